"""Multi-GPU sharding of the hot path: one process per GPU, independent 30 s chunks,
weights replicated by ONE broadcast (RCCL over xGMI when the tensors live on GPUs; gloo in CPU tests),
no collective inside the hot loop, results gathered on the host.

The ownership split mirrors the reference's whisper_full_parallel (shared read-only model, one state per
worker, results concatenated in chunk order; W/whisper.cpp:5817-5913)."""
from __future__ import annotations

import numpy as np


def chunks_for_rank(n_chunks: int, rank: int, world: int) -> list[int]:
    """chunk c -> rank c mod world."""
    return list(range(rank, n_chunks, world))


def broadcast_model(model_bytes: bytes | None, rank: int, world: int, dist, device) -> bytes:
    """Rank 0 holds the ggml model image; every rank returns the same bytes.  Two collectives in total:
    an 8-byte size and the image itself (base.en f16: 148 MB)."""
    import torch
    if world == 1:
        assert model_bytes is not None
        return model_bytes
    size = torch.tensor([len(model_bytes) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src=0)
    n = int(size.item())
    if rank == 0:
        image = torch.frombuffer(bytearray(model_bytes), dtype=torch.uint8).to(device)
    else:
        image = torch.empty(n, dtype=torch.uint8, device=device)
    dist.broadcast(image, src=0)
    return image.cpu().numpy().tobytes()


class _DevMem:
    """A device allocation that is not torch's (the context's weight arena) as a zero-copy torch tensor: torch.as_tensor reads
    the __cuda_array_interface__ protocol on ROCm builds as well."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def load_replicated(lib, model_bytes: bytes | None, rank: int, world: int, dist, device_index: int, torch_device=None,
                    force_collectives: bool = False):
    """Every rank ends up with a context holding the same weights; rank 0 is the only one that parses payloads.
    Collectives: the header image (~1 MB: hyper-parameters, mel filters, vocabulary, tensor directory) and ONE broadcast of
    the packed device arena straight into every rank's arena allocation (RCCL over xGMI: base.en 148 MB, large-v3 q5_1 1.18 GB)
    — no D2H, no re-parse, no re-quantisation on the other ranks.  Returns (ctx, seconds spent in the arena broadcast).
    force_collectives: issue the collectives at world size 1 too (tests: the RCCL calls on the non-torch arena allocation can
    be executed on a one-GPU box, where RCCL admits one rank per device)."""
    import ctypes as C
    import time

    import numpy as np
    import torch
    if world == 1 and not force_collectives:
        buf = C.create_string_buffer(model_bytes, len(model_bytes))
        ctx = lib.wmi_init_from_buffer_on_device(C.cast(buf, C.c_void_p), len(model_bytes), device_index)
        assert ctx, "model load failed"
        return ctx, 0.0
    dev = torch_device if torch_device is not None else torch.device("cuda", device_index)
    if rank == 0:
        buf = C.create_string_buffer(model_bytes, len(model_bytes))
        ctx = lib.wmi_init_from_buffer_on_device(C.cast(buf, C.c_void_p), len(model_bytes), device_index)
        assert ctx, "model load failed"
        n = lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model_bytes), None, 0)
        hdr = np.empty(n, np.uint8)
        assert lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model_bytes), hdr.ctypes.data_as(C.c_void_p), n) == n
        del buf
        meta = torch.tensor([n, lib.wmi_weights_bytes(ctx, 0)], dtype=torch.int64, device=dev)
    else:
        ctx = None; hdr = None
        meta = torch.zeros(2, dtype=torch.int64, device=dev)
    dist.broadcast(meta, src=0)
    n_hdr, n_arena = int(meta[0].item()), int(meta[1].item())
    h = torch.from_numpy(hdr).to(dev) if rank == 0 else torch.empty(n_hdr, dtype=torch.uint8, device=dev)
    dist.broadcast(h, src=0)
    if rank != 0:
        hb = h.cpu().numpy().tobytes()
        buf = C.create_string_buffer(hb, len(hb))
        ctx = lib.wmi_init_from_header(C.cast(buf, C.c_void_p), len(hb), device_index)      # arena zeroed and laid out, weights pending
        assert ctx, "header image rejected"
        assert lib.wmi_weights_pending(ctx) == 1
        assert lib.wmi_weights_bytes(ctx, 0) == n_arena, "arena layout differs between ranks"
    arena = torch.as_tensor(_DevMem(int(lib.wmi_arena_ptr(ctx)), n_arena), device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dist.broadcast(arena, src=0)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if rank != 0:
        assert lib.wmi_arena_commit(ctx) == 0, "arena commit failed"      # the context refuses to compute before this
    return ctx, dt


def gather_results(local: dict, world: int, dist, force_collectives: bool = False) -> dict:
    """local: {chunk_id: result}; returns the merged {chunk_id: result} on every rank (host-side gather)."""
    if world == 1 and not force_collectives:
        return dict(local)
    parts = [None] * world
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    return dict(sorted(merged.items()))
