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


def gather_results(local: dict, world: int, dist) -> dict:
    """local: {chunk_id: result}; returns the merged {chunk_id: result} on every rank (host-side gather)."""
    if world == 1:
        return dict(local)
    parts = [None] * world
    dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    return dict(sorted(merged.items()))
