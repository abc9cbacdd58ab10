"""Synthetic-input generators and the multi-GPU sharding helpers (CPU; the N > 1 path runs under gloo)."""
import os
import struct
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from godot_whisper_amd import shard, synth
from oracle import port


def test_slaney_formula_reproduces_the_reference_filterbank():
    # the 128-bin bank (large-v3 shape) has no fixture; the generator is validated on the 80-bin one
    ref = synth.mel_filters(80)
    mine = synth.slaney_mel_filters(80)
    assert ref.shape == mine.shape == (80, 201)
    np.testing.assert_allclose(mine, ref, atol=2e-6)
    assert synth.mel_filters(128).shape == (128, 201)


def test_model_writer_layout():
    mb = synth.make_model("micro.en", seed=3)
    assert struct.unpack_from("<I", mb, 0)[0] == 0x67676D6C
    hp = struct.unpack_from("<11i", mb, 4)
    assert hp == (51864, 1500, 128, 2, 2, 448, 128, 2, 3, 80, 1)
    assert synth.make_model("micro.en", seed=3) == mb                      # deterministic
    assert synth.make_model("micro.en", seed=4) != mb
    n_tensors = len(synth.tensor_specs(hp[:10]))
    assert n_tensors == 7 + 15 * 2 + 4 + 24 * 3                            # W/whisper.cpp:1304 slot count
    if port.available():
        ps = port.PortSide(mb)
        assert (ps.NV, ps.S, ps.L) == (51864, 128, 3)
        ps.close()


def test_pcm_generator():
    a = synth.make_pcm(2.0, seed=1); b = synth.make_pcm(2.0, seed=1); c = synth.make_pcm(2.0, seed=2)
    assert a.dtype == np.float32 and a.size == 32000 and np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.abs(a).max() <= 1.0
    g = synth.make_pcm(3.0, seed=1, gate=True)
    assert np.all(g[32000:] == 0.0) and np.any(g[:32000] != 0.0)


def test_chunk_partition_is_a_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(c for r in range(world) for c in shard.chunks_for_rank(64, r, world))
        assert seen == list(range(64))
        assert all(len(shard.chunks_for_rank(64, r, world)) == 64 // world for r in range(world))
    assert shard.chunks_for_rank(5, 1, 2) == [1, 3]


WORKER = textwrap.dedent("""
    import os, sys, hashlib
    sys.path.insert(0, {root!r})
    import __graft_entry__ as e
    e.load_package()
    import torch, torch.distributed as dist
    from godot_whisper_amd import shard, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = synth.make_model("micro.en", seed=1234) if rank == 0 else None
    got = shard.broadcast_model(model, rank, world, dist, torch.device("cpu"))
    want = hashlib.sha256(synth.make_model("micro.en", seed=1234)).hexdigest()
    assert hashlib.sha256(got).hexdigest() == want, "broadcast corrupted the model image"
    mine = {{c: (c * c, rank) for c in shard.chunks_for_rank(6, rank, world)}}
    merged = shard.gather_results(mine, world, dist)
    assert list(merged) == list(range(6)) and all(merged[c][0] == c * c and merged[c][1] == c % world for c in merged)
    dist.barrier(); dist.destroy_process_group()
    print("RANK_OK", rank)
""")


def test_two_process_broadcast_and_gather_over_gloo(tmp_path):
    root = str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o


@pytest.mark.parametrize("qtype", ["q5_1", "q4_0", "q8_0"])
def test_quantiser_is_byte_identical_to_the_reference_tool(qtype, tmp_path):
    """synth.quantize_model regenerates quantised models on the GPU box; here (build container) it is pinned against
    the reference's own `quantize` tool (W/examples/quantize), compiled by oracle/Makefile."""
    import pathlib
    tool = pathlib.Path(__file__).resolve().parent.parent / "oracle" / "_ref" / "quantize"
    if not tool.exists():
        pytest.skip("reference quantize tool not built")
    mb = synth.make_model("micro.en", seed=5)
    (tmp_path / "m.bin").write_bytes(mb)
    r = subprocess.run([str(tool), str(tmp_path / "m.bin"), str(tmp_path / "o.bin"), qtype], capture_output=True)
    assert r.returncode == 0
    assert (tmp_path / "o.bin").read_bytes() == synth.quantize_model(mb, qtype)


@pytest.mark.parametrize("qtype", [None, "q5_1"])
def test_header_image_reproduces_the_arena_layout(qtype):
    """Multi-GPU load: the other ranks get a payload-less header image and the packed device arena.  Without a
    GPU: the image parses to the same model (hyper-parameters, vocabulary, special tokens) and yields the same arena plan —
    byte sizes of the arena and of its matrices, quantisation kind — as the full file, so rank 0's arena fits every rank's."""
    import ctypes as C
    from godot_whisper_amd import runtime
    lib = runtime.load_library(); runtime.silence_logs(lib)
    model = synth.make_model("micro", seed=9)
    if qtype:
        model = synth.quantize_model(model, qtype)
    buf = C.create_string_buffer(model, len(model))
    n = lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model), None, 0)
    assert 0 < n < len(model) // 4
    hdr = C.create_string_buffer(n)
    assert lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model), C.cast(hdr, C.c_void_p), n) == n
    full = lib.wmi_init_host_only(C.cast(buf, C.c_void_p), len(model))
    img = lib.wmi_init_host_only(C.cast(hdr, C.c_void_p), n)
    assert full and img
    try:
        for which in (0, 1, 2):
            assert lib.wmi_weights_bytes(full, which) == lib.wmi_weights_bytes(img, which) and (which == 2 or lib.wmi_weights_bytes(full, which) > 0)
        assert lib.wmi_weights_bytes(full, 2) == (7 if qtype else 0)
        for fn in ("whisper_n_vocab", "whisper_n_audio_ctx", "whisper_model_n_text_layer", "whisper_token_eot", "whisper_token_beg", "whisper_is_multilingual", "whisper_model_ftype"):
            assert getattr(lib, fn)(full) == getattr(lib, fn)(img), fn
        assert lib.whisper_token_to_str(full, 1234) == lib.whisper_token_to_str(img, 1234)
        # an image is not a model: truncated or re-exported images are rejected
        assert lib.wmi_model_header(C.cast(hdr, C.c_void_p), n, None, 0) == 0
        assert not lib.wmi_init_host_only(C.cast(hdr, C.c_void_p), n - 5)
    finally:
        lib.whisper_free(full); lib.whisper_free(img)
