"""-m gpu: the N > 1 model load on real device memory (the reference's whisper_full_parallel ownership split
W/whisper.cpp:5837-5913 with one process per GPU).

Two processes share the one GPU of the test box (gloo rendezvous on 127.0.0.1 — RCCL refuses two ranks on one device), and run
shard.load_replicated exactly as bench.py --gpus N does: rank 0 parses the ggml file and builds its device arena, rank 1 gets the
~1 MB header image, lays out the same arena (weights pending: every compute call must refuse), receives the packed arena by ONE
broadcast straight into its own arena allocation (zero-copy through __cuda_array_interface__), commits it — and must then
transcribe its chunks exactly as a context loaded from the file does."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    sys.path.insert(0, {root!r})
    import __graft_entry__ as entry
    entry.load_package()
    import torch, torch.distributed as dist
    from godot_whisper_amd import host, runtime, shard, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = runtime.require_gpu(); runtime.silence_logs(lib)
    out = {{}}
    for shape, qtype in (("micro.en", None), ("micro", "q5_1")):
        model = None
        if rank == 0:
            model = synth.make_model(shape, seed=77)
            if qtype:
                model = synth.quantize_model(model, qtype)
        ctx, t_bcast = shard.load_replicated(lib, model, rank, world, dist, 0, dev)
        assert ctx and lib.wmi_weights_pending(ctx) == 0
        node = host.SpeechToText(lib); node.ctx = ctx
        if shape == "micro": node.language = "de"
        res = {{}}
        for c in range(4):                                   # every rank transcribes ALL chunks here: the results must be identical
            p = node.full_params("", 0); p.temperature_inc = 0.0
            r = node.transcribe(synth.make_pcm(6.0 + 5 * c, seed=300 + c), params=p)
            assert node.last_ret == 0
            res[c] = [[int(t["id"]), int(t["tid"]), float(t["p"]), float(t["plog"]), int(t["t0"]), int(t["t1"])] for t in r[1:]]
        # and the sharded run: chunk c -> rank c mod world, host-side gather
        mine = {{c: res[c] for c in shard.chunks_for_rank(4, rank, world)}}
        merged = shard.gather_results(mine, world, dist)
        assert list(merged) == [0, 1, 2, 3]
        out[shape + (":" + qtype if qtype else "")] = {{"all": res, "merged": {{str(k): v for k, v in merged.items()}}, "bcast_ms": 1e3 * t_bcast,
                                                     "arena_bytes": int(lib.wmi_weights_bytes(ctx, 0))}}
        node.close()
    dist.barrier(); dist.destroy_process_group()
    print("RESULT" + json.dumps(out))
""")

PENDING = textwrap.dedent("""
    import ctypes as C, sys
    sys.path.insert(0, {root!r})
    import __graft_entry__ as entry
    entry.load_package()
    import numpy as np
    from godot_whisper_amd import host, runtime, synth
    lib = runtime.require_gpu(); runtime.silence_logs(lib)
    model = synth.make_model("micro.en", seed=77)
    buf = C.create_string_buffer(model, len(model))
    n = lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model), None, 0)
    hdr = C.create_string_buffer(n)
    assert lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model), C.cast(hdr, C.c_void_p), n) == n
    # the header image is not accepted by the file loaders ...
    assert not lib.wmi_init_from_buffer_on_device(C.cast(hdr, C.c_void_p), n, 0)
    # ... and a context made from it refuses to compute until the arena is committed
    ctx = lib.wmi_init_from_header(C.cast(hdr, C.c_void_p), n, 0)
    assert ctx and lib.wmi_weights_pending(ctx) == 1
    node = host.SpeechToText(lib); node.ctx = ctx
    node.transcribe(synth.make_pcm(4.0, seed=1), params=node.full_params("", 0))
    assert node.last_ret != 0, "a context with pending weights transcribed"
    node.close()
    print("PENDING_OK")
""")


def _run_world(tmp_path, script_text, world, port):
    script = tmp_path / "worker.py"
    script.write_text(script_text.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs


def test_second_rank_receives_the_arena_and_transcribes_like_rank_0(tmp_path):
    outs = _run_world(tmp_path, WORKER, 2, 29633)
    res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT")][-1][len("RESULT"):]) for o in outs]
    for key in res[0]:
        a, b = res[0][key], res[1][key]
        assert a["arena_bytes"] == b["arena_bytes"] > 0
        assert any(len(v) for v in a["all"].values()), key               # something was decoded
        assert a["all"] == b["all"], key                                 # rank 1 (arena by broadcast) == rank 0 (parsed the file), bit for bit
        assert a["merged"] == b["merged"] == {k: v for k, v in a["all"].items()}, key      # the sharded run gathers to the same thing


def test_header_image_context_refuses_to_compute_until_committed(tmp_path):
    script = tmp_path / "pending.py"
    script.write_text(PENDING.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PENDING_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


NCCL_WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    sys.path.insert(0, {root!r})
    import __graft_entry__ as entry
    entry.load_package()
    import numpy as np
    import torch, torch.distributed as dist
    from godot_whisper_amd import host, runtime, shard, synth
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # exactly bench.py's call for N > 1
    assert dist.get_backend() == "nccl"
    lib = runtime.require_gpu(); runtime.silence_logs(lib)
    out = {{}}
    for shape, qtype in (("micro.en", None), ("micro", "q5_1")):
        model = synth.make_model(shape, seed=77)
        if qtype:
            model = synth.quantize_model(model, qtype)
        # rank 0's side of load_replicated with every collective issued: meta, header image, and the arena broadcast on the tensor
        # that wraps the context's own hipMalloc (not a torch allocation)
        ctx, t_bcast = shard.load_replicated(lib, model, 0, 1, dist, 0, dev, force_collectives=True)
        assert ctx and lib.wmi_weights_pending(ctx) == 0
        n_arena = int(lib.wmi_weights_bytes(ctx, 0))
        # a receiving rank's side, as far as one rank can play it: context from the header image, its pending arena wrapped the same
        # way, handed to the nccl backend as the broadcast's destination buffer, filled through the torch view, committed
        buf = C.create_string_buffer(model, len(model))
        n = lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model), None, 0)
        hdr = C.create_string_buffer(n)
        assert lib.wmi_model_header(C.cast(buf, C.c_void_p), len(model), C.cast(hdr, C.c_void_p), n) == n
        ctx2 = lib.wmi_init_from_header(C.cast(hdr, C.c_void_p), n, 0)
        assert ctx2 and lib.wmi_weights_pending(ctx2) == 1 and lib.wmi_weights_bytes(ctx2, 0) == n_arena
        a1 = torch.as_tensor(shard._DevMem(int(lib.wmi_arena_ptr(ctx)), n_arena), device=dev)
        a2 = torch.as_tensor(shard._DevMem(int(lib.wmi_arena_ptr(ctx2)), n_arena), device=dev)
        assert a1.data_ptr() == int(lib.wmi_arena_ptr(ctx)) and a2.data_ptr() == int(lib.wmi_arena_ptr(ctx2))      # zero-copy views
        dist.broadcast(a2, src=0)                        # RCCL sees the foreign pointer as a collective buffer
        work = dist.broadcast(a1, src=0, async_op=True); work.wait()
        got = [torch.empty_like(a1)]
        dist.all_gather(got, a1)                         # a collective that MOVES the foreign buffer's bytes through RCCL
        assert torch.equal(got[0], a1)
        a2.copy_(got[0]); torch.cuda.synchronize()
        assert lib.wmi_arena_commit(ctx2) == 0
        res = []
        for c_ in (ctx, ctx2):
            node = host.SpeechToText(lib); node.ctx = c_
            if shape == "micro": node.language = "de"
            p = node.full_params("", 0); p.temperature_inc = 0.0
            r = node.transcribe(synth.make_pcm(9.0, seed=301), params=p)
            assert node.last_ret == 0
            res.append([[int(t["id"]), int(t["tid"]), float(t["p"]), float(t["plog"]), int(t["t0"]), int(t["t1"])] for t in r[1:]])
            node.close()
        assert res[0] == res[1] and len(res[0]) > 0
        merged = shard.gather_results({{0: res[0]}}, 1, dist, force_collectives=True)      # all_gather_object over nccl
        assert merged == {{0: res[0]}}
        out[shape + (":" + qtype if qtype else "")] = {{"arena_bytes": n_arena, "bcast_ms": 1e3 * t_bcast, "tokens": len(res[0])}}
    dist.barrier(); dist.destroy_process_group()
    print("RESULT" + json.dumps(out))
""")


def test_rccl_backend_handles_the_foreign_arena_and_the_object_gather(tmp_path):
    """De-risk of the first multi-GPU lease on the one GPU of the test box (RCCL admits one rank per device, so world size 1):
    the `nccl` backend — RCCL on ROCm — is initialised exactly as bench.py --gpus N does and EXECUTES every collective
    shard.load_replicated / gather_results issue, on the tensors they issue them on: dist.broadcast of the zero-copy view of the
    context's own hipMalloc'd weight arena (as source and as destination buffer), an all_gather that moves that buffer's bytes
    through RCCL, dist.all_gather_object for the host-side result gather.  The gloo tests above cover the two-rank data flow
    (W/whisper.cpp:5837-5913 ownership split); this one covers the backend the 8-GPU run uses."""
    outs = _run_world(tmp_path, NCCL_WORKER, 1, 29641)
    res = json.loads([l for l in outs[0].splitlines() if l.startswith("RESULT")][-1][len("RESULT"):])
    assert set(res) == {"micro.en", "micro:q5_1"}
    for v in res.values():
        assert v["arena_bytes"] > 0 and v["tokens"] > 0
