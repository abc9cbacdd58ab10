"""Scratch: the three encoder-attention forms (WMI_ATTN_FORM 0 = round-2 kernel, 1 = 32-row wavefronts + exact maximum first,
2 = 32-row wavefronts + running maximum) against the compiled reference: encoder output / cross K / logits error and
us per launch at one and eight chunks.  One subprocess per form (the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import __graft_entry__ as entry
    entry.load_package(); entry.load_oracle()
    from godot_whisper_amd import host, runtime, synth
    from oracle import reflib
    import stage_compare as sc
    shape = os.environ.get("SHAPE", "base.en")
    lib = runtime.require_gpu(); runtime.silence_logs(lib)
    mb = synth.make_model(shape, seed=1234); pcm = synth.make_pcm(30.0, seed=1234)
    if os.environ.get("NOREF") != "1":
        prod = sc.ProductSide(lib, mb); ref = sc.RefSide(reflib.lib(), mb); ref.n_threads = 32
        r = sc.compare_stages(prod, ref, pcm, n_steps=2, log=None)
        print("  errors vs reference: " + "  ".join("%s rms-rel %.2e max %.2e" % (k, r[k]["rms_rel"], r[k]["max_abs"])
                                                     for k in ("embd_enc", "cross_k", "cross_v", "logits_prompt", "logits_step1")))
        prod.close(); ref.close()
    node = host.SpeechToText(lib); node.set_language_model(mb)
    params = node.full_params("", 0); nb = 8
    pcms = [synth.make_pcm(30.0, seed=100 + i) for i in range(nb)]
    ptrs = (C.c_void_p * nb)(*[p.ctypes.data for p in pcms]); lens = (C.c_int * nb)(*[p.size for p in pcms])
    node.transcribe(pcms[0], "", 0)
    assert lib.wmi_full_batch(node.ctx, params, ptrs, lens, nb, 0) == 0
    lib.wmi_bench_kernel.restype = C.c_double
    print("  attention layer: 1 chunk %.2f us, 8 chunks %.2f us" % (lib.wmi_bench_kernel(node.ctx, 2, 100), lib.wmi_bench_kernel(node.ctx, 5, 50)))
    t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)()
    lib.whisper_reset_timings(node.ctx)
    for _ in range(20): node.transcribe(pcms[0], "", 0)
    lib.wmi_get_timings(node.ctx, t6, n5)
    print("  encode ms (1 chunk): %.4f" % (t6[1] / 1e3 / n5[0]))
    node.close()
else:
    for form in os.environ.get("FORMS", "0 1 2").split():
        for ks in os.environ.get("KSPLITS", "-1").split():
            for cfg in os.environ.get("CFGS", "0").split():
                env = dict(os.environ, WMI_ATTN_FORM=form, WMI_ATTN_KSPLIT=ks)
                if cfg != "0": env["WMI_ATTN_CFG"] = cfg
                print("WMI_ATTN_FORM=%s WMI_ATTN_KSPLIT=%s WMI_ATTN_CFG=%s" % (form, ks, cfg), flush=True)
                subprocess.run([sys.executable, __file__, "child"], env=env)
