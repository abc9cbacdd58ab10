"""Body / boundary split of the greedy decode step from in-kernel time stamps (wmi_step_stamps), next to the whole-step chain
(wmi_bench_kernel 20) and the host-paced step (decode_ms_per_token of a transcription).  SHAPE=base.en by default;
WMI_LIB_PATH selects an A/B build of the library."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import __graft_entry__ as entry
entry.load_package()
from godot_whisper_amd import host, runtime, synth
lib = runtime.require_gpu(); runtime.silence_logs(lib)
shape = os.environ.get("SHAPE", "base.en")
model = synth.make_model(shape, seed=1234)
node = host.SpeechToText(lib); node.set_language_model(model)
pcm = synth.make_pcm(30.0, seed=1234)
for _ in range(80 if os.environ.get("LONG") else 6):
    node.transcribe(pcm, "", 0)
L = {"tiny.en": 4, "base.en": 6, "small": 12}.get(shape, 6)
names_layer = ["LN+qkv", "sa+out", "xattn", "comb+co", "LN+fc1", "fc2"]


def stamps(chained):
    cap = 256
    buf = (C.c_double * (6 * cap))()
    n = lib.wmi_step_stamps(node.ctx, buf, cap, chained)
    rows = [(buf[6 * i], buf[6 * i + 1], buf[6 * i + 2], int(buf[6 * i + 3]), buf[6 * i + 4], buf[6 * i + 5]) for i in range(max(n, 0))]
    return [r for r in rows if r[3] > 0]


for chained in (1, 0):
    rows = stamps(chained)
    if not rows:
        print("no stamps (chained=%d)" % chained); continue
    names = ([] if chained else ["embed"]) + [f"{nm}.{l}" for l in range(L) for nm in names_layer] + ["logits", "f.stats", "f.pick"]
    if len(names) == len(rows) + 1:
        names.remove("f.stats")                  # statistics pass folded into the vocabulary projection's epilogue
    if len(names) != len(rows):
        names = [f"k{i}" for i in range(len(rows))]
    print(f"--- chained={chained}: {len(rows)} stamped launches; step span {rows[-1][2] - rows[0][0]:.2f} us")
    agg = {}
    prev_end = None
    for nm, (s0, s1, e1, cnt, m1, m2) in zip(names, rows):
        gap = (s0 - prev_end) if prev_end is not None else 0.0
        key = nm.split(".")[0] if "." in nm and nm.split(".")[0] in names_layer else nm
        a = agg.setdefault(key, [0.0, 0.0, 0.0, 0, 0.0, 0.0])
        a[0] += e1 - s0; a[1] += gap; a[2] += s1 - s0; a[3] += 1
        if m1 == -2.0: a[5] += m2                      # effective shader clock in MHz (k_xattn_fused)
        elif m1 >= 0: a[4] += m1 - s0
        if m1 >= 0 and m2 >= 0: a[5] += m2 - s0
        if os.environ.get("VERBOSE"):
            print(f"  {nm:10s} start {s0:8.2f}  body {e1 - s0:5.2f}  wave-start spread {s1 - s0:5.2f}  gap before {gap:5.2f}  waves {cnt}")
        prev_end = e1
    tb = tg = 0.0
    for key, (b, g, sp, c, m1, m2) in agg.items():
        print(f"  {key:8s} x{c}: body {b / c:5.2f} us  gap-before {g / c:5.2f} us  start-spread {sp / c:4.2f}  row-ready +{m1 / c:4.2f}  tile-reduced +{m2 / c:4.2f}   (sum body {b:6.1f}, gaps {g:6.1f})")
        tb += b; tg += g
    print(f"  total body {tb:.1f} us + gaps {tg:.1f} us")

lib.wmi_bench_kernel.restype = C.c_double
libc = C.CDLL(None)
libc.setenv(b"WMI_STEP_MASK", b"0x1ff", 1)
print("whole step chain (graph, non-chained form): %.1f us" % lib.wmi_bench_kernel(node.ctx, 20, 200))
# per-kind chains with several copies per graph (a 6-launch graph is bounded by the replay itself)
if os.environ.get("KINDS"):
    kinds = [("qkv (LN)", 2, 6), ("self-attn + out", 4, 6), ("cross-attention (fused)", 8, 6), ("combine + cross out", 16, 6),
             ("mlp.0 (LN, GELU)", 32, 6), ("mlp.2 (K = 4S)", 64, 6), ("logits", 128, 1), ("filters (2 kernels)", 256, 2)]
    for reps in (1, 8):
        libc.setenv(b"WMI_CHAIN_REPS", str(reps).encode(), 1)
        for name, m, n in kinds:
            libc.setenv(b"WMI_STEP_MASK", str(m).encode(), 1)
            t = lib.wmi_bench_kernel(node.ctx, 20, 96)
            print("reps/graph %d  %-24s %6.1f us per step = %5.2f us per launch" % (reps, name, t, t / n))
    libc.unsetenv(b"WMI_CHAIN_REPS")
# host-paced: decode time per token inside whisper_full
t6 = (C.c_int64 * 6)(); n5 = (C.c_int32 * 5)()
lib.whisper_reset_timings(node.ctx)
for _ in range(100):
    node.transcribe(pcm, "", 0)
lib.wmi_get_timings(node.ctx, t6, n5)
print("timings us:", list(t6), "counts:", list(n5))
