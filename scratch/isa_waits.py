"""Print, per kernel of a gfx950 .s file, the order of vector loads and vmcnt waits (compressed): a `vmcnt(0)` between
loads that should be in flight together is a dependent round trip the source did not intend."""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name = None; ev = []
def flush():
    if name and pat in name and ev:
        out = []; last = None; n = 0
        for e in ev:
            if e == last: n += 1
            else:
                if last: out.append(f"{last}x{n}" if n > 1 else last)
                last, n = e, 1
        out.append(f"{last}x{n}" if n > 1 else last)
        print(name[:110]); print("   ", " ".join(out)[:1500])
for l in txt:
    m = re.match(r"^(_Z\w+):", l)
    if m: flush(); name = m.group(1); ev = []; continue
    l = l.strip()
    if l.startswith("global_load") or l.startswith("buffer_load"): ev.append("L" + ("4" if "dwordx4" in l else "2" if "dwordx2" in l else "h" if "short" in l else "1"))
    elif l.startswith("s_waitcnt") and "vmcnt" in l: ev.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
    elif l.startswith("s_barrier"): ev.append("|B|")
    elif l.startswith("global_store"): ev.append("S")
    elif l.startswith("s_endpgm"): ev.append("END")
flush()
