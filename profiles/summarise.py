#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of profiles/collect.sh into the small tracked summaries under profiles/.

    python profiles/summarise.py r01c

Writes profiles/<tag>_kernel_stats.csv (copy of the --stats table), profiles/<tag>_pmc_summary.json
(per-kernel average of FETCH_SIZE / WRITE_SIZE in bytes per launch and the MFMA-busy fraction) and prints the
figures used for `roofline.traffic` in bench.py.

Counter handling follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-byte requests of a wide coalesced stream as 64 bytes, i.e. reads HALF the bytes of a 16 B/lane
streaming read — the summary therefore lists both the raw and the x2-corrected read traffic.  WRITE_SIZE is
uncalibrated (reported raw)."""
import csv
import json
import pathlib
import re
import shutil
import sys
from collections import defaultdict

ROOT = pathlib.Path(__file__).resolve().parent.parent


def short(name: str) -> str:
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][-40:]


def per_kernel(path, counters):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] in counters:
                # key kernels additionally by grid size so that the k_gemv / k_gemm instances separate
                key = f'{short(r["Kernel_Name"])} grid={r["Grid_Size"]}'
                acc[key][(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    out = {}
    for k, d in acc.items():
        per_counter = defaultdict(list)
        for (c, _), vals in d.items():
            per_counter[c].append(sum(vals))          # sum over XCDs/instances of one dispatch
        out[k] = {c: sum(v) / len(v) for c, v in per_counter.items()}
        out[k]["launches"] = max(len(v) for v in per_counter.values())
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles"
    shutil.copy(src / "trace_kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")
    for line in (src / "bench_trace.log").read_text().splitlines():
        if line.startswith("{"):
            (dst / f"{tag}_bench_under_rocprof.json").write_text(line + "\n")
    fetch = per_kernel(src / "pmc_fetch_counter_collection.csv", {"FETCH_SIZE"})
    write = per_kernel(src / "pmc_write_counter_collection.csv", {"WRITE_SIZE"})
    mfma = per_kernel(src / "pmc_mfma_counter_collection.csv", {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"})
    summary = {}
    for k in sorted(set(fetch) | set(write) | set(mfma)):
        e = {"launches": (fetch.get(k) or write.get(k) or mfma.get(k))["launches"]}
        if k in fetch:
            e["fetch_bytes_raw"] = fetch[k]["FETCH_SIZE"] * 1024
            e["fetch_bytes_x2_corrected"] = 2 * e["fetch_bytes_raw"]
        if k in write:
            e["write_bytes_raw"] = write[k]["WRITE_SIZE"] * 1024
        if k in mfma and mfma[k].get("GRBM_GUI_ACTIVE"):
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            e["mfma_busy_frac"] = mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (mfma[k]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            e["mfma_busy_cycles"] = mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
            e["gui_active_cycles_sum"] = mfma[k]["GRBM_GUI_ACTIVE"]
        summary[k] = e
    (dst / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
    for k, e in summary.items():
        if "fetch_bytes_raw" in e and e["fetch_bytes_raw"] > 1e6:
            print(f'{k:48s} launches {e["launches"]:4d} fetch raw {e["fetch_bytes_raw"]/1e6:8.2f} MB (x2: {e["fetch_bytes_x2_corrected"]/1e6:8.2f}) '
                  f'write {e.get("write_bytes_raw", 0)/1e6:7.2f} MB  mfma_busy {e.get("mfma_busy_cycles", 0):.3g} / gui {e.get("gui_active_cycles_sum", 0):.3g}')


if __name__ == "__main__":
    main()
