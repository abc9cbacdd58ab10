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
    # normalisation of the MFMA-busy counter, measured rather than assumed: busy / gui-active of the MFMA-only calibration kernel
    # (scratch/lab/mfma_peak.hip k_peak at 2 workgroups per CU: the matrix pipes issue back to back, busy = 1 by construction).
    # Round 2 divided by "1024 SIMDs x gui-active / 8", which gave 15.5 % for a kernel whose FLOP rate alone implies 20 %.
    norm = None
    calib = src / "pmc_mfma_calib_counter_collection.csv"
    if calib.exists():
        c = per_kernel(calib, {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
        ratios = [e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["GRBM_GUI_ACTIVE"] for k2, e in c.items() if "k_peak" in k2 and e.get("GRBM_GUI_ACTIVE")]
        if ratios:
            norm = max(ratios)                       # the densest variant (16 accumulators, 2 workgroups per CU)
    summary = {}
    if norm:
        summary["_mfma_busy_calibration"] = {"launches": 0, "busy_per_gui_active_at_100_percent": norm,
                                             "source": "scratch/lab/mfma_peak.hip k_peak under the same counter group (profiles/collect.sh)"}
    for k in sorted(set(fetch) | set(write) | set(mfma)):
        e = {"launches": (fetch.get(k) or write.get(k) or mfma.get(k))["launches"]}
        if k in fetch:
            e["fetch_bytes_raw"] = fetch[k]["FETCH_SIZE"] * 1024
            e["fetch_bytes_x2_corrected"] = 2 * e["fetch_bytes_raw"]
        if k in write:
            e["write_bytes_raw"] = write[k]["WRITE_SIZE"] * 1024
        if k in mfma and mfma[k].get("GRBM_GUI_ACTIVE"):
            # fraction of the calibration kernel's busy / gui-active ratio (the MFMA-only loop = 1.0); without a calibration run the
            # round-2 guess (summed over 1024 SIMDs, gui-active over 8 XCDs) is kept and flagged
            r = mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / mfma[k]["GRBM_GUI_ACTIVE"]
            if norm:
                e["mfma_busy_frac"] = r / norm
            else:
                e["mfma_busy_frac_uncalibrated"] = r / (1024.0 / 8.0)
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
