#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory (rocprofv3 CSVs) into profiles/<tag>_summary.md
and copy the kernel-stats CSV.  Usage: tools/summarize_prof.py gpurun_out/prof_<tag> <tag> [note]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
lines = [f"# rocprofv3 summary {tag}", "", note, ""]

ks = os.path.join(src, "trace", "t_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline` (acq kernels)", "",
              "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks)):
        if "acq::" in r["Name"]:
            lines.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                         f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines.append("")
bl = os.path.join(src, "bench_traced.log")
if os.path.exists(bl):
    for l in open(bl):
        if l.startswith("{"):
            j = json.loads(l)
            lines += ["bench line of the traced run: " + json.dumps({k: j[k] for k in ("value", "ms_per_step", "roofline", "stage_ms") if k in j}), ""]

agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "acq::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
            agg[(r["Kernel_Name"][:48], "_vgpr")] = [float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"])]
            agg[(r["Kernel_Name"][:48], "_grid")] = [float(r["Grid_Size"])]
if agg:
    lines += ["## PMC passes (`rocprofv3 --pmc <set>`, separate runs, `bench.py --steps 2 --warmup 1 --blocks 1024`), per-launch averages", ""]
    kernels = sorted({k for k, _ in agg})
    for k in kernels:
        lines.append(f"### `{k}`")
        d = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
        for c in sorted(d):
            lines.append(f"- {c}: {d[c]:.6g}")
        if "FETCH_SIZE" in d:
            lines.append(f"- HBM read bytes (FETCH_SIZE KiB x 1024 x 2, gfx950 correction of MI355X_MICROARCH.md section HBM): {d['FETCH_SIZE']*1024*2:.4g}")
        if "WRITE_SIZE" in d:
            lines.append(f"- HBM write bytes (WRITE_SIZE KiB x 1024, uncalibrated): {d['WRITE_SIZE']*1024:.4g}")
        if "TCC_HIT_sum" in d:
            lines.append(f"- L2 hit rate: {d['TCC_HIT_sum']/(d['TCC_HIT_sum']+d['TCC_MISS_sum']):.4f}")
        if "SQ_WAVE_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
            lines.append(f"- VALU-active share of wave cycles: {d['SQ_ACTIVE_INST_VALU']/d['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_LDS_IDX_ACTIVE" in d:
            lines.append(f"- LDS bank-conflict share of LDS cycles: {d['SQ_LDS_BANK_CONFLICT']/max(d['SQ_LDS_IDX_ACTIVE'],1):.3f}")
        lines.append("")
# per-cell HBM traffic of the dominant kernel for bench.py's roofline.traffic
try:
    d = {c: sum(v) / len(v) for (kk, c), v in agg.items() if "k_corr" in kk}
    cells = d["_grid"] / 256.0 if "_grid" in d else None  # grid = workgroups*256 threads; one workgroup per cell (padded)
    if cells and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        onchip = {}
        for key, name in (("SQ_INSTS_VALU", "valu_wave_instr_per_cell"), ("SQ_INSTS_LDS", "lds_wave_instr_per_cell"),
                          ("SQ_INSTS_VMEM_RD", "vmem_read_wave_instr_per_cell"), ("SQ_LDS_IDX_ACTIVE", "lds_active_cycles_per_cell"),
                          ("SQ_LDS_BANK_CONFLICT", "lds_conflict_cycles_per_cell")):
            if key in d:
                onchip[name] = d[key] / cells
        if "TCC_HIT_sum" in d:
            onchip["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in d:
            for key, name in (("SQ_ACTIVE_INST_VALU", "valu_share_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "issue_stall_share_of_wave_cycles"),
                              ("SQ_WAIT_ANY", "waitcnt_barrier_share_of_wave_cycles")):
                if key in d:
                    onchip[name] = d[key] / d["SQ_WAVE_CYCLES"]
        json.dump({"tag": tag, "kernel": "k_corr", "cells_per_launch_profiled": cells,
                   "hbm_read_bytes_per_cell": d["FETCH_SIZE"] * 1024 * 2 / cells,
                   "hbm_write_bytes_per_cell": d["WRITE_SIZE"] * 1024 / cells,
                   "onchip_counters": onchip,
                   "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE KiB x 2 (gfx950 correction, "
                             "MI355X_MICROARCH.md section HBM); WRITE_SIZE uncalibrated"},
                  open(os.path.join(dst, "traffic.json"), "w"), indent=1)
except Exception as ex:  # noqa
    print("traffic.json not written:", ex)
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
