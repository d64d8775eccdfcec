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
    cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "bench.py --steps 5 --warmup 1 --no-cpu-baseline"
    lines += [f"## `rocprofv3 --kernel-trace --stats -- python {cmd}` (acq kernels; every launch of a kernel has the same size)", "",
              "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks)):
        if "acq::" in r["Name"]:
            lines.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                         f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines.append("")
cells_per_launch = None
clock_lines = []  # the box-independent figures (round 5): cycles one CU spends per cell, fraction of the peak at the clock held
for name, what in (("bench_traced.log", "traced"), ("bench_unprofiled.log", "un-profiled")):
    bl = os.path.join(src, name)
    if os.path.exists(bl):
        for l in open(bl):
            if l.startswith("{"):
                j = json.loads(l)
                cells_per_launch = j.get("roofline", {}).get("cells_per_launch", cells_per_launch)
                r = j.get("roofline", {})
                lines += [f"bench line of the {what} run of the same command: " + json.dumps(
                    {"value": j.get("value"), "ms_per_step": j.get("ms_per_step"), "stage_ms": j.get("stage_ms"),
                     "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "cells_per_launch", "flops_per_cell",
                                                        "sclk_mhz", "sclk_mhz_xcd_min", "sclk_mhz_xcd_max", "power_w", "cycles_per_cell_per_cu", "frac_at_clock")}}), ""]
                r = dict(r, compute_units=r.get("compute_units", 256))
                if r.get("sclk_mhz") and r.get("cycles_per_cell_per_cu"):
                    clock_lines.append(f"- {what} run, HIP events: kernel_ms {r['kernel_ms']:.3f} x sclk {r['sclk_mhz']:.0f} MHz (mean of the 8 XCDs' cycle counters over the timed steps, gpsacq_cycle_stamp_device: "
                                       f"{(r.get('clock_sampling') or {}).get('sclk_mhz_per_xcd')}; sysfs = XCD 0: {(r.get('clock_sampling') or {}).get('sclk_mhz_sysfs')} MHz, {r.get('power_w')} W) x {r.get('compute_units')} CUs / {r['cells_per_launch']} cells = **{r['cycles_per_cell_per_cu']:.0f} cycles per cell per CU**, "
                                       f"frac {r['frac']:.4f} of 157.3 TFLOP/s, **frac_at_clock {r['frac_at_clock']:.4f}**")
                    if what == "traced" and os.path.exists(ks):
                        for row in csv.DictReader(open(ks)):
                            if "k_corr" in row["Name"]:
                                avg_ms = float(row["AverageNs"]) / 1e6
                                cyc = avg_ms * 1e-3 * r["sclk_mhz"] * 1e6 * r.get("compute_units", 256) / r["cells_per_launch"]
                                clock_lines.append(f"- rocprofv3 trace of that run: k_corr avg {avg_ms:.3f} ms x the same sclk = **{cyc:.0f} cycles per cell per CU**")
                                break
agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "acq::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
            agg[(r["Kernel_Name"][:48], "_vgpr")] = [float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"])]
            agg[(r["Kernel_Name"][:48], "_grid")] = [float(r["Grid_Size"])]
if agg:
    lines += ["## PMC passes (`rocprofv3 --pmc <set>`, one run per set, the same command with --steps 2), per-launch averages", ""]
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
    cells = cells_per_launch or (d["_grid"] / 256.0 if "_grid" in d else None)  # the bench line's count; else grid / 256 (one workgroup per cell, padded)
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
                              ("SQ_WAIT_INST_LDS", "lds_issue_stall_share_of_wave_cycles"), ("SQ_WAIT_ANY", "waitcnt_barrier_share_of_wave_cycles")):
                if key in d:
                    onchip[name] = d[key] / d["SQ_WAVE_CYCLES"]
        if "TCP_TCC_READ_REQ_sum" in d:
            # one TCP -> TCC read request moves a 128-byte line on gfx950 (the same accounting MI355X_MICROARCH.md documents for FETCH_SIZE:
            # 128-byte requests tallied at 64).  Cross-check inside this profile: a cell's vector-memory read instructions are 16 bytes per
            # lane = 1 KiB = 8 lines per wave-instruction, and with ~0 % L1 reuse every one of them becomes requests
            onchip["l2_to_l1_read_requests_per_cell"] = d["TCP_TCC_READ_REQ_sum"] / cells
            onchip["l2_to_l1_bytes_per_cell"] = d["TCP_TCC_READ_REQ_sum"] * 128.0 / cells
            if "SQ_INSTS_VMEM_RD" in d:
                onchip["l2_to_l1_bytes_per_cell_from_vmem_instr"] = d["SQ_INSTS_VMEM_RD"] * 1024.0 / cells
                onchip["l1_lines_requested_over_lines_loaded"] = d["TCP_TCC_READ_REQ_sum"] / (d["SQ_INSTS_VMEM_RD"] * 8.0)  # ~1: no L1 reuse
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in d:  # (raw: the counter's access unit is not documented for gfx950 -- the line cross-check above is the L1-reuse evidence)
                onchip["tcp_total_cache_accesses_per_cell"] = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / cells
        if "GRBM_GUI_ACTIVE" in d:
            # GRBM_GUI_ACTIVE: shader-clock cycles the kernel kept the GPU busy, summed over the 8 XCDs -- a duration in CYCLES, no clock reading needed
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0 * 256.0 / cells
            onchip["cycles_per_cell_per_cu_grbm"] = cyc
            clock_lines.append(f"- PMC pass: GRBM_GUI_ACTIVE {d['GRBM_GUI_ACTIVE']:.6g} / 8 XCDs x 256 CUs / {cells:.0f} cells = **{cyc:.0f} cycles per cell per CU** (counted in cycles: independent of the box's clock)")
        # ties the counters to the kernel sources they were collected on: bench.py recomputes the hash and flags a mismatch
        # ("traffic_stale").  The summary is made in the authoring container from the GPU box's output of the SAME tree.
        import hashlib
        h = hashlib.sha256()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for name in ("acq_kernels.hip", "acq_phases.hpp", "acq_math.hpp"):
            h.update(open(os.path.join(root, "gnss-gps-sdr_amd", "csrc", name), "rb").read())
        sha_file = os.path.join(src, "kernel_source_sha.txt")  # written on the GPU box by tools/profile.sh
        sha = open(sha_file).read().strip() if os.path.exists(sha_file) else h.hexdigest()[:16]
        json.dump({"tag": tag, "kernel": "k_corr", "kernel_instance": "k_corr<22,3,2>", "kernel_source_sha": sha,
                   "cells_per_launch_profiled": cells,
                   "hbm_read_bytes_per_cell": d["FETCH_SIZE"] * 1024 * 2 / cells,
                   "hbm_write_bytes_per_cell": d["WRITE_SIZE"] * 1024 / cells,
                   "onchip_counters": onchip,
                   "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE KiB x 2 (gfx950 correction, "
                             "MI355X_MICROARCH.md section HBM); WRITE_SIZE uncalibrated"},
                  open(os.path.join(dst, "traffic.json"), "w"), indent=1)
except Exception as ex:  # noqa
    print("traffic.json not written:", ex)
if clock_lines:
    lines += ["## Box-independent roofline of k_corr (what makes this profile comparable with a bench line from another box)", ""] + clock_lines + [""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
