#!/usr/bin/env python3
"""Kernel experiments on the GPU box: times the k_corr2 variants (GPSACQ_KVAR, GPSACQ_CPW) against k_corr and
checks their cells against k_corr's.  tools/kvar_exp.py [variants...]   (needs an MI355X)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(%r, "gnss-gps-sdr_amd", "python"))
import torch, gpsacq
fc, fs, mfo = 4.092e6, 5.456e6, 5000.0
buf = open(os.path.join(%r, "tests", "golden", "synth_nott_fs5456.bin"), "rb").read()[:40 * 5120]
out = {}
with gpsacq.Engine(fc, fs, mfo) as eng:
    cells, peaks = eng.search(buf)
    ref = os.environ.get("KV_REF")
    if os.environ.get("KV_SAVE"):
        np.save(os.environ["KV_SAVE"], cells)
    elif ref:
        r = np.load(ref)
        out["max_rel_pwr"] = float(np.max(np.abs(cells["max_pwr"] / r["max_pwr"] - 1)))
        out["max_rel_tot"] = float(np.max(np.abs(cells["tot_pwr"] / r["tot_pwr"] - 1)))
        out["argmax_mismatch"] = int((cells["max_i"] != r["max_i"]).sum())
        out["bit_exact"] = bool(np.array_equal(cells, r))
    nblk = int(os.environ.get("KV_BLOCKS", "4096"))
    d_bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")
    d_peaks = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
    for _ in range(2):
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
    ms = []
    for _ in range(6):
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
        ms.append(eng.last_timing()["ms_correlate"])
    out["ms"] = ms
    out["ms_avg"] = sum(ms) / len(ms)
    out["mcells_s"] = nblk * eng.num_doppler / out["ms_avg"] / 1e3
print("KVOUT " + json.dumps(out))
''' % (ROOT, ROOT)


def run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("KVOUT "):
            return json.loads(line[6:]), r.stderr
    return {"error": (r.stdout + r.stderr)[-800:]}, r.stderr


def main():
    ref = "/tmp/kv_ref.npy"
    base, _ = run({"KV_SAVE": ref})
    print("k_corr (baseline)        ", json.dumps(base), flush=True)
    variants = [a for a in sys.argv[1:]] or ["1", "2", "3", "4", "5", "6", "7", "8"]
    for v in variants:
        kv, _, cpw = v.partition(":")
        env = {"GPSACQ_KVAR": kv, "KV_REF": ref}
        if kv.startswith("abl"):  # ablation build of the library (tools/ablate.sh), kernel variant 2
            env = {"GPSACQ_KVAR": "2", "KV_REF": ref, "GPSACQ_LIB": os.path.join(ROOT, "build", "abl", "libgpsacq_%s.so" % kv)}
            kv = "2"
        if cpw:
            env["GPSACQ_CPW"] = cpw
        if int(kv) >= 20 and int(kv) < 30:
            env["GPSACQ_PROF"] = "1"
            env["KV_BLOCKS"] = "1024"
        out, err = run(env)
        print("KVAR %-6s" % v, json.dumps(out), flush=True)
        if 20 <= int(kv) < 30:
            print("\n".join([l for l in err.splitlines() if "profile" in l][-2:]), flush=True)


if __name__ == "__main__":
    main()
