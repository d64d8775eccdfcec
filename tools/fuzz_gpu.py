#!/usr/bin/env python3
"""tools/fuzz_gpu.py [first_seed] [n_seeds] (env FUZZ_FS=lo,hi | FUZZ_CLI=1 | FUZZ_PLUMBING=1) -- wide seeded sweep of the engine against the oracle on the GPU box (one-off
hunting tool; the bounded version lives in tests/test_gpu_extras.py::test_random_grid_configurations).  Per seed: a random
(fs in 1.2..20 MHz, IF, Doppler range, Doppler step) and one of the modes
  coherent | ref_quirks | non-coherent (plain / creep re-aligned) | Doppler window | default schedule with stride
on random capture bits; a handful of grid points of random tasks against the oracle's restatement.  Prints one line per seed
and a summary; exit 1 on any mismatch."""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpsacq  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

REL = 2e-5


def close(a, b, what):
    err = float(np.max(np.abs(np.asarray(a, np.float64) / np.asarray(b, np.float64) - 1.0)))
    if not err <= REL:
        raise AssertionError(f"{what}: rel err {err:.3g}")
    return err


def iq_case(seed, rng, fs, fc, max_fo, mode):
    """8-bit IQ capture (rtl-sdr uint8 or HackRF int8, random DC, random mixer): the fused 1-bit search against the numpy conversion
    + the C oracle; the float paths against the float64 restatements, on the bin grid or a sub-bin one."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from iq8_oracle import iq8_to_bits, iq8_to_real, iq8_to_complex, multibit_cells, complex_cells
    from make_golden import code_replica
    from oracle_lib import lib, _p
    signed = bool(rng.integers(0, 2))
    remove_dc = bool(rng.integers(0, 4))
    mix = float(rng.choice([0.0, fc, rng.uniform(-0.3, 0.3) * fs]))
    nblk = 3
    ns = nblk * 40960
    z = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) * 25.0 + (rng.uniform(-6, 6) + 1j * rng.uniform(-6, 6))
    raw = np.empty(2 * ns, np.uint8)
    if signed:
        raw[0::2] = np.clip(np.rint(z.real), -128, 127).astype(np.int8).view(np.uint8)
        raw[1::2] = np.clip(np.rint(z.imag), -128, 127).astype(np.int8).view(np.uint8)
    else:
        raw[0::2] = np.clip(np.rint(z.real) + 128, 0, 255).astype(np.uint8)
        raw[1::2] = np.clip(np.rint(z.imag) + 128, 0, 255).astype(np.uint8)
    bin_hz = fs / 40000.0
    step = 0.0 if mode == "iq8" else float(rng.choice([0.0, 0.0, bin_hz / 2, bin_hz / 3]))
    desc = f"seed {seed}: fs {fs / 1e6:.4f} MHz fc {fc / 1e6:.4f} max_fo {max_fo:.0f} step {step:.2f} mode {mode} signed {signed} dc {remove_dc} mix {mix:.0f}"
    worst = 0.0
    with gpsacq.Engine(fc, fs, max_fo) as eng:
        mean = eng.iq8_mean(raw, signed=signed) if remove_dc else (0.0, 0.0)
        tasks = [(int(rng.integers(0, nblk)), int(rng.integers(0, 32))) for _ in range(2)]
        if mode == "iq8":
            inp = eng.iq8_input(signed=signed, remove_dc=remove_dc, mean=mean, mix_hz=mix, fs=fs, total_samples=ns)
            cells, peaks = eng.search_iq8(raw, inp, tasks=tasks)
            bits = iq8_to_bits(raw, signed=signed, remove_dc=remove_dc, mix_hz=mix, fs=fs)
            dev_bits = eng.iq8_to_bits(raw, signed=signed, remove_dc=remove_dc, mix_hz=mix, fs=fs)
            flips = int(np.unpackbits(bits ^ dev_bits).sum())
            if flips > 2:
                raise AssertionError(f"{flips} converted bits differ from the numpy restatement")
            orc = Oracle(fc, fs, max_fo)
            for t, (b, sv) in enumerate(tasks):
                oc, _ = orc.search_block(dev_bits[b * 5120:(b + 1) * 5120].tobytes(), sv)
                worst = max(worst, close(cells["max_pwr"][t], oc["max_pwr"], "iq8 max_pwr"), close(cells["tot_pwr"][t], oc["tot_pwr"], "iq8 tot_pwr"))
                if (cells["max_i"][t] != oc["max_i"]).sum() > 0:
                    raise AssertionError("iq8 argmax")
            return desc + f" flips {flips}", worst
        eng.set_doppler_step(step)
        sub, kmax = eng.doppler_sub, eng.kmax
        inp = eng.iq8_input(signed=signed, remove_dc=remove_dc, mean=mean, mix_hz=mix, fs=fs, total_samples=ns, multibit=(1 if mode == "multibit" else 2))
        cells, peaks = eng.search_iq8(raw, inp, tasks=tasks)
        quad = np.zeros(40960, np.uint8)
        lib().oracle_lo_quadrants(fc, fs, 40960, _p(quad))
        samples = (iq8_to_real if mode == "multibit" else iq8_to_complex)(raw, signed=signed, remove_dc=remove_dc, mix_hz=mix, fs=fs)
        dmax, nl = int(max_fo * 40000 / fs), eng.num_lags
        for t, (b, sv) in enumerate(tasks):
            pts = sorted(set([-kmax, kmax, 0] + [int(v) for v in rng.integers(-kmax, kmax + 1, 5)]))
            for r in range(sub):
                sel = [k for k in pts if k % sub == r]
                if not sel:
                    continue
                dops = [(k - r) // sub for k in sel]
                blk = samples[b * 40960:]
                if mode == "multibit":
                    mp, mi, tp = multibit_cells(blk, quad, code_replica(fs, sv), dmax, nl, eps=r / sub, dops=dops)
                else:
                    mp, mi, tp = complex_cells(blk, code_replica(fs, sv), dmax, nl, eps=r / sub, dops=dops)
                got = cells[t][np.array(sel) + kmax]
                worst = max(worst, close(got["max_pwr"], mp, mode + " max_pwr"), close(got["tot_pwr"], tp, mode + " tot_pwr"))
                if (got["max_i"] != mi).sum() > 0:
                    i = int(np.nonzero(got["max_i"] != mi)[0][0])
                    raise AssertionError(f"{mode} argmax at point {sel[i]}: {got['max_i'][i]} vs {mi[i]}")
    return desc, worst


def cli_case(seed, rng, fs, fc):
    """gps_test on a random file (any length: whole runs, a partial run, a partial block, nothing), random batch size, one to three
    engines on the GPU, reference quirk on or off, 1-bit or 8-bit IQ: stdout must be the banner, format_report of the engine's own
    peaks for the whole runs, and the reference's end-of-file line."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host import BANNER, GPS_TEST
    iq = bool(rng.integers(0, 3) == 0)
    quirks = bool(rng.integers(0, 2))
    per = 81920 if iq else 5120
    n_bytes = int(rng.integers(0, 5 * 32 * per)) if rng.integers(0, 4) else int(rng.integers(0, 6)) * 32 * per
    raw = rng.integers(0, 256, size=n_bytes, dtype=np.uint8)
    env = dict(os.environ, GPSACQ_BATCH_RUNS=str(int(rng.choice([1, 2, 3, 64]))), GPSACQ_REF_QUIRKS="1" if quirks else "0",
               GPSACQ_DEVICES=",".join(["0"] * int(rng.integers(1, 4))))
    mix = float(rng.choice([0.0, fc]))
    if iq:
        env.update(GPSACQ_INPUT="iq_u8", GPSACQ_MIX_HZ=repr(mix))
    desc = f"seed {seed}: fs {fs / 1e6:.4f} MHz fc {fc / 1e6:.4f} mode cli bytes {n_bytes} iq {iq} quirks {quirks} batch {env['GPSACQ_BATCH_RUNS']} devices {env['GPSACQ_DEVICES']}"
    n_runs = n_bytes // (32 * per)
    with tempfile.NamedTemporaryFile(suffix=".bin", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as f:
        raw.tofile(f)
        f.flush()
        r = subprocess.run([GPS_TEST, f.name, repr(fc), repr(fs), "5000"], capture_output=True, text=True, env=env, timeout=300)
    if r.returncode != 0:
        raise AssertionError(f"gps_test exit {r.returncode}: {r.stderr[-300:]}")
    want = BANNER
    if n_runs > 0:
        with gpsacq.Engine(fc, fs, 5000.0, ref_quirks=quirks) as eng:
            if iq:
                mean = eng.iq8_mean(raw)
                inp = eng.iq8_input(remove_dc=True, mean=mean, mix_hz=mix, fs=fs, total_samples=raw.size // 2)
                _, peaks = eng.search_iq8(raw[:n_runs * 32 * per], inp)
            else:
                _, peaks = eng.search(raw[:n_runs * 32 * per], want_cells=False)
        want += gpsacq.format_report(peaks)
    want += "run out of file!\n"
    if r.stdout != want:
        a, b = r.stdout.split("\n"), want.split("\n")
        i = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b)))
        raise AssertionError(f"stdout differs at line {i}: {a[i:i + 1]} vs {b[i:i + 1]} ({len(a)} vs {len(b)} lines)")
    return desc + f" runs {n_runs}", 0.0


def plumbing_case(seed, rng, fs, fc, max_fo):
    """The ways a batch reaches the kernels must not change a bit of the result: the pipeline (random cuts into slots, 1-bit or 8-bit
    IQ), several engines sharing the GPU (Doppler slabs with a random grid; whole runs), against one plain search."""
    kind = str(rng.choice(["pipe_bits", "pipe_iq", "multi_grid", "multi_blocks"]))
    desc = f"seed {seed}: fs {fs / 1e6:.4f} MHz fc {fc / 1e6:.4f} max_fo {max_fo:.0f} mode {kind}"
    bin_hz = fs / 40000.0
    if kind.startswith("pipe"):
        iq = kind == "pipe_iq"
        nblk = int(rng.integers(3, 40))
        per = 81920 if iq else int(rng.choice([5120, 5120, 5456, 6000]))
        with gpsacq.Engine(fc, fs, max_fo) as eng:
            if iq:
                raw = rng.integers(0, 256, size=nblk * per, dtype=np.uint8)
                mean = eng.iq8_mean(raw)
                inp = eng.iq8_input(remove_dc=True, mean=mean, mix_hz=fc, fs=fs, total_samples=raw.size // 2)
                _, want = eng.search_iq8(raw, inp)
            else:
                raw = rng.integers(0, 256, size=(nblk - 1) * per + 5120, dtype=np.uint8)
                _, want = eng.search(raw, stride=per, want_cells=False)
            got, b0, inflight = [], 0, []
            while b0 < nblk or inflight:
                while b0 < nblk and len(inflight) < 3:
                    n = int(min(nblk - b0, rng.integers(1, 12)))
                    slot = [s_ for s_ in range(3) if s_ not in [i[0] for i in inflight]][0]
                    nbytes = n * per if iq else (n - 1) * per + 5120
                    buf = eng.pipe_buffer(slot, max(nbytes, 12 * 81920 if iq else 12 * 6000))
                    buf[:nbytes] = raw[b0 * per:b0 * per + nbytes]
                    if iq:
                        sub = eng.iq8_input(remove_dc=True, mean=mean, mix_hz=fc, fs=fs, first_sample=b0 * 40960, total_samples=raw.size // 2)
                        eng.pipe_submit(slot, n, per, iq=sub)
                    else:
                        eng.pipe_submit(slot, n, per)
                    inflight.append((slot, b0, n))
                    b0 += n
                slot, first, n = inflight.pop(0)
                pk = eng.pipe_collect(slot)
                got.append((first, n, pk))
            # the pipeline's schedule is "task t = block t of the batch against PRN t % 32": the same tasks through the plain entry
            for first, n, pk in got:
                tasks = [(first + t, t % 32) for t in range(n)]
                _, ref = eng.search_iq8(raw, inp, tasks=tasks) if iq else eng.search(raw, tasks=tasks, stride=per)
                if pk.tobytes() != ref.tobytes():
                    raise AssertionError(f"pipeline peaks of blocks {first}..{first + n - 1} differ from the plain search")
                if first % 32 == 0 and pk.tobytes() != want[first:first + n].tobytes():
                    raise AssertionError("pipeline differs from the reference schedule")
        return desc + f" blocks {nblk} batches {len(got)}", 0.0
    ndev = int(rng.integers(2, 6))
    if kind == "multi_grid":
        step = float(rng.choice([0.0, bin_hz / 2, bin_hz / 3, 2.0 * bin_hz]))
        bits = rng.integers(0, 256, size=3 * 5120, dtype=np.uint8)
        tasks = [(int(rng.integers(0, 3)), int(rng.integers(0, 32))) for _ in range(int(rng.integers(1, 9)))]
        with gpsacq.Engine(fc, fs, max_fo) as eng:
            eng.set_doppler_step(step)
            _, want = eng.search(bits, tasks=tasks)
        with gpsacq.MultiEngine(fc, fs, max_fo, devices=(0,) * ndev) as me:
            me.set_doppler_step(step)
            got = me.search_grid(bits, tasks)
        for k in ("snr", "lo_shift", "ca_shift"):
            if not np.array_equal(got[k], want[k]):
                raise AssertionError(f"multi grid {k} differs with {ndev} engines")
        return desc + f" engines {ndev} step {step:.2f} tasks {len(tasks)}", 0.0
    runs = int(rng.integers(1, 5))
    bits = rng.integers(0, 256, size=runs * 32 * 5120, dtype=np.uint8)
    with gpsacq.Engine(fc, fs, max_fo) as eng:
        _, want = eng.search(bits, want_cells=False)
    with gpsacq.MultiEngine(fc, fs, max_fo, devices=(0,) * ndev) as me:
        peaks, best = me.search_blocks(bits)
    if not np.array_equal(peaks, want):
        raise AssertionError(f"multi blocks peaks differ with {ndev} engines")
    for sv in range(32):
        cand = want[sv::32]
        k = max((float(p["snr"]), -int(p["lo_shift"]), int(p["ca_shift"])) for p in cand)
        if (float(best["snr"][sv]), -int(best["lo_shift"][sv]), int(best["ca_shift"][sv])) != k:
            raise AssertionError(f"multi blocks best of PRN index {sv} differs")
    return desc + f" engines {ndev} runs {runs}", 0.0


def one(seed):
    rng = np.random.default_rng(50000 + seed)
    fs = float(rng.choice([rng.uniform(1.2e6, 4e6), rng.uniform(4e6, 10e6), rng.uniform(10e6, 20e6)]))
    if os.environ.get("FUZZ_FS"):  # FUZZ_FS=lo,hi: sampling rates of one kernel instance only (5.25e6,5.5e6: k_corr<22, FOLD>)
        lo, hi = (float(v) for v in os.environ["FUZZ_FS"].split(","))
        fs = float(rng.uniform(lo, hi))
    fc = float(rng.uniform(0.0, 0.49 * fs))
    bin_hz = fs / 40000.0
    max_fo = float(rng.uniform(2 * bin_hz, min(60 * bin_hz, 25000.0)))
    mode = str(rng.choice(["coherent", "coherent", "quirks", "noncoh", "noncoh_creep", "window", "stride", "iq8", "iq8", "multibit", "complex"]))
    if mode in ("iq8", "multibit", "complex") and not os.environ.get("FUZZ_PLUMBING") and not os.environ.get("FUZZ_CLI"):
        return iq_case(seed, rng, fs, fc, max_fo, mode)
    if os.environ.get("FUZZ_CLI"):  # FUZZ_CLI=1: front-end cases only
        return cli_case(seed, rng, fs, fc)
    if seed % 5 == 4 or os.environ.get("FUZZ_PLUMBING"):  # FUZZ_PLUMBING=1: plumbing cases only
        return plumbing_case(seed, rng, fs, fc, max_fo)
    if mode in ("noncoh", "noncoh_creep") and rng.integers(0, 2):
        fs = float(round(fs / 1000.0) * 1000.0)  # a whole number of samples per code period: block alignment is defined
        bin_hz = fs / 40000.0
    step = 0.0 if mode in ("quirks",) else float(rng.choice([0.0, bin_hz / 2, bin_hz / 3, bin_hz / 7, 2.0 * bin_hz, 3.3 * bin_hz]))
    nblk = 6
    bits = rng.integers(0, 256, size=nblk * 5120 + 4096, dtype=np.uint8)
    desc = f"seed {seed}: fs {fs / 1e6:.4f} MHz fc {fc / 1e6:.4f} max_fo {max_fo:.0f} step {step:.2f} mode {mode}"
    worst = 0.0
    orc = Oracle(fc, fs, max_fo, ref_quirks=(mode == "quirks"))
    with gpsacq.Engine(fc, fs, max_fo, ref_quirks=(mode == "quirks")) as eng:
        eng.set_doppler_step(step)
        sub, dstride, kmax = eng.doppler_sub, eng.doppler_stride, eng.kmax
        if eng.num_doppler != 2 * kmax + 1:
            raise AssertionError("grid size")
        if mode in ("noncoh", "noncoh_creep"):
            if sub != 1 or dstride != 1:
                eng.set_doppler_step(0.0)
                sub, dstride, kmax = 1, 1, eng.kmax
            n_acc, bstep = int(rng.integers(2, 4)), int(rng.integers(1, 3))
            stride = 5120
            eng.set_noncoherent(n_acc, bstep)
            eng.set_creep_compensation(mode == "noncoh_creep")
            align = bool(rng.integers(0, 2)) and float(eng.num_lags) * 1000.0 == fs  # whole samples per code period only
            eng.set_block_alignment(align)
            tasks = [(0, int(rng.integers(0, 32))), (int(rng.integers(0, nblk - (n_acc - 1) * bstep)), int(rng.integers(0, 32)))]
            cells, peaks = eng.search(bits.tobytes(), tasks=tasks, stride=stride)
            for t, (b, sv) in enumerate(tasks):
                want = orc.search_noncoherent(bits.tobytes(), stride, b, sv, n_acc, bstep, creep=(mode == "noncoh_creep"), align=align)
                worst = max(worst, close(cells["max_pwr"][t], want["max_pwr"], "nc max_pwr"), close(cells["tot_pwr"][t], want["tot_pwr"], "nc tot_pwr"))
                if (cells["max_i"][t] != want["max_i"]).sum() > 1:
                    raise AssertionError("nc argmax")
            return desc + f" n_acc {n_acc} step {bstep} align {align}", worst
        if mode == "window":
            first = int(rng.integers(-kmax, kmax + 1))
            n = int(rng.integers(1, kmax - first + 2))
            eng.set_doppler_window(first, n)
        else:
            first, n = -kmax, 2 * kmax + 1
        if mode == "stride":
            stride = int(rng.choice([5000, 5120, 5456, 6000]))
            nb = (bits.size - 5120) // stride + 1
            cells, peaks = eng.search(bits.tobytes(), stride=stride)
            tasks = [(b, b % 32) for b in range(nb)]
            check = [int(v) for v in rng.choice(nb, 2, replace=False)]
        else:
            stride = 5120
            tasks = [(int(rng.integers(0, nblk)), int(rng.integers(0, 32))) for _ in range(3)]
            if mode == "quirks":
                tasks[0] = (tasks[0][0], 0)  # the PRN index the overrun touches
            cells, peaks = eng.search(bits.tobytes(), tasks=tasks)
            check = range(len(tasks))
        for t in check:
            b, sv = tasks[t]
            pts = sorted(set([first, first + n - 1] + [int(v) for v in rng.integers(first, first + n, 6)]))
            blk = bits[b * stride:b * stride + 5120].tobytes()
            if mode == "quirks":
                oc_all, _ = orc.search_block(blk, sv)
                oc, ks = oc_all[np.array(pts) + orc.dmax], pts
            else:
                oc, ks = orc.search_grid(blk, sv, sub=sub, dstride=dstride, points=pts)
            got = cells[t][np.array(ks) - first]
            worst = max(worst, close(got["max_pwr"], oc["max_pwr"], "max_pwr"), close(got["tot_pwr"], oc["tot_pwr"], "tot_pwr"))
            if (got["max_i"] != oc["max_i"]).sum() > 0:
                # a tie to rounding is legitimate: the oracle's own power at our lag must equal its maximum to 1e-5
                bad = np.nonzero(got["max_i"] != oc["max_i"])[0]
                for i in bad:
                    raise AssertionError(f"argmax at point {ks[i]}: {got['max_i'][i]} vs {oc['max_i'][i]} (pwr {got['max_pwr'][i]} vs {oc['max_pwr'][i]})")
            k = int(np.argmax(cells[t]["snr"]))
            if peaks["lo_shift"][t] != k + first or peaks["ca_shift"][t] != cells[t]["max_i"][k]:
                raise AssertionError("peak != scan over the cells")
    return desc, worst


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    for seed in range(first, first + count):
        try:
            desc, worst = one(seed)
            print(f"ok   {desc}  worst rel {worst:.2e}", flush=True)
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print(f"FAIL seed {seed}: {type(ex).__name__}: {ex}", flush=True)
            traceback.print_exc(limit=3)
    print(f"fuzz: {count - bad} ok, {bad} failed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
