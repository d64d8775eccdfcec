#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Run in the authoring container:  python tests/golden/make_golden.py
Everything here is seeded; re-running reproduces the committed files bit for bit
(given the same numpy version).  Nothing reads /root/reference.

Fixtures:
  gps_sig_tmp.bin            DATA file bundled with the reference (its only test input:
                             README.md:45,57; made by gps_sig_gen.m, PRN 8, fs 8.184 MHz,
                             IF 2.046 MHz).  Copied verbatim (it is data, not source).
  ref_known_answers.json     reference outputs recorded in BASELINE.md section 2 and
                             SURVEY.md section 8c (gps_test on gps_sig_tmp.bin) plus the
                             JKS table (Raw GPS ... .html:79-83).  Transcribed by hand.
  synth_nott_fs5456.bin      seeded synthetic 1-bit real-IF capture, fs 5.456 MHz, IF 4.092 MHz,
                             64 blocks (2 runs), five PRNs of the JKS table injected
                             (SURVEY.md section 8d "stand-in set").
  synth_rtl_fs2800.bin       seeded synthetic capture, fs 2.8 MHz, IF 0.62 MHz (inexact float32
                             NCO rates, D=143), 33 blocks.
  synth_iq8_rtl.bin          seeded synthetic rtl-sdr style capture: uint8 offset-128 interleaved I,Q at
                             baseband, fs 2.8 MHz, 2 blocks, PRN 5 at +1023 Hz, DC offset added.
  synth_iq8_rtl_bits.bin     its 1-bit conversion by oracle/iq8_oracle.py (mix 0.62 MHz, DC removed).
  synth_weak_fs5456.bin      seeded weak-signal capture for the non-coherent extension: fs 5.456 MHz,
                             7 blocks laid out every 5456 bytes (8 whole C/A periods), PRN 12 weak, PRN 3 stronger.
  synth_weak_rtl_fs2800.bin  BASELINE configs[3] shape: fs 2.8 MHz, IF 0.62 MHz, 6 blocks every 5250 bytes (15 C/A
                             periods), PRN 9 weak at +40 kHz (receiver LO offset, bin 571 of +-1428).
  np64_cells_*.npz           per-cell {max_pwr, max_i, tot_pwr, second_pwr, second_i} from an INDEPENDENT float64
                             numpy restatement (np.fft / pocketfft) of
                             c/search_offline.cpp:121-201 for a few (block, sv) pairs.
"""
import json
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
N = 40000
CPS = 1.023e6
L1 = 1575.42e6
# c/search_offline.cpp:20-53 (T1, T2)
TAPS = [(2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9), (3, 10), (2, 3), (3, 4),
        (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9),
        (1, 3), (4, 6), (5, 7), (6, 8), (7, 9), (8, 10), (1, 6), (2, 7), (3, 8), (4, 9)]


def ca_chips(t1, t2):
    """c/cacode.h:9-35 restated with Python lists; returns 1023 chips (0/1)."""
    g1 = [0] + [1] * 10
    g2 = [0] + [1] * 10
    out = []
    for _ in range(1023):
        out.append(g1[10] ^ g2[t1] ^ g2[t2])
        g1[0] = g1[3] ^ g1[10]
        g2[0] = g2[2] ^ g2[3] ^ g2[6] ^ g2[8] ^ g2[9] ^ g2[10]
        g1 = [g1[0]] + g1[0:10]
        g2 = [g2[0]] + g2[0:10]
    return np.array(out, dtype=np.int64)


def code_replica(fs, sv):
    """c/search_offline.cpp:76,83-103: float32 sequential NCO + linear blend."""
    chips = ca_chips(*TAPS[sv])
    bip = np.where(chips == 1, -1.0, 1.0)
    ca_rate = np.float32(CPS / fs)
    ph = np.float32(0)
    k = 0
    out = np.empty(N, dtype=np.float32)
    for i in range(N):
        chip = np.float32(bip[k % 1023])
        ph = np.float32(ph + ca_rate)
        if float(ph) >= 1.0:
            ph = np.float32(float(ph) - 1.0)
            k += 1
            chip = np.float32(float(chip) * (1.0 - float(ph)))
            chip = np.float32(chip + np.float32(ph * np.float32(bip[k % 1023])))
        out[i] = chip
    return out


def lo_quadrants(fc, fs, n):
    """c/search_offline.cpp:127,131,155-156."""
    rate = np.float32(4 * fc / fs)
    ph = np.float32(0)
    q = np.empty(n, dtype=np.int64)
    for i in range(n):
        q[i] = int(ph)
        ph = np.float32(ph + rate)
        if ph >= 4:
            ph = np.float32(ph - np.float32(4))
    return q


def mix_block(bytes_, quad):
    """c/search_offline.cpp:141-153."""
    bits = np.unpackbits(np.frombuffer(bytes_, dtype=np.uint8), bitorder='little').astype(np.int64)
    lo_sin = np.array([1, 1, 0, 0])
    lo_cos = np.array([0, 1, 1, 0])
    i = 1.0 - 2.0 * (bits ^ lo_cos[quad[:bits.size]])
    q = 1.0 - 2.0 * (bits ^ lo_sin[quad[:bits.size]])
    return i + 1j * q


def np64_cells(fc, fs, max_fo, blocks, pairs, quirks=False):
    """Independent float64 restatement of Sample+Correlate (:121-201) for the given
    (block_index, sv) pairs.  float64 throughout except the float32 rounding of the
    stored spectra (fftwf_complex storage)."""
    dmax = int(max_fo * N / fs)
    S = int(np.ceil(fs / 1000.0))
    quad = lo_quadrants(fc, fs, 40960)
    res = {}
    code_cache = {}
    for b, sv in pairs:
        x = mix_block(blocks[b], quad)
        D = np.fft.fft(x[:N]).astype(np.complex64).astype(np.complex128)
        if sv not in code_cache:
            code_cache[sv] = np.fft.fft(code_replica(fs, sv).astype(np.float64)).astype(np.complex64).astype(np.complex128)
        C = code_cache[sv].copy()
        if quirks and sv == 0:
            C[:960] = x[N:N + 960]
        mp = np.empty(2 * dmax + 1, np.float64)
        mi = np.empty(2 * dmax + 1, np.int64)
        tp = np.empty(2 * dmax + 1, np.float64)
        p2 = np.empty(2 * dmax + 1, np.float64)  # the runner-up lag and its power: what a rounding tie of the argmax
        i2 = np.empty(2 * dmax + 1, np.int64)    # is proven against (tests/test_gpu_parity.py::test_cells_vs_numpy_golden)
        for d in range(-dmax, dmax + 1):
            prod = (np.conj(D) * np.roll(C, d)).astype(np.complex64).astype(np.complex128)
            y = np.fft.ifft(prod) * N
            pwr = (y[:S].real ** 2 + y[:S].imag ** 2)
            mp[d + dmax] = pwr.max()
            mi[d + dmax] = int(pwr.argmax())
            tp[d + dmax] = pwr.sum()
            rest = pwr.copy()
            rest[mi[d + dmax]] = -1.0
            i2[d + dmax] = int(rest.argmax())
            p2[d + dmax] = rest.max()
        res[(b, sv)] = (mp, mi, tp, p2, i2)
    return dmax, S, res


def synth_capture(fs, fc, nblk, sats, seed, amp=0.151):
    """Seeded 1-bit real-IF capture: noise + BPSK C/A signals."""
    rng = np.random.default_rng(seed)
    ns = nblk * 40960
    m = np.arange(ns, dtype=np.float64)
    y = rng.standard_normal(ns)
    for prn, lo, ca in sats:
        fd = lo * fs / N
        chips = 1.0 - 2.0 * ca_chips(*TAPS[prn - 1])
        code_rate = CPS * (1 + fd / L1)
        idx = np.floor((m + ca) * code_rate / fs).astype(np.int64) % 1023
        ph = 2 * np.pi * ((fc + fd) / fs * m + rng.random())
        y += amp * chips[idx] * np.cos(ph)
    bits = (y < 0).astype(np.uint8)
    return np.packbits(bits, bitorder='little').tobytes()


def main():
    # ---- recorded reference outputs (hand transcription; provenance in the docstring) ----
    known = {
        "gps_sig_tmp": {
            "args": ["gps_sig_tmp.bin", 2.046e6, 8.184e6, 5000],
            "source": "BASELINE.md section 2 / SURVEY.md section 8c (reference gps_test run during the survey)",
            "runs": 12,
            "sv7_snr": [713.6, 682.2, 668.3, 650.3, 612.1, 632.0, 647.8, 666.7, 682.8, 585.9, 634.4, 615.8],
            "sv7_lo_shift": [0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0],
            "sv7_ca_shift": [260, 1540, 2820, 4100, 5380, 6660, 7940, 1036, 2316, 3596, 4876, 6156],
            "run0_hits_sv": [0, 1, 5, 7, 27, 28, 31],
            "run0_hits_snr": [28.0, 25.6, 25.1, 713.6, 26.9, 25.4, 25.1],
            "run0_hits_lo": [1, -7, -1, 0, 1, -1, 3],
            "run0_hits_ca": [7844, 7612, 7571, 260, 6899, 7571, 7939],
            "other_best_snr_range": [12.8, 31.0],
            "note": "run-0 sv 0 entry (28.0) includes the fwd_buf overrun effect (SURVEY fact 5)",
        },
        "nottingham_jks_table": {
            "source": "Raw GPS signal samples data set for testing GPS receivers.html:79-83",
            "prn": [1, 21, 29, 30, 31],
            "lo_shift": [6, 8, -9, -9, -8],
            "ca_shift": [1465, 686, 3868, 2998, 2337],
            "snr": [108.7, 121.7, 167.2, 145.2, 121.3],
        },
    }
    with open(os.path.join(HERE, "ref_known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)

    # ---- synthetic stand-in for the (absent) Nottingham capture ----
    jks = [(1, 6, 1465), (21, 8, 686), (29, -9, 3868), (30, -9, 2998), (31, -8, 2337)]
    nott = synth_capture(5.456e6, 4.092e6, 64, jks, 20260927)
    open(os.path.join(HERE, "synth_nott_fs5456.bin"), "wb").write(nott)
    rtl = synth_capture(2.8e6, 0.62e6, 33, [(8, 20, 700), (12, -33, 2100), (1, 3, 5)], 7)
    open(os.path.join(HERE, "synth_rtl_fs2800.bin"), "wb").write(rtl)

    # ---- 8-bit IQ capture ----
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    from iq8_oracle import iq8_to_bits
    rng = np.random.default_rng(99)
    ns = 2 * 40960
    m = np.arange(ns, dtype=np.float64)
    fd = 1023.0
    chips = 1.0 - 2.0 * ca_chips(*TAPS[4])
    idx = np.floor((m + 321) * CPS * (1 + fd / L1) / 2.8e6).astype(np.int64) % 1023
    z = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) / np.sqrt(2) + 0.3 * chips[idx] * np.exp(2j * np.pi * (fd / 2.8e6 * m + 0.123))
    z = 30.0 * z + (3.7 + 1.2j)
    iq = np.empty(2 * ns, dtype=np.uint8)
    iq[0::2] = np.clip(np.rint(z.real) + 128, 0, 255).astype(np.uint8)
    iq[1::2] = np.clip(np.rint(z.imag) + 128, 0, 255).astype(np.uint8)
    iq.tofile(os.path.join(HERE, "synth_iq8_rtl.bin"))
    iq8_to_bits(iq, signed=False, remove_dc=True, mix_hz=0.62e6, fs=2.8e6).tofile(os.path.join(HERE, "synth_iq8_rtl_bits.bin"))

    # ---- weak-signal capture, block starts a whole number of code periods apart ----
    rng = np.random.default_rng(4242)
    ns = 7 * 5456 * 8
    m = np.arange(ns, dtype=np.float64)
    y = rng.standard_normal(ns)
    for prn, lo, ca, amp in [(12, 4, 1000, 0.055), (3, -11, 3333, 0.11)]:
        fd = lo * 5.456e6 / N
        chips = 1.0 - 2.0 * ca_chips(*TAPS[prn - 1])
        idx = np.floor((m + ca) * CPS * (1 + fd / L1) / 5.456e6).astype(np.int64) % 1023
        nav = np.where((np.floor(m / (20 * 5456)).astype(np.int64) % 2) == 0, 1.0, -1.0)  # a data-bit flip every 20 ms
        y += amp * nav * chips[idx] * np.cos(2 * np.pi * ((4.092e6 + fd) / 5.456e6 * m + rng.random()))
    np.packbits((y < 0).astype(np.uint8), bitorder='little').tofile(os.path.join(HERE, "synth_weak_fs5456.bin"))

    rng = np.random.default_rng(2800)
    ns = 6 * 5250 * 8
    m = np.arange(ns, dtype=np.float64)
    y = rng.standard_normal(ns)
    fo = 571 * 2.8e6 / N  # LO offset: shifts the carrier, not the code rate
    chips = 1.0 - 2.0 * ca_chips(*TAPS[8])
    idx = np.floor((m + 777) * CPS / 2.8e6).astype(np.int64) % 1023
    y += 0.06 * chips[idx] * np.cos(2 * np.pi * ((0.62e6 + fo) / 2.8e6 * m + 0.4))
    np.packbits((y < 0).astype(np.uint8), bitorder='little').tofile(os.path.join(HERE, "synth_weak_rtl_fs2800.bin"))

    def blocks_of(buf):
        return [buf[i * 5120:(i + 1) * 5120] for i in range(len(buf) // 5120)]

    # ---- independent float64 per-cell vectors ----
    sig = open(os.path.join(HERE, "gps_sig_tmp.bin"), "rb").read()
    jobs = [
        ("np64_cells_sigtmp.npz", 2.046e6, 8.184e6, 5000.0, blocks_of(sig), [(7, 7), (0, 0), (1, 1), (39, 7), (31, 31)], True),
        ("np64_cells_nott.npz", 4.092e6, 5.456e6, 5000.0, blocks_of(nott), [(0, 0), (20, 20), (28, 28), (29, 29), (30, 30), (3, 3), (32, 0)], False),
        ("np64_cells_rtl.npz", 0.62e6, 2.8e6, 5000.0, blocks_of(rtl), [(7, 7), (11, 11), (32, 0)], False),
    ]
    for name, fc, fs, mfo, blks, pairs, quirks in jobs:
        dmax, S, res = np64_cells(fc, fs, mfo, blks, pairs, quirks)
        out = {"fc": fc, "fs": fs, "max_fo": mfo, "dmax": dmax, "S": S, "quirks": int(quirks),
               "pairs": np.array(pairs, dtype=np.int64)}
        for (b, sv), (mp, mi, tp, p2, i2) in res.items():
            out[f"max_pwr_{b}_{sv}"] = mp
            out[f"max_i_{b}_{sv}"] = mi
            out[f"tot_pwr_{b}_{sv}"] = tp
            out[f"second_pwr_{b}_{sv}"] = p2
            out[f"second_i_{b}_{sv}"] = i2
        np.savez_compressed(os.path.join(HERE, name), **out)
        print(name, "dmax", dmax, "S", S, "pairs", len(pairs))


if __name__ == "__main__":
    main()
