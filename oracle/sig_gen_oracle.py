"""sig_gen_oracle.py -- numpy restatement of the reference's test-signal generator gps_sig_gen.m (TEST INFRASTRUCTURE:
only tests/ may import it; the product's generator is gnss-gps-sdr_amd/csrc/gen_kernels.hip, k_siggen).

What the script does (citations: /root/reference/gps_sig_gen.m):
  :8-16   C/A code of PRN `sv` as +-1, one impulse per chip at 8 samples per chip (upsample = zero insertion),
          repeated 20 times per navigation bit
  :18-19  num_data random +-1 navigation bits (UNSEEDED rand: the bits of the bundled gps_sig_tmp.bin are not
          reproducible from the script; recover_data_bits() reads them back out of the file)
  :22,35  pulse shaping by conv(data, rcosine(1, 8)): MATLAB's default raised-cosine FIR -- roll-off 0.5, delay 3
          symbols, 49 taps, h(t) = sinc(t) cos(pi R t) / (1 - (2 R t)^2), the singular points t = +-1/(2R) set to
          (R/2) sin(pi/(2R)) (= 3.06e-17 in double for R = 0.5, not 0)
  :34,36  y = real(data .* exp(1i*2*pi*fc*(0:L-1)*(1/ca_rate))), fc = ca_rate/4: the argument is evaluated left to right
          in double, ((2 pi fc) n) (1/ca_rate); on odd n cos() is the ~1e-10 rounding residue of that product, whose sign
          decides the written bit
  :37-41  bit = (1 - sign(y))/2 written as 'ubit1' (LSB first); y == 0 gives 0.5, which fwrite rounds to 1

  :21-30  the script's OTHER output, gps_sig_tmp_for_hackrf_tx.bin (README.md section 2.2 replays it through a HackRF):
          x = conv(rcosine(1, 8), [data data data data data]) .* 50 as I, zeros as Q, interleaved, fwrite 'int8' (round to
          nearest, ties away from zero, saturating) -- the same shaped baseband as the 1-bit file's, five times over, at IF 0
          (hackrf_tx() below; the file itself is not bundled)

Pin: with the 100 recovered bits generate() reproduces the reference's gps_sig_tmp.bin (2 046 006 bytes,
sha256 a6242849...) BIT FOR BIT (tests/test_siggen.py).  Two details that the file itself settles: conv() accumulates
oldest input first (MATLAB's filter order; the reverse order differs in 287 900 samples where shaped pulses cancel to
+-1e-17), and an exact zero is written as 1 (the last 7 samples of the file)."""
import numpy as np

CA_BASE_RATE = 1.023e6   # gps_sig_gen.m:8
OV = 8                   # ca_ov_ratio, :9
CA_PER_DATA = 20         # num_ca_per_data, :13
ROLLOFF, DELAY = 0.5, 3  # rcosine(1, 8) defaults


def rcosine_taps():
    """rcosine(1, 8) ('fir/normal', R = 0.5, delay 3): 49 taps, peak 1 (the scale does not matter to a 1-bit output)."""
    t = np.arange(-DELAY * OV, DELAY * OV + 1) / OV
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.sinc(t) * np.cos(np.pi * ROLLOFF * t) / (1.0 - (2.0 * ROLLOFF * t) ** 2)
    h[np.abs(np.abs(t) - 1.0 / (2.0 * ROLLOFF)) < 1e-12] = (ROLLOFF / 2.0) * np.sin(np.pi / (2.0 * ROLLOFF))
    return h


def shaped_pulses(dseq, h):
    """conv(upsample(dseq, 8), h), accumulated oldest input first (MATLAB filter order), plain double adds."""
    n_chip = len(dseq)
    n_out = n_chip * OV + len(h) - 1
    m = np.arange(n_out)
    jmax = m // OV
    acc = np.zeros(n_out)
    for o in range((len(h) - 1) // OV, -1, -1):  # o = 6: the oldest contributing chip
        j = jmax - o
        tap = m - OV * j
        ok = (j >= 0) & (j < n_chip) & (tap < len(h))
        term = np.zeros(n_out)
        term[ok] = dseq[j[ok]] * h[tap[ok]]
        acc = acc + term
    return acc


def generate(chips01, data_pm1):
    """chips01: the 1023 C/A chips (0/1) of the PRN; data_pm1: navigation bits (+-1).  Returns the packed capture bytes
    (LSB first, length ceil((len(data)*20*1023*8 + 48) / 8))."""
    g = 1.0 - 2.0 * np.asarray(chips01, dtype=np.float64)
    dseq = np.repeat(np.asarray(data_pm1, dtype=np.float64), CA_PER_DATA * 1023) * np.tile(g, CA_PER_DATA * len(data_pm1))
    sh = shaped_pulses(dseq, rcosine_taps())
    ca_rate = CA_BASE_RATE * OV
    fc = ca_rate / 4.0
    n = np.arange(len(sh), dtype=np.float64)
    x = ((2.0 * np.pi * fc) * n) * (1.0 / ca_rate)
    y = sh * np.cos(x)
    bits = np.where(y < 0, 1, 0).astype(np.uint8)
    bits[y == 0] = 1
    return np.packbits(bits, bitorder="little")


def recover_data_bits(capture_bytes, chips01):
    """The navigation bits inside a capture made by gps_sig_gen.m: at the centre of chip k (sample 8 k + 24, a multiple
    of 4: carrier +1) the shaped pulse is data * chip.  Returns (+-1 per bit, worst |mean| over a bit's 20460 chips = 1.0
    when every chip agrees)."""
    bits = np.unpackbits(np.frombuffer(capture_bytes, dtype=np.uint8), bitorder="little")
    n_data = (len(bits) - 48) // (OV * 1023 * CA_PER_DATA)
    n_chip = n_data * CA_PER_DATA * 1023
    s = 1.0 - 2.0 * bits[OV * np.arange(n_chip) + DELAY * OV].astype(np.float64)
    g = 1.0 - 2.0 * np.asarray(chips01, dtype=np.float64)
    per = (s * np.tile(g, CA_PER_DATA * n_data)).reshape(n_data, CA_PER_DATA * 1023).mean(axis=1)
    return np.sign(per), float(np.abs(per).min())


# ---- gps_sig_gen.m:21-30: the HackRF transmit file ------------------------------------------------------------------------
TX_REPEAT = 5     # x = [data, data, data, data, data], :23
TX_SCALE = 50.0   # .*50, :25


def tx_samples(n_data, n_repeat=TX_REPEAT):
    """complex samples of the transmit file: length(conv(num, x)) = n_repeat * n_data * 20 * 1023 * 8 + 48"""
    return n_repeat * n_data * CA_PER_DATA * 1023 * OV + 48


def matlab_int8(v):
    """fwrite(fid, v, 'int8') of doubles: round to nearest, ties away from zero (MATLAB round), saturate to [-128, 127]"""
    v = np.asarray(v, dtype=np.float64)
    t = np.trunc(v)
    r = np.where(np.abs(v - t) == 0.5, t + np.sign(v), np.rint(v))
    return np.clip(r, -128, 127).astype(np.int8)


def tx_baseband(chips01, data_pm1, first=0, count=None, n_repeat=TX_REPEAT, newest_first=False):
    """conv(num, x)[first : first + count] in double, x = the impulse train of n_repeat copies of the data-modulated code
    (:13-19,23-24).  Accumulated oldest input first like shaped_pulses(); newest_first=True adds the seven terms in the
    opposite order -- MATLAB's order for conv(short, long) is not documented, and the int8 file cannot tell (the two differ
    by ~1e-17 where pulses cancel; a written value only moves if x * 50 sits within that of a half-integer), which
    tests/test_siggen.py asserts on every sample it generates."""
    g = 1.0 - 2.0 * np.asarray(chips01, dtype=np.float64)
    data = np.asarray(data_pm1, dtype=np.float64)
    n_chip = n_repeat * len(data) * CA_PER_DATA * 1023
    total = n_chip * OV + 48
    if count is None:
        count = total - first
    assert 0 <= first and first + count <= total
    h = rcosine_taps()
    m = np.arange(first, first + count, dtype=np.int64)
    jmax = m // OV
    acc = np.zeros(count)
    order = range(0, 7) if newest_first else range(6, -1, -1)
    for o in order:
        j = jmax - o
        tap = m - OV * j
        ok = (j >= 0) & (j < n_chip) & (tap < len(h))
        jj = j[ok]
        d = data[(jj // (CA_PER_DATA * 1023)) % len(data)] * g[jj % 1023]
        term = np.zeros(count)
        term[ok] = d * h[tap[ok]]
        acc = acc + term
    return acc


def hackrf_tx(chips01, data_pm1, first=0, count=None, n_repeat=TX_REPEAT, newest_first=False):
    """gps_sig_tmp_for_hackrf_tx.bin, complex samples first .. first + count - 1, as interleaved int8 (I, Q = 0), :25-29"""
    x = tx_baseband(chips01, data_pm1, first, count, n_repeat, newest_first) * TX_SCALE
    out = np.zeros(2 * x.size, dtype=np.int8)
    out[0::2] = matlab_int8(x)
    return out
