"""tools/stamp_probe.py -- what gpsacq_cycle_stamp_device measures (GPU box): per-XCD clocks from the CUs' cycle counters over busy windows of
1 .. 400 steps of a 640-block search, idle windows (the counters stand still), and two back-to-back stamps (a few thousand ticks on every CU).
Output: profiles/r05_experiments/e_xcd_clocks.log, block 1."""
import sys, os, time, json
sys.path.insert(0, "gnss-gps-sdr_amd/python"); sys.path.insert(0, ".")
import torch, gpsacq, numpy as np
dev = torch.device("cuda", 0)
eng = gpsacq.Engine(4.092e6, 5.456e6, 5000.0)
st = torch.cuda.ExternalStream(eng.stream_ptr, device=dev)
nblk = 640
bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device=dev)
pk = torch.zeros((nblk, 4), dtype=torch.int32, device=dev)
def window(n_steps, busy=True, sleep_ms=0.0):
    s = torch.zeros((2, 512), dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.synchronize()
    eng.cycle_stamp_device(s[0].data_ptr()); e0.record(st)
    if busy:
        for _ in range(n_steps):
            eng.search_device(bits.data_ptr(), nblk, pk.data_ptr(), sync=False)
    else:
        eng.synchronize(); time.sleep(sleep_ms * 1e-3)
    eng.cycle_stamp_device(s[1].data_ptr()); e1.record(st)
    eng.synchronize(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); v = s.cpu().numpy()
    import bench
    chip, per_xcd, n = bench.cu_clocks(v, ms)
    a, b = v[0].astype(np.int64), v[1].astype(np.int64); okm = (a > 0) & (b > a)
    allm = (b - a)[okm] / (ms * 1e3)
    return ms, (chip, per_xcd, n, (float(allm.min()), float(allm.max())) if allm.size else None), v
for _ in range(3): window(20)
for n in (1, 2, 8, 40, 200, 400):
    ms, mhz, v = window(n)
    print("busy %4d steps %8.2f ms  chip %s per-XCD %s CUs %d all-CU min/max %s" % (n, ms, mhz[0] and round(mhz[0], 1), mhz[1], mhz[2], mhz[3]))
for sl in (5, 50, 500):
    ms, mhz, v = window(0, busy=False, sleep_ms=sl)
    print("idle %4d ms    %8.2f ms  chip %s CUs %d all-CU min/max %s" % (sl, ms, mhz[0], mhz[2], mhz[3]))
s = torch.zeros((2, 512), dtype=torch.int64, device=dev)
eng.cycle_stamp_device(s[0].data_ptr()); eng.cycle_stamp_device(s[1].data_ptr(), sync=True)
v = s.cpu().numpy(); d = (v[1] - v[0])[(v[0] > 0) & (v[1] > 0)]
print("back-to-back stamps: CUs", d.size, "delta cycles min/median/max", int(d.min()), int(np.median(d)), int(d.max()))
print("slots reached per XCD", [int(np.count_nonzero(v[0][64 * x:64 * (x + 1)])) for x in range(8)])
