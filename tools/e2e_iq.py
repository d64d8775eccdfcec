import numpy as np, subprocess, os, time, sys
sys.path.insert(0, 'tests')
rng = np.random.default_rng(1)
runs = 340
n = runs * 32 * 40960
path = '/dev/shm/e2e_iq_u8.bin'
with open(path, 'wb') as f:
    for r in range(runs):
        z = rng.integers(96, 160, size=2 * 32 * 40960, dtype=np.uint8)
        z.tofile(f)
exe = 'gnss-gps-sdr_amd/bin/gps_test'
for k in range(3):
    t0 = time.perf_counter()
    r = subprocess.run([exe, path, '4.092e6', '5.456e6', '5000'], capture_output=True, text=True,
                       env=dict(os.environ, GPSACQ_INPUT='iq_u8', GPSACQ_MIX_HZ='4.092e6', GPSACQ_TRACE='1'))
    w = time.perf_counter() - t0
    print('wall %.3f s rc %d runs %d' % (w, r.returncode, r.stdout.count('satellite:')))
    print([l for l in r.stderr.splitlines() if 'SearchTask' in l][-1])
os.remove(path)
