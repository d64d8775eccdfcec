"""Non-coherent accumulation (extension, SURVEY.md section 8f.2 / BASELINE config 4): the device path
against the oracle's restatement (sum of per-lag powers over blocks, then the reference's scan)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noncoherent_vs_oracle(golden_dir):
    import gpsacq
    from oracle_lib import Oracle
    buf = open(os.path.join(golden_dir, "synth_weak_fs5456.bin"), "rb").read()
    orc = Oracle(4.092e6, 5.456e6, 5000.0)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        stride = eng.aligned_stride()
        assert stride == 5456  # 8 C/A periods of 682 bytes
        tasks = [(0, 11), (0, 2), (0, 20), (1, 11)]
        coh_cells, coh_peaks = eng.search(buf, tasks=tasks, stride=stride)
        eng.set_noncoherent(5, 1)
        nc_cells, nc_peaks = eng.search(buf, tasks=tasks, stride=stride)
        for t, (b, sv) in enumerate(tasks):
            want = orc.search_noncoherent(buf, stride, b, sv, 5, 1)
            np.testing.assert_allclose(nc_cells["max_pwr"][t], want["max_pwr"], rtol=2e-5)
            np.testing.assert_allclose(nc_cells["tot_pwr"][t], want["tot_pwr"], rtol=2e-5)
            assert (nc_cells["max_i"][t] != want["max_i"]).sum() <= 1
        # the point of the mode: PRN 12 (weak) is buried coherently, stands out non-coherently
        noise_coh, noise_nc = coh_peaks["snr"][2], nc_peaks["snr"][2]
        assert coh_peaks["snr"][0] < 1.3 * noise_coh
        assert nc_peaks["snr"][0] > 2.0 * noise_nc
        assert int(nc_peaks["lo_shift"][0]) == 4 and abs(int(nc_peaks["ca_shift"][0]) - 1000) <= 1
        assert int(nc_peaks["lo_shift"][1]) == -11 and int(nc_peaks["ca_shift"][1]) == 3333
        # out-of-range accumulation span is rejected, n_acc = 1 restores the coherent results bit for bit
        with pytest.raises(gpsacq.GpsAcqError):
            eng.search(buf, tasks=[(3, 0)], stride=stride)
        eng.set_noncoherent(1)
        again_cells, again_peaks = eng.search(buf, tasks=tasks, stride=stride)
        assert np.array_equal(again_cells, coh_cells) and np.array_equal(again_peaks, coh_peaks)
        # default schedule with accumulation over a stride of blocks: task t = (block t, prn t % 32)
        eng.set_noncoherent(3, 2)
        c, p = eng.search(buf, stride=stride, n_tasks=3)
        want = orc.search_noncoherent(buf, stride, 2, 2, 3, 2)
        np.testing.assert_allclose(c["max_pwr"][2], want["max_pwr"], rtol=2e-5)
