"""GPU tests of the run-time hand-out of cells (k_corr<..., PERSIST>, include/gpsacq.h gpsacq_set_cell_handout): three resident
workgroups per CU draw (task, Doppler point) tickets instead of one workgroup per cell.  The arithmetic of a cell is the same, so the
cells and peaks must be the SAME BIT FOR BIT with the hand-out on (the default) and off -- for every instance that has a persistent
form (12, 22, 28, 33 accumulator columns; the 12-column one with non-coherent sums in registers), for fine grids whose tasks are
handed out in chunks, for one-point windows, single tasks, and device task lists with entries the kernel must reject.  No reference
equivalent: Correlate() (c/search_offline.cpp:169-201) is a serial loop; what a cell is stays pinned by the oracle tests."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpsacq_mod():
    import gpsacq
    gpsacq.load_library()
    return gpsacq


def _both(eng, run):
    """run(eng) with the hand-out on, off and on again: three identical results."""
    out = []
    for on in (True, False, True):
        eng.set_cell_handout(on)
        out.append(run(eng))
    for a, b in ((out[0], out[1]), (out[0], out[2])):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    return out[0]


@pytest.mark.parametrize("name,fc,fs,columns", [("synth_nott_fs5456.bin", 4.092e6, 5.456e6, 22), ("gps_sig_tmp.bin", 2.046e6, 8.184e6, 33),
                                               ("synth_rtl_fs2800.bin", 0.62e6, 2.8e6, 12)])
def test_reference_grid_is_bit_identical_with_and_without_the_handout(gpsacq_mod, golden_dir, name, fc, fs, columns):
    """Reference schedule (block t against PRN t % 32) over every block of the fixture, the reference's +-5 kHz grid: 64 / 399 / 33
    tasks (2409 ... 19 551 cells for 768 resident workgroups)."""
    buf = open(os.path.join(golden_dir, name), "rb").read()
    buf = buf[:(len(buf) // 5120) * 5120]
    with gpsacq_mod.Engine(fc, fs, 5000.0) as eng:
        assert eng.acc_columns == columns
        cells, peaks = _both(eng, lambda e: e.search(buf))
        assert cells.shape[0] == len(buf) // 5120 and (cells["max_i"] >= 0).all() and (peaks["snr"] > 0).all()


def test_28_column_instance(gpsacq_mod, golden_dir):
    """fs = 6.9 MHz: 6900 lags = 28 accumulator columns (the W1H instance)."""
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:40 * 5120]
    with gpsacq_mod.Engine(1.7e6, 6.9e6, 5000.0) as eng:
        assert eng.acc_columns == 28
        _both(eng, lambda e: e.search(buf))


def test_fine_grids_are_handed_out_in_chunks(gpsacq_mod, golden_dir):
    """+-100 kHz at fs 2.8 MHz: 2857 bins per task (23 chunks of 125 points), then 953 points at 210 Hz with 5 non-coherent sums in
    registers (8 chunks of 120), then 45.5 Hz sub-bin spectra at fs 5.456 MHz (3 spectra per block: 219 points = 2 chunks of 110):
    a task's chunks go to whichever XCD asks next, the cells do not care."""
    rtl = open(os.path.join(golden_dir, "synth_rtl_fs2800.bin"), "rb").read()
    tasks = [(b, sv) for b in range(3) for sv in (0, 5, 20, 31)]
    with gpsacq_mod.Engine(0.62e6, 2.8e6, 100000.0) as eng:
        assert eng.num_doppler == 2857 and eng.acc_columns == 12
        cells, _ = _both(eng, lambda e: e.search(rtl[:4 * 5120], tasks=tasks))
        assert cells.shape == (12, 2857)
        eng.set_doppler_step(250.0)
        stride = eng.aligned_stride()
        eng.set_noncoherent(5, 1)
        n = (len(rtl) - 5120) // stride + 1
        nc_tasks = [(b, sv) for b in range(min(3, n - 4)) for sv in (0, 20)]
        cells, _ = _both(eng, lambda e: e.search(rtl, tasks=nc_tasks, stride=stride))
        assert cells.shape[1] == 953
    nott = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:6 * 5120]
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        eng.set_doppler_step(50.0)
        assert eng.num_doppler == 219
        _both(eng, lambda e: e.search(nott, tasks=[(b, sv) for b in range(6) for sv in (0, 20, 28)]))


def test_windows_of_one_point_and_single_tasks(gpsacq_mod, golden_dir):
    """A Doppler window of one bin (every ticket is its own unit), one task of 73 cells (a tenth of the resident workgroups), and both."""
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:34 * 5120]
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        full, _ = _both(eng, lambda e: e.search(buf))
        one, _ = _both(eng, lambda e: e.search(buf[:5120]))
        assert np.array_equal(one[0], full[0])
        eng.set_doppler_window(6, 1)
        col, pk = _both(eng, lambda e: e.search(buf))
        assert col.shape == (34, 1) and np.array_equal(col[:, 0], full[:, 6 + eng.dmax]) and (pk["lo_shift"] == 6).all()
        single, _ = _both(eng, lambda e: e.search(buf[:5120], tasks=[(0, 0)]))
        assert np.array_equal(single[0, 0], full[0, 6 + eng.dmax])


def test_rejected_tasks_of_a_device_task_list(gpsacq_mod, golden_dir):
    """A task list handed over in device memory is bounded by the kernel itself (k_corr: max_i = -1, zero powers): a persistent
    workgroup that draws such a cell must move on to its next ticket like any other."""
    import torch
    buf = np.frombuffer(open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:8 * 5120], dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d_bits = torch.from_numpy(buf.copy()).to(dev)
    tasks = np.array([(0, 0), (9, 3), (1, 20), (-1, 2), (2, 32), (3, -1), (7, 31), (8, 0)] + [(b % 8, b % 32) for b in range(40)], dtype=np.int32)
    bad = [1, 3, 4, 5, 7]
    d_tasks = torch.from_numpy(tasks).to(dev)
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        def run(e):
            d_cells = torch.full((len(tasks), e.num_doppler, 4), 0x55, dtype=torch.int32, device=dev)
            d_pk = torch.zeros((len(tasks), 4), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e.search_device(d_bits.data_ptr(), 8, d_pk.data_ptr(), d_tasks_ptr=d_tasks.data_ptr(), n_tasks=len(tasks), d_cells_ptr=d_cells.data_ptr(), sync=True)
            return d_cells.cpu().numpy().view(gpsacq_mod.CELL_DTYPE).reshape(len(tasks), e.num_doppler), d_pk.cpu().numpy().view(gpsacq_mod.PEAK_DTYPE).reshape(-1)
        cells, peaks = _both(eng, run)
        for t in range(len(tasks)):
            if t in bad:
                assert (cells[t]["max_i"] == -1).all() and (cells[t]["max_pwr"] == 0).all() and peaks["snr"][t] == 0
            else:
                assert (cells[t]["max_i"] >= 0).all() and (cells[t]["tot_pwr"] > 0).all()


def test_environment_switch_and_repeated_launches(gpsacq_mod, golden_dir, monkeypatch):
    """GPSACQ_CORR_PERSIST=0 at gpsacq_create selects one workgroup per cell; the hand-out state is zeroed in front of every launch:
    twenty searches in a row (the state of one must not leak into the next), asynchronously enqueued, all equal."""
    import torch
    buf = np.frombuffer(open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:64 * 5120], dtype=np.uint8)
    monkeypatch.setenv("GPSACQ_CORR_PERSIST", "0")
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        ref_cells, ref_peaks = eng.search(buf)
    monkeypatch.delenv("GPSACQ_CORR_PERSIST")
    dev = torch.device("cuda", 0)
    d_bits = torch.from_numpy(buf.copy()).to(dev)
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        d_pk = [torch.zeros((64, 4), dtype=torch.int32, device=dev) for _ in range(20)]
        torch.cuda.synchronize()
        for k in range(20):
            eng.search_device(d_bits.data_ptr(), 64 - (k % 3), d_pk[k].data_ptr(), sync=False)
        eng.synchronize()
        for k in range(20):
            n = 64 - (k % 3)
            assert np.array_equal(d_pk[k].cpu().numpy().view(gpsacq_mod.PEAK_DTYPE).reshape(-1)[:n], ref_peaks[:n]), k
        cells, peaks = eng.search(buf)
        assert np.array_equal(cells, ref_cells) and np.array_equal(peaks, ref_peaks)


@pytest.mark.parametrize("name,fc,fs,max_fo", [("gps_sig_tmp.bin", 2.046e6, 8.184e6, 5000.0), ("synth_nott_fs5456.bin", 4.092e6, 5.456e6, 5000.0),
                                             ("synth_rtl_fs2800.bin", 0.62e6, 2.8e6, 100000.0)])
def test_every_cell_is_written_exactly_where_it_belongs(gpsacq_mod, golden_dir, name, fc, fs, max_fo):
    """The hand-out must give every (task, Doppler point) to exactly one workgroup: the cell buffer is poisoned (all bits set: NaN
    powers, lag -1) in front of every search, so a cell nobody drew would stay poisoned and a cell computed for the wrong slot would
    differ from the one-workgroup-per-cell result.  19 551 / 4672 / 94 281 cells for 768 workgroups; three searches each."""
    import torch
    buf = np.frombuffer(open(os.path.join(golden_dir, name), "rb").read(), dtype=np.uint8)
    n = buf.size // 5120
    dev = torch.device("cuda", 0)
    d_bits = torch.from_numpy(buf[:n * 5120].copy()).to(dev)
    with gpsacq_mod.Engine(fc, fs, max_fo) as eng:
        def run(e):
            d_cells = torch.full((n, e.num_doppler, 4), -1, dtype=torch.int32, device=dev)
            d_pk = torch.full((n, 4), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e.search_device(d_bits.data_ptr(), n, d_pk.data_ptr(), d_cells_ptr=d_cells.data_ptr(), sync=True)
            c = d_cells.cpu().numpy().view(gpsacq_mod.CELL_DTYPE).reshape(n, e.num_doppler)
            assert (c["max_i"] >= 0).all() and np.isfinite(c["max_pwr"]).all() and np.isfinite(c["tot_pwr"]).all() and (c["tot_pwr"] > 0).all()
            return c, d_pk.cpu().numpy().view(gpsacq_mod.PEAK_DTYPE).reshape(-1)
        _both(eng, run)
