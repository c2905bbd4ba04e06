"""The window path at the resolution the reference's dense configuration runs at: 1280 x 1024 (test/test_data/tummono/dense.yaml:21-22,
resize_ratio 1 on TUM-mono images).  Every other PBA test (and, until round 5, every PBA bench number) uses 640 x 480 or smaller
images; here the texel rows are 40 KB apart, image coordinates exceed 1023 and the intensity plane has 320 tiles per row."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fullres_window():
    return syn.make_window(num_frames=4, num_points=1600, width=1280, height=1024, seed=11)


def _close(a, b, rtol, atol):
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= atol + rtol * np.abs(np.asarray(b)).max()


def test_stage_parity_at_1280x1024(fullres_window):
    """calculateEnergy / linearize / calculateStep against the CPU checker, f64: energies 1e-10, systems 1e-9, step 1e-9"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = fullres_window
    o = po.OracleWindow(po.default_pba_options())
    syn.load_window(o, win)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    o.begin()
    g.begin()
    (eo, no), (eg, ng) = o.calculate_energy(), g.calculate_energy()
    assert no == ng and abs(eo - eg) <= 1e-10 * abs(eo)
    o.linearize()
    g.linearize()
    for a, b in zip(o.get_system(), g.get_system()):
        assert _close(b, a, 1e-9, 1e-9 * np.abs(a).max())
    so, sg = o.calculate_step(1e-5), g.calculate_step(1e-5)
    assert np.abs(so - sg).max() <= 1e-9 * max(1.0, np.abs(so).max())
    # every landmark sits where the generator put it: coordinates beyond the 640 x 480 range are exercised
    assert max(f.uv[:, 0].max() for f in win.frames) > 1100 and max(f.uv[:, 1].max() for f in win.frames) > 900
    g.close()


def test_full_solve_parity_at_1280x1024(fullres_window):
    """the fused loop (7 iterations, production settings) + point statuses + covariances: poses 1e-7, as at the small sizes"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = fullres_window
    o = po.OracleWindow(po.default_pba_options())
    syn.load_window(o, win)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    eo, ito, nvo = o.solve()
    eg, itg, nvg = g.solve()
    assert (ito, nvo) == (itg, nvg) and abs(eo - eg) <= 1e-7 * abs(eo)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
        lo, lg = o.get_landmarks(f.frame_id), g.get_landmarks(f.frame_id, False)
        assert _close(lg["idepth"], lo["idepth"], 1e-6, 1e-9)
        assert np.array_equal(lo["flags"] & 3, lg["flags"] & 3) and np.array_equal(lo["n_inliers"], lg["n_inliers"])
    g.close()


def test_f32_texels_at_1280x1024(fullres_window):
    """f32 storage (16-byte texels; the reference's -DUSE_FLOAT build): fp32 round-off class, as tests/test_gpu_pba.py at 320 x 240"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = fullres_window
    o = po.OracleWindow(po.default_pba_options())
    syn.load_window(o, win)
    g = capi.HipWindow(capi.default_pba_options(dtype=capi.F32))
    syn.load_window(g, win)
    o.begin()
    g.begin()
    (eo, no), (eg, ng) = o.calculate_energy(), g.calculate_energy()
    assert abs(no - ng) <= 2 and abs(eo - eg) <= 1e-4 * abs(eo)
    g.close()


@pytest.mark.parametrize("name,F,P", [("12kf_50k", 12, 50000), ("15kf_5k", 15, 5000)])
def test_bench_windows_at_1280x1024_against_the_checker(name, F, P):
    """The two windows bench.py's `roofline_large_fullres` rows time (the reference's dense configuration, test/test_data/tummono/dense.yaml:
    21-22,35,42: 1280 x 1024, up to 15 keyframes / 5000 points; and the C4 size, 12 KF / 50 000 points) against the CPU checker AT THEIR OWN
    SIZE: energies 1e-10, systems 1e-9, step 1e-9, then the production solve() — poses 1e-7, statuses identical.  12 x 42 MB of f64 texels
    live in HBM, not in the Infinity Cache; the two-stage Schur build and the coarse sweep table are the variants in use."""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    po.set_threads(16)
    try:
        win = syn.make_window(num_frames=F, num_points=P, width=1280, height=1024, seed=1, render_device="cuda")
        o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
        g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
        o.begin()
        g.begin()
        (eo, no), (eg, ng) = o.calculate_energy(), g.calculate_energy()
        assert no == ng and no > 0.9 * P * (F - 1) * 0.5 and abs(eo - eg) <= 1e-10 * abs(eo)
        o.linearize()
        g.linearize()
        for a, b in zip(o.get_system(), g.get_system()):
            assert _close(b, a, 1e-9, 1e-9 * np.abs(a).max())
        so, sg = o.calculate_step(1e-5), g.calculate_step(1e-5)
        assert np.abs(so - sg).max() <= 1e-9 * max(1.0, np.abs(so).max())
        g.close()
        # the production call from the same start: fused loop + relinearisation + covariances + point statuses
        o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
        g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
        eo, ito, nvo = o.solve()
        eg, itg, nvg = g.solve()
        assert (ito, nvo) == (itg, nvg) and abs(eo - eg) <= 1e-7 * abs(eo)
        for f in win.frames:
            To, abo = o.get_pose(f.frame_id)
            Tg, abg = g.get_pose(f.frame_id)
            assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
            lo, lg = o.get_landmarks(f.frame_id), g.get_landmarks(f.frame_id, False)
            assert _close(lg["idepth"], lo["idepth"], 1e-6, 1e-9)
            assert np.array_equal(lo["flags"] & 3, lg["flags"] & 3) and np.array_equal(lo["n_inliers"], lg["n_inliers"])
        g.close()
    finally:
        po.set_threads(max(1, min(__import__("os").cpu_count() or 1, 8) - 1))
