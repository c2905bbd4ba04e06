"""The appends of a keyframe step are queued on the window and applied by one copy + one launch in front of the next call that uses the
device (pba.hip: flushAppends, round 6).  These tests drive the queue's corner cases directly: the same frame appended to several times with
no device call in between (two queued operations on one array force an early flush), connections that grow in steps, flag updates of
existing landmarks stacked on each other, a capacity growth (arrays move) with operations pending — and hold the result against the
CPU checker loaded the same way and against a window loaded in one go."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _load_in_steps(backend, win, cuts):
    """every frame's landmarks and connections in len(cuts) + 1 batches: all frames first (no landmarks), then batch by batch"""
    intr = win.scene.intrinsics
    for f in win.frames:
        backend.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False)
    for frac in list(cuts) + [1.0]:
        for f in win.frames:
            n = int(round(frac * len(f.uv)))
            backend.set_landmarks(f.frame_id, f.uv[:n], f.idepth_init[:n], f.patch[:n], np.zeros(n, dtype=np.uint8))
        for f in win.frames:
            n = int(round(frac * len(f.uv)))
            for g in win.frames:
                if g is not f:
                    backend.set_connection(f.frame_id, g.frame_id, np.zeros(n, dtype=np.uint8))
    return backend


@pytest.mark.parametrize("cuts", [(0.5,), (0.1, 0.11, 0.6)])
def test_appends_in_steps_without_device_calls_in_between(cuts):
    from dsopp_amd import capi
    from oracle import pyoracle as po
    # 1500 landmarks per frame: the third batch of the second parametrisation crosses the first capacity (1024) with operations pending
    win = syn.make_window(num_frames=4, num_points=6000, width=320, height=240, seed=23)
    o = _load_in_steps(po.OracleWindow(po.default_pba_options()), win, cuts)
    g = _load_in_steps(capi.HipWindow(capi.default_pba_options()), win, cuts)
    one = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    eo, ito, nvo = o.solve()
    eg, itg, nvg = g.solve()
    e1, it1, nv1 = one.solve()
    assert (ito, nvo) == (itg, nvg) == (it1, nv1)
    assert abs(eo - eg) <= 1e-7 * abs(eo) and abs(e1 - eg) <= 1e-9 * abs(e1)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        T1, ab1 = one.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
        assert np.abs(T1 - Tg).max() <= 1e-9 and np.abs(ab1 - abg).max() <= 1e-9
        lo, lg, l1 = o.get_landmarks(f.frame_id), g.get_landmarks(f.frame_id, False), one.get_landmarks(f.frame_id, False)
        assert np.array_equal(lo["flags"] & 3, lg["flags"] & 3) and np.array_equal(l1["flags"], lg["flags"])
        assert np.allclose(lg["idepth"], lo["idepth"], rtol=1e-6, atol=1e-9) and np.allclose(lg["idepth"], l1["idepth"], rtol=1e-9, atol=1e-12)
    g.close()
    one.close()


def test_stacked_flag_updates_keep_the_last_one():
    """two flag updates of the same frame with nothing in between: the second must find what the first left (marginalized -> the
    to_marginalize bit is assigned by every update, local_frame.hpp:492-497)"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=3, num_points=600, width=320, height=240, seed=5)
    o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    o.solve()
    g.solve()
    f = win.frames[1]
    n = len(f.uv)
    first = np.zeros(n, dtype=np.uint8)
    first[::3] = 1
    second = np.zeros(n, dtype=np.uint8)
    second[::3] = 1
    second[1::5] = 1
    for b in (o, g):
        b.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, first)
        b.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, second)
    lo, lg = o.get_landmarks(f.frame_id), g.get_landmarks(f.frame_id, False)
    assert np.array_equal(lo["flags"], lg["flags"])
    # ... and the solve that follows folds the same landmarks
    eo, ito, nvo = o.solve()
    eg, itg, nvg = g.solve()
    assert (ito, nvo) == (itg, nvg) and abs(eo - eg) <= 1e-7 * abs(eo)
    g.close()


def test_connections_declared_before_their_target_is_pushed():
    """LocalFrame::update keeps the residual list of EVERY connection of the frame, whether or not the solver holds the target yet
    (local_frame.hpp:507-519): a connection declared towards an id that is not in the window waits on the host (no device table) and becomes
    one when a frame with that id is pushed; towards an id that never comes it costs nothing and changes nothing.  Loaded that way — in two
    steps, the second after the push — the window must equal the one loaded in the usual order (to summation order: the tables are created
    in another sequence, and with them the order in which a landmark's connections are added up)."""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=4, num_points=2400, width=320, height=240, seed=29)
    intr = win.scene.intrinsics
    early = capi.HipWindow(capi.default_pba_options())
    for i, f in enumerate(win.frames):
        early.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False)
        early.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
        half = len(f.uv) // 2
        for g in win.frames:
            if g is f:
                continue
            # towards the frames already there: everything; towards the ones to come: the first half now ...
            early.set_connection(f.frame_id, g.frame_id, np.zeros(len(f.uv) if g.frame_id in [h.frame_id for h in win.frames[:i]] else half, dtype=np.uint8))
        early.set_connection(f.frame_id, 10_000 + i, np.zeros(len(f.uv), dtype=np.uint8))  # an id that never comes
        for h in win.frames[:i]:
            # ... and the rest once the target is in the window (h declared f early)
            early.set_connection(h.frame_id, f.frame_id, np.zeros(len(h.uv), dtype=np.uint8))
    usual = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    early.begin()
    usual.begin()
    (ee, ne), (eu, nu) = early.calculate_energy(), usual.calculate_energy()
    assert ne == nu and abs(ee - eu) <= 1e-13 * abs(eu)
    early.linearize()
    usual.linearize()
    for a, b in zip(early.get_system(), usual.get_system()):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()
    ee, ite, nve = early.solve()
    eu, itu, nvu = usual.solve()
    assert (ite, nve) == (itu, nvu) and abs(ee - eu) <= 1e-10 * abs(eu)
    for f in win.frames:
        Te, abe = early.get_pose(f.frame_id)
        Tu, abu = usual.get_pose(f.frame_id)
        assert np.abs(Te - Tu).max() <= 1e-9 and np.abs(abe - abu).max() <= 1e-9
        le, lu = early.get_landmarks(f.frame_id, False), usual.get_landmarks(f.frame_id, False)
        assert np.array_equal(le["flags"], lu["flags"]) and np.allclose(le["idepth"], lu["idepth"], rtol=1e-9, atol=1e-12)
    early.close()
    usual.close()
