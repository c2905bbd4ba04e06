"""Error behaviour of the C-ABI (include/dsopp_hip.h): the contract violations the reference CHECKs or logs
(PROB_SRC/photometric_bundle_adjustment.cpp:66-69,101-102) come back as negative codes with a message, never as a crash or a
silent no-op, and leave the objects usable."""
import re

import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _code(exc):
    return int(re.search(r"error (-?\d+)", str(exc.value)).group(1))


def test_window_contract_violations(tiny_window):
    from dsopp_amd import capi
    win = tiny_window
    intr = win.scene.intrinsics
    g = capi.HipWindow(capi.default_pba_options())
    with pytest.raises(capi.HipError) as e:   # nothing pushed yet
        g.solve()
    assert _code(e) == -6
    f0, f1 = win.frames[0], win.frames[1]
    g.push_frame(f1.frame_id, f1.timestamp, f1.pixelinfo, None, intr, syn.mat_to_params(f1.T_w_c_init), 1.0, np.zeros(2), False, False)
    with pytest.raises(capi.HipError) as e:   # an older frame after a newer one
        g.push_frame(f0.frame_id, f0.timestamp, f0.pixelinfo, None, intr, syn.mat_to_params(f0.T_w_c_init), 1.0, np.zeros(2), True, False)
    assert _code(e) == -3 and "ascending" in str(e.value)
    with pytest.raises(capi.HipError) as e:   # unknown frame id
        g.set_landmarks(999, f1.uv, f1.idepth_init, f1.patch, np.zeros(len(f1.uv), dtype=np.uint8))
    assert _code(e) == -2
    g.set_landmarks(f1.frame_id, f1.uv, f1.idepth_init, f1.patch, np.zeros(len(f1.uv), dtype=np.uint8))
    with pytest.raises(capi.HipError) as e:   # landmarks can only be appended
        g.set_landmarks(f1.frame_id, f1.uv[:5], f1.idepth_init[:5], f1.patch[:5], np.zeros(5, dtype=np.uint8))
    assert _code(e) == -1
    # a connection to a frame that is not (yet, or no longer) in the window is legal and ignored by the solve: the
    # reference keeps such entries in the keyframe's connection map too and only builds residuals for window members
    g.set_connection(f1.frame_id, 12345, np.zeros(len(f1.uv), dtype=np.uint8))
    with pytest.raises(capi.HipError) as e:   # but not with more entries than the reference frame has landmarks
        g.set_connection(f1.frame_id, 12345, np.zeros(len(f1.uv) + 1, dtype=np.uint8))
    assert _code(e) == -1
    with pytest.raises(capi.HipError) as e:   # stage call before begin()
        g.calculate_step(1e-5)
    assert _code(e) == -6
    with pytest.raises(capi.HipError) as e:   # restore without snapshot
        g.restore()
    assert _code(e) == -6
    with pytest.raises(capi.HipError):        # depth maps need at least 1 level, at most 5
        g.create_reference_depth_maps(0)
    # the window is still usable after all of that: complete it and solve
    f2 = win.frames[2]
    g.push_frame(f2.frame_id, f2.timestamp, f2.pixelinfo, None, intr, syn.mat_to_params(f2.T_w_c_init), 1.0, np.zeros(2), False, False)
    g.set_landmarks(f2.frame_id, f2.uv, f2.idepth_init, f2.patch, np.zeros(len(f2.uv), dtype=np.uint8))
    g.set_connection(f1.frame_id, f2.frame_id, np.zeros(len(f1.uv), dtype=np.uint8))
    g.set_connection(f2.frame_id, f1.frame_id, np.zeros(len(f2.uv), dtype=np.uint8))
    e_, it, nv = g.solve()
    assert np.isfinite(e_) and nv > 0
    g.close()


def test_aligner_and_pyramid_contract_violations(tiny_window):
    from dsopp_amd import capi
    win = tiny_window
    fr, ft = win.frames[0], win.frames[1]
    H, W = fr.image_u8.shape
    intr = win.scene.intrinsics
    pyr = capi.Pyramid(W, H, 2)
    pyr.build(fr.image_u8)
    with pytest.raises(capi.HipError) as e:   # level beyond the pyramid
        pyr.get_level(2)
    assert _code(e) == -1
    a = capi.HipAligner()
    with pytest.raises(capi.HipError) as e:   # nothing pushed
        a.solve()
    assert _code(e) == -6
    T = syn.mat_to_params(fr.T_w_c_gt)
    a.push_target(2000, T, pyr, 0, intr, 1.0, np.zeros(2))
    ids, wgt = np.zeros((H, W)), np.zeros((H, W))
    with pytest.raises(capi.HipError) as e:   # the reference must come first after reset()
        a.push_reference_depth_map(1000, T, pyr, 0, intr, ids, wgt, 1.0, np.zeros(2))
    assert _code(e) == -3
    a.reset()
    a.push_reference_depth_map(3000, T, pyr, 0, intr, ids, wgt, 1.0, np.zeros(2))
    with pytest.raises(capi.HipError) as e:   # target older than the reference
        a.push_target(2000, T, pyr, 0, intr, 1.0, np.zeros(2))
    assert _code(e) == -3
    a.push_target(4000, T, pyr, 0, intr, 1.0, np.zeros(2))
    r = a.solve()                              # an empty depth map: no valid residual, zero iterations, pose unchanged
    assert r["n_valid"] == 0 and r["iterations"] == 0 and np.abs(r["T_w_target"] - T).max() < 1e-12
    f32 = capi.Pyramid(W, H, 1, dtype=capi.F32)
    f32.build(fr.image_u8)
    a.reset()
    with pytest.raises(capi.HipError) as e:   # dtype mismatch between aligner (f64) and pyramid (f32)
        a.push_target(5000, T, f32, 0, intr, 1.0, np.zeros(2))
    assert _code(e) == -1
    for o in (a, pyr, f32):
        o.close()


def test_empty_and_degenerate_inputs_of_the_tracker_side_calls(tiny_window):
    """empty sets, keyframes without immature landmarks, an empty depth map, a window without active landmarks: every call
    returns cleanly with the reference's result for the empty case"""
    from dsopp_amd import capi
    win = tiny_window
    intr = win.scene.intrinsics
    H, W = win.frames[0].image_u8.shape
    g = capi.HipWindow(capi.default_pba_options())
    pyrs = []
    for i, f in enumerate(win.frames):
        p = capi.Pyramid(W, H, 2)
        p.build(f.image_u8)
        pyrs.append(p)
        if i < 2:
            g.push_frame(f.frame_id, f.timestamp, None, None, intr, syn.mat_to_params(f.T_w_c_gt), 1.0, np.zeros(2), i == 0, False, pyramid=p)
            g.set_landmarks(f.frame_id, np.zeros((0, 2)), np.zeros(0), np.zeros((0, 8)), np.zeros(0, dtype=np.uint8))
    new = win.frames[2]
    empty = dict(projection=np.zeros((0, 2)), direction=np.zeros((0, 3)), patch=np.zeros((0, 8)), gradient=np.zeros((0, 2)),
                 status=np.zeros(0, dtype=np.uint8))
    s_empty = capi.ImmatureSet(empty)
    # activation: no active landmarks, one keyframe with an empty set, one without a set
    st, idp, res = g.activate_landmarks([win.frames[0].frame_id, win.frames[1].frame_id], [s_empty, None], pyrs[2], syn.mat_to_params(new.T_w_c_gt),
                                        1.0, (0, 0), 2000, 2.0, True)
    assert res["number_of_active_points"] == 0 and res["n_activated"] == 0 and len(st[0]) == 0 and len(st[1]) == 0
    assert abs(res["min_distance_to_neighbor"] - 0.0) < 1e-15        # 2.0 + (0 - 2000) * 0.001 clamps at 0 (:34-38)
    # depth estimation on an empty set, alone and inside a batch
    T = syn.mat_to_params(np.linalg.inv(new.T_w_c_gt) @ win.frames[0].T_w_c_gt)
    s_empty.estimate(pyrs[2], 0, intr, T)
    capi.estimate_depths_batched([s_empty, s_empty], pyrs[2], 0, intr, np.stack([T, T]), np.ones(2), np.zeros((2, 2)))
    assert len(s_empty.download()["status"]) == 0
    with pytest.raises(capi.HipError):
        capi.estimate_depths_batched([], pyrs[2], 0, intr, np.zeros((0, 7)), np.zeros(0), np.zeros((0, 2)))
    # reference depth maps of a window without landmarks: all-zero maps, optical flow of nothing = NaN (0 / 0 in the reference)
    maps = g.create_reference_depth_maps(2)
    ids, wgt = maps.get_level(0)
    assert not ids.any() and not wgt.any()
    assert np.isnan(maps.mean_square_optical_flow(0, intr, [T])[0])
    # aligning against the empty maps: no valid residual, zero iterations, failure reported through rmse
    a = capi.HipAligner()
    rl = np.full(2, 1e10)
    r = a.estimate_pose(win.frames[1].timestamp, syn.mat_to_params(win.frames[1].T_w_c_gt), pyrs[1], maps, 1.0, np.zeros(2), new.timestamp, pyrs[2], 1.0,
                        intr, syn.mat_to_params(new.T_w_c_gt)[None, :], np.zeros(2), rl)
    assert r["lm_iterations"] == 0 and not r["success"]
    for o in [a, maps, s_empty, g] + pyrs:
        o.close()
