"""GPU parity tests of the coarse-tracker path: image pyramid construction and two-frame direct
image alignment per pyramid level, HIP path vs CPU oracle on identical synthetic frames."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def two_frames():
    return syn.make_window(num_frames=2, num_points=20, width=320, height=240, seed=7)


def test_pyramid_bit_exact(two_frames):
    """f64 pyramid == scalar definition bit for bit (the reference asserts its AVX2 path against the same definition with
    EXPECT_EQ, test/test/features/test_dxdy_accelerated.cpp:11-85), incl. LUT + vignette correction."""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    img = two_frames.frames[0].image_u8
    H, W = img.shape
    rng = np.random.default_rng(0)
    lut = np.cumsum(rng.uniform(0.5, 1.5, 256))
    vig = rng.integers(120, 255, size=(H, W)).astype(np.uint8)
    for kwargs in (dict(), dict(lut=lut), dict(lut=lut, vignetting=vig)):
        infos, _ = po.build_pyramid(img, levels=4, **kwargs)
        p = capi.Pyramid(W, H, levels=4)
        p.build(img, **kwargs)
        for l in range(4):
            got = p.get_level(l)
            assert got.shape == infos[l].shape
            assert np.array_equal(got, infos[l]), (kwargs.keys(), l, np.abs(got - infos[l]).max())
        p.close()
    # odd mask + float storage mode: fp32 round-off class
    p = capi.Pyramid(W, H, levels=3, dtype=capi.F32)
    p.build(img)
    infos, _ = po.build_pyramid(img, levels=3)
    for l in range(3):
        assert np.abs(p.get_level(l) - infos[l]).max() <= 1e-4
    p.close()


@pytest.mark.parametrize("size", [(326, 250), (643, 481), (97, 70), (1280, 1024)])
def test_pyramid_one_launch_is_bit_exact_at_any_size(size):
    """without a vignette every level is built in ONE launch (region workgroups regenerate the upper levels from the 8-bit image with
    the chain's nested 2 x 2 means): bit-identical to the scalar definition for 1 .. 5 levels, at sizes that are not multiples of
    the region or of 2^levels (odd widths drop their last column at every halving, downscale_image.hpp:16-33), with and without
    the photometric LUT, and after a mask was set (the mask lane survives a rebuild)"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    W, H = size
    rng = np.random.default_rng(W + H)
    img = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    lut = np.cumsum(rng.uniform(0.5, 1.5, 256))
    for levels in ((5, 3, 1) if W > 1000 else (5, 4, 3, 2, 1)):
        if (W >> (levels - 1)) < 3 or (H >> (levels - 1)) < 3:
            continue
        for kwargs in (dict(), dict(lut=lut)):
            infos, _ = po.build_pyramid(img, levels=levels, **kwargs)
            p = capi.Pyramid(W, H, levels=levels)
            mask = (rng.uniform(size=(H >> (levels - 1), W >> (levels - 1))) > 0.2).astype(np.uint8)
            p.set_mask(levels - 1, mask)
            p.build(img, **kwargs)
            for l in range(levels):
                got = p.get_level(l)
                assert got.shape == infos[l].shape
                assert np.array_equal(got, infos[l]), (size, levels, kwargs.keys(), l, np.abs(got - infos[l]).max())
            p.close()


def _depth_map(frame, level, n, seed):
    H, W = frame.depth.shape
    h, w = H >> level, W >> level
    rng = np.random.default_rng(seed)
    idsum, wgt = np.zeros((h, w)), np.zeros((h, w))
    xs, ys = rng.integers(0, w, n), rng.integers(0, h, n)  # includes border pixels the scan must skip
    idsum[ys, xs] = 1.0 / frame.depth[np.minimum(ys << level, H - 1), np.minimum(xs << level, W - 1)]
    wgt[ys, xs] = 1.0
    return idsum, wgt


@pytest.mark.parametrize("lm_path", [0, 1])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_alignment_parity(two_frames, level, lm_path):
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = two_frames
    fr, ft = win.frames
    H, W = fr.image_u8.shape
    infos_r, _ = po.build_pyramid(fr.image_u8, levels=3)
    infos_t, _ = po.build_pyramid(ft.image_u8, levels=3)
    pr, pt = capi.Pyramid(W, H, 3), capi.Pyramid(W, H, 3)
    pr.build(fr.image_u8)
    pt.build(ft.image_u8)
    intr = win.scene.intrinsics / (1 << level)  # CameraCalibration::cameraModel(level), camera_calibration.cpp:66-70
    # level 2 stays below 1024 reference points: with lm_path 0 it runs the single-workgroup loop kernel, levels 0 / 1 the
    # launch-per-iteration kernel
    idsum, wgt = _depth_map(fr, level, 700 if level == 2 else 1500, seed=level)
    T_ref = syn.mat_to_params(fr.T_w_c_gt)
    T_init = syn.mat_to_params(ft.T_w_c_init)
    # oracle
    u, v, idp, inten = po.points_from_depth_map(infos_r[level], idsum, wgt)
    h, w = infos_r[level].shape[:2]
    ro = po.align_solve(po.default_align_options(), u, v, idp, inten, intr, (w, h), T_ref, 1.0, np.zeros(2), intr, infos_t[level], None,
                        T_init, 1.0, np.zeros(2))
    # HIP
    a = capi.HipAligner(capi.default_align_options())
    a.set_lm_path(lm_path)   # 0: the whole LM loop in one launch of one workgroup; 1: one launch per iteration
    a.reset()
    a.push_reference_depth_map(1000, T_ref, pr, level, intr, idsum, wgt, 1.0, np.zeros(2))
    assert a.num_points() == len(u) and (len(u) <= 1024) == (level == 2)
    a.push_target(2000, T_init, pt, level, intr, 1.0, np.zeros(2))
    rg = a.solve()
    assert rg["iterations"] == ro["iterations"], (rg["iterations"], ro["iterations"])
    assert rg["n_valid"] == ro["n_valid"]
    assert abs(rg["energy"] - ro["energy"]) <= 1e-8 * abs(ro["energy"])
    assert abs(rg["rmse"] - ro["rmse"]) <= 1e-8 * ro["rmse"]
    assert np.abs(rg["T_w_target"] - ro["T_w_target"]).max() <= 1e-8
    assert np.abs(rg["affine_brightness"] - ro["affine_brightness"]).max() <= 1e-8
    assert np.abs(rg["H"] - ro["H"]).max() <= 1e-8 * np.abs(ro["H"]).max()
    assert np.abs(rg["covariance"] - ro["covariance"]).max() <= 1e-6 * np.abs(ro["covariance"]).max()
    # and it actually aligns: closer to ground truth than the initial guess (reference bar: 5e-2 / 1 degree)
    gt = syn.mat_to_params(ft.T_w_c_gt)
    if level == 0:
        assert np.abs(rg["T_w_target"] - gt).max() < 0.5 * np.abs(T_init - gt).max()
    a.close()
    pr.close()
    pt.close()


def test_alignment_known_pose_and_masks(two_frames):
    from dsopp_amd import capi
    win = two_frames
    fr, ft = win.frames
    H, W = fr.image_u8.shape
    pr, pt = capi.Pyramid(W, H, 1), capi.Pyramid(W, H, 1)
    pr.build(fr.image_u8)
    pt.build(ft.image_u8)
    intr = win.scene.intrinsics
    idsum, wgt = _depth_map(fr, 0, 800, seed=3)
    a = capi.HipAligner()
    T_ref, T_t = syn.mat_to_params(fr.T_w_c_gt), syn.mat_to_params(ft.T_w_c_gt)
    a.push_reference_depth_map(1000, T_ref, pr, 0, intr, idsum, wgt, 1.0, np.zeros(2))
    a.push_target(2000, syn.mat_to_params(ft.T_w_c_init), pt, 0, intr, 1.0, np.zeros(2))
    a.push_known_pose(2000, T_t)
    r = a.solve()
    assert r["rmse"] == -1 and np.abs(r["T_w_target"] - T_t).max() < 1e-12  # kZeroCost path
    # a fully masked target leaves no valid residual
    a2 = capi.HipAligner()
    pt.set_mask(0, np.zeros((H, W), dtype=np.uint8))
    a2.push_reference_depth_map(1000, T_ref, pr, 0, intr, idsum, wgt, 1.0, np.zeros(2))
    a2.push_target(2000, syn.mat_to_params(ft.T_w_c_init), pt, 0, intr, 1.0, np.zeros(2))
    r2 = a2.solve()
    assert r2["n_valid"] == 0 and r2["iterations"] == 0
    for o in (a, a2, pr, pt):
        o.close()


@pytest.mark.parametrize("weak_prior", [False, True])
def test_alignment_photometric_parameters(two_frames, weak_prior):
    """exposure ratio != 1, non-zero affine brightness of both frames (the brightness-change scale, the a / b columns of the
    8x8 system and the affine prior block of eigen_pose_alignment.cpp:101-104,183-187)"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = two_frames
    fr, ft = win.frames
    H, W = fr.image_u8.shape
    level = 1
    infos_r, _ = po.build_pyramid(fr.image_u8, levels=2)
    infos_t, _ = po.build_pyramid(ft.image_u8, levels=2)
    pr, pt = capi.Pyramid(W, H, 2), capi.Pyramid(W, H, 2)
    pr.build(fr.image_u8)
    pt.build(ft.image_u8)
    intr = win.scene.intrinsics / (1 << level)
    idsum, wgt = _depth_map(fr, level, 1200, seed=11)
    T_ref, T_init = syn.mat_to_params(fr.T_w_c_gt), syn.mat_to_params(ft.T_w_c_init)
    ab_ref, ab_tgt = np.array([0.02, 1.5]), np.array([-0.01, -0.7])
    e_ref, e_tgt = 0.8, 1.1
    kw = dict(affine_brightness_regularizer=(1e1, 1e-3)) if weak_prior else {}
    u, v, idp, inten = po.points_from_depth_map(infos_r[level], idsum, wgt)
    h, w = infos_r[level].shape[:2]
    ro = po.align_solve(po.default_align_options(**kw), u, v, idp, inten, intr, (w, h), T_ref, e_ref, ab_ref, intr, infos_t[level], None, T_init,
                        e_tgt, ab_tgt)
    a = capi.HipAligner(capi.default_align_options(**kw))
    a.reset()
    a.push_reference_depth_map(1000, T_ref, pr, level, intr, idsum, wgt, e_ref, ab_ref)
    a.push_target(2000, T_init, pt, level, intr, e_tgt, ab_tgt)
    rg = a.solve()
    assert rg["iterations"] == ro["iterations"] and rg["n_valid"] == ro["n_valid"]
    assert abs(rg["energy"] - ro["energy"]) <= 1e-8 * abs(ro["energy"])
    assert np.abs(rg["T_w_target"] - ro["T_w_target"]).max() <= 1e-8
    assert np.abs(rg["affine_brightness"] - ro["affine_brightness"]).max() <= 1e-7
    assert np.abs(rg["H"] - ro["H"]).max() <= 1e-8 * np.abs(ro["H"]).max()
    if weak_prior:
        assert np.abs(rg["affine_brightness"] - ab_tgt).max() > 1e-3   # the photometric parameters really moved
    for obj in (a, pr, pt):
        obj.close()


def test_alignment_rotation_prior(two_frames):
    """setRotationPrior (eigen_pose_alignment.cpp:254-257,309-311): the rotation of t_target_reference is replaced by the prior
    projected onto SO(3) before the solve; reset() drops it.  The projection is checked against NumPy's SVD statement."""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = two_frames
    fr, ft = win.frames
    H, W = fr.image_u8.shape
    level = 1
    infos_r, _ = po.build_pyramid(fr.image_u8, levels=2)
    infos_t, _ = po.build_pyramid(ft.image_u8, levels=2)
    pr, pt = capi.Pyramid(W, H, 2), capi.Pyramid(W, H, 2)
    pr.build(fr.image_u8)
    pt.build(ft.image_u8)
    intr = win.scene.intrinsics / 2
    idsum, wgt = _depth_map(fr, level, 1500, seed=3)
    T_ref, T_init = syn.mat_to_params(fr.T_w_c_gt), syn.mat_to_params(ft.T_w_c_init)
    R_gt = (np.linalg.inv(ft.T_w_c_gt) @ fr.T_w_c_gt)[:3, :3]
    rng = np.random.default_rng(0)
    prior = R_gt + rng.normal(0, 1e-3, (3, 3))                  # not exactly a rotation: fitToSO3 has work to do
    U, _, Vt = np.linalg.svd(prior)
    R_fit = U @ np.diag([1, 1, np.linalg.det(U @ Vt)]) @ Vt
    u, v, idp, inten = po.points_from_depth_map(infos_r[level], idsum, wgt)
    h, w = infos_r[level].shape[:2]
    args = (po.default_align_options(), u, v, idp, inten, intr, (w, h), T_ref, 1.0, np.zeros(2), intr, infos_t[level], None, T_init, 1.0, np.zeros(2))
    r_plain = po.align_solve(*args)
    r_prior = po.align_solve(*args, rotation_prior=prior)
    r_fit = po.align_solve(*args, rotation_prior=R_fit)
    assert np.abs(r_prior["T_w_target"] - r_fit["T_w_target"]).max() <= 1e-9     # oracle's projection == SVD projection
    assert np.abs(r_prior["T_w_target"] - r_plain["T_w_target"]).max() > 1e-7    # the prior is not a no-op
    a = capi.HipAligner(capi.default_align_options())
    a.set_lm_path(1)
    res = []
    for use_prior in (True, False):      # the second pass runs after reset(): the prior must be gone
        a.reset()
        if use_prior:
            a.set_rotation_prior(prior)
        a.push_reference_depth_map(1000, T_ref, pr, level, intr, idsum, wgt, 1.0, np.zeros(2))
        a.push_target(2000, T_init, pt, level, intr, 1.0, np.zeros(2))
        res.append(a.solve())
    for rg, ro in zip(res, (r_prior, r_plain)):
        assert rg["iterations"] == ro["iterations"] and rg["n_valid"] == ro["n_valid"]
        assert np.abs(rg["T_w_target"] - ro["T_w_target"]).max() <= 1e-8
        assert abs(rg["rmse"] - ro["rmse"]) <= 1e-8 * ro["rmse"]
    with pytest.raises(capi.HipError):
        a.set_rotation_prior(np.diag([1.0, 1.0, -1.0]))
    for obj in (a, pr, pt):
        obj.close()
