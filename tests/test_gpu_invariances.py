"""Size-independent properties of the photometric bundle adjustment, checked on the HIP path alone at the full C1 size
(7 keyframes, 2000 points, 640x480) — statements that hold for the reference's energy by construction and do not involve
the oracle:
  * gauge: moving every keyframe by one rigid transform G (T_i -> G T_i) changes no relative pose, so energies, the number
    of valid residuals and the solved relative poses are unchanged;
  * monocular scale: t_i -> s t_i with idepth -> idepth / s reprojects every pattern pixel to the same place;
  * photometric gauge: an exposure factor common to all frames cancels in (e_t / e_r);
  * landmark order: permuting the landmarks of a frame (with their connection statuses) permutes nothing but the sums."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1():
    return syn.make_window(num_frames=7, num_points=2000, width=640, height=480, seed=0)


def _load(win, pose=lambda T: T, idepth=lambda d: d, exposure=1.0, perm=None):
    from dsopp_amd import capi
    g = capi.HipWindow(capi.default_pba_options())
    intr = win.scene.intrinsics
    for i, f in enumerate(win.frames):
        g.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(pose(f.T_w_c_init)), exposure, f.affine_init, f.fixed, False)
        p = np.arange(len(f.uv)) if perm is None else perm[i]
        g.set_landmarks(f.frame_id, f.uv[p], idepth(f.idepth_init[p]), f.patch[p], np.zeros(len(p), dtype=np.uint8))
        for j in range(i):
            h = win.frames[j]
            g.set_connection(h.frame_id, f.frame_id, np.zeros(len(h.uv), dtype=np.uint8))
            g.set_connection(f.frame_id, h.frame_id, np.zeros(len(f.uv), dtype=np.uint8))
    return g


def _solve(g, win):
    g.begin()
    e0, n0 = g.calculate_energy()
    e, it, nv = g.optimize()
    poses = [syn.params_to_mat(g.get_pose(f.frame_id)[0]) for f in win.frames]
    return dict(e0=e0, n0=n0, e=e, it=it, nv=nv, poses=poses)


@pytest.fixture(scope="module")
def base(c1):
    g = _load(c1)
    r = _solve(g, c1)
    g.close()
    return r


def _relative(poses):
    return [np.linalg.inv(poses[0]) @ T for T in poses]


def test_rigid_gauge(c1, base):
    G = syn.se3_exp(np.array([0.7, -1.3, 2.1, 0.4, -0.9, 0.25]))
    g = _load(c1, pose=lambda T: G @ T)
    r = _solve(g, c1)
    g.close()
    assert r["n0"] == base["n0"] and abs(r["e0"] - base["e0"]) <= 1e-9 * base["e0"]
    assert (r["it"], r["nv"]) == (base["it"], base["nv"]) and abs(r["e"] - base["e"]) <= 1e-7 * base["e"]
    for A, B in zip(_relative(r["poses"]), _relative(base["poses"])):
        assert np.abs(A - B).max() <= 1e-7


def test_monocular_scale(c1, base):
    s = 3.7

    def scaled(T):
        T = T.copy()
        T[:3, 3] *= s
        return T

    g = _load(c1, pose=scaled, idepth=lambda d: d / s)
    r = _solve(g, c1)
    g.close()
    assert r["n0"] == base["n0"] and abs(r["e0"] - base["e0"]) <= 1e-9 * base["e0"]
    assert (r["it"], r["nv"]) == (base["it"], base["nv"]) and abs(r["e"] - base["e"]) <= 1e-6 * base["e"]
    for A, B in zip(_relative(r["poses"]), _relative(base["poses"])):
        assert np.abs(A[:3, :3] - B[:3, :3]).max() <= 1e-6
        assert np.abs(A[:3, 3] / s - B[:3, 3]).max() <= 1e-6


def test_common_exposure_factor(c1, base):
    g = _load(c1, exposure=0.37)
    r = _solve(g, c1)
    g.close()
    assert r["n0"] == base["n0"] and abs(r["e0"] - base["e0"]) <= 1e-12 * base["e0"]
    assert abs(r["e"] - base["e"]) <= 1e-9 * base["e"]


def test_landmark_permutation(c1, base):
    rng = np.random.default_rng(1)
    perm = [rng.permutation(len(f.uv)) for f in c1.frames]
    g = _load(c1, perm=perm)
    r = _solve(g, c1)
    g.close()
    assert r["n0"] == base["n0"] and abs(r["e0"] - base["e0"]) <= 1e-11 * base["e0"]     # summation order only
    assert (r["it"], r["nv"]) == (base["it"], base["nv"]) and abs(r["e"] - base["e"]) <= 1e-8 * base["e"]
    for A, B in zip(r["poses"], base["poses"]):
        assert np.abs(A - B).max() <= 1e-8


def test_landmark_permutation_is_bitwise_in_the_deterministic_build(c1):
    """The device holds every batch of landmarks in its OWN order (32 x 32-pixel tiles, raster inside; ties by content — pba.hip:
    HostFrame::to_internal), so with the order-deterministic build (dsopp_hip_window_set_deterministic) the caller's order changes
    nothing at all: energies, poses and every landmark's inverse depth are bit-identical, and the getters answer in the caller's order."""
    rng = np.random.default_rng(2)
    perm = [rng.permutation(len(f.uv)) for f in c1.frames]
    res = []
    for p in (None, perm):
        g = _load(c1, perm=p)
        g.set_deterministic(True)
        r = _solve(g, c1)
        r["idepth"] = [g.get_landmarks(f.frame_id, False)["idepth"] for f in c1.frames]
        r["energy01"] = g.get_residuals(c1.frames[0].frame_id, c1.frames[1].frame_id)["energy"]
        g.close()
        res.append(r)
    a, b = res
    assert (a["e0"], a["n0"], a["e"], a["it"], a["nv"]) == (b["e0"], b["n0"], b["e"], b["it"], b["nv"])
    for A, B in zip(a["poses"], b["poses"]):
        assert np.array_equal(A, B)
    for i, (da, db) in enumerate(zip(a["idepth"], b["idepth"])):
        assert np.array_equal(da[perm[i]], db)   # landmark j of the permuted listing is landmark perm[j] of the original one
    assert np.array_equal(a["energy01"][perm[0]], b["energy01"])
