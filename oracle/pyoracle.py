"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper over oracle/libdsopp_oracle.so (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(dsopp_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdsopp_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp", ".h"))]
    stale = force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdsopp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Options(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("initial_trust_region_radius", C.c_double),
                ("function_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("affine_brightness_regularizer", C.c_double * 2), ("fixed_state_regularizer", C.c_double),
                ("sigma_huber_loss", C.c_double), ("estimate_uncertainty", C.c_int32), ("force_accept", C.c_int32),
                ("first_estimate_jacobians", C.c_int32), ("optimize_idepths", C.c_int32)]


class AlignResult(C.Structure):
    _fields_ = [("rmse", C.c_double), ("energy", C.c_double), ("n_valid", C.c_int32), ("iterations", C.c_int32),
                ("T_w_target", C.c_double * 7), ("affine_brightness", C.c_double * 2), ("covariance", C.c_double * 36),
                ("H", C.c_double * 64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_window_create.restype = C.c_void_p
        _lib.orc_window_create.argtypes = [C.POINTER(Options)]
        _lib.orc_window_destroy.argtypes = [C.c_void_p]
        _lib.orc_window_destroy.restype = None
    return _lib


def _p(a, dtype=np.float64):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_pba_options(**kw) -> Options:
    o = Options()
    lib().orc_default_pba_options(C.byref(o))
    for k, v in kw.items():
        if k == "affine_brightness_regularizer":
            o.affine_brightness_regularizer[0], o.affine_brightness_regularizer[1] = v
        else:
            setattr(o, k, v)
    return o


def default_align_options(**kw) -> Options:
    o = Options()
    lib().orc_default_align_options(C.byref(o))
    for k, v in kw.items():
        if k == "affine_brightness_regularizer":
            o.affine_brightness_regularizer[0], o.affine_brightness_regularizer[1] = v
        else:
            setattr(o, k, v)
    return o


def set_threads(n: int):
    lib().orc_set_threads(int(n))


class OracleWindow:
    """CPU restatement of EigenPhotometricBundleAdjustment behind the same Python interface as dsopp_amd.HipWindow."""

    def __init__(self, options: Options | None = None):
        self.options = options or default_pba_options()
        self._h = C.c_void_p(lib().orc_window_create(C.byref(self.options)))
        self._keep = []  # borrowed image / mask buffers
        self.frame_ids = []

    def close(self):
        if self._h:
            lib().orc_window_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def K(self) -> int:
        return 8 * lib().orc_window_num_frames(self._h)

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(f"oracle call failed: {rc}")
        return rc

    def push_frame(self, frame_id, timestamp, pixelinfo, mask, intrinsics, T_w_agent, exposure, affine, fixed, is_marginalized):
        pix = _f64(pixelinfo)
        H, W = pix.shape[:2]
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._keep.append((pix, m))
        self._chk(lib().orc_window_push_frame(self._h, int(frame_id), C.c_int64(int(timestamp)), W, H, _p(pix),
                                              _p(m, np.uint8), _p(_f64(intrinsics)), _p(_f64(T_w_agent)),
                                              C.c_double(exposure), _p(_f64(affine)), int(bool(fixed)), int(bool(is_marginalized))))
        self.frame_ids.append(int(frame_id))

    def set_landmarks(self, frame_id, uv, idepth, patch, flags):
        n = len(idepth)
        self._chk(lib().orc_window_set_landmarks(self._h, int(frame_id), n, _p(_f64(uv)), _p(_f64(idepth)), _p(_f64(patch)),
                                                 _p(np.ascontiguousarray(flags, dtype=np.uint8), np.uint8)))

    def set_connection(self, ref_id, tgt_id, statuses):
        st = np.ascontiguousarray(statuses, dtype=np.uint8)
        self._chk(lib().orc_window_set_connection(self._h, int(ref_id), int(tgt_id), len(st), _p(st, np.uint8)))

    def mark_frame_marginalized(self, frame_id):
        self._chk(lib().orc_window_mark_frame_marginalized(self._h, int(frame_id)))

    # --- stage level ---
    def begin(self):
        self._chk(lib().orc_window_begin(self._h))

    def calculate_energy(self):
        e, n = C.c_double(), C.c_int()
        self._chk(lib().orc_window_calculate_energy(self._h, C.byref(e), C.byref(n)))
        return e.value, n.value

    def linearize(self):
        self._chk(lib().orc_window_linearize(self._h))

    def get_system(self):
        K = self.K
        Hpp, bpp, Hsc, bsc = np.zeros((K, K)), np.zeros(K), np.zeros((K, K)), np.zeros(K)
        self._chk(lib().orc_window_get_system(self._h, _p(Hpp), _p(bpp), _p(Hsc), _p(bsc)))
        return Hpp, bpp, Hsc, bsc

    def calculate_step(self, lam):
        step = np.zeros(self.K)
        self._chk(lib().orc_window_calculate_step(self._h, C.c_double(lam), _p(step)))
        return step

    def accept_step(self):
        a, b = C.c_double(), C.c_double()
        self._chk(lib().orc_window_accept_step(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reject_step(self):
        self._chk(lib().orc_window_reject_step(self._h))

    def update_point_statuses(self):
        self._chk(lib().orc_window_update_point_statuses(self._h))

    def solve(self):
        e, it, nv = C.c_double(), C.c_int(), C.c_int()
        self._chk(lib().orc_window_solve(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def optimize(self):
        e, it, nv = C.c_double(), C.c_int(), C.c_int()
        self._chk(lib().orc_window_optimize(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def reset_state(self, frame_id, T_w_agent, affine, idepth):
        self._chk(lib().orc_window_reset_state(self._h, int(frame_id), _p(_f64(T_w_agent)), _p(_f64(affine)), _p(_f64(idepth))))

    # --- getters ---
    def get_frame_state(self, frame_id):
        T0, ab0, eps, step = np.zeros(7), np.zeros(2), np.zeros(8), np.zeros(8)
        self._chk(lib().orc_window_get_frame_state(self._h, int(frame_id), _p(T0), _p(ab0), _p(eps), _p(step)))
        return T0, ab0, eps, step

    def get_pose(self, frame_id):
        T, ab = np.zeros(7), np.zeros(2)
        self._chk(lib().orc_window_get_pose(self._h, int(frame_id), _p(T), _p(ab)))
        return T, ab

    def get_landmarks(self, frame_id, with_hpib=True):  # (with_hpib: signature parity with the HIP wrapper; the rows are always returned)
        n = self._chk(lib().orc_window_num_landmarks(self._h, int(frame_id)))
        K = self.K
        out = dict(idepth=np.zeros(n), idepth_step=np.zeros(n), inv_hdd=np.zeros(n), b_d=np.zeros(n),
                   relative_baseline=np.zeros(n), n_inliers=np.zeros(n, dtype=np.int32), flags=np.zeros(n, dtype=np.uint8),
                   hpib=np.zeros((n, K)))
        self._chk(lib().orc_window_get_landmarks(self._h, int(frame_id), _p(out["idepth"]), _p(out["idepth_step"]),
                                                 _p(out["inv_hdd"]), _p(out["b_d"]), _p(out["relative_baseline"]),
                                                 _p(out["n_inliers"], np.int32), _p(out["flags"], np.uint8), _p(out["hpib"])))
        return out

    def get_residuals(self, ref_id, tgt_id, full=False):
        n = self._chk(lib().orc_window_num_landmarks(self._h, int(ref_id)))
        out = dict(status=np.zeros(n, dtype=np.uint8), candidate=np.zeros(n, dtype=np.uint8), energy=np.zeros(n),
                   huber_weight=np.zeros(n))
        if full:
            out.update(residuals=np.zeros((n, 8)), J_ref=np.zeros((n, 8, 8)), J_tgt=np.zeros((n, 8, 8)), J_idepth=np.zeros((n, 8)))
        m = self._chk(lib().orc_window_get_residuals(
            self._h, int(ref_id), int(tgt_id), _p(out["status"], np.uint8), _p(out["candidate"], np.uint8), _p(out["energy"]),
            _p(out["huber_weight"]), _p(out.get("residuals")), _p(out.get("J_ref")), _p(out.get("J_tgt")), _p(out.get("J_idepth"))))
        return {k: v[:m] for k, v in out.items()}

    def get_marginalized(self):
        K = self.K
        H, b, e = np.zeros((K, K)), np.zeros(K), C.c_double()
        self._chk(lib().orc_window_get_marginalized(self._h, _p(H), _p(b), C.byref(e)))
        return H, b, e.value

    def get_covariance(self, ref_id, tgt_id):
        cov = np.zeros((6, 6))
        self._chk(lib().orc_window_get_covariance(self._h, int(ref_id), int(tgt_id), _p(cov)))
        return cov


def build_pyramid(image_u8, lut=None, vignetting=None, levels=4):
    img = np.ascontiguousarray(image_u8, dtype=np.uint8)
    H, W = img.shape
    levels = min(levels, 5)
    infos, planes = [], []
    w, h = W, H
    for _ in range(levels):
        infos.append(np.zeros((h, w, 3)))
        planes.append(np.zeros((h, w)))
        w //= 2
        h //= 2
    pi = (C.c_void_p * levels)(*[a.ctypes.data for a in infos])
    pp = (C.c_void_p * levels)(*[a.ctypes.data for a in planes])
    lutp = None if lut is None else _p(_f64(lut))
    vig = None if vignetting is None else np.ascontiguousarray(vignetting, dtype=np.uint8)
    lib().orc_build_pyramid(_p(img, np.uint8), W, H, lutp, _p(vig, np.uint8), levels, pi, pp)
    return infos, planes


def points_from_depth_map(pixelinfo, idepth_sum, weight):
    pix = _f64(pixelinfo)
    H, W = pix.shape[:2]
    cap = H * W
    u, v, d, inten = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap)
    n = lib().orc_points_from_depth_map(W, H, _p(pix), _p(_f64(idepth_sum)), _p(_f64(weight)), cap, _p(u), _p(v), _p(d), _p(inten))
    return u[:n].copy(), v[:n].copy(), d[:n].copy(), inten[:n].copy()


# (the status codes and the constructor-default struct of arrays are plain data: they live with the generators, re-exported here)
from dsopp_amd.synthetic import IMMATURE_STATUS, new_immature_landmarks  # noqa: E402,F401


def estimate_depths(lms, target_pixelinfo, mask, intrinsics, T_target_reference, reference_exposure=1.0, reference_affine=(0, 0),
                    target_exposure=1.0, target_affine=(0, 0), sigma_huber_loss=20.0):
    """DepthEstimation::estimate (depth_estimation.cpp:363-381); updates the landmark arrays in place"""
    pix = _f64(target_pixelinfo)
    H, W = pix.shape[:2]
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    n = len(lms["status"])
    for k in ("idepth_min", "idepth_max", "uniqueness", "search_pixel_interval"):
        lms[k] = _f64(lms[k])
    for k in ("status", "traced"):
        lms[k] = np.ascontiguousarray(lms[k], dtype=np.uint8)
    lib().orc_estimate_depths(W, H, _p(pix), _p(m, np.uint8), _p(_f64(intrinsics)), _p(_f64(T_target_reference)), C.c_double(reference_exposure),
                              _p(_f64(reference_affine)), C.c_double(target_exposure), _p(_f64(target_affine)), C.c_double(sigma_huber_loss), n,
                              _p(_f64(lms["projection"])), _p(_f64(lms["direction"])), _p(_f64(lms["patch"])), _p(_f64(lms["gradient"])),
                              _p(lms["idepth_min"]), _p(lms["idepth_max"]), _p(lms["uniqueness"]), _p(lms["search_pixel_interval"]),
                              _p(lms["status"], np.uint8), _p(lms["traced"], np.uint8))
    return lms


ACTIVATION_STATUS = dict(activate=0, skip=1, delete=2)


def activate_landmarks(frames, intrinsics, sigma_huber_loss=20.0, number_of_desired_points=2000, min_distance_to_neighbor=0.0, refine=True,
                       mask_sparsity_newest=None):
    """LandmarksActivator::activate (landmarks_activator.cpp:351-391).  frames: list of dicts, oldest first, newest keyframe
    last: pixelinfo (H x W x 3), mask (H x W u8 or None), T_w (7), exposure, affine (2); all but the last also active_uv,
    active_idepth, active_skip and `immature` (the dict of new_immature_landmarks, updated in place).
    Returns (statuses per frame, number_of_active_points, new min_distance_to_neighbor)."""
    F = len(frames)
    H, W = frames[0]["pixelinfo"].shape[:2]
    pix = [_f64(f["pixelinfo"]) for f in frames]
    masks = [None if f.get("mask") is None else np.ascontiguousarray(f["mask"], dtype=np.uint8) for f in frames]
    pix_arr = (C.c_void_p * F)(*[_p(p) for p in pix])
    mask_arr = (C.c_void_p * F)(*[_p(m, np.uint8) for m in masks])
    T_w = _f64(np.concatenate([_f64(f["T_w"]) for f in frames]))
    expo = _f64([f.get("exposure", 1.0) for f in frames])
    aff = _f64(np.concatenate([_f64(f.get("affine", (0, 0))) for f in frames]))
    old = frames[:-1]
    n_active = np.array([len(f["active_idepth"]) for f in old], dtype=np.int32)
    n_imm = np.array([len(f["immature"]["status"]) for f in old], dtype=np.int32)
    cat = lambda arrs, dt, w: np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=dt).reshape(-1, w) for a in arrs]).ravel(), dtype=dt)
    a_uv, a_id, a_skip = cat([f["active_uv"] for f in old], np.float64, 2), cat([f["active_idepth"] for f in old], np.float64, 1), \
        cat([f["active_skip"] for f in old], np.uint8, 1)
    im = lambda k, dt, w: cat([f["immature"][k] for f in old], dt, w)
    proj, patch = im("projection", np.float64, 2), im("patch", np.float64, 8)
    imin, imax, uniq, spi = im("idepth_min", np.float64, 1), im("idepth_max", np.float64, 1), im("uniqueness", np.float64, 1), \
        im("search_pixel_interval", np.float64, 1)
    status, traced = im("status", np.uint8, 1), im("traced", np.uint8, 1)
    act = np.zeros(int(n_imm.sum()), dtype=np.uint8)
    dist = C.c_double(min_distance_to_neighbor)
    ms = None if mask_sparsity_newest is None else np.ascontiguousarray(mask_sparsity_newest, dtype=np.uint8)
    fn = lib().orc_activate_landmarks
    fn.restype = C.c_int
    n_pts = fn(F, W, H, pix_arr, mask_arr, _p(ms, np.uint8), _p(T_w), _p(expo), _p(aff), _p(_f64(intrinsics)), _p(n_active, np.int32), _p(a_uv),
               _p(a_id), _p(a_skip, np.uint8), _p(n_imm, np.int32), _p(proj), _p(patch), _p(imin), _p(imax), _p(uniq), _p(spi),
               _p(status, np.uint8), _p(traced, np.uint8), C.c_double(sigma_huber_loss), int(number_of_desired_points), C.byref(dist),
               int(bool(refine)), _p(act, np.uint8))
    out, o = [], 0
    for f, n in zip(old, n_imm):
        f["immature"]["idepth_min"], f["immature"]["idepth_max"] = imin[o:o + n].copy(), imax[o:o + n].copy()
        f["immature"]["status"] = status[o:o + n].copy()
        out.append(act[o:o + n].copy())
        o += n
    return out, n_pts, dist.value


def build_epipolar_segment(width, height, intrinsics, T_target_reference, observed, idepth_min=0.0, idepth_max=1000.0, cap=8192):
    proj, idp = np.zeros((cap, 2)), np.zeros(cap)
    n = lib().orc_build_epipolar_segment(int(width), int(height), _p(_f64(intrinsics)), _p(_f64(T_target_reference)), _p(_f64(observed)),
                                         C.c_double(idepth_min), C.c_double(idepth_max), cap, _p(proj), _p(idp))
    return proj[:min(n, cap)].copy(), idp[:min(n, cap)].copy()


def create_reference_depth_maps(sources, T_w_newest, intrinsics, width, height, levels):
    """createReferenceDepthMaps (create_depth_maps.cpp:18-147).  sources: list of dicts with T_w (7), uv (n x 2), idepth,
    variance, skip (outlier | marginalized), status (connection statuses towards the newest keyframe).
    Returns [(idepth_sum, weight)] per level, H_l x W_l each."""
    counts = np.array([len(s["idepth"]) for s in sources], dtype=np.int32)
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(s[k], dtype=dt).reshape(len(s["idepth"]), -1) for s in sources]).ravel()
                                             if len(sources) else np.zeros(0, dtype=dt), dtype=dt)
    Ts = np.ascontiguousarray(np.concatenate([_f64(s["T_w"]) for s in sources]) if len(sources) else np.zeros(0))
    sizes, w, h = [], int(width), int(height)
    for _ in range(levels):
        sizes.append((h, w))
        w, h = w // 2, h // 2
    total = sum(a * b for a, b in sizes)
    ids, wgt = np.zeros(total), np.zeros(total)
    uv, idp, var = cat("uv", np.float64), cat("idepth", np.float64), cat("variance", np.float64)
    skip, status = cat("skip", np.uint8), cat("status", np.uint8)
    n = lib().orc_create_reference_depth_maps(len(sources), _p(Ts), counts.ctypes.data_as(C.POINTER(C.c_int32)), _p(uv), _p(idp), _p(var),
                                              _p(skip, np.uint8), _p(status, np.uint8), _p(_f64(T_w_newest)), _p(_f64(intrinsics)),
                                              int(width), int(height), int(levels), _p(ids), _p(wgt))
    assert n == levels
    out, o = [], 0
    for (hh, ww) in sizes:
        out.append((ids[o:o + hh * ww].reshape(hh, ww).copy(), wgt[o:o + hh * ww].reshape(hh, ww).copy()))
        o += hh * ww
    return out


def initialization_poses(T_w_previous, T_w_last, T_w_keyframe):
    """initializationPoses (monocular_tracker.cpp:136-176); T_w_previous None = fewer than two frames in the track"""
    out = np.zeros((128, 7))
    if T_w_previous is None:
        n = lib().orc_initialization_poses(0, None, None, None, 128, _p(out))
    else:
        n = lib().orc_initialization_poses(1, _p(_f64(T_w_previous)), _p(_f64(T_w_last)), _p(_f64(T_w_keyframe)), 128, _p(out))
    return out[:n].copy()


def se3_log(T):
    xi = np.zeros(6)
    lib().orc_se3_log(_p(_f64(T)), _p(xi))
    return xi


def mean_square_optical_flow(idepth_sum, weight, intrinsics, T_target_reference):
    """calculateMeanSquareOpticalFlow (monocular_tracker.cpp:104-134) of one depth-map level"""
    ids, wgt = _f64(idepth_sum), _f64(weight)
    H, W = ids.shape
    fn = lib().orc_mean_square_optical_flow
    fn.restype = C.c_double
    return fn(W, H, _p(ids), _p(wgt), _p(_f64(intrinsics)), _p(_f64(T_target_reference)))


def align_solve(options, u, v, idepth, intensity, ref_intr, ref_size, T_w_ref, ref_exposure, ref_ab, tgt_intr, tgt_pixelinfo,
                tgt_mask, T_w_tgt_init, tgt_exposure, tgt_ab, rotation_prior=None):
    pix = _f64(tgt_pixelinfo)
    H, W = pix.shape[:2]
    m = None if tgt_mask is None else np.ascontiguousarray(tgt_mask, dtype=np.uint8)
    out = AlignResult()
    prior = None if rotation_prior is None else _f64(np.asarray(rotation_prior).reshape(9))
    rc = lib().orc_align_solve_with_prior(C.byref(options), len(u), _p(_f64(u)), _p(_f64(v)), _p(_f64(idepth)), _p(_f64(intensity)),
                                          _p(_f64(ref_intr)), int(ref_size[0]), int(ref_size[1]), _p(_f64(T_w_ref)), C.c_double(ref_exposure),
                                          _p(_f64(ref_ab)), _p(_f64(tgt_intr)), W, H, _p(pix), _p(m, np.uint8), _p(_f64(T_w_tgt_init)),
                                          C.c_double(tgt_exposure), _p(_f64(tgt_ab)), _p(prior), C.byref(out))
    if rc < 0:
        raise RuntimeError(f"orc_align_solve failed {rc}")
    return dict(rmse=out.rmse, energy=out.energy, n_valid=out.n_valid, iterations=out.iterations,
                T_w_target=np.array(out.T_w_target), affine_brightness=np.array(out.affine_brightness),
                covariance=np.array(out.covariance).reshape(6, 6), H=np.array(out.H).reshape(8, 8))


def se3_exp(xi):
    T = np.zeros(7)
    lib().orc_se3_exp(_p(_f64(xi)), _p(T))
    return T


def se3_mul(A, B):
    Cc = np.zeros(7)
    lib().orc_se3_mul(_p(_f64(A)), _p(_f64(B)), _p(Cc))
    return Cc


def se3_inverse(A):
    B = np.zeros(7)
    lib().orc_se3_inverse(_p(_f64(A)), _p(B))
    return B


def se3_adj(A):
    M = np.zeros((6, 6))
    lib().orc_se3_adj(_p(_f64(A)), _p(M))
    return M


def reproject_pattern(ref_intr, ref_size, tgt_intr, tgt_size, T_t_r, u, v, idepth, with_jacobians=True):
    n = len(u)
    tu, tv = np.zeros(n), np.zeros(n)
    dui, dvi, duT, dvT = np.zeros(n), np.zeros(n), np.zeros((n, 6)), np.zeros((n, 6))
    ok = lib().orc_reproject_pattern(_p(_f64(ref_intr)), int(ref_size[0]), int(ref_size[1]), _p(_f64(tgt_intr)), int(tgt_size[0]),
                                     int(tgt_size[1]), _p(_f64(T_t_r)), n, _p(_f64(u)), _p(_f64(v)), C.c_double(idepth),
                                     int(with_jacobians), _p(tu), _p(tv), _p(dui), _p(dvi), _p(duT), _p(dvT))
    return bool(ok), tu, tv, dui, dvi, duT, dvT


def solve_system(H, b):
    n = len(b)
    x = np.zeros(n)
    lib().orc_solve_system(n, _p(_f64(H)), _p(_f64(b)), _p(x))
    return x


def reduce_system(H, b, elim):
    n = len(b)
    e = np.ascontiguousarray(elim, dtype=np.int32)
    nk = n - len(e)
    Ho, bo = np.zeros((nk, nk)), np.zeros(nk)
    lib().orc_reduce_system(n, _p(_f64(H)), _p(_f64(b)), len(e), _p(e, np.int32), _p(Ho), _p(bo))
    return Ho, bo


def pinv_drop(H, nullspaces):
    n = H.shape[0]
    out = np.zeros((n, n))
    lib().orc_pinv_drop(n, _p(_f64(H)), int(nullspaces), _p(out))
    return out


def relative_transformation_uncertainty(T_w_1, T_w_2, sigma_11, sigma_22, sigma_12):
    """Motion::relativeTransformationUncertainty (se3_motion.hpp:151-158) as the oracle restates it"""
    out = np.zeros((6, 6))
    lib().orc_relative_transformation_uncertainty(_p(_f64(T_w_1)), _p(_f64(T_w_2)), _p(_f64(sigma_11)), _p(_f64(sigma_22)), _p(_f64(sigma_12)), _p(out))
    return out


def ldlt_solve(A, b):
    """Eigen::LDLT (pivoted) as the oracle restates it, on its own (oracle/linalg.hpp: ldltSolve)"""
    n = len(b)
    x = np.zeros(n)
    lib().orc_ldlt_solve(n, _p(_f64(A)), _p(_f64(b)), _p(x))
    return x


def pinv_cod(H):
    """completeOrthogonalDecomposition().pseudoInverse() of a symmetric matrix as the oracle restates it"""
    n = H.shape[0]
    out = np.zeros((n, n))
    lib().orc_pinv_cod(n, _p(_f64(H)), _p(out))
    return out


def interpolate_linear(pixelinfo, x, y):
    """PixelMap<1>::Evaluate -> interpolateLinear<true,1> (pixel_map.hpp:20-40) at n positions: (n, 3) = (I, Ix, Iy)"""
    pix = _f64(pixelinfo)
    H, W = pix.shape[:2]
    x, y = _f64(x), _f64(y)
    out = np.zeros((len(x), 3))
    lib().orc_interpolate_linear(W, H, _p(pix), len(x), _p(x), _p(y), _p(out))
    return out


def mask_valid(mask, x, y):
    """CameraMask::valid at the rounded position with the border check (camera_mask.hpp:48-89)"""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = m.shape
    x, y = _f64(x), _f64(y)
    out = np.zeros(len(x), dtype=np.uint8)
    lib().orc_mask_valid(W, H, _p(m, np.uint8), len(x), _p(x), _p(y), _p(out, np.uint8))
    return out.astype(bool)
