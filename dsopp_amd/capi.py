"""ctypes binding of the product C-ABI (include/dsopp_hip.h -> dsopp_amd/lib/libdsopp_hip.so).

Python is only plumbing here (tests, bench, torch.distributed glue): every compute call goes through the C-ABI into
hand-written HIP kernels.  There is NO fallback: a missing library raises at import of the symbols, and every compute
entry point fails with DSOPP_HIP_ERR_HIP when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSOPP_HIP_LIB: measurement aid — another build of the same library (an earlier round's, for before / after counters in one run)
LIB_PATH = os.environ.get("DSOPP_HIP_LIB") or os.path.join(_HERE, "lib", "libdsopp_hip.so")

F64, F32 = 0, 1
NUM_KERNEL_CLASSES = 11


class Options(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("initial_trust_region_radius", C.c_double),
                ("function_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("affine_brightness_regularizer", C.c_double * 2), ("fixed_state_regularizer", C.c_double),
                ("sigma_huber_loss", C.c_double), ("estimate_uncertainty", C.c_int32), ("force_accept", C.c_int32),
                ("first_estimate_jacobians", C.c_int32), ("optimize_idepths", C.c_int32), ("dtype", C.c_int32)]


class AlignResult(C.Structure):
    _fields_ = [("rmse", C.c_double), ("energy", C.c_double), ("n_valid", C.c_int32), ("iterations", C.c_int32),
                ("T_world_target", C.c_double * 7), ("affine_brightness", C.c_double * 2), ("covariance", C.c_double * 36),
                ("H", C.c_double * 64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# every symbol include/dsopp_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "dsopp_hip_pyramid_group_build", "dsopp_hip_pyramid_group_create", "dsopp_hip_pyramid_group_destroy", "dsopp_hip_pyramid_group_get", "dsopp_hip_pyramid_group_set_level", "dsopp_hip_pyramid_group_set_mask", "dsopp_hip_window_group_accept_step", "dsopp_hip_window_group_begin", "dsopp_hip_window_group_calculate_energy", "dsopp_hip_window_group_calculate_step", "dsopp_hip_window_group_create", "dsopp_hip_window_group_create_reference_depth_maps", "dsopp_hip_window_group_destroy", "dsopp_hip_window_group_frame_ids", "dsopp_hip_window_group_get_covariance", "dsopp_hip_window_group_get_frame_state", "dsopp_hip_window_group_get_frame_update", "dsopp_hip_window_group_get_landmarks", "dsopp_hip_window_group_get_marginalized", "dsopp_hip_window_group_get_pose", "dsopp_hip_window_group_get_residuals", "dsopp_hip_window_group_get_system", "dsopp_hip_window_group_last_solve_ms", "dsopp_hip_window_group_linearize", "dsopp_hip_window_group_mark_frame_marginalized", "dsopp_hip_window_group_num_frames", "dsopp_hip_window_group_num_landmarks", "dsopp_hip_window_group_optimize", "dsopp_hip_window_group_optimize_repeated", "dsopp_hip_window_group_push_frame", "dsopp_hip_window_group_refill_reference_depth_maps", "dsopp_hip_window_group_reject_step", "dsopp_hip_window_group_restore", "dsopp_hip_window_group_set_connection", "dsopp_hip_window_group_set_deterministic", "dsopp_hip_window_group_set_landmarks", "dsopp_hip_window_group_set_lm_mode", "dsopp_hip_window_group_set_max_iterations", "dsopp_hip_window_group_shard", "dsopp_hip_window_group_size", "dsopp_hip_window_group_snapshot", "dsopp_hip_window_group_solve", "dsopp_hip_window_group_update_point_statuses",
    "dsopp_hip_window_frame_ids", "dsopp_hip_window_set_deterministic",
    "dsopp_hip_comm_unique_id", "dsopp_hip_comm_create", "dsopp_hip_comm_adopt", "dsopp_hip_comm_destroy", "dsopp_hip_comm_abort", "dsopp_hip_comm_rank",
    "dsopp_hip_comm_allreduce", "dsopp_hip_window_set_comm",
    "dsopp_hip_window_optimize_async", "dsopp_hip_window_optimize_wait",
    "dsopp_hip_immature_sets_estimate",
    "dsopp_hip_aligner_set_rotation_prior",
    "dsopp_hip_window_refill_reference_depth_maps",
    "dsopp_hip_initialization_poses",
    "dsopp_hip_depth_maps_mean_square_optical_flow",
    "dsopp_hip_window_activate_landmarks",
    "dsopp_hip_last_error", "dsopp_hip_device_count", "dsopp_hip_version", "dsopp_hip_default_pba_options",
    "dsopp_hip_default_align_options", "dsopp_hip_pyramid_create", "dsopp_hip_pyramid_destroy", "dsopp_hip_pyramid_build",
    "dsopp_hip_pyramid_build_device", "dsopp_hip_pyramid_set_level", "dsopp_hip_pyramid_set_mask", "dsopp_hip_pyramid_get_level",
    "dsopp_hip_pyramid_level_size", "dsopp_hip_window_create", "dsopp_hip_window_destroy", "dsopp_hip_window_push_frame",
    "dsopp_hip_window_set_landmarks", "dsopp_hip_window_set_connection", "dsopp_hip_window_mark_frame_marginalized",
    "dsopp_hip_window_num_frames", "dsopp_hip_window_solve", "dsopp_hip_window_begin", "dsopp_hip_window_calculate_energy",
    "dsopp_hip_window_linearize", "dsopp_hip_window_get_system", "dsopp_hip_window_calculate_step", "dsopp_hip_window_accept_step",
    "dsopp_hip_window_reject_step", "dsopp_hip_window_update_point_statuses", "dsopp_hip_window_get_frame_state",
    "dsopp_hip_window_get_pose", "dsopp_hip_window_num_landmarks", "dsopp_hip_window_get_landmarks", "dsopp_hip_window_get_residuals",
    "dsopp_hip_window_get_marginalized", "dsopp_hip_window_get_covariance", "dsopp_hip_window_set_allreduce",
    "dsopp_hip_window_last_solve_ms", "dsopp_hip_window_optimize", "dsopp_hip_window_optimize_repeated", "dsopp_hip_window_get_frame_update", "dsopp_hip_window_create_reference_depth_maps", "dsopp_hip_depth_maps_destroy", "dsopp_hip_depth_maps_level_size", "dsopp_hip_depth_maps_get_level", "dsopp_hip_aligner_push_reference_depth_maps", "dsopp_hip_aligner_estimate_pose", "dsopp_hip_aligner_set_hypothesis_width", "dsopp_hip_aligner_set_lm_path", "dsopp_hip_estimate_depths", "dsopp_hip_immature_set_create", "dsopp_hip_immature_set_destroy", "dsopp_hip_immature_set_upload_state", "dsopp_hip_immature_set_download_state", "dsopp_hip_immature_set_estimate", "dsopp_hip_window_set_max_iterations", "dsopp_hip_window_set_lm_mode", "dsopp_hip_window_time_kernel", "dsopp_hip_window_snapshot", "dsopp_hip_window_restore",
    "dsopp_hip_window_set_profiling", "dsopp_hip_window_get_profile", "dsopp_hip_kernel_class_name", "dsopp_hip_aligner_create", "dsopp_hip_aligner_destroy", "dsopp_hip_aligner_reset",
    "dsopp_hip_aligner_push_reference_depth_map", "dsopp_hip_aligner_push_reference_points", "dsopp_hip_aligner_push_target",
    "dsopp_hip_aligner_push_known_pose", "dsopp_hip_aligner_solve", "dsopp_hip_aligner_num_points",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                               "dsopp_amd has no CPU fallback")
        # PyTorch ships its own HIP runtime; if it is imported but its CUDA state is not up yet, bring it up BEFORE this library
        # touches the device: initialised second, torch reports "No HIP GPUs are available" (the other order works)
        torch = sys.modules.get("torch")
        if torch is not None:
            try:
                if torch.cuda.is_available() and not torch.cuda.is_initialized():
                    torch.cuda.init()
            except Exception:
                pass
        _lib = C.CDLL(LIB_PATH)
        _lib.dsopp_hip_last_error.restype = C.c_char_p
        _lib.dsopp_hip_version.restype = C.c_char_p
        _lib.dsopp_hip_kernel_class_name.restype = C.c_char_p
    return _lib


class HipError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise HipError(f"dsopp_hip error {rc}: {lib().dsopp_hip_last_error().decode()}")


def _p(a, dtype=np.float64):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def device_count() -> int:
    n = C.c_int()
    _chk(lib().dsopp_hip_device_count(C.byref(n)))
    return n.value


def default_pba_options(**kw) -> Options:
    o = Options()
    lib().dsopp_hip_default_pba_options(C.byref(o))
    for k, v in kw.items():
        if k == "affine_brightness_regularizer":
            o.affine_brightness_regularizer[0], o.affine_brightness_regularizer[1] = v
        else:
            setattr(o, k, v)
    return o


def default_align_options(**kw) -> Options:
    o = Options()
    lib().dsopp_hip_default_align_options(C.byref(o))
    for k, v in kw.items():
        if k == "affine_brightness_regularizer":
            o.affine_brightness_regularizer[0], o.affine_brightness_regularizer[1] = v
        else:
            setattr(o, k, v)
    return o


class Pyramid:
    """Device-resident image pyramid of one frame (dsopp_hip_pyramid)."""

    def __init__(self, width, height, levels=1, dtype=F64, device=0, stream=None):
        self._h = C.c_void_p()
        self.width, self.height, self.dtype, self.device = int(width), int(height), dtype, device
        _chk(lib().dsopp_hip_pyramid_create(int(device), C.c_void_p(stream or 0), int(width), int(height), int(levels), int(dtype),
                                            C.byref(self._h)))
        self.levels = min(int(levels), 5)

    def close(self):
        if self._h:
            lib().dsopp_hip_pyramid_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, image_u8, lut=None, vignetting=None):
        img = _u8(image_u8)
        assert img.shape == (self.height, self.width)
        _chk(lib().dsopp_hip_pyramid_build(self._h, _p(img, np.uint8), _p(None if lut is None else _f64(lut)), _p(_u8(vignetting), np.uint8)))

    def build_device(self, image_dev_ptr, lut=None, vignetting_dev_ptr=None, vignetting_max=0.0):
        _chk(lib().dsopp_hip_pyramid_build_device(self._h, C.c_void_p(image_dev_ptr), _p(None if lut is None else _f64(lut)),
                                                  C.c_void_p(vignetting_dev_ptr or 0), C.c_double(vignetting_max)))

    def set_level(self, level, pixelinfo):
        _chk(lib().dsopp_hip_pyramid_set_level(self._h, int(level), _p(_f64(pixelinfo))))

    def set_mask(self, level, mask):
        _chk(lib().dsopp_hip_pyramid_set_mask(self._h, int(level), _p(_u8(mask), np.uint8)))

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _chk(lib().dsopp_hip_pyramid_level_size(self._h, int(level), C.byref(w), C.byref(h)))
        return w.value, h.value

    def get_level(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w, 3))
        _chk(lib().dsopp_hip_pyramid_get_level(self._h, int(level), _p(out)))
        return out


class DepthMaps:
    """Device-resident reference depth maps of the newest keyframe (dsopp_hip_depth_maps)."""

    def __init__(self, handle, levels):
        self._h, self.levels = handle, int(levels)

    def close(self):
        if self._h:
            lib().dsopp_hip_depth_maps_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_depth_maps_level_size(self._h, int(level), C.byref(w), C.byref(h)))
        return w.value, h.value

    def mean_square_optical_flow(self, level, intrinsics, transforms):
        """calculateMeanSquareOpticalFlow for a list of T_target_reference (7-vectors, at most 4); returns an array"""
        T = _f64(np.concatenate([_f64(t) for t in transforms]))
        out = np.zeros(len(transforms))
        _chk(lib().dsopp_hip_depth_maps_mean_square_optical_flow(self._h, int(level), _p(_f64(intrinsics)), len(transforms), _p(T), _p(out)))
        return out

    def get_level(self, level):
        """(idepth_sum, weight), each H x W"""
        w, h = self.level_size(level)
        ids, wgt = np.zeros((h, w)), np.zeros((h, w))
        _chk(lib().dsopp_hip_depth_maps_get_level(self._h, int(level), _p(ids), _p(wgt)))
        return ids, wgt


class Comm:
    """One rank of the native RCCL communicator (dsopp_hip_comm): one process per GPU.  `exchange(id_bytes_or_None) -> bytes`
    moves the 128-byte id from rank 0 to every rank (e.g. torch.distributed.broadcast_object_list, MPI, a file)."""

    ID_BYTES = 128

    def __init__(self, rank: int, world_size: int, device: int, exchange):
        self._h = C.c_void_p()
        buf = (C.c_uint8 * self.ID_BYTES)()
        if rank == 0:
            _chk(lib().dsopp_hip_comm_unique_id(buf))
        raw = exchange(bytes(buf) if rank == 0 else None)
        buf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(raw)
        _chk(lib().dsopp_hip_comm_create(buf, int(rank), int(world_size), int(device), C.byref(self._h)))
        self.rank, self.world_size = int(rank), int(world_size)

    def size(self) -> int:
        """ranks of the communicator as RCCL itself reports them (ncclCommCount, read when the communicator was created)"""
        r, n = C.c_int(), C.c_int()
        _chk(lib().dsopp_hip_comm_rank(self._h, C.byref(r), C.byref(n)))
        return n.value

    def allreduce(self, device_ptr: int, count: int, stream: int = 0):
        _chk(lib().dsopp_hip_comm_allreduce(self._h, C.c_void_p(device_ptr), C.c_size_t(count), C.c_void_p(stream or 0)))

    def close(self):
        if self._h:
            lib().dsopp_hip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipWindow:
    """Sliding-window photometric BA on the GPU; same Python interface as oracle.pyoracle.OracleWindow."""

    def __init__(self, options: Options | None = None, device=0, stream=None):
        self.options = options or default_pba_options()
        self.device = device
        self._stream = stream
        self._h = C.c_void_p()
        _chk(lib().dsopp_hip_window_create(C.byref(self.options), int(device), C.c_void_p(stream or 0), C.byref(self._h)))
        self._pyramids = {}
        self._cb = None

    def close(self):
        if self._h:
            lib().dsopp_hip_window_destroy(self._h)
            self._h = C.c_void_p()
        for p, owned in self._pyramids.values():
            if owned:
                p.close()
        self._pyramids = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_frames(self) -> int:
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_num_frames(self._h, C.byref(n)))
        return n.value

    @property
    def K(self) -> int:
        return 8 * self.num_frames

    def push_frame(self, frame_id, timestamp, pixelinfo, mask, intrinsics, T_w_agent, exposure, affine, fixed, is_marginalized,
                   pyramid: Pyramid | None = None, level=0):
        """pixelinfo: H x W x 3 host array (uploaded into a 1-level pyramid owned by this wrapper) unless `pyramid` given."""
        owned = pyramid is None
        if pyramid is None:
            pix = _f64(pixelinfo)
            H, W = pix.shape[:2]
            pyramid = Pyramid(W, H, 1, self.options.dtype, self.device, self._stream)
            pyramid.set_level(0, pix)
            if mask is not None:
                pyramid.set_mask(0, mask)
            level = 0
        self._pyramids[int(frame_id)] = (pyramid, owned)
        _chk(lib().dsopp_hip_window_push_frame(self._h, int(frame_id), C.c_int64(int(timestamp)), pyramid._h, int(level),
                                               _p(_f64(intrinsics)), _p(_f64(T_w_agent)), C.c_double(exposure), _p(_f64(affine)),
                                               int(bool(fixed)), int(bool(is_marginalized))))
        # pushFrame folds the frames flagged for marginalisation into the prior and erases them: their pyramids are no longer
        # borrowed.  Wrapper-owned ones are freed here (a long sliding-window run would otherwise keep one per keyframe, ~10 MB each)
        alive = set(self.frame_ids())
        for fid in [k for k in self._pyramids if k not in alive]:
            p, was_owned = self._pyramids.pop(fid)
            if was_owned:
                p.close()

    def frame_ids(self):
        # sized from the window's own frame count (not from a constant mirrored here): a short buffer would make push_frame's
        # garbage collection close pyramids the window still borrows
        cap = max(1, self.num_frames)
        ids = np.zeros(cap, dtype=np.int32)
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_frame_ids(self._h, cap, ids.ctypes.data_as(C.c_void_p), C.byref(n)))
        return [int(x) for x in ids[:min(n.value, cap)]]

    def set_landmarks(self, frame_id, uv, idepth, patch, flags):
        n = len(idepth)
        _chk(lib().dsopp_hip_window_set_landmarks(self._h, int(frame_id), n, _p(_f64(uv)), _p(_f64(idepth)), _p(_f64(patch)),
                                                  _p(_u8(flags), np.uint8)))

    def set_connection(self, ref_id, tgt_id, statuses):
        st = _u8(statuses)
        _chk(lib().dsopp_hip_window_set_connection(self._h, int(ref_id), int(tgt_id), len(st), _p(st, np.uint8)))

    def mark_frame_marginalized(self, frame_id):
        _chk(lib().dsopp_hip_window_mark_frame_marginalized(self._h, int(frame_id)))

    def set_allreduce(self, fn, rank=0, world_size=1):
        """fn(device_ptr:int, count:int, stream:int) -> int.  Keeps the ctypes callback alive."""
        if fn is None:
            self._cb = None
            _chk(lib().dsopp_hip_window_set_allreduce(self._h, None, None, 0, 1))
            return
        self._cb = ALLREDUCE_FN(lambda user, ptr, count, stream: int(fn(ptr, count, stream) or 0))
        _chk(lib().dsopp_hip_window_set_allreduce(self._h, self._cb, None, int(rank), int(world_size)))

    def set_comm(self, comm: "Comm | None"):
        """native multi-GPU exchange: the library enqueues ncclAllReduce itself (no Python in the solve loop)"""
        self._comm = comm
        self._cb = None
        _chk(lib().dsopp_hip_window_set_comm(self._h, comm._h if comm is not None else None))

    def optimize(self):
        """LM loop only (no relinearisation / covariance / point statuses)."""
        e, it, nv = C.c_double(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_optimize(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def get_frame_update(self, frame_id, target_ids):
        """everything updateFrame reads back for one keyframe in one transfer: dict(idepth, inv_hdd, relative_baseline,
        n_inliers, flags, status={target_id: array})"""
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_num_landmarks(self._h, int(frame_id), C.byref(n)))
        n = n.value
        tids = np.ascontiguousarray(target_ids, dtype=np.int32)
        out = dict(idepth=np.zeros(n), inv_hdd=np.zeros(n), relative_baseline=np.zeros(n), n_inliers=np.zeros(n, dtype=np.int32),
                   flags=np.zeros(n, dtype=np.uint8))
        st = np.zeros((len(tids), n), dtype=np.uint8)
        _chk(lib().dsopp_hip_window_get_frame_update(self._h, int(frame_id), _p(out["idepth"]), _p(out["inv_hdd"]), _p(out["relative_baseline"]),
                                                     out["n_inliers"].ctypes.data_as(C.c_void_p), _p(out["flags"], np.uint8), len(tids),
                                                     tids.ctypes.data_as(C.c_void_p), _p(st, np.uint8)))
        out["status"] = {int(t): st[k] for k, t in enumerate(tids)}
        return out

    def create_reference_depth_maps(self, levels: int) -> DepthMaps:
        """createReferenceDepthMaps of the window's newest keyframe, on the device"""
        h = C.c_void_p()
        _chk(lib().dsopp_hip_window_create_reference_depth_maps(self._h, int(levels), C.byref(h)))
        return DepthMaps(h, levels)

    def activate_landmarks(self, frame_ids, immature_sets, newest_pyramid: "Pyramid", T_world_newest, exposure_newest=1.0,
                           affine_newest=(0, 0), number_of_desired_points=2000, min_distance_to_neighbor=0.0, refine=True,
                           sigma_huber_loss=20.0):
        """LandmarksActivator::activate for the listed window keyframes (oldest first) and their device-resident immature
        sets against a new keyframe.  Returns (statuses per keyframe, idepths per keyframe, result dict)."""
        n = len(frame_ids)
        ids = np.ascontiguousarray(frame_ids, dtype=np.int32)
        sets = (C.c_void_p * n)(*[s._h if s is not None else None for s in immature_sets])
        st = [np.zeros(s.n if s is not None else 0, dtype=np.uint8) for s in immature_sets]
        idp = [np.zeros(s.n if s is not None else 0) for s in immature_sets]
        st_p = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p) for a in st])
        id_p = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p) for a in idp])
        dist = C.c_double(min_distance_to_neighbor)
        res = ActivationResult()
        _chk(lib().dsopp_hip_window_activate_landmarks(self._h, n, _p(ids, np.int32), sets, newest_pyramid._h, _p(_f64(T_world_newest)),
                                                       C.c_double(exposure_newest), _p(_f64(affine_newest)), int(number_of_desired_points),
                                                       C.byref(dist), int(bool(refine)), C.c_double(sigma_huber_loss), st_p, id_p, C.byref(res)))
        return st, idp, {k: getattr(res, k) for k, _ in ActivationResult._fields_}

    def refill_reference_depth_maps(self, maps: DepthMaps):
        """createReferenceDepthMaps into an existing DepthMaps object (no allocation)"""
        _chk(lib().dsopp_hip_window_refill_reference_depth_maps(self._h, maps._h))

    def optimize_async(self):
        """enqueue the LM loop on the window's stream (no host synchronisation)"""
        _chk(lib().dsopp_hip_window_optimize_async(self._h))

    def optimize_wait(self):
        e, it, nv = C.c_double(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_optimize_wait(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def optimize_repeated(self, iterations_target: int):
        """{restore(); optimize()} from the snapshot until exactly `iterations_target` GN iterations ran; returns
        (iterations_done, last_energy).  Same work as the Python loop, without its per-call overhead between solves."""
        done, e = C.c_int32(), C.c_double()
        _chk(lib().dsopp_hip_window_optimize_repeated(self._h, int(iterations_target), C.byref(done), C.byref(e)))
        return done.value, e.value

    def set_max_iterations(self, n: int):
        _chk(lib().dsopp_hip_window_set_max_iterations(self._h, int(n)))
        self.options.max_iterations = int(n)

    KERNEL_CLASSES = {"pair_setup": 0, "fej": 1, "sweep_linearize": 2, "sweep_energy": 3, "schur": 4, "assemble": 5, "assemble_solve": 6,
                      "backsub": 7, "energy_reduce": 8, "accept_decide": 9,
                      "sweep_linearize_loop": 10}

    def time_kernel(self, name: str, repeats: int = 50) -> float:
        """average microseconds of `repeats` back-to-back launches of one kernel class (one HIP event pair)"""
        us = C.c_double()
        _chk(lib().dsopp_hip_window_time_kernel(self._h, self.KERNEL_CLASSES[name], int(repeats), C.byref(us)))
        return us.value

    def set_deterministic(self, enable: bool):
        """two-stage (atomic-free, bit-reproducible) build of the reduced normal equations at every window size"""
        _chk(lib().dsopp_hip_window_set_deterministic(self._h, int(bool(enable))))

    def set_lm_mode(self, mode: int):
        """0 fused device loop (default), 1 host-driven stages, 2 unfused device loop"""
        _chk(lib().dsopp_hip_window_set_lm_mode(self._h, int(mode)))

    def snapshot(self):
        _chk(lib().dsopp_hip_window_snapshot(self._h))

    def restore(self):
        _chk(lib().dsopp_hip_window_restore(self._h))

    def set_profiling(self, enable: bool):
        _chk(lib().dsopp_hip_window_set_profiling(self._h, int(bool(enable))))

    def get_profile(self):
        """{kernel class name: (total device ms, launches)} since profiling was enabled."""
        out = {}
        for k in range(NUM_KERNEL_CLASSES):
            ms, n = C.c_double(), C.c_int64()
            _chk(lib().dsopp_hip_window_get_profile(self._h, k, C.byref(ms), C.byref(n)))
            out[lib().dsopp_hip_kernel_class_name(k).decode()] = (ms.value, n.value)
        return out

    # stage level
    def begin(self):
        _chk(lib().dsopp_hip_window_begin(self._h))

    def calculate_energy(self):
        e, n = C.c_double(), C.c_int32()
        _chk(lib().dsopp_hip_window_calculate_energy(self._h, C.byref(e), C.byref(n)))
        return e.value, n.value

    def linearize(self):
        _chk(lib().dsopp_hip_window_linearize(self._h))

    def get_system(self):
        K = self.K
        Hpp, bpp, Hsc, bsc = np.zeros((K, K)), np.zeros(K), np.zeros((K, K)), np.zeros(K)
        _chk(lib().dsopp_hip_window_get_system(self._h, _p(Hpp), _p(bpp), _p(Hsc), _p(bsc)))
        return Hpp, bpp, Hsc, bsc

    def calculate_step(self, lam):
        step = np.zeros(self.K)
        _chk(lib().dsopp_hip_window_calculate_step(self._h, C.c_double(lam), _p(step)))
        return step

    def accept_step(self):
        a, b = C.c_double(), C.c_double()
        _chk(lib().dsopp_hip_window_accept_step(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reject_step(self):
        _chk(lib().dsopp_hip_window_reject_step(self._h))

    def update_point_statuses(self):
        _chk(lib().dsopp_hip_window_update_point_statuses(self._h))

    def solve(self):
        e, it, nv = C.c_double(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_solve(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def last_solve_ms(self) -> float:
        ms = C.c_float()
        _chk(lib().dsopp_hip_window_last_solve_ms(self._h, C.byref(ms)))
        return ms.value

    # getters
    def get_frame_state(self, frame_id):
        T0, ab0, eps, step = np.zeros(7), np.zeros(2), np.zeros(8), np.zeros(8)
        _chk(lib().dsopp_hip_window_get_frame_state(self._h, int(frame_id), _p(T0), _p(ab0), _p(eps), _p(step)))
        return T0, ab0, eps, step

    def get_pose(self, frame_id):
        T, ab = np.zeros(7), np.zeros(2)
        _chk(lib().dsopp_hip_window_get_pose(self._h, int(frame_id), _p(T), _p(ab)))
        return T, ab

    def num_landmarks(self, frame_id):
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_num_landmarks(self._h, int(frame_id), C.byref(n)))
        return n.value

    def get_landmarks(self, frame_id, with_hpib=True):
        n = self.num_landmarks(frame_id)
        K = self.K
        out = dict(idepth=np.zeros(n), idepth_step=np.zeros(n), inv_hdd=np.zeros(n), b_d=np.zeros(n), relative_baseline=np.zeros(n),
                   n_inliers=np.zeros(n, dtype=np.int32), flags=np.zeros(n, dtype=np.uint8))
        hp = np.zeros((n, K)) if with_hpib else None
        _chk(lib().dsopp_hip_window_get_landmarks(self._h, int(frame_id), _p(out["idepth"]), _p(out["idepth_step"]), _p(out["inv_hdd"]),
                                                  _p(out["b_d"]), _p(out["relative_baseline"]), _p(out["n_inliers"], np.int32),
                                                  _p(out["flags"], np.uint8), _p(hp)))
        if with_hpib:
            out["hpib"] = hp
        return out

    def get_residuals(self, ref_id, tgt_id, full=False):
        n = self.num_landmarks(ref_id)
        out = dict(status=np.zeros(n, dtype=np.uint8), candidate=np.zeros(n, dtype=np.uint8), energy=np.zeros(n))
        _chk(lib().dsopp_hip_window_get_residuals(self._h, int(ref_id), int(tgt_id), n, _p(out["status"], np.uint8),
                                                  _p(out["candidate"], np.uint8), _p(out["energy"])))
        return out

    def get_marginalized(self):
        K = self.K
        H, b, e, sz = np.zeros((K, K)), np.zeros(K), C.c_double(), C.c_int32()
        _chk(lib().dsopp_hip_window_get_marginalized(self._h, _p(H), _p(b), C.byref(e), C.byref(sz)))
        return H, b, e.value

    def get_covariance(self, ref_id, tgt_id):
        cov = np.zeros((6, 6))
        _chk(lib().dsopp_hip_window_get_covariance(self._h, int(ref_id), int(tgt_id), _p(cov)))
        return cov



TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_LOCAL, TRANSPORT_P2P = 0, 1, 2, 3


class PyramidGroup:
    """One keyframe's image pyramid on every distinct device of a window group (dsopp_hip_pyramid_group)."""

    def __init__(self, group: "HipWindowGroup", width, height, levels=1):
        self._h = C.c_void_p()
        self.width, self.height, self.levels = int(width), int(height), min(int(levels), 5)
        _chk(lib().dsopp_hip_pyramid_group_create(group._h, int(width), int(height), int(levels), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().dsopp_hip_pyramid_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, image_u8, lut=None, vignetting=None):
        img = _u8(image_u8)
        assert img.shape == (self.height, self.width)
        _chk(lib().dsopp_hip_pyramid_group_build(self._h, _p(img, np.uint8), _p(None if lut is None else _f64(lut)), _p(_u8(vignetting), np.uint8)))

    def set_level(self, level, pixelinfo):
        _chk(lib().dsopp_hip_pyramid_group_set_level(self._h, int(level), _p(_f64(pixelinfo))))

    def set_mask(self, level, mask):
        _chk(lib().dsopp_hip_pyramid_group_set_mask(self._h, int(level), _p(_u8(mask), np.uint8)))


class HipWindowGroup:
    """n landmark shards of ONE sliding window on n devices behind one object (dsopp_hip_window_group): the single-process
    multi-GPU form of the drop-in.  Same Python interface as HipWindow / oracle.pyoracle.OracleWindow; landmark arrays are the
    whole keyframe's, the group deals them over the shards and interleaves the read-backs."""

    def __init__(self, options: Options | None = None, devices=(0,), transport=TRANSPORT_AUTO):
        self.options = options or default_pba_options()
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        ids = np.ascontiguousarray(self.devices, dtype=np.int32)
        _chk(lib().dsopp_hip_window_group_create(C.byref(self.options), ids.ctypes.data_as(C.c_void_p), len(ids), int(transport), C.byref(self._h)))
        self._pyramids = {}

    def close(self):
        if self._h:
            lib().dsopp_hip_window_group_destroy(self._h)
            self._h = C.c_void_p()
        for p, owned in self._pyramids.values():
            if owned:
                p.close()
        self._pyramids = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        n, t = C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_size(self._h, C.byref(n), C.byref(t)))
        return n.value

    @property
    def transport(self):
        n, t = C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_size(self._h, C.byref(n), C.byref(t)))
        return t.value

    @property
    def num_frames(self) -> int:
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_group_num_frames(self._h, C.byref(n)))
        return n.value

    @property
    def K(self) -> int:
        return 8 * self.num_frames

    def frame_ids(self):
        cap = max(1, self.num_frames)
        ids = np.zeros(cap, dtype=np.int32)
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_group_frame_ids(self._h, cap, ids.ctypes.data_as(C.c_void_p), C.byref(n)))
        return [int(x) for x in ids[:min(n.value, cap)]]

    def shard_num_landmarks(self, shard, frame_id):
        """landmarks of `frame_id` held by one shard (introspection through the shard's own window)"""
        w, dev, n = C.c_void_p(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_shard(self._h, int(shard), C.byref(w), C.byref(dev)))
        _chk(lib().dsopp_hip_window_num_landmarks(w, int(frame_id), C.byref(n)))
        return n.value

    def push_frame(self, frame_id, timestamp, pixelinfo, mask, intrinsics, T_w_agent, exposure, affine, fixed, is_marginalized,
                   pyramid: PyramidGroup | None = None, level=0):
        owned = pyramid is None
        if pyramid is None:
            pix = _f64(pixelinfo)
            H, W = pix.shape[:2]
            pyramid = PyramidGroup(self, W, H, 1)
            pyramid.set_level(0, pix)
            if mask is not None:
                pyramid.set_mask(0, mask)
            level = 0
        self._pyramids[int(frame_id)] = (pyramid, owned)
        _chk(lib().dsopp_hip_window_group_push_frame(self._h, int(frame_id), C.c_int64(int(timestamp)), pyramid._h, int(level),
                                                     _p(_f64(intrinsics)), _p(_f64(T_w_agent)), C.c_double(exposure), _p(_f64(affine)),
                                                     int(bool(fixed)), int(bool(is_marginalized))))
        alive = set(self.frame_ids())
        for fid in [k for k in self._pyramids if k not in alive]:
            p, was_owned = self._pyramids.pop(fid)
            if was_owned:
                p.close()

    def set_landmarks(self, frame_id, uv, idepth, patch, flags):
        n = len(idepth)
        _chk(lib().dsopp_hip_window_group_set_landmarks(self._h, int(frame_id), n, _p(_f64(uv)), _p(_f64(idepth)), _p(_f64(patch)),
                                                        _p(_u8(flags), np.uint8)))

    def set_connection(self, ref_id, tgt_id, statuses):
        st = _u8(statuses)
        _chk(lib().dsopp_hip_window_group_set_connection(self._h, int(ref_id), int(tgt_id), len(st), _p(st, np.uint8)))

    def mark_frame_marginalized(self, frame_id):
        _chk(lib().dsopp_hip_window_group_mark_frame_marginalized(self._h, int(frame_id)))

    def solve(self):
        e, it, nv = C.c_double(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_solve(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def optimize(self):
        e, it, nv = C.c_double(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_optimize(self._h, C.byref(e), C.byref(it), C.byref(nv)))
        return e.value, it.value, nv.value

    def optimize_repeated(self, iterations_target: int):
        done, e = C.c_int32(), C.c_double()
        _chk(lib().dsopp_hip_window_group_optimize_repeated(self._h, int(iterations_target), C.byref(done), C.byref(e)))
        return done.value, e.value

    def last_solve_ms(self) -> float:
        ms = C.c_float()
        _chk(lib().dsopp_hip_window_group_last_solve_ms(self._h, C.byref(ms)))
        return ms.value

    def set_max_iterations(self, n: int):
        _chk(lib().dsopp_hip_window_group_set_max_iterations(self._h, int(n)))
        self.options.max_iterations = int(n)

    def set_deterministic(self, enable: bool):
        _chk(lib().dsopp_hip_window_group_set_deterministic(self._h, int(bool(enable))))

    def set_lm_mode(self, mode: int):
        _chk(lib().dsopp_hip_window_group_set_lm_mode(self._h, int(mode)))

    def snapshot(self):
        _chk(lib().dsopp_hip_window_group_snapshot(self._h))

    def restore(self):
        _chk(lib().dsopp_hip_window_group_restore(self._h))

    # stage level
    def begin(self):
        _chk(lib().dsopp_hip_window_group_begin(self._h))

    def calculate_energy(self):
        e, n = C.c_double(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_calculate_energy(self._h, C.byref(e), C.byref(n)))
        return e.value, n.value

    def linearize(self):
        _chk(lib().dsopp_hip_window_group_linearize(self._h))

    def get_system(self):
        K = self.K
        Hpp, bpp, Hsc, bsc = np.zeros((K, K)), np.zeros(K), np.zeros((K, K)), np.zeros(K)
        _chk(lib().dsopp_hip_window_group_get_system(self._h, _p(Hpp), _p(bpp), _p(Hsc), _p(bsc)))
        return Hpp, bpp, Hsc, bsc

    def calculate_step(self, lam):
        step = np.zeros(self.K)
        _chk(lib().dsopp_hip_window_group_calculate_step(self._h, C.c_double(lam), _p(step)))
        return step

    def accept_step(self):
        a, b = C.c_double(), C.c_double()
        _chk(lib().dsopp_hip_window_group_accept_step(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reject_step(self):
        _chk(lib().dsopp_hip_window_group_reject_step(self._h))

    def update_point_statuses(self):
        _chk(lib().dsopp_hip_window_group_update_point_statuses(self._h))

    # getters
    def get_frame_state(self, frame_id):
        T0, ab0, eps, step = np.zeros(7), np.zeros(2), np.zeros(8), np.zeros(8)
        _chk(lib().dsopp_hip_window_group_get_frame_state(self._h, int(frame_id), _p(T0), _p(ab0), _p(eps), _p(step)))
        return T0, ab0, eps, step

    def get_pose(self, frame_id):
        T, ab = np.zeros(7), np.zeros(2)
        _chk(lib().dsopp_hip_window_group_get_pose(self._h, int(frame_id), _p(T), _p(ab)))
        return T, ab

    def num_landmarks(self, frame_id):
        n = C.c_int32()
        _chk(lib().dsopp_hip_window_group_num_landmarks(self._h, int(frame_id), C.byref(n)))
        return n.value

    def get_landmarks(self, frame_id, with_hpib=True):
        n = self.num_landmarks(frame_id)
        K = self.K
        out = dict(idepth=np.zeros(n), idepth_step=np.zeros(n), inv_hdd=np.zeros(n), b_d=np.zeros(n), relative_baseline=np.zeros(n),
                   n_inliers=np.zeros(n, dtype=np.int32), flags=np.zeros(n, dtype=np.uint8))
        hp = np.zeros((n, K)) if with_hpib else None
        _chk(lib().dsopp_hip_window_group_get_landmarks(self._h, int(frame_id), _p(out["idepth"]), _p(out["idepth_step"]), _p(out["inv_hdd"]),
                                                        _p(out["b_d"]), _p(out["relative_baseline"]), _p(out["n_inliers"], np.int32),
                                                        _p(out["flags"], np.uint8), _p(hp)))
        if with_hpib:
            out["hpib"] = hp
        return out

    def get_frame_update(self, frame_id, target_ids):
        n = self.num_landmarks(frame_id)
        tids = np.ascontiguousarray(target_ids, dtype=np.int32)
        out = dict(idepth=np.zeros(n), inv_hdd=np.zeros(n), relative_baseline=np.zeros(n), n_inliers=np.zeros(n, dtype=np.int32),
                   flags=np.zeros(n, dtype=np.uint8))
        st = np.zeros((len(tids), n), dtype=np.uint8)
        _chk(lib().dsopp_hip_window_group_get_frame_update(self._h, int(frame_id), _p(out["idepth"]), _p(out["inv_hdd"]), _p(out["relative_baseline"]),
                                                           out["n_inliers"].ctypes.data_as(C.c_void_p), _p(out["flags"], np.uint8), len(tids),
                                                           tids.ctypes.data_as(C.c_void_p), _p(st, np.uint8)))
        out["status"] = {int(t): st[k] for k, t in enumerate(tids)}
        return out

    def get_residuals(self, ref_id, tgt_id, full=False):
        n = self.num_landmarks(ref_id)
        out = dict(status=np.zeros(n, dtype=np.uint8), candidate=np.zeros(n, dtype=np.uint8), energy=np.zeros(n))
        _chk(lib().dsopp_hip_window_group_get_residuals(self._h, int(ref_id), int(tgt_id), n, _p(out["status"], np.uint8),
                                                        _p(out["candidate"], np.uint8), _p(out["energy"])))
        return out

    def get_marginalized(self):
        K = self.K
        H, b, e, sz = np.zeros((K, K)), np.zeros(K), C.c_double(), C.c_int32()
        _chk(lib().dsopp_hip_window_group_get_marginalized(self._h, _p(H), _p(b), C.byref(e), C.byref(sz)))
        return H, b, e.value

    def get_covariance(self, ref_id, tgt_id):
        cov = np.zeros((6, 6))
        _chk(lib().dsopp_hip_window_group_get_covariance(self._h, int(ref_id), int(tgt_id), _p(cov)))
        return cov

    def create_reference_depth_maps(self, levels: int) -> DepthMaps:
        h = C.c_void_p()
        _chk(lib().dsopp_hip_window_group_create_reference_depth_maps(self._h, int(levels), C.byref(h)))
        return DepthMaps(h, levels)

    def refill_reference_depth_maps(self, maps: DepthMaps):
        _chk(lib().dsopp_hip_window_group_refill_reference_depth_maps(self._h, maps._h))


def estimate_depths(lms, target_pyramid: Pyramid, level, intrinsics, T_target_reference, reference_exposure=1.0, reference_affine=(0, 0),
                    target_exposure=1.0, target_affine=(0, 0), sigma_huber_loss=20.0):
    """DepthEstimation::estimate on the device; `lms` is the dict of arrays of oracle.pyoracle.new_immature_landmarks, updated in place"""
    n = len(lms["status"])
    for k in ("idepth_min", "idepth_max", "uniqueness", "search_pixel_interval"):
        lms[k] = _f64(lms[k])
    for k in ("status", "traced"):
        lms[k] = _u8(lms[k])
    _chk(lib().dsopp_hip_estimate_depths(target_pyramid._h, int(level), _p(_f64(intrinsics)), _p(_f64(T_target_reference)), C.c_double(reference_exposure),
                                         _p(_f64(reference_affine)), C.c_double(target_exposure), _p(_f64(target_affine)), C.c_double(sigma_huber_loss), n,
                                         _p(_f64(lms["projection"])), _p(_f64(lms["direction"])), _p(_f64(lms["patch"])), _p(_f64(lms["gradient"])),
                                         _p(lms["idepth_min"]), _p(lms["idepth_max"]), _p(lms["uniqueness"]), _p(lms["search_pixel_interval"]),
                                         _p(lms["status"], np.uint8), _p(lms["traced"], np.uint8)))
    return lms


def initialization_poses(T_world_previous, T_world_last, T_world_keyframe):
    """initializationPoses of the tracker (monocular_tracker.cpp:136-176); T_world_previous None = fewer than two frames"""
    out, n = np.zeros((128, 7)), C.c_int32()
    if T_world_previous is None:
        _chk(lib().dsopp_hip_initialization_poses(None, None, None, 128, _p(out), C.byref(n)))
    else:
        _chk(lib().dsopp_hip_initialization_poses(_p(_f64(T_world_previous)), _p(_f64(T_world_last)), _p(_f64(T_world_keyframe)), 128, _p(out), C.byref(n)))
    return out[:n.value].copy()


def estimate_depths_batched(sets, target_pyramid, level, intrinsics, T_target_reference, reference_exposure, reference_affine,
                            target_exposure=1.0, target_affine=(0, 0), sigma_huber_loss=20.0):
    """DepthEstimation::estimate for several keyframes' device-resident sets against one new frame in one launch"""
    n = len(sets)
    arr = (C.c_void_p * n)(*[s._h for s in sets])
    _chk(lib().dsopp_hip_immature_sets_estimate(n, arr, target_pyramid._h, int(level), _p(_f64(intrinsics)), _p(_f64(np.asarray(T_target_reference).reshape(-1))),
                                                _p(_f64(reference_exposure)), _p(_f64(np.asarray(reference_affine).reshape(-1))), C.c_double(target_exposure),
                                                _p(_f64(target_affine)), C.c_double(sigma_huber_loss)))


class ActivationResult(C.Structure):
    _fields_ = [("number_of_active_points", C.c_int32), ("n_activated", C.c_int32), ("n_skipped", C.c_int32), ("n_deleted", C.c_int32),
                ("selection_rounds", C.c_int32), ("min_distance_to_neighbor", C.c_double)]


class ImmatureSet:
    """Device-resident immature landmarks of one keyframe (dsopp_hip_immature_set)."""

    def __init__(self, lms, device=0, stream=None):
        self._h = C.c_void_p()
        self.n = len(lms["status"])
        _chk(lib().dsopp_hip_immature_set_create(int(device), C.c_void_p(stream or 0), self.n, _p(_f64(lms["projection"])), _p(_f64(lms["direction"])),
                                                 _p(_f64(lms["patch"])), _p(_f64(lms["gradient"])), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().dsopp_hip_immature_set_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def estimate(self, target_pyramid: Pyramid, level, intrinsics, T_target_reference, reference_exposure=1.0, reference_affine=(0, 0),
                 target_exposure=1.0, target_affine=(0, 0), sigma_huber_loss=20.0):
        _chk(lib().dsopp_hip_immature_set_estimate(self._h, target_pyramid._h, int(level), _p(_f64(intrinsics)), _p(_f64(T_target_reference)),
                                                   C.c_double(reference_exposure), _p(_f64(reference_affine)), C.c_double(target_exposure),
                                                   _p(_f64(target_affine)), C.c_double(sigma_huber_loss)))

    def sync(self):
        """wait for the set's stream (dsopp_hip_immature_set_download_state with no outputs)"""
        _chk(lib().dsopp_hip_immature_set_download_state(self._h, None, None, None, None, None, None))

    def upload(self, lms):
        """replace the estimator state (idepth interval, uniqueness, search interval, status, traced)"""
        _chk(lib().dsopp_hip_immature_set_upload_state(self._h, _p(_f64(lms["idepth_min"])), _p(_f64(lms["idepth_max"])), _p(_f64(lms["uniqueness"])),
                                                       _p(_f64(lms["search_pixel_interval"])), _p(np.ascontiguousarray(lms["status"], dtype=np.uint8), np.uint8),
                                                       _p(np.ascontiguousarray(lms["traced"], dtype=np.uint8), np.uint8)))

    def download(self):
        n = self.n
        out = dict(idepth_min=np.zeros(n), idepth_max=np.zeros(n), uniqueness=np.zeros(n), search_pixel_interval=np.zeros(n),
                   status=np.zeros(n, dtype=np.uint8), traced=np.zeros(n, dtype=np.uint8))
        _chk(lib().dsopp_hip_immature_set_download_state(self._h, _p(out["idepth_min"]), _p(out["idepth_max"]), _p(out["uniqueness"]),
                                                         _p(out["search_pixel_interval"]), _p(out["status"], np.uint8), _p(out["traced"], np.uint8)))
        return out


class HipAligner:
    """Two-frame direct image alignment of one pyramid level (dsopp_hip_aligner)."""

    def __init__(self, options: Options | None = None, device=0, stream=None):
        self.options = options or default_align_options()
        self._h = C.c_void_p()
        _chk(lib().dsopp_hip_aligner_create(C.byref(self.options), int(device), C.c_void_p(stream or 0), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().dsopp_hip_aligner_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _chk(lib().dsopp_hip_aligner_reset(self._h))

    def push_reference_depth_map(self, timestamp, T_w_agent, pyramid: Pyramid, level, intrinsics, idepth_sum, weight, exposure, affine):
        _chk(lib().dsopp_hip_aligner_push_reference_depth_map(self._h, C.c_int64(int(timestamp)), _p(_f64(T_w_agent)), pyramid._h, int(level),
                                                              _p(_f64(intrinsics)), _p(_f64(idepth_sum)), _p(_f64(weight)), C.c_double(exposure),
                                                              _p(_f64(affine))))

    def push_reference_depth_maps(self, timestamp, T_w_agent, pyramid: Pyramid, level, intrinsics, maps: DepthMaps, exposure, affine):
        _chk(lib().dsopp_hip_aligner_push_reference_depth_maps(self._h, C.c_int64(int(timestamp)), _p(_f64(T_w_agent)), pyramid._h, int(level),
                                                               _p(_f64(intrinsics)), maps._h, C.c_double(exposure), _p(_f64(affine))))

    def estimate_pose(self, ref_time, T_w_ref, ref_pyramid: Pyramid, ref_maps: DepthMaps, ref_exposure, ref_affine, tgt_time,
                      tgt_pyramid: Pyramid, tgt_exposure, intrinsics, T_w_inits, affine_init, rmse_last):
        """estimatePose of the tracker (monocular_tracker.cpp:179-245): coarse-to-fine over all levels, one C call per frame.
        T_w_inits: (n, 7) initialisations; rmse_last: per-level array, updated in place."""
        inits = _f64(np.atleast_2d(T_w_inits))
        rl = _f64(rmse_last)
        T, ab = np.zeros(7), np.zeros(2)
        ok, tries, its = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(lib().dsopp_hip_aligner_estimate_pose(self._h, C.c_int64(int(ref_time)), _p(_f64(T_w_ref)), ref_pyramid._h, ref_maps._h,
                                                   C.c_double(ref_exposure), _p(_f64(ref_affine)), C.c_int64(int(tgt_time)), tgt_pyramid._h,
                                                   C.c_double(tgt_exposure), _p(_f64(intrinsics)), len(inits), _p(inits), _p(_f64(affine_init)),
                                                   _p(rl), _p(T), _p(ab), C.byref(ok), C.byref(tries), C.byref(its)))
        rmse_last[:] = rl
        return dict(T_w_target=T, affine_brightness=ab, success=bool(ok.value), tries=tries.value, lm_iterations=its.value)

    def push_reference_points(self, timestamp, T_w_agent, pyramid: Pyramid, level, intrinsics, u, v, idepth, exposure, affine):
        _chk(lib().dsopp_hip_aligner_push_reference_points(self._h, C.c_int64(int(timestamp)), _p(_f64(T_w_agent)), pyramid._h, int(level),
                                                           _p(_f64(intrinsics)), len(u), _p(_f64(u)), _p(_f64(v)), _p(_f64(idepth)),
                                                           C.c_double(exposure), _p(_f64(affine))))

    def push_target(self, timestamp, T_w_agent_init, pyramid: Pyramid, level, intrinsics, exposure, affine):
        _chk(lib().dsopp_hip_aligner_push_target(self._h, C.c_int64(int(timestamp)), _p(_f64(T_w_agent_init)), pyramid._h, int(level),
                                                 _p(_f64(intrinsics)), C.c_double(exposure), _p(_f64(affine))))

    def set_rotation_prior(self, R_target_reference):
        """setRotationPrior (3x3, None clears); reset() clears it too"""
        R = None if R_target_reference is None else _f64(np.asarray(R_target_reference).reshape(9))
        _chk(lib().dsopp_hip_aligner_set_rotation_prior(self._h, _p(R)))

    def set_lm_path(self, path: int):
        _chk(lib().dsopp_hip_aligner_set_lm_path(self._h, int(path)))

    def set_hypothesis_width(self, width: int):
        """initialisations estimate_pose evaluates per launch: 0 automatic, 1 sequential, 2 .. 8 concurrent (one XCD each)"""
        _chk(lib().dsopp_hip_aligner_set_hypothesis_width(self._h, int(width)))

    def push_known_pose(self, timestamp, T_w_agent):
        _chk(lib().dsopp_hip_aligner_push_known_pose(self._h, C.c_int64(int(timestamp)), _p(_f64(T_w_agent))))

    def num_points(self) -> int:
        n = C.c_int32()
        _chk(lib().dsopp_hip_aligner_num_points(self._h, C.byref(n)))
        return n.value

    def solve(self):
        out = AlignResult()
        _chk(lib().dsopp_hip_aligner_solve(self._h, C.byref(out)))
        return dict(rmse=out.rmse, energy=out.energy, n_valid=out.n_valid, iterations=out.iterations,
                    T_w_target=np.array(out.T_world_target), affine_brightness=np.array(out.affine_brightness),
                    covariance=np.array(out.covariance).reshape(6, 6), H=np.array(out.H).reshape(8, 8))
