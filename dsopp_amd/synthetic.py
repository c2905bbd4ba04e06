"""Deterministic synthetic sliding windows and tracker frames.

The reference's solver tests and benchmarks all read the un-shipped ``track30seconds`` sequence through
``test_tools::SolverTestData`` (``test/tools/src/solver_test_data.cpp:31-143``).  This module produces the same kind of
input — a window of keyframes with images, ground-truth poses/depths, noisy initial guesses and a full clique of
connections — from an analytic scene, so tests, the CPU baseline and the GPU bench all see identical data.

Scene: a smooth depth surface ``z0(u0, v0)`` over frame-0 image coordinates (sum of low-frequency cosines, z in about
[2, 10]) textured by a band-limited sum of cosines (intensities in [0, 255]).  Both are analytic, so rendering a frame
needs no image resampling: for every pixel the ray/surface intersection is found by a fixed-point iteration.
"""
from __future__ import annotations

import dataclasses
import os

import numpy as np

# Pattern offsets (x_i, y_i): src/common/pattern/include/common/pattern/pattern.hpp:21-32
PATTERN = np.array([[0, 2], [-1, 1], [1, 1], [-2, 0], [0, 0], [2, 0], [-1, -1], [0, -2]], dtype=np.float64)


def se3_exp(xi: np.ndarray) -> np.ndarray:
    """Sophus convention: tangent (upsilon, omega); returns 4x4 homogeneous matrix."""
    xi = np.asarray(xi, dtype=np.float64)
    u, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    Om = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        R = np.eye(3) + Om + 0.5 * Om @ Om
        V = np.eye(3) + 0.5 * Om + Om @ Om / 6.0
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th**2 * Om @ Om
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * Om + (th - np.sin(th)) / th**3 * Om @ Om
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ u
    return T


def mat_to_params(T: np.ndarray) -> np.ndarray:
    """4x4 -> (qx, qy, qz, qw, tx, ty, tz), the Sophus::SE3 storage order."""
    R = T[:3, :3]
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    q = np.array([x, y, z, w])
    q /= np.linalg.norm(q)
    return np.concatenate([q, T[:3, 3]])


def params_to_mat(p: np.ndarray) -> np.ndarray:
    x, y, z, w = p[:4]
    R = np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p[4:7]
    return T


@dataclasses.dataclass
class Scene:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    depth_amp: np.ndarray
    depth_fu: np.ndarray
    depth_fv: np.ndarray
    depth_ph: np.ndarray
    tex_amp: np.ndarray
    tex_ku: np.ndarray
    tex_kv: np.ndarray
    tex_ph: np.ndarray

    @staticmethod
    def make(width: int, height: int, seed: int = 0, tex_terms: int = 40) -> "Scene":
        rng = np.random.default_rng(seed)
        fx = fy = 0.7 * width
        depth_amp = rng.uniform(0.2, 0.6, 6)
        depth_fu = rng.uniform(-1.5, 1.5, 6)
        depth_fv = rng.uniform(-1.5, 1.5, 6)
        depth_ph = rng.uniform(0, 2 * np.pi, 6)
        # spatial frequencies in cycles / pixel (of a 640-wide image; scaled with resolution), log-uniform
        f = np.exp(rng.uniform(np.log(1 / 160.0), np.log(1 / 7.0), tex_terms)) * (640.0 / width)
        ang = rng.uniform(0, 2 * np.pi, tex_terms)
        tex_amp = 1.0 / np.sqrt(f / f.min())
        tex_amp *= 45.0 / np.sqrt(0.5 * np.sum(tex_amp**2))
        return Scene(width, height, fx, fy, width / 2.0, height / 2.0, depth_amp, depth_fu, depth_fv, depth_ph, tex_amp,
                     f * np.cos(ang), f * np.sin(ang), rng.uniform(0, 2 * np.pi, tex_terms))

    @property
    def intrinsics(self) -> np.ndarray:
        return np.array([self.fx, self.fy, self.cx, self.cy])

    def depth0(self, u0: np.ndarray, v0: np.ndarray) -> np.ndarray:
        z = np.full(u0.shape, 6.0)
        for a, fu, fv, ph in zip(self.depth_amp, self.depth_fu, self.depth_fv, self.depth_ph):
            z += a * np.cos(2 * np.pi * (fu * u0 / self.width + fv * v0 / self.height) + ph)
        return z

    def texture(self, u0: np.ndarray, v0: np.ndarray) -> np.ndarray:
        t = np.full(u0.shape, 127.5)
        for a, ku, kv, ph in zip(self.tex_amp, self.tex_ku, self.tex_kv, self.tex_ph):
            t += a * np.cos(2 * np.pi * (ku * u0 + kv * v0) + ph)
        return t

    def render_torch(self, T_w_c: np.ndarray, a: float, b: float, device: str):
        """render() with the per-pixel loops on a torch device (float64): a 1280x1024 frame takes milliseconds instead of ~5 s.  The
        bench's full-resolution windows use it; cos / exp may differ from NumPy's in the last bit, so parity tests (which compare two
        backends on ONE window) may use either, but golden fixtures are made with render()."""
        import torch
        W, H = self.width, self.height
        f64 = dict(dtype=torch.float64, device=device)
        vv, uu = torch.meshgrid(torch.arange(H, **f64), torch.arange(W, **f64), indexing="ij")
        rx, ry = (uu - self.cx) / self.fx, (vv - self.cy) / self.fy
        R, t = T_w_c[:3, :3], T_w_c[:3, 3]

        def project(d):
            X = R[0, 0] * rx * d + R[0, 1] * ry * d + R[0, 2] * d + t[0]
            Y = R[1, 0] * rx * d + R[1, 1] * ry * d + R[1, 2] * d + t[1]
            Z = R[2, 0] * rx * d + R[2, 1] * ry * d + R[2, 2] * d + t[2]
            return self.fx * X / Z + self.cx, self.fy * Y / Z + self.cy, Z

        def depth0(u0, v0):
            z = torch.full_like(u0, 6.0)
            for amp, fu, fv, ph in zip(self.depth_amp, self.depth_fu, self.depth_fv, self.depth_ph):
                z += amp * torch.cos(2 * np.pi * (fu * u0 / self.width + fv * v0 / self.height) + ph)
            return z

        d = torch.full((H, W), 6.0, **f64)
        for _ in range(12):
            u0, v0, Z = project(d)
            d = d * depth0(u0, v0) / Z
        u0, v0, _ = project(d)
        tex = torch.full_like(u0, 127.5)
        for amp, ku, kv, ph in zip(self.tex_amp, self.tex_ku, self.tex_kv, self.tex_ph):
            tex += amp * torch.cos(2 * np.pi * (ku * u0 + kv * v0) + ph)
        return (float(np.exp(a)) * tex + b).cpu().numpy(), d.cpu().numpy()

    def render(self, T_w_c: np.ndarray, a: float = 0.0, b: float = 0.0):
        """Returns (radiance image float64 HxW = exp(a)*texture + b, z-depth map HxW) for camera pose T_w_c."""
        W, H = self.width, self.height
        uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        rx = (uu - self.cx) / self.fx
        ry = (vv - self.cy) / self.fy
        R, t = T_w_c[:3, :3], T_w_c[:3, 3]
        d = np.full((H, W), 6.0)
        for _ in range(12):
            X = R[0, 0] * rx * d + R[0, 1] * ry * d + R[0, 2] * d + t[0]
            Y = R[1, 0] * rx * d + R[1, 1] * ry * d + R[1, 2] * d + t[1]
            Z = R[2, 0] * rx * d + R[2, 1] * ry * d + R[2, 2] * d + t[2]
            u0 = self.fx * X / Z + self.cx
            v0 = self.fy * Y / Z + self.cy
            z0 = self.depth0(u0, v0)
            d = d * z0 / Z
        X = R[0, 0] * rx * d + R[0, 1] * ry * d + R[0, 2] * d + t[0]
        Y = R[1, 0] * rx * d + R[1, 1] * ry * d + R[1, 2] * d + t[1]
        Z = R[2, 0] * rx * d + R[2, 1] * ry * d + R[2, 2] * d + t[2]
        u0 = self.fx * X / Z + self.cx
        v0 = self.fy * Y / Z + self.cy
        img = np.exp(a) * self.texture(u0, v0) + b
        return img, d


def pixelinfo_from_plane(plane: np.ndarray) -> np.ndarray:
    """(I, dx, dy) AoS with central differences (one-sided x1.0 at borders) — the definition in
    src/features/src/calculate_pixelinfo.cpp:340-374.  Returns H x W x 3 float64."""
    H, W = plane.shape
    out = np.empty((H, W, 3))
    out[..., 0] = plane
    dx = np.empty_like(plane)
    dx[:, 1:-1] = 0.5 * (plane[:, 2:] - plane[:, :-2])
    dx[:, 0] = plane[:, 1] - plane[:, 0]
    dx[:, -1] = plane[:, -1] - plane[:, -2]
    dy = np.empty_like(plane)
    dy[1:-1, :] = 0.5 * (plane[2:, :] - plane[:-2, :])
    dy[0, :] = plane[1, :] - plane[0, :]
    dy[-1, :] = plane[-1, :] - plane[-2, :]
    out[..., 1] = dx
    out[..., 2] = dy
    return out


@dataclasses.dataclass
class SyntheticFrame:
    frame_id: int
    timestamp: int
    image_u8: np.ndarray  # H x W uint8
    pixelinfo: np.ndarray  # H x W x 3 float64 (level 0)
    depth: np.ndarray  # H x W z-depth (ground truth)
    T_w_c_gt: np.ndarray  # 4x4
    T_w_c_init: np.ndarray  # 4x4
    affine_gt: np.ndarray  # (a, b)
    affine_init: np.ndarray
    exposure: float
    fixed: bool
    uv: np.ndarray  # n x 2 landmark projections (integer valued float64)
    idepth_gt: np.ndarray
    idepth_init: np.ndarray
    patch: np.ndarray  # n x 8


@dataclasses.dataclass
class SyntheticWindow:
    scene: Scene
    frames: list

    @property
    def num_points(self) -> int:
        return int(sum(len(f.uv) for f in self.frames))


BASE_MOTION = np.array([0.08, 0.01, 0.02, 0.004, 0.012, 0.003])


def make_window(num_frames: int = 7, num_points: int = 2000, width: int = 640, height: int = 480, seed: int = 0,
                pose_noise: bool = True, idepth_noise: float = 2e-3, affine_jitter: bool = False,
                min_gradient: float = 4.0, quantize: bool = True, order: str = None, render_device: str = None) -> SyntheticWindow:
    """C1 of SURVEY.md §8d by default: 7 keyframes, 2000 active points, 640x480, full clique.
    `order`: order of a frame's landmarks — "random" (default: the order the pixels were drawn in), "raster" (row-major by pixel),
    "tileN" (N x N-pixel tiles in raster order, raster inside a tile: what a grid-cell feature extractor yields).
    `render_device`: render the frames with torch on that device (Scene.render_torch) — full-resolution bench windows."""
    scene = Scene.make(width, height, seed)
    rng = np.random.default_rng(seed + 1)
    frames = []
    per_frame = [num_points // num_frames + (1 if i < num_points % num_frames else 0) for i in range(num_frames)]
    for i in range(num_frames):
        T_gt = se3_exp(i * BASE_MOTION)
        a, b = (rng.uniform(-0.05, 0.05), rng.uniform(-5, 5)) if (affine_jitter and i > 0) else (0.0, 0.0)
        img, depth = scene.render_torch(T_gt, a, b, render_device) if render_device else scene.render(T_gt, a, b)
        if quantize:
            u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
            plane = u8.astype(np.float64)
        else:
            plane = np.clip(img, 0, 255)
            u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        pinfo = pixelinfo_from_plane(plane)
        if i > 0 and pose_noise:
            noise = np.concatenate([rng.normal(0, 2e-2, 3), rng.normal(0, 5e-3, 3)])
            T_init = T_gt @ se3_exp(noise)
        else:
            T_init = T_gt.copy()
        # landmarks: random integer pixels in [8, W-9] x [8, H-9] with |grad| > min_gradient
        n = per_frame[i]
        uv = np.zeros((0, 2))
        grad = np.hypot(pinfo[..., 1], pinfo[..., 2])
        while len(uv) < n:
            cand = np.stack([rng.integers(8, width - 8, 2 * n + 64), rng.integers(8, height - 8, 2 * n + 64)], axis=1)
            ok = grad[cand[:, 1], cand[:, 0]] > min_gradient
            uv = np.concatenate([uv, cand[ok].astype(np.float64)])
        uv = uv[:n]
        order = order or os.environ.get("DSOPP_SYN_ORDER", "random")  # (environment: experiments on unmodified callers)
        if order == "clump":  # experiment: every landmark inside one 48 x 48 window (cache-resident gather)
            uv = np.stack([200 + (uv[:, 0] % 48), 200 + (uv[:, 1] % 48)], axis=1)
        if order == "raster":
            uv = uv[np.lexsort((uv[:, 0], uv[:, 1]))]
        elif order.startswith("tile"):  # tiles of T x T pixels in raster order, raster inside a tile
            T = int(order[4:] or 32)
            key = (uv[:, 1] // T) * 100000 + (uv[:, 0] // T)
            uv = uv[np.lexsort((uv[:, 0], uv[:, 1], key))]
        ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
        idepth_gt = 1.0 / depth[vi, ui]
        idepth_init = idepth_gt * (1 + rng.uniform(-idepth_noise, idepth_noise, n))
        patch = np.stack([plane[vi + int(oy), ui + int(ox)] for ox, oy in PATTERN], axis=1)
        frames.append(SyntheticFrame(i, 1000 * (i + 1), u8, pinfo, depth, T_gt, T_init, np.array([a, b]), np.zeros(2), 1.0,
                                     i == 0, uv, idepth_gt, idepth_init, patch))
    return SyntheticWindow(scene, frames)


def load_window(backend, win: SyntheticWindow, statuses=None):
    """Push a synthetic window into a backend exposing push_frame / set_landmarks / set_connection
    (the oracle wrapper and the HIP wrapper share that interface).  Mirrors the call order of the reference:
    pushFrame creates the residual lists between the new frame and all previous ones
    (src/energy/problems/src/photometric_bundle_adjustment.cpp:98-124)."""
    intr = win.scene.intrinsics
    for i, f in enumerate(win.frames):
        backend.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, mat_to_params(f.T_w_c_init), f.exposure,
                           f.affine_init, f.fixed, False)
        backend.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
        for j in range(i):
            g = win.frames[j]
            for (r, t) in ((g, f), (f, g)):
                st = None if statuses is None else statuses.get((r.frame_id, t.frame_id))
                if st is None:
                    st = np.zeros(len(r.uv), dtype=np.uint8)
                backend.set_connection(r.frame_id, t.frame_id, st)
    return backend


# ImmatureLandmarkStatus (src/track/landmarks/include/track/landmarks/immature_tracking_landmark.hpp) as the C-ABI encodes it
IMMATURE_STATUS = dict(good=0, out_of_boundary=1, outlier=2, skipped=3, ill_conditioned=4, uninitialized=5, delete=6)


def new_immature_landmarks(uv, direction, patch, gradient):
    """struct-of-arrays ImmatureTrackingLandmark set with the constructor defaults (immature_tracking_landmark.hpp:93-106): the plain
    data both the device sets (capi.ImmatureSet) and the CPU checker take — no arithmetic, so it lives with the generators"""
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).copy()  # noqa: E731
    n = len(uv)
    return dict(projection=f64(uv), direction=f64(direction), patch=f64(patch), gradient=f64(gradient),
                idepth_min=np.zeros(n), idepth_max=np.full(n, 1.0 / 0.001), uniqueness=np.full(n, np.finfo(np.float64).max),
                search_pixel_interval=np.full(n, np.finfo(np.float64).max), status=np.full(n, IMMATURE_STATUS["uninitialized"], dtype=np.uint8),
                traced=np.zeros(n, dtype=np.uint8))
