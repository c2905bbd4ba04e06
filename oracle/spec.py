"""ORACLE — TEST INFRASTRUCTURE ONLY.  Independent NumPy float64 specification of the photometric BA maths.

Written from the equations (the cleanest statement of the residual in the reference is
``src/energy/problems/internal/energy/problems/cost_functors/bundle_adjustment_photometric_cost_functor.hpp:79-119``),
NOT from the C++ oracle: it shares no code with oracle/*.hpp and exists to cross-check it — residuals and energies
directly, Jacobians through finite differences of the geometry chained with the *interpolated stored* image gradients
(what ``PixelMap::Evaluate`` returns, ``src/features/include/features/camera/pixel_map.hpp:20-40,246-259``), and the
normal equations through brute-force dense ``J^T W J``.
"""
from __future__ import annotations

import numpy as np

PATTERN = np.array([[0, 2], [-1, 1], [1, 1], [-2, 0], [0, 0], [2, 0], [-1, -1], [0, -2]], dtype=np.float64)


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def exp_se3(xi):
    """4x4 matrix exponential of the twist (upsilon, omega) through its power series (independent of Rodrigues)."""
    xi = np.asarray(xi, dtype=np.float64)
    A = np.zeros((4, 4))
    A[:3, :3] = hat(xi[3:])
    A[:3, 3] = xi[:3]
    # scaling and squaring with a long Taylor series
    s = max(0, int(np.ceil(np.log2(max(np.abs(A).sum(), 1e-30)))) + 4)
    As = A / (2.0**s)
    E = np.eye(4)
    term = np.eye(4)
    for k in range(1, 20):
        term = term @ As / k
        E = E + term
    for _ in range(s):
        E = E @ E
    return E


def quat_to_mat(p):
    x, y, z, w = p[:4]
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n
    R = np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                  [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                  [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p[4:7]
    return T


def adjoint(T):
    R, t = T[:3, :3], T[:3, 3]
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[3:, 3:] = R
    A[:3, 3:] = hat(t) @ R
    return A


def bilinear(pixelinfo, x, y):
    """(I, Ix, Iy) blend at the 4 neighbours with truncation toward zero, as PixelMap does."""
    ix, iy = int(x), int(y)
    dx, dy = x - ix, y - iy
    return ((1 - dx) * (1 - dy) * pixelinfo[iy, ix] + dx * (1 - dy) * pixelinfo[iy, ix + 1] + (1 - dx) * dy * pixelinfo[iy + 1, ix] +
            dx * dy * pixelinfo[iy + 1, ix + 1])


def project_pattern(intr_r, intr_t, T_tr, uv, idepth):
    """target pixel coordinates of the 8 pattern points (8x2) and their z in the target camera."""
    fxr, fyr, cxr, cyr = intr_r
    fxt, fyt, cxt, cyt = intr_t
    pts = uv[None, :] + PATTERN
    rays = np.stack([(pts[:, 0] - cxr) / fxr, (pts[:, 1] - cyr) / fyr, np.ones(8)], axis=1)
    X = rays @ T_tr[:3, :3].T + idepth * T_tr[:3, 3][None, :]
    out = np.stack([fxt * X[:, 0] / X[:, 2] + cxt, fyt * X[:, 1] / X[:, 2] + cyt], axis=1)
    return out, X[:, 2]


def in_roi(pts, width, height):
    return bool(np.all(pts[:, 0] >= 4) and np.all(pts[:, 1] >= 4) and np.all(pts[:, 0] <= width - 5) and np.all(pts[:, 1] <= height - 5))


def relative_pose(T_r0, T_t0, xi_r, xi_t):
    """T_tr = exp(-xi_t) * inv(T_t0) * T_r0 * exp(xi_r)  (evaluate_jacobians.hpp:47-49)."""
    return exp_se3(-np.asarray(xi_t)) @ np.linalg.inv(T_t0) @ T_r0 @ exp_se3(xi_r)


class SpecFrame:
    def __init__(self, T0_params, ab0, eps, pixelinfo, intr, exposure=1.0):
        self.T0 = quat_to_mat(np.asarray(T0_params, dtype=np.float64))
        self.ab0 = np.asarray(ab0, dtype=np.float64)
        self.eps = np.asarray(eps, dtype=np.float64)
        self.pixelinfo = pixelinfo
        self.intr = np.asarray(intr, dtype=np.float64)
        self.exposure = exposure
        self.height, self.width = pixelinfo.shape[:2]


def residual8(fr: SpecFrame, ft: SpecFrame, uv, idepth, patch, xi_r=None, xi_t=None, ab_r=None, ab_t=None):
    """Returns (ok, r[8], target points 8x2, samples 8x3).  ok=False when the pattern leaves either ROI / z<=0."""
    xi_r = fr.eps[:6] if xi_r is None else xi_r
    xi_t = ft.eps[:6] if xi_t is None else xi_t
    ab_r = fr.ab0 + fr.eps[6:] if ab_r is None else ab_r
    ab_t = ft.ab0 + ft.eps[6:] if ab_t is None else ab_t
    T_tr = relative_pose(fr.T0, ft.T0, xi_r, xi_t)
    pts, z = project_pattern(fr.intr, ft.intr, T_tr, uv, idepth)
    ref_pts = uv[None, :] + PATTERN
    ok = (-1e-4 < idepth < 1010.0) and in_roi(ref_pts, fr.width, fr.height) and bool(np.all(z > 0)) and in_roi(pts, ft.width, ft.height)
    if not ok:
        return False, np.zeros(8), pts, None
    samples = np.stack([bilinear(ft.pixelinfo, p[0], p[1]) for p in pts])
    s = (ft.exposure / fr.exposure) * np.exp(ab_t[0] - ab_r[0])
    r = (samples[:, 0] - ab_t[1]) - s * (patch - ab_r[1])
    return True, r, pts, samples


def huber(r, sigma):
    n2 = float(r @ r)
    if n2 > sigma * sigma:
        n = np.sqrt(n2)
        return sigma * n - sigma * sigma / 2, sigma / n
    return n2 / 2, 1.0


def jacobians_fd(fr: SpecFrame, ft: SpecFrame, uv, idepth, patch, h=1e-6):
    """Jacobian blocks (8x8 ref, 8x8 tgt, 8 idepth) at the current state, non-FEJ:
    geometry differentiated by central differences, image gradient = interpolated stored gradient."""
    ok, r, pts, samples = residual8(fr, ft, uv, idepth, patch)
    assert ok
    gx, gy = samples[:, 1], samples[:, 2]

    def pts_of(xi_r, xi_t, d):
        T_tr = relative_pose(fr.T0, ft.T0, xi_r, xi_t)
        return project_pattern(fr.intr, ft.intr, T_tr, uv, d)[0]

    J_r, J_t = np.zeros((8, 8)), np.zeros((8, 8))
    for c in range(6):
        e = np.zeros(6)
        e[c] = h
        dp = (pts_of(fr.eps[:6] + e, ft.eps[:6], idepth) - pts_of(fr.eps[:6] - e, ft.eps[:6], idepth)) / (2 * h)
        J_r[:, c] = gx * dp[:, 0] + gy * dp[:, 1]
        dp = (pts_of(fr.eps[:6], ft.eps[:6] + e, idepth) - pts_of(fr.eps[:6], ft.eps[:6] - e, idepth)) / (2 * h)
        J_t[:, c] = gx * dp[:, 0] + gy * dp[:, 1]
    dp = (pts_of(fr.eps[:6], ft.eps[:6], idepth + h) - pts_of(fr.eps[:6], ft.eps[:6], idepth - h)) / (2 * h)
    J_d = gx * dp[:, 0] + gy * dp[:, 1]
    ab_r, ab_t = fr.ab0 + fr.eps[6:], ft.ab0 + ft.eps[6:]
    s = (ft.exposure / fr.exposure) * np.exp(ab_t[0] - ab_r[0])
    # d r / d a_r = +s (patch - b_r);  d r / d b_r = +s ; d r / d a_t = -s (patch - b_r); d r / d b_t = -1
    J_r[:, 6] = s * (patch - ab_r[1])
    J_r[:, 7] = s
    J_t[:, 6] = -s * (patch - ab_r[1])
    J_t[:, 7] = -1
    return r, J_r, J_t, J_d
