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


# ---------------------------------------------------------------------------------------------------------------------
# two-frame direct alignment (the coarse tracker's per-level problem, PatternSize = 1) — written from the definition:
#   r_i(T, a, b) = (I_t(pi(T_tr x_i)) - b_t) - (e_t / e_r) exp(a_t - a_r) (I_r,i - b_r),   E = sum_i rho_sigma(|r_i|) + priors,
# reference statement: src/energy/problems/src/eigen_pose_alignment.cpp:55-206.  Independent of oracle/pose_alignment.hpp.
# ---------------------------------------------------------------------------------------------------------------------
def align_project(intr_r, intr_t, T_tr, u, v, idepth):
    """target pixel coordinates (n x 2) and depths z of reference pixels (u, v) with inverse depth idepth"""
    fxr, fyr, cxr, cyr = intr_r
    fxt, fyt, cxt, cyt = intr_t
    rays = np.stack([(u - cxr) / fxr, (v - cyr) / fyr, np.ones_like(u)], axis=1)
    X = rays @ T_tr[:3, :3].T + idepth[:, None] * T_tr[:3, 3][None, :]
    return np.stack([fxt * X[:, 0] / X[:, 2] + cxt, fyt * X[:, 1] / X[:, 2] + cyt], axis=1), X[:, 2]


def align_residuals(pix_t, intr_r, size_r, intr_t, T_tr, ab_r, ab_t, e_r, e_t, u, v, idepth, intensity):
    """(valid mask, residuals, interpolated stored (I, Ix, Iy) samples n x 3).  A point is valid when its reference pixel and its
    reprojection lie inside the 4-pixel ROI border, z > 0 and the inverse depth is admissible."""
    ht, wt = pix_t.shape[:2]
    wr, hr = size_r
    pts, z = align_project(intr_r, intr_t, T_tr, u, v, idepth)
    with np.errstate(invalid="ignore"):
        ok = (idepth > -1e-4) & (idepth < 1010.0) & (u >= 4) & (v >= 4) & (u <= wr - 5) & (v <= hr - 5) & (z > 0)
        ok &= (pts[:, 0] >= 4) & (pts[:, 1] >= 4) & (pts[:, 0] <= wt - 5) & (pts[:, 1] <= ht - 5)
    r = np.zeros(len(u))
    samples = np.zeros((len(u), 3))
    s = (e_t / e_r) * np.exp(ab_t[0] - ab_r[0])
    for i in np.nonzero(ok)[0]:
        samples[i] = bilinear(pix_t, pts[i, 0], pts[i, 1])
        r[i] = (samples[i, 0] - ab_t[1]) - s * (intensity[i] - ab_r[1])
    return ok, r, samples


def align_energy(r, ok, sigma, ab_t, reg):
    a = np.abs(r[ok])
    e = np.where(a > sigma, sigma * a - sigma * sigma / 2, a * a / 2).sum()
    return float(e + (ab_t[0] * reg[0] * ab_t[0] + ab_t[1] * reg[1] * ab_t[1]) / 2), int(ok.sum())


def align_normal_equations(pix_t, intr_r, size_r, intr_t, T_tr, ab_r, ab_t, e_r, e_t, u, v, idepth, intensity, sigma, reg, h=1e-6):
    """J^T W J and J^T W r of the alignment energy in the parameters (eps: LEFT perturbation T <- exp(eps) T, a_t, b_t):
    the geometry is differentiated by central differences, the image by its interpolated stored gradient (what the reference's
    PixelMap::Evaluate returns); W is the IRLS weight of the Huber loss; the affine prior enters as reg * (a, b)."""
    ok, r, samples = align_residuals(pix_t, intr_r, size_r, intr_t, T_tr, ab_r, ab_t, e_r, e_t, u, v, idepth, intensity)
    n = len(u)
    J = np.zeros((n, 8))
    for c in range(6):
        e = np.zeros(6)
        e[c] = h
        dp = (align_project(intr_r, intr_t, exp_se3(e) @ T_tr, u, v, idepth)[0] - align_project(intr_r, intr_t, exp_se3(-e) @ T_tr, u, v, idepth)[0]) / (2 * h)
        J[:, c] = samples[:, 1] * dp[:, 0] + samples[:, 2] * dp[:, 1]
    s = (e_t / e_r) * np.exp(ab_t[0] - ab_r[0])
    J[:, 6] = -s * (intensity - ab_r[1])   # d r / d a_t
    J[:, 7] = -1.0                          # d r / d b_t
    J[~ok] = 0
    w = np.where(np.abs(r) > sigma, sigma / np.maximum(np.abs(r), 1e-300), 1.0) * ok
    H = J.T @ (w[:, None] * J)
    g = J.T @ (w * r)
    H[6, 6] += reg[0]
    H[7, 7] += reg[1]
    g[6] += reg[0] * ab_t[0]
    g[7] += reg[1] * ab_t[1]
    return H, g, ok, r


def align_lm_step(H, g, lam):
    """the damped Gauss-Newton step in the true parameters: delta = -(H + lam diag(H))^-1 g  (levenberg_marquardt_algorithm.hpp:100-112)"""
    return -np.linalg.solve(H + lam * np.diag(np.diag(H)), g)


# ---------------------------------------------------------------------------------------------------------------------
# uncertainty: pseudo-inverse with dropped directions, covariance of a relative pose
# ---------------------------------------------------------------------------------------------------------------------
def pinv_drop_smallest(H, n_drop):
    """pseudo-inverse of a symmetric matrix with its n_drop smallest singular values treated as zero
    (eigen_photometric_bundle_adjustment.cpp:31-45: JacobiSVD, the last singular values' inverses set to 0)"""
    w, V = np.linalg.eigh((H + H.T) / 2)
    order = np.argsort(np.abs(w))
    inv = np.zeros_like(w)
    keep = order[n_drop:]
    inv[keep] = 1.0 / w[keep]
    return (V * inv[None, :]) @ V.T


def log_se3(T):
    """twist (upsilon, omega) of a 4x4 rigid transform through the matrix logarithm's series on the rotation (|omega| < pi) —
    Rodrigues-free: atan2 of the skew part for the angle, the closed-form V^-1 is avoided by solving V upsilon = t with V from its series"""
    R, t = T[:3, :3], T[:3, 3]
    W = (R - R.T) / 2
    w = np.array([W[2, 1], W[0, 2], W[1, 0]])
    s, c = np.linalg.norm(w), (np.trace(R) - 1) / 2
    th = np.arctan2(s, c)
    om = w if s < 1e-12 else w * (th / s)
    Om = hat(om)
    V = np.eye(3)
    term = np.eye(3)
    for k in range(1, 25):
        term = term @ Om / (k + 1)
        V = V + term
    return np.concatenate([np.linalg.solve(V, t), om])


def relative_pose_covariance_fd(T_w_1, T_w_2, sigma_11, sigma_22, sigma_12, h=1e-6):
    """covariance of eps in T_12 = (T_w_1^-1 T_w_2) exp(eps) when T_w_i = T_w_i exp(eps_i), (eps_1, eps_2) ~ N(0, [[S11, S12], [S12^T, S22]]):
    first-order propagation with the Jacobian of eps(eps_1, eps_2) = log((T_1^-1 T_2)^-1 (T_1 exp(eps_1))^-1 (T_2 exp(eps_2))) taken by
    central differences (statement of se3_motion.hpp:140-158 without its closed form)"""
    T12_inv = np.linalg.inv(np.linalg.inv(T_w_1) @ T_w_2)

    def eps(e1, e2):
        return log_se3(T12_inv @ np.linalg.inv(T_w_1 @ exp_se3(e1)) @ (T_w_2 @ exp_se3(e2)))

    J = np.zeros((6, 12))
    for c in range(6):
        e = np.zeros(6)
        e[c] = h
        J[:, c] = (eps(e, np.zeros(6)) - eps(-e, np.zeros(6))) / (2 * h)
        J[:, 6 + c] = (eps(np.zeros(6), e) - eps(np.zeros(6), -e)) / (2 * h)
    S = np.block([[sigma_11, sigma_12], [sigma_12.T, sigma_22]])
    return J @ S @ J.T


# ---------------------------------------------------------------------------------------------------------------------
# point statuses: the 3rd-quartile rule
# ---------------------------------------------------------------------------------------------------------------------
def third_quartile_threshold(energies, sigma):
    """energy above which a residual becomes an outlier: the element of rank floor(0.75 n) (0-based) of the energies of all residuals
    in state OK, plus sigma^2 / 2  (photometric_bundle_adjustment.cpp:344-361)"""
    e = np.asarray(energies, dtype=np.float64)
    if e.size == 0:
        return 0.0
    k = int(e.size * 0.75)
    return float(np.partition(e, k)[k] + sigma * sigma / 2)
