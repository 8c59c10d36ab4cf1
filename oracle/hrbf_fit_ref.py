"""CPU reference of the OPTIONAL true Hermite-RBF fit (the extension BASELINE config 5 names: "batched-HRBF small-GEMM on MFMA").

TEST INFRASTRUCTURE (like everything under oracle/): only tests/ and bench.py's baseline leg may import it.

There is NO counterpart in the reference: `hrbfbase.glsl:132,153,173` use the closed-form coefficients `10 * n_i` where a
Hermite-RBF interpolant (Macedo, Gois, Velho 2011) solves a (4k x 4k) symmetric positive definite system for k centres.  This file
states the algorithm the HIP kernel `k_hrbf_fit` (hrbffusion3d_amd/csrc/k_fit.hip) implements, in float64 numpy, so that the kernel
(fp32, MFMA-tiled Cholesky) can be held to it within a tolerance, and both to analytic plane / sphere / cylinder answers:

  per pixel c with a valid vertex p_c and normal n_c, over the valid pixels q of its (2w+1)^2 window (row-major, |p_q - p_c| <=
  jump * w * z_c / fx), k >= MIN_CENTRES of them:
    rho   = support * max_q |p_q - p_c|                      common support radius  =>  the system is symmetric positive definite
    u_q   = (p_q - p_c) / rho                                dimensionless coordinates; g(u) = f(p_c + rho u) / rho keeps unit normals
    psi   = Wendland C4  phi(r) = (1 - r)^6 (35 r^2 + 18 r + 3) / 3   (psi in C^4 => the interpolant's Hessian is continuous)
    g(u)  = sum_j  alpha_j psi(u - u_j) - beta_j . grad psi(u - u_j)
    constraints  g(u_i) = 0,  grad g(u_i) = n_i     ->   A c = b,  blocks  A_ij = [[psi, -grad psi^T], [grad psi, -H psi]](u_i - u_j),
                                                         + ridge on the diagonal
    out:  grad g(0) (the fitted normal), H_x = H_u g(0) / rho, principal curvatures = eigenvalues of the shape operator
          (I - n n^T) H_x (I - n n^T) / |grad g| on the tangent plane, kmax >= kmin, with their directions.
"""
import numpy as np

MIN_CENTRES = 8
SENTINEL = 1000.0          # like the reference's curvature sentinel (depth_curvature_gradient.frag)


def wendland_c4(r):
    """(phi, F = phi'/r, G = F'/r, K = G'/r) of phi(r) = (1-r)^6 (35 r^2 + 18 r + 3) / 3 on [0, 1); zeros beyond"""
    r = np.asarray(r, np.float64)
    inside = r < 1.0
    t = np.where(inside, 1.0 - r, 0.0)
    phi = t ** 6 * (35.0 * r * r + 18.0 * r + 3.0) / 3.0
    F = -(56.0 / 3.0) * t ** 5 * (5.0 * r + 1.0)
    G = 560.0 * t ** 4
    with np.errstate(divide="ignore", invalid="ignore"):
        K = np.where(r > 0, -2240.0 * t ** 3 / np.where(r > 0, r, 1.0), 0.0)
    return phi, F, G, K


def gather_centres(vertex, normal, x, y, w, fx, jump=3.0):
    """window centres of pixel (x, y): (points (k,3), normals (k,3), index of the pixel itself) or None"""
    H, W = vertex.shape[:2]
    pc = vertex[y, x, :3].astype(np.float64)
    nc = normal[y, x, :3].astype(np.float64)
    if not (pc[2] > 0 and np.isfinite(pc).all() and np.isfinite(nc).all() and np.linalg.norm(nc) > 0.5):
        return None
    limit = jump * w * pc[2] / fx
    P, N, ic = [], [], -1
    for dy in range(-w, w + 1):
        for dx in range(-w, w + 1):
            xx, yy = x + dx, y + dy
            if xx < 0 or yy < 0 or xx >= W or yy >= H:
                continue
            p = vertex[yy, xx, :3].astype(np.float64); n = normal[yy, xx, :3].astype(np.float64)
            if not (p[2] > 0 and np.isfinite(p).all() and np.isfinite(n).all() and np.linalg.norm(n) > 0.5):
                continue
            if np.linalg.norm(p - pc) > limit:
                continue
            if dx == 0 and dy == 0:
                ic = len(P)
            P.append(p); N.append(n)
    if len(P) < MIN_CENTRES or ic < 0:
        return None
    return np.array(P), np.array(N), ic


def assemble(U, Nrm, ridge):
    """the (4k x 4k) Hermite system of centres U (dimensionless) with normals Nrm"""
    k = len(U)
    A = np.zeros((4 * k, 4 * k)); b = np.zeros(4 * k)
    for i in range(k):
        b[4 * i + 1:4 * i + 4] = Nrm[i]
        for j in range(k):
            d = U[i] - U[j]
            r = np.linalg.norm(d)
            phi, F, G, _ = wendland_c4(r)
            A[4 * i, 4 * j] = phi
            A[4 * i, 4 * j + 1:4 * j + 4] = -F * d
            A[4 * i + 1:4 * i + 4, 4 * j] = F * d
            A[4 * i + 1:4 * i + 4, 4 * j + 1:4 * j + 4] = -(F * np.eye(3) + G * np.outer(d, d))
    A += ridge * np.eye(4 * k)
    return A, b


def evaluate(U, coef, x):
    """(g, grad g, Hessian g) of the interpolant at the dimensionless point x"""
    g = 0.0; grad = np.zeros(3); Hs = np.zeros((3, 3))
    for j in range(len(U)):
        a, be = coef[4 * j], coef[4 * j + 1:4 * j + 4]
        d = x - U[j]
        r = np.linalg.norm(d)
        phi, F, G, K = wendland_c4(r)
        db = d @ be
        g += a * phi - F * db
        grad += a * F * d - (F * be + G * d * db)
        Hs += a * (F * np.eye(3) + G * np.outer(d, d))
        Hs -= G * (db * np.eye(3) + np.outer(be, d) + np.outer(d, be)) + K * np.outer(d, d) * db
    return g, grad, Hs


def curvature_from(grad, Hx):
    """principal curvatures (kmax, dir, kmin, dir) of the level set through the point"""
    gn = np.linalg.norm(grad)
    n = grad / gn
    a = np.array([1.0, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1.0, 0])
    t1 = np.cross(n, a); t1 /= np.linalg.norm(t1)
    t2 = np.cross(n, t1)
    M = np.array([[t1 @ Hx @ t1, t1 @ Hx @ t2], [t2 @ Hx @ t1, t2 @ Hx @ t2]]) / gn
    w, V = np.linalg.eigh(0.5 * (M + M.T))
    d_min = V[0, 0] * t1 + V[1, 0] * t2
    d_max = V[0, 1] * t1 + V[1, 1] * t2
    return w[1], d_max, w[0], d_min, n, gn


def fit_pixel(vertex, normal, x, y, w=2, fx=528.0, support=1.25, ridge=1e-6, jump=3.0):
    """-> (kmax, dir_max, kmin, dir_min, normal, |grad|, k, cond) or None (sentinel)"""
    got = gather_centres(vertex, normal, x, y, w, fx, jump)
    if got is None:
        return None
    P, Nrm, ic = got
    pc = P[ic]
    rho = support * np.max(np.linalg.norm(P - pc, axis=1))
    if not rho > 0:
        return None
    U = (P - pc) / rho
    A, b = assemble(U, Nrm, ridge)
    coef = np.linalg.solve(A, b)
    _, grad, Hu = evaluate(U, coef, np.zeros(3))
    kmax, dmax, kmin, dmin, n, gn = curvature_from(grad, Hu / rho)
    return kmax, dmax, kmin, dmin, n, gn, len(P), np.linalg.cond(A)


def fit_image(vertex, normal, w=2, fx=528.0, support=1.25, ridge=1e-6, jump=3.0, step=1):
    """FIT_CURV1 (dir_max, kmax), FIT_CURV2 (dir_min, kmin), FIT_NORMAL (n, |grad|) as (H, W, 4) float64; sentinel rows where no fit"""
    H, W = vertex.shape[:2]
    c1 = np.zeros((H, W, 4)); c2 = np.zeros((H, W, 4)); nn = np.zeros((H, W, 4))
    c1[..., 3] = SENTINEL; c2[..., 3] = SENTINEL
    for y in range(0, H, step):
        for x in range(0, W, step):
            r = fit_pixel(vertex, normal, x, y, w, fx, support, ridge, jump)
            if r is None:
                continue
            kmax, dmax, kmin, dmin, n, gn = r[:6]
            c1[y, x] = (*dmax, kmax); c2[y, x] = (*dmin, kmin); nn[y, x] = (*n, gn)
    return c1, c2, nn
