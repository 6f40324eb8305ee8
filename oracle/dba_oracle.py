"""CPU restatement (numpy, float64) of the dense bundle adjustment path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the shipped package.

PARITY UNPINNED.  The reference calls ``droid_backends.ba`` and friends
(/root/reference/scripts/frontend/depth_video.py:380-381,289-299,
/root/reference/scripts/frontend/modules/corr.py:12,19) whose CUDA sources are
in an un-vendored submodule (/root/reference/.gitmodules:7-9, pip
droid_backends 0.0.0, /root/reference/backup/piplist.txt:47).  The algorithm is
restated from the reference's own pure-PyTorch twin of that kernel, which cannot
be imported here because it needs lietorch and torch_scatter:
  system assembly   /root/reference/scripts/frontend/geom/ba.py:29-104
  damping / Schur   /root/reference/scripts/frontend/geom/chol.py:32-73
  projection + Jacobians /root/reference/scripts/frontend/geom/projective_ops.py:18-125
Caller conventions: poses are world->camera (tx,ty,tz,qx,qy,qz,qw); eta has one
row per depth variable; ``ba`` works in place
(/root/reference/scripts/frontend/covisible_graph.py:337-344).
Entry-point arguments the twin does not have (t0, t1, disps_sens,
motion_only) follow the call site at depth_video.py:380-381: poses in
[t0,t1) are free, depth variables are the sorted unique of arange(t0,t1) and
ii, and a positive disps_sens adds the prior 0.05*(disp - disps_sens)^2 in
place of eta.

The restatement is validated in tests/test_oracle_dba.py by finite differences
of the reprojection (Jacobians), a brute-force dense normal-equation solve, and
the zero-residual fixed point.
"""
import numpy as np

MIN_DEPTH = 0.2
SENSOR_ALPHA = 0.05


# ---------------------------------------------------------------------------
# SE3 on (t, q=xyzw) vectors
# ---------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_to_mat(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def se3_inv(p):
    t, q = p[..., :3], p[..., 3:]
    qi = q * np.array([-1.0, -1.0, -1.0, 1.0])
    ti = -np.einsum("...ij,...j->...i", quat_to_mat(qi), t)
    return np.concatenate([ti, qi], -1)


def se3_mul(a, b):
    ta, qa, tb, qb = a[..., :3], a[..., 3:], b[..., :3], b[..., 3:]
    t = np.einsum("...ij,...j->...i", quat_to_mat(qa), tb) + ta
    return np.concatenate([t, quat_mul(qa, qb)], -1)


def hat(v):
    M = np.zeros(v.shape[:-1] + (3, 3))
    M[..., 0, 1], M[..., 0, 2] = -v[..., 2], v[..., 1]
    M[..., 1, 0], M[..., 1, 2] = v[..., 2], -v[..., 0]
    M[..., 2, 0], M[..., 2, 1] = -v[..., 1], v[..., 0]
    return M


def se3_exp(xi):
    """Exponential of a twist (tau, phi) -> (t, q)."""
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1)
    th = np.sqrt(th2)
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * ths) / ths)          # sin(th/2)/th
    q = np.concatenate([a[..., None] * phi, np.cos(0.5 * th)[..., None]], -1)
    b = np.where(small, 0.5 - th2 / 24.0, (1 - np.cos(ths)) / (ths * ths))
    c = np.where(small, 1.0 / 6 - th2 / 120.0, (ths - np.sin(ths)) / (ths ** 3))
    Phi = hat(phi)
    V = np.eye(3) + b[..., None, None] * Phi + c[..., None, None] * (Phi @ Phi)
    t = np.einsum("...ij,...j->...i", V, tau)
    return np.concatenate([t, q], -1)


def adjoint(p):
    """6x6 adjoint for twists ordered (tau, phi)."""
    R = quat_to_mat(p[..., 3:])
    Ad = np.zeros(p.shape[:-1] + (6, 6))
    Ad[..., :3, :3] = R
    Ad[..., 3:, 3:] = R
    Ad[..., :3, 3:] = hat(p[..., :3]) @ R
    return Ad


def relative_pose(poses, ii, jj):
    """G_ij = G_j G_i^-1, composed as q_ij = q_j q_i^-1, t_ij = t_j - R(q_ij) t_i.
    For exactly unit quaternions this equals se3_mul(G_j, se3_inv(G_i)); for quaternions
    stored in float32 (norm off by ~1e-8) the two forms differ by that much, and the form
    with a single rotation is the one a kernel evaluates."""
    qi_inv = poses[ii, 3:] * np.array([-1.0, -1.0, -1.0, 1.0])
    q = quat_mul(poses[jj, 3:], qi_inv)
    t = poses[jj, :3] - np.einsum("...ij,...j->...i", quat_to_mat(q), poses[ii, :3])
    G = np.concatenate([t, q], -1)
    same = ii == jj
    if same.any():  # stereo baseline hack of projective_ops.py:105
        G[same] = np.array([-0.1, 0, 0, 0, 0, 0, 1.0])
    return G


# ---------------------------------------------------------------------------
# projection
# ---------------------------------------------------------------------------
def reproject(poses, disps, intr, ii, jj, jacobian=False):
    """coords[E,h,w,2], valid[E,h,w] (and Ji,Jj[E,h,w,2,6], Jz[E,h,w,2])."""
    poses = np.asarray(poses, np.float64)
    disps = np.asarray(disps, np.float64)
    fx, fy, cx, cy = (float(v) for v in intr)
    ht, wd = disps.shape[1:]
    y, x = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64), indexing="ij")
    X0 = np.stack([np.broadcast_to((x - cx) / fx, (len(ii), ht, wd)),
                   np.broadcast_to((y - cy) / fy, (len(ii), ht, wd)),
                   np.ones((len(ii), ht, wd)), disps[ii]], -1)
    G = relative_pose(poses, ii, jj)
    R = quat_to_mat(G[:, 3:])
    X1 = np.einsum("eij,ehwj->ehwi", R, X0[..., :3]) + G[:, None, None, :3] * X0[..., 3:4]
    X, Y, Z, D = X1[..., 0], X1[..., 1], X1[..., 2], X0[..., 3]
    Zc = np.where(Z < 0.5 * MIN_DEPTH, 1.0, Z)
    d = 1.0 / Zc
    coords = np.stack([fx * X * d + cx, fy * Y * d + cy], -1)
    valid = ((Z > MIN_DEPTH) & (X0[..., 2] > MIN_DEPTH)).astype(np.float64)
    if not jacobian:
        return coords, valid
    o = np.zeros_like(d)
    Jp = np.stack([np.stack([fx * d, o, -fx * X * d * d], -1),
                   np.stack([o, fy * d, -fy * Y * d * d], -1)], -2)          # [E,h,w,2,3]
    Ja = np.zeros(d.shape + (3, 6))
    Ja[..., 0, 0] = D; Ja[..., 1, 1] = D; Ja[..., 2, 2] = D
    Ja[..., 0, 4] = Z; Ja[..., 0, 5] = -Y
    Ja[..., 1, 3] = -Z; Ja[..., 1, 5] = X
    Ja[..., 2, 3] = Y; Ja[..., 2, 4] = -X
    Jj = Jp @ Ja                                                             # [E,h,w,2,6]
    Ad = adjoint(G)
    Ji = -np.einsum("ehwcm,emn->ehwcn", Jj, Ad)
    Jz = np.einsum("ehwcm,em->ehwc", Jp, G[:, :3])                          # d coords / d disp_i
    return coords, valid, (Ji, Jj, Jz)


# ---------------------------------------------------------------------------
# one Gauss-Newton iteration
# ---------------------------------------------------------------------------
def depth_variables(ii, t0, t1):
    kx = np.unique(np.concatenate([np.arange(t0, t1), np.asarray(ii)]))
    kk = np.searchsorted(kx, ii)
    return kx, kk


def build_system(poses, disps, intr, disps_sens, target, weight, eta, ii, jj, t0, t1):
    """Returns dict with H[P,P,6,6], v[P,6], E[P,M,6,HW], C[M,HW], w[M,HW], kx, kk and the
    per-edge pieces (Ei,Ej,Ck,wk,Hii,Hij,Hji,Hjj,vi,vj)."""
    ii = np.asarray(ii); jj = np.asarray(jj)
    ht, wd = disps.shape[1:]
    HW = ht * wd
    Ecount = len(ii)
    P = t1 - t0
    coords, valid, (Ji, Jj, Jz) = reproject(poses, disps, intr, ii, jj, jacobian=True)
    tgt = np.asarray(target, np.float64).transpose(0, 2, 3, 1)   # [E,h,w,2]
    wgt = np.asarray(weight, np.float64).transpose(0, 2, 3, 1)
    r = (tgt - coords).reshape(Ecount, HW, 2)
    w = (0.001 * valid[..., None] * wgt).reshape(Ecount, HW, 2)
    Ji = Ji.reshape(Ecount, HW, 2, 6); Jj = Jj.reshape(Ecount, HW, 2, 6); Jz = Jz.reshape(Ecount, HW, 2)
    wJi = w[..., None] * Ji; wJj = w[..., None] * Jj
    Hii = np.einsum("epcm,epcn->emn", wJi, Ji); Hij = np.einsum("epcm,epcn->emn", wJi, Jj)
    Hji = np.einsum("epcm,epcn->emn", wJj, Ji); Hjj = np.einsum("epcm,epcn->emn", wJj, Jj)
    vi = np.einsum("epcm,epc->em", wJi, r); vj = np.einsum("epcm,epc->em", wJj, r)
    Ei = np.einsum("epcm,epc->emp", wJi, Jz); Ej = np.einsum("epcm,epc->emp", wJj, Jz)
    wk = (w * r * Jz).sum(-1); Ck = (w * Jz * Jz).sum(-1)
    kx, kk = depth_variables(ii, t0, t1)
    M = len(kx)
    eta = np.asarray(eta, np.float64).reshape(-1, HW)
    assert eta.shape[0] == M, f"eta has {eta.shape[0]} rows but there are {M} depth variables"
    H = np.zeros((P, P, 6, 6)); v = np.zeros((P, 6)); E = np.zeros((P, M, 6, HW))
    C = np.zeros((M, HW)); wv = np.zeros((M, HW))
    ip, jp = ii - t0, jj - t0
    for e in range(Ecount):
        fi, fj = 0 <= ip[e] < P, 0 <= jp[e] < P
        if fi:
            H[ip[e], ip[e]] += Hii[e]; v[ip[e]] += vi[e]; E[ip[e], kk[e]] += Ei[e]
        if fj:
            H[jp[e], jp[e]] += Hjj[e]; v[jp[e]] += vj[e]; E[jp[e], kk[e]] += Ej[e]
        if fi and fj:
            H[ip[e], jp[e]] += Hij[e]; H[jp[e], ip[e]] += Hji[e]
        C[kk[e]] += Ck[e]; wv[kk[e]] += wk[e]
    ds = np.asarray(disps_sens, np.float64)[kx].reshape(M, HW)
    m = (ds > 0).astype(np.float64)
    C = C + m * SENSOR_ALPHA + (1 - m) * eta
    wv = wv - m * SENSOR_ALPHA * (np.asarray(disps, np.float64)[kx].reshape(M, HW) - ds)
    return dict(H=H, v=v, E=E, C=C, w=wv, kx=kx, kk=kk, Ei=Ei, Ej=Ej, Ck=Ck, wk=wk, Hii=Hii,
                Hij=Hij, Hji=Hji, Hjj=Hjj, vi=vi, vj=vj, P=P, M=M, HW=HW)


def reduced_system(sys_, lm=0.0, ep=0.0):
    """Damped pose block minus the Schur complement: S[6P,6P], rhs[6P]."""
    P, M, HW = sys_["P"], sys_["M"], sys_["HW"]
    H = sys_["H"].transpose(0, 2, 1, 3).reshape(6 * P, 6 * P).copy()
    d = np.diag(H).copy()
    H[np.diag_indices(6 * P)] += ep + lm * d
    E = sys_["E"].transpose(0, 2, 1, 3).reshape(6 * P, M * HW)
    Q = 1.0 / sys_["C"].reshape(-1)
    S = H - (E * Q) @ E.T
    rhs = sys_["v"].reshape(-1) - E @ (Q * sys_["w"].reshape(-1))
    return S, rhs, E, Q


def solve_spd(S, b):
    """Cholesky solve; zeros when the factorisation breaks down (chol.py:9-17)."""
    try:
        L = np.linalg.cholesky(S)
    except np.linalg.LinAlgError:
        return np.zeros_like(b)
    return np.linalg.solve(L.T, np.linalg.solve(L, b))


def ba_step(poses, disps, intr, disps_sens, target, weight, eta, ii, jj, t0, t1, lm, ep,
            motion_only):
    sys_ = build_system(poses, disps, intr, disps_sens, target, weight, eta, ii, jj, t0, t1)
    P, M, HW = sys_["P"], sys_["M"], sys_["HW"]
    poses = poses.copy(); disps = disps.copy()
    if motion_only:
        H = sys_["H"].transpose(0, 2, 1, 3).reshape(6 * P, 6 * P).copy()
        H[np.diag_indices(6 * P)] += ep + lm * np.diag(H)
        dx = solve_spd(H, sys_["v"].reshape(-1))
        dz = None
    else:
        S, rhs, E, Q = reduced_system(sys_, lm, ep)
        dx = solve_spd(S, rhs)
        dz = (Q * (sys_["w"].reshape(-1) - E.T @ dx)).reshape(M, HW)
    dxr = dx.reshape(P, 6)
    poses[t0:t1] = se3_mul(se3_exp(dxr), poses[t0:t1])
    if dz is not None:
        ht, wd = disps.shape[1:]
        new = disps[sys_["kx"]] + dz.reshape(M, ht, wd)
        new = np.where(new > 10, 0.0, new)          # ba.py:101-102
        disps[sys_["kx"]] = np.clip(new, 0.0, None)
    return poses, disps, dxr, dz, sys_


def ba(poses, disps, intr, disps_sens, target, weight, eta, ii, jj, t0, t1, itrs=2, lm=1e-4,
       ep=0.1, motion_only=False, state_dtype=np.float64):
    """Returns (poses, disps, last dx, last dz). Inputs are not modified.
    state_dtype=np.float32 rounds poses / disps after every iteration, as an
    implementation that keeps its state in float32 buffers does."""
    poses = np.asarray(poses, np.float64).copy()
    disps = np.asarray(disps, np.float64).copy()
    dx = dz = None
    for _ in range(itrs):
        poses, disps, dx, dz, _ = ba_step(poses, disps, intr, disps_sens, target, weight, eta,
                                          np.asarray(ii), np.asarray(jj), t0, t1, lm, ep,
                                          motion_only)
        poses = poses.astype(state_dtype).astype(np.float64)
        disps = disps.astype(state_dtype).astype(np.float64)
    return poses, disps, dx, dz


# ---------------------------------------------------------------------------
# frame distance (depth_video.py:289-299): mean induced flow magnitude
# ---------------------------------------------------------------------------
def frame_distance(poses, disps, intr, ii, jj, beta):
    """Mean induced-flow magnitude i->j, mixing the full relative motion (weight beta) and
    its translation-only part (weight 1-beta); a term counts only where the transformed
    point has z > MIN_DEPTH.  Pairs whose summed weight is below 75 % of the pixels return
    the sentinel 1000 (they are never selected as neighbours)."""
    poses = np.asarray(poses, np.float64)
    ii = np.asarray(ii); jj = np.asarray(jj)
    ht, wd = disps.shape[1:]
    fx, fy, cx, cy = (float(v) for v in intr)
    y, x = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64), indexing="ij")
    d = np.asarray(disps, np.float64)[ii]
    X0 = np.stack([np.broadcast_to((x - cx) / fx, d.shape), np.broadcast_to((y - cy) / fy, d.shape),
                   np.ones_like(d)], -1)
    G = relative_pose(poses, ii, jj)
    R = quat_to_mat(G[:, 3:])

    def flow(X1):
        Z = X1[..., 2]
        ok = Z > MIN_DEPTH
        Zs = np.where(ok, Z, 1.0)
        du = fx * X1[..., 0] / Zs + cx - x
        dv = fy * X1[..., 1] / Zs + cy - y
        return np.sqrt(du * du + dv * dv) * ok, ok.astype(np.float64)

    f1, v1 = flow(np.einsum("eij,ehwj->ehwi", R, X0) + G[:, None, None, :3] * d[..., None])
    f2, v2 = flow(X0 + G[:, None, None, :3] * d[..., None])
    num = (beta * f1 + (1 - beta) * f2).reshape(len(ii), -1).sum(1)
    den = (beta * v1 + (1 - beta) * v2).reshape(len(ii), -1).sum(1)
    return np.where(den / (ht * wd) < 0.75, 1000.0, num / np.maximum(den, 1e-30))


# ---------------------------------------------------------------------------
# correlation lookup (modules/corr.py:23-50): bilinear sample of a
# (2r+1)^2 window around coords in volume[e, y, x, :, :]
# ---------------------------------------------------------------------------
def corr_index_forward(volume, coords, r):
    """volume[E,h1,w1,h2,w2], coords[E,2,h1,w1] (x,y) -> [E,(2r+1),(2r+1),h1,w1].
    Window element (i,j) samples at (x + j - r, y + i - r); outside = 0."""
    volume = np.asarray(volume, np.float64)
    coords = np.asarray(coords, np.float64)
    E, h1, w1, h2, w2 = volume.shape
    rd = 2 * r + 1
    out = np.zeros((E, rd, rd, h1, w1))
    x0 = np.floor(coords[:, 0]); y0 = np.floor(coords[:, 1])
    dx = coords[:, 0] - x0; dy = coords[:, 1] - y0
    e_idx, yy, xx = np.meshgrid(np.arange(E), np.arange(h1), np.arange(w1), indexing="ij")

    def fetch(yi, xi):
        ok = (yi >= 0) & (yi < h2) & (xi >= 0) & (xi < w2)
        v = volume[e_idx, yy, xx, np.clip(yi, 0, h2 - 1).astype(int), np.clip(xi, 0, w2 - 1).astype(int)]
        return np.where(ok, v, 0.0)

    for i in range(rd):
        for j in range(rd):
            xi = x0 + (j - r); yi = y0 + (i - r)
            out[:, i, j] = ((1 - dx) * (1 - dy) * fetch(yi, xi) + dx * (1 - dy) * fetch(yi, xi + 1) +
                            (1 - dx) * dy * fetch(yi + 1, xi) + dx * dy * fetch(yi + 1, xi + 1))
    return out


def corr_index_backward(volume_shape, coords, corr_grad, r):
    """Adjoint of corr_index_forward with respect to the volume."""
    coords = np.asarray(coords, np.float64)
    g = np.asarray(corr_grad, np.float64)
    E, h1, w1, h2, w2 = volume_shape
    rd = 2 * r + 1
    out = np.zeros(volume_shape)
    x0 = np.floor(coords[:, 0]); y0 = np.floor(coords[:, 1])
    dx = coords[:, 0] - x0; dy = coords[:, 1] - y0
    e_idx, yy, xx = np.meshgrid(np.arange(E), np.arange(h1), np.arange(w1), indexing="ij")

    def push(yi, xi, val):
        ok = (yi >= 0) & (yi < h2) & (xi >= 0) & (xi < w2)
        np.add.at(out, (e_idx[ok], yy[ok], xx[ok], yi[ok].astype(int), xi[ok].astype(int)), val[ok])

    for i in range(rd):
        for j in range(rd):
            xi = x0 + (j - r); yi = y0 + (i - r)
            gg = g[:, i, j]
            push(yi, xi, (1 - dx) * (1 - dy) * gg); push(yi, xi + 1, dx * (1 - dy) * gg)
            push(yi + 1, xi, (1 - dx) * dy * gg); push(yi + 1, xi + 1, dx * dy * gg)
    return out
