"""Small numpy SE3 helpers used to synthesise dense-BA problems (host side only).

Poses are (tx,ty,tz,qx,qy,qz,qw), world->camera, as in
/root/reference/scripts/frontend/depth_video.py:65-86.
"""
import numpy as np

MIN_DEPTH = 0.2


def quat_rotate(q, v):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    u = np.stack([x, y, z], -1)
    t = 2.0 * np.cross(u, v)
    return v + w[..., None] * t + np.cross(u, t)


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def relative(poses, ii, jj):
    """G_ij = G_j * G_i^-1 as (t,q); a self edge becomes the fixed stereo baseline."""
    ti, qi = poses[ii, :3], poses[ii, 3:]
    tj, qj = poses[jj, :3], poses[jj, 3:]
    qi_inv = qi * np.array([-1.0, -1.0, -1.0, 1.0])
    q = quat_mul(qj, qi_inv)
    t = tj - quat_rotate(q, ti)
    same = np.asarray(ii) == np.asarray(jj)
    t[same] = np.array([-0.1, 0.0, 0.0])
    q[same] = np.array([0.0, 0.0, 0.0, 1.0])
    return t, q


def reproject(poses, disps, intr, ii, jj):
    """Pixel coordinates in frame j of the grid of frame i: [E,h,w,2], valid[E,h,w]."""
    fx, fy, cx, cy = (float(v) for v in intr)
    ht, wd = disps.shape[1:]
    y, x = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64),
                       indexing="ij")
    d = disps[ii]
    X0 = np.stack([np.broadcast_to((x - cx) / fx, d.shape),
                   np.broadcast_to((y - cy) / fy, d.shape), np.ones_like(d)], -1)
    t, q = relative(poses, ii, jj)
    X1 = quat_rotate(q[:, None, None, :], X0) + t[:, None, None, :] * d[..., None]
    Z = X1[..., 2]
    Zc = np.where(Z < 0.5 * MIN_DEPTH, 1.0, Z)
    coords = np.stack([fx * X1[..., 0] / Zc + cx, fy * X1[..., 1] / Zc + cy], -1)
    return coords, Z > MIN_DEPTH
