"""Seeded synthetic inputs for the mapping hot path: pinhole cameras, splat
clouds and dense-BA graphs.  Everything is generated on the CPU with a seeded
``torch.Generator`` so every device and every rank sees identical bits.

Camera conventions restate /root/reference/scripts/gaussian/cameras.py:7-67
(``viewmatrix = w2c.T``, ``projmatrix = viewmatrix @ P.T``, znear 0.01,
zfar 100, principal point ignored by the rasteriser path).
"""
import math

import numpy as np
import torch

SEED_BASE = 20250124
ZNEAR, ZFAR = 0.01, 100.0


def focal_to_fov(focal, pixels):
    return 2.0 * math.atan(pixels / (2.0 * focal))


def projection_matrix(fov_x, fov_y, znear=ZNEAR, zfar=ZFAR):
    """4x4 perspective matrix, column-vector convention, w_clip = z_view."""
    tx, ty = math.tan(fov_x / 2.0), math.tan(fov_y / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    P[3, 2] = 1.0
    return P


_PROJ_T = {}


def _projection_t(fov_x, fov_y, device):
    """Transposed projection matrix on `device`, built once per (field of view, device): a
    keyframe window renders with one intrinsic, and a pageable host-to-device copy per render
    makes the host wait for everything queued on the stream before it."""
    key = (float(fov_x), float(fov_y), str(device))
    if key not in _PROJ_T:
        if len(_PROJ_T) > 64:
            _PROJ_T.clear()
        _PROJ_T[key] = projection_matrix(fov_x, fov_y).T.contiguous().to(device)
    return _PROJ_T[key]


def _camera_center(w2c):
    """inverse(w2c.T)[3, :3] (cameras.py:62-64) = -A^-1 t for w2c = [A t; 0 1], with the 3x3
    inverse written out through cross products.  On the GPU `torch.linalg.inv` makes the host
    wait for the whole stream (its LU reports a status word), once per render; these are a
    handful of asynchronous elementwise launches."""
    A, t = w2c[:3, :3], w2c[:3, 3]
    c0, c1, c2 = A[:, 0], A[:, 1], A[:, 2]
    r0, r1, r2 = torch.linalg.cross(c1, c2), torch.linalg.cross(c2, c0), torch.linalg.cross(c0, c1)
    det = (c0 * r0).sum()
    inv = torch.stack((r0, r1, r2)) / det
    return (-(inv @ t)).contiguous()


class Camera:
    """Matrices in the layout the rasteriser settings expect."""

    def __init__(self, w2c, intrinsic):
        # intrinsic: dict fu,fv,cu,cv,H,W with fu/cu vertical, fv/cv horizontal
        # (/root/reference/scripts/gaussian/tf.py:27-45).
        self.width, self.height = int(intrinsic["W"]), int(intrinsic["H"])
        fov_y = focal_to_fov(intrinsic["fu"], self.height)
        fov_x = focal_to_fov(intrinsic["fv"], self.width)
        self.world_view_transform = w2c.T.contiguous()
        self.projection_matrix = _projection_t(fov_x, fov_y, w2c.device)
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).contiguous()
        self.camera_center = _camera_center(w2c)
        self.tanfovx = math.tan(fov_x * 0.5)
        self.tanfovy = math.tan(fov_y * 0.5)


def intrinsic_for(W, H):
    f = 0.6 * W
    return {"fu": f, "fv": f, "cu": (H - 1) / 2.0, "cv": (W - 1) / 2.0, "H": H, "W": W}


def window_poses(n_views, x_extent=None):
    """c2w [n,4,4]: view k at x = 0.4 k metres, yaw 1 degree * k, looking down +z."""
    out = torch.zeros(n_views, 4, 4, dtype=torch.float32)
    for k in range(n_views):
        a = math.radians(1.0 * k)
        c, s = math.cos(a), math.sin(a)
        R = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float32)
        out[k, :3, :3] = R
        out[k, 0, 3] = 0.4 * k
        out[k, 3, 3] = 1.0
    return out


def make_splats(n, W, H, n_views=1, seed=SEED_BASE, z_range=(1.5, 60.0)):
    """Raw (pre-activation) splat parameters filling the union frustum of a window."""
    g = torch.Generator().manual_seed(seed)
    intr = intrinsic_for(W, H)
    fx = intr["fv"]
    c2w = window_poses(n_views)
    u = (torch.rand(n, generator=g) * 1.1 - 0.05) * W
    v = (torch.rand(n, generator=g) * 1.1 - 0.05) * H
    lz0, lz1 = math.log(z_range[0]), math.log(z_range[1])
    z = torch.exp(torch.rand(n, generator=g) * (lz1 - lz0) + lz0)
    k = torch.randint(0, n_views, (n,), generator=g)
    xc = torch.stack([(u - (W - 1) / 2.0) / fx * z, (v - (H - 1) / 2.0) / fx * z, z], 1)
    Rk, tk = c2w[k, :3, :3], c2w[k, :3, 3]
    xyz = torch.einsum("nij,nj->ni", Rk, xc) + tk
    r_px = torch.exp(torch.randn(n, 2, generator=g) * 0.6 + math.log(2.5))
    scaling = torch.log(z[:, None] * r_px / fx)
    rotation = torch.randn(n, 4, generator=g)
    opacity = torch.randn(n, 1, generator=g) * 1.5
    rgb = torch.rand(n, 3, generator=g)
    return {"_xyz": xyz.contiguous(), "_rgb": rgb.contiguous(), "_opacity": opacity.contiguous(),
            "_scaling": scaling.contiguous(), "_rotation": rotation.contiguous()}


def _quat_z_to(normal):
    """Unit quaternions (w,x,y,z) that turn the splat's local +z axis into `normal` [n,3]."""
    nx, ny, nz = normal.unbind(1)
    q = torch.stack([1.0 + nz, -ny, nx, torch.zeros_like(nx)], 1)
    flip = (1.0 + nz) < 1e-6                      # normal = -z: half a turn about x
    q[flip] = torch.tensor([0.0, 1.0, 0.0, 0.0])
    return torch.nn.functional.normalize(q)


def make_surface_splats(n, W, H, n_views=1, seed=SEED_BASE, z_far=80.0):
    """A street instead of a cloud: opaque splats lying ON surfaces (a ground plane 1.6 m under
    the cameras, two side walls, three facades across the street at 12 / 25 / 45 m), the way
    a converged map looks.  Opacity ~ sigmoid(N(2.2, 0.7)) ~ 0.9, so the lists of a tile end
    early on the first surface; the ground near the horizon and the walls far down the street
    pile many splats into few tiles, so list lengths spread by more than an order of magnitude
    and the longest exceed the 2048-entry LDS sort.  Same dict as `make_splats`."""
    g = torch.Generator().manual_seed(seed)
    intr = intrinsic_for(W, H)
    fx = intr["fv"]
    c2w = window_poses(n_views)
    kind = torch.rand(n, generator=g)
    k = torch.randint(0, n_views, (n,), generator=g)
    u = (torch.rand(n, generator=g) * 1.1 - 0.05) * W
    v = (torch.rand(n, generator=g) * 1.1 - 0.05) * H
    lz0, lz1 = math.log(1.5), math.log(z_far)
    z = torch.exp(torch.rand(n, generator=g) * (lz1 - lz0) + lz0)
    cam_h, wall_x, wall_top = 1.6, 6.0, -5.0
    xs = (u - (W - 1) / 2.0) / fx
    ys = (v - (H - 1) / 2.0) / fx
    ground = kind < 0.40
    wall = (kind >= 0.40) & (kind < 0.70)
    facade = kind >= 0.70
    xc = torch.zeros(n, 3)
    normal = torch.zeros(n, 3)
    # ground: the pixel column and the distance are drawn, the height is the plane's
    xc[ground] = torch.stack([xs * z, torch.full((n,), cam_h), z], 1)[ground]
    normal[ground] = torch.tensor([0.0, -1.0, 0.0])
    # walls: left / right, the height drawn between the ground and the eaves
    side = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)
    yw = wall_top + torch.rand(n, generator=g) * (cam_h - wall_top)
    xc[wall] = torch.stack([side * wall_x, yw, z], 1)[wall]
    normal[wall] = torch.stack([-side, torch.zeros(n), torch.zeros(n)], 1)[wall]
    # facades: fronto-parallel planes across part of the street
    fz = torch.tensor([12.0, 25.0, 45.0])[torch.randint(0, 3, (n,), generator=g)]
    fx0 = torch.where(fz == 12.0, -6.0, torch.where(fz == 25.0, -1.0, -6.0))
    fx1 = torch.where(fz == 12.0, -2.5, torch.where(fz == 25.0, 6.0, 6.0))
    xf = fx0 + torch.rand(n, generator=g) * (fx1 - fx0)
    xc[facade] = torch.stack([xf, yw, fz], 1)[facade]
    normal[facade] = torch.tensor([0.0, 0.0, -1.0])
    dist = xc[:, 2].clone()
    Rk, tk = c2w[k, :3, :3], c2w[k, :3, 3]
    xyz = torch.einsum("nij,nj->ni", Rk, xc) + tk
    normal = torch.einsum("nij,nj->ni", Rk, normal)
    r_px = torch.exp(torch.randn(n, 2, generator=g) * 0.6 + math.log(2.5))
    scaling = torch.log(dist[:, None] * r_px / fx)
    rotation = _quat_z_to(normal) + torch.randn(n, 4, generator=g) * 0.03
    opacity = torch.randn(n, 1, generator=g) * 0.7 + 2.2
    rgb = torch.rand(n, 3, generator=g)
    return {"_xyz": xyz.contiguous(), "_rgb": rgb.contiguous(), "_opacity": opacity.contiguous(),
            "_scaling": scaling.contiguous(), "_rotation": rotation.contiguous()}


def perturb_splats(params, seed=SEED_BASE + 1):
    """Copy used to render ground-truth images: xyz += N(0,0.01), rgb += N(0,0.05)."""
    g = torch.Generator().manual_seed(seed)
    out = {k: v.clone() for k, v in params.items()}
    out["_xyz"] += torch.randn(out["_xyz"].shape, generator=g) * 0.01
    out["_rgb"] = (out["_rgb"] + torch.randn(out["_rgb"].shape, generator=g) * 0.05).clamp(0, 1)
    return out


def activate(params):
    """Activations of /root/reference/scripts/gaussian/gaussian_base.py:44-48,64-72."""
    return dict(
        means3D=params["_xyz"],
        colors=params["_rgb"],
        opacities=torch.sigmoid(params["_opacity"]),
        scales=torch.exp(params["_scaling"]),
        rotations=torch.nn.functional.normalize(params["_rotation"]),
    )


# ---------------------------------------------------------------------------
# dense bundle adjustment graphs
# ---------------------------------------------------------------------------
def quat_from_yaw(a):
    # rotation about +y by angle a, (x,y,z,w)
    return np.array([0.0, math.sin(a / 2), 0.0, math.cos(a / 2)])


def make_dba_problem(n_frames=9, ht=43, wd=77, n_edges=48, seed=SEED_BASE + 7, buffer=None,
                     noise_px=0.5):
    """Poses (world->cam, tx ty tz qx qy qz qw), inverse depths, edges, flow targets.

    Frame 0 is the gauge frame (t0 = 1).  Edges are all ordered pairs with
    1 <= |i-j| <= 3, padded with the longest-baseline pairs up to n_edges.
    Targets are the exact reprojections plus N(0, noise_px) pixels.
    """
    from . import se3np
    g = np.random.default_rng(seed)
    B = buffer or n_frames
    poses = np.zeros((B, 7), np.float64)
    poses[:, 6] = 1.0
    for k in range(n_frames):
        a = math.radians(1.0 * k)
        q = quat_from_yaw(a)          # c2w rotation
        t = np.array([0.4 * k, 0.0, 0.0])
        # world->cam = inverse of (R,t)
        qi = q * np.array([-1, -1, -1, 1.0])
        ti = -se3np.quat_rotate(qi, t)
        poses[k, :3], poses[k, 3:] = ti, qi
    disps = np.zeros((B, ht, wd), np.float64)
    disps[:n_frames] = g.uniform(0.02, 0.8, size=(n_frames, ht, wd))
    fx = 0.6 * (wd * 8) / 8.0
    intr = np.array([fx, fx, (wd - 1) / 2.0, (ht - 1) / 2.0])
    pairs = [(i, j) for i in range(n_frames) for j in range(n_frames) if 1 <= abs(i - j) <= 3]
    extra = sorted([(i, j) for i in range(n_frames) for j in range(n_frames) if abs(i - j) > 3],
                   key=lambda p: (-abs(p[0] - p[1]), p))
    pairs = (pairs + extra)[:n_edges]
    ii = np.array([p[0] for p in pairs], np.int64)
    jj = np.array([p[1] for p in pairs], np.int64)
    coords, _ = se3np.reproject(poses, disps, intr, ii, jj)
    target = coords + g.normal(0, noise_px, size=coords.shape)       # [E,ht,wd,2]
    weight = g.uniform(0, 1, size=coords.shape)
    kx = np.unique(np.concatenate([np.arange(1, n_frames), ii]))
    eta = 1e-3 * g.uniform(0.5, 1.5, size=(kx.shape[0], ht, wd))
    return dict(poses=poses, disps=disps, intr=intr, disps_sens=np.zeros_like(disps),
                target=np.ascontiguousarray(target.transpose(0, 3, 1, 2)),
                weight=np.ascontiguousarray(weight.transpose(0, 3, 1, 2)),
                eta=eta, ii=ii, jj=jj, t0=1, t1=n_frames)
