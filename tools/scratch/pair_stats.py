"""How many (pixel group, splat) pairs does one view of the bench scene have, for groups of
8x8, 8x4 (rows halves), 4x8 (column halves) and 4x4 pixels?  Exact per-pixel alpha >= 1/255 test
(saturation ignored)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vings_mono_amd import harness, synth, surfel
dev = torch.device("cuda:0")
W, H, n = 1920, 1080, 1_000_000
raw = synth.make_splats(n, W, H, n_views=8, seed=synth.SEED_BASE)
intr = synth.intrinsic_for(W, H)
w2c = torch.linalg.inv(synth.window_poses(8)[3]).to(dev)
P = {k: v.to(dev) for k, v in raw.items()}
act = synth.activate(P)
st, aux = surfel.rasterize_forward(P["_xyz"].contiguous(), P["_rgb"].contiguous(), act["opacities"].contiguous(),
                                   act["scales"].contiguous(), act["rotations"].contiguous(),
                                   harness.view_settings(w2c, intr), keep_keys=True)
D = st.D
keys = st.keys[:D]
tile = (keys >> 32).long()
vals = st.vals[:D].long()
gx = (W + 15) // 16
geom = st.geom
tot = {"8x8": 0, "8x4": 0, "4x8": 0, "4x4": 0, "pixels": 0, "inst": 0}
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
step = 200_000
sel = torch.randperm(D, device=dev)[:1_000_000]          # a sample of the instances
for c0 in range(0, sel.numel(), step):
    idx = sel[c0:c0 + step]
    g = geom[vals[idx]]                                   # [m,24]
    t = tile[idx]
    fx = ((t % gx) * 16)[:, None, None] + xx[None]        # [m,16,16]
    fy = ((t // gx) * 16)[:, None, None] + yy[None]
    fx, fy = fx.float(), fy.float()
    Tu, Tv, Tw = g[:, 0:3], g[:, 3:6], g[:, 6:9]
    e = lambda v, k: v[:, k][:, None, None]
    kx, ky, kz = fx * e(Tw, 0) - e(Tu, 0), fx * e(Tw, 1) - e(Tu, 1), fx * e(Tw, 2) - e(Tu, 2)
    lx, ly, lz = fy * e(Tw, 0) - e(Tv, 0), fy * e(Tw, 1) - e(Tv, 1), fy * e(Tw, 2) - e(Tv, 2)
    px, py, pz = ky * lz - kz * ly, kz * lx - kx * lz, kx * ly - ky * lx
    sx, sy = px / pz, py / pz
    rho3 = sx * sx + sy * sy
    dx, dy = g[:, 9][:, None, None] - fx, g[:, 10][:, None, None] - fy
    rho2 = 2.0 * (dx * dx + dy * dy)
    rho = torch.minimum(rho3, rho2)
    depth = torch.where(rho3 <= rho2, sx * e(Tw, 0) + sy * e(Tw, 1) + e(Tw, 2), e(Tw, 2).expand_as(sx))
    alpha = torch.clamp(g[:, 14][:, None, None] * torch.exp(-0.5 * rho), max=0.99)
    hit = (pz != 0) & ~(depth < 0.2) & (alpha >= 1.0 / 255.0) & (fx < W) & (fy < H)
    tot["pixels"] += int(hit.sum())
    tot["inst"] += int(hit.flatten(1).any(1).sum())
    for name, (bh, bw) in {"8x8": (8, 8), "8x4": (4, 8), "4x8": (8, 4), "4x4": (4, 4)}.items():
        m = hit.reshape(-1, 16 // bh, bh, 16 // bw, bw).any(4).any(2)
        tot[name] += int(m.sum())
print("instances sampled", sel.numel(), "of", D, "with any hit", tot["inst"])
for k in ("8x8", "8x4", "4x8", "4x4"):
    print(k, "pairs per instance %.3f" % (tot[k] / sel.numel()), " pixels per pair %.2f" % (tot["pixels"] / tot[k]))
print("8x4 pairs per 8x8 pair %.3f ; 4x8 per 8x8 %.3f ; 4x4 per 8x8 %.3f" % (tot["8x4"] / tot["8x8"], tot["4x8"] / tot["8x8"], tot["4x4"] / tot["8x8"]))
