"""Where the drop-in (autograd) view step spends its host time: cProfile over a few windows of
the bench's `autograd_api` loop (GaussianRasterizer through autograd, fused loss, PyTorch
bookkeeping, SparseGaussianAdam.step) at the bench size, next to the wall time per view."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vings_mono_amd import harness, synth  # noqa: E402

dev = torch.device("cuda:0")
N, W, H, V = 1_000_000, 1920, 1080, 8
if len(sys.argv) > 1:
    N, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
raw = synth.make_splats(N, W, H, n_views=V)
intr = synth.intrinsic_for(W, H)
mapper = harness.SurfelMapper(raw, dev)
c2w = synth.window_poses(V).to(dev)
gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
gts, w2cs, Rh = [], [], []
with torch.no_grad():
    for i in range(V):
        w = torch.linalg.inv(c2w[i])
        pr = harness.render(gtp, w, intr)
        gts.append({"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])})
        w2cs.append(w)
        Rh.append(w[:3, :3].cpu())


def window():
    for i in range(V):
        radii, cur = mapper.backward_view(w2cs[i], intr, gts[i], Rw2c_host=Rh[i])
        mapper.finish_view(radii, cur, i)


for _ in range(3):
    window()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    window()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / (5 * V)
print("autograd view step: %.3f ms per view = %.1f frames/s" % (dt * 1e3, 1 / dt))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    window()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
