import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vings_mono_amd import harness, synth, surfel
dev = torch.device("cuda:0")
W, H, n = 1920, 1080, 1_000_000
raw = synth.make_splats(n, W, H, n_views=8, seed=1)
intr = synth.intrinsic_for(W, H)
w2c = torch.linalg.inv(synth.window_poses(8)[3]).to(dev)
P = {k: v.to(dev) for k, v in raw.items()}
act = synth.activate(P)
st, aux = surfel.rasterize_forward(P["_xyz"].contiguous(), P["_rgb"].contiguous(), act["opacities"].contiguous(),
                                   act["scales"].contiguous(), act["rotations"].contiguous(),
                                   harness.view_settings(w2c, intr))
r = st.ranges.long()
ln = (r[:, 1] - r[:, 0]).float()
q = torch.tensor([0.1, 0.5, 0.9, 0.99, 0.999, 1.0], device=dev)
print("D", st.D, "tiles", ln.numel(), "mean", float(ln.mean()), "quantiles", torch.quantile(ln, q).tolist())
for cap in (64, 128, 256, 512, 1024, 2048, 4096):
    print(cap, int((ln <= cap).sum()))
