"""EXPERIMENT: does handing the blend backward its tiles longest-list-first shorten the launch?"""
import ctypes, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vings_mono_amd import _lib, harness, surfel, synth
from vings_mono_amd.harness import mapping_loss_hip

dev = torch.device("cuda:0")
W, H, N = 1920, 1080, 1_000_000
raw = synth.make_splats(N, W, H, n_views=8, seed=1)
intr = synth.intrinsic_for(W, H)
w2c = torch.linalg.inv(synth.window_poses(8)[3]).to(dev)
gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
with torch.no_grad():
    pr = harness.render(gtp, w2c, intr)
gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(), "depth_cov": torch.ones_like(pr["depth"])}
m = harness.SurfelMapper(raw, dev)
fb = m._fused_buffers()
P = m.params
settings = harness.view_settings(w2c, intr)
L = _lib.lib()
raw_lib = ctypes.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else L
fn = raw_lib.vings_experiment_tile_order
fn.argtypes = [ctypes.c_void_p]; fn.restype = None
with torch.no_grad():
    dgeom = torch.empty((N, surfel.GRAD_STRIDE), dtype=torch.float32, device=dev)
    st, aux = surfel.rasterize_forward(P["_xyz"].detach(), P["_rgb"].detach(), fb["opac"], fb["scales"], fb["rot"],
                                       settings, raw=(P["_opacity"].detach(), P["_scaling"].detach(), P["_rotation"].detach()),
                                       dgeom=dgeom, lean=True)
    loss_vec, d_rgb, d_allmap = mapping_loss_hip(st.color, st.allmap, gt, w2c[:3, :3].cpu(), intr, m.loss_weights, cache=True)
    lens = (st.ranges[:, 1] - st.ranges[:, 0]).to(torch.int64)
    T = lens.numel()
    lpt = torch.argsort(lens, descending=True).to(torch.int32).contiguous()
    ident = torch.arange(T, dtype=torch.int32, device=dev)
    def run(order, reps=20):
        fn(None if order is None else order.data_ptr())
        surfel.blend_backward(st, aux, d_rgb, d_allmap, dgeom=dgeom, lean=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            surfel.blend_backward(st, aux, d_rgb, d_allmap, dgeom=dgeom, lean=True)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    res = {"tiles": T, "len_mean": float(lens.float().mean()), "len_max": int(lens.max()),
           "len_p99": float(torch.quantile(lens.float(), 0.99))}
    for rep in range(2):
        res[f"xcd_contiguous_{rep}"] = run(None)
        res[f"identity_{rep}"] = run(ident)
        res[f"longest_first_{rep}"] = run(lpt)
    fn(None)
print(json.dumps(res))
