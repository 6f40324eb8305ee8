"""Worker of tests/test_multirank_gpu.py: one rank of a 2-rank mapping job.  Both ranks share
the single GPU of the test box and exchange over gloo (RCCL needs one GPU per rank); the
exchange code path (vings_mono_amd.parallel.GradReducer) is the one bench.py uses."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    from vings_mono_amd import harness, parallel, synth
    dev = torch.device("cuda:0")
    W, H, n, steps = 160, 96, 5000, 2
    raw = synth.make_splats(n, W, H, n_views=world * steps, seed=33)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(world * steps)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    red = parallel.GradReducer(n, dev, world)
    for s in range(steps):
        k = s * world + rank                      # this rank's keyframe of the step
        w2c = torch.linalg.inv(c2w[k]).to(dev)
        with torch.no_grad():
            pr = harness.render(gtp, w2c, intr)
            gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
                  "depth_cov": torch.ones_like(pr["depth"])}
        radii, cur = a.backward_view(w2c, intr, gt)
        a.finish_view(radii, cur, k, red)
        b.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k, red)
        if s == 0:
            a.sync_records(red)                    # blocking form after the first step ...
            b.sync_records(red)
    for m in (a, b):                               # ... and the overlapped form at the end
        fin = m.sync_records(red, async_op=True)
        fin()
    torch.cuda.synchronize()
    state = {"rank": rank}
    for name, m in (("autograd", a), ("fused", b)):
        state[name] = {k: m.params[k].detach().cpu() for k in m.NAMES}
        state[name]["global_scores"] = m.global_scores.cpu()
        state[name]["local_scores"] = m.local_scores.cpu()
        state[name]["kf_id"] = m.globalkf_id.cpu()
        state[name]["kf_max"] = m.globalkf_max_scores.cpu()
    state["raw"] = {k: raw[k].clone() for k in a.NAMES}
    torch.save(state, f"{out_path}.rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
