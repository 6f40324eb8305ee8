"""Worker of tests/test_multirank_gpu.py::test_exchange_over_rccl_one_rank: the multi-GPU view
step with its collectives going through RCCL.  The test box has one GPU and RCCL takes one rank
per GPU, so the communicator has ONE rank: every all-reduce is the identity, but it is issued
the way a multi-GPU job issues it (uint8 MAX on a side stream, float SUM per chunk as
asynchronous work, int64 MAX of the record keys), and the exchange code around it (touched mask,
slot scan, chunked pack, per-chunk scores + Adam) is the real one.  The result must equal the
single-GPU step."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_path = sys.argv[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from vings_mono_amd import harness, parallel, synth
    W, H, n, steps = 160, 96, 5000, 3
    raw = synth.make_splats(n, W, H, n_views=steps, seed=33)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(steps)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    single = harness.SurfelMapper(raw, dev)
    mappers = {"autograd": harness.SurfelMapper(raw, dev), "fused": harness.SurfelMapper(raw, dev),
               "fused_planar": harness.SurfelMapper(raw, dev)}
    # the reducers are told of two ranks so that the exchange runs; the sums are over the one
    # rank the communicator has
    reds = {"autograd": parallel.GradReducer(n, dev, 2, compact=False),
            "fused": parallel.GradReducer(n, dev, 2, n_chunks=3),
            "fused_planar": parallel.GradReducer(n, dev, 2, compact=False)}
    exchanged = []
    for k in range(steps):
        w2c = torch.linalg.inv(c2w[k]).to(dev)
        with torch.no_grad():
            pr = harness.render(gtp, w2c, intr)
            gt = {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
                  "depth_cov": torch.ones_like(pr["depth"])}
        cam, R = harness.view_settings(w2c, intr), w2c[:3, :3].cpu()
        single.fused_view(cam, intr, dict(gt), R, k, None)
        radii, cur = mappers["autograd"].backward_view(w2c, intr, dict(gt))
        mappers["autograd"].finish_view(radii, cur, k, reds["autograd"])
        for name in ("fused", "fused_planar"):
            mappers[name].fused_view(cam, intr, dict(gt), R, k, reds[name])
        exchanged.append(reds["fused"].last_exchanged_rows)
        if k == 0:
            for name, m in mappers.items():
                m.sync_records(reds[name])
    for name, m in mappers.items():
        fin = m.sync_records(reds[name], async_op=True)
        fin()
    torch.cuda.synchronize()
    state = {"exchanged_rows": exchanged, "backend": dist.get_backend()}
    for name, m in [("single", single)] + list(mappers.items()):
        state[name] = {k: m.params[k].detach().cpu() for k in m.NAMES}
        state[name]["global_scores"] = m.global_scores.cpu()
        state[name]["local_scores"] = m.local_scores.cpu()
        state[name]["kf_id"] = m.globalkf_id.cpu()
        state[name]["kf_max"] = m.globalkf_max_scores.cpu()
    state["raw"] = {k: raw[k].clone() for k in single.NAMES}
    torch.save(state, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
