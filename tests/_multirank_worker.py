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
    W, H, n, steps = 160, 96, 5000, 3
    raw = synth.make_splats(n, W, H, n_views=world * steps, seed=33)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(world * steps)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    from vings_mono_amd import maintenance
    a, b = harness.SurfelMapper(raw, dev), harness.SurfelMapper(raw, dev)
    # c = b, except that it merges its record keys over the ranks BEFORE the map maintenance
    # below; b carries its unmerged keys through the prune and the append
    c = harness.SurfelMapper(raw, dev)
    # d = b with the planar exchange (all N rows in one all-reduce) instead of the compact one
    d = harness.SurfelMapper(raw, dev)
    red = parallel.GradReducer(n, dev, world, n_chunks=3)
    red_c = parallel.GradReducer(n, dev, world)
    red_d = parallel.GradReducer(n, dev, world, compact=False)
    exchanged = []
    extra = {k: v[:300].clone() for k, v in synth.make_splats(600, W, H, n_views=2, seed=34).items()}
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
        exchanged.append(red.last_exchanged_rows)
        c.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k, red_c)
        d.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k, red_d)
        if s == 0:
            a.sync_records(red)                    # blocking form after the first step ...
            b.sync_records(red)
            c.sync_records(red_c)
            d.sync_records(red_d)
        if s == 1:
            # map maintenance between two view steps of a multi-rank job: every rank prunes the
            # same splats and appends the same new ones (the splat count changes twice)
            c.sync_records(red_c)
            for m in (a, b, c, d):
                mask = (torch.arange(m.n, device=dev) % 9) == 4
                maintenance.prune(m, mask)
                maintenance.add_splats(m, extra)
                assert m.n == n - int(mask.sum()) + 300
    for m, r in ((a, red), (b, red), (c, red_c), (d, red_d)):  # ... and the overlapped form at the end
        fin = m.sync_records(r, async_op=True)
        fin()
    # Checkpoints of a multi-rank job.  The records were merged just above, so one rank alone
    # may save without a collective; after one more view step the keys are unmerged again: a
    # save without the reducer is refused, with it (all ranks call) it merges first.
    from vings_mono_amd import io_formats
    if rank == 0:
        io_formats.save_pt_ckpt(b, out_path + ".merged.ckpt")
    b.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k, red)
    a.finish_view(*a.backward_view(w2c, intr, gt), k, red)       # (the other mappers take the step too)
    c.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k, red_c)
    d.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k, red_d)
    try:
        io_formats.save_pt_ckpt(b, out_path + ".refused.ckpt")
        refused = False
    except ValueError:
        refused = True
    io_formats.save_pt_ckpt(b, out_path + f".rank{rank}.ckpt", red)
    a.sync_records(red)
    c.sync_records(red_c)
    d.sync_records(red_d)
    # GPU side of a loop closure over 7 keyframes (not a multiple of 2 or 4: the last rank's
    # block is short or empty): scoring and tracker refresh dealt out over the ranks against the
    # same calls on this rank alone
    Kl = 7
    kf_c2w = [synth.window_poses(Kl)[i].to(dev) for i in range(Kl)]
    s_one, i_one = maintenance.calc_score(b, kf_c2w, list(range(20, 20 + Kl)), intr)
    s_all, i_all = maintenance.calc_score(b, kf_c2w, list(range(20, 20 + Kl)), intr, red)
    maps = []
    for r_ in (None, red):
        dense = torch.full((Kl, H, W), 0.5, device=dev)
        coarse = torch.zeros((Kl, H // 8, W // 8), device=dev)
        maintenance.rectify_tracker(b, kf_c2w, intr, dense, coarse, r_)
        maps.append((dense, coarse))
    torch.cuda.synchronize()
    state = {"rank": rank, "exchanged_rows": exchanged, "unmerged_save_refused": refused,
             "loop": {"scores_err": float(((s_one - s_all).abs() / (s_one.abs() + 1e-3)).max()),
                      "ids_match": float((i_one == i_all).float().mean()),
                      "dense_equal": bool(torch.equal(maps[0][0], maps[1][0])),
                      "coarse_equal": bool(torch.equal(maps[0][1], maps[1][1])),
                      "scored": int((s_all[:, 0] > 0).sum()), "kf_seen": int(len(torch.unique(i_all[:, 0]))),
                      "refreshed": int((maps[1][0] != 0.5).sum())}}
    for name, m in (("autograd", a), ("fused", b), ("fused_synced_first", c), ("fused_planar", d)):
        state[name] = {k: m.params[k].detach().cpu() for k in m.NAMES}
        state[name]["global_scores"] = m.global_scores.cpu()
        state[name]["local_scores"] = m.local_scores.cpu()
        state[name]["kf_id"] = m.globalkf_id.cpu()
        state[name]["kf_max"] = m.globalkf_max_scores.cpu()
    keep = (torch.arange(n) % 9) != 4
    state["raw"] = {k: torch.cat((raw[k][keep], extra[k])) for k in a.NAMES}
    torch.save(state, f"{out_path}.rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
