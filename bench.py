#!/usr/bin/env python3
"""Mapping throughput benchmark for the MI355X hot path.

Metric (BASELINE.json): mapping frames/sec (render + backward + BA) at 1.0 M
splats, 1920x1080.  One *step* is one pass over an 8-keyframe window on every
GPU: per view  render -> mapping loss -> rasteriser backward -> score
bookkeeping -> (N>1: RCCL all-reduce of the packed splat gradients) -> sparse
Adam, plus one dense bundle-adjustment call ``ba(itrs=2)`` per window.
``value`` = views processed by all ranks / wall time of the timed steps.

    python bench.py --gpus N --steps K --warmup W

For N>1 there is one rank per GPU.  Started under torch.distributed.run (WORLD_SIZE
set) this process is one of the ranks; started plainly with --gpus N > 1 it
starts the N ranks itself as children (before anything here touches a GPU) and
hands on rank 0's line and the children's exit code.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
F64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix rate, AMD's data-sheet figure (256 CUs x 4 SIMDs x
                               # 32 flop/clk x 2.4 GHz; v_mfma_f64_16x16x4_f64 = 2048 flop in 64 cycles)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", choices=("c3", "c4", "c5", "native"), default="c3",
                    help="BASELINE config: c3 (default at every N: the same scene and the same work "
                         "per GPU, so the 1/2/4/8 figures form one weak-scaling curve) = 8-keyframe "
                         "window + DBA on every GPU, the one the metric is quoted on; c4 = one "
                         "32-keyframe window whose keyframes are dealt out over the GPUs (4 per GPU "
                         "at 8); c5 = 5.0M splats, 8 keyframes dealt out over the GPUs + DBA + the "
                         "GPU side of one loop closure over 64 history keyframes per step, its "
                         "keyframes dealt out over the GPUs too; native = the reference's own "
                         "operating point (344x616, 300k splats, 43x77 DBA grid)")
    ap.add_argument("--scene", choices=("frustum", "surfaces"), default="frustum",
                    help="frustum: splats spread through the window's frustum union (the metric's "
                         "scene); surfaces: opaque walls and a ground plane with a 10x spread of "
                         "list lengths (a side workload)")
    ap.add_argument("--splats", type=int, default=None,
                    help="default 1.0M (c5: 5.0M, native: 0.3M)")
    ap.add_argument("--loop-keyframes", type=int, default=64, help="c5: history keyframes K")
    ap.add_argument("--width", type=int, default=None, help="default 1920 (native: 616)")
    ap.add_argument("--height", type=int, default=None, help="default 1080 (native: 344)")
    ap.add_argument("--views", type=int, default=None,
                    help="keyframes per GPU per step (default: 8 for c3, 32 / GPUs for c4)")
    ap.add_argument("--binning", choices=("tight", "reference"), default="tight",
                    help="tight: a splat enters only the tiles its alpha >= 1/255 footprint "
                         "reaches (same rendered bits, shorter instance list); reference: the "
                         "3-sigma square of the reference's rasteriser (bit-exact instance list)")
    ap.add_argument("--stable-frac", type=float, default=0.0,
                    help="put this random share of the splats under the stable mask for the whole "
                         "run (a side workload: the metric's map has none)")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="skip the reference_api / reference_binning figures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dba", action="store_true")
    ap.add_argument("--dba-stream", choices=("side", "main"), default="side",
                    help="side: the window's bundle adjustment runs on a second stream beside its "
                         "view steps (they share no data; in the reference the tracker's BA and the "
                         "mapper are separate workers too); main: after them on the same stream")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="all ranks share GPU 0 and exchange over gloo: checks the multi-rank "
                         "path on a one-GPU box; the figure printed is not a benchmark result")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="one rank, one GPU, a one-rank RCCL communicator, the multi-GPU view step "
                         "with every collective issued (each is the identity): what the exchange "
                         "costs before any byte crosses a link; not a benchmark result")
    ap.add_argument("--exchange", choices=("compact", "planar"), default="compact",
                    help="N > 1: compact = only the rows some rank touched travel, in chunks "
                         "applied while the next one is reduced; planar = all rows in one all-reduce")
    ap.add_argument("--exchange-chunks", type=int, default=2,
                    help="N > 1, compact exchange: chunks per view (one-rank RCCL rehearsal: each "
                         "chunk costs ~18 us of stream hand-overs, and larger messages use the links better)")
    ap.add_argument("--no-graph-run", action="store_true",
                    help="skip the graph_replay side figure (N = 1: the same step with every view "
                         "replayed from a HIP graph captured per keyframe)")
    ap.add_argument("--autograd", action="store_true",
                    help="drive the view step through torch autograd and PyTorch bookkeeping "
                         "instead of the fused C-ABI sequence")
    ap.add_argument("--torch-loss", action="store_true",
                    help="build the PyTorch loss graph instead of the fused HIP loss kernels")
    ap.add_argument("--dba-grid", type=str, default=None, help="default 135x240 (native: 43x77)")
    return ap.parse_args()


def algorithmic_bytes_blend_bwd(D, P, n_vis, C=9):
    # per launch of the blend-backward kernel: staged records 64 B per instance,
    # upstream gradients 4(3+C) B and saved forward state 4(C+3) B per pixel,
    # packed gradient rows 84 B per visible splat
    return 64 * D + 4 * (3 + C) * P + 4 * (C + 3) * P + 84 * n_vis


def measured_traffic(splats, W, H):
    """HBM bytes per blend-backward launch from the committed PMC passes of this same
    workload (rocprofv3 cannot run inside the bench); None for any other workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "blend_bwd_hbm_traffic.json")),
                       reverse=True):           # the latest round's passes first
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        wl = rec.get("workload", {})
        if (wl.get("splats"), wl.get("width"), wl.get("height")) == (splats, W, H):
            return rec.get("hbm_bytes_per_launch")
    return None


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as
    children of a process that never touches a GPU (one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1), pass rank 0's JSON line on and exit with
    the children's code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode == 0 and not lines:
        print("bench.py: the ranks exited cleanly without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.rehearse_rccl:
        raise SystemExit(self_launch(args))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU path")
    if args.rehearse_rccl:
        assert world == 1, "--rehearse-rccl is the one-rank form: start it without a launcher"
    if args.rehearse_gloo:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)      # a stuck rehearsal names its line
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    json_out = sys.stdout
    if world > 1 or args.rehearse_rccl:
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is for
        # the one JSON line, so everything else of this process goes to stderr
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    ranks_seen, comm = 1, None
    if args.rehearse_rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 or args.rehearse_rccl:
        # every rank adds a one: the line says how many ranks really took part
        ones = torch.ones(1, dtype=torch.int64, device="cpu" if args.rehearse_gloo else dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        comm = {"backend": dist.get_backend(), "ranks_seen": ranks_seen}
        if dist.get_backend() == "nccl":
            comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())

    from vings_mono_amd import harness, synth, parallel, profiling, surfel
    workload = args.workload
    native = workload == "native"
    if native:
        # The reference's own operating point: the mapper works at the tracker's image size
        # (frontend.image_size 344x616, /root/reference/configs/hierarchical/smallcity.yaml:42)
        # on a map that grows by <= 40k splats per keyframe, and the tracker's BA runs on the
        # 43x77 grid.  Same step as c3 otherwise.
        workload = "c3"
    W = args.width or (616 if native else 1920)
    H = args.height or (344 if native else 1080)
    args.width, args.height = W, H
    if args.splats is None:
        args.splats = 5_000_000 if workload == "c5" else 300_000 if native else 1_000_000
    if args.dba_grid is None:
        args.dba_grid = "43x77" if native else "135x240"
    make_scene = synth.make_surface_splats if args.scene == "surfaces" else synth.make_splats
    intr = synth.intrinsic_for(W, H)
    surfel.set_tight_binning(args.binning == "tight")
    loop = None
    if workload == "c5":
        # BASELINE config 5: a 5.0M-splat map along a street of K history keyframes; the 8 most
        # recent keyframes are the mapping window, dealt out over the GPUs (1 per GPU at 8).
        # Every step also runs the GPU side of one loop closure over all K keyframes
        # (loop_model.py:169-214, loop_rectify.py:300-437): the K scoring passes and the K depth
        # renders are dealt out over the ranks (records merged / maps gathered), the per-splat
        # re-anchor runs on every replica, with an identity correction so that the map of the
        # next step is the same.  The pose-graph solve itself (scipy / GTSAM on the host) is
        # not part of this path.
        K = args.loop_keyframes
        window = 8
        V = args.views or max(1, window // world)
        raw = synth.make_splats(args.splats, W, H, n_views=K, seed=5, z_range=(2.0, 120.0))
        c2w_all = synth.window_poses(K)
        c2w_rank = c2w_all[K - window + (rank * V) % window:][:V]
        assert c2w_rank.shape[0] == V, "views per GPU x GPUs must fit the 8-keyframe window"
        loop = {"K": K, "c2w": [p.clone() for p in c2w_all]}
    elif workload == "c3":
        # Identical splat replica on every rank, filling the frustum union of ONE V-keyframe
        # window.  Every rank films that window from its own height (5 cm apart), so all ranks
        # see the same share of the map: the work per GPU stays what it is at N = 1 (weak
        # scaling) instead of thinning out as the window would stretch over more keyframes.
        V = args.views or 8
        window = V
        raw = make_scene(args.splats, W, H, n_views=V, seed=synth.SEED_BASE)
        c2w_rank = synth.window_poses(V)
        c2w_rank[:, 1, 3] += 0.05 * rank
    else:
        # BASELINE config 4: one 32-keyframe window over the whole job, rank r owns keyframes
        # r*V .. r*V+V-1 of it (4 distinct keyframes per GPU at 8 GPUs); the map fills the
        # frustum union of all 32.
        window = 32
        V = args.views or max(1, window // world)
        raw = make_scene(args.splats, W, H, n_views=window, seed=synth.SEED_BASE)
        c2w_rank = synth.window_poses(window)[(rank * V) % window:][:V]
        assert c2w_rank.shape[0] == V, "views per GPU x GPUs must fit the 32-keyframe window"
    n_views_total = V * world
    mapper = harness.SurfelMapper(raw, dev)
    if args.stable_frac > 0:
        mapper.stable_mask[:] = torch.rand(mapper.n, generator=torch.Generator().manual_seed(9)).to(dev) < args.stable_frac
        mapper.note_stable_mask()
    gt_mapper_params = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    c2w_rank = c2w_rank.to(dev)
    my_views = list(range(rank * V, (rank + 1) * V))      # global keyframe ids
    gts = []
    with torch.no_grad():
        for i in range(V):
            pred = harness.render(gt_mapper_params, torch.linalg.inv(c2w_rank[i]), intr)
            depth = pred["depth"].clone()
            gts.append({"rgb": pred["rgb"].clone(), "depth": depth,
                        "depth_cov": torch.ones_like(depth)})
    del gt_mapper_params
    # (the one-rank RCCL rehearsal tells the reducer of two ranks so that the exchange runs)
    reducer = parallel.GradReducer(mapper.n, dev, 2 if args.rehearse_rccl else world,
                                   n_chunks=args.exchange_chunks,
                                   compact=args.exchange == "compact")
    if loop is not None:
        from vings_mono_amd import maintenance
        loop["c2w"] = [p.to(dev) for p in loop["c2w"]]
        loop["stack"] = torch.stack(loop["c2w"])
        loop["dense"] = torch.zeros((loop["K"], H, W), device=dev)
        loop["coarse"] = torch.zeros((loop["K"], H // 8, W // 8), device=dev)
        loop["ms"] = {}

    def loop_closure():
        """GPU side of one loop closure: top-3 keyframe scoring of every splat, re-anchoring
        (identity correction: the kernels run, the map stays), tracker depth refresh."""
        K = loop["K"]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        scores3, ids3 = maintenance.calc_score(mapper, loop["c2w"], list(range(K)), intr, reducer)
        ev[1].record()
        maintenance.rectify_gaussians(mapper, loop["stack"], loop["stack"], torch.ones(K), 0, K - 1,
                                      scores3, ids3, delete_below=-1.0)
        ev[2].record()
        maintenance.rectify_tracker(mapper, loop["c2w"], intr, loop["dense"], loop["coarse"], reducer)
        ev[3].record()
        loop["events"] = ev

    dba = None
    if not args.no_dba:
        from vings_mono_amd import dba as dba_mod
        ht, wd = (int(x) for x in args.dba_grid.split("x"))
        dba = dba_mod.make_bench_problem(dev, ht, wd)

    # keyframe poses are inputs of the mapper: their inverses (device) and the rotation block
    # the loss kernels take from the host are prepared once, like the ground-truth maps
    w2c_dev = [torch.linalg.inv(c2w_rank[i]) for i in range(V)]
    R_host = [w[:3, :3].cpu() for w in w2c_dev]
    cams = [harness.view_settings(w, intr) for w in w2c_dev]

    dba_side = torch.cuda.Stream(device=dev) if (dba is not None and args.dba_stream == "side") else None
    dba_done = torch.cuda.Event() if dba_side is not None else None

    def step(autograd=args.autograd, torch_loss=args.torch_loss, graphed=False, full_passes=False):
        if dba_side is not None:
            # the window's BA beside its view steps; it starts behind everything already queued
            dba_side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(dba_side):
                dba.run()
                dba_done.record(dba_side)
        for i, k in enumerate(my_views):
            if graphed:
                mapper.fused_view_graphed(k, cams[i], intr, gts[i], R_host[i], k)
            elif autograd or torch_loss:
                # the reference's formulation: PyTorch activations / autograd / bookkeeping
                # around the rasteriser (and, with --torch-loss, the PyTorch loss graph)
                radii, cur = mapper.backward_view(w2c_dev[i], intr, gts[i], Rw2c_host=R_host[i],
                                                  torch_loss=torch_loss)
                mapper.finish_view(radii, cur, k, reducer)
            else:
                mapper.fused_view(cams[i], intr, gts[i], R_host[i], k, reducer,
                                  lean=False if full_passes else None)
        # the record maxima of the window (best keyframe, error score) are merged over the ranks
        # here, once per window, behind the bundle adjustment
        finish_records = mapper.sync_records(reducer, async_op=True)
        if dba is not None and dba_side is None:
            dba.run()
        if finish_records is not None:
            finish_records()
        if loop is not None:
            loop_closure()
        if dba_side is not None:
            torch.cuda.current_stream(dev).wait_event(dba_done)   # the step ends when both have

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # inside the timed region only the dominant kernel is bracketed by events (two records
    # per launch); the per-stage table comes from one more, untimed, step afterwards
    profiling.select(["blend_bwd"])
    profiling.enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = profiling.read()
    profiling.select(None)
    profiling.enable(True)
    step()
    barrier()
    prof_all = profiling.read()
    profiling.enable(False)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = n_views_total * args.steps
    value = frames / elapsed
    stats = mapper.last_stats()

    def side_run(n_steps=2, **kw):
        """frames/s of a variant of the step, a few steps inside this same invocation."""
        step(**kw)
        barrier()
        t = time.perf_counter()
        for _ in range(n_steps):
            step(**kw)
        barrier()
        return round(n_views_total * n_steps / (time.perf_counter() - t), 3)

    reference_api = reference_binning = graph_replay = autograd_api = full_passes = stable_half = None
    if args.rehearse_rccl:
        args.no_side_runs = True
    if world == 1 and not args.no_side_runs and not args.no_graph_run and loop is None \
            and not (args.autograd or args.torch_loss):
        step(graphed=True)          # eager: learns the instance counts
        step(graphed=True)          # captures one graph per keyframe
        graph_replay = side_run(n_steps=min(args.steps, 10), graphed=True)
    if world == 1 and not args.no_side_runs and not (args.autograd or args.torch_loss) and loop is None \
            and args.stable_frac == 0:
        # the reference's own call pattern (gaussian_base.py:367-403): PyTorch activations,
        # GaussianRasterizer through autograd, the PyTorch loss graph, loss.backward(), PyTorch
        # score bookkeeping, SparseGaussianAdam.step -- through the drop-in names
        reference_api = side_run(autograd=True, torch_loss=True)
        # the same loop with one call changed: the loss through harness.mapping_loss_fused (the
        # fused HIP kernels behind a differentiable scalar) instead of the PyTorch loss graph
        autograd_api = side_run(autograd=True, torch_loss=False)
        full_passes = side_run(full_passes=True)
        # half of the map under the stable mask (no optimiser step for those splats,
        # gaussian_base.py:400-402): the backward pass skips their gradients
        mapper.stable_mask[:] = torch.rand(mapper.n, generator=torch.Generator().manual_seed(9)).to(dev) < 0.5
        mapper.note_stable_mask()
        stable_half = {"value": side_run()}
        profiling.select(["blend_bwd"])
        profiling.enable(True)
        step()
        barrier()
        pr = profiling.read().get("blend_bwd")
        profiling.select(None)
        profiling.enable(False)
        if pr and pr["count"]:
            stable_half["blend_bwd_ms"] = round(pr["ms"] / pr["count"], 4)
        mapper.stable_mask.zero_()
        mapper.note_stable_mask()
        if args.binning == "tight":
            surfel.set_tight_binning(False)
            reference_binning = side_run()
            reference_binning_D = mapper.last_stats()["D"]
            surfel.set_tight_binning(True)

    # ---- roofline of the dominant kernel (blend backward), measured with HIP events ----
    roof = None
    if "blend_bwd" in prof and prof["blend_bwd"]["count"] > 0:
        avg_ms = prof["blend_bwd"]["ms"] / prof["blend_bwd"]["count"]
        alg = algorithmic_bytes_blend_bwd(stats["D"], H * W, stats["n_vis"])
        achieved = alg / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "blend_bwd_kernel", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": measured_traffic(args.splats, W, H), "avg_kernel_ms": round(avg_ms, 4),
                "algorithmic_bytes": int(alg),
                "note": "achieved / avg_kernel_ms are measured in this run (HIP events on the "
                        "launch stream); traffic is NOT: it is the FETCH_SIZE / WRITE_SIZE "
                        "figure of the committed rocprofv3 PMC passes of this same workload "
                        "(profiles/, null for any other workload), as 2 x FETCH_SIZE + WRITE_SIZE: the doubling "
                        "is the guide's correction for wide streaming reads, an upper bound for this "
                        "kernel's 16-byte gathers (the un-doubled sum is in the same profile record).  "
                        "The kernel is bound by "
                        "vector-instruction issue, not by HBM (profiles/*/blend_sq_counters.txt: "
                        "7.46e8 VALU wave-instructions per launch on 1024 SIMDs, 3.3 cycles each "
                        "at 2.4 GHz = the measured millisecond); "
                        "more than half of the traffic is its write side, one 80-byte float atomic "
                        "per (8x8 pixel block, splat) pair where the algorithmic count has one "
                        "84-byte row per visible splat (an LDS accumulator per tile halves it and "
                        "was measured slower)"}
    # per view for the stages of the view step (binning is two timed calls: the sort and the
    # tile order), per launch for everything else
    views_timed = prof_all.get("blend_fwd", {}).get("count", 0)
    per_view = {"project_fwd", "bin", "blend_fwd", "blend_bwd", "project_bwd", "adam", "loss"}
    stage_ms = {k: round(v["ms"] / max(views_timed if (k in per_view and views_timed) else v["count"], 1), 4)
                for k, v in prof_all.items()}
    if loop is not None:
        ev = loop["events"]
        stage_ms.update({"loop_score_top3": round(ev[0].elapsed_time(ev[1]), 3),
                         "loop_reanchor": round(ev[1].elapsed_time(ev[2]), 3),
                         "loop_tracker_refresh": round(ev[2].elapsed_time(ev[3]), 3)})
    if roof is not None:
        stage_ms["blend_bwd"] = roof["avg_kernel_ms"]     # the timed region's own figure

    # ---- the dense-BA kernels on their own (inside the step they share the GPU with the view
    # steps of the window, so their stage_ms there are not kernel times) ----
    roof_dba = None
    if dba is not None and rank == 0:
        profiling.select(None)
        profiling.enable(True)
        for _ in range(5):
            dba.run()
        torch.cuda.synchronize()
        pd = profiling.read()
        profiling.enable(False)
        ds = dba.describe()
        ht, wd = (int(x) for x in ds["grid"].split("x"))
        HWd, E, P, K = ht * wd, ds["edges"], ds["free_poses"], ds["depth_maps"]
        per = lambda k: pd[k]["ms"] / max(pd[k]["count"], 1) if k in pd else None
        t_lin, t_schur, t_solve, t_retr = per("dba_jacobian"), per("dba_schur"), per("dba_solve"), per("dba_retract")
        if t_lin and t_schur:
            lin_bytes = 76 * E * HWd                      # 20 B read + 56 B written per edge and pixel
            schur_flop = 2 * (6 * P) ** 2 * K * HWd       # dense upper bound of E Q E^T
            roof_dba = {
                "linearize": {"bound": "hbm", "kernel": "dba_linearize_kernel",
                              "achieved": round(lin_bytes / (t_lin * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(lin_bytes / (t_lin * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "avg_ms": round(t_lin, 4), "algorithmic_bytes": lin_bytes},
                "schur": {"bound": "mfma", "kernel": "dba_schur_kernel", "dtype": "f64",
                          "achieved": round(schur_flop / (t_schur * 1e-3) / 1e12, 3),
                          "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(schur_flop / (t_schur * 1e-3) / 1e12 / F64_MATRIX_PEAK_TFLOPS, 5),
                          "avg_ms": round(t_schur, 4), "algorithmic_flop": schur_flop},
                "solve_ms": None if t_solve is None else round(t_solve, 4),
                "retract_ms": None if t_retr is None else round(t_retr, 4),
                "note": "per Gauss-Newton iteration, the bundle adjustment running alone on the GPU "
                        "(HIP events around each stage); counter evidence for the same kernels "
                        "(FETCH_SIZE / WRITE_SIZE, SQ_INSTS_VALU_MFMA_MOPS_F64, "
                        "SQ_VALU_MFMA_BUSY_CYCLES) is under profiles/r3/dba_counters.txt; the f64 "
                        "matrix peak is AMD's data-sheet figure"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and workload == "c3":
        from oracle import cpu_baseline
        cpu = cpu_baseline.run(args.splats, W, H, V, dba_problem=None if dba is None else dba.np,
                               dba_itrs=2 if dba is None else dba.itrs, window=V)

    if rank == 0:
        out = {
            "metric": "mapping frames/sec (render+bwd%s) @%gM splats %dx%d" % (
                "" if dba is None else "+BA", args.splats / 1e6, W, H),
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "comm": comm, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (gloo rehearsal on one GPU: not a benchmark result)" if args.rehearse_gloo else "")
                    + (" (one-rank RCCL rehearsal of the multi-GPU step: not a benchmark result)" if args.rehearse_rccl else ""),
            "config": {"workload": ("%s: %.1fM splats, %s %dx%d%s" % (
                                        "the reference's own operating point (smallcity.yaml), stepped like BASELINE config 3"
                                        if native else "BASELINE config " + workload[1], args.splats / 1e6,
                                        ("%d-keyframe window per GPU" % V) if workload == "c3" else
                                        ("one %d-keyframe window, %d keyframes per GPU" % (window, V)),
                                        W, H, ("" if dba is None else " + DBA step per GPU") +
                                        ("" if loop is None else " + loop closure over %d keyframes "
                                         "per step (scoring and tracker refresh dealt out over the GPUs, "
                                         "re-anchor on every replica; pose-graph solve excluded)" % loop["K"]))),
                       "scene": args.scene, "stable_frac": args.stable_frac, "binning": args.binning,
                       "blend": "lean (no median-depth / flow planes: the mapping loss has no term "
                                "in them; full_passes = the drop-in rasteriser's passes)",
                       "view_step": "autograd" if (args.autograd or args.torch_loss) else "fused C-ABI sequence",
                       "loss": "torch graph" if args.torch_loss else "fused HIP kernels",
                       "splats": args.splats, "width": W, "height": H, "views_per_gpu": V,
                       "dba": None if dba is None else dict(dba.describe(), stream=args.dba_stream),
                       "instances_D": stats["D"], "visible": stats["n_vis"],
                       "parallelism": f"views sharded over {world} GPU(s), grad all-reduce",
                       "exchange": None if reducer.world == 1 else (
                           {"form": "planar", "rows": mapper.n} if reducer.last_exchanged_rows is None
                           else {"form": "compact", "chunks": args.exchange_chunks,
                                 "rows": reducer.last_exchanged_rows, "of": mapper.n})},
            "roofline": roof, "roofline_dba": roof_dba, "cpu_baseline": cpu, "stage_ms": stage_ms,
            "reference_api": None if reference_api is None else {
                "value": reference_api, "unit": "frames/s",
                "what": "same step through GaussianRasterizer autograd + PyTorch loss graph + "
                        "PyTorch bookkeeping (the reference's call pattern, gaussian_base.py:367-403)"},
            "autograd_api": None if autograd_api is None else {
                "value": autograd_api, "unit": "frames/s",
                "what": "the reference's call pattern with the loss taken from "
                        "harness.mapping_loss_fused: GaussianRasterizer through autograd, "
                        "loss.backward(), PyTorch score bookkeeping, SparseGaussianAdam.step"},
            "full_passes": None if full_passes is None else {
                "value": full_passes, "unit": "frames/s",
                "what": "the fused step with the full blend passes (median depth and flow planes "
                        "computed) instead of the lean ones"},
            "stable_half": None if stable_half is None else dict(
                stable_half, unit="frames/s",
                what="the fused step with a random half of the splats under the stable mask: they "
                     "keep their place in the compositing and their scores, the backward pass "
                     "skips their gradient components and the optimiser their rows"),
            "graph_replay": None if graph_replay is None else {
                "value": graph_replay, "unit": "frames/s",
                "what": "the same fused step with every view replayed from a HIP graph captured "
                        "once per keyframe of the window (SurfelMapper.fused_view_graphed); per-kernel "
                        "events cannot be recorded inside a replay, so the headline value and the "
                        "roofline are taken on the eager launches"},
            "reference_binning": None if reference_binning is None else {
                "value": reference_binning, "unit": "frames/s", "instances_D": reference_binning_D,
                "what": "fused step with the reference's 3-sigma tile squares (bit-exact instance list)"},
            "loss": None if mapper.last_loss is None else round(float(mapper.last_loss), 6),
        }
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1 or args.rehearse_rccl:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
