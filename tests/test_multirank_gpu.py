"""Two, four and eight ranks of the sharded mapping loop on the GPU: replicas must stay bit-identical
after every exchanged step, the fused C-ABI view step must agree with the autograd-driven
formulation of the same step, and its compact exchange (touched rows only, in chunks) with
its planar one.  The ranks share the test box's single GPU and exchange over gloo; on a
multi-GPU node bench.py runs the same code over RCCL."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rank_view_steps(tmp_path, world):
    out = str(tmp_path / "state")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_multirank_worker.py"), out]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    s0 = torch.load(out + ".rank0.pt", weights_only=True)
    names = ("_xyz", "_rgb", "_opacity", "_scaling", "_rotation")
    for r in range(1, world):
        s1 = torch.load(out + f".rank{r}.pt", weights_only=True)
        for impl in ("autograd", "fused", "fused_synced_first", "fused_planar"):
            for k in names + ("global_scores", "local_scores", "kf_id", "kf_max"):
                assert torch.equal(s0[impl][k], s1[impl][k]), f"{impl}: replicas 0 and {r} differ in {k}"
        assert s1["exchanged_rows"] == s0["exchanged_rows"]
    # checkpoints: rank 0 alone saved the merged map; an unmerged one was refused without the
    # reducer and, with it, every rank wrote the same records
    assert os.path.exists(out + ".merged.ckpt") and not os.path.exists(out + ".refused.ckpt")
    ck = [torch.load(out + f".rank{r}.ckpt", weights_only=True) for r in range(world)]
    for r in range(world):
        assert torch.load(out + f".rank{r}.pt", weights_only=True)["unmerged_save_refused"]
        for key in ("_xyz", "_global_scores", "_local_scores", "_globalkf_id", "_globalkf_max_scores"):
            assert torch.equal(ck[0][key], ck[r][key]), key
    # loop closure dealt out over the ranks = the same calls on one rank: the depth maps bit for
    # bit; the scores to what two runs of the importance pass differ by (its per-tile partial
    # sums meet in float atomics; the merge itself is checked bit-exact in
    # test_maintenance_parity.py), near-ties between keyframes may then swap owners
    for r in range(world):
        lp = torch.load(out + f".rank{r}.pt", weights_only=True)["loop"]
        assert lp["dense_equal"] and lp["coarse_equal"], (r, lp)
        assert lp["scores_err"] <= 1e-4 and lp["ids_match"] >= 0.999, (r, lp)
        assert lp["scored"] > 1000 and lp["kf_seen"] > 2 and lp["refreshed"] > 1000, lp
    # the compact exchange moved the rows some rank saw, not the whole map
    assert all(0 < m < 5300 for m in s0["exchanged_rows"]), s0["exchanged_rows"]
    # compact against planar exchange of the same fused step (two runs of the blend backward:
    # float atomics, so a tolerance)
    for k in names:
        ub = s0["fused"][k] - s0["raw"][k]
        ud = s0["fused_planar"][k] - s0["raw"][k]
        assert float((ub - ud).abs().max()) <= 2e-2 * float(ub.abs().max()), k
    for k in ("global_scores", "local_scores", "kf_max"):
        scale = float(s0["fused"][k].abs().max())
        assert float((s0["fused"][k] - s0["fused_planar"][k]).abs().max()) <= 1e-4 * scale, k
    # Between the second and the third step every mapper was pruned and extended.  The record
    # keys a rank had not yet merged travelled with their splats: the mapper that merged them
    # before the maintenance ends with the same records and parameters, bit for bit.
    # (Two runs of the same step are not bit-identical, float atomics add in another order, so
    # this is the comparison the fused and autograd steps get below.)
    n_after = 5000 - len(range(4, 5000, 9)) + 300
    assert s0["fused"]["_xyz"].shape[0] == n_after
    for k in names:
        ub = s0["fused"][k] - s0["raw"][k]
        uc = s0["fused_synced_first"][k] - s0["raw"][k]
        assert float((ub - uc).abs().max()) <= 2e-2 * float(ub.abs().max()), k
    for k in ("global_scores", "local_scores", "kf_max"):
        scale = float(s0["fused"][k].abs().max())
        assert float((s0["fused"][k] - s0["fused_synced_first"][k]).abs().max()) <= 1e-4 * scale, k
    assert float((s0["fused"]["kf_id"] == s0["fused_synced_first"]["kf_id"]).float().mean()) >= 0.999
    for k in names:
        ua = s0["autograd"][k] - s0["raw"][k]
        ub = s0["fused"][k] - s0["raw"][k]
        scale = float(ua.abs().max())
        assert scale > 0, k
        assert float((ua - ub).abs().max()) <= 2e-2 * scale, k
    # the record maxima were merged over the ranks (once mid-run, once at the end)
    assert float(s0["fused"]["kf_max"].max()) > 0 and float(s0["fused"]["local_scores"][:, 1].max()) > 0
    assert len(torch.unique(s0["fused"]["kf_id"])) > 1
    for k in ("global_scores", "local_scores", "kf_max"):
        scale = float(s0["autograd"][k].abs().max())
        assert float((s0["autograd"][k] - s0["fused"][k]).abs().max()) <= 1e-4 * scale, k


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts the
    bench): the process starts its two ranks itself and hands on rank 0's line."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-gloo",
           "--splats", "200000", "--width", "960", "--height", "544", "--steps", "1", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["comm"]["backend"] == "gloo"
    assert rec["metric"] == "mapping frames/sec (render+bwd+BA) @0.2M splats 960x544"
    assert "config 3" in rec["config"]["workload"] and rec["value"] > 0
    assert rec["config"]["exchange"]["form"] == "compact"


def test_exchange_over_rccl_one_rank(tmp_path):
    """The exchange with its collectives issued through RCCL (backend "nccl") on a one-rank
    communicator — all this box can host: the sums are over one rank, so the step must equal
    the single-GPU step (to what two runs of the float-atomic backward differ by), and the
    record maxima must come back unchanged through the int64 MAX."""
    out = str(tmp_path / "rccl.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_worker.py"), out], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    s = torch.load(out, weights_only=True)
    assert s["backend"] == "nccl"
    assert all(0 < m < 5000 for m in s["exchanged_rows"]), s["exchanged_rows"]
    names = ("_xyz", "_rgb", "_opacity", "_scaling", "_rotation")
    for impl in ("autograd", "fused", "fused_planar"):
        for k in names:
            us = s["single"][k] - s["raw"][k]
            ui = s[impl][k] - s["raw"][k]
            scale = float(us.abs().max())
            assert scale > 0, k
            assert float((us - ui).abs().max()) <= 2e-2 * scale, (impl, k)
        for k in ("global_scores", "local_scores", "kf_max"):
            scale = float(s["single"][k].abs().max())
            assert float((s["single"][k] - s[impl][k]).abs().max()) <= 1e-4 * scale, (impl, k)
        assert float((s["single"]["kf_id"] == s[impl]["kf_id"]).float().mean()) >= 0.999, impl
