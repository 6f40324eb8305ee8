"""Multi-GPU sharding of the mapping loop: one process per GPU, full splat
replica per GPU, the keyframes of a window dealt out over the ranks.

The reference is single-GPU (every config sets device.tracker = device.mapper =
cuda:0, e.g. /root/reference/configs/hierarchical/smallcity.yaml:11-13), so
this exchange step has no counterpart there.  Per optimisation step every
rank renders and back-propagates its own view, then two collectives keep the
replicas bit-identical:

  SUM  float32 [N,16]: 14 score-weighted parameter gradients, the importance
       score of the view, and the visibility flag
  MAX  int64   [N,2] : (importance score bits << 32 | keyframe id) for the
       best-keyframe bookkeeping, and the error-score bits for its running max

Both run over RCCL on the GPUs of one node (backend "nccl"); the same code runs
over gloo on CPU tensors in the tests.
"""
import torch

PACK_WIDTH = 16
_SLICES = {"_xyz": (0, 3), "_rgb": (3, 6), "_opacity": (6, 7), "_scaling": (7, 9),
           "_rotation": (9, 13)}
_COL_SCORE0, _COL_VIS = 13, 14


def shard_views(n_views, world, rank):
    """Contiguous block of the window's keyframes owned by `rank`."""
    per = (n_views + world - 1) // world
    return list(range(rank * per, min(n_views, (rank + 1) * per)))


class GradReducer:
    def __init__(self, n, device, world, group=None):
        self.n, self.world, self.group = n, world, group
        if world > 1:
            self.fbuf = torch.zeros((n, PACK_WIDTH), dtype=torch.float32, device=device)
            self.ibuf = torch.zeros((n, 2), dtype=torch.int64, device=device)

    def all_reduce_packed(self):
        """SUM over `fbuf` and MAX over `ibuf`, both already filled on the device (the fused
        view step packs them with one kernel); the two collectives run concurrently."""
        import torch.distributed as dist
        h1 = dist.all_reduce(self.fbuf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h2 = dist.all_reduce(self.ibuf, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        h1.wait()
        h2.wait()

    def reduce(self, grads, cur, vis_local, kf_id):
        """grads: dict name -> [N,M] tensor (updated in place with the sum over ranks).
        Returns (vis_any[N] bool, score0_sum[N], score1_max[N], best_score0[N], best_kf[N])."""
        if self.world == 1:
            return (vis_local, cur[:, 0], cur[:, 1], cur[:, 0],
                    torch.full((self.n,), kf_id, dtype=torch.int64, device=cur.device))
        import torch.distributed as dist
        f = self.fbuf
        for name, (a, b) in _SLICES.items():
            f[:, a:b] = grads[name].reshape(self.n, b - a)
        f[:, _COL_SCORE0] = cur[:, 0]
        f[:, _COL_VIS] = vis_local.to(torch.float32)
        f[:, 15] = 0
        i = self.ibuf
        s0 = cur[:, 0].contiguous().clamp(min=0).view(torch.int32).to(torch.int64)
        s1 = cur[:, 1].contiguous().clamp(min=0).view(torch.int32).to(torch.int64)
        i[:, 0] = (s0 << 32) | int(kf_id)
        i[:, 1] = s1
        h1 = dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h2 = dist.all_reduce(i, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        h1.wait()
        h2.wait()
        for name, (a, b) in _SLICES.items():
            grads[name].copy_(f[:, a:b].reshape(grads[name].shape))
        best_score = (i[:, 0] >> 32).to(torch.int32).view(torch.float32)
        best_kf = i[:, 0] & 0xFFFFFFFF
        score1 = i[:, 1].to(torch.int32).view(torch.float32)
        return f[:, _COL_VIS] > 0, f[:, _COL_SCORE0].clone(), score1, best_score, best_kf
