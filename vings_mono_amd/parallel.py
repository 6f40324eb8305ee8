"""Multi-GPU sharding of the mapping loop: one process per GPU, full splat
replica per GPU, the keyframes of a window dealt out over the ranks.

The reference is single-GPU (every config sets device.tracker = device.mapper =
cuda:0, e.g. /root/reference/configs/hierarchical/smallcity.yaml:11-13), so
this exchange step has no counterpart there.  Per optimisation step every
rank renders and back-propagates its own view, then one collective keeps the
replicas bit-identical:

  SUM  float32 [15,N] (planar): 13 score-weighted parameter gradients, the
       importance score of the view and the visibility flag

The two record maxima that no view step reads (best keyframe per splat, error
score) stay rank-local between merges: every view folds
(importance score bits << 32 | keyframe id, error score bits) into a running
int64 [N,2] maximum, and `SurfelMapper.sync_records` maximises that over the
ranks (MAX all-reduce) once per window, or whenever the records are read.

Both run over RCCL on the GPUs of one node (backend "nccl"); the same code runs
over gloo on CPU tensors in the tests.
"""
import torch

PACK_WIDTH = 15
_SLICES = {"_xyz": (0, 3), "_rgb": (3, 6), "_opacity": (6, 7), "_scaling": (7, 9),
           "_rotation": (9, 13)}
_COL_SCORE0, _COL_VIS = 13, 14


def record_keys(cur, kf_id):
    """int64 [N,2] = (bits(max(s0,0)) << 32 | kf_id, bits(max(s1,0))): non-negative floats order
    like their bit patterns, so an integer MAX picks the best score and, with it, its keyframe
    (ties go to the larger keyframe id)."""
    s0 = cur[:, 0].contiguous().clamp(min=0).view(torch.int32).to(torch.int64)
    s1 = cur[:, 1].contiguous().clamp(min=0).view(torch.int32).to(torch.int64)
    return torch.stack([(s0 << 32) | int(kf_id), s1], dim=1)


def split_record_keys(keys):
    """(best score0 [N] float32, its keyframe id [N] int64, max score1 [N] float32)."""
    best = (keys[:, 0] >> 32).to(torch.int32).view(torch.float32)
    return best, keys[:, 0] & 0xFFFFFFFF, keys[:, 1].to(torch.int32).view(torch.float32)


def shard_views(n_views, world, rank):
    """Contiguous block of the window's keyframes owned by `rank`."""
    per = (n_views + world - 1) // world
    return list(range(rank * per, min(n_views, (rank + 1) * per)))


class GradReducer:
    def __init__(self, n, device, world, group=None):
        self.n, self.world, self.group = n, world, group
        if world > 1:
            self.fbuf = torch.zeros((PACK_WIDTH, n), dtype=torch.float32, device=device)

    def resize(self, n):
        """The splat count changed (pruning, new keyframe): a fresh exchange buffer.  Nothing
        of the old one is kept; it only lives from the pack to the unpack of one view."""
        if n != self.n:
            self.n = n
            if self.world > 1:
                self.fbuf = torch.zeros((PACK_WIDTH, n), dtype=torch.float32,
                                        device=self.fbuf.device)

    class _Done:
        """Handle of a collective that has already completed."""
        def wait(self):
            return True

    def _all_reduce(self, t, op, async_op=False):
        """RCCL reduces device tensors in place on its own stream, ordered after the current
        one.  gloo works on host memory: a device tensor is staged through the host here, in
        the calling thread (its own staging of device tensors, done from a worker thread, was
        seen to stall the GPU queue of a process for good when several processes share one
        card, which is how the tests run two ranks on a one-GPU box)."""
        import torch.distributed as dist
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()                       # waits for the kernels that produced t
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return self._Done() if async_op else None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def all_reduce_packed(self):
        """SUM over `fbuf`, already filled on the device (the fused view step packs it with one
        kernel)."""
        import torch.distributed as dist
        self._all_reduce(self.fbuf, dist.ReduceOp.SUM)

    def all_reduce_record_keys(self, keys, async_op=False):
        """MAX over the ranks of the running record keys ([N,2] int64), in place.  With
        async_op the collective's work handle is returned (wait() before reading `keys`)."""
        if self.world == 1:
            return None
        import torch.distributed as dist
        return self._all_reduce(keys, dist.ReduceOp.MAX, async_op=async_op)

    def reduce(self, grads, cur, vis_local):
        """grads: dict name -> [N,M] tensor (updated in place with the sum over ranks).
        Returns (vis_any [N] bool, score0_sum [N])."""
        if self.world == 1:
            return vis_local, cur[:, 0]
        f = self.fbuf
        for name, (a, b) in _SLICES.items():
            f[a:b] = grads[name].reshape(self.n, b - a).t()
        f[_COL_SCORE0] = cur[:, 0]
        f[_COL_VIS] = vis_local.to(torch.float32)
        self.all_reduce_packed()
        for name, (a, b) in _SLICES.items():
            grads[name].copy_(f[a:b].t().reshape(grads[name].shape))
        return f[_COL_VIS] > 0, f[_COL_SCORE0].clone()
