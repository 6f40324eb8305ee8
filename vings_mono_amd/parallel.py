"""Multi-GPU sharding of the mapping loop: one process per GPU, full splat
replica per GPU, the keyframes of a window dealt out over the ranks.

The reference is single-GPU (every config sets device.tracker = device.mapper =
cuda:0, e.g. /root/reference/configs/hierarchical/smallcity.yaml:11-13), so
this exchange step has no counterpart there.  Per optimisation step every
rank renders and back-propagates its own view, then the per-splat sums over
the ranks of the 13 score-weighted parameter gradients and of the view's
importance score keep the replicas bit-identical.  Two forms (`GradReducer`):

  planar   SUM of float32 [15,N] (13 gradients, score, visibility flag) in one
           all-reduce over all N splats
  compact  MAX of the uint8 [N] "touched" mask (known after the projection,
           reduced on a side stream under the rest of the view), then SUM of
           14 floats for the touched rows only, in chunks; chunk k is applied
           (scores + Adam) while chunk k+1 is being reduced

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


EXCHANGE_COLUMNS = 14      # the compact form: 13 gradients + importance score
CHUNK_ALIGN = 256


def chunk_layout(m, n_chunks):
    """(chunk_rows, chunks in use) for m exchanged rows: equal chunks of a multiple of 256 rows,
    the last one partly filled.  Chunk k is the contiguous float range
    [k * 14 * chunk_rows, (k + 1) * 14 * chunk_rows) of the exchange buffer."""
    per = -(-max(m, 1) // max(n_chunks, 1))
    chunk_rows = -(-per // CHUNK_ALIGN) * CHUNK_ALIGN
    return chunk_rows, -(-m // chunk_rows)


class GradReducer:
    """Exchange state of one mapper.  Two forms:

    planar   `fbuf` [15,N], one SUM all-reduce per view over all N rows (`reduce`,
             `all_reduce_packed`): what the autograd-driven view step uses.
    compact  only the rows some rank touched in this step travel, in `n_chunks` contiguous
             chunks, and chunk k is applied (scores + Adam) while chunk k+1 is still being
             reduced (`start_union` / `finish_union` / `exchange_chunks`): what the fused view
             step uses.  The set of touched rows is known right after the forward projection,
             so its MAX all-reduce and the slot scan run on a side stream under the rest of the
             view; by the time the backward pass has been queued the host has the row count.
    """

    def __init__(self, n, device, world, group=None, n_chunks=4, compact=True):
        self.n, self.world, self.group = n, world, group
        self.device = torch.device(device)
        self.n_chunks, self.compact = int(n_chunks), bool(compact)
        self._fbuf = None
        self._cx = None
        self.last_exchanged_rows = None

    @property
    def fbuf(self):
        if self._fbuf is None and self.world > 1:
            self._fbuf = torch.zeros((PACK_WIDTH, self.n), dtype=torch.float32, device=self.device)
        return self._fbuf

    def resize(self, n):
        """The splat count changed (pruning, new keyframe): fresh exchange buffers.  Nothing
        of the old ones is kept; they only live from the pack to the unpack of one view."""
        if n != self.n:
            self.n = n
            self._fbuf = None
            self._cx = None

    # ------------------------------------------------------------------ compact form
    def _compact_state(self):
        if self._cx is None:
            from . import _lib
            n, dev = self.n, self.device
            ws_bytes = int(_lib.lib().vings_splat_compact_workspace_bytes(n))
            cx = {
                "touched": torch.zeros(max(n, 1), dtype=torch.uint8, device=dev),
                "slot": torch.empty(max(n, 1), dtype=torch.int32, device=dev),
                "rows": torch.empty(max(n, 1), dtype=torch.int32, device=dev),
                "count_dev": torch.zeros(1, dtype=torch.int64, device=dev),
                "ws": torch.empty(ws_bytes, dtype=torch.uint8, device=dev), "ws_bytes": ws_bytes,
                # every chunk may carry up to CHUNK_ALIGN - 1 unused rows
                "packed": torch.zeros(EXCHANGE_COLUMNS * (n + self.n_chunks * CHUNK_ALIGN),
                                      dtype=torch.float32, device=dev),
            }
            if dev.type == "cuda":
                cx["count_host"] = torch.zeros(1, dtype=torch.int64).pin_memory()
                cx["side"] = torch.cuda.Stream(device=dev)
                cx["ev_fwd"] = torch.cuda.Event()
                cx["ev_plan"] = torch.cuda.Event()
            self._cx = cx
        return self._cx

    def start_union(self, radii):
        """Queued right after the forward projection of the view (radii [N] int32, on the
        current stream): touched = radii > 0, MAX over the ranks, slot scan, row count to the
        host, all on a side stream.  Returns at once."""
        import torch.distributed as dist
        from . import _lib
        L, cx, n = _lib.lib(), self._compact_state(), self.n
        main = torch.cuda.current_stream(self.device)
        _lib.check(L.vings_splat_touched(n, _lib.ptr(radii), _lib.ptr(cx["touched"]),
                                         main.cuda_stream), "vings_splat_touched")
        cx["ev_fwd"].record(main)
        side = cx["side"]
        with torch.cuda.stream(side):
            side.wait_event(cx["ev_fwd"])
            self._all_reduce(cx["touched"], dist.ReduceOp.MAX)
            _lib.check(L.vings_splat_compact_plan_device(
                n, _lib.ptr(cx["touched"]), _lib.ptr(cx["slot"]), _lib.ptr(cx["ws"]),
                cx["ws_bytes"], _lib.ptr(cx["count_dev"]), side.cuda_stream),
                "vings_splat_compact_plan_device")
            cx["count_host"].copy_(cx["count_dev"], non_blocking=True)
            cx["ev_plan"].record(side)

    def finish_union(self):
        """(rows to exchange, chunk_rows).  The host waits for the scan started by
        `start_union` (long finished when this is called after the backward pass has been
        queued); the current stream is ordered after it."""
        cx = self._cx
        cx["ev_plan"].synchronize()
        torch.cuda.current_stream(self.device).wait_event(cx["ev_plan"])
        m = int(cx["count_host"][0])
        self.last_exchanged_rows = m
        chunk_rows, _ = chunk_layout(m, self.n_chunks)
        return m, chunk_rows

    def exchange_chunks(self, m, chunk_rows, apply):
        """SUM all-reduce of every chunk in use, all started at once (they run one after the
        other on the collective's stream); `apply(chunk, first_slot, n_slots)` is called for
        chunk k once the current stream is ordered after its all-reduce, i.e. the kernels it
        launches run while chunk k+1 is still in flight."""
        import torch.distributed as dist
        cx = self._compact_state()
        per = EXCHANGE_COLUMNS * chunk_rows
        n_used = -(-m // chunk_rows)
        chunks = [cx["packed"][k * per:(k + 1) * per] for k in range(n_used)]
        works = [self._all_reduce(c, dist.ReduceOp.SUM, async_op=True) for c in chunks]
        for k, (c, w) in enumerate(zip(chunks, works)):
            w.wait()
            apply(c, k * chunk_rows, min(chunk_rows, m - k * chunk_rows))

    # ------------------------------------------------------------------ collectives

    class _Done:
        """Handle of a collective that has already completed."""
        def wait(self):
            return True

    def _all_reduce(self, t, op, async_op=False):
        """RCCL reduces device tensors in place on its own stream, ordered after the current
        one.  gloo works on host memory, and a device tensor is staged through the host HERE,
        in the calling thread, not handed to gloo: torch's gloo backend stages a device tensor
        itself (pinned buffer + asynchronous copy on a pool stream, set up inside
        `all_reduce`), and with two processes on one GPU that call never returns, in either
        rank, once the tensor is larger than a few hundred KB and the caller's stream still
        has kernels queued when it is made (tools/gloo_probe.py reproduces it with plain
        torch and no kernel of this library: `busy 100000` hangs, `idle` passes up to 60 MB).
        `t.cpu()` is a synchronous copy on the current stream: it drains the queue first, so
        gloo is only ever entered with nothing of this rank pending on the GPU and sees host
        memory, which is the case that passes at every size.  Only the test rig (several ranks
        on the one GPU of the test box) takes this branch."""
        import torch.distributed as dist
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()                       # waits for the kernels that produced t
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return self._Done() if async_op else None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def rank(self):
        import torch.distributed as dist
        return dist.get_rank(self.group) if self.world > 1 else 0

    def comm_world(self):
        """Ranks of the communicator itself (a rehearsal may drive the exchange of `world`
        ranks over a communicator of one)."""
        import torch.distributed as dist
        return dist.get_world_size(self.group) if self.world > 1 and dist.is_initialized() else 1

    def all_gather(self, out, local):
        """out [world * rows, ...] <- the `local` [rows, ...] of every rank, rank-major (local may
        be the calling rank's own slice of out).  gloo takes host memory, staged here like
        `_all_reduce` does."""
        import torch.distributed as dist
        if out.is_cuda and dist.get_backend(self.group) == "gloo":
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h, local.cpu(), group=self.group)
            out.copy_(h)
            return
        dist.all_gather_into_tensor(out, local, group=self.group)

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
