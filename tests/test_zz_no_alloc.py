"""The allocation-free plan step (vings_splat_compact_plan_static: scratch in a static device
array of the library) against vings_splat_compact_plan with a caller workspace, through the C
ABI: destination indices and kept-row count bit-exact, on two calls in a row, the second with
a smaller input (the scratch keeps the first call's contents)."""
import ctypes

import pytest
import torch

from vings_mono_amd import _lib

pytestmark = pytest.mark.gpu


def _plan_with_workspace(keep):
    L = _lib.lib()
    n = keep.numel()
    dev = keep.device
    dst = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
    ws = torch.empty(int(L.vings_splat_compact_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    m = ctypes.c_int64(0)
    _lib.check(L.vings_splat_compact_plan(n, _lib.ptr(keep), _lib.ptr(dst), _lib.ptr(ws), ws.numel(),
                                          ctypes.byref(m), _lib.stream_ptr(dev)), "plan")
    return dst[:n], int(m.value)


def _plan_static(keep, dst, count):
    L = _lib.lib()
    _lib.check(L.vings_splat_compact_plan_static(keep.numel(), _lib.ptr(keep), _lib.ptr(dst),
                                                 _lib.ptr(count), _lib.stream_ptr(keep.device)),
               "plan_static")


def test_static_plan_matches_workspace_plan_twice():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    first = (torch.rand(1_000_003, generator=g) > 0.3).to(torch.uint8).to(dev)
    second = (torch.rand(70_001, generator=g) > 0.6).to(torch.uint8).to(dev)
    # outputs allocated before the calls: the entry point itself allocates nothing
    dst = torch.empty(first.numel(), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    for keep in (first, second):
        want_dst, want_m = _plan_with_workspace(keep)
        dst.fill_(-7)
        count.fill_(-1)
        _plan_static(keep, dst, count)
        torch.cuda.synchronize()
        assert int(count.item()) == want_m == int(keep.sum().item())
        assert torch.equal(dst[:keep.numel()], want_dst)
        assert bool((dst[keep.numel():] == -7).all())          # nothing written past N


def test_static_plan_allocates_nothing_and_can_be_captured():
    """Device memory (the driver's free-byte count and torch's own books) is the same before
    and after the calls; and the call recorded into a HIP graph -- the scratch address looked up
    beforehand with vings_splat_compact_static_prepare -- replays on new mask contents."""
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    masks = [(torch.rand(300_007, generator=g) > t).to(torch.uint8).to(dev) for t in (0.2, 0.7, 0.5)]
    keep = masks[0].clone()
    dst = torch.empty(keep.numel(), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(L.vings_splat_compact_static_prepare(), "prepare")
    wants = [_plan_with_workspace(m) for m in masks]
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    alloc0, reserved0 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
    for _ in range(3):
        _plan_static(keep, dst, count)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0
    assert (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)) == (alloc0, reserved0)
    assert int(count.item()) == wants[0][1] and torch.equal(dst, wants[0][0])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _plan_static(keep, dst, count)
    for m, (want_dst, want_m) in zip(masks[1:], wants[1:]):
        keep.copy_(m)
        dst.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        assert int(count.item()) == want_m and torch.equal(dst, want_dst)


def test_static_plan_edges():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    count = torch.full((1,), 5, dtype=torch.int64, device=dev)
    # empty input: the count word is still written
    _lib.check(L.vings_splat_compact_plan_static(0, None, None, _lib.ptr(count), _lib.stream_ptr(dev)),
               "plan_static")
    torch.cuda.synchronize()
    assert int(count.item()) == 0
    # beyond the static scratch: refused with a clear message, nothing launched
    limit = int(L.vings_splat_compact_static_max_rows())
    assert limit == 1 << 23
    keep = torch.ones(8, dtype=torch.uint8, device=dev)
    dst = torch.empty(8, dtype=torch.int32, device=dev)
    rc = L.vings_splat_compact_plan_static(limit + 1, _lib.ptr(keep), _lib.ptr(dst), _lib.ptr(count),
                                           _lib.stream_ptr(dev))
    assert rc != 0 and b"static scratch" in L.vings_last_error()
    with pytest.raises(_lib.VingsError):
        _lib.check(rc, "plan_static")
    # exactly one block and a ragged tail
    for n in (1, 256, 257):
        k = torch.ones(n, dtype=torch.uint8, device=dev)
        k[::2] = 0
        d = torch.empty(n, dtype=torch.int32, device=dev)
        _plan_static(k, d, count)
        torch.cuda.synchronize()
        want_dst, want_m = _plan_with_workspace(k)
        assert int(count.item()) == want_m and torch.equal(d, want_dst)
