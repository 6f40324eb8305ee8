"""Argument checks of entry points through the C ABI, without a GPU: every check below fails (or
returns early) before the first HIP call, so the status codes and messages can be read on the
host.  Covers empty inputs (N = 0 is a no-op that dereferences nothing), missing pointers and
sizes out of range."""
import ctypes

import pytest
import torch

from vings_mono_amd import _lib, dba

FAKE = 0x1000          # a non-null address that must never be dereferenced by a failed check

# made-up addresses and a real device do not belong in one process, whatever the checks promise
pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="host-only argument checks")


def L():
    return _lib.lib()


def last_error():
    return L().vings_last_error().decode()


def test_top3_merge_arguments():
    f = L().vings_splat_top3_merge
    assert f(0, 1, None, None, None, None, None) == 0            # no splats: nothing to merge
    assert f(0, 4, None, None, None, None, None) == 0
    assert f(8, 0, FAKE, FAKE, FAKE, FAKE, None) != 0            # no rank records at all
    assert "sizes" in last_error()
    assert f(-1, 2, FAKE, FAKE, FAKE, FAKE, None) != 0
    for hole in range(4):
        args = [FAKE] * 4
        args[hole] = None
        assert f(8, 2, *args, None) != 0
        assert "null pointer" in last_error()


@pytest.mark.parametrize("name,lead", [
    ("vings_surfel_blend_backward", []),
    ("vings_surfel_blend_backward_accumulate", []),
    ("vings_surfel_blend_backward_accumulate_lean", []),
    ("vings_surfel_blend_backward_accumulate_ordered", [1]),
    ("vings_surfel_blend_backward_accumulate_stable", [1]),
    ("vings_surfel_blend_backward_auto", []),
])
def test_blend_backward_arguments(name, lead):
    f = getattr(L(), name)
    n_ptr = len(f.argtypes) - len(lead) - 3
    # N = 0: nothing to do, whatever the pointers
    assert f(*lead, 0, 16, 16, *([None] * n_ptr)) == 0
    # sizes
    assert f(*lead, -1, 16, 16, *([FAKE] * n_ptr)) != 0 and "sizes" in last_error()
    assert f(*lead, 4, 0, 16, *([FAKE] * n_ptr)) != 0 and "sizes" in last_error()
    # the gradient row buffer (tenth pointer in every variant) is required
    args = [FAKE] * n_ptr
    args[9] = None
    assert f(*lead, 4, 16, 16, *args) != 0 and "null pointer" in last_error()


def test_auto_backward_needs_its_flag_word():
    f = L().vings_surfel_blend_backward_auto
    args = [FAKE] * 13
    args[11] = None                                               # flag_word (stream is last)
    assert f(4, 16, 16, *args) != 0 and "null pointer" in last_error()


def test_static_compaction_limits_without_a_gpu():
    limit = int(L().vings_splat_compact_static_max_rows())
    assert limit == 1 << 23
    count = ctypes.c_int64(5)
    rc = L().vings_splat_compact_plan_static(limit + 1, FAKE, FAKE, ctypes.addressof(count), None)
    assert rc != 0 and "static scratch" in last_error()
    assert count.value == 5                                       # refused before anything ran


def test_dba_arguments():
    lay = dba._Layout()
    q = dba._fn("vings_dba_layout_query")
    assert q(0, 77, 48, 8, 9, ctypes.byref(lay)) != 0 and "layout args" in last_error()
    assert q(43, 77, 48, 8, 9, None) != 0
    assert q(43, 77, 0, 0, 0, ctypes.byref(lay)) == 0 and lay.total_bytes > 0
    assert q(43, 77, 48, 8, 9, ctypes.byref(lay)) == 0
    need = int(lay.total_bytes)
    ba = dba._fn("vings_dba_ba")
    # frames / window out of range, unaligned and short workspaces: all refused on the host
    base = dict(B=9, ht=43, wd=77, E=48, M=9, t0=1, t1=9)

    def call(ws=FAKE & ~255 or 256, ws_bytes=need, **over):
        a = dict(base, **over)
        return ba(FAKE, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, FAKE, a["B"], a["ht"], a["wd"],
                  a["E"], a["M"], a["t0"], a["t1"], 2, 1e-4, 0.1, 0, ws, ws_bytes, None)

    assert call(B=0) != 0 and "frames" in last_error()
    assert call(t0=5, t1=3) != 0 and "t0" in last_error()
    assert call(t1=12) != 0
    assert call(ws=0x1008) != 0 and "aligned" in last_error()
    assert call(ws=None) != 0
    assert call(ws_bytes=need - 1) != 0 and "too small" in last_error()
