"""Per-kernel timing through the HIP-event hooks of libvings_hip.so.

Events are recorded on the stream each kernel is launched on (torch's current
stream, handed to the C ABI), so the durations are those of the kernels
themselves, not of the Python around them.
"""
import ctypes

from . import _lib


def _fns():
    L = _lib.lib()
    en = _lib.bind("vings_profile_enable", ctypes.c_int, [ctypes.c_int])
    nt = _lib.bind("vings_profile_num_tags", ctypes.c_int, [])
    nm = _lib.bind("vings_profile_tag_name", ctypes.c_char_p, [ctypes.c_int])
    rd = _lib.bind("vings_profile_read", ctypes.c_int,
                   [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)])
    return L, en, nt, nm, rd


def enable(on=True):
    _, en, _, _, _ = _fns()
    _lib.check(en(1 if on else 0), "vings_profile_enable")


def select(tags=None):
    """Time only the named tags (None = all); each timed launch costs two event records."""
    _, _, nt, nm, _ = _fns()
    sel = _lib.bind("vings_profile_select", ctypes.c_int, [ctypes.c_uint32])
    if tags is None:
        mask = 0xFFFFFFFF
    else:
        names = [nm(t).decode() for t in range(nt())]
        mask = 0
        for t in tags:
            mask |= 1 << names.index(t)
    _lib.check(sel(mask), "vings_profile_select")


def read():
    """{tag: {"ms": total milliseconds, "count": launches}} for every tag that ran."""
    _, _, nt, nm, rd = _fns()
    out = {}
    for t in range(nt()):
        ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        _lib.check(rd(t, ctypes.byref(ms), ctypes.byref(cnt)), "vings_profile_read")
        if cnt.value > 0:
            out[nm(t).decode()] = {"ms": ms.value, "count": int(cnt.value)}
    return out
