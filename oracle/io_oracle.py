"""Byte layout of the map's on-disk formats, restated value by value with `struct`.

TEST INFRASTRUCTURE ONLY: imported by tests/.  The shipped package never imports this module.

Restates /root/reference/scripts/gaussian/vis_utils.py:186-249 (construct_list_of_attributes,
save_ply) and the writer it hands the table to (plyfile's PlyData([el]).write on a
little-endian host: text header, then the structured rows packed as little-endian float32), and
/root/reference/scripts/gaussian/gaussian_base.py:492-505 (save_pt_ckpt's key set).
PARITY UNPINNED: the reference tree holds no file written by either function and `plyfile` is
not importable here, so nothing but the cited source pins these bytes.
"""
import struct

C0 = 0.28209479177387814


def ply_attributes(save_mode):
    """vis_utils.py:186-204."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += ["f_dc_%d" % i for i in range(3)]
    names += ["f_rest_%d" % i for i in range(45)]
    names.append("opacity")
    if save_mode in ("3dgs", "sky"):
        names += ["scale_%d" % i for i in range(3)]
    elif save_mode == "2dgs":
        names += ["scale_%d" % i for i in range(2)]
    else:
        raise AssertionError("Invalid save_mode.")
    return names + ["rot_%d" % i for i in range(4)]


def _f32(x):
    """Round a Python float to float32 and back (the value a float32 array element holds)."""
    return struct.unpack("<f", struct.pack("<f", x))[0]


def sh_dc(rgb_value):
    """RGB2SH of vis_utils.py:212-216 the way the reference evaluates it: on a float32 CUDA
    tensor, where torch divides by a Python scalar by multiplying with the float32 reciprocal
    of the (float32-rounded) scalar."""
    inv = _f32(1.0 / _f32(C0))
    return _f32(_f32(_f32(rgb_value) - 0.5) * inv)


def ply_bytes(xyz, rgb, opacity, scaling, rotation, save_mode="2dgs"):
    """The file save_ply writes for these per-splat rows (lists of lists of floats)."""
    names = ply_attributes(save_mode)
    n = len(xyz)
    head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    head += "".join("property float %s\n" % a for a in names) + "end_header\n"
    out = [head.encode("ascii")]
    for i in range(n):
        row = list(xyz[i]) + [0.0, 0.0, 0.0]                       # normals: zeros (:214)
        row += [sh_dc(c) for c in rgb[i]] + [0.0] * 45             # SH: DC term only (:217-221)
        row += [opacity[i][0] if isinstance(opacity[i], (list, tuple)) else opacity[i]]
        row += list(scaling[i]) + ([-10.0] if save_mode != "2dgs" else [])   # (:225-226)
        row += list(rotation[i])
        assert len(row) == len(names)
        out.append(struct.pack("<%df" % len(row), *row))
    return b"".join(out)


CKPT_KEYS = ("_xyz", "_rgb", "_scaling", "_rotation", "_opacity", "_global_scores", "_local_scores",
             "_stable_mask", "_globalkf_id", "_globalkf_max_scores")          # gaussian_base.py:493-504
