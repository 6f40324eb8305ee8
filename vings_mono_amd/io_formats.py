"""On-disk formats of the map, so that maps trained here open in the reference's viewers and
its checkpoints load here.  Host-side only (numpy / torch.save); no GPU work.

  PLY   /root/reference/scripts/gaussian/vis_utils.py:186-310 (save_ply / load_ply):
        binary little-endian, one `vertex` element, every property float32, in the order
        x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..{1|2} rot_0..3;
        colour is stored as the degree-0 spherical-harmonic coefficient (rgb - 0.5) / C0,
        opacity / scale / rotation as the raw (pre-activation) parameters; save_mode "3dgs"
        appends a third log-scale of -10.  plyfile (which the reference writes with) is not
        needed: the layout is written and parsed directly.
  .pt   /root/reference/scripts/gaussian/gaussian_base.py:492-525 (save_pt_ckpt /
        load_pt_ckpt): a dict of the five parameter tensors and the five score records under
        the reference's attribute names.
No reference-written sample of either file exists in the reference tree, so the layout is
pinned only by the code cited above (round trip + header checked in tests/test_io_formats.py).
"""
import numpy as np
import torch

C0 = 0.28209479177387814
NAMES = ("_xyz", "_rgb", "_opacity", "_scaling", "_rotation")
RECORDS = {"_global_scores": "global_scores", "_local_scores": "local_scores",
           "_stable_mask": "stable_mask", "_globalkf_id": "globalkf_id",
           "_globalkf_max_scores": "globalkf_max_scores"}


def ply_attributes(save_mode):
    if save_mode not in ("2dgs", "3dgs", "sky"):
        raise ValueError("save_mode must be 2dgs, 3dgs or sky")
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + \
            [f"f_rest_{i}" for i in range(45)] + ["opacity"]
    names += [f"scale_{i}" for i in range(2 if save_mode == "2dgs" else 3)]
    return names + [f"rot_{i}" for i in range(4)]


def save_ply(params, path, save_mode="2dgs"):
    """params: dict of raw parameter tensors / arrays (_xyz, _rgb, _opacity, _scaling, _rotation)."""
    g = lambda k: np.asarray(params[k].detach().cpu().numpy() if torch.is_tensor(params[k])
                             else params[k], np.float32)
    xyz, rgb, opac, scale, rot = (g(k) for k in NAMES)
    n = xyz.shape[0]
    names = ply_attributes(save_mode)
    if save_mode != "2dgs":
        scale = np.concatenate((scale, -10.0 * np.ones((n, 1), np.float32)), axis=1)
    # (vis_utils.py:216 evaluates RGB2SH on a float32 CUDA tensor; there torch divides by a
    # Python scalar by multiplying with the float32 reciprocal of the scalar)
    sh = (rgb - np.float32(0.5)) * (np.float32(1.0) / np.float32(C0))
    table = np.concatenate((xyz, np.zeros((n, 3), np.float32), sh,
                            np.zeros((n, 45), np.float32), opac.reshape(n, 1), scale, rot),
                           axis=1).astype("<f4")
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join("property float %s\n" % a for a in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


def load_ply(path):
    """Returns the dict load_ply of the reference returns: float arrays _xyz [N,3], _rgb [N,3]
    (from the SH DC term), _opacity [N,1], _scaling [N,2] (first two scales), _rotation [N,4]."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header has no end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
                elif n is None:
                    raise ValueError("the vertex element must come first")
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise ValueError("only float32 vertex properties are supported")
                props.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("expected a binary_little_endian PLY with a vertex element")
        data = np.frombuffer(f.read(4 * n * len(props)), dtype="<f4").reshape(n, len(props))
    col = {name: i for i, name in enumerate(props)}
    take = lambda names: np.stack([data[:, col[k]] for k in names], axis=1).astype(np.float64)
    numbered = lambda prefix: sorted((p for p in props if p.startswith(prefix)),
                                     key=lambda s: int(s.split("_")[-1]))
    return {"_xyz": take(["x", "y", "z"]),
            "_rgb": take(["f_dc_0", "f_dc_1", "f_dc_2"]) * C0 + 0.5,
            "_opacity": take(["opacity"]),
            "_scaling": take(numbered("scale_"))[:, :2],
            "_rotation": take(numbered("rot"))}


def save_pt_ckpt(mapper, path, reducer=None):
    """The reference's checkpoint dict from a harness.SurfelMapper.  The five parameter entries
    are nn.Parameters like the reference's own (gaussian_base.py:492-505): its load_pt_ckpt
    assigns them straight to the model and builds the optimiser on them, so plain tensors would
    load with requires_grad=False and could not be trained further.

    Multi-rank jobs: the score records of a rank are complete only after the record keys the
    view steps gathered have been merged over the ranks (SurfelMapper.sync_records, a
    collective).  Two ways to save:
      * every rank calls `mapper.sync_records(reducer)`, then ONE rank (or all) calls
        save_pt_ckpt(mapper, path) -- no collective runs in here;
      * every rank calls save_pt_ckpt(mapper, path_of_this_rank, reducer): the merge runs in
        here, so ALL ranks must make the call (rank 0 alone would wait for the others forever).
    A mapper with unmerged keys and no reducer is refused."""
    if getattr(mapper, "_pending_keys", None) is not None and getattr(mapper, "_records_dirty", True):
        if reducer is None or reducer.world == 1:
            raise ValueError("this mapper holds record keys not yet merged over the ranks: call "
                             "mapper.sync_records(reducer) on every rank first, or pass the job's "
                             "GradReducer here on every rank")
        mapper.sync_records(reducer)
    d = {k: torch.nn.Parameter(mapper.params[k].detach().clone()) for k in NAMES}
    d.update({ref: getattr(mapper, ours) for ref, ours in RECORDS.items()})
    torch.save(d, path)


def load_pt_ckpt(path, device):
    """A harness.SurfelMapper from a checkpoint dict in the reference's layout.  Loaded with
    weights_only=True: nothing in the file is executed."""
    from .harness import SurfelMapper
    d = torch.load(path, map_location="cpu", weights_only=True)
    m = SurfelMapper({k: d[k].detach().float() for k in NAMES}, device)
    for ref, ours in RECORDS.items():
        cur = getattr(m, ours)
        setattr(m, ours, d[ref].to(device=m.device, dtype=cur.dtype).reshape(cur.shape).contiguous())
    return m
