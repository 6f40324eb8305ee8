"""On-disk formats (PLY, .pt checkpoint) in the reference's layout
(/root/reference/scripts/gaussian/vis_utils.py:186-310,
/root/reference/scripts/gaussian/gaussian_base.py:492-525).  Parity unpinned: the reference
tree holds no sample file; checked are the header text, field order, value encoding and the
round trip.  Runs on CPU."""
import numpy as np
import pytest
import torch

from vings_mono_amd import harness, io_formats, synth


@pytest.mark.parametrize("mode,nscale", [("2dgs", 2), ("3dgs", 3), ("sky", 3)])
def test_ply_layout_and_round_trip(tmp_path, mode, nscale):
    raw = synth.make_splats(257, 64, 48, seed=3)
    path = tmp_path / f"map_{mode}.ply"
    io_formats.save_ply(raw, str(path), mode)
    blob = path.read_bytes()
    head, body = blob.split(b"end_header\n", 1)
    lines = head.decode("ascii").splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    props = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    want = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + \
           [f"f_rest_{i}" for i in range(45)] + ["opacity"] + [f"scale_{i}" for i in range(nscale)] + \
           [f"rot_{i}" for i in range(4)]
    assert props == want
    table = np.frombuffer(body, "<f4").reshape(257, len(want))
    np.testing.assert_array_equal(table[:, 0:3], raw["_xyz"].numpy())
    assert not table[:, 3:6].any() and not table[:, 9:54].any()        # normals, higher SH
    np.testing.assert_allclose(table[:, 6:9], (raw["_rgb"].numpy() - 0.5) / io_formats.C0, rtol=1e-6)
    np.testing.assert_array_equal(table[:, 54], raw["_opacity"].numpy()[:, 0])
    np.testing.assert_array_equal(table[:, 55:57], raw["_scaling"].numpy())
    if nscale == 3:
        assert np.all(table[:, 57] == -10.0)
    np.testing.assert_array_equal(table[:, -4:], raw["_rotation"].numpy())
    back = io_formats.load_ply(str(path))
    assert back["_scaling"].shape == (257, 2) and back["_opacity"].shape == (257, 1)
    for k in ("_xyz", "_opacity", "_scaling", "_rotation"):
        np.testing.assert_array_equal(back[k].astype(np.float32), raw[k].numpy())
    np.testing.assert_allclose(back["_rgb"], raw["_rgb"].numpy(), atol=1e-6)


@pytest.mark.parametrize("mode", ["2dgs", "3dgs", "sky"])
def test_ply_bytes_equal_the_layout_oracle(tmp_path, mode):
    """save_ply against oracle/io_oracle.py, which packs the same file value by value with
    `struct` from the reference's source: byte for byte."""
    from oracle import io_oracle
    raw = synth.make_splats(193, 64, 48, seed=11)
    raw["_rgb"][:7] = torch.tensor([0.0, 1.0, 0.5, 0.1, 0.2, 0.3, 0.7])[:, None]   # round values too
    path = tmp_path / f"o_{mode}.ply"
    io_formats.save_ply(raw, str(path), mode)
    want = io_oracle.ply_bytes(*(raw[k].tolist() for k in ("_xyz", "_rgb", "_opacity", "_scaling", "_rotation")),
                               save_mode=mode)
    assert path.read_bytes() == want
    assert io_formats.ply_attributes(mode) == io_oracle.ply_attributes(mode)
    assert tuple(sorted(io_formats.NAMES + tuple(io_formats.RECORDS))) == tuple(sorted(io_oracle.CKPT_KEYS))


def test_ply_reader_rejects_other_layouts(tmp_path):
    p = tmp_path / "ascii.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    with pytest.raises(ValueError):
        io_formats.load_ply(str(p))
    p.write_text("plx\n")
    with pytest.raises(ValueError):
        io_formats.load_ply(str(p))


def test_checkpoint_uses_reference_names_and_round_trips(tmp_path):
    raw = synth.make_splats(100, 64, 48, seed=5)
    m = harness.SurfelMapper(raw, "cpu")
    m.global_scores[:, 0] = torch.arange(100.0)
    m.stable_mask[::3] = True
    m.globalkf_id[:] = 7
    path = str(tmp_path / "ckpt.pt")
    io_formats.save_pt_ckpt(m, path)
    d = torch.load(path, weights_only=True)
    assert sorted(d) == sorted(["_xyz", "_rgb", "_scaling", "_rotation", "_opacity", "_global_scores",
                                "_local_scores", "_stable_mask", "_globalkf_id", "_globalkf_max_scores"])
    assert d["_stable_mask"].dtype == torch.bool and d["_globalkf_id"].dtype == torch.int64
    # the parameter entries are trainable leaves, as the reference's load_pt_ckpt expects
    for k in m.NAMES:
        assert isinstance(d[k], torch.nn.Parameter) and d[k].requires_grad
    m2 = io_formats.load_pt_ckpt(path, "cpu")
    for k in m.NAMES:
        assert torch.equal(m2.params[k].detach(), m.params[k].detach())
    assert torch.equal(m2.global_scores, m.global_scores)
    assert torch.equal(m2.stable_mask, m.stable_mask) and torch.equal(m2.globalkf_id, m.globalkf_id)


@pytest.mark.gpu
def test_trained_map_survives_both_formats_on_the_gpu(tmp_path):
    """A map trained for three view steps on the GPU is written as checkpoint and as PLY and read
    back into fresh mappers: the checkpoint replica renders the same bits and carries the same
    records; the PLY replica (colour goes through the SH-DC encoding) renders within 1e-5.
    Resuming from the checkpoint takes the same fourth step as a mapper built from the same
    tensors (the reference's load_pt_ckpt builds a new optimiser too, gaussian_base.py:507-525)."""
    dev = torch.device("cuda:0")
    W, H = 160, 96
    raw = synth.make_splats(4000, W, H, n_views=4, seed=31)
    intr = synth.intrinsic_for(W, H)
    c2w = synth.window_poses(4)
    gtp = {k: v.to(dev) for k, v in synth.perturb_splats(raw).items()}
    views = []
    with torch.no_grad():
        for k in range(4):
            w2c = torch.linalg.inv(c2w[k]).to(dev)
            pr = harness.render(gtp, w2c, intr)
            views.append((w2c, {"rgb": pr["rgb"].clone(), "depth": pr["depth"].clone(),
                                "depth_cov": torch.ones_like(pr["depth"])}))
    m = harness.SurfelMapper(raw, dev)
    for k, (w2c, gt) in enumerate(views[:3]):
        m.fused_view(harness.view_settings(w2c, intr), intr, gt, w2c[:3, :3].cpu(), k)
    torch.cuda.synchronize()
    ck, ply = str(tmp_path / "map.pt"), str(tmp_path / "map.ply")
    io_formats.save_pt_ckpt(m, ck)
    io_formats.save_ply({k: m.params[k].detach().cpu() for k in m.NAMES}, ply)

    m2 = io_formats.load_pt_ckpt(ck, dev)
    for k in m.NAMES:
        assert m2.params[k].is_cuda and torch.equal(m2.params[k].detach(), m.params[k].detach())
    for rec in ("global_scores", "local_scores", "stable_mask", "globalkf_id", "globalkf_max_scores"):
        assert torch.equal(getattr(m2, rec), getattr(m, rec)), rec
    w2c = views[3][0]
    with torch.no_grad():
        a = harness.render({k: m.params[k].detach() for k in m.NAMES}, w2c, intr)
        b = harness.render({k: m2.params[k].detach() for k in m.NAMES}, w2c, intr)
        back = io_formats.load_ply(ply)
        c = harness.render({k: torch.from_numpy(np.ascontiguousarray(back[k], dtype=np.float32)).to(dev)
                            for k in m.NAMES}, w2c, intr)
    for key in ("rgb", "depth"):
        assert torch.equal(a[key], b[key]), key
        assert float((a[key] - c[key]).abs().max()) <= 1e-5 * max(1.0, float(a[key].abs().max())), key

    # resume: the loaded mapper and a mapper built from the same tensors take the same step
    m3 = harness.SurfelMapper({k: m.params[k].detach().cpu() for k in m.NAMES}, dev)
    for rec in ("global_scores", "local_scores", "stable_mask", "globalkf_id", "globalkf_max_scores"):
        setattr(m3, rec, getattr(m, rec).clone())
    for mm in (m2, m3):
        mm.fused_view(harness.view_settings(w2c, intr), intr, views[3][1], w2c[:3, :3].cpu(), 3)
    torch.cuda.synchronize()
    assert float(m2.last_loss) == float(m3.last_loss)
    for k in m.NAMES:
        # float atomics order the gradient sums differently from run to run
        d = (m2.params[k].detach() - m3.params[k].detach()).abs().max()
        assert float(d) <= 2e-3 * float((m2.params[k].detach() - m.params[k].detach()).abs().max()) + 1e-12, k
