"""Generates tests/golden/dba_solve_golden.npz by IMPORTING the reference's own
normal-equation solvers (run in the development container only; /root/reference does not
exist on the GPU box).  The fixture holds inputs and expected outputs only.

Reference functions exercised:
  frontend.geom.chol.schur_solve     chol.py:46-73   damping, Schur complement, Cholesky,
                                                     back-substitution of the depth update
  frontend.geom.chol.block_solve     chol.py:32-43   motion-only solve

chol.py imports frontend.geom.projective_ops, which imports the absent `lietorch` package at
module level.  The two solvers never touch it, so an EMPTY module of that name (three empty
classes, nothing callable) is registered before the import; no behaviour is substituted.
The system fed to the solvers comes from this build's CPU oracle on a small seeded problem,
so the fixture pins the linear-algebra half of the bundle adjustment; the geometric half
(projection and Jacobians) needs a real SE3 implementation and stays unpinned.

Usage:  python tests/golden/make_dba_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference/scripts")

_empty = types.ModuleType("lietorch")
for _name in ("SE3", "Sim3", "SO3"):
    setattr(_empty, _name, type(_name, (), {}))
sys.modules["lietorch"] = _empty

from frontend.geom import chol as ref_chol     # noqa: E402
from oracle import dba_oracle as O             # noqa: E402

OUT = os.path.join(HERE, "dba_solve_golden.npz")


def small_problem(seed):
    rng = np.random.default_rng(seed)
    B, ht, wd = 6, 9, 12
    poses = np.zeros((B, 7))
    poses[:, 6] = 1.0
    for k in range(B):
        xi = np.concatenate([[0.15 * k, 0.01 * k, 0.02 * k], rng.normal(0, 0.01, 3)])
        poses[k] = O.se3_inv(O.se3_mul(O.se3_exp(xi[None])[0], np.array([0, 0, 0, 0, 0, 0, 1.0])))
    disps = rng.uniform(0.1, 0.8, (B, ht, wd))
    intr = np.array([10.0, 10.5, wd / 2.0, ht / 2.0])
    ii, jj = [], []
    for i in range(B):
        for j in range(B):
            if 1 <= abs(i - j) <= 2:
                ii.append(i); jj.append(j)
    ii, jj = np.array(ii), np.array(jj)
    coords, _ = O.reproject(poses, disps, intr, ii, jj)
    target = (coords + rng.normal(0, 0.5, coords.shape)).transpose(0, 3, 1, 2)
    weight = rng.uniform(0, 1, target.shape)
    t0, t1 = 1, B
    kx, _ = O.depth_variables(ii, t0, t1)
    eta = 1e-3 * rng.uniform(0.5, 1.5, (len(kx), ht, wd))
    sens = np.zeros_like(disps)
    return O.build_system(poses, disps, intr, sens, target, weight, eta, ii, jj, t0, t1)


def main():
    out = {}
    for tag, seed in (("a", 11), ("b", 12)):
        s = small_problem(seed)
        H = torch.from_numpy(s["H"])[None]            # [1,P,P,6,6]
        E = torch.from_numpy(s["E"])[None]            # [1,P,M,6,HW]
        C = torch.from_numpy(s["C"])[None]            # [1,M,HW]
        v = torch.from_numpy(s["v"])[None]
        w = torch.from_numpy(s["w"])[None]
        for name, (ep, lm) in (("std", (0.1, 1e-4)), ("stiff", (1e-3, 1e-2))):
            dx, dz = ref_chol.schur_solve(H, E, C, v, w, ep=ep, lm=lm)
            out[f"{tag}_{name}_dx"], out[f"{tag}_{name}_dz"] = dx[0].numpy(), dz[0].numpy()
            out[f"{tag}_{name}_eplm"] = np.array([ep, lm])
        out[f"{tag}_motion_dx"] = ref_chol.block_solve(H, v, ep=0.1, lm=1e-4)[0].numpy()
        for k in ("H", "E", "C", "v", "w"):
            out[f"{tag}_{k}"] = s[k]
    # a matrix that is not positive definite: the solvers return zeros (chol.py:9-17)
    Hbad = -torch.eye(12, dtype=torch.float64).reshape(2, 6, 2, 6).permute(0, 2, 1, 3)[None].contiguous()
    out["bad_dx"] = ref_chol.block_solve(Hbad, torch.ones(1, 2, 6, dtype=torch.float64), ep=0.1, lm=0.0)[0].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
