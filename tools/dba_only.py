"""The bench's dense bundle-adjustment problem on its own (for rocprofv3 counter passes):
`ba(itrs=2)` on the 135x240 grid, 48 edges, a few calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vings_mono_amd import dba  # noqa: E402

grid = sys.argv[1] if len(sys.argv) > 1 else "135x240"
ht, wd = (int(x) for x in grid.split("x"))
prob = dba.make_bench_problem(torch.device("cuda:0"), ht, wd)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    prob.run()
torch.cuda.synchronize()
print("dba_only", grid, prob.describe())
