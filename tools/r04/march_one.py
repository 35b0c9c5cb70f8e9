"""One march launch set (5 x 4096 rays x 1536 nodes) at a given grid size / table layout: the program the --pmc passes of tools/r04/march_pmc.sh profile.
usage: python3 tools/r04/march_one.py G layout"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from samplenerfro_amd import ops, _lib, synthetic as syn
G, layout = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda:0")
B, N = 4096, 1536
o, d = syn.sphere_rays(B)
o = torch.from_numpy(o).to(dev); d = torch.from_numpy(d).to(dev)
spec = _lib.Grid.make([G] * 3, [-1.5] * 3, [1.5] * 3, layout)
table = ops.grid_build_table(torch.ones((G, G, G), device=dev), spec)
pd, dr, _, _ = ops.march(table, spec, o, d, 2.0, 6.0, N)
for i in range(4):
    ops.march(table, spec, o, d, 2.0, 6.0, N, out=(pd, dr))
torch.cuda.synchronize()
