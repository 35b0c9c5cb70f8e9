import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from samplenerfro_amd import ops, _lib
dev = torch.device("cuda:0")
B, N = 4096, 1536
rng = np.random.default_rng(0)
G = 512
spec = _lib.Grid.make([G] * 3, [-1.5] * 3, [1.5] * 3)
grid = torch.ones((G, G, G), device=dev)
table = ops.grid_build_table(grid, spec); del grid
for name, axis in (("x", 0), ("y", 1), ("z", 2), ("diag", -1)):
    o = rng.uniform(-1.2, 1.2, (B, 3)).astype(np.float32)
    d = np.zeros((B, 3), np.float32)
    if axis >= 0:
        o[:, axis] = -3.5; d[:, axis] = 1.0
    else:
        o[:] = o - 2.0; d[:] = 1 / np.sqrt(3)
    ot, dt = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    pd, dr, _, _ = ops.march(table, spec, ot, dt, 2.0, 6.0, N)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i in range(5):
        ops.march(table, spec, ot, dt, 2.0, 6.0, N, out=(pd, dr)); ev[i + 1].record()
    torch.cuda.synchronize()
    print(name, "rays: march ms", min(ev[i].elapsed_time(ev[i + 1]) for i in range(5)))
