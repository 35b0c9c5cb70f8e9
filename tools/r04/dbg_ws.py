import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_whole_path as W
from samplenerfro_amd.train import train_step
for Nf, B in ((0, 160), (24, 160)):
    res = []
    for fill in (0x00, 0xFF, 0x3F):
        model, state, batch, flags = W._train_setup(Nf, B)
        rng = np.array([5, 6], np.uint32)
        train_step(model, rng, W._train_setup(Nf, B)[1], batch, flags)
        model._ws["train"].fill_(fill)
        state, stats, _ = train_step(model, rng, state, batch, flags)
        torch.cuda.synchronize()
        res.append(state.grads.clone())
    n = state.theta.numel()
    for k, g in enumerate(res[1:], 1):
        d = (g != res[0]) & ~(torch.isnan(g) & torch.isnan(res[0]))
        idx = torch.nonzero(d).flatten()
        print("Nf", Nf, "fill", k, "differing", idx.numel(), "of", g.numel(), "n_theta", n, "first", idx[:8].tolist(), "nan in g", int(torch.isnan(g).sum()), "nan in ref", int(torch.isnan(res[0]).sum()))
        if idx.numel():
            i = idx[0].item(); print("   ", g[i].item(), res[0][i].item(), "last idx", idx[-8:].tolist())
