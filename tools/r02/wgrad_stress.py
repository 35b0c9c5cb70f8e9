#!/usr/bin/env python3
"""Determinism stress of the backward kernels: the same inputs through dgrad + wgrad N times must give bit-identical gradients
(a race in the DMA ring / vmcnt accounting of the transpose-read wgrad would show up as run-to-run differences).
python tools/r02/wgrad_stress.py [repeats=50] [rows=...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from samplenerfro_amd import _lib, ops, synthetic as syn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda:0"
lib = _lib.load()
ok = True
for mode in ("f32", "tf32"):
    for B, S in ((4096, 128), (300, 7), (4096, 9)):
        rows = B * S
        BW = _lib.BACKWARDS[mode]
        pf = torch.from_numpy(syn.init_params_flat(0, fine=False, bias_scale=0.1)["coarse_mlp"]).to(dev)
        packed = ops.nerfmlp_pack(pf, _lib.PREC_F16X3); pbwd = ops.nerfmlp_pack_bwd(pf, None, BW)
        g = torch.Generator(device=dev).manual_seed(1)
        pd = torch.rand((S, B, 4), device=dev, generator=g) * 2 - 1
        dr = torch.nn.functional.normalize(torch.randn((S, B, 4), device=dev, generator=g), dim=-1)
        d_raw = torch.randn((S, B, 4), device=dev, generator=g) * torch.exp(torch.randn((S, B, 1), device=dev, generator=g) * 3)
        ref = None
        for it in range(reps):
            raw, save = ops.nerfmlp_forward_train(packed, _lib.PREC_F16X3, pd, dr, None, S, B, BW)
            grads = ops.nerfmlp_backward(pbwd, packed, _lib.PREC_F16X3, save, d_raw, rows, backward=BW)
            if ref is None:
                ref = grads.clone(); raw0 = raw.clone()
            elif not (torch.equal(ref, grads) and torch.equal(raw0, raw)):
                ok = False
                print(f"[{mode}] rows {rows}: run {it} differs: max |d grad| {float((ref - grads).abs().max()):.3e}")
                break
        print(f"[{mode}] rows {rows}: {reps} runs bit-identical: {ref is not None and ok}", flush=True)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
