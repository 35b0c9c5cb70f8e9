#!/usr/bin/env python3
"""Times of the three NerfMLP training kernels alone, for every library given (variants of csrc/mlp.hip built by tools/r06/build_var.sh),
alternating on ONE box: python tools/r06/ab_wgrad.py [lib.so ...]  (no argument: the product library).  One process per library."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from samplenerfro_amd import _lib
    lib_path = sys.argv[2] if sys.argv[2] != "product" else None
    lib = _lib.load(lib_path)
    import torch
    from samplenerfro_amd import ops, synthetic as syn
    rows = 4096 * 128
    B, S = 4096, 128
    dev = "cuda:0"
    P = _lib.PREC_F16X3
    pf = torch.from_numpy(syn.init_params_flat(0, fine=False)["coarse_mlp"]).to(dev)
    packed = ops.nerfmlp_pack(pf, P)
    g = torch.Generator(device=dev).manual_seed(0)
    pd = torch.rand((S, B, 4), device=dev, generator=g) * 2 - 1
    dr = torch.nn.functional.normalize(torch.randn((S, B, 4), device=dev, generator=g), dim=-1)
    d_raw = torch.randn((S, B, 4), device=dev, generator=g) * 1e-3

    def timeit(fn, n=6):
        fn(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            fn(); ev[i + 1].record()
        torch.cuda.synchronize()
        return min(ev[i].elapsed_time(ev[i + 1]) for i in range(n))

    out = []
    ref = None
    for bw in sys.argv[3:]:
        BW = _lib.BACKWARDS[bw]
        pbwd = ops.nerfmlp_pack_bwd(pf, None, BW)
        raw, save = ops.nerfmlp_forward_train(packed, P, pd, dr, None, S, B, BW)
        dy = torch.empty(lib.rnerf_nerfmlp_dy_bytes(rows, BW), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.rnerf_nerfmlp_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
        grads = torch.empty(_lib.NERFMLP_PARAMS, device=dev)
        tf = timeit(lambda: lib.rnerf_nerfmlp_forward_train(packed.data_ptr(), P, pd.data_ptr(), dr.data_ptr(), None, S, B, raw.data_ptr(), save.data_ptr(), BW, 0, None))
        td = timeit(lambda: ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, dy=dy, stages="d", backward=BW))
        tw = timeit(lambda: ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, grads=grads, workspace=ws, dy=dy, stages="w", backward=BW))
        out.append(f"{bw}: fwd_train {tf:.3f} dgrad {td:.3f} wgrad+reduce {tw:.3f} ms  |g| {float(grads.abs().max()):.6e} sum {float(grads.double().sum()):.9e}")
        del save, dy
    print(f"{os.path.basename(sys.argv[2]):28s} " + "   ".join(out), flush=True)
    sys.exit(0)
libs = sys.argv[1:] or ["product"]
for rep in range(2):
    for l in libs:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", l, "f16x3", "f16x3lo8"], check=False)
