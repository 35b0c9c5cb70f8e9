#!/usr/bin/env python3
"""Numerics of a cheaper split of the fp32-grade product (a DESIGN NOTE for a future round, CPU only: nothing here is product code).

Today every NerfMLP product is x·W = x_hi·W_hi + x_hi·W_lo + x_lo·W_hi on f16 MFMAs (3 passes, error ~2^-22).  The two cross terms carry
2^-11 of the magnitude: they do not need 11-bit operands.  On gfx950 the scaled fp8 MFMA (32x32x64 f8f6f4) runs at twice the f16 rate
(profiles/r03/ubench_mfma_fp8.txt), so  x_hi·W_hi [f16]  +  fp8(x)·fp8(W_lo)  +  fp8(x_lo)·fp8(W) [MX block scales]  would cost 2/3 of
the matrix time.  This script measures what that does to the rendered colour, on the bench workload's weights and on scaled-up weights,
against float64, next to the shipped f16x3 / f16x2 arithmetics.  Emulation: operands rounded to the stated formats (e4m3 with one power-of-two
scale per 32 consecutive K elements, as the MX formats define), products and sums in float64 (the MFMA accumulates in fp32: its own
~1e-7 is below everything measured here).

usage: python tools/fp8_cross_term_error.py [out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_np as R                       # noqa: E402
from samplenerfro_amd import synthetic as syn        # noqa: E402


def f16(x):
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)


def e4m3_block(x, axis, block=32):
    """Round to fp8 e4m3 with one E8M0 (power-of-two) scale per `block` consecutive elements along `axis` (the K axis)."""
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    K = x.shape[-1]
    pad = (-K) % block
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, pad)])
    xb = xp.reshape(xp.shape[:-1] + (-1, block))
    amax = np.abs(xb).max(-1, keepdims=True)
    scale = 2.0 ** np.ceil(np.log2(np.maximum(amax, 1e-300) / 448.0))          # largest element <= 448 after scaling
    y = xb / scale
    e = np.clip(np.floor(np.log2(np.maximum(np.abs(y), 2.0 ** -9))), -6, 8)    # exponent (subnormals below 2^-6)
    q = np.round(y / 2.0 ** (e - 3)) * 2.0 ** (e - 3)
    q = np.clip(q, -448.0, 448.0) * scale
    q = q.reshape(xp.shape)[..., :K]
    return np.moveaxis(q, -1, axis)


def mx_block(x, axis, ebits, mbits, emax_val, block=32):
    """Round to a small MX float (1 sign, `ebits` exponent, `mbits` mantissa bits, largest finite value `emax_val`, subnormals, no inf / nan:
    e2m3 = fp6 (max 7.5), e2m1 = fp4 (max 6)) with one E8M0 scale per `block` consecutive elements along `axis`."""
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    K = x.shape[-1]
    pad = (-K) % block
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, pad)])
    xb = xp.reshape(xp.shape[:-1] + (-1, block))
    amax = np.abs(xb).max(-1, keepdims=True)
    scale = 2.0 ** np.ceil(np.log2(np.maximum(amax, 1e-300) / emax_val))
    y = xb / scale
    bias = 2 ** (ebits - 1) - 1
    emin = 1 - bias                                                            # exponent of the smallest normal
    e = np.maximum(np.floor(np.log2(np.maximum(np.abs(y), 2.0 ** (emin - mbits - 2)))), emin)
    q = np.round(y / 2.0 ** (e - mbits)) * 2.0 ** (e - mbits)
    q = np.clip(q, -emax_val, emax_val) * scale
    q = q.reshape(xp.shape)[..., :K]
    return np.moveaxis(q, -1, axis)


def e2m3_block(x, axis):
    return mx_block(x, axis, 2, 3, 7.5)


def e2m1_block(x, axis):
    return mx_block(x, axis, 2, 1, 6.0)


MODES = ("f16x3", "f16x2", "f16+fp8x2", "f16+fp6x2", "f16+fp4x2")


def make_dense(mode):
    def dense(p, x, acc_dtype=None):
        k, b = np.asarray(p["kernel"], np.float64), np.asarray(p["bias"], np.float64)
        x = np.asarray(x, np.float64)
        xh, kh = f16(x), f16(k)
        xl, kl = f16(x - xh), f16(k - kh)
        if mode == "f16x3":
            y = xh @ kh + xh @ kl + xl @ kh
        elif mode == "f16x2":                     # exact weights (hi + lo) x activations rounded to f16: the shipped opt-in precision
            y = xh @ kh + xh @ kl
        elif mode == "f16+fp8x2":                 # f16 main term + both cross terms on block-scaled fp8 operands
            y = xh @ kh + e4m3_block(xh, -1) @ e4m3_block(kl, 0) + e4m3_block(xl, -1) @ e4m3_block(kh, 0)
        elif mode == "f16+fp6x2":                 # the same with fp6 e2m3 operands (the f8f6f4 MFMA issues fp6 / fp4 at twice the fp8 rate)
            y = xh @ kh + e2m3_block(xh, -1) @ e2m3_block(kl, 0) + e2m3_block(xl, -1) @ e2m3_block(kh, 0)
        else:
            y = xh @ kh + e2m1_block(xh, -1) @ e2m1_block(kl, 0) + e2m1_block(xl, -1) @ e2m1_block(kh, 0)
        return y + b
    return dense


def render(params, table, mc, o, d, jitter, dense):
    old = R._dense
    R._dense = dense
    try:
        ret, _ = R.nerf_forward(mc, params, table, o, d, jitter, dtype=np.float64)
    finally:
        R._dense = old
    return ret[-1][0]


def main():
    G, ext, B, S, P = 64, 1.5, 192, 128, 12
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, ext, 0.6), 0.5).reshape(-1, 1), [G] * 3, 3, 1.0).reshape(G, G, G)
    table = R.build_table(grid.astype(np.float64), [G] * 3, [-ext] * 3, [ext] * 3, np.float64)
    o, d = syn.sphere_rays(B, seed=syn.SEED + 7)
    mc = R.ModelConfig([G] * 3, [-ext] * 3, [ext] * 3, near=2.0, far=6.0, num_coarse_samples=S, num_fine_samples=0, num_path_samples=P)
    jitter = np.arange(0, S * P, P) + P // 2
    results = []
    for name, scale, bias in (("bench initial weights (glorot)", 1.0, 0.0), ("hidden kernels x 1.5, biases 0.1", 1.5, 0.1), ("hidden kernels x 2", 2.0, 0.1)):
        pf = syn.init_params_flat(0, fine=False, bias_scale=bias)
        tree = syn.params_tree(pf)
        for l in range(1, 8):
            tree["coarse_mlp"][f"Dense_{l}"]["kernel"] = tree["coarse_mlp"][f"Dense_{l}"]["kernel"] * scale
        exact = render(tree, table, mc, o, d, jitter, lambda p, x, acc_dtype=None: np.asarray(x, np.float64) @ np.asarray(p["kernel"], np.float64) + np.asarray(p["bias"], np.float64))
        row = {"weights": name}
        for mode in MODES:
            rgb = render(tree, table, mc, o, d, jitter, make_dense(mode))
            row[mode] = float(np.abs(rgb - exact).max())
        results.append(row)
        print(row)
    out = {"what": "max |dRGB| against float64 over 192 rays x 128 samples (refractive sphere grid 64^3), NerfMLP products emulated per mode",
           "mfma_ticks_per_K64_product": {"f16x3": 384, "f16x2": 256, "f16+fp8x2": 256, "f16+fp6x2": 192, "f16+fp4x2": 192}, "results": results}
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
