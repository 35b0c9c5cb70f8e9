#!/usr/bin/env python3
"""Gradient error of NerfMLP backward arithmetics against float64 — including one that does not exist yet (DESIGN NOTE, CPU only).

Emulates the operand roundings of the dgrad chain (dX = dY W^T per layer) and of the wgrad (dW = X^T dY) for
    f32     hi + lo f16 parts of both operands, 3 products            (the shipped default: RNERF_BWD_F16X2)
    tf32    dgrad: exact weights x f16(dY); wgrad: f16(X) x f16(dY)   (RNERF_BWD_F16)
    fp8lo   f16 main term + both cross terms on block-scaled e4m3 operands (one power-of-two scale per 32 elements of the reduction
            axis): what a "lo planes stored as fp8" backward would compute (1/3 less matrix time, 1/4 less HBM traffic)
in float64 products / sums (the MFMA's fp32 accumulation is below what is measured), on the network and cotangent recipe of
tools/r02/bwd_err.py (rows with cotangents orders of magnitude apart), and prints the worst  max|g - g64| / max|g64|  over the 24 tensors.

usage: python tools/fp8_backward_error.py [rows ...]        (default: 581 4096)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import ref_np as R                       # noqa: E402
from samplenerfro_amd import synthetic as syn        # noqa: E402
from fp8_cross_term_error import e4m3_block, f16     # noqa: E402

SHAPES = syn.NERF_MLP_SHAPES


def unflatten(flat):
    out, off = [], 0
    for i, o in SHAPES:
        out.append((flat[off:off + i * o].reshape(i, o).astype(np.float64), flat[off + i * o:off + i * o + o].astype(np.float64)))
        off += i * o + o
    return out


def parts(x):
    h = f16(x)
    return h, f16(x - h)


def prod(a, b, mode, kaxis_a, kaxis_b):
    """a @ b with the operand roundings of `mode`; k axes given for the fp8 block scales."""
    if mode == "f64":
        return a @ b
    ah, al = parts(a); bh, bl = parts(b)
    if mode == "f32":
        return ah @ bh + ah @ bl + al @ bh
    if mode == "fp8lo":
        return ah @ bh + e4m3_block(ah, kaxis_a) @ e4m3_block(bl, kaxis_b) + e4m3_block(al, kaxis_a) @ e4m3_block(bh, kaxis_b)
    raise ValueError(mode)


def dgrad(dy, w, mode):
    """dX = dY W^T (reduction over the layer's outputs)."""
    mode = _split(mode)[0]
    if mode == "tf32":
        return f16(dy) @ w.T                       # exact weights (hi + lo), gradient rounded to f16
    return prod(dy, w.T, mode, -1, 0)


def wgrad(x, dy, m, mode):
    """dW = X^T diag(m) dY (reduction over the rows).  dY is the row-normalised gradient, m the per-row power of two it was divided by:
    like the kernels, the operands are rounded BEFORE the (exact) power-of-two row scale is applied."""
    mode = _split(mode)[1]
    if mode == "f64":
        return (x * m).T @ dy
    xh, xl = parts(x); dh, dl = parts(dy)
    if mode == "xy":
        return (xh * m).T @ dh + (xl * m).T @ dh
    if mode == "yx":
        return (xh * m).T @ dh + (xh * m).T @ dl
    if mode == "tf32":
        return (xh * m).T @ dh
    if mode == "f32":
        return (xh * m).T @ dh + (xh * m).T @ dl + (xl * m).T @ dh
    q = lambda a: e4m3_block(a, 0)                  # blocks of 32 consecutive ROWS (the reduction axis) share a scale
    return (xh * m).T @ dh + (q(xh) * m).T @ q(dl) + (q(xl) * m).T @ q(dh)


def _split(mode):
    """mode "f32" / "tf32" / "fp8lo", or a mixed "<dgrad>+<wgrad>" pair (round 4: which HALF of the backward needs which arithmetic?):
    dgrad in {f32, tf32 (exact weights x f16(dY): 2 passes), fp8lo}, wgrad in {f32, tf32, fp8lo, xy (X_hi + X_lo) x dY_hi, yx X_hi x (dY_hi + dY_lo)}."""
    return mode.split("+") if "+" in mode else (mode, mode)


def backward(P, enc, venc, cot, mode):
    """Manual forward (float64) + backward of NerfMLP (rnerf/model_utils.py:30-90) with the GEMMs of the backward in `mode`.
    Rows are normalised by a power of two of their largest cotangent like the kernels do (exact; keeps f16 / e4m3 in range)."""
    X, Z = {}, {}
    x = enc
    for l in range(8):
        X[l] = x
        z = x @ P[l][0] + P[l][1]; Z[l] = z
        x = np.maximum(z, 0)
        if l == 4:
            x = np.concatenate([x, enc], -1)
    X[8] = x; X[9] = x
    bott = x @ P[9][0] + P[9][1]
    X[10] = np.concatenate([bott, venc], -1)
    Z[10] = X[10] @ P[10][0] + P[10][1]
    X[11] = np.maximum(Z[10], 0)
    m = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(cot).max(-1, keepdims=True), 1e-300))) - 5)
    cot = np.where(m > 0, cot / m, 0.0)            # normalised cotangents; un-normalised at the wgrad (exact powers of two)
    g = [None] * 12
    dy11 = cot[:, :3]; dy8 = cot[:, 3:4]
    g[11] = (wgrad(X[11], dy11, m, mode), (dy11 * m).sum(0))
    g[8] = (wgrad(X[8], dy8, m, mode), (dy8 * m).sum(0))
    dy10 = dgrad(dy11, P[11][0], mode) * (Z[10] > 0)
    g[10] = (wgrad(X[10], dy10, m, mode), (dy10 * m).sum(0))
    dy9 = dgrad(dy10, P[10][0], mode)[:, :256]
    g[9] = (wgrad(X[9], dy9, m, mode), (dy9 * m).sum(0))
    dx = dgrad(dy9, P[9][0], mode) + dy8 @ P[8][0].T
    for l in range(7, -1, -1):
        dyl = dx[:, :256] * (Z[l] > 0) if l != 4 else dx[:, :256] * (Z[l] > 0)
        g[l] = (wgrad(X[l], dyl, m, mode), (dyl * m).sum(0))
        if l > 0:
            dx = dgrad(dyl, P[l][0], mode)
            if l == 5:
                dx = dx[:, :256]                   # the skip concat's encoding columns take no gradient (radiance stages)
    return g


MODES = ("f32", "fp8lo", "tf32", "tf32+f32", "fp8lo+f32", "f32+fp8lo", "f32+xy", "f32+yx", "f32+tf32")


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [581, 4096]
    pf = syn.init_params_flat(12, fine=False, bias_scale=0.1)["coarse_mlp"]
    P = unflatten(pf)
    for rows in rows_list:
        rng = np.random.default_rng(9)
        pos = rng.uniform(-3, 3, (rows, 3)); dirs = R.safe_l2_normalize(rng.standard_normal((rows, 3)))
        enc = R.pos_enc(pos, 0, 10, np.float64); venc = R.pos_enc(dirs, 0, 4, np.float64)
        cot = rng.standard_normal((rows, 4)) * np.array([1e-3, 1e-3, 1e-3, 3e-4])
        cot[5::83] = 0.0; cot[6::83] *= 1e-4; cot[7::83] *= 1e3          # rows whose gradients are orders of magnitude apart (bwd_err.py)
        ref = backward(P, enc, venc, cot, "f64")
        for mode in MODES:
            got = backward(P, enc, venc, cot, mode)
            worst, where = 0.0, ""
            for l in range(12):
                e = np.abs(got[l][0] - ref[l][0]).max() / np.abs(ref[l][0]).max()
                if e > worst:
                    worst, where = e, f"Dense_{l}.kernel"
            print(f"rows {rows:6d}  {mode:10s} worst max|g - g64| / max|g64| = {worst:.2e}  ({where})")


if __name__ == "__main__":
    main()
