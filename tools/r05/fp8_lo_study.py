"""What would fp8 lo planes cost the weight gradient?  (a numerical study on the host, no kernel.)
dW = X^T dY over R rows for one 256 x 256 layer; X = relu activations, dY = row-normalised gradients (|dY| <= 1 per row).
Arithmetics compared with the float64 product:
  f16x3      : (Xh + Xl)(dYh + dYl) without lo x lo, lo = f16                               — today's default (22 bits)
  f16 + fp8  : Xh dYh + q8(Xh) q8(dYl) + q8(Xl) q8(dYh), q8 = e4m3 with a per-row power-of-two scale for the lo planes — 3 bytes per element
               instead of 4; the cross terms are fp8 x fp8 (what v_mfma_f32_32x32x64_f8f6f4 / ..._fp8_fp8 can issue)
  f16 + int8 : the same with block-scaled int8 (one power-of-two scale per row and 32 features) for every cross-term operand (v_mfma_i32_32x32x32_i8;
               the group scales would have to be applied per 32-wide k-slice: a sketch of the arithmetic, not of a kernel)
  f16        : Xh dYh                                                                       — the single-plane backward
usage: python tools/r05/fp8_lo_study.py [rows]"""
import sys
import numpy as np
import torch
torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = N = 256
X = torch.relu(torch.randn(R, K, dtype=torch.float64) * 0.7 + 0.1)
dY = torch.randn(R, N, dtype=torch.float64) * torch.exp(torch.randn(R, 1, dtype=torch.float64))        # rows of very different size
m = 2.0 ** torch.ceil(torch.log2(dY.abs().amax(1, keepdim=True)))                                         # the dgrad's row normalisation
dYn = dY / m
exact = (X * m).T @ dYn
f16 = lambda t: t.to(torch.float16).to(torch.float64)
def q8(t, per_row_scale=True):
    s = 2.0 ** torch.floor(torch.log2(448.0 / t.abs().amax(1, keepdim=True).clamp_min(1e-300))) if per_row_scale else 1.0
    return (t * s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64) / s
Xh, dYh = f16(X), f16(dYn)
Xl, dYl = f16(X - Xh), f16(dYn - dYh)
res = {}
res["f16x3 (f16 lo planes)"] = ((Xh * m).T @ dYh) + ((Xh * m).T @ dYl) + ((Xl * m).T @ dYh)
res["f16 hi + fp8 lo, cross terms fp8 x fp8"] = ((Xh * m).T @ dYh) + ((q8(Xh) * m).T @ q8(dYn - dYh)) + ((q8(X - Xh) * m).T @ q8(dYh))
res["f16 hi + fp8 lo, cross terms f16 x fp8 (no such MFMA)"] = ((Xh * m).T @ dYh) + ((Xh * m).T @ q8(dYn - dYh)) + ((q8(X - Xh) * m).T @ dYh)
def i8(t, group=32):                      # int8 with one power-of-two scale per (row, group of 32 features): 7 bits for the group's largest entry
    r, c = t.shape
    g = t.reshape(r, c // group, group)
    s = 2.0 ** torch.floor(torch.log2(127.0 / g.abs().amax(2, keepdim=True).clamp_min(1e-300)))
    return (torch.round(g * s).clamp(-127, 127) / s).reshape(r, c)
res["f16 hi + int8 lo (block-scaled), cross terms int8 x int8"] = ((Xh * m).T @ dYh) + ((i8(Xh) * m).T @ i8(dYn - dYh)) + ((i8(X - Xh) * m).T @ i8(dYh))
res["f16 single plane"] = (Xh * m).T @ dYh
print(f"rows {R}, one 256 x 256 layer; error of dW relative to max |dW|:")
for k, v in res.items():
    print(f"  {k:58s} {float((v - exact).abs().max() / exact.abs().max()):.2e}")
