#!/usr/bin/env python3
"""Which (lane, byte) of the A / B operands of v_mfma_scale_f32_32x32x64_f8f6f4 is which (row, k) / (k, column), and what the per-lane scale
byte multiplies — probed on the device with one-hot operands (groundwork for the "f16 main term + fp8 cross terms" engine).
usage (GPU box): hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/f8f6f4_probe.hip -o /tmp/libf8probe.so && python tools/ubench/f8f6f4_probe.py /tmp/libf8probe.so"""
import ctypes, sys
import numpy as np
import torch

lib = ctypes.CDLL(sys.argv[1])
ONE = 0x38                       # 1.0 in e4m3 (bias 7)
dev = "cuda:0"


def run(a, b, sa=None, sb=None):
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    s1 = torch.full((64,), 0x7F, dtype=torch.int32, device=dev) if sa is None else torch.from_numpy(sa).to(dev)
    s2 = torch.full((64,), 0x7F, dtype=torch.int32, device=dev) if sb is None else torch.from_numpy(sb).to(dev)
    d = torch.zeros((64, 16), dtype=torch.float32, device=dev)
    rc = lib.f8_probe(ctypes.c_void_p(ta.data_ptr()), ctypes.c_void_p(tb.data_ptr()), ctypes.c_void_p(s1.data_ptr()), ctypes.c_void_p(s2.data_ptr()),
                      ctypes.c_void_p(d.data_ptr()))
    assert rc == 0
    acc = d.cpu().numpy()
    D = np.zeros((32, 32), np.float32)                 # accumulator layout of the 32x32 f32 result: row = (r&3) + 8 (r>>2) + 4 (lane>>5), col = lane & 31
    for l in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = acc[l, r]
    return D


ones = np.full((64, 32), ONE, np.uint8)
zeros = np.zeros((64, 32), np.uint8)
D = run(ones, ones)
print("all ones: D == 64 everywhere:", bool(np.all(D == 64.0)))
# rows of A / columns of B
rowA, colB = {}, {}
for L in (0, 1, 31, 32, 33, 63):
    a = zeros.copy(); a[L, 0] = ONE
    D = run(a, ones); r = np.nonzero(D.sum(1))[0]; rowA[L] = r.tolist()
    b = zeros.copy(); b[L, 0] = ONE
    D = run(ones, b); c = np.nonzero(D.sum(0))[0]; colB[L] = c.tolist()
print("row of A's lane L:", rowA)
print("column of B's lane L:", colB)
# k classes: A one-hot at (lane 0 / 32, byte p) against B one-hot at (lane 0 / 32, byte p'): non-zero iff the same k
match = {}
for La in (0, 32):
    for p in range(32):
        a = zeros.copy(); a[La, p] = ONE
        hits = []
        for Lb in (0, 32):
            for q in range(32):
                b = zeros.copy(); b[Lb, q] = ONE
                if run(a, b)[0, 0] != 0:
                    hits.append((Lb, q))
        match[(La, p)] = hits
same = all(match[(La, p)] == [(La, p)] for La in (0, 32) for p in range(32))
print("k of A (lane half, byte p) pairs with B (same half, same byte) only:", same)
if not same:
    for k, v in match.items():
        print("  A", k, "<-> B", v)
# scales: E8M0, 0x7F = 2^0; does lane L's scale byte (bits 7:0 of the scale operand) multiply its own 32 k-elements of its row?
sa = np.full(64, 0x7F, np.int32); sa[0] = 0x80                                       # lane 0 (row 0, first k half): x 2
D = run(ones, ones, sa=sa)
print("scale_a lane 0 = 2^1: row 0 =", D[0, :3], " other rows =", D[1, :3], " (expect 32*2 + 32 = 96 vs 64 if the scale covers that lane's 32 k)")
sa = np.full(64, 0x7F, np.int32); sa[32] = 0x7E
D = run(ones, ones, sa=sa)
print("scale_a lane 32 = 2^-1: row 0 =", D[0, :3], " (expect 32 + 16 = 48)")
sb = np.full(64, 0x7F, np.int32); sb[5] = 0x81
D = run(ones, ones, sb=sb)
print("scale_b lane 5 = 2^2: column 5 =", D[:3, 5], " column 6 =", D[:3, 6], " (expect 32*4 + 32 = 160 vs 64)")
# value check: e4m3 codes
vals = {0x38: 1.0, 0x40: 2.0, 0x30: 0.5, 0x3C: 1.5, 0x7E: 448.0, 0x08: 2.0 ** -6, 0x01: 2.0 ** -9}
for code, want in vals.items():
    a = zeros.copy(); a[0, 0] = code
    b = zeros.copy(); b[0, 0] = ONE
    print(f"code 0x{code:02x}: {run(a, b)[0, 0]} (e4m3 value {want})")
