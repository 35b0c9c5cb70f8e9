"""Decode the dgrad's dY planes and compare with torch float64 (debug helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_np as R, torch_ref as TR
from samplenerfro_amd import _lib, ops, synthetic as syn
F32 = np.float32
B, S = 83, 7
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
rng = np.random.default_rng(9)
pf = syn.init_params_flat(12, fine=False, bias_scale=0.1)["coarse_mlp"]
pos = rng.uniform(-3, 3, (B, S, 3)).astype(F32)
dirs = R.safe_l2_normalize(rng.standard_normal((B, S, 3)).astype(F32))
pd = np.concatenate([pos, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2)
dr = np.concatenate([dirs, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2)
cot = (rng.standard_normal((S, B, 4)) * np.array([1e-3, 1e-3, 1e-3, 3e-4])).astype(F32)
flat = torch.tensor(pf, dtype=torch.float64)
ps, off = [], 0
for i, o in TR.NERF_MLP_SHAPES:
    ps.append((flat[off:off + i * o].view(i, o), flat[off + i * o:off + i * o + o])); off += i * o + o
x = torch.tensor(R.pos_enc(pos.transpose(1, 0, 2).reshape(-1, 3), 0, 10), dtype=torch.float64)
cond = torch.tensor(R.pos_enc(dirs.transpose(1, 0, 2).reshape(-1, 3), 0, 4), dtype=torch.float64)
pre = []
h = x
for i in range(8):
    z = h @ ps[i][0] + ps[i][1]; z.requires_grad_(True); z.retain_grad(); pre.append(z)
    h = torch.relu(z)
    if i == 4: h = torch.cat([h, x], -1)
sigma = h @ ps[8][0] + ps[8][1]
bott = h @ ps[9][0] + ps[9][1]; bott.retain_grad() if bott.requires_grad else None
zv = torch.cat([bott, cond], -1) @ ps[10][0] + ps[10][1]
v = torch.relu(zv)
rgb = v @ ps[11][0] + ps[11][1]
out = torch.cat([rgb, sigma], -1)
c = torch.tensor(cot.reshape(-1, 4), dtype=torch.float64)
grads = torch.autograd.grad((out * c).sum(), pre + [bott, zv])
ref = {l: grads[l].numpy() for l in range(8)}; ref[8] = grads[8].numpy(); ref[9] = grads[9].numpy()
P = _lib.PRECISIONS["f16x3"]
flat_d = T(pf); packed = ops.nerfmlp_pack(flat_d, P)
rows = S * B; Rp = (rows + 255) // 256 * 256
for bwd in ("bf16", "tf32", "f32"):
    BW = _lib.BACKWARDS[bwd]
    raw, save = ops.nerfmlp_forward_train(packed, P, T(pd.astype(F32)), T(dr.astype(F32)), None, S, B, BW)
    dy = ops.nerfmlp_backward(ops.nerfmlp_pack_bwd(flat_d, None, BW), packed, P, save, T(cot), rows, stages="d", backward=BW)
    raw_b = dy.cpu().numpy()
    NPp = 2 if bwd == "f32" else 1
    plane = 153 * Rp * 2 * 16
    def decode(p):
        a = raw_b[p * plane:(p + 1) * plane].view(np.uint16).reshape(153, Rp, 2, 8)
        if bwd == "bf16":
            return (a.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        return a.view(np.float16).astype(np.float64)
    val = decode(0) + (decode(1) if NPp == 2 else 0)
    if bwd != "bf16":
        rs = raw_b[NPp * plane:NPp * plane + 4 * Rp].view(np.float32).astype(np.float64)
        val = val * rs[None, :, None, None]
        print("m_ref", raw_b[NPp * plane + 4 * Rp:NPp * plane + 4 * Rp + 4].view(np.float32), "row scales", rs[:4], rs[rows - 2:rows + 2])
    line = []
    for l in range(10):
        nk = 8 if l == 9 else 16
        slot0 = 144 if l == 9 else 16 * l
        got = np.zeros((rows, nk * 16))
        for s in range(nk):
            for hh in range(2):
                for j in range(8):
                    got[:, 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)] = val[slot0 + s, :rows, hh, j]
        r = ref[l]
        line.append(f"L{l} {np.abs(got - r).max() / np.abs(r).max():.1e}")
    print(f"[{bwd}]", " ".join(line))
    if bwd == "tf32":
        l = 9; nk = 8; slot0 = 144
        got = np.zeros((rows, nk * 16))
        for s in range(nk):
            for hh in range(2):
                for j in range(8):
                    got[:, 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)] = val[slot0 + s, :rows, hh, j]
        r = ref[9]
        for row in (17, 18, 19, 273):
            nz = np.nonzero(r[row])[0][:6]
            print("row", row, "rs", rs[row], "cot", cot.reshape(-1, 4)[row], "normalised", cot.reshape(-1, 4)[row] / rs[row], "\n   ref", r[row, nz], "got", got[row, nz], "ratio", got[row, nz] / r[row, nz])
        bad = np.abs(got - r).max(1) / np.abs(r).max()
        print("rows with large error:", np.nonzero(bad > 1e-2)[0][:40], "of", rows)
        for row in np.nonzero(bad > 1e-2)[0][:3]:
            e = np.abs(got[row] - r[row]) / np.abs(r).max()
            fb = np.nonzero(e > 1e-3)[0]
            print("bad row", row, "bad features", fb[:40], "\n   got", got[row, fb[:8]], "\n   ref", r[row, fb[:8]], "normalised got", got[row, fb[:8]] / rs[row])
        # where does the wrong value come from?  search the reference (pre-mask = all layers) for the same number
        z9 = grads[9].numpy()
        for row in np.nonzero(bad > 1e-2)[0][:6]:
            v = got[row, 28]
            cand = np.argwhere(np.abs(z9 - v) < 2e-3 * abs(v))
            print("row", row, "got", v, "ref", r[row, 28], "matches in dL/dzv:", cand[:6].tolist())
        hd = val[152, :rows, 0, :4]
        print("heads row0", hd[0], "cot", cot.reshape(-1, 4)[0])
