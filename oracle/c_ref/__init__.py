"""ctypes access to oracle/c_ref/march_ref.c (TEST INFRASTRUCTURE: the scalar C restatement of the table / lookup / march, second reading
beside oracle/ref_np.py).  build() compiles it with gcc into oracle/c_ref/libmarch_ref.so (git-ignored); load() builds on demand."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "march_ref.c")
LIB = os.path.join(HERE, "libmarch_ref.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        gcc = shutil.which("gcc") or shutil.which("cc")
        if gcc is None:
            raise RuntimeError("gcc not found: the C restatement of the oracle cannot be built")
        # -ffp-contract=off: no FMA contraction; no -ffast-math; SSE scalar float arithmetic on x86-64 = one IEEE operation per C operation
        subprocess.check_call([gcc, "-O2", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _i3(v):
    return (C.c_int32 * 3)(*[int(x) for x in v])


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_table(grid, ndim, nmin, nmax):
    g = np.ascontiguousarray(np.asarray(grid, np.float32).reshape(-1))
    out = np.empty((g.size, 4), np.float32)
    load().rnerf_ref_build_table(_p(g), _i3(ndim), _d3(nmin), _d3(nmax), _p(out))
    return out


def linear3(table, pts, ndim, nmin, nmax):
    t = np.ascontiguousarray(table, np.float32)
    p = np.ascontiguousarray(np.asarray(pts, np.float32).reshape(-1, 3))
    out = np.empty((p.shape[0], 4), np.float32); idx = np.empty((p.shape[0], 6), np.int32)
    load().rnerf_ref_linear3(_p(t), _i3(ndim), _d3(nmin), _d3(nmax), _p(p), C.c_int64(p.shape[0]), _p(out), _p(idx))
    return out, idx


def path_sampler(origins, viewdirs, table, ndim, nmin, nmax, near, far, num_samples):
    """-> (pos [B,N,3], dir [B,N,3], dist [B,N], ior [B,N,1], grad [B,N,3], vox [B,N,6]) like oracle.ref_np.path_sampler(..., return_idx=True)."""
    t = np.ascontiguousarray(table, np.float32)
    o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(viewdirs, np.float32)
    B, N = o.shape[0], int(num_samples)
    pos = np.empty((B, N, 3), np.float32); dr = np.empty((B, N, 3), np.float32); dist = np.empty((B, N), np.float32)
    ior = np.empty((B, N, 1), np.float32); grad = np.empty((B, N, 3), np.float32); vox = np.empty((B, N, 6), np.int32)
    load().rnerf_ref_path_sampler(_p(t), _i3(ndim), _d3(nmin), _d3(nmax), _p(o), _p(d), C.c_int32(B), C.c_double(near), C.c_double(far), C.c_int32(N),
                                  _p(pos), _p(dr), _p(dist), _p(ior), _p(grad), _p(vox))
    return pos, dr, dist, ior, grad, vox


def sample_pdf(u, bins, weights, origins, directions, z_vals, jitter):
    """-> (z [B,S+F], pos [B,S+F,3], dir [B,S+F,3], idx int32 [B,S+F]) like oracle.ref_np.sample_pdf (without the gradient gather)."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    u, bins, weights, o, d, zv = f(u), f(bins), f(weights), f(origins), f(directions), f(z_vals)
    j = np.ascontiguousarray(jitter, np.int32)
    B, F = u.shape
    nb, N, S = weights.shape[1], zv.shape[1], j.shape[0]
    assert bins.shape == (B, nb + 1) and o.shape == (B, N, 3)
    z = np.empty((B, S + F), np.float32); pos = np.empty((B, S + F, 3), np.float32); dr = np.empty((B, S + F, 3), np.float32)
    idx = np.empty((B, S + F), np.int32); scratch = np.empty(2 * (nb + 1) + 2, np.float32)
    load().rnerf_ref_sample_pdf(_p(u), _p(bins), _p(weights), C.c_int32(B), C.c_int32(nb), C.c_int32(F), _p(zv), _p(o), _p(d), C.c_int32(N), _p(j),
                                C.c_int32(S), _p(z), _p(pos), _p(dr), _p(idx), _p(scratch))
    return z, pos, dr, idx
