"""The C-ABI library builds for gfx950 (no GPU needed) and exports every symbol include/rnerf.h declares."""
import ctypes
import os
import re

from samplenerfro_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rnerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rnerf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib_path):
    syms = declared_symbols()
    assert len(syms) >= 13
    lib = ctypes.CDLL(lib_path)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rnerf.h but not exported"


def test_binding_covers_header(lib_path):
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.rnerf_version() == 3
    stream3 = 1160 * 2 * 1024 + 3468 * 4      # a three-pass operand stream + aux floats (biases, heads, the zero block, the range flag)
    al = lambda n: (n + 255) // 256 * 256
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_BF16X3) == stream3
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_BF16) == 1160 * 1024 + 3468 * 4
    # every f16-based evaluation buffer ends with the bf16x3 stream of the range-safe second pass
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_F16X3) == al(stream3) + stream3
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_F16X2) == al(stream3) + stream3
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_F16F8) == al(stream3) + al(stream3) + stream3      # its own stream, the f16x3 stream it falls back to, bf16x3
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_F16) == al(1160 * 1024 + 3468 * 4) + stream3            # the single-pass stream + bf16x3
    assert lib.rnerf_nerfmlp_packed_bytes(_lib.PREC_F32) == 595844 * 4                            # the exact-fp32 arbiter reads the flat buffer itself
    # the Python names follow enum rnerf_precision of the header
    hdr = open(os.path.join(ROOT, "include", "rnerf.h")).read()
    for name, val in _lib.PRECISIONS.items():
        assert re.search(r"RNERF_PREC_%s\s*=\s*%d\b" % (name.upper(), val), hdr), name
    assert lib.rnerf_nerfmlp_packed_bytes(99) == 0
    assert b"precision" in lib.rnerf_last_error()


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(_lib.RnerfError):
        _lib.load(str(tmp_path / "nope.so"))


def test_argument_errors_do_not_need_a_gpu(lib_path):
    lib = _lib.load()
    g = _lib.Grid.make([1, 8, 8], [-1, -1, -1], [1, 1, 1])
    rc = lib.rnerf_grid_build_table(ctypes.c_void_p(16), ctypes.c_void_p(32), ctypes.byref(g), None)
    assert rc == -1 and b"dims" in lib.rnerf_last_error()
    assert lib.rnerf_march(None, ctypes.byref(g), None, None, 4, 2.0, 6.0, 8, None, None, None, None, None) == -1


def test_whole_path_structs_match_the_header_layout(lib_path):
    """ctypes mirrors of rnerf_model / rnerf_train_cfg: the workspace queries read num_coarse / num_fine / num_path / bd_cut / backward
    / bg_patch_size out of them on the host — the sizes must move exactly as the fields say."""
    lib = _lib.load()
    m = _lib.Model()
    m.num_coarse, m.num_fine, m.num_path = 64, 0, 12
    assert lib.rnerf_forward_workspace_bytes(ctypes.byref(m), 0) == 0
    a = lib.rnerf_forward_workspace_bytes(ctypes.byref(m), 512)
    path = 2 * 64 * 12 * 512 * 16
    assert path < a < path + 64 * 512 * 24 + 512 * 64
    m.num_fine = 128
    b = lib.rnerf_forward_workspace_bytes(ctypes.byref(m), 512)
    assert b - a >= 192 * 512 * (16 + 16 + 16 + 4)                       # fine rows (pd, dr), raw, merged depths
    m.bd_cut = 1
    assert lib.rnerf_forward_workspace_bytes(ctypes.byref(m), 512) - b >= 2 * 9 * 512 * 4
    c = _lib.TrainCfg()
    c.backward = _lib.BWD_F16
    t1 = lib.rnerf_train_workspace_bytes(ctypes.byref(m), ctypes.byref(c), 512)
    c.backward = _lib.BWD_F16X2
    t2 = lib.rnerf_train_workspace_bytes(ctypes.byref(m), ctypes.byref(c), 512)
    assert t2 > t1 > b                                                   # hi + lo saves / dY planes
    c.bg_smooth_weight, c.bg_patch_size = 1.0, 128
    assert lib.rnerf_train_workspace_bytes(ctypes.byref(m), ctypes.byref(c), 512) > t2
    rc = lib.rnerf_adam_update(None, None, None, None, None, 0, None, 0, None, None, None)
    assert rc == -1 and b"rnerf_adam_update" in lib.rnerf_last_error()


def test_the_product_library_reads_no_environment(lib_path):
    """Experiment switches compile in only under -DRNERF_EXPERIMENTS (csrc/common.h RNERF_ENV): librnerf.so must not even import getenv,
    librnerf_experiments.so (same sources) does; no source file calls getenv() directly."""
    import glob
    import subprocess
    from samplenerfro_amd import build
    und = lambda p: subprocess.run(["nm", "-D", "--undefined-only", p], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und(lib_path)
    assert os.path.exists(build.LIB_EXPERIMENTS) and "getenv" in und(build.LIB_EXPERIMENTS)
    for src in glob.glob(os.path.join(ROOT, "samplenerfro_amd", "csrc", "**", "*.*"), recursive=True):
        text = re.sub(r"//.*", "", open(src).read())
        if not src.endswith("common.h"):
            assert not re.search(r"(?<![A-Z_])getenv\s*\(", text), src
    for py in glob.glob(os.path.join(ROOT, "samplenerfro_amd", "*.py")):        # the host layer: only the launcher's variables (distributed.py)
        for m in re.finditer(r"environ(?:\.get|\.setdefault)?\s*[\(\[]\s*[\"']([A-Z_0-9]+)", open(py).read()):
            assert m.group(1) in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"), (py, m.group(1))


def test_a_library_of_another_abi_version_is_refused(lib_path, monkeypatch):
    import pytest
    monkeypatch.setattr(_lib, "ABI_VERSION", 1)
    with pytest.raises(_lib.RnerfError, match="ABI version"):
        _lib.load(lib_path)
