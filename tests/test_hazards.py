"""The device code of the MLP engines must be free of MFMA <-> inline-asm VALU hazards (tools/hazard_check.py).

LLVM inserts the wait states gfx90a+ needs between a matrix instruction and a dependent VALU instruction only for instructions it knows
to be VALU; the one-line `asm("v_...")` statements the kernels place in MFMA shadows are opaque to it.  Round 2 met wrong values in a few
lanes, differently from run to run, in one schedule of the f16 dgrad kernel (SLP vectoriser on).  This test compiles csrc/mlp.hip with the
product flags (hipcc cross-compiles, no GPU) and scans the emitted ISA for the two hazard patterns; tests/test_gpu_backward.py checks on
the device that repeated runs of the backward kernels are bit-identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_mfma_inline_asm_hazards_in_the_product_build(tmp_path):
    import hazard_check as H
    src = os.path.join(ROOT, "samplenerfro_amd", "csrc", "mlp.hip")
    asm = H.compile_to_asm(src, [])
    try:
        funcs = H.parse(asm)
    finally:
        os.unlink(asm)
    n_asm = sum(1 for c in funcs.values() for ins in c if ins[2] and ins[0].startswith("v_"))
    n_mfma = sum(1 for c in funcs.values() for ins in c if ins[0].startswith("v_mfma"))
    assert n_mfma > 10000 and n_asm > 3000          # the scan saw the engines (forward, dgrad, wgrad) and their asm statements
    found = H.check(funcs)
    assert not found, found[:10]


def test_the_checker_flags_both_patterns():
    import hazard_check as H
    mk = lambda lines: {"k": [(l.split()[0], [o.strip() for o in l.split(None, 1)[1].split(",")], a, i) for i, (l, a) in enumerate(lines)]}
    # H1: asm VALU result read by an MFMA one wait state later
    f = H.check(mk([("v_fmac_f32_e32 v5, v1, v2", True), ("s_nop 0", False), ("v_mfma_f32_32x32x16_f16 a[0:15], v[4:7], v[8:11], a[0:15]", False)]))
    assert [x[0] for x in f] == ["H1"]
    f = H.check(mk([("v_fmac_f32_e32 v5, v1, v2", True), ("s_nop 1", False), ("v_mfma_f32_32x32x16_f16 a[0:15], v[4:7], v[8:11], a[0:15]", False)]))
    assert f == []
    # H2: asm VALU reading an MFMA's VGPR result inside the 8-pass window (11 wait states)
    f = H.check(mk([("v_mfma_f32_32x32x16_f16 v[0:15], v[20:23], v[24:27], v[0:15]", False), ("s_nop 7", False), ("v_fma_mix_f32 v30, v3, v31, v32", True)]))
    assert [x[0] for x in f] == ["H2"]
    f = H.check(mk([("v_mfma_f32_32x32x16_f16 v[0:15], v[20:23], v[24:27], v[0:15]", False), ("s_nop 7", False), ("s_nop 3", False),
                    ("v_fma_mix_f32 v30, v3, v31, v32", True)]))
    assert f == []
    # the incident of round 4 (csrc/bkgd16.hip, never shipped): a three-instruction f16 hi / lo split as inline asm whose last partial-register
    # write (v_fma_mixhi_f16) sat right in front of the MFMA that read the word — NaN outputs on the device, bit-correct in isolation
    f = H.check(mk([("v_cvt_pk_f16_f32 v30, v122, v123", True), ("v_fma_mixlo_f16 v33, v141, -1.0, v148 op_sel_hi:[1,0,0]", True),
                    ("v_fma_mixhi_f16 v33, v141, -1.0, v149 op_sel:[1,0,0] op_sel_hi:[1,0,0]", True),
                    ("v_mfma_f32_32x32x16_f16 a[0:15], v[138:141], v[30:33], a[0:15]", False)]))
    assert f and all(x[0] == "H1" for x in f)


def test_no_hazards_in_the_background_mlp_build():
    """csrc/bkgd16.hip (f16 hi + lo background MLP): MFMAs but, since the incident above, no inline asm — the scan must stay clean if any returns."""
    import hazard_check as H
    asm = H.compile_to_asm(os.path.join(ROOT, "samplenerfro_amd", "csrc", "bkgd16.hip"), [])
    try:
        funcs = H.parse(asm)
    finally:
        os.unlink(asm)
    assert sum(1 for c in funcs.values() for ins in c if ins[0].startswith("v_mfma")) > 300
    assert not H.check(funcs)

