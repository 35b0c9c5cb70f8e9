#!/usr/bin/env python3
"""Static check of the MFMA <-> inline-asm VALU data hazards in the device code of csrc/*.hip (gfx950).

Why: LLVM's hazard recognizer inserts the wait states gfx90a+ needs between a matrix instruction and a VALU instruction that touches
the same registers (GCNHazardRecognizer::checkMAIHazards90A / checkMAIVALUHazards) only for instructions it KNOWS to be VALU.  An
`asm("v_fmac_f32 ...")` statement is an opaque INLINEASM node to it: no wait states are inserted around it.  The kernels use one-line
asm VALU instructions in MFMA shadows (v_fma_mix_f32, v_bfe_i32, v_fmac_f32, v_pk_min_u16, v_max_f32, v_cvt_pk_f16_f32), so whether a
hazard exists depends on where the scheduler happens to put them.  Round 2 saw wrong values "in a few lanes, differently from run to run"
in the f16 dgrad kernel with the SLP vectoriser on; this tool checks every build for the two patterns that can produce that:

  H1  asm VALU writes vN, and an MFMA reads vN as SrcA/SrcB/SrcC fewer than 2 wait states later
      (checkMAIHazards90A: "VALU writes VGPR -> MFMA read": 2 wait states);
  H2  an MFMA writes v[a:b] (or a[a:b]), and an asm VALU reads or writes a register of that range fewer than PASSES + 3 wait states
      later (checkMAIVALUHazards: "XDL write VGPR -> VALU read/write": 5 / 7 / 11 / 19 wait states for 2 / 4 / 8 / 16 passes).

Every instruction counts one wait state, `s_nop N` counts N + 1.  The scan is linear per function (basic-block order of the .s file, branch
targets ignored): a conservative approximation, good for the straight-line MFMA loops these kernels consist of.

usage: python tools/hazard_check.py [--flags "<extra hipcc flags>"] [file.hip ...]      (default: csrc/mlp.hip with the product flags)
exit code 1 if a hazard is found.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
PASSES = {"32x32x16": 8, "32x32x8": 16, "32x32x2": 16, "32x32x4": 16, "16x16x32": 4, "16x16x16": 8, "16x16x4": 8, "4x4x4": 2, "32x32x1": 16, "16x16x1": 8}


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def parse(path):
    """-> {function: [(mnemonic, [operand strings], in_asm, line_no)]}"""
    funcs, cur, in_asm = {}, None, False
    for no, ln in enumerate(open(path), 1):
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
        if cur is None or not t:
            continue
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((parts[0], ops, in_asm, no))
    return funcs


def wait_states(ins):
    if ins[0] == "s_nop":
        try:
            return int(ins[1][0], 0) + 1
        except Exception:
            return 1
    return 1


def check(funcs):
    findings = []
    for fn, code in funcs.items():
        for i, (mn, ops, in_asm, no) in enumerate(code):
            is_asm_valu = in_asm and mn.startswith("v_") and not mn.startswith("v_mfma")
            if is_asm_valu and ops:
                dst = regs(ops[0])
                # H1: an MFMA reads the asm's result too early
                ws = 0
                for (mn2, ops2, _, no2) in code[i + 1:i + 4]:
                    if ws >= 2:
                        break
                    if mn2.startswith("v_mfma") and any(regs(o) & dst for o in ops2[1:]):
                        findings.append(("H1", fn, no, mn, no2, mn2, ws))
                    ws += wait_states((mn2, ops2))
            if mn.startswith("v_mfma") and ops:
                shape = re.search(r"(\d+x\d+x\d+)", mn)
                need = PASSES.get(shape.group(1), 16) + 3 if shape else 19
                dst = regs(ops[0])
                ws = 0
                for (mn2, ops2, asm2, no2) in code[i + 1:i + 1 + need]:
                    if ws >= need:
                        break
                    if asm2 and mn2.startswith("v_") and not mn2.startswith("v_mfma") and any(regs(o) & dst for o in ops2):
                        findings.append(("H2", fn, no, mn, no2, mn2, ws))
                    ws += wait_states((mn2, ops2))
    return findings


def compile_to_asm(src, extra):
    from samplenerfro_amd import build
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [build._hipcc()] + build.FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def main():
    args = sys.argv[1:]
    extra = []
    if args and args[0] == "--flags":
        extra = args[1].split()
        args = args[2:]
    srcs = args or [os.path.join(ROOT, "samplenerfro_amd", "csrc", "mlp.hip")]
    bad = 0
    for src in srcs:
        asm = src if src.endswith(".s") else compile_to_asm(src, extra)
        funcs = parse(asm)
        n_asm = sum(1 for c in funcs.values() for ins in c if ins[2] and ins[0].startswith("v_"))
        n_mfma = sum(1 for c in funcs.values() for ins in c if ins[0].startswith("v_mfma"))
        f = check(funcs)
        print(f"{os.path.basename(src)}: {len(funcs)} kernels, {n_mfma} MFMAs, {n_asm} inline-asm VALU instructions, {len(f)} hazard candidates"
              + (f" (extra flags: {' '.join(extra)})" if extra else ""))
        for kind, fn, no, mn, no2, mn2, ws in f[:40]:
            print(f"  {kind} {fn[:60]}: line {no} {mn} -> line {no2} {mn2} after {ws} wait states")
        bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
