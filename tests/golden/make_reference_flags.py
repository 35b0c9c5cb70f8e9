"""The reference's flag defaults and shipped config overrides as DATA: every `flags.DEFINE_*(name, default, help)` call of
rnerf/utils.py:define_flags (read with `ast`, literal defaults only — nothing is imported or executed), every configs/*.yaml and the
bindings of every configs/*.gin, written to
tests/golden/reference_flags.json.  tests/test_reference_flags.py holds samplenerfro_amd.utils.default_flags and the bench / test
workloads' overrides to them.  usage: python tests/golden/make_reference_flags.py"""
import ast
import glob
import hashlib
import json
import os

REF = os.environ.get("RNERF_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_flags.json")


def extract():
    src = os.path.join(REF, "rnerf", "utils.py")
    if not os.path.exists(src):
        return None
    text = open(src).read()
    tree = ast.parse(text, src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "define_flags")
    defaults = {}
    for call in (n for n in ast.walk(fn) if isinstance(n, ast.Call)):
        f = call.func
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == "flags" and f.attr.startswith("DEFINE_") and len(call.args) >= 2:
            name = ast.literal_eval(call.args[0])
            a = call.args[1]
            try:
                defaults[name] = {"kind": f.attr[len("DEFINE_"):], "default": ast.literal_eval(a)}
            except ValueError:
                defaults[name] = {"kind": f.attr[len("DEFINE_"):], "default": None, "expr": ast.unparse(a)}      # a computed default: kept as text, not compared
    import yaml
    configs = {}
    for p in sorted(glob.glob(os.path.join(REF, "configs", "*.yaml"))):
        configs[os.path.basename(p)[:-5]] = yaml.safe_load(open(p))
    gin = {}                      # `Scope.name = literal` bindings of configs/*.gin (comments dropped), as data
    for p in sorted(glob.glob(os.path.join(REF, "configs", "*.gin"))):
        b = {}
        for line in open(p):
            line = line.split("#", 1)[0].strip()
            if "=" in line:
                k, v = (t.strip() for t in line.split("=", 1))
                try:
                    b[k] = ast.literal_eval(v)
                except (ValueError, SyntaxError):
                    b[k] = v
        gin[os.path.basename(p)[:-4]] = b
    return {"source_sha256": hashlib.sha256(text.encode()).hexdigest(), "defaults": defaults, "configs": configs, "gin": gin}


def main():
    d = extract()
    if d is None:
        print("SKIPPED: the reference is not on this machine")
        return
    json.dump(d, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(d['defaults'])} flags, {len(d['configs'])} configs")


if __name__ == "__main__":
    main()
