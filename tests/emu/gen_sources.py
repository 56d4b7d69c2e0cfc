"""TEST INFRASTRUCTURE.  Writes the library's sources (galah_amd/csrc) into tests/emu/build/src in the form the emulated
build compiles: the text is the product's, with the three constructs a host compiler cannot take rewritten --

  extern __shared__ ... NAME[];      ->  a pointer to the emulator's per-workgroup dynamic LDS
  asm volatile(TEXT : out : in : clobbers) with gfx950 instructions  ->  hipemu_gcn::run(TEXT, bindings) (the interpreter
                                          executes the same instruction text; operands are read off the statement)
  "../../include/galah_hip.h"        ->  "galah_hip.h" (found through -I)
  __attribute__((amdgpu_waves_per_eu(..)))  ->  dropped (an occupancy hint)
  __launch_bounds__(EXPR) ... NAME(...) {  ->  the same, and `hipemu::check_launch_bound((EXPR), "NAME");` as the body's first statement:
                                          a launch with more work-items per workgroup than the kernel was compiled for fails on the GPU
                                          (the bound may depend on template parameters, hence a check inside the kernel)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "galah_amd", "csrc")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "src")   # (next to this script: a copy of tests/emu elsewhere under tests/ builds on its own)

LAUNCH_BOUND_CHECKS = {}
DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?unsigned char\s+(\w+)\[\];")


def operands_to_emu(text):
    """GHIP_MURMUR21_OPERANDS-style text `outs : ins : clobbers` -> (binds initialiser, output assignments)."""
    parts = [p.strip() for p in text.split("\n")]
    flat = " ".join(p.rstrip("\\").strip() for p in parts)
    fields, cur, quoted = [], "", False   # split at the colons outside string literals ("=&{v[40:41]}" has one inside)
    for ch in flat:
        if ch == '"':
            quoted = not quoted
        if ch == ":" and not quoted:
            fields.append(cur)
            cur = ""
        else:
            cur += ch
    fields.append(cur)
    outs, ins, _clob = [f.strip() for f in fields]
    binds, assigns = [], []
    for m in re.finditer(r'(?:\[(\w+)\]\s*)?"([^"]+)"\((\w+)\)', outs):
        name, cons, var = m.groups()
        fixed = re.search(r"\{v\[(\d+):(\d+)\]\}", cons)
        if fixed:
            assigns.append(f"{var} = hipemu_m_.reg64({fixed.group(1)});")
        else:
            assigns.append(f"{var} = (decltype({var}))hipemu_m_.out(\"{name}\");")
    for m in re.finditer(r'\[(\w+)\]\s*"([^"]+)"\(([^()]+)\)', ins):
        name, _cons, expr = m.groups()
        binds.append(f'{{"{name}", (uint64_t)({expr}), (int)(8 * sizeof({expr}))}}')
    return "{" + ", ".join(binds) + "}", " ".join(assigns)


def balanced(text, at):
    """text[at] == '(' -> index just past its closing parenthesis."""
    depth = 0
    for i in range(at, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses")


def check_launch_bounds(name, text):
    """Inserts the launch-bound check at the top of every kernel that declares one -> (text, number of kernels)."""
    out, at, n = "", 0, 0
    for m in re.finditer(r"__launch_bounds__\(", text):
        if m.start() < at:
            continue
        e_end = balanced(text, m.end() - 1)
        expr = text[m.end():e_end - 1]
        k = re.compile(r"(\w+)\s*\(").search(text, e_end)   # the kernel's name: the next identifier in front of a parameter list
        while k and k.group(1) in ("__attribute__", "amdgpu_waves_per_eu", "aligned"):
            k = re.compile(r"(\w+)\s*\(").search(text, balanced(text, k.end() - 1))
        assert k, (name, expr)
        p_end = balanced(text, k.end() - 1)
        rest = re.compile(r"\s*(\{|;)").match(text, p_end)
        assert rest, (name, k.group(1))
        if rest.group(1) == ";":
            continue   # a declaration
        out += text[at:rest.end()] + f' hipemu::check_launch_bound(({expr}), "{k.group(1)}");'
        at = rest.end()
        n += 1
    return out + text[at:], n


def transform(name, text):
    text = DYN.sub(lambda m: f"unsigned char *const {m.group(1)} = hipemu::dyn_lds();", text)
    text = text.replace('"../../include/galah_hip.h"', '"galah_hip.h"')
    text = re.sub(r"__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)\s*", "", text)   # an occupancy hint: no meaning on the host
    if name == "murmur21_asm.h":
        m = re.search(r"#define GHIP_MURMUR21_OPERANDS\s*\\\n((?:.*\\\n)*.*\n)", text)
        assert m, "murmur21_asm.h: operand macro not found"
        binds, assigns = operands_to_emu(m.group(1))
        text = text[: m.start()] + f"    const hipemu_gcn::Bind hipemu_b_[] = {binds};\n    hipemu_gcn::Machine hipemu_m_;\n" + text[m.end():]
        n_asm = 0

        def asm_sub(mm):
            nonlocal n_asm
            n_asm += 1
            body = mm.group(1).strip()
            return (f"hipemu_gcn::run({body}, hipemu_b_, (int)(sizeof hipemu_b_ / sizeof hipemu_b_[0]), hipemu_m_); {assigns}")
        text = re.sub(r"asm volatile\(((?:.|\n)*?)\s*:\s*GHIP_MURMUR21_OPERANDS\);", asm_sub, text)
        assert n_asm == 2, n_asm
        text = text.replace("#undef GHIP_MURMUR21_OPERANDS\n", "")
        text = text.replace("#include <hip/hip_runtime.h>", "#include <hip/hip_runtime.h>\n#include <hipemu_gcn_asm.h>")
    assert "GHIP_MURMUR21_OPERANDS" not in text
    if name.endswith(".hip"):
        text, n_bounds = check_launch_bounds(name, text)
        LAUNCH_BOUND_CHECKS[name] = n_bounds
    # any inline assembly left must be the empty scheduling fence of seed_common.h
    for m in re.finditer(r"asm volatile\(([^;]*)\);", text):
        assert m.group(1).strip() == '""', (name, m.group(0))
    return text


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in sorted(os.listdir(SRC)):
        if not f.endswith((".hip", ".cpp", ".h")):
            continue
        with open(os.path.join(SRC, f)) as fh:
            text = transform(f, fh.read())
        out = os.path.join(OUT, f + ".cpp" if f.endswith(".hip") else f)
        if os.path.exists(out) and open(out).read() == text:
            continue   # keep the timestamp: make rebuilds only what changed
        with open(out, "w") as fh:
            fh.write(text)
    assert sum(LAUNCH_BOUND_CHECKS.values()) >= 39, LAUNCH_BOUND_CHECKS   # the kernels that declare a launch bound (round 6: 39)


if __name__ == "__main__":
    sys.exit(main())
