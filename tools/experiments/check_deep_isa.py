"""Build-time check of gemm_deep_kernel's inline-asm loads (neat_amd/csrc/gemm.hip): between a `global_load_dwordx4` issued
from inline asm and the `; W_READY` marker behind the counted wait, no other instruction may name the load's destination
registers -- the compiler does not know that they are still in flight, and a copy or spill placed there would read them
before the data has returned.  Usage: python tools/check_deep_isa.py gemm.s (the -S output of the device compile)."""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    text = open(path).read()
    bad = 0
    kernels = 0
    for m in re.finditer(r"^(_ZN4neat16gemm_deep_kernel\w+):.*?\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        kernels += 1
        name, body = m.group(1), m.group(2).split("\n")
        pending = {}                      # register -> line of the asm load
        in_asm = False
        for i, line in enumerate(body):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") and "W_READY" not in t or t.startswith("."):
                if "W_READY" in t:
                    pending.clear()
                continue
            if "W_READY" in t:
                pending.clear()
                continue
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
            if in_asm and t.startswith("global_load_dwordx4"):
                for r in regs_of(toks[0]):
                    pending[r] = i
                continue
            used = set()
            for tok in toks:
                used |= regs_of(tok)
            hit = used & set(pending)
            if hit:
                bad += 1
                print(f"{name}: line {i}: `{t}` touches in-flight registers {sorted(hit)[:8]}")
            if t.startswith("s_barrier") or t.startswith("s_cbranch") or t.startswith("s_branch"):
                pass
    print(f"{kernels} gemm_deep_kernel instances checked, {bad} violations")
    return 1 if bad or not kernels else 0


if __name__ == "__main__":
    sys.exit(check(sys.argv[1]))
