#!/usr/bin/env python3
"""Build-time lint for kernels that keep in-flight loads in hand-named accumulator registers (csrc/lab/agpr_asm.h).

For every kernel of a device assembly listing (hipcc --cuda-device-only -S) it checks that
  * nothing is spilled to scratch (a spill of a register whose load is still in flight stores garbage), and
  * no compiler-generated instruction touches an accumulator register that the inline-asm blocks use
    (the compiler treats a clobbered register as free between statements).
usage: lint_agpr.py file.s [kernel-name-substring]      exit status 1 on a violation."""
import re
import sys


def regs(text):
    out = set()
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\ba(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    name, in_asm, mine, theirs, scratch = None, False, set(), {}, 0
    bad = False
    results = []

    def flush():
        nonlocal bad
        if name is None or want not in name:
            return
        clash = {r: l for r, l in theirs.items() if r in mine}
        ok = not clash and scratch == 0
        results.append((name, len(mine), len(theirs), scratch, ok))
        if clash:
            bad = True
            for r, l in sorted(clash.items())[:5]:
                print(f"  {name}: compiler code touches hand-managed a{r}: {l.strip()}")
        if scratch:
            bad = True
            print(f"  {name}: {scratch} scratch instructions (spills)")

    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            name, in_asm, mine, theirs, scratch = m.group(1), False, set(), {}, 0
            continue
        if name is None:
            continue
        if "s_endpgm" in line and False:
            pass
        if ";;#ASMSTART" in line or ";#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line or ";#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0]
        if not code.strip() or code.strip().startswith("."):
            continue
        if "scratch_" in code:
            scratch += 1
        r = regs(code)
        if in_asm:
            mine |= r
        else:
            for x in r:
                theirs.setdefault(x, line)
    flush()
    for n, a, b, s, ok in results:
        print(f"{'ok ' if ok else 'BAD'} {n[:90]}  asm-agprs={a} compiler-agprs={b} scratch={s}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
