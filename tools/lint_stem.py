#!/usr/bin/env python3
"""Static check on the first-layer + pool kernel's listing (csrc/conv_stem_i8.hip, conv_stem_pool_i8_kernel<7>): its operand
fragments are loaded by inline asm two items ahead and awaited by one counted `s_waitcnt vmcnt(7)` per item, so between the
issue of a fragment set and its wait NO instruction may read, copy or spill those registers (the compiler does not know a load
is in flight).  The listing is scanned linearly: for each of the two sets, from the last of its 7 loads inside the loop to the
end of the kernel and from the loop head to the set's wait, none of its registers may appear as an operand.

    python tools/lint_stem.py            (cross-compiles to assembly with hipcc; no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dlmc-quant_amd", "csrc", "conv_stem_i8.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-gpu-flush-denormals-to-zero", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def regs(line):
    line = line.split(";")[0]
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "stem.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, SRC], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    body, on = [], False
    for line in text:
        if re.match(r"^_ZN5dlmcq24conv_stem_pool_i8_kernelILi7E\w+:", line):
            on = True
        elif on:
            body.append(line)
            if "s_endpgm" in line:
                break
    assert body, "conv_stem_pool_i8_kernel<7> not found"
    head = next(i for i, l in enumerate(body) if "=>This Loop Header: Depth=1" in l)
    waits = [i for i, l in enumerate(body) if "s_waitcnt vmcnt(7)" in l]
    assert len(waits) == 2 and waits[0] > head, f"expected the two counted waits inside the loop, found {waits}"
    loads = [(i, re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", l)) for i, l in enumerate(body) if "global_load_dwordx4" in l and i > head]
    assert len(loads) == 14, f"expected 2 x 7 asm loads inside the loop, found {len(loads)}"
    sets = [loads[:7], loads[7:]]
    bad = 0
    for k, (st, w) in enumerate(zip(sets, waits)):
        mine = set()
        for _, m in st:
            mine.update(range(int(m.group(1)), int(m.group(2)) + 1))
        last = st[-1][0]
        spans = [(last + 1, len(body)), (head, w)]
        for a, b in spans:
            for i in range(a, b):
                if "global_load_dwordx4" in body[i]:
                    continue
                hit = regs(body[i]) & mine
                if hit:
                    print(f"set {k}: line {i}: `{body[i].strip()}` touches in-flight registers {sorted(hit)}")
                    bad += 1
        # a load of this set may use registers of its own set as ADDRESS only before they are loaded in that same sequence
    for i in range(head, len(body)):
        if "scratch_" in body[i]:
            print(f"line {i}: {body[i].strip()}  [spill inside the item loop]")
            bad += 1
    print(f"conv_stem_pool_i8_kernel<7>: 2 fragment sets checked, {bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
