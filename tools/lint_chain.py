#!/usr/bin/env python3
"""Static checks on the chain kernel's listing (csrc/conv_chain_i8.hip holds hand-counted waits and asm loads whose target
registers are in flight across most of a chunk, so what the compiler does around them matters):

  * no scratch (spill) instruction between the first and the last s_barrier of a kernel - a spill of an asm-loaded register
    while its load is in flight would save and restore garbage; prologue / tail spills of ordinary values are reported only;
  * every asm buffer_store_dwordx4 is followed by its two wait states (`s_nop 1`: the hazard recogniser does not look
    inside inline asm, gfx940+ needs two before a VALU write of the store data).

    python tools/lint_chain.py            (cross-compiles to assembly with hipcc; no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dlmc-quant_amd", "csrc", "conv_chain_i8.hip")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-gpu-flush-denormals-to-zero", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "chain.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, SRC], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur = {}, None
    for line in text:
        m = re.match(r"^(_ZN5dlmcq20conv_chain_i8_kernel\w+):", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    assert len(kernels) >= 7, f"expected the 7 chain kernel instantiations, found {len(kernels)}"
    bad = 0
    for name, body in kernels.items():
        bars = [i for i, l in enumerate(body) if "s_barrier" in l]
        assert len(bars) >= 4, name
        for i, l in enumerate(body):
            if "scratch_" in l:
                where = "inside the chunk loop" if bars[0] < i < bars[-1] else "outside the loop"
                print(f"{name}: line {i}: {l.strip()}  [{where}]")
                bad += bars[0] < i < bars[-1]
            if "buffer_store_dwordx4" in l:
                nxt = next(b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";"))
                if nxt != "s_nop 1":
                    print(f"{name}: line {i}: store not followed by s_nop 1 but by `{nxt}`")
                    bad += 1
    print(f"{len(kernels)} kernels checked, {bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
