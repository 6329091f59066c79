#!/usr/bin/env python3
"""Static checks on the chain kernel's listing (csrc/conv_chain_i8.hip holds hand-counted waits and asm loads whose target
registers are in flight across most of a chunk, so what the compiler does around them matters):

  * no scratch (spill) instruction between the first and the last s_barrier of a kernel - a spill of an asm-loaded register
    while its load is in flight would save and restore garbage; prologue / tail spills of ordinary values are reported only;
  * every asm buffer_store_dwordx4 is followed by its two wait states (`s_nop 1`: the hazard recogniser does not look
    inside inline asm, gfx940+ needs two before a VALU write of the store data);
  * the shortcut tile is loaded by asm buffer_load_dwordx4 a whole chunk before its counted wait, and for the compiler the
    destination registers hold their value from the asm statement on: between a load inside the chunk loop and the next
    `s_waitcnt vmcnt` in program order (wrapping around the loop) NO instruction may read, copy or overwrite them.

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


def regs(line):
    line = line.split(";")[0]
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def in_flight_check(name, body):
    """Registers of the in-loop asm loads must not be touched before the next vmcnt wait.  Returns the number of problems."""
    heads = [i for i, l in enumerate(body) if "=>This Loop Header: Depth=1" in l or ("Loop Header: Depth=1" in l and "=>" in l)]
    first_bar = next(i for i, l in enumerate(body) if "s_barrier" in l)     # (the prologue's: the chunk loop starts behind it)
    heads = [h for h in heads if h > first_bar]
    if not heads:
        print(f"{name}: chunk loop header not found")
        return 1
    # the chunk loop = the depth-1 loop that holds the in-loop asm loads
    loads = [(i, re.search(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", l)) for i, l in enumerate(body)
             if "buffer_load_dwordx4" in l and " lds" not in l]
    bad = checked = 0
    for head in heads:
        label = body[head].split(":")[0].strip()
        ends = [i for i, l in enumerate(body) if i > head and re.search(r"s_cbranch\w+\s+" + re.escape(label) + r"\b", l)]
        if not ends:
            continue
        end = ends[-1]
        inside = [(i, m) for i, m in loads if head < i < end and m]
        checked += len(inside)
        for i, m in inside:
            mine = set(range(int(m.group(1)), int(m.group(2)) + 1))
            j, steps = i + 1, 0
            while steps < 2 * (end - head):
                if j > end:
                    j = head
                line = body[j]
                if "s_waitcnt" in line and "vmcnt(" in line:
                    break
                if "buffer_load_dwordx4" not in line:
                    hit = regs(line) & mine
                    if hit:
                        print(f"{name}: line {j}: `{line.strip()}` touches registers {sorted(hit)} of the load at line {i} still in flight")
                        bad += 1
                j += 1
                steps += 1
            else:
                print(f"{name}: no vmcnt wait found behind the load at line {i}")
                bad += 1
    # round 5: chunk 0's shortcut tile is requested in the PROLOGUE, before the wave waits for its input tile (a counted wait that leaves
    # these loads in flight): from each such load to the first `s_waitcnt vmcnt(0)` in listing order nothing may touch its registers
    first_head = min(heads)
    pro = [(i, m) for i, m in loads if i < first_head and m]
    for i, m in pro:
        mine = set(range(int(m.group(1)), int(m.group(2)) + 1))
        for j in range(i + 1, len(body)):
            line = body[j]
            if "s_waitcnt" in line and "vmcnt(0)" in line:
                break
            if "buffer_load_dwordx4" not in line:
                hit = regs(line) & mine
                if hit:
                    print(f"{name}: line {j}: `{line.strip()}` touches registers {sorted(hit)} of the prologue load at line {i} still in flight")
                    bad += 1
    checked_pro = len(pro)
    fp32_shortcut = re.search(r"kernelILi\d+ELi\d+ELi0E", name) is not None     # (C2 = 0; the others reduce a convolution shortcut instead)
    if fp32_shortcut and checked < 4:   # at least one 4-load shortcut tile request inside the chunk loop: a vacuous pass is a failure
        print(f"{name}: only {checked} in-loop asm loads found - the check did not see the chunk loop")
        bad += 1
    if fp32_shortcut and checked_pro < 4:
        print(f"{name}: only {checked_pro} prologue asm loads found - the check did not see chunk 0's request")
        bad += 1
    return bad


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
        assert len(bars) >= 5, name
        # the chunk loop: from its header to the last barrier (the prologue has one barrier of its own - the input tile staged in LDS -
        # before any asm load is in flight: a spill there is only reported)
        heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and "=>" in l and i > bars[0]]
        lo = heads[0] if heads else bars[1]
        for i, l in enumerate(body):
            if "scratch_" in l:
                inside = lo < i < bars[-1]
                print(f"{name}: line {i}: {l.strip()}  [{'inside the chunk loop' if inside else 'outside the loop'}]")
                bad += inside
            if "buffer_store_dwordx4" in l:
                nxt = next(b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";"))
                if nxt != "s_nop 1":
                    print(f"{name}: line {i}: store not followed by s_nop 1 but by `{nxt}`")
                    bad += 1
        bad += in_flight_check(name, body)
    print(f"{len(kernels)} kernels checked, {bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
