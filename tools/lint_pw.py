#!/usr/bin/env python3
"""Static checks on the pointwise kernel's listing (csrc/conv_pw_i8.hip): its activation fragments are loaded by asm
buffer_load_dwordx4 in the middle of the previous block's epilogue and awaited by a counted s_waitcnt at the top of the loop, so

  * the kernel must not use scratch at all (a spill of a fragment register whose load is in flight saves garbage);
  * between a load and the next `s_waitcnt vmcnt` in program order (wrapping around the loop) no instruction may read, copy
    or overwrite its destination registers;
  * every asm buffer_store_dwordx4 is followed by its two wait states (`s_nop 1`).

    python tools/lint_pw.py            (cross-compiles to assembly with hipcc; no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lint_chain import FLAGS, ROOT, regs  # noqa: E402

SRC = os.path.join(ROOT, "dlmc-quant_amd", "csrc", "conv_pw_i8.hip")


def in_flight_check(name, body, nfrag):
    """The block loop's text = every basic block annotated `in Loop: Header=<the depth-1 loop behind the prologue's barriers>` (the
    latch may be laid out in front of the header).  From each fragment load the listing is walked forward to the end of that text,
    then on from its start, up to the counted wait; any touch of the load's registers on the way is a problem."""
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and "=>" in l and bars and i > bars[-1]]
    if not heads:
        print(f"{name}: block loop header not found")
        return 1
    head = heads[0]
    tag = "Header=" + body[head].split(":")[0].strip().lstrip(".L")
    labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    member = [i for i in labels if tag in body[i]] + [head]
    lo = min(member)
    last = max(member)
    hi = next((i for i in labels if i > last), len(body)) - 1
    loads = [(i, re.search(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", body[i])) for i in range(lo, hi + 1) if "buffer_load_dwordx4" in body[i]]
    bad = 0
    if len(loads) != nfrag:          # a vacuous pass is a failure
        print(f"{name}: {len(loads)} fragment loads inside the block loop, expected {nfrag}")
        bad += 1
    for i, m in loads:
        mine = set(range(int(m.group(1)), int(m.group(2)) + 1))
        j, steps = i + 1, 0
        while steps < 2 * (hi - lo + 1):
            if j > hi:
                j = lo
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
    return bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "pw.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, SRC], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur = {}, None
    for line in text:
        m = re.match(r"^(_ZN5dlmcq17conv_pw_i8_kernel\w+):", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    assert len(kernels) >= 10, f"expected the pointwise kernel instantiations, found {len(kernels)}"
    bad = 0
    for name, body in kernels.items():
        c = int(re.search(r"kernelILi(\d+)E", name).group(1))
        for i, l in enumerate(body):
            if "scratch_" in l:
                print(f"{name}: line {i}: {l.strip()}")
                bad += 1
            if "buffer_store_dwordx4" in l:
                nxt = next(b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";"))
                if nxt != "s_nop 1":
                    print(f"{name}: line {i}: store not followed by s_nop 1 but by `{nxt}`")
                    bad += 1
        bad += in_flight_check(name, body, c // 32)
    print(f"{len(kernels)} kernels checked, {bad} problem(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
