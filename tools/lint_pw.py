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
    """Control-flow walk: from every fragment load inside the block loop, every instruction reachable before a `s_waitcnt vmcnt`
    (fall-through, both ways of a conditional branch, the target of an unconditional one; rarely executed blocks are laid out far
    from the loop, so the listing's order says nothing) must leave the load's registers alone."""
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and "=>" in l and bars and i > bars[-1]]
    if not heads:
        print(f"{name}: block loop header not found")
        return 1
    tag = "Header=" + body[heads[0]].split(":")[0].strip().lstrip(".L")

    def in_loop(i):          # the basic block of line i belongs to the block loop (its label line says so; the header itself does)
        j = i
        while j >= 0 and not re.match(r"^\.LBB\d+_\d+:", body[j]) and "; %bb." not in body[j]:
            j -= 1
        return j >= 0 and (tag in body[j] or j == heads[0])
    loads = [(i, re.search(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", body[i])) for i in range(len(body))
             if "buffer_load_dwordx4" in body[i] and i > bars[-1] and in_loop(i)]
    bad = 0
    if len(loads) != nfrag:          # a vacuous pass is a failure
        print(f"{name}: {len(loads)} fragment loads inside the block loop, expected {nfrag}")
        bad += 1
    for i, m in loads:
        mine = set(range(int(m.group(1)), int(m.group(2)) + 1))
        seen, todo, waited = set(), [i + 1], 0
        while todo:
            j = todo.pop()
            while j < len(body) and j not in seen:
                seen.add(j)
                line = body[j].split(";")[0].strip()
                if not line or line.endswith(":"):
                    j += 1
                    continue
                if line.startswith("s_waitcnt") and "vmcnt(" in line:
                    waited += 1
                    break
                if "buffer_load_dwordx4" not in line:
                    hit = regs(line) & mine
                    if hit:
                        print(f"{name}: line {j}: `{line}` touches registers {sorted(hit)} of the load at line {i} still in flight")
                        bad += 1
                if line.startswith("s_endpgm"):
                    print(f"{name}: the kernel can end with the load at line {i} in flight (no vmcnt wait on the way)")
                    bad += 1
                    break
                mb = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", line)
                if mb:
                    todo.append(label_at[mb.group(2)])
                    if mb.group(1) == "s_branch":
                        break
                j += 1
        if not waited:
            print(f"{name}: no vmcnt wait reachable from the load at line {i}")
            bad += 1
    return bad


def queue_check(name, body, loads_per_block):
    """The residual-block kernel (csrc/conv_pwr_i8.hip) keeps loads in flight ACROSS counted waits, so the walk carries the whole
    vector-memory queue: every buffer / LDS-DMA instruction joins it, `s_waitcnt vmcnt(n)` retires all but the youngest n, and no
    instruction may touch the destination registers of a load that is still in it.  All paths from the kernel's entry (both ways
    of every conditional branch), to a fixpoint over (line, queue)."""
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    bad, reported = 0, set()
    seen, todo = set(), [(0, ())]
    inloop = set()
    while todo:
        j, q = todo.pop()
        while j < len(body) and (j, q) not in seen:
            seen.add((j, q))
            line = body[j].split(";")[0].strip()
            if not line or line.endswith(":"):
                j += 1
                continue
            mw = re.search(r"vmcnt\((\d+)\)", line) if line.startswith("s_waitcnt") else None
            if mw:
                n = int(mw.group(1))
                q = q[len(q) - n:] if n and len(q) > n else (q if n else ())
                j += 1
                continue
            for k, dst in q:
                hit = regs(line) & set(range(dst[0], dst[1] + 1)) if dst else set()
                if hit and (j, k) not in reported:
                    reported.add((j, k))
                    print(f"{name}: line {j}: `{line}` touches registers {sorted(hit)} of the load at line {k} still in flight")
                    bad += 1
            if line.startswith("s_endpgm"):
                if any(dst for _, dst in q):
                    print(f"{name}: the kernel can end with loads in flight: lines {[k for k, dst in q if dst]}")
                    bad += 1
                break
            if re.match(r"(buffer|global)_(load|store|atomic)", line):
                ml = re.match(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", line)
                q = (q + (((j, (int(ml.group(1)), int(ml.group(2)))) if ml else (None, None)),))[-63:]     # (stores and LDS-DMA: anonymous, so that the prologue's loops converge)
                if ml:
                    inloop.add(j)
            mb = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", line)
            if mb:
                todo.append((label_at[mb.group(2)], q))
                if mb.group(1) == "s_branch":
                    break
            j += 1
    if len(inloop) < loads_per_block:          # a vacuous pass is a failure
        print(f"{name}: {len(inloop)} asm loads seen, expected at least {loads_per_block}")
        bad += 1
    return bad


def listing(src, pattern):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-fno-slp-vectorize", "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    kernels, cur = {}, None
    for line in text:
        m = re.match(pattern, line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line and "s_endpgm" == line.split(";")[0].strip():
                pass
            if line.startswith(".Lfunc_end"):
                cur = None
    return kernels


def common_checks(name, body):
    bad = 0
    for i, l in enumerate(body):
        if "scratch_" in l:
            print(f"{name}: line {i}: {l.strip()}")
            bad += 1
        if "buffer_store_dwordx4" in l:
            nxt = next(b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";"))
            if nxt != "s_nop 1":
                print(f"{name}: line {i}: store not followed by s_nop 1 but by `{nxt}`")
                bad += 1
    return bad


def main_pwr():
    kernels = listing(os.path.join(ROOT, "dlmc-quant_amd", "csrc", "conv_pwr_i8.hip"), r"^(_ZN5dlmcq18conv_pwr_i8_kernel\w+):")
    assert len(kernels) >= 4, f"expected the residual-block kernel instantiations, found {len(kernels)}"
    bad = 0
    for name, body in kernels.items():
        c = int(re.search(r"kernelILi(\d+)E", name).group(1))
        bad += common_checks(name, body)
        bad += queue_check(name, body, 2 * (c // 32 + 4) + 12)     # prologue and loop: fragments + shortcut rows
    print(f"{len(kernels)} residual-block kernels checked, {bad} problem(s)")
    return bad


def main():
    bad_pwr = main_pwr()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "pw.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-fno-slp-vectorize", "-o", out, SRC], stderr=subprocess.DEVNULL)   # (csrc/Makefile: EXTRA_conv_pw_i8)
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
    return 1 if bad or bad_pwr else 0


if __name__ == "__main__":
    sys.exit(main())
