#!/usr/bin/env python3
"""The chain kernel on the SEVEN shapes of the ResNet-50 b512 plan (11 launches), launched as the plan launches them (plain
quantisers: null zero point, unsigned byte range, chunk-major second weights), under any build of the library:

    python tools/chain_ab.py [--lib path/to/libdlmcq_variant.so] [--lab 0,1,32] [--cases d1,s1,...] [--iters 7]

Prints per shape the median launch time, the algorithmic bytes, TB/s, the fraction of 8 TB/s and a checksum of the outputs
(equal checksums across builds = the same bytes), then the sum over the plan's 11 launches.  `--lab` flags are the lab library's
timing-only ablations (dlmcq_x_chain_lab; 1 = no shortcut loads, 32 = no weight DMA, 33 = neither; a lab build is needed).
Round 5: the tool the A/B runs of the chain variants under csrc/lab/ go through (one process per build, alternated by the caller)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--lab", default="0")
ap.add_argument("--cases", default="d1,s1,s1t,d2,s2,s2t,s3")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=7)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--cm", action="store_true", help="the fp32 tensors chunk-major (DLMCQ_FP32_CHUNK_MAJOR; same checksums: same values)")
ap.add_argument("--abcm", action="store_true", help="time row-major and chunk-major fp32 tensors alternately, launch by launch, in this process")
ap.add_argument("--noout", action="store_true", help="also time every case without its fp32 / code stores (lab-free ablation)")
args = ap.parse_args()
if args.lib:
    os.environ["DLMCQ_LIBRARY"] = os.path.abspath(args.lib)
    os.environ["DLMCQ_LAB_TOOLS"] = "1"
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

# name: (C1, H, KD, K2, want_out, want_codes, C2, H2, stride2, launches in the plan)
CASES = {
    "d1": (64, 56, 256, 64, True, False, 64, 56, 1, 1),       # stage 1's first block: conv3 + conv shortcut -> next conv1
    "s1": (64, 56, 256, 64, True, False, 0, 0, 0, 1),
    "s1t": (64, 56, 256, 128, False, True, 0, 0, 0, 1),
    "d2": (128, 28, 512, 128, True, False, 256, 56, 2, 1),    # stage 2's first block
    "s2": (128, 28, 512, 128, True, False, 0, 0, 0, 2),
    "s2t": (128, 28, 512, 256, False, True, 0, 0, 0, 1),
    "s3": (256, 14, 1024, 256, True, False, 0, 0, 0, 4),
}
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(11)
labs = [int(f) for f in args.lab.split(",") if f]
has_lab = hasattr(N.lib, "dlmcq_x_chain_lab")
if any(labs) and not has_lab:
    sys.exit("chain_ab: --lab needs a lab build (make -C dlmc-quant_amd/csrc lab)")
total = {fl: 0.0 for fl in labs}
total_bytes = 0


def layer(k, c):
    w = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=dev, dtype=torch.int8)
    return dict(wq=w, wsum=w.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k, generator=g, device=dev),
                w_scale=(torch.rand(k, generator=g, device=dev) * 0.004 + 0.001) * (64.0 / c))


def csum(t):
    if t is None:
        return 0
    if isinstance(t, K.ChunkMajor):
        t = t.buf
    return int(t.view(torch.uint8).to(torch.int64).sum().item()) if t.dtype != torch.float32 else int(t.view(torch.int32).to(torch.int64).sum().item())


for name in args.cases.split(","):
    c, h, k, k2, want_out, want_codes, c2, h2, st2, count = CASES[name]
    n = args.batch
    m = n * h * h
    nbytes = m * (c + k * (4 * (c2 == 0) + 4 * want_out + want_codes) + k2) + (m * c2 if c2 else 0) + k * (c + c2 + k2)
    nset = max(2, int(400e6 // nbytes) + 1)
    xs = [torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
          for _ in range(nset)]
    a = dict(layer(k, c), in_scale=torch.full((1,), 0.02, device=dev), in_zp=None)
    b = layer(k2, k)
    b["wq_chunk"] = K.chunk_major(b["wq"])     # what the plan passes (DLMCQ_W2_CHUNK_MAJOR)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
    emit2 = K.EmitCodes(torch.full((1,), 0.11, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
    if c2:
        x2s = [torch.randint(0, 256, (n, c2, h2, h2), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
               for _ in range(nset)]
        sc = dict(layer(k, c2), in_scale=torch.full((1,), 0.03, device=dev), in_zp=None, stride=st2)

        def run(i, wo=want_out, wc=want_codes, cm=args.cm):
            return K.conv2d_i8_dual_chain(dict(a, codes=xs[i]), dict(sc, codes=x2s[i]), b, relu=True, emit=emit, want_out=wo, want_codes=wc,
                                          relu3=True, emit3=emit2, rows_per_tile=args.rows, out_chunk_major=cm)
    else:
        ress = [torch.randn(n, k, h, h, generator=g, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nset)]
        ress_cm = [K.ChunkMajor.from_nhwc(r) for r in ress] if (args.cm or args.abcm) else None

        def run(i, wo=want_out, wc=want_codes, cm=args.cm):
            return K.conv2d_i8_chain(dict(a, codes=xs[i]), b, (ress_cm if cm else ress)[i], relu=True, emit=emit, want_out=wo, want_codes=wc,
                                     relu2=True, emit2=emit2, rows_per_tile=args.rows, out_chunk_major=cm)
    got = run(0)
    torch.cuda.synchronize()
    sums = "/".join(f"{csum(t) & 0xffffffff:08x}" for t in got)
    variants = [(f"lab={fl}" if fl else "as built", fl) for fl in labs]
    if args.noout:
        variants.append(("no stores", -1))
    if args.abcm:
        variants.append(("chunk-major", -2))
    times = {lbl: [] for lbl, _ in variants}
    for it in range(args.iters):
        for lbl, fl in variants:
            if fl > 0:
                N.lib.dlmcq_x_chain_lab(fl)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if fl == -2:
                run(it % nset, cm=True)
            elif fl == -1:
                run(it % nset, False, False)
            else:
                run(it % nset)
            e1.record()
            torch.cuda.synchronize()
            if fl > 0:
                N.lib.dlmcq_x_chain_lab(0)
            times[lbl].append(e0.elapsed_time(e1) * 1e3)
    line = f"{name:4s} C{c:<4d}{('+' + str(c2)) if c2 else '':5s} {h:>2d}^2 K{k:<5d} K2 {k2:<4d} x{count}  {nbytes / 1e6:7.1f} MB  sum {sums}"
    for lbl, fl in variants:
        t = sorted(times[lbl])[len(times[lbl]) // 2]
        line += f" | {lbl} {t:7.1f} us {nbytes / t / 1e6:5.2f} TB/s ({nbytes / t / 1e6 / 8:.3f})"
        if fl in total:
            total[fl] += t * count
        if fl == -2:
            total_cm = globals().get("total_cm", 0.0) + t * count
            globals()["total_cm"] = total_cm
    total_bytes += nbytes * count
    print(line, flush=True)
    del xs
    if c2:
        del x2s
    else:
        del ress
    torch.cuda.empty_cache()
if args.abcm and set(args.cases.split(",")) == set(CASES):
    print(f"SUM over the plan's 11 launches, chunk-major fp32 tensors: {total_cm:8.1f} us  {total_bytes / total_cm / 1e6:.2f} TB/s = {total_bytes / total_cm / 1e6 / 8:.3f} of 8 TB/s")
if set(args.cases.split(",")) == set(CASES):
    for fl in labs:
        print(f"SUM over the plan's 11 launches, {'lab=' + str(fl) if fl else 'as built'}: {total[fl]:8.1f} us  {total_bytes / 1e9:.3f} GB  "
              f"{total_bytes / total[fl] / 1e6:.2f} TB/s = {total_bytes / total[fl] / 1e6 / 8:.3f} of 8 TB/s")
