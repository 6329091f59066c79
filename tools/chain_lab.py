#!/usr/bin/env python3
"""The chain kernel (block end + next 1x1 reduction, dlmcq_conv2d_i8_nhwc_chain) against the two launches it replaces, on the
ResNet-50 shapes at batch 512: interleaved timing over rotating buffers, results compared bit for bit.

    python tools/chain_lab.py [--batch 512] [--cases s1,s1t,...] [--rows 0,64,...]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

if "--lab" in " ".join(sys.argv):
    os.environ["DLMCQ_LIBRARY"] = os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so")
    os.environ["DLMCQ_LAB_TOOLS"] = "1"
from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

# name: (C, H, K, K2, want_out, want_codes)   inside a stage: fp32 out, no codes; stage end: codes, no fp32 out
CASES = {
    "s1": (64, 56, 256, 64, True, False),
    "s1t": (64, 56, 256, 128, False, True),
    "s2": (128, 28, 512, 128, True, False),
    "s2t": (128, 28, 512, 256, False, True),
    "s3": (256, 14, 1024, 256, True, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--rows", default="0")
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--lab", default="", help="lab library flags to time as extra variants, e.g. 1,2,noout (1 = no shortcut loads, 4 / 8 = fp32 accesses as 16 rows x 64 bytes / 4 rows x 256 bytes per instruction; timing only)")
    args = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(11)
    rows_list = [int(r) for r in args.rows.split(",")]
    for name in args.cases.split(","):
        c, h, k, k2, want_out, want_codes = CASES[name]
        n = args.batch
        m = n * h * h
        per_set = m * (c + k * (4 + 4 * want_out + want_codes) + k2)
        nset = max(2, int(300e6 // per_set) + 1)
        xs = [torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
              for _ in range(nset)]
        ress = [torch.randn(n, k, h, h, generator=g, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nset)]
        w1 = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=dev, dtype=torch.int8)
        w2 = torch.randint(-127, 128, (k2, 1, 1, k), generator=g, device=dev, dtype=torch.int8)
        a = dict(wq=w1, wsum=w1.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k, generator=g, device=dev),
                 in_scale=torch.full((1,), 0.02, device=dev), in_zp=torch.full((1,), 3.0, device=dev),
                 w_scale=(torch.rand(k, generator=g, device=dev) * 0.004 + 0.001))
        b = dict(wq=w2, wsum=w2.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k2, generator=g, device=dev),
                 w_scale=(torch.rand(k2, generator=g, device=dev) * 0.002 + 0.0005))
        emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), torch.zeros(1, device=dev), 0, 255, N.FORM_ZEROPOINT)
        emit2 = K.EmitCodes(torch.full((1,), 0.11, device=dev), torch.zeros(1, device=dev), 0, 255, N.FORM_ZEROPOINT)

        def two(i):
            out, codes = K.conv2d_i8(xs[i], a["wq"], a["wsum"], a["bias"], a["in_scale"], a["in_zp"], a["w_scale"], residual=ress[i],
                                     relu=True, emit=emit, want_out=want_out)
            _, codes2 = K.conv2d_i8(codes, b["wq"], b["wsum"], b["bias"], emit.scale, emit.zero_point, b["w_scale"], relu=True,
                                    emit=emit2, want_out=False)
            return out, codes, codes2

        def one(i, rows):
            return K.conv2d_i8_chain(dict(a, codes=xs[i]), b, ress[i], relu=True, emit=emit, want_out=want_out, want_codes=want_codes,
                                     relu2=True, emit2=emit2, rows_per_tile=rows)
        ref = two(0)
        variants = [("two launches", lambda i: two(i))] + [(f"chain rows={r}", (lambda i, r=r: one(i, r))) for r in rows_list]
        for fl in [f for f in args.lab.split(",") if f]:
            if fl == "noout":
                variants.append(("chain no stores", lambda i: K.conv2d_i8_chain(dict(a, codes=xs[i]), b, ress[i], relu=True, emit=emit,
                                                                               want_out=False, want_codes=False, relu2=True, emit2=emit2,
                                                                               rows_per_tile=64)))
            else:
                def lab_run(i, fl=int(fl)):
                    N.lib.dlmcq_x_chain_lab(fl)
                    r = one(i, 64)
                    N.lib.dlmcq_x_chain_lab(0)
                    return r
                variants.append((f"chain lab={fl}", lab_run))
        for label, fn in variants[1:1 + len(rows_list)]:
            got = fn(0)
            torch.cuda.synchronize()
            ok = (not want_out or torch.equal(got[0], ref[0])) and (not want_codes or torch.equal(got[1], ref[1])) and torch.equal(got[2], ref[2])
            if not ok:
                print(f"  !! {name} {label}: differs from the two launches")
        times = {label: [] for label, _ in variants}
        for it in range(args.iters):
            for label, fn in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(it % nset)
                e1.record()
                torch.cuda.synchronize()
                times[label].append(e0.elapsed_time(e1) * 1e3)
        chain_bytes = m * (c + k * (4 + 4 * want_out + want_codes) + k2)
        line = f"{name:4s} C{c:<4d} {h:>2d}^2 K{k:<5d} K2 {k2:<4d}"
        for label, _ in variants:
            t = sorted(times[label])[len(times[label]) // 2]
            line += f" | {label} {t:7.1f} us ({chain_bytes / t / 1e3:5.0f} GB/s of the chain's bytes)"
        print(line, flush=True)
        del xs, ress
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
