#!/usr/bin/env python3
"""A/B timing of the int8 conv kernel on single ResNet-50 layers, through the LAB library (`make -C dlmc-quant_amd/csrc lab`).

Each case is one layer of the frozen ResNet-50 plan at batch 512 (shape + epilogue flavour); each knob set is
(tile width, A direct to registers, persistent kernel ring depth / waves per SIMD).  All knob sets of a case run interleaved
in one process over rotating buffers (> 256 MiB, past the Infinity Cache) and are checked against the first one bit for bit.

    python tools/conv_lab.py [--batch 512] [--cases d1,b2,...] [--knobs 128:1,64:1,...]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

# name: (C, H, K, R, stride, residual, out, codes)
CASES = {
    "b0": (64, 56, 64, 1, 1, False, False, True),
    "c1": (64, 56, 64, 3, 1, False, False, True),
    "d1": (64, 56, 256, 1, 1, True, True, True),
    "d1c": (64, 56, 256, 1, 1, True, False, True),
    "b1": (256, 56, 64, 1, 1, False, False, True),
    "b1w": (256, 56, 128, 1, 1, False, False, True),
    "c2": (128, 28, 128, 3, 1, False, False, True),
    "d2": (128, 28, 512, 1, 1, True, True, True),
    "b2": (512, 28, 128, 1, 1, False, False, True),
    "c3": (256, 14, 256, 3, 1, False, False, True),
    "d3": (256, 14, 1024, 1, 1, True, True, True),
    "b3": (1024, 14, 256, 1, 1, False, False, True),
    "b3w": (1024, 14, 512, 1, 1, False, False, True),
    "c3s": (256, 28, 256, 3, 2, False, False, True),
    "c4": (512, 7, 512, 3, 1, False, False, True),
    "d4": (512, 7, 2048, 1, 1, True, True, True),
    "b4": (2048, 7, 512, 1, 1, False, False, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--knobs", default="128:1,64:1,64:0")   # bn:adir[:pp_nbuf:pp_wps]
    ap.add_argument("--iters", type=int, default=6)
    args = ap.parse_args()
    lab = ctypes.CDLL(os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so"))
    fn = lab.dlmcq_x_conv2d_i8_tuned
    fn.restype = ctypes.c_int
    p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    fn.argtypes = [p] * 8 + [i64] * 7 + [i32] * 4 + [p, i32, p, p, p, i32, i32, i32, f32, p, i32, i32, i32, i32]
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(2333)
    knobs = []
    for k in args.knobs.split(","):
        v = [int(t) for t in k.split(":")]
        knobs.append(tuple(v + [0] * (4 - len(v))))
    for name in args.cases.split(","):
        c, h, k, r, stride, res, want_out, want_codes = CASES[name]
        n = args.batch
        pq = (h + 2 * (r // 2) - r) // stride + 1
        in_bytes, out_elems = n * c * h * h, n * k * pq * pq
        nset = max(2, int(300e6 // max(1, in_bytes + out_elems * (4 * res + 4 * want_out + want_codes))) + 1)
        xs = [torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
              for _ in range(nset)]
        wq = torch.randint(-127, 128, (k, r, r, c), generator=g, device=dev, dtype=torch.int8)
        wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
        s_w = (torch.rand(k, generator=g, device=dev) * 0.01 + 0.001).contiguous()
        bias = torch.randn(k, generator=g, device=dev)
        s_in = torch.full((1,), 0.02, device=dev)
        zp = torch.full((1,), 3.0, device=dev)
        q_s = torch.full((1,), 0.5, device=dev)
        q_z = torch.zeros(1, device=dev)
        ress = [torch.randn(n, k, pq, pq, generator=g, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nset)] if res else None
        outs = [torch.empty(n, k, pq, pq, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nset)] if want_out else None
        cods = [torch.empty(n, k, pq, pq, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last) for _ in range(nset)]

        def run(kn, i):
            rc = fn(N.ptr(xs[i]), N.ptr(wq), N.ptr(outs[i]) if want_out else None, N.ptr(bias), N.ptr(wsum), N.ptr(s_in), N.ptr(zp),
                    N.ptr(s_w), n, h, h, c, k, r, r, stride, r // 2, 1, 1, N.ptr(ress[i]) if res else None, 1, N.ptr(cods[i]),
                    N.ptr(q_s), N.ptr(q_z), 0, 255, N.FORM_ZEROPOINT, 0.0, N.stream_ptr(), kn[0], kn[1], kn[2], kn[3])
            if rc == -1:
                return False                      # this knob set does not exist for this layer (DLMCQ_EINVAL): skipped
            if rc:
                raise RuntimeError(f"{name} {kn}: rc {rc}")
            return True
        ref = None
        all_knobs = knobs
        knobs = [kn for kn in all_knobs if run(kn, 0)]
        times = {kn: [] for kn in knobs}
        for kn in knobs:
            run(kn, 0)
            torch.cuda.synchronize()
            got = (cods[0].clone(), outs[0].clone() if want_out else None)
            if ref is None:
                ref = got
            else:
                same = torch.equal(got[0], ref[0]) and (not want_out or torch.equal(got[1], ref[1]))
                if not same:
                    print(f"  !! {name} {kn}: result differs from {knobs[0]}")
        for it in range(args.iters):
            for kn in knobs:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run(kn, it % nset)
                b.record()
                torch.cuda.synchronize()
                times[kn].append(a.elapsed_time(b) * 1e3)
        nbytes = in_bytes + wq.numel() + out_elems * (4 * res + 4 * want_out + want_codes)
        macs = out_elems * c * r * r
        line = f"{name:4s} C{c:<4d} {h:>2d}^2 K{k:<4d} {r}x{r} {'res ' if res else '    '}{'out ' if want_out else '    '}"
        for kn in knobs:
            t = sorted(times[kn])[len(times[kn]) // 2]
            line += f" | {':'.join(str(v) for v in kn):>9s} {t:7.1f} us {nbytes / t / 1e3:5.0f} GB/s {2 * macs / t / 1e6:5.0f} TOP/s"
        print(line, flush=True)
        del xs, ress, outs, cods
        torch.cuda.empty_cache()
        knobs = all_knobs


if __name__ == "__main__":
    main()
