#!/usr/bin/env python3
"""What bounds the pointwise (1x1, asymmetric weights, codes in -> codes out) layers of MobileOne-S1: the swapped asymmetric
int8 kernel as built and with one ingredient removed at a time (lab library, `make -C dlmc-quant_amd/csrc lab`).

    python tools/pw_lab.py [--batch 1024] [--cases 192x28,512x14,...]

Variants: pw = the pointwise kernel of csrc/conv_pw_i8.hip; of the tiled kernel: 0 as built, 2 no activation loads, 3 no weight loads, 4 no MFMAs, 6 epilogue without its arithmetic, 7 no epilogue.
Results of the variants are garbage; only the time means something.  Rotating buffers past the Infinity Cache.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402

# name: (C, H, K, tile width)
CASES = {
    "96x56": (128, 56, 128, 128),        # (the plan pads 96 -> 128 channels)
    "192x28": (192, 28, 192, 192),
    "192to512x14": (192, 14, 512, 128),
    "512x14": (512, 14, 512, 128),
    "512to1280x7": (512, 7, 1280, 128),
}
VARIANTS = ["pw", 0, 2, 3, 4, 6, 7]      # "pw": csrc/conv_pw_i8.hip (what the product launches for these layers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--iters", type=int, default=6)
    args = ap.parse_args()
    lab = ctypes.CDLL(os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so"))
    fn = lab.dlmcq_x_conv2d_i8_tuned
    fn.restype = ctypes.c_int
    p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    fn.argtypes = [p] * 8 + [i64] * 7 + [i32] * 4 + [p, i32, p, p, p, i32, i32, i32, f32, p, i32, i32, i32, i32]
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(2333)
    print("case            " + "".join(f"{(v if v == 'pw' else 'LAB ' + str(v)):>12s}" for v in VARIANTS) + "   (us)")
    for name in args.cases.split(","):
        c, h, k, bn = CASES[name]
        n = args.batch
        nset = max(2, int(300e6 // (n * h * h * (c + k))) + 1)
        xs = [torch.randint(0, 256, (n, h, h, c), generator=g, device=dev, dtype=torch.uint8) for _ in range(nset)]
        cods = [torch.empty(n, h, h, k, device=dev, dtype=torch.uint8) for _ in range(nset)]
        wq = torch.randint(-8, 8, (k, c), generator=g, device=dev, dtype=torch.int8)
        wsum = wq.to(torch.int32).sum(dim=1).to(torch.int32).contiguous()
        s_w = (torch.rand(k, generator=g, device=dev) * 0.01 + 0.001).contiguous()
        woff = torch.randn(k, generator=g, device=dev) * 0.01
        s_in = torch.full((1,), 0.02, device=dev)
        zp = torch.zeros(1, device=dev)
        q_s = torch.full((1,), 0.05, device=dev)
        q_z = torch.zeros(1, device=dev)
        times = {v: [] for v in VARIANTS}
        for it in range(args.iters + 1):
            for v in VARIANTS:
                i = it % nset
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if v == "pw":
                    rc = lab.dlmcq_conv2d_i8_nhwc_asym(N.ptr(xs[i]), N.ptr(wq), None, None, N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), N.ptr(woff),
                                                       i64(n), i64(h), i64(h), i64(c), i64(k), i64(1), i64(1), 1, 0, 1, 1, None, 1, N.ptr(cods[i]),
                                                       N.ptr(q_s), N.ptr(q_z), 0, 255, N.FORM_ZEROPOINT, f32(0.0), N.stream_ptr())
                else:
                    rc = fn(N.ptr(xs[i]), N.ptr(wq), None, N.ptr(woff), N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, 1, 1, 1, 0, 1, 1,
                            None, 1, N.ptr(cods[i]), N.ptr(q_s), N.ptr(q_z), 0, 255, N.FORM_ZEROPOINT, 0.0, N.stream_ptr(), bn, 1, 0, -20 - v)
                b.record()
                torch.cuda.synchronize()
                if rc:
                    raise RuntimeError(f"{name} LAB {v}: rc {rc}")
                if it:
                    times[v].append(a.elapsed_time(b) * 1e3)
        med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
        nbytes = n * h * h * (c + k)
        print(f"{name:16s}" + "".join(f"{med[v]:12.1f}" for v in VARIANTS) +
              f"   tiled as built: {nbytes / med[0] / 1e3:5.0f} GB/s {2 * n * h * h * c * k / med[0] / 1e6:5.0f} TOP/s;"
              f" pointwise kernel: {nbytes / med['pw'] / 1e3:5.0f} GB/s", flush=True)
        del xs, cods
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
