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
VARIANTS = ["pw", "pw2", "pw4", "pw6", "pw7", "pw8", "pw9", 0, 2, 3, 4, 6, 7]      # "pw": csrc/conv_pw_i8.hip (what the product launches for these layers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--only", default="", help="comma-separated variants to run (default: all), e.g. pw,0")
    ap.add_argument("--trace", action="store_true", help="print one wave's clock stamps (LAB 1 build of the pointwise kernel)")
    args = ap.parse_args()
    global VARIANTS
    if args.only:
        VARIANTS = [v if v.startswith("pw") else int(v) for v in args.only.split(",")]
    lab = ctypes.CDLL(os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so"))
    fn = lab.dlmcq_x_conv2d_i8_tuned
    fn.restype = ctypes.c_int
    p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    fn.argtypes = [p] * 8 + [i64] * 7 + [i32] * 4 + [p, i32, p, p, p, i32, i32, i32, f32, p, i32, i32, i32, i32]
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(2333)
    print("case            " + "".join(f"{(v if isinstance(v, str) else 'LAB ' + str(v)):>9s}" for v in VARIANTS) + "   (us)")
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
        skip = set()
        def launch(v, i):
            if isinstance(v, str) and v != "pw":       # the pointwise kernel's own variants (2 no loads, 4 no MFMAs, 6 / 7 / 8 epilogue pieces, 9 nt stores)
                rc = fn(N.ptr(xs[i]), N.ptr(wq), None, N.ptr(woff), N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, 1, 1, 1, 0, 1, 1,
                        None, 1, N.ptr(cods[i]), N.ptr(q_s), None, 0, 255, N.FORM_ZEROPOINT, 0.0, N.stream_ptr(), bn, 1, 0, -40 - int(v[2:]))
                if rc == -1:
                    rc = 0
                    skip.add(v)
                return rc
            if v == "pw":
                return lab.dlmcq_conv2d_i8_nhwc_asym(N.ptr(xs[i]), N.ptr(wq), None, None, N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), N.ptr(woff),
                                                     i64(n), i64(h), i64(h), i64(c), i64(k), i64(1), i64(1), 1, 0, 1, 1, None, 1, N.ptr(cods[i]),
                                                     N.ptr(q_s), None, 0, 255, N.FORM_ZEROPOINT, f32(0.0), N.stream_ptr())
            return fn(N.ptr(xs[i]), N.ptr(wq), None, N.ptr(woff), N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, 1, 1, 1, 0, 1, 1,
                      None, 1, N.ptr(cods[i]), N.ptr(q_s), None, 0, 255, N.FORM_ZEROPOINT, 0.0, N.stream_ptr(), bn, 1, 0, -20 - v)
        # every timing: `reps` launches back to back on rotating buffers between two events (a single launch behind a synchronize
        # also measures the host's launch path: ~35 us for the pointwise kernel's 100 KB dynamic LDS, ~5 for the tiled kernel)
        for it in range(args.iters + 1):
            for v in VARIANTS:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for rep in range(args.reps):
                    rc = launch(v, (it * args.reps + rep) % nset)
                    if rc:
                        raise RuntimeError(f"{name} LAB {v}: rc {rc}")
                b.record()
                torch.cuda.synchronize()
                if it:
                    times[v].append(a.elapsed_time(b) * 1e3 / args.reps)
        med = {v: (float("nan") if v in skip else sorted(t)[len(t) // 2]) for v, t in times.items()}
        if args.trace and "pw" not in skip:
            tr = torch.zeros(256 + 4 * 4096, dtype=torch.int64, device=dev)
            for _ in range(2):
                rc = fn(N.ptr(xs[0]), N.ptr(wq), None, N.ptr(woff), N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, 1, 1, 1, 0, 1, 1,
                        N.ptr(tr), 1, N.ptr(cods[0]), N.ptr(q_s), None, 0, 255, N.FORM_ZEROPOINT, 0.0, N.stream_ptr(), bn, 1, 0, -41)
            torch.cuda.synchronize()
            t = [int(v) for v in tr.cpu()]
            if rc == 0 and t[2] > 5:
                ns, clk = t[2], t[3:t[2]]
                us = (t[1] - t[0]) / 100.0
                print(f"  {name}: one wave, entry to exit {us:.1f} us = {clk[-1] - clk[0]} clocks ({(clk[-1] - clk[0]) / us / 1e3:.2f} GHz); prologue {clk[1] - clk[0]} clocks;"
                      f" then per block (loop top, fragments landed, K loop / epilogue per pass, stores issued), clocks since the loop was entered:")
                print("   " + " ".join(str(v - clk[1]) for v in clk[2:]))
                wg = torch.tensor(t[256:]).reshape(-1, 4)
                wg = wg[wg[:, 0] > 0]
                t00 = int(wg[:, 0].min())
                st, en = (wg[:, 0] - t00).double() / 100.0, (wg[:, 1] - t00).double() / 100.0
                xcc, cu, se = wg[:, 3] & 15, (wg[:, 2] >> 8) & 15, (wg[:, 2] >> 13) & 7
                places = len({(int(x), int(s_), int(c_)) for x, s_, c_ in zip(xcc, se, cu)})
                print(f"   {len(wg)} workgroups on {places} (XCC, SE, CU) places; start {st.min():.1f} .. {st.max():.1f} us (median {st.median():.1f}), "
                      f"end {en.min():.1f} .. {en.max():.1f} us (median {en.median():.1f}); lifetime median {(en - st).median():.1f}, max {(en - st).max():.1f} us")
                late = st > st.median() + 5
                print(f"   {int(late.sum())} workgroups start more than 5 us after the median start")
        nbytes = n * h * h * (c + k)
        print(f"{name:16s}" + "".join(f"{med[v]:9.1f}" for v in VARIANTS) +
              (f"   tiled as built: {nbytes / med[0] / 1e3:5.0f} GB/s {2 * n * h * h * c * k / med[0] / 1e6:5.0f} TOP/s;" if 0 in med else "") +
              (f" pointwise kernel: {nbytes / med['pw'] / 1e3:5.0f} GB/s" if "pw" in med else ""), flush=True)
        del xs, cods
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
