#!/usr/bin/env python3
"""What bounds the halo-tile 3x3 kernel (csrc/conv3x3_i8.hip)?  Through the LAB library (`make -C dlmc-quant_amd/csrc lab`):
timing-only ablations (no weight DMA / no halo DMA / no MFMAs / no xor / no quantiser / unstaged stores) interleaved with the
product kernel in one process over rotating buffers, and a stamped build that writes the shader clock at the phase boundaries of
one wave's K steps and at every workgroup's start / loop / end.

    python tools/halo_lab.py [--batch 512] [--cases c2,c3,c4] [--stamps]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402

CASES = {"c1": (64, 56, 64), "c1w": (64, 56, 128), "c2": (128, 28, 128), "c3": (256, 14, 256), "c4": (512, 7, 512)}   # C, H, K
VARIANTS = {0: "product", 100: "8 waves of 64 x 64", 2: "no weight DMA", 3: "no halo DMA", 4: "no MFMA", 6: "no xor", 7: "no quantiser",
            8: "codes not staged", 9: "2nd slot starts 1/2 tile late", 10: "2nd slot starts 1/4 tile late", 11: "2nd slot starts 1 tile late",
            12: "slot s of 4 starts s x 8 k clocks late", 13: "slot s of 4 starts s x 4 k clocks late"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--cases", default="c2,c3,c4")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--generic", action="store_true", help="also time the generic kernel (128-wide, A direct)")
    args = ap.parse_args()
    lab = ctypes.CDLL(os.path.join(ROOT, "dlmc-quant_amd", "libdlmcq_lab.so"))
    fn = lab.dlmcq_x_conv2d_i8_tuned
    fn.restype = ctypes.c_int
    p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    fn.argtypes = [p] * 8 + [i64] * 7 + [i32] * 4 + [p, i32, p, p, p, i32, i32, i32, f32, p, i32, i32, i32, i32]
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(2333)
    for name in args.cases.split(","):
        c, h, k = CASES[name]
        n = args.batch
        nset = max(2, int(300e6 // (n * h * h * (c + k))) + 1)
        xs = [torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
              for _ in range(nset)]
        wq = torch.randint(-127, 128, (k, 3, 3, c), generator=g, device=dev, dtype=torch.int8)
        wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
        s_w = (torch.rand(k, generator=g, device=dev) * 0.01 + 0.001).contiguous()
        bias = torch.randn(k, generator=g, device=dev)
        s_in = torch.full((1,), 0.02, device=dev)
        zp = torch.full((1,), 3.0, device=dev)
        q_s = torch.full((1,), 0.5, device=dev)
        q_z = torch.zeros(1, device=dev)
        cods = [torch.empty(n, k, h, h, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last) for _ in range(nset)]
        trace = torch.zeros(64 * 8 + 6 * 65536, dtype=torch.int64, device=dev)

        def run(v, i):
            bn, adir, wps, res = (128 if k % 128 == 0 else 64), 5, 0, None
            if v == -1:
                adir = 1                       # the generic kernel
            elif v > 0:
                wps = -100 - v
                res = N.ptr(trace) if v == 1 else None
            rc = fn(N.ptr(xs[i]), N.ptr(wq), None, N.ptr(bias), N.ptr(wsum), N.ptr(s_in), N.ptr(zp), N.ptr(s_w), n, h, h, c, k, 3, 3,
                    1, 1, 1, 1, res, 1, N.ptr(cods[i]), N.ptr(q_s), N.ptr(q_z), 0, 255, N.FORM_ZEROPOINT, 0.0, N.stream_ptr(),
                    bn, adir, 0, wps)
            if rc:
                raise RuntimeError(f"{name} variant {v}: rc {rc}")
        vs = ([-1] if args.generic else []) + [v for v in VARIANTS if (v < 100 and (v < 9 or (k % 128 == 0 and v < 12) or (v >= 12 and c == 64 and k == 64))) or (v >= 100 and k % 128 == 0)]
        times = {v: [] for v in vs}
        for v in vs:
            run(v, 0)
        torch.cuda.synchronize()
        for it in range(args.iters):
            for v in vs:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run(v, it % nset)
                b.record()
                torch.cuda.synchronize()
                times[v].append(a.elapsed_time(b) * 1e3)
        macs = n * h * h * k * c * 9
        print(f"{name}: C{c} {h}^2 K{k} batch {n}")
        for v in vs:
            t = sorted(times[v])[len(times[v]) // 2]
            label = "generic kernel" if v == -1 else VARIANTS[v]
            print(f"    {label:20s} {t:7.1f} us   {2 * macs / t / 1e6:6.0f} TOP/s (real pixels)", flush=True)
        if args.stamps:
            trace.zero_()
            run(1, 0)
            torch.cuda.synchronize()
            tr = trace.cpu()
            st = tr[:64 * 8].view(64, 8)
            nstep = int((st[:, 4] > 0).sum())
            print(f"    stamps of one wave (clocks): wait / barrier / issue / multiply / (step total)   [{nstep} steps]")
            for sidx in range(min(nstep, 40)):
                r = st[sidx]
                nxt = st[sidx + 1][0] if sidx + 1 < nstep else r[4]
                print(f"      step {sidx:2d}: {int(r[1] - r[0]):6d} {int(r[2] - r[1]):6d} {int(r[3] - r[2]):6d} {int(r[4] - r[3]):6d}   ({int(nxt - r[0]):6d})")
            fs = (h + 1) * (h + 1)
            nwg = ((n * fs + 255) // 256) * (k // (128 if k % 128 == 0 else 64))
            wg = tr[64 * 8:64 * 8 + 6 * nwg].view(nwg, 6)
            pro = (wg[:, 1] - wg[:, 0]).float()
            loop = (wg[:, 2] - wg[:, 1]).float()
            epi = (wg[:, 3] - wg[:, 2]).float()
            life = (wg[:, 3] - wg[:, 0]).double()
            real = (wg[:, 5] - wg[:, 4]).double()              # 100 MHz ticks
            ghz = (life / real.clamp(min=1) * 0.1)
            span = (int(wg[:, 5].max()) - int(wg[:, 4].min())) / 100.0   # us, first start to last end (the constant clock is chip-wide)
            print(f"    {nwg} workgroups: prologue {pro.median():.0f} (max {pro.max():.0f}), K loop {loop.median():.0f} (min {loop.min():.0f} max {loop.max():.0f}), "
                  f"epilogue {epi.median():.0f} (max {epi.max():.0f}), lifetime {life.median():.0f} clocks = {real.median() / 100:.1f} us; "
                  f"shader clock {ghz.median():.2f} GHz; first start to last end {span:.1f} us")
            starts = ((wg[:, 4] - int(wg[:, 4].min())).double() / 100.0)
            print(f"    workgroup starts (us): median {starts.median():.1f}, 75 % {starts.quantile(0.75):.1f}, last {starts.max():.1f}")
        del xs, cods
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
