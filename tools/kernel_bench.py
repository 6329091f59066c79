#!/usr/bin/env python3
"""Stand-alone kernel timings at BASELINE config 2 (A = 64x256x56x56, W = 256x256x3x3) and a size
sweep: achieved algorithmic GB/s per kernel, HIP events on the launch stream, median of `--iters`.

    python tools/kernel_bench.py [--iters 30] [--sweep]
"""
import argparse
import json
import math
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]

import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

PEAK = 8000.0  # GB/s, HBM3E spec


REPS = 8  # launches per event pair: back-to-back, so host launch latency is hidden behind the GPU queue


def timeit(fn, iters, warmup=3, flush=None):
    for _ in range(warmup):
        fn()
    ts = []
    reps = 1 if flush is not None else REPS
    for _ in range(iters):
        if flush is not None:
            flush()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)  # us per launch
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--cold", action="store_true", help="evict the Infinity Cache between launches")
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(2333)
    A = torch.randn(64, 256, 56, 56, device=dev)
    W = torch.randn(256, 256, 3, 3, device=dev) * math.sqrt(2 / 2304)
    out = torch.empty_like(A)
    outw = torch.empty_like(W)
    junk = torch.empty(512 * 1024 * 1024 // 4, device=dev) if args.cold else None
    flush = (lambda: junk.add_(1.0)) if args.cold else None
    nA, nW = A.numel(), W.numel()
    s_t, o_t = K.observe_qparams(A, 8, True)
    s_c, o_c = K.observe_qparams(A, 8, False, ch_axis=1)
    s_w, _ = K.observe_qparams(W, 8, True, ch_axis=0, scale_eps=1e-6)
    g = 1 / math.sqrt(nA * 127)
    rows = []

    def rec(name, nbytes, fn):
        med, mn = timeit(fn, args.iters, flush=flush)
        rows.append(dict(kernel=name, bytes=nbytes, us_median=round(med, 2), us_min=round(mn, 2),
                         GBps=round(nbytes / med / 1e3, 1), frac_of_8TBps=round(nbytes / med / 1e3 / PEAK, 3)))
        print(json.dumps(rows[-1]), flush=True)

    rec("copy_(A) torch baseline (1R+1W)", 8 * nA, lambda: out.copy_(A))
    rec("fq A per-tensor QBASE", 8 * nA, lambda: K.fake_quant(A, s_t, o_t, -127, 127, N.FORM_QBASE, g=g, out=out))
    rec("fq A per-tensor EMULATE", 8 * nA, lambda: K.fake_quant(A, s_t, o_t, -127, 127, N.FORM_EMULATE, out=out))
    rec("fq A per-channel(ax1) ZEROPOINT", 8 * nA, lambda: K.fake_quant(A, s_c, o_c, 0, 255, N.FORM_ZEROPOINT, out=out))
    rec("fq A per-channel(ax1) EMULATE", 8 * nA, lambda: K.fake_quant(A, s_c, o_c, 0, 255, N.FORM_EMULATE, out=out))
    rec("fq A per-tensor QBASE + int8 codes", 9 * nA,
        lambda: K.fake_quant(A, s_t, o_t, -127, 127, N.FORM_QBASE, g=g, out=out, codes="i8"))
    rec("fq A per-tensor -> int8 codes only", 5 * nA,
        lambda: K.fake_quant(A, s_t, o_t, -127, 127, N.FORM_QBASE, g=g, codes="i8", want_y=False))
    rec("fq W per-channel(ax0) SYMMETRIC", 8 * nW, lambda: K.fake_quant(W, s_w, None, -127, 127, N.FORM_SYMMETRIC, out=outw))
    rec("observer A per-tensor absmax (1R)", 4 * nA, lambda: K.observe_qparams(A, 8, True))
    rec("observer A per-tensor minmax (1R)", 4 * nA, lambda: K.observe_qparams(A, 8, False))
    rec("observer A per-channel(ax1) absmax (1R)", 4 * nA, lambda: K.observe_qparams(A, 8, True, ch_axis=1))
    rec("observer A per-channel(ax1) minmax (1R)", 4 * nA, lambda: K.observe_qparams(A, 8, False, ch_axis=1))
    rec("observer W per-channel(ax0) absmax (1R)", 4 * nW, lambda: K.observe_qparams(W, 8, True, ch_axis=0))

    gy = torch.randn_like(A)
    gx = torch.empty_like(A)

    def bwd():
        from dlmc import _native as NN
        sc = K._scratch(NN.lib.dlmcq_fq_bwd_scratch_bytes(1, 1, nA), A.device)
        gs = torch.empty(1, device=dev)
        NN.check(NN.lib.dlmcq_fake_quant_bwd_f32(NN.ptr(A), NN.ptr(gy), NN.ptr(gx), NN.ptr(gs), NN.ptr(s_t.reshape(1)), NN.ptr(o_t.reshape(1)),
                                                 1, 1, nA, -127, 127, g, NN.ptr(sc), sc.numel() * 4, NN.stream_ptr()))
    rec("fq backward A per-tensor QBASE (2R+1W + scale grad)", 12 * nA, bwd)
    up, lw = torch.tensor(2.5, device=dev), torch.tensor(-2.5, device=dev)
    rec("rootq weight forward on A-sized tensor (1R+1W)", 8 * nA, lambda: K.rootq_weight(A, up, lw, 0, 15))
    _, codes8 = K.fake_quant(A, s_t, o_t, -127, 127, N.FORM_QBASE, g=g, codes="i8", want_y=False)
    rec("dequant int8 codes -> fp32 (1 B R + 4 B W)", 5 * nA, lambda: K.dequant_codes(codes8, A.shape, s_t, o_t, N.FORM_QBASE, "i8", True, g=g))
    c4 = torch.randint(0, 16, (nA,), dtype=torch.int8, device=dev)
    rec("pack int4 (1 B R + 0.5 B W)", int(1.5 * nA), lambda: K.pack_int4(c4))
    p4 = K.pack_int4(c4)
    rec("unpack int4 (0.5 B R + 1 B W)", int(1.5 * nA), lambda: K.unpack_int4(p4, nA, False))

    def obs_fq():
        s, o = K.observe_qparams(A, 8, True)
        K.fake_quant(A, s, o, -127, 127, N.FORM_QBASE, g=g, out=out)
    rec("observer + fq A per-tensor (2R+1W)", 12 * nA, obs_fq)
    rec("torch amax(A) baseline (1R)", 4 * nA, lambda: A.amax())

    if args.sweep:
        for mb in (1, 4, 16, 64, 256, 1024, 2048):
            n = mb * 1024 * 1024 // 4
            x = torch.randn(n, device=dev)
            y = torch.empty_like(x)
            rec(f"fq per-tensor QBASE {mb} MiB", 8 * n, lambda: K.fake_quant(x, s_t, o_t, -127, 127, N.FORM_QBASE, g=g, out=y))
            rec(f"copy_ {mb} MiB", 8 * n, lambda: y.copy_(x))
            del x, y
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kernel_bench.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
