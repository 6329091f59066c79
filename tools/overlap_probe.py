#!/usr/bin/env python3
"""What is running a chain kernel (HBM-bound) and a 3x3 halo kernel (matrix-pipe-bound) AT THE SAME TIME worth?  One ResNet-50 stage's
shapes at half the benchmark's batch (what each of StreamedPlan's two streams launches): `reps` chain launches on one stream, `reps`
3x3 launches on another - each alone, one after the other, and both streams at once.

    python tools/overlap_probe.py [--stage 1|2|3] [--batch 256] [--reps 4]

If "together" is close to max(chain, 3x3) the two kinds of launch hide each other and a plan that keeps its streams half a block apart
gains; if it is close to their sum the chip runs them one after the other whatever the streams say."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=0, help="0: all three")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--iters", type=int, default=7)
args = ap.parse_args()
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)


def layer(ko, r, ci, scale):
    w = torch.randint(-127, 128, (ko, r, r, ci), generator=g, device=dev, dtype=torch.int8)
    return dict(wq=w, wsum=w.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(ko, generator=g, device=dev) * 0.1,
                w_scale=torch.full((ko,), scale, device=dev))


def med(v):
    return sorted(v)[len(v) // 2]


for stage in ([args.stage] if args.stage else [1, 2, 3]):
    c, h, k = {1: (64, 56, 256), 2: (128, 28, 512), 3: (256, 14, 1024)}[stage]
    n = args.batch
    conv3, exp, red = layer(c, 3, c, 0.0004), layer(k, 1, c, 0.0008), layer(c, 1, k, 0.0004)
    red["wq_chunk"] = K.chunk_major(red["wq"])
    s_in = torch.full((1,), 0.02, device=dev)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    xs = [cl(torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8)) for _ in range(args.reps)]
    ys = [cl(torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8)) for _ in range(args.reps)]
    rs = [cl(torch.relu(torch.randn(n, k, h, h, generator=g, device=dev))) for _ in range(args.reps)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def chains():
        for i in range(args.reps):
            K.conv2d_i8_chain(dict(exp, codes=ys[i], in_scale=emit.scale, in_zp=None), red, rs[i], relu=True, emit=emit, want_out=True,
                              want_codes=False, relu2=True, emit2=emit)

    def convs():
        for i in range(args.reps):
            K.conv2d_i8(xs[i], conv3["wq"], conv3["wsum"], conv3["bias"], s_in, None, conv3["w_scale"], padding=1, relu=True, emit=emit, want_out=False)

    def timed(fa, fb):
        """fa on stream sa and fb on stream sb (either may be None), from a common start to both ends: microseconds."""
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(cur)
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        # (the 3x3 stream is issued first: its launches are the short ones, the host must not be what delays them)
        if fb:
            with torch.cuda.stream(sb):
                fb()
        if fa:
            with torch.cuda.stream(sa):
                fa()
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3

    t = {"chain": [], "3x3": [], "both": []}
    for it in range(args.iters + 1):
        a, b, ab = timed(chains, None), timed(None, convs), timed(chains, convs)
        if it:
            t["chain"].append(a), t["3x3"].append(b), t["both"].append(ab)
    a, b, ab = med(t["chain"]), med(t["3x3"]), med(t["both"])
    print(f"stage {stage} ({c} -> {k} at {h}^2, {n} images, {args.reps} launches of each): chains alone {a:7.1f} us   3x3 alone {b:7.1f} us   "
          f"sum {a + b:7.1f}   together {ab:7.1f} us   = max + {100 * (ab - max(a, b)) / min(a, b):.0f} % of the shorter", flush=True)
