#!/usr/bin/env python3
"""Round 5, review item 1(b): does a CACHE-RESIDENT schedule pay?  A ResNet-50 stage's mid-section - [3x3 convolution -> chain kernel] x B,
each chain reading the fp32 block tensor the previous chain wrote - walked by sub-batches small enough for that tensor (<= ~100 MB) to
still sit in the 256 MiB Infinity Cache when it is read back, with TEMPORAL fp32 loads / stores (a build with -DDLMCQ_FP32_TEMPORAL:
build_ab/libdlmcq_temporal.so) against the whole batch with the non-temporal hint (the product).

    python tools/cache_resident_lab.py [--lib build_ab/libdlmcq_temporal.so] [--stage 3] [--subs 512,256,128,64] [--blocks 4]

Per sub-batch size: the time of the whole batch of 512 images through the section (sub-batch after sub-batch, one stream), chain launches
and 3x3 launches separately (HIP events).  Run it once per library; compare `chains` at sub = 512 under the product with `chains` at
sub = 128 / 64 under the temporal build.  (rocprofv3 --pmc FETCH_SIZE on the same command shows whether the re-read is served on-die.)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--stage", type=int, default=3, help="3: 256 -> 1024 -> 256 at 14^2 (fp32 tensor 802 KB per image); 2: 128 -> 512 -> 128 at 28^2 (1.6 MB)")
ap.add_argument("--subs", default="512,256,128,64")
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
if args.lib:
    os.environ["DLMCQ_LIBRARY"] = os.path.abspath(args.lib)
    os.environ["DLMCQ_LAB_TOOLS"] = "1"
import torch  # noqa: E402

from dlmc import _native as N  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
c, h, k = {3: (256, 14, 1024), 2: (128, 28, 512)}[args.stage]
n = args.batch


def layer(ko, r, ci, scale):
    w = torch.randint(-127, 128, (ko, r, r, ci), generator=g, device=dev, dtype=torch.int8)
    return dict(wq=w, wsum=w.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(ko, generator=g, device=dev) * 0.1,
                w_scale=torch.full((ko,), scale, device=dev))


blocks = []
for _ in range(args.blocks):
    conv3 = layer(c, 3, c, 0.0004)                 # the block's 3x3 (codes -> codes, halo kernel)
    exp = layer(k, 1, c, 0.0008)                   # its last 1x1 (+ shortcut + ReLU)
    red = layer(c, 1, k, 0.0004)                   # the next block's first 1x1
    red["wq_chunk"] = K.chunk_major(red["wq"])
    blocks.append((conv3, exp, red))
s_in = torch.full((1,), 0.02, device=dev)
emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
emit_s = K.EmitCodes(torch.full((1,), 0.05, device=dev), None, 0, 255, N.FORM_ZEROPOINT, shift128=True)
x0 = torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
r0 = torch.relu(torch.randn(n, k, h, h, generator=g, device=dev)).contiguous(memory_format=torch.channels_last)
flush = torch.empty(384 * 1024 * 1024 // 4, device=dev)    # written between timed passes: nothing of the last pass stays on-die


def section(x, res, ev):
    """x: codes [m, c] of the block's first reduction; res: fp32 block tensor.  Returns the last block tensor."""
    for conv3, exp, red in blocks:
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        _, y = K.conv2d_i8(x, conv3["wq"], conv3["wsum"], conv3["bias"], s_in, None, conv3["w_scale"], padding=1, relu=True, emit=emit, want_out=False)
        e1.record()
        res, _, x = K.conv2d_i8_chain(dict(exp, codes=y, in_scale=emit.scale, in_zp=None), red, res, relu=True, emit=emit, want_out=True,
                                      want_codes=False, relu2=True, emit2=emit)
        e2.record()
        ev.append((e0, e1, e2))
    return res


cl = lambda t: t.contiguous(memory_format=torch.channels_last)
print(f"# stage {args.stage}: [{c} -> 3x3 -> {c}] + chain [{c} -> {k} (+ fp32 shortcut) -> {c}] at {h}^2, {args.blocks} blocks, batch {n}; "
      f"fp32 block tensor {k * h * h * 4 / 1e6:.2f} MB per image; library: {N.LIB_PATH}")
for sub in [int(v) for v in args.subs.split(",")]:
    parts = [(cl(x0[i:i + sub]), cl(r0[i:i + sub])) for i in range(0, n, sub)]
    best = None
    for it in range(args.iters):
        flush.fill_(float(it))
        ev = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for xs, rs in parts:
            section(xs, rs, ev)
        b.record()
        torch.cuda.synchronize()
        t = (a.elapsed_time(b) * 1e3, sum(e1.elapsed_time(e2) for _, e1, e2 in ev) * 1e3, sum(e0.elapsed_time(e1) for e0, e1, _ in ev) * 1e3)
        if it and (best is None or t[0] < best[0]):
            best = t
    print(f"sub-batch {sub:4d} ({sub * k * h * h * 4 / 1e6:6.1f} MB per block tensor, {len(parts)} sub-batches): section {best[0]:8.1f} us   "
          f"chains {best[1]:8.1f} us   3x3 {best[2]:8.1f} us", flush=True)
