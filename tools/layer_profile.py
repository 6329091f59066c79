#!/usr/bin/env python3
"""Per-launch timing of the ResNet-50 quantised forward (HIP events around each of this project's kernels):
which layers are far from the roofline.  python tools/layer_profile.py [--int8] [--batch 512]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
import workloads as W  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402
from dlmc.utils.merge_bn import merge_bn  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--int8", action="store_true")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(2333)
model = merge_bn(W.resnet50().to(dev).eval(), inplace=True, allow_missing=True)
quantize_model(model, json.loads(json.dumps(bench.QCFG)), None, quantization_type="FSPTQ", int8_gemm=args.int8)
x = torch.randn(args.batch, 3, 224, 224, device=dev)
if args.int8:
    x = x.contiguous(memory_format=torch.channels_last)
shapes = []
hooks = [m.register_forward_pre_hook(lambda mod, inp, n=n: shapes.append((n, tuple(inp[0].shape), tuple(mod.weight.shape),
                                                                          getattr(mod, "stride", (1,))[0])))
         for n, m in model.named_modules() if hasattr(m, "wt_scale")]
with torch.no_grad():
    model(x)
    model(x)
    shapes.clear()
    K.PROFILE.enabled = True
    K.PROFILE.reset()
    for _ in range(args.iters):
        model(x)
    torch.cuda.synchronize()
K.PROFILE.enabled = False
per = len(K.PROFILE.records) // args.iters
rows = {}
for i, (tag, nbytes, a, b, _) in enumerate(K.PROFILE.records):
    rows.setdefault(i % per, [tag, nbytes, []])[2].append(a.elapsed_time(b) * 1e3)
layers = shapes[:len(shapes) // args.iters]
tot = {}
print(f"{'#':>3} {'kernel':<11} {'MB':>9} {'us(med)':>9} {'GB/s':>8}")
for i in range(per):
    tag, nbytes, ts = rows[i]
    ts.sort()
    med = ts[len(ts) // 2]
    tot.setdefault(tag, [0, 0.0])
    tot[tag][0] += nbytes
    tot[tag][1] += med
    print(f"{i:>3} {tag:<11} {nbytes / 1e6:>9.1f} {med:>9.1f} {nbytes / med / 1e3:>8.0f}")
for tag, (b, t) in tot.items():
    print(f"TOTAL {tag:<11} {b / 1e9:8.2f} GB {t / 1e3:8.2f} ms {b / t / 1e3:8.0f} GB/s")
for n, ish, wsh, st in layers:
    print(n, ish, wsh, st)
