#!/usr/bin/env python3
"""The calibrating ("first batch") forward of ResNet-50 b512 through dlmc.utils.fuse.EagerFused, every observer re-armed: wall time, and
per kernel family of this project the launches / bytes / HIP-event time of ONE such forward (the rest is torch: max-pool, adaptive pool,
host waits).    python tools/first_batch_profile.py [reps]          (under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import workloads as W  # noqa: E402
from bench import QCFG  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402
from dlmc.utils.fuse import EagerFused  # noqa: E402
from dlmc.utils.merge_bn import merge_bn  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda:0"
torch.manual_seed(2333)
model = merge_bn(W.resnet50().to(dev).eval(), inplace=True, allow_missing=True)
quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=True)
x = torch.relu(torch.randn(512, 3, 224, 224, device=dev))


def rearm():
    for m in model.modules():
        if hasattr(m, "_init") and hasattr(m, "in_init_state"):
            m._init.mark(m, "in_init_state", False)
            m._init.mark(m, "wt_init_state", False)


with torch.no_grad():
    model(x)
    eager = EagerFused(model)
    rearm()
    eager(x)
    for r in range(reps):
        rearm()
        torch.cuda.synchronize()
        K.PROFILE.reset()
        K.PROFILE.enabled = r == reps - 1
        t0 = time.perf_counter()
        eager(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        K.PROFILE.enabled = False
        print(f"first batch {r}: host returns after {1e3 * (t1 - t0):.2f} ms, GPU done after {1e3 * (t2 - t0):.2f} ms", flush=True)
fam = {}
for tag, nbytes, a, b, _ in K.PROFILE.records:
    f = fam.setdefault(tag, [0, 0, 0.0])
    f[0] += 1
    f[1] += nbytes
    f[2] += a.elapsed_time(b)
tot = 0.0
for tag, (n, nb, ms) in sorted(fam.items(), key=lambda kv: -kv[1][2]):
    tot += ms
    print(f"  {tag:14s} {n:4d} launches {nb / 1e9:8.2f} GB {ms:8.3f} ms {nb / ms / 1e6 if ms else 0:8.0f} GB/s")
print(f"  this project's kernels together {tot:.3f} ms (the last forward ran with HIP events on every launch)")
