#!/usr/bin/env python3
"""Does splitting the batch over S HIP streams (each running the fused plan on its share) overlap the memory-bound and
the matrix-bound layers of different shares?  python tools/stream_probe.py [batch]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import workloads as W  # noqa: E402
from bench import QCFG  # noqa: E402
from dlmc.utils.fuse import fuse_inference  # noqa: E402
from dlmc.utils.merge_bn import merge_bn  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda:0"
torch.manual_seed(2333)
model = merge_bn(W.resnet50().to(dev).eval(), inplace=True, allow_missing=True)
quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=True)
x = torch.relu(torch.randn(batch, 3, 224, 224, device=dev)).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    model(x)
    plan = fuse_inference(model)
    ref = plan(x)
    import itertools
    for S, skew_us in itertools.chain(((s, 0) for s in (1, 2, 3, 4)), ((2, k) for k in (50, 100, 200, 400, 800))):
        streams = [torch.cuda.Stream() for _ in range(S)]
        parts = list(x.chunk(S, dim=0))

        def run():
            outs = [None] * S
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    if i and skew_us:
                        torch.cuda._sleep(int(skew_us * 1e-6 * 2.1e9) * i)     # start this share later (shader clocks)
                    outs[i] = plan(parts[i])
            for s in streams:
                cur.wait_stream(s)
            return torch.cat(outs)
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{S} stream(s) skew {skew_us} us: {ms:.3f} ms  {batch / ms * 1e3:.0f} images/s  identical={torch.equal(out, ref)}", flush=True)
    # free-running shares: no join between steps (each stream runs its share of every step back to back; one join at the end)
    for S in (2, 3):
        streams = [torch.cuda.Stream() for _ in range(S)]
        parts = list(x.chunk(S, dim=0))

        def run_free(steps):
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            outs = [None] * S
            for _ in range(steps):
                for i, s in enumerate(streams):
                    with torch.cuda.stream(s):
                        outs[i] = plan(parts[i])
            for s in streams:
                cur.wait_stream(s)
            return torch.cat(outs)
        run_free(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run_free(10)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{S} free-running stream(s), no join between steps: {ms:.3f} ms  {batch / ms * 1e3:.0f} images/s  identical={torch.equal(out, ref)}", flush=True)
