#!/usr/bin/env python3
"""Where does the calibrating ("first batch") forward spend its time?  ResNet-50 b512, every observer re-armed:
  as is (one host read of the zero point per layer) / the read skipped (decision assumed) / host time only (no GPU wait)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import workloads as W  # noqa: E402
from bench import QCFG  # noqa: E402
from dlmc.quantization.scalar.FSPTQuant.base import FSPTQBase  # noqa: E402
from dlmc.utils.merge_bn import merge_bn  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

dev = "cuda:0"
torch.manual_seed(2333)
model = merge_bn(W.resnet50().to(dev).eval(), inplace=True, allow_missing=True)
quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=True)
x = torch.relu(torch.randn(512, 3, 224, 224, device=dev)).contiguous(memory_format=torch.channels_last)


def rearm():
    for m in model.modules():
        if hasattr(m, "_init") and hasattr(m, "in_init_state"):
            m._init.mark(m, "in_init_state", False)
            m._init.mark(m, "wt_init_state", False)


def timed(tag):
    rearm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        model(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{tag}: host returns after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms", flush=True)


with torch.no_grad():
    model(x)
timed("as is")
timed("as is")
orig = FSPTQBase._int8_applicable


def no_read(self, input):
    if self._zp_is_int is None:
        self._zp_is_int = True
    return orig(self, input)


FSPTQBase._int8_applicable = no_read
timed("zero-point read skipped")
timed("zero-point read skipped")
FSPTQBase._int8_applicable = orig
# steady state module path for comparison
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.no_grad():
    model(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"steady-state module path: host {1e3 * (t1 - t0):.1f} ms, GPU done {1e3 * (time.perf_counter() - t0):.1f} ms")
if os.environ.get("PROBE_LOOP"):
    for _ in range(int(os.environ["PROBE_LOOP"])):
        timed("loop")
