#!/usr/bin/env python3
"""Training-step time (forward + backward + SGD) of the three wrapper families - the reference's QAT use of the path
(trainer/quantization_aware_training_trainer.py:50-80).  python tools/qat_step.py [model] [batch]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import workloads as W  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda:0"
out = {}
for family, wtype, asigned in (("QBase (LSQ-style)", "minmax_tensor", True), ("RootQ", "minmax_tensor", False), ("FSPTQ", "minmax_channel", False),
                               ("fp32 (no quantisation)", None, None)):
    torch.manual_seed(2333)
    net = W.MODELS[name]().to(dev).train()
    if wtype is not None:
        cfg = {"weight": {"enable": True, "type": wtype, "args": {"n_bits": 4, "signed": True}},
               "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 4, "signed": asigned}},
               "momentum": 0.1, "exclude_layers": [], "override_options": []}
        quantize_model(net, cfg, None, {"QBase (LSQ-style)": None, "RootQ": "RootQ", "FSPTQ": "FSPTQ"}[family])
    opt = torch.optim.SGD(net.parameters(), lr=1e-4)
    x = torch.randn(batch, 3, 224, 224, device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(net(x), y).backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    out[family] = round((time.perf_counter() - t0) / 10 * 1e3, 2)
    print(f"{name} b{batch} {family:24s} {out[family]:8.2f} ms/step  {batch / out[family] * 1e3:8.0f} images/s", flush=True)
