#!/usr/bin/env python3
"""Training-step time (forward + backward + SGD) of the three wrapper families - the reference's QAT use of the path
(trainer/quantization_aware_training_trainer.py:51-75: output = model(data); loss.backward(); optimizer.step()).

    python tools/qat_step.py [model] [batch] [--json FILE]

Prints ms per step and images/s per family and - round 5 - the fake-quant kernels' own rates inside one more, instrumented step
(HIP events per launch: forward fq_tensor / fq_channel at 8 algorithmic bytes per element, the one-pass backward `fq_bwd` at 12);
--json writes the whole record (profiles/r05_qat_step_<model>.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import workloads as W  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

argv = [a for a in sys.argv[1:]]
json_path = None
if "--json" in argv:
    i = argv.index("--json")
    json_path = argv[i + 1]
    del argv[i:i + 2]
name = argv[0] if len(argv) > 0 else "resnet18"
batch = int(argv[1]) if len(argv) > 1 else 128
from dlmc.quantization.scalar import kernels as K  # noqa: E402
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
    ms = round((time.perf_counter() - t0) / 10 * 1e3, 2)
    rec = {"ms_per_step": ms, "images_per_s": round(batch / ms * 1e3, 1)}
    if wtype is not None:       # one more step with HIP events on every launch of this project's kernels
        K.PROFILE.reset()
        K.PROFILE.enabled = True
        step()
        torch.cuda.synchronize()
        K.PROFILE.enabled = False
        fam = {}
        for tag, nbytes, e0, e1, _ in K.PROFILE.records:
            f = fam.setdefault(tag, {"launches": 0, "bytes": 0, "ms": 0.0})
            f["launches"] += 1
            f["bytes"] += nbytes
            f["ms"] += e0.elapsed_time(e1)
        K.PROFILE.reset()
        rec["kernels"] = {k: {"launches": f["launches"], "ms": round(f["ms"], 3), "GBps": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1),
                              "frac_of_8TBps": round(f["bytes"] / (f["ms"] * 1e-3) / 8e12, 4)} for k, f in fam.items() if f["ms"] > 0}
        rec["kernels_ms_total"] = round(sum(f["ms"] for f in fam.values()), 3)
    out[family] = rec
    extra = "  ".join(f"{k} {v['GBps']:.0f} GB/s x{v['launches']}" for k, v in rec.get("kernels", {}).items())
    print(f"{name} b{batch} {family:24s} {ms:8.2f} ms/step  {batch / ms * 1e3:8.0f} images/s   {extra}", flush=True)
if json_path:
    with open(json_path, "w") as fh:
        json.dump({"what": "QAT training step (forward + backward + SGD), W4A4, synthetic 224^2 batch; reference loop: "
                           "trainer/quantization_aware_training_trainer.py:51-75", "model": name, "batch": batch,
                   "device": torch.cuda.get_device_name(0), "families": out}, fh, indent=1)
