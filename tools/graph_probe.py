import json, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch
import workloads as W
from bench import QCFG
from dlmc.utils.fuse import fuse_inference
from dlmc.utils.graph import GraphedForward
from dlmc.utils.merge_bn import merge_bn
from dlmc.utils.quantize import quantize_model
dev = "cuda:0"
torch.manual_seed(2333)
model = merge_bn(W.resnet50().to(dev).eval(), inplace=True, allow_missing=True)
quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=True)
x = torch.relu(torch.randn(512, 3, 224, 224, device=dev)).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    model(x)
    plan = fuse_inference(model)
    for _ in range(3):
        plan(x)
    def t(f, n=20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f(x)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    e = t(plan)
    g = GraphedForward(plan, x)
    gm = t(g)
print(f"eager {e:.3f} ms  graph {gm:.3f} ms")
