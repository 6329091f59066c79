#!/usr/bin/env python3
"""The five BASELINE.json configurations in one report (GPU numbers; CPU port beside them where it is cheap).

    python tools/run_configs.py > profiles/rNN_configs.json

1  ResNet-18 W8A8 per-tensor symmetric, batch 1        (QBase family; the reference's CPU-runnable case)
2  stand-alone per-channel quant/dequant kernels        -> tools/kernel_bench.py (not repeated here)
3  ResNet-50 W8A8 per-channel, batch 512                (FSPTQ flow, int8 MFMA conv and fp32 conv)
4  RepVGG-A1 (deploy) RepAPQ W8A8, 512 per GPU          (FSPTQ flow; 8-GPU run is the driver's)
5  MobileOne-S1 W4A8 asymmetric per-channel, batch 1024 (QBase family, per-channel extension, fp32 conv)
"""
import copy
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
import workloads as W  # noqa: E402
from dlmc.quantization.scalar import kernels as K  # noqa: E402
from dlmc.utils.merge_bn import merge_bn  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

DEV = "cuda:0"


def time_model(model, x, iters):
    with torch.no_grad():
        model(x)
        model(x)
        K.PROFILE.enabled = True
        K.PROFILE.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        K.PROFILE.enabled = False
    fam = {}
    for tag, nbytes, a, b, _ in K.PROFILE.records:
        f = fam.setdefault(tag, [0, 0.0])
        f[0] += nbytes
        f[1] += a.elapsed_time(b)
    return dt, {k: {"ms_per_step": round(v[1] / iters, 3), "GBps": round(v[0] / (v[1] * 1e-3) / 1e9, 1)} for k, v in fam.items() if v[1] > 0}


def cfg(wtype, wbits, wsigned, abits=8, asigned=False, family_recon=True):
    c = {"weight": {"enable": True, "type": wtype, "args": {"n_bits": wbits, "signed": wsigned}},
         "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": abits, "signed": asigned}},
         "exclude_layers": [], "override_options": []}
    if family_recon:
        c["weight"]["recon_type"] = "None"
    return c


def main():
    out = {}
    torch.manual_seed(2333)
    # ---- config 1
    m = W.resnet18().to(DEV).eval()
    quantize_model(m, cfg("minmax_tensor", 8, True, 8, True, False), None)
    x = torch.randn(1, 3, 224, 224, device=DEV)
    dt, fam = time_model(m, x, 20)
    from dlmc.utils.graph import GraphedForward
    fwd = GraphedForward(m, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fwd(x)
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) / 50 * 1e3
    from oracle.ref_layers import port_model
    torch.set_num_threads(bench.usable_cores())
    mc = port_model(W.resnet18().eval(), "QBase", a_signed=True)
    xc = torch.randn(1, 3, 224, 224)
    with torch.no_grad():
        mc(xc)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            mc(xc)
            ts.append(time.perf_counter() - t0)
    from dlmc.utils.fuse import fuse_inference
    for mod in m.modules():
        if hasattr(mod, "int8_gemm"):
            mod.int8_gemm = True
    plan = fuse_inference(m)
    pfwd = GraphedForward(plan, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        pfwd(x)
    torch.cuda.synchronize()
    plan_graph_ms = (time.perf_counter() - t0) / 50 * 1e3
    out["config1_resnet18_b1_w8a8_per_tensor"] = {"gpu_ms": round(dt * 1e3, 3), "gpu_images_per_s": round(1 / dt, 1),
                                                  "gpu_hipgraph_ms": round(graph_ms, 3),
                                                  "gpu_fused_plan_hipgraph_ms": round(plan_graph_ms, 3),
                                                  "cpu_port_ms": round(statistics.median(ts) * 1e3, 1),
                                                  "cpu_cores": bench.usable_cores(), "kernels": fam}
    # ---- config 3 / 4
    for name, model_name, batch in (("config3_resnet50_b512", "resnet50", 512), ("config4_repvgg_a1_b512_per_gpu", "repvgg_a1", 512)):
        for mode in ("fused_plan", "int8_modules", "fp32conv_modules"):
            int8 = mode != "fp32conv_modules"
            m = merge_bn(W.MODELS[model_name]().to(DEV).eval(), inplace=True, allow_missing=True)
            quantize_model(m, cfg("minmax_channel", 8, True), None, quantization_type="FSPTQ", int8_gemm=int8)
            x = torch.relu(torch.randn(batch, 3, 224, 224, device=DEV))     # unsigned-activation configs
            if int8:
                x = x.contiguous(memory_format=torch.channels_last)
            if mode == "fused_plan":
                with torch.no_grad():
                    m(x)
                m = fuse_inference(m)
            dt, fam = time_model(m, x, 5)
            out[f"{name}_{mode}"] = {"ms_per_step": round(dt * 1e3, 2), "images_per_s": round(batch / dt, 1), "kernels": fam}
            del m, x
            torch.cuda.empty_cache()
    # ---- config 5
    # W4A8, asymmetric per-channel weights (QBase family): the reference's own op sequence (fp32 convolutions of the
    # fake-quantised operands, module by module) and the frozen plan (depthwise layers on conv_dw_i8.hip, pointwise layers on the
    # matrix cores with the weight-offset term, 96-channel tensors padded to 128; the 3-channel first layer on conv_stem_i8.hip's ASYM instantiation)
    for mode in ("fp32conv_modules", "fused_plan"):
        m = W.mobileone_s1_deploy().to(DEV).eval()
        quantize_model(m, cfg("minmax_channel", 4, False, 8, False, False), None)
        x = torch.relu(torch.randn(1024, 3, 224, 224, device=DEV))         # unsigned-activation configs
        with torch.no_grad():
            m(x)
            if mode == "fused_plan":
                from dlmc.utils.fuse import fuse_inference
                m = fuse_inference(m)
        dt, fam = time_model(m, x, 5)
        out[f"config5_mobileone_s1_b1024_w4a8_{mode}"] = {"ms_per_step": round(dt * 1e3, 2), "images_per_s": round(1024 / dt, 1), "kernels": fam}
        del m, x
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
