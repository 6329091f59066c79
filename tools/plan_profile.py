#!/usr/bin/env python3
"""Per-layer timing of the fused plan (HIP events around every plan node).  python tools/plan_profile.py [model] [batch]
(model mobileone_s1 is profiled as BASELINE config 5: QBase W4A8, asymmetric per-channel weights)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]
import torch  # noqa: E402

import workloads as W  # noqa: E402
from bench import QCFG  # noqa: E402
from dlmc.utils.fuse import ChainInt8Layer, DualInt8Layer, DwPwInt8Layer, _PlanLayer as Int8Layer, fuse_inference  # noqa: E402
from dlmc.utils.merge_bn import merge_bn  # noqa: E402
from dlmc.utils.quantize import quantize_model  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = "cuda:0"
torch.manual_seed(2333)
if name == "mobileone_s1":
    model = W.MODELS[name]().to(dev).eval()
    quantize_model(model, {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 4, "signed": False}},
                           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
                           "exclude_layers": [], "override_options": []}, None)
    x = torch.relu(torch.randn(batch, 3, 224, 224, device=dev))
else:
    model = merge_bn(W.MODELS[name]().to(dev).eval(), inplace=True, allow_missing=True)
    quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(batch, 3, 224, 224, device=dev)).contiguous(memory_format=torch.channels_last)
recs = []
with torch.no_grad():
    model(x)
    plan = fuse_inference(model)
    for _ in range(2):
        plan(x)

    def pre(mod, args):
        mod._ev = torch.cuda.Event(enable_timing=True)
        mod._ev.record()

    def post(mod, args, out):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        recs.append((mod, args, out, mod._ev, e))
    duals = [m for m in plan.modules() if isinstance(m, DualInt8Layer)]
    chains = [m for m in plan.modules() if isinstance(m, ChainInt8Layer)]
    inner = set()
    for d in duals:
        inner |= {id(d.a), id(d.b)}
    for c in chains:
        inner |= {id(c.a), id(c.b), id(c.main), id(c.short)}
    for p_ in plan.modules():
        if isinstance(p_, DwPwInt8Layer):
            inner |= {id(p_.dw), id(p_.pw)}
    for m in plan.modules():
        if isinstance(m, (DualInt8Layer, ChainInt8Layer, DwPwInt8Layer)) or (isinstance(m, Int8Layer) and id(m) not in inner):
            m.register_forward_pre_hook(pre)
            m.register_forward_hook(post)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    plan(x)
    b.record()
    torch.cuda.synchronize()
print(f"whole forward {a.elapsed_time(b):.2f} ms")
tot = 0.0
for mod, args, out, s, e in recs:
    us = s.elapsed_time(e) * 1e3
    tot += us
    if isinstance(mod, DwPwInt8Layer):
        xin, o = args[0], out[1]
        wd, wp = mod.dw.layer.weight, mod.pw.layer.weight
        m_ = o.numel() // o.shape[1]
        macs = m_ * (wd.shape[0] * 9 + wp.numel())
        nb = xin.numel() + o.numel() + wd.numel() + wp.numel()
        print(f"{us:8.1f} us  DWPW in {str(tuple(xin.shape)):20s} dw {str(tuple(wd.shape)):16s} -> pw {str(tuple(wp.shape)):18s} relu     codes  "
              f"{2 * macs / us / 1e6:6.0f} TOP/s {nb / us / 1e3:6.0f} GB/s")
        continue
    if isinstance(mod, ChainInt8Layer):
        w1, w3 = mod.main.layer.weight, mod.b.layer.weight
        o = out[2]
        m_ = o.numel() // o.shape[1]
        macs = m_ * (w1.numel() + w3.numel())
        xm, xs = (args[1], args[0]) if mod.swapped else (args[0], args[1])     # (a swapped dual chain reads its main operand second)
        nb = xm.numel() + w1.numel() + w3.numel() + m_ * w1.shape[0] * (4 * (out[0] is not None) + (out[1] is not None)) + o.numel()
        cmaj = lambda t: type(t).__name__ == "ChunkMajor"       # (the block tensor between two chain kernels: kernels.ChunkMajor)
        what = f"+ fp32 shortcut{'*' if cmaj(args[1]) else ''} {tuple(args[1].shape)}"
        if mod.short is not None:
            w2 = mod.short.layer.weight
            macs += m_ * w2.numel()
            nb += xs.numel() // (mod.short.layer.stride[0] ** 2) + w2.numel()     # the strided shortcut convolution samples 1 / stride^2 of its input
            what = f"+ {tuple(xs.shape)} x {tuple(w2.shape)}"
        else:
            nb += m_ * w1.shape[0] * 4
        print(f"{us:8.1f} us  CHAIN {str(tuple(xm.shape)):20s} x {str(tuple(w1.shape)):18s} {what:42s} -> x {str(tuple(w3.shape)):18s} "
              f"{('out*' if cmaj(out[0]) else 'out ') if out[0] is not None else '    '}{'codes ' if out[1] is not None else '      '}{2 * macs / us / 1e6:6.0f} TOP/s {nb / us / 1e3:6.0f} GB/s")
        continue
    if isinstance(mod, DualInt8Layer):
        wa, wb = mod.a.layer.weight, mod.b.layer.weight
        o = out[0] if out[0] is not None else out[1]
        macs = o.numel() * (wa.numel() // wa.shape[0] + wb.numel() // wb.shape[0])
        nb = args[0].numel() + args[1].numel() // (mod.b.layer.stride[0] ** 2) + wa.numel() + wb.numel() + o.numel() * (
            4 * (out[0] is not None) + (out[1] is not None))
        print(f"{us:8.1f} us  DUAL {str(tuple(args[0].shape)):20s} x {str(tuple(wa.shape)):18s} + {str(tuple(args[1].shape)):20s} x "
              f"{str(tuple(wb.shape)):18s} {'out ' if out[0] is not None else '    '}codes  {2 * macs / us / 1e6:6.0f} TOP/s {nb / us / 1e3:6.0f} GB/s")
        continue
    lay = mod.layer
    xin = args[0]
    o = out[0] if out[0] is not None else out[1]
    w = lay.weight
    macs = o.numel() * (w.numel() // w.shape[0])
    nbytes = xin.numel() * (1 if xin.dtype != torch.float32 else 5) + w.numel() + o.numel() * (
        4 * (out[0] is not None) + (out[1] is not None) + 4 * (len(args) > 1))
    print(f"{us:8.1f} us  in {str(tuple(xin.shape)):22s} {str(xin.dtype)[6:]:8s} w {str(tuple(w.shape)):20s} "
          f"{'res ' if len(args) > 1 else '    '}{'relu ' if mod.relu else '     '}{'out ' if out[0] is not None else '    '}"
          f"{'codes' if out[1] is not None else '     '}  {2 * macs / us / 1e6:6.0f} TOP/s {nbytes / us / 1e3:6.0f} GB/s")
print(f"plan nodes {tot / 1e3:.2f} ms")
