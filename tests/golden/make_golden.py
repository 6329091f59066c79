#!/usr/bin/env python3
"""Generate the golden vectors that pin the oracle (and the HIP path) to the reference.

Run ONLY in the build container, where the reference checkout is mounted at
/root/reference.  The reference's Python never travels to the GPU box; what travels is the
output of this script: `tests/golden/golden_v1.npz` (inputs + expected outputs) and
`tests/golden/golden_v1.json` (case descriptors).

    python tests/golden/make_golden.py                  # golden_v1
    python tests/golden/make_golden.py --supplement     # golden_v1_grad (round 2)
    python tests/golden/make_golden.py --supplement2    # golden_v2: LSQ initialisation, minmax_pixel (round 4)

How the reference is imported: the package does not import through
its normal graph (`trainer/__init__.py` pulls torchvision and five missing modules), so
empty `trainer` / `trainer.loss` package shells are registered and `trainer/loss/loss.py`
is loaded by path; then `dlmc.quantization.scalar.{ops,utils,modules,RootQ,FSPTQuant}`
import cleanly.  Wrappers are constructed the way `dlmc/utils/quantize.py:130-133` does
(`__new__` + `__dict__.update` + `initialize`), never through their constructors.
`FSPTQBase.initialize` hard-codes `device('cuda')` (FSPTQuant/base.py:47); for the duration
of that one call `torch.zeros` is wrapped to drop the `device=` keyword so the reference's own
`initialize` still runs on CPU.

Every expected array below is produced by the reference's code, not by the oracle.
"""
import copy
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 2333  # the reference's own seed (example/quantization/QAT_config.yaml:8)


def import_reference():
    sys.path.insert(0, REF)
    for name in ("trainer", "trainer.loss"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location(
        "trainer.loss.loss", os.path.join(REF, "trainer/loss/loss.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["trainer.loss.loss"] = mod
    spec.loader.exec_module(mod)
    from dlmc.quantization.scalar import ops, utils, modules, RootQ, FSPTQuant
    from dlmc.quantization.scalar.modules import function as mfunction
    from dlmc.quantization.scalar.RootQ import function as rfunction
    return ops, utils, modules, RootQ, FSPTQuant, mfunction, rfunction


ops, utils, modules, RootQ, FSPTQuant, mfunction, rfunction = import_reference()

ARR = {}
CASES = []


def put(name, **arrays):
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        ARR[f"{name}.{k}"] = np.asarray(v)


def gen(seed_offset=0):
    g = torch.Generator()
    g.manual_seed(SEED + seed_offset)
    return g


def special_values(scale, offset, lo, hi):
    """Values that hit ties, saturation, signed zeros, denormals and non-finite inputs."""
    ks = [lo - 3, lo - 0.5, lo, lo + 0.5, -2.5, -1.5, -0.5, -0.25, 0.0, 0.25, 0.5, 1.5, 2.5,
          hi - 0.5, hi, hi + 0.5, hi + 3]
    v = [k * scale + offset for k in ks]
    v += [0.0, -0.0, 1e-41, -1e-41, 1.17549435e-38, 3.0e38, -3.0e38,
          float("inf"), float("-inf"), float("nan")]
    return torch.tensor(v, dtype=torch.float32)


RANGES = [(True, 8), (False, 8), (True, 4), (False, 4), (False, 3), (False, 2), (True, 2)]


# ----------------------------------------------------------------------------- primitives
def case_primitives():
    """utils.py:1-22 quantize / dequantize / emulate_quantize / get_qrange."""
    idx = 0
    for signed, n_bits in RANGES:
        lo, hi = utils.get_qrange(signed, n_bits)
        # --- per-tensor, exact power-of-two scale so ties are real ties
        for scale, offset in ((0.25, 0.0), (0.0123, 0.0), (0.037, -0.41), (2.0 ** -20, 0.0)):
            g = gen(idx)
            x = torch.cat([torch.randn(300, generator=g) * scale * hi * 0.6 + offset,
                           special_values(scale, offset, lo, hi)])
            s = torch.tensor(scale, dtype=torch.float32)
            o = torch.tensor(offset, dtype=torch.float32)
            name = f"prim_t{idx}"
            put(name, x=x, scale=s, offset=o,
                q=utils.quantize(x, s, o, lo, hi),
                y=utils.emulate_quantize(x, s, o, lo, hi))
            CASES.append(dict(name=name, kind="primitive", layout="tensor", signed=signed,
                              n_bits=n_bits, lo=lo, hi=hi))
            idx += 1
        # --- per-channel, weights KCRS (scale [K,1,1,1]) and activations NCHW (scale [1,C,1,1])
        for layout, shape, ch_axis in (("kcrs", (6, 4, 3, 3), 0), ("nchw", (3, 5, 4, 4), 1),
                                       ("nchw7", (2, 3, 7, 7), 1), ("nc", (5, 7), 1),
                                       ("kc", (6, 10), 0)):
            g = gen(100 + idx)
            x = torch.randn(shape, generator=g)
            C = shape[ch_axis]
            bshape = [1] * len(shape)
            bshape[ch_axis] = C
            s = (torch.rand(C, generator=g) * 0.05 + 0.001).reshape(bshape)
            s.view(-1)[0] = 0.0  # a dead channel: scale 0 -> divisor is the bare 1e-7 epsilon
            s.view(-1)[1] = 2.0 ** -5  # exact ties
            o = (torch.randn(C, generator=g) * 0.1).reshape(bshape) if not signed else torch.zeros(bshape)
            flat = x.view(-1)
            flat[::7] = torch.round(flat[::7] * 8) / 8 + 1.0 / 64  # many exact k+0.5 ties on channel 1
            flat[5] = float("nan")
            flat[11] = float("inf")
            flat[17] = -0.0
            name = f"prim_c{idx}"
            put(name, x=x, scale=s, offset=o,
                q=utils.quantize(x, s, o, lo, hi),
                y=utils.emulate_quantize(x, s, o, lo, hi))
            CASES.append(dict(name=name, kind="primitive", layout=layout, ch_axis=ch_axis,
                              signed=signed, n_bits=n_bits, lo=lo, hi=hi))
            idx += 1


# ------------------------------------------------------------------------------ observers
def case_observers():
    """ops.py:20-34 quantize_minmax_tensor, ops.py:112-140 quantize_minmax_channel."""
    idx = 0
    shapes = [((6, 4, 3, 3), 0), ((3, 5, 4, 4), 1), ((2, 3, 7, 7), 1), ((4, 8, 14, 14), 1),
              ((5, 7), 1), ((6, 10), 0), ((1, 3, 9, 5), 1), ((33,), 0), ((2, 16, 28, 28), 1)]
    for shape, ch_axis in shapes:
        for signed, n_bits in ((True, 8), (False, 8), (True, 4), (False, 4)):
            g = gen(1000 + idx)
            x = torch.randn(shape, generator=g)
            if idx % 3 == 1:
                x = torch.relu(x)  # post-ReLU: min == 0 exactly
            if idx % 5 == 4:
                x.view(-1)[3] = -0.0
            name = f"obs{idx}"
            s_t, o_t = ops.get_qparams_tensor(x, "minmax_tensor", n_bits=n_bits, signed=signed)
            arrays = dict(x=x, t_scale=s_t, t_offset=o_t.to(torch.float32))
            desc = dict(name=name, kind="observer", shape=list(shape), ch_axis=ch_axis,
                        signed=signed, n_bits=n_bits, t_offset_dtype=str(o_t.dtype))
            if len(shape) >= 2:
                s_c, o_c = ops.get_qparams_tensor(x, "minmax_channel", n_bits=n_bits,
                                                  signed=signed, ch_axis=ch_axis)
                arrays.update(c_scale=s_c, c_offset=o_c)
            if (not signed) and bool((x.min() >= 0).item()):
                s_n, o_n = ops.get_qparams_tensor(x, "minmax_tensor", n_bits=n_bits, signed=signed,
                                                  allow_offset=False)
                arrays.update(t_scale_nooff=s_n, t_offset_nooff=o_n.to(torch.float32))
                if len(shape) >= 2:
                    s_n, o_n = ops.get_qparams_tensor(x.clone(), "minmax_channel", n_bits=n_bits,
                                                      signed=signed, ch_axis=ch_axis,
                                                      allow_offset=False)
                    arrays.update(c_scale_nooff=s_n, c_offset_nooff=o_n)
            put(name, **arrays)
            CASES.append(desc)
            idx += 1
    # NaN / inf propagation through max/min
    for j, bad in enumerate((float("nan"), float("inf"), float("-inf"))):
        x = torch.randn(4, 6, 5, 5, generator=gen(1500 + j))
        x[1, 2, 3, 4] = bad
        for signed in (True, False):
            name = f"obs_bad{j}_{int(signed)}"
            s_t, o_t = ops.get_qparams_tensor(x, "minmax_tensor", n_bits=8, signed=signed)
            s_c, o_c = ops.get_qparams_tensor(x, "minmax_channel", n_bits=8, signed=signed, ch_axis=1)
            put(name, x=x, t_scale=s_t, t_offset=o_t.to(torch.float32), c_scale=s_c, c_offset=o_c)
            CASES.append(dict(name=name, kind="observer", shape=[4, 6, 5, 5], ch_axis=1,
                              signed=signed, n_bits=8, t_offset_dtype=str(o_t.dtype)))


# ------------------------------------------------------------------------ wrapper helpers
def swap(cls, module, qconfig):
    """dlmc/utils/quantize.py:130-133."""
    q = cls.__new__(cls)
    q.__dict__.update(module.__dict__)
    q.initialize(copy.deepcopy(qconfig))
    return q


class Capture:
    """Record what the wrapper hands to `_forward_func` (the fake-quantised input and weight)."""

    def __init__(self, mod):
        self.mod = mod
        self.orig = mod._forward_func
        mod._forward_func = self

    def __call__(self, input, weight):
        self.input = input.detach().clone()
        self.weight = weight.detach().clone()
        # upstream gradients of the conv / linear w.r.t. its fake-quantised operands: the exact
        # inputs of the fake-quant backward, so that backward can be checked bit for bit
        if input.requires_grad:
            input.register_hook(lambda g: setattr(self, "g_input", g.detach().clone()))
        if weight.requires_grad:
            weight.register_hook(lambda g: setattr(self, "g_weight", g.detach().clone()))
        return self.orig(input, weight)


def make_layer(kind, g):
    if kind == "conv":
        m = torch.nn.Conv2d(4, 6, 3, padding=1, bias=True)
        x = torch.randn(2, 4, 6, 6, generator=g)
    elif kind == "conv_s2":
        m = torch.nn.Conv2d(3, 5, 3, stride=2, padding=1, bias=False)
        x = torch.randn(2, 3, 9, 9, generator=g)
    elif kind == "conv_reflect":
        m = torch.nn.Conv2d(3, 4, 3, padding=1, bias=True, padding_mode="reflect")
        x = torch.randn(2, 3, 6, 6, generator=g)
    elif kind == "conv_group":
        m = torch.nn.Conv2d(4, 8, 3, padding=1, groups=2, bias=True)
        x = torch.randn(2, 4, 5, 5, generator=g)
    else:
        m = torch.nn.Linear(10, 7, bias=True)
        x = torch.randn(5, 10, generator=g)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / m.weight[0].numel()))
        if m.bias is not None:
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return m, x


# ---------------------------------------------------------------------------------- QBase
def case_qbase():
    """modules/base.py:28-140 through QConv2d / QLinear (per-tensor only: defect 3)."""
    idx = 0
    for kind in ("conv", "conv_s2", "conv_reflect", "conv_group", "linear"):
        for (w_signed, w_bits), (i_signed, i_bits) in (((True, 8), (True, 8)), ((True, 8), (False, 8)),
                                                        ((True, 4), (False, 4)), ((False, 4), (True, 8)),
                                                        ((True, 2), (False, 3))):
            g = gen(2000 + idx)
            m, x = make_layer(kind, g)
            if not i_signed and idx % 2 == 0:
                x = torch.relu(x)
            qcfg = {"input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": i_bits, "signed": i_signed}},
                    "weight": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": w_bits, "signed": w_signed}},
                    "momentum": 0.1}
            cls = modules.QLinear if kind == "linear" else modules.QConv2d
            q = swap(cls, m, qcfg)
            cap = Capture(q)
            with torch.no_grad():
                out1 = q(x)            # first call: observer + fake-quant
                x2 = x * 0.7 + 0.05    # second call: frozen scales, different data
                out2 = q(x2)
                in2, wt2 = cap.input, cap.weight
                q(x)
                in1, wt1 = cap.input, cap.weight
            name = f"qbase{idx}"
            put(name, x=x, x2=x2, weight=m.weight, bias=(m.bias if m.bias is not None else torch.zeros(0)),
                in_scale=q.in_scale, in_offset=q.in_offset.to(torch.float32).reshape(-1),
                wt_scale=q.wt_scale, wt_offset=q.wt_offset.to(torch.float32).reshape(-1),
                fq_input=in1, fq_weight=wt1, fq_input2=in2, out=out1, out2=out2)
            CASES.append(dict(name=name, kind="qbase", layer=kind, qconfig=qcfg,
                              state_keys=sorted(q.state_dict().keys())))
            idx += 1
    # autograd through the live path (modules/base.py:96-102,131-133): K7's specification
    for j, ((w_signed, w_bits), (i_signed, i_bits)) in enumerate((((True, 8), (True, 8)), ((True, 4), (False, 4)))):
        g = gen(2500 + j)
        m, x = make_layer("conv", g)
        x = x * 1.5
        qcfg = {"input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": i_bits, "signed": i_signed}},
                "weight": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": w_bits, "signed": w_signed}},
                "momentum": 0.1}
        q = swap(modules.QConv2d, m, qcfg)
        with torch.no_grad():
            q(x)  # calibrate
            # shrink the scales so that part of the data saturates (the clamp mask matters then)
            q.in_scale.mul_(0.6)
            q.wt_scale.mul_(0.7)
        cap = Capture(q)
        xg = x.clone().requires_grad_(True)
        out = q(xg)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        name = f"qbase_grad{j}"
        put(name, g_fq_input=cap.g_input, g_fq_weight=cap.g_weight, x=x, weight=m.weight.detach(), bias=m.bias.detach(), gout=gout,
            in_scale=q.in_scale.detach(), in_offset=q.in_offset.to(torch.float32).reshape(-1),
            wt_scale=q.wt_scale.detach(), wt_offset=q.wt_offset.to(torch.float32).reshape(-1),
            out=out.detach(), grad_x=xg.grad, grad_weight=q.weight.grad, grad_bias=q.bias.grad,
            grad_in_scale=q.in_scale.grad, grad_wt_scale=q.wt_scale.grad)
        CASES.append(dict(name=name, kind="qbase_grad", layer="conv", qconfig=qcfg))
    # output-aware weight scale (ops.py:85-109) through QBase (modules/base.py:108-117), as PTQ_output_config.yaml uses it
    for j, kind in enumerate(("conv", "linear")):
        g = gen(2700 + j)
        m, x = make_layer(kind, g)
        qcfg = {"input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
                "weight": {"enable": True, "type": "l2norm_output", "args": {"n_bits": 4, "signed": True}},
                "momentum": 0.1}
        cls = modules.QLinear if kind == "linear" else modules.QConv2d
        q = swap(cls, m, qcfg)
        cap = Capture(q)
        with torch.no_grad():
            out = q(x)
        name = f"qbase_l2out{j}"
        put(name, x=x, weight=m.weight, bias=m.bias, in_scale=q.in_scale, wt_scale=q.wt_scale,
            wt_offset=q.wt_offset.to(torch.float32).reshape(-1), fq_input=cap.input, fq_weight=cap.weight, out=out)
        CASES.append(dict(name=name, kind="qbase_l2out", layer=kind, qconfig=qcfg))
    # the closed form in FunLSQ.backward (modules/function.py:37-49)
    for j, (lo, hi) in enumerate(((-127, 127), (0, 15))):
        g = gen(2600 + j)
        w = torch.randn(4, 3, 3, 3, generator=g)
        s = torch.tensor([0.02 if lo < 0 else 0.11])
        o = torch.zeros(1)
        gw = torch.randn(w.shape, generator=g)
        gg = 1.0 / math.sqrt(w.numel() * hi)
        wr = w.clone().requires_grad_(True)
        sr = s.clone().requires_grad_(True)
        y = mfunction.FunLSQ.apply(wr, sr, o, lo, hi, gg)
        y.backward(gw)
        name = f"funlsq{j}"
        put(name, w=w, scale=s, gout=gw, y=y.detach(), grad_w=wr.grad, grad_scale=sr.grad)
        CASES.append(dict(name=name, kind="funlsq", lo=lo, hi=hi, g=gg))


# ---------------------------------------------------------------------------------- FSPTQ
def fsptq_init(cls, module, qconfig):
    orig = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return orig(*a, **k)
    q = cls.__new__(cls)
    q.__dict__.update(module.__dict__)
    torch.zeros = zeros_cpu
    try:
        q.initialize(copy.deepcopy(qconfig))
    finally:
        torch.zeros = orig
    return q


def case_fsptq():
    """FSPTQuant/base.py:33-63,95-159 through FSPTQConv2d / FSPTQLinear (per-channel weights)."""
    idx = 0
    for kind in ("conv", "conv_s2", "conv_group", "linear"):
        for (w_signed, w_bits), (i_signed, i_bits), in_type, recon in (
                ((True, 8), (False, 8), "minmax_tensor", "None"),      # FSPTQ_config.yaml:40-53
                ((True, 8), (False, 8), "minmax_tensor", "None_relu"),  # min == 0: integer zero point
                ((True, 4), (True, 8), "minmax_tensor", "None"),
                ((True, 8), (False, 8), "minmax_tensor", "adaround"),
                ((True, 3), (False, 4), "minmax_tensor", "adaround")):
            g = gen(3000 + idx)
            m, x = make_layer(kind, g)
            if recon.endswith("_relu"):
                x = torch.relu(x)
            recon_type = recon.replace("_relu", "")
            qcfg = {"input": {"enable": True, "type": in_type, "args": {"n_bits": i_bits, "signed": i_signed}},
                    "weight": {"enable": True, "type": "minmax_channel", "recon_type": recon_type,
                               "args": {"n_bits": w_bits, "signed": w_signed}},
                    "momentum": 0.1}
            cls = FSPTQuant.FSPTQLinear if kind == "linear" else FSPTQuant.FSPTQConv2d
            q = fsptq_init(cls, m, qcfg)
            cap = Capture(q)
            arrays = {}
            with torch.no_grad():
                q.eval()
                out = q(x)
                arrays.update(fq_input=cap.input, fq_weight=cap.weight, out=out)
                if recon_type == "adaround":
                    arrays.update(alpha_init=q.alpha.detach().clone())
                    q.alpha.add_(torch.randn(q.alpha.shape, generator=g) * 0.5)
                    out_e = q(x)
                    arrays.update(alpha=q.alpha.detach().clone(), fq_weight_eval=cap.weight, out_eval=out_e)
                    q.train()
                    out_t = q(x)
                    arrays.update(fq_weight_train=cap.weight, out_train=out_t,
                                  soft_targets=q.get_soft_targets())
                    q.eval()
                x2 = x * 0.8 + 0.03
                q(x2)
                arrays.update(x2=x2, fq_input2=cap.input)
            name = f"fsptq{idx}"
            put(name, x=x, weight=m.weight, bias=(m.bias if m.bias is not None else torch.zeros(0)),
                in_scale=q.in_scale, in_offset=q.in_offset.to(torch.float32).reshape(-1),
                wt_scale=q.wt_scale, wt_offset=q.wt_offset, **arrays)
            CASES.append(dict(name=name, kind="fsptq", layer=kind, qconfig=qcfg,
                              state_keys=sorted(q.state_dict().keys())))
            idx += 1


def case_fsptq_grad():
    """Autograd through the FSPTQ live path (FSPTQuant/base.py:108-109,149-152: round_pass STE, clamp masks) with scales shrunk
    so that part of the data saturates: the specification of the one-pass HIP backward of the ZEROPOINT / SYMMETRIC forms."""
    for j, ((w_signed, w_bits), (i_signed, i_bits), relu_in) in enumerate((((True, 8), (False, 8), True), ((True, 4), (True, 8), False),
                                                                            ((True, 8), (False, 8), False))):
        g = gen(3500 + j)
        m, x = make_layer("conv", g)
        x = x * 1.5
        if relu_in:
            x = torch.relu(x)          # min == 0: integer zero point (the int8 path's case); otherwise the float minimum
        qcfg = {"input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": i_bits, "signed": i_signed}},
                "weight": {"enable": True, "type": "minmax_channel", "recon_type": "None", "args": {"n_bits": w_bits, "signed": w_signed}},
                "momentum": 0.1}
        q = fsptq_init(FSPTQuant.FSPTQConv2d, m, qcfg)
        with torch.no_grad():
            q.eval()
            q(x)  # calibrate
            q.in_scale.mul_(0.6)
            q.wt_scale.mul_(0.7)
        cap = Capture(q)
        xg = x.clone().requires_grad_(True)
        out = q(xg)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        name = f"fsptq_grad{j}"
        extra = {}
        if isinstance(q.in_scale, torch.nn.Parameter) and q.in_scale.grad is not None:
            extra["grad_in_scale"] = q.in_scale.grad
        if isinstance(q.wt_scale, torch.nn.Parameter) and q.wt_scale.grad is not None:
            extra["grad_wt_scale"] = q.wt_scale.grad
        put(name, g_fq_input=cap.g_input, g_fq_weight=cap.g_weight, fq_input=cap.input, fq_weight=cap.weight, x=x,
            weight=m.weight.detach(), bias=m.bias.detach(), gout=gout, in_scale=q.in_scale.detach(),
            in_offset=q.in_offset.to(torch.float32).reshape(-1), wt_scale=q.wt_scale.detach(), out=out.detach(), grad_x=xg.grad,
            grad_weight=q.weight.grad, grad_bias=q.bias.grad, **extra)
        CASES.append(dict(name=name, kind="fsptq_grad", layer="conv", qconfig=qcfg))


# ---------------------------------------------------------------------------------- RootQ
def case_rootq():
    """RootQ/base.py:37-156 + RootQ/function.py:15-32,58-67."""
    idx = 0
    for kind in ("conv", "linear", "conv_group"):
        for (w_signed, w_bits), (i_signed, i_bits) in (((False, 4), (False, 4)), ((False, 2), (False, 2)),
                                                        ((True, 4), (False, 8)), ((False, 8), (False, 8)),
                                                        ((False, 3), (True, 4))):
            g = gen(4000 + idx)
            m, x = make_layer(kind, g)
            if idx % 2 == 0:
                x = torch.relu(x)
            qcfg = {"input": {"enable": True, "type": None, "args": {"n_bits": i_bits, "signed": i_signed}},
                    "weight": {"enable": True, "type": None, "args": {"n_bits": w_bits, "signed": w_signed}},
                    "momentum": 0.1}
            cls = RootQ.RootQLinear if kind == "linear" else RootQ.RootQConv2d
            q = swap(cls, m, qcfg)
            cap = Capture(q)
            arrays = {}
            import contextlib, io
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                q.eval()
                out = q(x)  # init + eval forward
                arrays.update(out=out, fq_input=cap.input, fq_weight=cap.weight,
                              st_in_scale=q.in_scale.detach().clone(), st_in_run_scale=q.in_run_scale.clone(),
                              st_wt_upper=q.wt_upper.detach().clone(), st_wt_lower=q.wt_lower.detach().clone(),
                              st_wt_run_upper=q.wt_run_upper.clone(), st_wt_run_lower=q.wt_run_lower.clone())
                # move the learnable bounds, then one train-mode step (EMA update) and an eval step
                q.in_scale.mul_(0.8)
                q.wt_upper.mul_(0.9)
                q.wt_lower.mul_(0.85)
                q.train()
                x2 = x * 1.3 - 0.02
                out_t = q(x2)
                arrays.update(x2=x2, out_train=out_t, fq_input_train=cap.input, fq_weight_train=cap.weight,
                              tr_in_scale=q.in_scale.detach().clone(), tr_in_run_scale=q.in_run_scale.clone(),
                              tr_wt_upper=q.wt_upper.detach().clone(), tr_wt_lower=q.wt_lower.detach().clone(),
                              tr_wt_run_upper=q.wt_run_upper.clone(), tr_wt_run_lower=q.wt_run_lower.clone())
                q.eval()
                out_e = q(x2)
                arrays.update(out_eval2=out_e, fq_input_eval2=cap.input, fq_weight_eval2=cap.weight)
            name = f"rootq{idx}"
            put(name, x=x, weight=m.weight, bias=(m.bias if m.bias is not None else torch.zeros(0)),
                wt_alpha=q.wt_alpha.detach(), **arrays)
            CASES.append(dict(name=name, kind="rootq", layer=kind, qconfig=qcfg,
                              state_keys=sorted(q.state_dict().keys())))
            idx += 1
    # gradients of the RootQ live path (train mode)
    for j, ((w_signed, w_bits), (i_signed, i_bits)) in enumerate((((False, 4), (False, 4)), ((False, 2), (False, 8)))):
        g = gen(4500 + j)
        m, x = make_layer("conv", g)
        x = torch.relu(x) * 1.2
        qcfg = {"input": {"enable": True, "type": None, "args": {"n_bits": i_bits, "signed": i_signed}},
                "weight": {"enable": True, "type": None, "args": {"n_bits": w_bits, "signed": w_signed}},
                "momentum": 0.1}
        q = swap(RootQ.RootQConv2d, m, qcfg)
        import contextlib, io
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            q.eval()
            q(x)
            q.in_scale.mul_(0.7)
            q.wt_upper.mul_(0.8)
            q.wt_lower.mul_(0.8)
        q.train()
        pre = {k: v.detach().clone() for k, v in q.state_dict().items()}
        cap = Capture(q)
        xg = x.clone().requires_grad_(True)
        out = q(xg)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        name = f"rootq_grad{j}"
        put(name, g_fq_input=cap.g_input, g_fq_weight=cap.g_weight, fq_input=cap.input, fq_weight=cap.weight,
            x=x, weight=m.weight.detach(), bias=m.bias.detach(), gout=gout, out=out.detach(),
            grad_x=xg.grad, grad_weight=q.weight.grad, grad_in_scale=q.in_scale.grad,
            grad_wt_upper=q.wt_upper.grad, grad_wt_lower=q.wt_lower.grad, grad_wt_alpha=q.wt_alpha.grad,
            **{f"pre_{k}": v for k, v in pre.items() if k not in ("weight", "bias")})
        CASES.append(dict(name=name, kind="rootq_grad", layer="conv", qconfig=qcfg))


# ----------------------------------------------------------------------------- estimators
def case_estimators():
    """ops.py:36-83,169-215: iterative scale refinement (calibration only).  The sums are fp32
    reductions whose order is an ATen implementation detail, so these are tolerance fixtures."""
    idx = 0
    import contextlib, io
    for shape, ch_axis in (((6, 4, 3, 3), 0), ((5, 12), 0)):
        for signed, n_bits in ((True, 8), (True, 4), (False, 4)):
            g = gen(5000 + idx)
            x = torch.randn(shape, generator=g)
            if not signed:
                x = torch.relu(x) + 0.01 * torch.rand(shape, generator=g)
            name = f"est{idx}"
            arrays = dict(x=x)
            with contextlib.redirect_stdout(io.StringIO()):
                s, o = ops.get_qparams_tensor(x, "l2norm_tensor", n_bits=n_bits, signed=signed)
                arrays.update(l2norm_t_scale=s, l2norm_t_offset=o.to(torch.float32))
                s, o = ops.get_qparams_tensor(x, "l2norm_channel", n_bits=n_bits, signed=signed, ch_axis=ch_axis)
                arrays.update(l2norm_c_scale=s, l2norm_c_offset=o)
                s, o = ops.get_qparams_tensor(x, "l2loss_tensor", n_bits=n_bits, signed=signed)
                arrays.update(l2loss_t_scale=s, l2loss_t_offset=o.to(torch.float32))
                if not signed:
                    s, o = ops.get_qparams_tensor(x, "l2loss_channel", n_bits=n_bits, signed=signed, ch_axis=ch_axis)
                    arrays.update(l2loss_c_scale=s, l2loss_c_offset=o)
            put(name, **arrays)
            CASES.append(dict(name=name, kind="estimator", shape=list(shape), ch_axis=ch_axis,
                              signed=signed, n_bits=n_bits))
            idx += 1


# ------------------------------------------------------- weight transforms before the path
def case_weight_transforms():
    """dlmc/utils/merge_bn.py:45-113 and model/classification/repvgg.py:92-147 (switch_to_deploy)."""
    # merge_bn imports the BitMixer package, which is missing from the reference: give it a shell
    bm = types.ModuleType("dlmc.quantization.scalar.BitMixer")
    bm.BitMixerBatchNorm = bm.BitMixerSwitchableBatchNorm = type("Missing", (), {})
    sys.modules["dlmc.quantization.scalar.BitMixer"] = bm
    spec = importlib.util.spec_from_file_location("ref_access", os.path.join(REF, "dlmc/utils/access.py"))
    # dlmc/utils/__init__ pulls quantize.py (timm etc.): load merge_bn.py by path inside a package shell
    pkg = types.ModuleType("dlmc.utils")
    pkg.__path__ = [os.path.join(REF, "dlmc/utils")]
    sys.modules["dlmc.utils"] = pkg
    for sub in ("access", "merge_bn"):
        sp = importlib.util.spec_from_file_location(f"dlmc.utils.{sub}", os.path.join(REF, f"dlmc/utils/{sub}.py"))
        m = importlib.util.module_from_spec(sp)
        sys.modules[f"dlmc.utils.{sub}"] = m
        sp.loader.exec_module(m)
    ref_merge = sys.modules["dlmc.utils.merge_bn"]
    sp = importlib.util.spec_from_file_location("ref_repvgg", os.path.join(REF, "model/classification/repvgg.py"))
    ref_repvgg = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(ref_repvgg)
    import contextlib, io

    def randomise_bn(bn, g):
        with torch.no_grad():
            bn.weight.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.2)
            bn.running_var.copy_(torch.rand(bn.num_features, generator=g) + 0.3)

    for j, (bias, groups) in enumerate(((False, 1), (True, 1), (True, 2))):
        g = gen(6000 + j)
        net = torch.nn.Sequential()
        net.add_module("conv1", torch.nn.Conv2d(4, 6, 3, padding=1, bias=bias, groups=groups))
        net.add_module("bn1", torch.nn.BatchNorm2d(6))
        with torch.no_grad():
            net.conv1.weight.copy_(torch.randn(net.conv1.weight.shape, generator=g) * 0.3)
            if bias:
                net.conv1.bias.copy_(torch.randn(6, generator=g) * 0.1)
        randomise_bn(net.bn1, g)
        pre = {k: v.clone() for k, v in net.state_dict().items()}
        merged = ref_merge.merge_bn(net, inplace=True)
        name = f"mergebn{j}"
        put(name, weight=pre["conv1.weight"], bias=pre.get("conv1.bias", torch.zeros(0)), gamma=pre["bn1.weight"],
            beta=pre["bn1.bias"], mean=pre["bn1.running_mean"], var=pre["bn1.running_var"],
            out_weight=merged.conv1.weight, out_bias=merged.conv1.bias)
        CASES.append(dict(name=name, kind="merge_bn", groups=groups, has_bias=bias,
                          bn_replaced_by=type(merged.bn1).__name__))
    for j, (cin, cout, stride, groups) in enumerate(((8, 8, 1, 1), (8, 12, 2, 1), (8, 8, 1, 2), (6, 6, 2, 1))):
        g = gen(6100 + j)
        with contextlib.redirect_stdout(io.StringIO()):
            blk = ref_repvgg.RepVGGBlock(cin, cout, 3, stride=stride, padding=1, groups=groups)
        with torch.no_grad():
            blk.rbr_dense.conv.weight.copy_(torch.randn(blk.rbr_dense.conv.weight.shape, generator=g) * 0.2)
            blk.rbr_1x1.conv.weight.copy_(torch.randn(blk.rbr_1x1.conv.weight.shape, generator=g) * 0.2)
        for bn in (blk.rbr_dense.bn, blk.rbr_1x1.bn, blk.rbr_identity):
            if bn is not None:
                randomise_bn(bn, g)
        pre = {k: v.clone() for k, v in blk.state_dict().items()}
        blk.eval()
        blk.switch_to_deploy()
        name = f"repvgg{j}"
        put(name, out_kernel=blk.rbr_reparam.weight, out_bias=blk.rbr_reparam.bias,
            **{("pre_" + k.replace(".", "__")): v for k, v in pre.items() if "num_batches" not in k})
        CASES.append(dict(name=name, kind="repvgg", cin=cin, cout=cout, stride=stride, groups=groups,
                          has_identity=any(k.startswith("rbr_identity") for k in pre)))


# ------------------------------------------------- round-4 supplement: LSQ initialisation, minmax_pixel
def case_lsq():
    """`type: "LSQ"` through QConv2d / QLinear (modules/base.py:84-85 input, :118-121 weight): the QAT flow's default first-call
    initialiser 2 * mean|x| / sqrt(Qp) (example/quantization/LSQ_config.yaml: W s3, A u3), then the QBase forward with it.  The
    reference's LSQ branch allocates its zero offsets on device('cuda'): torch.zeros is wrapped for the duration of the forward
    (the same patch fsptq_init uses for FSPTQBase.initialize)."""
    orig = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return orig(*a, **k)
    idx = 0
    for kind in ("conv", "conv_s2", "linear"):
        for (w_signed, w_bits), (i_signed, i_bits), w_type, i_type in (
                ((True, 3), (False, 3), "LSQ", "LSQ"),              # LSQ_config.yaml
                ((True, 8), (True, 8), "LSQ", "LSQ"),
                ((True, 4), (False, 8), "LSQ", "minmax_tensor"),    # only the weights by LSQ
                ((True, 8), (False, 4), "minmax_tensor", "LSQ")):   # only the input by LSQ
            g = gen(7000 + idx)
            m, x = make_layer(kind, g)
            if not i_signed:
                x = torch.relu(x)
            qcfg = {"input": {"enable": True, "type": i_type, "args": {"n_bits": i_bits, "signed": i_signed}},
                    "weight": {"enable": True, "type": w_type, "args": {"n_bits": w_bits, "signed": w_signed}},
                    "momentum": 0.1}
            cls = modules.QLinear if kind == "linear" else modules.QConv2d
            q = swap(cls, m, qcfg)
            cap = Capture(q)
            torch.zeros = zeros_cpu
            try:
                with torch.no_grad():
                    out1 = q(x)
                    in1, wt1 = cap.input, cap.weight
                    x2 = x * 0.7 + 0.05
                    out2 = q(x2)
                    in2 = cap.input
            finally:
                torch.zeros = orig
            name = f"lsq{idx}"
            put(name, x=x, x2=x2, weight=m.weight, bias=(m.bias if m.bias is not None else torch.zeros(0)),
                in_scale=q.in_scale, in_offset=q.in_offset.to(torch.float32).reshape(-1),
                wt_scale=q.wt_scale, wt_offset=q.wt_offset.to(torch.float32).reshape(-1),
                fq_input=in1, fq_weight=wt1, fq_input2=in2, out=out1, out2=out2)
            CASES.append(dict(name=name, kind="lsq", layer=kind, qconfig=qcfg, state_keys=sorted(q.state_dict().keys())))
            idx += 1


def case_minmax_pixel():
    """ops.py:142-167 quantize_minmax_pixel: one (scale, offset) per kernel position, reduced over out- and in-channels -
    including the reference's quirk that the UNSIGNED branch takes the minimum of |x| (ops.py:156), and a 3-D tensor."""
    idx = 0
    for shape in ((6, 4, 3, 3), (5, 3, 1, 1), (4, 6, 5), (8, 2, 7, 7)):
        for signed, n_bits in ((True, 8), (False, 8), (True, 4), (False, 4)):
            g = gen(7500 + idx)
            x = torch.randn(shape, generator=g)
            if idx % 3 == 2:
                x = torch.relu(x) + 0.05
            arrays = dict(x=x)
            s, o = ops.get_qparams_tensor(x, "minmax_pixel", n_bits=n_bits, signed=signed)
            arrays.update(scale=s, offset=o)
            if not signed:
                s, o = ops.get_qparams_tensor(x.clone(), "minmax_pixel", n_bits=n_bits, signed=signed, allow_offset=False)
                arrays.update(scale_nooff=s, offset_nooff=o)
            name = f"pixel{idx}"
            put(name, **arrays)
            CASES.append(dict(name=name, kind="minmax_pixel", shape=list(shape), signed=signed, n_bits=n_bits))
            idx += 1


def main_supplement2():
    """Round 4: golden_v2.{npz,json} (golden_v1 and golden_v1_grad are not regenerated)."""
    torch.manual_seed(SEED)
    torch.set_num_threads(1)
    case_lsq()
    case_minmax_pixel()
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **ARR)
    with open(os.path.join(HERE, "golden_v2.json"), "w") as f:
        json.dump(dict(seed=SEED, torch=torch.__version__, cases=CASES), f, indent=1, default=str)
    print(f"{len(CASES)} cases, {len(ARR)} arrays (supplement 2)")


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(1)
    case_primitives()
    case_observers()
    case_qbase()
    case_fsptq()
    case_rootq()
    case_estimators()
    case_weight_transforms()
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **ARR)
    with open(os.path.join(HERE, "golden_v1.json"), "w") as f:
        json.dump(dict(seed=SEED, torch=torch.__version__, cases=CASES), f, indent=1, default=str)
    print(f"{len(CASES)} cases, {len(ARR)} arrays, "
          f"{os.path.getsize(os.path.join(HERE, 'golden_v1.npz')) / 1024:.0f} KiB")


def main_supplement():
    """Cases added after golden_v1 was frozen go to their own pair of files (golden_v1 itself is not regenerated)."""
    torch.manual_seed(SEED)
    torch.set_num_threads(1)
    case_fsptq_grad()
    np.savez_compressed(os.path.join(HERE, "golden_v1_grad.npz"), **ARR)
    with open(os.path.join(HERE, "golden_v1_grad.json"), "w") as f:
        json.dump(dict(seed=SEED, torch=torch.__version__, cases=CASES), f, indent=1, default=str)
    print(f"{len(CASES)} cases, {len(ARR)} arrays (supplement)")


if __name__ == "__main__":
    if "--supplement2" in sys.argv:
        main_supplement2()
    elif "--supplement" in sys.argv:
        main_supplement()
    else:
        main()
