"""The quant wrappers on the GPU, driven exactly the way the reference's trainers drive them
(`quantize_model` then `model(data)`), against the golden vectors recorded from the reference's own
wrappers: calibrated scales, the fake-quantised operands handed to conv / linear (bit for bit), layer
outputs and gradients (fp32 tolerance: the conv itself is MIOpen), state_dict layout and reload."""
import copy
import io
import math

import pytest
import torch
from torch import nn

from _cmp import assert_bits_equal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(case, golden):
    kind = case["layer"]
    w, b = golden.get(case, "weight"), golden.get(case, "bias")
    if kind == "linear":
        m = nn.Linear(w.shape[1], w.shape[0], bias=b.numel() > 0)
    else:
        kw = dict(conv=dict(padding=1), conv_s2=dict(stride=2, padding=1),
                  conv_reflect=dict(padding=1, padding_mode="reflect"), conv_group=dict(padding=1, groups=2))[kind]
        m = nn.Conv2d(w.shape[1] * kw.get("groups", 1), w.shape[0], w.shape[2], bias=b.numel() > 0, **kw)
    with torch.no_grad():
        m.weight.copy_(w)
        if b.numel():
            m.bias.copy_(b)
    return m


class Holder(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        return self.layer(x)


class Capture:
    def __init__(self, mod):
        self.orig = mod._forward_func
        mod._forward_func = self

    def __call__(self, input, weight):
        self.input, self.weight = input.detach().clone(), weight.detach().clone()
        return self.orig(input, weight)


def _quantized(case, golden, qtype=None):
    from dlmc.utils.quantize import quantize_model
    cfg = {"weight": copy.deepcopy(case["qconfig"]["weight"]), "input": copy.deepcopy(case["qconfig"]["input"]),
           "momentum": case["qconfig"].get("momentum", 0.1), "exclude_layers": [], "override_options": []}
    net = Holder(_layer(case, golden)).to(DEV)
    quantize_model(net, cfg, None, quantization_type=qtype)
    return net, Capture(net.layer)


def close(got, want, what="", rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(got.detach().cpu(), want, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def same_values(got, want, what=""):
    g, w = got.detach().cpu().reshape(-1), want.reshape(-1)
    assert g.shape == w.shape and bool(((g == w) | (g.isnan() & w.isnan())).all()), f"{what}: {g} vs {w}"


def test_qbase_wrappers(golden):
    from dlmc.quantization.scalar import modules
    for c in golden.of_kind("qbase"):
        net, cap = _quantized(c, golden)
        assert isinstance(net.layer, modules.QBase)
        x, x2 = golden.get(c, "x").to(DEV), golden.get(c, "x2").to(DEV)
        with torch.no_grad():
            out = net(x)
            same_values(net.layer.in_scale, golden.get(c, "in_scale"), c["name"] + ".in_scale")
            same_values(net.layer.wt_scale, golden.get(c, "wt_scale"), c["name"] + ".wt_scale")
            same_values(net.layer.in_offset, golden.get(c, "in_offset"))
            same_values(net.layer.wt_offset, golden.get(c, "wt_offset"))
            assert_bits_equal(cap.input, golden.get(c, "fq_input"), c["name"] + ".fq_input")
            assert_bits_equal(cap.weight, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
            close(out, golden.get(c, "out"), c["name"] + ".out")
            out2 = net(x2)                                    # frozen scales
            assert_bits_equal(cap.input, golden.get(c, "fq_input2"), c["name"] + ".fq_input2")
            close(out2, golden.get(c, "out2"), c["name"] + ".out2")
        assert sorted(net.layer.state_dict().keys()) == c["state_keys"]
        assert float(net.layer.in_init_state) == 1 and float(net.layer.wt_init_state) == 1


def test_qbase_autograd(golden):
    for c in golden.of_kind("qbase_grad"):
        net, cap = _quantized(c, golden)
        q = net.layer
        x = golden.get(c, "x").to(DEV)
        with torch.no_grad():
            net(x)
            q.in_scale.copy_(golden.get(c, "in_scale"))     # the shrunk scales of the recorded run
            q.wt_scale.copy_(golden.get(c, "wt_scale"))
        xg = x.clone().requires_grad_(True)
        out = net(xg)
        close(out, golden.get(c, "out"), c["name"] + ".out")
        out.backward(golden.get(c, "gout").to(DEV))
        close(xg.grad, golden.get(c, "grad_x"), c["name"] + ".grad_x", rtol=1e-3, atol=1e-5)
        close(q.weight.grad, golden.get(c, "grad_weight"), c["name"] + ".grad_weight", rtol=1e-3, atol=1e-4)
        close(q.bias.grad, golden.get(c, "grad_bias"), rtol=1e-3, atol=1e-4)
        close(q.in_scale.grad, golden.get(c, "grad_in_scale"), c["name"] + ".grad_in_scale", rtol=2e-3, atol=1e-5)
        close(q.wt_scale.grad, golden.get(c, "grad_wt_scale"), c["name"] + ".grad_wt_scale", rtol=2e-3, atol=1e-5)


def test_fsptq_autograd(golden):
    """The FSPTQ wrapper under autograd on the GPU (HIP forward, one-pass HIP backward of both operands) against the gradients
    the REFERENCE produced on the same inputs with the same (shrunk, partly saturating) scales."""
    cases = golden.of_kind("fsptq_grad")
    assert len(cases) == 3
    for c in cases:
        net, cap = _quantized(c, golden, "FSPTQ")
        q = net.layer
        x = golden.get(c, "x").to(DEV)
        with torch.no_grad():
            net.eval()
            net(x)
            q.in_scale.copy_(golden.get(c, "in_scale").reshape(q.in_scale.shape))
            q.wt_scale.copy_(golden.get(c, "wt_scale").reshape(q.wt_scale.shape))
        xg = x.clone().requires_grad_(True)
        out = net(xg)
        assert_bits_equal(cap.input, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        assert_bits_equal(cap.weight, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        close(out, golden.get(c, "out"), c["name"] + ".out")
        out.backward(golden.get(c, "gout").to(DEV))
        close(xg.grad, golden.get(c, "grad_x"), c["name"] + ".grad_x", rtol=1e-3, atol=1e-5)
        close(q.weight.grad, golden.get(c, "grad_weight"), c["name"] + ".grad_weight", rtol=1e-3, atol=1e-4)
        close(q.bias.grad, golden.get(c, "grad_bias"), rtol=1e-3, atol=1e-4)
        close(q.in_scale.grad.reshape(-1), golden.get(c, "grad_in_scale").reshape(-1), c["name"] + ".grad_in_scale", rtol=2e-3, atol=1e-5)
        close(q.wt_scale.grad.reshape(-1), golden.get(c, "grad_wt_scale").reshape(-1), c["name"] + ".grad_wt_scale", rtol=2e-3, atol=1e-4)


def test_fsptq_wrappers(golden):
    from dlmc.quantization.scalar import FSPTQuant
    for c in golden.of_kind("fsptq"):
        net, cap = _quantized(c, golden, "FSPTQ")
        q = net.layer
        assert isinstance(q, FSPTQuant.FSPTQBase)
        x = golden.get(c, "x").to(DEV)
        ada = c["qconfig"]["weight"]["recon_type"] == "adaround"
        with torch.no_grad():
            net.eval()
            out = net(x)
            same_values(q.in_scale, golden.get(c, "in_scale"))
            same_values(q.in_offset, golden.get(c, "in_offset"))
            same_values(q.wt_scale, golden.get(c, "wt_scale"), c["name"] + ".wt_scale")
            assert list(q.wt_scale.shape) == list(golden.get(c, "wt_scale").shape)
            same_values(q.wt_offset, golden.get(c, "wt_offset"))
            assert_bits_equal(cap.input, golden.get(c, "fq_input"), c["name"] + ".fq_input")
            if ada:
                close(q.alpha, golden.get(c, "alpha_init"), c["name"] + ".alpha_init", rtol=1e-4, atol=1e-4)
                q.alpha.copy_(golden.get(c, "alpha"))
                net(x)
                assert_bits_equal(cap.weight, golden.get(c, "fq_weight_eval"), c["name"] + ".fq_weight_eval")
                net.train()
                out_t = net(x)
                close(cap.weight, golden.get(c, "fq_weight_train"), c["name"] + ".fq_weight_train", rtol=1e-5, atol=1e-6)
                close(out_t, golden.get(c, "out_train"), rtol=1e-3, atol=1e-4)
                net.eval()
            else:
                assert_bits_equal(cap.weight, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
                close(out, golden.get(c, "out"), c["name"] + ".out")
            net(golden.get(c, "x2").to(DEV))
            assert_bits_equal(cap.input, golden.get(c, "fq_input2"), c["name"] + ".fq_input2")
        assert sorted(q.state_dict().keys()) == c["state_keys"]
    # FSPTQTrainer.change_model_state switches the first layer's activation quantisation off
    # (trainer/fsptq_trainer.py:158-159); the reference crashes there, this build uses the raw input.
    c = [k for k in golden.of_kind("fsptq") if k["qconfig"]["weight"]["recon_type"] != "adaround"][0]
    net, cap = _quantized(c, golden, "FSPTQ")
    net.layer.change_quant_state(True, False)
    x = golden.get(c, "x").to(DEV)
    with torch.no_grad():
        net(x)
    assert torch.equal(cap.input, x)
    assert_bits_equal(cap.weight, golden.get(c, "fq_weight"))


def test_fsptq_training_step_moves_alpha_and_scale(golden):
    """One optimiser step of block reconstruction: gradients reach alpha and in_scale."""
    c = [k for k in golden.of_kind("fsptq") if k["qconfig"]["weight"]["recon_type"] == "adaround"][0]
    net, _ = _quantized(c, golden, "FSPTQ")
    q = net.layer
    x = golden.get(c, "x").to(DEV)
    net.train()
    out = net(x)
    target = torch.randn_like(out)
    ((out - target) ** 2).mean().backward()
    assert q.alpha.grad is not None and float(q.alpha.grad.abs().sum()) > 0
    assert q.in_scale.grad is not None and torch.isfinite(q.in_scale.grad).all()


def test_rootq_wrappers(golden):
    from dlmc.quantization.scalar import RootQ
    for c in golden.of_kind("rootq"):
        net, cap = _quantized(c, golden, "RootQ")
        q = net.layer
        assert isinstance(q, RootQ.RootQBase)
        x, x2 = golden.get(c, "x").to(DEV), golden.get(c, "x2").to(DEV)
        with torch.no_grad():
            net.eval()
            out = net(x)
            same_values(q.in_scale, golden.get(c, "st_in_scale"), c["name"] + ".in_scale")
            same_values(q.in_run_scale, golden.get(c, "st_in_run_scale"))
            close(q.wt_upper, golden.get(c, "st_wt_upper"), rtol=1e-5, atol=0)     # mean|W|: reduction order
            close(q.wt_lower, golden.get(c, "st_wt_lower"), rtol=1e-5, atol=0)
            assert_bits_equal(cap.input, golden.get(c, "fq_input"), c["name"] + ".fq_input")
            # pin the bounds to the recorded ones so the weight grid is comparable bit for bit
            for name in ("wt_upper", "wt_lower", "wt_run_upper", "wt_run_lower"):
                getattr(q, name).copy_(golden.get(c, "st_" + name))
            out = net(x)
            assert_bits_equal(cap.weight, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
            close(out, golden.get(c, "out"), c["name"] + ".out")
            # one train-mode step after moving the learnable bounds, as in the recorded run
            q.in_scale.mul_(0.8)
            q.wt_upper.mul_(0.9)
            q.wt_lower.mul_(0.85)
            net.train()
            out_t = net(x2)
            for name in ("in_run_scale", "wt_run_upper", "wt_run_lower"):
                same_values(getattr(q, name), golden.get(c, "tr_" + name), c["name"] + ".tr_" + name)
            assert_bits_equal(cap.input, golden.get(c, "fq_input_train"), c["name"] + ".fq_input_train")
            assert_bits_equal(cap.weight, golden.get(c, "fq_weight_train"), c["name"] + ".fq_weight_train")
            close(out_t, golden.get(c, "out_train"))
            net.eval()
            net(x2)
            assert_bits_equal(cap.input, golden.get(c, "fq_input_eval2"))
            assert_bits_equal(cap.weight, golden.get(c, "fq_weight_eval2"))
        assert sorted(q.state_dict().keys()) == c["state_keys"]


def test_rootq_autograd(golden):
    for c in golden.of_kind("rootq_grad"):
        net, cap = _quantized(c, golden, "RootQ")
        q = net.layer
        x = golden.get(c, "x").to(DEV)
        with torch.no_grad():
            net.eval()
            net(x)
            for k in ("in_scale", "in_run_scale", "wt_upper", "wt_lower", "wt_alpha", "wt_run_upper", "wt_run_lower"):
                getattr(q, k).copy_(golden.get(c, "pre_" + k))
        net.train()
        xg = x.clone().requires_grad_(True)
        out = net(xg)
        assert_bits_equal(cap.input, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        assert_bits_equal(cap.weight, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        close(out, golden.get(c, "out"))
        out.backward(golden.get(c, "gout").to(DEV))
        close(xg.grad, golden.get(c, "grad_x"), c["name"] + ".grad_x", rtol=1e-3, atol=1e-5)
        close(q.weight.grad, golden.get(c, "grad_weight"), c["name"] + ".grad_weight", rtol=2e-3, atol=1e-4)
        close(q.in_scale.grad, golden.get(c, "grad_in_scale"), c["name"] + ".grad_in_scale", rtol=2e-3, atol=1e-5)
        close(q.wt_upper.grad, golden.get(c, "grad_wt_upper"), c["name"] + ".grad_wt_upper", rtol=5e-3, atol=1e-5)
        close(q.wt_lower.grad, golden.get(c, "grad_wt_lower"), c["name"] + ".grad_wt_lower", rtol=5e-3, atol=1e-5)
        close(q.wt_alpha.grad, golden.get(c, "grad_wt_alpha"), c["name"] + ".grad_wt_alpha", rtol=5e-3, atol=1e-5)


def test_checkpoint_roundtrip_and_reset(golden):
    """state_dict -> fresh quantised model: identical outputs with no re-calibration; reset re-arms."""
    c = golden.of_kind("qbase")[0]
    net, cap = _quantized(c, golden)
    x, x2 = golden.get(c, "x").to(DEV), golden.get(c, "x2").to(DEV)
    with torch.no_grad():
        net(x)
        ref2 = net(x2)
    buf = io.BytesIO()
    torch.save(net.state_dict(), buf)
    buf.seek(0)
    net_b, cap_b = _quantized(c, golden)
    net_b.load_state_dict(torch.load(buf))
    with torch.no_grad():
        out2 = net_b(x2)                                     # must NOT observe x2: scales come from the file
    assert torch.equal(out2, ref2)
    same_values(net_b.layer.in_scale, golden.get(c, "in_scale"))
    net_b.layer.reset_qparams()
    with torch.no_grad():
        net_b(x2)
    assert float(net_b.layer.in_scale) != float(golden.get(c, "in_scale"))  # re-observed on x2


def test_per_channel_qbase_extension():
    """`minmax_channel` through QBase raises in the reference (defect 3); here it works and equals the
    per-channel EMULATE-free QBASE form of the oracle."""
    from dlmc.utils.quantize import quantize_model
    from oracle import fakequant_oracle as O
    torch.manual_seed(2333)
    conv = nn.Conv2d(8, 16, 3, padding=1)
    net = Holder(conv).to(DEV)
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    quantize_model(net, cfg, None)
    cap = Capture(net.layer)
    x = torch.rand(4, 8, 12, 12)
    with torch.no_grad():
        net(x.to(DEV))
    s_in, o_in = O.minmax_channel(x, 8, False, ch_axis=1)
    s_wt, o_wt = O.minmax_channel(conv.weight.detach().cpu(), 8, True, ch_axis=0)
    assert net.layer.in_scale.shape == (1, 8, 1, 1) and net.layer.wt_scale.shape == (16, 1, 1, 1)
    want_x = O.fq_qbase(x, s_in, o_in, 0, 255, 1 / math.sqrt(x.numel() * 255))[1]
    want_w = O.fq_qbase(conv.weight.detach().cpu(), s_wt, o_wt, -127, 127, 1 / math.sqrt(conv.weight.numel() * 127))[1]
    assert_bits_equal(cap.input, want_x, "per-channel input")
    assert_bits_equal(cap.weight, want_w, "per-channel weight")


def test_estimators_on_device(golden):
    from dlmc.quantization.scalar import ops
    for c in golden.of_kind("estimator"):
        x = golden.get(c, "x").to(DEV)
        s, o = ops.get_qparams_tensor(x, "l2norm_tensor", n_bits=c["n_bits"], signed=c["signed"])
        close(s, golden.get(c, "l2norm_t_scale"), c["name"] + ".l2norm_t", rtol=1e-4, atol=0)
        s, o = ops.get_qparams_tensor(x, "l2norm_channel", n_bits=c["n_bits"], signed=c["signed"], ch_axis=c["ch_axis"])
        close(s, golden.get(c, "l2norm_c_scale"), c["name"] + ".l2norm_c", rtol=1e-4, atol=0)
        s, o = ops.get_qparams_tensor(x, "l2loss_tensor", n_bits=c["n_bits"], signed=c["signed"])
        close(s, golden.get(c, "l2loss_t_scale"), c["name"] + ".l2loss_t", rtol=1e-5, atol=0)
        close(o, golden.get(c, "l2loss_t_offset"), rtol=0, atol=0)
        if golden.has(c, "l2loss_c_scale"):
            s, o = ops.get_qparams_tensor(x, "l2loss_channel", n_bits=c["n_bits"], signed=c["signed"], ch_axis=c["ch_axis"])
            close(s, golden.get(c, "l2loss_c_scale"), c["name"] + ".l2loss_c", rtol=1e-5, atol=0)
            close(o, golden.get(c, "l2loss_c_offset"), rtol=0, atol=0)


def test_resnet18_config1_end_to_end():
    """BASELINE config 1: ResNet-18 W8A8 per-tensor symmetric, batch 1 - every layer's fake-quantised
    operands equal the oracle's when it is fed the same layer inputs."""
    import workloads as W
    from dlmc.quantization.scalar import modules
    from dlmc.utils.quantize import quantize_model
    from oracle import fakequant_oracle as O
    torch.manual_seed(2333)
    net = W.resnet18().to(DEV).eval()
    cfg = {"weight": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
           "exclude_layers": [], "override_options": []}
    quantize_model(net, cfg, None)
    layers = [(n, m) for n, m in net.named_modules() if isinstance(m, modules.QBase)]
    assert len(layers) == 21
    seen = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp, n=n: seen.__setitem__(n, inp[0].detach().cpu())) for n, m in layers]
    caps = {n: Capture(m) for n, m in layers}
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(2333))
    with torch.no_grad():
        out = net(x.to(DEV))
    assert out.shape == (1, 1000) and torch.isfinite(out).all()
    for n, m in layers:
        xin, w = seen[n], m.weight.detach().cpu()
        s_in, o_in = O.minmax_tensor(xin, 8, True)
        s_wt, o_wt = O.minmax_tensor(w, 8, True)
        want_x = O.fq_qbase(xin, s_in, o_in, -127, 127, 1 / math.sqrt(xin.numel() * 127))[1]
        want_w = O.fq_qbase(w, s_wt, o_wt, -127, 127, 1 / math.sqrt(w.numel() * 127))[1]
        assert_bits_equal(caps[n].input, want_x, n + ".fq_input")
        assert_bits_equal(caps[n].weight, want_w, n + ".fq_weight")
    for h in hooks:
        h.remove()


def test_merge_bn_and_repvgg_reparam(golden):
    """The two weight transforms that precede quantize_model in the RepAPQ flow (FSPTQuant.py:65-67)."""
    import workloads as W
    from dlmc.utils.merge_bn import DEFAULT_BN_MAPPING_FN, DEFAULT_CONV_MAPPING_FN, merge_bn
    from dlmc.utils.reparam import repvgg_model_convert
    assert DEFAULT_CONV_MAPPING_FN("layer1.conv1.1") == "layer1.conv1.0" and DEFAULT_CONV_MAPPING_FN("layer1.bn1") == "layer1.conv1"
    assert DEFAULT_BN_MAPPING_FN("layer1.conv1.0") == "layer1.conv1.1" and DEFAULT_BN_MAPPING_FN("fc") is None
    for c in golden.of_kind("merge_bn"):
        net = nn.Sequential()
        net.add_module("conv1", nn.Conv2d(4, 6, 3, padding=1, bias=c["has_bias"], groups=c["groups"]))
        net.add_module("bn1", nn.BatchNorm2d(6))
        with torch.no_grad():
            net.conv1.weight.copy_(golden.get(c, "weight"))
            if c["has_bias"]:
                net.conv1.bias.copy_(golden.get(c, "bias"))
            for dst, key in ((net.bn1.weight, "gamma"), (net.bn1.bias, "beta"), (net.bn1.running_mean, "mean"),
                             (net.bn1.running_var, "var")):
                dst.copy_(golden.get(c, key))
        net = net.to(DEV).eval()
        x = torch.randn(2, 4, 8, 8, device=DEV)
        with torch.no_grad():
            ref = net(x)
            merged = merge_bn(net, inplace=True)
            assert isinstance(merged.bn1, nn.Identity) and isinstance(net.bn1, nn.BatchNorm2d)  # a deep copy was merged
            assert_bits_equal(merged.conv1.weight, golden.get(c, "out_weight"), c["name"] + ".weight")
            assert_bits_equal(merged.conv1.bias, golden.get(c, "out_bias"), c["name"] + ".bias")
            close(merged(x), ref.cpu(), c["name"] + " output", rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        merge_bn(nn.Sequential(nn.BatchNorm2d(3)).to(DEV))
    for c in golden.of_kind("repvgg"):
        blk = W.RepVGGTrainBlock(c["cin"], c["cout"], c["stride"], c["groups"])
        sd = {k[4:].replace("__", "."): golden.get(c, k) for k in
              [f.split(".", 1)[1] for f in golden.arr.files if f.startswith(c["name"] + ".pre_")]}
        blk.load_state_dict(sd, strict=False)
        blk = blk.to(DEV).eval()
        x = torch.randn(2, c["cin"], 8, 8, device=DEV)
        with torch.no_grad():
            ref = blk(x)
            dep = repvgg_model_convert(blk)
            assert hasattr(dep, "rbr_reparam") and not hasattr(dep, "rbr_dense") and hasattr(blk, "rbr_dense")
            assert_bits_equal(dep.rbr_reparam.weight, golden.get(c, "out_kernel"), c["name"] + ".kernel")
            assert_bits_equal(dep.rbr_reparam.bias, golden.get(c, "out_bias"), c["name"] + ".bias")
            close(dep(x), ref.cpu(), c["name"] + " output", rtol=1e-4, atol=1e-4)


def test_config5_mobileone_w4a8_asymmetric_per_channel():
    """BASELINE config 5 at small batch: MobileOne-S1 (deploy form, build-owned shapes), W4 asymmetric per-channel
    (`minmax_channel`, unsigned 4 bit: codes 0..15, float offset = channel min, ops.py:129-136), A u8 per tensor;
    nibble pack/unpack of the weight codes; RootQ weight forward at 4 bit."""
    import workloads as W
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    from dlmc.quantization.scalar import modules
    from dlmc.utils.quantize import quantize_model
    from oracle import fakequant_oracle as O
    torch.manual_seed(2333)
    net = W.mobileone_s1_deploy().to(DEV).eval()
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 4, "signed": False}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    quantize_model(net, cfg, None)
    layers = [(n, m) for n, m in net.named_modules() if isinstance(m, modules.QBase)]
    assert len(layers) == 44
    seen = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp, n=n: seen.__setitem__(n, inp[0].detach().cpu())) for n, m in layers]
    caps = {n: Capture(m) for n, m in layers}
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = net(x.to(DEV))
    assert out.shape == (2, 1000) and torch.isfinite(out).all()
    for n, m in layers[:6] + layers[-3:]:
        xin, w = seen[n], m.weight.detach().cpu()
        s_in, o_in = O.minmax_tensor(xin, 8, False)
        s_wt, o_wt = O.minmax_channel(w, 4, False, ch_axis=0)
        assert m.wt_scale.shape == s_wt.shape
        want_x = O.fq_qbase(xin, s_in, o_in, 0, 255, 1 / math.sqrt(xin.numel() * 255))[1]
        qw, want_w = O.fq_qbase(w, s_wt, o_wt, 0, 15, 1 / math.sqrt(w.numel() * 15))
        assert_bits_equal(caps[n].input, want_x, n + ".fq_input")
        assert_bits_equal(caps[n].weight, want_w, n + ".fq_weight")
        # W4 codes: emitted packed (two per byte), unpacked, and dequantised back to the same fake-quant weights
        g_w = 1 / math.sqrt(w.numel() * 15)
        _, packed = K.fake_quant(m.weight.detach(), m.wt_scale, m.wt_offset, 0, 15, N.FORM_QBASE, g=g_w, codes="p4", want_y=False)
        assert packed.numel() == (w.numel() + 1) // 2
        codes = K.unpack_int4(packed, w.numel(), False).cpu().to(torch.float32).reshape(w.shape)
        assert torch.equal(codes, qw), n + ".w4 codes"
        assert torch.equal(K.pack_int4(codes.to(torch.int8).to(DEV)).cpu(), packed.cpu()), n + ".pack"
        back = K.dequant_codes(packed, w.shape, m.wt_scale, m.wt_offset, N.FORM_QBASE, "p4", False, g=g_w)
        assert_bits_equal(back, want_w, n + ".dequant(p4)")
    for h in hooks:
        h.remove()
    # RootQ weight forward at 4 bit on a pointwise weight of the net
    w = layers[4][1].weight.detach()
    up, lw = (2 * w.abs().mean() * math.sqrt(15)).cpu(), (-2 * w.abs().mean() * math.sqrt(15)).cpu()
    want = O.fq_rootq_weight(w.cpu(), up, lw, torch.tensor(0.25), 0, 15)[2]
    assert_bits_equal(K.rootq_weight(w, up.to(DEV), lw.to(DEV), 0, 15), want, "rootq w4")


def test_integer_checkpoint_roundtrip():
    """int8 / packed-int4 export -> load into a fresh quantised model: identical outputs, 4-8x smaller weights."""
    from dlmc.utils.export import export_quantized_state, load_quantized_state
    from dlmc.utils.quantize import quantize_model

    def build(family, wbits, wsigned, wtype):
        torch.manual_seed(2333)
        net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.ReLU(), nn.Conv2d(16, 32, 3, stride=2, padding=1), nn.ReLU(),
                            nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(32, 10)).to(DEV).eval()
        cfg = {"weight": {"enable": True, "type": wtype, "recon_type": "None", "args": {"n_bits": wbits, "signed": wsigned}},
               "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
               "exclude_layers": [], "override_options": []}
        quantize_model(net, cfg, None, quantization_type=family)
        return net

    x = torch.randn(4, 3, 16, 16, device=DEV)
    for family, wbits, wsigned, wtype in (("FSPTQ", 8, True, "minmax_channel"), ("FSPTQ", 4, True, "minmax_channel"),
                                          (None, 8, True, "minmax_tensor"), (None, 4, False, "minmax_channel")):
        a = build(family, wbits, wsigned, wtype)
        with torch.no_grad():
            ref = a(x)
        blob = export_quantized_state(a)
        buf = io.BytesIO()
        torch.save(blob, buf)
        buf.seek(0)
        blob = torch.load(buf, weights_only=False)
        for name, rec in blob["layers"].items():
            n = math.prod(rec["shape"])
            assert rec["codes"].numel() == ((n + 1) // 2 if wbits <= 4 else n) and rec["codes"].element_size() == 1
        b = build(family, wbits, wsigned, wtype)
        with torch.no_grad():
            b[0].weight.mul_(3.0)                      # make sure the weights really come from the file
        load_quantized_state(b, blob)
        with torch.no_grad():
            out = b(x)                                 # no re-calibration: init flags came with the file
        assert torch.equal(out, ref), (family, wbits, wsigned)
    with pytest.raises(RuntimeError, match="calibrated"):
        export_quantized_state(build("FSPTQ", 8, True, "minmax_channel"))


def test_graphed_forward_matches_eager():
    """The whole quantised forward as one HIP graph (launch-bound small-batch case, BASELINE config 1)."""
    import time
    import workloads as W
    from dlmc.utils.graph import GraphedForward
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    net = W.resnet18().to(DEV).eval()
    cfg = {"weight": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
           "exclude_layers": [], "override_options": []}
    quantize_model(net, cfg, None)
    x = torch.randn(1, 3, 224, 224, device=DEV)
    with torch.no_grad():
        net(x)                                   # calibrate eagerly
        ref = net(x * 0.9)
    fwd = GraphedForward(net, x)
    out = fwd(x * 0.9)
    assert torch.equal(out, ref)
    with pytest.raises(ValueError):
        fwd(torch.randn(2, 3, 224, 224, device=DEV))
    # it is also what makes batch 1 fast: report (not assert) the two latencies
    def bench(f, n=30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        eager_ms, graph_ms = bench(net), bench(fwd)
    print(f"resnet18 b1 W8A8: eager {eager_ms:.2f} ms, graph {graph_ms:.2f} ms")
    assert graph_ms < eager_ms


def test_adaround_fused_matches_the_reference_chain():
    """Fused AdaRound forward / backward against the reference's op chain run through autograd on the CPU."""
    from dlmc.quantization.scalar import kernels as K
    from oracle import fakequant_oracle as O
    g = torch.Generator().manual_seed(2333)
    for shape, (lo, hi) in (((16, 8, 3, 3), (-127, 127)), ((10, 33), (-7, 7)), ((4, 5, 1, 1), (-3, 3))):
        w = torch.randn(shape, generator=g) * 0.1
        s, _ = O.minmax_channel(w, 8 if hi == 127 else (4 if hi == 7 else 3), True, ch_axis=0)
        s = s + 1e-6
        alpha = torch.randn(shape, generator=g) * 2.0
        gy = torch.randn(shape, generator=g)
        # eval form: bit-exact
        _, y_eval = O.fq_adaround(w, s, alpha, lo, hi, training=False)
        assert_bits_equal(K.adaround_weight(w.to(DEV), alpha.to(DEV), s.to(DEV), lo, hi, False), y_eval, f"{shape} eval")
        # training form + gradients (sigmoid differs in the last ulp between CPU and GPU exp)
        a_ref, s_ref = alpha.clone().requires_grad_(True), s.clone().requires_grad_(True)
        q = torch.floor(w / s_ref) + torch.clamp(torch.sigmoid(a_ref) * (1.1 - (-0.1)) + (-0.1), 0, 1)
        y_ref = q.clamp(lo, hi) * s_ref
        y_ref.backward(gy)
        y = K.adaround_weight(w.to(DEV), alpha.to(DEV), s.to(DEV), lo, hi, True)
        close(y, y_ref.detach(), f"{shape} train", rtol=1e-5, atol=1e-6)
        ga, gs = K.adaround_weight_backward(w.to(DEV), alpha.to(DEV), s.to(DEV), gy.to(DEV), lo, hi)
        close(ga, a_ref.grad, f"{shape} g_alpha", rtol=1e-4, atol=1e-6)
        # the reference's scale gradient also receives the (zero-measure) floor(w/s) path: only the product term
        want_gs = (gy * q.detach().clamp(lo, hi)).sum(dim=tuple(range(1, w.dim())), keepdim=True)
        close(gs, want_gs, f"{shape} g_scale", rtol=1e-4, atol=1e-5)
        assert gs.shape == s.shape


def test_output_aware_weight_scale_and_function_api(golden):
    """`l2norm_output` through QBase (ops.py:85-109; PTQ_output_config.yaml) and the explicit autograd Functions."""
    from dlmc.quantization.scalar.modules import FunLQ, FunLSQ, FunRootQ, FunUniformQ
    for c in golden.of_kind("qbase_l2out"):
        net, cap = _quantized(c, golden)
        with torch.no_grad():
            out = net(golden.get(c, "x").to(DEV))
        assert_bits_equal(cap.input, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        close(net.layer.wt_scale, golden.get(c, "wt_scale"), c["name"] + ".wt_scale", rtol=2e-4, atol=0)
        close(out, golden.get(c, "out"), c["name"] + ".out", rtol=5e-2, atol=5e-2)   # 4-bit weights: a scale ulp moves codes
    for c in golden.of_kind("funlsq"):
        w, s, gout = (golden.get(c, k).to(DEV) for k in ("w", "scale", "gout"))
        wr, sr = w.clone().requires_grad_(True), s.clone().requires_grad_(True)
        y = FunLSQ.apply(wr, sr, torch.zeros(1, device=DEV), c["lo"], c["hi"], c["g"])
        assert_bits_equal(y, golden.get(c, "y"), c["name"] + ".y")
        y.backward(gout)
        assert_bits_equal(wr.grad, golden.get(c, "grad_w"), c["name"] + ".grad_w")
        close(sr.grad, golden.get(c, "grad_scale"), c["name"] + ".grad_scale", rtol=1e-4, atol=1e-6)
        for fn, nargs in ((FunUniformQ, 5), (FunRootQ, 5), (FunLQ, 6)):
            wr = w.clone().requires_grad_(True)
            args = (wr, s, torch.zeros(1, device=DEV), c["lo"], c["hi"]) + ((c["g"],) if nargs == 6 else ())
            y = fn.apply(*args)
            y.sum().backward()
            assert wr.grad is not None and wr.grad.shape == w.shape
            if fn is not FunLQ:
                assert_bits_equal(y, golden.get(c, "y"), fn.__name__)


def test_block_reconstruction_stays_on_the_device():
    """hooked activations collected in HBM, AdaRound optimised with the fused kernels; the block's
    output error against its fp32 twin must drop."""
    import copy
    import workloads as W
    from dlmc.quantization.scalar.FSPTQuant import FSPTQBase
    from dlmc.utils.quantize import quantize_model
    from dlmc.utils.reconstruct import collect_block_io, reconstruct_block
    torch.manual_seed(2333)
    from dlmc.utils.merge_bn import merge_bn
    fp = merge_bn(W.resnet18().to(DEV).eval(), inplace=True)      # FSPTQuant.py:67: BN is folded before quantize_model
    net = copy.deepcopy(fp)
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "recon_type": "adaround", "args": {"n_bits": 3, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    quantize_model(net, cfg, None, "FSPTQ")
    batches = [torch.relu(torch.randn(8, 3, 64, 64, device=DEV)) for _ in range(4)]
    with torch.no_grad():
        net(batches[0])                                   # calibrate (scales, alpha)
    block, fp_block = net.layer1[0], fp.layer1[0]
    xin, yout = collect_block_io(net, fp, block, fp_block, batches, total=32)
    assert xin.is_cuda and yout.is_cuda and xin.shape == (32, 64, 16, 16) and yout.shape == (32, 64, 16, 16)
    with pytest.raises(RuntimeError):
        collect_block_io(net, fp, block, fp_block, batches, total=31)
    for p in block.parameters():
        p.requires_grad_(False)
    for m in block.modules():
        if isinstance(m, FSPTQBase):
            m.alpha.requires_grad_(True)
    block.train()                                          # the loop optimises the soft-rounded block (base.py:136-141)
    with torch.no_grad():
        before = float((block(xin) - yout).pow(2).mean())
    groups = [{"params": [m.alpha for m in block.modules() if isinstance(m, FSPTQBase)], "lr": 1e-2}]
    last = reconstruct_block(block, xin, yout, iters=80, batch=16, param_groups=groups)
    assert block.training and torch.isfinite(last)
    with torch.no_grad():
        after = float((block(xin) - yout).pow(2).mean())
    print(f"block reconstruction (soft rounding): l2 {before:.5f} -> {after:.5f}")
    assert after < 0.9 * before


def test_lsq_initialisation_against_the_reference(golden):
    """`type: "LSQ"` (modules/base.py:84-85 input, :118-121 weight; LSQ_config.yaml, the QAT flow's default) through QConv2d /
    QLinear on the GPU against golden_v2 (the reference's own forward).  The scale 2 * mean|x| / sqrt(Qp) is one HIP launch
    (dlmcq_lsq_init_f32); its mean is summed in double precision in a fixed order, the reference's CPU run sums fp32 in ATen's
    order, so the SCALE is compared to 2e-6 relative (stated tolerance; everything after the mean is the same fp32 chain with a
    true division).  With the reference's scales loaded the fake-quantised operands are then compared bit for bit."""
    cases = golden.of_kind("lsq")
    assert len(cases) == 12
    exact = 0
    for c in cases:
        net, cap = _quantized(c, golden)
        q = net.layer
        x, x2 = golden.get(c, "x").to(DEV), golden.get(c, "x2").to(DEV)
        with torch.no_grad():
            out = net(x)
            for name in ("in_scale", "wt_scale"):
                got, want = getattr(q, name).detach().cpu().reshape(-1), golden.get(c, name).reshape(-1)
                torch.testing.assert_close(got, want, rtol=2e-6, atol=0, msg=lambda m: f"{c['name']}.{name}: {m}")
                exact += int(torch.equal(got, want))
            same_values(q.in_offset.reshape(-1), golden.get(c, "in_offset"), c["name"] + ".in_offset")
            same_values(q.wt_offset.reshape(-1), golden.get(c, "wt_offset"), c["name"] + ".wt_offset")
            assert float(q.in_init_state) == 1 and float(q.wt_init_state) == 1
            # a scale that differs in its last bit moves every fake-quantised value by that much, and a value on a rounding tie by one step
            step_in, step_wt = float(golden.get(c, "in_scale")), float(golden.get(c, "wt_scale"))
            for got, want, step, what in ((cap.input, golden.get(c, "fq_input"), step_in, "fq_input"),
                                          (cap.weight, golden.get(c, "fq_weight"), step_wt, "fq_weight")):
                d = (got.cpu() - want).abs()
                assert float(d.max()) <= step * 1.0001 and float((d > 1e-5 * want.abs() + 1e-9).float().mean()) <= 0.01, f"{c['name']}.{what}"
            close(out, golden.get(c, "out"), c["name"] + ".out", rtol=1e-3, atol=2 * (step_in + step_wt))
            # the reference's own scales in place: the operands bit for bit, on new data (frozen scales)
            q.in_scale.copy_(golden.get(c, "in_scale").to(DEV))
            q.wt_scale.copy_(golden.get(c, "wt_scale").to(DEV))
            out2 = net(x2)
            assert_bits_equal(cap.input, golden.get(c, "fq_input2"), c["name"] + ".fq_input2")
            assert_bits_equal(cap.weight, golden.get(c, "fq_weight"), c["name"] + ".fq_weight (reference scale)")
            close(out2, golden.get(c, "out2"), c["name"] + ".out2")
        assert sorted(q.state_dict().keys()) == c["state_keys"]
    print(f"LSQ scales bit-identical to the reference's: {exact} of {2 * len(cases)}")


def test_lsq_init_kernel_shapes_and_errors():
    """dlmcq_lsq_init_f32 on sizes around its vector width and grid (1, 3, 4, 255, 2^20 + 3 elements, a channels_last view) against
    float64 arithmetic; argument checks."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(5)
    for n in (1, 3, 4, 5, 255, 4096, (1 << 20) + 3):
        x = torch.randn(n, device=DEV, generator=g) * 3
        want = 2 * (x.double().abs().sum() / n).float() / torch.tensor(math.sqrt(127), dtype=torch.float32, device=DEV)
        torch.testing.assert_close(K.lsq_init(x, 127).reshape(()), want, rtol=3e-7, atol=0)
    x = torch.randn(8, 16, 14, 14, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(K.lsq_init(x, 7), K.lsq_init(x.contiguous(), 7), rtol=3e-7, atol=0)
    one = torch.zeros(4, device=DEV)
    assert N.lib.dlmcq_lsq_init_f32(N.ptr(one), N.ptr(one), 0, 1.0, N.ptr(one), 16384, None) == -1      # empty mean
    assert N.lib.dlmcq_lsq_init_f32(N.ptr(one), N.ptr(one), 4, 1.0, None, 0, None) == -3                  # no scratch


def test_minmax_pixel_against_the_reference(golden):
    """ops.py:142-167 through get_qparams_tensor(qtype="minmax_pixel") on the GPU against golden_v2: bit for bit (max / min are
    exact), the |x|-minimum of the unsigned branch (ops.py:156) included, 4-D and 3-D tensors, allow_offset=False."""
    from dlmc.quantization.scalar import ops
    cases = golden.of_kind("minmax_pixel")
    assert len(cases) == 16
    for c in cases:
        x = golden.get(c, "x").to(DEV)
        s, o = ops.get_qparams_tensor(x, "minmax_pixel", n_bits=c["n_bits"], signed=c["signed"])
        want = golden.get(c, "scale")
        assert list(s.shape) == list(want.shape)
        assert_bits_equal(s, want, c["name"] + ".scale")
        assert_bits_equal(o, golden.get(c, "offset"), c["name"] + ".offset")
        if golden.has(c, "scale_nooff"):
            s, o = ops.get_qparams_tensor(x, "minmax_pixel", n_bits=c["n_bits"], signed=c["signed"], allow_offset=False)
            assert_bits_equal(s, golden.get(c, "scale_nooff"), c["name"] + ".scale_nooff")
            assert_bits_equal(o, golden.get(c, "offset_nooff"), c["name"] + ".offset_nooff")
