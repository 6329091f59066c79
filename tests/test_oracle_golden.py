"""The oracle pinned to the reference: every function of oracle/fakequant_oracle.py against the
golden vectors that the reference's own code produced (tests/golden/make_golden.py)."""
import math

import pytest
import torch

from _cmp import assert_bits_equal


def assert_out_close(got, want, what=""):
    """Conv / linear outputs: the accumulation order inside F.conv2d / F.linear is a third-party
    detail (thread count, blocking), so layer outputs are compared to fp32 tolerance while the
    fake-quantised operands feeding them are compared bit for bit."""
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5, msg=lambda m: f"{what}: {m}")
from oracle import fakequant_oracle as O


def _layer(case, golden):
    kind = case["layer"]
    w = golden.get(case, "weight")
    b = golden.get(case, "bias")
    if kind == "linear":
        m = torch.nn.Linear(w.shape[1], w.shape[0], bias=b.numel() > 0)
    else:
        kw = dict(conv=dict(padding=1), conv_s2=dict(stride=2, padding=1),
                  conv_reflect=dict(padding=1, padding_mode="reflect"),
                  conv_group=dict(padding=1, groups=2))[kind]
        groups = kw.get("groups", 1)
        m = torch.nn.Conv2d(w.shape[1] * groups, w.shape[0], w.shape[2], bias=b.numel() > 0, **kw)
    with torch.no_grad():
        m.weight.copy_(w)
        if b.numel():
            m.bias.copy_(b)
    return m


def test_qrange():
    assert O.qrange(True, 8) == (-127, 127)
    assert O.qrange(False, 8) == (0, 255)
    assert O.qrange(True, 4) == (-7, 7)
    assert O.qrange(False, 4) == (0, 15)
    assert O.qrange(False, 2) == (0, 3)


def test_primitives(golden):
    cases = golden.of_kind("primitive")
    assert len(cases) >= 50
    for c in cases:
        x, s, o = (golden.get(c, k) for k in ("x", "scale", "offset"))
        q, y = O.fq_emulate(x, s, o, c["lo"], c["hi"])
        assert_bits_equal(q, golden.get(c, "q"), c["name"] + ".q")
        assert_bits_equal(y, golden.get(c, "y"), c["name"] + ".y")


def test_observers(golden):
    cases = golden.of_kind("observer")
    assert len(cases) >= 30
    for c in cases:
        x = golden.get(c, "x")
        s, o = O.minmax_tensor(x, c["n_bits"], c["signed"])
        assert_bits_equal(s, golden.get(c, "t_scale"), c["name"] + ".t_scale")
        assert_bits_equal(o, golden.get(c, "t_offset"), c["name"] + ".t_offset")
        if golden.has(c, "c_scale"):
            s, o = O.minmax_channel(x, c["n_bits"], c["signed"], ch_axis=c["ch_axis"])
            want = golden.get(c, "c_scale")
            assert list(s.shape) == list(want.shape)
            assert_bits_equal(s, want, c["name"] + ".c_scale")
            assert_bits_equal(o, golden.get(c, "c_offset"), c["name"] + ".c_offset")
        if golden.has(c, "t_scale_nooff"):
            s, o = O.minmax_tensor(x, c["n_bits"], c["signed"], allow_offset=False)
            assert_bits_equal(s, golden.get(c, "t_scale_nooff"))
            assert_bits_equal(o, golden.get(c, "t_offset_nooff"))
        if golden.has(c, "c_scale_nooff"):
            s, o = O.minmax_channel(x, c["n_bits"], c["signed"], ch_axis=c["ch_axis"], allow_offset=False)
            assert_bits_equal(s, golden.get(c, "c_scale_nooff"))
            assert_bits_equal(o, golden.get(c, "c_offset_nooff"))


def test_qbase_forward(golden):
    cases = golden.of_kind("qbase")
    assert len(cases) == 25
    for c in cases:
        m = _layer(c, golden)
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        irng, wrng = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x = golden.get(c, "x")
        # first call = observer (ops.py:20-34) then fake-quant
        s_in, o_in = O.minmax_tensor(x, ia["n_bits"], ia["signed"])
        s_wt, o_wt = O.minmax_tensor(m.weight.detach(), wa["n_bits"], wa["signed"])
        assert_bits_equal(s_in, golden.get(c, "in_scale"), c["name"] + ".in_scale")
        assert_bits_equal(o_in, golden.get(c, "in_offset"))
        assert_bits_equal(s_wt, golden.get(c, "wt_scale"))
        assert_bits_equal(o_wt, golden.get(c, "wt_offset"))
        xq, wq, out = O.qbase_layer_forward(m, x, s_in.reshape(1), o_in, s_wt.reshape(1), o_wt, irng, wrng)
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        assert_out_close(out, golden.get(c, "out"), c["name"] + ".out")
        xq2, _, out2 = O.qbase_layer_forward(m, golden.get(c, "x2"), s_in.reshape(1), o_in, s_wt.reshape(1), o_wt, irng, wrng)
        assert_bits_equal(xq2, golden.get(c, "fq_input2"))
        assert_out_close(out2, golden.get(c, "out2"))
        assert c["state_keys"] == sorted(
            ["in_init_state", "in_offset", "in_scale", "weight", "wt_init_state", "wt_offset", "wt_scale"]
            + (["bias"] if m.bias is not None else []))


def test_qbase_backward(golden):
    for c in golden.of_kind("qbase_grad"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        irng, wrng = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w, gout = golden.get(c, "x"), golden.get(c, "weight"), golden.get(c, "gout")
        s_in, o_in = golden.get(c, "in_scale"), golden.get(c, "in_offset")
        s_wt, o_wt = golden.get(c, "wt_scale"), golden.get(c, "wt_offset")
        g_i = 1 / math.sqrt(x.numel() * irng[1])
        g_w = 1 / math.sqrt(w.numel() * wrng[1])
        _, xq = O.fq_qbase(x, s_in, o_in, irng[0], irng[1], g_i)
        _, wq = O.fq_qbase(w, s_wt, o_wt, wrng[0], wrng[1], g_w)
        assert_bits_equal(xq, golden.get(c, "fq_input") if golden.has(c, "fq_input") else xq)
        out = torch.nn.functional.conv2d(xq, wq, golden.get(c, "bias"), padding=1)
        assert_out_close(out, golden.get(c, "out"))
        # upstream gradients captured from the reference run (the conv backward is third party)
        gx, gs_in = O.qbase_backward(x, s_in, o_in, golden.get(c, "g_fq_input"), irng[0], irng[1], g_i)
        gw, gs_wt = O.qbase_backward(w, s_wt, o_wt, golden.get(c, "g_fq_weight"), wrng[0], wrng[1], g_w)
        assert_bits_equal(gx, golden.get(c, "grad_x"), c["name"] + ".grad_x")
        assert_bits_equal(gw, golden.get(c, "grad_weight"), c["name"] + ".grad_weight")
        # scale gradients are fp32 sums over the whole tensor: order-dependent -> tolerance
        torch.testing.assert_close(gs_in.reshape(1), golden.get(c, "grad_in_scale"), rtol=2e-4, atol=1e-6)
        torch.testing.assert_close(gs_wt.reshape(1), golden.get(c, "grad_wt_scale"), rtol=2e-4, atol=1e-6)


def test_fsptq_backward(golden):
    """FSPTQ's live path under autograd (reference-made gradients, tests/golden/golden_v1_grad.*): the oracle's node-by-node
    restatement gives the input / weight gradients bit for bit and the scale gradients to summation order."""
    cases = golden.of_kind("fsptq_grad")
    assert len(cases) == 3
    for c in cases:
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        irng, wrng = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x"), golden.get(c, "weight")
        s_in, zp, s_wt = golden.get(c, "in_scale"), golden.get(c, "in_offset"), golden.get(c, "wt_scale")
        _, xq = O.fq_zeropoint(x, s_in, zp, irng[0], irng[1])
        _, wq = O.fq_symmetric(w, s_wt, wrng[0], wrng[1])
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        gx, gs_in = O.fsptq_act_backward(x, s_in, zp, golden.get(c, "g_fq_input"), irng[0], irng[1])
        gw, gs_wt = O.fsptq_weight_backward(w, s_wt, golden.get(c, "g_fq_weight"), wrng[0], wrng[1])
        assert_bits_equal(gx, golden.get(c, "grad_x"), c["name"] + ".grad_x")
        assert_bits_equal(gw, golden.get(c, "grad_weight"), c["name"] + ".grad_weight")
        torch.testing.assert_close(gs_in.reshape(-1), golden.get(c, "grad_in_scale").reshape(-1), rtol=2e-4, atol=1e-6)
        torch.testing.assert_close(gs_wt.reshape(-1), golden.get(c, "grad_wt_scale").reshape(-1), rtol=2e-4, atol=1e-5)


def test_funlsq_closed_form(golden):
    for c in golden.of_kind("funlsq"):
        w, s, gout = golden.get(c, "w"), golden.get(c, "scale"), golden.get(c, "gout")
        _, y = O.fq_emulate(w, s, torch.zeros(1), c["lo"], c["hi"])
        assert_bits_equal(y, golden.get(c, "y"))
        gw, gs = O.lsq_backward(w, s, gout, c["lo"], c["hi"], c["g"])
        assert_bits_equal(gw, golden.get(c, "grad_w"))
        assert_bits_equal(gs, golden.get(c, "grad_scale"))


def test_fsptq_forward(golden):
    cases = golden.of_kind("fsptq")
    assert len(cases) == 20
    for c in cases:
        m = _layer(c, golden)
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        irng, wrng = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x"), m.weight.detach()
        s_in, zp = O.minmax_tensor(x, ia["n_bits"], ia["signed"])
        s_wt, o_wt = O.minmax_channel(w, wa["n_bits"], wa["signed"], ch_axis=0)
        s_wt = s_wt + 1e-6  # FSPTQuant/base.py:129
        assert_bits_equal(s_in, golden.get(c, "in_scale"))
        assert_bits_equal(zp, golden.get(c, "in_offset"))
        assert_bits_equal(s_wt, golden.get(c, "wt_scale"), c["name"] + ".wt_scale")
        assert list(s_wt.shape) == list(golden.get(c, "wt_scale").shape)
        assert_bits_equal(o_wt, golden.get(c, "wt_offset"))
        _, xq = O.fq_zeropoint(x, s_in.reshape(1), zp, *irng)
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        _, xq2 = O.fq_zeropoint(golden.get(c, "x2"), s_in.reshape(1), zp, *irng)
        assert_bits_equal(xq2, golden.get(c, "fq_input2"))
        if c["qconfig"]["weight"]["recon_type"] == "adaround":
            a0 = O.adaround_init_alpha(w, s_wt)
            assert_bits_equal(a0, golden.get(c, "alpha_init"), c["name"] + ".alpha_init")
            _, wq = O.fq_adaround(w, s_wt, a0, *wrng, training=False)
            assert_bits_equal(wq, golden.get(c, "fq_weight"))
            alpha = golden.get(c, "alpha")
            _, wq_e = O.fq_adaround(w, s_wt, alpha, *wrng, training=False)
            assert_bits_equal(wq_e, golden.get(c, "fq_weight_eval"))
            _, wq_t = O.fq_adaround(w, s_wt, alpha, *wrng, training=True)
            assert_bits_equal(wq_t, golden.get(c, "fq_weight_train"))
            assert_bits_equal(O.adaround_soft_targets(alpha), golden.get(c, "soft_targets"))
            assert_out_close(O.conv_or_linear(m, xq, wq_t), golden.get(c, "out_train"))
        else:
            xq_, wq, out = O.fsptq_layer_forward(m, x, s_in.reshape(1), zp, s_wt, irng, wrng)
            assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
            assert_out_close(out, golden.get(c, "out"), c["name"] + ".out")
        keys = ["in_init_state", "in_offset", "in_scale", "org_weight", "weight", "wt_init_state",
                "wt_offset", "wt_scale"] + (["bias"] if m.bias is not None else []) + \
               (["alpha"] if c["qconfig"]["weight"]["recon_type"] == "adaround" else [])
        assert c["state_keys"] == sorted(keys)


def test_rootq_forward(golden):
    cases = golden.of_kind("rootq")
    assert len(cases) == 15
    mom = 0.1
    for c in cases:
        m = _layer(c, golden)
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        (ilo, ihi), (wlo, whi) = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w, alpha = golden.get(c, "x"), m.weight.detach(), golden.get(c, "wt_alpha")
        # init (RootQ/base.py:79-90,113-129)
        in_scale = (x.max() - x.min()) / (ihi - ilo)
        mean_abs = w.abs().mean()
        wt_max = 2 * mean_abs * math.sqrt(whi)
        wt_min = -2 * mean_abs * math.sqrt(whi)
        assert_bits_equal(in_scale, golden.get(c, "st_in_scale"))
        assert_bits_equal(wt_max, golden.get(c, "st_wt_upper"))
        assert_bits_equal(wt_min, golden.get(c, "st_wt_lower"))
        _, xq = O.fq_rootq_act(x, in_scale, ilo, ihi)
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        _, _, wq = O.fq_rootq_weight(w, wt_max, wt_min, alpha, wlo, whi)
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        assert_out_close(O.conv_or_linear(m, xq, wq), golden.get(c, "out"))
        # one train-mode step with moved parameters (EMA, RootQ/base.py:92-101,131-142)
        x2 = golden.get(c, "x2")
        p_in, p_up, p_lo = in_scale * 0.8, wt_max * 0.9, wt_min * 0.85
        assert_bits_equal(p_in, golden.get(c, "tr_in_scale"))
        g_i = 1 / math.sqrt(x2.numel() * ihi)
        g_w = 1 / math.sqrt(w.numel() * whi)
        run_in = O.rootq_ema(in_scale, p_in, mom, g_i)
        run_up = O.rootq_ema(wt_max, p_up, mom, g_w)
        run_lo = O.rootq_ema(wt_min, p_lo, mom, g_w)
        assert_bits_equal(run_in, golden.get(c, "tr_in_run_scale"), c["name"] + ".run_scale")
        assert_bits_equal(run_up, golden.get(c, "tr_wt_run_upper"))
        assert_bits_equal(run_lo, golden.get(c, "tr_wt_run_lower"))
        _, xq_t = O.fq_rootq_act(x2, run_in, ilo, ihi)
        _, _, wq_t = O.fq_rootq_weight(w, run_up, run_lo, alpha, wlo, whi)
        assert_bits_equal(xq_t, golden.get(c, "fq_input_train"))
        assert_bits_equal(wq_t, golden.get(c, "fq_weight_train"))
        assert_bits_equal(xq_t, golden.get(c, "fq_input_eval2"))  # eval re-uses the running values
        assert_bits_equal(wq_t, golden.get(c, "fq_weight_eval2"))
        assert c["state_keys"] == sorted(
            ["in_scale", "in_run_upper", "in_run_scale", "in_init_state", "wt_upper", "wt_lower",
             "wt_alpha", "wt_run_upper", "wt_run_lower", "wt_init_state", "weight"]
            + (["bias"] if m.bias is not None else []))


def test_estimators(golden):
    for c in golden.of_kind("estimator"):
        x = golden.get(c, "x")
        s, o = O.l2norm_tensor(x, c["n_bits"], c["signed"])
        torch.testing.assert_close(s, golden.get(c, "l2norm_t_scale"), rtol=1e-5, atol=0)
        s, o = O.l2norm_channel(x, c["n_bits"], c["signed"], ch_axis=c["ch_axis"])
        torch.testing.assert_close(s, golden.get(c, "l2norm_c_scale"), rtol=1e-5, atol=0)
        assert_bits_equal(o, golden.get(c, "l2norm_c_offset"))
        s, o = O.l2loss_tensor(x, c["n_bits"], c["signed"])
        assert_bits_equal(s, golden.get(c, "l2loss_t_scale"))
        assert_bits_equal(o, golden.get(c, "l2loss_t_offset"))
        if golden.has(c, "l2loss_c_scale"):
            s, o = O.l2loss_channel(x, c["n_bits"], c["signed"], ch_axis=c["ch_axis"])
            assert_bits_equal(s, golden.get(c, "l2loss_c_scale"))
            assert_bits_equal(o, golden.get(c, "l2loss_c_offset"))


def test_weight_transforms(golden):
    for c in golden.of_kind("merge_bn"):
        b = golden.get(c, "bias")
        w, bo = O.fold_bn(golden.get(c, "weight"), b if b.numel() else None, golden.get(c, "gamma"), golden.get(c, "beta"),
                          golden.get(c, "mean"), golden.get(c, "var"))
        assert_bits_equal(w, golden.get(c, "out_weight"), c["name"] + ".weight")
        assert_bits_equal(bo, golden.get(c, "out_bias"), c["name"] + ".bias")
        assert c["bn_replaced_by"] == "Identity"
    for c in golden.of_kind("repvgg"):
        def bn(prefix, eps=1e-5):
            return tuple(golden.get(c, f"pre_{prefix}__{k}") for k in ("weight", "bias", "running_mean", "running_var")) + (eps,)
        k, b = O.repvgg_fuse(golden.get(c, "pre_rbr_dense__conv__weight"), bn("rbr_dense__bn"),
                             golden.get(c, "pre_rbr_1x1__conv__weight"), bn("rbr_1x1__bn"),
                             bn("rbr_identity") if c["has_identity"] else None, groups=c["groups"])
        assert_bits_equal(k, golden.get(c, "out_kernel"), c["name"] + ".kernel")
        assert_bits_equal(b, golden.get(c, "out_bias"), c["name"] + ".bias")


def test_output_aware_estimator(golden):
    for c in golden.of_kind("qbase_l2out"):
        m = _layer(c, golden)
        x = golden.get(c, "x")
        s_in, o_in = O.minmax_tensor(x, 8, True)
        _, xq = O.fq_qbase(x, s_in.reshape(1), o_in, -127, 127, 1 / math.sqrt(x.numel() * 127))
        assert_bits_equal(xq, golden.get(c, "fq_input"))
        s, o = O.l2norm_output(m, xq, m.weight.detach(), 4, True)
        torch.testing.assert_close(s.reshape(1), golden.get(c, "wt_scale"), rtol=1e-4, atol=0)


def test_lsq_initialisation(golden):
    """`type: "LSQ"` (modules/base.py:84-85,118-121; the QAT flow's default, LSQ_config.yaml): 2 * mean|x| / sqrt(Qp), then the
    QBase forward with that scale - golden_v2, produced by the reference's own forward."""
    cases = golden.of_kind("lsq")
    assert len(cases) == 12
    for c in cases:
        m = _layer(c, golden)
        ic, wc = c["qconfig"]["input"], c["qconfig"]["weight"]
        ia, wa = ic["args"], wc["args"]
        irng, wrng = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x = golden.get(c, "x")
        if ic["type"] == "LSQ":
            s_in, o_in = O.lsq_init(x, irng[1]), torch.zeros(())
        else:
            s_in, o_in = O.minmax_tensor(x, ia["n_bits"], ia["signed"])
        assert_bits_equal(s_in.reshape(1), golden.get(c, "in_scale"), c["name"] + ".in_scale")
        assert_bits_equal(o_in.reshape(-1), golden.get(c, "in_offset"), c["name"] + ".in_offset")
        xq, _ = O.fq_qbase(x, s_in.reshape(1), o_in, irng[0], irng[1], 1 / math.sqrt(x.numel() * irng[1]))[::-1]
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        # the weight's LSQ scale (base.py:118-121) is taken AFTER the input was fake-quantised, from the raw weight
        w = m.weight.detach()
        s_wt, o_wt = (O.lsq_init(w, wrng[1]), torch.zeros(())) if wc["type"] == "LSQ" else O.minmax_tensor(w, wa["n_bits"], wa["signed"])
        assert_bits_equal(s_wt.reshape(1), golden.get(c, "wt_scale"), c["name"] + ".wt_scale")
        xq_, wq, out = O.qbase_layer_forward(m, x, s_in.reshape(1), o_in, s_wt.reshape(1), o_wt, irng, wrng)
        assert_bits_equal(xq_, golden.get(c, "fq_input"))
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        assert_out_close(out, golden.get(c, "out"), c["name"] + ".out")
        xq2, _, out2 = O.qbase_layer_forward(m, golden.get(c, "x2"), s_in.reshape(1), o_in, s_wt.reshape(1), o_wt, irng, wrng)
        assert_bits_equal(xq2, golden.get(c, "fq_input2"))
        assert_out_close(out2, golden.get(c, "out2"))


def test_minmax_pixel(golden):
    """ops.py:142-167, with the reference's |x|-minimum in the unsigned branch."""
    cases = golden.of_kind("minmax_pixel")
    assert len(cases) == 16
    for c in cases:
        x = golden.get(c, "x")
        s, o = O.minmax_pixel(x, c["n_bits"], c["signed"])
        want = golden.get(c, "scale")
        assert list(s.shape) == list(want.shape) == list(x.shape[2:])
        assert_bits_equal(s, want, c["name"] + ".scale")
        assert_bits_equal(o, golden.get(c, "offset"), c["name"] + ".offset")
        if golden.has(c, "scale_nooff"):
            s, o = O.minmax_pixel(x, c["n_bits"], c["signed"], allow_offset=False)
            assert_bits_equal(s, golden.get(c, "scale_nooff"))
            assert_bits_equal(o, golden.get(c, "offset_nooff"))
