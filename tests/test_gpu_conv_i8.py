"""The fused int8 conv / linear (K9) against an exact reference: float64 convolution of the dequantised
operands on the CPU.  With unit scales and small codes the fp32 result must be the exact integer."""
import pytest
import torch
import torch.nn.functional as F

from oracle import fakequant_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
    return torch.Generator().manual_seed(2333 + seed)


CASES = [  # N, C, H, W, K, R, stride, pad, dil
    (2, 64, 9, 9, 64, 1, 1, 0, 1),
    (3, 128, 8, 8, 256, 1, 2, 0, 1),      # 1x1 stride 2 (downsample), BN = 128 path
    (2, 64, 10, 10, 64, 3, 1, 1, 1),
    (2, 64, 11, 9, 128, 3, 2, 1, 1),
    (1, 128, 7, 7, 192, 3, 1, 1, 1),      # K % 128 != 0 -> BN = 64 with a tail tile
    (2, 64, 9, 9, 40, 3, 1, 2, 2),        # dilation, K < 64
    (5, 64, 14, 14, 128, 3, 1, 1, 1),     # M = 980: not a multiple of 128
]


@pytest.mark.parametrize("unsigned", [True, False])
def test_conv_i8_matches_float64_reference(unsigned):
    from dlmc.quantization.scalar import kernels as K
    for idx, (n, c, h, w, k, r, stride, pad, dil) in enumerate(CASES):
        g = gen(idx)
        lo, hi = (0, 255) if unsigned else (-127, 127)
        codes = torch.randint(lo, hi + 1, (n, c, h, w), generator=g).to(torch.uint8 if unsigned else torch.int8)
        zp = float(torch.randint(0, 9, (1,), generator=g)) if unsigned else 0.0
        s_in = torch.tensor(0.0173)
        wt = torch.randn(k, c, r, r, generator=g) * 0.05
        s_w, _ = O.minmax_channel(wt, 8, True, ch_axis=0)
        s_w = s_w + 1e-6
        bias = torch.randn(k, generator=g)
        qw_ref, wq_deq = O.fq_symmetric(wt, s_w, -127, 127)
        # weight quantisation kernel: KRSC codes + sums
        wq, wsum = K.quantize_weight_krsc(wt.to(DEV), s_w.to(DEV), -127, 127)
        assert torch.equal(wq.cpu().to(torch.float32), qw_ref.permute(0, 2, 3, 1).contiguous()), f"case {idx} wq"
        assert torch.equal(wsum.cpu().to(torch.float64), qw_ref.double().sum(dim=(1, 2, 3))), f"case {idx} wsum"
        xq = (codes.double() - zp) * s_in.double()
        ref = F.conv2d(xq, wq_deq.double(), bias.double(), stride=stride, padding=pad, dilation=dil)
        got = K.conv2d_i8(codes.to(DEV).contiguous(memory_format=torch.channels_last), wq, wsum, bias.to(DEV),
                          s_in.to(DEV), torch.tensor(zp).to(DEV), s_w.to(DEV), stride=stride, padding=pad, dilation=dil)
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-6, atol=2e-5, msg=lambda m: f"case {idx}: {m}")


def test_conv_i8_exact_integers_and_linear():
    from dlmc.quantization.scalar import kernels as K
    g = gen(50)
    # asymmetric small integers, unit scales: the fp32 output is the exact integer convolution
    codes = torch.randint(0, 16, (2, 64, 6, 6), generator=g).to(torch.uint8)
    wq_f = torch.randint(-7, 8, (96, 64, 3, 3), generator=g).float()
    wq, wsum = K.quantize_weight_krsc(wq_f.to(DEV), torch.ones(96, 1, 1, 1, device=DEV), -127, 127)
    ref = F.conv2d(codes.double() - 3.0, wq_f.double(), None, padding=1)
    got = K.conv2d_i8(codes.to(DEV).contiguous(memory_format=torch.channels_last), wq, wsum, None,
                      torch.ones(1, device=DEV), torch.tensor(3.0, device=DEV), torch.ones(96, device=DEV), padding=1)
    assert torch.equal(got.cpu().double(), ref)
    # linear: (N, C) x (K, C)^T with a K that is not a multiple of 64
    x = torch.randint(0, 256, (37, 128), generator=g).to(torch.uint8)
    wl = torch.randn(1000, 128, generator=g) * 0.03
    s_w, _ = O.minmax_channel(wl, 8, True, ch_axis=0)
    b = torch.randn(1000, generator=g)
    wq, wsum = K.quantize_weight_krsc(wl.to(DEV), s_w.to(DEV), -127, 127)
    ref = F.linear(x.double() * 0.02, O.fq_symmetric(wl, s_w, -127, 127)[1].double(), b.double())
    got = K.conv2d_i8(x.to(DEV), wq, wsum, b.to(DEV), torch.tensor(0.02, device=DEV), None, s_w.to(DEV))
    assert got.shape == (37, 1000)
    torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-6, atol=2e-5)
    # arguments the kernel cannot take are refused, not mis-computed
    from dlmc._native import DlmcqError
    with pytest.raises(DlmcqError):
        K.conv2d_i8(torch.zeros(1, 48, 4, 4, dtype=torch.uint8, device=DEV).contiguous(memory_format=torch.channels_last),
                    torch.zeros(8, 1, 1, 48, dtype=torch.int8, device=DEV), torch.zeros(8, dtype=torch.int32, device=DEV), None,
                    torch.ones(1, device=DEV), None, torch.ones(8, device=DEV))


def test_fsptq_model_int8_path_matches_fp32_path():
    """The same quantised network through both conv paths: fp32 MIOpen conv of the fake-quantised operands
    (reference-identical operands) vs the fused int8 MFMA path.  Ineligible layers (C = 3 stem) fall back."""
    import copy
    from torch import nn
    from dlmc.quantization.scalar import kernels as K
    from dlmc.utils.quantize import quantize_model
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "recon_type": "None", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    torch.manual_seed(2333)
    base = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 128, 3, stride=2, padding=1), nn.ReLU(),
                         nn.Conv2d(128, 64, 1), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(64, 10)).to(DEV).eval()
    a, b = copy.deepcopy(base), copy.deepcopy(base)
    quantize_model(a, copy.deepcopy(cfg), None, quantization_type="FSPTQ")
    quantize_model(b, copy.deepcopy(cfg), None, quantization_type="FSPTQ", int8_gemm=True)
    x = torch.randn(4, 3, 20, 20, device=DEV)
    with torch.no_grad():
        ra, rb = a(x), b(x)                      # calibration forward
        K.PROFILE.enabled = True
        K.PROFILE.reset()
        ra, rb = a(x), b(x)
        K.PROFILE.enabled = False
    tags = [t for t, *_ in K.PROFILE.records]
    assert tags.count("conv_i8") == 3           # conv 64->128, conv 128->64, linear 64->10; the C = 3 stem falls back
    torch.testing.assert_close(rb, ra, rtol=1e-4, atol=1e-4)
    # with autograd on, the int8 path steps aside (training uses the differentiable fp32 path)
    out = b(x)
    assert out.requires_grad


def test_qbase_model_int8_path_matches_fp32_path():
    """QBase family (signed symmetric W8A8, per tensor): int8 MFMA path vs the fp32 path."""
    import copy
    from torch import nn
    from dlmc.quantization.scalar import kernels as K
    from dlmc.utils.quantize import quantize_model
    cfg = {"weight": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
           "exclude_layers": [], "override_options": []}
    torch.manual_seed(2333)
    base = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 128, 1, stride=2), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(128, 10)).to(DEV).eval()
    a, b = copy.deepcopy(base), copy.deepcopy(base)
    quantize_model(a, copy.deepcopy(cfg), None)
    quantize_model(b, copy.deepcopy(cfg), None, int8_gemm=True)
    x = torch.randn(4, 64, 12, 12, device=DEV)
    with torch.no_grad():
        a(x), b(x)
        K.PROFILE.enabled = True
        K.PROFILE.reset()
        ra, rb = a(x), b(x)
        K.PROFILE.enabled = False
    assert [t for t, *_ in K.PROFILE.records].count("conv_i8") == 3
    torch.testing.assert_close(rb, ra, rtol=1e-4, atol=1e-4)
    # an asymmetric (unsigned, offset = min != 0) QBase layer is not int8-able and silently keeps the fp32 path
    cfg2 = copy.deepcopy(cfg)
    cfg2["input"]["args"]["signed"] = False
    c = copy.deepcopy(base)
    quantize_model(c, cfg2, None, int8_gemm=True)
    with torch.no_grad():
        c(x)
        K.PROFILE.enabled = True
        K.PROFILE.reset()
        c(x)
        K.PROFILE.enabled = False
    assert [t for t, *_ in K.PROFILE.records].count("conv_i8") <= 2   # only the post-ReLU (min = 0) layers qualify


def test_module_path_weight_code_cache_follows_the_weights():
    """The int8 module path keeps the integer weight codes between forwards and must notice every write to the weight
    or its scale (in-place update, load_state_dict, re-calibration)."""
    from torch import nn
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}

    def build(int8):
        torch.manual_seed(1)
        net = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1)).to(DEV).eval()
        quantize_model(net, cfg, None, "FSPTQ", int8_gemm=int8)
        return net
    a, b = build(True), build(False)
    x = torch.relu(torch.randn(2, 64, 8, 8, device=DEV))
    with torch.no_grad():
        for step in range(3):
            ya, yb = a(x), b(x)
            torch.testing.assert_close(ya, yb.contiguous(memory_format=torch.channels_last), rtol=1e-5, atol=1e-5)
            assert getattr(a[0], "_wq_cache", None) is not None
            for net in (a, b):                       # an "optimiser step": in-place, bumps the version counter
                net[0].weight.mul_(1.5 + step)
                net[0].wt_scale.mul_(1.5 + step)
        sd = build(False).state_dict()
        a.load_state_dict(sd)
        b.load_state_dict(sd)
        torch.testing.assert_close(a(x), b(x).contiguous(memory_format=torch.channels_last), rtol=1e-5, atol=1e-5)


def test_conv_i8_random_shapes_against_float64():
    """120 random small geometries (channels, sizes, taps, stride, padding, dilation, ragged tiles) through the default
    dispatch and the fused epilogue, against a float64 convolution of the dequantised operands."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = gen(77)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))
    done = 0
    while done < 120:
        n, c, k = ri(1, 3), 64 * ri(1, 3), [8, 32, 64, 72, 128, 192, 256][ri(0, 6)]
        r = [1, 3, 5][ri(0, 2)]
        stride, dil = ri(1, 2), ri(1, 2)
        pad = ri(0, 2)
        h, w = ri(1, 12), ri(1, 12)
        p, q = (h + 2 * pad - dil * (r - 1) - 1) // stride + 1, (w + 2 * pad - dil * (r - 1) - 1) // stride + 1
        if p < 1 or q < 1:
            continue
        done += 1
        unsigned = done % 2 == 0
        lo, hi = (0, 255) if unsigned else (-127, 127)
        codes = torch.randint(lo, hi + 1, (n, c, h, w), generator=g).to(torch.uint8 if unsigned else torch.int8)
        zp = float(ri(0, 9)) if unsigned else 0.0
        wt = torch.randn(k, c, r, r, generator=g) * 0.05
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        bias = torch.randn(k, generator=g)
        qw = torch.clamp(torch.round(wt / s_w.reshape(-1, 1, 1, 1)), -127, 127)
        ref = F.conv2d((codes.double() - zp) * 0.0173, qw.double() * s_w.double().reshape(-1, 1, 1, 1), bias.double(),
                       stride=stride, padding=pad, dilation=dil)
        wq, wsum = K.quantize_weight_krsc(wt.to(DEV), s_w.to(DEV), -127, 127)
        cd = codes.to(DEV).contiguous(memory_format=torch.channels_last)
        args = (cd, wq, wsum, bias.to(DEV), torch.tensor([0.0173], device=DEV), torch.tensor([zp], device=DEV), s_w.to(DEV))
        kw = dict(stride=stride, padding=pad, dilation=dil)
        tag = f"shape {done}: n{n} c{c} {h}x{w} k{k} r{r} s{stride} p{pad} d{dil} {'u8' if unsigned else 's8'}"
        got = K.conv2d_i8(*args, **kw)
        torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-6, atol=2e-5, msg=lambda m: f"{tag}: {m}")
        res = torch.randn(got.shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        q_s = torch.tensor([float(ref.abs().max()) / 255 + 1e-3], device=DEV)
        emit = K.EmitCodes(q_s, torch.tensor([0.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
        want = torch.relu(got + res)
        _, wc = K.fake_quant(want, q_s, emit.zero_point, 0, 255, N.FORM_ZEROPOINT, codes="i8", want_y=False)
        out, cds = K.conv2d_i8(*args, residual=res, relu=True, emit=emit, **kw)
        assert torch.equal(out, want) and torch.equal(cds, wc), tag + " fused epilogue"


def test_codes_only_epilogue_random_shapes_match_the_two_pass_path():
    """Layers that emit only their consumer's codes run in the swapped accumulator layout (conv_i8_mfma_kernel<..., SWAP>: a lane
    owns 16 channels of one pixel, the ReLU is folded into the quantiser's clamp).  4 fixed (the 256-wide tile rule) + 80 random geometries - widths that do and do
    not qualify (K % tile width), ragged row tiles, all four quantiser forms incl. signed ranges and non-zero offsets, ReLU on and
    off, bias on and off, symmetric and asymmetric weights - against the same convolution's fp32 output put through ReLU and the
    stand-alone fake-quant kernel: the codes must be the same bytes."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = gen(1234)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))
    forms = [N.FORM_ZEROPOINT, N.FORM_SYMMETRIC, N.FORM_EMULATE, N.FORM_QBASE]
    # the shapes whose tile rule is the 256-wide swapped kernel (A direct / A through the ring) come first, then random ones
    fixed = [(2, 512, 512, 3), (3, 2048, 512, 1), (1, 512, 256, 3), (2, 1024, 256, 1)]
    done = 0
    while done < 84:
        n, c, k = ri(1, 3), 64 * ri(1, 3), [64, 128, 192, 256, 72, 32][ri(0, 5)]
        r = [1, 3][ri(0, 1)]
        if done < len(fixed):
            n, c, k, r = fixed[done]
        stride, pad = ri(1, 2), ri(0, 1)
        h, w = ri(2, 13), ri(2, 13)
        p, q = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
        if p < 1 or q < 1:
            continue
        done += 1
        unsigned = done % 2 == 0
        lo, hi = (0, 255) if unsigned else (-127, 127)
        codes = torch.randint(lo, hi + 1, (n, c, h, w), generator=g).to(torch.uint8 if unsigned else torch.int8)
        zp = float(ri(0, 9)) if unsigned else 0.0
        wt = torch.randn(k, c, r, r, generator=g) * 0.05
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        bias = torch.randn(k, generator=g).to(DEV) if done % 3 else None
        w_off = (torch.randn(k, generator=g) * 0.01).to(DEV) if done % 4 == 1 else None
        wq, wsum = K.quantize_weight_krsc(wt.to(DEV), s_w.to(DEV), -127, 127)
        cd = codes.to(DEV).contiguous(memory_format=torch.channels_last)
        args = (cd, wq, wsum, bias, torch.tensor([0.0173], device=DEV), torch.tensor([zp], device=DEV), s_w.to(DEV))
        kw = dict(stride=stride, padding=pad, w_offset=w_off)
        relu = done % 5 != 0
        form = forms[done % 4]
        ref = K.conv2d_i8(*args, **kw)
        want = torch.relu(ref) if relu else ref
        spread = float(want.abs().max()) + 1e-3
        if form == N.FORM_ZEROPOINT:
            qlo, qhi, q_s, q_z = 0, 255, spread / 200, float(ri(0, 40))
        elif form == N.FORM_SYMMETRIC:
            qlo, qhi, q_s, q_z = -127, 127, spread / 100, 0.0
        elif form == N.FORM_EMULATE:
            qlo, qhi, q_s, q_z = (0, 255, spread / 300, -0.37 * spread) if done % 8 < 4 else (-128, 127, spread / 90, 0.11 * spread)
        else:
            qlo, qhi, q_s, q_z = -127, 127, spread / 110, 0.05 * spread
        gq = 1.0 / (want.numel() * 127) ** 0.5 if form == N.FORM_QBASE else 0.0
        q_st, q_zt = torch.tensor([q_s], device=DEV), torch.tensor([q_z], device=DEV)
        emit = K.EmitCodes(q_st, q_zt, qlo, qhi, form, gq)
        _, wc = K.fake_quant(want, q_st, q_zt, qlo, qhi, form, g=gq, codes="i8", want_y=False)
        tag = f"case {done}: n{n} c{c} {h}x{w} k{k} r{r} s{stride} p{pad} form {form} relu {relu} asym {w_off is not None}"
        none, cds = K.conv2d_i8(*args, relu=relu, emit=emit, want_out=False, **kw)
        assert none is None
        assert torch.equal(cds.view(torch.uint8), wc.view(torch.uint8)), tag
