"""The fused epilogue of the int8 kernel (residual add + ReLU + the consumer's activation codes) and the frozen
execution plan built on it must reproduce the separate kernels BIT FOR BIT: same arithmetic, fewer trips
through HBM."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
    return torch.Generator().manual_seed(2333 + seed)


def same(a, b, what):
    """Bit equality, treating +0 and -0 as one value (ReLU may pick either)."""
    assert a.shape == b.shape and a.dtype == b.dtype, what
    if a.dtype == torch.float32:
        a, b = a + 0.0, b + 0.0
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), \
            f"{what}: {(a.view(torch.int32) != b.view(torch.int32)).sum().item()} of {a.numel()} differ"
    else:
        assert torch.equal(a, b), f"{what}: {(a != b).sum().item()} of {a.numel()} differ"


EPI_CASES = [  # N, C, H, W, K, R, stride, pad
    (2, 64, 9, 9, 64, 1, 1, 0),
    (3, 128, 8, 8, 256, 1, 2, 0),
    (2, 64, 10, 10, 128, 3, 1, 1),
    (1, 128, 7, 7, 192, 3, 1, 1),      # BN = 64 with a tail tile
    (5, 64, 14, 14, 128, 3, 1, 1),     # ragged M
    (2, 64, 9, 9, 42, 3, 1, 1),        # K % 4 != 0: scalar epilogue
]


@pytest.mark.parametrize("form", ["zeropoint_u8", "zeropoint_s8", "qbase_s8", "emulate_u8", "symmetric_s4"])
def test_fused_epilogue_equals_the_separate_kernels(form):
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    qform, lo, hi, zp, g = {"zeropoint_u8": (N.FORM_ZEROPOINT, 0, 255, 3.0, 0.0),
                            "zeropoint_s8": (N.FORM_ZEROPOINT, -127, 127, -5.0, 0.0),
                            "qbase_s8": (N.FORM_QBASE, -127, 127, None, 1e-3),
                            "emulate_u8": (N.FORM_EMULATE, 0, 255, 0.25, 0.0),
                            "symmetric_s4": (N.FORM_SYMMETRIC, -7, 7, None, 0.0)}[form]
    for idx, (n, c, h, w, k, r, stride, pad) in enumerate(EPI_CASES):
        gg = gen(idx)
        codes = torch.randint(0, 256, (n, c, h, w), generator=gg).to(torch.uint8).to(DEV).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(k, c, r, r, generator=gg) * 0.05).to(DEV)
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        bias = torch.randn(k, generator=gg).to(DEV)
        wq, wsum = K.quantize_weight_krsc(wt, s_w, -127, 127)
        s_in, zp_in = torch.tensor([0.0173], device=DEV), torch.tensor([2.0], device=DEV)
        kw = dict(stride=stride, padding=pad)
        plain = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, s_w, **kw)
        res = torch.randn(plain.shape, generator=gg).to(DEV).contiguous(memory_format=torch.channels_last)
        q_s = torch.tensor([float(plain.abs().max()) / max(hi, 1) * 0.7], device=DEV)   # some values saturate
        q_zp = None if zp is None else torch.tensor([zp], device=DEV)
        emit = K.EmitCodes(q_s, q_zp, lo, hi, qform, g)
        for use_res in (False, True):
            for relu in (False, True):
                want = plain + res if use_res else plain.clone()
                if relu:
                    want = torch.relu(want)
                _, want_codes = K.fake_quant(want, q_s, q_zp, lo, hi, qform, g=g, codes="i8", want_y=False)
                tag = f"{form} case {idx} res={use_res} relu={relu}"
                out, got_codes = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, s_w, residual=res if use_res else None,
                                             relu=relu, emit=emit, **kw)
                same(out, want, tag + " out")
                same(got_codes, want_codes, tag + " codes")
                assert got_codes.is_contiguous(memory_format=torch.channels_last)
                none, only_codes = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, s_w, residual=res if use_res else None,
                                               relu=relu, emit=emit, want_out=False, **kw)
                assert none is None
                same(only_codes, want_codes, tag + " codes-only")
                if use_res or relu:
                    same(K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, s_w, residual=res if use_res else None, relu=relu, **kw),
                         want, tag + " no-emit")


def test_dual_kernel_equals_two_convolutions_and_an_add():
    """conv_a(x) + conv_b(y) (the block's last conv + the conv on its shortcut) in one kernel, every epilogue."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    cases = [  # (N, Ca, Ha, Ka, Ra, stride_a, pad_a), (Cb, Hb, Rb, stride_b, pad_b)
        ((2, 64, 14, 256, 1, 1, 0), (128, 28, 1, 2, 0)),     # Bottleneck: 1x1 + 1x1/2 downsample
        ((3, 128, 9, 128, 3, 1, 1), (64, 18, 1, 2, 0)),      # BasicBlock: 3x3 + 1x1/2 downsample, ragged M
        ((2, 64, 12, 64, 1, 1, 0), (64, 12, 3, 1, 1)),       # BN = 64 tiles, 3x3 on the shortcut
        ((1, 192, 7, 192, 1, 1, 0), (64, 7, 1, 1, 0)),       # K % 128 != 0
    ]
    for idx, ((n, ca, ha, k, ra, sa, pa), (cb, hb, rb, sb, pb)) in enumerate(cases):
        gg = gen(400 + idx)

        def operand(c, h, r, stride, pad, unsigned, seed_zp):
            lo, hi = (0, 256) if unsigned else (-127, 128)
            codes = torch.randint(lo, hi, (n, c, h, h), generator=gg).to(torch.uint8 if unsigned else torch.int8).to(DEV)
            codes = codes.contiguous(memory_format=torch.channels_last)
            wt = (torch.randn(k, c, r, r, generator=gg) * 0.05).to(DEV)
            s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
            wq, wsum = K.quantize_weight_krsc(wt, s_w, -127, 127)
            return dict(codes=codes, wq=wq, wsum=wsum, bias=torch.randn(k, generator=gg).to(DEV),
                        in_scale=torch.tensor([0.0173 + 0.001 * seed_zp], device=DEV),
                        in_zp=torch.tensor([float(seed_zp) if unsigned else 0.0], device=DEV), w_scale=s_w, stride=stride, padding=pad)
        a = operand(ca, ha, ra, sa, pa, True, 2)
        b = operand(cb, hb, rb, sb, pb, idx % 2 == 0, 5)

        def single(t):
            return K.conv2d_i8(t["codes"], t["wq"], t["wsum"], t["bias"], t["in_scale"], t["in_zp"], t["w_scale"],
                               stride=t["stride"], padding=t["padding"])
        plain = single(a) + single(b)
        same(K.conv2d_i8_dual(a, b), plain, f"dual case {idx} plain")
        q_s = torch.tensor([float(plain.abs().max()) / 255 * 0.7], device=DEV)
        emit = K.EmitCodes(q_s, torch.tensor([1.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
        want = torch.relu(plain)
        _, wc = K.fake_quant(want, q_s, emit.zero_point, 0, 255, N.FORM_ZEROPOINT, codes="i8", want_y=False)
        out, codes = K.conv2d_i8_dual(a, b, relu=True, emit=emit)
        same(out, want, f"dual case {idx} relu out")
        same(codes, wc, f"dual case {idx} codes")
        none, codes2 = K.conv2d_i8_dual(a, b, relu=True, emit=emit, want_out=False)
        assert none is None
        same(codes2, wc, f"dual case {idx} codes-only")
    with pytest.raises(ValueError):     # different output shapes
        K.conv2d_i8_dual(a, dict(b, stride=2))


def test_fused_entry_rejects_bad_arguments():
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    codes = torch.zeros(1, 64, 4, 4, dtype=torch.uint8, device=DEV).contiguous(memory_format=torch.channels_last)
    wq, wsum = K.quantize_weight_krsc(torch.randn(64, 64, 1, 1, device=DEV), torch.ones(64, device=DEV), -127, 127)
    one = torch.ones(1, device=DEV)
    with pytest.raises(ValueError):
        K.conv2d_i8(codes, wq, wsum, None, one, None, one, want_out=False)
    with pytest.raises(ValueError):
        K.conv2d_i8(codes, wq, wsum, None, one, None, one, residual=torch.zeros(1, 64, 5, 5, device=DEV))
    with pytest.raises(N.DlmcqError):   # a range wider than a byte
        K.conv2d_i8(codes, wq, wsum, None, one, None, one, emit=K.EmitCodes(one, None, -200, 200, N.FORM_SYMMETRIC))
    with pytest.raises(N.DlmcqError):   # RootQ activations are not an epilogue form
        K.conv2d_i8(codes, wq, wsum, None, one, None, one, emit=K.EmitCodes(one, None, 0, 255, N.FORM_ROOTQ_ACT))


FSPTQ = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
         "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
         "exclude_layers": [], "override_options": []}
QBASE = {"weight": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
         "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": True}},
         "exclude_layers": [], "override_options": []}


@pytest.mark.parametrize("arch,qtype,cfg,res,signed_image", [
    ("resnet50", "FSPTQ", FSPTQ, 64, False), ("resnet18", "FSPTQ", FSPTQ, 96, True), ("repvgg_a1", "FSPTQ", FSPTQ, 64, False),
    ("resnet18", None, QBASE, 64, True), ("resnet18-bn", "FSPTQ", FSPTQ, 64, False)])
def test_fused_plan_is_bit_identical_to_the_wrappers(arch, qtype, cfg, res, signed_image):
    """`signed_image`: N(0,1) pixels.  Under the FSPTQ u8 activation config the reference's zero point is then the
    (negative, non-integer) minimum (FSPTQuant/base.py:99-103 via ops.py:20-34), the first layer's codes are not
    integers and it keeps its fp32 path; with non-negative pixels (min = 0) it runs on the stem kernel."""
    import workloads as W
    from dlmc.utils.fuse import fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    keep_bn = arch.endswith("-bn")                # BatchNorm left in place: nothing may be folded across it
    arch = arch.split("-")[0]
    net = W.MODELS[arch]().to(DEV).eval()
    for m in net.modules():                       # non-trivial BN statistics, then fold them as FSPTQuant.py:67 does
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    if not keep_bn:
        net = merge_bn(net, inplace=True)
    quantize_model(net, cfg, None, qtype, int8_gemm=True)
    x = torch.randn(4, 3, res, res, device=DEV)
    if not signed_image:
        x = torch.relu(x)
    with torch.no_grad():
        net(x)                                    # calibrate
        want = net(x * 0.8)
    fused = fuse_inference(net)
    rep = fused.fusion_report
    print(arch, qtype, rep)
    with torch.no_grad():
        got = fused(x * 0.8)
    same(got, want, f"{arch} {qtype} logits")
    if arch == "resnet50":
        assert rep.layers == 54 and rep.residual == 16 and rep.relu == 49 and rep.skipped == []
        assert rep.stem == 1 and rep.pooled == 1           # conv1 -> ReLU -> max-pool, pooled as codes
        assert rep.emit == 48 and rep.fp32_outputs == 14   # 12 shortcuts + last block + fc
        assert rep.dual == 4                               # conv3 + downsample of each stage's first block: one kernel
    if arch == "repvgg_a1":
        assert rep.layers == 23 and rep.relu == 22 and rep.stem == 1 and rep.skipped == []
        assert rep.fp32_outputs == 2    # the last block (feeds the pool) and the classifier
    if keep_bn:
        assert rep.relu == 0 and rep.residual == 0 and rep.emit == 0 and rep.layers == 21
    elif arch == "resnet18" and qtype == "FSPTQ":
        assert rep.stem == 0 and rep.skipped == ["conv1"]       # non-integer zero point: fp32 first layer
    if arch == "resnet18" and qtype is None:
        assert rep.stem == 1 and rep.pooled == 1                # QBase: signed codes; the stem kernel pools in fp32, shortcut included
    # and under a HIP graph
    from dlmc.utils.graph import GraphedForward
    fwd = GraphedForward(fused, x)
    same(fwd(x * 0.8), want, f"{arch} {qtype} graphed")


@pytest.mark.parametrize("ch", [24, 96, 160, 128])
def test_residual_block_whose_width_is_no_multiple_of_64(ch):
    """MobileNetV2-style projection widths (24 / 96 / 160) and the fp32 `out[:, :k]` of such a layer used as a later shortcut:
    the plan pads those layers' output channels to a multiple of 64, so their shortcut adds stay outside the kernel
    (a k-wide fp32 shortcut does not fit a k_pad-wide tile); 128 channels take the absorbed path.  Either way the plan is
    bit-identical to the wrappers."""
    from dlmc.utils.fuse import fuse_inference
    from dlmc.utils.quantize import quantize_model

    class Narrow(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = torch.nn.Conv2d(64, ch, 1)
            self.a = torch.nn.Conv2d(ch, 128, 1)
            self.b = torch.nn.Conv2d(128, ch, 3, padding=1)
            self.c = torch.nn.Conv2d(ch, 64, 1)
            self.d = torch.nn.Conv2d(64, ch, 1)
            self.head = torch.nn.Conv2d(ch, 32, 1)

        def forward(self, x):
            y = torch.relu(self.stem(x))
            z = torch.relu(self.b(torch.relu(self.a(y))) + y)      # block 1: shortcut = a padded layer's fp32 output
            z = torch.relu(self.d(torch.relu(self.c(z))) + z)      # block 2: shortcut = the first block's result
            return self.head(z)
    torch.manual_seed(7 + ch)
    net = Narrow().to(DEV).eval()
    quantize_model(net, FSPTQ, None, "FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(3, 64, 12, 12, device=DEV))
    with torch.no_grad():
        net(x)
        want = net(x * 0.7)
        fused = fuse_inference(net)
        got = fused(x * 0.7)
    rep = fused.fusion_report
    assert rep.skipped == [] and rep.layers == 6 and rep.residual == (2 if ch % 64 == 0 else 0), rep
    if ch % 64 == 0:
        same(got, want, f"residual block, {ch} channels")
    else:
        # the wrappers run layers whose channel count is no multiple of 64 as fp32 convolutions of the fake-quantised operands
        # (no padding at module level); the plan pads them onto the int8 kernel: same codes, fp32 accumulation order differs
        # - and a value that sits on a rounding tie of the next quantiser may then land one code apart (the +-1-code effect of
        # tests/test_gpu_fullsize.py): nearly all elements agree to fp32 accuracy, the rest by about one quantisation step
        diff = (got - want).abs()
        assert float((diff > 2e-5).float().mean()) < 0.03 and float(diff.max()) < 0.05 * float(want.std()), \
            (float((diff > 2e-5).float().mean()), float(diff.max()), float(want.std()))


STEM_CASES = [  # N, C, H, W, K, R, S, stride, pad
    (2, 3, 32, 32, 64, 7, 7, 2, 3),      # ResNet
    (3, 3, 17, 23, 64, 3, 3, 2, 1),      # RepVGG / MobileOne stage0, odd sizes
    (2, 3, 16, 16, 96, 3, 3, 1, 1),      # two 64-channel slabs, the second half empty
    (1, 4, 12, 12, 32, 5, 5, 1, 2),
    (2, 1, 9, 9, 8, 1, 1, 1, 0),
    (9, 3, 20, 20, 64, 7, 8, 2, 3),      # 8 taps per row; M = 9*10*9 = 810 (ragged tile)
    (2, 4, 8, 12, 32, 3, 3, 1, 1),       # four channels through the four-pixels-per-thread image quantiser (NCHW, W % 4 = 0)
    (2, 1, 8, 8, 8, 1, 1, 1, 0),         # one channel through it
    (2, 2, 6, 16, 16, 3, 3, 1, 2),       # two channels, pad 2
    (3, 3, 10, 24, 64, 7, 7, 2, 3),      # channels_last image (odd index), three channels, W % 4 = 0: the one-pixel-per-thread quantiser
]


@pytest.mark.parametrize("unsigned", [True, False])
def test_stem_kernels_match_the_generic_path(unsigned):
    """quantize_pad_nhwc4 + conv2d_i8_stem against fake_quant codes and a float64 convolution of the dequantised
    operands; every epilogue option against the separate kernels, bit for bit."""
    import torch.nn.functional as F
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    lo, hi = (0, 255) if unsigned else (-127, 127)
    for idx, (n, c, h, w, k, r, s, stride, pad) in enumerate(STEM_CASES):
        gg = gen(100 + idx)
        x = torch.randn(n, c, h, w, generator=gg).to(DEV)
        if idx % 2:
            x = x.contiguous(memory_format=torch.channels_last)
        s_in = torch.tensor([float(x.abs().max()) / (127 if not unsigned else 200)], device=DEV)
        zp = torch.tensor([117.0 if unsigned else 0.0], device=DEV)
        form = N.FORM_ZEROPOINT
        if idx == 0:     # non-finite pixels take the reference's route (R(+-inf) = NaN -> code 0)
            x.view(-1)[torch.tensor([3, 77, 500])] = torch.tensor([float("inf"), -float("inf"), float("nan")], device=DEV)
        _, want_codes = K.fake_quant(x, s_in, zp, lo, hi, form, codes="i8", want_y=False)
        xpad = K.quantize_pad_nhwc4(x, s_in, zp, lo, hi, form, pad)
        assert xpad.shape == (n, h + 2 * pad, w + 2 * pad, 4)
        inner = xpad[:, pad:pad + h, pad:pad + w, :c].permute(0, 3, 1, 2)
        same(inner.contiguous(), want_codes.contiguous(), f"stem case {idx} image codes")
        if pad:
            border = torch.cat([xpad[:, :pad, :, :c].reshape(-1), xpad[:, :, :pad, :c].reshape(-1), xpad[:, -pad:, :, :c].reshape(-1),
                                xpad[:, :, -pad:, :c].reshape(-1)])
            assert bool((border.to(torch.float32) == float(zp)).all()), f"stem case {idx} border"
        wt = (torch.randn(k, c, r, s, generator=gg) * 0.1).to(DEV)
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        bias = torch.randn(k, generator=gg).to(DEV)
        wq, wsum = K.quantize_weight_stem(wt, s_w, -127, 127)
        qw = torch.clamp(torch.round(wt.cpu() / s_w.cpu().reshape(-1, 1, 1, 1)), -127, 127)
        assert torch.equal(wq[:, :, :s, :c].cpu().to(torch.float32), qw.permute(0, 2, 3, 1)), f"stem case {idx} wq"
        assert int(wq[:, :, s:, :].abs().sum()) == 0 and int(wq[:, :, :, c:].abs().sum()) == 0
        assert torch.equal(wsum.cpu().double(), qw.double().sum(dim=(1, 2, 3)))
        plain = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride)
        ref = F.conv2d((want_codes.cpu().double() - float(zp)) * float(s_in), qw.double() * s_w.cpu().double().reshape(-1, 1, 1, 1),
                       bias.cpu().double(), stride=stride, padding=pad)
        assert plain.shape == ref.shape and plain.is_contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(plain.cpu().double(), ref, rtol=2e-6, atol=2e-5, msg=lambda m: f"stem case {idx}: {m}")
        q_s = torch.tensor([float(plain.abs().max()) / 255 * 0.8], device=DEV)
        emit = K.EmitCodes(q_s, torch.tensor([0.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
        want = torch.relu(plain)
        _, wc = K.fake_quant(want, q_s, emit.zero_point, 0, 255, N.FORM_ZEROPOINT, codes="i8", want_y=False)
        out, codes = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride, relu=True, emit=emit)
        same(out, want, f"stem case {idx} relu out")
        same(codes, wc, f"stem case {idx} codes")
        none, codes2 = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride, relu=True, emit=emit, want_out=False)
        assert none is None
        same(codes2, wc, f"stem case {idx} codes-only")


def test_stem_with_the_pool_inside():
    """conv -> ReLU -> MaxPool2d(3, 2, 1) -> quantiser as one kernel against the separate ops (bit for bit; NaN wins)."""
    import torch.nn.functional as F
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    for idx, (n, c, h, w, k, r, s, stride, pad) in enumerate([(2, 3, 32, 32, 64, 7, 7, 2, 3), (3, 3, 17, 23, 64, 3, 3, 2, 1),
                                                               (1, 4, 13, 9, 32, 5, 5, 1, 2), (2, 1, 7, 50, 8, 1, 3, 1, 0),
                                                               (5, 3, 9, 9, 64, 3, 3, 1, 1)]):
        gg = gen(600 + idx)
        x = torch.randn(n, c, h, w, generator=gg).to(DEV)
        if idx == 1:
            x.view(-1)[123] = float("nan")
        s_in = torch.tensor([float(x[torch.isfinite(x)].abs().max()) / 120], device=DEV)
        zp = torch.tensor([128.0], device=DEV)
        wt = (torch.randn(k, c, r, s, generator=gg) * 0.1).to(DEV)
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        if idx == 4:
            s_w[5] = -s_w[5]          # a negative scale (no observer produces one): the pool's order flips for that channel
            s_w[33] = -s_w[33]
        bias = torch.randn(k, generator=gg).to(DEV)
        xpad = K.quantize_pad_nhwc4(x, s_in, zp, 0, 255, N.FORM_ZEROPOINT, pad)
        wq, wsum = K.quantize_weight_stem(wt, s_w, -127, 127)
        for relu in (True, False):
            conv = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride, relu=relu)
            want = F.max_pool2d(conv, 3, 2, 1)
            q_s = torch.tensor([float(want[torch.isfinite(want)].abs().max()) / 255 * 0.9 + 1e-3], device=DEV)
            emit = K.EmitCodes(q_s, torch.tensor([7.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
            _, wc = K.fake_quant(want, q_s, emit.zero_point, 0, 255, N.FORM_ZEROPOINT, codes="i8", want_y=False)
            out, codes = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride, relu=relu, emit=emit, pool=True)
            assert out.shape == want.shape and out.is_contiguous(memory_format=torch.channels_last)
            nan = torch.isnan(want)
            assert torch.equal(torch.isnan(out), nan), f"pooled stem {idx}: NaN pattern"
            same(torch.where(nan, torch.zeros_like(out), out), torch.where(nan, torch.zeros_like(want), want), f"pooled stem {idx} relu={relu} out")
            same(codes, wc, f"pooled stem {idx} relu={relu} codes")
            none, c2 = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride, relu=relu, emit=emit, want_out=False, pool=True)
            assert none is None
            same(c2, wc, f"pooled stem {idx} codes-only")


def test_resnet_first_layer_with_the_pool_in_registers():
    """csrc/conv_stem_pool7_i8.hip (7 filter rows, stride 2, 64 channels, codes only: pooling by DPP and running maxima, never
    through LDS) against conv_stem_pool_i8_kernel (the same entry point asked for the fp32 output as well stays on the old
    kernel): the same bytes.  Sizes: 224 (two x tiles of 31 + 25 pooled columns, four bands of 14 pooled rows), 32 (one 8-column
    tile: a single store instruction), 260 (three x tiles), 8 tap columns, images that do not fill the last workgroup, signed
    codes, a channel with a negative scale (the pool's order flips), no bias, no ReLU."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    cases = [(3, 224, 224, 7, True, False, True), (5, 32, 32, 7, True, True, False), (2, 64, 260, 7, True, False, True),
             (1, 48, 40, 8, False, False, True), (7, 36, 28, 7, True, True, True)]
    for idx, (n, h, w, s, unsigned, negscale, relu) in enumerate(cases):
        gg = gen(900 + idx)
        x = torch.randn(n, 3, h, w, generator=gg).to(DEV)
        s_in = torch.tensor([float(x.abs().max()) / 120], device=DEV)
        zp = torch.tensor([128.0 if unsigned else 0.0], device=DEV)
        lo, hi = (0, 255) if unsigned else (-127, 127)
        wt = (torch.randn(64, 3, 7, s, generator=gg) * 0.1).to(DEV)
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        if negscale:
            s_w[5] = -s_w[5]
            s_w[49] = -s_w[49]
        bias = None if idx == 3 else torch.randn(64, generator=gg).to(DEV)
        xpad = K.quantize_pad_nhwc4(x, s_in, zp, lo, hi, N.FORM_ZEROPOINT, 3)
        wq, wsum = K.quantize_weight_stem(wt, s_w, -127, 127)
        ref = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=2, relu=relu, pool=True)
        q_s = torch.tensor([float(ref.abs().max()) / 255 * 0.9 + 1e-3], device=DEV)
        emit = K.EmitCodes(q_s, torch.tensor([7.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
        out, old = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=2, relu=relu, emit=emit, pool=True)        # old kernel
        none, new = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=2, relu=relu, emit=emit, pool=True, want_out=False)
        assert none is None and new.shape == old.shape
        same(new, old, f"pool-in-registers case {idx}")


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int8])
def test_maxpool_on_codes_equals_quantised_maxpool(dtype):
    import torch.nn.functional as F
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    for idx, (n, c, h, w, k, s, p) in enumerate([(2, 64, 16, 16, 3, 2, 1), (3, 8, 9, 11, 2, 2, 0), (1, 128, 7, 7, 3, 1, 1),
                                                 (2, 4, 5, 6, 3, 3, 1)]):
        gg = gen(200 + idx)
        lo, hi = (0, 256) if dtype == torch.uint8 else (-128, 128)
        codes = torch.randint(lo, hi, (n, c, h, w), generator=gg).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
        got = K.maxpool_codes(codes, k, s, p)
        want = F.max_pool2d(codes.float(), k, s, p).to(dtype)
        same(got, want.contiguous(memory_format=torch.channels_last), f"pool case {idx}")
    # and the commutation itself: quantise(maxpool(v)) == maxpool(quantise(v))
    v = torch.randn(2, 64, 14, 14, generator=gen(300)).to(DEV).contiguous(memory_format=torch.channels_last)
    s_q, zp = torch.tensor([0.011], device=DEV), torch.tensor([3.0 if dtype == torch.uint8 else -2.0], device=DEV)
    lo, hi = (0, 255) if dtype == torch.uint8 else (-127, 127)
    _, a = K.fake_quant(F.max_pool2d(v, 3, 2, 1), s_q, zp, lo, hi, N.FORM_ZEROPOINT, codes="i8", want_y=False)
    _, b = K.fake_quant(v, s_q, zp, lo, hi, N.FORM_ZEROPOINT, codes="i8", want_y=False)
    same(K.maxpool_codes(b, 3, 2, 1), a, "commutation")


def test_streamed_plan_is_bit_identical_and_refuses_qbase():
    import workloads as W
    from dlmc.utils.fuse import StreamedPlan, fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    net = merge_bn(W.resnet18().to(DEV).eval(), inplace=True)
    quantize_model(net, FSPTQ, None, "FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(6, 3, 64, 64, device=DEV))
    with torch.no_grad():
        net(x)
        plan = fuse_inference(net)
        want = plan(x * 0.7)
        for n in (1, 2, 3, 8):          # 8 > batch: falls back to one stream
            same(StreamedPlan(plan, n)(x * 0.7), want, f"{n} streams")
    q = W.resnet18().to(DEV).eval()
    quantize_model(q, QBASE, None, None, int8_gemm=True)
    with torch.no_grad():
        q(x)
    with pytest.raises(ValueError):
        StreamedPlan(fuse_inference(q), 2)


def test_fuse_requires_a_calibrated_eval_model():
    import workloads as W
    from dlmc.utils.fuse import fuse_inference
    from dlmc.utils.quantize import quantize_model
    net = W.resnet18().to(DEV)
    quantize_model(net, FSPTQ, None, "FSPTQ")
    with pytest.raises(RuntimeError):
        fuse_inference(net.train())
    with pytest.raises(RuntimeError):
        fuse_inference(net.eval())               # never calibrated


def test_fused_plan_matches_the_cpu_port_end_to_end():
    """The whole fused ResNet-50 (stem kernel, dual kernels, pooled codes, epilogue quantisers) against the CPU port of
    the reference's op sequence (oracle/ref_layers.py: FSPTQ forms + F.conv2d in fp32).

    Layer by layer the operands are bit-exact (tests above and test_gpu_modules.py); end to end they cannot be: an fp32
    convolution summed in another order differs by ~1e-6, which moves an activation that sits within 1e-6 of a rounding
    tie to the neighbouring code - about 4 in 10 000 per layer - and a deep random-weight network amplifies those
    (measured: 1.5 % of the activation spread after the first block, 6 % after the third; the fp32-conv GPU path drifts
    from the CPU by the same amounts).  So this is a sanity bound on the drift, not a parity test: logits correlate
    > 0.995, mean |diff| < 3 % of their spread, same top class."""
    import copy
    import workloads as W
    from dlmc.utils.fuse import fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    from oracle.ref_layers import port_model
    torch.manual_seed(2333)
    base = W.resnet50().eval()
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = merge_bn(copy.deepcopy(base).to(DEV), inplace=True)
    cpu = port_model(copy.deepcopy(net).cpu(), "FSPTQ")
    quantize_model(net, FSPTQ, None, "FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(4, 3, 64, 64, generator=gen(500)))
    with torch.no_grad():
        net(x.to(DEV))
        plan = fuse_inference(net)
        got = plan(x.to(DEV)).cpu()
        cpu(x)                             # first call calibrates the port on the same batch
        want = cpu(x)
    spread = float(want.std())
    mean_err = float((got - want).abs().mean())
    corr = float(torch.corrcoef(torch.stack([got.flatten(), want.flatten()]))[0, 1])
    print(f"fused plan vs CPU port: mean |diff| {mean_err:.4g}, logit spread {spread:.4g}, correlation {corr:.5f}")
    assert corr > 0.995 and mean_err < 0.03 * spread
    assert torch.equal(got.argmax(dim=1), want.argmax(dim=1))


def test_epilogue_quantiser_on_exact_ties_saturation_and_nan():
    """The reciprocal fast path of the epilogue quantiser must hand exact rounding ties (x.5), values a few ulps either
    side of them, saturating values and NaN to the exact division: with unit scales the convolution result is an
    exact integer, the bias moves it onto (and just off) the ties, and the consumer's scale is a power of two or an
    awkward number.  Codes must equal the stand-alone kernel's (half-to-even, clamp, NaN -> 0) bit for bit."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, c, h, k = 2, 64, 6, 64
    g = gen(900)
    codes = torch.randint(0, 5, (n, c, h, h), generator=g).to(torch.uint8).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randint(-1, 2, (k, c, 1, 1), generator=g).float().to(DEV)
    one = torch.ones(k, device=DEV)
    wq, wsum = K.quantize_weight_krsc(wt, one, -127, 127)
    s_in, zp_in = torch.tensor([1.0], device=DEV), torch.tensor([0.0], device=DEV)
    ulp = 2.0 ** -20
    offs = torch.tensor([0.5, -0.5, 0.5 + ulp, 0.5 - ulp, 1.5, 2.5, 0.25, 0.0], device=DEV)
    bias = offs.repeat(k // offs.numel())
    bias[7] = float("nan")
    bias[15] = float("inf")
    bias[23] = -float("inf")
    bias[31] = 1e30
    plain = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, one)
    assert torch.isnan(plain[:, 7]).all() and bool((plain[:, 0] * 2 % 2 == 1).all())      # exact x.5 values are present
    for q_scale, zp, lo, hi, form in [(1.0, 0.0, -127, 127, N.FORM_SYMMETRIC), (1.0, 3.0, 0, 255, N.FORM_ZEROPOINT),
                                      (0.5, 0.0, 0, 255, N.FORM_ZEROPOINT), (0.1, 7.0, 0, 255, N.FORM_ZEROPOINT),
                                      (1.0, 0.0, -7, 7, N.FORM_QBASE), (3.0, 0.0, 0, 15, N.FORM_EMULATE),
                                      (1e-30, 0.0, 0, 255, N.FORM_ZEROPOINT), (1e-41, 0.0, 0, 255, N.FORM_ZEROPOINT)]:
        qs = torch.tensor([q_scale], device=DEV)
        qz = torch.tensor([zp], device=DEV)
        emit = K.EmitCodes(qs, qz, lo, hi, form, 0.0)
        for relu in (False, True):
            want = torch.relu(plain) if relu else plain
            _, wc = K.fake_quant(want, qs, qz, lo, hi, form, codes="i8", want_y=False)
            _, got = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, one, relu=relu, emit=emit, want_out=False)
            same(got, wc, f"scale {q_scale} zp {zp} [{lo},{hi}] form {form} relu={relu}")


def test_epilogue_quantiser_random_sweep():
    """The epilogue quantiser against the stand-alone kernel over many random consumers (scale, zero point, range, form):
    74 k values each, of which ~0.05 % land inside the tie margin and take the exact-division path."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = gen(901)
    n, c, h, k = 4, 64, 12, 128
    codes = torch.randint(0, 256, (n, c, h, h), generator=g).to(torch.uint8).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(k, c, 3, 3, generator=g) * 0.05).to(DEV)
    s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
    wq, wsum = K.quantize_weight_krsc(wt, s_w, -127, 127)
    bias = torch.randn(k, generator=g).to(DEV)
    s_in, zp_in = torch.tensor([0.0173], device=DEV), torch.tensor([2.0], device=DEV)
    plain = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, s_w, padding=1)
    vmax = float(plain.abs().max())
    forms = [N.FORM_EMULATE, N.FORM_QBASE, N.FORM_ZEROPOINT, N.FORM_SYMMETRIC]
    for i in range(48):
        form = forms[i % 4]
        bits = [8, 8, 4, 2][(i // 4) % 4]
        signed = (i // 16) % 2 == 1 or form in (N.FORM_QBASE, N.FORM_SYMMETRIC)
        lo, hi = (-(2 ** (bits - 1) - 1), 2 ** (bits - 1) - 1) if signed else (0, 2 ** bits - 1)
        scale = vmax / hi * float(torch.empty(1).uniform_(0.2, 1.5, generator=g))
        zp = 0.0 if form in (N.FORM_QBASE, N.FORM_SYMMETRIC) else float(torch.randint(lo, hi + 1, (1,), generator=g))
        if form == N.FORM_EMULATE:
            zp = float(torch.empty(1).uniform_(-1, 1, generator=g)) * scale       # EMULATE's offset is in value units
        qs, qz = torch.tensor([scale], device=DEV), torch.tensor([zp], device=DEV)
        gq = 1e-3 if form == N.FORM_QBASE else 0.0
        emit = K.EmitCodes(qs, qz, lo, hi, form, gq)
        relu = i % 3 == 0
        want = torch.relu(plain) if relu else plain
        _, wc = K.fake_quant(want, qs, qz, lo, hi, form, g=gq, codes="i8", want_y=False)
        _, got = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp_in, s_w, padding=1, relu=relu, emit=emit, want_out=False)
        same(got, wc, f"sweep {i}: form {form} [{lo},{hi}] scale {scale:.4g} zp {zp:.4g}")


def test_stem_and_dual_kernels_random_shapes():
    """Random geometries for the first-layer kernel (channels 1-4, filter rows 1-7, taps 1-8, stride 1-3, padding 0-3) and
    for the dual kernel (two operand pairs of different size / stride / taps giving one output shape)."""
    import torch.nn.functional as F
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = gen(1234)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))
    done = 0
    while done < 40:
        n, c, r, s, stride, pad = ri(1, 3), ri(1, 4), ri(1, 7), ri(1, 8), ri(1, 3), ri(0, 3)
        h, w, k = ri(1, 20), ri(1, 20), 4 * ri(1, 24)
        if h + 2 * pad < r or w + 2 * pad < s:
            continue
        done += 1
        x = torch.randn(n, c, h, w, generator=g).to(DEV)
        if done % 2:
            x = x.contiguous(memory_format=torch.channels_last)
        unsigned = done % 3 != 0
        lo, hi = (0, 255) if unsigned else (-127, 127)
        s_in = torch.tensor([float(x.abs().max()) / 120 + 1e-3], device=DEV)
        zp = torch.tensor([float(ri(100, 140)) if unsigned else 0.0], device=DEV)
        _, codes = K.fake_quant(x, s_in, zp, lo, hi, N.FORM_ZEROPOINT, codes="i8", want_y=False)
        wt = (torch.randn(k, c, r, s, generator=g) * 0.1).to(DEV)
        s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
        bias = torch.randn(k, generator=g).to(DEV)
        qw = torch.clamp(torch.round(wt.cpu() / s_w.cpu().reshape(-1, 1, 1, 1)), -127, 127)
        ref = F.conv2d((codes.cpu().double() - float(zp)) * float(s_in), qw.double() * s_w.cpu().double().reshape(-1, 1, 1, 1),
                       bias.cpu().double(), stride=stride, padding=pad)
        xpad = K.quantize_pad_nhwc4(x, s_in, zp, lo, hi, N.FORM_ZEROPOINT, pad)
        wq, wsum = K.quantize_weight_stem(wt, s_w, -127, 127)
        got = K.conv2d_i8_stem(xpad, wq, wsum, bias, s_in, zp, s_w, s, stride=stride)
        torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-6, atol=2e-5,
                                   msg=lambda m: f"stem {done}: n{n} c{c} {h}x{w} k{k} {r}x{s} s{stride} p{pad}: {m}")
    done = 0
    while done < 24:
        n, k = ri(1, 3), [32, 64, 128, 192, 256][ri(0, 4)]
        p_out = ri(1, 9)

        def operand(unsigned):
            c, r, stride = 64 * ri(1, 2), [1, 3][ri(0, 1)], ri(1, 2)
            pad = r // 2
            h = (p_out - 1) * stride + 1 + ri(0, stride - 1)      # any input size giving p_out rows
            lo, hi = (0, 256) if unsigned else (-127, 128)
            codes = torch.randint(lo, hi, (n, c, h, h), generator=g).to(torch.uint8 if unsigned else torch.int8).to(DEV)
            wt = (torch.randn(k, c, r, r, generator=g) * 0.05).to(DEV)
            s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
            wq, wsum = K.quantize_weight_krsc(wt, s_w, -127, 127)
            return dict(codes=codes.contiguous(memory_format=torch.channels_last), wq=wq, wsum=wsum,
                        bias=torch.randn(k, generator=g).to(DEV) if ri(0, 1) else None,
                        in_scale=torch.tensor([0.01 + 0.001 * ri(0, 9)], device=DEV),
                        in_zp=torch.tensor([float(ri(0, 9)) if unsigned else 0.0], device=DEV), w_scale=s_w, stride=stride, padding=pad)
        a, b = operand(True), operand(done % 2 == 0)
        done += 1

        def single(t):
            return K.conv2d_i8(t["codes"], t["wq"], t["wsum"], t["bias"], t["in_scale"], t["in_zp"], t["w_scale"],
                               stride=t["stride"], padding=t["padding"])
        want = torch.relu(single(a) + single(b))
        q_s = torch.tensor([float(want.max()) / 255 + 1e-3], device=DEV)
        emit = K.EmitCodes(q_s, torch.tensor([0.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
        _, wc = K.fake_quant(want, q_s, emit.zero_point, 0, 255, N.FORM_ZEROPOINT, codes="i8", want_y=False)
        out, codes = K.conv2d_i8_dual(a, b, relu=True, emit=emit)
        same(out, want, f"dual {done} out")
        same(codes, wc, f"dual {done} codes")


@pytest.mark.parametrize("name,family", [("resnet50", "FSPTQ"), ("resnet18", "FSPTQ"), ("resnet18", "QBase")])
def test_eager_fused_calibrates_and_runs_like_the_model(name, family):
    """dlmc.utils.fuse.EagerFused: the wrappers keep observing / calibrating / quantising as in `model(x)`, while every
    layer -> (+ shortcut) -> ReLU chain on the int8 route is one launch.  A calibrating forward through it must leave EVERY
    wrapper with the scales, offsets and init flags `model(x)` leaves, and return the same bits; so must later forwards."""
    import copy
    import workloads as W
    from dlmc.utils.fuse import EagerFused
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    if family == "QBase":      # (its int8 route: per-tensor scales, signed codes, zero offsets)
        cfg["weight"]["type"] = "minmax_tensor"
        cfg["input"]["args"]["signed"] = True
    torch.manual_seed(99)
    net = W.MODELS[name]().to(DEV).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = merge_bn(net, inplace=True)
    quantize_model(net, cfg, None, family, int8_gemm=True)
    twin = copy.deepcopy(net)
    x = torch.relu(torch.randn(6, 3, 64, 64, device=DEV))
    with torch.no_grad():
        want0 = net(x)                       # calibrates, module by module
        fused = EagerFused(twin)
        assert len(fused.chains) >= 16 and any(c[0] is not None for c in fused.chains.values())
        got0 = fused(x)                      # calibrates through the fused epilogues
        assert sum(v == "fused" for v in fused.last_states.values()) >= 16, fused.last_states      # (the route was really taken)
        sa, sb = net.state_dict(), twin.state_dict()
        assert sa.keys() == sb.keys()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
        assert torch.equal(want0.view(torch.int32), got0.view(torch.int32))
        want1, got1 = net(x * 0.7), fused(x * 0.7)
        assert torch.equal(want1.view(torch.int32), got1.view(torch.int32))
        assert torch.equal(twin(x * 0.7).view(torch.int32), got1.view(torch.int32))      # the wrappers themselves are untouched


def test_eager_fused_speculates_on_integer_zero_points_and_recovers_when_wrong():
    """Round 5: EagerFused lets freshly calibrated FSPTQ layers ASSUME an integer zero point (one host read per forward instead of one per
    layer: ZeroPointSpeculation).  With post-ReLU pixels every assumption holds; with N(0, 1) pixels the first layer's zero point is the
    (negative, non-integer) minimum - its guess is wrong, every layer behind it calibrated on garbage, and the forward must notice, re-arm
    and run again: in both cases the scales and the output are exactly those of `model(x)`."""
    import copy
    import workloads as W
    from dlmc.utils.fuse import EagerFused
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    for pixels, held in (("halfnormal", True), ("normal", False)):
        torch.manual_seed(5)
        net = merge_bn(W.MODELS["resnet18"]().to(DEV).eval(), inplace=True, allow_missing=True)
        quantize_model(net, cfg, None, "FSPTQ", int8_gemm=True)
        twin = copy.deepcopy(net)
        x = torch.randn(4, 3, 64, 64, device=DEV)
        if pixels == "halfnormal":
            x = torch.relu(x)
        with torch.no_grad():
            want = net(x)
            fused = EagerFused(twin)
            got = fused(x)
        assert fused.speculation["layers"] >= 16 and fused.speculation["held"] is held, (pixels, fused.speculation)
        sa, sb = net.state_dict(), twin.state_dict()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (pixels, k)
        assert torch.equal(want.view(torch.int32), got.view(torch.int32)), pixels


def test_eager_fused_keeps_its_promise_on_shortcuts_it_cannot_fuse():
    """EagerFused promises `y == model(x)`: a broadcast shortcut (not the layer's output shape) and an in-place add INTO the shortcut
    (`short += layer(x)`: the fused launch would not mutate `short`) must run layer, add and ReLU one by one - same bits, and the
    in-place add still mutates what it mutates in the model."""
    import copy
    from dlmc.utils.fuse import EagerFused
    from dlmc.utils.quantize import quantize_model

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv2d(64, 64, 1)
            self.b = torch.nn.Conv2d(64, 64, 3, padding=1)
            self.c = torch.nn.Conv2d(64, 64, 1)
            self.pool = torch.nn.AvgPool2d(3, 1, 1)
            self.relu = torch.nn.ReLU()

        def forward(self, x):
            y = self.relu(self.a(x) + x.mean(dim=(2, 3), keepdim=True))      # broadcast shortcut [N, 64, 1, 1]
            idt = self.pool(y)
            idt.add_(self.b(y))                                               # in-place add into the shortcut (`idt += ...` reaches torch.fx as a plain add)
            z = self.relu(idt)
            return self.relu(self.c(z) + z), idt                              # a shortcut the epilogue does take; idt as the model leaves it

    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    torch.manual_seed(7)
    net = Net().to(DEV).eval()
    quantize_model(net, cfg, None, "FSPTQ", int8_gemm=True)
    twin = copy.deepcopy(net)
    x = torch.relu(torch.randn(4, 64, 12, 12, device=DEV))
    with torch.no_grad():
        want = net(x)
        fused = EagerFused(twin)
        got = fused(x)
        st = fused.last_states
        assert st["a"] == "plain" and st.get("b", "plain") == "plain" and st["c"] == "fused", st
        for g, w in zip(got, want):
            assert torch.equal(g.view(torch.int32), w.view(torch.int32))
        for g, w in zip(fused(x * 0.5), net(x * 0.5)):
            assert torch.equal(g.view(torch.int32), w.view(torch.int32))


@pytest.mark.parametrize("shape", [(6, 64, 20, 20, 256, True, True), (5, 128, 9, 13, 64, False, False), (3, 64, 7, 7, 1000, True, False)], ids=str)
def test_observing_epilogue_gives_the_observer_s_min_and_max(shape):
    """dlmcq_conv2d_i8_nhwc_fused_observed + dlmcq_minmax_finalize_f32 (round 5, the first batch): the (max, min) - and max |x|, and a NaN -
    of the tiled kernel's fp32 output from its own epilogue equal dlmcq_minmax_f32 over the stored tensor bit for bit, and the scale / offset
    derived from them equal observe_qparams's."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    from dlmc.quantization.scalar._wrapper import observe_minmax
    n, c, h, w, k, relu, with_res = shape
    g = torch.Generator(device=DEV).manual_seed(n + c + k)
    codes = torch.randint(0, 256, (n, c, h, w), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 4e-4 + 5e-5).contiguous()
    bias = torch.randn(k, generator=g, device=DEV)
    res = torch.randn((n, k, h, w), generator=g, device=DEV).contiguous(memory_format=torch.channels_last) if with_res else None
    s_in = torch.full((1,), 0.021, device=DEV)
    for poison in (None, float("nan"), float("inf")):
        if poison is not None and res is None:
            continue
        if poison is not None:
            res = res.clone()
            res[n - 1, k - 1, h - 1, w - 1] = poison
        out = K.conv2d_i8(codes, wq, wsum, bias, s_in, None, s_w, residual=res, relu=relu, observe=True)
        hint = K.minmax_hint(out)
        assert hint is not None and hint[1] >= 1
        for mode in (N.MINMAX_MINMAX, N.MINMAX_NEGMIN, N.MINMAX_ABSMAX):
            a, b = K.minmax_from_partials(*hint, mode=mode)
            a0, b0 = K.minmax(out, mode=mode)
            assert torch.equal(a.view(torch.int32), a0.view(torch.int32)), (poison, mode)
            assert b is None or torch.equal(b.view(torch.int32), b0.view(torch.int32)), (poison, mode)
        for signed in (False, True):
            s1, o1 = observe_minmax(out, 8, signed, hint=hint)
            s0, o0 = observe_minmax(out, 8, signed)
            assert torch.equal(s1.view(torch.int32), s0.view(torch.int32)) and torch.equal(o1.view(torch.int32), o0.view(torch.int32)), (poison, signed)
        out.add_(1.0)                                     # an in-place write: the hint no longer describes the tensor
        assert K.minmax_hint(out) is None
