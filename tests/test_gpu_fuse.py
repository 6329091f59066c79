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


@pytest.mark.parametrize("arch,qtype,cfg,res", [("resnet50", "FSPTQ", FSPTQ, 64), ("resnet18", "FSPTQ", FSPTQ, 96),
                                                ("repvgg_a1", "FSPTQ", FSPTQ, 64), ("resnet18", None, QBASE, 64),
                                                ("mobileone_s1", "FSPTQ", FSPTQ, 64)])
def test_fused_plan_is_bit_identical_to_the_wrappers(arch, qtype, cfg, res):
    import workloads as W
    from dlmc.utils.fuse import fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    net = W.MODELS[arch]().to(DEV).eval()
    for m in net.modules():                       # non-trivial BN statistics, then fold them as FSPTQuant.py:67 does
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = merge_bn(net, inplace=True)
    quantize_model(net, cfg, None, qtype, int8_gemm=True)
    x = torch.randn(4, 3, res, res, device=DEV)
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
        assert rep.layers == 53 and rep.residual == 16 and rep.relu == 48 and rep.skipped == ["conv1"]
        assert rep.emit == 47 and rep.fp32_outputs == 18   # 12 shortcuts + 4 residual feeds + last block + fc
    if arch == "repvgg_a1":
        assert rep.layers == 22 and rep.relu == 21 and rep.fp32_outputs == 2    # the last block (feeds the pool) and the classifier
    # and under a HIP graph
    from dlmc.utils.graph import GraphedForward
    fwd = GraphedForward(fused, x)
    same(fwd(x * 0.8), want, f"{arch} {qtype} graphed")


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
