"""BASELINE configs[4]: MobileOne-S1 W4A8, asymmetric per-channel weights (ops.py:129-136: unsigned `minmax_channel`, the
offset is the channel minimum), 8-bit unsigned per-tensor activations - on the int8 kernels of the frozen plan:
depthwise 3x3 layers on csrc/conv_dw_i8.hip, pointwise 1x1 layers on the matrix cores with the weight-offset term of
dlmcq_conv2d_i8_nhwc_asym, 96-channel tensors zero-padded to 128.  Every plan node is checked against the float64
reference of tests/plan_reference.py computed from ITS OWN input (no drift): fp32 to rtol 2e-6 of the summed magnitude,
codes exact from the kernel's own fp32 value and within one code on < 1e-3 of the elements otherwise."""
import json

import pytest
import torch

from plan_reference import check_node as _check_node, close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

W4A8 = {  # QBase family (quantization_type=None), per-channel extension
    "weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 4, "signed": False}},
    "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
    "exclude_layers": [], "override_options": [],
}


W8A8_FSPTQ = {  # example/quantization/FSPTQ_config.yaml:40-53 (symmetric per-channel s8 weights, u8 activations)
    "weight": {"enable": True, "type": "minmax_channel", "recon_type": "None", "args": {"n_bits": 8, "signed": True}},
    "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
    "exclude_layers": [], "override_options": [],
}


# (None, 1024, 224): BASELINE configs[4] at its stated size - every node on windows of the first and the LAST image
@pytest.mark.parametrize("family,batch,size,full", [(None, 3, 64, True), (None, 64, 224, False), ("FSPTQ", 3, 64, True), (None, 1024, 224, False)])
def test_mobileone_s1_w4a8_plan_node_by_node(family, batch, size, full):
    import workloads as W
    from dlmc.utils.fuse import DwInt8Layer, Int8Layer, StemLayer, fuse_inference
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    net = W.mobileone_s1_deploy().to(DEV).eval()
    quantize_model(net, json.loads(json.dumps(W4A8 if family is None else W8A8_FSPTQ)), None, family)
    x = torch.relu(torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(5))).to(DEV)
    recs = []
    with torch.no_grad():
        want = net(x)                       # calibrates; the module path (fp32 convolutions of the fake-quantised operands)
        # node by node on the plan bench.py times (every layer its own launch, its own hook); the same plan with its 17 stride-1
        # depthwise + pointwise units as ONE kernel each (csrc/conv_dwpw_i8.hip; `dwpw=True`, not the default: measured no faster)
        # must give the same bits
        plan = fuse_inference(net)
        fused = fuse_inference(net, dwpw=True)
        assert fused.fusion_report.dwpw == 17 and plan.fusion_report.dwpw == 0, fused.fusion_report
        rep = plan.fusion_report
        for m in plan.modules():
            if isinstance(m, (Int8Layer, StemLayer)):
                m.register_forward_hook(lambda mod, args, out: recs.append((mod, args, out)))
        got = plan(x)
        got_fused = fused(x)
    assert torch.equal(got_fused.view(torch.int32), got.view(torch.int32)), "the plan with fused depthwise + pointwise units differs from the unfused plan"
    n_dw = sum(isinstance(r[0], DwInt8Layer) for r in recs)
    # the 3-channel first layer (matrix cores, with the weight-offset term for the asymmetric W4 weights), 21 units, the classifier
    assert n_dw == 21 and len(recs) == 1 + 21 + 21 + 1 and isinstance(recs[0][0], StemLayer), (n_dw, len(recs), rep)
    assert rep.skipped == [] and rep.stem == 1, rep
    rates = {}
    for idx, (mod, args, out) in enumerate(recs):
        if mod.layer.weight.dim() == 2:
            continue                         # the classifier reads fp32 features: checked through the logits below
        _check_node(idx, mod, args, out, full, rates=rates)
    worst = max(rates.values()) if rates else 0.0
    print(f"mobileone_s1 {family} batch {batch} @{size}: worst off-by-one rate of a codes-only node {worst:.2e} "
          f"(node {max(rates, key=rates.get) if rates else None})")
    assert worst <= 1.2e-4, rates            # twice what round 3 measured (5.9e-5 at batch 1024): a drift must not hide under the 2e-3 bound
    # logits against the module path: same codes layer by layer up to fp32 accumulation-order ties
    spread = float(want.std())
    # (round 5: `A and B or batch > 8` had skipped BOTH checks above batch 8.  The mean deviation is checked at every batch; the predicted
    #  class must agree on every image of a small batch and on all but a tie-breaking handful of a large one)
    assert float((got - want).abs().mean()) < 0.05 * spread, (float((got - want).abs().mean()), spread)
    agree = float((got.argmax(1) == want.argmax(1)).double().mean())
    assert agree == 1.0 if batch <= 8 else agree >= 0.98, agree


def test_depthwise_kernel_shapes_and_forms():
    """conv2d_dw_i8 alone: strides, paddings, 5x5 taps, signed codes, symmetric and asymmetric weights, a zero point."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    import torch.nn.functional as F
    for idx, (n, c, h, w, r, stride, pad, signed, asym) in enumerate([(2, 64, 9, 9, 3, 1, 1, False, True), (3, 8, 12, 7, 3, 2, 1, False, False),
                                                                      (1, 192, 14, 14, 5, 1, 2, True, True), (2, 12, 6, 6, 3, 1, 0, False, True),
                                                                      # the two-pixels-per-thread kernel (3x3, stride 1, padding 1, C % 16 == 0):
                                                                      # even / odd widths, a single column, signed codes, one row
                                                                      (1, 128, 5, 6, 3, 1, 1, False, False), (2, 32, 3, 1, 3, 1, 1, True, True),
                                                                      (3, 16, 1, 7, 3, 1, 1, False, True), (1, 96 + 16, 28, 28, 3, 1, 1, False, True),
                                                                      # the one-pixel 16-channel kernel (3x3, C % 16 == 0, any stride / padding): MobileOne's stride-2 layers
                                                                      (2, 64, 13, 12, 3, 2, 1, False, True), (2, 32, 9, 9, 3, 1, 0, False, False),
                                                                      (1, 48, 8, 8, 3, 2, 1, True, True), (2, 16, 7, 9, 3, 2, 1, False, True)]):
        g = torch.Generator().manual_seed(40 + idx)
        lo, hi = (-127, 127) if signed else (0, 255)
        codes = torch.randint(lo, hi + 1, (n, c, h, w), generator=g).to(torch.int8 if signed else torch.uint8)
        zp = 0.0 if (signed or idx % 2) else 4.0      # (zero point 0: the two-pixel kernel reads its border taps as out-of-range buffer loads)
        qw = torch.randint(0, 16, (c, 1, r, r), generator=g)
        s_w = torch.rand(c, generator=g) * 0.02 + 0.001
        o_w = (torch.randn(c, generator=g) * 0.05) if asym else None
        bias = torch.randn(c, generator=g)
        s_in = 0.03
        xd = (codes.double() - zp) * s_in
        wd = qw.double() * s_w.double().reshape(-1, 1, 1, 1) + (o_w.double().reshape(-1, 1, 1, 1) if asym else 0)
        ref = torch.relu(F.conv2d(xd, wd, bias.double(), stride=stride, padding=pad, groups=c))
        mag = F.conv2d(xd.abs(), wd.abs(), bias.double().abs(), stride=stride, padding=pad, groups=c)
        emit = K.EmitCodes(torch.tensor([float(ref.max()) / 255 + 1e-3], device=DEV), torch.tensor([2.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
        out, oc = K.conv2d_dw_i8(codes.to(DEV).contiguous(memory_format=torch.channels_last), qw[:, 0].permute(1, 2, 0).contiguous().to(torch.int8).to(DEV),
                                 bias.to(DEV), torch.tensor([s_in], device=DEV), torch.tensor([zp], device=DEV), s_w.to(DEV),
                                 None if o_w is None else o_w.to(DEV), stride=stride, padding=pad, relu=True, emit=emit)
        close(out.cpu(), ref, mag, f"dw case {idx}")
        from oracle import fakequant_oracle as O
        assert torch.equal(oc.cpu().float(), O.fq_zeropoint(out.cpu(), emit.scale.cpu(), emit.zero_point.cpu(), 0, 255)[0]), f"dw case {idx} codes"
        # codes only (the ReLU folded into the quantiser's clamp): the same bytes
        none, oc2 = K.conv2d_dw_i8(codes.to(DEV).contiguous(memory_format=torch.channels_last), qw[:, 0].permute(1, 2, 0).contiguous().to(torch.int8).to(DEV),
                                   bias.to(DEV), torch.tensor([s_in], device=DEV), torch.tensor([zp], device=DEV), s_w.to(DEV),
                                   None if o_w is None else o_w.to(DEV), stride=stride, padding=pad, relu=True, emit=emit, want_out=False)
        assert none is None and torch.equal(oc2, oc), f"dw case {idx} codes-only"
        # the plain unsigned-byte quantiser (no zero point: every post-ReLU tensor of the frozen plans) takes the two-pixel kernel's
        # pair-arithmetic path when only codes are wanted: the same bytes as the general path (fp32 output wanted as well), with and
        # without a bias
        plain = K.EmitCodes(emit.scale, None, 0, 255, N.FORM_ZEROPOINT)
        for bb in (bias.to(DEV), None):
            def run(want_out):
                return K.conv2d_dw_i8(codes.to(DEV).contiguous(memory_format=torch.channels_last), qw[:, 0].permute(1, 2, 0).contiguous().to(torch.int8).to(DEV),
                                      bb, torch.tensor([s_in], device=DEV), torch.tensor([zp], device=DEV), s_w.to(DEV),
                                      None if o_w is None else o_w.to(DEV), stride=stride, padding=pad, relu=True, emit=plain, want_out=want_out)
            out3, oc3 = run(True)
            none, oc4 = run(False)
            assert none is None and torch.equal(oc4, oc3), f"dw case {idx} plain quantiser, codes-only vs with fp32 output ({int((oc4 != oc3).sum())} differ)"
            assert torch.equal(oc3.cpu().float(), O.fq_zeropoint(out3.cpu(), plain.scale.cpu(), torch.zeros(1), 0, 255)[0]), f"dw case {idx} plain codes"


@pytest.mark.parametrize("n,c,h,w,signed,asym,has_bias,zp", [
    (8, 64, 28, 28, False, True, True, 0.0),        # 6 272 pixels, one chunk
    (3, 192, 40, 37, False, True, True, 0.0),       # MobileOne-S1 stage 2's width, an odd image width
    (2, 128, 56, 56, False, False, True, 3.0),      # the widest frame (57 positions per row), symmetric weights, a zero point
    (24, 512, 14, 14, True, True, False, 0.0),      # stage 3: eight chunks, signed codes (no re-centring of the fragments), no bias
    (5, 64, 33, 30, True, False, False, -7.0),
    (1024, 192, 28, 28, False, True, True, 0.0),    # BASELINE configs[4] at its stated size: a stage-2 depthwise layer at batch 1024
    (1024, 512, 14, 14, True, True, True, 0.0),     # ... and a stage-3 layer as the plan feeds it (re-centred codes)
])
def test_depthwise_on_the_matrix_cores_is_the_vector_kernel_bit_for_bit(n, c, h, w, signed, asym, has_bias, zp):
    """csrc/conv_dwm_i8.hip (3x3 / stride 1 / padding 1, codes only, the plain quantiser, >= 4 096 pixels) against
    conv_dw3p2_i8_kernel (the same call on sub-batches below that size) and against a float64 convolution."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n * 100 + c)
    lo, hi = (-128, 127) if signed else (0, 255)
    codes = torch.randint(lo, hi + 1, (n, c, h, w), generator=g).to(torch.int8 if signed else torch.uint8)
    qw = torch.randint(0, 16, (c, 1, 3, 3), generator=g) if asym else torch.randint(-8, 8, (c, 1, 3, 3), generator=g)
    s_w = torch.rand(c, generator=g) * 0.02 + 0.001
    o_w = (torch.randn(c, generator=g) * 0.05) if asym else None
    bias = torch.randn(c, generator=g) if has_bias else None
    s_in = 0.03
    xd = (codes.double() - zp) * s_in
    wd = qw.double() * s_w.double().reshape(-1, 1, 1, 1) + (o_w.double().reshape(-1, 1, 1, 1) if asym else 0)
    ref = torch.relu(F.conv2d(xd, wd, None if bias is None else bias.double(), padding=1, groups=c))
    scale = torch.tensor([float(ref.max()) / 255 + 1e-3], device=DEV)
    emit = K.EmitCodes(scale, None, 0, 255, N.FORM_ZEROPOINT)
    cd = codes.to(DEV).contiguous(memory_format=torch.channels_last)
    wq = qw[:, 0].permute(1, 2, 0).contiguous().to(torch.int8).to(DEV)

    def run(x):
        none, oc = K.conv2d_dw_i8(x, wq, None if bias is None else bias.to(DEV), torch.tensor([s_in], device=DEV), torch.tensor([zp], device=DEV),
                                  s_w.to(DEV), None if o_w is None else o_w.to(DEV), stride=1, padding=1, relu=True, emit=emit, want_out=False)
        assert none is None
        return oc
    assert n * h * w >= 4096
    K.PROFILE.reset()
    K.PROFILE.enabled = True
    try:
        got = run(cd)
        first = run(cd[:1].contiguous(memory_format=torch.channels_last))
    finally:
        K.PROFILE.enabled = False
    assert [r[0] for r in K.PROFILE.records] == ["conv_dwm", "conv_dw"]      # (the profile tag follows the library's own dispatch rule)
    K.PROFILE.reset()
    assert torch.equal(got[:1], first)
    # round 5: the whole tensor through the vector kernel (DLMCQ_FORCE_TILED on the depthwise entry point)
    none, whole = K.conv2d_dw_i8(cd, wq, None if bias is None else bias.to(DEV), torch.tensor([s_in], device=DEV), torch.tensor([zp], device=DEV),
                                 s_w.to(DEV), None if o_w is None else o_w.to(DEV), stride=1, padding=1, relu=True, emit=emit, want_out=False,
                                 force_tiled=True)
    assert torch.equal(got, whole), f"{int((got != whole).sum())} of {got.numel()} codes differ from the vector kernel"
    del whole
    nsub = max(1, 4095 // (h * w))
    for i0 in (0, n - nsub):            # the first and the last images through the vector kernel
        sub = run(cd[i0:i0 + nsub].contiguous(memory_format=torch.channels_last))
        assert torch.equal(got[i0:i0 + nsub], sub), f"{int((got[i0:i0 + nsub] != sub).sum())} codes differ from the vector kernel (images {i0}..)"
    want = torch.clamp(torch.round(ref / float(scale)), 0, 255)
    diff = (got.cpu().double() - want).abs()
    assert float(diff.max()) <= 1.0 and float((diff > 0).double().mean()) < 2e-3, (float(diff.max()), float((diff > 0).double().mean()))


@pytest.mark.parametrize("c,k,r,stride,pad", [(3, 64, 3, 2, 1), (1, 32, 5, 1, 2), (4, 96, 7, 2, 3), (2, 8, 1, 1, 0)])
def test_asymmetric_first_layer_kernel(c, k, r, stride, pad):
    """dlmcq_conv2d_i8_stem_asym alone: every channel count the padded NHWC4 buffer allows, taps up to 7, against the float64
    convolution of (q - zp) * s_in with w' = qw * s_w + o_w (the operand-sum term must skip the 4th byte and the taps beyond S)."""
    import torch.nn.functional as F
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator().manual_seed(c * 100 + k)
    n, h = 3, 19
    x = torch.relu(torch.randn(n, c, h, h, generator=g))
    s_in, zp = torch.tensor([float(x.max()) / 255]), torch.tensor([0.0])
    qw = torch.randint(0, 16, (k, c, r, r), generator=g)
    s_w, o_w = torch.rand(k, generator=g) * 0.05 + 0.01, -torch.rand(k, generator=g) * 0.3
    bias = torch.randn(k, generator=g)
    xpad = K.quantize_pad_nhwc4(x.to(DEV), s_in.to(DEV), zp.to(DEV), 0, 255, N.FORM_ZEROPOINT, pad)
    full = torch.zeros(k, r, 8, 4, dtype=torch.int8)
    full[:, :, :r, :c] = qw.permute(0, 2, 3, 1).to(torch.int8)
    wsum = qw.sum(dim=(1, 2, 3)).to(torch.int32)
    out = K.conv2d_i8_stem(xpad, full.to(DEV), wsum.to(DEV), bias.to(DEV), s_in.to(DEV), zp.to(DEV), s_w.to(DEV), r, stride=stride,
                           relu=False, w_offset=o_w.to(DEV), channels=c)
    codes = torch.clamp(torch.round(x / s_in), 0, 255)
    xd = (codes.double() - 0.0) * float(s_in)
    wd = qw.double() * s_w.double().reshape(-1, 1, 1, 1) + o_w.double().reshape(-1, 1, 1, 1)
    ref = F.conv2d(xd, wd, bias.double(), stride=stride, padding=pad)
    mag = F.conv2d(xd.abs(), wd.abs(), bias.double().abs(), stride=stride, padding=pad)
    close(out.cpu(), ref, mag, f"asym stem c={c} k={k} r={r}")


def test_plan_holds_packed_int4_weights_and_loads_the_integer_checkpoint():
    """BASELINE configs[4] names "sub-byte pack/unpack": the frozen plan stores its 4-bit weight codes two per byte and expands them
    with ONE dlmcq_unpack_int4 launch per forward (dlmc/utils/fuse.py PackedWeights4); an integer checkpoint of dlmc.utils.export
    (packed int4, KCRS order) loads into the plan on the device, no fp32 weights involved.  All three plans - unpacked, packed,
    packed from the checkpoint - give the same bits."""
    import workloads as W
    from dlmc.quantization.scalar import kernels as K
    from dlmc.utils.export import export_quantized_state
    from dlmc.utils.fuse import _PlanLayer, fuse_inference
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    net = W.mobileone_s1_deploy().to(DEV).eval()
    quantize_model(net, json.loads(json.dumps(W4A8)), None, None)
    x = torch.relu(torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(5))).to(DEV)
    with torch.no_grad():
        net(x)
        plain = fuse_inference(net, pack_int4=False)
        packed = fuse_inference(net)
        blob = export_quantized_state(net)
        assert all(rec["packed_int4"] for rec in blob["layers"].values())
        loaded = fuse_inference(net, weight_blob=blob)
        want = plain(x * 0.9)
        K.PROFILE.enabled = True
        K.PROFILE.reset()
        got = packed(x * 0.9)
        K.PROFILE.enabled = False
        got2 = loaded(x * 0.9)
    assert [t for t, *_ in K.PROFILE.records].count("unpack_int4") == 1          # one expansion per forward, whole network
    assert torch.equal(got, want) and torch.equal(got2, want)
    assert plain.packed_weights == [] and len(packed.packed_weights) == 1
    holder = packed.packed_weights[0]
    nodes = [m for m in packed.modules() if isinstance(m, _PlanLayer) and hasattr(m, "wq")]
    assert len(nodes) == 44
    lo, hi = holder.scratch.data_ptr(), holder.scratch.data_ptr() + holder.scratch.numel()
    assert all(lo <= m.wq.data_ptr() < hi for m in nodes)                        # every layer's codes live in the expanded scratch only
    assert holder.packed.numel() * 2 == holder.scratch.numel()
    # the same codes whichever way they came: quantised from fp32 at plan build, or expanded from the checkpoint
    for a, b in zip(nodes, [m for m in loaded.modules() if isinstance(m, _PlanLayer) and hasattr(m, "wq")]):
        assert torch.equal(a.wq, b.wq)
    # 8-bit networks are left alone
    from dlmc.utils.merge_bn import merge_bn
    r18 = merge_bn(W.resnet18().to(DEV).eval(), inplace=True, allow_missing=True)
    quantize_model(r18, json.loads(json.dumps(W8A8_FSPTQ)), None, "FSPTQ", int8_gemm=True)
    with torch.no_grad():
        r18(torch.relu(torch.randn(2, 3, 64, 64, device=DEV)))
        assert fuse_inference(r18).packed_weights == []


@pytest.mark.parametrize("n,c,k,h,w,asym,zp", [(2, 64, 128, 9, 9, True, 4.0), (3, 128, 128, 56, 56, True, 0.0), (5, 192, 192, 28, 28, True, 3.0),
                                            (4, 512, 512, 14, 14, True, 0.0), (1, 192, 512, 7, 5, False, 2.0), (7, 64, 192, 1, 1, True, 0.0),
                                            (2, 256, 192, 3, 61, True, 1.0), (9, 128, 512, 2, 7, False, 0.0)])
def test_depthwise_plus_pointwise_kernel_matches_the_two_launches(n, c, k, h, w, asym, zp):
    """dlmcq_conv2d_dwpw_i8_nhwc (one launch: depthwise 3x3 / 1 / 1 + ReLU + quantiser, then pointwise 1x1 + ReLU + quantiser) against
    dlmcq_conv2d_dw_i8_nhwc followed by dlmcq_conv2d_i8_nhwc_asym / _fused: the same bytes.  Shapes: all three widths, the widest
    image the kernel takes (61), images smaller than a tile, tiles crossing images, a partial last tile, symmetric and asymmetric
    weights on both layers, a non-zero zero point on the depthwise input (border code) and QBASE / ZEROPOINT quantisers."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(n * 1000 + c + k + h)
    codes = torch.randint(0, 256, (n, c, h, w), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wdw = torch.randint(0, 16, (3, 3, c), generator=g, device=DEV).to(torch.int8)
    s_dw = torch.rand(c, generator=g, device=DEV) * 0.02 + 0.001
    o_dw = torch.randn(c, generator=g, device=DEV) * 0.05 if asym else None
    b_dw = torch.randn(c, generator=g, device=DEV) * 0.3
    s_in, zpt = torch.tensor([0.013], device=DEV), torch.tensor([zp], device=DEV)
    wpw = torch.randint(-8 if not asym else 0, 8 if not asym else 16, (k, 1, 1, c), generator=g, device=DEV).to(torch.int8)
    pw = dict(wq=wpw, wsum=wpw.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k, generator=g, device=DEV),
              w_scale=torch.rand(k, generator=g, device=DEV) * 0.01 + 0.0005, w_offset=(torch.randn(k, generator=g, device=DEV) * 0.01 if asym else None))
    form = N.FORM_QBASE if asym else N.FORM_ZEROPOINT
    emit = K.EmitCodes(torch.tensor([0.021], device=DEV), torch.zeros(1, device=DEV), 0, 255, form, g=1e-3 if asym else 0.0)
    emit2 = K.EmitCodes(torch.tensor([0.05], device=DEV), torch.zeros(1, device=DEV), 0, 255, form, g=2e-3 if asym else 0.0)
    pw_in = torch.tensor([0.021], device=DEV)
    _, mid = K.conv2d_dw_i8(codes, wdw, b_dw, s_in, zpt, s_dw, o_dw, stride=1, padding=1, relu=True, emit=emit, want_out=False)
    _, want = K.conv2d_i8(mid, pw["wq"], pw["wsum"], pw["bias"], pw_in, emit.zero_point, pw["w_scale"], relu=True, emit=emit2, want_out=False,
                          w_offset=pw["w_offset"])
    assert K.dwpw_supported(c, k, h, w, 1, 1, 3)
    table = K.dwpw_table(wdw, b_dw, s_in, zpt, s_dw, o_dw)
    got = K.conv2d_dwpw_i8(codes, table, asym, True, True, zpt, emit, dict(pw, in_scale=pw_in), relu=True, emit2=emit2)
    assert got.shape == want.shape and got.dtype == want.dtype
    bad = (got != want)
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} codes differ, first at {bad.nonzero()[:4].tolist()}"


def test_depthwise_plus_pointwise_argument_checks():
    from dlmc import _native as N
    one = torch.zeros(64, device=DEV)
    p = N.ptr(one)
    f = N.lib.dlmcq_conv2d_dwpw_i8_nhwc
    args = lambda K_, W_, C_=64, lo=0: (p, p, 1, 1, 1, p, 1, 4, W_, C_, 1, p, p, lo, 255, 2, 0.0, p, p, p, p, p, p, K_, 1, p, p, p, 0, 255, 2, 0.0, None)   # noqa: E731
    assert f(*args(256, 8)) == -1          # a width the kernel is not built for
    assert f(*args(192, 63)) == -1         # an image wider than its halo buffers (W + 1 > 63)
    assert f(*args(192, 8, 96)) == -1      # channels not a multiple of 64
    assert f(*args(192, 8, 64, -128)) == -1   # the matrix step reads unsigned bytes
