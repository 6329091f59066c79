"""csrc/conv_pw_i8.hip - pointwise (1x1, stride 1) codes-to-codes layers with the weights resident in LDS - against the tiled
kernel of csrc/conv_i8.hip (the same call below the size at which the pointwise kernel takes over: bit-identical codes demanded)
and against a float64 convolution of the dequantised operands + the oracle's quantiser (within one code, rarely).
Reference arithmetic: modules/conv.py:13-19 on FSPTQuant/base.py:108-109,149-152 / ops.py:129-136 operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (N, H, W, C, K, asym, bias, relu, x signed (int8 codes), input zero point)
CASES = [
    (40, 14, 14, 192, 192, True, True, True, False, 0.0),      # MobileOne-S1 stage 2 (K = 192: one slice, two passes of 96 channels)
    (24, 14, 14, 512, 512, True, True, True, False, 0.0),      # stage 3 (four slices of 128, one big workgroup per CU)
    (24, 14, 15, 192, 512, True, True, True, False, 0.0),      # stage 2 -> 3; 5 040 pixels: the last block of 32 is half empty
    (8, 28, 28, 128, 128, True, True, True, False, 3.0),       # the padded 96-channel tensors of stage 1, a zero point
    (8, 28, 27, 64, 128, False, False, False, True, 0.0),      # symmetric weights, no bias, no ReLU, signed codes
    (6, 28, 28, 128, 256, False, True, True, False, 7.0),      # symmetric, two slices
    (12, 20, 20, 64, 192, True, False, True, False, 0.0),
    (9, 23, 23, 512, 128, False, True, True, False, 0.0),      # 4 761 pixels, one slice
    (10, 24, 24, 64, 64, False, True, True, False, 0.0),       # ResNet-50's first reduction (64 -> 64): a 64-wide slice
    (1024, 28, 28, 192, 192, True, True, True, False, 0.0),    # BASELINE configs[4] at its stated size: a stage-2 layer of MobileOne-S1 at batch 1024
    (1024, 14, 14, 512, 512, True, True, True, False, 0.0),    # ... and a stage-3 layer (154 / 103 M output elements: the 32-bit buffer offsets at size)
]


def _tagged(K, fn):
    """fn() with the launch profile on: (result, tags of the launches it made)."""
    K.PROFILE.reset()
    K.PROFILE.enabled = True
    try:
        r = fn()
    finally:
        K.PROFILE.enabled = False
    tags = [rec[0] for rec in K.PROFILE.records]
    K.PROFILE.reset()
    return r, tags


def _reference_codes(xq, zp, s_in, wq, s_w, w_off, bias, relu, q_scale, hi):
    x = (xq.double() - zp) * float(s_in)
    w = wq.double() * s_w.double()[:, None]
    if w_off is not None:
        w = w + w_off.double()[:, None]
    y = x @ w.t()
    if bias is not None:
        y = y + bias.double()
    if relu:
        y = torch.relu(y)
    return torch.clamp(torch.round(y / float(q_scale)), 0, hi)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(int(v)) if not isinstance(v, bool) else "ft"[v] for v in c))
def test_pointwise_kernel_is_the_tiled_kernel_bit_for_bit(case):
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, h, w, c, k, asym, has_bias, relu, signed, zp = case
    g = torch.Generator(device=DEV).manual_seed(n * 1000 + c + k)
    if signed:
        codes = torch.randint(-128, 128, (n, c, h, w), generator=g, device=DEV, dtype=torch.int8)
    else:
        codes = torch.randint(0, 256, (n, c, h, w), generator=g, device=DEV, dtype=torch.uint8)
    codes = codes.contiguous(memory_format=torch.channels_last)
    lo_w, hi_w = (0, 16) if asym else (-127, 128)
    wq = torch.randint(lo_w, hi_w, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 0.004 + 0.0005).contiguous()
    w_off = (torch.randn(k, generator=g, device=DEV) * 0.01).contiguous() if asym else None
    bias = torch.randn(k, generator=g, device=DEV).contiguous() if has_bias else None
    s_in = torch.full((1,), 0.021, device=DEV)
    in_zp = torch.full((1,), zp, device=DEV)
    q_scale = torch.full((1,), 0.043, device=DEV)
    emit = K.EmitCodes(q_scale, None, 0, 255, N.FORM_ZEROPOINT)

    def run(cd):
        _, got = K.conv2d_i8(cd, wq, wsum, bias, s_in, in_zp, s_w, relu=relu, emit=emit, want_out=False, w_offset=w_off)
        return got
    m = n * h * w
    assert m >= 4096
    got, tags = _tagged(K, lambda: run(codes))
    assert tags == ["conv_pw"], tags           # (the tag follows the library's own dispatch rule)
    torch.cuda.synchronize()
    # the tiled kernel: the same layer on the first images only (fewer than 4 096 pixels: conv_pw_applies declines)
    nsub = max(1, 4095 // (h * w))
    sub, tags = _tagged(K, lambda: run(codes[:nsub].contiguous(memory_format=torch.channels_last)))
    assert tags == ["conv_i8"], tags
    assert torch.equal(got[:nsub], sub), f"pointwise kernel differs from the tiled kernel on {int((got[:nsub] != sub).sum())} codes"
    # ... and on the LAST images (the half-empty last block, the last workgroup's tail)
    subl = run(codes[-nsub:].contiguous(memory_format=torch.channels_last))
    assert torch.equal(got[-nsub:], subl)
    # float64 reference: within one code, on very few elements
    xq = codes.permute(0, 2, 3, 1).reshape(m, c)
    want = _reference_codes(xq, zp, s_in, wq.reshape(k, c), s_w, w_off, bias, relu, q_scale, 255)
    diff = (got.permute(0, 2, 3, 1).reshape(m, k).double() - want).abs()
    assert float(diff.max()) <= 1.0 and float((diff > 0).double().mean()) < 1e-3, (float(diff.max()), float((diff > 0).double().mean()))


def test_pointwise_kernel_emits_recentred_codes_like_the_tiled_kernel():
    """DLMCQ_EMIT_SHIFT128 (codes handed to a matrix-core consumer as int8 `code - 128`: what MobileOne's pointwise layers now emit for
    the depthwise layers behind them): the same bytes from both kernels, and `code - 128` of the plain emission."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(77)
    n, h, w, c, k = 12, 20, 20, 192, 192
    codes = torch.randint(0, 256, (n, c, h, w), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wq = torch.randint(0, 16, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 0.004 + 0.0005).contiguous()
    w_off = (torch.randn(k, generator=g, device=DEV) * 0.01).contiguous()
    bias = torch.randn(k, generator=g, device=DEV)
    s_in = torch.full((1,), 0.021, device=DEV)
    q_scale = torch.full((1,), 0.043, device=DEV)

    def run(cd, shifted):
        emit = K.EmitCodes(q_scale, None, 0, 255, N.FORM_ZEROPOINT, shift128=shifted)
        return K.conv2d_i8(cd, wq, wsum, bias, s_in, None, s_w, relu=True, emit=emit, want_out=False, w_offset=w_off)[1]
    plain, shifted = run(codes, False), run(codes, True)
    assert shifted.dtype == torch.int8 and torch.equal(shifted.to(torch.int16) + 128, plain.to(torch.int16))
    sub = run(codes[:9].contiguous(memory_format=torch.channels_last), True)          # 3 600 pixels: the tiled kernel
    assert torch.equal(shifted[:9], sub)


def test_pointwise_kernel_leaves_the_rest_to_the_tiled_kernel():
    """Shapes outside its list (other widths, strides, an fp32 output, a shortcut) take the tiled kernel as before."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(5)
    codes = torch.randint(0, 256, (8, 256, 28, 28), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (128, 1, 1, 256), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = torch.full((128,), 0.002, device=DEV)
    one = torch.full((1,), 0.02, device=DEV)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=DEV), None, 0, 255, N.FORM_ZEROPOINT)

    def both():
        K.conv2d_i8(codes, wq, wsum, None, one, None, s_w, relu=True, emit=emit, want_out=False)           # 256 input channels
        K.conv2d_i8(codes[:, :128].contiguous(memory_format=torch.channels_last), wq[..., :128].contiguous(), wsum, None, one, None, s_w,
                    relu=True, emit=emit, want_out=True)                                                   # fp32 output wanted
    assert _tagged(K, both)[1] == ["conv_i8", "conv_i8"]
