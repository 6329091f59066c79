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
    (32, 14, 14, 1024, 256, False, True, True, True, 0.0),     # ResNet-50's stage-3 reductions (1 024 -> 256): a 128 KB weight slice, six waves per CU
    (24, 14, 14, 1024, 512, False, True, True, False, 0.0),    # ... and the first layer of stage 4 (1 024 -> 512)
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

    def run(cd, force_tiled=False):
        _, got = K.conv2d_i8(cd, wq, wsum, bias, s_in, in_zp, s_w, relu=relu, emit=emit, want_out=False, w_offset=w_off, force_tiled=force_tiled)
        return got
    m = n * h * w
    assert m >= 4096
    got, tags = _tagged(K, lambda: run(codes))
    assert tags == ["conv_pw"], tags           # (the tag is the library's own answer: the call again with DLMCQ_ROUTE_ONLY)
    torch.cuda.synchronize()
    # round 5: the WHOLE tensor through the tiled kernel (DLMCQ_FORCE_TILED) - every block of every workgroup faces the kernel it replaces,
    # at the stated sizes too (1 024 x 28^2 x 192, 1 024 x 14^2 x 512), not only the first and last sub-batch
    whole, tags = _tagged(K, lambda: run(codes, force_tiled=True))
    assert tags == ["conv_i8"], tags
    assert torch.equal(got, whole), f"pointwise kernel differs from the tiled kernel on {int((got != whole).sum())} of {got.numel()} codes"
    del whole
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


@pytest.mark.parametrize("c,k", [(128, 64), (192, 64), (512, 64), (1024, 64), (512, 192), (1024, 192)], ids=str)
def test_pointwise_pairs_without_an_instantiation_take_the_tiled_kernel(c, k):
    """conv_pw_applies admitted (C, K) pairs conv_pw_launch has no instantiation for - a codes-to-codes 1x1 layer such as
    128 -> 64 or 512 -> 192 with the plain quantiser and >= 4 096 pixels then failed with DLMCQ_EINVAL instead of running on the tiled kernel.
    The predicate and the launch table are one list now: these pairs route to the tiled kernel and give the float64 reference's codes."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(c + k)
    n, h, w = 6, 28, 28                                   # 4 704 pixels
    codes = torch.randint(0, 256, (n, c, h, w), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 0.0004 + 0.00005).contiguous()
    bias = torch.randn(k, generator=g, device=DEV).contiguous()
    s_in, q_scale = torch.full((1,), 0.021, device=DEV), torch.full((1,), 0.043, device=DEV)
    emit = K.EmitCodes(q_scale, None, 0, 255, N.FORM_ZEROPOINT)
    (_, got), tags = _tagged(K, lambda: K.conv2d_i8(codes, wq, wsum, bias, s_in, None, s_w, relu=True, emit=emit, want_out=False))
    assert tags == ["conv_i8"], tags
    (_, forced), _ = _tagged(K, lambda: K.conv2d_i8(codes, wq, wsum, bias, s_in, None, s_w, relu=True, emit=emit, want_out=False, force_tiled=True))
    assert torch.equal(got, forced)
    m = n * h * w
    want = _reference_codes(codes.permute(0, 2, 3, 1).reshape(m, c), 0.0, s_in, wq.reshape(k, c), s_w, None, bias, True, q_scale, 255)
    diff = (got.permute(0, 2, 3, 1).reshape(m, k).double() - want).abs()
    assert float(diff.max()) <= 1.0 and float((diff > 0).double().mean()) < 1e-3


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


# csrc/conv_pwr_i8.hip: (N, H, W, C, K, fp32 output stored as well, signed input codes, input zero point, codes emitted as `code - 128`)
RES_CASES = [
    (32, 14, 14, 256, 1024, False, True, 0.0, False),     # ResNet-50 stage 3's last block end (codes only: the next block's shortcut is a convolution)
    (32, 14, 14, 256, 1024, True, False, 0.0, False),
    (128, 7, 7, 512, 2048, True, True, 0.0, True),        # stage 4 (fp32 block output + codes), codes handed on re-centred
    (96, 7, 7, 512, 2048, False, False, 5.0, False),      # 4 704 pixels = 147 blocks, a zero point
    (24, 14, 14, 256, 128, True, True, 0.0, False),       # one slice; 147 blocks: fewer than the waves of its workgroups
    (64, 8, 8, 256, 4096, True, True, 0.0, False),        # the smallest pixel count (4 096) against the widest layer the kernel takes: 32 slices
    (16, 16, 16, 512, 128, False, False, 2.0, True),      # 4 096 pixels, one slice, a zero point, re-centred emission
    (512, 14, 14, 256, 1024, False, True, 0.0, False),    # BASELINE configs[2] at its stated size: the two shapes of the ResNet-50 plan at batch 512
    (512, 7, 7, 512, 2048, True, True, 0.0, False),
]


@pytest.mark.parametrize("case", RES_CASES, ids=lambda c: "x".join(str(int(v)) if not isinstance(v, bool) else "ft"[v] for v in c))
def test_residual_block_end_kernel_is_the_tiled_kernel_bit_for_bit(case):
    """The 1x1 block end with an fp32 shortcut (+ fp32 output) + ReLU + plain codes on the weight-resident kernel: the same bytes -
    codes AND fp32 values - as the tiled kernel's residual epilogue, and a float64 reference within one code."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, h, w, c, k, want_out, signed, zp, shifted = case
    g = torch.Generator(device=DEV).manual_seed(n * 1000 + c + k + int(want_out))
    if signed:
        codes = torch.randint(-128, 128, (n, c, h, w), generator=g, device=DEV, dtype=torch.int8)
    else:
        codes = torch.randint(0, 256, (n, c, h, w), generator=g, device=DEV, dtype=torch.uint8)
    codes = codes.contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 0.0004 + 0.00005).contiguous()
    bias = torch.randn(k, generator=g, device=DEV).contiguous()
    short = (torch.randn((n, k, h, w), generator=g, device=DEV) * 2.0).contiguous(memory_format=torch.channels_last)
    short[0, :8, 0, 0] = torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1e30, -1e30, 0.0215], device=DEV)
    s_in = torch.full((1,), 0.021, device=DEV)
    in_zp = torch.full((1,), zp, device=DEV)
    q_scale = torch.full((1,), 0.043, device=DEV)
    emit = K.EmitCodes(q_scale, None, 0, 255, N.FORM_ZEROPOINT, shift128=shifted)

    def run(cd, sc):
        return K.conv2d_i8(cd, wq, wsum, bias, s_in, in_zp, s_w, residual=sc, relu=True, emit=emit, want_out=want_out)
    (out, got), tags = _tagged(K, lambda: run(codes, short))
    assert tags == ["conv_pwr"], tags
    torch.cuda.synchronize()
    # round 5: the whole tensor through the tiled kernel (DLMCQ_FORCE_TILED): codes and fp32 bits, every pixel block of every slice
    (ow, gw), tags = _tagged(K, lambda: K.conv2d_i8(codes, wq, wsum, bias, s_in, in_zp, s_w, residual=short, relu=True, emit=emit, want_out=want_out,
                                                  force_tiled=True))
    assert tags == ["conv_i8"], tags
    assert torch.equal(got, gw), f"{int((got != gw).sum())} of {got.numel()} codes differ from the tiled kernel"
    if want_out:
        assert torch.equal(out.view(torch.int32), ow.view(torch.int32)), "fp32 block output differs from the tiled kernel (bitwise, whole tensor)"
    del ow, gw
    # round 5: the fp32 shortcut (and output) chunk-major - K.ChunkMajor, DLMCQ_FP32_IN / OUT_CHUNK_MAJOR: the same values in [K / 64][M][64] planes
    (oc, gc), tags = _tagged(K, lambda: K.conv2d_i8(codes, wq, wsum, bias, s_in, in_zp, s_w, residual=K.ChunkMajor.from_nhwc(short), relu=True,
                                                  emit=emit, want_out=want_out, out_chunk_major=True))
    assert tags == ["conv_pwr"], tags
    assert torch.equal(gc, got), "codes differ with a chunk-major shortcut"
    if want_out:
        assert isinstance(oc, K.ChunkMajor) and torch.equal(oc.to_nhwc().view(torch.int32), out.view(torch.int32)), "chunk-major fp32 output differs"
    del oc, gc
    nsub = max(1, 4095 // (h * w))
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for sl in (slice(0, nsub), slice(n - nsub, n)):        # the tiled kernel on the first and on the last images (fewer than 4 096 pixels)
        (osub, csub), tags = _tagged(K, lambda: run(cl(codes[sl]), cl(short[sl])))
        assert tags == ["conv_i8"], tags
        assert torch.equal(got[sl], csub), f"{int((got[sl] != csub).sum())} codes differ from the tiled kernel"
        if want_out:
            assert torch.equal(out[sl].view(torch.int32), osub.view(torch.int32)), "fp32 block output differs from the tiled kernel (bitwise)"
    m = n * h * w
    if m <= 8192:
        x = (codes.permute(0, 2, 3, 1).reshape(m, c).double() - zp) * 0.021
        y = x @ (wq.reshape(k, c).double() * s_w.double()[:, None]).t() + bias.double() + short.permute(0, 2, 3, 1).reshape(m, k).double()
        y = torch.relu(y)
        want = torch.clamp(torch.round(y / 0.043), 0, 255)
        ok = torch.isfinite(y)
        gq = got.permute(0, 2, 3, 1).reshape(m, k).double() + (128 if shifted else 0)
        diff = (gq - want).abs()[ok]
        assert float(diff.max()) <= 1.0 and float((diff > 0).double().mean()) < 1e-3, (float(diff.max()), float((diff > 0).double().mean()))
        if want_out:
            o = out.permute(0, 2, 3, 1).reshape(m, k).double()
            assert float((o - y).abs()[ok].max()) < 1e-3


def test_chunk_major_shortcut_where_the_tiled_kernel_takes_the_call():
    """A K.ChunkMajor shortcut handed to a call the block-end kernel does not take (fewer than 4 096 pixels): the wrapper converts it, the
    result is the ordinary call's; the C entry point itself refuses the layout bits there (DLMCQ_EINVAL) instead of reading row-major."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(9)
    n, c, k, h = 8, 256, 128, 14
    codes = torch.randint(0, 256, (n, c, h, h), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = torch.full((k,), 0.0002, device=DEV)
    one = torch.full((1,), 0.02, device=DEV)
    short = torch.randn((n, k, h, h), generator=g, device=DEV).contiguous(memory_format=torch.channels_last)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=DEV), None, 0, 255, N.FORM_ZEROPOINT)
    (o0, c0), t0 = _tagged(K, lambda: K.conv2d_i8(codes, wq, wsum, None, one, None, s_w, residual=short, relu=True, emit=emit, want_out=True))
    (o1, c1), t1 = _tagged(K, lambda: K.conv2d_i8(codes, wq, wsum, None, one, None, s_w, residual=K.ChunkMajor.from_nhwc(short), relu=True, emit=emit,
                                                want_out=True, out_chunk_major=True))
    assert t0 == t1 == ["conv_i8"] and isinstance(o1, torch.Tensor)
    assert torch.equal(o0.view(torch.int32), o1.view(torch.int32)) and torch.equal(c0, c1)
    out = torch.empty_like(o0)
    cd = torch.empty_like(c0)
    rc = N.lib.dlmcq_conv2d_i8_nhwc_fused(N.ptr(codes), N.ptr(wq), N.ptr(out), None, N.ptr(wsum), N.ptr(one), None, N.ptr(s_w), n, h, h, c, k, 1, 1, 1, 0, 1,
                                          1, N.ptr(short), 1, N.ptr(cd), N.ptr(emit.scale), None, 0, 255, emit.form | N.FP32_IN_CHUNK_MAJOR, 0.0,
                                          N.stream_ptr())
    assert rc == -1        # DLMCQ_EINVAL


def test_residual_block_end_kernel_declines_what_it_was_not_built_for():
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(6)
    codes = torch.randint(0, 256, (23, 256, 14, 14), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)   # 4 508 pixels: not a multiple of 32
    wq = torch.randint(-127, 128, (128, 1, 1, 256), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = torch.full((128,), 0.0002, device=DEV)
    one = torch.full((1,), 0.02, device=DEV)
    short = torch.randn((23, 128, 14, 14), generator=g, device=DEV).contiguous(memory_format=torch.channels_last)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=DEV), None, 0, 255, N.FORM_ZEROPOINT)
    emit_z = K.EmitCodes(torch.full((1,), 0.05, device=DEV), torch.full((1,), 3.0, device=DEV), 0, 255, N.FORM_ZEROPOINT)

    def three():
        K.conv2d_i8(codes, wq, wsum, None, one, None, s_w, residual=short, relu=True, emit=emit, want_out=False)          # ragged block count
        K.conv2d_i8(codes[:22].contiguous(memory_format=torch.channels_last), wq, wsum, None, one, None, s_w,
                    residual=short[:22].contiguous(memory_format=torch.channels_last), relu=True, emit=emit_z, want_out=False)   # a zero point
        K.conv2d_i8(codes[:22].contiguous(memory_format=torch.channels_last), wq, wsum, None, one, None, s_w,
                    residual=short[:22].contiguous(memory_format=torch.channels_last), relu=False, emit=emit, want_out=False)    # no ReLU
    assert _tagged(K, three)[1] == ["conv_i8"] * 3


@pytest.mark.parametrize("shape", [(128, 7, 7, 512, 2048), (32, 14, 14, 256, 384), (512, 7, 7, 512, 2048)], ids=str)
def test_residual_block_end_kernel_without_codes(shape):
    """A network's last block (fp32 output alone: nothing quantises the pooled features' producer): same fp32 bits as the tiled kernel."""
    from dlmc.quantization.scalar import kernels as K
    n, h, w, c, k = shape
    g = torch.Generator(device=DEV).manual_seed(n + c + k)
    codes = torch.randint(-128, 128, (n, c, h, w), generator=g, device=DEV, dtype=torch.int8).contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 0.0004 + 0.00005).contiguous()
    bias = torch.randn(k, generator=g, device=DEV).contiguous()
    short = (torch.randn((n, k, h, w), generator=g, device=DEV) * 2.0).contiguous(memory_format=torch.channels_last)
    short[0, :4, 0, 0] = torch.tensor([float("nan"), float("inf"), -float("inf"), -0.0], device=DEV)
    s_in = torch.full((1,), 0.021, device=DEV)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)

    def run(cd, sc):
        return K.conv2d_i8(cd, wq, wsum, bias, s_in, None, s_w, residual=sc, relu=True)
    out, tags = _tagged(K, lambda: run(codes, short))
    assert tags == ["conv_pwr"], tags
    nsub = max(1, 4095 // (h * w))
    for sl in (slice(0, nsub), slice(n - nsub, n)):
        osub, tags = _tagged(K, lambda: run(cl(codes[sl]), cl(short[sl])))
        assert tags == ["conv_i8"], tags
        assert torch.equal(out[sl].view(torch.int32), osub.view(torch.int32))


# the dual form: (N, output H = W, stride of the 512-channel pair, K, which pair is the call's first: "dense" (256 channels, row by row) or "sampled")
DUAL_CASES = [
    (32, 14, 2, 1024, "sampled"),      # ResNet-50 stage 3's first block as the plan calls it (shortcut convolution first)
    (32, 14, 2, 1024, "dense"),
    (24, 14, 1, 256, "dense"),         # an unstrided shortcut convolution, two slices
    (512, 14, 2, 1024, "sampled"),     # BASELINE configs[2] at its stated size
]


@pytest.mark.parametrize("case", DUAL_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_dual_block_end_kernel_is_the_tiled_kernel_bit_for_bit(case):
    """conv1x1(256 ch) + conv1x1(512 ch, strided) + ReLU -> fp32 + plain codes with both weight slices resident in LDS: the same
    fp32 bits and codes as the tiled kernel's DUAL instantiation (sub-batches below 4 096 pixels), float64 reference within one code."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, hw, st, k, first = case
    g = torch.Generator(device=DEV).manual_seed(n + hw + st + k)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    xd = cl(torch.randint(-128, 128, (n, 256, hw, hw), generator=g, device=DEV, dtype=torch.int8))
    xs = cl(torch.randint(0, 256, (n, 512, hw * st, hw * st), generator=g, device=DEV, dtype=torch.uint8))

    def operand(x, c, stride, seed):
        gg = torch.Generator(device=DEV).manual_seed(seed)
        wq = torch.randint(-127, 128, (k, 1, 1, c), generator=gg, device=DEV, dtype=torch.int8)
        return dict(codes=x, wq=wq, wsum=wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(),
                    bias=torch.randn(k, generator=gg, device=DEV), in_scale=torch.full((1,), 0.017 + 0.001 * seed, device=DEV),
                    in_zp=torch.full((1,), float(seed), device=DEV) if x.dtype == torch.uint8 else None,
                    w_scale=(torch.rand(k, generator=gg, device=DEV) * 0.0004 + 0.00005).contiguous(), stride=stride)
    dense, sampled = operand(xd, 256, 1, 3), operand(xs, 512, st, 5)
    emit = K.EmitCodes(torch.full((1,), 0.043, device=DEV), None, 0, 255, N.FORM_ZEROPOINT)

    def run(sl):
        d, s_ = dict(dense, codes=cl(xd[sl])), dict(sampled, codes=cl(xs[sl]))
        a, b = (d, s_) if first == "dense" else (s_, d)
        return K.conv2d_i8_dual(a, b, relu=True, emit=emit, want_out=True, out_chunk_major=cmaj)
    cmaj = False
    (out, got), tags = _tagged(K, lambda: run(slice(0, n)))
    assert tags == ["conv_pwr"], tags
    cmaj = True            # round 5: the block output chunk-major (K.ChunkMajor): same values
    (oc, gc), tags = _tagged(K, lambda: run(slice(0, n)))
    assert tags == ["conv_pwr"] and isinstance(oc, K.ChunkMajor) and torch.equal(gc, got) and torch.equal(oc.to_nhwc().view(torch.int32), out.view(torch.int32))
    del oc, gc
    nsub = max(1, 4095 // (hw * hw))
    (osub, csub), tags = _tagged(K, lambda: run(slice(0, nsub)))       # (where the tiled kernel takes the call the output stays an ordinary tensor)
    assert tags == ["conv_i8"] and isinstance(osub, torch.Tensor)
    cmaj = False
    for sl in (slice(0, nsub), slice(n - nsub, n)):
        (osub, csub), tags = _tagged(K, lambda: run(sl))
        assert tags == ["conv_i8"], tags
        assert torch.equal(got[sl], csub), f"{int((got[sl] != csub).sum())} codes differ from the tiled kernel"
        assert torch.equal(out[sl].view(torch.int32), osub.view(torch.int32)), "fp32 output differs from the tiled kernel (bitwise)"
    m = n * hw * hw
    if m <= 8192:
        def real(t, x):
            zp = 0.0 if t["in_zp"] is None else float(t["in_zp"])
            xx = x[:, :, ::t["stride"], ::t["stride"]].permute(0, 2, 3, 1).reshape(m, -1).double()
            return ((xx - zp) * float(t["in_scale"])) @ (t["wq"].reshape(k, -1).double() * t["w_scale"].double()[:, None]).t() + t["bias"].double()
        y = torch.relu(real(dense, xd) + real(sampled, xs))
        want = torch.clamp(torch.round(y / 0.043), 0, 255)
        diff = (got.permute(0, 2, 3, 1).reshape(m, k).double() - want).abs()
        assert float(diff.max()) <= 1.0 and float((diff > 0).double().mean()) < 1e-3, (float(diff.max()), float((diff > 0).double().mean()))
        assert float((out.permute(0, 2, 3, 1).reshape(m, k).double() - y).abs().max()) < 1e-3
