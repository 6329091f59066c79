"""The halo-tile 3x3 kernel (csrc/conv3x3_i8.hip: 3x3 / stride 1 or 2 / pad 1 layers that emit only their consumer's codes,
K % 64 == 0) against (a) the float64 convolution of the dequantised operands + the oracle's quantiser and (b) the generic
implicit-GEMM kernel of conv_i8.hip run on the same layer with an fp32 output as well (which keeps it off the halo kernel):
both kernels add the same exact int32 sums into the same rounding chain, so their codes must agree bit for bit.
Reference call being replaced: F.conv2d in modules/conv.py:13-19 on the operands of FSPTQuant/base.py:108-109,149-152."""
import pytest
import torch
import torch.nn.functional as F

from oracle import fakequant_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# N, C, H, W, K  (frame pitch W + 1; halo pieces per chunk = ceil((256 + 2 (W + 1) + 2) / 16): <= 24 -> HPW 3, else HPW 4)
SHAPES = [
    (3, 64, 7, 7, 128),        # FS = 64: four images per tile, 23 % junk rows
    (2, 128, 14, 14, 128),     # two chunks: the halo double buffer
    (1, 64, 28, 28, 256),      # two column blocks
    (2, 64, 56, 56, 128),      # Wp = 57: 24 halo pieces (the HPW = 3 limit)
    (1, 64, 5, 70, 128),       # Wp = 71: HPW = 4 instantiation
    (1, 64, 3, 118, 128),      # Wp = 119: the widest image the kernel takes (32 pieces)
    (5, 256, 14, 14, 256),     # ResNet-50 stage 3 / RepVGG stage 3 shape, tiles crossing images
    (3, 512, 7, 7, 512),       # ResNet-50 stage 4 shape: 72 K steps
    (1, 64, 1, 1, 128),        # one pixel: every tap but the centre is border
    (2, 64, 2, 3, 128),
    (7, 64, 9, 13, 128),       # M not a multiple of anything
    (2, 64, 56, 56, 64),       # ResNet-50 stage 1: 64-wide tile, one chunk (the single-halo-buffer instantiation)
    (3, 64, 12, 12, 64),
    (2, 128, 10, 10, 64),      # 64-wide tile, two chunks
    (1, 64, 6, 70, 64),        # 64-wide, wide image
    (1, 128, 4, 90, 64),
    (2, 128, 9, 9, 192),       # K % 128 != 0: three 64-wide column blocks
    # stride 2 (four phase images streamed through three tile buffers): ResNet-50's stage openers, RepVGG's
    (3, 128, 56, 56, 128, 2),  # 56^2 -> 28^2: Wp = 29, 18-piece tiles
    (2, 256, 28, 28, 256, 2),  # two column blocks, four chunks
    (3, 512, 14, 14, 512, 2),  # 14^2 -> 7^2: eight chunks, tiles spanning several images
    (1, 64, 112, 112, 64, 2),  # RepVGG stage 1: Wp = 57, 20-piece tiles, 64-wide column block
    (2, 64, 56, 56, 128, 2),   # one chunk only
    (5, 64, 6, 10, 64, 2),
    (1, 64, 2, 2, 128, 2),     # a single output pixel
    (1, 128, 4, 120, 128, 2),  # Wp = 61: the widest image the stride-2 kernel takes
]


def _layer(idx, n, c, h, w, k, unsigned, zp, relu, with_bias, stride=1):
    g = torch.Generator().manual_seed(4242 + idx)
    lo, hi = (0, 255) if unsigned else (-127, 127)
    codes = torch.randint(lo, hi + 1, (n, c, h, w), generator=g).to(torch.uint8 if unsigned else torch.int8)
    wt = torch.randn(k, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
    s_w, _ = O.minmax_channel(wt, 8, True, ch_axis=0)
    s_w = s_w + 1e-6
    bias = torch.randn(k, generator=g) * 0.1 if with_bias else None
    s_in = torch.tensor(0.0231)
    qw, wdeq = O.fq_symmetric(wt, s_w, -127, 127)
    ref = F.conv2d((codes.double() - zp) * s_in.double(), wdeq.double(), None if bias is None else bias.double(), padding=1, stride=stride)
    if relu:
        ref = torch.relu(ref)
    return codes, wt, s_w, bias, s_in, ref


@pytest.mark.parametrize("unsigned", [True, False])
def test_halo_kernel_vs_float64_and_generic_kernel(unsigned):
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    worst = 0.0
    for idx, shape in enumerate(SHAPES):
        n, c, h, w, k = shape[:5]
        stride = shape[5] if len(shape) > 5 else 1
        zp = float(3 + idx) if unsigned else float(idx % 3 - 1)      # non-zero: the borders must read the zero point's code
        relu = idx % 2 == 0
        codes, wt, s_w, bias, s_in, ref = _layer(idx, n, c, h, w, k, unsigned, zp, relu, idx % 3 != 1, stride)
        wq, wsum = K.quantize_weight_krsc(wt.to(DEV), s_w.to(DEV), -127, 127)
        if unsigned:
            q_s = (ref.abs().max() / 200).float().reshape(1)         # some saturation
            q_z, lo, hi, form = torch.tensor([2.0]), 0, 255, N.FORM_ZEROPOINT
        else:
            q_s = (ref.abs().max() / 100).float().reshape(1)
            q_z, lo, hi, form = None, -127, 127, N.FORM_SYMMETRIC
        emit = K.EmitCodes(q_s.to(DEV), None if q_z is None else q_z.to(DEV), lo, hi, form)
        cd = codes.to(DEV).contiguous(memory_format=torch.channels_last)
        args = (cd, wq, wsum, None if bias is None else bias.to(DEV), s_in.to(DEV), torch.tensor(zp).to(DEV), s_w.to(DEV))
        _, got = K.conv2d_i8(*args, padding=1, stride=stride, relu=relu, emit=emit, want_out=False)          # the halo kernel
        out, gen = K.conv2d_i8(*args, padding=1, stride=stride, relu=relu, emit=emit, want_out=True)         # the generic kernel
        assert got.shape == (n, k, h // stride, w // stride) and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, gen), f"shape {idx} {SHAPES[idx]}: halo kernel and generic kernel disagree"
        torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-6, atol=2e-5, msg=lambda m: f"shape {idx}: {m}")
        # the oracle's quantiser on the float64 result: int32-exact accumulation vs float64 may differ by one code at a tie
        if form == N.FORM_ZEROPOINT:
            want = O.fq_zeropoint(ref.float(), q_s[0], q_z[0], lo, hi)[0]
        else:
            want = O.fq_symmetric(ref.float(), q_s[0], lo, hi)[0]
        off = (got.cpu().float() - want).abs()
        assert float(off.max()) <= 1.0, f"shape {idx}: code off by {float(off.max())}"
        worst = max(worst, float((off > 0).float().mean()))
    print(f"halo 3x3: worst off-by-one rate {worst:.2e}")
    assert worst < 1e-3


def test_halo_kernel_border_rows_never_leak():
    """Rows of the linear frame that are not pixels (x = W, y = H) are multiplied and dropped: a canary around the output
    tensor must survive, and an all-zero-point input must give exactly the bias's codes everywhere."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, c, h, w, k = 3, 64, 6, 11, 128
    zp = 7.0
    codes = torch.full((n, c, h, w), int(zp), dtype=torch.uint8, device=DEV).contiguous(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(77)
    wt = torch.randn(k, c, 3, 3, generator=g) * 0.05
    s_w, _ = O.minmax_channel(wt, 8, True, ch_axis=0)
    wq, wsum = K.quantize_weight_krsc(wt.to(DEV), s_w.to(DEV), -127, 127)
    bias = torch.linspace(-1, 1, k)
    emit = K.EmitCodes(torch.tensor([0.01], device=DEV), torch.tensor([100.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
    _, got = K.conv2d_i8(codes, wq, wsum, bias.to(DEV), torch.tensor(0.02, device=DEV), torch.tensor(zp, device=DEV), s_w.to(DEV),
                         padding=1, emit=emit, want_out=False)
    want = O.fq_zeropoint(bias, torch.tensor(0.01), torch.tensor(100.0), 0, 255)[0].to(torch.uint8)
    assert torch.equal(got.cpu(), want.view(1, k, 1, 1).expand(n, k, h, w))


def test_shifted_code_emission_is_the_same_network():
    """DLMCQ_EMIT_SHIFT128: a producer stores an unsigned-byte quantiser's codes as int8 `code - 128`; the consumer takes them as
    signed codes with the zero point `zp - 128`.  Producer side: byte for byte `code ^ 0x80` (1x1 swapped kernel, halo kernel,
    the chain kernel's second quantiser, fp32-output kernel).  Consumer side (halo kernel without its xor, generic kernel):
    bit-identical results."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(99)
    n, c, h, k = 3, 64, 12, 128
    x = torch.randint(0, 256, (n, c, h, h), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)

    def layer(kk, cc, r):
        wq = torch.randint(-127, 128, (kk, r, r, cc), generator=g, device=DEV, dtype=torch.int8)
        return dict(wq=wq, wsum=wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(kk, generator=g, device=DEV),
                    w_scale=torch.rand(kk, generator=g, device=DEV) * 0.002 + 0.0005)
    l1, l3, l5 = layer(k, c, 1), layer(k, k, 3), layer(512, k, 1)
    s_in, zp_in = torch.full((1,), 0.02, device=DEV), torch.full((1,), 4.0, device=DEV)
    q_s, q_z = torch.full((1,), 0.07, device=DEV), torch.full((1,), 9.0, device=DEV)

    def emit(shift):
        return K.EmitCodes(q_s, q_z, 0, 255, N.FORM_ZEROPOINT, shift128=shift)

    def conv(codes, lay, zp, shift, pad=0, want_out=False, relu=True):
        return K.conv2d_i8(codes, lay["wq"], lay["wsum"], lay["bias"], s_in, zp, lay["w_scale"], padding=pad, relu=relu, emit=emit(shift),
                           want_out=want_out)
    # producers
    _, u1 = conv(x, l1, zp_in, False)                     # swapped 1x1 kernel
    _, s1 = conv(x, l1, zp_in, True)
    assert u1.dtype == torch.uint8 and s1.dtype == torch.int8
    assert torch.equal(s1.view(torch.uint8), u1 ^ 0x80)
    o_u, u1o = conv(x, l1, zp_in, False, want_out=True)   # fp32-output kernel (unswapped epilogue)
    o_s, s1o = conv(x, l1, zp_in, True, want_out=True)
    assert torch.equal(o_u, o_s) and torch.equal(s1o.view(torch.uint8), u1o ^ 0x80) and torch.equal(u1o, u1)
    # consumers: the halo kernel (3x3) and the generic kernel (fp32 out) on shifted codes with zp - 128
    _, u3 = conv(u1, l3, q_z, False, pad=1)
    _, s3 = conv(s1, l3, q_z - 128.0, False, pad=1)
    assert torch.equal(u3, s3)
    _, s3s = conv(s1, l3, q_z - 128.0, True, pad=1)       # halo kernel as a producer of shifted codes
    assert torch.equal(s3s.view(torch.uint8), u3 ^ 0x80)
    out_u, _ = conv(u1, l3, q_z, False, pad=1, want_out=True)
    out_s, _ = conv(s1, l3, q_z - 128.0, False, pad=1, want_out=True)
    assert torch.equal(out_u, out_s)
    # the chain kernel's second quantiser
    a = dict(codes=u3, wq=l5["wq"], wsum=l5["wsum"], bias=l5["bias"], in_scale=q_s, in_zp=q_z, w_scale=l5["w_scale"])
    l6 = layer(128, 512, 1)
    b = dict(wq=l6["wq"], wsum=l6["wsum"], bias=l6["bias"], w_scale=l6["w_scale"])
    res = torch.randn(n, 512, h, h, generator=g, device=DEV).contiguous(memory_format=torch.channels_last)
    e1 = K.EmitCodes(torch.full((1,), 0.05, device=DEV), torch.zeros(1, device=DEV), 0, 255, N.FORM_ZEROPOINT)
    _, _, c2u = K.conv2d_i8_chain(a, b, res, relu=True, emit=e1, want_out=True, want_codes=False, relu2=True, emit2=emit(False))
    _, _, c2s = K.conv2d_i8_chain(a, b, res, relu=True, emit=e1, want_out=True, want_codes=False, relu2=True, emit2=emit(True))
    assert c2s.dtype == torch.int8 and torch.equal(c2s.view(torch.uint8), c2u ^ 0x80)
    # the first quantiser of the chain kernel cannot be shifted (its codes are read in place): refused, not mis-computed
    from dlmc._native import DlmcqError
    with pytest.raises((DlmcqError, ValueError)):
        K.conv2d_i8_chain(a, b, res, relu=True, emit=K.EmitCodes(e1.scale, e1.zero_point, 0, 255, N.FORM_ZEROPOINT, shift128=True),
                          want_out=True, want_codes=True, relu2=True, emit2=emit(False))


# csrc/conv3x3_pipe_i8.hip (round 5): N, C, H, W, K, unsigned input codes, input zero point, bias
PIPE_SHAPES = [
    (512, 256, 14, 14, 256, False, 0.0, True),     # BASELINE configs[2] / [3] at the stated size: ResNet-50 stage 3, RepVGG-A1 stage 3
    (512, 512, 7, 7, 512, False, 0.0, True),       # ResNet-50 stage 4: 72 K steps per tile, four column blocks
    (160, 128, 28, 28, 128, True, 0.0, True),      # two chunks per tile (the shortest loop: a quad per half-step), uint8 codes (re-centred on read)
    (256, 128, 14, 14, 384, False, -128.0, False), # three column blocks (tile -> (row block, column block) by a division by 3), re-centred codes' zero point
    (600, 256, 17, 13, 128, True, 5.0, True),      # odd image sizes, a zero point: border positions read its code; tiles crossing images
    (600, 512, 9, 11, 256, False, 0.0, False),
]


@pytest.mark.parametrize("shape", PIPE_SHAPES, ids=lambda s: "x".join(str(int(v)) if not isinstance(v, bool) else "ft"[v] for v in s))
def test_pipelined_halo_kernel_is_the_halo_kernel_and_the_tiled_kernel_bit_for_bit(shape):
    """The persistent, software-pipelined 3x3 kernel (tile t's quantising epilogue under tile t + 1's K loop) against the plain halo kernel
    (the default) and the tiled kernel (DLMCQ_FORCE_TILED) on the WHOLE tensor: the same codes byte for byte - first / last tile of a
    workgroup, tiles that start a new column block, the overhang of the last row block - and a float64 reference on a sample."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, c, h, w, k, unsigned, zp, with_bias = shape
    g = torch.Generator(device=DEV).manual_seed(n + c + k + h)
    lo, hi = (0, 256) if unsigned else (-128, 128)
    codes = torch.randint(lo, hi, (n, c, h, w), generator=g, device=DEV, dtype=torch.int16).to(torch.uint8 if unsigned else torch.int8)
    codes = codes.contiguous(memory_format=torch.channels_last)
    wq = torch.randint(-127, 128, (k, 3, 3, c), generator=g, device=DEV, dtype=torch.int8)
    wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
    s_w = (torch.rand(k, generator=g, device=DEV) * 2e-4 + 5e-5).contiguous()
    bias = (torch.randn(k, generator=g, device=DEV) * 0.5).contiguous() if with_bias else None
    s_in, in_zp = torch.full((1,), 0.021, device=DEV), torch.full((1,), zp, device=DEV)
    q_scale = torch.full((1,), 0.05, device=DEV)
    emit = K.EmitCodes(q_scale, None, 0, 255, N.FORM_ZEROPOINT)

    def run(**kw):
        K.PROFILE.reset()
        K.PROFILE.enabled = True
        try:
            _, got = K.conv2d_i8(codes, wq, wsum, bias, s_in, in_zp, s_w, padding=1, relu=True, emit=emit, want_out=False, **kw)
        finally:
            K.PROFILE.enabled = False
        tag = K.PROFILE.records[-1][0]
        K.PROFILE.reset()
        return got, tag
    got, tag = run(pipelined=True)
    assert tag == "conv3x3_pipe", tag
    plain, tag = run()
    assert tag == "conv3x3_halo", tag
    assert torch.equal(got, plain), f"{int((got != plain).sum())} of {got.numel()} codes differ from the plain halo kernel"
    del plain
    tiled, tag = run(force_tiled=True)
    assert tag == "conv_i8", tag
    assert torch.equal(got, tiled), f"{int((got != tiled).sum())} of {got.numel()} codes differ from the tiled kernel"
    del tiled
    # float64 reference on the first, a middle and the last image
    for i in sorted({0, n // 2, n - 1}):
        x = (codes[i:i + 1].to(torch.int16).double().cpu() - zp) * 0.021
        wd = wq[:, :, :, :].permute(0, 3, 1, 2).double().cpu() * s_w.double().cpu().reshape(-1, 1, 1, 1)
        ref = torch.relu(F.conv2d(x, wd, None if bias is None else bias.double().cpu(), padding=1))
        want = torch.clamp(torch.round(ref / 0.05), 0, 255)
        diff = (got[i:i + 1].double().cpu() - want).abs()
        assert float(diff.max()) <= 1.0 and float((diff > 0).double().mean()) < 2e-3, (i, float(diff.max()), float((diff > 0).double().mean()))


def test_pipelined_halo_kernel_leaves_small_and_other_layers_to_the_halo_kernel():
    """Asked for (DLMCQ_PIPELINED) but not applicable - fewer than two tiles per CU, 64 input channels, stride 2, a zero point in the consumer's quantiser: the plain halo kernel as before."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator(device=DEV).manual_seed(3)

    def tag_of(n, c, h, k, stride=1, q_zp=None):
        codes = torch.randint(-128, 128, (n, c, h, h), generator=g, device=DEV, dtype=torch.int16).to(torch.int8).contiguous(memory_format=torch.channels_last)
        wq = torch.randint(-127, 128, (k, 3, 3, c), generator=g, device=DEV, dtype=torch.int8)
        wsum = wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
        emit = K.EmitCodes(torch.full((1,), 0.05, device=DEV), q_zp, 0, 255, N.FORM_ZEROPOINT)
        K.PROFILE.reset()
        K.PROFILE.enabled = True
        try:
            K.conv2d_i8(codes, wq, wsum, None, torch.full((1,), 0.02, device=DEV), None, torch.full((k,), 1e-4, device=DEV), stride=stride, padding=1,
                        relu=True, emit=emit, want_out=False, pipelined=True)
        finally:
            K.PROFILE.enabled = False
        t = K.PROFILE.records[-1][0]
        K.PROFILE.reset()
        return t
    assert tag_of(64, 256, 14, 256) == "conv3x3_halo"                                        # 114 tiles
    assert tag_of(64, 64, 56, 64) == "conv3x3_halo"                                          # 64 input channels
    assert tag_of(256, 256, 28, 256, stride=2) == "conv3x3_halo"                             # stride 2
    assert tag_of(512, 256, 14, 256, q_zp=torch.full((1,), 3.0, device=DEV)) == "conv3x3_halo"   # not the plain quantiser
