"""dlmcq_conv2d_i8_nhwc_chain (block end + next block's 1x1 reduction in one kernel) against the two separate
dlmcq_conv2d_i8_nhwc_fused calls it replaces: fp32 output, intermediate codes and final codes bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(n, h, c, k, k2, seed, zp1=3.0, zp2=0.0):
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randint(0, 256, (n, c, h, h), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    w1 = torch.randint(-127, 128, (k, 1, 1, c), generator=g, device=dev, dtype=torch.int8)
    w2 = torch.randint(-127, 128, (k2, 1, 1, k), generator=g, device=dev, dtype=torch.int8)
    a = dict(codes=x, wq=w1, wsum=w1.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(),
             bias=torch.randn(k, generator=g, device=dev), in_scale=torch.full((1,), 0.02, device=dev),
             in_zp=torch.full((1,), zp1, device=dev), w_scale=(torch.rand(k, generator=g, device=dev) * 0.004 + 0.001))
    b = dict(wq=w2, wsum=w2.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(),
             bias=torch.randn(k2, generator=g, device=dev), w_scale=(torch.rand(k2, generator=g, device=dev) * 0.002 + 0.0005))
    res = (torch.randn(n, k, h, h, generator=g, device=dev) * 2).contiguous(memory_format=torch.channels_last)
    emit = K.EmitCodes(torch.full((1,), 0.05, device=dev), torch.full((1,), zp2, device=dev), 0, 255, N.FORM_ZEROPOINT)
    emit2 = K.EmitCodes(torch.full((1,), 0.11, device=dev), torch.zeros(1, device=dev), 0, 255, N.FORM_ZEROPOINT)
    return K, a, b, res, emit, emit2


def _reference(K, a, b, res, emit, emit2):
    out, codes = K.conv2d_i8(a["codes"], a["wq"], a["wsum"], a["bias"], a["in_scale"], a["in_zp"], a["w_scale"], residual=res,
                             relu=True, emit=emit, want_out=True)
    _, codes2 = K.conv2d_i8(codes, b["wq"], b["wsum"], b["bias"], emit.scale, emit.zero_point, b["w_scale"], relu=True, emit=emit2,
                            want_out=False)
    return out, codes, codes2


@pytest.mark.parametrize("c,k,k2", [(64, 256, 64), (64, 256, 128), (128, 512, 128), (128, 512, 256), (256, 1024, 256)])
@pytest.mark.parametrize("rows", [0, 64, 49])
def test_chain_matches_two_calls(c, k, k2, rows):
    n, h = 3, 14                      # 588 pixels: partial last tile for every tile height
    K, a, b, res, emit, emit2 = _case(n, h, c, k, k2, seed=c + k2 + rows)
    out_r, codes_r, codes2_r = _reference(K, a, b, res, emit, emit2)
    for want_out, want_codes in ((True, False), (False, True), (True, True)):
        out, codes, codes2 = K.conv2d_i8_chain(a, b, res, relu=True, emit=emit, want_out=want_out, want_codes=want_codes,
                                               relu2=True, emit2=emit2, rows_per_tile=rows)
        torch.cuda.synchronize()
        if want_out:
            assert torch.equal(out.view(torch.int32), out_r.view(torch.int32))
        if want_codes:
            assert torch.equal(codes, codes_r)
        assert torch.equal(codes2, codes2_r)
    # the second layer's weights handed over chunk-major (DLMCQ_W2_CHUNK_MAJOR): same bytes out
    bc = dict(b, wq_chunk=K.chunk_major(b["wq"]))
    out, codes, codes2 = K.conv2d_i8_chain(a, bc, res, relu=True, emit=emit, want_out=True, want_codes=True, relu2=True, emit2=emit2,
                                           rows_per_tile=rows)
    assert torch.equal(out.view(torch.int32), out_r.view(torch.int32)) and torch.equal(codes, codes_r) and torch.equal(codes2, codes2_r)


@pytest.mark.parametrize("c,k,k2", [(64, 256, 64), (64, 256, 128), (128, 512, 128), (128, 512, 256), (256, 1024, 256)])
def test_chain_plain_quantisers_give_the_same_bytes(c, k, k2):
    """Both quantisers unsigned bytes without a zero point (what the frozen plans pass for post-ReLU tensors: a null pointer): the
    kernel's compile-time path (EpiQuant::code4n_plain, the pack's saturation as clamp and ReLU) against the general path (the
    same quantisers with a zero point TENSOR of 0) and against the two separate calls."""
    from dlmc import _native as N
    n, h = 5, 14
    K, a, b, res, emit, emit2 = _case(n, h, c, k, k2, seed=3 * c + k2, zp1=0.0, zp2=0.0)
    plain, plain2 = K.EmitCodes(emit.scale, None, 0, 255, N.FORM_ZEROPOINT), K.EmitCodes(emit2.scale, None, 0, 255, N.FORM_ZEROPOINT)
    out_r, codes_r, codes2_r = _reference(K, a, b, res, emit, emit2)
    for want_out, want_codes in ((True, True), (False, True), (True, False)):
        out, codes, codes2 = K.conv2d_i8_chain(a, b, res, relu=True, emit=plain, want_out=want_out, want_codes=want_codes, relu2=True, emit2=plain2)
        if want_out:
            assert torch.equal(out.view(torch.int32), out_r.view(torch.int32))
        if want_codes:
            assert torch.equal(codes, codes_r)
        assert torch.equal(codes2, codes2_r)


def test_chain_larger_batch_and_nonzero_zero_points():
    K, a, b, res, emit, emit2 = _case(64, 28, 128, 512, 128, seed=7, zp1=0.0, zp2=5.0)
    out_r, codes_r, codes2_r = _reference(K, a, b, res, emit, emit2)
    out, codes, codes2 = K.conv2d_i8_chain(a, b, res, emit=emit, want_out=True, want_codes=True, emit2=emit2)
    assert torch.equal(out.view(torch.int32), out_r.view(torch.int32))
    assert torch.equal(codes, codes_r) and torch.equal(codes2, codes2_r)


@pytest.mark.parametrize("c,c2,k,k3,stride", [(64, 64, 256, 64, 1), (128, 256, 512, 128, 2)])
def test_dual_chain_matches_dual_plus_fused(c, c2, k, k3, stride):
    """First block of a stage: conv3 + downsample convolution + ReLU + quantiser + the next block's conv1 in one kernel, against
    dlmcq_conv2d_i8_nhwc_dual followed by dlmcq_conv2d_i8_nhwc_fused."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(c + k3)
    n, h = 3, 13                                  # 507 pixels: a partial last tile
    h2 = (h - 1) * stride + 1 + (stride - 1)      # 13 (stride 1) / 26 (stride 2: the last row and column are skipped)

    def operand(ch, hh, kk, zp, st):
        x = torch.randint(0, 256, (n, ch, hh, hh), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
        wq = torch.randint(-127, 128, (kk, 1, 1, ch), generator=g, device=dev, dtype=torch.int8)
        return dict(codes=x, wq=wq, wsum=wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(),
                    bias=torch.randn(kk, generator=g, device=dev), in_scale=torch.full((1,), 0.02, device=dev),
                    in_zp=torch.full((1,), zp, device=dev), w_scale=(torch.rand(kk, generator=g, device=dev) * 0.004 + 0.001), stride=st)
    a, b = operand(c, h, k, 2.0, 1), operand(c2, h2, k, 5.0, stride)
    w3 = torch.randint(-127, 128, (k3, 1, 1, k), generator=g, device=dev, dtype=torch.int8)
    c3 = dict(wq=w3, wsum=w3.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=torch.randn(k3, generator=g, device=dev),
              w_scale=(torch.rand(k3, generator=g, device=dev) * 0.002 + 0.0005))
    emit = K.EmitCodes(torch.full((1,), 0.07, device=dev), torch.full((1,), 4.0, device=dev), 0, 255, N.FORM_ZEROPOINT)
    emit3 = K.EmitCodes(torch.full((1,), 0.13, device=dev), torch.zeros(1, device=dev), 0, 255, N.FORM_ZEROPOINT)
    out_r, codes_r = K.conv2d_i8_dual(a, b, relu=True, emit=emit, want_out=True)
    _, codes3_r = K.conv2d_i8(codes_r, c3["wq"], c3["wsum"], c3["bias"], emit.scale, emit.zero_point, c3["w_scale"], relu=True,
                              emit=emit3, want_out=False)
    assert K.dual_chain_supported(c, c2, k, k3, n * h * h)
    for want_out, want_codes in ((True, False), (True, True), (False, True)):
        out, codes, codes3 = K.conv2d_i8_dual_chain(a, b, c3, relu=True, emit=emit, want_out=want_out, want_codes=want_codes,
                                                    relu3=True, emit3=emit3)
        if want_out:
            assert torch.equal(out.view(torch.int32), out_r.view(torch.int32))
        if want_codes:
            assert torch.equal(codes, codes_r)
        assert torch.equal(codes3, codes3_r)
    c3c = dict(c3, wq_chunk=K.chunk_major(c3["wq"]))     # DLMCQ_W2_CHUNK_MAJOR on the third layer
    out, codes, codes3 = K.conv2d_i8_dual_chain(a, b, c3c, relu=True, emit=emit, want_out=True, want_codes=True, relu3=True, emit3=emit3)
    assert torch.equal(out.view(torch.int32), out_r.view(torch.int32)) and torch.equal(codes, codes_r) and torch.equal(codes3, codes3_r)


@pytest.mark.parametrize("c,k,k2", [(64, 256, 64), (64, 256, 128), (128, 512, 128), (128, 512, 256), (256, 1024, 256)])
@pytest.mark.parametrize("rows", [0, 56])
def test_chain_chunk_major_block_tensors_hold_the_same_values(c, k, k2, rows):
    """DLMCQ_FP32_IN / OUT_CHUNK_MAJOR: the fp32 shortcut read and / or the fp32 block output written as [K / 64][M][64] planes
    (kernels.ChunkMajor) - all four combinations (the 128 -> K -> 128 instantiation takes one layout per call: the wrapper converts
    the shortcut for the mixed ones) against the row-major call: same fp32 bits, same codes.  588 pixels: a partial last tile."""
    n, h = 3, 14
    K, a, b, res, emit, emit2 = _case(n, h, c, k, k2, seed=c + 2 * k2 + rows)
    out_r, codes_r, codes2_r = K.conv2d_i8_chain(a, b, res, emit=emit, want_out=True, want_codes=True, emit2=emit2, rows_per_tile=rows)
    res_cm = K.ChunkMajor.from_nhwc(res)
    assert torch.equal(res_cm.to_nhwc(), res) and res_cm.buf.shape == (k // 64, n * h * h, 64)
    assert torch.equal(res_cm.window(2, 3, 9, 5, 14), res[2:3, :, 3:9, 5:14])
    for r_in, out_cm in ((res_cm, True), (res_cm, False), (res, True)):
        out, codes, codes2 = K.conv2d_i8_chain(a, b, r_in, emit=emit, want_out=True, want_codes=True, emit2=emit2, rows_per_tile=rows,
                                               out_chunk_major=out_cm)
        assert isinstance(out, K.ChunkMajor) == out_cm
        got = out.to_nhwc() if out_cm else out
        assert torch.equal(got.view(torch.int32), out_r.view(torch.int32)), (isinstance(r_in, K.ChunkMajor), out_cm)
        assert torch.equal(codes, codes_r) and torch.equal(codes2, codes2_r)
    # a stage end: no fp32 output, the shortcut chunk-major
    _, codes, codes2 = K.conv2d_i8_chain(a, b, res_cm, emit=emit, want_out=False, want_codes=True, emit2=emit2, rows_per_tile=rows)
    assert torch.equal(codes, codes_r) and torch.equal(codes2, codes2_r)


def test_chain_refuses_mixed_layouts_where_it_has_one_offset_set():
    """The C entry point itself: 128 -> K -> 128 with the shortcut chunk-major and the output row-major is DLMCQ_EINVAL (the Python
    wrapper never asks for it: it converts)."""
    from dlmc import _native as N
    K, a, b, res, emit, emit2 = _case(2, 14, 128, 512, 128, seed=5, zp1=0.0)
    assert (128, 128) in K.CHAIN_ONE_LAYOUT
    n, _, h, w = a["codes"].shape
    m, k, k2 = n * h * w, 512, 128
    out = torch.empty((n, k, h, w), device="cuda:0").contiguous(memory_format=torch.channels_last)
    codes2 = torch.empty((n, k2, h, w), dtype=torch.uint8, device="cuda:0").contiguous(memory_format=torch.channels_last)
    ws1, ws2 = a["w_scale"].contiguous(), b["w_scale"].contiguous()

    def call(flags):
        return N.lib.dlmcq_conv2d_i8_nhwc_chain(
            N.ptr(a["codes"]), N.ptr(a["wq"]), N.ptr(out), N.ptr(a["bias"]), N.ptr(a["wsum"]), N.ptr(a["in_scale"]), None, N.ptr(ws1), m, 128, k,
            1, N.ptr(res), 1, None, N.ptr(emit.scale), None, 0, 255, emit.form, 0.0, N.ptr(b["wq"]), N.ptr(b["bias"]), N.ptr(b["wsum"]), N.ptr(ws2),
            k2, 1, N.ptr(codes2), N.ptr(emit2.scale), None, 0, 255, emit2.form | flags, 0.0, 0, N.stream_ptr())
    assert call(N.FP32_IN_CHUNK_MAJOR) == -1 and call(N.FP32_OUT_CHUNK_MAJOR) == -1      # DLMCQ_EINVAL
    assert call(0) == 0 and call(N.FP32_IN_CHUNK_MAJOR | N.FP32_OUT_CHUNK_MAJOR) == 0
    torch.cuda.synchronize()


def test_dual_chain_writes_its_block_tensor_chunk_major():
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(77)
    n, h, c, c2, k, k3 = 3, 13, 128, 256, 512, 128

    def operand(ch, hh, kk, st):
        x = torch.randint(0, 256, (n, ch, hh, hh), generator=g, device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
        wq = torch.randint(-127, 128, (kk, 1, 1, ch), generator=g, device=dev, dtype=torch.int8)
        return dict(codes=x, wq=wq, wsum=wq.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(),
                    bias=torch.randn(kk, generator=g, device=dev), in_scale=torch.full((1,), 0.02, device=dev), in_zp=None,
                    w_scale=(torch.rand(kk, generator=g, device=dev) * 0.004 + 0.001), stride=st)
    a, b = operand(c, h, k, 1), operand(c2, 2 * h, k, 2)
    w3 = torch.randint(-127, 128, (k3, 1, 1, k), generator=g, device=dev, dtype=torch.int8)
    c3 = dict(wq=w3, wsum=w3.to(torch.int32).sum(dim=(1, 2, 3)).to(torch.int32).contiguous(), bias=None,
              w_scale=(torch.rand(k3, generator=g, device=dev) * 0.002 + 0.0005))
    emit = K.EmitCodes(torch.full((1,), 0.07, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
    emit3 = K.EmitCodes(torch.full((1,), 0.13, device=dev), None, 0, 255, N.FORM_ZEROPOINT)
    out_r, _, codes3_r = K.conv2d_i8_dual_chain(a, b, c3, emit=emit, want_out=True, emit3=emit3)
    out, _, codes3 = K.conv2d_i8_dual_chain(a, b, c3, emit=emit, want_out=True, emit3=emit3, out_chunk_major=True)
    assert isinstance(out, K.ChunkMajor) and torch.equal(out.to_nhwc().view(torch.int32), out_r.view(torch.int32)) and torch.equal(codes3, codes3_r)



def test_chain_refuses_unsupported_shapes():
    K, a, b, res, emit, emit2 = _case(1, 7, 64, 256, 64, seed=1)
    assert K.chain_supported(64, 256, 64, 49) and not K.chain_supported(512, 2048, 512, 49)
    bad = dict(b, wq=torch.zeros(96, 1, 1, 256, dtype=torch.int8, device="cuda:0"), wsum=torch.zeros(96, dtype=torch.int32, device="cuda:0"),
               bias=torch.zeros(96, device="cuda:0"), w_scale=torch.ones(96, device="cuda:0"))
    with pytest.raises(RuntimeError):
        K.conv2d_i8_chain(a, bad, res, emit=emit, emit2=emit2)


def test_resnet50_plan_chains_the_block_boundaries_and_stays_bit_identical():
    """The plan pass pairs every non-dual block end with the next block's first 1x1 (stage ends keep their codes for the
    downsample convolution); logits equal the unchained plan's and the wrappers' bit for bit."""
    import workloads as W
    from dlmc.utils.fuse import ChainInt8Layer, fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    cfg = {"weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 8, "signed": True}},
           "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
           "exclude_layers": [], "override_options": []}
    torch.manual_seed(2333)
    net = W.MODELS["resnet50"]().to("cuda:0").eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = merge_bn(net, inplace=True)
    quantize_model(net, cfg, None, "FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(6, 3, 64, 64, device="cuda:0"))
    with torch.no_grad():
        net(x)                                  # calibrate
        want = net(x * 0.8)
        plain = fuse_inference(net, chain_pairs=False)
        chained = fuse_inference(net)
        rowmajor = fuse_inference(net, block_layout=False)
        rep = chained.fusion_report
        # 16 blocks: the very last one has no successor; stage 3's and stage 4's first blocks, the 1024 -> 512 stage end and the
        # rest of stage 4 are wider than the kernel is built for: 11 pairs, 2 of them first blocks of a stage (shortcut =
        # convolution), 2 stage ends that still write their codes for the next stage's downsample convolution
        assert plain.fusion_report.chained == 0 and rep.chained == 11, rep
        pairs = [m for m in chained.modules() if isinstance(m, ChainInt8Layer)]
        assert len(pairs) == 11 and sum(m.want_codes for m in pairs) == 2 and sum(m.short is not None for m in pairs) == 2
        # the fp32 block tensors between kernels that walk them chunk by chunk are chunk-major: the 9 outputs of chain kernels (the last
        # 14^2 chain's is read by the block-end kernel) and stage 3's first block end (the dual block-end kernel, read by the first 14^2
        # chain); stage 4's stay row-major (its first block end runs on the tiled kernel).  The plan that keeps them all row-major gives
        # the same logits
        assert rep.chunk_major == 10 and sum(m.out_cm for m in pairs) == 9 and rowmajor.fusion_report.chunk_major == 0, rep
        y0, y1, y2 = plain(x * 0.8), chained(x * 0.8), rowmajor(x * 0.8)
    assert torch.equal(y0.view(torch.int32), y1.view(torch.int32)) and torch.equal(y2.view(torch.int32), y1.view(torch.int32))
    assert torch.equal(y1.view(torch.int32), want.view(torch.int32))
