"""Parity of the fused plan at the BENCHMARK's own size (BASELINE configs[2]: ResNet-50 W8A8 per-channel, batch 512 at
224 x 224) and a teacher-forced end-to-end check.

Every plan node (all kernel instantiations the bench launches: 128x64 / 128x128 tiles, activations direct or through the
LDS ring, the dual kernels, stem + pool, every residual / fp32-out / codes epilogue combination) is given its REAL inputs
at full size - layer1's fp32 tensors are 1.64 GB, M = N*P*Q reaches 1.6 M rows - and its output is
compared on sampled windows (first image, a middle one, and the LAST rows of the last image, both corners) with a float64
convolution of the dequantised operands plus the oracle's quantisers (oracle/fakequant_oracle.py, pinned to the reference
by tests/golden/): fp32 values to rtol 2e-6, codes bit-exact where the kernel also wrote the fp32 value they come from,
within one code on < 1e-3 of the elements otherwise (fp32 accumulation-order ties).  "rtol 2e-6" is taken relative to
the magnitude the value was summed from (SUM |x'| |w'| + |bias| + |shortcut|): at this depth sums cancel to a thousandth of
their terms, and the reference's own operands (w' = q * s_w, an fp32 product) carry 6e-8 of each TERM.

Because each node is checked against a reference computed from ITS OWN input, no drift accumulates: this is per-node
parity at scale, not a bound on end-to-end drift.
"""
import json

import pytest
import torch
import torch.nn.functional as F

from oracle import fakequant_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FSPTQ = {  # example/quantization/FSPTQ_config.yaml:40-53
    "weight": {"enable": True, "type": "minmax_channel", "recon_type": "None", "args": {"n_bits": 8, "signed": True}},
    "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
    "exclude_layers": [], "override_options": [],
}


def _windows(n_img, P, Q, size=6):
    """(image, p0, q0, ph, qw): both corners of the first / a middle / the last image (the end of the tensor), and - round 5 - windows
    that STRADDLE TILE SEAMS: the kernels cut the pixel index m = (n P + p) Q + q into blocks of 32 (pointwise), 64 (chain), 128 (tiled)
    and 256 rows (halo, in frame positions), so in two interior images a window is laid around the first pixel whose index is a multiple
    of 256 (a seam of every one of those block sizes at once: pixels m - 1 and m sit in different tiles), and the first corner of image 1
    faces the last corner of image 0 (an image boundary inside a tile)."""
    ph, qw = min(size, P), min(size, Q)
    out = []
    for n in sorted({0, n_img // 2, n_img - 1}):
        out += [(n, 0, 0, ph, qw), (n, P - ph, Q - qw, ph, qw)]
    if n_img > 2:
        out.append((1, 0, 0, ph, qw))
    for n in sorted({n_img // 3, (2 * n_img) // 3}):
        m0 = n * P * Q
        seam = -(-m0 // 256) * 256 - m0            # first pixel of this image whose index is a multiple of 256
        if 0 < seam < P * Q:
            p, q = seam // Q, seam % Q
            out.append((n, max(0, min(P - ph, p - ph // 2)), max(0, min(Q - qw, q - qw // 2)), ph, qw))
    return out


def _conv_window_ref(codes, s_in, zp, weight_deq, bias, stride, pad, win):
    """float64 convolution of the dequantised operands on one output window.  codes: (N, C, H, W) uint8 on the GPU."""
    n, p0, q0, ph, qw = win
    _, _, H, W = codes.shape
    R, S = weight_deq.shape[2], weight_deq.shape[3]
    h0, w0 = p0 * stride - pad, q0 * stride - pad
    h1, w1 = (p0 + ph - 1) * stride - pad + R, (q0 + qw - 1) * stride - pad + S
    ch0, cw0, ch1, cw1 = max(h0, 0), max(w0, 0), min(h1, H), min(w1, W)
    x = codes[n:n + 1, :, ch0:ch1, cw0:cw1].to("cpu")
    if x.dtype == torch.int8:          # an unsigned quantiser's codes handed over as int8 `code - 128` (DLMCQ_EMIT_SHIFT128)
        x = x.to(torch.int16) + 128
    x = x.double()
    x = (x - float(zp)) * float(s_in)                      # x' = (q - zp) * s   (FSPTQuant/base.py:108-109)
    x = F.pad(x, (cw0 - w0, w1 - cw1, ch0 - h0, h1 - ch1))   # padded taps contribute x' = 0
    ref = F.conv2d(x, weight_deq, None if bias is None else bias.double(), stride=stride)
    # what the result was summed from: every fp32 rounding of the reference's operands (w' = q * s_w is an fp32 product) and of
    # the kernel's chain is relative to these magnitudes, not to a (possibly cancelling) sum
    mag = F.conv2d(x.abs(), weight_deq.abs(), None if bias is None else bias.double().abs(), stride=stride)
    return ref, mag


def _layer_params(node):
    lay = node.layer
    w = lay.weight.detach().float().cpu()
    if w.dim() == 2:
        w = w[:, :, None, None]
    scale = node.w_scale.detach().cpu().reshape(-1, 1, 1, 1)
    w_deq = O.fq_symmetric(w, scale, node.w_lo, node.w_hi)[1].double()   # FSPTQuant/base.py:149-152
    bias = None if lay.bias is None else lay.bias.detach().float().cpu()
    if lay.weight.dim() == 2:
        return w_deq, bias, 1, 0
    return w_deq, bias, lay.stride[0], lay.padding[0]


def _close(got, ref, mag, what):
    """|got - ref| <= 2e-6 * mag + 2e-5, mag = the magnitude of the addends the fp32 result was rounded from (a shortcut add
    of two large values of opposite sign leaves their rounding errors in a small sum)."""
    err = (got.double() - ref).abs()
    tol = 2e-6 * mag + 2e-5
    i = int((err - tol).argmax())
    assert bool((err <= tol).all()), (f"{what}: max excess {float((err - tol).max()):.3g} at {i}: got {float(got.flatten()[i])!r} "
                                      f"ref {float(ref.flatten()[i])!r} mag {float(mag.flatten()[i]):.6g}")


def _check_codes(got_codes, fp32_out, ref32, emit, what):
    """got_codes against the oracle's quantiser: exact from the kernel's own fp32 value, +-1 rarely from the reference's."""
    s, z = emit.scale.detach().cpu(), emit.zp.detach().cpu()
    if got_codes.dtype == torch.int8 and emit.lo >= 0:      # stored as `code - 128` (DLMCQ_EMIT_SHIFT128): compare the codes themselves
        got_codes = got_codes.to(torch.int16) + 128
    if fp32_out is not None:
        want = O.fq_zeropoint(fp32_out, s, z, emit.lo, emit.hi)[0]
        assert torch.equal(got_codes.float(), want), f"{what}: codes differ from the oracle's codes of the kernel's own fp32 output"
        return None
    want = O.fq_zeropoint(ref32, s, z, emit.lo, emit.hi)[0]
    off = (got_codes.float() - want).abs()
    assert float(off.max()) <= 1 and float((off > 0).float().mean()) < 1e-3, \
        f"{what}: codes off by {float(off.max())} / {float((off > 0).float().mean()):.2e} of elements"
    return int((off > 0).sum()), off.numel()


# Observed share of codes-only elements one code away from the float64 reference's code (int32-exact accumulation against
# float64 at rounding ties), summed over the sampled windows of every codes-only node.  Round 3 measured
# OFF_BY_ONE_SEEN; the tests fail at twice that, so that a drift from 1e-5 to 9e-4 cannot pass under the per-window bound of 1e-3.
OFF_BY_ONE_SEEN = {"resnet50": 5.0e-6, "repvgg_a1": 5.0e-5}    # (measured: 1.5e-6 = 3 of 2.0 M; 1.8e-5 = 1 of 55 k on the worst node)


def _rate_check(name, counts):
    bad, tot = sum(b for b, _ in counts), sum(t for _, t in counts)
    rate = bad / max(tot, 1)
    print(f"{name}: {bad} of {tot} sampled codes-only elements are one code off the float64 reference ({rate:.2e})")
    assert rate <= 2 * OFF_BY_ONE_SEEN[name], f"{name}: off-by-one rate {rate:.2e} above twice the recorded {OFF_BY_ONE_SEEN[name]:.1e}"


@pytest.mark.timeout(900)
def test_resnet50_batch512_every_plan_node_against_the_oracle():
    import workloads as W
    from dlmc import _native as N
    from dlmc.utils.fuse import ChainInt8Layer, DualInt8Layer, Int8Layer, StemLayer, fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    batch = 512
    torch.manual_seed(2333)
    model = merge_bn(W.resnet50().to(DEV).eval(), inplace=True, allow_missing=True)
    quantize_model(model, json.loads(json.dumps(FSPTQ)), None, quantization_type="FSPTQ", int8_gemm=True)
    # the image is NCHW, as bench.py hands it over (the reference's loaders do): the first node's quantiser then runs its
    # four-pixels-per-thread plane reader at full size; the channels_last form is checked against it below
    x = torch.relu(torch.randn(batch, 3, 224, 224, device=DEV))
    assert x.is_contiguous()
    recs = []
    with torch.no_grad():
        model(x)
        plan = fuse_inference(model)
        duals = [m for m in plan.modules() if isinstance(m, DualInt8Layer)]
        chains = [m for m in plan.modules() if isinstance(m, ChainInt8Layer)]
        inner = {id(d.a) for d in duals} | {id(d.b) for d in duals}
        for c in chains:
            inner |= {id(c.a), id(c.b), id(c.main), id(c.short)}
        for m in plan.modules():
            if isinstance(m, (ChainInt8Layer, DualInt8Layer, StemLayer)) or (isinstance(m, Int8Layer) and id(m) not in inner):
                m.register_forward_hook(lambda mod, args, out: recs.append((mod, args, out)))
        plan(x)
        torch.cuda.synchronize()
    # 54 layers: 11 chain kernels (2 of them with a convolution shortcut: 24 layers), 2 dual kernels (4), 1 stem, 25 single layers
    assert len(recs) == 39 and len(chains) == 11, (len(recs), len(chains))
    kinds = set()
    beyond_2g = 0
    counts = []
    for idx, (mod, args, out) in enumerate(recs):
        if isinstance(mod, ChainInt8Layer):
            kinds.add(_check_chain_node(mod, args, out, idx, counts))
            continue
        fp32, codes = out
        emit = mod.a.emit if isinstance(mod, DualInt8Layer) else mod.emit
        o = fp32 if fp32 is not None else codes
        if o.dim() == 2:
            o4 = (lambda t: None if t is None else t[:, :, None, None])
            fp32, codes, o = o4(fp32), o4(codes), o[:, :, None, None]
        n_img, k, P, Q = o.shape
        if fp32 is not None and fp32.numel() * 4 > 2 ** 31:
            beyond_2g += 1
        for win in _windows(n_img, P, Q):
            n, p0, q0, ph, qw = win
            what = f"node {idx} {type(mod).__name__} out {tuple(o.shape)} window {win}"
            if isinstance(mod, StemLayer):
                lay, act = mod.layer, mod.act
                w = lay.weight.detach().float().cpu()
                w_deq = O.fq_symmetric(w, mod.w_scale.detach().cpu().reshape(-1, 1, 1, 1), mod.w_lo, mod.w_hi)[1].double()
                bias = None if lay.bias is None else lay.bias.detach().float().cpu()
                # pooled window <- conv window (3x3 / 2 / 1) <- image window; quantise the image with the oracle
                c0p, c0q = 2 * p0 - 1, 2 * q0 - 1
                cp0, cq0 = max(c0p, 0), max(c0q, 0)
                cp1, cq1 = min(2 * (p0 + ph - 1) + 2, 112), min(2 * (q0 + qw - 1) + 2, 112)
                img_codes = O.fq_zeropoint(args[0][n:n + 1].float().cpu(), act.scale.cpu(), act.zp.cpu(), act.lo, act.hi)[0]
                conv, cmag = _conv_window_ref(img_codes.to(torch.uint8), act.scale.cpu(), act.zp.cpu(), w_deq, bias, 2, 3,
                                              (0, cp0, cq0, cp1 - cp0, cq1 - cq0))
                conv = torch.relu(conv) if mod.relu else conv
                conv = F.pad(conv, (cq0 - c0q, 0, cp0 - c0p, 0), value=-float("inf"))
                ref = F.max_pool2d(conv, 3, 2, 0, ceil_mode=False)[:, :, :ph, :qw]
                mag = F.max_pool2d(F.pad(cmag, (cq0 - c0q, 0, cp0 - c0p, 0)), 3, 2, 0)[:, :, :ph, :qw]
                kinds.add("stem+pool")
            elif isinstance(mod, DualInt8Layer):
                ref = mag = 0
                for part, xin in ((mod.a, args[0]), (mod.b, args[1])):
                    w_deq, bias, stride, pad = _layer_params(part)
                    one, omag = _conv_window_ref(xin, part.act.scale.cpu(), part.act.zp.cpu(), w_deq, bias, stride, pad, win)
                    ref, mag = ref + one.float().double(), mag + omag
                ref = torch.relu(ref) if mod.a.relu else ref
                kinds.add(f"dual bn{'64' if k <= 256 else '128'}")
            else:
                w_deq, bias, stride, pad = _layer_params(mod)
                xin = args[0] if args[0].dim() == 4 else args[0][:, :, None, None]
                if xin.dtype == torch.float32:      # the classifier head: fp32 features quantised by the node itself
                    xin = O.fq_zeropoint(xin.cpu(), mod.act.scale.cpu(), mod.act.zp.cpu(), mod.act.lo, mod.act.hi)[0].to(torch.uint8)
                ref, mag = _conv_window_ref(xin, mod.act.scale.cpu(), mod.act.zp.cpu(), w_deq, bias, stride, pad, win)
                if len(args) > 1:
                    res = _win4(args[1], n, p0, ph, q0, qw).cpu().double()
                    mag = mag + res.abs()
                    ref = ref.float().double() + res
                ref = torch.relu(ref) if mod.relu else ref
                kinds.add(f"{'1x1' if w_deq.shape[2] == 1 else '3x3'} K{k} {'res ' if len(args) > 1 else ''}"
                          f"{'out ' if fp32 is not None else ''}{'codes' if codes is not None else ''}")
            ref32 = ref.float()
            got32 = None
            if fp32 is not None:
                got32 = _win4(fp32, n, p0, ph, q0, qw).cpu()
                _close(got32, ref, mag, what)
            if codes is not None:
                c = _check_codes(codes[n:n + 1, :, p0:p0 + ph, q0:q0 + qw].cpu(), got32, ref32, emit, what)
                if c is not None:
                    counts.append(c)
    _rate_check("resnet50", counts)
    # the first node once more with the same image in channels_last memory (the other reader of the image quantiser): same bytes
    stem_mod, stem_args, stem_out = next(r for r in recs if isinstance(r[0], StemLayer))
    with torch.no_grad():
        again = stem_mod(stem_args[0].contiguous(memory_format=torch.channels_last))
    for a, b in zip(again, stem_out):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), "first node: NCHW and channels_last images disagree"
    assert beyond_2g == 0      # (layer1's fp32 tensors are 1.64 GB: byte offsets pass 2^30, not 2^31 - the next test does)
    print(f"checked {len(recs)} plan nodes x 6 windows; epilogue / kernel kinds seen: {sorted(kinds)}")


def _win4(t, n, p0, ph, q0, qw):
    """Window [1, K, ph, qw] of image n of an fp32 block tensor, channels_last or chunk-major (dlmc ... kernels.ChunkMajor)."""
    from dlmc.quantization.scalar import kernels as K
    if isinstance(t, K.ChunkMajor):
        return t.window(n, p0, p0 + ph, q0, q0 + qw)
    return t[n:n + 1, :, p0:p0 + ph, q0:q0 + qw]


def _check_chain_node(mod, args, out, idx, counts):
    """One chain kernel (block end + next block's first 1x1): its fp32 output / codes against the oracle's convolution(s) +
    shortcut + ReLU, and the second convolution's codes against the oracle's convolution of the FIRST layer's codes (the
    kernel's own, or the oracle's codes of the kernel's own fp32 output: they never leave the chip otherwise)."""
    fp32, codes, codes2 = out
    a, b = mod.a, mod.b
    n_img, k, P, Q = codes2.shape[0], a.k, codes2.shape[2], codes2.shape[3]
    for win in _windows(n_img, P, Q):
        n, p0, q0, ph, qw = win
        what = f"node {idx} chain {tuple(codes2.shape)} window {win}"
        if mod.short is None:
            w_deq, bias, stride, pad = _layer_params(a)
            ref, mag = _conv_window_ref(args[0], a.act.scale.cpu(), a.act.zp.cpu(), w_deq, bias, stride, pad, win)
            res = _win4(args[1], n, p0, ph, q0, qw).cpu().double()      # (between two chain kernels the block tensor is a K.ChunkMajor)
            ref, mag = ref.float().double() + res, mag + res.abs()
        else:
            xm, xs = (args[1], args[0]) if mod.swapped else (args[0], args[1])
            ref = mag = 0
            for part, xin in ((mod.main, xm), (mod.short, xs)):
                w_deq, bias, stride, pad = _layer_params(part)
                one, omag = _conv_window_ref(xin, part.act.scale.cpu(), part.act.zp.cpu(), w_deq, bias, stride, pad, win)
                ref, mag = ref + one.float().double(), mag + omag
        ref = torch.relu(ref) if a.relu else ref
        got32 = None
        if fp32 is not None:
            got32 = _win4(fp32, n, p0, ph, q0, qw).cpu()
            _close(got32, ref, mag, what)
        if codes is not None:
            c = _check_codes(codes[n:n + 1, :, p0:p0 + ph, q0:q0 + qw].cpu(), got32, ref.float(), a.emit, what)
            if c is not None:
                counts.append(c)
        e = a.emit
        mid = (codes[n:n + 1, :, p0:p0 + ph, q0:q0 + qw].cpu() if codes is not None else
               O.fq_zeropoint(got32, e.scale.detach().cpu(), e.zp.detach().cpu(), e.lo, e.hi)[0].to(torch.uint8))
        w_deq, bias, stride, pad = _layer_params(b)
        ref2, _ = _conv_window_ref(mid, b.act.scale.cpu(), b.act.zp.cpu(), w_deq, bias, stride, pad, (0, 0, 0, ph, qw))
        ref2 = torch.relu(ref2) if b.relu else ref2
        counts.append(_check_codes(codes2[n:n + 1, :, p0:p0 + ph, q0:q0 + qw].cpu(), None, ref2.float(), b.emit, what + " second convolution"))
    return f"chain K{k} {'conv shortcut ' if mod.short is not None else ''}{'out ' if fp32 is not None else ''}{'codes' if codes is not None else ''}"


@pytest.mark.timeout(900)
def test_repvgg_a1_batch512_every_plan_node_against_the_oracle():
    """BASELINE configs[3]'s one-GPU shard at its stated size: RepVGG-A1 in deploy form (model/classification/repvgg.py:132-147,
    205-207), the reference's few-shot-PTQ flow (example/quantization/FSPTQuant.py:65-67,80; FSPTQuant/base.py:95-159: W
    `minmax_channel` s8, A `minmax_tensor` u8), 512 images at 224 x 224.  Every plan node - the swapped 3x3 first-layer kernel,
    the stride-2 3x3 layers on the generic kernel, the stride-1 layers on the halo-tile kernel (64-wide and 128-wide tiles), the
    1280-wide last layer with its fp32 output - against the float64 convolution of the oracle's dequantised operands on windows of
    the first, a middle and the LAST image."""
    import workloads as W
    from dlmc.utils.fuse import Int8Layer, StemLayer, fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    from plan_reference import check_node
    torch.manual_seed(2333)
    model = merge_bn(W.MODELS["repvgg_a1"]().to(DEV).eval(), inplace=True, allow_missing=True)
    quantize_model(model, json.loads(json.dumps(FSPTQ)), None, quantization_type="FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(512, 3, 224, 224, device=DEV)).contiguous(memory_format=torch.channels_last)
    recs = []
    with torch.no_grad():
        model(x)
        plan = fuse_inference(model)
        for m in plan.modules():
            if isinstance(m, (Int8Layer, StemLayer)):
                m.register_forward_hook(lambda mod, args, out: recs.append((mod, args, out)))
        plan(x)
        torch.cuda.synchronize()
    assert plan.fusion_report.skipped == [] and isinstance(recs[0][0], StemLayer)
    convs = [r for r in recs if r[0].layer.weight.dim() == 4]
    assert len(convs) == 22, len(convs)          # 1 + 2 + 4 + 14 + 1 (the classifier reads fp32 features)
    rates = {}
    for idx, (mod, args, out) in enumerate(convs):
        check_node(idx, mod, args, out, False, rates=rates, max_rate=1e-3, windows=_windows)
    bad = sum(rates.values())
    print(f"repvgg_a1: per-node off-by-one rates of the codes-only nodes: max {max(rates.values()):.2e}, mean {bad / len(rates):.2e}")
    assert max(rates.values()) <= 2 * OFF_BY_ONE_SEEN["repvgg_a1"], rates      # (per node: 216 x K sampled elements each)


def test_streamed_plan_is_the_single_stream_plan_at_batch_512():
    """What bench.py times - StreamedPlan(plan, 2): the batch split over two HIP streams, 256 images each, hence other M, other tail
    tiles and other tile counts than the single-stream plan the test above checks - is bit-identical to that plan at full size."""
    import workloads as W
    from dlmc.utils.fuse import StreamedPlan, fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    model = merge_bn(W.resnet50().to(DEV).eval(), inplace=True, allow_missing=True)
    quantize_model(model, json.loads(json.dumps(FSPTQ)), None, quantization_type="FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(512, 3, 224, 224, device=DEV)).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        model(x)
        plan = fuse_inference(model)
        one = plan(x)
        two = StreamedPlan(plan, 2)(x)
        three = StreamedPlan(plan, 3)(x)          # 171 + 171 + 170 images: uneven shares
        torch.cuda.synchronize()
    assert torch.equal(one, two) and torch.equal(one, three)



def test_byte_offsets_beyond_2_to_the_31():
    """A block-end layer of layer1 (1x1, 64 -> 256 at 56 x 56, shortcut + ReLU + fp32 out + codes) at batch 768: the fp32
    shortcut and output are 2.47 GB each, so the last images lie beyond byte offset 2^31 (and element index 2^29)."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    n, c, k, h = 768, 64, 256, 56
    g = torch.Generator(device=DEV).manual_seed(77)
    codes = torch.randint(0, 256, (n, c, h, h), generator=g, device=DEV, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(k, c, 1, 1, generator=g, device=DEV) * 0.05
    s_w = wt.abs().amax(dim=(1, 2, 3)) / 127 + 1e-6
    bias = torch.randn(k, generator=g, device=DEV)
    wq, wsum = K.quantize_weight_krsc(wt, s_w, -127, 127)
    s_in, zp = torch.tensor([0.02], device=DEV), torch.tensor([3.0], device=DEV)
    res = torch.randn(n, k, h, h, generator=g, device=DEV).contiguous(memory_format=torch.channels_last)
    assert res.numel() * 4 > 2 ** 31
    emit = K.EmitCodes(torch.tensor([0.05], device=DEV), torch.tensor([9.0], device=DEV), 0, 255, N.FORM_ZEROPOINT)
    out, oc = K.conv2d_i8(codes, wq, wsum, bias, s_in, zp, s_w, residual=res, relu=True, emit=emit)
    w_deq = O.fq_symmetric(wt.cpu(), s_w.cpu().reshape(-1, 1, 1, 1), -127, 127)[1].double()
    for win in _windows(n, h, h):
        i, p0, q0, ph, qw = win
        ref, mag = _conv_window_ref(codes, s_in.cpu(), zp.cpu(), w_deq, bias.cpu(), 1, 0, win)
        r = res[i:i + 1, :, p0:p0 + ph, q0:q0 + qw].cpu().double()
        ref = torch.relu(ref.float().double() + r)
        got = out[i:i + 1, :, p0:p0 + ph, q0:q0 + qw].cpu()
        _close(got, ref, mag + r.abs(), f"window {win}")
        want = O.fq_zeropoint(got, emit.scale.cpu(), emit.zero_point.cpu(), 0, 255)[0]
        assert torch.equal(oc[i:i + 1, :, p0:p0 + ph, q0:q0 + qw].cpu().float(), want), f"window {win}: codes"


def test_fused_plan_teacher_forced_against_the_cpu_port():
    """End to end WITHOUT drift: every plan node of a small ResNet-50 (batch 4, 64 x 64) is fed the input the fused plan
    itself produced and compared, whole tensor, with the float64 reference of that node (as above).  Together with the
    bit-exact quantisers this is parity of the entire forward, node by node."""
    import copy
    import workloads as W
    from dlmc.utils.fuse import DualInt8Layer, Int8Layer, StemLayer, fuse_inference
    from dlmc.utils.merge_bn import merge_bn
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    base = W.resnet50().eval()
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = merge_bn(copy.deepcopy(base).to(DEV), inplace=True)
    quantize_model(net, json.loads(json.dumps(FSPTQ)), None, "FSPTQ", int8_gemm=True)
    x = torch.relu(torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(2833))).to(DEV)
    recs = []
    with torch.no_grad():
        net(x)
        # one node per launch-of-one-layer: the chained plan (block end + next 1x1 in one kernel) is bit-identical to this one
        # (tests/test_gpu_chain.py) and its chain kernels face the oracle at full size in the test above
        plan = fuse_inference(net, chain_pairs=False)
        duals = [m for m in plan.modules() if isinstance(m, DualInt8Layer)]
        inner = {id(d.a) for d in duals} | {id(d.b) for d in duals}
        for m in plan.modules():
            if isinstance(m, DualInt8Layer) or (isinstance(m, Int8Layer) and id(m) not in inner):
                m.register_forward_hook(lambda mod, args, out: recs.append((mod, args, out)))
        plan(x)
    assert len(recs) == 49
    for idx, (mod, args, out) in enumerate(recs):
        fp32, codes = out
        o = fp32 if fp32 is not None else codes
        if o.dim() == 2:
            fp32 = None if fp32 is None else fp32[:, :, None, None]
            codes = None if codes is None else codes[:, :, None, None]
            o = o[:, :, None, None]
        n_img, k, P, Q = o.shape
        refs, mags = [], []
        for n in range(n_img):
            win = (n, 0, 0, P, Q)
            if isinstance(mod, DualInt8Layer):
                ref = mag = 0
                for part, xin in ((mod.a, args[0]), (mod.b, args[1])):
                    w_deq, bias, stride, pad = _layer_params(part)
                    one, omag = _conv_window_ref(xin, part.act.scale.cpu(), part.act.zp.cpu(), w_deq, bias, stride, pad, win)
                    ref, mag = ref + one.float().double(), mag + omag
                relu, emit = mod.a.relu, mod.a.emit
            else:
                w_deq, bias, stride, pad = _layer_params(mod)
                xin = args[0] if args[0].dim() == 4 else args[0][:, :, None, None]
                if xin.dtype == torch.float32:
                    xin = O.fq_zeropoint(xin.cpu(), mod.act.scale.cpu(), mod.act.zp.cpu(), mod.act.lo, mod.act.hi)[0].to(torch.uint8)
                ref, mag = _conv_window_ref(xin, mod.act.scale.cpu(), mod.act.zp.cpu(), w_deq, bias, stride, pad, win)
                if len(args) > 1:
                    mag = mag + args[1][n:n + 1].cpu().double().abs()
                    ref = ref.float().double() + args[1][n:n + 1].cpu().double()
                relu, emit = mod.relu, mod.emit
            refs.append(torch.relu(ref) if relu else ref)
            mags.append(mag)
        ref, mag = torch.cat(refs), torch.cat(mags)
        what = f"node {idx} {type(mod).__name__} out {tuple(o.shape)}"
        got32 = None
        if fp32 is not None:
            got32 = fp32.cpu()
            _close(got32, ref, mag, what)
        if codes is not None:
            _check_codes(codes.cpu(), got32, ref.float(), emit, what)
