"""float64 reference of ONE node of the frozen plan, computed from the node's own inputs (teacher forcing): the oracle's
quantisers (oracle/fakequant_oracle.py, pinned to the reference by tests/golden/) dequantise the operands, a float64
convolution multiplies them.  Shared by the full-size and the MobileOne plan tests.  Test infrastructure only."""
import math

import torch
import torch.nn.functional as F

from oracle import fakequant_oracle as O


def close(got, ref, mag, what, rtol=2e-6):
    """|got - ref| <= rtol * mag + 2e-5, mag = the magnitude of what the value was summed from (SUM |x'| |w'| + |bias| +
    |shortcut|): sums cancel, and the reference's own fp32 operands carry 6e-8 of each TERM."""
    err = (got.double() - ref).abs()
    tol = rtol * mag + 2e-5
    i = int((err - tol).argmax())
    assert bool((err <= tol).all()), (f"{what}: max excess {float((err - tol).max()):.3g} at {i}: got {float(got.flatten()[i])!r} "
                                      f"ref {float(ref.flatten()[i])!r} mag {float(mag.flatten()[i]):.6g}")


def weight_dequant(node):
    """w' of the node's layer in float64, by the oracle: FSPTQ symmetric (FSPTQuant/base.py:149-152) or QBase (offset =
    channel minimum when asymmetric, modules/base.py:131-133)."""
    from dlmc.quantization.scalar.FSPTQuant.base import FSPTQBase
    lay = node.layer
    w = lay.weight.detach().float().cpu()
    if isinstance(lay, FSPTQBase):
        scale = node.w_scale[:node.k].detach().cpu().reshape([-1] + [1] * (w.dim() - 1))
        wd = O.fq_symmetric(w, scale, node.w_lo, node.w_hi)[1]
    else:
        off = lay.wt_offset.detach().float().cpu() if lay.wt_offset is not None else torch.zeros(())
        g_w = 1 / math.sqrt(w.numel() * lay.wt_max_val)
        wd = O.fq_qbase(w, lay.wt_scale.detach().float().cpu(), off, lay.wt_min_val, lay.wt_max_val, g_w)[1]
    wd = wd.double()
    return wd if wd.dim() == 4 else wd[:, :, None, None]


def plain_codes(codes, spec):
    """Codes as the reference counts them: an unsigned quantiser's codes may travel as int8 `code - 128` inside the plan
    (DLMCQ_EMIT_SHIFT128, dlmc/utils/fuse.py)."""
    if codes.dtype == torch.int8 and spec.lo >= 0:
        return codes.to(torch.int16) + 128
    return codes


def act_dequant(act, codes, numel):
    """x' of activation codes (float64): ZEROPOINT (q - zp) * s, QBASE q * s^ (+ 0)."""
    q = plain_codes(codes, act).double()
    if act.needs_g:
        s_hat = O.ste_scale(act.scale.detach().float().cpu(), act.g(numel))
        return q * float(s_hat.reshape(-1)[0])
    return (q - float(act.zp.reshape(-1)[0])) * float(act.scale.reshape(-1)[0])


def emit_codes(emit, v32, numel):
    """The consumer's codes of an fp32 tensor, by the oracle."""
    s = emit.scale.detach().float().cpu()
    if emit.needs_g:
        return O.fq_qbase(v32, s, torch.zeros(()), emit.lo, emit.hi, emit.g(numel))[0]
    return O.fq_zeropoint(v32, s, emit.zp.detach().float().cpu(), emit.lo, emit.hi)[0]


def conv_window(x_deq, w_deq, bias, stride, pad, groups, win, hw):
    """float64 convolution of dequantised operands on one output window.  x_deq: the input slab already cut to the rows /
    columns the window needs (clipped to the image), `hw` = (h0, w0, h1, w1) of the unclipped receptive field and
    (ch0, cw0, ch1, cw1) of the clipped one.  Returns (value, magnitude)."""
    (h0, w0, h1, w1), (ch0, cw0, ch1, cw1) = hw
    x = F.pad(x_deq, (cw0 - w0, w1 - cw1, ch0 - h0, h1 - ch1))       # padded taps contribute x' = 0
    b = None if bias is None else bias.double()
    ref = F.conv2d(x, w_deq, b, stride=stride, groups=groups)
    mag = F.conv2d(x.abs(), w_deq.abs(), None if b is None else b.abs(), stride=stride, groups=groups)
    return ref, mag


def node_window_ref(codes, act, numel, w_deq, bias, stride, pad, groups, win):
    """Reference value and magnitude of a conv node on output window (n, p0, q0, ph, qw); codes: (N, C, H, W) on any device."""
    n, p0, q0, ph, qw = win
    _, _, H, W = codes.shape
    R, S = w_deq.shape[2], w_deq.shape[3]
    h0, w0 = p0 * stride - pad, q0 * stride - pad
    h1, w1 = (p0 + ph - 1) * stride - pad + R, (q0 + qw - 1) * stride - pad + S
    ch0, cw0, ch1, cw1 = max(h0, 0), max(w0, 0), min(h1, H), min(w1, W)
    x = act_dequant(act, codes[n:n + 1, :, ch0:ch1, cw0:cw1].to("cpu"), numel)
    return conv_window(x, w_deq, bias, stride, pad, groups, win, ((h0, w0, h1, w1), (ch0, cw0, ch1, cw1)))


def check_node(idx, mod, args, out, full, rates=None, max_rate=2e-3, windows=None):
    """One conv node of a frozen plan (Int8Layer / DwInt8Layer / StemLayer without in-kernel pooling) against the float64
    reference computed from ITS OWN input: fp32 to rtol 2e-6 of the summed magnitude, codes exact from the kernel's own fp32
    value and within one code on < max_rate of the elements otherwise.  `rates[idx]` receives the observed off-by-one rate of a
    codes-only node.  `windows`: explicit (n, p0, q0, ph, qw) list (default: whole tensor if `full`, else four corners)."""
    from dlmc.utils.fuse import DwInt8Layer
    fp32, codes = out
    o = fp32 if fp32 is not None else codes
    if o.dim() == 2:
        fp32 = None if fp32 is None else fp32[:, :, None, None]
        codes = None if codes is None else codes[:, :, None, None]
        o = o[:, :, None, None]
    lay = mod.layer
    k = mod.k
    n_img, _, P, Q = o.shape
    w_deq = weight_dequant(mod)
    bias = None if lay.bias is None else lay.bias.detach().float().cpu()
    dw = isinstance(mod, DwInt8Layer)
    stride, pad = (lay.stride[0], lay.padding[0]) if lay.weight.dim() == 4 else (1, 0)
    xin = args[0] if args[0].dim() == 4 else args[0][:, :, None, None]
    xin = xin[:, :mod.c]                                     # drop the padding channels
    numel = xin.numel()
    if xin.dtype == torch.float32:                           # fed by a layer outside the plan: the node quantises its input itself
        xin = emit_codes(mod.act, xin.float().cpu(), numel).to(torch.uint8)
    wins = windows(n_img, P, Q) if windows is not None else [(n, 0, 0, P, Q) for n in range(n_img)] if full else \
        [(n, p0, q0, min(5, P), min(5, Q)) for n in (0, n_img - 1) for p0, q0 in ((0, 0), (P - min(5, P), Q - min(5, Q)))]
    bad = tot = 0
    emit_numel = n_img * k * P * Q
    for win in wins:
        n, p0, q0, ph, qw = win
        ref, mag = node_window_ref(xin, mod.act, numel, w_deq, bias, stride, pad, k if dw else 1, win)
        ref = torch.relu(ref) if mod.relu else ref
        what = f"node {idx} {type(mod).__name__} {tuple(lay.weight.shape)} window {win}"
        got32 = None
        if fp32 is not None:
            got32 = fp32[n:n + 1, :k, p0:p0 + ph, q0:q0 + qw].cpu()
            close(got32, ref, mag, what)
        if codes is not None:
            got = plain_codes(codes[n:n + 1, :, p0:p0 + ph, q0:q0 + qw].cpu(), mod.emit)
            if got.shape[1] > k:      # padding channels: the consumer's code of 0
                pad_code = emit_codes(mod.emit, torch.zeros(1), emit_numel)
                assert bool((got[:, k:].float() == float(pad_code)).all()), f"{what}: padding channels"
            if got32 is not None:
                assert torch.equal(got[:, :k].float(), emit_codes(mod.emit, got32, emit_numel)), f"{what}: codes of the kernel's own fp32 value"
            else:
                off = (got[:, :k].float() - emit_codes(mod.emit, ref.float(), emit_numel)).abs()
                assert float(off.max()) <= 1, f"{what}: codes off by {float(off.max())}"
                bad, tot = bad + int((off > 0).sum()), tot + off.numel()
    if tot:
        assert bad / tot < max_rate, f"node {idx} {type(mod).__name__} {tuple(lay.weight.shape)}: {bad / tot:.2e} of the codes off by one"
        if rates is not None:
            rates[idx] = bad / tot


