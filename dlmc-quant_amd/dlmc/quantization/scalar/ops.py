"""Observer and scale/offset estimators, dispatched by name exactly like the reference
(dlmc/quantization/scalar/ops.py:11-18: `globals()["quantize_" + qtype]`, unknown names raise KeyError).

minmax_* are the hot observers: one read of the tensor through the HIP reduction, the scale/offset
arithmetic on device, no host sync and no transposed copy.  The iterative estimators are
calibration-time only and keep the reference's control flow: l2norm_tensor / l2norm_channel run one fused
HIP launch per iteration (quantize + both reductions, `dlmcq_l2norm_step_f32`); the output-aware and
shrink-search variants run every tensor-sized step on the GPU with the HIP quantize kernel + device reductions.
"""
import torch

from . import kernels as K
from ._wrapper import observe_minmax
from .utils import get_qrange, quantize


def get_qparams_output(input, weight, module, qtype, **kwargs):
    return globals()[f"quantize_{qtype}"](input, weight, module, **kwargs)


def get_qparams_tensor(tensor, qtype, **kwargs):
    return globals()[f"quantize_{qtype}"](tensor, **kwargs)


def l2_loss(t1, t2):
    """trainer/loss/loss.py:22-24 (the reference imports it from the trainer package)."""
    return ((t1 - t2) ** 2).sum(axis=1).mean()


# ------------------------------------------------------------------------------ min / max
def quantize_minmax_tensor(tensor, n_bits, signed, allow_offset=True, sync=False):
    """ops.py:20-34.  Returns 0-dim fp32 device tensors (the reference's signed offset is an int64
    CPU `tensor(0)`; it only ever enters fp32 arithmetic)."""
    s, o = observe_minmax(tensor.detach(), n_bits, signed, None, allow_offset, sync=sync)
    if not signed and not allow_offset:
        _assert_nonneg(tensor)
    return s, o


def quantize_minmax_channel(tensor, n_bits, signed, ch_axis=0, allow_offset=True, sync=False):
    """ops.py:112-140, without the transpose copy; scale/offset shaped [1,..,C,..,1]."""
    s, o = observe_minmax(tensor.detach(), n_bits, signed, ch_axis, allow_offset, sync=sync)
    if not signed and not allow_offset:
        _assert_nonneg(tensor)
    return s, o


def _assert_nonneg(tensor):
    # ops.py:29,132 `assert (min_val >= 0).all()` - calibration-time check, one host sync
    _, mn = K.minmax(tensor.detach())
    assert bool((mn >= 0).item()), "allow_offset=False needs a non-negative tensor"


def _rows(tensor, ch_axis):
    shape = [1] * tensor.dim()
    shape[ch_axis] = -1
    return tensor.transpose(0, ch_axis).reshape(tensor.shape[ch_axis], -1), shape


def quantize_minmax_pixel(tensor, n_bits, signed, allow_offset=True):
    """ops.py:142-167: one scale per kernel position (reduce over out- and in-channels)."""
    new_shape = list(tensor.shape[2:4]) if tensor.dim() == 4 else [tensor.shape[2]]
    t = tensor.detach().reshape(tensor.shape[0], tensor.shape[1], -1)
    pix = t.permute(2, 0, 1).contiguous()  # [P, K, C]: pixel becomes the channel axis of the reduction
    if signed:
        s, o = K.observe_qparams(pix, n_bits, True, ch_axis=0)
    else:
        mx, mn = K.minmax(pix.abs(), ch_axis=0)  # the reference takes abs() here (ops.py:156-158)
        if not allow_offset:
            mn = torch.zeros_like(mn)
        s, o = K.qparams_from_minmax(mx, mn, n_bits, False)
    return s.reshape(new_shape), o.reshape(new_shape)


# ------------------------------------------------------------------- iterative refinement
def quantize_l2norm_tensor(tensor, n_bits, signed):
    """ops.py:71-83."""
    tensor = tensor.detach()
    scale, offset = quantize_minmax_tensor(tensor, n_bits, signed, allow_offset=True)
    lo, hi = get_qrange(signed, n_bits)
    diff = float("inf")
    while diff > 1e-5:
        new_scale = K.l2norm_step(tensor, scale, offset, lo, hi)     # quantize + both reductions, one read
        diff = float((new_scale - scale).abs() / scale)
        scale = new_scale
    return scale, offset


def quantize_l2norm_channel(tensor, n_bits, signed, ch_axis=0):
    """ops.py:198-215."""
    rows, new_shape = _rows(tensor.detach(), ch_axis)
    rows = rows.contiguous()
    scale, offset = quantize_minmax_channel(rows, n_bits, signed, ch_axis=0, allow_offset=True)
    lo, hi = get_qrange(signed, n_bits)
    diff = float("inf")
    while diff > 1e-5:
        new_scale = K.l2norm_step(rows, scale, offset, lo, hi)       # per-row sums, one read
        diff = float(((new_scale - scale) ** 2).sum().sqrt() / (scale ** 2).sum().sqrt())
        scale = new_scale
    return scale.reshape(new_shape), offset.reshape(new_shape)


def quantize_l2norm_output(input, weight, module, n_bits, signed, patience=1000):
    """ops.py:85-109: refine the weight scale against the layer OUTPUT."""
    output = module._forward_func(input, weight)
    scale, offset = quantize_minmax_tensor(weight, n_bits, signed, allow_offset=True)
    lo, hi = get_qrange(signed, n_bits)
    diff, best_mse, best_scale, count = float("inf"), float("inf"), scale, 0
    while diff > 1e-5 and count != patience:
        weight_q = quantize(weight, scale, offset, lo, hi)
        output_q = module._forward_func(input, weight_q)
        mse = l2_loss(output, output_q)
        new_scale = (output_q * output).mean(axis=0).sum() / (output_q * output_q + 1e-7).mean(axis=0).sum()
        diff = float((new_scale - scale).abs() / scale)
        scale = new_scale
        if mse < best_mse:
            best_mse, best_scale = mse, scale
        count += 1
    return best_scale, offset


def quantize_l2norm_output_channel(input, weight, module, n_bits, signed, ch_axis=0, patience=1000):
    """ops.py:252-292."""
    rows, new_shape = _rows(weight.detach(), ch_axis)
    output = module._forward_func(input, weight)
    batch, channel = output.shape[0], output.shape[1]
    output = output.reshape(batch, channel, -1)
    scale, offset = quantize_minmax_channel(rows.contiguous(), n_bits, signed, ch_axis=0, allow_offset=True)
    scale, offset = scale.reshape(new_shape), offset.reshape(new_shape)
    lo, hi = get_qrange(signed, n_bits)
    diff, best_mse, best_scale, count = float("inf"), float("inf"), scale, 0
    while diff > 1e-5 and count != patience:
        weight_q = quantize(weight, scale, offset, lo, hi)
        output_q = module._forward_func(input, weight_q).reshape(batch, channel, -1)
        new_scale = ((output * output_q).sum(axis=(0, 2)) / (output_q * output_q + 1e-7).sum(axis=(0, 2))).reshape(scale.shape)
        mse = l2_loss(output, output_q)
        diff = float(((new_scale - scale) ** 2).sum().sqrt() / (scale ** 2).sum().sqrt())
        if mse < best_mse:
            best_mse, best_scale = mse, scale
        scale = new_scale
        count += 1
    return best_scale.reshape(new_shape), offset


# -------------------------------------------------------------------------- shrink search
def _zp_fakequant(x, scale, zp, qmax):
    """ops.py:58-60: (clamp(round(x/s) + zp, 0, qmax) - zp) * s  ==  the ZEROPOINT form minus the
    round_pass identity; the search only compares losses, so it is evaluated with plain device ops."""
    return ((torch.round(x / scale) + zp).clamp(0, qmax) - zp) * scale


def quantize_l2loss_tensor(tensor, n_bits, signed, allow_offset=True):
    """ops.py:36-68: 80-step shrink search with a rounded integer zero point (unsigned only)."""
    tensor = tensor.detach()
    if signed:
        return quantize_minmax_tensor(tensor, n_bits, True)
    mx, mn = K.minmax(tensor)
    if not allow_offset:
        assert bool((mn >= 0).item())
        mn = torch.zeros_like(mn)
    qmax = 2 ** n_bits - 1
    # all 80 candidates at once on device (the reference loops in Python with one sync per step)
    shrink = 1 - 0.01 * torch.arange(80, device=tensor.device, dtype=torch.float32)
    cand_scale = (shrink * mx - shrink * mn) / qmax
    cand_zp = torch.round(-(shrink * mn) / cand_scale)
    x2 = tensor if tensor.dim() >= 2 else tensor.reshape(1, -1)
    losses = torch.stack([l2_loss(_zp_fakequant(x2, cand_scale[i], cand_zp[i], qmax), x2) for i in range(80)])
    # first strict improvement over 1000 wins, then strictly better ones: = first argmin below 1000
    best = int(torch.argmin(losses).item())
    if not bool(losses[best] < 1000):
        return mx / qmax, torch.zeros_like(mn)
    return cand_scale[best], cand_zp[best]


def quantize_l2loss_channel(tensor, n_bits, signed, ch_axis=0):
    """ops.py:169-196, including its aliasing quirk: `min_val` IS `offset`, so once a step is accepted
    for a channel the following candidates shrink the accepted zero point, not the original minimum.
    The C x 80 Python double loop becomes 80 vectorised steps over all channels."""
    rows, new_shape = _rows(tensor.detach(), ch_axis)
    rows = rows.contiguous()
    scale, offset = quantize_minmax_channel(rows, n_bits, signed, ch_axis=0, allow_offset=True)
    scale, offset = scale.clone(), offset.clone()       # [C,1]
    qmax = 2 ** n_bits - 1
    max_val = offset + scale * qmax
    best = torch.full_like(scale, 1000.0)
    for i in range(80):
        new_min = (1 - 0.01 * i) * offset                 # `offset` doubles as min_val (the alias)
        new_max = (1 - 0.01 * i) * max_val
        new_scale = (new_max - new_min) / qmax
        new_zp = torch.round(-new_min / new_scale)
        tq = _zp_fakequant(rows, new_scale, new_zp, qmax)
        loss = ((rows - tq) ** 2).sum(axis=1, keepdim=True)   # l2_loss of a single row = its squared error
        take = best > loss
        scale = torch.where(take, new_scale, scale)
        offset = torch.where(take, new_zp, offset)
        best = torch.where(take, loss, best)
    return scale.reshape(new_shape), offset.reshape(new_shape)


def quantize_l2norm_pixel(tensor, n_bits, signed, patience=1000):
    """ops.py:217-250 as evidently intended (the reference version cannot run: it calls an
    un-imported `emulate_quantize` and leaves `best_scale` unbound - SURVEY.md defect 7)."""
    from .utils import emulate_quantize
    new_shape = list(tensor.shape[2:4]) if tensor.dim() == 4 else ([tensor.shape[2]] if tensor.dim() == 3 else [1])
    t = tensor.detach().reshape(tensor.shape[0], tensor.shape[1], -1).contiguous()
    scale, offset = quantize_minmax_pixel(t, n_bits, signed)
    scale, offset = scale.reshape(1, 1, -1), offset.reshape(1, 1, -1)
    lo, hi = get_qrange(signed, n_bits)
    diff, best_mse, best_scale, count = float("inf"), float("inf"), scale, 0
    while diff > 1e-5 and count != patience:
        tq = emulate_quantize(t, scale, offset, lo, hi)
        new_scale = ((t * tq).sum(axis=(0, 1)) / (tq * tq + 1e-7).sum(axis=(0, 1))).reshape(scale.shape)
        mse = l2_loss(t, tq)
        diff = float(((new_scale - scale) ** 2).sum().sqrt() / (scale ** 2).sum().sqrt())
        if best_mse > mse:
            best_mse, best_scale = mse, scale
        scale = new_scale
        count += 1
    return best_scale.reshape(new_shape), offset.reshape(new_shape)
