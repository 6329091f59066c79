"""CPU oracle for the fake-quantize hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU and op for op, the arithmetic of the reference's fake-quantize
path so that the HIP kernels can be checked bit for bit.  It is the checker, never the product:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The
shipped package (`dlmc-quant_amd/`) never imports anything under `oracle/` and has no CPU
fallback.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here against
`tests/golden/golden_v1.npz`, vectors produced by the reference's own code
(`tests/golden/make_golden.py`, run in the build container against /root/reference).

Why torch-on-CPU and not numpy: the reference *is* a sequence of ATen CPU ops; restating it with
the same ATen ops makes the rounding of every intermediate identical by construction (IEEE fp32
sub/div/mul/add, `round` = half-to-even, NaN-propagating `clamp`/`max`/`min`), and gives the
"repo's own CPU path" its real multi-threaded speed for `cpu_baseline`.  An independent scalar C
restatement lives in `oracle/fq_oracle.c` and is checked against the same vectors.

All citations are relative to /root/reference.
"""
import math

import torch


# ------------------------------------------------------------------------------- ranges
def qrange(signed, n_bits):
    """dlmc/quantization/scalar/utils.py:14-22.  Signed range is symmetric: -127..127 for 8 bit."""
    if signed:
        hi = 2 ** (n_bits - 1) - 1
        return -hi, hi
    return 0, 2 ** n_bits - 1


# ----------------------------------------------------------------------- fake-quant forms
def ste_scale(scale, g):
    """utils.py:24-27 `grad_scale`: forward value is (s - s*g) + s*g, which is NOT always s
    (1 element in 1e5 differs by an ulp), so the HIP kernel reproduces it."""
    sg = scale * g
    return (scale - sg) + sg


def ste_round(v):
    """utils.py:29-32 `round_pass`: forward value is (round(v) - v) + v.  Equals round(v) except
    that -0.0 becomes +0.0 and +-inf becomes NaN."""
    return (v.round() - v) + v


def ste_floor(v):
    """utils.py:34-37 `floor_pass`."""
    return (v.floor() - v) + v


def fq_emulate(x, scale, offset, lo, hi):
    """Form A - utils.py:1-11 `emulate_quantize`.
    q = clamp(round((x - o) / (s + 1e-7)), lo, hi);  y = q * s + o.  Returns (q, y)."""
    q = ((x - offset) / (scale + 1e-7)).round().clamp(lo, hi)
    return q, q * scale + offset


def fq_qbase(x, scale, offset, lo, hi, g=0.0):
    """Form B - modules/base.py:96-102 (input) and :131-133 (weight).
    s^ = ste_scale(s, g);  q = ste_round(clamp((x - o) / s^, lo, hi));  y = q * s^ + o."""
    s_hat = ste_scale(scale, g)
    q = ste_round(((x - offset) / s_hat).clamp(lo, hi))
    return q, q * s_hat + offset


def fq_zeropoint(x, scale, zp, lo, hi):
    """Form C - FSPTQuant/base.py:108-109 (activations).
    q = clamp(ste_round(x / s) + zp, lo, hi);  y = (q - zp) * s."""
    q = (ste_round(x / scale) + zp).clamp(lo, hi)
    return q, (q - zp) * scale


def fq_symmetric(w, scale, lo, hi):
    """Form D - FSPTQuant/base.py:149-152 (weights, recon_type not adaround).
    q = clamp(ste_round(w / s), lo, hi);  y = q * s."""
    q = ste_round(w / scale).clamp(lo, hi)
    return q, q * scale


def adaround_soft_targets(alpha, gamma=-0.1, zeta=1.1):
    """FSPTQuant/base.py:78-79."""
    return torch.clamp(torch.sigmoid(alpha) * (zeta - gamma) + gamma, 0, 1)


def adaround_init_alpha(w, scale, gamma=-0.1, zeta=1.1):
    """FSPTQuant/base.py:69-76."""
    fl = torch.floor(w / scale)
    rest = w / scale - fl
    return -torch.log((zeta - gamma) / (rest - gamma) - 1)


def fq_adaround(w, scale, alpha, lo, hi, training):
    """FSPTQuant/base.py:136-141,151-152: q = floor(w/s) + h(alpha) (train) or + [alpha >= 0] (eval)."""
    q = torch.floor(w / scale)
    q = q + (adaround_soft_targets(alpha) if training else (alpha >= 0).float())
    q = q.clamp(lo, hi)
    return q, q * scale


def rootq_clip(x, upper, lower):
    """RootQ/function.py:15-20 - additive clipping; not bit-identical to clamp for huge |x|."""
    x = x + torch.relu(lower - x)
    x = x - torch.relu(x - upper)
    return x


def fq_rootq_act(x, run_scale, lo, hi):
    """Form E - RootQ/base.py:99,106,108-111: clip to [0, s*(hi-lo)], q = ste_round(xc / s), y = q*s."""
    upper = run_scale * (hi - lo)
    xc = rootq_clip(x, upper, 0)
    q = ste_round(xc / run_scale)
    return q, q * run_scale


def fq_rootq_weight(w, upper, lower, alpha, lo, hi):
    """Form F - RootQ/base.py:146-155 + RootQ/function.py:22-32,58-67.  Returns (interval, sgn, y)."""
    wc = rootq_clip(w, upper, lower)
    delta = (upper - lower) / (hi - lo)
    interval = ste_floor((wc - lower) / delta)
    mi = (interval + 0.5) * delta + lower
    a = alpha + torch.relu(1e-4 - alpha)
    a = a - torch.relu(a - 1)
    d = wc - mi
    sg = d / (torch.abs(d) + 1e-5)
    k = 2 / delta
    phi = torch.pow(k * abs(d) + 1e-5, a) * sg
    s = phi.sgn()
    y = ((s + 1) / 2 + interval) * delta + lower
    return interval, s, y


def rootq_ema(run, param, momentum, g):
    """RootQ/base.py:95-97,137-140: r = run*(1-m) + m*param, then g*r + (1-g)*r (forward value)."""
    r = run.mul(1 - momentum).add(momentum * param)
    return g * r + (1 - g) * r


# ------------------------------------------------------------------------------ observers
def minmax_tensor(x, n_bits, signed, allow_offset=True):
    """ops.py:20-34.  Returns (scale, offset) as fp32 zero-dim tensors (the reference's signed
    branch returns an int64 `tensor(0)`; it only ever enters fp32 arithmetic)."""
    if signed:
        return x.abs().max() / (2 ** (n_bits - 1) - 1), torch.zeros((), dtype=torch.float32)
    mn = x.min()
    if not allow_offset:
        assert bool((mn >= 0).all())
        mn = torch.zeros((), dtype=torch.float32)
    mx = x.max()
    return (mx - mn) / (2 ** n_bits - 1), mn


def minmax_channel(x, n_bits, signed, ch_axis=0, allow_offset=True):
    """ops.py:112-140.  Scale/offset come back shaped [1,..,C,..,1]."""
    shape = [1] * x.dim()
    shape[ch_axis] = -1
    rows = x.transpose(0, ch_axis).reshape(x.shape[ch_axis], -1)
    if signed:
        scale = rows.abs().max(dim=1)[0] / (2 ** (n_bits - 1) - 1)
        offset = torch.zeros_like(scale)
    else:
        mn = rows.min(dim=1)[0]
        if not allow_offset:
            assert bool((mn >= 0).all())
            mn = torch.zeros_like(mn)
        scale = (rows.max(dim=1)[0] - mn) / (2 ** n_bits - 1)
        offset = mn
    return scale.reshape(shape), offset.reshape(shape)


def minmax_pixel(x, n_bits, signed, allow_offset=True):
    """ops.py:142-167: one (scale, offset) per kernel position, reduced over out- and in-channels.  The unsigned branch takes
    the minimum of |x| (ops.py:156: `tensor.abs().min`), kept as the reference has it."""
    new_shape = [x.shape[2], x.shape[3]] if x.dim() == 4 else [x.shape[2]]
    t = x.reshape(x.shape[0], x.shape[1], -1)
    if signed:
        scale = t.abs().max(dim=0)[0].max(dim=0)[0] / (2 ** (n_bits - 1) - 1)
        offset = torch.zeros_like(scale)
    else:
        mn = t.abs().min(dim=0)[0].min(dim=0)[0]
        mx = t.abs().max(dim=0)[0].max(dim=0)[0]
        if not allow_offset:
            assert bool((mn >= 0).all())
            mn = torch.zeros_like(mn)
        scale = (mx - mn) / (2 ** n_bits - 1)
        offset = mn
    return scale.reshape(new_shape), offset.reshape(new_shape)


def lsq_init(x, qmax):
    """modules/base.py:84-85 (input), :118-121 (weight): the LSQ first-call scale 2 * mean|x| / sqrt(Qp); the offset is zero."""
    import math
    return 2 * x.detach().abs().mean() / math.sqrt(qmax)


def l2_loss(a, b):
    """trainer/loss/loss.py:22-24."""
    return ((a - b) ** 2).sum(axis=1).mean()


def quantize_codes(x, scale, offset, lo, hi):
    """utils.py:1-2 `quantize` (codes only, with the +1e-7 divisor)."""
    return ((x - offset) / (scale + 1e-7)).round().clamp(lo, hi)


def l2norm_tensor(x, n_bits, signed, max_iter=100000):
    """ops.py:71-83."""
    scale, offset = minmax_tensor(x, n_bits, signed)
    lo, hi = qrange(signed, n_bits)
    diff = float("inf")
    it = 0
    while diff > 1e-5 and it < max_iter:
        q = quantize_codes(x, scale, offset, lo, hi)
        new = (x * q).sum() / (q * q + 1e-7).sum()
        diff = float((new - scale).abs() / scale)
        scale = new
        it += 1
    return scale, offset


def l2norm_channel(x, n_bits, signed, ch_axis=0, max_iter=100000):
    """ops.py:198-215."""
    shape = [1] * x.dim()
    shape[ch_axis] = -1
    rows = x.transpose(0, ch_axis).reshape(x.shape[ch_axis], -1)
    scale, offset = minmax_channel(rows, n_bits, signed, ch_axis=0)
    lo, hi = qrange(signed, n_bits)
    diff = float("inf")
    it = 0
    while diff > 1e-5 and it < max_iter:
        q = quantize_codes(rows, scale, offset, lo, hi)
        new = ((rows * q).sum(axis=1) / (q * q + 1e-7).sum(axis=1)).reshape(scale.shape)
        diff = float(((new - scale) ** 2).sum().sqrt() / (scale ** 2).sum().sqrt())
        scale = new
        it += 1
    return scale.reshape(shape), offset.reshape(shape)


def l2loss_tensor(x, n_bits, signed):
    """ops.py:36-68: 80-step shrink search with a rounded integer zero point (unsigned only)."""
    if signed:
        return minmax_tensor(x, n_bits, True)
    mn, mx = x.min(), x.max()
    qmax = 2 ** n_bits - 1
    best = 1000
    scale, offset = mx / qmax, torch.zeros(())
    for i in range(80):
        nmx, nmn = (1 - 0.01 * i) * mx, (1 - 0.01 * i) * mn
        ns = (nmx - nmn) / qmax
        nz = torch.round(-nmn / ns)
        q = torch.round(x / ns) + nz
        q = (q.clamp(0, qmax) - nz) * ns
        loss = l2_loss(q, x)
        if loss < best:
            best, scale, offset = loss, ns, nz
    return scale, offset


def l2loss_channel(x, n_bits, signed, ch_axis=0):
    """ops.py:169-196 (the C x 80 Python double loop).  Faithful to an aliasing quirk of the
    reference: `min_val = offset` (:172) is the SAME tensor that `offset[c] = new_offset` (:193)
    writes, so after the first accepted step the channel's "min" is its zero point."""
    shape = [1] * x.dim()
    shape[ch_axis] = -1
    rows = x.transpose(0, ch_axis).reshape(x.shape[ch_axis], -1)
    scale, offset = minmax_channel(rows, n_bits, signed, ch_axis=0)
    scale, offset = scale.clone(), offset.clone()
    qmax = 2 ** n_bits - 1
    mn = offset  # alias, on purpose (see docstring)
    mx = offset + scale * qmax
    for c in range(rows.shape[0]):
        best = 1000
        for i in range(80):
            nmn, nmx = (1 - 0.01 * i) * mn[c], (1 - 0.01 * i) * mx[c]
            ns = (nmx - nmn) / qmax
            nz = torch.round(-nmn / ns)
            q = (torch.round(rows[c] / ns) + nz).clamp(0, qmax)
            q = (q - nz) * ns
            loss = l2_loss(rows[c].view(1, -1), q.view(1, -1))
            if best > loss:
                scale[c], offset[c], best = ns, nz, loss
    return scale.reshape(shape), offset.reshape(shape)


# ------------------------------------------------------------------- LSQ backward (K7 spec)
def lsq_backward(w, scale, gout, lo, hi, g):
    """modules/function.py:37-49 `FunLSQ.backward` (closed form; offset is not subtracted there)."""
    qw = w / scale
    m_lo = (qw < lo).float()
    m_hi = (qw > hi).float()
    m_mid = 1.0 - m_lo - m_hi
    gs = ((lo * m_lo + hi * m_hi + m_mid * (-qw + qw.round())) * gout).sum().unsqueeze(0) * g
    return m_mid * gout, gs


def qbase_backward(x, scale, offset, gy, lo, hi, g):
    """What autograd executes through modules/base.py:96-102 (the live path), node by node.
    Forward: v = (x - o)/s^; c = clamp(v); r = ste_round(c); y = r*s^ + o.  Backward of `gy`:
      mul:   g_r = gy * s^                     d/ds^ += sum(gy * r)
      clamp: g_v = where(lo <= v <= hi, g_r, 0)
      div:   g_x = g_v / s^                    d/ds^ += sum(-g_v * (v / s^))
      s^ = ste_scale(s, g) has d s^/d s = g    -> grad_s = g * (d/ds^)
    `g_x` is therefore (gy * s^) / s^ inside the range - two roundings, not `gy` - and exactly +0
    outside.  Returns (g_x, grad_s); g_x is bit-exact, grad_s is an fp32 sum (order-dependent)."""
    s_hat = ste_scale(scale, g)
    v = (x - offset) / s_hat
    r = ste_round(v.clamp(lo, hi))
    inside = (v >= lo) & (v <= hi)
    g_v = torch.where(inside, gy * s_hat, torch.zeros((), dtype=gy.dtype))
    gx = g_v / s_hat
    gs = ((gy * r).sum() + (-g_v * (v / s_hat)).sum()) * g
    return gx, gs


def fsptq_act_backward(x, scale, zp, gy, lo, hi):
    """What autograd executes through FSPTQuant/base.py:108-109 (activations), node by node.
    Forward: v = x / s; r = ste_round(v); c = clamp(r + zp, lo, hi); y = (c - zp) * s.  Backward of `gy`:
      mul:   g_c = gy * s                      d/ds += sum(gy * (c - zp))
      clamp: g_r = where(lo <= r + zp <= hi, g_c, 0)       (ste_round and the additions pass gradients through)
      div:   g_x = g_r / s                     d/ds += sum(-g_r * (v / s))
    Returns (g_x, grad_s): g_x bit-exact ((gy * s) / s inside the range, +0 outside), grad_s an fp32 sum."""
    v = x / scale
    pre = ste_round(v) + zp
    c = pre.clamp(lo, hi)
    inside = (pre >= lo) & (pre <= hi)
    g_r = torch.where(inside, gy * scale, torch.zeros((), dtype=gy.dtype))
    gx = g_r / scale
    gs = (gy * (c - zp)).sum() + (-g_r * (v / scale)).sum()
    return gx, gs


def fsptq_weight_backward(w, scale, gy, lo, hi):
    """The same through FSPTQuant/base.py:149-152 (weights, per-output-channel scale [K, 1, 1, 1] or [K, 1]):
    v = w / s; c = clamp(ste_round(v), lo, hi); y = c * s.  Returns (g_w, grad_s[K...]) with grad_s summed per channel."""
    v = w / scale
    r = ste_round(v)
    c = r.clamp(lo, hi)
    inside = (r >= lo) & (r <= hi)
    g_r = torch.where(inside, gy * scale, torch.zeros((), dtype=gy.dtype))
    gw = g_r / scale
    dims = tuple(range(1, w.dim()))
    gs = (gy * c).sum(dim=dims, keepdim=True) + (-g_r * (v / scale)).sum(dim=dims, keepdim=True)
    return gw, gs


# --------------------------------------------------------------------------- layer forwards
def conv_or_linear(layer, x_q, w_q):
    """modules/conv.py:13-19, modules/linear.py:12-13 (and the FSPTQuant/RootQ copies)."""
    import torch.nn.functional as F
    if isinstance(layer, torch.nn.Linear):
        return F.linear(x_q, w_q, layer.bias)
    if layer.padding_mode != "zeros":
        return F.conv2d(F.pad(x_q, layer._reversed_padding_repeated_twice, mode=layer.padding_mode),
                        w_q, layer.bias, layer.stride, (0, 0), layer.dilation, layer.groups)
    return F.conv2d(x_q, w_q, layer.bias, layer.stride, layer.padding, layer.dilation, layer.groups)


def qbase_layer_forward(layer, x, in_scale, in_offset, wt_scale, wt_offset, in_rng, wt_rng):
    """modules/base.py:67-140 steady state (scales frozen): returns (x_q, w_q, out)."""
    g_i = 1 / math.sqrt(x.numel() * in_rng[1])
    g_w = 1 / math.sqrt(layer.weight.numel() * wt_rng[1])
    _, xq = fq_qbase(x, in_scale, in_offset, in_rng[0], in_rng[1], g_i)
    _, wq = fq_qbase(layer.weight.detach(), wt_scale, wt_offset, wt_rng[0], wt_rng[1], g_w)
    return xq, wq, conv_or_linear(layer, xq, wq)


def fsptq_layer_forward(layer, x, in_scale, in_zp, wt_scale, in_rng, wt_rng):
    """FSPTQuant/base.py:95-159 steady state, recon_type None: returns (x_q, w_q, out)."""
    _, xq = fq_zeropoint(x, in_scale, in_zp, in_rng[0], in_rng[1])
    _, wq = fq_symmetric(layer.weight.detach(), wt_scale, wt_rng[0], wt_rng[1])
    return xq, wq, conv_or_linear(layer, xq, wq)


# ------------------------------------------------- weight transforms that precede the path
def fold_bn(weight, bias, gamma, beta, mean, var):
    """dlmc/utils/merge_bn.py:85-101: note var + 1e-7 (not bn.eps).  Returns (weight', bias')."""
    v = var + 1e-7
    cout = weight.shape[0]
    if bias is None or bias.numel() == 0:
        bias = torch.zeros(cout)
    b = gamma * (bias - mean) / v.sqrt() + beta
    w = (weight.reshape(cout, -1) * gamma.reshape(-1, 1) / v.sqrt().reshape(-1, 1)).reshape(weight.shape)
    return w, b


def repvgg_fuse(k3, bn3, k1, bn1, bnid, groups=1):
    """model/classification/repvgg.py:92-130.  bn* = (gamma, beta, mean, var, eps); bnid may be None."""
    def branch(kernel, bn):
        gamma, beta, mean, var, eps = bn
        std = (var + eps).sqrt()
        return kernel * (gamma / std).reshape(-1, 1, 1, 1), beta - mean * gamma / std
    ka, ba = branch(k3, bn3)
    kb, bb = branch(k1, bn1)
    kb = torch.nn.functional.pad(kb, [1, 1, 1, 1])
    if bnid is None:
        return ka + kb + 0, ba + bb + 0
    cin = k3.shape[0]
    cg = cin // groups
    ident = torch.zeros(cin, cg, 3, 3)
    for i in range(cin):
        ident[i, i % cg, 1, 1] = 1
    kc, bc = branch(ident, bnid)
    return ka + kb + kc, ba + bb + bc


def l2norm_output(layer, x, weight, n_bits, signed, patience=1000):
    """ops.py:85-109: refine the (per-tensor) weight scale against the layer output."""
    out = conv_or_linear(layer, x, weight)
    scale, offset = minmax_tensor(weight, n_bits, signed)
    lo, hi = qrange(signed, n_bits)
    diff, best_mse, best_scale, count = float("inf"), float("inf"), scale, 0
    while diff > 1e-5 and count != patience:
        wq = quantize_codes(weight, scale, offset, lo, hi)
        oq = conv_or_linear(layer, x, wq)
        mse = l2_loss(out, oq)
        new = (oq * out).mean(axis=0).sum() / (oq * oq + 1e-7).mean(axis=0).sum()
        diff = float((new - scale).abs() / scale)
        scale = new
        if mse < best_mse:
            best_mse, best_scale = mse, scale
        count += 1
    return best_scale, offset
