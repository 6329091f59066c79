"""Frozen-network execution plan: fold what FOLLOWS each quantised layer into that layer's int8 kernel.

In the reference every wrapper is an island (`modules/conv.py:13-19`, `FSPTQuant/base.py:95-159`): it reads an
fp32 activation, fake-quantises it, convolves, and writes an fp32 activation which a ReLU module re-reads and
re-writes, a residual add re-reads and re-writes, and the next wrapper re-reads to quantise again.  Once the
scales are frozen (after calibration / PTQ - the state `post_training_quantization.py:77` evaluates in) none of
those trips through HBM is needed: the value that leaves the matrix-core accumulator can be dequantised, added to
the shortcut, rectified and turned into the NEXT layer's activation code while it is still in a register
(`dlmcq_conv2d_i8_nhwc_fused`).  Arithmetic and order are those of the separate kernels, so the codes - and hence
every downstream value - are bit-identical to the unfused wrappers; only the memory traffic changes.

    model = ...; quantize_model(model, cfg, logger, "FSPTQ"); model(calib_batch)      # calibrated, on the GPU
    fused = fuse_inference(model)            # torch.fx GraphModule over the same parameters
    y = fused(x)                             # == model(x) bit for bit (up to the sign of zero)

`torch.fx` is used once, as a dataflow reader (which ReLU / add / wrapper consumes which tensor); execution is the
explicit HIP launches of the plan, and the result can be captured with `dlmc.utils.graph.GraphedForward`.
The plan snapshots weights and scales: re-fuse after changing them.
"""
import math
import operator

import torch
import torch.fx as fx
import torch.nn.functional as F
from torch import nn

from .. import _native as N
from ..quantization.scalar import kernels as K
from ..quantization.scalar._wrapper import int8_kind, ste_scale_value
from ..quantization.scalar.FSPTQuant.base import FSPTQBase
from ..quantization.scalar.modules.base import QBase
from ..quantization.scalar.RootQ.base import RootQBase

__all__ = ["fuse_inference", "StreamedPlan", "Int8Layer", "DualInt8Layer", "StemLayer", "FusionReport"]


# ---------------------------------------------------------------------------------- frozen quantiser specs
class _ActSpec:
    """One wrapper's frozen activation quantiser, as a producer has to evaluate it."""

    def __init__(self, scale, zp, lo, hi, form, needs_g):
        self.scale, self.zp, self.lo, self.hi, self.form, self.needs_g = scale, zp, int(lo), int(hi), form, needs_g
        self.key = (form, self.lo, self.hi, float(scale.reshape(-1)[0]), 0.0 if zp is None else float(zp.reshape(-1)[0]),
                    needs_g)
        # what a PRODUCER is told: a zero point of 0 (every post-ReLU tensor) goes as "none" - the kernels' plain-quantiser paths
        # (csrc/conv_epilogue.h epi_plain) key on the null pointer (read once, when the plan is built)
        self.zp_emit = None if self.key[4] == 0.0 else zp

    def g(self, numel):
        return 1 / math.sqrt(numel * self.hi) if self.needs_g else 0.0

    def emit(self, numel):
        return K.EmitCodes(self.scale, self.zp_emit, self.lo, self.hi, self.form, self.g(numel))


def _byte_range(lo, hi):
    return (0 <= lo and hi <= 255) or (-128 <= lo and hi <= 127)


def _ceil64(n):
    return (n + 63) // 64 * 64


def plan_kind(mod):
    """Which kernel of the frozen plan runs `mod`: "gemm" = conv_i8.hip (dense conv / linear; channel counts that are no
    multiple of 64 are zero-padded when the plan is built), "stem" = conv_stem_i8.hip (<= 4 input channels), "dw" =
    conv_dw_i8.hip (depthwise), or None."""
    k = int8_kind(mod)
    if k is not None:
        return k
    w = mod.weight
    sq = lambda t: len(set(t)) == 1  # noqa: E731
    if w.dim() != 4 or mod.padding_mode != "zeros" or isinstance(mod.padding, str) or not (sq(mod.stride) and sq(mod.padding)):
        return None
    if tuple(mod.dilation) != (1, 1):
        return None
    if mod.groups == 1 and w.shape[1] % 4 == 0 and w.shape[1] > 4:
        return "gemm"
    if mod.groups == w.shape[0] and w.shape[1] == 1 and w.shape[0] % 4 == 0 and w.shape[2] <= 7 and w.shape[3] <= 7:
        return "dw"
    return None


def _frozen_spec(mod):
    """(_ActSpec, weight scale, weight lo, weight hi, kind, weight offset, weight-code function) if `mod` can run on an
    int8 kernel with frozen scales.  The last two are None for symmetric per-tensor / per-channel weights quantised by
    quantize_weight_krsc; asymmetric (offset = channel minimum, ops.py:129-136) or QBase per-channel weights carry their
    float offsets [K] and a function returning the module's own integer codes [K, C, R, S]."""
    kind = plan_kind(mod)
    if kind is None:
        return None
    spec = _frozen_spec_(mod, kind)
    if spec is None:
        return None
    if len(spec) == 4:
        spec = spec + (None, None)
    return spec[:4] + (kind,) + spec[4:]


def _frozen_spec_(mod, kind):
    if isinstance(mod, FSPTQBase):
        if not (mod.act_quant and mod.wt_quant) or mod.in_scale.numel() != 1:
            return None
        if mod.qconfig["weight"].get("recon_type") in ("adaround", "dist_recon"):
            return None
        if not (mod._init.ready(mod, "in_init_state") and mod._init.ready(mod, "wt_init_state")):
            raise RuntimeError("fuse_inference: run a calibration forward first (scales are not initialised)")
        if not (_byte_range(mod.in_min_val, mod.in_max_val) and -128 <= mod.wt_min_val and mod.wt_max_val <= 127):
            return None
        zp = mod.in_offset.detach().to(torch.float32).reshape(-1)[:1].clone()
        z = float(zp[0])
        if z != round(z) or not (mod.in_min_val <= z <= mod.in_max_val):
            return None
        act = _ActSpec(mod.in_scale.detach().reshape(-1)[:1].clone(), zp, mod.in_min_val, mod.in_max_val,
                       N.FORM_ZEROPOINT, False)
        return act, mod.wt_scale.detach().clone(), mod.wt_min_val, mod.wt_max_val
    if isinstance(mod, QBase):
        cfg = mod.qconfig
        if not (cfg["input"]["enable"] and cfg["weight"]["enable"]):
            return None
        k = mod.weight.shape[0]
        per_channel = mod.wt_scale.numel() != 1
        if mod.in_scale.numel() != 1 or (per_channel and (mod.wt_scale.numel() != k or mod.wt_scale.shape[0] != k)):
            return None
        if not (mod._init.ready(mod, "in_init_state") and mod._init.ready(mod, "wt_init_state")):
            raise RuntimeError("fuse_inference: run a calibration forward first (scales are not initialised)")
        lo, hi = mod.wt_min_val, mod.wt_max_val
        if not (_byte_range(mod.in_min_val, mod.in_max_val) and _byte_range(lo, hi)):
            return None
        if float(mod.in_offset.abs().max()) != 0:
            return None                  # a float activation offset has no integer zero point (padding must be a code)
        asym = mod.wt_offset is not None and float(mod.wt_offset.abs().max()) != 0
        act = _ActSpec(mod.in_scale.detach().reshape(-1)[:1].clone(), None, mod.in_min_val, mod.in_max_val,
                       N.FORM_QBASE, True)
        g_w = 1 / math.sqrt(mod.weight.numel() * hi)
        s_hat = ste_scale_value(mod.wt_scale, g_w).clone()
        if not (per_channel or asym or hi > 127 or kind == "dw"):
            return act, s_hat, lo, hi
        w_off = mod.wt_offset.detach().to(torch.float32).reshape(-1).clone() if asym else None

        def codes(mod=mod, g_w=g_w, lo=lo, hi=hi):      # the module's own weight quantiser (form QBASE), as integers
            off = mod.wt_offset if mod.wt_offset is not None else None
            return K.fake_quant(mod.weight.detach(), mod.wt_scale.detach(), off, lo, hi, N.FORM_QBASE, g=g_w, codes="i8",
                                want_y=False)[1]
        return act, s_hat, lo, hi, w_off, codes
    return None


# -------------------------------------------------------------------------------------------- plan nodes
class _PlanLayer(nn.Module):
    """Common part of the plan nodes: frozen quantiser constants, the consumer's emit spec, pooling on codes."""

    def __init__(self, layer, spec, relu=False, emit=None, want_out=True, pool=None):
        super().__init__()
        self.layer = layer
        self.act, w_scale, self.w_lo, self.w_hi, self.kind, w_off, self._w_codes = spec
        self.relu, self.emit, self.want_out, self.pool = bool(relu), emit, bool(want_out), pool
        k = layer.weight.shape[0]
        # channel counts that are no multiple of 64 (the K step of the matrix-core kernel) are zero-padded: padded output
        # channels have zero weights and bias, so their value is 0 and their code is the consumer's code of 0
        self.k, self.k_pad = k, (_ceil64(k) if self.kind in ("gemm", "dw") and layer.weight.dim() == 4 else k)
        w_scale = w_scale.detach().to(torch.float32).reshape(-1)
        w_scale = w_scale.expand(k) if w_scale.numel() == 1 else w_scale
        self.register_buffer("w_scale", self._padk(w_scale, 1.0), persistent=False)
        if w_off is not None:
            w_off = w_off.expand(k) if w_off.numel() == 1 else w_off
        self.register_buffer("w_off", None if w_off is None else self._padk(w_off.to(w_scale.device), 0.0), persistent=False)
        self.register_buffer("bias_pad", None if layer.bias is None or self.k_pad == k else self._padk(layer.bias.detach().float(), 0.0),
                             persistent=False)
        self._zp_fill = int(0 if self.act.zp is None else float(self.act.zp.reshape(-1)[0]))   # (read once: no host sync in forward)
        # A producer may hand an unsigned-byte quantiser's codes over as int8 `code - 128` (EmitCodes.shift128: what the matrix
        # cores multiply anyway, so the consumer's kernel need not re-centre every operand byte it reads); this node then runs
        # with the zero point `zp - 128` - the same integers.  `emit_shift`: this node emits ITS consumers' codes that way.
        self.emit_shift = False
        zs = None
        if self.act.lo >= 0:
            zs = (torch.zeros(1, device=w_scale.device) if self.act.zp is None else self.act.zp.detach().float().reshape(-1)[:1]) - 128.0
        self.register_buffer("zp_shift", zs, persistent=False)
        self._deq = {}     # QBase dequantises with s^ = grad_scale(s, g(numel)): one tiny tensor per input size

    def _padk(self, v, fill):
        v = v.detach().reshape(-1)
        if self.k_pad == v.numel():
            return v.contiguous()
        return torch.cat([v, torch.full((self.k_pad - v.numel(),), fill, dtype=v.dtype, device=v.device)]).contiguous()

    def _bias(self):
        return self.layer.bias if self.bias_pad is None else self.bias_pad

    def _in_scale(self, numel):
        if not self.act.needs_g:
            return self.act.scale
        s = self._deq.get(numel)
        if s is None:
            s = self._deq[numel] = ste_scale_value(self.act.scale, self.act.g(numel)).contiguous()
        return s

    def _emit_for(self, n, k, p, q):
        """The consumer's quantiser; its g (QBase) is taken over ITS input = this node's (pooled) output."""
        if self.emit is None:
            return None
        if self.pool is not None:
            kk, ss, pp = self.pool
            p, q = (p + 2 * pp - kk) // ss + 1, (q + 2 * pp - kk) // ss + 1
        e = self.emit.emit(n * k * p * q)
        e.shift128 = self.emit_shift
        return e

    def _zp(self, codes):
        """The zero point that goes with `codes`: int8 codes of an unsigned quantiser are shifted codes (see __init__)."""
        return self.zp_shift if (codes.dtype == torch.int8 and self.act.lo >= 0) else self.act.zp

    def _finish(self, out, codes):
        if self.pool is not None:
            codes = K.maxpool_codes(codes, *self.pool)
        if out is not None and self.k_pad != self.k:
            out = out[:, :self.k]          # fp32 leaves the plan: drop the padding channels (codes stay padded for plan consumers)
        return out, codes


def _pad_channels(codes, c_pad, fill):
    """Activation codes (N, C, H, W) channels_last -> (N, c_pad, H, W), the new channels holding `fill` (the code of 0)."""
    n, c, h, w = codes.shape
    if c == c_pad:
        return codes
    out = torch.full((n, c_pad, h, w), fill, dtype=codes.dtype, device=codes.device).contiguous(memory_format=torch.channels_last)
    out[:, :c] = codes
    return out


def _weight_codes(node):
    """Integer weight codes [K, C, R, S] (int16) of a plan node whose spec carries its own quantiser, with codes above 127
    re-centred (qw - 128, offset + 128 * s: the matrix cores multiply signed bytes)."""
    q = node._w_codes().to(torch.int16)
    if q.dim() == 2:
        q = q[:, :, None, None]
    if node.w_hi > 127:
        q = q - 128
        shift = 128.0 * node.w_scale[:node.k]
        base = node.w_off[:node.k] if node.w_off is not None else torch.zeros_like(shift)
        node.w_off = node._padk(base + shift, 0.0)
    return q


class Int8Layer(_PlanLayer):
    """One quantised conv / linear of the frozen plan (input channels % 64 == 0).  Input: the producer's codes
    (uint8/int8) or an fp32 tensor (quantised here, one pass).  Output: `(fp32 or None, consumer codes or None)`."""

    def __init__(self, layer, spec, **kw):
        super().__init__(layer, spec, **kw)
        w = layer.weight.detach()
        c = w.shape[1]
        self.c, self.c_pad = c, (_ceil64(c) if w.dim() == 4 else c)
        if self._w_codes is None and self.c_pad == c and self.k_pad == self.k:
            wq, wsum = K.quantize_weight_krsc(w, self.w_scale, self.w_lo, self.w_hi)
        else:
            if self._w_codes is not None:
                q = _weight_codes(self)                                            # [K, C, R, S] int16
            else:
                q4 = w if w.dim() == 4 else w[:, :, None, None]
                q = K.quantize_weight_krsc(q4, self.w_scale[:self.k], self.w_lo, self.w_hi)[0].permute(0, 3, 1, 2).to(torch.int16)
            full = torch.zeros((self.k_pad, self.c_pad) + tuple(q.shape[2:]), dtype=torch.int16, device=q.device)
            full[:self.k, :c] = q
            wq = full.permute(0, 2, 3, 1).contiguous().to(torch.int8)                # KRSC
            wsum = full.sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
        self.register_buffer("wq", wq, persistent=False)
        self.register_buffer("wsum", wsum, persistent=False)

    def _codes(self, x):
        """The activation codes of `x` (already codes, or fp32 quantised here in one pass), channel-padded for the kernel."""
        act = self.act
        if x.dtype not in (torch.uint8, torch.int8):
            N.require_gpu(x)
            if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            x = K.fake_quant(x, act.scale, act.zp, act.lo, act.hi, act.form, g=act.g(x.numel()), codes="i8", want_y=False)[1]
        c_pad = getattr(self, "c_pad", x.shape[1])
        if x.dim() == 4 and x.shape[1] != c_pad:
            x = _pad_channels(x, c_pad, self._zp_fill)
        return x

    def _real_numel(self, codes):
        """Elements of the layer's real (unpadded) input: QBase's grad_scale factor is defined on it."""
        return codes.numel() // codes.shape[1] * self.c if codes.dim() == 4 else codes.numel()

    def _conv_kw(self):
        lay = self.layer
        return {} if lay.weight.dim() == 2 else dict(stride=lay.stride[0], padding=lay.padding[0], dilation=lay.dilation[0])

    def _out_hw(self, codes):
        lay = self.layer
        if lay.weight.dim() == 2:
            return 1, 1
        st, pd, dl = lay.stride[0], lay.padding[0], lay.dilation[0]
        r, s = lay.weight.shape[2], lay.weight.shape[3]
        return (codes.shape[2] + 2 * pd - dl * (r - 1) - 1) // st + 1, (codes.shape[3] + 2 * pd - dl * (s - 1) - 1) // st + 1

    def operand(self, x):
        """This layer as one addend of conv2d_i8_dual."""
        codes = self._codes(x)
        return dict(codes=codes, wq=self.wq, wsum=self.wsum, bias=self._bias(), in_scale=self._in_scale(self._real_numel(codes)),
                    in_zp=self._zp(codes), w_scale=self.w_scale, **self._conv_kw())

    def forward(self, x, residual=None):
        lay, act = self.layer, self.act
        linear = lay.weight.dim() == 2
        lead = None
        if linear and x.dim() != 2:
            lead = x.shape[:-1]
            x = x.reshape(-1, x.shape[-1])
            if residual is not None:
                residual = residual.reshape(-1, residual.shape[-1])
        codes = self._codes(x)
        numel = self._real_numel(codes)
        emit = self._emit_for(codes.shape[0], lay.weight.shape[0], *self._out_hw(codes))
        kw = self._conv_kw()
        if self.w_off is not None:
            kw["w_offset"] = self.w_off
        if self.relu or residual is not None or emit is not None or self.w_off is not None:
            res = K.conv2d_i8(codes, self.wq, self.wsum, self._bias(), self._in_scale(numel), self._zp(codes), self.w_scale,
                              residual=residual, relu=self.relu, emit=emit, want_out=self.want_out,
                              out_chunk_major=getattr(self, "out_cm", False), **kw)
            out, out_codes = res if emit is not None else (res, None)
        else:
            out, out_codes = K.conv2d_i8(codes, self.wq, self.wsum, self._bias(), self._in_scale(numel), self._zp(codes), self.w_scale, **kw), None
        if lead is not None:
            out = None if out is None else out.reshape(*lead, out.shape[-1])
            out_codes = None if out_codes is None else out_codes.reshape(*lead, out_codes.shape[-1])
        return self._finish(out, out_codes)


class DwInt8Layer(Int8Layer):
    """A depthwise convolution of the frozen plan (csrc/conv_dw_i8.hip): codes in, codes (and / or fp32) out."""

    def __init__(self, layer, spec, **kw):
        _PlanLayer.__init__(self, layer, spec, **kw)
        w = layer.weight.detach()
        self.c, self.c_pad = self.k, self.k_pad
        if self._w_codes is not None:
            q = _weight_codes(self)                                                # [C, 1, R, S]
        else:
            q = K.quantize_weight_krsc(w, self.w_scale[:self.k], self.w_lo, self.w_hi)[0].permute(0, 3, 1, 2).to(torch.int16)
        full = torch.zeros((self.k_pad,) + tuple(q.shape[2:]), dtype=torch.int16, device=q.device)
        full[:self.k] = q[:, 0]
        self.register_buffer("wq", full.permute(1, 2, 0).contiguous().to(torch.int8), persistent=False)   # [R, S, C]

    def forward(self, x):
        lay, act = self.layer, self.act
        codes = self._codes(x)
        numel = self._real_numel(codes)
        emit = self._emit_for(codes.shape[0], self.k, *self._out_hw(codes))
        res = K.conv2d_dw_i8(codes, self.wq, self._bias(), self._in_scale(numel), self._zp(codes), self.w_scale, self.w_off,
                             stride=lay.stride[0], padding=lay.padding[0], relu=self.relu, emit=emit, want_out=self.want_out)
        out, out_codes = res if emit is not None else (res, None)
        return self._finish(out, out_codes)


class DualInt8Layer(nn.Module):
    """`conv_a(x) + conv_b(y)` -> (ReLU) -> codes as ONE kernel: the last convolution of a residual block's first
    unit and the convolution on its shortcut.  Neither addend is written to memory."""

    def __init__(self, a, b):
        super().__init__()
        self.a, self.b = a, b          # `a` carries the epilogue options (relu / emit / want_out / pool)

    def forward(self, x, y):
        a = self.a
        oa, ob = a.operand(x), self.b.operand(y)
        emit = a._emit_for(oa["codes"].shape[0], a.layer.weight.shape[0], *a._out_hw(oa["codes"]))
        res = K.conv2d_i8_dual(oa, ob, relu=a.relu, emit=emit, want_out=a.want_out, out_chunk_major=getattr(self, "out_cm", False))
        out, out_codes = res if emit is not None else (res, None)
        return a._finish(out, out_codes)


class ChainInt8Layer(nn.Module):
    """A residual block's last 1x1 convolution (+ shortcut, ReLU) and the next block's first 1x1 convolution as ONE kernel
    (csrc/conv_chain_i8.hip): the activation codes between them stay in LDS.  The shortcut is an fp32 tensor, or - `short`,
    the first block of a stage - a second 1x1 convolution reduced into the same tile.  Returns
    `(fp32 or None, codes or None, codes2)`; shapes the kernel is not built for run the plan nodes one after the other."""

    def __init__(self, a, b, want_codes, short=None, main=None):
        super().__init__()
        # `a` carries the epilogue options; with a convolution shortcut, `main` / `short` are the unit-stride and the (possibly)
        # strided operand of the sum (fp32 addition commutes: which of the two the dual plan node called `a` does not matter)
        self.a, self.b, self.want_codes, self.short, self.main = a, b, bool(want_codes), short, (main if main is not None else a)
        self.swapped = short is not None and self.main is not a       # main reads the plan node's SECOND input
        self._w2cm = None      # the second layer's weight codes chunk-major (K.chunk_major), made at the first forward
        self.out_cm = False    # the fp32 block output as a K.ChunkMajor (set by _block_layout_pass where only chain kernels read it)

    def forward(self, x, y):
        a, b, sc, mn = self.a, self.b, self.short, self.main
        if self.swapped:
            x, y = y, x
        codes = mn._codes(x)
        n, c, h, w = codes.shape
        nxt = dict(wq=b.wq, wsum=b.wsum, bias=b._bias(), w_scale=b.w_scale)
        if not getattr(b, "_packed4", False):     # (4-bit weights live packed and are expanded per forward: they stay KRSC)
            if self._w2cm is None or self._w2cm.device != b.wq.device:
                self._w2cm = K.chunk_major(b.wq)  # the plan's weights are frozen: made once
            nxt["wq_chunk"] = self._w2cm
        kw = dict(relu=a.relu, emit=a._emit_for(n, a.k, h, w), want_out=a.want_out, want_codes=self.want_codes, emit2=b._emit_for(n, b.k, h, w),
                  out_chunk_major=self.out_cm)
        if sc is None:
            if not K.chain_supported(c, a.k, b.k, n * h * w):
                if isinstance(y, K.ChunkMajor):      # (the plan nodes one after the other know row-major tensors only)
                    y = y.to_nhwc()
                out, mid = a(x, y)
                return out, (mid if self.want_codes else None), b(mid)[1]
            oa = dict(codes=codes, wq=a.wq, wsum=a.wsum, bias=a._bias(), in_scale=a._in_scale(a._real_numel(codes)), in_zp=a._zp(codes),
                      w_scale=a.w_scale)
            return K.conv2d_i8_chain(oa, nxt, y, relu2=b.relu, **kw)
        if not K.dual_chain_supported(c, sc.c, a.k, b.k, n * h * w):
            out, mid = DualInt8Layer(a, sc if mn is a else mn)(*((x, y) if mn is a else (y, x)))
            return out, (mid if self.want_codes else None), b(mid)[1]      # (row-major: every reader takes that)
        kw["emit3"] = kw.pop("emit2")
        return K.conv2d_i8_dual_chain(mn.operand(x), sc.operand(y), nxt, relu3=b.relu, **kw)


class DwPwInt8Layer(nn.Module):
    """A depthwise 3x3 / stride 1 / pad 1 layer and the pointwise layer that alone reads its codes as ONE kernel
    (csrc/conv_dwpw_i8.hip: the MobileOne / MobileNet unit; the wide code tensor between them stays in LDS).  Returns
    `(None, codes)` like a codes-only Int8Layer; inputs the kernel is not built for run the two plan nodes one after the other."""

    def __init__(self, dw, pw):
        super().__init__()
        self.dw, self.pw = dw, pw
        self._tables = {}      # (elements of the input, unsigned?) -> the depthwise constants in the kernel's layout (QBase: s^ depends on numel)

    def forward(self, x):
        dw, pw = self.dw, self.pw
        codes = dw._codes(x)
        n, c, h, w = codes.shape
        lay = dw.layer
        if not K.dwpw_supported(c, pw.k_pad, h, w, lay.stride[0], lay.padding[0], lay.weight.shape[2]) or pw.c_pad != c:
            return pw(dw(x)[1])
        numel = dw._real_numel(codes)
        key = (numel, codes.dtype == torch.uint8, dw.wq.data_ptr())
        table = self._tables.get(key)
        if table is None:
            self._tables.clear()
            table = self._tables[key] = K.dwpw_table(dw.wq, dw._bias(), dw._in_scale(numel), dw._zp(codes), dw.w_scale, dw.w_off,
                                                     x_unsigned=codes.dtype == torch.uint8)
        emit = dw._emit_for(n, dw.k, h, w)                          # the depthwise output's quantiser = the pointwise layer's input quantiser
        emit2 = pw._emit_for(n, pw.layer.weight.shape[0], h, w)
        op = dict(wq=pw.wq, wsum=pw.wsum, bias=pw._bias(), w_scale=pw.w_scale, w_offset=pw.w_off, in_scale=pw._in_scale(n * h * w * pw.c))
        out = K.conv2d_dwpw_i8(codes, table, dw.w_off is not None, dw._bias() is not None, dw.relu, dw._zp(codes), emit, op, relu=pw.relu, emit2=emit2)
        return pw._finish(None, out)


def _dwpw_pass(gm, report):
    """Depthwise 3x3 -> pointwise 1x1 (a MobileOne / MobileNet unit): replace the two plan nodes by one DwPwInt8Layer where the pointwise
    layer is the only reader of the depthwise layer's codes and both emit codes only."""
    graph = gm.graph
    modules = dict(gm.named_modules())
    count = 0
    for nd in list(graph.nodes):
        dw = modules.get(nd.target) if nd.op == "call_module" else None
        if type(dw) is not DwInt8Layer or len(nd.args) != 1:
            continue
        lay = dw.layer
        if not (tuple(lay.weight.shape[2:]) == (3, 3) and lay.stride[0] == 1 and lay.padding[0] == 1 and lay.dilation[0] == 1 and dw.pool is None and
                dw.emit is not None and not dw.want_out and (dw.emit.lo, dw.emit.hi) == (0, 255) and not dw.emit_shift and dw.k_pad % 64 == 0):
            continue
        gets = {u.args[1]: u for u in nd.users if u.op == "call_function" and u.target is operator.getitem}
        if len(gets) != len(nd.users) or 1 not in gets or (0 in gets and gets[0].users):
            continue
        g1 = gets[1]
        if len(g1.users) != 1:
            continue
        npw = next(iter(g1.users))
        pw = modules.get(npw.target) if npw.op == "call_module" else None
        if type(pw) is not Int8Layer or npw.args != (g1,):
            continue
        pl = pw.layer
        if not (pl.weight.dim() == 4 and tuple(pl.weight.shape[2:]) == (1, 1) and pl.stride[0] == 1 and pl.padding[0] == 0 and pw.pool is None and
                pw.emit is not None and not pw.want_out and not pw.emit_shift and pw.c_pad == dw.k_pad and pw.k_pad in K.DWPW_WIDTHS):
            continue
        pgets = {u.args[1]: u for u in npw.users if u.op == "call_function" and u.target is operator.getitem}
        if len(pgets) != len(npw.users) or (0 in pgets and pgets[0].users):
            continue
        name = f"_int8_dwpw_{count}"
        count += 1
        gm.add_module(name, DwPwInt8Layer(dw, pw))
        with graph.inserting_after(npw):
            nc = graph.call_module(name, args=nd.args)
        with graph.inserting_after(nc):
            codes = graph.call_function(operator.getitem, (nc, 1))
        if 1 in pgets:
            pgets[1].replace_all_uses_with(codes)
        for n in list(pgets.values()) + [npw] + list(gets.values()) + [nd]:
            graph.erase_node(n)
    report.dwpw = count
    if count:
        graph.eliminate_dead_code()
        graph.lint()
        gm.recompile()


def _pointwise(plan):
    """A plan node the chain kernel can take as either half: a plain 1x1 / stride 1 / unpadded int8 convolution."""
    lay = plan.layer
    return (type(plan) is Int8Layer and lay.weight.dim() == 4 and tuple(lay.weight.shape[2:]) == (1, 1) and lay.stride[0] == 1 and
            lay.padding[0] == 0 and plan.w_off is None and plan.pool is None and plan.k_pad == plan.k and plan.c_pad == plan.c and
            not plan.act.needs_g)


def _chain_pass(gm, report):
    """Block end -> next block's first 1x1: replace the two plan nodes by one ChainInt8Layer where the second reads nothing
    but the first's codes (other readers of those codes - a stage's downsample convolution - keep getting them)."""
    graph = gm.graph
    modules = dict(gm.named_modules())
    count = 0
    for na in list(graph.nodes):
        if na.op != "call_module" or len(na.args) != 2 or not isinstance(modules.get(na.target), (Int8Layer, DualInt8Layer)):
            continue
        a, main, short = modules[na.target], None, None
        if isinstance(a, DualInt8Layer):      # the first block of a stage: its shortcut is a (possibly strided) 1x1 convolution
            a, other = a.a, a.b
            main, short = (a, other) if _pointwise(a) else (other, a)
            lay = short.layer
            if not (_pointwise(main) and type(short) is Int8Layer and lay.weight.dim() == 4 and tuple(lay.weight.shape[2:]) == (1, 1) and
                    lay.padding[0] == 0 and short.w_off is None and short.pool is None and short.k_pad == short.k and
                    short.c_pad == short.c and not short.act.needs_g and a.pool is None and a.w_off is None):
                continue
        elif not _pointwise(a):
            continue
        if not (a.emit is not None and (a.emit.lo, a.emit.hi) == (0, 255) and not a.emit.needs_g):
            continue
        mc = (main if main is not None else a).c       # channels of the unit-stride operand
        gets = {u.args[1]: u for u in na.users if u.op == "call_function" and u.target is operator.getitem}
        if len(gets) != len(na.users) or 1 not in gets:
            continue
        g1 = gets[1]
        nb = next((u for u in g1.users if u.op == "call_module" and u.args == (g1,) and isinstance(modules.get(u.target), Int8Layer) and
                   _pointwise(modules[u.target]) and modules[u.target].emit is not None and not modules[u.target].want_out and
                   not modules[u.target].emit.needs_g and modules[u.target].c == a.k and
                   ((mc, modules[u.target].k) in K.CHAIN_SHAPES if short is None else
                    (mc, short.c, modules[u.target].k) in K.DUAL_CHAIN_SHAPES)), None)
        if nb is None:
            continue
        b = modules[nb.target]
        bgets = {u.args[1]: u for u in nb.users if u.op == "call_function" and u.target is operator.getitem}
        if len(bgets) != len(nb.users) or (0 in bgets and bgets[0].users):
            continue
        name = f"_int8_chain_{count}"
        count += 1
        a.emit_shift = False        # the chain kernel's second GEMM reads those codes in place, as unsigned bytes
        gm.add_module(name, ChainInt8Layer(a, b, want_codes=len(g1.users) > 1, short=short, main=main))
        with graph.inserting_after(na):
            nc = graph.call_module(name, args=na.args)
        with graph.inserting_after(nc):
            outs = [graph.call_function(operator.getitem, (nc, i)) for i in (2, 1, 0)][::-1]
        if 0 in gets:
            gets[0].replace_all_uses_with(outs[0])
        for u in list(g1.users):
            if u is not nb:
                u.replace_input_with(g1, outs[1])
        if 1 in bgets:
            bgets[1].replace_all_uses_with(outs[2])
        for n in list(bgets.values()) + [nb] + list(gets.values()) + [na]:
            graph.erase_node(n)
    report.chained = count
    if count:
        graph.eliminate_dead_code()
        graph.lint()
        gm.recompile()


def _block_end_like(m):
    """A plan layer the library's block-end kernel (csrc/conv_pwr_i8.hip) can take: the only kernel besides the chain kernels that
    knows chunk-major block tensors.  (Whether it DOES take a call is the library's decision per call - K.conv2d_i8 asks and falls
    back to the ordinary layout.)"""
    return (type(m) is Int8Layer and _pointwise(m) and m.c in (256, 512) and m.k % 128 == 0 and m.k_pad == m.k and m.c_pad == m.c and m.relu
            and m.w_off is None and m.pool is None and m.layer.stride[0] == 1 and not m.act.needs_g and
            (m.emit is None or ((m.emit.lo, m.emit.hi) == (0, 255) and not m.emit.needs_g)))


def _block_layout_pass(gm, report):
    """The fp32 block tensor between two kernels that walk it chunk by chunk - written by a chain kernel or the block-end kernel, read
    as the shortcut by another of them, by nothing else - goes CHUNK-MAJOR (K.ChunkMajor, DLMCQ_FP32_*_CHUNK_MAJOR): HBM serves planes in
    which neighbouring workgroups' pieces are neighbours faster than 256-byte pieces of K * 4-byte rows (chain launches -11 ... -18 % at
    14^2 / 56^2, tools/chain_ab.py --abcm).  A private layout of the plan: same values.  Calls that keep their two fp32 tensors in one
    layout (K.CHAIN_ONE_LAYOUT; the block-end kernel) get both or neither."""
    graph = gm.graph
    modules = dict(gm.named_modules())

    def mod(node):
        return modules.get(node.target) if node.op == "call_module" else None

    def writer(m):      # can write its fp32 output chunk-major
        if isinstance(m, ChainInt8Layer):
            return m.a.k % 64 == 0
        if isinstance(m, DualInt8Layer):     # (the 256-deep addend read row by row, the 512-deep one sampled: conv_pwr_applies)
            one = lambda t: type(t) is Int8Layer and t.layer.weight.dim() == 4 and tuple(t.layer.weight.shape[2:]) == (1, 1)
            dense, other = (m.a, m.b) if m.a.c == 256 else (m.b, m.a)
            return (one(m.a) and one(m.b) and (dense.c, other.c) == (256, 512) and dense.layer.stride[0] == 1 and m.a.relu and m.a.k % 128 == 0
                    and m.a.k_pad == m.a.k and m.a.w_off is None and m.b.w_off is None and m.a.pool is None and m.a.emit is not None)
        return isinstance(m, Int8Layer) and _block_end_like(m)

    def reader(node, o):   # reads `o` as its fp32 shortcut, chunk-major if offered
        m = mod(node)
        if len(node.args) != 2 or node.args[1] is not o or node.args[0] is o:
            return False
        if isinstance(m, ChainInt8Layer):
            return m.short is None
        return isinstance(m, Int8Layer) and _block_end_like(m)

    def one_layout(m):
        if isinstance(m, ChainInt8Layer):
            return m.short is None and (m.main.c, m.b.k) in K.CHAIN_ONE_LAYOUT
        return True         # (the block-end kernel)
    nodes = [n for n in graph.nodes if isinstance(mod(n), (ChainInt8Layer, DualInt8Layer, Int8Layer))]
    out_node, src, cm = {}, {}, {}     # node -> getitem of its fp32 output; node -> the node whose output is its shortcut; node -> output chunk-major?
    for nd in nodes:
        for u in nd.users:
            if u.op == "call_function" and u.target is operator.getitem and u.args[1] == 0 and u.users:
                out_node[nd] = u
    for nd, o in out_node.items():
        cm[nd] = writer(mod(nd)) and all(reader(r, o) for r in o.users)
        if cm[nd]:
            for r in o.users:
                src[r] = nd
    changed = True
    while changed:                   # one layout per call where the kernel has a single set of offsets
        changed = False
        for nd in nodes:
            m = mod(nd)
            has_shortcut = len(nd.args) == 2 and not isinstance(m, DualInt8Layer) and not (isinstance(m, ChainInt8Layer) and m.short is not None)
            if nd not in out_node or not has_shortcut or not one_layout(m):
                continue
            icm, ocm = cm.get(src.get(nd), False), cm.get(nd, False)
            if icm and not ocm:
                cm[src[nd]] = False
                changed = True
            elif ocm and not icm:
                cm[nd] = False
                changed = True
    count = 0
    for nd, flag in cm.items():
        mod(nd).out_cm = bool(flag)
        count += bool(flag)
    report.chunk_major = count


class StemLayer(_PlanLayer):
    """The network's first convolution (<= 4 input channels) of the frozen plan: the image is quantised into a
    zero-point-padded NHWC4 code buffer and convolved on the matrix cores (csrc/conv_stem_i8.hip)."""

    def __init__(self, layer, spec, **kw):
        super().__init__(layer, spec, **kw)
        self.c = layer.weight.shape[1]
        if self._w_codes is None:
            wq, wsum = K.quantize_weight_stem(layer.weight, self.w_scale, self.w_lo, self.w_hi)
        else:     # the module's own integer codes (asymmetric / per-channel QBase weights) in the stem kernel's [K, R, 8 taps, 4] layout
            q = _weight_codes(self)                                                # [K, C, R, S] int16, re-centred
            k, c, r, s_ = q.shape
            full = torch.zeros((k, r, 8, 4), dtype=torch.int16, device=q.device)
            full[:, :, :s_, :c] = q.permute(0, 2, 3, 1)
            wq, wsum = full.to(torch.int8).contiguous(), q.sum(dim=(1, 2, 3)).to(torch.int32).contiguous()
        self.register_buffer("wq", wq, persistent=False)
        self.register_buffer("wsum", wsum, persistent=False)

    def forward(self, x):
        lay, act = self.layer, self.act
        numel = x.numel()
        pad, st = lay.padding[0], lay.stride[0]
        k, _, r, s = lay.weight.shape
        emit = self._emit_for(x.shape[0], k, (x.shape[2] + 2 * pad - r) // st + 1, (x.shape[3] + 2 * pad - s) // st + 1)
        in_kernel = self.pool == (3, 2, 1) and k <= 64 and self.w_off is None      # conv + ReLU + MaxPool2d(3, 2, 1) + quantiser: one kernel
        # an unsigned image quantiser's codes go into the padded buffer re-centred (`code - 128`: what the matrix cores multiply),
        # so the first-layer kernels need not xor every operand fragment they read; the zero point moves with them
        shifted = act.lo >= 0
        xpad = K.quantize_pad_nhwc4(x, act.scale, act.zp, act.lo, act.hi, act.form, pad, g=act.g(numel), shift128=shifted)
        res = K.conv2d_i8_stem(xpad, self.wq, self.wsum, lay.bias, self._in_scale(numel), self.zp_shift if shifted else act.zp, self.w_scale, s, stride=st,
                               relu=self.relu, emit=emit, want_out=self.want_out, pool=in_kernel, w_offset=self.w_off, channels=self.c)
        out, out_codes = res if emit is not None else (res, None)
        return (out, out_codes) if in_kernel else self._finish(out, out_codes)


class _DryNode(nn.Module):
    def forward(self, *args):
        raise RuntimeError("fuse_inference(dry_run=True) builds the plan's structure only")


class FusionReport:
    """What the pass did, for logs and tests."""

    def __init__(self):
        self.layers = self.relu = self.residual = self.emit = self.fp32_outputs = self.stem = self.pooled = self.dual = 0
        self.chained = 0      # block end + next block's 1x1 pairs running as one kernel
        self.chunk_major = 0  # fp32 block outputs kept chunk-major between two kernels that walk them chunk by chunk (_block_layout_pass)
        self.dwpw = 0         # depthwise 3x3 + pointwise 1x1 units running as one kernel
        self.skipped = []

    def __repr__(self):
        return (f"FusionReport(int8 layers={self.layers}, relu fused={self.relu}, residual fused={self.residual}, "
                f"code-emitting={self.emit}, fp32 outputs kept={self.fp32_outputs}, stem layers={self.stem}, "
                f"pools on codes={self.pooled}, dual (conv + shortcut conv) kernels={self.dual}, chained pairs={self.chained} (fp32 outputs chunk-major: {self.chunk_major}), "
                f"depthwise + pointwise units={self.dwpw}, "
                f"not eligible={self.skipped})")


class _Tracer(fx.Tracer):
    def is_leaf_module(self, m, qualname):
        return isinstance(m, (QBase, FSPTQBase, RootQBase)) or super().is_leaf_module(m, qualname)


_ADD_FNS = (operator.add, operator.iadd, torch.add)
_RELU_FNS = (F.relu, torch.relu, torch.relu_, F.relu_)


def _is_add(node):
    if node.kwargs:
        return False
    if node.op == "call_function" and node.target in _ADD_FNS:
        return len(node.args) == 2 and all(isinstance(a, fx.Node) for a in node.args)
    if node.op == "call_method" and node.target in ("add", "add_"):
        return len(node.args) == 2 and all(isinstance(a, fx.Node) for a in node.args)
    return False


def _inplace_add(node):
    return (node.op == "call_method" and node.target == "add_") or (node.op == "call_function" and node.target is operator.iadd)


def _is_relu(node, modules):
    if node.op == "call_module":
        return type(modules.get(node.target)) is nn.ReLU
    if node.op == "call_function":
        return node.target in _RELU_FNS
    return node.op == "call_method" and node.target in ("relu", "relu_")


def _pool_params(node, modules):
    """(kernel, stride, padding) if `node` is an nn.MaxPool2d the code-domain pool reproduces."""
    if node.op != "call_module" or type(modules.get(node.target)) is not nn.MaxPool2d:
        return None
    m = modules[node.target]
    one = lambda v: v[0] if isinstance(v, (tuple, list)) and len(set(v)) == 1 else (v if isinstance(v, int) else None)  # noqa: E731
    k, s, p, d = one(m.kernel_size), one(m.stride if m.stride is not None else m.kernel_size), one(m.padding), one(m.dilation)
    if None in (k, s, p, d) or d != 1 or m.ceil_mode or m.return_indices:
        return None
    return k, s, p


class PackedWeights4(nn.Module):
    """The plan's 4-bit weight codes as they are stored: two codes per byte (`dlmcq_pack_int4`: element 2i in the low nibble),
    every layer's codes in the plan's own layout (KRSC / RSC / first-layer), concatenated.  The int8 kernels read one byte per
    code, so each forward starts with ONE `dlmcq_unpack_int4` launch that expands the whole network's weights into a scratch
    buffer the plan nodes' `wq` tensors are views of (BASELINE configs[4]: "sub-byte pack/unpack" on the timed path; MobileOne-S1:
    2.4 MB packed, ~3 us per step).  `expand()` is idempotent: concurrent streams write identical bytes."""

    def __init__(self, nodes, signed):
        super().__init__()
        self.signed = bool(signed)
        sizes = [n.wq.numel() for n in nodes]
        offs, total = [], 0
        for sz in sizes:
            offs.append(total)
            total += (sz + 31) // 32 * 32                     # 16 packed bytes: the kernels want 16-byte-aligned weights
        dev = nodes[0].wq.device
        flat = torch.zeros(total, dtype=torch.int8, device=dev)
        for n, o, sz in zip(nodes, offs, sizes):
            flat[o:o + sz] = n.wq.reshape(-1)
        self.register_buffer("packed", K.pack_int4(flat), persistent=False)
        self.register_buffer("scratch", torch.empty(total, dtype=torch.int8, device=dev), persistent=False)
        self.n = total
        for n, o, sz in zip(nodes, offs, sizes):              # the nodes keep no copy of their own: `wq` becomes a view of the scratch
            n._buffers["wq"] = self.scratch[o:o + sz].view(n.wq.shape)
            n._packed4 = True
        self.expand()

    def expand(self):
        K.unpack_int4(self.packed, self.n, self.signed, out=self.scratch)


def _pack_plan_weights(gm):
    """Store every plan node's weight codes whose range fits 4 bits packed (see PackedWeights4).  Returns the holders."""
    groups = {True: [], False: []}
    for m in gm.modules():
        if isinstance(m, _PlanLayer) and hasattr(m, "wq") and m.wq.dtype == torch.int8 and -8 <= m.w_lo and m.w_hi <= 15:
            lo, hi = int(m.wq.min()), int(m.wq.max())          # (plan build time: one read per layer)
            if 0 <= lo and hi <= 15:
                groups[False].append(m)
            elif -8 <= lo and hi <= 7:
                groups[True].append(m)
    holders = [PackedWeights4(nodes, signed) for signed, nodes in groups.items() if nodes]
    for i, h in enumerate(holders):
        gm.add_module(f"_packed_weights_{i}", h)
    if holders:
        gm.register_forward_pre_hook(lambda mod, args: [h.expand() for h in holders] and None)
    return holders


class EagerFused:
    """The model's forward with its wrappers UNFROZEN - they observe, calibrate and re-quantise exactly as in `model(x)` - but with
    every `layer -> (+ shortcut) -> ReLU` chain whose layer takes its int8 route evaluated by ONE launch (the int8 kernel's
    fused epilogue) instead of the layer, torch's add and torch's ReLU: the same bits (the epilogue is the kernel the plan uses,
    tested against the separate ops), so the scales a calibrating forward derives are identical, at roughly half the HBM traffic.
    This is what makes the first, calibrating batch cheap (bench.py `first_batch`); the frozen plan (`fuse_inference`) is for
    the steady state.  torch.fx reads the dataflow once (wrappers are leaves); execution is an interpreter over that graph.

        fused = EagerFused(model)      # once
        y = fused(x)                   # == model(x), bit for bit, including what the observers see

    Limits (each keeps `y == model(x)` by NOT fusing): a shortcut that is not an fp32 tensor of the layer's exact output shape
    (broadcast adds, other dtypes) runs layer, add and ReLU one by one; an in-place add INTO the shortcut (`short += layer(x)`) is
    left to torch, because the fused launch would not mutate `short`.  Not preserved on fused chains: forward hooks registered on
    the wrapper, the add or the ReLU (`forward_fused` is called directly and the add / ReLU never run as modules)."""

    def __init__(self, model):
        try:
            graph = _Tracer().trace(model)
        except Exception as e:
            raise RuntimeError(f"EagerFused reads the model's dataflow with torch.fx and could not trace it ({type(e).__name__}: {e})") from e
        self.gm = fx.GraphModule(model, graph)
        modules = dict(self.gm.named_modules())
        for node in list(graph.nodes):      # folded BatchNorms / eval-mode Dropout are wires
            if node.op == "call_module" and isinstance(modules[node.target], (nn.Identity, nn.Dropout)) and len(node.args) == 1 and \
                    not (isinstance(modules[node.target], nn.Dropout) and model.training):
                node.replace_all_uses_with(node.args[0])
                graph.erase_node(node)
        self.gm.recompile()
        self.chains = {}      # layer node -> (add node or None, shortcut node or None, relu node or None)
        taken = set()         # an add belongs to the FIRST layer (in program order) that feeds it: the other operand is its shortcut
        for node in graph.nodes:
            if node.op != "call_module" or len(node.args) != 1 or node.kwargs or not hasattr(modules[node.target], "forward_fused"):
                continue
            add = short = relu = None
            users = list(node.users)
            if len(users) == 1 and _is_add(users[0]) and users[0].args[0] is not users[0].args[1]:
                if users[0] in taken:
                    continue
                if _inplace_add(users[0]) and users[0].args[0] is not node:
                    continue      # `short += layer(x)` mutates the shortcut tensor, which other readers may hold: left to torch
                add = users[0]
                taken.add(add)
                short = add.args[1] if add.args[0] is node else add.args[0]
                users = list(add.users)
            if len(users) == 1 and _is_relu(users[0], modules):
                relu = users[0]
            if add is not None or relu is not None:
                self.chains[node] = (add, short, relu)

    def __call__(self, *args):
        # Freshly calibrated layers decide their int8 route by a value on the device (an integer zero point?): the forward runs on the
        # assumption that they may and checks all of them with ONE host read at its end (54 synchronisations saved in ResNet-50's first batch);
        # a wrong guess - a tensor without a zero in front of an unsigned quantiser - re-arms what calibrated and runs again, unspeculated
        from ..quantization.scalar._wrapper import ZeroPointSpeculation
        with ZeroPointSpeculation() as sp:
            interp = _EagerInterp(self.gm, self.chains)
            out = interp.run(*args)
        self.speculation = {"layers": len(sp.pending), "held": True}
        if not sp.verify():
            sp.rearm()
            self.speculation["held"] = False
            interp = _EagerInterp(self.gm, self.chains)
            out = interp.run(*args)
        self.last_states = {n.target: ("fused" if st == "fused" else "plain") for n, st in interp.state.items()}   # (tests, reports)
        return out


class _EagerInterp(fx.Interpreter):
    def __init__(self, gm, chains):
        super().__init__(gm)
        self.chains = chains
        self.add_of = {c[0]: n for n, c in chains.items() if c[0] is not None}
        self.relu_of = {c[2]: n for n, c in chains.items() if c[2] is not None}
        self.state = {}       # layer node -> "fused" | "plain" | ("pending", layer, x)

    def _launch(self, node, layer, x, short):
        add, _, relu = self.chains[node]
        if add is not None and not self._fusable_shortcut(layer, x, short):
            self.state[node] = "plain"       # a broadcast / differently shaped / non-fp32 shortcut: the ops one by one
            return None
        # (observe_out: the launch also leaves the min / max partials of its output for the consumer's observer - the calibrating forward's
        #  extra read of every activation tensor is gone wherever the kernel that ran has the observing epilogue)
        y = layer.forward_fused(x, residual=short, relu=relu is not None, observe_out=True)
        self.state[node] = "plain" if y is None else "fused"
        return y

    @staticmethod
    def _fusable_shortcut(layer, x, short):
        """The fused epilogue adds an fp32 tensor of exactly the layer's output shape; anything else keeps `y == model(x)` by
        running layer, add and ReLU separately."""
        if not isinstance(short, torch.Tensor) or short.dtype != torch.float32 or short.device != x.device:
            return False
        w = layer.weight
        if w.dim() == 2:
            return tuple(short.shape) == (*x.shape[:-1], w.shape[0])
        if x.dim() != 4:
            return False
        def side(n, k, s, p, d):
            return (n + 2 * p - d * (k - 1) - 1) // s + 1
        p = layer.padding if not isinstance(layer.padding, str) else None
        if p is None:
            return False
        return tuple(short.shape) == (x.shape[0], w.shape[0], side(x.shape[2], w.shape[2], layer.stride[0], p[0], layer.dilation[0]),
                                      side(x.shape[3], w.shape[3], layer.stride[1], p[1], layer.dilation[1]))

    def run_node(self, n):
        if n in self.chains:
            add, short, _ = self.chains[n]
            layer = self.fetch_attr(n.target)
            (x,), _ = self.fetch_args_kwargs_from_env(n)
            if add is not None and short not in self.env:      # the shortcut is computed later in program order (a downsample branch): at the add
                self.state[n] = ("pending", layer, x)
                return None
            y = self._launch(n, layer, x, self.env[short] if add is not None else None)
            return layer(x) if y is None else y
        if n in self.add_of:
            c = self.add_of[n]
            st = self.state.get(c)
            if st == "fused":
                return self.env[c]
            if isinstance(st, tuple):
                _, layer, x = st
                short = self.env[self.chains[c][1]]
                y = self._launch(c, layer, x, short)
                if y is not None:
                    return y
                self.env[c] = layer(x)           # not on the int8 route after all: the ops one by one
        elif n in self.relu_of:
            if self.state.get(self.relu_of[n]) == "fused":
                (v,), _ = self.fetch_args_kwargs_from_env(n)
                return v
        return super().run_node(n)


def _codes_from_blob(mod_name, blob, layer):
    """A layer's integer weight codes [K, C, R, S] (int16, on the layer's device) from an integer checkpoint of
    dlmc.utils.export (packed int4 or int8), expanded ON THE DEVICE - the plan never sees fp32 weights for that layer."""
    rec = blob["layers"].get(mod_name)
    if rec is None:
        return None
    dev = layer.weight.device
    n = int(torch.tensor(rec["shape"]).prod())
    if rec["packed_int4"]:
        q = K.unpack_int4(rec["codes"].to(dev), n, rec["lo"] < 0)
    else:
        q = rec["codes"].to(dev)
    return q.reshape(rec["shape"]).to(torch.int16)


def fuse_inference(model, report=None, dry_run=False, chain_pairs=True, pack_int4=True, weight_blob=None, dwpw=False, block_layout=True):
    """Return a `torch.fx.GraphModule` executing `model`'s calibrated quantised forward as the fused int8 plan.
    `pack_int4`: weight codes whose range fits 4 bits are stored packed and expanded by one launch per forward (PackedWeights4).
    `weight_blob`: an integer checkpoint (`dlmc.utils.export.export_quantized_state`) of the same model - the plan takes the
    layers' weight codes from it (expanded on the device) instead of quantising the fp32 weights again.
    Layers that are not eligible (grouped / 3-channel convs, non-integer zero points, RootQ, ...) keep running
    their own wrapper.  `model` must be on the GPU, in eval mode, already calibrated.  `dry_run=True` only takes the
    fusion decisions (graph + `fusion_report`, placeholder nodes): it needs no GPU and the result cannot be run.
    `chain_pairs=False` keeps every block end and the 1x1 convolution behind it as two launches (A/B and tests).
    `dwpw=True` runs every depthwise 3x3 / stride 1 + pointwise 1x1 unit (MobileOne, MobileNet) as ONE launch
    (csrc/conv_dwpw_i8.hip: the code tensor between the two layers stays in LDS; bit-identical).  Off by default: at MobileOne-S1
    W4A8, batch 1024, the unit takes 270 us either way at 28^2 and 14^2 (148 + 123 and 108 + 114 us as two launches) - both halves
    are bound by their vector arithmetic (~24 instructions per depthwise element), which fusing does not remove.
    `block_layout=False` keeps every fp32 block tensor row-major (channels_last) instead of chunk-major between two chain kernels
    (_block_layout_pass; A/B and tests)."""
    if model.training:
        raise RuntimeError("fuse_inference: the plan is for inference - call model.eval() first")
    report = report if report is not None else FusionReport()
    try:
        graph = _Tracer().trace(model)
    except Exception as e:   # data-dependent control flow, *args signatures, ... - torch.fx says which line
        raise RuntimeError(f"fuse_inference reads the model's dataflow with torch.fx and could not trace it ({type(e).__name__}: "
                           f"{e}); the module-by-module path (quantize_model(..., int8_gemm=True)) needs no tracing") from e
    gm = fx.GraphModule(model, graph)
    modules = dict(gm.named_modules())
    specs = {}

    def spec_of(node):
        if node.op != "call_module" or len(node.args) != 1 or node.kwargs:
            return None
        if node.target not in specs:
            mod = modules[node.target]
            specs[node.target] = _frozen_spec(mod) if isinstance(mod, (QBase, FSPTQBase)) else None
            if specs[node.target] is None and isinstance(mod, (QBase, FSPTQBase, RootQBase)):
                report.skipped.append(node.target)
        return specs[node.target]

    # folded BatchNorms (merge_bn leaves nn.Identity) and eval-mode Dropout are wires, not operations
    for node in list(graph.nodes):
        if node.op == "call_module" and isinstance(modules[node.target], (nn.Identity, nn.Dropout)) and len(node.args) == 1:
            node.replace_all_uses_with(node.args[0])
            graph.erase_node(node)

    dual_inputs = {}   # dual plan node -> (activation spec of input 0, of input 1)

    def accepts(u, t):
        """The activation quantiser `u` applies to tensor `t`, if `u` can take `t` as codes instead."""
        if u in dual_inputs:
            acts = {dual_inputs[u][i].key: dual_inputs[u][i] for i in (0, 1) if u.args[i] is t}
            return next(iter(acts.values())) if len(acts) == 1 else None
        s = spec_of(u) if u.args and u.args[0] is t else None
        return s[0] if s is not None and s[4] in ("gemm", "dw") else None

    count = 0
    live = set(graph.nodes)
    for node in list(graph.nodes):
        if node not in live:            # absorbed into a dual kernel earlier in this loop
            continue
        spec = spec_of(node)
        if spec is None:
            continue
        # ---- the chain  layer -> (+ shortcut) -> ReLU, each link the sole user of the previous one ----
        chain, last, residual, relu = [node], node, None, False
        users = list(last.users)
        # (the stem kernel has no shortcut input; a layer whose output channels are zero-padded to a multiple of 64 - MobileNetV2's
        #  24 / 96 / 160-channel projections, CIFAR ResNets' 16 / 32 - computes a k_pad-wide tile: the k-wide fp32 shortcut does not
        #  fit it, so the add stays outside the kernel)
        wn = modules[node.target].weight
        unpadded = wn.dim() != 4 or wn.shape[0] % 64 == 0
        if spec[4] == "gemm" and unpadded and len(users) == 1 and _is_add(users[0]) and users[0].args[0] is not users[0].args[1]:
            add = users[0]
            residual = add.args[1] if add.args[0] is last else add.args[0]
            chain.append(add)
            last = add
            users = list(last.users)
        if len(users) == 1 and _is_relu(users[0], modules):
            relu = True
            chain.append(users[0])
            last = users[0]
            users = list(last.users)
        # ---- a max-pool read only by int8 layers of one quantiser runs on the codes (monotone quantiser) ----
        pool = None
        if len(users) == 1 and _pool_params(users[0], modules) is not None and modules[node.target].weight.shape[0] % 4 == 0:
            mp = users[0]
            cons = [accepts(u, mp) for u in mp.users]
            on_codes = cons and all(c is not None for c in cons) and len({c.key for c in cons}) == 1
            # the first-layer kernel pools in fp32 itself (any consumers); elsewhere the pool runs on the emitted codes
            in_stem = (spec[4] == "stem" and _pool_params(mp, modules) == (3, 2, 1) and modules[node.target].weight.shape[0] <= 64 and
                       spec[5] is None)       # (the pooling first-layer kernel has no weight-offset term)
            if on_codes or in_stem:
                pool = _pool_params(mp, modules)
                chain.append(mp)
                last = mp
        # ---- who reads the result: int8 layers fed ONLY through their activation argument take codes ----
        consumers = {}
        fp32_needed = False
        for u in last.users:
            a = accepts(u, last)
            if a is None:
                fp32_needed = True
            else:
                consumers.setdefault(a.key, []).append((u, a))
        emit, takers = None, []
        if consumers:
            key = max(consumers, key=lambda k: len(consumers[k]))
            takers = [u for u, _ in consumers[key]]
            emit = consumers[key][0][1]
            fp32_needed = fp32_needed or len(consumers) > 1
        if not last.users:
            fp32_needed = True
        name = f"_int8_plan_{count}"
        count += 1
        specs[name] = None
        modules[name] = None
        cls = {"gemm": Int8Layer, "dw": DwInt8Layer}.get(spec[4], StemLayer)
        # the shortcut is itself a not-yet-planned int8 convolution read by nobody else: one dual kernel
        other = spec_of(residual) if residual is not None and residual.op == "call_module" else None
        dual = (other is not None and other[4] == "gemm" and spec[5] is None and other[5] is None and list(residual.users) == [chain[1]] and
                modules[node.target].weight.dim() == 4 and modules[residual.target].weight.dim() == 4)
        if dry_run:       # decisions only (CPU-side tests): the node is a placeholder, nothing is quantised or launched
            gm.add_module(name, _DryNode())
        else:
            def from_blob(name, sp):       # the layer's weight codes come from the integer checkpoint, expanded on the device
                if weight_blob is None or name not in weight_blob["layers"]:
                    return sp
                return sp[:6] + ((lambda name=name: _codes_from_blob(name, weight_blob, modules[name])),)
            spec = from_blob(node.target, spec)
            if other is not None and dual:
                other = from_blob(residual.target, other)
            plan = cls(modules[node.target], spec, relu=relu, emit=emit, want_out=fp32_needed or emit is None, pool=pool)
            # codes of an unsigned-byte quantiser read only by matrix-core layers (no channel padding, no pooling on the way)
            # travel re-centred (see _PlanLayer.__init__); the consumers recognise them by dtype
            def takes_shifted(u):
                sp = spec_of(u) if u.op == "call_module" else None
                m = modules.get(u.target) if u.op == "call_module" else None
                if sp is None or m is None or m.weight.dim() != 4:
                    return False
                if sp[4] == "dw":      # the matrix-core depthwise kernel (csrc/conv_dwm_i8.hip) multiplies signed bytes: no re-centring of its fragments
                    # (not with `dwpw`: the fused unit's kernel emits plain codes only)
                    return (not dwpw and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.weight.shape[0] % 64 == 0)
                return sp[4] == "gemm" and m.groups == 1 and m.weight.shape[1] % 64 == 0
            plan.emit_shift = bool(cls is Int8Layer and emit is not None and 0 <= emit.lo and emit.hi <= 255 and pool is None and
                                   plan.k_pad == plan.k and takers and all(takes_shifted(u) for u in takers))
            gm.add_module(name, DualInt8Layer(plan, Int8Layer(modules[residual.target], other)) if dual else plan)
        if dual:
            args = (node.args[0], residual.args[0])
        else:
            args = (node.args[0],) if residual is None else (node.args[0], residual)
        with graph.inserting_after(last):
            fused = graph.call_module(name, args=args)
        if dual:
            dual_inputs[fused] = (spec[0], other[0])
            chain.insert(0, residual)       # erased last (its only user, the add, goes first)
            specs[residual.target] = None   # never planned on its own
            report.layers += 1
            report.dual += 1
        with graph.inserting_after(fused):
            out = graph.call_function(operator.getitem, (fused, 0))
            codes = graph.call_function(operator.getitem, (fused, 1))
        for u in list(last.users):
            if u in (out, codes):
                continue
            u.replace_input_with(last, codes if u in takers else out)
        for n in reversed(chain[1:] if dual else chain):
            graph.erase_node(n)
        if dual:
            graph.erase_node(residual)
            live.discard(residual)
        report.layers += 1
        report.stem += spec[4] == "stem"
        report.pooled += pool is not None
        report.relu += relu
        report.residual += residual is not None
        report.emit += emit is not None
        report.fp32_outputs += bool(fp32_needed or emit is None)
    graph.eliminate_dead_code()
    graph.lint()
    gm.recompile()
    if chain_pairs and not dry_run:
        _chain_pass(gm, report)
        if block_layout:
            _block_layout_pass(gm, report)
        if dwpw:      # (off by default: measured no faster than the two launches - both halves of a MobileOne unit are bound by their
            #            own vector arithmetic, not by the code tensor between them: LABNOTES round 4)
            _dwpw_pass(gm, report)
    gm.packed_weights = _pack_plan_weights(gm) if (pack_int4 and not dry_run) else []
    gm.fusion_report = report
    return gm


_STREAM_POOL = {}


def _plan_streams(device, n):
    """The side streams of every StreamedPlan on a device come from ONE pool: the runtime maps streams onto a few hardware queues,
    and a second plan's fresh streams can land on queues the first plan's already occupy - its shares then run one after the other
    (RepVGG-A1 behind ResNet-50 in one process: 2.28 ms per step on fresh streams, 1.85 on the shared ones = what it takes alone)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    pool = _STREAM_POOL.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


class StreamedPlan:
    """Run a frozen plan on `n_streams` HIP streams, each taking a contiguous share of the batch.

    A ResNet alternates layers bound by HBM (1x1 expansions that write a shortcut) with layers bound by the operand
    path of the matrix cores (3x3s, 1x1 reductions), and every launch has a ramp and a tail.  Two shares of the batch,
    one layer apart on two streams, fill each other's gaps: ResNet-50 at 512 images, 8.81 -> 8.14 ms (+8 %), three or
    four streams give less.  Outputs are bit-identical to the single-stream plan: nothing in the plan depends on the
    batch size once the scales are frozen - except the QBase family, whose `grad_scale` factor g = 1/sqrt(numel*hi)
    (modules/base.py:96-97) does; such plans are refused.

        fast = StreamedPlan(fuse_inference(model), n_streams=2)
        y = fast(x)
    """

    def __init__(self, plan, n_streams=2):
        if n_streams < 1:
            raise ValueError("n_streams must be >= 1")
        for m in plan.modules():
            if isinstance(m, _PlanLayer) and (m.act.needs_g or (m.emit is not None and m.emit.needs_g)):
                raise ValueError("StreamedPlan: this plan holds QBase quantisers whose scale depends on the number of "
                                 "elements per call (grad_scale); splitting the batch would change the result")
        self.plan, self.n = plan, int(n_streams)
        self.streams = None

    def __call__(self, x):
        if self.n == 1 or x.shape[0] < self.n:
            return self.plan(x)
        if self.streams is None:
            self.streams = _plan_streams(x.device, self.n)
        cur = torch.cuda.current_stream(x.device)
        parts = x.chunk(self.n, dim=0)
        outs = [None] * len(parts)
        for i, part in enumerate(parts):
            s = self.streams[i]
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = self.plan(part)
        for s in self.streams[:len(parts)]:
            cur.wait_stream(s)
        for o, s in zip(outs, self.streams):
            o.record_stream(cur)       # produced on a side stream, consumed (and later freed) on the caller's
        return torch.cat(outs, dim=0)
