"""FSPTQBase: the few-shot PTQ ("RepAPQ") wrapper - per-channel W8A8 (reference:
dlmc/quantization/scalar/FSPTQuant/base.py).

    activations  q = clamp(R(x/s) + zp, lo, hi);  x' = (q - zp)*s      reference: base.py:108-109
    weights      q = clamp(R(W/s_k), lo, hi);     W' = q*s_k            reference: base.py:149-152
                 s_k = per-output-channel minmax scale + 1e-6           reference: base.py:113-129
    adaround     q = floor(W/s_k) + h(alpha)   (train)  /  + [alpha >= 0]   (eval)   base.py:136-141

Each of the first two lines is one HIP launch; scale observers are one read each.  State_dict keys
and shapes follow the reference (`wt_scale`/`wt_offset` are [K,1,1,1] or [K,1], `org_weight`, `alpha`).
Differences from the reference, on paths where it crashes (SURVEY.md defects 4, 5): buffers are created
on the layer's own device instead of a hard-coded `device('cuda')`; with `act_quant` off the layer uses
the raw input (the reference leaves `q_input` unbound).
"""
import torch
from torch.nn import Module

from .... import _native as N
from .. import kernels as K
from .. import ops
from .._wrapper import InitState, ZeroPointSpeculation, fake_quant, fusable_epilogue, int8_forward, int8_gemm_default, int8_kind, set_scale
from ..utils import get_qrange


class _AdaRoundFn(torch.autograd.Function):
    """AdaRound weight path: one HIP launch forward, one backward (alpha and the per-channel scale learn;
    floor() passes no gradient to the weight, exactly as in the reference's op chain)."""

    @staticmethod
    def forward(ctx, weight, alpha, scale, lo, hi, training):
        ctx.save_for_backward(weight, alpha, scale)
        ctx.rng = (lo, hi, training)
        return K.adaround_weight(weight, alpha, scale, lo, hi, training)

    @staticmethod
    def backward(ctx, gy):
        weight, alpha, scale = ctx.saved_tensors
        lo, hi, training = ctx.rng
        need_a, need_s = ctx.needs_input_grad[1] and training, ctx.needs_input_grad[2]
        ga, gs = (None, None)
        if need_a or need_s:
            if training:
                ga, gs = K.adaround_weight_backward(weight, alpha, scale, gy, lo, hi, want_alpha=need_a, want_scale=need_s)
            else:   # eval form: [alpha >= 0] has no gradient; the scale still multiplies the clamped code
                y = K.adaround_weight(weight, alpha, scale, lo, hi, False)
                red = tuple(range(1, y.dim()))
                gs = (gy * (y / scale)).sum(dim=red, keepdim=True)
        return None, ga, gs, None, None, None


class FSPTQBase(Module):
    qconfig: dict

    def __init__(self, qconfig: dict = None):
        self.initialize(qconfig)

    def initialize(self, qconfig):
        self.qconfig = qconfig
        self.train_module = 0
        self.wt_min_val, self.wt_max_val = get_qrange(qconfig["weight"]["args"]["signed"],
                                                      qconfig["weight"]["args"]["n_bits"])
        self.in_min_val, self.in_max_val = get_qrange(qconfig["input"]["args"]["signed"],
                                                      qconfig["input"]["args"]["n_bits"])
        dev = self.weight.device
        k = self.weight.shape[0]
        per_k = (k, 1, 1, 1) if self.weight.dim() == 4 else (k, 1)
        self.register_parameter("in_scale", torch.nn.Parameter(torch.ones(1, device=dev)))
        self.register_buffer("in_offset", torch.zeros(1, device=dev))
        self.register_buffer("in_init_state", torch.zeros(1, device=dev))
        self.register_parameter("wt_scale", torch.nn.Parameter(torch.ones(per_k, device=dev)))
        self.register_buffer("wt_offset", torch.ones(per_k, device=dev))
        self.register_buffer("wt_init_state", torch.zeros(1, device=dev))
        self.register_buffer("org_weight", self.weight.clone().detach())
        self.act_quant = self.qconfig["input"]["enable"]
        self.wt_quant = self.qconfig["weight"]["enable"]
        self.soft_target = True
        if self.qconfig["weight"].get("recon_type") in ("adaround", "dist_recon"):
            self.register_parameter("alpha", torch.nn.Parameter(torch.ones_like(self.weight)))
            self.gamma, self.zeta = -0.1, 1.1
            self.beta = 2 / 3
        self._init = InitState()
        # fused int8 conv/linear on the matrix cores: opt-in (qconfig["int8_gemm"] or DLMC_INT8_GEMM=1); the
        # default is the reference-identical fp32 conv of the fake-quantised operands
        self.int8_gemm = bool(qconfig.get("int8_gemm", int8_gemm_default()))
        self._zp_is_int = None

    def _int8_applicable(self, input):
        if not (self.int8_gemm and self.act_quant and self.wt_quant) or torch.is_grad_enabled():
            return False
        if self.qconfig["weight"].get("recon_type") in ("adaround", "dist_recon") or self.in_scale.numel() != 1:
            return False
        if not (int8_kind(self) is not None and self.wt_min_val >= -128 and self.wt_max_val <= 127):
            return False
        if not (0 <= self.in_min_val and self.in_max_val <= 255) and not (-128 <= self.in_min_val and self.in_max_val <= 127):
            return False
        if self._zp_is_int is None:   # one host read, right after calibration - or none: the caller collects them (ZeroPointSpeculation)
            if ZeroPointSpeculation.active is not None:
                ZeroPointSpeculation.active.pending.append(self)
                return True
            zp = float(self.in_offset.reshape(-1)[0])
            self._zp_is_int = zp == round(zp) and self.in_min_val <= zp <= self.in_max_val
        return self._zp_is_int

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "in_offset"   # becomes 0-dim after calibration (base.py:99)
        if key in state_dict and state_dict[key].shape != self.in_offset.shape:
            self.in_offset = torch.zeros_like(state_dict[key], dtype=torch.float32, device=self.weight.device)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._init.invalidate()
        self._zp_is_int = None

    def _forward_func(self, input, weight):
        raise NotImplementedError

    # --------------------------------------------------------------------- adaround pieces
    def init_alpha(self):
        """alpha such that h(alpha) reproduces the fractional part of W/s (base.py:69-76)."""
        w = self.weight.detach()
        ratio = w / self.wt_scale.detach()
        rest = ratio - torch.floor(ratio)
        self.alpha.data.copy_(-torch.log((self.zeta - self.gamma) / (rest - self.gamma) - 1))

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def change_quant_state(self, wt_state, act_state):
        self.wt_quant = wt_state
        self.act_quant = act_state

    def reinit_parameters(self):
        self._init.mark(self, "in_init_state", False)
        self._init.mark(self, "wt_init_state", False)

    # ---------------------------------------------------------------------------- forward
    def _calibrate_input(self, input):
        cfg = self.qconfig["input"]
        kw = dict(cfg["args"])
        if str(cfg["type"]).startswith("minmax_") and "pixel" not in str(cfg["type"]):
            kw["sync"] = True
        if "channel" in str(cfg["type"]):
            kw.setdefault("ch_axis", 1)
        if cfg["type"] == "minmax_tensor":       # (round 5: the producing launch may have observed this very tensor in its epilogue)
            hint = K.minmax_hint(input)
            if hint is not None:
                kw["minmax_hint"] = hint
        scale, offset = ops.get_qparams_tensor(input.detach(), qtype=cfg["type"], **kw)
        set_scale(self.in_scale, scale)
        self.in_offset = offset.detach().to(torch.float32)
        self._zp_is_int = None
        self._init.mark(self, "in_init_state")
        if ZeroPointSpeculation.active is not None:
            ZeroPointSpeculation.active.calibrated.append(self)

    def _calibrate_weight(self):
        cfg = self.qconfig["weight"]
        scale, offset = ops.get_qparams_tensor(self.weight.detach(), qtype=cfg["type"], **cfg["args"])
        set_scale(self.wt_scale, scale + 1e-6)
        self.wt_offset = offset.detach().to(torch.float32)
        if cfg.get("recon_type") == "adaround":
            self.init_alpha()
        self._init.mark(self, "wt_init_state")

    def forward_fused(self, input, residual=None, relu=False, observe_out=False):
        """`forward(input)` followed by `+ residual` and ReLU as ONE int8 launch - observers and calibration exactly as in
        `forward` - when this layer takes its int8 route; None when it does not (the caller then runs the ops one by one;
        whatever calibration was due has been done).  Used by dlmc.utils.fuse.EagerFused."""
        N.require_gpu(input, self.weight)
        if self.act_quant and not self._init.ready(self, "in_init_state"):
            self._calibrate_input(input)
        if self.wt_quant and not self._init.ready(self, "wt_init_state"):
            self._calibrate_weight()
        if not (fusable_epilogue(self) and self._int8_applicable(input)):
            return None
        return int8_forward(self, input, self.in_scale, self.in_offset, self.in_min_val, self.in_max_val, N.FORM_ZEROPOINT,
                            self.wt_scale, self.wt_min_val, self.wt_max_val, residual=residual, relu=relu, observe_out=observe_out)

    def forward(self, input):
        N.require_gpu(input, self.weight)
        q_input = input
        if self.act_quant and not self._init.ready(self, "in_init_state"):
            self._calibrate_input(input)
        if self.wt_quant and not self._init.ready(self, "wt_init_state"):
            self._calibrate_weight()
        if self._int8_applicable(input):
            return int8_forward(self, input, self.in_scale, self.in_offset, self.in_min_val, self.in_max_val,
                                N.FORM_ZEROPOINT, self.wt_scale, self.wt_min_val, self.wt_max_val)
        if self.act_quant:
            q_input = fake_quant(input, self.in_scale, self.in_offset, self.in_min_val, self.in_max_val,
                                 N.FORM_ZEROPOINT)
        if not self.wt_quant:
            return self._forward_func(q_input, self.weight)
        recon = self.qconfig["weight"].get("recon_type")
        if recon == "adaround":
            # block reconstruction: alpha (and the scales) learn - fused soft-rounding kernel, forward and backward
            if torch.is_grad_enabled() and (self.alpha.requires_grad or self.wt_scale.requires_grad):
                weight = _AdaRoundFn.apply(self.weight, self.alpha, self.wt_scale, self.wt_min_val, self.wt_max_val,
                                           bool(self.training))
            else:
                weight = K.adaround_weight(self.weight, self.alpha, self.wt_scale, self.wt_min_val, self.wt_max_val,
                                           self.training)
        elif recon == "dist_recon":
            raise NotImplementedError("recon_type 'dist_recon' is unfinished in the reference "
                                      "(FSPTQuant/base.py:133,143 call undefined code)")
        else:
            weight = fake_quant(self.weight, self.wt_scale, None, self.wt_min_val, self.wt_max_val,
                                N.FORM_SYMMETRIC)
        return self._forward_func(q_input, weight)
