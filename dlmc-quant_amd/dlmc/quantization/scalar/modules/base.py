"""QBase: the generic QAT / PTQ wrapper (reference: dlmc/quantization/scalar/modules/base.py).

Same surface as the reference - `initialize(qconfig)`, `forward(input)`, `_forward_func(input, weight)`,
`reset_qparams()`, state_dict keys `in_scale, in_offset, in_init_state, wt_scale, wt_offset,
wt_init_state` - but every tensor-sized step is one HIP launch:

    first call   observer (one read, device-side scale/offset, all-reduce(MAX) across ranks for
                 activations) ...                                     reference: base.py:82-94,107-129
    every call   y = R(clamp((x - o)/s^, lo, hi)) * s^ + o            reference: base.py:96-102,131-133
                 with a HIP backward for QAT (LSQ-style scale gradient)

and the steady state never synchronises with the host (init flags are mirrored host-side).
Deliberate differences from the reference, all on paths where it crashes (SURVEY.md defects 3, 4, 5, 6):
per-channel qtypes work (the scale Parameter grows to [1,..,C,..,1]); nothing hard-codes
`device('cuda')`; `weight.enable: false` uses the fp32 weight; `reset_qparams()` re-arms the observer.
"""
import math
from fnmatch import fnmatch

import torch
from torch.nn import Module

from .... import _native as N
from .. import kernels as K
from .. import ops
from .._wrapper import (InitState, fake_quant, fusable_epilogue, int8_forward, int8_gemm_default, int8_kind, set_scale,
                        ste_scale_value)
from ..utils import get_qrange


class QBase(Module):
    qconfig: dict

    def __init__(self, qconfig: dict = None):
        # Reached only through QConv2d / QLinear, whose __init__ has already run the nn layer's.
        self.initialize(qconfig)

    # -------------------------------------------------------------------------- set-up
    def initialize(self, qconfig):
        if "channel" in str(qconfig["input"]["type"]):
            qconfig["input"]["args"]["ch_axis"] = 1   # activations are NCHW / (N, C): channel axis 1
        self.qconfig = qconfig
        self.wt_min_val, self.wt_max_val = get_qrange(qconfig["weight"]["args"]["signed"],
                                                      qconfig["weight"]["args"]["n_bits"])
        self.in_min_val, self.in_max_val = get_qrange(qconfig["input"]["args"]["signed"],
                                                      qconfig["input"]["args"]["n_bits"])
        dev = self.weight.device
        self.register_parameter("in_scale", torch.nn.Parameter(torch.ones(1, device=dev)))
        self.register_buffer("in_offset", None)
        self.register_buffer("in_init_state", torch.zeros(1, device=dev))
        self.register_parameter("wt_scale", torch.nn.Parameter(torch.ones(1, device=dev)))
        self.register_buffer("wt_offset", None)
        self.register_buffer("wt_init_state", torch.zeros(1, device=dev))
        self._init = InitState()
        # fused int8 conv/linear on the matrix cores: opt-in; needs symmetric (zero-offset) per-tensor quantisers
        self.int8_gemm = bool(qconfig.get("int8_gemm", int8_gemm_default()))
        self._int8_offsets_zero = None

    def _int8_applicable(self):
        cfg = self.qconfig
        if not (self.int8_gemm and cfg["input"]["enable"] and cfg["weight"]["enable"]) or torch.is_grad_enabled():
            return False
        if self.in_scale.numel() != 1 or self.wt_scale.numel() != 1 or int8_kind(self) is None:
            return False
        if not (-128 <= self.in_min_val and self.in_max_val <= 127 and -128 <= self.wt_min_val and self.wt_max_val <= 127):
            return False
        if self._int8_offsets_zero is None:   # one host read, right after calibration
            self._int8_offsets_zero = bool(float(self.in_offset.abs().max()) == 0 and float(self.wt_offset.abs().max()) == 0)
        return self._int8_offsets_zero

    def reset_qparams(self):
        """Forget the calibrated scales: the next forward observes again."""
        dev = self.weight.device
        with torch.no_grad():
            self.in_scale.data = torch.ones(1, device=dev)
            self.wt_scale.data = torch.ones(1, device=dev)
        self.in_offset = None
        self.wt_offset = None
        self._init.mark(self, "in_init_state", False)
        self._init.mark(self, "wt_init_state", False)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # offsets are registered as None until the first forward; accept them from a checkpoint
        for name in ("in_offset", "wt_offset"):
            key = prefix + name
            if key in state_dict and getattr(self, name) is None:
                setattr(self, name, torch.zeros_like(state_dict[key], dtype=torch.float32, device=self.weight.device))
        for name in ("in_scale", "wt_scale"):
            key = prefix + name
            if key in state_dict and state_dict[key].shape != getattr(self, name).shape:
                getattr(self, name).data = torch.ones_like(state_dict[key], dtype=torch.float32, device=self.weight.device)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._init.invalidate()
        self._int8_offsets_zero = None

    def _forward_func(self, input, weight):
        raise NotImplementedError

    # ------------------------------------------------------------------------- calibration
    def _calibrate_input(self, input):
        cfg = self.qconfig["input"]
        x = input.detach()
        if fnmatch(str(cfg["type"]), "LSQ"):
            # LSQ init 2*mean|x|/sqrt(Qp) (base.py:84-85): one read of x on the device (dlmcq_lsq_init_f32)
            scale = K.lsq_init(x, self.in_max_val)
            offset = torch.zeros((), device=x.device)
        else:
            kw = dict(cfg["args"])
            if str(cfg["type"]).startswith("minmax_") and "pixel" not in str(cfg["type"]):
                kw["sync"] = True   # data parallel: the observer must see the global batch
            scale, offset = ops.get_qparams_tensor(x, qtype=cfg["type"], **kw)
        set_scale(self.in_scale, scale)
        self.in_offset = offset.detach().to(torch.float32)
        self._int8_offsets_zero = None
        self._init.mark(self, "in_init_state")

    def _calibrate_weight(self, input):
        cfg = self.qconfig["weight"]
        w = self.weight.detach()
        if fnmatch(str(cfg["type"]), "*output*"):
            scale, offset = ops.get_qparams_output(input.detach(), w, self, qtype=cfg["type"], **cfg["args"])
        elif fnmatch(str(cfg["type"]), "LSQ"):
            scale = K.lsq_init(w, self.wt_max_val)         # base.py:118-121
            offset = torch.zeros((), device=w.device)
        else:
            scale, offset = ops.get_qparams_tensor(w, qtype=cfg["type"], **cfg["args"])
        set_scale(self.wt_scale, scale)
        self.wt_offset = offset.detach().to(torch.float32)
        self._init.mark(self, "wt_init_state")

    # ----------------------------------------------------------------------------- forward
    def forward_fused(self, input, residual=None, relu=False, observe_out=False):      # (observe_out: FSPTQ family only so far)
        """`forward(input)` followed by `+ residual` and ReLU as ONE int8 launch (observers and calibration as in `forward`)
        when this layer takes its int8 route; None when it does not.  Used by dlmc.utils.fuse.EagerFused."""
        N.require_gpu(input, self.weight)
        if not (self.int8_gemm and not torch.is_grad_enabled() and fusable_epilogue(self)):
            return None
        if self.qconfig["input"]["enable"] and not self._init.ready(self, "in_init_state"):
            self._calibrate_input(input)
        if self.qconfig["weight"]["enable"] and not self._init.ready(self, "wt_init_state") and \
                not fnmatch(str(self.qconfig["weight"]["type"]), "*output*"):
            self._calibrate_weight(input)
        if not (self._init.ready(self, "wt_init_state") and self._int8_applicable()):
            return None
        g_i = 1 / math.sqrt(input.numel() * self.in_max_val)
        g_w = 1 / math.sqrt(self.weight.numel() * self.wt_max_val)
        return int8_forward(self, input, self.in_scale, None, self.in_min_val, self.in_max_val, N.FORM_QBASE,
                            ste_scale_value(self.wt_scale, g_w), self.wt_min_val, self.wt_max_val, g_in=g_i,
                            wt_scale_key=(self.wt_scale.data_ptr(), self.wt_scale._version, g_w), residual=residual, relu=relu)

    def forward(self, input):
        N.require_gpu(input, self.weight)
        if self.int8_gemm and not torch.is_grad_enabled():
            if self.qconfig["input"]["enable"] and not self._init.ready(self, "in_init_state"):
                self._calibrate_input(input)
            if self.qconfig["weight"]["enable"] and not self._init.ready(self, "wt_init_state") and \
                    not fnmatch(str(self.qconfig["weight"]["type"]), "*output*"):
                self._calibrate_weight(input)
            if self._init.ready(self, "wt_init_state") and self._int8_applicable():
                g_i = 1 / math.sqrt(input.numel() * self.in_max_val)
                g_w = 1 / math.sqrt(self.weight.numel() * self.wt_max_val)
                return int8_forward(self, input, self.in_scale, None, self.in_min_val, self.in_max_val, N.FORM_QBASE,
                                    ste_scale_value(self.wt_scale, g_w), self.wt_min_val, self.wt_max_val, g_in=g_i,
                                    wt_scale_key=(self.wt_scale.data_ptr(), self.wt_scale._version, g_w))
        if self.qconfig["input"]["enable"]:
            if not self._init.ready(self, "in_init_state"):
                self._calibrate_input(input)
            g_i = 1 / math.sqrt(input.numel() * self.in_max_val)
            input = fake_quant(input, self.in_scale, self.in_offset, self.in_min_val, self.in_max_val,
                               N.FORM_QBASE, g_i)
        weight = self.weight
        if self.qconfig["weight"]["enable"]:
            if not self._init.ready(self, "wt_init_state"):
                self._calibrate_weight(input)
            g_w = 1 / math.sqrt(self.weight.numel() * self.wt_max_val)
            weight = fake_quant(self.weight, self.wt_scale, self.wt_offset, self.wt_min_val, self.wt_max_val,
                                N.FORM_QBASE, g_w)
        return self._forward_func(input, weight)
