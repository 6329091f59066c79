"""ctypes binding of the C ABI in include/dlmcq.h (libdlmcq.so, built by csrc/Makefile).

There is NO fallback: if the shared library is missing or a symbol is absent the import fails,
and every call on a non-GPU tensor raises.  The product path is the HIP path or nothing.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
import sys

_PRODUCT = os.path.join(os.path.dirname(_HERE), "libdlmcq.so")
# Another build of the library (same-box A/B timing, the lab library of tools/) is loaded ONLY when the caller names it AND
# declares itself a lab tool (DLMCQ_LAB_TOOLS=1; tools/*.py and tools/*.sh set both), and the path is printed: a stale
# DLMCQ_LIBRARY in a user's environment does not swap the product library silently - it is reported and ignored.
_OVERRIDE = os.environ.get("DLMCQ_LIBRARY")
LAB_OVERRIDE = bool(_OVERRIDE) and os.environ.get("DLMCQ_LAB_TOOLS") == "1"
if _OVERRIDE and not LAB_OVERRIDE:
    print(f"[dlmc] DLMCQ_LIBRARY={_OVERRIDE} ignored (set DLMCQ_LAB_TOOLS=1 to load a lab / A-B build); loading {_PRODUCT}", file=sys.stderr)
LIB_PATH = _OVERRIDE if LAB_OVERRIDE else _PRODUCT
if LAB_OVERRIDE:
    print(f"[dlmc] lab override: loading {LIB_PATH} instead of the product library", file=sys.stderr)

# enums of include/dlmcq.h
FORM_EMULATE, FORM_QBASE, FORM_ZEROPOINT, FORM_SYMMETRIC, FORM_ROOTQ_ACT = range(5)
EMIT_SHIFT128 = 0x100    # DLMCQ_EMIT_SHIFT128: OR-able into q_form (include/dlmcq.h)
W2_CHUNK_MAJOR = 0x200   # DLMCQ_W2_CHUNK_MAJOR: OR-able into the chain entry points' last quantiser form
FORCE_TILED = 0x400      # DLMCQ_FORCE_TILED: OR-able into q_form of conv2d_i8_nhwc_fused / _asym / _dual / conv2d_dw_i8_nhwc: the family's generic kernel
ROUTE_ONLY = 0x800       # DLMCQ_ROUTE_ONLY: launch nothing, return which kernel the dispatch picks (ROUTE_*)
FP32_IN_CHUNK_MAJOR = 0x2000   # DLMCQ_FP32_IN_CHUNK_MAJOR: a chain call's fp32 shortcut is [K / 64][M][64] (kernels.ChunkMajor)
FP32_OUT_CHUNK_MAJOR = 0x4000  # DLMCQ_FP32_OUT_CHUNK_MAJOR: ... and / or its fp32 block output
PIPELINED = 0x1000       # DLMCQ_PIPELINED (opt-in): the persistent, software-pipelined halo-tile 3x3 kernel where it applies
ROUTE_TILED, ROUTE_HALO3X3, ROUTE_PW, ROUTE_PWR, ROUTE_DW, ROUTE_DWM, ROUTE_HALO3X3_PIPE = 1, 2, 3, 4, 5, 6, 7
ROUTE_TAG = {ROUTE_TILED: "conv_i8", ROUTE_HALO3X3: "conv3x3_halo", ROUTE_PW: "conv_pw", ROUTE_PWR: "conv_pwr", ROUTE_DW: "conv_dw",
             ROUTE_DWM: "conv_dwm", ROUTE_HALO3X3_PIPE: "conv3x3_pipe"}     # the profile tag (bench.py's kernel families) of each route
Y_DEQUANT, Y_CODES = 0, 1
CODES_NONE, CODES_I8, CODES_P4 = 0, 1, 2
MINMAX_ABSMAX, MINMAX_MINMAX, MINMAX_NEGMIN = 0, 1, 2

_p, _i64, _i32, _f32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_size_t

# symbol -> (restype, argtypes); must list every function include/dlmcq.h declares
SIGNATURES = {
    "dlmcq_version": (ctypes.c_int, []),
    "dlmcq_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "dlmcq_fake_quant_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_dequant_codes_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_dequant_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "dlmcq_minmax_scratch_bytes": (_sz, [_i64, _i64, _i64]),
    "dlmcq_minmax_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i32, _p, _sz, _p]),
    "dlmcq_minmax_finalize_f32": (ctypes.c_int, [_p, _i64, _i64, _p, _p, _i32, _p]),
    "dlmcq_conv2d_i8_observed_partials": (_sz, [_i64, _i64]),
    "dlmcq_conv2d_i8_nhwc_fused_observed": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32,
                                                            _i32, _i32, _p, _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p, _i64, _p, _p]),
    "dlmcq_qparams_from_minmax": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_span_scale_f32": (ctypes.c_int, [_p, _p, _p, _i64, _f32, _i32, _p]),
    "dlmcq_lsq_init_scratch_bytes": (_sz, [_i64]),
    "dlmcq_lsq_init_f32": (ctypes.c_int, [_p, _p, _i64, _f32, _p, _sz, _p]),
    "dlmcq_observe_qparams_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    "dlmcq_pack_int4": (ctypes.c_int, [_p, _p, _i64, _p]),
    "dlmcq_unpack_int4": (ctypes.c_int, [_p, _p, _i64, _i32, _p]),
    "dlmcq_fq_bwd_scratch_bytes": (_sz, [_i64, _i64, _i64]),
    "dlmcq_fake_quant_bwd_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _f32, _p, _sz, _p]),
    "dlmcq_rootq_bwd_scratch_bytes": (ctypes.c_size_t, [_i64]),
    "dlmcq_rootq_weight_bwd_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _sz, _p]),
    "dlmcq_fake_quant_bwd_form_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    "dlmcq_rootq_weight_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i32, _i32, _p]),
    "dlmcq_l2norm_scratch_bytes": (_sz, [_i64, _i64, _i64]),
    "dlmcq_l2norm_step_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "dlmcq_l2norm_iterate_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    "dlmcq_l2out_scratch_bytes": (_sz, [_i64, _i64, _i64]),
    "dlmcq_l2out_update_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _f32, _f32, _p, _sz, _p]),
    "dlmcq_l2loss_scratch_bytes": (_sz, [_i64]),
    "dlmcq_l2loss_tensor_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _f32, _p, _sz, _p]),
    "dlmcq_l2loss_rows_f32": (ctypes.c_int, [_p, _p, _p, _i64, _i64, _i32, _p]),
    "dlmcq_adaround_weight_f32": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p]),
    "dlmcq_adaround_weight_bwd_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p]),
    "dlmcq_quantize_weight_krsc_i8": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _i32, _p]),
    "dlmcq_conv2d_i8_nhwc_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                               _i32, _i32, _i32, _i32, _p]),
    "dlmcq_conv2d_i8_nhwc_fused": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                                 _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_conv2d_i8_nhwc_asym": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                                _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_conv2d_dw_i8_nhwc": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32,
                                              _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_dwpw_pack_table": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "dlmcq_conv2d_dwpw_i8_nhwc": (ctypes.c_int, [_p, _p, _i32, _i32, _i32, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _i32, _i32, _i32, _f32,
                                                _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_conv2d_i8_nhwc_dual": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                                _i32, _i32, _i32, _i32,
                                                _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32,
                                                _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_conv2d_i8_nhwc_chain": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p, _i32, _p, _p, _p, _i32,
                                                 _i32, _i32, _f32, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _i32, _i32, _f32,
                                                 _i32, _p]),
    "dlmcq_conv2d_i8_nhwc_dual_chain": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i32,
                                                      _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32,
                                                      _i32, _p, _p, _p, _i32, _i32, _i32, _f32,
                                                      _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _i32, _p]),
    "dlmcq_quantize_pad_nhwc4": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32,
                                               _i32, _f32, _p]),
    "dlmcq_quantize_weight_stem_i8": (ctypes.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _i32, _p]),
    "dlmcq_conv2d_i8_stem_fused": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32,
                                                 _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_conv2d_i8_stem_asym": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32,
                                                _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_conv2d_i8_stem_pool_fused": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32,
                                                      _i32, _p, _p, _p, _i32, _i32, _i32, _f32, _p]),
    "dlmcq_maxpool_codes_nhwc": (ctypes.c_int, [_p, _p, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _p]),
    "dlmcq_fold_bn_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _f32, _p]),
    "dlmcq_repvgg_fuse_f32": (ctypes.c_int, [_p, _p, _p, _p, _p, _p, _p, _f32, _f32, _f32, _i64, _i64, _p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C dlmc-quant_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
    # torch ships its own libamdhip64 (soname libamdhip64.so.7); load it first so that libdlmcq's
    # DT_NEEDED resolves to the SAME runtime instance torch uses (pointers and streams are shared).
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tlib):
        ctypes.CDLL(tlib, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        except AttributeError:
            if LAB_OVERRIDE:   # an older build named for an A/B run: calls to what it lacks fail there
                continue
            raise
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def experimental(name, argtypes):
    """A symbol exported by the library but deliberately absent from include/dlmcq.h (tests / tuning only)."""
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = ctypes.c_int, argtypes
    return fn


class DlmcqError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise DlmcqError(f"{lib.dlmcq_strerror(rc).decode()} (code {rc})")


def route(rc):
    """Return value of an entry point called with ROUTE_ONLY: the route (> 0), or an error like any other call."""
    if rc <= 0:
        raise DlmcqError(f"{lib.dlmcq_strerror(rc).decode()} (code {rc})" if rc < 0 else "route query returned DLMCQ_OK (an empty problem)")
    return rc


def stream_ptr():
    """The hipStream_t torch is currently launching on (kernels are enqueued there, asynchronously)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DlmcqError(
                "dlmc (MI355X build) runs on the GPU only: got a tensor on "
                f"'{t.device}'.  There is no CPU fallback in this package.")
