"""GPU parity of the HIP kernels, through the C ABI: bit-exact against the golden vectors (made by
the reference) and against the oracle on seeded inputs; size-independent properties at full size."""
import math

import numpy as np
import pytest
import torch

from _cmp import assert_bits_equal, ulp_distance
from oracle import fakequant_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "the gpu suite needs a GPU"
    from dlmc.quantization.scalar import kernels
    return kernels


def N():
    from dlmc import _native
    return _native


def gen(seed):
    g = torch.Generator()
    g.manual_seed(2333 + seed)
    return g


def values_equal(got, want, what=""):
    """Equal as numbers (+0 == -0), NaN == NaN: for observer outputs, where torch itself does not
    define which zero a max/min returns."""
    g = got.detach().cpu().reshape(-1).double()
    w = want.detach().cpu().reshape(-1).double()
    assert g.shape == w.shape, f"{what}: {g.shape} vs {w.shape}"
    ok = (g == w) | (g.isnan() & w.isnan())
    assert bool(ok.all()), f"{what}: got {g[~ok][:6]} want {w[~ok][:6]}"


# --------------------------------------------------------------------------- golden: primitives
def test_golden_primitives_form_a(K, golden):
    n = N()
    for c in golden.of_kind("primitive"):
        x, s, o = (golden.get(c, k).to(DEV) for k in ("x", "scale", "offset"))
        y = K.fake_quant(x, s, o, c["lo"], c["hi"], n.FORM_EMULATE)
        q = K.fake_quant(x, s, o, c["lo"], c["hi"], n.FORM_EMULATE, y_kind=n.Y_CODES)
        assert_bits_equal(y, golden.get(c, "y"), c["name"] + ".y")
        assert_bits_equal(q, golden.get(c, "q"), c["name"] + ".q")
        # integer codes == the fp32 codes (NaN -> 0 by definition of the int8 emission)
        _, codes = K.fake_quant(x, s, o, c["lo"], c["hi"], n.FORM_EMULATE, codes="i8")
        want = torch.nan_to_num(golden.get(c, "q"), nan=0.0).to(torch.int32)
        assert torch.equal(codes.cpu().to(torch.int32), want), c["name"] + ".codes"
        # reference `dequantize` on fp32 codes
        yd = K.dequant(q, s, o)
        assert_bits_equal(yd, golden.get(c, "y"), c["name"] + ".dequant")


def test_golden_observers(K, golden):
    for c in golden.of_kind("observer"):
        x = golden.get(c, "x").to(DEV)
        s, o = K.observe_qparams(x, c["n_bits"], c["signed"])
        values_equal(s, golden.get(c, "t_scale"), c["name"] + ".t_scale")
        values_equal(o, golden.get(c, "t_offset"), c["name"] + ".t_offset")
        assert s.dim() == 0
        if golden.has(c, "c_scale"):
            s, o = K.observe_qparams(x, c["n_bits"], c["signed"], ch_axis=c["ch_axis"])
            want = golden.get(c, "c_scale")
            assert list(s.shape) == list(want.shape)
            values_equal(s, want, c["name"] + ".c_scale")
            values_equal(o, golden.get(c, "c_offset"), c["name"] + ".c_offset")
        if golden.has(c, "t_scale_nooff"):
            s, o = K.observe_qparams(x, c["n_bits"], c["signed"], allow_offset=False)
            values_equal(s, golden.get(c, "t_scale_nooff"))
            values_equal(o, golden.get(c, "t_offset_nooff"))
        if golden.has(c, "c_scale_nooff"):
            s, o = K.observe_qparams(x, c["n_bits"], c["signed"], ch_axis=c["ch_axis"], allow_offset=False)
            values_equal(s, golden.get(c, "c_scale_nooff"))
            values_equal(o, golden.get(c, "c_offset_nooff"))


def test_golden_layer_operands(K, golden):
    """The fake-quantised input / weight the reference's wrappers hand to conv / linear."""
    n = N()
    for c in golden.of_kind("qbase"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        (ilo, ihi), (wlo, whi) = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x").to(DEV), golden.get(c, "weight").to(DEV)
        s_in, o_in = K.observe_qparams(x, ia["n_bits"], ia["signed"])
        s_wt, o_wt = K.observe_qparams(w, wa["n_bits"], wa["signed"])
        values_equal(s_in, golden.get(c, "in_scale"))
        values_equal(s_wt, golden.get(c, "wt_scale"))
        xq = K.fake_quant(x, s_in, o_in, ilo, ihi, n.FORM_QBASE, g=1 / math.sqrt(x.numel() * ihi))
        wq = K.fake_quant(w, s_wt, o_wt, wlo, whi, n.FORM_QBASE, g=1 / math.sqrt(w.numel() * whi))
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        x2 = golden.get(c, "x2").to(DEV)
        xq2 = K.fake_quant(x2, s_in, o_in, ilo, ihi, n.FORM_QBASE, g=1 / math.sqrt(x2.numel() * ihi))
        assert_bits_equal(xq2, golden.get(c, "fq_input2"), c["name"] + ".fq_input2")
    for c in golden.of_kind("fsptq"):
        if c["qconfig"]["weight"]["recon_type"] == "adaround":
            continue
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        (ilo, ihi), (wlo, whi) = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x").to(DEV), golden.get(c, "weight").to(DEV)
        s_in, zp = K.observe_qparams(x, ia["n_bits"], ia["signed"])
        s_wt, _ = K.observe_qparams(w, wa["n_bits"], wa["signed"], ch_axis=0, scale_eps=1e-6)
        values_equal(s_in, golden.get(c, "in_scale"))
        values_equal(zp, golden.get(c, "in_offset"))
        values_equal(s_wt, golden.get(c, "wt_scale"), c["name"] + ".wt_scale")
        xq = K.fake_quant(x, s_in, zp, ilo, ihi, n.FORM_ZEROPOINT)
        wq = K.fake_quant(w, s_wt, None, wlo, whi, n.FORM_SYMMETRIC)
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
    for c in golden.of_kind("rootq"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        (ilo, ihi), (wlo, whi) = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x").to(DEV), golden.get(c, "weight").to(DEV)
        xq = K.fake_quant(x, golden.get(c, "st_in_run_scale").to(DEV), None, ilo, ihi, n.FORM_ROOTQ_ACT)
        assert_bits_equal(xq, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        wq = K.rootq_weight(w, golden.get(c, "st_wt_run_upper").to(DEV), golden.get(c, "st_wt_run_lower").to(DEV), wlo, whi)
        assert_bits_equal(wq, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
        x2 = golden.get(c, "x2").to(DEV)
        xq = K.fake_quant(x2, golden.get(c, "tr_in_run_scale").to(DEV), None, ilo, ihi, n.FORM_ROOTQ_ACT)
        assert_bits_equal(xq, golden.get(c, "fq_input_train"), c["name"] + ".fq_input_train")
        wq = K.rootq_weight(w, golden.get(c, "tr_wt_run_upper").to(DEV), golden.get(c, "tr_wt_run_lower").to(DEV), wlo, whi)
        assert_bits_equal(wq, golden.get(c, "fq_weight_train"), c["name"] + ".fq_weight_train")


def test_golden_backward(K, golden):
    for c in golden.of_kind("qbase_grad"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        (ilo, ihi), (wlo, whi) = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x").to(DEV), golden.get(c, "weight").to(DEV)
        gx, gs = K.fake_quant_backward(x, golden.get(c, "g_fq_input").to(DEV), golden.get(c, "in_scale").to(DEV),
                                       golden.get(c, "in_offset").to(DEV), ilo, ihi, 1 / math.sqrt(x.numel() * ihi))
        assert_bits_equal(gx, golden.get(c, "grad_x"), c["name"] + ".grad_x")
        torch.testing.assert_close(gs.cpu(), golden.get(c, "grad_in_scale"), rtol=2e-4, atol=1e-6)
        gw, gs = K.fake_quant_backward(w, golden.get(c, "g_fq_weight").to(DEV), golden.get(c, "wt_scale").to(DEV),
                                       golden.get(c, "wt_offset").to(DEV), wlo, whi, 1 / math.sqrt(w.numel() * whi))
        assert_bits_equal(gw, golden.get(c, "grad_weight"), c["name"] + ".grad_weight")
        torch.testing.assert_close(gs.cpu(), golden.get(c, "grad_wt_scale"), rtol=2e-4, atol=1e-6)


def test_golden_backward_fsptq(K, golden):
    """The one-pass HIP backward of the ZEROPOINT (activations) and SYMMETRIC (per-channel weights) forms against gradients
    the REFERENCE produced under autograd (tests/golden/golden_v1_grad.*), fed the upstream gradients captured in that run."""
    from dlmc import _native as n
    for c in golden.of_kind("fsptq_grad"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        (ilo, ihi), (wlo, whi) = O.qrange(ia["signed"], ia["n_bits"]), O.qrange(wa["signed"], wa["n_bits"])
        x, w = golden.get(c, "x").to(DEV), golden.get(c, "weight").to(DEV)
        gx, gs = K.fake_quant_backward(x, golden.get(c, "g_fq_input").to(DEV), golden.get(c, "in_scale").to(DEV),
                                       golden.get(c, "in_offset").to(DEV), ilo, ihi, 0.0, form=n.FORM_ZEROPOINT)
        assert_bits_equal(gx, golden.get(c, "grad_x"), c["name"] + ".grad_x")
        torch.testing.assert_close(gs.cpu().reshape(-1), golden.get(c, "grad_in_scale").reshape(-1), rtol=2e-4, atol=1e-6)
        s_wt = golden.get(c, "wt_scale").to(DEV)
        gw, gs = K.fake_quant_backward(w, golden.get(c, "g_fq_weight").to(DEV), s_wt.reshape(-1), torch.zeros_like(s_wt).reshape(-1),
                                       wlo, whi, 0.0, ch_axis=0, form=n.FORM_SYMMETRIC)
        assert_bits_equal(gw, golden.get(c, "grad_weight"), c["name"] + ".grad_weight")
        torch.testing.assert_close(gs.cpu().reshape(-1), golden.get(c, "grad_wt_scale").reshape(-1), rtol=2e-4, atol=1e-5)


# ------------------------------------------------------------------- seeded inputs vs the oracle
LAYOUTS = [  # (shape, ch_axis)   ch_axis None = per tensor
    ((4099,), None),                 # numel % 4 != 0
    ((3, 16, 28, 28), None),
    ((64, 32, 3, 3), 0),             # KCRS, inner = 288
    ((256, 64, 1, 1), 0),            # 1x1 conv weights, inner = 64
    ((10, 2048), 0),                 # fc weights
    ((3, 16, 28, 28), 1),            # NCHW, inner % 4 == 0
    ((2, 24, 7, 7), 1),              # inner = 49: float4 straddles channels
    ((5, 3, 7, 7), 1),               # slab = 147: not a multiple of 4 -> generic kernel
    ((33, 130), 1),                  # (N, C): inner = 1
    ((2, 6000, 2, 2), 1),            # many channels per chunk
]
RANGES = [(True, 8), (False, 8), (True, 4), (False, 4), (False, 2)]


def _oracle_form(form, x, s, o, lo, hi, g):
    n = N()
    if form == n.FORM_EMULATE:
        return O.fq_emulate(x, s, o, lo, hi)
    if form == n.FORM_QBASE:
        return O.fq_qbase(x, s, o, lo, hi, g)
    if form == n.FORM_ZEROPOINT:
        return O.fq_zeropoint(x, s, o, lo, hi)
    if form == n.FORM_SYMMETRIC:
        return O.fq_symmetric(x, s, lo, hi)
    return O.fq_rootq_act(x, s, lo, hi)


@pytest.mark.parametrize("form", range(5))
def test_forms_vs_oracle(K, form):
    n = N()
    k = 0
    for shape, ch_axis in LAYOUTS:
        for signed, bits in RANGES:
            k += 1
            lo, hi = O.qrange(signed, bits)
            g = gen(k)
            x = torch.randn(shape, generator=g)
            if not signed:
                x = torch.relu(x) + (0.02 if k % 2 else 0.0)
            flat = x.view(-1)
            flat[::11] = torch.round(flat[::11] * 16) / 16      # plenty of exact ties for dyadic scales
            flat[3] = float("nan")
            flat[7] = float("inf")
            flat[9] = -0.0
            if ch_axis is None:
                s = torch.tensor(2.0 ** -4 if k % 3 == 0 else 0.0371, dtype=torch.float32)
                o = torch.tensor(0.0 if signed else (3.0 if form == n.FORM_ZEROPOINT else 0.0137))
            else:
                cs = [1] * len(shape)
                cs[ch_axis] = shape[ch_axis]
                s = (torch.rand(cs, generator=g) * 0.05 + 1e-3)
                s.view(-1)[0] = 2.0 ** -5
                o = torch.zeros(cs) if signed else (
                    torch.randint(0, 5, cs, generator=g).float() if form == n.FORM_ZEROPOINT
                    else torch.randn(cs, generator=g) * 0.05)
            gg = 1 / math.sqrt(x.numel() * hi)
            q_ref, y_ref = _oracle_form(form, x, s, o, lo, hi, gg)
            xd, sd, od = x.to(DEV), s.to(DEV), o.to(DEV)
            tag = f"form{form} {shape} ax{ch_axis} {'s' if signed else 'u'}{bits}"
            y = K.fake_quant(xd, sd, od, lo, hi, form, g=gg)
            assert_bits_equal(y, y_ref, tag + " y")
            q = K.fake_quant(xd, sd, od, lo, hi, form, g=gg, y_kind=n.Y_CODES)
            assert_bits_equal(q, q_ref, tag + " q")
            # int8 codes, and their dequantisation == y (where the code is an integer in range)
            if form != n.FORM_ROOTQ_ACT or lo == 0:
                _, codes = K.fake_quant(xd, sd, od, lo, hi, form, g=gg, codes="i8")
                fin = torch.isfinite(q_ref)
                assert torch.equal(codes.cpu()[fin].to(torch.int32), q_ref[fin].to(torch.int32)), tag + " i8"
                yd = K.dequant_codes(codes, x.shape, sd, od, form, "i8", lo < 0, g=gg)
                assert_bits_equal(yd.cpu()[fin], y_ref[fin], tag + " dequant(i8)")
                if hi <= 15 and lo >= -8:
                    _, p4 = K.fake_quant(xd, sd, od, lo, hi, form, g=gg, codes="p4", want_y=False)
                    un = K.unpack_int4(p4, x.numel(), lo < 0).cpu().to(torch.int32).reshape(x.shape)
                    assert torch.equal(un[fin], q_ref[fin].to(torch.int32)), tag + " p4"
                    yd = K.dequant_codes(p4, x.shape, sd, od, form, "p4", lo < 0, g=gg)
                    assert_bits_equal(yd.cpu()[fin], y_ref[fin], tag + " dequant(p4)")


def test_unaligned_inplace_and_empty(K):
    n = N()
    g = gen(77)
    base = torch.randn(4097 + 3, generator=g)
    s, o = torch.tensor(0.03), torch.tensor(0.01)
    for off in (1, 2, 3):                       # 4-, 8-, 12-byte misaligned views -> generic kernel
        x = base[off:off + 4097]
        xd = base.to(DEV)[off:off + 4097]
        assert xd.data_ptr() % 16 != 0
        y = K.fake_quant(xd, s.to(DEV), o.to(DEV), -127, 127, n.FORM_QBASE, g=1e-4)
        assert_bits_equal(y, O.fq_qbase(x, s, o, -127, 127, 1e-4)[1], f"offset {off}")
        mx, mn = K.minmax(xd)
        assert float(mx) == float(x.max()) and float(mn) == float(x.min())
    # in place
    x = torch.randn(3, 8, 14, 14, generator=g)
    xd = x.to(DEV)
    K.fake_quant(xd, s.to(DEV), o.to(DEV), 0, 255, n.FORM_EMULATE, out=xd)
    assert_bits_equal(xd, O.fq_emulate(x, s, o, 0, 255)[1], "in place")
    # empty tensors are a no-op, not an error
    e = torch.empty(0, 4, 3, 3, device=DEV)
    assert K.fake_quant(e, s.to(DEV), None, -127, 127, n.FORM_SYMMETRIC).shape == e.shape
    # an empty reduction has no value: refused, as torch refuses it
    with pytest.raises(n.DlmcqError):
        K.minmax(torch.empty(0, device=DEV))


def test_pack_unpack_roundtrip(K):
    g = gen(5)
    for nel in (1, 2, 15, 16, 17, 4096, 100003):
        for signed in (True, False):
            codes = torch.randint(-7 if signed else 0, 8 if signed else 16, (nel,), generator=g, dtype=torch.int8)
            packed = K.pack_int4(codes.to(DEV))
            assert packed.numel() == (nel + 1) // 2
            c = codes.to(torch.int32) & 0xF
            want = c[0::2].clone()
            want[: c[1::2].numel()] |= c[1::2] << 4
            assert torch.equal(packed.cpu().to(torch.int32), want), f"pack n={nel}"
            back = K.unpack_int4(packed, nel, signed)
            assert torch.equal(back.cpu().to(torch.int32), codes.to(torch.int32)), f"unpack n={nel}"


@pytest.mark.parametrize("ch_axis", [None, 0, 1])
def test_observer_vs_oracle_shapes(K, ch_axis):
    k = 0
    shapes = [(7,), (4100,), (1, 3, 224, 224), (8, 64, 56, 56), (4, 256, 14, 14), (6, 512, 7, 7), (64, 2048),
              (256, 64, 1, 1), (64, 3, 7, 7), (1000, 2048), (2, 4, 60, 60), (3, 5, 33, 9)]
    for shape in shapes:
        if ch_axis is not None and len(shape) <= ch_axis:
            continue
        for signed in (True, False):
            k += 1
            x = torch.randn(shape, generator=gen(100 + k))
            if k % 3 == 0:
                x = torch.relu(x)
            xd = x.to(DEV)
            if ch_axis is None:
                s, o = K.observe_qparams(xd, 8, signed)
                ws, wo = O.minmax_tensor(x, 8, signed)
            else:
                s, o = K.observe_qparams(xd, 8, signed, ch_axis=ch_axis)
                ws, wo = O.minmax_channel(x, 8, signed, ch_axis=ch_axis)
                assert list(s.shape) == list(ws.shape)
            values_equal(s, ws, f"{shape} ax{ch_axis} scale")
            values_equal(o, wo, f"{shape} ax{ch_axis} offset")
            # raw min/max and the packed [max | -min] form used by the cross-rank all-reduce
            mx, mn = K.minmax(xd, ch_axis=ch_axis, mode=N().MINMAX_NEGMIN)
            red = tuple(i for i in range(x.dim()) if i != ch_axis) if ch_axis is not None else None
            amax = x.max() if red is None else (x.amax(dim=red) if red else x)
            amin = x.min() if red is None else (x.amin(dim=red) if red else x)
            values_equal(mx, amax)
            values_equal(mn, -amin)
            s2, o2 = K.qparams_from_minmax(mx.reshape(-1), mn.reshape(-1), 8, signed, min_is_negated=True) if not signed \
                else K.qparams_from_minmax(K.minmax(xd, ch_axis=ch_axis, mode=N().MINMAX_ABSMAX)[0].reshape(-1), None, 8, True)
            values_equal(s2, ws, "split observer scale")
            values_equal(o2, wo, "split observer offset")


def test_observer_nan_inf(K):
    for bad in (float("nan"), float("inf"), float("-inf")):
        x = torch.randn(4, 6, 5, 5, generator=gen(9))
        x[1, 2, 3, 4] = bad
        for signed in (True, False):
            s, o = K.observe_qparams(x.to(DEV), 8, signed)
            ws, wo = O.minmax_tensor(x, 8, signed)
            values_equal(s, ws, f"{bad} tensor")
            s, o = K.observe_qparams(x.to(DEV), 8, signed, ch_axis=1)
            ws, wo = O.minmax_channel(x, 8, signed, ch_axis=1)
            values_equal(s, ws, f"{bad} channel scale")
            values_equal(o, wo, f"{bad} channel offset")


def test_backward_vs_oracle(K):
    k = 0
    for shape, ch_axis in [((4099,), None), ((8, 16, 14, 14), None), ((16, 8, 3, 3), 0), ((4, 8, 6, 6), 1), ((3, 5, 7, 7), 1)]:
        for lo, hi in ((-127, 127), (0, 15)):
            k += 1
            g = gen(300 + k)
            x = torch.randn(shape, generator=g) * (1.0 if lo < 0 else 0.5) + (0.0 if lo < 0 else 0.4)
            gy = torch.randn(shape, generator=g)
            if ch_axis is None:
                s, o = torch.tensor([0.012 if lo < 0 else 0.07]), torch.tensor([0.0 if lo < 0 else 0.01])
            else:
                cs = [1] * len(shape)
                cs[ch_axis] = shape[ch_axis]
                s = torch.rand(cs, generator=g) * 0.02 + 0.005
                o = torch.zeros(cs) if lo < 0 else torch.rand(cs, generator=g) * 0.02
            gg = 1 / math.sqrt(x.numel() * hi)
            s_hat = O.ste_scale(s, gg)
            v = (x - o) / s_hat
            inside = (v >= lo) & (v <= hi)
            gv = torch.where(inside, gy * s_hat, torch.zeros(()))
            want_gx = gv / s_hat
            contrib = (gy * O.ste_round(v.clamp(lo, hi)) + (-gv) * (v / s_hat)).double()
            red = None if ch_axis is None else tuple(i for i in range(x.dim()) if i != ch_axis)
            want_gs = (contrib.sum() if red is None else contrib.sum(dim=red)).float().reshape(-1) * gg
            gx, gs = K.fake_quant_backward(x.to(DEV), gy.to(DEV), s.to(DEV), o.to(DEV), lo, hi, gg, ch_axis=ch_axis)
            assert_bits_equal(gx, want_gx, f"{shape} ax{ch_axis} gx")
            torch.testing.assert_close(gs.cpu(), want_gs, rtol=1e-4, atol=1e-7)
            gx2, gs2 = K.fake_quant_backward(x.to(DEV), gy.to(DEV), s.to(DEV), o.to(DEV), lo, hi, gg, ch_axis=ch_axis)
            assert torch.equal(gs, gs2), "scale gradient must be reproducible run to run"


def test_rootq_weight_vs_oracle(K):
    for k, (lo, hi) in enumerate(((0, 15), (0, 3), (-7, 7), (0, 255))):
        g = gen(400 + k)
        w = torch.randn(64, 32, 3, 3, generator=g) * 0.05
        w.view(-1)[5] = float("nan")
        w.view(-1)[6] = 1e8   # additive clipping of a huge value is not clamp
        up, lw = torch.tensor(0.08), torch.tensor(-0.075)
        _, _, want = O.fq_rootq_weight(w, up, lw, torch.tensor(0.25), lo, hi)
        got = K.rootq_weight(w.to(DEV), up.to(DEV), lw.to(DEV), lo, hi)
        assert_bits_equal(got, want, f"rootq weight {lo}..{hi}")


# ------------------------------------------------------------- full-size, size-independent checks
def test_full_size_properties(K):
    """BASELINE config 2 sizes: A = 64x256x56x56, W = 256x256x3x3.  The oracle cannot run all of A in
    seconds, so: (1) sampled slabs against the oracle, (2) idempotence, (3) code-histogram checksum,
    (4) observer against torch's exact amax/amin on the same device data."""
    n = N()
    torch.manual_seed(2333)
    A = torch.randn(64, 256, 56, 56, device=DEV)
    W = torch.randn(256, 256, 3, 3, device=DEV) * math.sqrt(2 / 2304)
    # observers
    s_t, o_t = K.observe_qparams(A, 8, True)
    assert float(s_t) == float(A.abs().max().cpu() / 127)  # divide on the CPU: torch-GPU multiplies by 1/127
    s_c, o_c = K.observe_qparams(A, 8, False, ch_axis=1)
    mx, mn = A.amax(dim=(0, 2, 3)), A.amin(dim=(0, 2, 3))
    values_equal(s_c.reshape(-1), (mx - mn).cpu() / 255, "per-channel scale")
    values_equal(o_c.reshape(-1), mn, "per-channel offset")
    s_w, _ = K.observe_qparams(W, 8, True, ch_axis=0)
    values_equal(s_w.reshape(-1), W.abs().amax(dim=(1, 2, 3)).cpu() / 127)
    # per-tensor QBase form on A
    gg = 1 / math.sqrt(A.numel() * 127)
    y = K.fake_quant(A, s_t, o_t, -127, 127, n.FORM_QBASE, g=gg)
    for nidx in (0, 17, 63):
        ref = O.fq_qbase(A[nidx].cpu(), s_t.cpu(), o_t.cpu(), -127, 127, gg)[1]
        assert_bits_equal(y[nidx], ref, f"A[{nidx}] per tensor")
    y2 = K.fake_quant(y, s_t, o_t, -127, 127, n.FORM_QBASE, g=gg)
    assert torch.equal(y, y2), "fake-quant must be idempotent at fixed scale"
    _, codes = K.fake_quant(A, s_t, o_t, -127, 127, n.FORM_QBASE, g=gg, codes="i8")
    hist = torch.bincount(codes.reshape(-1).to(torch.int64) + 128, minlength=256)
    assert int(hist.sum()) == A.numel() and int(hist[0]) == 0          # -128 is never produced
    assert int(hist[1]) >= 1 or int(hist[255]) >= 1                    # the absmax element saturates at +-127
    # per-channel zero-point form on A (FSPTQ) and symmetric weights
    yc = K.fake_quant(A, s_c, o_c, 0, 255, n.FORM_EMULATE)
    for nidx in (1, 40):
        ref = O.fq_emulate(A[nidx].cpu(), s_c.cpu()[0], o_c.cpu()[0], 0, 255)[1]
        assert_bits_equal(yc[nidx], ref, f"A[{nidx}] per channel")
    wq = K.fake_quant(W, s_w + 1e-6, None, -127, 127, n.FORM_SYMMETRIC)
    assert_bits_equal(wq, O.fq_symmetric(W.cpu(), s_w.cpu() + 1e-6, -127, 127)[1], "W per channel")
    assert ulp_distance(wq, O.fq_symmetric(W.cpu(), s_w.cpu() + 1e-6, -127, 127)[1]) == 0


def test_beyond_2_to_31_elements(K):
    """Maximum sizes: a per-tensor pass over more than 2^31 elements (8.6 GB in, 8.6 GB out) - the index math is
    64-bit; checked on the first and last megabyte against the oracle, and the observer must see an extreme
    planted in the very last element."""
    n = (1 << 31) + 4096 + 3
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * n * 4:
        pytest.skip("not enough free HBM for the 2^31-element case")
    nn_ = N()
    x = torch.empty(n, device=DEV)
    x.normal_(generator=torch.Generator(device=DEV).manual_seed(2333))
    x[-1] = 77.0
    x[n // 2] = -55.0
    s, o = K.observe_qparams(x, 8, False)
    assert float(o) == -55.0 and float(s) == float((torch.tensor(77.0) - torch.tensor(-55.0)) / 255)
    sa, _ = K.observe_qparams(x, 8, True)
    assert float(sa) == float(torch.tensor(77.0) / 127)
    y = K.fake_quant(x, s, o, 0, 255, nn_.FORM_EMULATE)
    _, codes = K.fake_quant(x, s, o, 0, 255, nn_.FORM_EMULATE, codes="i8", want_y=False)
    m = 1 << 18
    for sl in (slice(0, m), slice(n - m, n), slice((1 << 31) - m // 2, (1 << 31) + m // 2)):
        q_ref, y_ref = O.fq_emulate(x[sl].cpu(), s.cpu(), o.cpu(), 0, 255)
        assert_bits_equal(y[sl], y_ref, f"slice {sl}")
        assert torch.equal(codes[sl].cpu().to(torch.float32), q_ref)
    del y, codes, x
    torch.cuda.empty_cache()


def test_l2norm_step_vs_oracle(K):
    """One fused refinement iteration == quantize + two reductions of the reference (ops.py:78-79, :206-207)."""
    for k, (shape, ch_axis) in enumerate([((70001,), None), ((8, 16, 28, 28), None), ((64, 32, 3, 3), 0), ((6, 24, 14, 14), 1),
                                          ((33, 130), 1)]):
        for signed, bits in ((True, 8), (False, 4)):
            lo, hi = O.qrange(signed, bits)
            g = gen(700 + k)
            x = torch.randn(shape, generator=g)
            if not signed:
                x = torch.relu(x) + 0.01
            if ch_axis is None:
                s, o = O.minmax_tensor(x, bits, signed)
                q = O.quantize_codes(x, s, o, lo, hi)
                want = (x * q).sum().double() / (q * q + 1e-7).sum().double()
            else:
                s, o = O.minmax_channel(x, bits, signed, ch_axis=ch_axis)
                q = O.quantize_codes(x, s, o, lo, hi)
                red = tuple(i for i in range(x.dim()) if i != ch_axis)
                want = ((x * q).double().sum(dim=red) / (q * q + 1e-7).double().sum(dim=red)).reshape(s.shape)
            got = K.l2norm_step(x.to(DEV), s.to(DEV), o.to(DEV), lo, hi)
            assert got.shape == s.shape
            torch.testing.assert_close(got.cpu().double(), want, rtol=2e-5, atol=0)


def test_entry_points_are_graph_capturable(K):
    """include/dlmcq.h promises no allocation / synchronisation inside the library: the observer, the fake-quant
    and the int8 conv can be captured into one HIP graph (through torch.cuda.CUDAGraph) and replayed."""
    n = N()
    g = gen(900)
    x = torch.rand(8, 64, 14, 14, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(DEV)
    s_w, _ = K.observe_qparams(w, 8, True, ch_axis=0, scale_eps=1e-6)
    static_x = x.clone(memory_format=torch.preserve_format)
    outs = {}
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):            # warm-up on the side stream, as graph capture requires
        for _ in range(2):
            s, o = K.observe_qparams(static_x, 8, False)
            _, codes = K.fake_quant(static_x, s, o, 0, 255, n.FORM_ZEROPOINT, codes="i8", want_y=False)
            wq, wsum = K.quantize_weight_krsc(w, s_w, -127, 127)
            K.conv2d_i8(codes, wq, wsum, None, s, o, s_w, padding=1)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        s, o = K.observe_qparams(static_x, 8, False)
        y = K.fake_quant(static_x, s, o, 0, 255, n.FORM_ZEROPOINT)
        _, codes = K.fake_quant(static_x, s, o, 0, 255, n.FORM_ZEROPOINT, codes="i8", want_y=False)
        wq, wsum = K.quantize_weight_krsc(w, s_w, -127, 127)
        outs["conv"] = K.conv2d_i8(codes, wq, wsum, None, s, o, s_w, padding=1)
        outs["y"], outs["s"] = y, s
    for scale in (1.0, 0.5):                 # replay on new data written into the captured input buffer
        static_x.copy_(x * scale)
        graph.replay()
        torch.cuda.synchronize()
        xs = (x * scale)
        s_ref, o_ref = O.minmax_tensor(xs.cpu().contiguous(), 8, False)
        assert float(outs["s"]) == float(s_ref)
        y_ref = O.fq_zeropoint(xs.cpu().contiguous(), s_ref, o_ref, 0, 255)[1]
        assert_bits_equal(outs["y"].contiguous(), y_ref, f"graph replay scale {scale}")
        eager = torch.nn.functional.conv2d(outs["y"], O.fq_symmetric(w.cpu(), s_w.cpu(), -127, 127)[1].to(DEV), padding=1)
        torch.testing.assert_close(outs["conv"], eager, rtol=1e-4, atol=1e-4)


def test_fuzz_shapes_forms_alignments(K):
    """Seeded fuzz: 150 random (shape, channel axis, form, range, storage offset) combinations, every output
    (y, fp32 codes, int8 codes, observer) against the oracle bit for bit - ragged sizes, slabs that are not a
    multiple of 4, single-element channels, misaligned views."""
    n = N()
    rng = np.random.default_rng(2333)
    for case in range(150):
        rank = int(rng.integers(1, 5))
        shape = tuple(int(rng.integers(1, 9)) for _ in range(rank - 1)) + (int(rng.integers(1, 70)),)
        if rank >= 3 and rng.random() < 0.3:
            shape = shape[:-2] + (7, 7)
        per_channel = rank >= 2 and rng.random() < 0.6
        ch_axis = int(rng.integers(0, min(rank, 2))) if per_channel else None
        signed, bits = RANGES[int(rng.integers(0, len(RANGES)))]
        lo, hi = O.qrange(signed, bits)
        form = int(rng.integers(0, 5))
        g = gen(1000 + case)
        numel = int(np.prod(shape))
        off = int(rng.integers(0, 4)) if rng.random() < 0.3 else 0
        base = torch.randn(numel + off, generator=g)
        if not signed:
            base = torch.relu(base) + 0.01
        x = base[off:].reshape(shape)
        if ch_axis is None:
            s = torch.tensor(float(rng.uniform(0.005, 0.05)), dtype=torch.float32)
            o = torch.tensor(0.0 if signed else (2.0 if form == n.FORM_ZEROPOINT else 0.01), dtype=torch.float32)
        else:
            cs = [1] * rank
            cs[ch_axis] = shape[ch_axis]
            s = torch.rand(cs, generator=g) * 0.05 + 1e-3
            o = torch.zeros(cs) if signed else (torch.randint(0, 4, cs, generator=g).float() if form == n.FORM_ZEROPOINT
                                                else torch.rand(cs, generator=g) * 0.02)
        gg = 1 / math.sqrt(numel * hi)
        q_ref, y_ref = _oracle_form(form, x, s, o, lo, hi, gg)
        xd = base.to(DEV)[off:].reshape(shape)
        tag = f"fuzz {case}: {shape} ax{ch_axis} form{form} {'s' if signed else 'u'}{bits} off{off}"
        assert_bits_equal(K.fake_quant(xd, s.to(DEV), o.to(DEV), lo, hi, form, g=gg), y_ref, tag + " y")
        assert_bits_equal(K.fake_quant(xd, s.to(DEV), o.to(DEV), lo, hi, form, g=gg, y_kind=n.Y_CODES), q_ref, tag + " q")
        if form != n.FORM_ROOTQ_ACT or lo == 0:
            _, codes = K.fake_quant(xd, s.to(DEV), o.to(DEV), lo, hi, form, g=gg, codes="i8")
            assert torch.equal(codes.cpu().to(torch.float32), q_ref), tag + " i8"
        so, oo = K.observe_qparams(xd, bits, signed, ch_axis=ch_axis)
        ws, wo = (O.minmax_tensor(x, bits, signed) if ch_axis is None else O.minmax_channel(x, bits, signed, ch_axis=ch_axis))
        values_equal(so, ws, tag + " scale")
        values_equal(oo, wo, tag + " offset")


def test_fsptq_forms_backward_matches_autograd_of_the_reference_chain():
    """One-pass HIP backward of the FSPTQ forms (ZEROPOINT activations per tensor, SYMMETRIC weights per channel) against
    autograd through the reference's op chain on the same device: gx bit for bit, the scale gradient to sum tolerance."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import _wrapper as Wr
    from dlmc.quantization.scalar import kernels as K
    g = torch.Generator().manual_seed(2333)
    cases = [("zeropoint", N.FORM_ZEROPOINT, (4, 16, 9, 9), None, 0, 255, 3.0),
             ("zeropoint s8", N.FORM_ZEROPOINT, (3, 8, 5, 7), None, -127, 127, 0.0),
             ("symmetric ch0", N.FORM_SYMMETRIC, (32, 16, 3, 3), 0, -127, 127, 0.0),
             ("symmetric 4 bit", N.FORM_SYMMETRIC, (24, 40), 0, -7, 7, 0.0)]
    for name, form, shape, ch_axis, lo, hi, zp in cases:
        x = (torch.randn(shape, generator=g) * 2).to(DEV)
        gy = torch.randn(shape, generator=g).to(DEV)
        if ch_axis is None:
            scale = torch.tensor([float(x.abs().max()) / 100], device=DEV)
            bshape = (1,)
        else:
            scale = (x.abs().amax(dim=tuple(range(1, x.dim()))) / (hi * 0.8)).reshape(-1)
            bshape = (shape[0],) + (1,) * (len(shape) - 1)
        offset = torch.tensor([zp], device=DEV) if ch_axis is None else None
        xr = x.clone().requires_grad_(True)
        sr = scale.clone().reshape(bshape).requires_grad_(True)
        off = offset if offset is not None else torch.zeros((), device=DEV)
        y = Wr._composite(form, xr, sr, off, lo, hi, 0.0)
        y.backward(gy)
        gx, gs = K.fake_quant_backward(x, gy, scale, offset, lo, hi, 0.0, ch_axis=ch_axis, form=form)
        assert_bits_equal(gx, xr.grad, name + " gx")
        torch.testing.assert_close(gs.reshape(-1), sr.grad.reshape(-1), rtol=2e-4, atol=1e-4, msg=lambda m: f"{name} gscale: {m}")
        # and through the autograd Function the wrappers use
        xr2 = x.clone().requires_grad_(True)
        sr2 = scale.clone().reshape(bshape).requires_grad_(True)
        Wr.fake_quant(xr2, sr2, offset, lo, hi, form).backward(gy)
        assert_bits_equal(xr2.grad, xr.grad, name + " Function gx")
        torch.testing.assert_close(sr2.grad, sr.grad, rtol=2e-4, atol=1e-4)


def test_rootq_activation_backward_matches_autograd_of_the_reference_chain():
    """Fused backward of the RootQ activation form against autograd through RootQ/base.py:106-111's ops."""
    from dlmc import _native as N
    from dlmc.quantization.scalar import kernels as K
    from dlmc.quantization.scalar.RootQ import base as RB
    g = torch.Generator().manual_seed(77)
    for shape, lo, hi, smul in (((4, 16, 9, 9), 0, 15, 0.15), ((3, 7, 5), 0, 3, 0.5), ((2, 8, 4, 4), 0, 255, 0.01)):
        x = (torch.randn(shape, generator=g) * 1.5).to(DEV)
        x.view(-1)[:4] = torch.tensor([0.0, -0.0, 1e-9, -1e-9], device=DEV)
        gy = torch.randn(shape, generator=g).to(DEV)
        s = torch.tensor(smul, device=DEV)
        xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
        RB._act_composite(xr, sr, lo, hi).backward(gy)
        gx, gs = K.fake_quant_backward(x, gy, s.reshape(1), None, lo, hi, 0.0, form=N.FORM_ROOTQ_ACT)
        assert_bits_equal(gx, xr.grad, f"rootq act gx {shape}")
        torch.testing.assert_close(gs.reshape(()), sr.grad, rtol=2e-4, atol=1e-3)
        xr2, sr2 = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
        RB._RootQActFn.apply(xr2, sr2, lo, hi).backward(gy)
        assert_bits_equal(xr2.grad, xr.grad, "Function gx")
        torch.testing.assert_close(sr2.grad, sr.grad, rtol=2e-4, atol=1e-3)


def test_rootq_weight_backward_matches_autograd_of_the_reference_chain():
    """Fused backward of the RootQ weight transform (gw, g_upper, g_lower, g_alpha) against autograd through the
    reference's op chain on the device (pow / log: fp32 tolerance)."""
    import math
    from dlmc.quantization.scalar import kernels as K
    from dlmc.quantization.scalar.RootQ import base as RB
    g = torch.Generator().manual_seed(4242)
    for shape, bits, alpha in (((64, 32, 3, 3), 4, 0.25), ((40, 24), 2, 0.7), ((16, 8, 3, 3), 3, 1.5), ((8, 16), 4, 5e-5)):
        lo, hi = -(2 ** (bits - 1) - 1), 2 ** (bits - 1) - 1
        w = (torch.randn(shape, generator=g) * 0.1).to(DEV)
        gy = torch.randn(shape, generator=g).to(DEV)
        bound = 2 * float(w.abs().mean()) * math.sqrt(hi)
        leaves = [w.clone().requires_grad_(True)] + [torch.tensor(v, device=DEV, requires_grad=True) for v in (bound, -bound, alpha)]
        RB._weight_composite(*leaves, lo, hi).backward(gy)
        gw, gu, gl, ga = K.rootq_weight_backward(w, gy, leaves[1], leaves[2], leaves[3], lo, hi)
        tag = f"rootq weight bwd {shape} b{bits} alpha {alpha}"
        torch.testing.assert_close(gw, leaves[0].grad, rtol=2e-4, atol=2e-5, msg=lambda m: f"{tag} gw: {m}")
        scale = float(gy.abs().sum())
        for name, got, want in (("g_upper", gu, leaves[1].grad), ("g_lower", gl, leaves[2].grad), ("g_alpha", ga, leaves[3].grad)):
            assert abs(float(got) - float(want)) <= 2e-4 * abs(float(want)) + 1e-5 * scale, f"{tag} {name}: {float(got)} vs {float(want)}"
        # through the autograd Function of the wrapper
        l2 = [w.clone().requires_grad_(True)] + [torch.tensor(v, device=DEV, requires_grad=True) for v in (bound, -bound, alpha)]
        RB._RootQWeightFn.apply(*l2, lo, hi).backward(gy)
        torch.testing.assert_close(l2[0].grad, leaves[0].grad, rtol=2e-4, atol=2e-5)
        for a, b in zip(l2[1:], leaves[1:]):
            assert abs(float(a.grad) - float(b.grad)) <= 2e-4 * abs(float(b.grad)) + 1e-5 * scale


def test_output_aware_scale_small_batch_few_channels_large_map():
    """quantize_l2norm_output (ops.py:85-109) on an output of [2, 16, 224, 224]: per tensor, the fused update reduces ONE flat
    row whose segment count exceeds the per-channel plan's (round 2 sized the scratch for the per-channel plan only and the
    entry point refused the call with DLMCQ_ESCRATCH).  Against the oracle's loop, same iteration count."""
    import torch.nn.functional as F
    from dlmc.quantization.scalar import ops

    class Layer:                       # what the estimator needs of a wrapper: _forward_func (modules/conv.py:13-19)
        def _forward_func(self, x, w):
            return F.conv2d(x, w, None, 1, 1)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 224, 224, generator=g)
    w = torch.randn(16, 3, 3, 3, generator=g) * 0.2
    conv = torch.nn.Conv2d(3, 16, 3, padding=1, bias=False)
    s_ref, _ = O.l2norm_output(conv, x, w, 4, True, patience=8)
    s, _ = ops.quantize_l2norm_output(x.to(DEV), w.to(DEV), Layer(), 4, True, patience=8)
    torch.testing.assert_close(s.cpu().reshape(()), s_ref.reshape(()), rtol=2e-4, atol=0)
    s_ch, _ = ops.quantize_l2norm_output_channel(x.to(DEV), w.to(DEV), Layer(), 4, True, patience=4)      # the per-channel plan still fits
    assert s_ch.shape == (16, 1, 1, 1) and bool(torch.isfinite(s_ch).all())
