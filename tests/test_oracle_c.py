"""The scalar C restatement (oracle/fq_oracle.c) pinned to the same golden vectors as the torch oracle."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from _cmp import assert_bits_equal
from oracle import fakequant_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def C():
    so = os.path.join(ROOT, "oracle", "_build", "libfqoracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(so)


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def f32(t):
    return np.ascontiguousarray(t.detach().numpy().astype(np.float32))


def geom(x, s):
    if s.numel() == 1:
        return 1, x.numel()
    ax = [i for i, d in enumerate(s.shape) if d != 1][0]
    return x.shape[ax], int(np.prod(x.shape[ax + 1:]))


I64, F = ctypes.c_int64, ctypes.c_float


def test_c_emulate_against_golden(C, golden):
    for c in golden.of_kind("primitive"):
        x, s, o = golden.get(c, "x"), golden.get(c, "scale"), golden.get(c, "offset")
        ch, inner = geom(x, s)
        xs, ss = f32(x).reshape(-1), f32(s).reshape(-1)
        os_ = f32(o.expand_as(s) if o.numel() != s.numel() else o).reshape(-1)
        q, y = np.empty_like(xs), np.empty_like(xs)
        C.fqo_emulate(fp(xs), fp(q), fp(y), fp(ss), fp(os_), I64(xs.size), I64(ch), I64(inner), F(c["lo"]), F(c["hi"]))
        assert_bits_equal(q, golden.get(c, "q"), c["name"] + ".q")
        assert_bits_equal(y, golden.get(c, "y"), c["name"] + ".y")


def test_c_observers_against_golden(C, golden):
    for c in golden.of_kind("observer"):
        x = golden.get(c, "x")
        for per_channel in (False, True):
            if per_channel and not golden.has(c, "c_scale"):
                continue
            if per_channel:
                ax = c["ch_axis"]
                outer, ch, inner = int(np.prod(x.shape[:ax])), x.shape[ax], int(np.prod(x.shape[ax + 1:]))
            else:
                outer, ch, inner = 1, 1, x.numel()
            xs = f32(x).reshape(-1)
            mx, mn, ab = (np.empty(ch, np.float32) for _ in range(3))
            C.fqo_minmax(fp(xs), fp(mx), fp(mn), fp(ab), I64(outer), I64(ch), I64(inner))
            sc, of = np.empty(ch, np.float32), np.empty(ch, np.float32)
            C.fqo_qparams(fp(mx), fp(mn), fp(ab), fp(sc), fp(of), I64(ch), c["n_bits"], int(c["signed"]), 1, F(0.0))
            want_s = golden.get(c, "c_scale" if per_channel else "t_scale").reshape(-1)
            want_o = golden.get(c, "c_offset" if per_channel else "t_offset").reshape(-1)
            for got, want in ((sc, want_s), (of, want_o)):
                g, w = torch.from_numpy(got).double(), want.double()
                assert bool(((g == w) | (g.isnan() & w.isnan())).all()), c["name"]


def test_c_layer_forms_against_golden(C, golden):
    for c in golden.of_kind("qbase"):
        ia = c["qconfig"]["input"]["args"]
        lo, hi = O.qrange(ia["signed"], ia["n_bits"])
        x = f32(golden.get(c, "x")).reshape(-1)
        s, o = f32(golden.get(c, "in_scale")).reshape(-1), f32(golden.get(c, "in_offset")).reshape(-1)
        q, y = np.empty_like(x), np.empty_like(x)
        C.fqo_qbase(fp(x), fp(q), fp(y), fp(s), fp(o), I64(x.size), I64(1), I64(x.size), F(lo), F(hi),
                    F(1 / math.sqrt(x.size * hi)))
        assert_bits_equal(y, golden.get(c, "fq_input"), c["name"])
    for c in golden.of_kind("fsptq"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        lo, hi = O.qrange(ia["signed"], ia["n_bits"])
        x = f32(golden.get(c, "x")).reshape(-1)
        s, z = f32(golden.get(c, "in_scale")).reshape(-1), f32(golden.get(c, "in_offset")).reshape(-1)
        q, y = np.empty_like(x), np.empty_like(x)
        C.fqo_zeropoint(fp(x), fp(q), fp(y), fp(s), fp(z), I64(x.size), I64(1), I64(x.size), F(lo), F(hi))
        assert_bits_equal(y, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        if c["qconfig"]["weight"]["recon_type"] != "adaround":
            wlo, whi = O.qrange(wa["signed"], wa["n_bits"])
            w = golden.get(c, "weight")
            ws, sc = f32(w).reshape(-1), f32(golden.get(c, "wt_scale")).reshape(-1)
            q, y = np.empty_like(ws), np.empty_like(ws)
            C.fqo_symmetric(fp(ws), fp(q), fp(y), fp(sc), I64(ws.size), I64(w.shape[0]), I64(ws.size // w.shape[0]), F(wlo), F(whi))
            assert_bits_equal(y, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")
    for c in golden.of_kind("rootq"):
        ia, wa = c["qconfig"]["input"]["args"], c["qconfig"]["weight"]["args"]
        lo, hi = O.qrange(ia["signed"], ia["n_bits"])
        x = f32(golden.get(c, "x")).reshape(-1)
        s = f32(golden.get(c, "st_in_run_scale")).reshape(-1)
        q, y = np.empty_like(x), np.empty_like(x)
        C.fqo_rootq_act(fp(x), fp(q), fp(y), fp(s), I64(x.size), I64(1), I64(x.size), F(lo), F(hi))
        assert_bits_equal(y, golden.get(c, "fq_input"), c["name"] + ".fq_input")
        wlo, whi = O.qrange(wa["signed"], wa["n_bits"])
        w = f32(golden.get(c, "weight")).reshape(-1)
        y = np.empty_like(w)
        C.fqo_rootq_weight(fp(w), fp(y), I64(w.size), F(float(golden.get(c, "st_wt_run_upper"))),
                           F(float(golden.get(c, "st_wt_run_lower"))), F(float(golden.get(c, "wt_alpha"))), F(wlo), F(whi))
        assert_bits_equal(y, golden.get(c, "fq_weight"), c["name"] + ".fq_weight")


def test_c_backward_against_golden(C, golden):
    for c in golden.of_kind("qbase_grad"):
        ia = c["qconfig"]["input"]["args"]
        lo, hi = O.qrange(ia["signed"], ia["n_bits"])
        x, gy = f32(golden.get(c, "x")).reshape(-1), f32(golden.get(c, "g_fq_input")).reshape(-1)
        gx = np.empty_like(x)
        acc = ctypes.c_double(0)
        g = 1 / math.sqrt(x.size * hi)
        C.fqo_qbase_backward(fp(x), fp(gy), fp(gx), ctypes.byref(acc), F(float(golden.get(c, "in_scale"))),
                             F(float(golden.get(c, "in_offset"))), I64(x.size), F(lo), F(hi), F(g))
        assert_bits_equal(gx, golden.get(c, "grad_x"), c["name"] + ".grad_x")
        assert abs(acc.value * g - float(golden.get(c, "grad_in_scale"))) <= 2e-4 * abs(float(golden.get(c, "grad_in_scale"))) + 1e-6
