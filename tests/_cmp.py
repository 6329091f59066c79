"""Bitwise comparison helpers shared by the CPU and GPU suites."""
import numpy as np
import torch


def _np(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(t, dtype=np.float32))


def assert_bits_equal(got, want, what=""):
    """fp32 arrays equal bit for bit (signed zeros distinguished); every NaN equals every NaN."""
    g, w = _np(got).reshape(-1), _np(want).reshape(-1)
    assert g.shape == w.shape, f"{what}: shape {g.shape} vs {w.shape}"
    gn, wn = np.isnan(g), np.isnan(w)
    assert np.array_equal(gn, wn), f"{what}: NaN pattern differs at {np.flatnonzero(gn != wn)[:8]}"
    gb, wb = g.view(np.uint32)[~gn], w.view(np.uint32)[~wn]
    bad = np.flatnonzero(gb != wb)
    if bad.size:
        i = bad[:8]
        raise AssertionError(f"{what}: {bad.size}/{g.size} values differ, e.g. got {g[~gn][i]} want {w[~wn][i]}")


def ulp_distance(got, want):
    """Max distance in units in the last place over the finite, non-NaN entries."""
    g, w = _np(got).reshape(-1), _np(want).reshape(-1)
    ok = np.isfinite(g) & np.isfinite(w)

    def key(a):
        i = a.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    if not ok.any():
        return 0
    return int(np.abs(key(g[ok]) - key(w[ok])).max())
