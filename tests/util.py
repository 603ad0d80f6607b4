"""Bit-exact comparison helpers for the parity tests."""
import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    if a.dtype == np.float64:
        return a.view(np.uint64)
    return a


def assert_bits_equal(got, want, what=""):
    """Bit-for-bit equality (NaNs compare equal when both are NaN; -0.0 != +0.0)."""
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    gb, wb = bits(got), bits(want)
    bad = gb != wb
    if got.dtype.kind == "f":
        bad &= ~(np.isnan(got) & np.isnan(want))
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError("%s: %d of %d values differ, first at %s: got %s want %s" % (
            what, int(bad.sum()), bad.size, idx.tolist(), got[tuple(idx[0])], want[tuple(idx[0])]))


def miss_equal(got_t, want_t, what=""):
    """Closest-hit distances: negative means miss (any negative value), non-negative must match bit for bit."""
    gm, wm = got_t < 0, want_t < 0
    assert (gm == wm).all(), "%s: hit/miss differs for %d rays" % (what, int((gm != wm).sum()))
    assert_bits_equal(got_t[~wm], want_t[~wm], what)
