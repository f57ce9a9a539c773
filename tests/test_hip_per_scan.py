"""GPU: the cumulative table of the prioritised samplers' std::discrete_distribution (Sampling.cpp:101-296; libstdc++'s
param_type::_M_initialize: a sequential fp64 accumulate, a division, a sequential partial_sum) built WITHOUT walking it element by
element (per.hip: per_scan_kernel and its grid form -- integer prefix sums inside a binade, real fp64 additions at binade
crossings and exact ties).  The drawn indices depend on every rounding of those chains, so the table must equal the sequential one bit for bit:
against numpy's sequential cumsum on the host and against the sequential walk kernel, on a million elements of the samplers'
value ranges and on inputs made of ties and binade crossings."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _table(api, p, which):
    f = api.lib.hl_debug_per_scan
    f.restype = C.c_double; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    p = np.ascontiguousarray(p, dtype=np.float32); cp = np.zeros(len(p), dtype=np.float64)
    ms = f(p.ctypes.data, cp.ctypes.data, len(p), which)
    assert ms >= 0
    return cp, ms


def _host_table(p):
    x = p.astype(np.float32).astype(np.float64)
    s = np.cumsum(x)[-1]                      # (numpy's cumsum is the sequential chain; its sum() is pairwise and would not do)
    cp = np.cumsum(x / s); cp[-1] = 1.0
    return cp


def _cases():
    g = np.random.default_rng(11)
    n = 1 << 20
    yield "PERerr-like", np.sqrt(np.sqrt(g.standard_normal(n) ** 2 + np.finfo(np.float32).eps))
    yield "PERrank-like", (1.0 / np.sqrt(np.sqrt(np.arange(1, n + 1, dtype=np.float64))))[g.permutation(n)]
    yield "wide range", g.random(n) ** 12 * 1e4 + 1e-7
    yield "ragged end", g.random(n - 12345) + 0.25
    yield "short", g.random(5000) + 0.01
    yield "two", np.array([0.3, 0.9])
    # sums that leave the range where adding floats is exact (bits below the accumulator's last place: ties in the FIRST chain)
    yield "inexact sum", np.concatenate([[3e7], g.integers(1, 1 << 10, 300000) * 2.0 ** -7])
    # powers of two and small integers: quotients with few mantissa bits -- ties and exact binade hits in the second chain
    yield "dyadic", 2.0 ** g.integers(-6, 3, 262144 + 77)
    yield "small integers", g.integers(1, 4, 1 << 19).astype(np.float64)


@pytest.mark.parametrize("name,p", list(_cases()), ids=[c[0] for c in _cases()])
def test_scanned_table_equals_the_sequential_chains_bit_for_bit(hip_api, name, p):
    ref = _host_table(np.asarray(p))
    got, ms = _table(hip_api, p, 0)              # the grid form where the table is long enough (else one workgroup)
    assert np.array_equal(got, ref), (name, int(np.sum(got != ref)), int(np.argmax(got != ref)))
    one, ms_one = _table(hip_api, p, 2)          # one workgroup whatever the length
    assert np.array_equal(one, ref), (name, int(np.sum(one != ref)), int(np.argmax(one != ref)))
    if len(ref) <= (1 << 18) or name == "PERerr-like":      # (the walk takes 23 ms per million elements)
        seq, ms_seq = _table(hip_api, p, 1)
        assert np.array_equal(seq, ref)
        if name == "PERerr-like": print("\ngrid form %.3f ms, one workgroup %.3f ms, sequential walk %.3f ms on %d elements" % (ms, ms_one, ms_seq, len(ref)))


def test_grid_form_on_ten_million_elements(hip_api):
    """A replay ten times the BASELINE size: 610 chunks, most of them stepped over by their integer totals."""
    g = np.random.default_rng(5)
    p = np.sqrt(np.sqrt(g.standard_normal(10_000_000) ** 2 + np.finfo(np.float32).eps))
    ref = _host_table(p)
    got, ms = _table(hip_api, p, 0)
    assert np.array_equal(got, ref), (int(np.sum(got != ref)), int(np.argmax(got != ref)))
    print("\ngrid form on 1e7 elements: %.3f ms" % ms)
