"""CPU: the arithmetic behind per.hip's per_scan_kernel, restated in numpy.  libstdc++'s std::discrete_distribution builds its table
with two SEQUENTIAL fp64 chains acc_i = fl(acc_{i-1} + x_i) (Sampling.cpp:101-296 rebuilds it before every prioritised minibatch).
Claim the kernel rests on: while acc stays inside one binade [2^e, 2^(e+1)) it is an integer multiple M of u = 2^(e-52) and a
round-to-nearest-even addition is the INTEGER step M += floor(x/u) + [frac(x/u) > 1/2], except for (a) the addition that carries acc
into the next binade and (b) an exact tie frac == 1/2 -- both done as one real fp64 addition.  The emulation below takes a chain in
chunks exactly as the kernel does (integer prefix sums, first crossing / tie found, rest of the chunk rescanned) and must equal
numpy's sequential cumsum bit for bit, on the samplers' value ranges, on quotients, and on inputs where every element is a tie."""
import numpy as np
import pytest

TOP, SAT = 1 << 53, 1 << 54


def chain(x, CH=4096, HEAD=64):
    n = len(x); out = np.empty(n); acc = 0.0
    head = min(n, HEAD)
    for j in range(head):
        acc = acc + x[j]; out[j] = acc
    pos = head; passes = 0
    while pos < n:
        cn = min(CH, n - pos); xs = x[pos:pos + cn]; done = 0
        while done < cn:
            passes += 1
            _, ex = np.frexp(acc); ex = int(ex) - 1                      # acc in [2^ex, 2^(ex+1))
            scale = 2.0 ** (52 - ex); u = 2.0 ** (ex - 52); M0 = int(acc * scale)
            assert (1 << 52) <= M0 < TOP
            y = xs[done:] * scale                                        # exact: a power of two
            big = y >= 9007199254740992.0
            k = np.where(big, 0, np.floor(np.where(big, 0, y))).astype(np.int64)
            f = np.where(big, 0.0, y - k)
            cc = np.where(big, SAT, k + (f > 0.5)).astype(object)      # python ints: the kernel saturates at 2^54 instead
            tie = (~big) & (f == 0.5)
            M = M0 + np.cumsum(cc)
            cross = np.nonzero(np.array([int(v) >= TOP for v in M]))[0]
            ti = np.nonzero(tie)[0]
            stop = min(cross[0] if len(cross) else cn - done, ti[0] if len(ti) else cn - done)
            for g in range(stop):
                out[pos + done + g] = float(int(M[g])) * u
            before = int(M[stop - 1]) if stop > 0 else M0
            if done + stop < cn:                                         # the crossing / tie element: one real addition
                acc = float(before) * u + xs[done + stop]; out[pos + done + stop] = acc; done = done + stop + 1
            else:
                acc = float(before) * u; done = cn
        pos += cn
    return out, passes


def _inputs():
    g = np.random.default_rng(1)
    n = 20000
    yield "PERerr-like", np.sqrt(np.sqrt(g.standard_normal(n) ** 2 + np.finfo(np.float32).eps)).astype(np.float32).astype(np.float64)
    yield "PERrank-like", (1.0 / np.sqrt(np.sqrt(np.arange(1, n + 1, dtype=np.float64))))[g.permutation(n)].astype(np.float32).astype(np.float64)
    yield "wide range", g.random(n) ** 8 + 1e-12
    yield "dyadic", 2.0 ** g.integers(-6, 3, n)
    yield "every element a tie", np.concatenate([[0.75], (2 * g.integers(0, 50, 600) + 1) * 2.0 ** -54, g.random(300) * 1e-3,
                                                 (2 * g.integers(0, 50, 100) + 1) * 2.0 ** -53])


@pytest.mark.parametrize("name,x", list(_inputs()), ids=[c[0] for c in _inputs()])
def test_integer_steps_inside_a_binade_reproduce_the_sequential_chain(name, x):
    ref = np.cumsum(x)                           # numpy's cumsum IS the sequential chain
    got, passes = chain(x, CH=256 if name.startswith("every") else 4096, HEAD=1 if name.startswith("every") else 64)
    assert np.array_equal(ref, got), int(np.sum(ref != got))
    q = x / ref[-1]                              # the second chain: partial sums of the quotients
    got2, _ = chain(q)
    assert np.array_equal(np.cumsum(q), got2)
    if name.startswith("every"): assert passes > 600          # (each tie costs a pass: they really are ties)
