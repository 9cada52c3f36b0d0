"""The extended-precision adjudicator (oracle/lpvs_oracle_ld.c) and the fixture made with it.  CPU only.

The adjudicator is lpvo_admm_gram (src/lasso.jl:136-171 on the Gram form) carried in x87 extended precision; at sizes where f64
rounding is far below the tolerances it must reproduce the f64 oracle's iterates for every prox operator, and with a Gram whose
entries are small integers (every product and sum exact in both formats for the first iterations) it must agree with it bit for
bit in the x-update's right-hand side.  The committed cfg3 fixture is data: counts, iterates, the fingerprint of its G, b."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["l1", "l0", "ball", "group"])
def test_extended_precision_adjudicator_reproduces_the_f64_oracle(oracle, kind):
    rng = np.random.default_rng(7)
    m, n = 900, 192
    A = rng.standard_normal((m, n)) * np.logspace(0, -1.5, n)[None, :]
    xt = np.zeros(n); xt[rng.choice(n, 12, replace=False)] = 3 * rng.standard_normal(12)
    y = A @ xt + 0.05 * rng.standard_normal(m)
    G, b = A.T @ A, A.T @ y
    pg = {"l1": oracle.NormL1(0.3), "l0": oracle.NormL0(0.05), "ball": oracle.IndBallL0(10), "group": oracle.GroupL2(0.4, 16)}[kind]
    snaps = [1, 40, 150]
    ld = oracle.admm_gram_ld(G, b, pg, snaps, mu=0.05)
    for c in snaps:
        r = oracle.admm_gram(G, b, pg, iters=c, tol=0.0, mu=0.05)
        for k, name in enumerate(("x", "z", "u")):
            d = np.linalg.norm(r[name] - ld[c][k]) / max(np.linalg.norm(ld[c][k]), 1e-300)
            assert d <= 2e-12, (kind, c, name, d)
        assert np.array_equal(r["z"] != 0, ld[c][1] != 0), (kind, c)
    # a start vector is honoured the same way (z = x0, u = 0: src/lasso.jl:146-147)
    x0 = rng.standard_normal(n)
    l0 = oracle.admm_gram_ld(G, b, pg, [5], x0=x0, mu=0.05)[5]
    r0 = oracle.admm_gram(G, b, pg, x0=x0, iters=5, tol=0.0, mu=0.05)
    assert np.linalg.norm(r0["z"] - l0[1]) <= 1e-12 * max(np.linalg.norm(l0[1]), 1.0)


def test_cfg3_extended_precision_fixture_is_wellformed():
    f = np.load(os.path.join(ROOT, "tests", "golden", "cfg3_extended_precision_iterates.npz"))
    assert list(f["counts"]) == [200, 500, 1000, 2000]
    assert f["x"].shape == f["z"].shape == f["u"].shape == (4, 8192) and len(str(f["sha256"])) == 64
    assert f["oracle_x"].shape == f["oracle_z"].shape == f["oracle_u"].shape == (4, 8192)
    # the f64 oracle's own distance to the exact iterates: the part of "device vs oracle" no device can remove
    d = [float(np.linalg.norm(a - b) / np.linalg.norm(b)) for a, b in zip(f["oracle_z"], f["z"])]
    assert 1e-10 < d[0] < 3e-10 and 6e-10 < d[3] < 8e-10 and d[0] < d[1] < d[2] < d[3]
    nnz = [int(np.count_nonzero(z)) for z in f["z"]]
    assert nnz[0] == 7904 and nnz[1:] == [8192] * 3 and all(np.isfinite(f[k]).all() for k in "xzu")
    # ADMM invariants that hold for any exact run: groups of z are zero or full, and x - z shrinks as the run proceeds
    for z in f["z"]:
        g = z.reshape(512, 16) != 0
        assert np.all(g.all(axis=1) | (~g).all(axis=1))
    d = [np.linalg.norm(x - z) for x, z in zip(f["x"], f["z"])]
    assert d[0] > d[1] > d[2] > d[3] > 0
