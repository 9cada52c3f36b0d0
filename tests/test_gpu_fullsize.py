"""Size-independent properties of the HIP path at BASELINE.json's sizes (where the CPU oracle cannot run)
and one mid-size comparison against the oracle's Gram form.  GPU only.

Properties used (each holds exactly or to rounding for the reference algorithm as well):
  * additivity over samples:  G(all rows) = G(first part) + G(second part), b likewise          (1e-12)
  * the two Gram forms (n x n lower triangle vs symmetric-pair contraction) agree                (1e-12)
  * linearity of b in y;  symmetry of G;  pad / edge tiles carry no garbage
  * ADMM invariants at any iteration: z = prox_g(x + u_prev) recomputed on the host bit-for-bit from the
    returned iterates is a fixed point of the prox (prox is idempotent on its own output), the reported
    ||x - z|| equals the norm of the returned vectors, u = sum of (x - z) increments stays consistent
  * optimality at convergence (solver independent, mu cancels): group lasso KKT
        ||Phi_g'(y - Phi z)|| <= lam on inactive groups, = lam z_g/||z_g|| on active ones
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def cfg3_inputs(log2n, Nf=512, seed=0):
    import bench
    return bench.synth_signal(1 << log2n, Nf, seed, torch.device("cuda"))


def test_cfg3_gram_additivity_and_forms(L):
    """N = 2^18 rows at the full n = 8192: G and b are sums over samples (interleaved halves); both Gram forms agree."""
    y, X, V, w = cfg3_inputs(18)
    N = len(y)
    Vfix = V.clone()
    grams = {}
    for form in ("ap", "krs", "kr"):          # structured (w is an arithmetic progression), symmetric-pair MFMA, plain MFMA
        os.environ["LPVS_GRAM_FORM"] = form
        try:
            with L.Problem.lpv(y, X, Vfix, w, 8) as p:
                grams[form] = p.get_gram()
                n = p.n
        finally:
            del os.environ["LPVS_GRAM_FORM"]
    G, b = grams["krs"]
    assert n == 8192 and np.array_equal(G, G.T) and np.isfinite(G).all()
    scale = np.abs(G).max()
    assert np.abs(G - grams["kr"][0]).max() <= 1e-12 * scale and np.array_equal(b, grams["kr"][1])   # same operands, same b kernel
    # structured vs dense: exact phases vs the reference's rounded phases fl(w*x): |w x| 2^-53 rad per term
    tol = 1e-12 + 4.5e-16 * float(w.max().item() * X.max().item())
    assert np.abs(grams["ap"][0] - G).max() <= tol * scale
    assert np.abs(grams["ap"][1] - b).max() <= tol * np.abs(b).max() * 10
    with L.Problem.lpv(y, X, Vfix, w, 8) as p:        # default = auto -> structured here
        Gd, bd = p.get_gram()
    assert np.array_equal(Gd, grams["ap"][0]) and np.array_equal(bd, grams["ap"][1])
    # linearity of b in y
    with L.Problem.lpv(2.0 * y, X, Vfix, w, 8) as p:
        _, b3 = p.get_gram()
    assert np.abs(b3 - 2.0 * bd).max() <= 1e-13 * np.abs(bd).max()
    # additivity over samples.  The basis centres come from the range of the V a problem is given (src/utilities.jl:24-25), so
    # both interleaved halves keep BOTH end points: A = even rows + the last row, B = odd rows + the first row; the two rows
    # counted twice are the two-row problem E.   G(all) + G(E) = G(A) + G(B),  b likewise -- in the default and in the dense form.
    dev = y.device
    iA = torch.cat([torch.arange(0, N, 2, device=dev), torch.tensor([N - 1], device=dev)])
    iB = torch.cat([torch.tensor([0], device=dev), torch.arange(1, N, 2, device=dev)])
    iE = torch.tensor([0, N - 1], device=dev)
    for form, (Gall, ball) in (("auto", (Gd, bd)), ("krs", (G, b))):
        parts = []
        for idx in (iA, iB, iE):
            with L.default_options(gram_form=None if form == "auto" else form):
                with L.Problem.lpv(y[idx].contiguous(), X[idx].contiguous(), Vfix[idx].contiguous(), w, 8) as p:
                    parts.append(p.get_gram())
        (GA, bA), (GB, bB), (GE, bE) = parts
        eg = np.abs(Gall + GE - (GA + GB)).max() / scale
        eb = np.abs(ball + bE - (bA + bB)).max() / np.abs(ball).max()
        assert eg <= 1e-12 and eb <= 1e-12, (form, eg, eb)


def test_fourier_gram_additivity_cfg2(L):
    """cfg2 size (N = 2^18, Nf = 512, n = 1024): G(all) = G(part 1) + G(part 2) for the Fourier panel form."""
    N, Nf = 1 << 18, 512
    g = torch.Generator(device="cuda").manual_seed(2)
    t = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * N).values
    f = torch.tensor(np.arange(1, Nf + 1) / 1024.0, dtype=torch.float64, device="cuda")
    y = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    with L.Problem.fourier(y, t, f) as p:
        G, b = p.get_gram()
    h = N // 2 + 12345                       # ragged split: exercises pad rows in both parts
    with L.Problem.fourier(y[:h].contiguous(), t[:h].contiguous(), f) as p:
        G1, b1 = p.get_gram()
    with L.Problem.fourier(y[h:].contiguous(), t[h:].contiguous(), f) as p:
        G2, b2 = p.get_gram()
    assert np.abs(G - (G1 + G2)).max() <= 1e-12 * np.abs(G).max()
    assert np.abs(b - (b1 + b2)).max() <= 1e-12 * np.abs(b).max()
    # diagonal of A'A: sum_n cos^2 + sin^2 pairs -> G[k,k] + G[k+Nf,k+Nf] = N / (2 Nf) exactly in exact arithmetic
    d = np.diag(G)
    assert np.abs(d[:Nf] + d[Nf:] - N / (2.0 * Nf)).max() <= 1e-9


def _group_prox_host(v, lam, mu, glen):
    z = np.zeros_like(v)
    for s in range(0, len(v), glen):
        ss = 0.0
        for q in range(glen):
            ss += v[s + q] * v[s + q]
        nv = np.sqrt(ss)
        scale = 1.0 - lam * mu / nv if nv > 0 else 0.0
        z[s:s + glen] = max(scale, 0.0) * v[s:s + glen]
    return z


def test_cfg3_admm_invariants_fullsize(L):
    """N = 2^20, Nf = 512, Nv = 8 (the judged size), a few hundred iterations: iterate invariants."""
    y, X, V, w = cfg3_inputs(20)
    lam, mu, Nv = 5.0, 0.05, 8
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, len(w), 2 * Nv))
        p.admm_init(None, μ=mu, tol=0.0)
        it, nxz, conv = p.admm_run(150)
        x1, z1, u1 = p.admm_get()
        it2, nxz2, _ = p.admm_run(1)
        x2, z2, u2 = p.admm_get()
        G, b = p.get_gram()
    assert it == 150 and it2 == 151 and not conv
    assert abs(np.linalg.norm(x2 - z2) - nxz2) <= 1e-12 * max(nxz2, 1e-30)       # reported norm = norm of iterates
    assert np.array_equal(u2, u1 + (x2 - z2))                                    # u += x - z, bit for bit
    assert np.array_equal(z2, _group_prox_host(x2 + u1, lam, mu, 2 * Nv))        # z = prox(x + u_prev), bit for bit
    # x-update: (G + I/mu) x2 = b + (z1 - u1)/mu  (solved by the explicit inverse: residual ~ cond * eps)
    r = G @ x2 + x2 / mu - (b + (z1 - u1) / mu)
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b)
    nz = np.count_nonzero(np.abs(z2).reshape(-1, 2 * Nv).sum(1))
    assert 3 <= nz <= len(w)


def test_group_lasso_kkt_at_convergence_midsize(L, oracle):
    """n = 2048 (Nf = 128, Nv = 8), N = 2^15: run to convergence, then check the KKT conditions and compare
    with the oracle's Gram-form ADMM at equal iteration count."""
    Nf, Nv, lam, mu = 128, 8, 8.0, 0.5
    y, X, V, w = cfg3_inputs(15, Nf=Nf, seed=1)
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, Nf, 2 * Nv))
        p.admm_init(None, μ=mu, tol=1e-9)
        it, nxz, conv = p.admm_run(20000)
        x, z, u = p.admm_get()
        G, b = p.get_gram()
    assert conv and it < 20000
    grad = b - G @ z                              # Phi'(y - Phi z)
    gn = np.linalg.norm(grad.reshape(Nf, 2 * Nv), axis=1)
    zn = np.linalg.norm(z.reshape(Nf, 2 * Nv), axis=1)
    active = zn > 0
    assert 1 <= active.sum() < Nf
    assert (gn[~active] <= lam * (1 + 1e-6)).all()
    for g_ in np.nonzero(active)[0]:
        sl = slice(g_ * 2 * Nv, (g_ + 1) * 2 * Nv)
        assert np.linalg.norm(grad[sl] - lam * z[sl] / zn[g_]) <= 1e-5 * lam
    ro = oracle.admm_gram(G, b, oracle.GroupL2(lam, 2 * Nv), iters=20000, tol=1e-9, mu=mu)
    assert ro["iters"] == it
    assert rel(z, ro["z"]) <= 1e-9 and np.array_equal(z != 0, ro["z"] != 0)
    # and the Gram itself against the oracle regressor (column-major Phi on the host, BLAS syrk)
    Phi = oracle.lpv_regressor(X.cpu().numpy(), V.cpu().numpy(), w.cpu().numpy(), Nv)
    Go = Phi.T @ Phi
    assert np.abs(G - Go).max() <= 1e-12 * np.abs(Go).max()


def test_cfg4_window_shards_reproduce_the_whole(L):
    """cfg4 shape (windows of 2^16 samples, Nf = 256 with the zero frequency, L1, mu = 1e-4), 48 windows: disjoint window
    ranges -- what the ranks of a node own -- reproduce the whole run bit for bit, S is the in-order sum of |x_i|^2, and a
    window solved alone through the single-problem path agrees with its batched solution."""
    n, nwin, Nf = 1 << 16, 48, 256
    g = torch.Generator(device="cuda").manual_seed(4)
    t = torch.arange(nwin * n, dtype=torch.float64, device="cuda")
    f = np.arange(Nf) / 512.0
    y = (torch.sin(2 * np.pi * f[33] * t) + 0.5 * torch.sin(2 * np.pi * f[100] * t)
         + 0.3 * torch.randn(nwin * n, dtype=torch.float64, device="cuda", generator=g))
    kw = dict(λ=0.2, μ=1e-4, tol=0.0, iters=300)
    x, S, its = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
    assert x.shape == (nwin, Nf) and np.all(its == 300)
    Sref = np.zeros(Nf)
    for i in range(nwin):
        Sref += x[i].real ** 2 + x[i].imag ** 2
    assert np.array_equal(S, Sref)                                   # window order, src/lsfft.jl:122
    parts = [L.windowpsd_sparse_batched(y, t, f, n, 0, None, win_lo=lo, win_hi=hi, **kw)[0] for lo, hi in ((0, 17), (17, 40), (40, 48))]
    assert np.array_equal(np.vstack(parts), x)                       # shards == whole, bit for bit
    assert int(np.argmax(np.abs(x[5]))) == 33
    i = 29                                                           # one window alone (single-problem path, np = 512)
    yi, ti = y[i * n:(i + 1) * n].cpu().numpy(), t[i * n:(i + 1) * n].cpu().numpy()
    with L.Problem.fourier(yi, ti, f, np.ones(n)) as p:
        p.set_prox(L.NormL1(0.2))
        p.admm_init(None, μ=1e-4, tol=0.0, linear_sign=-1)           # Quadratic(Q, +q) as written, src/lasso.jl:119-121
        p.admm_run(300)
        xi = p.params(0)
    assert rel(xi, x[i]) <= 1e-9


def test_cfg5_shape_multichannel_equals_single_channel(L):
    """cfg5 shape (Nf = 1024, Nv = 16 -> n = 32768, IndBallL0(32)) at N = 2^20 rows, 3 channels sharing (X, V): every channel of the
    multi-signal solve (matrix-core tile product) equals its own single-signal solve (scalar tile product) to summation
    order, with the same support of 32 coefficients."""
    import bench
    N, Nf, Nv, ns = 1 << 20, 1024, 16, 3                          # (the bench's own cfg5 inputs: two thirds of its inverse's tiles qualify for fixed point)
    Y, X, V, w = bench.synth_channels(N, Nf, ns, torch.device("cuda"))
    kw = dict(proxg=L.IndBallL0(32), iters=40, tol=0.0, μ=0.05, printerval=100000)
    ses = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, **kw)
    q = 1
    se = L.ls_sparse_spectral_lpv(Y[:, q].contiguous(), X, V, w, Nv, **kw)
    assert np.count_nonzero(se.x) <= 32 and np.count_nonzero(se.x) > 0
    assert np.array_equal(np.abs(ses[q].x) > 0, np.abs(se.x) > 0)
    assert rel(ses[q].x, se.x) <= 1e-10
    # the multi-signal handle streams the MIXED storage here (36-bit fixed-point tiles wherever a tile's rows are small, decoded by the
    # loader waves of the matrix-core tile product); with uniform 6-byte tiles (storage="split") and with doubles the same iterates
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        p.set_prox(L.IndBallL0(32))
        p.admm_init(None, μ=0.05, tol=0.0)
        info = p.matvec_info()
        assert info["kernel"] == "symv_tile_mfma_ws_kernel" and "36-bit fixed point" in info["storage"], info
        us, nbytes = p.time_matvec(3)
        assert nbytes < 0.95 * 6 * 32768 * (32768 + 128) / 2          # fewer bytes than uniform 6-byte tiles
    for st, bound in (("split", 1e-10), ("f64", 1e-10)):
        sa = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, storage=st, **kw)
        assert np.array_equal(np.abs(sa[q].x) > 0, np.abs(ses[q].x) > 0) and rel(ses[q].x, sa[q].x) <= bound, (st, rel(ses[q].x, sa[q].x))


def test_window_shards_reproduce_the_whole_dense_form(L, monkeypatch):
    """Same sharding invariance on the dense (MFMA panel) batched path, taken for non-uniform frequency grids."""
    monkeypatch.setenv("LPVS_GRAM_FORM", "panel")
    n, nwin, Nf = 1 << 13, 23, 64
    rng = np.random.default_rng(3)
    t = np.arange(nwin * n, dtype=np.float64)
    f = np.arange(1, Nf + 1) / 200.0
    y = np.sin(2 * np.pi * f[20] * t) + 0.3 * rng.standard_normal(nwin * n)
    kw = dict(λ=0.2, μ=1e-3, tol=0.0, iters=100)
    x, S, its = L.windowpsd_sparse_batched(y, t, f, n, 0, None, **kw)
    parts = [L.windowpsd_sparse_batched(y, t, f, n, 0, None, win_lo=lo, win_hi=hi, **kw)[0] for lo, hi in ((0, 1), (1, 12), (12, 23))]
    assert np.array_equal(np.vstack(parts), x)


def test_lpv_batch_multi_equals_single_channel_solves(L, monkeypatch):
    """lpvs_lpv_batch_multi_f64 (channels over devices, one host thread per device): with one device, with three shards
    sharing the device (a rehearsal of three GPUs: each shard its own handle, Gram and factorisation), and against the per-channel
    single-signal solves -- same coefficients, same per-channel stopping iterations."""
    rng = np.random.default_rng(77)
    N, Nf, Nv, ns = 6000, 24, 4, 5
    X = np.sort(rng.random(N) * 80); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) / 6
    Y = np.stack([np.cos(w[(5 * q + 2) % Nf] * X) * (1 + q * V) + 0.5 * np.cos(w[(3 * q + 7) % Nf] * X + q) + 0.05 * rng.standard_normal(N) for q in range(ns)], axis=1)
    kw = dict(proxg=L.IndBallL0(6), μ=0.05, tol=1e-7, iters=3000)
    P1, it1 = L.lpv_batch_multi(Y, X, V, w, Nv, ngpus=1, **kw)
    P3, it3 = L.lpv_batch_multi(Y, X, V, w, Nv, devices=[0, 0, 0], **kw)
    assert P1.shape == (Nf * Nv, ns) and len(set(it1.tolist())) > 1 and it1.min() < 3000      # every channel stops on its own (or runs out of iterations)
    assert np.array_equal(it1, it3) and np.abs(P1 - P3).max() <= 1e-12 * np.abs(P1).max()
    import io
    for q in range(ns):
        se = L.ls_sparse_spectral_lpv(Y[:, q], X, V, w, Nv, proxg=L.IndBallL0(6), μ=0.05, tol=1e-7, iters=3000, printerval=100000, out=io.StringIO())
        assert np.abs(se.x - P1[:, q]).max() <= 1e-10 * np.abs(se.x).max(), q
        assert np.array_equal(se.x != 0, P1[:, q] != 0) and np.count_nonzero(np.concatenate([se.x.real, se.x.imag])) == 6
    G, _ = L.lpv_batch_multi(Y, X, V, w, Nv, λ=3.0, ngpus=0, μ=0.05, tol=0.0, iters=200)    # the reference's group lasso, every visible device
    S = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, λ=3.0, μ=0.05, tol=0.0, iters=200, printerval=100000, out=io.StringIO())
    assert all(np.abs(G[:, q] - S[q].x).max() <= 1e-10 * np.abs(S[q].x).max() for q in range(ns))


def test_lpv_signals_multi_equals_the_loop_over_signals(L):
    """lpvs_lpv_signals_multi_f64 (independent signals, each with its own X and V; contiguous ranges over devices, several solves in
    flight per device): the same coefficients and stopping iterations as the loop of single-signal calls, whatever the sharding --
    one solve at a time, three in flight, two shards sharing the device with two in flight each; host and device-resident inputs."""
    rng = np.random.default_rng(91)
    N, Nf, Nv, nsig = 5000, 160, 8, 5                              # n = 2560: the tile-packed path, one launch per iteration
    w = 2 * np.pi * (np.arange(Nf) + 1.0) / 8
    X = np.sort(rng.random((N, nsig)) * 60, axis=0); V = np.tile(np.linspace(0, 1, N)[:, None], (1, nsig)) ** np.arange(1, nsig + 1)
    Y = np.stack([np.cos(w[(7 * q + 3) % Nf] * X[:, q]) * (1 + q * V[:, q]) + 0.4 * np.cos(w[(11 * q + 50) % Nf] * X[:, q] + q)
                  + 0.05 * rng.standard_normal(N) for q in range(nsig)], axis=1)
    kw = dict(λ=2.0, μ=0.05, tol=1e-6, iters=800)
    ref = [L.ls_sparse_spectral_lpv(Y[:, q].copy(), X[:, q].copy(), V[:, q].copy(), w, Nv, printerval=100000, **kw).x for q in range(nsig)]
    assert all(0 < np.count_nonzero(r) < r.size for r in ref)
    P1, it1 = L.lpv_signals_multi(Y, X, V, w, Nv, ngpus=1, in_flight=1, **kw)
    P3, it3 = L.lpv_signals_multi(Y, X, V, w, Nv, ngpus=1, in_flight=3, **kw)
    P22, it22 = L.lpv_signals_multi(Y, X, V, w, Nv, devices=[0, 0], in_flight=2, **kw)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a.T), device="cuda").T       # column-major N x nsig on the device
    Pd, itd = L.lpv_signals_multi(dev(Y), dev(X), dev(V), w, Nv, ngpus=1, in_flight=2, **kw)
    for q in range(nsig):
        assert np.array_equal(P1[:, q], ref[q]), q
    for P, it in ((P3, it3), (P22, it22), (Pd, itd)):
        assert np.array_equal(P, P1) and np.array_equal(it, it1)
    assert len(set(it1.tolist())) > 1 and it1.max() <= 800          # every signal stops on its own
    with pytest.raises(ValueError):
        L.lpv_signals_multi(Y, X, V, w, Nv, in_flight=0, **kw)
