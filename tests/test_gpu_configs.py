"""Parity of the HIP path at the five BASELINE.json configurations, against the CPU oracle (or, where the oracle cannot
run at the size, against the dense-MFMA path and size-independent invariants).  GPU only.

  cfg1  ls_spectral, N = 4096, equidistant t, Nf = 256 (zero frequency first)            vs oracle SVD solve
  cfg2  ls_sparse_spectral NormL1(0.01), N = 2^18, Nf = 512, 5000 iterations (hipGraph)  vs oracle.admm_gram on the device Gram
  cfg3  ls_sparse_spectral_lpv, N = 2^20, Nf = 512, Nv = 8, 2000 iterations              structured Gram vs dense MFMA Gram, end to end
  cfg4  (windows) is covered in test_gpu_fullsize.py / test_gpu_parity.py
  cfg5  multichannel LPV, IndBallL0: oracle-size problems vs oracle.admm_gram (per-signal stopping iteration), and the
        full shape (8 channels x N = 2^20, Nf = 1024, Nv = 16, n = 32768) through per-channel invariants

Tolerances: rel-L2 <= 1e-9 with identical support and identical stopping iteration against the Gram-form oracle
(SURVEY.md section 8(d)); the cfg3 structured-vs-dense figure is explained in the test.
"""
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


# ------------------------------------------------------------------ cfg1
def test_cfg1_ls_spectral_exact_config(L, oracle):
    """N = 4096, t = 0:1:4095, f = (0:255)/512, y = sin(2 pi f[33] t) + 0.1 N(0,1), lam = 1e-10."""
    rng = np.random.default_rng(0x1B5EC + 1)
    N = 4096
    t = np.arange(N, dtype=np.float64)
    f = np.arange(256) / 512.0
    y = np.sin(2 * np.pi * f[32] * t) + 0.1 * rng.standard_normal(N)
    x, fr = L.ls_spectral(y, t, f, λ=1e-10)
    xo, _ = oracle.ls_spectral(y, t, f, lam=1e-10)
    assert x.shape == (256,) and np.array_equal(fr, f)
    assert rel(x, xo) <= 1e-10, rel(x, xo)
    assert int(np.argmax(np.abs(x))) == 32 and x[0].imag == 0
    A, zf = L.get_fourier_regressor(t, f)
    Ao, zo = oracle.get_fourier_regressor(t, f)
    assert zf == zo == 1 and A.shape == (N, 511)
    assert np.abs(A - Ao).max() <= 4e-16 * (1 + 2 * np.pi * f.max() * t.max()) / np.sqrt(512)    # phases up to 1.3e4 rad


# ------------------------------------------------------------------ cfg2
def cfg2_inputs(equidistant):
    N, Nf = 1 << 18, 512
    g = torch.Generator(device="cuda").manual_seed(0x1B5EC + 2)
    if equidistant:
        t = torch.arange(N, dtype=torch.float64, device="cuda")
    else:
        t = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=g) * N).values
    f = np.arange(1, Nf + 1) / 1024.0
    ph = torch.rand(5, dtype=torch.float64, device="cuda", generator=g) * 2 * np.pi
    y = sum(a * torch.sin(2 * np.pi * f[i - 1] * t + ph[k]) for k, (i, a) in enumerate(zip((17, 100, 257, 300, 480), (2, 1, .5, .25, .1))))
    y = y + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    return y, t, f


@pytest.mark.parametrize("equidistant", [False, True])
def test_cfg2_fullsize_admm_vs_oracle(L, oracle, equidistant):
    """cfg2 at its full size: NormL1(0.01), mu = 0.05, exactly 5000 iterations (tol = 0).  The 5000 device iterations (hipGraph
    replay of the ONE-launch iteration for np < 2048, admm_small_iter_kernel -- the kernel bench.py's cfg2 line times; its name is
    asserted) against the oracle's Gram-form ADMM on the host, started from the same Gram."""
    y, t, f = cfg2_inputs(equidistant)
    with L.Problem.fourier(y, t, f) as p:
        G, b = p.get_gram()
        p.set_prox(L.NormL1(0.01))
        p.admm_init(None, μ=0.05, tol=0.0)
        info = p.matvec_info()
        assert info["kernel"] == "admm_small_iter_kernel" and info["one_launch_iteration"], info
        it, nxz, conv = p.admm_run(5000)
        x, z, u = p.admm_get()
        params = p.params(0)
    assert p.n == 1024 and it == 5000 and not conv
    ro = oracle.admm_gram(G, b, oracle.NormL1(0.01), iters=5000, tol=0.0, mu=0.05, history=True)
    assert ro["iters"] == 5000
    assert rel(z, ro["z"]) <= 1e-9 and rel(x, ro["x"]) <= 1e-9 and rel(u, ro["u"]) <= 1e-9, (rel(z, ro["z"]), rel(u, ro["u"]))
    assert np.array_equal(z != 0, ro["z"] != 0)
    assert abs(nxz - ro["nxz"][-1]) <= 1e-6 * ro["nxz"][-1] + 1e-13 * np.linalg.norm(ro["x"])
    top = np.sort(np.argsort(-np.abs(params))[:4] + 1)
    assert list(top) == [17, 100, 257, 300]
    # the Gram itself: diag pairs of A'A sum to N/(2Nf) exactly in exact arithmetic; symmetric; finite
    d = np.diag(G)
    assert np.array_equal(G, G.T) and np.abs(d[:512] + d[512:] - (1 << 18) / 1024.0).max() <= 1e-9


# ------------------------------------------------------------------ cfg3
CFG3_FLOOR_BOUND = 3e-10    # the SAME Gram through two factorisation schedules (different roundings of M): measured 0.8e-10 / 1.1e-10 (round 4, before
                             # the x-update correction: 1.1 .. 1.5e-9)
CFG3_PAIR_BOUND = 5e-9       # two solves on DIFFERENT Grams (structured / dense, exact / rounded phases: max|dG| = 5e-14 max|G|, a few hundred ulps of
                             # its large entries): measured 1.6 .. 2.7e-9 -- the conditioning of the problem (ONE ulp of input uncertainty moves the
                             # iterate by 1.0e-10: profiles/r05_cfg3_error_directions.txt), not an evaluation error
def test_cfg3_fullsize_structured_vs_dense_end_to_end(L):
    """The benchmarked path (structured Gram -> factorisation -> 2000 iterations at N = 2^20) against the same solve on the
    dense f64-MFMA Gram (LPVS_GRAM_FORM=krs), as an experiment that separates WHAT makes two solves of this size differ:

      structured    phases of the real products w*x (double-double slot frequencies, FMA-exact products; nudft.hip / nufft.hip)
      dense-exact   the dense MFMA Gram on a trig table with the same unrounded phases (LPVS_PHASE=exact, basis.hip)
      dense-rounded the dense MFMA Gram on the reference's own phases fl(w*x) (src/lasso.jl:39 rounds the product first)
      ... and `structured` / `dense-exact` once more with the round-2 factorisation schedule (LPVS_FACTOR_SCHEME=steps): the SAME Gram
      bit for bit (both Gram paths are deterministic), the same algorithm, only the order in which the inverse's sums are rounded.

    (0) SAME INPUTS, two evaluation orders: the same Gram through two summation orders of the factorisation.  |M H - I| is 2e-13 either
        way; rounds 3 and 4 measured 1.1 .. 1.5e-9 between the two after 2000 iterations and called it a floor.  It was the explicit
        inverse's systematic error E w, a constant forcing of the not-yet-converged map; with the x-update
        correction the two schedules agree to 1e-10 (CFG3_FLOOR_BOUND).
    (1) structured vs dense-exact: the same mathematical problem through entirely different kernel chains, whose Grams differ by
        max|dG| = 5e-14 max|G| -- DIFFERENT INPUTS to the iteration, a few hundred ulps of G's large entries apart.
    (2) dense-exact vs dense-rounded: ONE kernel chain, the phases perturbed by <= ulp(w*x)/2 = 3.7e-10 rad: what the reference's
        fl(w*x) costs.
    (1) and (2) measure 1.6 .. 2.7e-9: the conditioning of the problem -- one ulp of uncertainty in G, b moves the 2000th iterate by
    1.0e-10 (tools/cfg3_vs_oracle.py --perturb) -- and no evaluation can be closer to another than their inputs allow: CFG3_PAIR_BOUND.
    Identical supports throughout.  Against the ORACLE on the device's own Gram the benchmarked path is within 1e-9 at every count
    (tests/test_gpu_judged_size.py)."""
    import bench
    y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
    out = {}
    for name, form, phase, scheme in (("structured", "ap", None, None), ("dense-exact", "krs", "exact", None), ("dense-rounded", "krs", None, None),
                                      ("structured/steps", "ap", None, "steps"), ("dense-exact/steps", "krs", "exact", "steps")):
        os.environ["LPVS_GRAM_FORM"] = form
        if phase:
            os.environ["LPVS_PHASE"] = phase
        if scheme:
            os.environ["LPVS_FACTOR_SCHEME"] = scheme
        try:
            with L.Problem.lpv(y, X, V, w, 8) as p:
                G = p.device_gram()[0].clone() if name.startswith(("structured", "dense-exact")) else None
                p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, 512, 16))
                p.admm_init(None, μ=0.05, tol=0.0)
                assert p.matvec_info()["kernel"] == "admm_iter_mixed_kernel"
                it, nxz, conv = p.admm_run(2000)
                x, z, u = p.admm_get()
                out[name] = dict(z=z, x=x, it=it, nxz=nxz, form=p.timing()["gram_form"], G=G)
        finally:
            for v in ("LPVS_GRAM_FORM", "LPVS_PHASE", "LPVS_FACTOR_SCHEME"):
                os.environ.pop(v, None)
    assert out["structured"]["form"] in ("ap", "ap-nufft") and out["dense-exact"]["form"] == out["dense-rounded"]["form"] == "krs"
    assert all(o["it"] == 2000 for o in out.values())
    assert torch.equal(out["structured"]["G"], out["structured/steps"]["G"]) and torch.equal(out["dense-exact"]["G"], out["dense-exact/steps"]["G"])
    gdiff = float((out["structured"]["G"] - out["dense-exact"]["G"]).abs().max() / out["dense-exact"]["G"].abs().max())
    z = {k: o["z"] for k, o in out.items()}
    groups = lambda v: np.abs(v).reshape(512, 16).sum(1) > 0
    for k in z:
        assert np.array_equal(groups(z[k]), groups(z["structured"])) and np.array_equal(z[k] != 0, z["structured"] != 0), k   # identical support, all five
    floor_s, floor_e = rel(z["structured"], z["structured/steps"]), rel(z["dense-exact"], z["dense-exact/steps"])
    r_se, r_er, r_sr = rel(z["structured"], z["dense-exact"]), rel(z["dense-exact"], z["dense-rounded"]), rel(z["structured"], z["dense-rounded"])
    print(f"cfg3 N=2^20, 2000 iterations, rel-L2(z): noise floor (same Gram, two factorisation orders) {floor_s:.3e} / {floor_e:.3e} | "
          f"structured vs dense-exact {r_se:.3e} (max|dG|/max|G| = {gdiff:.1e}) | dense-exact vs dense-rounded {r_er:.3e} | "
          f"structured vs dense-rounded {r_sr:.3e}; active groups {int(groups(z['structured']).sum())}, "
          f"phase bound 2^-53*max|w x| = {2.0 ** -53 * float(w.max() * X.max()):.2e} rad")
    for name, r in (("floor/structured", floor_s), ("floor/dense", floor_e)):
        assert r <= CFG3_FLOOR_BOUND, (name, r)
    for name, r in (("structured vs dense-exact", r_se), ("exact vs rounded", r_er), ("structured vs rounded", r_sr)):
        assert r <= CFG3_PAIR_BOUND, (name, r)
    assert max(floor_s, floor_e) > 1e-12                                           # (the floor is real: the two schedules do round differently)
    assert {40, 204, 409} <= set(np.nonzero(groups(z["structured"]))[0])          # the three true frequencies are active


# ------------------------------------------------------------------ cfg5 at oracle size
@pytest.mark.parametrize("Nf", [12, 140])
def test_cfg5_indball_multichannel_vs_oracle(L, oracle, Nf):
    """Multichannel LPV with IndBallL0(r) (the cfg5 estimator; an API extension, SURVEY 8(b) "Gaps") at a size the oracle
    runs: ns = 3 channels sharing (X, V, w), N = 1500, Nv = 8 -- n = 192 (plain mat-vec path) and n = 2240 (tile-packed
    path, matrix-core multi-signal product).  Every channel against oracle.admm_gram on the same Gram: rel-L2 <= 1e-9,
    identical support, identical per-channel stopping iteration; and the single-signal call likewise."""
    rng = np.random.default_rng(13)
    N, ns, Nv, r = 1500, 3, 8, 6
    X = np.sort(10 * rng.random(N)); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf
    Y = np.stack([np.cos(w[(3 * q + 1) % Nf] * X) * (1 + q * V) + 0.3 * np.sin(w[(5 * q + 2) % Nf] * X) + 0.05 * rng.standard_normal(N)
                  for q in range(ns)], axis=1)
    kw = dict(iters=600, tol=1e-4, μ=0.05)
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        G, _ = p.get_gram()
        B = p.get_rhs()
        p.set_prox(L.IndBallL0(r))
        p.admm_init(None, μ=0.05, tol=1e-4)
        p.admm_run(600)
        per = [p.admm_status(q) for q in range(ns)]
        x, z, u = p.admm_get()
        P = p.params(0).reshape(-1, ns, order="F")
    Phi = oracle.lpv_regressor(X, V, w, Nv)
    Go = Phi.T @ Phi
    assert np.abs(G - Go).max() <= 1e-12 * np.abs(Go).max()
    stops = []
    for q in range(ns):
        assert np.abs(B[:, q] - Phi.T @ Y[:, q]).max() <= 1e-12 * np.abs(B[:, q]).max()
        ro = oracle.admm_gram(G, B[:, q], oracle.IndBallL0(r), iters=600, tol=1e-4, mu=0.05)
        assert per[q][0] == ro["iters"] and per[q][2] == (ro["iters"] < 600), (q, per[q], ro["iters"])
        assert rel(z[:, q], ro["z"]) <= 1e-9 and rel(x[:, q], ro["x"]) <= 1e-9 and rel(u[:, q], ro["u"]) <= 1e-9, (q, rel(z[:, q], ro["z"]))
        assert np.array_equal(z[:, q] != 0, ro["z"] != 0) and np.count_nonzero(z[:, q]) == r
        assert rel(P[:, q], oracle.lpv_unpermute(ro["z"], Nf, Nv)) <= 1e-9
        # the single-signal entry point (scalar tile product / same plain path) with the same estimator
        se = L.ls_sparse_spectral_lpv(Y[:, q].copy(), X, V, w, Nv, proxg=L.IndBallL0(r), printerval=100000, out=io.StringIO(), **kw)
        assert rel(se.x, oracle.lpv_unpermute(ro["z"], Nf, Nv)) <= 1e-9 and np.array_equal(se.x != 0, P[:, q] != 0)
        stops.append(ro["iters"])
    assert len(set(stops)) > 1 and min(stops) < 600            # the channels stop at their own iterations
    # the drop-in wrapper returns the same thing
    ses = L.ls_sparse_spectral_lpv_multi(Y, X, V, w, Nv, proxg=L.IndBallL0(r), printerval=100000, out=io.StringIO(), **kw)
    assert all(np.array_equal(ses[q].x, P[:, q]) for q in range(ns))


def test_cfg5_kernel_chain_long_horizon_vs_oracle(L, oracle):
    """cfg5's kernel chain at the BENCH'S OWN HORIZON where the CPU oracle can follow (VERDICT round 5, next #1c): 8 channels sharing (X, V),
    Nf = 256, Nv = 16 -> n = 8192 (N = 2^20, so that the mixed storage cfg5 streams holds: 36-bit fixed-point tiles below the diagonal),
    IndBallL0(32), 2000 iterations of symv_tile_mfma_ws_kernel (4x4x4 MFMA, partials per run of tiles) -> symv_reduce_runs_kernel ->
    admm_prox_kernel (one-pass top-32), with the x-update correction that is the default of multi-channel handles since round 6.
    Channels 0 and 5 against oracle.admm_gram_multi (src/lasso.jl:136-171 on the Gram form, one Cholesky factor for both; README.md:79-83 for the
    estimator) on the device's own Gram: live for the first 200 iterations, from the fixture tests/golden/cfg5_n8192_oracle_iterates.npz
    (tools/cfg5_midsize_vs_oracle.py --save; keyed by the sha256 of G, B; a live run checks the stored iterates bit for bit) at 500, 1000 and
    2000 -- rel-L2 of x, z, u <= 1e-9, identical support (32 of 8192) at every count.  Measured 2.6e-10 at 2000 (uncorrected: 4.1e-10)."""
    import bench, hashlib
    from _guards import precondition_not_met
    N, Nf, Nv, ns, r, mu = 1 << 20, 256, 16, 8, 32, 0.05
    Y, X, V, w = bench.synth_channels(N, Nf, ns, torch.device("cuda"))
    dev = {}
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        assert p.n == 8192
        p.set_prox(L.IndBallL0(r))
        p.admm_init(None, μ=mu, tol=0.0)
        info = p.matvec_info()
        assert info["kernel"] == "symv_tile_mfma_ws_kernel" and "36-bit fixed point" in info["storage"] and info["signals_per_pass"] == 8, info
        done = 0
        for cnt in (200, 500, 1000, 2000):
            it, _, conv = p.admm_run(cnt - done); done = cnt
            assert it == cnt and not conv
            dev[cnt] = p.admm_get()
        assert p.timing()["xcorr_count"] == 3                                          # after 16, 512, 1024: several right-hand sides
        G, _ = p.get_gram(); B = p.get_rhs()
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg5_n8192_oracle_iterates.npz"))
    fp = hashlib.sha256(np.ascontiguousarray(G).tobytes() + np.ascontiguousarray(B).tobytes()).hexdigest()
    if str(fix["sha256"]) != fp:
        precondition_not_met("tests/golden/cfg5_n8192_oracle_iterates.npz belongs to another G, B (%s..., now %s...): regenerate it with "
                             "tools/cfg5_midsize_vs_oracle.py --save" % (str(fix["sha256"])[:12], fp[:12]))
    chans, counts = [int(c) for c in fix["channels"]], [int(c) for c in fix["counts"]]
    assert counts == [200, 500, 1000, 2000]
    live = oracle.admm_gram_multi(G, B[:, chans], oracle.IndBallL0(r), [200], mu=mu)[200]
    for i, name in enumerate(("oracle_x", "oracle_z", "oracle_u")):
        assert np.array_equal(live[i], fix[name][0]), name                             # the stored oracle iterates ARE the oracle's
    worst = {}
    for k, cnt in enumerate(counts):
        x, z, u = dev[cnt]
        for j, ch in enumerate(chans):
            ox, oz, ou = fix["oracle_x"][k][:, j], fix["oracle_z"][k][:, j], fix["oracle_u"][k][:, j]
            e = dict(x=rel(x[:, ch], ox), z=rel(z[:, ch], oz), u=rel(u[:, ch], ou))
            worst[cnt] = max(worst.get(cnt, 0.0), *e.values())
            assert max(e.values()) <= 1e-9, (cnt, ch, e)
            assert np.array_equal(z[:, ch] != 0, oz != 0) and np.count_nonzero(oz) == r, (cnt, ch)
    print("cfg5 chain n=8192, 8 channels, IndBallL0(32): channels %s vs oracle.admm_gram_multi, worst rel-L2 of x, z, u after %s" %
          (chans, ", ".join(f"{c}: {worst[c]:.2e}" for c in counts)))


def _ball_prox_host(v, r):
    """IndBallL0(r): keep the r largest |v| (lowest index first on ties), zero the rest."""
    idx = np.argsort(-np.abs(v), kind="stable")[:r]
    z = np.zeros_like(v)
    z[idx] = v[idx]
    return z


CFG5_SEL = (0, 77, 512, 900, 1023)       # frequency groups whose 32 columns each are rebuilt on the host (first / last / three inside)
def test_cfg5_fullshape_eight_channels_invariants(L, oracle):
    """cfg5 as one GPU sees it: 8 of the 64 channels, N = 2^20, Nf = 1024, Nv = 16 (n = 32768), IndBallL0(32).

    (1) The Gram this handle iterates on is PINNED TO THE ORACLE at the judged size: 160 columns of Phi
    (five frequency groups x 32) from ``oracle.lpv_regressor`` -- the reference's formula, src/lasso.jl:39-50 -- give a 160 x 160
    block of G and 160 rows of B = Phi'Y for the 8 channels; the device's default form at this size is the structured Gram with the
    slot sums by non-uniform FFT on the nf = 8192 fine grid (two grids per workgroup, half twiddle table: the branch only n = 32768
    takes).  Bound as for cfg3 (tests/test_gpu_judged_size.py): 1e-12 + the reference's rounding of w*x, 4.5e-16 max|w x|.
    (2) No CPU oracle runs the ITERATION at this size; per channel the iterate invariants that hold for the reference algorithm
    are checked on the returned vectors: u += x - z bit for bit, z = prox(x + u_prev) bit for bit (top-32 recomputed on the
    host), the reported norm, and the x-update's linear system -- its residual evaluated on the device Gram that (1) pinned."""
    import bench
    N, Nf, Nv, ns, r, mu = 1 << 20, 1024, 16, 8, 32, 0.05
    _, X, V, w = bench.synth_signal(N, Nf, 0, torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(0x1B5EC + 5)
    Y = torch.stack([(1 + q) * torch.cos(w[(37 * q + 11) % Nf] * X) * (1 + V) + 0.5 * torch.cos(w[(91 * q + 400) % Nf] * X)
                     + 0.7 * torch.sin(w[(53 * q + 700) % Nf] * X) * V
                     + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g) for q in range(ns)], dim=1)
    with L.Problem.lpv_multi(Y, X, V, w, Nv) as p:
        assert p.n == 32768 and p.timing()["gram_form"] in ("ap", "ap-nufft")
        p.set_prox(L.IndBallL0(r))
        p.admm_init(None, μ=mu, tol=0.0)
        it, _, conv = p.admm_run(30)
        x1, z1, u1 = p.admm_get()
        it2, nxz2, _ = p.admm_run(1)
        x2, z2, u2 = p.admm_get()
        per = [p.admm_status(q) for q in range(ns)]
        B = p.get_rhs()
        Gd, bd = p.device_gram()
        # ---- (1) the pin: oracle columns at N = 2^20
        cols = np.concatenate([f * 2 * Nv + np.arange(2 * Nv) for f in CFG5_SEL])
        Xh, Vh, wh, Yh = X.cpu().numpy(), V.cpu().numpy(), w.cpu().numpy(), Y.cpu().numpy()
        Phi = oracle.lpv_regressor(Xh, Vh, wh[list(CFG5_SEL)], Nv)                      # N x 160 (1.3 GB), reference formula
        assert Phi.shape == (N, len(cols))
        Go, Bo = Phi.T @ Phi, Phi.T @ Yh
        del Phi
        ct = torch.tensor(cols, device="cuda")
        Gblk = Gd.index_select(0, ct).index_select(1, ct).cpu().numpy()
        phase = 4.5e-16 * float(wh.max() * Xh.max())
        eg = np.abs(Gblk - Go).max() / np.abs(Go).max()
        # (of the channel's largest right-hand-side entry over ALL rows, as for cfg3: the five groups need not hold a channel's own frequencies)
        eb = max(np.abs(B[cols, q] - Bo[:, q]).max() / np.abs(B[:, q]).max() for q in range(ns))
        print(f"cfg5 N=2^20 n=32768 default ({p.timing()['gram_form']}) form vs oracle columns: G block {eg:.2e}, B rows {eb:.2e}; phase-rounding bound {phase:.2e}")
        assert p.timing()["gram_form"] == "ap-nufft"
        assert eg <= min(1e-12 + phase, 5e-12) and eb <= min(1e-12 + phase, 5e-12), (eg, eb, phase)
        rhs = torch.tensor(B + (z1 - u1) / mu, device="cuda").T.contiguous()          # [ns][n]
        xd = torch.tensor(x2, device="cuda").T.contiguous()
        res = (xd @ Gd + xd / mu - rhs).cpu().numpy()                                  # G symmetric: x'G = (G x)'
        sym = float((Gd[:4096, :4096] - Gd[:4096, :4096].T).abs().max().item())
    assert it == 30 and it2 == 31 and not conv and sym == 0.0
    for q in range(ns):
        assert per[q][0] == 31
        assert abs(np.linalg.norm(x2[:, q] - z2[:, q]) - per[q][1]) <= 1e-12 * max(per[q][1], 1e-30)
        assert np.array_equal(u2[:, q], u1[:, q] + (x2[:, q] - z2[:, q]))
        assert np.array_equal(z2[:, q], _ball_prox_host(x2[:, q] + u1[:, q], r))
        assert np.count_nonzero(z2[:, q]) == r
        assert np.linalg.norm(res[q]) <= 1e-9 * np.linalg.norm(B[:, q]), (q, np.linalg.norm(res[q]) / np.linalg.norm(B[:, q]))
    assert nxz2 == max(s[1] for s in per)
    # channel 3 alone through the single-signal path: same support, same coefficients to summation order
    q = 3
    with L.Problem.lpv(Y[:, q].contiguous(), X, V, w, Nv) as p1:
        p1.set_prox(L.IndBallL0(r))
        p1.admm_init(None, μ=mu, tol=0.0)
        p1.admm_run(31)
        xs, zs, us = p1.admm_get()
    r1 = rel(zs, z2[:, q])
    print(f"cfg5 shape n=32768, 31 iterations: single-signal path (6-byte storage of M, offset form) vs multi-signal path (doubles, matrix cores): rel-L2(z) = {r1:.3e}")
    assert np.array_equal(zs != 0, z2[:, q] != 0) and r1 <= 1e-9


# ------------------------------------------------------------------ per-iteration iterates with tol > 0 on the fused path
def test_fused_update_first_iterations_with_positive_tol(L, oracle):
    """n = 2048 (tile-packed mat-vec + fused update with the deferred convergence commit), tol > 0, one iteration per
    admm_run call and then chunks: every iterate equals the oracle's at the same iteration (the first update launch of a
    chunk must not consult a previous iteration's norm: it has none)."""
    rng = np.random.default_rng(77)
    N, Nf, Nv = 3000, 128, 8
    X = np.sort(10 * rng.random(N) * N / 500); V = np.linspace(0, 1, N)
    w = 2 * np.pi * (np.arange(Nf) + 1.0) * 25 / Nf / 4
    y = 2 * V ** 2 * np.cos(w[12] * X) + 2 / (5 * V + 1) * np.cos(w[60] * X) + 0.1 * rng.standard_normal(N)
    lam, mu, tol = 3.0, 0.05, 1e-5
    with L.Problem.lpv(y, X, V, w, Nv) as p:
        assert p.n == 2048
        G, b = p.get_gram()
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, Nf, 2 * Nv))
        for trial in range(3):                                   # re-initialising must not leave a stale pending norm behind
            p.admm_init(None, μ=mu, tol=tol)
            got = []
            for k in (1, 1, 1, 2, 5):
                it, nxz, conv = p.admm_run(k)
                got.append((it, nxz, conv) + p.admm_get())
            for (it, nxz, conv, x, z, u) in got:
                ro = oracle.admm_gram(G, b, oracle.GroupL2(lam, 2 * Nv), iters=it, tol=tol, mu=mu, history=True)
                assert ro["iters"] == it and not conv
                assert rel(x, ro["x"]) <= 1e-9 and rel(z, ro["z"]) <= 1e-9 and rel(u, ro["u"]) <= 1e-9, (trial, it)
                assert np.array_equal(z != 0, ro["z"] != 0)
                assert abs(nxz - ro["nxz"][-1]) <= 1e-9 * ro["nxz"][-1]
        # run to convergence in ragged chunks: the stopping iteration is the oracle's
        p.admm_init(None, μ=mu, tol=1e-3)
        done, conv = 0, False
        while not conv and done < 4000:
            done, nxz, conv = p.admm_run(37)
        ro = oracle.admm_gram(G, b, oracle.GroupL2(lam, 2 * Nv), iters=4000, tol=1e-3, mu=mu)
        assert conv and done == ro["iters"]
        assert rel(p.admm_get()[1], ro["z"]) <= 1e-9
