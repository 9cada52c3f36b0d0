"""The judged sizes held to the CPU ORACLE, not to another HIP path.  GPU only.

  cfg3 (N = 2^20, Nf = 512, Nv = 8, n = 8192)
    * Gram: 80 columns of Phi (five frequency groups x 16) built on the host by ``oracle.lpv_regressor`` -- the reference's own
      formula, src/lasso.jl:35-50, at |w x| up to 3.3e6 rad -- pin an 80 x 80 block of G and 80 entries of b of the device's
      default (structured / NUFFT) form and of the dense MFMA form.
    * Iteration: G, b read back at n = 8192 and ``oracle.admm_gram`` (group prox, lam = 5, mu = 0.05) against
      ``admm_iter_mixed_kernel`` (name asserted): rel-L2 of x, z, u <= 1e-9 with identical support after 200, 500, 1000 and the
      bench's own 2000 iterations (SURVEY 8(d)'s tolerance, met at every count since round 5's x-update correction), and against
      the EXACT iterates -- the same algorithm in extended precision, a committed fixture -- to 2e-10 at 2000.
  cfg4 (1024 windows x 2^16, Nf = 256 with the zero frequency, L1, mu = 1e-4), 2000 iterations per window as bench.py runs it
    * the DEFAULT execution plan (cache-sized chunks, two parts in flight, 32-bit reads + stale nibble product) against the uncut
      single launch sequence bit for bit, and the raw state x, z, u of four spot windows against ``oracle.admm_quadratic`` /
      ``oracle.admm_gram`` / the oracle's whole host pipeline.

Tolerances.  Dense form vs oracle: 1e-12 of max|G| (same rounded phases fl(w x) as the reference, only the summation order differs).
Structured form vs oracle: the reference rounds the phase w*x to a double BEFORE cos/sin (src/lasso.jl:42), an error of up to
2^-53 |w x| rad per sample that the structured form (exact progression phases) does not make; the bound below is that term,
4.5e-16 * max|w x| (= 1.5e-9 at this size) -- and the same block against a host Phi with the phase product carried in long double
(no rounding of w*x) is held to 1e-12, which shows the difference to the oracle IS the reference's phase rounding."""
import numpy as np
import pytest
import torch

from _guards import precondition_not_met

pytestmark = pytest.mark.gpu

SEL = (0, 40, 204, 409, 511)          # first / last group, the three true frequencies of the generator


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def cfg3():
    import bench
    y, X, V, w = bench.synth_signal(1 << 20, 512, 0, torch.device("cuda"))
    return dict(y=y, X=X, V=V, w=w, yh=y.cpu().numpy(), Xh=X.cpu().numpy(), Vh=V.cpu().numpy(), wh=w.cpu().numpy())


def _cols(Nv):
    return np.concatenate([f * 2 * Nv + np.arange(2 * Nv) for f in SEL])


def _phi_long_double_phase(oracle, Xh, Vh, wsel, Nv):
    """The same columns with the phase w*x formed and reduced in long double (x87 80-bit: 64-bit mantissa), then cos / sin in double:
    the reference's formula without its rounding of the product."""
    K = oracle.basis_activation(Vh, Nv, True, False)                       # N x Nv, src/utilities.jl:23-36
    N = len(Xh)
    Phi = np.empty((N, len(wsel) * 2 * Nv), order="F")
    xl = Xh.astype(np.longdouble)
    twopi = 2 * np.longdouble(np.pi) if np.finfo(np.longdouble).nmant < 63 else np.longdouble("6.283185307179586476925286766559005768")
    for k, wf in enumerate(wsel):
        ph = np.longdouble(wf) * xl
        ph = ph - np.floor(ph / twopi) * twopi
        c, s = np.cos(ph.astype(np.float64)), np.sin(ph.astype(np.float64))
        Phi[:, k * 2 * Nv:k * 2 * Nv + Nv] = c[:, None] * K
        Phi[:, k * 2 * Nv + Nv:(k + 1) * 2 * Nv] = -s[:, None] * K
    return Phi


def test_cfg3_gram_block_against_oracle_columns(L, oracle, cfg3):
    """N = 2^20: G[cols, cols] and b[cols] of the device against Phi_cols' Phi_cols, Phi_cols' y with Phi_cols from the oracle."""
    Nv = 8
    c = cfg3
    cols = _cols(Nv)
    Phi = oracle.lpv_regressor(c["Xh"], c["Vh"], c["wh"][list(SEL)], Nv)                    # N x 80 (671 MB), reference formula
    assert Phi.shape == (1 << 20, len(cols))
    Go, bo = Phi.T @ Phi, Phi.T @ c["yh"]
    gs, bs = np.abs(Go).max(), np.abs(bo).max()
    # dense MFMA form (general w): same rounded phases as the oracle
    with L.default_options(gram_form="krs"):
        with L.Problem.lpv(c["y"], c["X"], c["V"], c["w"], Nv) as p:
            assert p.timing()["gram_form"] == "krs" and p.n == 8192
            G, b = p.get_gram()
    eg, eb = np.abs(G[np.ix_(cols, cols)] - Go).max() / gs, np.abs(b[cols] - bo).max() / bs
    print(f"cfg3 N=2^20 dense form vs oracle columns: G block {eg:.2e}, b {eb:.2e} (of max)")
    assert eg <= 1e-12 and eb <= 1e-12, (eg, eb)
    # default form (structured, slot sums by non-uniform FFT)
    with L.Problem.lpv(c["y"], c["X"], c["V"], c["w"], Nv) as p:
        assert p.timing()["gram_form"] == "ap-nufft"
        Gd, bd = p.get_gram()
    phase = 4.5e-16 * float(c["wh"].max() * c["Xh"].max())                                  # the reference's rounding of w*x
    eg, eb = np.abs(Gd[np.ix_(cols, cols)] - Go).max() / gs, np.abs(bd[cols] - bo).max() / bs
    print(f"cfg3 N=2^20 default (NUFFT) form vs oracle columns: G block {eg:.2e}, b {eb:.2e}; phase-rounding bound {phase:.2e}")
    assert eg <= min(1e-12 + phase, 5e-12) and eb <= min(1e-12 + phase, 5e-12), (eg, eb, phase)   # measured 5.9e-13 / 2.3e-13: random signs, far below the worst case
    del Phi
    if np.finfo(np.longdouble).nmant >= 63:
        Pl = _phi_long_double_phase(oracle, c["Xh"], c["Vh"], c["wh"][list(SEL)], Nv)
        Gl, bl = Pl.T @ Pl, Pl.T @ c["yh"]
        eg, eb = np.abs(Gd[np.ix_(cols, cols)] - Gl).max() / gs, np.abs(bd[cols] - bl).max() / bs
        print(f"cfg3 N=2^20 default (NUFFT) form vs host columns with unrounded phases: G block {eg:.2e}, b {eb:.2e}")
        assert eg <= 1e-12 and eb <= 1e-12, (eg, eb)
    # the whole matrices: the two device forms differ by the same phase term, nothing else (full 8192 x 8192)
    assert np.abs(Gd - G).max() / gs <= 1e-12 + phase


# ---- cfg3's iteration at the bench's own count.  Two references, both CPU restatements of src/lasso.jl:136-171 on the Gram form:
#   oracle.admm_gram        f64, Cholesky x-update -- run here at 200 iterations; its iterates after 200 / 500 / 1000 / 2000 iterations on
#                           THIS G, b are also in the fixture (they are a deterministic function of G, b: the test checks the 200-iteration
#                           run against them bit for bit), so the 2000-iteration comparison costs no three CPU-minutes per test run;
#   oracle.admm_gram_ld     the same algorithm in x87 extended precision (64-bit mantissa): the ADJUDICATOR between two f64 paths.
# Fixture: tests/golden/cfg3_extended_precision_iterates.npz (tools/cfg3_vs_oracle.py --longdouble --save, then --reuse-ld --save: 9 + 3
# CPU-minutes), keyed by the sha256 of G, b -- the device Gram is bit-reproducible (fixed-point accumulation, fixed summation orders).
# Measured (round 6: offset vector refined at lpvs_admm_init, corrections after 16, 128, 256, 512, 1024; max over x, z, u of rel-L2):
#   iterations                        200        500        1000       2000
#   device  vs exact                  1.2e-10    2.4e-10    1.9e-10    9.7e-11      (round 5's schedule 16, 512, 1024: 1.9e-10 4.8e-10 2.6e-10 1.2e-10;
#                                                                                    round 4, uncorrected: 2.4e-10 5.4e-10 8.8e-10 1.22e-9)
#   oracle  vs exact (z)              2.0e-10    3.7e-10    5.6e-10    7.2e-10
#   device  vs oracle                 2.8e-10    5.1e-10    7.1e-10    8.0e-10      (round 4: 4.7e-10 9.4e-10 1.5e-9 1.9e-9)
# SURVEY 8(d)'s 1e-9 against the f64 oracle holds at every count, and what is left of it is the ORACLE's own distance to the exact
# iterates (its Cholesky solves commit the same kind of systematic error the device's explicit inverse did: DESIGN.md section 6.1).
CFG3_EXACT_BOUND = {200: 2.5e-10, 500: 4e-10, 1000: 3.5e-10, 2000: 2e-10}   # device vs the extended-precision iterate (measured x 1.6 .. 2.1)
CFG3_ORACLE_BOUND = 1e-9                                                  # device vs the f64 oracle, every count
def test_cfg3_one_launch_iteration_against_oracle_at_n8192(L, oracle, cfg3):
    """n = 8192 (64 row blocks, 2080 tiles, float-head diagonal tiles at their real scale): the benchmarked kernel against
    oracle.admm_gram (Cholesky x-update, src/lasso.jl:136-171 on the Gram form of :51) on the Gram read back -- live at 200 iterations
    (1e-9, identical support, same norm), from the fixture at 500 / 1000 and the bench's own 2000 -- and against the extended-precision
    iterates of the same algorithm (CFG3_EXACT_BOUND).  lam = 5 leaves every group active from ~500 iterations on at this N: the
    support comparison is only non-trivial in the 200-iteration leg."""
    import hashlib, os
    Nv, lam, mu = 8, 5.0, 0.05
    c = cfg3
    dev = {}
    with L.Problem.lpv(c["y"], c["X"], c["V"], c["w"], Nv) as p:
        p.set_prox(L.SlicedSeparableSum.frequency_groups(lam, 512, 2 * Nv))
        p.admm_init(None, μ=mu, tol=0.0)
        info = p.matvec_info()
        assert info["kernel"] == "admm_iter_mixed_kernel" and info["one_launch_iteration"], info
        # the default: 32 of the fixed-point tiles' 36 bits read per iteration (1992 tiles x 66048 B + 88 float-head tiles x 98304 B = 140.2 MB),
        # the 4-bit planes through the stale nibble product (103 refreshes in 2000 iterations)
        assert "32-bit fixed point reads" in info["storage"] and p.time_matvec(10)[1] == 1992 * 66048 + 88 * 98304, info
        done = 0
        for cnt in (200, 500, 1000, 2000):
            it, nxz, conv = p.admm_run(cnt - done)
            done = cnt
            assert it == cnt and not conv
            dev[cnt] = p.admm_get() + (nxz,)
        G, b = p.get_gram()
    x, z, u, nxz = dev[200]
    ro = oracle.admm_gram(G, b, oracle.GroupL2(lam, 2 * Nv), iters=200, tol=0.0, mu=mu, history=True)
    errs = {k: rel(v, ro[k]) for k, v in (("x", x), ("z", z), ("u", u))}
    nz = np.count_nonzero(ro["z"])
    print(f"cfg3 n=8192, 200 iterations, admm_iter_mixed_kernel vs oracle.admm_gram: x {errs['x']:.2e} z {errs['z']:.2e} u {errs['u']:.2e}; nnz {nz}")
    assert max(errs.values()) <= CFG3_ORACLE_BOUND, errs
    assert np.array_equal(z != 0, ro["z"] != 0) and 0 < nz < z.size                  # (a support that could differ)
    assert abs(nxz - ro["nxz"][-1]) <= 1e-7 * ro["nxz"][-1]
    # ---- the fixture: exact (extended-precision) and f64-oracle iterates of the same G, b
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_extended_precision_iterates.npz"))
    fp = hashlib.sha256(np.ascontiguousarray(G).tobytes() + np.ascontiguousarray(b).tobytes()).hexdigest()
    if str(fix["sha256"]) != fp:
        precondition_not_met("tests/golden/cfg3_extended_precision_iterates.npz belongs to another G, b (%s..., now %s...): the Gram's bits changed -- "
                    "regenerate it with tools/cfg3_vs_oracle.py --longdouble --save, then --reuse-ld --save" % (str(fix["sha256"])[:12], fp[:12]))
    assert np.array_equal(ro["z"], fix["oracle_z"][0]) and np.array_equal(ro["x"], fix["oracle_x"][0])   # the stored oracle iterates ARE the oracle's
    for k, cnt in enumerate(int(q) for q in fix["counts"]):
        x, z, u, _ = dev[cnt]
        e = {"x": rel(x, fix["x"][k]), "z": rel(z, fix["z"][k]), "u": rel(u, fix["u"][k])}
        eo = {"x": rel(x, fix["oracle_x"][k]), "z": rel(z, fix["oracle_z"][k]), "u": rel(u, fix["oracle_u"][k])}
        print(f"cfg3 n=8192, {cnt} iterations, admm_iter_mixed_kernel vs the exact iterate: x {e['x']:.2e} z {e['z']:.2e} u {e['u']:.2e} | "
              f"vs the f64 oracle: x {eo['x']:.2e} z {eo['z']:.2e} u {eo['u']:.2e} | oracle vs exact z {rel(fix['oracle_z'][k], fix['z'][k]):.2e}")
        assert max(e.values()) <= CFG3_EXACT_BOUND[cnt], (cnt, e)
        assert max(eo.values()) <= CFG3_ORACLE_BOUND, (cnt, eo)
        assert np.array_equal(z != 0, fix["z"][k] != 0) and np.array_equal(z != 0, fix["oracle_z"][k] != 0)


def test_cfg3_36_bit_reads_by_name_and_without_the_nibble_refresh(L, cfg3, monkeypatch):
    """The default handle of cfg3 READS 32 of the 36 bits of its fixed-point tiles (140.2 MB per iteration) and carries the product of the
    4-bit planes with a right-hand side at most 32 iterations old in the offset vector (LPVS_STORAGE_MIXED32, the stale nibble product:
    test_cfg3_one_launch_iteration_against_oracle_at_n8192 holds it to the exact iterates).  Here the two things beside it:
    (a) storage="mixed" BY NAME: all 36 bits every iteration (156.5 MB), the same bounds;
    (b) the reason the refresh exists: 32-bit reads WITHOUT it (LPVS_FIX_BITS=32: the planes are packed as zeros) leave x and z where they
        are -- the x-update correction removes the truncation's systematic part -- but the DUAL variable integrates the rest
        (measured 2.2e-9 / 6.6e-9 / 2.5e-9 / 1.0e-9 from the exact iterates; with the refresh 1.8e-10 / 5.0e-10 / 2.5e-10 / 9.8e-11)."""
    import hashlib, os
    c = cfg3
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_extended_precision_iterates.npz"))
    counts = [int(q) for q in fix["counts"]]
    def leg(storage):
        with L.Problem.lpv(c["y"], c["X"], c["V"], c["w"], 8) as p:
            G, b = p.get_gram()
            if hashlib.sha256(np.ascontiguousarray(G).tobytes() + np.ascontiguousarray(b).tobytes()).hexdigest() != str(fix["sha256"]):
                precondition_not_met("the fixture belongs to another G, b (see test_cfg3_one_launch_iteration_against_oracle_at_n8192)")
            del G
            if storage:
                p.set_option("storage", storage)
            p.set_prox(L.SlicedSeparableSum.frequency_groups(5.0, 512, 16))
            p.admm_init(None, μ=0.05, tol=0.0)
            info = p.matvec_info()
            assert info["kernel"] == "admm_iter_mixed_kernel", info
            nbytes = p.time_matvec(20)[1]
            out, done = [], 0
            for k, cnt in enumerate(counts):
                p.admm_run(cnt - done); done = cnt
                x, z, u = p.admm_get()
                out.append((max(rel(x, fix["x"][k]), rel(z, fix["z"][k])), max(rel(x, fix["oracle_x"][k]), rel(z, fix["oracle_z"][k])),
                            rel(u, fix["u"][k]), rel(u, fix["oracle_u"][k]), np.array_equal(z != 0, fix["z"][k] != 0)))
            return info, nbytes, out
    # (a) 1992 fixed-point tiles x 74240 B + 88 float-head tiles x 98304 B
    info, nbytes, out = leg("mixed")
    assert nbytes == 156536832 == 1992 * 74240 + 88 * 98304 and "32-bit fixed point reads" not in info["storage"], (nbytes, info)
    for cnt, (ex, eo, eu, euo, same) in zip(counts, out):
        print(f"cfg3, 36-bit reads, {cnt} iterations: x, z vs exact {ex:.2e}, vs the f64 oracle {eo:.2e}; u vs exact {eu:.2e}, vs the oracle {euo:.2e}")
        assert max(ex, eu) <= CFG3_EXACT_BOUND[cnt] and max(eo, euo) <= CFG3_ORACLE_BOUND and same, (cnt, ex, eo, eu, euo)
    # (b) plain truncation: 8192 bytes of nibbles less for each of the 1992 fixed-point tiles, no refresh
    monkeypatch.setenv("LPVS_FIX_BITS", "32"); monkeypatch.setenv("LPVS_NIB_PERIOD", "0")
    info, nbytes, out = leg(None)
    assert nbytes == 156536832 - 8192 * 1992 == 1992 * 66048 + 88 * 98304 and "32-bit fixed point reads" in info["storage"], (nbytes, info)
    for cnt, (ex, eo, eu, euo, same) in zip(counts, out):
        print(f"cfg3, 32-bit tiles without the nibble refresh, {cnt} iterations: x, z vs exact {ex:.2e}, vs the f64 oracle {eo:.2e}; u vs exact {eu:.2e}")
        assert ex <= CFG3_EXACT_BOUND[cnt] and eo <= CFG3_ORACLE_BOUND and same, (cnt, ex, eo)
        assert eu <= 1.5e-8, (cnt, eu)
    assert max(o[2] for o in out) > 1e-9                                   # (the day this fails the refresh is no longer needed)


def test_cfg4_default_plan_fullsize_against_uncut_and_oracle(L, oracle, monkeypatch):
    """1024 windows x 2^16 under the engine's DEFAULT plan (what bench.py's cfg4 record times: cache-sized chunks, two parts in flight, 32 of
    the fixed-point tiles' 36 bits read + the stale nibble product with its period ramped to 32 from launch 256 on) AT THE BENCH'S OWN
    2000 ITERATIONS PER WINDOW (src/lsfft.jl:112-126 -> src/lasso.jl:105-126; VERDICT round 5, next #1a):
      * bit for bit against one uncut launch sequence;
      * windows 0, 341, 342 (a chunk boundary of the default plan) and 1023: the raw state x, z AND u (lpvs_windows_estimate_state_f64)
        against the oracle's ADMM on the window's device Gram -- oracle.admm_quadratic (Quadratic(Q, +q) with its CG x-update, as written)
        and oracle.admm_gram (the same iteration with the exact Cholesky x-update) -- rel-L2 <= 1e-9, identical support; and the packed
        coefficients against the oracle's whole host pipeline (1e-8: its Gram is formed on the host)."""
    import bench
    from lpvspectral_jl_amd import _lib, api
    n, nwin, Nf, iters = 1 << 16, 1024, 256, 2000
    y, t, f = bench.synth_windows(nwin, n, Nf, torch.device("cuda"))
    eng = dict(estimator=_lib.EST_SPARSE, lam=0.0, prox=(_lib.PROX_L1, 0.2, 0), μ=1e-4, tol=0.0, iters=iters, sign=_lib.LINEAR_QUADRATIC_AS_WRITTEN)
    for v in ("LPVS_WINDOW_CHUNK_MB", "LPVS_WINDOWS_IN_FLIGHT", "LPVS_ITERATION", "LPVS_NT_LOADS", "LPVS_NIB_PERIOD", "LPVS_NIB_RAMP", "LPVS_NIB_FUSED", "LPVS_M_STORAGE"):
        monkeypatch.delenv(v, raising=False)
    x1, its1 = api.windows_estimate([y], t, f, n, 0, None, eng)
    tm = api.windowpsd_last_timing()
    assert tm["one_launch_iteration"] and tm["reads_32_bits"] and tm["windows"] == nwin, tm
    xs, zs, us, its_s = api.windows_estimate_state([y], t, f, n, 0, None, eng)          # the default plan again, raw state out
    monkeypatch.setenv("LPVS_WINDOW_CHUNK_MB", "0"); monkeypatch.setenv("LPVS_WINDOWS_IN_FLIGHT", "1")
    x0, its0 = api.windows_estimate([y], t, f, n, 0, None, eng)
    monkeypatch.delenv("LPVS_WINDOW_CHUNK_MB"); monkeypatch.delenv("LPVS_WINDOWS_IN_FLIGHT")
    assert x1.shape == (1, nwin, Nf) and np.all(its1 == iters)
    assert np.array_equal(x1, x0) and np.array_equal(its1, its0) and np.array_equal(its1, its_s)
    assert zs.shape == (1, nwin, 2 * Nf - 1)
    assert all(np.array_equal(oracle.fourier2complex(zs[0, i], 1), x1[0, i]) for i in range(nwin))    # the state call IS the same run
    S = (np.abs(x1[0]) ** 2).sum(0)
    assert int(np.argmax(S)) == 33
    yh, th = y.cpu().numpy(), t.cpu().numpy()
    W = np.ones(n)
    worst = {"cg": 0.0, "chol": 0.0}
    for i in (0, 341, 342, 1023):
        yi, ti = yh[i * n:(i + 1) * n], th[i * n:(i + 1) * n]
        with L.Problem.fourier(yi, ti, f, W) as p:
            Q, q = p.get_gram()
        dev = {"x": xs[0, i], "z": zs[0, i], "u": us[0, i]}
        assert np.count_nonzero(dev["u"]) > 0
        for name, ro in (("cg", oracle.admm_quadratic(Q, q, oracle.NormL1(0.2), iters=iters, tol=0.0, mu=1e-4)),
                         ("chol", oracle.admm_gram(Q, -q, oracle.NormL1(0.2), iters=iters, tol=0.0, mu=1e-4))):
            assert ro["iters"] == iters
            e = {k: rel(dev[k], ro[k]) for k in ("x", "z", "u")}
            worst[name] = max(worst[name], *e.values())
            assert max(e.values()) <= 1e-9, (i, name, e)
            assert np.array_equal(dev["z"] != 0, ro["z"] != 0) and 0 < np.count_nonzero(ro["z"]) < ro["z"].size, (i, name)
        xo = oracle.ls_sparse_spectral(yi, ti, f, W, proxg=oracle.NormL1(0.2), iters=iters, tol=0.0, mu=1e-4)[0]   # the whole host pipeline
        assert rel(x1[0, i], xo) <= 1e-8, (i, rel(x1[0, i], xo))
        assert np.array_equal(x1[0, i] != 0, xo != 0), i
    print(f"cfg4 1024 x 2^16, default plan, {iters} iterations per window: bit-identical to the uncut call; spot windows' x, z, u vs "
          f"oracle.admm_quadratic worst rel-L2 {worst['cg']:.2e}, vs oracle.admm_gram (Cholesky x-update) {worst['chol']:.2e}")
