"""GPU parity tests of the pose-graph path (BASELINE.json configs[4]) against the CPU oracle and golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(soslam):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    from stereo_orb_slam_amd import pg, synth
    return pg, synth, soslam


def _oracle_solve(oracle_lib, g, iters=10):
    L = oracle_lib.lib()
    o = oracle_lib.PgOptions()
    L.oracle_pg_options_default(C.byref(o))
    o.max_iterations = iters
    est = g.est.copy()
    s = oracle_lib.PgSummary()
    log = (oracle_lib.PgIteration * iters)()
    rc = L.oracle_pg_solve(len(est), len(g.e_from), est, g.fixed, g.e_from, g.e_to, np.ascontiguousarray(g.meas),
                           np.ascontiguousarray(g.info), C.byref(o), C.byref(s), C.cast(log, C.c_void_p))
    assert rc == 0
    return est, s, list(log)[: s.iterations]


def test_edge_blocks_match_autograd_golden(gpu, golden_dir):
    pg, synth, L = gpu
    g = np.load(os.path.join(golden_dir, "pg_edge_jacobian.npz"))
    n = len(g["e"])
    est = np.concatenate([g["xi"], g["xj"]])                       # vertex k and n + k form edge k
    ef, et = np.arange(n, dtype=np.uint32), np.arange(n, 2 * n, dtype=np.uint32)
    with pg.PoseGraph() as h:
        h.set_graph(est, np.zeros(2 * n, np.uint8), ef, et, g["z"], np.eye(6).reshape(36))
        out = h.debug_linearize(dense=False)
    np.testing.assert_allclose(out["e"], g["e"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(out["ji"], g["ji"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["jj"], g["jj"], rtol=1e-9, atol=1e-11)


def test_linearized_system_matches_oracle(gpu, oracle_lib):
    pg, synth, L = gpu
    g = synth.generate_pg(6)
    n6 = 6 * (len(g.est) - 1)
    H, b = np.zeros((n6, n6)), np.zeros(n6)
    chi = oracle_lib.lib().oracle_pg_linearize(len(g.est), len(g.e_from), np.ascontiguousarray(g.est), g.fixed, g.e_from, g.e_to,
                                               np.ascontiguousarray(g.meas), np.ascontiguousarray(g.info), 1.0, H, b)
    with pg.PoseGraph() as h:
        h.load(g)
        out = h.debug_linearize()
    assert out["chi2"] == pytest.approx(chi, rel=1e-11)
    np.testing.assert_allclose(out["H"], H, rtol=1e-9, atol=1e-11 * np.abs(H).max())
    np.testing.assert_allclose(out["b"], b, rtol=1e-9, atol=1e-11 * np.abs(b).max())


def _same_rotation(qa, qb, atol):
    sign = np.sign((qa * qb).sum(1))[:, None]
    np.testing.assert_allclose(qa * sign, qb, atol=atol)


def test_solve_small_graph_matches_oracle(gpu, oracle_lib):
    pg, synth, L = gpu
    g = synth.generate_pg(6)
    oest, osum, olog = _oracle_solve(oracle_lib, g)
    with pg.PoseGraph() as h:
        h.load(g)
        s = h.optimize()
        est, log = h.estimates(), h.iteration_log()
    assert s.initial_chi2 == pytest.approx(osum.initial_chi2, rel=1e-11)
    assert s.final_chi2 == pytest.approx(osum.final_chi2, rel=1e-5)
    assert s.iterations == osum.iterations and s.termination == osum.termination
    assert [e.trials for e in log] == [e.trials for e in olog]
    np.testing.assert_allclose([e.chi2 for e in log], [e.chi2 for e in olog], rtol=1e-5)
    np.testing.assert_allclose(est[:, :3], oest[:, :3], atol=1e-4)
    _same_rotation(est[:, 3:], oest[:, 3:], 1e-5)
    np.testing.assert_array_equal(est[0], g.est[0])               # vertex 0 fixed


def test_solve_reaches_scipy_minimum(gpu, golden_dir):
    pg, synth, L = gpu
    g = np.load(os.path.join(golden_dir, "pg_minimum_scipy.npz"))
    fixed = np.zeros(len(g["est0"]), np.uint8)
    fixed[0] = 1
    with pg.PoseGraph(pg.default_options(max_iterations=40)) as h:
        h.set_graph(g["est0"], fixed, g["e_from"], g["e_to"], g["meas"], g["info"])
        s = h.optimize()
    assert s.initial_chi2 == pytest.approx(float(g["chi2_0"]), rel=1e-10)
    assert s.final_chi2 == pytest.approx(float(g["chi2"]), rel=1e-6)


def test_config5_full_size_matches_oracle(gpu, oracle_lib):
    """BASELINE.json configs[4]: 5 000 SE(3) nodes / 20 000 edges, the reference's 10 Levenberg iterations
    (/root/reference/src/pose_graph_optimizer.cpp:69) against the oracle (whose linear solve is PCG at 1e-12): every
    iteration's chi2, lambda and trial count, poses to 1e-4 (the north_star's bar on pose parameters), monotone chi2."""
    pg, synth, L = gpu
    g = synth.generate_pg(5)
    assert (len(g.est), len(g.e_from)) == (5000, 20000)
    oest, osum, olog = _oracle_solve(oracle_lib, g, iters=10)
    with pg.PoseGraph(pg.default_options(max_iterations=10)) as h:
        h.load(g)
        s = h.optimize()
        est, log = h.estimates(), h.iteration_log()
    assert s.iterations == osum.iterations == 10
    assert s.final_chi2 == pytest.approx(osum.final_chi2, rel=1e-5)
    chis = [e.chi2 for e in log]
    assert all(b <= a * (1 + 1e-12) for a, b in zip([s.initial_chi2] + chis, chis))
    for a, b in zip(log, olog):
        assert a.chi2 == pytest.approx(b.chi2, rel=1e-5) and a.trials == b.trials and a.lam == pytest.approx(b.lam, rel=1e-4)
    np.testing.assert_allclose(est[:, :3], oest[:, :3], atol=1e-4)
    _same_rotation(est[:, 3:], oest[:, 3:], 1e-4)


def test_two_level_preconditioner_cuts_the_iterations_not_the_answer(gpu):
    """The reference solves the Levenberg system directly (LinearSolverEigen, /root/reference/src/pose_graph_optimizer.cpp:14-18);
    here PCG does, and its preconditioner must not show in the result: block-Jacobi alone and block-Jacobi plus the
    rigid-body coarse space (the default from 64 free vertices on) give the same ten iterations - chi2, lambda, trials, poses -
    while the coarse space takes a fraction of the PCG iterations (task 4: <= 100 per Levenberg step).  Also a
    chain with few loop closures, the shape the reference's own graphs have (odometry + loop edges)."""
    pg, synth, L = gpu
    g = synth.generate_pg(5)
    res = {}
    for pre in (1, 2):
        with pg.PoseGraph(pg.default_options(max_iterations=10, preconditioner=pre)) as h:
            h.load(g)
            s = h.optimize()
            res[pre] = (s, h.estimates(), h.iteration_log())
    (s1, e1, l1), (s2, e2, l2) = res[1], res[2]
    assert s1.iterations == s2.iterations == 10
    for a, b in zip(l1, l2):
        assert a.chi2 == pytest.approx(b.chi2, rel=1e-8) and a.trials == b.trials and a.lam == pytest.approx(b.lam, rel=1e-6)
    np.testing.assert_allclose(e1[:, :3], e2[:, :3], atol=1e-6)
    _same_rotation(e1[:, 3:], e2[:, 3:], 1e-6)
    per_solve = [it.linear_iterations / max(1, it.trials) for it in l2]
    assert max(per_solve) <= 100, per_solve
    assert s2.linear_iterations * 5 < s1.linear_iterations, (s1.linear_iterations, s2.linear_iterations)
    # a chain with a handful of closures: 600 vertices in one long row (no lattice), 12 loop edges
    c = synth.generate_pg(5, n_node=600, row_len=600, n_loop_max=12, min_gap=50, radius=80.0)
    out = {}
    for pre in (1, 2):
        with pg.PoseGraph(pg.default_options(max_iterations=10, preconditioner=pre)) as h:
            h.load(c)
            s = h.optimize()
            out[pre] = (s, h.estimates())
    assert out[1][0].final_chi2 == pytest.approx(out[2][0].final_chi2, rel=1e-7)
    np.testing.assert_allclose(out[1][1][:, :3], out[2][1][:, :3], atol=1e-5)
    # (on a one-dimensional chain the piecewise-rigid coarse space gains less than on the lattice: measured 5 080 -> 2 789)
    assert out[2][0].linear_iterations * 3 < out[1][0].linear_iterations * 2, (out[1][0].linear_iterations, out[2][0].linear_iterations)


def test_chain_with_loop_closures_takes_the_band_factor(gpu, oracle_lib):
    """The reference's own graphs: keyframes in order - an odometry chain - and a few loop-closure edges
    (/root/reference/src/pose_graph_optimizer.cpp:56-66).  Numbered breadth-first (Cuthill-McKee) such a graph is a band whatever
    the length of its loops: a loop's two arms are numbered alternately and the closure joins neighbours.  H then has an exact band
    factor (block cyclic reduction, as the BA path's reduced camera matrix) and a solve is ONE round, where the two-level PCG needs
    hundreds of iterations on a one-dimensional chain.  Same ten iterations as the oracle (direct sparse Cholesky): chi2, lambda,
    trials, poses - with three closures and with twelve."""
    pg, synth, L = gpu
    for loops in (3, 12):
        c = synth.generate_pg(5, n_node=2000, row_len=2000, n_loop_max=loops, min_gap=50, radius=80.0)
        assert len(c.e_from) == 1999 + loops
        oest, osum, olog = _oracle_solve(oracle_lib, c, iters=10)
        res = {}
        for pre in (0, 2):
            with pg.PoseGraph(pg.default_options(max_iterations=10, preconditioner=pre)) as h:
                h.load(c)
                s = h.optimize()
                res[pre] = (s, h.estimates(), h.iteration_log())
        s, est, log = res[0]
        assert s.iterations == osum.iterations
        for a, b in zip(log, olog):
            assert a.chi2 == pytest.approx(b.chi2, rel=1e-5) and a.trials == b.trials and a.lam == pytest.approx(b.lam, rel=1e-4)
        np.testing.assert_allclose(est[:, :3], oest[:, :3], atol=1e-4)
        _same_rotation(est[:, 3:], oest[:, 3:], 1e-4)
        per_solve = [it.linear_iterations / max(1, it.trials) for it in log]
        assert max(per_solve) <= 2 and min(per_solve) >= 1, per_solve      # every edge inside the band: the factor is exact
        assert s.linear_iterations * 50 < res[2][0].linear_iterations, (s.linear_iterations, res[2][0].linear_iterations)
        assert s.final_chi2 == pytest.approx(res[2][0].final_chi2, rel=1e-7)
    # the order given, band factor asked for: the closures lie outside the band and stay in the matrix-vector product - about a dozen
    # rounds apiece, same answer (this is the path of graphs whose breadth-first order still leaves edges outside the band)
    c = synth.generate_pg(5, n_node=2000, row_len=2000, n_loop_max=2, min_gap=50, radius=80.0)
    import os
    outs = {}
    for env in ("", "1"):
        if env:
            os.environ["SOSLAM_PG_NO_BFS"] = "1"
        else:
            os.environ.pop("SOSLAM_PG_NO_BFS", None)
        with pg.PoseGraph(pg.default_options(max_iterations=10, preconditioner=3)) as h:
            h.load(c)
            outs[env] = (h.optimize(), h.estimates())
    os.environ.pop("SOSLAM_PG_NO_BFS", None)
    assert outs["1"][0].linear_iterations > 5 * outs[""][0].linear_iterations
    assert outs["1"][0].final_chi2 == pytest.approx(outs[""][0].final_chi2, rel=1e-7)
    np.testing.assert_allclose(outs["1"][1][:, :3], outs[""][1][:, :3], atol=1e-6)
    # the lattice of configs[4] is not a band in any order: AUTO keeps the two-level PCG there (same iteration counts)
    g = synth.generate_pg(5)
    res = {}
    for pre in (0, 2):
        with pg.PoseGraph(pg.default_options(max_iterations=2, preconditioner=pre)) as h:
            h.load(g)
            res[pre] = h.optimize().linear_iterations
    assert res[0] == res[2] and res[0] > 100, res


@pytest.mark.parametrize("case", ["fixed_in_the_middle", "two_components", "shuffled_ids", "several_fixed"])
def test_band_factor_on_awkward_chains(gpu, oracle_lib, case):
    """Shapes the breadth-first numbering of the band path must cope with: the constant vertex in the middle of the chain (two free
    arms), two chains that are not connected (the second without a constant vertex: only the Levenberg shift makes its block
    definite), vertex ids that do not follow the chain (the caller's order is arbitrary), several constant vertices.  Each against the
    oracle's ten iterations."""
    pg, synth, L = gpu
    c = synth.generate_pg(5, n_node=400, row_len=400, n_loop_max=3, min_gap=50, radius=80.0)
    est, ef, et, meas, fixed = c.est.copy(), c.e_from.copy(), c.e_to.copy(), c.meas.copy(), c.fixed.copy()
    if case == "fixed_in_the_middle":
        fixed[:] = 0
        fixed[200] = 1
    elif case == "two_components":
        keep = ~(((ef < 250) & (et >= 250)) | ((et < 250) & (ef >= 250)))     # cut every edge across vertex 250
        ef, et, meas = ef[keep], et[keep], meas[keep]
    elif case == "shuffled_ids":
        perm = np.random.default_rng(3).permutation(len(est))
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        est, fixed = est[perm], fixed[perm]                                   # new id k holds old vertex perm[k]
        ef, et = inv[ef].astype(np.uint32), inv[et].astype(np.uint32)
    else:
        fixed[[0, 133, 399]] = 1
    g = synth.PgProblem(est, ef, et, meas, fixed, c.info)
    oest, osum, olog = _oracle_solve(oracle_lib, g, iters=10)
    with pg.PoseGraph(pg.default_options(max_iterations=10)) as h:
        h.load(g)
        s = h.optimize()
        out, log = h.estimates(), h.iteration_log()
    assert s.iterations == osum.iterations
    for a, b in zip(log, olog):
        assert a.chi2 == pytest.approx(b.chi2, rel=1e-5, abs=1e-12) and a.trials == b.trials
    np.testing.assert_allclose(out[:, :3], oest[:, :3], atol=1e-4)
    _same_rotation(out[:, 3:], oest[:, 3:], 1e-4)
    assert max(it.linear_iterations / max(1, it.trials) for it in log) <= 2, [(it.linear_iterations, it.trials) for it in log]


def test_linearisation_is_bitwise_reproducible(gpu):
    """H and b are summed from per-edge records through fixed-order lists (pg_gather), not by floating-point atomics: two
    linearisations of the same estimates agree bit for bit, and so do two whole solves."""
    pg, synth, L = gpu
    g = synth.generate_pg(5)
    outs = []
    for _ in range(2):
        with pg.PoseGraph(pg.default_options(max_iterations=3)) as h:
            h.load(g)
            lin = h.debug_linearize(dense=False)
            h.optimize()
            outs.append((lin["b"], lin["chi2"], h.estimates()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    np.testing.assert_array_equal(outs[0][2], outs[1][2])


def test_one_call_and_errors(gpu):
    pg, synth, L = gpu
    g = synth.generate_pg(6)
    est = g.est.copy()
    s = pg.PgSummary()
    o = pg.default_options(max_iterations=3)
    st = L.lib().soslam_pg_solve(C.byref(o), len(est), L.ptr(est), L.ptr(g.fixed), len(g.e_from), L.ptr(g.e_from), L.ptr(g.e_to),
                                 L.ptr(np.ascontiguousarray(g.meas)), L.ptr(np.ascontiguousarray(g.info)), C.byref(s))
    assert st == 0 and s.final_chi2 < s.initial_chi2 and not np.array_equal(est, g.est)
    bad = g.e_to.copy()
    bad[3] = len(est) + 5
    est2 = g.est.copy()
    st = L.lib().soslam_pg_solve(C.byref(o), len(est2), L.ptr(est2), L.ptr(g.fixed), len(g.e_from), L.ptr(g.e_from), L.ptr(bad),
                                 L.ptr(np.ascontiguousarray(g.meas)), L.ptr(np.ascontiguousarray(g.info)), C.byref(s))
    assert st == L.ERR_INVALID_ARGUMENT
    np.testing.assert_array_equal(est2, g.est)


def test_degenerate_graphs(gpu):
    """What PoseGraphOptimizer::Optimize can hand over at the start of a run: a single fixed vertex without edges, and
    two vertices joined by one odometry edge (the free one must move onto the measurement, chi2 -> 0)."""
    pg, synth, L = gpu
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    info = np.diag([0.01, 0.01, 0.01, 1.0, 1.0, 1.0]).reshape(36)
    with pg.PoseGraph() as h:
        h.set_graph(ident[None, :].copy(), np.array([1], np.uint8), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 7)), info)
        s = h.optimize()
        assert s.initial_chi2 == 0.0 and s.final_chi2 == 0.0
        np.testing.assert_array_equal(h.estimates(), ident[None, :])
    est = np.stack([ident, np.array([0.3, -0.1, 0.2, 0, 0, 0, 1.0])])
    meas = np.array([[1.0, 0.0, 0.0, 0.0, 0.0, np.sin(0.05), np.cos(0.05)]])
    with pg.PoseGraph() as h:
        h.set_graph(est.copy(), np.array([1, 0], np.uint8), np.array([0], np.uint32), np.array([1], np.uint32), meas, info)
        s = h.optimize()
        out = h.estimates()
    assert s.final_chi2 < 1e-12 * max(1.0, s.initial_chi2) + 1e-14
    np.testing.assert_array_equal(out[0], ident)
    np.testing.assert_allclose(out[1], meas[0], atol=1e-7)


def test_appended_graph_equals_the_whole_graph(gpu):
    """soslam_pg_append (the reference's optimizer persists and grows, /root/reference/src/pose_graph_optimizer.cpp:56-59):
    a graph uploaded in two pieces is the whole graph, bit for bit; and vertices that were optimised before an append keep
    their optimised estimates, exactly as if the whole graph had been uploaded with them."""
    pg, synth, L = gpu
    g = synth.generate_pg(6)
    nv, ne = len(g.est), len(g.e_from)
    v1 = nv // 2
    first = (g.e_from < v1) & (g.e_to < v1)                 # edges inside the first half of the vertices
    with pg.PoseGraph(pg.default_options(max_iterations=4)) as h:
        h.load(g)
        h.optimize()
        whole = h.estimates()
    order = np.concatenate([np.nonzero(first)[0], np.nonzero(~first)[0]])
    with pg.PoseGraph(pg.default_options(max_iterations=4)) as h:
        h.append(g.est[:v1], g.fixed[:v1], g.e_from[first], g.e_to[first], g.meas[first], g.info)
        h.append(g.est[v1:], g.fixed[v1:], g.e_from[~first], g.e_to[~first], g.meas[~first])
        assert (h.n_vertex, h.n_edge) == (nv, ne)
        h.optimize()
        pieces = h.estimates()
    with pg.PoseGraph(pg.default_options(max_iterations=4)) as h:     # the same edge order, uploaded at once
        h.set_graph(g.est, g.fixed, g.e_from[order], g.e_to[order], g.meas[order], g.info)
        h.optimize()
        reordered = h.estimates()
    np.testing.assert_array_equal(pieces, reordered)
    np.testing.assert_allclose(pieces, whole, atol=1e-9)               # edge order only changes the order of the sums
    # optimise, then grow: the old vertices continue from their optimised estimates
    with pg.PoseGraph(pg.default_options(max_iterations=4)) as h:
        h.append(g.est[:v1], g.fixed[:v1], g.e_from[first], g.e_to[first], g.meas[first], g.info)
        h.optimize()
        e1 = h.estimates()
        h.append(g.est[v1:], g.fixed[v1:], g.e_from[~first], g.e_to[~first], g.meas[~first])
        h.optimize()
        grown = h.estimates()
    with pg.PoseGraph(pg.default_options(max_iterations=4)) as h:
        h.set_graph(np.concatenate([e1, g.est[v1:]]), g.fixed, g.e_from[order], g.e_to[order], g.meas[order], g.info)
        h.optimize()
        np.testing.assert_array_equal(grown, h.estimates())


def test_levenberg_schedule_matches_the_independent_restatement(gpu, golden_dir):
    """The device path's Levenberg controller against the numpy / autograd restatement of g2o's loop
    (oracle/gen_controller_golden.py): chi2, lambda and the trials of every iteration, one rejected trial on the way."""
    pg, synth, L = gpu
    g = np.load(os.path.join(golden_dir, "pg_lm_trajectory.npz"), allow_pickle=False)
    n_it = len(g["chi2"])
    fixed = np.zeros(len(g["est0"]), np.uint8)
    fixed[0] = 1
    with pg.PoseGraph(pg.default_options(max_iterations=n_it, pcg_tolerance=1e-13)) as h:
        h.set_graph(g["est0"], fixed, g["e_from"], g["e_to"], g["meas"], g["info"])
        s = h.optimize()
        est, log = h.estimates(), h.iteration_log()
    assert s.iterations == n_it
    assert [e.trials for e in log] == g["trials"].tolist()
    np.testing.assert_allclose([e.chi2 for e in log], g["chi2"], rtol=1e-6)
    np.testing.assert_allclose([e.lam for e in log], g["lam"], rtol=1e-5)
    np.testing.assert_allclose(est[:, :3], g["est"][:, :3], atol=1e-6)
