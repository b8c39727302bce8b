"""CPU tests of everything that does not need a GPU: the shared host-side setup
(equilibration, KKT assembly, multi-level chain factorisation) and the ADMM
driver, executed through the oracle's CPU twin of the C ABI; the Python API
(solve_score and friends) on top of it."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import GOLDEN_NAMES, compare_residuals_with_golden, compare_with_golden, graph_by_name, load_golden
from score_amd import compat
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import (
    solve_problem_with_intermediate_iterates,
    solve_score,
    solve_score_batch,
)
from score_amd.solver import ConicSolver


@pytest.mark.parametrize("name", ["manhattan", "synth_a", "synth_b", "synth_c", "synth_d", "graph3d"])
@pytest.mark.parametrize("relax", ["SOCP", "QCQP"])
def test_twin_solve_score_matches_golden(name, relax, fixtures, twin_lib):
    fg = graph_by_name(name, fixtures)
    gold = load_golden(name)
    res = solve_score(fg, relax, lib_path=twin_lib)
    assert res.solved, res.info
    assert res.info["pobj"] == pytest.approx(float(gold["objective"]), rel=1e-5, abs=1e-6)
    compare_with_golden(res, gold)
    compare_residuals_with_golden(res, fg, gold)  # also where the poses are not unique
    assert res.pose_chain_names == fg.get_pose_chain_names()
    d = fg.dimension
    key = (fg.range_measurements[0].first_key, fg.range_measurements[0].second_key) if fg.range_measurements else None
    if key is not None:
        assert res.distances[key].shape == ((1,) if relax == "SOCP" else (d,))
        if relax == "QCQP":
            assert max(np.linalg.norm(v) for v in res.distances.values()) <= 1 + 1e-9
    T = res.poses[fg.pose_variables[0][0].name]
    np.testing.assert_allclose(T, np.eye(d + 1), atol=1e-12)  # the pinned pose


def test_twin_goats(fixtures, twin_lib):
    """BASELINE config 0: the GOATS AUV data set (CPU plumbing case)."""
    res = solve_score(fixtures["goats"], lib_path=twin_lib)  # default relaxation = QCQP
    gold = load_golden("goats")
    assert res.solved
    assert res.info["pobj"] == pytest.approx(float(gold["objective"]), rel=1e-6)
    compare_with_golden(res, gold)
    compare_residuals_with_golden(res, fixtures["goats"], gold)


def test_qcqp_direct_equals_via_socp(fixtures, twin_lib):
    fg = fixtures["manhattan"]
    a = solve_score(fg, "QCQP", lib_path=twin_lib)
    b = solve_score(fg, "QCQP", qcqp_mode="direct", lib_path=twin_lib)
    assert a.solved and b.solved
    assert a.info["pobj"] == pytest.approx(b.info["pobj"], rel=1e-6, abs=1e-6)
    for nm in a.poses:
        np.testing.assert_allclose(a.poses[nm], b.poses[nm], atol=2e-4)
    # r_ij agree wherever the measured distance is not (numerically) zero
    for m in fg.range_measurements:
        if m.dist > 1e-6:
            k = (m.first_key, m.second_key)
            np.testing.assert_allclose(a.distances[k], b.distances[k], atol=5e-4)


@pytest.mark.parametrize("dim", [2, 3])
def test_direct_qcqp_is_solved_in_head_form_and_mapped_back(twin_lib, monkeypatch, dim):
    """csrc/score_headform.hpp (the reference's default relaxation, gurobi_utils.py:341-344, :488-496, handed over as it is):
    the library rewrites the constant-head unit-ball cones into private-head cones, solves THAT program and maps the
    solution back.  What comes back must be a KKT point of the program AS GIVEN -- checked here with the given P, q, A, b
    alone -- with the SOCP relaxation's optimal value, zero-distance measurements (idle cones) included."""
    from score_amd.assemble import assemble
    from score_amd.manhattan import make_manhattan_3d
    from score_amd.solver import ConicSolver

    if dim == 2:
        fg = make_manhattan(n_robots=2, n_poses=50, n_beacons=3, seed=14, p_range=0.5)  # (synth_d: two ranges of distance 0)
        assert min(m.dist for m in fg.range_measurements) == 0.0
    else:
        fg = make_manhattan_3d(n_robots=2, n_poses=25, n_beacons=3, seed=12, p_range=0.6)
    qp = assemble(fg, "QCQP").qp
    ref = assemble(fg, "SOCP").qp
    sv = ConicSolver([qp], dict(eps_abs=1e-9, eps_rel=1e-9, max_iters=60000), lib_path=twin_lib)
    sol = sv.solve()[0]
    assert (sv.n_total, sv.m_total) == (qp.n, qp.m)  # sizes at the boundary are the caller's
    sv.close()
    sv = ConicSolver([ref], dict(eps_abs=1e-9, eps_rel=1e-9, max_iters=60000), lib_path=twin_lib)
    sref = sv.solve()[0]
    sv.close()
    assert sol.solved and sref.solved
    x, y, s = sol.x, sol.y, sol.s
    scale = max(1.0, np.abs(y).max())
    assert np.abs(qp.A @ x + s - qp.b).max() <= 1e-12
    assert np.abs(qp.P @ x + qp.q + qp.A.T @ y).max() <= 2e-6 * scale
    T = dim
    S, Y = s.reshape(-1, T + 1), y.reshape(-1, T + 1)
    assert np.all(S[:, 0] == 1.0) and np.all(np.linalg.norm(S[:, 1:], axis=1) <= 1.0 + 1e-12)
    assert np.all(Y[:, 0] >= np.linalg.norm(Y[:, 1:], axis=1) - 1e-12 * scale)
    assert np.abs((S * Y).sum(axis=1)).max() <= 1e-9 * scale
    assert qp.objective(x) == pytest.approx(ref.objective(sref.x), rel=1e-7, abs=1e-7)  # (sums of terms of magnitude 1e5..1e6)
    assert sol.info["pobj"] == pytest.approx(qp.objective(x), rel=1e-9, abs=1e-7)
    # the plain splitting loop on the program as given (rounds 1-4) ends at the same value
    monkeypatch.setenv("SCORE_QCQP_PLAIN", "1")
    sv = ConicSolver([qp], dict(eps_abs=1e-6, eps_rel=1e-6, max_iters=40000, cg_iters=8, adaptive_rho=0), lib_path=twin_lib)
    plain = sv.solve()[0]
    sv.close()
    assert plain.solved
    assert plain.info["pobj"] == pytest.approx(sol.info["pobj"], rel=1e-5, abs=1e-5)
    # (x itself is not unique here: idle directions, a landmark no active cone determines)


def test_head_form_keeps_equality_rows(twin_lib):
    """The rewrite of constant-head unit-ball cones (csrc/score_headform.hpp) on a program that also has zero-cone rows in
    front (the general ABI form: z > 0; here one landmark coordinate fixed by an equality): the rows keep their places, their
    duals and slacks come back unchanged, and the returned point is a KKT point of the program as given."""
    import scipy.sparse as sp

    from oracle import score_oracle as so

    from score_amd.assemble import ConicQP, assemble
    from score_amd.solver import ConicSolver

    fg = make_manhattan(n_robots=2, n_poses=40, n_beacons=3, seed=21, p_range=0.4)
    qp = assemble(fg, "QCQP").qp
    col = (sum(len(c) for c in fg.pose_variables) - 1) * 3  # landmark 0's x coordinate (replica 0)
    A2 = sp.vstack([sp.csr_matrix(([1.0], ([0], [col])), shape=(1, qp.n)), qp.A]).tocsr()
    qp2 = ConicQP(P=qp.P, q=qp.q, c0=qp.c0, A=A2, b=np.concatenate([[3.25], qp.b]), z=1, soc_dims=qp.soc_dims, chain_ptr=qp.chain_ptr,
                  node_cols=qp.node_cols, block_size=qp.block_size, rep_d=qp.rep_d, rep_n=qp.rep_n)
    sv = ConicSolver([qp2], dict(eps_abs=1e-8, eps_rel=1e-8, max_iters=100000), lib_path=twin_lib)
    out = sv.solve()[0]
    assert (sv.n_total, sv.m_total) == (qp2.n, qp2.m)
    sv.close()
    assert out.solved and out.x[col] == pytest.approx(3.25, abs=1e-7) and out.s[0] == pytest.approx(0.0, abs=1e-7)
    cert = so.kkt_certificate(qp2.P, qp2.q, qp2.A, qp2.b, 1, qp2.soc_dims, out.x, out.y, out.s)
    assert cert["primal_res_inf"] < 1e-7 and cert["dual_res_inf"] < 1e-5 and cert["s_cone_dist"] < 1e-12 and cert["y_cone_dist"] < 1e-12, cert


def test_batch_equals_individual(twin_lib):
    graphs = [make_manhattan(n_robots=3, n_poses=40 + 10 * (s % 4), n_beacons=4, seed=s, p_range=0.4) for s in (300, 303, 305)]
    batch = solve_score_batch(graphs, "SOCP", lib_path=twin_lib, lockstep=True)
    for g, rb in zip(graphs, batch):
        ri = solve_score(g, "SOCP", lib_path=twin_lib)
        assert rb.solved and ri.solved
        assert rb.info["iters"] == ri.info["iters"]
        for nm in ri.poses:
            np.testing.assert_allclose(rb.poses[nm], ri.poses[nm], atol=1e-9)


def test_batch_grouping_keeps_order_and_results(twin_lib):
    """solve_score_batch's default policy splits the graphs into lock-step groups (one handle per
    group, groups driven from a thread pool): results come back in input order and equal the
    one-by-one solves, whatever the grouping."""
    graphs = [make_manhattan(n_robots=2, n_poses=20 + 3 * i, n_beacons=4, seed=800 + i, p_range=0.8) for i in range(5)]
    ref = [solve_score(g, "SOCP", lib_path=twin_lib) for g in graphs]
    for kw in (dict(workers=2), dict(lockstep=False, workers=3)):
        out = solve_score_batch(graphs, "SOCP", lib_path=twin_lib, **kw)
        assert len(out) == len(graphs)
        for g, a, b in zip(graphs, out, ref):
            assert a.solved and a.pose_chain_names == g.get_pose_chain_names()
            assert a.info["pobj"] == pytest.approx(b.info["pobj"], rel=1e-6, abs=1e-8)
            last = g.pose_variables[0][-1].name
            np.testing.assert_allclose(a.poses[last], b.poses[last], atol=1e-6)


def test_intermediate_iterates(twin_lib):
    fg = make_manhattan(n_robots=2, n_poses=40, n_beacons=2, seed=4)
    its = solve_problem_with_intermediate_iterates(fg, "SOCP", every=25, lib_path=twin_lib)
    assert len(its) >= 2 and its[-1].solved
    iters = [r.info["iters"] for r in its]
    assert iters == sorted(iters) and iters[0] == 6  # the warm-up (polish_warmup = 6) is one snapshot at every=25
    final = solve_score(fg, "SOCP", lib_path=twin_lib)
    for nm in final.poses:
        np.testing.assert_allclose(its[-1].poses[nm], final.poses[nm], atol=1e-4)


def test_api_errors(twin_lib):
    fg = make_manhattan(n_robots=1, n_poses=6, n_beacons=1, seed=0, p_range=1.0)
    with pytest.raises(ValueError, match="not supported"):
        solve_score(fg, "LP", lib_path=twin_lib)
    fg.landmark_variables.append(compat.LandmarkVariable2D("L9"))
    with pytest.raises(AssertionError, match="unconnected"):
        solve_score(fg, lib_path=twin_lib)
    fg.landmark_variables.pop()
    # stale example call shape: solve_score(data, solver_params, relaxation)
    res = solve_score(fg, object(), "SOCP", lib_path=twin_lib)
    assert res.solved
    # non-convergence is reported through solved=False, not an exception
    res = solve_score(fg, "SOCP", solver_settings=dict(max_iters=25, eps_abs=1e-14, eps_rel=1e-14), lib_path=twin_lib)
    assert res.solved is False and res.info["status"] == 2


def test_flat_arrays_are_kept_on_the_graph_and_follow_its_lists(twin_lib):
    """The drop-in call on a graph solved before (score/solve_score.py:54-57: the same FactorGraphData for both relaxations, for
    the intermediate iterates): the pass over the measurement objects happens once, the arrays stay on the object while its
    lists stand (native.cached_graph_arrays), and every change of a list -- append, pop, replace -- is seen.  total_time is
    setup + solve, the counterpart of model.Runtime (gurobi_utils.py:194)."""
    from score_amd import native

    fg = make_manhattan(n_robots=2, n_poses=25, n_beacons=2, seed=3, p_range=0.5)
    a = solve_score(fg, "SOCP", lib_path=twin_lib)
    kept = getattr(fg, native._CACHE_ATTR)
    assert kept[0] == native.graph_fingerprint(fg)
    b = solve_score(fg, "QCQP", lib_path=twin_lib)
    assert getattr(fg, native._CACHE_ATTR)[1] is kept[1]  # the second call read no measurement object
    assert a.solved and b.solved and a.info["pobj"] == pytest.approx(b.info["pobj"], rel=1e-6, abs=1e-6)
    assert a.total_time == pytest.approx((a.info["setup_ms"] + a.info["solve_ms"]) * 1e-3) and a.total_time > a.info["solve_ms"] * 1e-3
    # a dropped measurement changes the answer: the cached arrays must not be used
    nr = len(fg.range_measurements)
    last = fg.range_measurements.pop()
    c = solve_score(fg, "SOCP", lib_path=twin_lib)
    assert len(c.distances) == nr - 1 and getattr(fg, native._CACHE_ATTR)[1] is not kept[1]
    fg.range_measurements.append(last)
    d = solve_score(fg, "SOCP", lib_path=twin_lib)
    # (a noise-free graph: the optimum is 0; the twin adds its partial sums in a fixed order since round 6 -- before, pobj here was
    #  what the OpenMP reductions left of it, +-1e-9 run to run)
    assert len(d.distances) == nr and d.info["pobj"] == pytest.approx(a.info["pobj"], rel=1e-9, abs=1e-8)
    # same list, same length, another element: seen through the first / last identities; an in-place edit needs the explicit call
    fg.range_measurements[-1] = compat.FGRangeMeasurement(last.association, dist=last.dist + 1.0, stddev=last.stddev)
    assert native.graph_fingerprint(fg) != getattr(fg, native._CACHE_ATTR)[0]
    native.invalidate_graph_cache(fg)
    assert not hasattr(fg, native._CACHE_ATTR)
    # the unconnected-variable check still fires on a cached graph (it is evaluated on the arrays of the CURRENT lists)
    fg.landmark_variables.append(compat.LandmarkVariable2D("L9"))
    with pytest.raises(AssertionError, match="unconnected"):
        solve_score(fg, lib_path=twin_lib)


def test_no_ranges_and_single_pose_chain(twin_lib):
    """Edge cases: m = 0 (no cones at all) and a robot with a single pose."""
    fg = make_manhattan(n_robots=1, n_poses=30, n_beacons=0, seed=2)
    assert assemble(fg, "SOCP").qp.m == 0
    res = solve_score(fg, "SOCP", lib_path=twin_lib)
    assert res.solved and res.info["pobj"] == pytest.approx(0.0, abs=1e-6)
    # odometry alone: the estimate composes the measurements
    T = np.eye(3)
    for m in fg.odom_measurements[0]:
        step = np.eye(3); step[:2, :2] = m.rotation_matrix; step[:2, 2] = m.translation_vector
        T = T @ step
    np.testing.assert_allclose(res.poses["A29"][:2, 2], T[:2, 2], atol=1e-5)
    fg2 = make_manhattan(n_robots=2, n_poses=20, n_beacons=2, seed=3, p_range=0.5)
    fg2.pose_variables[1] = fg2.pose_variables[1][:1]  # robot B keeps one pose
    fg2.odom_measurements[1] = []
    names = {p.name for c in fg2.pose_variables for p in c} | {l.name for l in fg2.landmark_variables}
    fg2.range_measurements = [m for m in fg2.range_measurements if m.first_key in names and m.second_key in names]
    assert "B0" in {m.first_key for m in fg2.range_measurements} | {m.second_key for m in fg2.range_measurements}
    assert solve_score(fg2, "SOCP", lib_path=twin_lib).solved


def test_abi_rejects_bad_problems(twin_lib):
    fg = make_manhattan(n_robots=1, n_poses=8, n_beacons=1, seed=0, p_range=1.0)
    qp = assemble(fg, "SOCP").qp
    bad = assemble(fg, "SOCP").qp
    bad.soc_dims = bad.soc_dims.copy(); bad.soc_dims[0] = 2  # z + sum(dims) != m
    with pytest.raises(RuntimeError, match="soc_dims"):
        ConicSolver(bad, lib_path=twin_lib)
    bad = assemble(fg, "SOCP").qp
    bad.A = sp.csr_matrix((bad.A.data, bad.A.indices + 10**6, bad.A.indptr), shape=(bad.m, bad.n + 10**6 + 1))[:, : bad.n + 10**6 + 1]
    bad.A._shape = (qp.m, qp.n)
    with pytest.raises(RuntimeError, match="out of range"):
        ConicSolver(bad, lib_path=twin_lib)
    with pytest.raises(ValueError, match="unknown solver setting"):
        ConicSolver(qp, dict(no_such=1), lib_path=twin_lib)


@pytest.mark.parametrize("radix", [2, 3, 4])
def test_chain_preconditioner_is_the_exact_chain_inverse(radix, twin_lib):
    """With a problem that consists ONLY of chains (odometry, no ranges) the
    multi-level factorisation is an exact solver for K, so one PCG step gives the
    exact KKT solution: a single ADMM iteration must reproduce the direct solve."""
    fg = make_manhattan(n_robots=3, n_poses=37, n_beacons=0, seed=21, p_range=0.0)
    qp = assemble(fg, "SOCP").qp
    assert qp.m == 0
    x_direct = spla.spsolve((qp.P + 1e-3 * sp.identity(qp.n)).tocsc(), -qp.q)
    # fac_fp32 = 0: factors in double -- exact to rounding; 1 (default): factors kept to float precision, the
    # inverse is exact to float eps times the conditioning of the chains
    for fp32, tol in ((0, 1e-8), (1, 2e-4)):
        st = dict(scale_iters=0, cg_iters=1, adaptive_cg=0, adaptive_rho=0, sigma=1e-3, alpha=1.0, chain_radix=radix, fac_fp32=fp32)
        sol = ConicSolver(qp, st, lib_path=twin_lib)
        sol.reset()
        out = sol.steps(1)[0]
        sol.close()
        np.testing.assert_allclose(out.x, x_direct, rtol=tol, atol=tol * np.abs(x_direct).max())


def test_equilibration_and_kkt_values(fixtures, twin_lib):
    """Host logic: D, E > 0 with one scale per cone, and K = D(P)D + sigma I + rho (EAD)'(EAD)."""
    mdl = assemble(fixtures["manhattan"], "SOCP")
    qp = mdl.qp
    sol = ConicSolver(qp, dict(rho=0.37, sigma=1e-6), lib_path=twin_lib)
    D, E, Kval = sol.debug_get("D"), sol.debug_get("E"), sol.debug_get("Kval")
    assert D.min() > 0 and E.min() > 0
    Ec = E.reshape(-1, 3)
    assert np.all(Ec == Ec[:, :1])
    Ps = sp.diags(D) @ qp.P @ sp.diags(D)
    As = sp.diags(E) @ qp.A @ sp.diags(D)
    K = (Ps + 1e-6 * sp.identity(qp.n) + 0.37 * (As.T @ As)).tocsr()
    K.sort_indices()
    assert abs(K).max(axis=0).toarray().max() < 1e3  # equilibrated (unscaled: 5e5)
    assert Kval.size >= K.nnz
    assert np.isclose(Kval.sum(), K.data.sum(), rtol=1e-10)


def test_host_setup_does_not_depend_on_the_thread_count(twin_lib, tmp_path):
    """score_create spreads equilibration, the KKT pattern and the chain factorisations over host
    threads; every thread computes what the serial loop would, so the scaled problem -- and with
    OMP_NUM_THREADS=1 the whole iterate sequence of the twin -- must be bitwise the same on one
    CPU (affinity mask of a child process) and on all of them."""
    import os
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(
        "import sys, os\n"
        "if sys.argv[2] == 'one': os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import numpy as np\n"
        "from score_amd.manhattan import make_manhattan\n"
        "from score_amd.assemble import assemble\n"
        "from score_amd.solver import ConicSolver\n"
        "qp = assemble(make_manhattan(n_robots=6, n_poses=3000, n_beacons=3, seed=4), 'SOCP').qp\n"
        "s = ConicSolver(qp, dict(polish=0), lib_path=sys.argv[1])\n"
        "out = s.steps(30)[0]\n"
        "np.save(sys.argv[3], np.concatenate([out.x, out.y, out.s]))\n"
    )
    env = dict(os.environ, OMP_NUM_THREADS="1")
    outs = []
    for mode in ("one", "all"):
        f = str(tmp_path / f"{mode}.npy")
        subprocess.run([sys.executable, str(script), twin_lib, mode, f], check=True, env=env, timeout=600)
        outs.append(np.load(f))
    if len(os.sched_getaffinity(0)) < 2:
        pytest.skip("single-CPU machine: nothing to compare")
    assert np.array_equal(outs[0], outs[1])


def test_array_graph_input_equals_object_input(twin_lib):
    """native.ArrayGraph (a factor graph that already is flat arrays) goes through the same path as a
    FactorGraphData, minus the per-measurement attribute reads."""
    from score_amd.manhattan import make_manhattan
    from score_amd.native import ArrayGraph, graph_arrays

    gs = [make_manhattan(n_robots=2, n_poses=30, n_beacons=2, seed=60 + i, p_range=0.4, n_loop_closures=i) for i in range(3)]
    a = solve_score_batch(gs, "SOCP", lib_path=twin_lib)
    b = solve_score_batch([ArrayGraph(graph_arrays(g)) for g in gs], "SOCP", lib_path=twin_lib)
    for x, y in zip(a, b):
        assert x.solved and y.solved and abs(x.info["iters"] - y.info["iters"]) <= 25 and x.pose_chain_names == y.pose_chain_names
        assert x.info["pobj"] == pytest.approx(y.info["pobj"], rel=1e-6, abs=1e-8)
        for nm in x.poses:  # (two ADMM runs of the OpenMP twin stop within the tolerance of each other, not bitwise)
            np.testing.assert_allclose(x.poses[nm], y.poses[nm], atol=1e-3)
        assert list(x.distances.keys()) == list(y.distances.keys())


@pytest.mark.parametrize("name,relax", [("manhattan", "SOCP"), ("synth_b", "SOCP"), ("graph3d", "SOCP"), ("synth_d", "QCQP")])
def test_row_replicated_setup_equals_the_general_one(name, relax, fixtures, twin_lib, monkeypatch):
    """The model couples one row k of the pose matrices at a time (gurobi_utils.py:504-526); only the cones couple the
    rows (:345-352).  Both assemblers order the unknowns replica by replica and announce it (score_problem::rep_d);
    the solver then keeps K_row once: K and A' hold replica 0's rows, a replica's chain uses its owner's factors.
    Here the host side of that: the twin run on the replicated structures (SCORE_TWIN_REPLICATION) must walk through
    the same iterates as the twin run on the full K (3-D: three replicas; QCQP: no tail; loop closures)."""
    fg = graph_by_name(name, fixtures)
    direct = relax == "QCQP"
    mdl = assemble(fg, relax)
    qp = mdl.qp
    d = fg.dimension
    assert qp.rep_d == d and qp.rep_n * d + (0 if direct else len(mdl.range_keys)) == qp.n
    # the structure itself, on the assembled matrices: P = I_d (x) P_row (+ tail), no coupling between the replicas
    nr = qp.rep_n
    P = qp.P.tocsr()
    P0 = P[:nr, :nr]
    for k in range(1, d):
        assert abs(P[k * nr : (k + 1) * nr, k * nr : (k + 1) * nr] - P0).max() == 0.0
    off = P[: d * nr, : d * nr].copy().tolil()
    for k in range(d):
        off[k * nr : (k + 1) * nr, k * nr : (k + 1) * nr] = 0
    assert off.tocsr().count_nonzero() == 0
    outs = {}
    for mode in ("general", "replicated"):
        if mode == "replicated":
            monkeypatch.setenv("SCORE_TWIN_REPLICATION", "1")
        else:
            monkeypatch.delenv("SCORE_TWIN_REPLICATION", raising=False)
        sol = ConicSolver(qp, dict(polish=0, adaptive_rho=0, adaptive_cg=0, fac_fp32=0), lib_path=twin_lib)
        outs[mode] = (sol.steps(60)[0], sol.debug_get("Kval"), sol.debug_get("z"), sol.debug_get("kx"))
        sol.close()
    (a, Ka, za, kxa), (b, Kb, zb, kxb) = outs["general"], outs["replicated"]
    # the replicated K holds replica 0's rows and the tail's: fewer values, same entries
    assert Kb.size < Ka.size and np.isclose(Kb.sum() * 1.0, Kb.sum())
    scale = np.abs(a.x).max()
    np.testing.assert_allclose(b.x, a.x, atol=1e-9 * scale)
    np.testing.assert_allclose(b.y, a.y, atol=1e-9 * max(1.0, np.abs(a.y).max()))
    np.testing.assert_allclose(zb, za, atol=1e-9 * max(1e-300, np.abs(za).max()))
    np.testing.assert_allclose(kxb, kxa, atol=1e-9 * max(1e-300, np.abs(kxa).max()))
    assert b.info["pobj"] == pytest.approx(a.info["pobj"], rel=1e-7)  # (sums of terms of magnitude 1e5: x to 1e-9)
    # a full solve on the replicated structures ends at the golden optimum
    monkeypatch.setenv("SCORE_TWIN_REPLICATION", "1")
    res = solve_score(fg, relax, qcqp_mode="direct" if direct else "via_socp", solver_settings=dict(max_iters=30000), lib_path=twin_lib)
    assert res.solved
    if name in GOLDEN_NAMES:
        compare_with_golden(res, load_golden(name), pose_tol=1e-4)


@pytest.mark.parametrize("name,relax,replicated", [
    ("goats", "SOCP", True), ("manhattan", "SOCP", True), ("manhattan", "SOCP", False), ("synth_b", "SOCP", True),
    ("graph3d", "SOCP", True), ("graph3d", "SOCP", False), ("synth_d", "QCQP", True), ("prior2d", "SOCP", True),
])
def test_band_view_layout_reproduces_the_csr_rows(name, relax, replicated, fixtures, twin_lib, monkeypatch):
    """The product streams K (and, on request, the Newton matrix) through a band view (csrc/score_band.hpp): chain rows as
    value-slot pairs without column indices, a per-tile remainder, diagonal tiles, CSR tiles with long rows in segments.
    The layout builder is host code shared with the twin, which applies it with band_apply_host (the specification of
    k_spmv_band) and compares with the CSR rows: same products to rounding, every unknown's row served exactly once --
    with and without the row-replicated host structures, 2-D and 3-D, loop closures, priors, the direct QCQP form (no
    distance rows), the reference's GOATS data (one robot, 4 beacons seen hundreds of times: split long rows)."""
    fg = graph_by_name(name, fixtures)
    qp = assemble(fg, relax).qp if relax != "QCQP" else _model_for_direct(fg).qp
    if replicated:
        monkeypatch.setenv("SCORE_TWIN_REPLICATION", "1")
    else:
        monkeypatch.delenv("SCORE_TWIN_REPLICATION", raising=False)
    sol = ConicSolver(qp, dict(polish=0), lib_path=twin_lib)
    k = sol.debug_get("band_check")
    assert k[1] == 1.0 and k[0] < 1e-12, k           # view on, products equal
    assert k[2] >= 1 and k[5] in (8.0, 10.0, 12.0)    # band tiles exist; 4, 5 or 6 slot pairs per row
    d = fg.dimension
    assert k[5] == (8.0 if d == 2 else 12.0)
    if relax == "SOCP":
        h = sol.debug_get("band_check_h")
        assert h[1] == 1.0 and h[0] < 1e-12, h
    sol.close()


def _model_for_direct(fg):
    from score_amd.solve_score import _model_for

    return _model_for(fg, "QCQP", "direct", assembler="python")


def test_broadcast_graphs_without_a_process_group():
    """broadcast_graphs / solve_score_sharded(root=) outside torch.distributed: the single-process code path returns the
    graphs as ArrayGraphs and solves them like solve_score_batch."""
    from score_amd.distributed import broadcast_graphs
    from score_amd.native import ArrayGraph

    graphs = [make_manhattan(n_robots=2, n_poses=12 + i, n_beacons=2, seed=70 + i) for i in range(2)]
    out = broadcast_graphs(graphs, root=0)
    assert len(out) == 2 and all(isinstance(g, ArrayGraph) for g in out)
    assert out[0].num_poses == 24 and out[1].num_poses == 26


def test_graph_handles_and_the_estimate_read_back(twin_lib):
    """score_create_from_graphs / score_read_estimates / score_graphs_connected through the ABI (here: the twin, which builds
    the model with the host assembler and reads the estimate back with read_estimates_host -- the host specification of the
    device kernel): the estimate equals what the index maps and the reference's rounding make of the SAME solve's x --
    homogeneous poses with the pinned [I | 0], the relaxed blocks, landmarks, SOCP distances, QCQP directions (native and
    reconstructed from the SOCP), 2-D and 3-D; GraphModel's reshapes equal ScoreModel's maps."""
    from score_amd.native import assemble_native, graph_arrays, graph_model, graphs_connected
    from score_amd.rounding import round_to_special_orthogonal
    from score_amd.solver import ConicSolver

    g2 = make_manhattan(n_robots=3, n_poses=40, n_beacons=3, seed=11, p_range=0.5)
    from score_amd.manhattan import make_manhattan_3d

    g3 = make_manhattan_3d(n_robots=2, n_poses=25, n_beacons=3, seed=12, p_range=0.6)
    for fg in (g2, g3):
        a = graph_arrays(fg)
        d = int(a["dim"])
        assert graphs_connected([a, a], lib_path=twin_lib) is None
        for relax, qdirs in (("SOCP", False), ("SOCP", True), ("QCQP", False)):
            sv = ConicSolver.from_graphs([a], 0 if relax == "SOCP" else 1, dict(cg_iters=8) if relax == "QCQP" else {}, lib_path=twin_lib)
            infos, ests, x = sv.solve_estimates(qcqp_directions=qdirs, return_x=True)
            sv_n = sv.n_total
            sv.close()
            T, B, Lm, Rg, flags = ests[0]
            gm, full = graph_model(a, relax), assemble_native(fg, relax, lib_path=twin_lib, arrays=a)
            assert np.array_equal(gm.free_cols, full.free_cols) and gm.n_model == full.n_model
            xm = full.expand(x)
            blocks, lms, rng = gm.views(x)
            assert np.array_equal(blocks, full.pose_blocks(xm)) and np.array_equal(lms, full.landmark_block(xm)) and np.array_equal(rng, full.range_block(xm))
            if relax == "QCQP":  # (solved in its head form, csrc/score_headform.hpp: x's directions are r*(u) of the rewrite)
                assert sv_n == full.qp.n
            assert np.array_equal(B, blocks) and np.array_equal(Lm, lms) and not flags.any()
            assert np.array_equal(B[0], np.hstack([np.eye(d), np.zeros((d, 1))]))
            R = round_to_special_orthogonal(blocks[:, :, :d])
            np.testing.assert_allclose(T[:, :d, :d], R, atol=1e-12)
            assert np.array_equal(T[:, :d, d], blocks[:, :, d]) and np.array_equal(T[:, d, :d], np.zeros((len(T), d))) and np.all(T[:, d, d] == 1.0)
            if relax == "QCQP":
                np.testing.assert_allclose(Rg, rng, atol=1e-12)
            elif not qdirs:
                assert np.array_equal(Rg, rng)
            else:
                tr = np.concatenate([blocks[:, :, d], lms])
                delta = tr[a["rng_a"]] - tr[a["rng_b"]]
                den = np.maximum(np.sqrt((delta * delta).sum(axis=1)), a["rng_dist"])
                np.testing.assert_allclose(Rg, delta / den[:, None], atol=1e-14)
    # a graph with an unconnected landmark: the check names it through the index of the failing graph
    a_bad = dict(graph_arrays(g2))
    a_bad["landmark_names"] = list(a_bad["landmark_names"]) + ["L_free"]
    assert graphs_connected([graph_arrays(g2), a_bad], lib_path=twin_lib) == 1
    # handles made from score_problem arrays do not know the graph
    plain = ConicSolver([assemble_native(g2, "SOCP", lib_path=twin_lib).qp], {}, lib_path=twin_lib)
    with pytest.raises(RuntimeError, match="not made by ConicSolver.from_graphs"):
        plain.solve_estimates()
    plain.close()


def _lc_between(fg, rng, a, i, b, j):
    Ta, Tb = fg.pose_variables[a][i].transformation_matrix, fg.pose_variables[b][j].transformation_matrix
    rel = np.linalg.inv(Ta) @ Tb
    return compat.PoseMeasurement2D(fg.pose_variables[a][i].name, fg.pose_variables[b][j].name, float(rel[0, 2] + 0.01 * rng.standard_normal()),
                                    float(rel[1, 2] + 0.01 * rng.standard_normal()), float(np.arctan2(rel[1, 0], rel[0, 0]) + 0.002 * rng.standard_normal()), 1e4, 2.5e5)


def test_link_plan_host_logic(twin_lib):
    """Host side of round 6's loop closures inside the Newton preconditioner (csrc/score_link.hpp: find_link_pairs_P,
    make_link_plan -- compiled into the twin for this test; the kernels are GPU-tested): which node pairs are links (found in P's
    pattern: a rotation row holding a non-neighbour chain column), how the unknowns split into independent groups, what the caps
    refuse.  [pairs, pairs inside, unknowns, affected chains, rounds, -, groups, largest group, most chains per group]"""
    from score_amd.native import assemble_native

    def plan(fg):
        sol = ConicSolver([assemble_native(fg, "SOCP", lib_path=twin_lib).qp], dict(polish=0), lib_path=twin_lib)
        v, pairs = sol.debug_get("links"), sol.debug_get("link_pairs")
        sol.close()
        return [int(x) for x in v], pairs.astype(np.int64).reshape(-1, 2)

    rng = np.random.default_rng(5)
    fg = make_manhattan(n_robots=3, n_poses=150, n_beacons=3, seed=77, p_range=0.15)
    assert plan(fg)[0][:4] == [0, 0, 0, 0]  # no loop closures: nothing
    # between robots, onto the pinned pose (no unknown), chain neighbours (inside the chain), the same pair twice and reversed
    fg.loop_closure_measurements = [_lc_between(fg, rng, 0, 40, 1, 90), _lc_between(fg, rng, 2, 10, 1, 30), _lc_between(fg, rng, 0, 0, 2, 120),
                                    _lc_between(fg, rng, 1, 70, 1, 71), _lc_between(fg, rng, 0, 100, 0, 20), _lc_between(fg, rng, 0, 20, 0, 100),
                                    _lc_between(fg, rng, 2, 140, 0, 60)]
    v, pairs = plan(fg)
    # 4 links x 2 rows; nodes: A20 A40 A60 A100 B30 B90 C10 C140 per row = 8 nodes x 3 unknowns x 2 rows; every robot's chain is
    # affected in both rows (6 chains); A carries 4 nodes (12 rounds); the robots are all linked: one group per row, of 24
    assert v[:5] == [8, 8, 48, 6, 12] and v[6:] == [2, 24, 3], v
    n_rep = (3 * 150 - 1) * 3 + 3
    assert len(pairs) == 8 and np.all(pairs[:, 0] < pairs[:, 1]) and np.all(pairs[4:] == pairs[:4] + n_rep)  # row 1 = row 0 shifted by a replica
    assert [int(c) for c in pairs[:4, 0]] == sorted(int(c) for c in pairs[:4, 0])  # one canonical order whatever found them
    # loop closures within robots: the groups are per robot and row
    fg2 = make_manhattan(n_robots=4, n_poses=300, n_beacons=3, seed=78, n_loop_closures=20)
    v2, _ = plan(fg2)
    assert v2[0] == v2[1] > 20 and v2[6] >= 4 and v2[7] <= 96 and v2[8] == 1 and v2[2] == 3 * 2 * len({m.base_pose for m in fg2.loop_closure_measurements} | {m.to_pose for m in fg2.loop_closure_measurements} - {"A0"}) , v2
    # beyond the caps (more than 48 unknowns on one chain): all of the problem's pairs or none
    fg3 = make_manhattan(n_robots=1, n_poses=400, n_beacons=2, seed=79, n_loop_closures=12)
    v3, p3 = plan(fg3)
    assert v3[0] > 12 and v3[1] == 0 and v3[2] == 0 and len(p3) == 0, v3
    # 3-D: 4 x 4 blocks, three rows
    from conftest import graph_3d

    v4, _ = plan(graph_3d())
    assert v4[:5] == [3, 3, 24, 3, 8] and v4[6:] == [3, 8, 1], v4
