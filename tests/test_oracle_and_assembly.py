"""CPU tests: the oracle against its anchors, and the product assembler against
the oracle's literal restatement of score/utils/gurobi_utils.py."""
import numpy as np
import pytest

from conftest import GOLDEN_NAMES, SYNTH, graph_by_name, load_golden
from oracle import score_oracle as so
from score_amd import compat
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan


def _random_values(mdl, rng):
    """A random model-space point that satisfies the pin, as named containers."""
    d = mdl.dim
    xs = rng.normal(size=mdl.qp.n)
    xm = mdl.expand(xs)
    vals = {
        "poses": {nm: mdl.pose_blocks(xm)[i] for i, nm in enumerate(mdl.pose_names)},
        "landmarks": {nm: mdl.landmark_block(xm)[i] for i, nm in enumerate(mdl.landmark_names)},
        "dists": {k: mdl.range_block(xm)[i] for i, k in enumerate(mdl.range_keys)},
    }
    return xs, vals


from conftest import graph_3d as _graph_3d  # noqa: E402  (shared with tests/golden/make_fixture_golden.py)


@pytest.mark.parametrize("relax", ["SOCP", "QCQP"])
@pytest.mark.parametrize("name", ["manhattan", "goats", "synth_b", "3d"])
def test_assembler_matches_literal_model(name, relax, fixtures):
    fg = _graph_3d() if name == "3d" else graph_by_name(name, fixtures)
    mdl = assemble(fg, relax)
    lit = so.LiteralModel(fg, relax)
    rng = np.random.default_rng(1)
    for _ in range(3):
        xs, vals = _random_values(mdl, rng)
        direct = lit.direct_cost(vals)
        assert lit.pin_violation(vals) == 0.0
        assert mdl.qp.objective(xs) == pytest.approx(direct, rel=1e-11)
        # cone rows: s = b - A x must be (d_ij, t_i - t_j) resp. (1, r_ij)
        s = (mdl.qp.b - mdl.qp.A @ xs).reshape(-1, mdl.dim + 1)
        for r, key in enumerate(mdl.range_keys[:50]):
            if relax == "SOCP":
                diff = lit.translation(vals, key[0]) - lit.translation(vals, key[1])
                np.testing.assert_allclose(s[r], np.concatenate([vals["dists"][key], diff]), atol=1e-12)
            else:
                np.testing.assert_allclose(s[r], np.concatenate([[1.0], vals["dists"][key]]), atol=1e-12)


def test_fixture_statistics(fixtures):
    """Counts of the reference's shipped data files."""
    m, g = fixtures["manhattan"], fixtures["goats"]
    assert (m.num_poses, m.num_landmarks, len(m.range_measurements)) == (1600, 6, 1160)
    assert sum(len(c) for c in m.odom_measurements) == 1596 and len(m.pose_priors) == 1
    assert (g.num_poses, g.num_landmarks, len(g.range_measurements)) == (679, 4, 1558)
    assert g.range_measurements[0].precision == pytest.approx(1.0 / 0.75 ** 2)
    ms = assemble(m, "SOCP").qp
    assert (ms.n, ms.m) == (10772 - 6, 3 * 1160)  # pinned pose eliminated; lb rows implied by the cone
    assert assemble(m, "QCQP").qp.n == 11932 - 6
    assert assemble(g, "SOCP").qp.n == 5640 - 6 and assemble(g, "QCQP").qp.n == 7198 - 6


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_newton_oracle_reproduces_golden(name, fixtures):
    fg = graph_by_name(name, fixtures)
    gold = load_golden(name)
    rp, u, info = so.newton_solve(fg, tol=1e-13, max_iter=300)
    vals = so.reduced_to_values(rp, u, "SOCP")
    obj = so.LiteralModel(fg, "SOCP").direct_cost(vals)
    assert obj == pytest.approx(float(gold["objective"]), rel=1e-8, abs=1e-9)
    P = np.stack([vals["poses"][str(n)] for n in gold["pose_names"]])
    det = gold["pose_determined"]
    np.testing.assert_allclose(P[det], gold["poses"][det], atol=1e-6 * max(1.0, np.abs(gold["poses"]).max()))
    # the masks come from the oracle alone: null space of the generalised Hessian at the optimum
    pm, lm, ninfo = so.determined_masks(rp, u)
    assert np.array_equal(pm, gold["pose_determined"]) and np.array_equal(lm, gold["landmark_determined"])
    res, ex = so.optimal_residuals(rp, u)
    np.testing.assert_allclose(res, gold["quad_residuals"], atol=1e-7)
    np.testing.assert_allclose(ex, gold["range_excess"], atol=1e-7)


def test_determined_masks_follow_the_null_space():
    """A direction the analysis calls free must leave the objective unchanged, and the variables it
    calls determined must not move along it; moving a determined variable must raise the objective."""
    fg = graph_by_name("synth_b", {})
    rp, u, _ = so.newton_solve(fg, tol=1e-13, max_iter=300)
    pm, lm, ninfo = so.determined_masks(rp, u)
    assert ninfo["null_dim"] > 0 and not pm.all() and pm[: len(fg.pose_variables[0])].all()
    _, H = rp.grad_hess(u)
    N = so.gauge_basis(rp)
    d = rp.dim
    n_odo = sum(len(c) for c in fg.odom_measurements) * (d + d * d)  # odometry rows come first in J (loop closures follow)
    assert np.abs((rp.J @ N)[:n_odo]).max() < 1e-9  # odometry residuals are blind to the gauge directions
    G = N.T @ (H @ N)
    w, V = np.linalg.eigh(0.5 * (G + G.T))
    z = N @ V[:, 0]
    z /= np.abs(z).max()
    f0 = rp.value(u)
    assert abs(rp.value(u + 1e-3 * z) - f0) < 1e-9 * max(1.0, f0)  # flat along a null direction
    for i, nm in enumerate(rp.pose_names):
        if pm[i] and nm != rp.first_pose:
            assert np.abs(z[rp.col[nm] : rp.col[nm] + d * (d + 1)]).max() < 1e-6
    e = np.zeros(rp.n)
    e[rp.col[rp.pose_names[5]] + d] = 1.0  # translation of a determined pose of the pinned robot
    assert rp.value(u + 1e-3 * e) > f0 + 1e-6


def test_survey_anchor_values():
    """Survey-time anchors: 33.66585 and 330.487."""
    assert float(load_golden("manhattan")["objective"]) == pytest.approx(33.66585, abs=2e-5)
    assert float(load_golden("goats")["objective"]) == pytest.approx(330.487, abs=1e-3)


@pytest.mark.parametrize("name", ["manhattan", "synth_c"])
def test_oracle_optimum_certified_on_product_matrices(name, fixtures):
    """Solver-independent certificate: the Newton optimum, with the multipliers the
    reduced problem implies, satisfies the KKT conditions of the conic program the
    PRODUCT assembler built -- for both relaxations (they share poses/landmarks)."""
    fg = graph_by_name(name, fixtures)
    rp, u, _ = so.newton_solve(fg, tol=1e-13)
    for relax in ("SOCP", "QCQP"):
        mdl = assemble(fg, relax)
        vals = so.reduced_to_values(rp, u, relax)
        lit = so.LiteralModel(fg, relax)
        assert lit.cone_violation(vals) < 1e-9
        d = mdl.dim
        xm = np.zeros(mdl.n_model)
        for i, nm in enumerate(mdl.pose_names):
            xm[i * d * (d + 1) : (i + 1) * d * (d + 1)] = vals["poses"][nm].ravel()
        for i, nm in enumerate(mdl.landmark_names):
            xm[mdl.lm_base + i * d : mdl.lm_base + (i + 1) * d] = vals["landmarks"][nm]
        for i, k in enumerate(mdl.range_keys):
            xm[mdl.rng_base + i * mdl.rng_width : mdl.rng_base + (i + 1) * mdl.rng_width] = vals["dists"][k]
        x = mdl.reduce(xm)
        dl = rp.deltas(u)
        rho = np.linalg.norm(dl, axis=1)
        uh = np.where(rho[:, None] > 0, dl / np.maximum(rho, 1e-300)[:, None], 0.0)
        ex = np.maximum(0.0, rho - rp.dist)
        y = np.zeros((rp.nr, d + 1))
        if relax == "SOCP":  # y = 2w max(0,|D|-dist) (1, -u)
            y[:, 0] = 2 * rp.wr * ex
            y[:, 1:] = -(2 * rp.wr * ex)[:, None] * uh
        else:  # y_vec = -2 w dist (D - dist r), y_0 = |y_vec|
            rr = np.stack([vals["dists"][k] for k in mdl.range_keys])
            yv = -2 * (rp.wr * rp.dist)[:, None] * (dl - rp.dist[:, None] * rr)
            y[:, 1:] = yv
            y[:, 0] = np.linalg.norm(yv, axis=1)
        cert = so.kkt_certificate(mdl.qp.P, mdl.qp.q, mdl.qp.A, mdl.qp.b, 0, mdl.qp.soc_dims, x, y.ravel())
        scale = max(1.0, np.abs(mdl.qp.q).max())
        assert cert["dual_res_inf"] < 1e-9 * scale, cert
        assert cert["s_cone_dist"] < 1e-9 and cert["y_cone_dist"] < 1e-9, cert
        assert cert["complementarity"] < 1e-7 * scale, cert


def test_relaxations_share_poses(fixtures):
    """SOCP / QCQP equivalence: same reduced problem, so one Newton
    solve serves both; the literal costs of the two expansions agree."""
    fg = fixtures["manhattan"]
    rp, u, _ = so.newton_solve(fg, tol=1e-13)
    c_socp = so.LiteralModel(fg, "SOCP").direct_cost(so.reduced_to_values(rp, u, "SOCP"))
    c_qcqp = so.LiteralModel(fg, "QCQP").direct_cost(so.reduced_to_values(rp, u, "QCQP"))
    assert c_socp == pytest.approx(c_qcqp, rel=1e-10)


def test_input_validation(fixtures):
    fg = make_manhattan(n_robots=1, n_poses=5, n_beacons=1, seed=0, p_range=1.0)
    with pytest.raises(ValueError, match="not supported"):
        assemble(fg, "SDP")
    fg.dimension = 4
    with pytest.raises(ValueError, match="not 2 or 3"):
        assemble(fg, "SOCP")
    fg.dimension = 2
    fg.landmark_variables.append(compat.LandmarkVariable2D("A1"))
    with pytest.raises(ValueError, match="already exists"):
        assemble(fg, "SOCP")
    fg.landmark_variables.pop()
    fg.range_measurements.append(fg.range_measurements[0])
    with pytest.raises(ValueError, match="already exists in distance_vars"):
        assemble(fg, "SOCP")


def test_generator_statistics():
    """The synthetic generator reproduces the shipped fixture's statistics."""
    fg = make_manhattan(n_robots=4, n_poses=400, n_beacons=6, seed=3)
    assert fg.unconnected_variable_names == []
    n_pb = sum(1 for m in fg.range_measurements if m.second_key.startswith("L"))
    n_rr = len(fg.range_measurements) - n_pb
    assert 0.08 < n_pb / (4 * 400 * 6) < 0.12 and 0.07 < n_rr / (6 * 400) < 0.13
    th = np.array([m.theta for c in fg.odom_measurements for m in c])
    turn = np.mean(np.abs(th) > 0.5)
    assert 0.12 < turn < 0.30
    assert all(m.dist >= 0 for m in fg.range_measurements)
    p0 = fg.pose_variables[0][0]
    assert p0.true_position == (0.0, 0.0) and p0.true_theta == 0.0
    a = make_manhattan(n_robots=2, n_poses=30, n_beacons=2, seed=9)
    b = make_manhattan(n_robots=2, n_poses=30, n_beacons=2, seed=9)
    assert [m.dist for m in a.range_measurements] == [m.dist for m in b.range_measurements]


@pytest.mark.parametrize("relax", ["SOCP", "QCQP"])
@pytest.mark.parametrize("name", ["manhattan", "goats", "synth_b", "graph3d"])
def test_native_assembler_equals_the_python_one(name, relax, fixtures, twin_lib):
    """score_assemble (C++, behind the C ABI) builds the same conic program as score_amd/assemble.py --
    same column layout, P / q / c0 / A / b / cones / chain hint -- on the reference's two data sets, a
    graph with loop closures and a 3-D graph with a landmark prior; and hence the same literal objective
    (gurobi_utils.py:358-526) as the oracle."""
    from score_amd.native import assemble_native

    fg = graph_by_name(name, fixtures)
    a, b = assemble(fg, relax), assemble_native(fg, relax, lib_path=twin_lib)
    qa, qb = a.qp, b.qp
    assert (qa.n, qa.m, qa.z, qa.block_size) == (qb.n, qb.m, qb.z, qb.block_size)
    scale = abs(qa.P).max()
    assert abs(qa.P - qb.P).max() <= 1e-13 * scale
    np.testing.assert_allclose(qb.q, qa.q, rtol=0, atol=1e-13 * max(1.0, np.abs(qa.q).max()))
    assert qb.c0 == pytest.approx(qa.c0, rel=1e-12, abs=1e-9)
    if qa.m:
        assert abs(qa.A - qb.A).max() == 0.0 and np.array_equal(qa.b, qb.b)
        assert qb.A.has_sorted_indices and qb.P.has_sorted_indices
    assert np.array_equal(qa.soc_dims, qb.soc_dims)
    assert np.array_equal(qa.chain_ptr, qb.chain_ptr) and np.array_equal(qa.node_cols, qb.node_cols)
    assert np.array_equal(a.free_cols, b.free_cols) and a.n_model == b.n_model
    assert a.pose_names == b.pose_names and a.range_keys == b.range_keys
    if a.range_ends is not None:
        assert np.array_equal(a.range_ends, b.range_ends)
    rng = np.random.default_rng(3)
    xs, vals = _random_values(b, rng)
    assert qb.objective(xs) == pytest.approx(so.LiteralModel(fg, relax).direct_cost(vals), rel=1e-11)


def test_native_assembler_errors(twin_lib):
    from score_amd import compat
    from score_amd.manhattan import make_manhattan
    from score_amd.native import assemble_native

    fg = make_manhattan(n_robots=1, n_poses=5, n_beacons=1, seed=1, p_range=1.0)
    with pytest.raises(ValueError, match="not supported"):
        assemble_native(fg, "LP", lib_path=twin_lib)
    fg.range_measurements.append(compat.FGRangeMeasurement(("A1", "nope"), 1.0, 1.0))
    with pytest.raises(ValueError, match="Variable name nope not found"):
        assemble_native(fg, "SOCP", lib_path=twin_lib)
    fg.range_measurements.pop()
    fg.pose_variables[0].append(compat.PoseVariable2D("A1", (0.0, 0.0), 0.0))
    with pytest.raises(ValueError, match="already exists in pose_vars"):
        assemble_native(fg, "SOCP", lib_path=twin_lib)


def test_one_pass_object_reader_equals_the_attribute_passes(fixtures, monkeypatch):
    """score_amd/csrc/_objread.c (one pass over every measurement list, plain attributes straight from the instance dict)
    yields the same flat arrays as one numpy.fromiter pass per attribute -- on the reference's data sets, a 3-D graph, a
    graph whose measurement classes answer through properties only, and with the same errors for unknown names."""
    import score_amd.native as nat
    from score_amd import compat
    from score_amd.manhattan import make_manhattan

    assert nat._objread is not None, "score_amd/_objread*.so is not built (python -c 'import __graft_entry__ as g; g.build()')"

    class PropRange:  # first_key / second_key / precision only, association as a list, values behind properties
        def __init__(self, m):
            self._m = m
        association = property(lambda self: list(self._m.association))
        dist = property(lambda self: self._m.dist)
        stddev = property(lambda self: self._m.stddev)

    fgs = [graph_by_name(nm, fixtures) for nm in ("goats", "synth_b", "graph3d", "prior2d")] + [make_manhattan(n_robots=3, n_poses=60, n_beacons=2, seed=5)]
    wrapped = make_manhattan(n_robots=2, n_poses=30, n_beacons=2, seed=6)
    wrapped.range_measurements = [PropRange(m) for m in wrapped.range_measurements]
    fgs.append(wrapped)
    for fg in fgs:
        fast = nat.graph_arrays(fg)
        with monkeypatch.context() as mp:
            mp.setattr(nat, "_objread", None)
            slow = nat.graph_arrays(fg)
        assert fast.keys() == slow.keys()
        for k in fast:
            if isinstance(fast[k], np.ndarray):
                assert fast[k].dtype == slow[k].dtype and np.array_equal(fast[k], slow[k]), k
            else:
                assert fast[k] == slow[k], k
        assert all(isinstance(k, tuple) for k in fast["range_keys"])
    fg = make_manhattan(n_robots=1, n_poses=5, n_beacons=1, seed=1, p_range=1.0)
    fg.range_measurements.append(compat.FGRangeMeasurement(("A1", "nope"), 1.0, 1.0))
    with pytest.raises(ValueError, match="Variable name nope not found"):
        nat.graph_arrays(fg)
    # a duplicate key anywhere in the list is reported before an unknown endpoint, with the helper and without it
    # (gurobi_utils.py:62-67 raises while the distance variables are added, :103-109 only when the costs are built)
    fg.range_measurements.append(fg.range_measurements[0])
    with pytest.raises(ValueError, match="already exists in distance_vars"):
        nat.graph_arrays(fg)
    with monkeypatch.context() as mp:
        mp.setattr(nat, "_objread", None)
        with pytest.raises(ValueError, match="already exists in distance_vars"):
            nat.graph_arrays(fg)
    fg.range_measurements.pop()
    fg.range_measurements.pop()
    fg.odom_measurements[0][0].to_pose = "nowhere"
    with pytest.raises(KeyError, match="nowhere"):
        nat.graph_arrays(fg)
    fg.odom_measurements[0][0].to_pose = "A1"
    fg.odom_measurements[0][0].x = "not a number"
    with pytest.raises(TypeError):
        nat.graph_arrays(fg)


def test_batch_assembler_equals_graph_by_graph_calls(fixtures, twin_lib):
    """score_assemble_batch (one foreign call, one graph per host thread of the library) returns, graph for graph, what
    score_assemble returns: mixed sizes and dimensions, SOCP and QCQP; a bad graph fails the call, names its index and
    leaves nothing behind."""
    from score_amd.manhattan import make_manhattan
    from score_amd.native import assemble_native, assemble_native_batch, graph_arrays

    fgs = [graph_by_name(nm, fixtures) for nm in ("synth_a", "synth_b", "graph3d", "prior2d")]
    fgs += [make_manhattan(n_robots=1 + t % 3, n_poses=40 + 17 * t, n_beacons=2, seed=70 + t) for t in range(9)]
    for relax in ("SOCP", "QCQP"):
        arrs = [graph_arrays(fg) for fg in fgs]
        got = assemble_native_batch(arrs, relax, lib_path=twin_lib)
        assert len(got) == len(fgs)
        for fg, a, m in zip(fgs, arrs, got):
            ref = assemble_native(fg, relax, lib_path=twin_lib, arrays=a)
            qa, qb = ref.qp, m.qp
            assert (qa.n, qa.m, qa.block_size, qa.rep_d, qa.rep_n, qa.c0) == (qb.n, qb.m, qb.block_size, qb.rep_d, qb.rep_n, qb.c0)
            for x, y in ((qa.P, qb.P), (qa.A, qb.A)):
                assert np.array_equal(x.indptr, y.indptr) and np.array_equal(x.indices, y.indices) and np.array_equal(x.data, y.data)
            assert np.array_equal(qa.q, qb.q) and np.array_equal(qa.b, qb.b)
            assert np.array_equal(qa.chain_ptr, qb.chain_ptr) and np.array_equal(qa.node_cols, qb.node_cols)
            assert np.array_equal(ref.free_cols, m.free_cols) and ref.n_model == m.n_model
    assert assemble_native_batch([], "SOCP", lib_path=twin_lib) == []
    bad = [graph_arrays(fg) for fg in fgs[:3]]
    bad[1] = dict(bad[1], rng_a=bad[1]["rng_a"] + 10**6)
    with pytest.raises(ValueError, match="graph 1: .*out of range"):
        assemble_native_batch(bad, "SOCP", lib_path=twin_lib)


def test_headline_config_golden_is_an_optimum_of_the_literal_model():
    """tests/golden/config3_golden.npz (BASELINE configs[3], 20 robots x 1000 poses; produced offline by the oracle's
    Newton method): the stored estimate belongs to the graph make_config(3) regenerates, satisfies the pin, evaluates
    to the stored objective under the reference's objective taken literally (gurobi_utils.py:358-526), and the
    gradient of the reduced problem vanishes there -- checked without running the solve again."""
    from score_amd.manhattan import make_config

    gold = load_golden("config3")
    fg = make_config(3)
    names = [p.name for chain in fg.pose_variables for p in chain]
    assert names == [str(n) for n in gold["pose_names"]]
    assert [l.name for l in fg.landmark_variables] == [str(n) for n in gold["landmark_names"]]
    d = fg.dimension
    P, L = gold["poses"], gold["landmarks"]
    assert P.shape == (20000, d, d + 1)
    assert np.array_equal(P[0], np.hstack([np.eye(d), np.zeros((d, 1))]))  # pinned pose (gurobi_utils.py:316-333)
    rp = so.ReducedProblem(fg)
    u = np.zeros(rp.n)
    for i, nm in enumerate(names):
        if nm != rp.first_pose:
            u[rp.col[nm] : rp.col[nm] + d * (d + 1)] = P[i].ravel()
    for i, l in enumerate(fg.landmark_variables):
        u[rp.col[l.name] : rp.col[l.name] + d] = L[i]
    g, _ = rp.grad_hess(u)
    assert np.abs(g).max() <= 1e-8 * max(1.0, np.abs(rp.g0).max())
    vals = so.reduced_to_values(rp, u, "SOCP")
    lit = so.LiteralModel(fg, "SOCP")
    assert lit.pin_violation(vals) == 0.0 and lit.cone_violation(vals) <= 1e-12
    assert lit.direct_cost(vals) == pytest.approx(float(gold["objective"]), rel=1e-12)
    res, ex = so.optimal_residuals(rp, u)
    np.testing.assert_allclose(ex, gold["range_excess"], atol=1e-12)
    assert int((ex > 1e-9).sum()) == 588
