"""Wire / disk formats around the path and the intermediate-iterate API (f1), on the CPU.

Fixtures (DATA files of the reference, byte for byte): tests/golden/goats_14_6_2002_15_20.pkl -- the
PyFactorGraph pickle the reference's example loads (examples/solve_goats_example_score.py:18,40) -- and
tests/golden/gt_traj_A.tum, the ground-truth trajectory beside it (TUM rows ``t x y z qx qy qz qw``)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, compare_with_golden, graph_by_name, load_golden
from score_amd.io import load_fg_npz, load_pyfg_pickle, load_tum, save_to_tum
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_problem_with_intermediate_iterates, solve_score

PKL = os.path.join(GOLDEN, "goats_14_6_2002_15_20.pkl")
TUM = os.path.join(GOLDEN, "gt_traj_A.tum")


def test_pickle_ingest_equals_the_array_fixture():
    """load_pyfg_pickle reads the reference's pickle WITHOUT the py_factor_graph package (restricted
    unpickler, stub classes) and yields the graph the npz fixture encodes, measurement by measurement."""
    a, b = load_pyfg_pickle(PKL), load_fg_npz(os.path.join(GOLDEN, "goats_fg.npz"))
    assert a.dimension == b.dimension == 2
    assert a.get_pose_chain_names() == b.get_pose_chain_names()
    assert [l.name for l in a.landmark_variables] == [l.name for l in b.landmark_variables]
    assert (a.num_poses, a.num_landmarks, len(a.range_measurements)) == (679, 4, 1558)
    for ca, cb in zip(a.odom_measurements, b.odom_measurements):
        assert len(ca) == len(cb)
        for ma, mb in zip(ca, cb):
            assert (ma.base_pose, ma.to_pose) == (mb.base_pose, mb.to_pose)
            va = (ma.x, ma.y, ma.theta, ma.translation_precision, ma.rotation_precision)
            assert va == (mb.x, mb.y, mb.theta, mb.translation_precision, mb.rotation_precision)
    for ra, rb in zip(a.range_measurements, b.range_measurements):
        assert ra.association == rb.association and ra.dist == rb.dist and ra.stddev == rb.stddev
    assert a.range_measurements[0].precision == pytest.approx(1.0 / 0.75 ** 2)  # derived property
    assert len(a.pose_priors) == len(b.pose_priors)
    assert a.unconnected_variable_names == []


def test_pickle_ingest_refuses_foreign_globals(tmp_path):
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    p = tmp_path / "evil.pkl"
    p.write_bytes(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError, match="not allowed"):
        load_pyfg_pickle(str(p))


def test_tum_format_and_round_trip(twin_lib, tmp_path):
    """The reference's ground-truth file parses as N x 8 (t x y z qx qy qz qw, planar: z = qx = qy = 0,
    unit quaternions); save_to_tum writes the same layout, one file per pose chain, and load_tum reads
    back exactly the poses of the result."""
    gt = load_tum(TUM)
    assert gt.shape == (679, 8)
    assert np.all(gt[:, 3] == 0) and np.all(gt[:, 4] == 0) and np.all(gt[:, 5] == 0)
    np.testing.assert_allclose(np.linalg.norm(gt[:, 4:], axis=1), 1.0, atol=1e-12)
    assert np.array_equal(gt[:, 0], np.arange(679))
    fg = make_manhattan(n_robots=2, n_poses=25, n_beacons=2, seed=9, p_range=0.4)
    res = solve_score(fg, "SOCP", lib_path=twin_lib)
    files = save_to_tum(res, str(tmp_path / "est"))
    assert [os.path.basename(f) for f in files] == ["est_A.tum", "est_B.tum"]
    for f, chain in zip(files, res.pose_chain_names):
        with open(f) as fh:
            lines = fh.read().strip().splitlines()
        assert len(lines) == len(chain) and all(len(l.split()) == 8 for l in lines)
        arr = load_tum(f)
        for row, name in zip(arr, chain):
            T = res.poses[name]
            np.testing.assert_allclose(row[1:3], T[:2, 2], atol=1e-8)
            assert row[3] == 0.0 and row[4] == 0.0 and row[5] == 0.0
            th = 2.0 * np.arctan2(row[6], row[7])
            np.testing.assert_allclose([np.cos(th), np.sin(th)], [T[0, 0], T[1, 0]], atol=1e-8)


def test_goats_pickle_through_solve_score_and_tum(twin_lib, tmp_path):
    """BASELINE configs[0] end to end on the CPU twin: pickle in, trajectory file out, same rows as the
    ground truth file has."""
    fg = load_pyfg_pickle(PKL)
    res = solve_score(fg, lib_path=twin_lib)  # reference default relaxation (QCQP)
    gold = load_golden("goats")
    assert res.solved and res.info["pobj"] == pytest.approx(float(gold["objective"]), rel=1e-6)
    compare_with_golden(res, gold)
    (f,) = save_to_tum(res, str(tmp_path / "goats"))
    est, gt = load_tum(f), load_tum(TUM)
    assert est.shape == gt.shape


def test_intermediate_iterates_follow_the_solver(twin_lib):
    """solve_score.py:89-116.  On the CPU twin (no polish) the trajectory is ADMM only: the warm-up
    snapshots every `every` iterations, then every `every` more; `solved` is the solver's own status and
    the list ends with the first solved iterate, which is the optimum."""
    fg = graph_by_name("synth_a", {})
    its = solve_problem_with_intermediate_iterates(fg, "SOCP", every=3, lib_path=twin_lib)
    counts = [r.info["iters"] for r in its]
    # (polish_warmup = 6 by default: two warm-up snapshots, then ADMM goes on in steps of `every`)
    assert counts == sorted(counts) and counts[:3] == [3, 6, 9] and len(set(counts)) == len(counts)
    assert [r.solved for r in its[:-1]] == [False] * (len(its) - 1) and its[-1].solved
    assert all(r.info["status"] in (1, 2) for r in its)
    res_p = [r.info["res_pri"] for r in its]
    assert res_p[-1] < 1e-3 * max(res_p)
    compare_with_golden(its[-1], load_golden("synth_a"))
    ref = solve_score(fg, "SOCP", lib_path=twin_lib)
    # the same iteration, paused: the first solved snapshot comes no later than solve()'s own check
    # (solve() tests every 25 iterations, the snapshots every 3)
    assert ref.solved and its[-1].info["iters"] <= ref.info["iters"] + 3
