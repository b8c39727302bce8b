"""The batched generator of synthetic Manhattan-world graphs (csrc/score_generate.hpp).

CPU: the host loops of the CPU twin -- the generator's specification -- against the statistics SURVEY 8(d) measured from
the reference's shipped fixture (examples/manhattan/factor_graph.pickle: 4 x 400 poses, 6 beacons, 1 160 ranges), the
structural rules of the walk, and the solver.  GPU: the device kernels against those host loops (integers bit for bit,
reals to the rounding of the math library) and through the product path."""
import ctypes as C

import numpy as np
import pytest

from score_amd.generate import GeneratedBatch, ManhattanSpec, generate_manhattan
from score_amd.solve_score import solve_score, solve_score_batch


def test_generated_worlds_have_the_fixtures_statistics(twin_lib):
    R, T, Nb, side, p = 4, 400, 6, 20, 0.10
    count = 24
    B = GeneratedBatch(count, seed=7000, n_robots=R, n_poses=T, n_beacons=Nb, side=side, p_range=p, lib_path=twin_lib)
    dturn, n_rb, n_rr, odo_t, odo_th, rng_err = [], 0, 0, [], [], []
    for i in range(count):
        a = B.arrays(i)
        poses, beacons = B.truth(i)
        xy = poses[:, :2].reshape(R, T, 2)
        hd = (np.round(poses[:, 2] / (np.pi / 2)).astype(int) % 4).reshape(R, T)
        assert np.array_equal(xy[0, 0], [0.0, 0.0]) and hd[0, 0] == 0  # the pinned pose
        assert xy.min() >= 0 and xy.max() <= side and beacons.min() >= 0 and beacons.max() <= side
        step = xy[:, 1:] - xy[:, :-1]
        dirs = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]], dtype=float)
        assert np.array_equal(step, dirs[hd[:, :-1]])  # unit steps along the heading of the pose they leave
        dturn.append(((hd[:, 1:] - hd[:, :-1]) % 4).ravel())
        # measurements: odometry chain by chain, indices as score_graph wants them
        assert np.array_equal(a["rel_base"], np.concatenate([r * T + np.arange(T - 1) for r in range(R)]))
        assert np.array_equal(a["rel_to"], a["rel_base"] + 1)
        assert np.all(a["rel_kappa"] == 1e4) and np.allclose(a["rel_tau"], 2.5e5)
        th = np.arctan2(a["rel_R"][:, 1, 0], a["rel_R"][:, 0, 0])
        true_dth = (((hd[:, 1:] - hd[:, :-1] + 1) % 4 - 1) * (np.pi / 2)).ravel()
        odo_th.append(np.arctan2(np.sin(th - true_dth), np.cos(th - true_dth)))
        odo_t.append(a["rel_t"] - np.array([1.0, 0.0]))
        np.testing.assert_allclose(a["rel_R"][:, 0, 0], a["rel_R"][:, 1, 1]); np.testing.assert_allclose(a["rel_R"][:, 0, 1], -a["rel_R"][:, 1, 0])
        # ranges: robot by robot against the beacons (time-major), then the robot pairs at equal timesteps
        ra, rb = a["rng_a"], a["rng_b"]
        is_rb = rb >= R * T
        n_rb += int(is_rb.sum()); n_rr += int((~is_rb).sum())
        assert np.all(np.diff(is_rb.astype(int)) <= 0)  # all robot-beacon measurements first
        assert np.all(ra[~is_rb] % T == rb[~is_rb] % T) and np.all(ra[~is_rb] // T < rb[~is_rb] // T)
        pts = np.concatenate([xy.reshape(-1, 2), beacons])
        true = np.linalg.norm(pts[ra] - pts[rb], axis=1)
        assert np.all(a["rng_dist"] >= 0.0) and np.all(a["rng_prec"] == 1.0)
        keep = a["rng_dist"] > 0.0
        rng_err.append((a["rng_dist"] - true)[keep & (true > 4.0)])  # (away from the clamp)
        assert len({tuple(k) for k in a["range_keys"]}) == len(ra)  # no duplicate keys
    d = np.concatenate(dturn)
    left_right, back = ((d == 1) | (d == 3)).mean(), (d == 2).mean()
    assert 0.17 <= left_right <= 0.22 and 0.005 <= back <= 0.03, (left_right, back)  # ~18 % turns, ~1 % U-turns (+ the walls)
    exp_rb, exp_rr = count * R * T * Nb * p, count * (R * (R - 1) // 2) * T * p
    assert abs(n_rb - exp_rb) < 4 * np.sqrt(exp_rb) and abs(n_rr - exp_rr) < 4 * np.sqrt(exp_rr), (n_rb, exp_rb, n_rr, exp_rr)
    ot, oth, re_ = np.concatenate(odo_t), np.concatenate(odo_th), np.concatenate(rng_err)
    assert abs(ot.std() - 0.01) < 3e-4 and abs(ot.mean()) < 2e-4 and abs(oth.std() - 0.002) < 6e-5
    assert abs(re_.std() - 1.0) < 0.02 and abs(re_.mean()) < 0.02
    # ... and against the reference's own simulation file (tests/golden/manhattan_fg.npz = examples/manhattan/factor_graph.pickle:
    # 4 x 400 poses, 6 beacons, the shape generated above)
    import os

    from score_amd.io import load_fg_npz

    fx = load_fg_npz(os.path.join(os.path.dirname(__file__), "golden", "manhattan_fg.npz"))
    od = [m_ for ch in fx.odom_measurements for m_ in ch]
    fth = np.array([m_.theta for m_ in od])
    fk = np.round(fth / (np.pi / 2)).astype(int) % 4
    f_lr, f_back = ((fk == 1) | (fk == 3)).mean(), (fk == 2).mean()
    assert abs(left_right - f_lr) < 0.03 and abs(back - f_back) < 0.02, (left_right, f_lr, back, f_back)
    assert abs(oth.std() - (fth - np.round(fth / (np.pi / 2)) * (np.pi / 2)).std()) < 1.5e-4
    assert od[0].translation_precision == 1e4 and od[0].rotation_precision == pytest.approx(2.5e5)
    lm = {v.name for v in fx.landmark_variables}
    f_rb = sum(1 for m_ in fx.range_measurements if m_.first_key in lm or m_.second_key in lm)
    f_rr = len(fx.range_measurements) - f_rb
    assert abs(n_rb / count - f_rb) < 0.08 * f_rb and abs(n_rr / count - f_rr) < 0.15 * f_rr, (n_rb / count, f_rb, n_rr / count, f_rr)
    assert fx.range_measurements[0].precision == 1.0
    # world t of a batch is the world of seed + t, whatever the batch
    one = GeneratedBatch(1, seed=7005, n_robots=R, n_poses=T, n_beacons=Nb, side=side, p_range=p, lib_path=twin_lib)
    a5, b0 = B.arrays(5), one.arrays(0)
    for k in ("rel_t", "rel_R", "rng_a", "rng_b", "rng_dist"):
        assert np.array_equal(a5[k], b0[k])
    assert not np.array_equal(B.arrays(4)["rng_dist"][:50], a5["rng_dist"][:50])


def test_generator_against_the_independent_restatement(twin_lib):
    """oracle/generate_oracle.py restates the generator in plain Python (its own Philox4x32-10, pinned here by the known-answer
    vectors of Random123; walk, measurements and their order written from the description, not from the C++): the library's
    host loops -- which the device kernels are tested against -- must draw the same worlds: every integer equal, reals to 1e-12."""
    from oracle.generate_oracle import philox4x32_10, world

    kat = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kat:  # Random123 kat_vectors, philox4x32 10 rounds
        assert " ".join("%08x" % v for v in philox4x32_10(key[0] | (key[1] << 32), *ctr)) == want
    spec = dict(n_robots=3, n_poses=60, n_beacons=5, side=7, p_range=0.3, sigma_t=0.02, sigma_theta=0.01, sigma_range=0.7)
    B = GeneratedBatch(3, seed=123456789012, lib_path=twin_lib, **spec)
    for t in range(3):
        w = world(123456789012, t, **spec)
        a = B.arrays(t)
        poses, beacons = B.truth(t)
        R, T = spec["n_robots"], spec["n_poses"]
        assert np.array_equal(poses[:, :2], np.array([p for r in range(R) for p in w["pos"][r]], dtype=float))
        hd = np.array([h for r in range(R) for h in w["hd"][r]])
        np.testing.assert_allclose(poses[:, 2], np.arctan2(np.sin(hd * np.pi / 2), np.cos(hd * np.pi / 2)), atol=1e-15)
        assert np.array_equal(beacons, np.array(w["beacons"], dtype=float))
        od = np.array(w["odom"])
        assert np.array_equal(a["rel_base"], od[:, 0].astype(np.int32)) and np.array_equal(a["rel_to"], od[:, 1].astype(np.int32))
        np.testing.assert_allclose(a["rel_t"], od[:, 2:4], rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.arctan2(a["rel_R"][:, 1, 0], a["rel_R"][:, 0, 0]), od[:, 4], rtol=0, atol=1e-12)
        rg = np.array(w["ranges"])
        assert len(rg) == len(a["rng_a"]) > 50
        assert np.array_equal(a["rng_a"], rg[:, 0].astype(np.int32)) and np.array_equal(a["rng_b"], rg[:, 1].astype(np.int32))
        np.testing.assert_allclose(a["rng_dist"], rg[:, 2], rtol=0, atol=1e-12)
        assert np.all(a["rng_prec"] == pytest.approx(1.0 / 0.49)) and np.all(a["rel_kappa"] == pytest.approx(2500.0))


def test_generated_worlds_solve_and_errors(twin_lib):
    graphs = generate_manhattan(3, seed=11, n_robots=2, n_poses=60, n_beacons=3, p_range=0.4, lib_path=twin_lib)
    B = graphs[0].arrays["_owner"]
    res = solve_score_batch(graphs, "SOCP", lib_path=twin_lib)
    assert all(r.solved for r in res)
    poses, _ = B.truth(1)
    est = np.array([res[1].poses[f"A{i}"][:2, 2] for i in range(60)])
    assert np.abs(est - poses[:60, :2]).max() < 1.0  # the pinned robot's trajectory: odometry noise of 1 cm per step
    assert res[1].pose_chain_names[1][3] == "B3" and set(res[1].landmarks) == {"L0", "L1", "L2"}
    # ground truth as results (TUM export of the reference trajectory) and trajectory errors against it
    from score_amd.io import load_tum, save_to_tum

    gt = B.truth_results(1)
    assert np.array_equal(gt.poses["A0"], np.eye(3)) and list(gt.landmarks) == ["L0", "L1", "L2"]
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        files = save_to_tum(gt, tmp + "/gt")
        assert len(files) == 2 and load_tum(files[0]).shape == (60, 8)
        np.testing.assert_allclose(load_tum(files[1])[:, 1:3], poses[60:, :2])
    err = B.trajectory_errors(1, res[1])
    assert set(err) == {"A", "B"} and err["A"]["translation_rmse"] < 0.5 and err["A"]["heading_max"] < 0.2
    assert B.trajectory_errors(1, gt)["B"]["translation_max"] == 0.0
    rq = solve_score(graphs[2], lib_path=twin_lib)  # the reference's default relaxation
    assert rq.solved and len(rq.distances) == graphs[2].num_ranges
    with pytest.raises(ValueError, match="robots"):
        GeneratedBatch(1, n_robots=0, lib_path=twin_lib)
    with pytest.raises(ValueError, match="p_range"):
        GeneratedBatch(1, p_range=1.5, lib_path=twin_lib)
    with pytest.raises(ValueError, match="2\\^30 possible range measurements"):  # (32-bit range offsets: refused, not wrapped)
        GeneratedBatch(16, n_robots=64, n_poses=1000, n_beacons=4096, lib_path=twin_lib)
    with pytest.raises(IndexError):
        B.arrays(3)
    assert C.sizeof(ManhattanSpec) == 64  # (ABI v7: + dim, reserved)
    with pytest.raises(ValueError, match="dimension must be 2 or 3"):
        GeneratedBatch(1, dim=4, lib_path=twin_lib)
    # score_create_from_generated: argument errors come back as errors, not crashes
    from score_amd.solver import ScoreSettings

    lib = B.lib
    lib.score_create_from_generated.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    for first, count, relax, what in ((-1, 1, 0, b"range"), (2, 2, 0, b"range"), (0, 0, 0, b"range"), (0, 1, 2, b"relaxation")):
        assert lib.score_create_from_generated(B._h, first, count, relax, None, C.byref(h)) != 0
        assert what in lib.score_last_error()
    assert lib.score_create_from_generated(None, 0, 1, 0, None, C.byref(h)) != 0
    assert lib.score_create_from_generated(B._h, 1, 2, 1, None, C.byref(h)) == 0  # worlds 1-2, the QCQP form
    n, m, cnt = C.c_int64(), C.c_int64(), C.c_int32()
    lib.score_dims(h, C.byref(n), C.byref(m), C.byref(cnt))
    a1, a2 = B.arrays(1), B.arrays(2)
    assert cnt.value == 2 and m.value == 3 * (len(a1["rng_a"]) + len(a2["rng_a"]))
    assert n.value == sum(2 * ((2 * 60 - 1) * 3 + 3 + len(a["rng_a"])) for a in (a1, a2))  # the QCQP program's columns
    lib.score_destroy(h)


def test_3d_generator_against_the_independent_restatement(twin_lib):
    """The generator draws 3-D worlds too (round 6; the reference's model is dimension-generic, gurobi_utils.py:37-50): walks on the
    lattice of a cube with axis-aligned orientations, odometry with rotation-vector noise, ranges in space.  Against
    oracle/generate_oracle.py::world3 (written from the description): every integer equal -- positions, orientations as
    rotation matrices, beacons, endpoints, counts --, reals to 1e-12; every odometry rotation is a rotation; the worlds solve
    (3 x 4 pose matrices, 4 x 4 chain blocks) and the estimate follows the truth; trial t is the same world in any batch."""
    from oracle.generate_oracle import rot3, world3

    spec = dict(n_robots=3, n_poses=50, n_beacons=4, side=6, p_range=0.3, sigma_t=0.02, sigma_theta=0.01, sigma_range=0.7)
    B = GeneratedBatch(3, seed=987654321, lib_path=twin_lib, dim=3, **spec)
    for t in range(3):
        w = world3(987654321, t, **spec)
        a = B.arrays(t)
        poses, beacons = B.truth(t)
        R, T = spec["n_robots"], spec["n_poses"]
        assert a["dim"] == 3 and poses.shape == (R * T, 12) and beacons.shape == (4, 3)
        assert np.array_equal(poses[:, :3], np.array([p for r in range(R) for p in w["pos"][r]], dtype=float))
        assert np.array_equal(poses[:, 3:].reshape(-1, 3, 3), np.array([rot3(o) for r in range(R) for o in w["ori"][r]], dtype=float))
        assert np.array_equal(beacons, np.array(w["beacons"], dtype=float))
        assert np.array_equal(a["rel_base"], np.array([o[0] for o in w["odom"]], np.int32)) and np.array_equal(a["rel_to"], np.array([o[1] for o in w["odom"]], np.int32))
        np.testing.assert_allclose(a["rel_t"], np.array([o[2] for o in w["odom"]]), rtol=0, atol=1e-12)
        np.testing.assert_allclose(a["rel_R"], np.array([o[3] for o in w["odom"]]), rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.einsum("nij,nkj->nik", a["rel_R"], a["rel_R"]), np.broadcast_to(np.eye(3), a["rel_R"].shape), atol=1e-12)
        assert np.all(np.linalg.det(a["rel_R"]) > 0.999)
        rg = w["ranges"]
        assert len(rg) == len(a["rng_a"]) > 30
        assert np.array_equal(a["rng_a"], np.array([x[0] for x in rg], np.int32)) and np.array_equal(a["rng_b"], np.array([x[1] for x in rg], np.int32))
        np.testing.assert_allclose(a["rng_dist"], np.array([x[2] for x in rg]), rtol=0, atol=1e-12)
        # the walk: unit steps along the body x axis, inside the cube; turns are quarter turns
        P3 = poses[:, :3].reshape(R, T, 3)
        Rw = poses[:, 3:].reshape(R, T, 3, 3)
        assert P3.min() >= 0 and P3.max() <= spec["side"]
        assert np.array_equal(P3[:, 1:] - P3[:, :-1], Rw[:, :-1, :, 0])
    assert np.array_equal(poses[0], [0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1])  # robot A starts at the origin with the identity orientation (the pinned pose)
    one = GeneratedBatch(1, seed=987654321 + 2, lib_path=twin_lib, dim=3, **spec)
    assert np.array_equal(one.arrays(0)["rng_dist"], B.arrays(2)["rng_dist"]) and np.array_equal(one.truth(0)[0], B.truth(2)[0])
    res = solve_score_batch(B.graphs(), "SOCP", lib_path=twin_lib)
    assert all(r.solved for r in res)
    err = B.trajectory_errors(1, res[1])
    assert set(err) == {"A", "B", "C"} and err["A"]["translation_rmse"] < 1.0 and err["A"]["heading_max"] < 0.3
    gt = B.truth_results(1)
    assert gt.poses["A0"].shape == (4, 4) and B.trajectory_errors(1, gt)["C"]["translation_max"] == 0.0 and B.trajectory_errors(1, gt)["C"]["heading_max"] < 1e-7


@pytest.mark.gpu
def test_device_generator_equals_the_host_loops(hip_lib, twin_lib):
    """The kernels (one thread per robot walk / per (trial, group, timestep) of the ranges, a scan in between) against the host
    loops of the twin: every integer (positions, headings, beacons, endpoints, counts) bit for bit, every real to a few ulps of
    the math libraries (sin, cos, log, sqrt, atan2 of device and host); BASELINE configs[4]'s shape and an odd one."""
    for spec in (dict(n_robots=4, n_poses=1000, n_beacons=4), dict(n_robots=7, n_poses=129, n_beacons=9, side=11, p_range=0.33, sigma_range=0.5),
                 dict(n_robots=4, n_poses=600, n_beacons=4, side=12, dim=3), dict(n_robots=5, n_poses=77, n_beacons=7, side=5, p_range=0.4, dim=3)):
        dev = GeneratedBatch(5, seed=4000, lib_path=hip_lib, **spec)
        ref = GeneratedBatch(5, seed=4000, lib_path=twin_lib, **spec)
        for i in range(5):
            a, b = dev.arrays(i), ref.arrays(i)
            for k in ("rel_base", "rel_to", "rng_a", "rng_b", "chain_len"):
                assert np.array_equal(a[k], b[k]), (spec, i, k)
            for k in ("rel_t", "rel_R", "rel_kappa", "rel_tau", "rng_dist", "rng_prec"):
                np.testing.assert_allclose(a[k], b[k], rtol=0, atol=1e-12, err_msg=f"{spec} {i} {k}")
            (pa, ba), (pb, bb) = dev.truth(i), ref.truth(i)
            if spec.get("dim") == 3:
                assert np.array_equal(pa, pb) and np.array_equal(ba, bb)  # (positions and integer rotation matrices)
                continue
            assert np.array_equal(pa[:, :2], pb[:, :2]) and np.array_equal(ba, bb)
            np.testing.assert_allclose(pa[:, 2], pb[:, 2], atol=1e-15)
    # worlds in a row of one batch: the handle is built from the arrays the generator left ON THE DEVICE (score_create_from_generated);
    # the same worlds handed over as host arrays (score_create_from_graphs) give the same program and the same solutions, bit for bit
    from score_amd.solver import ConicSolver

    B = GeneratedBatch(6, seed=4100, n_robots=4, n_poses=500, n_beacons=4, lib_path=hip_lib)
    arrs = [B.arrays(i) for i in range(6)]
    for relax in (0, 1):
        res_dev = ConicSolver.from_graphs(arrs[1:5], relax, {}, lib_path=hip_lib)
        assert res_dev._keep and res_dev._keep[0] is B  # (took the resident path)
        plain = [{k: v for k, v in a.items() if k not in ("_owner", "_index")} for a in arrs[1:5]]
        res_host = ConicSolver.from_graphs(plain, relax, {}, lib_path=hip_lib)
        assert not res_host._keep
        if relax == 0:
            for nm in ("qs", "bs", "Aval", "K0", "K1"):
                assert np.array_equal(res_dev.debug_get(nm), res_host.debug_get(nm)), nm
        for x, y in zip(res_dev.solve(), res_host.solve()):
            assert x.solved and y.solved and np.array_equal(x.x, y.x) and np.array_equal(x.y, y.y) and x.info["pobj"] == y.info["pobj"]
        res_dev.close(); res_host.close()
    # through the product: generated on the device, model + setup + solve + estimates on the device
    graphs = generate_manhattan(8, seed=4000, n_robots=4, n_poses=1000, n_beacons=4, lib_path=hip_lib)
    res = solve_score_batch(graphs, "SOCP")
    assert all(r.solved and r.info["newton_iters"] > 0 for r in res)
    again = solve_score(graphs[3], "SOCP")
    assert again.info["pobj"] == pytest.approx(res[3].info["pobj"], rel=1e-9)
    # 3-D worlds through the product (resident arrays, device assembler for 3 x 4 pose matrices, 4 x 4 chain blocks), certified
    B3 = GeneratedBatch(4, seed=5100, n_robots=3, n_poses=300, n_beacons=4, side=10, p_range=0.2, dim=3, lib_path=hip_lib)
    g3 = B3.graphs()
    r3 = solve_score_batch(g3, "SOCP")
    assert all(r.solved and r.info["newton_iters"] > 0 for r in r3)
    err = B3.trajectory_errors(2, r3[2])
    assert max(v["translation_rmse"] for v in err.values()) < 3.0
    from oracle import score_oracle as so
    from score_amd.native import assemble_native

    qp = assemble_native(g3[0], "SOCP", arrays=B3.arrays(0)).qp
    sv = ConicSolver([qp], {})
    out = sv.solve()[0]
    sv.close()
    cert = so.kkt_certificate(qp.P, qp.q, qp.A, qp.b, 0, qp.soc_dims, out.x, out.y, out.s)
    assert out.solved and cert["primal_res_inf"] < 1e-5 and cert["dual_res_inf"] < 1e-4, cert
    assert out.info["pobj"] == pytest.approx(r3[0].info["pobj"], rel=1e-6)


@pytest.mark.gpu
def test_config5_from_seeds_is_certified(hip_lib):
    """BASELINE configs[4]'s shape from nothing but seeds: 64 worlds of 4 robots x 1000 poses drawn on the device, built there
    (score_create_from_generated) and solved in lock-step groups; every world solved, the solver-independent KKT certificate of
    the conic program (built by the host assembler from the same arrays) for a sample of them, the pinned robot's trajectory
    within the odometry's drift of the ground truth, both relaxations' direct forms agreeing."""
    from oracle import score_oracle as so
    from score_amd.native import assemble_native
    from score_amd.solver import ConicSolver

    B = GeneratedBatch(64, seed=64000, n_robots=4, n_poses=1000, n_beacons=4, lib_path=hip_lib)
    graphs = B.graphs()
    res = solve_score_batch(graphs, "SOCP")
    assert len(res) == 64 and all(r.solved and r.info["newton_iters"] > 0 for r in res)
    for i in (0, 17, 63):
        truth, beacons = B.truth(i)
        est = res[i].poses.array[:1000, :2, 2]  # robot A (pinned at the origin)
        assert np.abs(est - truth[:1000, :2]).max() < 10.0, i  # (1000 steps of 0.002 rad heading noise on a 20 m grid: metres)
        a = B.arrays(i)
        qp = assemble_native(graphs[i], "SOCP", arrays=a).qp
        sv = ConicSolver.from_graphs([a], 0, {})
        out = sv.solve()[0]
        sv.close()
        cert = so.kkt_certificate(qp.P, qp.q, qp.A, qp.b, 0, qp.soc_dims, out.x, out.y, out.s)
        assert cert["primal_res_inf"] < 1e-5 and cert["dual_res_inf"] < 1e-4 and cert["s_cone_dist"] < 1e-9 and cert["y_cone_dist"] < 1e-9, (i, cert)
        assert out.info["pobj"] == pytest.approx(res[i].info["pobj"], rel=1e-8)
    rq = solve_score_batch(graphs[:8], "QCQP", qcqp_mode="direct")
    for a_, b_ in zip(rq, res[:8]):
        assert a_.solved and a_.info["pobj"] == pytest.approx(b_.info["pobj"], rel=1e-7)
        np.testing.assert_allclose(a_.poses.array, b_.poses.array, atol=1e-6 * max(1.0, np.abs(b_.poses.array).max()))


def test_zero_distance_ranges_get_one_direction_convention(twin_lib):
    """A range measured as exactly 0 (the generator clamps at 0; the shipped fixture holds five) leaves its QCQP direction free:
    w |t_i - t_j - 0 r|^2 does not depend on r.  Every path returns r = 0 there -- the device read-back (k_read_estimates /
    read_estimates_host), the Python closed form and the direct QCQP's headform_expand -- so the answer does not depend on the
    assembler (advisor finding, round 5: 0.708 between 'device' and 'native' on this world)."""
    G = generate_manhattan(3, seed=5, lib_path=twin_lib)[2]
    zero = np.nonzero(G.arrays["rng_dist"] == 0)[0]
    assert len(zero) >= 1
    a = solve_score(G, "QCQP", lib_path=twin_lib, assembler="device")
    b = solve_score(G, "QCQP", lib_path=twin_lib, assembler="native")
    c = solve_score(G, "QCQP", lib_path=twin_lib, assembler="native", qcqp_mode="direct")
    A, B, Cc = (r.variables.distances.array for r in (a, b, c))
    assert np.all(A[zero] == 0.0) and np.all(B[zero] == 0.0) and np.all(Cc[zero] == 0.0)
    np.testing.assert_allclose(A, B, atol=1e-8)
    np.testing.assert_allclose(A, Cc, atol=1e-4)
