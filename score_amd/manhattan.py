"""Seeded synthetic multi-robot Manhattan-world RA-SLAM generator.

Reproduces the statistics of the reference's shipped simulation fixture
(examples/manhattan/factor_graph.pickle, measured in SURVEY.md section 8d):
integer-lattice walks with unit steps and headings in {0, +-pi/2, pi} (a turn
on ~18 % of steps, a U-turn on ~1 %), odometry noise sigma_t = 0.01 m
(precision 1e4) and sigma_theta = 0.002 rad (precision 2.5e5), range noise
sigma = 1 m (precision 1) clamped at >= 0, every robot-beacon and same-timestep
robot-robot pair measured with probability ~0.10, no loop closures, robot A's
first pose at the origin with identity heading (it is the pinned pose).
Robots are named A, B, ... (skipping 'L', which PyFactorGraph reserves for
landmarks), poses ``<robot><step>``, landmarks ``L<i>``.
"""
from __future__ import annotations

import numpy as np

from . import compat

_TURNS = np.array([0.0, np.pi / 2, -np.pi / 2, np.pi])
_TURN_P = np.array([0.81, 0.09, 0.09, 0.01])


def robot_letters(n: int):
    out = []
    c = ord("A")
    while len(out) < n:
        ch = chr(c) if c <= ord("Z") else f"R{c - ord('Z')}_"
        if ch != "L":
            out.append(ch)
        c += 1
    return out


def _walk(rng, n_poses: int, side: int, start, heading: int):
    """Lattice walk; heading index h in {0,1,2,3} = h * pi/2.  Returns integer
    positions (n, 2) and heading indices (n,)."""
    dirs = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]])
    turn_steps = np.array([0, 1, -1, 2])
    pos = np.zeros((n_poses, 2), dtype=np.int64)
    hd = np.zeros(n_poses, dtype=np.int64)
    pos[0] = start
    hd[0] = heading
    for i in range(1, n_poses):
        pos[i] = pos[i - 1] + dirs[hd[i - 1]]
        # choose the next heading so that the following forward step stays inside
        order = rng.choice(4, size=4, replace=False, p=_TURN_P)
        for o in order:
            h = (hd[i - 1] + turn_steps[o]) % 4
            nxt = pos[i] + dirs[h]
            if 0 <= nxt[0] <= side and 0 <= nxt[1] <= side:
                hd[i] = h
                break
        else:  # cannot happen on a grid with side >= 1
            hd[i] = (hd[i - 1] + 2) % 4
    return pos, hd


def make_manhattan(
    n_robots: int = 4,
    n_poses: int = 400,
    n_beacons: int = 6,
    seed: int = 0,
    side: int = 20,
    p_range: float = 0.10,
    sigma_t: float = 0.01,
    sigma_theta: float = 0.002,
    sigma_range: float = 1.0,
    n_loop_closures: int = 0,
) -> compat.FactorGraphData:
    rng = np.random.default_rng(seed)
    dirs = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]])
    letters = robot_letters(n_robots)
    fg = compat.FactorGraphData(dimension=2)
    all_pos, all_hd = [], []
    for r in range(n_robots):
        if r == 0:
            start, h0 = np.array([0, 0]), 0
        else:
            while True:
                start = rng.integers(0, side + 1, size=2)
                h0 = int(rng.integers(0, 4))
                nxt = start + dirs[h0]
                if 0 <= nxt[0] <= side and 0 <= nxt[1] <= side:
                    break
        pos, hd = _walk(rng, n_poses, side, start, h0)
        all_pos.append(pos)
        all_hd.append(hd)
        theta = hd * (np.pi / 2)
        theta = np.arctan2(np.sin(theta), np.cos(theta))
        fg.pose_variables.append(
            [
                compat.PoseVariable2D(f"{letters[r]}{i}", (float(pos[i, 0]), float(pos[i, 1])), float(theta[i]))
                for i in range(n_poses)
            ]
        )
        # odometry: unit step forward in the base frame, then the turn
        dth = ((hd[1:] - hd[:-1] + 1) % 4 - 1) * (np.pi / 2)  # in {-pi/2, 0, pi/2, pi}
        nx = 1.0 + sigma_t * rng.standard_normal(n_poses - 1)
        ny = sigma_t * rng.standard_normal(n_poses - 1)
        nth = dth + sigma_theta * rng.standard_normal(n_poses - 1)
        nth = np.arctan2(np.sin(nth), np.cos(nth))
        fg.odom_measurements.append(
            [
                compat.PoseMeasurement2D(
                    f"{letters[r]}{i}", f"{letters[r]}{i + 1}", float(nx[i]), float(ny[i]), float(nth[i]),
                    1.0 / sigma_t ** 2, 1.0 / sigma_theta ** 2,
                )
                for i in range(n_poses - 1)
            ]
        )
    beacons = rng.integers(0, side + 1, size=(n_beacons, 2)).astype(np.float64)
    fg.landmark_variables = [
        compat.LandmarkVariable2D(f"L{i}", (float(b[0]), float(b[1]))) for i, b in enumerate(beacons)
    ]
    P = np.stack(all_pos).astype(np.float64)  # (R, T, 2)
    # robot-beacon ranges
    for r in range(n_robots):
        if n_beacons == 0:
            break
        hit = rng.random((n_poses, n_beacons)) < p_range
        ti, bi = np.nonzero(hit)
        true = np.linalg.norm(P[r, ti] - beacons[bi], axis=1)
        meas = np.maximum(0.0, true + sigma_range * rng.standard_normal(true.size))
        for t, b, dd in zip(ti, bi, meas):
            fg.range_measurements.append(
                compat.FGRangeMeasurement((f"{letters[r]}{t}", f"L{b}"), float(dd), float(sigma_range))
            )
    # same-timestep robot-robot ranges
    for a in range(n_robots):
        for b in range(a + 1, n_robots):
            ti = np.nonzero(rng.random(n_poses) < p_range)[0]
            true = np.linalg.norm(P[a, ti] - P[b, ti], axis=1)
            meas = np.maximum(0.0, true + sigma_range * rng.standard_normal(true.size))
            for t, dd in zip(ti, meas):
                fg.range_measurements.append(
                    compat.FGRangeMeasurement((f"{letters[a]}{t}", f"{letters[b]}{t}"), float(dd), float(sigma_range))
                )
    # optional loop closures (the shipped fixture has none)
    for _ in range(n_loop_closures):
        r = int(rng.integers(0, n_robots))
        i, j = sorted(rng.choice(n_poses, size=2, replace=False))
        Ti = fg.pose_variables[r][i].transformation_matrix
        Tj = fg.pose_variables[r][j].transformation_matrix
        rel = np.linalg.inv(Ti) @ Tj
        th = np.arctan2(rel[1, 0], rel[0, 0]) + sigma_theta * rng.standard_normal()
        fg.loop_closure_measurements.append(
            compat.PoseMeasurement2D(
                f"{letters[r]}{i}", f"{letters[r]}{j}",
                float(rel[0, 2] + sigma_t * rng.standard_normal()),
                float(rel[1, 2] + sigma_t * rng.standard_normal()),
                float(th), 1.0 / sigma_t ** 2, 1.0 / sigma_theta ** 2,
            )
        )
    return fg


def _rotvec(v: np.ndarray) -> np.ndarray:
    """Rotation matrix of the rotation vector v (Rodrigues)."""
    th = float(np.linalg.norm(v))
    if th < 1e-15:
        return np.eye(3)
    k = v / th
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


def make_manhattan_3d(
    n_robots: int = 4,
    n_poses: int = 400,
    n_beacons: int = 6,
    seed: int = 0,
    side: int = 12,
    p_range: float = 0.10,
    sigma_t: float = 0.01,
    sigma_theta: float = 0.002,
    sigma_range: float = 1.0,
    p_turn: float = 0.25,
    n_loop_closures: int = 0,
) -> compat.FactorGraphData:
    """The 3-D counterpart of `make_manhattan` (the reference's model is dimension-generic,
    gurobi_utils.py:37-50, :53-60; it ships no 3-D data): every robot walks the integer lattice of a cube of
    the given side with an axis-aligned orientation -- one unit step along its body x axis per pose, a
    quarter turn about its body z or y axis with probability p_turn (and whenever the step would leave the
    cube) -- odometry = that motion in the base frame plus noise (translation sigma_t per axis, rotation a
    random rotation vector of sigma_theta per axis), ranges to beacons and between robots at equal timestamps
    with probability p_range each.  Same statistics and naming as the 2-D generator."""
    rng = np.random.default_rng(seed)
    letters = robot_letters(n_robots)
    fg = compat.FactorGraphData(dimension=3)
    quarter = [_rotvec(np.array(a) * (np.pi / 2)) for a in ((0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0))]
    quarter = [np.rint(q) for q in quarter]
    all_pos = []
    for r in range(n_robots):
        pos = np.zeros(3) if r == 0 else rng.integers(0, side + 1, size=3).astype(np.float64)
        R = np.eye(3)
        if r > 0:
            for _ in range(int(rng.integers(0, 6))):
                R = R @ quarter[int(rng.integers(0, 4))]
        P, Rs = [pos.copy()], [R.copy()]
        for _ in range(n_poses - 1):
            # the motion of this step in the base frame: one unit forward, then possibly a quarter turn
            tries = 0
            while True:
                Rn = R @ quarter[int(rng.integers(0, 4))] if (rng.random() < p_turn or tries > 0) else R
                nxt = pos + R[:, 0]
                ahead = nxt + Rn[:, 0]  # the step after this one must stay inside as well
                if np.all(nxt >= 0) and np.all(nxt <= side) and np.all(ahead >= 0) and np.all(ahead <= side):
                    break
                tries += 1
                if tries > 64:  # boxed in along the current heading: turn in place first
                    R = R @ quarter[int(rng.integers(0, 4))]
                    tries = 1
            pos, R = nxt, Rn
            P.append(pos.copy()); Rs.append(R.copy())
        P = np.stack(P); all_pos.append(P)
        fg.pose_variables.append([
            compat.PoseVariable3D(f"{letters[r]}{i}", tuple(float(x) for x in P[i]), Rs[i].copy()) for i in range(n_poses)])
        odo = []
        for i in range(n_poses - 1):
            rel_R = Rs[i].T @ Rs[i + 1]
            rel_t = Rs[i].T @ (P[i + 1] - P[i])
            odo.append(compat.PoseMeasurement3D(
                f"{letters[r]}{i}", f"{letters[r]}{i + 1}", rel_t + sigma_t * rng.standard_normal(3),
                rel_R @ _rotvec(sigma_theta * rng.standard_normal(3)), 1.0 / sigma_t ** 2, 1.0 / sigma_theta ** 2))
        fg.odom_measurements.append(odo)
    beacons = rng.integers(0, side + 1, size=(n_beacons, 3)).astype(np.float64)
    fg.landmark_variables = [compat.LandmarkVariable3D(f"L{i}", tuple(float(x) for x in b)) for i, b in enumerate(beacons)]
    P = np.stack(all_pos)
    for r in range(n_robots):
        if n_beacons == 0:
            break
        ti, bi = np.nonzero(rng.random((n_poses, n_beacons)) < p_range)
        true = np.linalg.norm(P[r, ti] - beacons[bi], axis=1)
        meas = np.maximum(0.0, true + sigma_range * rng.standard_normal(true.size))
        for t, b, dd in zip(ti, bi, meas):
            fg.range_measurements.append(compat.FGRangeMeasurement((f"{letters[r]}{t}", f"L{b}"), float(dd), float(sigma_range)))
    for a in range(n_robots):
        for b in range(a + 1, n_robots):
            ti = np.nonzero(rng.random(n_poses) < p_range)[0]
            true = np.linalg.norm(P[a, ti] - P[b, ti], axis=1)
            meas = np.maximum(0.0, true + sigma_range * rng.standard_normal(true.size))
            for t, dd in zip(ti, meas):
                fg.range_measurements.append(compat.FGRangeMeasurement((f"{letters[a]}{t}", f"{letters[b]}{t}"), float(dd), float(sigma_range)))
    # optional loop closures within a robot's trajectory (drawn last: graphs without them are unchanged)
    for _ in range(n_loop_closures):
        r = int(rng.integers(0, n_robots))
        i, j = sorted(int(x) for x in rng.choice(n_poses, size=2, replace=False))
        Ri, Rj = fg.pose_variables[r][i].rotation_matrix, fg.pose_variables[r][j].rotation_matrix
        Pi, Pj = all_pos[r][i], all_pos[r][j]
        fg.loop_closure_measurements.append(compat.PoseMeasurement3D(
            f"{letters[r]}{i}", f"{letters[r]}{j}", Ri.T @ (Pj - Pi) + sigma_t * rng.standard_normal(3),
            Ri.T @ Rj @ _rotvec(sigma_theta * rng.standard_normal(3)), 1.0 / sigma_t ** 2, 1.0 / sigma_theta ** 2))
    return fg


# BASELINE.json configs (index -> generator arguments); seed = index*1000 + trial
CONFIGS = {
    1: dict(n_robots=1, n_poses=500, n_beacons=2),
    2: dict(n_robots=4, n_poses=1000, n_beacons=4),
    3: dict(n_robots=20, n_poses=1000, n_beacons=4),
    4: dict(n_robots=4, n_poses=1000, n_beacons=4),  # batch of 64 trials of config 2's shape
}


def make_config(index: int, trial: int = 0, **overrides) -> compat.FactorGraphData:
    kw = dict(CONFIGS[index])
    kw.update(overrides)
    return make_manhattan(seed=index * 1000 + trial, **kw)
