"""Factor-graph ingest/egress for the SCORE path.

* ``load_pyfg_pickle``: reads a PyFactorGraph pickle (the format of the
  reference's fixtures, examples/solve_goats_example_score.py:18,40) WITHOUT
  ``py_factor_graph`` installed, via a restricted unpickler that maps the six
  ``py_factor_graph.*`` globals those pickles contain onto ``score_amd.compat``
  classes and refuses everything else.
* ``save_fg_npz`` / ``load_fg_npz``: a neutral array format (what is committed
  under tests/golden/, since pickles of foreign classes cannot travel).
* ``save_to_tum``: trajectory export (the reference imports
  ``py_factor_graph.utils.solver_utils.save_to_tum`` at gurobi_utils.py:17;
  format of examples/goats_14_data/gt_traj_A.tum: ``t x y z qx qy qz qw``).
"""
from __future__ import annotations

import io as _io
import pickle
from typing import Dict, List

import numpy as np

from . import compat

_ALLOWED_PYFG = {
    ("py_factor_graph.factor_graph", "FactorGraphData"),
    ("py_factor_graph.variables", "PoseVariable2D"),
    ("py_factor_graph.variables", "LandmarkVariable2D"),
    ("py_factor_graph.measurements", "PoseMeasurement2D"),
    ("py_factor_graph.measurements", "FGRangeMeasurement"),
    ("py_factor_graph.priors", "PosePrior2D"),
    ("py_factor_graph.priors", "LandmarkPrior2D"),
}
_ALLOWED_OTHER = {
    ("numpy.core.multiarray", "scalar"),
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "dtype"),
    ("numpy", "ndarray"),
    ("builtins", "set"),
    ("builtins", "frozenset"),
}


class _Raw:
    """Receives pickled state verbatim; converted to compat objects afterwards."""

    _kind = ""

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif (
            isinstance(state, tuple)
            and len(state) == 2
            and isinstance(state[1], dict)
            and (state[0] is None or isinstance(state[0], dict))
        ):
            if state[0]:
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self.__dict__["_state"] = state


class _PyfgUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED_PYFG:
            return type(name, (_Raw,), {"_kind": name})
        if (module, name) in _ALLOWED_OTHER:
            if module in ("numpy.core.multiarray", "numpy._core.multiarray"):
                # numpy 2 moved numpy.core to numpy._core (pickles carry either name); import
                # whichever this numpy provides
                import importlib.util

                for cand in ("numpy._core.multiarray", "numpy.core.multiarray"):
                    try:
                        if importlib.util.find_spec(cand) is not None:
                            module = cand
                            break
                    except ModuleNotFoundError:
                        continue
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"global {module}.{name} is not allowed")


def _conv_pose_var(r) -> compat.PoseVariable2D:
    return compat.PoseVariable2D(
        name=r.name,
        true_position=tuple(float(v) for v in r.true_position),
        true_theta=float(r.true_theta),
        timestamp=getattr(r, "timestamp", None),
    )


def _conv_pose_meas(r) -> compat.PoseMeasurement2D:
    return compat.PoseMeasurement2D(
        base_pose=r.base_pose,
        to_pose=r.to_pose,
        x=float(r.x),
        y=float(r.y),
        theta=float(r.theta),
        translation_precision=float(r.translation_precision),
        rotation_precision=float(r.rotation_precision),
        timestamp=getattr(r, "timestamp", None),
    )


def _conv_range(r) -> compat.FGRangeMeasurement:
    return compat.FGRangeMeasurement(
        association=tuple(r.association),
        dist=float(r.dist),
        stddev=float(r.stddev),
        timestamp=getattr(r, "timestamp", None),
    )


def _conv_pose_prior(r) -> compat.PosePrior2D:
    if hasattr(r, "_state"):
        name, pos, theta, tp, rp, ts = r._state
    else:
        name, pos, theta = r.name, r.position, r.theta
        tp, rp, ts = r.translation_precision, r.rotation_precision, getattr(r, "timestamp", None)
    return compat.PosePrior2D(name, tuple(float(v) for v in pos), float(theta), float(tp), float(rp), ts)


def _conv_landmark_prior(r) -> compat.LandmarkPrior2D:
    if hasattr(r, "_state"):
        name, pos, tp, ts = r._state
    else:
        name, pos, tp, ts = r.name, r.position, r.translation_precision, getattr(r, "timestamp", None)
    return compat.LandmarkPrior2D(name, tuple(float(v) for v in pos), float(tp), ts)


def load_pyfg_pickle(path: str) -> compat.FactorGraphData:
    """Parse a PyFactorGraph 2-D pickle into a ``compat.FactorGraphData``."""
    with open(path, "rb") as f:
        raw = _PyfgUnpickler(_io.BytesIO(f.read())).load()
    if getattr(raw, "_kind", "") != "FactorGraphData":
        raise ValueError(f"{path}: not a pickled FactorGraphData")
    if int(raw.dimension) != 2:
        raise ValueError("only 2-D PyFactorGraph pickles are supported by the ingest")
    fg = compat.FactorGraphData(dimension=int(raw.dimension))
    fg.pose_variables = [[_conv_pose_var(p) for p in chain] for chain in raw.pose_variables]
    fg.landmark_variables = [
        compat.LandmarkVariable2D(l.name, tuple(float(v) for v in l.true_position))
        for l in raw.landmark_variables
    ]
    fg.odom_measurements = [[_conv_pose_meas(m) for m in chain] for chain in raw.odom_measurements]
    fg.loop_closure_measurements = [_conv_pose_meas(m) for m in raw.loop_closure_measurements]
    fg.range_measurements = [_conv_range(m) for m in raw.range_measurements]
    fg.pose_priors = [_conv_pose_prior(p) for p in getattr(raw, "pose_priors", [])]
    fg.landmark_priors = [_conv_landmark_prior(p) for p in getattr(raw, "landmark_priors", [])]
    return fg


# ---------------------------------------------------------------------------
# neutral array format (2-D graphs)
# ---------------------------------------------------------------------------
def save_fg_npz(path: str, fg: compat.FactorGraphData) -> None:
    assert fg.dimension == 2, "npz fixture format covers 2-D graphs"
    names = fg.get_pose_chain_names()
    chain_len = np.array([len(c) for c in names], dtype=np.int64)
    flat = [p for chain in fg.pose_variables for p in chain]
    meas = [m for chain in fg.odom_measurements for m in chain]
    odom_len = np.array([len(c) for c in fg.odom_measurements], dtype=np.int64)

    def pm(ms):
        return dict(
            base=np.array([m.base_pose for m in ms], dtype="U"),
            to=np.array([m.to_pose for m in ms], dtype="U"),
            xyt=np.array([[m.x, m.y, m.theta] for m in ms], dtype=np.float64).reshape(-1, 3),
            prec=np.array([[m.translation_precision, m.rotation_precision] for m in ms], dtype=np.float64).reshape(-1, 2),
        )

    od, lc = pm(meas), pm(fg.loop_closure_measurements)
    np.savez_compressed(
        path,
        dimension=np.int64(fg.dimension),
        chain_len=chain_len,
        pose_names=np.array([p.name for p in flat], dtype="U"),
        pose_true=np.array([[*p.true_position, p.true_theta] for p in flat], dtype=np.float64).reshape(-1, 3),
        lm_names=np.array([l.name for l in fg.landmark_variables], dtype="U"),
        lm_true=np.array([l.true_position for l in fg.landmark_variables], dtype=np.float64).reshape(-1, 2),
        odom_len=odom_len,
        odom_base=od["base"], odom_to=od["to"], odom_xyt=od["xyt"], odom_prec=od["prec"],
        lc_base=lc["base"], lc_to=lc["to"], lc_xyt=lc["xyt"], lc_prec=lc["prec"],
        rng_a=np.array([m.first_key for m in fg.range_measurements], dtype="U"),
        rng_b=np.array([m.second_key for m in fg.range_measurements], dtype="U"),
        rng_dist=np.array([m.dist for m in fg.range_measurements], dtype=np.float64),
        rng_std=np.array([m.stddev for m in fg.range_measurements], dtype=np.float64),
        pp_names=np.array([p.name for p in fg.pose_priors], dtype="U"),
        pp_vals=np.array([[*p.position, p.theta, p.translation_precision, p.rotation_precision] for p in fg.pose_priors], dtype=np.float64).reshape(-1, 5),
        lp_names=np.array([p.name for p in fg.landmark_priors], dtype="U"),
        lp_vals=np.array([[*p.position, p.translation_precision] for p in fg.landmark_priors], dtype=np.float64).reshape(-1, 3),
    )


def load_fg_npz(path: str) -> compat.FactorGraphData:
    z = np.load(path, allow_pickle=False)
    fg = compat.FactorGraphData(dimension=int(z["dimension"]))
    k = 0
    for n in z["chain_len"]:
        chain = []
        for i in range(k, k + int(n)):
            x, y, th = z["pose_true"][i]
            chain.append(compat.PoseVariable2D(str(z["pose_names"][i]), (float(x), float(y)), float(th)))
        fg.pose_variables.append(chain)
        k += int(n)
    fg.landmark_variables = [
        compat.LandmarkVariable2D(str(n), (float(p[0]), float(p[1]))) for n, p in zip(z["lm_names"], z["lm_true"])
    ]

    def pm(prefix, i):
        x, y, th = z[prefix + "_xyt"][i]
        tp, rp = z[prefix + "_prec"][i]
        return compat.PoseMeasurement2D(
            str(z[prefix + "_base"][i]), str(z[prefix + "_to"][i]), float(x), float(y), float(th), float(tp), float(rp)
        )

    k = 0
    for n in z["odom_len"]:
        fg.odom_measurements.append([pm("odom", i) for i in range(k, k + int(n))])
        k += int(n)
    fg.loop_closure_measurements = [pm("lc", i) for i in range(len(z["lc_base"]))]
    fg.range_measurements = [
        compat.FGRangeMeasurement((str(a), str(b)), float(d), float(s))
        for a, b, d, s in zip(z["rng_a"], z["rng_b"], z["rng_dist"], z["rng_std"])
    ]
    fg.pose_priors = [
        compat.PosePrior2D(str(n), (float(v[0]), float(v[1])), float(v[2]), float(v[3]), float(v[4]))
        for n, v in zip(z["pp_names"], z["pp_vals"])
    ]
    fg.landmark_priors = [
        compat.LandmarkPrior2D(str(n), (float(v[0]), float(v[1])), float(v[2])) for n, v in zip(z["lp_names"], z["lp_vals"])
    ]
    return fg


# ---------------------------------------------------------------------------
# TUM export / import
# ---------------------------------------------------------------------------
def save_to_tum(results, filepath_prefix: str, timestamps: List[float] = None) -> List[str]:
    """One ``<prefix>_<chain letter>.tum`` file per pose chain, rows
    ``t x y z qx qy qz qw``.  2-D poses get z = 0 and a yaw-only quaternion."""
    files = []
    chains = results.pose_chain_names or [list(results.poses.keys())]
    for chain in chains:
        if not chain:
            continue
        letter = chain[0][0]
        path = f"{filepath_prefix}_{letter}.tum"
        with open(path, "w") as f:
            for i, name in enumerate(chain):
                T = results.poses[name]
                d = T.shape[0] - 1
                t = timestamps[i] if timestamps is not None else float(i)
                if d == 2:
                    th = np.arctan2(T[1, 0], T[0, 0])
                    q = (0.0, 0.0, np.sin(th / 2), np.cos(th / 2))
                    xyz = (T[0, 2], T[1, 2], 0.0)
                else:
                    q = _quat_from_rot(T[:3, :3])
                    xyz = tuple(T[:3, 3])
                f.write(f"{t:.6f} {xyz[0]:.9f} {xyz[1]:.9f} {xyz[2]:.9f} {q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f}\n")
        files.append(path)
    return files


def load_tum(path: str) -> np.ndarray:
    """Rows ``t x y z qx qy qz qw`` -> (N, 8) array."""
    return np.loadtxt(path, dtype=np.float64).reshape(-1, 8)


def _quat_from_rot(R: np.ndarray):
    tr = np.trace(R)
    if tr > 0:
        s = 2.0 * np.sqrt(tr + 1.0)
        return ((R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s)
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
    q = [0.0, 0.0, 0.0, 0.0]
    q[i] = 0.25 * s
    q[j] = (R[j, i] + R[i, j]) / s
    q[k] = (R[k, i] + R[i, k]) / s
    q[3] = (R[k, j] - R[j, k]) / s
    return tuple(q)
