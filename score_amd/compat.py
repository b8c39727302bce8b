"""Duck-typed stand-ins for the PyFactorGraph types the SCORE solve path touches.

The reference consumes ``py_factor_graph`` objects (not vendored in the
reference repo, not installable here).  These classes expose exactly the
attribute surface the reference reads -- see SURVEY.md section 8(b):

* ``FactorGraphData``: ``dimension`` (score/solve_score.py:72),
  ``unconnected_variable_names`` (:29), ``pose_variables``
  (score/utils/gurobi_utils.py:181,237), ``landmark_variables`` (:253),
  ``odom_measurements`` (:398), ``loop_closure_measurements`` (:425),
  ``range_measurements`` (:281,288,463), ``landmark_priors`` (:438),
  ``get_pose_chain_names()`` (:196).
* pose measurement: ``base_pose``, ``to_pose`` (:400-401),
  ``translation_precision``, ``translation_vector``, ``rotation_precision``,
  ``rotation_matrix`` (:514-522).
* range measurement: ``first_key``, ``second_key`` (:288,464), ``dist``,
  ``precision`` (:487,500).
* landmark prior: ``name``, ``translation_vector``, ``translation_precision``
  (:441-444).
* results: ``VariableValues(dim, poses, landmarks, distances)`` (:136) and
  ``SolverResults(variables=, total_time=, solved=, pose_chain_names=)``
  (:197-202).

A real ``py_factor_graph.FactorGraphData`` works wherever these do: the solve
path only uses the attributes above (duck typing), never ``isinstance``.
"""
from __future__ import annotations

from collections.abc import Mapping
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def _rot2(theta: float) -> np.ndarray:
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s], [s, c]], dtype=np.float64)


@dataclass
class PoseVariable2D:
    name: str
    true_position: Tuple[float, float] = (0.0, 0.0)
    true_theta: float = 0.0
    timestamp: Optional[float] = None

    @property
    def rotation_matrix(self) -> np.ndarray:
        return _rot2(self.true_theta)

    @property
    def transformation_matrix(self) -> np.ndarray:
        T = np.eye(3)
        T[:2, :2] = self.rotation_matrix
        T[:2, 2] = self.true_position
        return T


@dataclass
class PoseVariable3D:
    name: str
    true_position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    true_rotation: np.ndarray = field(default_factory=lambda: np.eye(3))
    timestamp: Optional[float] = None

    @property
    def rotation_matrix(self) -> np.ndarray:
        return np.asarray(self.true_rotation, dtype=np.float64)

    @property
    def transformation_matrix(self) -> np.ndarray:
        T = np.eye(4)
        T[:3, :3] = self.rotation_matrix
        T[:3, 3] = self.true_position
        return T


@dataclass
class LandmarkVariable2D:
    name: str
    true_position: Tuple[float, float] = (0.0, 0.0)


@dataclass
class LandmarkVariable3D:
    name: str
    true_position: Tuple[float, float, float] = (0.0, 0.0, 0.0)


@dataclass
class PoseMeasurement2D:
    """Relative SE(2) measurement; pickled state is x, y, theta + precisions."""

    base_pose: str
    to_pose: str
    x: float
    y: float
    theta: float
    translation_precision: float
    rotation_precision: float
    timestamp: Optional[float] = None

    @property
    def translation_vector(self) -> np.ndarray:
        return np.array([self.x, self.y], dtype=np.float64)

    @property
    def rotation_matrix(self) -> np.ndarray:
        return _rot2(self.theta)


@dataclass
class PoseMeasurement3D:
    base_pose: str
    to_pose: str
    translation: np.ndarray
    rotation: np.ndarray
    translation_precision: float
    rotation_precision: float
    timestamp: Optional[float] = None

    @property
    def translation_vector(self) -> np.ndarray:
        return np.asarray(self.translation, dtype=np.float64)

    @property
    def rotation_matrix(self) -> np.ndarray:
        return np.asarray(self.rotation, dtype=np.float64)


@dataclass
class FGRangeMeasurement:
    """Range measurement; pickled state is association, dist, stddev."""

    association: Tuple[str, str]
    dist: float
    stddev: float
    timestamp: Optional[float] = None

    @property
    def first_key(self) -> str:
        return self.association[0]

    @property
    def second_key(self) -> str:
        return self.association[1]

    @property
    def variance(self) -> float:
        return self.stddev ** 2

    @property
    def precision(self) -> float:
        return 1.0 / (self.stddev ** 2)

    @property
    def weight(self) -> float:
        return self.precision


@dataclass
class PosePrior2D:
    """Carried for fidelity with the pickles; the SCORE objective ignores pose
    priors (nothing in score/utils/gurobi_utils.py reads ``pose_priors``)."""

    name: str
    position: Tuple[float, float]
    theta: float
    translation_precision: float
    rotation_precision: float
    timestamp: Optional[float] = None


@dataclass
class LandmarkPrior2D:
    name: str
    position: Tuple[float, float]
    translation_precision: float
    timestamp: Optional[float] = None

    @property
    def translation_vector(self) -> np.ndarray:
        return np.asarray(self.position, dtype=np.float64)


@dataclass
class LandmarkPrior3D:
    name: str
    position: Tuple[float, float, float]
    translation_precision: float
    timestamp: Optional[float] = None

    @property
    def translation_vector(self) -> np.ndarray:
        return np.asarray(self.position, dtype=np.float64)


@dataclass
class FactorGraphData:
    dimension: int = 2
    pose_variables: List[list] = field(default_factory=list)
    landmark_variables: list = field(default_factory=list)
    odom_measurements: List[list] = field(default_factory=list)
    loop_closure_measurements: list = field(default_factory=list)
    range_measurements: list = field(default_factory=list)
    pose_priors: list = field(default_factory=list)
    landmark_priors: list = field(default_factory=list)

    # ---- the surface read by the solve path -------------------------------
    @property
    def num_poses(self) -> int:
        return sum(len(c) for c in self.pose_variables)

    @property
    def num_landmarks(self) -> int:
        return len(self.landmark_variables)

    @property
    def all_variable_names(self) -> List[str]:
        names = [p.name for chain in self.pose_variables for p in chain]
        names += [l.name for l in self.landmark_variables]
        return names

    @property
    def unconnected_variable_names(self) -> List[str]:
        """Variables no factor that SCORE uses touches (solve_score.py:28-32)."""
        touched = set()
        for chain in self.odom_measurements:
            touched.update(m.base_pose for m in chain)
            touched.update(m.to_pose for m in chain)
        touched.update(m.base_pose for m in self.loop_closure_measurements)
        touched.update(m.to_pose for m in self.loop_closure_measurements)
        for m in self.range_measurements:
            touched.update(m.association)
        touched.update(p.name for p in self.landmark_priors)
        return [n for n in self.all_variable_names if n not in touched]

    def get_pose_chain_names(self) -> List[List[str]]:
        return [[p.name for p in chain] for chain in self.pose_variables]

    # ---- convenience ------------------------------------------------------
    @property
    def true_poses(self) -> Dict[str, np.ndarray]:
        return {
            p.name: p.transformation_matrix
            for chain in self.pose_variables
            for p in chain
        }

    @property
    def true_landmarks(self) -> Dict[str, np.ndarray]:
        return {
            l.name: np.asarray(l.true_position, dtype=np.float64)
            for l in self.landmark_variables
        }


# ---------------------------------------------------------------------------
# results (py_factor_graph.utils.solver_utils counterparts)
# ---------------------------------------------------------------------------
class ArrayDict(Mapping):
    """A read-only ``dict``-like view ``name -> array[i]`` over one stacked array.  Building 20 000
    small arrays and a real dict for every solve costs more than the solve itself; the name index
    is built on the first lookup, rows are views into the stacked array (``dict(view)`` gives a
    plain dict where one is wanted)."""

    __slots__ = ("_names", "_array", "_index")

    def __init__(self, names, array):
        self._names = names
        self._array = array
        self._index = None

    def _idx(self):
        if self._index is None:
            self._index = {nm: i for i, nm in enumerate(self._names)}
        return self._index

    def __getitem__(self, key):
        return self._array[self._idx()[key]]

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, key):
        return key in self._idx()

    # keys() / items() / values() are Mapping's own views (sized, re-iterable, like a dict's): code written against
    # the reference's Dict[str, ndarray] -- len(res.poses.values()), two passes over items() -- keeps working.  Rows
    # are VIEWS into one stacked array: an in-place edit of a row edits the stack (dict(view) gives independent keys,
    # {k: v.copy() ...} independent values).

    @property
    def array(self) -> np.ndarray:
        """The stacked values, in ``keys()`` order."""
        return self._array

    def __repr__(self):
        return f"ArrayDict({len(self._names)} entries, item shape {self._array.shape[1:]})"


@dataclass
class VariableValues:
    dim: int
    poses: Dict[str, np.ndarray]
    landmarks: Dict[str, np.ndarray]
    distances: Optional[Dict[Tuple[str, str], np.ndarray]] = None

    @property
    def rotations_theta(self) -> Dict[str, float]:
        assert self.dim == 2
        return {k: float(np.arctan2(T[1, 0], T[0, 0])) for k, T in self.poses.items()}

    @property
    def rotations_matrix(self) -> Dict[str, np.ndarray]:
        return {k: T[: self.dim, : self.dim] for k, T in self.poses.items()}

    @property
    def translations(self) -> Dict[str, np.ndarray]:
        return {k: T[: self.dim, self.dim] for k, T in self.poses.items()}


@dataclass
class SolverResults:
    variables: VariableValues
    total_time: float
    solved: bool
    pose_chain_names: Optional[list] = None
    solver_cost: Optional[float] = None
    info: Optional[dict] = None  # ADMM statistics (not in the reference type)
    # the relaxation's own variables before SO(d) rounding (not in the reference type): the d x (d+1)
    # blocks [R | t] the convex program returned, for diagnostics and parity checks
    relaxed_poses: Optional[dict] = None

    @property
    def poses(self):
        return self.variables.poses

    @property
    def translations(self):
        return self.variables.translations

    @property
    def rotations_theta(self):
        return self.variables.rotations_theta

    @property
    def rotations_matrix(self):
        return self.variables.rotations_matrix

    @property
    def landmarks(self):
        return self.variables.landmarks

    @property
    def distances(self):
        return self.variables.distances
