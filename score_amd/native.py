"""Native (C++) model construction through the C ABI: ``score_assemble``.

``assemble_native(data, relaxation)`` is a drop-in for ``score_amd.assemble.assemble`` -- it
returns a ``ScoreModel`` with the same column layout and read-back maps -- but the conic program
(P, q, A, b, cones, chain hint) is built by ``score_amd/csrc/score_assemble.hpp`` from flat
measurement arrays instead of SciPy sparse algebra: a few milliseconds per 20-robot graph, with
the GIL released (model construction is what bounds ``solve_score_batch`` once the solve takes
milliseconds; reference: score/utils/gurobi_utils.py:173-187 ``initialize_model``).

The assembled program lives in the library's host memory; ``NativeQP`` hands its ``score_problem``
view to ``score_create[_batch]`` directly and only materialises SciPy matrices when a caller asks
for ``.P`` / ``.A`` (tests, the oracle's certificate).
"""
from __future__ import annotations

import ctypes as C
from itertools import chain
from operator import attrgetter
from typing import Dict, Optional

import numpy as np
import scipy.sparse as sp

from .assemble import (
    QCQP_RELAXATION,
    SOCP_RELAXATION,
    ScoreModel,
    _check_unique,
    _pose_meas_arrays,
    check_dimension,
    check_valid_relaxation,
)
from .solver import ScoreProblem, _f64p, _i32p, load_library


class ScoreGraph(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("relaxation", C.c_int32), ("n_chains", C.c_int32), ("chain_len", _i32p),
        ("n_landmarks", C.c_int32),
        ("n_rel", C.c_int64), ("rel_base", _i32p), ("rel_to", _i32p), ("rel_t", _f64p), ("rel_R", _f64p),
        ("rel_kappa", _f64p), ("rel_tau", _f64p),
        ("n_rng", C.c_int64), ("rng_a", _i32p), ("rng_b", _i32p), ("rng_dist", _f64p), ("rng_prec", _f64p),
        ("n_lprior", C.c_int64), ("lprior_lm", _i32p), ("lprior_t", _f64p), ("lprior_prec", _f64p),
    ]


def _bind(lib: C.CDLL) -> None:
    if getattr(lib, "_score_assemble_bound", False):
        return
    lib.score_assemble.argtypes = [C.POINTER(ScoreGraph), C.POINTER(C.c_void_p)]
    lib.score_assemble_batch.argtypes = [C.POINTER(ScoreGraph), C.c_int32, C.POINTER(C.c_void_p)]
    lib.score_assembled_view.argtypes = [C.c_void_p, C.POINTER(ScoreProblem)]
    lib.score_assembled_free.argtypes = [C.c_void_p]
    lib.score_assembled_free.restype = None
    lib._score_assemble_bound = True


try:  # CPython helper (score_amd/csrc/_objread.c, built by __graft_entry__.build()): host glue, optional
    from . import _objread
except ImportError:  # pragma: no cover - the fromiter passes below are the same reads, one attribute at a time
    _objread = None


def _pose_meas_fast(meas: list, pose_idx: Dict[str, int], d: int):
    """``assemble._pose_meas_arrays`` in one pass over the measurement objects (``_objread.gather``)."""
    ne = len(meas)
    m0 = meas[0]
    planar = d == 2 and all(hasattr(m0, a) for a in ("x", "y", "theta"))
    if _objread is None or not planar:
        return _pose_meas_arrays(meas, pose_idx, d)
    bi, tj = np.empty(ne, np.int32), np.empty(ne, np.int32)
    kap, tau, x, y, th = (np.empty(ne) for _ in range(5))
    try:
        _objread.gather(meas, ("base_pose", "to_pose", "translation_precision", "rotation_precision", "x", "y", "theta"), "iiddddd",
                        (bi, tj, kap, tau, x, y, th), (pose_idx, pose_idx, None, None, None, None, None))
    except KeyError as exc:
        raise KeyError(exc.args[0]) from None
    tm = np.empty((ne, 2))
    Rm = np.empty((ne, 2, 2))
    tm[:, 0] = x; tm[:, 1] = y
    cs, sn = np.cos(th), np.sin(th)
    Rm[:, 0, 0] = cs; Rm[:, 0, 1] = -sn
    Rm[:, 1, 0] = sn; Rm[:, 1, 1] = cs
    return bi, tj, kap, tau, tm, Rm


def graph_arrays(data) -> Dict[str, np.ndarray]:
    """FactorGraphData -> the flat arrays of ``score_graph`` (the only per-measurement Python work
    left on the path: attribute reads -- one pass over every measurement list with ``_objread.gather``, or one
    ``numpy.fromiter`` pass per attribute without it).  Raises the reference's errors for duplicate / unknown
    variable names (gurobi_utils.py:62-80, :103-109)."""
    d = data.dimension
    check_dimension(d)
    chain_len = [len(c) for c in data.pose_variables]
    if _objread is not None:
        pvars = list(chain.from_iterable(data.pose_variables))
        pose_names = [None] * len(pvars)
        _objread.gather(pvars, ("name",), "o", (pose_names,), (None,))
    else:
        pose_names = [p.name for chain_ in data.pose_variables for p in chain_]
    landmark_names = [l.name for l in data.landmark_variables]
    pose_idx = dict(zip(pose_names, range(len(pose_names))))
    if len(pose_idx) != len(pose_names):
        _check_unique(pose_names, "pose_vars")  # (names the duplicate)
    if len(set(landmark_names)) != len(landmark_names):
        _check_unique(landmark_names, "landmark_vars")
    for nm in landmark_names:
        if nm in pose_idx:
            raise ValueError(f"Variable name {nm} already exists in pose_vars")
    Np = len(pose_names)
    if Np == 0:
        raise ValueError("factor graph has no pose variables")
    var_idx = dict(pose_idx)
    var_idx.update((nm, Np + i) for i, nm in enumerate(landmark_names))
    rm = data.range_measurements
    nr = len(rm)
    has_assoc = bool(nr) and hasattr(rm[0], "association")
    has_std = bool(nr) and hasattr(rm[0], "stddev")  # PyFactorGraph stores stddev; precision = 1 / stddev^2 is a derived property
    ra = rb = dist = prec = None
    if _objread is not None and has_assoc:
        range_keys = [None] * nr
        ab = np.empty((nr, 2), np.int32)
        dist, w = np.empty(nr), np.empty(nr)
        try:
            _objread.gather(rm, ("association", "association", "dist", "stddev" if has_std else "precision"), "Opdd",
                            (range_keys, ab, dist, w), (None, var_idx, None, None))
        except KeyError as exc:
            # the reference adds every distance variable (duplicate keys raise there, gurobi_utils.py:62-67, :288) before it
            # resolves an endpoint name (:103-109): a duplicate key anywhere in the list is reported first, as on the
            # fromiter path below
            seen = set()
            for k in (tuple(a) for a in map(attrgetter("association"), rm)):
                if k in seen:
                    raise ValueError(f"Variable name {k} already exists in distance_vars") from None
                seen.add(k)
            raise ValueError(f"Variable name {exc.args[0]} not found") from None
        ra, rb = np.ascontiguousarray(ab[:, 0]), np.ascontiguousarray(ab[:, 1])
        prec = 1.0 / (w * w) if has_std else w
    elif has_assoc:
        range_keys = [tuple(a) for a in map(attrgetter("association"), rm)]
    else:
        range_keys = [(m.first_key, m.second_key) for m in rm]
    if ra is not None:  # (the keys as index pairs: a sort of 64-bit codes instead of a set of tuples of strings)
        codes = np.sort(ra.astype(np.int64) * np.int64(len(var_idx)) + rb.astype(np.int64))
        duplicate_keys = bool(nr > 1 and (codes[1:] == codes[:-1]).any())
    else:
        duplicate_keys = len(set(range_keys)) != len(range_keys)
    if duplicate_keys:
        seen = set()
        for k in range_keys:
            if k in seen:
                raise ValueError(f"Variable name {k} already exists in distance_vars")
            seen.add(k)
    meas = list(chain.from_iterable(data.odom_measurements))
    meas += list(data.loop_closure_measurements)
    ne = len(meas)
    if ne:
        bi, tj, kap, tau, tm, Rm = _pose_meas_fast(meas, pose_idx, d)
    else:
        bi = tj = np.zeros(0, np.int64); kap = tau = np.zeros(0); tm = np.zeros((0, d)); Rm = np.zeros((0, d, d))
    if ra is None:
        try:  # (dict lookups in C: itemgetter over all keys at once)
            ra = np.fromiter(map(var_idx.__getitem__, (k[0] for k in range_keys)), dtype=np.int32, count=nr)
            rb = np.fromiter(map(var_idx.__getitem__, (k[1] for k in range_keys)), dtype=np.int32, count=nr)
        except KeyError as exc:
            raise ValueError(f"Variable name {exc.args[0]} not found") from None
        dist = np.fromiter(map(attrgetter("dist"), rm), dtype=np.float64, count=nr)
        if has_std:
            std = np.fromiter(map(attrgetter("stddev"), rm), dtype=np.float64, count=nr)
            prec = 1.0 / (std * std)
        else:
            prec = np.fromiter(map(attrgetter("precision"), rm), dtype=np.float64, count=nr)
    lm_idx = {nm: i for i, nm in enumerate(landmark_names)}
    pri = list(data.landmark_priors)
    for p in pri:
        if p.name not in lm_idx and p.name not in pose_idx:
            raise ValueError(f"Variable name {p.name} not found")
        if p.name not in lm_idx:
            raise ValueError(f"landmark prior on {p.name}: not a landmark")
    return dict(
        dim=d, pose_names=pose_names, landmark_names=landmark_names, range_keys=range_keys,
        n_loop_closures=len(data.loop_closure_measurements),
        # (= data.get_pose_chain_names(): the names just read, cut chain by chain)
        pose_chain_names=[pose_names[o - n_:o] for o, n_ in zip(np.cumsum(chain_len).tolist(), chain_len)],
        chain_len=np.array(chain_len, dtype=np.int32),
        rel_base=bi.astype(np.int32, copy=False), rel_to=tj.astype(np.int32, copy=False), rel_t=np.ascontiguousarray(tm, dtype=np.float64),
        rel_R=np.ascontiguousarray(Rm, dtype=np.float64), rel_kappa=np.ascontiguousarray(kap, dtype=np.float64),
        rel_tau=np.ascontiguousarray(tau, dtype=np.float64),
        rng_a=ra, rng_b=rb, rng_dist=dist, rng_prec=prec,
        lprior_lm=np.array([lm_idx[p.name] for p in pri], dtype=np.int32),
        lprior_t=np.array([np.asarray(p.translation_vector, dtype=np.float64) for p in pri], dtype=np.float64).reshape(-1, d),
        lprior_prec=np.array([float(p.translation_precision) for p in pri], dtype=np.float64),
    )


_CACHE_ATTR = "_score_amd_graph_arrays"


def graph_fingerprint(data) -> tuple:
    """What ``cached_graph_arrays`` keys the flat arrays of a graph object on: the dimension and, for every variable / measurement
    list of the graph, its identity, its length and the identities of its first and last element.  Appending, removing or
    replacing elements, or swapping a list, changes it; editing a field of a measurement object IN PLACE does not -- call
    ``invalidate_graph_cache(data)`` after such an edit (PyFactorGraph's measurement classes are immutable attrs classes)."""
    def one(lst):
        n = len(lst)
        return (id(lst), n, id(lst[0]) if n else 0, id(lst[-1]) if n else 0)

    return (int(data.dimension), one(data.pose_variables), tuple(one(c) for c in data.pose_variables), one(data.landmark_variables),
            one(data.odom_measurements), tuple(one(c) for c in data.odom_measurements), one(data.loop_closure_measurements),
            one(data.range_measurements), one(data.landmark_priors))


def cached_graph_arrays(data) -> Dict[str, np.ndarray]:
    """``graph_arrays(data)``, kept on the graph object: the reference's callers solve the same FactorGraphData again and again
    (both relaxations of one graph, score/solve_score.py:54-57; the intermediate iterates, :89-116), and the pass over 47 k
    measurement objects of the headline graph is 7.5 ms of attribute reads in front of a 3.8 ms solve.  The arrays are reused
    while ``graph_fingerprint(data)`` stands; they are never written to by the solve."""
    fp = graph_fingerprint(data)
    hit = getattr(data, _CACHE_ATTR, None)
    if hit is not None and hit[0] == fp:
        return hit[1]
    arrays = graph_arrays(data)
    try:
        object.__setattr__(data, _CACHE_ATTR, (fp, arrays))  # (also for frozen attrs classes; objects with __slots__: no cache)
    except (AttributeError, TypeError):
        pass
    return arrays


def invalidate_graph_cache(data) -> None:
    """Drop the flat arrays kept on ``data`` (after editing a measurement object in place)."""
    try:
        object.__delattr__(data, _CACHE_ATTR)
    except (AttributeError, TypeError):
        pass


class ArrayGraph:
    """A factor graph that already IS flat arrays (``graph_arrays(data)``, or a producer that never builds
    per-measurement Python objects): accepted wherever ``solve_score`` / ``solve_score_batch`` accept a
    FactorGraphData.  Skips the per-measurement attribute reads that bound the object path (they hold
    the GIL: ~5 ms per 4-robot x 1000-pose graph)."""

    def __init__(self, arrays: Dict[str, np.ndarray]):
        self.arrays = arrays

    @property
    def dimension(self) -> int:
        return int(self.arrays["dim"])

    @property
    def num_poses(self) -> int:
        return len(self.arrays["pose_names"])

    @property
    def num_ranges(self) -> int:
        return len(self.arrays["range_keys"])

    @property
    def n_loop_closures(self) -> int:
        return int(self.arrays.get("n_loop_closures", 0))

    def get_pose_chain_names(self):
        return self.arrays["pose_chain_names"]


def unconnected_variable_names(a: Dict[str, np.ndarray]):
    """``FactorGraphData.unconnected_variable_names`` (score/solve_score.py:28-32) from the flat arrays:
    variables that no odometry / loop-closure / range measurement and no landmark prior touches."""
    Np, Nl = len(a["pose_names"]), len(a["landmark_names"])
    touched = np.zeros(Np + Nl, dtype=bool)
    for key in ("rel_base", "rel_to", "rng_a", "rng_b"):
        touched[a[key]] = True
    touched[Np + a["lprior_lm"]] = True
    if touched.all():
        return []
    names = list(a["pose_names"]) + list(a["landmark_names"])
    return [names[i] for i in np.nonzero(~touched)[0]]


class NativeQP:
    """The assembled conic program held by the native library (duck-types ``assemble.ConicQP``)."""

    def __init__(self, lib: C.CDLL, handle: C.c_void_p):
        self._lib, self._h = lib, handle
        self.problem = ScoreProblem()
        if lib.score_assembled_view(handle, C.byref(self.problem)) != 0:
            raise RuntimeError(lib.score_last_error().decode())
        p = self.problem
        self.n, self.m, self.z = int(p.n), int(p.m), int(p.z)
        self.c0 = float(p.c0)
        self.block_size = int(p.block_size)
        self.rep_d, self.rep_n = int(p.rep_d), int(p.rep_n)
        self._cache = {}

    def _arr(self, ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(count,)).copy()

    def _get(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def P(self) -> sp.csr_matrix:
        def build():
            p = self.problem
            ptr = self._arr(p.P_rowptr, self.n + 1, np.int32)
            return sp.csr_matrix((self._arr(p.P_val, int(ptr[-1]), np.float64), self._arr(p.P_col, int(ptr[-1]), np.int32), ptr),
                                 shape=(self.n, self.n))
        return self._get("P", build)

    @property
    def A(self) -> sp.csr_matrix:
        def build():
            p = self.problem
            ptr = self._arr(p.A_rowptr, self.m + 1, np.int32)
            return sp.csr_matrix((self._arr(p.A_val, int(ptr[-1]), np.float64), self._arr(p.A_col, int(ptr[-1]), np.int32), ptr),
                                 shape=(self.m, self.n))
        return self._get("A", build)

    @property
    def q(self) -> np.ndarray:
        return self._get("q", lambda: self._arr(self.problem.q, self.n, np.float64))

    @property
    def b(self) -> np.ndarray:
        return self._get("b", lambda: self._arr(self.problem.b, self.m, np.float64))

    @property
    def soc_dims(self) -> np.ndarray:
        return self._get("soc", lambda: self._arr(self.problem.soc_dims, int(self.problem.n_soc), np.int32))

    @property
    def chain_ptr(self) -> np.ndarray:
        return self._get("cp", lambda: self._arr(self.problem.chain_ptr, int(self.problem.n_chains) + 1, np.int32))

    @property
    def node_cols(self) -> np.ndarray:
        def build():
            first = self._arr(self.problem.node_first_col, int(self.chain_ptr[-1]), np.int32)
            return (first[:, None] + np.arange(self.block_size, dtype=np.int32)[None, :]).ravel()
        return self._get("nc", build)

    def objective(self, x: np.ndarray) -> float:
        return 0.5 * float(np.einsum("i,i->", x, self.P @ x)) + float(np.einsum("i,i->", self.q, x)) + self.c0

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.score_assembled_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def score_graph_struct(a: Dict[str, np.ndarray], relaxation: int = 0) -> ScoreGraph:
    """``struct score_graph`` over the flat arrays of ``graph_arrays`` (borrowed: keep ``a`` alive).  Kept on the dict while its
    arrays are the same objects: fourteen ``ndarray.ctypes`` look-ups are 40 us a graph -- 0.65 ms of a 16-trial handle's create,
    under the interpreter lock."""
    ids = tuple(id(a[k]) for k in _STRUCT_ARRAYS)
    kept = a.get("_cstruct")
    if kept is not None and kept[0] == (int(relaxation), ids):
        return kept[1]
    g = _score_graph_struct(a, relaxation)
    a["_cstruct"] = ((int(relaxation), ids), g)
    return g


_STRUCT_ARRAYS = ("chain_len", "rel_base", "rel_to", "rel_t", "rel_R", "rel_kappa", "rel_tau", "rng_a", "rng_b", "rng_dist", "rng_prec",
                  "lprior_lm", "lprior_t", "lprior_prec")


def _score_graph_struct(a: Dict[str, np.ndarray], relaxation: int = 0) -> ScoreGraph:
    g = ScoreGraph()
    g.dim, g.relaxation = int(a["dim"]), int(relaxation)
    g.n_chains, g.chain_len = len(a["chain_len"]), a["chain_len"].ctypes.data_as(_i32p)
    g.n_landmarks = len(a["landmark_names"])
    g.n_rel = len(a["rel_base"])
    g.rel_base, g.rel_to = a["rel_base"].ctypes.data_as(_i32p), a["rel_to"].ctypes.data_as(_i32p)
    g.rel_t, g.rel_R = a["rel_t"].ctypes.data_as(_f64p), a["rel_R"].ctypes.data_as(_f64p)
    g.rel_kappa, g.rel_tau = a["rel_kappa"].ctypes.data_as(_f64p), a["rel_tau"].ctypes.data_as(_f64p)
    g.n_rng = len(a["rng_a"])
    g.rng_a, g.rng_b = a["rng_a"].ctypes.data_as(_i32p), a["rng_b"].ctypes.data_as(_i32p)
    g.rng_dist, g.rng_prec = a["rng_dist"].ctypes.data_as(_f64p), a["rng_prec"].ctypes.data_as(_f64p)
    g.n_lprior = len(a["lprior_lm"])
    g.lprior_lm, g.lprior_t = a["lprior_lm"].ctypes.data_as(_i32p), a["lprior_t"].ctypes.data_as(_f64p)
    g.lprior_prec = a["lprior_prec"].ctypes.data_as(_f64p)
    return g


def _read_back_maps(a: Dict[str, np.ndarray], relaxation: str, qp: "NativeQP") -> ScoreModel:
    """The ScoreModel around an assembled program: the same read-back maps as assemble.py (model space keeps the
    Gurobi layout; solver space: replica by replica, see score_assemble.hpp)."""
    d = int(a["dim"])
    D1, PB = d + 1, d * (d + 1)
    Np, Nl, Nr = len(a["pose_names"]), len(a["landmark_names"]), len(a["range_keys"])
    rw = 1 if relaxation == SOCP_RELAXATION else d
    lm_base = Np * PB
    rng_base = lm_base + Nl * d
    n_model = rng_base + Nr * rw
    # replica 0: every pose but the pinned one (pose 0 of chain 0), chain by chain = poses 1..Np-1 in order; then
    # the landmarks; then (QCQP) the range vectors' component 0.  Replica k: the same columns shifted.
    pose0 = (np.arange(1, Np, dtype=np.int64)[:, None] * PB + np.arange(D1, dtype=np.int64)[None, :]).ravel()
    lm0 = lm_base + np.arange(Nl, dtype=np.int64) * d
    rq0 = rng_base + np.arange(Nr, dtype=np.int64) * d if relaxation != SOCP_RELAXATION else np.zeros(0, np.int64)
    pieces = []
    for k in range(d):
        pieces += [pose0 + k * D1, lm0 + k]
        if relaxation != SOCP_RELAXATION:
            pieces.append(rq0 + k)
    if relaxation == SOCP_RELAXATION:
        pieces.append(rng_base + np.arange(Nr, dtype=np.int64))
    free_cols = np.concatenate(pieces)
    assert free_cols.size == qp.n, (free_cols.size, qp.n)
    ends = dist = None
    if Nr:
        va, vb = a["rng_a"].astype(np.int64), a["rng_b"].astype(np.int64)

        def tcol(v):
            pose = v < Np
            return np.where(pose, v * PB + d, lm_base + (v - Np) * d), np.where(pose, D1, 1)

        ta, sa = tcol(va)
        tb, sb = tcol(vb)
        ends = np.stack([ta, sa, tb, sb], axis=1)
        dist = a["rng_dist"]
    return ScoreModel(
        dim=d, relaxation=relaxation, qp=qp, n_model=n_model, free_cols=free_cols,
        fixed_cols=np.arange(PB), fixed_vals=np.hstack([np.eye(d), np.zeros((d, 1))]).ravel(),
        pose_names=a["pose_names"], landmark_names=a["landmark_names"], range_keys=a["range_keys"],
        lm_base=lm_base, rng_base=rng_base, rng_width=rw, range_ends=ends, range_dist=dist,
        pose_chain_names=a.get("pose_chain_names"),
    )


def graphs_connected(arrays: list, lib_path: Optional[str] = None) -> Optional[int]:
    """``score_graphs_connected``: None when every variable of every graph is touched by a measurement or a prior
    (score/solve_score.py:28-32), else the index of the first graph that has unconnected variables."""
    if not arrays:
        return None
    lib = load_library(lib_path)
    lib.score_graphs_connected.argtypes = [C.POINTER(ScoreGraph), C.c_int32]
    n = len(arrays)
    gs = (ScoreGraph * n)()
    keep = []
    for i, a in enumerate(arrays):
        g = score_graph_struct(a, 0)
        keep.append(g)
        C.memmove(C.byref(gs[i]), C.byref(g), C.sizeof(ScoreGraph))
    rc = lib.score_graphs_connected(gs, n)
    if rc < 0:
        raise RuntimeError(lib.score_last_error().decode())
    return None if rc == 0 else rc - 1


class GraphQP:
    """Sizes of the conic program of a graph whose model is built inside ``score_create_from_graphs`` (on the device): what
    ``ScoreModel`` needs of a ``ConicQP`` when nobody asks for the matrices."""

    def __init__(self, a: Dict[str, np.ndarray], relaxation: str):
        d = int(a["dim"])
        Np, Nl, Nr = len(a["pose_names"]), len(a["landmark_names"]), len(a["range_keys"])
        socp = relaxation == SOCP_RELAXATION
        self.rep_d, self.rep_n = d, (Np - 1) * (d + 1) + Nl + (0 if socp else Nr)
        self.n = d * self.rep_n + (Nr if socp else 0)
        self.m, self.z = Nr * (d + 1), 0
        self.block_size = d + 1


class GraphModel:
    """``ScoreModel``'s interface for a graph whose program is built inside ``score_create_from_graphs``.  The native column
    layout is regular (replica by replica: the pose entries, the landmark coordinates, the QCQP range components; then the SOCP
    distances), so a solution is read back with reshapes (``views``) instead of index maps; the maps themselves
    (``free_cols``, ``range_ends``: model space, the Gurobi layout) are made on first use -- per graph they cost more than
    the read-back itself, and a Monte-Carlo sweep never asks for them."""

    def __init__(self, a: Dict[str, np.ndarray], relaxation: str):
        check_valid_relaxation(relaxation)
        self.graph_arrays = a
        self.relaxation = relaxation
        self.dim = d = int(a["dim"])
        self.qp = GraphQP(a, relaxation)
        self.pose_names, self.landmark_names, self.range_keys = a["pose_names"], a["landmark_names"], a["range_keys"]
        self.pose_chain_names = a.get("pose_chain_names")
        Np, Nl, Nr = len(self.pose_names), len(self.landmark_names), len(self.range_keys)
        self.rng_width = 1 if relaxation == SOCP_RELAXATION else d
        self.lm_base = Np * d * (d + 1)
        self.rng_base = self.lm_base + Nl * d
        self.n_model = self.rng_base + Nr * self.rng_width
        self.range_dist = a["rng_dist"] if Nr else None
        self._full = None

    def _maps(self) -> ScoreModel:
        if self._full is None:
            self._full = _read_back_maps(self.graph_arrays, self.relaxation, self.qp)
        return self._full

    free_cols = property(lambda self: self._maps().free_cols)
    fixed_cols = property(lambda self: self._maps().fixed_cols)
    fixed_vals = property(lambda self: self._maps().fixed_vals)
    range_ends = property(lambda self: self._maps().range_ends)

    def expand(self, x_solver: np.ndarray) -> np.ndarray:
        return self._maps().expand(x_solver)

    def reduce(self, x_model: np.ndarray) -> np.ndarray:
        return self._maps().reduce(x_model)

    def pose_blocks(self, x_model: np.ndarray) -> np.ndarray:
        return self._maps().pose_blocks(x_model)

    def landmark_block(self, x_model: np.ndarray) -> np.ndarray:
        return self._maps().landmark_block(x_model)

    def range_block(self, x_model: np.ndarray) -> np.ndarray:
        return self._maps().range_block(x_model)

    def views(self, x_solver: np.ndarray):
        """(pose blocks (Np, d, d + 1) with the pinned pose, landmarks (Nl, d), range variables (Nr, 1 or d)) of a solver-space
        vector -- what ``pose_blocks / landmark_block / range_block`` give for ``expand(x_solver)``, by reshapes."""
        d, D1 = self.dim, self.dim + 1
        Np, Nl, Nr = len(self.pose_names), len(self.landmark_names), len(self.range_keys)
        n_rep = self.qp.rep_n
        X = x_solver[: d * n_rep].reshape(d, n_rep)
        blocks = np.empty((Np, d, D1))
        blocks[0] = np.hstack([np.eye(d), np.zeros((d, 1))])
        blocks[1:] = X[:, : (Np - 1) * D1].reshape(d, Np - 1, D1).transpose(1, 0, 2)
        lm0 = (Np - 1) * D1
        lms = np.ascontiguousarray(X[:, lm0 : lm0 + Nl].T)
        if self.relaxation == SOCP_RELAXATION:
            rng = x_solver[d * n_rep :].reshape(Nr, 1).copy()
        else:
            rng = np.ascontiguousarray(X[:, lm0 + Nl : lm0 + Nl + Nr].T)
        return blocks, lms, rng


def graph_model(a: Dict[str, np.ndarray], relaxation: str) -> GraphModel:
    """The read-back side of ``assemble_native`` without the program itself (``solver.ConicSolver.from_graphs`` builds it)."""
    return GraphModel(a, relaxation)


def assemble_native(data, relaxation: str = QCQP_RELAXATION, lib_path: Optional[str] = None, arrays: Optional[dict] = None) -> ScoreModel:
    check_valid_relaxation(relaxation)
    lib = load_library(lib_path)
    _bind(lib)
    a = arrays if arrays is not None else graph_arrays(data)
    g = score_graph_struct(a, 0 if relaxation == SOCP_RELAXATION else 1)
    h = C.c_void_p()
    if lib.score_assemble(C.byref(g), C.byref(h)) != 0:
        raise ValueError(lib.score_last_error().decode())
    return _read_back_maps(a, relaxation, NativeQP(lib, h))


def assemble_native_batch(arrays: list, relaxation: str = QCQP_RELAXATION, lib_path: Optional[str] = None) -> list:
    """``assemble_native`` for a list of flat-array graphs in ONE foreign call (``score_assemble_batch``: one graph per
    host thread of the library's team).  A thread that builds the models of its lock-step group this way leaves the
    interpreter lock once per group: with one call per graph from a pool of Python threads the lock's hand-overs, not
    the model construction, set the pace (64 four-robot graphs: 1.6 ms each in the library, 7-15 ms each as seen from
    the pool; profiles/scripts/r04_e2e_timeline.py)."""
    check_valid_relaxation(relaxation)
    if not arrays:
        return []
    lib = load_library(lib_path)
    _bind(lib)
    n = len(arrays)
    gs = (ScoreGraph * n)()
    keep = []
    for i, a in enumerate(arrays):
        g = score_graph_struct(a, 0 if relaxation == SOCP_RELAXATION else 1)
        keep.append(g)
        C.memmove(C.byref(gs[i]), C.byref(g), C.sizeof(ScoreGraph))
    hs = (C.c_void_p * n)()
    if lib.score_assemble_batch(gs, n, hs) != 0:
        raise ValueError(lib.score_last_error().decode())
    qps = [NativeQP(lib, C.c_void_p(hs[i])) for i in range(n)]
    return [_read_back_maps(a, relaxation, qp) for a, qp in zip(arrays, qps)]
