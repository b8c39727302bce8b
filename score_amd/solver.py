"""ctypes binding of the C ABI in include/score_hip.h.

``ConicSolver`` loads ``score_amd/csrc/libscore_hip.so`` (the HIP/gfx950
library).  There is no CPU fallback on the product path: if the library is
missing or no HIP device is usable, construction raises.  (Tests may point
``lib_path`` at the oracle's CPU twin, which implements the same ABI.)
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "csrc", "libscore_hip.so")

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class ScoreProblem(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32),
        ("P_rowptr", _i32p), ("P_col", _i32p), ("P_val", _f64p),
        ("q", _f64p), ("c0", C.c_double),
        ("A_rowptr", _i32p), ("A_col", _i32p), ("A_val", _f64p),
        ("b", _f64p),
        ("z", C.c_int32), ("n_soc", C.c_int32), ("soc_dims", _i32p),
        ("block_size", C.c_int32), ("n_chains", C.c_int32),
        ("chain_ptr", _i32p), ("node_first_col", _i32p),
        ("rep_d", C.c_int32), ("rep_n", C.c_int32),
    ]


class ScoreSettings(C.Structure):
    _fields_ = [
        ("eps_abs", C.c_double), ("eps_rel", C.c_double),
        ("max_iters", C.c_int32), ("check_interval", C.c_int32),
        ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
        ("scale_iters", C.c_int32), ("cg_iters", C.c_int32),
        ("adaptive_cg", C.c_int32), ("max_cg_iters", C.c_int32), ("cg_target", C.c_double),
        ("adaptive_rho", C.c_int32), ("adaptive_rho_interval", C.c_int32),
        ("adaptive_rho_tol", C.c_double),
        ("chain_radix", C.c_int32), ("device", C.c_int32),
        ("use_graph", C.c_int32), ("polish", C.c_int32), ("polish_start", C.c_double),
        ("polish_warmup", C.c_int32), ("verbose", C.c_int32), ("chain_split", C.c_int32), ("fac_fp32", C.c_int32),
    ]


class ScoreRefineInfo(C.Structure):
    _fields_ = [
        ("cost_initial", C.c_double), ("cost_final", C.c_double), ("grad_inf", C.c_double),
        ("iterations", C.c_int32), ("linear_solves", C.c_int32), ("pcg_iters", C.c_int32),
        ("setup_ms", C.c_double), ("solve_ms", C.c_double),
    ]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class ScoreInfo(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("iters", C.c_int32), ("cg_iters", C.c_int32), ("rho_updates", C.c_int32),
        ("rho", C.c_double), ("pobj", C.c_double), ("dobj", C.c_double),
        ("res_pri", C.c_double), ("res_dual", C.c_double), ("gap", C.c_double),
        ("setup_ms", C.c_double), ("solve_ms", C.c_double), ("kkt_bytes", C.c_double),
        ("newton_iters", C.c_int32), ("newton_cg_iters", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


STATUS_NAMES = {0: "unsolved", 1: "solved", 2: "max_iters", 3: "numerical"}

# every symbol include/score_hip.h declares
ABI_SYMBOLS = [
    "score_assemble", "score_assemble_batch", "score_assembled_view", "score_assembled_free", "score_round_to_so",
    "score_default_settings", "score_create", "score_create_batch", "score_create_from_graphs", "score_read_estimates", "score_graphs_connected", "score_dims", "score_solve",
    "score_reset", "score_solve_steps", "score_newton_steps", "score_linear_create", "score_linear_solve", "score_refine_create", "score_refine_run", "score_refine_destroy", "score_time_kkt_apply", "score_time_iteration", "score_debug_time", "score_debug_get", "score_destroy",
    "score_trim_caches", "score_host_counters", "score_last_error", "score_backend", "score_abi_version",
    "score_generate_manhattan", "score_generated_graph", "score_generated_truth", "score_generated_free", "score_create_from_generated",
]

ABI_VERSION = 7  # SCORE_ABI_VERSION of include/score_hip.h this binding's structs follow


def load_library(path: Optional[str] = None) -> C.CDLL:
    path = path or HIP_LIB
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "The SCORE solver has no CPU fallback."
        )
    lib = C.CDLL(path)
    lib.score_default_settings.argtypes = [C.POINTER(ScoreSettings)]
    lib.score_default_settings.restype = None
    lib.score_create.argtypes = [C.POINTER(ScoreProblem), C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
    lib.score_create_batch.argtypes = [C.POINTER(ScoreProblem), C.c_int32, C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
    lib.score_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.score_solve.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, C.POINTER(ScoreInfo)]
    lib.score_reset.argtypes = [C.c_void_p]
    lib.score_solve_steps.argtypes = [C.c_void_p, C.c_int32, _f64p, _f64p, _f64p, C.POINTER(ScoreInfo)]
    lib.score_newton_steps.argtypes = [C.c_void_p, C.c_int32, _f64p, _f64p, _f64p, C.POINTER(ScoreInfo)]
    lib.score_time_kkt_apply.argtypes = [C.c_void_p, C.c_int32, _f64p, _f64p]
    lib.score_debug_time.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, _f64p]
    lib.score_time_iteration.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f64p, C.c_int32]
    lib.score_time_iteration.restype = C.c_int
    lib.score_debug_get.argtypes = [C.c_void_p, C.c_char_p, _f64p, C.c_int64]
    lib.score_debug_get.restype = C.c_int64
    lib.score_destroy.argtypes = [C.c_void_p]
    lib.score_destroy.restype = None
    lib.score_linear_create.argtypes = [C.POINTER(ScoreProblem), C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
    lib.score_linear_solve.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, C.c_double, C.c_int32, C.POINTER(C.c_int32), _f64p]
    lib.score_refine_create.argtypes = [C.c_void_p, C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
    lib.score_refine_run.argtypes = [C.c_void_p, _f64p, _f64p, C.c_int32, C.c_double, _f64p, _f64p, C.POINTER(ScoreRefineInfo)]
    lib.score_refine_destroy.argtypes = [C.c_void_p]
    lib.score_refine_destroy.restype = None
    lib.score_round_to_so.argtypes = [C.c_int32, C.c_int64, _f64p, _f64p, C.POINTER(C.c_int32), C.c_int32]
    lib.score_round_to_so.restype = C.c_int
    lib.score_trim_caches.argtypes = []
    lib.score_trim_caches.restype = C.c_int64
    lib.score_host_counters.argtypes = [C.POINTER(C.c_double), C.c_int32]
    lib.score_host_counters.restype = C.c_int32
    lib.score_last_error.restype = C.c_char_p
    lib.score_backend.restype = C.c_char_p
    lib.score_abi_version.restype = C.c_int32
    want = ABI_VERSION * 1000 + C.sizeof(ScoreProblem)
    if lib.score_abi_version() != want:
        raise RuntimeError(f"{path}: ABI {lib.score_abi_version()} (version * 1000 + sizeof(score_problem)), this binding expects {want}: "
                           "rebuild the library or update the binding")
    return lib


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


@dataclass
class ConicSolution:
    x: np.ndarray
    y: np.ndarray
    s: np.ndarray
    info: dict

    @property
    def solved(self) -> bool:
        return self.info["status"] == 1


def host_counters(lib_path: Optional[str] = None) -> dict:
    """What the library's threads spent waiting for the device since the process started (``score_host_counters``):
    milliseconds spinning (CPU busy), milliseconds asleep (economy waits), number of waits and of sleeps."""
    v = (C.c_double * 4)()
    load_library(lib_path).score_host_counters(v, 4)
    return dict(spin_ms=float(v[0]), sleep_ms=float(v[1]), waits=int(v[2]), sleeps=int(v[3]))


def trim_caches(lib_path: Optional[str] = None) -> int:
    """Release the device / pinned blocks and streams the library keeps parked between handles
    (``score_trim_caches``); returns the bytes freed."""
    return int(load_library(lib_path).score_trim_caches())


class ConicSolver:
    """A batch of conic QPs (``score_amd.assemble.ConicQP``) resident on one GPU."""

    @classmethod
    def from_graphs(cls, arrays: Sequence[dict], relaxation: int, settings: Optional[dict] = None, lib_path: Optional[str] = None) -> "ConicSolver":
        """A handle straight from factor graphs (``score_create_from_graphs``): ``arrays`` are the flat arrays of
        ``native.graph_arrays``, ``relaxation`` 0 = "SOCP" / 1 = "QCQP".  Model construction (the reference's
        ``initialize_model``, gurobi_utils.py:173-187) happens inside the call, on the device; the unknowns are ordered as
        ``score_assemble`` orders them, so ``native.graph_model`` supplies the read-back maps."""
        from .native import ScoreGraph, score_graph_struct  # (native imports this module)

        self = cls.__new__(cls)
        self.lib = load_library(lib_path)
        self.lib.score_create_from_graphs.argtypes = [C.POINTER(ScoreGraph), C.c_int32, C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
        self.count = len(arrays)
        st = ScoreSettings()
        self.lib.score_default_settings(C.byref(st))
        for k, v in (settings or {}).items():
            if not hasattr(st, k):
                raise ValueError(f"unknown solver setting {k}")
            setattr(st, k, v)
        self.settings = st
        # worlds of ONE generated batch (score_amd.generate), in a row: the library builds the handle from the arrays the
        # generator left on the device (score_create_from_generated) -- the same program, nothing of a world uploaded again
        # (and no ``struct score_graph`` made here: 0.1 ms a graph under the interpreter lock)
        owner = arrays[0].get("_owner") if self.count else None
        first = arrays[0].get("_index") if owner is not None else None
        resident = (owner is not None and getattr(owner, "_h", None) and getattr(getattr(owner, "lib", None), "_handle", None) == self.lib._handle
                    and all(a.get("_owner") is owner and a.get("_index") == first + i for i, a in enumerate(arrays)))
        gs = None if resident else (ScoreGraph * self.count)()
        self.ns, self.ms = [], []
        for i, a in enumerate(arrays):
            if not resident:
                g = score_graph_struct(a, int(relaxation))
                C.memmove(C.byref(gs[i]), C.byref(g), C.sizeof(ScoreGraph))
            d = int(a["dim"])
            Np, Nl, Nr = len(a["pose_names"]), len(a["landmark_names"]), len(a["rng_a"])
            n_rep = (Np - 1) * (d + 1) + Nl + (Nr if relaxation else 0)
            self.ns.append(d * n_rep + (0 if relaxation else Nr))
            self.ms.append(Nr * (d + 1))
            est_per = getattr(self, "_est_per", None)
            if est_per is None:
                est_per = self._est_per = []
            est_per.append((Np, Nl, Nr))
            self._est_dims = (d, int(relaxation), est_per)
        self._keep = []
        self._h = C.c_void_p()
        if resident:
            self.lib.score_create_from_generated.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ScoreSettings), C.POINTER(C.c_void_p)]
            rc = self.lib.score_create_from_generated(owner._h, int(first), self.count, int(relaxation), C.byref(st), C.byref(self._h))
            self._keep.append(owner)
        else:
            rc = self.lib.score_create_from_graphs(gs, self.count, C.byref(st), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise ValueError(f"score_create_from_graphs failed: {self.lib.score_last_error().decode()}")
        self.n_total, self.m_total = sum(self.ns), sum(self.ms)
        nt, mt, cnt = C.c_int64(), C.c_int64(), C.c_int32()
        self.lib.score_dims(self._h, C.byref(nt), C.byref(mt), C.byref(cnt))
        assert (nt.value, mt.value, cnt.value) == (self.n_total, self.m_total, self.count), "score_create_from_graphs: sizes differ from the column layout"
        return self

    def __init__(self, qps: Sequence, settings: Optional[dict] = None, lib_path: Optional[str] = None):
        if not isinstance(qps, (list, tuple)):
            qps = [qps]
        self.lib = load_library(lib_path)
        self.count = len(qps)
        self.ns = [int(qp.n) for qp in qps]
        self.ms = [int(qp.m) for qp in qps]
        st = ScoreSettings()
        self.lib.score_default_settings(C.byref(st))
        for k, v in (settings or {}).items():
            if not hasattr(st, k):
                raise ValueError(f"unknown solver setting {k}")
            setattr(st, k, v)
        self.settings = st
        self._keep = []  # borrowed arrays must outlive score_create
        probs = (ScoreProblem * self.count)()
        for i, qp in enumerate(qps):
            native = getattr(qp, "problem", None)
            if isinstance(native, ScoreProblem):
                # assembled by the library itself (score_amd.native): hand its view over as it is
                C.memmove(C.byref(probs[i]), C.byref(native), C.sizeof(ScoreProblem))
                self._keep.append(qp)
                continue
            P = qp.P.tocsr()
            A = qp.A.tocsr()
            if not P.has_sorted_indices:
                P = P.sorted_indices()
            if not A.has_sorted_indices:
                A = A.sorted_indices()
            arrs = dict(
                Pp=_i32(P.indptr), Pc=_i32(P.indices), Pv=_f64(P.data), q=_f64(qp.q),
                Ap=_i32(A.indptr), Ac=_i32(A.indices), Av=_f64(A.data), b=_f64(qp.b),
                soc=_i32(qp.soc_dims), cp=_i32(qp.chain_ptr), nc=_i32(qp.node_cols[:: max(1, qp.block_size)] if qp.block_size else qp.node_cols),
            )
            if qp.block_size:
                nodes = np.asarray(qp.node_cols).reshape(-1, qp.block_size)
                if nodes.size and not np.all(np.diff(nodes, axis=1) == 1):
                    raise ValueError("chain nodes must own consecutive columns")
            self._keep.append(arrs)
            p = probs[i]
            p.n, p.m = qp.n, qp.m
            p.P_rowptr, p.P_col, p.P_val = _ptr(arrs["Pp"], _i32p), _ptr(arrs["Pc"], _i32p), _ptr(arrs["Pv"], _f64p)
            p.q, p.c0 = _ptr(arrs["q"], _f64p), float(qp.c0)
            p.A_rowptr, p.A_col, p.A_val = _ptr(arrs["Ap"], _i32p), _ptr(arrs["Ac"], _i32p), _ptr(arrs["Av"], _f64p)
            p.b = _ptr(arrs["b"], _f64p)
            p.z, p.n_soc, p.soc_dims = int(qp.z), int(len(qp.soc_dims)), _ptr(arrs["soc"], _i32p)
            nch = int(len(qp.chain_ptr) - 1) if qp.block_size else 0
            p.block_size, p.n_chains = int(qp.block_size), nch
            p.chain_ptr, p.node_first_col = _ptr(arrs["cp"], _i32p), _ptr(arrs["nc"], _i32p)
            p.rep_d, p.rep_n = int(getattr(qp, "rep_d", 0)), int(getattr(qp, "rep_n", 0))
        self._h = C.c_void_p()
        rc = self.lib.score_create_batch(probs, self.count, C.byref(st), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise RuntimeError(f"score_create_batch failed: {self.lib.score_last_error().decode()}")
        self._keep.clear()
        self.n_total, self.m_total = sum(self.ns), sum(self.ms)

    @property
    def backend(self) -> str:
        return self.lib.score_backend().decode()

    def _split(self, x, y, s, infos) -> List[ConicSolution]:
        out, xo, ro = [], 0, 0
        backend = self.backend  # which library produced the numbers ("hip-gfx950"; "cpu-twin" only in tests)
        for i in range(self.count):
            out.append(ConicSolution(
                x[xo : xo + self.ns[i]].copy(), y[ro : ro + self.ms[i]].copy(), s[ro : ro + self.ms[i]].copy(),
                dict(infos[i].as_dict(), backend=backend),
            ))
            xo += self.ns[i]
            ro += self.ms[i]
        return out

    def solve(self) -> List[ConicSolution]:
        x = np.empty(self.n_total); y = np.empty(self.m_total); s = np.empty(self.m_total)
        infos = (ScoreInfo * self.count)()
        rc = self.lib.score_solve(self._h, _ptr(x, _f64p), _ptr(y, _f64p), _ptr(s, _f64p), infos)
        if rc != 0:
            raise RuntimeError(f"score_solve failed: {self.lib.score_last_error().decode()}")
        return self._split(x, y, s, infos)

    def solve_estimates(self, qcqp_directions: bool = False, return_x: bool = False):
        """Cold-start solve of a handle made by ``from_graphs``, the estimate read back in the reference's own shapes straight
        from the device (``score_read_estimates``; replaces ``get_variable_values``, gurobi_utils.py:114-136) -- no x / y / s
        copies, no index maps.  Returns ``(infos, estimates)``: per problem a dict of ``info`` and a tuple
        ``(poses (Np, d+1, d+1), relaxed (Np, d, d+1), landmarks (Nl, d), ranges (Nr, 1 | d), degenerate (Np,))`` of views into
        the handle-wide arrays."""
        dims = getattr(self, "_est_dims", None)
        if dims is None:
            raise RuntimeError("solve_estimates: the handle was not made by ConicSolver.from_graphs")
        d, relax, per = dims
        infos = (ScoreInfo * self.count)()
        x = np.empty(self.n_total) if return_x else None  # (tests: the same solve's solver-space solution beside the estimate)
        if self.lib.score_solve(self._h, _ptr(x, _f64p) if return_x else None, None, None, infos) != 0:
            raise RuntimeError(f"score_solve failed: {self.lib.score_last_error().decode()}")
        rw = d if (relax or qcqp_directions) else 1
        nP, nL, nR = sum(p[0] for p in per), sum(p[1] for p in per), sum(p[2] for p in per)
        T = np.empty((nP, d + 1, d + 1)); B = np.empty((nP, d, d + 1)); Lm = np.empty((nL, d)); Rg = np.empty((nR, rw))
        flags = np.empty(nP, dtype=np.int32)
        self.lib.score_read_estimates.argtypes = [C.c_void_p, C.c_int32, _f64p, _f64p, _f64p, _f64p, C.POINTER(C.c_int32)]
        if self.lib.score_read_estimates(self._h, 1 if qcqp_directions else 0, _ptr(T, _f64p), _ptr(B, _f64p), _ptr(Lm, _f64p), _ptr(Rg, _f64p),
                                         flags.ctypes.data_as(C.POINTER(C.c_int32))) != 0:
            raise RuntimeError(f"score_read_estimates failed: {self.lib.score_last_error().decode()}")
        backend = self.backend
        out_i, out_e, po, lo, ro = [], [], 0, 0, 0
        for i, (np_, nl_, nr_) in enumerate(per):
            out_i.append(dict(infos[i].as_dict(), backend=backend))
            out_e.append((T[po : po + np_], B[po : po + np_], Lm[lo : lo + nl_], Rg[ro : ro + nr_], flags[po : po + np_]))
            po += np_; lo += nl_; ro += nr_
        if return_x:
            return out_i, out_e, x
        return out_i, out_e

    def reset(self) -> None:
        if self.lib.score_reset(self._h) != 0:
            raise RuntimeError(self.lib.score_last_error().decode())

    def steps(self, iters: int) -> List[ConicSolution]:
        x = np.empty(self.n_total); y = np.empty(self.m_total); s = np.empty(self.m_total)
        infos = (ScoreInfo * self.count)()
        rc = self.lib.score_solve_steps(self._h, int(iters), _ptr(x, _f64p), _ptr(y, _f64p), _ptr(s, _f64p), infos)
        if rc != 0:
            raise RuntimeError(f"score_solve_steps failed: {self.lib.score_last_error().decode()}")
        return self._split(x, y, s, infos)

    def newton_steps(self, iters: int) -> List[ConicSolution]:
        """At most ``iters`` Newton iterations of the polish from the current iterate, then a snapshot."""
        x = np.empty(self.n_total); y = np.empty(self.m_total); s = np.empty(self.m_total)
        infos = (ScoreInfo * self.count)()
        rc = self.lib.score_newton_steps(self._h, int(iters), _ptr(x, _f64p), _ptr(y, _f64p), _ptr(s, _f64p), infos)
        if rc != 0:
            raise RuntimeError(f"score_newton_steps failed: {self.lib.score_last_error().decode()}")
        return self._split(x, y, s, infos)

    def time_kkt_apply(self, reps: int = 200):
        ms, by = C.c_double(), C.c_double()
        if self.lib.score_time_kkt_apply(self._h, int(reps), C.byref(ms), C.byref(by)) != 0:
            raise RuntimeError(self.lib.score_last_error().decode())
        return ms.value, by.value

    ITERATION_KERNELS = ("rhs", "prec_init", "kp", "prec_step", "kpb", "cone")

    def time_iteration(self, warmup: int = 50, iters: int = 200, dispatch: bool = False):
        """In-loop microseconds of the six kernels of one ADMM iteration on the device wall clock
        (first workgroup in to last workgroup out).  With ``dispatch=True`` a second dict holds the
        begin-to-end time of each DISPATCH (start/stop HIP events bound to the launches on the
        handle's stream: what ``rocprofv3 --kernel-trace`` reports).  Resets and advances the iterates."""
        us = np.zeros(12)
        if self.lib.score_time_iteration(self._h, int(warmup), int(iters), _ptr(us, _f64p), 1 if dispatch else 0) != 0:
            raise RuntimeError(self.lib.score_last_error().decode())
        dev = dict(zip(self.ITERATION_KERNELS, us[:6].tolist()))
        if not dispatch:
            return dev
        return dev, dict(zip(self.ITERATION_KERNELS, us[6:].tolist()))

    def debug_time(self, kernel: str, reps: int = 200) -> float:
        ms = C.c_double()
        if self.lib.score_debug_time(self._h, kernel.encode(), int(reps), C.byref(ms)) != 0:
            raise RuntimeError(self.lib.score_last_error().decode())
        return ms.value

    def debug_get(self, name: str) -> np.ndarray:
        sz = self.lib.score_debug_get(self._h, name.encode(), None, 0)
        if sz < 0:
            raise KeyError(name)
        out = np.empty(sz)
        self.lib.score_debug_get(self._h, name.encode(), _ptr(out, _f64p), sz)
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.score_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LinearSolver:
    """Chain-preconditioned PCG on the GPU for SPD systems on a fixed sparsity pattern
    (``score_linear_create`` / ``score_linear_solve``): the linear solves of the local refinement that
    follows SCORE (score_amd/refine.py).  ``pattern``: scipy CSR with sorted indices and a full diagonal;
    ``chain_ptr`` / ``node_first_col`` / ``block_size``: the block-tridiagonal hint (one chain per robot,
    one node per pose)."""

    def __init__(self, pattern, chain_ptr, node_first_col, block_size: int, settings: Optional[dict] = None,
                 lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        P = pattern.tocsr()
        if not P.has_sorted_indices:
            P = P.sorted_indices()
        self.n = int(P.shape[0])
        self.nnz = int(P.nnz)
        self.indptr, self.indices = _i32(P.indptr), _i32(P.indices)
        st = ScoreSettings()
        self.lib.score_default_settings(C.byref(st))
        for k, v in (settings or {}).items():
            if not hasattr(st, k):
                raise ValueError(f"unknown solver setting {k}")
            setattr(st, k, v)
        self.settings = st
        cp, nc = _i32(chain_ptr), _i32(node_first_col)
        zero_i, zero_d = np.zeros(1, np.int32), np.zeros(1)
        p = ScoreProblem()
        p.n, p.m = self.n, 0
        p.P_rowptr, p.P_col, p.P_val = _ptr(self.indptr, _i32p), _ptr(self.indices, _i32p), None
        p.q, p.c0 = None, 0.0
        p.A_rowptr, p.A_col, p.A_val, p.b = _ptr(zero_i, _i32p), _ptr(zero_i, _i32p), _ptr(zero_d, _f64p), _ptr(zero_d, _f64p)
        p.z, p.n_soc, p.soc_dims = 0, 0, _ptr(zero_i, _i32p)
        p.block_size, p.n_chains = int(block_size), int(len(cp) - 1)
        p.chain_ptr, p.node_first_col = _ptr(cp, _i32p), _ptr(nc, _i32p)
        self._h = C.c_void_p()
        rc = self.lib.score_linear_create(C.byref(p), C.byref(st), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise RuntimeError(f"score_linear_create failed: {self.lib.score_last_error().decode()}")

    def solve(self, values, rhs, rel_tol: float = 1e-8, max_iters: int = 500, residual: bool = False):
        """x with K x = rhs; returns (x, info) with info = {converged, iters[, rel_residual]}."""
        v, b = _f64(values), _f64(rhs)
        if v.shape != (self.nnz,) or b.shape != (self.n,):
            raise ValueError("values / rhs do not match the pattern")
        x = np.empty(self.n)
        used = C.c_int32(0)
        rr = C.c_double(0.0)
        rc = self.lib.score_linear_solve(self._h, _ptr(v, _f64p), _ptr(b, _f64p), _ptr(x, _f64p), float(rel_tol), int(max_iters),
                                         C.byref(used), C.byref(rr) if residual else None)
        if rc < 0:
            raise RuntimeError(f"score_linear_solve failed: {self.lib.score_last_error().decode()}")
        info = {"converged": rc == 0, "iters": int(used.value)}
        if residual:
            info["rel_residual"] = float(rr.value)
        return x, info

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.score_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
