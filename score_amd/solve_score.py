"""Drop-in ``solve_score`` for the SCORE convex relaxation on MI355X.

Mirrors ``score/solve_score.py:54-86`` of the reference:

    solve_score(data: FactorGraphData, relaxation_type: str = "QCQP") -> SolverResults

check the graph (:28-32) -> build the model (gurobi_utils.py:173-187; here
``score_amd.assemble``) -> optimise (here: the HIP ADMM solver behind the C ABI
of include/score_hip.h instead of ``model.optimize()``) -> extract results with
SO(d)-rounded rotations (gurobi_utils.py:114-136, :190-203).

Relaxations.  "SOCP" and "QCQP" are the same convex program after minimising
out the per-range auxiliary variable (min_{d>=|D|} w(d-dist)^2 =
w*max(0,|D|-dist)^2 = min_{|r|<=1} w|D - dist*r|^2), so they
share their optimal poses and landmarks.  By default a "QCQP" request is
solved through that equivalent SOCP and the optimal ``r_ij`` are reconstructed
in closed form (r = D / max(|D|, dist)); ``qcqp_mode="direct"`` hands the QCQP
cone program itself to the solver.
"""
from __future__ import annotations

import logging
import os
import time
from typing import List, Optional, Sequence

import numpy as np

from . import compat
from .assemble import (
    ACCEPTABLE_RELAXATIONS,
    QCQP_RELAXATION,
    SOCP_RELAXATION,
    ScoreModel,
    assemble,
    check_valid_relaxation,
)
from .rounding import round_to_special_orthogonal
from .solver import ConicSolver

logger = logging.getLogger(__name__)

DEFAULT_SOLVER_SETTINGS = dict(eps_abs=1e-7, eps_rel=1e-7, max_iters=20000)


def _check_factor_graph(data) -> None:
    """score/solve_score.py:28-32."""
    unconnected_variables = data.unconnected_variable_names
    assert len(unconnected_variables) == 0, f"Found {unconnected_variables} unconnected variables. "


def _qcqp_dists_from_socp(model: ScoreModel, x_model: np.ndarray, data) -> np.ndarray:
    """Optimal QCQP range directions for fixed translations: r = D / max(|D|, dist)."""
    d = model.dim
    nr = len(model.range_keys)
    if nr == 0:
        return np.zeros((0, d))
    k = np.arange(d)
    e = model.range_ends
    delta = x_model[e[:, 0:1] + e[:, 1:2] * k] - x_model[e[:, 2:3] + e[:, 3:4] * k]
    den = np.maximum(np.sqrt(np.einsum("ij,ij->i", delta, delta)), model.range_dist)
    out = np.zeros((nr, d))
    # (a range measured as exactly 0 leaves r free -- w |D - 0 r|^2: every path returns r = 0 there)
    np.divide(delta, den[:, None], out=out, where=(den[:, None] > 0) & (np.asarray(model.range_dist)[:, None] > 0))
    return out


def extract_solver_results(
    model: ScoreModel, x_solver: np.ndarray, data, total_time: float, solved: bool,
    requested_relaxation: str, info: Optional[dict] = None, lib=None, device: int = 0,
) -> compat.SolverResults:
    """gurobi_utils.py:190-203 + VariableCollection.get_variable_values (:114-136)."""
    d = model.dim
    fast = getattr(model, "views", None)  # (native.GraphModel: the regular native layout, read back by reshapes)
    if fast is not None:
        blocks, lm_vals, rng_vals = fast(x_solver)
    else:
        xm = model.expand(x_solver)
        blocks = model.pose_blocks(xm)  # (Np, d, d+1)
        lm_vals = model.landmark_block(xm).copy()
    R = round_to_special_orthogonal(blocks[:, :, :d], lib=lib, device=device)  # lib: on the device
    T = np.zeros((blocks.shape[0], d + 1, d + 1))
    T[:, d, d] = 1.0
    T[:, :d, :d] = R
    T[:, :d, d] = blocks[:, :, d]
    # dict-like views over the stacked arrays (compat.ArrayDict): no per-pose Python objects
    poses = compat.ArrayDict(model.pose_names, T)
    landmarks = compat.ArrayDict(model.landmark_names, lm_vals)
    if requested_relaxation == model.relaxation:
        dists = compat.ArrayDict(model.range_keys, rng_vals if fast is not None else model.range_block(xm).copy())
    elif fast is not None:  # QCQP answered through the SOCP: r = D / max(|D|, dist) from the stacked translations
        a = model.graph_arrays
        tr = np.concatenate([blocks[:, :, d], lm_vals]) if len(lm_vals) else blocks[:, :, d]
        delta = tr[a["rng_a"]] - tr[a["rng_b"]]
        den = np.maximum(np.sqrt(np.einsum("ij,ij->i", delta, delta)), a["rng_dist"]) if len(delta) else np.zeros(0)
        dv = np.zeros((len(delta), d))
        np.divide(delta, den[:, None], out=dv, where=(den[:, None] > 0) & (np.asarray(a["rng_dist"])[:, None] > 0))
        dists = compat.ArrayDict(model.range_keys, dv)
    else:  # QCQP answered through the SOCP
        dists = compat.ArrayDict(model.range_keys, _qcqp_dists_from_socp(model, xm, data))
    values = compat.VariableValues(d, poses, landmarks, dists)
    return compat.SolverResults(
        variables=values, total_time=total_time, solved=solved,
        pose_chain_names=model.pose_chain_names if model.pose_chain_names is not None else data.get_pose_chain_names(),
        solver_cost=(info or {}).get("pobj"), info=info,
        relaxed_poses=compat.ArrayDict(model.pose_names, blocks if fast is not None else blocks.copy()),
    )


def _runtime_s(info: dict) -> float:
    """``SolverResults.total_time``: the reference reports ``model.Runtime`` (gurobi_utils.py:194) -- Gurobi's optimize() wall
    time, presolve, ordering and factorisation included.  The counterpart here is score_create (model construction, equilibration,
    K, factors: ``setup_ms``) + the solve (``solve_ms``), both of the handle the graph was solved in (a lock-step group reports its
    handle's times for every member); both stay in ``SolverResults.info``."""
    return (float(info.get("setup_ms", 0.0)) + float(info.get("solve_ms", 0.0))) * 1e-3


def _loop_closure_settings(settings: dict, user: Optional[dict], has_loop_closures: bool, lib_path: Optional[str]) -> None:
    """Loop closures (gurobi_utils.py:407-430) are stiff couplings outside the per-robot chains.  Since round 6 both preconditioners
    carry them (csrc/score_link.hpp: the ADMM loop's K and the Newton matrix; the oracle's CPU twin restates the correction), and
    such graphs run with the default 2 PCG iterations per KKT solve -- measured on the twin, 2 x 397 poses with 2 loop closures,
    ADMM alone: 3 900 PCG iterations against 37 600 with the 16 per step of rounds 1-5.  Only with the correction switched off
    (SCORE_NO_LINKS) does a run in which the ADMM loop has to converge by itself get those 16 back."""
    if not has_loop_closures or "cg_iters" in (user or {}) or not os.environ.get("SCORE_NO_LINKS"):
        return
    alone = not int(settings.get("polish", 1)) or bool(os.environ.get("SCORE_QCQP_PLAIN"))
    if not alone:
        from .solver import load_library

        alone = load_library(lib_path).score_backend().decode() != "hip-gfx950"
    if alone:
        settings.update(cg_iters=16, cg_target=0.1)


def _resolve_args(args, relaxation_type):
    """Accept both the reference signature ``solve_score(data, relaxation)`` and
    the stale example's ``solve_score(data, solver_params, relaxation)``
    (examples/solve_goats_example_score.py:42-44)."""
    params = None
    for a in args:
        if isinstance(a, str):
            relaxation_type = a
        elif a is not None:
            params = a
    return params, relaxation_type


def _model_for(data, relaxation_type: str, qcqp_mode: str, lib_path: Optional[str] = None, assembler: str = "native") -> ScoreModel:
    """Model construction (gurobi_utils.py:173-187).  ``assembler="native"`` (default): the C++
    assembler behind the C ABI (score_assemble, score_amd/native.py); ``"python"``: the NumPy/SciPy
    one (score_amd/assemble.py) -- same program, same column layout."""
    relax = SOCP_RELAXATION if (relaxation_type == QCQP_RELAXATION and qcqp_mode == "via_socp") else relaxation_type
    if assembler == "native":
        from .native import ArrayGraph, assemble_native, cached_graph_arrays, unconnected_variable_names

        arrays = data.arrays if isinstance(data, ArrayGraph) else cached_graph_arrays(data)
        # score/solve_score.py:28-32, evaluated on the arrays just extracted (one pass over the graph objects)
        unconnected_variables = unconnected_variable_names(arrays)
        assert len(unconnected_variables) == 0, f"Found {unconnected_variables} unconnected variables. "
        return assemble_native(data, relax, lib_path=lib_path, arrays=arrays)
    if assembler != "python":
        raise ValueError(f"assembler {assembler} is not supported")
    return assemble(data, relax)


def _models_for(datas: Sequence, relaxation_type: str, qcqp_mode: str, lib_path: Optional[str] = None, assembler: str = "native") -> List[ScoreModel]:
    """``_model_for`` for a list of graphs; the native assembler takes them in one foreign call
    (``native.assemble_native_batch`` -> ``score_assemble_batch``: one graph per host thread of the library);
    ``assembler="device"``: the read-back maps only -- the model itself is built inside ``score_create_from_graphs``."""
    if assembler == "device":
        from .native import ArrayGraph, cached_graph_arrays, graph_model, unconnected_variable_names

        relax = SOCP_RELAXATION if (relaxation_type == QCQP_RELAXATION and qcqp_mode == "via_socp") else relaxation_type
        # (objects -> flat arrays once per graph object: kept on it while its lists stand, native.cached_graph_arrays)
        arrays = [data.arrays if isinstance(data, ArrayGraph) else cached_graph_arrays(data) for data in datas]
        # score/solve_score.py:28-32 on the flat arrays: one foreign call for the whole group (score_graphs_connected); the names
        # for the reference's message only when a graph fails
        from .native import graphs_connected

        bad = graphs_connected(arrays, lib_path=lib_path)
        if bad is not None:
            unconnected_variables = unconnected_variable_names(arrays[bad])
            assert len(unconnected_variables) == 0, f"Found {unconnected_variables} unconnected variables. "
        return [graph_model(a, relax) for a in arrays]  # (model.graph_arrays: what ConicSolver.from_graphs hands to the library)
    if assembler != "native":
        out = []
        for data in datas:
            _check_factor_graph(data)
            out.append(_model_for(data, relaxation_type, qcqp_mode, lib_path, assembler))
        return out
    from .native import ArrayGraph, assemble_native_batch, cached_graph_arrays, unconnected_variable_names

    relax = SOCP_RELAXATION if (relaxation_type == QCQP_RELAXATION and qcqp_mode == "via_socp") else relaxation_type
    arrays = [data.arrays if isinstance(data, ArrayGraph) else cached_graph_arrays(data) for data in datas]
    for a in arrays:  # score/solve_score.py:28-32, on the flat arrays
        unconnected_variables = unconnected_variable_names(a)
        assert len(unconnected_variables) == 0, f"Found {unconnected_variables} unconnected variables. "
    return assemble_native_batch(arrays, relax, lib_path=lib_path)


def solve_score(
    data, *args, relaxation_type: str = QCQP_RELAXATION, qcqp_mode: str = "via_socp",
    solver_settings: Optional[dict] = None, lib_path: Optional[str] = None, assembler: str = "device",
) -> compat.SolverResults:
    """MLE estimate of poses and landmarks from the SCORE relaxation.

    args:
        data (FactorGraphData): the data describing the problem
        relaxation_type (str): "QCQP" (default, as the reference) or "SOCP"
    returns:
        SolverResults: poses are homogeneous (d+1)x(d+1) matrices with rotations
        rounded to SO(d); ``solved`` is False when the solver did not reach its
        tolerances (no exception, as in the reference).
    """
    _params, relaxation_type = _resolve_args(args, relaxation_type)
    return solve_score_batch([data], relaxation_type=relaxation_type, qcqp_mode=qcqp_mode,
                             solver_settings=solver_settings, lib_path=lib_path, assembler=assembler)[0]


def solve_score_batch(
    datas: Sequence, relaxation_type: str = QCQP_RELAXATION, qcqp_mode: str = "via_socp",
    solver_settings: Optional[dict] = None, lib_path: Optional[str] = None, lockstep: Optional[bool] = None,
    workers: int = 4, assembler: str = "device", _models: Optional[list] = None, group_size: Optional[int] = None,
) -> List[compat.SolverResults]:
    """Independent factor graphs on one GPU.

    ``assembler``: where the model (gurobi_utils.py:173-187 ``initialize_model``) is built -- ``"device"`` (default): inside
    ``score_create_from_graphs``, on the GPU, from the graphs' flat arrays; ``"native"``: the C++ assembler on the host
    (``score_assemble``), the program handed to ``score_create``; ``"python"``: the NumPy / SciPy twin of it.  Same program,
    same column layout, bit-equal P, q, A, b between the first two.

    ``lockstep=True``: all graphs in ONE handle, advancing through the same kernel launches --
    ADMM warm-up and semismooth-Newton polish alike, with per-problem penalties, step lengths,
    line searches and termination (best device utilisation; the batch takes as many Newton
    iterations as its slowest member).  ``lockstep=False``: one handle (own HIP stream) per graph,
    driven from a pool of ``workers`` host threads.  Default: lock-step groups of up to 16 graphs,
    one group per worker thread, so that host-side setup (``score_create``) and the kernels of
    different groups overlap (measured on 64 four-robot trials: 2490 problems/s in groups of 16
    on 4 threads, 770 problems/s with one handle per graph).  ``group_size`` overrides the group length
    (default: the graphs spread evenly over the workers, at most 16 per group)."""
    check_valid_relaxation(relaxation_type)
    if len(datas) == 0:
        return []
    order = list(range(len(datas)))

    def size_of(i):
        d_ = datas[i]
        if hasattr(d_, "arrays"):  # native.ArrayGraph
            return d_.num_poses + d_.num_ranges
        return sum(len(c) for c in d_.pose_variables) + len(d_.range_measurements)

    if lockstep is None and len(datas) > 1:
        # a lock-step handle needs one block size (2-D and 3-D graphs never share a group); within a
        # dimension, graphs of similar size share a group: it runs as long as its slowest member
        chunks = []
        group = max(1, min(16, -(-len(datas) // max(1, workers)))) if group_size is None else max(1, int(group_size))
        # (worlds of one generated batch -- score_amd.generate -- stay in their order: they are of one shape, and a run of
        #  consecutive worlds is built from the arrays the generator left on the device)
        own = [getattr(d_, "arrays", {}).get("_owner") if hasattr(d_, "arrays") else None for d_ in datas]
        keep_order = own[0] is not None and all(o is own[0] for o in own)
        for dim in sorted({int(datas[i].dimension) for i in order}):
            sub = [i for i in order if int(datas[i].dimension) == dim]
            if not keep_order:
                sub = sorted(sub, key=size_of)
            chunks += [sub[i : i + group] for i in range(0, len(sub), group)]
    elif not lockstep and len(datas) > 1:
        chunks = [[i] for i in order]
    elif len({int(d.dimension) for d in datas}) > 1:
        # lockstep=True was asked for a mixed 2-D / 3-D list: one handle per dimension
        chunks = [[i for i in order if int(datas[i].dimension) == dim] for dim in sorted({int(d.dimension) for d in datas})]
    else:
        chunks = None
    if chunks is not None and (len(chunks) > 1 or len(chunks[0]) != len(datas)):
        first_error = []
        if qcqp_mode not in ("via_socp", "direct"):
            raise ValueError(f"qcqp_mode {qcqp_mode} is not supported")
        def one(idx):
            try:
                # every group builds its own models, in one foreign call for the whole group (_models_for): the group's
                # thread leaves the interpreter lock once for all of them
                models = _models_for([datas[i] for i in idx], relaxation_type, qcqp_mode, lib_path, assembler)
                return solve_score_batch([datas[i] for i in idx], relaxation_type, qcqp_mode, solver_settings, lib_path,
                                         lockstep=True, assembler=assembler, _models=models)
            except ValueError as exc:  # keep what the other graphs of this group produced
                partial = getattr(exc, "partial_results", None)
                if partial is None:
                    if len(idx) > 1:
                        raise
                    partial = [None]
                first_error.append(exc)
                return partial

        if workers <= 1:
            parts = [one(c) for c in chunks]
        else:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=min(workers, len(chunks))) as gpool:
                parts = list(gpool.map(one, chunks))
        out = [None] * len(datas)
        for idx, rs in zip(chunks, parts):
            for i, r in zip(idx, rs):
                out[i] = r
        if first_error:
            bad = [i for i, r in enumerate(out) if r is None]
            err = ValueError(f"graphs {bad} could not be extracted (first error: {first_error[0]})")
            err.partial_results = out  # type: ignore[attr-defined]
            raise err
        return out
    if qcqp_mode not in ("via_socp", "direct"):
        raise ValueError(f"qcqp_mode {qcqp_mode} is not supported")
    models = list(_models) if _models is not None else _models_for(datas, relaxation_type, qcqp_mode, lib_path, assembler)
    settings = dict(DEFAULT_SOLVER_SETTINGS)
    if relaxation_type == QCQP_RELAXATION and qcqp_mode == "direct" and os.environ.get("SCORE_QCQP_PLAIN"):
        # (the plain splitting loop on the program as given; by default the library solves the direct form in its head form,
        #  csrc/score_headform.hpp, with the default settings)
        settings.update(cg_iters=8, adaptive_rho=0)
    settings.update(solver_settings or {})
    _loop_closure_settings(settings, solver_settings, any((d.n_loop_closures if hasattr(d, "arrays") else len(d.loop_closure_measurements)) for d in datas), lib_path)
    if assembler == "device":
        # model construction inside score_create_from_graphs, the estimate straight from the device (score_read_estimates):
        # rounded poses, landmarks, range variables -- no x / y / s copies, no index maps on the host
        from .rounding import finish_device_poses

        solver = ConicSolver.from_graphs([m.graph_arrays for m in models], 0 if models[0].relaxation == SOCP_RELAXATION else 1,
                                         settings, lib_path=lib_path)
        try:
            infos, ests = solver.solve_estimates(qcqp_directions=(relaxation_type != models[0].relaxation))
        finally:
            solver.close()
        out, errors = [], []
        for k, (data, model, info, (T, B, Lm, Rg, flags)) in enumerate(zip(datas, models, infos, ests)):
            solved = info["status"] == 1
            if not solved:
                logger.warning("SCORE solve did not converge: %s", info)
            try:
                T = finish_device_poses(T, B, flags)
            except ValueError as exc:
                if len(datas) == 1:
                    raise
                errors.append((k, exc))
                out.append(None)
                continue
            values = compat.VariableValues(model.dim, compat.ArrayDict(model.pose_names, T), compat.ArrayDict(model.landmark_names, Lm),
                                           compat.ArrayDict(model.range_keys, Rg))
            out.append(compat.SolverResults(
                variables=values, total_time=_runtime_s(info), solved=solved,
                pose_chain_names=model.pose_chain_names if model.pose_chain_names is not None else data.get_pose_chain_names(),
                solver_cost=info.get("pobj"), info=info, relaxed_poses=compat.ArrayDict(model.pose_names, B),
            ))
        if errors:
            k, exc = errors[0]
            err = ValueError(f"{len(errors)} of {len(datas)} graphs could not be extracted (first: #{k}: {exc})")
            err.partial_results = out  # type: ignore[attr-defined]
            raise err
        return out
    solver = ConicSolver([m.qp for m in models], settings, lib_path=lib_path)
    lib, device = solver.lib, int(solver.settings.device)
    try:
        sols = solver.solve()
    finally:
        solver.close()
    out = []
    errors = []
    for k, (data, model, sol) in enumerate(zip(datas, models, sols)):
        if not sol.solved:
            logger.warning("SCORE solve did not converge: %s", sol.info)
        try:
            out.append(extract_solver_results(
                model, sol.x, data, total_time=_runtime_s(sol.info), solved=sol.solved,
                requested_relaxation=relaxation_type, info=sol.info, lib=lib, device=device,
            ))
        except ValueError as exc:
            # an estimate that cannot be rounded (NaN iterate, non-converged rotation block): as in
            # the reference, one graph's failure is that graph's -- the rest of the batch is kept
            if len(datas) == 1:
                raise
            errors.append((k, exc))
            out.append(None)
    if errors:
        k, exc = errors[0]
        err = ValueError(f"{len(errors)} of {len(datas)} graphs could not be extracted (first: #{k}: {exc})")
        err.partial_results = out  # type: ignore[attr-defined]
        raise err
    return out


def solve_problem_with_intermediate_iterates(
    data, relaxation_type: str = QCQP_RELAXATION, every: int = 5, qcqp_mode: str = "via_socp",
    solver_settings: Optional[dict] = None, lib_path: Optional[str] = None, max_snapshots: int = 400,
) -> List[compat.SolverResults]:
    """score/solve_score.py:89-116: one ``SolverResults`` per iteration cap.  The reference restarts
    Gurobi's barrier solver with BarIterLimit = 0, 1, 2, ... until it reports OPTIMAL; here ONE run of
    the product's default solver is paused along its own trajectory: every ``every`` ADMM iterations
    during the warm-up (``polish_warmup`` iterations, 6 by default), then after every semismooth-Newton
    iteration of the polish.  Where the polish does not apply (polish=0) the run continues with ADMM snapshots.  ``solved`` is the solver's own verdict (its three termination tests,
    the counterpart of ``model.status == GRB.OPTIMAL``, gurobi_utils.py:195); the list ends with the
    first solved iterate."""
    check_valid_relaxation(relaxation_type)
    model = _model_for(data, relaxation_type, qcqp_mode, lib_path)
    settings = dict(DEFAULT_SOLVER_SETTINGS)
    n_lc = int(data.n_loop_closures) if hasattr(data, "arrays") else len(data.loop_closure_measurements)  # (ArrayGraph or objects)
    if relaxation_type == QCQP_RELAXATION and qcqp_mode == "direct" and os.environ.get("SCORE_QCQP_PLAIN"):
        settings.update(cg_iters=8, adaptive_rho=0)
    settings.update(solver_settings or {})
    _loop_closure_settings(settings, solver_settings, bool(n_lc), lib_path)
    every = max(1, int(every))
    solver = ConicSolver([model.qp], settings, lib_path=lib_path)
    warmup = int(solver.settings.polish_warmup) if solver.settings.polish else 0
    iterates = []
    try:
        solver.reset()
        t0 = time.time()

        def snapshot(sol) -> bool:
            try:
                iterates.append(extract_solver_results(
                    model, sol.x, data, total_time=time.time() - t0, solved=sol.solved,
                    requested_relaxation=relaxation_type, info=sol.info, lib=solver.lib, device=int(solver.settings.device),
                ))
            except ValueError:
                pass  # an early iterate whose rotation block cannot be rounded yet
            return sol.solved

        # (attempted snapshots are counted, not stored ones: an iterate that cannot be rounded yet still uses up a slot,
        #  so the loops below end whatever extract_solver_results does)
        done, admm, attempts, newton_seen = False, 0, 0, 0
        while not done and admm < warmup and attempts < max_snapshots:
            k = min(every, warmup - admm)
            done = snapshot(solver.steps(k)[0])
            admm += k
            attempts += 1
        newton_possible = warmup > 0
        while not done and attempts < max_snapshots:
            if newton_possible:
                sol = solver.newton_steps(1)[0]
                if sol.info["newton_iters"] == newton_seen:
                    newton_possible = False  # no polish for this program / backend, or Newton has stalled: ADMM goes on
                    continue
                newton_seen = sol.info["newton_iters"]
            else:
                sol = solver.steps(every)[0]
                admm += every
            done = snapshot(sol)
            attempts += 1
            if admm >= settings["max_iters"]:
                break
    finally:
        solver.close()
    return iterates
