"""Multi-GPU execution of independent SCORE problems (Monte-Carlo trials).

``solve_score`` has no cross-problem state (score/solve_score.py:54-86), so a
set of factor graphs shards embarrassingly: one process per GPU
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on
CPU), problem i goes to one rank (longest-processing-time assignment by
problem size), every rank solves its share as ONE lock-step batch on its own
device, and a single all_gather of fixed-stride float64 records returns the
estimates.  There is no collective on the data path.  When only one rank holds
the data set (a pickle loaded on rank 0), ``root=`` broadcasts the graphs first:
one object broadcast of names and sizes, one tensor broadcast of every numeric
array, concatenated.
"""
from __future__ import annotations

import inspect
import os
from typing import List, Optional, Sequence

import numpy as np

from . import compat
from .native import ArrayGraph, graph_arrays
from .solve_score import solve_score_batch

_HDR = 10  # status, iters, cg_iters, pobj, res_pri, res_dual, solve_ms, n_values, n_distances, distance width


def shard_assignment(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment; deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += float(costs[i])
    return [sorted(s) for s in shards]


def _meta(data):
    """(dimension, pose names, landmark names, number of ranges) of a FactorGraphData or an ArrayGraph."""
    if isinstance(data, ArrayGraph):
        a = data.arrays
        return int(a["dim"]), list(a["pose_names"]), list(a["landmark_names"]), len(a["range_keys"])
    return (data.dimension, [p.name for chain in data.pose_variables for p in chain],
            [l.name for l in data.landmark_variables], len(data.range_measurements))


def _range_keys(data) -> list:
    """The keys of the distance variables, in measurement order (gurobi_utils.py:288: (first_key, second_key))."""
    if isinstance(data, ArrayGraph):
        return [tuple(k) for k in data.arrays["range_keys"]]
    rm = data.range_measurements
    if rm and hasattr(rm[0], "association"):
        return [tuple(m.association) for m in rm]
    return [(m.first_key, m.second_key) for m in rm]


def problem_cost(data) -> float:
    _, poses, _, n_ranges = _meta(data)
    return float(len(poses) * 6 + 3 * n_ranges)


def _pack(res: compat.SolverResults, data) -> np.ndarray:
    _, names, lm_names, _ = _meta(data)
    vals = [res.poses[n].ravel() for n in names]
    vals += [np.asarray(res.landmarks[l]).ravel() for l in lm_names]
    # the range variables, every one of them as the reference returns them (gurobi_utils.py:127-136: a 1-array per SOCP
    # distance, a d-vector per QCQP direction), in measurement order
    dists = res.variables.distances
    nd, wd = 0, 0
    if dists is not None and len(dists):
        stack = dists.array if hasattr(dists, "array") else np.array([np.asarray(dists[k]).ravel() for k in _range_keys(data)])
        stack = np.asarray(stack, dtype=np.float64).reshape(len(dists), -1)
        nd, wd = stack.shape
        vals.append(stack.ravel())
    v = np.concatenate(vals) if vals else np.zeros(0)
    info = res.info or {}
    hdr = np.array([
        float(info.get("status", 1 if res.solved else 0)), float(info.get("iters", 0)), float(info.get("cg_iters", 0)),
        float(info.get("pobj", np.nan)), float(info.get("res_pri", np.nan)), float(info.get("res_dual", np.nan)),
        float(info.get("solve_ms", res.total_time * 1e3)), float(v.size), float(nd), float(wd),
    ])
    return np.concatenate([hdr, v])


def _unpack(rec: np.ndarray, data) -> compat.SolverResults:
    d, names, lm_names, _ = _meta(data)
    nv = int(rec[7])
    v = rec[_HDR : _HDR + nv]
    k = (d + 1) * (d + 1)
    poses = {n: v[i * k : (i + 1) * k].reshape(d + 1, d + 1).copy() for i, n in enumerate(names)}
    off = len(names) * k
    lms = {l: v[off + i * d : off + (i + 1) * d].copy() for i, l in enumerate(lm_names)}
    off += len(lm_names) * d
    nd, wd = int(rec[8]), int(rec[9])
    dists = compat.ArrayDict(_range_keys(data), v[off : off + nd * wd].reshape(nd, wd).copy()) if nd else {}
    info = dict(status=int(rec[0]), iters=int(rec[1]), cg_iters=int(rec[2]), pobj=float(rec[3]),
                res_pri=float(rec[4]), res_dual=float(rec[5]), solve_ms=float(rec[6]))
    return compat.SolverResults(
        variables=compat.VariableValues(d, poses, lms, dists), total_time=info["solve_ms"] * 1e-3,
        solved=info["status"] == 1, pose_chain_names=data.get_pose_chain_names(), solver_cost=info["pobj"], info=info,
    )


def record_stride(datas: Sequence) -> int:
    worst = 0
    for data in datas:
        d, names, lm_names, n_ranges = _meta(data)
        # (room for d values per range: the QCQP directions; the SOCP distances use one)
        worst = max(worst, len(names) * (d + 1) ** 2 + len(lm_names) * d + n_ranges * d)
    return _HDR + worst


# numeric fields of ``native.graph_arrays`` (everything else in that dict is names / small integers)
_NUMERIC = ("chain_len", "rel_base", "rel_to", "rel_t", "rel_R", "rel_kappa", "rel_tau", "rng_a", "rng_b", "rng_dist",
            "rng_prec", "lprior_lm", "lprior_t", "lprior_prec")


def broadcast_graphs(datas: Optional[Sequence], root: int = 0, device: Optional[int] = None) -> List[ArrayGraph]:
    """Rank ``root`` holds the factor graphs (FactorGraphData or ArrayGraph), the other ranks pass ``None``; every rank
    returns the list as ``ArrayGraph``s.  Two collectives: ONE object broadcast (names, range keys, shapes and dtypes
    of the numeric arrays) and ONE broadcast of all numeric arrays concatenated into a float64 buffer (the int32 index
    arrays travel as exact float64 values; backend "nccl" = RCCL: through this rank's GPU).  Without a process
    group: the arrays of ``datas``."""
    import torch
    import torch.distributed as dist

    def arrays_of(g):
        return g.arrays if isinstance(g, ArrayGraph) else graph_arrays(g)

    if not (dist.is_available() and dist.is_initialized()):
        return [ArrayGraph(arrays_of(g)) for g in (datas or [])]
    rank = dist.get_rank()
    nccl = dist.get_backend() == "nccl"
    if nccl and device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    header, flat, error = None, None, None
    if rank == root:
        # Whatever goes wrong while the root reads its graphs (no graphs at all, duplicate or unknown variable names, a
        # graph without poses: graph_arrays raises the reference's errors), the root must still reach the collective --
        # the other ranks are waiting in it.  The error travels as a string and is raised on EVERY rank afterwards.
        try:
            if datas is None:
                raise ValueError("broadcast_graphs: the root rank must hold the graphs")
            header, chunks = [], []
            for g in datas:
                a = arrays_of(g)
                # (private keys -- a generated world's '_owner' holds a CDLL and the generator's handle -- stay here; the lazy
                #  name tables of generated worlds travel as plain lists: the receivers get ordinary array graphs)
                meta = {k: v for k, v in a.items() if k not in _NUMERIC and not k.startswith("_")}
                meta["pose_names"] = list(meta["pose_names"])
                meta["range_keys"] = [tuple(k) for k in meta["range_keys"]]
                meta["landmark_names"] = list(meta["landmark_names"])
                meta["pose_chain_names"] = [list(c) for c in meta["pose_chain_names"]]
                meta["_shapes"] = {k: (tuple(np.shape(a[k])), np.asarray(a[k]).dtype.str) for k in _NUMERIC}
                header.append(meta)
                chunks += [np.asarray(a[k], dtype=np.float64).ravel() for k in _NUMERIC]
            flat = np.concatenate(chunks) if chunks else np.zeros(0)
        except Exception as exc:  # noqa: BLE001 - re-raised on every rank after the collective
            error = f"{type(exc).__name__}: {exc}"
            header, flat = None, np.zeros(0)
    box = [header, int(flat.size) if flat is not None else 0, error]
    # (backend "nccl" moves the pickled payload through a GPU: this rank's, not whatever torch.cuda.current_device() is)
    dist.broadcast_object_list(box, src=root, device=torch.device(f"cuda:{device}") if nccl else None)
    header, total, error = box
    if error is not None:
        raise ValueError(f"broadcast_graphs: rank {root} could not read its graphs ({error})")
    t = torch.from_numpy(flat) if rank == root else torch.empty(total, dtype=torch.float64)
    if nccl:
        t = t.to(f"cuda:{device}")
    if total:
        dist.broadcast(t, src=root)
    flat = t.cpu().numpy()
    out, o = [], 0
    for meta in header:
        a = {k: v for k, v in meta.items() if k != "_shapes"}
        for k in _NUMERIC:
            shape, dt = meta["_shapes"][k]
            n = int(np.prod(shape)) if len(shape) else 1
            a[k] = flat[o : o + n].astype(np.dtype(dt)).reshape(shape)
            o += n
        out.append(ArrayGraph(a))
    return out


def all_gather_records(buf: np.ndarray, device: Optional[int] = None) -> List[np.ndarray]:
    """One all_gather of a rank's fixed-shape float64 record block; returns every rank's block (rank order).
    Backend "nccl" (= RCCL over xGMI on ROCm): the block travels through this rank's GPU; "gloo": host memory.
    Without an initialised process group: [buf] -- the single-process case takes the same code path."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [np.asarray(buf)]
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
    if dist.get_backend() == "nccl":
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        t = t.to(f"cuda:{device}")
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    return [g.cpu().numpy() for g in gathered]


def solve_score_sharded(
    datas: Optional[Sequence], relaxation_type: str = "QCQP", solver_settings: Optional[dict] = None,
    lib_path: Optional[str] = None, device: Optional[int] = None, root: Optional[int] = None,
) -> List[compat.SolverResults]:
    """Every rank passes the SAME list of factor graphs and gets ALL results -- or, with ``root=r``, only rank r holds
    the graphs (a data set loaded from disk, e.g. the reference's examples/goats_14_data pickle), the others pass
    ``None``, and the graphs are broadcast first (``broadcast_graphs``: two collectives).

    (Monte-Carlo graphs are generated from seeds, so holding the whole list on every rank costs
    nothing and spares a broadcast of problem data; a rank only ASSEMBLES and solves its own
    shard.  The gathered record carries the rounded poses, landmarks and solver statistics --
    what ``SolverResults`` exposes -- not the raw conic iterates x / y / s.)  If any problem fails
    anywhere, every rank raises after the collective; no rank is left waiting."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return solve_score_batch(datas, relaxation_type, solver_settings=solver_settings, lib_path=lib_path)
    world, rank = dist.get_world_size(), dist.get_rank()
    if root is not None:
        datas = broadcast_graphs(datas if rank == root else None, root, device)
    shards = shard_assignment([problem_cost(d) for d in datas], world)
    mine = shards[rank]
    settings = dict(solver_settings or {})
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    settings.setdefault("device", device)
    stride = record_stride(datas)
    per_rank = max(len(s) for s in shards)
    buf = np.zeros((per_rank, stride))
    # Whatever happens on this rank (score_create / score_solve failing, a graph whose estimate
    # cannot be packed), it must still reach the all_gather below -- the other ranks are waiting in
    # it.  A failed slot travels as a failure record (status 3 = numerical, NaN values); the error
    # is raised on every rank AFTER the collective.
    failure = None
    try:
        results = solve_score_batch([datas[i] for i in mine], relaxation_type, solver_settings=settings,
                                    lib_path=lib_path) if mine else []
    except Exception as exc:  # noqa: BLE001 - re-raised after the collective
        failure = f"rank {rank}: {type(exc).__name__}: {exc}"
        results = [None] * len(mine)
    for slot, (i, res) in enumerate(zip(mine, results)):
        try:
            if res is None:
                raise RuntimeError("no result")
            rec = _pack(res, datas[i])
            buf[slot, : rec.size] = rec
        except Exception as exc:  # noqa: BLE001
            if failure is None:
                failure = f"rank {rank}, problem {i}: {type(exc).__name__}: {exc}"
            buf[slot, :] = np.nan
            buf[slot, 0] = 3.0   # status: numerical
            buf[slot, 7] = -1.0  # no values
    gathered = all_gather_records(buf, device)
    out: List[Optional[compat.SolverResults]] = [None] * len(datas)
    failed = []
    for r in range(world):
        g = gathered[r]
        for slot, i in enumerate(shards[r]):
            if g[slot, 7] < 0:
                failed.append(i)
                continue
            out[i] = _unpack(g[slot], datas[i])
    if failed or failure:
        raise RuntimeError(f"solve_score_sharded: problems {failed} failed" + (f" ({failure})" if failure else ""))
    return out  # type: ignore[return-value]


def solve_generated_sharded(
    total: int, seed: int = 0, relaxation_type: str = "QCQP", solver_settings: Optional[dict] = None,
    lib_path: Optional[str] = None, device: Optional[int] = None, **spec,
) -> List[compat.SolverResults]:
    """A Monte-Carlo study from seeds over the ranks: ``total`` synthetic Manhattan worlds (``score_amd.generate``: world t is
    the world of ``seed + t``), a block of consecutive worlds per rank -- DRAWN THERE, on that rank's device, so that no problem
    data is distributed at all --, solved as lock-step batches, and one all_gather of fixed-stride records returns every estimate
    to every rank (two small collectives in front of it agree on the record stride).  A record carries what ``_pack`` carries
    plus the endpoints of the world's range measurements: a foreign world's distance keys are rebuilt from them.  Without an
    initialised process group: the single-process run of the same code."""
    import torch
    import torch.distributed as dist

    from .generate import GeneratedBatch, _PoseNames, _RangeKeys
    from .manhattan import robot_letters

    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    per = -(-int(total) // world)
    lo, hi = min(total, rank * per), min(total, (rank + 1) * per)
    settings = dict(solver_settings or {})
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    settings.setdefault("device", device)
    failure = None
    graphs, results = [], []
    try:
        if hi > lo:
            batch = GeneratedBatch(hi - lo, seed=int(seed) + lo, device=int(settings["device"]), lib_path=lib_path, **spec)
            graphs = batch.graphs()
            results = solve_score_batch(graphs, relaxation_type, solver_settings=settings, lib_path=lib_path)
    except Exception as exc:  # noqa: BLE001 - re-raised after the collectives
        failure = f"rank {rank}: {type(exc).__name__}: {exc}"
        results = [None] * len(graphs)
    # the record stride: the largest world anywhere (ranges differ from world to world)
    recs = []
    for g, res in zip(graphs, results):
        try:
            if res is None:
                raise RuntimeError("no result")
            a = g.arrays
            recs.append(np.concatenate([_pack(res, g), a["rng_a"].astype(np.float64), a["rng_b"].astype(np.float64)]))
        except Exception as exc:  # noqa: BLE001
            if failure is None:
                failure = f"rank {rank}: {type(exc).__name__}: {exc}"
            recs.append(None)
    stride = max([r.size for r in recs if r is not None] + [_HDR])
    if on:
        t = torch.tensor([float(stride)], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.to(f"cuda:{int(settings['device'])}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        stride = int(t.item())
    buf = np.zeros((per, stride))
    buf[:, 7] = -2.0  # empty slot
    for slot, r in enumerate(recs):
        if r is None:
            buf[slot, :] = np.nan
            buf[slot, 0], buf[slot, 7] = 3.0, -1.0
        else:
            buf[slot, : r.size] = r
    gathered = all_gather_records(buf, int(settings["device"]))
    # (the shape of a world: the generator's own defaults, from its signature -- nothing restated here)
    gen_defaults = {k: v.default for k, v in inspect.signature(GeneratedBatch.__init__).parameters.items() if v.default is not inspect.Parameter.empty}
    R, T, Nb, d = (int(spec.get(k, gen_defaults[k])) for k in ("n_robots", "n_poses", "n_beacons", "dim"))
    poses = _PoseNames(robot_letters(R), T)
    lms = [f"L{k}" for k in range(Nb)]
    chains = [_PoseNames([ch], T) for ch in robot_letters(R)]
    out: List[Optional[compat.SolverResults]] = [None] * int(total)
    failed = []
    for r in range(world):
        for slot in range(per):
            i = r * per + slot
            if i >= total:
                break
            rec = gathered[r][slot]
            if rec[7] < 0:
                failed.append(i)
                continue
            nv, nd, wd = int(rec[7]), int(rec[8]), int(rec[9])
            k = (d + 1) * (d + 1)
            if nv != len(poses) * k + Nb * d + nd * wd:
                raise RuntimeError(f"solve_generated_sharded: world {i}: record of {nv} values does not match {R} robots x {T} poses, {Nb} beacons, {nd} ranges")
            v = rec[_HDR : _HDR + nv]
            P = v[: len(poses) * k].reshape(len(poses), d + 1, d + 1).copy()
            off = len(poses) * k
            L = v[off : off + Nb * d].reshape(Nb, d).copy()
            off += Nb * d
            D = v[off : off + nd * wd].reshape(nd, wd).copy()
            ends = rec[_HDR + nv : _HDR + nv + 2 * nd]
            keys = _RangeKeys(ends[:nd].astype(np.int32), ends[nd:].astype(np.int32), poses)
            info = dict(status=int(rec[0]), iters=int(rec[1]), cg_iters=int(rec[2]), pobj=float(rec[3]), res_pri=float(rec[4]),
                        res_dual=float(rec[5]), solve_ms=float(rec[6]), seed=int(seed) + i, rank=r)
            out[i] = compat.SolverResults(
                variables=compat.VariableValues(d, compat.ArrayDict(poses, P), compat.ArrayDict(lms, L), compat.ArrayDict(keys, D) if nd else {}),
                total_time=info["solve_ms"] * 1e-3, solved=info["status"] == 1, pose_chain_names=chains, solver_cost=info["pobj"], info=info,
            )
    if failed or failure:
        raise RuntimeError(f"solve_generated_sharded: worlds {failed} failed" + (f" ({failure})" if failure else ""))
    return out  # type: ignore[return-value]
