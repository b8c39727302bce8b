#!/usr/bin/env python3
"""Headline benchmark: SOCP (ADMM) iterations/s and problems/s of the SCORE
relaxation on the 20-robot / 4-beacon Manhattan RA-SLAM workload
(BASELINE.json configs[3]), with the HBM roofline of the KKT SpMV and a CPU
baseline timed on the same box.

A "step" = one cold-start solve to tolerance of one synthetic factor graph per
GPU (per rank: its own Monte-Carlo trial, weak scaling; no data-path
collective).  Problem data are assembled and resident in HBM before the timed
region.  One JSON line on stdout (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--robots", type=int, default=20)
    ap.add_argument("--poses", type=int, default=1000)
    ap.add_argument("--beacons", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1, help="independent trials per GPU solved in lock-step")
    ap.add_argument("--relaxation", default="SOCP")
    ap.add_argument("--eps", type=float, default=1e-7)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline time budget")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--kkt-reps", type=int, default=2000)
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even for one rank (testing)")
    ap.add_argument("--montecarlo", type=int, default=0, metavar="TRIALS",
                    help="BASELINE config 4: TRIALS independent 4-robot x 1000-pose trials sharded over the ranks; "
                         "reports problems/s of the full solver with every problem resident in HBM")
    ap.add_argument("--mc-threads", type=int, default=4, help="host threads driving handles concurrently (montecarlo)")
    ap.add_argument("--mc-batch", type=int, default=16,
                    help="montecarlo: trials per handle (>1: lock-step batches, one launch serves the whole batch)")
    return ap.parse_args()


def make_workload(args, rank: int):
    from score_amd.assemble import assemble
    from score_amd.manhattan import make_manhattan

    models = []
    for j in range(args.batch):
        trial = rank * args.batch + j
        fg = make_manhattan(n_robots=args.robots, n_poses=args.poses, n_beacons=args.beacons, seed=3000 + trial)
        models.append(assemble(fg, args.relaxation))
    return models


def pmc_traffic(n: int, nnz_p: int):
    """HBM bytes per launch of the KKT SpMV from the committed PMC run
    (profiles/kkt_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    gfx950 FETCH correction applied) -- only when it was taken on this workload."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "kkt_traffic.json")))
        if rec["workload"]["n"] == n and rec["workload"]["nnz_P"] == nnz_p:
            return float(rec["traffic_bytes_per_launch"])
    except Exception:
        pass
    return None


def profiler_kkt_us(n: int, nnz_p: int):
    """Average duration of the KKT SpMV in the committed rocprofv3 kernel trace of this command
    (profiles/kkt_traffic.json carries it) -- only when it was taken on this workload."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "kkt_traffic.json")))
        if rec["workload"]["n"] == n and rec["workload"]["nnz_P"] == nnz_p:
            return float(rec["rocprofv3_kernel_trace"]["average_us"])
    except Exception:
        pass
    return None


def cpu_baseline(args, models):
    """The oracle's CPU twin (same algorithm, OpenMP loops) on a bounded sample of
    the same workload: a fixed number of cold-start ADMM iterations."""
    import __graft_entry__ as g
    from score_amd.solver import ConicSolver

    lib = g.build_oracle()
    import ctypes

    cores = os.cpu_count() or 1
    sol = ConicSolver([m.qp for m in models[:1]], dict(eps_abs=args.eps, eps_rel=args.eps), lib_path=lib)
    # this sparse, memory-bound iteration stops scaling long before a big host runs
    # out of cores: calibrate the OpenMP team size and use the fastest
    try:
        omp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        omp = None
    best = (None, 1e30)
    for nt in sorted({1, 4, 8, 16, 32, 64, cores}):
        if nt > cores or omp is None and nt != cores:
            continue
        if omp is not None:
            omp.omp_set_num_threads(nt)
        sol.reset()
        sol.steps(25)
        t0 = time.perf_counter()
        sol.steps(25)
        dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (nt, dt)
    threads = best[0] or cores
    if omp is not None:
        omp.omp_set_num_threads(threads)
    per25 = max(1e-4, best[1])
    n_it = int(max(25, min(5000, (args.cpu_seconds / per25) * 25)) // 25 * 25)
    sol.reset()
    t0 = time.perf_counter()
    out = sol.steps(n_it)[0]
    dt = time.perf_counter() - t0
    sol.close()
    return {
        "value": n_it / dt, "unit": "iters/s", "cores": int(threads), "host_cores": int(cores), "kind": "port",
        "sample": f"{n_it} cold-start ADMM iterations of trial 0 of the same workload "
                  f"(oracle/cpu_twin, OpenMP, {out.info['cg_iters']} PCG iterations)",
        "seconds": dt,
    }


def montecarlo_summary(eps: float, device: int, trials: int = 64, per_handle: int = 16, threads: int = 4, sweeps: int = 3):
    """BASELINE config 5 on this GPU, as an extra figure of the default single-GPU run: `trials`
    four-robot Monte-Carlo problems in lock-step handles of `per_handle`, full solver."""
    from concurrent.futures import ThreadPoolExecutor

    import torch

    from score_amd.assemble import assemble
    from score_amd.manhattan import make_manhattan
    from score_amd.solver import ConicSolver

    qps = [assemble(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t), "SOCP").qp for t in range(trials)]
    solvers = [ConicSolver(qps[i : i + per_handle], dict(eps_abs=eps, eps_rel=eps, device=device))
               for i in range(0, trials, per_handle)]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        def sweep():
            return [r for rs in pool.map(lambda s: s.solve(), solvers) for r in rs]
        sweep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(sweeps):
            last = sweep()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for s in solvers:
        s.close()
    return {"problems_per_sec": trials * sweeps / dt, "trials": trials, "trials_per_handle": per_handle,
            "host_threads": threads, "solved_last_sweep": int(sum(1 for r in last if r.solved)),
            "workload": "4 robots x 1000 poses, 4 beacons, SOCP, full solver (ADMM warm-up + Newton polish in lock-step)"}


def montecarlo(args, rank, world, local_rank):
    """Config 5: independent Monte-Carlo trials, trial i on rank i % world; a rank's trials are
    grouped into lock-step handles of --mc-batch trials (own stream each); the timed region solves
    all of them `steps` times from a pool of host threads (full solver: ADMM warm-up + Newton
    polish)."""
    from concurrent.futures import ThreadPoolExecutor

    import torch
    import torch.distributed as dist

    from score_amd.assemble import assemble
    from score_amd.manhattan import make_manhattan
    from score_amd.solver import ConicSolver

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    mine = [t for t in range(args.montecarlo) if t % world == rank]
    qps = [assemble(make_manhattan(n_robots=4, n_poses=1000, n_beacons=4, seed=4000 + t), "SOCP").qp for t in mine]
    nb = max(1, args.mc_batch)
    solvers = [ConicSolver(qps[i : i + nb], dict(eps_abs=args.eps, eps_rel=args.eps, device=local_rank))
               for i in range(0, len(qps), nb)]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def sweep(pool):
        return [r for rs in pool.map(lambda s: s.solve(), solvers) for r in rs]

    with ThreadPoolExecutor(max_workers=max(1, args.mc_threads)) as pool:
        for _ in range(args.warmup):
            sweep(pool)
        barrier()
        t0 = time.perf_counter()
        last = None
        for _ in range(args.steps):
            last = sweep(pool)
        barrier()
        dt = time.perf_counter() - t0
    stats = torch.tensor([dt, float(len(mine) * args.steps), float(sum(1 for r in last if r.solved)),
                          float(sum(r.info["iters"] for r in last)), float(sum(r.info["newton_iters"] for r in last))],
                         dtype=torch.float64)
    if use_dist:
        stats = stats.cuda()
        tmax = stats[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone(); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, tot = float(tmax.item()), tot.cpu().tolist()
    else:
        dt_max, tot = dt, stats[1:].tolist()
    if rank == 0:
        print(json.dumps({
            "metric": "problems_per_sec", "value": tot[0] / dt_max, "unit": "problems/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.montecarlo} Monte-Carlo trials, manhattan RA-SLAM 4 robots x 1000 poses, 4 beacons, "
                                   f"SOCP, full solver (ADMM warm-up + Newton polish), {args.mc_threads} host threads/GPU, "
                                   f"{nb} trial(s) per handle" + (" (lock-step)" if nb > 1 else ""),
                       "eps": args.eps, "parallelism": f"trials sharded x{world}"},
            "problems_solved_last_sweep": int(tot[1]), "admm_iters_last_sweep": int(tot[2]),
            "newton_iters_last_sweep": int(tot[3]),
        }), flush=True)
    for s in solvers:
        s.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def chain_factor_doubles(N: int, bs: int, radix: int = 4):
    """Doubles of chain-factor data one preconditioner application reads for a chain of N nodes:
    level 0 run + separator blocks (its spikes are not read by k_prec_pre), every block of the
    coarser levels (layout of score_host.hpp factor_chain_levels)."""
    b2 = bs * bs
    total, level = 0, 0
    while True:
        last = N <= radix - 1
        nsep = 0 if last else N // radix
        nruns = nsep + 1
        P = N if last else radix - 1
        total += 2 * b2 * P * nruns + 2 * b2 * nsep
        if level > 0 and not last:
            total += 2 * b2 * N
        if last:
            return total
        N, level = nsep, level + 1


def algorithmic_bytes(qp, kkt_bytes: float):
    """Algorithmic HBM bytes per launch of the six kernels of one ADMM iteration."""
    n, m, nnzA = int(qp.n), int(qp.m), int(qp.A.nnz)
    bs = int(qp.block_size)
    cp = np.asarray(qp.chain_ptr)
    lens = np.diff(cp)
    fac = 8.0 * sum(chain_factor_doubles(int(L), bs) for L in lens if L > 0)
    n_chain = int(lens.sum()) * bs
    n_jac = n - n_chain
    prec_init = fac + 24.0 * n_chain + 32.0 * n_jac            # r in, z and p out (+ 1/diag for Jacobi columns)
    prec_step = fac + 72.0 * n_chain + 80.0 * n_jac            # + w, p, xt, kx in; r, xt, kx out
    return {
        "rhs": 12.0 * nnzA + 8.0 * (9 * n + m),
        "prec_init": prec_init,
        "kp": float(kkt_bytes),
        "prec_step": prec_step,
        "kpb": float(kkt_bytes) + 24.0 * n,
        "cone": 12.0 * nnzA + 56.0 * m,
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.montecarlo > 0:
        montecarlo(args, rank, world, local_rank)
        return
    workload = (f"manhattan RA-SLAM, {args.robots} robots x {args.poses} poses, {args.beacons} beacons, "
                f"{args.relaxation} relaxation, {args.batch} trial(s)/GPU")
    models = make_workload(args, rank)

    if args.cpu_baseline_only:
        print(json.dumps({"cpu_baseline": cpu_baseline(args, models), "config": {"workload": workload}}))
        return

    import torch
    import torch.distributed as dist

    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from score_amd.solver import ConicSolver

    # Leg A (the timed region of the contract): the operator-splitting loop alone, polish off --
    # this is what "SOCP iterations/s" measures.  Leg B below: the full solver (ADMM warm-up +
    # semismooth-Newton polish), reported as extra fields.
    settings = dict(eps_abs=args.eps, eps_rel=args.eps, device=local_rank, polish=0)
    solver = ConicSolver([m.qp for m in models], settings)  # HIP library; fails loudly without it
    assert solver.backend == "hip-gfx950"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.solve()
    barrier()
    t0 = time.perf_counter()
    iters = 0
    cg = 0
    solved = 0
    last = None
    for _ in range(args.steps):
        last = solver.solve()
        iters += sum(s.info["iters"] for s in last)
        cg += sum(s.info["cg_iters"] for s in last)
        solved += sum(1 for s in last if s.solved)
    barrier()
    dt = time.perf_counter() - t0

    # ---- leg B: full solver (polish on); single problems only ----
    polished = None
    if args.batch == 1:
        ps = ConicSolver([m.qp for m in models], dict(eps_abs=args.eps, eps_rel=args.eps, device=local_rank, polish=1))
        for _ in range(args.warmup):
            ps.solve()
        barrier()
        tp0 = time.perf_counter()
        pl = None
        for _ in range(args.steps):
            pl = ps.solve()
        barrier()
        tp = time.perf_counter() - tp0
        pi = pl[0].info
        polished = {"ms_per_solve": 1e3 * tp / args.steps, "problems_per_sec_this_rank": args.steps / tp,
                    "solved": bool(pl[0].solved), "admm_iters": pi["iters"], "newton_iters": pi["newton_iters"],
                    "newton_pcg_iters": pi["newton_cg_iters"], "pobj": pi["pobj"], "res_pri": pi["res_pri"],
                    "res_dual": pi["res_dual"]}
        ps.close()

    kkt_ms, kkt_bytes = solver.time_kkt_apply(args.kkt_reps)
    alg_bytes = algorithmic_bytes(models[0].qp, kkt_bytes / args.batch)
    # back-to-back duration of every kernel of the iteration (HIP events, solver's stream)
    kernel_us = {k: 1e3 * solver.debug_time(k, 200) for k in
                 ("rhs", "prec_init", "kp", "prec_step", "kpb", "xupdate", "cone")}
    # the same kernels IN THE LOOP of real ADMM iterations (device wall clock, first workgroup in
    # to last workgroup out -- a profiler's kernel duration): caches as the loop leaves them
    inloop_us = solver.time_iteration(warmup=50, iters=200)
    stats = torch.tensor([dt, float(iters), float(args.steps * args.batch), float(solved), float(cg)], dtype=torch.float64)
    if use_dist:
        stats = stats.cuda()
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, tot = float(tmax.item()), tot.cpu().tolist()
    else:
        dt_max, tot = dt, stats[1:].tolist()
    if rank == 0:
        info = last[0].info
        achieved_b2b = kkt_bytes / (kkt_ms * 1e-3) / 1e9
        achieved = kkt_bytes / (inloop_us["kp"] * 1e-6) / 1e9
        rec = {
            "metric": "socp_iters_per_sec", "value": tot[0] / dt_max, "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n": int(models[0].qp.n), "m": int(models[0].qp.m),
                       "nnz_P": int(models[0].qp.P.nnz), "eps": args.eps, "parallelism": f"independent x{world}"},
            "problems_per_sec": tot[1] / dt_max, "problems_solved": int(tot[2]), "problems_total": int(tot[1]),
            "admm_iters_per_solve": tot[0] / max(1.0, tot[1]), "pcg_iters_per_admm_iter": tot[3] / max(1.0, tot[0]),
            "final": {"pobj": info["pobj"], "res_pri": info["res_pri"], "res_dual": info["res_dual"], "rho": info["rho"]},
            "roofline": {"bound": "hbm", "kernel": "k_spmv<KP> (w = K p, KKT operator)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(int(models[0].qp.n), int(models[0].qp.P.nnz)) if args.batch == 1 else None,
                         "bytes_per_launch": kkt_bytes, "us_per_launch": inloop_us["kp"],
                         "timing": "in the ADMM loop: device wall clock, first workgroup in to last workgroup "
                                   "out, averaged over 200 iterations (what rocprofv3 --kernel-trace reports)",
                         "rocprofv3_average_us": profiler_kkt_us(int(models[0].qp.n), int(models[0].qp.P.nnz)) if args.batch == 1 else None,
                         "back_to_back": {"us_per_launch": kkt_ms * 1e3, "achieved": achieved_b2b,
                                          "frac": achieved_b2b / HBM_PEAK_GBS,
                                          "note": "500 consecutive launches of this kernel alone: K stays in the XCD L2s"}},
            "kernel_us_in_loop": inloop_us,
            "kernel_us_back_to_back": kernel_us,
            "full_solver_with_newton_polish": polished,
        }
        # every kernel of the iteration against the same HBM roofline (algorithmic bytes as in
        # DESIGN.md section 4, back-to-back duration), and the whole iteration in the loop
        per_kernel = {}
        for k, b in alg_bytes.items():
            gbs = args.batch * b / (inloop_us[k] * 1e-6) / 1e9
            per_kernel[k] = {"bytes": args.batch * b, "us": inloop_us[k], "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS}
        rec["roofline_by_kernel"] = per_kernel
        it_bytes = args.batch * sum(alg_bytes.values())
        it_us = 1e6 * dt_max / max(1.0, tot[0] / world) * args.batch
        rec["roofline_iteration"] = {"bytes": it_bytes, "us": it_us, "GB/s": it_bytes / (it_us * 1e-6) / 1e9,
                                     "frac": it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "note": "six kernels of one ADMM iteration (2 PCG iterations), time in the launch graph incl. convergence checks"}
        if world == 1 and not args.no_cpu_baseline and args.batch == 1:
            rec["montecarlo_64_trials_this_gpu"] = montecarlo_summary(args.eps, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args, models)
            rec["speedup_vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]
        print(json.dumps(rec), flush=True)
    solver.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
