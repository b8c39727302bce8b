#!/usr/bin/env python3
"""Benchmarks of the MI355X SCORE solver (BASELINE.json's metric: SOCP iterations/s and
problems/s on synthetic multi-robot Manhattan RA-SLAM graphs, with the HBM roofline of the KKT
SpMV and a CPU baseline timed on the same box).

    python bench.py [--gpus N] [--steps K] [--warmup W]

The primary metric is the same at every N: SOCP iterations/s on BASELINE.json configs[3] (one 20-robot x
1000-pose x 4-beacon SOCP per GPU, resident in HBM), weak scaling -- every rank solves its own headline
problem, a step = one cold-start solve to eps = 1e-7 with the operator-splitting loop alone, `value` = ADMM
iterations of all ranks / the slowest rank's time.  N = 1 is the line the driver records as BENCH.
Every N also carries BASELINE configs[4] as `config5_montecarlo`: 64 independent 4-robot x 1000-pose
trials, trial t on rank t mod N (strong scaling), the product's default solver, and after every sweep ONE
all_gather of the result records (RCCL over xGMI; score_amd.distributed.all_gather_records) -- problems/s of
the whole job.
N = 1 additionally carries: the product's default solver (ADMM warm-up + semismooth-Newton polish) on the
headline problem, the in-loop roofline of the KKT SpMV (single problem and a lock-step batch of 16), the
roofline of the kernel that dominates GPU time (the chain preconditioner), the end-to-end legs and the CPU
baselines.  (`--workload montecarlo`: configs[4] alone as the primary metric -- profiling runs only.)

Launching: `python bench.py --gpus N` starts N ranks itself (one child process per GPU, spawned
BEFORE anything in this process touches the GPU; RCCL rendezvous on 127.0.0.1); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (RANK / WORLD_SIZE already in
the environment) it is one of the ranks.  Rank 0 prints ONE JSON line on stdout: the compact contract record
(<= 4 KB: contract keys, config, roofline of the KKT SpMV + the dominant kernel + the batch of 16, cpu_baseline, a
few scalars per leg).  The FULL record (every leg, per-call arrays, notes) goes to --full-out
(gpurun_out/bench_full.json) and, as one line, to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs import (  # noqa: E402  (the legs: workloads, Monte-Carlo, CPU baselines, roofline helpers -- bench_legs.py)
    HBM_PEAK_GBS, MC_SEED0, Dist, MonteCarlo, algorithmic_bytes, config5_leg, cpu_baseline, economy_leg, end_to_end, long_chain_leg, make_headline, mc_groups,
    mc_models, newton_probe, pmc_traffic, roofline_block, roofline_newton, run_montecarlo_leg, solver_lib, survey_prec_bytes, three_d_leg,
)

def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("auto", "headline", "montecarlo"), default="auto",
                    help="auto: headline on one GPU, montecarlo (BASELINE configs[4]) on several")
    ap.add_argument("--robots", type=int, default=20)
    ap.add_argument("--poses", type=int, default=1000)
    ap.add_argument("--beacons", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1, help="headline: independent trials per GPU solved in lock-step")
    ap.add_argument("--relaxation", default="SOCP")
    ap.add_argument("--eps", type=float, default=1e-7)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of EACH CPU-baseline entry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end leg (profiler runs: eight host threads creating handles at once)")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--no-probes", action="store_true",
                    help="timed region only (loop-only profiler traces): no roofline probes, no extra legs")
    ap.add_argument("--kkt-reps", type=int, default=1000)
    ap.add_argument("--roofline-batch", type=int, default=16, help="problems in the lock-step handle of roofline_batch16")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for one rank (testing)")
    ap.add_argument("--montecarlo", type=int, default=0, metavar="TRIALS",
                    help="run BASELINE configs[4] with TRIALS trials (default 64 when the workload is montecarlo)")
    ap.add_argument("--mc-threads", type=int, default=4, help="host threads driving handles concurrently (montecarlo)")
    ap.add_argument("--mc-batch", type=int, default=16, help="montecarlo: at most this many trials per lock-step handle")
    ap.add_argument("--fresh-batch", type=int, default=16, help="fresh-graph leg: at most this many graphs per lock-step handle")
    ap.add_argument("--fresh-threads", type=int, default=4, help="fresh-graph leg: host threads creating / solving handles concurrently")
    ap.add_argument("--mc-robots", type=int, default=4)
    ap.add_argument("--mc-poses", type=int, default=1000)
    ap.add_argument("--mc-beacons", type=int, default=4)
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="where rank 0 writes the FULL record (every leg, every note); stdout carries the compact contract line only")
    ap.add_argument("--test-cpu-twin", action="store_true",
                    help="TEST ONLY (tests/test_bench_contract.py): gloo + the oracle's CPU twin instead of RCCL + "
                         "the HIP library, to exercise the N-rank launcher without GPUs; its numbers are not measurements")
    args = ap.parse_args(argv)
    if args.test_cpu_twin and os.environ.get("SCORE_BENCH_TEST_MODE") != "1":
        ap.error("--test-cpu-twin runs the oracle's CPU twin in place of the HIP library: it exists for the launcher tests "
                 "only and needs SCORE_BENCH_TEST_MODE=1 in the environment (nothing it prints is a measurement)")
    return args


# ---------------------------------------------------------------------------------------------
# the contract line (round 6): the LAST stdout line is a compact record (<= 4 KB: the driver reads a bounded tail of
# stdout -- round 5's 20 KB line was not parsed); everything else goes to a side file and to stderr
# ---------------------------------------------------------------------------------------------
CONTRACT_MAX_BYTES = 4096


def _r(x, sig=5):
    """floats to `sig` significant digits (ints, bools, None, strings unchanged)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _pick(d, *keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d}


def _roof3(block):
    """a roofline block as the contract wants it: numbers only"""
    if not block:
        return None
    out = _pick(block, "bound", "kernel", "bytes_per_launch", "us_per_launch", "achieved", "peak", "unit", "frac", "traffic")
    if "kernel" in out:
        out["kernel"] = out["kernel"].split(" (")[0].split(",")[0][:48]
    return out


def compact_record(rec):
    """The contract line: metric / value / ... / config / roofline / cpu_baseline + a handful of scalars of the other
    legs.  No prose, no per-call arrays (those are in the full record: --full-out)."""
    out = {k: _r(rec.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                         "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(rec.get("config") or {})
    cfg["workload"] = str(cfg.get("workload", ""))[:200]
    out["config"] = {k: _r(v) for k, v in cfg.items()}
    out["roofline"] = _roof3(rec.get("roofline"))
    for k in ("roofline_dominant", "roofline_batch16"):
        if rec.get(k):
            out[k] = _roof3(rec[k])
    cb = rec.get("cpu_baseline")
    out["cpu_baseline"] = None if not cb else dict(_pick(cb, "value", "unit", "cores", "kind", "seconds"), sample=str(cb.get("sample", ""))[:160])
    legs = {}
    tr = rec.get("timed_region") or {}
    legs.update({"admm_iters_per_solve": _r(tr.get("admm_iters_per_solve")), "problems_solved": tr.get("problems_solved"),
                 "problems_total": tr.get("problems_total")} if tr else {})
    pd = rec.get("product_default_solver") or {}
    if pd:
        legs.update({"default_solve_ms": _r(pd.get("ms_per_solve")), "default_solve_solved": pd.get("solved"),
                     "default_newton_pcg_iters": pd.get("newton_pcg_iters")})
    c5 = rec.get("config5_montecarlo") or {}
    if c5:
        legs.update({"config5_resolve_problems_per_sec": _r(c5.get("problems_per_sec")),
                     "config5_fresh_graphs_problems_per_sec": _r(c5.get("fresh_graphs_problems_per_sec")),
                     "config5_generated_graphs_problems_per_sec": _r(c5.get("generated_graphs_problems_per_sec")),
                     "config5_solved_last_sweep": c5.get("solved_last_sweep"), "config5_trials": c5.get("trials")})
        fr = c5.get("fresh_graphs") or {}
        if fr:
            legs["host_cpu_ms_per_fresh_problem"] = _r(fr.get("host_cpu_ms_per_problem_mean_over_ranks"))
            if "host_cpu_ms_work_wait" in fr:
                legs["host_cpu_ms_work_wait"] = [_r(v) for v in fr["host_cpu_ms_work_wait"]]
        ec = c5.get("fresh_graphs_economy_waits") or {}
        if "problems_per_sec" in ec:
            legs["economy_waits_fresh_problems_per_sec"] = _r(ec["problems_per_sec"])
            legs["economy_waits_host_cpu_ms_per_problem"] = _r(ec["host_cpu_ms_per_problem"])
    e2e = rec.get("end_to_end") or {}
    if e2e:
        legs.update({"solve_score_ms": _r(e2e.get("headline_solve_score_ms")), "solve_score_repeat_ms": _r(e2e.get("headline_solve_score_repeat_ms")),
                     "score_create_ms": _r(e2e.get("score_create_ms_best")),
                     "e2e_from_objects_problems_per_sec": _r(e2e.get("config4_end_to_end_problems_per_sec")),
                     "e2e_from_objects_first_pass_problems_per_sec": _r(e2e.get("config4_end_to_end_first_pass_problems_per_sec")),
                     "e2e_from_arrays_problems_per_sec": _r(e2e.get("config4_end_to_end_from_arrays_problems_per_sec"))})
    for k in ("problems_total_per_step", "problems_solved_last_sweep"):  # (--workload montecarlo)
        if k in rec:
            legs[k] = rec[k]
    for k, key in (("three_d", "three_d_default_ms"), ("long_chains", "long_chains_default_ms")):
        if rec.get(k):
            legs[key] = _r(rec[k].get("product_default_ms"))
    if rec.get("speedup_vs_cpu_baseline") is not None:
        legs["speedup_vs_cpu_baseline"] = _r(rec["speedup_vs_cpu_baseline"])
    sp = rec.get("speedup_time_to_solution") or {}
    if sp.get("C2_twin_best_team") is not None:
        legs["speedup_time_to_solution_vs_best_cpu_team"] = _r(sp["C2_twin_best_team"])
    out["legs"] = {k: v for k, v in legs.items() if v is not None}
    for k in ("test_mode", "full_record"):
        if k in rec:
            out[k] = rec[k]
    return out


def emit(args, rec):
    """Rank 0's output: the full record to --full-out (and one stderr line), the compact contract record as the ONE stdout line."""
    path = args.full_out
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(rec, fh)
        rec["full_record"] = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as e:  # a read-only tree must not cost the contract line
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    print(json.dumps(rec), file=sys.stderr, flush=True)
    line = json.dumps(compact_record(rec), separators=(",", ":"))
    if len(line) > CONTRACT_MAX_BYTES:  # never again: drop the optional legs before the contract fields
        c = compact_record(rec)
        c.pop("legs", None)
        line = json.dumps(c, separators=(",", ":"))
    assert len(line) <= CONTRACT_MAX_BYTES, len(line)
    sys.stderr.flush()
    print(line, flush=True)


# ---------------------------------------------------------------------------------------------
# N-rank launcher
# ---------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """Start one child process per GPU (this process has not touched the GPU and never will),
    relay rank 0's stdout, return the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's stdout is collected by a reader thread; every child is polled, and the first one to fail takes
    # the others (exactly the processes this launcher started) down at once instead of leaving them in a
    # collective until the backend's own timeout
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = max(rc, abs(code))
                sys.stderr.write(f"[bench] rank {r} exited with code {code}; stopping the other ranks\n")
                for o in sorted(live):
                    procs[o].terminate()
        if live:
            time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write("".join(x or "" for x in out0))
    sys.stdout.flush()
    return rc


def run_headline(args, D: Dist):
    from score_amd.solver import ConicSolver

    workload = (f"BASELINE configs[3]: manhattan RA-SLAM, {args.robots} robots x {args.poses} poses, {args.beacons} beacons, "
                f"{args.relaxation} relaxation, {args.batch} trial(s)/GPU, ADMM loop alone (polish off) to eps {args.eps:g}")
    # the end-to-end leg's graph is built FIRST, as an application that loads its graph and calls solve_score()
    # would hold it: 47 k measurement objects allocated late in this process sit scattered over a fragmented
    # object heap and their attribute reads take twice as long
    e2e_graph = None
    if D.world == 1 and args.batch == 1 and not args.no_probes and not args.cpu_baseline_only:
        from score_amd.manhattan import make_manhattan

        e2e_graph = make_manhattan(n_robots=args.robots, n_poses=args.poses, n_beacons=args.beacons, seed=3000)
    models = make_headline(args, D.rank, args.batch)
    if args.cpu_baseline_only:
        emit(args, {"cpu_baseline": cpu_baseline(args, models), "config": {"workload": workload}})
        return
    dev = D.device
    base = dict(eps_abs=args.eps, eps_rel=args.eps, device=dev)
    # The Monte-Carlo legs run FIRST (round 5).  They share nothing with the legs below, but the order matters to the HIP
    # runtime: a process that has destroyed a launch-graph executable -- the ADMM-only legs below replay graphs of 25 iterations
    # and rebuild them when the PCG count adapts -- runs every later lock-step batch ~15 % slower (64 config-5 trials 4130-4220
    # -> 3520-3580 problems/s after ONE graph-using solve of any handle; profiles/r05_graph_after_effect.txt).  The product
    # default path itself no longer builds graphs (its one ADMM block is 6 iterations: launched directly).
    early = {}
    if not args.no_probes and (D.world > 1 or args.test_cpu_twin or args.batch == 1):
        early["config5_montecarlo"] = config5_leg(args, D, *([args.montecarlo or 64] if (D.world > 1 or args.test_cpu_twin) else []))
        if D.world == 1 and not args.test_cpu_twin and not args.no_e2e:
            early["end_to_end"] = end_to_end(args, dev, e2e_graph)
    # Leg A (the timed region of the contract): the operator-splitting loop alone -- what "SOCP
    # iterations/s" measures.  Leg B: the product default (ADMM warm-up + Newton polish).
    solver = ConicSolver([m.qp for m in models], dict(base, polish=0), lib_path=solver_lib(args))  # HIP library; fails loudly without it
    assert solver.backend == ("cpu-twin" if args.test_cpu_twin else "hip-gfx950")
    for _ in range(args.warmup):
        solver.solve()
    D.barrier()
    t0 = time.perf_counter()
    iters = cg = solved = 0
    last = None
    for _ in range(args.steps):
        last = solver.solve()
        iters += sum(s.info["iters"] for s in last)
        cg += sum(s.info["cg_iters"] for s in last)
        solved += sum(1 for s in last if s.solved)
    D.barrier()
    dt = time.perf_counter() - t0
    dt_max, tot = D.reduce(dt, [float(iters), float(args.steps * args.batch), float(solved), float(cg)])
    rec = None
    if D.rank == 0:
        info = last[0].info
        rec = {
            "metric": "socp_iters_per_sec", "value": tot[0] / dt_max, "unit": "iters/s",
            "n_gpus": D.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n": int(models[0].qp.n), "m": int(models[0].qp.m),
                       "nnz_P": int(models[0].qp.P.nnz), "eps": args.eps, "parallelism": f"independent x{D.world}"},
            "timed_region": {"what": "cold-start solves with the ADMM loop alone (settings.polish = 0); NOT the product default",
                             "problems_per_sec_admm_only": tot[1] / dt_max, "problems_solved": int(tot[2]),
                             "problems_total": int(tot[1]), "admm_iters_per_solve": tot[0] / max(1.0, tot[1]),
                             "pcg_iters_per_admm_iter": tot[3] / max(1.0, tot[0]),
                             "final": {"pobj": info["pobj"], "res_pri": info["res_pri"], "res_dual": info["res_dual"], "rho": info["rho"]}},
        }
    if args.no_probes:
        if rec is not None:
            rec["roofline"] = None
            rec["cpu_baseline"] = None
            emit(args, rec)
        solver.close()
        D.close()
        return
    if D.world > 1 or args.test_cpu_twin:
        # several ranks: the same primary metric (every rank its own headline problem), BASELINE configs[4] sharded over
        # the ranks with its result gather; roofline and CPU baselines are measured by the single-GPU line
        solver.close()
        c5 = early["config5_montecarlo"]
        if rec is not None:
            rec["config5_montecarlo"] = c5
            rec["roofline"] = None
            rec["cpu_baseline"] = None
            rec["note"] = "roofline and cpu_baseline are measured by the single-GPU line (python bench.py)"
            if args.test_cpu_twin:
                rec["test_mode"] = "gloo + oracle CPU twin: launcher test, NOT a measurement"
            emit(args, rec)
        D.close()
        return

    # ---- leg B: product default (polish on); single problems only ----
    polished = None
    if args.batch == 1:
        ps = ConicSolver([m.qp for m in models], dict(base, polish=1))
        for _ in range(args.warmup):
            ps.solve()
        D.barrier()
        tp0 = time.perf_counter()
        pl = None
        for _ in range(args.steps):
            pl = ps.solve()
        D.barrier()
        tp = time.perf_counter() - tp0
        pi = pl[0].info
        polished = {"what": "product default: a short ADMM warm-up (polish_warmup iterations) + semismooth-Newton polish, same problem, same eps",
                    "ms_per_solve": 1e3 * tp / args.steps, "problems_per_sec": args.steps / tp,
                    "solved": bool(pl[0].solved), "admm_iters": pi["iters"], "newton_iters": pi["newton_iters"],
                    "newton_pcg_iters": pi["newton_cg_iters"], "pobj": pi["pobj"], "res_pri": pi["res_pri"],
                    "res_dual": pi["res_dual"]}
        try:
            _, polished["newton_probe"] = newton_probe(ps)
        except KeyError:
            pass
        ps.close()

    # ---- roofline probes, all on the solver's stream ----
    kkt_ms, kkt_bytes = solver.time_kkt_apply(args.kkt_reps)
    repinfo = solver.debug_get("rep")
    rep, nnz_at = int(repinfo[0]), int(repinfo[2]) // args.batch
    alg = algorithmic_bytes(models[0].qp, kkt_bytes / args.batch, rep, nnz_at)
    b2b_us = {k: 1e3 * solver.debug_time(k, 200) for k in ("rhs", "prec_init", "kp", "prec_step", "kpb", "xupdate", "cone")}
    dev_us, disp_us = solver.time_iteration(warmup=50, iters=200, dispatch=True)
    if D.rank == 0:
        rec["product_default_solver"] = polished
        rec["config"]["row_replication"] = rep  # K = I_rep (x) K_row (+ tail): K_row and one factor set per robot streamed once
        rec["config"]["fac_fp32"] = int(solver.settings.fac_fp32)  # chain factors streamed as 4-byte values (arithmetic stays f64)
        rec["config"]["cg_iters"] = int(solver.settings.cg_iters)  # PCG iterations per ADMM iteration (adaptive from there)
        headline_shape = (args.robots, args.poses, args.beacons, args.batch, args.relaxation) == (20, 1000, 4, 1, "SOCP")
        rec["roofline"] = roofline_block(
            "k_spmv<KP> (w = K p, the KKT operator), single problem", kkt_bytes, disp_us["kp"], dev_us["kp"],
            {"back_to_back": {"us_per_launch": kkt_ms * 1e3, "frac": kkt_bytes / (kkt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": f"{args.kkt_reps} consecutive launches of this kernel alone (HIP events): K stays in the XCD L2s"},
             "traffic": pmc_traffic("kp", "headline_loop" if headline_shape else None)})
        # the kernel that owns the iteration's GPU time: the chain preconditioner inside the PCG step
        rec["roofline_dominant"] = roofline_block(
            "k_prec_pre<STEP> (PCG step + z = M^-1 r, block-tridiagonal chains by nested dissection, coarse-level factors register-resident)",
            args.batch * alg["prec_step"], disp_us["prec_step"], dev_us["prec_step"],
            {"traffic": pmc_traffic("prec_step", "headline_loop" if headline_shape else None),
             "bytes_survey_8d": args.batch * (survey_prec_bytes(models[0].qp) + 56.0 * int(models[0].qp.n)),
             "frac_survey_8d": args.batch * (survey_prec_bytes(models[0].qp) + 56.0 * int(models[0].qp.n)) / (disp_us["prec_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
             "bytes_note": "bytes_per_launch = what this kernel reads and writes: the radix-4 nested-dissection factors as "
                           "4-byte values (fac_fp32) + the fused vector update in fp64; bytes_survey_8d = SURVEY 8(d)'s B_prec "
                           "(a plain fp64 block-tridiagonal factor, blocks shared by the d rows) + the same vector update. "
                           "One factor set per robot serves its d chains (row replication); 40 workgroups on 256 CUs: the kernel is "
                           "bound by its dependent phases and launch shell, not by bytes"})
        per_kernel = {}
        for k, b in alg.items():
            gbs = args.batch * b / (disp_us[k] * 1e-6) / 1e9
            per_kernel[k] = {"bytes": args.batch * b, "us_dispatch": disp_us[k], "us_device": dev_us[k], "us_back_to_back": b2b_us.get(k),
                             "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS}
        rec["roofline_by_kernel"] = per_kernel
        it_bytes = args.batch * sum(alg.values())
        it_us = 1e6 * dt_max / max(1.0, tot[0] / D.world) * args.batch
        rec["roofline_iteration"] = {"bytes": it_bytes, "us": it_us, "GB/s": it_bytes / (it_us * 1e-6) / 1e9,
                                     "frac": it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "note": "six kernels of one ADMM iteration (2 PCG iterations), time in the launch graph incl. convergence checks"}
    solver.close()

    # ---- the same kernels over a lock-step batch (the regime where the SpMV leaves the latency floor) ----
    if D.world == 1 and args.batch == 1 and args.roofline_batch > 1:
        B = args.roofline_batch
        bm = models + make_headline(args, 1, B - 1)
        bs_ = ConicSolver([m.qp for m in bm], dict(base, polish=0, adaptive_cg=0))
        _, kb = bs_.time_kkt_apply(10)
        bdev, bdisp = bs_.time_iteration(warmup=10, iters=40, dispatch=True)
        bs_.close()
        rec["roofline_batch16"] = roofline_block(
            f"k_spmv_band<KP>, {B} headline problems in one lock-step handle", kb, bdisp["kp"], bdev["kp"],
            {"batch": B, "kernel_us_dispatch": bdisp, "kernel_us_device": bdev,
             "us_per_problem_iteration": sum(bdisp.values()) / B,
             "traffic": pmc_traffic("kp", "batch16_loop" if (headline_shape and B == 16) else None)})
        del bm

    if D.world == 1 and args.batch == 1 and polished and polished.get("newton_probe"):
        rec["roofline_newton"] = roofline_newton(args, dev, models[0].qp, polished["newton_probe"], headline_shape)
    if D.world == 1 and args.batch == 1:
        rec["three_d"] = three_d_leg(args, dev)
        rec["long_chains"] = long_chain_leg(args, dev)
        rec["config5_montecarlo"] = early["config5_montecarlo"]
        if not args.no_e2e:
            rec["config5_montecarlo"]["fresh_graphs_economy_waits"] = economy_leg(args)
        if not args.no_e2e:
            rec["end_to_end"] = early["end_to_end"]
    if D.world == 1 and not args.no_cpu_baseline:
        small = mc_models(args, [0])[0]
        gs = ConicSolver([small.qp], dict(base))  # product default on BASELINE configs[2]
        gs.solve()
        t0 = time.perf_counter()
        for _ in range(5):
            gout = gs.solve()[0]
        gpu_small_s = (time.perf_counter() - t0) / 5
        gs.close()
        cb = cpu_baseline(args, models, gpu_iters_per_solve=tot[0] / max(1.0, tot[1]), small_qp=small.qp)
        rec["cpu_baseline"] = cb
        rec["speedup_vs_cpu_baseline"] = rec["value"] / cb["value"]
        if polished:
            gpu_s = polished["ms_per_solve"] * 1e-3
            sp = {}
            for name, e in cb["entries"].items():
                ref = gpu_small_s if name.endswith("_config2") else gpu_s
                sp[name] = e["seconds_to_eps"] / ref if e["seconds_to_eps"] else None
            sp["gpu_seconds_headline"] = gpu_s
            sp["gpu_seconds_config2"] = gpu_small_s
            sp["gpu_config2_solved"] = bool(gout.solved)
            sp["note"] = ("ALGORITHM against ALGORITHM: CPU seconds to eps of each baseline (operator splitting alone) / GPU seconds of the "
                          "product DEFAULT solver (ADMM warm-up + Newton polish) on the SAME problem instance, same stopping rule, eps %g; "
                          "*_config2 entries on BASELINE configs[2], the others on the headline problem.  The hardware-against-hardware "
                          "figure is speedup_like_for_like" % args.eps)
            rec["speedup_time_to_solution"] = sp
        # like for like: the same algorithm (ADMM loop alone, same iteration count to eps) on both sides
        gpu_admm_s = dt_max / args.steps
        e = cb["entries"]["C2_twin_best_team"]
        rec["speedup_like_for_like"] = {
            "gpu_seconds_admm_only": gpu_admm_s, "cpu_seconds_admm_only": e["seconds_to_eps"], "cpu_threads": e["cores"],
            "ratio": (e["seconds_to_eps"] / gpu_admm_s) if e["seconds_to_eps"] else None,
            "what": "the same operator-splitting loop to the same eps on the headline problem: HIP kernels against the C++ twin at its "
                    "best OpenMP team on this host (= speedup_vs_cpu_baseline in time instead of iterations/s)"}
    if D.rank == 0:
        emit(args, rec)
    D.close()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    D = Dist(args)
    workload = args.workload
    if workload == "auto":  # the same primary metric at every N (a 1 -> 8 curve compares like with like)
        workload = "headline"
    if workload == "montecarlo":
        rec = run_montecarlo_leg(args, D)
        if rec is not None:
            emit(args, rec)
        D.close()
    else:
        run_headline(args, D)


if __name__ == "__main__":
    main()

