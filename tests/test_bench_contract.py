"""bench.py host logic that can run without a GPU (argument handling, workload
construction, JSON schema helpers)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(argv, tmp_path, env=None, timeout=900, **kw):
    """bench.py as the driver runs it; returns (process, compact record = LAST stdout line, full record = --full-out file).
    The contract (round 6, after BENCH_r05.json came back unparsed): stdout carries ONE json line, <= 4 KB."""
    full = os.path.join(str(tmp_path), "bench_full.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--full-out", full],
                         capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env, **kw)
    if out.returncode != 0:
        return out, None, None
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # ONE json line, from rank 0
    last = out.stdout.strip().splitlines()[-1]
    assert last == lines[0] and len(last.encode()) <= 4096, len(last)
    tail = out.stdout.encode()[-8192:].decode()
    compact = json.loads(tail.strip().splitlines()[-1])  # what a reader of the last 8 KB of stdout gets
    for v in compact.values():
        assert not (isinstance(v, str) and len(v) > 200)  # no prose in the contract line
    with open(full) as fh:
        return out, compact, json.load(fh)


def test_bench_cpu_baseline_leg_runs_and_reports(twin_lib, tmp_path):
    """The cpu_baseline leg of bench.py (oracle's CPU twin on a bounded sample)."""
    out, compact, rec = run_bench(["--cpu-baseline-only", "--robots", "2", "--poses", "60",
                                   "--mc-robots", "3", "--mc-poses", "80", "--cpu-seconds", "5"], tmp_path, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    cc = compact["cpu_baseline"]
    assert set(cc) == {"value", "unit", "cores", "kind", "seconds", "sample"} and cc["kind"] == "port" and cc["value"] > 0
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0
    assert cb["cores"] >= 1 and "sample" in cb
    # C1 (SciPy direct-KKT ADMM, 1 thread) and C2 (C++ twin, 1 thread / best team),
    # each with seconds to eps on the same problem
    assert set(cb["entries"]) == {"C1_scipy_direct_kkt_admm_config2", "C2_twin_1_thread_config2", "C2_twin_1_thread", "C2_twin_best_team"}
    for e in cb["entries"].values():
        assert e["kind"] == "port" and e["seconds_to_eps"] > 0 and e["cores"] >= 1
    for grp in (("C1_scipy_direct_kkt_admm_config2", "C2_twin_1_thread_config2"), ("C2_twin_1_thread", "C2_twin_best_team")):
        objs = [cb["entries"][k]["pobj"] for k in grp]
        assert max(objs) - min(objs) < 1e-4 * max(1.0, abs(objs[0]))  # baselines on one problem solve the same program


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def test_bench_gpus_flag_launches_that_many_ranks(twin_lib, tmp_path):
    """`python bench.py --gpus 2` must start two ranks by itself (no torchrun) and report n_gpus = 2 with the SAME
    primary metric as one rank (SOCP iterations/s, every rank its own headline problem: weak scaling), plus BASELINE
    configs[4] sharded t mod N with ONE all_gather of the result records per sweep.  Run here on gloo + the oracle's
    CPU twin (--test-cpu-twin); on the GPU box the same launcher starts RCCL ranks."""
    env = dict(os.environ, OMP_NUM_THREADS="2", SCORE_BENCH_TEST_MODE="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out, compact, rec = run_bench(
        ["--gpus", "2", "--test-cpu-twin", "--robots", "2", "--poses", "30",
         "--montecarlo", "5", "--mc-robots", "2", "--mc-poses", "30", "--mc-batch", "2", "--mc-threads", "1", "--steps", "1",
         "--warmup", "0"], tmp_path, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    for key in CONTRACT_KEYS + ("roofline", "cpu_baseline"):
        assert key in rec and key in compact
    assert compact["n_gpus"] == 2 and compact["metric"] == "socp_iters_per_sec" and compact["value"] > 0 and compact["scaling"] == "weak"
    assert compact["legs"]["config5_trials"] == 5 and compact["legs"]["config5_solved_last_sweep"] == 5
    assert compact["legs"]["config5_resolve_problems_per_sec"] > 0 and compact["legs"]["config5_fresh_graphs_problems_per_sec"] > 0
    assert rec["n_gpus"] == 2 and rec["metric"] == "socp_iters_per_sec" and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["timed_region"]["problems_total"] == 2 and rec["timed_region"]["problems_solved"] == 2  # one per rank
    c5 = rec["config5_montecarlo"]
    assert c5["n_gpus"] == 2 and c5["scaling"] == "strong" and c5["trials"] == 5 and c5["problems_per_sec"] > 0
    assert c5["results_gathered"] == 5 and c5["solved_last_sweep"] == 5  # 3 trials on rank 0, 2 on rank 1, all records on rank 0
    assert "gloo" in c5["gather"]
    # ... and what a Monte-Carlo study gets: every graph solved once, handle creation and model construction inside the timer
    # (score_create_from_graphs), beside the re-solve figure and labelled as such; every rank builds ITS shard's models only
    fr = c5["fresh_graphs"]
    assert c5["fresh_graphs_problems_per_sec"] == fr["problems_per_sec"] > 0 and "RE-SOLVES" in c5["problems_per_sec_is"]
    assert "generated_graphs_problems_per_sec" in c5 and c5["generated_graphs"] is None  # (the twin run skips the generated-worlds leg)
    assert fr["solved_last_sweep"] == 5 and fr["host_cpu_ms_per_problem_mean_over_ranks"] > 0
    assert sum(fr["trials_per_handle_rank0"]) == 3  # (rank 0 holds trials 0, 2, 4 of 5)
    assert "test_mode" in rec


def test_bench_is_one_of_the_ranks_under_a_launcher(twin_lib, tmp_path):
    """With RANK / WORLD_SIZE already in the environment (torch.distributed.run) bench.py must not
    spawn anything: world size 1 here, process group forced on (the gather then runs as a 1-rank collective)."""
    env = dict(os.environ, OMP_NUM_THREADS="2", SCORE_BENCH_TEST_MODE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29533")
    out, compact, rec = run_bench(
        ["--gpus", "1", "--test-cpu-twin", "--force-dist", "--robots", "1",
         "--poses", "30", "--beacons", "2", "--montecarlo", "2", "--mc-robots", "1", "--mc-poses", "30", "--mc-beacons", "2",
         "--steps", "1", "--warmup", "0"], tmp_path, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert compact["n_gpus"] == 1 and compact["metric"] == "socp_iters_per_sec"
    assert rec["n_gpus"] == 1 and rec["metric"] == "socp_iters_per_sec"
    assert rec["config5_montecarlo"]["solved_last_sweep"] == 2 and rec["config5_montecarlo"]["results_gathered"] == 2


def test_a_failing_rank_takes_the_launch_down_promptly(twin_lib):
    """The launcher polls every child: a rank that dies (here: an impossible workload on every rank) ends the whole
    launch with a non-zero code at once -- no rank is left waiting in a collective for the backend's timeout."""
    import time

    env = dict(os.environ, OMP_NUM_THREADS="2", SCORE_BENCH_TEST_MODE="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--test-cpu-twin", "--robots", "0", "--poses", "30",
         "--steps", "1", "--warmup", "0"],
        capture_output=True, text=True, timeout=300, cwd=ROOT, env=env,
    )
    assert out.returncode != 0 and time.time() - t0 < 120
    assert "exited with code" in out.stderr


def test_montecarlo_group_sizes():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.mc_groups(64, 16, 4) == [16, 16, 16, 16]
    assert bench.mc_groups(0, 16, 4) == []
    assert bench.mc_groups(8, 16, 4) == [4, 4] and bench.mc_groups(16, 16, 4) == [4, 4, 4, 4] and bench.mc_groups(32, 16, 4) == [8] * 4
    for n in (1, 2, 3, 8, 9, 32, 33, 100):
        g = bench.mc_groups(n, 16, 4)
        assert sum(g) == n and max(g) <= 16 and min(g) >= 1 and max(g) - min(g) <= 1


def test_cpu_twin_flag_needs_the_test_mode_switch():
    """`--test-cpu-twin` routes the timed region through the oracle's twin: bench.py refuses it unless
    SCORE_BENCH_TEST_MODE=1 is set as well, so that no measurement can be taken through oracle/ by accident."""
    env = dict(os.environ)
    env.pop("SCORE_BENCH_TEST_MODE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--test-cpu-twin", "--steps", "1"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "SCORE_BENCH_TEST_MODE" in out.stderr


def test_four_ranks_on_four_cpus_do_not_starve_each_other(twin_lib, tmp_path):
    """Host-side safety of the N-rank run: `bench.py --gpus 4` (gloo + the CPU twin, no GPU involved) with the launcher's
    affinity narrowed to four CPUs -- one per rank -- must not take much longer than with all CPUs: every rank sizes its
    host teams to its share of the CPUs it may use (host_threads(): affinity / cgroup quota / LOCAL_WORLD_SIZE), waits
    yield instead of spinning, and the launcher's children inherit the mask."""
    import time

    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 8:
        import pytest
        pytest.skip("needs 8 CPUs for the unpinned run")
    cmd = ["--gpus", "4", "--test-cpu-twin", "--robots", "2", "--poses", "40",
           "--montecarlo", "8", "--mc-robots", "2", "--mc-poses", "30", "--mc-batch", "2", "--steps", "1", "--warmup", "0"]
    env = dict(os.environ, OMP_NUM_THREADS="1", SCORE_BENCH_TEST_MODE="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)

    def run(mask):
        t0 = time.perf_counter()
        out, compact, rec = run_bench(cmd, tmp_path, env=env, preexec_fn=(lambda: os.sched_setaffinity(0, mask)) if mask else None)
        dt = time.perf_counter() - t0
        assert out.returncode == 0, out.stderr[-3000:]
        assert rec["n_gpus"] == 4 and rec["config5_montecarlo"]["results_gathered"] == 8
        return dt

    free = run(None)
    pinned = run(set(cpus[:4]))
    assert pinned <= 1.5 * free + 5.0, (free, pinned)


def test_the_contract_line_of_a_full_single_gpu_record_fits_4k():
    """BENCH_r05.json.parsed was null: the one stdout line had grown to 20.6 KB.  The compact record of the LARGEST full record
    this bench has produced (round 5's committed single-GPU run, every leg present) must fit 4 KB, carry the contract keys, a
    numeric roofline and cpu_baseline, and no prose."""
    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 16000
    c = bench.compact_record(full)
    line = json.dumps(c, separators=(",", ":"))
    assert len(line.encode()) <= bench.CONTRACT_MAX_BYTES, len(line)
    for key in CONTRACT_KEYS:
        assert key in c
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["us_per_launch"] * 1e-3) < 1e-2 * r["achieved"]
    assert c["roofline_dominant"]["frac"] > 0 and c["roofline_batch16"]["frac"] > 0
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["value"] > 0
    assert c["legs"]["config5_fresh_graphs_problems_per_sec"] > 0 and c["legs"]["solve_score_ms"] > 0
    assert abs(c["value"] - full["value"]) < 1e-4 * full["value"]
