"""bench.py host logic that can run without a GPU (argument handling, workload
construction, JSON schema helpers)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_cpu_baseline_leg_runs_and_reports(twin_lib):
    """The cpu_baseline leg of bench.py (oracle's CPU twin on a bounded sample)."""
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--robots", "2", "--poses", "60",
         "--cpu-seconds", "2"],
        capture_output=True, text=True, timeout=600, cwd=ROOT,
    )
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0
    assert rec["cpu_baseline"]["cores"] >= 1 and "sample" in rec["cpu_baseline"]
