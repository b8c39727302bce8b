"""world_size-2 gloo test of the sharded Monte-Carlo path (CPU twin as backend)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from score_amd.distributed import shard_assignment


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lib, outdir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from score_amd.distributed import solve_score_sharded
    from score_amd.manhattan import make_manhattan

    graphs = [make_manhattan(n_robots=2, n_poses=30 + 7 * i, n_beacons=2, seed=51 + i) for i in range(5)]
    res = solve_score_sharded(graphs, "SOCP", lib_path=lib, device=0)
    flat = np.concatenate([np.concatenate([r.poses[n].ravel() for n in sorted(r.poses)]) for r in res])
    np.save(os.path.join(outdir, f"rank{rank}.npy"), flat)
    np.save(os.path.join(outdir, f"solved{rank}.npy"), np.array([r.solved for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_assignment_is_balanced_and_deterministic():
    costs = [10, 1, 7, 3, 8, 2, 9]
    a = shard_assignment(costs, 3)
    assert a == shard_assignment(costs, 3)
    assert sorted(i for s in a for i in s) == list(range(7))
    loads = [sum(costs[i] for i in s) for s in a]
    assert max(loads) - min(loads) <= max(costs)


def test_two_rank_gloo_matches_single_process(twin_lib, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, twin_lib, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(r0, r1)  # every rank holds all results
    assert np.load(tmp_path / "solved0.npy").all()
    from score_amd.manhattan import make_manhattan
    from score_amd.solve_score import solve_score

    graphs = [make_manhattan(n_robots=2, n_poses=30 + 7 * i, n_beacons=2, seed=51 + i) for i in range(5)]
    single = [solve_score(g, "SOCP", lib_path=twin_lib) for g in graphs]
    flat = np.concatenate([np.concatenate([r.poses[n].ravel() for n in sorted(r.poses)]) for r in single])
    np.testing.assert_allclose(r0, flat, atol=1e-6)
