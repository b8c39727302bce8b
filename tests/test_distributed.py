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
    # every distance variable, keyed as the reference keys them (gurobi_utils.py:127-136)
    dflat = np.concatenate([np.concatenate([np.asarray(r.variables.distances[k]).ravel() for k in sorted(r.variables.distances)]) for r in res])
    np.save(os.path.join(outdir, f"dist{rank}.npy"), dflat)
    resq = solve_score_sharded(graphs[:2], "QCQP", lib_path=lib, device=0)  # the QCQP directions: d values per range
    np.save(os.path.join(outdir, f"distq{rank}.npy"),
            np.concatenate([np.concatenate([np.asarray(r.variables.distances[k]).ravel() for k in sorted(r.variables.distances)]) for r in resq]))
    np.save(os.path.join(outdir, f"solved{rank}.npy"), np.array([r.solved for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_root(rank, world, port, lib, outdir):
    """Only rank 0 holds the graphs (one of them the reference's GOATS pickle); rank 1 passes None."""
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from score_amd.distributed import solve_score_sharded

    graphs = _root_graphs() if rank == 0 else None
    res = solve_score_sharded(graphs, "SOCP", lib_path=lib, device=0, root=0)
    flat = np.concatenate([np.concatenate([r.poses[n].ravel() for n in sorted(r.poses)] + [np.asarray(r.landmarks[n]).ravel() for n in sorted(r.landmarks)]) for r in res])
    np.save(os.path.join(outdir, f"root_rank{rank}.npy"), flat)
    np.save(os.path.join(outdir, f"root_solved{rank}.npy"), np.array([r.solved for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def _root_graphs():
    from score_amd.io import load_pyfg_pickle
    from score_amd.manhattan import make_manhattan

    graphs = [make_manhattan(n_robots=2, n_poses=30 + 7 * i, n_beacons=2, seed=51 + i) for i in range(3)]
    graphs.append(load_pyfg_pickle(os.path.join(ROOT, "tests", "golden", "goats_14_6_2002_15_20.pkl")))
    return graphs


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
    # the sharded results carry every field solve_score returns: the distance variables too, under the same keys
    d0, d1 = np.load(tmp_path / "dist0.npy"), np.load(tmp_path / "dist1.npy")
    np.testing.assert_array_equal(d0, d1)
    for r, g in zip(single, graphs):
        assert sorted(r.variables.distances) == sorted((m.first_key, m.second_key) for m in g.range_measurements)
    dflat = np.concatenate([np.concatenate([np.asarray(r.variables.distances[k]).ravel() for k in sorted(r.variables.distances)]) for r in single])
    assert dflat.size == sum(len(g.range_measurements) for g in graphs)
    np.testing.assert_allclose(d0, dflat, atol=1e-6)
    singleq = [solve_score(g, "QCQP", lib_path=twin_lib) for g in graphs[:2]]
    qflat = np.concatenate([np.concatenate([np.asarray(r.variables.distances[k]).ravel() for k in sorted(r.variables.distances)]) for r in singleq])
    assert qflat.size == 2 * sum(len(g.range_measurements) for g in graphs[:2])
    np.testing.assert_allclose(np.load(tmp_path / "distq0.npy"), qflat, atol=1e-6)
    np.testing.assert_array_equal(np.load(tmp_path / "distq0.npy"), np.load(tmp_path / "distq1.npy"))


def _worker_bad_root(rank, world, port, lib, outdir):
    """The root's graphs cannot be read (a duplicate pose name): every rank must raise, none may hang in the collective."""
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from score_amd.distributed import solve_score_sharded
    from score_amd.manhattan import make_manhattan

    graphs = None
    if rank == 0:
        graphs = [make_manhattan(n_robots=2, n_poses=20, n_beacons=2, seed=5)]
        graphs[0].pose_variables[1][3].name = graphs[0].pose_variables[1][2].name
    try:
        solve_score_sharded(graphs, "SOCP", lib_path=lib, device=0, root=0)
        msg = "no error"
    except ValueError as exc:
        msg = f"ValueError: {exc}"
    with open(os.path.join(outdir, f"bad{rank}.txt"), "w") as f:
        f.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_a_root_that_cannot_read_its_graphs_fails_on_every_rank(twin_lib, tmp_path):
    port = _free_port()
    mp.spawn(_worker_bad_root, args=(2, port, twin_lib, str(tmp_path)), nprocs=2, join=True)
    for rank in (0, 1):
        msg = (tmp_path / f"bad{rank}.txt").read_text()
        assert msg.startswith("ValueError: broadcast_graphs: rank 0 could not read its graphs"), msg
        assert "already exists" in msg, msg


def test_root_rank_broadcasts_the_graphs(twin_lib, tmp_path):
    """north_star's "broadcast/gather of problem data and results": rank 0 alone holds the data set (three synthetic graphs
    and the reference's goats_14 pickle), rank 1 is given None; both return every result, equal to a single-process solve."""
    port = _free_port()
    mp.spawn(_worker_root, args=(2, port, twin_lib, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "root_rank0.npy"), np.load(tmp_path / "root_rank1.npy")
    np.testing.assert_array_equal(r0, r1)
    assert np.load(tmp_path / "root_solved0.npy").all() and np.load(tmp_path / "root_solved1.npy").all()
    from score_amd.solve_score import solve_score

    single = [solve_score(g, "SOCP", lib_path=twin_lib) for g in _root_graphs()]
    flat = np.concatenate([np.concatenate([r.poses[n].ravel() for n in sorted(r.poses)] + [np.asarray(r.landmarks[n]).ravel() for n in sorted(r.landmarks)]) for r in single])
    np.testing.assert_allclose(r0, flat, atol=1e-6)


_NCCL_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r}, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from score_amd.distributed import solve_score_sharded
from score_amd.manhattan import make_manhattan
from score_amd.solve_score import solve_score_batch
graphs = [make_manhattan(n_robots=2, n_poses=30 + 7 * i, n_beacons=2, seed=51 + i) for i in range(5)]
assert dist.get_backend() == "nccl"
res = solve_score_sharded(graphs, "SOCP", device=0)          # solve on this GPU, records through ONE RCCL all_gather
ref = solve_score_batch(graphs, "SOCP", solver_settings=dict(device=0))
worst = 0.0
for a, b in zip(res, ref):
    assert a.solved and b.solved
    for n in b.poses:
        worst = max(worst, float(np.abs(a.poses[n] - b.poses[n]).max()))
    for n in b.landmarks:
        worst = max(worst, float(np.abs(a.landmarks[n] - b.landmarks[n]).max()))
assert worst <= 1e-9, worst
res2 = solve_score_sharded(graphs, "SOCP", device=0, root=0)  # the graphs broadcast from rank 0 first (RCCL broadcast through the GPU)
for a, b in zip(res2, ref):
    for n in b.poses:
        worst = max(worst, float(np.abs(a.poses[n] - b.poses[n]).max()))
    assert list(a.variables.distances) == list(b.variables.distances)
    for k in b.variables.distances:
        worst = max(worst, float(np.abs(np.asarray(a.variables.distances[k]) - np.asarray(b.variables.distances[k])).max()))
assert worst <= 1e-9, worst
# a Monte-Carlo study from seeds: the worlds drawn on this rank's GPU, built from there, records through RCCL
from score_amd.distributed import solve_generated_sharded
from score_amd.generate import generate_manhattan
gen = solve_generated_sharded(6, seed=77, relaxation_type="SOCP", device=0, n_robots=3, n_poses=200, n_beacons=3)
ref3 = solve_score_batch(generate_manhattan(6, seed=77, n_robots=3, n_poses=200, n_beacons=3), "SOCP", solver_settings=dict(device=0))
for a, b in zip(gen, ref3):
    assert a.solved and b.solved and list(a.variables.distances) == list(b.variables.distances)
    worst = max(worst, float(np.abs(a.poses.array - b.poses.array).max()), float(np.abs(a.variables.distances.array - b.variables.distances.array).max()))
assert worst <= 1e-9, worst
dist.barrier()
dist.destroy_process_group()
print("NCCL_SHARDED_OK", worst)
"""


@pytest.mark.gpu
def test_sharded_solve_under_rccl_equals_the_batch_solve(hip_lib):
    """The N > 1 path on real hardware as far as one GPU allows: solve_score_sharded in a process group with backend
    "nccl" (= RCCL) and world size 1 -- device binding, the product's HIP solves, and the all_gather of the result
    records through the GPU -- equals solve_score_batch.  Runs in a child process started before this one has a
    process group (never a re-exec of a GPU-initialised process)."""
    import subprocess

    code = _NCCL_CHILD.format(root=ROOT, port=str(_free_port()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "NCCL_SHARDED_OK" in out.stdout


def _worker_generated(rank, world, port, lib, outdir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from score_amd.distributed import solve_generated_sharded

    res = solve_generated_sharded(5, seed=900, relaxation_type="SOCP", lib_path=lib, device=0, n_robots=2, n_poses=40, n_beacons=3, p_range=0.3)
    np.save(os.path.join(outdir, f"gen{rank}.npy"), np.concatenate([r.poses.array.ravel() for r in res]))
    np.save(os.path.join(outdir, f"gend{rank}.npy"), np.concatenate([r.variables.distances.array.ravel() for r in res]))
    with open(os.path.join(outdir, f"genk{rank}.txt"), "w") as f:
        f.write(repr([list(r.variables.distances)[:3] for r in res]) + repr([r.info["rank"] for r in res]))
    dist.destroy_process_group()


def test_generated_worlds_sharded_over_two_ranks(twin_lib, tmp_path):
    """A Monte-Carlo study from seeds over two ranks (gloo, CPU twin): every rank draws and solves ITS block of worlds (no problem
    data is distributed), every rank ends with every estimate -- equal to the single-process run, distance keys included."""
    port = _free_port()
    mp.spawn(_worker_generated, args=(2, port, twin_lib, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "gen0.npy"), np.load(tmp_path / "gen1.npy")
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_array_equal(np.load(tmp_path / "gend0.npy"), np.load(tmp_path / "gend1.npy"))
    assert open(tmp_path / "genk0.txt").read() == open(tmp_path / "genk1.txt").read()
    assert open(tmp_path / "genk0.txt").read().endswith("[0, 0, 0, 1, 1]")  # worlds 0-2 on rank 0, 3-4 on rank 1
    from score_amd.distributed import solve_generated_sharded
    from score_amd.generate import generate_manhattan
    from score_amd.solve_score import solve_score_batch

    single = solve_generated_sharded(5, seed=900, relaxation_type="SOCP", lib_path=twin_lib, n_robots=2, n_poses=40, n_beacons=3, p_range=0.3)
    np.testing.assert_allclose(g0, np.concatenate([r.poses.array.ravel() for r in single]), atol=1e-6)
    direct = solve_score_batch(generate_manhattan(5, seed=900, n_robots=2, n_poses=40, n_beacons=3, p_range=0.3, lib_path=twin_lib), "SOCP", lib_path=twin_lib)
    for a, b in zip(single, direct):
        assert a.solved and list(a.variables.distances) == list(b.variables.distances) and a.pose_chain_names[1][5] == "B5"
        np.testing.assert_allclose(a.poses.array, b.poses.array, atol=1e-6)
        np.testing.assert_allclose(a.variables.distances.array, b.variables.distances.array, atol=1e-6)


def _worker_root_generated(rank, world, port, lib, outdir):
    """Only rank 0 holds the worlds -- GENERATED ones, whose arrays carry the generator's handle ('_owner') and lazy name tables."""
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from score_amd.distributed import broadcast_graphs, solve_score_sharded
    from score_amd.generate import generate_manhattan

    graphs = generate_manhattan(3, seed=901, n_robots=2, n_poses=30, n_beacons=2, p_range=0.3, lib_path=lib) if rank == 0 else None
    got = broadcast_graphs(graphs, root=0)
    assert all("_owner" not in g.arrays and isinstance(g.arrays["pose_names"], list) for g in got)  # ordinary array graphs everywhere
    res = solve_score_sharded(graphs, "SOCP", lib_path=lib, device=0, root=0)
    np.save(os.path.join(outdir, f"rootgen{rank}.npy"), np.concatenate([np.concatenate([r.poses[n].ravel() for n in sorted(r.poses)]) for r in res]))
    with open(os.path.join(outdir, f"rootgenk{rank}.txt"), "w") as f:
        f.write(repr([sorted(r.variables.distances)[:4] for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_root_rank_broadcasts_generated_worlds(twin_lib, tmp_path):
    """Advisor finding (round 5): broadcast_graphs could not ship worlds of score_amd.generate -- their arrays hold '_owner' (a
    GeneratedBatch with a CDLL inside), which does not pickle.  Private keys now stay on the root and the lazy name tables travel
    as lists: two ranks, rank 0 alone holds three generated worlds; both return every estimate, equal to the single-process solve."""
    port = _free_port()
    mp.spawn(_worker_root_generated, args=(2, port, twin_lib, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rootgen0.npy"), np.load(tmp_path / "rootgen1.npy")
    np.testing.assert_array_equal(r0, r1)
    assert open(tmp_path / "rootgenk0.txt").read() == open(tmp_path / "rootgenk1.txt").read()
    from score_amd.generate import generate_manhattan
    from score_amd.solve_score import solve_score_batch

    single = solve_score_batch(generate_manhattan(3, seed=901, n_robots=2, n_poses=30, n_beacons=2, p_range=0.3, lib_path=twin_lib), "SOCP", lib_path=twin_lib)
    np.testing.assert_allclose(r0, np.concatenate([np.concatenate([r.poses[n].ravel() for n in sorted(r.poses)]) for r in single]), atol=1e-6)
