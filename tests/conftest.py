import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
HIP_LIB = os.path.join(ROOT, "score_amd", "csrc", "libscore_hip.so")
TWIN_LIB = os.path.join(ROOT, "oracle", "cpu_twin", "libscore_cpu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count() -> int:
    """Devices the HIP runtime sees (hipGetDeviceCount through ctypes: no torch.cuda initialisation)."""
    import ctypes

    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            lib = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        try:
            if lib.hipGetDeviceCount(ctypes.byref(n)) != 0:
                return 0
        except Exception:
            return 0
        return int(n.value)
    return 0


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing in
    score_create ("no HIP device available"); `-m gpu` on such a box therefore reports skips."""
    if not any("gpu" in it.keywords for it in items):
        return
    # a machine with the amdgpu compute node is a GPU box: never skip there (a broken runtime must
    # fail loudly, not turn the parity tests into skips)
    if os.path.exists("/dev/kfd") or _hip_device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device on this machine (gpu-marked tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def twin_lib():
    """The oracle's CPU twin of the solver (same C ABI), built on demand."""
    import __graft_entry__ as g

    return g.build_oracle()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library.  Built here if absent (hipcc cross-compiles)."""
    import __graft_entry__ as g

    return g.build_hip()


@pytest.fixture(scope="session")
def fixtures():
    from score_amd.io import load_fg_npz

    return {
        "manhattan": load_fg_npz(os.path.join(GOLDEN, "manhattan_fg.npz")),
        "goats": load_fg_npz(os.path.join(GOLDEN, "goats_fg.npz")),
    }


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, f"{name}_golden.npz"))


SYNTH = {
    "synth_a": dict(n_robots=1, n_poses=60, n_beacons=2, seed=11),
    "synth_b": dict(n_robots=3, n_poses=50, n_beacons=3, seed=12, n_loop_closures=4),
    "synth_c": dict(n_robots=2, n_poses=120, n_beacons=0, seed=13, p_range=0.3),
}


def graph_by_name(name, fixtures):
    from score_amd.manhattan import make_manhattan

    if name in fixtures:
        return fixtures[name]
    return make_manhattan(**SYNTH[name])


def compare_with_golden(res, gold, pose_tol=1e-4, check_landmarks=True):
    """Poses (rounded rotation + translation) and uniquely determined landmarks
    of a SolverResults against a golden optimum; relative to the largest
    coordinate, as north_star states the tolerance (1e-4 relative)."""
    import numpy as np

    from oracle import score_oracle as so

    P = gold["poses"]
    d = P.shape[1]
    scale = max(1.0, float(np.max(np.abs(P[:, :, d]))))
    worst_t, worst_R = 0.0, 0.0
    det = gold["pose_determined"] if "pose_determined" in gold.files else np.ones(len(P), dtype=bool)
    for i, nm in enumerate(gold["pose_names"]):
        if not det[i]:
            continue  # gauge freedom: compare through the objective instead
        T = res.poses[str(nm)]
        worst_t = max(worst_t, float(np.max(np.abs(T[:d, d] - P[i, :, d]))) / scale)
        worst_R = max(worst_R, float(np.max(np.abs(T[:d, :d] - so.round_to_special_orthogonal(P[i, :, :d])))))
    assert worst_t < pose_tol, f"translations differ by {worst_t:.3e} (relative)"
    assert worst_R < pose_tol, f"rounded rotations differ by {worst_R:.3e}"
    if check_landmarks:
        for i, nm in enumerate(gold["landmark_names"]):
            if gold["landmark_determined"][i]:
                err = float(np.max(np.abs(res.landmarks[str(nm)] - gold["landmarks"][i]))) / scale
                assert err < pose_tol, f"landmark {nm} differs by {err:.3e}"
    return worst_t, worst_R
