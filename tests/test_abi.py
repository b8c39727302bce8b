"""The product library loads without a GPU, exports every symbol of
include/score_hip.h, and fails LOUDLY (no CPU fallback) when no device exists."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from score_amd.assemble import assemble
from score_amd.manhattan import make_manhattan
from score_amd.solver import ABI_SYMBOLS, ConicSolver, load_library


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "score_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(score_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(ABI_SYMBOLS)


def test_hip_library_exports_every_symbol(hip_lib):
    lib = load_library(hip_lib)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    assert lib.score_backend().decode() == "hip-gfx950"


def test_twin_exports_the_same_abi(twin_lib):
    lib = load_library(twin_lib)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    assert lib.score_backend().decode() == "cpu-twin"


def test_product_path_has_no_cpu_fallback(hip_lib):
    """Without a HIP device score_create must fail (this container has none);
    on a GPU box the same call succeeds, which the gpu tests cover."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    fg = make_manhattan(n_robots=1, n_poses=6, n_beacons=1, seed=0, p_range=1.0)
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        ConicSolver(assemble(fg, "SOCP").qp)  # default lib = the HIP library


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        load_library(str(tmp_path / "libscore_hip.so"))


def test_product_package_does_not_import_the_oracle():
    """score_amd/ must not reference oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "score_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_score_import_shim_matches_reference_paths():
    """`from score.solve_score import solve_score` etc. keep working (drop-in)."""
    from score.solve_score import solve_problem_with_intermediate_iterates, solve_score  # noqa: F401
    from score.utils.gurobi_utils import ACCEPTABLE_RELAXATIONS, QCQP_RELAXATION, SOCP_RELAXATION
    from score.utils.matrix_utils import round_to_special_orthogonal  # noqa: F401
    from score.utils.solver_utils import ScoreSolverParams

    assert (SOCP_RELAXATION, QCQP_RELAXATION) == ("SOCP", "QCQP") and len(ACCEPTABLE_RELAXATIONS) == 2
    assert ScoreSolverParams(solver="gurobi", verbose=True, save_results=True, init_technique="none",
                             custom_init_file=None).init_technique == "none"
    import inspect

    sig = inspect.signature(solve_score)
    assert list(sig.parameters)[0] == "data" and sig.parameters["relaxation_type"].default == "QCQP"


def test_the_switch_list_is_the_one_the_sources_read():
    """No switch enters the library without entering SURVIVING_SWITCHES (and INTEGRATION.md): the getenv calls of the product's
    sources are exactly that list (+ LOCAL_WORLD_SIZE, which the launchers export)."""
    import glob
    import re

    from conftest import ROOT, SURVIVING_SWITCHES

    found = set()
    for path in glob.glob(os.path.join(ROOT, "score_amd", "csrc", "*")):
        if path.endswith((".hpp", ".hip", ".c")):
            found |= set(re.findall(r'getenv\("(SCORE_[A-Z0-9_]+)"\)', open(path).read()))
    assert found == {s for s, _ in SURVIVING_SWITCHES}, found ^ {s for s, _ in SURVIVING_SWITCHES}
    assert len(found) <= 15
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for s in found:
        assert s in doc, s
