"""Names user code imports from the reference's score/utils/gurobi_utils.py
(relaxation constants, :26-34); the Gurobi model builders themselves are
replaced by score_amd.assemble + the HIP solver."""
from score_amd.assemble import (  # noqa: F401
    ACCEPTABLE_RELAXATIONS,
    QCQP_RELAXATION,
    SOCP_RELAXATION,
)

RANDOM_INIT, ZERO_INIT, ODOM_INIT, GT_INIT = "random", "zero", "odom", "gt"
ACCEPTABLE_INIT = [RANDOM_INIT, ZERO_INIT, ODOM_INIT, GT_INIT]
