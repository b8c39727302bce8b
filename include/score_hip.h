/*
 * score_hip.h -- C ABI of the MI355X conic solver that replaces the reference's
 * solver boundary for the SCORE relaxation.
 *
 * What it replaces.  The reference builds a Gurobi model
 * (score/utils/gurobi_utils.py:173-187 initialize_model) and crosses into
 * native code exactly once, at `model.optimize()` (score/solve_score.py:76),
 * then reads `.X`, `.Runtime`, `.status` back (gurobi_utils.py:133-135,
 * :194-195).  gurobipy's object API cannot be re-bound, so the boundary is
 * restated as "hand over the assembled convex program, get the optimiser
 * back":
 *
 *     minimise    1/2 x'Px + q'x + c0
 *     subject to  A x + s = b,   s in {0}^z  x  SOC(d_1) x ... x SOC(d_k)
 *
 *   score_create[_batch]   <-  gp.Model() + addMVar/addVars/addConstr/
 *                              setObjective/update
 *                              (gurobi_utils.py:206-215, :233-352, :186-187)
 *   score_solve            <-  model.optimize()           (solve_score.py:76)
 *   outputs x / y / s      <-  Var.X                (gurobi_utils.py:133-135)
 *   score_info.status      <-  model.status == GRB.OPTIMAL        (:195)
 *   score_info.solve_ms    <-  model.Runtime                      (:194)
 *   score_solve_steps      <-  BarIterLimit = k; optimize()
 *                              (solve_score.py:103-105, intermediate iterates)
 *   score_destroy          <-  model disposal
 *
 * Conventions.  Plain pointers and sizes only.  All input arrays are BORROWED
 * for the duration of the call and copied to the device in score_create*;
 * outputs are written into caller-owned buffers.  Return value 0 = ok,
 * negative = error (message via score_last_error(), thread-local).  No
 * exceptions cross the ABI.  One handle = one HIP device + one stream; a
 * handle is not thread-safe, distinct handles are independent.  A handle may
 * hold a BATCH of independent problems that advance in lock-step through the
 * same kernel launches (per-problem step sizes, penalty and termination).
 * There is no CPU fallback: without a usable HIP device score_create* fails.
 */
#ifndef SCORE_HIP_H
#define SCORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct score_problem {
    int32_t n;                 /* unknowns                                   */
    int32_t m;                 /* constraint rows                            */
    /* P: FULL symmetric n x n matrix in CSR (both triangles), sorted columns */
    const int32_t* P_rowptr;   /* n + 1                                      */
    const int32_t* P_col;
    const double*  P_val;
    const double*  q;          /* n                                          */
    double         c0;         /* objective constant                         */
    /* A: m x n in CSR; rows ordered: z zero-cone rows, then the SOC blocks  */
    const int32_t* A_rowptr;   /* m + 1                                      */
    const int32_t* A_col;
    const double*  A_val;
    const double*  b;          /* m                                          */
    int32_t        z;          /* leading zero-cone (equality) rows          */
    int32_t        n_soc;      /* number of second-order cones               */
    const int32_t* soc_dims;   /* n_soc entries, each >= 1, sum == m - z     */
    /* Optional block-tridiagonal preconditioner hint (0 / NULL = none):
     * chain c consists of nodes chain_ptr[c] .. chain_ptr[c+1]-1 in order;
     * node j owns the block_size consecutive columns
     * node_first_col[j] .. node_first_col[j] + block_size - 1.  Consecutive
     * nodes of a chain are assumed to be the strongly coupled ones (for SCORE:
     * one chain per robot and pose-matrix row, one node per pose).  Columns in
     * no node get a Jacobi preconditioner.                                   */
    int32_t        block_size; /* 1..4                                       */
    int32_t        n_chains;
    const int32_t* chain_ptr;  /* n_chains + 1                               */
    const int32_t* node_first_col;
    /* Optional row-replication hint (0 = none).  The SCORE model is a sum of terms in ONE row k of the
     * pose matrices [R | t] at a time (gurobi_utils.py:504-526: row k of R_j - R_i R~ and of
     * t_j - t_i - R_i t~ only holds row k of the variables), and only the cones couple the d rows
     * (:345-352).  With the unknowns ordered replica by replica,
     *     columns [k * rep_n, (k + 1) * rep_n), k = 0..rep_d-1  = the rep_n unknowns of matrix row k,
     *     columns [rep_d * rep_n, n)                            = the remaining ("tail") unknowns,
     * the quadratic part is P = I_{rep_d} (x) P_row (+ a tail block), every cone has rep_d + 1 rows --
     * a head row that holds tail columns only and rep_d rows that hold the same entries, replica by
     * replica -- and the chains come replica by replica as well.  The KKT operator then is
     * K = I (x) K_row (+ tail): the solver stores and streams K_row ONCE and applies it to the rep_d
     * right-hand sides together, and factors one set of chains for all replicas.  The hint is CHECKED
     * at score_create (pattern and values, to 1e-12); a problem that does not have the structure is
     * solved as a general one -- the hint never changes the result.                                   */
    int32_t        rep_d;      /* replicas (SCORE: the dimension d); 0 or 1 = no hint */
    int32_t        rep_n;      /* unknowns per replica                        */
} score_problem;

typedef struct score_settings {
    double  eps_abs;           /* absolute tolerance (unscaled residuals)    */
    double  eps_rel;           /* relative tolerance                         */
    int32_t max_iters;         /* ADMM iteration cap                         */
    int32_t check_interval;    /* iterations per launch graph / convergence test */
    double  rho;               /* initial penalty                            */
    double  sigma;             /* proximal weight on x                       */
    double  alpha;             /* over-relaxation in (0, 2)                  */
    int32_t scale_iters;       /* Ruiz equilibration passes (0 = off)        */
    int32_t cg_iters;          /* PCG iterations per ADMM iteration (initial) */
    int32_t adaptive_cg;       /* 0/1: adapt cg_iters to the measured reduction */
    int32_t max_cg_iters;      /* cap for adaptive_cg                        */
    double  cg_target;         /* wanted M^-1-norm residual reduction per KKT solve */
    int32_t adaptive_rho;      /* 0/1                                        */
    int32_t adaptive_rho_interval; /* in ADMM iterations (multiple of check_interval) */
    double  adaptive_rho_tol;  /* refactor when rho changes by this factor   */
    int32_t chain_radix;       /* partition radix of the chain solver (2..4) */
    int32_t device;            /* HIP device ordinal                         */
    int32_t use_graph;         /* replay iterations from a hipGraph          */
    int32_t polish;            /* 0/1: semismooth-Newton polish once ADMM is close (programs whose cones have
                                  private head variables: the SCORE SOCP form, and the QCQP form after the
                                  library's rewrite, score_headform.hpp)                                 */
    double  polish_start;      /* start it when both relative residuals are below this                  */
    int32_t polish_warmup;     /* ADMM iterations before the first polish attempt (0: one check_interval) */
    int32_t verbose;
    int32_t chain_split;       /* chain preconditioner: 0 = one workgroup per chain (default); 1 = split every long
                                  chain over 2-4 workgroups (k_prec_wave) when the whole launch fits the device at
                                  once -- correct and tested, measured 10-15 % SLOWER than the default on the
                                  headline problem, kept as an option                      */
    int32_t fac_fp32;          /* chain factors kept to float precision (rounded after every factorisation) and
                                  streamed as 4-byte values by the LDS-resident chain kernel -- half the bytes of the
                                  kernel that owns the iteration; M^-1 stays a fixed linear operator, PCG converges
                                  to the same tolerances.  1 (default): the ADMM loop's factors (of K);  2: the Newton
                                  polish's too (same iteration counts on the BASELINE sizes, ~10 % more Newton PCG
                                  iterations on small ill-conditioned graphs);  0: double throughout.
                                  With 1 the Newton factors follow the 4-byte stream by themselves when every
                                  chain has >= 256 nodes (same iteration counts there, 7 % faster default solve).
                                  3-D problems (4 x 4 chain blocks): 1 covers the Newton factors as well -- the
                                  LDS-resident chain kernel exists for the 4-byte stream only there, the streaming
                                  kernel that double factors need is three times slower.                        */
} score_settings;

enum {
    SCORE_STATUS_UNSOLVED = 0,
    SCORE_STATUS_SOLVED = 1,          /* all three termination tests passed  */
    SCORE_STATUS_MAX_ITERS = 2,
    SCORE_STATUS_NUMERICAL = 3        /* NaN/Inf encountered                 */
};

typedef struct score_info {
    int32_t status;
    int32_t iters;             /* ADMM ("SOCP") iterations                   */
    int32_t cg_iters;          /* total PCG iterations                       */
    int32_t rho_updates;
    double  rho;               /* final penalty                              */
    double  pobj;              /* primal objective incl. c0                  */
    double  dobj;              /* dual objective incl. c0                    */
    double  res_pri;           /* |Ax + s - b|_inf                           */
    double  res_dual;          /* |Px + q + A'y|_inf                         */
    double  gap;               /* |pobj - dobj|                              */
    double  setup_ms;          /* score_create time                          */
    double  solve_ms;          /* wall time of the last score_solve          */
    double  kkt_bytes;         /* algorithmic bytes of one K-apply (this problem) */
    int32_t newton_iters;      /* Newton iterations of the polish (0 = not run) */
    int32_t newton_cg_iters;   /* PCG iterations spent inside the polish     */
} score_info;

typedef struct score_handle score_handle;

void score_default_settings(score_settings* s);

/* Build a solver for ONE problem / for a batch of `count` independent problems.
 * Programs whose cones all have a CONSTANT head and private tail columns (the reference's default "QCQP" relaxation:
 * ||r_ij|| <= 1, gurobi_utils.py:341-344, cost :488-496) are rewritten into private-head cones and solved in that form
 * (score_amd/csrc/score_headform.hpp) -- the same optimum, with the Newton polish; sizes, x, y, s at this boundary stay those
 * of the program as given.  SCORE_QCQP_PLAIN=1 in the environment: the plain ADMM loop on the program as given.            */
int  score_create(const score_problem* p, const score_settings* s, score_handle** out);
int  score_create_batch(const score_problem* p, int32_t count, const score_settings* s,
                        score_handle** out);

/* A handle for `count` factor graphs (struct score_graph, below), model construction included: the whole of the reference's
 * `initialize_model` (gurobi_utils.py:173-187: variables :221-310, pin :316-333, cones :336-352, objective :358-526) plus the
 * model's creation (:206-215) in one call.  The graphs' flat arrays are all that crosses to the device: every measurement
 * writes its terms there, the conic program (P, q, A, b) is merged from them and equilibrated, K, A' and the Newton matrix are
 * built from it -- the program score_assemble would build, entry by entry (bit-equal P, q, A, b; the host assembler stays as
 * the specification and takes over when the device path is switched off: SCORE_HOST_ASSEMBLE / SCORE_HOST_SETUP).  Unknowns
 * are ordered as score_assemble orders them; solutions come back through score_solve as for any other handle.            */
struct score_graph;
int  score_create_from_graphs(const struct score_graph* graphs, int32_t count, const score_settings* s,
                              score_handle** out);

/* The reference's graph check (score/solve_score.py:28-32: assert data.unconnected_variable_names == []) on the flat arrays:
 * 0 = every variable of every graph is touched by a measurement or a prior, i > 0 = graph i - 1 is the first that has
 * unconnected variables, < 0 = error.  (score_create_from_graphs itself builds whatever it is given, like initialize_model.) */
int  score_graphs_connected(const struct score_graph* graphs, int32_t count);

/* After a solve of a handle made by score_create_from_graphs: the estimate in the reference's own shapes, straight from the
 * solution on the device -- replaces VariableCollection.get_variable_values (gurobi_utils.py:114-136; extract_solver_results
 * :190-203 wraps it).  Problem after problem:
 *   poses      n_poses x (d+1) x (d+1)  homogeneous [[R, t], [0, 1]], R = the relaxed block rounded onto SO(d) as
 *                                       score_round_to_so rounds it (gurobi_utils.py:115-125); pose 0 is the pinned [I | 0]
 *   relaxed    n_poses x d x (d+1)      the relaxation's own blocks [R | t]
 *   landmarks  n_landmarks x d
 *   ranges     n_ranges x 1 (SOCP distances) or x d (QCQP directions; qcqp_directions != 0 on an SOCP program: the optimal
 *                                       directions r = D / max(|D|, dist) of the equivalent QCQP, gurobi_utils.py:488-496)
 *   degenerate n_poses                  1 where the rounding is not unique (the block is the identity then)
 * Any output may be NULL.  Handles made from score_problem arrays do not know the graph: error.                            */
int  score_read_estimates(score_handle* h, int32_t qcqp_directions, double* poses, double* relaxed, double* landmarks,
                          double* ranges, int32_t* degenerate);

/* Concatenated sizes of the handle's problems (sum of n, sum of m, count).  */
int  score_dims(const score_handle* h, int64_t* n_total, int64_t* m_total, int32_t* count);

/* Cold-start solve.  x: sum n, y and s: sum m (problem after problem), any may
 * be NULL; infos: `count` entries (may be NULL).                             */
int  score_solve(score_handle* h, double* x, double* y, double* s, score_info* infos);

/* Run exactly `iters` more ADMM iterations from the current iterate (after
 * score_reset: from zero), then report the iterate -- the intermediate-iterate
 * interface (solve_score.py:89-116).                                         */
int  score_reset(score_handle* h);
int  score_solve_steps(score_handle* h, int32_t iters, double* x, double* y, double* s,
                       score_info* infos);

/* The same for the second phase of the default solver: at most `iters` semismooth-Newton iterations
 * of the polish from the current iterate (a backend or problem without the polish leaves the
 * iterate alone), then the report.  score_info.newton_iters accumulates.               */
int  score_newton_steps(score_handle* h, int32_t iters, double* x, double* y, double* s,
                        score_info* infos);

/* Time `reps` applications of the KKT operator w = K p on the handle's stream
 * with HIP events (the roofline probe bench.py reports); returns the average
 * milliseconds per launch in *ms_per_apply and the algorithmic bytes of one
 * launch in *bytes_per_apply.                                                */
int  score_time_kkt_apply(score_handle* h, int32_t reps, double* ms_per_apply,
                          double* bytes_per_apply);

/* Time the six kernels of the ADMM iteration IN THE LOOP: resets the solver (iterates, penalties,
 * PCG count), runs `warmup` + `iters` iterations (2 PCG iterations each: rhs, prec_init, kp,
 * prec_step, kpb, cone) on the handle's stream and returns each kernel's average duration in
 * microseconds (same order):
 *   us[0..5]   on the device -- first workgroup in to last workgroup out, constant-rate wall clock;
 *              no command is inserted between two kernels;
 *   us[6..11]  (with_events != 0; a second pass over further iterations) begin to end of each
 *              DISPATCH as the runtime records it: start/stop HIP events bound to the launch itself
 *              (hipExtLaunchKernel) on the handle's stream -- the interval a profiler's kernel trace
 *              (rocprofv3 --kernel-trace) reports for the same launches; zeros otherwise.
 * Unlike back-to-back launches of one kernel, every kernel finds the caches as the loop leaves
 * them.  Requires settings.cg_iters == 2; leaves the iterates advanced.  `us` holds 12 doubles.  */
int  score_time_iteration(score_handle* h, int32_t warmup, int32_t iters, double* us, int32_t with_events);

/* Debug: average milliseconds of `reps` back-to-back launches of one kernel of the
 * iteration ("rhs", "prec_init", "prec_step", "kp", "kpb", "xupdate", "cone"),
 * HIP events on the handle's stream.  Leaves the iterates in an undefined state
 * (call score_reset afterwards).                                              */
int  score_debug_time(score_handle* h, const char* kernel, int32_t reps, double* ms);

/* Debug/test access to an internal device vector by name ("x", "xt", "s", "y",
 * "u", "r", "z", "p", "w"); copies min(len, size) doubles, returns the size. */
int64_t score_debug_get(score_handle* h, const char* name, double* out, int64_t len);

void score_destroy(score_handle* h);

/* Handles come and go at a high rate in Monte-Carlo use, so device blocks, pinned host blocks and streams of
 * destroyed handles are parked in a process-wide cache (device blocks: at most an eighth of the device memory, 32 GiB at most;
 * pinned host blocks: at most 4 GiB; SCORE_CACHE_MB overrides both) and reused by the
 * next score_create.  score_trim_caches releases everything parked (live handles are not touched) and returns the
 * bytes freed; the cache also releases itself and retries once when an allocation fails.                        */
int64_t score_trim_caches(void);
/* What the library's threads spent waiting for the device since the process started (process-wide, all handles):
 * out[0] = milliseconds SPINNING on host-mapped result words (a thread alone in its solve: lowest latency; burns a CPU),
 * out[1] = milliseconds ASLEEP between looks (several solves at once, or several ranks on the node: "economy" waits,
 * csrc/score_hip.hip), out[2] = waits, out[3] = sleeps.  Returns the number of counters (4); fills min(len, 4).
 * The reference blocks inside model.optimize() (score/solve_score.py:76) and reports nothing of the kind; bench.py uses the
 * counters to split the host CPU per problem into work and wait.                                                  */
int32_t score_host_counters(double* out, int32_t len);

/* ---------------------------------------------------------------------------
 * Native model construction: the factor graph as flat arrays -> the conic program above.
 * Replaces, for the same path, the reference's `initialize_model`
 * (score/utils/gurobi_utils.py:173-187: variables :221-310, pinned first pose :316-333, cones
 * :336-352, objective :358-526), which the Python host otherwise performs in
 * score_amd/assemble.py.  Poses are numbered chain by chain (chain 0 first); the first pose of
 * chain 0 is the pinned one.  Range endpoints are variable ids: 0..n_poses-1 = poses,
 * n_poses..n_poses+n_landmarks-1 = landmarks.  Column layout of the result: see
 * score_amd/csrc/score_assemble.hpp (identical to score_amd/assemble.py).
 * ------------------------------------------------------------------------- */
typedef struct score_graph {
    int32_t dim;                /* 2 or 3                                      */
    int32_t relaxation;         /* 0 = "SOCP", 1 = "QCQP" (gurobi_utils.py:139-144) */
    int32_t n_chains;
    const int32_t* chain_len;   /* poses per chain                             */
    int32_t n_landmarks;
    int64_t n_rel;              /* relative-pose measurements: odometry, then loop closures */
    const int32_t* rel_base;    /* pose index of base_pose / to_pose           */
    const int32_t* rel_to;
    const double*  rel_t;       /* n_rel * dim   translation_vector            */
    const double*  rel_R;       /* n_rel * dim * dim  rotation_matrix, row-major */
    const double*  rel_kappa;   /* translation_precision                       */
    const double*  rel_tau;     /* rotation_precision                          */
    int64_t n_rng;              /* range measurements                          */
    const int32_t* rng_a;       /* variable id of first_key / second_key       */
    const int32_t* rng_b;
    const double*  rng_dist;
    const double*  rng_prec;    /* precision = 1 / stddev^2                    */
    int64_t n_lprior;           /* landmark priors (gurobi_utils.py:433-446)   */
    const int32_t* lprior_lm;   /* landmark index                              */
    const double*  lprior_t;    /* n_lprior * dim                              */
    const double*  lprior_prec;
} score_graph;

typedef struct score_assembled score_assembled;

/* Build the program (host memory owned by *out).  Inputs are borrowed for the call. */
int  score_assemble(const score_graph* g, score_assembled** out);
/* The same for `count` graphs in one call, one graph per host thread of the library's team (the Monte-Carlo path:
 * score_amd.solve_score.solve_score_batch builds the models of a lock-step group this way -- one foreign call
 * per group instead of one per graph keeps the callers' interpreter lock out of the picture).  out[0..count-1];
 * on failure nothing is left allocated and the message names the first graph that failed.   */
int  score_assemble_batch(const score_graph* graphs, int32_t count, score_assembled** out);
/* Fill `view` with pointers into the assembled program (valid until score_assembled_free);
 * pass it to score_create / score_create_batch like any other score_problem.   */
int  score_assembled_view(const score_assembled* a, score_problem* view);
void score_assembled_free(score_assembled* a);

/* ---------------------------------------------------------------------------
 * Synthetic Manhattan-world RA-SLAM graphs, generated where they are solved (the batched generator).
 * The reference's simulation study draws such worlds one at a time in Python (its shipped fixture
 * examples/manhattan/factor_graph.pickle is one of them; statistics measured in SURVEY 8(d) and restated in
 * score_amd/csrc/score_generate.hpp: lattice walks with 81 / 9 / 9 / 1 % straight / left / right / back, odometry and range
 * noise, every robot-beacon and same-timestep robot-robot pair measured with probability p_range).  score_generate_manhattan
 * makes `count` worlds on the device -- trial t is the world of seed + t whatever the batch (counter-based Philox4x32-10) -- and
 * hands back their flat arrays: score_generated_graph fills a score_graph view (valid until score_generated_free) that
 * score_create_from_graphs / score_assemble / score_refine_create take like any other; score_generated_truth the ground
 * truth (2-D: poses n_robots * n_poses x (x, y, theta), beacons n_beacons x (x, y); 3-D: poses x (x, y, z, R row-major --
 * 12 values), beacons x (x, y, z); either may be NULL).
 * ------------------------------------------------------------------------- */
typedef struct score_manhattan_spec {
    int32_t  n_robots;      /* 1..64; robot 0's first pose is the pinned one (origin, identity heading) */
    int32_t  n_poses;       /* poses per robot (>= 2)                       */
    int32_t  n_beacons;
    int32_t  side;          /* grid [0, side]^2                             */
    double   p_range;       /* probability of a range measurement per pair and timestep */
    double   sigma_t;       /* odometry translation noise (precision 1 / sigma^2) */
    double   sigma_theta;   /* odometry rotation noise                      */
    double   sigma_range;   /* range noise; measurements clamped at >= 0    */
    uint64_t seed;          /* world t of the call: seed + t                */
    int32_t  dim;           /* 2 (0 = 2): the shipped fixture's worlds; 3: walks on the lattice of the cube [0, side]^3 with
                               axis-aligned orientations (the model is dimension-generic, gurobi_utils.py:37-50; the reference
                               ships no 3-D data: statistics carried over, csrc/score_generate.hpp)                         */
    int32_t  reserved;
} score_manhattan_spec;
typedef struct score_generated score_generated;
int  score_generate_manhattan(const score_manhattan_spec* spec, int32_t count, int32_t device, score_generated** out);
int  score_generated_graph(const score_generated* g, int32_t index, struct score_graph* view);
int  score_generated_truth(const score_generated* g, int32_t index, double* poses, double* beacons);
void score_generated_free(score_generated* g);
/* A handle for worlds first .. first + count - 1 of a generated batch, as score_create_from_graphs builds it for their views --
 * but when the batch lives on the handle's device the measurement arrays are read where the generator left them: nothing of a
 * world crosses the link again (the host lays out sizes, cones and chains from its copy).  relaxation: 0 = "SOCP", 1 = "QCQP". */
int  score_create_from_generated(const score_generated* g, int32_t first, int32_t count, int32_t relaxation,
                                 const score_settings* s, score_handle** out);

/* SO(d) rounding of the relaxed rotation blocks: replaces the per-pose
 * round_to_special_orthogonal(...) calls of VariableCollection.get_variable_values
 * (score/utils/gurobi_utils.py:115-125; score/utils/matrix_utils.py:59-79).
 * `blocks`, `rotations`: n row-major dim x dim matrices (host memory); rotations[i] is the
 * maximiser of tr(R' blocks[i]) over SO(dim) -- what the reference's SVD + determinant fix
 * computes.  degenerate[i] = 1 where that maximiser is not unique (rank-deficient or
 * reflection-like input, NaN): rotations[i] is then the identity and the caller decides
 * (score_amd falls back to the reference's SVD formula for those blocks).  dim = 2 or 3. */
int  score_round_to_so(int32_t dim, int64_t n, const double* blocks, double* rotations,
                       int32_t* degenerate, int32_t device);

/* Linear mode: the handle's chain-preconditioned PCG as a solver for symmetric positive definite
 * systems K x = rhs -- the sparse solves inside the local refinement that follows SCORE (the reference's
 * README.md:63-67 hands the SCORE estimate to GTSAM; score_amd/refine.py runs Gauss-Newton /
 * Levenberg-Marquardt on SE(2) and solves its damped normal equations here).
 * score_linear_create: `pattern` carries n, the CSR pattern of K in P_rowptr / P_col (columns strictly
 * increasing, every row holds its diagonal), the chain hint (one chain per robot, one node per pose,
 * block_size = 3 for SE(2)), and m = 0; values are not read.  score_linear_solve: `values` on that
 * pattern; K is factored along the chains on the device (k_factor), PCG runs until
 * r'M^-1 r <= rel_tol^2 r0'M^-1 r0 (tested on the device) or max_iters.  Returns 0 converged, 1 iteration
 * cap, < 0 error; rel_residual (optional) = |rhs - K x|_2 / |rhs|_2.  Free with score_destroy.      */
int  score_linear_create(const score_problem* pattern, const score_settings* s, score_handle** out);
int  score_linear_solve(score_handle* h, const double* values, const double* rhs, double* x, double rel_tol,
                        int32_t max_iters, int32_t* iters_used, double* rel_residual);

/* Local refinement after SCORE, whole loop behind the ABI: Gauss-Newton / Levenberg-Marquardt on SE(2) or SE(3)
 * from a given estimate (README.md:63-67 of the reference: "SCORE's estimate initialises a local solver").
 * The graph is the score_graph of score_assemble (dim = 2 or 3; `relaxation` is not read).  Per-measurement
 * Jacobian blocks, J'J / J'r on a fixed pattern and trial points are device kernels; the damped normal
 * equations run in linear mode (above).
 *   dim 2: poses n_poses x (theta, x, y) in chain order; landmarks n_landmarks x (x, y);
 *   dim 3: poses n_poses x 12 = [R (3 x 3, row-major) | t]; landmarks n_landmarks x 3; steps live in the tangent
 *          space (R <- R Exp(omega), t <- t + v), the chains of the preconditioner are the omega blocks and the
 *          v blocks of every robot (3 x 3).
 * Pose 0 stays fixed.  tol: stop when |J'r|_inf <= tol * max(1, cost).                                        */
typedef struct score_refine score_refine;
typedef struct score_refine_info {
    double  cost_initial, cost_final, grad_inf;
    int32_t iterations, linear_solves, pcg_iters;
    double  setup_ms, solve_ms;
} score_refine_info;
int  score_refine_create(const score_graph* g, const score_settings* s, score_refine** out);
int  score_refine_run(score_refine* r, const double* poses_in, const double* landmarks_in, int32_t max_iters,
                      double tol, double* poses_out, double* landmarks_out, score_refine_info* info);
void score_refine_destroy(score_refine* r);

const char* score_last_error(void);
const char* score_backend(void);   /* "hip-gfx950" or "cpu-twin"             */

/* Layout version of the structs above.  A binding built against another header (an out-of-tree ctypes
 * struct, say) would be misread silently -- score_create_batch walks an ARRAY of score_problem, so a
 * stale stride goes wrong from the second problem on.  Bump SCORE_ABI_VERSION whenever a struct changes;
 * loaders compare (score_amd.solver.load_library does).  History: 1 = rounds 1-2, 2 = score_problem
 * gained rep_d / rep_n, 3 = this function, 4 = score_assemble_batch, 5 = score_create_from_graphs.                                                        */
#define SCORE_ABI_VERSION 7
int32_t score_abi_version(void);   /* SCORE_ABI_VERSION of the library's build, times 1000, plus sizeof(score_problem) */

#ifdef __cplusplus
}
#endif
#endif /* SCORE_HIP_H */
