/* minilp_hip.h — C ABI of the MI355X-native simplex pivot engine (libminilp_hip.so).
 *
 * This is the drop-in boundary for ztlpn/minilp's Problem / Solution API.  The reference has
 * no FFI seam (pure Rust); these are the entry points a `minilp-sys` Rust shim
 * binds.  Each function cites the reference interface it replaces.
 *
 * Conventions
 *   - opaque handles, plain pointers and sizes, no C++/torch types;
 *   - status codes: 0 OK, 1 Infeasible, 2 Unbounded (lib.rs:172-178 `Error`), <0 internal
 *     (-1 invalid argument / reference panic condition, -2 singular basis, -3 HIP error,
 *      -4 no GPU / extension unavailable, -5 the dense nucleus inverse (8 k^2 bytes) does not fit in HBM).
 *     mlp_last_error() returns the message;
 *   - not thread-safe per handle; distinct handles are independent;
 *   - Solution mutators follow the reference's consume-on-error rule (lib.rs:359, 385): on a
 *     non-zero status the solution is freed and *s is set to NULL;
 *   - all solver state (x_B, d, gamma, beta, the basis inverse, A in CSR+CSC) lives in HBM;
 *     only scalars cross the boundary per pivot.
 */
#ifndef MINILP_HIP_H
#define MINILP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mlp_problem mlp_problem;   /* lib.rs:193-200 `Problem`  */
typedef struct mlp_solution mlp_solution; /* lib.rs:313-318 `Solution` (owns the device-resident Solver) */

enum { MLP_MINIMIZE = 0, MLP_MAXIMIZE = 1 };      /* lib.rs:61-68  OptimizationDirection */
enum { MLP_EQ = 0, MLP_LE = 1, MLP_GE = 2 };      /* lib.rs:160-169 ComparisonOp */
enum { MLP_OK = 0, MLP_INFEASIBLE = 1, MLP_UNBOUNDED = 2,
       MLP_EINVAL = -1, MLP_ESINGULAR = -2, MLP_EHIP = -3, MLP_ENOGPU = -4, MLP_ENOMEM = -5 };

/* ABI version of this header: bumped whenever a struct below changes.  From version 4 on mlp_stats only GROWS AT ITS END
 * (fields are appended, never inserted or removed), so a host built against an older version-4+ header reads a valid
 * prefix; mlp_stats_size() is sizeof(mlp_stats) as the LIBRARY was built — a host checks it (and mlp_abi_version())
 * against its own header before trusting the layout. */
#define MLP_ABI_VERSION 4u
uint32_t mlp_abi_version(void);
uint64_t mlp_stats_size(void);

const char* mlp_last_error(void);
/* Number of visible HIP devices (0 => every solve returns MLP_ENOGPU; there is no CPU fallback). */
int mlp_device_count(void);
/* Select the HIP device used by solutions created afterwards on this thread (one process per GPU). */
int mlp_set_device(int device);

/* ---- Problem (lib.rs:215-305) ------------------------------------------------------------ */
mlp_problem* mlp_problem_new(int direction);                                   /* Problem::new      lib.rs:217 */
mlp_problem* mlp_problem_clone(const mlp_problem* p);                          /* #[derive(Clone)]  lib.rs:193 */
void mlp_problem_free(mlp_problem* p);
/* returns the Variable index (lib.rs:72, 79) */
uint32_t mlp_problem_add_var(mlp_problem* p, double obj_coeff, double min, double max); /* add_var lib.rs:233 */
uint32_t mlp_problem_num_vars(const mlp_problem* p);
/* duplicate or out-of-range variable => MLP_EINVAL (the reference panics, lib.rs:247-249) */
int mlp_problem_add_constraint(mlp_problem* p, const uint32_t* vars, const double* coeffs, uint64_t k,
                               int cmp_op, double rhs);                        /* add_constraint lib.rs:276 */
/* bulk forms of add_var / add_constraint (same semantics, one call): n variables; m rows in CSR */
int mlp_problem_add_vars(mlp_problem* p, uint64_t n, const double* obj_coeffs, const double* mins, const double* maxs);
int mlp_problem_add_constraints_csr(mlp_problem* p, uint64_t m, const uint64_t* indptr, const uint32_t* vars,
                                    const double* coeffs, const int32_t* cmp_ops, const double* rhs);
int mlp_problem_solve(const mlp_problem* p, mlp_solution** out);               /* solve lib.rs:291 */
/* read-back of the model data (Problem is plain data in the reference too, lib.rs:194-200) */
uint64_t mlp_problem_num_constraints(const mlp_problem* p);
int mlp_problem_var(const mlp_problem* p, uint32_t var, double* obj_coeff, double* min, double* max);
/* returns the number of terms; copies min(terms, cap) of them (sorted by variable) */
uint64_t mlp_problem_constraint(const mlp_problem* p, uint64_t c, uint32_t* vars, double* coeffs, uint64_t cap,
                                int* cmp_op, double* rhs);

/* ---- Solution (lib.rs:332-424) ----------------------------------------------------------- */
mlp_solution* mlp_solution_clone(const mlp_solution* s);  /* #[derive(Clone)] lib.rs:313: deep copy of device state */
void mlp_solution_free(mlp_solution* s);
double mlp_solution_objective(const mlp_solution* s);                          /* objective lib.rs:334 */
uint32_t mlp_solution_num_vars(const mlp_solution* s);
int mlp_solution_var_value(const mlp_solution* s, uint32_t var, double* out);  /* var_value lib.rs:344 / Index lib.rs:426 */
int mlp_solution_values(const mlp_solution* s, double* out, uint32_t n);       /* iter lib.rs:350 (bulk form) */
int mlp_solution_add_constraint(mlp_solution** s, const uint32_t* vars, const double* coeffs, uint64_t k,
                                int cmp_op, double rhs);                       /* add_constraint lib.rs:368 */
int mlp_solution_fix_var(mlp_solution** s, uint32_t var, double val);          /* fix_var lib.rs:390 */
int mlp_solution_unfix_var(mlp_solution** s, uint32_t var, int* was_fixed);    /* unfix_var lib.rs:399 */
int mlp_solution_add_gomory_cut(mlp_solution** s, uint32_t var);               /* add_gomory_cut lib.rs:419 */

/* ---- Engine-level controls (no counterpart in the reference: its pivot loop exposes no
 *      counter; these implement the fixed-pivot-budget measurement of §8d) ---- */
/* Like mlp_problem_solve but stops after `budget` simplex iterations (budget < 0: run to optimality).
 * flags: bit0 = record a pivot trace, bit1 = time the dominant kernels with HIP events (sampled iterations),
 * bit2 = with bit1: every iteration is a sampled one. */
int mlp_problem_solve_ex(const mlp_problem* p, mlp_solution** out, int64_t budget, uint32_t flags);
int mlp_solution_continue(mlp_solution* s, int64_t budget);
/* Basis checkpoint ("pivots from a saved mid-solve basis"; the reference has no basis I/O — its
 * Solver state is solver.rs:14-58).  mlp_solution_save_basis returns the size of the blob and copies it when
 * buf != NULL and cap is large enough (0 on error).  mode 0: basic / non-basic sets (solver.rs:37, 44), non-basic
 * flags and values (solver.rs:45-49); mode 1: + the steepest-edge weights as f32 (they only steer pricing);
 * mode 2: + x_B, d, gamma, beta and the objective as f64 — a solve loaded from a mode-2 blob continues pivot for
 * pivot like the uninterrupted one.  mlp_problem_solve_from_basis builds the same problem (Solver::try_new),
 * installs the basis, re-inverts the nucleus on the device (BasisSolver::reset, solver.rs:1286-1303), recomputes
 * x_B and the reduced costs from the basis (modes 0/1; solver.rs:1177-1231) and continues like mlp_problem_solve_ex.
 * A blob of another model (row / variable counts differ, sets do not partition the variables) => MLP_EINVAL.
 * A sharded solution (mlp_solution_enable_sharding) saves mode 0 only: the partition is what every rank holds in full. */
uint64_t mlp_solution_save_basis(const mlp_solution* s, int mode, void* buf, uint64_t cap);
int mlp_problem_solve_from_basis(const mlp_problem* p, const void* blob, uint64_t len, mlp_solution** out, int64_t budget,
                                 uint32_t flags);
/* flags bit1 (HIP-event timing): 1 = sample EVERY iteration as an eager, event-bracketed one (measurement passes),
 * 0 = the default cadence (every 8th / 4th batch), < 0 = switch the sampling off (until the next call with a value >= 0, which
 * also switches HIP-event timing on for a solution created without flag bit1) */
int mlp_solution_set_sampling(mlp_solution* s, int every_iteration);
int mlp_solution_budget_exhausted(const mlp_solution* s);
/* Recompute the dense nucleus inverse from A (the counterpart of BasisSolver::reset,
 * solver.rs:1286-1303); returns max |W_incremental - W_fresh| through *max_diff when non-NULL.  While the basis is held as the
 * compact factor (mlp_stats.factor_active) there is no incremental inverse to compare with: the factor is rebuilt from the basis
 * and *max_diff is NaN ("nothing compared"), never a false 0. */
int mlp_solution_reinvert(mlp_solution* s, double* max_diff);

/* The objective value and the reduced costs are recomputed for the new point as well (solver.rs:1199-1231).
 * x_B = B^-1 (b - N x_N) recomputed from the basis, with two steps of iterative refinement (the reference's
 * recalc_basic_var_vals, solver.rs:1177-1197, which it leaves unused; here it is the polish step of long runs and is
 * exported for hosts that want it after many warm-start pivots).  A dense-rhs FTRAN: with a large nucleus it is one
 * streaming read of the nucleus inverse (8 k^2 bytes) per step; in profile mode mlp_stats.dense_ftran_* time it. */
int mlp_solution_recompute_basic_values(mlp_solution* s);

/* Column-block sharding of the pricing path across the GPUs of one node (one process per GPU,
 * DESIGN.md §6).  Every rank builds the SAME problem, calls mlp_problem_solve_ex(budget = 0), then
 * this function with its rank, the world size (<= 16) and the name of a POSIX shared-memory object of
 * 896 * world zeroed bytes created by the launcher (the rendezvous), and then the same sequence of
 * mlp_solution_continue calls.  Rank r owns non-basic positions [n*r/world, n*(r+1)/world): its
 * tableau-row sweep, d/gamma update and pricing scan cover only that block; candidates are exchanged once or
 * twice per pivot (primal: pricing all-gather + ratio decision; dual: leaving row, pass-1 minimum, pass-2
 * candidate) from inside the pivot kernels.  Transport: every rank keeps a mailbox in its own HBM, the peers map
 * it through HIP IPC (the handles travel through the rendezvous object) and write their 64-byte records into it
 * over xGMI; polling is local.  MLP_MAILBOX=host selects the older host-memory mailbox (PCIe) instead.  The call
 * returns when every rank has mapped every mailbox (bounded waits: a missing rank is an error, MLP_EHIP).
 * Primal and dual loops; the Solution mutators are refused.  mlp_solution_transport names the transport in use.
 * Deferred sharding (default; MLP_SHARD_DEFER=0 turns it off): while the nucleus is small — the sparse-tableau-row regime, a few
 * hundred pivots from the slack basis, where a pivot is tens of microseconds of latency-bound launches and per-pivot exchanges can
 * only slow it down — the ranks run as bit-identical REPLICAS (the deterministic unsharded iteration on every rank, no exchange);
 * the column blocks and the exchanges go live at the first batch that leaves that regime, at the same pivot on every rank, and
 * stay live.  Nothing changes for the caller: the same calls, the same pivots. */
int mlp_solution_enable_sharding(mlp_solution* s, int rank, int world, const char* shm_name);
/* The same with the transport named by the caller: NULL / "" = the default above (or MLP_TRANSPORT), "peer", "host",
 * "rccl" — north_star's literal transport: the mailbox records of every exchange (16-byte pricing candidates, the ratio decision,
 *          the dual loop's minimum / candidate) are delivered by ncclAllGather over xGMI: the pivot kernels post into and poll their
 *          own device box, and while a batch of pivots is in flight the host pumps stage -> ncclAllGather -> deliver rounds on a
 *          second stream until every rank's batch has drained.  rccl_id = the 128-byte ncclUniqueId made by rank 0
 *          (mlp_rccl_unique_id) and distributed by the launcher; librccl.so is loaded at run time.  Slower per exchange than
 *          the peer stores (a collective per round), so it is the fallback for nodes where the peer mappings do not deliver;
 * "pump" — the rccl transport's protocol with peer copies between IPC-mapped staging buffers in place of the collective (RCCL
 *          refuses two ranks on one device; this is how the protocol is tested on a one-GPU box).
 * The pump transports accept world = 1. */
int mlp_solution_enable_sharding_ex(mlp_solution* s, int rank, int world, const char* shm_name, const char* transport,
                                    const void* rccl_id);
int mlp_rccl_unique_id(void* out128);
const char* mlp_solution_transport(const mlp_solution* s);

typedef struct mlp_stats {
    uint64_t iterations, basis_changes, bound_flips, primal_iters, dual_iters, reinversions;
    uint64_t num_constraints, num_total_vars, nucleus_size, nucleus_capacity, nnz;
    /* algorithmic bytes and HIP-event time of the two dominant kernels */
    double fused_bytes, fused_ms, sweep_bytes, sweep_ms;
    uint64_t fused_launches, sweep_launches;
    double solve_wall_s; /* host wall time spent inside the pivot loops */
    uint64_t kase[5]; /* basis changes by partition case: nucleus->nucleus, singleton->nucleus (grow), nucleus->singleton
                         (shrink), singleton->singleton (column swap), same-row singleton swap */
    double update_ms; uint64_t update_launches; /* K8 (x_B/d/gamma/beta update + next pricing scan), HIP-event time */
    uint64_t banded_sweep;    /* 1 when the tableau-row pass runs as the banded sweep (large m), 0 for the CSC pull */
    uint64_t final_refreshes; /* times optimality was re-examined on recomputed reduced costs (long runs only) */
    double max_pivot_err; /* drift monitor: max |alpha_q[r] - alpha_r[q]| / max(1,|alpha_q[r]|) seen so far */
    /* FTRAN of the entering column (head + gather of the listed columns of the nucleus inverse + F push):
     * algorithmic bytes 8 k |list| + 12 nnz(nucleus columns) + 12 nnz(a_q), HIP-event time of sampled iterations */
    double ftran_bytes, ftran_ms; uint64_t ftran_launches;
    double iter_ms; uint64_t iter_samples; /* whole sampled iterations, first kernel to last (HIP events) */
    uint64_t beta_rebuilds; /* lazy dual steepest edge: exact rebuilds of the dual edge norms from the basis inverse (the primal
                               loop skips their per-pivot recurrence, solver.rs:1153-1174, because nothing reads them there) */
    double fold_bytes, fold_ms; uint64_t fold_launches; /* sampled folds of the pending rank-1 terms into the nucleus inverse (HIP events) */
    double dense_ftran_bytes, dense_ftran_ms; uint64_t dense_ftran_launches; /* dense-rhs FTRAN of mlp_solution_recompute_basic_values /
                              the polish step: algorithmic bytes (8 k^2 per solve) and kernel-exact time of the pass over the nucleus inverse */
    double str_ms; uint64_t str_launches; /* sampled sparse tableau rows (small nucleus: only the columns that meet supp(rho)): HIP-event time */
    uint64_t hyper_iters, hyper_bails; /* iterations run by the hypersparse single-workgroup kernel (support-restricted work, sparse models);
                                          iterations it declined (list overflow / too much work for one workgroup) and handed to the multi-kernel path */
    uint64_t ratio_stalls; /* in-kernel waits of the one-launch Harris tests that timed out (grid not co-resident); each one is
                              retried with the two-launch form, which then stays selected */
    /* ---- appended in ABI version 4 ---- */
    uint64_t reinversion_fallbacks; /* rounds 1-3: re-inversions a library call reported singular and the Gauss-Jordan kernels then
                                       re-examined; always 0 since the blocked inversion is hand-written (kept for the layout);
                                       ratio_stalls also counts a stalled wait of the one-launch small-nucleus form */
    /* compact factor of the basis (a peeled triangular factor + additive eta terms instead of the explicit
     * nucleus inverse; selected by the measured shape of the basis, MLP_FACTOR=1 / 0 forces it on / off) */
    uint64_t factor_active;     /* 1 while B^-1 is held as the compact factor */
    uint64_t factor_refactors;  /* refactorisations (peels of the current basis) so far */
    uint64_t factor_levels;     /* levels of the last peel = dependent steps of one triangular solve */
    uint64_t factor_switches;   /* switches between the two representations */
    uint64_t factor_bump;       /* columns the last peel left over (cycles of the basis graph; their inverse is kept explicitly) */
    uint64_t factor_bump_max;   /* largest bump of any refactorisation so far */
} mlp_stats;
void mlp_solution_stats(const mlp_solution* s, mlp_stats* out);
void mlp_solution_reset_stats(mlp_solution* s);

/* pivot trace (flags bit0): phase 0 primal / 1 dual; row = -1 for a bound flip */
uint64_t mlp_solution_trace_len(const mlp_solution* s);
void mlp_solution_trace_get(const mlp_solution* s, uint64_t i, int32_t* phase, int64_t* col, int64_t* row,
                            int64_t* entering_var, int64_t* leaving_var, double* pivot_coeff, double* obj_after);
/* white-box state for the differential tests (names follow solver.rs:14-58): returns the length,
 * copies min(len, cap) doubles into out when out != NULL; (uint64_t)-1 for an unknown name. */
uint64_t mlp_solution_state(const mlp_solution* s, const char* what, double* out, uint64_t cap);

/* ---- MPS (mps.rs:39 MpsFile::parse) ------------------------------------------------------ */
typedef struct mlp_mps mlp_mps;
int mlp_mps_parse(const char* text, uint64_t len, int direction, mlp_mps** out);
void mlp_mps_free(mlp_mps* f);
const char* mlp_mps_name(const mlp_mps* f);                 /* MpsFile::problem_name mps.rs:11 */
uint32_t mlp_mps_num_vars(const mlp_mps* f);
const char* mlp_mps_var_name(const mlp_mps* f, uint32_t i); /* MpsFile::variables mps.rs:13 */
int64_t mlp_mps_var_index(const mlp_mps* f, const char* name);
mlp_problem* mlp_mps_problem(const mlp_mps* f);             /* MpsFile::problem mps.rs:15 (a clone) */

/* ---- engine-level stepping (second table) --------------------------------------------
 * The iteration of Solver::optimize / restore_feasibility (solver.rs:487-547) one stage per call, paced
 * by the host: what a host-side Solver would bind instead of BasisSolver::{solve, solve_transp,
 * push_eta_matrix} (solver.rs:1273-1339) and the scans of choose_pivot / pivot.  Every stage runs the
 * same kernels as the replayed graph; vectors stay on the device (read them with mlp_solution_state:
 * "col_coeffs" after FTRAN, "row_coeffs" after ROW, ...); only the scalars below cross the boundary.
 *
 *   mlp_problem_solve_ex(p, &s, 0, 0);                         // set up, no pivot yet
 *   while (mlp_engine_open(s, &info) == MLP_ITER_PIVOT)        // pricing: entering column (primal) / leaving row (dual)
 *       do st = mlp_engine_stage(s, info.next_stage, &info);   // FTRAN, RATIO, BTRAN, BASIS, ROW, APPLY in the phase's order
 *       while (st == MLP_ITER_PIVOT || st == MLP_ITER_FLIP);   // after APPLY the next iteration is already priced
 *
 * mlp_engine_open picks the phase initial_solve would run next (dual loop while primal-infeasible,
 * then recalc_obj_coeffs + primal loop) and returns the status of the pricing decision; a terminal
 * status (OPTIMAL, FEASIBLE, INFEASIBLE, UNBOUNDED) closes the loop, after FEASIBLE call open again.
 * Return values < 0 are errors (mlp_last_error), e.g. a stage called out of order. */
enum { MLP_STAGE_FTRAN = 0, MLP_STAGE_RATIO = 1, MLP_STAGE_BTRAN = 2, MLP_STAGE_BASIS = 3, MLP_STAGE_ROW = 4, MLP_STAGE_APPLY = 5 };
enum { MLP_ITER_PIVOT = 0, MLP_ITER_FLIP = 1, MLP_ITER_OPTIMAL = 2, MLP_ITER_UNBOUNDED = 3, MLP_ITER_FEASIBLE = 4,
       MLP_ITER_INFEASIBLE = 5, MLP_ITER_SINGULAR = 6 };
typedef struct mlp_iter_info {
    int32_t status;      /* MLP_ITER_* of the open iteration (after APPLY: of the next one) */
    int32_t phase;       /* 0 primal (optimize), 1 dual (restore_feasibility) */
    int32_t next_stage;  /* the stage mlp_engine_stage expects next, -1 when no iteration is open */
    int32_t reserved;
    int64_t col, row;    /* entering non-basic position q, leaving basic position r (-1 while undecided) */
    int64_t entering_var, leaving_var;
    double pivot_coeff;  /* alpha_rq (solver.rs:1073) */
    double step;         /* change of the entering variable (solver.rs:828) */
    double objective;    /* cur_obj_val after the decision (solver.rs:1027) */
    uint64_t nucleus_size;
} mlp_iter_info;
int mlp_engine_open(mlp_solution* s, mlp_iter_info* out);
int mlp_engine_stage(mlp_solution* s, int stage, mlp_iter_info* out);

/* ---- driver helper (host only; examples/tsp.rs:437-539) -------------------------------------------
 * Stoer-Wagner global minimum cut of a dense symmetric n x n weight matrix (row-major, zero diagonal):
 * returns the cut weight and marks one side of the cut in side_out[n] (0/1).  Used by the TSP
 * cutting-plane driver to separate subtour-elimination constraints. */
double mlp_util_min_cut(uint32_t n, const double* weights, uint8_t* side_out);

#ifdef __cplusplus
}
#endif
#endif /* MINILP_HIP_H */
