// engine.h — host orchestration of the device-resident revised simplex (the "Solver" of
// solver.rs:14-58 with every vector in HBM and every hot loop a HIP kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"

namespace mlp {

enum Stage { STAGE_FTRAN = 0, STAGE_RATIO = 1, STAGE_BTRAN = 2, STAGE_BASIS = 3, STAGE_ROW = 4, STAGE_APPLY = 5 };

struct MlpError : std::runtime_error {
    int code;
    MlpError(int c, const std::string& s) : std::runtime_error(s), code(c) {}
};
struct LpFail {  // lib.rs:172-178
    int code;    // 1 infeasible, 2 unbounded
};

#define HIPCHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess)                                                                           \
            throw ::mlp::MlpError(-3, std::string(#expr) + " failed: " + hipGetErrorString(e__) + " at " + \
                                          __FILE__ + ":" + std::to_string(__LINE__));                    \
    } while (0)

struct Constraint {  // lib.rs:199: (CsVec, ComparisonOp, f64) with sorted unique indices
    std::vector<int> idx;
    std::vector<double> val;
    int op;
    double rhs;
};
struct ProblemData {  // lib.rs:193-200
    int direction = 0;
    std::vector<double> obj, lo, hi;  // obj already negated for Maximize (lib.rs:235-238)
    std::vector<Constraint> cons;
    int add_var(double c, double mn, double mx);
    void add_constraint(const uint32_t* vars, const double* coeffs, uint64_t k, int op, double rhs);
};

// Process-wide cache of small device blocks.  A Solution owns ~50 device arrays; `Solution::clone`
// (one per branch-and-bound node in the TSP driver, lib.rs:313) and `drop` would otherwise pay one
// hipMalloc / hipFree each (hipFree synchronises the device).  Blocks are rounded up to a power of two
// and recycled; large blocks (W) go straight to the runtime.  A block is returned only after the
// owning stream has been synchronised, so the next owner may use it on any stream.
struct DevPool {
    static constexpr size_t kMaxBlock = (size_t)64 << 20;   // larger blocks are not cached
    static constexpr size_t kMaxCached = (size_t)4 << 30;   // total bytes kept in the cache
    static void* get(size_t bytes, size_t* got_bytes, int* device);   // on the current device
    static void* get_raw(size_t bytes, size_t* got_bytes, int* device);
    static void put(void* p, size_t bytes, int device);                // back to the free list of ITS device
    static void trim();  // hipFree everything cached
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;        // elements
    size_t bytes = 0;      // size of the block as obtained from the pool
    int dev = 0;           // device the block lives on
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) DevPool::put(p, bytes, dev);
        p = nullptr;
        cap = 0;
        bytes = 0;
    }
    // grow to hold n elements, preserving the first `keep` ones
    void ensure(size_t n, size_t keep, hipStream_t st) {
        if (n <= cap) return;
        size_t want = n + n / 2 + 64, got = 0;
        int ndev = 0;
        T* np = static_cast<T*>(DevPool::get(want * sizeof(T), &got, &ndev));
        if (p && keep) HIPCHECK(hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (p) {
            HIPCHECK(hipStreamSynchronize(st));
            DevPool::put(p, bytes, dev);
        }
        p = np;
        dev = ndev;
        bytes = got;
        cap = got / sizeof(T);
    }
    void alloc_exact(size_t n) {  // fresh allocation of (at least) n elements (contents undefined)
        release();
        size_t got = 0;
        p = static_cast<T*>(DevPool::get(n * sizeof(T), &got, &dev));
        bytes = got;
        cap = n;
    }
    void upload(const std::vector<T>& h, hipStream_t st) {
        ensure(h.size(), 0, st);
        if (!h.empty()) HIPCHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
    }
    void swap(DevBuf<T>& o) {
        std::swap(p, o.p); std::swap(cap, o.cap); std::swap(bytes, o.bytes); std::swap(dev, o.dev);
    }
    void copy_from(const DevBuf<T>& o, size_t n, hipStream_t st) {
        ensure(n, 0, st);
        if (n) HIPCHECK(hipMemcpyAsync(p, o.p, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    }
};

struct PivotRecord {
    int32_t phase;
    int64_t col, row, entering_var, leaving_var;
    double pivot_coeff, obj_after;
};

struct Stats {
    uint64_t iterations = 0, basis_changes = 0, bound_flips = 0, primal_iters = 0, dual_iters = 0, reinversions = 0;
    double fused_bytes = 0, fused_ms = 0, sweep_bytes = 0, sweep_ms = 0;
    uint64_t fused_launches = 0, sweep_launches = 0, final_refreshes = 0;
    // FTRAN of the entering column (head + gather + F push): HIP-event time and algorithmic bytes
    double ftran_bytes = 0, ftran_ms = 0;
    uint64_t ftran_launches = 0;
    double iter_ms = 0;  // whole sampled iterations (first kernel to last), HIP events
    uint64_t iter_samples = 0;
    double update_ms = 0;
    uint64_t update_launches = 0;
    double solve_wall_s = 0;
    double max_pivot_err = 0;
    uint64_t reinversion_fallbacks = 0;  // (rounds 1-3: rocSOLVER inversions that reported a false zero pivot and were redone; always 0 since the blocked inversion is hand-written)
    double str_ms = 0;  // sampled sparse tableau rows (k_row_touch + k_row_pull, launch-bracketed HIP events)
    uint64_t str_launches = 0;
    uint64_t hyper_bail_reason[10] = {};  // by reason code of the kernel (hyper.inc)
    uint64_t hyper_iters = 0, hyper_bails = 0;  // iterations taken by the hypersparse kernel; iterations it handed back
    uint64_t ratio_stalls = 0;   // in-kernel waits of the fused ratio test that timed out (each one retried with two launches)
    uint64_t beta_rebuilds = 0;  // lazy dual steepest edge: exact rebuilds of beta from the basis inverse
    uint64_t fac_refactors = 0, fac_levels = 0, fac_switches = 0, fac_bump = 0, fac_bump_max = 0, fac_sb_factors = 0, fac_sb_fallbacks = 0, fac_sb_rounds = 0, fac_sb_tail = 0, fac_sb_skipped = 0;  // compact factor: refactorisations (peels), levels of the last one, mode switches
    // dense-rhs FTRAN x_B = B^-1 (b - N x_N) (recalc_basic_vals): the streaming read of the nucleus inverse, kernel-exact
    double dense_ftran_bytes = 0, dense_ftran_ms = 0;
    uint64_t dense_ftran_launches = 0;
    double fold_bytes = 0, fold_ms = 0;  // sampled folds of W0 (HIP events stamped by the kernel): read + write of the matrix
    uint64_t fold_launches = 0;
    uint64_t kase[5] = {0, 0, 0, 0, 0};  // partition-change cases: nuc->nuc, grow, shrink, col swap, same-row
};

class Engine {
public:
    Engine();
    ~Engine();
    Engine(const Engine&) = delete;

    // Solver::try_new (solver.rs:108-369): throws LpFail{1} for min > max / contradictory empty rows
    void try_new(const ProblemData& pd);
    void initial_solve();                                                   // solver.rs:470-485
    void add_constraint(Constraint c);                                      // solver.rs:549-634 (indices < num_total_vars)
    void fix_var(int var, double val);                                      // solver.rs:378-415
    bool unfix_var(int var);                                                // solver.rs:418-438
    void add_gomory_cut(int var);                                           // solver.rs:440-460
    double get_value(int var);                                              // solver.rs:371-376
    void get_values(double* out, int n);
    double cur_obj_val();                                                   // solver.rs:51
    Engine* clone();                                                        // #[derive(Clone)] solver.rs:14
    double reinvert(bool replace);  // from-scratch nucleus inversion; returns max |W - W_fresh|
    double last_reinvert_scale = 0.0;  // max |W_fresh| of that comparison (state "reinvert_scale")
    // Basis checkpoint (include/minilp_hip.h: mlp_solution_save_basis / mlp_problem_solve_from_basis).
    // mode 0: basic / non-basic sets, flags and x_N; 1: + steepest-edge weights as f32; 2: + x_B, d, gamma,
    // beta, objective as f64 (a loaded solve continues pivot for pivot).
    std::vector<uint8_t> save_basis(int mode);
    void load_basis(const uint8_t* blob, size_t len);  // call after try_new on the same problem
    // column-block pricing across ranks.  transport: nullptr / "" = MLP_TRANSPORT or the default (device mailboxes written by the peers,
    // MLP_TRANSPORT=host for the host mailbox); "rccl" = records delivered by ncclAllGather (rccl_id: the 128-byte ncclUniqueId of
    // rank 0); "pump" = the same pump protocol with peer copies in place of the collective (ranks sharing one GPU: tests)
    void enable_sharding(int rank, int world, const char* shm_name, const char* transport_name = nullptr, const void* rccl_id = nullptr);
    static void rccl_unique_id(void* out128);
    bool sharded() const { return shard_world > 1; }
    bool sharding_live() const { return shard_is_live(); }

    int num_vars = 0;
    int direction = 0;
    int64_t pivot_budget = -1;
    bool budget_exhausted = false;
    bool resume_in_optimize = false;
    bool trace = false, profile = false;
    int sample_every = 0;  // profile mode: 0 = default cadence (every 8th / 4th batch), 1 = every iteration is a sampled eager one
    std::vector<PivotRecord> trace_log;
    Stats stats;
    void resolve_events() {}
    uint64_t state(const char* what, double* out, uint64_t cap);
    int m() const { return m_; }
    int total_vars() const { return N_; }
    int nucleus() const { return k_; }
    int nucleus_cap() const { return cap_; }
    size_t nnz() const { return h_rcol.size(); }

private:
    // --- host mirror (integer bookkeeping + the matrix for rebuilds)
    int m_ = 0, N_ = 0, k_ = 0, cap_ = 0;
    std::vector<double> h_obj, h_lo, h_hi, h_rhs;
    std::vector<int> h_rptr, h_rcol;
    std::vector<double> h_rval;
    // Host view of the columns: only what the host logic needs — the length of every column and, for singleton
    // columns, their single entry (classification of the basis in rebuild_inverse).  The CSC itself lives on the
    // device only (built there from the uploaded CSR-side arrays at try_new, re-laid out there by add_constraint).
    std::vector<int> h_cptr, h_crow;       // initial build only (try_new); dropped after the first device-side append
    std::vector<double> h_cval;
    int max_col_nnz_ = 0, max_row_nnz_ = 0;   // longest column / row of A (in-kernel stage heads need them to fit an LDS list)
    double amax_ = 0.0;                       // max |A_ij| (incl. the slack identity): scale bound of the deterministic blocked push
    bool force_det_push_ = false;             // set around recalc_basic_vals: the blocked push in its deterministic form
    bool pb_det_default = false;              // unsharded solves: deterministic blocked push by default (set from the measured A/B)
    bool ratio_two = false;                  // MLP_RATIO_TWO_KERNELS, or latched by an ITER_STALL: two launches for the two Harris passes
    long long ratio_spin_limit = 20000000LL; // MLP_RATIO_SPIN_LIMIT: polls before a fused ratio test gives up (0: the first launch stalls; tests)
    bool ranks_share_device = false;
    // sparse tableau row (k_row_touch / k_row_pull) while the nucleus is small
    int str_kmax = 230;                      // MLP_STR_K: largest nucleus the sparse form is used for (0: never)
    int sb_kmax = 160;                       // MLP_SMALL_BASIS_K: largest nucleus (at the end of a full record ring) k_small_basis is used for; measured
                                             // crossover with the three launches at k ~ 110-120 (tools/small_basis_curve.py)
    bool sb_now = false;                     // ... decided per batch like str_now (the geometry is baked into the graphs)
    // small-nucleus primal head (primal_head.inc): FTRAN + Harris test + BTRAN + inverse update + touched columns in ONE workgroup, while the
    // nucleus stays within ph_kmax slots for the whole batch (the batch is cut short for that); MLP_PRIMAL_HEAD_K overrides the bound (0: off)
    int ph_kmax = -1;                        // -1: primal_head_kmax(longest column)
    bool ph_now = false;
    bool touch_done = false;                 // the BASIS stage of the iteration being recorded carried the touched-column list
    bool str_now = false, str_clean = false; // geometry of the batch being run; alpha_r / helper are zero outside touched entries
    DevBuf<int> d_ar_list;   // non-basic positions with alpha_r != 0 (dual iteration, CSC-pull tableau row): the dual Harris test walks the list
    bool ar_built_ = false;  // the ROW stage of the iteration being launched listed them
    DevBuf<int> d_str_list;
    // hypersparse single-workgroup iteration (hyper.inc)
    int hyper_mode = -1;                     // MLP_HYPER: 1 on wherever the kernel applies, 0 off, -1 auto (<= 16 non-zeros per row on average)
    long hyper_heavy = 0;                    // MLP_HYPER_HEAVY: eta-update entries one workgroup takes on (0: the kernel's default)
    int hyper_backoff_max = 8;   // after bail-outs in a row the multi-kernel path keeps going for up to 2 << this pivots (swept 2..8 on config 3)
    uint64_t hyper_off_until = 0;            // lifetime pivot count until which the multi-kernel path runs (after bail-outs)
    int hyper_bail_streak = 0;
    DevBuf<int> d_hy_stamp;                  // n + m epoch stamps (alpha_r list / singleton part of the alpha_q list) + the kernel's derived maps
    DevBuf<double> d_hy_score;               // m dual pricing scores (derived by the kernel at every launch)
    size_t hy_stamp_len = 0;
    bool hyper_capable(int phase) const;
    // ---- compact factor of the basis (csrc/factor.inc, HISTORY.md §2.6): B^-1 = (peeled triangular factor of the
    // basis at the last refactorisation)^-1 + at most fac_J_ additive rank-1 terms, instead of the explicit nucleus inverse
    int fac_mode = -1;                       // MLP_FACTOR: 1 from the start (falls back to the explicit inverse when the peel leaves a bump),
                                             // 0 never, -1 auto: tried when the nucleus would need more than fac_auto_cap_ slots
    bool fac_on_ = false;
    int fac_J_ = 64;                         // rank-1 terms the factor has room for (MLP_FACTOR_J: room = period = that value)
    int fac_period_ = 48;                    // pivots between refactorisations: 48, or 64 while a refactorisation is expensive (many levels, a bump)
    bool fac_period_auto_ = true;
    int fac_nlev_ = 0;
    int fac_auto_cap_ = 8192;                // MLP_FACTOR_FROM
    uint64_t fac_tried_at_ = 0;              // lifetime pivot count of the last attempt that found a bump
    bool fac_tried_ = false;
    DevBuf<int> d_fac_pos_of_var, d_fac_var_of_pos, d_fac_prow, d_fac_items, d_fac_lptr, d_fac_meta, d_fac_tmp, d_fac_counters;
    DevBuf<double> d_fac_pval, d_fac_U, d_fac_V, d_fac_rhs, d_fac_x0, d_fac_coef, d_fac_part;
    DevBuf<unsigned> d_fac_bar;
    DevBuf<int> d_fac_bpos, d_fac_brow, d_fac_bslot_of_row, d_fac_irow, d_fac_fptr, d_fac_fidx, d_fac_bptr, d_fac_bidx;
    DevBuf<double> d_fac_ipiv, d_fac_fval, d_fac_bval;
    DevBuf<int> d_fac_lcount;
    DevBuf<double> d_fac_Kd, d_fac_Wtmp, d_fac_gjval;  // bump inversion: K, the work copy of the inverse, partial maxima | scratch
    DevBuf<int> d_fac_gjrow;
    DevBuf<int> d_fac_eslot;  // 2 nz: slots of the targets of the long edge lists (FTRAN | BTRAN)
    DevBuf<int> d_fac_ltslot, d_fac_segs;  // per level: first LDS slot (small levels) | the segments of a solve's walk
    DevBuf<double> d_fac_WbT;               // transpose of the bump inverse
    DevBuf<FacTailRec> d_fac_tprog;  // the tail of the solves as records: FTRAN | BTRAN, FAC_TAIL_CAP each
    DevBuf<int> d_fac_lev3;   // 3 m: level of a position | level of a row's pivot position | reach of a position
    bool fac_skip_ = true;    // every solve walks only the levels its right-hand side can reach
    bool fac_flow_ = false;   // (unused: the data-flow walk of round 5 was removed)
    DevBuf<double> d_fac_Wb;   // allocated with the first bump
    bool fac_pair_ = true;                   // the two solves of a stage share one walk over the levels
    int fac_bump_ = 0;
    int fac_bump_max_ = FAC_BMAX;            // MLP_FACTOR_BUMP: largest bump the compact factor carries through its DENSE inverse
    // sparse factor of the bump (factor_sb.inc): bumps of fac_sb_from_ .. fac_sb_max_ columns are eliminated sparsely (LU with fill in
    // rounds of independent pivots); one whose rows outgrow their slots falls back to the dense inverse (b <= fac_bump_max_)
    int fac_sb_max_ = FAC_SB_MAX;            // MLP_FACTOR_SB: 0 = never
    int fac_sb_from_ = 48;                   // MLP_FACTOR_SB_FROM: smaller bumps keep the dense inverse (one wave-sized product per solve)
    int fac_sb_fail_b_ = 0, fac_sb_fail_skip_ = 0;  // size of the last bump that failed the sparse elimination / refactorisations left before it is tried again
    bool fac_sb_on_ = false;                 // the current factor carries its bump sparsely
    DevBuf<int> d_fac_sb_int;                // integer scratch + outputs of the factorisation (FacSbWork)
    DevBuf<double> d_fac_sb_dbl;
    DevBuf<FacSbRec> d_fac_sb_rec;
    FacSbWork fac_sb_work(int m) const;
    void fac_alloc();
    void fac_fill_view(DevView& v) const;
    bool fac_refactor(int bump_limit = -1);  // the peel + level lists from the current basis; false when it leaves a bump beyond the limit (-1: MLP_FACTOR_BUMP)
    bool fac_enter(int bump_limit = -1);     // switch to the compact factor (false: the basis does not peel; nothing changed)
    void fac_make_room(int need);
    void fac_leave();                        // back to the explicit nucleus inverse (re-inversion from A)
    void launch_stage_fac(int phase, int stage, bool with_events);
    void ensure_hyper();         // sharded solve with another rank on this GPU (or unknown): same
    // Lazy dual steepest edge: the primal loop never reads beta, so its iterations skip tau = B^-1 rho (solver.rs:1157)
    // and the beta recurrence; beta is rebuilt exactly from the basis inverse (k_exact_beta) when something next needs it
    bool lazy_dse = true;                    // MLP_LAZY_DSE=0: maintain beta in every pivot like the reference
    bool beta_stale = false;
    bool batch_lazy = false;                 // the iterations being recorded skip the beta recurrence
    bool lazy_now(int phase) const { return lazy_dse && enable_dse && phase == 0 && !stepping && !fac_on_ && max_row_nnz_ <= HEAD_LIST_CAP; }
    void ensure_beta();
    std::vector<int> h_colnnz, h_single_row;
    std::vector<double> h_single_val;
    std::vector<int> h_basic_vars, h_nb_vars, h_var_loc;
    // slot maps live on the device; pulled on demand (reinvert, add_constraint, clone)
    std::vector<int> h_kslot_of_pos, h_srow_of_pos, h_kslot_of_row, h_pos_of_srow, h_pos_of_kslot, h_row_of_kslot;
    std::vector<double> h_sdiag_of_pos;
    std::vector<uint8_t> h_nb_fixed;
    bool enable_pse = false, enable_dse = false, primal_feasible = false, dual_feasible = false;
    size_t nnz_nonbasic = 0;

    // --- device state
    hipStream_t st = nullptr, st2 = nullptr;  // st2: side branch of the iteration graph
    hipEvent_t evFork[3] = {nullptr, nullptr, nullptr}, evJoin[3] = {nullptr, nullptr, nullptr};
    uint64_t batches_run = 0;
    int eager_iters_in_geom = 0;
    DevBuf<int> d_cptr, d_crow, d_rptr, d_rcol;
    DevBuf<double> d_cval, d_rval, d_lo, d_hi, d_obj;
    DevBuf<int> d_var_loc, d_basic_vars, d_nb_vars;
    DevBuf<double> d_xB, d_loB, d_hiB, d_beta, d_d, d_xN, d_gamma;
    DevBuf<uint8_t> d_nbflags;
    DevBuf<int> d_kslot_of_pos, d_srow_of_pos, d_kslot_of_row, d_pos_of_srow, d_pos_of_kslot, d_row_of_kslot;
    DevBuf<RowInfo> d_rowinfo;
    DevBuf<int> d_bptr;             // banded sweep: band-major copy of A
    DevBuf<unsigned short> d_brow;
    DevBuf<double> d_bval;
    DevBuf<double2> d_band_part;
    bool banded_dirty = true;
    // Locality order of the banded sweep: after many pivots nb_vars is a random permutation and the pass over the
    // band-major copy degenerates into 80-byte gathers (PMC: 521 MB per launch at pivot 240 000 against 150 MB at pivot
    // 200).  The pass therefore visits the positions sorted by the variable they hold; the host rebuilds that order from
    // its mirror of var_loc (O(N)) every `order_every` pivots — a stale order only costs locality, never correctness.
    DevBuf<int> d_nb_order;
    // packed non-basic copy of the band-major matrix in the locality order (DevView.pk_*), rebuilt with the order
    DevBuf<int> d_pk_ptr;
    DevBuf<unsigned short> d_pk_row;
    DevBuf<double> d_pk_val;
    DevBuf<unsigned char> d_pk_valid;
    DevBuf<unsigned char> d_fmark;   // det_pull: row marks of the FTRAN pull (zero outside an FTRAN)
    bool use_pack = true;           // MLP_SWEEP_PACKED=0: the sweep always takes the indirect path through the full copy
    bool pack_built = false;
    size_t band_total_ = 0;         // entries of the band-major copy (incl. pad entries)
    bool use_order = true;          // banded sweep in the locality order (positions sorted by the variable they hold)
    uint64_t order_built_at = 0;    // lifetime pivot count at the last rebuild
    uint64_t lifetime_pivots = 0;   // (stats can be reset by the caller)
    uint64_t order_every = 2048;
    bool order_valid = false;
    uint64_t order_from = 4096;     // pivots after which the order is switched on
    bool order_force = false;       // (loaded bases: from the start)
    void refresh_nb_order(bool force);
    int det_mode = -1;              // MLP_DETERMINISTIC: 1 force the pulled F products, 0 never, -1 auto (<= 2^21 non-zeros)
    int banded_mode = -1;           // MLP_BANDED: 1 force on, 0 off, -1 auto (m >= 4 bands and >= 2^22 non-zeros)
    bool use_banded() const;
  public:
    bool banded_active() const { return use_banded(); }
    bool factor_active() const { return fac_on_; }
    void restart_sampling() { batches_run = 0; }
  private:
    void ensure_banded();
    // pulled F product of the large-nucleus primal iteration (fpull.inc): row-major packed copy of the nucleus columns, rebuilt at the
    // start of every solve / continue call and every fpk_every_ pivots (entering columns are appended in between); alpha_K by variable
    DevBuf<int> d_fpk_cnt, d_fpk_var;
    DevBuf<double> d_fpk_val, d_fpk_x, d_fpk_part;
    DevBuf<unsigned char> d_fpk_in;
    bool fpull_on_ = true;       // MLP_FPULL=0: the blocked push + k_ratio_primal_fused (A/B, tests)
    bool fpk_valid_ = false;
    uint64_t fpk_built_at_ = 0, fpk_builds_ = 0;
    int fpk_every_ = 1024;
    bool fpk_wanted() const;
    void fpk_rebuild();
    DevBuf<int> d_colblk;        // blocked F push (large nucleus): row-block offsets per column
    DevBuf<double> d_push_part;  // ... and its PB_CHUNKS x m partial sums
    bool colblk_dirty = true;
    void ensure_colblk();
    DevBuf<double> d_sdiag_of_pos, d_W, d_U, d_V, d_Ut;
    int ld_pad = 16;    // MLP_LDPAD: extra doubles per row of a large W (row pitch = cap + pad): breaks the power-of-two stride
    int pad_for(int cap) const { return (cap >= 8192 || force_big_tiles) ? ld_pad : 0; }
    int ld() const { return cap_ + pad_for(cap_); }
    int lr_force = -1;  // MLP_LOWRANK: force the delayed-update period (0 = off); default: 32 from cap 8192
    DevBuf<double> d_work;  // alpha_q | tau | rv (2m) | hS  — one memset per batch
    DevBuf<double> d_alpha_r, d_helper;
    DevBuf<int2> d_nb_rng;
    int sweep_variant = 0;
    int sw_balanced = 1;  // MLP_STREAM_BALANCED=0: fixed 128-row strips in the streaming pass (A/B runs); N > 1: that many blocks
    int rt_device = 0;
    void acquire_runtime();  // streams, events, pinned Ctl mirror: recycled across Solutions
    void release_runtime();
    bool force_big_tiles = false;  // MLP_BIGTILE: use the large-nucleus tiling of the fused W pass at any size (tests)
    int shard_rank = 0, shard_world = 1;
    // Round 5: DEFERRED sharding.  While the nucleus is small (the sparse-tableau-row regime, a few hundred pivots from the slack basis)
    // a pivot is 40-60 us of latency-bound launches: two or three per-pivot exchanges over xGMI can only slow it down (measured,
    // oversubscribed: 3 665 against 19 095 pivots/s).  Until the tableau row becomes a pass over A the ranks therefore run as
    // bit-identical REPLICAS — the deterministic unsharded iteration on every rank, no exchange — and the column-block sharding goes
    // live at the first batch that leaves that regime (never back).  MLP_SHARD_DEFER=0: sharded from the first pivot (protocol tests).
    bool shard_defer_ = true;
    bool shard_live_ = false;
    bool shard_is_live() const { return shard_world > 1 && shard_live_; }
    MailRec* d_mail = nullptr;
    void* mail_host = nullptr;
    bool mail_registered = false;
    void* own_box = nullptr;               // peer transport: this rank's mailbox in its own HBM
    void* peer_box[MAX_WORLD] = {};        // ... and the peers' boxes as mapped through HIP IPC
    int mail_fanout = 1;
    size_t xb_cap_ = 0;                    // doubles per vector slot of the exchange buffer (>= the largest possible nucleus)
    bool no_wshard = false;                // MLP_NO_WSHARD: keep the streaming pass replicated on every rank
    void release_mailboxes();
    // pump transport (MLP_TRANSPORT=rccl | pump): the kernels post into / poll this rank's own device box; while a batch is in flight
    // the host moves the records between the ranks on a second stream: stage -> all-gather -> deliver (kernels.hip: k_mail_stage)
    int pump_backend_ = 0;                 // 0 off, 1 RCCL all-gather, 2 peer copies through HIP IPC + a shared-memory barrier
    hipStream_t st_pump = nullptr;
    void* pump_stage_ = nullptr;           // world x MAIL_PUMP_RECS records
    void* pump_stage_peer_[MAX_WORLD] = {};
    unsigned long long* pump_host_ = nullptr;  // pinned: [0, world) gathered batch words, [world] this rank's word
    unsigned long long pump_seq_ = 0, pump_gen_ = 0;
    uint64_t pump_rounds_ = 0;
    void* rccl_comm_ = nullptr;
    void enable_pump(int rank, int world, void* shm, size_t host_bytes, int backend, const void* rccl_id);
    void pump_until_idle();                // call after enqueueing kernels that contain exchanges, before synchronising the stream
    void pump_round(bool done_local, bool* all_done);
    void shm_barrier();
    bool fac_allowed_under_sharding() const;  // the compact factor shards on distinct devices only (enable_sharding)
    void golive_check();                   // deferred sharding: the ranks compare fingerprints of their replicated state before the switch
    uint64_t golive_seq_ = 0, golive_checks_ = 0;
  public:
    std::string transport = "none";        // human-readable name of the exchange transport
  private:
    size_t mail_bytes = 0;
    double refresh_tol = 1e-7;  // re-invert W when the two-way pivot check disagrees by more than this
    DevBuf<double> d_aK, d_rK, d_tK, d_tauK, d_vK, d_klist_a, d_blist_a, d_part_tau, d_part_v;
    DevBuf<int> d_klist_s, d_blist_s;
    DevBuf<double> d_red_key, d_red_key2;
    DevBuf<int> d_red_idx;
    DevBuf<unsigned> d_ticket;
    DevBuf<Ctl> d_ctl;
    DevView hview;
    bool view_dirty = true;
    Ctl* h_ctl = nullptr;  // pinned

    // cached x for get_value
    std::vector<double> h_xB, h_xN;
    bool values_dirty = true;

    // --- iteration graphs: [phase][pse]
    bool use_graph = true;
    int batch = 32;  // (round 4: 16 -> 32: one host round trip per 32 replayed iterations; the driver's 20-pivot window is then ONE batch)
    long final_refresh_pivots = 50000;  // MLP_FINAL_REFRESH: re-examine optimality on recomputed reduced costs after this many pivots (0 = never)
    uint64_t iters_since_recalc = 0, iters_since_polish = 0;
    bool basic_values_feasible();
    bool reduced_costs_feasible();
    bool cold_start_ = true;  // the solve started from the slack basis / a loaded basis (try_new, load_basis); add_constraint, fix_var ... clear it:
                              // warm-start re-solves are short and keep the lazy capture policy
    int graph_iters = 10;  // (round 5: 8 -> 10: the driver's 20-pivot window is two graphs) MLP_GRAPH_ITERS: iterations per graph once a geometry has run for a while (1 = off)
    // graph slots: [0] one iteration per graph, [1] graph_iters iterations per graph (long runs)
    hipGraphExec_t gexec[2][2][2] = {};
    hipGraph_t ggraph[2][2][2] = {};
    Geom ggeom[2][2][2];
    uint64_t graph_batches_in_geom = 0;
    hipEvent_t ev[12] = {};  // sweep0/1, W pass0/1, update0/1, ftran0/1, iteration0/1, fold0/1 (kernel-exact: arm_kernel_timing)
    size_t nnz_nucleus_cols();  // non-zeros of the nucleus basic columns (algorithmic bytes of the F products)
    void drop_graphs();
    hipGraphExec_t get_graph(int phase, int multi);

    Geom geom() const;
    DevView* sync_view();
    void build_csc();
    void upload_matrix();
    void alloc_row_buffers(int m_new);
    void ensure_nucleus_cap(int need);
    void ensure_red();
    void flush_lowrank();
    void pull_ctl();
    void pull_maps();
    void push_maps();
    int col_nnz(int var) const { return h_colnnz[var]; }
    DevBuf<int> d_cptr_alt, d_crow_alt, d_row_idx, d_scan_tmp;  // second CSC buffer set of the device-side append, staging
    DevBuf<double> d_cval_alt, d_row_val;
    void append_row_on_device(const Constraint& c, int slack, int row);

    void record_iteration(int phase, bool with_events);  // enqueue the kernel sequence of ONE iteration
    void launch_stage(int phase, int stage, bool with_events);
  public:
    // engine-level stepping (include/minilp_hip.h: mlp_engine_open / mlp_engine_stage)
    struct StepInfo {
        int32_t status, phase, next_stage;
        int64_t col, row, entering_var, leaving_var;
        double pivot_coeff, step, objective;
        uint64_t nucleus_size;
    };
    int step_phase = 0, step_pos = -1;
    bool stepping = false;
    void recalc_basic_vals();  // x_B recomputed from the basis (polish of long runs)
    // solver.rs:261-270: neither primal nor dual feasible => the feasibility phase runs on an ARTIFICIAL objective (d = +-1 / 0 from
    // try_new); recomputing d from the real costs inside that phase would resume the dual loop on dual-infeasible reduced costs
    bool in_artificial_phase() const { return !primal_feasible && !dual_feasible; }
    void refresh_objective() { recalc_obj_coeffs(); }  // objective + reduced costs of the current point, from the basis (solver.rs:1199-1231)
    int step_open(StepInfo* out);
    int step_stage(int stage, StepInfo* out);
    int step_finish(int phase, int status);
    void fill_step_info(StepInfo* out, int phase) const;

  private:
    int run_loop(int phase);                             // batches of replays until terminal / budget
    int process_records(int phase, int launched);
    void optimize();              // solver.rs:487-511
    void restore_feasibility();   // solver.rs:513-547
    void recalc_obj_coeffs();     // solver.rs:1199-1231
    void calc_col_coeffs(int col);                          // solver.rs:671-677
    void calc_row_coeffs(int row, bool with_sweep);         // solver.rs:680-693
    void fetch_values();
    void rebuild_inverse();       // BasisSolver::reset counterpart (solver.rs:1286-1303)
};

// ---- MPS (mps.rs) on the host side of the product
struct MpsData {
    std::string name;
    std::vector<std::string> var_names;
    ProblemData problem;
};
MpsData parse_mps(const std::string& text, int direction);  // throws MlpError(-1, ..) on syntax errors

}  // namespace mlp
