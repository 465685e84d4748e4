// kernels.h — device-side data layout and launch wrappers of the pivot hot path.
// All arrays live in HBM for the lifetime of a Solution; see DESIGN.md §3 for the layout.
//
// Every kernel takes the DevView BY VALUE (pointers and sizes only: it sits in the kernarg segment,
// so reaching it costs no dependent load).  Nothing a kernel needs changes its launch arguments
// from pivot to pivot — the nucleus size, the pivot scalars and the partition-change plan live in
// the device-resident Ctl block — so a whole simplex iteration is a fixed kernel sequence that is
// captured once into a hipGraph and replayed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mlp {

constexpr double EPS = 1e-8;  // solver.rs:12

enum IterStatus : int {
    ITER_PIVOT = 0,       // (q, r) chosen, basis change
    ITER_FLIP = 1,        // bound flip: q chosen, no leaving row (solver.rs:846-851)
    ITER_OPTIMAL = 2,     // primal: no eligible entering column (solver.rs:737)
    ITER_UNBOUNDED = 3,   // solver.rs:842-844
    ITER_FEASIBLE = 4,    // dual: no infeasible row (solver.rs:534)
    ITER_INFEASIBLE = 5,  // dual: no entering column (solver.rs:1019)
    ITER_SINGULAR = 6,    // partition plan found two singleton columns on one row
    ITER_NONE = 7,        // nothing recorded (halted replay)
    ITER_COMM = 8,        // sharded mode: a peer did not answer within the spin bound
    ITER_STALL = 9,       // an in-kernel wait between two passes of one launch timed out (grid not co-resident)
};

// Per-iteration scalars, written by kernels.
struct IterState {
    int status;
    int q;             // entering non-basic position (index into nb_* arrays)
    int r;             // leaving basic position (-1: bound flip)
    int entering_var;  // nb_vars[q]
    int leaving_var;   // basic_vars[r]
    int sign;          // entering_diff_sign (solver.rs:743)
    int klist_n;       // FTRAN: number of (col-slot, coeff) pairs of a_q that hit nucleus rows
    int blist_n;       // BTRAN: number of (row-slot, coeff) pairs
    double pivot_coeff, leaving_new_val, entering_new_val, entering_diff;
    double entering_cur, entering_other, max_step, pivot_obj;
    double obj;        // cur_obj_val (solver.rs:51)
    double rho_sq;     // ||rho_r||^2      (solver.rs:1160)
    double alpha_sq;   // ||alpha_q||^2+1  (solver.rs:1136)
    double inv_alpha;  // 1 / alpha_q[r], the FTRAN-side pivot used by the inverse update
};

struct StructUpdate {  // HISTORY.md §3.3: how the (P_K, R_K) partition changes at this pivot
    int kase;  // 0 nuc->nuc, 1 sing->nuc (grow), 2 nuc->sing (shrink), 3 sing->sing, 4 sing->sing same row, -1 none
    int r;     // leaving position
    int sr;    // row slot of r (cases 0, 2), -1 otherwise
    int i_r;   // row of the leaving singleton (cases 1, 3, 4)
    int i_q;   // row of the entering singleton (cases 2, 3, 4)
    int cq;    // col slot of i_q (cases 2, 3)
    int kold;  // nucleus size before the change
    int jn;    // delayed-update mode: index the new rank-1 term gets (0 after a fold)
    int pad0;
    int pad;
    double diag_q;      // value of the entering singleton's entry
    double inv_diag_r;  // rho[i_r] = 1/diag of the leaving singleton
};

struct PivotRec {  // one per executed iteration, host reads them back in batches
    int status, phase, q, r, entering_var, leaving_var, kase, k_after;
    int klist_n, blist_n;  // lengths of the FTRAN / BTRAN input lists of this iteration (algorithmic bytes)
    double pivot_coeff, obj;
};
constexpr int RING = 64;
constexpr int LR_MAX = 32;  // capacity of the pending rank-1 list of the delayed-update mode

// Mutable control block (device memory; copied to the host once per batch of replays).
struct Ctl {
    IterState it;
    StructUpdate up;
    int k;       // nucleus size
    int halt;    // set when an iteration ends the loop (optimal / unbounded / ...): later replays no-op
    int ring_n;  // records written since the host last reset it
    int forced;  // dual iteration with a host-forced row (fix_var): skip dual pricing
    // delayed-update mode (HISTORY.md §2.1): W = W0 + sum_{j<nlow} U[j] V[j]^T
    int nlow;   // number of pending rank-1 terms
    int fold;   // this pivot's fused pass folds the pending terms into W0 (primal: set by the FTRAN head, dual: by the plan)
    double lr_c[LR_MAX], lr_e[LR_MAX], lr_g[LR_MAX], lr_h[LR_MAX];  // V[j].a_list, U[j].b_list, V[j].rho_K, U[j].t_K
    int ratio_epoch;          // fused primal ratio test: launch counter published by pass 1's last block
    double ratio_max_step;    // ... and the step bound it publishes
    unsigned long long xepoch[5];  // sharded mode: exchange counters (kind 0: pricing, 1: primal ratio decision,
                                   // 2: dual ratio pass-1 minimum, 3: dual ratio pass-2 candidate,
                                   // 4: tau_K / v_K vectors of the row-sharded streaming pass)
    double max_pivot_err;  // max over the batch of |alpha_q[r] - alpha_r[q]| / max(1,|alpha_q[r]|): drift monitor of W
    long long ratio_spin_limit;  // fused ratio tests: polls of the in-kernel wait before it gives up with ITER_STALL (set by the
                                 // host: 20 M ~ seconds; MLP_RATIO_SPIN_LIMIT=0 makes the first launch stall, for the retry test)
    // hypersparse single-workgroup iteration (hyper.inc): epoch of the per-position stamps, and the flag with which the
    // kernel hands an iteration it will not take (list overflow / too much work for one workgroup) to the multi-kernel path
    int hyper_epoch;
    int hyper_bail;
    int aq_n;        // sparse primal ratio test: listed positions of supp(alpha_q) this iteration
    int str_n;       // sparse tableau row (k_row_touch / k_row_pull): non-basic columns touched by the rows of supp(rho) this iteration
                     // (aq_n and str_n are adjacent: the host clears the pair with one 8-byte memset)
    // compact factor (factor.inc): level ranges a solve has to walk, produced by the solve / head that precedes it (a right-hand side
    // that is zero on every level beyond L leaves those levels at zero: they are skipped)
    int fac_rho_part_n;  // compact factor: the BTRAN that produced rho also left the per-workgroup partial sums of V_j . rho for this many
    int fac_rho_part_ok; // pending terms (region 2 of fac_part); consumed (and cleared) by the FTRAN that solves tau = B^-1 rho
    double fac_inv[64];  // compact factor: 1 / alpha_q[r] of pending term j — fac_U holds (alpha_q - e_r) unscaled, readers form U_j = -(...) * fac_inv[j]
    int fac_aq_hi;     // FTRAN of the entering column: highest level among the pivot positions of its rows
    int fac_rho_hi;    // FTRAN of rho (tau = B^-1 rho): highest level among the rows of supp(rho)        (atomicMax by the BTRAN's epilogue)
    int fac_aq_lo;     // BTRAN of alpha_q (v = B^-T alpha_q): lowest level among supp(alpha_q)           (atomicMin by the FTRAN's epilogue)
    int fac_aq_reach;  // ... and the highest level its dependents reach                                   (atomicMax)
    int kprof_on;    // MLP_KPROF=1: kernels stamp the wall clock into hy_prof (KMARK; state("kernel_timeline"))
    int side_go;     // v branch of the late primal iteration (engine.hip launch_stage): the iteration was live when its FTRAN head ran — the
                     // side kernels (t_K, fold, streaming pass) test this instead of `status`, which the ratio test rewrites beside them
    double sb_obj0;  // running objective before the primal ratio test added this iteration's step (k_small_basis restores it when its wait stalls:
                     // the iteration is then re-run from the top)
    int sb_count;    // launches of k_small_basis that ran an iteration (state("small_basis_launches"): tests check the path was taken)
    int ph_count;    // iterations k_primal_head carried through all its stages (state("primal_head_launches"))
    int ph_rng_x, ph_rng_y;  // k_primal_head with `apply`: CSC range of the leaving variable's column, for nb_rng[q] (set by the update kernel)
    int ar_used;     // dual Harris tests that ran over the listed non-zeros of alpha_r since the Solution was created (state("dual_list_tests"))
    int ar_n;        // entries of DevView.ar_list this iteration (zeroed by the dual ratio test's finaliser and at every batch start)
    int rv_n;        // k_primal_head: rows of (rho, v) the previous iteration's head left non-zero, listed in aq_list (the next head zeroes them:
                     // with the tableau row pulled inside the update kernel nothing else may clear rv while other workgroups still read it)
    unsigned long long hy_prof[24];  // ticks of the 100 MHz wall clock per stage of the hypersparse iteration (diagnostics)
    PivotRec ring[RING];
    // (behind the ring: on lines nothing arrives at during a sweep, and no other field moved)
    int ar_off;   // > 0: the tableau row of a dual iteration is not listed (DevView.ar_list) for this many iterations — the last list
                  // passed AR_CAP entries, and a dense row's appends arrive at ONE address (on a line the sweep only reads: not next to ar_n)
    int ar_back;  // ... the pause grows with every list that overflows again: 16, 32 ... 256 iterations; a list that fits ends it
    int ar_pauses;  // times the listing was paused since the Solution was created (state("dual_list_tests")[1])
    int ar_keep;  // >= 0: the dual Harris test of this iteration ran over that many listed non-zeros of alpha_r (ar_list still holds them): the
                  // update kernel walks the list instead of all n positions (solver.rs:1073-1080 updates d on the non-zeros of row_coeffs); -1: no list
};

// Sharded pricing: one 64-byte mailbox record per (kind, parity, rank) in host
// memory mapped into every rank's GPU; a rank writes only its own slot and polls the others.
constexpr int MAIL_KINDS = 6;  // 0-3 candidate exchanges, 4 vector-exchange flags, 5 transport handshake
constexpr int MAX_WORLD = 16;  // ranks of one sharded solve (one node: 8 GPUs; test rigs oversubscribe one GPU)
struct alignas(64) MailRec {
    unsigned long long epoch;
    double f[7];
};

// Non-basic flags (solver.rs:66-70 NonBasicVarState + nb_var_is_fixed)
constexpr uint8_t NB_AT_MIN = 1, NB_AT_MAX = 2, NB_FIXED = 4;

// Everything the kernels need (passed by value); it changes only when a buffer is re-allocated
// (growth of W, add_constraint), which also invalidates the captured graphs.
// Row-indexed copy of (sdiag_of_pos[pos_of_srow[i]], pos_of_srow[i], kslot_of_row[i]): a singleton row has
// kslot = -1 and (pos, diag) of its covering column; a nucleus row has kslot >= 0, pos = -1, diag = 0.
struct alignas(16) RowInfo {
    double diag;
    int pos;
    int kslot;
};

// One item of a small level of a compact-factor solve, ready to execute: where the result goes (FTRAN: position, BTRAN: row) with the
// number of edges in the top byte, the index of its right-hand-side entry, the pivot, and its (at most six) edges — for each the index
// of the target in the solve vector, the target's slot among the items of the small levels (-1: not one of them), the coefficient.
constexpr int FAC_TAIL = 256;       // levels of at most this many positions are walked by one workgroup (one item per lane)
constexpr int FAC_TAIL_CAP = 1024;  // ... in pieces of at most this many positions: the piece's part of the solve vectors and its inner edges live in LDS
constexpr int FAC_TAIL_K = 4;       // inner edges per position kept in LDS (more: that position walks its record)
constexpr int FAC_TAIL_INLINE = 6;
struct alignas(16) FacTailRec {
    int out_n;   // (n << 24) | out
    int rhs;
    double piv;
    int tgt[FAC_TAIL_INLINE];
    int slot[FAC_TAIL_INLINE];
    double val[FAC_TAIL_INLINE];
    int ovf;     // a longer edge list: place of edge number FAC_TAIL_INLINE in the lists in memory (fac_fidx / fac_fslot / fac_fval ...), else -1
    int pad[3];
};
static_assert(sizeof(FacTailRec) == 128, "16 + 6 x 16 + 16 bytes");
// SPARSE FACTOR OF THE BUMP (round 5, factor_sb.inc; lu.rs:118-304 with the pivot rule of 194-233): the bump K the two-sided peel
// leaves is eliminated right-looking in ROUNDS of mutually independent pivots (threshold 0.1, lowest Markowitz count first); fill is
// stored, rows that would outgrow their slots send the bump back to the dense inverse.  One item of a round, ready to execute in
// pull form: out = (rhs - sum val[e] * x[idx[e]]) / piv  (edges beyond the inline ones at ovf in fac_sb_oidx / fac_sb_oval).
constexpr int FAC_SB_MAX = 4096;   // bump columns the sparse factor carries (four solve vectors of that length live in LDS)
constexpr int FAC_SB_RC = 64;      // entries of an active row, fill included
constexpr int FAC_SB_CC = 64;      // rows ever listed for a column (stale ones included)
constexpr int FAC_SB_LC = 64;      // multipliers of a row
constexpr int FAC_SB_INL = 8;      // edges inside a record
constexpr int FAC_SB_OVS = 64;     // overflow edges per item (>= the longest list)
constexpr int FAC_SB_ROUNDS = 1022;
constexpr int FAC_SB_DT = 128;      // dense tail: once this few columns are left the elimination stops and the rest is inverted densely
constexpr int FAC_SB_KINDS = 6;     // record arrays: L, U, U^T, L^T, FTRAN right-hand side, BTRAN right-hand side
struct alignas(16) FacSbRec {
    int out, rhs, n, ovf;
    double piv, pad;
    int idx[FAC_SB_INL];
    double val[FAC_SB_INL];
};
static_assert(sizeof(FacSbRec) == 128, "32 + 32 + 64 bytes");
struct FacSbWork {  // scratch and outputs of the factorisation (all by bump slot)
    int* slot_of_pos;                         // m: bump slot of a position (valid at bump positions only)
    int* rcol; double* rval; int* rcnt;       // b x RC | b: the active matrix by rows
    int* crow; int* ccnt;                     // b x CC | b: rows listed per column (never shrinks: inactive rows are skipped)
    int* rstate; int* cstate;                 // b: 0 active, else the round (1-based) that pivoted the row / column
    int* lidx; double* lval; int* lcnt;       // b x LC | b: multipliers of a row (pivot row slot, factor) in elimination order
    int* pivcol; double* piv;                 // by row slot
    int* cand_u; int* cand_cost; int* won; int* bid; int* taken;  // selection of a round
    int* place;                               // by row slot: place in round order
    int* flags;                               // [0] overflow, [1] singular / stuck, [2] rounds, [3] columns left
    FacSbRec* rec;                            // 6 x FAC_SB_MAX: L | U | U^T | L^T records in round order, the right-hand sides of the two solves by slot
    int* lptr;                                // FAC_SB_ROUNDS + 2
    int* oidx; double* oval;                  // 6 x FAC_SB_MAX x FAC_SB_OVS overflow edges
    // dense tail (the last <= FAC_SB_DT rows x columns, by ascending slot): their slots, the work copy, K_t^-1 and its transpose
    int* trow; int* tcol; int* tidx;          // DT | DT | b (tail index of a column slot)
    double* tK; double* tW; double* tinv; double* tinvT;  // DT x DT each: [a][c] work arrays, tinv[c][a] = (K_t^-1)[column c, row a], tinvT[a][c]
};
struct DevView {
    int m, n;  // constraints (= basic positions), non-basic positions (= num_vars)
    int ld;    // leading dimension (= capacity) of W
    int pad0;
    // constraint matrix, both orientations (solver.rs:21-22), int32 indices, f64 values
    const int* csc_ptr; const int* csc_row; const double* csc_val;  // N+1, nnz, nnz
    const int* csr_ptr; const int* csr_col; const double* csr_val;  // m+1, nnz, nnz
    const double* var_lo; const double* var_hi; const double* obj_c;  // N (orig_var_mins/maxs/obj_coeffs)
    int* var_loc;  // N: >=0 basic position, <0: -1-nb position   (solver.rs:33 var_states)
    // basic side, by position (solver.rs:37-41)
    int* basic_vars; double* xB; double* loB; double* hiB; double* beta;
    // non-basic side, by nb position (solver.rs:44-49)
    int* nb_vars; double* d; double* xN; double* gamma; uint8_t* nbflags;
    int2* nb_rng;  // n: CSC [begin, end) of the column at each non-basic position (cache of csc_ptr[nb_vars[c]])
    const int* nb_order;  // n or null: locality order of the banded sweep (positions sorted by the variable they hold)
    // Packed copy of the band-major matrix for the CURRENT non-basic set, in the locality order (rebuilt with it): index
    // i of the pass reads its segments front to back — no holes where basic variables sit, no position -> variable ->
    // offset chain.  pk_valid[i] is cleared by the update kernel when the position served by i changes its variable
    // (at most one per pivot until the next rebuild); those indices take the indirect path.
    const int* pk_ptr;             // nbands x (n + 1) offsets, or null
    const unsigned short* pk_row;
    const double* pk_val;
    unsigned char* pk_valid;       // n
    // basis inverse: singleton split + dense nucleus inverse W (HISTORY.md §3.2)
    int* kslot_of_pos;     // m: row slot of W for a nucleus position, -1 for a singleton position
    int* srow_of_pos;      // m: the row of the single entry of a singleton basic column
    double* sdiag_of_pos;  // m: its value
    int* kslot_of_row;     // m: col slot of W for a nucleus row, -1 for a row covered by a singleton
    int* pos_of_srow;      // m: position of the singleton covering that row
    RowInfo* rowinfo;      // m: the three row maps packed for the F pushes (one 16-byte gather per matrix entry)
    // blocked F push of the large-nucleus regime (pb_on): per column, the CSC offset at which each block of
    // PB_ROWS rows starts (N x (pb_rb + 1)), and PB_CHUNKS x m partial sums
    int* colblk;
    double* push_part;
    int pb_rb, pb_on;
    // Deterministic form of the blocked push (pb_det; sharded solves, where the ranks must stay bit-identical replicas, and —
    // measured no slower — the default everywhere): the LDS accumulators are fixed-point integers (a 64-bit high limb and a
    // 32-bit low limb per row, scaled per slot chunk by the largest |x_K| of the chunk times pb_amax = max |A|), so that the
    // LDS atomics are exact, associative integer additions: the result does not depend on their order.  pb_hbits = bits of
    // headroom for the number of terms one accumulator can receive (ceil(log2(longest row + 1))).
    int pb_det, pb_hbits;
    double pb_amax;
    // banded tableau-row sweep (large m): a second copy of A in band-major order (bands of BAND_ROWS rows;
    // per band a CSC with 16-bit local row indices), so that a workgroup can hold its band of (rho, v) in LDS
    int* bptr;              // nbands x (N + 1)
    unsigned short* brow;   // row index inside the band, 16 bits; every (band, column) segment has an even length
                            // (zero-valued pad entry), so that 8 of them load as one 4-byte-aligned 16-byte word
    double* bval;           // same length as brow (+ 8 spare entries)
    double2* band_part;     // nbands x n partial (alpha_r, helper)
    int nbands, banded;
    unsigned char* fmark;  // det_pull: singleton positions the current FTRAN's nucleus columns reach (zero outside an FTRAN); nullptr: pull every row
    int det_pull;  // small models: the F products are pulled per singleton row (fixed summation order) instead of
                   // pushed with float atomics, so that a solve is reproducible bit for bit from run to run
    int* pos_of_kslot;     // cap: row slot -> position
    int* row_of_kslot;     // cap: col slot -> row
    double* W;             // cap x ld, row-major: W[rowslot(p)][colslot(i)] = (B^-1)[p, i]  (W0 in delayed-update mode)
    double* U;             // LR_MAX x ld: pending rank-1 terms, row-slot side   (delayed-update mode)
    double* Ut;            // ld x LR_MAX: the same terms slot-major (Ut[slot][j] = U[j][slot]), kept in step by lowrank_append: the fold
                           // reads the LR_MAX values of a row slot as ONE contiguous, wave-uniform (scalar) load
    double* V;             // LR_MAX x ld: pending rank-1 terms, col-slot side
    int lrJ;               // 0: every pivot updates W in place; J > 0: fold every J pivots
    // Balanced strips of the streaming pass: with sw_nbal > 0 (the number of k_stream_w blocks the device holds at once)
    // the strip height follows k so that every co-resident block gets ONE tile of equal size — the fixed 128-row strips
    // leave 3 381 tiles for 1 024 slots at k = 20 500, i.e. some CUs stream four tiles while others stream three.
    int sw_nbal, sw_pad;
    int* hy_stamp_n;  // n: hypersparse iteration: epoch at which a non-basic position last entered the alpha_r list
    int* hy_stamp_p;  // m: ... a singleton basic position last entered the alpha_q list
    int* hy_var_slot; // n + m: nucleus slot of a variable (-1: non-basic or singleton basic); rebuilt at every launch of the kernel
    int2* hy_brng;    // m: CSC range of the basic column at each position (same)
    double* hy_score; // m: dual pricing score infeasibility^2 / beta of each row, -inf when feasible (same)
    // per-pivot vectors
    double* alpha_q;  // m by position  (col_coeffs,            solver.rs:54)
    double* tau;      // m by position  (B^-1 rho,              solver.rs:1157)
    double2* rv;      // m by row: .x = rho (inv_basis_row_coeffs, solver.rs:56), .y = v = B^-T alpha_q (solver.rs:1114)
    double* alpha_r;  // n by nb pos    (row_coeffs,            solver.rs:57)
    double* helper;   // n by nb pos    (sq_norms_update_helper solver.rs:55)
    // nucleus-slot work vectors (cap each)
    double* aK;    // alpha_q restricted to nucleus positions (by row slot)
    double* rK;    // rho restricted to nucleus rows (by col slot)
    double* tK;    // rhs of the transposed nucleus solve (by row slot)
    double* tauK;  // W rK (by row slot)
    double* vK;    // W^T tK (by col slot)
    int* klist_s; double* klist_a;  // FTRAN input list (col slots)
    int* blist_s; double* blist_a;  // BTRAN input list (row slots)
    double* part_tau;  // [ncolchunks][ld]
    double* part_v;    // [nrowstripes][ld]
    // reductions
    double* red_key; double* red_key2; int* red_idx; unsigned* ticket;
    Ctl* ctl;
    // column-block sharding of the pricing path: this rank owns non-basic positions [nb_lo, nb_hi)
    int nb_lo, nb_hi, rank, world;
    // Mailboxes of the per-pivot exchanges, one box of [MAIL_KINDS][2 parities][world] records per rank.  `mail` is
    // the box this rank POLLS.  Peer transport (default): the box lives in this GPU's own HBM (uncached /
    // fine-grained), every peer maps it through a HIP IPC handle and writes its slot into it over xGMI, so a post
    // is `mail_fanout` = world remote stores and a wait polls local memory.  Host transport (MLP_TRANSPORT=host):
    // one box in host memory shared by all ranks (mail_fanout = 1, every poll crosses PCIe).
    MailRec* mail;                   // null when world == 1
    MailRec* mail_peer[MAX_WORLD];   // mail_peer[r]: rank r's box as mapped into this process (mail_peer[rank] == mail)
    int mail_fanout;
    // Row-sharded streaming pass of the large-nucleus mode (wshard = 1, peer transport only): rank g streams the strips
    // s with s % world == g of W0, writes its tau_K rows and its partial v_K into EVERY rank's exchange buffer (xGMI
    // stores), and every rank assembles tau_K (rows from their owners) and v_K (partials summed in rank order) from its
    // own buffer.  xbuf[parity][source rank][0: tau_K, 1: v_K partial][xb_cap] doubles, behind the mailbox of the same
    // device allocation (so the peers' IPC mappings cover it).
    int wshard;
    double* xbuf;                    // this rank's exchange buffer (polled / read locally)
    double* xbuf_peer[MAX_WORLD];    // rank r's exchange buffer as mapped into this process
    int xb_cap, pad3;
    // Sparse tableau row (str_on; small nucleus, one GPU): alpha_r = rho^T N (and the PSE helper N^T v on the same columns)
    // over the columns that meet a row of supp(rho) only, listed in str_list through the epoch stamps hy_stamp_n, instead of
    // a pass over all of A.  alpha_r / helper are then kept ZERO outside the touched entries (the update kernel zeroes what it
    // used).
    int* ar_list;    // n or null: non-basic positions with alpha_r != 0, listed by the CSC-pull tableau row of a dual iteration (k_sweep MODE 0);
                     // Ctl.ar_n entries.  The dual Harris test runs over the list in ONE block when it is short (solver.rs:962-1002 walks the
                     // non-zeros of row_coeffs only) instead of two grid-wide passes over all n positions
    int* str_list;   // n
    int* aq_list;    // m: positions of supp(alpha_q) (listed by the FTRAN when str_on and the F products are pushed)
    int str_on, pad4;
    // PULLED F product of the large-nucleus primal iteration (round 6, fpull.inc): a ROW-major packed copy of the nucleus columns of A.
    // Row i owns the slots csr_ptr[i] .. csr_ptr[i + 1] of fpk_var / fpk_val (a row of the copy is a subset of the row of A) and holds
    // fpk_cnt[i] entries (variable, value): the columns that were nucleus basics when the copy was built, in CSR order, then one entry
    // per column that entered the basis since, in pivot order (appended by the pivot's partition-change blocks).  alpha_S[i] =
    // (a_q[i] - sum_e fpk_val[e] * fpk_x[fpk_var[e]]) / diag_i is then ONE gather pass per singleton row (fixed summation order, no
    // atomics, no partial sums).  fpk_x = alpha_K scattered BY VARIABLE (zero for every variable that is not a nucleus basic: a column
    // that has LEFT the basis contributes exact zeros until the next build drops its entries).
    int* fpk_cnt;            // m: entries of row i in use
    int* fpk_var;            // nnz: variable (column of A)
    double* fpk_val;         // nnz: coefficient
    unsigned char* fpk_in;   // N: the variable's column is part of the packed copy
    double* fpk_x;           // N: alpha_K by variable
    double* fpk_part;        // 2 per block of k_fpull_p1: its part of Harris pass 1 (minimum) and of ||alpha_q||^2 — a buffer of their own: the
                             // blocks of k_fpull_p2 fold them while earlier blocks of the same launch already post pass-2 candidates in red_key
    int fpk_on, pad5;
    // ---- compact factor of the basis (csrc/factor.inc, HISTORY.md §2.6) -------------------------------------
    // fac_on: B^-1 is NOT held as singleton split + dense nucleus inverse but as a frozen PEELED TRIANGULAR FACTOR of the
    // basis B0 of the last refactorisation — an iterated column-singleton peel orders (pivot row, position) pairs into levels
    // such that B0 is upper triangular in that order; nothing is stored beyond A itself and the order (lu.rs:118-304 without
    // fill: the peel IS the factorisation when the bump is empty) — plus the eta transformations since then in additive form,
    // B^-1 = B0^-1 + sum_{j < nlow} U_j V_j^T (solver.rs:1274-1284 kept as full-space rank-1 terms, dense, at most fac_J).
    // Solves are level-scheduled pulls (k_fac_solve; lu.rs:79-106, 432-463): FTRAN walks CSR rows in descending level
    // order, BTRAN walks CSC columns in ascending level order; the maps below are SNAPSHOTS taken at the refactorisation
    // (basic_vars / var_loc move on with every pivot).
    int fac_on, fac_J;
    int* fac_meta;           // [0] number of levels, [1] peeled positions, [2] bump columns, [3] first level of the single-workgroup tail
                             // [4] column steps of the peel (= level of the bump), [5] row steps, [6] positions in small levels, [7] segments of the walk
    int* fac_pos_of_var;     // N: position of a variable in B0, -1 when it was non-basic
    int* fac_var_of_pos;     // m: variable at a position in B0
    int* fac_prow;           // m: pivot row of a position
    double* fac_pval;        // m: pivot element A[prow[p], var_of_pos[p]]
    int* fac_items;          // m: positions in level order
    int* fac_lptr;           // levels + 1 offsets into fac_items
    // per place in fac_items (level order), rebuilt with the peel: pivot row and pivot element, and the RESOLVED edge lists of the
    // two solves — FTRAN: the other basic columns of the pivot row as (position, value), BTRAN: the other rows of the column
    int* fac_lev_of_pos;     // m: level of a position (-1: bump)
    int* fac_lev_of_row;     // m: level of the position that pivots on a row (-1: bump row)
    int* fac_reach_of_pos;   // m: highest level any BTRAN dependent of the position reaches
    int fac_skip, fac_flow;  // fac_skip: walk only the levels a right-hand side can reach (always on); fac_flow: unused (the data-flow walk of round 5 was removed)
    // the single-workgroup tail of the solves (factor.inc): place of a position / of a row's pivot position in fac_items (-1: bump),
    // and the tail's items as fixed-size records in walking order (one per direction)
    int* fac_idx_of_pos; int* fac_idx_of_row;
    const FacTailRec* fac_tprog_f; const FacTailRec* fac_tprog_b;
    int* fac_fslot; int* fac_bslot;  // per edge of the lists in memory: slot of the target among the items of the small levels (-1: not one); filled for the long lists only
    int* fac_ltslot;   // per level: first LDS slot of its positions when workgroup 0 walks it alone (a small level), else -1
    int* fac_segs;     // the walk of a solve as fac_meta[7] segments (kind, first level, last level), ascending: 0 grid, 1 small levels, 2 bump
    const double* fac_WbT;   // transpose of fac_Wb (the FTRAN walks columns)
    int* fac_irow; double* fac_ipiv;
    int* fac_fptr; int* fac_fidx; double* fac_fval;   // m + 1 | entries of the basis
    int* fac_bptr; int* fac_bidx; double* fac_bval;
    double* fac_U;           // fac_J x m, by position: U_j = -(alpha_q - e_r) / alpha_q[r] of the j-th pivot since the refactorisation
    double* fac_V;           // fac_J x m, by row: V_j = rho_r of that pivot
    double* fac_rhs;         // m, by row: right-hand side of a column FTRAN (zero outside a solve: the solve clears what the head scattered)
    double* fac_x0;          // m: result of the B0 solve before the rank-1 terms are added (by position for FTRAN, by row for BTRAN)
    double* fac_coef;        // 2 * fac_J + 1: coefficients V_j . rhs / U_j . c of the running solve
    double* fac_part;        // fac_J x 1024: per-workgroup partial sums of those dot products (dense right-hand sides)
    unsigned* fac_bar;       // [0] grid barrier counter, [1] exit ticket, [2] reduction ticket, [FAC_BAR_FLAG] the barrier's release word
    // the BUMP: what the peel leaves (columns on cycles of the basis graph: a generalised network has one-cycle components).
    // In peel order B0 = [[U, F], [0, K]] with K the bump (fac_meta[2] columns, at most FAC_BMAX): its inverse is kept explicitly
    // (fac_Wb, Gauss-Jordan at the refactorisation); FTRAN solves the bump first, BTRAN last.
    int* fac_bpos;           // FAC_BMAX: position of a bump slot
    int* fac_brow;           // FAC_BMAX: row of a bump slot
    int* fac_bslot_of_row;   // m: bump slot of a row, -1 for a pivot row of the peel
    double* fac_Wb;          // FAC_BMAX x FAC_BMAX, row-major: Wb[s][u] = (K^-1)[position slot s, row slot u]
    // sparse factor of the bump (fac_meta[8] = 1: in use, fac_meta[9] = rounds): records L | U | U^T | L^T, FAC_SB_MAX each, in round order
    const FacSbRec* fac_sb_rec; const int* fac_sb_lptr; const int* fac_sb_oidx; const double* fac_sb_oval;
    // ... its dense tail (fac_meta[10] rows x columns, the last level of the rounds): slots and the inverse in both layouts
    const int* fac_sb_trow; const int* fac_sb_tcol; const double* fac_sb_tinv; const double* fac_sb_tinvT;
};
constexpr int FAC_BMAX = 1024;
constexpr int AR_CAP = 2048;     // entries of DevView.ar_list the one-block dual Harris test takes (ratio_dual_list); k_sweep stops listing beyond
constexpr int TK_ONE = 128;      // a ticket (last_block_arrives) counts up to this many arrivals at one address, more in two steps
constexpr int TK_GROUPS = 16;    // first-step tickets of the two-step form
constexpr int TK_STRIDE = 1024;  // words between them: ticket[TK_STRIDE * (1 + x)], x < TK_GROUPS
constexpr int TK_WORDS = TK_STRIDE * (TK_GROUPS + 1) + 8;  // words a buffer that holds tickets in its first words needs
constexpr int FAC_BAR_FLAG = TK_WORDS + 1024;  // fac_bar[FAC_BAR_FLAG]: the grid barrier's release word, on a line neither the arrival counter nor a ticket uses

// fused pass tiling
constexpr int FW_TR = 8;     // minimum rows per block (sizes the partial buffers; kernels use 8 or 16)
constexpr int FW_TC = 1024;  // columns per block (256 threads x 4)

// Launch geometry that is baked into a captured graph.
constexpr int BAND_ROWS = 8192;  // rows per band of the banded sweep: 128 KB of (rho, v) pairs in LDS
constexpr int BAND_THREADS = 1024;
constexpr int PB_ROWS = 4096;   // rows per LDS block of the blocked F push (32 KB of doubles)
constexpr int PB_CHUNKS = 48;   // capacity of the partial-sum buffer of the blocked F push (column chunks = slot ranges)
constexpr int PB_CHUNKS_DEFAULT = 24;  // chunks used

struct Geom {
    int m, n, cap;
    int lanes;  // lanes per CSC column in the pull kernels (4, 16 or 64; from the average column length)
    int sweep_one;  // average column below 3 entries: the CSC-pull tableau row takes ONE lane per column (a quarter of the workgroups; sums in storage order)
    int sweep_variant;  // 0 (the lane / gather-chain variants of round 1 are gone as a choice)
    int big;            // fused W pass: 64-row x 1024-column blocks, non-temporal (cap > 4096, or forced by MLP_BIGTILE)
    int head_fused;     // stage heads run inside the consuming kernel (delayed-update mode off, every column / row fits the LDS list)
    int str;            // sparse tableau row instead of the sweep over all of A (nucleus of at most MLP_STR_K columns, one GPU)
    int sb;             // nucleus small enough for the one-launch BTRAN + pass + v tail + touch of the lazy primal iteration (k_small_basis; MLP_SMALL_BASIS_K)
    int fac;            // compact factor of the basis instead of the explicit nucleus inverse (factor.inc)
    int ph;             // small nucleus, lazy primal iteration: FTRAN + Harris test + BTRAN + inverse update + touched columns in ONE workgroup (k_primal_head)
    int ratio_two;      // the two Harris passes as two launches (no in-kernel wait): MLP_RATIO_TWO_KERNELS, ranks sharing a device, after an ITER_STALL
    int fp;             // large nucleus, lazy primal iteration: the F product of the FTRAN is PULLED inside the ratio test's launch (fpull.inc)
};

// ---- launch wrappers (all asynchronous on `st`; the DevView is passed to the kernels by value) ----
void launch_clear_work(const DevView& dv, hipStream_t st);  // alpha_q, tau, rv := 0 (one memset)
void launch_price_primal(const DevView& dv, const Geom& g, int use_pse, hipStream_t st);  // standalone K1 (first iteration of a batch)
void launch_price_dual(const DevView& dv, const Geom& g, int use_dse, hipStream_t st);    // standalone K6
void launch_ftran_prep(const DevView& dv, int derive_primal, hipStream_t st);             // FTRAN head (one wave)
void launch_ftran_gather(const DevView& dv, const Geom& g, hipStream_t st, int ys = 0);               // alpha_q = B^-1 a_q
void launch_ftran_fused(const DevView& dv, const Geom& g, int derive_primal, hipStream_t st);
// delayed-update mode (large nucleus), primal iteration, one GPU: head + gather (+ blocked push) without the one-wave launch in front
bool ftran_head_rides_gather(const DevView& dv, const Geom& g);
void launch_ftran_gather_lrh(const DevView& dv, const Geom& g, hipStream_t st, int ys = 0, int fpk = 0);   // FTRAN head + gather in one launch; fpk: no blocked push (fpull.inc)
// pulled F product + both Harris passes + BTRAN head + plan | t_K in ONE launch (fpull.inc); the packed copy is built by the two passes below
bool fpull_supported(const DevView& dv, const Geom& g);
void launch_fpull_ratio(const DevView& dv, const Geom& g, hipStream_t st, hipEvent_t ftran_done = nullptr);
void launch_fpk_build(const DevView& dv, hipStream_t st);   // the packed copy from the CSR of A and the current maps (fpk_in cleared beforehand)
void launch_btran_fused(const DevView& dv, const Geom& g, int with_rhs, int derive_dual, hipStream_t st);  // BTRAN head + gather (dual iteration)
constexpr int HEAD_LIST_CAP = 1024;  // entries an in-kernel stage head can hold (longest column / row of A)
void launch_ratio_primal(const DevView& dv, const Geom& g, int use_pse, hipStream_t st, int tk_ride = 0);  // K5 p1 (+alpha_sq, y_S), p2 (+BTRAN head, plan) [| t_K blocks]
bool tk_rides_ratio(const DevView& dv, const Geom& g);
bool tk_rides_ratio_small(const DevView& dv, const Geom& g);  // the same for a small nucleus (k_small_basis), y_S formed on the fly  // large nucleus, lazy primal iteration: t_K is formed by blocks riding behind the ratio blocks
void launch_post_ftran(const DevView& dv, const Geom& g, int use_pse, hipStream_t st);    // dual path: alpha_sq, y_S, plan
void launch_btran_prep(const DevView& dv, int derive_dual, int plan_after, hipStream_t st);  // BTRAN head (one wave)
void launch_btran(const DevView& dv, const Geom& g, int with_rhs, hipStream_t st, int after_fold = 0);  // rho, rK, rho_sq [| tK]
void launch_btran_rhs(const DevView& dv, const Geom& g, hipStream_t st);                  // tK alone
bool launch_sweep(const DevView& dv, const Geom& g, int mode, int with_struct, hipStream_t st, int inline_combine = 0, int list = 0);  // K4 [| partition change]; true: mode 0 and the non-zeros of alpha_r were listed (DevView.ar_list)
// sparse tableau row: touched-column list, then the pull of alpha_r / helper on the listed columns (| partition change)
void launch_row_sparse(const DevView& dv, const Geom& g, int mode, int with_struct, int touch, hipStream_t st);
// small nucleus (first capacity), lazy primal iteration: BTRAN + pass over W + v tail + touched-column list in ONE launch
bool small_basis_supported(const DevView& dv, const Geom& g);
void launch_small_basis(const DevView& dv, const Geom& g, hipStream_t st, int tk_inside = 1);  // tk_inside = 0: t_K rode in the ratio launch
// small nucleus (at most primal_head_kmax slots for the whole batch), lazy primal steepest-edge iteration, pushed F products: everything between
// the pricing decision and the tableau row in ONE launch of one workgroup (primal_head.inc)
bool primal_head_supported(const DevView& dv, const Geom& g);
int primal_head_kmax(int longest_column);  // largest nucleus the kernel serves for a model whose longest column has that many entries (0: none)
void launch_primal_head(const DevView& dv, const Geom& g, hipStream_t st);
void launch_init_nb_rng(const DevView& dv, const Geom& g, hipStream_t st);
void launch_ratio_dual(const DevView& dv, const Geom& g, hipStream_t st, int list_ok = 0); // K7 p1, p2 (+FTRAN head); list_ok: the tableau row of this iteration listed its non-zeros
void launch_fused_w(const DevView& dv, const Geom& g, int with_v, hipStream_t st, int with_tau = 1);  // tauK/vK partials + eta update of W
int launch_post_fused(const DevView& dv, const Geom& g, int with_v, hipStream_t st, int classic = 0, int skip_push = 0, int with_tau = 1, int touch = 0, int rk_ride = 0);
bool rk_rides_post(const DevView& dv, const Geom& g);  // (strip-tiled tail of the large-nucleus pass: the form that carries the rho_K blocks)
void launch_exact_beta(const DevView& dv, hipStream_t st);  // beta_p = ||e_p^T B^-1||^2 for every basic position (lazy dual steepest edge)
// hypersparse iteration (hyper.inc): up to max_iters dual iterations (no primal steepest edge) in ONE launch of one workgroup
void launch_hyper_dual(const DevView& dv, int use_dse, int max_iters, long heavy, hipStream_t st);  // heavy <= 0: default work bound
void launch_mail_handshake(const DevView& dv, int* out, hipStream_t st);  // transport self-test at enable_sharding
// pump transport: stage this rank's records / deliver the peers' records (between them: an all-gather of the staging blocks)
constexpr int MAIL_PUMP_RECS = MAIL_KINDS * 2 + 1;  // 64-byte records per rank in the staging buffer (the last one: the host's batch word)
void launch_mail_stage(const DevView& dv, void* stage, hipStream_t st);
void launch_mail_deliver(const DevView& dv, const void* stage, hipStream_t st);
// sampled iterations: stamp (t0, t1) at the start / end of the next kernel of a slot (1 tableau-row sweep, 2 pass over the
// nucleus inverse, 3 fold) instead of bracketing its launch; (nullptr, nullptr) disarms
void arm_kernel_timing(int slot, hipEvent_t t0, hipEvent_t t1);
bool stream_strips_enabled();
bool fold_fuses_v(const DevView& dv, int with_v, int with_tau, int fold_only);  // a folding pivot's fold also produces the v partials (its streaming pass is skipped)
int stream_coresident_blocks();  // blocks of the default k_stream_w instance the device holds at once (0: unknown)
void launch_structure_update(const DevView& dv, const Geom& g, hipStream_t st);
void launch_update_pivot(const DevView& dv, const Geom& g, int phase, int use_dse, int use_pse, hipStream_t st, int inline_comb = 0,
                         int with_struct = 0, int pull_inside = 0);  // K8 + clear + next pricing [pull_inside: + the sparse tableau row, per workgroup]
bool update_pulls_inside(const DevView& dv, const Geom& g);
bool head_applies(const DevView& dv, const Geom& g);  // the head also applies the basic side of the pivot and position q (update kernel: pull_inside = 2)  // small-nucleus primal head: the touched columns are pulled by the update kernel's own workgroups
// non-graph helpers
void launch_set_iter(const DevView& dv, int status, int q, int r, double lnv, int forced, hipStream_t st);
void launch_reset_ring(const DevView& dv, hipStream_t st);
void launch_btran_dense(const DevView& dv, const Geom& g, hipStream_t st);  // y = B^-T c_B -> rv.x (c_B gathered into alpha_q)
void launch_recalc_d(const DevView& dv, const Geom& g, hipStream_t st);     // d_c = obj - a_c.y ; obj
void launch_recalc_basic_vals(const DevView& dv, const Geom& g, const double* rhs, double* r_tmp, int refine, hipStream_t st);  // x_B = B^-1 (b - N x_N); refine: x_B += B^-1 (b - A x)
void launch_shift_nonbasic(const DevView& dv, const Geom& g, int col, double val, hipStream_t st);
void launch_sq_norms_add_row(const DevView& dv, const Geom& g, hipStream_t st);
void launch_copy_rho_sq_to_beta(const DevView& dv, int row, hipStream_t st);
void launch_build_nucleus(const DevView& dv, const Geom& g, double* Kd, int k, hipStream_t st);
// compact factor (factor.inc)
// one level-scheduled solve: dir 0 FTRAN (src 0 entering column | 1 rho | 2 src_ptr by row; dst 0 alpha_q | 1 tau),
// dir 1 BTRAN (src 0 e_r | 1 alpha_q | 2 src_ptr by position; dst 0 rho + ||rho||^2 | 1 v); always: whatever the iteration status
// fuse (dual iteration without primal steepest edge, not stepping): 1 = the leaving row's scalars (k_btran_prep) in the head of the BTRAN,
// 2 = the plan (k_post_ftran) and 4 = the new rank-1 term (k_fac_append) in the epilogue of the FTRAN; 8 = the BTRAN does NOT leave the
// partial sums of V_j . rho behind (MLP_FACTOR_RHO_PART=0: A/B, tests)
void launch_fac_solve(const DevView& dv, const Geom& g, int dir, int src, int dst, const double* src_ptr, int always, hipStream_t st, int fuse = 0);
void launch_fac_solve2(const DevView& dv, const Geom& g, int dir, int srcA, int dstA, int srcB, int dstB, hipStream_t st, int fuse = 0);  // two right-hand sides, one walk over the levels
int fac_solve_grid_blocks();  // workgroups of k_fac_solve's grid: workgroup j reduces the coefficient of pending term j, so fac_J must not exceed it
void launch_fac_append(const DevView& dv, hipStream_t st);     // U_nlow, V_nlow from alpha_q / rho of this pivot; nlow += 1
void launch_fac_gather_cb(const DevView& dv, hipStream_t st);  // alpha_q[p] = c[basic_vars[p]]
// refactorisation (host-paced peel): init, then claim + commit per level, then the level lists
void launch_fac_peel_init(const DevView& dv, int* cnt, int* level, int* row_lev, int* claim, int* rcnt, int* claim_r, hipStream_t st);
void launch_fac_peel_level(const DevView& dv, int lev, int* cnt, int* level, int* row_lev, int* claim, int* cand_row, int* counters, hipStream_t st);
// the whole peel in one launch (column steps and row steps): lsteps[0] = steps that removed something, lsteps[t] = +/- positions peeled
// up to step t (+ column step, - row step)
void launch_fac_peel_all(const DevView& dv, int* cnt, int* level, int* row_lev, int* claim, int* cand_row, int* counters, int* lsteps, int max_levels,
                         int* rcnt, int* claim_r, int* cand_col, hipStream_t st);
void launch_fac_peel_fill(const DevView& dv, const int* level, int* cursor, hipStream_t st);
void launch_fac_edges(const DevView& dv, int pass, int* fcnt, int* bcnt, const int* level, hipStream_t st);  // resolved edge lists in level order: pass 0 counts, pass 1 fills
void launch_fac_bump_invert(const DevView& dv, double* K, double* W, double* out, double* outT, int b, int* flag, double* part_val, int* part_row, hipStream_t st);
void launch_fac_bump_transpose(const double* in, double* outT, int b, hipStream_t st);  // K^-1 of the bump, one launch
void launch_fac_tail_prog(const DevView& dv, FacTailRec* pf, FacTailRec* pb, int nlev, hipStream_t st);
void launch_fac_plan(const DevView& dv, int* ltslot, int* segs, hipStream_t st);  // small levels, their LDS slots, the segments of a solve's walk (device-side)  // the tail's items as records (both directions)
void launch_fac_reach_all(const DevView& dv, int lev_hi, int lev_lo, hipStream_t st);
void launch_fac_reach_level(const DevView& dv, int lev, int size, hipStream_t st);  // one large level by the grid  // reach_of_pos of every position (levels in descending order, one launch); level of the bump
void launch_fac_sb_factor(const DevView& dv, const int* level, const FacSbWork& w, int b, hipStream_t st);  // sparse factor of the bump (one workgroup)
void launch_fac_bump_build(const DevView& dv, double* Kd, int b, hipStream_t st);  // K = B0[bump rows, bump columns], dense, row-major with pitch FAC_BMAX
void launch_str_reset(const DevView& dv, hipStream_t st);  // sparse tableau row: new stamp epoch, empty lists
void launch_checksum_w(const DevView& dv, unsigned long long* out, hipStream_t st);  // order-independent checksum of W[0:k, 0:k] and the slot maps (tests)
void launch_fold_lowrank(const DevView& dv, const Geom& g, hipStream_t st);  // W0 += U^T V, nlow := 0 (host-requested flush)
void launch_gauss_jordan(double* Kd, double* Winv, int k, int ld, int* d_flag, double* d_scratch, hipStream_t st);
// blocked in-place Gauss-Jordan inversion of the nucleus held in dv.W (inverse.inc); *flag = 1: singular
void launch_blocked_inverse(const DevView& dv, int k, double* rowbuf, int nrowbuf, int* piv, int* src, double* prow, double* ckey,
                            int* cidx, int* flag, hipStream_t st);

// device-side matrix maintenance (add_constraint without a host pass over the non-zeros; also the initial builds)
void launch_csc_append_row(const int* optr, const int* orow, const double* oval, int n_old, int new_row, const int* ncols,
                           const double* nvals, int kn, int* nptr, int* nrow, double* nval, hipStream_t st);
void launch_exclusive_scan(const int* in, int* out, long n, int* sums, hipStream_t st);  // sums: ceil(n / 4096) + 1 ints; sums[last] = total
void launch_band_count(const int* cptr, const int* crow, int N, int nbands, int* cnt, hipStream_t st);
// packed non-basic copy of the banded sweep: segment lengths by index of the pass (then an exclusive scan), then the copy
void launch_pack_count(const DevView& dv, int* cnt, hipStream_t st);
void launch_pack_fill(const DevView& dv, const int* pkptr, unsigned short* prow, double* pval, unsigned char* valid, hipStream_t st);
void launch_band_fill(const int* cptr, const int* crow, const double* cval, int N, int nbands, const int* bptr, unsigned short* brow,
                      double* bval, hipStream_t st);
void launch_build_colblk(const int* cptr, const int* crow, int N, int rb, int* colblk, hipStream_t st);

}  // namespace mlp
