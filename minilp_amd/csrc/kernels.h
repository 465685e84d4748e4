// kernels.h — device-side data layout and launch wrappers of the pivot hot path.
// All arrays live in HBM for the lifetime of a Solution; see DESIGN.md §3 for the layout.
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
};

// Per-iteration scalars, written by kernels, copied to the host once per pivot.
struct IterState {
    int status;
    int q;             // entering non-basic position (index into nb_* arrays)
    int r;             // leaving basic position / constraint row slot of x_B (-1: bound flip)
    int entering_var;  // nb_vars[q]
    int leaving_var;   // basic_vars[r]
    int sign;          // entering_diff_sign (solver.rs:743)
    int klist_n;       // FTRAN: number of (col-slot, coeff) pairs of a_q that hit nucleus rows
    int blist_n;       // BTRAN: number of (row-slot, coeff) pairs
    double pivot_coeff, leaving_new_val, entering_new_val, entering_diff;
    double entering_cur, entering_other, max_step, pivot_obj;
    double obj;       // cur_obj_val (solver.rs:51)
    double rho_sq;    // ||rho_r||^2      (solver.rs:1160)
    double alpha_sq;  // ||alpha_q||^2    (solver.rs:1136)
    double best_key;  // scratch
};

// Non-basic flags (solver.rs:66-70 NonBasicVarState + nb_var_is_fixed)
constexpr uint8_t NB_AT_MIN = 1, NB_AT_MAX = 2, NB_FIXED = 4;

// Everything the kernels need, passed by value.
struct DevView {
    int m, n;  // constraints (= basic positions), non-basic positions (= num_vars)
    int k;     // nucleus size
    int ld;    // leading dimension (= capacity) of W
    // constraint matrix, both orientations (solver.rs:21-22), int32 indices, f64 values
    const int* csc_ptr; const int* csc_row; const double* csc_val;  // N+1, nnz, nnz
    const int* csr_ptr; const int* csr_col; const double* csr_val;  // m+1, nnz, nnz
    const double* var_lo; const double* var_hi; const double* obj_c;  // N (orig_var_mins/maxs/obj_coeffs)
    int* var_loc;  // N: >=0 basic position, <0: -1-nb position   (solver.rs:33 var_states)
    // basic side, by position (solver.rs:37-41)
    int* basic_vars; double* xB; double* loB; double* hiB; double* beta;
    // non-basic side, by nb position (solver.rs:44-49)
    int* nb_vars; double* d; double* xN; double* gamma; uint8_t* nbflags;
    // basis inverse: singleton split + dense nucleus inverse W
    int* kslot_of_pos;     // m: row slot of W for a nucleus position, -1 for a singleton position
    int* srow_of_pos;      // m: the row of the single entry of a singleton basic column
    double* sdiag_of_pos;  // m: its value
    int* kslot_of_row;     // m: col slot of W for a nucleus row, -1 for a row covered by a singleton
    int* pos_of_srow;      // m: position of the singleton covering that row
    int* pos_of_kslot;     // cap: row slot -> position
    int* row_of_kslot;     // cap: col slot -> row
    double* W;             // cap x ld, row-major: W[rowslot(p)][colslot(i)] = (B^-1)[p, i]
    // per-pivot vectors
    double* alpha_q;  // m by position  (col_coeffs,            solver.rs:54)
    double* rho;      // m by row       (inv_basis_row_coeffs,  solver.rs:56)
    double* tau;      // m by position  (B^-1 rho,              solver.rs:1157)
    double* vvec;     // m by row       (B^-T alpha_q,          solver.rs:1114)
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
    double* part_tau;  // [ncolchunks][cap]
    double* part_v;    // [nrowstripes][cap]
    // reductions
    double* red_key; int* red_idx; unsigned* ticket;
    IterState* it;
};

// fused pass tiling
constexpr int FW_TR = 16;    // rows per block
constexpr int FW_TC = 1024;  // columns per block (256 threads x 4)

struct StructUpdate {  // how the (P_K, R_K) partition changes at this pivot
    int kase;       // 0: nuc->nuc, 1: sing->nuc (grow), 2: nuc->sing (shrink), 3: sing->sing, 4: sing->sing on the same row
    int r;          // leaving position
    int sr;         // row slot of r (cases 0, 2)
    int i_r;        // row of the leaving singleton (cases 1, 3)
    int i_q;        // row of the entering singleton (cases 2, 3)
    int cq;         // col slot of i_q (cases 2, 3)
    double diag_q;  // value of the entering singleton's entry
    double inv_diag_r;  // rho[i_r] = 1/diag of the leaving singleton
};

// ---- launch wrappers (all asynchronous on `st`) ----
void launch_price_primal(const DevView& v, int use_pse, hipStream_t st);
void launch_price_dual(const DevView& v, int use_dse, hipStream_t st);
void launch_ftran_col(const DevView& v, hipStream_t st);     // alpha_q = B^-1 a_{nb_vars[it->q]}
void launch_ratio_primal(const DevView& v, hipStream_t st);  // -> it->r / flip / unbounded
void launch_btran_unit(const DevView& v, hipStream_t st);    // rho = B^-T e_{it->r}, rK, rho_sq
void launch_sweep(const DevView& v, int with_helper, int only_helper, hipStream_t st);
void launch_ratio_dual(const DevView& v, hipStream_t st);    // -> it->q / infeasible
void launch_prep_v(const DevView& v, hipStream_t st);        // vvec[S rows], tK
void launch_fused_w(const DevView& v, int with_v, int do_update, int rslot, hipStream_t st);
void launch_finish_tau(const DevView& v, hipStream_t st);
void launch_finish_v(const DevView& v, hipStream_t st);
void launch_structure_update(const DevView& v, const StructUpdate& u, hipStream_t st);
void launch_update_pivot(const DevView& v, int use_dse, int use_pse, hipStream_t st);
void launch_update_flip(const DevView& v, hipStream_t st);
// dense transposed solve y = B^-T c with c given by position in alpha_q-like buffer `c_pos`; result by row in `y_row`
void launch_btran_dense(const DevView& v, const double* c_pos, double* y_row, hipStream_t st);
void launch_ftran_dense(const DevView& v, const double* b_row, double* x_pos, hipStream_t st);
void launch_gather_basic_obj(const DevView& v, double* c_pos, hipStream_t st);  // c_pos[p] = obj[basic_vars[p]]
void launch_recalc_d(const DevView& v, const double* y_row, hipStream_t st);  // d_c = obj - a_c.y ; it->obj
void launch_shift_nonbasic(const DevView& v, int col, double val, hipStream_t st);  // fix_var on a non-basic var
void launch_sq_norms_add_row(const DevView& v, hipStream_t st);  // gamma[c] += alpha_r[c]^2 (solver.rs:620-624)
// from-scratch inversion of the nucleus (Gauss-Jordan, partial pivoting) into Wout (k x ld); returns via *flag (device int) 0 ok / 1 singular
void launch_build_nucleus(const DevView& v, double* Kd, hipStream_t st);
void launch_gauss_jordan(double* Kd, double* Winv, int k, int ld, int* d_flag, double* d_scratch, hipStream_t st);
void launch_max_abs_diff(const double* A, const double* B, int k, int ld, double* d_out, hipStream_t st);

}  // namespace mlp
