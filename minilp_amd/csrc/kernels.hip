// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the simplex pivot hot path.
//
// Reference loops replaced (file:line in ztlpn/minilp 0.2.2) are cited per kernel.  None of this
// is GEMM-shaped: every kernel is an HBM/L2-bound stream with wave64 shuffle reductions, so there
// is no MFMA here (DESIGN.md §4 gives the algorithmic bytes per kernel).
#include "kernels.h"

#include <limits.h>
#include <math.h>

namespace mlp {

#define NONE_IDX INT_MAX
constexpr int BLK = 256;

// ------------------------------------------------------------------------------------ helpers
struct Cand {
    double key;
    int idx;
};
__device__ __forceinline__ Cand cand_none() { return Cand{-INFINITY, NONE_IDX}; }
// strict '>' with lowest index on ties: the reference's scans keep the first maximum
// (solver.rs:719, 727, 811, 878, 996).
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {
    return a.key > b.key || (a.key == b.key && a.idx < b.idx);
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand t;
        t.key = __shfl_down(c.key, o, 64);
        t.idx = __shfl_down(c.idx, o, 64);
        if (cand_better(t, c)) c = t;
    }
    return c;
}
__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_down(x, o, 64);
        if (t < x) x = t;
    }
    return x;
}
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}
__device__ __forceinline__ Cand block_best(Cand c) {  // result valid in thread 0
    __shared__ double s_key[BLK / 64];
    __shared__ int s_idx[BLK / 64];
    c = wave_best(c);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) {
        s_key[w] = c.key;
        s_idx[w] = c.idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < BLK / 64; ++i) {
            Cand t{s_key[i], s_idx[i]};
            if (cand_better(t, c)) c = t;
        }
    }
    return c;
}
__device__ __forceinline__ double block_min(double x) {  // thread 0
    __shared__ double s[BLK / 64];
    x = wave_min(x);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) s[w] = x;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < BLK / 64; ++i)
            if (s[i] < x) x = s[i];
    return x;
}
__device__ __forceinline__ double block_sum(double x) {  // thread 0, fixed tree => deterministic
    __shared__ double s[BLK / 64];
    x = wave_sum(x);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) s[w] = x;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < BLK / 64; ++i) x += s[i];
    return x;
}

// Cross-workgroup hand-off (cdna_hip_programming.md §6 G16): thread 0 has stored this block's
// partial; agent-scope release, ticket, and the last arriver does an agent-scope acquire that
// drops this CU's stale L1 lines before the block re-reads every partial.
__device__ __forceinline__ bool last_block_arrives(unsigned* ticket) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1);
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return s_last != 0;
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid-wide arg-best; returns true for every thread of the finalising block, result in thread 0
__device__ __forceinline__ bool grid_best(Cand& c, const DevView& v) {
    c = block_best(c);
    if (threadIdx.x == 0) {
        v.red_key[blockIdx.x] = c.key;
        v.red_idx[blockIdx.x] = c.idx;
    }
    if (!last_block_arrives(v.ticket)) return false;
    Cand x = cand_none();
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) {
        Cand t{ld_agent(&v.red_key[i]), ld_agent(&v.red_idx[i])};
        if (cand_better(t, x)) x = t;
    }
    c = block_best(x);
    if (threadIdx.x == 0) *v.ticket = 0;
    return true;
}
__device__ __forceinline__ bool grid_min(double& x, const DevView& v) {
    x = block_min(x);
    if (threadIdx.x == 0) v.red_key[blockIdx.x] = x;
    if (!last_block_arrives(v.ticket)) return false;
    double y = INFINITY;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) {
        double t = ld_agent(&v.red_key[i]);
        if (t < y) y = t;
    }
    x = block_min(y);
    if (threadIdx.x == 0) *v.ticket = 0;
    return true;
}
__device__ __forceinline__ bool grid_sum(double& x, const DevView& v) {
    x = block_sum(x);
    if (threadIdx.x == 0) v.red_key[blockIdx.x] = x;
    if (!last_block_arrives(v.ticket)) return false;
    double y = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) y += ld_agent(&v.red_key[i]);
    x = block_sum(y);
    if (threadIdx.x == 0) *v.ticket = 0;
    return true;
}

static inline int grid_for(int n, int per_thread = 4, int max_blocks = 512) {
    long b = ((long)n + (long)BLK * per_thread - 1) / ((long)BLK * per_thread);
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

template <int G>
__device__ __forceinline__ double group_sum(double x) {  // xor tree inside G consecutive lanes
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// ------------------------------------------------------------------- K1: primal pricing
// solver.rs:696-739: argmax over eligible non-basic columns of d^2/gamma (PSE) or |d| (Dantzig).
__global__ void __launch_bounds__(BLK) k_price_primal(DevView v, int use_pse) {
    Cand best = cand_none();
    for (int c = blockIdx.x * BLK + threadIdx.x; c < v.n; c += gridDim.x * BLK) {
        double dd = v.d[c];
        uint8_t f = v.nbflags[c];
        if (((f & NB_AT_MIN) && dd > -EPS) || ((f & NB_AT_MAX) && dd < EPS)) continue;  // solver.rs:705-706
        double score = use_pse ? dd * dd / v.gamma[c] : fabs(dd);
        Cand t{score, c};
        if (cand_better(t, best)) best = t;
    }
    if (!grid_best(best, v)) return;
    if (threadIdx.x == 0) {
        IterState* it = v.it;
        if (best.idx == NONE_IDX) {
            it->status = ITER_OPTIMAL;
            it->q = -1;
            it->r = -1;
        } else {
            int q = best.idx;
            int var = v.nb_vars[q];
            double dq = v.d[q];
            it->status = ITER_PIVOT;
            it->q = q;
            it->r = -1;
            it->entering_var = var;
            it->sign = dq < 0.0;  // solver.rs:743
            it->entering_cur = v.xN[q];
            it->entering_other = (dq < 0.0) ? v.var_hi[var] : v.var_lo[var];  // solver.rs:744-748
        }
        it->klist_n = 0;
        it->blist_n = 0;
    }
}

// ------------------------------------------------------------------- K6: dual pricing
// solver.rs:855-917: argmax over infeasible rows of infeas^2/beta.
__global__ void __launch_bounds__(BLK) k_price_dual(DevView v, int use_dse) {
    Cand best = cand_none();
    for (int r = blockIdx.x * BLK + threadIdx.x; r < v.m; r += gridDim.x * BLK) {
        double val = v.xB[r], mn = v.loB[r], mx = v.hiB[r];
        double infeas;
        if (val < mn - EPS) infeas = mn - val;
        else if (val > mx + EPS) infeas = val - mx;
        else continue;
        double score = use_dse ? infeas * infeas / v.beta[r] : infeas;
        Cand t{score, r};
        if (cand_better(t, best)) best = t;
    }
    if (!grid_best(best, v)) return;
    if (threadIdx.x == 0) {
        IterState* it = v.it;
        if (best.idx == NONE_IDX) {
            it->status = ITER_FEASIBLE;
            it->r = -1;
            it->q = -1;
        } else {
            int r = best.idx;
            double val = v.xB[r], mn = v.loB[r];
            it->status = ITER_PIVOT;
            it->r = r;
            it->q = -1;
            it->leaving_new_val = (val < mn) ? mn : v.hiB[r];  // solver.rs:908-914
            it->leaving_var = v.basic_vars[r];
        }
        it->klist_n = 0;
        it->blist_n = 0;
    }
}

// ------------------------------------------------------------------- K2: FTRAN of one column
// alpha_q = B^-1 a_q  (solver.rs:671-677 -> 1305-1319 -> lu.rs:79-106).  B^-1 is held as a
// singleton split + dense nucleus inverse W, so the solve is:
//   prep   : singleton rows of a_q land directly; entries on nucleus rows become a short list
//   gather : aK = W[:, list] * coeffs  (only the touched columns of W are read)
//            + push of -F*aK into the singleton positions (CSC columns of the nucleus basics)
__global__ void __launch_bounds__(64) k_ftran_prep(DevView v) {
    IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int lane = threadIdx.x;
    int var = it->entering_var;
    int base = v.csc_ptr[var], end = v.csc_ptr[var + 1];
    int cnt = 0;
    for (int e0 = base; e0 < end; e0 += 64) {
        int e = e0 + lane;
        bool valid = e < end;
        int s = -1;
        double a = 0.0;
        if (valid) {
            int i = v.csc_row[e];
            a = v.csc_val[e];
            s = v.kslot_of_row[i];
            if (s < 0) {
                int p = v.pos_of_srow[i];
                v.alpha_q[p] = a / v.sdiag_of_pos[p];
            }
        }
        bool isk = valid && s >= 0;
        unsigned long long mask = __ballot(isk);
        if (isk) {
            int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            v.klist_s[off] = s;
            v.klist_a[off] = a;
        }
        cnt += __popcll(mask);
    }
    if (lane == 0) it->klist_n = cnt;
}

// xK[slot] = sum_j list_a[j] * W[slot][list_s[j]]; out_pos[pos(slot)] = xK; then the F push.
template <int G>
__global__ void __launch_bounds__(BLK) k_ftran_gather(DevView v) {
    const IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (slot >= v.k) return;
    int n = it->klist_n;
    double acc = 0.0;
    const double* wrow = v.W + (size_t)slot * v.ld;
    for (int j = gl; j < n; j += G) acc += v.klist_a[j] * wrow[v.klist_s[j]];
    acc = group_sum<G>(acc);
    int p = v.pos_of_kslot[slot];
    if (gl == 0) {
        v.aK[slot] = acc;
        v.alpha_q[p] = acc;
    }
    if (acc != 0.0) {
        int var = v.basic_vars[p];
        int end = v.csc_ptr[var + 1];
        for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
            int i = v.csc_row[e];
            if (v.kslot_of_row[i] < 0) {
                int ps = v.pos_of_srow[i];
                unsafeAtomicAdd(&v.alpha_q[ps], -v.csc_val[e] * acc / v.sdiag_of_pos[ps]);
            }
        }
    }
}

// Generic finish of a dense FTRAN: xK (by row slot) -> out (by position) incl. the F push.
__global__ void __launch_bounds__(BLK) k_ftran_init_single(DevView v, const double* b_row, double* out_pos) {
    if (v.it->status != ITER_PIVOT) return;
    int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= v.m) return;
    if (v.kslot_of_pos[p] < 0) out_pos[p] = b_row[v.srow_of_pos[p]] / v.sdiag_of_pos[p];
}
template <int G>
__global__ void __launch_bounds__(BLK) k_ftran_push(DevView v, const double* xK, double* out_pos) {
    if (v.it->status != ITER_PIVOT) return;
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (slot >= v.k) return;
    double x = xK[slot];
    int p = v.pos_of_kslot[slot];
    if (gl == 0) out_pos[p] = x;
    if (x != 0.0) {
        int var = v.basic_vars[p];
        int end = v.csc_ptr[var + 1];
        for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
            int i = v.csc_row[e];
            if (v.kslot_of_row[i] < 0) {
                int ps = v.pos_of_srow[i];
                unsafeAtomicAdd(&out_pos[ps], -v.csc_val[e] * x / v.sdiag_of_pos[ps]);
            }
        }
    }
}

// ------------------------------------------------------------------- K5: primal Harris ratio test
// solver.rs:752-771 get_leaving_var_step
__device__ __forceinline__ double leaving_step(const DevView& v, int p, double coeff, int sign, bool& to_max) {
    double val = v.xB[p];
    to_max = (sign && coeff < 0.0) || (!sign && coeff > 0.0);
    if (to_max) {
        double mx = v.hiB[p];
        return val < mx ? mx - val : 0.0;
    } else {
        double mn = v.loB[p];
        return val > mn ? val - mn : 0.0;
    }
}
// pass 1 (solver.rs:782-795)
__global__ void __launch_bounds__(BLK) k_ratio_primal_p1(DevView v) {
    IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int sign = it->sign;
    double mn = INFINITY;
    for (int p = blockIdx.x * BLK + threadIdx.x; p < v.m; p += gridDim.x * BLK) {
        double coeff = v.alpha_q[p];
        double ca = fabs(coeff);
        if (ca < EPS) continue;
        bool tm;
        double cur = (leaving_step(v, p, coeff, sign, tm) + EPS) / ca;
        if (cur < mn) mn = cur;
    }
    if (!grid_min(mn, v)) return;
    if (threadIdx.x == 0) {
        double max_step = fabs(it->entering_other - it->entering_cur);
        if (mn < max_step) max_step = mn;
        it->max_step = max_step;
    }
}
// pass 2 (solver.rs:800-853)
__global__ void __launch_bounds__(BLK) k_ratio_primal_p2(DevView v) {
    IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int sign = it->sign;
    double max_step = it->max_step;
    Cand best = cand_none();
    for (int p = blockIdx.x * BLK + threadIdx.x; p < v.m; p += gridDim.x * BLK) {
        double coeff = v.alpha_q[p];
        double ca = fabs(coeff);
        if (ca < EPS) continue;
        bool tm;
        double cur = leaving_step(v, p, coeff, sign, tm) / ca;
        if (cur <= max_step) {
            Cand t{ca, p};
            if (cand_better(t, best)) best = t;
        }
    }
    if (!grid_best(best, v)) return;
    if (threadIdx.x == 0) {
        int q = it->q;
        double dq = v.d[q];
        if (best.idx != NONE_IDX) {
            int r = best.idx;
            double coeff = v.alpha_q[r];
            bool tm;
            leaving_step(v, r, coeff, sign, tm);
            double lnv = tm ? v.hiB[r] : v.loB[r];
            double diff = (v.xB[r] - lnv) / coeff;  // solver.rs:828
            it->r = r;
            it->pivot_coeff = coeff;
            it->leaving_new_val = lnv;
            it->entering_diff = diff;
            it->entering_new_val = it->entering_cur + diff;
            it->leaving_var = v.basic_vars[r];
            it->pivot_obj = dq / coeff;  // solver.rs:1073
            it->obj += dq * diff;        // solver.rs:1027
            it->status = ITER_PIVOT;
        } else if (isinf(it->entering_other)) {
            it->status = ITER_UNBOUNDED;  // solver.rs:842-844
        } else {
            double diff = it->entering_other - it->entering_cur;  // solver.rs:846-851
            it->r = -1;
            it->entering_new_val = it->entering_other;
            it->entering_diff = diff;
            it->obj += dq * diff;
            it->status = ITER_FLIP;
        }
    }
}

// ------------------------------------------------------------------- K3: BTRAN of a unit vector
// rho = B^-T e_r (solver.rs:680-683 -> 1322-1338).  With W explicit this is one row of W when r
// is a nucleus position, or a short combination of rows (those nucleus columns that have an
// entry in the leaving singleton's row, read from the CSR row) otherwise.
__global__ void __launch_bounds__(64) k_btran_prep(DevView v) {
    IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int lane = threadIdx.x;
    int r = it->r;
    int sr = v.kslot_of_pos[r];
    if (sr >= 0) {
        if (lane == 0) {
            v.blist_s[0] = sr;
            v.blist_a[0] = 1.0;
            it->blist_n = 1;
        }
        return;
    }
    int i_r = v.srow_of_pos[r];
    double inv = 1.0 / v.sdiag_of_pos[r];
    if (lane == 0) v.rho[i_r] = inv;
    int base = v.csr_ptr[i_r], end = v.csr_ptr[i_r + 1];
    int cnt = 0;
    for (int e0 = base; e0 < end; e0 += 64) {
        int e = e0 + lane;
        bool valid = e < end;
        int s = -1;
        double a = 0.0;
        if (valid) {
            int loc = v.var_loc[v.csr_col[e]];
            a = v.csr_val[e];
            if (loc >= 0) s = v.kslot_of_pos[loc];
        }
        bool isk = valid && s >= 0;
        unsigned long long mask = __ballot(isk);
        if (isk) {
            int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            v.blist_s[off] = s;
            v.blist_a[off] = -a * inv;
        }
        cnt += __popcll(mask);
    }
    if (lane == 0) it->blist_n = cnt;
}
__global__ void __launch_bounds__(BLK) k_btran_gather(DevView v) {
    const IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int s = blockIdx.x * BLK + threadIdx.x;
    if (s >= v.k) return;
    int n = it->blist_n;
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc += v.blist_a[j] * v.W[(size_t)v.blist_s[j] * v.ld + s];
    v.rK[s] = acc;
    v.rho[v.row_of_kslot[s]] = acc;
}
// ||x||^2 over n entries -> *out (fixed reduction tree)
__global__ void __launch_bounds__(BLK) k_sqnorm(DevView v, const double* x, int n, double* out, int add_one) {
    if (v.it->status != ITER_PIVOT) return;
    double s = 0.0;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) s += x[i] * x[i];
    if (!grid_sum(s, v)) return;
    if (threadIdx.x == 0) *out = s + (add_one ? 1.0 : 0.0);
}

// ------------------------------------------------------------------- K4: tableau row  rho^T N
// solver.rs:685-692 (and 1117-1132 for the PSE helper).  The reference pushes rows of supp(rho)
// through the CSR; here every non-basic column PULLS its dot product from the CSC: no atomics,
// fixed summation order, one streaming pass over A that yields alpha_r and (PSE) N^T v together.
template <int G, int MODE>  // MODE 0: alpha_r only, 1: alpha_r + helper, 2: helper only
__global__ void __launch_bounds__(BLK) k_sweep(DevView v) {
    if (v.it->status != ITER_PIVOT) return;
    int c = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (c >= v.n) return;
    int var = v.nb_vars[c];
    int end = v.csc_ptr[var + 1];
    double a1 = 0.0, a2 = 0.0;
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
        int i = v.csc_row[e];
        double a = v.csc_val[e];
        if (MODE != 2) a1 += a * v.rho[i];
        if (MODE != 0) a2 += a * v.vvec[i];
    }
    if (MODE != 2) a1 = group_sum<G>(a1);
    if (MODE != 0) a2 = group_sum<G>(a2);
    if (gl == 0) {
        if (MODE != 2) v.alpha_r[c] = a1;
        if (MODE != 0) v.helper[c] = a2;
    }
}

// ------------------------------------------------------------------- K7: dual Harris ratio test
__device__ __forceinline__ bool dual_eligible(double coeff, uint8_t f, int lsign) {  // solver.rs:937-951
    int esign;
    if (coeff >= EPS) esign = !lsign;
    else if (coeff <= -EPS) esign = lsign;
    else return false;
    return esign ? !(f & NB_AT_MAX) : !(f & NB_AT_MIN);
}
__device__ __forceinline__ double clamp_obj(double d, uint8_t f) {  // solver.rs:927-935
    if ((f & NB_AT_MIN) && d < 0.0) d = 0.0;
    if ((f & NB_AT_MAX) && d > 0.0) d = 0.0;
    return d;
}
__global__ void __launch_bounds__(BLK) k_ratio_dual_p1(DevView v) {  // solver.rs:962-974
    IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int lsign = it->leaving_new_val > v.xB[it->r];
    double mn = INFINITY;
    for (int c = blockIdx.x * BLK + threadIdx.x; c < v.n; c += gridDim.x * BLK) {
        double coeff = v.alpha_r[c];
        uint8_t f = v.nbflags[c];
        if (!dual_eligible(coeff, f, lsign)) continue;
        double cur = (fabs(clamp_obj(v.d[c], f)) + EPS) / fabs(coeff);
        if (cur < mn) mn = cur;
    }
    if (!grid_min(mn, v)) return;
    if (threadIdx.x == 0) it->max_step = mn;
}
__global__ void __launch_bounds__(BLK) k_ratio_dual_p2(DevView v) {  // solver.rs:979-1021
    IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int r = it->r;
    int lsign = it->leaving_new_val > v.xB[r];
    double max_step = it->max_step;
    Cand best = cand_none();
    for (int c = blockIdx.x * BLK + threadIdx.x; c < v.n; c += gridDim.x * BLK) {
        double coeff = v.alpha_r[c];
        uint8_t f = v.nbflags[c];
        if (!dual_eligible(coeff, f, lsign)) continue;
        double cur = fabs(clamp_obj(v.d[c], f)) / fabs(coeff);
        if (cur <= max_step) {
            Cand t{fabs(coeff), c};
            if (cand_better(t, best)) best = t;
        }
    }
    if (!grid_best(best, v)) return;
    if (threadIdx.x == 0) {
        if (best.idx == NONE_IDX) {
            it->status = ITER_INFEASIBLE;
            return;
        }
        int q = best.idx;
        double coeff = v.alpha_r[q];
        double dq = v.d[q];
        double diff = (v.xB[r] - it->leaving_new_val) / coeff;  // solver.rs:1005
        it->q = q;
        it->entering_var = v.nb_vars[q];
        it->pivot_coeff = coeff;
        it->entering_diff = diff;
        it->entering_cur = v.xN[q];
        it->entering_new_val = v.xN[q] + diff;
        it->pivot_obj = dq / coeff;
        it->obj += dq * diff;
        it->leaving_var = v.basic_vars[r];
    }
}

// ------------------------------------------------------------------- v = B^-T alpha_q, stage 1
// (solver.rs:1114).  y_S on singleton rows, then the rhs tK of the transposed nucleus solve.
__global__ void __launch_bounds__(BLK) k_btran_single(DevView v, const double* c_pos, double* y_row) {
    if (v.it->status != ITER_PIVOT) return;
    int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= v.m) return;
    if (v.kslot_of_pos[p] < 0) y_row[v.srow_of_pos[p]] = c_pos[p] / v.sdiag_of_pos[p];
}
template <int G>
__global__ void __launch_bounds__(BLK) k_btran_rhs(DevView v, const double* c_pos, const double* y_row) {
    if (v.it->status != ITER_PIVOT) return;
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (slot >= v.k) return;
    int p = v.pos_of_kslot[slot];
    int var = v.basic_vars[p];
    int end = v.csc_ptr[var + 1];
    double acc = 0.0;
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
        int i = v.csc_row[e];
        if (v.kslot_of_row[i] < 0) acc += v.csc_val[e] * y_row[i];
    }
    acc = group_sum<G>(acc);
    if (gl == 0) v.tK[slot] = c_pos[p] - acc;
}
__global__ void __launch_bounds__(BLK) k_scatter_cols(DevView v, const double* xK, double* y_row) {
    if (v.it->status != ITER_PIVOT) return;
    int s = blockIdx.x * BLK + threadIdx.x;
    if (s < v.k) y_row[v.row_of_kslot[s]] = xK[s];
}

// ------------------------------------------------------------------- fused pass over W
// One read + one write of the dense nucleus inverse per pivot does three things at once:
//   tauK = W * rK          (FTRAN #2 for dual steepest edge, solver.rs:1157)
//   vK   = W^T * tK        (BTRAN #2 for primal steepest edge, solver.rs:1114)
//   W   -= (aK - e_r) rK^T / alpha_r   (the eta transformation of solver.rs:1274-1284 applied
//                                       eagerly: B'^-1 = E B^-1)
// Block = FW_TR rows x FW_TC columns; per-block partials are reduced by k_fused_reduce in a fixed
// order (no float atomics => bitwise reproducible).
template <bool WITH_TAU, bool WITH_V, bool DO_UPDATE>
__global__ void __launch_bounds__(BLK) k_fused_w(DevView v, int rslot, double inv_alpha_override) {
    if (v.it->status != ITER_PIVOT) return;
    __shared__ double s_tau[FW_TR][BLK / 64];
    const int k = v.k, ld = v.ld;
    const int row0 = blockIdx.x * FW_TR;
    const int col0 = blockIdx.y * FW_TC;
    const int tid = threadIdx.x;
    int cidx[4];
    cidx[0] = col0 + 2 * tid;
    cidx[1] = cidx[0] + 1;
    cidx[2] = col0 + 512 + 2 * tid;
    cidx[3] = cidx[2] + 1;
    double rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rk[j] = (cidx[j] < k) ? v.rK[cidx[j]] : 0.0;
    double inv_alpha = 0.0;
    if (DO_UPDATE) inv_alpha = (inv_alpha_override != 0.0) ? inv_alpha_override : 1.0 / v.alpha_q[v.it->r];
    double vacc[4] = {0.0, 0.0, 0.0, 0.0};
    double tacc[FW_TR];
    const bool pair0 = cidx[1] < k, pair1 = cidx[3] < k;
    const bool one0 = cidx[0] < k, one1 = cidx[2] < k;
#pragma unroll
    for (int a = 0; a < FW_TR; ++a) {
        int row = row0 + a;
        tacc[a] = 0.0;
        if (row >= k) continue;
        double* wp = v.W + (size_t)row * ld;
        double w[4] = {0.0, 0.0, 0.0, 0.0};
        if (pair0) {
            double2 t = *reinterpret_cast<const double2*>(wp + cidx[0]);
            w[0] = t.x;
            w[1] = t.y;
        } else if (one0) {
            w[0] = wp[cidx[0]];
        }
        if (pair1) {
            double2 t = *reinterpret_cast<const double2*>(wp + cidx[2]);
            w[2] = t.x;
            w[3] = t.y;
        } else if (one1) {
            w[2] = wp[cidx[2]];
        }
        if (WITH_TAU) tacc[a] = w[0] * rk[0] + w[1] * rk[1] + w[2] * rk[2] + w[3] * rk[3];
        if (WITH_V) {
            double t = v.tK[row];
#pragma unroll
            for (int j = 0; j < 4; ++j) vacc[j] += w[j] * t;
        }
        if (DO_UPDATE) {
            double u = (v.aK[row] - (row == rslot ? 1.0 : 0.0)) * inv_alpha;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] -= u * rk[j];
            if (pair0) *reinterpret_cast<double2*>(wp + cidx[0]) = make_double2(w[0], w[1]);
            else if (one0) wp[cidx[0]] = w[0];
            if (pair1) *reinterpret_cast<double2*>(wp + cidx[2]) = make_double2(w[2], w[3]);
            else if (one1) wp[cidx[2]] = w[2];
        }
    }
    if (WITH_V) {
        double* pv = v.part_v + (size_t)blockIdx.x * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (cidx[j] < k) pv[cidx[j]] = vacc[j];
    }
    if (WITH_TAU) {
        int wv = tid >> 6, l = tid & 63;
#pragma unroll
        for (int a = 0; a < FW_TR; ++a) {
            double s = wave_sum(tacc[a]);
            if (l == 0) s_tau[a][wv] = s;
        }
        __syncthreads();
        if (tid < FW_TR) {
            int row = row0 + tid;
            if (row < k) {
                double s = s_tau[tid][0];
                for (int i = 1; i < BLK / 64; ++i) s += s_tau[tid][i];
                v.part_tau[(size_t)blockIdx.y * ld + row] = s;
            }
        }
    }
}
__global__ void __launch_bounds__(BLK) k_fused_reduce(DevView v, int nstripes, int nchunks, int with_tau, int with_v) {
    if (v.it->status != ITER_PIVOT) return;
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= v.k) return;
    if (with_tau) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += v.part_tau[(size_t)c * v.ld + i];
        v.tauK[i] = s;
    }
    if (with_v) {
        double s = 0.0;
        for (int t = 0; t < nstripes; ++t) s += v.part_v[(size_t)t * v.ld + i];
        v.vK[i] = s;
    }
}

// ------------------------------------------------------------------- partition change
// B'^-1[p,i] = B^-1[p,i] - (alpha_p - [p==r]) rho_i / alpha_r, restricted to the new nucleus.
__global__ void __launch_bounds__(BLK) k_struct_grow(DevView v, StructUpdate u) {  // case 1: sing -> nuc
    if (v.it->status != ITER_PIVOT) return;
    int s = blockIdx.x * BLK + threadIdx.x;
    int kold = v.k;  // view still carries the old k
    double inv_alpha = 1.0 / v.alpha_q[u.r];
    if (s < kold) {
        v.W[(size_t)kold * v.ld + s] = v.rK[s] * inv_alpha;                    // new row: rho / alpha_r
        v.W[(size_t)s * v.ld + kold] = -v.aK[s] * u.inv_diag_r * inv_alpha;    // new column
    } else if (s == kold) {
        v.W[(size_t)kold * v.ld + kold] = u.inv_diag_r * inv_alpha;
        v.kslot_of_pos[u.r] = kold;
        v.pos_of_kslot[kold] = u.r;
        v.kslot_of_row[u.i_r] = kold;
        v.row_of_kslot[kold] = u.i_r;
    }
}
__global__ void __launch_bounds__(BLK) k_struct_newcol(DevView v, StructUpdate u) {  // case 3: sing -> sing
    if (v.it->status != ITER_PIVOT) return;
    int s = blockIdx.x * BLK + threadIdx.x;
    double inv_alpha = 1.0 / v.alpha_q[u.r];
    if (s < v.k) v.W[(size_t)s * v.ld + u.cq] = -v.aK[s] * u.inv_diag_r * inv_alpha;
    if (s == 0) {
        v.srow_of_pos[u.r] = u.i_q;
        v.sdiag_of_pos[u.r] = u.diag_q;
        v.kslot_of_row[u.i_q] = -1;
        v.pos_of_srow[u.i_q] = u.r;
        v.kslot_of_row[u.i_r] = u.cq;
        v.row_of_kslot[u.cq] = u.i_r;
    }
}
// case 4: a singleton replaces the singleton of the same row: only the diagonal entry changes
__global__ void k_struct_newdiag(DevView v, StructUpdate u) {
    if (v.it->status != ITER_PIVOT) return;
    v.sdiag_of_pos[u.r] = u.diag_q;
}
// case 2: nuc -> sing.  Drop row slot sr and col slot cq; keep slots compact by moving the last in.
__global__ void __launch_bounds__(BLK) k_struct_shrink_row(DevView v, StructUpdate u) {
    if (v.it->status != ITER_PIVOT) return;
    int s = blockIdx.x * BLK + threadIdx.x;
    int last = v.k - 1;
    if (s < v.k && u.sr != last) v.W[(size_t)u.sr * v.ld + s] = v.W[(size_t)last * v.ld + s];
    if (s == 0) {
        if (u.sr != last) {
            int pl = v.pos_of_kslot[last];
            v.pos_of_kslot[u.sr] = pl;
            v.kslot_of_pos[pl] = u.sr;
        }
        v.kslot_of_pos[u.r] = -1;
        v.srow_of_pos[u.r] = u.i_q;
        v.sdiag_of_pos[u.r] = u.diag_q;
    }
}
__global__ void __launch_bounds__(BLK) k_struct_shrink_col(DevView v, StructUpdate u) {
    if (v.it->status != ITER_PIVOT) return;
    int s = blockIdx.x * BLK + threadIdx.x;
    int last = v.k - 1;
    if (s < last && u.cq != last) v.W[(size_t)s * v.ld + u.cq] = v.W[(size_t)s * v.ld + last];
    if (s == 0) {
        if (u.cq != last) {
            int il = v.row_of_kslot[last];
            v.row_of_kslot[u.cq] = il;
            v.kslot_of_row[il] = u.cq;
        }
        v.kslot_of_row[u.i_q] = -1;
        v.pos_of_srow[u.i_q] = u.r;
    }
}

// ------------------------------------------------------------------- K8: updates after the pivot
// basic side: solver.rs:1049-1058 (x_B, bounds), 1164-1173 (dual steepest-edge norms)
__global__ void __launch_bounds__(BLK) k_update_basic(DevView v, int use_dse) {
    const IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int p = blockIdx.x * BLK + threadIdx.x;
    if (p >= v.m) return;
    int r = it->r;
    double pc = it->pivot_coeff;
    double a = v.alpha_q[p];
    if (p == r) {
        int ev = it->entering_var;
        v.xB[r] = it->entering_new_val;
        v.loB[r] = v.var_lo[ev];
        v.hiB[r] = v.var_hi[ev];
        if (use_dse) v.beta[r] = it->rho_sq / (pc * pc);
        v.basic_vars[r] = ev;
        v.var_loc[ev] = r;
        v.var_loc[it->leaving_var] = -1 - it->q;
    } else if (a != 0.0) {
        v.xB[p] -= it->entering_diff * a;
        if (use_dse) v.beta[p] += -2.0 * a * v.tau[p] / pc + it->rho_sq * a * a / (pc * pc);
    }
}
// non-basic side: solver.rs:1068-1080 (value/state of the leaving var, reduced costs), 1140-1150 (PSE)
__global__ void __launch_bounds__(BLK) k_update_nonbasic(DevView v, int use_pse) {
    const IterState* it = v.it;
    if (it->status != ITER_PIVOT) return;
    int c = blockIdx.x * BLK + threadIdx.x;
    if (c >= v.n) return;
    int q = it->q;
    double pc = it->pivot_coeff;
    if (c == q) {
        int lv = it->leaving_var;
        double lnv = it->leaving_new_val;
        v.d[q] = -it->pivot_obj;
        if (use_pse) v.gamma[q] = it->alpha_sq / (pc * pc);
        v.nb_vars[q] = lv;
        v.xN[q] = lnv;
        v.nbflags[q] = (uint8_t)((lnv == v.var_lo[lv] ? NB_AT_MIN : 0) | (lnv == v.var_hi[lv] ? NB_AT_MAX : 0));
    } else {
        double ar = v.alpha_r[c];
        if (ar != 0.0) {
            v.d[c] -= it->pivot_obj * ar;
            if (use_pse) v.gamma[c] += -2.0 * ar * v.helper[c] / pc + it->alpha_sq * ar * ar / (pc * pc);
        }
    }
}
// bound flip: solver.rs:1031-1042
__global__ void __launch_bounds__(BLK) k_update_flip(DevView v) {
    const IterState* it = v.it;
    if (it->status != ITER_FLIP) return;
    int p = blockIdx.x * BLK + threadIdx.x;
    if (p < v.m) {
        double a = v.alpha_q[p];
        if (a != 0.0) v.xB[p] -= it->entering_diff * a;
    }
    if (p == 0) {
        int q = it->q, ev = it->entering_var;
        double nv = it->entering_new_val;
        v.xN[q] = nv;
        v.nbflags[q] = (uint8_t)((v.nbflags[q] & NB_FIXED) | (nv == v.var_lo[ev] ? NB_AT_MIN : 0) |
                                 (nv == v.var_hi[ev] ? NB_AT_MAX : 0));
    }
}

// ------------------------------------------------------------------- K9: recalc reduced costs
// solver.rs:1216-1231: d_c = c_c - a_c . y for every non-basic c, then the objective from scratch.
template <int G>
__global__ void __launch_bounds__(BLK) k_recalc_d(DevView v, const double* y_row) {
    int c = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (c >= v.n) return;
    int var = v.nb_vars[c];
    int end = v.csc_ptr[var + 1];
    double acc = 0.0;
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) acc += v.csc_val[e] * y_row[v.csc_row[e]];
    acc = group_sum<G>(acc);
    if (gl == 0) v.d[c] = v.obj_c[var] - acc;
}
__global__ void __launch_bounds__(BLK) k_recalc_obj(DevView v) {
    double s = 0.0;
    for (int p = blockIdx.x * BLK + threadIdx.x; p < v.m; p += gridDim.x * BLK) s += v.obj_c[v.basic_vars[p]] * v.xB[p];
    for (int c = blockIdx.x * BLK + threadIdx.x; c < v.n; c += gridDim.x * BLK) s += v.obj_c[v.nb_vars[c]] * v.xN[c];
    if (!grid_sum(s, v)) return;
    if (threadIdx.x == 0) v.it->obj = s;
}
__global__ void k_set_status(DevView v, int status) { v.it->status = status; }
__global__ void k_gather_basic_obj(DevView v, double* c_pos) {
    int p = blockIdx.x * BLK + threadIdx.x;
    if (p < v.m) c_pos[p] = v.obj_c[v.basic_vars[p]];
}

// fix_var on a non-basic variable (solver.rs:393-404): x_B -= diff * alpha_q, obj += diff * d, x_N = val
__global__ void __launch_bounds__(BLK) k_shift_nonbasic(DevView v, int col, double val) {
    int p = blockIdx.x * BLK + threadIdx.x;
    double diff = val - v.xN[col];
    if (p < v.m) {
        double a = v.alpha_q[p];
        if (a != 0.0) v.xB[p] -= diff * a;
    }
    __syncthreads();
    if (p == 0) v.it->obj += diff * v.d[col];
}
__global__ void k_set_xn(DevView v, int col, double val) { v.xN[col] = val; }
// add_constraint (solver.rs:620-624): gamma[c] += alpha_r[c]^2
__global__ void __launch_bounds__(BLK) k_gamma_add_row(DevView v) {
    int c = blockIdx.x * BLK + threadIdx.x;
    if (c < v.n) {
        double a = v.alpha_r[c];
        v.gamma[c] += a * a;
    }
}

// ------------------------------------------------------------------- from-scratch inversion
// Counterpart of BasisSolver::reset (solver.rs:1286-1303): rebuild the nucleus K from the CSC
// columns of the basic variables and invert it by Gauss-Jordan with partial pivoting.
template <int G>
__global__ void __launch_bounds__(BLK) k_build_nucleus(DevView v, double* Kd) {
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;  // row slot <-> position (column of K)
    int gl = threadIdx.x & (G - 1);
    if (slot >= v.k) return;
    int var = v.basic_vars[v.pos_of_kslot[slot]];
    int end = v.csc_ptr[var + 1];
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
        int a = v.kslot_of_row[v.csc_row[e]];
        if (a >= 0) Kd[(size_t)a * v.ld + slot] = v.csc_val[e];
    }
}
__global__ void __launch_bounds__(BLK) k_set_identity(double* Wv, int k, int ld) {
    size_t i = (size_t)blockIdx.x * BLK + threadIdx.x;
    if (i < (size_t)k * k) {
        int r = (int)(i / k), c = (int)(i % k);
        Wv[(size_t)r * ld + c] = (r == c) ? 1.0 : 0.0;
    }
}
// one block: pivot search in column j (rows >= j); scratch[0] = pivot row, flag=1 if singular
__global__ void __launch_bounds__(BLK) k_gj_pivot(const double* Kd, int k, int ld, int j, int* piv_row, int* flag) {
    Cand best = cand_none();
    for (int a = j + threadIdx.x; a < k; a += BLK) {
        Cand t{fabs(Kd[(size_t)a * ld + j]), a};
        if (cand_better(t, best)) best = t;
    }
    best = block_best(best);
    if (threadIdx.x == 0) {
        *piv_row = best.idx == NONE_IDX ? j : best.idx;
        if (!(best.key >= 1e-11)) *flag = 1;
    }
}
// elimination factors of step j: Kd[a][j] as it will be after the row swap (0 for the pivot row)
__global__ void __launch_bounds__(BLK) k_gj_factors(const double* Kd, int k, int ld, int j, const int* piv_row, double* factors) {
    int a = blockIdx.x * BLK + threadIdx.x;
    if (a >= k) return;
    int pr = *piv_row;
    double f;
    if (a == j) f = 0.0;
    else if (a == pr) f = Kd[(size_t)j * ld + j];
    else f = Kd[(size_t)a * ld + j];
    factors[a] = f;
    if (a == 0) factors[k] = 1.0 / Kd[(size_t)pr * ld + j];
}
// swap rows j <-> piv in both matrices and scale the new row j by 1/pivot
__global__ void __launch_bounds__(BLK) k_gj_swap_scale(double* Kd, double* Wv, int k, int ld, int j, const int* piv_row,
                                                     const double* factors) {
    int c = blockIdx.x * BLK + threadIdx.x;
    if (c >= k) return;
    int pr = *piv_row;
    double inv = factors[k];
    double a = Kd[(size_t)pr * ld + c], b = Kd[(size_t)j * ld + c];
    double x = Wv[(size_t)pr * ld + c], y = Wv[(size_t)j * ld + c];
    if (pr != j) {
        Kd[(size_t)pr * ld + c] = b;
        Wv[(size_t)pr * ld + c] = y;
    }
    Kd[(size_t)j * ld + c] = a * inv;
    Wv[(size_t)j * ld + c] = x * inv;
}
__global__ void __launch_bounds__(BLK) k_gj_eliminate(double* Kd, double* Wv, int k, int ld, int j, const double* factors) {
    int c = blockIdx.x * BLK + threadIdx.x;
    int a = blockIdx.y;
    if (c >= k || a == j) return;
    double f = factors[a];
    if (f == 0.0) return;
    Kd[(size_t)a * ld + c] -= f * Kd[(size_t)j * ld + c];
    Wv[(size_t)a * ld + c] -= f * Wv[(size_t)j * ld + c];
}
__global__ void __launch_bounds__(BLK) k_max_abs_diff(const double* A, const double* B, int k, int ld, double* out) {
    double mx = 0.0;
    size_t tot = (size_t)k * k;
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < tot; i += (size_t)gridDim.x * BLK) {
        int r = (int)(i / k), c = (int)(i % k);
        double dlt = fabs(A[(size_t)r * ld + c] - B[(size_t)r * ld + c]);
        if (dlt != dlt) dlt = INFINITY;
        if (dlt > mx) mx = dlt;
    }
    // max via atomics on the bit pattern (non-negative doubles order like uint64)
    mx = -wave_min(-mx);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(mx));
}

// ===================================================================================== launchers
static inline int blocks_for(int n, int per_block = BLK) { return n <= 0 ? 1 : (n + per_block - 1) / per_block; }

void launch_price_primal(const DevView& v, int use_pse, hipStream_t st) {
    hipLaunchKernelGGL(k_price_primal, dim3(grid_for(v.n)), dim3(BLK), 0, st, v, use_pse);
}
void launch_price_dual(const DevView& v, int use_dse, hipStream_t st) {
    hipLaunchKernelGGL(k_price_dual, dim3(grid_for(v.m)), dim3(BLK), 0, st, v, use_dse);
}
void launch_ftran_col(const DevView& v, hipStream_t st) {
    (void)hipMemsetAsync(v.alpha_q, 0, sizeof(double) * (size_t)v.m, st);
    hipLaunchKernelGGL(k_ftran_prep, dim3(1), dim3(64), 0, st, v);
    if (v.k > 0) hipLaunchKernelGGL(k_ftran_gather<16>, dim3(blocks_for(v.k * 16)), dim3(BLK), 0, st, v);
}
void launch_ratio_primal(const DevView& v, hipStream_t st) {
    hipLaunchKernelGGL(k_ratio_primal_p1, dim3(grid_for(v.m)), dim3(BLK), 0, st, v);
    hipLaunchKernelGGL(k_ratio_primal_p2, dim3(grid_for(v.m)), dim3(BLK), 0, st, v);
}
void launch_btran_unit(const DevView& v, hipStream_t st) {
    (void)hipMemsetAsync(v.rho, 0, sizeof(double) * (size_t)v.m, st);
    hipLaunchKernelGGL(k_btran_prep, dim3(1), dim3(64), 0, st, v);
    if (v.k > 0) hipLaunchKernelGGL(k_btran_gather, dim3(blocks_for(v.k)), dim3(BLK), 0, st, v);
    hipLaunchKernelGGL(k_sqnorm, dim3(grid_for(v.m)), dim3(BLK), 0, st, v, (const double*)v.rho, v.m, &v.it->rho_sq, 0);
}
void launch_sweep(const DevView& v, int with_helper, int only_helper, hipStream_t st) {
    dim3 g(blocks_for(v.n * 16)), b(BLK);
    if (only_helper) hipLaunchKernelGGL((k_sweep<16, 2>), g, b, 0, st, v);
    else if (with_helper) hipLaunchKernelGGL((k_sweep<16, 1>), g, b, 0, st, v);
    else hipLaunchKernelGGL((k_sweep<16, 0>), g, b, 0, st, v);
}
void launch_ratio_dual(const DevView& v, hipStream_t st) {
    hipLaunchKernelGGL(k_ratio_dual_p1, dim3(grid_for(v.n)), dim3(BLK), 0, st, v);
    hipLaunchKernelGGL(k_ratio_dual_p2, dim3(grid_for(v.n)), dim3(BLK), 0, st, v);
}
void launch_prep_v(const DevView& v, hipStream_t st) {
    hipLaunchKernelGGL(k_sqnorm, dim3(grid_for(v.m)), dim3(BLK), 0, st, v, (const double*)v.alpha_q, v.m, &v.it->alpha_sq, 1);
    hipLaunchKernelGGL(k_btran_single, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v, (const double*)v.alpha_q, v.vvec);
    if (v.k > 0) hipLaunchKernelGGL(k_btran_rhs<16>, dim3(blocks_for(v.k * 16)), dim3(BLK), 0, st, v, (const double*)v.alpha_q, (const double*)v.vvec);
}
void launch_fused_w(const DevView& v, int with_v, int do_update, int rslot, hipStream_t st) {
    if (v.k <= 0) return;
    int nstripes = (v.k + FW_TR - 1) / FW_TR, nchunks = (v.k + FW_TC - 1) / FW_TC;
    dim3 g(nstripes, nchunks), b(BLK);
    if (do_update) {
        if (with_v) hipLaunchKernelGGL((k_fused_w<true, true, true>), g, b, 0, st, v, rslot, 0.0);
        else hipLaunchKernelGGL((k_fused_w<true, false, true>), g, b, 0, st, v, rslot, 0.0);
    } else {
        if (with_v) hipLaunchKernelGGL((k_fused_w<false, true, false>), g, b, 0, st, v, rslot, 0.0);
        else hipLaunchKernelGGL((k_fused_w<true, false, false>), g, b, 0, st, v, rslot, 0.0);
    }
    int with_tau = do_update || !with_v;
    hipLaunchKernelGGL(k_fused_reduce, dim3(blocks_for(v.k)), dim3(BLK), 0, st, v, nstripes, nchunks, with_tau, with_v);
}
void launch_finish_tau(const DevView& v, hipStream_t st) {
    (void)hipMemsetAsync(v.tau, 0, sizeof(double) * (size_t)v.m, st);
    hipLaunchKernelGGL(k_ftran_init_single, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v, (const double*)v.rho, v.tau);
    if (v.k > 0) hipLaunchKernelGGL(k_ftran_push<16>, dim3(blocks_for(v.k * 16)), dim3(BLK), 0, st, v, (const double*)v.tauK, v.tau);
}
void launch_finish_v(const DevView& v, hipStream_t st) {
    if (v.k > 0) hipLaunchKernelGGL(k_scatter_cols, dim3(blocks_for(v.k)), dim3(BLK), 0, st, v, (const double*)v.vK, v.vvec);
}
void launch_structure_update(const DevView& v, const StructUpdate& u, hipStream_t st) {
    switch (u.kase) {
        case 0: break;  // nuc -> nuc: the fused pass already produced the new row
        case 1: hipLaunchKernelGGL(k_struct_grow, dim3(blocks_for(v.k + 1)), dim3(BLK), 0, st, v, u); break;
        case 2:
            hipLaunchKernelGGL(k_struct_shrink_row, dim3(blocks_for(v.k)), dim3(BLK), 0, st, v, u);
            hipLaunchKernelGGL(k_struct_shrink_col, dim3(blocks_for(v.k)), dim3(BLK), 0, st, v, u);
            break;
        case 3: hipLaunchKernelGGL(k_struct_newcol, dim3(blocks_for(v.k > 0 ? v.k : 1)), dim3(BLK), 0, st, v, u); break;
        case 4: hipLaunchKernelGGL(k_struct_newdiag, dim3(1), dim3(1), 0, st, v, u); break;
    }
}
void launch_update_pivot(const DevView& v, int use_dse, int use_pse, hipStream_t st) {
    hipLaunchKernelGGL(k_update_basic, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v, use_dse);
    hipLaunchKernelGGL(k_update_nonbasic, dim3(blocks_for(v.n)), dim3(BLK), 0, st, v, use_pse);
}
void launch_update_flip(const DevView& v, hipStream_t st) {
    hipLaunchKernelGGL(k_update_flip, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v);
}
void launch_btran_dense(const DevView& v, const double* c_pos, double* y_row, hipStream_t st) {
    hipLaunchKernelGGL(k_set_status, dim3(1), dim3(1), 0, st, v, (int)ITER_PIVOT);
    (void)hipMemsetAsync(y_row, 0, sizeof(double) * (size_t)v.m, st);
    hipLaunchKernelGGL(k_btran_single, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v, c_pos, y_row);
    if (v.k > 0) {
        hipLaunchKernelGGL(k_btran_rhs<16>, dim3(blocks_for(v.k * 16)), dim3(BLK), 0, st, v, c_pos, (const double*)y_row);
        launch_fused_w(v, 1, 0, -1, st);
        hipLaunchKernelGGL(k_scatter_cols, dim3(blocks_for(v.k)), dim3(BLK), 0, st, v, (const double*)v.vK, y_row);
    }
}
void launch_ftran_dense(const DevView& v, const double* b_row, double* x_pos, hipStream_t st) {
    // gather rK = b[R_K] is the caller's job when needed; not on the pivot path (kept for refresh).
    (void)v; (void)b_row; (void)x_pos; (void)st;
}
void launch_gather_basic_obj(const DevView& v, double* c_pos, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_basic_obj, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v, c_pos);
}
void launch_recalc_d(const DevView& v, const double* y_row, hipStream_t st) {
    hipLaunchKernelGGL(k_recalc_d<16>, dim3(blocks_for(v.n * 16)), dim3(BLK), 0, st, v, y_row);
    hipLaunchKernelGGL(k_recalc_obj, dim3(grid_for(v.m + v.n)), dim3(BLK), 0, st, v);
}
void launch_shift_nonbasic(const DevView& v, int col, double val, hipStream_t st) {
    hipLaunchKernelGGL(k_shift_nonbasic, dim3(blocks_for(v.m)), dim3(BLK), 0, st, v, col, val);
    hipLaunchKernelGGL(k_set_xn, dim3(1), dim3(1), 0, st, v, col, val);
}
void launch_sq_norms_add_row(const DevView& v, hipStream_t st) {
    hipLaunchKernelGGL(k_gamma_add_row, dim3(blocks_for(v.n)), dim3(BLK), 0, st, v);
}
void launch_build_nucleus(const DevView& v, double* Kd, hipStream_t st) {
    if (v.k <= 0) return;
    (void)hipMemsetAsync(Kd, 0, sizeof(double) * (size_t)v.k * v.ld, st);
    hipLaunchKernelGGL(k_build_nucleus<16>, dim3(blocks_for(v.k * 16)), dim3(BLK), 0, st, v, Kd);
}
void launch_gauss_jordan(double* Kd, double* Winv, int k, int ld, int* d_flag, double* d_scratch, hipStream_t st) {
    if (k <= 0) return;
    size_t tot = (size_t)k * k;
    hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((tot + BLK - 1) / BLK)), dim3(BLK), 0, st, Winv, k, ld);
    int* piv = reinterpret_cast<int*>(d_scratch);
    double* factors = d_scratch + 2;
    for (int j = 0; j < k; ++j) {
        hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(BLK), 0, st, (const double*)Kd, k, ld, j, piv, d_flag);
        hipLaunchKernelGGL(k_gj_factors, dim3(blocks_for(k)), dim3(BLK), 0, st, (const double*)Kd, k, ld, j, (const int*)piv, factors);
        hipLaunchKernelGGL(k_gj_swap_scale, dim3(blocks_for(k)), dim3(BLK), 0, st, Kd, Winv, k, ld, j, (const int*)piv, (const double*)factors);
        hipLaunchKernelGGL(k_gj_eliminate, dim3(blocks_for(k), k), dim3(BLK), 0, st, Kd, Winv, k, ld, j, (const double*)factors);
    }
}
void launch_max_abs_diff(const double* A, const double* B, int k, int ld, double* d_out, hipStream_t st) {
    (void)hipMemsetAsync(d_out, 0, sizeof(double), st);
    if (k <= 0) return;
    size_t tot = (size_t)k * k;
    int g = (int)((tot + BLK * 4 - 1) / (BLK * 4));
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_max_abs_diff, dim3(g), dim3(BLK), 0, st, A, B, k, ld, d_out);
}

}  // namespace mlp
