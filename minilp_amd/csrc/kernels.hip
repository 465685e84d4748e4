// kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the simplex pivot hot path.
//
// Reference loops replaced (file:line in ztlpn/minilp 0.2.2) are cited per kernel.  None of this
// is GEMM-shaped: every kernel is an HBM/L2-bound stream with wave64 shuffle reductions, so there
// is no MFMA here (DESIGN.md §4 gives the algorithmic bytes per kernel).
//
// All pivot kernels take the DevView BY VALUE (pointers and sizes) and read the nucleus size and the pivot
// scalars from the device-resident Ctl block, so one iteration is a fixed launch sequence (hipGraph).
//
// Map of the file: helpers and grid-wide reductions; mailbox exchanges of the sharded mode; partition plan;
// stage heads (FTRAN / BTRAN); K1/K6 pricing; K2 FTRAN (+ blocked and pulled forms of the F product);
// K5 primal ratio test; K3 BTRAN; partition change; K4 tableau row (CSC pull and banded sweep); K7 dual
// ratio test; fused pass over the nucleus inverse (in place / streaming / fold) and its tails; K8 update
// (+ next pricing); helpers for recalc / re-inversion; launch wrappers.
#include "kernels.h"
#include <hip/hip_ext.h>

#include <limits.h>
#include <math.h>

#include <cstdlib>
#include <string>

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
// MLP_KPROF=1 (diagnostics): thread 0 of the marking block stamps the 100 MHz wall clock into Ctl.hy_prof[i]; after a batch the
// slots hold the timeline of its LAST iteration (kernel entries and the marks inside the two chain-bound kernels), read
// through state("kernel_timeline").  Off: one scalar load and a not-taken branch per mark.
#define KMARK(c, i)                                                                   \
    do {                                                                              \
        if ((c)->kprof_on == 1 && threadIdx.x == 0) (c)->hy_prof[i] = wall_clock64();        \
    } while (0)
#define KMARK0(c, i)                                                                  \
    do {                                                                              \
        if (blockIdx.x == 0 && blockIdx.y == 0) KMARK(c, i);                          \
    } while (0)
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

__device__ __forceinline__ double block_max(double x) {  // thread 0 (diagnostics)
    __shared__ double s[BLK / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_down(x, o, 64));
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) s[w] = x;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < BLK / 64; ++i) x = fmax(x, s[i]);
    return x;
}

// Cross-workgroup hand-off without fences (cdna_hip_programming.md §6 G16, form "8-B agent atomics
// both sides"): every partial is stored write-through (relaxed agent-scope atomic store = sc1), the
// storing lane drains its stores (s_waitcnt vmcnt(0)) and takes a ticket; the last arriver re-reads
// all partials with agent-scope (sc1, L1-bypassing) loads.  No buffer_wbl2 / buffer_inv needed.
__device__ __forceinline__ void st_agent(double* p, double x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Arrivals at ONE address serialise at ~15 ns each (391 of them ~8 us, 1 563 — the update of a 400 000-column model — most of that kernel's
// 18 us): above TK_ONE arrivals they count in two steps — TK_GROUPS first-step tickets 4 KB apart (arrival index mod TK_GROUPS; the last
// arrival at one resets it and arrives on the ticket itself).  The caller's reset of `ticket` is the same in both forms.  `my`: the
// arrival's index in [0, nblocks) when it is not the block index.
__device__ __forceinline__ bool last_block_arrives(unsigned* ticket, unsigned nblocks, int my = -1) {
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool last;
        if (nblocks <= (unsigned)TK_ONE) {
            last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
        } else {
            const unsigned me = my < 0 ? blockIdx.x : (unsigned)my, x = me % (unsigned)TK_GROUPS;
            const unsigned cnt = (nblocks - x + (unsigned)TK_GROUPS - 1u) / (unsigned)TK_GROUPS;
            unsigned* sub = ticket + (size_t)TK_STRIDE * (1u + x);
            last = false;
            if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cnt - 1u) {
                __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a launch may use the ticket twice: the reset has landed before anyone can hear of this arrival)
                last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)TK_GROUPS - 1u;
            }
        }
        s_last = last ? 1 : 0;
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
__device__ __forceinline__ bool grid_best(Cand& c, const DevView& v, int nblocks = -1) {
    if (nblocks < 0) nblocks = (int)gridDim.x;  // (a launch may carry extra, horizontally fused blocks behind the reducing ones)
    c = block_best(c);
    if (threadIdx.x == 0) {
        st_agent(&v.red_key[blockIdx.x], c.key);
        st_agent(&v.red_idx[blockIdx.x], c.idx);
    }
    if (!last_block_arrives(v.ticket, (unsigned)nblocks)) return false;
    Cand x = cand_none();
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        Cand t{ld_agent(&v.red_key[i]), ld_agent(&v.red_idx[i])};
        if (cand_better(t, x)) x = t;
    }
    c = block_best(x);
    if (threadIdx.x == 0) *v.ticket = 0;
    return true;
}
// arg-best with a payload travelling with the winner (used by pricing: the winner's reduced cost
// must come through the reduction, not through a re-read of memory another block has just written)
__device__ __forceinline__ void wave_best_p(Cand& c, double& pay) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand t;
        t.key = __shfl_down(c.key, o, 64);
        t.idx = __shfl_down(c.idx, o, 64);
        double tp = __shfl_down(pay, o, 64);
        if (cand_better(t, c)) {
            c = t;
            pay = tp;
        }
    }
}
__device__ __forceinline__ void block_best_p(Cand& c, double& pay) {  // result valid in thread 0
    __shared__ double s_key[BLK / 64], s_pay[BLK / 64];
    __shared__ int s_idx[BLK / 64];
    wave_best_p(c, pay);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) {
        s_key[w] = c.key;
        s_idx[w] = c.idx;
        s_pay[w] = pay;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < BLK / 64; ++i) {
            Cand t{s_key[i], s_idx[i]};
            if (cand_better(t, c)) {
                c = t;
                pay = s_pay[i];
            }
        }
    }
}
__device__ __forceinline__ bool grid_best_p(Cand& c, double& pay, const DevView& v, int nblocks = -1) {
    if (nblocks < 0) nblocks = (int)gridDim.x;  // (a launch may carry extra, horizontally fused blocks behind the reducing ones)
    block_best_p(c, pay);
    if (threadIdx.x == 0) {
        st_agent(&v.red_key[blockIdx.x], c.key);
        st_agent(&v.red_idx[blockIdx.x], c.idx);
        st_agent(&v.red_key2[blockIdx.x], pay);
    }
    if (!last_block_arrives(v.ticket, (unsigned)nblocks)) return false;
    KMARK(v.ctl, 16);
    Cand x = cand_none();
    double xp = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        Cand t{ld_agent(&v.red_key[i]), ld_agent(&v.red_idx[i])};
        if (cand_better(t, x)) {
            x = t;
            xp = ld_agent(&v.red_key2[i]);
        }
    }
    KMARK(v.ctl, 17);
    block_best_p(x, xp);
    c = x;
    pay = xp;
    if (threadIdx.x == 0) *v.ticket = 0;
    return true;
}
// grid-wide (min, sum) pair in one pass
__device__ __forceinline__ bool grid_min_sum(double& mn, double& sm, const DevView& v, int nblocks = -1) {
    if (nblocks < 0) nblocks = (int)gridDim.x;
    mn = block_min(mn);
    sm = block_sum(sm);
    if (threadIdx.x == 0) {
        st_agent(&v.red_key[blockIdx.x], mn);
        st_agent(&v.red_key2[blockIdx.x], sm);
    }
    if (!last_block_arrives(v.ticket, (unsigned)nblocks)) return false;
    double y = INFINITY, z = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        double t = ld_agent(&v.red_key[i]);
        if (t < y) y = t;
        z += ld_agent(&v.red_key2[i]);
    }
    mn = block_min(y);
    sm = block_sum(z);
    if (threadIdx.x == 0) *v.ticket = 0;
    return true;
}
__device__ __forceinline__ bool grid_sum(double& x, const DevView& v, int nblocks = -1, int my = -1) {
    if (nblocks < 0) nblocks = (int)gridDim.x;
    if (my < 0) my = (int)blockIdx.x;  // (the reducing blocks may sit behind other blocks of the launch: their own index then)
    x = block_sum(x);
    if (threadIdx.x == 0) st_agent(&v.red_key[my], x);
    if (!last_block_arrives(v.ticket, (unsigned)nblocks, my)) return false;
    double y = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) y += ld_agent(&v.red_key[i]);
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
// Kernel-exact timing of the sampled iterations (bench.py's roofline entries): when a pair of events is armed for a
// slot, the launch goes through hipExtLaunchKernelGGL, which stamps the events at the start and at the end of the
// KERNEL — the figure rocprofv3 reports — instead of bracketing the launch with hipEventRecord (which adds the
// dispatch gap of an eager launch: 27.5 against 23.9 us on the banded sweep).
// thread_local: two Solutions in profile mode on different host threads must not stamp each other's events (a Solution is
// driven by one thread at a time; arming and launching happen on that thread)
static thread_local hipEvent_t g_tev[4][2] = {};
void arm_kernel_timing(int slot, hipEvent_t t0, hipEvent_t t1) {
    if (slot > 0 && slot < 4) {
        g_tev[slot][0] = t0;
        g_tev[slot][1] = t1;
    }
}
#define LAUNCH_T(slot, kern, grid, block, lds, st, ...)                                                          \
    do {                                                                                                         \
        if (g_tev[slot][0]) hipExtLaunchKernelGGL(kern, grid, block, lds, st, g_tev[slot][0], g_tev[slot][1], 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                        \
    } while (0)
static inline int blocks_for(long n, int per_block = BLK) { return n <= 0 ? 1 : (int)((n + per_block - 1) / per_block); }

template <int G>
__device__ __forceinline__ double group_sum(double x) {  // xor tree inside G consecutive lanes
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ void push_rec(Ctl* c, int phase) {  // single thread
    int n = c->ring_n;
    if (n < RING) {
        PivotRec& r = c->ring[n];
        r.status = c->it.status;
        r.phase = phase;
        r.q = c->it.q;
        r.r = c->it.r;
        r.entering_var = c->it.entering_var;
        r.leaving_var = c->it.leaving_var;
        r.kase = c->up.kase;
        r.k_after = c->k;
        r.klist_n = c->it.klist_n;
        r.blist_n = c->it.blist_n;
        r.pivot_coeff = c->it.pivot_coeff;
        // MLP_KPROF=2..5 (diagnostics): the record carries a per-rank quantity of the iteration in place of the pivot element, so
        // that the traces of the ranks of a sharded solve can be compared: ||alpha_q||^2 + 1, ||rho||^2, 1 / alpha_q[r], sum of U[j].t_K
        if (c->kprof_on == 2) r.pivot_coeff = c->it.alpha_sq;
        else if (c->kprof_on == 3) r.pivot_coeff = c->it.rho_sq;
        else if (c->kprof_on == 4) r.pivot_coeff = c->it.inv_alpha;
        else if (c->kprof_on == 5) {
            double sh = 0.0;
            for (int j = 0; j < c->nlow && j < LR_MAX; ++j) sh += c->lr_h[j];
            r.pivot_coeff = sh;
        }
        r.obj = c->it.obj;
    }
    c->ring_n = n + 1;
}

// ---- sharded pricing: mailbox exchange between the ranks of one solve (one thread) ------------
// Records live in host memory mapped into every GPU (fine-grained, system-scope coherent).  A rank
// writes only its own slot (payload, then the epoch with system-scope release) and polls the other
// slots for the same epoch; two parity buffers per kind keep a fast rank from overwriting a record
// a slow rank has not read yet.  Spins are bounded: a missing peer ends the solve with ITER_COMM.
__device__ __forceinline__ MailRec* mail_slot(const DevView& v, int kind, unsigned long long epoch, int rank) {
    return v.mail + ((size_t)(kind * 2 + (int)(epoch & 1ull)) * v.world + rank);
}
// this rank's slot inside the box of `peer` (the box that peer polls)
__device__ __forceinline__ MailRec* mail_slot_at(const DevView& v, int peer, int kind, unsigned long long epoch) {
    return v.mail_peer[peer] + ((size_t)(kind * 2 + (int)(epoch & 1ull)) * v.world + v.rank);
}
__device__ __forceinline__ void mail_store(MailRec* slot, unsigned long long epoch, const double* f) {
#pragma unroll
    for (int i = 0; i < 7; ++i) __hip_atomic_store(&slot->f[i], f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&slot->epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Post this rank's record of (kind, epoch) to every box that must see it.  Wave-parallel form: lane l < fanout
// writes into peer l's box (one round of xGMI stores for all peers); call with the whole wave.
__device__ __forceinline__ void mail_post_wave(const DevView& v, int kind, unsigned long long epoch, const double* f, int lane) {
    if (v.mail_fanout <= 1) {
        if (lane == 0) mail_store(mail_slot(v, kind, epoch, v.rank), epoch, f);
    } else if (lane < v.mail_fanout) {
        mail_store(mail_slot_at(v, lane, kind, epoch), epoch, f);
    }
}
// Single-thread form: all payloads first, one release fence, then the epochs.
__device__ __forceinline__ void mail_post(const DevView& v, int kind, unsigned long long epoch, const double* f) {
    if (v.mail_fanout <= 1) {
        mail_store(mail_slot(v, kind, epoch, v.rank), epoch, f);
        return;
    }
    for (int r = 0; r < v.mail_fanout; ++r) {
        MailRec* slot = mail_slot_at(v, r, kind, epoch);
#pragma unroll
        for (int i = 0; i < 7; ++i) __hip_atomic_store(&slot->f[i], f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);  // system scope: the payloads are visible before any epoch
    for (int r = 0; r < v.mail_fanout; ++r)
        __hip_atomic_store(&mail_slot_at(v, r, kind, epoch)->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool mail_wait(MailRec* slot, unsigned long long epoch, double* f) {
    for (long spins = 0;; ++spins) {
        if (__hip_atomic_load(&slot->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) break;
        if (spins > 60000000L) return false;  // ~10+ s
        __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) f[i] = __hip_atomic_load(&slot->f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
}
__device__ __forceinline__ void comm_fail(Ctl* c, int phase) {
    c->it.status = ITER_COMM;
    c->halt = 1;
    push_rec(c, phase);
}
// all-gather of per-shard candidates, run by ONE WAVE: lane r polls rank r's slot (one PCIe round trip
// for all peers), then the wave reduces them identically on every rank (score desc, position asc) and
// adopts the winner's payloads.  kind 0: primal pricing (payload d_q); kind 3: dual ratio pass 2
// (payloads alpha_rq and d_q).  Result valid in lane 0.
__device__ bool exchange_best_wave(const DevView& v, Ctl* c, int phase, Cand& best, double pay_local, int lane,
                                   int kind = 0, double pay2_local = 0.0) {
    if (v.world <= 1) return true;
    unsigned long long ep = c->xepoch[kind] + 1ull;
    {   // `best` and the payloads are valid in lane 0 only: broadcast them, then lane l posts into peer l's box
        double f[7] = {__shfl(best.key, 0, 64), (double)__shfl(best.idx, 0, 64), __shfl(pay_local, 0, 64),
                       __shfl(pay2_local, 0, 64), 0.0, 0.0, 0.0};
        mail_post_wave(v, kind, ep, f, lane);
    }
    Cand g = cand_none();
    double dq = 0.0, p2 = 0.0;
    int owner = -1;
    bool ok = true;
    for (int r2 = lane; r2 < v.world; r2 += 64) {
        Cand t = best;
        double td = pay_local, t2 = pay2_local;
        if (r2 != v.rank) {
            double h[7];
            if (!mail_wait(mail_slot(v, kind, ep, r2), ep, h)) ok = false;
            t = Cand{h[0], (int)h[1]};
            td = h[2];
            t2 = h[3];
        }
        if (owner < 0 || cand_better(t, g)) {
            g = t;
            dq = td;
            p2 = t2;
            owner = r2;
        }
    }
    ok = __all(ok);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand t;
        t.key = __shfl_down(g.key, o, 64);
        t.idx = __shfl_down(g.idx, o, 64);
        double td = __shfl_down(dq, o, 64);
        double t2 = __shfl_down(p2, o, 64);
        int to = __shfl_down(owner, o, 64);
        if (to >= 0 && (owner < 0 || cand_better(t, g))) {
            g = t;
            dq = td;
            p2 = t2;
            owner = to;
        }
    }
    if (lane == 0) {
        c->xepoch[kind] = ep;
        if (!ok) {
            comm_fail(c, phase);
        } else {
            if (g.idx != NONE_IDX && owner != v.rank) {  // non-owners hold no valid d / alpha_r outside their block
                v.d[g.idx] = dq;
                if (kind == 3) v.alpha_r[g.idx] = p2;
            }
            best = g;
        }
    }
    return ok;
}
// all-reduce MIN of one double per rank (dual ratio pass 1, kind 2); same wave-parallel poll.
__device__ bool exchange_min_wave(const DevView& v, Ctl* c, double& mn, int lane) {
    if (v.world <= 1) return true;
    unsigned long long ep = c->xepoch[2] + 1ull;
    {
        double f[7] = {__shfl(mn, 0, 64), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        mail_post_wave(v, 2, ep, f, lane);
    }
    double g = INFINITY;
    bool ok = true;
    for (int r2 = lane; r2 < v.world; r2 += 64) {
        double t = mn;
        if (r2 != v.rank) {
            double h[7];
            if (!mail_wait(mail_slot(v, 2, ep, r2), ep, h)) ok = false;
            t = h[0];
        }
        if (t < g) g = t;
    }
    ok = __all(ok);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_down(g, o, 64);
        if (t < g) g = t;
    }
    if (lane == 0) {
        c->xepoch[2] = ep;
        if (!ok) comm_fail(c, 1);
        else mn = g;
    }
    return ok;
}
// Dual pricing runs over the (replicated) basic side, but x_B carries float atomics and is reproducible
// only to rounding across ranks: every rank adopts rank 0's leaving row (kind 0, one lane).
__device__ bool adopt_rank0_candidate(const DevView& v, Ctl* c, Cand& best) {
    if (v.world <= 1) return true;
    const unsigned long long ep = ++c->xepoch[0];
    if (v.rank == 0) {
        double f[7] = {best.key, (double)best.idx, 0.0, 0.0, 0.0, 0.0, 0.0};
        mail_post(v, 0, ep, f);
        return true;
    }
    double h[7];
    if (!mail_wait(mail_slot(v, 0, ep, 0), ep, h)) {
        comm_fail(c, 1);
        return false;
    }
    best = Cand{h[0], (int)h[1]};
    return true;
}

// Transport handshake (Engine::enable_sharding): one wave posts a record of the handshake kind (MAIL_KINDS - 1, used
// by nothing else) with epoch 1 to every box and waits for every rank's record with a short bound; out[0] = 1 on
// success, out[1] = number of ranks heard from.
__global__ void __launch_bounds__(64) k_mail_handshake(DevView v, int* out) {
    const int lane = threadIdx.x;
    const unsigned long long ep = 1ull;
    double f[7] = {(double)v.rank, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    mail_post_wave(v, MAIL_KINDS - 1, ep, f, lane);
    bool ok = true;
    int heard = 0;
    for (int r = lane; r < v.world; r += 64) {
        MailRec* slot = mail_slot(v, MAIL_KINDS - 1, ep, r);
        bool got = false;
        for (long spins = 0; spins < 12000000L; ++spins) {  // ~2 s
            if (__hip_atomic_load(&slot->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == ep) {
                got = true;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (got && __hip_atomic_load(&slot->f[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == (double)r) heard += 1;
        else ok = false;
    }
    ok = __all(ok);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) heard += __shfl_down(heard, o, 64);
    if (lane == 0) {
        out[1] = heard;
        out[0] = ok ? 1 : 0;
    }
}
// ---- pump transport (MLP_TRANSPORT=rccl | pump): delivery of the mailbox records by a collective instead of peer stores -------
// The pivot kernels post into and poll their OWN device box (mail_fanout = 1, exactly as with the host mailbox); a pump on a second
// stream moves the records between the ranks while a batch of iterations is in flight:
//   k_mail_stage   : this rank's records of every (kind, parity) are copied, seqlock-consistently, into its block of a staging buffer;
//   all-gather     : ncclAllGather of the blocks (RCCL over xGMI), or — ranks sharing one GPU, which RCCL refuses — peer copies;
//   k_mail_deliver : the other ranks' records go from the staging buffer into this rank's box, payload first, then the epoch
//                    with release, which is what the polling kernels acquire on.
// A record is only overwritten (epoch e + 2 into the slot of e) after every rank has consumed e, so a staged record is never torn
// with respect to an epoch a consumer still waits for.  The last record of every staging block carries the host's "my batch is
// done" word of the pump protocol (Engine::pump_until_idle).
constexpr int PUMP_RECS = MAIL_KINDS * 2 + 1;  // records per rank in the staging buffer
__global__ void __launch_bounds__(64) k_mail_stage(DevView v, MailRec* stage) {
    const int t = threadIdx.x;
    if (t >= MAIL_KINDS * 2) return;
    const MailRec* src = v.mail + (size_t)t * v.world + v.rank;  // [kind * 2 + parity][rank]
    MailRec* dst = stage + (size_t)v.rank * PUMP_RECS + t;
    const unsigned long long e1 = __hip_atomic_load(&src->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    double f[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) f[i] = __hip_atomic_load(&src->f[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long e2 = __hip_atomic_load(&src->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (e1 != e2) return;  // being rewritten: the previous staged copy stands, the next round takes the new one
    dst->epoch = e1;
#pragma unroll
    for (int i = 0; i < 7; ++i) dst->f[i] = f[i];
}
__global__ void __launch_bounds__(256) k_mail_deliver(DevView v, const MailRec* stage) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int t = idx % (MAIL_KINDS * 2), r = idx / (MAIL_KINDS * 2);
    if (r >= v.world || r == v.rank) return;
    const MailRec* src = stage + (size_t)r * PUMP_RECS + t;
    MailRec* dst = v.mail + (size_t)t * v.world + r;
    const unsigned long long e = src->epoch;
    if (e == 0ull || e == __hip_atomic_load(&dst->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
    mail_store(dst, e, src->f);
}
void launch_mail_stage(const DevView& dv, void* stage, hipStream_t st) {
    hipLaunchKernelGGL(k_mail_stage, dim3(1), dim3(64), 0, st, dv, reinterpret_cast<MailRec*>(stage));
}
void launch_mail_deliver(const DevView& dv, const void* stage, hipStream_t st) {
    const int n = dv.world * MAIL_KINDS * 2;
    hipLaunchKernelGGL(k_mail_deliver, dim3((n + 255) / 256), dim3(256), 0, st, dv, reinterpret_cast<const MailRec*>(stage));
}
void launch_mail_handshake(const DevView& dv, int* out, hipStream_t st) {
    hipLaunchKernelGGL(k_mail_handshake, dim3(1), dim3(64), 0, st, dv, out);
}

// Partition-change plan (HISTORY.md §3.3), run by ONE thread once q, r and the final alpha_q are
// known.  The host never needs (q, r): this is what makes the iteration graph-replayable.
__device__ void plan_update(const DevView& v, Ctl* c, int phase) {
    StructUpdate& u = c->up;
    u.kase = -1;
    if (c->it.status != ITER_PIVOT) return;
    if (v.fac_on) {  // compact factor (factor.inc): no partition to maintain; the pivot becomes a rank-1 term (k_fac_append)
        c->it.inv_alpha = 1.0 / ld_agent(&v.alpha_q[c->it.r]);
        c->fold = 0;
        u.r = c->it.r;
        return;
    }
    const int r = c->it.r, ev = c->it.entering_var;
    const int sr = v.kslot_of_pos[r];
    const bool old_nuc = sr >= 0;
    const int cb = v.csc_ptr[ev];
    const bool new_sing = (v.csc_ptr[ev + 1] - cb) == 1;
    u.r = r;
    u.sr = sr;
    u.kold = c->k;
    u.i_r = -1;
    u.i_q = -1;
    u.cq = -1;
    u.diag_q = 0.0;
    u.inv_diag_r = 0.0;
    if (!old_nuc) {
        u.i_r = v.srow_of_pos[r];
        u.inv_diag_r = 1.0 / v.sdiag_of_pos[r];
    }
    c->it.inv_alpha = 1.0 / ld_agent(&v.alpha_q[r]);  // (L1-bypassing: k_primal_head reads entries that L2 atomics of the same launch produced)
    // (primal iteration: the FTRAN head took the fold decision — with the v branch the fold of this pivot may already be running
    // beside this plan, and its last block clears nlow)
    if (phase != 0 || !v.lrJ) c->fold = (v.lrJ > 0 && c->nlow >= v.lrJ) ? 1 : 0;
    u.jn = c->fold ? 0 : c->nlow;
    u.pad0 = 0;
    u.pad = 0;
    if (new_sing) {
        u.i_q = v.csc_row[cb];
        u.diag_q = v.csc_val[cb];
        u.cq = v.kslot_of_row[u.i_q];
        if (u.cq < 0) {
            // an entering singleton on an S-row is legal only when it replaces the singleton
            // covering that row (then alpha_K = 0 and only the diagonal entry changes)
            if (!old_nuc && u.i_q == u.i_r) {
                u.kase = 4;
            } else {
                c->it.status = ITER_SINGULAR;
                c->halt = 1;
                push_rec(c, phase);
            }
        } else {
            u.kase = old_nuc ? 2 : 3;
        }
    } else {
        if (!old_nuc && c->k >= v.ld) {  // no room to grow: the host failed to reserve capacity
            c->it.status = ITER_SINGULAR;
            c->halt = 1;
            push_rec(c, phase);
        } else {
            u.kase = old_nuc ? 0 : 1;
        }
    }
}

// acc += sum over the lanes b of `mask` (ascending) of coef_b * row[slot_b]: the listed entries of one 64-entry chunk
// travel by shuffle in list order; four gathers are issued before the four adds (same summation order as a sequential
// walk, a quarter of the load waits)
__device__ __forceinline__ void lr_dot_listed(unsigned long long mask, int slot, double coef, const double* row, bool mine, double& acc) {
    while (mask) {
        int sb[4];
        double ab[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool have = mask != 0ull;
            const int b = have ? __ffsll((long long)mask) - 1 : 0;
            sb[u] = __shfl(slot, b, 64);
            ab[u] = have ? __shfl(coef, b, 64) : 0.0;
            if (!have) sb[u] = 0;
            mask &= mask - 1ull;  // (0 stays 0)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = mine ? row[sb[u] < 0 ? 0 : sb[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ab[u] != 0.0) acc += ab[u] * x[u];
    }
}
// Sparse primal ratio test (small nucleus, v.str_on, pushed F products): the FTRAN lists the positions of supp(alpha_q) as it
// touches them — each once, through the epoch stamp of the position — so that the Harris test, ||alpha_q||^2 and the
// singleton part of v run over that list in ONE block instead of two grid-wide passes over all m positions.
__device__ __forceinline__ bool aq_listing(const DevView& v) { return v.str_on && !v.pb_on && !v.det_pull && v.world <= 1; }
__device__ __forceinline__ void aq_list_add(const DevView& v, Ctl* c, int p) {
    const int ep = c->hyper_epoch + 1;
    if (atomicExch(&v.hy_stamp_p[p], ep) != ep) {
        const int o = atomicAdd(&c->aq_n, 1);
        if (o < v.m) v.aq_list[o] = p;
    }
}
// ------------------------------------------------------------------- wave-level stage heads
// FTRAN head (one wave): derive the entering column's scalars, land its singleton-row entries in
// alpha_q and list its entries on nucleus rows.  alpha_q = B^-1 a_q  (solver.rs:671-677).
__device__ void ftran_prep_wave(const DevView& v, Ctl* c, int lane, int derive_primal) {
    IterState* it = &c->it;
    const int q = it->q;
    const int var = v.nb_vars[q];
    if (lane == 0) {
        it->entering_var = var;
        if (derive_primal) {  // solver.rs:741-748
            double dq = v.d[q];
            it->sign = dq < 0.0;
            it->entering_cur = v.xN[q];
            it->entering_other = (dq < 0.0) ? v.var_hi[var] : v.var_lo[var];
            it->r = -1;
            it->leaving_var = -1;
            // delayed-update mode: a full list of pending terms is folded by this pivot — decided here, ahead of everything
            // that reads W0, so that the fold may run beside the ratio test (v branch) instead of behind it
            if (v.lrJ) c->fold = c->nlow >= v.lrJ ? 1 : 0;
        }
    }
    int base = v.csc_ptr[var], end = v.csc_ptr[var + 1];
    if (v.fac_on) {  // compact factor: the entering column becomes the right-hand side (by row) of the level-scheduled solve
        int hi = -1;
        for (int e = base + lane; e < end; e += 64) {
            const int row = v.csc_row[e];
            v.fac_rhs[row] = v.csc_val[e];
            const int lv = v.fac_lev_of_row[row];
            hi = max(hi, lv < 0 ? INT_MAX : lv);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hi = max(hi, __shfl_down(hi, o, 64));
        if (lane == 0) {
            it->klist_n = 0;
            c->fac_aq_hi = hi;  // the FTRAN walks the levels from this one down
        }
        return;
    }
    int cnt = 0;
    // delayed-update mode: c_j = V[j] . (listed entries of a_q), lane j serves pending term j (LR_MAX <= 64).  The
    // listed entries travel by shuffle in list order (the same summation order as a walk over the stored list), so
    // the gathers of one chunk are independent loads instead of a chain of re-reads of the list
    const int nlow = v.lrJ ? c->nlow : 0;
    const double* Vrow = v.V + (size_t)(lane < nlow ? lane : 0) * v.ld;
    double lr_acc = 0.0;
    // Two chunks of 64 entries per trip, their loads issued together: a 100-entry column is ONE round of the dependent
    // chain entry -> row -> packed row map (diag, position, slot) instead of two rounds of a four-deep one.
    for (int e0 = base; e0 < end; e0 += 128) {
        int sx[2] = {-1, -1};
        double ax[2] = {0.0, 0.0};
        int ix[2];
        bool vx[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = e0 + 64 * h + lane;
            vx[h] = e < end;
            ix[h] = vx[h] ? v.csc_row[e] : 0;
            ax[h] = vx[h] ? v.csc_val[e] : 0.0;
        }
        RowInfo rx[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) rx[h] = v.rowinfo[ix[h]];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (vx[h]) {
                sx[h] = rx[h].kslot;
                if (sx[h] < 0) {
                    v.alpha_q[rx[h].pos] = ax[h] / rx[h].diag;
                    if (aq_listing(v)) aq_list_add(v, c, rx[h].pos);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && e0 + 64 >= end) break;  // (uniform)
            const int s = sx[h];
            const double a = ax[h];
            bool isk = vx[h] && s >= 0;
            unsigned long long mask = __ballot(isk);
            if (isk) {
                int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                v.klist_s[off] = s;
                v.klist_a[off] = a;
            }
            cnt += __popcll(mask);
            if (nlow > 0) lr_dot_listed(mask, s, a, Vrow, lane < nlow, lr_acc);
        }
    }
    if (lane == 0) it->klist_n = cnt;
    if (lane < nlow) c->lr_c[lane] = lr_acc;
}
// BTRAN head (one wave): rho = B^-T e_r (solver.rs:680-683) as a short list of rows of W.
__device__ void btran_prep_wave(const DevView& v, Ctl* c, int lane, int r, int derive_dual, int plan_after, int phase) {
    IterState* it = &c->it;
    if (derive_dual && !c->forced && lane == 0) {  // solver.rs:892-916 (a host-forced row keeps the host's value)
        double val = v.xB[r], mn = v.loB[r];
        it->leaving_new_val = (val < mn) ? mn : v.hiB[r];
        it->leaving_var = v.basic_vars[r];
        it->q = -1;
        it->entering_var = -1;
    }
    if (v.fac_on) {  // compact factor: rho = B^-T e_r comes from the level-scheduled solve (k_fac_solve), nothing to list
        if (lane == 0) {
            it->blist_n = 0;
            if (plan_after) plan_update(v, c, phase);
        }
        return;
    }
    int sr = v.kslot_of_pos[r];
    const int nlow = v.lrJ ? c->nlow : 0;
    const double* Urow = v.U + (size_t)(lane < nlow ? lane : 0) * v.ld;
    double lr_acc = 0.0;
    if (sr >= 0) {
        if (lane == 0) {
            v.blist_s[0] = sr;
            v.blist_a[0] = 1.0;
            it->blist_n = 1;
        }
    } else {
        int i_r = v.srow_of_pos[r];
        double inv = 1.0 / v.sdiag_of_pos[r];
        if (lane == 0) {
            v.rv[i_r].x = inv;
            v.tau[r] = inv * inv;  // tau_S = (rho_S - F tauK)/diag: only row i_r of rho_S is non-zero
        }
        int base = v.csr_ptr[i_r], end = v.csr_ptr[i_r + 1];
        int cnt = 0;
        for (int e0 = base; e0 < end; e0 += 128) {  // two chunks per trip, their three-deep load chains side by side
            int sx[2] = {-1, -1}, cx[2], lx[2];
            double ax[2];
            bool vx[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = e0 + 64 * h + lane;
                vx[h] = e < end;
                cx[h] = vx[h] ? v.csr_col[e] : 0;
                ax[h] = vx[h] ? v.csr_val[e] : 0.0;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) lx[h] = v.var_loc[cx[h]];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ks = v.kslot_of_pos[lx[h] >= 0 ? lx[h] : 0];
                if (vx[h] && lx[h] >= 0) sx[h] = ks;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && e0 + 64 >= end) break;  // (uniform)
                const int s = sx[h];
                bool isk = vx[h] && s >= 0;
                unsigned long long mask = __ballot(isk);
                const double coef = -ax[h] * inv;
                if (isk) {
                    int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                    v.blist_s[off] = s;
                    v.blist_a[off] = coef;
                }
                cnt += __popcll(mask);
                // e_j = U[j] . (listed rows): lane j serves pending term j, entries travel by shuffle in list order
                if (nlow > 0) lr_dot_listed(mask, s, coef, Urow, lane < nlow, lr_acc);
            }
        }
        if (lane == 0) it->blist_n = cnt;
    }
    if (nlow > 0) {  // delayed-update mode
        if (sr >= 0 && lane < nlow) lr_acc = Urow[sr];
        if (lane < nlow) c->lr_e[lane] = lr_acc;
    }
    if (plan_after && lane == 0) plan_update(v, c, phase);
}

// ------------------------------------------------------------------- K1 / K6: pricing
// K1 solver.rs:696-739: argmax over eligible non-basic columns of d^2/gamma (PSE) or |d| (Dantzig).
// K6 solver.rs:855-917: argmax over infeasible rows of infeas^2/beta.
__device__ __forceinline__ Cand price_primal_one(double dd, double gm, uint8_t f, int j, int use_pse) {
    if (((f & NB_AT_MIN) && dd > -EPS) || ((f & NB_AT_MAX) && dd < EPS)) return cand_none();  // solver.rs:705-706
    return Cand{use_pse ? dd * dd / gm : fabs(dd), j};
}
__device__ __forceinline__ Cand price_dual_one(double val, double mn, double mx, double bt, int r, int use_dse) {
    double infeas;
    if (val < mn - EPS) infeas = mn - val;
    else if (val > mx + EPS) infeas = val - mx;
    else return cand_none();
    return Cand{use_dse ? infeas * infeas / bt : infeas, r};
}
// single thread: open the next iteration with the pricing decision (derived scalars are filled by
// the head wave of the next kernel, after the kernel boundary has made this launch's stores visible)
__device__ __forceinline__ void open_iteration(Ctl* c, int phase, Cand best) {
    IterState* it = &c->it;
    it->klist_n = 0;
    it->blist_n = 0;
    c->up.kase = -1;
    if (best.idx == NONE_IDX) {
        it->status = phase == 0 ? ITER_OPTIMAL : ITER_FEASIBLE;
        it->q = -1;
        it->r = -1;
        c->halt = 1;
        push_rec(c, phase);
    } else {
        it->status = ITER_PIVOT;
        if (phase == 0) {
            it->q = best.idx;
            it->r = -1;
        } else {
            it->r = best.idx;
            it->q = -1;
        }
    }
}
// Finalising block of a pricing reduction: optionally close the current iteration (record), then
// open the next one with the (sharded: globally exchanged) pricing decision.
__device__ __forceinline__ void close_and_open(const DevView& v, Ctl* c, int phase, Cand best, double pay, bool close_current) {
    __shared__ double s_key, s_pay;
    __shared__ int s_idx, s_go;
    if (threadIdx.x == 0) {
        if (close_current) push_rec(c, phase);
        if (close_current && v.str_on) {  // sparse tableau row: a new epoch for the stamps, an empty list
            c->hyper_epoch += 1;
            c->str_n = 0;
            c->aq_n = 0;
        }
        s_go = 1;
        if (close_current && c->forced) {
            c->halt = 1;  // a host-forced iteration (fix_var) is a single step
            s_go = 0;
        }
        s_key = best.key;
        s_idx = best.idx;
        s_pay = pay;
    }
    __syncthreads();
    if (!s_go || threadIdx.x >= 64) return;
    Cand b{s_key, s_idx};
    bool ok = true;
    if (phase == 0) ok = exchange_best_wave(v, c, phase, b, s_pay, threadIdx.x);
    else if (threadIdx.x == 0) ok = adopt_rank0_candidate(v, c, b);
    if (threadIdx.x == 0 && ok) open_iteration(c, phase, b);
}
__global__ void __launch_bounds__(BLK) k_price_primal(DevView v, int use_pse) {
    Ctl* c = v.ctl;
    if (c->halt) return;
    Cand best = cand_none();
    double pay = 0.0;
    for (int j = v.nb_lo + blockIdx.x * BLK + threadIdx.x; j < v.nb_hi; j += gridDim.x * BLK) {
        double dj = v.d[j];
        Cand t = price_primal_one(dj, use_pse ? v.gamma[j] : 1.0, v.nbflags[j], j, use_pse);
        if (cand_better(t, best)) {
            best = t;
            pay = dj;
        }
    }
    if (!grid_best_p(best, pay, v)) return;
    close_and_open(v, c, 0, best, pay, false);
}
__global__ void __launch_bounds__(BLK) k_price_dual(DevView v, int use_dse) {
    Ctl* c = v.ctl;
    if (c->halt || c->forced) return;
    Cand best = cand_none();
    for (int r = blockIdx.x * BLK + threadIdx.x; r < v.m; r += gridDim.x * BLK) {
        Cand t = price_dual_one(v.xB[r], v.loB[r], v.hiB[r], use_dse ? v.beta[r] : 1.0, r, use_dse);
        if (cand_better(t, best)) best = t;
    }
    if (!grid_best(best, v)) return;
    if (threadIdx.x == 0 && adopt_rank0_candidate(v, c, best)) open_iteration(c, 1, best);
}

// ------------------------------------------------------------------- K2: FTRAN of one column
// alpha_q = B^-1 a_q  (solver.rs:671-677 -> 1305-1319 -> lu.rs:79-106).  B^-1 is held as a
// singleton split + dense nucleus inverse W (HISTORY.md §3.2), so the solve is:
//   head   : singleton rows of a_q land directly; entries on nucleus rows become a short list
//   gather : aK = W[:, list] * coeffs  (only the touched columns of W are read)
//            + push of -F*aK into the singleton positions (CSC columns of the nucleus basics)
__global__ void __launch_bounds__(64) k_ftran_prep(DevView v, int derive_primal) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) {
        if (threadIdx.x == 0) c->side_go = 0;
        return;
    }
    if (threadIdx.x == 0) c->side_go = 1;
    KMARK0(c, 0);
    ftran_prep_wave(v, c, threadIdx.x, derive_primal);
}
// push of -x * (column of the basic variable at `p`) into the singleton positions
template <int G>
__device__ __forceinline__ void push_F(const DevView& v, int p, double x, double* out_pos, int gl, Ctl* list_ctl = nullptr) {
    int var = v.basic_vars[p];
    int end = v.csc_ptr[var + 1];
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
        const RowInfo ri = v.rowinfo[v.csc_row[e]];
        if (ri.kslot < 0) {
            unsafeAtomicAdd(&out_pos[ri.pos], -v.csc_val[e] * x / ri.diag);
            if (list_ctl) aq_list_add(v, list_ctl, ri.pos);
        }
    }
}
// Deterministic FTRAN of a large model with a small nucleus (sharded solves of config 4: DESIGN.md §6): the pull per singleton
// row walks every entry of the row, 10^7 entries per pivot whatever the nucleus size.  The gather MARKS the singleton positions
// that a nucleus column with alpha_K != 0 reaches (plain idempotent stores); k_pull_F then skips the unmarked rows — their sum
// would be an exact zero — and clears the marks it consumes.  Same sums in the same order: results are bit-identical.
template <int G>
__device__ __forceinline__ void mark_F(const DevView& v, int p, int gl) {
    const int var = v.basic_vars[p];
    const int end = v.csc_ptr[var + 1];
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
        const RowInfo ri = v.rowinfo[v.csc_row[e]];
        if (ri.kslot < 0) v.fmark[ri.pos] = 1;
    }
}
template <int G>
__global__ void __launch_bounds__(BLK) k_ftran_gather(DevView v) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 18);
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (slot >= c->k) return;
    int n = c->it.klist_n;
    double acc = 0.0;
    const double* wrow = v.W + (size_t)slot * v.ld;
    if (G == 4) {
        // Four lanes per slot with the lane's loads of a trip issued together (round 4; the delayed-update mode with the blocked
        // push launches this instance whatever the model's lane count): with 64 lanes per slot the 32 pending terms U[j][slot] were
        // read one row per lane — 32 cache lines per slot, 5 120 workgroups — and the gather took 17 us at k = 20 500.
        for (int j0 = gl; j0 < n; j0 += 16) {
            double a[4], w[4];
            int si[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 4 * u;
                a[u] = j < n ? v.klist_a[j] : 0.0;
                si[u] = j < n ? v.klist_s[j] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = j0 + 4 * u < n ? wrow[si[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + 4 * u < n) acc += a[u] * w[u];
        }
        if (v.lrJ) {
            const int nlow = c->nlow;
            double e[LR_MAX / 4], x[LR_MAX / 4];
#pragma unroll
            for (int u = 0; u < LR_MAX / 4; ++u) {
                const int j = gl + 4 * u;
                e[u] = j < nlow ? c->lr_c[j] : 0.0;
                x[u] = j < nlow ? v.U[(size_t)j * v.ld + slot] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < LR_MAX / 4; ++u)
                if (gl + 4 * u < nlow) acc += x[u] * e[u];
        }
    } else {
        for (int j = gl; j < n; j += G) acc += v.klist_a[j] * wrow[v.klist_s[j]];
        if (v.lrJ) {
            const int nlow = c->nlow;
            for (int j = gl; j < nlow; j += G) acc += v.U[(size_t)j * v.ld + slot] * c->lr_c[j];
        }
    }
    acc = group_sum<G>(acc);
    int p = v.pos_of_kslot[slot];
    if (gl == 0) {
        v.aK[slot] = acc;
        v.alpha_q[p] = acc;
        if (acc != 0.0 && aq_listing(v)) aq_list_add(v, c, p);
    }
    if (acc != 0.0 && !v.pb_on && !v.det_pull) push_F<G>(v, p, acc, v.alpha_q, gl, aq_listing(v) ? c : nullptr);
    else if (acc != 0.0 && v.det_pull && v.fmark) mark_F<G>(v, p, gl);
}

// Blocked F push of the large-nucleus regime.  y_S -= D^-1 F x_K through device-scope f64 atomics costs
// ~10^6 memory-side atomics per call at k = 10^4 (~25 G/s: 45-55 us).  Here block (b, c) accumulates the
// entries of its slot range c that fall into row block b in LDS (ds_add_f64), the per-chunk sums go to
// push_part[c][row] with plain stores, and k_push_combine adds them up per singleton row: no global
// atomics, and the global summation order is fixed (only the LDS order within a block is not).
constexpr int PB_TILE = 512;  // slot descriptors staged in LDS per round
__global__ void __launch_bounds__(BLK) k_push_stage1(DevView v, int which) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 19);
    __shared__ double acc[PB_ROWS];
    __shared__ double s_x[PB_TILE];
    __shared__ int s_beg[PB_TILE], s_len[PB_TILE];
    // XCD-aware tile map (speed only): the row blocks of one slot chunk read the same columns — 25 workgroups, ~4 entries
    // of every column each — so they are placed on ONE XCD (block L runs on XCD L % 8), where the second to 25th reader
    // of a 128-byte line find it in that XCD's L2.  Needs the number of chunks to be a multiple of 8.
    int b = blockIdx.x, cc = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, idx = L >> 3;
        cc = (L & 7) + 8 * (idx / (int)gridDim.x);
        b = idx % (int)gridDim.x;
    }
    const int tid = threadIdx.x;
    const int row0 = b * PB_ROWS;
    const int nrows = min(PB_ROWS, v.m - row0);
    for (int t = tid; t < nrows; t += BLK) acc[t] = 0.0;
    const int k = c->k;
    const int per = (k + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_lo = cc * per, s_hi = min(k, s_lo + per);
    const double* xK = which ? v.tauK : v.aK;
    const int lane = tid & 7, grp = tid >> 3;  // 8 lanes per slot, 32 slots side by side
    const int stride = v.pb_rb + 1;
    for (int tile0 = s_lo; tile0 < s_hi; tile0 += PB_TILE) {
        const int nt = min(PB_TILE, s_hi - tile0);
        __syncthreads();  // acc zeroed / the previous tile's descriptors are no longer read
        // phase A: one thread per slot fetches (x, segment of the column inside this row block); the
        // three-deep dependent chain slot -> position -> variable -> offsets is paid once, in parallel
        for (int t = tid; t < nt; t += BLK) {
            const int slot = tile0 + t;
            const double x = xK[slot];
            const int var = v.basic_vars[v.pos_of_kslot[slot]];
            const int beg = v.colblk[(size_t)var * stride + b], end = v.colblk[(size_t)var * stride + b + 1];
            s_x[t] = x;
            s_beg[t] = beg;
            s_len[t] = (x != 0.0) ? end - beg : 0;
        }
        __syncthreads();
        // phase B: four slots per 8-lane group in flight (independent loads), then the LDS atomics
        for (int base = grp * 4; base < nt; base += (BLK / 8) * 4) {
            int r[4];
            double a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = base + j;
                const int len = t < nt ? s_len[t] : 0;
                r[j] = -1;
                a[j] = 0.0;
                if (lane < len) {
                    const int e = s_beg[t] + lane;
                    r[j] = v.csc_row[e] - row0;
                    a[j] = v.csc_val[e] * s_x[t];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r[j] >= 0) unsafeAtomicAdd(&acc[r[j]], a[j]);
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {  // segments longer than 8 entries (rare)
                const int t = base + j;
                const int len = t < nt ? s_len[t] : 0;
                for (int o = 8 + lane; o < len; o += 8) {
                    const int e = s_beg[t] + o;
                    unsafeAtomicAdd(&acc[v.csc_row[e] - row0], v.csc_val[e] * s_x[t]);
                }
            }
        }
    }
    __syncthreads();
    double* dst = v.push_part + (size_t)cc * v.m + row0;
    for (int t = tid; t < nrows; t += BLK) dst[t] = acc[t];
}
// Deterministic form of the same kernel (DevView.pb_det).  The only order-dependent arithmetic of k_push_stage1 is the sum
// an LDS accumulator receives through float atomics.  Here every product a * x is split into two fixed-point limbs relative
// to a bound of the chunk — 2^e > max |x_K| of the chunk's slots times max |A| —
//     hi = rint(term * 2^(P - e))                           (|hi| <= 2^P, P = min(52, 62 - pb_hbits): exact as a double)
//     lo = rint((term - hi * 2^(e - P)) * 2^(P - e + LS))   (the remainder is exact; |lo| <= 2^(LS - 1), LS = min(54, 62 - pb_hbits))
// and the limbs are added with INTEGER LDS atomics (ds_add_u64, two's complement): exact and associative, so the accumulated
// pair does not depend on the order in which the atomics land, and the value written to push_part —
// H * 2^(e - P) + L * 2^(e - P - LS), one rounding — is a pure function of the inputs.  Resolution 2^(e - P - LS): 106 bits
// below the largest possible term, so a row whose terms are all 10^-16 of the chunk's largest (the rounding noise of
// "structural zeros" of alpha_K) still keeps 53 significant bits — the componentwise backward error of the solve is that
// of the float sum (tests/test_late_regime.py checks it against the matrix itself).  Ranks of a sharded solve thereby stay
// bit-identical replicas without the pull over every singleton row (k_pull_F: nnz(A) work per call), and unsharded runs of
// the large-nucleus regime become reproducible bit for bit.
constexpr int PBD_TILE = 256;  // slot descriptors per round
constexpr int PBD_CHUNKS_DEFAULT = 16;  // slot chunks of the deterministic form
constexpr size_t PBD_LDS = sizeof(unsigned long long) * 2 * PB_ROWS + sizeof(double) * PBD_TILE + sizeof(int) * 2 * PBD_TILE;
__global__ void __launch_bounds__(BLK) k_push_stage1_det(DevView v, int which) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 19);
    extern __shared__ unsigned long long s_pbd[];  // 68 KB: two workgroups per CU
    unsigned long long* accH = s_pbd;                      // PB_ROWS
    unsigned long long* accL = s_pbd + PB_ROWS;            // PB_ROWS
    double* s_x = reinterpret_cast<double*>(s_pbd + 2 * PB_ROWS);  // PBD_TILE
    int* s_beg = reinterpret_cast<int*>(s_x + PBD_TILE);   // PBD_TILE
    int* s_len = s_beg + PBD_TILE;                         // PBD_TILE
    __shared__ double s_bound;
    int b = blockIdx.x, cc = blockIdx.y;
    if ((gridDim.y & 7) == 0) {  // XCD-aware tile map (see k_push_stage1)
        const int L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, idx = L >> 3;
        cc = (L & 7) + 8 * (idx / (int)gridDim.x);
        b = idx % (int)gridDim.x;
    }
    const int tid = threadIdx.x;
    const int row0 = b * PB_ROWS;
    const int nrows = min(PB_ROWS, v.m - row0);
    for (int t = tid; t < nrows; t += BLK) {
        accH[t] = 0ull;
        accL[t] = 0ull;
    }
    const int k = c->k;
    const int per = (k + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_lo = cc * per, s_hi = min(k, s_lo + per);
    const double* xK = which ? v.tauK : v.aK;
    {   // bound of the chunk: max |x_K| over its slots (max is order-independent: every row block of the chunk finds the same)
        double mx = 0.0;
        for (int s = s_lo + tid; s < s_hi; s += BLK) {
            const double ax = fabs(xK[s]);
            mx = (ax == ax) ? fmax(mx, ax) : INFINITY;  // (fmax drops a NaN: a NaN slot must poison the bound like an infinite one)
        }
        mx = block_max(mx);
        if (tid == 0) s_bound = mx * v.pb_amax;
    }
    __syncthreads();
    const double bound = s_bound;
    double* dst = v.push_part + (size_t)cc * v.m + row0;
    if (!(bound > 0.0)) {  // nothing to push
        for (int t = tid; t < nrows; t += BLK) dst[t] = 0.0;
        return;
    }
    if (!(bound < INFINITY)) {  // a non-finite x_K (broken inverse, overflow): PROPAGATE it as the float form would, so that the
                                // drift monitor / polish checks see the breakdown instead of a silent zero
        for (int t = tid; t < nrows; t += BLK) dst[t] = NAN;
        return;
    }
    int e;
    (void)frexp(bound, &e);  // bound < 2^e
    if (e < -900) e = -900;  // (keeps every scale below a finite power of two)
    const int P = min(52, 62 - v.pb_hbits), LS = min(54, 62 - v.pb_hbits);
    const double S1 = ldexp(1.0, P - e), iS1 = ldexp(1.0, e - P), SL = ldexp(1.0, P - e + LS), iSL = ldexp(1.0, e - P - LS);
    const int lane = tid & 7, grp = tid >> 3;  // 8 lanes per slot, 32 slots side by side
    const int stride = v.pb_rb + 1;
    for (int tile0 = s_lo; tile0 < s_hi; tile0 += PBD_TILE) {
        const int nt = min(PBD_TILE, s_hi - tile0);
        __syncthreads();
        for (int t = tid; t < nt; t += BLK) {  // phase A: one thread per slot fetches (x, segment of the column inside this row block)
            const int slot = tile0 + t;
            const double x = xK[slot];
            const int var = v.basic_vars[v.pos_of_kslot[slot]];
            const int beg = v.colblk[(size_t)var * stride + b], end = v.colblk[(size_t)var * stride + b + 1];
            s_x[t] = x;
            s_beg[t] = beg;
            s_len[t] = (x != 0.0) ? end - beg : 0;
        }
        __syncthreads();
        for (int base = grp * 4; base < nt; base += (BLK / 8) * 4) {  // phase B: four slots per 8-lane group in flight
            int r[4];
            double a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = base + j;
                const int len = t < nt ? s_len[t] : 0;
                r[j] = -1;
                a[j] = 0.0;
                if (lane < len) {
                    const int en = s_beg[t] + lane;
                    r[j] = v.csc_row[en] - row0;
                    a[j] = v.csc_val[en] * s_x[t];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r[j] >= 0) {
                    const long long hi = __double2ll_rn(a[j] * S1);
                    const long long lo = __double2ll_rn((a[j] - (double)hi * iS1) * SL);
                    atomicAdd(&accH[r[j]], (unsigned long long)hi);
                    atomicAdd(&accL[r[j]], (unsigned long long)lo);
                }
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {  // segments longer than 8 entries (rare)
                const int t = base + j;
                const int len = t < nt ? s_len[t] : 0;
                for (int o = 8 + lane; o < len; o += 8) {
                    const int en = s_beg[t] + o;
                    const double term = v.csc_val[en] * s_x[t];
                    const long long hi = __double2ll_rn(term * S1);
                    const long long lo = __double2ll_rn((term - (double)hi * iS1) * SL);
                    atomicAdd(&accH[v.csc_row[en] - row0], (unsigned long long)hi);
                    atomicAdd(&accL[v.csc_row[en] - row0], (unsigned long long)lo);
                }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < nrows; t += BLK) dst[t] = (double)(long long)accH[t] * iS1 + (double)(long long)accL[t] * iSL;
}
__global__ void __launch_bounds__(BLK) k_push_combine(DevView v, int which, int nchunks, int ys = 0) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 20);
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= v.m) return;
    const RowInfo ri = v.rowinfo[i];
    if (ri.kslot >= 0) return;
    double s = 0.0;
    for (int c0 = 0; c0 < nchunks; c0 += 8) {  // eight independent loads in flight, summed in chunk order
        double pp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pp[u] = (c0 + u < nchunks) ? v.push_part[(size_t)(c0 + u) * v.m + i] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += pp[u];
    }
    if (ys) {
        // v branch: the singleton part y_S = alpha_S / diag of v = B^-T alpha_q (solver.rs:1114), by row, is left behind for t_K
        // (the ratio test, which otherwise forms it, now runs beside the kernels that need it: the same quotient, bit for bit)
        const double a = v.alpha_q[ri.pos] - (s != 0.0 ? s / ri.diag : 0.0);
        if (s != 0.0) v.alpha_q[ri.pos] = a;
        v.rv[i].y = a / ri.diag;
        return;
    }
    if (s != 0.0) {
        double* out = which ? v.tau : v.alpha_q;
        out[ri.pos] -= s / ri.diag;
    }
}
// Deterministic form of the same product for small models: y_S[p] -= (sum over the nucleus columns in row
// i_p of A[i_p, col] * x_K[slot(col)]) / diag_p, pulled through the CSR row with a fixed reduction tree.
// It touches every entry of the singleton rows per call (nnz(A) work instead of nnz of the nucleus columns),
// which is affordable only while the matrix is small; in exchange two runs of the same solve take the
// same pivots, which matters on degenerate models (a branch-and-bound tree over TSP relaxations varied
// between 460 and 10 700 nodes from run to run with the atomics).
template <int G>
__global__ void __launch_bounds__(BLK) k_pull_F(DevView v, int which) {
    const Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int p = (blockIdx.x * BLK + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (p >= v.m || v.kslot_of_pos[p] >= 0) return;
    if (!which && v.fmark) {  // (the G lanes of a position read the mark before lane 0 clears it: same wave, in order)
        const unsigned char mk = v.fmark[p];
        if (!mk) return;
        if (gl == 0) v.fmark[p] = 0;
    }
    const int i = v.srow_of_pos[p];
    const double* xK = which ? v.tauK : v.aK;
    double acc = 0.0;
    const int end = v.csr_ptr[i + 1];
    for (int e = v.csr_ptr[i] + gl; e < end; e += G) {
        const int loc = v.var_loc[v.csr_col[e]];
        if (loc >= 0) {
            const int s = v.kslot_of_pos[loc];
            if (s >= 0) acc += v.csr_val[e] * xK[s];
        }
    }
    acc = group_sum<G>(acc);
    if (gl == 0 && acc != 0.0) {
        double* out = which ? v.tau : v.alpha_q;
        out[p] -= acc / v.sdiag_of_pos[p];
    }
}
// The marked pull of the FTRAN (which = 0 with marks) at 64 lanes per row, in scan form: k_pull_F<64> launches m / 4 blocks — 25 000
// on config 4 — of which all but the few holding a marked row return at once; the grid alone made the sharded early FTRAN cost 64 us
// against 11 unsharded (round-3 review, weak 9).  Here 2 048 waves stride over the positions 64 at a time: one coalesced read of the
// mark bytes, a ballot, then the whole wave walks each marked row exactly as the 64 lanes of k_pull_F<64> do (same lane -> entry
// assignment, same xor tree: the same bits).
__global__ void __launch_bounds__(BLK) k_pull_F_scan(DevView v) {
    const Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int lane = threadIdx.x & 63;
    const int nw = (int)gridDim.x * (BLK / 64);
    for (int chunk = (int)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6); chunk * 64 < v.m; chunk += nw) {
        const int p0 = chunk * 64 + lane;
        unsigned char mk = 0;
        if (p0 < v.m) {
            mk = v.fmark[p0];
            if (mk) {
                v.fmark[p0] = 0;
                if (v.kslot_of_pos[p0] >= 0) mk = 0;  // (k_pull_F leaves such a mark in place; nothing sets one)
            }
        }
        unsigned long long mask = __ballot(mk != 0);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int p = chunk * 64 + b;
            const int i = v.srow_of_pos[p];
            double acc = 0.0;
            const int end = v.csr_ptr[i + 1];
            for (int e = v.csr_ptr[i] + lane; e < end; e += 64) {
                const int loc = v.var_loc[v.csr_col[e]];
                if (loc >= 0) {
                    const int s = v.kslot_of_pos[loc];
                    if (s >= 0) acc += v.csr_val[e] * v.aK[s];
                }
            }
            acc = group_sum<64>(acc);
            if (lane == 0 && acc != 0.0) v.alpha_q[p] -= acc / v.sdiag_of_pos[p];
        }
    }
}
static void launch_pull_F(const DevView& dv, const Geom& g, int which, hipStream_t st) {
    if (!which && dv.fmark && g.lanes > 16 && dv.m >= 16384) {
        hipLaunchKernelGGL(k_pull_F_scan, dim3(512), dim3(BLK), 0, st, dv);
        return;
    }
    if (g.lanes <= 4) hipLaunchKernelGGL(k_pull_F<4>, dim3(blocks_for((long)dv.m * 4)), dim3(BLK), 0, st, dv, which);
    else if (g.lanes <= 16) hipLaunchKernelGGL(k_pull_F<16>, dim3(blocks_for((long)dv.m * 16)), dim3(BLK), 0, st, dv, which);
    else hipLaunchKernelGGL(k_pull_F<64>, dim3(blocks_for((long)dv.m * 64)), dim3(BLK), 0, st, dv, which);
}
static void launch_blocked_push(const DevView& dv, int which, hipStream_t st, int ys = 0) {
    // (a band form of this push — LDS blocks of the band-major copy — was measured in round 2: 34.6 + 9.5 us against 34.8 + 6.4 us,
    // neither bound by its traffic; removed.  Round 6: in the lazy primal iteration the product is PULLED instead: fpull.inc)
    // slot chunks: as many as keep (row blocks x chunks) within one round of workgroups (3 per CU by LDS), at most PB_CHUNKS
    const int want = 0;
    int chunks = PB_CHUNKS_DEFAULT;  // (24 measured best: 727 / 731 / 732 / 746 us per pivot at k = 20 500 for 24 / 30 / 36 / 48)
    if (chunks > PB_CHUNKS) chunks = PB_CHUNKS;
    if (dv.pb_det) {  // fixed-point limbs: order-independent
        static bool det_attr = false;
        if (!det_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_push_stage1_det), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PBD_LDS);
            det_attr = true;
        }
        if (want <= 0) chunks = PBD_CHUNKS_DEFAULT;  // (two workgroups per CU by LDS: row blocks x chunks should fit one round)
        hipLaunchKernelGGL(k_push_stage1_det, dim3(dv.pb_rb, chunks), dim3(BLK), PBD_LDS, st, dv, which);
    } else hipLaunchKernelGGL(k_push_stage1, dim3(dv.pb_rb, chunks), dim3(BLK), 0, st, dv, which);
    hipLaunchKernelGGL(k_push_combine, dim3((dv.m + BLK - 1) / BLK), dim3(BLK), 0, st, dv, which, chunks, ys);
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
// pass 1 (solver.rs:782-795); with PSE also ||alpha_q||^2 (solver.rs:1136) and the singleton part
// y_S of v = B^-T alpha_q (solver.rs:1114), both of which need the same stream over alpha_q.
__global__ void __launch_bounds__(BLK) k_ratio_primal_p1(DevView v, int use_pse) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    int sign = c->it.sign;
    double mn = INFINITY, sq = 0.0;
    for (int p = blockIdx.x * BLK + threadIdx.x; p < v.m; p += gridDim.x * BLK) {
        double coeff = v.alpha_q[p];
        if (use_pse) {
            sq += coeff * coeff;
            if (v.kslot_of_pos[p] < 0) v.rv[v.srow_of_pos[p]].y = coeff / v.sdiag_of_pos[p];
        }
        double ca = fabs(coeff);
        if (ca < EPS) continue;
        bool tm;
        double cur = (leaving_step(v, p, coeff, sign, tm) + EPS) / ca;
        if (cur < mn) mn = cur;
    }
    if (!grid_min_sum(mn, sq, v)) return;
    if (threadIdx.x == 0) {
        double max_step = fabs(c->it.entering_other - c->it.entering_cur);
        if (mn < max_step) max_step = mn;
        c->it.max_step = max_step;
        c->it.alpha_sq = sq + 1.0;
    }
}
// Finalising block of the primal ratio test: the decision (solver.rs:820-853), in sharded mode the adoption
// of rank 0's decision, then the BTRAN head and the partition plan.
__device__ void ratio_primal_finish(const DevView& v, Ctl* c, Cand best) {
    IterState* it = &c->it;
    const int sign = it->sign;
    __shared__ int s_r;
    if (threadIdx.x == 0) {
        const int q = it->q;
        const double dq = v.d[q];
        // decide locally without side effects ...
        int status = ITER_PIVOT, r = -1;
        double coeff = 0.0, lnv = 0.0, diff = 0.0, enew = 0.0, pobj = 0.0;
        if (best.idx != NONE_IDX) {
            r = best.idx;
            coeff = ld_agent(&v.alpha_q[r]);
            bool tm;
            leaving_step(v, r, coeff, sign, tm);
            lnv = tm ? v.hiB[r] : v.loB[r];
            diff = (v.xB[r] - lnv) / coeff;  // solver.rs:828
            enew = it->entering_cur + diff;
            pobj = dq / coeff;               // solver.rs:1073
        } else if (isinf(it->entering_other)) {
            status = ITER_UNBOUNDED;         // solver.rs:842-844
        } else {
            status = ITER_FLIP;              // solver.rs:846-851
            diff = it->entering_other - it->entering_cur;
            enew = it->entering_other;
        }
        // ... in sharded mode every rank adopts rank 0's decision (alpha_q is replicated, but its
        // float atomics make it reproducible only to rounding; the decision must be identical)
        bool ok = true;
        if (v.world > 1) {
            const unsigned long long ep = ++c->xepoch[1];
            if (v.rank == 0) {
                double f[7] = {(double)status, (double)r, coeff, lnv, diff, enew, pobj};
                mail_post(v, 1, ep, f);
            } else {
                double h[7];
                ok = mail_wait(mail_slot(v, 1, ep, 0), ep, h);
                if (ok) {
                    status = (int)h[0]; r = (int)h[1]; coeff = h[2]; lnv = h[3]; diff = h[4]; enew = h[5]; pobj = h[6];
                } else {
                    comm_fail(c, 0);
                }
            }
        }
        s_r = -1;
        if (ok) {
            it->status = status;
            it->r = r;
            it->pivot_coeff = coeff;
            it->leaving_new_val = lnv;
            it->entering_diff = diff;
            it->entering_new_val = enew;
            it->pivot_obj = pobj;
            if (status == ITER_UNBOUNDED) {
                c->halt = 1;
                push_rec(c, 0);
            } else {
                c->sb_obj0 = it->obj;
                it->obj += dq * diff;  // solver.rs:1027
                if (status == ITER_PIVOT) {
                    it->leaving_var = v.basic_vars[r];
                    s_r = r;
                }
            }
        }
    }
    KMARK(c, 5);
    __syncthreads();
    if (s_r >= 0) {
        // the BTRAN head (wave 0) and the partition plan (first lane of wave 1) are independent chains of dependent
        // loads over disjoint outputs: side by side they cost the longer of the two instead of their sum
        if (threadIdx.x < 64) btran_prep_wave(v, c, threadIdx.x, s_r, 0, 0, 0);
        else if (threadIdx.x == 64) plan_update(v, c, 0);
    }
}

// pass 2 (solver.rs:800-853); the finalising block goes straight on with the BTRAN head + plan
__global__ void __launch_bounds__(BLK) k_ratio_primal_p2(DevView v) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    IterState* it = &c->it;
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
    ratio_primal_finish(v, c, best);
}

// Small models (<= 16 384 positions, one GPU): both primal Harris passes, ||alpha_q||^2 and the singleton part of v in ONE BLOCK
// (two loops over alpha_q; the second re-reads from L2).
__global__ void __launch_bounds__(BLK) k_ratio_primal_one(DevView v, int use_pse) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int sign = c->it.sign;
    double mn = INFINITY, sq = 0.0;
    for (int p0 = threadIdx.x; p0 < v.m; p0 += 4 * BLK) {
        double co[4], xb[4], lob[4], hib[4], sd[4];
        int ks[4], sr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * BLK;
            const int pc = p < v.m ? p : 0;
            co[u] = p < v.m ? v.alpha_q[pc] : 0.0;
            xb[u] = v.xB[pc];
            lob[u] = v.loB[pc];
            hib[u] = v.hiB[pc];
            ks[u] = v.kslot_of_pos[pc];
            sr[u] = v.srow_of_pos[pc];
            sd[u] = v.sdiag_of_pos[pc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * BLK >= v.m) continue;
            if (use_pse) {
                sq += co[u] * co[u];
                if (ks[u] < 0) v.rv[sr[u]].y = co[u] / sd[u];
            }
            const double a = fabs(co[u]);
            if (a < EPS) continue;
            const bool tm = (sign && co[u] < 0.0) || (!sign && co[u] > 0.0);  // solver.rs:752-771 (as leaving_step)
            const double st = tm ? (xb[u] < hib[u] ? hib[u] - xb[u] : 0.0) : (xb[u] > lob[u] ? xb[u] - lob[u] : 0.0);
            const double cur = (st + EPS) / a;
            if (cur < mn) mn = cur;
        }
    }
    __shared__ double s_ms;
    mn = block_min(mn);
    sq = block_sum(sq);
    if (threadIdx.x == 0) {
        double max_step = fabs(c->it.entering_other - c->it.entering_cur);
        if (mn < max_step) max_step = mn;
        c->it.max_step = max_step;
        c->it.alpha_sq = sq + 1.0;
        s_ms = max_step;
    }
    __syncthreads();
    const double max_step = s_ms;
    Cand best = cand_none();
    for (int p0 = threadIdx.x; p0 < v.m; p0 += 4 * BLK) {
        double co[4], xb[4], lob[4], hib[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * BLK;
            const int pc = p < v.m ? p : 0;
            co[u] = p < v.m ? v.alpha_q[pc] : 0.0;
            xb[u] = v.xB[pc];
            lob[u] = v.loB[pc];
            hib[u] = v.hiB[pc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // ascending positions per thread: ties keep the lowest position
            const double a = fabs(co[u]);
            if (a < EPS) continue;
            const bool tm = (sign && co[u] < 0.0) || (!sign && co[u] > 0.0);
            const double st = tm ? (xb[u] < hib[u] ? hib[u] - xb[u] : 0.0) : (xb[u] > lob[u] ? xb[u] - lob[u] : 0.0);
            if (st / a <= max_step) {
                Cand t{a, p0 + u * BLK};
                if (cand_better(t, best)) best = t;
            }
        }
    }
    best = block_best(best);
    ratio_primal_finish(v, c, best);
}
constexpr int AQ_CAP = 1024;  // positions of supp(alpha_q) the single-block form of the primal ratio test takes on (8 per thread: beyond that
                              // the two grid-wide passes are faster — measured on config 4: 97 against 77 us per pivot at k = 135 with a cap of 8 192)
// Both Harris passes in ONE launch (primal): pass 1's grid-wide minimum is published by its last-arriving
// block, every block waits for it (a 98-block grid is always co-resident) and runs pass 2 on the elements it
// still holds in registers; the second ticketed reduction ends in ratio_primal_finish as before.  Saves a
// kernel boundary and the second read of alpha_q / x_B / bounds.
// t_K = alpha_K - F^T y_S (solver.rs:1114) for the slots of one block, G lanes per slot, as k_btran's second part — for the
// blocks that ride behind the ratio blocks of k_ratio_primal_fused (below)
// ONFLY (small nucleus: the ratio test of the SAME launch is the one that writes y_S by row): y_S[i] = alpha_q[pos] / diag of the
// singleton covering row i is formed from the packed row map — the quotient the ratio test writes, zero where alpha_q is zero
template <int G, bool ONFLY>
__device__ __forceinline__ void tk_ride_body(const DevView& v, const Ctl* c, int block) {
    const int slot = (block * BLK + (int)threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (slot >= c->k) return;
    const int p = v.pos_of_kslot[slot];
    const int var = v.basic_vars[p];
    const int end = v.csc_ptr[var + 1];
    double acc = 0.0;
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
        double y;
        if (ONFLY) {
            const RowInfo ri = v.rowinfo[v.csc_row[e]];
            const double a = ri.kslot < 0 ? v.alpha_q[ri.pos] : 0.0;
            y = a != 0.0 ? a / ri.diag : 0.0;
        } else {
            y = v.rv[v.csc_row[e]].y;  // (zero on nucleus rows, see k_btran)
        }
        acc += v.csc_val[e] * y;
    }
    acc = group_sum<G>(acc);
    if (gl == 0) v.tK[slot] = v.alpha_q[p] - acc;
}
// n_ratio > 0 (large nucleus, lazy primal iteration; launch_ratio_primal): only the first n_ratio blocks run the ratio test; the
// blocks behind them form t_K, which needs alpha_q and y_S but not the leaving row — the combine of the blocked F push has left
// y_S by row (its ys form: the quotient the ratio test would write, use_pse = 2 tells pass 1 not to), so the 2 M-entry pull that
// made k_btran a 23 us kernel runs in the shadow of the ratio test's two grid-wide hand-offs.  The ratio blocks come first in
// dispatch order and wait only for each other: the co-residency argument of the in-kernel wait is unchanged.
__global__ void __launch_bounds__(BLK) k_ratio_primal_fused(DevView v, int use_pse, int n_ratio = 0, int tk_lanes = 0, int tk_onfly = 0) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int nrb = n_ratio > 0 ? n_ratio : (int)gridDim.x;
    if ((int)blockIdx.x >= nrb) {
        const int tb = (int)blockIdx.x - nrb;
        if (tk_onfly) {  // (small nucleus: k_small_basis then carries no t_K blocks and waits for none)
            if (tk_lanes <= 4) tk_ride_body<4, true>(v, c, tb);
            else if (tk_lanes <= 16) tk_ride_body<16, true>(v, c, tb);
            else tk_ride_body<64, true>(v, c, tb);
        } else {
            if (tk_lanes <= 4) tk_ride_body<4, false>(v, c, tb);
            else if (tk_lanes <= 16) tk_ride_body<16, false>(v, c, tb);
            else tk_ride_body<64, false>(v, c, tb);
        }
        return;
    }
    const int sign = c->it.sign;
    KMARK0(c, 1);
    if (aq_listing(v) && c->aq_n <= AQ_CAP) {
        // Sparse form: block 0 alone runs both Harris passes (solver.rs:782-853), ||alpha_q||^2 and the singleton part of v
        // over the listed positions of supp(alpha_q); no grid-wide reduction, no in-kernel wait.  Ties keep the lowest
        // position (cand_better), whatever the order of the list.
        if (blockIdx.x != 0) return;
        const int n_l = c->aq_n;
        constexpr int PL = AQ_CAP / BLK;
        double ca[PL], stp[PL];
        int pos[PL];
        double mn = INFINITY, sq = 0.0;
        // every input of the thread's (up to PL) listed positions is loaded before anything is stored: with the loads behind the
        // per-element `continue` and the rv.y stores between them (possible aliases for the compiler) the PL elements became PL
        // serial round trips of two dependent levels each — measured with MLP_KPROF: 20.5 us from kernel entry to "loads done"
        // at k = 107, of a 76 us iteration
        double co[PL], xbv[PL], lov[PL], hiv[PL], sdv[PL];
        int ksv[PL], srv[PL];
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int a = threadIdx.x + u * BLK;
            pos[u] = a < n_l ? v.aq_list[a] : -1;
        }
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int p = pos[u] < 0 ? 0 : pos[u];
            co[u] = v.alpha_q[p];
            ksv[u] = v.kslot_of_pos[p];
            srv[u] = v.srow_of_pos[p];
            sdv[u] = v.sdiag_of_pos[p];
            xbv[u] = v.xB[p];
            lov[u] = v.loB[p];
            hiv[u] = v.hiB[p];
        }
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            ca[u] = 0.0;
            stp[u] = 0.0;
            if (pos[u] < 0) continue;
            const double coeff = co[u];
            if (use_pse) {
                sq += coeff * coeff;
                if (ksv[u] < 0) v.rv[srv[u]].y = coeff / sdv[u];
            }
            const double aa = fabs(coeff);
            if (aa < EPS) {
                pos[u] = -1;
                continue;
            }
            const bool tm = (sign && coeff < 0.0) || (!sign && coeff > 0.0);
            const double st = tm ? (xbv[u] < hiv[u] ? hiv[u] - xbv[u] : 0.0) : (xbv[u] > lov[u] ? xbv[u] - lov[u] : 0.0);
            ca[u] = aa;
            stp[u] = st;
            const double cur = (st + EPS) / aa;
            if (cur < mn) mn = cur;
        }
        __shared__ double s_ms;
        KMARK(c, 2);
        mn = block_min(mn);
        sq = block_sum(sq);
        if (threadIdx.x == 0) {
            double max_step = fabs(c->it.entering_other - c->it.entering_cur);
            if (mn < max_step) max_step = mn;
            c->it.max_step = max_step;
            c->it.alpha_sq = sq + 1.0;
            s_ms = max_step;
        }
        __syncthreads();
        KMARK(c, 3);
        const double max_step = s_ms;
        Cand best = cand_none();
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            if (pos[u] < 0) continue;
            if (stp[u] / ca[u] <= max_step) {
                Cand t{ca[u], pos[u]};
                if (cand_better(t, best)) best = t;
            }
        }
        best = block_best(best);
        KMARK(c, 4);
        ratio_primal_finish(v, c, best);
        KMARK(c, 6);
        return;
    }
    const int epoch0 = c->ratio_epoch;  // written by the previous launch of this kernel: stable here
    constexpr int PT = 4;               // elements per thread (grid_for's default)
    double ca[PT], stp[PT];
    int pos[PT];
    double mn = INFINITY, sq = 0.0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int p = (int)blockIdx.x * BLK + threadIdx.x + u * nrb * BLK;
        pos[u] = -1;
        ca[u] = 0.0;
        stp[u] = 0.0;
        if (p >= v.m) continue;
        // every input of the element is loaded up front (independent loads in flight together) instead of behind the
        // branches that use it: one dependent memory round trip less on the critical path of the iteration
        const double coeff = v.alpha_q[p];
        const int ks = v.kslot_of_pos[p], sr = v.srow_of_pos[p];
        const double sd = v.sdiag_of_pos[p];
        const double xb = v.xB[p], lob = v.loB[p], hib = v.hiB[p];
        if (use_pse) {
            sq += coeff * coeff;
            if (ks < 0 && use_pse != 2) v.rv[sr].y = coeff / sd;
        }
        const double a = fabs(coeff);
        if (a < EPS) continue;
        const bool tm = (sign && coeff < 0.0) || (!sign && coeff > 0.0);  // solver.rs:752-771 (as leaving_step)
        const double st = tm ? (xb < hib ? hib - xb : 0.0) : (xb > lob ? xb - lob : 0.0);
        pos[u] = p;
        ca[u] = a;
        stp[u] = st;
        const double cur = (st + EPS) / a;
        if (cur < mn) mn = cur;
    }
    __shared__ double s_max_step;
    if (grid_min_sum(mn, sq, v, nrb)) {  // last arriver of pass 1: publish the step bound
        if (threadIdx.x == 0) {
            double max_step = fabs(c->it.entering_other - c->it.entering_cur);
            if (mn < max_step) max_step = mn;
            c->it.max_step = max_step;
            c->it.alpha_sq = sq + 1.0;
            st_agent(&c->ratio_max_step, max_step);
            // the ticket is reused by pass 2 INSIDE this launch: its reset must be visible to the other blocks'
            // atomics (grid_min_sum resets it with a plain store, which only a kernel boundary publishes)
            __hip_atomic_store(v.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_agent(&c->ratio_epoch, epoch0 + 1);
        }
    }
    __shared__ int s_gave_up;
    if (threadIdx.x == 0) {
        s_gave_up = 0;
        long spins = 0;
        while (ld_agent(&c->ratio_epoch) != epoch0 + 1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > c->ratio_spin_limit) {  // seconds: cannot happen while the grid is co-resident; never hang the GPU
                s_gave_up = 1;
                break;
            }
        }
        s_max_step = ld_agent(&c->ratio_max_step);
    }
    __syncthreads();
    if (s_gave_up) {
        // ANY block that gives up declares the stall, the first one writes the record (atomic on halt): block 0 may be the one
        // block that did not give up — the last arriver of pass 1 — and then nobody would (the iteration went on with a stale
        // leaving row; found when a diagnostic mark delayed block 0: GPU memory fault in the update kernel)
        if (threadIdx.x == 0 && atomicExch(&c->halt, 1) == 0) {
            c->it.status = ITER_STALL;  // not a property of the model: the grid was not co-resident (see launch_ratio_primal)
            push_rec(c, 0);
        }
        return;
    }
    const double max_step = s_max_step;
    Cand best = cand_none();
#pragma unroll
    for (int u = 0; u < PT; ++u) {  // ascending positions per thread, so ties keep the lowest position
        if (pos[u] < 0) continue;
        if (stp[u] / ca[u] <= max_step) {
            Cand t{ca[u], pos[u]};
            if (cand_better(t, best)) best = t;
        }
    }
    if (!grid_best(best, v, nrb)) return;
    ratio_primal_finish(v, c, best);
}

// dual path, after FTRAN: the FTRAN-side pivot, ||alpha_q||^2 and y_S (PSE), then the plan
__global__ void __launch_bounds__(BLK) k_post_ftran(DevView v, int use_pse) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    double sq = 0.0;
    if (use_pse) {
        for (int p = blockIdx.x * BLK + threadIdx.x; p < v.m; p += gridDim.x * BLK) {
            double coeff = v.alpha_q[p];
            sq += coeff * coeff;
            if (!v.fac_on && v.kslot_of_pos[p] < 0) v.rv[v.srow_of_pos[p]].y = coeff / v.sdiag_of_pos[p];
        }
    }
    if (!grid_sum(sq, v)) return;
    if (threadIdx.x == 0) {
        c->it.alpha_sq = sq + 1.0;
        plan_update(v, c, 1);
    }
}

// ------------------------------------------------------------------- K3: BTRAN of a unit vector
// rho = B^-T e_r (solver.rs:680-683 -> 1322-1338).  With W explicit this is one row of W when r
// is a nucleus position, or a short combination of rows (those nucleus columns that have an
// entry in the leaving singleton's row, read from the CSR row) otherwise.
__global__ void __launch_bounds__(64) k_btran_prep(DevView v, int derive_dual, int plan_after) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    btran_prep_wave(v, c, threadIdx.x, c->it.r, derive_dual, plan_after, 1);
}
// Horizontally fused: blocks [0, n_gather) do rK = sum_j blist_a[j] * W[blist_s[j], :], the rho
// scatter and ||rho||^2; the remaining blocks (PSE) build tK = alpha_K - F^T y_S (solver.rs:1114).
// rho_K = sum_j blist_a[j] W[blist_s[j], :] (+ the pending terms of the delayed-update mode), rho by row, ||rho||^2: the first part
// of k_btran, for `n_gather` blocks numbered `block` — of k_btran itself, or riding behind the v tail of k_post_fused (round 4)
__device__ __forceinline__ void btran_rk_body(const DevView& v, Ctl* c, int block, int n_gather, int after_fold) {
    const int k = c->k;
    const int n = c->it.blist_n;
    double sq = 0.0;
    if (v.lrJ) {
        // delayed-update mode (large nucleus): four lanes share a slot — up to 20 listed rows plus 32 pending terms are
        // 52 loads per slot, a serial chain for one lane (fixed summation order: lane-strided, then two shuffles)
        // (v branch: this launch waited for the fold of a folding pivot — W0 then holds every term)
        const int nlow = (after_fold && c->fold) ? 0 : c->nlow;
        for (int g4 = block * BLK + (int)threadIdx.x; g4 < 4 * k; g4 += n_gather * BLK) {  // (whole groups of 4 lanes)
            const int s = g4 >> 2, gl = g4 & 3;
            double acc = 0.0;
            // (round 4: the lane's loads of a trip are issued together — four listed rows, then its eight pending terms —
            // and added in the old order: as one load -> add chain per term the 13 dependent round trips were the kernel,
            // 22 us at k = 20 500 for 8 MB)
            for (int j0 = gl; j0 < n; j0 += 16) {
                double a[4], w[4];
                int si[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 4 * u;
                    a[u] = j < n ? v.blist_a[j] : 0.0;
                    si[u] = j < n ? v.blist_s[j] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = j0 + 4 * u < n ? v.W[(size_t)si[u] * v.ld + s] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j0 + 4 * u < n) acc += a[u] * w[u];
            }
            {
                double e[LR_MAX / 4], x[LR_MAX / 4];
#pragma unroll
                for (int u = 0; u < LR_MAX / 4; ++u) {
                    const int j = gl + 4 * u;
                    e[u] = j < nlow ? c->lr_e[j] : 0.0;
                    x[u] = j < nlow ? v.V[(size_t)j * v.ld + s] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < LR_MAX / 4; ++u)
                    if (gl + 4 * u < nlow) acc += e[u] * x[u];
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (gl == 0) {
                v.rK[s] = acc;
                v.rv[v.row_of_kslot[s]].x = acc;
                sq += acc * acc;
            }
        }
    } else
    for (int s = block * BLK + (int)threadIdx.x; s < k; s += n_gather * BLK) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += v.blist_a[j] * v.W[(size_t)v.blist_s[j] * v.ld + s];
        v.rK[s] = acc;
        v.rv[v.row_of_kslot[s]].x = acc;
        sq += acc * acc;
    }
    if (!grid_sum(sq, v, n_gather, block)) return;
    if (threadIdx.x == 0) {
        int r = c->it.r;
        if (v.kslot_of_pos[r] < 0) {
            double inv = 1.0 / v.sdiag_of_pos[r];
            sq += inv * inv;
        }
        c->it.rho_sq = sq;
    }
}
template <int G>
__global__ void __launch_bounds__(BLK) k_btran(DevView v, int n_gather, int after_fold = 0) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 7);
    const int k = c->k;
    if ((int)blockIdx.x < n_gather) {
        btran_rk_body(v, c, (int)blockIdx.x, n_gather, after_fold);
    } else {
        int slot = (((int)blockIdx.x - n_gather) * BLK + threadIdx.x) / G;
        int gl = threadIdx.x & (G - 1);
        if (slot >= k) return;
        int p = v.pos_of_kslot[slot];
        int var = v.basic_vars[p];
        int end = v.csc_ptr[var + 1];
        double acc = 0.0;
        // (rv.y is zero on the nucleus rows at this point — the update kernel cleared it, v_K is scattered later — so the
        // entries on nucleus rows add exact zeros and the row-map lookup that would skip them is not needed)
        for (int e = v.csc_ptr[var] + gl; e < end; e += G) acc += v.csc_val[e] * v.rv[v.csc_row[e]].y;
        acc = group_sum<G>(acc);
        if (gl == 0) v.tK[slot] = v.alpha_q[p] - acc;
    }
}


// v branch of the late primal iteration: tK = alpha_K - F^T y_S (solver.rs:1114) on its own, as soon as the FTRAN has landed.
// ------------------------------------------------------------------- stage heads inside the consuming kernel
// A stage head (one wave: a five-deep chain of dependent loads that turns the pivot column / row into a short list)
// used to be its own launch in front of the kernel that consumes the list: k_ftran_prep -> k_ftran_gather in the primal
// iteration, k_btran_prep -> k_btran in the dual one.  A launch boundary costs ~4.7 us in a replayed graph, the chain
// itself about as much; here EVERY block of the consuming kernel runs the head for itself into LDS (the inputs are a
// column or a row of A and the slot maps: a few hundred bytes, L2-resident), and block 0 alone performs the global side
// effects (pivot scalars, entries on singleton rows, the list for the records).  Used while the delayed-update mode is
// off and every column / row fits the LDS list (Geom.head_fused).
constexpr int HEAD_CAP = 1024;
__device__ void ftran_head_lds(const DevView& v, Ctl* c, int lane, int derive_primal, bool primary, int* ls, double* la, int* ln) {
    IterState* it = &c->it;
    const int q = it->q;
    const int var = v.nb_vars[q];
    if (primary && lane == 0) {
        it->entering_var = var;
        if (derive_primal) {  // solver.rs:741-748
            double dq = v.d[q];
            it->sign = dq < 0.0;
            it->entering_cur = v.xN[q];
            it->entering_other = (dq < 0.0) ? v.var_hi[var] : v.var_lo[var];
            it->r = -1;
            it->leaving_var = -1;
        }
    }
    // other blocks of this launch push -F alpha_K into the singleton positions with atomics while block 0 lands the
    // column's own singleton-row entries: both must then be atomic adds onto the zeroed vector
    const bool atomic_land = !v.pb_on && !v.det_pull;
    const int base = v.csc_ptr[var], end = v.csc_ptr[var + 1];
    int cnt = 0;
    for (int e0 = base; e0 < end; e0 += 64) {
        const int e = e0 + lane;
        const bool valid = e < end;
        int s = -1;
        double a = 0.0;
        if (valid) {
            const int i = v.csc_row[e];
            a = v.csc_val[e];
            s = v.kslot_of_row[i];
            if (s < 0 && primary) {
                const int p = v.pos_of_srow[i];
                const double x = a / v.sdiag_of_pos[p];
                if (atomic_land) unsafeAtomicAdd(&v.alpha_q[p], x);
                else v.alpha_q[p] = x;
                if (aq_listing(v)) aq_list_add(v, c, p);
            }
        }
        const bool isk = valid && s >= 0;
        const unsigned long long mask = __ballot(isk);
        if (isk) {
            const int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (off < HEAD_CAP) {
                ls[off] = s;
                la[off] = a;
            }
            if (primary) {
                v.klist_s[off] = s;
                v.klist_a[off] = a;
            }
        }
        cnt += __popcll(mask);
    }
    if (lane == 0) {
        *ln = cnt < HEAD_CAP ? cnt : HEAD_CAP;
        if (primary) it->klist_n = cnt;
    }
}
// The FTRAN head of the DELAYED-UPDATE mode inside the gather (round 5): k_ftran_prep was a launch of ONE wave in front of
// k_ftran_gather<4> — 9.0 us of launch floor and a five-deep chain of dependent loads per late pivot (profiles/r04e_late_kernel_stats.csv).
// Here every block of the gather runs the head for itself into LDS — the entering column is ~100 entries, the row map and the 32
// pending V rows are L2-resident — and block 0 alone performs the global side effects (pivot scalars, fold decision, entries on
// singleton rows, the list and the coefficients c_j = V[j] . list for the records and the tails).  Same list order, same sums: the
// bits of k_ftran_prep + k_ftran_gather<4>.  Primal iteration, blocked push, one GPU.
__global__ void __launch_bounds__(BLK) k_ftran_gather_lrh(DevView v, int fpk) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) c->side_go = 0;
        return;
    }
    KMARK0(c, 0);
    __shared__ int s_ls[HEAD_CAP];
    __shared__ double s_la[HEAD_CAP];
    __shared__ double s_lrc[LR_MAX];
    __shared__ int s_n;
    const bool primary = blockIdx.x == 0;
    const int nlow = c->nlow;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        IterState* it = &c->it;
        const int q = it->q;
        const int var = v.nb_vars[q];
        if (primary && lane == 0) {
            c->side_go = 1;
            it->entering_var = var;
            const double dq = v.d[q];  // solver.rs:741-748
            it->sign = dq < 0.0;
            it->entering_cur = v.xN[q];
            it->entering_other = (dq < 0.0) ? v.var_hi[var] : v.var_lo[var];
            it->r = -1;
            it->leaving_var = -1;
            c->fold = nlow >= v.lrJ ? 1 : 0;  // (decided ahead of everything that reads W0)
        }
        const int base = v.csc_ptr[var], end = v.csc_ptr[var + 1];
        int cnt = 0;
        const double* Vrow = v.V + (size_t)(lane < nlow ? lane : 0) * v.ld;
        double lr_acc = 0.0;
        for (int e0 = base; e0 < end; e0 += 128) {  // (as ftran_prep_wave: two chunks of 64 entries per trip, their loads issued together)
            int sx[2] = {-1, -1};
            double ax[2] = {0.0, 0.0};
            int ix[2];
            bool vx[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = e0 + 64 * h + lane;
                vx[h] = e < end;
                ix[h] = vx[h] ? v.csc_row[e] : 0;
                ax[h] = vx[h] ? v.csc_val[e] : 0.0;
            }
            RowInfo rx[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) rx[h] = v.rowinfo[ix[h]];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (vx[h]) {
                    sx[h] = rx[h].kslot;
                    if (sx[h] < 0 && primary) v.alpha_q[rx[h].pos] = ax[h] / rx[h].diag;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && e0 + 64 >= end) break;  // (uniform)
                const int s = sx[h];
                const double a = ax[h];
                const bool isk = vx[h] && s >= 0;
                const unsigned long long mask = __ballot(isk);
                if (isk) {
                    const int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                    if (off < HEAD_CAP) {
                        s_ls[off] = s;
                        s_la[off] = a;
                    }
                    if (primary) {
                        v.klist_s[off] = s;
                        v.klist_a[off] = a;
                    }
                }
                cnt += __popcll(mask);
                if (nlow > 0) lr_dot_listed(mask, s, a, Vrow, lane < nlow, lr_acc);
            }
        }
        if (lane == 0) {
            s_n = cnt < HEAD_CAP ? cnt : HEAD_CAP;
            if (primary) it->klist_n = cnt;
        }
        if (lane < LR_MAX) s_lrc[lane] = lane < nlow ? lr_acc : 0.0;
        if (primary && lane < nlow) c->lr_c[lane] = lr_acc;
    }
    __syncthreads();
    const int slot = (blockIdx.x * BLK + threadIdx.x) / 4;
    const int gl = threadIdx.x & 3;
    if (slot >= c->k) return;
    const int n = s_n;
    double acc = 0.0;
    const double* wrow = v.W + (size_t)slot * v.ld;
    // (fpull.inc: the slot's variable — a chain of two loads that overlaps the gather)
    const int p = v.pos_of_kslot[slot];
    const int fvar = fpk ? v.basic_vars[p] : 0;
    for (int j0 = gl; j0 < n; j0 += 16) {  // (k_ftran_gather<4>: the lane's loads of a trip issued together, added in list order)
        double a[4], w[4];
        int si[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            a[u] = j < n ? s_la[j] : 0.0;
            si[u] = j < n ? s_ls[j] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = j0 + 4 * u < n ? wrow[si[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + 4 * u < n) acc += a[u] * w[u];
    }
    {
        double e[LR_MAX / 4], x[LR_MAX / 4];
#pragma unroll
        for (int u = 0; u < LR_MAX / 4; ++u) {
            const int j = gl + 4 * u;
            e[u] = j < nlow ? s_lrc[j] : 0.0;
            x[u] = j < nlow ? v.U[(size_t)j * v.ld + slot] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < LR_MAX / 4; ++u)
            if (gl + 4 * u < nlow) acc += x[u] * e[u];
    }
    acc = group_sum<4>(acc);
    if (gl == 0) {
        v.aK[slot] = acc;
        v.alpha_q[p] = acc;
        if (fpk) v.fpk_x[fvar] = acc;  // alpha_K by variable: what the pull of the F product gathers
    }
}
template <int G>
__global__ void __launch_bounds__(BLK) k_ftran_fused(DevView v, int derive_primal) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 0);
    __shared__ int s_ls[HEAD_CAP];
    __shared__ double s_la[HEAD_CAP];
    __shared__ int s_n;
    if (threadIdx.x < 64) ftran_head_lds(v, c, threadIdx.x, derive_primal, blockIdx.x == 0, s_ls, s_la, &s_n);
    __syncthreads();
    const int slot = (blockIdx.x * BLK + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (slot >= c->k) return;
    const int n = s_n;
    double acc = 0.0;
    const double* wrow = v.W + (size_t)slot * v.ld;
    for (int j = gl; j < n; j += G) acc += s_la[j] * wrow[s_ls[j]];
    acc = group_sum<G>(acc);
    const int p = v.pos_of_kslot[slot];
    if (gl == 0) {
        v.aK[slot] = acc;
        v.alpha_q[p] = acc;
        if (acc != 0.0 && aq_listing(v)) aq_list_add(v, c, p);
    }
    if (acc != 0.0 && !v.pb_on && !v.det_pull) push_F<G>(v, p, acc, v.alpha_q, gl, aq_listing(v) ? c : nullptr);
    else if (acc != 0.0 && v.det_pull && v.fmark) mark_F<G>(v, p, gl);
}
__device__ void btran_head_lds(const DevView& v, Ctl* c, int lane, int derive_dual, bool primary, int* ls, double* la, int* ln) {
    IterState* it = &c->it;
    const int r = it->r;
    if (derive_dual && !c->forced && primary && lane == 0) {  // solver.rs:892-916 (a host-forced row keeps the host's value)
        const double val = v.xB[r], mn = v.loB[r];
        it->leaving_new_val = (val < mn) ? mn : v.hiB[r];
        it->leaving_var = v.basic_vars[r];
        it->q = -1;
        it->entering_var = -1;
    }
    const int sr = v.kslot_of_pos[r];
    if (sr >= 0) {
        if (lane == 0) {
            ls[0] = sr;
            la[0] = 1.0;
            *ln = 1;
            if (primary) {
                v.blist_s[0] = sr;
                v.blist_a[0] = 1.0;
                it->blist_n = 1;
            }
        }
        return;
    }
    const int i_r = v.srow_of_pos[r];
    const double inv = 1.0 / v.sdiag_of_pos[r];
    if (primary && lane == 0) {
        v.rv[i_r].x = inv;
        v.tau[r] = inv * inv;  // tau_S = (rho_S - F tauK)/diag: only row i_r of rho_S is non-zero
    }
    const int base = v.csr_ptr[i_r], end = v.csr_ptr[i_r + 1];
    int cnt = 0;
    for (int e0 = base; e0 < end; e0 += 64) {
        const int e = e0 + lane;
        const bool valid = e < end;
        int s = -1;
        double a = 0.0;
        if (valid) {
            const int loc = v.var_loc[v.csr_col[e]];
            a = v.csr_val[e];
            if (loc >= 0) s = v.kslot_of_pos[loc];
        }
        const bool isk = valid && s >= 0;
        const unsigned long long mask = __ballot(isk);
        if (isk) {
            const int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (off < HEAD_CAP) {
                ls[off] = s;
                la[off] = -a * inv;
            }
            if (primary) {
                v.blist_s[off] = s;
                v.blist_a[off] = -a * inv;
            }
        }
        cnt += __popcll(mask);
    }
    if (lane == 0) {
        *ln = cnt < HEAD_CAP ? cnt : HEAD_CAP;
        if (primary) it->blist_n = cnt;
    }
}
// k_btran with the BTRAN head inside (dual iteration): gather blocks build the list for themselves
template <int G>
__global__ void __launch_bounds__(BLK) k_btran_fused(DevView v, int n_gather, int derive_dual) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int k = c->k;
    if ((int)blockIdx.x < n_gather) {
        __shared__ int s_ls[HEAD_CAP];
        __shared__ double s_la[HEAD_CAP];
        __shared__ int s_n;
        if (threadIdx.x < 64) btran_head_lds(v, c, threadIdx.x, derive_dual, blockIdx.x == 0, s_ls, s_la, &s_n);
        __syncthreads();
        const int n = s_n;
        double sq = 0.0;
        for (int s = blockIdx.x * BLK + threadIdx.x; s < k; s += n_gather * BLK) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += s_la[j] * v.W[(size_t)s_ls[j] * v.ld + s];
            v.rK[s] = acc;
            v.rv[v.row_of_kslot[s]].x = acc;
            sq += acc * acc;
        }
        if (!grid_sum(sq, v, n_gather)) return;
        if (threadIdx.x == 0) {
            const int r = c->it.r;
            if (v.kslot_of_pos[r] < 0) {
                const double inv = 1.0 / v.sdiag_of_pos[r];
                sq += inv * inv;
            }
            c->it.rho_sq = sq;
        }
    } else {  // (PSE) tK = alpha_K - F^T y_S, as in k_btran
        const int slot = (((int)blockIdx.x - n_gather) * BLK + threadIdx.x) / G;
        const int gl = threadIdx.x & (G - 1);
        if (slot >= k) return;
        const int p = v.pos_of_kslot[slot];
        const int var = v.basic_vars[p];
        const int end = v.csc_ptr[var + 1];
        double acc = 0.0;
        for (int e = v.csc_ptr[var] + gl; e < end; e += G) {
            const int i = v.csc_row[e];
            if (v.kslot_of_row[i] < 0) acc += v.csc_val[e] * v.rv[i].y;
        }
        acc = group_sum<G>(acc);
        if (gl == 0) v.tK[slot] = v.alpha_q[p] - acc;
    }
}

// ------------------------------------------------------------------- partition change
// B'^-1[p,i] = B^-1[p,i] - (alpha_p - [p==r]) rho_i / alpha_r, restricted to the new nucleus.
// The case comes from the device-side plan.  Shrinking keeps the slots compact:
// W_new[a][b] = W_old[src_row(a)][src_col(b)] with src_row(sr) = last, src_col(cq) = last; all reads
// come from row/column `last`, all writes go to row sr / column cq, so there is no hazard.
// Delayed-update mode: instead of touching W0, append this pivot's rank-1 term to (U, V) and keep the
// pending terms consistent with the slot changes (new slot: zeros in the old terms; dropped slot: the
// last slot's entries move in; replaced column: zeros in every term's V).
__device__ __forceinline__ void lowrank_append(const DevView& v, Ctl* c, const StructUpdate& u, int s) {
    const int kold = u.kold, ld = v.ld, last = kold - 1;
    const double inv_alpha = c->it.inv_alpha;
    const int jn = u.jn;  // index of the new term (0 when this pivot's fused pass folded the pending ones)
    if (u.kase == 4) {
        if (s == 0) c->nlow = jn;
        return;
    }
    double* Un = v.U + (size_t)jn * ld;
    double* Vn = v.V + (size_t)jn * ld;
    if (u.kase == 0 || u.kase == 2) {
        const int lim = (u.kase == 2) ? last : kold;
        if (s < lim) {
            int sr_src = (u.kase == 2 && s == u.sr) ? last : s;   // row slot whose term value lands in slot s
            int sc_src = (u.kase == 2 && s == u.cq) ? last : s;   // col slot whose term value lands in slot s
            const double un = -(v.aK[sr_src] - (sr_src == u.sr ? 1.0 : 0.0)) * inv_alpha;
            Un[s] = un;
            v.Ut[(size_t)s * LR_MAX + jn] = un;
            Vn[s] = v.rK[sc_src];
        }
        if (u.kase == 2 && s < jn) {  // pending terms follow the move of the last slot
            double* Uj = v.U + (size_t)s * ld;
            double* Vj = v.V + (size_t)s * ld;
            if (u.sr != last) {
                const double x = Uj[last];
                Uj[u.sr] = x;
                v.Ut[(size_t)u.sr * LR_MAX + s] = x;
            }
            if (u.cq != last) Vj[u.cq] = Vj[last];
        }
    } else if (u.kase == 1) {
        if (s < kold) {
            const double un = -v.aK[s] * inv_alpha;
            Un[s] = un;
            v.Ut[(size_t)s * LR_MAX + jn] = un;
            Vn[s] = v.rK[s];
        } else if (s == kold) {
            Un[kold] = 0.0;
            v.Ut[(size_t)kold * LR_MAX + jn] = 0.0;
            Vn[kold] = 0.0;
        }
        if (s < jn) {  // the new slot does not exist in the pending terms
            v.U[(size_t)s * ld + kold] = 0.0;
            v.Ut[(size_t)kold * LR_MAX + s] = 0.0;
            v.V[(size_t)s * ld + kold] = 0.0;
        }
    } else if (u.kase == 3) {
        if (s < kold) {
            const double un = -v.aK[s] * inv_alpha;
            Un[s] = un;
            v.Ut[(size_t)s * LR_MAX + jn] = un;
            Vn[s] = (s == u.cq) ? 0.0 : v.rK[s];
        }
        if (s < jn) v.V[(size_t)s * ld + u.cq] = 0.0;
    }
    if (s == 0) c->nlow = jn + 1;
}
__device__ __forceinline__ void fpk_append_wave(const DevView& v, const Ctl* c, int lane);  // (fpull.inc)
__device__ __forceinline__ void struct_update_body(const DevView& v, Ctl* c, int s) {
    const StructUpdate u = c->up;
    if (u.kase < 0) return;
    if (v.fpk_on && s < 64) fpk_append_wave(v, c, s);  // the entering column joins the packed copy of the pulled F product (threads 0..63: one wave)
    if (v.lrJ) lowrank_append(v, c, u, s);
    if (u.kase == 0) return;
    const int kold = u.kold;
    const int ld = v.ld;
    const double inv_alpha = c->it.inv_alpha;
    if (u.kase == 1) {  // singleton -> nucleus: append row slot kold (position r) and col slot kold (row i_r)
        if (s < kold) {
            v.W[(size_t)kold * ld + s] = v.rK[s] * inv_alpha;
            v.W[(size_t)s * ld + kold] = -v.aK[s] * u.inv_diag_r * inv_alpha;
        } else if (s == kold) {
            v.W[(size_t)kold * ld + kold] = u.inv_diag_r * inv_alpha;
            v.kslot_of_pos[u.r] = kold;
            v.pos_of_kslot[kold] = u.r;
            v.kslot_of_row[u.i_r] = kold;
            v.rowinfo[u.i_r] = RowInfo{0.0, -1, kold};
            v.row_of_kslot[kold] = u.i_r;
            c->k = kold + 1;
        }
    } else if (u.kase == 2) {  // nucleus -> singleton: drop row slot sr and col slot cq
        const int last = kold - 1;
        if (s < last) {
            if (u.sr != last) v.W[(size_t)u.sr * ld + s] = v.W[(size_t)last * ld + (s == u.cq ? last : s)];
            if (u.cq != last) v.W[(size_t)s * ld + u.cq] = v.W[(size_t)(s == u.sr ? last : s) * ld + last];
        }
        if (s == 0) {
            if (u.sr != last) {
                int pl = v.pos_of_kslot[last];
                v.pos_of_kslot[u.sr] = pl;
                v.kslot_of_pos[pl] = u.sr;
            }
            v.kslot_of_pos[u.r] = -1;
            v.srow_of_pos[u.r] = u.i_q;
            v.sdiag_of_pos[u.r] = u.diag_q;
            if (u.cq != last) {
                int il = v.row_of_kslot[last];
                v.row_of_kslot[u.cq] = il;
                v.kslot_of_row[il] = u.cq;
                v.rowinfo[il] = RowInfo{0.0, -1, u.cq};
            }
            v.kslot_of_row[u.i_q] = -1;
            v.pos_of_srow[u.i_q] = u.r;
            v.rowinfo[u.i_q] = RowInfo{u.diag_q, u.r, -1};
            c->k = last;
        }
    } else if (u.kase == 3) {  // singleton -> singleton on another row: col slot cq now stands for row i_r
        if (s < kold) v.W[(size_t)s * ld + u.cq] = -v.aK[s] * u.inv_diag_r * inv_alpha;
        if (s == 0) {
            v.srow_of_pos[u.r] = u.i_q;
            v.sdiag_of_pos[u.r] = u.diag_q;
            v.kslot_of_row[u.i_q] = -1;
            v.pos_of_srow[u.i_q] = u.r;
            v.rowinfo[u.i_q] = RowInfo{u.diag_q, u.r, -1};
            v.kslot_of_row[u.i_r] = u.cq;
            v.rowinfo[u.i_r] = RowInfo{0.0, -1, u.cq};
            v.row_of_kslot[u.cq] = u.i_r;
        }
    } else if (u.kase == 4) {  // singleton replaced by another singleton of the same row
        if (s == 0) {
            v.sdiag_of_pos[u.r] = u.diag_q;
            v.rowinfo[v.srow_of_pos[u.r]] = RowInfo{u.diag_q, u.r, -1};
        }
    }
}
__global__ void __launch_bounds__(BLK) k_struct_update(DevView v) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    struct_update_body(v, c, blockIdx.x * BLK + threadIdx.x);
}

// ------------------------------------------------------------------- K4: tableau row  rho^T N
// solver.rs:685-692 (and 1117-1132 for the PSE helper).  The reference pushes rows of supp(rho)
// through the CSR; here every non-basic column PULLS its dot product from the CSC: no atomics,
// fixed summation order, one streaming pass over A that yields alpha_r and (PSE) N^T v together.
// G lanes per column, U independent (index -> gather) chains per lane per trip; rho and v are
// interleaved (double2) so one 16-byte gather serves both products.  nb_rng[c] caches the CSC
// range of the column at non-basic position c.  Blocks beyond n_sweep apply the partition change
// (it touches W and the slot maps only, nothing the sweep reads).
// Measured (PMC, config 4): 11.9 M L2 requests per launch, 90 % L2 hits, FETCH_SIZE 76 MB — A lives
// in the 256 MB Infinity Cache and the kernel is bound by the L2 request rate of the 10^7 gathers.
// Non-temporal loads on the A stream were tried: 82 us vs 57 us (worse).
template <int G, int U, int MODE>  // MODE 0: alpha_r, 1: alpha_r + helper, 2: helper
__global__ void __launch_bounds__(BLK) k_sweep(DevView v, int n_sweep, int build_list) {
    Ctl* c = v.ctl;
    const int halt0 = c->halt, status0 = c->it.status, ar_off0 = c->ar_off;  // (one clause of scalar loads)
    if (halt0 || status0 != ITER_PIVOT) return;
    if ((int)blockIdx.x >= n_sweep) {
        struct_update_body(v, c, ((int)blockIdx.x - n_sweep) * BLK + threadIdx.x);
        return;
    }
    int col = v.nb_lo + (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (col >= v.nb_hi) return;
    const int2 rg = v.nb_rng[col];
    const int beg = rg.x, end = rg.y;
    double a1 = 0.0, a2 = 0.0;
    for (int e0 = beg + gl; e0 < end; e0 += U * G) {
        int idx[U];
        double a[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            int e = e0 + j * G;
            bool ok = e < end;
            idx[j] = ok ? v.csc_row[e] : 0;
            a[j] = ok ? v.csc_val[e] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (MODE == 1) {
                double2 t = v.rv[idx[j]];
                a1 += a[j] * t.x;
                a2 += a[j] * t.y;
            } else if (MODE == 0) {
                a1 += a[j] * v.rv[idx[j]].x;
            } else {
                a2 += a[j] * v.rv[idx[j]].y;
            }
        }
    }
    if (MODE != 2) a1 = group_sum<G>(a1);
    if (MODE != 0) a2 = group_sum<G>(a2);
    if (gl == 0) {
        if (MODE != 2) v.alpha_r[col] = a1;
        if (MODE != 0) v.helper[col] = a2;
    }
    if (MODE == 0 && build_list && v.ar_list && ar_off0 == 0) {  // (dual iteration: the Harris test that follows consumes and resets the list) the non-zeros of the tableau row as a list (wave-aggregated counter; the order of the list is free)
        const bool nz = gl == 0 && a1 != 0.0;
        const unsigned long long mask = __ballot(nz);
        if (mask) {
            const int lane = threadIdx.x & 63, lead = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == lead) base = atomicAdd(&c->ar_n, __popcll(mask));
            base = __shfl(base, lead, 64);
            if (nz) v.ar_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = col;  // (at most n entries: one per column)
        }
    }
}
// Banded sweep: the pull above is bound by the L2 request rate of its 10^7 row-indexed
// gathers.  Here workgroup (band b, chunk c) first copies its band of the interleaved (rho, v) vector into
// LDS (128 KB), then every thread walks one non-basic column's entries INSIDE the band (band-major copy
// of A, ~nnz/nbands entries per column; caching the per-position segments like nb_rng was tried and only
// moved the cost into k_update_pivot) and
// gathers from LDS; the per-band partial dot products go to band_part[b][j] and k_band_combine adds them
// up in band order (fixed summation order, no atomics).
typedef unsigned int uint4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));
// VORD: the columns are visited in VARIABLE order (the storage order of the band-major copy) instead of non-basic
// POSITION order.  After many pivots nb_vars is a random permutation, so position order turns the read of the copy into
// 80-byte gathers out of 128-byte lines: the PMC counters show 521 MB of HBM traffic per launch at pivot 240 000 against
// 150 MB at pivot 200, for the same 116 MB copy.  In variable order the copy is read front to back; basic variables are
// skipped (var_loc >= 0) and the partials are scattered to band_part[b][position] (16-byte stores).  Unsharded solves
// only: a rank of a sharded solve owns a range of POSITIONS, which in variable order would be a scattered subset.
template <int MODE, bool VORD>
__global__ void __launch_bounds__(BAND_THREADS) k_sweep_band(DevView v, int chunks) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 22);
    extern __shared__ double2 s_rv[];
    const int n_band_blocks = v.nbands * chunks;
    if ((int)blockIdx.x >= n_band_blocks) {  // horizontally fused: the partition change (primal iteration)
        struct_update_body(v, c, ((int)blockIdx.x - n_band_blocks) * BAND_THREADS + threadIdx.x);
        return;
    }
    const int b = (int)blockIdx.x / chunks, chunk = (int)blockIdx.x % chunks, tid = threadIdx.x;
    const int row0 = b * BAND_ROWS;
    const int nrows = min(BAND_ROWS, v.m - row0);
    for (int t = tid; t < nrows; t += BAND_THREADS) s_rv[t] = v.rv[row0 + t];
    // The entry loop reads whole groups of 8 entries and MASKS the ones beyond a segment's end by a zero value: their row
    // indices (a neighbour's, or spare entries) still index this LDS array, and 0 * (stale LDS holding a NaN pattern) is NaN.
    // The last band is short: give the rows beyond it a defined value.  (Found in round 4: the integer accumulators the
    // deterministic blocked push leaves behind in LDS read as NaNs; three entries of a tableau row came out non-finite.)
    for (int t = nrows + tid; t < BAND_ROWS; t += BAND_THREADS) s_rv[t] = make_double2(0.0, 0.0);
    __syncthreads();
    const int base = VORD ? 0 : v.nb_lo;
    const int span = VORD ? v.m + v.n : v.nb_hi - v.nb_lo;  // variables, or this rank's non-basic positions
    const int per = (span + chunks - 1) / chunks;
    const int c_lo = base + chunk * per;
    const int c_hi = min(base + span, c_lo + per);
    const int* bp = v.bptr + (size_t)b * (size_t)(v.m + v.n + 1);
    double2* out = v.band_part + (size_t)b * (size_t)v.n;
    constexpr int CPT = 4;  // columns per thread resolved together: their index chains overlap
    for (int j0 = c_lo + tid; j0 < c_hi; j0 += CPT * BAND_THREADS) {
    int vars[CPT], poss[CPT], begs[CPT], ends[CPT];
    bool pkd[CPT];
    const int* pkp = (!VORD && v.pk_ptr) ? v.pk_ptr + (size_t)b * (size_t)(v.n + 1) : nullptr;
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int i = j0 + u * BAND_THREADS;
        vars[u] = -1;
        poss[u] = -1;
        pkd[u] = false;
        if (i < c_hi) {
            if (!VORD && pkp && v.pk_valid[i]) {  // packed copy: the segment of index i, straight from its offsets
                pkd[u] = true;
                vars[u] = 0;
                poss[u] = i;
                continue;
            }
            if (VORD) {
                const int loc = v.var_loc[i];
                if (loc < 0) {  // non-basic: position -1 - loc
                    vars[u] = i;
                    poss[u] = -1 - loc;
                }
            } else {
                // locality order: index i of the pass serves position nb_order[i] (positions sorted by the variable
                // they hold, refreshed by the host now and then), and the partials are indexed by i
                vars[u] = v.nb_vars[v.nb_order ? v.nb_order[i] : i];
                poss[u] = i;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        if (pkd[u]) {
            begs[u] = pkp[poss[u]];
            ends[u] = pkp[poss[u] + 1];
        } else {
            begs[u] = vars[u] >= 0 ? bp[vars[u]] : 0;
            ends[u] = vars[u] >= 0 ? bp[vars[u] + 1] : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        if (vars[u] < 0) continue;
        const int j = poss[u];
        const int beg = begs[u], end = ends[u];
        const unsigned short* __restrict__ Rw = pkd[u] ? v.pk_row : v.brow;
        const double* __restrict__ Vl = pkd[u] ? v.pk_val : v.bval;
        double a1 = 0.0, a2 = 0.0;
        for (int e0 = beg; e0 < end; e0 += 8) {
            // Eight entries per step as 1 + 4 sixteen-byte loads per lane (a lane's entries are contiguous;
            // eight scalar loads each touch 64 different lines per wave and the kernel becomes bound by the
            // texture-address unit: 35.5 us).  Reading past `end` is harmless: the arrays carry 8 spare
            // entries, a neighbour's rows are valid indices of this band, and the values are masked.
            const uint4u rr = *reinterpret_cast<const uint4u*>(Rw + e0);  // eight 16-bit rows; e0 is even
            const dbl2u x0 = *reinterpret_cast<const dbl2u*>(Vl + e0);
            const dbl2u x1 = *reinterpret_cast<const dbl2u*>(Vl + e0 + 2);
            const dbl2u x2 = *reinterpret_cast<const dbl2u*>(Vl + e0 + 4);
            const dbl2u x3 = *reinterpret_cast<const dbl2u*>(Vl + e0 + 6);
            const unsigned r[8] = {rr.x & 0xffffu, rr.x >> 16, rr.y & 0xffffu, rr.y >> 16,
                                   rr.z & 0xffffu, rr.z >> 16, rr.w & 0xffffu, rr.w >> 16};
            double a[8] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y, x3.x, x3.y};
            const int nv = end - e0;
#pragma unroll
            for (int u = 1; u < 8; ++u)
                if (u >= nv) a[u] = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double2 t = s_rv[r[u]];
                if (MODE != 2) a1 += a[u] * t.x;
                if (MODE != 0) a2 += a[u] * t.y;
            }
        }
        out[j] = make_double2(a1, a2);
    }
    }
}
template <int MODE>
__global__ void __launch_bounds__(BLK) k_band_combine(DevView v, int n_comb) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    if ((int)blockIdx.x >= n_comb) {  // horizontally fused: the partition change (as in k_sweep)
        struct_update_body(v, c, ((int)blockIdx.x - n_comb) * BLK + threadIdx.x);
        return;
    }
    const int i = v.nb_lo + blockIdx.x * BLK + threadIdx.x;  // index of the pass (= position without a locality order)
    if (i >= v.nb_hi) return;
    const int j = v.nb_order ? v.nb_order[i] : i;
    double a1 = 0.0, a2 = 0.0;
    for (int b = 0; b < v.nbands; ++b) {
        const double2 t = v.band_part[(size_t)b * (size_t)v.n + i];
        a1 += t.x;
        a2 += t.y;
    }
    if (MODE != 2) v.alpha_r[j] = a1;
    if (MODE != 0) v.helper[j] = a2;
}
__global__ void __launch_bounds__(BLK) k_init_nb_rng(DevView v) {
    int j = blockIdx.x * BLK + threadIdx.x;
    if (j < v.n) {
        int var = v.nb_vars[j];
        v.nb_rng[j] = make_int2(v.csc_ptr[var], v.csc_ptr[var + 1]);
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
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    int lsign = c->it.leaving_new_val > v.xB[c->it.r];
    double mn = INFINITY, dummy = 0.0;
    for (int j = v.nb_lo + blockIdx.x * BLK + threadIdx.x; j < v.nb_hi; j += gridDim.x * BLK) {
        double coeff = v.alpha_r[j];
        uint8_t f = v.nbflags[j];
        if (!dual_eligible(coeff, f, lsign)) continue;
        double cur = (fabs(clamp_obj(v.d[j], f)) + EPS) / fabs(coeff);
        if (cur < mn) mn = cur;
    }
    if (!grid_min_sum(mn, dummy, v)) return;
    __shared__ double s_mn;
    if (threadIdx.x == 0) s_mn = mn;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    double g = s_mn;
    bool ok = exchange_min_wave(v, c, g, threadIdx.x);  // sharded: minimum over all column blocks
    if (threadIdx.x == 0 && ok) c->it.max_step = g;
}
struct ArState { int n, off, back; };  // Ctl.ar_n / ar_off / ar_back as the kernel found them (loaded once, in its first clause of scalar loads)
__device__ void ratio_dual_finish(const DevView& v, Ctl* c, Cand best, ArState ar);
// solver.rs:979-1021; the finalising block goes straight on with the FTRAN head
__global__ void __launch_bounds__(BLK) k_ratio_dual_p2(DevView v) {
    Ctl* c = v.ctl;
    const ArState ar{c->ar_n, c->ar_off, c->ar_back};
    if (c->halt || c->it.status != ITER_PIVOT) return;
    IterState* it = &c->it;
    int r = it->r;
    int lsign = it->leaving_new_val > v.xB[r];
    double max_step = it->max_step;
    Cand best = cand_none();
    for (int j = v.nb_lo + blockIdx.x * BLK + threadIdx.x; j < v.nb_hi; j += gridDim.x * BLK) {
        double coeff = v.alpha_r[j];
        uint8_t f = v.nbflags[j];
        if (!dual_eligible(coeff, f, lsign)) continue;
        double cur = fabs(clamp_obj(v.d[j], f)) / fabs(coeff);
        if (cur <= max_step) {
            Cand t{fabs(coeff), j};
            if (cand_better(t, best)) best = t;
        }
    }
    if (!grid_best(best, v)) return;
    ratio_dual_finish(v, c, best, ar);
}
// Finalising block of the dual ratio test: (sharded: candidate all-gather,) the decision (solver.rs:1000-1021) and the
// FTRAN head of the entering column.
__device__ void ratio_dual_finish(const DevView& v, Ctl* c, Cand best, ArState ar) {
    IterState* it = &c->it;
    const int r = it->r;
    const int ar_off0 = ar.off, ar_back0 = ar.back, ar_n0 = ar.n;
    __shared__ int s_ok;
    __shared__ double s_key;
    __shared__ int s_idx;
    if (v.world > 1) {  // sharded: all-gather of the per-block candidates with (alpha_rq, d_q) as payload
        if (threadIdx.x == 0) {
            s_key = best.key;
            s_idx = best.idx;
        }
        __syncthreads();
        bool ok = true;
        if (threadIdx.x < 64) {
            Cand b{s_key, s_idx};
            const bool have = b.idx != NONE_IDX;
            ok = exchange_best_wave(v, c, 1, b, have ? v.d[b.idx] : 0.0, threadIdx.x, 3, have ? v.alpha_r[b.idx] : 0.0);
            if (threadIdx.x == 0) {
                s_key = b.key;
                s_idx = ok ? b.idx : NONE_IDX;
                s_ok = ok ? 1 : 0;
            }
        }
        __syncthreads();
        if (!s_ok) return;  // communication failure: already recorded
        best = Cand{s_key, s_idx};
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        s_ok = 0;
        // (the list of alpha_r's non-zeros has been consumed: the next tableau row starts an empty one — or none for a while when this one
        // passed the cap: a dense row's appends, one arrival per wave at one address, cost up to 110 us on the 160 000-column config-3 family)
        if (ar_off0 > 0) {
            c->ar_off = ar_off0 - 1;
        } else if (ar_n0 > 8 * AR_CAP) {  // (a DENSE row: a list that just misses the cap costs little and says little about the next one)
            const int nb2 = ar_back0 < 16 ? 16 : (ar_back0 < 256 ? 2 * ar_back0 : 256);
            c->ar_back = nb2;
            c->ar_off = nb2;
            c->ar_pauses += 1;
        } else if (ar_back0 != 0) {
            c->ar_back = 0;
        }
        c->ar_n = 0;
        c->ar_keep = -1;  // (ratio_dual_list sets it behind this call)
        if (best.idx == NONE_IDX) {
            it->status = ITER_INFEASIBLE;
            c->halt = 1;
            push_rec(c, 1);
        } else {
            int q = best.idx;
            double coeff = v.alpha_r[q];
            double dq = v.d[q];
            double diff = (v.xB[r] - it->leaving_new_val) / coeff;  // solver.rs:1005
            it->q = q;
            it->pivot_coeff = coeff;
            it->entering_diff = diff;
            it->entering_cur = v.xN[q];
            it->entering_new_val = v.xN[q] + diff;
            it->pivot_obj = dq / coeff;
            it->obj += dq * diff;
            it->leaving_var = v.basic_vars[r];
            s_ok = 1;
        }
    }
    __syncthreads();
    if (s_ok && threadIdx.x < 64) ftran_prep_wave(v, c, threadIdx.x, 0);
}
// Small models (<= RATIO_ONE_MAX positions, one GPU): both dual Harris passes in ONE BLOCK — two loops over alpha_r (the second
// re-reads it from L2), no ticketed grid reductions, no in-kernel wait.  Measured on config 3 (n = 10 000): the one-launch
// grid form costs 12.9 us, of which the scan itself is a fraction.
constexpr int RATIO_ONE_MAX = 16384;
// Dual Harris test over the LISTED non-zeros of the tableau row (round 6; solver.rs:962-1002 walks the non-zeros of row_coeffs only): on sparse
// models alpha_r has a handful to a few hundred entries of n = 10^4 ... 4 10^5, and the two grid-wide passes of k_ratio_dual_fused are then
// two ticketed reductions plus an in-kernel wait over nothing (19-22 us on the 400 000-column transport instance), the one-block form a
// 40-trip walk over zeros (16 us on config 3).  Block 0 alone takes a list of up to AR_CAP entries; true = handled (every block returns).
__device__ __forceinline__ bool ratio_dual_list(const DevView& v, Ctl* c, int list_ok, ArState ar) {
    if (!list_ok || !v.ar_list || v.world > 1) return false;
    const int nl = ar.n, off = ar.off;
    if (off > 0 || nl > AR_CAP) return false;
    if (blockIdx.x != 0) return true;
    IterState* it = &c->it;
    const int lsign = it->leaving_new_val > v.xB[it->r];
    constexpr int PL = AR_CAP / BLK;
    int pos[PL];
    double ca[PL], dj[PL];
    double mn = INFINITY;
    int jj[PL];
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const int a = (int)threadIdx.x + u * BLK;
        jj[u] = a < nl ? v.ar_list[a] : -1;
    }
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const int j = jj[u] < 0 ? v.nb_lo : jj[u];
        const double coeff = v.alpha_r[j];
        const uint8_t f = v.nbflags[j];
        const double d = v.d[j];
        pos[u] = -1;
        ca[u] = 0.0;
        dj[u] = 0.0;
        if (jj[u] < 0 || !dual_eligible(coeff, f, lsign)) continue;
        pos[u] = j;
        ca[u] = fabs(coeff);
        dj[u] = fabs(clamp_obj(d, f));
        const double cur = (dj[u] + EPS) / ca[u];
        if (cur < mn) mn = cur;
    }
    __shared__ double s_lstep;
    mn = block_min(mn);
    if (threadIdx.x == 0) {
        it->max_step = mn;
        s_lstep = mn;
        c->ar_used += 1;
    }
    __syncthreads();
    const double max_step = s_lstep;
    Cand best = cand_none();
#pragma unroll
    for (int u = 0; u < PL; ++u) {  // (ties keep the lowest position whatever the order of the list: cand_better)
        if (pos[u] < 0) continue;
        if (dj[u] / ca[u] <= max_step) {
            Cand t{ca[u], pos[u]};
            if (cand_better(t, best)) best = t;
        }
    }
    best = block_best(best);
    ratio_dual_finish(v, c, best, ar);
    if (threadIdx.x == 0) c->ar_keep = nl;  // (the update kernel walks the same list)
    return true;
}
__global__ void __launch_bounds__(BLK) k_ratio_dual_one(DevView v, int list_ok) {
    Ctl* c = v.ctl;
    const ArState ar{c->ar_n, c->ar_off, c->ar_back};
    if (c->halt || c->it.status != ITER_PIVOT) return;
    if (ratio_dual_list(v, c, list_ok, ar)) return;
    IterState* it = &c->it;
    const int lsign = it->leaving_new_val > v.xB[it->r];
    double mn = INFINITY;
    for (int j0 = v.nb_lo + threadIdx.x; j0 < v.nb_hi; j0 += 4 * BLK) {
        double co[4], dd[4];
        uint8_t ff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * BLK;
            const int jc = j < v.nb_hi ? j : v.nb_lo;
            co[u] = j < v.nb_hi ? v.alpha_r[jc] : 0.0;
            ff[u] = v.nbflags[jc];
            dd[u] = v.d[jc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!dual_eligible(co[u], ff[u], lsign)) continue;
            const double cur = (fabs(clamp_obj(dd[u], ff[u])) + EPS) / fabs(co[u]);
            if (cur < mn) mn = cur;
        }
    }
    __shared__ double s_step;
    mn = block_min(mn);
    if (threadIdx.x == 0) {
        it->max_step = mn;
        s_step = mn;
    }
    __syncthreads();
    const double max_step = s_step;
    Cand best = cand_none();
    for (int j0 = v.nb_lo + threadIdx.x; j0 < v.nb_hi; j0 += 4 * BLK) {
        double co[4], dd[4];
        uint8_t ff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * BLK;
            const int jc = j < v.nb_hi ? j : v.nb_lo;
            co[u] = j < v.nb_hi ? v.alpha_r[jc] : 0.0;
            ff[u] = v.nbflags[jc];
            dd[u] = v.d[jc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // ascending positions per thread: ties keep the lowest position
            if (!dual_eligible(co[u], ff[u], lsign)) continue;
            if (fabs(clamp_obj(dd[u], ff[u])) / fabs(co[u]) <= max_step) {
                Cand t{fabs(co[u]), j0 + u * BLK};
                if (cand_better(t, best)) best = t;
            }
        }
    }
    best = block_best(best);
    ratio_dual_finish(v, c, best, ar);
}
// Both dual Harris passes in ONE launch, like k_ratio_primal_fused: pass 1's last-arriving block (after the all-reduce
// over the ranks of a sharded solve) publishes the step bound, every block waits for it and runs pass 2 on the
// elements it still holds in registers.  Launched only when the grid is co-resident (launch_ratio_dual).
// PT positions per thread: 4, or 16 on models with more than 131 072 non-basic positions — every block takes two tickets at ONE L2 address
// (~15 ns per arrival when they arrive together): 391 blocks on the 400 000-column transport instance queued for ~12 us of a 22 us kernel
template <int PT>
__global__ void __launch_bounds__(BLK) k_ratio_dual_fused(DevView v, int list_ok) {
    Ctl* c = v.ctl;
    const ArState ar{c->ar_n, c->ar_off, c->ar_back};
    if (c->halt || c->it.status != ITER_PIVOT) return;
    if (ratio_dual_list(v, c, list_ok, ar)) return;
    IterState* it = &c->it;
    const int lsign = it->leaving_new_val > v.xB[it->r];
    const int epoch0 = c->ratio_epoch;
    double ca[PT], dj[PT];
    int pos[PT];
    double mn = INFINITY, dummy = 0.0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int j = v.nb_lo + (int)blockIdx.x * BLK + threadIdx.x + u * (int)gridDim.x * BLK;
        pos[u] = -1;
        ca[u] = 0.0;
        dj[u] = 0.0;
        if (j >= v.nb_hi) continue;
        const double coeff = v.alpha_r[j];
        const uint8_t f = v.nbflags[j];
        if (!dual_eligible(coeff, f, lsign)) continue;
        pos[u] = j;
        ca[u] = fabs(coeff);
        dj[u] = fabs(clamp_obj(v.d[j], f));
        const double cur = (dj[u] + EPS) / ca[u];
        if (cur < mn) mn = cur;
    }
    __shared__ double s_step;
    __shared__ int s_state;
    if (grid_min_sum(mn, dummy, v)) {  // last arriver of pass 1
        if (threadIdx.x == 0) s_step = mn;
        __syncthreads();
        bool ok = true;
        double g = s_step;
        if (threadIdx.x < 64) ok = exchange_min_wave(v, c, g, threadIdx.x);  // sharded: minimum over all column blocks
        if (threadIdx.x == 0) {
            if (ok) it->max_step = g;
            st_agent(&c->ratio_max_step, ok ? g : -1.0);  // a negative bound tells the waiters to give up (failure recorded)
            __hip_atomic_store(v.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_agent(&c->ratio_epoch, epoch0 + 1);
        }
    }
    if (threadIdx.x == 0) {
        int state = 0;
        long spins = 0;
        while (ld_agent(&c->ratio_epoch) != epoch0 + 1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > c->ratio_spin_limit) {
                state = 1;
                break;
            }
        }
        const double stp = ld_agent(&c->ratio_max_step);
        if (state == 0 && stp < 0.0) state = 2;
        s_step = stp;
        s_state = state;
    }
    __syncthreads();
    if (s_state) {
        if (s_state == 1 && threadIdx.x == 0 && atomicExch(&c->halt, 1) == 0) {  // (any block: see k_ratio_primal_fused)
            it->status = ITER_STALL;
            push_rec(c, 1);
        }
        return;
    }
    const double max_step = s_step;
    Cand best = cand_none();
#pragma unroll
    for (int u = 0; u < PT; ++u) {  // ascending positions per thread: ties keep the lowest position
        if (pos[u] < 0) continue;
        if (dj[u] / ca[u] <= max_step) {
            Cand t{ca[u], pos[u]};
            if (cand_better(t, best)) best = t;
        }
    }
    if (!grid_best(best, v)) return;
    ratio_dual_finish(v, c, best, ar);
}
// tK = alpha_K - F^T y_S on its own (dual path with PSE, where it cannot ride on k_btran)
template <int G>
__global__ void __launch_bounds__(BLK) k_btran_rhs(DevView v) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (slot >= c->k) return;
    int p = v.pos_of_kslot[slot];
    int var = v.basic_vars[p];
    int end = v.csc_ptr[var + 1];
    double acc = 0.0;
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) acc += v.csc_val[e] * v.rv[v.csc_row[e]].y;  // (zero on nucleus rows, see k_btran)
    acc = group_sum<G>(acc);
    if (gl == 0) v.tK[slot] = v.alpha_q[p] - acc;
}

// ------------------------------------------------------------------- fused pass over W
// One read + one write of the dense nucleus inverse per pivot does three things at once:
//   tauK = W * rK          (FTRAN #2 for dual steepest edge, solver.rs:1157)
//   vK   = W^T * tK        (BTRAN #2 for primal steepest edge, solver.rs:1114)
//   W   -= (aK - e_r) rK^T / alpha_r   (the eta transformation of solver.rs:1274-1284 applied
//                                       eagerly: B'^-1 = E B^-1)
// Block = FW_TR rows x FW_TC columns; per-block partials are reduced by k_post_fused in a fixed
// order (no float atomics => bitwise reproducible).
// W accesses of the fused pass.  NT = true: non-temporal loads and stores.  Measured on config 4 at
// k = 6 500 (W = 340 MB > the 256 MB Infinity Cache): the pass itself is a little slower (138 vs 125 us)
// but it stops evicting A, so the tableau-row sweep stays at 59 us instead of 90 us: +9 % pivots/s.
// While W fits the cache (cap <= 4096) plain accesses are faster.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 fw_load2(const double* p) {
    if (NT) {
        dbl2_t t = __builtin_nontemporal_load(reinterpret_cast<const dbl2_t*>(p));
        return make_double2(t.x, t.y);
    }
    return *reinterpret_cast<const double2*>(p);
}
template <bool NT>
__device__ __forceinline__ void fw_store2(double* p, double a, double b) {
    if (NT) {
        dbl2_t t = {a, b};
        __builtin_nontemporal_store(t, reinterpret_cast<dbl2_t*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = make_double2(a, b);
    }
}
// LR = delayed-update mode, normal (non-folding) pivot: the same streaming pass with DO_UPDATE off;
// it leaves folding pivots to k_fused_lr, and one extra block row computes the low-rank dots
// g_j = V[j].rho_K, h_j = U[j].t_K that k_post_fused adds to the two products.
// TILED: 1-D grid; the tile blocks stride over the (stripe, chunk) tiles of the CURRENT k, so neither a
// pass nor an early exit pays for dispatching the cap-sized grid (16 384 blocks at cap 16 384: ~20 us).
template <int TR, bool WITH_TAU, bool WITH_V, bool DO_UPDATE, bool NT = false, bool LR = false, int RL = 1, bool TILED = false>
__global__ void __launch_bounds__(BLK) k_fused_w(DevView v) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 8);
    if (LR && c->fold) return;
    const int k = c->k, ld = v.ld;
    const int n_tile_blocks = TILED ? (int)gridDim.x - (LR ? LR_MAX : 0) : 0;
    if (LR && (TILED ? (int)blockIdx.x >= n_tile_blocks : blockIdx.y == gridDim.y - 1)) {
        const int j = TILED ? (int)blockIdx.x - n_tile_blocks : (int)blockIdx.x;
        if (j >= c->nlow) return;
        const double* Vj = v.V + (size_t)j * ld;
        const double* Uj = v.U + (size_t)j * ld;
        double g = 0.0, h = 0.0;
        for (int s = threadIdx.x; s < k; s += BLK) {
            g += Vj[s] * v.rK[s];
            if (WITH_V) h += Uj[s] * v.tK[s];
        }
        g = block_sum(g);
        h = block_sum(h);
        if (threadIdx.x == 0) {
            c->lr_g[j] = g;
            c->lr_h[j] = h;
        }
        return;
    }
    // a block owns RL consecutive row tiles of TR rows (RL > 1 for a large nucleus: RL times fewer
    // v partials to write here and to reduce in k_post_fused) and one chunk of FW_TC columns
    __shared__ double s_tau[TR * RL][BLK / 64];
    const int tid = threadIdx.x;
    const int nchunks_k = (k + FW_TC - 1) / FW_TC;
    const int ntiles = TILED ? ((k + TR * RL - 1) / (TR * RL)) * nchunks_k : 1;
    for (int tile = TILED ? (int)blockIdx.x : 0; tile < ntiles; tile += TILED ? n_tile_blocks : 1) {
    const int stripe = TILED ? tile / nchunks_k : (int)blockIdx.x;
    const int chunk = TILED ? tile % nchunks_k : (int)blockIdx.y;
    const int row0 = stripe * (TR * RL);
    const int col0 = chunk * FW_TC;
    if (row0 >= k || col0 >= k) return;  // untiled grids are sized by the capacity
    int cidx[4];
    cidx[0] = col0 + 2 * tid;
    cidx[1] = cidx[0] + 1;
    cidx[2] = col0 + 512 + 2 * tid;
    cidx[3] = cidx[2] + 1;
    double rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rk[j] = (cidx[j] < k) ? v.rK[cidx[j]] : 0.0;
    const double inv_alpha = DO_UPDATE ? c->it.inv_alpha : 0.0;
    const int rslot = DO_UPDATE ? c->up.sr : -1;
    double vacc[4] = {0.0, 0.0, 0.0, 0.0};
    const bool pair0 = cidx[1] < k, pair1 = cidx[3] < k;
    const bool one0 = cidx[0] < k, one1 = cidx[2] < k;
    const int wv = tid >> 6, l = tid & 63;
#pragma unroll 1
    for (int sub = 0; sub < RL; ++sub) {
        double tacc[TR];
#pragma unroll
        for (int a = 0; a < TR; ++a) {
            int row = row0 + sub * TR + a;
            tacc[a] = 0.0;
            if (row >= k) continue;
            double* wp = v.W + (size_t)row * ld;
            double w[4] = {0.0, 0.0, 0.0, 0.0};
            if (pair0) {
                double2 t = fw_load2<NT>(wp + cidx[0]);
                w[0] = t.x;
                w[1] = t.y;
            } else if (one0) {
                w[0] = wp[cidx[0]];
            }
            if (pair1) {
                double2 t = fw_load2<NT>(wp + cidx[2]);
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
                if (pair0) fw_store2<NT>(wp + cidx[0], w[0], w[1]);
                else if (one0) wp[cidx[0]] = w[0];
                if (pair1) fw_store2<NT>(wp + cidx[2], w[2], w[3]);
                else if (one1) wp[cidx[2]] = w[2];
            }
        }
        if (WITH_TAU) {
#pragma unroll
            for (int a = 0; a < TR; ++a) {
                double sacc = wave_sum(tacc[a]);
                if (l == 0) s_tau[sub * TR + a][wv] = sacc;
            }
        }
    }
    if (WITH_V) {
        double* pv = v.part_v + (size_t)stripe * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (cidx[j] < k) pv[cidx[j]] = vacc[j];
    }
    if (WITH_TAU) {
        __syncthreads();
        for (int t = tid; t < TR * RL; t += BLK) {
            int row = row0 + t;
            if (row < k) {
                double sacc = s_tau[t][0];
                for (int i = 1; i < BLK / 64; ++i) sacc += s_tau[t][i];
                v.part_tau[(size_t)chunk * ld + row] = sacc;
            }
        }
    }
    if (TILED) __syncthreads();  // s_tau is reused by the next tile
    }
}
// Delayed-update mode (HISTORY.md §2.1): W = W0 + sum_j U[j] V[j]^T with at most J pending rank-1
// terms.  A normal pivot only READS W0 here (tau/v partials; the low-rank part of the two products is
// added in k_post_fused from the dots g_j = V[j].rho_K, h_j = U[j].t_K computed by the extra block
// row of this launch); every J-th pivot FOLDS: w = w0 + sum_j U[j][row] V[j][col] is formed in
// registers, used for the partials and written back.  W traffic per pivot: 8 k^2 (+ 16 k^2 / J)
// instead of 16 k^2.
template <int TR, bool WITH_V, bool NT, int RL = 1>
__global__ void __launch_bounds__(BLK) k_fused_lr(DevView v, int fold_only) {
    Ctl* c = v.ctl;
    if (!fold_only && (c->halt || c->it.status != ITER_PIVOT)) return;
    const int k = c->k, ld = v.ld;
    const int nlow = c->nlow;
    const bool fold = fold_only || c->fold;
    if (!fold) return;  // normal pivots are served by k_fused_w<.., LR = true>
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * (TR * RL);
    const int col0 = blockIdx.y * FW_TC;
    if (row0 >= k || col0 >= k) return;
    __shared__ double s_tau[TR * RL][BLK / 64];
    __shared__ double s_u[LR_MAX][TR];
    int cidx[4];
    cidx[0] = col0 + 2 * tid;
    cidx[1] = cidx[0] + 1;
    cidx[2] = col0 + 512 + 2 * tid;
    cidx[3] = cidx[2] + 1;
    const bool pair0 = cidx[1] < k, pair1 = cidx[3] < k;
    const bool one0 = cidx[0] < k, one1 = cidx[2] < k;
    double rk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rk[q] = (!fold_only && cidx[q] < k) ? v.rK[cidx[q]] : 0.0;
    double vacc[4] = {0.0, 0.0, 0.0, 0.0};
    const int wv = tid >> 6, l = tid & 63;
#pragma unroll 1
    for (int sub = 0; sub < RL; ++sub) {
        const int rbase = row0 + sub * TR;
        if (rbase >= k) break;
        double w[TR][4];
#pragma unroll
        for (int a = 0; a < TR; ++a) {
            int row = rbase + a;
            w[a][0] = w[a][1] = w[a][2] = w[a][3] = 0.0;
            if (row >= k) continue;
            const double* wp = v.W + (size_t)row * ld;
            if (pair0) {
                double2 t = fw_load2<NT>(wp + cidx[0]);
                w[a][0] = t.x;
                w[a][1] = t.y;
            } else if (one0) {
                w[a][0] = wp[cidx[0]];
            }
            if (pair1) {
                double2 t = fw_load2<NT>(wp + cidx[2]);
                w[a][2] = t.x;
                w[a][3] = t.y;
            } else if (one1) {
                w[a][2] = wp[cidx[2]];
            }
        }
        if (nlow > 0) {
            __syncthreads();  // the previous sub-tile's readers of s_u are done
            for (int i = tid; i < nlow * TR; i += BLK) {
                int j = i / TR, a = i % TR;
                int row = rbase + a;
                s_u[j][a] = row < k ? v.U[(size_t)j * ld + row] : 0.0;
            }
            __syncthreads();
            for (int j = 0; j < nlow; ++j) {
                const double* Vj = v.V + (size_t)j * ld;
                double vj[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) vj[q] = (cidx[q] < k) ? Vj[cidx[q]] : 0.0;
#pragma unroll
                for (int a = 0; a < TR; ++a) {
                    double u = s_u[j][a];
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[a][q] += u * vj[q];
                }
            }
#pragma unroll
            for (int a = 0; a < TR; ++a) {
                int row = rbase + a;
                if (row >= k) continue;
                double* wp = v.W + (size_t)row * ld;
                if (pair0) fw_store2<NT>(wp + cidx[0], w[a][0], w[a][1]);
                else if (one0) wp[cidx[0]] = w[a][0];
                if (pair1) fw_store2<NT>(wp + cidx[2], w[a][2], w[a][3]);
                else if (one1) wp[cidx[2]] = w[a][2];
            }
        }
        if (fold_only) continue;
#pragma unroll
        for (int a = 0; a < TR; ++a) {
            int row = rbase + a;
            double tacc = w[a][0] * rk[0] + w[a][1] * rk[1] + w[a][2] * rk[2] + w[a][3] * rk[3];
            if (WITH_V && row < k) {
                double t = v.tK[row];
#pragma unroll
                for (int q = 0; q < 4; ++q) vacc[q] += w[a][q] * t;
            }
            double sacc = wave_sum(tacc);
            if (l == 0) s_tau[sub * TR + a][wv] = sacc;
        }
    }
    if (fold_only) return;
    if (WITH_V) {
        double* pv = v.part_v + (size_t)blockIdx.x * ld;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cidx[q] < k) pv[cidx[q]] = vacc[q];
    }
    __syncthreads();
    for (int t = tid; t < TR * RL; t += BLK) {
        int row = row0 + t;
        if (row < k) {
            double sacc = s_tau[t][0];
            for (int i = 1; i < BLK / 64; ++i) sacc += s_tau[t][i];
            v.part_tau[(size_t)blockIdx.y * ld + row] = sacc;
        }
    }
}

// Streaming pass of the large-nucleus delayed-update mode (every pivot): tau_K = W0 rho_K and v_K = W0^T t_K partials
// in ONE read of W0, nothing written back.  Shape found with tools/stream_bench.hip (MI355X, k = 20 480: 546 us =
// 6.1 TB/s against 645 us for the 16 x 1024 tiles of k_fused_w; k = 10 240: 127 us = 6.6 TB/s against 161 us; a bare
// sum of W with the same loads runs at 7.0 / 6.4 TB/s):
//   * a block owns a strip of SW_CH = 512 columns x SW_RB = 512 rows and walks it in steps of SW_RS = 8 rows; a thread
//     owns one pair of columns, so a step is 8 independent 16-byte non-temporal loads per lane, and the loads of step
//     i + 1 are issued before step i is consumed (register double buffer);
//   * v accumulates in registers over the whole strip: part_v has k / 512 rows instead of k / 16 (the old tiling wrote
//     and re-read 8 k^2 / 16 bytes of partials per pivot: 100 us of k_post_fused at k = 20 000);
//   * the 8 row sums of a step go through an LDS transpose (two alternating buffers, one barrier per step).
// The last LR_MAX blocks compute the low-rank dots g_j = V[j].rho_K, h_j = U[j].t_K for k_post_fused.  On a folding
// pivot the fold kernel (mode 2) has already applied the pending terms to W0, so the dots are skipped and the pass
// reads the folded matrix.
constexpr int SW_MAX_BLOCKS = 8192;
// grid of the fold (tiles are strided over by its blocks; tools/rw_bench.hip: a bare in-place
// read + write pass over the same matrix runs in 1 140 us with 4 096 blocks against 1 237 / 1 265 with 2 048 / 8 192)
static int fold_max_blocks() {
    return SW_MAX_BLOCKS;  // (1 024 ... 8 192 measured alike in situ)
}
// rows per strip of the streaming pass: fixed (template value) or balanced over the co-resident blocks (DevView.sw_nbal)
__device__ __forceinline__ int sw_strip_rows(const DevView& v, int k, int ch, int rs, int rb_fixed) {
    if (!v.sw_nbal || rb_fixed != 128) return rb_fixed;  // (only the partials of k_stream_w's default geometry are balanced)
    const int nch = (k + ch - 1) / ch;
    const int mult = v.wshard ? v.world : 1;     // row-sharded pass: every rank streams 1 / world of the strips
    const int spc = max(1, (v.sw_nbal * mult) / nch);  // strips per column chunk
    int rb = (k + spc - 1) / spc;
    rb = ((rb + rs - 1) / rs) * rs;
    return max(rb, 32);  // (part_v holds cap / 8 rows of partials)
}
template <bool WITH_V, int SW_CH, int SW_RB, int SW_RS>
__global__ void __launch_bounds__(BLK, 4) k_stream_w(DevView v, int with_tau, int skip_on_fold = 0) {  // 4 waves per SIMD: keep the double buffer within 128 VGPRs
    static_assert(SW_CH == 2 * BLK || SW_CH == 4 * BLK, "one or two column pairs per thread");
    constexpr int NP = SW_CH / (2 * BLK);
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    if (skip_on_fold && c->fold) return;  // the fold of this pivot produced the v partials itself (k_fold_w, fuse_v)
    KMARK0(c, 21);
    const int k = c->k, ld = v.ld;
    const int tid = threadIdx.x;
    const int n_tile_blocks = (int)gridDim.x - LR_MAX;
    if ((int)blockIdx.x >= n_tile_blocks) {
        const int j = (int)blockIdx.x - n_tile_blocks;
        if (c->fold || j >= c->nlow) return;
        const double* Vj = v.V + (size_t)j * ld;
        const double* Uj = v.U + (size_t)j * ld;
        double g = 0.0, h = 0.0;
        for (int s = tid; s < k; s += BLK) {
            g += Vj[s] * v.rK[s];
            if (WITH_V) h += Uj[s] * v.tK[s];
        }
        g = block_sum(g);
        h = block_sum(h);
        if (tid == 0) {
            c->lr_g[j] = g;
            c->lr_h[j] = h;
        }
        return;
    }
    __shared__ double s_t[2][SW_RS][BLK + 1];
    constexpr int G = BLK / SW_RS;  // lanes that share a row in the reduction of a step (32)
    const int rrow = tid / G, gl = tid % G;
    const int rb = sw_strip_rows(v, k, SW_CH, SW_RS, SW_RB);
    const int nch = (k + SW_CH - 1) / SW_CH, nstr_all = (k + rb - 1) / rb;
    // row-sharded pass: this rank streams only the strips s with s % world == rank (k_post_exchange completes the products)
    const int sw = v.wshard ? v.world : 1, sr = v.wshard ? v.rank : 0;
    const int nstr = nstr_all > sr ? (nstr_all - sr + sw - 1) / sw : 0;
    const double* __restrict__ Wp = v.W;
    const double* __restrict__ tKp = v.tK;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += n_tile_blocks) {
        const int strip = (tile / nch) * sw + sr, chunk = tile % nch;
        const int rbeg = strip * rb, rend = min(k, rbeg + rb);
        int c0[NP];
        bool pair[NP], one[NP];
        double rk0[NP], rk1[NP], vacc0[NP], vacc1[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c0[p] = chunk * SW_CH + p * 2 * BLK + 2 * tid;
            pair[p] = c0[p] + 1 < k;
            one[p] = c0[p] < k;
            rk0[p] = one[p] ? v.rK[c0[p]] : 0.0;
            rk1[p] = pair[p] ? v.rK[c0[p] + 1] : 0.0;
            vacc0[p] = vacc1[p] = 0.0;
        }
        // Branch-free loads: rows beyond the strip are clamped to its last row (their products are masked through
        // t = 0 and never stored), columns beyond k to column 0 (masked by select); a lone last column (odd k) is
        // read as a pair whose second half lies in the row's padding (ld > k there) and is masked.  Conditional
        // loads would split the loop into many basic blocks (see k_fold_w).
        const double* wcol[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) wcol[p] = Wp + (one[p] ? c0[p] : 0);
        dbl2_t w[SW_RS][NP], wn[SW_RS][NP];
        auto load_step = [&](dbl2_t (&dst)[SW_RS][NP], int r0) {
#pragma unroll
            for (int a = 0; a < SW_RS; ++a) {
                const size_t roff = (size_t)min(r0 + a, rend - 1) * ld;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const dbl2_t t = __builtin_nontemporal_load(reinterpret_cast<const dbl2_t*>(wcol[p] + roff));
                    dst[a][p].x = one[p] ? t.x : 0.0;
                    dst[a][p].y = pair[p] ? t.y : 0.0;
                }
            }
        };
        load_step(w, rbeg);
        int buf = 0;
        for (int r0 = rbeg; r0 < rend; r0 += SW_RS, buf ^= 1) {
            if (r0 + SW_RS < rend) load_step(wn, r0 + SW_RS);  // (uniform branch: the last step has no successor)
#pragma unroll
            for (int a = 0; a < SW_RS; ++a) {
                double sacc = 0.0;
                const double t = (WITH_V && r0 + a < rend) ? tKp[r0 + a] : 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    sacc += w[a][p].x * rk0[p] + w[a][p].y * rk1[p];
                    if (WITH_V) {
                        vacc0[p] += w[a][p].x * t;
                        vacc1[p] += w[a][p].y * t;
                    }
                }
                if (with_tau) s_t[buf][a][tid] = sacc;
            }
            if (with_tau) {  // (uniform) lazy dual steepest edge: the primal loop skips tau = B^-1 rho altogether — no
                             // row sums, no barrier: the pass is then a pure accumulation of v in registers
                __syncthreads();  // one barrier per step: the two LDS buffers alternate
                double sum = 0.0;
#pragma unroll
                for (int j = 0; j < BLK / G; ++j) sum += s_t[buf][rrow][gl + j * G];
                sum = group_sum<G>(sum);
                if (gl == 0 && r0 + rrow < rend) v.part_tau[(size_t)chunk * ld + r0 + rrow] = sum;
            }
#pragma unroll
            for (int a = 0; a < SW_RS; ++a)
#pragma unroll
                for (int p = 0; p < NP; ++p) w[a][p] = wn[a][p];
        }
        if (with_tau) __syncthreads();  // the next tile's first step may reuse the buffer the slowest wave is still reducing
        if (WITH_V) {
            double* pv = v.part_v + (size_t)strip * ld;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (one[p]) pv[c0[p]] = vacc0[p];
                if (pair[p]) pv[c0[p] + 1] = vacc1[p];
            }
        }
    }
}

// Fold of the pending rank-1 terms into W0 for the large-nucleus mode:  W0[r][c] += sum_j U[j][r] V[j][c], j < nlow
// (one explicit FMA per term, in term order).  One read and one write of W0, 2 JM flops per
// element: memory-bound as long as the FMAs hide behind the stream, so the kernel is shaped like k_stream_w:
//   * a block owns FD_CH = 256 columns x FD_RB rows; a thread owns ONE column and keeps the JM values V[j][col] in
//     registers for the whole strip (a pair of columns would need 2 JM registers and halve the occupancy);
//   * rows are walked in steps of FD_RS = 8: the 8 x JM values U[j][row] of a step are staged in LDS (double buffer,
//     one barrier per step) and read back as broadcast 16-byte words; the loads of step i + 1 are issued before step i
//     is consumed.
// mode 1: host-requested (flush outside the pivot loop); mode 2: a folding pivot (Ctl.fold), k_stream_w follows.
constexpr int FD_CH = 256, FD_RB = 256, FD_RS = 8;
// The hot loop is branch-free: rows and columns beyond the edge are CLAMPED (the loads are unconditional, their
// results masked or never stored).  With `cond ? load : 0` forms the loop breaks into ~20 basic blocks and the
// register allocator spills the V values (measured: 1.8 KB of scratch per lane).
// The same fold with the U side read through the SCALAR unit (round 3).  k_fold_w stages the FD_RS x JM values U[j][row]
// of a step in LDS and every thread reads them back as broadcasts: 16 ds_read_b128 per element row — at 420 M elements per
// fold that is ~105 M LDS wave-instructions, which bound the kernel (1.43 ms = 4.7 TB/s at k = 20 500 where a read + write
// stream of the matrix takes ~1.2 ms).  The values of one row are the same for every lane of the wave, so they belong in
// scalar registers: lowrank_append keeps a slot-major copy Ut[slot][j], a row's JM values are one contiguous 256-byte
// block, loaded with s_load (constant address space: nothing in this kernel writes Ut or t_K) and used as the scalar operand
// of the FMAs.  No LDS, no barriers.  Measured at k = 20 500 (rocprofv3): 1 338 us against 1 426 us, 695 vs 707 us per pivot
// of the late window (the LDS form was kept for the A/B until round 6).  What bounds it now is the scalar-load latency per
// row (one batch, one wait: s_load returns out of order, there is no partial wait), hidden only by the 3 waves per SIMD.
typedef const __attribute__((address_space(4))) double const_f64;
typedef double sreg8 __attribute__((ext_vector_type(8)));  // 16 consecutive SGPRs
template <int JM>
__global__ void __launch_bounds__(BLK, (JM <= 16 ? 4 : JM <= 32 ? 3 : 2)) k_fold_w2(DevView v, int mode, int fuse_v) {
    Ctl* c = v.ctl;
    if (mode != 1 && (c->halt || c->it.status != ITER_PIVOT)) return;
    if (!(mode == 1 || c->fold)) return;
    const int k = c->k, ld = v.ld;
    const int nlow = min(c->nlow, JM);
    if (nlow <= 0 || k <= 0) return;
    const int tid = threadIdx.x;
    const int nch = (k + FD_CH - 1) / FD_CH, nstr = (k + FD_RB - 1) / FD_RB;
    double* __restrict__ Wp = v.W;
    const double* __restrict__ Vp = v.V;
    const_f64* Utc = (const_f64*)(uintptr_t)v.Ut;
    const_f64* tKc = (const_f64*)(uintptr_t)v.tK;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += gridDim.x) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * FD_RB, rend = min(k, rbeg + FD_RB);
        const int col = chunk * FD_CH + tid;
        const bool active = col < k;
        const int colc = active ? col : k - 1;
        double vj[JM];
#pragma unroll
        for (int j = 0; j < JM; ++j) {
            const double x = Vp[(size_t)min(j, nlow - 1) * ld + colc];
            vj[j] = (active && j < nlow) ? x : 0.0;
        }
        double* wcol = Wp + colc;
        double w[FD_RS], wn[FD_RS];
        double vacc = 0.0;
#pragma unroll
        for (int a = 0; a < FD_RS; ++a) w[a] = __builtin_nontemporal_load(wcol + (size_t)min(rbeg + a, rend - 1) * ld);
        for (int r0 = rbeg; r0 < rend; r0 += FD_RS) {
            if (r0 + FD_RS < rend) {  // uniform branch: the last step has no successor
#pragma unroll
                for (int a = 0; a < FD_RS; ++a) wn[a] = __builtin_nontemporal_load(wcol + (size_t)min(r0 + FD_RS + a, rend - 1) * ld);
            }
#pragma unroll
            for (int a = 0; a < FD_RS; ++a) {
                const int row = min(r0 + a, rend - 1);  // (uniform: rows beyond the strip repeat its last row, their results are dropped)
                // Scalar loads return out of order — the only wait there is is "all of them" — so a row's values are fetched
                // as ONE batch, spelled out: left to the scheduler the 64-byte groups of several rows are interleaved, each
                // with its own full wait (measured 716 vs 697 us per pivot).
                double acc = w[a], tk;
                const_f64* ur = Utc + (size_t)row * LR_MAX;
                const_f64* tp = tKc + row;
                if constexpr (JM == 16) {
                    sreg8 u0, u1;
                    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx2 %2, %4, 0x0\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(u0), "=&s"(u1), "=&s"(tk) : "s"(ur), "s"(tp) : "memory");
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = __builtin_fma(u0[j], vj[j], acc);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = __builtin_fma(u1[j], vj[8 + j], acc);
                } else {
#pragma unroll
                    for (int h = 0; h < JM / 32; ++h) {
                        sreg8 u0, u1, u2, u3;
                        asm volatile("s_load_dwordx16 %0, %5, 0x0\n\ts_load_dwordx16 %1, %5, 0x40\n\ts_load_dwordx16 %2, %5, 0x80\n\t"
                                     "s_load_dwordx16 %3, %5, 0xc0\n\ts_load_dwordx2 %4, %6, 0x0\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&s"(u0), "=&s"(u1), "=&s"(u2), "=&s"(u3), "=&s"(tk) : "s"(ur + 32 * h), "s"(tp) : "memory");
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = __builtin_fma(u0[j], vj[32 * h + j], acc);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = __builtin_fma(u1[j], vj[32 * h + 8 + j], acc);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = __builtin_fma(u2[j], vj[32 * h + 16 + j], acc);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = __builtin_fma(u3[j], vj[32 * h + 24 + j], acc);
                    }
                }
                w[a] = acc;
                vacc = __builtin_fma(acc, (fuse_v && r0 + a < rend) ? tk : 0.0, vacc);
            }
            if (active) {
                if (r0 + FD_RS <= rend) {
#pragma unroll
                    for (int a = 0; a < FD_RS; ++a) __builtin_nontemporal_store(w[a], wcol + (size_t)(r0 + a) * ld);
                } else {
#pragma unroll
                    for (int a = 0; a < FD_RS; ++a)
                        if (r0 + a < rend) __builtin_nontemporal_store(w[a], wcol + (size_t)(r0 + a) * ld);
                }
            }
#pragma unroll
            for (int a = 0; a < FD_RS; ++a) w[a] = wn[a];
        }
        if (fuse_v && active) v.part_v[(size_t)strip * ld + col] = vacc;
    }
}
// (Measured and rejected in round 3: the same update on the matrix cores — v_mfma_f64_16x16x4_f64, operands one f64 per lane
// from plain vector loads, 235 VGPRs, 2 waves per SIMD: correct, 707 us per pivot of the late window against 698 for this
// form.  The f64 matrix rate of gfx950 equals its vector rate, and the 16 x 16 tile layout turns every load of W into four
// 128-byte row segments.)
// Sparse tableau row / sparse ratio test: a batch that starts on state it cannot trust (a dense sweep ran in between, or an
// iteration stamped and listed columns without closing — ITER_STALL retry, halted batch) advances the stamp epoch and clears
// both list counters, so that no stale stamp can hide a column from the next iteration's lists.
__global__ void k_str_reset(DevView v) {
    v.ctl->hyper_epoch += 1;
    v.ctl->aq_n = 0;
    v.ctl->str_n = 0;
}
void launch_str_reset(const DevView& dv, hipStream_t st) { hipLaunchKernelGGL(k_str_reset, dim3(1), dim3(1), 0, st, dv); }
__global__ void k_reset_nlow(DevView v) {
    v.ctl->nlow = 0;
    v.ctl->fold = 0;
}
// Order-independent checksum of the k x k nucleus inverse and of its slot maps (tests of the sharded path: every rank of a
// solve must hold the same bits): the sum over (i, j) of bits(W[i][j]) * (2 (i k + j) + 1) mod 2^64, plus the two maps.
__global__ void __launch_bounds__(BLK) k_checksum_w(DevView v, unsigned long long* out) {
    const int k = v.ctl->k;
    unsigned long long acc = 0ull;
    const long total = (long)k * k;
    for (long e = (long)blockIdx.x * BLK + threadIdx.x; e < total; e += (long)gridDim.x * BLK) {
        const int i = (int)(e / k), j = (int)(e % k);
        acc += (unsigned long long)__double_as_longlong(v.W[(size_t)i * v.ld + j]) * (2ull * (unsigned long long)e + 1ull);
    }
    for (int s = blockIdx.x * BLK + threadIdx.x; s < k; s += gridDim.x * BLK)
        acc += (unsigned long long)(unsigned)v.pos_of_kslot[s] * 0x9E3779B97F4A7C15ull + (unsigned long long)(unsigned)v.row_of_kslot[s] * 0xC2B2AE3D27D4EB4Full * (unsigned long long)(s + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
void launch_checksum_w(const DevView& dv, unsigned long long* out, hipStream_t st) {
    hipLaunchKernelGGL(k_checksum_w, dim3(2048), dim3(BLK), 0, st, dv, out);
}

// Row-sharded streaming pass, step 2 of 3 (k_stream_w -> k_post_exchange -> k_post_fused): reduce this rank's partials
// (tau_K on the rows of its own strips, complete; v_K over its own strips, partial) and write them into every rank's
// exchange buffer; the last block to finish makes the stores visible system-wide and raises this rank's flag (mailbox
// kind 4) in every box.  k_post_fused then waits for all flags and reads only local memory.
template <int TR, int TC>
__global__ void __launch_bounds__(BLK) k_post_exchange(DevView v, int with_v, int with_tau) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int k = c->k, ld = v.ld;
    const unsigned long long ep = c->xepoch[4] + 1ull;
    const size_t par = (size_t)(ep & 1ull);
    const int i = blockIdx.x * BLK + threadIdx.x;
    const int tr = sw_strip_rows(v, k, TC, 4, TR);
    if (i < k) {
        const int strip = i / tr;
        const bool mine = with_tau && strip % v.world == v.rank;
        double x = 0.0;
        if (mine) {
            const int nchunks = (k + TC - 1) / TC;
            for (int j = 0; j < nchunks; ++j) x += v.part_tau[(size_t)j * ld + i];
        }
        double sv = 0.0;
        if (with_v) {
            const int nstripes = (k + tr - 1) / tr;
            for (int t = v.rank; t < nstripes; t += v.world) sv += v.part_v[(size_t)t * ld + i];
        }
        const size_t off = ((par * v.world + v.rank) * 2) * (size_t)v.xb_cap + i;
        for (int r = 0; r < v.world; ++r) {
            double* dst = v.xbuf_peer[r] + off;
            if (mine) __hip_atomic_store(dst, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (with_v) __hip_atomic_store(dst + v.xb_cap, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();  // this thread's stores have reached the peers ...
    __syncthreads();         // ... and so have the whole block's, before its first thread takes the ticket
    if (!last_block_arrives(v.ticket, gridDim.x)) return;
    if (threadIdx.x == 0) {
        *v.ticket = 0;
        c->xepoch[4] = ep;
    }
    if (threadIdx.x < 64) {
        double f[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        mail_post_wave(v, 4, ep, f, threadIdx.x);
        // ... and this ONE wave waits for every rank's flag, so that the next kernel (k_post_fused) finds all the data
        // in local memory without waiting: a wait inside k_post_fused would park thousands of spinning blocks on the
        // CUs (and starve the peers' kernels when several ranks share a GPU)
        bool ok = true;
        for (int r = threadIdx.x; r < v.world; r += 64) {
            double h[7];
            if (!mail_wait(mail_slot(v, 4, ep, r), ep, h)) ok = false;
        }
        ok = __all(ok);
        if (!ok && threadIdx.x == 0) comm_fail(c, 0);
    }
}
// After the fused pass (horizontally fused): blocks [0, n_push) finish tau = B^-1 rho by position
// (tau_K from the per-chunk partials in a fixed order, then the push of -F tau_K, solver.rs:1157);
// the remaining blocks reduce the v partials in a fixed order and scatter v_K by row (solver.rs:1114).
template <bool OWN_RK = false>
__device__ __forceinline__ void row_touch_body(const DevView& v, Ctl* c, int block, double own_rk = 0.0);  // (sparse tableau row, below)
template <int G, bool WITH_V, int TR, int TC = FW_TC>
__global__ void __launch_bounds__(BLK) k_post_fused(DevView v, int n_push, int touch_from = -1, int fold_fused = 0, int rk_from = -1) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 9);
    if (rk_from >= 0 && (int)blockIdx.x >= rk_from) {
        // rho_K rides behind the v tail (round 4; large nucleus, lazy primal iteration): nothing between the ratio test and the
        // tableau row reads rho — the streaming pass forms v only — so the BTRAN launch (10.6 us at k = 20 500) is gone and its
        // blocks run beside this 6 us reduction.  On a folding pivot the fold has already run: W0 holds every term (after_fold).
        btran_rk_body(v, c, (int)blockIdx.x - rk_from, (int)gridDim.x - rk_from, 1);
        return;
    }
    if (touch_from >= 0 && (int)blockIdx.x >= touch_from) {  // horizontally fused: the touched-column list of the sparse tableau
        row_touch_body(v, c, (int)blockIdx.x - touch_from);   // row needs rho only (the BTRAN before this launch), nothing of this pass
        return;
    }
    const int k = c->k;
    // row-sharded streaming pass: the partials were exchanged by k_post_exchange; read tau_K (each row from its owner's
    // slot) and the v_K partials (summed in rank order) from the local exchange buffer
    // (only the strip tiling goes through the exchange: the classic 8- / 16-row tiling of the dense solves outside the pivot loop —
    // recalc_basic_vals of the polish step, recalc_obj_coeffs — stays replicated.  Round 4: a sharded solve that reached the polish
    // with a large nucleus read the stale exchange buffer here and reported an objective of zero.)
    const bool xsh = v.wshard != 0 && TR > 16;
    const double* xb = nullptr;
    if (xsh) {  // k_post_exchange (previous launch) returned only after every rank's flag had arrived (or set halt)
        const unsigned long long ep = c->xepoch[4];
        xb = v.xbuf + (size_t)(ep & 1ull) * v.world * 2 * (size_t)v.xb_cap;
    }
    if ((int)blockIdx.x < n_push) {
        // G lanes per slot serve the push of -F tau_K; when that product is computed elsewhere (blocked push, pulled
        // form) one lane per slot does the reduction and the blocks beyond k / 256 exit at once
        const bool solo = v.pb_on || v.det_pull;
        int slot = solo ? (int)(blockIdx.x * BLK + threadIdx.x) : (int)(blockIdx.x * BLK + threadIdx.x) / G;
        int gl = solo ? 0 : (int)(threadIdx.x & (G - 1));
        if (slot >= k) return;
        const int nchunks = (k + TC - 1) / TC;
        double x = 0.0;
        if (xsh) {
            const int owner = (slot / sw_strip_rows(v, k, TC, 4, TR)) % v.world;
            x = __hip_atomic_load(xb + ((size_t)owner * 2) * v.xb_cap + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            for (int j = 0; j < nchunks; ++j) x += v.part_tau[(size_t)j * v.ld + slot];
        }
        if (v.lrJ && !c->fold) {  // low-rank part of W_eff * rho_K
            const int nlow = c->nlow;
            for (int j = 0; j < nlow; ++j) x += v.U[(size_t)j * v.ld + slot] * c->lr_g[j];
        }
        int p = v.pos_of_kslot[slot];
        if (gl == 0) {
            v.tauK[slot] = x;
            v.tau[p] = x;
        }
        if (x != 0.0 && !v.pb_on && !v.det_pull) push_F<G>(v, p, x, v.tau, gl);
        return;
    }
    if (!WITH_V) return;
    // 32 slots x 8 stripe groups per block; LDS combines the 8 partial sums in a fixed order
    const int b = (int)blockIdx.x - n_push;
    const int lane32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = b * 32 + lane32;
    if (b * 32 >= k) return;
    __shared__ double s_part[8][33];
    // (balanced strips exist for the default geometry only: 4 rows per step; on a folding pivot whose fold produced the
    // partials itself they come in the fold kernel's strips of FD_RB rows)
    const int tr = (fold_fused && c->fold) ? FD_RB : sw_strip_rows(v, k, TC, 4, TR);
    const int nstripes = (k + tr - 1) / tr;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (xsh) {
        if (grp == 0 && i < k)  // rank order: every rank forms the identical sum
            for (int r = 0; r < v.world; ++r)
                s0 += __hip_atomic_load(xb + ((size_t)r * 2 + 1) * v.xb_cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (i < k) {
        int t = grp;
        for (; t + 24 < nstripes; t += 32) {  // four independent loads in flight per lane
            s0 += v.part_v[(size_t)t * v.ld + i];
            s1 += v.part_v[(size_t)(t + 8) * v.ld + i];
            s2 += v.part_v[(size_t)(t + 16) * v.ld + i];
            s3 += v.part_v[(size_t)(t + 24) * v.ld + i];
        }
        for (; t < nstripes; t += 8) s0 += v.part_v[(size_t)t * v.ld + i];
    }
    if (i < k && v.lrJ && !c->fold) {  // low-rank part of W_eff^T * t_K, spread over the eight stripe groups as well
        const int nlow = c->nlow;
        for (int j = grp; j < nlow; j += 8) s1 += v.V[(size_t)j * v.ld + i] * c->lr_h[j];
    }
    s_part[grp][lane32] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < k) {
        double sv = s_part[0][lane32];
#pragma unroll
        for (int g2 = 1; g2 < 8; ++g2) sv += s_part[g2][lane32];
        v.vK[i] = sv;
        v.rv[v.row_of_kslot[i]].y = sv;
    }
}
// v-only reduction for the dense transposed solve of recalc_obj_coeffs
template <int TR>
__global__ void __launch_bounds__(BLK) k_reduce_v(DevView v) {
    Ctl* c = v.ctl;
    const int k = c->k;
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= k) return;
    const int nstripes = (k + TR - 1) / TR;
    double s = 0.0;
    for (int t = 0; t < nstripes; ++t) s += v.part_v[(size_t)t * v.ld + i];
    v.vK[i] = s;
    v.rv[v.row_of_kslot[i]].y = s;
}

// ------------------------------------------------------------------- K8: updates after the pivot
// basic side: solver.rs:1049-1058 (x_B, bounds), 1164-1173 (dual steepest-edge norms)
// non-basic side: solver.rs:1068-1080 (value/state of the leaving var, reduced costs), 1140-1150 (PSE)
// bound flip: solver.rs:1031-1042
// The same pass (a) zeroes this iteration's work vectors behind itself (alpha_q, tau, rho|v are each
// read for the last time here), and (b) prices the NEXT iteration from the values it has just
// written (K1 when next_phase = 0, K6 when 1), so neither a memset nor a pricing kernel is needed
// inside the replayed graph.
__global__ void __launch_bounds__(BLK) k_update_pivot(DevView v, int phase, int use_dse, int use_pse, int inline_comb, int n_upd, int pull_inside) {
    Ctl* c = v.ctl;
    if (c->halt) return;
    const IterState* it = &c->it;
    const int status = it->status;
    if (status != ITER_PIVOT && status != ITER_FLIP) return;
    KMARK0(c, 12);
    if ((int)blockIdx.x >= n_upd) {  // horizontally fused (dual iteration without PSE): the partition change, which touches
        if (status == ITER_PIVOT)    // W and the slot maps only — nothing the update blocks read or write
            struct_update_body(v, c, ((int)blockIdx.x - n_upd) * BLK + threadIdx.x);
        return;
    }
    const bool flip = status == ITER_FLIP;
    const int r = flip ? -1 : it->r, q = it->q;
    const double pc = it->pivot_coeff;
    // pull_inside (small-nucleus primal head, one position per thread): the sparse tableau row — alpha_rj = rho . a_j and the PSE helper
    // v . a_j on the columns the head stamped (solver.rs:685-692, 1126-1132) — is pulled HERE, by the workgroup that updates those
    // positions, instead of by a launch of its own (k_row_pull, 5.8 us + a kernel boundary for ~3 MB of gathers): each workgroup
    // compacts the stamped positions of its 256 into LDS, 32 lanes pull each column in storage order (k_row_pull<32, 1>'s sums, bit
    // for bit), and the update below reads the pair from LDS.  alpha_r / helper are not materialised; (rho, v) is not zeroed here —
    // other workgroups are still reading it — but by the next head, from the list this iteration's head left (Ctl.rv_n).
    constexpr int UPT = 4;  // positions per thread a pulling launch may take at most (launch_update_pivot chooses: one up to 512 workgroups)
    __shared__ int s_tl[BLK * UPT];
    __shared__ double s_ta[BLK * UPT], s_th[BLK * UPT];
    __shared__ int s_tcnt;
    int my_slot[UPT] = {-1, -1, -1, -1};
    // (a pulling launch loads its positions' d / gamma / flags HERE, ahead of the pull: behind it, one trip's stores kept the next trip's loads
    // from being issued early — four dependent round trips, 3 us of the 7 a workgroup took in the early window of config 4)
    double pre_d[UPT] = {0.0, 0.0, 0.0, 0.0}, pre_g[UPT] = {1.0, 1.0, 1.0, 1.0};
    uint8_t pre_f[UPT] = {0, 0, 0, 0};
    const bool pre = pull_inside && !flip;
    if (pull_inside) {
        if (threadIdx.x == 0) s_tcnt = 0;
        __syncthreads();
        if (!flip) {
            const int ep = c->hyper_epoch + 1;
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const int t = blockIdx.x * BLK + threadIdx.x + i * n_upd * BLK;
                if (t < v.n) {
                    const int tn = v.nb_order ? v.nb_order[t] : t;
                    pre_d[i] = v.d[tn];
                    if (use_pse) pre_g[i] = v.gamma[tn];
                    pre_f[i] = v.nbflags[tn];
                    if (tn >= v.nb_lo && tn < v.nb_hi && v.hy_stamp_n[tn] == ep) {
                        my_slot[i] = atomicAdd(&s_tcnt, 1);
                        s_tl[my_slot[i]] = tn;
                    }
                }
            }
        }
        __syncthreads();
        const int cnt = s_tcnt;
        const int gl = threadIdx.x & 31;
        for (int i = threadIdx.x >> 5; i < cnt; i += BLK / 32) {
            const int2 rg = v.nb_rng[s_tl[i]];
            double a1 = 0.0, a2 = 0.0;
            // (four entries of the lane at a time: the loads of a trip are in flight together — one entry per trip was two dependent
            // round trips per 32 entries of the column, and the workgroup with the most stamped columns closes the launch; per lane
            // the same sums in the same order)
            for (int e0 = rg.x + gl; e0 < rg.y; e0 += 4 * 32) {
                double a[4];
                int rw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 32;
                    a[u] = e < rg.y ? v.csc_val[e] : 0.0;
                    rw[u] = e < rg.y ? v.csc_row[e] : -1;
                }
                double2 tt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) tt[u] = rw[u] >= 0 ? v.rv[rw[u]] : make_double2(0.0, 0.0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (rw[u] >= 0) {
                        a1 += a[u] * tt[u].x;
                        a2 += a[u] * tt[u].y;
                    }
            }
            a1 = group_sum<32>(a1);
            a2 = group_sum<32>(a2);
            if (gl == 0) {
                s_ta[i] = a1;
                s_th[i] = a2;
            }
        }
        __syncthreads();
    }
    Cand cand = cand_none();
    double cand_d = 0.0;
    // one position per thread on the usual grid; the loop only strides for very large models and for a pulling launch (launch_update_pivot)
    const int tmax = v.m > v.n ? v.m : v.n;
    // pull_inside == 2: the head has applied the basic side (x_B on supp(alpha_q), position r, alpha_q back to zero) and position q's own
    // entries (d, gamma, flags, nb_vars, x_N) itself: here only the non-basic side remains — the touched columns' d / gamma, the drift
    // check at q, and the pricing scan
    const bool head_applied = pull_inside == 2;
    const int tmax_eff = head_applied ? v.n : tmax;
    // Dual iteration whose Harris test ran over the listed non-zeros of alpha_r (ratio_dual_list): the non-basic side walks that list —
    // d changes on the non-zeros of the row only (solver.rs:1073-1080), and nothing prices the non-basic side in a dual iteration — instead
    // of loading d / flags / alpha_r of all n positions (6.8 MB of the kernel's 26 on the 400 000-column transport instance).
    const int keep = (phase == 1 && !use_pse && !inline_comb && !pull_inside && !flip && !v.nb_order && v.ar_list) ? c->ar_keep : -1;
    int trip = 0;
    for (int t = blockIdx.x * BLK + threadIdx.x; t < tmax_eff; t += n_upd * BLK, ++trip) {
        Cand tc = cand_none();
        double tc_d = 0.0;
        // (a select chain, not an indexed read: the array stays in registers)
        const int slot_t = !pull_inside ? -1 : (trip == 0 ? my_slot[0] : (trip == 1 ? my_slot[1] : (trip == 2 ? my_slot[2] : (trip == 3 ? my_slot[3] : -1))));
        if (t < v.m && !head_applied) {
            double a = v.alpha_q[t];
            double xb = v.xB[t], lo = v.loB[t], hi = v.hiB[t], bt = use_dse ? v.beta[t] : 1.0;
            const double tau_t = (use_dse && !flip) ? v.tau[t] : 0.0;  // loaded with the others, not behind `a != 0`
            if (t == r) {
                int ev = it->entering_var;
                xb = it->entering_new_val;
                lo = v.var_lo[ev];
                hi = v.var_hi[ev];
                v.xB[r] = xb;
                v.loB[r] = lo;
                v.hiB[r] = hi;
                if (use_dse) {
                    bt = it->rho_sq / (pc * pc);
                    v.beta[r] = bt;
                }
                v.basic_vars[r] = ev;
                v.var_loc[ev] = r;
                v.var_loc[it->leaving_var] = -1 - q;
                if (v.fpk_x) v.fpk_x[it->leaving_var] = 0.0;  // (fpull.inc: alpha_K by variable is zero outside the nucleus basics)
            } else if (a != 0.0) {
                xb -= it->entering_diff * a;
                v.xB[t] = xb;
                if (use_dse && !flip) {
                    bt += -2.0 * a * tau_t / pc + it->rho_sq * a * a / (pc * pc);
                    v.beta[t] = bt;
                }
            }
            v.alpha_q[t] = 0.0;
            v.tau[t] = 0.0;
            if (!pull_inside) v.rv[t] = make_double2(0.0, 0.0);
            if (phase == 1 && !c->forced) tc = price_dual_one(xb, lo, hi, bt, t, use_dse);
        }
        if (keep >= 0 ? t < keep : t < v.n) {
            // non-basic side: thread t serves position tn = nb_order[t] (locality order of the banded sweep: its partials
            // are indexed by t), or position t itself without an order, or entry t of the list of alpha_r's non-zeros
            const int tn = keep >= 0 ? v.ar_list[t] : (v.nb_order ? v.nb_order[t] : t);
            double dd, gm;
            uint8_t f;
            if (pre && trip < UPT) {  // (select chains, not indexed reads: the arrays stay in registers)
                dd = trip == 0 ? pre_d[0] : (trip == 1 ? pre_d[1] : (trip == 2 ? pre_d[2] : pre_d[3]));
                gm = trip == 0 ? pre_g[0] : (trip == 1 ? pre_g[1] : (trip == 2 ? pre_g[2] : pre_g[3]));
                f = trip == 0 ? pre_f[0] : (trip == 1 ? pre_f[1] : (trip == 2 ? pre_f[2] : pre_f[3]));
            } else {
                dd = v.d[tn];
                gm = use_pse ? v.gamma[tn] : 1.0;
                f = v.nbflags[tn];
            }
            if (tn == q && head_applied) {
                if (!flip && q >= v.nb_lo && q < v.nb_hi) {
                    // the pivot element computed two ways (FTRAN side / BTRAN side) measures the drift of W
                    const double ba = slot_t >= 0 ? s_ta[slot_t] : 0.0;
                    const double fa = 1.0 / it->inv_alpha;
                    const double err = fabs(fa - ba) / fmax(1.0, fabs(fa));
                    if (err > c->max_pivot_err || err != err) c->max_pivot_err = err;
                    v.nb_rng[q] = make_int2(c->ph_rng_x, c->ph_rng_y);
                }
            } else if (tn == q) {
                if (flip) {
                    int ev = it->entering_var;
                    double nv = it->entering_new_val;
                    v.xN[q] = nv;
                    f = (uint8_t)((f & NB_FIXED) | (nv == v.var_lo[ev] ? NB_AT_MIN : 0) | (nv == v.var_hi[ev] ? NB_AT_MAX : 0));
                    v.nbflags[q] = f;
                } else {
                    int lv = it->leaving_var;
                    double lnv = it->leaving_new_val;
                    // the pivot element computed two ways (FTRAN side / BTRAN side) measures the drift of W
                    if (q >= v.nb_lo && q < v.nb_hi) {
                        double ba;
                        if (inline_comb) {
                            ba = 0.0;
                            for (int b = 0; b < v.nbands; ++b) ba += v.band_part[(size_t)b * (size_t)v.n + t].x;
                        } else if (pull_inside) {
                            ba = slot_t >= 0 ? s_ta[slot_t] : 0.0;
                        } else {
                            ba = v.alpha_r[q];
                        }
                        double fa = 1.0 / it->inv_alpha;
                        double err = fabs(fa - ba) / fmax(1.0, fabs(fa));
                        if (err > c->max_pivot_err || err != err) c->max_pivot_err = err;
                    }
                    if (v.str_on && !pull_inside) v.alpha_r[q] = 0.0;
                    dd = -it->pivot_obj;
                    v.d[q] = dd;
                    if (use_pse) {
                        gm = it->alpha_sq / (pc * pc);
                        v.gamma[q] = gm;
                    }
                    v.nb_vars[q] = lv;
                    if (v.pk_valid) v.pk_valid[keep >= 0 ? q : t] = 0;  // index t of the pass serves position q: its packed segment is stale now
                    v.nb_rng[q] = make_int2(v.csc_ptr[lv], v.csc_ptr[lv + 1]);
                    v.xN[q] = lnv;
                    f = (uint8_t)((lnv == v.var_lo[lv] ? NB_AT_MIN : 0) | (lnv == v.var_hi[lv] ? NB_AT_MAX : 0));
                    v.nbflags[q] = f;
                }
            } else if (!flip && tn >= v.nb_lo && tn < v.nb_hi) {
                double ar, hp = 0.0;
                if (inline_comb) {  // banded sweep: sum the per-band partials here (band order) instead of a combine launch
                    double s1 = 0.0, s2 = 0.0;
                    for (int b0 = 0; b0 < v.nbands; b0 += 16) {  // sixteen independent loads in flight (all 13 bands of
                        double2 pb[16];                            // config 4 in one round), summed in band order
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            pb[u] = (b0 + u < v.nbands) ? v.band_part[(size_t)(b0 + u) * (size_t)v.n + t] : make_double2(0.0, 0.0);
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            s1 += pb[u].x;
                            s2 += pb[u].y;
                        }
                    }
                    ar = s1;  // (row_coeffs / the PSE helper are not materialised on this path: nothing reads them — the
                    hp = s2;  // host-paced stepping API, which exposes them, keeps the separate combine kernel)
                } else if (pull_inside) {
                    ar = slot_t >= 0 ? s_ta[slot_t] : 0.0;
                    hp = slot_t >= 0 ? s_th[slot_t] : 0.0;
                } else {
                    ar = v.alpha_r[tn];
                    if (use_pse) hp = v.helper[tn];
                }
                if (ar != 0.0) {
                    dd -= it->pivot_obj * ar;
                    v.d[tn] = dd;
                    if (use_pse) {
                        gm += -2.0 * ar * hp / pc + it->alpha_sq * ar * ar / (pc * pc);
                        v.gamma[tn] = gm;
                    }
                    if (v.str_on && !pull_inside) v.alpha_r[tn] = 0.0;  // sparse tableau row: the vector stays zero outside this iteration's touched entries
                }
            }
            if (phase == 0 && tn >= v.nb_lo && tn < v.nb_hi) {
                tc = price_primal_one(dd, gm, f, tn, use_pse);
                tc_d = dd;
            }
        }
        if (cand_better(tc, cand)) {  // (ties keep the lowest position whatever the visiting order)
            cand = tc;
            cand_d = tc_d;
        }
    }
    // every block arrives here only after its own updates; the last arriver closes this iteration
    // (record) and opens the next one with the pricing decision
    KMARK0(c, 13);
    if (!grid_best_p(cand, cand_d, v, n_upd)) return;
    KMARK(c, 14);
    close_and_open(v, c, phase, cand, cand_d, true);
    KMARK(c, 15);
}

// ------------------------------------------------------------------- helpers outside the pivot graph
__global__ void k_set_iter(DevView v, int status, int q, int r, double lnv, int forced) {
    Ctl* c = v.ctl;
    IterState* it = &c->it;
    it->status = status;
    it->q = q;
    it->r = r;
    it->leaving_new_val = lnv;
    it->entering_var = q >= 0 ? v.nb_vars[q] : -1;
    it->leaving_var = r >= 0 ? v.basic_vars[r] : -1;
    it->klist_n = 0;
    it->blist_n = 0;
    c->up.kase = -1;
    c->forced = forced;
    c->halt = 0;
}
__global__ void k_reset_ring(DevView v) {
    Ctl* c = v.ctl;
    c->ring_n = 0;
    c->halt = 0;
    c->forced = 0;
    c->max_pivot_err = 0.0;
    c->hyper_bail = 0;
    c->rv_n = 0;  // (every batch starts from zeroed work vectors: launch_clear_work)
    c->ar_n = 0;  // (... and from an empty list of alpha_r's non-zeros)
    c->ar_keep = -1;
}
// K9: recalc reduced costs (solver.rs:1216-1231): d_c = c_c - a_c . y, then the objective from scratch
__global__ void __launch_bounds__(BLK) k_gather_basic_obj(DevView v) {
    int p = blockIdx.x * BLK + threadIdx.x;
    if (p < v.m) {
        double cb = v.obj_c[v.basic_vars[p]];
        v.alpha_q[p] = cb;
        const int ks = v.kslot_of_pos[p];
        if (ks < 0) v.rv[v.srow_of_pos[p]].y = cb / v.sdiag_of_pos[p];
        else v.rv[v.row_of_kslot[ks]].y = 0.0;  // (the t_K pull adds every entry of a column times rv.y: zero on nucleus rows)
    }
    if (p == 0) {
        v.ctl->it.status = ITER_PIVOT;
        v.ctl->halt = 0;
        v.ctl->up.kase = -1;
    }
}
template <int G>
__global__ void __launch_bounds__(BLK) k_recalc_d(DevView v) {
    int col = (blockIdx.x * BLK + threadIdx.x) / G;
    int gl = threadIdx.x & (G - 1);
    if (col >= v.n) return;
    int var = v.nb_vars[col];
    int end = v.csc_ptr[var + 1];
    double acc = 0.0;
    for (int e = v.csc_ptr[var] + gl; e < end; e += G) acc += v.csc_val[e] * v.rv[v.csc_row[e]].y;
    acc = group_sum<G>(acc);
    if (gl == 0) v.d[col] = v.obj_c[var] - acc;
}
__global__ void __launch_bounds__(BLK) k_recalc_obj(DevView v) {
    double s = 0.0;
    for (int p = blockIdx.x * BLK + threadIdx.x; p < v.m; p += gridDim.x * BLK) s += v.obj_c[v.basic_vars[p]] * v.xB[p];
    for (int j = blockIdx.x * BLK + threadIdx.x; j < v.n; j += gridDim.x * BLK) s += v.obj_c[v.nb_vars[j]] * v.xN[j];
    if (!grid_sum(s, v)) return;
    if (threadIdx.x == 0) v.ctl->it.obj = s;
}
// fix_var on a non-basic variable (solver.rs:393-404): x_B -= diff * alpha_q, obj += diff * d, x_N = val
__global__ void __launch_bounds__(BLK) k_shift_nonbasic(DevView v, int col, double val) {
    int p = blockIdx.x * BLK + threadIdx.x;
    double diff = val - v.xN[col];
    if (p < v.m) {
        double a = v.alpha_q[p];
        if (a != 0.0) v.xB[p] -= diff * a;
    }
    if (p == 0) v.ctl->it.obj += diff * v.d[col];
}
__global__ void k_set_xn(DevView v, int col, double val) { v.xN[col] = val; }
// add_constraint (solver.rs:620-624): gamma[c] += alpha_r[c]^2
__global__ void __launch_bounds__(BLK) k_gamma_add_row(DevView v) {
    int j = blockIdx.x * BLK + threadIdx.x;
    if (j < v.n) {
        double a = v.alpha_r[j];
        v.gamma[j] += a * a;
    }
}
__global__ void k_copy_rho_sq_to_beta(DevView v, int row) { v.beta[row] = v.ctl->it.rho_sq; }

// ------------------------------------------------------------------- from-scratch inversion
// Counterpart of BasisSolver::reset (solver.rs:1286-1303): rebuild the nucleus K from the CSC
// columns of the basic variables and invert it by Gauss-Jordan with partial pivoting.
template <int G>
__global__ void __launch_bounds__(BLK) k_build_nucleus(DevView v, double* Kd, int k) {
    int slot = (blockIdx.x * BLK + threadIdx.x) / G;  // row slot <-> position (column of K)
    int gl = threadIdx.x & (G - 1);
    if (slot >= k) return;
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
        int r = (int)(i / k), cc = (int)(i % k);
        Wv[(size_t)r * ld + cc] = (r == cc) ? 1.0 : 0.0;
    }
}
// one block: pivot search in column j (rows >= j); flag=1 if singular
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
    int cc = blockIdx.x * BLK + threadIdx.x;
    if (cc >= k) return;
    int pr = *piv_row;
    double inv = factors[k];
    double a = Kd[(size_t)pr * ld + cc], b = Kd[(size_t)j * ld + cc];
    double x = Wv[(size_t)pr * ld + cc], y = Wv[(size_t)j * ld + cc];
    if (pr != j) {
        Kd[(size_t)pr * ld + cc] = b;
        Wv[(size_t)pr * ld + cc] = y;
    }
    Kd[(size_t)j * ld + cc] = a * inv;
    Wv[(size_t)j * ld + cc] = x * inv;
}
__global__ void __launch_bounds__(BLK) k_gj_eliminate(double* Kd, double* Wv, int k, int ld, int j, const double* factors) {
    int cc = blockIdx.x * BLK + threadIdx.x;
    int a = blockIdx.y;
    if (cc >= k || a == j) return;
    double f = factors[a];
    if (f == 0.0) return;
    Kd[(size_t)a * ld + cc] -= f * Kd[(size_t)j * ld + cc];
    Wv[(size_t)a * ld + cc] -= f * Wv[(size_t)j * ld + cc];
}



// Exact dual steepest-edge weights from the basis inverse: beta_p = || e_p^T B^-1 ||^2 (the quantity the recurrence of
// solver.rs:1153-1174 maintains).  One block per basic position: the BTRAN head builds the position's list of rows of W
// (no side effects), the block forms rho_K = sum_j a_j W[s_j, :] column by column and sums its squares; a singleton
// position adds 1 / diag^2 for its own row.  Used by the LAZY dual steepest edge: the primal loop never reads beta, so it
// skips tau = B^-1 rho (the second FTRAN of every pivot and its push) and beta is rebuilt here when the dual simplex,
// a checkpoint or the white-box state next needs it.  Requires W0 to be the whole inverse (pending terms folded).
__global__ void __launch_bounds__(BLK) k_exact_beta(DevView v) {
    const int p = blockIdx.x;
    if (p >= v.m) return;
    __shared__ int s_ls[HEAD_CAP];
    __shared__ double s_la[HEAD_CAP];
    __shared__ int s_n;
    __shared__ double s_inv2;
    const int k = v.ctl->k, ld = v.ld;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int sr = v.kslot_of_pos[p];
        if (sr >= 0) {
            if (lane == 0) {
                s_ls[0] = sr;
                s_la[0] = 1.0;
                s_n = 1;
                s_inv2 = 0.0;
            }
        } else {
            const int i_r = v.srow_of_pos[p];
            const double inv = 1.0 / v.sdiag_of_pos[p];
            const int base = v.csr_ptr[i_r], end = v.csr_ptr[i_r + 1];
            int cnt = 0;
            for (int e0 = base; e0 < end; e0 += 64) {
                const int e = e0 + lane;
                const bool valid = e < end;
                int s = -1;
                double a = 0.0;
                if (valid) {
                    const int loc = v.var_loc[v.csr_col[e]];
                    a = v.csr_val[e];
                    if (loc >= 0) s = v.kslot_of_pos[loc];
                }
                const bool isk = valid && s >= 0;
                const unsigned long long mask = __ballot(isk);
                if (isk) {
                    const int off = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                    if (off < HEAD_CAP) {
                        s_ls[off] = s;
                        s_la[off] = -a * inv;
                    }
                }
                cnt += __popcll(mask);
            }
            if (lane == 0) {
                s_n = cnt < HEAD_CAP ? cnt : HEAD_CAP;
                s_inv2 = inv * inv;
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    double sq = 0.0;
    for (int s = threadIdx.x; s < k; s += BLK) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += s_la[j] * v.W[(size_t)s_ls[j] * ld + s];
        sq += acc * acc;
    }
    sq = block_sum(sq);
    if (threadIdx.x == 0) v.beta[p] = sq + s_inv2;
}
void launch_exact_beta(const DevView& dv, hipStream_t st) {
    if (dv.m > 0) hipLaunchKernelGGL(k_exact_beta, dim3(dv.m), dim3(BLK), 0, st, dv);
}

// ------------------------------------------------------------------- device-side matrix maintenance
// Solution::add_constraint appends ONE row (solver.rs:597-613 rebuilds CSR and CSC on the host, O(nnz)).  Here the
// matrix stays on the device: the CSR row is appended in place (capacity-doubling buffers), the CSC is re-laid out by
// one copy kernel (a column's entries move by the number of touched columns before it, found by binary search in the
// sorted new row; the new row index is the largest, so appending keeps every column's rows ascending), and the
// derived copies (band-major copy of the banded sweep, row-block offsets of the blocked F push) are rebuilt by the
// kernels below, which also serve the initial build.  No host pass over the non-zeros, no re-upload.
template <int G>
__global__ void __launch_bounds__(BLK) k_csc_append_row(const int* __restrict__ optr, const int* __restrict__ orow,
                                                        const double* __restrict__ oval, int n_old, int new_row,
                                                        const int* __restrict__ ncols, const double* __restrict__ nvals, int kn,
                                                        int* __restrict__ nptr, int* __restrict__ nrow, double* __restrict__ nval) {
    const int j = (blockIdx.x * BLK + threadIdx.x) / G;  // old column (variable)
    const int gl = threadIdx.x & (G - 1);
    if (j > n_old) return;
    if (j == n_old) {  // the new slack column: one entry (new_row, +1), solver.rs:250
        if (gl == 0) {
            const int b = optr[n_old] + kn;
            nptr[n_old] = b;
            nptr[n_old + 1] = b + 1;
            nrow[b] = new_row;
            nval[b] = 1.0;
        }
        return;
    }
    int lo = 0, hi = kn;  // shift = number of touched columns < j
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ncols[mid] < j) lo = mid + 1;
        else hi = mid;
    }
    const int ob = optr[j], oe = optr[j + 1];
    const int nb = ob + lo;
    for (int e = ob + gl; e < oe; e += G) {
        nrow[nb + (e - ob)] = orow[e];
        nval[nb + (e - ob)] = oval[e];
    }
    if (gl == 0) {
        nptr[j] = nb;
        if (lo < kn && ncols[lo] == j) {
            nrow[nb + (oe - ob)] = new_row;
            nval[nb + (oe - ob)] = nvals[lo];
        }
    }
}
void launch_csc_append_row(const int* optr, const int* orow, const double* oval, int n_old, int new_row, const int* ncols,
                           const double* nvals, int kn, int* nptr, int* nrow, double* nval, hipStream_t st) {
    hipLaunchKernelGGL(k_csc_append_row<16>, dim3(blocks_for((long)(n_old + 1) * 16)), dim3(BLK), 0, st, optr, orow, oval, n_old,
                       new_row, ncols, nvals, kn, nptr, nrow, nval);
}

// exclusive scan of n ints (in place allowed), three phases: 4096-element block scans, scan of the block sums by
// one block, add-back.  `sums` holds ceil(n / 4096) + 1 ints.
constexpr int SCAN_TILE = 4096;
__global__ void __launch_bounds__(1024) k_scan_blocks(const int* in, int* out, long n, int* sums) {
    __shared__ int s_w[16];
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * 4;
    int x[4], t = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        x[u] = base + u < n ? in[base + u] : 0;
        t += x[u];
    }
    int incl = t;  // inclusive scan of the per-thread totals across the block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < 16; ++i) {
            const int y = s_w[i];
            s_w[i] = acc;
            acc += y;
        }
        if (sums) sums[blockIdx.x] = acc;
    }
    __syncthreads();
    int excl = incl - t + s_w[wv];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (base + u < n) out[base + u] = excl;
        excl += x[u];
    }
}
__global__ void __launch_bounds__(1024) k_scan_sums(int* sums, int nb) {  // one block: exclusive scan of up to 2^20 block sums
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int x = i < nb ? sums[i] : 0;
        int incl = x;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wv; ++w2) woff += s_w[w2];
        const int carry = s_carry;
        if (i < nb) sums[i] = carry + woff + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = s_carry;  // grand total
}
__global__ void __launch_bounds__(1024) k_scan_add(int* out, long n, const int* sums) {
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * 4;
    const int off = sums[blockIdx.x];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u < n) out[base + u] += off;
}
void launch_exclusive_scan(const int* in, int* out, long n, int* sums, hipStream_t st) {
    const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, st, in, out, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, sums, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, st, out, n, (const int*)sums);
}

// Band-major copy of A for the banded sweep, built on the device from the CSC (columns hold ascending rows, so the
// entries of (column, band) are a contiguous run): count (even-rounded) per (band, column) -> exclusive scan in
// band-major order -> fill.  bptr has nbands * (N + 1) entries; the slot [b][N] of every band carries a zero count so
// that bptr[b][N] = end of band b's last column.
__global__ void __launch_bounds__(BLK) k_band_count(const int* __restrict__ cptr, const int* __restrict__ crow, int N, int nbands,
                                                     int* __restrict__ cnt) {
    const int var = blockIdx.x * BLK + threadIdx.x;
    if (var > N) return;
    if (var == N) {
        for (int b = 0; b < nbands; ++b) cnt[(size_t)b * (N + 1) + N] = 0;
        return;
    }
    int e = cptr[var];
    const int end = cptr[var + 1];
    for (int b = 0; b < nbands; ++b) {
        const int lim = (b + 1) * BAND_ROWS;
        int c = 0;
        while (e < end && crow[e] < lim) {
            ++e;
            ++c;
        }
        cnt[(size_t)b * (N + 1) + var] = (c + 1) & ~1;
    }
}
__global__ void __launch_bounds__(BLK) k_band_fill(const int* __restrict__ cptr, const int* __restrict__ crow,
                                                    const double* __restrict__ cval, int N, int nbands, const int* __restrict__ bptr,
                                                    unsigned short* __restrict__ brow, double* __restrict__ bval) {
    const int var = blockIdx.x * BLK + threadIdx.x;
    if (var >= N) return;
    int e = cptr[var];
    const int end = cptr[var + 1];
    for (int b = 0; b < nbands; ++b) {
        const int lim = (b + 1) * BAND_ROWS, row0 = b * BAND_ROWS;
        int dst = bptr[(size_t)b * (N + 1) + var];
        const int dend = bptr[(size_t)b * (N + 1) + var + 1];
        while (e < end && crow[e] < lim) {
            brow[dst] = (unsigned short)(crow[e] - row0);
            bval[dst] = cval[e];
            ++e;
            ++dst;
        }
        if (dst < dend) {  // pad entry of an odd-length segment: row 0 of the band, value 0
            brow[dst] = 0;
            bval[dst] = 0.0;
        }
    }
}
void launch_band_count(const int* cptr, const int* crow, int N, int nbands, int* cnt, hipStream_t st) {
    hipLaunchKernelGGL(k_band_count, dim3(blocks_for(N + 1)), dim3(BLK), 0, st, cptr, crow, N, nbands, cnt);
}
void launch_band_fill(const int* cptr, const int* crow, const double* cval, int N, int nbands, const int* bptr, unsigned short* brow,
                      double* bval, hipStream_t st) {
    hipLaunchKernelGGL(k_band_fill, dim3(blocks_for(N)), dim3(BLK), 0, st, cptr, crow, cval, N, nbands, bptr, brow, bval);
}
// Packed non-basic copy (DevView.pk_*): one thread per (band, index of the pass)
__global__ void __launch_bounds__(BLK) k_pack_count(DevView v, int* __restrict__ cnt) {
    const long g = (long)blockIdx.x * BLK + threadIdx.x;
    const int n1 = v.n + 1;
    if (g >= (long)v.nbands * n1) return;
    const int b = (int)(g / n1), i = (int)(g % n1);
    int len = 0;
    if (i < v.n) {
        const int var = v.nb_vars[v.nb_order[i]];
        const int* bp = v.bptr + (size_t)b * (size_t)(v.m + v.n + 1);
        len = bp[var + 1] - bp[var];
    }
    cnt[g] = len;
}
__global__ void __launch_bounds__(BLK) k_pack_fill(DevView v, const int* __restrict__ pkptr, unsigned short* __restrict__ prow,
                                                    double* __restrict__ pval, unsigned char* __restrict__ valid) {
    const long g = (long)blockIdx.x * BLK + threadIdx.x;
    if (g >= (long)v.nbands * v.n) return;
    const int b = (int)(g / v.n), i = (int)(g % v.n);
    const int var = v.nb_vars[v.nb_order[i]];
    const int* bp = v.bptr + (size_t)b * (size_t)(v.m + v.n + 1);
    const int beg = bp[var], end = bp[var + 1];
    int dst = pkptr[(size_t)b * (size_t)(v.n + 1) + i];
    for (int e = beg; e < end; ++e, ++dst) {
        prow[dst] = v.brow[e];
        pval[dst] = v.bval[e];
    }
    if (b == 0) valid[i] = 1;
}
void launch_pack_count(const DevView& dv, int* cnt, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_count, dim3(blocks_for((long)dv.nbands * (dv.n + 1))), dim3(BLK), 0, st, dv, cnt);
}
void launch_pack_fill(const DevView& dv, const int* pkptr, unsigned short* prow, double* pval, unsigned char* valid, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_fill, dim3(blocks_for((long)dv.nbands * dv.n)), dim3(BLK), 0, st, dv, pkptr, prow, pval, valid);
}
// Row-block offsets of every column for the blocked F push: colblk[var][b] = first CSC index of column var whose row is
// >= b * PB_ROWS, colblk[var][rb] = end of the column.
__global__ void __launch_bounds__(BLK) k_build_colblk(const int* __restrict__ cptr, const int* __restrict__ crow, int N, int rb,
                                                       int* __restrict__ colblk) {
    const int var = blockIdx.x * BLK + threadIdx.x;
    if (var >= N) return;
    int e = cptr[var];
    const int end = cptr[var + 1];
    int* dst = colblk + (size_t)var * (rb + 1);
    for (int b = 0; b <= rb; ++b) {
        const long lim = (long)b * PB_ROWS;
        while (e < end && crow[e] < lim) ++e;
        dst[b] = e;
    }
    dst[rb] = end;
}
void launch_build_colblk(const int* cptr, const int* crow, int N, int rb, int* colblk, hipStream_t st) {
    hipLaunchKernelGGL(k_build_colblk, dim3(blocks_for(N)), dim3(BLK), 0, st, cptr, crow, N, rb, colblk);
}

// ------------------------------------------------------------------- K4, sparse form (small nucleus)
// rho = B^-T e_r is non-zero on the nucleus rows (k col slots) and on the leaving singleton's own row only, so while the
// nucleus is small the tableau row alpha_r = rho^T N (solver.rs:685-692) touches just the columns that meet those few rows —
// the reference's loop over the rows of supp(rho), and SURVEY §8(d)'s 16 * sum_{i in supp rho} nnz(A_i) bytes — instead
// of all of A (123 MB on config 4, of which < 6 % lies in the support while k <= 57).  With primal steepest edge the
// helper N^T v (solver.rs:1126-1132) is needed on the columns with alpha_rj != 0 only, i.e. on the same list.
//   k_row_touch: one wave per row of supp(rho): its CSR entries -> non-basic positions, each listed once (epoch stamp);
//   k_row_pull : G lanes per listed column pull alpha_rj (and helper_j) from the CSC in storage order (no float atomics).
// one wave lists the non-basic columns of one row of supp(rho) (each once: epoch stamp of the position)
__device__ __forceinline__ void row_touch_row(const DevView& v, Ctl* c, int row, int lane, int ep) {
    const int end = v.csr_ptr[row + 1];
    for (int e0 = v.csr_ptr[row]; e0 < end; e0 += 64) {
        const int e = e0 + lane;
        int j = -1;
        if (e < end) {
            const int loc = v.var_loc[v.csr_col[e]];
            if (loc < 0) {
                j = -1 - loc;
                if (j < v.nb_lo || j >= v.nb_hi || atomicExch(&v.hy_stamp_n[j], ep) == ep) j = -1;
            }
        }
        const unsigned long long mask = __ballot(j >= 0);  // one counter update per wave and trip
        if (mask) {
            int base = 0;
            if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&c->str_n, __popcll(mask));
            base = __shfl(base, __ffsll((long long)mask) - 1, 64);
            if (j >= 0) v.str_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = j;
        }
    }
}
template <bool OWN_RK>
__device__ __forceinline__ void row_touch_body(const DevView& v, Ctl* c, int block, double own_rk) {
    const int k = c->k;
    const int w = (int)((block * BLK + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (w > k) return;
    int row;
    if (w < k) {
        if ((OWN_RK ? own_rk : v.rK[w]) == 0.0) return;  // (k_small_basis: the wave formed its own entry of rho_K)
        row = v.row_of_kslot[w];
    } else {  // the leaving singleton's own row (rho there is 1 / its diagonal entry)
        const int r = c->it.r;
        if (v.kslot_of_pos[r] >= 0) return;
        row = v.srow_of_pos[r];
    }
    row_touch_row(v, c, row, lane, c->hyper_epoch + 1);  // (the epoch is advanced by the update kernel's finaliser)
}
__global__ void __launch_bounds__(BLK) k_row_touch(DevView v) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 10);
    row_touch_body(v, c, (int)blockIdx.x);
}
// ------------------------------------------------------------------- small nucleus: BTRAN + pass over W + v tail + touch, one launch
// While the nucleus inverse still has its first capacity (256 slots) the lazy primal iteration's three launches between the
// ratio test and the tableau row — k_btran (rho_K, ||rho||^2 | t_K), k_fused_w<8> (v partials + eta update of W),
// k_post_fused (v_K reduce + scatter | touched-column list) — move a few kilobytes each and cost 4.9 + 6.2 + 5.7 us of
// launch ramps, boundaries and dependent-load chains (profiles/r04_early_kernel_stats.csv).  Here they are ONE launch:
//   block 0            rho_K from the listed rows of W (as k_btran), then — once the t_K blocks have arrived — the whole pass
//                      over W from LDS-staged vectors, the v_K reduction and its scatter by row;
//   blocks 1..n_rhs    t_K = alpha_K - F^T y_S, G lanes per slot (as k_btran's second part), write-through, then a ticket;
//   the rest           the touched-column list of the sparse tableau row; each wave forms its own rho_K entry, so these
//                      blocks wait for nobody.
// The only in-kernel wait is block 0's for the t_K tickets, BEFORE anything of the iteration has been applied to W: if it
// times out (grid not co-resident) the batch ends with ITER_STALL like a fused ratio test and the engine goes back to the
// three launches for good (Geom.ratio_two).  Every sum is formed in the order of the kernels it replaces — the stripes of
// eight rows, then k_post_fused's eight groups — so the results are bit-identical to theirs (tests/test_small_basis.py).
constexpr int SB_CAP = 256;  // capacity this kernel serves (one workgroup walks all of W: 512 KB at most)
template <int G>
__global__ void __launch_bounds__(BLK) k_small_basis(DevView v, int n_rhs) {  // n_rhs = 0: t_K was formed by blocks riding in the ratio launch
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    const int k = c->k, ld = v.ld;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x > n_rhs) {
        // rho_K is being formed by block 0 of this launch: each wave forms its own entry (the same sum in the same order,
        // hence the same bits — only "is it zero" is needed) from the rows of W block 0 is going to update, so block 0 waits
        // for this block's ticket before it writes W
        const int tb = (int)blockIdx.x - 1 - n_rhs;
        if (tb * (BLK / 64) > k) return;  // (no wave of this block has a row: block 0 does not count it)
        const int w = (tb * BLK + tid) >> 6;
        double acc = 0.0;
        if (w < k) {
            const int n = c->it.blist_n;
            for (int j = 0; j < n; ++j) acc += v.blist_a[j] * v.W[(size_t)v.blist_s[j] * ld + w];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&v.ticket[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        row_touch_body<true>(v, c, tb, acc);
        return;
    }
    if (blockIdx.x > 0) {  // t_K (solver.rs:1114), as k_btran
        const int slot = (((int)blockIdx.x - 1) * BLK + tid) / G;
        const int gl = tid & (G - 1);
        if (((int)blockIdx.x - 1) * (BLK / G) >= k) return;  // (whole block beyond the nucleus: block 0 does not count it)
        if (slot < k) {
            const int p = v.pos_of_kslot[slot];
            const int var = v.basic_vars[p];
            const int end = v.csc_ptr[var + 1];
            double acc = 0.0;
            for (int e = v.csc_ptr[var] + gl; e < end; e += G) acc += v.csc_val[e] * v.rv[v.csc_row[e]].y;
            acc = group_sum<G>(acc);
            if (gl == 0) {
                st_agent(&v.tK[slot], v.alpha_q[p] - acc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&v.ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    KMARK0(c, 7);
    __shared__ double s_part[32][128];  // per-stripe v partials of one half of the columns
    __shared__ double s_rk[SB_CAP], s_tk[SB_CAP], s_u[SB_CAP];
    __shared__ int s_state;
    // (1) rho_K = sum_j blist_a[j] W[blist_s[j], :], rho by row, ||rho||^2
    {
        const int n = c->it.blist_n;
        double sq = 0.0;
        if (tid < k) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += v.blist_a[j] * v.W[(size_t)v.blist_s[j] * ld + tid];
            v.rK[tid] = acc;
            v.rv[v.row_of_kslot[tid]].x = acc;
            s_rk[tid] = acc;
            sq += acc * acc;
        }
        sq = block_sum(sq);
        {   // (the detour of k_btran's one-block grid reduction: a second tree over (sq, 0, 0, ...) — the same bits)
            double y = tid == 0 ? sq : 0.0;
            sq = block_sum(y);
        }
        if (tid == 0) {
            const int r = c->it.r;
            if (v.kslot_of_pos[r] < 0) {
                const double inv = 1.0 / v.sdiag_of_pos[r];
                sq += inv * inv;
            }
            c->it.rho_sq = sq;
        }
    }
    // (2) the eta column by slot while the t_K blocks finish, then wait for their tickets
    const double inv_alpha = c->it.inv_alpha;
    const int rslot = c->up.sr;
    if (tid < k) s_u[tid] = (v.aK[tid] - (tid == rslot ? 1.0 : 0.0)) * inv_alpha;
    const int row_of_mine = tid < k ? v.row_of_kslot[tid] : 0;  // (128 columns per half: thread i serves column cb + i below)
    const int row_of_mine2 = tid + 128 < k && tid < 128 ? v.row_of_kslot[tid + 128] : 0;
    if (tid == 0) {
        const unsigned want = n_rhs > 0 ? (unsigned)((k + BLK / G - 1) / (BLK / G)) : 0u;  // t_K blocks with a slot below k
        const unsigned want_touch = (unsigned)(k / (BLK / 64) + 1);          // touch blocks with a wave at or below k
        int state = 0;
        long long spins = 0;
        while (__hip_atomic_load(&v.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want ||
               __hip_atomic_load(&v.ticket[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want_touch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > c->ratio_spin_limit) {
                state = 1;
                break;
            }
        }
        if (!state) {
            __hip_atomic_store(&v.ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&v.ticket[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_state = state;
    }
    __syncthreads();
    if (s_state) {
        if (tid == 0 && atomicExch(&c->halt, 1) == 0) {
            c->it.obj = c->sb_obj0;  // (the ratio test of this iteration has already added its step: the iteration is re-run from the top)
            c->it.status = ITER_STALL;
            push_rec(c, 0);
        }
        return;
    }
    KMARK0(c, 8);
    if (tid < k) s_tk[tid] = ld_agent(&v.tK[tid]);
    __syncthreads();
    // (3) v partials per stripe of eight rows + eta update of W (k_fused_w<8>), reduced in k_post_fused's order
    const int nstripes = (k + 7) / 8;
    const int cp = tid & 63, wv = tid >> 6;
    for (int cb = 0; cb < k; cb += 128) {
        const int c0 = cb + 2 * cp;
        const bool pair = c0 + 1 < k, one = c0 < k;
        const double rk0 = one ? s_rk[c0] : 0.0, rk1 = pair ? s_rk[c0 + 1] : 0.0;
        // (two stripes of the wave per trip: 32 independent loads in flight per lane — the pass is one workgroup's chain of
        // load -> use -> store round trips, not bandwidth)
        for (int st0 = wv; st0 < nstripes; st0 += 2 * (BLK / 64)) {
            double w0[16], w1[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                const int row = (st0 + (a >> 3) * (BLK / 64)) * 8 + (a & 7);
                w0[a] = 0.0;
                w1[a] = 0.0;
                if (row < k && st0 + (a >> 3) * (BLK / 64) < nstripes) {
                    const double* wp = v.W + (size_t)row * ld;
                    if (pair) {
                        const double2 t = *reinterpret_cast<const double2*>(wp + c0);
                        w0[a] = t.x;
                        w1[a] = t.y;
                    } else if (one) {
                        w0[a] = wp[c0];
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int stripe = st0 + h * (BLK / 64);
                if (stripe >= nstripes) break;
                double va0 = 0.0, va1 = 0.0;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int row = stripe * 8 + a;
                    if (row >= k) continue;
                    const double t = s_tk[row], u = s_u[row];
                    va0 += w0[h * 8 + a] * t;
                    va1 += w1[h * 8 + a] * t;
                    double* wp = v.W + (size_t)row * ld;
                    const double n0 = w0[h * 8 + a] - u * rk0, n1 = w1[h * 8 + a] - u * rk1;
                    if (pair) *reinterpret_cast<double2*>(wp + c0) = make_double2(n0, n1);
                    else if (one) wp[c0] = n0;
                }
                s_part[stripe][2 * cp] = va0;
                s_part[stripe][2 * cp + 1] = va1;
            }
        }
        __syncthreads();
        if (tid < 128 && cb + tid < k) {
            double sv = 0.0;
#pragma unroll 1
            for (int grp = 0; grp < 8; ++grp) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int t = grp;
                for (; t + 24 < nstripes; t += 32) {
                    s0 += s_part[t][tid];
                    s1 += s_part[t + 8][tid];
                    s2 += s_part[t + 16][tid];
                    s3 += s_part[t + 24][tid];
                }
                for (; t < nstripes; t += 8) s0 += s_part[t][tid];
                const double g = (s0 + s1) + (s2 + s3);
                sv = grp == 0 ? g : sv + g;
            }
            v.vK[cb + tid] = sv;
            v.rv[cb == 0 ? row_of_mine : row_of_mine2].y = sv;
        }
        __syncthreads();
    }
    if (tid == 0) c->sb_count += 1;
}
template <int G, int MODE>  // MODE 0: alpha_r, 1: alpha_r + helper, 2: helper
__global__ void __launch_bounds__(BLK) k_row_pull(DevView v, int n_pull) {
    Ctl* c = v.ctl;
    if (c->halt || c->it.status != ITER_PIVOT) return;
    KMARK0(c, 11);
    if ((int)blockIdx.x >= n_pull) {
        struct_update_body(v, c, ((int)blockIdx.x - n_pull) * BLK + threadIdx.x);
        return;
    }
    const int n_t = c->str_n;
    const int gl = threadIdx.x & (G - 1);
    for (int idx = (int)(blockIdx.x * BLK + threadIdx.x) / G; idx < n_t; idx += n_pull * (BLK / G)) {
        const int col = v.str_list[idx];
        const int2 rg = v.nb_rng[col];
        double a1 = 0.0, a2 = 0.0;
        for (int e = rg.x + gl; e < rg.y; e += G) {
            const double a = v.csc_val[e];
            const double2 t = v.rv[v.csc_row[e]];
            if (MODE != 2) a1 += a * t.x;
            if (MODE != 0) a2 += a * t.y;
        }
        if (MODE != 2) a1 = group_sum<G>(a1);
        if (MODE != 0) a2 = group_sum<G>(a2);
        if (gl == 0) {
            if (MODE != 2) v.alpha_r[col] = a1;
            if (MODE != 0) v.helper[col] = a2;
        }
    }
}

#include "fpull.inc"  // large-nucleus primal iteration: the F product of the FTRAN pulled inside the ratio test's launch
#include "hyper.inc"  // the hypersparse single-workgroup iteration (uses the stage helpers above)
#include "primal_head.inc"  // small-nucleus primal iteration: FTRAN + Harris test + BTRAN + inverse update + touched columns in ONE workgroup (uses hyper.inc's DPP reductions)
#include "factor.inc"  // the compact factor of the basis: peel, level-scheduled solves, additive eta terms
#include "factor_sb.inc"  // ... sparse factor (LU with fill, rounds of independent pivots) of the bump the peel leaves
#include "inverse.inc"  // blocked in-place inversion of a dense-filling nucleus (the refactorisation of the explicit inverse)

// ===================================================================================== launchers
#define LANES_SWITCH(L, STMT4, STMT16, STMT32) \
    do {                                        \
        if ((L) <= 4) { STMT4; }                \
        else if ((L) <= 16) { STMT16; }         \
        else { STMT32; }                        \
    } while (0)

void launch_hyper_dual(const DevView& dv, int use_dse, int max_iters, long heavy, hipStream_t st) {
    const int prof_on = std::getenv("MLP_HYPER_PROF") != nullptr ? 1 : 0;  // per-stage wall-clock marks (each costs a timer read)
    static bool attr_set = false;
    if (!attr_set) {  // 136 KB of LDS lists (more than the default 64 KB per workgroup)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hyper_dual), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HY_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_hyper_dual, dim3(1), dim3(HB), HY_LDS_BYTES, st, dv, use_dse, max_iters, heavy > 0 ? heavy : HY_HEAVY, prof_on);
}
void launch_clear_work(const DevView& hv, hipStream_t st) {
    // alpha_q | tau | rv | hS are carved from one allocation (engine): a single memset
    (void)hipMemsetAsync(hv.alpha_q, 0, sizeof(double) * 5 * (size_t)hv.m, st);
}
void launch_price_primal(const DevView& dv, const Geom& g, int use_pse, hipStream_t st) {
    hipLaunchKernelGGL(k_price_primal, dim3(grid_for(dv.nb_hi - dv.nb_lo)), dim3(BLK), 0, st, dv, use_pse);
}
void launch_price_dual(const DevView& dv, const Geom& g, int use_dse, hipStream_t st) {
    hipLaunchKernelGGL(k_price_dual, dim3(grid_for(g.m)), dim3(BLK), 0, st, dv, use_dse);
}
void launch_ftran_prep(const DevView& dv, int derive_primal, hipStream_t st) {
    hipLaunchKernelGGL(k_ftran_prep, dim3(1), dim3(64), 0, st, dv, derive_primal);
}
void launch_ftran_fused(const DevView& dv, const Geom& g, int derive_primal, hipStream_t st) {  // head + gather in one launch
    LANES_SWITCH(g.lanes,
                 hipLaunchKernelGGL(k_ftran_fused<4>, dim3(blocks_for((long)g.cap * 4)), dim3(BLK), 0, st, dv, derive_primal),
                 hipLaunchKernelGGL(k_ftran_fused<16>, dim3(blocks_for((long)g.cap * 16)), dim3(BLK), 0, st, dv, derive_primal),
                 hipLaunchKernelGGL(k_ftran_fused<64>, dim3(blocks_for((long)g.cap * 64)), dim3(BLK), 0, st, dv, derive_primal));
    if (dv.pb_on) launch_blocked_push(dv, 0, st);
    else if (dv.det_pull) launch_pull_F(dv, g, 0, st);
}
void launch_btran_fused(const DevView& dv, const Geom& g, int with_rhs, int derive_dual, hipStream_t st) {  // head + BTRAN in one launch
    int n_gather = blocks_for(g.cap);
    if (n_gather > 512) n_gather = 512;
#define BTRANF(G)                                                                                                  \
    do {                                                                                                           \
        int n_rhs = with_rhs ? blocks_for((long)g.cap * G) : 0;                                                    \
        hipLaunchKernelGGL(k_btran_fused<G>, dim3(n_gather + n_rhs), dim3(BLK), 0, st, dv, n_gather, derive_dual); \
    } while (0)
    LANES_SWITCH(g.lanes, BTRANF(4), BTRANF(16), BTRANF(64));
#undef BTRANF
}
bool ftran_head_rides_gather(const DevView& dv, const Geom& g) {  // delayed-update mode: the FTRAN head inside the gather (k_ftran_gather_lrh)
    return dv.lrJ > 0 && dv.pb_on && dv.world <= 1 && !g.fac && !dv.det_pull && dv.rowinfo != nullptr;
}
void launch_ftran_gather_lrh(const DevView& dv, const Geom& g, hipStream_t st, int ys, int fpk) {
    hipLaunchKernelGGL(k_ftran_gather_lrh, dim3(blocks_for((long)g.cap * 4)), dim3(BLK), 0, st, dv, fpk);
    if (!fpk) launch_blocked_push(dv, 0, st, ys);  // (fpk: the product is pulled inside the ratio test's launch, fpull.inc)
}
void launch_ftran_gather(const DevView& dv, const Geom& g, hipStream_t st, int ys) {
    // (delayed-update mode with the blocked push: nothing after the gather needs the lanes of a slot — four lanes per slot)
    LANES_SWITCH((dv.lrJ && dv.pb_on) ? 4 : g.lanes,
                 hipLaunchKernelGGL(k_ftran_gather<4>, dim3(blocks_for((long)g.cap * 4)), dim3(BLK), 0, st, dv),
                 hipLaunchKernelGGL(k_ftran_gather<16>, dim3(blocks_for((long)g.cap * 16)), dim3(BLK), 0, st, dv),
                 hipLaunchKernelGGL(k_ftran_gather<64>, dim3(blocks_for((long)g.cap * 64)), dim3(BLK), 0, st, dv));
    if (dv.pb_on) launch_blocked_push(dv, 0, st, ys);
    else if (dv.det_pull) launch_pull_F(dv, g, 0, st);
}
// Blocks of `fn` (BLK threads, no dynamic LDS) that the CURRENT device holds at once, halved as a margin for kernels of
// other queues sharing the CUs; cached per device (mlp_set_device may move a process to another GPU or partition).
static int coresident_half(const void* fn, int slot) {
    static int cache[3][64];
    static bool known[3][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!known[slot][dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BLK, 0) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            cache[slot][dev] = per_cu * cus / 2;
        else
            cache[slot][dev] = 0;  // unknown: always take the two-kernel path
        known[slot][dev] = true;
    }
    return cache[slot][dev];
}
static bool ratio_one_enabled() {  // MLP_RATIO_ONE=0: the grid forms at any size (tests of their in-kernel wait on small instances)
    const char* e = std::getenv("MLP_RATIO_ONE");  // (read per launch: tests switch it inside one process)
    return !(e && e[0] == '0');
}
// t_K rides in the ratio launch (k_ratio_primal_fused, n_ratio > 0): the one-launch grid form of the test is the one that runs,
// the blocked F push is on (its combine leaves y_S by row) and the delayed-update mode is on (large nucleus)
bool tk_rides_ratio(const DevView& dv, const Geom& g) {
    const char* e = std::getenv("MLP_TK_RIDE");
    if (e && e[0] == '0') return false;
    if (dv.world > 1 || !dv.pb_on || !dv.lrJ || g.ratio_two || g.cap <= 0) return false;
    if (g.m <= 16384 && ratio_one_enabled()) return false;
    const int nb = grid_for(g.m);
    const int max_coresident = coresident_half(reinterpret_cast<const void*>(k_ratio_primal_fused), 0);
    return nb <= max_coresident && (long)nb * BLK * 4 >= (long)g.m;
}
// ... and for a small nucleus (k_small_basis): the grid form of the test runs (in its sparse form block 0 alone works), t_K with y_S on the fly
bool tk_rides_ratio_small(const DevView& dv, const Geom& g) {
    const char* e = std::getenv("MLP_TK_RIDE");
    if (e && e[0] == '0') return false;
    if (dv.world > 1 || g.ratio_two || g.cap <= 0 || !dv.rowinfo) return false;
    if (g.m <= 16384 && ratio_one_enabled()) return false;
    const int nb = grid_for(g.m);
    const int max_coresident = coresident_half(reinterpret_cast<const void*>(k_ratio_primal_fused), 0);
    return nb <= max_coresident && (long)nb * BLK * 4 >= (long)g.m;
}
// Pulled F product (fpull.inc): two launches, neither with an in-kernel wait.  The first reduces over one block per 32 items (m rows + cap
// slots): the partials travel through red_key / red_key2, which Engine::ensure_red sizes for it.
bool fpull_supported(const DevView& dv, const Geom& g) {
    return dv.fpk_on && dv.fpk_cnt && dv.rowinfo && dv.lrJ > 0;
}
void launch_fpull_ratio(const DevView& dv, const Geom& g, hipStream_t st, hipEvent_t ftran_done) {
    const long items = (long)g.m + (long)g.cap;
    const int n1 = blocks_for(items * FP_G);
    hipLaunchKernelGGL(k_fpull_p1, dim3(n1), dim3(BLK), 0, st, dv);
    if (ftran_done) (void)hipEventRecord(ftran_done, st);  // (sampled iteration: alpha_q is complete here — the FTRAN bracket closes)
    const int nb = grid_for(g.m);
    const int lanes = g.lanes <= 4 ? 4 : (g.lanes <= 16 ? 16 : 64);
    // (grid: nb ratio blocks | the t_K blocks | one block that appends the entering column to the packed copy)
    LANES_SWITCH(lanes, hipLaunchKernelGGL(k_fpull_p2<4>, dim3(nb + blocks_for((long)g.cap * 4) + 1), dim3(BLK), 0, st, dv, nb, n1),
                 hipLaunchKernelGGL(k_fpull_p2<16>, dim3(nb + blocks_for((long)g.cap * 16) + 1), dim3(BLK), 0, st, dv, nb, n1),
                 hipLaunchKernelGGL(k_fpull_p2<64>, dim3(nb + blocks_for((long)g.cap * 64) + 1), dim3(BLK), 0, st, dv, nb, n1));
}
void launch_ratio_primal(const DevView& dv, const Geom& g, int use_pse, hipStream_t st, int tk_ride) {
    if (tk_ride) {  // (the caller asked tk_rides_ratio / tk_rides_ratio_small first); 2: small nucleus, y_S on the fly
        const int nb = grid_for(g.m);
        const int lanes = g.lanes <= 4 ? 4 : (g.lanes <= 16 ? 16 : 64);
        hipLaunchKernelGGL(k_ratio_primal_fused, dim3(nb + blocks_for((long)g.cap * lanes)), dim3(BLK), 0, st, dv, tk_ride == 2 ? use_pse : 2, nb, lanes,
                           tk_ride == 2 ? 1 : 0);
        return;
    }
    if (dv.world <= 1 && g.m <= 16384 && ratio_one_enabled()) {  // small model: one block, no grid-wide reduction (RATIO_ONE_MAX)
        hipLaunchKernelGGL(k_ratio_primal_one, dim3(1), dim3(BLK), 0, st, dv, use_pse);
        return;
    }
    const int nb = grid_for(g.m);
    // The fused kernel's blocks wait inside the launch for its last-arriving block, which is only safe while the
    // WHOLE grid is co-resident: bound the grid by what this device (or partition: CPX mode, CU mask) can hold at
    // once, per the occupancy calculator, with a 2x margin for kernels of other queues sharing the CUs.  The engine
    // selects the two-launch form (Geom.ratio_two) when MLP_RATIO_TWO_KERNELS is set, when the ranks of a sharded solve
    // share one device, and for good after a wait has ever timed out (ITER_STALL).
    const int max_coresident = g.ratio_two ? 0 : coresident_half(reinterpret_cast<const void*>(k_ratio_primal_fused), 0);
    if (nb <= max_coresident && (long)nb * BLK * 4 >= (long)g.m) {  // every element fits the fused kernel's registers
        hipLaunchKernelGGL(k_ratio_primal_fused, dim3(nb), dim3(BLK), 0, st, dv, use_pse, 0, 0, 0);  // both passes + BTRAN head + plan
        return;
    }
    hipLaunchKernelGGL(k_ratio_primal_p1, dim3(nb), dim3(BLK), 0, st, dv, use_pse);
    hipLaunchKernelGGL(k_ratio_primal_p2, dim3(nb), dim3(BLK), 0, st, dv);  // + BTRAN head + plan
}
void launch_post_ftran(const DevView& dv, const Geom& g, int use_pse, hipStream_t st) {
    hipLaunchKernelGGL(k_post_ftran, dim3(use_pse ? grid_for(g.m) : 1), dim3(BLK), 0, st, dv, use_pse);
}
void launch_btran_prep(const DevView& dv, int derive_dual, int plan_after, hipStream_t st) {
    hipLaunchKernelGGL(k_btran_prep, dim3(1), dim3(64), 0, st, dv, derive_dual, plan_after);
}
void launch_btran(const DevView& dv, const Geom& g, int with_rhs, hipStream_t st, int after_fold) {
    int n_gather = blocks_for(dv.lrJ ? (long)g.cap * 4 : (long)g.cap);  // delayed-update mode: 4 lanes per slot
    if (n_gather > (dv.lrJ ? 1024 : 512)) n_gather = dv.lrJ ? 1024 : 512;  // (the reduction buffers hold >= 1024 partials)
#define BTRAN(G)                                                                                           \
    do {                                                                                                   \
        int n_rhs = with_rhs ? blocks_for((long)g.cap * G) : 0;                                            \
        hipLaunchKernelGGL(k_btran<G>, dim3(n_gather + n_rhs), dim3(BLK), 0, st, dv, n_gather, after_fold); \
    } while (0)
    LANES_SWITCH(g.lanes, BTRAN(4), BTRAN(16), BTRAN(64));
#undef BTRAN
}
void launch_btran_rhs(const DevView& dv, const Geom& g, hipStream_t st) {
    LANES_SWITCH(g.lanes,
                 hipLaunchKernelGGL(k_btran_rhs<4>, dim3(blocks_for((long)g.cap * 4)), dim3(BLK), 0, st, dv),
                 hipLaunchKernelGGL(k_btran_rhs<16>, dim3(blocks_for((long)g.cap * 16)), dim3(BLK), 0, st, dv),
                 hipLaunchKernelGGL(k_btran_rhs<64>, dim3(blocks_for((long)g.cap * 64)), dim3(BLK), 0, st, dv));
}
static void launch_sweep_banded(const DevView& dv, const Geom& g, int mode, int with_struct, int inline_combine, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = sizeof(double2) * (size_t)BAND_ROWS;
    if (!attr_set) {  // more than the default 64 KB of LDS per workgroup
        // (a failure here makes the launches below fail, which Engine::pull_ctl reports through hipGetLastError)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_band<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_band<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_band<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_band<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_band<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sweep_band<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // column order of the pass
    // Measured (round 2): variable order does NOT pay — late window 77.5 us against 73.4 us in position order, early window
    // 112.4 against 103.9 us per pivot: the 16-byte partial stores scattered by position cost what the sequential read
    // saves. Position order it is.
    const bool vord = false;  // (variable order of the pass: measured slower — 112.4 against 103.9 us per pivot — and removed as a choice)
    // One workgroup per CU (LDS-bound): all blocks of the launch must fit the 256 CUs at once, or a second,
    // almost empty round of workgroups doubles the kernel time (260 workgroups: 43 us, 247: 36 us).  In the
    // primal iteration the per-band partials are summed by k_update_pivot itself (inline_combine) and the
    // partition change rides in the tail blocks of this launch; otherwise k_band_combine does both.
    const int struct_blocks = (inline_combine && with_struct) ? (g.cap + BAND_THREADS - 1) / BAND_THREADS : 0;
    int chunks = (256 - struct_blocks) / dv.nbands;
    if (chunks < 1) chunks = 1;
    const dim3 gr(dv.nbands * chunks + struct_blocks), b(BAND_THREADS);
    if (vord) {
        if (mode == 0) LAUNCH_T(1, (k_sweep_band<0, true>), gr, b, lds, st, dv, chunks);
        else if (mode == 1) LAUNCH_T(1, (k_sweep_band<1, true>), gr, b, lds, st, dv, chunks);
        else LAUNCH_T(1, (k_sweep_band<2, true>), gr, b, lds, st, dv, chunks);
    } else {
        if (mode == 0) LAUNCH_T(1, (k_sweep_band<0, false>), gr, b, lds, st, dv, chunks);
        else if (mode == 1) LAUNCH_T(1, (k_sweep_band<1, false>), gr, b, lds, st, dv, chunks);
        else LAUNCH_T(1, (k_sweep_band<2, false>), gr, b, lds, st, dv, chunks);
    }
    if (inline_combine) return;
    const int nc = blocks_for(dv.nb_hi - dv.nb_lo);
    const dim3 gc(nc + (with_struct ? blocks_for(g.cap) : 0));
    if (mode == 0) hipLaunchKernelGGL(k_band_combine<0>, gc, dim3(BLK), 0, st, dv, nc);
    else if (mode == 1) hipLaunchKernelGGL(k_band_combine<1>, gc, dim3(BLK), 0, st, dv, nc);
    else hipLaunchKernelGGL(k_band_combine<2>, gc, dim3(BLK), 0, st, dv, nc);
}
bool launch_sweep(const DevView& dv, const Geom& g, int mode, int with_struct, hipStream_t st, int inline_combine, int list) {
    if (dv.banded) {
        launch_sweep_banded(dv, g, mode, with_struct, inline_combine, st);
        return false;
    }
#define SWEEP(G, U)                                                                                               \
    do {                                                                                                          \
        int n_sweep = blocks_for((long)(dv.nb_hi - dv.nb_lo) * G);                                                \
        dim3 gr(n_sweep + (with_struct ? blocks_for(g.cap) : 0)), b(BLK);                                         \
        if (mode == 0) LAUNCH_T(1, (k_sweep<G, U, 0>), gr, b, 0, st, dv, n_sweep, list);                         \
        else if (mode == 1) LAUNCH_T(1, (k_sweep<G, U, 1>), gr, b, 0, st, dv, n_sweep, 0);                    \
        else LAUNCH_T(1, (k_sweep<G, U, 2>), gr, b, 0, st, dv, n_sweep, 0);                                   \
    } while (0)
    if (g.sweep_one) { SWEEP(1, 4); }
    else if (g.sweep_variant == 1) { LANES_SWITCH(g.lanes, SWEEP(4, 4), SWEEP(16, 4), SWEEP(32, 4)); }
    else if (g.sweep_variant == 2) { LANES_SWITCH(g.lanes, SWEEP(4, 8), SWEEP(8, 8), SWEEP(32, 8)); }
    else { LANES_SWITCH(g.lanes, SWEEP(4, 4), SWEEP(16, 4), SWEEP(16, 8)); }
#undef SWEEP
    return mode == 0 && list && dv.ar_list != nullptr;  // (k_sweep MODE 0 listed the non-zeros of alpha_r)
}
void launch_row_sparse(const DevView& dv, const Geom& g, int mode, int with_struct, int touch, hipStream_t st) {
    if (touch) hipLaunchKernelGGL(k_row_touch, dim3(blocks_for((long)(g.cap + 1) * 64)), dim3(BLK), 0, st, dv);
    const int n_pull = 256;
    const dim3 gr(n_pull + (with_struct ? blocks_for(g.cap) : 0));
#define ROWPULL(G)                                                                                   \
    do {                                                                                             \
        if (mode == 0) hipLaunchKernelGGL((k_row_pull<G, 0>), gr, dim3(BLK), 0, st, dv, n_pull);      \
        else if (mode == 1) hipLaunchKernelGGL((k_row_pull<G, 1>), gr, dim3(BLK), 0, st, dv, n_pull); \
        else hipLaunchKernelGGL((k_row_pull<G, 2>), gr, dim3(BLK), 0, st, dv, n_pull);                \
    } while (0)
    LANES_SWITCH(g.lanes, ROWPULL(4), ROWPULL(16), ROWPULL(32));
#undef ROWPULL
}
bool small_basis_supported(const DevView& dv, const Geom& g) {
    const char* sb = std::getenv("MLP_SMALL_BASIS");  // (read per call, i.e. per captured graph: tests toggle it inside one process)
    const bool off = sb && sb[0] == '0';
    return !off && g.sb && g.cap > 0 && g.cap <= SB_CAP && !g.big && !dv.lrJ && g.str && !g.ratio_two && !g.fac && dv.world <= 1;
}
void launch_small_basis(const DevView& dv, const Geom& g, hipStream_t st, int tk_inside) {
    const int n_touch = blocks_for((long)(g.cap + 1) * 64);
#define SMALLB(G)                                                                                                  \
    do {                                                                                                           \
        const int n_rhs = tk_inside ? blocks_for((long)g.cap * G) : 0;                                             \
        LAUNCH_T(2, k_small_basis<G>, dim3(1 + n_rhs + n_touch), dim3(BLK), 0, st, dv, n_rhs);                      \
    } while (0)
    LANES_SWITCH(g.lanes, SMALLB(4), SMALLB(16), SMALLB(64));
#undef SMALLB
}
bool primal_head_supported(const DevView& dv, const Geom& g) {
    const char* e = std::getenv("MLP_PRIMAL_HEAD");  // (read per call, i.e. per captured graph: tests toggle it inside one process)
    const bool off = e && e[0] == '0';
    return !off && g.ph && g.str && g.cap > 0 && !g.big && !g.fac && !dv.lrJ && dv.world <= 1 && !dv.pb_on && !dv.det_pull && dv.rowinfo &&
           dv.hy_stamp_p && dv.str_list;
}
int primal_head_kmax(int longest_column) {
    if (longest_column <= 0 || longest_column > PH_COL_MAX) return 0;  // (every nucleus column sits in the registers of 16 lanes)
    const int by_list = PH_AQ_CAP / longest_column - 1;  // supp(alpha_q) <= (k + 1) columns' worth of rows
    return by_list < PH_KMAX ? by_list : PH_KMAX;
}
void launch_primal_head(const DevView& dv, const Geom& g, hipStream_t st) {
    // (the touched columns as a LIST only when k_row_pull follows; the update kernel that pulls them itself reads the stamps)
    LAUNCH_T(2, k_primal_head, dim3(1), dim3(PH_T), 0, st, dv, update_pulls_inside(dv, g) ? 0 : 1, head_applies(dv, g) ? 1 : 0);
}
void launch_init_nb_rng(const DevView& dv, const Geom& g, hipStream_t st) {
    hipLaunchKernelGGL(k_init_nb_rng, dim3(blocks_for(g.n)), dim3(BLK), 0, st, dv);
}
void launch_ratio_dual(const DevView& dv, const Geom& g, hipStream_t st, int list_ok) {
    if (dv.world <= 1 && g.n <= RATIO_ONE_MAX && ratio_one_enabled()) {  // small model: one block, no grid-wide reduction
        hipLaunchKernelGGL(k_ratio_dual_one, dim3(1), dim3(BLK), 0, st, dv, list_ok);
        return;
    }
    const int span = dv.nb_hi - dv.nb_lo;
    const int pt = span > 131072 ? 16 : 4;
    const int nb = grid_for(span, pt);
    const int max_coresident = g.ratio_two ? 0 : coresident_half(reinterpret_cast<const void*>(k_ratio_dual_fused<16>), 1);  // see launch_ratio_primal
    if (nb <= max_coresident && (long)nb * BLK * pt >= (long)span) {
        if (pt == 16) hipLaunchKernelGGL(k_ratio_dual_fused<16>, dim3(nb), dim3(BLK), 0, st, dv, list_ok);  // both passes + FTRAN head
        else hipLaunchKernelGGL(k_ratio_dual_fused<4>, dim3(nb), dim3(BLK), 0, st, dv, list_ok);
        return;
    }
    hipLaunchKernelGGL(k_ratio_dual_p1, dim3(grid_for(g.n)), dim3(BLK), 0, st, dv);
    hipLaunchKernelGGL(k_ratio_dual_p2, dim3(grid_for(g.n)), dim3(BLK), 0, st, dv);  // + FTRAN head
}
// rows per tile of the fused pass: 8 while W is small (more blocks in flight, 14 vs 20 us at
// k = 1 800), 16 from cap 8192 on (half the v partials; measured 5.0-5.2 TB/s at k = 6 500 either way)
// row tiles per block of the large-nucleus tiling
constexpr int FW_RL = 1;  // measured: 4 row tiles per block slow the stream down (251 vs 163 us at k = 10 000) and the v reduce is not the bottleneck
static inline int fw_rows(const Geom& g) { return g.big ? 16 * FW_RL : 8; }
// 1-D grid of the tiled large-nucleus passes: enough blocks to fill the chip a few times over
constexpr int FW_TILE_BLOCKS = 4096;
// strip geometry of k_stream_w (columns x rows per block, rows per step)
struct SwGeom { int ch, rb, rs; };
static constexpr SwGeom kSwGeoms[] = {{512, 512, 8}, {1024, 256, 4}, {1024, 128, 4}, {512, 256, 8}, {1024, 64, 4}, {512, 128, 4}};
static int stream_variant() {
    return 2;  // 1 024 columns x 128 rows, 4 rows per step (the other shapes were A/B runs of round 2: tools/stream_bench.hip)
}
static int sw_ch() { return kSwGeoms[stream_variant()].ch; }
static int sw_rb() { return kSwGeoms[stream_variant()].rb; }
bool stream_strips_enabled();
static bool stream_strips() { return stream_strips_enabled(); }
int stream_coresident_blocks() {
    static int n = -1;
    if (n < 0) {
        int dev = 0, per_cu = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_stream_w<true, 1024, 128, 4>), BLK, 0) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            n = (per_cu < 3 ? per_cu : 3) * cus;  // measured: 3 per CU beats 4 (733 vs 740 us per pivot at k = 20 500); a
        else                                      // count that is not a multiple of the CUs loads them unevenly (759 us)
            n = 0;
    }
    return (stream_variant() == 2) ? n : 0;
}
bool stream_strips_enabled() {
    return true;
}
// A folding pivot of the lazy primal iteration (v only, no tau), unsharded: the fold kernel produces the v partials of the
// freshly folded inverse itself and the streaming pass of that pivot is skipped (one read of the inverse less in every
// lrJ pivots: 0.49 ms of 1.92 ms at k = 20 500).
bool fold_fuses_v(const DevView& dv, int with_v, int with_tau, int fold_only) {
    return with_v && !with_tau && !fold_only && !dv.wshard && dv.lrJ > 0 && stream_strips_enabled();
}
static void launch_fused_lr(const DevView& dv, const Geom& g, int with_v, int fold_only, hipStream_t st, int with_tau = 1) {
    const int rows = fw_rows(g);
    int nstripes = (g.cap + rows - 1) / rows, nchunks = (g.cap + FW_TC - 1) / FW_TC;
    dim3 b(BLK);
    if (rows == 8) {
        // the streaming pass carries one extra block row: one block per pending term (its low-rank dots)
        dim3 gr(nstripes < LR_MAX ? LR_MAX : nstripes, nchunks + 1), gf(nstripes, nchunks);
        if (!fold_only) {  // normal pivot: read-only streaming pass (exits at once on a folding pivot)
            if (with_v) LAUNCH_T(2, (k_fused_w<8, true, true, false, false, true>), gr, b, 0, st, dv);
            else LAUNCH_T(2, (k_fused_w<8, true, false, false, false, true>), gr, b, 0, st, dv);
        }
        if (with_v) hipLaunchKernelGGL((k_fused_lr<8, true, false>), gf, b, 0, st, dv, fold_only);
        else hipLaunchKernelGGL((k_fused_lr<8, false, false>), gf, b, 0, st, dv, fold_only);
        return;
    }
    // large tiles: a folding pivot folds first (k_fold_w2: U through the scalar unit; with `fuse` it also yields that pivot's v
    // partials and the streaming pass below exits at once), then every pivot streams W0 once (k_stream_w: 1 024 columns x 128 rows,
    // 4 rows per step, one balanced tile per co-resident block)
    const int fuse = fold_fuses_v(dv, with_v, with_tau, fold_only) ? 1 : 0;
    const long ftiles = (long)((g.cap + FD_RB - 1) / FD_RB) * ((g.cap + FD_CH - 1) / FD_CH);
    const int nf = (int)(ftiles < fold_max_blocks() ? ftiles : fold_max_blocks());
    if (dv.lrJ <= 16) LAUNCH_T(3, k_fold_w2<16>, dim3(nf), b, 0, st, dv, fold_only ? 1 : 2, fuse);
    else LAUNCH_T(3, k_fold_w2<LR_MAX>, dim3(nf), b, 0, st, dv, fold_only ? 1 : 2, fuse);
    if (!fold_only) {
        long tiles = (long)((g.cap + sw_rb() - 1) / sw_rb()) * ((g.cap + sw_ch() - 1) / sw_ch());
        if (dv.sw_nbal > tiles) tiles = dv.sw_nbal;  // balanced strips: one tile per co-resident block
        const int nt = (int)(tiles < SW_MAX_BLOCKS ? tiles : SW_MAX_BLOCKS);
        if (with_v) LAUNCH_T(2, (k_stream_w<true, 1024, 128, 4>), dim3(nt + LR_MAX), b, 0, st, dv, with_tau, fuse);
        else if (with_tau) LAUNCH_T(2, (k_stream_w<false, 1024, 128, 4>), dim3(nt + LR_MAX), b, 0, st, dv, 1, 0);
    }
}
void launch_fold_lowrank(const DevView& dv, const Geom& g, hipStream_t st) {
    if (!dv.lrJ || g.cap <= 0) return;
    launch_fused_lr(dv, g, 0, 1, st);
    hipLaunchKernelGGL(k_reset_nlow, dim3(1), dim3(1), 0, st, dv);
}
void launch_fused_w(const DevView& dv, const Geom& g, int with_v, hipStream_t st, int with_tau) {
    if (g.cap <= 0) return;  // a model without kept rows has no nucleus: nothing to stream (and no valid grid)
    if (dv.lrJ) {
        launch_fused_lr(dv, g, with_v, 0, st, with_tau);
        return;
    }
    const int rows = fw_rows(g);
    int nstripes = (g.cap + rows - 1) / rows, nchunks = (g.cap + FW_TC - 1) / FW_TC;
    dim3 gr(nstripes, nchunks), b(BLK);
    if (rows == 8) {
        if (with_v && with_tau) LAUNCH_T(2, (k_fused_w<8, true, true, true>), gr, b, 0, st, dv);
        else if (with_v) LAUNCH_T(2, (k_fused_w<8, false, true, true>), gr, b, 0, st, dv);
        else if (with_tau) LAUNCH_T(2, (k_fused_w<8, true, false, true>), gr, b, 0, st, dv);
        else LAUNCH_T(2, (k_fused_w<8, false, false, true>), gr, b, 0, st, dv);
    } else {
        const long tiles_cap = (long)nstripes * nchunks;
        const int nb = (int)(tiles_cap < FW_TILE_BLOCKS ? tiles_cap : FW_TILE_BLOCKS);
        if (with_v && with_tau) LAUNCH_T(2, (k_fused_w<16, true, true, true, true, false, FW_RL, true>), dim3(nb), b, 0, st, dv);
        else if (with_v) LAUNCH_T(2, (k_fused_w<16, false, true, true, true, false, FW_RL, true>), dim3(nb), b, 0, st, dv);
        else if (with_tau) LAUNCH_T(2, (k_fused_w<16, true, false, true, true, false, FW_RL, true>), dim3(nb), b, 0, st, dv);
        else LAUNCH_T(2, (k_fused_w<16, false, false, true, true, false, FW_RL, true>), dim3(nb), b, 0, st, dv);
    }
}
bool rk_rides_post(const DevView& dv, const Geom& g) {
    const char* e = std::getenv("MLP_RK_RIDE");
    if (e && e[0] == '0') return false;
    return dv.lrJ && fw_rows(g) != 8 && FW_RL == 1 && stream_strips() && !dv.wshard && dv.world <= 1;
}
int launch_post_fused(const DevView& dv, const Geom& g, int with_v, hipStream_t st, int classic, int skip_push, int with_tau, int touch, int rk_ride) {
    // touch: also build the touched-column list of the sparse tableau row in this launch (extra blocks); returns 1 when it
    // did, 0 when the caller still has to launch k_row_touch on its own
    if (!with_tau) skip_push = 1;  // lazy dual steepest edge: no tau, hence no push of -F tau_K
    if (!with_tau && !with_v) return 0;
    const int touch_asked = touch;
    if (!classic && dv.lrJ && fw_rows(g) != 8 && FW_RL == 1 && stream_strips()) {  // partials of k_stream_w's strips
#define POSTX(RB, CH)                                                                                             \
    if (sw_rb() == RB && sw_ch() == CH) hipLaunchKernelGGL((k_post_exchange<RB, CH>), dim3(blocks_for(g.cap)), dim3(BLK), 0, st, dv, with_v, with_tau);
        if (dv.wshard) {  // row-sharded pass: exchange the partials first (k_post_fused then reads the exchange buffer)
            POSTX(512, 512) POSTX(256, 1024) POSTX(128, 1024) POSTX(256, 512) POSTX(128, 512) POSTX(64, 1024)
        }
#undef POSTX
#define POSTS2(G, RB, CH)                                                                                         \
    if (sw_rb() == RB && sw_ch() == CH) {                                                                         \
        if (with_v && rk_ride) {                                                                                  \
            const int n_tail = n_push + blocks_for(g.cap, 32);                                                    \
            int n_rk = blocks_for((long)g.cap * 4);                                                               \
            if (n_rk > 1024) n_rk = 1024;                                                                         \
            hipLaunchKernelGGL((k_post_fused<G, true, RB, CH>), dim3(n_tail + n_rk), dim3(BLK), 0, st, dv, n_push, -1, fold_fuses_v(dv, with_v, with_tau, 0) ? 1 : 0, n_tail); \
        } else if (with_v) hipLaunchKernelGGL((k_post_fused<G, true, RB, CH>), dim3(n_push + blocks_for(g.cap, 32)), dim3(BLK), 0, st, dv, n_push, -1, fold_fuses_v(dv, with_v, with_tau, 0) ? 1 : 0, -1); \
        else hipLaunchKernelGGL((k_post_fused<G, false, RB, CH>), dim3(n_push), dim3(BLK), 0, st, dv, n_push);    \
    }
#define POSTS(G)                                                                                                  \
    do {                                                                                                          \
        int n_push = with_tau ? blocks_for((long)g.cap * G) : 0;                                                  \
        POSTS2(G, 512, 512); POSTS2(G, 256, 1024); POSTS2(G, 128, 1024); POSTS2(G, 256, 512); POSTS2(G, 128, 512); POSTS2(G, 64, 1024);  \
    } while (0)
        LANES_SWITCH(g.lanes, POSTS(4), POSTS(16), POSTS(64));
#undef POSTS
#undef POSTS2
        if (dv.pb_on && !skip_push) launch_blocked_push(dv, 1, st);
        else if (dv.det_pull && with_tau) launch_pull_F(dv, g, 1, st);
        return 0;
    }
#define POSTF(G)                                                                                                  \
    do {                                                                                                          \
        int n_push = with_tau ? blocks_for((long)g.cap * G) : 0;                                                  \
        if (with_v && fw_rows(g) == 8 && touch) {                                                                 \
            const int n_tail = n_push + blocks_for(g.cap, 32);                                                    \
            hipLaunchKernelGGL((k_post_fused<G, true, 8>), dim3(n_tail + blocks_for((long)(g.cap + 1) * 64)), dim3(BLK), 0, st, dv, n_push, n_tail); \
            touch = 0;                                                                                            \
        } else if (with_v && fw_rows(g) == 8) hipLaunchKernelGGL((k_post_fused<G, true, 8>), dim3(n_push + blocks_for(g.cap, 32)), dim3(BLK), 0, st, dv, n_push); \
        else if (with_v) hipLaunchKernelGGL((k_post_fused<G, true, 16 * FW_RL>), dim3(n_push + blocks_for(g.cap, 32)), dim3(BLK), 0, st, dv, n_push); \
        else hipLaunchKernelGGL((k_post_fused<G, false, 16>), dim3(n_push), dim3(BLK), 0, st, dv, n_push);        \
    } while (0)
    LANES_SWITCH(g.lanes, POSTF(4), POSTF(16), POSTF(64));
#undef POSTF
    if (dv.pb_on && !skip_push) launch_blocked_push(dv, 1, st);
    else if (dv.det_pull && with_tau) launch_pull_F(dv, g, 1, st);
    return (touch_asked && !touch) ? 1 : 0;
}
void launch_structure_update(const DevView& dv, const Geom& g, hipStream_t st) {
    hipLaunchKernelGGL(k_struct_update, dim3(blocks_for(g.cap)), dim3(BLK), 0, st, dv);
}
bool update_pulls_inside(const DevView& dv, const Geom& g) {
    const char* e = std::getenv("MLP_PULL_INSIDE");  // (tests: the three forms of the small-nucleus iteration must agree, tests/test_primal_head.py)
    if (e && e[0] == '0') return false;
    const int t = g.m > g.n ? g.m : g.n;
    return primal_head_supported(dv, g) && blocks_for(t, BLK * 4) <= 2048 && dv.hy_stamp_n != nullptr;  // at most four positions per thread
}
bool head_applies(const DevView& dv, const Geom& g) {  // ... and the head applies the basic side and position q itself (the update kernel: non-basic side only)
    const char* e = std::getenv("MLP_HEAD_APPLY");
    if (e && e[0] == '0') return false;
    return update_pulls_inside(dv, g) && dv.nb_order == nullptr && dv.pk_valid == nullptr;
}
void launch_update_pivot(const DevView& dv, const Geom& g, int phase, int use_dse, int use_pse, hipStream_t st, int inline_comb,
                         int with_struct, int pull_inside) {
    int t = g.m > g.n ? g.m : g.n;
    // inline_comb (primal iteration with the banded sweep): the update kernel sums the per-band partials itself
    // with_struct (dual iteration without PSE): the partition change rides in the tail blocks of this launch
    // one position per thread: measured against 4 per thread (a quarter of the blocks and tickets) the longer per-thread
    // chain of band-partial loads costs more than the tickets save (109.0 vs 104.3 us per pivot); the kernel strides
    // only beyond 512 * 4 * 256 positions
    // positions per thread: one up to 512 workgroups (measured against 2 and 4 on config 4, mid / late windows: 244.1 / 244.7 / 248.2 and 663.3 /
    // 664.0 / 667.8 us per pivot), beyond that as many as keep the grid within 512 (the 400 000-column transport instance: 164.6 / 156.2 /
    // 154.0 us per pivot at 1 / 2 / 4 — 1 563 workgroups queue on the ticket and the launch ramp)
    // (re-measured with the two-step ticket on the transport instance: 123.8 / 126.9 / 119.3 us per pivot at 1 / 2 / 4 — unchanged choice)
    const int upd_pt = std::max(1, std::min(8, (blocks_for(t) + 511) / 512));
    int n_upd = blocks_for(t, BLK * upd_pt) <= 2048 ? blocks_for(t, BLK * upd_pt) : 2048;
    // a pulling launch: one position per thread up to 512 workgroups, at most four (k_update_pivot: UPT).  Four — 98 workgroups on config 4 —
    // was the choice while 391 arrivals queued ~8 us on the one-address ticket; with the two-step ticket the early window of config 4
    // runs 38.1 / 37.9 us per pivot at two / one against 39.7 at four
    const int pull_pt = std::max(1, std::min(4, (blocks_for(pull_inside == 2 ? g.n : t) + 511) / 512));
    if (pull_inside == 2) n_upd = blocks_for(g.n, BLK * pull_pt);   // non-basic side only
    else if (pull_inside) n_upd = blocks_for(t, BLK * pull_pt);
    hipLaunchKernelGGL(k_update_pivot, dim3(n_upd + (with_struct ? blocks_for(g.cap) : 0)), dim3(BLK), 0, st, dv, phase, use_dse, use_pse,
                       inline_comb, n_upd, pull_inside);
}
void launch_set_iter(const DevView& dv, int status, int q, int r, double lnv, int forced, hipStream_t st) {
    hipLaunchKernelGGL(k_set_iter, dim3(1), dim3(1), 0, st, dv, status, q, r, lnv, forced);
}
void launch_reset_ring(const DevView& dv, hipStream_t st) { hipLaunchKernelGGL(k_reset_ring, dim3(1), dim3(1), 0, st, dv); }
void launch_btran_dense(const DevView& dv, const Geom& g, hipStream_t st) {
    if (dv.fac_on) {  // compact factor: c_B by position -> alpha_q, y = B^-T c_B -> rv.y
        launch_fac_gather_cb(dv, st);
        launch_fac_solve(dv, g, 1, 2, 1, dv.alpha_q, 1, st);  // (src 2: an arbitrary dense vector — every level is walked)
        return;
    }
    if (g.cap <= 0) return;
    // c_B by position -> alpha_q, y_S -> rv.y; then tK, vK = W^T tK, scatter into rv.y
    hipLaunchKernelGGL(k_gather_basic_obj, dim3(blocks_for(g.m)), dim3(BLK), 0, st, dv);
    launch_btran_rhs(dv, g, st);
    int nstripes = (g.cap + 16 - 1) / 16, nchunks = (g.cap + FW_TC - 1) / FW_TC;
    hipLaunchKernelGGL((k_fused_w<16, false, true, false>), dim3(nstripes, nchunks), dim3(BLK), 0, st, dv);
    hipLaunchKernelGGL(k_reduce_v<16>, dim3(blocks_for(g.cap)), dim3(BLK), 0, st, dv);
}
// ---- recomputation of the basic values from the basis: x_B = B^-1 (b - N x_N)
// (the reference has recalc_basic_var_vals, solver.rs:1177-1197, as dead code; used here to polish long runs)
// step 1: r_i = b_i - sum over the NON-basic entries of row i of a * x_N   (G lanes per CSR row)
// (refine = 1: the full residual b - A x of the current point, for one step of iterative refinement)
template <int G>
__global__ void __launch_bounds__(BLK) k_residual_rhs(DevView v, const double* rhs, double* r, int refine) {
    const int i = (blockIdx.x * BLK + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (i >= v.m) return;
    double acc = 0.0;
    const int end = v.csr_ptr[i + 1];
    for (int e = v.csr_ptr[i] + gl; e < end; e += G) {
        const int loc = v.var_loc[v.csr_col[e]];
        if (loc < 0) acc += v.csr_val[e] * v.xN[-1 - loc];
        else if (refine) acc += v.csr_val[e] * v.xB[loc];
    }
    acc = group_sum<G>(acc);
    if (gl == 0) r[i] = rhs[i] - acc;
}
// step 2: seed the tau path of the fused pass with r: rK by nucleus slot, tau = D^-1 r_S by singleton position
__global__ void __launch_bounds__(BLK) k_seed_dense_ftran(DevView v, const double* r) {
    const int t = blockIdx.x * BLK + threadIdx.x;
    Ctl* c = v.ctl;
    if (t == 0) {
        c->it.status = ITER_PIVOT;
        c->halt = 0;
        c->up.kase = -1;
        c->fold = 0;
    }
    if (t < c->k) v.rK[t] = r[v.row_of_kslot[t]];
    if (t < v.m && v.kslot_of_pos[t] < 0) v.tau[t] = r[v.srow_of_pos[t]] / v.sdiag_of_pos[t];
}
__global__ void __launch_bounds__(BLK) k_copy_tau_to_xb(DevView v, int refine) {
    const int p = blockIdx.x * BLK + threadIdx.x;
    if (p < v.m) v.xB[p] = refine ? v.xB[p] + v.tau[p] : v.tau[p];
}
// x_B = B^-1 r through the tau path: tau_K = W r_K (fused pass without update), then -F tau_K into the
// singleton positions (post-fused tail), exactly as tau = B^-1 rho is computed every pivot.
void launch_recalc_basic_vals(const DevView& dv, const Geom& g, const double* rhs, double* r_tmp, int refine, hipStream_t st) {
    if (g.lanes <= 4) hipLaunchKernelGGL(k_residual_rhs<4>, dim3(blocks_for((long)g.m * 4)), dim3(BLK), 0, st, dv, rhs, r_tmp, refine);
    else if (g.lanes <= 16) hipLaunchKernelGGL(k_residual_rhs<16>, dim3(blocks_for((long)g.m * 16)), dim3(BLK), 0, st, dv, rhs, r_tmp, refine);
    else hipLaunchKernelGGL(k_residual_rhs<64>, dim3(blocks_for((long)g.m * 64)), dim3(BLK), 0, st, dv, rhs, r_tmp, refine);
    launch_clear_work(dv, st);
    if (dv.fac_on) {  // compact factor: tau = B^-1 r by one level-scheduled solve
        launch_fac_solve(dv, g, 0, 2, 1, r_tmp, 1, st);
        hipLaunchKernelGGL(k_copy_tau_to_xb, dim3(blocks_for(g.m)), dim3(BLK), 0, st, dv, refine);
        launch_clear_work(dv, st);
        return;
    }
    const int t = g.m > g.cap ? g.m : g.cap;
    hipLaunchKernelGGL(k_seed_dense_ftran, dim3(blocks_for(t)), dim3(BLK), 0, st, dv, (const double*)r_tmp);
    // Large-nucleus regime: the dense-rhs FTRAN x_K = W0 r_K is ONE streaming read of the nucleus inverse through the
    // strip kernel of the pivot loop (k_stream_w, tau side only; the caller has folded the pending terms).  A sharded
    // solve keeps the replicated classic pass (no exchange outside the pivot loop).
    const bool strips = dv.lrJ && fw_rows(g) != 8 && FW_RL == 1 && stream_strips() && !dv.wshard;
    if (g.cap > 0) {
        if (strips) {
            launch_fused_lr(dv, g, 0, 0, st, 1);
        } else {
            const int nstripes = (g.cap + 16 - 1) / 16, nchunks = (g.cap + FW_TC - 1) / FW_TC;
            LAUNCH_T(2, (k_fused_w<16, true, false, false>), dim3(nstripes, nchunks), dim3(BLK), 0, st, dv);
        }
    }
    launch_post_fused(dv, g, 0, st, strips ? 0 : 1);  // partials in the strip tiling / in the classic 16 x 1024 tiling
    hipLaunchKernelGGL(k_copy_tau_to_xb, dim3(blocks_for(g.m)), dim3(BLK), 0, st, dv, refine);
    launch_clear_work(dv, st);
}
void launch_recalc_d(const DevView& dv, const Geom& g, hipStream_t st) {
    LANES_SWITCH(g.lanes,
                 hipLaunchKernelGGL(k_recalc_d<4>, dim3(blocks_for((long)g.n * 4)), dim3(BLK), 0, st, dv),
                 hipLaunchKernelGGL(k_recalc_d<16>, dim3(blocks_for((long)g.n * 16)), dim3(BLK), 0, st, dv),
                 hipLaunchKernelGGL(k_recalc_d<64>, dim3(blocks_for((long)g.n * 64)), dim3(BLK), 0, st, dv));
    hipLaunchKernelGGL(k_recalc_obj, dim3(grid_for(g.m + g.n)), dim3(BLK), 0, st, dv);
}
void launch_shift_nonbasic(const DevView& dv, const Geom& g, int col, double val, hipStream_t st) {
    hipLaunchKernelGGL(k_shift_nonbasic, dim3(blocks_for(g.m)), dim3(BLK), 0, st, dv, col, val);
    hipLaunchKernelGGL(k_set_xn, dim3(1), dim3(1), 0, st, dv, col, val);
}
void launch_sq_norms_add_row(const DevView& dv, const Geom& g, hipStream_t st) {
    hipLaunchKernelGGL(k_gamma_add_row, dim3(blocks_for(g.n)), dim3(BLK), 0, st, dv);
}
void launch_copy_rho_sq_to_beta(const DevView& dv, int row, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_rho_sq_to_beta, dim3(1), dim3(1), 0, st, dv, row);
}
void launch_build_nucleus(const DevView& dv, const Geom& g, double* Kd, int k, hipStream_t st) {
    if (k <= 0) return;
    (void)hipMemsetAsync(Kd, 0, sizeof(double) * (size_t)k * dv.ld, st);
    hipLaunchKernelGGL(k_build_nucleus<16>, dim3(blocks_for((long)k * 16)), dim3(BLK), 0, st, dv, Kd, k);
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

}  // namespace mlp
