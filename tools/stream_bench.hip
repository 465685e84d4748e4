// stream_bench.hip — micro-benchmark of the read-only pass over the dense nucleus inverse W
// (tau = W rho, v = W^T t; the streaming pass of the delayed-update mode, HISTORY.md §2.1).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_bench.hip -o tools/stream_bench
// Run  : tools/stream_bench [k] [ld] [reps]     (prints GB/s of 8 k^2 algorithmic bytes per variant)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));          \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef double dbl2_t __attribute__((ext_vector_type(2)));
typedef double dbl4_t __attribute__((ext_vector_type(4)));
constexpr int BLK = 256;

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

// ---- ceiling: plain sum of W (rows k, pitch ld), 16-byte non-temporal loads, grid-stride over 16x1024 tiles
template <bool NT>
__global__ void __launch_bounds__(BLK) k_read_only(const double* __restrict__ W, int k, int ld, double* out) {
    const int nch = (k + 1023) / 1024;
    const int ntiles = ((k + 15) / 16) * nch;
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = (tile / nch) * 16, col0 = (tile % nch) * 1024 + 2 * threadIdx.x;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const int row = row0 + a;
            if (row < k) {
                const double* wp = W + (size_t)row * ld;
                if (col0 + 1 < k) {
                    dbl2_t t = NT ? __builtin_nontemporal_load((const dbl2_t*)(wp + col0)) : *(const dbl2_t*)(wp + col0);
                    acc += t.x + t.y;
                }
                if (col0 + 513 < k) {
                    dbl2_t t = NT ? __builtin_nontemporal_load((const dbl2_t*)(wp + col0 + 512)) : *(const dbl2_t*)(wp + col0 + 512);
                    acc += t.x + t.y;
                }
            }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

// ---- V0: the shape of the current k_fused_w<16, tau, v, no update, NT> (2-D grid, one tile per block)
__global__ void __launch_bounds__(BLK) k_v0(const double* __restrict__ W, int k, int ld, const double* rK, const double* tK,
                                            double* part_tau, double* part_v) {
    __shared__ double s_tau[16][BLK / 64];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * 16, col0 = blockIdx.y * 1024;
    if (row0 >= k || col0 >= k) return;
    int cidx[4] = {col0 + 2 * tid, col0 + 2 * tid + 1, col0 + 512 + 2 * tid, col0 + 512 + 2 * tid + 1};
    double rk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rk[j] = cidx[j] < k ? rK[cidx[j]] : 0.0;
    double vacc[4] = {0, 0, 0, 0};
    const bool pair0 = cidx[1] < k, pair1 = cidx[3] < k;
    const int wv = tid >> 6, l = tid & 63;
    double tacc[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int row = row0 + a;
        tacc[a] = 0.0;
        if (row >= k) continue;
        const double* wp = W + (size_t)row * ld;
        double w[4] = {0, 0, 0, 0};
        if (pair0) {
            dbl2_t t = __builtin_nontemporal_load((const dbl2_t*)(wp + cidx[0]));
            w[0] = t.x; w[1] = t.y;
        }
        if (pair1) {
            dbl2_t t = __builtin_nontemporal_load((const dbl2_t*)(wp + cidx[2]));
            w[2] = t.x; w[3] = t.y;
        }
        tacc[a] = w[0] * rk[0] + w[1] * rk[1] + w[2] * rk[2] + w[3] * rk[3];
        const double t = tK[row];
#pragma unroll
        for (int j = 0; j < 4; ++j) vacc[j] += w[j] * t;
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        double s = wave_sum(tacc[a]);
        if (l == 0) s_tau[a][wv] = s;
    }
    double* pv = part_v + (size_t)blockIdx.x * ld;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (cidx[j] < k) pv[cidx[j]] = vacc[j];
    __syncthreads();
    if (tid < 16 && row0 + tid < k) part_tau[(size_t)blockIdx.y * ld + row0 + tid] = s_tau[tid][0] + s_tau[tid][1] + s_tau[tid][2] + s_tau[tid][3];
}

// ---- V1: column strip: a block owns CH columns x RB rows and walks the rows in steps of RS with all RS x (CH/BLK/2)
// 16-byte loads of a step issued before any of them is consumed; v accumulates in registers over the whole strip
// (part_v has k / RB rows instead of k / 16), tau partials per row go through a wave reduction.
// CH = 1024: 2 pairs per thread (as V0); CH = 512: 1 pair per thread.
template <int CH, int RB, int RS, bool NT>
__global__ void __launch_bounds__(BLK) k_v1(const double* __restrict__ W, int k, int ld, const double* __restrict__ rK,
                                            const double* __restrict__ tK, double* part_tau, double* part_v) {
    constexpr int NP = CH / (2 * BLK);  // pairs per thread
    __shared__ double s_tau[RS][BLK / 64];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int nch = (k + CH - 1) / CH;
    const int nstr = (k + RB - 1) / RB;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += gridDim.x) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        int c0[NP];
        double rk[NP][2], vacc[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c0[p] = chunk * CH + p * 2 * BLK + 2 * tid;
            rk[p][0] = c0[p] < k ? rK[c0[p]] : 0.0;
            rk[p][1] = c0[p] + 1 < k ? rK[c0[p] + 1] : 0.0;
            vacc[p][0] = vacc[p][1] = 0.0;
            if (c0[p] + 1 >= k) c0[p] = -1;  // (edge columns: k is even in this benchmark)
        }
        for (int r0 = rbeg; r0 < rend; r0 += RS) {
            dbl2_t w[RS][NP];
            double t[RS];
            if (r0 + RS <= rend) {  // interior step: no row checks
#pragma unroll
                for (int a = 0; a < RS; ++a) {
                    const double* wp = W + (size_t)(r0 + a) * ld;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        if (c0[p] >= 0) w[a][p] = NT ? __builtin_nontemporal_load((const dbl2_t*)(wp + c0[p])) : *(const dbl2_t*)(wp + c0[p]);
                        else w[a][p] = dbl2_t{0.0, 0.0};
                    }
                    t[a] = tK[r0 + a];
                }
            } else {
#pragma unroll
                for (int a = 0; a < RS; ++a) {
                    const bool ok = r0 + a < rend;
                    const double* wp = W + (size_t)(ok ? r0 + a : rbeg) * ld;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        if (ok && c0[p] >= 0) w[a][p] = *(const dbl2_t*)(wp + c0[p]);
                        else w[a][p] = dbl2_t{0.0, 0.0};
                    }
                    t[a] = ok ? tK[r0 + a] : 0.0;
                }
            }
            double tacc[RS];
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                double s = 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    s += w[a][p].x * rk[p][0] + w[a][p].y * rk[p][1];
                    vacc[p][0] += w[a][p].x * t[a];
                    vacc[p][1] += w[a][p].y * t[a];
                }
                tacc[a] = s;
            }
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                double s = wave_sum(tacc[a]);
                if (l == 0) s_tau[a][wv] = s;
            }
            __syncthreads();
            if (tid < RS && r0 + tid < rend)
                part_tau[(size_t)chunk * ld + r0 + tid] = s_tau[tid][0] + s_tau[tid][1] + s_tau[tid][2] + s_tau[tid][3];
            __syncthreads();
        }
        double* pv = part_v + (size_t)strip * ld;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (c0[p] >= 0) {
                pv[c0[p]] = vacc[p][0];
                pv[c0[p] + 1] = vacc[p][1];
            }
    }
}

// ---- V2: like V1 but the tau reduction is deferred: per-thread partial row sums for a step go to LDS transposed and
// one wave reduces them (fewer shuffles in the streaming loop)
template <int CH, int RB, int RS, bool NT>
__global__ void __launch_bounds__(BLK) k_v2(const double* __restrict__ W, int k, int ld, const double* __restrict__ rK,
                                            const double* __restrict__ tK, double* part_tau, double* part_v) {
    constexpr int NP = CH / (2 * BLK);
    __shared__ double s_t[RS][BLK + 1];
    const int tid = threadIdx.x;
    const int nch = (k + CH - 1) / CH;
    const int nstr = (k + RB - 1) / RB;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += gridDim.x) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        int c0[NP];
        double rk[NP][2], vacc[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c0[p] = chunk * CH + p * 2 * BLK + 2 * tid;
            rk[p][0] = c0[p] < k ? rK[c0[p]] : 0.0;
            rk[p][1] = c0[p] + 1 < k ? rK[c0[p] + 1] : 0.0;
            vacc[p][0] = vacc[p][1] = 0.0;
            if (c0[p] + 1 >= k) c0[p] = -1;
        }
        for (int r0 = rbeg; r0 < rend; r0 += RS) {
            dbl2_t w[RS][NP];
            double t[RS];
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const bool ok = r0 + a < rend;
                const double* wp = W + (size_t)(ok ? r0 + a : rbeg) * ld;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (ok && c0[p] >= 0) w[a][p] = NT ? __builtin_nontemporal_load((const dbl2_t*)(wp + c0[p])) : *(const dbl2_t*)(wp + c0[p]);
                    else w[a][p] = dbl2_t{0.0, 0.0};
                }
                t[a] = ok ? tK[r0 + a] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                double s = 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    s += w[a][p].x * rk[p][0] + w[a][p].y * rk[p][1];
                    vacc[p][0] += w[a][p].x * t[a];
                    vacc[p][1] += w[a][p].y * t[a];
                }
                s_t[a][tid] = s;
            }
            __syncthreads();
            // RS rows x 256 partials: 256 threads = RS groups of 256/RS lanes
            {
                constexpr int G = BLK / RS;  // lanes per row (16 for RS = 16)
                const int row = tid / G, gl = tid % G;
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < BLK / G; ++j) s += s_t[row][gl + j * G];
#pragma unroll
                for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (gl == 0 && r0 + row < rend) part_tau[(size_t)chunk * ld + r0 + row] = s;
            }
            __syncthreads();
        }
        double* pv = part_v + (size_t)strip * ld;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (c0[p] >= 0) {
                pv[c0[p]] = vacc[p][0];
                pv[c0[p] + 1] = vacc[p][1];
            }
    }
}


// ---- V3: wave-autonomous strips, no LDS, no barriers.  A wave owns NP*128 columns x RB rows and walks the rows in
// steps of RS; the loads of step i+1 are issued before step i is consumed (register double buffer); the RS row sums of
// a step are reduced across the 64 lanes with a halving butterfly (RS + 5 shuffle-adds for RS = 16 instead of 6 RS):
// after the exchange with lane^32 a lane keeps RS/2 sums, after lane^16 RS/4, ..., so that finally lane (j * 64 / RS)
// holds row j's total.
template <int RS>
__device__ __forceinline__ void butterfly_reduce(double (&x)[RS], int lane) {
    // invariant: after the step with offset o, x[0 .. cnt) are the live partial sums of this lane
    int cnt = RS;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        if (cnt > 1) {
            const int half = cnt / 2;
            const bool upper = (lane & o) != 0;
#pragma unroll
            for (int j = 0; j < RS / 2; ++j) {
                if (j < half) {
                    // lower lanes keep rows [0, half), upper lanes keep rows [half, cnt): send the other half
                    const double send = upper ? x[j] : x[j + half];
                    const double keep = upper ? x[j + half] : x[j];
                    x[j] = keep + __shfl_xor(send, o, 64);
                }
            }
            cnt = half;
        } else {
            x[0] += __shfl_xor(x[0], o, 64);
        }
    }
}
// which row (0 .. RS-1) ends up in x[0] of `lane` after butterfly_reduce: bit b of the row index, counted from the top,
// is bit (5 - b) of the lane for the first log2(RS) steps
template <int RS>
__device__ __forceinline__ int butterfly_row(int lane) {
    int row = 0, cnt = RS, o = 32;
    while (cnt > 1) {
        cnt /= 2;
        if (lane & o) row += cnt;
        o >>= 1;
    }
    return row;
}
template <int NP, int RB, int RS, bool PF>
__global__ void __launch_bounds__(BLK) k_v3(const double* __restrict__ W, int k, int ld, const double* __restrict__ rK,
                                            const double* __restrict__ tK, double* part_tau, double* part_v) {
    constexpr int CW = NP * 128;  // columns per wave
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = (gridDim.x * BLK) >> 6;
    const int nch = (k + CW - 1) / CW, nstr = (k + RB - 1) / RB;
    constexpr int LOG = RS == 16 ? 4 : (RS == 8 ? 3 : (RS == 4 ? 2 : 5));
    const bool writer = (lane & ((64 >> LOG) - 1)) == 0;   // lanes that hold a finished row sum
    const int myrow = butterfly_row<RS>(lane);
    for (int tile = wave; tile < nstr * nch; tile += nwaves) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        int c0[NP];
        double rk[NP][2], vacc[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c0[p] = chunk * CW + p * 128 + 2 * lane;
            const bool ok = c0[p] + 1 < k;
            rk[p][0] = ok ? rK[c0[p]] : 0.0;
            rk[p][1] = ok ? rK[c0[p] + 1] : 0.0;
            vacc[p][0] = vacc[p][1] = 0.0;
            if (!ok) c0[p] = -1;
        }
        dbl2_t w[RS][NP], wn[RS][NP];
        auto load_step = [&](dbl2_t (&dst)[RS][NP], int r0) {
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const bool ok = r0 + a < rend;
                const double* wp = W + (size_t)(ok ? r0 + a : rbeg) * ld;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (ok && c0[p] >= 0) dst[a][p] = __builtin_nontemporal_load((const dbl2_t*)(wp + c0[p]));
                    else dst[a][p] = dbl2_t{0.0, 0.0};
                }
            }
        };
        load_step(w, rbeg);
        for (int r0 = rbeg; r0 < rend; r0 += RS) {
            if (PF && r0 + RS < rend) load_step(wn, r0 + RS);
            double tacc[RS];
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const double t = (r0 + a < rend) ? tK[r0 + a] : 0.0;
                double s = 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    s += w[a][p].x * rk[p][0] + w[a][p].y * rk[p][1];
                    vacc[p][0] += w[a][p].x * t;
                    vacc[p][1] += w[a][p].y * t;
                }
                tacc[a] = s;
            }
            butterfly_reduce<RS>(tacc, lane);
            if (writer && r0 + myrow < rend) part_tau[(size_t)chunk * ld + r0 + myrow] = tacc[0];
            if (PF) {
#pragma unroll
                for (int a = 0; a < RS; ++a)
#pragma unroll
                    for (int p = 0; p < NP; ++p) w[a][p] = wn[a][p];
            } else if (r0 + RS < rend) {
                load_step(w, r0 + RS);
            }
        }
        double* pv = part_v + (size_t)strip * ld;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (c0[p] >= 0) {
                pv[c0[p]] = vacc[p][0];
                pv[c0[p] + 1] = vacc[p][1];
            }
    }
}

// ---- V4: wave-autonomous strips with a wave-private LDS transpose for the row sums (no block barrier, few shuffles):
// every lane writes its RS row partials to the wave's LDS tile [RS][64+1], then lane l sums row (l / (64/RS))'s
// partials over its (64/RS)-lane group slice and a short xor tree finishes.  PF: loads of step i+1 issued before step i
// is consumed.
template <int NP, int RB, int RS, bool PF>
__global__ void __launch_bounds__(BLK) k_v4(const double* __restrict__ W, int k, int ld, const double* __restrict__ rK,
                                            const double* __restrict__ tK, double* part_tau, double* part_v) {
    constexpr int CW = NP * 128;
    constexpr int G = 64 / RS;  // lanes that share a row in the reduction
    __shared__ double s_t[BLK / 64][RS][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = (gridDim.x * BLK) >> 6;
    const int nch = (k + CW - 1) / CW, nstr = (k + RB - 1) / RB;
    const int myrow = lane / G, gl = lane % G;
    for (int tile = wave; tile < nstr * nch; tile += nwaves) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        int c0[NP];
        double rk[NP][2], vacc[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c0[p] = chunk * CW + p * 128 + 2 * lane;
            const bool ok = c0[p] + 1 < k;
            rk[p][0] = ok ? rK[c0[p]] : 0.0;
            rk[p][1] = ok ? rK[c0[p] + 1] : 0.0;
            vacc[p][0] = vacc[p][1] = 0.0;
            if (!ok) c0[p] = -1;
        }
        dbl2_t w[RS][NP], wn[RS][NP];
        auto load_step = [&](dbl2_t (&dst)[RS][NP], int r0) {
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const bool ok = r0 + a < rend;
                const double* wp = W + (size_t)(ok ? r0 + a : rbeg) * ld;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (ok && c0[p] >= 0) dst[a][p] = __builtin_nontemporal_load((const dbl2_t*)(wp + c0[p]));
                    else dst[a][p] = dbl2_t{0.0, 0.0};
                }
            }
        };
        load_step(w, rbeg);
        for (int r0 = rbeg; r0 < rend; r0 += RS) {
            if (PF && r0 + RS < rend) load_step(wn, r0 + RS);
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const double t = (r0 + a < rend) ? tK[r0 + a] : 0.0;
                double sacc = 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    sacc += w[a][p].x * rk[p][0] + w[a][p].y * rk[p][1];
                    vacc[p][0] += w[a][p].x * t;
                    vacc[p][1] += w[a][p].y * t;
                }
                s_t[wv][a][lane] = sacc;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < 64 / G; ++j) sum += s_t[wv][myrow][gl + j * G];
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (gl == 0 && r0 + myrow < rend) part_tau[(size_t)chunk * ld + r0 + myrow] = sum;
            __builtin_amdgcn_wave_barrier();
            if (PF) {
#pragma unroll
                for (int a = 0; a < RS; ++a)
#pragma unroll
                    for (int p = 0; p < NP; ++p) w[a][p] = wn[a][p];
            } else if (r0 + RS < rend) {
                load_step(w, r0 + RS);
            }
        }
        double* pv = part_v + (size_t)strip * ld;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (c0[p] >= 0) {
                pv[c0[p]] = vacc[p][0];
                pv[c0[p] + 1] = vacc[p][1];
            }
    }
}
// ---- V5: V2 (block strips, LDS transpose behind block barriers) with the loads of the next step issued first
template <int CH, int RB, int RS>
__global__ void __launch_bounds__(BLK) k_v5(const double* __restrict__ W, int k, int ld, const double* __restrict__ rK,
                                            const double* __restrict__ tK, double* part_tau, double* part_v) {
    constexpr int NP = CH / (2 * BLK);
    __shared__ double s_t[2][RS][BLK + 1];
    const int tid = threadIdx.x;
    const int nch = (k + CH - 1) / CH;
    const int nstr = (k + RB - 1) / RB;
    constexpr int G = BLK / RS;
    const int row = tid / G, gl = tid % G;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += gridDim.x) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        int c0[NP];
        double rk[NP][2], vacc[NP][2];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c0[p] = chunk * CH + p * 2 * BLK + 2 * tid;
            rk[p][0] = c0[p] < k ? rK[c0[p]] : 0.0;
            rk[p][1] = c0[p] + 1 < k ? rK[c0[p] + 1] : 0.0;
            vacc[p][0] = vacc[p][1] = 0.0;
            if (c0[p] + 1 >= k) c0[p] = -1;
        }
        dbl2_t w[RS][NP], wn[RS][NP];
        auto load_step = [&](dbl2_t (&dst)[RS][NP], int r0) {
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const bool ok = r0 + a < rend;
                const double* wp = W + (size_t)(ok ? r0 + a : rbeg) * ld;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (ok && c0[p] >= 0) dst[a][p] = __builtin_nontemporal_load((const dbl2_t*)(wp + c0[p]));
                    else dst[a][p] = dbl2_t{0.0, 0.0};
                }
            }
        };
        load_step(w, rbeg);
        int buf = 0;
        for (int r0 = rbeg; r0 < rend; r0 += RS, buf ^= 1) {
            if (r0 + RS < rend) load_step(wn, r0 + RS);
#pragma unroll
            for (int a = 0; a < RS; ++a) {
                const double t = (r0 + a < rend) ? tK[r0 + a] : 0.0;
                double sacc = 0.0;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    sacc += w[a][p].x * rk[p][0] + w[a][p].y * rk[p][1];
                    vacc[p][0] += w[a][p].x * t;
                    vacc[p][1] += w[a][p].y * t;
                }
                s_t[buf][a][tid] = sacc;
            }
            __syncthreads();  // one barrier per step: the two LDS buffers alternate
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < BLK / G; ++j) sum += s_t[buf][row][gl + j * G];
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (gl == 0 && r0 + row < rend) part_tau[(size_t)chunk * ld + r0 + row] = sum;
#pragma unroll
            for (int a = 0; a < RS; ++a)
#pragma unroll
                for (int p = 0; p < NP; ++p) w[a][p] = wn[a][p];
        }
        __syncthreads();
        double* pv = part_v + (size_t)strip * ld;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (c0[p] >= 0) {
                pv[c0[p]] = vacc[p][0];
                pv[c0[p] + 1] = vacc[p][1];
            }
    }
}

// check kernel for the butterfly: every lane contributes lane * 100 + row; row j must total sum_l (100 l + j)
__global__ void k_check_butterfly(double* out) {
    double x[16];
    const int lane = threadIdx.x & 63;
    for (int j = 0; j < 16; ++j) x[j] = lane * 100.0 + j;
    butterfly_reduce<16>(x, lane);
    if ((lane & 3) == 0) out[butterfly_row<16>(lane)] = x[0];
    double y[8];
    for (int j = 0; j < 8; ++j) y[j] = lane * 100.0 + j;
    butterfly_reduce<8>(y, lane);
    if ((lane & 7) == 0) out[16 + butterfly_row<8>(lane)] = y[0];
}

// pseudo-random fill (splitmix-style hash of the element index): all-zero data streams measurably faster than real data
__global__ void k_fill(double* W, size_t n, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        W[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3;
    }
}
template <class F>
static double time_it(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 20480;
    const int ld = argc > 2 ? atoi(argv[2]) : 32784;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    double *W, *rK, *tK, *pt, *pv, *out;
    CK(hipMalloc(&W, (size_t)k * ld * 8));
    CK(hipMalloc(&rK, (size_t)ld * 8));
    CK(hipMalloc(&tK, (size_t)ld * 8));
    CK(hipMalloc(&pt, (size_t)(k / 128 + 2) * ld * 8));
    CK(hipMalloc(&pv, (size_t)(k / 16 + 2) * ld * 8));
    CK(hipMalloc(&out, 64));
    const bool zeros = argc > 4 && atoi(argv[4]) == 1;
    if (zeros) CK(hipMemset(W, 0, (size_t)k * ld * 8));
    else hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, W, (size_t)k * ld, 12345ull);
    CK(hipDeviceSynchronize());
    printf("W filled with %s\n", zeros ? "zeros" : "pseudo-random values");
    std::vector<double> h(ld, 1.0);
    CK(hipMemcpy(rK, h.data(), (size_t)ld * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tK, h.data(), (size_t)ld * 8, hipMemcpyHostToDevice));
    const double bytes = 8.0 * k * (double)k;
    auto report = [&](const char* name, double us) { printf("%-44s %9.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0); fflush(stdout); };
    printf("k = %d, ld = %d, W = %.2f GB, algorithmic bytes per pass = %.2f GB\n", k, ld, (double)k * ld * 8e-9, bytes * 1e-9);
    for (int nb : {2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read-only sum, NT, %d blocks", nb);
        report(nm, time_it([&] { hipLaunchKernelGGL(k_read_only<true>, dim3(nb), dim3(BLK), 0, 0, W, k, ld, out); }, reps));
    }
    report("read-only sum, plain loads, 4096 blocks", time_it([&] { hipLaunchKernelGGL(k_read_only<false>, dim3(4096), dim3(BLK), 0, 0, W, k, ld, out); }, reps));
    {
        dim3 gr((k + 15) / 16, (k + 1023) / 1024);
        report("V0 current shape (16x1024 tile / block, 2-D)", time_it([&] { hipLaunchKernelGGL(k_v0, gr, dim3(BLK), 0, 0, W, k, ld, rK, tK, pt, pv); }, reps));
    }
#define RUN_V(K, CH, RB, RS, NT, NB)                                                                                          \
    {                                                                                                                         \
        char nm[96];                                                                                                          \
        snprintf(nm, sizeof nm, #K " CH=%d RB=%d RS=%d %s, %d blocks", CH, RB, RS, NT ? "NT" : "plain", NB);                  \
        report(nm, time_it([&] { hipLaunchKernelGGL((K<CH, RB, RS, NT>), dim3(NB), dim3(BLK), 0, 0, W, k, ld, rK, tK, pt, pv); }, reps)); \
    }
    RUN_V(k_v2, 512, 512, 16, true, 8192)
    {
        double* chk;
        CK(hipMalloc(&chk, 24 * 8));
        hipLaunchKernelGGL(k_check_butterfly, dim3(1), dim3(64), 0, 0, chk);
        double hc[24];
        CK(hipMemcpy(hc, chk, sizeof hc, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int j = 0; j < 16; ++j) ok = ok && hc[j] == 100.0 * (63 * 64 / 2) + 64.0 * j;
        for (int j = 0; j < 8; ++j) ok = ok && hc[16 + j] == 100.0 * (63 * 64 / 2) + 64.0 * j;
        printf("butterfly reduction check: %s\n", ok ? "ok" : "WRONG");
    }
#define RUN_V3(NP, RB, RS, PF, NB)                                                                                            \
    {                                                                                                                         \
        char nm[96];                                                                                                          \
        snprintf(nm, sizeof nm, "k_v3 NP=%d RB=%d RS=%d %s, %d blocks", NP, RB, RS, PF ? "prefetch" : "no-pf", NB);           \
        report(nm, time_it([&] { hipLaunchKernelGGL((k_v3<NP, RB, RS, PF>), dim3(NB), dim3(BLK), 0, 0, W, k, ld, rK, tK, pt, pv); }, reps)); \
    }
#define RUN_V4(NP, RB, RS, PF, NB)                                                                                            \
    {                                                                                                                         \
        char nm[96];                                                                                                          \
        snprintf(nm, sizeof nm, "k_v4 NP=%d RB=%d RS=%d %s, %d blocks", NP, RB, RS, PF ? "prefetch" : "no-pf", NB);           \
        report(nm, time_it([&] { hipLaunchKernelGGL((k_v4<NP, RB, RS, PF>), dim3(NB), dim3(BLK), 0, 0, W, k, ld, rK, tK, pt, pv); }, reps)); \
    }
    RUN_V4(2, 256, 8, true, 4096)
#define RUN_V5(CH, RB, RS, NB)                                                                                                \
    {                                                                                                                         \
        char nm[96];                                                                                                          \
        snprintf(nm, sizeof nm, "k_v5 CH=%d RB=%d RS=%d prefetch, %d blocks", CH, RB, RS, NB);                                \
        report(nm, time_it([&] { hipLaunchKernelGGL((k_v5<CH, RB, RS>), dim3(NB), dim3(BLK), 0, 0, W, k, ld, rK, tK, pt, pv); }, reps)); \
    }
    RUN_V5(512, 512, 8, 8192)
    RUN_V5(512, 512, 8, 1024)
    RUN_V5(512, 256, 8, 8192)
    RUN_V5(512, 128, 8, 8192)
    RUN_V5(512, 64, 8, 8192)
    RUN_V5(1024, 256, 8, 8192)
    RUN_V5(1024, 128, 8, 8192)
    RUN_V5(1024, 64, 8, 8192)
    RUN_V5(1024, 256, 4, 8192)
    RUN_V5(1024, 128, 4, 8192)
    return 0;
}
