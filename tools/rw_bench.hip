// rw_bench.hip — ceiling of an in-place read + write pass over the dense nucleus inverse W on this box: what the fold of the
// delayed-update mode (k_fold_w2: W0 += U V^T, 16 k^2 algorithmic bytes) and with it the blocked re-inversion (inverse.inc)
// can reach at best.  Variants: tile shapes of 16-byte non-temporal loads / stores, with and without a register prefetch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rw_bench.hip -o tools/rw_bench
// Run  : tools/rw_bench [k] [ld] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

typedef double dbl2_t __attribute__((ext_vector_type(2)));
constexpr int BLK = 256;

// (a) the fold's own access shape: a block owns CH = 256 columns x RB rows, a thread one column, RS rows per step, next step prefetched
template <int RB, int RS, bool NT>
__global__ void __launch_bounds__(BLK) k_rw_col(double* __restrict__ W, int k, int ld, double c) {
    const int nch = (k + BLK - 1) / BLK, nstr = (k + RB - 1) / RB;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += gridDim.x) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        const int col = chunk * BLK + threadIdx.x;
        if (col >= k) continue;
        double* wcol = W + col;
        double w[RS], wn[RS];
#pragma unroll
        for (int a = 0; a < RS; ++a) {
            const double* p = wcol + (size_t)min(rbeg + a, rend - 1) * ld;
            w[a] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        for (int r0 = rbeg; r0 < rend; r0 += RS) {
            if (r0 + RS < rend) {
#pragma unroll
                for (int a = 0; a < RS; ++a) {
                    const double* p = wcol + (size_t)min(r0 + RS + a, rend - 1) * ld;
                    wn[a] = NT ? __builtin_nontemporal_load(p) : *p;
                }
            }
#pragma unroll
            for (int a = 0; a < RS; ++a)
                if (r0 + a < rend) {
                    double* p = wcol + (size_t)(r0 + a) * ld;
                    if (NT) __builtin_nontemporal_store(w[a] * c + 1.0, p);
                    else *p = w[a] * c + 1.0;
                }
#pragma unroll
            for (int a = 0; a < RS; ++a) w[a] = wn[a];
        }
    }
}
// (b) 16 bytes per lane: a block owns 512 columns x RB rows, a thread two adjacent columns
template <int RB, int RS, bool NT>
__global__ void __launch_bounds__(BLK) k_rw_pair(double* __restrict__ W, int k, int ld, double c) {
    const int nch = (k + 2 * BLK - 1) / (2 * BLK), nstr = (k + RB - 1) / RB;
    for (int tile = blockIdx.x; tile < nstr * nch; tile += gridDim.x) {
        const int strip = tile / nch, chunk = tile % nch;
        const int rbeg = strip * RB, rend = min(k, rbeg + RB);
        const int col = chunk * 2 * BLK + 2 * threadIdx.x;
        if (col + 1 >= k) continue;
        double* wcol = W + col;
        dbl2_t w[RS], wn[RS];
#pragma unroll
        for (int a = 0; a < RS; ++a) {
            const dbl2_t* p = (const dbl2_t*)(wcol + (size_t)min(rbeg + a, rend - 1) * ld);
            w[a] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        for (int r0 = rbeg; r0 < rend; r0 += RS) {
            if (r0 + RS < rend) {
#pragma unroll
                for (int a = 0; a < RS; ++a) {
                    const dbl2_t* p = (const dbl2_t*)(wcol + (size_t)min(r0 + RS + a, rend - 1) * ld);
                    wn[a] = NT ? __builtin_nontemporal_load(p) : *p;
                }
            }
#pragma unroll
            for (int a = 0; a < RS; ++a)
                if (r0 + a < rend) {
                    dbl2_t* p = (dbl2_t*)(wcol + (size_t)(r0 + a) * ld);
                    dbl2_t x = {w[a].x * c + 1.0, w[a].y * c + 1.0};
                    if (NT) __builtin_nontemporal_store(x, p);
                    else *p = x;
                }
#pragma unroll
            for (int a = 0; a < RS; ++a) w[a] = wn[a];
        }
    }
}

template <typename F>
static void run(const char* name, F launch, int k, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = 16.0 * k * (double)k;
    printf("%-64s %8.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
}

int main(int argc, char** argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 20480;
    const int ld = argc > 2 ? atoi(argv[2]) : k + 16;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    double* W;
    CK(hipMalloc(&W, sizeof(double) * (size_t)k * ld));
    CK(hipMemset(W, 0, sizeof(double) * (size_t)k * ld));
    printf("k = %d, ld = %d, W = %.2f GB, algorithmic bytes per in-place pass = %.2f GB (read + write)\n", k, ld, 8e-9 * k * ld, 16e-9 * k * (double)k);
    for (int nb : {2048, 4096, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "one column / lane, 256 x 256 tiles, 8 rows / step, NT, %d blocks", nb);
        run(nm, [&] { hipLaunchKernelGGL((k_rw_col<256, 8, true>), dim3(nb), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    }
    run("one column / lane, 256 x 256 tiles, 8 rows / step, plain, 8192", [&] { hipLaunchKernelGGL((k_rw_col<256, 8, false>), dim3(8192), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    run("one column / lane, 256 x 128 tiles, 4 rows / step, NT, 8192", [&] { hipLaunchKernelGGL((k_rw_col<128, 4, true>), dim3(8192), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    run("one column / lane, 256 x 512 tiles, 16 rows / step, NT, 8192", [&] { hipLaunchKernelGGL((k_rw_col<512, 16, true>), dim3(8192), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    for (int nb : {2048, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "two columns / lane (16 B), 512 x 128 tiles, 4 rows / step, NT, %d blocks", nb);
        run(nm, [&] { hipLaunchKernelGGL((k_rw_pair<128, 4, true>), dim3(nb), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    }
    run("two columns / lane (16 B), 512 x 256 tiles, 8 rows / step, NT, 8192", [&] { hipLaunchKernelGGL((k_rw_pair<256, 8, true>), dim3(8192), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    run("two columns / lane (16 B), 512 x 256 tiles, 8 rows / step, plain, 8192", [&] { hipLaunchKernelGGL((k_rw_pair<256, 8, false>), dim3(8192), dim3(BLK), 0, 0, W, k, ld, 0.5); }, k, reps);
    CK(hipFree(W));
    return 0;
}
