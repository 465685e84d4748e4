// Latency probe for the single-workgroup iteration design (tools/experiments): dependent global loads (pointer chase)
// in a small / a large array, workgroup barriers, global atomics, as seen by ONE workgroup of 512 threads on an idle chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void __launch_bounds__(512) probe(const int* __restrict__ chase, int n_steps, int* stamps, double* out, int* sink) {
    __shared__ int s_x[512];
    const int tid = threadIdx.x;
    unsigned long long t0, t1;
    // (1) pointer chase by one thread
    int p = tid == 0 ? 0 : -1;
    __syncthreads();
    t0 = wall_clock64();
    if (tid == 0) for (int i = 0; i < n_steps; ++i) p = chase[p];
    t1 = wall_clock64();
    if (tid == 0) { out[0] = (double)(t1 - t0) * 10.0 / n_steps; sink[0] = p; }
    // (2) pointer chase by all 512 threads, each its own chain (memory-level parallelism across lanes)
    int q = tid;
    __syncthreads();
    t0 = wall_clock64();
    for (int i = 0; i < n_steps; ++i) q = chase[q];
    t1 = wall_clock64();
    if (tid == 0) out[1] = (double)(t1 - t0) * 10.0 / n_steps;
    sink[1 + tid] = q;
    // (3) __syncthreads round
    __syncthreads();
    t0 = wall_clock64();
    for (int i = 0; i < 256; ++i) { s_x[tid] = i; __syncthreads(); }
    t1 = wall_clock64();
    if (tid == 0) out[2] = (double)(t1 - t0) * 10.0 / 256;
    // (4) global atomicExch round trip (one thread, dependent)
    int a = 0;
    __syncthreads();
    t0 = wall_clock64();
    if (tid == 0) for (int i = 0; i < 256; ++i) a = atomicExch(&stamps[(a + i) & 1023], i + 1);
    t1 = wall_clock64();
    if (tid == 0) { out[3] = (double)(t1 - t0) * 10.0 / 256; sink[600] = a; }
    // (5) store then barrier then load by another thread (the hand-over pattern through global memory)
    __syncthreads();
    t0 = wall_clock64();
    int acc = 0;
    for (int i = 0; i < 128; ++i) {
        stamps[2048 + tid] = i + tid;
        __syncthreads();
        acc += stamps[2048 + ((tid + 64) & 511)];
        __syncthreads();
    }
    t1 = wall_clock64();
    if (tid == 0) out[4] = (double)(t1 - t0) * 10.0 / 128;
    sink[700 + tid] = acc;
}

int main() {
    for (int pass = 0; pass < 2; ++pass) {
        const size_t n = pass == 0 ? (1u << 16) : (1u << 26);  // 256 KB (L2) / 256 MB (beyond L2 and most of the MALL)
        std::vector<int> perm(n), chase(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::mt19937_64 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (size_t i = 0; i < n; ++i) chase[perm[i]] = perm[(i + 1) % n];
        int *d_chase, *d_stamps, *d_sink; double* d_out;
        hipMalloc(&d_chase, n * sizeof(int)); hipMalloc(&d_stamps, 4096 * sizeof(int)); hipMalloc(&d_sink, 2048 * sizeof(int)); hipMalloc(&d_out, 8 * sizeof(double));
        hipMemcpy(d_chase, chase.data(), n * sizeof(int), hipMemcpyHostToDevice);
        hipMemset(d_stamps, 0, 4096 * sizeof(int));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d_chase, 2000, d_stamps, d_out, d_sink);
        double out[8];
        hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
        printf("array %zu MB: dependent load, 1 thread %.0f ns | 512 threads %.0f ns | __syncthreads %.0f ns | atomicExch round trip %.0f ns | store+barrier+load+barrier %.0f ns\n",
               n * 4 >> 20, out[0], out[1], out[2], out[3], out[4]);
        hipFree(d_chase); hipFree(d_stamps); hipFree(d_sink); hipFree(d_out);
    }
    return 0;
}
