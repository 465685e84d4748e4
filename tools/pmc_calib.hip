// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the simplex
// kernels use (MI355X_MICROARCH.md §HBM: only 16-B/lane streams are calibrated there).  Each kernel
// moves a KNOWN number of bytes from a buffer far larger than the 256 MB Infinity Cache; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- ./pmc_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- ./pmc_calib
// and compare the per-dispatch counter with the byte count printed here (tools/pmc_calib.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) calib_read_i32(const int* p, size_t n, int* out) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x7fffffff) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_read_f64(const double* p, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_read_f64x2(const double2* p, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double2 t = p[i];
        acc += t.x + t.y;
    }
    if (acc == 1.2345e300) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_write_f64(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.0;
}
__global__ void __launch_bounds__(256) calib_write_f64x2(double2* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_double2(1.0, 2.0);
}
// the sweep's pattern: a 12-byte-per-entry stream (4-B index + 8-B value) and a 16-byte gather from a
// 1.6 MB table (100 000 rows) that stays in L2
__global__ void __launch_bounds__(256) calib_stream_gather(const int* idx, const double* val, const double2* table, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double2 t = table[idx[i]];
        acc += val[i] * (t.x + t.y);
    }
    if (acc == 1.2345e300) out[0] = acc;
}

int main() {
    const size_t BYTES = (size_t)2 << 30;  // 2 GiB per stream, 8x the Infinity Cache
    void *a = nullptr, *b = nullptr, *tab = nullptr, *out = nullptr;
    CK(hipMalloc(&a, BYTES));
    CK(hipMalloc(&b, BYTES));
    CK(hipMalloc(&tab, 100000 * 16));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, BYTES));
    CK(hipMemset(b, 0, BYTES));
    CK(hipMemset(tab, 0, 100000 * 16));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        calib_read_i32<<<grid, 256>>>((const int*)a, BYTES / 4, (int*)out);
        calib_read_f64<<<grid, 256>>>((const double*)a, BYTES / 8, (double*)out);
        calib_read_f64x2<<<grid, 256>>>((const double2*)a, BYTES / 16, (double*)out);
        calib_write_f64<<<grid, 256>>>((double*)b, BYTES / 8);
        calib_write_f64x2<<<grid, 256>>>((double2*)b, BYTES / 16);
        // 128 Mi entries: 0.5 GiB of indices (all zero -> row 0.. fine: hits L2) + 1 GiB of values
        calib_stream_gather<<<grid, 256>>>((const int*)a, (const double*)b, (const double2*)tab, (size_t)128 << 20, (double*)out);
    }
    CK(hipDeviceSynchronize());
    printf("calib_read_i32 read_bytes=%zu\ncalib_read_f64 read_bytes=%zu\ncalib_read_f64x2 read_bytes=%zu\n", BYTES, BYTES, BYTES);
    printf("calib_write_f64 write_bytes=%zu\ncalib_write_f64x2 write_bytes=%zu\n", BYTES, BYTES);
    printf("calib_stream_gather read_bytes=%zu (stream only; the 16-B gathers hit one L2 line)\n", ((size_t)128 << 20) * 12);
    return 0;
}
