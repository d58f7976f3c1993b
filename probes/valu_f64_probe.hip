// Probe 2 (round 1): lone-wave fp64 VALU issue rate, fp64 divide cost, LDS broadcast reads,
// and MFMA||VALU overlap inside ONE wave on gfx950.  Numbers feed DESIGN.md's cycle budget.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// 16 independent chains, 64 FMAs per iteration
__global__ void k_fma16(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1e-9;
    double c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.5 + 0.01 * i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_fma(c[i], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// 64 dependent FMAs per iteration
__global__ void k_fma_dep64(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1e-9, c = 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) c = __builtin_fma(c, a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + l] = c;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// f32 16 independent chains for comparison
__global__ void k_fma16_f32(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    float a = 1.0f + 1e-6f * l, b = 1e-6f;
    float c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.5f + 0.01f * i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_fmaf(c[i], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// 16 dependent divides per iteration (counted as 64 "ops" -> divide by 16 on host)
__global__ void k_div(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    double a = 1.0000001 + 1e-9 * l, c = 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c = a / (c + 1.0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + l] = c;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// 64 independent LDS broadcast reads (uniform address) of f64 per iteration, summed
__global__ void k_lds_bcast(double* out, int iters, long long* cyc) {
    __shared__ double buf[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += blockDim.x) buf[i] = 1e-6 * i;
    __syncthreads();
    double s = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int base = (it & 7) * 64;
#pragma unroll
        for (int i = 0; i < 64; ++i) s += buf[base + i];
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// 64 independent per-lane LDS reads (lane-consecutive addresses) per iteration
__global__ void k_lds_lane(double* out, int iters, long long* cyc) {
    __shared__ double buf[64 * 72];
    int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 72; i += blockDim.x) buf[i] = 1e-6 * i;
    __syncthreads();
    double s = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int base = (it & 7) * 64;
#pragma unroll
        for (int i = 0; i < 64; ++i) s += buf[base + i * 64 + l];
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// MFMA (4 per iter, chained accumulators x2) interleaved with 32 independent FMAs in the SAME wave
template <int NFMA>
__global__ void k_overlap(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1e-9;
    double c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.5 + 0.01 * i;
    d4 m0 = {0, 0, 0, 0}, m1 = m0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q & 1) m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m1, 0, 0, 0);
            else m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NFMA / 4; ++i) c[(q * (NFMA / 4) + i) & 15] = __builtin_fma(c[(q * (NFMA / 4) + i) & 15], a, b);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = m0[0] + m0[1] + m1[2] + m1[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
static void timeit(const char* name, K kern, int blocks, int threads, int iters, int ops_per_iter) {
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * threads * blocks)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    double wall_cyc_per_iter = ms * 1e-3 * 2.4e9 / iters;
    printf("%-26s blocks=%5d thr=%4d  ms=%8.3f  wave0 clk/iter=%8.1f (%.2f clk/op)  wall clk/iter=%8.1f (%.2f clk/op)\n", name, blocks, threads, ms,
           (double)h / iters, (double)h / iters / ops_per_iter, wall_cyc_per_iter, wall_cyc_per_iter / ops_per_iter);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    int it = 4000;
    timeit("fma_f64 x16 indep", k_fma16, 1, 64, it, 64);
    timeit("fma_f64 x16 indep", k_fma16, 1024, 64, it, 64);
    timeit("fma_f64 x16 indep", k_fma16, 1024, 128, it, 64);
    timeit("fma_f64 x16 indep", k_fma16, 1024, 256, it, 64);
    timeit("fma_f64 x16 indep", k_fma16, 1024, 512, it, 64);
    timeit("fma_f64 dep64", k_fma_dep64, 1, 64, it, 64);
    timeit("fma_f32 x16 indep", k_fma16_f32, 1, 64, it, 64);
    timeit("div_f64 dep x16", k_div, 1, 64, it, 16);
    timeit("lds bcast f64 x64", k_lds_bcast, 1, 64, it, 64);
    timeit("lds bcast f64 x64", k_lds_bcast, 1024, 256, it, 64);
    timeit("lds lane  f64 x64", k_lds_lane, 1, 64, it, 64);
    timeit("lds lane  f64 x64", k_lds_lane, 1024, 256, it, 64);
    timeit("mfma x4 + 0 fma", k_overlap<0>, 1, 64, it, 4);
    timeit("mfma x4 + 16 fma", k_overlap<16>, 1, 64, it, 4);
    timeit("mfma x4 + 32 fma", k_overlap<32>, 1, 64, it, 4);
    timeit("mfma x4 + 64 fma", k_overlap<64>, 1, 64, it, 4);
    timeit("mfma x4 + 32 fma 2w/simd", k_overlap<32>, 1024, 128, it, 4);
    return 0;
}
