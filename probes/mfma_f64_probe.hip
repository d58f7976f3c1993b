// Probe (round 1): verifies the register-resident "D-layout" tile algebra used by the PDP
// Riccati kernels on gfx950 and measures v_mfma_f64_16x16x4_f64 issue/latency cycles.
//   tile = 16x16 row-major fp64 matrix, lane l holds elements flat[64*r + l], r = 0..3
//   (row = (l>>4) + 4r, col = l&15) == the C/D layout of v_mfma_f64_16x16x4_f64.
//   Claim under test: sum_c mfma(X.r[c], Y.r[c], D) == D + X^T * Y.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ d4 mma_tn(const d4 x, const d4 y, d4 c) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], c, 0, 0, 0);
    return c;
}

__global__ void k_layout(const double* X, const double* Y, const double* C, double* D) {
    int l = threadIdx.x;
    d4 x, y, c;
    for (int r = 0; r < 4; ++r) { x[r] = X[64 * r + l]; y[r] = Y[64 * r + l]; c[r] = C[64 * r + l]; }
    d4 d = mma_tn(x, y, c);
    for (int r = 0; r < 4; ++r) D[64 * r + l] = d[r];
}

// dependent chain of MFMAs (accumulator chained)
__global__ void k_mfma_dep(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c[0] + c[1] + c[2] + c[3];
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// A/B operand depends on previous result (X^T Y chains: result feeds next as operand)
__global__ void k_mfma_dep_ab(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    d4 c = {1e-3, 2e-3, 3e-3, 4e-3};
    d4 z = {0, 0, 0, 0};
    double b = 1e-2;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        d4 d = z;
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(c[0], b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(c[1], b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(c[2], b, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(c[3], b, d, 0, 0, 0);
        c = d;
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c[0] + c[1] + c[2] + c[3];
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// 4 independent accumulators
__global__ void k_mfma_indep(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// dependent fp64 FMA chain and 4 independent chains
__global__ void k_fma_dep(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1e-9, c = 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c = __builtin_fma(c, a, b); c = __builtin_fma(c, a, b); c = __builtin_fma(c, a, b); c = __builtin_fma(c, a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_indep(double* out, int iters, long long* cyc) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1e-9, c0 = 0.5, c1 = 0.6, c2 = 0.7, c3 = 0.8;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_fma(c0, a, b); c1 = __builtin_fma(c1, a, b); c2 = __builtin_fma(c2, a, b); c3 = __builtin_fma(c3, a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c0 + c1 + c2 + c3;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent LDS read chain (pointer chase) f64
__global__ void k_lds_dep(double* out, int iters, long long* cyc) {
    __shared__ int nxt[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) nxt[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    int p = l;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) { p = nxt[p]; p = nxt[p]; p = nxt[p]; p = nxt[p]; }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = p;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
static void timeit(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_wave) {
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * 64 * blocks)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    double tf = flop_per_iter_per_wave * iters * blocks / (ms * 1e-3) / 1e12;
    printf("%-22s blocks=%5d iters=%d  ms=%.3f  clk64/iter(4 ops)=%.1f  -> %.1f clk/op ; %.2f TFLOP/s\n", name, blocks, iters, ms,
           (double)h / iters, (double)h / iters / 4.0, tf);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s arch=%s CUs=%d clock=%d kHz smemPerBlock=%zu maxSmemPerMP=%zu regsPerBlock=%d l2=%d wave=%d mem=%.1f GB\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor,
           p.regsPerBlock, p.l2CacheSize, p.warpSize, p.totalGlobalMem / 1e9);
    // ---- layout test with asymmetric matrices
    std::vector<double> X(256), Y(256), C(256), D(256), R(256);
    srand(1);
    for (int i = 0; i < 256; ++i) { X[i] = (rand() % 2001 - 1000) / 1000.0; Y[i] = (rand() % 2001 - 1000) / 700.0; C[i] = (rand() % 2001 - 1000) / 300.0; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = C[i * 16 + j];
        for (int k = 0; k < 16; ++k) s = fma(X[k * 16 + i], Y[k * 16 + j], s);   // (X^T Y)[i][j], k ascending
        R[i * 16 + j] = s;
    }
    double *dX, *dY, *dC, *dD;
    CK(hipMalloc(&dX, 2048)); CK(hipMalloc(&dY, 2048)); CK(hipMalloc(&dC, 2048)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dX, X.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Y.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dX, dY, dC, dD);
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    double md = 0, mt = 0; int bitexact = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        md = fmax(md, fabs(D[i * 16 + j] - R[i * 16 + j])); mt = fmax(mt, fabs(D[i * 16 + j] - R[j * 16 + i]));
        bitexact += (D[i * 16 + j] == R[i * 16 + j]);
    }
    printf("layout: max|D - (C + X^T Y)| = %.3e   (vs transposed ref: %.3e)  bit-exact elems vs k-ascending fma chain: %d/256\n", md, mt, bitexact);
    printf("LAYOUT_%s\n", md < 1e-12 ? "OK" : "FAIL");
    // ---- timing
    int iters = 20000;
    timeit("mfma_f64 dep(C)  1wave", k_mfma_dep, 1, iters, 4 * 2048.0);
    timeit("mfma_f64 dep(AB) 1wave", k_mfma_dep_ab, 1, iters, 4 * 2048.0);
    timeit("mfma_f64 indep   1wave", k_mfma_indep, 1, iters, 4 * 2048.0);
    timeit("mfma_f64 indep 1024w", k_mfma_indep, 1024, iters, 4 * 2048.0);
    timeit("mfma_f64 indep 4096w", k_mfma_indep, 4096, iters, 4 * 2048.0);
    timeit("mfma_f64 dep   4096w", k_mfma_dep, 4096, iters, 4 * 2048.0);
    timeit("fma_f64 dep    1wave", k_fma_dep, 1, iters, 4 * 128.0);
    timeit("fma_f64 indep  1wave", k_fma_indep, 1, iters, 4 * 128.0);
    timeit("fma_f64 indep 8192w", k_fma_indep, 8192, iters, 4 * 128.0);
    timeit("lds dep chain  1wave", k_lds_dep, 1, iters, 0.0);
    return 0;
}
