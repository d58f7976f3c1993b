// Probe 5: do TWO wavefronts on one SIMD overlap where one cannot?  ("whether wave B's MFMA overlaps wave A's
// int-VALU / LDS / SALU / wait time is untested".)
// Every wave runs `iters` x [4 chained v_mfma_f64_16x16x4 + 4 x K filler instructions of one kind].  Grid = 1024 single-wave workgroups
// (one per SIMD of the 256 CUs) against 2048 and 4096 (two / four per SIMD; the kernel needs 24 VGPRs and 8 KB of LDS, so all are
// resident at once).  If the fillers of one wave hide under the MFMAs of the other, the 2048-wave launch takes as long as the
// 1024-wave launch; with no overlap it takes twice as long.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int KIND, int K>
__global__ void __launch_bounds__(64) k_mix(double* out, int iters) {
    __shared__ double lds[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = i * 1e-6;
    __syncthreads();
    d4 c = {0, 0, 0, 0};
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    int vi = l, vj = 3;
    int sacc = 0;
    double dsum = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vi) : "v"(vj));
                if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dsum) : "v"(b));
                if (KIND == 2) { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((l * 8 + k * 512) & 8191)); asm volatile("" :: "v"(t)); }
                if (KIND == 3) asm volatile("s_sleep 1");
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    out[(size_t)blockIdx.x * 64 + l] = c[0] + c[1] + c[2] + c[3] + vi + sacc + dsum;
}

template <typename Kn>
static float time_grid(Kn kern, int grid, double* out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}
template <typename Kn>
static void run(const char* name, Kn kern, int K, double* out) {
    const int iters = 4000;
    float t1 = time_grid(kern, 1024, out, iters), t2 = time_grid(kern, 2048, out, iters), t4 = time_grid(kern, 4096, out, iters);
    printf("%-12s K=%2d fillers/MFMA : 1 wave/SIMD %7.3f ms | 2 waves/SIMD %7.3f ms (x%.2f) | 4 waves/SIMD %7.3f ms (x%.2f)   [x1 = full overlap, x2 / x4 = none]\n",
           name, K, t1, t2, t2 / t1, t4, t4 / t1);
}
#define RUNK(name, KIND) run(name, k_mix<KIND, 0>, 0, out); run(name, k_mix<KIND, 4>, 4, out); run(name, k_mix<KIND, 8>, 8, out); run(name, k_mix<KIND, 16>, 16, out);
int main() {
    double* out;
    CK(hipMalloc(&out, 8 * 64 * 4096));
    RUNK("v_add_u32", 0)
    RUNK("v_fma_f64", 1)
    RUNK("ds_read_b64", 2)
    RUNK("s_sleep", 3)
    return 0;
}
