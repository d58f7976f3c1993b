// Probe 3: can NON-fp64 instructions issue in the shadow of a dependent chain of fp64 MFMAs inside one wave?
// Each iteration: 4 chained v_mfma_f64_16x16x4 with K filler instructions placed between consecutive MFMAs in program order
// (asm volatile to pin the order).  Fillers: v_add_u32 (int VALU), v_mov_b32, ds_read_b64, s_add_u32 (SALU), v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int KIND, int K>
__global__ void k_mix(double* out, int iters, long long* cyc) {
    __shared__ double lds[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = i * 1e-6;
    __syncthreads();
    d4 c = {0, 0, 0, 0};
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    int vi = l, vj = 3;
    float vf = 1.0f + l, vg = 0.5f;
    int sacc = 0;
    double dsum = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vi) : "v"(vj));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(vf) : "v"(vg));
                if (KIND == 2) { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((l * 8 + k * 512) & 8191)); asm volatile("" :: "v"(t)); }
                if (KIND == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dsum) : "v"(b));
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = c[0] + c[1] + c[2] + c[3] + vi + vf + sacc + dsum;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename Kn>
static void run(const char* name, Kn kern, int K) {
    double* out; long long* cyc;
    int iters = 5000;
    CK(hipMalloc(&out, 8 * 64)); CK(hipMalloc(&cyc, 8));
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-12s K=%2d fillers per MFMA : %7.1f cycles per 4-MFMA chain (256 = fully hidden)  -> %.1f cycles per filler beyond MFMA time\n", name, K,
           (double)h / iters, K ? ((double)h / iters - 256.0) / (4.0 * K) : 0.0);
    CK(hipFree(out)); CK(hipFree(cyc));
}
#define RUNK(name, KIND) run(name, k_mix<KIND, 0>, 0); run(name, k_mix<KIND, 4>, 4); run(name, k_mix<KIND, 8>, 8); run(name, k_mix<KIND, 12>, 12); run(name, k_mix<KIND, 16>, 16);
int main() {
    RUNK("v_add_u32", 0)
    RUNK("v_fma_f32", 1)
    RUNK("ds_read_b64", 2)
    RUNK("s_add_u32", 3)
    RUNK("v_fma_f64", 4)
    return 0;
}
