// Probe 7: does an fp64 VALU instruction get cheaper when only some lanes of the wavefront are enabled (EXEC mask)?  The rollout of the fused
// OC kernels is a scalar recursion that all 64 lanes execute redundantly; if the SIMD skipped disabled 16-lane quarters, running it on a
// quarter (or on one lane) would be faster.  One wavefront, chains of v_fma_f64: 1 dependent chain, 4 and 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int CH>
__global__ void k(double* out, int iters, int lanes, long long* cyc) {
    const int l = threadIdx.x;
    double a[CH];
    for (int c = 0; c < CH; ++c) a[c] = 1.0 + 1e-3 * (l + c);
    const double b = 1.0 - 1e-9 * l, d = 1e-7;
    long long t0 = 0, t1 = 0;
    if (l < lanes) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < CH; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d));
        }
        t1 = __builtin_readcyclecounter();
    }
    double s = 0; for (int c = 0; c < CH; ++c) s += a[c];
    out[l] = s;
    if (l == 0) cyc[0] = t1 - t0;
}

template <int CH>
static void run(const char* name) {
    double* out; long long* cyc; CK(hipMalloc(&out, 8 * 64)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    for (int lanes : {64, 32, 16, 1}) {
        hipLaunchKernelGGL(k<CH>, dim3(1), dim3(64), 0, 0, out, iters, lanes, cyc);
        hipLaunchKernelGGL(k<CH>, dim3(1), dim3(64), 0, 0, out, iters, lanes, cyc);
        CK(hipDeviceSynchronize());
        long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("%-22s lanes enabled %2d : %6.2f cycles per v_fma_f64\n", name, lanes, (double)h / ((double)iters * 8 * CH));
    }
}
int main() { run<1>("1 dependent chain"); run<4>("4 independent chains"); run<8>("8 independent chains"); return 0; }
