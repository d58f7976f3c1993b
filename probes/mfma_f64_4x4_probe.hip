// Layout and timing of v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A (lane la) x one-hot B (lane lb) -> which D lanes light up.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void onehot(double* out) {
    const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0, c = 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    out[((size_t)la * 64 + lb) * 64 + lane] = d;
}
__global__ void timing(double* out, long long* cyc, int iters) {
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002, c = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    // dependent on A (result fed back as the A operand)
    double e = c * 1e-30;
    long long t2 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        e = __builtin_amdgcn_mfma_f64_4x4x4f64(e, b, 0.0, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_4x4x4f64(e, b, 0.0, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_4x4x4f64(e, b, 0.0, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_4x4x4f64(e, b, 0.0, 0, 0, 0);
    }
    long long t3 = __builtin_readcyclecounter();
    out[threadIdx.x] = c + e;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
    onehot<<<dim3(64, 64), 64>>>(d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // print for every (la, lb) the lit lanes, compactly: "la lb : lanes..."
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        bool any = false;
        for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) any = true;
        if (!any) continue;
        printf("%d %d :", la, lb);
        for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    long long* c; hipMalloc(&c, 16); double* o; hipMalloc(&o, 512);
    timing<<<1, 64>>>(o, c, 1000); timing<<<1, 64>>>(o, c, 1000);
    long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    printf("TIMING 4x4x4: dependent-on-C %.1f cycles/op, dependent-on-A %.1f cycles/op\n", hc[0] / 4000.0, hc[1] / 4000.0);
    return 0;
}
