// Probe 6: where do the wavefronts of a workgroup land?  Each wave records HW_REG_HW_ID (gfx9 layout: wave_id[3:0] simd_id[5:4] pipe[7:6]
// cu_id[11:8] sh[12] se[15:13] ...) and XCC_ID.  Question: in a 512-thread workgroup (8 waves), do waves w and w + 4 share a SIMD, so that
// a "runner" wave and its "evaluator" wave can be paired on one SIMD by construction?  Also 128-thread workgroups (2 waves), 4 per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_where(unsigned* out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep every wave resident long enough that the whole grid is on the chip at once
    double a = threadIdx.x * 1e-3;
    for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
    lds[threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}

static void run(int threads, int blocks, size_t ldsb) {
    const int wpb = threads / 64, nw = blocks * wpb;
    unsigned* d; CK(hipMalloc(&d, 8 * nw));
    CK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(threads), ldsb, 0, d, 20000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * nw); CK(hipMemcpy(h.data(), d, 8 * nw, hipMemcpyDeviceToHost));
    printf("== %d threads per workgroup, %d workgroups, %zu KB LDS each\n", threads, blocks, ldsb / 1024);
    for (int b = 0; b < 3; ++b) {
        printf("  workgroup %d:", b);
        for (int w = 0; w < wpb; ++w) { unsigned v = h[2 * (b * wpb + w)]; printf("  w%d->simd%u slot%u cu%u se%u xcc%u", w, (v >> 4) & 3, v & 15, (v >> 8) & 15, (v >> 13) & 7, h[2 * (b * wpb + w) + 1] & 15); }
        printf("\n");
    }
    // statistics: per workgroup, the SIMD pattern of its waves; per (xcc,se,sh,cu,simd): how many waves
    std::map<std::vector<int>, int> pat; std::map<unsigned long long, int> per_simd; int pair_ok = 0, pairs = 0;
    for (int b = 0; b < blocks; ++b) {
        std::vector<int> p;
        for (int w = 0; w < wpb; ++w) {
            unsigned v = h[2 * (b * wpb + w)], x = h[2 * (b * wpb + w) + 1] & 15;
            p.push_back((v >> 4) & 3);
            unsigned long long key = ((unsigned long long)x << 32) | (v & 0xFFF0u & ~0xC0u);      // xcc | se sh cu simd (pipe bits masked)
            per_simd[key]++;
        }
        pat[p]++;
        if (wpb == 8) for (int w = 0; w < 4; ++w) { pairs++; pair_ok += p[w] == p[w + 4]; }
        if (wpb == 2) { pairs++; pair_ok += p[0] == p[1]; }
    }
    for (auto& kv : pat) { printf("  SIMD pattern"); for (int s : kv.first) printf(" %d", s); printf(" : %d workgroups\n", kv.second); }
    std::map<int, int> hist; for (auto& kv : per_simd) hist[kv.second]++;
    for (auto& kv : hist) printf("  SIMDs hosting %d waves: %d\n", kv.first, kv.second);
    if (pairs) printf("  wave pairs (w, w + %d) on the same SIMD: %d of %d\n", wpb / 2, pair_ok, pairs);
    CK(hipFree(d));
}

int main() {
    run(512, 256, 160 * 1024);
    run(512, 512, 160 * 1024);
    run(128, 1024, 40 * 1024);
    run(64, 1024, 40 * 1024);
    return 0;
}
