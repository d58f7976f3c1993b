// Stand-in for RCCL's collective kernel in scheduling experiments on ONE GPU (probes/exchange_policies.py).
// A one-rank RCCL group turns all_gather_into_tensor into a device-to-device copy, which needs no compute unit; the kernel an N-rank group launches does: a few
// workgroups (one per channel), each holding a CU slot, some LDS and registers for as long as the exchange lasts (tens of microseconds for 80 KB over xGMI).  This
// kernel has that footprint and nothing else: `nwg` workgroups of 256 threads with `lds_bytes` of dynamic LDS spin for `us` microseconds on the 100 MHz wall clock,
// then copy `n` doubles (the rows) from src to dst so that the ordering of the pipeline can be checked on the data.
// The footprint is RCCL's: the gfx950 code object of librccl.so 2.26.6 (llvm-readelf --notes) holds 126 kernels, 114 of them with 4.7 - 21 KB of static LDS per workgroup
// (rcclGenericKernel<1 | 2 | 4>: 19,744 bytes, max_flat_workgroup_size 256, 261 - 280 registers incl. AGPRs); oc_pdp_fused3_kernel<quadrotor, 4> leaves 0 bytes of a CU's LDS free.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void standin_collective_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, int ticks) {
    extern __shared__ double lds[];
    const uint64_t t0 = wall_clock64();
    lds[threadIdx.x] = (double)threadIdx.x;
    while ((int64_t)(wall_clock64() - t0) < (int64_t)ticks) __builtin_amdgcn_s_sleep(8);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i] + 0.0 * lds[threadIdx.x];
}

extern "C" int standin_collective(void* stream, const double* src, double* dst, int n, int nwg, int lds_bytes, int us) {
    hipLaunchKernelGGL(standin_collective_kernel, dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, src, dst, n, us * 100);
    return (int)hipGetLastError();
}
