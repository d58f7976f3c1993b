// PMC calibration probe (round-4 verdict, evidence hygiene 9): FETCH_SIZE / WRITE_SIZE of rocprofv3 on gfx950 against KNOWN byte counts, in the access shapes
// this repository's kernels use.  MI355X_MICROARCH.md calibrates one case only - wide coalesced streaming reads of 16 B/lane report exactly 1/2 - and says
// "calibrate on a known byte count in your own access pattern before trusting an absolute".  The kernels here move a known number of bytes each:
//
//   read16      16 B / lane global_load_dwordx4, streaming                    (the guide's calibrated case: the control)
//   read8        8 B / lane global_load_dwordx2, streaming                    (how every kernel of csrc/ reads its inputs and matrices)
//   write8       8 B / lane range-checked raw buffer store, streaming          (how the fused kernels write x, lambda, gains, records: LQS_RSRC / rsX)
//   copy8_lds    8 B / lane load -> LDS -> 8 B / lane buffer store             (the streamer wave of lqr_solve_stream_kernel, the chunk hand-over of the solver)
//   rows712      one wave per "trajectory": 50 rows of 712 B written, then read back 3 rows ahead by the same wave (the gain scratch of oc_pdp_fused3_kernel:
//                712 B per stage, written by the backward sweep, re-read by the forward sweep)  - total bytes as given on the command line
//   rows712_big  the same with a footprint far beyond the 256 MB Infinity Cache
//
// Every kernel runs REPS times on a buffer of `MB` megabytes (default 1024: four times the Infinity Cache); stdout lists the true bytes read / written per
// dispatch.  probes/profile_r05.sh runs this binary under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and divides.
//     hipcc --offload-arch=gfx950 -O3 -o probes/pmc_calibrate probes/pmc_calibrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)

typedef double double2_ __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---- streaming reads: every workgroup of 256 threads walks its own contiguous slab; the sum goes to one word per workgroup (negligible writes)
extern "C" __global__ void __launch_bounds__(256) read16(const double2_* __restrict__ src, double* __restrict__ sink, int64_t n16_per_block) {
    const double2_* p = src + (int64_t)blockIdx.x * n16_per_block;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n16_per_block; i += 256) { double2_ v = __builtin_nontemporal_load(p + i); acc += v.x + v.y; }
    if (acc == 123.456) sink[blockIdx.x] = acc;
}

extern "C" __global__ void __launch_bounds__(256) read8(const double* __restrict__ src, double* __restrict__ sink, int64_t n8_per_block) {
    const double* p = src + (int64_t)blockIdx.x * n8_per_block;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n8_per_block; i += 256) acc += p[i];
    if (acc == 123.456) sink[blockIdx.x] = acc;
}

// ---- streaming range-checked buffer stores, 8 B / lane; the resource covers one slab of < 2 GB per workgroup
extern "C" __global__ void __launch_bounds__(256) write8(double* __restrict__ dst, int64_t n8_per_block) {
    double* p = dst + (int64_t)blockIdx.x * n8_per_block;
    const auto rs = RSRC(p, n8_per_block * 8);
    const double v = (double)blockIdx.x;
    u32x2 w = __builtin_bit_cast(u32x2, v);
    // the last iteration's lanes beyond the slab carry an out-of-range offset: dropped by the range check, as in the product kernels
    for (int64_t i = threadIdx.x; i < n8_per_block + 255; i += 256) __builtin_amdgcn_raw_buffer_store_b64(w, rs, (int)(i * 8), 0, 0);
}

extern "C" __global__ void __launch_bounds__(256) copy8_lds(const double* __restrict__ src, double* __restrict__ dst, int64_t n8_per_block) {
    __shared__ double stage[4][256];
    const double* p = src + (int64_t)blockIdx.x * n8_per_block;
    double* q = dst + (int64_t)blockIdx.x * n8_per_block;
    const auto rs = RSRC(q, n8_per_block * 8);
    for (int64_t i0 = 0; i0 < n8_per_block; i0 += 1024) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int64_t i = i0 + 256 * k + threadIdx.x; v[k] = p[i < n8_per_block ? i : 0]; }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) stage[k][threadIdx.x] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = i0 + 256 * k + threadIdx.x;
            const double o = stage[k][threadIdx.x ^ 1];                       // (a different lane's word: the LDS hop is real)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, i < n8_per_block ? (int)((i ^ 1) * 8) : 0x7fffffff, 0, 0);
        }
        __syncthreads();
    }
}

// ---- the gain scratch: one 64-lane workgroup per trajectory, `rows` rows of 89 doubles (712 B); backward sweep writes row t = rows-1 .. 0, forward sweep reads them
// back t = 0 .. rows-1, requested 3 rows ahead, exactly as oc_pdp_fused3_kernel does.  `work` dependent FMAs per row stand for the Riccati step.
extern "C" __global__ void __launch_bounds__(64) rows712(double* __restrict__ ws, double* __restrict__ sink, int rows, int work) {
    double* g = ws + (int64_t)blockIdx.x * rows * 89;
    const auto rs = RSRC(g, (int64_t)rows * 712);
    const int lane = threadIdx.x;
    double a = 1.0 + lane;
    for (int t = rows - 1; t >= 0; --t) {
        for (int k = 0; k < work; ++k) a = fma(a, 1.0000001, 1e-9);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rs, t * 712 + lane * 8, 0, 0);
        if (lane < 25) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), rs, t * 712 + 512 + lane * 8, 0, 0);
    }
    double acc = 0.0, r0[4], r1[4];
#pragma unroll
    for (int t = 0; t < 3; ++t) { r0[t] = t < rows ? g[t * 89 + lane] : 0.0; r1[t] = (t < rows && lane < 25) ? g[t * 89 + 64 + lane] : 0.0; }
    for (int t0 = 0; t0 < rows; t0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                         // (register slots by compile-time index: four sets in rotation, as in the product kernel)
            const int t = t0 + k;
            if (t < rows) {
                if (t + 3 < rows) { r0[(k + 3) & 3] = g[(t + 3) * 89 + lane]; r1[(k + 3) & 3] = lane < 25 ? g[(t + 3) * 89 + 64 + lane] : 0.0; }
                double b = r0[k] + r1[k];
                for (int q = 0; q < work; ++q) b = fma(b, 1.0000001, 1e-9);
                acc += b;
            }
        }
    }
    if (acc == 123.456) sink[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int64_t MB = argc > 1 ? atoll(argv[1]) : 1024;
    const int REPS = argc > 2 ? atoi(argv[2]) : 4;
    const int64_t bytes = MB << 20;
    double *a, *b, *sink;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&sink, 1 << 20));
    CHECK(hipMemset(a, 0, bytes));
    CHECK(hipMemset(b, 0, bytes));
    const int blocks = 2048;                                                   // 8 workgroups per CU
    const int64_t n8 = bytes / 8 / blocks, n16 = bytes / 16 / blocks;
    printf("# buffer %lld MB, %d workgroups x 256 threads, %d dispatches per kernel\n", (long long)MB, blocks, REPS);
    printf("# kernel true_read_bytes true_write_bytes\n");
    for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(read16, dim3(blocks), dim3(256), 0, 0, (const double2_*)a, sink, n16);
    printf("read16 %lld 0\n", (long long)(n16 * 16 * blocks));
    for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(read8, dim3(blocks), dim3(256), 0, 0, (const double*)a, sink, n8);
    printf("read8 %lld 0\n", (long long)(n8 * 8 * blocks));
    for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(write8, dim3(blocks), dim3(256), 0, 0, b, n8);
    printf("write8 0 %lld\n", (long long)(n8 * 8 * blocks));
    for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(copy8_lds, dim3(blocks), dim3(256), 0, 0, (const double*)a, b, n8);
    printf("copy8_lds %lld %lld\n", (long long)(n8 * 8 * blocks), (long long)(n8 * 8 * blocks));
    // the headline kernel's gain scratch: 1024 trajectories x 50 stages x 712 B = 36.5 MB written and read back (fits the Infinity Cache, not the L2)
    for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(rows712, dim3(1024), dim3(64), 0, 0, b, sink, 50, 64);
    printf("rows712 %lld %lld\n", 1024LL * 50 * 712, 1024LL * 50 * 712);
    // the same pattern beyond the Infinity Cache: as many trajectories as the buffer holds
    const int64_t nbig = bytes / (50 * 712);
    for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(rows712, dim3((unsigned)nbig), dim3(64), 0, 0, b, sink, 50, 0);
    printf("rows712_big %lld %lld\n", (long long)(nbig * 50 * 712), (long long)(nbig * 50 * 712));
    CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    printf("# done\n");
    return 0;
}
