// Probe 8: what bounds the rollout recursion x+ = f(x, u) of the fused OC kernels (one wavefront, all lanes redundant)?  Runs the generated
// quadrotor dynamics for T steps: (A) one trajectory, state in registers; (B) TWO independent trajectories interleaved in the same wave.
// If B costs about as much as A the recursion is latency-bound (dependent fp64 ops: 11.5 cycles; independent: ~5) and instruction-level
// parallelism is what is missing; if B costs twice A it is issue-bound.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I <csrc> -DPDP_MODEL_HEADER='"generated/<quadrotor>.h"' probes/rollout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../include/pdp_hip.h"
#define PDP_HD __host__ __device__ inline
#include PDP_MODEL_HEADER
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using Mdl = PdpModel;

template <int NTR>
__global__ void k(double* out, int T, const double* x0, const double* u, const double* theta, long long* cyc) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    double th[NP], pc[Mdl::NPC];
    for (int i = 0; i < NP; ++i) th[i] = theta[i];
    Mdl::precompute(th, pc);
    double xc[NTR][NX], xn[NTR][NX], uc[NTR][NU];
    for (int k = 0; k < NTR; ++k) for (int i = 0; i < NX; ++i) xc[k][i] = x0[k * NX + i];
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int k = 0; k < NTR; ++k) {
#pragma unroll
            for (int i = 0; i < NU; ++i) uc[k][i] = u[(k * T + t) * NU + i];
        }
#pragma unroll
        for (int k = 0; k < NTR; ++k) Mdl::dyn(xc[k], uc[k], th, pc, xn[k]);
#pragma unroll
        for (int k = 0; k < NTR; ++k)
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[k][i] = xn[k][i];
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int k = 0; k < NTR; ++k) for (int i = 0; i < NX; ++i) s += xc[k][i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// (C) as in the fused kernels: u staged in LDS and read one step ahead, x written to LDS by lane 0, theta / pc parked in LDS; FLAGS: 1 = no x store,
// 2 = theta / pc straight from registers, 4 = x stored to global memory instead of LDS
template <int FLAGS>
__global__ void kc(double* out, int T, const double* x0, const double* u, const double* theta, long long* cyc) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    __shared__ double lds[2048];
    double* par = lds; double* xs = lds + 64; double* us = xs + (T + 1) * NX;
    const int lane = threadIdx.x;
    {
        double th0[NP], pc0[Mdl::NPC];
        for (int i = 0; i < NP; ++i) th0[i] = theta[i];
        Mdl::precompute(th0, pc0);
        if (lane == 0) { for (int i = 0; i < NP; ++i) par[i] = th0[i]; for (int i = 0; i < Mdl::NPC; ++i) par[NP + i] = pc0[i]; }
    }
    for (int i = lane; i < T * NU; i += 64) us[i] = u[i];
    __syncthreads();
    double th[NP], pc[Mdl::NPC];
    if (FLAGS & 2) { for (int i = 0; i < NP; ++i) th[i] = theta[i]; Mdl::precompute(th, pc); }
    else { for (int i = 0; i < NP; ++i) th[i] = par[i]; for (int i = 0; i < Mdl::NPC; ++i) pc[i] = par[NP + i]; }
    double xc[NX], xn[NX], uc[NU], un[NU];
    for (int i = 0; i < NX; ++i) xc[i] = x0[i];
    for (int i = 0; i < NU; ++i) un[i] = us[i];
    long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
        for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = us[tn * NU + i]; }
        Mdl::dyn(xc, uc, th, pc, xn);
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = xn[i];
        if (!(FLAGS & 1) && !(FLAGS & 12) && lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[(t + 1) * NX + i] = xn[i];
        }
        if (FLAGS & 8) {                         // every lane stores: lane 0 to the staging row, the others to a dump area (no EXEC branch)
            double* dst = lane == 0 ? xs + (t + 1) * NX : xs + (T + 2) * NX + NU * T + lane;
#pragma unroll
            for (int i = 0; i < NX; ++i) dst[i] = xn[i];
        }
        if ((FLAGS & 4) && lane == 0) {          // x_{t+1} straight to global memory (vmcnt, not the LDS counter the u reads wait on)
#pragma unroll
            for (int i = 0; i < NX; ++i) out[64 + (t + 1) * NX + i] = xn[i];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    double s = xs[lane];
    for (int i = 0; i < NX; ++i) s += xc[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    const int T = 50;
    double hx[2 * NX], hu[2 * 50 * NU], hth[NP > 0 ? NP : 1];
    for (int i = 0; i < 2 * NX; ++i) hx[i] = 0.01 * (i % 7);
    hx[6] = 1.0; hx[NX + 6] = 1.0;
    for (int i = 0; i < 2 * T * NU; ++i) hu[i] = 2.4 + 0.01 * (i % 5);
    for (int i = 0; i < NP; ++i) hth[i] = 1.0 + 0.1 * i;
    double *dx, *du, *dth, *out; long long* cyc;
    CK(hipMalloc(&dx, sizeof hx)); CK(hipMalloc(&du, sizeof hu)); CK(hipMalloc(&dth, sizeof hth)); CK(hipMalloc(&out, 8 * (64 + 64 * 16))); CK(hipMalloc(&cyc, 8));
    CK(hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu, sizeof hu, hipMemcpyHostToDevice)); CK(hipMemcpy(dth, hth, sizeof hth, hipMemcpyHostToDevice));
    long long h;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("one trajectory            : %6.1f cycles per step\n", (double)h / T);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("two interleaved in a wave : %6.1f cycles per step (of both)\n", (double)h / T);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kc<0>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("as in the fused kernel    : %6.1f cycles per step\n", (double)h / T);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kc<1>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("  ... without the x store : %6.1f cycles per step\n", (double)h / T);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kc<2>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("  ... theta, pc not via LDS: %6.1f cycles per step\n", (double)h / T);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kc<4>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("  ... x stored to global  : %6.1f cycles per step\n", (double)h / T);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kc<8>, dim3(1), dim3(64), 0, 0, out, T, dx, du, dth, cyc);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("  ... all lanes store (lane 0 real, others dump): %6.1f cycles per step\n", (double)h / T);
    return 0;
}
