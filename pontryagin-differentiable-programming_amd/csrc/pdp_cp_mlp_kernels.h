// pdp_cp_mlp_kernels.h - fused ControlPlanning.step for the tanh-MLP policy (reference PDP/PDP.py:727-759 setNeuralPolicy, 763-786 integrateSys,
// 850-878 step) with the network held in REGISTERS: cp_step_mlp16_kernel.  Same adjoint formulation, same arithmetic in the same order as
// cp_step_adjoint_kernel (pdp_model_kernels.h) - loss and trajectory bit-identical, gradient equal to 1e-16 (tests/test_gpu_cp_mlp.py) - for networks of at most 4 weight layers of
// width <= 16 (the reference's examples: hidden [n] or [n, n], PDP.py:730 / cartpole_PDP_neural.py:49; C5: [13, 13] -> 4, p = 420).
//
// Why.  At C5b (quadrotor, T = 100, p = 420, B = 1024) the general kernel spent ~18 k cycles per time step and trajectory: every layer product was a
// loop of run-time length whose trip fetched one weight and one activation from LDS (~100 cycles per trip, 13 trips, 6 products per step), the
// backward sweep re-ran the whole network on ONE lane per time step (26 fp64 tanh in sequence) only to recover u_t, and the parameters took 3.4 KB of
// LDS per trajectory.  Here:
//   * lane 16 k + r holds ROW r of A_k (forward product) and COLUMN r of A_k (transposed product of the adjoint sweep), zero-padded to 16: 33 doubles
//     per lane for up to four layers, loaded once from theta (column-major vec, PDP.py:739);
//   * a layer product is 16 FMAs on those registers with the input element broadcast from the lane that owns it (v_readlane -> scalar operand): the
//     trip count is a compile-time 16, there is no LDS access in it; the wave computes layer k for all four lane groups at once, only group k's
//     result is kept (the layers are sequential anyway);
//   * activations stay in the lane that computed them: stored to / re-read from the workspace by that same lane (one 8-byte access per lane and
//     step, requested one step ahead in the adjoint sweep); u_t is kept in LDS by the rollout, so the backward chunk evaluation is just the generated
//     Jacobian code;
//   * tanh is evaluated once per layer and step (one call for all lanes), never in the adjoint sweep (1 - z^2 from the stored z).
// Measured (profiles/r03_cp_mlp_kernel.txt): C5b, B = 1024: 0.844 -> 0.281 ms (17.7 k -> 5.9 k cycles per time step), 3.6 - 4.0 M trajectories/s from B = 1024 up.
// One wavefront per trajectory as before (the batch of C5 is 1024 per GPU = one trajectory per SIMD).  A batched-over-trajectories MFMA form (16
// trajectories' activations as the columns of one 16x16x16 product) would cut the instruction count per trajectory further, but only pays for batches
// well beyond 16 x 1024 trajectories per GPU, which no BASELINE configuration has; see DESIGN.md section 4.3.
#pragma once
#include "pdp_model_kernels.h"

namespace pdp {

constexpr int MLP16_W = 16, MLP16_MAXL = 4;
PDP_DEV int mlp_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

struct Mlp16Layout { int xs, us, zs, ds, misc, blk, total, actw; };
template <class Mdl>
__host__ __device__ inline Mlp16Layout cp_mlp16_layout(int T, int rows) {
    Mlp16Layout L;
    int o = 0;
    L.xs = o; o += (T + 1) * Mdl::NX;
    L.us = o; o += T * Mdl::NU;
    L.zs = o; o += MLP16_MAXL * MLP16_W + 2;              // layer inputs of the current step | 1.0 (bias factor) | 0.0
    L.ds = o; o += MLP16_MAXL * MLP16_W;                  // layer deltas of the current step
    L.misc = o; o += Mdl::NX + Mdl::NU + 2;               // (d pi/dx)' v | spare
    L.blk = o; o += rows * ((Mdl::PATH_NVAR + 1 + Mdl::PATH_NCONST) | 1);      // pool rows [entries | 0.0 | constants]: every row carries its own zero and constants
    L.total = o + 8;
    L.actw = 64;                                           // workspace doubles per time step: one per lane
    return L;
}
template <class Mdl>
__host__ inline bool cp_mlp16_ok(const pdp_policy& pol) {
    if (pol.kind != PDP_POLICY_MLP || pol.n_layers < 1 || pol.n_layers > MLP16_MAXL || Mdl::NX > MLP16_W || Mdl::NU > MLP16_W) return false;
    for (int k = 0; k < pol.n_layers; ++k) if (pol.sizes[k] < 1 || pol.sizes[k] > MLP16_W) return false;
    return true;
}
// pool rows per evaluation pass: as many as fit beside the trajectory in 40 KB (four wavefronts, one per SIMD, share a CU's 160 KB)
// budget_kb: LDS per workgroup the rows are sized for - 40 (four wavefronts per CU, one per SIMD) or 20 (eight: two per SIMD, for batches beyond one trajectory per SIMD;
// the kernel needs 240 VGPRs: two waves fit a SIMD)
template <class Mdl>
__host__ inline int cp_mlp16_rows(int T, int budget_kb = 40) {
    const int stride = (Mdl::PATH_NVAR + 1 + Mdl::PATH_NCONST) | 1;
    const int fixed = cp_mlp16_layout<Mdl>(T, 0).total;
    int fit = (budget_kb * 1024 / 8 - fixed) / stride;
    if (fit < 4 && budget_kb < 40) fit = (40 * 1024 / 8 - fixed) / stride;
    if (fit < 8) { const int f2 = (160 * 1024 / 8 - fixed) / stride; if (f2 > fit) fit = f2; }     // long horizons: one workgroup per CU
    return fit < 1 ? 0 : (fit > 64 ? 64 : fit);
}

template <class Mdl>
__global__ void __launch_bounds__(64) cp_step_mlp16_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0, const double* __restrict__ theta,
                                                            int tb, double* __restrict__ loss, double* __restrict__ grad, double* __restrict__ xo,
                                                            double* __restrict__ uo, double* __restrict__ ws_acts, int CH) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, W = MLP16_W;
    constexpr int NV = Mdl::PATH_NVAR, STRIDE = (NV + 1 + Mdl::PATH_NCONST) | 1;      // row: [entries | 0.0 | constants]
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Mlp16Layout L = cp_mlp16_layout<Mdl>(T, CH);
    double *xs = lds + L.xs, *us = lds + L.us, *zs = lds + L.zs, *ds = lds + L.ds, *dpx = lds + L.misc, *pool = lds + L.blk;
    const int b = blockIdx.x, lane = threadIdx.x, grp = lane >> 4, idx = lane & 15;
    double* actg = ws_acts + (int64_t)b * T * 64;            // [T][64]: element (t, lane) written and re-read by the same lane
    const int nl = pol.n_layers;
    const double* thb = theta + (int64_t)b * tb;
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    // layer tables (uniform): parameter offset, rows, cols
    int loff[MLP16_MAXL], lrows[MLP16_MAXL], lcols[MLP16_MAXL];
    {
        int cols = NX, off = 0;
#pragma unroll
        for (int k = 0; k < MLP16_MAXL; ++k) {
            loff[k] = off; lcols[k] = cols; lrows[k] = (k < nl) ? pol.sizes[k] : 0;
            if (k < nl) { off += lrows[k] * cols + lrows[k]; cols = lrows[k]; }
        }
    }
    // this lane's layer (its group), and its row / column of that layer's weights: A_k is stored column-major, vec_F(A_k)[r + c rows]   (PDP.py:739)
    int my_off = 0, my_rows = 0, my_cols = 0;
#pragma unroll
    for (int k = 0; k < MLP16_MAXL; ++k) if (grp == k) { my_off = loff[k]; my_rows = lrows[k]; my_cols = lcols[k]; }
    double Wrow[W], Wcol[W], bias;
#pragma unroll
    for (int c = 0; c < W; ++c) {
        Wrow[c] = (idx < my_rows && c < my_cols) ? thb[my_off + idx + c * my_rows] : 0.0;          // A[idx][c]
        Wcol[c] = (idx < my_cols && c < my_rows) ? thb[my_off + c + idx * my_rows] : 0.0;          // A[c][idx]
    }
    bias = idx < my_rows ? thb[my_off + my_rows * my_cols + idx] : 0.0;
    if (lane == 0) { zs[MLP16_MAXL * W] = 1.0; zs[MLP16_MAXL * W + 1] = 0.0; }
    wave_lds_sync();

    // ---------------- forward rollout: x_{t+1} = f(x_t, pi(x_t)), executed uniformly by the wave
    double J = 0.0;
    {
        double xc[NX], xn[NX], uc[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)b * NX + i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = xc[i];
        }
        for (int t = 0; t < T; ++t) {
            double zprev = 0.0, zkeep = 0.0;
#pragma unroll
            for (int k = 0; k < MLP16_MAXL; ++k) {
                if (k < nl) {
                    double a = bias;
#pragma unroll
                    for (int c = 0; c < W; ++c) {
                        const double zc = (k == 0) ? (c < NX ? xc[c < NX ? c : 0] : 0.0) : readlane_f64(zprev, 16 * (k > 0 ? k - 1 : 0) + c);
                        a = fma(Wrow[c], zc, a);               // a += A[r][c] z[c], c ascending (as policy_eval)
                    }
                    if (k + 1 < nl) {
                        const double zk = pdp_tanh(a);
                        zprev = zk;
                        zkeep = grp == k ? zk : zkeep;
                    } else {
#pragma unroll
                        for (int j = 0; j < NU; ++j) uc[j] = readlane_f64(a, 16 * k + j);
                    }
                }
            }
            actg[t * 64 + lane] = zkeep;                        // this lane's hidden activation of step t (groups >= nl - 1: unused)
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < NU; ++j) us[t * NU + j] = uc[j];
            }
            if (uo) {
                double uv = 0.0;
#pragma unroll
                for (int j = 0; j < NU; ++j) uv = lane == j ? uc[j] : uv;
                if (lane < NU) uo[((int64_t)b * T + t) * NU + lane] = uv;
            }
            Mdl::dyn(xc, uc, nullptr, pc, xn);
            J += Mdl::path_cost(xc, uc, nullptr, pc);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[(t + 1) * NX + i] = xn[i];
            }
        }
        J += Mdl::final_cost(xc, nullptr, pc);
        wave_lds_sync();
    }
    if (xo) for (int i = lane; i < (T + 1) * NX; i += 64) xo[(int64_t)b * (T + 1) * NX + i] = xs[i];

    // (computed HERE, from a lane id the optimiser cannot see through: hoisted above the rollout these 16 registers - and the 50 of the pool offsets below - stay live
    // across it and the kernel needs 266 VGPRs, one more than lets two wavefronts share a SIMD)
    const int ln_ = mlp_opaque(lane);
    // per-lane parameter slots q = 0..7 (parameter index lane + 64 q): LDS offsets of the two factors of d cost / d theta_j (delta_k[r] * z_k[c], or 1.0)
    int pr[8], pz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = ln_ + 64 * q;
        pr[q] = L.zs + MLP16_MAXL * W + 1; pz[q] = L.zs + MLP16_MAXL * W + 1;       // 0.0 * 0.0
        if (j < p) {
#pragma unroll
            for (int k = 0; k < MLP16_MAXL; ++k) {
                if (k < nl && j >= loff[k] && j < loff[k] + lrows[k] * lcols[k] + lrows[k]) {
                    const int e = j - loff[k], nw = lrows[k] * lcols[k];
                    if (e < nw) { pr[q] = L.ds + k * W + e % lrows[k]; pz[q] = L.zs + k * W + e / lrows[k]; }
                    else { pr[q] = L.ds + k * W + (e - nw); pz[q] = L.zs + MLP16_MAXL * W; }
                }
            }
        }
    }
    // ---------------- adjoint sweep: mu_T = h_x ; v_t = c_u + G_t' mu_{t+1} ; grad += (d pi/d theta)' v_t ; mu_t = c_x + F_t' mu_{t+1} + (d pi/d x)' v_t
    double mu[NX];
    {
        double xT[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xT[i] = xs[T * NX + i];
        Mdl::dhx(xT, nullptr, pc, mu);
    }
    double gacc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) gacc[q] = 0.0;
    // per-lane pool offsets: column idx of F (group 0, idx < NX), column idx of G (the output layer's group, idx < NU), c_x[idx], c_u[idx]
    const int grp_ = ln_ >> 4, idx_ = ln_ & 15;
    const bool lane_x = grp_ == 0 && idx_ < NX, lane_u = grp_ == nl - 1 && idx_ < NU;
    // (slots inside a row: an entry, the row's 0.0, or one of its constants - the same distance from row to row whatever the element is, so the offsets need no stride
    // of their own: 28 registers less than with a shared constant pool, which is what lets two wavefronts share a SIMD)
    int fo[NX], go[NX], cxo = 0, cuo = 0;
    auto enc = [&](int code, int& o) { o = code >= 0 ? code : (code == -1 ? NV : NV + 1 + (-2 - code)); };
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        enc(lane_x ? Mdl::path_code(0, k * NX + idx_) : -1, fo[k]);
        enc(lane_u ? Mdl::path_code(1, k * NU + idx_) : -1, go[k]);
    }
    enc(lane_x ? Mdl::path_code(2, idx_) : -1, cxo);
    enc(lane_u ? Mdl::path_code(3, idx_) : -1, cuo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the rollout's activation stores have landed (re-read by the same lanes below)
    double znext = actg[(T - 1) * 64 + lane];
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
    for (int c = nchunk - 1; c >= 0; --c) {
        const int t0 = c * ch, cnt = min(ch, T - t0);
        wave_lds_sync();
        if (lane < cnt) {                           // lane = time step: F, G, c_x, c_u at (x_t, u_t) of the stored trajectory
            const int t = t0 + lane;
            double xc[NX], uc[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xs[t * NX + i];
#pragma unroll
            for (int j = 0; j < NU; ++j) uc[j] = us[t * NU + j];
            double* row = pool + lane * STRIDE;
            PackedSink sk{row};
            Mdl::eval_path(xc, uc, nullptr, nullptr, pc, sk);
            row[NV] = 0.0;
#pragma unroll
            for (int i = 0; i < Mdl::PATH_NCONST; ++i) row[NV + 1 + i] = Mdl::path_const(i);
        }
        wave_lds_sync();
        for (int tl = cnt - 1; tl >= 0; --tl) {
            const int t = t0 + tl;
            const double zk_own = znext;                          // this lane's activation of step t
            znext = actg[(t > 0 ? t - 1 : 0) * 64 + lane];         // ... and of the step the sweep visits next, requested now - WITHOUT a branch: a conditional block
                                                                  // around the load makes the wait behind it a vmcnt(0), i.e. every step waits for the load it has just issued
            // layer inputs of the step into LDS (factors of the parameter gradient): z_0 = x_t, z_{k+1} = the activations of group k
            if (lane < NX) zs[lane] = xs[t * NX + lane];
            if (grp + 1 < nl) zs[(grp + 1) * W + idx] = zk_own;
            // v = c_u + G' mu   (lanes of the output layer's group)
            const double* rowt = pool + tl * STRIDE;
            double delta = rowt[cuo];
#pragma unroll
            for (int k = 0; k < NX; ++k) delta = fma(rowt[go[k]], mu[k], delta);
            double dkeep = 0.0, back0 = 0.0;
#pragma unroll
            for (int k = MLP16_MAXL - 1; k >= 0; --k) {
                if (k < nl) {
                    dkeep = grp == k ? delta : dkeep;             // delta_k lives in group k
                    double back = 0.0;                            // (A_k' delta_k)[idx] on group k's lanes, r ascending (as the general kernel)
#pragma unroll
                    for (int r = 0; r < W; ++r) back = fma(Wcol[r], readlane_f64(delta, 16 * k + r), back);
                    if (k > 0) {
                        const double moved = __shfl_down(back, 16, 64);      // to the lane of group k - 1 that owns z_k[idx]
                        delta = moved * (1.0 - zk_own * zk_own);
                    } else back0 = back;
                }
            }
            ds[grp * W + idx] = dkeep;
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < 8; ++q) gacc[q] += lds[pr[q]] * lds[pz[q]];
            // mu_t = c_x + F' mu_{t+1} + (d pi/dx)' v   (lanes idx < NX of group 0), then broadcast
            double m_new = rowt[cxo] + back0;
#pragma unroll
            for (int k = 0; k < NX; ++k) m_new = fma(rowt[fo[k]], mu[k], m_new);
#pragma unroll
            for (int i = 0; i < NX; ++i) mu[i] = readlane_f64(m_new, i);
            wave_lds_sync();                                      // (zs / ds are rewritten by the next step)
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int j = lane + 64 * q; if (j < p) grad[(int64_t)b * p + j] = gacc[q]; }
    if (lane == 0) loss[b] = J;
    (void)dpx;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// cp_step_mlp4t_kernel (round 5): FOUR trajectories per wavefront, the layer products on the 4-block fp64 MFMA.
//
// What cp_step_mlp16_kernel leaves on the table (round-4 verdict, item 4): its wavefront computes every layer product for all four 16-lane groups and keeps one
// (the layers are sequential: 4-fold redundancy), its rollout runs one trajectory uniformly on 64 lanes, and its two tanh per step serve one trajectory.  Here a
// wavefront owns four trajectories j = lane & 3 that SHARE the parameters (theta stride 0 - the reference's case: one policy, PDP.py:850-878):
//   * v_mfma_f64_4x4x4_4b computes four independent 4x4x4 products D_b = C_b + A_b B_b (A lane = i + 4 b + 16 k, B lane = j + 4 b + 16 k, D lane = j + 4 b + 16 i,
//     pdp_tile.h).  Block b takes ROWS 4 b .. 4 b + 3 of a layer, the columns j of B are the four TRAJECTORIES, four such instructions (inner index 4 q + k, q = 0..3)
//     give the whole 16-wide layer for four trajectories: A_q[lane] = W[4 b + i][4 q + k], B_q[lane] = z[4 q + k][trajectory j] (the same for every b),
//     D[lane (j, b, i)] = a[4 b + i][trajectory j].  The accumulator starts at the bias and the inner index ascends: the same fma chain as the register kernel, bit for bit;
//   * a layer's output - one double per lane: row 4 b + i of trajectory j - goes through ONE tanh for all four trajectories and becomes the next layer's B operands by a
//     broadcast of quad q inside each row of 16 lanes (B_q[lane] = D[(lane & 0x33) | 4 q]) - done through LDS, transposed on the way in (one write, two 128-bit reads);
//   * the adjoint products A_k' delta_k are the same instruction with the transposed weights as A operands; the deltas and activations never leave that layout;
//   * the rollout evaluates the dynamics for trajectory j on all 16 lanes of its group (redundant, free), so the states are in registers where layer 0 needs them
//     (B_q = x[4 q + k]: a select on k = lane >> 4) and the controls come back from the output layer through the same LDS hop;
//   * the parameter gradient of a trajectory is the sum over time of the outer products delta_k z_k': lane (j, row r) accumulates row r of every layer (16 doubles per
//     layer), the z_k of the step staged in LDS and read back by the 16 lanes of the trajectory as 128-bit broadcasts.
// Trajectories and controls live in global memory (the API outputs or the workspace), the activations in the workspace in D layout (one coalesced 512-byte store per
// hidden layer and step for four trajectories), the Jacobian pool in LDS: 16 time steps x 4 trajectories per evaluation pass.
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
struct Mlp4tLayout { int zst, mus, zt, pool, total; };
template <class Mdl>
__host__ __device__ inline Mlp4tLayout cp_mlp4t_layout() {
    Mlp4tLayout L;
    int o = 0;
    L.zst = o; o += 4 * MLP16_MAXL * MLP16_W;            // layer inputs of the current step, [trajectory][layer][16]
    L.mus = o; o += 4 * MLP16_W;                         // mu of the four trajectories
    L.zt = o; o += 3 * 64;                               // D layout -> B operands (forward | backward) and the controls: one row of 16 per trajectory
    L.pool = o; o += 64 * ((Mdl::PATH_NVAR + 1 + Mdl::PATH_NCONST) | 1);
    L.total = o + 8;
    return L;
}
// workspace: activations [ceil(B / 4)][T][MAXL - 1][64] | x [B][T + 1][NX] | u [B][T][NU]   (the trajectory parts are used when the caller does not ask for x / u)
template <class Mdl>
__host__ inline int64_t cp_mlp4t_ws_doubles(int B, int T) {
    return (int64_t)((B + 3) / 4) * T * (MLP16_MAXL - 1) * 64 + (int64_t)B * ((int64_t)(T + 1) * Mdl::NX + (int64_t)T * Mdl::NU);
}

template <class Mdl, int NL>
__global__ void __launch_bounds__(64) cp_step_mlp4t_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0, const double* __restrict__ theta,
                                                            double* __restrict__ loss, double* __restrict__ grad, double* __restrict__ xo, double* __restrict__ uo,
                                                            double* __restrict__ ws) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, W = MLP16_W, ML = NL, AS = MLP16_MAXL - 1;        // NL weight layers (a template parameter: the gradient rows of absent layers would cost 32 registers each); AS: activation slots per step in the workspace
    constexpr int NV = Mdl::PATH_NVAR, STRIDE = (NV + 1 + Mdl::PATH_NCONST) | 1;      // pool row: [entries | 0.0 | constants]
    static_assert(NX <= W && NU <= 4, "states in one 16-wide layer input, controls in the rows of one block");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Mlp4tLayout L = cp_mlp4t_layout<Mdl>();
    double *zst = lds + L.zst, *mus = lds + L.mus, *ztf = lds + L.zt, *ztb = ztf + 64, *ztu = ztf + 128, *pool = lds + L.pool;
    const int lane = threadIdx.x, j = lane & 3, bq = (lane >> 2) & 3, kq = lane >> 4, rowid = 4 * bq + kq;      // D layout: this lane holds row `rowid` of trajectory j
    const int wv = blockIdx.x, b = 4 * wv + j;
    const bool live = b < B;
    const int bb = live ? b : B - 1;                      // (a padding trajectory repeats the last one; its stores are dropped)
    constexpr int nl = NL;
    double* actg = ws + (int64_t)wv * T * AS * 64;                                                   // [T][ML - 1][64]
    double* wx = ws + (int64_t)((B + 3) / 4) * T * AS * 64;
    double* xs = xo ? xo + (int64_t)bb * (T + 1) * NX : wx + (int64_t)bb * (T + 1) * NX;
    double* us = uo ? uo + (int64_t)bb * T * NU : wx + (int64_t)B * (T + 1) * NX + (int64_t)bb * T * NU;
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    // layer tables (uniform)
    int loff[ML], lrows[ML], lcols[ML];
    {
        int cols = NX, off = 0;
#pragma unroll
        for (int k = 0; k < ML; ++k) {
            loff[k] = off; lcols[k] = cols; lrows[k] = (k < nl) ? pol.sizes[k] : 0;
            if (k < nl) { off += lrows[k] * cols + lrows[k]; cols = lrows[k]; }
        }
    }
    // operands: AF[k][q] = W_k[4 b + i][4 q + k'] and AT[k][q] = W_k[4 q + k'][4 b + i]  with i = lane & 3, b = bq, k' = kq  (column-major vec, PDP.py:739); bias in D layout
    double AF[ML][4], AT[ML][4], bias[ML];
    {
        const int ai = lane & 3;
#pragma unroll
        for (int k = 0; k < ML; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * bq + ai, c = 4 * q + kq;
                AF[k][q] = (k < nl && r < lrows[k] && c < lcols[k]) ? theta[loff[k] + r + c * lrows[k]] : 0.0;
                AT[k][q] = (k < nl && c < lrows[k] && r < lcols[k]) ? theta[loff[k] + c + r * lrows[k]] : 0.0;
            }
            bias[k] = (k < nl && rowid < lrows[k]) ? theta[loff[k] + lrows[k] * lcols[k] + rowid] : 0.0;
        }
    }
    // D layout -> the four B operands of the next product: B_q[lane (j, b, k)] = d[row 4 q + k][trajectory j], whatever b.  Through LDS, transposed on the way in: the lane
    // that holds row 4 b + i writes slot [j][i][b], a reader finds its four values side by side at [j][k][0..3] (one 64-bit write, two 128-bit reads; as eight
    // ds_bpermute - two per operand - this was 17 % of the kernel, probes/mlp4t_timing.py)
    typedef double pdp_d2 __attribute__((ext_vector_type(2)));
    auto to_b = [&](double d, double* zt, double (&Bq)[4]) {
        zt[j * 16 + kq * 4 + bq] = d;
        wave_lds_sync();
        const pdp_d2 lo = *(const pdp_d2*)(zt + j * 16 + kq * 4), hi = *(const pdp_d2*)(zt + j * 16 + kq * 4 + 2);
        Bq[0] = lo.x; Bq[1] = lo.y; Bq[2] = hi.x; Bq[3] = hi.y;
    };

    // ---------------- forward rollout: x_{t+1} = f(x_t, pi(x_t)) for trajectory j on the 16 lanes of its group
    double J = 0.0;
    {
        double xc[NX], xn[NX], uc[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)bb * NX + i];
        if (lane < 4 && live) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = xc[i];
        }
        for (int t = 0; t < T; ++t) {
            double zD = 0.0;
#pragma unroll
            for (int k = 0; k < ML; ++k) {
                if (k < nl) {
                    double a = bias[k], Bq[4];
                    if (k == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double c0 = 4 * q + 0 < NX ? xc[4 * q + 0 < NX ? 4 * q + 0 : 0] : 0.0, c1 = 4 * q + 1 < NX ? xc[4 * q + 1 < NX ? 4 * q + 1 : 0] : 0.0,
                                         c2 = 4 * q + 2 < NX ? xc[4 * q + 2 < NX ? 4 * q + 2 : 0] : 0.0, c3 = 4 * q + 3 < NX ? xc[4 * q + 3 < NX ? 4 * q + 3 : 0] : 0.0;
                            Bq[q] = kq == 0 ? c0 : (kq == 1 ? c1 : (kq == 2 ? c2 : c3));
                        }
                    } else to_b(zD, ztf, Bq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (4 * q < lcols[k]) a = mma4_blk(AF[k][q], Bq[q], a);
                    if (k + 1 < nl) {
                        zD = pdp_tanh(a);
                        actg[((int64_t)t * AS + k) * 64 + lane] = zD;
                    } else {                              // the controls: rows 0 .. NU - 1 of the output layer, to every lane of the trajectory
                        ztu[j * 16 + rowid] = a;
                        wave_lds_sync();
                        const pdp_d2 lo = *(const pdp_d2*)(ztu + j * 16), hi = *(const pdp_d2*)(ztu + j * 16 + 2);
                        const double uu[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                        for (int i = 0; i < NU; ++i) uc[i] = uu[i];
                    }
                }
            }
            if (lane < 4 && live) {
#pragma unroll
                for (int i = 0; i < NU; ++i) us[(int64_t)t * NU + i] = uc[i];
            }
            Mdl::dyn(xc, uc, nullptr, pc, xn);
            J += Mdl::path_cost(xc, uc, nullptr, pc);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            if (lane < 4 && live) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[(int64_t)(t + 1) * NX + i] = xn[i];
            }
        }
        J += Mdl::final_cost(xc, nullptr, pc);
        // mu_T = h_x(x_T), all 13 entries in every lane of the trajectory
        double h[NX];
        Mdl::dhx(xc, nullptr, pc, h);
        if (lane < 4) {
#pragma unroll
            for (int i = 0; i < NX; ++i) mus[lane * W + i] = h[i];
        }
    }
    if (lane < 4 && live) loss[b] = J;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trajectory and the activations have landed (re-read below by other lanes of this wavefront)
    wave_lds_sync();

    // ---------------- adjoint sweep, four trajectories at once
    const int ln_ = mlp_opaque(lane);                     // (per-lane maps from an opaque lane id: not hoisted above the rollout)
    const int j_ = ln_ & 3, rid_ = 4 * ((ln_ >> 2) & 3) + (ln_ >> 4);
    int fo[NX], go[NX], cxo, cuo;
    auto enc = [&](int code) { return code >= 0 ? code : (code == -1 ? NV : NV + 1 + (-2 - code)); };
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        fo[k] = enc(rid_ < NX ? Mdl::path_code(0, k * NX + rid_) : -1);
        go[k] = enc(rid_ < NU ? Mdl::path_code(1, k * NU + rid_) : -1);
    }
    cxo = enc(rid_ < NX ? Mdl::path_code(2, rid_) : -1);
    cuo = enc(rid_ < NU ? Mdl::path_code(3, rid_) : -1);
    double mu[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) mu[i] = mus[j_ * W + i];
    double gacc[ML][W], gbias[ML];
#pragma unroll
    for (int k = 0; k < ML; ++k) { gbias[k] = 0.0;
#pragma unroll
        for (int c = 0; c < W; ++c) gacc[k][c] = 0.0; }
    // the layer inputs of step t in D layout: z_0 = x_t (row < NX), z_k = stored activation of layer k - 1; requested one step ahead
    double zpre[ML];
    auto request = [&](int t, double (&z)[ML]) {
        z[0] = xs[(int64_t)t * NX + (rid_ < NX ? rid_ : 0)];
#pragma unroll
        for (int k = 1; k < ML; ++k) z[k] = actg[((int64_t)t * AS + (k - 1)) * 64 + lane];
    };
    request(T - 1, zpre);
    constexpr int CH = 16;                                // time steps per evaluation pass: 16 steps x 4 trajectories = 64 pool rows
    for (int t0 = ((T - 1) / CH) * CH; t0 >= 0; t0 -= CH) {
        const int cnt = min(CH, T - t0);
        wave_lds_sync();
        {                                                 // lane = (trajectory j, step s): F, G, c_x, c_u at (x_t, u_t) of the stored trajectory
            const int s_ = lane >> 2;
            if (s_ < cnt) {
                const int t = t0 + s_;
                double xc[NX], uc[NU];
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = xs[(int64_t)t * NX + i];
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = us[(int64_t)t * NU + i];
                double* row = pool + lane * STRIDE;       // row index = j + 4 s
                PackedSink sk{row};
                Mdl::eval_path(xc, uc, nullptr, nullptr, pc, sk);
                row[NV] = 0.0;
#pragma unroll
                for (int i = 0; i < Mdl::PATH_NCONST; ++i) row[NV + 1 + i] = Mdl::path_const(i);
            }
        }
        wave_lds_sync();
        for (int tl = cnt - 1; tl >= 0; --tl) {
            const int t = t0 + tl;
            double zin[ML];
#pragma unroll
            for (int k = 0; k < ML; ++k) zin[k] = zpre[k];
            request(t > 0 ? t - 1 : 0, zpre);             // (no branch around the loads: the wait behind a conditional block would be a vmcnt(0))
            if (rid_ >= NX) zin[0] = 0.0;
#pragma unroll
            for (int k = 1; k < ML; ++k) if (k >= nl) zin[k] = 0.0;
            // stage the layer inputs for the gradient's outer products
#pragma unroll
            for (int k = 0; k < ML; ++k) if (k < nl) zst[(j_ * MLP16_MAXL + k) * W + rid_] = zin[k];
            const double* rowt = pool + (j_ + 4 * tl) * STRIDE;
            // v = c_u + G' mu  (rows < NU of the output layer's delta)
            double delta = rowt[cuo];
#pragma unroll
            for (int k = 0; k < NX; ++k) delta = fma(rowt[go[k]], mu[k], delta);
            if (rid_ >= NU) delta = 0.0;
            double dk[ML], back0 = 0.0;
#pragma unroll
            for (int k = ML - 1; k >= 0; --k) {
                dk[k] = 0.0;
                if (k < nl) {
                    dk[k] = delta;
                    double back = 0.0, Dq[4];             // (A_k' delta_k)[row], inner index ascending
                    to_b(delta, ztb, Dq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (4 * q < lrows[k]) back = mma4_blk(AT[k][q], Dq[q], back);
                    if (k > 0) delta = back * (1.0 - zin[k] * zin[k]);
                    else back0 = back;
                }
            }
            wave_lds_sync();
            // parameter gradient: row `rid_` of every layer, z_k of this trajectory from the staging (128-bit broadcast reads)
#pragma unroll
            for (int k = 0; k < ML; ++k) {
                if (k < nl) {
                    const double* zk = zst + (j_ * MLP16_MAXL + k) * W;
#pragma unroll
                    for (int c = 0; c < W; ++c) gacc[k][c] += dk[k] * zk[c];
                    gbias[k] += dk[k];
                }
            }
            // mu_t = c_x + F' mu_{t+1} + (d pi/dx)' v   (rows < NX), then to every lane of the trajectory
            double m_new = rowt[cxo] + back0;
#pragma unroll
            for (int k = 0; k < NX; ++k) m_new = fma(rowt[fo[k]], mu[k], m_new);
            if (rid_ < NX) mus[j_ * W + rid_] = m_new;
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < NX; ++i) mu[i] = mus[j_ * W + i];
        }
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < ML; ++k) {
            if (k < nl && rid_ < lrows[k]) {
#pragma unroll
                for (int c = 0; c < W; ++c) if (c < lcols[k]) grad[(int64_t)b * p + loff[k] + rid_ + c * lrows[k]] = gacc[k][c];
                grad[(int64_t)b * p + loff[k] + lrows[k] * lcols[k] + rid_] = gbias[k];
            }
        }
    }
}

}  // namespace pdp
