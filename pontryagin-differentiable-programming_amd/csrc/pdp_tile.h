// pdp_tile.h - register-resident 16x16 fp64 tile algebra for one gfx950 wavefront.
//
// A tile is a 16x16 row-major fp64 matrix spread over the 64 lanes: lane l, register r (0..3) holds the
// element with flat index 64*r + l, i.e. row = (l >> 4) + 4r, col = l & 15.  This is exactly the C/D layout
// of v_mfma_f64_16x16x4_f64, and (measured, profiles/r01_probe_mfma_f64.txt) feeding register r of two tiles
// X, Y as the A and B operands of the r-th MFMA accumulates X^T * Y - bit-exact with a k-ascending fma chain.
// So every product of the Riccati recursion is phrased as  C + X^T Y  and never leaves the register file:
// no LDS traffic, no shuffles.  A transposed operand is obtained by loading the source transposed.
//
// v_mfma_f64_16x16x4_f64 on MI355X: 64 cycles issue, so a 16x16x16 product costs 256 cycles; products with at most 4 OUTPUT rows
// (everything multiplied by the m x m control block) run on the 4-block form v_mfma_f64_4x4x4_4b (25-32 cycles each, mma4_* below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define PDP_DEV __device__ __forceinline__

namespace pdp {

PDP_DEV int tile_row(int lane, int r) { return (lane >> 4) + 4 * r; }
PDP_DEV int tile_col(int lane) { return lane & 15; }

PDP_DEV d4 zero4() { d4 z = {0.0, 0.0, 0.0, 0.0}; return z; }

// C + X^T Y, inner dimension 16 (4 MFMAs)
PDP_DEV d4 mma_tn(const d4 x, const d4 y, d4 c) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], c, 0, 0, 0);
    return c;
}
// C + X^T Y restricted to the first 4 rows of X and Y (1 MFMA): products over the control dimension m <= 4
PDP_DEV d4 mma_tn_r0(const d4 x, const d4 y, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], c, 0, 0, 0);
}
// C - X^T Y over the first 4 rows
PDP_DEV d4 mms_tn_r0(const d4 x, const d4 y, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(-x[0], y[0], c, 0, 0, 0);
}

// Products with only 4 output rows (everything multiplied by the m x m control block) run on the 4-block MFMA
// v_mfma_f64_4x4x4_4b_f64 - four independent 4x4x4 products, 25 cycles dependent-on-C / 32 dependent-on-A against 64 / 90 of the
// 16x16x4 form (profiles/r01_probe_mfma_f64_4x4.txt; operand layouts probed there: A lane = i + 4 b + 16 k, B lane = j + 4 b + 16 k,
// D lane = j + 4 b + 16 i).  In the register conventions of this file (lane = 16 row + col):
//      D[i][c] = C[i][c] + sum_{k<4} X[k][4 (c/4) + i] * Y[k][c]          i < 4;   D, C, X, Y: one register ("rows 0..3") each
// i.e. C + M^T Y for a 4 x 4 block M when X holds M in each of its four column blocks, X = [M|M|M|M].
PDP_DEV double mma4_blk(double xrep, double y, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(xrep, y, c, 0, 0, 0); }
// rows 0..3 of C + X^T Y over the full inner dimension 16 (4 small MFMAs), X column-block replicated: Xrep[k][4 b + i] = X[k][i]
PDP_DEV double mma4_tn(const d4 xrep, const d4 y, double c) {
    c = mma4_blk(xrep[0], y[0], c);
    c = mma4_blk(xrep[1], y[1], c);
    c = mma4_blk(xrep[2], y[2], c);
    c = mma4_blk(xrep[3], y[3], c);
    return c;
}

// Load a dense row-major R x C matrix (leading dimension ld) into the tile at offset (roff, coff); elements
// outside the matrix are 0.  TRANS loads the transpose (tile(i,j) = M[j-coff'...]) - see callers.
template <bool TRANS>
PDP_DEV d4 load_dense(const double* __restrict__ M, int R, int C, int ld, int roff, int coff, int lane) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = tile_row(lane, r) - roff, col = tile_col(lane) - coff;
        bool ok = TRANS ? (row >= 0 && row < C && col >= 0 && col < R) : (row >= 0 && row < R && col >= 0 && col < C);
        double x = 0.0;
        if (ok) x = TRANS ? M[col * ld + row] : M[row * ld + col];
        v[r] = x;
    }
    return v;
}
PDP_DEV void store_dense(double* __restrict__ M, int R, int C, int ld, int roff, int coff, int lane, const d4 v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = tile_row(lane, r) - roff, col = tile_col(lane) - coff;
        if (row >= 0 && row < R && col >= 0 && col < C) M[row * ld + col] = v[r];
    }
}
// Loop-invariant map between a tile and a dense R x C block (leading dimension ld) placed at (roff, coff) in the tile:
// built once per kernel, so that per-step loads/stores cost one address add each.
struct TileMap {
    int off[4];       // element offset in the dense block, or -1
};
PDP_DEV TileMap make_tile_map(int R, int C, int ld, int roff, int coff, int lane) {
    TileMap m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = tile_row(lane, r) - roff, col = tile_col(lane) - coff;
        m.off[r] = (row >= 0 && row < R && col >= 0 && col < C) ? row * ld + col : -1;
    }
    return m;
}
// Branch-free variant: elements outside the block are directed to one `sink` offset.  A tile that is exactly zero outside the
// block (every product of zero-padded operands is) stores zeros there, so the same map loads zeros back for those elements:
// no exec-mask branch around each register's store / load.
template <bool TRANS>
PDP_DEV TileMap make_dense_map(int R, int C, int ld, int roff, int coff, int lane) {      // TRANS: tile = (R x C block)^T
    if (!TRANS) return make_tile_map(R, C, ld, roff, coff, lane);
    TileMap m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int row = tile_row(lane, r) - roff, col = tile_col(lane) - coff;
        m.off[r] = (row >= 0 && row < C && col >= 0 && col < R) ? col * ld + row : -1;
    }
    return m;
}
// map of an R x C block (C <= 4) replicated into the four column blocks of the tile: element (row, col) <- M[row][col & 3]
PDP_DEV TileMap make_rep4_map(int R, int C, int ld, int lane) {
    TileMap m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tile_row(lane, r), c = tile_col(lane) & 3;
        m.off[r] = (row < R && c < C) ? row * ld + c : -1;
    }
    return m;
}
// the same replicated tile read from the TRANSPOSED block: element (row, col) <- Mt[col & 3][row], Mt being C x R with leading dim ld
PDP_DEV TileMap make_rep4_map_transposed(int R, int C, int ld, int lane) {
    TileMap m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tile_row(lane, r), c = tile_col(lane) & 3;
        m.off[r] = (row < R && c < C) ? c * ld + row : -1;
    }
    return m;
}
// Loads / stores through a loop-invariant map: the per-step cost is one address add per register (load_dense / store_dense redo the
// index arithmetic and bound checks of every element on every call).  Loads are branch-free (absent elements read element 0 of the
// block and are zeroed by a select), stores are predicated.
template <int NR = 4>
PDP_DEV d4 load_map(const double* __restrict__ base, const TileMap& m) {
    d4 v = zero4();
#pragma unroll
    for (int r = 0; r < NR; ++r) { const int o = m.off[r]; const double x = base[o >= 0 ? o : 0]; v[r] = o >= 0 ? x : 0.0; }
    return v;
}
template <int NR = 4>
PDP_DEV void store_map(double* __restrict__ base, const TileMap& m, const d4 v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) if (m.off[r] >= 0) base[m.off[r]] = v[r];
}
struct TileMapBytes { unsigned off[4]; };   // unsigned BYTE offsets: uniform base (SGPR pair) + 32-bit lane offset addressing, no 64-bit VALU add
PDP_DEV TileMapBytes to_bytes_sink(const TileMap& m, int sink) {
    TileMapBytes b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b.off[r] = 8u * (unsigned)(m.off[r] < 0 ? sink : m.off[r]);
    return b;
}
PDP_DEV TileMapBytes make_tile_map_sink(int R, int C, int ld, int roff, int coff, int lane, int sink) {
    TileMap m = make_tile_map(R, C, ld, roff, coff, lane);
    TileMapBytes b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b.off[r] = 8u * (unsigned)(m.off[r] < 0 ? sink : m.off[r]);
    return b;
}
template <int NR = 4>
PDP_DEV void store_all(double* __restrict__ base, const TileMapBytes& m, const d4 v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        unsigned o = m.off[r];
        asm("" : "+v"(o));      // keeps the zero-extension next to the access (hoisted out of the loop it hides the base + u32 form)
        *(double*)((char*)base + o) = v[r];
    }
}
template <int NR = 4>
PDP_DEV d4 load_all(const double* __restrict__ base, const TileMapBytes& m) {
    d4 v = zero4();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        unsigned o = m.off[r];
        asm("" : "+v"(o));
        v[r] = *(const double*)((const char*)base + o);
    }
    return v;
}
// zero every column outside [c0, c1)
PDP_DEV d4 keep_cols(const d4 v, int c0, int c1, int lane) {
    int col = tile_col(lane);
    return (col >= c0 && col < c1) ? v : zero4();
}
PDP_DEV bool tile_finite(const d4 v) {
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) ok = ok && (fabs(v[r]) <= 1.7e308);
    return ok;
}

// Cross-lane hand-off through LDS inside ONE wavefront (all kernels here run one wave per workgroup): the LDS unit
// executes a wave's DS instructions in issue order, so a later ds_read sees an earlier ds_write of any lane; what is
// needed is (a) the compiler must not reorder the accesses (it reasons per thread) and (b) outstanding DS results are
// waited for.  No s_barrier, and - unlike __syncthreads() - no vmcnt(0): pending global stores keep draining.
PDP_DEV void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// tanh for the policy networks (PDP.py:733-751): sign(x) (1 - 2 / (exp(2 |x|) + 1)), the reciprocal by v_rcp_f64 + two Newton steps.  ABSOLUTE error <= 2e-16 (the
// subtraction from 1 loses relative accuracy for |x| << 1, where the value itself is that small) - far inside the 1e-10 parity tolerance, and what the activations, their
// derivatives 1 - z^2 and every product they enter need.  The library tanh (double-double exp, (e - 1/e) / (e + 1/e)) costs ~680 cycles per call on gfx950
// (probes/mlp4t_timing.py with the call stubbed out: 23 % of the C5b step kernel); this one ~1/3 of that.  EVERY kernel that evaluates a policy uses this function, so the
// kernels stay bit-identical among themselves (tests/test_gpu_cp_mlp.py); against numpy's tanh the activations differ by <= 2 ulp of 1.
PDP_DEV double pdp_tanh(double x) {
    const double ax = fmin(fabs(x), 20.0);              // tanh(20) = 1 - 8e-18 = 1.0 in fp64; keeps exp finite
    const double d = exp(2.0 * ax) + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double t = fma(-2.0, r, 1.0);
    return x != x ? x : copysign(t, x);                 // (a NaN stays a NaN: the status words report it)
}

// broadcast lane `src`'s value to the whole wave (2 x v_readlane_b32, no LDS)
PDP_DEV double readlane_f64(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// In-register inverse of an M x M matrix (M <= 4) by Gauss-Jordan with partial pivoting, executed
// redundantly (uniformly) by every lane.  a is row-major a[i*M+j]; returns false on a tiny/non-finite pivot.
template <int M>
PDP_DEV bool inverse_small(const double* a_in, double* inv) {
    double a[M][M], b[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) { a[i][j] = a_in[i * M + j]; b[i][j] = (i == j) ? 1.0 : 0.0; }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        // partial pivoting: bring the largest |a[i][k]|, i >= k, to row k (branch-free swaps)
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            bool sw = fabs(a[i][k]) > fabs(a[k][k]);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                double t = a[k][j], s = a[i][j];
                a[k][j] = sw ? s : t; a[i][j] = sw ? t : s;
                t = b[k][j]; s = b[i][j];
                b[k][j] = sw ? s : t; b[i][j] = sw ? t : s;
            }
        }
        double piv = a[k][k];
        ok = ok && (fabs(piv) > 1e-300) && (fabs(piv) <= 1.7e308);
        double ip = 1.0 / piv;
#pragma unroll
        for (int j = 0; j < M; ++j) { a[k][j] *= ip; b[k][j] *= ip; }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i == k) continue;
            double f = a[i][k];
#pragma unroll
            for (int j = 0; j < M; ++j) { a[i][j] -= f * a[k][j]; b[i][j] -= f * b[k][j]; }
        }
    }
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) inv[i * M + j] = b[i][j];
    return ok;
}

// Inverse of an M x M matrix (M <= 4), uniform over the wave.  Fast path: cofactor (adjugate) formulas with ONE
// division (fp64 division costs ~74 cycles on gfx950; Gauss-Jordan needs M of them in sequence); guarded by the
// determinant test |det| > 1e-10 * |prod diag|, otherwise the pivoted Gauss-Jordan above takes over (uniform branch).
template <int M>
PDP_DEV bool inverse_small_fast(const double* a, double* inv) {
    double dprod = 1.0;                        // |det| <= prod diag for SPD matrices (Hadamard): det / dprod measures conditioning
#pragma unroll
    for (int i = 0; i < M; ++i) dprod *= a[i * M + i];
    double det, c[M * M];
    if constexpr (M == 1) {
        det = a[0]; c[0] = 1.0;
    } else if constexpr (M == 2) {
        det = a[0] * a[3] - a[1] * a[2];
        c[0] = a[3]; c[1] = -a[1]; c[2] = -a[2]; c[3] = a[0];
    } else if constexpr (M == 3) {
        c[0] = a[4] * a[8] - a[5] * a[7]; c[1] = a[2] * a[7] - a[1] * a[8]; c[2] = a[1] * a[5] - a[2] * a[4];
        c[3] = a[5] * a[6] - a[3] * a[8]; c[4] = a[0] * a[8] - a[2] * a[6]; c[5] = a[2] * a[3] - a[0] * a[5];
        c[6] = a[3] * a[7] - a[4] * a[6]; c[7] = a[1] * a[6] - a[0] * a[7]; c[8] = a[0] * a[4] - a[1] * a[3];
        det = a[0] * c[0] + a[1] * c[3] + a[2] * c[6];
    } else {
        // 2x2 minors of rows (0,1) and rows (2,3)
        const double s0 = a[0] * a[5] - a[4] * a[1], s1 = a[0] * a[6] - a[4] * a[2], s2 = a[0] * a[7] - a[4] * a[3];
        const double s3 = a[1] * a[6] - a[5] * a[2], s4 = a[1] * a[7] - a[5] * a[3], s5 = a[2] * a[7] - a[6] * a[3];
        const double c5 = a[10] * a[15] - a[14] * a[11], c4 = a[9] * a[15] - a[13] * a[11], c3 = a[9] * a[14] - a[13] * a[10];
        const double c2 = a[8] * a[15] - a[12] * a[11], c1 = a[8] * a[14] - a[12] * a[10], c0 = a[8] * a[13] - a[12] * a[9];
        det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
        c[0] = a[5] * c5 - a[6] * c4 + a[7] * c3;   c[1] = -a[1] * c5 + a[2] * c4 - a[3] * c3;
        c[2] = a[13] * s5 - a[14] * s4 + a[15] * s3; c[3] = -a[9] * s5 + a[10] * s4 - a[11] * s3;
        c[4] = -a[4] * c5 + a[6] * c2 - a[7] * c1;  c[5] = a[0] * c5 - a[2] * c2 + a[3] * c1;
        c[6] = -a[12] * s5 + a[14] * s2 - a[15] * s1; c[7] = a[8] * s5 - a[10] * s2 + a[11] * s1;
        c[8] = a[4] * c4 - a[5] * c2 + a[7] * c0;   c[9] = -a[0] * c4 + a[1] * c2 - a[3] * c0;
        c[10] = a[12] * s4 - a[13] * s2 + a[15] * s0; c[11] = -a[8] * s4 + a[9] * s2 - a[11] * s0;
        c[12] = -a[4] * c3 + a[5] * c1 - a[6] * c0; c[13] = a[0] * c3 - a[1] * c1 + a[2] * c0;
        c[14] = -a[12] * s3 + a[13] * s1 - a[14] * s0; c[15] = a[8] * s3 - a[9] * s1 + a[10] * s0;
    }
    if (!(fabs(det) > 1e-10 * fabs(dprod)) || !(fabs(det) <= 1.7e308)) return inverse_small<M>(a, inv);   // ill-conditioned / singular: pivoted path
    // 1 / det: hardware reciprocal + one Newton step - the IEEE division sequence is a 12-instruction dependent chain in the middle of every backward step, and det
    // (guarded above against vanishing / overflowing) is nowhere near the ranges that sequence exists for; same form as the 4 x 4 path of riccati_backward
    double id = __builtin_amdgcn_rcp(det);
    id = fma(fma(-det, id, 1.0), id, id);
#pragma unroll
    for (int i = 0; i < M * M; ++i) inv[i] = c[i] * id;
    return true;
}

// wave-level sum over the 4 lane groups that share a column (lanes l, l+16, l+32, l+48)
PDP_DEV double sum_over_rowgroups(double v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
PDP_DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace pdp
