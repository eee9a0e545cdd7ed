// lmi_kernels.hpp -- LDLTMgr (src/oracles/ldlt_mgr.rs:3-141) and the LMI oracles built on it
// (src/oracles/lmi_oracle.rs, lmi0_oracle.rs) on the device.  SURVEY section 8, row f4.
//
// One oracle call is, for an m x m matrix pencil F(x) = B - sum_k x_k F_k with n variables:
//   form     A[i][j] = B[i][j] - sum_k F_k[i][j] x_k           lower triangle, streamed over the n matrices
//   factor   LDL^T with an exit at the first pivot <= 0          (the feasibility decision)
//   witness  back substitution for v with v'Av = -ep < 0
//   quad     g_k = v' F_k v  for every k                        second stream over the n matrices
// The two streams are HBM bound (n m^2 8 bytes each at most); the factorisation is O(m^3 / 3).
//
// Rounding contract.  The DECISION (which pivot fails, `pos`) must not depend on how the device orders its
// sums, so `form` and `factor` keep the reference's per-element order exactly:
//   form    s = B[i][j]; for k ascending: s -= F_k[i][j] * x_k                     (lmi_oracle.rs:29-35)
//   factor  T[i][j] = A[i][j] - s,  s = sum_{k<j} L[i][k] T[j][k] folded from 0.0 in ascending k   (ldlt_mgr.rs:34-47)
// The reference walks row by row; the same per-element folds are obtained here by a right-looking blocked
// sweep in which every element owns its accumulator s (kept in the lower triangle of `storage` until the
// element's column is factored): panels are applied in ascending order and, inside a panel, k runs ascending,
// so each s sees its products in the reference's order -- `storage`, `pos` and ep are bit-identical to the CPU
// path on every row the reference touches.  witness / quad feed continuous outputs only (the cut gradient) and
// use parallel reductions (rounding-level differences).
#pragma once

#include "ell_kernels.hpp"

namespace ellhip {

constexpr int LMI_NB = 32;       // factorisation panel width
constexpr int LMI_FORM_W = 256;  // columns formed at a time (lazily: a failing pivot stops the forming too)

struct LmiState {
    int pos1;       // 0 while every pivot so far is > 0; else index of the failing row + 1 (pub pos.1)
    int pad;
    double ep;      // -storage[pos1-1][pos1-1]
};

// A[i][j] for j in [c0, c1), i in [max(j, c0) .. m): one row per workgroup, one column per thread.
// mode 0: s = B; s -= F_k x_k   (LMIOracle)      mode 1: s = 0; s += F_k x_k   (LMI0Oracle)
__global__ __launch_bounds__(LMI_FORM_W) void k_lmi_form(const double* __restrict__ F, const double* __restrict__ B,
                                                         const double* __restrict__ x, double* __restrict__ A,
                                                         long long m, long long n, long long c0, int mode,
                                                         const LmiState* __restrict__ st) {
    if (st->pos1) return;
    const long long i = c0 + blockIdx.x;
    const long long j = c0 + threadIdx.x;
    if (i >= m || j >= m || j > i) return;
    const long long off = i * m + j, mm = m * m;
    double s = (mode == 0) ? B[off] : 0.0;
    if (mode == 0) {
#pragma unroll 8
        for (long long k = 0; k < n; ++k) s = s - F[k * mm + off] * x[k];
    } else {
#pragma unroll 8
        for (long long k = 0; k < n; ++k) s = s + F[k * mm + off] * x[k];
    }
    A[off] = s;
}

// storage lower triangle + diagonal = 0.0 (the accumulators s start from 0.0, ldlt_mgr.rs:42)
__global__ __launch_bounds__(256) void k_ldlt_clear(double* __restrict__ S, long long m, LmiState* __restrict__ st) {
    const long long total = m * m;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long i = idx / m, j = idx - i * m;
        if (j <= i) S[idx] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->pos1 = 0;
        st->ep = 0.0;
    }
}

// Diagonal block [k0, k1) x [k0, k1): one wave, lane r = row k0 + r.  Column by column:
//   t = A[i][j] - (s continued over the block's earlier columns);  D[j] = t on the diagonal (exit if <= 0);
//   below it  storage[j][i] = t  (kept for later, :37)  and  storage[i][j] = t / D[j]  (L[i][j], :38-39).
__global__ __launch_bounds__(64) void k_ldlt_diag(const double* __restrict__ A, double* __restrict__ S, long long m,
                                                  long long k0, LmiState* __restrict__ st) {
    if (st->pos1) return;
    __shared__ double Ls[LMI_NB][LMI_NB + 1], Ts[LMI_NB][LMI_NB + 1], Dj;
    __shared__ int failed;
    const int r = threadIdx.x;
    const long long nb = (m - k0 < LMI_NB) ? m - k0 : LMI_NB;
    const long long i = k0 + r;
    const bool row = r < nb;
    if (r == 0) failed = 0;
    __syncthreads();
    for (int jj = 0; jj < nb; ++jj) {
        const long long j = k0 + jj;
        double t = 0.0;
        if (row && r >= jj) {
            double s = S[i * m + j];
            for (int kk = 0; kk < jj; ++kk) s += Ls[r][kk] * Ts[jj][kk];
            t = A[i * m + j] - s;
            if (r == jj) {
                S[j * m + j] = t;  // self.storage[i * ndim + i] = diag (:49)
                Dj = t;
                if (t <= 0.0) {    // :50-53
                    failed = 1;
                    st->pos1 = (int)(j + 1);
                    st->ep = -t;
                }
            }
        }
        __syncthreads();
        if (failed) return;
        if (row && r > jj) {
            const double l = t / Dj;
            Ts[r][jj] = t;
            Ls[r][jj] = l;
            S[j * m + i] = t;
            S[i * m + j] = l;
        }
        __syncthreads();
    }
}

// Rows below the diagonal block (i >= k1), one row per thread, the block's T values and pivots in LDS.
__global__ __launch_bounds__(128) void k_ldlt_panel(const double* __restrict__ A, double* __restrict__ S, long long m,
                                                    long long k0, const LmiState* __restrict__ st) {
    if (st->pos1) return;
    __shared__ double Ts[LMI_NB][LMI_NB + 1], D[LMI_NB];
    const long long k1 = k0 + LMI_NB;  // only called when the block is full (k1 <= m)
    for (int idx = threadIdx.x; idx < LMI_NB * LMI_NB; idx += 128) {
        const int jj = idx / LMI_NB, kk = idx - jj * LMI_NB;
        if (kk < jj) Ts[jj][kk] = S[(k0 + kk) * m + (k0 + jj)];  // T[j][k] lives at storage[k][j]
        if (kk == jj) D[jj] = S[(k0 + jj) * m + (k0 + jj)];
    }
    __syncthreads();
    const long long i = k1 + (long long)blockIdx.x * 128 + threadIdx.x;
    if (i >= m) return;
    double l[LMI_NB];
#pragma unroll
    for (int jj = 0; jj < LMI_NB; ++jj) {
        double s = S[i * m + k0 + jj];
#pragma unroll
        for (int kk = 0; kk < jj; ++kk) s += l[kk] * Ts[jj][kk];
        const double t = A[i * m + k0 + jj] - s;
        l[jj] = t / D[jj];
        S[(k0 + jj) * m + i] = t;
    }
#pragma unroll
    for (int jj = 0; jj < LMI_NB; ++jj) S[i * m + k0 + jj] = l[jj];
}

// Trailing accumulators: s[i][j] += sum_{k in panel, ascending} L[i][k] T[j][k]  for k1 <= j <= i < m.
// 64 x 64 tiles, 4 x 4 elements per thread, the panel's L rows and T rows staged in LDS.
__global__ __launch_bounds__(256) void k_ldlt_update(double* __restrict__ S, long long m, long long k0,
                                                     const LmiState* __restrict__ st) {
    if (st->pos1) return;
    const long long k1 = k0 + LMI_NB;
    const long long ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const long long i0 = k1 + ti * 64, j0 = k1 + tj * 64;
    if (i0 >= m) return;
    __shared__ double Lt[64][LMI_NB + 1];  // L[i0 + a][k0 + kk]
    __shared__ double Tt[LMI_NB][64 + 1];  // T[j0 + b][k0 + kk] = storage[k0 + kk][j0 + b]
    for (int idx = threadIdx.x; idx < 64 * LMI_NB; idx += 256) {
        const int a = idx / LMI_NB, kk = idx - a * LMI_NB;
        Lt[a][kk] = (i0 + a < m) ? S[(i0 + a) * m + k0 + kk] : 0.0;
        const int kk2 = idx / 64, b = idx - kk2 * 64;
        Tt[kk2][b] = (j0 + b < m) ? S[(k0 + kk2) * m + j0 + b] : 0.0;
    }
    __syncthreads();
    const int ta = (threadIdx.x >> 4) * 4, tb = (threadIdx.x & 15) * 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long long i = i0 + ta + a, j = j0 + tb + b;
            acc[a][b] = (i < m && j <= i) ? S[i * m + j] : 0.0;
        }
#pragma unroll 4
    for (int kk = 0; kk < LMI_NB; ++kk) {
        double la[4], tbv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) la[a] = Lt[ta + a][kk];
#pragma unroll
        for (int b = 0; b < 4; ++b) tbv[b] = Tt[kk][tb + b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += la[a] * tbv[b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const long long i = i0 + ta + a, j = j0 + tb + b;
            if (i < m && j <= i) S[i * m + j] = acc[a][b];  // never the strict upper triangle: it holds T values
        }
}

// witness (:99-112): v[p-1] = 1; v[c] = -sum_{k > c} L[k][c] v[k].  One workgroup; thread t owns the columns
// c = t, t + 1024, ...; at step k every owner of a column c < k adds L[k][c] v[k] (row k of storage: coalesced).
constexpr int LMI_WIT_PER = 8;  // columns per thread: m <= 8192
__global__ __launch_bounds__(1024) void k_lmi_witness(const double* __restrict__ S, long long m,
                                                      double* __restrict__ v, const LmiState* __restrict__ st) {
    const int p = st->pos1;
    if (!p) return;
    __shared__ double vk;
    const int t = threadIdx.x;
    double s[LMI_WIT_PER];
#pragma unroll
    for (int q = 0; q < LMI_WIT_PER; ++q) s[q] = 0.0;
    for (long long c = t; c < m; c += 1024) v[c] = 0.0;  // Arr::new(ndim): zeros outside [start, pos)
    if (t == 0) vk = 1.0;
    __syncthreads();
    if (t == 0) v[p - 1] = 1.0;
    for (int k = p - 1; k >= 1; --k) {
        const double w = vk;
        const double* row = S + (long long)k * m;
#pragma unroll
        for (int q = 0; q < LMI_WIT_PER; ++q) {
            const int c = t + q * 1024;
            if (c < k) s[q] += row[c] * w;
        }
        __syncthreads();  // everyone has read vk
        const int own = (k - 1) & 1023, oq = (k - 1) >> 10;
        if (t == own) {
            double val = 0.0;
#pragma unroll
            for (int q = 0; q < LMI_WIT_PER; ++q)
                if (q == oq) val = -s[q];
            vk = val;
            v[k - 1] = val;
        }
        __syncthreads();
    }
}

// partial[k][chunk] = sum over rows i of the chunk, i < p, of sum_j (v_i F_k[i][j]) v_j      (:116-125)
constexpr int LMI_QUAD_CHUNKS = 8;
__global__ __launch_bounds__(256) void k_lmi_quad(const double* __restrict__ F, long long m,
                                                  const double* __restrict__ v, double* __restrict__ partial,
                                                  const LmiState* __restrict__ st) {
    const int p = st->pos1;
    if (!p) return;
    __shared__ double red[4];
    const long long k = blockIdx.x;
    const int chunk = blockIdx.y;
    const long long rows_per = (p + LMI_QUAD_CHUNKS - 1) / LMI_QUAD_CHUNKS;
    const long long r0 = chunk * rows_per, r1 = (r0 + rows_per < p) ? r0 + rows_per : p;
    const double* Fk = F + k * m * m;
    double acc = 0.0;
    for (long long i = r0; i < r1; ++i) {
        const double vi = v[i];
        const double* row = Fk + i * m;
        for (long long j = threadIdx.x; j < p; j += 256) acc += (vi * row[j]) * v[j];
    }
    acc = wave_allreduce_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[k * LMI_QUAD_CHUNKS + chunk] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void k_lmi_quad_reduce(long long n, const double* __restrict__ partial,
                                                         double* __restrict__ g, int mode,
                                                         const LmiState* __restrict__ st) {
    if (!st->pos1) return;
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < LMI_QUAD_CHUNKS; ++c) s += partial[k * LMI_QUAD_CHUNKS + c];
    g[k] = (mode == 0) ? s : -s;  // lmi_oracle.rs:42 / lmi0_oracle.rs:31
}

// sqrt (:129-140): R[i][i] = sqrt(D_i), R[i][j] = storage[j][i] * sqrt(D_i) for j > i, zeros below
__global__ __launch_bounds__(256) void k_ldlt_sqrt(const double* __restrict__ S, long long m, double* __restrict__ R) {
    const long long total = m * m;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long i = idx / m, j = idx - i * m;
        double r = 0.0;
        if (j >= i) {
            const double val = __builtin_sqrt(S[i * m + i]);
            r = (j == i) ? val : S[j * m + i] * val;
        }
        R[idx] = r;
    }
}

}  // namespace ellhip
