// ellstable_kernels.hpp -- CDNA4 kernels for EllStable::update_core (src/ell_stable.rs:52-125).
//
// State (one n x n row-major buffer M, leading dimension ld, as in the reference):
//   diagonal      d[i]    = M[i][i]          ("inv(D)" entries,                     :72-75)
//   strict upper  U[j][i] = M[j][i], j < i   (= L[i][j] of the unit lower factor,   :65)
//   strict lower  S[i][j] = M[i][j], j < i   (scratch: the products U[j][i]*w[j],   :66)
// The reference is BUG-COMPATIBLY reproduced: the back substitution reads the scratch
// triangle (:96) and the factor update adds beta2 * S[l][j] (:116).
//
// One update, blocked by B = 64 rows/columns (block kb = indices [64 kb, 64 kb + 64)):
//
//   forward   w = L^-1 g            right-looking: for each block, a one-wave register solve of the
//             S <- U .* w            64x64 diagonal block, then a grid-wide panel update of every
//                                    column to its right (reads 64 rows of U coalesced, writes the
//                                    products transposed through LDS as full 128-byte lines of S).
//                                    The workgroup that owns the next block's columns runs that
//                                    block's diagonal solve in the same launch.     reads 4n^2, writes 4n^2
//   mid       z = d.*w, gg = z.*w (in the diagonal solves), omega = sum gg, tsq, EllCalc, kappa,
//             t_j = omega/mu + prefix(gg), beta2_j = z_j/t_j, d_j *= t_{j-1}/t_j        O(n)
//   backward  q = z; for j descending: q[t] -= S[j][t]*q[j], t < j   (rows of S, coalesced)  reads 4n^2
//   xc        xc -= (rho/omega) q
//   factor    U[j][l] += beta2_j * S[l][j], l > j   (64x64 tiles transposed through LDS)  reads 4n^2 + RMW 8n^2
//
// 24 n^2 algorithmic bytes per successful update, like Ell; but the two triangular solves are
// length-n dependency chains (n sequential multiply-subtract steps), which bounds this variant well
// below the HBM roofline.  Summation order inside a column differs from the reference's
// strict left-to-right order only by the grouping into 16-row partial sums (deterministic).
#pragma once

#include <hip/hip_runtime.h>

#include "ell_kernels.hpp"

namespace ellhip {

constexpr int SB = 64;        // block size of the triangular solves
constexpr int SPANEL = 128;   // columns per panel workgroup (2 per lane)
constexpr int SLDS_PAD = 18;  // doubles per LDS tile row (16 + 2: keeps 16-byte alignment)

// ------------------------------------------------------------------------------ forward ------
// One wave: finish w for block J0..J0+63 given its partial values, park the products in S, emit z, gg.
// Lane l owns column J0 + l.  src/ell_stable.rs:61-83 restricted to the diagonal block.
__device__ __forceinline__ void st_fwd_diag_wave(double* __restrict__ M, long long ld, long long n,
                                                 long long J0, double wi, double* __restrict__ w,
                                                 double* __restrict__ z, double* __restrict__ gg) {
    const int lane = threadIdx.x & 63;
    const long long c = J0 + lane;
    const bool live = c < n;
    const long long cc = live ? c : n - 1;
    double u[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        long long r = J0 + j;
        if (r > n - 1) r = n - 1;
        u[j] = M[r * ld + cc];  // U[J0+j][c]; only j < lane is used
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const double wj = __shfl(wi, j, 64);
        if (lane > j) {
            const double v = u[j] * wj;  // :65
            u[j] = v;                    // parked product, :66
            wi = wi - v;                 // :67
        }
    }
    if (live) {
        // S[c][J0 + j] for j < lane: this lane's own row, contiguous
        double* srow = M + c * ld + J0;
#pragma unroll
        for (int j = 0; j < SB; ++j)
            if (j < lane) srow[j] = u[j];
        const double d = M[c * ld + c];
        const double zi = wi * d;  // :74
        w[c] = wi;
        z[c] = zi;
        gg[c] = zi * wi;  // :81
    }
}

__global__ __launch_bounds__(64) void k_st_fwd_first(double* __restrict__ M, long long ld, long long n,
                                                     const double* __restrict__ g, double* __restrict__ w,
                                                     double* __restrict__ z, double* __restrict__ gg,
                                                     const DevState* __restrict__ st) {
    if (st->halted) return;
    const int lane = threadIdx.x;
    const double wi = (lane < n) ? g[lane] : 0.0;
    st_fwd_diag_wave(M, ld, n, 0, wi, w, z, gg);
}

// Panel update for block kb (rows J0..J0+63, already final in w) over columns >= J0 + 64, 128 columns
// per workgroup; 4 waves take 16 rows each.  Workgroup 0 then solves the next diagonal block.
// `wsrc` is g for columns that have not been touched yet (first panel) -- handled by k_st_copy.
__global__ __launch_bounds__(256) void k_st_fwd_step(double* __restrict__ M, long long ld, long long n,
                                                     long long kb, double* __restrict__ w,
                                                     double* __restrict__ z, double* __restrict__ gg,
                                                     const DevState* __restrict__ st) {
    if (st->halted) return;
    __shared__ __attribute__((aligned(16))) double tile[4][SPANEL * SLDS_PAD];
    __shared__ double part[4][SPANEL];
    __shared__ double wnext[SPANEL];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long J0 = kb * SB;
    const long long Jend = J0 + SB;
    const long long c0 = Jend + (long long)blockIdx.x * SPANEL;
    const long long c = c0 + 2 * lane;  // this lane's two columns c, c+1 (ld is even: 16-byte aligned)
    const bool live0 = c < n, live1 = c + 1 < n;

    // rows of this wave: J0 + 16 wv + r
    double p0 = 0.0, p1 = 0.0;
    double2_t u[16];
    const long long cl = live0 ? c : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = J0 + 16 * wv + r;  // < Jend <= n here because c0 >= Jend exists only if Jend < n
        u[r] = *reinterpret_cast<const double2_t*>(M + row * ld + cl);
    }
    double* t = tile[wv];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double wj = w[J0 + 16 * wv + r];  // wave-uniform
        const double v0 = u[r].x * wj;
        const double v1 = u[r].y * wj;
        p0 += v0;
        p1 += v1;
        t[(2 * lane) * SLDS_PAD + r] = v0;
        t[(2 * lane + 1) * SLDS_PAD + r] = v1;
    }
    part[wv][2 * lane] = p0;
    part[wv][2 * lane + 1] = p1;
    __syncthreads();
    // S[col][J0 + 16 wv .. +16) for the 128 columns of this workgroup: 8 lanes x 16 B = one 128-byte line
    {
        const int piece = lane & 7;  // which 16-byte piece of the 128-byte line
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int col_local = 8 * k + (lane >> 3);
            const long long col = c0 + col_local;
            if (col < n) {
                const double2_t v = *reinterpret_cast<const double2_t*>(&t[col_local * SLDS_PAD + 2 * piece]);
                *reinterpret_cast<double2_t*>(M + col * ld + J0 + 16 * wv + 2 * piece) = v;
            }
        }
    }
    // w[col] -= (((p_0 + p_1) + p_2) + p_3): threads 0..127
    if (threadIdx.x < SPANEL) {
        const long long col = c0 + threadIdx.x;
        double wn = 0.0;
        if (col < n) {
            const double s = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) +
                             part[3][threadIdx.x];
            wn = w[col] - s;
            w[col] = wn;
        }
        wnext[threadIdx.x] = wn;
    }
    (void)live1;
    if (blockIdx.x != 0 || Jend >= n) return;
    __syncthreads();
    if (wv == 0) st_fwd_diag_wave(M, ld, n, Jend, wnext[lane], w, z, gg);
}

// ---------------------------------------------------------------------------------- mid -------
// omega, tsq, EllCalc, kappa; prefix sums t_j; beta2_j; diagonal rescale; q <- z.
// One workgroup of 1024 threads.  src/ell_stable.rs:78-90,107-113,120-122.
__global__ __launch_bounds__(1024) void k_st_mid(double* __restrict__ M, long long ld, long long n,
                                                 const double* __restrict__ z, const double* __restrict__ gg,
                                                 double* __restrict__ q, double* __restrict__ beta2,
                                                 DevState* __restrict__ st, EllCalcDev calc,
                                                 const CutParams* __restrict__ cp_dev, CutParams cp_val,
                                                 int queue_mode, int* __restrict__ q_status,
                                                 double* __restrict__ q_tsq) {
    __shared__ double red[16];
    __shared__ double tot[1024];
    __shared__ double bc[2];
    __shared__ int bc_status;
    const int tid = threadIdx.x;
    if (st->halted) {
        if (tid == 0 && q_status) {
            *q_status = ST_UNKNOWN;
            *q_tsq = st->tsq;
        }
        return;
    }
    // chunked sums: thread t owns the contiguous chunk [t*m, (t+1)*m)
    const long long m = (n + 1023) / 1024;
    const long long lo = (long long)tid * m;
    const long long hi = (lo + m < n) ? lo + m : n;
    double s = 0.0;
    for (long long i = lo; i < hi; ++i) s += gg[i];
    // inclusive scan of the chunk totals inside each wave (shuffle ladder), then across the 16 waves
    const int lane = tid & 63;
    double x = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) red[tid >> 6] = x;
    __syncthreads();
    double wave_off = 0.0;
    for (int k = 0; k < (tid >> 6); ++k) wave_off += red[k];
    tot[tid] = wave_off + (x - s);  // exclusive prefix of this thread's chunk
    if (tid == 0) {
        double omega = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) omega += red[k];
        const double kappa = st->kappa;
        const double tsq = kappa * omega;  // :85
        Coef cf;
        const CutParams cp = cp_dev ? *cp_dev : cp_val;
        const int status = calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);  // :86
        st->tsq = tsq;
        st->omega = omega;
        st->status = status;
        double t0 = 0.0;
        if (status == ST_SUCCESS) {
            st->rho_over_omega = cf.rho / omega;              // :101
            const double mu = cf.sigma / (1.0 - cf.sigma);    // :107
            t0 = omega / mu;                                   // :108
            st->kappa = kappa * cf.delta;                      // :122
            st->apply = 1;
        } else {
            st->apply = 0;  // :88-90 (the scratch triangle has already been rewritten, as in the reference)
            if (queue_mode) st->halted = 1;
        }
        if (q_status) {
            *q_status = status;
            *q_tsq = tsq;
        }
        bc[0] = t0;
        bc_status = status;
    }
    __syncthreads();
    if (bc_status != ST_SUCCESS) return;
    double told = bc[0] + tot[tid];  // t_{lo-1}
    for (long long j = lo; j < hi; ++j) {
        const double tnew = told + gg[j];      // :111
        beta2[j] = z[j] / tnew;                // :112
        M[j * ld + j] = M[j * ld + j] * (told / tnew);  // :113 / :121
        q[j] = z[j];                           // :93
        told = tnew;
    }
}

// ------------------------------------------------------------------------------ backward ------
// One wave: finish q for block J0..J0+63 (its partial values already hold the contributions of all
// later blocks): for j descending, q[t] -= S[J0+j][J0+t] * q[J0+j], t < j.   src/ell_stable.rs:93-98
__device__ __forceinline__ void st_bwd_diag_wave(const double* __restrict__ M, long long ld, long long n,
                                                 long long J0, double qi, double* __restrict__ q) {
    const int lane = threadIdx.x & 63;
    const long long c = J0 + lane;
    const bool live = c < n;
    const long long cc = live ? c : n - 1;
    double s[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        long long r = J0 + j;
        if (r > n - 1) r = n - 1;
        s[j] = M[r * ld + cc];  // S[J0+j][c]; only j > lane is used
    }
#pragma unroll
    for (int j = SB - 1; j >= 1; --j) {
        const double qj = __shfl(qi, j, 64);
        if (lane < j && J0 + j < n) qi = qi - s[j] * qj;
    }
    if (live) q[c] = qi;
}

__global__ __launch_bounds__(64) void k_st_bwd_last(const double* __restrict__ M, long long ld, long long n,
                                                    long long kb_last, double* __restrict__ q,
                                                    const DevState* __restrict__ st) {
    if (!st->apply) return;
    const long long c = kb_last * SB + threadIdx.x;
    const double qi = (c < n) ? q[c] : 0.0;
    st_bwd_diag_wave(M, ld, n, kb_last * SB, qi, q);
}

// Panel for block kb (rows J0..J0+63 of S, q final there) over columns t < J0, 128 per workgroup, then
// the workgroup that owns block kb-1 solves it.
__global__ __launch_bounds__(256) void k_st_bwd_step(const double* __restrict__ M, long long ld, long long n,
                                                     long long kb, double* __restrict__ q,
                                                     const DevState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ double part[4][SPANEL];
    __shared__ double qnext[SPANEL];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long J0 = kb * SB;
    const long long c0 = (long long)blockIdx.x * SPANEL;
    const long long c = c0 + 2 * lane;  // columns c, c+1 < J0 (J0 is a multiple of 64, so both or neither)
    const bool live = c < J0;
    double p0 = 0.0, p1 = 0.0;
    if (live) {
        double2_t sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long long row = J0 + 16 * wv + r;
            if (row > n - 1) row = n - 1;
            sv[r] = *reinterpret_cast<const double2_t*>(M + row * ld + c);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = J0 + 16 * wv + r;
            const double qj = (row < n) ? q[row] : 0.0;  // wave-uniform
            p0 += sv[r].x * qj;
            p1 += sv[r].y * qj;
        }
    }
    part[wv][2 * lane] = p0;
    part[wv][2 * lane + 1] = p1;
    __syncthreads();
    if (threadIdx.x < SPANEL) {
        const long long col = c0 + threadIdx.x;
        double qn = 0.0;
        if (col < J0) {
            const double s = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) +
                             part[3][threadIdx.x];
            qn = q[col] - s;
            q[col] = qn;
        }
        qnext[threadIdx.x] = qn;
    }
    // block kb-1 = columns [J0-64, J0): owned by workgroup (J0-64)/128
    const long long Jp = J0 - SB;
    if (Jp < 0 || (long long)blockIdx.x != Jp / SPANEL) return;
    __syncthreads();
    if (wv == 0) st_bwd_diag_wave(M, ld, n, Jp, qnext[(Jp - c0) + lane], q);
}

// xc -= (rho/omega) q   (src/ell_stable.rs:101-104)
__global__ __launch_bounds__(256) void k_st_xc(long long n, const double* __restrict__ q,
                                               double* __restrict__ xc, const DevState* __restrict__ st) {
    if (!st->apply) return;
    const double roo = st->rho_over_omega;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        xc[i] = xc[i] - roo * q[i];
}

// -------------------------------------------------------------------------------- factor ------
// U[j][l] += beta2[j] * S[l][j] for l > j (src/ell_stable.rs:114-117), 64x64 tiles: tile (tj, tl),
// tl >= tj, reads S rows l in tile tl / columns j in tile tj, transposes through LDS and updates the
// U rows j / columns l.  The last row j = n-1 has no l > j, so the reference's `0..last_idx` bound
// needs no special case.
__global__ __launch_bounds__(256) void k_st_factor(double* __restrict__ M, long long ld, long long n,
                                                   const double* __restrict__ beta2,
                                                   const DevState* __restrict__ st) {
    if (!st->apply) return;
    const long long tj = blockIdx.y, tl = blockIdx.x;
    if (tl < tj) return;
    __shared__ __attribute__((aligned(16))) double tile[64][66];
    const int tx = threadIdx.x & 31;  // column pair
    const int ty = threadIdx.x >> 5;  // 0..7
    const long long j0 = tj * 64, l0 = tl * 64;
    // load S[l0 + r][j0 + 2tx .. +1] -> tile[r][2tx..]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = ty + 8 * k;
        const long long l = l0 + r, j = j0 + 2 * tx;
        double2_t v = {0.0, 0.0};
        if (l < n && j < n) v = *reinterpret_cast<const double2_t*>(M + l * ld + j);  // ld even: j+1 <= ld-1
        tile[r][2 * tx] = v.x;
        tile[r][2 * tx + 1] = v.y;
    }
    __syncthreads();
    // update U[j0 + r][l0 + 2tx .. +1]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = ty + 8 * k;
        const long long j = j0 + r, l = l0 + 2 * tx;
        if (j >= n || l >= n) continue;
        const double b = beta2[j];
        double* p = M + j * ld + l;
        double2_t uv = *reinterpret_cast<double2_t*>(p);
        bool touched = false;
        if (l > j) {
            uv.x = uv.x + b * tile[2 * tx][r];
            touched = true;
        }
        if (l + 1 > j && l + 1 < n) {
            uv.y = uv.y + b * tile[2 * tx + 1][r];
            touched = true;
        }
        if (touched) {
            // never write an element with l <= j (diagonal / scratch of the same tile)
            if (l > j)
                *reinterpret_cast<double2_t*>(p) = uv;
            else
                p[1] = uv.y;
        }
    }
}

}  // namespace ellhip
