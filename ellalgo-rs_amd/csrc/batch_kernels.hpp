// batch_kernels.hpp -- many independent small ellipsoids (n <= 128), SURVEY section 8 row f3.
//
// One ellipsoid of dimension n <= 128 cannot fill an MI355X (n = 16: Q is 2 KiB), but the reference's
// small-n users run MANY of them: BSearchAdaptor clones the search space for every probe
// (src/cutting_plane.rs:410), the quickcheck-style sweeps and BASELINE config 1 (n = 16) run one tiny
// ellipsoid after the other.  Here B ellipsoids live side by side in HBM ([B][n][n] row-major) and one
// launch applies K cuts to each of them: a workgroup owns T / n ellipsoids (at most 64; one when n > 64),
// loads its Q into LDS once, runs Ell::update_core (src/ell.rs:97-137) K times out of LDS and writes Q back
// -- 16 n^2 / K bytes of HBM traffic per update instead of the large engine's 24 n^2.
//
// Arithmetic: with the matrix in LDS there is no reason to reorder anything, so every step follows the
// reference's statement order literally -- one thread per row folds `acc += Q[i][j] * g[j]` left to right
// (Arr::dot_mv, src/arr.rs:426-442), the row-0 thread folds omega left to right (Arr::dot, :443-451), the
// rank-1 runs over j <= i with the mirror store (src/ell.rs:117-128), `no_defer_trick` multiplies every
// element afterwards (:132-135) -- and the results are BIT-IDENTICAL to the CPU path (the parity tests
// use exact equality), including for a caller-supplied non-symmetric Q.
#pragma once

#include "ell_kernels.hpp"

namespace ellhip {

struct BatchParams {
    long long B;      // ellipsoids
    int n;            // dimension, 1..128
    int pitch;        // LDS row pitch in doubles (odd: conflict-free column access)
    int epw;          // ellipsoids per workgroup
    int K;            // cuts per ellipsoid in this launch
    int no_defer_trick;
};

constexpr int BATCH_NMAX = 128;

__host__ __device__ inline int batch_pitch(int n) { return n | 1; }
// doubles of LDS one ellipsoid needs: Q, g, gt, 8 scalars
// (odd, so that the lanes of the scalar stage, one ellipsoid each, hit different LDS banks)
__host__ __device__ inline size_t batch_lds_doubles(int n) {
    return ((size_t)n * batch_pitch(n) + 2 * (size_t)n + 8) | 1;
}

// Copy the workgroup's `count` contiguous doubles between HBM ([e][i][j] dense) and LDS ([e] blocks of `per`
// doubles, rows of `pitch`): element idx = tid + t*T.  The (e, i, j) decomposition is advanced incrementally --
// no integer division in the loop (a 64-bit division costs more than the whole per-element work).
template <int T, bool TO_LDS>
__device__ __forceinline__ void batch_copy(double* __restrict__ sm, double* __restrict__ glob, int count, int n,
                                           int pitch, int per, int tid) {
    const int nn = n * n;
    int e = tid / nn, r = tid - e * nn;
    int i = r / n, j = r - i * n;
    const int de = T / nn, dr = T - de * nn;  // step of T elements = de ellipsoids + dr elements
    const int di = dr / n, dj = dr - di * n;
    for (int idx = tid; idx < count; idx += T) {
        double* l = sm + (size_t)e * per + (size_t)i * pitch + j;
        if (TO_LDS) *l = glob[idx];
        else glob[idx] = *l;
        e += de;
        i += di;
        j += dj;
        if (j >= n) {
            j -= n;
            i += 1;
        }
        if (i >= n) {
            i -= n;
            e += 1;
        }
    }
}

// Cut k of ellipsoid b: kinds / beta arrays are [K][B], grads [K][B][n]; status / tsq outputs [K][B].
template <int T>
__global__ __launch_bounds__(T) void k_batch_update(BatchParams P, double* __restrict__ Q, double* __restrict__ xc,
                                                    double* __restrict__ kappa, double* __restrict__ tsq,
                                                    const int* __restrict__ kinds, const double* __restrict__ grads,
                                                    const double* __restrict__ beta0, const int* __restrict__ has_b1,
                                                    const double* __restrict__ beta1, int* __restrict__ status_out,
                                                    double* __restrict__ tsq_out, EllCalcDev calc) {
    extern __shared__ double sm[];
    const int n = P.n, pitch = P.pitch;
    const int tid = threadIdx.x;
    const int e = tid / n, i = tid - e * n;              // local ellipsoid, row
    const long long b = (long long)blockIdx.x * P.epw + e;
    const bool active = e < P.epw && b < P.B;
    const size_t per = batch_lds_doubles(n);
    double* q = sm + (size_t)(e < P.epw ? e : 0) * per;
    double* g = q + (size_t)n * pitch;
    double* gt = g + n;
    double* sc = gt + n;  // [0] rho/omega  [1] sigma/omega  [2] scale  [3] status  [4] kappa  [5] tsq

    // ---- Q -> LDS (coalesced over the workgroup's contiguous ellipsoids)
    const long long b_first = (long long)blockIdx.x * P.epw;
    const int nb = (int)((P.B - b_first < P.epw) ? P.B - b_first : P.epw);
    double* Qwg = Q + b_first * (long long)n * n;
    batch_copy<T, true>(sm, Qwg, nb * n * n, n, pitch, (int)per, tid);
    double xci = 0.0;
    if (active) xci = xc[b * n + i];
    if (active && i == 0) {
        sc[4] = kappa[b];
        sc[5] = tsq[b];
    }
    __syncthreads();

    // Scalar stage (omega fold + EllCalc, ~700 instructions with its divisions and square roots): lane t of wave 0
    // does it for local ellipsoid t, so it is issued once per workgroup instead of once per wave.
    const bool scalar_lane = tid < P.epw && b_first + tid < P.B;
    const long long bs = b_first + tid;
    const double* g_s = sm + (size_t)(tid < P.epw ? tid : 0) * per + (size_t)n * pitch;
    const double* gt_s = g_s + n;
    double* sc_s = const_cast<double*>(gt_s) + n;
    // the next cut's gradient and cut values are requested one cut ahead, so their HBM latency overlaps the work
    double g_next = 0.0, b0_next = 0.0, b1_next = 0.0;
    int kind_next = 0, hb1_next = 0;
    if (active) g_next = grads[b * n + i];
    if (scalar_lane) {
        kind_next = kinds[bs];
        b0_next = beta0[bs];
        hb1_next = has_b1[bs];
        b1_next = beta1[bs];
    }
    for (int k = 0; k < P.K; ++k) {
        const long long cut = (long long)k * P.B + b;
        const int kind_k = kind_next, hb1_k = hb1_next;
        const double b0_k = b0_next, b1_k = b1_next;
        if (active) g[i] = g_next;
        __syncthreads();
        if (k + 1 < P.K) {
            if (active) g_next = grads[(cut + P.B) * n + i];
            if (scalar_lane) {
                const long long nxt = (long long)(k + 1) * P.B + bs;
                kind_next = kinds[nxt];
                b0_next = beta0[nxt];
                hb1_next = has_b1[nxt];
                b1_next = beta1[nxt];
            }
        }
        if (active) {  // gt = Q g                                            src/ell.rs:102
            double acc = 0.0;
            const double* row = q + (size_t)i * pitch;
#pragma unroll 8
            for (int j = 0; j < n; ++j) acc += row[j] * g[j];  // loads run ahead, the adds stay in order
            gt[i] = acc;
        }
        __syncthreads();
        if (scalar_lane) {
            const long long cut_s = (long long)k * P.B + bs;
            double omega = 0.0;  //                                           :103
#pragma unroll 8
            for (int j = 0; j < n; ++j) omega += g_s[j] * gt_s[j];
            const double kap = sc_s[4];
            const double t = kap * omega;  //                                 :105
            Coef cf;
            const int st = calc.dispatch(kind_k, b0_k, hb1_k, b1_k, t, cf);  // :106
            sc_s[5] = t;
            sc_s[3] = (double)st;
            if (st == ST_SUCCESS) {
                sc_s[0] = cf.rho / omega;    //                               :112
                sc_s[1] = cf.sigma / omega;  //                               :117
                const double knew = kap * cf.delta;  //                       :130
                if (P.no_defer_trick) {      //                               :132-135
                    sc_s[2] = knew;
                    sc_s[4] = 1.0;
                } else {
                    sc_s[2] = 1.0;
                    sc_s[4] = knew;
                }
            }
            status_out[cut_s] = st;
            if (tsq_out) tsq_out[cut_s] = t;
        }
        __syncthreads();
        const bool ok = active && sc[3] == (double)ST_SUCCESS;
        if (ok) {
            xci = xci - sc[0] * gt[i];  //                                    :113-115
            const double r = sc[1] * gt[i];
            double* row = q + (size_t)i * pitch;
#pragma unroll 4
            for (int j = 0; j <= i; ++j) {  //                                :119-128
                const double v = row[j] - r * gt[j];
                row[j] = v;
                q[(size_t)j * pitch + i] = v;  // mirror store; nobody reads the upper triangle in this phase
            }
        }
        if (P.no_defer_trick) {
            __syncthreads();
            if (ok) {
                const double s = sc[2];
                double* row = q + (size_t)i * pitch;
                for (int j = 0; j < n; ++j) row[j] = row[j] * s;
            }
        }
        __syncthreads();
    }

    // ---- LDS -> Q, xc, kappa, tsq
    if (active) xc[b * n + i] = xci;
    if (active && i == 0) {
        kappa[b] = sc[4];
        tsq[b] = sc[5];
    }
    batch_copy<T, false>(sm, Qwg, nb * n * n, n, pitch, (int)per, tid);
}

// Q[b] = diag(d[b]) or kappa-free identity; used by the constructors.
__global__ __launch_bounds__(256) void k_batch_fill(double* __restrict__ Q, long long B, int n,
                                                    const double* __restrict__ diag) {
    const long long total = B * n * n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / ((long long)n * n), r = idx - b * n * n;
        const int i = (int)(r / n), j = (int)(r - (long long)i * n);
        Q[idx] = (i == j) ? (diag ? diag[b * n + i] : 1.0) : 0.0;
    }
}

}  // namespace ellhip
