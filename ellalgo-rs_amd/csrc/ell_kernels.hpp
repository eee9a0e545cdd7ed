// ell_kernels.hpp -- hand-written CDNA4 (gfx950) kernels for Ell::update_core (src/ell.rs:97-137).
//
// One update = two streaming passes over Q with a scalar stage in between:
//
//   k_gemv    gt[r] = sum_c Q[r][c] * g[c]                 reads  8*n^2 B     (src/arr.rs:426-442)
//   k_scalar  omega = g.gt ; tsq = kappa*omega ; EllCalc ; xc -= (rho/omega) gt ; kappa *= delta
//                                                          O(n)   (src/ell.rs:103-115,130-135)
//   k_rank1   Q[r][c] = (Q[r][c] - (ratio*gt[max(r,c)])*gt[min(r,c)]) [* kappa_new]
//                                                          reads 8*n^2 B, writes 8*n^2 B
//                                                          (src/ell.rs:117-128,132-135)
//
// This is BLAS-2: 4*n^2 flop against 24*n^2 bytes, i.e. HBM-bound by a factor ~60 on MI355X, so no
// MFMA; what matters is 16-byte-per-lane coalesced streams, enough bytes in flight per CU, and no
// wasted re-reads.  Layout and mapping:
//   * Q is row-major with leading dimension ld (>= n, multiple of 2 when n is even) in HBM.
//   * a 256-thread workgroup = 4 wave64s; each wave owns RW consecutive rows and sweeps them
//     left to right, 64 lanes x 16 B = 1 KiB of one row per load instruction, RW*UNR loads in
//     flight per lane.  The vector operand (g or gt) is loaded once per column step and reused
//     for the RW rows (it lives in L2; a Q element is touched exactly once per pass).
//   * per-row dot products are reduced with a fixed xor-butterfly of wave shuffles, so the result
//     depends only on (n, VEC): the same bits for any grid size, row partition or GPU count.
//   * the rank-1 pass walks the row tiles in the opposite direction to the GEMV pass, so the tail
//     of each pass is still in the 256 MiB Infinity Cache when the next pass starts there.
//   * the symmetric update is evaluated per element as (ratio*gt[hi])*gt[lo], hi = max(r,c):
//     for symmetric Q this is bit-identical to the reference's "lower triangle, then mirror"
//     loop and needs no transposed traffic.  Multiplication and subtraction are separate
//     roundings (-ffp-contract=off), as in the reference.
#pragma once

#include <hip/hip_runtime.h>

#include "ellcalc_device.hpp"

namespace ellhip {

struct CutParams {
    int kind;
    int has_b1;
    double b0;
    double b1;
};

// Device-resident scalar state of one search space (kappa/tsq of src/ell.rs:13,15 plus what the
// scalar stage hands to the rank-1 pass).
struct DevState {
    double kappa;
    double tsq;
    double omega;
    double ratio;           // sigma / omega            (src/ell.rs:117)
    double rho_over_omega;  //                           (src/ell.rs:112)
    double scale;           // kappa_new if no_defer_trick else 1.0 (src/ell.rs:132-135)
    int status;             // CutStatus of the last cut
    int apply;              // 1 -> the second pass over Q runs
    int halted;             // queue mode: set at the first non-Success cut
    int pad_;
};

typedef double double2_t __attribute__((ext_vector_type(2)));

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = double;
    static __device__ __forceinline__ double get(const double& v, int) { return v; }
    static __device__ __forceinline__ void set(double& v, int, double x) { v = x; }
};
template <>
struct VecT<2> {
    using type = double2_t;
    static __device__ __forceinline__ double get(const double2_t& v, int i) { return i ? v.y : v.x; }
    static __device__ __forceinline__ void set(double2_t& v, int i, double x) {
        if (i) v.y = x; else v.x = x;
    }
};

// Streaming accesses: NT selects the non-temporal (`nt`) cache policy for the once-touched Q stream.
template <bool NT, typename V>
__device__ __forceinline__ V ld_stream(const double* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    else return *reinterpret_cast<const V*>(p);
}
template <bool NT, typename V>
__device__ __forceinline__ void st_stream(double* p, const V& v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
    else *reinterpret_cast<V*>(p) = v;
}

__device__ __forceinline__ double wave_allreduce_sum(double v) {
    // fixed xor butterfly over the 64 lanes: every lane ends with the same bits
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------ k_gemv ---
// gt_out[r] = sum_c Q[r*ld + c] * g[c], r in [0, nrows).  grid.x = ceil(nrows / (4*RW)).
template <int RW, int UNR, int VEC, bool NT = false>
__global__ __launch_bounds__(256) void k_gemv(const double* __restrict__ Q, long long ld, long long n,
                                              long long nrows, const double* __restrict__ g,
                                              double* __restrict__ gt_out,
                                              const DevState* __restrict__ st) {
    if (st->halted) return;
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long row_base = ((long long)blockIdx.x * 4 + wave) * RW;
    if (row_base >= nrows) return;

    const double* rp[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        long long rr = row_base + r;
        if (rr > nrows - 1) rr = nrows - 1;  // clamp: harmless duplicate read, result not stored
        rp[r] = Q + rr * ld;
    }
    double acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = 0.0;

    constexpr long long STEP = 64 * VEC;
    long long c = (long long)lane * VEC;
    const long long n_main = n - (n % (STEP * UNR));  // columns covered by full unrolled steps
    for (; c < n_main; c += STEP * UNR) {
        V gv[UNR];
        V qv[UNR][RW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            gv[u] = *reinterpret_cast<const V*>(g + c + u * STEP);
#pragma unroll
            for (int r = 0; r < RW; ++r) qv[u][r] = ld_stream<NT, V>(rp[r] + c + u * STEP);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    acc[r] += VecT<VEC>::get(qv[u][r], v) * VecT<VEC>::get(gv[u], v);
    }
    for (; c < n; c += STEP) {  // tail steps; n % VEC == 0 so a lane's VEC columns are all valid
        const V gv = *reinterpret_cast<const V*>(g + c);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const V qv = ld_stream<NT, V>(rp[r] + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[r] += VecT<VEC>::get(qv, v) * VecT<VEC>::get(gv, v);
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const double s = wave_allreduce_sum(acc[r]);
        if (lane == 0 && row_base + r < nrows) gt_out[row_base + r] = s;
    }
}

// ---------------------------------------------------------------------------------- k_scalar ---
// One workgroup of 1024 threads.  omega = sum_i g[i]*gt[i] (src/arr.rs:443-451) with a fixed
// reduction shape, then the coefficient stage and the O(n) vector updates.
__global__ __launch_bounds__(1024) void k_scalar(long long n, const double* __restrict__ g,
                                                 const double* __restrict__ gt, double* __restrict__ xc,
                                                 DevState* __restrict__ st, EllCalcDev calc,
                                                 const CutParams* __restrict__ cp, int no_defer_trick,
                                                 int queue_mode, int* __restrict__ q_status,
                                                 double* __restrict__ q_tsq) {
    __shared__ double red[16];
    __shared__ double bc_roo;
    __shared__ int bc_status;
    const int tid = threadIdx.x;
    if (st->halted) {
        if (tid == 0 && q_status) {
            *q_status = ST_UNKNOWN;
            *q_tsq = st->tsq;
        }
        return;
    }
    double s = 0.0;
    for (long long i = tid; i < n; i += 1024) s += g[i] * gt[i];
    s = wave_allreduce_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double omega = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) omega += red[w];
        const double kappa = st->kappa;
        const double tsq = kappa * omega;  // src/ell.rs:105
        Coef cf;
        const int status = calc.dispatch(cp->kind, cp->b0, cp->has_b1, cp->b1, tsq, cf);  // :106
        st->tsq = tsq;
        st->omega = omega;
        st->status = status;
        double roo = 0.0;
        if (status == ST_SUCCESS) {
            roo = cf.rho / omega;                  // :112
            st->rho_over_omega = roo;
            st->ratio = cf.sigma / omega;          // :117
            const double knew = kappa * cf.delta;  // :130
            if (no_defer_trick) {                  // :132-135
                st->scale = knew;
                st->kappa = 1.0;
            } else {
                st->scale = 1.0;
                st->kappa = knew;
            }
            st->apply = 1;
        } else {
            st->apply = 0;  // :107-109  Q, xc, kappa untouched
            if (queue_mode) st->halted = 1;
        }
        if (q_status) {
            *q_status = status;
            *q_tsq = tsq;
        }
        bc_roo = roo;
        bc_status = status;
    }
    __syncthreads();
    if (bc_status != ST_SUCCESS) return;
    const double roo = bc_roo;
    for (long long i = tid; i < n; i += 1024) xc[i] = xc[i] - roo * gt[i];  // :113-115
}

// ----------------------------------------------------------------------------------- k_rank1 ---
// Qout[r][c] = Q[r][c] - (ratio*gt[hi])*gt[lo] (then * scale if SCALE) for the local rows (Qout may be
// Q itself: every element is read and written by the same lane); row0 = global index
// of local row 0 (row-partitioned multi-GPU).  grid.x = ceil(nrows / (4*RW)); tiles are walked in
// reverse block order (see header comment).
template <int RW, int UNR, int VEC, bool SCALE, bool NT = false, bool REVERSE = true>
__global__ __launch_bounds__(256) void k_rank1(const double* Q, double* Qout, long long ld, long long n,
                                               long long nrows, long long row0,
                                               const double* __restrict__ gt,
                                               const DevState* __restrict__ st) {
    if (!st->apply) return;
    using V = typename VecT<VEC>::type;
    const double ratio = st->ratio;
    const double scale = st->scale;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long tile = REVERSE ? (long long)gridDim.x - 1 - blockIdx.x : (long long)blockIdx.x;
    const long long row_base = (tile * 4 + wave) * RW;
    if (row_base >= nrows) return;

    const double* rp[RW];
    double* wp[RW];
    long long grow[RW];
    double gtr[RW], rgr[RW];
    bool valid[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        long long rr = row_base + r;
        valid[r] = rr < nrows;
        if (!valid[r]) rr = nrows - 1;
        rp[r] = Q + rr * ld;
        wp[r] = Qout + rr * ld;
        grow[r] = row0 + rr;
        gtr[r] = gt[grow[r]];
        rgr[r] = ratio * gtr[r];  // r_qg of src/ell.rs:119
    }

    constexpr long long STEP = 64 * VEC;
    long long c = (long long)lane * VEC;
    const long long n_main = n - (n % (STEP * UNR));
    for (; c < n_main; c += STEP * UNR) {
        V gv[UNR];
        V qv[UNR][RW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            gv[u] = *reinterpret_cast<const V*>(gt + c + u * STEP);
#pragma unroll
            for (int r = 0; r < RW; ++r) qv[u][r] = ld_stream<NT, V>(rp[r] + c + u * STEP);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                V o;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const long long col = c + u * STEP + v;
                    const double gc = VecT<VEC>::get(gv[u], v);
                    const double upd = (col <= grow[r]) ? rgr[r] * gc : (ratio * gc) * gtr[r];
                    double x = VecT<VEC>::get(qv[u][r], v) - upd;  // src/ell.rs:121-123
                    if (SCALE) x = x * scale;                       // src/ell.rs:133
                    VecT<VEC>::set(o, v, x);
                }
                if (valid[r]) st_stream<NT, V>(wp[r] + c + u * STEP, o);
            }
        }
    }
    for (; c < n; c += STEP) {
        const V gv = *reinterpret_cast<const V*>(gt + c);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const V qv = ld_stream<NT, V>(rp[r] + c);
            V o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const long long col = c + v;
                const double gc = VecT<VEC>::get(gv, v);
                const double upd = (col <= grow[r]) ? rgr[r] * gc : (ratio * gc) * gtr[r];
                double x = VecT<VEC>::get(qv, v) - upd;
                if (SCALE) x = x * scale;
                VecT<VEC>::set(o, v, x);
            }
            if (valid[r]) st_stream<NT, V>(wp[r] + c, o);
        }
    }
}

// -------------------------------------------------------------------------- small helpers ----
// Q = diag(d) or identity, written on the device (Arr::eye / Arr::from_diag, src/arr.rs:40-55).
__global__ __launch_bounds__(256) void k_fill_diag(double* __restrict__ Q, long long ld, long long n,
                                                   long long nrows, long long row0,
                                                   const double* __restrict__ diag) {
    const long long total = nrows * ld;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld, c = i - r * ld;
        double v = 0.0;
        if (c == row0 + r && c < n) v = diag ? diag[c] : 1.0;
        Q[i] = v;
    }
}

// upper[r][c] = lower[c][r] for c > r: what the reference's mirror store (src/ell.rs:124-126) does
// to a caller-supplied non-symmetric matrix on its first successful update.  Unsharded only.
__global__ __launch_bounds__(256) void k_mirror_lower(double* __restrict__ Q, long long ld, long long n,
                                                      const DevState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ double tile[32][33];
    const long long bx = blockIdx.x, by = blockIdx.y;  // tile (by, bx) of the LOWER triangle: by >= bx
    if (by < bx) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const long long r = by * 32 + k, c = bx * 32 + tx;
        tile[k][tx] = (r < n && c < n) ? Q[r * ld + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long long r = bx * 32 + k, c = by * 32 + tx;  // transposed position
        if (r < n && c < n && c > r) Q[r * ld + c] = tile[tx][k];
    }
}

// EllCalc on one lane, for ellhip_calc().
__global__ void k_calc_one(EllCalcDev calc, CutParams cp, double tsq, double* __restrict__ out4) {
    Coef cf;
    const int status = calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);
    out4[0] = (double)status;
    out4[1] = cf.rho;
    out4[2] = cf.sigma;
    out4[3] = cf.delta;
}

}  // namespace ellhip
