// lowpass_kernels.hpp -- LowpassOracle (src/oracles/lowpass_oracle.rs:22-151) evaluated on the device.
//
// The oracle owns a 15n x n table `spectrum` (row i = [1, 2cos(w_i), 2cos(2 w_i), ...]) and, per call,
// walks three frequency bands in round-robin order (cursors idx1 / idx3 / idx2 persist across calls),
// computes row.x for each visited row and returns a cut at the FIRST violated constraint:
//   passband  rows [0, nwpass)        val > up_sq | val < lp_sq        (:64-82)
//   stopband  rows [nwstop, mdim)     val > sp_sq | val < 0, tracks (fmax, kmax)   (:84-107)
//   transition rows [nwpass, nwstop)  val < 0                          (:109-122)
//   then x[0] < 0                                                      (:126-130)
// The walk is a sequential search with an early exit, so the device version numbers the rows by their
// VISITING POSITION p (0 .. mdim-1: passband cycle, stopband cycle, transition cycle) and finds
//   pos = min { p : row(p) violates }
// with an atomicMin.  Workgroups take 16 positions at a time in ascending order and stop as soon as the
// minimum found so far lies before their next chunk, so the rows read are the rows the reference reads
// plus at most one grid-wide round of chunks -- never the whole table unless the walk really gets
// that far (in which case all of the stopband is needed for fmax anyway).  The result does not depend
// on which workgroup wins a race: min is order independent and every position before `pos` has been
// evaluated.  k_lp_final (one workgroup) turns `pos` into the cut (gradient = +/- the row, copied
// bit for bit; beta0/beta1 exactly as the reference's expressions), moves the cursors, and in the
// device-resident drivers also plays the caller: records x_best / gamma and picks central vs bias cut
// (src/cutting_plane.rs:300-307).
//
// Row dot products: per lane ascending columns, fixed xor butterfly -> deterministic; the reference
// (Arr::dot, src/arr.rs:443-451) folds left to right, so values agree to rounding (1e-16 relative), which
// can only change the outcome when a value sits within rounding of a threshold.
#pragma once

#include "ell_kernels.hpp"

namespace ellhip {

struct LpParams {
    long long n, ld;
    int mdim, nwpass, nwstop;
    double lp_sq, up_sq;
};

struct LpState {
    int idx1, idx2, idx3;  // round-robin cursors (lowpass_oracle.rs:9,17,18)
    int more_alt;          // :8
    int kmax;              // :20
    unsigned pos;          // first violating visiting position of the scan in flight (LP_NONE: none)
    int shrunk;            // assess_optim's second return value of the last call
    int has_cut;           // 0: assess_feas returned None
    int has_best;          // device-resident drivers: x_best is Some
    int error;             // 1: feasible point but kmax = -1 (the reference would index spectrum[-1] and panic)
    double fmax;           // :19
    double sp_sq;          // :15 (and the caller's gamma in assess_optim)
    long long rows_total;  // rows the reference's walk visits, summed over calls (work measure for the roofline)
    unsigned hint;         // visiting positions the first scan launch of the NEXT call covers (1.5 x this walk's length)
};

// A scan is two launches over the visiting order: [0, a) and [a, mdim), a = clamp(hint, LP_FIRST_MIN chunks, all).
// The walk usually ends about where the previous one did -- a few rows behind the cursor for long stretches of a
// run, thousands of rows later in others -- so the first launch is sized from the previous walk: a short walk costs
// one small launch (a full-grid launch reads a chip-wide round of rows, 134 MB at n = 4096, before anybody can stop),
// a long one runs at full width at once, and a misprediction is caught by the second launch (which leaves at once
// when the first one found the violation).
constexpr unsigned LP_FIRST_MIN = 64;  // chunks
__device__ __forceinline__ void lp_range(const LpState* ls, unsigned total, unsigned per_step, int phase,
                                         unsigned& cb, unsigned& ce) {
    const unsigned nchunks = (total + per_step - 1) / per_step;
    unsigned a = (ls->hint + per_step - 1) / per_step;
    if (a < LP_FIRST_MIN) a = LP_FIRST_MIN;
    if (a > nchunks) a = nchunks;
    cb = phase == 0 ? 0u : a;
    ce = phase == 0 ? a : nchunks;
}

constexpr unsigned LP_NONE = 0xffffffffu;
constexpr int LP_RPW = 4;     // rows per wave and step
constexpr int LP_CHUNK = 16;  // visiting positions per workgroup step (4 waves x LP_RPW rows)

// visiting position -> (band, row).  Cursor o = idx - base is in [-1, len-1]; the t-th row visited
// (t = 1, 2, ...) is base + (o + t) mod len  (the `idx += 1; if idx == end { idx = base }` of :66-69).
__device__ __forceinline__ int lp_row_of(const LpParams& P, int idx1, int idx2, int idx3, unsigned p, int& band) {
    const int P1 = P.nwpass, P3 = P.mdim - P.nwstop;
    int q = (int)p;
    if (q < P1) {
        band = 0;
        return (idx1 + 1 + q) % P1;
    }
    q -= P1;
    if (q < P3) {
        band = 1;
        return P.nwstop + (idx3 - P.nwstop + 1 + q) % P3;
    }
    q -= P3;
    band = 2;
    const int P2 = P.nwstop - P.nwpass;
    return P.nwpass + (idx2 - P.nwpass + 1 + q) % P2;
}

__device__ __forceinline__ bool lp_violates(const LpParams& P, double sp_sq, int band, double val) {
    if (band == 0) return val > P.up_sq || val < P.lp_sq;  // :72,76
    if (band == 1) return val > sp_sq || val < 0.0;        // :92,95
    return val < 0.0;                                       // :116
}

// cursor after a full cycle over a band of `len` rows starting at `base`
__device__ __forceinline__ int lp_full_cycle(int idx, int base, int len) {
    if (len <= 0) return idx;
    return (idx - base == -1) ? base + len - 1 : idx;
}

template <int VEC, bool NT>
__global__ __launch_bounds__(256) void k_lp_scan(const double* __restrict__ A, LpParams P,
                                                 const double* __restrict__ x, double* __restrict__ vals,
                                                 LpState* __restrict__ ls, const int* __restrict__ halted,
                                                 int phase) {
    if (*halted) return;
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx1 = ls->idx1, idx2 = ls->idx2, idx3 = ls->idx3;
    const double sp_sq = ls->sp_sq;
    const unsigned total = (unsigned)P.mdim;
    unsigned chunk_begin, chunk_end;  // this launch's share of the walk (lp_range)
    lp_range(ls, total, LP_CHUNK, phase, chunk_begin, chunk_end);
    for (unsigned c = chunk_begin + blockIdx.x; c < chunk_end; c += gridDim.x) {
        const unsigned p0 = c * LP_CHUNK;
        // a violation before this chunk has already been found: nothing from here on can be the first
        if (__hip_atomic_load(&ls->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p0) return;
        const double* rowp[LP_RPW];
        int rows[LP_RPW], bands[LP_RPW];
        bool valid[LP_RPW];
#pragma unroll
        for (int r = 0; r < LP_RPW; ++r) {
            const unsigned p = p0 + (unsigned)(wave * LP_RPW + r);
            valid[r] = p < total;
            rows[r] = lp_row_of(P, idx1, idx2, idx3, valid[r] ? p : p0, bands[r]);
            rowp[r] = A + (long long)rows[r] * P.ld;
        }
        double acc[LP_RPW];
#pragma unroll
        for (int r = 0; r < LP_RPW; ++r) acc[r] = 0.0;
#pragma unroll 2
        for (long long j = (long long)VEC * lane; j < P.n; j += 64 * VEC) {
            const V xv = *reinterpret_cast<const V*>(x + j);
#pragma unroll
            for (int r = 0; r < LP_RPW; ++r) {
                const V av = ld_stream<NT, V>(rowp[r] + j);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[r] += VecT<VEC>::get(av, e) * VecT<VEC>::get(xv, e);
            }
        }
#pragma unroll
        for (int r = 0; r < LP_RPW; ++r) {
            const double val = wave_allreduce_sum(acc[r]);
            if (lane == 0 && valid[r]) {
                vals[rows[r]] = val;
                if (lp_violates(P, sp_sq, bands[r], val)) atomicMin(&ls->pos, p0 + (unsigned)(wave * LP_RPW + r));
            }
        }
    }
}

// Long rows (n >= LP_WIDE_N): the 4 waves of a workgroup split the COLUMNS of LP_RPW rows (16 B per lane,
// 4 KiB of each row per step), so a workgroup step covers 4 visiting positions instead of 16 and takes a
// quarter of the time: the early exit is four times finer grained (a grid-wide round of chunks is what a
// walk that stops early pays on top of the rows it needs).  Row sums: per thread ascending columns, xor
// butterfly, ((w0+w1)+w2)+w3 -- the shape of the GEMV pass.
constexpr long long LP_WIDE_N = 1024;

template <int VEC, bool NT>
__global__ __launch_bounds__(256) void k_lp_scan_wide(const double* __restrict__ A, LpParams P,
                                                      const double* __restrict__ x, double* __restrict__ vals,
                                                      LpState* __restrict__ ls, const int* __restrict__ halted,
                                                      int phase) {
    if (*halted) return;
    using V = typename VecT<VEC>::type;
    __shared__ double red[4][LP_RPW];
    __shared__ unsigned s_pos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx1 = ls->idx1, idx2 = ls->idx2, idx3 = ls->idx3;
    const double sp_sq = ls->sp_sq;
    const unsigned total = (unsigned)P.mdim;
    unsigned chunk_begin, chunk_end;
    lp_range(ls, total, LP_RPW, phase, chunk_begin, chunk_end);
    for (unsigned c = chunk_begin + blockIdx.x; c < chunk_end; c += gridDim.x) {
        const unsigned p0 = c * LP_RPW;
        if (tid == 0) s_pos = __hip_atomic_load(&ls->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();  // one decision per workgroup (also fences the reuse of red[])
        if (s_pos < p0) return;
        const double* rowp[LP_RPW];
        int rows[LP_RPW], bands[LP_RPW];
        bool valid[LP_RPW];
#pragma unroll
        for (int r = 0; r < LP_RPW; ++r) {
            const unsigned p = p0 + (unsigned)r;
            valid[r] = p < total;
            rows[r] = lp_row_of(P, idx1, idx2, idx3, valid[r] ? p : p0, bands[r]);
            rowp[r] = A + (long long)rows[r] * P.ld;
        }
        double acc[LP_RPW];
#pragma unroll
        for (int r = 0; r < LP_RPW; ++r) acc[r] = 0.0;
#pragma unroll 2
        for (long long j = (long long)VEC * tid; j < P.n; j += 256 * VEC) {
            const V xv = *reinterpret_cast<const V*>(x + j);
#pragma unroll
            for (int r = 0; r < LP_RPW; ++r) {
                const V av = ld_stream<NT, V>(rowp[r] + j);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[r] += VecT<VEC>::get(av, e) * VecT<VEC>::get(xv, e);
            }
        }
#pragma unroll
        for (int r = 0; r < LP_RPW; ++r) {
            const double w = wave_allreduce_sum(acc[r]);
            if (lane == 0) red[wave][r] = w;
        }
        __syncthreads();
        if (tid < LP_RPW && valid[tid]) {
            const double val = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
            // rows[] / bands[] are indexed with a compile-time r above; pick this thread's entry
            int row = rows[0], band = bands[0];
#pragma unroll
            for (int r = 1; r < LP_RPW; ++r)
                if (tid == r) {
                    row = rows[r];
                    band = bands[r];
                }
            vals[row] = val;
            if (lp_violates(P, sp_sq, band, val)) atomicMin(&ls->pos, p0 + (unsigned)tid);
        }
    }
}

// (fmax, kmax) over the stopband rows at visiting positions [lo, hi): first strict maximum in visiting
// order (`if val > self.fmax`, :100-103).  All 256 threads call; result valid in thread 0.
__device__ inline void lp_stopband_max(const LpParams& P, int idx1, int idx2, int idx3, const double* vals,
                                       unsigned lo, unsigned hi, double& fmax, int& kmax) {
    __shared__ double s_val[256];
    __shared__ unsigned s_pos[256];
    const int tid = threadIdx.x;
    double bv = -__builtin_inf();
    unsigned bp = LP_NONE;
    for (unsigned p = lo + tid; p < hi; p += 256) {
        int band;
        const int row = lp_row_of(P, idx1, idx2, idx3, p, band);
        const double v = vals[row];
        if (v > bv) {  // ascending p within a thread: the earliest position wins ties
            bv = v;
            bp = p;
        }
    }
    s_val[tid] = bv;
    s_pos[tid] = bp;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) {
            const double ov = s_val[tid + off];
            const unsigned op = s_pos[tid + off];
            if (op != LP_NONE && (s_pos[tid] == LP_NONE || ov > s_val[tid] || (ov == s_val[tid] && op < s_pos[tid]))) {
                s_val[tid] = ov;
                s_pos[tid] = op;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        fmax = -__builtin_inf();
        kmax = -1;
        if (s_pos[0] != LP_NONE) {
            int band;
            fmax = s_val[0];
            kmax = lp_row_of(P, idx1, idx2, idx3, s_pos[0], band);
        }
    }
    __syncthreads();
}

// mode 0 = assess_feas (:58-133), 1 = assess_optim (:139-150).  `drv` != nullptr: this call is one
// iteration of a device-resident cutting-plane loop on that search space (src/cutting_plane.rs:205-227,
// 286-313): x is the space's centre, x_best / gamma are recorded here, a feasible point ends a
// feasibility loop.
__global__ __launch_bounds__(256) void k_lp_final(const double* __restrict__ A, LpParams P,
                                                  const double* __restrict__ x, const double* __restrict__ vals,
                                                  LpState* __restrict__ ls, double* __restrict__ g,
                                                  CutParams* __restrict__ cp, double* __restrict__ xbest, int mode,
                                                  DevState* __restrict__ drv, const int* __restrict__ halted) {
    if (*halted) return;
    __shared__ int sh_row, sh_sign, sh_copy_x, sh_e0;
    const int tid = threadIdx.x;
    const unsigned pos = ls->pos;
    const int idx1 = ls->idx1, idx2 = ls->idx2, idx3 = ls->idx3;
    const int P1 = P.nwpass, P3 = P.mdim - P.nwstop, P2 = P.nwstop - P.nwpass;
    __syncthreads();  // everyone has read the scan result before thread 0 rewrites the state

    // stopband maximum over what the walk visited (:86-87 reset, :100-103)
    double fmax = ls->fmax;
    int kmax = ls->kmax;
    if (pos == LP_NONE || pos >= (unsigned)P1) {
        const unsigned hi = (pos == LP_NONE || pos >= (unsigned)(P1 + P3)) ? (unsigned)(P1 + P3) : pos;
        lp_stopband_max(P, idx1, idx2, idx3, vals, (unsigned)P1, hi, fmax, kmax);
    }
    if (tid == 0) {
        int row = -1, sign = 1, copy_x = 0, e0 = 0;
        CutParams c;
        c.kind = 0;  // update_bias_cut unless the caller obtained a better gamma
        c.has_b1 = 0;
        c.b0 = 0.0;
        c.b1 = 0.0;
        int shrunk = 0, has_cut = 1;
        int n1 = idx1, n2 = idx2, n3 = idx3;
        if (pos != LP_NONE) {
            int band;
            row = lp_row_of(P, idx1, idx2, idx3, pos, band);
            const double val = vals[row];
            if (band == 0) {
                n1 = row;
                if (val > P.up_sq) {  // :72-75
                    c.b0 = val - P.up_sq;
                    c.b1 = val - P.lp_sq;
                } else {              // :76-82
                    sign = -1;
                    c.b0 = -val + P.lp_sq;
                    c.b1 = -val + P.up_sq;
                }
                c.has_b1 = 1;
            } else if (band == 1) {
                n1 = lp_full_cycle(idx1, 0, P1);
                n3 = row;
                if (val > ls->sp_sq) {  // :92-94
                    c.b0 = val - ls->sp_sq;
                    c.b1 = val;
                } else {                // :95-99
                    sign = -1;
                    c.b0 = -val;
                    c.b1 = -val + ls->sp_sq;
                }
                c.has_b1 = 1;
            } else {  // :116-121
                n1 = lp_full_cycle(idx1, 0, P1);
                n3 = lp_full_cycle(idx3, P.nwstop, P3);
                n2 = row;
                sign = -1;
                c.b0 = -val;
            }
            ls->more_alt = 1;
        } else {
            n1 = lp_full_cycle(idx1, 0, P1);
            n3 = lp_full_cycle(idx3, P.nwstop, P3);
            n2 = lp_full_cycle(idx2, P.nwpass, P2);
            ls->more_alt = 0;  // :124
            const double x0 = x[0];
            if (x0 < 0.0) {  // :126-130
                e0 = 1;
                c.b0 = -x0;
            } else if (mode == 0) {
                has_cut = 0;  // feasible: None (:132)
                if (drv) {    // cutting_plane_feas returns (Some(xc), niter) (src/cutting_plane.rs:216-219)
                    copy_x = 1;
                    drv->halted = 1;
                    drv->stop = STOP_FEASIBLE;
                }
            } else {  // :145-150
                if (kmax < 0) {
                    ls->error = 1;
                    has_cut = 0;
                    if (drv) {
                        drv->halted = 1;
                        drv->stop = STOP_STATUS;
                    }
                } else {
                    row = kmax;
                    c.b0 = 0.0;
                    c.b1 = fmax;
                    c.has_b1 = 1;
                    c.kind = 1;  // update_central_cut (src/cutting_plane.rs:304)
                    ls->sp_sq = fmax;
                    shrunk = 1;
                    copy_x = drv ? 1 : 0;  // x_best = Some(space.xc()) (src/cutting_plane.rs:303)
                }
            }
        }
        ls->idx1 = n1;
        ls->idx2 = n2;
        ls->idx3 = n3;
        ls->fmax = fmax;
        ls->kmax = kmax;
        ls->shrunk = shrunk;
        ls->has_cut = has_cut;
        if (copy_x) ls->has_best = 1;
        ls->pos = LP_NONE;  // ready for the next scan
        const long long walked = (pos == LP_NONE) ? (long long)P.mdim : (long long)pos + 1;
        ls->rows_total += walked;
        ls->hint = (unsigned)(walked + walked / 2);
        *cp = c;
        sh_row = row;
        sh_sign = sign;
        sh_copy_x = copy_x;
        sh_e0 = e0;
    }
    __syncthreads();
    const int row = sh_row, sign = sh_sign;
    if (sh_e0) {
        for (long long j = tid; j < P.n; j += 256) g[j] = (j == 0) ? -1.0 : 0.0;
    } else if (row >= 0) {
        const double* a = A + (long long)row * P.ld;
        if (sign > 0)
            for (long long j = tid; j < P.n; j += 256) g[j] = a[j];
        else
            for (long long j = tid; j < P.n; j += 256) g[j] = -a[j];
    }
    if (sh_copy_x)
        for (long long j = tid; j < P.n; j += 256) xbest[j] = x[j];
}

}  // namespace ellhip
