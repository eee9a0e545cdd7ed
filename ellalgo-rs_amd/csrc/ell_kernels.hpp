// ell_kernels.hpp -- hand-written CDNA4 (gfx950) kernels for Ell::update_core (src/ell.rs:97-137).
//
// One update = two streaming passes over Q with a scalar stage in between:
//
//   k_sweep<GV>  gt[r] = sum_c Q[r][c] * g[c]              reads  8*n^2 B     (src/arr.rs:426-442)
//   k_scalar     omega = g.gt ; tsq = kappa*omega ; EllCalc ; xc -= (rho/omega) gt ; kappa *= delta
//                                                          O(n)   (src/ell.rs:103-115,130-135)
//   k_sweep<R1>  Q[r][c] = (Q[r][c] - (ratio*gt[max(r,c)])*gt[min(r,c)]) [* kappa_new]
//                                                          reads 8*n^2 B, writes 8*n^2 B
//                                                          (src/ell.rs:117-128,132-135)
//   k_sweep<R1,GV> both at once for consecutive updates (pipelined schedule, 16*n^2 B per update)
//
// and, for the recorded ("deferred") schedules that the larger handles run by default (see MAXPEND below):
//
//   k_symv + k_symv_reduce<NP>   y = Q_base*g through the lower triangle only (4*n^2 B), plus the scalar stage's
//                                dot products g.y and v_j.g
//   k_sweep_gemv_dots            the same for handles without the lower-triangle schedule: full-row GEMV with the
//                                v_j.g partial sums formed beside it
//   k_scalar_apply_def<NP,GY>    gt = y - sum_j (c_j v_j.g) v_j, omega, EllCalc, xc; records (c, gt) as update NP'
//   k_apply_lower / k_sweep_apply   one pass applies the 8 / 16 recorded updates element by element, in order
//
// This is BLAS-2: 4*n^2 flop against 24*n^2 bytes, i.e. HBM-bound by a factor ~60 on MI355X, so no
// MFMA; what matters is 16-byte-per-lane coalesced streams, enough bytes in flight per CU, and no
// wasted re-reads.  Layout and mapping:
//   * Q is row-major with leading dimension ld (>= n, multiple of 2 when n is even) in HBM.
//   * a 256-thread workgroup = 4 wave64s owns RW consecutive rows and sweeps them left to right,
//     4 KiB of each row per step (16 B per lane), RW*UNR loads in flight per lane.  The vector
//     operand (g or gt) is loaded once per column step and reused for the RW rows (it lives in
//     L2; a Q element is touched exactly once per pass).
//   * per-row dot products: per-thread sequential accumulation, fixed xor-butterfly of wave
//     shuffles, fixed order across the 4 waves, so the result depends only on (n, VEC): the same
//     bits for any grid size, row partition or GPU count.
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
    int halted_in;          // snapshot of `halted` for kernels whose lead workgroup rewrites it
    double kappa_in;        // snapshot of kappa taken by k_scalar_dot for k_scalar_apply
    int solve_err;          // EllStable persistent solves: nonzero if a bounded flag wait timed out
    int npend;              // deferred mode: number of rank-1 updates recorded but not yet applied to Q
    // device-resident cutting-plane loops (lowpass_kernels.hpp; src/cutting_plane.rs:205-227,286-313)
    int stop;               // why the loop halted: 0 running, 1 cut not Success, 2 tsq < tol, 3 oracle: feasible
    double tol;             // Options::tolerance; negative = no tolerance test (plain queues)
    long long niter;        // iterations completed without stopping = the `niter` the reference returns
};
constexpr int STOP_NONE = 0, STOP_STATUS = 1, STOP_TOL = 2, STOP_FEASIBLE = 3;

// What a direct update hands to its caller through pinned host memory (k_publish, and the scalar stage's own tail below).
struct LiveMirror {
    DevState st;
    unsigned long long seq;
};
static_assert(sizeof(DevState) % sizeof(long long) == 0, "k_publish copies DevState in 8-byte words");

// The tail of a scalar-stage kernel on a LIVE update (ellhip_update / ellhip_cut; csrc/ellhip_capi.hip live_publish / live_wait): every
// workgroup pushes ITS slice of the centre into pinned host memory right where it has just written it, and the last one to arrive adds the
// scalar state and the update's sequence number (system-scope release) -- the host polls that word.  Saves the separate k_publish launch
// behind the stage (4.6 us of a 56 us iteration at n = 4096, 6.8 of 252 at n = 16384).  All threads of the workgroup call it.
__device__ __forceinline__ void live_tail(const DevState* st, const double* xc, long long lo, long long hi, LiveMirror* m,
                                          double* h_xc, unsigned long long seq, unsigned* arrived) {
    __shared__ int live_last;
    __syncthreads();  // (this workgroup's stores to its slice of xc are complete)
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x)
        h_xc[i] = __hip_atomic_load(xc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        live_last = (old + 1 == gridDim.x);
        if (live_last) __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!live_last) return;
    if (threadIdx.x < (int)(sizeof(DevState) / sizeof(long long)))   // (the lead workgroup's stores: agent-scope loads, not this CU's L1)
        reinterpret_cast<long long*>(&m->st)[threadIdx.x] =
            __hip_atomic_load(reinterpret_cast<const long long*>(st) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Queue-mode bookkeeping of one cut (lead thread of the scalar stage): the loop test of
// src/cutting_plane.rs:222,308 `status != Success || tsq < tolerance`, evaluated on the device so
// that a whole batch of iterations can be enqueued without a host round trip.
__device__ __forceinline__ void queue_bookkeeping(DevState* st, int status, double tsq, int queue_mode) {
    if (!queue_mode) return;
    if (status != 0) {
        st->halted = 1;
        st->stop = STOP_STATUS;
    } else if (tsq < st->tol) {
        st->halted = 1;  // this cut's shrink still has to be applied: see lowpass driver / DevState.apply
        st->stop = STOP_TOL;
    } else {
        st->niter += 1;
    }
}

// Deferred rank-1 updates ("lazy shrink").  Instead of rewriting Q after every cut, up to MAXPEND
// updates are kept as pairs (c_j = sigma_j/omega_j, v_j = the gt of that cut):
//     Q_true = Q_base - sum_j c_j v_j v_j^T
// The GEMV of the next cut needs only Q_base*g (a READ-ONLY pass) plus O(n * pending) corrections
//     gt = Q_base*g - sum_j (c_j (v_j.g)) v_j ,   omega = g.gt = g.(Q_base*g) - sum_j c_j (v_j.g)^2
// and one pass applies all pending updates element by element IN ORDER, so every matrix element goes
// through exactly the roundings of the reference's one-update-at-a-time loop (src/ell.rs:117-128).
// Per update: 8 n^2 (1 + 1/MAXPEND) + ... bytes instead of 16 n^2 (pipelined) or 24 n^2 (two-pass).
// Unused slots hold c_j = 0 and v_j = 0, which makes every formula above an exact no-op for them.
constexpr int MAXPEND = 48;  // capacity of the pending-update buffers; the depth in force (NP) is 8, 16 or 24 (48: only inside
                             // a queue run whose cuts go through the group stage, ELLHIP_OPT_QUEUE_DEPTH)

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

// Streaming accesses: NT selects the non-temporal (`nt`) cache policy for the LOADS of the once-touched
// Q stream.  Stores always use the default policy: measured on MI355X at n = 16384, nt loads + plain
// stores 5.40 TB/s, nt loads + nt stores 5.29, plain loads + nt stores 5.35 (profiles/r01).
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

// ----------------------------------------------------------------------------------- k_sweep ---
// ONE streaming pass over the local rows of Q, in up to two roles at once:
//   R1 (rank-1 shrink)   Qout[r][c] = (Q[r][c] - (ratio*gt[hi])*gt[lo]) [* scale]     src/ell.rs:117-135
//   GV (GEMV)            gv_out[r]  = sum_c Qout[r][c] * gvec[c]                       src/arr.rs:426-442
//   R1 only  = the second pass of an update            (16 n^2 bytes: 8 read + 8 write)
//   GV only  = the first pass of an update             ( 8 n^2 bytes read)
//   R1 + GV  = "fused": the shrink of update k and the GEMV of update k+1 in one pass (16 n^2 bytes
//              for what the two-pass schedule moves 24 n^2 for); legal because the next gradient only
//              depends on xc_{k+1}, which the scalar stage of update k has already produced.
// Mapping ("column split"): a 256-thread workgroup owns RW consecutive rows; its 4 waves interleave
// 1 KiB column chunks, so one step of the workgroup touches 4 KiB contiguous of each row, 16 B per
// lane, RW*UNR loads in flight per lane.  A row's dot product is accumulated per thread in ascending
// column order (thread t: columns VEC*t + 256*VEC*k), reduced inside each wave by a fixed xor
// butterfly and across the 4 waves as ((s0+s1)+s2)+s3: the bits depend only on (n, VEC), not on RW,
// UNR, the grid, the row partition or the number of GPUs, and are the same in every role.
// `reverse` flips the order in which row tiles are visited (consecutive passes run in opposite
// directions, so a pass starts where the previous one ended: Infinity-Cache reuse).
template <int RW, int UNR, int VEC, bool NT, bool R1, bool GV, bool SCALE>
__device__ __forceinline__ void sweep_rows(const double* Q, double* Qout, long long ld, long long n,
                                           long long nrows, long long row0, long long row_base,
                                           const double* __restrict__ gt, const double* __restrict__ gvec,
                                           double* __restrict__ gv_out, double ratio, double scale,
                                           double (*red)[RW]) {
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const double* rp[RW];
    double* wp[RW];
    long long grow[RW];
    double gtr[RW], rgr[RW], acc[RW];
    bool valid[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        long long rr = row_base + r;
        valid[r] = rr < nrows;
        if (!valid[r]) rr = nrows - 1;  // clamp: harmless duplicate read, nothing stored
        rp[r] = Q + rr * ld;
        wp[r] = Qout + rr * ld;
        grow[r] = row0 + rr;
        acc[r] = 0.0;
        if (R1) {
            gtr[r] = gt[grow[r]];
            rgr[r] = ratio * gtr[r];  // r_qg of src/ell.rs:119
        }
    }
    constexpr long long STEP = 256 * VEC;
    long long c = (long long)threadIdx.x * VEC;
    const long long n_main = n - (n % (STEP * UNR));  // columns covered by full unrolled steps

    auto element = [&](int r, long long col, double qx, double gc) -> double {
        // (ratio*gt[hi])*gt[lo], hi = max(row, col): the reference's lower-triangle value, mirrored
        const double upd = (col <= grow[r]) ? rgr[r] * gc : (ratio * gc) * gtr[r];
        double x = qx - upd;      // src/ell.rs:121-123 (two roundings: no FMA)
        if (SCALE) x = x * scale;  // src/ell.rs:133
        return x;
    };

    for (; c < n_main; c += STEP * UNR) {
        V gv[UNR], hv[UNR];
        V qv[UNR][RW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (R1) gv[u] = *reinterpret_cast<const V*>(gt + c + u * STEP);
            if (GV) hv[u] = *reinterpret_cast<const V*>(gvec + c + u * STEP);
#pragma unroll
            for (int r = 0; r < RW; ++r) qv[u][r] = ld_stream<NT, V>(rp[r] + c + u * STEP);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                V o = qv[u][r];
                if (R1) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        VecT<VEC>::set(o, v, element(r, c + u * STEP + v, VecT<VEC>::get(qv[u][r], v),
                                                     VecT<VEC>::get(gv[u], v)));
                    if (valid[r]) st_stream<false, V>(wp[r] + c + u * STEP, o);
                }
                if (GV) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[r] += VecT<VEC>::get(o, v) * VecT<VEC>::get(hv[u], v);
                }
            }
        }
    }
    for (; c < n; c += STEP) {  // tail steps; n % VEC == 0 so a thread's VEC columns are all valid
        V gv, hv;
        if (R1) gv = *reinterpret_cast<const V*>(gt + c);
        if (GV) hv = *reinterpret_cast<const V*>(gvec + c);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            V o = ld_stream<NT, V>(rp[r] + c);
            if (R1) {
                const V qin = o;
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    VecT<VEC>::set(o, v, element(r, c + v, VecT<VEC>::get(qin, v), VecT<VEC>::get(gv, v)));
                if (valid[r]) st_stream<false, V>(wp[r] + c, o);
            }
            if (GV) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[r] += VecT<VEC>::get(o, v) * VecT<VEC>::get(hv, v);
            }
        }
    }
    if (GV) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double s = wave_allreduce_sum(acc[r]);
            if (lane == 0) red[wave][r] = s;
        }
        __syncthreads();
        if (threadIdx.x < RW && row_base + threadIdx.x < nrows) {
            const int r = threadIdx.x;
            gv_out[row_base + r] = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
        }
    }
}

// grid.x = ceil(nrows / RW).  gv_out already points at this shard's first row.
template <int RW, int UNR, int VEC, bool NT, bool R1, bool GV, bool SCALE>
__global__ __launch_bounds__(256) void k_sweep(const double* Q, double* Qout, long long ld, long long n,
                                               long long nrows, long long row0,
                                               const double* __restrict__ gt,
                                               const double* __restrict__ gvec,
                                               double* __restrict__ gv_out,
                                               const DevState* __restrict__ st, int reverse) {
    __shared__ double red[4][RW];
    if (st->halted) return;  // queue mode: a previous cut failed, nothing further runs
    const bool apply = R1 && st->apply != 0;
    if (!GV && !apply) return;  // failed cut: Q stays untouched (src/ell.rs:107-109)
    const long long tile = reverse ? (long long)gridDim.x - 1 - blockIdx.x : (long long)blockIdx.x;
    const long long row_base = tile * RW;
    if (row_base >= nrows) return;
    if (R1 && apply) {
        sweep_rows<RW, UNR, VEC, NT, true, GV, SCALE>(Q, Qout, ld, n, nrows, row0, row_base, gt, gvec, gv_out,
                                                      st->ratio, st->scale, red);
    } else if (GV) {
        sweep_rows<RW, UNR, VEC, NT, false, true, false>(Q, Qout, ld, n, nrows, row0, row_base, gt, gvec, gv_out,
                                                         0.0, 1.0, red);
    }
}

// ------------------------------------------------------------------------------------ k_symv ---
// GEMV through the lower triangle only (deferred mode, unsharded handle, Q bit-symmetric -- which the
// update formula guarantees): y = Q g with every stored element Q[r][c], c < r, used twice,
//     y[r] += Q[r][c] g[c]   (row sums)        y[c] += Q[r][c] g[r]   (column sums),
// so a pass reads 4 n^2 bytes instead of 8 n^2.  Tiles: strip I = SYMV_H rows, segment J = SYMV_SEG
// columns (only segments that reach the diagonal or lie left of it).  A workgroup keeps, per thread,
// the column sums of its own columns over the strip's rows (no cross-thread traffic) and reduces each
// row's partial sum once; it writes
//     rowpart[J][r]  = sum over the segment's columns c <= r of Q[r][c] g[c]
//     colpart[I][c]  = sum over the strip's rows r > c of Q[r][c] g[r]
// and k_symv_reduce forms y[i] = sum_J rowpart[J][i] + sum_{I >= i/H} colpart[I][i] in a fixed order.
#ifndef ELLHIP_SYMV_H
#define ELLHIP_SYMV_H 64
#endif
#ifndef ELLHIP_SYMV_SEG
#define ELLHIP_SYMV_SEG 2048
#endif
constexpr int SYMV_H = ELLHIP_SYMV_H;      // (macros: tuning builds only, see profiles/r01/tune_symv.txt)
constexpr int SYMV_SEG = ELLHIP_SYMV_SEG;  // default segment width; row shards with few tiles use SYMV_SEG_SMALL
constexpr int SYMV_SEG_SMALL = 512;

// SEG: segment width (2048 by default; 512 where a shard's trapezoid would give fewer tiles than the GPU has
// room for -- at P = 8 a rank has ~130 tiles of 64 x 2048 but ~520 of 64 x 512).
// Agent-scope relaxed (sc1) accesses for partial sums that are consumed INSIDE the launch that produced them
// (k_symv_tail): write-through stores, loads that are never served from the reading CU's L1 (cdna_hip_programming.md
// Guideline 16).
__device__ __forceinline__ void ho_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ho_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte write-through store (sc1).  Two 8-byte atomic stores per pair (64 lanes x 8 bytes at a 16-byte stride, twice)
// made a tile's 16 KiB of column sums cost more than the tile itself: k_symv_tail 0.32 ms against 0.19 + 0.012 ms.
__device__ __forceinline__ void ho_store2(double* p, double2_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// One tile (strip I, segment J) of the lower-triangle GEMV; returns false when the tile lies wholly above the diagonal
// or outside the local rows (nothing done).  HANDOFF: the partial sums are read by another workgroup of the same launch.
template <int RW, bool NT, int ABL, int SEG, bool HANDOFF>
__device__ __forceinline__ bool symv_tile(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                          long long nrows, const double* __restrict__ g, double* __restrict__ rowpart,
                                          double* __restrict__ colpart, long long I, long long J, double (*red)[SYMV_H]) {
    constexpr int SYMV_NCH = SEG / 512;  // 16-byte column chunks per thread
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // Row shard (symmetric multi-GPU mode): Q holds the rows [row0, row0 + nrows) only; I counts local strips,
    // every row / column index below is global.  Unsharded: row0 = 0, nrows = n.
    const long long r0 = row0 + I * SYMV_H;
    const long long c0 = J * SEG;
    const long long rend = row0 + nrows;  // one past the last local row
    if (r0 >= rend || c0 > r0 + SYMV_H - 1) return false;  // nothing at or left of the diagonal in this segment
    const long long rlast = (r0 + SYMV_H - 1 < rend - 1) ? r0 + SYMV_H - 1 : rend - 1;
    Q -= row0 * ld;  // so that Q + r * ld addresses global row r
    const bool full = c0 + SEG - 1 < r0;  // every column of the segment is strictly left of every row

    long long ck[SYMV_NCH];
    double2_t gc[SYMV_NCH], accc[SYMV_NCH];
#pragma unroll
    for (int k = 0; k < SYMV_NCH; ++k) {
        ck[k] = c0 + 512 * k + 2 * (long long)threadIdx.x;
        const bool in = ck[k] <= rlast;  // n is even and ck is even: ck <= n - 2, so the pair is inside the matrix
        gc[k] = in ? *reinterpret_cast<const double2_t*>(g + ck[k]) : double2_t{0.0, 0.0};
        accc[k] = double2_t{0.0, 0.0};
    }
    for (int rg = 0; rg < SYMV_H / RW; ++rg) {
        double accr[RW];
        double gr[RW];
        long long rr[RW];
        double2_t q[RW][SYMV_NCH];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            rr[r] = r0 + rg * RW + r;
            const bool rv = rr[r] < rend;
            gr[r] = rv ? g[rr[r]] : 0.0;
            accr[r] = 0.0;
            const double* row = Q + (rv ? rr[r] : rend - 1) * ld;
#pragma unroll
            for (int k = 0; k < SYMV_NCH; ++k) {
                // load the pair when its first column is at or left of the diagonal of this row
                if (rv && (full || ck[k] <= rr[r])) q[r][k] = ld_stream<NT, double2_t>(row + ck[k]);
                else q[r][k] = double2_t{0.0, 0.0};
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int k = 0; k < SYMV_NCH; ++k) {
                double qx = q[r][k].x, qy = q[r][k].y;
                if (!full) {
                    // pair loaded iff ck <= r; its second element is above the diagonal when ck + 1 > r
                    if (ck[k] + 1 > rr[r]) qy = 0.0;
                }
                accr[r] += qx * gc[k].x;
                accr[r] += qy * gc[k].y;
                // column sums take strictly-below-diagonal elements only (the diagonal is counted once, in the row sum)
                const double cx = (full || ck[k] < rr[r]) ? qx : 0.0;
                const double cy = (full || ck[k] + 1 < rr[r]) ? qy : 0.0;
                if (ABL != 2) {
                    accc[k].x += cx * gr[r];
                    accc[k].y += cy * gr[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double s = (ABL == 1) ? accr[r] : wave_allreduce_sum(accr[r]);
            if (lane == 0) red[wave][rg * RW + r] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < SYMV_H && r0 + threadIdx.x < rend) {
        const int r = threadIdx.x;
        const double v = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
        if (HANDOFF) ho_store(rowpart + J * n + r0 + r, v); else rowpart[J * n + r0 + r] = v;
    }
#pragma unroll
    for (int k = 0; k < SYMV_NCH; ++k)
        if (ck[k] <= rlast) {
            if (HANDOFF) {
                ho_store2(colpart + I * n + ck[k], accc[k]);
            } else {
                *reinterpret_cast<double2_t*>(colpart + I * n + ck[k]) = accc[k];
            }
        }
    return true;
}

template <int RW, bool NT, int ABL = 0, int SEG = SYMV_SEG>  // ABL: timing-only ablations for tools/tune_ell.hip
__global__ __launch_bounds__(256) void k_symv(const double* __restrict__ Q, long long ld, long long n,
                                              long long row0, long long nrows,
                                              const double* __restrict__ g, double* __restrict__ rowpart,
                                              double* __restrict__ colpart, const DevState* __restrict__ st) {
    __shared__ double red[4][SYMV_H];
    if (st->halted) return;
    // grid = (strips, segments), strips in descending order (the widest first).  Measured alternative: making
    // the segment index the fast one (row-major traversal) is 19 % slower (profiles/r01/tune_symv.txt).
    symv_tile<RW, NT, ABL, SEG, false>(Q, ld, n, row0, nrows, g, rowpart, colpart, (long long)gridDim.x - 1 - blockIdx.x,
                                       (long long)blockIdx.y, red);
}

// ------------------------------------------------------------------------------ k_symv_multi ---
// The lower-triangle GEMV for LV queued gradients in ONE pass over Q_base: Y = Q_base [g_0 ... g_{LV-1}].  On the
// recorded schedule the products of the next cuts all refer to the same Q_base (the cuts in between are recorded, not
// applied; the scalar stage corrects each y with them), and the queue holds the gradients already, so a group of LV
// cuts costs one read of the lower triangle instead of LV: (4 / LV) n^2 bytes per update.  Per vector the arithmetic
// is k_symv's, operation for operation (same tile, same thread-to-element map, same order of every sum), and vector l
// writes its own set of partial sums (rowpart + l * rowpart_stride, colpart + l * colpart_stride) for k_symv_reduce:
// every y_l is bit-identical to what k_symv yields for g_l alone.  The vector ALU and the registers (LV column sums and
// LV gradient pairs per column pair stay live over the tile) carry this to LV = 3 (0.25 ms per pass at n = 16384,
// against 0.19 for one vector; LV = 4: 0.39); larger groups go to the matrix cores (k_symm_mfma).  Tried and dropped
// (tools/experiments/symv_multi.hip, profiles/r03/symv_multi_*): next row group's loads issued ahead from a second
// register buffer, and all RW * LV row sums of a row group through one halving butterfly (bit-identical, 10 shuffles
// instead of 48) -- the extra live registers cost more than they bought.
template <int RW, bool NT, int SEG, int LV>
__global__ __launch_bounds__(256) void k_symv_multi(const double* __restrict__ Q, long long ld, long long n,
                                                    long long row0, long long nrows, const double* __restrict__ g,
                                                    long long g_stride, double* __restrict__ rowpart,
                                                    double* __restrict__ colpart, long long rowpart_stride,
                                                    long long colpart_stride, const DevState* __restrict__ st) {
    __shared__ double red[LV][4][SYMV_H];
    if (st->halted) return;
    constexpr int SYMV_NCH = SEG / 512;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = (long long)blockIdx.y;
    const long long r0 = row0 + I * SYMV_H;
    const long long c0 = J * SEG;
    const long long rend = row0 + nrows;
    if (r0 >= rend || c0 > r0 + SYMV_H - 1) return;
    const long long rlast = (r0 + SYMV_H - 1 < rend - 1) ? r0 + SYMV_H - 1 : rend - 1;
    Q -= row0 * ld;
    const bool full = c0 + SEG - 1 < r0;

    long long ck[SYMV_NCH];
    double2_t gc[LV][SYMV_NCH], accc[LV][SYMV_NCH];
#pragma unroll
    for (int k = 0; k < SYMV_NCH; ++k) {
        ck[k] = c0 + 512 * k + 2 * (long long)threadIdx.x;
        const bool in = ck[k] <= rlast;
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            gc[l][k] = in ? *reinterpret_cast<const double2_t*>(g + l * g_stride + ck[k]) : double2_t{0.0, 0.0};
            accc[l][k] = double2_t{0.0, 0.0};
        }
    }
    for (int rg = 0; rg < SYMV_H / RW; ++rg) {
        double accr[LV][RW];
        double gr[LV][RW];
        long long rr[RW];
        double2_t q[RW][SYMV_NCH];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            rr[r] = r0 + rg * RW + r;
            const bool rv = rr[r] < rend;
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                gr[l][r] = rv ? g[l * g_stride + rr[r]] : 0.0;
                accr[l][r] = 0.0;
            }
            const double* row = Q + (rv ? rr[r] : rend - 1) * ld;
#pragma unroll
            for (int k = 0; k < SYMV_NCH; ++k) {
                if (rv && (full || ck[k] <= rr[r])) q[r][k] = ld_stream<NT, double2_t>(row + ck[k]);
                else q[r][k] = double2_t{0.0, 0.0};
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int k = 0; k < SYMV_NCH; ++k) {
                double qx = q[r][k].x, qy = q[r][k].y;
                if (!full) {
                    if (ck[k] + 1 > rr[r]) qy = 0.0;
                }
                const double cx = (full || ck[k] < rr[r]) ? qx : 0.0;
                const double cy = (full || ck[k] + 1 < rr[r]) ? qy : 0.0;
#pragma unroll
                for (int l = 0; l < LV; ++l) {
                    accr[l][r] += qx * gc[l][k].x;
                    accr[l][r] += qy * gc[l][k].y;
                    accc[l][k].x += cx * gr[l][r];
                    accc[l][k].y += cy * gr[l][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                const double sum = wave_allreduce_sum(accr[l][r]);
                if (lane == 0) red[l][wave][rg * RW + r] = sum;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < SYMV_H && r0 + threadIdx.x < rend) {
        const int r = threadIdx.x;
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const double v = ((red[l][0][r] + red[l][1][r]) + red[l][2][r]) + red[l][3][r];
            rowpart[l * rowpart_stride + J * n + r0 + r] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < SYMV_NCH; ++k)
        if (ck[k] <= rlast) {
#pragma unroll
            for (int l = 0; l < LV; ++l)
                *reinterpret_cast<double2_t*>(colpart + l * colpart_stride + I * n + ck[k]) = accc[l][k];
        }
}

// y[i] = sum_{J <= i/SEG} rowpart[J][i] + sum_{I >= i/H} colpart[I][i]   (fixed order)
// One workgroup per 128 columns: lane = column pair (16-byte loads, 1 KiB per wave-instruction), wave w
// takes the strips I0 + w, I0 + w + 4, ...; the four wave sums are combined as ((w0+w1)+w2)+w3.
//
// NP > 0 (unsharded handle, deferred depth NP): the kernel also produces the dot products the scalar stage needs,
// so that k_scalar_dot_def's launch disappears from the update's dependency chain:
//     partial[b][0]     = sum over this workgroup's 128 columns of g[i] * y[i]
//     partial[b][1 + j] = sum over the same columns of pend[j][i] * g[i]          (v_j . g, j < NP)
// (per lane x-then-y, xor butterfly over the wave; k_scalar_apply_def adds the workgroups' values in index order)
// and workgroup 0 takes the halted / kappa snapshots k_scalar_dot_def would have taken.
// The reduction of one block of 128 columns (blk) -- the body of k_symv_reduce.  HANDOFF: the partial sums were written
// by other workgroups of THIS launch (k_symv_tail): they are read with agent-scope loads.
// WT: y and the partial dot products are stored write-through (agent scope): other workgroups of the SAME launch read
// them after an in-launch wait (k_symv_reduce_scalar).
template <int NP, bool HANDOFF, bool WT = false>
__device__ __forceinline__ void symv_reduce_block(long long blk, long long n, long long row0, long long nrows,
                                                  long long seg, const double* __restrict__ rowpart,
                                                  const double* __restrict__ colpart, double* __restrict__ y,
                                                  const double* __restrict__ g, const double* __restrict__ pend,
                                                  double* __restrict__ partial, double2_t (*part)[64], int np_used = NP) {
    // (np_used: the recorded slots that hold a vector right now; the dot products with the empty ones are not formed -- nobody
    // reads them -- and their 8 n bytes per slot not loaded: a group stage right after an apply pass has none)
    auto ldp = [](const double* p) -> double2_t {
        if constexpr (HANDOFF) return double2_t{ho_load(p), ho_load(p + 1)};
        else return *reinterpret_cast<const double2_t*>(p);
    };
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long i = blk * 128 + 2 * lane;  // columns i, i+1 (n is even)
    const long long nstrips = (nrows + SYMV_H - 1) / SYMV_H;     // local strips
    // operands of the dot products: requested first, so their latency hides behind the strip loop
    constexpr int NPW = (NP + 3) / 4;  // pending vectors per wave (wave w takes j = w, w + 4, ...)
    double2_t gi = {0.0, 0.0};
    double2_t pv[NPW > 0 ? NPW : 1];
    if (NP > 0) {
        if (i < n) gi = *reinterpret_cast<const double2_t*>(g + i);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int j = wave + 4 * k;
            pv[k] = (i < n && j < NP && j < np_used) ? *reinterpret_cast<const double2_t*>(pend + (long long)j * n + i)
                                                     : double2_t{0.0, 0.0};
        }
    }
    // A row shard yields its PARTIAL sums for every column (zeros right of its last row); the ranks' partial
    // vectors are added by the caller's all-reduce.  (row0 is a multiple of SYMV_H, so pairs never straddle.)
    double2_t s = {0.0, 0.0};
    if (i < n) {
        // loads are independent of the running sums: keep 16 (then 8) of them in flight, add in strip order (the
        // first columns have n / 64 strips to add, 64 per wave: with 8 in flight the kernel took 8 round trips)
        long long I = (i < row0 ? 0 : (i - row0) / SYMV_H) + wave;
        if constexpr (!HANDOFF) {  // (inside k_symv_tail the tile phase's 96 VGPRs bound the kernel: 8 in flight there)
            for (; I + 60 < nstrips; I += 64) {
                double2_t v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = ldp(colpart + (I + 4 * u) * n + i);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    s.x += v[u].x;
                    s.y += v[u].y;
                }
            }
        }
        for (; I + 28 < nstrips; I += 32) {
            double2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ldp(colpart + (I + 4 * u) * n + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += v[u].x;
                s.y += v[u].y;
            }
        }
        for (; I < nstrips; I += 4) {
            const double2_t v = ldp(colpart + I * n + i);
            s.x += v.x;
            s.y += v.y;
        }
    }
    // row partial sums (wave 0): requested eight at a time BEFORE the barrier, added in segment order (one load per
    // loop trip after the barrier was a chain of up to n / seg dependent round trips)
    double2_t r = {0.0, 0.0};
    if (wave == 0 && i < n && i >= row0 && i < row0 + nrows) {
        const long long nJ = i / seg + 1;
        for (long long J = 0; J < nJ; J += 8) {
            double2_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // unconditional loads (clamped), masked afterwards: adding +0.0 changes nothing
                const long long Ju = (J + u < nJ) ? J + u : nJ - 1;
                v[u] = ldp(rowpart + Ju * n + i);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                r.x += (J + u < nJ) ? v[u].x : 0.0;
                r.y += (J + u < nJ) ? v[u].y : 0.0;
            }
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    double2_t yv = {0.0, 0.0};  // this lane's pair of y (wave 0 only)
    if (wave == 0 && i < n) {
        const double2_t c0 = part[0][lane], c1 = part[1][lane], c2 = part[2][lane], c3 = part[3][lane];
        r.x += ((c0.x + c1.x) + c2.x) + c3.x;
        r.y += ((c0.y + c1.y) + c2.y) + c3.y;
        if constexpr (WT) ho_store2(y + i, r);
        else *reinterpret_cast<double2_t*>(y + i) = r;
        if (NP > 0) yv = r;
    }
    if (NP > 0) {
        double* out = partial + blk * (NP + 1);
        if (wave == 0) {
            double sgy = gi.x * yv.x;
            sgy += gi.y * yv.y;
            sgy = wave_allreduce_sum(sgy);
            if (lane == 0) {
                if constexpr (WT) ho_store(out, sgy);
                else out[0] = sgy;
            }
        }
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int j = wave + 4 * k;
            double sv = pv[k].x * gi.x;
            sv += pv[k].y * gi.y;
            sv = wave_allreduce_sum(sv);
            if (lane == 0 && j < NP) {
                if constexpr (WT) ho_store(out + 1 + j, sv);
                else out[1 + j] = sv;
            }
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256) void k_symv_reduce(long long n, long long row0, long long nrows, long long seg,
                                                     const double* __restrict__ rowpart,
                                                     const double* __restrict__ colpart,
                                                     double* __restrict__ y, DevState* __restrict__ st,
                                                     const double* __restrict__ g, const double* __restrict__ pend,
                                                     double* __restrict__ partial) {
    __shared__ double2_t part[4][64];
    const int halted = st->halted;
    if (NP > 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        st->halted_in = halted;
        st->kappa_in = st->kappa;
    }
    if (halted) return;
    symv_reduce_block<NP, false>((long long)blockIdx.x, n, row0, nrows, seg, rowpart, colpart, y, g, pend, partial, part);
}

// ----------------------------------------------------------------------------- k_sweep_apply ---
// Deferred mode: apply the MAXPEND pending rank-1 updates to the local rows in one pass (and, when GV,
// accumulate the GEMV of the next gradient on the freshly written values).  pend: MAXPEND vectors of
// length n (stride n); cpend: their coefficients.  Same mapping and summation shape as k_sweep.
// LOWER: only columns up to the tile's last row are touched (8 n^2 bytes for the whole pass instead of
// 16 n^2).  Legal while every GEMV in between reads the lower triangle only (k_symv): the strict upper
// triangle is then stale and is rebuilt from the lower one (k_mirror_lower_now) before anything reads it
// -- which is exactly what the reference's mirror store `Q[j][i] = Q[i][j]` (src/ell.rs:124-126) makes of it.
template <int RW, int UNR, int VEC, bool NT, bool GV, bool LOWER = false, int NP = 8>
__global__ __launch_bounds__(256) void k_sweep_apply(const double* Q, double* Qout, long long ld, long long n,
                                                     long long nrows, long long row0,
                                                     const double* __restrict__ pend,
                                                     const double* __restrict__ cpend,
                                                     const double* __restrict__ gvec,
                                                     double* __restrict__ gv_out,
                                                     const DevState* __restrict__ st, int reverse) {
    __shared__ double red[4][RW];
    // (no `halted` test here or in k_apply_lower / k_pend_reset: updates recorded BEFORE a queue halted belong to
    // successful cuts and must reach Q whenever something observes it; applying them early is always legal)
    (void)st;
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long tile = reverse ? (long long)gridDim.x - 1 - blockIdx.x : (long long)blockIdx.x;
    const long long row_base = tile * RW;
    if (row_base >= nrows) return;

    double ratio[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) ratio[j] = cpend[j];
    const double* rp[RW];
    double* wp[RW];
    long long grow[RW];
    double gtr[NP][RW], rgr[NP][RW], acc[RW];
    bool valid[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        long long rr = row_base + r;
        valid[r] = rr < nrows;
        if (!valid[r]) rr = nrows - 1;
        rp[r] = Q + rr * ld;
        wp[r] = Qout + rr * ld;
        grow[r] = row0 + rr;
        acc[r] = 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            gtr[j][r] = pend[(long long)j * n + grow[r]];
            rgr[j][r] = ratio[j] * gtr[j][r];  // r_qg of src/ell.rs:119, update j
        }
    }
    const long long grow_min = row0 + row_base;           // rows of this tile: grow_min .. grow_max
    const long long grow_max = grow[RW - 1];
    constexpr long long STEP = 256 * VEC;
    long long cend = n;
    if (LOWER) {
        cend = (grow_max / VEC + 1) * VEC;  // first column past the tile's diagonal, rounded up to the access width
        if (cend > n) cend = n;
    }
    for (long long c = (long long)threadIdx.x * VEC; c < cend; c += STEP * UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long cc = c + u * STEP;
            if (cc >= cend) break;
            V vj[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) vj[j] = *reinterpret_cast<const V*>(pend + (long long)j * n + cc);
            V hv;
            if (GV) hv = *reinterpret_cast<const V*>(gvec + cc);
            V qv[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) qv[r] = ld_stream<NT, V>(rp[r] + cc);
            // Which triangle the thread's VEC columns lie in, for ALL rows of the tile: away from the
            // diagonal (almost everywhere) the per-element select and the unused product disappear.
            const bool all_lower = cc + VEC - 1 <= grow_min;  // col <= row for every (row, col) here
            const bool all_upper = cc > grow_max;             // col >  row for every (row, col) here
            V o[RW];
            if (all_lower) {
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        double x = VecT<VEC>::get(qv[r], v);
#pragma unroll
                        for (int j = 0; j < NP; ++j) x = x - rgr[j][r] * VecT<VEC>::get(vj[j], v);
                        VecT<VEC>::set(o[r], v, x);
                    }
            } else if (all_upper) {
                V rv[NP];  // (ratio_j * gt_j[col]): the mirrored element's r_qg
#pragma unroll
                for (int j = 0; j < NP; ++j)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) VecT<VEC>::set(rv[j], v, ratio[j] * VecT<VEC>::get(vj[j], v));
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        double x = VecT<VEC>::get(qv[r], v);
#pragma unroll
                        for (int j = 0; j < NP; ++j) x = x - VecT<VEC>::get(rv[j], v) * gtr[j][r];
                        VecT<VEC>::set(o[r], v, x);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const bool lower = cc + v <= grow[r];
                        double x = VecT<VEC>::get(qv[r], v);
#pragma unroll
                        for (int j = 0; j < NP; ++j) {  // in recording order: the reference's roundings
                            const double gc = VecT<VEC>::get(vj[j], v);
                            const double upd = lower ? rgr[j][r] * gc : (ratio[j] * gc) * gtr[j][r];
                            x = x - upd;
                        }
                        VecT<VEC>::set(o[r], v, x);
                    }
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (valid[r]) st_stream<false, V>(wp[r] + cc, o[r]);
                if (GV) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[r] += VecT<VEC>::get(o[r], v) * VecT<VEC>::get(hv, v);
                }
            }
        }
    }
    if (GV) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double s = wave_allreduce_sum(acc[r]);
            if (lane == 0) red[wave][r] = s;
        }
        __syncthreads();
        if (threadIdx.x < RW && row_base + threadIdx.x < nrows) {
            const int r = threadIdx.x;
            gv_out[row_base + r] = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
        }
    }
}

// ----------------------------------------------------------------------------- k_apply_lower ---
// Lower-trapezoid apply pass for depth NP (8 or 16), used while every GEMV of the handle is a k_symv.  A
// workgroup owns APL_TR consecutive rows and sweeps the columns up to the tile's diagonal in chunks of 512; per
// chunk each thread loads its 16-byte pair of the NP pending vectors ONCE (registers) and runs all APL_TR rows
// through it, so the vectors cost one L2 read per 16 rows (k_sweep_apply: one per 4) -- that is what makes
// depth 16 pay: twice the updates per pass of Q for the same L2 traffic as depth 8 had.  The row coefficients
// c_j * v_j[row] are workgroup-uniform and live in LDS.  Every element left of or on the diagonal goes through
// the reference's sequence of roundings, x <- x - (c_j v_j[row]) v_j[col], j in recording order; the few
// elements right of the diagonal inside the last chunk get the same expression (they are stale by contract:
// nothing reads the strict upper triangle before k_mirror_lower_now rebuilds it).
// History: the first version (row groups fully unrolled, row-outer / j-inner) needed 255 VGPRs + 68 AGPRs
// (occupancy 1) and took 0.62 ms at depth 8 and 1.36 ms at depth 16 -- slower than k_sweep_apply<LOWER>
// (0.44 ms); with the row-group loop rolled and j outermost it takes 0.40 ms at depth 8 and 0.41 ms at depth 16,
// which makes depth 16 the fastest schedule (4120 vs 3810 updates/s at n = 16384).
constexpr int APL_TR = 16;

// APL_RG = rows in flight per thread (measured at n = 16384, ms per pass: depth 8: RG 4 0.426, 8 0.404, 16 0.397;
// depth 16: RG 4 0.409, 8 0.471, 16 1.57 (spills)).
template <int NP, bool NT, int APL_RG = (NP >= 16 ? 4 : 8)>
__global__ __launch_bounds__(256) void k_apply_lower(double* __restrict__ Q, long long ld, long long n,
                                                     long long nrows, long long row0,
                                                     const double* __restrict__ pend,
                                                     const double* __restrict__ cpend,
                                                     const DevState* __restrict__ st) {
    __shared__ double coef[NP][APL_TR];
    (void)st;
    const long long tile = (long long)gridDim.x - 1 - blockIdx.x;  // last (longest) rows first
    const long long lr0 = tile * APL_TR;                            // first local row of the tile
    if (lr0 >= nrows) return;
    const int nr = (int)((nrows - lr0 < APL_TR) ? nrows - lr0 : APL_TR);
    for (int idx = threadIdx.x; idx < NP * APL_TR; idx += 256) {
        const int j = idx / APL_TR, r = idx - j * APL_TR;
        coef[j][r] = (r < nr) ? cpend[j] * pend[(long long)j * n + row0 + lr0 + r] : 0.0;  // r_qg of src/ell.rs:119
    }
    __syncthreads();
    const long long gmax = row0 + lr0 + nr - 1;  // last global row of the tile
    long long cend = (gmax / 2 + 1) * 2;         // first column past the tile's diagonal, pair aligned
    if (cend > n) cend = n;
    double* base = Q + lr0 * ld;
    for (long long c = 2 * (long long)threadIdx.x; c < cend; c += 512) {
        double2_t vj[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) vj[j] = *reinterpret_cast<const double2_t*>(pend + (long long)j * n + c);
#pragma unroll 1  // (fully unrolled, the row groups kept 255 VGPRs + 68 AGPRs alive: occupancy 1)
        for (int r0 = 0; r0 < APL_TR; r0 += APL_RG) {
            double2_t x[APL_RG];
            bool on[APL_RG];
#pragma unroll
            for (int u = 0; u < APL_RG; ++u) {
                const int r = r0 + u;
                // a row takes part while the pair starts at or left of its diagonal (keeps the traffic at the trapezoid)
                on[u] = r < nr && c <= row0 + lr0 + r;
                if (on[u]) x[u] = ld_stream<NT, double2_t>(base + (long long)r * ld + c);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int u = 0; u < APL_RG; ++u) {  // per element still j ascending: the reference's order of roundings
                    const double cf = coef[j][r0 + u];
                    x[u].x = x[u].x - cf * vj[j].x;
                    x[u].y = x[u].y - cf * vj[j].y;
                }
            }
#pragma unroll
            for (int u = 0; u < APL_RG; ++u)
                if (on[u]) *reinterpret_cast<double2_t*>(base + (long long)(r0 + u) * ld + c) = x[u];
        }
    }
}

// ------------------------------------------------------------------------------ k_apply_mfma ---
// The same lower-trapezoid apply pass on the FP64 matrix cores: the NP recorded updates are ONE rank-NP update
//     Q[rows, cols] += A B,   A[r][k] = -(c_k v_k[r])  (16 x NP per wave),   B[k][c] = v_k[c]  (NP x 16 per tile),
// NP / 4 v_mfma_f64_16x16x4_f64 per 16 x 16 tile.  Why: k_apply_lower keeps NP pending vectors (2 NP registers) AND
// NP x rows-in-flight coefficients in registers -- 216 VGPRs at NP = 16 (two workgroups per CU), 256 + spills into the
// accumulation file at NP = 24 (0.62 ms per pass against 0.40) -- so the depth could not grow; here the accumulators are
// 4 doubles per tile and the operands one double each, whatever NP is.  The matrix pipe runs at the vector FMA rate
// (64 cycles per instruction and SIMD): at NP = 24 about a fifth of it is used, the pass stays HBM-bound.
// Mapping: workgroup = 64 rows (wave w: rows 16 w ..) x APM_COLS columns of the trapezoid; the pending vectors' 128-column
// block is staged through LDS (double-buffered) and shared by the four waves; a lane loads 16-byte pairs (columns 2m,
// 2m + 1 of four rows: the C layout of the f64 MFMA is col = lane & 15, row = (lane >> 4) + 4 i), the even columns feed
// one tile, the odd ones a second.
// Numerics: every element receives x + sum_k (-(c_k v_k[r])) v_k[c] as a chain of fused multiply-adds in recording
// order (one rounding per update) where the reference and k_apply_lower round the product and the difference
// separately (two): the results differ in the last bit per update and stay far inside the 1e-10 contract; the
// coefficient c_k v_k[r] itself is rounded exactly as src/ell.rs:119 rounds r_qg.  Elements right of the diagonal inside
// the diagonal 64-column block get the same expression (stale by contract, as in k_apply_lower).
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int APM_ROWS = 64;
constexpr int APM_COLS = 1024;

template <int NP, bool NT>
__global__ __launch_bounds__(256) void k_apply_mfma(double* __restrict__ Q, long long ld, long long n, long long nrows,
                                                    long long row0, const double* __restrict__ pend,
                                                    const double* __restrict__ cpend, const DevState* __restrict__ st) {
    static_assert(NP % 8 == 0, "rank of the update: a multiple of 8 (k = 4 per MFMA, whole pairs per staging thread)");
    constexpr int KS = NP / 4;
    constexpr int SPT = NP / 8;               // 16-byte pairs of the pending vectors a thread stages per 64-column block
    __shared__ double sh_v[2][NP][APM_ROWS];  // the pending vectors on a 64-column block (this one | the next)
    (void)st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long strip = (long long)gridDim.x - 1 - blockIdx.x;  // last (longest) strips first
    const long long lr0 = strip * APM_ROWS;
    if (lr0 >= nrows) return;
    const long long gr0 = row0 + lr0;                                     // first global row of the strip
    const long long glast = (lr0 + APM_ROWS <= nrows ? lr0 + APM_ROWS : nrows) - 1 + row0;  // last global row
    long long cend = (glast / 2 + 1) * 2;                                 // first column past the strip's diagonal (pair aligned)
    if (cend > n) cend = n;
    const long long c_begin = (long long)blockIdx.y * APM_COLS;
    if (c_begin >= cend) return;
    const long long c_stop = (c_begin + APM_COLS < cend) ? c_begin + APM_COLS : cend;
    // A operand: A[row = lane & 15][k = lane >> 4] of k-step s = -(c_k v_k[row]), k = 4 s + (lane >> 4)
    const long long arow = gr0 + 16 * wave + (lane & 15);
    double a[KS];
#pragma unroll
    for (int sidx = 0; sidx < KS; ++sidx) {
        const int k = 4 * sidx + (lane >> 4);
        a[sidx] = (arow <= glast) ? -(cpend[k] * pend[(long long)k * n + arow]) : 0.0;  // r_qg of src/ell.rs:119, negated
    }
    // staging of the pending vectors' block [cb, cb + 64): NP x 32 pairs, SPT per thread -- requested into registers at
    // the top of a step, parked in LDS after the step's MFMAs (their latency hides behind the step)
    const int sk = tid >> 5, sj = (tid & 31) * 2;     // vector sk + 8 t, columns sj, sj + 1 of the block
    auto fetch = [&](double2_t (&v)[SPT], long long cb) {
#pragma unroll
        for (int t = 0; t < SPT; ++t)
            v[t] = (cb + sj < n) ? *reinterpret_cast<const double2_t*>(pend + (long long)(sk + 8 * t) * n + cb + sj) : double2_t{0.0, 0.0};
    };
    auto park = [&](int buf, const double2_t (&v)[SPT]) {
#pragma unroll
        for (int t = 0; t < SPT; ++t) *reinterpret_cast<double2_t*>(&sh_v[buf][sk + 8 * t][sj]) = v[t];
    };
    int buf = 0;
    {
        double2_t v0[SPT];
        fetch(v0, c_begin);
        park(0, v0);
    }
    __syncthreads();
    double* rbase = Q + (lr0 + 16 * wave + (lane >> 4)) * ld + 2 * (lane & 15);  // this lane's pair of row (lane >> 4) of the wave
    // The wave's 16 x 64 block of Q for one step: two tiles (h) x four row groups (i), 16 bytes each.  The block of step
    // s + 1 is REQUESTED before the MFMA chain of step s starts (round 3 loaded, multiplied and stored one tile after the
    // other: the pass ran at the SUM of its HBM time and its matrix-pipe time -- 0.60 ms at rank 48, n = 16384, where the
    // two are 0.39 and 0.16 ms; with the loads in flight behind 2 x 2 KS MFMAs they overlap).
    auto loadq = [&](double2_t (&x)[2][4], long long cb) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long c = cb + 32 * h + 2 * (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long grow = gr0 + 16 * wave + (lane >> 4) + 4 * i;
                const bool on = grow <= glast && c < c_stop;
                x[h][i] = on ? ld_stream<NT, double2_t>(rbase + (long long)(4 * i) * ld + cb + 32 * h) : double2_t{0.0, 0.0};
            }
        }
    };
    auto step = [&](double2_t (&x)[2][4], double2_t (&xn)[2][4], long long cb) __attribute__((always_inline)) {
        const bool more = cb + APM_ROWS < c_stop;
        double2_t vn[SPT];
        if (more) {
            loadq(xn, cb + APM_ROWS);
            fetch(vn, cb + APM_ROWS);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long c = cb + 32 * h + 2 * (lane & 15);
            double4_t ce = {x[h][0].x, x[h][1].x, x[h][2].x, x[h][3].x}, co = {x[h][0].y, x[h][1].y, x[h][2].y, x[h][3].y};
#pragma unroll
            for (int sidx = 0; sidx < KS; ++sidx) {
                const double2_t b = *reinterpret_cast<const double2_t*>(&sh_v[buf][4 * sidx + (lane >> 4)][32 * h + 2 * (lane & 15)]);
                ce = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sidx], b.x, ce, 0, 0, 0);
                co = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sidx], b.y, co, 0, 0, 0);
            }
            const double oe[4] = {ce.x, ce.y, ce.z, ce.w}, oo[4] = {co.x, co.y, co.z, co.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long grow = gr0 + 16 * wave + (lane >> 4) + 4 * i;
                if (grow <= glast && c < c_stop)
                    *reinterpret_cast<double2_t*>(rbase + (long long)(4 * i) * ld + cb + 32 * h) = double2_t{oe[i], oo[i]};
            }
        }
        if (more) park(buf ^ 1, vn);
        __syncthreads();  // everybody is done with sh_v[buf]; sh_v[buf ^ 1] is complete
        buf ^= 1;
    };
    double2_t xa[2][4], xb[2][4];
    loadq(xa, c_begin);
    for (long long cb = c_begin; cb < c_stop; cb += 2 * APM_ROWS) {
        step(xa, xb, cb);
        if (cb + APM_ROWS < c_stop) step(xb, xa, cb + APM_ROWS);
    }
}

// ------------------------------------------------------------------------------- k_symm_mfma ---
// The lower-triangle product for up to 16 queued gradients at once on the FP64 matrix cores: Y = Q_base G, G = n x 16
// (k_pack_grads lays the group's gradients out as gT[c][v], zero-padded to 16 columns).  k_symv_multi runs out of
// vector ALU and registers near four gradients (every element costs 4 LV operations and LV live column sums); the
// matrix pipe does a 16 x 16 x 4 block product in 16 cycles, so sixteen gradients ride on the time one pass over the
// triangle takes from HBM anyway: (4 / 16) n^2 bytes per update.
// Tile = 64 rows x SYMV_SEG columns (k_symv's grid); wave w takes the 16-column blocks w, w + 4, ... over all 64 rows.
// Per block the wave loads 16 registers N[j] = Q[r0 + 4 j + (lane >> 4)][cb + (lane & 15)] (4 rows x 128 contiguous
// bytes per instruction), which IS the B operand of the column product (contraction over rows):
//     Dc[v][c] += sum_r gT[r][v] Q[r][c]            A = gT rows (lane & 15 = v, lane >> 4 = r), 16 MFMAs
// and, transposed through the wave's private LDS patch (pitch 17: the reads of 16 rows at one column hit 16 banks),
// the B operand of the row product (contraction over columns):
//     Dr[j'][v][r] += sum_c gT[c][v] Q[r][c]        A = gT columns, 16 MFMAs
// Dc leaves as colpart_v[I][c] per block, Dr is added over the four waves as ((w0 + w1) + w2) + w3 and leaves as
// rowpart_v[J][r]: the partial sums k_symv_reduce expects (for a row shard: of its trapezoid).  Numerics: the MFMA sums four products per instruction in
// its own association and fuses the multiply-add; y differs from k_symv's by a few ulp (inside the 1e-10 contract, not
// bit-identical to the vector-ALU kernels).  Requires n % 64 == 0 (and shard boundaries at multiples of 64).
constexpr int SMM_NV = 16;
constexpr int SMM_PITCH = 17;

// (k_pack_grads also rewinds the tile queue of the pass that follows it on the same stream: k_symm_mfma_q.  nvw = 16 or 32: the
// gradients side by side, gT[c][v], zero-padded to the width the pass multiplies)
__global__ __launch_bounds__(256) void k_pack_grads(const double* __restrict__ g, long long g_stride, int lv, long long n,
                                                    double* __restrict__ gT, unsigned* __restrict__ queue = nullptr,
                                                    int nvw = SMM_NV) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // element of gT: c = i / nvw, v = i % nvw
    if (i == 0 && queue) *queue = 0u;
    if (i >= n * nvw) return;
    const long long c = i / nvw;
    const int v = (int)(i % nvw);
    gT[i] = v < lv ? g[(long long)v * g_stride + c] : 0.0;
}

// Round 4 tried the cross-block register pipeline here (loads of block b + 4, and b + 8, in flight behind the 32 MFMAs of block
// b, wave-uniform row addresses through the scalar unit): at two waves per SIMD the 256 registers do not hold a second 8 KiB
// block beside gr / dr / the LDS reads in flight -- 336-490 bytes of scratch per lane, 0.65 ms per pass against 0.31
// (profiles/r04/symm_mfma_pipeline_attempt.txt).  The apply pass (k_apply_mfma), whose per-step state is 16 registers, took the
// same change and gained 15 %.  An LDS-DMA ring per wave, and loader / consumer waves, followed (tools/experiments/symm_glds.hip,
// symm_lc_kernel.hpp, profiles/r04/symm_mfma_overlap_probes.txt): same bits, no gain -- on a SIMD that executes f64 MFMAs every
// vector-memory instruction costs the matrix pipe ~70 cycles whichever wave issues it, so the two phases add up however the
// loads are buffered.  What did pay is how the tiles reach the CUs: k_symm_mfma_q below.
//
// One tile: strip I of the shard (rows r0 = row0 + 64 I ...), column segment J; sh = the workgroup's 4 x 64 x 17 doubles.
template <bool NT, int SEG>
__device__ __forceinline__ void symm_tile(const double* __restrict__ Q, long long ld, long long n, long long row0, long long I,
                                          long long J, const double* __restrict__ gT, int lv, double* __restrict__ rowpart,
                                          double* __restrict__ colpart, long long rowpart_stride, long long colpart_stride,
                                          double (*sh)[SYMV_H * SMM_PITCH]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    const long long r0 = row0 + I * SYMV_H;
    const long long c0 = J * SEG;
    const bool full = c0 + SEG - 1 < r0;
    // A operand of the column product: gT rows of the strip
    double gr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) gr[j] = gT[(r0 + 4 * j + lr) * SMM_NV + lc];
    double4_t dr[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dr[jj] = double4_t{0.0, 0.0, 0.0, 0.0};
    // blocks at or left of the strip's diagonal
    const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;  // first column past the tile's part
    const int nblk = (int)((cend - c0) / 16);
    double* mysh = sh[wave];
    const double* qbase = Q + (r0 + lr) * ld + lc;
    for (int b = wave; b < nblk; b += 4) {
        const long long cb = c0 + 16 * (long long)b;
        double x[16], gc[4];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = ld_stream<NT, double>(qbase + (long long)(4 * j) * ld + cb);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) gc[kb] = gT[(cb + 4 * kb + lr) * SMM_NV + lc];
        const bool diag = !full && cb + 15 >= r0;  // some element of the block is on or right of the diagonal
        double4_t dc = {0.0, 0.0, 0.0, 0.0};
        if (diag) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long r = r0 + 4 * j + lr, c = cb + lc;
                const double below = (c < r) ? x[j] : 0.0;   // column sums: strictly below the diagonal
                dc = __builtin_amdgcn_mfma_f64_16x16x4f64(gr[j], below, dc, 0, 0, 0);
                x[j] = (c <= r) ? x[j] : 0.0;                // row sums: the diagonal counts once, here
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dc = __builtin_amdgcn_mfma_f64_16x16x4f64(gr[j], x[j], dc, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) mysh[(4 * j + lr) * SMM_PITCH + lc] = x[j];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const double t = mysh[(16 * jj + lc) * SMM_PITCH + 4 * kb + lr];
                dr[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[kb], t, dr[jj], 0, 0, 0);
            }
        const double o[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = lr + 4 * i;
            if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[i];
        }
    }
    // row sums of the four waves, in wave order
    __syncthreads();
    double* red = &sh[0][0];  // [wave][jj][i][lane]: 4 * 16 * 64 doubles = 32 KiB of the 34 KiB
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const double o[4] = {dr[jj].x, dr[jj].y, dr[jj].z, dr[jj].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((wave * 4 + jj) * 4 + i) * 64 + lane] = o[i];
    }
    __syncthreads();
    // thread (wave, lane) takes row group jj = wave: rows 16 wave + lc, vectors lr + 4 i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = lr + 4 * i;
        const int jj = wave;
        const double s0 = red[((0 * 4 + jj) * 4 + i) * 64 + lane], s1 = red[((1 * 4 + jj) * 4 + i) * 64 + lane];
        const double s2 = red[((2 * 4 + jj) * 4 + i) * 64 + lane], s3 = red[((3 * 4 + jj) * 4 + i) * 64 + lane];
        if (v < lv) rowpart[(long long)v * rowpart_stride + J * n + r0 + 16 * jj + lc] = ((s0 + s1) + s2) + s3;
    }
}

// One workgroup per tile of k_symv's grid (the form round 3 shipped; the experiments and the queue form's check use it).
template <bool NT, int SEG>
__global__ __launch_bounds__(256) void k_symm_mfma(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                   long long nrows, const double* __restrict__ gT, int lv,
                                                   double* __restrict__ rowpart,
                                                   double* __restrict__ colpart, long long rowpart_stride,
                                                   long long colpart_stride, const DevState* __restrict__ st) {
    __shared__ double sh[4][SYMV_H * SMM_PITCH];
    if (st->halted) return;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = (long long)blockIdx.y;
    // a symmetric row shard holds the rows [row0, row0 + nrows) (both multiples of 64): I counts its local strips, row and
    // column indices are global, the result is this shard's PARTIAL sums (as k_symv's for a shard)
    const long long r0 = row0 + I * SYMV_H;
    if (r0 >= row0 + nrows || J * SEG > r0 + SYMV_H - 1) return;
    symm_tile<NT, SEG>(Q - row0 * ld, ld, n, row0, I, J, gT, lv, rowpart, colpart, rowpart_stride, colpart_stride, sh);
}

// The same tiles handed out from a QUEUE.  A 64 x 2048 tile keeps its workgroup ~100 us and a card holds 512 such workgroups,
// so the grid form's pass lasts whole workgroup lifetimes: 1152 tiles (n = 16384; 1028 full ones' worth of work) take three.
// Here 3 workgroups per CU are launched once, each draws tile after tile from a counter -- largest first, the host-built table
// `tiles` -- until none is left, and the pass ends with the smallest tiles: 0.272-0.275 ms against 0.306-0.317 (interleaved
// rounds in one process, tools/experiments/symm_glds.hip).  Same tiles, same arithmetic per tile: the partial sums are
// bit-identical to the grid form's.  The counter is rewound by the k_pack_grads launch ahead of the pass.
struct SymmTile {
    int I, J;  // strip of the shard, column segment
};
template <bool NT, int SEG>
__global__ __launch_bounds__(256) void k_symm_mfma_q(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                     const double* __restrict__ gT, int lv, double* __restrict__ rowpart,
                                                     double* __restrict__ colpart, long long rowpart_stride,
                                                     long long colpart_stride, const DevState* __restrict__ st,
                                                     const SymmTile* __restrict__ tiles, int ntiles,
                                                     unsigned* __restrict__ queue) {
    __shared__ double sh[4][SYMV_H * SMM_PITCH];
    __shared__ int s_t;
    if (st->halted) return;
    Q -= row0 * ld;
    for (;;) {
        __syncthreads();  // (everybody has read the previous index and is done with the tile's LDS)
        if (threadIdx.x == 0) {
            const unsigned t = atomicAdd(queue, 1u);
            s_t = t < (unsigned)ntiles ? (int)t : -1;
        }
        __syncthreads();
        const int t = s_t;
        if (t < 0) return;  // (uniform)
        symm_tile<NT, SEG>(Q, ld, n, row0, (long long)tiles[t].I, (long long)tiles[t].J, gT, lv, rowpart, colpart, rowpart_stride,
                           colpart_stride, sh);
    }
}

// Up to 32 gradients per pass: two 16-wide column tiles of the MFMA over the same block of Q (gT[c][32]).  The A operands of the
// column product (the gT rows of the strip, 64 x 32) live in LDS, shared by the four waves -- 64 registers per lane would not fit
// beside the block and the eight row-sum accumulators at two waves per SIMD; every accumulator takes its MFMAs back to back (the
// pipe forwards the accumulator it has just written: 68 cycles per MFMA against 74-83 in rotation).  Per vector the arithmetic is
// k_symm_mfma's: the partial sums of 32 gradients are bit-identical to those of two 16-wide passes, in 0.445 ms against 0.55-0.62
// (n = 16384, tools/experiments/symm32_queue.hip).
constexpr int SMM_NV2 = 2 * SMM_NV;
template <bool NT, int SEG>
__device__ __forceinline__ void symm_tile2(const double* __restrict__ Q, long long ld, long long n, long long row0, long long I,
                                           long long J, const double* __restrict__ gT, int lv, double* __restrict__ rowpart,
                                           double* __restrict__ colpart, long long rowpart_stride, long long colpart_stride,
                                           double (*sh)[SYMV_H * SMM_PITCH], double (*sgr)[SMM_NV2 + 1]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    const long long r0 = row0 + I * SYMV_H;
    const long long c0 = J * SEG;
    const bool full = c0 + SEG - 1 < r0;
    for (int k = threadIdx.x; k < SYMV_H * SMM_NV2; k += 256) sgr[k / SMM_NV2][k % SMM_NV2] = gT[(r0 + k / SMM_NV2) * SMM_NV2 + k % SMM_NV2];
    __syncthreads();
    double4_t dr[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dr[t][jj] = double4_t{0.0, 0.0, 0.0, 0.0};
    const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;
    const int nblk = (int)((cend - c0) / 16);
    double* mysh = sh[wave];
    const double* qbase = Q + (r0 + lr) * ld + lc;
    for (int b = wave; b < nblk; b += 4) {
        const long long cb = c0 + 16 * (long long)b;
        double x[16], gc[2][4];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = ld_stream<NT, double>(qbase + (long long)(4 * j) * ld + cb);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[t][kb] = gT[(cb + 4 * kb + lr) * SMM_NV2 + 16 * t + lc];
        const bool diag = !full && cb + 15 >= r0;
        double4_t dc[2] = {double4_t{0.0, 0.0, 0.0, 0.0}, double4_t{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long r = r0 + 4 * j + lr, c = cb + lc;
                const double below = (!diag || c < r) ? x[j] : 0.0;  // column sums: strictly below the diagonal
                dc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(sgr[4 * j + lr][16 * t + lc], below, dc[t], 0, 0, 0);
            }
        if (diag) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long r = r0 + 4 * j + lr, c = cb + lc;
                x[j] = (c <= r) ? x[j] : 0.0;  // row sums: the diagonal counts once, here
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) mysh[(4 * j + lr) * SMM_PITCH + lc] = x[j];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const double tv = mysh[(16 * jj + lc) * SMM_PITCH + 4 * kb + lr];
                    dr[t][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[t][kb], tv, dr[t][jj], 0, 0, 0);
                }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double o[4] = {dc[t].x, dc[t].y, dc[t].z, dc[t].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 16 * t + lr + 4 * i;
                if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[i];
            }
        }
    }
    __syncthreads();
    double* red = &sh[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const double o[4] = {dr[t][jj].x, dr[t][jj].y, dr[t][jj].z, dr[t][jj].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave * 4 + jj) * 4 + i) * 64 + lane] = o[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = 16 * t + lr + 4 * i;
            const int jj = wave;
            const double s0 = red[((0 * 4 + jj) * 4 + i) * 64 + lane], s1 = red[((1 * 4 + jj) * 4 + i) * 64 + lane];
            const double s2 = red[((2 * 4 + jj) * 4 + i) * 64 + lane], s3 = red[((3 * 4 + jj) * 4 + i) * 64 + lane];
            if (v < lv) rowpart[(long long)v * rowpart_stride + J * n + r0 + 16 * jj + lc] = ((s0 + s1) + s2) + s3;
        }
        __syncthreads();
    }
}

template <bool NT, int SEG>
__global__ __launch_bounds__(256, 2) void k_symm_mfma_q2(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                         const double* __restrict__ gT, int lv, double* __restrict__ rowpart,
                                                         double* __restrict__ colpart, long long rowpart_stride,
                                                         long long colpart_stride, const DevState* __restrict__ st,
                                                         const SymmTile* __restrict__ tiles, int ntiles,
                                                         unsigned* __restrict__ queue) {
    __shared__ double sh[4][SYMV_H * SMM_PITCH];
    __shared__ double sgr[SYMV_H][SMM_NV2 + 1];  // the gT rows of the strip (odd pitch)
    __shared__ int s_t;
    if (st->halted) return;
    Q -= row0 * ld;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = atomicAdd(queue, 1u);
            s_t = t < (unsigned)ntiles ? (int)t : -1;
        }
        __syncthreads();
        const int t = s_t;
        if (t < 0) return;
        symm_tile2<NT, SEG>(Q, ld, n, row0, (long long)tiles[t].I, (long long)tiles[t].J, gT, lv, rowpart, colpart, rowpart_stride,
                            colpart_stride, sh, sgr);
    }
}

// ----------------------------------------------------------------------------------- k_publish ---
// The live SearchSpace loop (src/cutting_plane.rs:299-311: xc() -> oracle -> update_*_cut) needs, after every update, the
// cut's status / tsq (the caller's next branch, :308) and the new centre (the oracle's next argument, :300) -- and nothing
// else: the shrink of Q may still be running when the call returns.  One workgroup, right behind the scalar stage, writes
// the 128 KiB centre and the scalar state straight into pinned host memory (fine-grained) and then the update's sequence
// number with a system-scope release; the host polls that word instead of issuing a device-to-host copy and waiting for
// the whole stream (ellhip_update_end / ellhip_cut: live_publish, live_wait).
// (LiveMirror: defined beside DevState)
constexpr int PUB_WGS = 16;  // one workgroup pushes ~12 GB/s over PCIe (11.4 us for the 128 KiB centre at n = 16384); 16: ~3 us
__global__ __launch_bounds__(256) void k_publish(const DevState* __restrict__ st, const double* __restrict__ xc, long long n,
                                                 LiveMirror* __restrict__ m, double* __restrict__ h_xc,
                                                 unsigned long long seq, unsigned* __restrict__ arrived) {
    __shared__ int last;
    const long long n2 = n / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256)
        *reinterpret_cast<double2_t*>(h_xc + 2 * i) = *reinterpret_cast<const double2_t*>(xc + 2 * i);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) h_xc[n - 1] = xc[n - 1];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (old + 1 == gridDim.x);
        if (last) __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!last) return;
    // the last workgroup to arrive: every slice of the centre is in host memory; the scalar state, then the sequence number
    if (threadIdx.x < (int)(sizeof(DevState) / sizeof(long long)))
        reinterpret_cast<long long*>(&m->st)[threadIdx.x] = reinterpret_cast<const long long*>(st)[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Host gradient in: the caller's n doubles sit in a pinned staging buffer; the chip pulls them over PCIe itself.  A copy
// engine transfer (hipMemcpyAsync) costs 7.8 us + 8 us of cross-engine hand-over before the GEMV can start; this launch
// sits in the same queue as the GEMV behind it (profiles/r04/live_loop_timeline_*.txt).
constexpr int STAGE_WGS = 32;
__global__ __launch_bounds__(256) void k_stage(const double* __restrict__ h_src, double* __restrict__ dst, long long n) {
    const long long n2 = n / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256)
        *reinterpret_cast<double2_t*>(dst + 2 * i) = *reinterpret_cast<const double2_t*>(h_src + 2 * i);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = h_src[n - 1];
}

// After a flush: forget the pending updates (unused slots must read as exact zeros).
__global__ __launch_bounds__(256) void k_pend_reset(double* __restrict__ pend, double* __restrict__ cpend,
                                                    long long total, DevState* __restrict__ st) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        pend[i] = 0.0;
    if (blockIdx.x == 0 && threadIdx.x < MAXPEND) cpend[threadIdx.x] = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->npend = 0;
}

// ---------------------------------------------------------------------------------- k_scalar ---
// The scalar stage between the two passes, spread over G workgroups (one CU moves only ~70 GB/s, and
// the stage touches ~56 n bytes) in two launches:
//   k_scalar_dot    partial[b] = sum over slice b of g[i]*gt[i]          (src/arr.rs:443-451)
//   k_scalar_apply  every workgroup: omega = sum_b partial[b]; tsq = kappa*omega; EllCalc; then its slice
//                   of xc -= (rho/omega) gt; workgroup 0 publishes kappa / status / ratio (src/ell.rs:105-135)
// The reduction shape is fixed by n alone (slice length, 256 sequential-per-thread lanes, xor
// butterfly, ((w0+w1)+w2)+w3, then the partials in index order), so omega has the same bits on every
// rank of a row-partitioned run and in every schedule.
#ifndef ELLHIP_SCALAR_SLICE
#define ELLHIP_SCALAR_SLICE 1024
#endif
__host__ __device__ inline int scalar_groups(long long n) {
    long long g = n / ELLHIP_SCALAR_SLICE;  // elements per workgroup of the scalar stage (tuning builds: -DELLHIP_SCALAR_SLICE=)
    if (g < 1) g = 1;
    if (g > 64) g = 64;
    return (int)g;
}
__host__ __device__ inline long long scalar_slice(long long n) {
    const long long g = scalar_groups(n);
    long long m = (n + g - 1) / g;
    return (m + 1) & ~1LL;  // even, so a slice never splits a 16-byte pair
}

__global__ __launch_bounds__(256) void k_scalar_dot(long long n, const double* __restrict__ g,
                                                    const double* __restrict__ gt,
                                                    double* __restrict__ partial, DevState* __restrict__ st) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int halted = st->halted;
    if (blockIdx.x == 0 && tid == 0) {
        // snapshots for k_scalar_apply, whose lead workgroup rewrites kappa / halted while the others read
        st->halted_in = halted;
        st->kappa_in = st->kappa;
    }
    if (halted) return;
    const long long m = scalar_slice(n);
    const long long lo = (long long)blockIdx.x * m;
    const long long hi = (lo + m < n) ? lo + m : n;
    double s = 0.0;
    for (long long i = lo + tid; i < hi; i += 256) s += g[i] * gt[i];
    s = wave_allreduce_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void k_scalar_apply(long long n, const double* __restrict__ gt,
                                                      double* __restrict__ xc,
                                                      const double* __restrict__ partial,
                                                      DevState* __restrict__ st, EllCalcDev calc,
                                                      const CutParams* __restrict__ cp_dev, CutParams cp_val,
                                                      int no_defer_trick, int queue_mode,
                                                      int* __restrict__ q_status, double* __restrict__ q_tsq) {
    __shared__ double bc_roo;
    __shared__ int bc_status;
    const int tid = threadIdx.x;
    const bool lead = blockIdx.x == 0;
    if (st->halted_in) {  // snapshot taken before this launch (see below): uniform across workgroups
        if (lead && tid == 0 && q_status) {
            *q_status = ST_UNKNOWN;
            *q_tsq = st->tsq;
        }
        return;
    }
    if (tid == 0) {
        const int G = scalar_groups(n);
        double omega = 0.0;
        for (int b = 0; b < G; ++b) omega += partial[b];
        const double kappa = st->kappa_in;
        const double tsq = kappa * omega;  // src/ell.rs:105
        Coef cf;
        const CutParams cp = cp_dev ? *cp_dev : cp_val;  // queued cuts live in HBM, direct ones arrive by value
        const int status = calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);  // :106
        double roo = 0.0;
        if (status == ST_SUCCESS) roo = cf.rho / omega;  // :112
        if (lead) {
            st->tsq = tsq;
            st->omega = omega;
            st->status = status;
            if (status == ST_SUCCESS) {
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
            }
            queue_bookkeeping(st, status, tsq, queue_mode);
            if (q_status) {
                *q_status = status;
                *q_tsq = tsq;
            }
        }
        bc_roo = roo;
        bc_status = status;
    }
    __syncthreads();
    if (bc_status != ST_SUCCESS) return;
    const double roo = bc_roo;
    const long long m = scalar_slice(n);
    const long long lo = (long long)blockIdx.x * m;
    const long long hi = (lo + m < n) ? lo + m : n;
    for (long long i = lo + tid; i < hi; i += 256) xc[i] = xc[i] - roo * gt[i];  // :113-115
}

constexpr int SC_LP_ROWS = 128;  // rows of partial sums staged through LDS per pass (a multiple of 8)

// Deferred-mode scalar stage (see MAXPEND above).  y = Q_base*g comes from the GEMV pass.
//   k_scalar_dot_def    partial[b][0] = slice of g.y ; partial[b][1+j] = slice of v_j.g
//   k_scalar_apply_def  every workgroup: gy, d_j = v_j.g; omega = gy - sum_j c_j d_j^2; tsq, EllCalc; its slice
//                       of gt = y - sum_j (c_j d_j) v_j is recorded as the new pending vector and applied
//                       to xc; workgroup 0 records c = sigma/omega, bumps npend, publishes kappa / status.
template <int NP>
__global__ __launch_bounds__(256) void k_scalar_dot_def(long long n, const double* __restrict__ g,
                                                        const double* __restrict__ y,
                                                        const double* __restrict__ pend,
                                                        double* __restrict__ partial, DevState* __restrict__ st) {
    __shared__ double red[4][NP + 1];
    const int tid = threadIdx.x;
    const int halted = st->halted;
    if (blockIdx.x == 0 && tid == 0) {
        st->halted_in = halted;
        st->kappa_in = st->kappa;
    }
    if (halted) return;
    const long long m = scalar_slice(n);
    const long long lo = (long long)blockIdx.x * m;
    const long long hi = (lo + m < n) ? lo + m : n;
    double s[NP + 1];
#pragma unroll
    for (int k = 0; k <= NP; ++k) s[k] = 0.0;
    for (long long i = lo + tid; i < hi; i += 256) {
        const double gi = g[i];
        s[0] += gi * y[i];
#pragma unroll
        for (int j = 0; j < NP; ++j) s[1 + j] += pend[(long long)j * n + i] * gi;
    }
#pragma unroll
    for (int k = 0; k <= NP; ++k) {
        const double w = wave_allreduce_sum(s[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = w;
    }
    __syncthreads();
    if (tid <= NP)
        partial[(long long)blockIdx.x * (NP + 1) + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// GY: the partial sums' column 0 (g . y) does not exist yet (k_sweep_gemv_dots produced the other columns beside the
// GEMV that produced y): every workgroup forms the scalar_groups(n) slice sums itself, in k_scalar_dot_def's shape, and
// they enter the reduction below exactly where the stored column would -- the same bits either way.
// (the body of the kernel, for workgroup `wg` of the stage: also run by the last workgroups of k_update_fused_def)
template <int NP, bool GY>
__device__ __forceinline__ void scalar_apply_def_body(const long long wg, const long long lo_sl, const long long hi_sl,
                                                      long long n, const double* __restrict__ y,
                                                      double* __restrict__ xc, double* __restrict__ pend,
                                                      double* __restrict__ cpend,
                                                      const double* __restrict__ partial,
                                                      DevState* __restrict__ st, EllCalcDev calc,
                                                      const CutParams* __restrict__ cp_dev, CutParams cp_val,
                                                      int slot, int queue_mode, int* __restrict__ q_status,
                                                      double* __restrict__ q_tsq, int npart,
                                                      const double* __restrict__ g_own) {
    // npart = number of partial-sum rows: scalar_groups(n) after k_scalar_dot_def, ceil(n / 128) after k_symv_reduce<NP>
    // `slot` = number of updates already pending = index of the (all-zero) slot this cut records into.
    // The host passes it: it equals the device's count as long as the queue has not halted, and a halted
    // queue ignores every later launch.
    __shared__ double bc_roo;
    __shared__ double bc_cd[NP];
    __shared__ int bc_status;
    const int tid = threadIdx.x;
    const bool lead = wg == 0;
    if (st->halted_in) {
        if (lead && tid == 0 && q_status) {
            *q_status = ST_UNKNOWN;
            *q_tsq = st->tsq;
        }
        return;
    }
    __shared__ double dsum[NP + 1];
    // Sum the npart rows of partial sums, per column: 8 interleaved running sums (thread group q adds rows q, q + 8,
    // ... in ascending order), combined in a fixed order.  The rows come through LDS, SC_LP_ROWS at a time, fetched by
    // all 256 threads at once: one memory round trip per pass (the threads of a column used to fetch their rows
    // themselves, four at a time: npart / 32 round trips on the update's dependency chain).
    __shared__ double psum[8][32];
    __shared__ double gy_part[64];
    __shared__ double gy_w[64][4];
    __shared__ double lp[SC_LP_ROWS * (NP + 1)];
    // The slice's operands do not depend on the coefficients: request them now, so that they arrive while the sums and
    // the coefficient stage run (up to 4 elements per thread: slices are 1024 long up to n = 65536).
    constexpr int EPT = 4;
    // [lo_sl, hi_sl): the elements this workgroup updates (its slice of scalar_slice(n) elements in the stand-alone stage)
    const bool pre = hi_sl - lo_sl <= 256 * EPT;
    double py[EPT], px[EPT], pp[EPT][NP];
    if (pre) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const long long i = lo_sl + tid + 256 * e;
            const bool in = i < hi_sl;
            py[e] = in ? y[i] : 0.0;
            px[e] = in ? xc[i] : 0.0;
#pragma unroll
            for (int j = 0; j < NP; ++j) pp[e][j] = in ? pend[(long long)j * n + i] : 0.0;
        }
    }
    if (GY) {  // (npart = scalar_groups(n) <= 64 in this mode; <= 8 wherever the host selects it: n <= 8192)
        const long long m = scalar_slice(n);
        if (npart <= 8) {
            // all slices in one sweep: 8 independent running sums per thread (slice b: i = b m + tid, + 256, ... as
            // k_scalar_dot_def sums it), their wave reductions back to back -- not npart rounds of load / reduce / store
            double sgy[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                sgy[b] = 0.0;
                const long long lo = (long long)b * m;
                const long long hi = (b < npart) ? ((lo + m < n) ? lo + m : n) : lo;
                for (long long i = lo + tid; i < hi; i += 256) sgy[b] += g_own[i] * y[i];
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const double w = wave_allreduce_sum(sgy[b]);
                if ((tid & 63) == 0 && b < npart) gy_w[b][tid >> 6] = w;
            }
        } else {
            for (int b = 0; b < npart; ++b) {
                const long long lo = (long long)b * m;
                const long long hi = (lo + m < n) ? lo + m : n;
                double sgy = 0.0;
#pragma unroll 4
                for (long long i = lo + tid; i < hi; i += 256) sgy += g_own[i] * y[i];
                sgy = wave_allreduce_sum(sgy);
                if ((tid & 63) == 0) gy_w[b][tid >> 6] = sgy;
            }
        }
        __syncthreads();
        if (tid < npart) gy_part[tid] = ((gy_w[tid][0] + gy_w[tid][1]) + gy_w[tid][2]) + gy_w[tid][3];
        __syncthreads();
    }
    {
        const int c = tid & 31, q = tid >> 5;
        double a = 0.0;
        for (int b0 = 0; b0 < npart; b0 += SC_LP_ROWS) {
            const int nb = (npart - b0 < SC_LP_ROWS) ? npart - b0 : SC_LP_ROWS;
            const int total = nb * (NP + 1);
            for (int k = tid; k < total; k += 256) lp[k] = partial[(long long)b0 * (NP + 1) + k];
            __syncthreads();
            if (c <= NP) {
                if (GY && c == 0) {
                    for (int b = q; b < nb; b += 8) a += gy_part[b0 + b];
                } else {
                    for (int b = q; b < nb; b += 8) a += lp[b * (NP + 1) + c];
                }
            }
            __syncthreads();
        }
        if (c <= NP) psum[q][c] = a;
    }
    __syncthreads();
    if (tid <= NP)
        dsum[tid] = ((((((psum[0][tid] + psum[1][tid]) + psum[2][tid]) + psum[3][tid]) + psum[4][tid]) + psum[5][tid]) +
                     psum[6][tid]) + psum[7][tid];
    __syncthreads();
    if (tid == 0) {
        double d[NP + 1];
#pragma unroll
        for (int k = 0; k <= NP; ++k) d[k] = dsum[k];
        double omega = d[0];  // g.(Q_base g)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            // slot `slot` is being written by the lead workgroup in this very launch: it is empty by
            // definition, so it is not read
            const double cd = (j == slot) ? 0.0 : cpend[j] * d[1 + j];  // c_j (v_j.g)
            bc_cd[j] = cd;
            omega = omega - cd * d[1 + j];
        }
        const double kappa = st->kappa_in;
        const double tsq = kappa * omega;  // src/ell.rs:105
        Coef cf;
        const CutParams cp = cp_dev ? *cp_dev : cp_val;
        const int status = calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);  // :106
        double roo = 0.0;
        if (status == ST_SUCCESS) roo = cf.rho / omega;  // :112
        if (lead) {
            st->tsq = tsq;
            st->omega = omega;
            st->status = status;
            if (status == ST_SUCCESS) {
                st->rho_over_omega = roo;
                st->ratio = cf.sigma / omega;   // :117
                st->kappa = kappa * cf.delta;   // :130 (deferred mode never runs with no_defer_trick)
                st->scale = 1.0;
                st->apply = 1;
                cpend[slot] = cf.sigma / omega;
                st->npend = slot + 1;
            } else {
                st->apply = 0;  // :107-109
            }
            queue_bookkeeping(st, status, tsq, queue_mode);
            if (q_status) {
                *q_status = status;
                *q_tsq = tsq;
            }
        }
        bc_roo = roo;
        bc_status = status;
    }
    __syncthreads();
    if (bc_status != ST_SUCCESS) return;
    const double roo = bc_roo;
    double cd[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) cd[j] = bc_cd[j];
    double* vnew = pend + (long long)slot * n;
    if (pre) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const long long i = lo_sl + tid + 256 * e;
            if (i < hi_sl) {
                double gt = py[e];
#pragma unroll
                for (int j = 0; j < NP; ++j) gt = gt - cd[j] * pp[e][j];
                vnew[i] = gt;                // slot `slot` was all zeros until now: its own term above was an exact 0
                xc[i] = px[e] - roo * gt;    // :113-115
            }
        }
        return;
    }
    for (long long i = lo_sl + tid; i < hi_sl; i += 256) {
        double gt = y[i];
#pragma unroll
        for (int j = 0; j < NP; ++j) gt = gt - cd[j] * pend[(long long)j * n + i];
        vnew[i] = gt;
        xc[i] = xc[i] - roo * gt;
    }
}

template <int NP, bool GY = false>
__global__ __launch_bounds__(256) void k_scalar_apply_def(long long n, const double* __restrict__ y,
                                                          double* __restrict__ xc, double* __restrict__ pend,
                                                          double* __restrict__ cpend,
                                                          const double* __restrict__ partial,
                                                          DevState* __restrict__ st, EllCalcDev calc,
                                                          const CutParams* __restrict__ cp_dev, CutParams cp_val,
                                                          int slot, int queue_mode, int* __restrict__ q_status,
                                                          double* __restrict__ q_tsq, int npart,
                                                          const double* __restrict__ g_own, LiveMirror* live_m = nullptr,
                                                          double* live_xc = nullptr, unsigned long long live_seq = 0,
                                                          unsigned* live_arrived = nullptr) {
    const long long m_sl = scalar_slice(n), lo_sl = (long long)blockIdx.x * m_sl;
    const long long hi_sl = (lo_sl + m_sl < n) ? lo_sl + m_sl : n;
    scalar_apply_def_body<NP, GY>((long long)blockIdx.x, lo_sl, hi_sl, n, y, xc, pend, cpend,
                                  partial, st, calc, cp_dev, cp_val, slot, queue_mode, q_status, q_tsq, npart, g_own);
    if (live_m) {   // a live update: results to the host from here (live_tail); `st` / `xc` without __restrict__ semantics from here on
        __threadfence();   // (the lead workgroup's state, this workgroup's slice: visible to the last workgroup's agent-scope loads)
        live_tail(st, xc, lo_sl, hi_sl, live_m, live_xc, live_seq, live_arrived);
    }
}

// Full-row GEMV pass of the deferred schedule (handles without the lower-triangle schedule: n < 8192, odd n) with
// the scalar stage's v_j . g dot products computed BESIDE it: the grid has scalar_groups(n) extra workgroups behind
// the row tiles; extra workgroup b does for slice b exactly what k_scalar_dot_def does for columns 1..NP (same thread
// mapping, same reduction shape, same bits) and takes its halted / kappa snapshots.  g . y cannot be formed here (y is
// this very launch's output): k_scalar_apply_def<NP, true> forms it, again in k_scalar_dot_def's shape.  Net effect:
// the scalar stage is one launch on the update's dependency chain instead of two (n = 4096: 13.6 -> ~9 us).
template <int RW, int UNR, int VEC, bool NT, int NP>
__global__ __launch_bounds__(256) void k_sweep_gemv_dots(const double* Q, long long ld, long long n, long long nrows,
                                                         long long row0, const double* __restrict__ gvec,
                                                         double* __restrict__ gv_out, DevState* __restrict__ st,
                                                         int reverse, unsigned ntiles, const double* __restrict__ pend,
                                                         double* __restrict__ partial) {
    __shared__ double red[4][RW > NP ? RW : NP];
    const int halted = st->halted;
    const int tid = threadIdx.x;
    if (blockIdx.x >= ntiles) {
        const long long b = (long long)blockIdx.x - ntiles;
        if (b == 0 && tid == 0) {
            st->halted_in = halted;
            st->kappa_in = st->kappa;
        }
        if (halted) return;
        const long long m = scalar_slice(n);
        const long long lo = b * m;
        const long long hi = (lo + m < n) ? lo + m : n;
        double sd[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) sd[j] = 0.0;
        for (long long i = lo + tid; i < hi; i += 256) {
            const double gi = gvec[i];
#pragma unroll
            for (int j = 0; j < NP; ++j) sd[j] += pend[(long long)j * n + i] * gi;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const double w = wave_allreduce_sum(sd[j]);
            if ((tid & 63) == 0) red[tid >> 6][j] = w;
        }
        __syncthreads();
        if (tid < NP) partial[b * (NP + 1) + 1 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        return;
    }
    if (halted) return;
    const long long tile = reverse ? (long long)ntiles - 1 - blockIdx.x : (long long)blockIdx.x;
    const long long row_base = tile * RW;
    if (row_base >= nrows) return;
    sweep_rows<RW, UNR, VEC, NT, false, true, false>(Q, const_cast<double*>(Q), ld, n, nrows, row0, row_base, nullptr, gvec,
                                                     gv_out, 0.0, 1.0, reinterpret_cast<double(*)[RW]>(&red[0][0]));
}

constexpr long long SCALAR_SPLIT_N = 8192;

// Small n (< SCALAR_SPLIT_N): the whole scalar stage in ONE workgroup of 1024 threads -- a second
// launch would cost more (~5 us) than one CU's bandwidth limit does.  Same arithmetic, its own fixed
// reduction shape (which form runs depends on n only, so results stay reproducible).
__global__ __launch_bounds__(1024) void k_scalar(long long n, const double* __restrict__ g,
                                                 const double* __restrict__ gt, double* __restrict__ xc,
                                                 DevState* __restrict__ st, EllCalcDev calc,
                                                 const CutParams* __restrict__ cp_dev, CutParams cp_val,
                                                 int no_defer_trick, int queue_mode,
                                                 int* __restrict__ q_status, double* __restrict__ q_tsq) {
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
        const CutParams cp = cp_dev ? *cp_dev : cp_val;  // queued cuts live in HBM, direct ones arrive by value
        const int status = calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);  // :106
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
        }
        queue_bookkeeping(st, status, tsq, queue_mode);
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

// Unconditional form: rebuild the strict upper triangle after lower-triangle-only apply passes.
__global__ __launch_bounds__(256) void k_mirror_lower_now(double* __restrict__ Q, long long ld, long long n) {
    __shared__ double tile[32][33];
    const long long bx = blockIdx.x, by = blockIdx.y;
    if (by < bx) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const long long r = by * 32 + k, c = bx * 32 + tx;
        tile[k][tx] = (r < n && c < n) ? Q[r * ld + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long long r = bx * 32 + k, c = by * 32 + tx;
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
