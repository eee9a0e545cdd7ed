// ellstable_kernels.hpp -- CDNA4 kernels for EllStable::update_core (src/ell_stable.rs:52-125).
//
// State (one n x n row-major buffer M, leading dimension ld, as in the reference):
//   diagonal      d[i]    = M[i][i]          ("inv(D)" entries,                     :72-75)
//   strict upper  U[j][i] = M[j][i], j < i   (= L[i][j] of the unit lower factor,   :65)
//   strict lower  S[i][j] = M[i][j], j < i   (scratch: the products U[j][i]*w[j],   :66)
// The reference is BUG-COMPATIBLY reproduced: the back substitution reads the scratch
// triangle (:96) and the factor update adds beta2 * S[l][j] (:116).
//
// One update, blocked by B = 128 rows/columns (block kb = indices [128 kb, 128 kb + 128)):
//
//   forward   w = L^-1 g            right-looking: for each block, a one-wave register solve of the
//             S <- U .* w            128x128 diagonal block (two 64-wide halves + a 64x64 mini panel),
//                                    then a grid-wide panel update of every column to its right
//                                    (reads 128 rows of U coalesced, writes the products transposed
//                                    through LDS as full 128-byte lines of S).
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
// strict left-to-right order only by the grouping into 32-row partial sums (deterministic).
#pragma once

#include <hip/hip_runtime.h>

#include "ell_kernels.hpp"

namespace ellhip {

constexpr int SB = 128;       // block size of the triangular solves (two 64-wide halves per diagonal block)
constexpr int SH = 64;        // half block = one wave's register solve
constexpr int SPANEL = 128;   // columns per panel workgroup (2 per lane)
constexpr int SLDS_PAD = 18;  // doubles per LDS tile row (16 + 2: keeps 16-byte alignment)
// Measurement builds only.  -DST_EXP_CHAIN_STEPS=k (results are then WRONG): the diagonal-block chains take k of their
// 64 steps, which shows how much of a block's time is the chain arithmetic (profiles/r02/ellstable_chain_share.txt).
// ST_STAMP(slot): tools/experiments/st_fwd_timeline.hip defines it to record wall-clock stamps per workgroup and step.
#ifndef ST_EXP_CHAIN_STEPS
#define ST_EXP_CHAIN_STEPS SH
#endif
#ifndef ST_STAMP
#define ST_STAMP(kb, slot)        // thread 0 of the workgroup
#define ST_STAMP_LANE0(kb, slot)  // lane 0 of the calling wave
#endif

// ------------------------------------------------------------------------------ forward ------
// The diagonal-block solve is a length-128 dependency chain, so it is kept free of memory latency:
// all 4 waves of the owning workgroup prefetch the three 64x64 pieces of the block (AA, AB, BB) and
// the 128 diagonal entries into registers before the panel work, park them in LDS, three waves share the
// chain (st_fwd_diag_block) out of registers and leave the products in LDS, and all 4 waves write them back to
// the scratch triangle as coalesced rows.
constexpr int BLK_PITCH = 65;  // doubles per LDS row of a 64x64 piece: odd, so the column reads AND the
                               // transposed product writes of the chain wave are both bank-conflict-free

struct Blk3 {  // one thread's share (8 x 16 B per piece) of the three 64x64 pieces of a diagonal block
    double2_t aa[8], ab[8], bb[8];
};

// Broadcast lane j's value to the whole wave through the scalar unit (v_readlane_b32 x2, a few
// cycles) -- __shfl would go through the LDS crossbar (ds_bpermute, ~100 cycles) on every chain step.
__device__ __forceinline__ double lane_bcast(double v, int j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}

// piece rows r = (tid >> 5) + 8k, column pair cp = tid & 31
// (`tid`: the thread's index inside the group of 256 threads that shares the block -- threadIdx.x everywhere except in
// the paired persistent solves, where a workgroup of 512 threads holds two such groups)
__device__ __forceinline__ void st_load_piece(const double* __restrict__ M, long long ld, long long n,
                                              long long r0, long long c0, double2_t (&v)[8], int tid) {
    const int tr = tid >> 5, cp = tid & 31;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        long long r = r0 + tr + 8 * k, c = c0 + 2 * cp;
        if (r > n - 1) r = n - 1;
        if (c > n - 1) c = 0;  // outside the matrix: any valid address, the value is never used
        v[k] = *reinterpret_cast<const double2_t*>(M + r * ld + c);
    }
}
__device__ __forceinline__ void st_park_piece(double* __restrict__ lds, const double2_t (&v)[8], int tid) {
    const int tr = tid >> 5, cp = tid & 31;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // two 8-byte writes: rows are not 16-byte aligned with the odd pitch
        lds[(tr + 8 * k) * BLK_PITCH + 2 * cp] = v[k].x;
        lds[(tr + 8 * k) * BLK_PITCH + 2 * cp + 1] = v[k].y;
    }
}

// Chain over 64 columns held in registers (u[j] = piece[j][lane]); the products stay in u and are parked transposed
// (piece[lane][j]) by the caller.  src/ell_stable.rs:61-69
// Critical path per step: subtract -> broadcast lane j+1 -> multiply -> subtract.  The broadcast reads the unselected
// difference `t` (lane j+1 is always an active lane of step j), so the select that protects the already-final lanes
// <= j runs beside the chain, not in it.
__device__ __forceinline__ double st_fwd_chain_regs(double (&u)[SH], double wi) {
    const int lane = threadIdx.x & 63;
    double t = wi;
#pragma unroll
    for (int j = 0; j < ST_EXP_CHAIN_STEPS; ++j) {
        const double wj = lane_bcast(t, j);  // lane j's value is final after step j-1
        const double v = u[j] * wj;          // :65
        u[j] = v;                            // :66 (meaningful for j < lane only)
        t = wi - v;                          // :67
        wi = (lane > j) ? t : wi;
    }
    return wi;
}
// Rows A (final; wa[j] = w[A0+j] in LDS, read as broadcasts) applied to the 64 columns of B held in registers.
__device__ __forceinline__ double st_fwd_mini_regs(double (&u)[SH], const double* __restrict__ wa, double wb) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four interleaved partial sums (16 steps deep instead of 64)
#pragma unroll
    for (int j = 0; j < SH; j += 4) {
        const double v0 = u[j] * wa[j];
        const double v1 = u[j + 1] * wa[j + 1];
        const double v2 = u[j + 2] * wa[j + 2];
        const double v3 = u[j + 3] * wa[j + 3];
        u[j] = v0, u[j + 1] = v1, u[j + 2] = v2, u[j + 3] = v3;
        a0 += v0, a1 += v1, a2 += v2, a3 += v3;
    }
    return wb - ((a0 + a1) + (a2 + a3));
}
// All threads: write the parked products of a piece to S rows R0.., columns C0..; `lower_only`: only
// the strict lower triangle of the piece belongs to S (the rest is factor / diagonal, never touched).
__device__ __forceinline__ void st_store_piece(double* __restrict__ M, long long ld, long long n,
                                               long long R0, long long C0, const double* __restrict__ piece,
                                               bool lower_only, int tid) {
    const int tr = tid >> 5, cp = tid & 31;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = tr + 8 * k;
        const long long row = R0 + r;
        if (row >= n) continue;
        const double2_t v = {piece[r * BLK_PITCH + 2 * cp], piece[r * BLK_PITCH + 2 * cp + 1]};
        double* dst = M + row * ld + C0 + 2 * cp;
        if (!lower_only) {
            *reinterpret_cast<double2_t*>(dst) = v;  // C0 + 63 < R0 <= row: always inside the matrix
        } else {
            if (2 * cp < r) dst[0] = v.x;
            if (2 * cp + 1 < r) dst[1] = v.y;
        }
    }
}

// ---- in-launch hand-off of a finished 128-vector between workgroups (persistent solves) ------------
// cdna_hip_programming.md Guideline 16, write-through form: the payload is stored with agent-scope
// relaxed atomics (sc1, 8 bytes each) by ONE wave, that wave drains its stores (s_waitcnt vmcnt(0)) and
// its lane 0 stores the flag; a consumer polls the flag with ONE lane (relaxed, agent), then a
// workgroup barrier, then every load of the payload is again an sc1 load.  Spins are bounded: on
// time-out the error word is set and the workgroup leaves (results are then invalid, the GPU is not hung).
__device__ __forceinline__ void st_publish_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double st_published_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_raise_flag(int* flag, int epoch) {  // called by the storing wave, all lanes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Data-as-flag form of the hand-off (persistent BACKWARD solve): the publish buffer qpub holds ST_SENTINEL (a
// signalling-NaN payload no arithmetic produces) until its owner stores the value, so a consumer polls the VALUE
// it needs and the separate flag round trip (store flag -> poll flag -> barrier -> load payload) disappears
// (backward solve 1.16 -> 1.07 ms at n = 16384).  k_st_post re-arms qpub before every backward solve.
constexpr unsigned long long ST_SENTINEL_BITS = 0x7FF4DEADC0DEBEEFull;
__device__ __forceinline__ double st_sentinel() { return __longlong_as_double((long long)ST_SENTINEL_BITS); }
__device__ __forceinline__ bool st_poll_value(const double* p, double& out) {
    for (int spin = 0; spin < (1 << 20); ++spin) {
        const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned long long)__double_as_longlong(v) != ST_SENTINEL_BITS) {
            out = v;
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    out = 0.0;
    return false;
}
// One lane polls; returns false on time-out.  Call from thread 0 only.
__device__ __forceinline__ bool st_wait_flag(const int* flag, int epoch) {
    for (int spin = 0; spin < (1 << 20); ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// Whole 128-wide diagonal block at J0, called by all 256 threads of one workgroup.
//   blk: prefetched pieces; dreg: this thread's diagonal entry M[J0+t][J0+t] (t < 128);
//   lds: 3 * 64 * BLK_PITCH doubles; wpart[128]: partial w of the block's columns (LDS).
//   PUBLISH: w is handed to other workgroups inside this launch (write-through stores + flag) before the
//   parked products are written back.
// The length-128 dependency chain is split over three waves so that only the chain steps themselves are serial:
//   wave 0  column loads of AA, chain A                      | wave 1: column loads of AB | wave 2: column loads of BB
//   -- barrier --   (w_A in LDS)
//   wave 0  products of AA back to LDS, z / gg of A          | wave 1: mini panel A -> B (64 x 64, tree-summed)
//   -- barrier --   (partial w_B in LDS)
//   wave 1  products of AB back to LDS                       | wave 2: chain B, publish all 128 values + flag, z / gg of B
// (one wave doing everything in sequence spent ~0.7 us of its ~4 us per block on the two later column loads and the
// two earlier product stores; the arithmetic and its order are unchanged, so are the bits).
template <bool PUBLISH>
__device__ __forceinline__ void st_fwd_diag_block(double* __restrict__ M, long long ld, long long n, long long J0,
                                                  const Blk3& blk, double dreg, double* __restrict__ lds,
                                                  double* __restrict__ dlds, const double* __restrict__ wpart,
                                                  double* __restrict__ w, double* __restrict__ z,
                                                  double* __restrict__ gg, int* flag = nullptr, int epoch = 0,
                                                  bool parked = false, int tid = threadIdx.x,
                                                  double* __restrict__ wout = nullptr) {
    // tid: index inside the 256-thread group that runs this block (all of them call; nobody else may).
    // wout (LDS, 128 doubles): the block's final w for a consumer in the same workgroup, complete at the last barrier.
    __shared__ double wab[2][SH];  // [0]: final w of half A; [1]: partial w of half B after the mini panel
    double* pAA = lds;
    double* pAB = lds + SH * BLK_PITCH;
    double* pBB = lds + 2 * SH * BLK_PITCH;
    if (!parked) {  // (the persistent solve parks while it waits for the previous block: st_fwd_park)
        st_park_piece(pAA, blk.aa, tid);
        st_park_piece(pAB, blk.ab, tid);
        st_park_piece(pBB, blk.bb, tid);
        if (tid < SB) dlds[tid] = dreg;
        __syncthreads();
    }
    const bool has_b = J0 + SH < n;
    const int wave = tid >> 6, lane = tid & 63;
    double* mine = wave == 0 ? pAA : (wave == 1 ? pAB : pBB);
    const bool busy = wave == 0 || (has_b && wave <= 2);
    double u[SH];
    if (busy) {
#pragma unroll
        for (int j = 0; j < SH; ++j) u[j] = mine[j * BLK_PITCH + lane];
    }
    if (wave == 0) {
        ST_STAMP_LANE0(J0 / SB, 8);
        wab[0][lane] = st_fwd_chain_regs(u, wpart[lane]);
        ST_STAMP_LANE0(J0 / SB, 9);
    }
    __syncthreads();
    if (wave == 1) ST_STAMP_LANE0(J0 / SB, 10);
    if (wave == 0) {
        const double wA = wab[0][lane];
        if (wout) wout[lane] = wA;
        const long long c = J0 + lane;
        if (c < n) {
            const double zi = wA * dlds[lane];  // :74
            if (!PUBLISH) w[c] = wA;
            else if (!has_b) st_publish_store(w + c, wA);
            z[c] = zi;
            gg[c] = zi * wA;  // :81
        }
        if (PUBLISH && !has_b) st_raise_flag(flag, epoch);
    } else if (wave == 1 && has_b) {
        wab[1][lane] = st_fwd_mini_regs(u, wab[0], wpart[lane + SH]);
        ST_STAMP_LANE0(J0 / SB, 11);
    }
    __syncthreads();
    if (wave == 2 && has_b) {
        ST_STAMP_LANE0(J0 / SB, 12);
        const double wB = st_fwd_chain_regs(u, wab[1][lane]);
        ST_STAMP_LANE0(J0 / SB, 13);
        if (wout) wout[SH + lane] = wB;
        const long long c = J0 + SH + lane;
        if (PUBLISH) {  // all 128 values by this wave, then its flag store (half A always lies inside the matrix here)
            st_publish_store(w + J0 + lane, wab[0][lane]);
            if (c < n) st_publish_store(w + c, wB);
            ST_STAMP_LANE0(J0 / SB, 7);
            st_raise_flag(flag, epoch);
        } else if (c < n) {
            w[c] = wB;
        }
        if (c < n) {
            const double zi = wB * dlds[lane + SH];
            z[c] = zi;
            gg[c] = zi * wB;
        }
    }
    if (busy) {  // parked products: piece[lane][j] (transposed in place; every lane read its whole column long ago)
#pragma unroll
        for (int j = 0; j < SH; ++j) mine[lane * BLK_PITCH + j] = u[j];
    }
    __syncthreads();
    st_store_piece(M, ld, n, J0, J0, pAA, true, tid);                  // S[A][A], strict lower
    if (has_b) {
        st_store_piece(M, ld, n, J0 + SH, J0, pAB, false, tid);        // S[B][A], full
        st_store_piece(M, ld, n, J0 + SH, J0 + SH, pBB, true, tid);    // S[B][B], strict lower
    }
}

// The same block for a workgroup that has TIME before the values it waits for arrive (k_st_fwd_helped's chain
// workgroups): the waves fetch their columns of the parked pieces into registers beforehand (st_fwd_diag_cols: 0.5 us
// of LDS reads that otherwise sit between the arrival of the values and chain A, and compete with it), and the mini
// panel A -> B is shared by waves 1 and 3 (interleaved partial sums a0, a1 in wave 1 and a2, a3 in wave 3, combined
// as (a0 + a1) + (a2 + a3) exactly like st_fwd_mini_regs: 0.7 -> 0.35 us).  Same bits.  Always publishes.
//   per block measured before: values seen 0.7 | sums 0.6 | columns 0.5 | chain A 1.26 | mini 0.69 | chain B 0.83 us
// Half of the mini panel: the residues O, O + 1 of j mod 4 (O = 0: the partial sums a0, a1 of st_fwd_mini_regs; O = 2: a2, a3)
template <int O>
__device__ __forceinline__ double st_fwd_mini_half(double (&u)[SH], const double* __restrict__ wa) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = O; j < SH; j += 4) {
        const double v0 = u[j] * wa[j];
        const double v1 = u[j + 1] * wa[j + 1];
        u[j] = v0, u[j + 1] = v1;
        a0 += v0, a1 += v1;
    }
    return a0 + a1;
}
__device__ __forceinline__ void st_fwd_diag_cols(const double* __restrict__ lds, bool has_b, double (&u)[SH]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double* mine = lds + (wave == 0 ? 0 : (wave == 2 ? 2 : 1)) * (SH * BLK_PITCH);  // waves 1 and 3: piece AB
    if (wave == 0 || has_b) {
#pragma unroll
        for (int j = 0; j < SH; ++j) u[j] = mine[j * BLK_PITCH + lane];
    }
}
template <bool STORE = true>
__device__ __forceinline__ void st_fwd_diag_block_pre(double* __restrict__ M, long long ld, long long n, long long J0,
                                                      double* __restrict__ lds, const double* __restrict__ dlds,
                                                      const double* __restrict__ wpart, double* __restrict__ w,
                                                      double* __restrict__ z, double* __restrict__ gg, int* flag,
                                                      int epoch, double (&u)[SH]) {
    __shared__ double wa[SH];        // final w of half A
    __shared__ double mini[2][SH];   // (a0 + a1), (a2 + a3) of the mini panel
    double* pAA = lds;
    double* pAB = lds + SH * BLK_PITCH;
    double* pBB = lds + 2 * SH * BLK_PITCH;
    const bool has_b = J0 + SH < n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        ST_STAMP_LANE0(J0 / SB, 8);
        wa[lane] = st_fwd_chain_regs(u, wpart[lane]);
        ST_STAMP_LANE0(J0 / SB, 9);
    }
    __syncthreads();
    if (wave == 1) ST_STAMP_LANE0(J0 / SB, 10);
    if (wave == 0) {
        const double wA = wa[lane];
        const long long c = J0 + lane;
        if (c < n) {
            const double zi = wA * dlds[lane];  // :74
            if (!has_b) st_publish_store(w + c, wA);
            z[c] = zi;
            gg[c] = zi * wA;  // :81
        }
        if (!has_b) st_raise_flag(flag, epoch);
    } else if (has_b && wave == 1) {
        mini[0][lane] = st_fwd_mini_half<0>(u, wa);
        ST_STAMP_LANE0(J0 / SB, 11);
    } else if (has_b && wave == 3) {
        mini[1][lane] = st_fwd_mini_half<2>(u, wa);
    }
    __syncthreads();
    if (wave == 2 && has_b) {
        ST_STAMP_LANE0(J0 / SB, 12);
        const double wB = st_fwd_chain_regs(u, wpart[lane + SH] - (mini[0][lane] + mini[1][lane]));
        ST_STAMP_LANE0(J0 / SB, 13);
        const long long c = J0 + SH + lane;
        st_publish_store(w + J0 + lane, wa[lane]);  // all 128 values by this wave, then its flag store
        if (c < n) st_publish_store(w + c, wB);
        ST_STAMP_LANE0(J0 / SB, 7);
        st_raise_flag(flag, epoch);
        if (c < n) {
            const double zi = wB * dlds[lane + SH];
            z[c] = zi;
            gg[c] = zi * wB;
        }
    }
    if constexpr (!STORE) return;  // mirrored layout: nobody parks the products (the scratch triangle is not kept)
    // parked products: piece[lane][j] (transposed in place; every lane read its whole column long ago)
    if (wave == 0 || (has_b && wave == 2)) {
        double* mine = wave == 0 ? pAA : pBB;
#pragma unroll
        for (int j = 0; j < SH; ++j) mine[lane * BLK_PITCH + j] = u[j];
    } else if (has_b && wave == 1) {
#pragma unroll
        for (int j = 0; j < SH; j += 4) pAB[lane * BLK_PITCH + j] = u[j], pAB[lane * BLK_PITCH + j + 1] = u[j + 1];
    } else if (has_b) {
#pragma unroll
        for (int j = 2; j < SH; j += 4) pAB[lane * BLK_PITCH + j] = u[j], pAB[lane * BLK_PITCH + j + 1] = u[j + 1];
    }
    __syncthreads();
    st_store_piece(M, ld, n, J0, J0, pAA, true, threadIdx.x);                  // S[A][A], strict lower
    if (has_b) {
        st_store_piece(M, ld, n, J0 + SH, J0, pAB, false, threadIdx.x);        // S[B][A], full
        st_store_piece(M, ld, n, J0 + SH, J0 + SH, pBB, true, threadIdx.x);    // S[B][B], strict lower
    }
}

// Mirrored layout: a 64 x 64 piece of a diagonal block held in registers (rows R0 + r, column pairs as st_load_piece laid
// them out) times the row scales r[row] -- the factor entries the eager kernels would have found in memory.  Entries on or
// left of the diagonal get multiplied too; the chains never look at them.
__device__ __forceinline__ void st_mul_piece_rows(long long n, long long R0, double2_t (&v)[8], const double* __restrict__ rs,
                                                  int tid) {
    const int tr = tid >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        long long r = R0 + tr + 8 * k;
        if (r > n - 1) r = n - 1;
        const double x = rs[r];
        v[k].x = v[k].x * x;
        v[k].y = v[k].y * x;
    }
}

__device__ __forceinline__ void st_fwd_park(const Blk3& blk, double dreg, double* __restrict__ lds,
                                            double* __restrict__ dlds, int tid = threadIdx.x) {
    st_park_piece(lds, blk.aa, tid);
    st_park_piece(lds + SH * BLK_PITCH, blk.ab, tid);
    st_park_piece(lds + 2 * SH * BLK_PITCH, blk.bb, tid);
    if (tid < SB) dlds[tid] = dreg;
}

__device__ __forceinline__ void st_prefetch_block(const double* __restrict__ M, long long ld, long long n,
                                                  long long J0, Blk3& blk, double& dreg, int tid = threadIdx.x) {
    st_load_piece(M, ld, n, J0, J0, blk.aa, tid);
    st_load_piece(M, ld, n, J0, J0 + SH, blk.ab, tid);
    st_load_piece(M, ld, n, J0 + SH, J0 + SH, blk.bb, tid);
    long long dj = J0 + (tid & (SB - 1));
    if (dj > n - 1) dj = n - 1;
    dreg = M[dj * ld + dj];
}

constexpr int ST_LDS_DOUBLES = 3 * SH * BLK_PITCH;  // 96 KiB: the three parked pieces (the panel tile aliases it)

// ------------------------------------------------------------------ the mirrored layout (ELLHIP_OPT_STABLE_SOLVE = 3) ---
// The reference moves 24 n^2 bytes per update, the eager kernels above 20 n^2, and in both solves the scratch triangle is
// pure overhead: the forward solve WRITES S[i][j] = fl(U[j][i] w[j]) (src/ell_stable.rs:66, transposed through LDS: what
// paces its helper workgroups), the backward solve reads it back (:96), the factor update reads or recomputes it a third
// time (:116) and rewrites U.  Two observations remove all of that:
//   (1) the factor update is a ROW SCALING.  U[j][l] += beta2[j] S[l][j] = beta2[j] fl(U[j][l] w[j]) (:107-121) multiplies
//       row j of U by (1 + beta2[j] w[j]) -- the same factor for the whole row.  So the matrix need not be rewritten at all:
//       the handle keeps U_base and one running product per row, r[j] <- r[j] (1 + beta2[j] w[j]) (k_st_post, O(n)), and
//       every kernel that loads a tile uses fl(U_base[j][l] r[j]).  One rounding per use instead of two per update: the
//       factor the solves see differs from the eager kernels' by ~1e-16 relative per update (not bit-identical; the states
//       stay within the parity tolerance of the CPU path by five orders of magnitude, tests/test_gpu_ellstable_factor.py).
//   (2) the backward solve's operand S[j][t] = fl(U[t][j] w[t]) is a COLUMN of U: with the factor stored a second time below
//       the diagonal (L[j][t] = U[t][j], copied once when the layout is entered -- U_base never changes afterwards) the solve
//       reads the addresses it used to read S from, with the lane's own constants r[t] and w[t]: the old kernel with two more
//       multiplies per element, the same sums in the same order.
// Per update: forward reads 4 n^2, backward reads 4 n^2 -- 8 n^2 bytes, nothing is written but O(n) vectors; every tile is
// loaded by exactly one workgroup per solve.  The reference's buffer (U scaled, scratch triangle of the last forward solve)
// is materialised when somebody observes it (k_st_mirror_leave, k_st_unscale_upper) and the layout entered again by the
// next update; the host does the same every 256 updates so that r stays a product of few factors.
// Which r buffer each kernel reads is device state (StPend, kept by k_st_mid): the backward solve of a successful cut still
// needs the scales its forward solve used, while k_st_post already writes the next ones.
struct StPend {
    int rsel;        // r buffer (0 / 1) the next forward solve reads (k_st_post of a successful cut writes it)
    int r_bwd;       // r buffer the solves of the update in flight read (set by k_st_mid)
    int r_fwd_last;  // r buffer the last forward solve that really ran read: the scratch triangle is made of it and w_keep
    int mirrored;    // the lower triangle holds the mirrored factor (k_st_mirror_mark; 0: it holds the scratch triangle)
    int have_w;      // a forward solve has run since the layout was entered: w_keep holds its result
    int keep_now;    // the update in flight ran its forward solve (not a no-op of a halted queue): k_st_post keeps its w
    int pad_[2];
};

__global__ __launch_bounds__(256) void k_st_fwd_first(double* __restrict__ M, long long ld, long long n,
                                                      const double* __restrict__ g, double* __restrict__ w,
                                                      double* __restrict__ z, double* __restrict__ gg,
                                                      const DevState* __restrict__ st) {
    if (st->halted) return;
    __shared__ __attribute__((aligned(16))) double lds[ST_LDS_DOUBLES];
    __shared__ double dlds[SB];
    __shared__ double wpart[SB];
    Blk3 blk;
    double dreg;
    st_prefetch_block(M, ld, n, 0, blk, dreg);
    if (threadIdx.x < SB) wpart[threadIdx.x] = (threadIdx.x < n) ? g[threadIdx.x] : 0.0;
    st_fwd_diag_block<false>(M, ld, n, 0, blk, dreg, lds, dlds, wpart, w, z, gg);
}

// Panel update for block kb (rows J0..J0+127, already final in w) over columns >= J0 + 128, 128 columns
// per workgroup; each of the 4 waves takes 32 rows in two 16-row passes.  Workgroup 0 then solves the
// next diagonal block (its 128 columns are exactly that block).
__global__ __launch_bounds__(256) void k_st_fwd_step(double* __restrict__ M, long long ld, long long n,
                                                     long long kb, double* __restrict__ w,
                                                     double* __restrict__ z, double* __restrict__ gg,
                                                     const DevState* __restrict__ st) {
    if (st->halted) return;
    __shared__ __attribute__((aligned(16))) double lds[ST_LDS_DOUBLES];  // panel tiles, then the parked pieces
    __shared__ double part[4][SPANEL];
    __shared__ double wnext[SPANEL];
    __shared__ double dlds[SB];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long J0 = kb * SB;
    const long long Jend = J0 + SB;  // < n: there are columns to the right, so all 128 rows exist
    const long long c0 = Jend + (long long)blockIdx.x * SPANEL;
    const long long c = c0 + 2 * lane;  // this lane's two columns c, c+1 (ld is even: 16-byte aligned)
    const long long cl = (c < n) ? c : 0;
    const int piece = lane & 7;  // which 16-byte piece of a 128-byte line this lane stores
    const bool owner = blockIdx.x == 0;

    Blk3 blk;
    double dreg = 0.0;
    if (owner) st_prefetch_block(M, ld, n, Jend, blk, dreg);  // independent of w: overlaps the panel work

    double p0 = 0.0, p1 = 0.0;
    double* t = lds + wv * (SPANEL * SLDS_PAD);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long r0 = J0 + 32 * wv + 16 * h;  // 16 rows of this pass
        double2_t u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = *reinterpret_cast<const double2_t*>(M + (r0 + r) * ld + cl);
        if (h) __syncthreads();  // the tile of the previous pass has been drained
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double wj = w[r0 + r];  // wave-uniform
            const double v0 = u[r].x * wj;
            const double v1 = u[r].y * wj;
            p0 += v0;
            p1 += v1;
            t[(2 * lane) * SLDS_PAD + r] = v0;
            t[(2 * lane + 1) * SLDS_PAD + r] = v1;
        }
        __syncthreads();
        // S[col][r0 .. r0+16) for the 128 columns: 8 lanes x 16 B = one full 128-byte line per column
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int col_local = 8 * k + (lane >> 3);
            const long long col = c0 + col_local;
            if (col < n) {
                const double2_t v = *reinterpret_cast<const double2_t*>(&t[col_local * SLDS_PAD + 2 * piece]);
                *reinterpret_cast<double2_t*>(M + col * ld + r0 + 2 * piece) = v;
            }
        }
    }
    part[wv][2 * lane] = p0;
    part[wv][2 * lane + 1] = p1;
    __syncthreads();
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
    if (!owner) return;
    __syncthreads();  // panel tiles are dead from here on: `lds` is reused for the parked pieces
    st_fwd_diag_block<false>(M, ld, n, Jend, blk, dreg, lds, dlds, wnext, w, z, gg);
}

// Persistent forward solve: ONE launch, workgroup s owns the 128 columns of block s.  It applies the row
// blocks kb = 0..s-1 to its strip as soon as their w is published (flag kb), keeping the strip's partial w
// in LDS, then solves its own diagonal block and publishes.  The dependency chain (n steps + one hand-off
// per block) is the critical path; all panel traffic overlaps with it.  grid = ceil(n/128) <= #CUs.
__global__ __launch_bounds__(256) void k_st_fwd_persist(double* __restrict__ M, long long ld, long long n,
                                                        const double* __restrict__ g, double* __restrict__ w,
                                                        double* __restrict__ z, double* __restrict__ gg,
                                                        int* __restrict__ flags, int* __restrict__ err, int epoch,
                                                        const DevState* __restrict__ st) {
    if (st->halted) return;
    __shared__ __attribute__((aligned(16))) double lds[ST_LDS_DOUBLES];  // panel tiles, then the parked pieces
    __shared__ double part[4][SPANEL];
    __shared__ double wstrip[SPANEL];
    __shared__ double wblk[SB];
    __shared__ double dlds[SB];
    __shared__ int ok;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long sblk = blockIdx.x;
    const long long c0 = sblk * SB;
    const long long c = c0 + 2 * lane;
    const long long cl = (c < n) ? c : 0;
    const int piece = lane & 7;

    Blk3 blk;
    double dreg = 0.0;
    if (sblk == 0) st_prefetch_block(M, ld, n, c0, blk, dreg);  // no panel work: fetch the own block right away
    if (threadIdx.x < SPANEL) wstrip[threadIdx.x] = (c0 + threadIdx.x < n) ? g[c0 + threadIdx.x] : 0.0;

    double* t = lds + wv * (SPANEL * SLDS_PAD);
    // The products S[col][row] = U[row][col] * w[row] of a row block are needed by nobody before the backward
    // solve, so for the LAST row block (the one this workgroup's own solve is waiting for) they are written back
    // only after the own block has been solved and published: the two LDS transposes, the global stores and their
    // four barriers leave the critical path of the dependency chain.  They are RECOMPUTED then (the 128 x 128
    // factor entries are re-read, L2-resident; w of that block is still in LDS) rather than kept in registers
    // across the diagonal-block solve: keeping them cost 128 VGPRs there and made the kernel spill (472 B/lane).
    double2_t u[2][16];
    auto write_back = [&](long long J0, bool recompute) {
        if (recompute) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double2_t f = *reinterpret_cast<const double2_t*>(M + (r0 + r) * ld + cl);
                    const double wj = wblk[32 * wv + 16 * h + r];
                    u[h][r].x = f.x * wj;
                    u[h][r].y = f.y * wj;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
            if (h) __syncthreads();  // the tile of the previous pass has been drained
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[(2 * lane) * SLDS_PAD + r] = u[h][r].x;
                t[(2 * lane + 1) * SLDS_PAD + r] = u[h][r].y;
            }
            __syncthreads();
            // S[col][r0 .. r0+16) for the 128 columns: 8 lanes x 16 B = one full 128-byte line per column
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int col_local = 8 * k + (lane >> 3);
                const long long col = c0 + col_local;
                if (col < n) {
                    const double2_t v = *reinterpret_cast<const double2_t*>(&t[col_local * SLDS_PAD + 2 * piece]);
                    *reinterpret_cast<double2_t*>(M + col * ld + r0 + 2 * piece) = v;
                }
            }
        }
    };
    for (long long kb = 0; kb < sblk; ++kb) {
        const long long J0 = kb * SB;  // all 128 rows exist: J0 + 128 <= c0 < n
        ST_STAMP(kb, 0);
        // nothing below depends on the flag except w: request both passes' rows of U (and, in the last
        // iteration, the own diagonal block) before waiting, so their HBM latency is hidden behind the wait
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[h][r] = *reinterpret_cast<const double2_t*>(M + (r0 + r) * ld + cl);
        }
        if (kb == sblk - 1) {
            // the own diagonal block: fetched AND parked in LDS while the previous block's solve is still running
            // (`lds` is free: the last row block's products are written back after the own solve)
            st_prefetch_block(M, ld, n, c0, blk, dreg);
            st_fwd_park(blk, dreg, lds, dlds);
            ST_STAMP(kb, 4);
        }
        // (the data-as-flag hand-off of the backward solve was tried here too: 255 workgroups x 128 lanes polling the
        // values slowed the publishing wave down, forward solve 1.25 -> 1.34 ms; one polling lane per workgroup it is)
        if (kb == sblk - 1) {
            // next in the chain: poll the 128 values themselves (the buffer is all-sentinel when the launch starts),
            // no flag round trip and no wait for the publisher's store drain.  Only ONE workgroup polls a block's
            // values at any time -- all of them doing so slowed the publishing wave down (1.25 -> 1.34 ms).
            if (threadIdx.x == 0) ok = 1;
            __syncthreads();
            if (threadIdx.x < SB) {
                double v = 0.0;
                if (!st_poll_value(w + J0 + threadIdx.x, v)) ok = 0;
                wblk[threadIdx.x] = v;
            }
            __syncthreads();
            if (!ok) {
                if (threadIdx.x == 0) atomicExch(err, 1);
                return;
            }
        } else {
            if (threadIdx.x == 0) ok = st_wait_flag(flags + kb, epoch) ? 1 : 0;
            __syncthreads();
            if (!ok) {
                if (threadIdx.x == 0) atomicExch(err, 1);
                return;
            }
            if (threadIdx.x < SB) wblk[threadIdx.x] = st_published_load(w + J0 + threadIdx.x);
            __syncthreads();
        }
        ST_STAMP(kb, 1);
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double wj = wblk[32 * wv + 16 * h + r];
                const double v0 = u[h][r].x * wj;
                const double v1 = u[h][r].y * wj;
                p0 += v0;
                p1 += v1;
                u[h][r].x = v0;  // the product replaces the factor entry (src/ell_stable.rs:66)
                u[h][r].y = v1;
            }
        }
        part[wv][2 * lane] = p0;
        part[wv][2 * lane + 1] = p1;
        __syncthreads();
        if (threadIdx.x < SPANEL) {
            const double s4 = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) +
                              part[3][threadIdx.x];
            wstrip[threadIdx.x] = wstrip[threadIdx.x] - s4;
        }
        ST_STAMP(kb, 2);
        if (kb + 1 < sblk) {
            write_back(J0, false);
            __syncthreads();  // tiles drained, part reusable
        }
        ST_STAMP(kb, 3);
    }
    __syncthreads();
    st_fwd_diag_block<true>(M, ld, n, c0, blk, dreg, lds, dlds, wstrip, w, z, gg, flags + sblk, epoch, sblk > 0);
    ST_STAMP(sblk, 5);
    if (sblk > 0) {
        __syncthreads();  // the parked pieces have been written back: `lds` is free for the transposes
        write_back((sblk - 1) * SB, true);
    }
    ST_STAMP(sblk, 6);
}

// Persistent forward solve with HELPER workgroups: two workgroups per 128-block s, on two CUs.
//   helper s (dispatched first)  applies the row blocks kb = 0 .. s-2 to the strip's partial w as their w is published
//                                (the rows of the next step are requested a step ahead, two register buffers) and hands
//                                the 128 partial sums to
//   chain s                      whose diagonal block is parked in LDS and whose rows of the one row block it still has
//                                to apply (s-1) are in registers well before w of block s-1 comes out.  It takes the
//                                helper's sums, applies block s-1, solves, publishes.
// The products S[i][j] = U[j][i] w[j] (src/ell_stable.rs:66) are needed by nobody before the backward solve; writing
// them (transposed through LDS, 2.3 us per row block) is shared so that neither workgroup sets the pace: the helper
// writes every 3rd row block it has just applied (ST_HELPER_WRITES), the chain workgroup recomputes and writes the others
// while it has nothing else to do (it stops once block s-5 is solved) and, after its own hand-over, whatever is left plus
// row block s-1.  A CU moves ~30 GB/s whatever its threads keep in flight, so the split is chosen by BYTES per step:
// helper 128 KB of rows + 128/3 KB of products, chain workgroup 2/3 of (128 KB re-read + 128 KB written) -- 171 KB
// each (every 2nd: 192 / 128 KB, 5.9 us per block; every 3rd or 4th: 5.7).
// Why: with one workgroup per block the chain period is (R + C) / 2, R = what the workgroup still has to do for row
// block s-2 after that block's hand-over plus fetching and parking its own block (10.7 us at n = 16384: flag 1.7, sums
// 0.6, write-back 2.2, 224 KB of rows + own block 6.2), C = 4.0 us from the arrival of block s-1's values to its own
// hand-over (tools/experiments/st_fwd_timeline.hip, profiles/r02/ellstable_fwd_timeline.txt).  A workgroup moves
// 128 KB in about 1 us at best (64 B / clock / CU): a helper that wrote every row block back (4.8 us per step) set the
// pace itself (6.4 us per block); with every other one it averages 3.7 us per step, ahead of the chain.
// Same arithmetic in the same order per column, same products: identical bits.  grid = 2 * ceil(n/128) <= the resident
// limit (every workgroup waits only for workgroups dispatched before it); beyond that the handle uses k_st_fwd_persist.
// hpart: n doubles, all-sentinel at launch (k_st_post re-arms it after every solve), the helpers' hand-over buffer.
constexpr long long ST_DUTY_STOP = 5;  // the chain workgroup of block s stops writing once block s - 5 is solved
#ifndef ST_HELPER_WRITES
#define ST_HELPER_WRITES 3             // the helper writes every 3rd row block's products, the chain workgroup the others
#endif
// MIRROR: the mirrored layout (see StPend above) -- nothing is stored but the vectors: every tile of U_base this launch loads
// is multiplied by its rows' running scales r[row] (buffer pend->rsel) and used.  Each tile is loaded by exactly one
// workgroup (helper s: row blocks 0 .. s - 2, chain s: row block s - 1 and the diagonal block).
template <bool MIRROR = false>
__global__ __launch_bounds__(256) void k_st_fwd_helped(double* __restrict__ M, long long ld, long long n,
                                                       const double* __restrict__ g, double* __restrict__ w,
                                                       double* __restrict__ hpart, double* __restrict__ z,
                                                       double* __restrict__ gg, int* __restrict__ flags,
                                                       int* __restrict__ err, int epoch,
                                                       const DevState* __restrict__ st,
                                                       const StPend* __restrict__ pend = nullptr,
                                                       const double* __restrict__ rbuf = nullptr) {
    if (st->halted) return;
    __shared__ double pr[SB];  // MIRROR: the running row scales on the row block being applied
    const double* rs = nullptr;
    if constexpr (MIRROR) rs = rbuf + (long long)pend->rsel * n;
    __shared__ __attribute__((aligned(16))) double lds[ST_LDS_DOUBLES];  // panel tiles / the parked pieces
    __shared__ double part[4][SPANEL];
    __shared__ double wstrip[SPANEL];
    __shared__ double wblk[SB];
    __shared__ double dlds[SB];
    __shared__ int ok, duty_stop;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long sblk = blockIdx.x >> 1;
    const bool chain = (blockIdx.x & 1) != 0;
    const long long c0 = sblk * SB;
    const long long c = c0 + 2 * lane;
    const long long cl = (c < n) ? c : 0;
    const int piece = lane & 7;
    double* t = lds + wv * (SPANEL * SLDS_PAD);

    auto load_rows = [&](long long J0, double2_t (&u)[2][16]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[h][r] = *reinterpret_cast<const double2_t*>(M + (r0 + r) * ld + cl);
        }
    };
    // MIRROR: request the row scales of row block J0 (LDS; complete at the caller's next barrier)
    auto load_pending = [&](long long J0) __attribute__((always_inline)) {
        if constexpr (MIRROR) {
            if (threadIdx.x < SB) pr[threadIdx.x] = rs[J0 + threadIdx.x];
        }
    };
    // MIRROR: the factor entries as the eager kernels would have found them, fl(U_base[row][c] r[row]) (src/ell_stable.rs:114-117
    // folded into one running product per row)
    auto scale_rows = [&](double2_t (&u)[2][16]) __attribute__((always_inline)) {
        if constexpr (MIRROR) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double x = pr[32 * wv + 16 * h + r];
                    u[h][r].x = u[h][r].x * x;
                    u[h][r].y = u[h][r].y * x;
                }
            }
        }
    };
    // products (u <- u .* w of the row block in wblk); SUMS: also the strip's partial w (leaves with them applied)
    auto apply_rows = [&](double2_t (&u)[2][16], auto sums_c) __attribute__((always_inline)) {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double wj = wblk[32 * wv + 16 * h + r];
                const double v0 = u[h][r].x * wj;
                const double v1 = u[h][r].y * wj;
                p0 += v0;
                p1 += v1;
                u[h][r].x = v0;  // the product replaces the factor entry (src/ell_stable.rs:66)
                u[h][r].y = v1;
            }
        }
        if constexpr (decltype(sums_c)::value) {
            part[wv][2 * lane] = p0;
            part[wv][2 * lane + 1] = p1;
            __syncthreads();
            if (threadIdx.x < SPANEL) {
                const double s4 = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) +
                                  part[3][threadIdx.x];
                wstrip[threadIdx.x] = wstrip[threadIdx.x] - s4;
            }
        }
    };
    // S[col][J0 + ..] <- the products, transposed through LDS (as in k_st_fwd_persist); ends with the tiles drained
    auto write_back = [&](long long J0, double2_t (&u)[2][16]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[(2 * lane) * SLDS_PAD + r] = u[h][r].x;
                t[(2 * lane + 1) * SLDS_PAD + r] = u[h][r].y;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int col_local = 8 * k + (lane >> 3);
                const long long col = c0 + col_local;
                if (col < n) {
                    const double2_t v = *reinterpret_cast<const double2_t*>(&t[col_local * SLDS_PAD + 2 * piece]);
                    *reinterpret_cast<double2_t*>(M + col * ld + r0 + 2 * piece) = v;
                }
            }
            __syncthreads();
        }
    };

    if (threadIdx.x < SPANEL) wstrip[threadIdx.x] = (c0 + threadIdx.x < n) ? g[c0 + threadIdx.x] : 0.0;

    if (!chain) {
        // ------------------------------------------- helper: sums over the row blocks 0 .. sblk - 2, products of the even ones
        const long long last = sblk - 2;
        if (last < 0) return;
        double2_t ua[2][16], ub[2][16];
        // One step: wait for row block kb, request the rows of the step after it into the other buffer (AFTER the wait:
        // loads return in order, a request in front of the poll would delay the poll's answer by its own latency), apply.
        auto step = [&](long long kb, double2_t (&cur)[2][16], double2_t (&nxt)[2][16]) __attribute__((always_inline)) -> bool {
            ST_STAMP(kb, 0);
            load_pending(kb * SB);
            if (threadIdx.x == 0) ok = st_wait_flag(flags + kb, epoch) ? 1 : 0;
            __syncthreads();
            if (!ok) return false;
            ST_STAMP(kb, 1);
            if (threadIdx.x < SB) wblk[threadIdx.x] = st_published_load(w + kb * SB + threadIdx.x);
            __syncthreads();
            ST_STAMP(kb, 2);
            scale_rows(cur);
            apply_rows(cur, std::true_type{});
            if (kb == last && threadIdx.x < SPANEL && c0 + threadIdx.x < n)
                st_publish_store(hpart + c0 + threadIdx.x, wstrip[threadIdx.x]);  // the value is its own flag
            // the rows of the next step: requested AFTER the wait (loads return in order, a request in front of the poll
            // would delay the poll's answer by its own latency) and after the sums (issuing 128 KB of loads takes ~1 us)
            load_rows(((kb + 1 <= last) ? kb + 1 : last) * SB, nxt);  // (past the end: row block `last` again, unused)
            ST_STAMP(kb, 3);
            if (!MIRROR && kb % ST_HELPER_WRITES == 0) write_back(kb * SB, cur);
            else __syncthreads();  // part / wblk are rewritten by the next step
            ST_STAMP(kb, 4);
            return true;
        };
        load_rows(0, ua);
        bool fine = true;
        for (long long kb = 0; fine && kb <= last; kb += 2) {
            fine = step(kb, ua, ub);
            if (fine && kb + 1 <= last) fine = step(kb + 1, ub, ua);
        }
        if (!fine && threadIdx.x == 0) atomicExch(err, 1);
        return;
    }

    // -------------------------------------------------------------------- chain
    double2_t u[2][16];
    // (1) writing duty: the products of the row blocks the helper leaves out, up to sblk - 5, until block sblk - 5 is solved
    long long duty_next = 0;  // the chain workgroup's row blocks below this one have their products in S (uniform)
    if constexpr (!MIRROR) {
        const long long dlast = sblk - ST_DUTY_STOP;
        for (; duty_next <= dlast; ++duty_next) {
            const long long kb = duty_next;
            if (kb % ST_HELPER_WRITES == 0) continue;  // the helper's
            if (threadIdx.x == 0) {
                ok = st_wait_flag(flags + kb, epoch) ? 1 : 0;
                duty_stop = (__hip_atomic_load(flags + dlast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) ? 1 : 0;
            }
            __syncthreads();
            if (!ok) {
                if (threadIdx.x == 0) atomicExch(err, 1);
                return;
            }
            if (threadIdx.x < SB) wblk[threadIdx.x] = st_published_load(w + kb * SB + threadIdx.x);
            const int stop = duty_stop;
            load_rows(kb * SB, u);
            __syncthreads();
            apply_rows(u, std::false_type{});
            write_back(kb * SB, u);
            if (stop) {  // block sblk - 5 is solved: time to get ready for the own solve
                ++duty_next;
                break;
            }
        }
    }
    // (2) the own solve
    Blk3 blk;
    double dreg = 0.0;
    st_prefetch_block(M, ld, n, c0, blk, dreg);
    if (sblk > 0) {
        load_rows((sblk - 1) * SB, u);
        load_pending((sblk - 1) * SB);
    }
    if constexpr (MIRROR) {  // the own diagonal block's factor entries, scaled like every other tile
        st_mul_piece_rows(n, c0, blk.aa, rs, threadIdx.x);
        st_mul_piece_rows(n, c0, blk.ab, rs, threadIdx.x);
        st_mul_piece_rows(n, c0 + SH, blk.bb, rs, threadIdx.x);
    }
    st_fwd_park(blk, dreg, lds, dlds);
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    double ucol[SH];  // this wave's columns of the parked pieces, in registers before the values arrive
    st_fwd_diag_cols(lds, c0 + SH < n, ucol);
    ST_STAMP(sblk, 0);
    if (sblk >= 2 && threadIdx.x < SPANEL && c0 + threadIdx.x < n) {  // the helper's sums over the row blocks 0 .. sblk - 2
        double v = 0.0;
        if (!st_poll_value(hpart + c0 + threadIdx.x, v)) ok = 0;
        wstrip[threadIdx.x] = v;
    }
    ST_STAMP(sblk, 1);
    if (sblk >= 1) {
        // next in the chain: poll the 128 values themselves (all-sentinel when the launch starts)
        if (threadIdx.x < SB) {
            double v = 0.0;
            if (!st_poll_value(w + (sblk - 1) * SB + threadIdx.x, v)) ok = 0;
            wblk[threadIdx.x] = v;
        }
        __syncthreads();
        if (!ok) {
            if (threadIdx.x == 0) atomicExch(err, 1);
            return;
        }
        ST_STAMP(sblk, 2);
        scale_rows(u);
        apply_rows(u, std::true_type{});
    }
    __syncthreads();
    ST_STAMP(sblk, 3);
    st_fwd_diag_block_pre<!MIRROR>(M, ld, n, c0, lds, dlds, wstrip, w, z, gg, flags + sblk, epoch, ucol);
    if constexpr (MIRROR) return;
    // (3) the products not written yet, off the chain (every w they need is published): the remaining odd row blocks and
    // row block sblk - 1 (the helper stops at sblk - 2)
    __syncthreads();  // the parked pieces have been written back: `lds` is free for the transposes
    auto late = [&](long long kb) __attribute__((always_inline)) {
        if (threadIdx.x < SB) wblk[threadIdx.x] = st_published_load(w + kb * SB + threadIdx.x);
        load_rows(kb * SB, u);
        __syncthreads();
        apply_rows(u, std::false_type{});
        write_back(kb * SB, u);
    };
    ST_STAMP(sblk, 5);
    for (long long kb = duty_next; kb < sblk; ++kb)
        if (kb % ST_HELPER_WRITES != 0 || kb == sblk - 1) late(kb);
    ST_STAMP(sblk, 6);
}

// ---------------------------------------------------------------------------------- mid -------
// omega, tsq, EllCalc, kappa; prefix sums t_j; beta2_j; diagonal rescale; q <- z.
// src/ell_stable.rs:78-90,107-113,120-122.  Two launches:
//   k_st_mid   one workgroup of 1024 threads: thread t sums its contiguous chunk of gg (m = ceil(n/1024) elements),
//              the chunk totals are scanned, thread 0 runs the coefficient stage; the exclusive prefix of every
//              chunk and t_0 = omega/mu go to `cpre` (1025 doubles)
//   k_st_post  one thread per element, spread over the chip: t_{j-1} = (cpre[1024] + cpre[chunk]) + gg[lo] + ... in
//              the same left-to-right order as a sequential walk of the chunk, then beta2_j, d_j *= t_{j-1}/t_j,
//              q_j = z_j -- the two divisions per element no longer sit in one workgroup (mid stage 77 -> 25 us at
//              n = 16384); it also re-arms the publish buffer of the persistent backward solve.
constexpr int ST_MID_T = 1024;

__global__ __launch_bounds__(ST_MID_T) void k_st_mid(long long n, const double* __restrict__ gg,
                                                     double* __restrict__ cpre, DevState* __restrict__ st,
                                                     EllCalcDev calc, const CutParams* __restrict__ cp_dev,
                                                     CutParams cp_val, int queue_mode, int* __restrict__ q_status,
                                                     double* __restrict__ q_tsq, StPend* __restrict__ pend = nullptr) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    if (st->halted) {
        if (tid == 0) {
            if (pend) pend->keep_now = 0;
            st->apply = 0;  // a tolerance stop leaves apply = 1 for the update that triggered it only
            if (q_status) {
                *q_status = ST_UNKNOWN;
                *q_tsq = st->tsq;
            }
        }
        return;
    }
    // chunked sums: thread t owns the contiguous chunk [t*m, (t+1)*m)
    const long long m = (n + ST_MID_T - 1) / ST_MID_T;
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
    cpre[tid] = wave_off + (x - s);  // exclusive prefix of this thread's chunk
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
        }
        queue_bookkeeping(st, status, tsq, queue_mode);
        if (q_status) {
            *q_status = status;
            *q_tsq = tsq;
        }
        cpre[ST_MID_T] = t0;
        if (pend) {
            // mirrored layout: the forward solve of this update read the r buffer `rsel`; its backward solve (a successful cut
            // only) must read the same one, while k_st_post writes the scales the NEXT update sees into the other buffer
            pend->have_w = 1;
            pend->keep_now = 1;
            pend->r_fwd_last = pend->rsel;
            if (status == ST_SUCCESS) {
                pend->r_bwd = pend->rsel;
                pend->rsel ^= 1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_st_arm(double* __restrict__ p, long long n) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) p[j] = st_sentinel();
}

// One thread per element j: t_{j-1}, t_j, beta2_j, d_j *= t_{j-1}/t_j (src/ell_stable.rs:111-113,120-121), q_j = z_j
// (:93).  qpub_rearm (persistent backward solve): every entry back to the sentinel before each solve that runs.
__global__ __launch_bounds__(256) void k_st_post(double* __restrict__ M, long long ld, long long n,
                                                 const double* __restrict__ z, const double* __restrict__ gg,
                                                 const double* __restrict__ cpre, double* __restrict__ q,
                                                 double* __restrict__ beta2, double* __restrict__ qpub_rearm,
                                                 double* __restrict__ w_rearm, const DevState* __restrict__ st,
                                                 double* __restrict__ hpart_rearm = nullptr,
                                                 int* __restrict__ fnext_reset = nullptr,
                                                 double* __restrict__ qhpart_rearm = nullptr,
                                                 const StPend* __restrict__ pend = nullptr, double* __restrict__ rbuf = nullptr,
                                                 const double* __restrict__ w_cur = nullptr, double* __restrict__ w_keep = nullptr) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (qhpart_rearm) qhpart_rearm[j] = st_sentinel();  // the backward helpers' hand-over buffer (k_st_bwd_factor_helped)
    if (fnext_reset && j == 0) *fnext_reset = 0;  // the factor-tile queue of k_st_bwd_factor
    if (hpart_rearm) hpart_rearm[j] = st_sentinel();  // the helpers' hand-over buffer (k_st_fwd_helped), like w_rearm
    // the publish buffer of the NEXT persistent forward solve: armed whatever happened to this update (a failed cut and
    // a halted loop still alternate the buffers, see ellstable_issue)
    if (w_rearm) w_rearm[j] = st_sentinel();
    // mirrored layout: the w of the forward solve that has just run (successful cut or not) is what the scratch triangle
    // would hold products of, should somebody ask for it (the solves' own two w buffers are re-armed in turn by every later
    // update, also by the no-op updates of a halted queue)
    if (pend && pend->keep_now) w_keep[j] = w_cur[j];
    if (!st->apply) return;  // failed cut / halted loop: nothing below runs, and neither does the backward solve
    if (qpub_rearm) qpub_rearm[j] = st_sentinel();
    const long long m = (n + ST_MID_T - 1) / ST_MID_T;
    const long long chunk = j / m, lo = chunk * m;
    double told = cpre[ST_MID_T] + cpre[chunk];  // t_{lo-1}
    for (long long i = lo; i < j; ++i) told = told + gg[i];  // :111, left to right inside the chunk
    const double tnew = told + gg[j];       // :111
    const double zj = z[j];
    const double b2 = zj / tnew;            // :112
    beta2[j] = b2;
    M[j * ld + j] = M[j * ld + j] * (told / tnew);  // :113 / :121
    q[j] = zj;                              // :93
    if (pend)  // mirrored layout: the factor update U[j][l] += beta2[j] fl(U[j][l] w[j]), l > j (:114-117) as row j's running scale
        rbuf[(long long)pend->rsel * n + j] = rbuf[(long long)pend->r_bwd * n + j] * (1.0 + b2 * w_cur[j]);
}

// ------------------------------------------------------------------------------ backward ------
// q = z; for j descending: q[t] -= S[j][t] * q[j], t < j   (src/ell_stable.rs:93-98; rows of the scratch
// triangle, bug-compatible).  Same structure as the forward solve, without stores: the three pieces of
// the diagonal block (BB, BA, AA) are prefetched by all 4 waves and parked in LDS; three waves share the chain.
struct Blk3b {
    double2_t bb[8], ba[8], aa[8];
};
__device__ __forceinline__ void st_prefetch_block_bwd(const double* __restrict__ M, long long ld, long long n,
                                                      long long J0, Blk3b& blk, int tid = threadIdx.x) {
    st_load_piece(M, ld, n, J0 + SH, J0 + SH, blk.bb, tid);
    st_load_piece(M, ld, n, J0 + SH, J0, blk.ba, tid);
    st_load_piece(M, ld, n, J0, J0, blk.aa, tid);
}

// Register forms (sv[j] = piece[j][lane] already loaded).  nvalid = rows of the half that exist (64 except in the
// ragged last block).  Full halves take the select off the dependency chain, as the forward chain does: the
// broadcast for step j-1 reads the unselected difference of step j (lane j-1 is an active lane of step j).
__device__ __forceinline__ double st_bwd_chain_regs(const double (&sv)[SH], int nvalid, double qi) {
    const int lane = threadIdx.x & 63;
    if (nvalid == SH) {
        double t = qi;
#pragma unroll
        for (int j = SH - 1; j >= 1 + (SH - ST_EXP_CHAIN_STEPS); --j) {
            const double qj = lane_bcast(t, j);
            const double v = sv[j] * qj;
            t = qi - v;
            qi = (lane < j) ? t : qi;
        }
        return qi;
    }
#pragma unroll
    for (int j = SH - 1; j >= 1; --j) {
        const double qj = lane_bcast(qi, j);
        const double v = sv[j] * qj;
        qi = (lane < j && j < nvalid) ? qi - v : qi;
    }
    return qi;
}
__device__ __forceinline__ double st_bwd_mini_regs(const double (&sv)[SH], const double* __restrict__ qb, int nvalid,
                                                   double qa) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;  // four interleaved partial sums, not a 64-deep chain
#pragma unroll
    for (int j = 0; j < SH; j += 4) {
        const double v0 = sv[j] * qb[j];
        const double v1 = sv[j + 1] * qb[j + 1];
        const double v2 = sv[j + 2] * qb[j + 2];
        const double v3 = sv[j + 3] * qb[j + 3];
        a0 += (j < nvalid) ? v0 : 0.0;
        a1 += (j + 1 < nvalid) ? v1 : 0.0;
        a2 += (j + 2 < nvalid) ? v2 : 0.0;
        a3 += (j + 3 < nvalid) ? v3 : 0.0;
    }
    return qa - ((a0 + a1) + (a2 + a3));
}

// Whole 128-wide diagonal block at J0 (upper half first), called by all 256 threads.  Like the forward block the
// chain is spread over three waves so that the column loads of the later pieces overlap the earlier chain:
//   wave 0: BB columns, chain B | wave 1: BA columns | wave 2: AA columns   -- barrier (q_B in LDS) --
//   wave 0: publishes q_B       | wave 1: mini panel B -> A                 -- barrier (partial q_A in LDS) --
//   wave 2: chain A, publishes q_A.     Same arithmetic in the same order as the one-wave form.
template <bool PUBLISH>
__device__ __forceinline__ void st_bwd_diag_block(long long n, long long J0, const Blk3b& blk,
                                                  double* __restrict__ lds, const double* __restrict__ qpart,
                                                  double* __restrict__ q, double* __restrict__ qpub = nullptr,
                                                  bool parked = false, int tid = threadIdx.x,
                                                  double* __restrict__ qout = nullptr) {
    // tid: index inside the 256-thread group that runs this block; qout (LDS, 128 doubles): the block's final q for a
    // consumer in the same workgroup, complete once every wave of the group has left this function and met a barrier.
    __shared__ double qab[2][SH];  // [0]: final q of half B; [1]: partial q of half A after the mini panel
    double* pBB = lds;
    double* pBA = lds + SH * BLK_PITCH;
    double* pAA = lds + 2 * SH * BLK_PITCH;
    if (!parked) {  // (the persistent solve parks while it waits for the previous block's values)
        st_park_piece(pBB, blk.bb, tid);
        st_park_piece(pBA, blk.ba, tid);
        st_park_piece(pAA, blk.aa, tid);
        __syncthreads();
    }
    const int wave = tid >> 6, lane = tid & 63;
    const bool has_b = J0 + SH < n;
    if (!has_b) {  // ragged last block with one half only: one wave, one chain
        if (wave != 0) return;
        double sv[SH];
#pragma unroll
        for (int j = 0; j < SH; ++j) sv[j] = pAA[j * BLK_PITCH + lane];
        const int nvalid = (n - J0 < SH) ? (int)(n - J0) : SH;
        const double qA = st_bwd_chain_regs(sv, nvalid, qpart[lane]);
        if (qout) qout[lane] = qA;
        if (J0 + lane < n) {
            q[J0 + lane] = qA;
            if (PUBLISH) st_publish_store(qpub + J0 + lane, qA);
        }
        return;
    }
    const int nvalid_b = (n - (J0 + SH) < SH) ? (int)(n - (J0 + SH)) : SH;
    const double* mine = wave == 0 ? pBB : (wave == 1 ? pBA : pAA);
    double sv[SH];
    if (wave <= 2) {
#pragma unroll
        for (int j = 0; j < SH; ++j) sv[j] = mine[j * BLK_PITCH + lane];  // S[..+j][..+lane]
    }
    if (wave == 0) qab[0][lane] = st_bwd_chain_regs(sv, nvalid_b, qpart[lane + SH]);
    __syncthreads();
    if (wave == 0) {
        if (qout) qout[SH + lane] = qab[0][lane];
        if (J0 + SH + lane < n) {
            const double qB = qab[0][lane];
            q[J0 + SH + lane] = qB;
            if (PUBLISH) st_publish_store(qpub + J0 + SH + lane, qB);  // the value is its own flag
        }
    } else if (wave == 1) {
        qab[1][lane] = st_bwd_mini_regs(sv, qab[0], nvalid_b, qpart[lane]);
    }
    __syncthreads();
    if (wave == 2) {
        const double qA = st_bwd_chain_regs(sv, SH, qab[1][lane]);  // half A is complete whenever half B exists
        if (qout) qout[lane] = qA;
        q[J0 + lane] = qA;
        if (PUBLISH) st_publish_store(qpub + J0 + lane, qA);
    }
}

// The same block for a workgroup that has time before the values it waits for arrive (k_st_bwd_factor): columns of the
// parked pieces fetched into registers beforehand, mini panel B -> A shared by waves 1 and 3 -- see st_fwd_diag_block_pre.
// Both halves exist (the caller takes st_bwd_diag_block for a ragged single half).  Always publishes.  Same bits.
__device__ __forceinline__ void st_bwd_diag_cols(const double* __restrict__ lds, double (&sv)[SH]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double* mine = lds + (wave == 0 ? 0 : (wave == 2 ? 2 : 1)) * (SH * BLK_PITCH);  // BB | BA (waves 1, 3) | AA
#pragma unroll
    for (int j = 0; j < SH; ++j) sv[j] = mine[j * BLK_PITCH + lane];
}
template <int O>
__device__ __forceinline__ double st_bwd_mini_half(const double (&sv)[SH], const double* __restrict__ qb, int nvalid) {
    double a0 = 0.0, a1 = 0.0;  // the residues O, O + 1 of j mod 4 of st_bwd_mini_regs' four partial sums
#pragma unroll
    for (int j = O; j < SH; j += 4) {
        const double v0 = sv[j] * qb[j];
        const double v1 = sv[j + 1] * qb[j + 1];
        a0 += (j < nvalid) ? v0 : 0.0;
        a1 += (j + 1 < nvalid) ? v1 : 0.0;
    }
    return a0 + a1;
}
__device__ __forceinline__ void st_bwd_diag_block_pre(long long n, long long J0, const double* __restrict__ qpart,
                                                      double* __restrict__ q, double* __restrict__ qpub,
                                                      const double (&sv)[SH]) {
    __shared__ double qb[SH];       // final q of half B
    __shared__ double mini[2][SH];  // (a0 + a1), (a2 + a3) of the mini panel
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nvalid_b = (n - (J0 + SH) < SH) ? (int)(n - (J0 + SH)) : SH;
    if (wave == 0) qb[lane] = st_bwd_chain_regs(sv, nvalid_b, qpart[lane + SH]);
    __syncthreads();
    if (wave == 0) {
        if (J0 + SH + lane < n) {
            const double qB = qb[lane];
            q[J0 + SH + lane] = qB;
            st_publish_store(qpub + J0 + SH + lane, qB);  // the value is its own flag
        }
    } else if (wave == 1) {
        mini[0][lane] = st_bwd_mini_half<0>(sv, qb, nvalid_b);
    } else if (wave == 3) {
        mini[1][lane] = st_bwd_mini_half<2>(sv, qb, nvalid_b);
    }
    __syncthreads();
    if (wave == 2) {
        const double qA = st_bwd_chain_regs(sv, SH, qpart[lane] - (mini[0][lane] + mini[1][lane]));
        q[J0 + lane] = qA;
        st_publish_store(qpub + J0 + lane, qA);
    }
}

constexpr int ST_LDS_DOUBLES_B = 3 * SH * BLK_PITCH;

__global__ __launch_bounds__(256) void k_st_bwd_last(const double* __restrict__ M, long long ld, long long n,
                                                     long long kb_last, double* __restrict__ q,
                                                     const DevState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ double lds[ST_LDS_DOUBLES_B];
    __shared__ double qpart[SB];
    const long long J0 = kb_last * SB;
    Blk3b blk;
    st_prefetch_block_bwd(M, ld, n, J0, blk);
    if (threadIdx.x < SB) qpart[threadIdx.x] = (J0 + threadIdx.x < n) ? q[J0 + threadIdx.x] : 0.0;
    st_bwd_diag_block<false>(n, J0, blk, lds, qpart, q);
}

// Panel for block kb (rows J0..J0+127 of S, q final there) over columns t < J0, 128 per workgroup (each
// wave 32 rows in two passes), then the workgroup that owns block kb-1 solves it.
__global__ __launch_bounds__(256) void k_st_bwd_step(const double* __restrict__ M, long long ld, long long n,
                                                     long long kb, double* __restrict__ q,
                                                     const DevState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ double lds[ST_LDS_DOUBLES_B];
    __shared__ double part[4][SPANEL];
    __shared__ double qnext[SPANEL];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long J0 = kb * SB;
    const long long c0 = (long long)blockIdx.x * SPANEL;
    const long long c = c0 + 2 * lane;  // columns c, c+1 < J0 (J0 is a multiple of 128, so both or neither)
    const bool owner = (long long)blockIdx.x == kb - 1;  // block kb-1 = columns [J0-128, J0)
    Blk3b blk;
    if (owner) st_prefetch_block_bwd(M, ld, n, J0 - SB, blk);  // independent of q: overlaps the panel work
    double p0 = 0.0, p1 = 0.0;
    if (c < J0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
            double2_t sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long row = r0 + r;
                if (row > n - 1) row = n - 1;
                sv[r] = *reinterpret_cast<const double2_t*>(M + row * ld + c);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = r0 + r;
                const double qj = (row < n) ? q[row] : 0.0;  // wave-uniform
                p0 += sv[r].x * qj;
                p1 += sv[r].y * qj;
            }
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
    if (!owner) return;
    __syncthreads();
    st_bwd_diag_block<false>(n, J0 - SB, blk, lds, qnext, q);
}

// Persistent backward solve: ONE launch; workgroup b owns strip nblk-1-b (dispatch order = dependency
// order).  It applies the row blocks kb = nblk-1 .. strip+1 of the scratch triangle to its 128 columns as
// their q is published, then solves its diagonal block and publishes.
__global__ __launch_bounds__(256) void k_st_bwd_persist(const double* __restrict__ M, long long ld, long long n,
                                                        double* __restrict__ q, double* __restrict__ qpub,
                                                        int* __restrict__ err,
                                                        const DevState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ double lds[ST_LDS_DOUBLES_B];
    __shared__ double part[4][SPANEL];
    __shared__ double qstrip[SPANEL];
    __shared__ double qblk[SB];
    __shared__ int ok;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long nblk = gridDim.x;
    const long long sblk = nblk - 1 - blockIdx.x;
    const long long c0 = sblk * SB;
    const long long c = c0 + 2 * lane;  // columns c, c+1 < c0 + 128 <= J0 of every row block applied here

    Blk3b blk;
    if (sblk == nblk - 1) st_prefetch_block_bwd(M, ld, n, c0, blk);
    if (threadIdx.x < SPANEL) qstrip[threadIdx.x] = (c0 + threadIdx.x < n) ? q[c0 + threadIdx.x] : 0.0;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();

    for (long long kb = nblk - 1; kb > sblk; --kb) {
        const long long J0 = kb * SB;
        double2_t sv[2][16];  // both passes' rows requested before the flag wait
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long row = r0 + r;
                if (row > n - 1) row = n - 1;
                sv[h][r] = *reinterpret_cast<const double2_t*>(M + row * ld + c);
            }
        }
        if (kb == sblk + 1) {  // own diagonal block: fetched and parked in LDS while the values it waits for are computed
            st_prefetch_block_bwd(M, ld, n, c0, blk);
            st_park_piece(lds, blk.bb, threadIdx.x);
            st_park_piece(lds + SH * BLK_PITCH, blk.ba, threadIdx.x);
            st_park_piece(lds + 2 * SH * BLK_PITCH, blk.aa, threadIdx.x);
        }
        if (threadIdx.x < SB) {
            double v = 0.0;
            if (J0 + threadIdx.x < n && !st_poll_value(qpub + J0 + threadIdx.x, v)) ok = 0;
            qblk[threadIdx.x] = v;
        }
        __syncthreads();
        if (!ok) {
            if (threadIdx.x == 0) atomicExch(err, 2);
            return;
        }
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double qj = qblk[32 * wv + 16 * h + r];  // 0 for rows beyond n
                p0 += sv[h][r].x * qj;
                p1 += sv[h][r].y * qj;
            }
        }
        part[wv][2 * lane] = p0;
        part[wv][2 * lane + 1] = p1;
        __syncthreads();
        if (threadIdx.x < SPANEL) {
            const double s4 = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) +
                              part[3][threadIdx.x];
            qstrip[threadIdx.x] = qstrip[threadIdx.x] - s4;
        }
        __syncthreads();
    }
    __syncthreads();
    st_bwd_diag_block<true>(n, c0, blk, lds, qstrip, q, qpub, sblk < nblk - 1);
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

// Mirrored layout: a piece of L times a vector of per-COLUMN constants (the row scales r[t], then w[t]: together the products
// the reference parked, S[j][t] = fl(fl(L[j][t] r[t]) w[t])).
__device__ __forceinline__ void st_mul_piece_cols(long long n, long long C0, double2_t (&v)[8], const double* __restrict__ w,
                                                  int tid) {
    const long long c = C0 + 2 * (tid & 31);
    const double w0 = (c < n) ? w[c] : 0.0, w1 = (c + 1 < n) ? w[c + 1] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k].x = v[k].x * w0;
        v[k].y = v[k].y * w1;
    }
}

// Between the two layouts: lower[i][j] = fl(fl(U[j][i] r[j]) w[j]) for i > j (r, w != NULL: the scratch triangle exactly as a
// forward solve on the scaled factor with that w stores it, src/ell_stable.rs:66) or = U[j][i] (r = w = NULL: the mirrored
// copy of the factor).  64 x 64 tiles transposed through LDS; grid (tiles, tiles), tiles left of the diagonal leave at once.
__device__ __forceinline__ void st_transpose_lower_tile(double* __restrict__ M, long long ld, long long n,
                                                        const double* __restrict__ r, const double* __restrict__ w) {
    const long long tj = blockIdx.y, ti = blockIdx.x;  // source tile rows j (tj), columns i (ti); destination rows i, columns j
    if (ti < tj) return;
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows per pass
    const long long j0 = tj * 64, i0 = ti * 64;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long long j = j0 + ty + 4 * k, i = i0 + tx;
        double v = 0.0;
        if (j < n && i < n && i > j) {
            v = M[j * ld + i];
            if (r) v = v * r[j];
            if (w) v = v * w[j];
        }
        tile[ty + 4 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long long i = i0 + ty + 4 * k, j = j0 + tx;
        if (i < n && j < n && i > j) M[i * ld + j] = tile[tx][ty + 4 * k];
    }
}
// reference layout -> mirrored layout: the factor copied below the diagonal, every row scale 1.  Not on a halted queue (every
// solve behind it is a no-op as well, and the scratch triangle of the failing cut has to survive until its results are
// read); whether it ran is device state (pend->mirrored, set by k_st_mirror_mark behind it).
__global__ __launch_bounds__(256) void k_st_mirror_enter(double* __restrict__ M, long long ld, long long n,
                                                         const DevState* __restrict__ st, double* __restrict__ rbuf) {
    if (st->halted) return;
    if (blockIdx.x == blockIdx.y && threadIdx.x < 64) {
        const long long j = (long long)blockIdx.y * 64 + threadIdx.x;
        if (j < n) rbuf[j] = 1.0, rbuf[n + j] = 1.0;
    }
    st_transpose_lower_tile(M, ld, n, nullptr, nullptr);
}
__global__ void k_st_mirror_mark(StPend* __restrict__ pend, const DevState* __restrict__ st) {
    StPend p;
    p.rsel = 0, p.r_bwd = 0, p.r_fwd_last = 0, p.mirrored = st->halted ? 0 : 1, p.have_w = 0, p.keep_now = 0, p.pad_[0] = p.pad_[1] = 0;
    *pend = p;
}
// mirrored layout -> reference layout, for an observer of the buffer: first the scratch triangle of the last forward solve
// (from U_base, the scales that solve used and its w), then U itself as the eager kernels would hold it, fl(U_base[j][l] r[j])
// with the current scales (k_st_unscale_upper), then k_st_mirror_clear.
__global__ __launch_bounds__(256) void k_st_mirror_leave(double* __restrict__ M, long long ld, long long n,
                                                         const StPend* __restrict__ pend, const double* __restrict__ rbuf,
                                                         const double* __restrict__ w_keep) {
    if (!pend->mirrored || !pend->have_w) return;
    st_transpose_lower_tile(M, ld, n, rbuf + (long long)pend->r_fwd_last * n, w_keep);
}
__global__ __launch_bounds__(256) void k_st_unscale_upper(double* __restrict__ M, long long ld, long long n,
                                                          const StPend* __restrict__ pend, const double* __restrict__ rbuf) {
    if (!pend->mirrored) return;
    const long long j = blockIdx.y;
    const double x = rbuf[(long long)pend->rsel * n + j];
    for (long long l = j + 1 + (long long)blockIdx.x * 256 + threadIdx.x; l < n; l += (long long)gridDim.x * 256)
        M[j * ld + l] = M[j * ld + l] * x;
}
__global__ void k_st_mirror_clear(StPend* __restrict__ pend) {
    StPend p;
    p.rsel = 0, p.r_bwd = 0, p.r_fwd_last = 0, p.mirrored = 0, p.have_w = 0, p.keep_now = 0, p.pad_[0] = p.pad_[1] = 0;
    *pend = p;
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

// The same update WITHOUT reading the scratch triangle.  S[l][j] is, bit for bit, the product fl(U[j][l] * w[j]) the
// forward solve of THIS update stored (src/ell_stable.rs:66; every path above writes exactly that product, and nothing
// touches U or S between the solve and here), so
//     U[j][l] <- U[j][l] + beta2[j] * fl(U[j][l] * w[j]),   l > j
// is the reference's `+= beta2 * S[l][j]` (:114-117) with the factor of the product re-read from the element itself:
// a row-wise read-modify-write of the strict upper triangle -- 8 n^2 bytes instead of 12 n^2, no transposes.
// Tiles of FROW_H rows x SEG columns (one workgroup each; 2-D grid, tiles left of the diagonal leave at once); a thread
// holds SEG / 512 column pairs of RW rows at a time.
constexpr int FROW_H = 64;
// PIPE (the workers of k_st_bwd_factor: ONE workgroup per CU, so the loads in flight have to come from the thread
// itself): full tiles with two register buffers of RW rows -- the next RW rows are requested before the current ones
// are updated and stored (RW = 8, SEG = 2048: 2 x 128 KB per workgroup in flight).
template <int SEG, int RW, bool PIPE = false, int H = FROW_H>
__device__ __forceinline__ void st_factor_tile(double* __restrict__ M, long long ld, long long n,
                                               const double* __restrict__ beta2, const double* __restrict__ w,
                                               long long I, long long J) {
    constexpr int NCH = SEG / 512;
    const long long r0 = I * H, c0 = J * SEG;
    if (r0 >= n || c0 >= n || c0 + SEG - 1 <= r0) return;  // no column of the segment right of the strip's first row
    const long long rlast = (r0 + H - 1 < n - 1) ? r0 + H - 1 : n - 1;
    // every element of the tile is right of the diagonal and every pair inside the matrix: no masks
    const bool full = c0 > rlast && c0 + SEG <= n;
    long long col[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) col[k] = c0 + 512 * k + 2 * (long long)threadIdx.x;
    if constexpr (PIPE) {
        if (full) {
            double2_t va[RW][NCH], vb[RW][NCH];
            auto fetch = [&](long long j0, double2_t (&v)[RW][NCH]) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const long long j = (j0 + r <= rlast) ? j0 + r : rlast;  // (past the strip: the last row again, not stored)
#pragma unroll
                    for (int k = 0; k < NCH; ++k) v[r][k] = ld_stream<true, double2_t>(M + j * ld + col[k]);
                }
            };
            auto update = [&](long long j0, const double2_t (&v)[RW][NCH]) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    if (j0 + r > rlast) break;
                    const double bj = beta2[j0 + r], wv = w[j0 + r];
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        double2_t o;
                        o.x = v[r][k].x + bj * (v[r][k].x * wv);
                        o.y = v[r][k].y + bj * (v[r][k].y * wv);
                        *reinterpret_cast<double2_t*>(M + (j0 + r) * ld + col[k]) = o;
                    }
                }
            };
            fetch(r0, va);
            for (long long j0 = r0;; j0 += 2 * RW) {
                const bool more1 = j0 + RW <= rlast;
                if (more1) fetch(j0 + RW, vb);
                update(j0, va);
                if (!more1) break;
                const bool more2 = j0 + 2 * RW <= rlast;
                if (more2) fetch(j0 + 2 * RW, va);
                update(j0 + RW, vb);
                if (!more2) break;
            }
            return;
        }
    }
    for (long long j0 = r0; j0 <= rlast; j0 += RW) {
        double2_t v[RW][NCH];
        double b[RW], wj[RW];
        if (full) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const long long j = (j0 + r <= rlast) ? j0 + r : rlast;  // (past the strip: the last row again, not stored)
                b[r] = beta2[j];
                wj[r] = w[j];
#pragma unroll
                for (int k = 0; k < NCH; ++k) v[r][k] = ld_stream<true, double2_t>(M + j * ld + col[k]);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                if (j0 + r > rlast) break;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    double2_t o;
                    o.x = v[r][k].x + b[r] * (v[r][k].x * wj[r]);
                    o.y = v[r][k].y + b[r] * (v[r][k].y * wj[r]);
                    *reinterpret_cast<double2_t*>(M + (j0 + r) * ld + col[k]) = o;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const long long j = j0 + r;
                if (j > rlast) break;
                const double bj = beta2[j], wv = w[j];
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const long long l = col[k];
                    if (l + 1 <= j || l >= n) continue;  // both left of / on the diagonal, or outside the matrix
                    double* p = M + j * ld + l;
                    if (l > j && l + 1 < n) {
                        double2_t uv = *reinterpret_cast<double2_t*>(p);
                        uv.x = uv.x + bj * (uv.x * wv);
                        uv.y = uv.y + bj * (uv.y * wv);
                        *reinterpret_cast<double2_t*>(p) = uv;
                    } else if (l > j) {        // l == n - 1: the last column alone
                        if (l < n) p[0] = p[0] + bj * (p[0] * wv);
                    } else if (l + 1 < n) {    // l == j: only the second element of the pair is right of the diagonal
                        p[1] = p[1] + bj * (p[1] * wv);
                    }
                }
            }
        }
    }
}
template <int SEG, int RW>
__global__ __launch_bounds__(256) void k_st_factor_rows(double* __restrict__ M, long long ld, long long n,
                                                        const double* __restrict__ beta2,
                                                        const double* __restrict__ w,
                                                        const DevState* __restrict__ st) {
    if (!st->apply) return;
    st_factor_tile<SEG, RW>(M, ld, n, beta2, w, (long long)blockIdx.x, (long long)blockIdx.y);
}

// The same launch with a HELPER workgroup per block (as in k_st_fwd_helped) and no dedicated factor workers: 2 * nblk
// workgroups, one per CU.  helper s: the strip's partial sums over the row blocks nblk-1 .. s+2, handed to chain s
// (qhpart, all-sentinel at launch); chain s: own block parked, columns and the rows of row block s+1 in registers before
// block s+1's values come out.  Factor tiles are SMALL here (FQ_H = 16 rows x SEG columns, ~10 us of work) and every
// workgroup pulls them whenever it has nothing else to do: a chain workgroup BEFORE its turn (it stops when block
// s + FQ_STOP is solved -- one tile and the 6 us of fetching and parking fit in the remaining steps) and after it, a
// helper after its last row block.  So nearly all of the chain workgroups' CU time goes to the factor update instead
// of waiting, and the chain itself has nothing in front of it when its turn comes.
constexpr int FQ_H = 16;
constexpr long long FQ_STOP = 6;
// MIRROR (the mirrored layout, see StPend): the addresses the scratch triangle used to occupy hold L[j][t] = U_base[t][j]; every
// element loaded becomes the product the reference parked there, fl(fl(L[j][t] r[t]) w[t]) -- the lane's own constants: the
// running scale of row t of the factor (the buffer this update's forward solve read, pend->r_bwd) and this update's w
// (src/ell_stable.rs:66) -- and enters the same sums in the same order.  Nothing is stored, there are no factor tiles.
template <int SEG, int RW, bool MIRROR = false>
__global__ __launch_bounds__(256) void k_st_bwd_factor_helped(double* __restrict__ M, long long ld, long long n,
                                                              double* __restrict__ q, double* __restrict__ qpub,
                                                              double* __restrict__ qhpart, int* __restrict__ err,
                                                              const DevState* __restrict__ st, long long nblk,
                                                              const double* __restrict__ beta2,
                                                              const double* __restrict__ w,
                                                              const int* __restrict__ ftiles, int nftiles,
                                                              int* __restrict__ fnext, long long fq_stop,
                                                              const StPend* __restrict__ pend = nullptr,
                                                              const double* __restrict__ rbuf = nullptr) {
    if (!st->apply) return;
    __shared__ double lds[ST_LDS_DOUBLES_B];
    __shared__ double part[4][SPANEL];
    __shared__ double qstrip[SPANEL];
    __shared__ double qblk[SB];
    __shared__ int ok, ftile;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long sblk = nblk - 1 - (long long)(blockIdx.x >> 1);
    const bool chain = (blockIdx.x & 1) != 0;
    const long long c0 = sblk * SB;
    const long long c = c0 + 2 * lane;  // columns c, c+1 < c0 + 128 <= J0 of every row block applied here
    const bool has_b = c0 + SH < n;
    // MIRROR: this lane's two columns t = c, c + 1: the running scale r[t] and w[t] of the update in flight
    double wc0 = 0.0, wc1 = 0.0, rc0 = 0.0, rc1 = 0.0;
    const double* rb = nullptr;
    if constexpr (MIRROR) {
        rb = rbuf + (long long)pend->r_bwd * n;
        if (c < n) wc0 = w[c], rc0 = rb[c];
        if (c + 1 < n) wc1 = w[c + 1], rc1 = rb[c + 1];
    }

    // factor tiles until the queue is empty or (deadline >= 0) block `deadline` has been solved
    auto work = [&](long long deadline) __attribute__((always_inline)) {
        if constexpr (MIRROR) return;
        for (;;) {
            if (threadIdx.x == 0) {
                int stop = 0;
                if (deadline >= 0) {
                    const double v = __hip_atomic_load(qpub + deadline * SB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stop = (unsigned long long)__double_as_longlong(v) != ST_SENTINEL_BITS;
                }
                ftile = stop ? nftiles : atomicAdd(fnext, 1);
            }
            __syncthreads();
            const int k = ftile;
            __syncthreads();
            if (k >= nftiles) break;
            const int tl = ftiles[k];
            st_factor_tile<SEG, RW, true, FQ_H>(M, ld, n, beta2, w, (long long)(tl >> 4), (long long)(tl & 0xf));
        }
    };
    auto load_rows = [&](long long J0, double2_t (&sv)[2][16]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long row = r0 + r;
                if (row > n - 1) row = n - 1;
                sv[h][r] = *reinterpret_cast<const double2_t*>(M + row * ld + c);
            }
        }
    };
    // MIRROR: the rows just loaded hold L[j][t] = U_base[t][j]: the parked product fl(fl(L r[t]) w[t])
    auto mirror_rows = [&](double2_t (&sv)[2][16]) __attribute__((always_inline)) {
        if constexpr (MIRROR) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sv[h][r].x = (sv[h][r].x * rc0) * wc0;
                    sv[h][r].y = (sv[h][r].y * rc1) * wc1;
                }
            }
        }
    };
    // row block J0's q (in qblk) applied to the strip's partial sums
    auto apply_rows = [&](const double2_t (&sv)[2][16]) __attribute__((always_inline)) {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double qj = qblk[32 * wv + 16 * h + r];  // 0 for rows beyond n
                p0 += sv[h][r].x * qj;
                p1 += sv[h][r].y * qj;
            }
        }
        part[wv][2 * lane] = p0;
        part[wv][2 * lane + 1] = p1;
        __syncthreads();
        if (threadIdx.x < SPANEL) {
            const double s4 = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) +
                              part[3][threadIdx.x];
            qstrip[threadIdx.x] = qstrip[threadIdx.x] - s4;
        }
        __syncthreads();
    };
    // the 128 values of row block kb, polled (data-as-flag); false on time-out
    auto wait_block = [&](long long kb) __attribute__((always_inline)) -> bool {
        if (threadIdx.x < SB) {
            double v = 0.0;
            if (kb * SB + threadIdx.x < n && !st_poll_value(qpub + kb * SB + threadIdx.x, v)) ok = 0;
            qblk[threadIdx.x] = v;
        }
        __syncthreads();
        return ok != 0;
    };

    if (threadIdx.x < SPANEL) qstrip[threadIdx.x] = (c0 + threadIdx.x < n) ? q[c0 + threadIdx.x] : 0.0;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();

    if (!chain) {
        // -------------------------------------------------------- helper: row blocks nblk-1 .. sblk+2, then factor tiles
        double2_t sv[2][16];
        for (long long kb = nblk - 1; kb >= sblk + 2; --kb) {
            // Matrix on-die (n < 8192, SEG = 512): far from its hand-over the helper does not wait for a row block, it
            // pulls factor tiles until the block is there (a tile in progress makes it late by a few us; it catches up at
            // 1.6 us per step against the chain's 4-6): n = 4096 147.6 -> 139.5 us.  From HBM (n = 16384) the extra
            // streaming slows the chain's hand-offs more than it helps: 783 -> 847 us (chain alone 515).
            if (SEG == 512 && kb > sblk + 2 + fq_stop) work(kb);
            load_rows(kb * SB, sv);
            if (!wait_block(kb)) {
                if (threadIdx.x == 0) atomicExch(err, 2);
                return;
            }
            mirror_rows(sv);
            apply_rows(sv);
        }
        if (sblk + 2 <= nblk - 1 && threadIdx.x < SPANEL && c0 + threadIdx.x < n)
            st_publish_store(qhpart + c0 + threadIdx.x, qstrip[threadIdx.x]);  // the value is its own flag
        work(-1);
        return;
    }

    // ------------------------------------------------------------ chain: factor tiles until block sblk + FQ_STOP is solved
    if (sblk + fq_stop <= nblk - 1) work(sblk + fq_stop);
    Blk3b blk;
    double svc[SH];  // this wave's columns of the parked diagonal block
    double2_t sv[2][16];
    st_prefetch_block_bwd(M, ld, n, c0, blk);
    if (sblk + 1 <= nblk - 1) load_rows((sblk + 1) * SB, sv);
    if constexpr (MIRROR) {  // the own diagonal block: scaled, then the parked products
        st_mul_piece_cols(n, c0 + SH, blk.bb, rb, threadIdx.x);
        st_mul_piece_cols(n, c0, blk.ba, rb, threadIdx.x);
        st_mul_piece_cols(n, c0, blk.aa, rb, threadIdx.x);
        st_mul_piece_cols(n, c0 + SH, blk.bb, w, threadIdx.x);
        st_mul_piece_cols(n, c0, blk.ba, w, threadIdx.x);
        st_mul_piece_cols(n, c0, blk.aa, w, threadIdx.x);
    }
    st_park_piece(lds, blk.bb, threadIdx.x);
    st_park_piece(lds + SH * BLK_PITCH, blk.ba, threadIdx.x);
    st_park_piece(lds + 2 * SH * BLK_PITCH, blk.aa, threadIdx.x);
    __syncthreads();
    if (has_b) st_bwd_diag_cols(lds, svc);
    if (sblk + 2 <= nblk - 1 && threadIdx.x < SPANEL && c0 + threadIdx.x < n) {  // the helper's sums
        double v = 0.0;
        if (!st_poll_value(qhpart + c0 + threadIdx.x, v)) ok = 0;
        qstrip[threadIdx.x] = v;
    }
    if (sblk + 1 <= nblk - 1) {
        if (!wait_block(sblk + 1)) {
            if (threadIdx.x == 0) atomicExch(err, 2);
            return;
        }
        mirror_rows(sv);
        apply_rows(sv);
    }
    __syncthreads();
    if (has_b) st_bwd_diag_block_pre(n, c0, qstrip, q, qpub, svc);
    else st_bwd_diag_block<true>(n, c0, blk, lds, qstrip, q, qpub, true);  // ragged last block, one half
    work(-1);  // this block is solved and handed over: the CU goes back to the factor tiles
}

}  // namespace ellhip
