// resident_kernels.hpp -- Ell::update_core (src/ell.rs:97-137) for a whole QUEUE of cuts in ONE persistent launch with the
// matrix parked ON-CHIP: the loop `for cut in queue { update }` of src/cutting_plane.rs:299-311 around the update, as
// SURVEY section 7's k_ell_fused asks, for the sizes where the lower triangle fits the chip's register files.
//
// Why: at n = 4096 the lower triangle of Q is 64 MiB, the 256 CUs hold 128 MiB of vector registers, and the streamed
// schedule still moves the matrix from the Infinity Cache every update: 39 us per update for 23 + 11 us of kernels
// (profiles/r02/bench_n4096_kernel_stats.csv).  Here the lower triangle is cut into SUPER-TILES of R x R tiles of 64 x 64
// doubles (R = 1, 2 or 3, the smallest that leaves no more super-tiles than CUs: n <= 1408 / 2816 / 4224), one
// workgroup per super-tile (one per CU, 512 threads = two waves per SIMD with 256 registers each), and every thread
// keeps its 4 x 2 block (rows 4 bi .., columns 2 bj ..) of each of the R x R tiles in registers for the whole launch
// (R = 3: 72 doubles = 144 registers; a first version with 256 threads and 4 x 4 blocks needed 288 + temporaries, i.e.
// the accumulation half of the register file plus scratch, and spilled 392 bytes per lane).  A diagonal super-tile
// holds its whole symmetric block (the tiles above the diagonal are mirrored copies and go through the same
// arithmetic).  One update:
//
//   1. GEMV on the resident tiles: per thread the 4 x 2 blocks feed R x 4 row sums (accumulated across the tiles of a
//      tile row) and, off the diagonal, R x 2 column sums (the mirrored half); the cross-lane sums are transposing
//      reductions (4 DPP exchanges + 1 shuffle over the 32 lanes of a row of blocks, 1 shuffle over the 2 row blocks of a
//      wave, LDS across the 8 waves).  The workgroup publishes its 2 R x 64 partial sums and its share of omega = g'Qg
//      to L2 (write-through), double-buffered by cut parity.                                     src/arr.rs:426-451
//   -- ONE grid barrier --
//   2. every workgroup forms, for ITS R row blocks and R column blocks, y = Q g from the S partial vectors each block
//      has (row sums of the super-tiles left of the diagonal in that block row, column sums of those below it in that
//      block column; index order: the same bits in every workgroup that needs the block), adds the omega shares in one
//      fixed tree and runs EllCalc -- redundantly, identical bits everywhere.                  src/ell.rs:105-106
//   3. the diagonal workgroups update their entries of xc, and every tile gets its rank-1 in registers with the
//      reference's roundings: x <- x - (ratio * gt[hi]) * gt[lo], hi = max(row, col), separate multiply and subtract.
//                                                                                              src/ell.rs:111-130
//
// A failing cut (status != Success) leaves Q, xc and kappa untouched and ends the loop on every workgroup at the same
// cut (they all computed the same status): src/cutting_plane.rs:308.  Results: the rank-1 follows the reference's
// roundings for the gt it is given; gt itself is summed in this kernel's own fixed association, so states agree with the
// other schedules to ~1e-15 and with the CPU path within the 1e-10 contract, and are bit-reproducible run to run (no
// atomics in the data path, static assignment).
//
// The grid is one workgroup per CU and must be resident as a whole (a workgroup that waits holds its CU): the host
// launches it COOPERATIVELY (hipLaunchCooperativeKernel refuses a grid the device cannot hold at once, and the runtime
// does not run two cooperative grids side by side -- two handles on one card, BSearchAdaptor's clone per probe
// src/cutting_plane.rs:409-418, cannot starve each other of CUs); every wait is bounded all the same.
// Failure contract (a bounded wait gave up): the batch is ABANDONED AS A WHOLE.  The tiles are written back only behind a
// commit word on which every workgroup either arrives or aborts, never both (rs_commit): if any workgroup aborted, none
// writes, Q in HBM is still the pre-batch matrix, DevState.solve_err = RS_WAIT_ERR, and the host restores xc / DevState /
// the batch's queue results from its snapshots and reruns the batch on the streamed schedule (resident_run).
#pragma once

#include <hip/hip_runtime.h>

#include "ell_kernels.hpp"

namespace ellhip {

constexpr int RS_TS = 64;     // tile size
constexpr int RS_RMAX = 3;    // tiles per super-tile edge: R x R x 8 doubles per thread = 144 VGPRs at R = 3
constexpr int RS_SMAX = 22;   // super-tile rows: S (S + 1) / 2 = 253 workgroups <= 256 CUs
constexpr int RS_THREADS = 512;
constexpr int RS_WAIT_ERR = 9;
// tools/experiments/resident_bench.hip -DRS_TIMELINE: phase stamps of workgroup 0 (s_memrealtime, 100 MHz)
#ifdef RS_TIMELINE
#define RS_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0 && A.stamps) A.stamps[(cut - A.first) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RS_STAMP(slot)
#endif

struct ResidentArgs {
    double* Q;                 // n x ld row-major; the lower triangle is read at entry and written at exit
    long long ld, n;
    int T;                     // tile rows = ceil(n / 64)
    int R, S;                  // super-tile edge in tiles, super-tile rows = ceil(T / R); grid = S (S + 1) / 2
    const double* qgrads;      // queue: gradients [k][n]
    const CutParams* qparams;  // queue: cut scalars
    int* qstatus;              // queue: per-cut status out
    double* qtsq;              // queue: per-cut tsq out
    long long first, count;    // cuts [first, first + count)
    double* xc;                // n
    DevState* st;
    double* part;              // [2][grid][2 R 64] partial sums (row sums | column sums), by cut parity
    double* omega_part;        // [2][grid] shares of omega, by cut parity
    unsigned* ctr;             // barrier words (RS_BAR_WORDS unsigned), zero at launch
    unsigned long long* stamps;  // RS_TIMELINE builds only
    EllCalcDev calc;
    long long fault_at;        // test hook (ELLHIP_OPT_RESIDENT_FAULT): workgroup 1 % grid abandons the batch at this cut of
                               // it as if its wait had timed out; < 0: never
};

__device__ __forceinline__ int rs_super_index(int SI, int SJ) { return SI * (SI + 1) / 2 + SJ; }

// DPP move of a double (two 32-bit halves); CTRL: quad_perm 0x00-0xFF, row_ror:n 0x120 + n
template <int CTRL>
__device__ __forceinline__ double rs_dpp(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// v[0..3] summed over the 32 lanes that share lane bit 5 (lane bits 0-4): on return every lane holds the total of
// v[2 * bit0 + bit1] (bit0, bit1 of its own lane id).  4 DPP exchanges + 1 shuffle.
__device__ __forceinline__ double rs_reduce4_over32(const double (&v)[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    double k2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b0 ? v[2 + i] : v[i], send = b0 ? v[i] : v[2 + i];
        k2[i] = keep + rs_dpp<0xB1>(send);  // quad_perm [1,0,3,2]: lane ^ 1
    }
    double k = (b1 ? k2[1] : k2[0]) + rs_dpp<0x4E>(b1 ? k2[0] : k2[1]);  // quad_perm [2,3,0,1]: lane ^ 2
    k += rs_dpp<0x124>(k);  // row_ror:4
    k += rs_dpp<0x128>(k);  // row_ror:8
    k += __shfl_xor(k, 16, 64);
    return k;
}

// v[0..1] summed over the two halves of a wave (lane bit 5): every lane ends with the total of v[bit5].  1 exchange.
__device__ __forceinline__ double rs_reduce2_halves(const double (&v)[2], int lane) {
    const bool b5 = lane & 32;
    return (b5 ? v[1] : v[0]) + __shfl_xor(b5 ? v[0] : v[1], 32, 64);
}

// Grid barrier number `k` (1, 2, ...) of a launch.  Everything that crosses workgroups in this kernel is stored
// write-through (ho_store: agent-scope relaxed atomic stores, sc1) and read with agent-scope loads, so the barrier itself
// needs no fence -- a release / acquire pair at agent scope writes back and invalidates the XCD's whole L2 (7.4 us per
// barrier).  256 atomic adds to ONE address serialise at the memory side (3.9 us per barrier for a single counter), so the
// arrival is a tree on separate 128-byte lines: 16 group counters, the group's last arrival adds to the root, the
// root's last arrival stores the barrier number to 16 release words, and a workgroup polls only its group's release
// word with one lane: 1.8 us per barrier (tools/experiments/grid_barrier_bench.hip, profiles/r03/grid_barrier.txt).
// `bar`: RS_BAR_WORDS unsigned words, zero at launch: root at [0], group counter g at [32 (1 + g)], release word g at
// [32 (17 + g)].  Waits are bounded.
constexpr int RS_BAR_GROUPS = 16;
constexpr int RS_BAR_COMMIT = 32 * (1 + 2 * RS_BAR_GROUPS);  // the commit word (its own 128-byte line)
constexpr int RS_BAR_WORDS = RS_BAR_COMMIT + 32;
constexpr unsigned RS_ABORT = 0x80000000u;
constexpr int RS_SPINS = 1 << 22;

// The commit word: arrivals in the low bits (atomic adds), RS_ABORT on top (a compare-and-swap that succeeds only while
// fewer than G workgroups have arrived).  A workgroup either arrives or aborts, so the word ends in exactly one of the states
// {G arrivals, bit clear} | {bit set}, readers test the bit first, and every workgroup reads the same verdict off it.
__device__ __forceinline__ bool rs_aborted(const unsigned* bar) {
    return (__hip_atomic_load(bar + RS_BAR_COMMIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & RS_ABORT) != 0;
}
// returns true when the batch is (now) aborted, false when all G workgroups had already arrived (the batch commits)
__device__ __forceinline__ bool rs_abort(unsigned* bar, unsigned G) {
    unsigned* cw = bar + RS_BAR_COMMIT;
    unsigned old = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (old & RS_ABORT) return true;
        if ((old & ~RS_ABORT) == G) return false;
        if (__hip_atomic_compare_exchange_strong(cw, &old, old | RS_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return true;
    }
}

__device__ __forceinline__ bool rs_grid_barrier(unsigned* bar, unsigned k, int* sh_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing waves drain their write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const int G = (int)gridDim.x, g = (int)blockIdx.x % RS_BAR_GROUPS;
        const int ngrp = G < RS_BAR_GROUPS ? G : RS_BAR_GROUPS;
        const unsigned members = (unsigned)((G - g + RS_BAR_GROUPS - 1) / RS_BAR_GROUPS);
        const unsigned old = __hip_atomic_fetch_add(bar + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        if (old + 1 == members * k) {
            const unsigned r = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r + 1 == (unsigned)ngrp * k) {
                for (int j = 0; j < ngrp; ++j)
                    __hip_atomic_store(bar + 32 * (1 + RS_BAR_GROUPS + j), k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 1;
            }
        }
        const unsigned* rel = bar + 32 * (1 + RS_BAR_GROUPS + g);
        for (int spin = 0; spin < RS_SPINS && !good; ++spin) {
            if ((int)(__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - k) >= 0) good = 1;
            else if ((spin & 63) == 63 && rs_aborted(bar)) break;  // somebody has abandoned the batch: so does this workgroup
            else __builtin_amdgcn_s_sleep(1);
        }
        *sh_ok = good;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// End of the batch: true = every workgroup got here (write the tiles back), false = the batch was abandoned (write
// nothing).  `failed`: this workgroup gave up inside the loop.
__device__ __forceinline__ bool rs_commit(unsigned* bar, bool failed, int* sh_ok) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned G = gridDim.x;
        unsigned* cw = bar + RS_BAR_COMMIT;
        int ok = -1;
        if (failed) {
            ok = rs_abort(bar, G) ? 0 : 1;  // (it has not arrived, so G arrivals are impossible: always an abort)
        } else {
            // arrive: ONE atomic add (253 compare-and-swap loops on one word retried each other for 0.55 ms per batch).  An
            // arrival behind an abort still counts up, which is harmless: every reader looks at RS_ABORT first, and the bit
            // is only ever set while the count is below G (rs_abort), so "G arrivals without the bit" still means that
            // nobody gave up.
            const unsigned old = __hip_atomic_fetch_add(cw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old & RS_ABORT) ok = 0;
            for (int spin = 0; spin < RS_SPINS && ok < 0; ++spin) {
                const unsigned v = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & RS_ABORT) ok = 0;
                else if (v == G) ok = 1;
                else __builtin_amdgcn_s_sleep(1);
            }
            if (ok < 0) ok = rs_abort(bar, G) ? 0 : 1;
        }
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// Before a batch: snapshot of the centre and the scalar state (restored by the host when the batch is abandoned) and the
// barrier words zeroed -- one launch (three copy-engine operations in front of the batch cost ~0.4 ms of cross-engine
// hand-overs: 78 000 instead of 106 000 updates/s for batches of 200 cuts at n = 4096).
__global__ __launch_bounds__(256) void k_rs_prepare(const double* __restrict__ xc, double* __restrict__ xc0, long long n,
                                                    const DevState* __restrict__ st, DevState* __restrict__ st0,
                                                    unsigned* __restrict__ bar) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) xc0[i] = xc[i];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < RS_BAR_WORDS; i += 256) bar[i] = 0u;
        if (threadIdx.x < (int)(sizeof(DevState) / sizeof(long long)))
            reinterpret_cast<long long*>(st0)[threadIdx.x] = reinterpret_cast<const long long*>(st)[threadIdx.x];
    }
}

template <int R>
__global__ __launch_bounds__(RS_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ell_resident(ResidentArgs A) {
    constexpr int NV = 2 * R * RS_TS;            // partial sums a workgroup publishes (row sums | column sums)
    constexpr int NW = RS_THREADS / 64;          // waves
    __shared__ double sh_gr[2][R][RS_TS];        // g on this super-tile's row blocks (this cut | next cut)
    __shared__ double sh_gc[2][R][RS_TS];        // ... and on its column blocks
    __shared__ double sh_row[R][RS_TS];          // row sums of phase 1, then y on the row blocks
    __shared__ double sh_colw[R][NW][RS_TS];     // column sums per wave
    __shared__ double sh_col[R][RS_TS];          // y on the column blocks
    __shared__ double sh_w[NW];
    __shared__ int sh_ok;
    // R = 3: the third tile column lives in LDS (96 KiB), [tile row][block row i][thread] pairs: 144 resident registers
    // per thread left the compiler ~110 for everything else and it spilled 260-490 bytes per lane into scratch inside the
    // loop; 96 resident registers do not, at ~1 us of LDS traffic per cut.
    constexpr int RB = (R == 3) ? 2 : R;          // tile columns held in registers
    __shared__ double2_t sh_q[(R == 3) ? R * 4 * RS_THREADS : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const long long n = A.n, ld = A.ld;
    int SI = 0;
    while ((SI + 1) * (SI + 2) / 2 <= wg) ++SI;
    const int SJ = wg - SI * (SI + 1) / 2;
    const bool diag = SI == SJ;
    // this thread's 4 x 2 block inside every tile: rows 4 bi + i, columns 2 bj + j
    const int bi = (lane >> 5) + 2 * wave, bj = lane & 31;
    // ---- park the tiles.  Lower and diagonal tiles come from the lower triangle as stored; the strict upper triangle
    // of Q may be stale (lower-triangle apply passes), so whatever lies above the diagonal is read from its mirror image.
    double q[R][RB][4][2];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double e[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const long long r = (long long)(R * SI + a) * RS_TS + 4 * bi + i, c = (long long)(R * SJ + b) * RS_TS + 2 * bj + j;
                    e[j] = (r < n && c < n) ? (c <= r ? A.Q[r * ld + c] : A.Q[c * ld + r]) : 0.0;
                }
                if constexpr (R == 3) {
                    if (b < RB) { q[a][b < RB ? b : 0][i][0] = e[0]; q[a][b < RB ? b : 0][i][1] = e[1]; }
                    else sh_q[(a * 4 + i) * RS_THREADS + tid] = double2_t{e[0], e[1]};
                } else {
                    q[a][b][i][0] = e[0];
                    q[a][b][i][1] = e[1];
                }
            }
    // element pair (row i of this thread's block) of tile (a, b): from registers, or from LDS for the third column
    auto qget = [&](int a, int b, int i) -> double2_t {
        if (R == 3 && b >= RB) return sh_q[(a * 4 + i) * RS_THREADS + tid];
        return double2_t{q[a][b < RB ? b : 0][i][0], q[a][b < RB ? b : 0][i][1]};
    };
    auto qset = [&](int a, int b, int i, double2_t v) {
        if (R == 3 && b >= RB) {
            sh_q[(a * 4 + i) * RS_THREADS + tid] = v;
        } else {
            q[a][b < RB ? b : 0][i][0] = v.x;
            q[a][b < RB ? b : 0][i][1] = v.y;
        }
    };
    double kappa = A.st->kappa;
    unsigned bar = 0;
    int halted = A.st->halted;
    int err = 0;
    int cur = 0;
    // g on the 2 R blocks of this super-tile: NV <= 384 values, at most one per thread
    auto g_at = [&](const double* g, int idx) -> double {   // idx in [0, NV): row blocks first
        const int blk = idx / RS_TS, off = idx % RS_TS;
        const long long i = (long long)((blk < R ? R * SI + blk : R * SJ + (blk - R))) * RS_TS + off;
        return i < n ? g[i] : 0.0;
    };
    auto g_put = [&](int buf, int idx, double v) {
        const int blk = idx / RS_TS, off = idx % RS_TS;
        if (blk < R) sh_gr[buf][blk][off] = v; else sh_gc[buf][blk - R][off] = v;
    };
    if (A.count > 0 && !halted && tid < NV) g_put(0, tid, g_at(A.qgrads + A.first * n, tid));
    __syncthreads();
    long long cut = A.first;
    for (; cut < A.first + A.count && !halted; ++cut) {
        const int par = (int)((cut - A.first) & 1);
        RS_STAMP(0);
        // ---- 1. partial sums of y = Q g on the resident tiles, and this workgroup's share of omega
        double rs[R][4], cs[R][2];
        {
            double gr[R][4], gc[R][2];
#pragma unroll
            for (int a = 0; a < R; ++a) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    gr[a][i] = sh_gr[cur][a][4 * bi + i];
                    rs[a][i] = 0.0;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gc[a][j] = sh_gc[cur][a][2 * bj + j];
                    cs[a][j] = 0.0;
                }
            }
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double2_t e = qget(a, b, i);
                        rs[a][i] = __builtin_fma(e.x, gc[b][0], rs[a][i]);
                        cs[b][0] = __builtin_fma(e.x, gr[a][i], cs[b][0]);
                        rs[a][i] = __builtin_fma(e.y, gc[b][1], rs[a][i]);
                        cs[b][1] = __builtin_fma(e.y, gr[a][i], cs[b][1]);
                    }
            // omega = g'Qg: this super-tile's block once (diagonal: the whole symmetric block is here) or twice
            double w = 0.0;
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) w = __builtin_fma(gr[a][i], rs[a][i], w);
            w = wave_allreduce_sum(w);
            if (lane == 0) sh_w[wave] = w;
        }
#pragma unroll
        for (int a = 0; a < R; ++a) {
            const double rtot = rs_reduce4_over32(rs[a], lane);   // row 4 bi + (2 b0 + b1) over the super-tile's columns
            if (bj < 4) sh_row[a][4 * bi + 2 * (bj & 1) + ((bj >> 1) & 1)] = rtot;
            if (!diag) {
                const double ctot = rs_reduce2_halves(cs[a], lane);  // column 2 bj + b5 over this wave's 8 rows
                sh_colw[a][wave][2 * bj + ((lane >> 5) & 1)] = ctot;
            }
        }
        RS_STAMP(1);
        __syncthreads();
        double* mypart = A.part + ((long long)par * G + wg) * NV;
        if (tid < (diag ? NV / 2 : NV)) {
            const int blk = tid / RS_TS, off = tid % RS_TS;
            double v;
            if (blk < R) {
                v = sh_row[blk][off];
            } else {
                v = 0.0;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) v += sh_colw[blk - R][wv][off];
            }
            ho_store(mypart + tid, v);
        }
        if (tid == 0) {
            double wsum = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) wsum += sh_w[wv];
            ho_store(A.omega_part + (long long)par * G + wg, diag ? wsum : wsum + wsum);
        }
        // the next cut's gradient blocks: requested now, parked after the barrier
        const bool more = cut + 1 < A.first + A.count;
        const double gn = (more && tid < NV) ? g_at(A.qgrads + (cut + 1) * n, tid) : 0.0;
        RS_STAMP(2);
        if (cut - A.first == A.fault_at && wg == 1 % G) {  // test hook: as if this workgroup's wait had timed out
            err = 1;
            break;
        }
        if (!rs_grid_barrier(A.ctr, ++bar, &sh_ok)) {
            err = 1;
            break;
        }
        RS_STAMP(3);
        // ---- 2. y on this super-tile's blocks; omega; coefficients
        {
            const double* pbase = A.part + (long long)par * G * NV;
            // entry tid of [0, NV): block row (R SI + blk) for blk < R, block column (R SJ + blk - R) otherwise; its S
            // partial vectors in index order: row sums of the super-tiles (SB, sj <= SB), then column sums of (si > SB, SB)
            if (tid < (diag ? NV / 2 : NV)) {
                const int blk = tid / RS_TS, off = tid % RS_TS;
                const int SB = blk < R ? SI : SJ, sub = blk < R ? blk : blk - R;
                // 32-bit offsets from one uniform base; RS_SMAX loads in flight at R <= 2, half of them at a time at R = 3
                // (72 resident doubles per thread leave no room for 22 values + 22 offsets: it spilled 490 bytes per lane)
                constexpr int NB = (R == 3) ? 2 : 1, PER = RS_SMAX / NB;
                // (`opq`: an opaque zero formed inside the loop, so that the 22 offsets are recomputed per cut -- hoisted out
                // of the loop as invariants they occupied 22+ registers for the whole launch)
                unsigned opq = 0;
                asm volatile("" : "+v"(opq));
                const unsigned o_row = (unsigned)(sub * RS_TS + off) + opq, o_col = (unsigned)((R + sub) * RS_TS + off) + opq;
                double yv = 0.0;
#pragma unroll
                for (int h = 0; h < NB; ++h) {
                    double v[PER];
#pragma unroll
                    for (int u = 0; u < PER; ++u) {
                        const int k = h * PER + u;
                        const unsigned at = (k <= SB) ? (unsigned)rs_super_index(SB, k) * (unsigned)NV + o_row
                                                      : (unsigned)rs_super_index(k, SB) * (unsigned)NV + o_col;
                        v[u] = (k < A.S) ? ho_load(pbase + at) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < PER; ++u) yv += v[u];
                }
                if (blk < R) sh_row[blk][off] = yv; else sh_col[blk - R][off] = yv;
            }
            double w = (tid < G) ? ho_load(A.omega_part + (long long)par * G + tid) : 0.0;
            w = wave_allreduce_sum(w);
            if (lane == 0) sh_w[wave] = w;
            if (more && tid < NV) g_put(cur ^ 1, tid, gn);
        }
        __syncthreads();
        RS_STAMP(4);
        double omega = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) omega += sh_w[wv];
        const double tsq = kappa * omega;  // src/ell.rs:105
        Coef cf;
        const CutParams cp = A.qparams[cut];
        const int status = A.calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);  // :106
        if (wg == 0 && tid == 0) {
            A.qstatus[cut] = status;
            A.qtsq[cut] = tsq;
            A.st->tsq = tsq;
            A.st->omega = omega;
            A.st->status = status;
            queue_bookkeeping(A.st, status, tsq, 1);
        }
        if (status != 0) {  // :107-109: Q, xc, kappa untouched; the queue halts here (src/cutting_plane.rs:308)
            halted = 1;
            ++cut;
            break;
        }
        RS_STAMP(5);
        const double roo = cf.rho / omega, ratio = cf.sigma / omega;  // :112, :117
        kappa = kappa * cf.delta;                                     // :130
        // ---- 3. xc (the diagonal workgroups own their row blocks' entries), rank-1 in registers
        if (diag && tid < R * RS_TS) {
            const long long i = (long long)(R * SI) * RS_TS + tid;
            if (i < n) A.xc[i] = A.xc[i] - roo * sh_row[tid / RS_TS][tid % RS_TS];  // :113-115
        }
        {
            double vr[R][4], vc[R][2];
#pragma unroll
            for (int a = 0; a < R; ++a) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vr[a][i] = sh_row[a][4 * bi + i];
#pragma unroll
                for (int j = 0; j < 2; ++j) vc[a][j] = diag ? sh_row[a][2 * bj + j] : sh_col[a][2 * bj + j];
            }
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    // Off-diagonal super-tile, or a tile below the diagonal of a diagonal one: every element has row > col,
                    // (ratio gt[row]) gt[col] (r_qg of src/ell.rs:119).  Tile above the diagonal: the mirrored product
                    // (ratio gt[col]) gt[row].  Diagonal tile of a diagonal super-tile: per element, by a select (the lanes
                    // of a wave lie on both sides of the diagonal there: no divergent branches).
                    if (!diag || a > b) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double rr = ratio * vr[a][i];
                            double2_t e = qget(a, b, i);
                            e.x = e.x - rr * vc[b][0];
                            e.y = e.y - rr * vc[b][1];
                            qset(a, b, i, e);
                        }
                    } else if (a < b) {
                        const double rc0 = ratio * vc[b][0], rc1 = ratio * vc[b][1];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            double2_t e = qget(a, b, i);
                            e.x = e.x - rc0 * vr[a][i];
                            e.y = e.y - rc1 * vr[a][i];
                            qset(a, b, i, e);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            double2_t e = qget(a, b, i);
                            const double lo0 = (ratio * vr[a][i]) * vc[b][0], up0 = (ratio * vc[b][0]) * vr[a][i];
                            const double lo1 = (ratio * vr[a][i]) * vc[b][1], up1 = (ratio * vc[b][1]) * vr[a][i];
                            e.x = e.x - ((4 * bi + i >= 2 * bj) ? lo0 : up0);
                            e.y = e.y - ((4 * bi + i >= 2 * bj + 1) ? lo1 : up1);
                            qset(a, b, i, e);
                        }
                    }
                }
        }
        cur ^= 1;
        __syncthreads();  // sh_row / sh_col / sh_w are rewritten by the next cut
        RS_STAMP(6);
    }
    // ---- commit or abandon (rs_commit): the tiles go back only when EVERY workgroup has finished the batch
    if (!rs_commit(A.ctr, err != 0, &sh_ok)) {
        if (tid == 0) atomicExch(&A.st->solve_err, RS_WAIT_ERR);  // Q untouched; the host restores the rest and reruns
        return;
    }
    // ---- write the tiles back: the lower triangle (diagonal tiles whole); the copies above the diagonal are dropped
    // (the pitch goes through an opaque copy so that the 72 element addresses are formed HERE: otherwise the compiler
    // keeps the ones it formed for parking alive across the whole loop -- 144 registers)
    long long ldw = ld;
    asm volatile("" : "+s"(ldw));
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double2_t e = qget(a, b, i);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int I = R * SI + a, J = R * SJ + b;
                    const long long r = (long long)I * RS_TS + 4 * bi + i, c = (long long)J * RS_TS + 2 * bj + j;
                    if (J <= I && r < n && c < n) A.Q[r * ldw + c] = j ? e.y : e.x;
                }
            }
    if (wg == 0 && tid == 0) {
        A.st->kappa = kappa;
        // cuts behind a halt never ran: the status the streamed scalar stage reports for them (ELLHIP_UNKNOWN)
        const double t_last = A.st->tsq;
        for (long long c = cut; c < A.first + A.count; ++c) {
            A.qstatus[c] = ST_UNKNOWN;
            A.qtsq[c] = t_last;
        }
    }
}

}  // namespace ellhip
