// resident_kernels.hpp -- Ell::update_core (src/ell.rs:97-137) for a whole QUEUE of cuts in ONE persistent launch with the
// matrix parked ON-CHIP: the loop `for cut in queue { update }` of src/cutting_plane.rs:299-311 around the update, as
// SURVEY section 7's k_ell_fused asks, for the sizes where the lower triangle fits the chip's register files.
//
// Why: at n = 4096 the lower triangle of Q is 64 MiB, the 256 CUs hold 128 MiB of vector registers, and the streamed
// schedule still moves the matrix from the Infinity Cache every update: 39 us per update for 23 + 11 us of kernels
// (profiles/r02/bench_n4096_kernel_stats.csv).  Here each workgroup (one per CU, 256 threads, one wave per SIMD with the
// whole 512-register budget) keeps up to RS_TPW tiles of 64 x 64 doubles of the lower triangle in registers for the
// whole launch -- thread (bi, bj) of a tile owns its 4 x 4 block (rows 4 bi .., columns 4 bj ..) -- and an update is
//
//   1. GEMV on the resident tiles: per tile 64 row sums (sum over the tile's columns) and, off the diagonal, 64 column
//      sums (the mirrored half); 4 x 4 blocks, the cross-lane sums as transposing reductions (5 DPP exchanges over the 16
//      lanes of a row of blocks, 3 shuffles over the 4 row blocks of a wave, LDS across the 4 waves).  Partial sums go to
//      part[tile][128] in L2 (write-through).                                                          src/arr.rs:426-442
//   -- grid barrier 1 --
//   2. every workgroup reduces ITS slices of 16 entries of y = Q g from the T partial vectors each entry has (fixed
//      order), writes them (y doubles as gt: nothing is recorded here, the rank-1 is applied at once) together with its
//      share of omega = g . y.                                                                       src/arr.rs:443-451
//   -- grid barrier 2 --
//   3. every workgroup adds the omega shares in one fixed tree, runs EllCalc redundantly (identical bits everywhere),
//      the slice owners update xc, and every tile gets its rank-1 in registers with the reference's roundings:
//      x <- x - (ratio * gt[hi]) * gt[lo], hi = max(row, col) (separate multiply and subtract).  src/ell.rs:105-130
//
// A failing cut (status != Success) leaves Q, xc and kappa untouched and ends the loop on every workgroup at the same
// cut (they all computed the same status): src/cutting_plane.rs:308.  Results: the rank-1 follows the reference's
// roundings for the gt it is given; gt itself is summed in this kernel's own fixed association (per 4 x 4 block, per
// tile, then T partial vectors in index order), so states agree with the other schedules to ~1e-15 and with the CPU path
// within the 1e-10 contract, and are bit-reproducible run to run (no atomics in the data path, static assignment).
//
// In-launch synchronisation: two grid-wide barriers per cut on one counter that only grows (arrive = write-through
// stores drained, workgroup barrier, one atomic add; wait = one lane polls with agent-scope loads, bounded; then an
// acquire fence).  The grid is one workgroup per CU and must be resident as a whole: the host checks the occupancy
// (1 x CU count) and the tile capacity before it chooses this path, and a time-out sets DevState.solve_err (the call
// fails, the GPU is not hung).
#pragma once

#include <hip/hip_runtime.h>

#include "ell_kernels.hpp"

namespace ellhip {

constexpr int RS_TS = 64;     // tile size
constexpr int RS_TPW = 9;     // tiles a workgroup can hold: 9 x 16 doubles per thread = 288 VGPRs
constexpr int RS_SLICE = 16;  // entries of y a reduce task covers
constexpr int RS_WAIT_ERR = 9;

struct ResidentArgs {
    double* Q;                 // n x ld row-major; the lower triangle (tiles J <= I) is read at entry and written at exit
    long long ld, n;
    int T;                     // tile rows = ceil(n / 64)
    int ntiles;                // T (T + 1) / 2
    const double* qgrads;      // queue: gradients [k][n]
    const CutParams* qparams;  // queue: cut scalars
    int* qstatus;              // queue: per-cut status out
    double* qtsq;              // queue: per-cut tsq out
    long long first, count;    // cuts [first, first + count)
    double* xc;                // n
    DevState* st;
    double* part;              // [ntiles][128] partial sums (row sums | column sums)
    double* y;                 // npad = 64 T doubles: Q g of the current cut
    double* omega_part;        // [nslices] shares of omega
    unsigned* ctr;             // barrier counter, zero at launch
    EllCalcDev calc;
};

__device__ __forceinline__ int rs_tile_index(int I, int J) { return I * (I + 1) / 2 + J; }

// DPP move of a double (two 32-bit halves); CTRL: quad_perm 0x00-0xFF, row_ror:n 0x120 + n
template <int CTRL>
__device__ __forceinline__ double rs_dpp(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// v[0..3] summed over the 16 lanes of a DPP row (lane bits 0-3): on return every lane holds the total of
// v[2 * bit0 + bit1] (bit0, bit1 of its own lane id).  5 exchanges.
__device__ __forceinline__ double rs_reduce4_row16(const double (&v)[4], int lane) {
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
    return k;
}

// v[0..3] summed over the 4 groups of 16 lanes of a wave (lane bits 4-5): every lane ends with the total of
// v[2 * bit4 + bit5].  3 exchanges.
__device__ __forceinline__ double rs_reduce4_groups(const double (&v)[4], int lane) {
    const bool b4 = lane & 16, b5 = lane & 32;
    double k2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b4 ? v[2 + i] : v[i], send = b4 ? v[i] : v[2 + i];
        k2[i] = keep + __shfl_xor(send, 16, 64);
    }
    return (b5 ? k2[1] : k2[0]) + __shfl_xor(b5 ? k2[0] : k2[1], 32, 64);
}

// one lane waits until the counter has reached `target`; bounded
__device__ __forceinline__ bool rs_wait(const unsigned* ctr, unsigned target) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// grid barrier: everything this workgroup stored write-through is visible to the others when they pass it
__device__ __forceinline__ bool rs_grid_barrier(unsigned* ctr, unsigned target, int* sh_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        *sh_ok = rs_wait(ctr, target) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return *sh_ok != 0;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ell_resident(ResidentArgs A) {
    __shared__ double sh_col[RS_TPW][4][RS_TS];  // column sums per wave (18 KiB)
    __shared__ double sh_row[RS_TPW][RS_TS];     // row sums (4.5 KiB)
    __shared__ double sh_red[4][RS_SLICE];
    __shared__ double sh_w[4];
    __shared__ int sh_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const long long n = A.n, ld = A.ld;
    // tiles of this workgroup: a contiguous run of the row-major list of lower-triangle tiles
    const int base = A.ntiles / G, rem = A.ntiles % G;
    const int t0 = wg * base + (wg < rem ? wg : rem);
    const int nt = base + (wg < rem ? 1 : 0);
    int tI[RS_TPW], tJ[RS_TPW];
    {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t0) ++I;
        int J = t0 - I * (I + 1) / 2;
#pragma unroll
        for (int t = 0; t < RS_TPW; ++t) {
            tI[t] = I;
            tJ[t] = J;
            if (++J > I) {
                J = 0;
                ++I;
            }
        }
    }
    const int bi = (lane >> 4) + 4 * wave, bj = lane & 15;  // this thread's 4 x 4 block inside every tile
    // ---- park the tiles
    double q[RS_TPW][4][4];
#pragma unroll
    for (int t = 0; t < RS_TPW; ++t) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long long r = (long long)tI[t] * RS_TS + 4 * bi + a, c = (long long)tJ[t] * RS_TS + 4 * bj;
#pragma unroll
            for (int b = 0; b < 4; ++b) q[t][a][b] = (t < nt && r < n && c + b < n) ? A.Q[r * ld + c + b] : 0.0;
        }
    }
    double kappa = A.st->kappa;
    const int nslices = (int)((n + RS_SLICE - 1) / RS_SLICE);
    unsigned bar = 0;
    int halted = A.st->halted;
    int err = 0;
    for (long long cut = A.first; cut < A.first + A.count && !halted; ++cut) {
        const double* g = A.qgrads + cut * n;
        // ---- 1. partial sums of y = Q g on the resident tiles
#pragma unroll
        for (int t = 0; t < RS_TPW; ++t) {
            if (t < nt) {
                const long long r0 = (long long)tI[t] * RS_TS + 4 * bi, c0 = (long long)tJ[t] * RS_TS + 4 * bj;
                double gr[4], gc[4], rs[4], cs[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    gr[a] = (r0 + a < n) ? g[r0 + a] : 0.0;
                    gc[a] = (c0 + a < n) ? g[c0 + a] : 0.0;
                    rs[a] = 0.0;
                    cs[a] = 0.0;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        rs[a] = __builtin_fma(q[t][a][b], gc[b], rs[a]);
                        cs[b] = __builtin_fma(q[t][a][b], gr[a], cs[b]);
                    }
                const double rtot = rs_reduce4_row16(rs, lane);   // row 4 bi + (2 b0 + b1) over the tile's 64 columns
                if (bj < 4) sh_row[t][4 * bi + 2 * (bj & 1) + ((bj >> 1) & 1)] = rtot;
                const double ctot = rs_reduce4_groups(cs, lane);  // column 4 bj + (2 b4 + b5) over this wave's 16 rows
                sh_col[t][wave][4 * bj + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1)] = ctot;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < nt * 128; idx += 256) {
            const int t = idx >> 7, e = idx & 127;
            // (tile index: the run is contiguous in the list)
            double v;
            if (e < RS_TS) v = sh_row[t][e];
            else v = ((sh_col[t][0][e - RS_TS] + sh_col[t][1][e - RS_TS]) + sh_col[t][2][e - RS_TS]) + sh_col[t][3][e - RS_TS];
            ho_store(A.part + (long long)(t0 + t) * 128 + e, v);
        }
        if (!rs_grid_barrier(A.ctr, (unsigned)G * ++bar, &sh_ok)) {
            err = 1;
            break;
        }
        // ---- 2. y on this workgroup's slices, and its share of omega
        for (int sl = wg; sl < nslices; sl += G) {
            const int e = tid & 15, p = tid >> 4;   // entry of the slice, partial-vector group
            const long long i = (long long)sl * RS_SLICE + e;
            const int b = (int)(i / RS_TS), off = (int)(i % RS_TS);
            double s = 0.0;
            // T partial vectors per entry: row sums of the tiles (b, J <= b), column sums of the tiles (I > b, b)
            for (int k = p; k < A.T; k += 16) {
                const long long at = (k <= b) ? (long long)rs_tile_index(b, k) * 128 + off
                                              : (long long)rs_tile_index(k, b) * 128 + RS_TS + off;
                s += ho_load(A.part + at);
            }
            // p = (lane >> 4) + 4 wave: over the lane groups, then over the waves
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) sh_red[wave][lane] = s;
            __syncthreads();
            if (tid < RS_SLICE) {
                const double yv = ((sh_red[0][tid] + sh_red[1][tid]) + sh_red[2][tid]) + sh_red[3][tid];
                double w = (i < n) ? g[i] * yv : 0.0;
                ho_store(A.y + i, yv);
                w += rs_dpp<0xB1>(w);
                w += rs_dpp<0x4E>(w);
                w += rs_dpp<0x124>(w);
                w += rs_dpp<0x128>(w);
                if (tid == 0) ho_store(A.omega_part + sl, w);
            }
            __syncthreads();
        }
        if (!rs_grid_barrier(A.ctr, (unsigned)G * ++bar, &sh_ok)) {
            err = 1;
            break;
        }
        // ---- 3. omega, coefficients (every workgroup, identical bits), xc, rank-1 in registers
        double w = 0.0;
        for (int k = tid; k < nslices; k += 256) w += ho_load(A.omega_part + k);
        w = wave_allreduce_sum(w);
        if (lane == 0) sh_w[wave] = w;
        __syncthreads();
        const double omega = ((sh_w[0] + sh_w[1]) + sh_w[2]) + sh_w[3];
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
            break;
        }
        const double roo = cf.rho / omega, ratio = cf.sigma / omega;  // :112, :117
        kappa = kappa * cf.delta;                                     // :130
        for (int sl = wg; sl < nslices; sl += G)
            if (tid < RS_SLICE) {
                const long long i = (long long)sl * RS_SLICE + tid;
                if (i < n) A.xc[i] = A.xc[i] - roo * ho_load(A.y + i);  // :113-115
            }
#pragma unroll
        for (int t = 0; t < RS_TPW; ++t) {
            if (t < nt) {
                const long long r0 = (long long)tI[t] * RS_TS + 4 * bi, c0 = (long long)tJ[t] * RS_TS + 4 * bj;
                double vr[4], vc[4], rr[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    vr[a] = ho_load(A.y + r0 + a);
                    vc[a] = ho_load(A.y + c0 + a);
                }
                if (tI[t] != tJ[t] || bi > bj) {  // every element strictly below the diagonal: (ratio gt[row]) gt[col]
#pragma unroll
                    for (int a = 0; a < 4; ++a) rr[a] = ratio * vr[a];  // r_qg of src/ell.rs:119
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) q[t][a][b] = q[t][a][b] - rr[a] * vc[b];
                } else {  // diagonal tile, block on or above the diagonal: hi = max(row, col)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const bool lower = r0 + a >= c0 + b;
                            const double upd = lower ? (ratio * vr[a]) * vc[b] : (ratio * vc[b]) * vr[a];
                            q[t][a][b] = q[t][a][b] - upd;
                        }
                }
            }
        }
    }
    // ---- write the tiles back (lower triangle; diagonal tiles whole) and the scalars
#pragma unroll
    for (int t = 0; t < RS_TPW; ++t) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const long long r = (long long)tI[t] * RS_TS + 4 * bi + a, c = (long long)tJ[t] * RS_TS + 4 * bj;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (t < nt && r < n && c + b < n) A.Q[r * ld + c + b] = q[t][a][b];
        }
    }
    if (wg == 0 && tid == 0) {
        A.st->kappa = kappa;
        if (err) atomicExch(&A.st->solve_err, RS_WAIT_ERR);
    }
}

}  // namespace ellhip
