// symvq_kernels.hpp -- one launch per cut on the recorded ("deferred") schedule of an unsharded Ell handle:
//     y = Q_base g  (lower triangle only, 4 n^2 bytes)  ->  gt, omega, EllCalc, xc, kappa  (src/ell.rs:102-115,130)
// replacing the three launches k_symv -> k_symv_reduce<NP> -> k_scalar_apply_def<NP> of ell_kernels.hpp.
//
// What bounds the static (strip, segment) grid of k_symv (measured, tools/tune_symvq.hip): a CU streams at most
// ~27 GB/s from HBM when every CU streams, the whole grid is resident at once (1152 tiles on 1280 slots at
// n = 16384), so CUs hold 4 or 5 one-MiB tiles and the launch ends when the CUs with 5 are done (190 us against
// 152 us for an even split); tiles pulled from a queue did not help (a 256 KiB tile takes 25-40 us of a 190 us launch:
// workgroups end 25-50 us apart).  So the triangle is cut into PIECES OF EQUAL BYTES instead:
//
//   segment J = columns [512 J, 512 J + 512); its rows are J*512 .. n-1 (row r holds min(512, r - 512 J + 1) elements
//   at or left of the diagonal).  Walking the segments in order and each segment's rows top to bottom, the stream of
//   16-byte-per-lane loads is cut into 1024 runs of equal length; a run that crosses a segment boundary is two pieces.
//   Workgroup w owns run w: every workgroup streams the same number of bytes (to one row), all of them start at
//   once and, the per-CU rate being what bounds them, end together.  The partition depends on n only.
//
//   piece (J, ra, rb): wave w of the workgroup streams rows ra + w, ra + w + 4, ... (4 KiB contiguous per row),
//   two rows in flight; every element feeds its row sum (y_r += q g_c) and, below the diagonal, its column sum
//   (y_c += q g_r).  Row sums go to rowpart[J][r]; the four waves' column sums are combined through LDS in the fixed
//   order ((w0+w1)+w2)+w3 into colpart[piece][512] -- 4 MiB of column partial sums per launch where the strip tiling
//   wrote 17.
//
// The rest of the cut runs in the same launch, as tasks pulled from a small queue by whoever has finished its run:
//   reduce b   y[i] = sum_{J <= i/512} rowpart[J][i] + sum_{pieces p of segment i/512} colpart[p][i mod 512] for 128
//              columns (fixed order), plus the block's share of the dot products g.y and v_j.g; waits (bounded) until
//              every piece of the segments 0..i/512 has been counted done.
//   scalar s   waits for all reduce tasks; every scalar task forms omega, tsq and the EllCalc coefficients
//              redundantly (identical bits), then updates its slice: gt = y - sum_j (c_j d_j) v_j recorded as the
//              new pending vector, xc -= (rho/omega) gt.  Task 0 also publishes kappa / status / DevState.
// Critical path after the last run ends: one reduce task + one scalar task (~10 us) instead of two more launches with
// their boundaries (27 us).  Every partial sum has a fixed slot and every sum a fixed order: the bits depend on n only,
// not on the grid size, the dispatch order or who executes a task.
//
// Hand-offs inside the launch follow cdna_hip_programming.md Guideline 16 (write-through form): payload stored with
// agent-scope relaxed atomic stores (sc1), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane
// adds to the counter; the consumer polls the counter with ONE lane (relaxed agent loads, bounded, s_sleep), barrier,
// then loads the payload with agent-scope relaxed atomic loads (sc1: never served from this CU's L1).  The counters are
// re-armed by the last workgroup to leave.  A time-out sets DevState.solve_err (the call fails, the GPU is not hung).
// The grid must be resident as a whole (a workgroup that waits holds its slot): the host sizes it from the occupancy
// query and the CU count; with fewer than 1024 workgroups a workgroup takes several runs.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"

namespace ellhip {

constexpr int SQ_SEG = 512;      // columns per segment: 4 x 16 bytes per lane per row, 4 KiB contiguous per wave and row
constexpr int SQ_NCH = SQ_SEG / 128;
constexpr int SQ_RUNS = 1024;    // runs of equal bytes (4 workgroups per CU on MI355X); fixed: the bits depend on n only
constexpr int SQ_MAXSEG = 64;    // n <= 32768
constexpr int SQ_LPART_BLOCKS = 128;  // reduce blocks whose dot products pass through LDS at a time (a multiple of 8)

struct SymvqCtl {                // device-resident, zeroed at creation, re-armed by the last workgroup of every launch
    unsigned next;               // queue head of the reduce / scalar tasks
    unsigned exited;             // workgroups that have left this launch
    unsigned reduce_done;
    unsigned pad;
    unsigned seg_done[SQ_MAXSEG];  // pieces completed per segment
};

struct SymvqPiece {
    int J, ra, rb, pad;          // segment, rows [ra, rb) (global row indices)
};

struct SymvqPlan {               // host-built, lives in device memory next to the pieces
    int nseg, npiece, nblock, nslice;
    int ndyn_first, ndyn;        // pieces [ndyn_first, npiece) are pulled from the queue after the static runs
    int run_first[SQ_RUNS + 1];  // pieces of run w: [run_first[w], run_first[w + 1])
    int seg_pfirst[SQ_MAXSEG + 1];  // pieces of segment J: [seg_pfirst[J], seg_pfirst[J + 1])
};

__device__ __forceinline__ void sq_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double sq_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool sq_wait_ge(const unsigned* ctr, unsigned want) {  // ONE lane; bounded
    for (int spin = 0; spin < (1 << 21); ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// ------------------------------------------------------------------------------------------ piece ----
// Rows [ra, rb) of segment J.  HANDOFF: the partial sums are consumed inside this launch (write-through stores).
// EVERY load is unconditional and in bounds (a branch around a load makes hipcc wait for it at the join): a row beyond
// rb - 1 is clamped to it, a 1 KiB chunk that lies wholly right of the row's diagonal becomes a one-line broadcast load
// of the row's first pair (no HBM traffic), a lane beyond column n - 2 reads the pair at n - 2; what must not count is
// masked when it is consumed.
template <int RW, bool NT, bool HANDOFF>
__device__ __forceinline__ void sq_piece(const double* __restrict__ Q, long long ld, long long n, int J, int ra, int rb,
                                         int piece, const double* __restrict__ g, double* __restrict__ rowpart,
                                         double* __restrict__ colpart, double (*lcol)[SQ_SEG]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = J * SQ_SEG;
    const int lim = (int)(n - 2) - c0;   // largest in-bounds pair offset inside this segment
    const double* Qseg = Q + c0;
    double2_t gc[SQ_NCH], accc[SQ_NCH];
#pragma unroll
    for (int k = 0; k < SQ_NCH; ++k) {
        const int c = c0 + 128 * k + 2 * lane;
        gc[k] = (c <= n - 2) ? *reinterpret_cast<const double2_t*>(g + c) : double2_t{0.0, 0.0};
        accc[k] = double2_t{0.0, 0.0};
    }
    const int nrow = rb - ra;
    const int ngrp = (nrow + 4 * RW - 1) / (4 * RW);   // groups of RW rows per wave
#pragma unroll 1
    for (int grp = 0; grp < ngrp; ++grp) {
        double2_t q[RW][SQ_NCH];
        double gr[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int idx = wave + 4 * (grp * RW + r);           // wave-uniform
            const int rr = ra + (idx < nrow ? idx : nrow - 1);
            const double* row = Qseg + (long long)rr * ld;
            const int dg = rr - c0;                               // segment-local column of the diagonal
            gr[r] = idx < nrow ? g[rr] : 0.0;                     // clamped rows count nothing
#pragma unroll
            for (int k = 0; k < SQ_NCH; ++k) {
                // a chunk wholly right of the diagonal: every lane re-reads the row's first pair (ONE cache line for the
                // wave instead of sixteen -- re-reading chunk 0 lane by lane made the diagonal blocks' runs 1.4x slower)
                int off = (128 * k <= dg) ? 128 * k + 2 * lane : 0;
                off = off < lim ? off : lim;
                q[r][k] = ld_stream<NT, double2_t>(row + off);
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int idx = wave + 4 * (grp * RW + r);
            const int rr = ra + idx;
            const int dg = rr - c0;
            double accr = 0.0;
#pragma unroll
            for (int k = 0; k < SQ_NCH; ++k) {
                const int c = 128 * k + 2 * lane;
                double qx = q[r][k].x, qy = q[r][k].y;
                if (c > dg) qx = 0.0;        // elements above the diagonal take no part (they may hold anything)
                if (c + 1 > dg) qy = 0.0;
                accr += qx * gc[k].x;
                accr += qy * gc[k].y;
                // column sums take strictly-below-diagonal elements only (the diagonal counts once, in the row sum)
                const double cx = (c < dg) ? qx : 0.0;
                const double cy = (c + 1 < dg) ? qy : 0.0;
                accc[k].x += cx * gr[r];
                accc[k].y += cy * gr[r];
            }
            const double s = wave_allreduce_sum(accr);
            if (lane == 0 && idx < nrow) {
                double* p = rowpart + (long long)J * n + rr;
                if (HANDOFF) sq_store(p, s); else *p = s;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < SQ_NCH; ++k) *reinterpret_cast<double2_t*>(&lcol[wave][128 * k + 2 * lane]) = accc[k];
    __syncthreads();
#pragma unroll
    for (int c = threadIdx.x; c < SQ_SEG; c += 256) {
        const double v = ((lcol[0][c] + lcol[1][c]) + lcol[2][c]) + lcol[3][c];
        double* p = colpart + (long long)piece * SQ_SEG + c;
        if (HANDOFF) sq_store(p, v); else *p = v;
    }
    __syncthreads();  // lcol is reused by the next piece
}

// Stand-alone run phase (tuning harness; plain stores, a later launch reduces).
template <int RW, bool NT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_symvq_runs(const double* __restrict__ Q, long long ld, long long n,
                                                         const double* __restrict__ g, double* __restrict__ rowpart,
                                                         double* __restrict__ colpart,
                                                         const SymvqPlan* __restrict__ plan,
                                                         const SymvqPiece* __restrict__ pieces,
                                                         SymvqCtl* __restrict__ ctl,
                                                         unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) double lcol[4][SQ_SEG];
    __shared__ int sh_task;
    const unsigned long long t_begin = stamps ? wall_clock64() : 0ull;
    for (int w = blockIdx.x; w < SQ_RUNS; w += gridDim.x) {
        const int p0 = plan->run_first[w], p1 = plan->run_first[w + 1];
        for (int p = p0; p < p1; ++p) {
            const SymvqPiece pc = pieces[p];
            sq_piece<RW, NT, false>(Q, ld, n, pc.J, pc.ra, pc.rb, p, g, rowpart, colpart, lcol);
        }
    }
    const unsigned long long t_static = stamps ? wall_clock64() : 0ull;
    const int ndyn = plan->ndyn, dfirst = plan->ndyn_first;
    int ndone = 0;
    for (;;) {
        if (threadIdx.x == 0) sh_task = (int)atomicAdd(&ctl->next, 1u);
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(sh_task);
        __syncthreads();
        if (t >= ndyn) break;
        const SymvqPiece pc = pieces[dfirst + t];
        sq_piece<RW, NT, false>(Q, ld, n, pc.J, pc.ra, pc.rb, dfirst + t, g, rowpart, colpart, lcol);
        ++ndone;
    }
    if (threadIdx.x == 0) {
        if (stamps) {
            stamps[4 * blockIdx.x] = t_begin;
            stamps[4 * blockIdx.x + 1] = t_static;
            stamps[4 * blockIdx.x + 2] = wall_clock64();
            stamps[4 * blockIdx.x + 3] = (unsigned long long)ndone;
        }
        const unsigned e = atomicAdd(&ctl->exited, 1u);
        if (e == gridDim.x - 1) {
            __hip_atomic_store(&ctl->next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------- reduce task ----
// 128 columns i = 128 b + 2 lane (+1).  Terms of y[i], in this fixed order: rowpart[J'][i], J' = 0..J, then
// colpart[p][i - c0] for the pieces p of segment J in order; wave w takes terms w, w + 4, ... (8 loads in flight),
// the four wave sums are combined as ((w0+w1)+w2)+w3.  Then the block's dot products (as k_symv_reduce<NP>).
template <int NP>
__device__ __forceinline__ void sq_reduce_task(int b, long long n, const SymvqPlan* __restrict__ plan,
                                               const double* __restrict__ rowpart, const double* __restrict__ colpart,
                                               const double* __restrict__ g, const double* __restrict__ pend,
                                               double* __restrict__ y, double* __restrict__ partial,
                                               double2_t (*part)[64]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long i = (long long)b * 128 + 2 * lane;
    const int J = (int)(((long long)b * 128) / SQ_SEG);
    const int p0 = plan->seg_pfirst[J], p1 = plan->seg_pfirst[J + 1];
    const int nterm = (J + 1) + (p1 - p0);
    const int cl = (int)(i - (long long)J * SQ_SEG);
    // operands of the dot products that do not depend on y: requested first
    constexpr int NPW = (NP + 3) / 4;
    double2_t gi = {0.0, 0.0};
    double2_t pv[NPW > 0 ? NPW : 1];
    if (i < n) gi = *reinterpret_cast<const double2_t*>(g + i);
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int j = wave + 4 * k;
        pv[k] = (i < n && j < NP) ? *reinterpret_cast<const double2_t*>(pend + (long long)j * n + i) : double2_t{0.0, 0.0};
    }
    auto term = [&](int t, double& vx, double& vy) {
        const double* p = (t <= J) ? rowpart + (long long)t * n + i : colpart + (long long)(p0 + t - (J + 1)) * SQ_SEG + cl;
        vx = sq_load(p);
        vy = sq_load(p + 1);
    };
    double2_t s = {0.0, 0.0};
    if (i < n) {
        int t = wave;
        for (; t + 28 < nterm; t += 32) {
            double vx[8], vy[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) term(t + 4 * u, vx[u], vy[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += vx[u];
                s.y += vy[u];
            }
        }
        for (; t < nterm; t += 4) {
            double vx, vy;
            term(t, vx, vy);
            s.x += vx;
            s.y += vy;
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    double2_t yv = {0.0, 0.0};
    if (wave == 0 && i < n) {
        const double2_t c0 = part[0][lane], c1 = part[1][lane], c2 = part[2][lane], c3 = part[3][lane];
        yv.x = ((c0.x + c1.x) + c2.x) + c3.x;
        yv.y = ((c0.y + c1.y) + c2.y) + c3.y;
        sq_store(y + i, yv.x);
        sq_store(y + i + 1, yv.y);
    }
    double* out = partial + (long long)b * (NP + 1);
    if (wave == 0) {
        double sgy = gi.x * yv.x;
        sgy += gi.y * yv.y;
        sgy = wave_allreduce_sum(sgy);
        if (lane == 0) sq_store(out, sgy);
    }
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int j = wave + 4 * k;
        double sv = pv[k].x * gi.x;
        sv += pv[k].y * gi.y;
        sv = wave_allreduce_sum(sv);
        if (lane == 0 && j < NP) sq_store(out + 1 + j, sv);
    }
    __syncthreads();  // part is reused by the next task
}

// ------------------------------------------------------------------------------------- scalar task ----
// Slice s of the scalar stage (k_scalar_apply_def's arithmetic; the partial sums of the nblock reduce blocks are
// brought into LDS in one round trip and added per column as 8 interleaved running sums in a fixed order).
template <int NP>
__device__ __forceinline__ void sq_scalar_task(int sl, long long n, int nblock, const double* __restrict__ y,
                                               double* __restrict__ xc, double* __restrict__ pend,
                                               double* __restrict__ cpend, const double* __restrict__ partial,
                                               DevState* __restrict__ st, double kappa_in, const EllCalcDev& calc,
                                               const CutParams* __restrict__ cp_dev, const CutParams& cp_val, int slot,
                                               int queue_mode, int* __restrict__ q_status, double* __restrict__ q_tsq,
                                               double* lpart /* [SQ_LPART_BLOCKS * (NP + 1)] */, double (*psum)[32],
                                               double* bc /* [NP + 2] */, int* bc_status) {
    const int tid = threadIdx.x;
    const bool lead = sl == 0;
    // the reduce blocks' partial sums come through LDS 128 blocks at a time (one memory round trip per pass); thread
    // (q, c) adds the rows q, q + 8, ... of column c in ascending order, whatever the pass boundaries
    const int c = tid & 31, q = tid >> 5;
    double a = 0.0;
    for (int b0 = 0; b0 < nblock; b0 += SQ_LPART_BLOCKS) {
        const int nb = (nblock - b0 < SQ_LPART_BLOCKS) ? nblock - b0 : SQ_LPART_BLOCKS;
        const int total = nb * (NP + 1);
        for (int k = tid; k < total; k += 256) lpart[k] = sq_load(partial + (long long)b0 * (NP + 1) + k);
        __syncthreads();
        if (c <= NP)
            for (int b = q; b < nb; b += 8) a += lpart[b * (NP + 1) + c];
        __syncthreads();
    }
    if (c <= NP) psum[q][c] = a;
    __syncthreads();
    if (tid == 0) {
        double d[NP + 1];
#pragma unroll
        for (int k = 0; k <= NP; ++k)
            d[k] = ((((((psum[0][k] + psum[1][k]) + psum[2][k]) + psum[3][k]) + psum[4][k]) + psum[5][k]) + psum[6][k]) +
                   psum[7][k];
        double omega = d[0];  // g.(Q_base g)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            // slot `slot` is written by this very launch (below): it is empty by definition, so it is not read
            const double cd = (j == slot) ? 0.0 : cpend[j] * d[1 + j];  // c_j (v_j.g)
            bc[j] = cd;
            omega = omega - cd * d[1 + j];
        }
        const double tsq = kappa_in * omega;  // src/ell.rs:105
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
                st->ratio = cf.sigma / omega;      // :117
                st->kappa = kappa_in * cf.delta;   // :130 (the recorded schedule never runs with no_defer_trick)
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
        bc[NP] = roo;
        *bc_status = status;
    }
    __syncthreads();
    if (*bc_status == ST_SUCCESS) {
        const double roo = bc[NP];
        double cd[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) cd[j] = bc[j];
        double* vnew = pend + (long long)slot * n;
        const long long m = scalar_slice(n);
        const long long lo = (long long)sl * m;
        const long long hi = (lo + m < n) ? lo + m : n;
        for (long long i = lo + tid; i < hi; i += 256) {
            double gt = sq_load(y + i);
#pragma unroll
            for (int j = 0; j < NP; ++j) gt = gt - cd[j] * pend[(long long)j * n + i];
            vnew[i] = gt;               // slot `slot` was all zeros until now: its own term above was an exact 0
            xc[i] = xc[i] - roo * gt;   // :113-115
        }
    }
    __syncthreads();  // LDS is reused by the next task
}

// ------------------------------------------------------------------------------------ the launch ----
template <int NP, bool NT>
__global__ __launch_bounds__(256, 4) void k_symvq(const double* __restrict__ Q, long long ld, long long n,
                                                  const double* __restrict__ g, double* __restrict__ y,
                                                  double* __restrict__ xc, double* __restrict__ pend,
                                                  double* __restrict__ cpend, double* __restrict__ rowpart,
                                                  double* __restrict__ colpart, double* __restrict__ partial,
                                                  DevState* __restrict__ st, const SymvqPlan* __restrict__ plan,
                                                  const SymvqPiece* __restrict__ pieces, SymvqCtl* __restrict__ ctl,
                                                  EllCalcDev calc, const CutParams* __restrict__ cp_dev,
                                                  CutParams cp_val, int slot, int queue_mode,
                                                  int* __restrict__ q_status, double* __restrict__ q_tsq) {
    __shared__ __attribute__((aligned(16))) double lds[SQ_LPART_BLOCKS * (MAXPEND + 1)];  // lcol[4][512] | lpart
    __shared__ double2_t part[4][64];
    __shared__ double psum[8][32];
    __shared__ double bc[MAXPEND + 2];
    __shared__ int bc_status;
    __shared__ int sh_task, sh_ok;
    static_assert(SQ_LPART_BLOCKS * (MAXPEND + 1) >= 4 * SQ_SEG && SQ_LPART_BLOCKS % 8 == 0, "LDS carve");
    double (*lcol)[SQ_SEG] = reinterpret_cast<double (*)[SQ_SEG]>(lds);
    const int tid = threadIdx.x;
    // `halted` and kappa are read before anything in this launch can have rewritten them: scalar task 0 runs only
    // after every run is done, i.e. after every workgroup of the grid has started (each owns at least one run).
    const int halted = st->halted;
    const double kappa_in = st->kappa;
    int err = 0;
    if (!halted) {
        for (int w = blockIdx.x; w < SQ_RUNS; w += gridDim.x) {
            const int p0 = plan->run_first[w], p1 = plan->run_first[w + 1];
            for (int p = p0; p < p1; ++p) {
                const SymvqPiece pc = pieces[p];
                sq_piece<2, NT, true>(Q, ld, n, pc.J, pc.ra, pc.rb, p, g, rowpart, colpart, lcol);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
                __syncthreads();
                if (tid == 0) atomicAdd(&ctl->seg_done[pc.J], 1u);
            }
        }
        // queue: [dynamic pieces][reduce tasks][scalar tasks]
        const int ndyn = plan->ndyn, dfirst = plan->ndyn_first;
        const int nblock = plan->nblock, ntask = ndyn + nblock + plan->nslice;
        for (;;) {
            if (tid == 0) sh_task = (int)atomicAdd(&ctl->next, 1u);
            __syncthreads();
            const int t = __builtin_amdgcn_readfirstlane(sh_task);
            if (t >= ntask) break;
            if (t < ndyn) {
                const SymvqPiece pc = pieces[dfirst + t];
                sq_piece<2, NT, true>(Q, ld, n, pc.J, pc.ra, pc.rb, dfirst + t, g, rowpart, colpart, lcol);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) atomicAdd(&ctl->seg_done[pc.J], 1u);
            } else if (t < ndyn + nblock) {
                const int b = t - ndyn;
                if (tid == 0) {
                    const int J = (int)(((long long)b * 128) / SQ_SEG);
                    int ok = 1;
                    for (int j = 0; j <= J && ok; ++j)
                        ok = sq_wait_ge(&ctl->seg_done[j], (unsigned)(plan->seg_pfirst[j + 1] - plan->seg_pfirst[j])) ? 1 : 0;
                    sh_ok = ok;
                }
                __syncthreads();
                if (sh_ok) sq_reduce_task<NP>(b, n, plan, rowpart, colpart, g, pend, y, partial, part);
                else err = 3;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) atomicAdd(&ctl->reduce_done, 1u);   // (also after a time-out: nobody may wait forever)
            } else {
                if (tid == 0) sh_ok = sq_wait_ge(&ctl->reduce_done, (unsigned)nblock) ? 1 : 0;
                __syncthreads();
                if (sh_ok && !err)
                    sq_scalar_task<NP>(t - ndyn - nblock, n, nblock, y, xc, pend, cpend, partial, st, kappa_in, calc,
                                       cp_dev, cp_val, slot, queue_mode, q_status, q_tsq, lds, psum, bc, &bc_status);
                else err = 4;
            }
            __syncthreads();  // sh_task / sh_ok are rewritten by the next round
        }
    } else if (blockIdx.x == 0 && tid == 0 && q_status) {
        *q_status = ST_UNKNOWN;
        *q_tsq = st->tsq;
    }
    if (tid == 0) {
        if (err) atomicExch(&st->solve_err, err);
        const unsigned e = atomicAdd(&ctl->exited, 1u);
        if (e == gridDim.x - 1) {  // last one out re-arms the counters for the next launch
            const int nseg = plan->nseg;
            for (int j = 0; j < nseg; ++j) __hip_atomic_store(&ctl->seg_done[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->reduce_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Host side: cut the triangle of an unsharded n x n matrix (n even) into pieces.  Unit of work = one 1 KiB chunk load
// of a wave (a row costs ceil((min(dg, 511) + 1) / 128) chunks).  The walk (segments in order, rows top to bottom) is
// cut into SQ_RUNS static runs of equal cost covering the first `1 - dyn_share` of the work, and the rest into small
// pieces of at most `dyn_cost` chunks that whoever finishes its run pulls from the queue: workgroups with equal byte
// counts end up to 25 % apart (measured: the fastest at 130 us, the median at 153 us, the slowest at 165 us of a launch),
// the small pieces absorb that.
inline void symvq_make_plan(long long n, SymvqPlan& plan, std::vector<SymvqPiece>& pieces, double dyn_share = 0.15,
                            long long dyn_cost = 64) {
    const int nseg = (int)((n + SQ_SEG - 1) / SQ_SEG);
    auto row_cost = [&](int J, long long r) -> long long {
        const long long dg = r - (long long)J * SQ_SEG;   // >= 0
        const long long last = dg < SQ_SEG - 1 ? dg : SQ_SEG - 1;
        return last / 128 + 1;
    };
    long long total = 0;
    for (int J = 0; J < nseg; ++J)
        for (long long r = (long long)J * SQ_SEG; r < n; ++r) total += row_cost(J, r);
    const long long stat_total = (long long)((1.0 - dyn_share) * (double)total);
    plan.nseg = nseg;
    plan.nblock = (int)((n + 127) / 128);
    plan.nslice = scalar_groups(n);
    pieces.clear();
    int run = 0;
    long long done = 0;          // cost handed out so far
    bool dynamic = false;
    plan.run_first[0] = 0;
    plan.ndyn_first = -1;
    for (int J = 0; J < nseg; ++J) {
        plan.seg_pfirst[J] = (int)pieces.size();
        long long r = (long long)J * SQ_SEG;
        while (r < n) {
            const long long ra = r;
            if (!dynamic) {
                // run `run` ends when the cumulative cost reaches stat_total * (run + 1) / SQ_RUNS
                const long long target = (stat_total * (run + 1)) / SQ_RUNS;
                while (r < n && done < target) done += row_cost(J, r++);
                if (r > ra) pieces.push_back(SymvqPiece{J, (int)ra, (int)r, 0});
                if (done >= target) {
                    plan.run_first[++run] = (int)pieces.size();
                    if (run == SQ_RUNS) {
                        dynamic = true;
                        plan.ndyn_first = (int)pieces.size();
                    }
                }
            } else {
                long long c = 0;
                while (r < n && c < dyn_cost) c += row_cost(J, r++);
                pieces.push_back(SymvqPiece{J, (int)ra, (int)r, 0});
            }
        }
    }
    plan.seg_pfirst[nseg] = (int)pieces.size();
    while (run < SQ_RUNS) plan.run_first[++run] = (int)pieces.size();   // (tiny n: trailing runs are empty)
    plan.npiece = (int)pieces.size();
    if (plan.ndyn_first < 0) plan.ndyn_first = plan.npiece;
    plan.ndyn = plan.npiece - plan.ndyn_first;
}

}  // namespace ellhip
