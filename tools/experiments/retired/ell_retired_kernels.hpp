// ell_retired_kernels.hpp -- kernels that were built, tested bit-identical and MEASURED SLOWER than the production
// path in round 2, moved out of libellhip.so in round 3 together with their host plumbing (last
// tree that had them wired in: commit 8411b35).  Kept compilable against the product headers for reference:
//   k_symv_tail            lower-triangle GEMV tiles + their reduction in one launch              (-10 %)
//   k_update_fused_def     GEMV pass + scalar stage in one launch, recorded full-row schedule     (22 800 vs 25 400 updates/s at n = 4096)
//   k_symv_reduce_scalar   reduction of the lower-triangle GEMV + scalar stage in one launch      (4190 vs 4230 updates/s at n = 16384)
#pragma once

#include "../../../ellalgo-rs_amd/csrc/ell_kernels.hpp"

namespace ellhip {

// ------------------------------------------------------------------------------- k_symv_tail ---
// k_symv and k_symv_reduce<NP> in ONE launch (unsharded handle): a workgroup that has finished its tile counts it done
// for its segment and -- if it belongs to the last workgroups the device was given -- pulls reduce tasks (128 columns
// each, symv_reduce_block) from a queue: a task waits, bounded, until every tile of the segments 0 .. J(task) is counted
// (segment-major dispatch order makes that a prefix condition; a task needs row sums from the segments left of its own
// and column sums from its own).  One launch and one kernel boundary fewer on every update's dependency chain, and the
// reductions of the early segments run in the tile phase's tail.  Same arithmetic in the same order as the two-launch
// form: identical bits.  Hand-off as in cdna_hip_programming.md Guideline 16 (write-through stores, storing waves drain,
// barrier, one lane adds to the counter; one lane polls, barrier, agent-scope loads).  Workgroups that wait hold their
// slots and poll: only the last `npull` workgroups to FINISH their tile pull tasks (a finishing ticket tells), so the
// waits are short and cannot starve tiles that still wait for a slot.  The last workgroup out re-arms the counters.
constexpr int SYMV_MAXSEGS = 64;
struct SymvTailCtl {
    unsigned next, exited, finished, pad1;
    unsigned seg_done[SYMV_MAXSEGS];
};
__device__ __forceinline__ bool symv_wait_ge(const unsigned* ctr, unsigned want) {  // ONE lane; bounded (~0.5 s)
    for (int spin = 0; spin < (1 << 19); ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(32);   // ~1 us between polls: a hundred pollers at 60 ns slowed the tiles still streaming
    }
    return false;
}

template <int RW, bool NT, int NP>
__global__ __launch_bounds__(256, RW == 2 ? 5 : 4) /* RW = 2: 96 VGPRs, the residency of k_symv's tile phase */ void k_symv_tail(const double* __restrict__ Q, long long ld, long long n,
                                                   const double* __restrict__ g, double* __restrict__ rowpart,
                                                   double* __restrict__ colpart, double* __restrict__ y,
                                                   const double* __restrict__ pend, double* __restrict__ partial,
                                                   DevState* __restrict__ st, SymvTailCtl* __restrict__ ctl,
                                                   unsigned nactive, unsigned npull) {
    __shared__ double red[4][SYMV_H];
    __shared__ double2_t part[4][64];
    __shared__ int sh_task, sh_ok;
    __shared__ unsigned sh_ticket;
    const int halted = st->halted;
    if (NP > 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        st->halted_in = halted;   // snapshots for k_scalar_apply_def (as k_symv_reduce<NP> takes them)
        st->kappa_in = st->kappa;
    }
    if (halted) return;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = blockIdx.y;
    if (!symv_tile<RW, NT, 0, SYMV_SEG, true>(Q, ld, n, 0, n, g, rowpart, colpart, I, J, red)) return;  // (not counted)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid == 0) {
        atomicAdd(&ctl->seg_done[J], 1u);
        sh_ticket = atomicAdd(&ctl->finished, 1u);   // this workgroup is the (ticket + 1)-th to finish its tile
    }
    __syncthreads();
    // Only the LAST npull workgroups to finish pull tasks: whoever finishes early would poll the counters for most of
    // the launch (128 early finishers polling made the whole launch 1.7x slower: 0.32 ms against 0.19 + 0.012 ms), the
    // last ones wait microseconds; and however many rounds the grid needs, they belong to its last one.
    if (sh_ticket + npull >= nactive) {
        const int nblock = (int)((n + 127) / 128);
        const long long nstrips = (n + SYMV_H - 1) / SYMV_H;
        for (;;) {
            if (tid == 0) sh_task = (int)atomicAdd(&ctl->next, 1u);
            __syncthreads();
            const int t = sh_task;
            if (t >= nblock) break;
            if (tid == 0) {
                const long long Jt = ((long long)t * 128) / SYMV_SEG;
                int ok = 1;
                for (long long j = 0; j <= Jt && ok; ++j)   // tiles of segment j: the strips that reach its first column
                    ok = symv_wait_ge(&ctl->seg_done[j], (unsigned)(nstrips - (j * SYMV_SEG) / SYMV_H)) ? 1 : 0;
                sh_ok = ok;
            }
            __syncthreads();
            if (sh_ok)
                symv_reduce_block<NP, true>(t, n, 0, n, SYMV_SEG, rowpart, colpart, y, g, pend, partial, part);
            else if (tid == 0)
                atomicExch(&st->solve_err, 5);
            __syncthreads();  // sh_task / sh_ok / part are rewritten by the next round
        }
    }
    if (tid == 0) {
        const unsigned e = atomicAdd(&ctl->exited, 1u);
        if (e == nactive - 1) {  // last one out re-arms the counters for the next launch
            const int nseg = (int)((n + SYMV_SEG - 1) / SYMV_SEG);
            for (int j = 0; j < nseg; ++j) __hip_atomic_store(&ctl->seg_done[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ONE launch per update on the recorded full-row schedule (unsharded handle, depth 8, n < 8192 or odd n): the GEMV
// pass of k_sweep_gemv_dots AND the scalar stage k_scalar_apply_def<NP, true> -- `Ell::update_core` (src/ell.rs:97-137)
// up to the recorded shrink in a single kernel, as SURVEY section 7's k_ell_fused asks.  The row tiles run first (lower
// block indices); the last scalar_groups(n) workgroups form the v_j . g partial sums beside them exactly as before, then
// wait until every workgroup of the launch has ARRIVED (one counter that only grows: `target` = launches so far x grid
// size; arrival = barrier, agent-scope release fence by one thread, atomic add; the waiters poll with one lane, then an
// acquire fence) and run the scalar stage's body on the fresh y.  Nobody else waits, so the grid need not be resident
// (a workgroup waits only for workgroups dispatched before it), and the wait is bounded (DevState.solve_err = 7).
// Same code on the same data in the same order as the two launches: identical bits.  What it saves is the second
// launch's latency on the update's dependency chain: n = 4096 24.6 + 11.5 us -> see DESIGN.md section 5.1.
constexpr int FUSED_WAIT_ERR = 7;
template <int RW, int UNR, int VEC, bool NT, int NP>
__global__ __launch_bounds__(256) void k_update_fused_def(const double* Q, long long ld, long long n, long long nrows,
                                                          long long row0, const double* __restrict__ gvec,
                                                          double* __restrict__ gv_out, DevState* __restrict__ st,
                                                          int reverse, unsigned ntiles, double* __restrict__ pend,
                                                          double* __restrict__ partial, double* __restrict__ xc,
                                                          double* __restrict__ cpend, EllCalcDev calc,
                                                          const CutParams* __restrict__ cp_dev, CutParams cp_val, int slot,
                                                          int queue_mode, int* __restrict__ q_status,
                                                          double* __restrict__ q_tsq, unsigned* __restrict__ arrived,
                                                          unsigned target) {
    __shared__ double red[4][RW > NP ? RW : NP];
    __shared__ int wait_ok;
    const int halted = st->halted;
    const int tid = threadIdx.x;
    if (blockIdx.x >= ntiles) {
        const long long b = (long long)blockIdx.x - ntiles;
        if (b == 0 && tid == 0) {  // (write-through, like everything a waiter reads after the arrivals)
            __hip_atomic_store(&st->halted_in, halted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->kappa_in, st->kappa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!halted) {
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
            if (tid < NP)
                __hip_atomic_store(&partial[b * (NP + 1) + 1 + tid], ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores above have landed
        __syncthreads();
        if (tid == 0) {
            atomicAdd(arrived, 1u);
            int ok = 0;
            for (int spin = 0; spin < (1 << 22); ++spin) {
                const unsigned a = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(a - target) >= 0) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // y and the partial sums of the others
            if (!ok) atomicExch(&st->solve_err, FUSED_WAIT_ERR);
            wait_ok = ok;
        }
        __syncthreads();
        if (!wait_ok) return;
        const long long m_sl = scalar_slice(n), lo_sl = b * m_sl;
        scalar_apply_def_body<NP, true>(b, lo_sl, (lo_sl + m_sl < n) ? lo_sl + m_sl : n, n, gv_out - row0, xc, pend, cpend,
                                        partial, st, calc, cp_dev, cp_val, slot, queue_mode, q_status, q_tsq,
                                        (int)scalar_groups(n), gvec);
        return;
    }
    if (!halted) {
        const long long tile = reverse ? (long long)ntiles - 1 - blockIdx.x : (long long)blockIdx.x;
        const long long row_base = tile * RW;
        if (row_base < nrows)
            sweep_rows<RW, UNR, VEC, NT, false, true, false>(Q, const_cast<double*>(Q), ld, n, nrows, row0, row_base, nullptr,
                                                             gvec, gv_out, 0.0, 1.0,
                                                             reinterpret_cast<double(*)[RW]>(&red[0][0]));
    }
    // The tile's rows of y once more, WRITE-THROUGH (sweep_rows stored them with plain stores; its wave sums are still in
    // `red`): a release fence here would write back the XCD's whole L2 in every one of the ~500 tile workgroups (measured:
    // 69 us per launch instead of 25 + 12).
    if (!halted && tid < RW) {
        const long long tile = reverse ? (long long)ntiles - 1 - blockIdx.x : (long long)blockIdx.x;
        const long long r = tile * RW + tid;
        double(*rr)[RW] = reinterpret_cast<double(*)[RW]>(&red[0][0]);  // sweep_rows' view of the buffer
        if (r < nrows)
            __hip_atomic_store(&gv_out[r], ((rr[0][tid] + rr[1][tid]) + rr[2][tid]) + rr[3][tid], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) atomicAdd(arrived, 1u);
}

// k_symv_reduce<NP> AND the scalar stage in ONE launch (unsharded lower-triangle schedule): every workgroup reduces its
// 128 columns of y and its share of the dot products (write-through), arrives, waits until all ceil(n/128) workgroups
// have (they do the same amount of work and are all resident, so the wait is short; bounded), then forms omega / the
// coefficients redundantly from the same partial sums in the same order as k_scalar_apply_def -- identical bits -- and
// updates ITS OWN 128 elements (gt, the recorded vector, xc).  One launch and one kernel boundary fewer on the
// update's dependency chain.
template <int NP>
__global__ __launch_bounds__(256) void k_symv_reduce_scalar(long long n, long long seg, const double* __restrict__ rowpart,
                                                            const double* __restrict__ colpart, double* __restrict__ y,
                                                            DevState* __restrict__ st, const double* __restrict__ g,
                                                            double* __restrict__ pend, double* __restrict__ partial,
                                                            double* __restrict__ xc, double* __restrict__ cpend,
                                                            EllCalcDev calc, const CutParams* __restrict__ cp_dev,
                                                            CutParams cp_val, int slot, int queue_mode,
                                                            int* __restrict__ q_status, double* __restrict__ q_tsq,
                                                            unsigned* __restrict__ arrived, unsigned target) {
    __shared__ double2_t part[4][64];
    __shared__ int wait_ok;
    const int halted = st->halted;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) {  // the snapshots the stage's body reads (its lead workgroup rewrites kappa / halted)
        __hip_atomic_store(&st->halted_in, halted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->kappa_in, st->kappa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!halted)
        symv_reduce_block<NP, false, true>((long long)blockIdx.x, n, 0, n, seg, rowpart, colpart, y, g, pend, partial, part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have landed
    __syncthreads();
    if (tid == 0) {
        atomicAdd(arrived, 1u);
        int ok = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            const unsigned a = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(a - target) >= 0) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the others' partial sums
        if (!ok) atomicExch(&st->solve_err, FUSED_WAIT_ERR);
        wait_ok = ok;
    }
    __syncthreads();
    if (!wait_ok) return;
    const long long lo = (long long)blockIdx.x * 128;
    scalar_apply_def_body<NP, false>((long long)blockIdx.x, lo, (lo + 128 < n) ? lo + 128 : n, n, y, xc, pend, cpend, partial,
                                     st, calc, cp_dev, cp_val, slot, queue_mode, q_status, q_tsq, (int)gridDim.x, nullptr);
}

}  // namespace ellhip
