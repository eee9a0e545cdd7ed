// symm_lc_kernel.hpp -- round-4 experiment: k_symm_mfma with LOADER and CONSUMER waves.
//
// What symm_glds.hip's probes showed (profiles/r04/symm_mfma_overlap_probes.txt): on a SIMD that is executing f64 MFMAs every
// vector-memory instruction issued by ANY wave of that SIMD costs the matrix pipe ~70 cycles (2585 -> 3307 cycles per block with
// ten LDS-DMA instructions per block beside it), so a kernel whose waves both load and multiply lasts the SUM of the two phases
// however the loads are buffered; MFMA waves on SIMDs that issue no loads keep their pace (4184 vs 4164).  Waves w and w + 4 of
// a workgroup share a SIMD.  So: 8 waves, waves 0 and 4 (one SIMD) only move bytes -- LDS-DMA of 64 x 16 blocks of Q (+ the gT
// rows of their columns) into a 12-slot ring -- and the six waves of the other three SIMDs only read LDS and multiply.
//
// Hand-over per slot, single writer each way: ready[s] = uses landed (loader, after s_waitcnt vmcnt), done[s] = uses read out
// (consumer, after lgkmcnt(0) on its operand reads).  Block b of a tile lives in slot b % 12, belongs to loader b % 2 and to
// consumer b % 6.  Tiles come from a queue (largest first); the six consumers' row sums are added in wave order.
// colpart is bit-identical to k_symm_mfma's, rowpart sums six partials instead of four (~1e-16 relative).
#pragma once
#include "symm_queue_kernel.hpp"
namespace ellhip {

constexpr int SLC_SLOTS = 12;
constexpr int SLC_NCONS = 6;
constexpr int SLC_AHEAD = 5;  // blocks a loader keeps in flight (<= 6: its slots; 10 LDS-DMAs each, vmcnt counts to 63)
constexpr int SLC_SPIN_MAX = 1 << 22;
constexpr size_t SLC_LDS_BYTES = (size_t)SLC_SLOTS * SGL_SLOT * sizeof(double) + 2 * SLC_SLOTS * sizeof(int) + 2 * sizeof(int);

// flags through LDS instructions proper: a volatile access through a generic pointer compiles to flat_load / flat_store, which
// count on vmcnt as well and complete out of order -- the loader's counted waits would no longer say which block has landed
typedef __attribute__((address_space(3))) volatile int lds_vint_t;
__device__ __forceinline__ lds_vint_t* lds_flag_ptr(int* p) { return (lds_vint_t*)(unsigned)(size_t)p; }
__device__ __forceinline__ int lds_poll(int* p) { return *lds_flag_ptr(p); }
__device__ __forceinline__ void lds_post(int* p, int v) { *lds_flag_ptr(p) = v; }

template <int SEG>
__global__ __launch_bounds__(512) void k_symm_lc(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                 long long nrows, const double* __restrict__ gT, int lv,
                                                 double* __restrict__ rowpart, double* __restrict__ colpart,
                                                 long long rowpart_stride, long long colpart_stride,
                                                 const DevState* __restrict__ st, const SymmTile* __restrict__ tiles, int ntiles,
                                                 unsigned* __restrict__ counter, int* __restrict__ err) {
    extern __shared__ double ring[];  // [12][SGL_SLOT] | ready[12] | done[12] | tile | fail
    if (st->halted) return;
    int* ready = reinterpret_cast<int*>(ring + (size_t)SLC_SLOTS * SGL_SLOT);
    int* done = ready + SLC_SLOTS;
    int* s_t = done + SLC_SLOTS;
    int* s_fail = s_t + 1;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = (wave8 & 3) == 0;
    const int lidx = wave8 >> 2;                                   // loader 0 / 1
    const int cidx = (wave8 & 3) - 1 + 3 * (wave8 >> 2);           // consumer 0 .. 5 (waves 1 2 3 5 6 7)
    const int lr = lane >> 4, lc = lane & 15;
    Q -= row0 * ld;
    if (threadIdx.x < 2 * SLC_SLOTS) ready[threadIdx.x] = 0;  // (ready and done are adjacent)
    if (threadIdx.x == 0) *s_fail = 0;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);
    int gbase = 0;  // generations handed out to earlier tiles (12 blocks = one generation of every slot)
    // consumer: LDS offsets of this lane's operands inside a slot (doubles)
    int xoff[16], toff[16], goff[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = 4 * j + lr;
        xoff[j] = row * 16 + ((((lc >> 1) ^ (row & 7)) << 1) | (lc & 1));
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int row = 16 * jj + lc, col = 4 * kb + lr;
            toff[4 * jj + kb] = row * 16 + ((((col >> 1) ^ (row & 7)) << 1) | (col & 1));
        }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) goff[kb] = 1024 + (4 * kb + lr) * 16 + lc;
    const int prow = lane >> 3;
    for (;;) {
        const int t = symm_next_tile(counter, ntiles, s_t);
        if (t < 0) break;
        const long long I = tiles[t].I, J = tiles[t].J;
        const long long r0 = row0 + I * SYMV_H;
        const long long c0 = J * SEG;
        const bool full = c0 + SEG - 1 < r0;
        const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;
        const int nblk = (int)((cend - c0) / 16);
        double4_t dr[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dr[jj] = double4_t{0.0, 0.0, 0.0, 0.0};
        if (loader) {
            const int nbl = nblk > lidx ? (nblk - lidx + 1) / 2 : 0;  // blocks lidx, lidx + 2, ...
            const double* qsrc = Q + (r0 + prow) * ld + 2 * ((lane & 7) ^ prow);
            const double* gsrc = gT + 2 * lane;
            bool failed = false;
            for (int idx = 0; idx < nbl + SLC_AHEAD && !failed; ++idx) {
                if (idx < nbl) {
                    const int b = lidx + 2 * idx;
                    const int s = b % SLC_SLOTS;
                    if (b >= SLC_SLOTS) {  // (the first twelve blocks of a tile find the ring idle: the barrier at the tile's end)
                        const int want = gbase + b / SLC_SLOTS;  // the generation of block b - 12
                        int spins = 0;
                        while (lds_poll(done + s) < want) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SLC_SPIN_MAX) {
                                failed = true;
                                break;
                            }
                        }
                        if (failed) break;
                    }
                    const long long cb = c0 + 16LL * b;
                    const unsigned dst = ring_lds + (unsigned)(s * SGL_SLOT * 8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) glds16(qsrc + (long long)(8 * q) * ld + cb, dst + q * 1024);
#pragma unroll
                    for (int q = 0; q < 2; ++q) glds16(gsrc + (cb + 8 * q) * SMM_NV, dst + 8192 + q * 1024);
                }
                if (idx >= SLC_AHEAD) {
                    const int j = idx - SLC_AHEAD;  // this block has to have landed
                    const int issued = idx + 1 < nbl ? idx + 1 : nbl;
                    wait_vmcnt(__builtin_amdgcn_readfirstlane(10 * (issued - (j + 1))));
                    const int bj = lidx + 2 * j;
                    if (lane == 0) lds_post(ready + bj % SLC_SLOTS, gbase + bj / SLC_SLOTS + 1);
                }
            }
            if (failed && lane == 0) lds_post(s_fail, 1);
        } else {
            double gr[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) gr[j] = gT[(r0 + 4 * j + lr) * SMM_NV + lc];
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) HERE, not at the first uses inside the loop (behind the block's stores)
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(gr[j]));
            bool failed = false;
            for (int b = cidx; b < nblk && !failed; b += SLC_NCONS) {
                const int s = b % SLC_SLOTS;
                const int gen = gbase + b / SLC_SLOTS + 1;
                int spins = 0;
                while (lds_poll(ready + s) < gen) {
                    __builtin_amdgcn_s_sleep(1);
                    if (lds_poll(s_fail) || ++spins > SLC_SPIN_MAX) {
                        failed = true;
                        break;
                    }
                }
                if (failed) break;
                const long long cb = c0 + 16LL * b;
                const double* blk = ring + (size_t)s * SGL_SLOT;
                double gc[4], xv[16], tv[16];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) gc[kb] = blk[goff[kb]];
#pragma unroll
                for (int j = 0; j < 16; ++j) xv[j] = blk[xoff[j]];
#pragma unroll
                for (int q = 0; q < 16; ++q) tv[q] = blk[toff[q]];
                // the slot goes back to the loader as soon as the operands are in registers
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(xv[j]), "+v"(tv[j]));
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) asm volatile("" : "+v"(gc[kb]));
                if (lane == 0) lds_post(done + s, gen);
                const bool diag = !full && cb + 15 >= r0;
                if (diag) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const long long r = r0 + 4 * j + lr, c = cb + lc;
                        xv[j] = (c < r) ? xv[j] : 0.0;  // column sums: strictly below the diagonal
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const long long r = r0 + 16 * jj + lc, c = cb + 4 * kb + lr;
                            tv[4 * jj + kb] = (c <= r) ? tv[4 * jj + kb] : 0.0;  // row sums: the diagonal counts here
                        }
                }
                double4_t dc = {0.0, 0.0, 0.0, 0.0};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) dc = __builtin_amdgcn_mfma_f64_16x16x4f64(gr[j], xv[j], dc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) dr[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[kb], tv[4 * jj + kb], dr[jj], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const double o[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int v = lr + 4 * q;
                    if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[q];
                }
            }
            if (failed && lane == 0) lds_post(s_fail, 1);
        }
        gbase += (nblk + SLC_SLOTS - 1) / SLC_SLOTS;
        __syncthreads();  // every block of the tile has been read out of the ring
        double* red = ring;  // [consumer][jj][i][lane]: 6 * 8 KiB
        if (!loader) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const double o[4] = {dr[jj].x, dr[jj].y, dr[jj].z, dr[jj].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((cidx * 4 + jj) * 4 + q) * 64 + lane] = o[q];
            }
        }
        __syncthreads();
        // 64 rows x 16 vectors = 1024 sums of six partials, two per thread; thread -> (jj, q, lane) as red's layout
        for (int e = threadIdx.x; e < 1024; e += 512) {
            const int jj = e >> 8, q = (e >> 6) & 3, ln = e & 63;
            const int v = (ln >> 4) + 4 * q;
            double sum = red[((0 * 4 + jj) * 4 + q) * 64 + ln];
#pragma unroll
            for (int w = 1; w < SLC_NCONS; ++w) sum += red[((w * 4 + jj) * 4 + q) * 64 + ln];
            if (v < lv) rowpart[(long long)v * rowpart_stride + J * n + r0 + 16 * jj + (ln & 15)] = sum;
        }
    }
    if (threadIdx.x == 0 && lds_poll(s_fail)) atomicExch(err, 1);
}

}  // namespace ellhip
