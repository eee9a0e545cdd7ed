// symm_glds_kernel.hpp -- round-4 experiment: k_symm_mfma with the blocks of Q (and the gT rows of their columns) brought
// into a per-wave LDS ring by LDS-DMA (global_load_lds_dwordx4), D blocks deep, so that the loads of the blocks to come stay
// in flight behind the 32 MFMAs of the block in hand WITHOUT register buffers (the register pipeline spilled at two waves per
// SIMD, profiles/r04/symm_mfma_pipeline_attempt.txt).  Same MFMAs, same operands, same order as k_symm_mfma: the partial sums
// are bit-identical.
//
// Ring slot (10 KiB): 64 rows x 16 doubles of Q, the 16-byte pieces of a row XOR-swizzled by (row & 7) -- done on the SOURCE
// address, the LDS image of an LDS-DMA instruction is lane-linear -- so that both the row-major reads of the column product
// (4 rows x 128 B per instruction) and the column-major reads of the row product (16 rows x 4 columns) are conflict free;
// then 16 rows x 16 doubles of gT (linear).  A wave issues 10 LDS-DMA instructions per block and waits with a counted
// s_waitcnt vmcnt(N): N = the LDS-DMAs of the blocks issued after this one + the colpart stores in between (all count on
// the one counter, in issue order).  No barrier in the loop: every wave has its own ring.
#pragma once
#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
namespace ellhip {

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void wait_vmcnt(int n) {  // n wave-uniform; a smaller count is a stronger wait
    switch (n) {
#define ELLHIP_W(k) \
    case k:         \
        asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); \
        break;
        ELLHIP_W(0) ELLHIP_W(1) ELLHIP_W(2) ELLHIP_W(3) ELLHIP_W(4) ELLHIP_W(5) ELLHIP_W(6) ELLHIP_W(7) ELLHIP_W(8) ELLHIP_W(9)
        ELLHIP_W(10) ELLHIP_W(11) ELLHIP_W(12) ELLHIP_W(13) ELLHIP_W(14) ELLHIP_W(15) ELLHIP_W(16) ELLHIP_W(17) ELLHIP_W(18)
        ELLHIP_W(19) ELLHIP_W(20) ELLHIP_W(21) ELLHIP_W(22) ELLHIP_W(23) ELLHIP_W(24) ELLHIP_W(25) ELLHIP_W(26) ELLHIP_W(27)
        ELLHIP_W(28) ELLHIP_W(29) ELLHIP_W(30) ELLHIP_W(31) ELLHIP_W(32) ELLHIP_W(33) ELLHIP_W(34) ELLHIP_W(35) ELLHIP_W(36)
        ELLHIP_W(37) ELLHIP_W(38) ELLHIP_W(39) ELLHIP_W(40) ELLHIP_W(41) ELLHIP_W(42) ELLHIP_W(43) ELLHIP_W(44) ELLHIP_W(45)
        ELLHIP_W(46) ELLHIP_W(47) ELLHIP_W(48) ELLHIP_W(49) ELLHIP_W(50) ELLHIP_W(51) ELLHIP_W(52) ELLHIP_W(53) ELLHIP_W(54)
        ELLHIP_W(55) ELLHIP_W(56) ELLHIP_W(57) ELLHIP_W(58) ELLHIP_W(59) ELLHIP_W(60) ELLHIP_W(61) ELLHIP_W(62)
#undef ELLHIP_W
        default:
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    }
}

constexpr int SGL_SLOT = 64 * 16 + 16 * 16;  // doubles per ring slot
// diagnostic builds (STAMP): shader clock / 100 MHz reference clock around the block loop of every workgroup's wave 0
__device__ unsigned long long g_sgl_clk[8192][3];

template <int SEG, int D, int MODE = 0, bool STAMP = false>
__global__ __launch_bounds__(MODE >= 6 ? 512 : 256) void k_symm_glds(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                   long long nrows, const double* __restrict__ gT, int lv,
                                                   double* __restrict__ rowpart, double* __restrict__ colpart,
                                                   long long rowpart_stride, long long colpart_stride,
                                                   const DevState* __restrict__ st) {
    extern __shared__ double ring[];  // [4][D][SGL_SLOT]
    if (st->halted) return;
    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = wave8 & 3;
    // (6, 7: throughput probes -- some waves only load, the others only compute, nothing ties them.  6: waves 4-7 load, one
    // beside each computing wave; 7: waves 0 and 4 load (waves w and w + 4 of a workgroup share a SIMD), twice each, and the six
    // waves of the other three SIMDs compute)
    const bool loader = (MODE == 6 && wave8 >= 4) || (MODE == 7 && wave == 0);
    const int lr = lane >> 4, lc = lane & 15;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = (long long)blockIdx.y;
    const long long r0 = row0 + I * SYMV_H;
    const long long c0 = J * SEG;
    if (r0 >= row0 + nrows || c0 > r0 + SYMV_H - 1) return;
    Q -= row0 * ld;
    const bool full = c0 + SEG - 1 < r0;
    double gr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) gr[j] = gT[(r0 + 4 * j + lr) * SMM_NV + lc];
    // (the compiler must see these loads retired HERE: left to its own bookkeeping it waits for them at their first uses
    // inside the loop, vmcnt(15) .. vmcnt(0) per block, and drains the ring with them)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(gr[j]));
    double4_t dr[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dr[jj] = double4_t{0.0, 0.0, 0.0, 0.0};
    const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;
    const int nblk = (int)((cend - c0) / 16);
    const int nbw = (nblk > wave ? (nblk - wave + 3) / 4 : 0) * ((MODE == 7 && loader) ? 2 : 1);  // this wave's blocks: wave, wave + 4, ...
    const int ns = (lv + 3) / 4;                                // colpart store instructions per block
    double* myring = ring + (size_t)wave * D * SGL_SLOT;
    const unsigned my_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)myring);
    // LDS-DMA sources: instruction k of a block covers rows 8 k .. 8 k + 7; lane -> row 8 k + (lane >> 3), LDS piece
    // lane & 7, which holds the row's piece (lane & 7) ^ (row & 7)
    const int prow = lane >> 3;
    const double* qsrc = Q + (r0 + prow) * ld + 2 * ((lane & 7) ^ prow);
    const double* gsrc = gT + 2 * lane;
    auto issue = [&](int i) {
        const long long cb = c0 + 16LL * (wave + 4 * ((MODE == 7 && loader) ? (i >> 1) : i));  // (7: every block twice, in range)
        const unsigned dst = my_lds + (unsigned)((i % D) * SGL_SLOT * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) glds16(qsrc + (long long)(8 * k) * ld + cb, dst + k * 1024);
#pragma unroll
        for (int k = 0; k < 2; ++k) glds16(gsrc + (cb + 8 * k) * SMM_NV, dst + 8192 + k * 1024);
    };
    unsigned long long t0 = 0, r0t = 0;
    if (STAMP) {
        t0 = __builtin_amdgcn_s_memtime();
        r0t = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    if ((MODE != 2 && MODE != 5 && MODE < 6) || loader)
        for (int i = 0; i < D - 1 && i < nbw; ++i) issue(i);
    // the blocks left of the diagonal: one straight-line body (the diagonal's four blocks are the LAST block of each wave)
    const int ndiag = (!full && nbw > 0) ? 1 : 0;
    auto arrive = [&](int i) {
        if (((MODE != 2 && MODE != 5 && MODE < 6) || loader) && i + D - 1 < nbw) issue(i + D - 1);  // into the slot block i - 1 has left
        const int ahead = (nbw - 1 - i < D - 1) ? nbw - 1 - i : D - 1;
        const int behind = (i < D - 1) ? i : D - 1;
        if (MODE < 2) wait_vmcnt(__builtin_amdgcn_readfirstlane(10 * ahead + ns * behind));
        if (loader) wait_vmcnt(__builtin_amdgcn_readfirstlane(10 * ahead));  // (2: no loads; 3: loads never waited for)
    };
    auto store_cols = [&](const double4_t& dc, long long cb) {
        const double o[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * q < lv) {  // (wave-uniform: exactly ns store instructions per block)
                const int v = lr + 4 * q;
                if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[q];
            }
        }
    };
    // LDS offsets of this lane's operands inside a slot (doubles)
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
    for (int i = 0; i < nbw - ndiag; ++i) {
        arrive(i);
        const long long cb = c0 + 16LL * (wave + 4 * i);
        const double* blk = myring + (i % D) * SGL_SLOT;
        if (MODE == 1 || loader) {  // the loading skeleton alone: one LDS read per block, no matrix-core work
            dr[0].x += blk[lane];
            continue;
        }
        double4_t dc = {0.0, 0.0, 0.0, 0.0};
        double gc[4], xv[16], tv[16];
        // every operand of the block first (36 LDS reads in flight), then the 32 MFMAs: left alone the compiler issues read,
        // wait, MFMA one by one through a single register pair
        if (MODE == 5) {  // no LDS reads either: the MFMAs, the stores and the loop
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[kb] = gr[kb];
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = gr[15 - j], tv[j] = gr[j] + 1.0;
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[kb] = blk[goff[kb]];
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = blk[xoff[j]];
#pragma unroll
            for (int k = 0; k < 16; ++k) tv[k] = blk[toff[k]];
        }
        // chains on ONE accumulator back to back: the pipe forwards the accumulator (68 cycles per MFMA); alternating
        // accumulators costs 74-83 (tools/experiments/mfma_f64_rate.hip), and the scheduler interleaves them if it may
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
        store_cols(dc, cb);
    }
    if (ndiag) {
        const int i = nbw - 1;
        arrive(i);
        const long long cb = c0 + 16LL * (wave + 4 * i);
        const double* blk = myring + (i % D) * SGL_SLOT;
        if (MODE == 1 || loader) {
            dr[0].x += blk[lane];
        } else {
            double4_t dc = {0.0, 0.0, 0.0, 0.0};
            double gc[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[kb] = blk[goff[kb]];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long r = r0 + 4 * j + lr, c = cb + lc;
                const double x = blk[xoff[j]];
                dc = __builtin_amdgcn_mfma_f64_16x16x4f64(gr[j], (c < r) ? x : 0.0, dc, 0, 0, 0);  // strictly below the diagonal
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const long long r = r0 + 16 * jj + lc, c = cb + 4 * kb + lr;
                    const double t = blk[toff[4 * jj + kb]];
                    dr[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[kb], (c <= r) ? t : 0.0, dr[jj], 0, 0, 0);  // the diagonal counts here
                }
            store_cols(dc, cb);
        }
    }
    if (STAMP) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1t = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == (MODE == 7 ? 64 : 0) && wg < 8192) {
            g_sgl_clk[wg][0] = t1 - t0;
            g_sgl_clk[wg][1] = r1t - r0t;
            g_sgl_clk[wg][2] = (unsigned long long)nbw;
        }
    }
    __syncthreads();
    double* red = ring;  // [wave][jj][i][lane]: 32 KiB of the ring
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const double o[4] = {dr[jj].x, dr[jj].y, dr[jj].z, dr[jj].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((wave * 4 + jj) * 4 + i) * 64 + lane] = o[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = lr + 4 * i;
        const int jj = wave;
        const double s0 = red[((0 * 4 + jj) * 4 + i) * 64 + lane], s1 = red[((1 * 4 + jj) * 4 + i) * 64 + lane];
        const double s2 = red[((2 * 4 + jj) * 4 + i) * 64 + lane], s3 = red[((3 * 4 + jj) * 4 + i) * 64 + lane];
        if (v < lv) rowpart[(long long)v * rowpart_stride + J * n + r0 + 16 * jj + lc] = ((s0 + s1) + s2) + s3;
    }
}

}  // namespace ellhip
