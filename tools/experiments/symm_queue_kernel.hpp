// symm_queue_kernel.hpp -- round-4 experiment: k_symm_mfma's tiles handed out from a queue.
//
// What the stamps of symm_glds.hip showed: a 64 x 2048 tile keeps its workgroup ~100 us, the card holds 512 of them (two
// per CU), the lower triangle of n = 16384 is 1152 tiles (1028 full ones' worth of work): the pass lasts THREE workgroup
// lifetimes where the work is 2.0 -- the third round runs a quarter of the slots.  Here a launch is 2 workgroups per CU
// that draw tiles from a counter, largest first (host-built table), until none is left: the pass lasts work / slots plus
// the last (smallest) tiles.
//
// Two bodies: `reg` = the production k_symm_mfma block loop (register loads, transposition through an LDS patch) and
// `glds` = the LDS-DMA ring of symm_glds_kernel.hpp.  Both give k_symm_mfma's partial sums to the bit.
#pragma once
#include "symm_glds_kernel.hpp"
namespace ellhip {

// (SymmTile: ell_kernels.hpp -- the queue form went into the product as k_symm_mfma_q)

// the next tile of the queue for the whole workgroup, or -1 (s_t: one int of LDS)
__device__ __forceinline__ int symm_next_tile(unsigned* counter, int ntiles, int* s_t) {
    __syncthreads();  // (everybody has read the previous value, and is done with the LDS the body used)
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(counter, 1u);
        *s_t = t < (unsigned)ntiles ? (int)t : -1;
    }
    __syncthreads();
    return *s_t;
}

template <bool NT, int SEG>
__global__ __launch_bounds__(256) void k_symm_q_reg(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                    long long nrows, const double* __restrict__ gT, int lv,
                                                    double* __restrict__ rowpart, double* __restrict__ colpart,
                                                    long long rowpart_stride, long long colpart_stride,
                                                    const DevState* __restrict__ st, const SymmTile* __restrict__ tiles,
                                                    int ntiles, unsigned* __restrict__ counter, int budget = 1 << 30) {
    __shared__ double sh[4][SYMV_H * SMM_PITCH];
    __shared__ int s_t;
    if (st->halted) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    Q -= row0 * ld;
    for (int done_tiles = 0; done_tiles < budget; ++done_tiles) {
        const int t = symm_next_tile(counter, ntiles, &s_t);
        if (t < 0) break;
        const long long I = tiles[t].I, J = tiles[t].J;
        const long long r0 = row0 + I * SYMV_H;
        const long long c0 = J * SEG;
        const bool full = c0 + SEG - 1 < r0;
        double gr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) gr[j] = gT[(r0 + 4 * j + lr) * SMM_NV + lc];
        double4_t dr[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dr[jj] = double4_t{0.0, 0.0, 0.0, 0.0};
        const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;
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
            const bool diag = !full && cb + 15 >= r0;
            double4_t dc = {0.0, 0.0, 0.0, 0.0};
            if (diag) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const long long r = r0 + 4 * j + lr, c = cb + lc;
                    const double below = (c < r) ? x[j] : 0.0;
                    dc = __builtin_amdgcn_mfma_f64_16x16x4f64(gr[j], below, dc, 0, 0, 0);
                    x[j] = (c <= r) ? x[j] : 0.0;
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
                    const double tt = mysh[(16 * jj + lc) * SMM_PITCH + 4 * kb + lr];
                    dr[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[kb], tt, dr[jj], 0, 0, 0);
                }
            const double o[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = lr + 4 * i;
                if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[i];
            }
        }
        __syncthreads();
        double* red = &sh[0][0];
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
}

template <int SEG, int D>
__global__ __launch_bounds__(256) void k_symm_q_glds(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                                     long long nrows, const double* __restrict__ gT, int lv,
                                                     double* __restrict__ rowpart, double* __restrict__ colpart,
                                                     long long rowpart_stride, long long colpart_stride,
                                                     const DevState* __restrict__ st, const SymmTile* __restrict__ tiles,
                                                     int ntiles, unsigned* __restrict__ counter) {
    extern __shared__ double ring[];  // [4][D][SGL_SLOT]: D = 2 is 80 KiB, two workgroups fill a CU's LDS to the byte
    if (st->halted) return;
    int* s_t = reinterpret_cast<int*>(ring);  // (the tile index passes through the idle ring; symm_next_tile's barriers + one more)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane >> 4, lc = lane & 15;
    Q -= row0 * ld;
    const int ns = (lv + 3) / 4;  // colpart store instructions per block
    double* myring = ring + (size_t)wave * D * SGL_SLOT;
    const unsigned my_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)myring);
    const int prow = lane >> 3;
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
    for (;;) {
        const int t = symm_next_tile(counter, ntiles, s_t);
        __syncthreads();  // (everybody has the index before the first LDS-DMA may land on it)
        if (t < 0) break;
        const long long I = tiles[t].I, J = tiles[t].J;
        const long long r0 = row0 + I * SYMV_H;
        const long long c0 = J * SEG;
        const bool full = c0 + SEG - 1 < r0;
        double gr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) gr[j] = gT[(r0 + 4 * j + lr) * SMM_NV + lc];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see symm_glds_kernel.hpp
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(gr[j]));
        double4_t dr[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dr[jj] = double4_t{0.0, 0.0, 0.0, 0.0};
        const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;
        const int nblk = (int)((cend - c0) / 16);
        const int nbw = nblk > wave ? (nblk - wave + 3) / 4 : 0;
        const double* qsrc = Q + (r0 + prow) * ld + 2 * ((lane & 7) ^ prow);
        const double* gsrc = gT + 2 * lane;
        auto issue = [&](int i) {
            const long long cb = c0 + 16LL * (wave + 4 * i);
            const unsigned dst = my_lds + (unsigned)((i % D) * SGL_SLOT * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) glds16(qsrc + (long long)(8 * k) * ld + cb, dst + k * 1024);
#pragma unroll
            for (int k = 0; k < 2; ++k) glds16(gsrc + (cb + 8 * k) * SMM_NV, dst + 8192 + k * 1024);
        };
        for (int i = 0; i < D - 1 && i < nbw; ++i) issue(i);
        const int ndiag = (!full && nbw > 0) ? 1 : 0;
        auto arrive = [&](int i) {
            if (i + D - 1 < nbw) issue(i + D - 1);
            const int ahead = (nbw - 1 - i < D - 1) ? nbw - 1 - i : D - 1;
            const int behind = (i < D - 1) ? i : D - 1;
            wait_vmcnt(__builtin_amdgcn_readfirstlane(10 * ahead + ns * behind));
        };
        auto store_cols = [&](const double4_t& dc, long long cb) {
            const double o[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (4 * q < lv) {
                    const int v = lr + 4 * q;
                    if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[q];
                }
            }
        };
        for (int i = 0; i < nbw - ndiag; ++i) {
            arrive(i);
            const long long cb = c0 + 16LL * (wave + 4 * i);
            const double* blk = myring + (i % D) * SGL_SLOT;
            double4_t dc = {0.0, 0.0, 0.0, 0.0};
            double gc[4], xv[16], tv[16];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[kb] = blk[goff[kb]];
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = blk[xoff[j]];
#pragma unroll
            for (int k = 0; k < 16; ++k) tv[k] = blk[toff[k]];
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
            double4_t dc = {0.0, 0.0, 0.0, 0.0};
            double gc[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[kb] = blk[goff[kb]];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long r = r0 + 4 * j + lr, c = cb + lc;
                const double x = blk[xoff[j]];
                dc = __builtin_amdgcn_mfma_f64_16x16x4f64(gr[j], (c < r) ? x : 0.0, dc, 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const long long r = r0 + 16 * jj + lc, c = cb + 4 * kb + lr;
                    const double tt = blk[toff[4 * jj + kb]];
                    dr[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[kb], (c <= r) ? tt : 0.0, dr[jj], 0, 0, 0);
                }
            store_cols(dc, cb);
        }
        __syncthreads();
        double* red = ring;
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
}

}  // namespace ellhip
