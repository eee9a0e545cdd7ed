// symm_split_kernel.hpp -- round-3 experiment, measured slower than k_symm_mfma (0.42 against 0.32 ms at n = 16384, 16
// gradients; profiles/r03/symm_mfma_variants_n16384.txt): the column and the row product on different waves.  The register
// allocation of a kernel is the maximum over its branches, so the split bought no occupancy, and only two of the four waves
// issue the loads.  Included by tools/experiments/symv_multi.hip.
#pragma once
#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
namespace ellhip {
// k_symm_mfma with the two products on different waves: waves 0 / 1 load the blocks, form the COLUMN product and park
// the block in LDS; waves 2 / 3 read it transposed and form the ROW product -- one workgroup barrier per round of two
// blocks, two patches per loading wave (the row waves read round r while the column waves fill round r + 1).  Each wave's
// instruction stream holds one MFMA chain and half of the registers, so three to four workgroups fit a CU and their
// phases interleave.  Same operands and per-product summation order as k_symm_mfma: the column sums are bit-identical,
// the row sums add the waves' partial sums in another order (two row waves instead of four).
template <bool NT, int SEG>
__global__ __launch_bounds__(256) void k_symm_mfma_split(const double* __restrict__ Q, long long ld, long long n,
                                                         const double* __restrict__ gT, int lv,
                                                         double* __restrict__ rowpart, double* __restrict__ colpart,
                                                         long long rowpart_stride, long long colpart_stride,
                                                         const DevState* __restrict__ st) {
    __shared__ double sh[2][2][SYMV_H * SMM_PITCH];  // [loading wave][round parity]
    if (st->halted) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = (long long)blockIdx.y;
    const long long r0 = I * SYMV_H;
    const long long c0 = J * SEG;
    if (r0 >= n || c0 > r0 + SYMV_H - 1) return;
    const bool full = c0 + SEG - 1 < r0;
    const long long cend = (c0 + SEG < r0 + SYMV_H) ? c0 + SEG : r0 + SYMV_H;
    const int nblk = (int)((cend - c0) / 16);
    const int rounds = (nblk + 1) / 2;  // round r: blocks 2 r (pair 0) and 2 r + 1 (pair 1)
    const int pair = wave & 1;          // column wave `pair` feeds row wave 2 + pair
    double4_t dr[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dr[jj] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (wave < 2) {
        // ---- column waves
        double gr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) gr[j] = gT[(r0 + 4 * j + lr) * SMM_NV + lc];
        const double* qbase = Q + (r0 + lr) * ld + lc;
        double xa[16], xb[16];
        auto fetch = [&](int b, double (&x)[16]) {
            const long long cb = c0 + 16 * (long long)b;
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = ld_stream<NT, double>(qbase + (long long)(4 * j) * ld + cb);
        };
        auto colwork = [&](int b, double (&x)[16], double* patch) {
            const long long cb = c0 + 16 * (long long)b;
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
            for (int j = 0; j < 16; ++j) patch[(4 * j + lr) * SMM_PITCH + lc] = x[j];
            const double o[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = lr + 4 * i;
                if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[i];
            }
        };
        if (pair < nblk) fetch(pair, xa);
        for (int r = 0; r < rounds; r += 2) {
            {
                const int b = 2 * r + pair, bn = b + 2;
                if (bn < nblk) fetch(bn, xb);
                if (b < nblk) colwork(b, xa, sh[pair][0]);
                __syncthreads();  // round r parked
            }
            if (r + 1 < rounds) {
                const int b = 2 * (r + 1) + pair, bn = b + 2;
                if (bn < nblk) fetch(bn, xa);
                if (b < nblk) colwork(b, xb, sh[pair][1]);
                __syncthreads();  // round r + 1 parked
            }
        }
        __syncthreads();  // the row waves' last read
    } else {
        // ---- row waves: one round behind
        auto rowwork = [&](int b, const double* patch) {
            const long long cb = c0 + 16 * (long long)b;
            double gc[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) gc[kb] = gT[(cb + 4 * kb + lr) * SMM_NV + lc];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const double t = patch[(16 * jj + lc) * SMM_PITCH + 4 * kb + lr];
                    dr[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[kb], t, dr[jj], 0, 0, 0);
                }
        };
        for (int r = 0; r < rounds; r += 2) {
            __syncthreads();  // round r parked
            {
                const int b = 2 * r + pair;
                if (b < nblk) rowwork(b, sh[pair][0]);
            }
            if (r + 1 < rounds) {
                __syncthreads();  // round r + 1 parked (and nobody overwrites parity 0 before the barrier after next)
                const int b = 2 * (r + 1) + pair;
                if (b < nblk) rowwork(b, sh[pair][1]);
            }
        }
        __syncthreads();
    }
    // row sums of the two row waves, in wave order
    double* red = &sh[0][0][0];  // [pair][jj][i][lane]: 2 * 16 * 64 doubles = 16 KiB
    if (wave >= 2) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const double o[4] = {dr[jj].x, dr[jj].y, dr[jj].z, dr[jj].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((pair * 4 + jj) * 4 + i) * 64 + lane] = o[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = lr + 4 * i;
        const int jj = wave;
        const double s0 = red[((0 * 4 + jj) * 4 + i) * 64 + lane], s1 = red[((1 * 4 + jj) * 4 + i) * 64 + lane];
        if (v < lv) rowpart[(long long)v * rowpart_stride + J * n + r0 + 16 * jj + lc] = s0 + s1;
    }
}

}  // namespace ellhip
