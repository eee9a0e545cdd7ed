// symm32_kernel.hpp -- round-3 experiment: k_symm_mfma for up to 32 gradients per pass (two N-tiles of the 16 x 16 x 4
// MFMA over the same block of Q).  The A operands of the column product (gT rows of the strip, 64 x 32) live in LDS, shared
// by the four waves; the block, its transposed patch and the accumulators of both N-tiles stay per wave.  Measured by
// tools/experiments/symv_multi.hip; not in the product (profiles/r03/symm_mfma_variants_n16384.txt).
#pragma once
#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"
namespace ellhip {

// (SMM_NV2 = 32: ell_kernels.hpp, since the queue form of this kernel went into the product as k_symm_mfma_q2)

__global__ __launch_bounds__(256) void k_pack_grads32(const double* __restrict__ g, long long g_stride, int lv, long long n,
                                                      double* __restrict__ gT) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * SMM_NV2) return;
    const long long c = i / SMM_NV2;
    const int v = (int)(i % SMM_NV2);
    gT[i] = v < lv ? g[(long long)v * g_stride + c] : 0.0;
}

template <bool NT, int SEG>
__global__ __launch_bounds__(256, 2) void k_symm_mfma32(const double* __restrict__ Q, long long ld, long long n,
                                                     const double* __restrict__ gT, int lv, double* __restrict__ rowpart,
                                                     double* __restrict__ colpart, long long rowpart_stride,
                                                     long long colpart_stride, const DevState* __restrict__ st) {
    __shared__ double sh[4][SYMV_H * SMM_PITCH];
    __shared__ double sgr[SYMV_H][SMM_NV2 + 1];  // gT rows of the strip (odd pitch)
    if (st->halted) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = (long long)blockIdx.y;
    const long long r0 = I * SYMV_H;
    const long long c0 = J * SEG;
    if (r0 >= n || c0 > r0 + SYMV_H - 1) return;
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
        for (int j = 0; j < 16; ++j) {
            const long long r = r0 + 4 * j + lr, c = cb + lc;
            const double below = (!diag || c < r) ? x[j] : 0.0;
            dc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sgr[4 * j + lr][lc], below, dc[0], 0, 0, 0);
            dc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sgr[4 * j + lr][16 + lc], below, dc[1], 0, 0, 0);
            if (diag) x[j] = (c <= r) ? x[j] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) mysh[(4 * j + lr) * SMM_PITCH + lc] = x[j];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const double t = mysh[(16 * jj + lc) * SMM_PITCH + 4 * kb + lr];
                dr[0][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[0][kb], t, dr[0][jj], 0, 0, 0);
                dr[1][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[1][kb], t, dr[1][jj], 0, 0, 0);
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


// the same from a tile queue (symm_queue_kernel.hpp): 3 workgroups per CU draw tiles, largest first
template <bool NT, int SEG, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void k_symm_q32(const double* __restrict__ Q, long long ld, long long n,
                                                     const double* __restrict__ gT, int lv, double* __restrict__ rowpart,
                                                     double* __restrict__ colpart, long long rowpart_stride,
                                                     long long colpart_stride, const DevState* __restrict__ st,
                                                     const SymmTile* __restrict__ tiles, int ntiles, unsigned* __restrict__ queue) {
    __shared__ double sh[4][SYMV_H * SMM_PITCH];
    __shared__ double sgr[SYMV_H][SMM_NV2 + 1];
    __shared__ int s_t;
    if (st->halted) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = atomicAdd(queue, 1u);
            s_t = t < (unsigned)ntiles ? (int)t : -1;
        }
        __syncthreads();
        const int t = s_t;
        if (t < 0) return;
        const long long I = tiles[t].I, J = tiles[t].J;
        const long long r0 = I * SYMV_H;
        const long long c0 = J * SEG;
        const bool full = c0 + SEG - 1 < r0;
        for (int k = threadIdx.x; k < SYMV_H * SMM_NV2; k += 256) sgr[k / SMM_NV2][k % SMM_NV2] = gT[(r0 + k / SMM_NV2) * SMM_NV2 + k % SMM_NV2];
        __syncthreads();
        double4_t dr[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dr[tt][jj] = double4_t{0.0, 0.0, 0.0, 0.0};
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
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) gc[tt][kb] = gT[(cb + 4 * kb + lr) * SMM_NV2 + 16 * tt + lc];
            const bool diag = !full && cb + 15 >= r0;
            double4_t dc[2] = {double4_t{0.0, 0.0, 0.0, 0.0}, double4_t{0.0, 0.0, 0.0, 0.0}};
            // (one accumulator at a time: the pipe forwards it, mfma_f64_rate.hip)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const long long r = r0 + 4 * j + lr, c = cb + lc;
                    const double below = (!diag || c < r) ? x[j] : 0.0;
                    dc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sgr[4 * j + lr][16 * tt + lc], below, dc[tt], 0, 0, 0);
                }
            if (diag) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const long long r = r0 + 4 * j + lr, c = cb + lc;
                    x[j] = (c <= r) ? x[j] : 0.0;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) mysh[(4 * j + lr) * SMM_PITCH + lc] = x[j];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const double tv = mysh[(16 * jj + lc) * SMM_PITCH + 4 * kb + lr];
                        dr[tt][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(gc[tt][kb], tv, dr[tt][jj], 0, 0, 0);
                    }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const double o[4] = {dc[tt].x, dc[tt].y, dc[tt].z, dc[tt].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = 16 * tt + lr + 4 * i;
                    if (v < lv) colpart[(long long)v * colpart_stride + I * n + cb + lc] = o[i];
                }
            }
        }
        __syncthreads();
        double* red = &sh[0][0];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const double o[4] = {dr[tt][jj].x, dr[tt][jj].y, dr[tt][jj].z, dr[tt][jj].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) red[((wave * 4 + jj) * 4 + i) * 64 + lane] = o[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 16 * tt + lr + 4 * i;
                const int jj = wave;
                const double s0 = red[((0 * 4 + jj) * 4 + i) * 64 + lane], s1 = red[((1 * 4 + jj) * 4 + i) * 64 + lane];
                const double s2 = red[((2 * 4 + jj) * 4 + i) * 64 + lane], s3 = red[((3 * 4 + jj) * 4 + i) * 64 + lane];
                if (v < lv) rowpart[(long long)v * rowpart_stride + J * n + r0 + 16 * jj + lc] = ((s0 + s1) + s2) + s3;
            }
            __syncthreads();
        }
    }
}

}  // namespace ellhip
