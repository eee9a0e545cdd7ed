// ellstable_retired_kernels.hpp -- EllStable kernel forms that were built, tested bit-identical and measured SLOWER than
// the production forms in round 2, moved out of libellhip.so in round 3 with their host plumbing
// (last tree that had them wired in: commit 8411b35):
//   k_st_fwd_persist2 / k_st_bwd_persist2   two 128-blocks per workgroup               (profiles/r02/ellstable_pair_experiment.txt)
//   k_st_bwd_factor                         backward solve + DEDICATED factor workers  (profiles/r02/ellstable_fused_workers.txt)
#pragma once

#include "../../../ellalgo-rs_amd/csrc/ellstable_kernels.hpp"

namespace ellhip {

// Paired persistent forward solve: workgroup t owns TWO consecutive 128-blocks, A = 2t and B = 2t + 1 (256 columns; a
// lane holds two columns of each).  The row blocks above the pair are applied to both column blocks as their w arrives
// from the workgroups before; then block A is solved, its result is applied to block B's columns straight from LDS
// (no store / poll round trip through memory, ~3 us under load), and block B is solved: the chain pays one
// inter-workgroup hand-off per 256 columns instead of one per 128, and block A is published while block B is still
// being solved (the next workgroup applies it meanwhile).  Same arithmetic in the same order per column as
// k_st_fwd_persist / k_st_fwd_step: identical bits.
//   256 threads, one workgroup per CU: the waves keep the 512-register budget the row panels (two column blocks: 256
//   VGPRs) and the chains need.  (A first version with 512 threads -- one half of the workgroup per block, running
//   side by side -- had 256 registers per wave, spilled, and scratch memory limits how many workgroups the device keeps
//   resident: the chain serialised, 4.1 ms instead of 0.95.)
//   LDS: ONE diagonal block is parked at a time (A's, then B's, in the same 97.5 KB); the transpose tiles of the two
//   column blocks' write-backs alias it (144 KB in all).
constexpr int ST_LDS2_DOUBLES = 2 * 4 * SPANEL * SLDS_PAD;  // 18432 doubles; the parked pieces (12480) alias its start

// Bring the 128 x 128 diagonal block at J0 into L2 without holding it: one 8-byte load per 128-byte line, four per
// thread.  The paired solves fetch their SECOND block for real only when the first is solved (its 96 registers per
// thread do not fit beside the chain and the inner panel's rows); this makes that fetch an L2 hit.  The caller keeps
// the returned value alive.
__device__ __forceinline__ double st_touch_block(const double* __restrict__ M, long long ld, long long n, long long J0) {
    const long long row = threadIdx.x >> 2, off = 16 * (threadIdx.x & 3);
    auto at = [&](long long r, long long c) {
        if (r > n - 1) r = n - 1;
        if (c > n - 1) c = 0;
        return M[r * ld + c];
    };
    return at(J0 + row, J0 + off) + at(J0 + row, J0 + SH + off) + at(J0 + SH + row, J0 + SH + off) +
           at(J0 + SH + row, J0 + off);
}

__global__ __launch_bounds__(256) void k_st_fwd_persist2(double* __restrict__ M, long long ld, long long n,
                                                         const double* __restrict__ g, double* __restrict__ w,
                                                         double* __restrict__ z, double* __restrict__ gg,
                                                         int* __restrict__ flags, int* __restrict__ err, int epoch,
                                                         const DevState* __restrict__ st) {
    if (st->halted) return;
    __shared__ __attribute__((aligned(16))) double lds[ST_LDS2_DOUBLES];
    __shared__ double part[2][4][SPANEL];
    __shared__ double wstrip[2][SPANEL];
    __shared__ double wblk[SB];   // w of the row block being applied (from another workgroup)
    __shared__ double wA[SB];     // w of this pair's block A, for the inner panel
    __shared__ double dlds[SB];
    __shared__ int ok;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long nblk = (n + SB - 1) / SB;
    const long long blkA = 2 * (long long)blockIdx.x;
    const bool hasB = blkA + 1 < nblk;   // an odd block count leaves the last workgroup with block A only
    const long long cA = blkA * SB, cB = cA + SB;
    const long long clA = (cA + 2 * lane < n) ? cA + 2 * lane : 0;
    const long long clB = (hasB && cB + 2 * lane < n) ? cB + 2 * lane : 0;
    const long long dAB = clB - clA;   // (128, except for lanes beyond column n - 1 of a ragged last block)
    const int piece = lane & 7;

    Blk3 blk;
    double dreg = 0.0;
    if (threadIdx.x < SPANEL) {
        wstrip[0][threadIdx.x] = (cA + threadIdx.x < n) ? g[cA + threadIdx.x] : 0.0;
        wstrip[1][threadIdx.x] = (hasB && cB + threadIdx.x < n) ? g[cB + threadIdx.x] : 0.0;
    }
    double2_t u[2][2][16];   // [column block][pass][row]
    // products S[col][row] = U[row][col] * w[row] of a row block, transposed through LDS into full 128-byte lines of S;
    // `which`: bit 0 = column block A, bit 1 = column block B; recompute: from the factor entries (L2) and wsrc
    auto write_back = [&](long long J0, bool recompute, const double* wsrc, int which) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (!(which & (1 << cb))) continue;
            const long long cl = cb ? clB : clA;
            if (recompute) {
                const double* rp = M + (J0 + 32 * wv) * ld + cl;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const double2_t f = *reinterpret_cast<const double2_t*>(rp);
                        rp += ld;
                        const double wj = wsrc[32 * wv + 16 * h + r];
                        u[cb][h][r].x = f.x * wj;
                        u[cb][h][r].y = f.y * wj;
                    }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
            if (h) __syncthreads();  // the tiles of the previous pass have been drained
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (!(which & (1 << cb))) continue;
                double* t = lds + (cb * 4 + wv) * (SPANEL * SLDS_PAD);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    t[(2 * lane) * SLDS_PAD + r] = u[cb][h][r].x;
                    t[(2 * lane + 1) * SLDS_PAD + r] = u[cb][h][r].y;
                }
            }
            __syncthreads();
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (!(which & (1 << cb))) continue;
                const double* t = lds + (cb * 4 + wv) * (SPANEL * SLDS_PAD);
                const long long c0 = cb ? cB : cA;
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
        }
    };
    const int both = hasB ? 3 : 1;

    for (long long kb = 0; kb < blkA; ++kb) {
        const long long J0 = kb * SB;  // all 128 rows exist: J0 + 128 <= 128 blkA < n
        if (kb == blkA - 1) {
            // block A's diagonal block: fetched AND parked while the block before is still being solved (`lds` is free:
            // the last row block's products are written back at the very end)
            st_prefetch_block(M, ld, n, cA, blk, dreg);
            st_fwd_park(blk, dreg, lds, dlds);
        }
        {   // ONE running row pointer; block B's pair sits `dAB` doubles to the right (64 row addresses would be 128 VGPRs)
            const double* rp = M + (J0 + 32 * wv) * ld + clA;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    u[0][h][r] = *reinterpret_cast<const double2_t*>(rp);
                    if (hasB) u[1][h][r] = *reinterpret_cast<const double2_t*>(rp + dAB);
                    rp += ld;
                }
            }
        }
        if (kb >= blkA - 2) {
            // the two blocks of the workgroup right before this one: next in the chain, poll the values themselves
            // (the buffer is all-sentinel when the launch starts); one workgroup per block does, see k_st_fwd_persist
            if (threadIdx.x == 0) ok = 1;
            __syncthreads();
            if (threadIdx.x < SB) {
                double v = 0.0;
                if (!st_poll_value(w + J0 + threadIdx.x, v)) ok = 0;
                wblk[threadIdx.x] = v;
            }
            __syncthreads();
        } else {
            if (threadIdx.x == 0) ok = st_wait_flag(flags + kb, epoch) ? 1 : 0;
            __syncthreads();
            if (ok && threadIdx.x < SB) wblk[threadIdx.x] = st_published_load(w + J0 + threadIdx.x);
            __syncthreads();
        }
        if (!ok) {
            if (threadIdx.x == 0) atomicExch(err, 1);
            return;
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (cb && !hasB) continue;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double wj = wblk[32 * wv + 16 * h + r];
                    const double v0 = u[cb][h][r].x * wj;
                    const double v1 = u[cb][h][r].y * wj;
                    p0 += v0;
                    p1 += v1;
                    u[cb][h][r].x = v0;  // the product replaces the factor entry (src/ell_stable.rs:66)
                    u[cb][h][r].y = v1;
                }
            }
            part[cb][wv][2 * lane] = p0;
            part[cb][wv][2 * lane + 1] = p1;
        }
        __syncthreads();
        {
            const int cb = threadIdx.x >> 7, c = threadIdx.x & (SPANEL - 1);   // 256 threads: both column blocks at once
            if (!cb || hasB) {
                const double s4 = ((part[cb][0][c] + part[cb][1][c]) + part[cb][2][c]) + part[cb][3][c];
                wstrip[cb][c] = wstrip[cb][c] - s4;
            }
        }
        if (kb + 1 < blkA) {
            write_back(J0, false, wblk, both);
            __syncthreads();  // tiles drained, part / wblk reusable
        }
    }
    __syncthreads();
    // ---- block A; the inner panel's rows (block A's rows, block B's columns) and B's diagonal block are requested first
    double2_t u2[2][16];
    Blk3 blkB;
    double dregB = 0.0;
    if (hasB) {
        const double* rp = M + (cA + 32 * wv) * ld + clB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                u2[h][r] = *reinterpret_cast<const double2_t*>(rp);
                rp += ld;
            }
        }
        const double keep = st_touch_block(M, ld, n, cB);   // B's diagonal block: into L2 for now
        if (keep == 1.2345e300 && threadIdx.x == 999) atomicExch(err, 9);   // (keeps the loads alive; never true)
    }
    if (blkA == 0) st_prefetch_block(M, ld, n, cA, blk, dreg);  // no panel work before the first block: nothing parked yet
    st_fwd_diag_block<true>(M, ld, n, cA, blk, dreg, lds, dlds, wstrip[0], w, z, gg, flags + blkA, epoch, blkA > 0,
                            threadIdx.x, wA);
    if (!hasB) {
        __syncthreads();  // A's parked products have been written back: `lds` is free for the transposes
        if (blkA > 0) write_back((blkA - 1) * SB, true, wblk, 1);
        return;
    }
    // ---- inner panel: w_A straight from LDS (B's diagonal block is requested first: an L2 hit by now)
    st_prefetch_block(M, ld, n, cB, blkB, dregB);
    {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double wj = wA[32 * wv + 16 * h + r];
                p0 += u2[h][r].x * wj;   // (the products themselves are recomputed for the write-back at the end)
                p1 += u2[h][r].y * wj;
            }
        }
        part[1][wv][2 * lane] = p0;
        part[1][wv][2 * lane + 1] = p1;
    }
    __syncthreads();  // (A's parked products have also been written back by now: the pieces are free)
    if (threadIdx.x < SPANEL) {
        const int c = threadIdx.x;
        const double s4 = ((part[1][0][c] + part[1][1][c]) + part[1][2][c]) + part[1][3][c];
        wstrip[1][c] = wstrip[1][c] - s4;
    }
    st_fwd_park(blkB, dregB, lds, dlds);
    __syncthreads();
    // ---- block B
    st_fwd_diag_block<true>(M, ld, n, cB, blkB, dregB, lds, dlds, wstrip[1], w, z, gg, flags + blkA + 1, epoch, true);
    __syncthreads();  // B's parked products have been written back: `lds` is free for the transposes
    // ---- deferred write-backs: the row block right above the pair (both column blocks), then block A's rows (B's columns)
    if (blkA > 0) {
        write_back((blkA - 1) * SB, true, wblk, 3);
        __syncthreads();
    }
    write_back(cA, true, wA, 2);
}

// Paired persistent backward solve (see k_st_fwd_persist2): workgroup t owns the blocks H = nblk - 1 - 2t (solved first)
// and L = H - 1; block H's q reaches block L's inner panel through LDS.  Nothing is stored but q, so only one diagonal
// block is ever parked and there are no transposes.  256 threads, a lane holds two columns of each block.  Identical bits
// to k_st_bwd_persist / k_st_bwd_step.
__global__ __launch_bounds__(256) void k_st_bwd_persist2(const double* __restrict__ M, long long ld, long long n,
                                                         double* __restrict__ q, double* __restrict__ qpub,
                                                         int* __restrict__ err, const DevState* __restrict__ st) {
    if (!st->apply) return;
    __shared__ double lds[ST_LDS_DOUBLES_B];
    __shared__ double part[2][4][SPANEL];
    __shared__ double qstrip[2][SPANEL];
    __shared__ double qblk[SB];   // q of the row block being applied (from another workgroup)
    __shared__ double qH[SB];     // q of this pair's block H, for the inner panel
    __shared__ int ok;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long nblk = (n + SB - 1) / SB;
    const long long blkH = nblk - 1 - 2 * (long long)blockIdx.x;
    const bool hasL = blkH >= 1;             // an odd block count leaves the last workgroup with block 0 only
    const long long cH = blkH * SB, cL = hasL ? cH - SB : 0;
    // columns c, c + 1 < c0 + 128 <= J0 of every row block applied here: inside the matrix
    const long long clH = cH + 2 * lane, clL = cL + 2 * lane;

    Blk3b blk;
    if (threadIdx.x < SPANEL) {
        qstrip[0][threadIdx.x] = (cH + threadIdx.x < n) ? q[cH + threadIdx.x] : 0.0;
        qstrip[1][threadIdx.x] = hasL ? q[cL + threadIdx.x] : 0.0;
    }
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();

    for (long long kb = nblk - 1; kb > blkH; --kb) {
        const long long J0 = kb * SB;
        if (kb == blkH + 1) {  // block H's diagonal block: fetched and parked while the values it waits for are computed
            st_prefetch_block_bwd(M, ld, n, cH, blk);
            st_park_piece(lds, blk.bb, threadIdx.x);
            st_park_piece(lds + SH * BLK_PITCH, blk.ba, threadIdx.x);
            st_park_piece(lds + 2 * SH * BLK_PITCH, blk.aa, threadIdx.x);
        }
        double2_t sv[2][2][16];  // [column block][pass][row]: all rows requested before the wait
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = J0 + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long row = r0 + r;
                if (row > n - 1) row = n - 1;
                sv[0][h][r] = *reinterpret_cast<const double2_t*>(M + row * ld + clH);
                if (hasL) sv[1][h][r] = *reinterpret_cast<const double2_t*>(M + row * ld + clL);
            }
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
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (cb && !hasL) continue;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double qj = qblk[32 * wv + 16 * h + r];  // 0 for rows beyond n
                    p0 += sv[cb][h][r].x * qj;
                    p1 += sv[cb][h][r].y * qj;
                }
            }
            part[cb][wv][2 * lane] = p0;
            part[cb][wv][2 * lane + 1] = p1;
        }
        __syncthreads();
        {
            const int cb = threadIdx.x >> 7, c = threadIdx.x & (SPANEL - 1);
            if (!cb || hasL) {
                const double s4 = ((part[cb][0][c] + part[cb][1][c]) + part[cb][2][c]) + part[cb][3][c];
                qstrip[cb][c] = qstrip[cb][c] - s4;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const bool first = blockIdx.x == 0;   // block H is the matrix' last block: nothing above it, nothing parked yet
    // ---- the inner panel's rows (block H's rows of the scratch triangle, block L's columns) and block L's diagonal block
    // are requested before block H is solved
    double2_t s2[2][16];
    Blk3b blkL;
    if (hasL) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r0 = cH + 32 * wv + 16 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long long row = r0 + r;
                if (row > n - 1) row = n - 1;
                s2[h][r] = *reinterpret_cast<const double2_t*>(M + row * ld + clL);
            }
        }
        const double keep = st_touch_block(M, ld, n, cL);   // block L's diagonal block: into L2 for now
        if (keep == 1.2345e300 && threadIdx.x == 999) atomicExch(err, 9);   // (keeps the loads alive; never true)
    }
    if (first) st_prefetch_block_bwd(M, ld, n, cH, blk);
    st_bwd_diag_block<true>(n, cH, blk, lds, qstrip[0], q, qpub, !first, threadIdx.x, qH);
    if (!hasL) return;
    __syncthreads();   // q_H complete in LDS; block H's parked pieces are no longer read
    st_prefetch_block_bwd(M, ld, n, cL, blkL);   // (an L2 hit by now)
    {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = cH + 32 * wv + 16 * h + r;
                const double qj = (row < n) ? qH[32 * wv + 16 * h + r] : 0.0;  // rows beyond n count nothing
                p0 += s2[h][r].x * qj;
                p1 += s2[h][r].y * qj;
            }
        }
        part[1][wv][2 * lane] = p0;
        part[1][wv][2 * lane + 1] = p1;
    }
    st_park_piece(lds, blkL.bb, threadIdx.x);
    st_park_piece(lds + SH * BLK_PITCH, blkL.ba, threadIdx.x);
    st_park_piece(lds + 2 * SH * BLK_PITCH, blkL.aa, threadIdx.x);
    __syncthreads();
    if (threadIdx.x < SPANEL) {
        const int c = threadIdx.x;
        const double s4 = ((part[1][0][c] + part[1][1][c]) + part[1][2][c]) + part[1][3][c];
        qstrip[1][c] = qstrip[1][c] - s4;
    }
    __syncthreads();
    st_bwd_diag_block<true>(n, cL, blkL, lds, qstrip[1], q, qpub, true);
}

// Backward solve AND factor update in ONE launch (they are independent: the solve reads the scratch triangle and
// writes q, the update rewrites the strict upper triangle from itself).  Workgroups [0, nblk) run the persistent
// backward solve exactly as k_st_bwd_persist does (dispatch order = dependency order), with the diagonal block's
// columns in registers before the values arrive (st_bwd_diag_block_pre); the workgroups after them are factor WORKERS:
// worker k takes the tiles k, k + nworker, ... of the host-built list `ftiles` (strip << 8 | segment, the active tiles
// of k_st_factor_rows<SEG, RW>'s grid, largest first).  One launch, so the overlap does not depend on the device
// running two streams side by side (measured: on some boxes of the pool the two launches ran one after the other,
// 0.68 + 0.47 ms instead of 0.93 together), and -- the static LDS of the solve limits every workgroup of this kernel to
// one per CU -- the workers sit on OTHER CUs than the chain, which two separate kernels do not guarantee (the factor
// kernel's waves then share the chain's SIMDs: backward solve 0.68 -> 0.92-0.96 ms).
template <int SEG, int RW>
__global__ __launch_bounds__(256) void k_st_bwd_factor(double* __restrict__ M, long long ld, long long n,
                                                       double* __restrict__ q, double* __restrict__ qpub,
                                                       int* __restrict__ err, const DevState* __restrict__ st,
                                                       long long nblk, const double* __restrict__ beta2,
                                                       const double* __restrict__ w, const int* __restrict__ ftiles,
                                                       int nftiles, int* __restrict__ fnext) {
    if (!st->apply) return;
    // Factor tiles are PULLED (one atomic per 1 MiB tile): the workers start at once, and every chain workgroup joins
    // them when its block is solved -- a CU streams ~30 GB/s at most (its outstanding misses x latency), so the 128
    // worker CUs alone needed 0.85 ms for the 2.1 GB; the chain's CUs come free one by one, 5 us apart.
    __shared__ int ftile;
    auto work = [&]() __attribute__((always_inline)) {
        for (;;) {
            if (threadIdx.x == 0) ftile = atomicAdd(fnext, 1);
            __syncthreads();
            const int k = ftile;
            __syncthreads();
            if (k >= nftiles) break;
            const int tl = ftiles[k];
            st_factor_tile<SEG, RW, true>(M, ld, n, beta2, w, (long long)(tl >> 8), (long long)(tl & 0xff));
        }
    };
    if ((long long)blockIdx.x >= nblk) {
        work();
        return;
    }
    __shared__ double lds[ST_LDS_DOUBLES_B];
    __shared__ double part[4][SPANEL];
    __shared__ double qstrip[SPANEL];
    __shared__ double qblk[SB];
    __shared__ int ok;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long sblk = nblk - 1 - blockIdx.x;
    const long long c0 = sblk * SB;
    const long long c = c0 + 2 * lane;  // columns c, c+1 < c0 + 128 <= J0 of every row block applied here
    const bool has_b = c0 + SH < n;

    Blk3b blk;
    double svc[SH];  // this wave's columns of the parked diagonal block
    if (sblk == nblk - 1) {
        st_prefetch_block_bwd(M, ld, n, c0, blk);
        st_park_piece(lds, blk.bb, threadIdx.x);
        st_park_piece(lds + SH * BLK_PITCH, blk.ba, threadIdx.x);
        st_park_piece(lds + 2 * SH * BLK_PITCH, blk.aa, threadIdx.x);
    }
    if (threadIdx.x < SPANEL) qstrip[threadIdx.x] = (c0 + threadIdx.x < n) ? q[c0 + threadIdx.x] : 0.0;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    if (sblk == nblk - 1 && has_b) st_bwd_diag_cols(lds, svc);

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
        if (kb == sblk + 1) {  // own diagonal block: fetched, parked in LDS and its columns taken into registers while
                               // the values it waits for are computed (sblk < nblk - 1: both halves exist)
            st_prefetch_block_bwd(M, ld, n, c0, blk);
            st_park_piece(lds, blk.bb, threadIdx.x);
            st_park_piece(lds + SH * BLK_PITCH, blk.ba, threadIdx.x);
            st_park_piece(lds + 2 * SH * BLK_PITCH, blk.aa, threadIdx.x);
            __syncthreads();
            st_bwd_diag_cols(lds, svc);
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
    if (has_b) st_bwd_diag_block_pre(n, c0, qstrip, q, qpub, svc);
    else st_bwd_diag_block<true>(n, c0, blk, lds, qstrip, q, qpub, true);  // ragged last block, one half
    work();  // this block is solved and handed over: the CU joins the factor workers
}

}  // namespace ellhip
