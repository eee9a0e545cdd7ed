// symv_units_kernels.hpp -- round-3 experiment, NOT in the product: the lower-triangle GEMV's tiles as work units of
// equal size with capped residency (k_symv_units) and a predicate-free body for full tiles (FULLT).  Bit-identical to
// the production k_symv and the whole GPU suite passed with it as the default -- but it is SLOWER: 0.195-0.199 ms against
// 0.186-0.188 ms at n = 16384 (profiles/r03/symv_units_residency_n16384.txt, symv_units_pipelined_n16384.txt): with every
// CU holding exactly four equal units the launch runs at 5.5 TB/s; balance is not what bounds k_symv.
#pragma once

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"

namespace ellhip {

// One tile (strip I, segment J) of the lower-triangle GEMV; returns false when the tile lies wholly above the diagonal
// or outside the local rows (nothing done).  HANDOFF: the partial sums are read by another workgroup of the same launch
// (tools/experiments/retired only).  FULLT: the caller guarantees that every column of the segment lies strictly left
// of every row of the strip and that all SYMV_H rows exist -- no predicate is left in the loop (the same arithmetic in
// the same order: identical bits).
template <int RW, bool NT, int ABL, int SEG, bool HANDOFF, bool FULLT = false>
__device__ __forceinline__ bool symv_tile_x(const double* __restrict__ Q, long long ld, long long n, long long row0,
                                          long long nrows, const double* __restrict__ g, double* __restrict__ rowpart,
                                          double* __restrict__ colpart, long long I, long long J, double (*red)[SYMV_H]) {
    constexpr int SYMV_NCH = SEG / 512;  // 16-byte column chunks per thread
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // Row shard (symmetric multi-GPU mode): Q holds the rows [row0, row0 + nrows) only; I counts local strips,
    // every row / column index below is global.  Unsharded: row0 = 0, nrows = n.
    const long long r0 = row0 + I * SYMV_H;
    const long long c0 = J * SEG;
    const long long rend = row0 + nrows;  // one past the last local row
    if (r0 >= rend || c0 > r0 + SYMV_H - 1) return false;  // nothing at or left of the diagonal in this segment
    const long long rlast = (r0 + SYMV_H - 1 < rend - 1) ? r0 + SYMV_H - 1 : rend - 1;
    Q -= row0 * ld;  // so that Q + r * ld addresses global row r
    const bool full = FULLT || c0 + SEG - 1 < r0;  // every column of the segment is strictly left of every row

    long long ck[SYMV_NCH];
    double2_t gc[SYMV_NCH], accc[SYMV_NCH];
#pragma unroll
    for (int k = 0; k < SYMV_NCH; ++k) {
        ck[k] = c0 + 512 * k + 2 * (long long)threadIdx.x;
        const bool in = FULLT || ck[k] <= rlast;  // n is even and ck is even: ck <= n - 2, so the pair is inside the matrix
        gc[k] = in ? *reinterpret_cast<const double2_t*>(g + ck[k]) : double2_t{0.0, 0.0};
        accc[k] = double2_t{0.0, 0.0};
    }
    for (int rg = 0; rg < SYMV_H / RW; ++rg) {
        double accr[RW];
        double gr[RW];
        long long rr[RW];
        double2_t q[RW][SYMV_NCH];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            rr[r] = r0 + rg * RW + r;
            const bool rv = FULLT || rr[r] < rend;
            gr[r] = rv ? g[rr[r]] : 0.0;
            accr[r] = 0.0;
            const double* row = Q + (rv ? rr[r] : rend - 1) * ld;
#pragma unroll
            for (int k = 0; k < SYMV_NCH; ++k) {
                // load the pair when its first column is at or left of the diagonal of this row
                if (rv && (full || ck[k] <= rr[r])) q[r][k] = ld_stream<NT, double2_t>(row + ck[k]);
                else q[r][k] = double2_t{0.0, 0.0};
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
#pragma unroll
            for (int k = 0; k < SYMV_NCH; ++k) {
                double qx = q[r][k].x, qy = q[r][k].y;
                if (!full) {
                    // pair loaded iff ck <= r; its second element is above the diagonal when ck + 1 > r
                    if (ck[k] + 1 > rr[r]) qy = 0.0;
                }
                accr[r] += qx * gc[k].x;
                accr[r] += qy * gc[k].y;
                // column sums take strictly-below-diagonal elements only (the diagonal is counted once, in the row sum)
                const double cx = (full || ck[k] < rr[r]) ? qx : 0.0;
                const double cy = (full || ck[k] + 1 < rr[r]) ? qy : 0.0;
                if (ABL != 2) {
                    accc[k].x += cx * gr[r];
                    accc[k].y += cy * gr[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const double s = (ABL == 1) ? accr[r] : wave_allreduce_sum(accr[r]);
            if (lane == 0) red[wave][rg * RW + r] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < SYMV_H && (FULLT || r0 + threadIdx.x < rend)) {
        const int r = threadIdx.x;
        const double v = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
        if (HANDOFF) ho_store(rowpart + J * n + r0 + r, v); else rowpart[J * n + r0 + r] = v;
    }
#pragma unroll
    for (int k = 0; k < SYMV_NCH; ++k)
        if (FULLT || ck[k] <= rlast) {
            if (HANDOFF) {
                ho_store2(colpart + I * n + ck[k], accc[k]);
            } else {
                *reinterpret_cast<double2_t*>(colpart + I * n + ck[k]) = accc[k];
            }
        }
    return true;
}


// The same tiles as WORK UNITS of equal size (what runs by default).  The static (strip, segment) grid above is resident
// as a whole (1152 tiles on 1280 slots at n = 16384), a CU streams ~27 GB/s whatever it holds, and the dispatcher hands
// the CUs 3, 4 or 5 tiles: the launch ends when the CUs with 5 are done -- 0.188 ms where the same bytes stream in
// 0.154 ms (profiles/r03/symv_morph_n16384.txt: an all-full SQUARE grid of the same tiles runs at 6.6-7.0 TB/s, the
// triangle at 5.3-6.0; the storage layout and DRAM offsets make no difference, profiles/r03/layout_n16384_*.txt).  The
// tiles that are not full are the diagonal ones, one per strip, and their sizes pair up: strip I's diagonal tile and
// strip (nstrips - 1 - I)'s together hold exactly one full tile's elements.  So the host lists UNITS -- a full tile, or
// two diagonal tiles (largest with smallest) run one after the other by the same workgroup -- and the launch is a 1-D
// grid over that list: n = 16384: 896 + 128 = 1024 units = 4 per CU, and `lds_cap` bytes of (unused) dynamic LDS per
// workgroup keep a CU from taking a fifth.  Every tile is computed by symv_tile exactly as before and writes the same
// partial sums: identical bits, and k_symv_reduce does not change.  units[u] = {I1, J1, I2, J2}: I2 = -2 one FULL tile
// (predicate-free body), I2 = -1 one diagonal tile, I2 >= 0 two.
template <int RW, bool NT, int SEG>
__global__ __launch_bounds__(256) void k_symv_units(const double* __restrict__ Q, long long ld, long long n,
                                                    long long row0, long long nrows, const double* __restrict__ g,
                                                    double* __restrict__ rowpart, double* __restrict__ colpart,
                                                    const DevState* __restrict__ st, const int4* __restrict__ units) {
    __shared__ double red[4][SYMV_H];
    if (st->halted) return;
    const int4 u = units[blockIdx.x];
    if (u.z == -2) {
        symv_tile_x<RW, NT, 0, SEG, false, true>(Q, ld, n, row0, nrows, g, rowpart, colpart, (long long)u.x, (long long)u.y, red);
        return;
    }
    symv_tile_x<RW, NT, 0, SEG, false, false>(Q, ld, n, row0, nrows, g, rowpart, colpart, (long long)u.x, (long long)u.y, red);
    if (u.z >= 0) {
        __syncthreads();  // `red` is reused
        symv_tile_x<RW, NT, 0, SEG, false, false>(Q, ld, n, row0, nrows, g, rowpart, colpart, (long long)u.z, (long long)u.w, red);
    }
}

}  // namespace ellhip
