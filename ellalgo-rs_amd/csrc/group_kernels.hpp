// group_kernels.hpp -- the scalar stage of a GROUP of queued cuts whose products y_l = Q_base g_l came out of one pass
// (k_symm_mfma, ELLHIP_OPT_LOOKAHEAD > 3): four parallel launches and one one-wave recurrence per group (k_group_reduce,
// k_group_gram, k_group_sums, k_group_scalar, k_group_apply; for a row shard k_group_dots after the owner's all-reduce)
// instead of a reduction and a scalar stage per cut.
//
// What Ell::update_core needs per cut (src/ell.rs:97-137 on the recorded schedule, ell_kernels.hpp "Deferred rank-1
// updates"):   gt_l = y_l - sum_j (c_j d_jl) v_j,   omega_l = g_l.y_l - sum_j c_j d_jl^2,   d_jl = v_j . g_l
// over the vectors v_j recorded before cut l -- including the ones the group itself records, v_m = gt_m (m < l), which is
// what chains the cuts: cut l's dot products wait for cut l - 1's vector.  But those dot products are linear in what is
// known up front:
//     v_m . g_l = y_m . g_l - sum_{j before m} (c_j d_jm) (v_j . g_l)
// so with   A_l  = g_l . y_l,   B_jl = v_j . g_l (j recorded before the group),   C_ml = y_m . g_l (m < l)
// (all n-length dot products of vectors that exist when the group starts: k_group_reduce, k_group_gram) the whole chain
// of omegas, EllCalc coefficients and correction factors is O(G^2 NP) scalar work (k_group_scalar, one wave), and
// the vectors follow in one elementwise pass (k_group_apply).  The matrix-core products differ from the vector-ALU
// schedules by a few ulp already; this stage adds the rounding of the recurrence above (same size: every term is a dot
// product of the same vectors), far inside the 1e-10 contract and not bit-identical to the per-cut stage.
#pragma once

#include "ell_kernels.hpp"

namespace ellhip {

constexpr int GRP_MAX = 32;  // cuts per group (= 2 * SMM_NV: one k_symm_mfma_q<.., .., 2> pass carries two 16-wide column tiles)

struct GroupOut {
    double cd[GRP_MAX][MAXPEND];  // cd[l][j] = c_j d_jl: the correction factors of cut l (0 for empty slots)
    double roo[GRP_MAX];          // rho / omega of cut l (src/ell.rs:112)
    int nok;                      // cuts of the group that succeeded (all of them unless the queue halted inside)
};

// y_l for every cut of the group (grid.y = l) + the dot products with what was recorded before the group: the body of
// k_symv_reduce<NP>, G of them in one launch.  gpart[l][b][0] = slice of g_l.y_l, [1 + j] = slice of v_j.g_l.
template <int NP>
__global__ __launch_bounds__(256) void k_group_reduce(long long n, long long row0, long long nrows, long long seg,
                                                      const double* __restrict__ rowpart,
                                                      const double* __restrict__ colpart, long long rowpart_stride,
                                                      long long colpart_stride, double* __restrict__ Y,
                                                      const double* __restrict__ g, long long g_stride,
                                                      const double* __restrict__ pend, double* __restrict__ gpart,
                                                      const DevState* __restrict__ st, int slot0 = NP) {
    __shared__ double2_t part[4][64];
    if (st->halted) return;
    const long long l = blockIdx.y, nb = gridDim.x;
    // (NP = 0, a row shard: its partial y_l only -- the dot products wait for the all-reduce, k_group_dots)
    // (slot0: the slots recorded before the group; k_group_scalar reads B[j][l] for j < slot0 only)
    symv_reduce_block<NP, false>((long long)blockIdx.x, n, row0, nrows, seg, rowpart + l * rowpart_stride,
                                 colpart + l * colpart_stride, Y + l * n, g + l * g_stride, pend, gpart + l * nb * (NP + 1), part, slot0);
}

// The same for a symmetric row shard: k_group_reduce<0> yields the shard's PARTIAL y_l, the owner's ONE all-reduce of
// the G vectors completes them, and the dot products follow from the complete vectors (every rank computes the same).
// gpart as above; lane pairs and wave sums as in symv_reduce_block's dot section.
template <int NP>
__global__ __launch_bounds__(256) void k_group_dots(long long n, const double* __restrict__ Y, const double* __restrict__ g,
                                                    long long g_stride, const double* __restrict__ pend,
                                                    double* __restrict__ gpart, const DevState* __restrict__ st, int slot0 = NP) {
    if (st->halted) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long l = blockIdx.y, nb = gridDim.x, blk = blockIdx.x;
    const long long i = blk * 128 + 2 * lane;
    double* out = gpart + (l * nb + blk) * (NP + 1);
    double2_t gi = {0.0, 0.0};
    if (i < n) gi = *reinterpret_cast<const double2_t*>(g + l * g_stride + i);
    if (wave == 0) {
        double2_t yv = {0.0, 0.0};
        if (i < n) yv = *reinterpret_cast<const double2_t*>(Y + l * n + i);
        double sgy = gi.x * yv.x;
        sgy += gi.y * yv.y;
        sgy = wave_allreduce_sum(sgy);
        if (lane == 0) out[0] = sgy;
    }
    for (int j = wave; j < NP; j += 4) {
        double2_t pv = {0.0, 0.0};
        if (i < n && j < slot0) pv = *reinterpret_cast<const double2_t*>(pend + (long long)j * n + i);
        double sv = pv.x * gi.x;
        sv += pv.y * gi.y;
        sv = wave_allreduce_sum(sv);
        if (lane == 0) out[1 + j] = sv;
    }
}

// cpart[b][m][l] = slice (128 columns) of y_m . g_l, m < l.  The slice passes through LDS in two halves of 64 columns (32 cuts x 128
// columns x 2 arrays would not fit the 64 KiB a kernel may declare); every (m, l) keeps ONE running sum over the 128 columns in
// ascending order, so the bits do not depend on the halves.
__global__ __launch_bounds__(256) void k_group_gram(long long n, int G, const double* __restrict__ Y,
                                                    const double* __restrict__ g, long long g_stride,
                                                    double* __restrict__ cpart, const DevState* __restrict__ st) {
    __shared__ double sy[GRP_MAX][64];
    __shared__ double sg[GRP_MAX][65];  // (odd pitch: the threads of one m read different l at the same column)
    if (st->halted) return;
    const int tid = threadIdx.x;
    constexpr int PAIRS = GRP_MAX * GRP_MAX / 256;  // (m, l) pairs per thread at most
    double acc[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) acc[q] = 0.0;
    for (int h = 0; h < 2; ++h) {
        const long long base = (long long)blockIdx.x * 128 + 64 * h;
        if (h) __syncthreads();
        for (int idx = tid; idx < G * 64; idx += 256) {
            const int l = idx >> 6, c = idx & 63;
            const long long i = base + c;
            sy[l][c] = i < n ? Y[(long long)l * n + i] : 0.0;
            sg[l][c] = i < n ? g[(long long)l * g_stride + i] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            const int t = tid + 256 * q;
            if (t < G * G) {
                const int m = t / G, l = t - m * G;
                if (m < l) {
                    double s = acc[q];
#pragma unroll 8
                    for (int c = 0; c < 64; ++c) s += sy[m][c] * sg[l][c];
                    acc[q] = s;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PAIRS; ++q) {
        const int t = tid + 256 * q;
        if (t < G * G) {
            const int m = t / G, l = t - m * G;
            cpart[(long long)blockIdx.x * (GRP_MAX * GRP_MAX) + m * GRP_MAX + l] = acc[q];
        }
    }
}

// Column sums of the slices: workgroup l < G adds gpart[l][0 .. nb)[0 .. NP] (contiguous: coalesced), workgroups G .. G + 3 add
// cpart[0 .. nb)[m][l'], 256 of its GRP_MAX^2 columns each (grid = G + GRP_MAX^2 / 256).  sums: [G][NP + 1] then [GRP_MAX][GRP_MAX].  Thread (c, q) adds rows q, q + R, ... of column c in
// ascending order, the R partial sums are combined in index order.
template <int NP>
__global__ __launch_bounds__(256) void k_group_sums(int G, int nb, const double* __restrict__ gpart,
                                                    const double* __restrict__ cpart, double* __restrict__ sums,
                                                    const DevState* __restrict__ st) {
    __shared__ double red[256];
    if (st->halted) return;
    const int tid = threadIdx.x;
    const int l = blockIdx.x;
    if (l < G) {
        constexpr int W = NP + 1;
        constexpr int R = 256 / W;  // row classes (NP = 48: 5, 24: 10, 16: 15, 8: 28)
        const double* p = gpart + (long long)l * nb * W;
        const int c = tid % W, q = tid / W;
        // (eight loads in flight, added in the order of the plain loop: beside the next group's pass -- which saturates HBM -- a
        // load takes microseconds, and one load per trip made this launch last 300 us instead of 14)
        double a = 0.0;
        if (q < R) {
            int b = q;
            for (; b + 7 * R < nb; b += 8 * R) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(long long)(b + u * R) * W + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; b < nb; b += R) a += p[(long long)b * W + c];
        }
        red[tid] = a;
        __syncthreads();
        if (tid < W) {
            double sum = red[tid];
            for (int r = 1; r < R; ++r) sum += red[r * W + tid];
            sums[l * W + tid] = sum;
        }
    } else {  // the Gram slices: one column (m, l') per thread, rows in ascending order, four running sums
        constexpr int W2 = GRP_MAX * GRP_MAX;
        const int col = (l - G) * 256 + tid;
        if (col / GRP_MAX >= G || col % GRP_MAX >= G) {  // (beyond the group: nobody reads it)
            sums[G * (NP + 1) + col] = 0.0;
            return;
        }
        const double* p = cpart + col;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = 0;
        for (; b + 15 < nb; b += 16) {  // (sixteen loads in flight; the four running sums take them in the order of the loop below)
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long long)(b + u) * W2];
#pragma unroll
            for (int u = 0; u < 16; u += 4) a0 += v[u], a1 += v[u + 1], a2 += v[u + 2], a3 += v[u + 3];
        }
        for (; b + 3 < nb; b += 4) {
            const double v0 = p[(long long)b * W2], v1 = p[(long long)(b + 1) * W2];
            const double v2 = p[(long long)(b + 2) * W2], v3 = p[(long long)(b + 3) * W2];
            a0 += v0, a1 += v1, a2 += v2, a3 += v3;
        }
        for (; b < nb; ++b) a0 += p[(long long)b * W2];
        sums[G * (NP + 1) + col] = (a0 + a1) + (a2 + a3);
    }
}

// One wave: cut by cut omega, EllCalc (src/ell_calc.rs:627-931 through EllCalcDev::dispatch), the correction factors and
// the dot products of the vector the cut records with the gradients still to come.  Lane j looks after recorded slot j
// (slot0 + G <= the run's depth <= MAXPEND <= 64: a group never reaches across an apply pass), lane t < G after cut t.  Queue semantics as k_scalar_apply_def: the first cut that is not Success halts the queue,
// every later one reports ELLHIP_UNKNOWN (src/cutting_plane.rs:222,308).
template <int NP>
__global__ __launch_bounds__(64) void k_group_scalar(int G, const double* __restrict__ sums, double* __restrict__ cpend,
                                                     DevState* __restrict__ st, EllCalcDev calc,
                                                     const CutParams* __restrict__ cp_dev, int slot0,
                                                     int* __restrict__ q_status, double* __restrict__ q_tsq,
                                                     GroupOut* __restrict__ out) {
    static_assert(MAXPEND <= 64 && GRP_MAX <= 64 && NP <= MAXPEND, "one lane per recorded slot, one per cut of the group");
    __shared__ double A[GRP_MAX];
    __shared__ double B[NP][GRP_MAX];
    __shared__ double C[GRP_MAX][GRP_MAX];
    __shared__ double D[GRP_MAX][GRP_MAX];
    __shared__ double bc[4];  // roo, cnew, kappa | status in s_status
    __shared__ double cdv[MAXPEND];
    // results collected in LDS and stored after the loop: __syncthreads waits for outstanding global stores, and the loop has
    // two barriers per cut
    __shared__ double o_cd[GRP_MAX][MAXPEND], o_roo[GRP_MAX], o_cnew[GRP_MAX], o_tsq[GRP_MAX];
    __shared__ int o_status[GRP_MAX];
    __shared__ int s_status, s_halt;
    __shared__ CutParams s_cp[GRP_MAX];  // (one round trip for all of them, not one per cut on the serial path)
    const int lane = threadIdx.x;
    const bool was_halted = st->halted != 0;
    if (lane < G) s_cp[lane] = cp_dev[lane];
    const double tol = st->tol;
    if (!was_halted) {
        // (eight loads in flight per trip: this wave runs beside the next group's pass, where a load takes microseconds)
        for (int k0 = 0; k0 < G * (NP + 1); k0 += 8 * 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u + lane;
                v[u] = k < G * (NP + 1) ? sums[k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u + lane;
                if (k < G * (NP + 1)) {
                    const int l = k / (NP + 1), c = k - l * (NP + 1);
                    if (c == 0) A[l] = v[u];
                    else B[c - 1][l] = v[u];
                }
            }
        }
        for (int k0 = 0; k0 < GRP_MAX * GRP_MAX; k0 += 8 * 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sums[G * (NP + 1) + k0 + 64 * u + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u + lane;
                C[k / GRP_MAX][k % GRP_MAX] = v[u];
            }
        }
    }
    double cc = (lane < NP && !was_halted) ? cpend[lane] : 0.0;  // c_j of this lane's slot
    if (lane == 0) {
        s_halt = was_halted ? 1 : 0;
        s_status = ST_SUCCESS;
    }
    __syncthreads();
    double kappa = st->kappa;
    const double tsq_in = st->tsq;
    int nok = 0;
    // what the cuts leave in DevState: kept in lane 0's registers and stored once at the end (a store followed by a load of
    // the same structure costs the serial path a memory round trip per cut)
    double f_tsq = tsq_in, f_omega = 0.0, f_roo = 0.0, f_ratio = 0.0;
    int f_status = ST_SUCCESS, f_apply = 0, f_any = 0, f_stop = STOP_NONE;
    long long f_niter = 0;
    for (int l = 0; l < G; ++l) {
        const int slot = slot0 + l;  // every earlier cut of the group succeeded, or the queue has halted
        if (s_halt) {
            if (lane == 0) {
                o_status[l] = ST_UNKNOWN;
                o_tsq[l] = f_tsq;
            }
            continue;  // (s_halt is not written again: uniform)
        }
        // d_jl of this lane's slot, its correction factor, and omega = g.y - sum_j c_j d_jl^2
        const double dj = lane < slot0 ? B[lane < NP ? lane : 0][l] : (lane < slot ? D[lane - slot0][l] : 0.0);
        const double cd = (lane < slot) ? cc * dj : 0.0;
        if (lane < MAXPEND) {
            o_cd[l][lane] = cd;
            cdv[lane] = cd;
        }
        const double omega = A[l] - wave_allreduce_sum(cd * dj);
        if (lane == 0) {
            const double tsq = kappa * omega;  // src/ell.rs:105
            Coef cf;
            const CutParams cp = s_cp[l];
            const int status = calc.dispatch(cp.kind, cp.b0, cp.has_b1, cp.b1, tsq, cf);  // :106
            f_any = 1;
            f_tsq = tsq;
            f_omega = omega;
            f_status = status;
            if (status == ST_SUCCESS) {
                const double roo = cf.rho / omega;     // :112
                const double cnew = cf.sigma / omega;  // :117
                o_roo[l] = roo;
                o_cnew[l] = cnew;
                f_roo = roo;
                f_ratio = cnew;
                f_apply = 1;
                bc[1] = cnew;
                bc[2] = kappa * cf.delta;  // :130
            } else {
                f_apply = 0;  // :107-109
            }
            // queue_bookkeeping (src/cutting_plane.rs:222,308)
            int halt = 0;
            if (status != 0) {
                halt = 1;
                f_stop = STOP_STATUS;
            } else if (tsq < tol) {
                halt = 1;  // (this cut's shrink is still recorded)
                f_stop = STOP_TOL;
            } else {
                f_niter += 1;
            }
            o_status[l] = status;
            o_tsq[l] = tsq;
            s_status = status;
            s_halt = halt;
        }
        __syncthreads();
        if (s_status == ST_SUCCESS) {
            nok = l + 1;
            kappa = bc[2];
            if (lane == slot) cc = bc[1];
            // v_l . g_t = y_l . g_t - sum_j cd_j (v_j . g_t) for the cuts t still to come: lane t its own, slots in order
            if (lane > l && lane < G) {
                // (the products first, eight LDS reads in flight; then the subtractions in slot order)
                double d = C[l][lane];
                int jq = 0;
                for (; jq + 8 <= slot0; jq += 8) {
                    double pr[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pr[u] = cdv[jq + u] * B[jq + u][lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) d = d - pr[u];
                }
                for (; jq < slot0; ++jq) d = d - cdv[jq] * B[jq][lane];
                for (; jq + 4 <= slot; jq += 4) {
                    double pr[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pr[u] = cdv[jq + u] * D[jq + u - slot0][lane];
#pragma unroll
                    for (int u = 0; u < 4; ++u) d = d - pr[u];
                }
                for (; jq < slot; ++jq) d = d - cdv[jq] * D[jq - slot0][lane];
                D[l][lane] = d;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int k = lane; k < G * MAXPEND; k += 64) out->cd[k / MAXPEND][k % MAXPEND] = (k / MAXPEND < nok) ? o_cd[k / MAXPEND][k % MAXPEND] : 0.0;
    if (lane < G) {
        q_status[lane] = o_status[lane];
        q_tsq[lane] = o_tsq[lane];
        if (lane < nok) {
            out->roo[lane] = o_roo[lane];
            cpend[slot0 + lane] = o_cnew[lane];
        }
    }
    if (lane == 0 && f_any) {
        st->tsq = f_tsq;
        st->omega = f_omega;
        st->status = f_status;
        st->apply = f_apply;
        if (nok > 0) {
            st->rho_over_omega = f_roo;   // of the last successful cut, as the cut-by-cut stage leaves them
            st->ratio = f_ratio;
            st->kappa = kappa;
            st->scale = 1.0;
            st->npend = slot0 + nok;
        }
        if (f_stop != STOP_NONE) {
            st->halted = 1;
            st->stop = f_stop;
        }
        st->niter += f_niter;
    }
    if (lane == 0) out->nok = nok;
}

// The vectors: v_m = y_m - sum_j cd[m][j] v_j in slot order (the recorded ones first, then the group's own), recorded
// into slot slot0 + m, and xc -= (rho / omega)_m v_m (src/ell.rs:113-115), element by element.
template <int NP>
__global__ __launch_bounds__(256) void k_group_apply(long long n, int G, const double* __restrict__ Y,
                                                     double* __restrict__ pend, double* __restrict__ xc, int slot0,
                                                     const GroupOut* __restrict__ out) {
    __shared__ double s_cd[GRP_MAX][MAXPEND];
    __shared__ double s_roo[GRP_MAX];
    const int tid = threadIdx.x;
    const int nok = out->nok;
    if (nok == 0) return;
    for (int k = tid; k < GRP_MAX * MAXPEND; k += 256) s_cd[k / MAXPEND][k % MAXPEND] = out->cd[k / MAXPEND][k % MAXPEND];
    if (tid < GRP_MAX) s_roo[tid] = out->roo[tid];
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + tid;
    if (i >= n) return;
    double pold[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) pold[j] = (j < slot0) ? pend[(long long)j * n + i] : 0.0;
    double ym[GRP_MAX];
#pragma unroll
    for (int m = 0; m < GRP_MAX; ++m) ym[m] = (m < nok) ? Y[(long long)m * n + i] : 0.0;
    double x = xc[i];
    double vnew[GRP_MAX];
#pragma unroll
    for (int m = 0; m < GRP_MAX; ++m) {
        if (m < nok) {
            double v = ym[m];
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (j < slot0) v = v - s_cd[m][j] * pold[j];
#pragma unroll
            for (int mm = 0; mm < GRP_MAX; ++mm)
                if (mm < m) v = v - s_cd[m][slot0 + mm] * vnew[mm];
            vnew[m] = v;
            pend[(long long)(slot0 + m) * n + i] = v;
            x = x - s_roo[m] * v;
        } else {
            vnew[m] = 0.0;
        }
    }
    xc[i] = x;
    (void)G;
}

}  // namespace ellhip
