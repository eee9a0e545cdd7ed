// lowpass_capi.inc.hpp -- C ABI of the device-side LowpassOracle and of the device-resident
// cutting-plane loops built on it.  Included at the end of ellhip_capi.hip (same translation unit: it
// issues the search-space primitives do_prime / do_cut / do_commit directly).
//
// Reference: src/oracles/lowpass_oracle.rs:22-151 (oracle), src/cutting_plane.rs:205-227,286-313 (loops).
#include "../../include/ellhip_lowpass.h"

#include <cmath>
#include <thread>

#include "lowpass_kernels.hpp"

struct ellhip_lowpass {
    int device = 0;
    LpParams P{};
    bool nt = false;
    bool wide = false;  // k_lp_scan_wide (4 positions per workgroup step) instead of k_lp_scan (16)
    unsigned grid = 1;
    double* d_A = nullptr;
    double* d_vals = nullptr;
    double* d_x = nullptr;
    double* d_g = nullptr;
    double* d_xbest = nullptr;
    LpState* d_ls = nullptr;
    CutParams* d_cp = nullptr;
    int* d_zero = nullptr;
    hipStream_t stream = nullptr;
    LpState* h_ls = nullptr;
    CutParams* h_cp = nullptr;
    double* h_vec = nullptr;
};

namespace {

// row i of the table: [1, 2cos(w_i), 2cos(2 w_i), ...], w = linspace(0, pi, mdim)
// (src/oracles/lowpass_oracle.rs:25-34, src/arr.rs:506-520)
void lp_fill_rows(double* dst, long long n, long long mdim, long long r0, long long r1) {
    const double pi = 3.14159265358979323846264338327950288;
    const double step = (mdim > 1) ? (pi - 0.0) / (double)(mdim - 1) : 0.0;
    for (long long i = r0; i < r1; ++i) {
        const double w = (mdim > 1) ? 0.0 + step * (double)i : 0.0;
        double* row = dst + (i - r0) * n;
        row[0] = 1.0;
        for (long long j = 1; j < n; ++j) row[j] = 2.0 * std::cos(w * (double)j);
    }
}

int lp_issue(ellhip_lowpass* o, hipStream_t st, const double* x_dev, int mode, DevState* drv, const int* halted,
             ellhip_space* prof) {
    {
        std::unique_ptr<ProfScope> ps;
        if (prof) ps.reset(new ProfScope(prof, CLS_LP_SCAN));
        const bool vec2 = (o->P.n % 2) == 0;
        // two launches per scan: the first one sized from the previous walk, the second one the rest (lp_range)
#define LP_LAUNCH(KERNEL)                                                                                             \
    for (int ph = 0; ph < 2; ++ph)                                                                                    \
        hipLaunchKernelGGL(KERNEL, dim3(o->grid), dim3(256), 0, st, (const double*)o->d_A, o->P, x_dev, o->d_vals,    \
                           o->d_ls, halted, ph);
        if (o->wide) {
            if (vec2 && o->nt) LP_LAUNCH((k_lp_scan_wide<2, true>))
            else if (vec2) LP_LAUNCH((k_lp_scan_wide<2, false>))
            else LP_LAUNCH((k_lp_scan_wide<1, false>))
        } else {
            if (vec2 && o->nt) LP_LAUNCH((k_lp_scan<2, true>))
            else if (vec2) LP_LAUNCH((k_lp_scan<2, false>))
            else LP_LAUNCH((k_lp_scan<1, false>))
        }
#undef LP_LAUNCH
        HIPCHK(hipGetLastError());
    }
    std::unique_ptr<ProfScope> ps;
    if (prof) ps.reset(new ProfScope(prof, CLS_LP_FINAL));
    hipLaunchKernelGGL(k_lp_final, dim3(1), dim3(256), 0, st, (const double*)o->d_A, o->P, x_dev,
                       (const double*)o->d_vals, o->d_ls, o->d_g, o->d_cp, o->d_xbest, mode, drv, halted);
    HIPCHK(hipGetLastError());
    return 0;
}

int lp_assess_host(ellhip_lowpass* o, int mode, const double* x, double* gamma_inout, double* grad_out, double* beta0,
                   int* has_beta1, double* beta1, int* shrunk) {
    if (!o || !x || !grad_out || !beta0 || !has_beta1 || !beta1) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(o->device);
    const size_t vbytes = (size_t)o->P.n * sizeof(double);
    memcpy(o->h_vec, x, vbytes);
    HIPCHK(hipMemcpyAsync(o->d_x, o->h_vec, vbytes, hipMemcpyHostToDevice, o->stream));
    if (mode == 1) {  // self.sp_sq = *sp_sq (:140)
        if (!gamma_inout || !shrunk) return fail(ELLHIP_E_INVALID, "NULL argument");
        HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(o->d_ls) + offsetof(LpState, sp_sq), gamma_inout, sizeof(double),
                              hipMemcpyHostToDevice, o->stream));
    }
    int rc = lp_issue(o, o->stream, o->d_x, mode, nullptr, o->d_zero, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(o->h_ls, o->d_ls, sizeof(LpState), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(o->h_cp, o->d_cp, sizeof(CutParams), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(o->h_vec, o->d_g, vbytes, hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    if (o->h_ls->error)
        return fail(ELLHIP_E_STATE, "lowpass oracle: feasible point without a stopband maximum (kmax = -1)");
    if (!o->h_ls->has_cut) return 0;
    memcpy(grad_out, o->h_vec, vbytes);
    *beta0 = o->h_cp->b0;
    *has_beta1 = o->h_cp->has_b1;
    *beta1 = o->h_cp->b1;
    if (mode == 1) {
        *shrunk = o->h_ls->shrunk;
        if (o->h_ls->shrunk) *gamma_inout = o->h_ls->sp_sq;  // *sp_sq = self.fmax (:149)
    }
    return 1;
}

// One device-resident loop: mode 1 = cutting_plane_optim (src/cutting_plane.rs:286-313), 0 = cutting_plane_feas
// (:205-227).  Iterations are enqueued in batches; every kernel of an iteration (oracle scan, oracle
// finish, GEMV, scalar stage, shrink) is a no-op once the loop has halted on the device, so the host
// looks at the state once per batch only.
int lp_drive(ellhip_space* s, ellhip_lowpass* o, int mode, double* gamma_inout, long long max_iters, double tol,
             double* x_best_out, int* has_best_out, int64_t* niter_out) {
    if (!s || !o || !has_best_out || !niter_out || (mode == 1 && !gamma_inout))
        return fail(ELLHIP_E_INVALID, "NULL argument");
    if (s->n != o->P.n) return fail(ELLHIP_E_INVALID, "oracle and search space dimensions differ");
    if (s->device != o->device) return fail(ELLHIP_E_INVALID, "oracle and search space live on different devices");
    if (s->sharded) return fail(ELLHIP_E_INVALID, "device-resident loops need an unsharded search space");
    if (max_iters < 0) return fail(ELLHIP_E_INVALID, "max_iters < 0");
    DeviceGuard guard(s->device);
    int rc = ensure_committed(s);
    if (rc) return rc;
    drop_prime(s);
    HIPCHK(hipStreamSynchronize(o->stream));
    hipStream_t st = s->stream;
    // loop state on the device: tolerance, iteration counter, stop reason; the oracle's gamma
    rc = read_back(s);
    if (rc) return rc;
    s->h_result->tol = tol;
    s->h_result->niter = 0;
    s->h_result->stop = STOP_NONE;
    s->h_result->halted = 0;
    HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(o->h_ls, o->d_ls, sizeof(LpState), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    o->h_ls->has_best = 0;
    o->h_ls->error = 0;
    if (mode == 1) o->h_ls->sp_sq = *gamma_inout;
    HIPCHK(hipMemcpyAsync(o->d_ls, o->h_ls, sizeof(LpState), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));

    int* d_halted = reinterpret_cast<int*>(reinterpret_cast<char*>(s->d_st) + offsetof(DevState, halted));
    const long long BATCH = 64;
    std::vector<int> slot_of((size_t)BATCH);
    long long done = 0;
    CutParams none{};
    bool stopped = false;
    while (done < max_iters && !stopped) {
        const long long nb = (max_iters - done < BATCH) ? max_iters - done : BATCH;
        for (long long i = 0; i < nb; ++i) {
            // oracle at the current centre; for every iteration but the first of a batch it runs between the
            // scalar stage of the previous cut and that cut's shrink, which then carries this cut's GEMV
            rc = lp_issue(o, st, s->d_xc, mode, s->d_st, d_halted, s);
            if (rc) return rc;
            if (s->shrink_pending || (deferring(s) && i > 0)) {
                rc = do_commit(s, s->shrink_pending, o->d_g);
                if (rc) return rc;
                s->shrink_pending = false;
                s->cur ^= 1;
            } else {
                rc = do_prime(s, o->d_g, s->cur);
                if (rc) return rc;
            }
            slot_of[(size_t)i] = s->cur;
            rc = do_cut(s, o->d_g, o->d_cp, none, 1, nullptr, nullptr);
            if (rc) return rc;
            s->shrink_pending = s->variant == ELLHIP_SPACE_ELL && !deferring(s);
        }
        // end of batch: apply the last shrink (it has no next gradient yet), then look at the loop state
        rc = do_commit(s, s->shrink_pending, nullptr);
        if (rc) return rc;
        s->shrink_pending = false;
        rc = read_back(s);
        if (rc) return rc;
        const DevState hs = *s->h_result;
        if (s->needs_mirror && (hs.niter > 0 || hs.stop == STOP_TOL)) s->needs_mirror = false;
        if (hs.halted) {
            stopped = true;
            const long long at = hs.niter - done;  // index of the stopping iteration inside this batch
            if (s->variant == ELLHIP_SPACE_ELL) s->npend = hs.npend;
            // clear the halt so that the space is usable again (and so that the shrink below runs)
            s->h_result->halted = 0;
            HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, st));
            if (hs.stop == STOP_TOL && s->variant == ELLHIP_SPACE_ELL && !deferring(s)) {
                // src/cutting_plane.rs:308 tests tsq AFTER the update: the update that hit the tolerance is
                // complete in the reference.  Its scalar stage set `halted`, which turned the shrink pass into a
                // no-op; gt of that cut is still in its slot and DevState.apply is still 1.
                if (at < 0 || at >= nb) return fail(ELLHIP_E_STATE, "lowpass driver: inconsistent iteration count");
                s->cur = slot_of[(size_t)at];
                rc = do_commit(s, true, nullptr);
                if (rc) return rc;
            }
            HIPCHK(hipStreamSynchronize(st));
        }
        done += nb;
    }
    drop_prime(s);
    rc = read_back(s);
    if (rc) return rc;
    // results
    HIPCHK(hipMemcpyAsync(o->h_ls, o->d_ls, sizeof(LpState), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (o->h_ls->error)
        return fail(ELLHIP_E_STATE, "lowpass oracle: feasible point without a stopband maximum (kmax = -1)");
    *niter_out = stopped ? s->h_result->niter : max_iters;
    *has_best_out = o->h_ls->has_best;
    if (o->h_ls->has_best && x_best_out) {
        HIPCHK(hipMemcpyAsync(o->h_vec, o->d_xbest, (size_t)o->P.n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        memcpy(x_best_out, o->h_vec, (size_t)o->P.n * sizeof(double));
    }
    if (mode == 1) *gamma_inout = o->h_ls->sp_sq;
    // plain queues and direct updates do not test a tolerance
    s->h_result->tol = -1.0;
    s->h_result->stop = STOP_NONE;
    s->h_result->niter = 0;
    HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

}  // namespace

extern "C" {

int ellhip_lowpass_create(ellhip_lowpass** out, int64_t ndim, double wpass, double wstop, double lp_sq, double up_sq,
                          double sp_sq, const double* spectrum, int device) {
    if (!out) return fail(ELLHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (ndim < 1 || ndim > (1 << 20)) return fail(ELLHIP_E_INVALID, "bad dimension");
    const long long mdim = 15 * ndim;  // :24
    const long long nwpass = (long long)std::floor(wpass * (double)(mdim - 1)) + 1;  // :36
    const long long nwstop = (long long)std::floor(wstop * (double)(mdim - 1)) + 1;  // :37
    if (!(nwpass >= 1 && nwpass <= nwstop && nwstop <= mdim))
        return fail(ELLHIP_E_INVALID, "band edges must satisfy 0 <= wpass <= wstop <= 1");
    const int ndev = ellhip_device_count();
    if (ndev <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device: the lowpass oracle has no CPU path");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) return fail(ELLHIP_E_INVALID, "device index out of range");
    ellhip_lowpass* o = new (std::nothrow) ellhip_lowpass();
    if (!o) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    o->device = device;
    o->P.n = ndim;
    o->P.ld = ndim + (ndim & 1);  // 16-byte aligned rows
    o->P.mdim = (int)mdim;
    o->P.nwpass = (int)nwpass;
    o->P.nwstop = (int)nwstop;
    o->P.lp_sq = lp_sq;
    o->P.up_sq = up_sq;
    const double a_bytes = (double)mdim * (double)o->P.ld * 8.0;
    o->nt = a_bytes > 200.0 * 1024 * 1024;  // same rule as the Q stream: larger than the Infinity Cache share
    o->wide = g_defaults.lp_wide >= 0 ? g_defaults.lp_wide != 0 : ndim >= LP_WIDE_N;  // ELLHIP_OPT_LP_WIDE
    const long long per_step = o->wide ? LP_RPW : LP_CHUNK;
    const long long nchunks = (mdim + per_step - 1) / per_step;
    o->grid = (unsigned)(nchunks < 1024 ? nchunks : 1024);
    if (g_defaults.lp_grid > 0) o->grid = (unsigned)g_defaults.lp_grid;  // ELLHIP_OPT_LP_GRID
    if (o->grid < 1) o->grid = 1;
    DeviceGuard guard(device);
    auto bail = [&](int code) {
        ellhip_lowpass_destroy(o);
        return code;
    };
    const size_t vbytes = (size_t)ndim * sizeof(double);
    hipError_t e = hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&o->d_A, (size_t)mdim * (size_t)o->P.ld * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&o->d_vals, (size_t)mdim * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&o->d_x, vbytes);
    if (e == hipSuccess) e = hipMalloc(&o->d_g, vbytes);
    if (e == hipSuccess) e = hipMalloc(&o->d_xbest, vbytes);
    if (e == hipSuccess) e = hipMalloc(&o->d_ls, sizeof(LpState));
    if (e == hipSuccess) e = hipMalloc(&o->d_cp, sizeof(CutParams));
    if (e == hipSuccess) e = hipMalloc(&o->d_zero, sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc(&o->h_ls, sizeof(LpState), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&o->h_cp, sizeof(CutParams), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&o->h_vec, vbytes, hipHostMallocDefault);
    if (e != hipSuccess) return bail(fail(e == hipErrorOutOfMemory ? ELLHIP_E_NOMEM : ELLHIP_E_HIP, "lowpass allocation", e));
    if (o->P.ld != ndim) e = fill_now(o->d_A, 0, (size_t)mdim * (size_t)o->P.ld * sizeof(double), o->stream);
    if (e == hipSuccess) e = fill_now(o->d_vals, 0, (size_t)mdim * sizeof(double), o->stream);
    if (e == hipSuccess) e = fill_now(o->d_zero, 0, sizeof(int), o->stream);
    if (e != hipSuccess) return bail(fail(ELLHIP_E_HIP, "lowpass memset", e));
    // the table: the caller's own (the reference's `spectrum` field, row-major mdim x ndim) or computed
    // here slab by slab with the host libm, exactly as LowpassOracle::new does
    const long long slab_rows = std::max<long long>(1, std::min<long long>(mdim, (64LL << 20) / (ndim * 8)));
    std::vector<double> slab;
    if (!spectrum) slab.resize((size_t)slab_rows * (size_t)ndim);
    for (long long r0 = 0; r0 < mdim; r0 += slab_rows) {
        const long long r1 = std::min(mdim, r0 + slab_rows);
        const double* src = spectrum ? spectrum + (size_t)r0 * (size_t)ndim : slab.data();
        if (!spectrum) {
            unsigned nthr = std::thread::hardware_concurrency();
            nthr = nthr < 1 ? 1 : (nthr > 16 ? 16 : nthr);
            if ((r1 - r0) * ndim < (1 << 16)) nthr = 1;
            std::vector<std::thread> pool;
            const long long per = (r1 - r0 + nthr - 1) / nthr;
            for (unsigned t = 0; t < nthr; ++t) {
                const long long a = r0 + (long long)t * per, b = std::min(r1, a + per);
                if (a >= b) break;
                pool.emplace_back(lp_fill_rows, slab.data() + (size_t)(a - r0) * (size_t)ndim, (long long)ndim, mdim, a, b);
            }
            for (auto& th : pool) th.join();
        }
        e = hipMemcpy2D(o->d_A + (size_t)r0 * (size_t)o->P.ld, (size_t)o->P.ld * sizeof(double), src,
                        (size_t)ndim * sizeof(double), (size_t)ndim * sizeof(double), (size_t)(r1 - r0),
                        hipMemcpyHostToDevice);
        if (e != hipSuccess) return bail(fail(ELLHIP_E_HIP, "lowpass table upload", e));
    }
    LpState ls;
    memset(&ls, 0, sizeof ls);
    ls.more_alt = 1;                  // :41
    ls.idx1 = -1;                     // :42
    ls.idx2 = (int)nwpass - 1;        // :49
    ls.idx3 = (int)nwstop - 1;        // :50
    ls.fmax = -__builtin_inf();       // :51
    ls.kmax = -1;                     // :52
    ls.sp_sq = sp_sq;
    ls.pos = LP_NONE;
    *o->h_ls = ls;
    e = hipMemcpy(o->d_ls, o->h_ls, sizeof(LpState), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(fail(ELLHIP_E_HIP, "lowpass state upload", e));
    *out = o;
    return 0;
}

void ellhip_lowpass_destroy(ellhip_lowpass* o) {
    if (!o) return;
    DeviceGuard guard(o->device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    if (o->d_A) (void)hipFree(o->d_A);
    if (o->d_vals) (void)hipFree(o->d_vals);
    if (o->d_x) (void)hipFree(o->d_x);
    if (o->d_g) (void)hipFree(o->d_g);
    if (o->d_xbest) (void)hipFree(o->d_xbest);
    if (o->d_ls) (void)hipFree(o->d_ls);
    if (o->d_cp) (void)hipFree(o->d_cp);
    if (o->d_zero) (void)hipFree(o->d_zero);
    if (o->h_ls) (void)hipHostFree(o->h_ls);
    if (o->h_cp) (void)hipHostFree(o->h_cp);
    if (o->h_vec) (void)hipHostFree(o->h_vec);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
}

int ellhip_lowpass_assess_feas(ellhip_lowpass* o, const double* x, double* grad_out, double* beta0, int* has_beta1,
                               double* beta1) {
    return lp_assess_host(o, 0, x, nullptr, grad_out, beta0, has_beta1, beta1, nullptr);
}

int ellhip_lowpass_assess_optim(ellhip_lowpass* o, const double* x, double* gamma_inout, double* grad_out,
                                double* beta0, int* has_beta1, double* beta1, int* shrunk) {
    return lp_assess_host(o, 1, x, gamma_inout, grad_out, beta0, has_beta1, beta1, shrunk);
}

int ellhip_lowpass_state(ellhip_lowpass* o, int32_t* ints7, double* doubles2) {
    if (!o) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(o->device);
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipMemcpy(o->h_ls, o->d_ls, sizeof(LpState), hipMemcpyDeviceToHost));
    if (ints7) {
        ints7[0] = o->h_ls->more_alt;
        ints7[1] = o->h_ls->idx1;
        ints7[2] = o->h_ls->idx2;
        ints7[3] = o->h_ls->idx3;
        ints7[4] = o->h_ls->kmax;
        ints7[5] = o->P.nwpass;
        ints7[6] = o->P.nwstop;
    }
    if (doubles2) {
        doubles2[0] = o->h_ls->fmax;
        doubles2[1] = o->h_ls->sp_sq;
    }
    return 0;
}

int64_t ellhip_lowpass_rows_visited(ellhip_lowpass* o, int reset) {
    if (!o) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(o->device);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(o->h_ls, o->d_ls, sizeof(LpState), hipMemcpyDeviceToHost));
    const long long total = o->h_ls->rows_total;
    if (reset) {
        o->h_ls->rows_total = 0;
        HIPCHK(hipMemcpy(o->d_ls, o->h_ls, sizeof(LpState), hipMemcpyHostToDevice));
    }
    return total;
}

int ellhip_lowpass_get_spectrum(ellhip_lowpass* o, double* out) {
    if (!o || !out) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(o->device);
    HIPCHK(hipMemcpy2D(out, (size_t)o->P.n * sizeof(double), o->d_A, (size_t)o->P.ld * sizeof(double),
                       (size_t)o->P.n * sizeof(double), (size_t)o->P.mdim, hipMemcpyDeviceToHost));
    return 0;
}

int ellhip_lowpass_optim(ellhip_space* s, ellhip_lowpass* o, double* gamma_inout, int64_t max_iters, double tol,
                         double* x_best_out, int* has_best_out, int64_t* niter_out) {
    return lp_drive(s, o, 1, gamma_inout, max_iters, tol, x_best_out, has_best_out, niter_out);
}

int ellhip_lowpass_feas(ellhip_space* s, ellhip_lowpass* o, int64_t max_iters, double tol, double* x_out,
                        int* feasible_out, int64_t* niter_out) {
    return lp_drive(s, o, 0, nullptr, max_iters, tol, x_out, feasible_out, niter_out);
}

}  // extern "C"
