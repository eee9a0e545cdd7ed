// batch_capi.inc.hpp -- C ABI of the batched small-n engine (include/ellhip_batch.h).  Included at the end of
// ellhip_capi.hip.
#include "../../include/ellhip_batch.h"

#include "batch_kernels.hpp"

struct ellhip_batch {
    int device = 0;
    long long B = 0;
    int n = 0;
    int T = 64;    // threads per workgroup
    int epw = 1;   // ellipsoids per workgroup
    size_t lds_bytes = 0;
    int no_defer_trick = 0;
    int use_parallel_cut = 1;
    double* d_Q = nullptr;
    double* d_xc = nullptr;
    double* d_kappa = nullptr;
    double* d_tsq = nullptr;
    hipStream_t stream = nullptr;
};

namespace {

int batch_alloc(ellhip_batch* h) {
    const size_t B = (size_t)h->B, n = (size_t)h->n;
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&h->d_Q, B * n * n * sizeof(double)));
    HIPCHK(hipMalloc(&h->d_xc, B * n * sizeof(double)));
    HIPCHK(hipMalloc(&h->d_kappa, B * sizeof(double)));
    HIPCHK(hipMalloc(&h->d_tsq, B * sizeof(double)));
    HIPCHK(fill_now(h->d_tsq, 0, B * sizeof(double), h->stream));
    return 0;
}

int batch_shape(ellhip_batch* h) {
    h->T = h->n <= 64 ? 256 : 128;
    if (g_defaults.batch_threads > 0) h->T = g_defaults.batch_threads;  // ELLHIP_OPT_BATCH_THREADS
    if (h->T != 64 && h->T != 128 && h->T != 256) return fail(ELLHIP_E_INVALID, "ELLHIP_OPT_BATCH_THREADS must be 64, 128 or 256");
    if (h->T < h->n) h->T = 128;
    h->epw = h->T / h->n;
    if (h->epw > 64) h->epw = 64;  // one wave runs the scalar stage, one lane per ellipsoid
    const size_t per_bytes = batch_lds_doubles(h->n) * sizeof(double);
    while (h->epw > 1 && (size_t)h->epw * per_bytes > 64 * 1024) h->epw -= 1;  // keep >= 2 workgroups per CU
    if (h->epw < 1) h->epw = 1;
    h->lds_bytes = (size_t)h->epw * per_bytes;
    if (h->lds_bytes > 160 * 1024) return fail(ELLHIP_E_INVALID, "batched engine: n too large for LDS");
    // more than the default 64 KiB of dynamic LDS needs an opt-in per kernel AND per device (a function attribute
    // belongs to the device's copy of the code object).  It is only ever RAISED, so that a handle with a larger
    // footprint created earlier keeps launching after a smaller one has been set up; the high-water marks are kept
    // per (device, block size).
    constexpr int MAXDEV = 64;
    static std::atomic<int> granted[MAXDEV][3];
    const int slot = h->T == 64 ? 0 : (h->T == 128 ? 1 : 2);
    const int dev = (h->device >= 0 && h->device < MAXDEV) ? h->device : -1;
    if (dev < 0 || (int)h->lds_bytes > granted[dev][slot].load()) {   // (a device beyond the table: set it every time)
#define BATCH_ATTR(TT)                                                                         \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_batch_update<TT>),             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes))
        if (h->T == 64) BATCH_ATTR(64);
        else if (h->T == 128) BATCH_ATTR(128);
        else BATCH_ATTR(256);
#undef BATCH_ATTR
        if (dev >= 0) {
            int seen = granted[dev][slot].load();
            while (seen < (int)h->lds_bytes && !granted[dev][slot].compare_exchange_weak(seen, (int)h->lds_bytes)) {}
        }
    }
    return 0;
}

int batch_launch(ellhip_batch* h, long long K, const int* kinds, const double* grads, const double* b0, const int* hb1,
                 const double* b1, int* status, double* tsq_out) {
    BatchParams P;
    P.B = h->B;
    P.n = h->n;
    P.pitch = batch_pitch(h->n);
    P.epw = h->epw;
    P.K = (int)K;
    P.no_defer_trick = h->no_defer_trick;
    const unsigned grid = (unsigned)((h->B + h->epw - 1) / h->epw);
    const EllCalcDev calc = EllCalcDev::make(h->n, h->use_parallel_cut);
#define BATCH_GO(TT)                                                                                             \
    hipLaunchKernelGGL(k_batch_update<TT>, dim3(grid), dim3(TT), h->lds_bytes, h->stream, P, h->d_Q, h->d_xc,    \
                       h->d_kappa, h->d_tsq, kinds, grads, b0, hb1, b1, status, tsq_out, calc)
    if (h->T == 64) BATCH_GO(64);
    else if (h->T == 128) BATCH_GO(128);
    else BATCH_GO(256);
#undef BATCH_GO
    HIPCHK(hipGetLastError());
    return 0;
}

int batch_new(ellhip_batch** out, long long B, long long n, int device) {
    if (!out) return fail(ELLHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (B < 1 || n < 1 || n > BATCH_NMAX) return fail(ELLHIP_E_INVALID, "batched engine: need B >= 1 and 1 <= n <= 128");
    if ((double)B * (double)n * (double)n * 8.0 > 200e9) return fail(ELLHIP_E_NOMEM, "batched engine: B*n*n too large");
    const int ndev = ellhip_device_count();
    if (ndev <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device: the batched engine has no CPU path");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) return fail(ELLHIP_E_INVALID, "device index out of range");
    ellhip_batch* h = new (std::nothrow) ellhip_batch();
    if (!h) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    h->device = device;
    h->B = B;
    h->n = (int)n;
    DeviceGuard guard(device);
    int rc = batch_shape(h);
    if (!rc) rc = batch_alloc(h);
    if (rc) {
        ellhip_batch_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

}  // namespace

extern "C" {

int ellhip_batch_create(ellhip_batch** out, int64_t B, int64_t n, const double* kappa, const double* mq,
                        const double* diag, const double* xc, int device) {
    ellhip_batch* h = nullptr;
    int rc = batch_new(&h, B, n, device);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    auto bail = [&](int code) {
        ellhip_batch_destroy(h);
        return code;
    };
    const size_t sB = (size_t)B, sn = (size_t)n;
    hipError_t e = hipSuccess;
    if (mq) {
        e = hipMemcpy(h->d_Q, mq, sB * sn * sn * sizeof(double), hipMemcpyHostToDevice);
    } else {
        double* d_diag = nullptr;
        if (diag) {
            e = hipMalloc(&d_diag, sB * sn * sizeof(double));
            if (e == hipSuccess) e = hipMemcpy(d_diag, diag, sB * sn * sizeof(double), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_batch_fill, dim3(1024), dim3(256), 0, h->stream, h->d_Q, (long long)B, (int)n,
                               (const double*)d_diag);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
        if (d_diag) (void)hipFree(d_diag);
    }
    if (e != hipSuccess) return bail(fail(ELLHIP_E_HIP, "batch create: matrix", e));
    if (xc) e = hipMemcpy(h->d_xc, xc, sB * sn * sizeof(double), hipMemcpyHostToDevice);
    else e = fill_now(h->d_xc, 0, sB * sn * sizeof(double), h->stream);
    if (e != hipSuccess) return bail(fail(ELLHIP_E_HIP, "batch create: xc", e));
    std::vector<double> ones;
    if (!kappa) {
        ones.assign(sB, 1.0);
        kappa = ones.data();
    }
    e = hipMemcpy(h->d_kappa, kappa, sB * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(fail(ELLHIP_E_HIP, "batch create: kappa", e));
    *out = h;
    return 0;
}

int ellhip_batch_from_space(ellhip_batch** out, const ellhip_space* space_c, int64_t B) {
    if (!space_c) return fail(ELLHIP_E_INVALID, "NULL handle");
    ellhip_space* s = const_cast<ellhip_space*>(space_c);
    if (s->variant != ELLHIP_SPACE_ELL || s->sharded)
        return fail(ELLHIP_E_INVALID, "batched engine: clones of an unsharded Ell only");
    DeviceGuard guard(s->device);
    int rc = make_q_current(s);  // recorded (deferred) shrinks belong to the matrix that is cloned
    if (rc) return rc;
    rc = read_back(s);
    if (rc) return rc;
    ellhip_batch* h = nullptr;
    rc = batch_new(&h, B, s->n, s->device);
    if (rc) return rc;
    h->no_defer_trick = s->no_defer_trick;
    h->use_parallel_cut = s->use_parallel_cut;
    const size_t n = (size_t)s->n;
    hipError_t e = hipSuccess;
    std::vector<double> kap((size_t)B, s->kappa), ts((size_t)B, s->tsq);
    for (int64_t b = 0; b < B && e == hipSuccess; ++b) {
        e = hipMemcpy2DAsync(h->d_Q + (size_t)b * n * n, n * sizeof(double), s->d_Q, (size_t)s->ld * sizeof(double),
                             n * sizeof(double), n, hipMemcpyDeviceToDevice, s->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(h->d_xc + (size_t)b * n, s->d_xc, n * sizeof(double), hipMemcpyDeviceToDevice, s->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess) e = hipMemcpy(h->d_kappa, kap.data(), (size_t)B * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_tsq, ts.data(), (size_t)B * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ellhip_batch_destroy(h);
        return fail(ELLHIP_E_HIP, "batch from_space", e);
    }
    *out = h;
    return 0;
}

void ellhip_batch_destroy(ellhip_batch* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_Q) (void)hipFree(h->d_Q);
    if (h->d_xc) (void)hipFree(h->d_xc);
    if (h->d_kappa) (void)hipFree(h->d_kappa);
    if (h->d_tsq) (void)hipFree(h->d_tsq);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ellhip_batch_update_dev(ellhip_batch* h, int64_t K, const int32_t* kinds_dev, const double* grads_dev,
                            const double* beta0_dev, const int32_t* has_beta1_dev, const double* beta1_dev,
                            int32_t* status_out_dev, double* tsq_out_dev) {
    if (!h || K < 1 || K > (1 << 20) || !kinds_dev || !grads_dev || !beta0_dev || !has_beta1_dev || !beta1_dev ||
        !status_out_dev)
        return fail(ELLHIP_E_INVALID, "bad argument");
    DeviceGuard guard(h->device);
    return batch_launch(h, K, kinds_dev, grads_dev, beta0_dev, has_beta1_dev, beta1_dev, status_out_dev, tsq_out_dev);
}

int ellhip_batch_update(ellhip_batch* h, int64_t K, const int32_t* kinds, const double* grads, const double* beta0,
                        const int32_t* has_beta1, const double* beta1, int32_t* status_out, double* tsq_out) {
    if (!h || K < 1 || K > (1 << 20) || !kinds || !grads || !beta0 || !status_out)
        return fail(ELLHIP_E_INVALID, "bad argument");
    DeviceGuard guard(h->device);
    const size_t KB = (size_t)K * (size_t)h->B, n = (size_t)h->n;
    for (size_t c = 0; c < KB; ++c)
        if (kinds[c] < 0 || kinds[c] > 2) return fail(ELLHIP_E_INVALID, "bad cut kind");
    std::vector<int32_t> hb(KB, 0);
    std::vector<double> b1(KB, 0.0);
    if (has_beta1 && beta1)
        for (size_t c = 0; c < KB; ++c)
            if (has_beta1[c]) {
                hb[c] = 1;
                b1[c] = beta1[c];
            }
    int *d_kinds = nullptr, *d_hb = nullptr, *d_status = nullptr;
    double *d_g = nullptr, *d_b0 = nullptr, *d_b1 = nullptr, *d_tsq = nullptr;
    hipError_t e = hipMalloc(&d_kinds, KB * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&d_hb, KB * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&d_status, KB * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&d_g, KB * n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_b0, KB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_b1, KB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_tsq, KB * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_kinds, kinds, KB * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_hb, hb.data(), KB * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_g, grads, KB * n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b0, beta0, KB * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b1, b1.data(), KB * sizeof(double), hipMemcpyHostToDevice);
    int rc = 0;
    if (e == hipSuccess) {
        rc = batch_launch(h, K, d_kinds, d_g, d_b0, d_hb, d_b1, d_status, d_tsq);
        if (!rc) e = hipStreamSynchronize(h->stream);
    }
    if (!rc && e == hipSuccess) e = hipMemcpy(status_out, d_status, KB * sizeof(int), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && tsq_out) e = hipMemcpy(tsq_out, d_tsq, KB * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_kinds);
    (void)hipFree(d_hb);
    (void)hipFree(d_status);
    (void)hipFree(d_g);
    (void)hipFree(d_b0);
    (void)hipFree(d_b1);
    (void)hipFree(d_tsq);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? ELLHIP_E_NOMEM : ELLHIP_E_HIP, "ellhip_batch_update", e);
    return 0;
}

int ellhip_batch_synchronize(ellhip_batch* h) {
    if (!h) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(h->device);
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

void* ellhip_batch_stream(ellhip_batch* h) { return h ? (void*)h->stream : nullptr; }

#define BATCH_GET(NAME, FIELD, COUNT)                                                                   \
    int NAME(ellhip_batch* h, double* out) {                                                            \
        if (!h || !out) return fail(ELLHIP_E_INVALID, "NULL argument");                                 \
        DeviceGuard guard(h->device);                                                                   \
        HIPCHK(hipStreamSynchronize(h->stream));                                                        \
        HIPCHK(hipMemcpy(out, h->FIELD, (size_t)(COUNT) * sizeof(double), hipMemcpyDeviceToHost));      \
        return 0;                                                                                       \
    }
BATCH_GET(ellhip_batch_get_xc, d_xc, h->B * h->n)
BATCH_GET(ellhip_batch_get_mq, d_Q, h->B * h->n * h->n)
BATCH_GET(ellhip_batch_get_kappa, d_kappa, h->B)
BATCH_GET(ellhip_batch_get_tsq, d_tsq, h->B)
#undef BATCH_GET

int ellhip_batch_set_xc(ellhip_batch* h, const double* xc) {
    if (!h || !xc) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(h->device);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_xc, xc, (size_t)h->B * (size_t)h->n * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int64_t ellhip_batch_size(const ellhip_batch* h) { return h ? h->B : 0; }
int64_t ellhip_batch_ndim(const ellhip_batch* h) { return h ? h->n : 0; }

int ellhip_batch_set_no_defer_trick(ellhip_batch* h, int flag) {
    if (!h) return fail(ELLHIP_E_INVALID, "NULL handle");
    h->no_defer_trick = flag ? 1 : 0;
    return 0;
}

int ellhip_batch_set_use_parallel_cut(ellhip_batch* h, int flag) {
    if (!h) return fail(ELLHIP_E_INVALID, "NULL handle");
    h->use_parallel_cut = flag ? 1 : 0;
    return 0;
}

}  // extern "C"
