// lmi_capi.inc.hpp -- C ABI of the device-side LDLTMgr / LMI oracles (include/ellhip_lmi.h).  Included at the end
// of ellhip_capi.hip.
#include "../../include/ellhip_lmi.h"

#include "lmi_kernels.hpp"

struct ellhip_lmi {
    int device = 0;
    int mode = 0;  // 0 LMIOracle, 1 LMI0Oracle
    long long n = 0, m = 0;
    double* d_F = nullptr;
    double* d_B = nullptr;
    double* d_A = nullptr;   // F(x), lower triangle, formed lazily
    double* d_S = nullptr;   // LDLTMgr::storage
    double* d_x = nullptr;
    double* d_v = nullptr;   // wit
    double* d_g = nullptr;
    double* d_partial = nullptr;
    LmiState* d_st = nullptr;
    LmiState* h_st = nullptr;
    hipStream_t stream = nullptr;
    bool factored = false;
};

extern "C" {

void ellhip_lmi_destroy(ellhip_lmi* o) {
    if (!o) return;
    DeviceGuard guard(o->device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    for (double* p : {o->d_F, o->d_B, o->d_A, o->d_S, o->d_x, o->d_v, o->d_g, o->d_partial})
        if (p) (void)hipFree(p);
    if (o->d_st) (void)hipFree(o->d_st);
    if (o->h_st) (void)hipHostFree(o->h_st);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
}

int ellhip_lmi_create(ellhip_lmi** out, int64_t n, int64_t m, const double* mat_f, const double* mat_b, int device) {
    if (!out) return fail(ELLHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (m < 1 || m > ELLHIP_LMI_MMAX || n < 0) return fail(ELLHIP_E_INVALID, "lmi: need 1 <= m <= 8192 and n >= 0");
    if (n > 0 && !mat_f) return fail(ELLHIP_E_INVALID, "lmi: mat_f is NULL");
    if (n == 0 && !mat_b) return fail(ELLHIP_E_INVALID, "lmi: nothing to factor");
    const int ndev = ellhip_device_count();
    if (ndev <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device: the LMI oracle has no CPU path");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) return fail(ELLHIP_E_INVALID, "device index out of range");
    ellhip_lmi* o = new (std::nothrow) ellhip_lmi();
    if (!o) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    o->device = device;
    o->mode = mat_b ? 0 : 1;
    o->n = n;
    o->m = m;
    DeviceGuard guard(device);
    const size_t mm = (size_t)m * (size_t)m * sizeof(double);
    hipError_t e = hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking);
    if (e == hipSuccess && n > 0) e = hipMalloc(&o->d_F, (size_t)n * mm);
    if (e == hipSuccess && mat_b) e = hipMalloc(&o->d_B, mm);
    if (e == hipSuccess) e = hipMalloc(&o->d_A, mm);
    if (e == hipSuccess) e = hipMalloc(&o->d_S, mm);
    if (e == hipSuccess) e = hipMalloc(&o->d_x, (size_t)(n > 0 ? n : 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&o->d_v, (size_t)m * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&o->d_g, (size_t)(n > 0 ? n : 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&o->d_partial, (size_t)(n > 0 ? n : 1) * LMI_QUAD_CHUNKS * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&o->d_st, sizeof(LmiState));
    if (e == hipSuccess) e = hipHostMalloc(&o->h_st, sizeof(LmiState), hipHostMallocDefault);
    if (e == hipSuccess && n > 0) e = hipMemcpy(o->d_F, mat_f, (size_t)n * mm, hipMemcpyHostToDevice);
    if (e == hipSuccess && mat_b) e = hipMemcpy(o->d_B, mat_b, mm, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = fill_now(o->d_S, 0, mm, o->stream);
    if (e == hipSuccess) e = fill_now(o->d_A, 0, mm, o->stream);
    if (e == hipSuccess) e = fill_now(o->d_v, 0, (size_t)m * sizeof(double), o->stream);
    if (e == hipSuccess) e = fill_now(o->d_st, 0, sizeof(LmiState), o->stream);
    if (e != hipSuccess) {
        ellhip_lmi_destroy(o);
        return fail(e == hipErrorOutOfMemory ? ELLHIP_E_NOMEM : ELLHIP_E_HIP, "lmi create", e);
    }
    *out = o;
    return 0;
}

int ellhip_lmi_assess_feas(ellhip_lmi* o, const double* x, double* g_out, double* ep_out) {
    if (!o) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (o->n > 0 && (!x || !g_out)) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(o->device);
    hipStream_t st = o->stream;
    const long long m = o->m, n = o->n;
    if (n > 0) HIPCHK(hipMemcpyAsync(o->d_x, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ldlt_clear, dim3(1024), dim3(256), 0, st, o->d_S, m, o->d_st);
    for (long long k0 = 0; k0 < m; k0 += LMI_NB) {
        if (k0 % LMI_FORM_W == 0)
            hipLaunchKernelGGL(k_lmi_form, dim3((unsigned)(m - k0)), dim3(LMI_FORM_W), 0, st, (const double*)o->d_F,
                               (const double*)o->d_B, (const double*)o->d_x, o->d_A, m, n, k0, o->mode,
                               (const LmiState*)o->d_st);
        hipLaunchKernelGGL(k_ldlt_diag, dim3(1), dim3(64), 0, st, (const double*)o->d_A, o->d_S, m, k0, o->d_st);
        const long long k1 = k0 + LMI_NB;
        if (k1 < m) {
            hipLaunchKernelGGL(k_ldlt_panel, dim3((unsigned)((m - k1 + 127) / 128)), dim3(128), 0, st,
                               (const double*)o->d_A, o->d_S, m, k0, (const LmiState*)o->d_st);
            const unsigned tiles = (unsigned)((m - k1 + 63) / 64);
            hipLaunchKernelGGL(k_ldlt_update, dim3(tiles, tiles), dim3(256), 0, st, o->d_S, m, k0,
                               (const LmiState*)o->d_st);
        }
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_lmi_witness, dim3(1), dim3(1024), 0, st, (const double*)o->d_S, m, o->d_v,
                       (const LmiState*)o->d_st);
    if (n > 0) {
        hipLaunchKernelGGL(k_lmi_quad, dim3((unsigned)n, LMI_QUAD_CHUNKS), dim3(256), 0, st, (const double*)o->d_F, m,
                           (const double*)o->d_v, o->d_partial, (const LmiState*)o->d_st);
        hipLaunchKernelGGL(k_lmi_quad_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                           (const double*)o->d_partial, o->d_g, o->mode, (const LmiState*)o->d_st);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(o->h_st, o->d_st, sizeof(LmiState), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    o->factored = true;
    if (o->h_st->pos1 == 0) return 0;
    if (n > 0) HIPCHK(hipMemcpy(g_out, o->d_g, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (ep_out) *ep_out = o->h_st->ep;
    return 1;
}

int ellhip_lmi_pos(ellhip_lmi* o, int64_t* pos2) {
    if (!o || !pos2) return fail(ELLHIP_E_INVALID, "NULL argument");
    pos2[0] = 0;  // `factor` never moves start (ldlt_mgr.rs:30)
    pos2[1] = o->factored ? o->h_st->pos1 : 0;
    return 0;
}

int ellhip_lmi_get_witness(ellhip_lmi* o, double* out) {
    if (!o || !out) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(o->device);
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipMemcpy(out, o->d_v, (size_t)o->m * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int ellhip_lmi_get_storage(ellhip_lmi* o, double* out) {
    if (!o || !out) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(o->device);
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipMemcpy(out, o->d_S, (size_t)o->m * (size_t)o->m * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int ellhip_lmi_sqrt(ellhip_lmi* o, double* r_out) {
    if (!o || !r_out) return fail(ELLHIP_E_INVALID, "NULL argument");
    if (!o->factored || o->h_st->pos1 != 0) return fail(ELLHIP_E_STATE, "sqrt called on a non-SPD matrix");
    DeviceGuard guard(o->device);
    // d_A is free once the factorisation is done: use it for R
    hipLaunchKernelGGL(k_ldlt_sqrt, dim3(1024), dim3(256), 0, o->stream, (const double*)o->d_S, o->m, o->d_A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(o->stream));
    HIPCHK(hipMemcpy(r_out, o->d_A, (size_t)o->m * (size_t)o->m * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
