// ellhip_capi.hip -- implementation of the C ABI declared in include/ellhip.h.
//
// Host-side runtime of the engine: owns the device buffers of one search space, stages the caller's
// host vectors through pinned memory, issues the kernels of ell_kernels.hpp / ellstable_kernels.hpp
// on one HIP stream and reads the scalar state back.  No torch types, no CPU compute path: if there
// is no HIP device every entry point that needs one fails (ELLHIP_E_NODEVICE).
#include "../../include/ellhip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ell_kernels.hpp"
#include "ellstable_kernels.hpp"

using namespace ellhip;

namespace {

thread_local std::string g_last_error = "";

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return code;
}

#define HIPCHK(expr)                                              \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return fail(ELLHIP_E_HIP, #expr, _e); \
    } while (0)

struct ProfEvent {
    hipEvent_t a, b;
    int cls;
};

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

}  // namespace

struct ellhip_space {
    int variant = ELLHIP_SPACE_ELL;
    long long n = 0, ld = 0, row0 = 0, nrows = 0;
    int device = 0;
    bool sharded = false;

    double* d_Q = nullptr;       // nrows * ld
    double* d_xc = nullptr;      // n
    double* d_stage = nullptr;   // n doubles (grad) + CutParams
    double* d_gt_own = nullptr;  // n
    double* d_gt = nullptr;      // gt buffer in use (own or caller's)
    double* d_work = nullptr;    // EllStable vectors: w, z, gg, q (4n)
    DevState* d_st = nullptr;

    double* h_stage = nullptr;    // pinned: n doubles + CutParams
    DevState* h_result = nullptr; // pinned

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    int no_defer_trick = 0;
    int use_parallel_cut = 1;
    bool needs_mirror = false;  // caller-supplied non-symmetric matrix, no successful update yet
    bool pending = false;       // update_begin issued, update_end not yet

    // cached scalars (refreshed by every synchronous read-back)
    double kappa = 1.0, tsq = 0.0;

    // device-resident cut queue
    long long qk = 0;
    CutParams* d_qparams = nullptr;
    double* d_qgrads = nullptr;
    int* d_qstatus = nullptr;
    double* d_qtsq = nullptr;

    // per-kernel event timing
    bool profile = false;
    std::vector<ProfEvent> prof_events;
    size_t prof_used = 0;
    double prof_ms[ELLHIP_NKERNEL_CLASSES] = {0};
    long long prof_cnt[ELLHIP_NKERNEL_CLASSES] = {0};

    // launch shape (tunable through the environment for experiments)
    int rw_gemv = 0, unr_gemv = 0, rw_rank1 = 0, unr_rank1 = 0;
    int nt_gemv = 0, nt_rank1 = 0;  // non-temporal cache policy on the Q stream
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            switched = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

size_t stage_bytes(long long n) { return (size_t)n * sizeof(double) + sizeof(CutParams); }

// ---- profiling helpers ---------------------------------------------------------------------
int prof_flush(ellhip_space* s) {
    for (size_t i = 0; i < s->prof_used; ++i) {
        ProfEvent& pe = s->prof_events[i];
        HIPCHK(hipEventSynchronize(pe.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pe.a, pe.b));
        s->prof_ms[pe.cls] += ms;
        s->prof_cnt[pe.cls] += 1;
    }
    s->prof_used = 0;
    return 0;
}

struct ProfScope {
    ellhip_space* s;
    ProfEvent* pe = nullptr;
    ProfScope(ellhip_space* sp, int cls) : s(sp) {
        if (!s->profile) return;
        if (s->prof_used == s->prof_events.size()) {
            if (s->prof_events.size() >= 8192) {
                if (prof_flush(s) != 0) return;
            } else {
                ProfEvent e;
                if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
                e.cls = cls;
                s->prof_events.push_back(e);
            }
        }
        pe = &s->prof_events[s->prof_used++];
        pe->cls = cls;
        (void)hipEventRecord(pe->a, s->stream);
    }
    ~ProfScope() {
        if (pe) (void)hipEventRecord(pe->b, s->stream);
    }
};

// ---- launch-shape selection ------------------------------------------------------------------
// A workgroup covers 4*RW rows; each lane keeps RW*UNR 16-byte loads in flight.  Defaults come from
// in-process A/B sweeps on MI355X (tools/tune_ell.hip, numbers in DESIGN.md):
//   * local Q block fits the 256 MiB Infinity Cache (n = 4096: 128 MiB): keep the default cache
//     policy so the other pass finds Q on-die, 4 rows per workgroup, deep unroll;
//   * larger: the Q stream is touched once per pass, so use the non-temporal policy (GEMV 4.8 ->
//     6.3 TB/s at n = 16384) and more rows per wave.
void pick_shape(ellhip_space* s) {
    const double q_bytes = (double)s->nrows * (double)s->ld * 8.0;
    const bool fits_mall = q_bytes <= 200.0 * 1024 * 1024;
    int rwg, ung, rwr, unr, ntg, ntr;
    if (fits_mall) {
        rwg = 1; ung = 4; rwr = 1; unr = 8; ntg = 0; ntr = 0;
    } else {
        rwg = 4; ung = 4; rwr = 2; unr = 4; ntg = 1; ntr = 1;
    }
    s->rw_gemv = env_int("ELLHIP_GEMV_RW", rwg);
    s->unr_gemv = env_int("ELLHIP_GEMV_UNR", ung);
    s->rw_rank1 = env_int("ELLHIP_RANK1_RW", rwr);
    s->unr_rank1 = env_int("ELLHIP_RANK1_UNR", unr);
    s->nt_gemv = env_int("ELLHIP_GEMV_NT", ntg);
    s->nt_rank1 = env_int("ELLHIP_RANK1_NT", ntr);
}

template <int VEC, bool NT>
int launch_gemv_v(ellhip_space* s, const double* g, double* gt_out) {
    const long long nr = s->nrows;
    const int rw = s->rw_gemv, unr = s->unr_gemv;
    const unsigned grid = (unsigned)((nr + 4LL * rw - 1) / (4LL * rw));
#define GEMV_CASE(RW, UNR)                                                                       \
    if (rw == RW && unr == UNR) {                                                                \
        hipLaunchKernelGGL((k_gemv<RW, UNR, VEC, NT>), dim3(grid), dim3(256), 0, s->stream, s->d_Q,  \
                           s->ld, s->n, nr, g, gt_out, s->d_st);                                 \
        return 0;                                                                                \
    }
    GEMV_CASE(1, 1) GEMV_CASE(1, 2) GEMV_CASE(1, 4) GEMV_CASE(1, 8)
    GEMV_CASE(2, 1) GEMV_CASE(2, 2) GEMV_CASE(2, 4) GEMV_CASE(2, 8)
    GEMV_CASE(4, 1) GEMV_CASE(4, 2) GEMV_CASE(4, 4)
    GEMV_CASE(8, 1) GEMV_CASE(8, 2)
#undef GEMV_CASE
    return fail(ELLHIP_E_INVALID, "unsupported ELLHIP_GEMV_RW/UNR combination");
}

int launch_gemv(ellhip_space* s, const double* g) {
    ProfScope ps(s, 0);
    double* gt_out = s->d_gt + s->row0;
    int rc;
    if (s->n % 2 == 0)
        rc = s->nt_gemv ? launch_gemv_v<2, true>(s, g, gt_out) : launch_gemv_v<2, false>(s, g, gt_out);
    else
        rc = launch_gemv_v<1, false>(s, g, gt_out);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

template <int VEC, bool SCALE, bool NT>
int launch_rank1_v(ellhip_space* s) {
    const long long nr = s->nrows;
    const int rw = s->rw_rank1, unr = s->unr_rank1;
    const unsigned grid = (unsigned)((nr + 4LL * rw - 1) / (4LL * rw));
#define RANK1_CASE(RW, UNR)                                                                          \
    if (rw == RW && unr == UNR) {                                                                    \
        hipLaunchKernelGGL((k_rank1<RW, UNR, VEC, SCALE, NT>), dim3(grid), dim3(256), 0, s->stream,      \
                           s->d_Q, s->d_Q, s->ld, s->n, nr, s->row0, s->d_gt, s->d_st);              \
        return 0;                                                                                    \
    }
    RANK1_CASE(1, 1) RANK1_CASE(1, 2) RANK1_CASE(1, 4) RANK1_CASE(1, 8)
    RANK1_CASE(2, 1) RANK1_CASE(2, 2) RANK1_CASE(2, 4) RANK1_CASE(2, 8)
    RANK1_CASE(4, 1) RANK1_CASE(4, 2) RANK1_CASE(4, 4)
    RANK1_CASE(8, 1) RANK1_CASE(8, 2)
#undef RANK1_CASE
    return fail(ELLHIP_E_INVALID, "unsupported ELLHIP_RANK1_RW/UNR combination");
}

int launch_rank1(ellhip_space* s) {
    if (s->needs_mirror) {
        const unsigned t = (unsigned)((s->n + 31) / 32);
        hipLaunchKernelGGL(k_mirror_lower, dim3(t, t), dim3(256), 0, s->stream, s->d_Q, s->ld, s->n,
                           s->d_st);
        HIPCHK(hipGetLastError());
    }
    ProfScope ps(s, 2);
    int rc;
    const bool even = s->n % 2 == 0;
    const bool nt = even && s->nt_rank1;
    if (s->no_defer_trick)
        rc = even ? (nt ? launch_rank1_v<2, true, true>(s) : launch_rank1_v<2, true, false>(s))
                  : launch_rank1_v<1, true, false>(s);
    else
        rc = even ? (nt ? launch_rank1_v<2, false, true>(s) : launch_rank1_v<2, false, false>(s))
                  : launch_rank1_v<1, false, false>(s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_scalar(ellhip_space* s, const double* g, const CutParams* cp, int queue_mode, int* qst,
                  double* qtsq) {
    ProfScope ps(s, 1);
    EllCalcDev calc = EllCalcDev::make(s->n, s->use_parallel_cut);
    hipLaunchKernelGGL(k_scalar, dim3(1), dim3(1024), 0, s->stream, s->n, g, s->d_gt, s->d_xc, s->d_st,
                       calc, cp, s->no_defer_trick, queue_mode, qst, qtsq);
    HIPCHK(hipGetLastError());
    return 0;
}

// EllStable::update_core as a fixed sequence of launches (ellstable_kernels.hpp).
int ellstable_issue(ellhip_space* s, const double* g_dev, const CutParams* cp_dev, int queue_mode, int* qst,
                    double* qtsq) {
    const long long n = s->n, ld = s->ld;
    const long long nb = (n + SB - 1) / SB;
    double* w = s->d_work;
    double* z = w + n;
    double* gg = z + n;
    double* q = gg + n;
    double* beta2 = q + n;
    hipStream_t st = s->stream;
    {
        ProfScope ps(s, 3);
        HIPCHK(hipMemcpyAsync(w, g_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_st_fwd_first, dim3(1), dim3(64), 0, st, s->d_Q, ld, n, g_dev, w, z, gg, s->d_st);
        for (long long kb = 0; kb + 1 < nb; ++kb) {
            const long long rest = n - (kb + 1) * SB;
            const unsigned grid = (unsigned)((rest + SPANEL - 1) / SPANEL);
            hipLaunchKernelGGL(k_st_fwd_step, dim3(grid), dim3(256), 0, st, s->d_Q, ld, n, kb, w, z, gg, s->d_st);
        }
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps(s, 1);
        EllCalcDev calc = EllCalcDev::make(n, s->use_parallel_cut);
        hipLaunchKernelGGL(k_st_mid, dim3(1), dim3(1024), 0, st, s->d_Q, ld, n, z, gg, q, beta2, s->d_st, calc,
                           cp_dev, queue_mode, qst, qtsq);
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps(s, 4);
        hipLaunchKernelGGL(k_st_bwd_last, dim3(1), dim3(64), 0, st, s->d_Q, ld, n, nb - 1, q, s->d_st);
        for (long long kb = nb - 1; kb >= 1; --kb) {
            const unsigned grid = (unsigned)((kb * SB + SPANEL - 1) / SPANEL);
            hipLaunchKernelGGL(k_st_bwd_step, dim3(grid), dim3(256), 0, st, s->d_Q, ld, n, kb, q, s->d_st);
        }
        const unsigned gx = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_st_xc, dim3(gx < 256 ? gx : 256), dim3(256), 0, st, n, q, s->d_xc, s->d_st);
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps(s, 5);
        hipLaunchKernelGGL(k_st_factor, dim3((unsigned)nb, (unsigned)nb), dim3(256), 0, st, s->d_Q, ld, n, beta2,
                           s->d_st);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---- one cut, phase 1 / phase 2, for either variant ------------------------------------------
int issue_phase1(ellhip_space* s, const double* g_dev) {
    if (s->variant == ELLHIP_SPACE_ELL) return launch_gemv(s, g_dev);
    return 0;  // EllStable does everything in phase 2
}

int issue_phase2(ellhip_space* s, const double* g_dev, const CutParams* cp_dev, int queue_mode, int* qst,
                 double* qtsq) {
    if (s->variant == ELLHIP_SPACE_ELL) {
        int rc = launch_scalar(s, g_dev, cp_dev, queue_mode, qst, qtsq);
        if (rc) return rc;
        return launch_rank1(s);
    }
    return ellstable_issue(s, g_dev, cp_dev, queue_mode, qst, qtsq);
}

int read_back(ellhip_space* s) {
    HIPCHK(hipMemcpyAsync(s->h_result, s->d_st, sizeof(DevState), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->kappa = s->h_result->kappa;
    s->tsq = s->h_result->tsq;
    return 0;
}

int alloc_common(ellhip_space* s) {
    const long long n = s->n;
    HIPCHK(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    HIPCHK(hipMalloc(&s->d_Q, (size_t)s->nrows * (size_t)s->ld * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_xc, (size_t)n * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_stage, stage_bytes(n)));
    HIPCHK(hipMalloc(&s->d_gt_own, (size_t)n * sizeof(double)));
    s->d_gt = s->d_gt_own;
    HIPCHK(hipMalloc(&s->d_st, sizeof(DevState)));
    if (s->variant == ELLHIP_SPACE_ELL_STABLE) HIPCHK(hipMalloc(&s->d_work, (size_t)n * 6 * sizeof(double)));
    HIPCHK(hipHostMalloc(&s->h_stage, stage_bytes(n), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&s->h_result, sizeof(DevState), hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(s->d_gt_own, 0, (size_t)n * sizeof(double), s->stream));
    return 0;
}

int write_state(ellhip_space* s) {
    DevState st;
    memset(&st, 0, sizeof st);
    st.kappa = s->kappa;
    st.tsq = s->tsq;
    st.scale = 1.0;
    st.status = ELLHIP_SUCCESS;
    *s->h_result = st;
    HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

bool host_is_symmetric(const double* mq, long long n) {
    for (long long i = 0; i < n; ++i)
        for (long long j = 0; j < i; ++j)
            if (memcmp(&mq[i * n + j], &mq[j * n + i], sizeof(double)) != 0) return false;
    return true;
}

int create_impl(ellhip_space** out, int variant, long long n, long long row0, long long nrows,
                bool sharded, double kappa, const double* mq, const double* diag, const double* xc,
                int device) {
    if (!out) return fail(ELLHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 1 || nrows < 1 || row0 < 0 || row0 + nrows > n) return fail(ELLHIP_E_INVALID, "bad dimensions");
    if (variant != ELLHIP_SPACE_ELL && variant != ELLHIP_SPACE_ELL_STABLE)
        return fail(ELLHIP_E_INVALID, "unknown variant");
    if (sharded && variant != ELLHIP_SPACE_ELL)
        return fail(ELLHIP_E_INVALID, "EllStable does not shard (replicas only)");
    int ndev = ellhip_device_count();
    if (ndev <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device: the ellipsoid engine has no CPU path");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) return fail(ELLHIP_E_INVALID, "device index out of range");

    ellhip_space* s = new (std::nothrow) ellhip_space();
    if (!s) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    s->variant = variant;
    s->n = n;
    s->row0 = row0;
    s->nrows = nrows;
    s->sharded = sharded;
    s->device = device;
    s->kappa = kappa;
    s->tsq = 0.0;
    // Leading dimension: keep 16-byte row alignment for even n; break the power-of-two row pitch
    // (all rows of a tile on one HBM channel group) with one extra 128-byte line per row.
    s->ld = n;
    if ((n % 512) == 0 && (double)nrows * (double)n * 8.0 > 200.0 * 1024 * 1024) s->ld = n + 16;
    s->ld = n + env_int("ELLHIP_PAD", (int)(s->ld - n));
    if ((n % 2) == 0 && (s->ld % 2) != 0) s->ld += 1;
    if (variant == ELLHIP_SPACE_ELL_STABLE) s->ld = n + (n & 1);  // 16-byte aligned rows for the 2-column lanes
    pick_shape(s);

    DeviceGuard guard(device);
    int rc = alloc_common(s);
    if (rc) {
        ellhip_destroy(s);
        return rc;
    }
    auto bail = [&](int code) {
        ellhip_destroy(s);
        return code;
    };
    // centre
    if (xc) {
        memcpy(s->h_stage, xc, (size_t)n * sizeof(double));
        if (hipMemcpyAsync(s->d_xc, s->h_stage, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s->stream) !=
            hipSuccess)
            return bail(fail(ELLHIP_E_HIP, "upload xc"));
        if (hipStreamSynchronize(s->stream) != hipSuccess) return bail(fail(ELLHIP_E_HIP, "sync"));
    } else {
        if (hipMemsetAsync(s->d_xc, 0, (size_t)n * sizeof(double), s->stream) != hipSuccess)
            return bail(fail(ELLHIP_E_HIP, "memset xc"));
    }
    // matrix
    if (mq) {
        if (hipMemcpy2DAsync(s->d_Q, (size_t)s->ld * sizeof(double), mq, (size_t)n * sizeof(double),
                             (size_t)n * sizeof(double), (size_t)nrows, hipMemcpyHostToDevice,
                             s->stream) != hipSuccess)
            return bail(fail(ELLHIP_E_HIP, "upload mq"));
        if (s->ld != n) {
            // zero the padding columns so they never hold NaNs
            if (hipMemset2DAsync(s->d_Q + n, (size_t)s->ld * sizeof(double), 0, (size_t)(s->ld - n) * sizeof(double),
                                 (size_t)nrows, s->stream) != hipSuccess)
                return bail(fail(ELLHIP_E_HIP, "memset pad"));
        }
        if (variant == ELLHIP_SPACE_ELL && !sharded) s->needs_mirror = !host_is_symmetric(mq, n);
    } else {
        double* d_diag = nullptr;
        if (diag) {
            memcpy(s->h_stage, diag, (size_t)n * sizeof(double));
            if (hipMemcpyAsync(s->d_stage, s->h_stage, (size_t)n * sizeof(double), hipMemcpyHostToDevice,
                               s->stream) != hipSuccess)
                return bail(fail(ELLHIP_E_HIP, "upload diag"));
            d_diag = s->d_stage;
        }
        hipLaunchKernelGGL(k_fill_diag, dim3(2048), dim3(256), 0, s->stream, s->d_Q, s->ld, n, nrows, row0,
                           (const double*)d_diag);
        if (hipGetLastError() != hipSuccess) return bail(fail(ELLHIP_E_HIP, "k_fill_diag"));
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess) return bail(fail(ELLHIP_E_HIP, "sync after upload"));
    rc = write_state(s);
    if (rc) return bail(rc);
    *out = s;
    return 0;
}

int stage_cut(ellhip_space* s, int kind, const double* grad, double b0, int has_b1, double b1) {
    if (!grad) return fail(ELLHIP_E_INVALID, "grad is NULL");
    if (kind < 0 || kind > 2) return fail(ELLHIP_E_INVALID, "bad cut kind");
    const long long n = s->n;
    memcpy(s->h_stage, grad, (size_t)n * sizeof(double));
    CutParams cp;
    cp.kind = kind;
    cp.has_b1 = has_b1 ? 1 : 0;
    cp.b0 = b0;
    cp.b1 = has_b1 ? b1 : 0.0;
    memcpy(reinterpret_cast<char*>(s->h_stage) + (size_t)n * sizeof(double), &cp, sizeof cp);
    HIPCHK(hipMemcpyAsync(s->d_stage, s->h_stage, stage_bytes(n), hipMemcpyHostToDevice, s->stream));
    return 0;
}

const CutParams* stage_params_dev(const ellhip_space* s) {
    return reinterpret_cast<const CutParams*>(reinterpret_cast<const char*>(s->d_stage) +
                                              (size_t)s->n * sizeof(double));
}

void queue_free(ellhip_space* s) {
    if (s->d_qparams) (void)hipFree(s->d_qparams);
    if (s->d_qgrads) (void)hipFree(s->d_qgrads);
    if (s->d_qstatus) (void)hipFree(s->d_qstatus);
    if (s->d_qtsq) (void)hipFree(s->d_qtsq);
    s->d_qparams = nullptr;
    s->d_qgrads = nullptr;
    s->d_qstatus = nullptr;
    s->d_qtsq = nullptr;
    s->qk = 0;
}

}  // namespace

// ================================================================================= C ABI ======

extern "C" {

int ellhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ellhip_last_error(void) { return g_last_error.c_str(); }

const char* ellhip_version(void) { return "ellhip 0.1.0 gfx950"; }

int ellhip_create(ellhip_space** out, int variant, int64_t n, double kappa, const double* mq,
                  const double* diag, const double* xc, int device) {
    return create_impl(out, variant, n, 0, n, false, kappa, mq, diag, xc, device);
}

int ellhip_create_shard(ellhip_space** out, int64_t n, int64_t row0, int64_t nrows, double kappa,
                        const double* mq_rows, const double* diag, const double* xc, int device) {
    return create_impl(out, ELLHIP_SPACE_ELL, n, row0, nrows, true, kappa, mq_rows, diag, xc, device);
}

void ellhip_destroy(ellhip_space* s) {
    if (!s) return;
    DeviceGuard guard(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto& pe : s->prof_events) {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    queue_free(s);
    if (s->d_Q) (void)hipFree(s->d_Q);
    if (s->d_xc) (void)hipFree(s->d_xc);
    if (s->d_stage) (void)hipFree(s->d_stage);
    if (s->d_gt_own) (void)hipFree(s->d_gt_own);
    if (s->d_work) (void)hipFree(s->d_work);
    if (s->d_st) (void)hipFree(s->d_st);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->h_result) (void)hipHostFree(s->h_result);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
}

int ellhip_clone(const ellhip_space* src, ellhip_space** out) {
    if (!src || !out) return fail(ELLHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    ellhip_space* s = new (std::nothrow) ellhip_space();
    if (!s) return fail(ELLHIP_E_NOMEM, "host allocation failed");
    s->variant = src->variant;
    s->n = src->n;
    s->ld = src->ld;
    s->row0 = src->row0;
    s->nrows = src->nrows;
    s->sharded = src->sharded;
    s->device = src->device;
    s->no_defer_trick = src->no_defer_trick;
    s->use_parallel_cut = src->use_parallel_cut;
    s->needs_mirror = src->needs_mirror;
    s->kappa = src->kappa;
    s->tsq = src->tsq;
    s->rw_gemv = src->rw_gemv;
    s->unr_gemv = src->unr_gemv;
    s->rw_rank1 = src->rw_rank1;
    s->unr_rank1 = src->unr_rank1;
    s->nt_gemv = src->nt_gemv;
    s->nt_rank1 = src->nt_rank1;
    DeviceGuard guard(s->device);
    int rc = alloc_common(s);
    if (rc) {
        ellhip_destroy(s);
        return rc;
    }
    // order the copies after everything the source has in flight
    hipError_t e = hipStreamSynchronize(src->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_Q, src->d_Q, (size_t)s->nrows * (size_t)s->ld * sizeof(double),
                           hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_xc, src->d_xc, (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_st, src->d_st, sizeof(DevState), hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
        ellhip_destroy(s);
        return fail(ELLHIP_E_HIP, "clone copy", e);
    }
    *out = s;
    return 0;
}

int ellhip_update_begin(ellhip_space* s, int kind, const double* grad, double beta0, int has_beta1,
                        double beta1) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->pending) return fail(ELLHIP_E_STATE, "update_begin called twice");
    DeviceGuard guard(s->device);
    int rc = stage_cut(s, kind, grad, beta0, has_beta1, beta1);
    if (rc) return rc;
    rc = issue_phase1(s, s->d_stage);
    if (rc) return rc;
    s->pending = true;
    return 0;
}

int ellhip_update_end(ellhip_space* s) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (!s->pending) return fail(ELLHIP_E_STATE, "update_end without update_begin");
    DeviceGuard guard(s->device);
    s->pending = false;
    int rc = issue_phase2(s, s->d_stage, stage_params_dev(s), 0, nullptr, nullptr);
    if (rc) return rc;
    rc = read_back(s);
    if (rc) return rc;
    const int status = s->h_result->status;
    if (status == ELLHIP_SUCCESS) s->needs_mirror = false;
    return status;
}

int ellhip_update(ellhip_space* s, int kind, const double* grad, double beta0, int has_beta1, double beta1) {
    int rc = ellhip_update_begin(s, kind, grad, beta0, has_beta1, beta1);
    if (rc) return rc;
    return ellhip_update_end(s);
}

double ellhip_tsq(const ellhip_space* s) { return s ? s->tsq : 0.0; }
double ellhip_kappa(const ellhip_space* s) { return s ? s->kappa : 0.0; }
int64_t ellhip_ndim(const ellhip_space* s) { return s ? s->n : 0; }

int ellhip_get_xc(const ellhip_space* s, double* xc_out) {
    if (!s || !xc_out) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(s->device);
    HIPCHK(hipMemcpyAsync(xc_out, s->d_xc, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_set_xc(ellhip_space* s, const double* xc) {
    if (!s || !xc) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(s->device);
    HIPCHK(hipStreamSynchronize(s->stream));
    memcpy(s->h_stage, xc, (size_t)s->n * sizeof(double));
    HIPCHK(hipMemcpyAsync(s->d_xc, s->h_stage, (size_t)s->n * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_get_mq(const ellhip_space* s, double* mq_out) {
    if (!s || !mq_out) return fail(ELLHIP_E_INVALID, "NULL argument");
    DeviceGuard guard(s->device);
    HIPCHK(hipMemcpy2DAsync(mq_out, (size_t)s->n * sizeof(double), s->d_Q, (size_t)s->ld * sizeof(double),
                            (size_t)s->n * sizeof(double), (size_t)s->nrows, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_set_no_defer_trick(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->variant != ELLHIP_SPACE_ELL) return fail(ELLHIP_E_INVALID, "no_defer_trick exists on Ell only");
    s->no_defer_trick = flag ? 1 : 0;
    return 0;
}

int ellhip_set_use_parallel_cut(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    s->use_parallel_cut = flag ? 1 : 0;
    return 0;
}

int ellhip_calc(int64_t n, int use_parallel_cut, int kind, double beta0, int has_beta1, double beta1,
                double tsq, double* out3, int device) {
    if (!out3 || n < 1 || kind < 0 || kind > 2) return fail(ELLHIP_E_INVALID, "bad argument");
    if (ellhip_device_count() <= 0) return fail(ELLHIP_E_NODEVICE, "no HIP device");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    DeviceGuard guard(device);
    double* d_out = nullptr;
    HIPCHK(hipMalloc(&d_out, 4 * sizeof(double)));
    CutParams cp;
    cp.kind = kind;
    cp.has_b1 = has_beta1 ? 1 : 0;
    cp.b0 = beta0;
    cp.b1 = has_beta1 ? beta1 : 0.0;
    hipLaunchKernelGGL(k_calc_one, dim3(1), dim3(1), 0, 0, EllCalcDev::make(n, use_parallel_cut), cp, tsq, d_out);
    double h[4];
    hipError_t e = hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ELLHIP_E_HIP, "ellhip_calc", e);
    out3[0] = h[1];
    out3[1] = h[2];
    out3[2] = h[3];
    return (int)h[0];
}

double* ellhip_gt_dev(ellhip_space* s) { return s ? s->d_gt : nullptr; }

int ellhip_set_gt_dev(ellhip_space* s, double* gt_dev) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    if (s->pending) return fail(ELLHIP_E_STATE, "update in flight");
    s->d_gt = gt_dev ? gt_dev : s->d_gt_own;
    return 0;
}

// ---- queue ------------------------------------------------------------------------------------

int ellhip_queue_upload(ellhip_space* s, int64_t k, const int32_t* kinds, const double* grads,
                        const double* beta0, const int32_t* has_beta1, const double* beta1) {
    if (!s || k < 1 || !kinds || !grads || !beta0) return fail(ELLHIP_E_INVALID, "bad argument");
    DeviceGuard guard(s->device);
    HIPCHK(hipStreamSynchronize(s->stream));
    queue_free(s);
    const long long n = s->n;
    std::vector<CutParams> hp((size_t)k);
    for (int64_t i = 0; i < k; ++i) {
        if (kinds[i] < 0 || kinds[i] > 2) return fail(ELLHIP_E_INVALID, "bad cut kind in queue");
        hp[(size_t)i].kind = kinds[i];
        hp[(size_t)i].has_b1 = (has_beta1 && has_beta1[i]) ? 1 : 0;
        hp[(size_t)i].b0 = beta0[i];
        hp[(size_t)i].b1 = (has_beta1 && has_beta1[i] && beta1) ? beta1[i] : 0.0;
    }
    HIPCHK(hipMalloc(&s->d_qparams, (size_t)k * sizeof(CutParams)));
    HIPCHK(hipMalloc(&s->d_qgrads, (size_t)k * (size_t)n * sizeof(double)));
    HIPCHK(hipMalloc(&s->d_qstatus, (size_t)k * sizeof(int)));
    HIPCHK(hipMalloc(&s->d_qtsq, (size_t)k * sizeof(double)));
    HIPCHK(hipMemcpy(s->d_qparams, hp.data(), (size_t)k * sizeof(CutParams), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_qgrads, grads, (size_t)k * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(s->d_qstatus, 0xff, (size_t)k * sizeof(int)));  // -1 = not run yet
    HIPCHK(hipMemset(s->d_qtsq, 0, (size_t)k * sizeof(double)));
    s->qk = k;
    return 0;
}

int ellhip_queue_begin(ellhip_space* s, int64_t index) {
    if (!s || index < 0 || index >= s->qk) return fail(ELLHIP_E_INVALID, "queue index out of range");
    DeviceGuard guard(s->device);
    return issue_phase1(s, s->d_qgrads + (size_t)index * (size_t)s->n);
}

int ellhip_queue_end(ellhip_space* s, int64_t index) {
    if (!s || index < 0 || index >= s->qk) return fail(ELLHIP_E_INVALID, "queue index out of range");
    DeviceGuard guard(s->device);
    return issue_phase2(s, s->d_qgrads + (size_t)index * (size_t)s->n, s->d_qparams + index, 1,
                        s->d_qstatus + index, s->d_qtsq + index);
}

int ellhip_queue_run(ellhip_space* s, int64_t first, int64_t count) {
    if (!s || first < 0 || count < 0 || first + count > s->qk) return fail(ELLHIP_E_INVALID, "queue range");
    for (int64_t i = first; i < first + count; ++i) {
        int rc = ellhip_queue_begin(s, i);
        if (rc) return rc;
        rc = ellhip_queue_end(s, i);
        if (rc) return rc;
    }
    return 0;
}

int ellhip_queue_results(ellhip_space* s, int32_t* status_out, double* tsq_out) {
    if (!s || s->qk < 1) return fail(ELLHIP_E_INVALID, "no queue");
    DeviceGuard guard(s->device);
    int rc = read_back(s);
    if (rc) return rc;
    if (status_out) HIPCHK(hipMemcpy(status_out, s->d_qstatus, (size_t)s->qk * sizeof(int), hipMemcpyDeviceToHost));
    if (tsq_out) HIPCHK(hipMemcpy(tsq_out, s->d_qtsq, (size_t)s->qk * sizeof(double), hipMemcpyDeviceToHost));
    if (s->needs_mirror && status_out) {
        for (int64_t i = 0; i < s->qk; ++i)
            if (status_out[i] == ELLHIP_SUCCESS) {
                s->needs_mirror = false;
                break;
            }
    }
    // a halted queue stays halted until the next upload; clear the flag so direct updates work again
    if (s->h_result->halted) {
        s->h_result->halted = 0;
        HIPCHK(hipMemcpyAsync(s->d_st, s->h_result, sizeof(DevState), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return 0;
}

// ---- streams / timing -------------------------------------------------------------------------

int ellhip_set_stream(ellhip_space* s, void* hip_stream) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    HIPCHK(hipStreamSynchronize(s->stream));
    s->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : s->own_stream;
    return 0;
}

int ellhip_synchronize(ellhip_space* s) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

int ellhip_profile_enable(ellhip_space* s, int flag) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    s->profile = flag != 0;
    return 0;
}

int ellhip_profile_read(ellhip_space* s, double* ms_out, int64_t* count_out) {
    if (!s) return fail(ELLHIP_E_INVALID, "NULL handle");
    DeviceGuard guard(s->device);
    int rc = prof_flush(s);
    if (rc) return rc;
    for (int i = 0; i < ELLHIP_NKERNEL_CLASSES; ++i) {
        if (ms_out) ms_out[i] = s->prof_ms[i];
        if (count_out) count_out[i] = s->prof_cnt[i];
        s->prof_ms[i] = 0.0;
        s->prof_cnt[i] = 0;
    }
    return 0;
}

}  // extern "C"
