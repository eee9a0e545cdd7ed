// symv_layout.hip -- round-3 experiment ("next" item 3): does the STORAGE LAYOUT / access pattern bound the
// lower-triangle GEMV k_symv (0.70-0.73 of the 8 TB/s peak where the full-row GEMV of the same matrix reaches 0.86)?
// Every variant computes the same y = Q g through the lower triangle of a symmetric Q (4 n^2 bytes) and is checked
// against the full-row GEMV.  Variants (tile = H rows x SEG columns, one workgroup of 256 threads per tile):
//   prod      the production kernel k_symv<2> (ell_kernels.hpp), row-major with pitch ld            -- baseline (a)
//   x rm      this file's tile body on the same row-major matrix (64 x 2048 and 128 x 1024 = (c))
//   x packed  the same tile body on a TILE-PACKED copy: every tile's H x SEG elements contiguous     -- (b)
//   +rot      row blocks of a tile visited starting at a tile-dependent offset (tiles that progress in lock step do
//             not hit the same DRAM offsets)
//   red8      row sums of 8 rows reduced together (10 shuffles per 8 rows instead of 48)
// plus two read-only streaming references over the same bytes (contiguous 4 n^2 bytes; full-row GEMV over 8 n^2).
// Usage: symv_layout [n] [rounds] [pad]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../ellalgo-rs_amd/csrc/ell_kernels.hpp"

using namespace ellhip;

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

__global__ void k_fill_sym(double* Q, long long ld, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n * ld; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld, c = i - r * ld;
        if (c >= n) { Q[i] = 0.0; continue; }
        const unsigned long long lo = r < c ? r : c, hi = r < c ? c : r;
        unsigned long long h = (hi * 0x9E3779B97F4A7C15ull) ^ (lo * 0xBF58476D1CE4E5B9ull);
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        Q[i] = (double)(h & 0xFFFFF) / 1048576.0 - 0.5 + (r == c ? 2.0 : 0.0);
    }
}

// number of active tiles in the segments before J, and the first active strip of segment J (SEG a multiple of H)
template <int H, int SEG>
__host__ __device__ inline long long tile_prefix(long long nstrips, long long J) {
    return J * nstrips - (long long)(SEG / H) * (J * (J - 1) / 2);
}

// row-major -> tile-packed: tile (I, J) of the lower triangle (active: J*SEG <= I*H + H - 1) at Qp + id * H * SEG
template <int H, int SEG>
__global__ void k_pack(const double* Q, long long ld, long long n, double* Qp) {
    const long long nstrips = n / H;
    const long long I = blockIdx.x, J = blockIdx.y;
    if (J * SEG > I * H + H - 1) return;
    const long long id = tile_prefix<H, SEG>(nstrips, J) + (I - J * (SEG / H));
    double* dst = Qp + id * (long long)H * SEG;
    for (int idx = threadIdx.x; idx < H * SEG; idx += blockDim.x) {
        const int r = idx / SEG, c = idx - r * SEG;
        dst[idx] = Q[(I * H + r) * ld + J * SEG + c];
    }
}

template <int H, int SEG>
__global__ void k_check_reduce(long long n, const double* rowpart, const double* colpart, double* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (long long J = 0; J <= i / SEG; ++J) s += rowpart[J * n + i];
    for (long long I = i / H; I < n / H; ++I) s += colpart[I * n + i];
    y[i] = s;
}

__device__ __forceinline__ double shx(double v, int m) { return __shfl_xor(v, m, 64); }

// sums of 8 rows over the 64 lanes: 10 shuffles; lane l ends with the wave's sum of row ((l >> 5) & 1) * 4 + ((l >> 4) & 1) * 2 + ((l >> 3) & 1)
__device__ __forceinline__ double reduce8(const double (&a)[8], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    double k4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = b5 ? a[4 + i] : a[i], send = b5 ? a[i] : a[4 + i];
        k4[i] = keep + shx(send, 32);
    }
    double k2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = b4 ? k4[2 + i] : k4[i], send = b4 ? k4[i] : k4[2 + i];
        k2[i] = keep + shx(send, 16);
    }
    double v = (b3 ? k2[1] : k2[0]) + shx(b3 ? k2[0] : k2[1], 8);
    v += shx(v, 4);
    v += shx(v, 2);
    v += shx(v, 1);
    return v;
}

// One tile.  tb: address of the tile's element (row r0, column c0); ts: its row pitch.  RED8: rows in blocks of 8.
template <int H, int SEG, int RW, bool RED8, bool NT, bool FMA = false>
__device__ __forceinline__ void tile_x(const double* __restrict__ tb, long long ts, long long n, long long r0, long long c0,
                                       const double* __restrict__ g, double* __restrict__ rowpart, double* __restrict__ colpart,
                                       long long I, long long J, int rot, double (*red)[H]) {
    constexpr int NCH = SEG / 512;
    constexpr int RB = RED8 ? 8 : RW;       // rows per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool full = c0 + SEG - 1 < r0;
    long long ck[NCH];
    double2_t gc[NCH], accc[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        ck[k] = c0 + 512 * k + 2 * (long long)threadIdx.x;
        gc[k] = (ck[k] <= r0 + H - 1) ? *reinterpret_cast<const double2_t*>(g + ck[k]) : double2_t{0.0, 0.0};
        accc[k] = double2_t{0.0, 0.0};
    }
    const double* tcol = tb + 2 * (long long)threadIdx.x;   // + 512 k + lr * ts
    for (int b = 0; b < H / RB; ++b) {
        const int rb = ((b + rot) % (H / RB)) * RB;
        double a8[RB];
#pragma unroll
        for (int sub = 0; sub < RB / RW; ++sub) {
            double2_t q[RW][NCH];
            double gr[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int lr = rb + sub * RW + r;
                gr[r] = g[r0 + lr];
                const double* row = tcol + (long long)lr * ts;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if (full || ck[k] <= r0 + lr) q[r][k] = ld_stream<NT, double2_t>(row + 512 * k);
                    else q[r][k] = double2_t{0.0, 0.0};
                }
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const long long rr = r0 + rb + sub * RW + r;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    double qx = q[r][k].x, qy = q[r][k].y;
                    if (!full && ck[k] + 1 > rr) qy = 0.0;
                    const double cx = (full || ck[k] < rr) ? qx : 0.0;
                    const double cy = (full || ck[k] + 1 < rr) ? qy : 0.0;
                    if (FMA) {
                        acc = __builtin_fma(qx, gc[k].x, acc);
                        acc = __builtin_fma(qy, gc[k].y, acc);
                        accc[k].x = __builtin_fma(cx, gr[r], accc[k].x);
                        accc[k].y = __builtin_fma(cy, gr[r], accc[k].y);
                    } else {
                        acc += qx * gc[k].x;
                        acc += qy * gc[k].y;
                        accc[k].x += cx * gr[r];
                        accc[k].y += cy * gr[r];
                    }
                }
                a8[sub * RW + r] = acc;
            }
        }
        if constexpr (RED8) {
            const double v = reduce8(a8, lane);
            if ((lane & 7) == 0) red[wave][rb + (lane >> 3)] = v;
        } else {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const double s = wave_allreduce_sum(a8[r]);
                if (lane == 0) red[wave][rb + r] = s;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < H) {
        const int r = threadIdx.x;
        rowpart[J * n + r0 + r] = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k)
        if (ck[k] <= r0 + H - 1) *reinterpret_cast<double2_t*>(colpart + I * n + ck[k]) = accc[k];
}

template <int H, int SEG, int RW, bool RED8, bool PACKED, bool ROT, bool FMA = false>
__global__ __launch_bounds__(256) void k_symvx(const double* __restrict__ Q, long long ld, long long n, const double* __restrict__ g,
                                               double* __restrict__ rowpart, double* __restrict__ colpart) {
    __shared__ double red[4][H];
    const long long nstrips = n / H;
    const long long I = (long long)gridDim.x - 1 - blockIdx.x, J = blockIdx.y;
    const long long r0 = I * H, c0 = J * SEG;
    if (c0 > r0 + H - 1) return;
    const double* tb;
    long long ts;
    if (PACKED) {
        tb = Q + (tile_prefix<H, SEG>(nstrips, J) + (I - J * (SEG / H))) * (long long)H * SEG;
        ts = SEG;
    } else {
        tb = Q + r0 * ld + c0;
        ts = ld;
    }
    const int rot = ROT ? (int)((I * 5 + J * 3) & 1023) : 0;
    tile_x<H, SEG, RW, RED8, true, FMA>(tb, ts, n, r0, c0, g, rowpart, colpart, I, J, rot, red);
}

// plain contiguous read of `count` double2 (sum into out so that nothing is optimised away)
template <int UNR>
__global__ __launch_bounds__(256) void k_stream_read(const double2_t* __restrict__ x, size_t count, double* out) {
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; i < count; i += stride) {
        double2_t v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = (i + 256 * u < count) ? __builtin_nontemporal_load(x + i + 256 * u) : double2_t{0.0, 0.0};
#pragma unroll
        for (int u = 0; u < UNR; ++u) s += v[u].x + v[u].y;
    }
    if (s == 12345.678) out[0] = s;
}

struct Variant {
    std::string name;
    double bytes;
    std::function<void(hipStream_t)> launch;
    std::function<void(hipStream_t)> reduce;  // fills ychk (empty: no check)
    std::vector<float> ms;
};

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 16384;
    const int rounds = argc > 2 ? atoi(argv[2]) : 10;
    const long long ld = n + (argc > 3 ? atoll(argv[3]) : 16);
    if (n % 2048) { fprintf(stderr, "n must be a multiple of 2048\n"); return 1; }
    double *Q, *g, *yref, *ychk, *Qp64, *Qp128;
    DevState* st;
    CK(hipMalloc(&Q, (size_t)n * ld * 8));
    CK(hipMalloc(&g, n * 8));
    CK(hipMalloc(&yref, n * 8));
    CK(hipMalloc(&ychk, n * 8));
    CK(hipMalloc(&st, sizeof(DevState)));
    const long long nt64 = tile_prefix<64, 2048>(n / 64, n / 2048), nt128 = tile_prefix<128, 1024>(n / 128, n / 1024);
    CK(hipMalloc(&Qp64, (size_t)nt64 * 64 * 2048 * 8));
    CK(hipMalloc(&Qp128, (size_t)nt128 * 128 * 1024 * 8));
    {
        std::vector<double> h((size_t)n);
        for (long long i = 0; i < n; ++i) h[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
        CK(hipMemcpy(g, h.data(), n * 8, hipMemcpyHostToDevice));
        DevState s{};
        s.kappa = 1.0;
        CK(hipMemcpy(st, &s, sizeof s, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fill_sym, dim3(4096), dim3(256), 0, 0, Q, ld, n);
        hipLaunchKernelGGL((k_pack<64, 2048>), dim3((unsigned)(n / 64), (unsigned)(n / 2048)), dim3(256), 0, 0, Q, ld, n, Qp64);
        hipLaunchKernelGGL((k_pack<128, 1024>), dim3((unsigned)(n / 128), (unsigned)(n / 1024)), dim3(256), 0, 0, Q, ld, n, Qp128);
        CK(hipDeviceSynchronize());
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipLaunchKernelGGL((k_sweep<4, 4, 2, true, false, true, false>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, Q, Q, ld, n, n,
                       0LL, (const double*)nullptr, g, yref, st, 0);
    CK(hipStreamSynchronize(s));
    std::vector<double> href((size_t)n), hchk((size_t)n);
    CK(hipMemcpy(href.data(), yref, n * 8, hipMemcpyDeviceToHost));
    double *rowpart, *colpart;
    const size_t rp_bytes = (size_t)(n / 1024) * n * 8, cp_bytes = (size_t)(n / 32) * n * 8;
    CK(hipMalloc(&rowpart, rp_bytes));
    CK(hipMalloc(&colpart, cp_bytes));
    const double tri = 4.0 * (double)n * (double)n;

    std::vector<Variant> vs;
    vs.push_back({"prod k_symv<2> 64x2048 row-major (a)", tri,
                  [=](hipStream_t q) {
                      dim3 grid((unsigned)(n / SYMV_H), (unsigned)(n / SYMV_SEG));
                      hipLaunchKernelGGL((k_symv<2, true, 0, SYMV_SEG>), grid, dim3(256), 0, q, Q, ld, n, 0LL, n, g, rowpart, colpart, st);
                  },
                  [=](hipStream_t q) { hipLaunchKernelGGL((k_check_reduce<64, 2048>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, n, rowpart, colpart, ychk); },
                  {}});
#define XV(H, SEG, RW, RED8, PACKED, ROT, SRC, LABEL)                                                                                   \
    vs.push_back({LABEL, tri,                                                                                                          \
                  [=](hipStream_t q) {                                                                                                 \
                      hipLaunchKernelGGL((k_symvx<H, SEG, RW, RED8, PACKED, ROT>), dim3((unsigned)(n / H), (unsigned)(n / SEG)), dim3(256), 0, q, \
                                         (const double*)SRC, ld, n, g, rowpart, colpart);                                               \
                  },                                                                                                                   \
                  [=](hipStream_t q) { hipLaunchKernelGGL((k_check_reduce<H, SEG>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, n, rowpart, colpart, ychk); }, \
                  {}});
    XV(64, 2048, 2, false, false, false, Q, "x 64x2048 rw2 row-major")
#define XF(H, SEG, RW, LABEL)                                                                                                           \
    vs.push_back({LABEL, tri,                                                                                                          \
                  [=](hipStream_t q) {                                                                                                 \
                      hipLaunchKernelGGL((k_symvx<H, SEG, RW, false, false, false, true>), dim3((unsigned)(n / H), (unsigned)(n / SEG)), dim3(256), 0, q, \
                                         (const double*)Q, ld, n, g, rowpart, colpart);                                                \
                  },                                                                                                                   \
                  [=](hipStream_t q) { hipLaunchKernelGGL((k_check_reduce<H, SEG>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, n, rowpart, colpart, ychk); }, \
                  {}});
    XF(64, 2048, 2, "x 64x2048 rw2 row-major FMA")
    XF(64, 2048, 4, "x 64x2048 rw4 row-major FMA")
    XV(64, 2048, 4, false, false, false, Q, "x 64x2048 rw4 row-major")
    XF(128, 1024, 4, "x 128x1024 rw4 row-major FMA")
    XF(128, 2048, 2, "x 128x2048 rw2 row-major FMA")
    XF(32, 2048, 4, "x 32x2048 rw4 row-major FMA")
    vs.push_back({"ref: contiguous read of 4n^2 bytes x4 grid 4096", tri,
                  [=](hipStream_t q) { hipLaunchKernelGGL((k_stream_read<4>), dim3(4096), dim3(256), 0, q, (const double2_t*)Q, (size_t)(n * n / 4), ychk); }, {}, {}});
    vs.push_back({"ref: contiguous read of 4n^2 bytes x8 grid 1024", tri,
                  [=](hipStream_t q) { hipLaunchKernelGGL((k_stream_read<8>), dim3(1024), dim3(256), 0, q, (const double2_t*)Q, (size_t)(n * n / 4), ychk); }, {}, {}});
    vs.push_back({"ref: full-row GEMV k_sweep<4,4,nt> (8n^2 bytes)", 2 * tri,
                  [=](hipStream_t q) { hipLaunchKernelGGL((k_sweep<4, 4, 2, true, false, true, false>), dim3((unsigned)(n / 4)), dim3(256), 0, q, Q, Q, ld, n, n, 0LL, (const double*)nullptr, g, yref, st, 0); }, {}, {}});

    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("n=%lld ld=%lld rounds=%d   4n^2 = %.1f MB; tiles 64x2048: %lld, 128x1024: %lld\n", n, ld, rounds, tri / 1e6, nt64, nt128);
    for (auto& v : vs) {
        if (!v.reduce) continue;
        CK(hipMemsetAsync(rowpart, 0xff, rp_bytes, s));
        CK(hipMemsetAsync(colpart, 0xff, cp_bytes, s));
        v.launch(s);
        v.reduce(s);
        CK(hipStreamSynchronize(s));
        CK(hipGetLastError());
        CK(hipMemcpy(hchk.data(), ychk, n * 8, hipMemcpyDeviceToHost));
        double err = 0.0, sc = 0.0;
        for (long long i = 0; i < n; ++i) {
            const double d = std::fabs(hchk[i] - href[i]);
            err = (d > err || d != d) ? (d != d ? INFINITY : d) : err;
            sc = std::max(sc, std::fabs(href[i]));
        }
        printf("check %-46s max|y - y_gemv| / max|y| = %.3e %s\n", v.name.c_str(), err / sc, err / sc < 1e-12 ? "ok" : "MISMATCH");
    }
    for (int r = 0; r < rounds + 1; ++r)
        for (auto& v : vs) {
            // realistic cache state: a pass over other data precedes every timed launch (the apply pass / pending vectors in production)
            hipLaunchKernelGGL((k_stream_read<4>), dim3(4096), dim3(256), 0, s, (const double2_t*)colpart, cp_bytes / 16, ychk);
            CK(hipEventRecord(a, s));
            v.launch(s);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (r > 0) v.ms.push_back(ms);
        }
    CK(hipGetLastError());
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms.front(), mx = v.ms.back();
        printf("%-50s med %.4f ms  min %.4f  max %.4f  %7.1f GB/s (med) %7.1f GB/s (best)\n", v.name.c_str(), med, mn, mx, v.bytes / med / 1e6,
               v.bytes / mn / 1e6);
    }
    return 0;
}
