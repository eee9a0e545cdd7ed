// mfma_f64_rate.hip -- round-4 probe: cycles per v_mfma_f64_16x16x4_f64 on gfx950 with K independent accumulator chains per
// wave and W waves per SIMD, no memory traffic (what the FP64 matrix pipe gives k_symm_mfma / k_apply_mfma at best).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

template <int K>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double a0, double b0, unsigned long long* clk) {
    double4_t acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 32 / K; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K>
static void run(int wgs_per_cu, int iters) {
    const int grid = 256 * wgs_per_cu;
    double* out;
    unsigned long long* clk;
    CK(hipMalloc(&out, (size_t)grid * 256 * 8));
    CK(hipMalloc(&clk, (size_t)grid * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<K>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37, 0.91, clk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[8];
    CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    const double mfmas = (double)iters * 32;
    printf("K=%d chains, %d wave(s) per SIMD: %.1f shader cycles per MFMA per wave (s_memtime), %.3f ms, %.1f TFLOP/s chip-wide\n", K,
           wgs_per_cu, (double)h[0] / mfmas, ms, mfmas * 2048.0 * grid * 4 / (ms * 1e-3) / 1e12);
    CK(hipFree(out));
    CK(hipFree(clk));
}

int main() {
    const int iters = 4000;
    for (int w = 1; w <= 2; ++w) {
        run<1>(w, iters);
        run<2>(w, iters);
        run<4>(w, iters);
        run<8>(w, iters);
    }
    return 0;
}
