// grid_barrier_bench.hip -- what does a grid-wide barrier of one workgroup per CU cost on MI355X (for the resident
// update kernel, csrc/resident_kernels.hpp)?  256 workgroups x 256 threads, K barriers in a row, several forms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE 0: stores drained (vmcnt 0), syncthreads, relaxed add, one lane polls with s_sleep 1, syncthreads
// MODE 1: the same without the payload stores (pure barrier)
// MODE 2: poll without s_sleep
// MODE 3: release add + acquire fence (the first version of k_ell_resident)
// MODE 4: four lanes poll, staggered
// MODE 5: per-workgroup flags instead of one counter: everybody writes flag[wg] = epoch, wave 0 polls 4 flags per lane
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, unsigned* flags, double* payload, int K, int* err) {
    __shared__ int ok;
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    for (int k = 1; k <= K; ++k) {
        if (MODE != 1) st_wt(payload + (size_t)wg * 256 + tid, (double)k);
        if (MODE == 5) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + wg, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 64) {
                bool done = false;
                for (int spin = 0; spin < (1 << 20) && !done; ++spin) {
                    bool all = true;
                    for (int j = tid; j < G; j += 64) all = all && (int)(__hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)k) >= 0;
                    done = __all(all);
                }
                if (tid == 0) ok = done;
            }
            __syncthreads();
            if (!ok) { if (tid == 0) *err = 1; return; }
            continue;
        }
        if (MODE == 6 || MODE == 7) {
            // tree: 16 group counters on their own 128-byte lines (flags + 32 * (1 + g)), root counter at flags[0];
            // MODE 6: everybody polls the root; MODE 7: the last arrival at the root releases 16 group flags (flags + 32 * (32 + g))
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const int ngrp = 16, g = wg % ngrp;
                const unsigned members = (unsigned)((G - g + ngrp - 1) / ngrp);
                const unsigned old = __hip_atomic_fetch_add(flags + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool released = false;
                if (old + 1 == members * (unsigned)k) {
                    const unsigned r = __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (MODE == 7 && r + 1 == (unsigned)ngrp * (unsigned)k) {
                        for (int j = 0; j < ngrp; ++j) __hip_atomic_store(flags + 32 * (32 + j), (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        released = true;
                    }
                }
                int good = released ? 1 : 0;
                const unsigned* pollp = MODE == 6 ? flags : flags + 32 * (32 + g);
                const unsigned want = MODE == 6 ? (unsigned)ngrp * (unsigned)k : (unsigned)k;
                for (int spin = 0; spin < (1 << 22) && !good; ++spin) {
                    if ((int)(__hip_atomic_load(pollp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) good = 1;
                    else __builtin_amdgcn_s_sleep(1);
                }
                ok = good;
            }
            __syncthreads();
            if (!ok) { if (tid == 0) *err = 1; return; }
            continue;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned target = (unsigned)G * (unsigned)k;
        if (MODE == 4) {
            if (tid < 64) {
                if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool done = false;
                for (int z = 0; z < (tid & 3); ++z) __builtin_amdgcn_s_sleep(4);
                for (int spin = 0; spin < (1 << 20) && !done; ++spin) {
                    bool mine = false;
                    if (tid < 4) mine = (int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
                    done = __any(mine);
                }
                if (tid == 0) ok = done;
            }
        } else if (tid == 0) {
            if (MODE == 3) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int good = 0;
            for (int spin = 0; spin < (1 << 22); ++spin) {
                if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { good = 1; break; }
                if (MODE != 2) __builtin_amdgcn_s_sleep(1);
            }
            if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ok = good;
        }
        __syncthreads();
        if (!ok) { if (tid == 0) *err = 1; return; }
    }
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    unsigned *ctr, *flags;
    double* payload;
    int* err;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&flags, 65536)); CK(hipMalloc(&payload, (size_t)G * 256 * 8)); CK(hipMalloc(&err, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char* names[8] = {"payload stores + drain + relaxed add + 1-lane poll (s_sleep 1)", "no payload (pure barrier)", "poll without s_sleep",
                            "RELEASE add + ACQUIRE fence", "4 staggered polling lanes", "per-workgroup flags, wave 0 polls them all",
                            "tree 16 x 16, everybody polls the root counter", "tree 16 x 16, root's last arrival releases 16 group flags"};
    for (int mode = 0; mode < 8; ++mode) {
        std::vector<float> ms;
        for (int r = 0; r < 4; ++r) {
            CK(hipMemset(ctr, 0, 4)); CK(hipMemset(flags, 0, 65536)); CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(a, 0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_bar<0>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                case 1: hipLaunchKernelGGL(k_bar<1>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                case 2: hipLaunchKernelGGL(k_bar<2>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                case 3: hipLaunchKernelGGL(k_bar<3>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                case 4: hipLaunchKernelGGL(k_bar<4>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                case 5: hipLaunchKernelGGL(k_bar<5>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                case 6: hipLaunchKernelGGL(k_bar<6>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
                default: hipLaunchKernelGGL(k_bar<7>, dim3(G), dim3(256), 0, 0, ctr, flags, payload, K, err); break;
            }
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float t;
            CK(hipEventElapsedTime(&t, a, b));
            if (r) ms.push_back(t);
        }
        int herr;
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        std::sort(ms.begin(), ms.end());
        printf("mode %d  %-66s %7.3f us per barrier%s\n", mode, names[mode], ms[ms.size() / 2] / K * 1e3, herr ? "  TIMED OUT" : "");
    }
    return 0;
}
