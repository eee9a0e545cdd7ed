// handoff_latency.hip -- ping-pong of one 8-byte value between two workgroups of one launch:
//   (a) on the same XCD, stores and polls with sc0 only (L1 bypassed, the XCD's L2 is the meeting point),
//   (b) on the same XCD, agent scope (sc1: what the persistent solves use), (c) on different XCDs, agent scope;
// each without and with the rest of the chip streaming a 2 GiB buffer (read-modify-write) beside it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void st_sc0(unsigned long long* p, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long ld_sc0(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid: blocks 0 and `partner` play; everyone else streams `buf` when asked.  Blocks are dealt round-robin over the
// 8 XCDs: partner = 8 -> same XCD as block 0, partner = 1 -> the next XCD.  Both report their XCC_ID.
template <bool SC0>
__global__ __launch_bounds__(256) void k_pingpong(unsigned long long* box, int partner, int rounds, double* buf,
                                                  size_t nbuf, unsigned long long* out, int* stopflag) {
    const int b = blockIdx.x;
    if (b == 0 || b == partner) {
        if (threadIdx.x != 0) return;
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* mine = box + (b == 0 ? 0 : 16);     // separate 128-byte lines
        unsigned long long* theirs = box + (b == 0 ? 16 : 0);
        const unsigned long long t0 = wall_clock64();
        int done = 0;
        for (int r = 1; r <= rounds; ++r) {
            long spin = 0;
            if (b == 0) {
                if (SC0) st_sc0(mine, (unsigned long long)r); else st_sc1(mine, (unsigned long long)r);
                while ((SC0 ? ld_sc0(theirs) : ld_sc1(theirs)) < (unsigned long long)r && ++spin < (1 << 15)) {}
            } else {
                while ((SC0 ? ld_sc0(theirs) : ld_sc1(theirs)) < (unsigned long long)r && ++spin < (1 << 15)) {}
                if (SC0) st_sc0(mine, (unsigned long long)r); else st_sc1(mine, (unsigned long long)r);
            }
            if (spin >= (1 << 15)) break;   // the value never arrived: give up (reported as rounds done < rounds)
            done = r;
        }
        const unsigned long long t1 = wall_clock64();
        out[b == 0 ? 4 : 5] = (unsigned long long)done;
        out[b == 0 ? 0 : 2] = t1 - t0;
        out[b == 0 ? 1 : 3] = xcc & 0xf;
        if (b == 0) __hip_atomic_store(stopflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (!buf) return;
    // background: stream until the players are done (bounded)
    const size_t per = nbuf / gridDim.x;
    double2* p = reinterpret_cast<double2*>(buf + per * b);
    for (int pass = 0; pass < 16; ++pass) {
        for (size_t i = threadIdx.x; i < per / 2; i += 256 * 4) {
            double2 a = p[i], c = (i + 256 < per / 2) ? p[i + 256] : a, d = (i + 512 < per / 2) ? p[i + 512] : a,
                    e = (i + 768 < per / 2) ? p[i + 768] : a;
            a.x += 1.0; c.x += 1.0; d.x += 1.0; e.x += 1.0;
            p[i] = a;
            if (i + 256 < per / 2) p[i + 256] = c;
            if (i + 512 < per / 2) p[i + 512] = d;
            if (i + 768 < per / 2) p[i + 768] = e;
        }
        if (__hip_atomic_load(stopflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}

int main() {
    unsigned long long *box, *out; int* stopflag; double* buf;
    const size_t nbuf = (size_t)256 << 20;  // 2 GiB of doubles
    CK(hipMalloc(&box, 4096)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&stopflag, 4)); CK(hipMalloc(&buf, nbuf * 8));
    CK(hipMemset(buf, 0, nbuf * 8));
    const int rounds = 500;
    for (int load = 0; load < 2; ++load)
        for (int variant = 0; variant < 3; ++variant) {
            const int partner = (variant == 2) ? 1 : 8;
            const bool sc0 = variant == 0;
            CK(hipMemset(box, 0, 4096)); CK(hipMemset(stopflag, 0, 4)); CK(hipMemset(out, 0, 64));
            if (sc0) hipLaunchKernelGGL(k_pingpong<true>, dim3(256), dim3(256), 0, 0, box, partner, rounds, load ? buf : nullptr, nbuf, out, stopflag);
            else hipLaunchKernelGGL(k_pingpong<false>, dim3(256), dim3(256), 0, 0, box, partner, rounds, load ? buf : nullptr, nbuf, out, stopflag);
            CK(hipDeviceSynchronize());
            unsigned long long h[8]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
            printf("%-46s %s: %.3f us per one-way hand-off (XCC %llu <-> XCC %llu; %llu / %llu of %d rounds completed)\n",
                   variant == 0 ? "same XCD, sc0 stores / polls (through its L2)" : variant == 1 ? "same XCD, agent scope (sc1)" : "different XCDs, agent scope (sc1)",
                   load ? "chip streaming 2 GiB RMW beside it" : "idle chip", h[0] / 100.0 / (h[4] ? h[4] : 1) / 2.0, h[1], h[3], h[4], h[5], rounds);
            fflush(stdout);
        }
    return 0;
}
