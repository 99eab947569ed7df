// anyorder_probe: can two kernels of ONE stream overlap on gfx950 / ROCm 7.2 when the second is launched with
// hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch)?  (hip_ext.h says the flag "is not supported on AMD GFX9xx boards";
// this measures what actually happens.)  If they overlap, a chain of dependent launches can be software-pipelined: the successor
// is resident, has run its prologue and spins on a device word while the predecessor still computes -- no launch gap.
//
//   test 1: k1 spins SPIN_US and stamps [start, end]; k2 (ordinary launch, same stream) stamps its start     -> start2 - end1  = the boundary
//   test 2: the same with k2 launched any-order                                                                 -> start2 - start1 (overlap if << SPIN_US)
//   test 3: a chain of NCH any-order launches, each waiting in-kernel for its predecessor's word and then "working" WORK_US:
//           total time per link against the same chain as ordinary launches (no in-kernel wait needed there)
//   every test on a plain stream and on a CU-masked stream (hipExtStreamCreateWithCUMask, 16 CUs of every XCD)
// build: hipcc --offload-arch=gfx950 -O3 profiles/anyorder_probe.hip -o /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_k(long long ticks, long long* stamp) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}

// link i of a chain: (optionally) wait until word >= i, work, raise word to i + 1
__global__ void link_k(unsigned* word, int i, int wait, long long work_ticks, long long* stamp) {
    const long long t_in = wall_clock64();
    __shared__ int ok;
    if (wait) {
        if (threadIdx.x == 0) {
            long long t0 = wall_clock64();
            while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)i) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 2000000LL) break;        // 20 ms: the predecessor never ran -> would be a deadlock
            }
            ok = 1;
        }
        __syncthreads();
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(word + 16 + i, 1u) + 1u;
        if (done == gridDim.x) __hip_atomic_store(word, (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0) { stamp[2 * i] = t_in; stamp[2 * i + 1] = wall_clock64(); }
    }
}

static hipStream_t masked_stream() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> m(words, 0);
    for (int i = 0; i < ncu; ++i) if ((i / 8) % 2 == 0) m[i / 32] |= 1u << (i % 32);       // CU i sits on XCC i % 8: 16 CUs of every XCC
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, words, m.data()));
    return s;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    long long* st; unsigned* word;
    CK(hipMalloc(&st, 4096 * 8)); CK(hipMalloc(&word, 4096 * 4));
    std::vector<long long> h(4096);
    const long long SPIN = 20000;      // 200 us of the 100 MHz clock
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t s;
        if (masked) s = masked_stream(); else CK(hipStreamCreate(&s));
        for (int any = 0; any < 2; ++any) {
            double best = 1e30, best0 = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(st, 0, 64, s));
                CK(hipStreamSynchronize(s));
                hipLaunchKernelGGL(spin_k, dim3(8), dim3(64), 0, s, SPIN, st);
                hipExtLaunchKernelGGL(spin_k, dim3(8), dim3(64), 0, s, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, (long long)100, st + 2);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h.data(), st, 64, hipMemcpyDeviceToHost));
                const double d = (h[2] - h[1]) * 0.01, d0 = (h[2] - h[0]) * 0.01;
                if (d < best) { best = d; best0 = d0; }
            }
            printf("%s stream, second launch %-9s: start2 - end1 = %8.2f us, start2 - start1 = %8.2f us  (%s)\n", masked ? "CU-masked" : "plain    ",
                   any ? "any-order" : "ordinary", best, best0, best < 0 ? "OVERLAP" : "serialised");
        }
        // test 3: chains
        const int NCH = 48;
        for (int wg : {8, 128}) for (double work_us : {2.0, 8.0}) {
            for (int mode = 0; mode < 2; ++mode) {      // 0: ordinary launches, no in-kernel wait   1: any-order launches + in-kernel wait
                double best = 1e30;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipMemsetAsync(word, 0, 4096 * 4, s));
                    CK(hipMemsetAsync(st, 0, 4096 * 8, s));
                    CK(hipStreamSynchronize(s));
                    for (int i = 0; i < NCH; ++i) {
                        if (mode == 0) hipLaunchKernelGGL(link_k, dim3(wg), dim3(256), 0, s, word, i, 0, (long long)(work_us * 100), st);
                        else hipExtLaunchKernelGGL(link_k, dim3(wg), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, word, i, 1, (long long)(work_us * 100), st);
                    }
                    CK(hipStreamSynchronize(s));
                    CK(hipMemcpy(h.data(), st, NCH * 16, hipMemcpyDeviceToHost));
                    const double per = (h[2 * (NCH - 1) + 1] - h[2 * 8 + 1]) * 0.01 / (NCH - 1 - 8);      // steady state: links 8 .. NCH-1
                    if (per < best) best = per;
                }
                printf("%s stream, chain of %d links x %3d workgroups, %.0f us of work each, %-32s: %6.2f us per link (overhead %5.2f)\n",
                       masked ? "CU-masked" : "plain    ", NCH, wg, work_us, mode ? "any-order + in-kernel wait" : "ordinary launches", best, best - work_us);
            }
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}
