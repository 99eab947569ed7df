// pingpong_probe: round trip between a long-running (persistent) kernel on stream A and a chain of short operations
// on stream B, the pattern of dec_loop + its side stream:
//   A: for i in 1..N: write ping = i (system-scope atomic), spin until pong >= i
//   B: for i in 1..N: [wait ping >= i] ; [set pong = i]        -- as stream value ops or as one-wave kernels
// Prints the average round trip per iteration.  Variants: plain streams or CU-masked streams (as oph_create makes them),
// host enqueue far ahead (everything queued before A starts) or throttled (host follows a pinned progress word).
// build: hipcc --offload-arch=gfx950 -O2 profiles/pingpong_probe.hip -o /tmp/pingpong_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

// grid > 1: every wave of every workgroup spins on pong like the loop kernel's waves do; block 0 lane 0 drives
__global__ void ping_kernel(unsigned* ping, const unsigned* pong, int n, long long* out, volatile int* host_prog) {
    const bool driver = blockIdx.x == 0 && threadIdx.x == 0;
    if (!driver && (threadIdx.x & 63) != 0) return;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        if (driver) {
            __hip_atomic_store(ping, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (host_prog) __hip_atomic_store((int*)host_prog, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        long long spins = 0;
        while ((int)(__hip_atomic_load(pong, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)i) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 40000000) { if (driver) { out[1] = i; out[0] = -1; } return; }
        }
    }
    if (driver) { out[0] = wall_clock64() - t0; out[1] = n; }
}
__global__ void wait_kernel(const unsigned* sig, unsigned want) {
    long long spins = 0;
    while ((int)(__hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > 40000000) return;
    }
}
__global__ void set_kernel(unsigned* sig, unsigned v) { __hip_atomic_fetch_max(sig, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void filler_kernel(float* p) { p[threadIdx.x] += 1.0f; }

static int g_grid = 1, g_block = 64;
static void run(const char* name, hipStream_t sa, hipStream_t sb, int mode, bool throttle, int fillers) {
    const int N = 200;
    unsigned* sig; long long* out; float* fill;
    hipMalloc(&sig, 256); hipMalloc(&out, 16); hipMalloc(&fill, 4096);
    hipMemset(sig, 0, 256); hipMemset(out, 0, 16);
    volatile int* hp = nullptr; void* dp = nullptr;
    hipHostMalloc((void**)&hp, 64, hipHostMallocMapped);
    hp[0] = 0;
    hipHostGetDevicePointer(&dp, (void*)hp, 0);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(ping_kernel, dim3(g_grid), dim3(g_block), 0, sa, sig, sig + 16, N, out, (volatile int*)dp);
    for (int i = 1; i <= N; ++i) {
        if (throttle) while (hp[0] < i - 8) {}
        if (mode == 0) hipStreamWaitValue32(sb, sig, i, hipStreamWaitValueGte, 0xffffffffu);
        else hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, sb, sig, (unsigned)i);
        for (int f = 0; f < fillers; ++f) hipLaunchKernelGGL(filler_kernel, dim3(64), dim3(256), 0, sb, fill);
        if (mode == 0) hipStreamWriteValue32(sb, sig + 16, i, 0);
        else hipLaunchKernelGGL(set_kernel, dim3(1), dim3(64), 0, sb, sig + 16, (unsigned)i);
    }
    const double enq = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
    hipStreamSynchronize(sa); hipStreamSynchronize(sb);
    long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    if (h[0] < 0) printf("%-64s TIMED OUT at iteration %lld\n", name, h[1]);
    else printf("%-64s %7.2f us per round trip (host enqueue %.1f us per iteration)\n", name, h[0] * 0.01 / N, enq / N);
    hipFree(sig); hipFree(out); hipFree(fill); hipHostFree((void*)hp);
}

int main() {
    hipStream_t a, b, ma, mb;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    uint32_t m1[8] = {0}, m2[8] = {0};
    for (int i = 0; i < 256; ++i) (i < 64 ? m1 : m2)[i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(&ma, 8, m1) != hipSuccess || hipExtStreamCreateWithCUMask(&mb, 8, m2) != hipSuccess) { printf("masked streams failed\n"); return 1; }
    for (int big = 0; big < 2; ++big)
    for (int throttle = 0; throttle < 2; ++throttle) {
        g_grid = big ? 128 : 1; g_block = big ? 256 : 64;
        printf("--- ping kernel: %d workgroups x %d threads, every wave polling\n", g_grid, g_block);
        const char* th = throttle ? "host throttled to 8 ahead" : "host enqueues everything ahead";
        char name[128];
        snprintf(name, sizeof name, "plain streams, stream value ops, %s", th); run(name, a, b, 0, throttle, 0);
        snprintf(name, sizeof name, "plain streams, wait/set kernels, %s", th); run(name, a, b, 1, throttle, 0);
        snprintf(name, sizeof name, "CU-masked streams, stream value ops, %s", th); run(name, ma, mb, 0, throttle, 0);
        snprintf(name, sizeof name, "CU-masked streams, wait/set kernels, %s", th); run(name, ma, mb, 1, throttle, 0);
        snprintf(name, sizeof name, "CU-masked, wait/set kernels + 13 fillers, %s", th); run(name, ma, mb, 1, throttle, 13);
    }
    return 0;
}
