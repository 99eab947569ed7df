// Host-side cost of enqueuing work on this box (MI355X, ROCm 7.2): how many microseconds of HOST time one launch /
// one event operation takes when the device is not the limit (kernels spin ~22 us so that the queue never drains).
// Used for DESIGN.md section 4 (decode loop: host enqueue vs device drain).   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
template <int N> struct Args { int a[N]; };
template <int N> __global__ void spin_k(Args<N> b, int iters) {
    extern __shared__ float sm[];
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(20);
    if (b.a[0] == 12345) sm[0] = 1.f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s, s2;
    uint32_t m[8] = {0}, m2[8] = {0};
    for (int i = 0; i < 64; ++i) m[i / 32] |= 1u << (i % 32);
    for (int i = 64; i < 192; ++i) m2[i / 32] |= 1u << (i % 32);
    (void)hipExtStreamCreateWithCUMask(&s, 8, m);
    (void)hipExtStreamCreateWithCUMask(&s2, 8, m2);
    hipEvent_t e1, e2;
    (void)hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    (void)hipFuncSetAttribute((const void*)spin_k<100>, hipFuncAttributeMaxDynamicSharedMemorySize, 66000);
    const int N = 4000;
    uint32_t* sig = nullptr;
    if (hipMalloc((void**)&sig, 64) != hipSuccess || hipMemset(sig, 0, 64) != hipSuccess) return 1;
    if (hipStreamWriteValue32(s, sig, 0, 0) != hipSuccess) { printf("stream value ops not usable\n"); return 1; }
    for (int mode = 0; mode < 8; ++mode) {
        (void)hipDeviceSynchronize();
        const double t0 = now();
        for (int i = 0; i < N; ++i) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(spin_k<16>, dim3(32), dim3(1024), 0, s, Args<16>{}, 30); break;          // 64-byte args
                case 1: hipLaunchKernelGGL(spin_k<100>, dim3(32), dim3(1024), 0, s, Args<100>{}, 30); break;        // 400-byte args
                case 2: hipLaunchKernelGGL(spin_k<100>, dim3(32), dim3(1024), 65000, s, Args<100>{}, 30); break;    // + 65 KB dynamic LDS
                case 3: hipLaunchKernelGGL(spin_k<100>, dim3(32), dim3(1024), 65000, (i / 8) % 2 ? s2 : s, Args<100>{}, 30); break;  // two streams, runs of 8
                case 4: (void)hipEventRecord(e1, s); (void)hipStreamWaitEvent(s2, e1, 0); break;                    // 2 event ops
                case 5: hipLaunchKernelGGL(spin_k<100>, dim3(32), dim3(1024), 65000, s, Args<100>{}, 30);
                        if (i % 16 == 0) { (void)hipEventRecord(e1, s); (void)hipStreamWaitEvent(s2, e1, 0); (void)hipEventRecord(e2, s2); (void)hipStreamWaitEvent(s, e2, 0); }
                        break;
                case 6: (void)hipStreamWriteValue32(s, sig, (uint32_t)(i + 1), 0); (void)hipStreamWaitValue32(s2, sig, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu); break;
                case 7: hipLaunchKernelGGL(spin_k<100>, dim3(32), dim3(1024), 65000, s, Args<100>{}, 30);
                        if (i % 16 == 0) {
                            (void)hipStreamWriteValue32(s, sig, (uint32_t)(2 * N + 2 * i + 1), 0); (void)hipStreamWaitValue32(s2, sig, (uint32_t)(2 * N + 2 * i + 1), hipStreamWaitValueGte, 0xffffffffu);
                            (void)hipStreamWriteValue32(s2, sig + 8, (uint32_t)(i + 1), 0); (void)hipStreamWaitValue32(s, sig + 8, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu);
                        }
                        break;
            }
        }
        const double t1 = now();
        (void)hipDeviceSynchronize();
        const double t2 = now();
        static const char* names[] = {"64B args", "400B args", "400B args + 65KB dyn LDS", "same, two masked streams in runs of 8",
                                      "eventRecord + streamWaitEvent pair", "launch + 4 event ops every 16 launches",
                                      "streamWriteValue32 + streamWaitValue32 pair", "launch + 4 stream-value ops every 16 launches"};
        printf("%-44s host %.2f us per iteration (enqueue %.1f ms, drained %.1f ms)\n", names[mode], (t1 - t0) * 1e6 / N,
               (t1 - t0) * 1e3, (t2 - t0) * 1e3);
    }
    return 0;
}
