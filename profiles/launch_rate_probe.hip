#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
struct Big { int a[64]; };
__global__ void empty_k(Big b) { if (b.a[0] == 12345) __builtin_trap(); }
__global__ void spin_k(Big b, int iters) { for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(20); if (b.a[0] == 12345) __builtin_trap(); }
int main() {
    hipStream_t s;
    uint32_t m[8] = {0};
    for (int i = 0; i < 64; ++i) m[i / 32] |= 1u << (i % 32);
    hipExtStreamCreateWithCUMask(&s, 8, m);
    Big b{};
    for (int rep = 0; rep < 3; ++rep) {
        for (int mode = 0; mode < 2; ++mode) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 6400; ++i) {
                if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(32), dim3(1024), 0, s, b);
                else hipLaunchKernelGGL(spin_k, dim3(32), dim3(1024), 0, s, b, 40);
            }
            auto t1 = std::chrono::steady_clock::now();
            hipStreamSynchronize(s);
            auto t2 = std::chrono::steady_clock::now();
            printf("%s: enqueue %.2f ms (%.2f us/launch), drained %.2f ms\n", mode ? "spin " : "empty",
                   std::chrono::duration<double>(t1 - t0).count() * 1e3, std::chrono::duration<double>(t1 - t0).count() * 1e6 / 6400,
                   std::chrono::duration<double>(t2 - t0).count() * 1e3);
        }
    }
    return 0;
}
