// glds_probe: where does global_load_lds_dwordx4 put each lane's 16 bytes?  Expectation: LDS base (wave-uniform, M0) +
// 16 * lane.  build: hipcc --offload-arch=gfx950 -O2 profiles/glds_probe.hip -o /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* g, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 4 + 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 64 * 4 + 64; i += blockDim.x) lds[i] = -1.f;
    __syncthreads();
    // wave w loads global floats [w*256 + 4*perm(lane), +4) to LDS base = lds + w*256 (+ 16 B * lane implied)
    const int src = w * 256 + 4 * (lane ^ 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src),
                                     (__attribute__((address_space(3))) void*)(lds + w * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
    float *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o);
    float r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        const float want = (float)(w * 256 + 4 * (l ^ 3) + e);
        if (r[w * 256 + l * 4 + e] != want) { if (bad < 5) printf("w %d lane %d e %d: got %g want %g\n", w, l, e, r[w * 256 + l * 4 + e], want); ++bad; }
    }
    printf("glds probe: %d mismatches (0 = lane l's 16 bytes land at base + 16 l)\n", bad);
    return bad != 0;
}
