// Lane layout of v_mfma_f32_4x4x1_16B_f32 on gfx950 (16 independent 4x4 outer products per instruction):
// prints, for A[lane] = lane-coded row values and B[lane] = lane-coded column values, which (block, i, j) every
// output register holds.  build: hipcc --offload-arch=gfx950 -O2 profiles/mfma4x4_probe.hip -o /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float a = 1.0f + lane;            // A value of this lane
    const float b = 100.0f * (1 + lane);    // B value of this lane
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = acc[e];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // D = a_src * b_src with a_src = 1 + laneA, b_src = 100 (1 + laneB): decode (laneA, laneB) per output
    int ok = 1;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const float v = h[lane * 4 + e];
            int la = -1, lb = -1;
            for (int x = 0; x < 64 && la < 0; ++x) for (int y = 0; y < 64; ++y) if (v == (1.0f + x) * 100.0f * (1 + y)) { la = x; lb = y; break; }
            // hypothesis: block = lane/4; D reg e = row i = e -> A lane 4*block + e ; column j = lane%4 -> B lane = lane
            const int want_a = (lane / 4) * 4 + e, want_b = lane;
            if (la != want_a || lb != want_b) { ok = 0; if (lane < 8) printf("lane %d reg %d: A lane %d B lane %d (expected %d, %d)\n", lane, e, la, lb, want_a, want_b); }
        }
    printf("mfma_f32_4x4x1_16B layout: D[lane][e] = A[4*(lane/4)+e] * B[lane]  -> %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    return 0;
}
