#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) {
        int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
        unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = (int)hwid;
    }
    __builtin_amdgcn_s_sleep(100);
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
    int* d; hipMalloc(&d, 4096 * 2 * 4);
    std::vector<int> h(4096 * 2);
    auto run = [&](const char* name, uint32_t* mask) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; }
        hipLaunchKernelGGL(probe, dim3(512), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, 512 * 2 * 4, hipMemcpyDeviceToHost);
        int hist[16] = {0};
        for (int i = 0; i < 512; ++i) hist[h[2 * i] & 15]++;
        printf("%-14s xcc hist:", name);
        for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
        printf("   distinct CUs per xcc:");
        for (int x = 0; x < 8; ++x) {
            std::vector<int> ids;
            for (int i = 0; i < 512; ++i) if ((h[2 * i] & 15) == x) {
                int id = (h[2 * i + 1] >> 8) & 0xff;     // CU_ID[11:8] SH_ID[12] SE_ID[15:13]
                bool seen = false; for (int v : ids) seen = seen || v == id;
                if (!seen) ids.push_back(id);
            }
            printf(" %zu", ids.size());
        }
        printf("\n");
        // do not destroy masked streams (re-creation after destroy hung before)
    };
    uint32_t m[8];
    for (int k = 0; k < 5; ++k) {
        for (auto& w : m) w = 0;
        const char* name = k == 0 ? "i%8==0" : k == 1 ? "i<32" : k == 2 ? "i%8==3" : k == 3 ? "i<64" : "all";
        for (int i = 0; i < 256; ++i) {
            bool on = k == 0 ? (i % 8 == 0) : k == 1 ? (i < 32) : k == 2 ? (i % 8 == 3) : k == 3 ? (i < 64) : true;
            if (on) m[i / 32] |= 1u << (i % 32);
        }
        run(name, m);
    }
    return 0;
}
