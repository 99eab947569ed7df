// aql_probe (round 5): do two kernel-dispatch packets of ONE user-mode AQL queue overlap on gfx950 when the second has no barrier
// bit?  Uses the library's own queue code (ophelia_amd/csrc/oph_aql.hip).  Variants: barrier bit on / off; acquire-release fence
// scope agent / none; grid 8 / 128 workgroups.
// build: see profiles/r05_aql_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../ophelia_amd/csrc/oph_aql.h"
using namespace oph;
struct SpinArgs { long long ticks; long long* stamp; };
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* co = argc > 1 ? argv[1] : "/tmp/aql_probe_kernels.co";
    hipSetDevice(0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> m(words, 0);
    for (int i = 0; i < ncu; ++i) if ((i / 8) % 2 == 0) m[i / 32] |= 1u << (i % 32);
    long long* st; hipMalloc(&st, 4096);
    SpinArgs* dargs; hipMalloc(&dargs, 4096);
    for (int masked = 0; masked < 2; ++masked) {
        std::string err;
        AqlQueue* q = aql_create(0, masked ? m.data() : nullptr, masked ? words : 0, co, 256, 2, &err);
        if (!q) { printf("aql_create failed: %s\n", err.c_str()); return 1; }
        AqlKernel k;
        if (!aql_kernel(q, "probe_spin", &k, &err)) { printf("%s\n", err.c_str()); return 1; }
        printf("%s queue: kernel object %llx kernarg %u group %u private %u\n", masked ? "CU-masked" : "plain", (unsigned long long)k.object, k.kernarg_size, k.group_static, k.private_size);
        for (int lanes2 = 0; lanes2 < 2; ++lanes2) for (int wgs : {8, 128}) for (int barrier = 1; barrier >= 0; --barrier) {
            double best = 1e30, best0 = 0;
            for (int rep = 0; rep < 5; ++rep) {
                SpinArgs ha[2] = {{20000, st}, {100, st + 2}};
                hipMemset(st, 0, 64);
                hipMemcpy(dargs, &ha[0], sizeof(SpinArgs), hipMemcpyHostToDevice);
                hipMemcpy((char*)dargs + 256, &ha[1], sizeof(SpinArgs), hipMemcpyHostToDevice);
                hipDeviceSynchronize();
                aql_dispatch(q, 0, k, wgs, 64, 0, dargs, true);
                aql_dispatch(q, lanes2 ? 1 : 0, k, wgs, 64, 0, (char*)dargs + 256, barrier != 0);
                aql_ring(q);
                if (!aql_wait_idle(q, 5.0)) { printf("wait_idle failed: %s\n", aql_error(q)); return 1; }
                long long h[4]; hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
                const double d = (h[2] - h[1]) * 0.01, d0 = (h[2] - h[0]) * 0.01;
                if (d < best) { best = d; best0 = d0; }
            }
            printf("  %s, %3d workgroups, second packet %s: start2 - end1 = %8.2f us, start2 - start1 = %8.2f us  (%s)\n", lanes2 ? "second packet on ANOTHER lane" : "both packets on one lane      ", wgs, barrier ? "WITH barrier bit" : "no barrier bit  ",
                   best, best0, best < 0 ? "OVERLAP" : "serialised");
        }
        aql_destroy(q);
    }
    // which queues run side by side?  8 queues (two groups of 4 lanes), a 200 us spin on queue 0, a short kernel on queue j: the
    // hypothesis is that queues share the compute pipes round-robin (4 pipes) and a pipe runs one queue at a time
    {
        std::string err;
        AqlQueue* g[2] = {aql_create(0, nullptr, 0, co, 256, 4, &err), aql_create(0, nullptr, 0, co, 256, 4, &err)};
        if (!g[0] || !g[1]) { printf("aql_create failed: %s\n", err.c_str()); return 1; }
        AqlKernel k[2];
        aql_kernel(g[0], "probe_spin", &k[0], &err); aql_kernel(g[1], "probe_spin", &k[1], &err);
        for (int i = 0; i < 8; ++i) {
            printf("  spin on queue %d; short kernel on queue j starts (us after the spin's start): ", i);
            for (int j = 0; j < 8; ++j) {
                if (j == i) { printf("   -   "); continue; }
                double best = 1e30;
                for (int rep = 0; rep < 3; ++rep) {
                    SpinArgs ha[2] = {{20000, st}, {100, st + 2}};
                    hipMemset(st, 0, 64);
                    hipMemcpy(dargs, &ha[0], sizeof(SpinArgs), hipMemcpyHostToDevice);
                    hipMemcpy((char*)dargs + 256, &ha[1], sizeof(SpinArgs), hipMemcpyHostToDevice);
                    hipDeviceSynchronize();
                    aql_dispatch(g[i / 4], i % 4, k[i / 4], 8, 64, 0, dargs, true);
                    aql_ring(g[i / 4]);
                    aql_dispatch(g[j / 4], j % 4, k[j / 4], 8, 64, 0, (char*)dargs + 256, true);
                    aql_ring(g[j / 4]);
                    aql_wait_idle(g[0], 5.0); aql_wait_idle(g[1], 5.0);
                    long long h[4]; hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
                    const double d0 = (h[2] - h[0]) * 0.01;
                    if (d0 < best) best = d0;
                }
                printf("%6.1f ", best);
            }
            printf("\n");
        }
        aql_destroy(g[0]); aql_destroy(g[1]);
    }
    return 0;
}
