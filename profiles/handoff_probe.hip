// handoff_probe: what does the all-to-all hand-off of one decoder layer cost on MI355X?
// Pattern of dec_run: G workgroups x 16 waves; per "layer" every workgroup publishes a 16x16 slice as 8-byte
// {epoch, value} granules and every wave then needs one full row (ncols granules).  No compute in between.
// Variants:
//   placement  0 = G workgroups wherever the dispatcher puts them (one per XCD round-robin: cross-XCD hand-offs)
//              1 = 8*G workgroups launched, only those on XCC `target` take a ticket and take part (same-XCD)
//   store      0 = agent-scope relaxed atomic store (sc1, write-through)   1 = workgroup-scope store (stays in the XCD's L2)
//   gather     0 = every wave polls its own row   1 = wave 0 of the workgroup polls all 16 rows, LDS broadcast
// build: hipcc --offload-arch=gfx950 -O3 profiles/handoff_probe.hip -o /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
constexpr int NL = 64, G = 32, NCOLS = 512;

__device__ __forceinline__ u64 gload(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int STORE>
__device__ __forceinline__ void gstore(u64* p, u64 v) {
    if (STORE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int PLACE, int STORE, int GATHER>
__global__ __launch_bounds__(1024) void xchg(u64* buf, int* ticket, int target, long long* out, int* bad, int ncols) {
    __shared__ int slice_s;
    __shared__ float stage[16 * NCOLS];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int slice = blockIdx.x;
    if (PLACE == 1) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
        if (xcc != target) return;
        if (tid == 0) slice_s = atomicAdd(ticket, 1);
        __syncthreads();
        slice = slice_s;
        if (slice >= G) return;
    }
    const long long t0 = wall_clock64();
    long long passes = 0;
    int nbad = 0;
    for (int l = 0; l < NL; ++l) {
        u64* slot = buf + (size_t)l * 16 * NCOLS;
        const unsigned epoch = l + 1;
        // publish: 256 threads, (row, col) of this slice -- columns beyond ncols are not published
        if (tid < 256) {
            const int row = tid >> 4, col = slice * 16 + (tid & 15);
            if (col < ncols) gstore<STORE>(slot + row * NCOLS + col, ((u64)epoch << 32) | (unsigned)(l * 7 + row * 1000 + col));
        }
        // gather
        if (GATHER == 0) {
            const u64* row = slot + w * NCOLS;
            for (;;) {
                bool ok = true;
                u64 v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = (e >> 2) * 256 + lane * 4 + (e & 3);
                    v[e] = c < ncols ? gload(row + c) : ((u64)epoch << 32);
                    ok = ok && (unsigned)(v[e] >> 32) == epoch;
                }
                ++passes;
                if (__all(ok)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = (e >> 2) * 256 + lane * 4 + (e & 3);
                        if (c < ncols && (unsigned)v[e] != (unsigned)(l * 7 + w * 1000 + c)) ++nbad;
                    }
                    break;
                }
                if (passes > 4000000) { ++nbad; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
            if (w == 0) {
                for (int r = 0; r < 16; ++r) {
                    const u64* row = slot + r * NCOLS;
                    for (;;) {
                        bool ok = true;
                        u64 v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int c = (e >> 2) * 256 + lane * 4 + (e & 3);
                            v[e] = c < ncols ? gload(row + c) : ((u64)epoch << 32);
                            ok = ok && (unsigned)(v[e] >> 32) == epoch;
                        }
                        ++passes;
                        if (__all(ok)) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int c = (e >> 2) * 256 + lane * 4 + (e & 3);
                                if (c < ncols) stage[r * NCOLS + c] = __uint_as_float((unsigned)v[e]);
                            }
                            break;
                        }
                        if (passes > 4000000) { ++nbad; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            __syncthreads();
            for (int c = lane; c < ncols; c += 64)
                if (__float_as_uint(stage[w * NCOLS + c]) != (unsigned)(l * 7 + w * 1000 + c)) ++nbad;
            __syncthreads();
        }
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { out[slice * 2] = t1 - t0; out[slice * 2 + 1] = passes; }
    if (nbad) atomicAdd(bad, nbad);
}

template <int PLACE, int STORE, int GATHER>
static void run(const char* name, int ncols, u64* buf, int* ticket, long long* out, int* bad) {
    std::vector<long long> h(G * 2);
    double best = 1e30, bp = 0;
    int hb = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(buf, 0, (size_t)NL * 16 * NCOLS * 8);
        hipMemset(ticket, 0, 4);
        hipMemset(bad, 0, 4);
        hipMemset(out, 0, G * 2 * 8);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((xchg<PLACE, STORE, GATHER>), dim3(PLACE ? 8 * G : G), dim3(1024), 0, 0, buf, ticket, rep % 8, out, bad, ncols);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
        hipMemcpy(h.data(), out, G * 2 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        double mx = 0, ps = 0;
        int n = 0;
        for (int i = 0; i < G; ++i) if (h[i * 2] > 0) { mx = mx > (double)h[i * 2] ? mx : (double)h[i * 2]; ps += (double)h[i * 2 + 1]; ++n; }
        if (n < G) { printf("%s: only %d of %d slices took part (xcc %d)\n", name, n, G, rep % 8); continue; }
        const double us = mx * 0.01 / NL;
        if (us < best) { best = us; bp = ps / n / NL; }
    }
    printf("%-58s ncols %3d: %6.2f us per layer, %5.1f polls per wave-layer, bad=%d\n", name, ncols, best, bp, hb);
}

int main() {
    u64* buf; int *ticket, *bad; long long* out;
    hipMalloc(&buf, (size_t)NL * 16 * NCOLS * 8);
    hipMalloc(&ticket, 4); hipMalloc(&bad, 4); hipMalloc(&out, G * 2 * 8);
    for (int ncols : {512, 256}) {
        run<0, 0, 0>("cross-XCD, sc1 stores, every wave polls its row", ncols, buf, ticket, out, bad);
        run<0, 0, 1>("cross-XCD, sc1 stores, wave 0 gathers 16 rows", ncols, buf, ticket, out, bad);
        run<1, 0, 0>("same-XCD,  sc1 stores, every wave polls its row", ncols, buf, ticket, out, bad);
        run<1, 1, 0>("same-XCD,  L2-resident stores, every wave polls its row", ncols, buf, ticket, out, bad);
        run<1, 1, 1>("same-XCD,  L2-resident stores, wave 0 gathers 16 rows", ncols, buf, ticket, out, bad);
    }
    return 0;
}
