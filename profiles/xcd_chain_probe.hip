// xcd_chain_probe (round 6): would dec_chain gain from being CONFINED TO ONE (OR TWO) XCDs?
//
// dec_chain is 24 dependent layers per decode step; a layer's hand-off is an all-to-all among the 32 column slices of a row group
// (tag-free 4-byte values + sentinel, sc1 stores / 16-byte sc1 loads: handoff2_probe's mode B, what ships).  Spread over 8 XCDs (the
// chain partition is 8 CUs of every XCD: workgroup b goes to XCD b % 8 whatever the CU mask) the exchange crosses the fabric; inside
// one XCD the sc1 accesses meet in that XCD's L2.  A launch cannot be placed on one XCD by its CU mask (a mask that leaves an XCD
// without CUs is ignored, xcc_mask_probe.hip) -- but it can over-launch 8 x, read HW_REG_XCC_ID, let the workgroups that landed
// elsewhere exit and hand the others their slice by ticket.  This probe measures exactly that, in dec_chain's geometry (32 slices
// x 2 row groups x 8 rows, 8 waves per workgroup, 512 columns), with and without the layer's weight stream (each workgroup pulls
// its 16 columns x 768 K x 4 B = 48 KB of every layer: 24 distinct layers = 1.15 MB per workgroup and step, 28.7 MB per row group --
// far more than one XCD's 4 MB L2, which is the price of confinement):
//
//   P0  today's placement: 64 workgroups on a stream masked to 8 CUs of every XCD
//   P1  all 64 workgroups on XCD 0 (512 launched, 448 exit): 2 per CU on its 32 CUs
//   P2  row group 0 on XCD 0, row group 1 on XCD 1 (the all-to-all never crosses: row groups do not talk), 1 per CU
//
// build: hipcc --offload-arch=gfx950 -O3 profiles/xcd_chain_probe.hip -o /tmp/xcd_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int G = 32, RG = 2, R = 8, NCOLS = 512, NSLOT = 4, NLAYER = 24;
constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr int WCHUNK = 16 * 768;                 // floats of one workgroup's weight slice of one layer (48 KB)

__device__ __forceinline__ f32x4 ld16(const float* base, unsigned off) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16));
}
__device__ __forceinline__ void st16(float* base, unsigned off, f32x4 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)off, 0, 16);
}
__device__ __forceinline__ float wsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// place: 0 = (blockIdx.x, blockIdx.y) as launched; 1 = all on XCD 0 by ticket; 2 = row group rg on XCD rg by ticket
// tickets[0..1]: per row group (place 2) or [0] for all (place 1); xcd_seen[ticket] = XCC of the workgroup that took it
__global__ __launch_bounds__(512) void chain(float* buf, const float* W, long long* out, int* bad, int* tickets, int* xcd_seen, int nl, int place, int weights) {
    __shared__ float part[R];
    __shared__ int slot_s;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcc = xcc_id();
    int g, rg;
    if (place == 0) { g = blockIdx.x % G; rg = blockIdx.x / G; if (tid == 0) xcd_seen[rg * G + g] = xcc; }
    else {
        if (place == 1 && xcc != 0) return;
        if (place == 2 && xcc > 1) return;
        if (tid == 0) slot_s = atomicAdd(tickets + (place == 2 ? xcc : 0), 1);
        __syncthreads();
        const int tk = slot_s;
        if (place == 1) { if (tk >= G * RG) return; g = tk % G; rg = tk / G; }
        else { if (tk >= G) return; g = tk; rg = xcc; }
        if (tid == 0) xcd_seen[rg * G + g] = xcc;
    }
    const int grow = rg * R + w;
    const long long t0 = wall_clock64();
    long long passes = 0;
    int nbad = 0;
    float wacc = 0.f;
    f32x4 wv[6];
    auto wload = [&](int l) {                    // this workgroup's 48 KB of layer l: 6 x 16 bytes per lane (plain loads, as dec_chain's fragments)
        const float* src = W + ((size_t)(l % NLAYER) * G + g) * WCHUNK;
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = *(const f32x4*)(src + (size_t)(i * 512 + tid) * 4);
    };
    if (weights) wload(0);
    for (int l = 0; l < nl; ++l) {
        const int slot = l % NSLOT;
        float rowsum = 0.f;
        if (l > 0) {
            const int ps = (l - 1) % NSLOT;
            const float* row = buf + ((size_t)ps * 16 + grow) * NCOLS;
            for (int it = 0;; ++it) {
                const f32x4 a = ld16(row, lane * 16), b = ld16(row, 1024 + lane * 16);
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) ok = ok && __float_as_uint(a[e]) != SENT && __float_as_uint(b[e]) != SENT;
                ++passes;
                if (__all(ok)) { rowsum = (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]); break; }
                if (it > 2000000) { ++nbad; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            rowsum = wsum(rowsum);
            if (fabsf(rowsum - (float)(l - 1 + grow)) > 1e-2f * (float)(l + 16)) ++nbad;
        }
        if (weights) {                           // the layer's "contraction": consume this layer's fragments (they were requested a layer ago)
#pragma unroll
            for (int i = 0; i < 6; ++i) wacc += (wv[i][0] + wv[i][1]) + (wv[i][2] + wv[i][3]);
        }
        if (lane == 0) part[w] = rowsum;
        __syncthreads();
        if (tid < 32) {
            const int row = tid >> 2, c4 = (tid & 3) * 4;
            const float v = ((float)(l + rg * R + row) + 0.f * part[row]) * (1.0f / NCOLS);
            st16(buf, (unsigned)(((((size_t)slot * 16 + rg * R + row) * NCOLS) + g * 16 + c4) * 4), f32x4{v, v, v, v});
            const int rs = (l + 2) % NSLOT;
            const float s = __uint_as_float(SENT);
            st16(buf, (unsigned)(((((size_t)rs * 16 + rg * R + row) * NCOLS) + g * 16 + c4) * 4), f32x4{s, s, s, s});
        }
        if (weights && l + 1 < nl) wload(l + 1); // the next layer's fragments behind the publish (dec_chain's order)
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { out[(rg * G + g) * 2] = t1 - t0; out[(rg * G + g) * 2 + 1] = passes; }
    if (nbad) atomicAdd(bad, nbad);
    if (wacc == 12345.678f) out[0] = 0;
}

// a second tenant: `nwg` workgroups that stream memory for ~`us` microseconds each, launched again and again on another stream (the
// cone's shape: 168 workgroups of 512 threads with 64 KB of LDS) -- what happens to them, and to the chain, when 1/8 land on XCD 0?
__global__ __launch_bounds__(512) void tenant(const float* src, float* sink, long long ticks, int* per_xcc) {
    extern __shared__ float lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicAdd(per_xcc + xcc_id(), 1);
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    while (wall_clock64() - t0 < ticks) { acc += src[i & ((1u << 22) - 1)]; i += 512 * 168; lds[threadIdx.x] = acc; }
    if (acc == 1.2345f) sink[0] = acc;
}

static hipStream_t g_masked, g_plain, g_side;

static double run(const char* name, int place, int weights, bool with_tenant) {
    const int NL = NLAYER * 80;
    float* buf; float* W; long long* out; int* bad; int* tickets; int* seen; int* per_xcc; float* tsrc;
    const size_t bytes = (size_t)NSLOT * 16 * NCOLS * 4, wbytes = (size_t)NLAYER * G * WCHUNK * 4;
    hipMalloc(&buf, bytes); hipMalloc(&W, wbytes); hipMalloc(&out, G * RG * 2 * 8); hipMalloc(&bad, 4); hipMalloc(&tickets, 8); hipMalloc(&seen, G * RG * 4);
    hipMalloc(&per_xcc, 64); hipMalloc(&tsrc, (size_t)4 << 22);
    hipMemset(W, 0, wbytes); hipMemset(tsrc, 0, (size_t)4 << 22);
    std::vector<long long> h(G * RG * 2);
    std::vector<int> hs(G * RG);
    double best = 1e30, bp = 0; int hb = 0; int placed = 0; unsigned xmask = 0; int tenant_launches = 0; int txcc[16] = {0};
    for (int rep = 0; rep < 4; ++rep) {
        hipStream_t s = place == 0 ? g_masked : g_plain;
        hipMemsetAsync(buf, 0xFF, bytes, s); hipMemsetAsync(bad, 0, 4, s); hipMemsetAsync(tickets, 0, 8, s); hipMemsetAsync(seen, 0xFF, G * RG * 4, s);
        hipMemsetAsync(out, 0, G * RG * 2 * 8, s); hipMemset(per_xcc, 0, 64);
        hipStreamSynchronize(s);
        const int grid = place == 0 ? G * RG : (place == 1 ? 8 * G * RG : 8 * G);
        hipLaunchKernelGGL(chain, dim3(grid), dim3(512), 0, s, buf, W, out, bad, tickets, seen, NL, place, weights);
        if (with_tenant) {
            // keep the side stream busy for as long as the chain runs: ~15 us launches of 168 workgroups with 64 KB of LDS each
            hipFuncSetAttribute((const void*)tenant, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            for (int k = 0; k < 400 && hipStreamQuery(s) == hipErrorNotReady; ++k) { hipLaunchKernelGGL(tenant, dim3(168), dim3(512), 65536, g_side, tsrc, buf + 4096 * 64, 1500LL, per_xcc); ++tenant_launches; if ((k & 7) == 7) hipStreamSynchronize(g_side); }
            hipStreamSynchronize(g_side);
        }
        if (hipStreamSynchronize(s) != hipSuccess) { printf("%s: failed\n", name); return 0; }
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hs.data(), seen, hs.size() * 4, hipMemcpyDeviceToHost);
        int b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost); hb += b;
        int tx[16]; hipMemcpy(tx, per_xcc, 64, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) txcc[i] += tx[i];
        double mx = 0, ps = 0; placed = 0; xmask = 0;
        for (int i = 0; i < G * RG; ++i) { mx = mx > (double)h[2 * i] ? mx : (double)h[2 * i]; ps += (double)h[2 * i + 1]; if (hs[i] >= 0) { ++placed; xmask |= 1u << hs[i]; } }
        const double us = mx * 0.01 / NL;
        if (placed == G * RG && us < best) { best = us; bp = ps / (G * RG) / 8 / NL; }
    }
    printf("%-58s %s%s: %6.3f us per layer = %6.1f us per 24-layer step, %5.2f passes, placed %d/64 on XCD mask 0x%02x, bad=%d", name, weights ? "+weights" : "        ",
           with_tenant ? "+tenant" : "       ", best, best * NLAYER, bp, placed, xmask, hb);
    if (with_tenant) printf("  [tenant: %d launches, workgroups per XCC %d %d %d %d %d %d %d %d]", tenant_launches, txcc[0], txcc[1], txcc[2], txcc[3], txcc[4], txcc[5], txcc[6], txcc[7]);
    printf("\n");
    hipFree(buf); hipFree(W); hipFree(out); hipFree(bad); hipFree(tickets); hipFree(seen); hipFree(per_xcc); hipFree(tsrc);
    return best;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> m(words, 0), mc(words, 0);
    for (int i = 0; i < ncu; ++i) { if ((i / 8) % 4 == 0) m[i / 32] |= 1u << (i % 32); else if ((i / 8) % 4 != 3) mc[i / 32] |= 1u << (i % 32); }      // chain: 8 CUs of every XCC; side: the cone's 16
    hipExtStreamCreateWithCUMask(&g_masked, words, m.data());
    hipStreamCreateWithFlags(&g_plain, hipStreamNonBlocking);
    hipExtStreamCreateWithCUMask(&g_side, words, mc.data());
    printf("device: %s, %d CUs\n", p.name, ncu);
    for (int weights : {0, 1}) {
        run("P0  64 workgroups on 8 CUs of every XCD (today)", 0, weights, false);
        run("P1  64 workgroups on XCD 0 (2 per CU), 448 exit", 1, weights, false);
        run("P2  row group 0 on XCD 0, row group 1 on XCD 1, 192 exit", 2, weights, false);
    }
    // the other partitions' launches land on every XCD (workgroup b -> XCD b % 8): beside a confined chain
    run("P0  today, beside a cone-shaped tenant on its own CU mask", 0, 1, true);
    run("P1  XCD 0, beside the same tenant (masked to the cone's CUs)", 1, 1, true);
    run("P2  XCDs 0 + 1, beside the same tenant", 2, 1, true);
    return 0;
}
