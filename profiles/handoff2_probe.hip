// handoff2_probe (round 5): the dec_chain hand-off in isolation, in dec_chain's own geometry -- 32 column slices x 2 row groups of
// 8 utterance rows, 8 waves per workgroup, wave w needs ALL 512 columns of row w of the previous layer, then 128 threads publish
// this workgroup's 8 x 16 slice.  A dependent chain of NL layers with (almost) no arithmetic: what does one all-to-all cost, by
// transport?
//   A  8-byte {epoch, value} granules, sc1 stores / sc1 loads (what dec_chain does): 8 x dwordx2 loads per lane and pass
//   B  tag-free: 4-byte values, a slot is reset to a sentinel (a NaN pattern arithmetic never produces) two layers before it is
//      rewritten; 2 x 16-byte sc1 loads per lane and pass, 32 lanes publish 16 bytes each (+ 32 reset stores)
//   C  as A, but the 8-byte granules leave as 16-byte sc1 stores (two granules per lane)
//   D  as B, the sweep by ONE wave per row pair ... (not built)
// build: hipcc --offload-arch=gfx950 -O3 profiles/handoff2_probe.hip -o /tmp/handoff2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int G = 32, RG = 2, R = 8, NCOLS = 512, NSLOT = 4;
constexpr unsigned SENT = 0xFFFFFFFFu;

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

// buf: A/C: u64 [NSLOT][16 rows][NCOLS]; B: float [NSLOT][16 rows][NCOLS]
template <int MODE>
__global__ __launch_bounds__(512) void chain(void* buf, long long* out, int* bad, int nl, int work) {
    __shared__ float part[R];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x, rg = blockIdx.y, grow = rg * R + w;
    const long long t0 = wall_clock64();
    long long passes = 0;
    int nbad = 0;
    float carry = 0.f;
    for (int l = 0; l < nl; ++l) {
        const int slot = l % NSLOT;
        // ---- gather the previous layer's row (layer 0 has no input)
        float rowsum = 0.f;
        if (l > 0) {
            const int ps = (l - 1) % NSLOT;
            const unsigned ep = (unsigned)l;
            if (MODE == 1) {
                const float* row = (const float*)buf + ((size_t)ps * 16 + grow) * NCOLS;
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
            } else {
                const u64* row = (const u64*)buf + ((size_t)ps * 16 + grow) * NCOLS;
                for (int it = 0;; ++it) {
                    u64 v[8];
                    bool ok = true;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = __hip_atomic_load(row + (e >> 2) * 256 + lane * 4 + (e & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(v[e] >> 32) == ep;
                    }
                    ++passes;
                    if (__all(ok)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) rowsum += __uint_as_float((unsigned)v[e]);
                        break;
                    }
                    if (it > 2000000) { ++nbad; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            rowsum = wsum(rowsum);
            // every value of layer l-1 is 1 / NCOLS * (l - 1 + row): the row sum is (l - 1 + row)
            if (fabsf(rowsum - (float)(l - 1 + grow)) > 1e-2f * (float)(l + 16)) ++nbad;
        }
        // ---- "work" that depends on the gathered row
        for (int i = 0; i < work; ++i) carry = __builtin_fmaf(carry, 0.999f, rowsum * 1e-9f);
        if (lane == 0) part[w] = rowsum;
        __syncthreads();
        // ---- publish this workgroup's 8 x 16 slice of layer l: value (l + row) / NCOLS (+ a dependence on the gathered data)
        if (MODE == 1) {
            if (tid < 32) {
                const int row = tid >> 2, c4 = (tid & 3) * 4;
                const float v = ((float)(l + rg * R + row) + 0.f * part[row]) * (1.0f / NCOLS);
                float* base = (float*)buf;
                st16(base, (unsigned)(((((size_t)slot * 16 + rg * R + row) * NCOLS) + g * 16 + c4) * 4), f32x4{v, v, v, v});
                const int rs = (l + 2) % NSLOT;
                const float s = __uint_as_float(SENT);
                st16(base, (unsigned)(((((size_t)rs * 16 + rg * R + row) * NCOLS) + g * 16 + c4) * 4), f32x4{s, s, s, s});
            }
        } else if (MODE == 0) {
            if (tid < 128) {
                const int row = tid >> 4, col = tid & 15;
                const float v = ((float)(l + rg * R + row) + 0.f * part[row]) * (1.0f / NCOLS);
                __hip_atomic_store((u64*)buf + ((size_t)slot * 16 + rg * R + row) * NCOLS + g * 16 + col, ((u64)(l + 1) << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (tid < 64) {
                const int row = tid >> 3, c2 = (tid & 7) * 2;
                const float v = ((float)(l + rg * R + row) + 0.f * part[row]) * (1.0f / NCOLS);
                const float e = __uint_as_float((unsigned)(l + 1));
                st16((float*)buf, (unsigned)(((((size_t)slot * 16 + rg * R + row) * NCOLS) + g * 16 + c2) * 8), f32x4{v, e, v, e});
            }
        }
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { out[(rg * G + g) * 2] = t1 - t0; out[(rg * G + g) * 2 + 1] = passes; }
    if (nbad) atomicAdd(bad, nbad);
    if (carry == 12345.f) out[0] = 0;
}

template <int MODE>
static void run(const char* name, hipStream_t s, int work) {
    const int NL = 2000;
    void* buf; long long* out; int* bad;
    const size_t bytes = (size_t)NSLOT * 16 * NCOLS * 8;
    hipMalloc(&buf, bytes); hipMalloc(&out, G * RG * 2 * 8); hipMalloc(&bad, 4);
    std::vector<long long> h(G * RG * 2);
    double best = 1e30, bp = 0; int hb = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemsetAsync(buf, MODE == 1 ? 0xFF : 0, bytes, s);
        hipMemsetAsync(bad, 0, 4, s);
        hipLaunchKernelGGL(chain<MODE>, dim3(G, RG), dim3(512), 0, s, buf, out, bad, NL, work);
        if (hipStreamSynchronize(s) != hipSuccess) { printf("%s: failed\n", name); return; }
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        int b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost); hb += b;
        double mx = 0, ps = 0;
        for (int i = 0; i < G * RG; ++i) { mx = mx > (double)h[2 * i] ? mx : (double)h[2 * i]; ps += (double)h[2 * i + 1]; }
        const double us = mx * 0.01 / NL;
        if (us < best) { best = us; bp = ps / (G * RG) / 8 / NL; }
    }
    printf("%-64s work %3d: %6.3f us per layer, %5.2f passes per wave-layer, bad=%d\n", name, work, best, bp, hb);
    hipFree(buf); hipFree(out); hipFree(bad);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> m(words, 0);
    for (int i = 0; i < ncu; ++i) if ((i / 8) % 4 == 0) m[i / 32] |= 1u << (i % 32);       // 8 CUs of every XCC: the chain partition's shape
    hipStream_t s; hipExtStreamCreateWithCUMask(&s, words, m.data());
    for (int work : {0, 200}) {
        run<0>("A: 8-byte {epoch,value} granules, sc1 (dec_chain today)", s, work);
        run<2>("C: the same granules published as 16-byte sc1 stores", s, work);
        run<1>("B: tag-free 4-byte values + sentinel reset, 16-byte sc1 loads/stores", s, work);
    }
    return 0;
}
