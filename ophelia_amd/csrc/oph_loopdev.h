// Device-side pieces shared by the decode kernels that keep their workgroups resident over many layers (oph_decrun.hip: dec_run /
// dec_loop, oph_decchain.hip: dec_chain): the 8-byte {epoch, value} granule hand-off and the packed layer descriptor.  gfx950 only.
#pragma once
#include "oph_internal.h"
#include "oph_device.h"

namespace oph {

typedef unsigned long long u64;
constexpr int RUN_KMAX = 768;                     // largest contraction length of a layer (3 taps x 256)
constexpr long long RUN_TIMEOUT_TICKS = 200000000LL;   // 2 s of the 100 MHz constant clock: a hand-off that takes
                                                       // longer means a workgroup of the run never became resident

static __device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave re-reads its row's granules until every tag carries this layer's epoch.  Lane l owns columns c..c+3 (and
// c2..c2+3 of the second half when `two`).  Bounded: on a time-out (or when another wave already failed) the error word
// is set and the run continues with whatever was read, so the launch always terminates.
static __device__ __forceinline__ int sweep_row(const u64* row, int c, bool cok, int c2, bool two, unsigned epoch, int lane,
                                                int* err, f32x4& av, f32x4& uv, bool once = false) {
    long long t0 = 0;
    const u64 want = (u64)epoch << 32;
    for (int it = 0;; ++it) {
        u64 ga[4], gu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ga[e] = cok ? granule_load(row + c + e) : want;
#pragma unroll
        for (int e = 0; e < 4; ++e) gu[e] = (cok && two) ? granule_load(row + c2 + e) : want;
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 4; ++e) ok = ok && (unsigned)(ga[e] >> 32) == epoch && (unsigned)(gu[e] >> 32) == epoch;
        bool give_up = false;
        if (!__all(ok) && it >= 64 && (it & 63) == 0) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            give_up = now - t0 > RUN_TIMEOUT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (give_up && lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__all(ok) || give_up || once) {      // once: timing ablation only
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] = __uint_as_float((unsigned)ga[e]);
                uv[e] = __uint_as_float((unsigned)gu[e]);
            }
            return it + 1;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Packed layer descriptor (oph_internal.h: LOOP_DESC_WORDS) held in scalar registers.  desc_load only ISSUES the scalar
// loads; desc_pin is the one place their results are waited for (an empty asm that needs every word in an SGPR) -- put
// after a wait that is long anyway (the hand-off sweep), so that no field access later stalls on the scalar cache.
typedef const __attribute__((address_space(4))) unsigned* LoopDescPtr;
struct LoopDesc {
    unsigned w[LOOP_DESC_WORDS];
    // global address space spelled out: a pointer assembled from two words is otherwise generic (flat_load, which also
    // ties up the LDS counter)
    __device__ __forceinline__ const float* ptr(int i) const {
        return (const float*)(const __attribute__((address_space(1))) float*)(((u64)w[i + 1] << 32) | (u64)w[i]);
    }
    __device__ __forceinline__ const float* Wt() const { return ptr(0); }
    __device__ __forceinline__ const float* bias() const { return ptr(2); }
    __device__ __forceinline__ const float* lnp() const { return ptr(4); }
    __device__ __forceinline__ const float* cat_table() const { return ptr(6); }
    __device__ __forceinline__ float* hist() const { return (float*)ptr(8); }
    __device__ __forceinline__ const float* cone(int odd) const { return odd ? ptr(12) : ptr(10); }
    __device__ __forceinline__ int pre() const { return w[14] & 15; }
    __device__ __forceinline__ int act() const { return (w[14] >> 4) & 15; }
    __device__ __forceinline__ int nonorm() const { return (w[14] >> 8) & 1; }
    __device__ __forceinline__ int ntaps() const { return (w[14] >> 12) & 3; }
    __device__ __forceinline__ int tapkind() const { return (w[14] >> 16) & 3; }
    __device__ __forceinline__ int next_pre() const { return (w[14] >> 20) & 15; }
    __device__ __forceinline__ int next_level() const { return (w[14] >> 28) & 15; }      // 1 + cone level read by the next layer's taps
    __device__ __forceinline__ int cin() const { return w[15] & 0xffff; }
    __device__ __forceinline__ int kc() const { return w[15] >> 16; }
    __device__ __forceinline__ int N() const { return w[16] & 0xffff; }
    __device__ __forceinline__ int ldw() const { return w[16] >> 16; }
    __device__ __forceinline__ int ccat() const { return w[17] & 0xffff; }
    __device__ __forceinline__ int ls() const { return w[17] >> 16; }
    __device__ __forceinline__ int off0() const { return w[18] & 0xffff; }
    __device__ __forceinline__ int off1() const { return w[18] >> 16; }
    __device__ __forceinline__ int idx0() const { return w[19] & 0xffff; }
    __device__ __forceinline__ int idx1() const { return w[19] >> 16; }
};
static __device__ __forceinline__ void desc_load(LoopDescPtr base, int l, LoopDesc& d) {
    LoopDescPtr p = base + l * LOOP_DESC_STRIDE;
#pragma unroll
    for (int i = 0; i < LOOP_DESC_WORDS; ++i) d.w[i] = p[i];
}
static __device__ __forceinline__ void desc_pin(LoopDesc& d) {
#pragma unroll
    for (int i = 0; i < LOOP_DESC_WORDS; ++i) asm volatile("" : "+s"(d.w[i]));
}

}  // namespace oph
