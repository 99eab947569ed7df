// Weight repacking ON THE DEVICE (oph_finalize_weights): every variable arrives as the TF tensor it is -- conv kernels (size, Cin, Cout),
// transposed-conv kernels (1, 3, Cout, Cin), vectors, lookup tables -- either uploaded from the host copies oph_set_weight collected or
// sitting in the flat buffer oph_set_weights_device was handed (the RCCL receive buffer of the start-up broadcast: consumed in place,
// no 210 MB device -> host -> device round trip per rank).  One thread per DESTINATION element; launches of a few microseconds, once.
#include "oph_internal.h"
#include "oph_device.h"

namespace oph {

// conv kernel (size, cin, cout) -> Wt[Nalloc][size * kc], k contiguous, tap order = kernel order, zero padding
__global__ void pack_conv_k(const float* k, float* Wt, int size, int cin, int cout, int kc, int Nalloc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n_el = (size_t)Nalloc * size * kc;
    if (i >= n_el) return;
    const int n = (int)(i / ((size_t)size * kc)), rem = (int)(i % ((size_t)size * kc)), t = rem / kc, c = rem % kc;
    Wt[i] = (n < cout && c < cin) ? k[((size_t)t * cin + c) * cout + n] : 0.f;
}
// transposed-conv kernel (1, 3, cout, cin): even phase We[Nalloc][2 kc] = [Kt[0,0] | Kt[0,2]], odd phase Wo[Nalloc][kc] = Kt[0,1]
// ([TF-sem] o[2t] = x[t].Kt[0,0]^T + x[t-1].Kt[0,2]^T ; o[2t+1] = x[t].Kt[0,1]^T, modules.py:242-250)
__global__ void pack_convT_k(const float* kt, float* We, float* Wo, int cin, int cout, int kc, int Nalloc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n_el = (size_t)Nalloc * 3 * kc;
    if (i >= n_el) return;
    const int n = (int)(i / ((size_t)3 * kc)), rem = (int)(i % ((size_t)3 * kc)), slot = rem / kc, c = rem % kc;      // slot 0, 1: even phase; 2: odd
    const bool in = n < cout && c < cin;
    if (slot < 2) We[(size_t)n * 2 * kc + slot * kc + c] = in ? kt[((size_t)(slot == 0 ? 0 : 2) * cout + n) * cin + c] : 0.f;
    else Wo[(size_t)n * kc + c] = in ? kt[((size_t)1 * cout + n) * cin + c] : 0.f;
}
// k = 1 conv kernel (1, cin, N) -> Wkn[kc][ldn], n contiguous (row_chain / cone_head)
__global__ void pack_wkn_k(const float* k, float* Wkn, int cin, int N, int kc, int ldn) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)kc * ldn) return;
    const int c = (int)(i / ldn), n = (int)(i % ldn);
    Wkn[i] = (c < cin && n < N) ? k[(size_t)c * N + n] : 0.f;
}
// the context rows of AudioDec C_1's kernel (1, 2d, d) -> Wc[ldvw][kc_c], k contiguous (cone head: V . Wc)
__global__ void pack_wtc_k(const float* k, float* Wc, int d, int kc_c, int ldvw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)ldvw * kc_c) return;
    const int n = (int)(i / kc_c), c = (int)(i % kc_c);
    Wc[i] = (n < d && c < d) ? k[(size_t)c * d + n] : 0.f;
}
// highway kernel (3, 256, 512) + bias (512) -> hc_fused's order: [column tile jt][K-step][64 columns][64 k], a tile's columns =
// [32 H1 channels | the same 32 channels of H2]; bias in the same column order
__global__ void pack_hcf_k(const float* kr, const float* bs, float* wp, float* bp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)512 * 768) return;
    const int kk = (int)(i & 63), q = (int)((i >> 6) & 63), ks = (int)((i >> 12) % 12), jt = (int)(i / ((size_t)12 * 4096));
    const int col = q < 32 ? 32 * jt + q : 256 + 32 * jt + (q - 32), k = ks * 64 + kk, tap = k >> 8, c = k & 255;
    wp[i] = kr[((size_t)tap * 256 + c) * 512 + col];
    if (i < 512) { const int jb = (int)(i >> 6), qb = (int)(i & 63); bp[i] = bs[qb < 32 ? 32 * jb + qb : 256 + 32 * jb + (qb - 32)]; }
}
// Wt[rows][ldw] (k contiguous; rows beyond `rows_have` read as zeros) -> the whole-decode launches' lane order (dec_loop / dec_chain):
// [column slice g][wave w][chunk slot pf][lane][4] with chunk = min(w + R pf, nch - 1), column = 16 g + 4 (lane >> 4) + (lane & 3),
// k = 16 chunk + 4 ((lane >> 2) & 3) + e -- one fragment request of a wave is 1 KB contiguous.  One thread per 16-byte piece.
__global__ void pack_loop_k(const float* Wt, int ldw, int rows_have, int nch, int slices, int R, int PF, float* dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)slices * R * PF * 64) return;
    const int lane = (int)(i & 63), pf = (int)((i >> 6) % PF), w = (int)((i >> 6) / PF % R), g = (int)((i >> 6) / ((size_t)PF * R));
    const int ch = min(w + R * pf, nch - 1), col = 16 * g + 4 * (lane >> 4) + (lane & 3), k = 16 * ch + 4 * ((lane >> 2) & 3);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < rows_have) v = *(const f32x4*)(Wt + (size_t)col * ldw + k);
    *(f32x4*)(dst + i * 4) = v;
}
// vectors and tables: dst[i] = i < n ? f(src[i]) : 0; mode 1: learned channel contributions -- sigmoid(x), row 0 of the table reads as
// zeros (modules.embed zero-pads it, modules.py:38-40)
__global__ void pad_copy_k(const float* src, float* dst, size_t n, size_t npad, int mode, int row0) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    float v = i < n ? src[i] : 0.f;
    if (mode == 1) v = 1.0f / (1.0f + expf(-(i < (size_t)row0 ? 0.0f : v)));
    dst[i] = v;
}
// *out = max(*out, bits of max |x|)  (non-negative floats order like their bit patterns; NaN patterns sort above every finite value)
__global__ void maxabs_k(const float* x, size_t n, unsigned* out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(x[i])));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
void launch_pack_conv(const float* k, float* Wt, int size, int cin, int cout, int kc, int Nalloc, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_k, dim3(blocks_for((size_t)Nalloc * size * kc)), dim3(256), 0, s, k, Wt, size, cin, cout, kc, Nalloc);
}
void launch_pack_convT(const float* kt, float* We, float* Wo, int cin, int cout, int kc, int Nalloc, hipStream_t s) {
    hipLaunchKernelGGL(pack_convT_k, dim3(blocks_for((size_t)Nalloc * 3 * kc)), dim3(256), 0, s, kt, We, Wo, cin, cout, kc, Nalloc);
}
void launch_pack_wkn(const float* k, float* Wkn, int cin, int N, int kc, int ldn, hipStream_t s) {
    hipLaunchKernelGGL(pack_wkn_k, dim3(blocks_for((size_t)kc * ldn)), dim3(256), 0, s, k, Wkn, cin, N, kc, ldn);
}
void launch_pack_wtc(const float* k, float* Wc, int d, int kc_c, int ldvw, hipStream_t s) {
    hipLaunchKernelGGL(pack_wtc_k, dim3(blocks_for((size_t)ldvw * kc_c)), dim3(256), 0, s, k, Wc, d, kc_c, ldvw);
}
void launch_pack_hcf(const float* kr, const float* bs, float* wp, float* bp, hipStream_t s) {
    hipLaunchKernelGGL(pack_hcf_k, dim3(blocks_for((size_t)512 * 768)), dim3(256), 0, s, kr, bs, wp, bp);
}
void launch_pad_copy(const float* src, float* dst, size_t n, size_t npad, int mode, int row0, hipStream_t s) {
    hipLaunchKernelGGL(pad_copy_k, dim3(blocks_for(npad)), dim3(256), 0, s, src, dst, n, npad, mode, row0);
}
void launch_maxabs(const float* x, size_t n, unsigned* out, hipStream_t s) {
    hipLaunchKernelGGL(maxabs_k, dim3(std::min<unsigned>(blocks_for(n), 1024u)), dim3(256), 0, s, x, n, out);
}

void launch_pack_loop(const float* Wt, int ldw, int rows_have, int nch, int slices, int R, int PF, float* dst, hipStream_t s) {
    const size_t n = (size_t)slices * R * PF * 64;
    hipLaunchKernelGGL(pack_loop_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wt, ldw, rows_have, nch, slices, R, PF, dst);
}

}  // namespace oph
