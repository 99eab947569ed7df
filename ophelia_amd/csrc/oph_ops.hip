// libophelia_hip.so -- per-operator entry points of the C ABI (unit parity: the same kernels as the model path on host buffers) and
// the device-resident timing of modules.conv1d_transpose behind bench.py's kernel_rooflines.
#include "oph_host.h"

extern "C" {

// ---- per-operator entry points (unit parity) ---------------------------------------------------
const char* oph_op_last_error(void) { return g_op_error.c_str(); }

}  // extern "C"

namespace {
struct OpCtx {
    hipStream_t s = nullptr;
    std::vector<void*> bufs;
    bool ok = true;
    explicit OpCtx(int device) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || hipSetDevice(device) != hipSuccess ||
            hipStreamCreate(&s) != hipSuccess) { ok = false; g_op_error = "no usable HIP device (no CPU fallback)"; }
    }
    ~OpCtx() { for (void* p : bufs) hipFree(p); if (s) hipStreamDestroy(s); }
    template <class T> T* alloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { ok = false; g_op_error = "hipMalloc failed"; return nullptr; }
        hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(T), s);
        bufs.push_back(p);
        return (T*)p;
    }
    template <class T> T* up(const T* src, size_t n) {
        T* p = alloc<T>(n);
        if (p) hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
        return p;
    }
    float* up_pad(const float* src, size_t n, size_t padto) {
        std::vector<float> t((n + padto - 1) / padto * padto, 0.f);
        std::copy(src, src + n, t.begin());
        float* p = up(t.data(), t.size());
        hipStreamSynchronize(s);
        return p;
    }
    int finish() {
        hipError_t e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { g_op_error = hipGetErrorString(e); return OPH_ERR_DEVICE; }
        return ok ? OPH_OK : OPH_ERR_DEVICE;
    }
};
}  // namespace

extern "C" {

int oph_op_embed(int device, const int32_t* ids, int64_t n, const float* table, int vocab, int units, float* out) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    for (int64_t i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= vocab) { g_op_error = "id out of range"; return OPH_ERR_INVALID; }
    const int ldo = round_up(units, 4);
    int* dids = c.up(ids, (size_t)n);
    float* dt = c.up(table, (size_t)vocab * units);
    float* dout = c.alloc<float>((size_t)n * ldo);
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_embed(dids, n, dt, units, dout, ldo, c.s);
    hipMemcpy2DAsync(out, (size_t)units * 4, dout, (size_t)ldo * 4, (size_t)units * 4, (size_t)n, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

int oph_op_layernorm(int device, const float* x, int64_t rows, int C, const float* gamma, const float* beta, float* y) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (C < 1 || C > 1280) { g_op_error = "C out of range (<=1280)"; return OPH_ERR_UNSUPPORTED; }
    const int ld = round_up(C, 128);
    float* dx = c.up(x, (size_t)rows * C);
    float* dh = c.alloc<float>((size_t)rows * ld);
    float* dy = c.alloc<float>((size_t)rows * C);
    float* g = c.up_pad(gamma, C, 256); float* b = c.up_pad(beta, C, 256);
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_pad_rows(dx, C, dh, ld, rows, C, c.s);
    EpiArgs e{};
    e.H = dh; e.ldh = ld; e.M = (int)rows; e.C = C; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = g; e.b1 = b; e.Y = dy; e.ldy = C; e.ypad = C;
    launch_epilogue(e, c.s);
    hipMemcpyAsync(y, dy, (size_t)rows * C * 4, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

static int op_conv_common(int device, const float* x, int B, int T, int Cin, int Cout, int size, int rate, int padding,
                          const float* kernel, const float* bias, const float* g1, const float* b1, const float* g2,
                          const float* b2, int act, bool is_hc, float* y) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (size != 1 && size != 3) { g_op_error = "size must be 1 or 3"; return OPH_ERR_UNSUPPORTED; }
    if (Cout > 1280 || (is_hc && (Cout > 1024 || Cout % 4 || Cin != Cout))) { g_op_error = "channels out of range"; return OPH_ERR_UNSUPPORTED; }
    const int kc = round_up(Cin, 32), N = is_hc ? 2 * Cout : Cout, Nalloc = round_up(N, 128), M = B * T;
    std::vector<float> wt = pack_conv(kernel, size, Cin, N, kc, Nalloc);
    float* dx = c.up(x, (size_t)M * Cin);
    float* dxp = c.alloc<float>((size_t)M * kc);
    float* dw = c.up(wt.data(), wt.size());
    float* dbias = c.up_pad(bias, N, Nalloc);
    float* dh = c.alloc<float>((size_t)M * Nalloc);
    float* dy = c.alloc<float>((size_t)M * Cout);
    float* dg1 = c.up_pad(g1, Cout, 256); float* db1 = c.up_pad(b1, Cout, 256);
    float* dg2 = is_hc ? c.up_pad(g2, Cout, 256) : nullptr; float* db2 = is_hc ? c.up_pad(b2, Cout, 256) : nullptr;
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_pad_rows(dx, Cin, dxp, kc, M, Cin, c.s);
    GemmArgs g{};
    g.X = dxp; g.ldx = kc; g.Wt = dw; g.ldw = size * kc; g.bias = dbias; g.H = dh; g.ldh = Nalloc; g.M = M; g.N = N; g.kc = kc;
    g.ntaps = size; g.mode = 0; g.T = T;
    for (int t = 0; t < size; ++t) g.off[t] = padding == 1 ? -(size - 1 - t) * rate : (t - (size - 1) / 2) * rate;
    launch_conv_gemm(g, c.s);
    EpiArgs e{};
    e.H = dh; e.ldh = Nalloc; e.M = M; e.C = Cout; e.mode = is_hc ? PRE_HC : PRE_CONV; e.act = act;
    e.g1 = dg1; e.b1 = db1; e.g2 = dg2; e.b2 = db2; e.Xres = dxp; e.ldres = kc; e.Y = dy; e.ldy = Cout; e.ypad = Cout;
    launch_epilogue(e, c.s);
    hipMemcpyAsync(y, dy, (size_t)M * Cout * 4, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

int oph_op_conv1d(int device, const float* x, int B, int T, int Cin, int Cout, int size, int rate, int padding,
                  const float* kernel, const float* bias, const float* gamma, const float* beta, int act, float* y) {
    return op_conv_common(device, x, B, T, Cin, Cout, size, rate, padding, kernel, bias, gamma, beta, nullptr, nullptr, act, false, y);
}
int oph_op_hc(int device, const float* x, int B, int T, int C, int size, int rate, int padding, const float* kernel,
              const float* bias, const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, float* y) {
    return op_conv_common(device, x, B, T, C, C, size, rate, padding, kernel, bias, gamma1, beta1, gamma2, beta2, ACT_NONE, true, y);
}


int oph_op_conv1d_transpose(int device, const float* x, int B, int T, int Cin, int Cout, const float* kernel,
                            const float* bias, const float* gamma, const float* beta, float* y) {
    return oph_op_conv1d_transpose_prec(device, x, B, T, Cin, Cout, kernel, bias, gamma, beta, 0, y);
}
int oph_op_conv1d_transpose_prec(int device, const float* x, int B, int T, int Cin, int Cout, const float* kernel,
                                 const float* bias, const float* gamma, const float* beta, int precision, float* y) {
    if (precision < 0 || precision > 2) { g_op_error = "precision must be 0 (fp32 MFMA), 1 (split-bf16 x3) or 2 (split-fp16 x3)"; return OPH_ERR_INVALID; }
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (Cout > 1280) { g_op_error = "channels out of range"; return OPH_ERR_UNSUPPORTED; }
    const int kc = round_up(Cin, 32), Nalloc = round_up(Cout, 128), M = B * T;
    std::vector<float> we((size_t)Nalloc * 2 * kc, 0.f), wo((size_t)Nalloc * kc, 0.f);
    for (int n = 0; n < Cout; ++n)
        for (int ci = 0; ci < Cin; ++ci) {
            we[(size_t)n * 2 * kc + ci] = kernel[((size_t)0 * Cout + n) * Cin + ci];
            we[(size_t)n * 2 * kc + kc + ci] = kernel[((size_t)2 * Cout + n) * Cin + ci];
            wo[(size_t)n * kc + ci] = kernel[((size_t)1 * Cout + n) * Cin + ci];
        }
    float* dx = c.up(x, (size_t)M * Cin);
    float* dxp = c.alloc<float>((size_t)M * kc);
    float* dwe = c.up(we.data(), we.size()); float* dwo = c.up(wo.data(), wo.size());
    float* dbias = c.up_pad(bias, Cout, Nalloc);
    float* dh = c.alloc<float>((size_t)2 * M * Nalloc);
    float* dy = c.alloc<float>((size_t)2 * M * Cout);
    float* dg = c.up_pad(gamma, Cout, 256); float* db = c.up_pad(beta, Cout, 256);
    if (!c.ok) return OPH_ERR_DEVICE;
    launch_pad_rows(dx, Cin, dxp, kc, M, Cin, c.s);
    GemmArgs g{};
    g.X = dxp; g.ldx = kc; g.bias = dbias; g.ldh = 2 * Nalloc; g.M = M; g.N = Cout; g.kc = kc; g.mode = 0; g.T = T;
    g.Wt = dwe; g.ldw = 2 * kc; g.ntaps = 2; g.off[0] = 0; g.off[1] = -1; g.H = dh;
    GemmArgs g2 = g;
    g2.Wt = dwo; g2.ldw = kc; g2.ntaps = 1; g2.off[0] = 0; g2.H = dh + Nalloc;
    if (precision == 0) {
        launch_conv_gemm(g, c.s);
        launch_conv_gemm(g2, c.s);
    } else {        // the SSRN path's launch for this layer: both phases in one, on the split 16-bit planes
        unsigned short* dweh = c.alloc<unsigned short>(we.size()); unsigned short* dwel = c.alloc<unsigned short>(we.size());
        unsigned short* dwoh = c.alloc<unsigned short>(wo.size()); unsigned short* dwol = c.alloc<unsigned short>(wo.size());
        if (!c.ok) return OPH_ERR_DEVICE;
        if (precision == 2) { launch_split_f16(dwe, dweh, dwel, we.size(), c.s); launch_split_f16(dwo, dwoh, dwol, wo.size(), c.s); }
        else { launch_split_bf16(dwe, dweh, dwel, we.size(), c.s); launch_split_bf16(dwo, dwoh, dwol, wo.size(), c.s); }
        g.Wh = dweh; g.Wl = dwel; g.f16 = precision == 2; g.nprod = 3;
        g2.Wh = dwoh; g2.Wl = dwol; g2.f16 = g.f16; g2.nprod = 3;
        if (precision == 2) {       // split-fp16: the input as hi / lo planes (what the previous layer's LayerNorm launch writes in SSRN), both phases as one problem
            unsigned short* dxh = c.alloc<unsigned short>((size_t)M * kc); unsigned short* dxl = c.alloc<unsigned short>((size_t)M * kc);
            if (!c.ok) return OPH_ERR_DEVICE;
            launch_rows_to_planes(dxp, kc, M, kc, dxh, dxl, c.s);
            unsigned short* kweh = c.alloc<unsigned short>(we.size()); unsigned short* kwel = c.alloc<unsigned short>(we.size());
            unsigned short* kwoh = c.alloc<unsigned short>(wo.size()); unsigned short* kwol = c.alloc<unsigned short>(wo.size());
            if (!c.ok) return OPH_ERR_DEVICE;
            launch_kblock_planes(dweh, kweh, Nalloc, 2 * kc, c.s); launch_kblock_planes(dwel, kwel, Nalloc, 2 * kc, c.s);
            launch_kblock_planes(dwoh, kwoh, Nalloc, kc, c.s); launch_kblock_planes(dwol, kwol, Nalloc, kc, c.s);
            PlaneGemmArgs pg{};
            pg.Ah = (const _Float16*)dxh; pg.Al = (const _Float16*)dxl; pg.Wh = (const _Float16*)kweh; pg.Wl = (const _Float16*)kwel;
            pg.Wh2 = (const _Float16*)kwoh; pg.Wl2 = (const _Float16*)kwol; pg.bias = dbias; pg.H = dh; pg.M = M; pg.N = Cout; pg.kc = kc; pg.T = T;
            pg.nalloc = Nalloc; pg.ldh = 2 * Nalloc; pg.ntaps = 2; pg.off[0] = 0; pg.off[1] = -1; pg.convt = 1;
            if (Cout % 64 == 0 && Cout <= 1024) {       // the SSRN path's launch since round 6: LayerNorm inside (the statistics cross the column tiles)
                float* dst = (float*)c.alloc<unsigned char>(plane_gemm_ln_stats_bytes(M, Cout));
                int* derr = c.alloc<int>(1);
                if (!c.ok) return OPH_ERR_DEVICE;
                hipMemsetAsync(dst, 0, plane_gemm_ln_stats_bytes(M, Cout), c.s); hipMemsetAsync(derr, 0, 4, c.s);
                pg.ln_gamma = dg; pg.ln_beta = db; pg.Y = dy; pg.ldy = Cout; pg.ln_stats = dst; pg.ln_epoch = 1; pg.ln_err = derr;
                launch_plane_gemm(pg, c.s);
                int herr = 0;
                hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, c.s);
                hipMemcpyAsync(y, dy, (size_t)2 * M * Cout * 4, hipMemcpyDeviceToHost, c.s);
                const int rc = c.finish();
                if (rc == OPH_OK && herr) { g_op_error = "conv1d_transpose + LayerNorm: statistics exchange timed out"; return OPH_ERR_DEVICE; }
                return rc;
            }
            launch_plane_gemm(pg, c.s);
        } else
            launch_conv_gemm_pair(g, g2, precision, c.s);
    }
    EpiArgs e{};
    e.H = dh; e.ldh = Nalloc; e.M = 2 * M; e.C = Cout; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = dg; e.b1 = db; e.Y = dy; e.ldy = Cout; e.ypad = Cout;
    launch_epilogue(e, c.s);
    hipMemcpyAsync(y, dy, (size_t)2 * M * Cout * 4, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

// Device-resident timing of modules.conv1d_transpose (SSRN D_4 / D_7, networks.py:483-486) for the roofline report:
// the same launches as oph_op_conv1d_transpose / the SSRN path (even-phase GEMM, odd-phase GEMM, LayerNorm rows), on
// seeded random device data, `iters` repetitions bracketed by HIP events after `warmup` untimed ones.
int oph_bench_conv1d_transpose(int device, int B, int T, int Cin, int Cout, int precision, int warmup, int iters,
                               double* avg_us, double* alg_bytes, double* alg_flops) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (Cout > 1280 || B < 1 || T < 1 || iters < 1 || !avg_us) { g_op_error = "bad argument"; return OPH_ERR_INVALID; }
    const int kc = round_up(Cin, 32), Nalloc = round_up(Cout, 128), M = B * T;
    std::vector<float> we((size_t)Nalloc * 2 * kc, 0.f), wo((size_t)Nalloc * kc, 0.f), xh((size_t)M * kc, 0.f), bh((size_t)Nalloc, 0.f);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    const float ws = sqrtf(2.6f / (3.0f * Cin));
    for (int n = 0; n < Cout; ++n) {
        bh[n] = 0.02f * rnd();
        for (int ci = 0; ci < Cin; ++ci) { we[(size_t)n * 2 * kc + ci] = ws * rnd(); we[(size_t)n * 2 * kc + kc + ci] = ws * rnd(); wo[(size_t)n * kc + ci] = ws * rnd(); }
    }
    for (int m = 0; m < M; ++m) for (int ci = 0; ci < Cin; ++ci) xh[(size_t)m * kc + ci] = rnd();
    std::vector<float> gh((size_t)round_up(Cout, 256), 1.f), zh((size_t)round_up(Cout, 256), 0.f);
    float* dx = c.up(xh.data(), xh.size());
    float* dwe = c.up(we.data(), we.size()); float* dwo = c.up(wo.data(), wo.size());
    float* dbias = c.up(bh.data(), bh.size());
    float* dh = c.alloc<float>((size_t)2 * M * Nalloc);
    float* dy = c.alloc<float>((size_t)2 * M * Cout);
    float* dg = c.up(gh.data(), gh.size()); float* db = c.up(zh.data(), zh.size());
    // the weights' hi / lo bf16 planes, split once as at load time
    unsigned short* dweh = c.alloc<unsigned short>(we.size()); unsigned short* dwel = c.alloc<unsigned short>(we.size());
    unsigned short* dwoh = c.alloc<unsigned short>(wo.size()); unsigned short* dwol = c.alloc<unsigned short>(wo.size());
    if (!c.ok) return OPH_ERR_DEVICE;
    if (precision >= 2) { launch_split_f16(dwe, dweh, dwel, we.size(), c.s); launch_split_f16(dwo, dwoh, dwol, wo.size(), c.s); }
    else { launch_split_bf16(dwe, dweh, dwel, we.size(), c.s); launch_split_bf16(dwo, dwoh, dwol, wo.size(), c.s); }
    // precision 2 (the SSRN path's default): the input arrives as fp16 hi / lo planes (written by the previous layer's LayerNorm launch),
    // and this layer's LayerNorm launch writes planes for the next layer beside its fp32 rows; precision 5: the round-3 launches
    // (fp32 rows split inside the paired contraction)
    int pg_dbg = 0, ln_dbg = 0;
    if (precision == 11 || precision == 12) {          // measurement builds: the fused launch without its wait (11) / without its plane stores (12)
#ifdef OPH_ABLATE
        ln_dbg = precision == 11 ? 16 : 32; precision = 2;
#else
        g_op_error = "precisions 11, 12 exist only in measurement builds of the library (ABLATE)"; return OPH_ERR_UNSUPPORTED;
#endif
    }
    if (precision >= 6 && precision <= 9) {
#ifdef OPH_ABLATE
        pg_dbg = 1 << (precision - 6); precision = 2;
#else
        g_op_error = "precisions 6..9 (ablation builds of plane_gemm) exist only in measurement builds of the library (ABLATE)"; return OPH_ERR_UNSUPPORTED;
#endif
    }      // measurement only: plane_gemm without its MFMAs / without its operand stream
    const bool planes = precision == 2 || precision == 10;      // (10: the round-5 form -- the same operands, plane_gemm + ln_rows)
    unsigned short *dxh = nullptr, *dxl = nullptr, *dyh = nullptr, *dyl = nullptr, *kweh = nullptr, *kwel = nullptr, *kwoh = nullptr, *kwol = nullptr;
    if (planes) {
        dxh = c.alloc<unsigned short>((size_t)M * kc); dxl = c.alloc<unsigned short>((size_t)M * kc);
        dyh = c.alloc<unsigned short>((size_t)2 * M * round_up(Cout, 32)); dyl = c.alloc<unsigned short>((size_t)2 * M * round_up(Cout, 32));
        if (!c.ok) return OPH_ERR_DEVICE;
        launch_rows_to_planes(dx, kc, M, kc, dxh, dxl, c.s);
        kweh = c.alloc<unsigned short>(we.size()); kwel = c.alloc<unsigned short>(we.size());
        kwoh = c.alloc<unsigned short>(wo.size()); kwol = c.alloc<unsigned short>(wo.size());
        if (!c.ok) return OPH_ERR_DEVICE;
        launch_kblock_planes(dweh, kweh, Nalloc, 2 * kc, c.s); launch_kblock_planes(dwel, kwel, Nalloc, 2 * kc, c.s);
        launch_kblock_planes(dwoh, kwoh, Nalloc, kc, c.s); launch_kblock_planes(dwol, kwol, Nalloc, kc, c.s);
    }
    // precision 10: the round-5 launches of the default arithmetic (plane_gemm writing raw rows + ln_rows), for comparison with the
    // fused launch (precision 2 since round 6)
    const bool two_launch = precision == 10;
    if (precision == 5 || precision == 10) precision = 2;
    const bool fused = planes && !two_launch && pg_dbg == 0 && Cout % 64 == 0 && Cout <= 1024;
    float* dst = nullptr; int* derr = nullptr; unsigned epoch = 0;
    if (fused) {
        dst = (float*)c.alloc<unsigned char>(plane_gemm_ln_stats_bytes(M, Cout)); derr = c.alloc<int>(1);
        if (!c.ok) return OPH_ERR_DEVICE;
        hipMemsetAsync(dst, 0, plane_gemm_ln_stats_bytes(M, Cout), c.s); hipMemsetAsync(derr, 0, 4, c.s);
    }
    hipStreamSynchronize(c.s);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { g_op_error = "event creation failed"; return OPH_ERR_DEVICE; }
    auto once = [&]() {
        if (planes) {
            PlaneGemmArgs pg{};
            pg.Ah = (const _Float16*)dxh; pg.Al = (const _Float16*)dxl; pg.Wh = (const _Float16*)kweh; pg.Wl = (const _Float16*)kwel;
            pg.Wh2 = (const _Float16*)kwoh; pg.Wl2 = (const _Float16*)kwol; pg.bias = dbias; pg.H = dh; pg.M = M; pg.N = Cout; pg.kc = kc; pg.T = T;
            pg.nalloc = Nalloc; pg.ldh = 2 * Nalloc; pg.ntaps = 2; pg.off[0] = 0; pg.off[1] = -1; pg.convt = 1; pg.dbg = pg_dbg;
            if (fused) {
                pg.ln_gamma = dg; pg.ln_beta = db; pg.Y = dy; pg.ldy = Cout; pg.Yh = (_Float16*)dyh; pg.Yl = (_Float16*)dyl;
                pg.ln_stats = dst; pg.ln_epoch = ++epoch; pg.ln_err = derr; pg.dbg = ln_dbg;
                launch_plane_gemm(pg, c.s);
                return;
            }
            launch_plane_gemm(pg, c.s);
            EpiArgs e{};
            e.H = dh; e.ldh = Nalloc; e.M = 2 * M; e.C = Cout; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = dg; e.b1 = db; e.Y = dy; e.ldy = Cout; e.ypad = round_up(Cout, 32);
            e.planes = 1; e.Yh = dyh; e.Yl = dyl;
            if (e.ldy < e.ypad) e.ypad = e.ldy;
            launch_epilogue(e, c.s);
            return;
        }
        GemmArgs g{};
        g.X = dx; g.ldx = kc; g.bias = dbias; g.ldh = 2 * Nalloc; g.M = M; g.N = Cout; g.kc = kc; g.mode = 0; g.T = T;
        g.Wt = dwe; g.Wh = dweh; g.Wl = dwel; g.f16 = precision >= 2; g.nprod = precision == 3 ? 2 : (precision == 4 ? 1 : 3); g.ldw = 2 * kc; g.ntaps = 2; g.off[0] = 0; g.off[1] = -1; g.H = dh;
        GemmArgs g2 = g;
        g2.Wt = dwo; g2.Wh = dwoh; g2.Wl = dwol; g2.ldw = kc; g2.ntaps = 1; g2.off[0] = 0; g2.H = dh + Nalloc;
        launch_conv_gemm_pair(g, g2, precision < 0 || precision > 4 ? 0 : std::min(precision, 2), c.s);
        EpiArgs e{};
        e.H = dh; e.ldh = Nalloc; e.M = 2 * M; e.C = Cout; e.mode = PRE_CONV; e.act = ACT_NONE; e.g1 = dg; e.b1 = db; e.Y = dy; e.ldy = Cout; e.ypad = Cout;
        launch_epilogue(e, c.s);
    };
    for (int i = 0; i < warmup; ++i) once();
    hipEventRecord(e0, c.s);
    for (int i = 0; i < iters; ++i) once();
    hipEventRecord(e1, c.s);
    float ms = 0.f;
    hipError_t er = hipEventSynchronize(e1);
    if (er == hipSuccess) er = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (er != hipSuccess) { g_op_error = hipGetErrorString(er); return OPH_ERR_DEVICE; }
    if (fused) {
        int herr = 0;
        hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost);
        if (herr) { g_op_error = "conv1d_transpose + LayerNorm: statistics exchange timed out"; return OPH_ERR_DEVICE; }
    }
    *avg_us = (double)ms * 1e3 / iters;
    // per input row Cin*4 B in + 2*Cout*4 B out, + the 3*Cin*Cout weights once per call; 2*3*Cin*Cout flop per input row
    if (alg_bytes) *alg_bytes = ((double)M * Cin + 2.0 * M * Cout + 3.0 * Cin * Cout) * 4.0;
    if (alg_flops) *alg_flops = 2.0 * 3.0 * (double)M * Cin * Cout;
    return c.finish();
}

int oph_op_attention(int device, const float* Q, const float* K, const float* V, const int32_t* prev_max, int B, int T,
                     int N, int d, int win, float* R, float* alignments, int64_t* max_attentions) {
    OpCtx c(device);
    if (!c.ok) return OPH_ERR_DEVICE;
    if (d % 4 || d > 512 || win < 1 || win > 8) { g_op_error = "d/win out of range"; return OPH_ERR_UNSUPPORTED; }
    for (int b = 0; b < B; ++b) if (prev_max[b] < 0 || prev_max[b] >= N) { g_op_error = "prev_max out of range"; return OPH_ERR_INVALID; }
    float* dq = c.up(Q, (size_t)B * T * d); float* dk = c.up(K, (size_t)B * N * d); float* dv = c.up(V, (size_t)B * N * d);
    int* dp = c.up(prev_max, (size_t)B);
    float* dr = c.alloc<float>((size_t)B * T * 2 * d); float* da = c.alloc<float>((size_t)B * N * T);
    long long* dm = c.alloc<long long>((size_t)B * T);
    if (!c.ok) return OPH_ERR_DEVICE;
    AttnRowsArgs a{};
    a.mode = 1; a.Q = dq; a.ldq = d; a.K = dk; a.V = dv; a.ldkv = d; a.N = N; a.d = d; a.win = win; a.p = dp; a.B = B; a.Bpad = B;
    a.nrows = B * T; a.T = T; a.R = dr; a.ldr = 2 * d; a.align = da; a.amax = dm;
    launch_attn_rows(a, c.s);
    hipMemcpyAsync(R, dr, (size_t)B * T * 2 * d * 4, hipMemcpyDeviceToHost, c.s);
    hipMemcpyAsync(alignments, da, (size_t)B * N * T * 4, hipMemcpyDeviceToHost, c.s);
    hipMemcpyAsync(max_attentions, dm, (size_t)B * T * 8, hipMemcpyDeviceToHost, c.s);
    return c.finish();
}

}  // extern "C"

