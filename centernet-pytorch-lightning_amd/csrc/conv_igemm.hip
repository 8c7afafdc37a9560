// Implicit-GEMM convolution for gfx950 (NHWC activations, packed weights [Co_pad][taps*Ci]).
//
//   D[channel][pixel] = sum_k Wp[channel][k] * Xg[pixel][k]        (k = tap*Ci + ci)
//
// One workgroup = 4 waves (256 threads) computes a BM=128 pixel x BN channel tile.  K is walked in
// BK-element slices that never straddle a tap (Ci % BK == 0), so an A slice is BK contiguous
// channels of one (possibly out-of-image -> zero) input pixel: 16-byte coalesced NHWC loads.
// Slices are register-prefetched one step ahead and staged through double-buffered LDS (one
// barrier per step); rows are padded by 16 B so the ds_read_b128 fragment reads are conflict-free.
// bf16 uses v_mfma_f32_32x32x16_bf16, fp32 (parity mode) v_mfma_f32_32x32x2_f32 — both accumulate
// in fp32.  The weight tile is the FIRST MFMA operand so each lane ends up with 4 consecutive
// output channels of one pixel per accumulator quad -> 8/16-byte NHWC stores.
//
// "transposed" mode (ConvTranspose2d forward and the data gradient of a strided conv) is
// decomposed by output parity class (blockIdx.z): only the taps whose offset divides the stride
// are visited, so no MACs are wasted.  Geometry per class comes as small tap tables in the
// kernel arguments (built by the host below).
#include "conv_common.h"
#include <stdlib.h>

template <typename T, int BN, int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvGeom g) {
    constexpr int BM = 128;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int VPR = BK / VEC;         // 16-byte vectors per tile row
    constexpr int RPP = 256 / VPR;        // rows loaded per pass
    constexpr int APASS = BM / RPP;
    constexpr int BPASS = (BN + RPP - 1) / RPP;
    constexpr int PITCH = BK + Mma<T>::PAD;
    constexpr int WGN = (BN >= 64) ? 2 : 1;   // waves along channels
    constexpr int WGM = 4 / WGN;              // waves along pixels
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = BK / Mma<T>::KSTEP;

    __shared__ __attribute__((aligned(16))) T lds[2 * (BM + BN) * PITCH];
    constexpr int BUF = (BM + BN) * PITCH;  // elements per pipeline stage: [A tile | B tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cls = blockIdx.z;
    const int ph = cls / g.so, pw = cls % g.so;
    const int OHc = (g.OH - ph + g.so - 1) / g.so, OWc = (g.OW - pw + g.so - 1) / g.so;
    const int Mc = g.N * OHc * OWc;
    const int m0 = blockIdx.x * BM;
    if (m0 >= Mc) return;
    const int n0 = blockIdx.y * BN;
    const int ntaps = g.ntaps[cls];
    const int cpt = g.Ci / BK;  // K slices per tap
    const int nsteps = ntaps * cpt;

    const T* __restrict__ X = reinterpret_cast<const T*>(g.x);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(g.w);

    // ---- loader coordinates (fixed per thread) ----
    const int vcol = (tid % VPR) * VEC;
    const int lrow = tid / VPR;
    int a_ihb[APASS], a_iwb[APASS];
    int64_t a_img[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        int m = m0 + lrow + p * RPP;
        if (m < Mc) {
            int n = m / (OHc * OWc);
            int r = m - n * (OHc * OWc);
            int oh = r / OWc, ow = r - oh * OWc;
            a_ihb[p] = oh * g.sm;
            a_iwb[p] = ow * g.sm;
            a_img[p] = (int64_t)n * g.H * g.W;
        } else {
            a_ihb[p] = -100000;  // never in bounds
            a_iwb[p] = 0;
            a_img[p] = 0;
        }
    }

    uint4 ra[APASS], rb[BPASS];
    auto gload = [&](int step) {
        const int t = step / cpt;
        const int c0 = (step - t * cpt) * BK;
        const int dh = g.dh[cls][t], dw = g.dw[cls][t];
        const int wofs = (int)g.wt[cls][t] * g.Ci + c0 + vcol;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            int ih = a_ihb[p] + dh, iw = a_iwb[p] + dw;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
                v = *reinterpret_cast<const uint4*>(X + (a_img[p] + (int64_t)ih * g.W + iw) * g.x_ld + c0 + vcol);
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int row = lrow + p * RPP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < BN && n0 + row < g.co_pad)
                v = *reinterpret_cast<const uint4*>(Wp + (int64_t)(n0 + row) * g.ktot + wofs);
            rb[p] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) lds_store_vec<T, PITCH>((lds + buf * BUF), lrow + p * RPP, vcol, ra[p]);
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int row = lrow + p * RPP;
            if (row < BN) lds_store_vec<T, PITCH>((lds + buf * BUF + BM * PITCH), row, vcol, rb[p]);
        }
    };

    f32x16_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;

    if (nsteps > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < nsteps; ++step) {
        const bool more = step + 1 < nsteps;
        if (more) gload(step + 1);
        const T* at = (lds + cur * BUF);
        const T* bt = (lds + cur * BUF + BM * PITCH);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            typename Mma<T>::Frag fa[MI], fb[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = Mma<T>::load(at, PITCH, wm + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = Mma<T>::load(bt, PITCH, wn + j * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = Mma<T>::mma(fb[j], fa[i], acc[j][i]);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----
    int64_t pix[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm + i * 32 + (lane & 31);
        if (m >= Mc) { pix[i] = -1; continue; }
        if (g.so == 1) {
            pix[i] = m;
        } else {
            int n = m / (OHc * OWc);
            int r = m - n * (OHc * OWc);
            int oh = r / OWc, ow = r - oh * OWc;
            pix[i] = ((int64_t)n * g.OH + oh * g.so + ph) * g.OW + ow * g.so + pw;
        }
    }
    conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane);
}

// ------------------------------------------------------------------------------------------------ host
static int build_geom(ConvGeom& g, int KH, int KW, int stride, int pad, int transposed) {
    if (!transposed) {
        g.sm = stride;
        g.so = 1;
        int nt = KH * KW;
        if (nt > CN_MAX_TAPS) return -1;
        g.ntaps[0] = nt;
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) {
                int t = kh * KW + kw;
                g.dh[0][t] = (signed char)(kh - pad);
                g.dw[0][t] = (signed char)(kw - pad);
                g.wt[0][t] = (unsigned char)t;
            }
        return 1;
    }
    if (stride * stride > CN_MAX_CLS) return -1;
    g.sm = 1;
    g.so = stride;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            int c = ph * stride + pw, nt = 0;
            for (int kh = 0; kh < KH; ++kh) {
                if (((ph + pad - kh) % stride + stride) % stride != 0) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    if (((pw + pad - kw) % stride + stride) % stride != 0) continue;
                    if (nt >= CN_MAX_TAPS) return -1;
                    // ih = (ohc*stride + ph + pad - kh)/stride = ohc + (ph + pad - kh)/stride  (exact)
                    int qh = ph + pad - kh, qw = pw + pad - kw;
                    g.dh[c][nt] = (signed char)(qh >= 0 ? qh / stride : -((-qh) / stride));
                    g.dw[c][nt] = (signed char)(qw >= 0 ? qw / stride : -((-qw) / stride));
                    g.wt[c][nt] = (unsigned char)(kh * KW + kw);
                    ++nt;
                }
            }
            g.ntaps[c] = nt;
        }
    return stride * stride;
}

template <typename T, int BN, int BK>
static void launch_igemm(const ConvGeom& g, int ncls, hipStream_t st) {
    int so = g.so;
    int OHc = (g.OH + so - 1) / so, OWc = (g.OW + so - 1) / so;
    int64_t Mc = (int64_t)g.N * OHc * OWc;
    dim3 grid(cdiv(Mc, 128), cdiv(g.Co, BN), ncls);
    hipLaunchKernelGGL((conv_igemm_kernel<T, BN, BK>), grid, dim3(256), 0, st, g);
}

// tile choice: BN = smallest padded waste over {32,64,128} (ties -> larger); BK by divisibility of Ci
static void pick_tile(int Ci, int Co, int dtype, int* bn_out, int* bk_out) {
    int co32 = (Co + 31) / 32 * 32;
    int best = 32, bestw = co32;
    for (int bn : {64, 128}) {
        int w = (co32 + bn - 1) / bn * bn;
        if (w <= bestw) { best = bn; bestw = w; }
    }
    int bk = 16;
    if (dtype == CN_BF16) {
        bk = (Ci % 64 == 0) ? 64 : (Ci % 32 == 0 ? 32 : 16);
        if (best == 128 && bk == 64) bk = 32;  // keep LDS <= 40 KB/block for 128x128 tiles
    }
    *bn_out = best;
    *bk_out = bk;
}

static bool use_conv3x3(int KH, int KW, int stride, int pad, int H, int W, int OH, int OW) {
    static const bool disabled = getenv("CN_DISABLE_CONV3X3") != nullptr;   // A/B switch for profiling
    return !disabled && KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == H && OW == W;
}

extern "C" int cn_conv2d_variant(int Ci, int Co, int KH, int KW, int stride, int pad, int dtype) {
    int bn, bk;
    pick_tile(Ci, Co, dtype, &bn, &bk);
    if (use_conv3x3(KH, KW, stride, pad, 1, 1, 1, 1)) {
        int ck = dtype == CN_BF16 ? (Ci % 64 == 0 ? 64 : (Ci % 32 == 0 ? 32 : 16)) : 16;
        return 3000000 + bn * 1000 + ck;        // conv3x3s1_kernel<T, BN, CK>
    }
    return bn * 1000 + bk;                      // conv_igemm_kernel<T, BN, BK>
}

template <typename T>
static int dispatch_igemm(const ConvGeom& g, int ncls, hipStream_t st) {
    int best, bk;
    pick_tile(g.Ci, g.Co, sizeof(T) == 2 ? CN_BF16 : CN_F32, &best, &bk);
#define CN_IG(BN_, BK_) launch_igemm<T, BN_, BK_>(g, ncls, st)
    if constexpr (sizeof(T) == 2) {
        if (best == 128) { if (bk == 32) CN_IG(128, 32); else CN_IG(128, 16); }
        else if (best == 64) { if (bk == 64) CN_IG(64, 64); else if (bk == 32) CN_IG(64, 32); else CN_IG(64, 16); }
        else { if (bk == 64) CN_IG(32, 64); else if (bk == 32) CN_IG(32, 32); else CN_IG(32, 16); }
    } else {
        if (best == 128) CN_IG(128, 16); else if (best == 64) CN_IG(64, 16); else CN_IG(32, 16);
    }
#undef CN_IG
    return 0;
}

extern "C" int cn_conv2d_fwd(const void* x, const void* wp, const float* bias, const void* residual, void* y,
                             int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int y_ld, int res_ld,
                             int KH, int KW, int stride, int pad, int transposed, int relu, int dtype, int out_dtype,
                             void* stream) {
    CN_CHECK_ARG(x && wp && y, "cn_conv2d_fwd: null pointer");
    CN_CHECK_ARG(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && Co > 0, "cn_conv2d_fwd: bad dims");
    if (Ci % 16 != 0 || Ci <= 0) CN_UNSUPPORTED("cn_conv2d_fwd: Ci=%d must be a positive multiple of 16", Ci);
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(x_ld % V == 0 && x_ld >= Ci && y_ld >= Co, "cn_conv2d_fwd: bad pitches x_ld=%d y_ld=%d", x_ld, y_ld);
    CN_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                     ((uintptr_t)residual & 15) == 0,
                 "cn_conv2d_fwd: pointers must be 16-byte aligned");
    if (stride != 1 && stride != 2) CN_UNSUPPORTED("cn_conv2d_fwd: stride %d", stride);
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.w = wp; g.bias = bias; g.res = residual; g.y = y;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.OH = OH; g.OW = OW; g.Co = Co; g.y_ld = y_ld;
    g.res_ld = res_ld; g.ktot = KH * KW * Ci; g.co_pad = (Co + 31) / 32 * 32; g.relu = relu;
    g.y_f32 = (out_dtype == CN_F32);
    CN_CHECK_ARG(out_dtype == dtype || out_dtype == CN_F32, "cn_conv2d_fwd: out_dtype must be the compute dtype or fp32");
    CN_CHECK_ARG(!(residual && out_dtype != dtype), "cn_conv2d_fwd: residual needs out_dtype == dtype");
    int ncls = build_geom(g, KH, KW, stride, pad, transposed);
    if (ncls < 0) CN_UNSUPPORTED("cn_conv2d_fwd: kernel %dx%d stride %d not supported", KH, KW, stride);
    if (dtype != CN_F32 && dtype != CN_BF16) CN_CHECK_ARG(false, "cn_conv2d_fwd: bad dtype %d", dtype);
    if (use_conv3x3(KH, KW, stride, pad, H, W, OH, OW) && conv3x3s1_launch(g, dtype, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(3x3)");
        return CN_OK;
    }
    if (dtype == CN_F32) dispatch_igemm<float>(g, ncls, (hipStream_t)stream);
    else if (dtype == CN_BF16) dispatch_igemm<bf16_t>(g, ncls, (hipStream_t)stream);
    else CN_CHECK_ARG(false, "cn_conv2d_fwd: bad dtype %d", dtype);
    CN_LAUNCH_CHECK("cn_conv2d_fwd");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ packing
// mode 0: rows = B, k = t*inner_pad + a   (Wp[b][t*ip + a] = W[a][b][t])
// mode 1: rows = A, k = t*inner_pad + b   (Wp[a][t*ip + b] = W[a][b][t])
// mode 2: rows = (t, b) = t*B + b, k = a  (Wp[t*B + b][a]  = W[a][b][t])   inner_pad pads k
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp, int A, int B,
                                                          int taps, int mode, int rows_pad, int inner_pad,
                                                          const float* __restrict__ row_scale) {
    const int64_t ktot = mode == 2 ? inner_pad : (int64_t)taps * inner_pad;
    const int64_t total = (int64_t)rows_pad * ktot;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ktot);
        const int k = (int)(i - (int64_t)r * ktot);
        int a = -1, b = -1, t = 0;
        if (mode == 2) {
            t = r / B; b = r - t * B; a = k;
            if (t >= taps) b = -1;
        } else {
            t = k / inner_pad;
            const int c = k - t * inner_pad;
            if (mode == 1) { a = r; b = c; } else { a = c; b = r; }
        }
        float v = 0.f;
        if (a >= 0 && a < A && b >= 0 && b < B) {
            v = w[((int64_t)a * B + b) * taps + t];
            if (row_scale) v *= row_scale[r];
        }
        Elem<T>::st(wp + i, v);
    }
}

extern "C" int cn_pack_weight(const float* w, void* wp, int A, int B, int KH, int KW, int mode, int rows_pad, int inner_pad,
                              const float* row_scale, int dtype, void* stream) {
    CN_CHECK_ARG(w && wp && A > 0 && B > 0 && KH > 0 && KW > 0 && mode >= 0 && mode <= 2, "cn_pack_weight: bad args");
    const int taps = KH * KW;
    const int rows = mode == 0 ? B : (mode == 1 ? A : taps * B);
    const int inner = mode == 0 ? A : (mode == 1 ? B : A);
    CN_CHECK_ARG(rows_pad >= rows && inner_pad >= inner, "cn_pack_weight: rows_pad %d < %d or inner_pad %d < %d", rows_pad, rows,
                 inner_pad, inner);
    int64_t total = (int64_t)rows_pad * (mode == 2 ? inner_pad : (int64_t)taps * inner_pad);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                                                   (T*)wp, A, B, taps, mode, rows_pad, inner_pad, row_scale));
    CN_LAUNCH_CHECK("cn_pack_weight");
    return CN_OK;
}

// dw[a][b][t] = dwp[a][t*inner_pad + b]   (inverse of mode 1)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int A,
                                                           int B, int taps, int inner_pad) {
    const int64_t total = (int64_t)A * B * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % taps);
        int64_t ab = i / taps;
        int b = (int)(ab % B), a = (int)(ab / B);
        dw[i] = dwp[(int64_t)a * taps * inner_pad + (int64_t)t * inner_pad + b];
    }
}

extern "C" int cn_unpack_wgrad(const float* dwp, float* dw, int A, int B, int KH, int KW, int inner_pad, void* stream) {
    CN_CHECK_ARG(dwp && dw && A > 0 && B > 0 && inner_pad >= B, "cn_unpack_wgrad: bad args");
    int64_t total = (int64_t)A * B * KH * KW;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dwp, dw, A, B, KH * KW, inner_pad);
    CN_LAUNCH_CHECK("cn_unpack_wgrad");
    return CN_OK;
}
