// Weight gradients of the two 512x512 layers with <= 16 channels on both sides (bf16 compute mode):
//   * DLA level0   conv3x3 16 -> 16   (x: NHWC bf16)                dW[co][kh,kw][ci] += sum_p dY[p][co] x[p+(kh,kw)][ci]
//   * the 7x7 stem conv 3 -> 16       (x: the fp32 NCHW input image)
// Both are HBM streams (0.6-1 GB, ~80 GFLOP); the VALU kernel they used to share spent 2-3.5 ms on LDS reads.  Here one
// WAVE owns an 8x16 pixel tile: dY tile [128 px][16 co] and the x halo tile [px][XPIX] sit in the wave's own LDS slab in
// their natural pixel-major order and both MFMA operands come out of them with the transposing LDS read
// (ds_read_b64_tr_b16), feeding v_mfma_f32_16x16x32_bf16 with M = co, K = 32 pixels, N = 16 columns:
//   XPIX = 16 (NHWC x):   N = ci, one MFMA per (tap, 32-pixel chunk);
//   XPIX = 4  (stem):     the halo is stored as {c0, c1, c2, 0} per pixel, so 16 consecutive bf16 starting at pixel q are
//                         the 4 pixels q..q+3 x 4 channels = the columns n = (kw & 3, ci) of FOUR taps at once; two MFMAs
//                         cover the 7 (+1 masked) kw of a kernel row.
// The next tile is fetched into registers while the current one is multiplied; a workgroup's four waves fold their
// accumulators through LDS and flush with one round of fp32 atomics.
#include "dcn_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define C16_TH 8
#define C16_TW 16

struct WgC16Geom {
    int target;               // workgroup budget of this call (cn_hooks.wgrad_blocks)
    const void* x;
    const bf16_t* dy;
    float* dw;
    int N, H, W, Ci, x_ld, Co, dy_ld, OH, OW;      // dy is [N][OH][OW][dy_ld]; a launch's grid.y walks 16-channel blocks of Co
    int os_co, os_ci, os_tap;          // dw[co*os_co + ci*os_ci + (kh*KW+kw)*os_tap]
    int tiles_h, tiles_w, iters;
    // BNB: dy is the gradient w.r.t. the OUTPUT of the training-mode BatchNorm (+ ReLU) that follows this conv, bn_x the conv's own raw
    // output; the kernel forms the BN input gradient on load: g = relu ? (fma(x, sc, sh) > 0 ? dy : 0) : dy, dx = fma(ca, g, fma(cp, x, cq))
    // (bn_bwd_apply_kernel's arithmetic), bn_coef = fp32 [5][bn_C]: ca | cp | cq | sc | sh (cn_bn_bwd_coef_sink)
    const bf16_t* bn_x; const float* bn_coef; int bn_C, bn_relu;
    const float* pre_ss; int pre_relu;   // input pre-affine (cn_hooks.pre_ss): x' = bf16(fma(x, ss[c], ss[16 + c])), relu: max(., 0); padding stays 0
};

// 8 consecutive K (pixel) values of 16 columns out of a pixel-major LDS tile: rows are LDS element offsets row_of(k)
template <typename RowFn>
__device__ static inline bf16x8_t tr_frag_k32(const bf16_t* tile, int lane, RowFn row_of) {
    const int r = lane & 15, g4 = lane >> 4;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + row_of(8 * g4 + (r >> 2)) + 4 * (r & 3)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + row_of(8 * g4 + 4 + (r >> 2)) + 4 * (r & 3)));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// VEC4 (XPIX == 4, S == 1, W % 4 == 0): the image halo is fetched with 16-byte loads — four pixels of one fp32 plane, the window starting at
// the aligned column tw0 - 4 — instead of one 4-byte load per pixel and plane (18 -> 6 wave-level load instructions per tile: the texture
// path issues one every ~16 cycles whatever its width)
template <int XPIX, int KH, int KW, int S = 1, bool AFF = false, bool BNB = false, bool VEC4 = false>
__global__ __launch_bounds__(256) void wgrad_c16_kernel(const WgC16Geom g) {
    static_assert(!AFF || XPIX == 16, "the pre-affine is for the NHWC bf16 input");
    static_assert(!VEC4 || (XPIX == 4 && S == 1 && KW == 7), "VEC4 is the stride-1 7x7 stem");
    constexpr int PAD = KH / 2;
    constexpr int HH = (C16_TH - 1) * S + KH;
    constexpr int KWG = XPIX == 16 ? KW : (KW + 3) / 4;                       // MFMA column groups per kernel row
    constexpr int HWD = VEC4 ? 24 : (XPIX == 16 ? (C16_TW - 1) * S + KW : (C16_TW - 1) * S + 4 * KWG);   // halo row length (the last 4-pixel window starts at S*tx + 4*(KWG-1)); VEC4: columns tw0 - 4 .. tw0 + 19
    constexpr int XO = VEC4 ? 1 : 0;                                          // column of the halo's first USED pixel (tw0 - PAD)
    constexpr int HP = HH * HWD;
    constexpr int DY_ELEMS = C16_TH * C16_TW * 16;
    constexpr int X_ELEMS = HP * XPIX;
    constexpr int SLAB = DY_ELEMS + X_ELEMS;
    constexpr int NACC = KH * KWG;
    constexpr int RED_ELEMS = NACC * 4 * 64 * 4 * 2;                          // fp32 [wave][NACC*4][64] of the final fold, in bf16 units
    static_assert(SLAB % 8 == 0, "wave slabs must stay 16-byte aligned");
    __shared__ __attribute__((aligned(16))) bf16_t lds[4 * SLAB > RED_ELEMS ? 4 * SLAB : RED_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16_t* const dyt = lds + wave * SLAB;
    bf16_t* const xh = dyt + DY_ELEMS;
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int co0 = blockIdx.y * 16;

    f32x4_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int DYV = DY_ELEMS / 8 / 64;                                    // 16-byte vectors per lane (4)
    constexpr int XV = XPIX == 16 ? (HP * 2 + 63) / 64 : (VEC4 ? (HH * 6 + 63) / 64 : (HP + 63) / 64);      // XPIX 16: 16-byte vectors;  XPIX 4: pixels (3 floats each); VEC4: (row, 4-pixel group) items
    uint4 rdy[DYV];
    uint4 rx16[XPIX == 16 ? XV : 1];
    float rx4[(XPIX == 4 && !VEC4) ? XV : 1][3];
    float4 rv4[VEC4 ? XV : 1][3];
    uint4 rbx[BNB ? DYV : 1];                           // BNB: the conv's raw output at the dy positions
    uint32_t dym = 0;                                   // BNB: which of this lane's dy vectors are real (the others stay zero)
    float bca[BNB ? 8 : 1], bcp[BNB ? 8 : 1], bcq[BNB ? 8 : 1], bsc[BNB ? 8 : 1], bsh[BNB ? 8 : 1];   // this lane's 8 channels: co0 + 8 (lane & 1) .. +7
    if constexpr (BNB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = co0 + (lane & 1) * 8 + j;
            const bool in = c < g.bn_C;
            bca[j] = in ? g.bn_coef[c] : 0.f; bcp[j] = in ? g.bn_coef[g.bn_C + c] : 0.f; bcq[j] = in ? g.bn_coef[2 * g.bn_C + c] : 0.f;
            bsc[j] = in ? g.bn_coef[3 * g.bn_C + c] : 0.f; bsh[j] = in ? g.bn_coef[4 * g.bn_C + c] : 0.f;
        }
    }
    uint32_t okm = 0;                                   // AFF: which of this lane's halo vectors lie inside the image (the others stay zero)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef short s16x2_ __attribute__((ext_vector_type(2)));
    f32x2_ sc[4], sh[4];                                // this lane's 8 channels: 8 (lane & 1) .. +7
    if constexpr (AFF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = f32x2_{g.pre_ss[(lane & 1) * 8 + 2 * i], g.pre_ss[(lane & 1) * 8 + 2 * i + 1]};
            sh[i] = f32x2_{g.pre_ss[16 + (lane & 1) * 8 + 2 * i], g.pre_ss[16 + (lane & 1) * 8 + 2 * i + 1]};
        }
    }

    auto gload = [&](int64_t tile) {
        const bool tv = tile < ntiles;
        const int64_t tc = tv ? tile : 0;
        const int n = (int)(tc / (g.tiles_h * g.tiles_w));
        const int r = (int)(tc - (int64_t)n * g.tiles_h * g.tiles_w);
        const int th0 = (r / g.tiles_w) * C16_TH, tw0 = (r % g.tiles_w) * C16_TW;
#pragma unroll
        for (int v = 0; v < DYV; ++v) {
            const int idx = lane + v * 64;
            const int px = idx >> 1, hf = idx & 1;
            const int oh = th0 + px / C16_TW, ow = tw0 + px % C16_TW;
            const bool ok = tv && oh < g.OH && ow < g.OW && co0 + hf * 8 < g.dy_ld;
            rdy[v] = ldg16_masked(g.dy, ((((int64_t)n * g.OH + oh) * g.OW + ow) * g.dy_ld + co0 + hf * 8) * 2, ok);
            if constexpr (BNB) {
                if (v == 0) dym = 0;
                rbx[v] = ldg16_masked(g.bn_x, ((((int64_t)n * g.OH + oh) * g.OW + ow) * g.dy_ld + co0 + hf * 8) * 2, ok);
                if (ok) dym |= 1u << v;
            }
        }
        if constexpr (XPIX == 16) {
            okm = 0;
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                const int idx = lane + v * 64;
                const int hp = idx >> 1, hf = idx & 1;
                const int ih = th0 * S - PAD + hp / HWD, iw = tw0 * S - PAD + hp % HWD;
                const bool ok = tv && hp < HP && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
                rx16[v] = ldg16_masked(g.x, ((((int64_t)n * g.H + ih) * g.W + iw) * g.x_ld + hf * 8) * 2, ok);
                if (AFF && ok) okm |= 1u << v;
            }
        } else if constexpr (VEC4) {
            const float* X = reinterpret_cast<const float*>(g.x);
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                const int it = lane + v * 64;
                const int row = it / 6, q4 = it - row * 6;
                const int ih = th0 - PAD + row, iw = tw0 - 4 + 4 * q4;
                const bool ok = tv && it < HH * 6 && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const bool okc = ok && c < g.Ci;
                    const float4 q = *reinterpret_cast<const float4*>(X + (okc ? (((int64_t)n * g.Ci + c) * g.H + ih) * g.W + iw : 0));
                    rv4[v][c] = okc ? q : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
            const float* X = reinterpret_cast<const float*>(g.x);
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                const int hp = lane + v * 64;
                const int ih = th0 * S - PAD + hp / HWD, iw = tw0 * S - PAD + hp % HWD;
                const bool ok = tv && hp < HP && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const bool okc = ok && c < g.Ci;
                    const float val = X[okc ? (((int64_t)n * g.Ci + c) * g.H + ih) * g.W + iw : 0];
                    rx4[v][c] = okc ? val : 0.f;
                }
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int v = 0; v < DYV; ++v) {
            if constexpr (BNB) {                        // BN input gradient from (dy, x): what bn_bwd_apply_kernel would have stored
                float gv[8], xv[8];
                Vec16<bf16_t>::unpack(rdy[v], gv);
                Vec16<bf16_t>::unpack(rbx[v], xv);
                const bool ok = (dym >> v) & 1u;
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float d[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float gg = (!g.bn_relu || fmaf(xv[j + e], bsc[j + e], bsh[j + e]) > 0.f) ? gv[j + e] : 0.f;
                        d[e] = fmaf(bca[j + e], gg, fmaf(bcp[j + e], xv[j + e], bcq[j + e]));
                    }
                    o[j >> 1] = ok ? pk_bf16(d[0], d[1]) : 0u;
                }
                rdy[v] = make_uint4(o[0], o[1], o[2], o[3]);
            }
            *reinterpret_cast<uint4*>(dyt + (lane + v * 64) * 8) = rdy[v];
        }
        if constexpr (XPIX == 16) {
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                const int idx = lane + v * 64;
                if constexpr (AFF) {                    // the previous layer's BN (+ ReLU), as bn_fwd_apply_sink_kernel would have stored it
                    const bool ok = (okm >> v) & 1u;
                    const uint32_t w[4] = {rx16[v].x, rx16[v].y, rx16[v].z, rx16[v].w};
                    uint32_t o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2_ xv = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
                        const f32x2_ yv = __builtin_elementwise_fma(xv, sc[i], sh[i]);
                        uint32_t pk = pk_bf16(yv[0], yv[1]);
                        if (g.pre_relu) pk = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_, pk), s16x2_{0, 0}));
                        o[i] = ok ? pk : 0u;
                    }
                    rx16[v] = make_uint4(o[0], o[1], o[2], o[3]);
                }
                if (idx < HP * 2) *reinterpret_cast<uint4*>(xh + idx * 8) = rx16[v];
            }
        } else if constexpr (VEC4) {
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                const int it = lane + v * 64;
                if (it < HH * 6) {
                    uint4* d = reinterpret_cast<uint4*>(xh + it * 16);        // 4 pixels x {c0, c1, c2, 0}
                    d[0] = make_uint4(pk_bf16(rv4[v][0].x, rv4[v][1].x), pk_bf16(rv4[v][2].x, 0.f), pk_bf16(rv4[v][0].y, rv4[v][1].y), pk_bf16(rv4[v][2].y, 0.f));
                    d[1] = make_uint4(pk_bf16(rv4[v][0].z, rv4[v][1].z), pk_bf16(rv4[v][2].z, 0.f), pk_bf16(rv4[v][0].w, rv4[v][1].w), pk_bf16(rv4[v][2].w, 0.f));
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                const int hp = lane + v * 64;
                if (hp < HP) *reinterpret_cast<uint2*>(xh + hp * 4) = make_uint2(pk_bf16(rx4[v][0], rx4[v][1]), pk_bf16(rx4[v][2], 0.f));
            }
        }
    };

    const int64_t stride_t = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    gload(tile);
#pragma unroll 1
    for (int it = 0; it < g.iters; ++it, tile += stride_t) {
        __syncthreads();                       // every wave is done reading its previous tile
        lstore();
        __syncthreads();
        gload(tile + stride_t);                // in flight while this tile is multiplied (masked past the end)
#pragma unroll
        for (int c = 0; c < 4; ++c) {          // 32-pixel chunk = tile rows 2c, 2c+1
            const bf16x8_t fa = tr_frag_k32(dyt, lane, [&](int kk) { return (32 * c + kk) * 16; });
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                for (int kwg = 0; kwg < KWG; ++kwg) {
                    const int kwb = XPIX == 16 ? kwg : 4 * kwg;
                    const bf16x8_t fb = tr_frag_k32(xh, lane, [&](int kk) { return ((S * (2 * c + (kk >> 4)) + kh) * HWD + S * (kk & 15) + kwb + XO) * XPIX; });
                    acc[kh * KWG + kwg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[kh * KWG + kwg], 0, 0, 0);
                }
        }
    }

    // fold the four waves, then one round of atomics per workgroup.  D[co = 4*(lane>>4)+r][n = lane&15]
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);                               // [wave][NACC*4][64]
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * NACC * 4 + a * 4 + r) * 64 + lane] = acc[a][r];
    __syncthreads();
    for (int i = tid; i < NACC * 4 * 64; i += 256) {
        const float s = red[i] + red[NACC * 4 * 64 + i] + red[2 * NACC * 4 * 64 + i] + red[3 * NACC * 4 * 64 + i];
        const int l = i & 63, r = (i >> 6) & 3, a = i >> 8;
        const int kh = a / KWG, kwg = a % KWG;
        const int co = co0 + 4 * (l >> 4) + r, nn = l & 15;
        const int ci = XPIX == 16 ? nn : (nn & 3);
        const int kw = XPIX == 16 ? kwg : 4 * kwg + (nn >> 2);
        if (co < g.Co && ci < g.Ci && kw < KW) atomicAdd(g.dw + (int64_t)co * g.os_co + (int64_t)ci * g.os_ci + (int64_t)(kh * KW + kw) * g.os_tap, s);
    }
}

template <int XPIX, int KH, int KW, int S = 1, bool AFF = false, bool BNB = false, bool VEC4 = false>
static void launch_c16(WgC16Geom& g, hipStream_t st) {
    g.tiles_h = cdiv(g.OH, C16_TH); g.tiles_w = cdiv(g.OW, C16_TW);
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    int64_t blocks = (ntiles + 3) / 4;
    const int cap = g.target / 3 < 128 ? 128 : g.target / 3;   // default 512
    if (blocks > cap) blocks = cap;
    g.iters = (int)((ntiles + blocks * 4 - 1) / (blocks * 4));
    hipLaunchKernelGGL((wgrad_c16_kernel<XPIX, KH, KW, S, AFF, BNB, VEC4>), dim3((unsigned)blocks, (unsigned)cdiv(g.Co, 16)), dim3(256), 0, st, g);
}

// bf16 NHWC x, 3x3 / stride 1|2 / pad 1, Ci == 16, Co in 16-channel blocks -> packed dwp[co][tap*16 + ci]
bool wgrad_c16_nhwc_launch(const void* x, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                           int stride, int OH, int OW, hipStream_t st, const float* pre_ss, int pre_relu, int target) {
    static const bool disabled = getenv("CN_DISABLE_WGRAD_C16") != nullptr;
    if (disabled || Ci != 16 || (x_ld & 7) || (dy_ld & 7) || dy_ld < ((Co + 15) & ~15) || (stride != 1 && stride != 2)) return false;
    WgC16Geom g;
    g.x = x; g.dy = (const bf16_t*)dy; g.dw = dwp; g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.Co = Co; g.dy_ld = dy_ld;
    g.OH = OH; g.OW = OW;
    g.os_co = 9 * Ci; g.os_ci = 1; g.os_tap = Ci; g.target = target;
    g.pre_ss = pre_ss; g.pre_relu = pre_relu;
    g.bn_x = nullptr; g.bn_coef = nullptr; g.bn_C = 0; g.bn_relu = 0;
    if (pre_ss) { if (stride == 1) launch_c16<16, 3, 3, 1, true>(g, st); else launch_c16<16, 3, 3, 2, true>(g, st); }
    else if (stride == 1) launch_c16<16, 3, 3, 1>(g, st); else launch_c16<16, 3, 3, 2>(g, st);
    return true;
}

// fp32 NCHW x (the image), 7x7 / stride 1|2 / pad 3, Ci <= 3, Co % 16 == 0 -> dw[co][ci][kh][kw]
// bn_x != nullptr: dy is the gradient w.r.t. the output of the BatchNorm (+ ReLU) behind the stem, see WgC16Geom
bool wgrad_c16_stem_launch(const float* x, const void* dy, float* dw, int N, int Ci, int H, int W, int Co, int dy_ld, int stride,
                           int OH, int OW, hipStream_t st, const void* bn_x, const float* bn_coef, int bn_relu, int target) {
    static const bool disabled = getenv("CN_DISABLE_WGRAD_C16") != nullptr;
    if (disabled || Ci > 3 || (Co & 15) || (dy_ld & 7) || dy_ld < Co || (stride != 1 && stride != 2)) return false;
    WgC16Geom g;
    g.x = x; g.dy = (const bf16_t*)dy; g.dw = dw; g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = 0; g.Co = Co; g.dy_ld = dy_ld;
    g.OH = OH; g.OW = OW;
    g.os_co = Ci * 49; g.os_ci = 49; g.os_tap = 1; g.target = target;
    g.pre_ss = nullptr; g.pre_relu = 0;
    g.bn_x = (const bf16_t*)bn_x; g.bn_coef = bn_coef; g.bn_C = Co; g.bn_relu = bn_relu;
    static const bool no_vec4 = getenv("CN_DISABLE_STEM_WGRAD_VEC4") != nullptr;
    const bool vec4 = !no_vec4 && stride == 1 && (W & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    if (bn_x) {
        if (dy_ld != Co) return false;               // bn_x is read with dy's pitch
        if (vec4) launch_c16<4, 7, 7, 1, false, true, true>(g, st);
        else if (stride == 1) launch_c16<4, 7, 7, 1, false, true>(g, st); else launch_c16<4, 7, 7, 2, false, true>(g, st);
        return true;
    }
    if (vec4) { launch_c16<4, 7, 7, 1, false, false, true>(g, st); return true; }
    if (stride == 1) launch_c16<4, 7, 7, 1>(g, st); else launch_c16<4, 7, 7, 2>(g, st);
    return true;
}

// ---- 7x7 stem forward on the matrix cores (bf16 mode) --------------------------------------------------------------
// y[p][co] = sum_{kh} sum_{(kw,ci)} w[co][ci][kh][kw] * x[ci][p*S + (kh-3, kw-3)]   (S = 1: DLA base layer, S = 2: ResNet stem).
// With the halo stored as {c0,c1,c2,0} per input pixel, a whole kernel ROW (8 kw slots x 4 channel slots, 7 x 3 real) is
// exactly K = 32 of one v_mfma_f32_16x16x32_bf16: A = weights [co][k = 4*kw + ci] (one fragment per kernel row and 16-channel
// block, built once per workgroup in LDS), B[k][px]: lane (px = lane&15, g4 = lane>>4) needs k = 8*g4..+7 = input pixels
// px*S+2*g4, px*S+2*g4+1 x 4 channels = 16 contiguous bytes of the halo row.  One wave = one 8x16 output tile = 8 rows x 7
// MFMAs per 16-channel block; D[co = 4*(lane>>4)+r][px]: 8-byte stores.
#define ST7_MAXCB 4
template <int S>
__global__ __launch_bounds__(256) void stem7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                        int N, int Ci, int H, int W, int Co, int y_ld, int OH, int OW, int tiles_h,
                                                        int tiles_w, int iters, const float* __restrict__ scale, const float* __restrict__ bias,
                                                        int relu, float* __restrict__ bn_part, int bn_slots) {
    CN_MAIN_PRIO_SET();
    constexpr int HH = (C16_TH - 1) * S + 7, HWD = (C16_TW - 1) * S + 8, HP = HH * HWD;   // +1 column: the 8th (masked) kw slot
    constexpr int XV = (HP + 63) / 64;
    __shared__ __attribute__((aligned(16))) uint4 wfrag[ST7_MAXCB * 7 * 64];
    __shared__ __attribute__((aligned(16))) bf16_t halo[4 * HP * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16_t* const xh = halo + wave * HP * 4;
    const int px = lane & 15, g4 = lane >> 4;
    const int ncb = (Co + 15) / 16;
    const int64_t ntiles = (int64_t)N * tiles_h * tiles_w;

    for (int f = tid; f < ncb * 7 * 64; f += 256) {        // weight fragments: [cb][kh][lane]
        const int l = f & 63, kh = (f >> 6) % 7, cb = f / (7 * 64);
        const int co = cb * 16 + (l & 15), gg = l >> 4;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kw = 2 * gg + (j >> 2), ci = j & 3;
            const bool ok = co < Co && ci < Ci && kw < 7;
            const float val = w[ok ? ((int64_t)(co * Ci + ci) * 7 + kh) * 7 + kw : 0];
            v[j] = ok ? val : 0.f;
        }
        wfrag[f] = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
    }

    // BatchNorm statistics of the stored values (sink protocol of bn.hip; launched with Co <= 16 only: one channel block)
    const bool stats = bn_part != nullptr;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    float rx[XV][3];
    auto gload = [&](int64_t tile) {
        const bool tv = tile < ntiles;
        const int64_t tc = tv ? tile : 0;
        const int n = (int)(tc / (tiles_h * tiles_w));
        const int r = (int)(tc - (int64_t)n * tiles_h * tiles_w);
        const int th0 = (r / tiles_w) * C16_TH, tw0 = (r % tiles_w) * C16_TW;
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int hp = lane + v * 64;
            const int ih = th0 * S - 3 + hp / HWD, iw = tw0 * S - 3 + hp % HWD;
            const bool ok = tv && hp < HP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool okc = ok && c < Ci;
                const float val = x[okc ? (((int64_t)n * Ci + c) * H + ih) * W + iw : 0];
                rx[v][c] = okc ? val : 0.f;
            }
        }
    };
    const int64_t stride_t = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    gload(tile);
#pragma unroll 1
    for (int it = 0; it < iters; ++it, tile += stride_t) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int hp = lane + v * 64;
            if (hp < HP) *reinterpret_cast<uint2*>(xh + hp * 4) = make_uint2(pk_bf16(rx[v][0], rx[v][1]), pk_bf16(rx[v][2], 0.f));
        }
        __syncthreads();
        const bool tv = tile < ntiles;
        const int64_t tc = tv ? tile : 0;
        const int n = (int)(tc / (tiles_h * tiles_w));
        const int r = (int)(tc - (int64_t)n * tiles_h * tiles_w);
        const int th0 = (r / tiles_w) * C16_TH, tw0 = (r % tiles_w) * C16_TW;
        gload(tile + stride_t);
#pragma unroll 1
        for (int ty = 0; ty < C16_TH; ++ty) {
            bf16x8_t fb[7];
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const bf16_t* src = xh + ((ty * S + kh) * HWD + px * S + 2 * g4) * 4;      // 8-byte aligned: two ds_read_b64
                const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 4);
                fb[kh] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
            const int oh = th0 + ty, ow = tw0 + px;
            for (int cb = 0; cb < ncb; ++cb) {
                const int c0 = cb * 16 + 4 * g4;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kh = 0; kh < 7; ++kh)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag[(cb * 7 + kh) * 64 + lane]), fb[kh], acc, 0, 0, 0);
                if (scale || bias || relu) {                // folded eval-mode BN (+ReLU): y = act(fma(conv, scale, shift))
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool in = c0 + q < Co;
                        const float v = fmaf(acc[q], (scale && in) ? scale[c0 + q] : 1.f, (bias && in) ? bias[c0 + q] : 0.f);
                        acc[q] = relu ? fmaxf(v, 0.f) : v;
                    }
                }
                if (tv && oh < OH && ow < OW && c0 < Co) {
                    bf16_t* dst = y + (((int64_t)n * OH + oh) * OW + ow) * y_ld + c0;
                    if (c0 + 4 <= Co) {
                        const uint2 o = make_uint2(pk_bf16(acc[0], acc[1]), pk_bf16(acc[2], acc[3]));
                        *reinterpret_cast<uint2*>(dst) = o;
                        if (stats) {
                            const float a0 = __uint_as_float(o.x << 16), a1 = __uint_as_float(o.x & 0xffff0000u);
                            const float a2 = __uint_as_float(o.y << 16), a3 = __uint_as_float(o.y & 0xffff0000u);
                            s0[0] += a0; s0[1] += a1; s0[2] += a2; s0[3] += a3;
                            s1[0] = fmaf(a0, a0, s1[0]); s1[1] = fmaf(a1, a1, s1[1]); s1[2] = fmaf(a2, a2, s1[2]); s1[3] = fmaf(a3, a3, s1[3]);
                        }
                    }
                    else
                        for (int q = 0; q < 4 && c0 + q < Co; ++q) dst[q] = f2bf(acc[q]);
                }
            }
        }
    }
    if (stats) bn_stats_flush_c16<256>(s0, s1, reinterpret_cast<float*>(halo), bn_part, bn_slots, y_ld, Co, blockIdx.x, tid);
}

// ---- the same contraction, ROW-WALKING (stride 1, <= 16 output channels: DLA base_layer, 512x512 -> 16 channels) ----------------
// stem7_fwd_kernel moves every 8x16 tile's halo global -> registers -> LDS -> registers behind two barriers, with 4-byte loads from the
// fp32 planes (the texture path takes a wave-level load instruction every ~16 cycles whatever its width: 18 of them per 128 outputs);
// at 16 output channels the layer is an HBM stream (0.2 GB in, 0.54 GB out) and ran at 2.5 TB/s.  Here a workgroup owns a region of R
// output rows x 128 columns: its (R + 6) x 136 input window is fetched ONCE with 16-byte loads (four pixels of one plane), converted
// and parked in LDS as {c0, c1, c2, 0} bf16 per pixel, zeros outside the image; then every wave walks down a 32-pixel-wide column of
// output rows.  The B operand of an INPUT row — lane (px, g4): pixels (ow + 2 g4 - 3, + 1) x 4 channel slots = 16 bytes of the LDS row
// — is the same for all seven kernel rows that use that row, so a wave keeps the last eight input rows' operands in a register ring:
// one LDS read per lane, 16-pixel group and output row, 7 MFMAs, no VALU on the input side.  Images are dealt to the XCDs as in
// conv3x3_c16r_kernel.
#define ST7R_G 2
#define ST7R_RMAX 32
#define ST7R_COLS 136                                      // 4 + 128 + 4: the window starts at the (16-byte aligned) column x0 - 4
__global__ __launch_bounds__(256, 2) void stem7_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                            int N, int Ci, int H, int W, int Co, int y_ld, int R, int sblocks, int rblocks,
                                                            const float* __restrict__ scale, const float* __restrict__ bias, int relu,
                                                            float* __restrict__ bn_part, int bn_slots) {
    CN_MAIN_PRIO_SET();
    __shared__ __attribute__((aligned(16))) uint2 win[(ST7R_RMAX + 6) * ST7R_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 15, g4 = lane >> 4;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int bpi = sblocks * rblocks;
    const int n = (jb / bpi) * 8 + xcd;
    if (n >= N) return;                                   // whole workgroup
    const int within = jb % bpi, rb = within / sblocks, sb = within % sblocks;
    const int oh0 = rb * R, x0 = sb * 128, ow0 = x0 + wave * 16 * ST7R_G;

    // ---- the window: rows oh0 - 3 .. oh0 + R + 2, columns x0 - 4 .. x0 + 131 (W is a multiple of 4: a 4-pixel vector is in or out) ----
    const int64_t plane = (int64_t)H * W;
    const float* const xn = x + (int64_t)n * Ci * plane;
    const int nrows = R + 6;
#pragma unroll 1
    for (int base = tid; base < nrows * (ST7R_COLS / 4); base += 256 * 2) {
        float4 v[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = base + u * 256;
            const int row = it / (ST7R_COLS / 4), v4 = it - row * (ST7R_COLS / 4);
            const int ih = oh0 - 3 + row, iw = x0 - 4 + 4 * v4;
            const bool ok = it < nrows * (ST7R_COLS / 4) && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool okc = ok && c < Ci;
                const float4 q = *reinterpret_cast<const float4*>(xn + (okc ? c * plane + (int64_t)ih * W + iw : 0));
                v[u][c] = okc ? q : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = base + u * 256;
            if (it < nrows * (ST7R_COLS / 4)) {
                uint2* d = win + it * 4;                    // (row * COLS + 4 v4) pixels
                d[0] = make_uint2(pk_bf16(v[u][0].x, v[u][1].x), pk_bf16(v[u][2].x, 0.f));
                d[1] = make_uint2(pk_bf16(v[u][0].y, v[u][1].y), pk_bf16(v[u][2].y, 0.f));
                d[2] = make_uint2(pk_bf16(v[u][0].z, v[u][1].z), pk_bf16(v[u][2].z, 0.f));
                d[3] = make_uint2(pk_bf16(v[u][0].w, v[u][1].w), pk_bf16(v[u][2].w, 0.f));
            }
        }
    }

    // weight fragments of the seven kernel rows: A[co = lane & 15][k = 8 g4 + j] = w[co][ci = j & 3][kh][kw = 2 g4 + (j >> 2)]
    bf16x8_t wa[7];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kw = 2 * g4 + (j >> 2), ci = j & 3;
            const bool ok = px < Co && ci < Ci && kw < 7;
            const float val = w[ok ? ((int64_t)(px * Ci + ci) * 7 + kh) * 7 + kw : 0];
            v[j] = ok ? val : 0.f;
        }
        wa[kh] = __builtin_bit_cast(bf16x8_t, make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])));
    }
    float sc4[4], bi4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool in = 4 * g4 + q < Co;
        sc4[q] = (scale && in) ? scale[4 * g4 + q] : 1.f;
        bi4[q] = (bias && in) ? bias[4 * g4 + q] : 0.f;
    }
    const bool affine = scale || bias || relu;
    const bool stats = bn_part != nullptr;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // window column of this lane's first pixel: (ow - 3) - (x0 - 4) = wave * 32 + gq * 16 + px + 2 g4 + 1
    const uint2* const wl = win + wave * 16 * ST7R_G + px + 2 * g4 + 1;
    uint4 ring[8][ST7R_G];
    auto fetch = [&](uint4 (&slot)[ST7R_G], int wrow) {    // window row wrow (clamped: rows past the region are never multiplied into a stored row)
        const uint2* p = wl + (wrow < nrows ? wrow : nrows - 1) * ST7R_COLS;
#pragma unroll
        for (int gq = 0; gq < ST7R_G; ++gq) {
            const uint2 lo = p[gq * 16], hi = p[gq * 16 + 1];
            slot[gq] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
#pragma unroll
    for (int u = 0; u < 7; ++u) fetch(ring[u], u);
#pragma unroll 1
    for (int r = 0; r < R; r += 8) {
        if (oh0 + r >= H) break;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int oh = oh0 + r + u;
            uint4 nxt[ST7R_G];
            fetch(nxt, r + u + 7);                         // input row oh + 4 = window row (oh - oh0) + 7
            if (oh < H && r + u < R) {
                bf16_t* const yrow = y + ((int64_t)n * H + oh) * W * y_ld;
#pragma unroll
                for (int gq = 0; gq < ST7R_G; ++gq) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kh = 0; kh < 7; ++kh)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[kh], __builtin_bit_cast(bf16x8_t, ring[(u + kh) & 7][gq]), acc, 0, 0, 0);
                    if (affine) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v = fmaf(acc[q], sc4[q], bi4[q]);
                            acc[q] = relu ? fmaxf(v, 0.f) : v;
                        }
                    }
                    const int ow = ow0 + gq * 16 + px, c0 = 4 * g4;
                    if (ow < W && c0 < Co) {
                        bf16_t* dst = yrow + ow * y_ld + c0;
                        if (c0 + 4 <= Co) {
                            const uint2 o = make_uint2(pk_bf16(acc[0], acc[1]), pk_bf16(acc[2], acc[3]));
                            *reinterpret_cast<uint2*>(dst) = o;
                            if (stats) {
                                const float a0 = __uint_as_float(o.x << 16), a1 = __uint_as_float(o.x & 0xffff0000u);
                                const float a2 = __uint_as_float(o.y << 16), a3 = __uint_as_float(o.y & 0xffff0000u);
                                s0[0] += a0; s0[1] += a1; s0[2] += a2; s0[3] += a3;
                                s1[0] = fmaf(a0, a0, s1[0]); s1[1] = fmaf(a1, a1, s1[1]); s1[2] = fmaf(a2, a2, s1[2]); s1[3] = fmaf(a3, a3, s1[3]);
                            }
                        } else {
                            for (int q = 0; q < 4 && c0 + q < Co; ++q) dst[q] = f2bf(acc[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int gq = 0; gq < ST7R_G; ++gq) ring[(u + 7) & 7][gq] = nxt[gq];      // row oh + 4 replaces row oh - 4
        }
    }
    if (stats) {
        __syncthreads();                                   // the window is dead: its head is the reduction scratch
        bn_stats_flush_c16<256>(s0, s1, reinterpret_cast<float*>(win), bn_part, bn_slots, y_ld, Co, blockIdx.x, tid);
    }
}

// bf16 output, 7x7 / stride 1|2 / pad 3, Ci <= 3, Co a multiple of 4; more than 64 output channels (Hourglass: 128) run as
// 64-channel chunks over the same image (the 3-channel fp32 input is small next to the output).
bool stem7_fwd_launch(const float* x, const float* w, const float* scale, const float* bias, int relu, void* y, int N, int Ci, int H, int W, int Co,
                      int stride, int OH, int OW, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_WGRAD_C16") != nullptr;
    if (disabled || Ci > 3 || (Co & 3) || (stride != 1 && stride != 2)) return false;
    const int tiles_h = cdiv(OH, C16_TH), tiles_w = cdiv(OW, C16_TW);
    const int64_t ntiles = (int64_t)N * tiles_h * tiles_w;
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const int iters = (int)((ntiles + blocks * 4 - 1) / (blocks * 4));
    constexpr int CHUNK = 16 * ST7_MAXCB;
    if (bn_part && Co <= 16 && (Co & 3) == 0) mark_taken(bn_taken); else bn_part = nullptr;     // statistics hook: one channel block only
    static const bool no_rows = getenv("CN_DISABLE_STEM_ROWS") != nullptr;
    if (!no_rows && stride == 1 && Co <= 16 && OH == H && OW == W && (W & 3) == 0 && (((uintptr_t)x) & 15) == 0) {      // row-walking kernel (DLA base_layer)
        const int R = OH >= 256 ? 32 : 16;
        const int sblocks = cdiv(OW, 128), rblocks = cdiv(OH, R);
        const int64_t nb = (int64_t)8 * ((N + 7) / 8) * sblocks * rblocks;
        if (nb <= 0x7fffffff) {
            hipLaunchKernelGGL(stem7_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, w, (bf16_t*)y, N, Ci, H, W, Co, Co, R, sblocks, rblocks,
                               scale, bias, relu, bn_part, bn_slots);
            return true;
        }
    }
    for (int c0 = 0; c0 < Co; c0 += CHUNK) {
        const int cc = Co - c0 < CHUNK ? Co - c0 : CHUNK;
        const float* wc = w + (int64_t)c0 * Ci * 49;
        const float* sc = scale ? scale + c0 : nullptr;
        const float* bc = bias ? bias + c0 : nullptr;
        bf16_t* yc = (bf16_t*)y + c0;
        if (stride == 1)
            hipLaunchKernelGGL(stem7_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, wc, yc, N, Ci, H, W, cc, Co, OH, OW, tiles_h, tiles_w, iters, sc, bc, relu, bn_part, bn_slots);
        else
            hipLaunchKernelGGL(stem7_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, x, wc, yc, N, Ci, H, W, cc, Co, OH, OW, tiles_h, tiles_w, iters, sc, bc, relu, bn_part, bn_slots);
    }
    return true;
}
