// 3x3 convolutions with 16 INPUT channels at full resolution (DLA-34 level0: 16 -> 16 @512^2 and its data gradient; level1: 16 -> 32,
// stride 2, 512^2 -> 256^2; pose_dla_dcn.py:283-296).  These layers are HBM streams (2.5 GFLOP per MB): what counts is that every input
// byte crosses the memory pipeline as few times as possible.
//   * MFMA shape v_mfma_f32_16x16x32_bf16: M = 16 output channels, N = 16 output pixels, K = 32 = TWO taps x 16 input channels, both
//     operands straight from registers filled by global loads (no LDS).  B operand lane (px = lane & 15, kc = lane >> 4): 16 bytes =
//     channels 8 (kc & 1) .. +7 of the pixel tap (kc >> 1) of the pair needs — a 16-pixel group is one contiguous run.
//   * tap pairing chosen for ROW REUSE: per input row ih and pixel group a lane holds
//         La(ih) = pixel (ih, S ow - 1 + t)          -> taps (kh, 0), (kh, 1) of whichever output row has ih = S oh + kh - 1
//         Lb(ih) = pixel (ih + t, S ow + 1)          -> taps (0, 2), (1, 2) of the output row with S oh - 1 = ih;  t = 0 half alone: tap (2, 2)
//     (t = kc >> 1).  An output row is five MFMAs per 16 output channels: La(S oh - 1), La(S oh), La(S oh + 1), Lb(S oh - 1) and
//     Lb(S oh + 1) with a weight operand whose t = 1 half is zero.  A wave walks DOWN a 32-pixel-wide column of output rows with the
//     rows in a register ring, so every input row is loaded once per wave (two 16-byte loads per lane and group, against five per
//     OUTPUT row in the strip kernel this replaces) and the loads of the next row step are in flight while this one multiplies.
//   * images are dealt to the XCDs (blockIdx % 8 -> image % 8): the halo rows that vertically adjacent workgroups share are fetched
//     into ONE L2.
//   * AFF: the input is the RAW output of the previous convolution and the training-mode BatchNorm (+ ReLU) of that layer is applied
//     on the way into the MFMA operand — x' = bf16(max(fma(x, scale[c], shift[c]), 0)), bit-identical to what bn_fwd_apply_sink_kernel
//     would have stored — so the normalised activation is never written or read (cn_hooks.pre_ss).  Zero padding is applied
//     AFTER the affine map.  The transform runs when a ring slot is first used, not where its loads are issued.
// The zero half of the fifth weight operand multiplies real data (the row below): a non-finite input there would leak as NaN into a
// row it does not belong to — not a concern for activations that are finite, which everything downstream needs anyway.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float c16_f32x4;
typedef __attribute__((ext_vector_type(2))) float c16_f32x2;
typedef __attribute__((ext_vector_type(2))) short c16_s16x2;

#define CR_G 2                                   // 16-pixel groups per wave (a wave's column is 32 output pixels wide)

struct C16Slot { uint4 a[CR_G], b[CR_G]; };

// AFF: 0 = plain input, 1 = affine, 2 = affine + ReLU
template <int AFF>
__device__ static inline uint4 c16_fix(uint4 v, const c16_f32x2 (&sc)[4], const c16_f32x2 (&sh)[4], bool ok) {
    if constexpr (AFF == 0) return v;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const c16_f32x2 x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
        const c16_f32x2 y = __builtin_elementwise_fma(x, sc[i], sh[i]);
        uint32_t p = pk_bf16(y[0], y[1]);
        if constexpr (AFF == 2)                  // ReLU on the rounded pair: negative bf16 = negative int16
            p = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(c16_s16x2, p), c16_s16x2{0, 0}));
        o[i] = ok ? p : 0u;
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

template <int S, int NCB, int AFF>
__global__ __launch_bounds__(256, S == 2 ? 2 : 3) void conv3x3_c16r_kernel(const ConvGeom g, int R, int sblocks, int rblocks) {
    CN_MAIN_PRIO_SET();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kc = lane >> 4;
    const int half = kc & 1, tsel = kc >> 1;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int bpi = sblocks * rblocks;
    const int n = (jb / bpi) * 8 + xcd;
    if (n >= g.N) return;                        // whole workgroup
    const int within = jb % bpi, rb = within / sblocks, sb = within % sblocks;
    const int oh0 = rb * R, ow0 = (sb * 4 + wave) * 16 * CR_G;
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);
    bf16_t* __restrict__ Y = reinterpret_cast<bf16_t*>(g.y);

    // weight tap index of window position (dh, dw) (the geometry's tap table may be mirrored: data gradients)
    auto wt_of = [&](int dh, int dw) {
        int r = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (g.dh[0][t] == dh && g.dw[0][t] == dw) r = g.wt[0][t];
        return r;
    };
    bf16x8_t wa[NCB][5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        int wt;
        bool live = true;
        if (m < 3) wt = tsel ? wt_of(m - 1, 0) : wt_of(m - 1, -1);
        else if (m == 3) wt = tsel ? wt_of(0, 1) : wt_of(-1, 1);
        else { wt = wt_of(1, 1); live = tsel == 0; }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            wa[cb][m] = __builtin_bit_cast(bf16x8_t, ldg16_masked(Wp, ((int64_t)(cb * 16 + px) * g.ktot + wt * 16 + half * 8) * 2,
                                                                live && cb * 16 + px < g.co_pad));
    }
    float bias4[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[cb][r] = (g.bias && cb * 16 + 4 * kc + r < g.Co) ? g.bias[cb * 16 + 4 * kc + r] : 0.f;
    c16_f32x2 sc[4], sh[4];
    if constexpr (AFF != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = c16_f32x2{g.pre_ss[half * 8 + 2 * i], g.pre_ss[half * 8 + 2 * i + 1]};
            sh[i] = c16_f32x2{g.pre_ss[16 + half * 8 + 2 * i], g.pre_ss[16 + half * 8 + 2 * i + 1]};
        }
    }

    // per-lane column offsets (bytes inside an input row) of the two loads of each group
    const int rowpitch = g.W * g.x_ld * 2;
    int colA[CR_G], colB[CR_G];
    bool okA[CR_G], okB[CR_G];
#pragma unroll
    for (int gq = 0; gq < CR_G; ++gq) {
        const int ow = ow0 + gq * 16 + px;
        const int iwA = S * ow - 1 + tsel, iwB = S * ow + 1;
        okA[gq] = (unsigned)iwA < (unsigned)g.W;
        okB[gq] = (unsigned)iwB < (unsigned)g.W;
        colA[gq] = (iwA * g.x_ld + half * 8) * 2;
        colB[gq] = (iwB * g.x_ld + half * 8) * 2 + (tsel ? rowpitch : 0);
    }
    const int64_t img = (int64_t)n * g.H * rowpitch;
    const char* const Xb = reinterpret_cast<const char*>(X) + img;
    // raw loads of input row ih: a = La(ih) (with_b: and b = Lb(ih)).  FAST (an INTERIOR wave, see below): every row and column it touches
    // is inside the image — a scalar row base plus this lane's 32-bit column offset, no masks, no address selects
    auto load_row = [&](auto fast, C16Slot& s, int ih, bool with_b) {
        if constexpr (decltype(fast)::value) {
            const char* const rp = Xb + (int64_t)ih * rowpitch;            // uniform
#pragma unroll
            for (int gq = 0; gq < CR_G; ++gq) {
                s.a[gq] = ldg16(rp + (unsigned)colA[gq]);
                if (with_b) s.b[gq] = ldg16(rp + (unsigned)colB[gq]);
            }
            return;
        }
        const bool rv0 = (unsigned)ih < (unsigned)g.H, rv1 = (unsigned)(ih + 1) < (unsigned)g.H;
        const bool rvb = tsel ? rv1 : rv0;
        const int64_t rbase = img + (int64_t)ih * rowpitch;
#pragma unroll
        for (int gq = 0; gq < CR_G; ++gq) {
            if constexpr (AFF == 0) {
                s.a[gq] = ldg16_masked(X, rbase + colA[gq], rv0 && okA[gq]);
                if (with_b) s.b[gq] = ldg16_masked(X, rbase + colB[gq], rvb && okB[gq]);
            } else {                                       // masked by fix_row, after the affine map
                s.a[gq] = ldg16(reinterpret_cast<const char*>(X) + ((rv0 && okA[gq]) ? rbase + colA[gq] : (int64_t)0));
                if (with_b) s.b[gq] = ldg16(reinterpret_cast<const char*>(X) + ((rvb && okB[gq]) ? rbase + colB[gq] : (int64_t)0));
            }
        }
    };
    // the affine map of a slot whose loads were issued a step ago (no-op without AFF)
    auto fix_row = [&](auto fast, C16Slot& s, int ih, bool with_b) {
        if constexpr (AFF != 0) {
            if constexpr (decltype(fast)::value) {
#pragma unroll
                for (int gq = 0; gq < CR_G; ++gq) {
                    s.a[gq] = c16_fix<AFF>(s.a[gq], sc, sh, true);
                    if (with_b) s.b[gq] = c16_fix<AFF>(s.b[gq], sc, sh, true);
                }
                return;
            }
            const bool rv0 = (unsigned)ih < (unsigned)g.H, rv1 = (unsigned)(ih + 1) < (unsigned)g.H;
            const bool rvb = tsel ? rv1 : rv0;
#pragma unroll
            for (int gq = 0; gq < CR_G; ++gq) {
                s.a[gq] = c16_fix<AFF>(s.a[gq], sc, sh, rv0 && okA[gq]);
                if (with_b) s.b[gq] = c16_fix<AFF>(s.b[gq], sc, sh, rvb && okB[gq]);
            }
        }
    };

    // BatchNorm statistics of the stored values (sink protocol of bn.hip): this lane's four channels per block over every pixel it stores
    const bool bnb = NCB == 1 && g.bnb_part != nullptr;   // y is a gradient w.r.t. a BN output: the BN's backward statistics instead (ConvGeom)
    const bool stats = g.bn_part != nullptr || bnb;
    BnbLane bl;
    bnb_lane_init(bl, bnb ? g.bnb_stats : nullptr, g.y_ld, 4 * kc);
    float s0[NCB][4], s1[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[cb][r] = 0.f; s1[cb][r] = 0.f; }

    // one output row from the slots holding input rows S oh - 1 (m), S oh (z: only .a) and S oh + 1 (p)
    // epilogue constants: ReLU as max(v, lo) with a uniform bound, this lane's store offset inside an output row
    const float relu_lo = g.relu == 1 ? 0.f : -INFINITY;
    unsigned yofs[CR_G][NCB];
#pragma unroll
    for (int gq = 0; gq < CR_G; ++gq)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) yofs[gq][cb] = (unsigned)(((ow0 + gq * 16 + px) * g.y_ld + cb * 16 + 4 * kc) * 2);
    auto out_row = [&](auto fast, const C16Slot& m, const C16Slot& z, const C16Slot& p, int oh) {
        constexpr bool FAST = decltype(fast)::value;   // every pixel of the wave inside the row, every channel of the row real: no guards
        if (!FAST && oh >= g.OH) return;
        const int64_t yrow = ((int64_t)n * g.OH + oh) * g.OW;
        char* const yrp = reinterpret_cast<char*>(Y) + yrow * g.y_ld * 2;                          // uniform
        const char* const xrp = reinterpret_cast<const char*>(g.bnb_x) + yrow * g.y_ld * 2;       // (bnb only)
#pragma unroll
        for (int gq = 0; gq < CR_G; ++gq) {
            const int ow = ow0 + gq * 16 + px;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                c16_f32x4 acc = {bias4[cb][0], bias4[cb][1], bias4[cb][2], bias4[cb][3]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][0], __builtin_bit_cast(bf16x8_t, m.a[gq]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][1], __builtin_bit_cast(bf16x8_t, z.a[gq]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][2], __builtin_bit_cast(bf16x8_t, p.a[gq]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][3], __builtin_bit_cast(bf16x8_t, m.b[gq]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb][4], __builtin_bit_cast(bf16x8_t, p.b[gq]), acc, 0, 0, 0);
                const int ch = cb * 16 + 4 * kc;
                if (FAST || (ow < g.OW && ch < g.y_ld)) {      // y_ld is a multiple of 4; padding channels are written as zeros
                    // ReLU as a lower bound (0, or -inf when off) that keeps a NaN (fmaxf returns the non-NaN operand: a NaN accumulator
                    // would be stored as -inf / 0 and an isnan screen further down would miss it; round-4 ADVICE)
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[r] < relu_lo ? relu_lo : acc[r];
                    if constexpr (!FAST) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ch + r >= g.Co) v[r] = 0.f;
                    }
                    uint2 o;
                    o.x = pk_bf16(v[0], v[1]); o.y = pk_bf16(v[2], v[3]);
                    *reinterpret_cast<uint2*>(yrp + yofs[gq][cb]) = o;
                    if (bnb) {
                        const uint2 xq = *reinterpret_cast<const uint2*>(xrp + yofs[gq][cb]);
                        bnb_lane_add(bl, o, xq, g.bnb_relu, s0[0], s1[0]);
                    } else if (stats) {
                        const float a0 = __uint_as_float(o.x << 16), a1 = __uint_as_float(o.x & 0xffff0000u);
                        const float a2 = __uint_as_float(o.y << 16), a3 = __uint_as_float(o.y & 0xffff0000u);
                        s0[cb][0] += a0; s0[cb][1] += a1; s0[cb][2] += a2; s0[cb][3] += a3;
                        s1[cb][0] = fmaf(a0, a0, s1[cb][0]); s1[cb][1] = fmaf(a1, a1, s1[cb][1]);
                        s1[cb][2] = fmaf(a2, a2, s1[cb][2]); s1[cb][3] = fmaf(a3, a3, s1[cb][3]);
                    }
                }
            }
        }
    };

    // INTERIOR waves (wave-uniform: all of the wave's columns and rows, halo included, inside the image; the whole output block inside the
    // map; no channel padding) run the guard-free instantiation of the row loop.  With the branch-free masked form everywhere the kernel
    // issued ~120 VALU instructions per 16-pixel group and row next to 5 MFMAs (40 of them selects): VALU-issue bound at 4.3 TB/s.
    const bool interior = S * ow0 - 1 >= 0 && S * (ow0 + 16 * CR_G - 1) + 1 < g.W && ow0 + 16 * CR_G <= g.OW && g.Co == g.y_ld &&
                          g.y_ld == 16 * NCB && S * oh0 - 1 >= 0 && S * (oh0 + R - 1) + 2 + S < g.H && oh0 + R <= g.OH;   // (+ S: the rows the last prefetch touches)
    auto run = [&](auto fast) {
        if constexpr (S == 1) {
            // ring of four row slots: rows h - 1, h, h + 1 multiply while row h + 2 is in flight
            C16Slot s[4];
            load_row(fast, s[0], oh0 - 1, true); load_row(fast, s[1], oh0, true); load_row(fast, s[2], oh0 + 1, true);
            fix_row(fast, s[0], oh0 - 1, true); fix_row(fast, s[1], oh0, true);
#pragma unroll 1
            for (int r = 0; r < R; r += 4) {
                const int h = oh0 + r;
                if (h >= g.OH) break;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    load_row(fast, s[(u + 3) & 3], h + u + 2, true);
                    fix_row(fast, s[(u + 2) & 3], h + u + 1, true);
                    out_row(fast, s[u], s[(u + 1) & 3], s[(u + 2) & 3], h + u);
                }
            }
        } else {
            // odd input rows 2 k + 1 (both loads) and even rows 2 k (La only) in rings of three, indexed by the local step
            C16Slot od[3], ev[3];
            load_row(fast, od[2], 2 * oh0 - 1, true); load_row(fast, ev[0], 2 * oh0, false); load_row(fast, od[0], 2 * oh0 + 1, true);
            fix_row(fast, od[2], 2 * oh0 - 1, true);
#pragma unroll 1
            for (int r = 0; r < R; r += 3) {
                const int oh = oh0 + r;
                if (oh >= g.OH) break;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    load_row(fast, ev[(u + 1) % 3], 2 * (oh + u) + 2, false);
                    load_row(fast, od[(u + 1) % 3], 2 * (oh + u) + 3, true);
                    fix_row(fast, ev[u], 2 * (oh + u), false);
                    fix_row(fast, od[u], 2 * (oh + u) + 1, true);
                    out_row(fast, od[(u + 2) % 3], ev[u], od[u], oh + u);
                }
            }
        }
    };
    if (ow0 < g.OW) {
        if (interior) run(std::true_type{}); else run(std::false_type{});
    }
    if (stats) {
        __shared__ float red[4 * 32];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
            bn_stats_flush_c16<256>(s0[cb], s1[cb], red, (bnb ? g.bnb_part : g.bn_part) + cb * 16, bnb ? g.bnb_slots : g.bn_slots, g.y_ld,
                                    g.Co - cb * 16, blockIdx.x, threadIdx.x);
    }
}

// caller: 3x3 / pad 1 geometry in class 0 of g, stride S in {1, 2} (S == 2: the nine taps relative to input pixel (2 oh, 2 ow))
bool conv_c16r_launch(const ConvGeom& g, int dtype, int S, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV_C16R") != nullptr;
    if (disabled || dtype != CN_BF16 || g.Ci != 16 || (g.x_ld & 7) || g.res || g.res32 || g.y_f32 || g.nsrc != 0 || g.dcn_x || g.head_nc ||
        (g.y_ld & 3) || g.ntaps[0] != 9 || g.relu > 1)
        return false;
    for (int t = 0; t < 9; ++t)
        if (g.dh[0][t] < -1 || g.dh[0][t] > 1 || g.dw[0][t] < -1 || g.dw[0][t] > 1) return false;
    if (S == 1 && !(g.Co <= 16 && g.y_ld <= 16 && g.co_pad >= 16 && g.OH == g.H && g.OW == g.W)) return false;
    if (S == 2 && !(g.Co > 16 && g.Co <= 32 && g.y_ld <= 32 && g.co_pad >= 32 && g.sm == 2 && g.so == 1)) return false;
    if (g.pre_ss && g.pre_relu != 0 && g.pre_relu != 1) return false;
    if ((int64_t)g.W * g.x_ld * 2 * g.H >= (int64_t)1 << 31 || (int64_t)g.OW * g.y_ld * 2 >= (int64_t)1 << 31) return false;
    int R = S == 1 ? (g.OH >= 256 ? 32 : 16) : 12;
    const int sblocks = (g.OW + 64 * CR_G - 1) / (64 * CR_G), rblocks = (g.OH + R - 1) / R;
    const int64_t blocks = (int64_t)8 * ((g.N + 7) / 8) * sblocks * rblocks;
    if (blocks > 0x7fffffff) return false;
    if (g.bn_part) {
        if (g.bn_slots > 0) mark_taken(g.bn_taken); else const_cast<ConvGeom&>(g).bn_part = nullptr;
    }
    if (g.bnb_part) {
        if (S == 1 && g.bnb_slots > 0 && !g.bn_part) mark_taken(g.bnb_taken); else const_cast<ConvGeom&>(g).bnb_part = nullptr;
    }
    const int aff = g.pre_ss ? 1 + g.pre_relu : 0;
#define CR_GO(S_, NCB_, AFF_) hipLaunchKernelGGL((conv3x3_c16r_kernel<S_, NCB_, AFF_>), dim3((unsigned)blocks), dim3(256), 0, st, g, R, sblocks, rblocks)
    if (S == 1) { if (aff == 2) CR_GO(1, 1, 2); else if (aff == 1) CR_GO(1, 1, 1); else CR_GO(1, 1, 0); }
    else { if (aff == 2) CR_GO(2, 2, 2); else if (aff == 1) CR_GO(2, 2, 1); else CR_GO(2, 2, 0); }
#undef CR_GO
    return true;
}
