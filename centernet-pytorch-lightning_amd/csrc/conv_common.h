// Pieces shared by the convolution kernels: geometry block, MFMA traits, LDS vector store, fused epilogue.
#pragma once
#include "dcn_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define CN_MAX_TAPS 16
#define CN_MAX_CLS 4
#define CN_MAX_SRC 6

struct ConvGeom {
    const void* x;
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int N, H, W, Ci, x_ld;
    int OH, OW, Co, y_ld, res_ld;
    int ktot;   // weight row length = total taps * Ci
    int co_pad; // weight rows available
    int relu;
    int y_f32;  // store the output as fp32 even in bf16 compute mode (DCN offsets / mask logits)
    int sm;     // input index = out_class_index * sm + d[tap]
    int so;     // output index = out_class_index * so + parity
    int ntaps[CN_MAX_CLS];
    signed char dh[CN_MAX_CLS][CN_MAX_TAPS];
    signed char dw[CN_MAX_CLS][CN_MAX_TAPS];
    unsigned char wt[CN_MAX_CLS][CN_MAX_TAPS];
    // fused DCNv2 offset/mask-gradient epilogue (cn_dcn_bwd_dom): the GEMM result dcol[p][k*Ci+c] is never stored
    const void* dcn_x;
    const float* dcn_om;
    float* dcn_dom;
    float* dcn_far;
    int* far_flag;        // nullable: set by the dom kernels when a far sample was scattered; read (and dx_far restored to 0) by the dx kernel
    int dcn_Ci, dcn_H, dcn_W, dcn_xld, dcn_omld;
    const float* res32;   // optional fp32 residual added in the epilogue (pitch res32_ld)
    int res32_ld;
    int epi_tile;         // bf16 output rows are 16-byte aligned vectors: use the LDS-staged epilogue
    // 1x1 conv over the channel CONCATENATION of nsrc tensors of the same N,H,W (DLA Root, pose_dla_dcn.py:180-188): the K loop
    // walks the sources one after the other, so torch.cat(x, 1) is never materialised.  nsrc == 0: the single input `x`.
    int nsrc;
    // BatchNorm statistics sink (training-mode conv + BN, bn.hip: cn_hooks.bn_part): every workgroup adds sum / sum of squares of the
    // values it stores to part[workgroup % bn_slots][2][y_ld]; only the LDS-staged bf16 epilogue has the hook
    float* bn_part;
    int bn_slots;
    int* bn_taken;            // host pointer (cn_hooks.bn_taken), set to 1 by the launch function whose kernel fills bn_part; nullable
    int* bnb_taken;           // the same for bnb_part
    const void* xs[CN_MAX_SRC];
    int xs_c[CN_MAX_SRC];     // channels (= pixel pitch) of each source, multiples of the K slice
    int xs_k0[CN_MAX_SRC];    // first K index of each source
    // fused task head (cn_head2_fwd; conv3x3_ws_kernel<..., HEAD = 2>): y is the public fp32 NCHW map [N, 2, H, W], all-zero at launch;
    // every wave adds its 32 hidden channels' share of conv1x1(relu(conv3x3(x) + bias)) with fp32 atomics, head_w = fp32 [2][Co]
    const float* head_w;
    const float* head_b;
    int head_nc;
    // pre-affine of the INPUT (cn_hooks.pre_ss; conv_c16.hip): x is the raw output of the previous conv, the kernel applies
    // x' = bf16(fma(x, pre_ss[c], pre_ss[Ci + c])) (pre_relu: max(., 0)) — that layer's training-mode BatchNorm — on the way in
    const float* pre_ss;
    int pre_relu;
    // BatchNorm BACKWARD statistics sink (cn_hooks.bnb_part; the 16-channel data-gradient kernels have the hook): y is the gradient
    // w.r.t. the output of a training-mode BN (+ ReLU) whose input was bnb_x (pitch y_ld); every workgroup adds, over the values it
    // STORES, sum g and sum g * xhat (g = relu ? (fma(x, sc, sh) > 0 ? y : 0) : y, xhat = (x - mean) * invstd) to bnb_part[slots][2][y_ld]
    float* bnb_part;
    int bnb_slots;
    const void* bnb_x;
    const float* bnb_stats;   // fp32 [4][y_ld]: mean | invstd | scale | shift
    int bnb_relu;
};

// the hook for kernels whose lanes hold 4 consecutive channels of one pixel (conv3x3_c16r_kernel, dgrad_s2_c32to16_kernel)
struct BnbLane { float mu[4], is[4], sc[4], sh[4]; };
__device__ static inline void bnb_lane_init(BnbLane& b, const float* __restrict__ stats, int C, int ch) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool in = stats != nullptr && ch + r < C;
        b.mu[r] = in ? stats[ch + r] : 0.f; b.is[r] = in ? stats[C + ch + r] : 0.f;
        b.sc[r] = in ? stats[2 * C + ch + r] : 0.f; b.sh[r] = in ? stats[3 * C + ch + r] : 0.f;
    }
}
// dy, x: the 4 bf16 this lane stores / the BN input at the same position (packed pairs); bn_partial_kernel<T, 1>'s arithmetic
__device__ static inline void bnb_lane_add(const BnbLane& b, uint2 dy, uint2 x, int relu, float (&s0)[4], float (&s1)[4]) {
    const float gv[4] = {__uint_as_float(dy.x << 16), __uint_as_float(dy.x & 0xffff0000u), __uint_as_float(dy.y << 16), __uint_as_float(dy.y & 0xffff0000u)};
    const float xv[4] = {__uint_as_float(x.x << 16), __uint_as_float(x.x & 0xffff0000u), __uint_as_float(x.y << 16), __uint_as_float(x.y & 0xffff0000u)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float gg = (!relu || fmaf(xv[r], b.sc[r], b.sh[r]) > 0.f) ? gv[r] : 0.f;
        s0[r] += gg;
        s1[r] = fmaf(gg, (xv[r] - b.mu[r]) * b.is[r], s1[r]);
    }
}


template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KSTEP = 16;  // k per MFMA
    static constexpr int PAD = 8;     // elements of row padding (16 bytes)
    typedef bf16x8_t Frag;
    __device__ static inline Frag load(const bf16_t* tile, int pitch, int row, int kk, int lane) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + (row + (lane & 31)) * pitch + kk * 16 + (lane >> 5) * 8);
        return __builtin_bit_cast(bf16x8_t, v);
    }
    __device__ static inline f32x16_t mma(Frag a, Frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 2;
    static constexpr int PAD = 1;
    typedef float Frag;
    __device__ static inline Frag load(const float* tile, int pitch, int row, int kk, int lane) {
        return tile[(row + (lane & 31)) * pitch + kk * 2 + (lane >> 5)];
    }
    __device__ static inline f32x16_t mma(Frag a, Frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

template <typename T, int PITCH>
__device__ static inline void lds_store_vec(T* tile, int row, int col, uint4 v) {
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(tile + row * PITCH + col) = v;  // PITCH*2 is a multiple of 16
    } else {
        float* p = reinterpret_cast<float*>(tile) + row * PITCH + col;  // odd pitch: scalar stores
        p[0] = __uint_as_float(v.x); p[1] = __uint_as_float(v.y); p[2] = __uint_as_float(v.z); p[3] = __uint_as_float(v.w);
    }
}


// Fused epilogue.  Accumulator layout of one 32x32 MFMA block (weights = first operand): lane holds pixel (lane&31) and
// channels (r&3) + 8*(r>>2) + 4*(lane>>5), i.e. four groups of 4 consecutive channels -> 8/16-byte NHWC stores.
// pix[i] = flat NHWC pixel index of this lane's pixel in M-subtile i, or -1 when it is outside the problem.
template <typename T, int MI, int NJ>
__device__ static inline void conv_epilogue(const ConvGeom& g, f32x16_t (&acc)[NJ][MI], const int64_t (&pix)[MI], int ch0, int lane,
                                            bool use_res32 = true) {
    // channels Co .. y_ld-1 (the activation's zero padding) are WRITTEN as zeros, so callers allocate with empty()
    T* __restrict__ Y = reinterpret_cast<T*>(g.y);
    const T* __restrict__ R = reinterpret_cast<const T*>(g.res);
    const bool mask_mode = g.relu == 2;          // ReLU-backward mask: y = (res > 0) ? y : 0 instead of y += res
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (pix[i] < 0) continue;
        const int64_t px = pix[i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = ch0 + j * 32 + 8 * q + 4 * (lane >> 5);
                if (ch >= g.y_ld) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][i][q * 4 + e];
                const bool full = (ch + 4 <= g.y_ld) && ((g.y_ld & 3) == 0) && (g.res == nullptr || (g.res_ld & 3) == 0);
                if (g.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.Co) v[e] += g.bias[ch + e];
                }
                if (R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.Co) {
                            const float r = Elem<T>::ld(R + px * g.res_ld + ch + e);
                            v[e] = mask_mode ? (r > 0.f ? v[e] : 0.f) : v[e] + r;
                        }
                }
                if (g.res32 && use_res32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.Co) {
                            float* r = const_cast<float*>(g.res32) + px * g.res32_ld + ch + e;
                            v[e] += *r;
                            if (g.far_flag) *r = 0.f;       // lazy dx_far protocol: consumed once, left clean
                        }
                }
                if (g.relu == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ch + e >= g.Co) v[e] = 0.f;
                if (sizeof(T) == 2 && g.y_f32) {
                    float* dstf = reinterpret_cast<float*>(g.y) + px * g.y_ld + ch;
                    if (full) {
                        *reinterpret_cast<float4*>(dstf) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ch + e < g.y_ld) dstf[e] = v[e];
                    }
                    continue;
                }
                T* dst = Y + px * g.y_ld + ch;
                if (full) {
                    if constexpr (sizeof(T) == 2) {
                        uint2 o;
                        o.x = pk_bf16(v[0], v[1]);
                        o.y = pk_bf16(v[2], v[3]);
                        *reinterpret_cast<uint2*>(dst) = o;
                    } else {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.y_ld) Elem<T>::st(dst + e, v[e]);
                }
            }
        }
    }
}

// LDS-staged epilogue (bf16 outputs).  The direct epilogue above writes 8-byte pieces at a 32-pixel stride per store
// instruction — on the 256-channel head convs that scatter cost 40 % of the kernel.  Here the accumulators pass through
// an fp32 LDS slab (one 32-pixel row block per wave row at a time, pitch BN+4 floats = conflict-free for both sides)
// and leave as 16-byte vectors along the channel axis: bias / residual / ReLU are applied in fp32 on the way out, so the
// rounding is identical to the direct path.  Block tile = (WGM*MI*32) pixels x (WGN*NJ*32) channels, 256 threads.
// `ot` needs WGM*32*(BN+4) floats and must not be read by anyone else (caller barriers before the call).
static inline bool conv_epi_tile_ok(const ConvGeom& g, int dtype) {
    // Co == y_ld: this epilogue only writes channels < Co, the activation's zero padding (Co = 40 -> y_ld = 48) is the direct one's job
    return dtype == CN_BF16 && !g.y_f32 && g.res32 == nullptr && (g.Co & 7) == 0 && g.Co == g.y_ld &&
           (g.res == nullptr || (g.res_ld & 7) == 0) && (((uintptr_t)g.y | (uintptr_t)g.res) & 15) == 0;
}
template <int MI, int NJ, int WGM, int WGN, int NT = 256, typename PixFn>
__device__ static inline void conv_epilogue_tile(const ConvGeom& g, f32x16_t (&acc)[NJ][MI], float* ot, int n0, int tid, PixFn pixel_of) {
    constexpr int BN = WGN * NJ * 32, P = BN + 4, R = WGM * 32, WM = MI * 32;
    constexpr int CPR = BN / 8, PASSES = (R * CPR + NT - 1) / NT;
    const int lane = tid & 63, wave = tid >> 6;
    const int wgm = wave / WGN, wgn = wave % WGN;
    bf16_t* __restrict__ Y = reinterpret_cast<bf16_t*>(g.y);
    const bf16_t* __restrict__ Rr = reinterpret_cast<const bf16_t*>(g.res);
    // BN statistics of the stored values: a thread visits the same channel vector in every pass (NT is a multiple of CPR)
    const bool stats = g.bn_part != nullptr;
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
#pragma unroll
    for (int ii = 0; ii < MI; ++ii) {
        if (ii > 0) __syncthreads();
        float* dst = ot + (wgm * 32 + (lane & 31)) * P + wgn * NJ * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dst + j * 32 + 8 * q) = make_float4(acc[j][ii][q * 4], acc[j][ii][q * 4 + 1], acc[j][ii][q * 4 + 2], acc[j][ii][q * 4 + 3]);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int id = tid + p * NT;
            const int rr = id / CPR, c8 = (id % CPR) * 8;
            if (id >= R * CPR) continue;
            const int ch = n0 + c8;
            if (ch >= g.Co) continue;
            const int64_t px = pixel_of((rr >> 5) * WM + ii * 32 + (rr & 31));
            if (px < 0) continue;
            float v[8];
            const float4 a = *reinterpret_cast<const float4*>(ot + rr * P + c8), b = *reinterpret_cast<const float4*>(ot + rr * P + c8 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if (g.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ch), b1 = *reinterpret_cast<const float4*>(g.bias + ch + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (Rr) {
                float r[8];
                Vec16<bf16_t>::load(Rr + px * g.res_ld + ch, r);
                if (g.relu == 2) {                       // ReLU-backward mask instead of a residual add
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = r[e] > 0.f ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r[e];
                }
            }
            if (g.relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pk_bf16(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<uint4*>(Y + px * g.y_ld + ch) = make_uint4(w[0], w[1], w[2], w[3]);
            if (stats) bn_stat_add(s0, s1, w);
        }
    }
    if (stats) {
        static_assert(NT % CPR == 0 && (CPR & (CPR - 1)) == 0 && CPR <= 64, "thread -> channel-vector map of the statistics");
        bn_stats_flush<CPR, NT>(s0, s1, ot, g.bn_part, g.bn_slots, g.y_ld, n0, g.Co, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * 7u, tid);
    }
}

// DCNv2 backward, source side, fused into the GEMM that produces dcol = dY x W^T (rows = tap*Ci + c):
//   dom[p][2k]   = m * sum_c dcol[p,k,c] * d(bilinear)/d(py)      dom[p][2k+1] likewise for px
//   dom[p][18+k] = m(1-m) * sum_c dcol[p,k,c] * bilinear_unmasked
// plus the rare samples displaced by more than DCN_FAR_R pixels, scattered into dx_far with global atomics (everything
// nearer is handled by the atomic-free adjoint-gather kernel).  Lane layout as in conv_epilogue: pixel = lane&31,
// channels (r&3) + 8*(r>>2) + 4*(lane>>5).  `red` = LDS scratch [128 pixels][TPB taps][3], zeroed by the caller.
#define DCN_FAR_R 3
template <typename T, int MI, int NJ>
__device__ static inline void dcn_dom_accumulate(const ConvGeom& g, f32x16_t (&acc)[NJ][MI], const int64_t (&pix)[MI],
                                                 const int (&mloc)[MI], int n0, int ch0, int lane, float* red, int tpb) {
    const T* __restrict__ X = reinterpret_cast<const T*>(g.dcn_x);
    const int Ci = g.dcn_Ci, H = g.dcn_H, W = g.dcn_W;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (pix[i] < 0) continue;
        const int64_t p = pix[i];
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const int64_t img = p - ((int64_t)h * W + w);
        const float* o = g.dcn_om + p * g.dcn_omld;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int chb = ch0 + j * 32;
            if (chb >= g.Co) continue;
            const int k = chb / Ci, cb = chb - k * Ci;
            const float py = (float)(h - 1 + k / 3) + o[2 * k];
            const float px = (float)(w - 1 + k % 3) + o[2 * k + 1];
            const float m = sigmoidf_(o[18 + k]);
            const Tap t = make_tap(py, px, H, W);
            const int64_t i00 = img + (int64_t)t.h0 * W + t.w0;
            const int dh0 = t.h0 - h, dw0 = t.w0 - w;
            const bool far_h0 = dh0 > DCN_FAR_R || dh0 < -DCN_FAR_R, far_h1 = dh0 + 1 > DCN_FAR_R || dh0 + 1 < -DCN_FAR_R;
            const bool far_w0 = dw0 > DCN_FAR_R || dw0 < -DCN_FAR_R, far_w1 = dw0 + 1 > DCN_FAR_R || dw0 + 1 < -DCN_FAR_R;
            const bool any_far = far_h0 || far_h1 || far_w0 || far_w1;
            float s_m = 0.f, s_y = 0.f, s_x = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cb + 8 * q + 4 * (lane >> 5);
                float x00[4] = {0.f, 0.f, 0.f, 0.f}, x01[4] = {0.f, 0.f, 0.f, 0.f}, x10[4] = {0.f, 0.f, 0.f, 0.f}, x11[4] = {0.f, 0.f, 0.f, 0.f};
                auto ld4 = [&](int64_t pixel, float* out) {
                    const T* src = X + pixel * g.dcn_xld + c;
                    if constexpr (sizeof(T) == 2) {
                        const uint2 v = *reinterpret_cast<const uint2*>(src);
                        out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
                        out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
                    } else {
                        const float4 v = *reinterpret_cast<const float4*>(src);
                        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
                    }
                };
                if (t.ok00) ld4(i00, x00);
                if (t.ok01) ld4(i00 + 1, x01);
                if (t.ok10) ld4(i00 + W, x10);
                if (t.ok11) ld4(i00 + W + 1, x11);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gj = acc[j][i][q * 4 + e];
                    const float val = x00[e] * t.w00 + x01[e] * t.w01 + x10[e] * t.w10 + x11[e] * t.w11;
                    s_m = fmaf(gj, val, s_m);
                    s_y = fmaf(gj, (1.f - t.lw) * (x10[e] - x00[e]) + t.lw * (x11[e] - x01[e]), s_y);
                    s_x = fmaf(gj, (1.f - t.lh) * (x01[e] - x00[e]) + t.lh * (x11[e] - x10[e]), s_x);
                    if (any_far) {
                        if (g.far_flag) *g.far_flag = 1;
                        const float gm = gj * m;
                        float* far = g.dcn_far + c + e;
                        if (t.w00 != 0.f && (far_h0 || far_w0)) atomicAdd(far + i00 * Ci, gm * t.w00);
                        if (t.w01 != 0.f && (far_h0 || far_w1)) atomicAdd(far + (i00 + 1) * Ci, gm * t.w01);
                        if (t.w10 != 0.f && (far_h1 || far_w0)) atomicAdd(far + (i00 + W) * Ci, gm * t.w10);
                        if (t.w11 != 0.f && (far_h1 || far_w1)) atomicAdd(far + (i00 + W + 1) * Ci, gm * t.w11);
                    }
                }
            }
            s_m += __shfl_xor(s_m, 32, 64);
            s_y += __shfl_xor(s_y, 32, 64);
            s_x += __shfl_xor(s_x, 32, 64);
            if (lane < 32) {
                float* r = red + (mloc[i] * tpb + (chb - n0) / Ci) * 3;
                atomicAdd(r + 0, s_y); atomicAdd(r + 1, s_x); atomicAdd(r + 2, s_m);
            }
        }
    }
}

// fused DCNv2 data-gradient kernel (dcn_fused.hip)
void dcn_bwd_dx_launch(const ConvGeom& g, int dtype, hipStream_t st);
void dcn_fwd_launch(const ConvGeom& g, int dtype, hipStream_t st);
bool dcn_fwd_tile_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                         int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st);
bool dcn_fwd_bm_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                       int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st);   // dcn_bm.hip: blend on the matrix cores
bool dcn_fwd_bm_shape_ok(int Ci, int x_ld, int Co, int y_ld, int om_ld);
bool dcn_fwd_b2_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                       int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st);   // dcn_b2.hip: 16x16 tiles, four waves per SIMD
bool dcn_fwd_b2_shape_ok(int Ci, int x_ld, int Co, int y_ld, int om_ld, int H, int W);
bool dcn_dx_bm_shape_ok(int Ci, int dy_ld, int om_ld);
bool dcn_fwd_tile_shape_ok(int Ci, int x_ld, int Co, int y_ld);
bool dcn_wgrad_bm_shape_ok(int Ci, int x_ld, int Co, int dy_ld, int om_ld);
bool dcn_wgrad_bm_launch(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                         int om_ld, int target_blocks, hipStream_t st);                     // dcn_bm.hip
bool dcn_dx_bm_launch(const void* dy, const void* wpd0, const float* om, float* far, int* far_flag, void* dx, int N, int H, int W, int Ci,
                      int dy_ld, int om_ld, hipStream_t st);                                // dcn_bm.hip
bool dcn_dom_bm_shape_ok(int Ci, int dy_ld, int x_ld, int om_ld);
bool dcn_dom_bm_launch(const void* dy, const void* wd2, const void* x, const float* om, float* dom, int dom_slabs, float* far, int* far_flag,
                       int N, int H, int W, int Ci, int dy_ld, int x_ld, int om_ld, hipStream_t st);
bool dcn_bwd_dom_tile_launch(const void* dy, const void* wd2, const void* x, const float* om, float* dom, int dom_slabs, float* far, int* far_flag,
                             int N, int H, int W, int Ci, int Co, int dy_ld, int x_ld, int om_ld, hipStream_t st);

// 3x3 / stride 1 / pad 1 halo-tile kernel (conv3x3.hip); returns false when the shape is not handled there
bool conv3x3s1_launch(const ConvGeom& g, int dtype, hipStream_t st);
bool conv_c16r_launch(const ConvGeom& g, int dtype, int S, hipStream_t st);
bool dgrad3x3s2_launch(const ConvGeom& g, int dtype, hipStream_t st);     // conv_dgrad_s2.hip: data gradient of the stride-2 3x3 convs, all four parity classes per workgroup
bool conv3x3s2_launch(const ConvGeom& g, int dtype, hipStream_t st);   // conv3x3.hip: the same halo-tile skeleton for the stride-2 forward convs
bool conv3x3_ws_launch(const ConvGeom& g, int dtype, hipStream_t st);   // conv3x3_ws.hip: 64 input channels, weights in registers
bool conv1x1_stream_launch(const ConvGeom& g, int dtype, hipStream_t st); // conv1x1_stream.hip: weights in LDS, activations straight from global memory
bool dgrad_s2_c32to16_launch(const ConvGeom& g, hipStream_t st);
