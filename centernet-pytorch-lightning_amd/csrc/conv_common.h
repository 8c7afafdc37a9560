// Pieces shared by the convolution kernels: geometry block, MFMA traits, LDS vector store, fused epilogue.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define CN_MAX_TAPS 16
#define CN_MAX_CLS 4

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
};

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
__device__ static inline void conv_epilogue(const ConvGeom& g, f32x16_t (&acc)[NJ][MI], const int64_t (&pix)[MI], int ch0, int lane) {
    T* __restrict__ Y = reinterpret_cast<T*>(g.y);
    const T* __restrict__ R = reinterpret_cast<const T*>(g.res);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (pix[i] < 0) continue;
        const int64_t px = pix[i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = ch0 + j * 32 + 8 * q + 4 * (lane >> 5);
                if (ch >= g.Co) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][i][q * 4 + e];
                const bool full = (ch + 4 <= g.Co) && ((g.y_ld & 3) == 0) && (g.res == nullptr || (g.res_ld & 3) == 0);
                if (g.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.Co) v[e] += g.bias[ch + e];
                }
                if (R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.Co) v[e] += Elem<T>::ld(R + px * g.res_ld + ch + e);
                }
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (sizeof(T) == 2 && g.y_f32) {
                    float* dstf = reinterpret_cast<float*>(g.y) + px * g.y_ld + ch;
                    if (full) {
                        *reinterpret_cast<float4*>(dstf) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ch + e < g.Co) dstf[e] = v[e];
                    }
                    continue;
                }
                T* dst = Y + px * g.y_ld + ch;
                if (full) {
                    if constexpr (sizeof(T) == 2) {
                        uint2 o;
                        o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                        o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                        *reinterpret_cast<uint2*>(dst) = o;
                    } else {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ch + e < g.Co) Elem<T>::st(dst + e, v[e]);
                }
            }
        }
    }
}

// 3x3 / stride 1 / pad 1 halo-tile kernel (conv3x3.hip); returns false when the shape is not handled there
bool conv3x3s1_launch(const ConvGeom& g, int dtype, hipStream_t st);
