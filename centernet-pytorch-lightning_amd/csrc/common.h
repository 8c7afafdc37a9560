// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/centernet_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits

void cn_set_error(const char* fmt, ...);
// target workgroup count of the split-K weight-gradient kernels of ONE call (cn_hooks.wgrad_blocks; <= 0: the library default)
#define CN_WGRAD_DEFAULT_BLOCKS 1536
// (clamped to 1 .. 65536: launch_wgrad scales the target by 4 / 3 in int arithmetic)
static inline int cn_wgrad_target(const cn_hooks* h) {
    const int b = (h && h->wgrad_blocks > 0) ? h->wgrad_blocks : CN_WGRAD_DEFAULT_BLOCKS;
    return b > 65536 ? 65536 : b;
}

#define CN_CHECK_ARG(cond, ...)                                                                     \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            cn_set_error(__VA_ARGS__);                                                              \
            return CN_EINVAL;                                                                       \
        }                                                                                           \
    } while (0)

#define CN_UNSUPPORTED(...)                                                                         \
    do {                                                                                            \
        cn_set_error(__VA_ARGS__);                                                                  \
        return CN_EUNSUPPORTED;                                                                     \
    } while (0)

#define CN_LAUNCH_CHECK(name)                                                                       \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) {                                                                     \
            cn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));                     \
            return (int)e_;                                                                         \
        }                                                                                           \
    } while (0)

#define CN_HIP(call)                                                                                \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cn_set_error("%s failed: %s", #call, hipGetErrorString(e_));                            \
            return (int)e_;                                                                         \
        }                                                                                           \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 (round-to-nearest-even) -------------------------------------------------------
__host__ __device__ static inline float bf2f(bf16_t b) {
    union { uint32_t u; float f; } v;
    v.u = ((uint32_t)b) << 16;
    return v.f;
}
__host__ __device__ static inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(bf16_t, (__bf16)f);      // v_cvt_pk_bf16_f32: round-to-nearest-even in hardware
#endif
    union { uint32_t u; float f; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair (low half = a) in one v_cvt_pk_bf16_f32
__device__ static inline uint32_t pk_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
}

// statistics sink of the training-mode BatchNorm (bn.hip: cn_hooks.bn_part): part[slots][2][C] fp32, all-zero when handed over
#define BN_STAT_SLOTS 128
struct BnSink { float* part; int slots; int C; };
// The per-call hooks of include/centernet_hip.h (cn_hooks), unpacked.  Nothing is remembered between calls: an entry point reads the
// struct it is handed and a launch function that honours a sink reports it through the `taken` pointer of its geometry.
static inline BnSink hooks_sink(cn_hooks* h) {
    if (h) h->bn_taken = 0;
    const bool ok = h && h->bn_part && h->bn_slots > 0 && h->bn_slots <= 1024 && h->bn_C > 0 && h->bn_C % 8 == 0 && ((uintptr_t)h->bn_part & 15) == 0;
    return ok ? BnSink{h->bn_part, h->bn_slots, h->bn_C} : BnSink{nullptr, 0, 0};
}
static inline void mark_taken(int* p) { if (p) *p = 1; }
struct BnbArm { float* part; int slots; int C; const void* x; const float* stats; int relu; };
static inline BnbArm hooks_bnb(cn_hooks* h) {
    if (h) h->bnb_taken = 0;
    const bool ok = h && h->bnb_part && h->bnb_slots > 0 && h->bnb_slots <= 1024 && h->bnb_C > 0 && h->bnb_x && h->bnb_stats && ((uintptr_t)h->bnb_x & 15) == 0;
    return ok ? BnbArm{h->bnb_part, h->bnb_slots, h->bnb_C, h->bnb_x, h->bnb_stats, h->bnb_relu} : BnbArm{nullptr, 0, 0, nullptr, nullptr, 0};
}
struct PreAffine { const float* ss; int C; int relu; };
static inline PreAffine hooks_pre(const cn_hooks* h) { return (h && h->pre_ss) ? PreAffine{h->pre_ss, h->pre_C, h->pre_relu} : PreAffine{nullptr, 0, 0}; }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kDtype = CN_F32;
    static constexpr int kVec = 4;  // elements per 16-byte vector
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kDtype = CN_BF16;
    static constexpr int kVec = 8;
    __device__ static inline float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static inline void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector <-> float[kVec]
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static inline void load(const float* p, float* out) {
        float4 v = *reinterpret_cast<const float4*>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    }
    __device__ static inline void store(float* p, const float* in) {
        *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
    }
    __device__ static inline void unpack(const uint4& v, float* out) {
        out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y); out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ static inline void load(const bf16_t* p, float* out) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = __uint_as_float(w[i] << 16);
            out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static inline void unpack(const uint4& v, float* out) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = __uint_as_float(w[i] << 16);
            out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static inline void store(bf16_t* p, const float* in) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pk_bf16(in[2 * i], in[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// 16-byte load / store through a NATIVE vector type.  `uint4` is a struct (HIP_vector_type): copying one between a global / LDS
// pointer and an element of a private ARRAY is emitted as llvm.memcpy, which SROA does not promote — the array stays in scratch
// memory and every reload waits with vmcnt(0) behind all in-flight prefetches (found in the DCN kernels' weight rings).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ static inline uint4 ldg16(const void* p) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ static inline void st16(void* p, const uint4& v) { *reinterpret_cast<u32x4_t*>(p) = u32x4_t{v.x, v.y, v.z, v.w}; }

// Branch-free guarded 16-byte global load.  A predicated load compiles to an exec-masked branch, after which the
// compiler can no longer count the loads in flight and falls back to s_waitcnt vmcnt(0) — that serialises every software
// prefetch behind it.  Here the caller passes an address that is readable either way and the value is masked instead.
__device__ static inline uint4 ldg16_masked(const void* base, int64_t byte_ofs, bool ok) {
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + (ok ? byte_ofs : (int64_t)0));
    const uint32_t m = ok ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

// ---- wave / block reductions (wave = 64 lanes) -----------------------------------------------------
__device__ static inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ static inline unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave priority of the launch-stream (critical chain) kernels: the weight-gradient kernels of the side stream stay at 0, so on a
// CU that hosts workgroups of both streams the chain's waves win the instruction arbitration (measurement: DESIGN 6b)
#ifndef CN_MAIN_PRIO
#define CN_MAIN_PRIO 2
#endif
#define CN_MAIN_PRIO_SET() do { if (CN_MAIN_PRIO) __builtin_amdgcn_s_setprio(CN_MAIN_PRIO); } while (0)

// ---- BatchNorm statistics in a producer's epilogue (sink protocol: bn.hip, cn_hooks.bn_part) ----
// a thread adds the 8 bf16 values it is about to store (packed pairs w) to its running (sum, sum of squares)
__device__ static inline void bn_stat_add(float (&s0)[8], float (&s1)[8], const uint32_t (&w)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = __uint_as_float(w[i] << 16), b = __uint_as_float(w[i] & 0xffff0000u);
        s0[2 * i] += a; s1[2 * i] = fmaf(a, a, s1[2 * i]);
        s0[2 * i + 1] += b; s1[2 * i + 1] = fmaf(b, b, s1[2 * i + 1]);
    }
}
// Workgroup flush: every thread holds (s0, s1) of ONE 8-channel vector, threads with equal tid % CPR hold the same vector (CPR = a
// power of two <= 64 = channel vectors per row of the workgroup's tile, first channel ch0).  Wave butterfly over the lanes that
// share a vector, the NT / 64 waves meet in `lds` (>= (NT / 64) * CPR * 16 floats; barriers inside), then one fp32 atomic per
// (channel, statistic) into row `slot % slots` of part[slots][2][C].  Channels >= ch_lim are dropped.
template <int CPR, int NT>
__device__ static inline void bn_stats_flush(float (&s0)[8], float (&s1)[8], float* lds, float* part, int slots, int C, int ch0, int ch_lim,
                                             unsigned slot, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o >= CPR; o >>= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] += __shfl_xor(s0[e], o, 64); s1[e] += __shfl_xor(s1[e], o, 64); }
    }
    __syncthreads();
    if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { lds[(wave * CPR + lane) * 16 + e] = s0[e]; lds[(wave * CPR + lane) * 16 + 8 + e] = s1[e]; }
    }
    __syncthreads();
    if (tid < CPR * 16) {
        const int cv = tid >> 4, e = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) v += lds[(w * CPR + cv) * 16 + e];
        const int ch = ch0 + cv * 8 + (e & 7);
        if (ch < ch_lim) atomicAdd(part + ((int64_t)(slot % (unsigned)slots) * 2 + (e >> 3)) * C + ch, v);
    }
}

// Flush for the 16-channel direct kernels (conv3x3_c16_kernel, stem7_fwd_kernel): lane = (pixel px = lane & 15, channel quad kc =
// lane >> 4), a lane holds (sum, sum of squares) of channels 4 kc .. 4 kc + 3.  `red`: >= (NT / 64) * 32 floats of LDS.
template <int NT>
__device__ static inline void bn_stats_flush_c16(float (&s0)[4], float (&s1)[4], float* red, float* part, int slots, int C, int ch_lim,
                                                 unsigned slot, int tid) {
    const int lane = tid & 63, wave = tid >> 6, kc = lane >> 4;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[r] += __shfl_xor(s0[r], o, 64); s1[r] += __shfl_xor(s1[r], o, 64); }
    }
    __syncthreads();
    if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[wave * 32 + 4 * kc + r] = s0[r]; red[wave * 32 + 16 + 4 * kc + r] = s1[r]; }
    }
    __syncthreads();
    if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) v += red[w * 32 + tid];
        const int ch = tid & 15;
        if (ch < ch_lim) atomicAdd(part + ((int64_t)(slot % (unsigned)slots) * 2 + (tid >> 4)) * C + ch, v);
    }
}

// dispatch on dtype
#define CN_DISPATCH_DTYPE(dtype, T, ...)                                                            \
    do {                                                                                            \
        if ((dtype) == CN_F32) {                                                                    \
            typedef float T;                                                                        \
            __VA_ARGS__;                                                                            \
        } else if ((dtype) == CN_BF16) {                                                            \
            typedef bf16_t T;                                                                       \
            __VA_ARGS__;                                                                            \
        } else {                                                                                    \
            cn_set_error("bad dtype %d", (int)(dtype));                                             \
            return CN_EINVAL;                                                                       \
        }                                                                                           \
    } while (0)
