// Streaming 1x1 / stride 1 convolution (bf16): y[p][co] = act(bias[co] + sum_k x[p][k] * W[co][k] [+ res]).
// The DLA 1x1 convs (tree projections, roots, the heads' output convs) and their data gradients are HBM streams: 64-250 FLOP per
// byte, far below the ridge.  The implicit-GEMM kernel walks K in 32-wide slices, each one global -> registers -> LDS -> barrier
// (2-3.5 TB/s on these shapes).  Here nothing of the activation touches the LDS:
//   * the whole weight matrix [Co_pad][K] (<= 64 KB) is loaded into LDS ONCE per workgroup (persistent grid);
//   * a wave owns 32-pixel strips; a pixel's channels are the MFMA B operand straight from global memory — K/16 independent
//     16-byte loads per lane, all in flight at once, the next strip's issued before the current strip's MFMAs;
//   * A fragments (weights) come from the LDS with one conflict-free ds_read_b128 per MFMA; no barrier in the strip loop, the
//     four waves of a workgroup (two workgroups per CU) run independently and hide each other's latency;
//   * epilogue as in conv3x3_ws_kernel: bias is the accumulator's initial value, v_permlane32_swap turns accumulator quads into
//     8-channel vectors, 16-byte stores; residual / ReLU mask / ReLU / fp32 rows are compile-time variants.
#include "conv_common.h"
#include <algorithm>

#define S1_NT 256

// KS = K / 16 (K steps of one 32x32x16 MFMA), NJ = Co_pad / 32
// NCHW: the output is the public fp32 map [N][Co][OH*OW] with the Co real channels (a head's last conv, heads.py:15-17 — the
// reference's layout): every accumulator register is one channel of 32 consecutive pixels, i.e. a 128-byte run of that channel's
// plane, stored straight from the accumulators (no NHWC bf16 intermediate, no layout-change pass).  OH*OW % 32 == 0.
template <int KS, int NJ, bool YF32, int RES, bool RELU, bool NCHW = false>
__global__ __launch_bounds__(S1_NT) void conv1x1_stream_kernel(const ConvGeom g, int64_t npix) {
    CN_MAIN_PRIO_SET();
    constexpr int K = KS * 16, P = K + 8;                       // LDS pitch of a weight row (elements): rows 16 B apart mod 256 B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const Ws = reinterpret_cast<bf16_t*>(smem);         // [NJ * 32][P]
    float* const bias_l = reinterpret_cast<float*>(smem + (size_t)NJ * 32 * P * 2);   // [NJ * 32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);

    for (int v = tid; v < NJ * 32 * (K / 8); v += S1_NT) {
        const int row = v / (K / 8), c8 = (v % (K / 8)) * 8;
        const uint4 w = ldg16(Wp + (int64_t)min(row, g.co_pad - 1) * g.ktot + c8);
        st16(Ws + row * P + c8, row < g.co_pad ? w : make_uint4(0, 0, 0, 0));
    }
    for (int c = tid; c < NJ * 32; c += S1_NT) bias_l[c] = (g.bias && c < g.Co) ? g.bias[c] : 0.f;
    __syncthreads();

    const int64_t nstrips = (npix + 31) / 32, stride = (int64_t)gridDim.x * 4;
    int64_t s = (int64_t)blockIdx.x * 4 + wave;
    bf16x8_t fb[2][KS];
    auto xload = [&](bf16x8_t (&f)[KS], int64_t strip) {
        int64_t p = strip * 32 + (lane & 31);
        p = p < npix ? p : npix - 1;                              // rows past the end re-read the last pixel, never stored
        const bf16_t* row = X + p * g.x_ld + 8 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) f[kk] = __builtin_bit_cast(bf16x8_t, ldg16(row + kk * 16));
    };
    if (s < nstrips) xload(fb[0], s);
    const bf16_t* const wa = Ws + (lane & 31) * P + 8 * (lane >> 5);
#pragma unroll 1
    for (int it = 0; s < nstrips; s += stride, ++it) {
        const bool more = s + stride < nstrips;
        f32x16_t acc[NJ];
        auto compute = [&](const bf16x8_t (&f)[KS], bf16x8_t (&fn)[KS]) {
            if (more) xload(fn, s + stride);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float* bl = bias_l + j * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *reinterpret_cast<const float4*>(bl + 8 * q);
                    acc[j][4 * q] = bq.x; acc[j][4 * q + 1] = bq.y; acc[j][4 * q + 2] = bq.z; acc[j][4 * q + 3] = bq.w;
                }
            }
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(wa + j * 32 * P + kk * 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, f[kk], acc[j], 0, 0, 0);
                }
        };
        if (it & 1) compute(fb[1], fb[0]); else compute(fb[0], fb[1]);

        const int64_t p = s * 32 + (lane & 31);
        const bool live = p < npix;
        if constexpr (NCHW) {
            if (live) {
                const int64_t hw = (int64_t)g.OH * g.OW;
                const int64_t b = p / hw, pix = p - b * hw;
                float* const dst = reinterpret_cast<float*>(g.y) + b * g.Co * hw + pix;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (ch < g.Co) dst[(int64_t)ch * hw] = acc[j][r];
                    }
            }
            continue;
        }
        const int chl = 8 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int ch = j * 32 + 16 * qq + chl;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][(2 * qq) * 4 + e]), __float_as_uint(acc[j][(2 * qq + 1) * 4 + e]), false, false);
                    v[e] = __uint_as_float(sw2[0]);
                    v[4 + e] = __uint_as_float(sw2[1]);
                }
                if (!live || ch >= g.y_ld) continue;
                if constexpr (RES != 0) {
                    float rv[8];
                    Vec16<bf16_t>::load(reinterpret_cast<const bf16_t*>(g.res) + p * g.res_ld + ch, rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = RES == 2 ? (rv[e] > 0.f ? v[e] : 0.f) : v[e] + rv[e];
                }
                if constexpr (RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, INFINITY);
                }
                if constexpr (YF32) {
                    float* dst = reinterpret_cast<float*>(g.y) + p * g.y_ld + ch;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    Vec16<bf16_t>::store(reinterpret_cast<bf16_t*>(g.y) + p * g.y_ld + ch, v);
                }
            }
    }
}

#define CN_MAX_DEVICES 16
template <int KS, int NJ>
static bool s1_go(const ConvGeom& g, int64_t npix, int grid, int dev, hipStream_t st) {
    const size_t smem = (size_t)NJ * 32 * (KS * 16 + 8) * 2 + (size_t)NJ * 32 * 4;
    const int res = g.res == nullptr ? 0 : (g.relu == 2 ? 2 : 1);
    if (g.y_f32 && res != 0) return false;
#define S1_GO(F32_, RES_, RELU_)                                                                                         \
    do {                                                                                                                 \
        auto kfn = conv1x1_stream_kernel<KS, NJ, F32_, RES_, RELU_>;                                                     \
        static bool attr[CN_MAX_DEVICES] = {};       /* function attributes are per device */                            \
        if (!attr[dev]) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[dev] = true; } \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(S1_NT), smem, st, g, npix);                                             \
    } while (0)
    if (g.y_f32) { if (g.relu == 1) S1_GO(true, 0, true); else S1_GO(true, 0, false); }
    else if (res == 0) { if (g.relu == 1) S1_GO(false, 0, true); else S1_GO(false, 0, false); }
    else if (res == 1) { if (g.relu == 1) S1_GO(false, 1, true); else S1_GO(false, 1, false); }
    else S1_GO(false, 2, false);
#undef S1_GO
    return true;
}

// y = public NCHW fp32 map [N][Co][H*W] (g.y; g.y_ld is ignored), bias in, no residual / ReLU.  K = 256 (the heads' hidden width)
bool conv1x1_stream_nchw_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV1X1_STREAM") != nullptr || getenv("CN_DISABLE_CONV1X1_NCHW") != nullptr;
    static int cus_of[CN_MAX_DEVICES] = {};
    if (disabled || dtype != CN_BF16 || g.nsrc != 0 || g.res != nullptr || g.relu != 0 || (g.x_ld & 7)) return false;
    if ((reinterpret_cast<uintptr_t>(g.x) | reinterpret_cast<uintptr_t>(g.w)) & 15) return false;
    const int K = g.Ci, nj = (g.Co + 31) / 32;
    const int64_t hw = (int64_t)g.OH * g.OW, npix = (int64_t)g.N * hw;
    if (K != 256 || K != g.ktot || nj < 1 || nj > 4 || g.co_pad < nj * 32 || hw % 32 != 0 || npix < 64 * 1024) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CN_MAX_DEVICES) return false;
    if (cus_of[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) return false;
        cus_of[dev] = v;
    }
    const int grid = (int)std::min<int64_t>(2 * (int64_t)cus_of[dev], (npix / 32 + 3) / 4);
#define S1_NCHW(NJ_)                                                                                                     \
    do {                                                                                                                 \
        auto kfn = conv1x1_stream_kernel<16, NJ_, true, 0, false, true>;                                                 \
        static bool attr[CN_MAX_DEVICES] = {};                                                                           \
        if (!attr[dev]) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[dev] = true; } \
        const size_t smem = (size_t)NJ_ * 32 * (16 * 16 + 8) * 2 + (size_t)NJ_ * 32 * 4;                                 \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(S1_NT), smem, st, g, npix);                                             \
        return true;                                                                                                     \
    } while (0)
    switch (nj) {
        case 1: S1_NCHW(1);
        case 2: S1_NCHW(2);
        case 3: S1_NCHW(3);
        default: S1_NCHW(4);
    }
#undef S1_NCHW
}

// caller guarantees: 1x1 / stride 1 / pad 0 geometry (normal or transposed: one tap, no shift), OH == H, OW == W
bool conv1x1_stream_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV1X1_STREAM") != nullptr;
    static int cus_of[CN_MAX_DEVICES] = {};       // CU count per device (a single process may drive several)
    if (disabled || dtype != CN_BF16 || g.nsrc != 0 || g.dcn_x != nullptr || g.res32 != nullptr) return false;
    if ((g.x_ld & 7) || (g.y_ld & 7) || (g.res != nullptr && (g.res_ld & 7))) return false;
    if ((reinterpret_cast<uintptr_t>(g.x) | reinterpret_cast<uintptr_t>(g.w) | reinterpret_cast<uintptr_t>(g.y) | reinterpret_cast<uintptr_t>(g.res)) & 15) return false;
    const int K = g.Ci, nj = (std::min(g.Co, g.y_ld) + 31) / 32;
    if (K != g.ktot || (K != 32 && K != 64 && K != 128 && K != 256) || nj < 1 || nj > 8 || g.co_pad < 32) return false;
    if ((size_t)nj * 32 * (K + 8) * 2 > 72 * 1024) return false;           // two workgroups per CU
    if (nj * 16 + 2 * (K / 16) * 4 > 200) return false;                    // accumulators + two fragment sets must leave 2 waves per SIMD
    const int64_t npix = (int64_t)g.N * g.OH * g.OW;
    if (npix < 64 * 1024) return false;                                    // small maps: the weight load per workgroup dominates
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CN_MAX_DEVICES) return false;
    if (cus_of[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) return false;
        cus_of[dev] = v;
    }
    const int cus = cus_of[dev];
    const int grid = (int)std::min<int64_t>(2 * (int64_t)cus, (npix / 32 + 3) / 4);
#define S1_K(KS_)                                                                              \
    switch (nj) {                                                                              \
        case 1: return s1_go<KS_, 1>(g, npix, grid, dev, st);                                       \
        case 2: return s1_go<KS_, 2>(g, npix, grid, dev, st);                                       \
        case 3: return s1_go<KS_, 3>(g, npix, grid, dev, st);                                       \
        case 4: return s1_go<KS_, 4>(g, npix, grid, dev, st);                                       \
        case 8: if (KS_ <= 8) return s1_go<(KS_ <= 8 ? KS_ : 8), 8>(g, npix, grid, dev, st); return false; \
        default: return false;                                                                 \
    }
    if (K == 32) { S1_K(2) }
    if (K == 64) { S1_K(4) }
    if (K == 128) { S1_K(8) }
    S1_K(16)
#undef S1_K
}
