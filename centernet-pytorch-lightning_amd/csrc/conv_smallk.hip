// 1x1 convolution with a 16-wide (padded) contraction — the data gradients of the 2-channel CenterNet heads
// (width_height / regression / keypoint offsets: dY [P][16] x W^T [16][256] -> dH [P][256], heads.py:9-15 backwards).
// On the implicit-GEMM kernel these launches are pure overhead: one 16-wide K slice per 128x128 tile, a 64 KB fp32 LDS slab
// for the epilogue and one workgroup per CU — 502 us for 1.07 GB of traffic (mask read + output write), 17 TFLOP/s.
// Here it is a streaming VALU kernel (cn_conv1x1_smallk, real contraction length K <= 4 given by the caller): a thread owns 8
// output channels, keeps their K x 8 weights in registers and per pixel does K x 8 FMAs between one 16-byte mask load and one
// 16-byte store; 4 pixels are in flight per thread, ~4 long-running workgroups per CU.
#include "conv_common.h"

template <int KE>
__device__ static inline void smallk_body(const ConvGeom& g, const uint4 (&wr)[16], int cg, int pr, int ppb, int64_t P) {
    float w[KE][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t d[8] = {wr[2 * e].x, wr[2 * e].y, wr[2 * e].z, wr[2 * e].w, wr[2 * e + 1].x, wr[2 * e + 1].y, wr[2 * e + 1].z, wr[2 * e + 1].w};
#pragma unroll
        for (int k = 0; k < KE; ++k) w[k][e] = __uint_as_float((k & 1) ? (d[k >> 1] & 0xffff0000u) : (d[k >> 1] << 16));
    }
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ R = reinterpret_cast<const bf16_t*>(g.res);
    bf16_t* __restrict__ Y = reinterpret_cast<bf16_t*>(g.y);
    const int ch = cg * 8;
    const bool mask_mode = g.relu == 2;
    constexpr int U = 4;
    const int64_t p_begin = (int64_t)blockIdx.x * ppb, p_end = min(P, p_begin + ppb);
    const int rows = blockDim.x / (g.Co / 8);                   // pixels per pass
    for (int64_t p0 = p_begin + pr; p0 < p_end; p0 += (int64_t)rows * U) {
        uint4 xv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                           // branch-free: rows past the end re-read the last row and are not stored
            const int64_t p = min(p0 + (int64_t)u * rows, p_end - 1);
            xv[u] = KE > 8 ? ldg16(X + p * g.x_ld) : ldg16(X + p * g.x_ld);
            rv[u] = R ? ldg16(R + p * g.res_ld + ch) : make_uint4(0, 0, 0, 0);
        }
        uint4 xv2[U];
        if constexpr (KE > 8) {
#pragma unroll
            for (int u = 0; u < U; ++u) xv2[u] = ldg16(X + min(p0 + (int64_t)u * rows, p_end - 1) * g.x_ld + 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + (int64_t)u * rows;
            const uint32_t dx[8] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w, KE > 8 ? xv2[u].x : 0u, KE > 8 ? xv2[u].y : 0u, KE > 8 ? xv2[u].z : 0u, KE > 8 ? xv2[u].w : 0u};
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
            for (int k = 0; k < KE; ++k) {
                const float xk = __uint_as_float((k & 1) ? (dx[k >> 1] & 0xffff0000u) : (dx[k >> 1] << 16));
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(xk, w[k][e], v[e]);
            }
            if (R) {
                const uint32_t dr[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r = __uint_as_float((e & 1) ? (dr[e >> 1] & 0xffff0000u) : (dr[e >> 1] << 16));
                    v[e] = mask_mode ? (r > 0.f ? v[e] : 0.f) : v[e] + r;
                }
            }
            if (g.relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p < p_end) st16(Y + p * g.y_ld + ch, make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])));
        }
    }
}

template <int KE>
__global__ __launch_bounds__(256) void conv1x1_smallk_kernel(const ConvGeom g, int ppb, int64_t P) {
    CN_MAIN_PRIO_SET();
    const int cpg = g.Co / 8;
    const int cg = threadIdx.x % cpg, pr = threadIdx.x / cpg;
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);      // [co_pad][16], columns >= K are zero padding
    uint4 wr[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        wr[2 * e] = ldg16(Wp + (int64_t)(cg * 8 + e) * 16);
        wr[2 * e + 1] = make_uint4(0, 0, 0, 0);
    }
    if (pr >= 256 / cpg) return;
    smallk_body<KE>(g, wr, cg, pr, ppb, P);
}

// ---- K = 17 .. 96 (padded to a multiple of 16), Co = 256: the 80-class / keypoint heads' data gradient ------------------------
// Still a stream (1.07 GB of mask + output against 43 GFLOP), but too much arithmetic for the VALU: a wave owns 32 pixels and
// all 256 output channels (8 accumulators of v_mfma_f32_32x32x16_bf16), takes its pixel fragments straight from global memory
// (a pixel's K_pad channels are one contiguous run), the weights [256][K_pad] sit in LDS for the workgroup's lifetime, and
// the result leaves through a per-wave LDS tile so that the mask is read and the output written as 16-byte vectors.
template <int KS>      // K_pad / 16
__global__ __launch_bounds__(256, 2) void conv1x1_midk_kernel(const ConvGeom g, int64_t P, int ngroups_per_block) {
    CN_MAIN_PRIO_SET();
    constexpr int KP = KS * 16, WP = KP + 8, CO = 256, EP = 128 + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    bf16_t* const Ws = lds;                                  // [256][WP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16_t* const Es = Ws + CO * WP + wave * 32 * EP;        // per-wave [32 px][128 ch (+8)] staging tile (two passes)
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ R = reinterpret_cast<const bf16_t*>(g.res);
    bf16_t* __restrict__ Y = reinterpret_cast<bf16_t*>(g.y);
    for (int v = tid; v < CO * (KP / 8); v += 256) {
        const int row = v / (KP / 8), col = (v % (KP / 8)) * 8;
        st16(Ws + row * WP + col, ldg16(reinterpret_cast<const bf16_t*>(g.w) + (int64_t)row * KP + col));
    }
    __syncthreads();
    const int px = lane & 31, half = lane >> 5;
    const int64_t G = (P + 31) / 32;                         // 32-pixel groups
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + wave) * ngroups_per_block;
    const bool mask_mode = g.relu == 2;
    uint4 xb[KS];
    auto xload = [&](int64_t grp) {
        const int64_t p = min(grp * 32 + px, P - 1);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) xb[kk] = ldg16(X + p * g.x_ld + kk * 16 + half * 8);
    };
    if (g0 < G) xload(g0);
#pragma unroll 1
    for (int it = 0; it < ngroups_per_block; ++it) {
        const int64_t grp = g0 + it;
        if (grp >= G) break;
        f32x16_t acc[8];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        bf16x8_t fx[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) fx[kk] = __builtin_bit_cast(bf16x8_t, xb[kk]);
        if (it + 1 < ngroups_per_block && grp + 1 < G) xload(grp + 1);          // next group's pixels while this one is multiplied
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int rb = 0; rb < 8; ++rb)
                acc[rb] = Mma<bf16_t>::mma(Mma<bf16_t>::load(Ws, WP, rb * 32, kk, lane), fx[kk], acc[rb]);
        // lane holds pixel px, channels 32*rb + 8*q + 4*half + e  ->  two 128-channel passes through the wave's LDS tile
#pragma unroll
        for (int hpass = 0; hpass < 2; ++hpass) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x16_t& a = acc[hpass * 4 + rb];
                    *reinterpret_cast<uint2*>(Es + px * EP + rb * 32 + 8 * q + 4 * half) =
                        make_uint2(pk_bf16(a[q * 4], a[q * 4 + 1]), pk_bf16(a[q * 4 + 2], a[q * 4 + 3]));
                }
            // 32 px x 128 ch = 512 vectors of 16 B: 8 per lane; lane -> (pixel = v / 16, channel vector = v % 16): 256-B runs per pixel
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int v = lane + j * 64;
                const int pl = v >> 4, cv = v & 15;
                const int64_t p = grp * 32 + pl;
                if (p >= P) continue;
                const int ch = hpass * 128 + cv * 8;
                uint4 o = *reinterpret_cast<const uint4*>(Es + pl * EP + cv * 8);
                if (R) {
                    const uint4 rr = ldg16(R + p * g.res_ld + ch);
                    const uint32_t od[4] = {o.x, o.y, o.z, o.w}, rd[4] = {rr.x, rr.y, rr.z, rr.w};
                    uint32_t nd[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r0 = __uint_as_float(rd[e] << 16), r1 = __uint_as_float(rd[e] & 0xffff0000u);
                        if (mask_mode) nd[e] = (r0 > 0.f ? (od[e] & 0xffffu) : 0u) | (r1 > 0.f ? (od[e] & 0xffff0000u) : 0u);
                        else nd[e] = pk_bf16(__uint_as_float(od[e] << 16) + r0, __uint_as_float(od[e] & 0xffff0000u) + r1);
                    }
                    o = make_uint4(nd[0], nd[1], nd[2], nd[3]);
                }
                st16(Y + p * g.y_ld + ch, o);
            }
        }
    }
}

// y[P][Co] = act(x[P][0..K-1] . wp[Co][0..K-1] (+ residual | masked by residual > 0)); bf16; x pitch x_ld, weights packed with
// x_ld-wide rows (cn_pack_weight).  Replaces conv_igemm for the heads' 1x1 data gradients (heads.py:9-15 backwards): K = 1 | 2 on
// the VALU kernel, K = 17 .. 96 with 256 outputs on the MFMA stream.
extern "C" int cn_conv1x1_smallk(const void* x, const void* wp, const void* residual, void* y, int64_t P, int K, int x_ld, int Co,
                                 int y_ld, int res_ld, int relu, int dtype, void* stream) {
    CN_CHECK_ARG(x && wp && y && P > 0 && K > 0 && Co > 0, "cn_conv1x1_smallk: bad args");
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.w = wp; g.res = residual; g.y = y; g.Co = Co; g.y_ld = y_ld; g.x_ld = x_ld; g.res_ld = res_ld; g.relu = relu;
    if (dtype == CN_BF16 && K > 4 && K <= 96 && Co == 256 && y_ld == 256 && x_ld % 16 == 0 && x_ld >= K && x_ld <= 96 && relu != 1 &&
        !(residual && (res_ld & 7))) {
        // MFMA stream: weights packed with x_ld-wide rows (cn_pack_weight pads the contraction to the activation pitch)
        const int ks = x_ld / 16;
        const int64_t G = (P + 31) / 32;
        int64_t blocks = (G + 3) / 4;
        if (blocks > 512) blocks = 512;                     // two workgroups per CU, each keeps the weights for its whole run
        const int gpb = (int)((G + blocks * 4 - 1) / (blocks * 4));
        const size_t smem = ((size_t)256 * (x_ld + 8) + 4 * 32 * (128 + 8)) * sizeof(bf16_t);
#define CN_MIDK(KS_)                                                                                                              \
    case KS_:                                                                                                                     \
        (void)hipFuncSetAttribute((const void*)conv1x1_midk_kernel<KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
        hipLaunchKernelGGL(conv1x1_midk_kernel<KS_>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, g, P, gpb);    \
        break;
        switch (ks) { CN_MIDK(1) CN_MIDK(2) CN_MIDK(3) CN_MIDK(4) CN_MIDK(5) CN_MIDK(6) default: CN_UNSUPPORTED("cn_conv1x1_smallk: x_ld=%d", x_ld); }
#undef CN_MIDK
        CN_LAUNCH_CHECK("cn_conv1x1_smallk(mfma)");
        return CN_OK;
    }
    if (dtype != CN_BF16 || K > 4 || (Co & 7) || Co < 64 || Co > 2048 || 256 % (Co / 8) != 0 || y_ld != Co || (x_ld & 7) || x_ld < 8 ||
        (residual && (res_ld & 7)))
        CN_UNSUPPORTED("cn_conv1x1_smallk: bf16; K <= 4 with Co in {64,...,2048} = y_ld, or K <= 96 with Co = 256 (K=%d Co=%d y_ld=%d x_ld=%d)", K, Co, y_ld, x_ld);
    const int rows = 256 / (Co / 8);
    // ~4 workgroups per CU, each streaming a long run of pixels: the weight prologue (one L2 round trip) is paid once
    int64_t blocks = (P + rows * 16 - 1) / (rows * 16);
    if (blocks > 1024) blocks = 1024;
    const int ppb = (int)(((P + blocks - 1) / blocks + rows * 4 - 1) / (rows * 4)) * rows * 4;
    blocks = (P + ppb - 1) / ppb;
    if (K <= 2) hipLaunchKernelGGL(conv1x1_smallk_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, ppb, P);
    else hipLaunchKernelGGL(conv1x1_smallk_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, ppb, P);
    CN_LAUNCH_CHECK("cn_conv1x1_smallk");
    return CN_OK;
}
