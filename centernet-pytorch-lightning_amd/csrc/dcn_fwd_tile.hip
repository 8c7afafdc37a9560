// DCNv2 forward with every sampled operand tile-resident (bf16, Ci % 64 == 0, Co % 64 == 0).
//   y[p][co] = act(bias[co] + sum_k sum_ci W[co][ci][k] * mask[p,k] * bilinear(x[:, :, ci], pos(p,k)))       
// The first fused forward (dcn_fwd_kernel, dcn_fused.hip) gathers the 4 corners of every (pixel, tap) from global memory:
// 36 16-byte L2 gathers per pixel and 64-channel block, issued and waited for inside the tap loop (395 us on 64->64 @128^2,
// batch 64, 3.3x the plain 3x3 conv).  Here, like the offset/mask-gradient kernel, a PERSISTENT 512-thread workgroup (one per
// CU, XCD-contiguous tile runs) owns 8x16-pixel tiles and keeps in LDS
//   * the x HALO tile [(8+7) x (16+7)][64 ch] of the current channel block (zeros outside the image), loaded once and
//     prefetched into registers one (tile, block) ahead;
//   * the bilinear geometry of all 9 taps of the tile (h0, w0, lh, lw, mask), computed once per tile from offsets that were
//     prefetched during the previous tile;
//   * a double-buffered A tile [128 px][64 ch]: 4 lanes per pixel blend 16 channels each from four 32-byte LDS corner reads
//     (fp32 weights, v_pk_fma_f32), while the matrix cores multiply the previous tap's tile with its weight slice (3-deep
//     register ring, L2-resident).
// Corners outside the halo (offsets beyond about +-2 px) fall back to masked global loads.  The output tile leaves through LDS
// as 16-byte vectors with bias / ReLU applied in fp32.
#include "conv_common.h"
#include "dcn_common.h"

#define FT_TH 8
#define FT_TW 16
#define FT_HR 3                       // halo reaches from -3 to +4 around the tile
#define FT_HH (FT_TH + 7)
#define FT_HW (FT_TW + 7)
#define FT_HP (FT_HH * FT_HW)

struct FwdTileGeom {
    const bf16_t* x; const bf16_t* w; const float* om; const float* bias; bf16_t* y;
    int N, H, W, Ci, Co, x_ld, y_ld, om_ld, ktot, relu;
    float* bn_part; int bn_slots;     // BatchNorm statistics sink (cn_hooks.bn_part), nullable: sum / sum of squares of the stored values
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// acc[0..7] (8 fp32 pairs = 16 channels) += w * x[0..15]  (x: 2 x uint4 of bf16)
__device__ static inline void blend16(f32x2_t (&acc)[8], const uint4& lo, const uint4& hi, float w) {
    const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const f32x2_t ww = {w, w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2_t v = {__uint_as_float(d[i] << 16), __uint_as_float(d[i] & 0xffff0000u)};
        acc[i] = __builtin_elementwise_fma(v, ww, acc[i]);
    }
}

template <int BN>   // output channels per workgroup: 64 or 128
__global__ __launch_bounds__(512) void dcn_fwd_tile_kernel(const FwdTileGeom g) {
    CN_MAIN_PRIO_SET();
    constexpr int BM = FT_TH * FT_TW;           // 128 pixels
    constexpr int CP = 64 + 8;                  // pitch of the halo / A / weight tiles (bf16 elements)
    constexpr int NJ = BN / 64;                 // 32-channel MFMA blocks per wave
    constexpr int EP = BN + 8;                  // pitch of the output staging tile
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    bf16_t* const Xh = lds;                                  // [FT_HP][CP]
    bf16_t* const As = Xh + FT_HP * CP;                      // 2 x [128][CP]   (re-used as the [128][EP] output tile)
    bf16_t* const Bs = As + 2 * BM * CP;                     // 2 x [BN][CP]
    float4* const Geo = reinterpret_cast<float4*>(Bs + 2 * BN * CP);   // [9][128]: {h0 | w0 (int16 pair), lh, lw, mask}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + FT_TW - 1) / FT_TW, tiles_h = (g.H + FT_TH - 1) / FT_TH;
    const int tiles_img = tiles_w * tiles_h, ntiles = tiles_img * g.N;
    const int n0 = blockIdx.y * BN;
    const int ncb = g.Ci / 64;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * (BN / 2);
    const int G = gridDim.x;
    const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int tpb = (ntiles + G - 1) / G;
    const int t_begin = lb * tpb, t_end = min(ntiles, t_begin + tpb);
    if (t_begin >= t_end) return;
    // BatchNorm statistics of the stored values (sink protocol of bn.hip): this thread's channel vector over every pixel it stores
    const bool stats = g.bn_part != nullptr;
    float bs0[8], bs1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs0[e] = 0.f; bs1[e] = 0.f; }

    constexpr int XV = (FT_HP * 8 + 511) / 512;      // halo vectors per thread
    constexpr int GV = (BM * 9 + 511) / 512;         // (pixel, tap) geometry items per thread
    constexpr int BV = BN * 8 / 512;                 // weight-slice vectors per thread
    uint4 rx[XV], rb[3][BV];
    float rom[GV][3];

    auto halo_load = [&](int t, int cb) {            // branch-free: out-of-image pieces are masked, not skipped
        const int n = t / tiles_img, r = t % tiles_img;
        const int th0 = (r / tiles_w) * FT_TH, tw0 = (r % tiles_w) * FT_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * 512;
            const int hp = v >> 3, col = (v & 7) * 8;
            const int h = th0 - FT_HR + hp / FT_HW, w = tw0 - FT_HR + hp % FT_HW;
            rx[i] = ldg16_masked(g.x, ((img + (int64_t)h * g.W + w) * g.x_ld + cb * 64 + col) * 2,
                                 hp < FT_HP && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W);
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * 512;
            if (v < FT_HP * 8) st16(Xh + (v >> 3) * CP + (v & 7) * 8, rx[i]);
        }
    };
    auto om_load = [&](int t) {                      // item e -> (tap = e / 128, pixel = e % 128)
        const int n = t / tiles_img, r = t % tiles_img;
        const int th0 = (r / tiles_w) * FT_TH, tw0 = (r % tiles_w) * FT_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int e = tid + i * 512;
            const int tap = e >> 7, pl = e & 127;
            const int h = th0 + pl / FT_TW, w = tw0 + pl % FT_TW;
            const bool ok = tap < 9 && h < g.H && w < g.W;
            const float* o = g.om + (ok ? (img + (int64_t)h * g.W + w) * g.om_ld : 0);
            const int tp = ok ? tap : 0;
            const float a = o[2 * tp], b = o[2 * tp + 1], c = o[18 + tp];
            rom[i][0] = ok ? a : 0.f; rom[i][1] = ok ? b : 0.f; rom[i][2] = ok ? c : -INFINITY;    // dead pixel: mask = sigmoid(-inf) = 0
        }
    };
    auto geo_store = [&](int t) {
        const int r = t % tiles_img;
        const int th0 = (r / tiles_w) * FT_TH, tw0 = (r % tiles_w) * FT_TW;
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int e = tid + i * 512;
            const int tap = e >> 7, pl = e & 127;
            if (tap >= 9) continue;
            const int h = th0 + pl / FT_TW, w = tw0 + pl % FT_TW;
            const float py = (float)(h - 1 + tap / 3) + rom[i][0], px = (float)(w - 1 + tap % 3) + rom[i][1];
            // h0 <= -2 or h0 >= H: both corner rows are outside the image whatever the exact value is -> clamp into int16 range
            const float fh = fminf(fmaxf(floorf(py), -2.f), (float)g.H), fw = fminf(fmaxf(floorf(px), -2.f), (float)g.W);
            const int h0 = (int)fh, w0 = (int)fw;
            Geo[tap * BM + pl] = make_float4(__int_as_float((int)(((unsigned)h0 << 16) | ((unsigned)w0 & 0xffffu))), py - fh, px - fw, sigmoidf_(rom[i][2]));
        }
    };
    auto bload = [&](uint4 (&r)[BV], int q) {        // step q = cb * 9 + tap: rows n0.., k = tap*Ci + cb*64 .. +63
        const int cb = q / 9, tap = q % 9;
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            const int v = tid + i * 512;
            r[i] = ldg16(g.w + (int64_t)(n0 + (v >> 3)) * g.ktot + tap * g.Ci + cb * 64 + (v & 7) * 8);
        }
    };
    auto bstore = [&](const uint4 (&r)[BV], int buf) {
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            const int v = tid + i * 512;
            st16(Bs + buf * BN * CP + (v >> 3) * CP + (v & 7) * 8, r[i]);
        }
    };
    const int nsteps = ncb * 9;

    // builds the A tile of one tap: lane quad (pl, lq) blends channels lq*16..+15 of pixel pl
    auto build = [&](int tap, int buf, int th0, int tw0, int64_t img, int cb) {
        const int pl = tid >> 2, lq = tid & 3;
        const float4 gq = Geo[tap * BM + pl];
        const int hw = __float_as_int(gq.x);
        const int h0 = hw >> 16, w0 = (int)(short)(hw & 0xffff);
        const float lh = gq.y, lw = gq.z, mk = gq.w;
        const int hy = h0 - (th0 - FT_HR), hx = w0 - (tw0 - FT_HR);
        uint4 c00[2], c01[2], c10[2], c11[2];
        if (hy >= 0 && hy + 1 < FT_HH && hx >= 0 && hx + 1 < FT_HW) {          // all four corners inside the LDS halo
            const bf16_t* b = Xh + (hy * FT_HW + hx) * CP + lq * 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c00[u] = *reinterpret_cast<const uint4*>(b + u * 8);
                c01[u] = *reinterpret_cast<const uint4*>(b + CP + u * 8);
                c10[u] = *reinterpret_cast<const uint4*>(b + FT_HW * CP + u * 8);
                c11[u] = *reinterpret_cast<const uint4*>(b + FT_HW * CP + CP + u * 8);
            }
        } else {                                                              // rare: global memory, zero outside the image
            auto gl = [&](int hh, int ww, int u) {
                return ldg16_masked(g.x, ((img + (int64_t)hh * g.W + ww) * g.x_ld + cb * 64 + lq * 16 + u * 8) * 2,
                                    (unsigned)hh < (unsigned)g.H && (unsigned)ww < (unsigned)g.W);
            };
#pragma unroll
            for (int u = 0; u < 2; ++u) { c00[u] = gl(h0, w0, u); c01[u] = gl(h0, w0 + 1, u); c10[u] = gl(h0 + 1, w0, u); c11[u] = gl(h0 + 1, w0 + 1, u); }
        }
        f32x2_t a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f32x2_t{0.f, 0.f};
        blend16(a, c00[0], c00[1], (1.f - lh) * (1.f - lw) * mk);
        blend16(a, c01[0], c01[1], (1.f - lh) * lw * mk);
        blend16(a, c10[0], c10[1], lh * (1.f - lw) * mk);
        blend16(a, c11[0], c11[1], lh * lw * mk);
        bf16_t* dst = As + buf * BM * CP + pl * CP + lq * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(pk_bf16(a[0].x, a[0].y), pk_bf16(a[1].x, a[1].y), pk_bf16(a[2].x, a[2].y), pk_bf16(a[3].x, a[3].y));
        *reinterpret_cast<uint4*>(dst + 8) = make_uint4(pk_bf16(a[4].x, a[4].y), pk_bf16(a[5].x, a[5].y), pk_bf16(a[6].x, a[6].y), pk_bf16(a[7].x, a[7].y));
    };

    // ---- prologue: first tile's halo + offsets, first three weight slices ----
    halo_load(t_begin, 0);
    om_load(t_begin);
#pragma unroll
    for (int d = 0; d < 3; ++d) bload(rb[d], d % nsteps);

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int n = t / tiles_img, r = t % tiles_img;
        const int th0 = (r / tiles_w) * FT_TH, tw0 = (r % tiles_w) * FT_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
        const int tn = (t + 1 < t_end) ? t + 1 : t;            // the last tile prefetches itself again: no branches around loads
        f32x16_t acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[j][rr] = 0.f;

        geo_store(t);                                          // everyone is past the previous tile's last build (sync below)
#pragma unroll 1
        for (int cb = 0; cb < ncb; ++cb) {
            halo_store();                                      // previous block's builds ended before the last barrier
            __syncthreads();
            // prefetch the next halo (next channel block, or block 0 of the next tile) and, on the last block, the next offsets
            if (cb + 1 < ncb) halo_load(t, cb + 1);
            else { halo_load(tn, 0); om_load(tn); }
            // invariant at step q = cb*9 + tap: ring slot q%3 (= tap%3) holds slice q, the other two hold q+1, q+2
            build(0, 0, th0, tw0, img, cb);
            bstore(rb[0], 0);
            bload(rb[0], (cb * 9 + 3) % nsteps);
            __syncthreads();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int buf = tap & 1;
                if (tap < 8) {
                    build(tap + 1, buf ^ 1, th0, tw0, img, cb);
                    bstore(rb[(tap + 1) % 3], buf ^ 1);
                    bload(rb[(tap + 1) % 3], (cb * 9 + tap + 4) % nsteps);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8_t fa = Mma<bf16_t>::load(As + buf * BM * CP, CP, wm, kk, lane);              // pixels
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const bf16x8_t fb = Mma<bf16_t>::load(Bs + buf * BN * CP, CP, wn + j * 32, kk, lane);   // output channels
                        acc[j] = Mma<bf16_t>::mma(fb, fa, acc[j]);
                    }
                }
                __syncthreads();
            }
        }
        // ---- epilogue: bias / ReLU in fp32 -> bf16 tile in LDS -> 16-byte NHWC stores ----
        bf16_t* const Es = As;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = wn + j * 32 + 8 * q + 4 * (lane >> 5);     // channel within the workgroup's block
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = n0 + cl + e;
                    float val = acc[j][q * 4 + e] + ((g.bias && c < g.Co) ? g.bias[c] : 0.f);
                    v[e] = g.relu ? fmaxf(val, 0.f) : val;
                }
                *reinterpret_cast<uint2*>(Es + (wm + (lane & 31)) * EP + cl) = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BM * (BN / 8) / 512; ++i) {
            const int v = tid + i * 512;
            const int pl = v / (BN / 8), col = (v % (BN / 8)) * 8;
            const int h = th0 + pl / FT_TW, w = tw0 + pl % FT_TW;
            if (h < g.H && w < g.W && n0 + col < g.y_ld) {
                const uint4 o = *reinterpret_cast<const uint4*>(Es + pl * EP + col);
                *reinterpret_cast<uint4*>(g.y + (img + (int64_t)h * g.W + w) * g.y_ld + n0 + col) = o;
                if (stats) {                                   // a thread stores the same channel vector in every pass of every tile
                    const uint32_t wv[4] = {o.x, o.y, o.z, o.w};
                    bn_stat_add(bs0, bs1, wv);
                }
            }
        }
        __syncthreads();                                       // Es (= As) and Geo are rewritten by the next tile
    }
    if (stats) {
        static_assert(512 % (BN / 8) == 0, "thread -> channel-vector map of the statistics");
        bn_stats_flush<BN / 8, 512>(bs0, bs1, reinterpret_cast<float*>(As), g.bn_part, g.bn_slots, g.y_ld, n0, g.Co, blockIdx.x + blockIdx.y * gridDim.x, tid);
    }
}

// returns false when the shape is not handled by the tile-resident kernel
bool dcn_fwd_tile_shape_ok(int Ci, int x_ld, int Co, int y_ld) {
    static const bool disabled = getenv("CN_DISABLE_DCN_FWD_TILE") != nullptr;
    static const bool forced = getenv("CN_FORCE_DCN_FWD_TILE") != nullptr;
    if (disabled || Ci % 64 != 0 || Co % 64 != 0 || (x_ld & 7) || (y_ld & 7) || y_ld != Co) return false;
    // measured (MI355X, batch 64): 1.2-2.1x over the global-gather kernel once the halo is re-used by >= 2 channel blocks and 128
    // output channels (230 vs 279 us 128->128@64^2, 100 vs 210 us 512->256@16^2); 64->64@128^2 is LDS + VALU bound here (431 vs
    // 395 us: 64 KB of corner reads + 64 KB of MFMA operand reads per tap) and stays on the gather / blend-matrix kernels.
    return forced || (Ci >= 128 && Co >= 128);
}

bool dcn_fwd_tile_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                         int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st) {
    if (!dcn_fwd_tile_shape_ok(Ci, x_ld, Co, y_ld) || relu > 1) return false;
    FwdTileGeom g;
    g.x = (const bf16_t*)x; g.w = (const bf16_t*)wp; g.om = om; g.bias = bias; g.y = (bf16_t*)y;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.Co = Co; g.x_ld = x_ld; g.y_ld = y_ld; g.om_ld = om_ld; g.ktot = ktot; g.relu = relu;
    static const bool no_stats = getenv("CN_DISABLE_DCN_TILE_STATS") != nullptr;
    g.bn_part = (bn_slots > 0 && !no_stats) ? bn_part : nullptr; g.bn_slots = bn_slots;
    if (g.bn_part) mark_taken(bn_taken);            // the LDS-staged epilogue has the statistics hook
    const int bn = (Co % 128 == 0) ? 128 : 64;
    const int nco = Co / bn;
    const int ntiles = ((H + FT_TH - 1) / FT_TH) * ((W + FT_TW - 1) / FT_TW) * N;
    int gx = 256 / nco;                          // one persistent workgroup per CU
    if (gx < 8) gx = 8;
    if (gx > ntiles) gx = ntiles;
    dim3 grid(gx, nco, 1);
    const size_t smem = ((size_t)FT_HP * 72 + 2 * 128 * 72 + 2 * (size_t)bn * 72) * sizeof(bf16_t) + (size_t)9 * 128 * sizeof(float4);
    if (bn == 64) {
        (void)hipFuncSetAttribute((const void*)dcn_fwd_tile_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_fwd_tile_kernel<64>, grid, dim3(512), smem, st, g);
    } else {
        (void)hipFuncSetAttribute((const void*)dcn_fwd_tile_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_fwd_tile_kernel<128>, grid, dim3(512), smem, st, g);
    }
    return true;
}

