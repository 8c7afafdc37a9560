// Weight-gradient GEMM for gfx950:  dWp[co][tap*Ci + ci] += sum_p dY[p][co] * Xg[p, tap][ci]
//
// The contraction runs over PIXELS, which is the slow axis of both NHWC operands, so MFMA
// fragments (k-contiguous per lane) need a transpose on the way in:
//   * bf16: tiles are staged in their natural [pixel][channel] order with plain 16-byte LDS stores and the
//     fragments are fetched with gfx950's transposing LDS read ds_read_b64_tr_b16 (a 16-lane group reads a
//     4-pixel x 16-channel block and each lane receives the 4 pixels of ITS channel; semantics probed in
//     tools/probe/tr_read.hip).  Rows are padded by 64 B so the four pixel rows of a read hit disjoint banks.
//     MFMA: v_mfma_f32_32x32x16_bf16.
//   * fp32 (parity mode): v_mfma_f32_32x32x2_f32 takes ONE f32 per lane, so the natural
//     [pixel][channel] image is already conflict-free — no transpose.
// Split-K over pixel chunks (blockIdx.x) with fp32 atomics into the (pre-zeroed) packed gradient;
// blockIdx.z = tap, blockIdx.y = (co tile, ci tile).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// 32(channel) x 16(pixel) bf16 MFMA operand out of a [pixel][channel] LDS tile (pitch in elements).
// lane l: channel r0 + (l&31), pixels k0 + 8*(l>>5) .. +7.
__device__ static inline bf16x8_t tr_frag(const bf16_t* tile, int pitch, int r0, int k0, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const bf16_t* p = tile + (k0 + 8 * (g >> 1) + (r >> 2)) * pitch + r0 + 16 * (g & 1) + 4 * (r & 3);
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * pitch));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

bool small_wgrad_packed(const void* x, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co,
                        int dy_ld, int KH, int KW, int stride, int pad, int dtype, hipStream_t st, const float* pre_ss, int pre_relu, int target);   // stem.hip

bool wgrad3x3s1_launch(const void* x, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                       hipStream_t st, int target);                                                     // conv_wgrad3x3.hip

struct WgradGeom {
    const void* x;
    const void* dy;
    float* dwp;
    int N, H, W, Ci, x_ld;
    int OH, OW, Co, dy_ld;
    int KW, stride, pad;
    int ktot, co_pad;
    int ci_tiles;
    int64_t P;          // N*OH*OW
    int64_t chunk;      // pixels per split-K chunk (multiple of the K tile)
    int target;         // workgroups wanted over the whole launch (host side only: cn_hooks.wgrad_blocks)
};

template <typename T, int BMW, int BNW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradGeom g) {
    constexpr bool BF = sizeof(T) == 2;
    // pixels per K tile: 64 where two pipeline stages of it fit the 64 KB of static LDS (every tile but 128 x 128) — four MFMAs per wave
    // and barrier instead of two
    constexpr int BKP = BF ? ((BMW + BNW <= 192) ? 64 : 32) : 16;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int YV = BKP * BMW / VEC / 256;      // 16-byte loads per thread for the dY tile
    constexpr int XV = BKP * BNW / VEC / 256;
    constexpr int YCG = BMW / VEC, XCG = BNW / VEC;  // channel groups per pixel row
    constexpr int WM = BMW / 2, WN = BNW / 2;      // 2x2 waves
    constexpr int MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = BF ? BKP / 16 : 8;
    static_assert(YV >= 1 && XV >= 1, "tile too small");

    constexpr int PADE = BF ? 32 : 0;               // bf16 rows padded by 64 B (bank spread for the transposing reads)
    constexpr int YP = BMW + PADE, XP = BNW + PADE; // row pitches in elements
    constexpr int BUF = (YP + XP) * BKP;            // elements per pipeline stage: [dY tile | X tile]
    __shared__ __attribute__((aligned(16))) T lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tap = blockIdx.z;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    const int co0 = (blockIdx.y / g.ci_tiles) * BMW;
    const int ci0 = (blockIdx.y % g.ci_tiles) * BNW;
    const int64_t pbeg = (int64_t)blockIdx.x * g.chunk;
    const int64_t pend = (pbeg + g.chunk < g.P) ? pbeg + g.chunk : g.P;
    if (pbeg >= g.P) return;
    const int ntiles = (int)((pend - pbeg + BKP - 1) / BKP);

    const T* __restrict__ X = reinterpret_cast<const T*>(g.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(g.dy);

    // per-thread load slots: slot v -> (pixel-in-tile, channel group)
    int y_px[YV], y_cg[YV], x_px[XV], x_cg[XV];
    // running (n, oh, ow) of the X-loader pixels (advance by BKP per tile)
    int xo_n[XV], xo_h[XV], xo_w[XV];
#pragma unroll
    for (int v = 0; v < YV; ++v) {
        int idx = tid + v * 256;
        y_px[v] = idx / YCG;
        y_cg[v] = idx % YCG;
    }
#pragma unroll
    for (int v = 0; v < XV; ++v) {
        int idx = tid + v * 256;
        x_px[v] = idx / XCG;
        x_cg[v] = idx % XCG;
        int64_t p = pbeg + x_px[v];
        int n = (int)(p / ((int64_t)g.OH * g.OW));
        int r = (int)(p - (int64_t)n * g.OH * g.OW);
        xo_n[v] = n;
        xo_h[v] = r / g.OW;
        xo_w[v] = r - xo_h[v] * g.OW;
    }

    uint4 ry[YV], rx[XV];
    auto gload = [&](int tile) {
        const int64_t p0 = pbeg + (int64_t)tile * BKP;
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            int64_t p = p0 + y_px[v];
            int c = co0 + y_cg[v] * VEC;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (p < pend && c < g.Co) val = *reinterpret_cast<const uint4*>(DY + p * g.dy_ld + c);  // dy_ld padded to VEC
            ry[v] = val;
        }
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            int64_t p = p0 + x_px[v];
            int c = ci0 + x_cg[v] * VEC;
            int ih = xo_h[v] * g.stride - g.pad + kh, iw = xo_w[v] * g.stride - g.pad + kw;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (p < pend && c < g.Ci && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
                val = *reinterpret_cast<const uint4*>(X + (((int64_t)xo_n[v] * g.H + ih) * g.W + iw) * g.x_ld + c);
            rx[v] = val;
            // advance this slot's pixel by one K tile
            xo_w[v] += BKP;
            while (xo_w[v] >= g.OW) { xo_w[v] -= g.OW; xo_h[v] += 1; }
            while (xo_h[v] >= g.OH) { xo_h[v] -= g.OH; xo_n[v] += 1; }
        }
    };
    auto lstore = [&](int buf) {
        if constexpr (BF) {
            bf16_t* ys = reinterpret_cast<bf16_t*>(lds + buf * BUF);
            bf16_t* xs = reinterpret_cast<bf16_t*>(lds + buf * BUF + YP * BKP);
#pragma unroll
            for (int v = 0; v < YV; ++v) *reinterpret_cast<uint4*>(ys + y_px[v] * YP + y_cg[v] * 8) = ry[v];
#pragma unroll
            for (int v = 0; v < XV; ++v) *reinterpret_cast<uint4*>(xs + x_px[v] * XP + x_cg[v] * 8) = rx[v];
        } else {
#pragma unroll
            for (int v = 0; v < YV; ++v)
                *reinterpret_cast<uint4*>(reinterpret_cast<float*>((lds + buf * BUF)) + y_px[v] * BMW + y_cg[v] * 4) = ry[v];
#pragma unroll
            for (int v = 0; v < XV; ++v)
                *reinterpret_cast<uint4*>(reinterpret_cast<float*>((lds + buf * BUF + YP * BKP)) + x_px[v] * BNW + x_cg[v] * 4) = rx[v];
        }
    };

    f32x16_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;

    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int tile = 0; tile < ntiles; ++tile) {
        const bool more = tile + 1 < ntiles;
        if (more) gload(tile + 1);
        if constexpr (BF) {
            const bf16_t* yt = reinterpret_cast<const bf16_t*>(lds + cur * BUF);
            const bf16_t* xt = reinterpret_cast<const bf16_t*>(lds + cur * BUF + YP * BKP);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                bf16x8_t fa[MI], fb[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = tr_frag(yt, YP, wm + i * 32, kk * 16, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = tr_frag(xt, XP, wn + j * 32, kk * 16, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const float* yt = reinterpret_cast<const float*>((lds + cur * BUF));
            const float* xt = reinterpret_cast<const float*>(lds + cur * BUF + YP * BKP);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                float fa[MI], fb[NJ];
                const int k = kk * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = yt[k * BMW + wm + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = xt[k * BNW + wn + j * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // D rows = co: (r&3) + 8*(r>>2) + 4*(lane>>5); D col = ci: lane&31
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ci = ci0 + wn + j * 32 + (lane & 31);
            if (ci >= g.Ci) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.Co) atomicAdd(g.dwp + (int64_t)co * g.ktot + (int64_t)tap * g.Ci + ci, acc[i][j][r]);
            }
        }
}

// bias gradient: db[c] += sum_p dy[p][c].  16-byte loads, lanes = (channel vector, pixel row), LDS fold over rows,
// one atomic per channel per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, float* __restrict__ db, int64_t P, int C,
                                                     int ld, int CVB, int64_t rows_per_blk) {
    constexpr int V = Vec16<T>::N;
    __shared__ float red[256][V + 1];
    const int tid = threadIdx.x, cvl = tid % CVB, prow = tid / CVB, RPB = 256 / CVB;
    const int cv = blockIdx.y * CVB + cvl;
    float s[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = 0.f;
    if (prow < RPB && cv * V < C) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk < P ? r0 + rows_per_blk : P;
        int64_t r = r0 + prow;
        for (; r + 15 * RPB < r1; r += 16 * RPB) {        // 16 independent 16-byte loads in flight per thread (64 KB per workgroup)
            uint4 q[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) q[u] = ldg16(dy + (r + u * RPB) * ld + cv * V);           // ld is padded to a vector multiple
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8][V];
#pragma unroll
                for (int u = 0; u < 8; ++u) Vec16<T>::unpack(q[h * 8 + u], v[u]);
#pragma unroll
                for (int j = 0; j < V; ++j) s[j] += ((v[0][j] + v[1][j]) + (v[2][j] + v[3][j])) + ((v[4][j] + v[5][j]) + (v[6][j] + v[7][j]));
            }
        }
        for (; r < r1; r += RPB) {
            float v[V];
            Vec16<T>::load(dy + r * ld + cv * V, v);
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) red[tid][j] = s[j];
    __syncthreads();
    int top = 1;
    while (top < RPB) top <<= 1;
    for (int st = top >> 1; st > 0; st >>= 1) {           // tree over the row slots (a serial fold by CVB threads took longer than the loads)
        if (prow < st && prow + st < RPB) {
#pragma unroll
            for (int j = 0; j < V; ++j) red[tid][j] += red[tid + st * CVB][j];
        }
        __syncthreads();
    }
    if (prow == 0 && cv * V < C) {
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (cv * V + j < C) atomicAdd(db + cv * V + j, red[tid][j]);
    }
}

// Weight-gradient kernels split K (= pixels) over cn_hooks.wgrad_blocks workgroups (cn_wgrad_target: CN_WGRAD_DEFAULT_BLOCKS without
// hooks).  Alone on the GPU they want ~6 per CU (1536); when they run on a side stream next to the data-gradient chain a THIN
// grid (~384) leaves the CUs to the chain that is on the critical path: measured +4 % step throughput on DLA-34 (the side work
// has 2x slack).  The count is an argument of the call — there is no process-wide setting.

template <typename T, int BMW, int BNW>
static void launch_wgrad(WgradGeom& g, int taps, hipStream_t st) {
    constexpr int BKP = sizeof(T) == 2 ? ((BMW + BNW <= 192) ? 64 : 32) : 16;      // as in the kernel
    int co_tiles = cdiv(g.Co, BMW);
    g.ci_tiles = cdiv(g.Ci, BNW);
    int base = co_tiles * g.ci_tiles * taps;
    int64_t want = (g.target * 4 / 3 + base - 1) / base;   // default: >= ~2048 workgroups
    int64_t maxk = (g.P + 8 * BKP - 1) / (8 * BKP);    // at least 8 K tiles per workgroup
    if (want > maxk) want = maxk;
    if (want < 1) want = 1;
    g.chunk = ((g.P + want - 1) / want + BKP - 1) / BKP * BKP;
    int kchunks = (int)((g.P + g.chunk - 1) / g.chunk);
    dim3 grid(kchunks, co_tiles * g.ci_tiles, taps);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, BMW, BNW>), grid, dim3(256), 0, st, g);
}

static int launch_colsum(const void* dy, float* db, int64_t P, int Co, int dy_ld, int dtype, hipStream_t st) {
    const int V = dtype == CN_F32 ? 4 : 8;
    const int CV = (Co + V - 1) / V;
    const int CVB = CV < 256 ? CV : 256;
    const int RPB = 256 / CVB;
    int64_t nblk = (P + (int64_t)RPB * 32 - 1) / ((int64_t)RPB * 32);
    // same-address atomics serialise (~0.2 us each: 256 workgroups spent 50 us in the chain on a 16 MB input): few workgroups,
    // each with 64 KB of loads in flight
    static const int cap = getenv("CN_COLSUM_BLOCKS") ? atoi(getenv("CN_COLSUM_BLOCKS")) : 64;
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    const int64_t rows_per_blk = (P + nblk - 1) / nblk;
    dim3 grid((int)nblk, (CV + CVB - 1) / CVB);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, st, (const T*)dy, db, P, Co, dy_ld, CVB,
                                                   rows_per_blk));
    CN_LAUNCH_CHECK("colsum");
    return CN_OK;
}

// db[c] += sum_p dy[p][c]  (bias gradient on its own; db is accumulated into)
extern "C" int cn_colsum(const void* dy, float* db, int64_t P, int Co, int dy_ld, int dtype, void* stream) {
    CN_CHECK_ARG(dy && db && P > 0 && Co > 0, "cn_colsum: bad args");
    const int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(dy_ld % V == 0 && dy_ld >= ((Co + V - 1) / V) * V, "cn_colsum: dy_ld must be a vector multiple covering Co");
    return launch_colsum(dy, db, P, Co, dy_ld, dtype, (hipStream_t)stream);
}

// Weight gradient straight into the PARAMETER layout dw[Co][Ci][KH][KW] (fp32; accumulate != 0 adds to it), for the shapes whose
// kernel has the slab form (bf16, 3x3 / stride 1 or 2 / pad 1, Ci > 16): split-K partials leave as private slabs in `ws` and one
// reduction launch sums them — no fp32 atomics, no pre-zeroed packed gradient, no unpack launch, fixed summation order.
// cn_conv2d_wgrad_direct_bytes: scratch size, 0 = shape not handled here (use cn_conv2d_wgrad + cn_unpack_wgrad).
bool wgrad3x3s1_slab_launch(const void* x, const void* dy, float* slabs, float* dw, int accumulate, int N, int H, int W, int Ci, int x_ld,
                            int Co, int dy_ld, int stride, hipStream_t st, int target);
size_t wgrad3x3s1_slab_bytes(int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld, int stride, int target);
extern "C" size_t cn_conv2d_wgrad_direct_bytes_h(int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld, int KH, int KW,
                                                 int stride, int pad, int dtype, int wgrad_blocks) {
    if (dtype != CN_BF16 || KH != 3 || KW != 3 || (stride != 1 && stride != 2) || pad != 1 || Ci <= 16) return 0;
    if (OH != (H + 2 - 3) / stride + 1 || OW != (W + 2 - 3) / stride + 1) return 0;
    return wgrad3x3s1_slab_bytes(N, H, W, Ci, x_ld, Co, dy_ld, stride, wgrad_blocks > 0 ? wgrad_blocks : CN_WGRAD_DEFAULT_BLOCKS);
}
extern "C" size_t cn_conv2d_wgrad_direct_bytes(int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld, int KH, int KW,
                                               int stride, int pad, int dtype) {
    return cn_conv2d_wgrad_direct_bytes_h(N, H, W, Ci, x_ld, OH, OW, Co, dy_ld, KH, KW, stride, pad, dtype, 0);
}
extern "C" int cn_conv2d_wgrad_direct_h(const void* x, const void* dy, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                                        int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                                        int KH, int KW, int stride, int pad, int dtype, cn_hooks* hooks, void* stream) {
    CN_CHECK_ARG(x && dy && dw && ws, "cn_conv2d_wgrad_direct: null pointer");
    const int target = cn_wgrad_target(hooks);
    const size_t need = cn_conv2d_wgrad_direct_bytes_h(N, H, W, Ci, x_ld, OH, OW, Co, dy_ld, KH, KW, stride, pad, dtype, target);
    if (need == 0) CN_UNSUPPORTED("cn_conv2d_wgrad_direct: shape not handled (cn_conv2d_wgrad_direct_bytes == 0)");
    if (ws_bytes < need) { cn_set_error("cn_conv2d_wgrad_direct: workspace too small"); return CN_EWORKSPACE; }
    CN_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15) == 0 && x_ld >= Ci && dy_ld >= Co, "cn_conv2d_wgrad_direct: bad pointers / pitches");
    hipStream_t st = (hipStream_t)stream;
    if (!wgrad3x3s1_slab_launch(x, dy, (float*)ws, dw, accumulate, N, H, W, Ci, x_ld, Co, dy_ld, stride, st, target))
        CN_UNSUPPORTED("cn_conv2d_wgrad_direct: shape not handled");
    CN_LAUNCH_CHECK("cn_conv2d_wgrad_direct");
    if (db) {
        int rc = launch_colsum(dy, db, (int64_t)N * OH * OW, Co, dy_ld, dtype, st);
        if (rc) return rc;
    }
    return CN_OK;
}
extern "C" int cn_conv2d_wgrad_direct(const void* x, const void* dy, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                                      int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                                      int KH, int KW, int stride, int pad, int dtype, void* stream) {
    return cn_conv2d_wgrad_direct_h(x, dy, dw, db, accumulate, ws, ws_bytes, N, H, W, Ci, x_ld, OH, OW, Co, dy_ld, KH, KW, stride, pad, dtype,
                                    nullptr, stream);
}

extern "C" int cn_conv2d_wgrad_h(const void* x, const void* dy, float* dwp, float* db,
                                 int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                                 int KH, int KW, int stride, int pad, int dtype, cn_hooks* hooks, void* stream) {
    const PreAffine pre = hooks_pre(hooks);      // input pre-affine of this call (cn_hooks.pre_ss)
    const int target = cn_wgrad_target(hooks);
    CN_CHECK_ARG(x && dy && dwp, "cn_conv2d_wgrad: null pointer");
    CN_CHECK_ARG(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && Co > 0 && Ci > 0, "cn_conv2d_wgrad: bad dims");
    int V = dtype == CN_F32 ? 4 : 8;
    if (Ci % V != 0) CN_UNSUPPORTED("cn_conv2d_wgrad: Ci=%d must be a multiple of %d", Ci, V);
    CN_CHECK_ARG(x_ld % V == 0 && dy_ld % V == 0 && x_ld >= Ci && dy_ld >= ((Co + V - 1) / V) * V,
                 "cn_conv2d_wgrad: pitches must be multiples of %d and cover the (vector-padded) channels: x_ld=%d dy_ld=%d Co=%d",
                 V, x_ld, dy_ld, Co);
    CN_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "cn_conv2d_wgrad: pointers must be 16-byte aligned");
    WgradGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.dy = dy; g.dwp = dwp;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.OH = OH; g.OW = OW; g.Co = Co; g.dy_ld = dy_ld;
    g.KW = KW; g.stride = stride; g.pad = pad; g.ktot = KH * KW * Ci; g.co_pad = (Co + 31) / 32 * 32;
    g.P = (int64_t)N * OH * OW; g.target = target;
    hipStream_t st = (hipStream_t)stream;
    bool done_small = false;
    if (pre.ss) CN_CHECK_ARG(pre.C == Ci, "cn_conv2d_wgrad: pre-affine given for %d channels, conv has %d", pre.C, Ci);
    if (Ci <= 16 && small_wgrad_packed(x, dy, dwp, N, H, W, Ci, x_ld, OH, OW, Co, dy_ld, KH, KW, stride, pad, dtype, st, pre.ss, pre.relu, target)) {
        CN_LAUNCH_CHECK("cn_conv2d_wgrad(small)");
        done_small = true;
    }
    if (pre.ss && !done_small)
        CN_UNSUPPORTED("cn_conv2d_wgrad: an input pre-affine is given but this shape has no kernel with the hook (bf16, 3x3 / pad 1, 16 input channels)");
    if (!done_small && dtype == CN_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == H && OW == W &&
        wgrad3x3s1_launch(x, dy, dwp, N, H, W, Ci, x_ld, Co, dy_ld, st, target)) {
        CN_LAUNCH_CHECK("cn_conv2d_wgrad(3x3)");
        done_small = true;
    }
    const bool bigm = Co > 64, bign = Ci > 64;
#define CN_WG(T) \
    do { if (bigm && bign) launch_wgrad<T, 128, 128>(g, KH * KW, st); \
         else if (bigm) launch_wgrad<T, 128, 64>(g, KH * KW, st); \
         else if (bign) launch_wgrad<T, 64, 128>(g, KH * KW, st); \
         else launch_wgrad<T, 64, 64>(g, KH * KW, st); } while (0)
    if (done_small) { /* handled by the small-channel VALU kernel */ }
    else if (dtype == CN_F32) CN_WG(float);
    else if (dtype == CN_BF16) CN_WG(bf16_t);
    else CN_CHECK_ARG(false, "cn_conv2d_wgrad: bad dtype %d", dtype);
#undef CN_WG
    CN_LAUNCH_CHECK("cn_conv2d_wgrad");
    if (db) {
        int rc = launch_colsum(dy, db, g.P, Co, dy_ld, dtype, st);
        if (rc) return rc;
    }
    return CN_OK;
}
extern "C" int cn_conv2d_wgrad(const void* x, const void* dy, float* dwp, float* db,
                               int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                               int KH, int KW, int stride, int pad, int dtype, void* stream) {
    return cn_conv2d_wgrad_h(x, dy, dwp, db, N, H, W, Ci, x_ld, OH, OW, Co, dy_ld, KH, KW, stride, pad, dtype, nullptr, stream);
}
