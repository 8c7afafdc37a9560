// Weight gradient of 3x3 / stride 1 / pad 1 convolutions (bf16), halo-tile form:
//   dWp[co][tap*Ci + ci] += sum_{pixels p of a tile} dY[p][co] * X[p + shift(tap)][ci]
// The generic split-K kernel walks 32-pixel K slices (4 MFMAs per wave per barrier) and re-reads X once per tap, which
// left it bound by L2->LDS latency (~250 TFLOP/s).  Here a workgroup takes an 8x16 pixel tile: the dY tile [128][BMW]
// and the X HALO tile [(8+2)x(16+2)][BNW] are staged ONCE in their natural [pixel][channel] order, and every tap of the
// tap group reads its shifted X rows straight out of the halo with the transposing LDS read (ds_read_b64_tr_b16) —
// 9x (or 3x) more MFMA work per byte staged.  The next tile is prefetched into registers while the current one is
// being multiplied; workgroups walk several tiles and flush their fp32 accumulators with one round of atomics.
#include "dcn_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define W3_TH 8
#define W3_TW 16
#define W3_HW (W3_TW + 2)
#define W3_HP ((W3_TH + 2) * W3_HW)

struct Wgrad3Geom {
    const bf16_t* x;
    const bf16_t* dy;
    float* dwp;
    int N, H, W, Ci, x_ld, Co, dy_ld, ktot;
    int tiles_h, tiles_w, ci_tiles;
    int tiles_per_block;
    int OH, OW;          // output (dY) size: H, W for stride 1; (H - 1) / 2 + 1 ... for the stride-2 variant
    int target;          // workgroups wanted over the whole launch (cn_hooks.wgrad_blocks, cn_wgrad_target)
};

// 32(channel) x 16(pixel) operand from a [row][channel] LDS tile; the 16 pixels are rows row0 .. row0+15.
__device__ static inline bf16x8_t tr_frag16(const bf16_t* tile, int pitch, int c0, int row0, int lane) {
    // (pixels `step` rows apart — the stride-2 weight gradient — are read by passing step * pitch as the pitch and row0 / step ... see tr_frag16s)
    const int r = lane & 15, g = lane >> 4;
    const bf16_t* p = tile + (row0 + 8 * (g >> 1) + (r >> 2)) * pitch + c0 + 16 * (g & 1) + 4 * (r & 3);
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * pitch));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// the same operand from 16 pixels that lie S tile rows apart, starting at row0 (stride-2 convolution: every other halo pixel)
template <int S>
__device__ static inline bf16x8_t tr_frag16s(const bf16_t* tile, int pitch, int c0, int row0, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const bf16_t* p = tile + (row0 + S * (8 * (g >> 1) + (r >> 2))) * pitch + c0 + 16 * (g & 1) + 4 * (r & 3);
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * S * pitch));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// TAPS = 9: one workgroup accumulates all taps;  TAPS = 3: blockIdx.z selects the tap row kh.
// SLAB: instead of fp32 atomics into the packed gradient, every workgroup stores its accumulators as ONE private slab (fragment
// order: a wave's store instruction is 256 contiguous bytes) and wgrad3x3_reduce_kernel sums the slabs of a (channel tile, tap
// row) straight into the parameter-layout gradient.  Measured (tools/wgrad_bench.py, 64->64 @128^2, batch 64): the atomic flush
// of the 147 KB accumulator tile costs 0.25 us PER WORKGROUP, serialised chip-wide (device-scope fp32 atomics execute at the
// memory side, ~600 GB/s) — 65 us of a 135 us launch with 256 workgroups, 98 of 166 us with the 384 the train step uses.
// NW = 8 (512 threads, 4 x 2 waves): all nine taps of a 128 x 64 channel tile with 144 accumulator registers per wave — dY is then
// read ONCE per launch instead of once per tap row (the 64 -> 256 head convs: 2.4 GB -> 0.8 GB of operand traffic per launch).
// S = 2: the stride-2 3x3 / pad 1 convs (the first conv of every DLA / ResNet stage): 4x16 output tiles, a 9x33 halo, every other halo
// pixel per operand (tr_frag16s) — these ran on the generic split-K kernel, one workgroup column per TAP (x re-read nine times,
// two MFMAs per wave and barrier): 32->64 @256^2 886 us, 64->128 284, 128->256 163, 256->512 168 on the step's thin grids.
template <int BMW, int BNW, int TAPS, bool SLAB = false, int NW = 4, int S = 1>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void wgrad3x3s1_kernel(const Wgrad3Geom g) {
    constexpr int NT = NW * 64;
    constexpr int TH = S == 1 ? W3_TH : 4;                    // output rows per tile
    constexpr int HW = (W3_TW - 1) * S + 3;                   // halo columns
    constexpr int YP = BMW + 32, XP = BNW + 32;               // +64 B: the 4 rows of a transposing read hit disjoint banks
    constexpr int YCG = BMW / 8, XCG = BNW / 8;
    constexpr int HROWS = TAPS == 9 ? (TH - 1) * S + 3 : (TH - 1) * S + 1;      // halo rows staged
    constexpr int YV = (TH * W3_TW * YCG + NT - 1) / NT;      // 16-byte loads per thread
    constexpr int XV = (HROWS * HW * XCG + NT - 1) / NT;
    constexpr int WM = BMW / (NW / 2), WN = BNW / 2;          // (NW / 2) x 2 waves
    constexpr int MI = WM / 32, NJ = WN / 32;

    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // [128][YP] dY tile | [HROWS*18][XP] X halo
    bf16_t* const yt = lds;
    bf16_t* const xt = lds + TH * W3_TW * YP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = (blockIdx.y / g.ci_tiles) * BMW, ci0 = (blockIdx.y % g.ci_tiles) * BNW;
    const int kh0 = TAPS == 9 ? 0 : (int)blockIdx.z;          // first tap row handled here
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int64_t t_beg = (int64_t)blockIdx.x * g.tiles_per_block;
    const int64_t t_end = t_beg + g.tiles_per_block < ntiles ? t_beg + g.tiles_per_block : ntiles;
    if (t_beg >= ntiles) return;

    f32x16_t acc[TAPS][MI][NJ];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    uint4 ry[YV], rx[XV];
    auto gload = [&](int64_t tile) {
        const int n = (int)(tile / (g.tiles_h * g.tiles_w));
        const int r = (int)(tile - (int64_t)n * g.tiles_h * g.tiles_w);
        const int th0 = (r / g.tiles_w) * TH, tw0 = (r % g.tiles_w) * W3_TW;
        const int64_t img = (int64_t)n * g.H * g.W, oimg = (int64_t)n * g.OH * g.OW;
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * NT;
            const int px = idx / YCG, c = co0 + (idx % YCG) * 8;
            const int oh = th0 + px / W3_TW, ow = tw0 + px % W3_TW;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (idx < TH * W3_TW * YCG && oh < g.OH && ow < g.OW && c < g.Co)
                val = *reinterpret_cast<const uint4*>(g.dy + (oimg + (int64_t)oh * g.OW + ow) * g.dy_ld + c);
            ry[v] = val;
        }
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int idx = tid + v * NT;
            const int hp = idx / XCG, c = ci0 + (idx % XCG) * 8;
            const int hr = hp / HW;                            // staged halo row: image row th0 * S - 1 + kh0 + hr
            const int ih = th0 * S - 1 + kh0 + hr, iw = tw0 * S - 1 + hp % HW;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (idx < HROWS * HW * XCG && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W && c < g.Ci)
                val = *reinterpret_cast<const uint4*>(g.x + (img + (int64_t)ih * g.W + iw) * g.x_ld + c);
            rx[v] = val;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * NT;
            if (idx < TH * W3_TW * YCG) *reinterpret_cast<uint4*>(yt + (idx / YCG) * YP + (idx % YCG) * 8) = ry[v];
        }
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int idx = tid + v * NT;
            if (idx < HROWS * HW * XCG) *reinterpret_cast<uint4*>(xt + (idx / XCG) * XP + (idx % XCG) * 8) = rx[v];
        }
    };

    gload(t_beg);
    for (int64_t tile = t_beg; tile < t_end; ++tile) {
        __syncthreads();                      // every wave is done reading the previous tile
        lstore();
        __syncthreads();
        if (tile + 1 < t_end) gload(tile + 1);   // in flight while this tile is multiplied
#pragma unroll
        for (int kk = 0; kk < TH; ++kk) {     // one k16 step = one 16-pixel tile row
            bf16x8_t fa[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = tr_frag16(yt, YP, wm + i * 32, kk * W3_TW, lane);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int kh = t / 3, kw = t % 3;             // TAPS == 3: kh is relative to kh0 (the halo starts at that row)
                bf16x8_t fb[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = tr_frag16s<S>(xt, XP, wn + j * 32, (kk * S + kh) * HW + kw, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[t][i][j], 0, 0, 0);
            }
        }
    }

    if constexpr (SLAB) {
        constexpr int PER_WAVE = TAPS * MI * NJ * 16 * 64;
        float* sl = g.dwp + (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (NW * PER_WAVE) + wave * PER_WAVE + lane;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sl[(((t * MI + i) * NJ + j) * 16 + r) * 64] = acc[t][i][j][r];
        return;
    }
    // D rows = co: (r&3) + 8*(r>>2) + 4*(lane>>5); D col = ci: lane&31
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int tap = TAPS == 9 ? t : kh0 * 3 + t;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ci = ci0 + wn + j * 32 + (lane & 31);
                if (ci >= g.Ci) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (co < g.Co) atomicAdd(g.dwp + (int64_t)co * g.ktot + (int64_t)tap * g.Ci + ci, acc[t][i][j][r]);
                }
            }
    }
}

// Sum of the k-split slabs of wgrad3x3s1_kernel<.., SLAB = true> into the PARAMETER layout dw[Co][Ci][3][3] (fp32; accumulate != 0
// adds, e.g. straight into the flat gradient buffer).  One workgroup per 64-float slab row (= one accumulator register of one
// wave): 64 lanes x 4 groups of k-splits, the groups meet in LDS.  Deterministic (fixed summation order).
template <int BMW, int BNW, int TAPS, int NW>
__global__ __launch_bounds__(256) void wgrad3x3_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int gx, int ci_tiles,
                                                              int Co, int Ci, int accumulate) {
    constexpr int WM = BMW / (NW / 2), WN = BNW / 2, MI = WM / 32, NJ = WN / 32;
    constexpr int ROWS_PER_WAVE = TAPS * MI * NJ * 16, TILE = NW * ROWS_PER_WAVE * 64;
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int row = blockIdx.x;                                     // slab row: (wave, t, i, j, r)
    const int64_t part = (int64_t)blockIdx.y * gridDim.z + blockIdx.z;
    const int64_t nparts = (int64_t)gridDim.y * gridDim.z;
    const float* p = slabs + part * TILE + (int64_t)row * 64 + lane;
    float v = 0.f;
    int s = sg;
    for (; s + 12 < gx; s += 16) {                                  // four independent loads in flight
        const float a = p[(int64_t)s * nparts * TILE], b = p[(int64_t)(s + 4) * nparts * TILE];
        const float c = p[(int64_t)(s + 8) * nparts * TILE], d = p[(int64_t)(s + 12) * nparts * TILE];
        v += (a + b) + (c + d);
    }
    for (; s < gx; s += 4) v += p[(int64_t)s * nparts * TILE];
    red[sg][lane] = v;
    __syncthreads();
    if (sg != 0) return;
    v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const int wave = row / ROWS_PER_WAVE, rem = row % ROWS_PER_WAVE;
    const int r = rem & 15, j = (rem >> 4) % NJ, i = (rem >> 4) / NJ % MI, t = (rem >> 4) / (NJ * MI);
    const int co0 = ((int)blockIdx.y / ci_tiles) * BMW, ci0 = ((int)blockIdx.y % ci_tiles) * BNW;
    const int co = co0 + (wave >> 1) * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int ci = ci0 + (wave & 1) * WN + j * 32 + (lane & 31);
    const int tap = TAPS == 9 ? t : (int)blockIdx.z * 3 + t;
    if (co < Co && ci < Ci) {
        float* d = dw + ((int64_t)co * Ci + ci) * 9 + tap;
        *d = accumulate ? *d + v : v;
    }
}

template <int BMW, int BNW, int TAPS>
static size_t w3_slab_plan(Wgrad3Geom& g) {              // fills the tiling fields; -> bytes of slab workspace
    const int co_tiles = cdiv(g.Co, BMW);
    g.ci_tiles = cdiv(g.Ci, BNW);
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int par = co_tiles * g.ci_tiles * (TAPS == 9 ? 1 : 3);
    // the 8-wave all-taps tile runs one workgroup per CU and its slab is 295 KB: one round of workgroups (64 -> 256 @128^2: 298 us
    // with 256 workgroups, 368 with 384)
    static const int env_blocks = getenv("CN_WGRAD3X3_BLOCKS") ? atoi(getenv("CN_WGRAD3X3_BLOCKS")) : 0;            // A/B: this family's own grid
    const int base_target = env_blocks > 0 ? env_blocks : g.target;
    const int target = (BMW == 128 && TAPS == 9 && base_target > 256) ? 256 : base_target;
    int64_t want = (target + par - 1) / par;
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    g.tiles_per_block = (int)((ntiles + want - 1) / want);
    const int64_t gx = (ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    return (size_t)gx * par * TAPS * BMW * BNW * sizeof(float);
}

template <int BMW, int BNW, int TAPS, int NW = 4, int S = 1>
static void launch_w3_slab(Wgrad3Geom& g, float* dw, int accumulate, hipStream_t st) {
    (void)w3_slab_plan<BMW, BNW, TAPS>(g);
    const int co_tiles = cdiv(g.Co, BMW);
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int gx = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    constexpr int TH = S == 1 ? W3_TH : 4, HW = (W3_TW - 1) * S + 3;
    const int hrows = TAPS == 9 ? (TH - 1) * S + 3 : (TH - 1) * S + 1;
    const size_t smem = ((size_t)TH * W3_TW * (BMW + 32) + (size_t)hrows * HW * (BNW + 32)) * sizeof(bf16_t);
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)wgrad3x3s1_kernel<BMW, BNW, TAPS, true, NW, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const dim3 grid(gx, co_tiles * g.ci_tiles, TAPS == 9 ? 1 : 3);
    hipLaunchKernelGGL((wgrad3x3s1_kernel<BMW, BNW, TAPS, true, NW, S>), grid, dim3(NW * 64), smem, st, g);
    hipLaunchKernelGGL((wgrad3x3_reduce_kernel<BMW, BNW, TAPS, NW>), dim3(TAPS * BMW * BNW / 64, grid.y, grid.z), dim3(256), 0, st, g.dwp, dw, gx,
                       g.ci_tiles, g.Co, g.Ci, accumulate);
}

template <int BMW, int BNW, int TAPS>
static void launch_w3(Wgrad3Geom& g, hipStream_t st) {
    const int co_tiles = cdiv(g.Co, BMW);
    g.ci_tiles = cdiv(g.Ci, BNW);
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int par = co_tiles * g.ci_tiles * (TAPS == 9 ? 1 : 3);
    int64_t want = (g.target + par - 1) / par;   // workgroups over the whole launch (cn_hooks.wgrad_blocks)
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    g.tiles_per_block = (int)((ntiles + want - 1) / want);
    const int gx = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    const int hrows = TAPS == 9 ? W3_TH + 2 : W3_TH;
    const size_t smem = ((size_t)W3_TH * W3_TW * (BMW + 32) + (size_t)hrows * W3_HW * (BNW + 32)) * sizeof(bf16_t);
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)wgrad3x3s1_kernel<BMW, BNW, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((wgrad3x3s1_kernel<BMW, BNW, TAPS>), dim3(gx, co_tiles * g.ci_tiles, TAPS == 9 ? 1 : 3), dim3(256), smem, st, g);
}

// bf16, 3x3 / stride 1 / pad 1, Ci > 16.  Returns false if the shape is not handled here.
bool wgrad3x3s1_launch(const void* x, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                       hipStream_t st, int target) {
    static const bool disabled = getenv("CN_DISABLE_WGRAD3X3") != nullptr;
    if (disabled || Ci % 8 != 0 || x_ld % 8 != 0 || dy_ld % 8 != 0) return false;
    Wgrad3Geom g;
    g.x = (const bf16_t*)x; g.dy = (const bf16_t*)dy; g.dwp = dwp;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.Co = Co; g.dy_ld = dy_ld; g.ktot = 9 * Ci;
    g.OH = H; g.OW = W; g.target = target;
    g.tiles_h = cdiv(H, W3_TH); g.tiles_w = cdiv(W, W3_TW);
    if (Co > 64) launch_w3<128, 64, 3>(g, st);      // 68.6 KB LDS, 96 accumulator registers
    else launch_w3<64, 64, 9>(g, st);               // 59 KB LDS, 144 accumulator registers
    return true;
}


// Slab form of the above: `slabs` is scratch (wgrad3x3s1_slab_bytes), the result lands in the parameter-layout gradient dw.
// stride 1 (OH = H) or 2 (OH = (H - 1) / 2 + 1), pad 1.
static void w3_tiles(Wgrad3Geom& g, int stride) {
    g.OH = stride == 1 ? g.H : (g.H - 1) / 2 + 1;
    g.OW = stride == 1 ? g.W : (g.W - 1) / 2 + 1;
    g.tiles_h = cdiv(g.OH, stride == 1 ? W3_TH : 4); g.tiles_w = cdiv(g.OW, W3_TW);
}
size_t wgrad3x3s1_slab_bytes(int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld, int stride, int target) {
    static const bool disabled = getenv("CN_DISABLE_WGRAD3X3") != nullptr || getenv("CN_DISABLE_WGRAD_SLABS") != nullptr;
    static const bool no_s2 = getenv("CN_DISABLE_WGRAD3X3_S2") != nullptr;
    if (disabled || Ci % 8 != 0 || x_ld % 8 != 0 || dy_ld % 8 != 0 || (stride != 1 && stride != 2) || (stride == 2 && no_s2)) return 0;
    Wgrad3Geom g;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.Co = Co; g.target = target;
    w3_tiles(g, stride);
    static const bool taps9 = getenv("CN_WGRAD3X3_WIDE_TAPS3") == nullptr;      // A/B: the 4-wave 128 x 64 tile with one tap row per workgroup
    return Co > 64 ? ((taps9 || stride == 2) ? w3_slab_plan<128, 64, 9>(g) : w3_slab_plan<128, 64, 3>(g)) : w3_slab_plan<64, 64, 9>(g);
}

bool wgrad3x3s1_slab_launch(const void* x, const void* dy, float* slabs, float* dw, int accumulate, int N, int H, int W, int Ci, int x_ld,
                            int Co, int dy_ld, int stride, hipStream_t st, int target) {
    if (wgrad3x3s1_slab_bytes(N, H, W, Ci, x_ld, Co, dy_ld, stride, target) == 0) return false;
    Wgrad3Geom g;
    g.x = (const bf16_t*)x; g.dy = (const bf16_t*)dy; g.dwp = slabs;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.Co = Co; g.dy_ld = dy_ld; g.ktot = 9 * Ci; g.target = target;
    w3_tiles(g, stride);
    static const bool taps9 = getenv("CN_WGRAD3X3_WIDE_TAPS3") == nullptr;
    if (stride == 2) {
        if (Co > 64) launch_w3_slab<128, 64, 9, 8, 2>(g, dw, accumulate, st);
        else launch_w3_slab<64, 64, 9, 4, 2>(g, dw, accumulate, st);
    } else if (Co > 64 && taps9) launch_w3_slab<128, 64, 9, 8>(g, dw, accumulate, st);     // 8 waves, all taps: dY read once
    else if (Co > 64) launch_w3_slab<128, 64, 3>(g, dw, accumulate, st);
    else launch_w3_slab<64, 64, 9>(g, dw, accumulate, st);
    return true;
}


// ================================================================================================ DCNv2 weight gradient
// dWp[co][tap*Ci + ci] += sum_p dY[p][co] * (mask[p,tap] * bilinear(x[:, :, ci], pos(p,tap)))
// Same halo-free tile scheme: the dY tile [128][BMW] is staged once per pixel tile; for every tap the sampled operand
// [128][BNW] is rebuilt in LDS (geometry by 128 lanes, then four unconditional 16-byte corner loads per item) and
// consumed with the transposing LDS read.  The 9x-wide column tensor is never materialised (nor kept from the forward).
struct DcnWgradGeom {
    const bf16_t* x;
    const bf16_t* dy;
    const float* om;
    float* dwp;
    int N, H, W, Ci, x_ld, Co, dy_ld, om_ld, ktot;
    int tiles_h, tiles_w, ci_tiles;
    int tiles_per_block;
    int target;          // workgroups wanted over the whole launch (cn_hooks.wgrad_blocks)
};

template <int BMW, int BNW, int TAPS>
__global__ __launch_bounds__(256) void dcn_wgrad_kernel(const DcnWgradGeom g) {
    constexpr int YP = BMW + 32, XP = BNW + 32;
    constexpr int YCG = BMW / 8, XCG = BNW / 8;
    constexpr int BMP = W3_TH * W3_TW;                        // 128 pixels
    constexpr int YV = (BMP * YCG + 255) / 256;
    constexpr int XI = (BMP * XCG + 255) / 256;               // sampling items per thread
    constexpr int WM = BMW / 2, WN = BNW / 2;
    constexpr int MI = WM / 32, NJ = WN / 32;

    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // [128][YP] dY | [128][XP] sampled x
    __shared__ int s_idx[4][BMP];
    __shared__ float s_w[4][BMP];
    bf16_t* const yt = lds;
    bf16_t* const xt = lds + BMP * YP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = (blockIdx.y / g.ci_tiles) * BMW, ci0 = (blockIdx.y % g.ci_tiles) * BNW;
    const int tap0 = TAPS == 9 ? 0 : (int)blockIdx.z * 3;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int64_t t_beg = (int64_t)blockIdx.x * g.tiles_per_block;
    const int64_t t_end = t_beg + g.tiles_per_block < ntiles ? t_beg + g.tiles_per_block : ntiles;
    if (t_beg >= ntiles) return;

    f32x16_t acc[TAPS][MI][NJ];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    uint4 ry[YV];
    auto yload = [&](int64_t tile) {
        const int n = (int)(tile / (g.tiles_h * g.tiles_w));
        const int r = (int)(tile - (int64_t)n * g.tiles_h * g.tiles_w);
        const int th0 = (r / g.tiles_w) * W3_TH, tw0 = (r % g.tiles_w) * W3_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * 256;
            const int px = idx / YCG, c = co0 + (idx % YCG) * 8;
            const int oh = th0 + px / W3_TW, ow = tw0 + px % W3_TW;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (idx < BMP * YCG && oh < g.H && ow < g.W && c < g.Co)
                val = *reinterpret_cast<const uint4*>(g.dy + (img + (int64_t)oh * g.W + ow) * g.dy_ld + c);
            ry[v] = val;
        }
    };

    yload(t_beg);
    for (int64_t tile = t_beg; tile < t_end; ++tile) {
        const int n = (int)(tile / (g.tiles_h * g.tiles_w));
        const int rr = (int)(tile - (int64_t)n * g.tiles_h * g.tiles_w);
        const int th0 = (rr / g.tiles_w) * W3_TH, tw0 = (rr % g.tiles_w) * W3_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
        const bf16_t* __restrict__ X = g.x + img * g.x_ld + ci0;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < YV; ++v) {
            const int idx = tid + v * 256;
            if (idx < BMP * YCG) *reinterpret_cast<uint4*>(yt + (idx / YCG) * YP + (idx % YCG) * 8) = ry[v];
        }
        // offsets / mask logit are fetched one tap ahead
        const int gh_ = th0 + (tid & (BMP - 1)) / W3_TW, gw_ = tw0 + (tid & (BMP - 1)) % W3_TW;
        const bool glive = tid < BMP && gh_ < g.H && gw_ < g.W;
        const float* const orow = g.om + (img + (int64_t)(glive ? gh_ : 0) * g.W + (glive ? gw_ : 0)) * g.om_ld;
        float ro[3] = {0.f, 0.f, 0.f};
        if (glive) { ro[0] = orow[2 * tap0]; ro[1] = orow[2 * tap0 + 1]; ro[2] = orow[18 + tap0]; }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int tap = tap0 + t;
            if (tid < BMP) {
                const int h = gh_, w = gw_;
                int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
                float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
                if (glive) {
                    const float py = (float)(h - 1 + tap / 3) + ro[0];
                    const float px = (float)(w - 1 + tap % 3) + ro[1];
                    const float m = sigmoidf_(ro[2]);
                    const Tap tp = make_tap(py, px, g.H, g.W);
                    const int hc0 = min(max(tp.h0, 0), g.H - 1), hc1 = min(max(tp.h0 + 1, 0), g.H - 1);
                    const int wc0 = min(max(tp.w0, 0), g.W - 1), wc1 = min(max(tp.w0 + 1, 0), g.W - 1);
                    i0 = hc0 * g.W + wc0; i1 = hc0 * g.W + wc1; i2 = hc1 * g.W + wc0; i3 = hc1 * g.W + wc1;
                    w0 = tp.w00 * m; w1 = tp.w01 * m; w2 = tp.w10 * m; w3 = tp.w11 * m;
                }
                s_idx[0][tid] = i0; s_idx[1][tid] = i1; s_idx[2][tid] = i2; s_idx[3][tid] = i3;
                s_w[0][tid] = w0; s_w[1][tid] = w1; s_w[2][tid] = w2; s_w[3][tid] = w3;
            }
            __syncthreads();                               // geometry (and, for t == 0, the dY tile) visible
            if (glive && t + 1 < TAPS) { ro[0] = orow[2 * (tap + 1)]; ro[1] = orow[2 * (tap + 1) + 1]; ro[2] = orow[18 + tap + 1]; }
#pragma unroll
            for (int ip = 0; ip < XI; ++ip) {
                const int it = tid + ip * 256;
                if (it < BMP * XCG) {
                    const int pl = it / XCG, col = (it % XCG) * 8;
                    float v0[8], v1[8], v2[8], v3[8], a[8];
                    const bool cok = ci0 + col < g.Ci;
                    const bf16_t* src = X + (cok ? col : 0);
                    Vec16<bf16_t>::load(src + (int64_t)s_idx[0][pl] * g.x_ld, v0);
                    Vec16<bf16_t>::load(src + (int64_t)s_idx[1][pl] * g.x_ld, v1);
                    Vec16<bf16_t>::load(src + (int64_t)s_idx[2][pl] * g.x_ld, v2);
                    Vec16<bf16_t>::load(src + (int64_t)s_idx[3][pl] * g.x_ld, v3);
                    const float w0 = cok ? s_w[0][pl] : 0.f, w1 = cok ? s_w[1][pl] : 0.f, w2 = cok ? s_w[2][pl] : 0.f, w3 = cok ? s_w[3][pl] : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] = v0[j] * w0 + v1[j] * w1 + v2[j] * w2 + v3[j] * w3;
                    Vec16<bf16_t>::store(xt + pl * XP + col, a);
                }
            }
            __syncthreads();
            if (t == 0 && tile + 1 < t_end) yload(tile + 1);   // next dY tile in flight during this tile's MFMAs
#pragma unroll
            for (int kk = 0; kk < W3_TH; ++kk) {
                bf16x8_t fa[MI], fb[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = tr_frag16(yt, YP, wm + i * 32, kk * W3_TW, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = tr_frag16(xt, XP, wn + j * 32, kk * W3_TW, lane);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[t][i][j], 0, 0, 0);
            }
            __syncthreads();                               // xt / geometry may be overwritten by the next tap
        }
    }

#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int tap = tap0 + t;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ci = ci0 + wn + j * 32 + (lane & 31);
                if (ci >= g.Ci) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (co < g.Co) atomicAdd(g.dwp + (int64_t)co * g.ktot + (int64_t)tap * g.Ci + ci, acc[t][i][j][r]);
                }
            }
    }
}

template <int BMW, int BNW, int TAPS>
static void launch_dw(DcnWgradGeom& g, hipStream_t st) {
    const int co_tiles = cdiv(g.Co, BMW);
    g.ci_tiles = cdiv(g.Ci, BNW);
    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int par = co_tiles * g.ci_tiles * (TAPS == 9 ? 1 : 3);
    int64_t want = (g.target + par - 1) / par;
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    g.tiles_per_block = (int)((ntiles + want - 1) / want);
    const int gx = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    const size_t smem = ((size_t)W3_TH * W3_TW * (BMW + 32) + (size_t)W3_TH * W3_TW * (BNW + 32)) * sizeof(bf16_t);
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)dcn_wgrad_kernel<BMW, BNW, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((dcn_wgrad_kernel<BMW, BNW, TAPS>), dim3(gx, co_tiles * g.ci_tiles, TAPS == 9 ? 1 : 3), dim3(256), smem, st, g);
}

// dcn_bm.hip: blend-matrix sampler, one wave per tap (64 -> 64 layers); false = shape not handled there
bool dcn_wgrad_bm_launch(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                         int om_ld, int target_blocks, hipStream_t st);

extern "C" int cn_dcn_wgrad_h(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld,
                              int Co, int dy_ld, int om_ld, int dtype, cn_hooks* hooks, void* stream) {
    CN_CHECK_ARG(x && om && dy && dwp && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, "cn_dcn_wgrad: bad args");
    if (dtype != CN_BF16) CN_UNSUPPORTED("cn_dcn_wgrad: bf16 only (fp32 parity mode goes through cn_dcn_im2col + cn_conv2d_wgrad)");
    if (Ci % 8 != 0 || x_ld % 8 != 0 || dy_ld % 8 != 0) CN_UNSUPPORTED("cn_dcn_wgrad: channel counts must be multiples of 8");
    static const int env_blocks = getenv("CN_DCN_WGRAD_BLOCKS") ? atoi(getenv("CN_DCN_WGRAD_BLOCKS")) : 0;      // A/B: this family's own grid
    const int target = env_blocks > 0 ? env_blocks : cn_wgrad_target(hooks);
    if (dcn_wgrad_bm_launch(x, om, dy, dwp, N, H, W, Ci, x_ld, Co, dy_ld, om_ld, target, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_dcn_wgrad(bm)");
        return CN_OK;
    }
    DcnWgradGeom g;
    g.x = (const bf16_t*)x; g.dy = (const bf16_t*)dy; g.om = om; g.dwp = dwp;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.Co = Co; g.dy_ld = dy_ld; g.om_ld = om_ld; g.ktot = 9 * Ci;
    g.target = cn_wgrad_target(hooks);
    g.tiles_h = cdiv(H, W3_TH); g.tiles_w = cdiv(W, W3_TW);
    // Co <= 64: all nine taps per workgroup (dY tile read once).  CN_DCN_WGRAD_TAPS3 = three taps per workgroup (blockIdx.z =
    // kernel row; 48 instead of 144 accumulator registers, three workgroups per CU): 12-28 % faster in isolation (545 -> 480 us on
    // 64->64 @128^2), but no faster inside the step, where this kernel runs with a background-shaped grid next to the
    // data-gradient chain (interleaved bench runs: 50.93 vs 50.85 ms) — kept as a switch.
    static const bool taps3 = getenv("CN_DCN_WGRAD_TAPS3") != nullptr;
    if (Co > 64) launch_dw<128, 64, 3>(g, (hipStream_t)stream);
    else if (taps3) launch_dw<64, 64, 3>(g, (hipStream_t)stream);
    else launch_dw<64, 64, 9>(g, (hipStream_t)stream);
    CN_LAUNCH_CHECK("cn_dcn_wgrad");
    return CN_OK;
}
extern "C" int cn_dcn_wgrad(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld,
                            int Co, int dy_ld, int om_ld, int dtype, void* stream) {
    return cn_dcn_wgrad_h(x, om, dy, dwp, N, H, W, Ci, x_ld, Co, dy_ld, om_ld, dtype, nullptr, stream);
}
