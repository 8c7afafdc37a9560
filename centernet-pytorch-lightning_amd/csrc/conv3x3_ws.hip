// Weight-stationary 3x3 / stride 1 / pad 1 convolution for 64 input channels (the head convs 64 -> 256, the 64 -> 64 layers at
// 128x128, the DCN offset convs 64 -> 27, and the data gradients of the 64-channel layers).  With Ci = 64 the whole weight slice of
// a 32-output-channel wave tile is 9 taps x 64 channels x 32 rows = 36 MFMA fragments = 144 VGPRs: it is loaded ONCE per workgroup
// into registers and the workgroup then walks over pixel tiles (persistent, one workgroup per CU), so that
//   * weights cost no LDS traffic and no barriers at all (the tile kernel in conv3x3.hip stages 16 KB of weights per tap through
//     LDS behind one barrier per tap: LDS-pipe-bound at ~0.36 of the MFMA peak);
//   * the LDS holds only the (16+2)x(16+2) input halo tile, double-buffered and filled by LDS-DMA (global_load_lds_dwordx4: no
//     staging registers, no ds_write issue) one tile ahead of the MFMAs; one raw s_barrier per tile, the DMA stays in flight
//     across it and is drained by the issuing wave's own vmcnt wait right before the barrier of the tile that reads it;
//   * a wave owns 64 (or 32) pixels x 32 output channels: one A-fragment ds_read_b128 per MFMA.
// LDS image of a halo tile (unpadded: the DMA writes lane-linearly): four K-step planes [kk][halo pixel q = hr * 18 + hc][32 B], the
// two 16-byte halves of a pixel's 16 channels swapped on odd halo rows.  A 32x32x16 fragment read touches, per ds_read_b128 lane
// group, 8 + 8 pixels of two consecutive halo rows, all with the same half: the 32-byte pixel pitch puts each row's eight pixels on
// the eight even (or odd) 16-byte slots of the 256-byte bank row, and the row-parity swap sends the two rows to opposite parities —
// conflict-free for every tap.  The swap is applied to the SOURCE address of the DMA and to the read address (the same involution
// on both sides).  Tap and K-step enter the read address as IMMEDIATE offsets: two address registers per 32-pixel fragment.
#include "conv_common.h"
#include <algorithm>

#define WS_HW 18
#define WS_PLANE (WS_HW * WS_HW * 32)         // bytes of one K-step plane (16 channels of every halo pixel)
#define WS_SLOTS (WS_HW * WS_HW * 8)          // 16-byte slots of one halo tile
#define WS_DMA_I ((WS_SLOTS + 63) / 64)       // wave-level DMA instructions per halo tile (41)
#define WS_HALO (WS_DMA_I * 1024)             // bytes per halo buffer
#define WS_NT 512
#define WS_SLAB (32 * 36 * 4)                 // per-wave fp32 staging slab of the epilogue: 32 pixels x (32 + 4) channels

__device__ uint4 ws_zero_page[8];             // 128 zero bytes: DMA source of the halo slots that lie outside the image

template <int BN>                             // output channels per workgroup: 64 (waves 4 x 2, 64-pixel wave tiles) or 32 (8 x 1, 32 pixels)
__global__ __launch_bounds__(WS_NT) void conv3x3_ws_kernel(const ConvGeom g, int nblk, int tiles_h, int tiles_w, uint64_t wmap) {
    constexpr int WGN = BN / 32, WGM = 8 / WGN, WM = 256 / WGM, MI = WM / 32;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * WS_HALO + 8 * WS_SLAB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * 32;
    // workgroups b, b+8, b+16 ... share an XCD (round-robin dispatch): the nblk channel blocks of one tile stream sit on one L2
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int nb = rr % nblk, stream = xcd + 8 * (rr / nblk), nstreams = gridDim.x / nblk;
    const int n0 = nb * BN;
    const int tiles_img = tiles_h * tiles_w, T = g.N * tiles_img;

    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);

    // the wave's weight fragments: rows = output channels n0 + wn + (lane & 31), k = tap * 64 + kk * 16 + 8 * (lane >> 5) .. + 7;
    // indexed by WINDOW POSITION pos = (dh + 1) * 3 + (dw + 1); wmap holds the weight tap of each position (4 bits each), so normal
    // and mirrored (data-gradient) taps run the same code with compile-time halo shifts
    bf16x8_t wr[9][4];
    {
        const int row = min(n0 + wn + (lane & 31), g.co_pad - 1);     // rows past the packed matrix: never stored
        const bf16_t* wrow = Wp + (int64_t)row * g.ktot + (lane >> 5) * 8;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wr[tap][kk] = __builtin_bit_cast(bf16x8_t, ldg16(wrow + (int)((wmap >> (4 * tap)) & 15) * 64 + kk * 16));
        // a use of every fragment in front of the tile loop: otherwise the loads are still "pending" at the loop header and the
        // compiler's vmcnt(0) before the first MFMA of EVERY tile also drains the halo DMA issued a moment earlier
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(wr[tap][kk]));
    }

    // byte offset of the top-left pixel of the lane's 3x3 window (halo pixel (row, col) of output pixel (row, col)) in plane 0, for
    // window rows of even / odd halo-row parity
    int a_ev[MI], a_od[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        const int row = m >> 4, q = row * WS_HW + (m & 15), h = lane >> 5;
        a_ev[i] = q * 32 + ((h ^ (row & 1)) << 4);       // window row ph even: halo row parity = row's
        a_od[i] = q * 32 + ((h ^ (row & 1) ^ 1) << 4);
    }

    // Per-lane values of the DMA and of the epilogue are recomputed per tile from an opaque copy of the lane id: hoisted out of the
    // tile loop they would pin ~50 registers next to the 144 of the weights and the compiler then serialises every ds_read with
    // its MFMA for want of a fragment to prefetch into.
    auto issue_halo = [&](int t, int buf) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int n = t / tiles_img, rem = t - n * tiles_img;
        const int th0 = (rem / tiles_w) * 16, tw0 = (rem % tiles_w) * 16;
        const char* base = reinterpret_cast<const char*>(X) + (((int64_t)n * g.H + th0 - 1) * g.W + (tw0 - 1)) * g.x_ld * 2;
#pragma unroll
        for (int p = 0; p < (WS_DMA_I + 7) / 8; ++p) {
            const int I = p * 8 + wave;
            if (I < WS_DMA_I) {
                const int L = I * 64 + lane;
                const int kk = L / (WS_HW * WS_HW * 2), s = L - kk * (WS_HW * WS_HW * 2), q = s >> 1;
                const int hr = (q * 3641) >> 16, hc = q - hr * WS_HW;           // q / 18 for q < 512
                const int c = kk * 2 + ((s & 1) ^ (hr & 1));
                const bool ok = kk < 4 && (unsigned)(th0 - 1 + hr) < (unsigned)g.H && (unsigned)(tw0 - 1 + hc) < (unsigned)g.W;
                const char* src = ok ? base + ((int64_t)(hr * g.W + hc) * g.x_ld + c * 8) * 2 : reinterpret_cast<const char*>(ws_zero_page) + (lane & 7) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + buf * WS_HALO + I * 1024), 16, 0, 0);
            }
        }
    };

    int t = stream;
    if (t < T) issue_halo(t, 0);
#pragma unroll 1
    for (int it = 0; t < T; t += nstreams, ++it) {
        const int buf = it & 1;
        // my share of this tile's halo has landed; past the barrier everybody's has, and everybody is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + nstreams < T) issue_halo(t + nstreams, buf ^ 1);

        f32x16_t acc[1][MI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][i][r] = 0.f;
        const unsigned char* hb = lds + buf * WS_HALO;
        // 36 * MI (window position, K step, fragment) MFMAs in groups of four; the fragments of group n + 2 are requested before the
        // MFMAs of group n issue (a ring of three register sets).  Left to itself the compiler reads every fragment into the same
        // four registers right in front of its MFMA (read, lgkmcnt(0), MFMA: the LDS latency exposed 72 times per tile).
        constexpr int NG = 9 * MI;
        bf16x8_t fr[3][4];
        auto gload = [&](bf16x8_t (&f)[4], int grp) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = grp * 4 + j, i = idx % MI, kk = (idx / MI) % 4, pos = idx / (4 * MI), ph = pos / 3, pw = pos % 3;
                f[j] = *reinterpret_cast<const bf16x8_t*>(hb + ((ph & 1) ? a_od[i] : a_ev[i]) + (ph * WS_HW + pw) * 32 + kk * WS_PLANE);
            }
        };
        gload(fr[0], 0);
        gload(fr[1], 1);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 2 < NG) gload(fr[(grp + 2) % 3], grp + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = grp * 4 + j, i = idx % MI, kk = (idx / MI) % 4, pos = idx / (4 * MI);
                acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[pos][kk], fr[grp % 3][j], acc[0][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        const int n = t / tiles_img, rem = t - n * tiles_img;
        const int th0 = (rem / tiles_w) * 16, tw0 = (rem % tiles_w) * 16;
        const int64_t img0 = (int64_t)n * g.OH * g.OW;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        if (g.epi_tile) {
            // per-wave LDS-staged epilogue: 32 pixels x 32 channels of fp32 through the wave's own slab (no workgroup barrier),
            // leaving as 16-byte vectors along the channel axis; arithmetic as in conv_epilogue_tile
            float* slab = reinterpret_cast<float*>(lds + 2 * WS_HALO + wave * WS_SLAB);
            bf16_t* __restrict__ Y = reinterpret_cast<bf16_t*>(g.y);
            const bf16_t* __restrict__ Rr = reinterpret_cast<const bf16_t*>(g.res);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(slab + (lane & 31) * 36 + 8 * q + 4 * (lane >> 5)) =
                        make_float4(acc[0][i][q * 4], acc[0][i][q * 4 + 1], acc[0][i][q * 4 + 2], acc[0][i][q * 4 + 3]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int r = ps * 16 + (lane >> 2), c8 = (lane & 3) * 8;
                    const int ch = n0 + wn + c8;
                    const float4 a = *reinterpret_cast<const float4*>(slab + r * 36 + c8), b = *reinterpret_cast<const float4*>(slab + r * 36 + c8 + 4);
                    if (ch >= g.Co) continue;
                    const int m = wm + i * 32 + r;
                    const int64_t px = img0 + (int64_t)(th0 + (m >> 4)) * g.OW + tw0 + (m & 15);
                    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    if (g.bias) {
                        const float4 b0 = *reinterpret_cast<const float4*>(g.bias + ch), b1 = *reinterpret_cast<const float4*>(g.bias + ch + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    if (Rr) {
                        float rv[8];
                        Vec16<bf16_t>::load(Rr + px * g.res_ld + ch, rv);
                        if (g.relu == 2) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = rv[e] > 0.f ? v[e] : 0.f;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += rv[e];
                        }
                    }
                    if (g.relu == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    Vec16<bf16_t>::store(Y + px * g.y_ld + ch, v);
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            int64_t pix[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = wm + i * 32 + (lane & 31);
                pix[i] = img0 + (int64_t)(th0 + (m >> 4)) * g.OW + tw0 + (m & 15);
            }
            conv_epilogue<bf16_t, MI, 1>(g, acc, pix, n0 + wn, lane);
        }
    }
}

// caller guarantees: 3x3 / stride 1 / pad 1 geometry in class 0 of g (normal or mirrored taps), OH == H, OW == W, g.epi_tile set
bool conv3x3_ws_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV_WS") != nullptr;
    static int cus = 0;
    if (disabled || dtype != CN_BF16 || g.Ci != 64 || (g.x_ld & 7) || (g.H & 15) || (g.W & 15) || g.nsrc != 0 || g.dcn_x != nullptr) return false;
    if ((reinterpret_cast<uintptr_t>(g.x) | reinterpret_cast<uintptr_t>(g.w)) & 15) return false;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) return false;
        cus = v;
    }
    uint64_t wmap = 0;
    unsigned seen = 0;
    if (g.ntaps[0] != 9) return false;
    for (int t = 0; t < 9; ++t) {
        const int ph = g.dh[0][t] + 1, pw = g.dw[0][t] + 1;
        if ((unsigned)ph > 2 || (unsigned)pw > 2) return false;
        seen |= 1u << (ph * 3 + pw);
        wmap |= (uint64_t)g.wt[0][t] << (4 * (ph * 3 + pw));
    }
    if (seen != 0x1ffu) return false;
    const int bn = g.Co <= 32 ? 32 : 64;
    const int nblk = (g.Co + bn - 1) / bn;
    const int group = 8 * nblk;
    int grid = (cus / group) * group;
    // CN_CONV_WS_FORCE=<workgroups>: tests run small problems through this kernel, several tiles per workgroup (lifts the size rule)
    const char* force = getenv("CN_CONV_WS_FORCE");
    if (force && grid) grid = std::min(grid, std::max(group, atoi(force) / group * group));
    const int tiles_h = g.H / 16, tiles_w = g.W / 16;
    const int64_t tiles = (int64_t)g.N * tiles_h * tiles_w;
    // the weights are loaded once per workgroup: only worth it when a workgroup gets a few tiles
    if (grid == 0 || (tiles * nblk < 4 * (int64_t)grid && force == nullptr)) return false;
    if (bn == 64) hipLaunchKernelGGL(conv3x3_ws_kernel<64>, dim3(grid), dim3(WS_NT), 0, st, g, nblk, tiles_h, tiles_w, wmap);
    else hipLaunchKernelGGL(conv3x3_ws_kernel<32>, dim3(grid), dim3(WS_NT), 0, st, g, nblk, tiles_h, tiles_w, wmap);
    return true;
}
