// Weight-stationary 3x3 / stride 1 / pad 1 convolution for 64 input channels (the head convs 64 -> 256, the 64 -> 64 layers at
// 128x128, the DCN offset convs 64 -> 27, and the data gradients of the 64-channel layers).  With Ci = 64 the whole weight slice of
// a 32-output-channel wave tile is 9 taps x 64 channels x 32 rows = 36 MFMA fragments = 144 VGPRs: it is loaded ONCE per workgroup
// into registers and the workgroup then walks over 16x16-pixel tiles (persistent, one workgroup of 8 waves per CU), so that
//   * weights cost no LDS traffic and no barriers at all (the tile kernel in conv3x3.hip stages 16 KB of weights per tap through
//     LDS behind one barrier per tap: LDS-pipe-bound at ~0.36 of the MFMA peak);
//   * the LDS holds only the (16+2)x(16+2) input halo tiles, three of them in rotation, filled by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write issue) two tiles ahead of the MFMAs;
//   * a wave owns 64 (or 32) pixels x 32 output channels: one A-fragment ds_read_b128 per MFMA, requested two MFMA groups ahead;
//     the epilogue goes straight from the accumulators to global memory;
//   * the two waves of a SIMD run HALF A TILE APART: waves 0-3 do their MFMAs while waves 4-7 run the epilogue of the previous
//     tile and issue their share of the DMA, then the roles swap (two raw s_barriers per tile).  A wave issues one VALU
//     instruction every four cycles, so the few hundred instructions of epilogue + DMA addressing are as long as a tile's MFMAs:
//     with all eight waves in lockstep (a first version) the matrix pipe idled through them and the kernel ran at 0.27 of peak.
// LDS image of a halo tile (unpadded: the DMA writes lane-linearly): pixel-major, 128 B per halo pixel q = hr * 18 + hc (eight
// 16-byte slots = 64 bf16 channels), slot c stored at position c ^ ((hc >> 1) & 7).  Eight consecutive lanes of a DMA instruction
// fetch one pixel's whole 128-byte line (a K-step-major image needed 32 lines per instruction for the same 1 KB: the address
// path, not the LDS, then bounded the DMA).  A 32x32x16 fragment read touches, per ds_read_b128 lane group, 8 + 8 pixels of two
// halo rows whose hc are 16 consecutive integers, all with the same slot c; the halo row pitch 18 x 128 B is a multiple of the
// 256-byte bank row, so (hc & 1, c ^ swizzle) covers the 16 slot positions of a bank row exactly once: conflict-free for every
// tap.  The permutation is applied to the SOURCE address of the DMA and to the read address (the same involution on both sides).
// Read address = A[fragment][window column] ^ (kk << 5) + immediate(window row, column): the K step only flips bits 5-6, which the
// swizzle owns, so one XOR per read replaces the address arithmetic; the buffer rotation is added to the six A registers per tile.
#include "conv_common.h"
#include <algorithm>

#define WS_HW 18
#define WS_SLOTS (WS_HW * WS_HW * 8)          // 16-byte slots of one halo tile
#define WS_DMA_I ((WS_SLOTS + 63) / 64)       // wave-level DMA instructions per halo tile (41)
#define WS_DMA_P ((WS_DMA_I + 7) / 8)         // ... per wave (6, the last one on wave 0 only)
#define WS_HALO (WS_DMA_I * 1024)             // bytes per halo buffer
#define WS_NT 512
#define WS_LDS (3 * WS_HALO + 64 * 4 + 2 * 64 * 4)     // + bias + the fused head's two weight rows of this channel block

__device__ uint4 ws_zero_page[8];             // 128 zero bytes: DMA source of the halo slots that lie outside the image
#ifdef WS_PROBE   // development build only (tools/ws_probe.py): per-phase cycle stamps of waves 0 and 4 of every workgroup, first 8 tiles
__device__ unsigned long long ws_ts[512 * 8 * 8];
#define WS_STAMP(k) do { if ((wave & 3) == 0 && it < 8 && (tid & 63) == 0) ws_ts[((blockIdx.x * 2 + (wave >> 2)) * 8 + it) * 8 + (k)] = clock64(); } while (0)
extern "C" int ws_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ws_ts), sizeof(ws_ts)); }
#else
#define WS_STAMP(k) do { } while (0)
#endif

// BN: output channels per workgroup, 64 (waves 4 x 2, 64-pixel wave tiles) or 32 (8 x 1, 32 pixels); YF32: fp32 output rows;
// RES: 0 no residual, 1 y += res, 2 ReLU-backward mask y = res > 0 ? y : 0; RELU: y = max(y, 0) (compile-time: the epilogue
// is branch-free)
// HEAD = 2 (round 4, cn_head2_fwd): a 2-channel task head in ONE launch (heads.py:9-15: conv3x3 -> ReLU -> conv1x1): the hidden
// activation never leaves the accumulators — each wave multiplies its 32 hidden channels with the head's two weight rows, the two halves
// of the wave meet through a lane swap and add one output channel each to the public fp32 NCHW map with fp32 atomics (8 partial sums per
// pixel and channel: 256 hidden channels / 32).  No 537 MB hidden tensor, no second launch.
template <int BN, bool YF32, int RES, bool RELU, int HEAD = 0>
__global__ __launch_bounds__(WS_NT) void conv3x3_ws_kernel(const ConvGeom g, int nblk, int tiles_h, int tiles_w, uint64_t wmap) {
    CN_MAIN_PRIO_SET();
    constexpr int WGN = BN / 32, WGM = 8 / WGN, WM = 256 / WGM, MI = WM / 32;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[WS_LDS];
    float* const bias_l = reinterpret_cast<float*>(lds + 3 * WS_HALO);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * 32;
    // workgroups b, b+8, b+16 ... share an XCD (round-robin dispatch): the nblk channel blocks of one tile stream sit on one L2
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int nb = rr % nblk, stream = xcd + 8 * (rr / nblk), nstreams = gridDim.x / nblk;
    const int n0 = nb * BN;
    const int tiles_img = tiles_h * tiles_w, T = g.N * tiles_img;

    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);

    if (tid < BN) bias_l[tid] = (g.bias && n0 + tid < g.Co) ? g.bias[n0 + tid] : 0.f;
    float* const head_l = bias_l + 64;        // [2][64]: the fused head's weight rows over this workgroup's hidden channels
    if constexpr (HEAD == 2) {
        if (tid < 2 * BN) head_l[(tid / BN) * 64 + tid % BN] = n0 + tid % BN < g.Co ? g.head_w[(int64_t)(tid / BN) * g.Co + n0 + tid % BN] : 0.f;
    }

    // the wave's weight fragments: rows = output channels n0 + wn + (lane & 31), k = tap * 64 + kk * 16 + 8 * (lane >> 5) .. + 7;
    // indexed by WINDOW POSITION pos = (dh + 1) * 3 + (dw + 1); wmap holds the weight tap of each position (4 bits each), so normal
    // and mirrored (data-gradient) taps run the same code with compile-time halo shifts
    bf16x8_t wr[9][4];
    // per-lane constants of the halo DMA: byte offset of the lane's 16-byte piece relative to the halo's top-left pixel, and five
    // flag bits per instruction (piece on halo row 0 / 17, column 0 / 17 — outside the image when the tile touches that border —
    // or no piece at all): the per-tile work is then one AND + compare + two selects + a 64-bit add per instruction
    int doff[WS_DMA_P];
    unsigned ring = 0;
    int ad[MI][3];                            // read address (K step 0, window row 0) of fragment i at window column pw, current buffer
    {
        const int lane = tid & 63;
        const int row = min(n0 + wn + (lane & 31), g.co_pad - 1);     // rows past the packed matrix: never stored
        const bf16_t* wrow = Wp + (int64_t)row * g.ktot + (lane >> 5) * 8;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wr[tap][kk] = __builtin_bit_cast(bf16x8_t, ldg16(wrow + (int)((wmap >> (4 * tap)) & 15) * 64 + kk * 16));
#pragma unroll
        for (int p = 0; p < WS_DMA_P; ++p) {
            const int I = p * 8 + wave, L = I * 64 + lane;
            const int q = L >> 3, hr = q / WS_HW, hc = q - hr * WS_HW;
            const int c = (L & 7) ^ ((hc >> 1) & 7);
            doff[p] = ((hr * g.W + hc) * g.x_ld + c * 8) * 2;
            const unsigned bits = (hr == 0 ? 1u : 0u) | (hr == WS_HW - 1 ? 2u : 0u) | (hc == 0 ? 4u : 0u) | (hc == WS_HW - 1 ? 8u : 0u) | (q >= WS_HW * WS_HW ? 16u : 0u);
            ring |= bits << (5 * p);
        }
        // the lane's 3x3 window of output pixel (prow, col) starts at halo pixel (prow, col); window column pw -> hc = col + pw
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wm + i * 32 + (lane & 31);
            const int prow = m >> 4, col = m & 15, h = lane >> 5;
#pragma unroll
            for (int pw = 0; pw < 3; ++pw) {
                const int sw = ((col + pw) >> 1) & 7;
                ad[i][pw] = (prow * WS_HW + col) * 128 + ((sw >> 1) << 5) + ((h ^ (sw & 1)) << 4);
            }
        }
        // a use of every weight fragment in front of the tile loop: otherwise the loads are still "pending" at the loop header and
        // the compiler's vmcnt(0) before the first MFMA of EVERY tile also drains the halo DMA in flight
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(wr[tap][kk]));
    }

    // tile coordinates (image, tile row, tile column) advance by nstreams tiles per step with carries: no divisions in the loop
    struct Tile { int n, th, tw; };
    const int step_n = nstreams / tiles_img, step_r = nstreams - step_n * tiles_img, step_h = step_r / tiles_w, step_w = step_r - step_h * tiles_w;
    auto advance = [&](Tile c) {
        c.tw += step_w;
        if (c.tw >= tiles_w) { c.tw -= tiles_w; ++c.th; }
        c.th += step_h;
        if (c.th >= tiles_h) { c.th -= tiles_h; ++c.n; }
        c.n += step_n;
        return c;
    };
    auto issue_halo = [&](Tile c, int bufbase) {
        const int n = c.n, th = c.th, tw = c.tw;
        const unsigned edge = (th == 0 ? 1u : 0u) | (th == tiles_h - 1 ? 2u : 0u) | (tw == 0 ? 4u : 0u) | (tw == tiles_w - 1 ? 8u : 0u) | 16u;
        const char* base = reinterpret_cast<const char*>(X) + (((int64_t)n * g.H + th * 16 - 1) * g.W + (tw * 16 - 1)) * g.x_ld * 2;
        const char* zp = reinterpret_cast<const char*>(ws_zero_page) + (tid & 7) * 16;
#pragma unroll
        for (int p = 0; p < WS_DMA_P; ++p) {
            const int I = p * 8 + wave;
            if (I < WS_DMA_I) {
                const bool ok = ((ring >> (5 * p)) & edge) == 0;
                int off = doff[p];
                asm volatile("" : "+v"(off));       // keep the 32-bit offset in its register: pre-extended to 64 bits the six spill
                const char* src = ok ? base + off : zp;
                // The DMA goes through inline asm, not __builtin_amdgcn_global_load_lds: with the builtin anywhere in the loop the
                // compiler stops counting LDS reads (every wait in front of an MFMA becomes lgkmcnt(0): the fragment prefetch
                // ring then exposes a full LDS latency once per nine MFMAs).  hipcc does not see this load: its completion is
                // the explicit vmcnt(0) in front of the barrier below.  M0 = LDS byte address of the instruction's 1 KB.
                unsigned keep;
                const unsigned dst = lds_base + (unsigned)(bufbase + I * 1024);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        }
    };

    // Tile loop of one wave:   MFMAs(i) | vmcnt(0), barrier | epilogue(i), DMA share of tile i+2 | barrier
    // Waves 4-7 pass one extra barrier in front of the loop (waves 0-3 one behind it), i.e. run one phase later.  Halo i lives in
    // buffer i % 3.  RAW: a wave's share of DMA(i+2) is issued behind epilogue(i) and retired by the vmcnt(0) behind its
    // MFMAs(i+1); the first reader of halo i+2 is two barriers further down.  WAR: buffer (i+2) % 3 held halo i-1, whose last
    // reader (a wave of the late group, MFMAs(i-1)) passed two barriers before the earliest writer gets here.  The vmcnt(0) also
    // covers the stores of epilogue(i-1) — issued a whole MFMA phase earlier — and the epilogue's own loads (residual) are issued
    // while no DMA of this wave is in flight, so the compiler's vmcnt(0) in front of their use drains nothing else.
    int t = stream;
    Tile cur, nx1, nx2;                       // tiles t, t + nstreams, t + 2 nstreams
    cur.n = t / tiles_img; cur.th = (t - cur.n * tiles_img) / tiles_w; cur.tw = t - cur.n * tiles_img - cur.th * tiles_w;
    nx1 = advance(cur);
    nx2 = advance(nx1);
    if (t < T) issue_halo(cur, 0);
    if (t + nstreams < T) issue_halo(nx1, WS_HALO);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // lgkmcnt: the bias staged in the LDS above
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(CN_MAIN_PRIO + 1);        // the later-dispatched half loses every VALU arbitration against its SIMD partner otherwise
    }
    int bufbase = 0;
#pragma unroll 1
    for (int it = 0; t < T; t += nstreams, ++it) {
        WS_STAMP(0);
        // accumulators start at the bias of their channel (8q + 4 (lane >> 5) + e of the wave's 32): four LDS reads instead of
        // sixteen adds per fragment in the epilogue
        f32x16_t acc[MI];
        {
            const float* bl = bias_l + wn + 4 * ((tid & 63) >> 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bl + 8 * q);
#pragma unroll
                for (int i = 0; i < MI; ++i) { acc[i][4 * q] = bq.x; acc[i][4 * q + 1] = bq.y; acc[i][4 * q + 2] = bq.z; acc[i][4 * q + 3] = bq.w; }
            }
        }
        // 36 * MI (window position, K step, fragment) MFMAs in groups of GS; the fragments of group n + 2 are requested before the
        // MFMAs of group n issue (a ring of three register sets: 36 registers; with groups of four the kernel spills, and a
        // scratch reload inside the DMA phase waits for every DMA in flight).  Left to itself the compiler reads every fragment
        // into the same four registers right in front of its MFMA (read, lgkmcnt(0), MFMA: the LDS latency exposed 72 times per tile).
        constexpr int GS = 3, NG = 36 * MI / GS;
        bf16x8_t fr[3][GS];
        auto gload = [&](bf16x8_t (&f)[GS], int grp) {
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const int idx = grp * GS + j, i = idx % MI, kk = (idx / MI) % 4, pos = idx / (4 * MI), ph = pos / 3, pw = pos % 3;
                f[j] = *reinterpret_cast<const bf16x8_t*>(lds + (ad[i][pw] ^ (kk << 5)) + (ph * WS_HW + pw) * 128);
            }
        };
        gload(fr[0], 0);
        gload(fr[1], 1);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 2 < NG) gload(fr[(grp + 2) % 3], grp + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const int idx = grp * GS + j, i = idx % MI, kk = (idx / MI) % 4, pos = idx / (4 * MI);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[pos][kk], fr[grp % 3][j], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        WS_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        WS_STAMP(2);

        // Epilogue straight from the accumulators.  Lane l holds pixel (l & 31), channels 8q + 4 (l >> 5) + 0..3 of quad q: one
        // v_permlane32_swap per value trades quad q+1 of the low half-wave for quad q of the high half-wave, after which every
        // lane owns EIGHT consecutive channels of its pixel (low half: quad q, high half: quad q+1) -> 16-byte bf16 (2 x 16-byte
        // fp32) stores, two per 32-pixel fragment.  (8-byte stores straight from the quads touch twice as many cache lines per
        // byte, and the address path is what the stores and the halo DMA share; an LDS transpose to channel vectors, as in
        // conv_epilogue_tile, costs four dependent LDS round trips: with one wave per SIMD in this phase that latency was longer
        // than the MFMA phase it hides behind.)  fp32 bias / residual / ReLU, then one rounding, as in conv_epilogue.  Channels
        // Co .. y_ld-1 (the activation's zero padding) come out as zeros without a mask: their weight rows are the zero rows of
        // the packed matrix, their bias is staged as 0 and the residual's own padding is zero.
        if constexpr (HEAD == 2) {
            int lane = tid & 63;
            asm volatile("" : "+v"(lane));
            const float* hl = head_l + wn + 4 * (lane >> 5);
            float* const ymap = reinterpret_cast<float*>(g.y) + ((int64_t)cur.n * 2 + (lane >> 5)) * g.OH * g.OW;     // half 0 -> channel 0, half 1 -> channel 1
            const float b2 = (n0 + wn == 0) ? g.head_b[lane >> 5] : 0.f;            // the head's bias enters once per pixel and channel
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w0 = *reinterpret_cast<const float4*>(hl + 8 * q), w1 = *reinterpret_cast<const float4*>(hl + 64 + 8 * q);
                    const float v0 = __builtin_amdgcn_fmed3f(acc[i][4 * q], 0.f, INFINITY), v1 = __builtin_amdgcn_fmed3f(acc[i][4 * q + 1], 0.f, INFINITY);
                    const float v2 = __builtin_amdgcn_fmed3f(acc[i][4 * q + 2], 0.f, INFINITY), v3 = __builtin_amdgcn_fmed3f(acc[i][4 * q + 3], 0.f, INFINITY);
                    p0 = fmaf(v0, w0.x, fmaf(v1, w0.y, fmaf(v2, w0.z, fmaf(v3, w0.w, p0))));
                    p1 = fmaf(v0, w1.x, fmaf(v1, w1.y, fmaf(v2, w1.z, fmaf(v3, w1.w, p1))));
                }
                // lanes l and l + 32 hold the same pixel: the low half keeps channel 0 (its p0 + the high half's p0), the high half channel 1
                const float o0 = __shfl_xor(p0, 32, 64), o1 = __shfl_xor(p1, 32, 64);
                const float mine = (lane >> 5) ? p1 : p0, other = (lane >> 5) ? o1 : o0;
                const int m = wm + i * 32 + (lane & 31);
                atomicAdd(ymap + (int64_t)(cur.th * 16 + (m >> 4)) * g.OW + cur.tw * 16 + (m & 15), mine + other + b2);
            }
        } else {
            int lane = tid & 63;
            asm volatile("" : "+v"(lane));      // recomputed per tile: hoisted, the per-lane addressing would pin registers
            // tile base in scalar registers, 32-bit per-lane byte offsets (a tile spans < 16 rows of the map)
            const int64_t pix0 = ((int64_t)cur.n * g.OH + cur.th * 16) * g.OW + cur.tw * 16;
            char* const ytile = reinterpret_cast<char*>(g.y) + pix0 * g.y_ld * (YF32 ? 4 : 2);
            const char* const rtile = reinterpret_cast<const char*>(g.res) + pix0 * g.res_ld * 2;
            const int chl = wn + 8 * (lane >> 5);                      // + 16 qq: the lane's 8 channels of quad pair qq
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = wm + i * 32 + (lane & 31);
                const unsigned px = (unsigned)((m >> 4) * g.OW + (m & 15));       // pixel inside the tile's 16 rows
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int ch = n0 + chl + 16 * qq;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][(2 * qq) * 4 + e]), __float_as_uint(acc[i][(2 * qq + 1) * 4 + e]), false, false);
                        v[e] = __uint_as_float(sw2[0]);
                        v[4 + e] = __uint_as_float(sw2[1]);
                    }
                    if (ch >= g.y_ld) continue;
                    if constexpr (RES != 0) {
                        float rv[8];
                        Vec16<bf16_t>::load(reinterpret_cast<const bf16_t*>(rtile + (px * (unsigned)g.res_ld + (unsigned)ch) * 2u), rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = RES == 2 ? (rv[e] > 0.f ? v[e] : 0.f) : v[e] + rv[e];
                    }
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, INFINITY);     // max(v, 0) in one instruction
                    }
                    if constexpr (YF32) {
                        float* dst = reinterpret_cast<float*>(ytile + (px * (unsigned)g.y_ld + (unsigned)ch) * 4u);
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        Vec16<bf16_t>::store(reinterpret_cast<bf16_t*>(ytile + (px * (unsigned)g.y_ld + (unsigned)ch) * 2u), v);
                    }
                }
            }
        }
        WS_STAMP(3);
        const int b2 = bufbase >= WS_HALO ? bufbase - WS_HALO : bufbase + 2 * WS_HALO;      // buffer (i + 2) % 3
        if (t + 2 * nstreams < T) issue_halo(nx2, b2);
        cur = nx1; nx1 = nx2; nx2 = advance(nx2);
        const int nb_ = bufbase == 2 * WS_HALO ? 0 : bufbase + WS_HALO;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int pw = 0; pw < 3; ++pw) ad[i][pw] += nb_ - bufbase;       // halo buffers are 1 KB multiples: bits 4-6 untouched
        bufbase = nb_;
        WS_STAMP(4);
        __builtin_amdgcn_s_barrier();
        WS_STAMP(5);
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
}

// caller guarantees: 3x3 / stride 1 / pad 1 geometry in class 0 of g (normal or mirrored taps), OH == H, OW == W
bool conv3x3_ws_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV_WS") != nullptr;
    static int cus = 0;
    if (disabled || dtype != CN_BF16 || g.Ci != 64 || (g.x_ld & 7) || (g.H & 15) || (g.W & 15) || g.nsrc != 0 || g.dcn_x != nullptr) return false;
    // what the kernel's own epilogue covers: bf16 or fp32 rows made of 8-channel vectors, optional bf16 residual / ReLU mask
    if (g.res32 != nullptr || (g.head_nc == 0 && (g.y_ld & 7)) || (g.res != nullptr && (g.res_ld & 7))) return false;
    if ((reinterpret_cast<uintptr_t>(g.x) | reinterpret_cast<uintptr_t>(g.w) | reinterpret_cast<uintptr_t>(g.y) | reinterpret_cast<uintptr_t>(g.res)) & 15) return false;
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
    const dim3 gr(grid), bl(WS_NT);
    const int res = g.res == nullptr ? 0 : (g.relu == 2 ? 2 : 1);
    if (g.y_f32 && res != 0) return false;
    if (g.head_nc != 0) {                     // fused 2-channel task head: 64-wide channel blocks, ReLU, no residual; the map is fp32 NCHW
        if (g.head_nc != 2 || bn != 64 || (g.Co & 63) || g.relu != 1 || res != 0 || !g.head_w || !g.head_b) return false;
        hipLaunchKernelGGL((conv3x3_ws_kernel<64, false, 0, true, 2>), gr, bl, 0, st, g, nblk, tiles_h, tiles_w, wmap);
        return true;
    }
#define WS_GO(BN_, F32_, RES_, RELU_) hipLaunchKernelGGL((conv3x3_ws_kernel<BN_, F32_, RES_, RELU_>), gr, bl, 0, st, g, nblk, tiles_h, tiles_w, wmap)
#define WS_PICK(BN_)                                                          \
    do {                                                                      \
        if (g.y_f32) { if (g.relu == 1) WS_GO(BN_, true, 0, true); else WS_GO(BN_, true, 0, false); } \
        else if (res == 0) { if (g.relu == 1) WS_GO(BN_, false, 0, true); else WS_GO(BN_, false, 0, false); } \
        else if (res == 1) { if (g.relu == 1) WS_GO(BN_, false, 1, true); else WS_GO(BN_, false, 1, false); } \
        else WS_GO(BN_, false, 2, false);                                     \
    } while (0)
    if (bn == 64) WS_PICK(64); else WS_PICK(32);
#undef WS_PICK
#undef WS_GO
    return true;
}
