// K-pipelined 3x3 / stride 1 / pad 1 convolution for >= 128 input channels (DLA-34 levels 3-5: 128 -> 128 @64^2, 256 -> 256 @32^2,
// 512 -> 512 @16^2, and their data gradients).  The weight-stationary kernel (conv3x3_ws.hip) stops at 64 input channels — a wave's
// weight slice no longer fits its registers — and the halo-tile kernel (conv3x3.hip) moves 16 KB of weights per tap through
// registers -> ds_write -> __syncthreads for a 128-pixel tile: LDS-pipe bound at 0.36 of the MFMA peak.  Here
//   * one persistent workgroup of 8 waves per CU owns a 16x16-pixel tile x 128 output channels (64x64 wave tiles: one LDS fragment
//     read per MFMA, the weight slice of a step shared by 256 pixels instead of 128);
//   * BOTH operands reach the LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write issue): the (16+2)^2 halo
//     tile of a 64-channel slice in two buffers (the weight-stationary kernel's swizzled pixel-major image), the per-(window
//     position, channel slice) weight slices [128 rows][64 k] through a ring of four 16 KB slots, three steps ahead of their use,
//     retired by COUNTED s_waitcnt vmcnt(n) — never 0 in the steady state — and raw s_barriers;
//   * the two waves of a SIMD run ONE PHASE APART (cdna_hip_programming.md, 8-phase template): a step is a read phase (16
//     ds_read_b128 = all fragments of the step, the wave's share of the DMA, one piece of the previous tile's epilogue) and an MFMA
//     phase (16 MFMAs), separated by barriers; waves 0-3 read while waves 4-7 multiply and vice versa, so the matrix pipe of a SIMD
//     always has exactly one wave's MFMA phase to run;
//   * roles: waves 4-7 (a phase later, three steps of lead) issue the weight DMA, waves 0-3 the halo DMA of the next slice;
//   * the epilogue of tile i (straight from a second accumulator set: v_permlane32_swap pairs -> 16-byte stores, as in the
//     weight-stationary kernel) is spread over the read phases of the first eight steps of tile i+1.
// Step g of a workgroup (global over its tiles): window position pos = g % 9 of channel slice c = (g / 9) % (Ci / 64).
//   time slots (one barrier each):   waves 0-3: R(g) = T[2g], M(g) = T[2g+1];   waves 4-7: R(g) = T[2g+1], M(g) = T[2g+2]
//   weights W(g) live in slot g % 4.  RAW: waves 4-7 issue W(g+3) in their R(g), wait for THEIR pieces of W(g+1) at the end of
//   the same phase (vmcnt = pieces of W(g+2), W(g+3) still in flight) and pass the barrier that precedes the first read of slot
//   (g+1) % 4 (waves 0-3, T[2g+2]).  WAR: slot (g+3) % 4 = (g-1) % 4 was last read in T[2g-1] (waves 4-7), those reads were
//   waited for (lgkmcnt(0)) at the head of T[2g], a barrier before T[2g+1].
//   halo slice k lives in buffer k % 2: waves 0-3 issue slice k+1 in R(pos 1..6) of slice k (the buffer's last readers, waves 4-7 in
//   R(pos 8) of slice k-1, are two barriers behind) and wait vmcnt(0) in R(pos 8).
// vmcnt bookkeeping relies on gfx9's in-order return of vector-memory operations; the compiler does not see the DMA (inline asm),
// so the waits it inserts for its own loads / stores only ever over-wait.
#include "conv_common.h"
#include <algorithm>
#include <type_traits>

#define KP_HW 18
#define KP_DMA_I 41                            // 1 KB DMA instructions per halo slice: 324 pixels x 128 B = 40.5 KB
#define KP_HALO (KP_DMA_I * 1024)
#define KP_WSLOT 16384                         // 128 rows x 64 k x 2 B
#define KP_NSLOT 4
#define KP_NT 512
#ifndef KP_ABL
#define KP_ABL 0                               // timing ablations (wrong results): bit 0 no weight DMA, 1 no halo DMA, 2 no vmcnt waits, 3 fragment reads only in the first step
#endif
#ifndef KP_SPLIT
#define KP_SPLIT 0                             // fragments of K steps 2, 3 are read in the MFMA phase (behind the MFMAs of K steps 0, 1)
#endif
#ifndef KP_INM
#define KP_INM 0                               // weight pieces (of the wave's two per step) issued inside the MFMA phase instead of the read phase
#endif
#define KP_LDS (2 * KP_HALO + KP_NSLOT * KP_WSLOT + 128 * 4)

__device__ uint4 kp_zero_page[8];              // 128 zero bytes: DMA source of halo pixels outside the image
#ifdef KP_PROBE   // development build only (tools/kp_probe.py): per-phase cycle stamps of waves 0 and 4 of every workgroup, first 40 steps
__device__ unsigned long long kp_ts[256 * 2 * 40 * 8];
#define KP_STAMP(k) do { if ((wave & 3) == 0 && gstep < 40 && lane == 0 && blockIdx.x < 256) kp_ts[((blockIdx.x * 2 + (wave >> 2)) * 40 + gstep) * 8 + (k)] = clock64(); } while (0)
extern "C" int kp_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(kp_ts), sizeof(kp_ts)); }
#else
#define KP_STAMP(k) do { } while (0)
#endif

// one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at `dst` (M0), lane-linear
__device__ static inline void kp_dma(const char* src, unsigned dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory", "m0");
}

// RES: 0 none, 1 y += res (bf16), 2 ReLU-backward mask y = res > 0 ? y : 0; RELU: y = max(y, 0)
// LOCK: the lockstep form (all eight waves in the same step, fragment ring across steps, ONE barrier per step) instead of the
// phase-staggered one — see the second half of the kernel body
template <int RES, bool RELU, bool LOCK>
__global__ __launch_bounds__(KP_NT) void conv3x3_kp_kernel(const ConvGeom g, int nblk, int tiles_h, int tiles_w, uint64_t wmap) {
    CN_MAIN_PRIO_SET();
    __shared__ __attribute__((aligned(1024))) unsigned char lds[KP_LDS];
    float* const bias_l = reinterpret_cast<float*>(lds + 2 * KP_HALO + KP_NSLOT * KP_WSLOT);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    constexpr unsigned WOFF = 2 * KP_HALO;     // byte offset of the weight ring

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = !LOCK && wave >= 4;
    const int wm = (wave & 3) * 64, wn = (wave >> 2) * 64;        // wave tile: 64 pixels (4 tile rows) x 64 output channels
    // waves 0-3 and 4-7 differ in the CHANNEL half, so a SIMD's two waves (w, w + 4) share their pixels' A fragments' LDS lines
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int nb = rr % nblk, stream = xcd + 8 * (rr / nblk), nstreams = gridDim.x / nblk;
    const int n0 = nb * 128;
    const int tiles_img = tiles_h * tiles_w, T = g.N * tiles_img;
    const int nsl = g.Ci >> 6, S = 9 * nsl;
    const int nitems = stream < T ? (T - stream + nstreams - 1) / nstreams : 0;
    const int Gtot = nitems * S;

    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const char* const Wb = reinterpret_cast<const char*>(g.w);

    if (tid < 128) bias_l[tid] = (g.bias && n0 + tid < g.Co) ? g.bias[n0 + tid] : 0.f;

    // ---- per-lane constants ----
    // A fragments (the weight-stationary kernel's halo image): byte address of fragment i at window column pw, K step 0, window row 0
    int ad[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        const int prow = m >> 4, col = m & 15, h = lane >> 5;
#pragma unroll
        for (int pw = 0; pw < 3; ++pw) {
            const int sw = ((col + pw) >> 1) & 7;
            ad[i][pw] = (prow * KP_HW + col) * 128 + ((sw >> 1) << 5) + ((h ^ (sw & 1)) << 4);
        }
    }
    // B fragments: row r = wn + 32 j + (lane & 31) of the slot, 16-byte chunk (2 kk + (lane >> 5)) stored at chunk ^ ((r >> 1) & 7): the
    // 8 + 8 rows of a ds_read_b128 lane group then cover the 16 slots of a 256-byte bank row exactly once
    int bd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wn + j * 32 + (lane & 31), s = (r >> 1) & 7, h = lane >> 5;
        bd[j] = (int)WOFF + r * 128 + ((s >> 1) << 5) + ((h ^ (s & 1)) << 4);
    }
    // weight DMA: every wave owns 16 rows of a slot (two 1 KB pieces): rows 16 wave + 8 pp + (lane >> 3); the lane's LDS chunk position
    // lane & 7 holds source chunk (lane & 7) ^ ((row >> 1) & 7)
    int woff[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int r = 16 * wave + 8 * pp + (lane >> 3);
        const int row = min(n0 + r, g.co_pad - 1);                   // rows past the packed matrix: never stored
        woff[pp] = row * g.ktot * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    // halo DMA: wave w owns pieces I = w + 8 p, p = 0..5 (I >= 41 does not exist: the wave re-issues its piece p = 4 instead, the same
    // bytes to the same place, so that every wave has the same number of DMA instructions in flight — the vmcnt bookkeeping below
    // counts them).  Per piece: byte offset of the lane's 16 bytes relative to the halo's top-left pixel, and four border bits.
    int hoff[6];
    unsigned hbits = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int I = wave + 8 * p < KP_DMA_I ? wave + 8 * p : wave + 8 * 4;
        const int L = I * 64 + lane, q = min(L >> 3, KP_HW * KP_HW - 1), hr = q / KP_HW, hc = q - hr * KP_HW;
        hoff[p] = ((hr * g.W + hc) * g.x_ld + ((L & 7) ^ ((hc >> 1) & 7)) * 8) * 2;
        const unsigned bits = (hr == 0 ? 1u : 0u) | (hr == KP_HW - 1 ? 2u : 0u) | (hc == 0 ? 4u : 0u) | (hc == KP_HW - 1 ? 8u : 0u) | ((L >> 3) >= KP_HW * KP_HW ? 16u : 0u);
        hbits |= bits << (5 * p);
    }

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
    // halo source of a slice: scalar pointer to the halo's top-left pixel (channel slice included) + the tile's border bits
    struct HaloSrc { const char* base; unsigned edge; };
    auto halo_src = [&](Tile c, int c0) {
        HaloSrc h;
        h.edge = (c.th == 0 ? 1u : 0u) | (c.th == tiles_h - 1 ? 2u : 0u) | (c.tw == 0 ? 4u : 0u) | (c.tw == tiles_w - 1 ? 8u : 0u) | 16u;
        h.base = reinterpret_cast<const char*>(X) + ((((int64_t)c.n * g.H + c.th * 16 - 1) * g.W + (c.tw * 16 - 1)) * g.x_ld + c0) * 2;
        // computed ONCE per slice: without the laundering the compiler re-derives these ~45 scalar instructions in front of every piece
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)h.base), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)h.base >> 32));
        h.edge = __builtin_amdgcn_readfirstlane(h.edge);
        h.base = reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi << 32) | lo));
        return h;
    };
    // the wave's halo piece p of slice `h` into the halo buffer at byte offset hbo (branch-free: pixels outside the image read the zero page)
    auto halo_piece = [&](const HaloSrc& h, int hbo, int p) {
        if (KP_ABL & 2) return;
        const int I = wave + 8 * p < KP_DMA_I ? wave + 8 * p : wave + 8 * 4;
        const bool ok = ((hbits >> (5 * p)) & h.edge) == 0;
        int off = hoff[p], ln = lane;
        asm volatile("" : "+v"(off), "+v"(ln));
        const uint64_t zpage = (uint64_t)(uintptr_t)reinterpret_cast<const char*>(kp_zero_page) + (uint64_t)((ln & 7) * 16);
        const uint64_t a = (uint64_t)(uintptr_t)h.base + (uint64_t)(int64_t)off;
        const unsigned lo = ok ? (unsigned)a : (unsigned)zpage, hi = ok ? (unsigned)(a >> 32) : (unsigned)(zpage >> 32);
        kp_dma(reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi << 32) | lo)), lds_base + (unsigned)(hbo + I * 1024));
    };
    // weight DMA piece pp of the wave for (window position pos, channel slice c) into ring slot sl
    const int ci2 = g.Ci * 2;
    auto weight_piece = [&](int pos, int c, int sl, int pp) {
        if (KP_ABL & 1) return;
        const int col = (int)((wmap >> (4 * pos)) & 15) * ci2 + c * 128;
        kp_dma(Wb + col + woff[pp], lds_base + WOFF + (unsigned)(sl * KP_WSLOT + (wave * 2 + pp) * 1024));
    };

    f32x16_t acc[2][2], prev[2][2];
    auto init_acc = [&]() {
        const float* bl = bias_l + wn + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(bl + j * 32 + 8 * q);
#pragma unroll
                for (int i = 0; i < 2; ++i) { acc[j][i][4 * q] = bq.x; acc[j][i][4 * q + 1] = bq.y; acc[j][i][4 * q + 2] = bq.z; acc[j][i][4 * q + 3] = bq.w; }
            }
    };
    // epilogue addressing: scalar pointer to the tile's first pixel (channel block included) + per-lane byte offsets of the lane's pixel
    // of fragment i and its 8-channel group
    char* ybase = nullptr;                       // ... of the tile whose results sit in `prev`
    const char* rbase = nullptr;
    auto tile_bases = [&](Tile c) {
        const int64_t pix0 = ((int64_t)c.n * g.OH + c.th * 16) * g.OW + c.tw * 16;
        ybase = reinterpret_cast<char*>(g.y) + (pix0 * g.y_ld + n0) * 2;
        if constexpr (RES != 0) rbase = reinterpret_cast<const char*>(g.res) + (pix0 * g.res_ld + n0) * 2;
    };
    // one eighth of a tile's epilogue: fragment (j, i), quad pair qq -> one 16-byte store per lane
    auto epilogue_chunk = [&](int e) {
        const int i = e & 1, j = (e >> 1) & 1, qq = e >> 2;
        int ln = lane;
        asm volatile("" : "+v"(ln));              // per-lane offsets recomputed per chunk (32-bit): kept live they cost five registers all loop long
        const int m = wm + i * 32 + (ln & 31), cl = wn + 8 * (ln >> 5) + 32 * j + 16 * qq;
        const int pixl = (m >> 4) * g.OW + (m & 15);
        float v[8];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(prev[j][i][(2 * qq) * 4 + e4]), __float_as_uint(prev[j][i][(2 * qq + 1) * 4 + e4]), false, false);
            v[e4] = __uint_as_float(sw2[0]);
            v[4 + e4] = __uint_as_float(sw2[1]);
        }
        if (n0 + cl >= g.y_ld) return;
        if constexpr (RES != 0) {
            float rv[8];
            Vec16<bf16_t>::load(reinterpret_cast<const bf16_t*>(rbase + (unsigned)((pixl * g.res_ld + cl) * 2)), rv);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = RES == 2 ? (rv[k] > 0.f ? v[k] : 0.f) : v[k] + rv[k];
        }
        if constexpr (RELU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k], 0.f, INFINITY);
        }
        Vec16<bf16_t>::store(reinterpret_cast<bf16_t*>(ybase + (unsigned)((pixl * g.y_ld + cl) * 2)), v);
    };

    if constexpr (LOCK) {
        // ======================================= lockstep form =======================================
        // All eight waves run the same step.  A step's 16 MFMAs go in four groups (K steps kk = 0..3: fragments B j = 0,1 and A i = 0,1);
        // the fragments of group G + 2 are requested before the MFMAs of group G issue (a ring of three register sets that runs ACROSS
        // steps: 36 groups per slice = 0 mod 3, so the ring position is a compile-time function of (pos, kk)).  Groups 2, 3 of step s
        // therefore read slot (s+1) % 4 and — in step 8 of a slice — the other halo buffer: ONE barrier per step, between groups 1 and
        // 2, in front of which every wave has waited for ITS pieces of W(s+1) (and, in step 8, of the next slice's halo).  Behind that
        // barrier slot (s-1) % 4 is free (its last reads fed the MFMAs of step s-1): W(s+3) is issued there, two steps ahead of the
        // barrier that publishes it; the next slice's halo goes out in steps 0..5 (the buffer's last reads fed step 8 of the slice before).
        Tile cur;
        cur.n = stream / tiles_img; cur.th = (stream - cur.n * tiles_img) / tiles_w; cur.tw = stream - cur.n * tiles_img - cur.th * tiles_w;
        if (nitems > 0) {
            const HaloSrc h0 = halo_src(cur, 0);
#pragma unroll
            for (int p = 0; p < 6; ++p) halo_piece(h0, 0, p);
            for (int s = 0; s < 3 && s < Gtot; ++s) { weight_piece(s, 0, s, 0); weight_piece(s, 0, s, 1); }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int gstep = 0;
        bf16x8_t fr[3][4];                       // [ring position][B j = 0, B j = 1, A i = 0, A i = 1]
        auto request = [&](bf16x8_t (&f)[4], unsigned hbuf, unsigned wslot, int pos_, int kk) {
            if ((KP_ABL & 8) && gstep > 0) return;
            const int ph = pos_ / 3, pw = pos_ % 3;
            int a0 = ad[0][pw] + (int)hbuf, a1 = ad[1][pw] + (int)hbuf, b0 = bd[0] + (int)wslot, b1 = bd[1] + (int)wslot;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));     // recomputed per group: hoisted, 9 positions x 4 K steps of addresses pin registers
            f[0] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)(b0 ^ (kk << 5)));
            f[1] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)(b1 ^ (kk << 5)));
            f[2] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)((a0 ^ (kk << 5)) + (ph * KP_HW + pw) * 128));
            f[3] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)((a1 ^ (kk << 5)) + (ph * KP_HW + pw) * 128));
        };
        unsigned hb = 0;
        bool has_prev = false;
        init_acc();
        request(fr[0], 0u, 0u, 0, 0);
        request(fr[1], 0u, 0u, 0, 1);
#pragma unroll 1
        for (int item = 0; item < nitems; ++item) {
            const Tile nxt = advance(cur);
#pragma unroll 1
            for (int c = 0; c < nsl; ++c) {
                const bool more = c + 1 < nsl || item + 1 < nitems;
                const HaloSrc nsrc = halo_src(c + 1 < nsl ? cur : nxt, c + 1 < nsl ? (c + 1) * 64 : 0);
                const int cnx = c + 1 < nsl ? c + 1 : 0;
                const unsigned nhb = hb ? 0u : (unsigned)KP_HALO;
                auto do_step = [&](auto POS_) __attribute__((always_inline)) {
                    constexpr int pos = decltype(POS_)::value;
                    const unsigned wsl = (unsigned)(gstep & 3) * KP_WSLOT, wsl1 = (unsigned)((gstep + 1) & 3) * KP_WSLOT;
                    const unsigned hb1 = pos == 8 ? nhb : hb;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        constexpr int dummy = 0; (void)dummy;
                        const int G = pos * 4 + kk;                       // group index inside the slice; ring position G % 3
                        if (kk == 2) {
                            // ---- the step's barrier: W(s+1) (and in step 8 the next slice's halo) published, slot (s-1) % 4 released ----
                            const int rem = Gtot - 1 - gstep;
                            constexpr int h1 = (pos >= 1 && pos <= 6) ? 1 : 0, h2 = (pos >= 2 && pos <= 7) ? 1 : 0;
                            if (KP_ABL & 4) { }
                            else if (rem >= 2) { if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + h2 + h1) : "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (!(KP_ABL & 16)) __builtin_amdgcn_s_barrier();
                            if (gstep + 3 < Gtot) {
                                weight_piece((pos + 3) % 9, pos + 3 >= 9 ? cnx : c, (gstep + 3) & 3, 0);
                                weight_piece((pos + 3) % 9, pos + 3 >= 9 ? cnx : c, (gstep + 3) & 3, 1);
                            }
                            if (pos <= 5 && more) halo_piece(nsrc, (int)nhb, pos);
                            if (has_prev && c == 0 && pos < 8) epilogue_chunk(pos);
                        }
                        // fragments of group G + 2: K steps 2, 3 of this step, or K steps 0, 1 of the next one
                        if (kk < 2) request(fr[(G + 2) % 3], hb, wsl, pos, kk + 2);
                        else request(fr[(G + 2) % 3], hb1, wsl1, (pos + 1) % 9, kk - 2);
                        __builtin_amdgcn_sched_barrier(0);
                        {
                            bf16x8_t (&f)[4] = fr[G % 3];
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[j], f[2 + i], acc[j][i], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ++gstep;
                };
                do_step(std::integral_constant<int, 0>{}); do_step(std::integral_constant<int, 1>{}); do_step(std::integral_constant<int, 2>{});
                do_step(std::integral_constant<int, 3>{}); do_step(std::integral_constant<int, 4>{}); do_step(std::integral_constant<int, 5>{});
                do_step(std::integral_constant<int, 6>{}); do_step(std::integral_constant<int, 7>{}); do_step(std::integral_constant<int, 8>{});
                hb = nhb;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) prev[j][i] = acc[j][i];
            tile_bases(cur);
            has_prev = true;
            init_acc();
            cur = nxt;
        }
        if (has_prev) {
#pragma unroll
            for (int e = 0; e < 8; ++e) epilogue_chunk(e);
        }
        return;
    }

    // ---- prologue: halo slice 0 of the first tile (all waves), weights of steps 0..2 (waves 4-7) ----
    Tile cur;
    cur.n = stream / tiles_img; cur.th = (stream - cur.n * tiles_img) / tiles_w; cur.tw = stream - cur.n * tiles_img - cur.th * tiles_w;
    if (nitems > 0) {
        const HaloSrc h0 = halo_src(cur, 0);
#pragma unroll
        for (int p = 0; p < 6; ++p) halo_piece(h0, 0, p);
        // waves 0-3 run two steps ahead with the weights (they issue W(g+2) in R(g)), waves 4-7 three (W(g+3) in their R(g), a slot later)
        for (int s = 0; s < (late ? 3 : 2) && s < Gtot; ++s) { weight_piece(s, 0, s, 0); weight_piece(s, 0, s, 1); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (late) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(CN_MAIN_PRIO + 1);      // the later-dispatched half loses every VALU arbitration against its SIMD partner otherwise
    }

#if KP_ABL & 8
    bf16x8_t fa[2][4] = {}, fb[2][4] = {};
#endif
    int gstep = 0;                               // global step of this workgroup
    unsigned hb = 0;                             // byte offset of the current halo buffer
    bool has_prev = false;
    init_acc();
#pragma unroll 1
    for (int item = 0; item < nitems; ++item) {
        const Tile nxt = advance(cur);
#pragma unroll 1
        for (int c = 0; c < nsl; ++c) {
            // the slice after this one: the tile's next channel slice, or slice 0 of the workgroup's next tile
            const bool more = c + 1 < nsl || item + 1 < nitems;
            const HaloSrc nsrc = halo_src(c + 1 < nsl ? cur : nxt, c + 1 < nsl ? (c + 1) * 64 : 0);
            const int cnx = c + 1 < nsl ? c + 1 : 0;                  // channel slice of the steps that wrap past this slice's ninth
            const unsigned nhb = hb ? 0u : (unsigned)KP_HALO;
            // one step (window position pos, a compile-time constant: the vmcnt immediates below depend on it)
            auto do_step = [&](auto POS_) __attribute__((always_inline)) {
                constexpr int pos = decltype(POS_)::value;
                // ================= read phase =================
                const int ph = pos / 3, pw = pos % 3;
                const unsigned wsl = (unsigned)(gstep & 3) * KP_WSLOT;
#if !(KP_ABL & 8)
                bf16x8_t fa[2][4], fb[2][4];
#endif
                if (!(KP_ABL & 8) || gstep == 0) {
                    // base addresses laundered per step: the XORs below are then recomputed (1 VALU per read) instead of 32 hoisted registers
                    int a0 = ad[0][pw] + (int)hb, a1 = ad[1][pw] + (int)hb, b0 = bd[0] + (int)wsl, b1 = bd[1] + (int)wsl;
                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
#pragma unroll
                    for (int kk = 0; kk < (KP_SPLIT ? 2 : 4); ++kk) {
                        fb[0][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)(b0 ^ (kk << 5)));
                        fb[1][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)(b1 ^ (kk << 5)));
                        fa[0][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)((a0 ^ (kk << 5)) + (ph * KP_HW + pw) * 128));
                        fa[1][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)((a1 ^ (kk << 5)) + (ph * KP_HW + pw) * 128));
                    }
                }
                if (has_prev && c == 0 && pos < 8) epilogue_chunk(pos);
                {
                    // DMA share of this read phase: weight pieces (waves 0-3: of W(g+2), waves 4-7: of W(g+3); KP_INM of the wave's two go
                    // out in the MFMA phase instead) and, in steps 1..6 of a slice, one piece of the next slice's halo
                    if (late) {
                        if (gstep + 3 < Gtot) {
                            constexpr int p3 = (pos + 3) % 9;
                            if (KP_INM < 2) weight_piece(p3, pos + 3 >= 9 ? cnx : c, (gstep + 3) & 3, 0);
                            if (KP_INM < 1) weight_piece(p3, pos + 3 >= 9 ? cnx : c, (gstep + 3) & 3, 1);
                        }
                    } else if (gstep + 2 < Gtot) {
                        constexpr int p2 = (pos + 2) % 9;
                        if (KP_INM < 2) weight_piece(p2, pos + 2 >= 9 ? cnx : c, (gstep + 2) & 3, 0);
                        if (KP_INM < 1) weight_piece(p2, pos + 2 >= 9 ? cnx : c, (gstep + 2) & 3, 1);
                    }
                    if (pos >= 1 && pos <= 6 && more) halo_piece(nsrc, (int)nhb, pos - 1);
                }
                if (late) {
                    // this wave's pieces of W(g+1) (issued in R(g-2), before that phase's halo piece) have landed: what may still be in flight
                    // is everything issued after them
                    const int rem = Gtot - 1 - gstep;
                    // h(q) = 1 when a halo piece is issued in R(q) (q = 1..6).  Issue order of a step: R: [weight pieces issued in R, halo
                    // piece], M: [weight pieces issued in M].  KP_INM = 0: the pieces of W(g+1) sit in R(g-2) in FRONT of that phase's halo
                    // piece, which may therefore still be in flight (except in R(8): the next slice is read two slots from here).
                    constexpr int h0 = (pos >= 1 && pos <= 6) ? 1 : 0, h1 = (pos >= 2 && pos <= 7) ? 1 : 0, h2 = (pos >= 3 && pos <= 8) ? 1 : 0;
                    constexpr int inR = 2 - KP_INM;       // pieces of W(g+3) already issued in this phase
                    constexpr int nfull = KP_INM == 0 ? (pos == 8 ? 4 : 4 + h2 + h1 + h0) : 2 + inR + h1 + h0;
                    constexpr int nlast = 2;              // rem == 2: W(g+2) is the last group (the tail issues no halo)
                    if (KP_ABL & 4) { }
                    else if (rem >= 3) { if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nfull) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 + inR) : "memory"); }
                    else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nlast) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                // ================= MFMA phase =================
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                KP_STAMP(5);
                __builtin_amdgcn_sched_barrier(0);
                if (KP_SPLIT) {
                    // the second half of the step's fragments is read here, in the shadow of the first half's MFMAs (the slot and the halo
                    // buffer stay valid through this phase: their next writers are two barriers away)
                    int a0 = ad[0][pw] + (int)hb, a1 = ad[1][pw] + (int)hb, b0 = bd[0] + (int)wsl, b1 = bd[1] + (int)wsl;
                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
#pragma unroll
                    for (int kk = 2; kk < 4; ++kk) {
                        fb[0][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)(b0 ^ (kk << 5)));
                        fb[1][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)(b1 ^ (kk << 5)));
                        fa[0][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)((a0 ^ (kk << 5)) + (ph * KP_HW + pw) * 128));
                        fa[1][kk] = *reinterpret_cast<const bf16x8_t*>(lds + (unsigned)((a1 ^ (kk << 5)) + (ph * KP_HW + pw) * 128));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (KP_SPLIT && kk == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][kk], fa[i][kk], acc[j][i], 0, 0, 0);
                    if (KP_INM >= 1 && (kk == 0 || (kk == 2 && KP_INM >= 2))) {
                        // part of the wave's weight DMA rides in the shadow of its own MFMAs (same order of issue as in the read phase)
                        __builtin_amdgcn_sched_barrier(0);
                        const int pp = kk == 0 ? (KP_INM >= 2 ? 0 : 1) : 1;
                        if (late) {
                            if (gstep + 3 < Gtot) weight_piece((pos + 3) % 9, pos + 3 >= 9 ? cnx : c, (gstep + 3) & 3, pp);
                        } else if (gstep + 2 < Gtot) {
                            weight_piece((pos + 2) % 9, pos + 2 >= 9 ? cnx : c, (gstep + 2) & 3, pp);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                KP_STAMP(6);
                if (!late) {
                    // waves 0-3: pieces of W(g+1), issued in step g-1; behind them: (KP_INM = 0 only) that step's halo piece, this step's
                    // two weight pieces and halo piece
                    const int rem = Gtot - 1 - gstep;
                    constexpr int h0 = (pos >= 1 && pos <= 6) ? 1 : 0, h1 = (pos >= 2 && pos <= 7) ? 1 : 0;
                    constexpr int nfull = KP_INM == 0 ? 2 + h1 + h0 : 2 + h0;
                    if (KP_ABL & 4) { }
                    else if (rem >= 2) { if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(nfull) : "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                KP_STAMP(7);
                ++gstep;
            };
            do_step(std::integral_constant<int, 0>{}); do_step(std::integral_constant<int, 1>{}); do_step(std::integral_constant<int, 2>{});
            do_step(std::integral_constant<int, 3>{}); do_step(std::integral_constant<int, 4>{}); do_step(std::integral_constant<int, 5>{});
            do_step(std::integral_constant<int, 6>{}); do_step(std::integral_constant<int, 7>{}); do_step(std::integral_constant<int, 8>{});
            hb = nhb;
        }
        // tile done: its accumulators move to the second set (stored during the next tile's first steps, or below)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) prev[j][i] = acc[j][i];
        tile_bases(cur);
        has_prev = true;
        init_acc();
        cur = nxt;
    }
    if (!late) __builtin_amdgcn_s_barrier();
    if (has_prev) {
#pragma unroll
        for (int e = 0; e < 8; ++e) epilogue_chunk(e);
    }
}

// caller guarantees: 3x3 / stride 1 / pad 1 geometry in class 0 of g (normal or mirrored taps), OH == H, OW == W
bool conv3x3_kp_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    // EXPERIMENTAL (round 4): correct on every tested shape and 5-7 % faster than the halo-tile kernel alone (0.32 / 0.355 / 0.375 of
    // the MFMA peak on 128 -> 128 @64^2 / 256 -> 256 @32^2 / 512 -> 512 @16^2 vs 0.30 / 0.33 / 0.355), but it has no BatchNorm
    // statistics hook, which costs a training step about what the kernel wins — so it is opt-in: CN_ENABLE_CONV_KP=1 (read per call:
    // the tests switch it inside one process).  DESIGN.md section 6b has the ablation table (the phase-staggered skeleton alone —
    // MFMAs, barriers, epilogue — runs at 0.47: one wave per SIMD in its MFMA phase issues a 32x32x16 MFMA every 37.5 cycles, not 32,
    // and every slot ends in a ~130-cycle barrier turnover).
    const bool disabled = getenv("CN_ENABLE_CONV_KP") == nullptr || getenv("CN_DISABLE_CONV_KP") != nullptr;
    static int cus = 0;
    if (disabled || dtype != CN_BF16 || g.Ci < 128 || (g.Ci & 63) || (g.x_ld & 7) || (g.H & 15) || (g.W & 15) || g.nsrc != 0 || g.dcn_x != nullptr) return false;
    if (g.res32 != nullptr || g.y_f32 || (g.y_ld & 7) || (g.res != nullptr && (g.res_ld & 7))) return false;
    if ((reinterpret_cast<uintptr_t>(g.x) | reinterpret_cast<uintptr_t>(g.w) | reinterpret_cast<uintptr_t>(g.y) | reinterpret_cast<uintptr_t>(g.res)) & 15) return false;
    if ((int64_t)g.co_pad * g.ktot * 2 >= (1ll << 31)) return false;          // 32-bit weight offsets
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
    const int nblk = (g.Co + 127) / 128;
    if (nblk * 128 - g.Co >= 64) return false;          // half of the last 128-channel block would be wasted MFMAs: the halo-tile kernel's 64-wide tile
    const int group = 8 * nblk;
    const int tiles_h = g.H / 16, tiles_w = g.W / 16;
    const int64_t tiles = (int64_t)g.N * tiles_h * tiles_w;
    int grid = (cus / group) * group;
    grid = (int)std::min<int64_t>(grid, (tiles + 7) / 8 * group);
    const char* force = getenv("CN_CONV_KP_FORCE");       // tests: fewer workgroups -> several tiles per workgroup on small problems
    if (force && grid) grid = std::min(grid, std::max(group, atoi(force) / group * group));
    if (grid <= 0) return false;
    const int res = g.res == nullptr ? 0 : (g.relu == 2 ? 2 : 1);
    const dim3 gr(grid), bl(KP_NT);
    const char* mode = getenv("CN_CONV_KP_MODE");        // "stagger": the phase-staggered form; default: the lockstep form
    const bool lock = !(mode && mode[0] == 's');
#define KP_GO(RES_, RELU_) do { if (lock) hipLaunchKernelGGL((conv3x3_kp_kernel<RES_, RELU_, true>), gr, bl, 0, st, g, nblk, tiles_h, tiles_w, wmap); \
                               else hipLaunchKernelGGL((conv3x3_kp_kernel<RES_, RELU_, false>), gr, bl, 0, st, g, nblk, tiles_h, tiles_w, wmap); } while (0)
    if (res == 0) { if (g.relu == 1) KP_GO(0, true); else KP_GO(0, false); }
    else if (res == 1) { if (g.relu == 1) KP_GO(1, true); else KP_GO(1, false); }
    else KP_GO(2, false);
#undef KP_GO
    return true;
}
