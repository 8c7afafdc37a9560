// DCNv2 blend-matrix kernels, second skeleton (round 6): 16x16-pixel tiles, eight waves per workgroup, FOUR waves per SIMD.
//
// The round-3..5 family (dcn_bm.hip) keeps a pixel group's 12x16 sampling window as MFMA fragments in registers (96 VGPRs) and
// parks a geometry table per wave in LDS: 250-256 VGPRs and 79 KB of LDS per four-wave workgroup = two waves per SIMD.  The probe of
// round 5 (tools/probe/valu_rate.hip) says what that costs: a wave issues one VALU instruction per ~4.9 cycles, a SIMD needs four
// resident waves to issue at its own rate, and every kernel of the family is a per-wave dependent chain (geometry -> blend-matrix
// rows -> blend MFMAs -> pack -> contraction MFMAs) that two waves cannot overlap.  This skeleton is built for occupancy instead:
//   * tile 16x16 pixels, 8 waves (wave = 4x8 pixel group), window margin 3 (reach: |offset| < 2 px for every pixel and tap, more
//     on three sides of most pixels; beyond -> the exact per-lane far path): the halo image is 22x22 pixels = 61 KB instead of 48 KB
//     per 8x16 tile (1.9 instead of 3.0 halo pixels loaded per output pixel), filled by LDS-DMA (no staging registers);
//   * the window fragments are read from the halo image per TOUCHED row (4 transposing reads, as dcn_wgrad_bm_kernel does), nothing
//     of the window lives in registers;
//   * no geometry table: a lane fetches the three numbers of (its pixel, tap) — dy, dx, mask logit — from global memory one tap
//     ahead (inline-asm loads with counted waits: the compiler sees no load in the tap loop and emits no vmcnt(0) of its own);
//   * weights: one 16-byte slot per thread and tap, global -> registers -> LDS double buffer, one block barrier per tap.
// <= 128 VGPRs and 77.4 KB of LDS per 512-thread workgroup: two workgroups = 16 waves per CU.
#include "conv_common.h"
#include <stdlib.h>

#define B2_TH 16
#define B2_TW 16
#define B2_MG 3
#define B2_HR (B2_TH + 2 * B2_MG)        // 22 halo rows
#define B2_HC (B2_TW + 2 * B2_MG)        // 22 halo columns
#define B2_GR 10                         // window rows of one 4x8 group (4 + 2 * 3)
#define B2_PIXB 128                      // bytes per halo pixel (64 bf16)
#define B2_NPIECE 61                     // 1 KB LDS-DMA pieces of the halo image (484 pixels = 60.5 pieces; the last one half slack)
#define B2_XB (B2_NPIECE * 1024)
#define B2_WSB 8192                      // bytes per weight buffer: [4 k-steps][2 halves][64 co][8] bf16

#ifdef B2_PROBE   // development build only (tools/dev/b2_probe.py): cycle stamps of wave B2_PROBE_WAVE of the first workgroups
#ifndef B2_PROBE_WAVE
#define B2_PROBE_WAVE 0
#endif
__device__ unsigned long long b2_ts[2048 * 48];
#define B2_STAMP(k) do { if (threadIdx.x == 64 * B2_PROBE_WAVE && blockIdx.x < 2048) b2_ts[blockIdx.x * 48 + (k)] = clock64(); } while (0)
extern "C" int b2_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(b2_ts), sizeof(b2_ts)); }
#else
#define B2_STAMP(k) do { } while (0)
#endif
__device__ uint4 b2_zero_page[8];        // 128 zero bytes: DMA source of everything outside the image

typedef short b2_s16x4 __attribute__((ext_vector_type(4)));
typedef short b2_s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t b2_u32x4 __attribute__((ext_vector_type(4)));

// LDS byte address of (halo row, halo column, channel c): 128 B per pixel, the two 64-byte halves swapped on every other column
// pair (any four consecutive columns of a transposing read then cover the 64 banks exactly once, whatever the first column)
__device__ static inline int b2_ofs(int hr, int hc, int c) {
    return (hr * B2_HC + hc) * B2_PIXB + ((((c >> 5) ^ (hc >> 1)) & 1) << 6) + (c & 31) * 2;
}

struct B2Geom {
    const bf16_t* x; const float* om; const bf16_t* wp; const float* bias; bf16_t* y;
    int N, H, W, x_ld, y_ld, ktot, Co, relu, Ci, tiles_w, tiles_img;
    float* bn_part; int bn_slots;
};

// one 1 KB piece of the halo image by LDS-DMA: lane l fills physical 16-byte chunk l & 7 of pixel 8 J + (l >> 3)
__device__ static inline void b2_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

template <bool MB>   // MB: more than one 64-channel block of x (Ci = 128 / 256), one after the other into the same accumulators
__global__ __launch_bounds__(512, 4) void dcn_fwd_b2_kernel(const B2Geom g) {
    CN_MAIN_PRIO_SET();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xw = smem;                                   // halo image [22][22] pixels x 128 B
    unsigned char* const Ws = smem + B2_XB;                           // 2 weight buffers
    b2_u32x4* const Lut = reinterpret_cast<b2_u32x4*>(Ws + 2 * B2_WSB);   // [23] v_perm selectors (see dcn_fwd_bm_kernel)
    if (threadIdx.x < 23) {
        const int c = (int)threadIdx.x - 8;
        b2_u32x4 sel;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int sft = c - 2 * d;
            sel[d] = sft == 0 ? 0x03020100u : (sft == 1 ? 0x01000c0cu : (sft == -1 ? 0x0c0c0302u : 0x0c0c0c0cu));
        }
        Lut[threadIdx.x] = sel;
    }
    B2_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // consecutive tiles share halo rows / columns: keep them on one XCD (workgroups are dealt round-robin to the 8 XCDs)
    const int G = gridDim.x;
    const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int n = lb / g.tiles_img, rt = lb - n * g.tiles_img;
    const int ty0 = (rt / g.tiles_w) * B2_TH, tx0 = (rt % g.tiles_w) * B2_TW;
    const int64_t img = (int64_t)n * g.H * g.W;
    const bf16_t* __restrict__ X = g.x + img * g.x_ld;

    const int grow = (wave >> 1) * 4, gcol = (wave & 1) * 8;       // this wave's pixel group inside the tile
    const int wxo = gcol ? 5 : 3;                                   // its 16-column window starts wxo columns left of the group (halo column gcol + 3 - wxo)
    const int nl = lane & 31, hh = lane >> 5;
    const int prow = nl >> 3, pcol = nl & 7;
    const int gy = ty0 + grow + prow, gx = tx0 + gcol + pcol;
    const bool live = gy < g.H && gx < g.W;
    const float* const po = g.om + (img + (int64_t)(live ? gy : 0) * g.W + (live ? gx : 0)) * 32;   // this lane's row of offsets / mask logits

    const int nblk = MB ? g.Ci >> 6 : 1;
    const int nstep = 9 * nblk;

    // ---- LDS-DMA of the halo image of channel block `blk`: rows ty0-3 .. ty0+18, columns tx0-3 .. tx0+18, zeros outside the image ----
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue_halo = [&](int blk) {
        const char* const Xb = reinterpret_cast<const char*>(X + blk * 64);
        const char* const zp = reinterpret_cast<const char*>(b2_zero_page) + (lane & 7) * 16;
        const int pl = lane >> 3, q = lane & 7;
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const int J = i * 8 + wave_s;
            if (J >= B2_NPIECE) break;
            const int pix = J * 8 + pl, r = pix / B2_HC, c = pix - r * B2_HC;
            const int hy = ty0 - B2_MG + r, hx = tx0 - B2_MG + c;
            const bool ok = pix < B2_HR * B2_HC && (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
            const int ql = q ^ (((c >> 1) & 1) << 2);       // the logical chunk stored at physical chunk q (b2_ofs)
            const char* src = ok ? Xb + (((int64_t)hy * g.W + hx) * g.x_ld + ql * 8) * 2 : zp;
            b2_dma16(src, lds_base + (unsigned)(J * 1024));
        }
    };
    // ---- weight slot of this thread: global (mode-1 pack [co][tap*Ci + ci]) -> two 8-byte registers -> LDS in fragment order ----
    // slot = (s, h, co): 8 bf16 = W[co][ci(s,h,e)], ci = 32*(s>>1) + 16*(s&1) + 8*(e>>2) + 4*h + (e&3)   (the order S^T's registers have)
    const int w_co = tid & 63, w_sh = tid >> 6;
    const bf16_t* const wrow = g.wp + (int64_t)w_co * g.ktot + 32 * (w_sh >> 2) + 16 * ((w_sh >> 1) & 1) + 4 * (w_sh & 1);
    uint64_t wq0, wq1;
    auto w_issue = [&](int step) {       // step = blk * 9 + tap
        const int blk = MB ? step / 9 : 0, tap = step - 9 * blk;
        const bf16_t* p = wrow + tap * g.Ci + blk * 64;
        asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16" : "=&v"(wq0), "=&v"(wq1) : "v"(p) : "memory");
    };
    auto w_store = [&](int buf) {
        *reinterpret_cast<b2_u32x4*>(Ws + buf * B2_WSB + tid * 16) = b2_u32x4{(uint32_t)wq0, (uint32_t)(wq0 >> 32), (uint32_t)wq1, (uint32_t)(wq1 >> 32)};
    };
    // ---- (dy, dx, mask logit) of (own pixel, tap) ----
    uint64_t on_pos; uint32_t on_m;
    auto om_issue = [&](int tap) {
        asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(on_pos), "=&v"(on_m) : "v"(po + 2 * tap), "v"(po + 18 + tap) : "memory");
    };

    issue_halo(0);
    w_issue(0);
    om_issue(0);
    // the bias is the accumulators' initial value (lane = pixel, register v of block cb = channel 32 cb + 8 (v >> 2) + 4 hh + (v & 3))
    f32x16_t acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4*>(g.bias + 32 * cb + 8 * q + 4 * hh);
            acc[cb][4 * q] = bv.x; acc[cb][4 * q + 1] = bv.y; acc[cb][4 * q + 2] = bv.z; acc[cb][4 * q + 3] = bv.w;
        }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wq0), "+v"(wq1), "+v"(on_pos), "+v"(on_m) :: "memory");
    w_store(0);
    float oc_y = __uint_as_float((uint32_t)on_pos), oc_x = __uint_as_float((uint32_t)(on_pos >> 32)), oc_m = __uint_as_float(on_m);
    __syncthreads();        // halo image, first weight slice, selector table
    B2_STAMP(1);

    // base addresses of the transposing fragment reads (see dcn_fwd_bm_kernel): lane (r16, g16) addresses 4 channels of halo column
    // wc0 (+4: second read); rows are added per touched row
    const int r16 = lane & 15, g16 = lane >> 4;
    typedef __attribute__((address_space(3))) b2_s16x4* lds_ptr;
    const int wc0 = gcol + B2_MG - wxo + 8 * (g16 >> 1) + (r16 >> 2);
    const unsigned char* const b0 = Xw + b2_ofs(grow, wc0, 16 * (g16 & 1) + 4 * (r16 & 3));
    const unsigned char* const b1 = Xw + b2_ofs(grow, wc0, 32 + 16 * (g16 & 1) + 4 * (r16 & 3));
    const float fy0 = (float)(prow + 2), fx0 = (float)(pcol + wxo - 1);     // window position of the pixel itself, minus one (tap (0, 0) = -1)

#pragma unroll 1
    for (int step = 0; step < nstep; ++step) {
        const int blk = MB ? step / 9 : 0, tap = step - 9 * blk;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        // next step's weights and geometry numbers on their way (consumed at the bottom of this iteration)
        {
            const int ns = step + 1 < nstep ? step + 1 : step;
            w_issue(ns);
            const int nb = MB ? ns / 9 : 0;
            om_issue(ns - 9 * nb);
        }
        B2_STAMP(2 + 4 * (step % 9));
        // ---- geometry of (own pixel, tap): window position of corner 00 and the two packed weight pairs ----
        const float pyr = oc_y + (fy0 + (float)ky), pxr = oc_x + (fx0 + (float)kx);
        const float m = live ? __builtin_amdgcn_rcpf(1.f + __expf(-oc_m)) : 0.f;
        const float fy = floorf(pyr), fx = floorf(pxr);
        const int wr = (int)fy, wc = (int)fx;
        const float ly = pyr - fy, lx = pxr - fx;
        const float wa = (1.f - ly) * m, wbt = ly * m;
        const bool inwin = (unsigned)wr <= (unsigned)(B2_GR - 2) && (unsigned)wc <= 14u;
        const bool far = !inwin && m != 0.f;
        const uint32_t P0 = inwin ? pk_bf16(wa * (1.f - lx), wa * lx) : 0u, P1 = inwin ? pk_bf16(wbt * (1.f - lx), wbt * lx) : 0u;
        const int wr_top = (P0 != 0u) ? wr : -1, wr_bot = (P1 != 0u) ? wr + 1 : -1;
        const b2_u32x4 sel = Lut[min(max(wc - 8 * hh + 8, 0), 22)];
        uint32_t V0[4], V1[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            V0[d] = __builtin_amdgcn_perm(P0, P0, sel[d]);
            V1[d] = __builtin_amdgcn_perm(P1, P1, sel[d]);
        }
        uint32_t rows = ((P0 != 0u) ? (1u << (wr & 15)) : 0u) | ((P1 != 0u) ? (2u << (wr & 15)) : 0u);
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x141, 0xF, 0xF, true);    // row_half_mirror
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x140, 0xF, 0xF, true);    // row_mirror
        uint32_t rowmask = (uint32_t)__builtin_amdgcn_readlane((int)rows, 0) | (uint32_t)__builtin_amdgcn_readlane((int)rows, 16) |
                           (uint32_t)__builtin_amdgcn_readlane((int)rows, 32) | (uint32_t)__builtin_amdgcn_readlane((int)rows, 48);
        const uint64_t farmask = __builtin_amdgcn_ballot_w64(far);
        B2_STAMP(3 + 4 * (step % 9));

        if (rowmask != 0u || farmask != 0ull) {
            // ---- S^T[ci][p] = sum over the touched window rows (K = the row's 16 source columns), fragments straight from the halo image ----
            f32x16_t st[2];
            // fragments of window row r (4 transposing reads) and this lane's 8 blend-matrix entries for it
            auto row_read = [&](int r, bf16x8_t& x0, bf16x8_t& x1) {
                const int ro_ = r * (B2_HC * B2_PIXB);
                const b2_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b0 + ro_));
                const b2_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b0 + ro_ + 4 * B2_PIXB));
                const b2_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b1 + ro_));
                const b2_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b1 + ro_ + 4 * B2_PIXB));
                const b2_s16x8 v0 = {l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
                const b2_s16x8 v1 = {l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
                x0 = __builtin_bit_cast(bf16x8_t, v0); x1 = __builtin_bit_cast(bf16x8_t, v1);
            };
            auto row_blend = [&](int r) {
                const bool t0 = wr_top == r, t1 = wr_bot == r;
                b2_u32x4 b;
#pragma unroll
                for (int d = 0; d < 4; ++d) b[d] = t0 ? V0[d] : (t1 ? V1[d] : 0u);
                return __builtin_bit_cast(bf16x8_t, b);
            };
            if (rowmask != 0u) {
                // the first touched row starts the accumulators (C = 0 is an inline constant: no 32 v_mov per tap).  (Requesting the next
                // row's fragments before this row's MFMAs was tried: 8 more live registers spill at the 128-register cap, 280 -> 329 us.)
                const int r = __builtin_ctz(rowmask);
                rowmask &= rowmask - 1;
                bf16x8_t x0, x1;
                row_read(r, x0, x1);
                const bf16x8_t bf = row_blend(r);
                f32x16_t z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0, bf, z, 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, bf, z, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) { st[0][i] = 0.f; st[1][i] = 0.f; }
            }
#pragma unroll 1
            while (rowmask) {
                const int r = __builtin_ctz(rowmask);
                rowmask &= rowmask - 1;
                bf16x8_t x0, x1;
                row_read(r, x0, x1);
                const bf16x8_t bf = row_blend(r);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0, bf, st[0], 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, bf, st[1], 0, 0, 0);
            }
            B2_STAMP(4 + 4 * (step % 9));
            // ---- samples that leave the window (rare): exact fp32 VALU blend from global memory, every lane its own 32 channels
            //      (register v of block mb = channel 32 mb + 8 (v >> 2) + 4 hh + (v & 3)) ----
            if (farmask != 0ull) {
                if (far) {
                    const int h0 = wr + (ty0 + grow - B2_MG), w0 = wc + (tx0 + gcol - wxo);
                    const bool h0ok = (unsigned)h0 < (unsigned)g.H, h1ok = (unsigned)(h0 + 1) < (unsigned)g.H;
                    const bool w0ok = (unsigned)w0 < (unsigned)g.W, w1ok = (unsigned)(w0 + 1) < (unsigned)g.W;
                    const float w00 = (h0ok && w0ok) ? wa * (1.f - lx) : 0.f, w01 = (h0ok && w1ok) ? wa * lx : 0.f;
                    const float w10 = (h1ok && w0ok) ? wbt * (1.f - lx) : 0.f, w11 = (h1ok && w1ok) ? wbt * lx : 0.f;
                    const int hc0 = min(max(h0, 0), g.H - 1), hc1 = min(max(h0 + 1, 0), g.H - 1);
                    const int wc0_ = min(max(w0, 0), g.W - 1), wc1_ = min(max(w0 + 1, 0), g.W - 1);
                    const bf16_t* p00 = X + ((int64_t)hc0 * g.W + wc0_) * g.x_ld + blk * 64 + 4 * hh;
                    const bf16_t* p01 = X + ((int64_t)hc0 * g.W + wc1_) * g.x_ld + blk * 64 + 4 * hh;
                    const bf16_t* p10 = X + ((int64_t)hc1 * g.W + wc0_) * g.x_ld + blk * 64 + 4 * hh;
                    const bf16_t* p11 = X + ((int64_t)hc1 * g.W + wc1_) * g.x_ld + blk * 64 + 4 * hh;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int co_ = 32 * mb + 8 * q;
                            const uint2 a = *reinterpret_cast<const uint2*>(p00 + co_), b = *reinterpret_cast<const uint2*>(p01 + co_);
                            const uint2 cc = *reinterpret_cast<const uint2*>(p10 + co_), d = *reinterpret_cast<const uint2*>(p11 + co_);
                            st[mb][4 * q] += __uint_as_float(a.x << 16) * w00 + __uint_as_float(b.x << 16) * w01 + __uint_as_float(cc.x << 16) * w10 + __uint_as_float(d.x << 16) * w11;
                            st[mb][4 * q + 1] += __uint_as_float(a.x & 0xffff0000u) * w00 + __uint_as_float(b.x & 0xffff0000u) * w01 + __uint_as_float(cc.x & 0xffff0000u) * w10 + __uint_as_float(d.x & 0xffff0000u) * w11;
                            st[mb][4 * q + 2] += __uint_as_float(a.y << 16) * w00 + __uint_as_float(b.y << 16) * w01 + __uint_as_float(cc.y << 16) * w10 + __uint_as_float(d.y << 16) * w11;
                            st[mb][4 * q + 3] += __uint_as_float(a.y & 0xffff0000u) * w00 + __uint_as_float(b.y & 0xffff0000u) * w01 + __uint_as_float(cc.y & 0xffff0000u) * w10 + __uint_as_float(d.y & 0xffff0000u) * w11;
                            __builtin_amdgcn_sched_barrier(0);      // one (block, quad) at a time: the rare path must not raise the kernel's register count
                        }
                }
            }
            // ---- y^T[co][p] += W_k^T[co][ci] S^T[ci][p]: S^T's registers ARE the B operand (k-step s = registers 8*(s&1)..+7 of block s>>1) ----
            const unsigned char* wb = Ws + (step & 1) * B2_WSB;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                b2_u32x4 sb;
#pragma unroll
                for (int d = 0; d < 4; ++d) sb[d] = pk_bf16(st[s >> 1][8 * (s & 1) + 2 * d], st[s >> 1][8 * (s & 1) + 2 * d + 1]);
                const bf16x8_t sf = __builtin_bit_cast(bf16x8_t, sb);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const b2_u32x4 wv = *reinterpret_cast<const b2_u32x4*>(wb + (((s * 2 + hh) * 64) + cb * 32 + nl) * 16);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv), sf, acc[cb], 0, 0, 0);
                }
            }
        }
        B2_STAMP(5 + 4 * (step % 9));
        // ---- bottom: the next step's weights into the other buffer (last read a step ago, a barrier ago), its geometry numbers ----
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(wq0), "+v"(wq1) :: "memory");      // younger: the two om loads
        w_store((step + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(on_pos), "+v"(on_m) :: "memory");
        oc_y = __uint_as_float((uint32_t)on_pos); oc_x = __uint_as_float((uint32_t)(on_pos >> 32)); oc_m = __uint_as_float(on_m);
        if (MB && tap == 8 && blk + 1 < nblk) {
            // next 64-channel block of x: everybody is done with the halo image after this barrier
            __syncthreads();
            issue_halo(blk + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    B2_STAMP(38);
    // ---- epilogue: lane = pixel, registers = 4 consecutive channels per (block, quad): ReLU, through the wave's slice of the dead halo
    //      image ([32 px][72] bf16) so that a pixel's 128 bytes leave as eight 16-byte lanes; BN statistics of the stored values ----
    {
        unsigned char* const Y = Xw + wave * 4608;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                uint2 o;
                o.x = pk_bf16(v[0], v[1]); o.y = pk_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(Y + nl * 144 + (32 * cb + 8 * q + 4 * hh) * 2) = o;
            }
        __builtin_amdgcn_wave_barrier();
        constexpr int CPP = 8;                       // 16-byte chunks per pixel
        const bool stats = g.bn_part != nullptr;
        float s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 32 * CPP / 64; ++i) {
            const int idx = lane + 64 * i, p = idx / CPP, ch = idx % CPP;
            const int oy = ty0 + grow + (p >> 3), ox = tx0 + gcol + (p & 7);
            const b2_u32x4 o = *reinterpret_cast<const b2_u32x4*>(Y + p * 144 + ch * 16);
            if (oy < g.H && ox < g.W) {
                *reinterpret_cast<b2_u32x4*>(g.y + (img + (int64_t)oy * g.W + ox) * g.y_ld + ch * 8) = o;
                if (stats) { const uint32_t w[4] = {o[0], o[1], o[2], o[3]}; bn_stat_add(s0, s1, w); }
            }
        }
        if (stats)      // (the flush barriers first: every wave is done with its slice of the halo image, which becomes the scratch)
            bn_stats_flush<CPP, 512>(s0, s1, reinterpret_cast<float*>(Xw), g.bn_part, g.bn_slots, g.y_ld, 0, g.Co, (unsigned)lb, tid);
    }
    B2_STAMP(39);
}

bool dcn_fwd_b2_shape_ok(int Ci, int x_ld, int Co, int y_ld, int om_ld, int H, int W) {
    static const bool disabled = getenv("CN_DISABLE_DCN_B2") != nullptr || getenv("CN_DISABLE_DCN_FWD_B2") != nullptr;
    // 16x16 tiles: measured ahead of the 8x16 kernel down to 32x32 maps at batch 64 (256->64 @32^2: 74 vs 88 us); 16x16 maps stay there
    static const int min_hw = [] { const char* e = getenv("CN_DCN_B2_MIN_HW"); return e ? atoi(e) : 32 * 32; }();
    return !disabled && (Ci == 64 || Ci == 128 || Ci == 256) && x_ld == Ci && om_ld == 32 && Co == 64 && y_ld == Co && H * W >= min_hw;
}

// returns false when the shape is not handled here (caller falls back to dcn_fwd_bm_kernel / the gather kernels)
bool dcn_fwd_b2_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                       int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st) {
    if (!dcn_fwd_b2_shape_ok(Ci, x_ld, Co, y_ld, om_ld, H, W) || bias == nullptr || ktot != 9 * Ci) return false;
    if (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y | (uintptr_t)om | (uintptr_t)bias) & 15) return false;
    B2Geom g;
    g.x = (const bf16_t*)x; g.om = om; g.wp = (const bf16_t*)wp; g.bias = bias; g.y = (bf16_t*)y;
    g.N = N; g.H = H; g.W = W; g.x_ld = x_ld; g.y_ld = y_ld; g.ktot = ktot; g.Co = Co; g.relu = relu; g.Ci = Ci;
    g.tiles_w = (W + B2_TW - 1) / B2_TW;
    g.tiles_img = g.tiles_w * ((H + B2_TH - 1) / B2_TH);
    const int64_t tiles = (int64_t)g.tiles_img * N;
    if (tiles > 0x7fffffff) return false;
    g.bn_part = bn_part; g.bn_slots = bn_slots;
    if (bn_part) mark_taken(bn_taken);
    const size_t smem = (size_t)B2_XB + 2 * B2_WSB + 23 * 16;
    if (Ci > 64) {
        (void)hipFuncSetAttribute((const void*)dcn_fwd_b2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_fwd_b2_kernel<true>, dim3((unsigned)tiles), dim3(512), smem, st, g);
    } else {
        (void)hipFuncSetAttribute((const void*)dcn_fwd_b2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_fwd_b2_kernel<false>, dim3((unsigned)tiles), dim3(512), smem, st, g);
    }
    return true;
}

