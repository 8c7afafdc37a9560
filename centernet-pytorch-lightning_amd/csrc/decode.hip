// Decode path (utils/decode.py, decode/ctdet.py, decode/multi_pose.py): 3x3 max-pool pseudo-NMS, top-K, gathers.
//
// Top-K is an exact radix SELECT (4 x 8-bit passes over order-preserving integer keys held in LDS) followed by a
// rank sort of the K survivors, so the result is a total order: score descending, ties -> lower index first
// (torch.topk leaves tie order unspecified; tests/ pin bit-exactness on tie-free inputs and the tie rule on ties).
// One workgroup (1024 threads) owns one (batch, class) map: the map is read from HBM exactly once (16-byte loads),
// the peak test and all select passes run out of LDS, and the only HBM writes are K (score, index) pairs.
#include "common.h"
#include "topk_stream.h"
#include <stdlib.h>

#define TK_THREADS 1024
#define TK_MAXK 256

__device__ static inline uint32_t f2key(float v) {
    v = v + 0.0f;  // -0.0 -> +0.0
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ static inline float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

struct __attribute__((aligned(16))) TkShared {   // sizeof is a multiple of 16 so the dynamic LDS behind it stays 16-B aligned
    uint32_t hist[256];
    uint32_t prefix, mask, need, cnt_g, cnt_e, n_eq, idx_thr, pad_;
    uint32_t wsum[4];
    uint32_t sel_key[TK_MAXK];
    uint32_t sel_idx[TK_MAXK];
};
static_assert(sizeof(TkShared) % 16 == 0, "TkShared must keep the dynamic LDS base aligned");

// Exact top-K of keys[0..L) (LDS) by (key desc, index asc).  Results (sorted) land in sh.sel_key / sh.sel_idx.
// All threads of the workgroup must call this.  K <= min(L, TK_MAXK).
__device__ static void block_topk(const uint32_t* keys, int L, int K, TkShared& sh) {
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    if (tid == 0) { sh.prefix = 0; sh.mask = 0; sh.need = K; sh.cnt_g = 0; sh.cnt_e = 0; sh.n_eq = 0; sh.idx_thr = 0xffffffffu; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += nthr) sh.hist[i] = 0;
        __syncthreads();
        // sh.need is read HERE, a barrier away from where the crossing thread rewrites it below.  (It used to be read in that
        // same phase: a wave that ran late saw the already-reduced count, found a second "crossing" bin and overwrote
        // prefix / need — a wrong threshold, more than K survivors, sel_* overrun in LDS and garbage indices.  It only ever
        // happened with the decode running next to another stream's kernels: 1 in ~10^6 maps.)
        const uint32_t prefix = sh.prefix, mask = sh.mask, need = sh.need;
        for (int i0 = 0; i0 < L; i0 += nthr) {
            const int i = i0 + tid;
            const bool ok = i < L && ((keys[i] & mask) == prefix);
            const uint32_t bin = ok ? ((keys[i] >> shift) & 255u) : 0xffffffffu;
            // Histogram update, robust to both extremes: a wave whose keys share a few bins (the zeros left by the
            // pseudo-NMS, a flat map, one exponent) aggregates them — one LDS atomic per DISTINCT bin, up to four bins in the first
            // pass and the leading bin in the later ones —
            // and whatever is left (survivors spread over many bins) adds itself directly.
            unsigned long long todo = __ballot(ok);
            bool mine = ok;
#pragma unroll 1
            for (int rep = 0; rep < (shift == 24 ? 4 : 1) && todo; ++rep) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t lb = __shfl(bin, leader, 64);
                const unsigned long long same = __ballot(mine && bin == lb);
                if (lane == leader) atomicAdd(&sh.hist[lb], (uint32_t)__popcll(same));
                if (bin == lb) mine = false;
                todo &= ~same;
            }
            if (mine) atomicAdd(&sh.hist[bin], 1u);
        }
        __syncthreads();
        // which bin holds the need-th largest key: descending inclusive scan of the histogram by 256 threads (a serial
        // walk by one thread cost ~12 us per pass, 4 passes per map, 5120 maps per batch)
        uint32_t h = 0, incl = 0;
        if (tid < 256) {
            h = sh.hist[255 - tid];
            incl = h;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (lane == 63) sh.wsum[tid >> 6] = incl;   // waves 0..3 own bins 255..192, 191..128, 127..64, 63..0
        }
        __syncthreads();
        if (tid < 256) {
            for (int w = 0; w < (tid >> 6); ++w) incl += sh.wsum[w];
            if (incl >= need && incl - h < need) {      // exactly one thread: the crossing bin
                sh.need = need - (incl - h);            // how many are still needed inside that bin
                sh.prefix = prefix | ((uint32_t)(255 - tid) << shift);
            }
            if (tid == 0) sh.mask = mask | (255u << shift);
        }
        __syncthreads();
    }
    const uint32_t thr = sh.prefix;         // the K-th largest key
    const uint32_t need_eq = sh.need;       // how many elements == thr belong to the top-K
    const uint32_t n_gt = (uint32_t)K - need_eq;
    // count the elements equal to the threshold
    {
        uint32_t c = 0;
        for (int i = tid; i < L; i += nthr) c += keys[i] == thr;
        c = wave_sum_u(c);
        if (lane == 0 && c) atomicAdd(&sh.n_eq, c);
    }
    __syncthreads();
    if (sh.n_eq > need_eq) {
        // ties straddle the cut: keep the need_eq LOWEST indices among the equal keys (2 x 8-bit select on the index)
        uint32_t ipre = 0, imask = 0, ineed = need_eq;
        for (int shift = 8; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += nthr) sh.hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < L; i += nthr)
                if (keys[i] == thr && (((uint32_t)i) & imask) == ipre) atomicAdd(&sh.hist[(i >> shift) & 255], 1u);
            __syncthreads();
            uint32_t acc = 0;
            int b = 0;
            for (; b < 255; ++b) {          // every thread walks the same histogram -> same answer, no broadcast needed
                if (acc + sh.hist[b] >= ineed) break;
                acc += sh.hist[b];
            }
            ineed -= acc;
            ipre |= (uint32_t)b << shift;
            imask |= 255u << shift;
            __syncthreads();
        }
        if (tid == 0) sh.idx_thr = ipre;    // indices <= idx_thr among the equal keys are kept (L <= 65536)
        __syncthreads();
    }
    const uint32_t idx_thr = sh.idx_thr;
    for (int i = tid; i < L; i += nthr) {
        const uint32_t k = keys[i];
        if (k > thr) {
            const uint32_t s = atomicAdd(&sh.cnt_g, 1u);
            if (s < (uint32_t)K) { sh.sel_key[s] = k; sh.sel_idx[s] = i; }      // always true for a consistent threshold
        } else if (k == thr && (uint32_t)i <= idx_thr) {
            const uint32_t s = n_gt + atomicAdd(&sh.cnt_e, 1u);
            if (s < (uint32_t)K) { sh.sel_key[s] = k; sh.sel_idx[s] = i; }
        }
    }
    __syncthreads();
    // rank sort (K <= 256)
    uint32_t mk = 0, mi = 0, rank = 0;
    if (tid < K) {
        mk = sh.sel_key[tid]; mi = sh.sel_idx[tid];
        for (int j = 0; j < K; ++j) {
            const uint32_t ok = sh.sel_key[j], oi = sh.sel_idx[j];
            rank += (ok > mk) || (ok == mk && oi < mi);
        }
    }
    __syncthreads();
    if (tid < K) { sh.sel_key[rank] = mk; sh.sel_idx[rank] = mi; }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ per-class top-K
template <int EPT>
__global__ __launch_bounds__(TK_THREADS) void topk_channel_kernel(const float* __restrict__ heat, float* __restrict__ scores,
                                                                  int32_t* __restrict__ inds, int H, int W, int K,
                                                                  int apply_nms) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    __shared__ TkShared sh;
    float* fmap = reinterpret_cast<float*>(dyn);
    const int HW = H * W, tid = threadIdx.x;
    const float* src = heat + (int64_t)blockIdx.x * HW;
    if ((HW & 3) == 0) {
        for (int i = tid; i < HW / 4; i += TK_THREADS) reinterpret_cast<float4*>(fmap)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = tid; i < HW; i += TK_THREADS) fmap[i] = src[i];
    }
    __syncthreads();
    uint32_t kreg[EPT];
    // Element ownership.  Column strips (thread = one x, RPS consecutive rows) let the 3x3 max be separable — a running
    // window of three row-maxima: 3 LDS reads per row instead of 9 per element; otherwise threads stride over the map.
    const int strips = (W > 0 && TK_THREADS % W == 0) ? TK_THREADS / W : 0;
    const int RPS = strips ? (H + strips - 1) / strips : 0;
    const bool by_strip = apply_nms && strips > 0 && RPS <= EPT;
    if (by_strip) {
        const int x = tid % W, y0 = (tid / W) * RPS;
        auto rowmax = [&](int y, float& c) {
            c = -INFINITY;
            if ((unsigned)y >= (unsigned)H) return -INFINITY;
            const float* row = fmap + y * W;
            c = row[x];
            float m = c;
            if (x > 0) m = fmaxf(m, row[x - 1]);
            if (x + 1 < W) m = fmaxf(m, row[x + 1]);
            return m;
        };
        float c_prev, c_cur, c_next;
        float m_prev = rowmax(y0 - 1, c_prev), m_cur = rowmax(y0, c_cur);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            uint32_t key = 0;
            if (e < RPS) {
                const float m_next = rowmax(y0 + e + 1, c_next);
                if (y0 + e < H) {
                    const float m = fmaxf(fmaxf(m_prev, m_cur), m_next);
                    key = f2key(c_cur * (m == c_cur ? 1.f : 0.f));       // heat * keep  (utils/decode.py:9-10)
                }
                m_prev = m_cur; m_cur = m_next; c_cur = c_next;
            }
            kreg[e] = key;
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * TK_THREADS;
            uint32_t key = 0;
            if (i < HW) {
                const float v = fmap[i];
                float out = v;
                if (apply_nms) {
                    const int y = i / W, x = i - y * W;
                    float m = v;
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = y + dy;
                        if ((unsigned)yy >= (unsigned)H) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int xx = x + dx;
                            if ((unsigned)xx >= (unsigned)W) continue;
                            m = fmaxf(m, fmap[yy * W + xx]);
                        }
                    }
                    out = v * (m == v ? 1.f : 0.f);   // heat * keep  (utils/decode.py:9-10)
                }
                key = f2key(out);
            }
            kreg[e] = key;
        }
    }
    __syncthreads();
    if (by_strip) {
        const int x = tid % W, y0 = (tid / W) * RPS;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (e < RPS && y0 + e < H) dyn[(y0 + e) * W + x] = kreg[e];
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + e * TK_THREADS;
            if (i < HW) dyn[i] = kreg[e];
        }
    }
    __syncthreads();
    block_topk(dyn, HW, K, sh);
    if (tid < K) {
        scores[(int64_t)blockIdx.x * K + tid] = key2f(sh.sel_key[tid]);
        inds[(int64_t)blockIdx.x * K + tid] = (int32_t)sh.sel_idx[tid];
    }
}

// ------------------------------------------------------------------------------------------------ streaming per-class top-K
// 128x128 maps (512x512 inputs at stride 4: every BASELINE config).  Thread (x4 = tid % 32, rb = tid / 32) owns the 4 columns
// 4*x4.. of the 16 rows rb*16..: its 16 row loads plus the two halo rows are 18 independent 16-byte loads in flight (a wave's
// load is two contiguous 512-byte row segments), the vertical 3-max stays in registers and the horizontal one takes the two
// neighbour columns from the adjacent lanes.  Bytes per map: 64 KB read once (+12 % halo rows, L2 hits), 800 B written.
__device__ static inline float ts_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// SIG: `heat` holds LOGITS; the kernel applies y = clamp(1 / (1 + expf(-x)), lo, 1 - lo) — the arithmetic of sigmoid_clamp_fwd_vec_kernel,
// bit for bit — to every value it loads, so the sigmoid pass over the map (read + 2 writes of 335 MB at C3) disappears from inference
template <bool NMS, bool SIG = false>
__global__ __launch_bounds__(TS_THREADS, 3) void topk_map128_kernel(const float* __restrict__ heat, float* __restrict__ scores,
                                                                    int32_t* __restrict__ inds, int K, float lo = 0.f) {
    auto act = [&](float4 v) {
        if constexpr (SIG) {
            const float hi = 1.f - lo;
            float* e = &v.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = fminf(fmaxf(1.f / (1.f + expf(-e[j])), lo), hi);
        }
        return v;
    };
    __shared__ TsShared sh;
    const int tid = threadIdx.x, x4 = tid & 31, row0 = (tid >> 5) * 16;
    const float* src = heat + (int64_t)blockIdx.x * 16384 + x4 * 4;
    if (tid == 0) sh.ctl[3] = 0;
    uint32_t key[64];
    if (NMS) {
        float4 raw[18];
#pragma unroll
        for (int j = 0; j < 18; ++j) {             // branch-free: rows outside the map re-read a border row and are replaced by -inf
            const int r = row0 - 1 + j;
            const int rc = r < 0 ? 0 : (r > 127 ? 127 : r);
            raw[j] = act(*reinterpret_cast<const float4*>(src + rc * 128));
        }
        const float ninf = -INFINITY;
        if (row0 == 0) raw[0] = make_float4(ninf, ninf, ninf, ninf);
        if (row0 == 112) raw[17] = make_float4(ninf, ninf, ninf, ninf);
        auto hmax = [&](const float4& v) {         // max over (x-1, x, x+1) inside the row; -inf beyond the row ends (max_pool2d padding)
            float left = __shfl_up(v.w, 1, 64), right = __shfl_down(v.x, 1, 64);
            left = x4 == 0 ? ninf : left;
            right = x4 == 31 ? ninf : right;
            return make_float4(ts_max3(left, v.x, v.y), ts_max3(v.x, v.y, v.z), ts_max3(v.y, v.z, v.w), ts_max3(v.z, v.w, right));
        };
        float4 hp = hmax(raw[0]), hc = hmax(raw[1]);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 hn = hmax(raw[j + 2]);
            const float4 c = raw[j + 1];
            const float mx = ts_max3(hp.x, hc.x, hn.x), my = ts_max3(hp.y, hc.y, hn.y);
            const float mz = ts_max3(hp.z, hc.z, hn.z), mw = ts_max3(hp.w, hc.w, hn.w);
            key[4 * j + 0] = ts_f2key(c.x * (mx == c.x ? 1.f : 0.f));       // heat * keep  (utils/decode.py:9-10)
            key[4 * j + 1] = ts_f2key(c.y * (my == c.y ? 1.f : 0.f));
            key[4 * j + 2] = ts_f2key(c.z * (mz == c.z ? 1.f : 0.f));
            key[4 * j + 3] = ts_f2key(c.w * (mw == c.w ? 1.f : 0.f));
            hp = hc; hc = hn;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 c = act(*reinterpret_cast<const float4*>(src + (row0 + j) * 128));
            key[4 * j + 0] = ts_f2key(c.x); key[4 * j + 1] = ts_f2key(c.y); key[4 * j + 2] = ts_f2key(c.z); key[4 * j + 3] = ts_f2key(c.w);
        }
    }
    __syncthreads();
    float* so = scores + (int64_t)blockIdx.x * K;
    int32_t* io = inds + (int64_t)blockIdx.x * K;
    ts_select<64>(key, [&](int s) { return (row0 + (s >> 2)) * 128 + x4 * 4 + (s & 3); }, K, 16384, sh,
                  [&](int rank, uint32_t k, uint32_t id) { so[rank] = ts_key2f(k); io[rank] = (int32_t)id; });
}

static bool topk_stream_enabled() {
    static const bool on = getenv("CN_DISABLE_TOPK_STREAM") == nullptr;
    return on;
}

// sig_lo >= 0: heat holds logits, sigmoid + clamp(sig_lo, 1 - sig_lo) on load (streaming kernel only: CN_EUNSUPPORTED otherwise)
static int launch_topk_channel(const float* heat, float* scores, int32_t* inds, int BC, int H, int W, int K, int apply_nms,
                               hipStream_t st, float sig_lo = -1.f) {
    const int HW = H * W;
    if (sig_lo >= 0.f) {
        if (!(H == 128 && W == 128 && K >= 1 && K <= TS_CAND && topk_stream_enabled() && (((uintptr_t)heat) & 15) == 0 && apply_nms)) {
            cn_set_error("top-K on logits: 128x128 maps with the pseudo-NMS only");
            return CN_EUNSUPPORTED;
        }
        hipLaunchKernelGGL((topk_map128_kernel<true, true>), dim3(BC), dim3(TS_THREADS), 0, st, heat, scores, inds, K, sig_lo);
        return CN_OK;
    }
    if (H == 128 && W == 128 && K >= 1 && K <= TS_CAND && topk_stream_enabled() && (((uintptr_t)heat) & 15) == 0) {
        if (apply_nms) hipLaunchKernelGGL(topk_map128_kernel<true>, dim3(BC), dim3(TS_THREADS), 0, st, heat, scores, inds, K);
        else hipLaunchKernelGGL(topk_map128_kernel<false>, dim3(BC), dim3(TS_THREADS), 0, st, heat, scores, inds, K);
        return CN_OK;
    }
    if (K > TK_MAXK || K > HW || K < 1) { cn_set_error("top-K: need 1 <= K <= min(%d, H*W) (K=%d)", TK_MAXK, K); return CN_EUNSUPPORTED; }
    if (HW > 32 * TK_THREADS) { cn_set_error("top-K: H*W=%d exceeds %d", HW, 32 * TK_THREADS); return CN_EUNSUPPORTED; }
    const size_t smem = (size_t)HW * 4;
#define TKC(E)                                                                                                         \
    do {                                                                                                               \
        if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)topk_channel_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL(topk_channel_kernel<E>, dim3(BC), dim3(TK_THREADS), smem, st, heat, scores, inds, H, W, K, apply_nms); \
    } while (0)
    if (HW <= 4 * TK_THREADS) TKC(4);
    else if (HW <= 8 * TK_THREADS) TKC(8);
    else if (HW <= 16 * TK_THREADS) TKC(16);
    else TKC(32);
#undef TKC
    return CN_OK;
}

extern "C" int cn_topk_channel(const float* heat, float* scores, int32_t* inds, int B, int C, int H, int W, int K, int apply_nms,
                               void* stream) {
    CN_CHECK_ARG(heat && scores && inds && B > 0 && C > 0 && H > 0 && W > 0, "cn_topk_channel: bad args");
    int rc = launch_topk_channel(heat, scores, inds, B * C, H, W, K, apply_nms, (hipStream_t)stream);
    if (rc) return rc;
    CN_LAUNCH_CHECK("cn_topk_channel");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ generic row top-K
__global__ __launch_bounds__(TK_THREADS) void topk_rows_kernel(const float* __restrict__ x, float* __restrict__ vals,
                                                               int32_t* __restrict__ idx, int L, int K) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    __shared__ TkShared sh;
    const int tid = threadIdx.x;
    const float* src = x + (int64_t)blockIdx.x * L;
    for (int i = tid; i < L; i += TK_THREADS) dyn[i] = f2key(src[i]);
    __syncthreads();
    block_topk(dyn, L, K, sh);
    if (tid < K) {
        vals[(int64_t)blockIdx.x * K + tid] = key2f(sh.sel_key[tid]);
        idx[(int64_t)blockIdx.x * K + tid] = (int32_t)sh.sel_idx[tid];
    }
}

extern "C" int cn_topk_rows(const float* x, float* vals, int32_t* idx, int R, int L, int K, void* stream) {
    CN_CHECK_ARG(x && vals && idx && R > 0 && L > 0, "cn_topk_rows: bad args");
    if (K > TK_MAXK || K > L || K < 1) CN_UNSUPPORTED("cn_topk_rows: need 1 <= K <= min(%d, L) (K=%d, L=%d)", TK_MAXK, K, L);
    if (L > 32768) CN_UNSUPPORTED("cn_topk_rows: L=%d exceeds 32768", L);
    const size_t smem = (size_t)L * 4;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)topk_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(topk_rows_kernel, dim3(R), dim3(TK_THREADS), smem, (hipStream_t)stream, x, vals, idx, L, K);
    CN_LAUNCH_CHECK("cn_topk_rows");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ __launch_bounds__(256) void nms3x3_kernel(const float* __restrict__ heat, float* __restrict__ out, int64_t total,
                                                     int H, int W) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const float v = heat[i];
        float m = v;
        for (int dy = -1; dy <= 1; ++dy) {
            if ((unsigned)(y + dy) >= (unsigned)H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                if ((unsigned)(x + dx) >= (unsigned)W) continue;
                m = fmaxf(m, heat[i + dy * W + dx]);
            }
        }
        out[i] = v * (m == v ? 1.f : 0.f);
    }
}

// any odd window (utils/decode.py:5-10 with kernel != 3: an option the reference defines and never uses): plain window scan
__global__ __launch_bounds__(256) void nms_k_kernel(const float* __restrict__ heat, float* __restrict__ out, int64_t total, int H,
                                                    int W, int r) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const float v = heat[i];
        float m = v;
        for (int dy = -r; dy <= r; ++dy) {
            if ((unsigned)(y + dy) >= (unsigned)H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                if ((unsigned)(x + dx) >= (unsigned)W) continue;
                m = fmaxf(m, heat[i + (int64_t)dy * W + dx]);
            }
        }
        out[i] = v * (m == v ? 1.f : 0.f);
    }
}

extern "C" int cn_nms(const float* heat, float* out, int B, int C, int H, int W, int kernel, void* stream) {
    CN_CHECK_ARG(heat && out && B > 0 && C > 0 && H > 0 && W > 0, "cn_nms: bad args");
    CN_CHECK_ARG(kernel >= 1 && (kernel & 1) == 1 && kernel <= 31, "cn_nms: kernel %d must be odd (1..31)", kernel);
    int64_t total = (int64_t)B * C * H * W;
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(nms_k_kernel, dim3((int)(g > 32768 ? 32768 : g)), dim3(256), 0, (hipStream_t)stream, heat, out, total, H, W,
                       (kernel - 1) / 2);
    CN_LAUNCH_CHECK("cn_nms");
    return CN_OK;
}

extern "C" int cn_nms3x3(const float* heat, float* out, int B, int C, int H, int W, void* stream) {
    CN_CHECK_ARG(heat && out && B > 0 && C > 0 && H > 0 && W > 0, "cn_nms3x3: bad args");
    int64_t total = (int64_t)B * C * H * W;
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(nms3x3_kernel, dim3((int)(g > 32768 ? 32768 : g)), dim3(256), 0, (hipStream_t)stream, heat, out, total, H, W);
    CN_LAUNCH_CHECK("cn_nms3x3");
    return CN_OK;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ feat, const int64_t* __restrict__ ind,
                                                          float* __restrict__ out, int B, int C, int64_t HW, int N) {
    const int64_t total = (int64_t)B * N * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bn = i / C;
        const int b = (int)(bn / N);
        int64_t id = ind[bn];
        id = id < 0 ? 0 : (id >= HW ? HW - 1 : id);
        out[i] = feat[((int64_t)b * C + c) * HW + id];
    }
}

extern "C" int cn_gather_rows(const float* feat, const int64_t* ind, float* out, int B, int C, int64_t HW, int N, void* stream) {
    CN_CHECK_ARG(feat && ind && out && B > 0 && C > 0 && HW > 0 && N > 0, "cn_gather_rows: bad args");
    int64_t total = (int64_t)B * N * C;
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, feat, ind, out, B, C, HW, N);
    CN_LAUNCH_CHECK("cn_gather_rows");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ ctdet_decode
// stage 2: one workgroup per image: top-K over the C*K per-class survivors, then gather wh/reg and build the boxes.
__global__ __launch_bounds__(TK_THREADS) void ctdet_stage2_kernel(const float* __restrict__ s1, const int32_t* __restrict__ i1,
                                                                  const float* __restrict__ wh, const float* __restrict__ reg,
                                                                  float* __restrict__ det, int64_t* __restrict__ inds_out,
                                                                  int32_t* __restrict__ cls_out, int C, int H, int W, int K) {
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    __shared__ TkShared sh;
    const int tid = threadIdx.x, b = blockIdx.x, L = C * K;
    const int64_t HW = (int64_t)H * W;
    for (int i = tid; i < L; i += TK_THREADS) dyn[i] = f2key(s1[(int64_t)b * L + i]);
    __syncthreads();
    block_topk(dyn, L, K, sh);
    if (tid < K) {
        const float score = key2f(sh.sel_key[tid]);
        const int pos = (int)sh.sel_idx[tid];
        const int cls = pos / K;
        const int ind = i1[(int64_t)b * L + pos];
        float xs = (float)(ind % W), ys = (float)(ind / W);
        if (reg) {
            xs = xs + reg[((int64_t)b * 2 + 0) * HW + ind];
            ys = ys + reg[((int64_t)b * 2 + 1) * HW + ind];
        } else {
            xs = xs + 0.5f;
            ys = ys + 0.5f;
        }
        const float w = wh[((int64_t)b * 2 + 0) * HW + ind], h = wh[((int64_t)b * 2 + 1) * HW + ind];
        float* d = det + ((int64_t)b * K + tid) * 6;
        d[0] = xs - w / 2; d[1] = ys - h / 2; d[2] = xs + w / 2; d[3] = ys + h / 2; d[4] = score; d[5] = (float)cls;
        if (inds_out) inds_out[(int64_t)b * K + tid] = ind;
        if (cls_out) cls_out[(int64_t)b * K + tid] = cls;
    }
}

// stage 2, streaming form (C*K <= 8192): the per-class survivors sit in registers (32 per thread), same select as the map kernel
__global__ __launch_bounds__(TS_THREADS) void ctdet_stage2_stream_kernel(const float* __restrict__ s1, const int32_t* __restrict__ i1,
                                                                         const float* __restrict__ wh, const float* __restrict__ reg,
                                                                         float* __restrict__ det, int64_t* __restrict__ inds_out,
                                                                         int32_t* __restrict__ cls_out, int C, int H, int W, int K) {
    __shared__ TsShared sh;
    __shared__ uint32_t okey[TS_CAND], opos[TS_CAND];
    const int tid = threadIdx.x, b = blockIdx.x, L = C * K;
    const int64_t HW = (int64_t)H * W;
    if (tid == 0) sh.ctl[3] = 0;
    uint32_t key[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int pos = s * TS_THREADS + tid;
        key[s] = pos < L ? ts_f2key(s1[(int64_t)b * L + (pos < L ? pos : 0)]) : 0u;
    }
    __syncthreads();
    ts_select<32>(key, [&](int s) { return s * TS_THREADS + tid; }, K, L, sh,
                  [&](int rank, uint32_t k, uint32_t id) { okey[rank] = k; opos[rank] = id; });
    __syncthreads();
    if (tid < K) {
        const float score = ts_key2f(okey[tid]);
        const int pos = (int)opos[tid];
        const int cls = pos / K;
        const int ind = i1[(int64_t)b * L + pos];
        float xs = (float)(ind % W), ys = (float)(ind / W);
        if (reg) {
            xs = xs + reg[((int64_t)b * 2 + 0) * HW + ind];
            ys = ys + reg[((int64_t)b * 2 + 1) * HW + ind];
        } else {
            xs = xs + 0.5f;
            ys = ys + 0.5f;
        }
        const float w = wh[((int64_t)b * 2 + 0) * HW + ind], h = wh[((int64_t)b * 2 + 1) * HW + ind];
        float* d = det + ((int64_t)b * K + tid) * 6;
        d[0] = xs - w / 2; d[1] = ys - h / 2; d[2] = xs + w / 2; d[3] = ys + h / 2; d[4] = score; d[5] = (float)cls;
        if (inds_out) inds_out[(int64_t)b * K + tid] = ind;
        if (cls_out) cls_out[(int64_t)b * K + tid] = cls;
    }
}

extern "C" size_t cn_ctdet_decode_workspace_bytes(int B, int C, int K) { return (size_t)B * C * K * 8; }

static int ctdet_decode_impl(const float* heat, const float* wh, const float* reg, float* det, int64_t* inds, int32_t* clses,
                            int B, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream, float sig_lo);
extern "C" int cn_ctdet_decode(const float* heat, const float* wh, const float* reg, float* det, int64_t* inds, int32_t* clses,
                               int B, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream) {
    return ctdet_decode_impl(heat, wh, reg, det, inds, clses, B, C, H, W, K, ws, ws_bytes, stream, -1.f);
}
// cn_ctdet_decode on the LOGITS of the class heat map: scores = clamp(sigmoid(logit), lo, 1 - lo) computed by the top-K kernel on load
// (same arithmetic as cn_sigmoid_clamp_fwd: the detections are bit-identical to cn_sigmoid_clamp_fwd + cn_ctdet_decode), the map itself
// is left untouched.  128x128 maps only (CN_EUNSUPPORTED otherwise: run the two calls).
extern "C" int cn_ctdet_decode_logits(const float* heat_logits, const float* wh, const float* reg, float* det, int64_t* inds, int32_t* clses,
                                      int B, int C, int H, int W, int K, float lo, void* ws, size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(lo >= 0.f && lo < 0.5f, "cn_ctdet_decode_logits: clamp %f", (double)lo);
    return ctdet_decode_impl(heat_logits, wh, reg, det, inds, clses, B, C, H, W, K, ws, ws_bytes, stream, lo);
}
static int ctdet_decode_impl(const float* heat, const float* wh, const float* reg, float* det, int64_t* inds, int32_t* clses,
                            int B, int C, int H, int W, int K, void* ws, size_t ws_bytes, void* stream, float sig_lo) {
    CN_CHECK_ARG(heat && wh && det && ws && B > 0 && C > 0 && H > 0 && W > 0, "cn_ctdet_decode: bad args");
    if (ws_bytes < cn_ctdet_decode_workspace_bytes(B, C, K)) { cn_set_error("cn_ctdet_decode: workspace too small"); return CN_EWORKSPACE; }
    if ((int64_t)C * K > 32768) CN_UNSUPPORTED("cn_ctdet_decode: C*K=%d exceeds 32768", C * K);
    hipStream_t st = (hipStream_t)stream;
    float* s1 = (float*)ws;
    int32_t* i1 = (int32_t*)(s1 + (size_t)B * C * K);
    int rc = launch_topk_channel(heat, s1, i1, B * C, H, W, K, 1, st, sig_lo);
    if (rc) return rc;
    CN_LAUNCH_CHECK("cn_ctdet_decode(stage1)");
    if (C * K <= 32 * TS_THREADS && K <= TS_CAND && topk_stream_enabled()) {
        hipLaunchKernelGGL(ctdet_stage2_stream_kernel, dim3(B), dim3(TS_THREADS), 0, st, (const float*)s1, (const int32_t*)i1, wh, reg,
                           det, inds, clses, C, H, W, K);
        CN_LAUNCH_CHECK("cn_ctdet_decode(stage2)");
        return CN_OK;
    }
    const size_t smem = (size_t)C * K * 4;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)ctdet_stage2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(ctdet_stage2_kernel, dim3(B), dim3(TK_THREADS), smem, st, (const float*)s1, (const int32_t*)i1, wh, reg, det,
                       inds, clses, C, H, W, K);
    CN_LAUNCH_CHECK("cn_ctdet_decode(stage2)");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ multi_pose_decode
// One workgroup per image.  s0/i0: top-K of the (single-class) centre heat map; sj/ij: per-joint top-K of hm_hp.
__global__ __launch_bounds__(256) void pose_assemble_kernel(const float* __restrict__ s0, const int32_t* __restrict__ i0,
                                                            const float* __restrict__ sj, const int32_t* __restrict__ ij,
                                                            const float* __restrict__ wh, const float* __restrict__ kps,
                                                            const float* __restrict__ reg, const float* __restrict__ hpo,
                                                            float* __restrict__ det, int J, int H, int W, int K) {
    extern __shared__ float sm[];
    float* cx = sm;               // [J][K] candidate x
    float* cy = cx + J * K;       // [J][K] candidate y
    float* cs = cy + J * K;       // [J][K] candidate score (thresholded)
    float* bb = cs + J * K;       // [K][4] boxes
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t HW = (int64_t)H * W;
    const int D = 5 + 2 * J + 1 + J;
    const float thresh = 0.1f;
    for (int i = tid; i < J * K; i += 256) {
        const int j = i / K;
        const float s = sj[(int64_t)b * J * K + i];
        const int ind = ij[(int64_t)b * J * K + i];
        float x = (float)(ind % W), y = (float)(ind / W);
        if (hpo) { x = x + hpo[((int64_t)b * 2 + 0) * HW + ind]; y = y + hpo[((int64_t)b * 2 + 1) * HW + ind]; }
        else { x = x + 0.5f; y = y + 0.5f; }
        const float m = s > thresh ? 1.f : 0.f;
        cs[i] = (1.f - m) * -1.f + m * s;
        cy[i] = (1.f - m) * (-10000.f) + m * y;
        cx[i] = (1.f - m) * (-10000.f) + m * x;
        (void)j;
    }
    for (int k = tid; k < K; k += 256) {
        const int ind = i0[(int64_t)b * K + k];
        float xs = (float)(ind % W), ys = (float)(ind / W);
        if (reg) { xs = xs + reg[((int64_t)b * 2 + 0) * HW + ind]; ys = ys + reg[((int64_t)b * 2 + 1) * HW + ind]; }
        else { xs = xs + 0.5f; ys = ys + 0.5f; }
        const float w = wh[((int64_t)b * 2 + 0) * HW + ind], h = wh[((int64_t)b * 2 + 1) * HW + ind];
        float* d = det + ((int64_t)b * K + k) * D;
        bb[k * 4 + 0] = d[0] = xs - w / 2;
        bb[k * 4 + 1] = d[1] = ys - h / 2;
        bb[k * 4 + 2] = d[2] = xs + w / 2;
        bb[k * 4 + 3] = d[3] = ys + h / 2;
        d[4] = s0[(int64_t)b * K + k];
        d[5 + 2 * J] = 0.f;  // class of a 1-class map
    }
    __syncthreads();
    for (int i = tid; i < J * K; i += 256) {
        const int j = i / K, k = i - j * K;
        const int ind = i0[(int64_t)b * K + k];
        const float px = kps[((int64_t)b * 2 * J + 2 * j) * HW + ind] + (float)(ind % W);      // regressed joint
        const float py = kps[((int64_t)b * 2 * J + 2 * j + 1) * HW + ind] + (float)(ind / W);
        float best = INFINITY;
        int bi = 0;
        for (int c = 0; c < K; ++c) {
            const float dx = px - cx[j * K + c], dy = py - cy[j * K + c];
            const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));   // no FMA contraction: match ATen bit for bit
            if (d < best) { best = d; bi = c; }
        }
        const float hx = cx[j * K + bi], hy = cy[j * K + bi];
        float hs = cs[j * K + bi];
        const float l = bb[k * 4 + 0], t = bb[k * 4 + 1], r = bb[k * 4 + 2], bt = bb[k * 4 + 3];
        const bool bad = (hx < l) || (hx > r) || (hy < t) || (hy > bt) || (hs < thresh) || (best > fmaxf(bt - t, r - l) * 0.3f);
        hs = bad ? hs * 0.f : hs;
        float* d = det + ((int64_t)b * K + k) * D;
        d[5 + 2 * j] = bad ? px : hx;
        d[5 + 2 * j + 1] = bad ? py : hy;
        // the reference RESHAPES hm_score [B,J,K,1] -> [B,K,J] (decode/multi_pose.py:90), it does not permute
        const int f = j * K + k;
        det[((int64_t)b * K + f / J) * D + 5 + 2 * J + 1 + f % J] = hs;
    }
}

extern "C" size_t cn_multi_pose_decode_workspace_bytes(int B, int J, int K) { return (size_t)B * (J + 1) * K * 8; }

extern "C" int cn_multi_pose_decode(const float* heat, const float* wh, const float* kps, const float* reg, const float* hm_hp,
                                    const float* hp_offset, float* det, int B, int J, int H, int W, int K, void* ws,
                                    size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(heat && wh && kps && hm_hp && det && ws && B > 0 && J > 0 && H > 0 && W > 0, "cn_multi_pose_decode: bad args (hm_hp is required)");
    if (ws_bytes < cn_multi_pose_decode_workspace_bytes(B, J, K)) { cn_set_error("cn_multi_pose_decode: workspace too small"); return CN_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    float* s0 = (float*)ws;
    float* sj = s0 + (size_t)B * K;
    int32_t* i0 = (int32_t*)(sj + (size_t)B * J * K);
    int32_t* ij = i0 + (size_t)B * K;
    int rc = launch_topk_channel(heat, s0, i0, B, H, W, K, 1, st);
    if (rc) return rc;
    rc = launch_topk_channel(hm_hp, sj, ij, B * J, H, W, K, 1, st);
    if (rc) return rc;
    CN_LAUNCH_CHECK("cn_multi_pose_decode(topk)");
    const size_t smem = ((size_t)3 * J * K + 4 * K) * sizeof(float);
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)pose_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(pose_assemble_kernel, dim3(B), dim3(256), smem, st, (const float*)s0, (const int32_t*)i0, (const float*)sj,
                       (const int32_t*)ij, wh, kps, reg, hp_offset, det, J, H, W, K);
    CN_LAUNCH_CHECK("cn_multi_pose_decode(assemble)");
    return CN_OK;
}
