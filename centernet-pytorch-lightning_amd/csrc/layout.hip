// Layout boundary + trivial element-wise kernels (HBM-bound; 16-byte vector accesses).
#include "common.h"

static thread_local char g_err[512] = "";

void cn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cn_version(void) { return 100; }
extern "C" const char* cn_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// NCHW fp32 -> NHWC T, tile = 64 pixels x 32 channels through LDS (both sides coalesced).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                           int C, int HW, int Cpad) {
    __shared__ float tile[32][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = (tid >> 6) + 4 * i, p = tid & 63;
        float v = 0.f;
        if (c0 + c < C && p0 + p < HW) v = src[((int64_t)n * C + c0 + c) * HW + p0 + p];
        tile[c][p] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int p = (tid >> 5) + 8 * i, c = tid & 31;
        if (c0 + c < Cpad && p0 + p < HW) Elem<T>::st(dst + ((int64_t)n * HW + p0 + p) * Cpad + c0 + c, tile[c][p]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int ld) {
    __shared__ float tile[32][65];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int p = (tid >> 5) + 8 * i, c = tid & 31;
        float v = 0.f;
        if (c0 + c < C && p0 + p < HW) v = Elem<T>::ld(src + ((int64_t)n * HW + p0 + p) * ld + c0 + c);
        tile[c][p] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = (tid >> 6) + 4 * i, p = tid & 63;
        if (c0 + c < C && p0 + p < HW) dst[((int64_t)n * C + c0 + c) * HW + p0 + p] = tile[c][p];
    }
}

extern "C" int cn_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int dtype,
                               void* stream) {
    CN_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "cn_nchw_to_nhwc: bad args");
    int HW = H * W;
    dim3 grid(cdiv(HW, 64), cdiv(Cpad, 32), N);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                                   src, (T*)dst, C, HW, Cpad));
    CN_LAUNCH_CHECK("cn_nchw_to_nhwc");
    return CN_OK;
}

extern "C" int cn_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int ld, int dtype,
                               void* stream) {
    CN_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "cn_nhwc_to_nchw: bad args");
    int HW = H * W;
    dim3 grid(cdiv(HW, 64), cdiv(C, 32), N);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)src, dst, C, HW, ld));
    CN_LAUNCH_CHECK("cn_nhwc_to_nchw");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void copy_channels_kernel(const T* __restrict__ src, int src_ld, int src_off,
                                                            T* __restrict__ dst, int dst_ld, int dst_off,
                                                            int64_t npix, int nvec /* vectors per pixel */) {
    constexpr int V = Vec16<T>::N;
    int64_t total = npix * nvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i / nvec;
        int v = (int)(i - p * nvec);
        uint4 val = *reinterpret_cast<const uint4*>(src + p * src_ld + src_off + v * V);
        *reinterpret_cast<uint4*>(dst + p * dst_ld + dst_off + v * V) = val;
    }
}

extern "C" int cn_copy_channels(const void* src, int src_ld, int src_off, void* dst, int dst_ld, int dst_off,
                                int64_t npix, int nch, int dtype, void* stream) {
    CN_CHECK_ARG(src && dst && npix > 0 && nch > 0, "cn_copy_channels: bad args");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(nch % V == 0 && src_ld % V == 0 && dst_ld % V == 0 && src_off % V == 0 && dst_off % V == 0,
                 "cn_copy_channels: channel counts/offsets must be multiples of %d (got nch=%d src %d+%d dst %d+%d)", V,
                 nch, src_ld, src_off, dst_ld, dst_off);
    int nvec = nch / V;
    int grid = (int)((npix * nvec + 255) / 256);
    if (grid > 8192) grid = 8192;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(copy_channels_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)src, src_ld, src_off, (T*)dst, dst_ld, dst_off, npix, nvec));
    CN_LAUNCH_CHECK("cn_copy_channels");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                  int64_t n) {
    constexpr int V = Vec16<T>::N;
    int64_t nv = n / V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float x[V], y[V];
        Vec16<T>::load(a + i * V, x);
        Vec16<T>::load(b + i * V, y);
#pragma unroll
        for (int j = 0; j < V; ++j) x[j] += y[j];
        Vec16<T>::store(out + i * V, x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - nv * V)) {
        int64_t i = nv * V + threadIdx.x;
        Elem<T>::st(out + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
    }
}

extern "C" int cn_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    CN_CHECK_ARG(a && b && out && n > 0, "cn_add: bad args");
    int V = dtype == CN_F32 ? 4 : 8;
    int grid = (int)((n / V + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(add_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)a, (const T*)b, (T*)out, n));
    CN_LAUNCH_CHECK("cn_add");
    return CN_OK;
}

// zero a buffer: 16-byte stores over the aligned body, byte stores for the ragged head / tail (any pointer, any size)
__global__ __launch_bounds__(256) void zero_kernel(unsigned char* __restrict__ p, int64_t nbytes, long long* stamp) {
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp = (long long)wall_clock64();       // measurement aid (cn_zero_stamps)
    const int64_t head = (int64_t)((16 - ((uintptr_t)p & 15)) & 15);
    const int64_t h = head < nbytes ? head : nbytes;
    const int64_t nvec = (nbytes - h) / 16;
    uint4* v = (uint4*)(p + h);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) v[i] = z;
    if (blockIdx.x == 0) {
        const int64_t tail0 = h + nvec * 16;
        for (int64_t i = threadIdx.x; i < h; i += blockDim.x) p[i] = 0;
        for (int64_t i = tail0 + threadIdx.x; i < nbytes; i += blockDim.x) p[i] = 0;
    }
}

// measurement aid: while a buffer is registered, the k-th cn_zero launch (host order, i.e. capture order) also stores the device wall
// clock at which it starts into buf[k] — a timeline of both streams of a replayed step without adding a single graph node
static long long* zero_stamp_buf = nullptr;
static int zero_stamp_n = 0, zero_stamp_i = 0;
extern "C" int cn_zero_stamps(int64_t* buf, int n) {
    zero_stamp_buf = (long long*)buf; zero_stamp_n = buf ? n : 0; zero_stamp_i = 0;
    return CN_OK;
}
extern "C" int cn_zero_stamps_used(void) { return zero_stamp_i; }

extern "C" int cn_zero(void* p, int64_t nbytes, void* stream) {
    CN_CHECK_ARG(p && nbytes > 0, "cn_zero: bad args");
    int64_t g = (nbytes / 16 + 255) / 256;
    int grid = (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
    long long* stamp = (zero_stamp_buf && zero_stamp_i < zero_stamp_n) ? zero_stamp_buf + zero_stamp_i++ : nullptr;
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, nbytes, stamp);
    CN_LAUNCH_CHECK("cn_zero");
    return CN_OK;
}

// measurement aid: the constant-rate (100 MHz) wall clock at the moment this one-thread kernel runs on its stream, e.g. where the
// launch stream and the weight-gradient stream end inside a REPLAYED hipGraph (the profiler's per-kernel overhead distorts that)
__global__ void stamp_kernel(long long* dst) { *dst = (long long)wall_clock64(); }
extern "C" int cn_stamp(int64_t* dst, void* stream) {
    CN_CHECK_ARG(dst, "cn_stamp: null pointer");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)dst);
    CN_LAUNCH_CHECK("cn_stamp");
    return CN_OK;
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<D>::st(d + i, Elem<S>::ld(s + i));
}

extern "C" int cn_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    CN_CHECK_ARG(src && dst && n > 0, "cn_cast: bad args");
    int grid = (int)((n + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipStream_t st = (hipStream_t)stream;
    if (sd == CN_F32 && dd == CN_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (sd == CN_BF16 && dd == CN_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (sd == CN_F32 && dd == CN_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else if (sd == CN_BF16 && dd == CN_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    else
        CN_CHECK_ARG(false, "cn_cast: bad dtypes %d -> %d", sd, dd);
    CN_LAUNCH_CHECK("cn_cast");
    return CN_OK;
}

// dst[i] = sum_s src[s][i]  (fp32 slabs -> activation dtype): folds the per-channel-block copies of the DCN offset gradient
template <typename D>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ src, D* __restrict__ dst, int S, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(src)[i];
        for (int sIdx = 1; sIdx < S; ++sIdx) {
            const float4 b = reinterpret_cast<const float4*>(src)[(int64_t)sIdx * n4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if constexpr (sizeof(D) == 4) reinterpret_cast<float4*>(dst)[i] = a;
        else *reinterpret_cast<uint2*>(dst + i * 4) = make_uint2(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w));
    }
}

extern "C" int cn_sum_slabs(const float* src, void* dst, int S, int64_t n, int dtype, void* stream) {
    CN_CHECK_ARG(src && dst && S >= 1 && n > 0 && n % 4 == 0, "cn_sum_slabs: bad args");
    int grid = (int)((n / 4 + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (dtype == CN_F32) hipLaunchKernelGGL(sum_slabs_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, S, n / 4);
    else if (dtype == CN_BF16) hipLaunchKernelGGL(sum_slabs_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, S, n / 4);
    else CN_CHECK_ARG(false, "cn_sum_slabs: bad dtype %d", dtype);
    CN_LAUNCH_CHECK("cn_sum_slabs");
    return CN_OK;
}

template <typename D>
__global__ __launch_bounds__(256) void add_f32_to_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         D* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        const float r[4] = {u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w};
        if constexpr (sizeof(D) == 4) {
            reinterpret_cast<float4*>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            uint2 o;
            o.x = pk_bf16(r[0], r[1]);
            o.y = pk_bf16(r[2], r[3]);
            reinterpret_cast<uint2*>(out)[i] = o;
        }
    }
}

extern "C" int cn_add_f32_to(const float* a, const float* b, void* out, int64_t n, int dtype, void* stream) {
    CN_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0, "cn_add_f32_to: bad args (n must be a multiple of 4)");
    int64_t g = (n / 4 + 255) / 256;
    int grid = (int)(g > 16384 ? 16384 : g);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(add_f32_to_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b,
                                                   (T*)out, n / 4));
    CN_LAUNCH_CHECK("cn_add_f32_to");
    return CN_OK;
}
