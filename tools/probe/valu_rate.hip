// Issue-rate probe: cycles per wave instruction of a few VALU opcodes on gfx950, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o tools/probe/valu_rate && tools/probe/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_IT 256
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(uint32_t* out, unsigned long long* cyc) {
    uint32_t a[8], b = threadIdx.x * 3 + 1, c = threadIdx.x * 7 + 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    __syncthreads();
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < N_IT; ++it) {
        // 16 x 8 independent chains
        if (OP == 0) { REP16(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_dot2_f32_bf16 %0, %0, %8, %9\n v_dot2_f32_bf16 %1, %1, %8, %9\n v_dot2_f32_bf16 %2, %2, %8, %9\n v_dot2_f32_bf16 %3, %3, %8, %9\n v_dot2_f32_bf16 %4, %4, %8, %9\n v_dot2_f32_bf16 %5, %5, %8, %9\n v_dot2_f32_bf16 %6, %6, %8, %9\n v_dot2_f32_bf16 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 2) { REP16(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 5) { REP16(asm volatile("v_dot2c_f32_bf16 %0, %8, %9\n v_dot2c_f32_bf16 %1, %8, %9\n v_dot2c_f32_bf16 %2, %8, %9\n v_dot2c_f32_bf16 %3, %8, %9\n v_dot2c_f32_bf16 %4, %8, %9\n v_dot2c_f32_bf16 %5, %8, %9\n v_dot2c_f32_bf16 %6, %8, %9\n v_dot2c_f32_bf16 %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 6) { REP16(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 7) { REP16(asm volatile("v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 8) { REP16(asm volatile("v_dot2_f32_f16 %0, %0, %8, %9\n v_dot2_f32_f16 %1, %1, %8, %9\n v_dot2_f32_f16 %2, %2, %8, %9\n v_dot2_f32_f16 %3, %3, %8, %9\n v_dot2_f32_f16 %4, %4, %8, %9\n v_dot2_f32_f16 %5, %5, %8, %9\n v_dot2_f32_f16 %6, %6, %8, %9\n v_dot2_f32_f16 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
        if (OP == 9) { REP16(asm volatile("v_fma_f16 %0, %0, %8, %9\n v_fma_f16 %1, %1, %8, %9\n v_fma_f16 %2, %2, %8, %9\n v_fma_f16 %3, %3, %8, %9\n v_fma_f16 %4, %4, %8, %9\n v_fma_f16 %5, %5, %8, %9\n v_fma_f16 %6, %6, %8, %9\n v_fma_f16 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
    }
    const unsigned long long t1 = clock64();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name) {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8 * 8);
    for (int waves : {1, 2, 4}) {          // waves per SIMD on one CU (block = 256 * waves threads)
        const int threads = 256 * waves > 1024 ? 1024 : 256 * waves;
        k<OP><<<1, threads>>>(out, cyc);
        k<OP><<<1, threads>>>(out, cyc);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-22s %d wave(s)/SIMD: %6.2f cycles per instruction per wave, %6.2f per SIMD-instruction\n", name, threads / 256, (double)c / (N_IT * 128.0), (double)c / (N_IT * 128.0) / (threads / 256));
    }
}
int main() {
    run<3>("v_fma_f32"); run<6>("v_xor_b32"); run<2>("v_perm_b32"); run<4>("v_cvt_pk_bf16_f32"); run<1>("v_dot2_f32_bf16"); run<5>("v_dot2c_f32_bf16");
    run<8>("v_dot2_f32_f16"); run<0>("v_pk_fma_f16"); run<7>("v_pk_mul_f16"); run<9>("v_fma_f16");
    return 0;
}
