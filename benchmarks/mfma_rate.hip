// mfma_rate.hip -- issue rate of the fp32 / bf16 matrix instructions with nothing else in the way: W waves per SIMD,
// each a chain of 8 independent accumulators, operands in registers.  The ceiling the tile kernels are priced against.
//   hipcc -O3 --offload-arch=gfx950 benchmarks/mfma_rate.hip -o benchmarks/mfma_rate && benchmarks/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[8];
    f32x4 acc4[8];
    for (int i = 0; i < 8; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    bf16x8 ab, bb;
    for (int e = 0; e < 8; ++e) {
        ab[e] = (__bf16)(a + e);
        bb[e] = (__bf16)(b - e);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    }
    if (s == 12345.678f) out[0] = s;  // never true: keeps the chain alive
}

template <int MODE>
void run(const char* name, double flops_per_instr, int threads) {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 20000, blocks = 256 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, threads>>>(out, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, threads>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * (threads / 64) * iters * 8;
    printf("%s, %d waves per block, 2 blocks per CU: %.3f ms, %.1f TFLOP/s\n", name, threads / 64, ms,
           instr * flops_per_instr / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("v_mfma_f32_32x32x2_f32", 4096.0, threads);
        run<1>("v_mfma_f32_16x16x4_f32", 2048.0, threads);
        run<2>("v_mfma_f32_32x32x16_bf16", 32768.0, threads);
    }
    return 0;
}
