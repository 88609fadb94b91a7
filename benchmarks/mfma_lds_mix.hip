// mfma_lds_mix.hip -- does LDS fragment traffic slow the fp32 matrix pipe down when nothing depends on it?
// 8 waves per CU issue batches of 32 v_mfma_f32_32x32x2_f32 (8 accumulators x 4) with R ds_read_b128 per batch whose
// results feed the NEXT batch (the tile kernels' 9 reads per 32 MFMAs = R 9).  Same for bf16 32x32x16 (8 per batch).
//   hipcc -O3 --offload-arch=gfx950 benchmarks/mfma_lds_mix.hip -o benchmarks/mfma_lds_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int R, bool BF>
__global__ __launch_bounds__(512) void mix_kernel(float* out, int iters) {
    __shared__ f32x4 lds[4096];  // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{1.f * i, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 cur[R > 0 ? R : 1], nxt[R > 0 ? R : 1];
    for (int r = 0; r < (R > 0 ? R : 1); ++r) cur[r] = lds[(threadIdx.x + 64 * r) & 4095];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) nxt[r] = lds[(threadIdx.x + 64 * r + 17 * it) & 4095];
        if (!BF) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i % (R > 0 ? R : 1)][cc], cur[0][cc], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[i % (R > 0 ? R : 1)]),
                                                                 __builtin_bit_cast(bf16x8, cur[0]), acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < (R > 0 ? R : 1); ++r) cur[r] = R > 0 ? nxt[r] : cur[r];
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <int R, bool BF>
void run() {
    float* out;
    (void)hipMalloc(&out, 4);
    const int iters = BF ? 40000 : 5000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    mix_kernel<R, BF><<<blocks, 512>>>(out, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mix_kernel<R, BF><<<blocks, 512>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 8 * iters * (BF ? 8 : 32);
    printf("%s, %d ds_read_b128 per batch of %d: %.3f ms, %.1f TFLOP/s\n", BF ? "bf16 32x32x16" : "fp32 32x32x2", R, BF ? 8 : 32, ms,
           instr * (BF ? 32768.0 : 4096.0) / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    run<0, false>(); run<6, false>(); run<9, false>(); run<18, false>();
    run<0, true>(); run<6, true>(); run<9, true>(); run<12, true>();
    return 0;
}
