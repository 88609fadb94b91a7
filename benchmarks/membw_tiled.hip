// membw_tiled.hip -- how much of the streaming read bandwidth survives when a row tile is read in
// K-chunks: a block owns 256 consecutive rows of a (n x 384 fp32) matrix and reads them chunk by
// chunk; within one wave instruction each row contributes RL contiguous bytes (RL = 64 ... 1024).
// Models the access pattern of the MFMA tile kernel (flat_mfma.hip).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int RL>  // bytes per row per wave instruction
__global__ __launch_bounds__(256) void tiled_read(const float* __restrict__ X, long long ntiles, int ld, float* out) {
    constexpr int LPR = RL / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;         // rows per wave instruction
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nsteps = ld * 4 / RL;
    float acc = 0.f;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const float* tile = X + t * 256 * (long long)ld;
        for (int ks = 0; ks < nsteps; ++ks) {
            // wave `wid` covers rows [64*wid, +64) of the tile, RPI rows per instruction
#pragma unroll 4
            for (int it = 0; it < 64 / RPI; ++it) {
                const int row = 64 * wid + it * RPI + lane / LPR;
                const f32x4 v = __builtin_nontemporal_load(
                    reinterpret_cast<const f32x4*>(tile + (long long)row * ld + ks * (RL / 4) + (lane % LPR) * 4));
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// Same walk, but through LDS-DMA (global_load_lds_dwordx4) into a ring of 1 KiB slots per wave,
// with the 16-byte chunks of a row permuted inside its RL-byte run (as the XOR swizzle does).
template <int RL, int DEPTH, bool PERMUTE>
__global__ __launch_bounds__(256) void tiled_glds(const float* __restrict__ X, long long ntiles, int ld, float* out) {
    extern __shared__ __align__(16) unsigned char sm[];
    constexpr int LPR = RL / 16, RPI = 64 / LPR;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* ring = reinterpret_cast<float*>(sm) + wid * DEPTH * 256;
    const int nsteps = ld * 4 / RL;
    int slot = 0;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const float* tile = X + t * 256 * (long long)ld;
        for (int ks = 0; ks < nsteps; ++ks) {
            for (int it = 0; it < 64 / RPI; ++it) {
                const int row = 64 * wid + it * RPI + lane / LPR;
                int ch = lane % LPR;
                if (PERMUTE) ch ^= (row >> 1) & (LPR - 1);
                glds16(tile + (long long)row * ld + ks * (RL / 4) + ch * 4, ring + slot * 256);
                slot = slot + 1 == DEPTH ? 0 : slot + 1;
                __builtin_amdgcn_s_waitcnt(0x0F70 | ((DEPTH - 1) & 0xF) | (((DEPTH - 1) >> 4) << 14));
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (ring[lane] == 123.456f) out[0] = 1.f;
}
template <int RL, int DEPTH, bool PERMUTE>
double run_glds(const float* X, long long ntiles, int ld, float* out, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t smem = 4 * DEPTH * 1024;
    tiled_glds<RL, DEPTH, PERMUTE><<<blocks, 256, smem>>>(X, ntiles, ld, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) tiled_glds<RL, DEPTH, PERMUTE><<<blocks, 256, smem>>>(X, ntiles, ld, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return (double)ntiles * 256 * ld * 4 * 5 / (ms * 1e-3) / 1e9;
}

template <int RL>
double run(const float* X, long long ntiles, int ld, float* out, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    tiled_read<RL><<<blocks, 256>>>(X, ntiles, ld, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) tiled_read<RL><<<blocks, 256>>>(X, ntiles, ld, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return (double)ntiles * 256 * ld * 4 * 5 / (ms * 1e-3) / 1e9;
}

int main() {
    const int ld = 384; const long long n = 10000000 / 256 * 256, ntiles = n / 256;
    float *X, *out;
    CK(hipMalloc(&X, n * ld * 4)); CK(hipMalloc(&out, 4)); CK(hipMemset(X, 1, n * ld * 4));
    for (int blocks : {512, 1024, 2048}) {
        printf("blocks=%4d  RL64 %.0f | RL128 %.0f | RL256 %.0f | RL512 %.0f | RL1024 %.0f | RL1536 n/a  GB/s\n", blocks,
               run<64>(X, ntiles, ld, out, blocks), run<128>(X, ntiles, ld, out, blocks), run<256>(X, ntiles, ld, out, blocks),
               run<512>(X, ntiles, ld, out, blocks), run<1024>(X, ntiles, ld, out, blocks));
        fflush(stdout);
        printf("   glds (LDS-DMA) depth 8: RL64 %.0f | RL128 %.0f | RL128 permuted %.0f | RL256 %.0f | RL256 permuted %.0f ; depth 4: RL128 %.0f  GB/s\n",
               run_glds<64, 8, false>(X, ntiles, ld, out, blocks), run_glds<128, 8, false>(X, ntiles, ld, out, blocks),
               run_glds<128, 8, true>(X, ntiles, ld, out, blocks), run_glds<256, 8, false>(X, ntiles, ld, out, blocks),
               run_glds<256, 8, true>(X, ntiles, ld, out, blocks), run_glds<128, 4, false>(X, ntiles, ld, out, blocks));
        fflush(stdout);
    }
    return 0;
}
