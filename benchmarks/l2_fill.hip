// l2_fill.hip -- how fast can every CU pull 16-byte chunks that HIT in L2 / MALL / HBM,
// (a) into registers with plain global_load_dwordx4, (b) into LDS with global_load_lds_dwordx4?
// Decides whether the MFMA batch kernel's operand fill should go through LDS-DMA or registers.
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/l2_fill benchmarks/l2_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave reads `iters` KiB-sized pieces; piece p of wave w is at ((w * 7919 + p) % npieces) KiB
template <int DEPTH>
__global__ __launch_bounds__(256) void fill_glds(const float* __restrict__ X, long long npieces, int iters, float* out) {
    extern __shared__ __align__(16) unsigned char sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* ring = reinterpret_cast<float*>(sm) + wid * DEPTH * 256;
    long long p = ((long long)(blockIdx.x * 4 + wid) * 7919) % npieces;
    int slot = 0;
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + p * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(ring + slot * 256), 16, 0, 0);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        p = p + 1 == npieces ? 0 : p + 1;
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((DEPTH - 1) & 0xF) | (((DEPTH - 1) >> 4) << 14));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (ring[lane] == 123.456f) out[0] = 1.f;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void fill_regs(const float* __restrict__ X, long long npieces, int iters, float* out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long long p = ((long long)(blockIdx.x * 4 + wid) * 7919) % npieces;
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i += DEPTH) {
        f32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            v[u] = *reinterpret_cast<const f32x4*>(X + p * 256 + lane * 4);
            p = p + 1 == npieces ? 0 : p + 1;
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

template <class F>
double timeit(F launch, double bytes) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return bytes * 3 / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t total = (size_t)4 << 30;
    float *X, *out;
    CK(hipMalloc(&X, total)); CK(hipMalloc(&out, 4)); CK(hipMemset(X, 1, total));
    const int iters = 4096;
    for (int bpc : {1, 2, 4}) {
        const int blocks = 256 * bpc;
        const double bytes = (double)blocks * 4 * iters * 1024;
        for (size_t foot : {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, total}) {
            const long long np = foot / 1024;
            double g8 = timeit([&] { fill_glds<8><<<blocks, 256, 4 * 8 * 1024>>>(X, np, iters, out); }, bytes);
            double g16 = timeit([&] { fill_glds<16><<<blocks, 256, 4 * 16 * 1024>>>(X, np, iters, out); }, bytes);
            double r8 = timeit([&] { fill_regs<8><<<blocks, 256>>>(X, np, iters, out); }, bytes);
            double r16 = timeit([&] { fill_regs<16><<<blocks, 256>>>(X, np, iters, out); }, bytes);
            printf("blocks/CU %d footprint %6zu MiB: LDS-DMA depth8 %.2f depth16 %.2f | registers depth8 %.2f depth16 %.2f TB/s\n",
                   bpc, foot >> 20, g8, g16, r8, r16);
            fflush(stdout);
        }
    }
    return 0;
}
