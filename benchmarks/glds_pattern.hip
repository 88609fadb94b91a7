// glds_pattern.hip -- does the SHAPE of a 1 KiB LDS-DMA instruction (global_load_lds_dwordx4) change
// what the CU's load path delivers from L2 / MALL?  l2_fill.hip measured 29-33 TB/s chip-wide with
// every instruction reading 1 KiB of consecutive bytes; the tile kernels read 8 rows x 128 B (or 16
// rows x 64 B) of a row-major matrix whose rows are 1536 B apart, in XOR-swizzled lane order.
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/glds_pattern benchmarks/glds_pattern.hip
// Patterns (lane -> byte offset inside a piece; pieces tile a matrix of 1536-byte rows):
//   0: 16*lane                                   1 KiB contiguous
//   1: (lane>>3)*1536 + 16*(lane&7)              8 rows x one 128-byte line
//   2: as 1, chunk ^ swizzle(row)                what flat_mfma_pipe.h issues
//   3: (lane>>2)*1536 + 16*(lane&3)              16 rows x 64 B
//   4: (lane>>4)*1536 + 16*(lane&15)             4 rows x 256 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ROWB = 1536;

template <int PAT>
__device__ __forceinline__ long long piece_base(long long p) {
    // pieces of pattern PAT laid over the matrix so that together they cover it exactly once
    if (PAT == 0) return p * 1024;
    if (PAT == 1 || PAT == 2) return (p / 12) * 8 * ROWB + (p % 12) * 128;
    if (PAT == 3) return (p / 24) * 16 * ROWB + (p % 24) * 64;
    return (p / 6) * 4 * ROWB + (p % 6) * 256;
}
template <int PAT>
__device__ __forceinline__ int lane_off(int lane, int wid) {
    if (PAT == 0) return 16 * lane;
    if (PAT == 1) return (lane >> 3) * ROWB + 16 * (lane & 7);
    if (PAT == 2) {
        const int r = wid * 8 + (lane >> 3);
        return (lane >> 3) * ROWB + 16 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    if (PAT == 3) return (lane >> 2) * ROWB + 16 * (lane & 3);
    return (lane >> 4) * ROWB + 16 * (lane & 15);
}

// every wave reads `iters` pieces, DEPTH in flight; piece p of wave w is ((w * 7919 + p) % npieces)
template <int PAT, int DEPTH>
__global__ __launch_bounds__(512) void fill_glds(const char* __restrict__ X, long long npieces, int iters, float* out) {
    extern __shared__ __align__(16) unsigned char sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* ring = reinterpret_cast<float*>(sm) + wid * DEPTH * 256;
    long long p = ((long long)(blockIdx.x * 8 + wid) * 7919) % npieces;
    const int lo = lane_off<PAT>(lane, wid);
    int slot = 0;
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + piece_base<PAT>(p) + lo),
                                         (__attribute__((address_space(3))) void*)(ring + slot * 256), 16, 0, 0);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        p = p + 1 == npieces ? 0 : p + 1;
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((DEPTH - 1) & 0xF) | (((DEPTH - 1) >> 4) << 14));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (ring[lane] == 123.456f) out[0] = 1.f;
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

template <int PAT>
void run(const char* X, float* out, const char* name) {
    const int iters = 4096, blocks = 256;
    const double bytes = (double)blocks * 8 * iters * 1024;
    for (size_t foot : {(size_t)3 << 19, (size_t)24 << 20, (size_t)192 << 20, (size_t)3 << 30}) {
        const long long np = foot / 1024;
        double g4 = timeit([&] { fill_glds<PAT, 4><<<blocks, 512, 8 * 4 * 1024>>>(X, np, iters, out); }, bytes);
        double g12 = timeit([&] { fill_glds<PAT, 12><<<blocks, 512, 8 * 12 * 1024>>>(X, np, iters, out); }, bytes);
        printf("%-28s footprint %7.1f MiB: 8 waves/CU, depth 4: %6.2f TB/s (%5.1f B/clk/CU @2.1)  depth 12: %6.2f TB/s (%5.1f)\n", name,
               foot / 1048576.0, g4, g4 * 1e12 / 256 / 2.1e9, g12, g12 * 1e12 / 256 / 2.1e9);
        fflush(stdout);
    }
}

int main() {
    const size_t total = (size_t)3 << 30;
    char* X; float* out;
    CK(hipMalloc(&X, total + (1 << 20))); CK(hipMalloc(&out, 4)); CK(hipMemset(X, 1, total + (1 << 20)));
    run<0>(X, out, "1 KiB contiguous");
    run<1>(X, out, "8 rows x 128 B");
    run<2>(X, out, "8 rows x 128 B, swizzled");
    run<3>(X, out, "16 rows x 64 B");
    run<4>(X, out, "4 rows x 256 B");
    return 0;
}
