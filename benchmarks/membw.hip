// membw.hip -- read-only HBM streaming probe: what a kernel that only loads (16 B per lane,
// fully coalesced) and adds can reach on this box.  Gives the practical ceiling the dense scan
// kernel (csrc/flat.hip) is compared with in DESIGN.md.  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 benchmarks/membw.hip -o benchmarks/membw && ./benchmarks/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// Each wave reads chunks of U KiB (U wave-instructions of 1 KiB), interleaved over all waves.
template <int U, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS) void read_kernel(const f32x4* __restrict__ p, long long n4, float* out) {
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * (THREADS / 64);
    const long long gw = (long long)blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    const long long nchunks = n4 / (64 * U);
    float acc = 0.f;
    for (long long c = gw; c < nchunks; c += W) {
        const f32x4* q = p + c * (64 * U) + lane;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(q + 64 * u) : q[64 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;  // keep the loads alive
}

template <int U, bool NT, int THREADS>
double run(const f32x4* p, long long n4, float* out, int blocks, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    read_kernel<U, NT, THREADS><<<blocks, THREADS>>>(p, n4, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) read_kernel<U, NT, THREADS><<<blocks, THREADS>>>(p, n4, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)n4 * 16 * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const long long bytes = argc > 1 ? atoll(argv[1]) : 15360000000LL;
    const long long n4 = bytes / 16;
    f32x4* p;
    float* out;
    CK(hipMalloc(&p, n4 * 16));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 1, n4 * 16));
    const int reps = 10;
    printf("bytes=%lld\n", bytes);
    for (int blocks : {512, 1024, 2048, 4096, 8192}) {
        printf("blocks=%5d  U1 nt %.0f | U2 nt %.0f | U4 nt %.0f | U6 nt %.0f | U8 nt %.0f | U4 plain %.0f | U8 plain %.0f | U4 nt 512thr %.0f | U4 nt 1024thr %.0f GB/s\n",
               blocks, run<1, true, 256>(p, n4, out, blocks, reps), run<2, true, 256>(p, n4, out, blocks, reps),
               run<4, true, 256>(p, n4, out, blocks, reps), run<6, true, 256>(p, n4, out, blocks, reps),
               run<8, true, 256>(p, n4, out, blocks, reps), run<4, false, 256>(p, n4, out, blocks, reps),
               run<8, false, 256>(p, n4, out, blocks, reps), run<4, true, 512>(p, n4, out, blocks / 2, reps),
               run<4, true, 1024>(p, n4, out, blocks / 4, reps));
        fflush(stdout);
    }
    return 0;
}
