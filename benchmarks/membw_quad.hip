// membw_quad.hip -- read-only probe of the lane -> address patterns a register-fed MFMA scan
// could use over a row-major n x d fp32 matrix (rows of 16-byte chunks).  One wave instruction
// reads 4 rows x 256 B; what differs is which lane reads which 16 bytes:
//   P=0 "rowquad":  lane l -> row l%4, chunk l/4        (the 4x4x1 MFMA operand layout: the four
//                   lanes of a quad read four different rows)
//   P=1 "quadline": lane l -> row (l/4)%4, chunk 4*(l/16) + l%4   (a quad reads 64 contiguous bytes)
//   P=2 "row16":    lane l -> row l/16, chunk l%16      (16 lanes read 256 contiguous bytes)
//   P=3 "scan":     lane l -> 1 KiB contiguous of one row per instruction (csrc/flat.hip)
//   P=4 "row16x64": lane l -> row l%16, chunk l/16: 16 rows x 64 B per instruction (the operand
//                   layout of v_mfma_f32_16x16x4_f32 / 16x16x32_bf16); RG counts 16-row groups
//   P=6 "row16quad": lane l -> row l/4, chunk l%4: the same 16 rows x 64 B, a quad reading one row's 64 B
//   P=7 "row8x128": lane l -> row l%8, chunk l/8: 8 rows x 128 B (whole cache lines) per instruction
//   P=5 "row32x32": lane l -> row l%32, chunk l/32: 32 rows x 32 B per instruction (32x32 MFMA)
// Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 benchmarks/membw_quad.hip -o benchmarks/membw_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// RG row groups of 4 rows per wave tile; d4 = 16-byte chunks per row (multiple of 16).
template <int P, int RG>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ X, long long n, int d4, float* out) {
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * 4;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long ntiles = P == 7 ? n / (8 * RG) : (P == 4 || P == 6) ? n / (16 * RG) : (P == 5 ? n / (32 * RG) : n / (4 * RG));
    int r, c;
    if (P == 0) { r = lane & 3; c = lane >> 2; }
    else if (P == 1) { r = (lane >> 2) & 3; c = 4 * (lane >> 4) + (lane & 3); }
    else { r = lane >> 4; c = lane & 15; }
    float acc = 0.f;
    for (long long t = gw; t < ntiles; t += W) {
        const long long row0 = t * 4 * RG;
        if (P == 4 || P == 5 || P == 6 || P == 7) {
            // tile = RG groups of (P == 4 ? 16 : 32) rows; 24 (48) instructions cover 384 floats of a group
            constexpr int ROWS = P == 5 ? 32 : (P == 7 ? 8 : 16), CPI = 64 / ROWS, NI = 96 / CPI;
            const long long rbase = t * 4 * RG * (ROWS / 4);
            const int rr = P == 6 ? lane / 4 : lane % ROWS, ch = P == 6 ? lane % 4 : lane / ROWS;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
#pragma unroll
                for (int i0 = 0; i0 < NI; i0 += 12) {
                    f32x4 v[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        v[i] = __builtin_nontemporal_load(X + (rbase + g * ROWS + rr) * 96 + (i0 + i) * CPI + ch);
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
                }
            }
        } else if (P == 3) {
            for (int cc = 0; cc < d4; cc += 64) {
                if (cc + lane < d4) {
                    f32x4 v[4 * RG];
#pragma unroll
                    for (int g = 0; g < 4 * RG; ++g) v[g] = __builtin_nontemporal_load(X + (row0 + g) * d4 + cc + lane);
#pragma unroll
                    for (int g = 0; g < 4 * RG; ++g) acc += v[g].x + v[g].y + v[g].z + v[g].w;
                }
            }
        } else {
            // d4 = 96 assumed here: all RG x 6 loads of a tile are issued before the first use
            f32x4 v[RG][6];
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc)
                    v[g][cc] = __builtin_nontemporal_load(X + (row0 + 4 * g + r) * 96 + cc * 16 + c);
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) acc += v[g][cc].x + v[g][cc].y + v[g][cc].z + v[g][cc].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int P, int RG>
double run(const f32x4* p, long long n, int d4, float* out, int blocks, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    read_kernel<P, RG><<<blocks, 256>>>(p, n, d4, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) read_kernel<P, RG><<<blocks, 256>>>(p, n, d4, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return (double)n * d4 * 16 * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 10000000LL;
    const int d4 = argc > 2 ? atoi(argv[2]) : 96;
    f32x4* p;
    float* out;
    CK(hipMalloc(&p, n * d4 * 16));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 1, n * d4 * 16));
    const int reps = 10;
    printf("n=%lld d4=%d\n", n, d4);
    for (int blocks : {512, 768, 1024, 1536, 2048}) {
        printf("blocks=%5d row16x64 RG1 %.0f RG2 %.0f | row32x32 RG1 %.0f | row16quad RG1 %.0f RG2 %.0f GB/s\n", blocks,
               run<4, 1>(p, n, d4, out, blocks, reps), run<4, 2>(p, n, d4, out, blocks, reps),
               run<5, 1>(p, n, d4, out, blocks, reps), run<6, 1>(p, n, d4, out, blocks, reps),
               run<6, 2>(p, n, d4, out, blocks, reps));
        printf("blocks=%5d row8x128 RG1 %.0f RG2 %.0f RG4 %.0f GB/s\n", blocks, run<7, 1>(p, n, d4, out, blocks, reps),
               run<7, 2>(p, n, d4, out, blocks, reps), run<7, 4>(p, n, d4, out, blocks, reps));
        printf("blocks=%5d rowquad RG1 %.0f RG2 %.0f RG4 %.0f | quadline RG1 %.0f RG2 %.0f RG4 %.0f | row16 RG2 %.0f RG4 %.0f | scan R4 %.0f GB/s\n",
               blocks, run<0, 1>(p, n, d4, out, blocks, reps), run<0, 2>(p, n, d4, out, blocks, reps),
               run<0, 4>(p, n, d4, out, blocks, reps), run<1, 1>(p, n, d4, out, blocks, reps),
               run<1, 2>(p, n, d4, out, blocks, reps), run<1, 4>(p, n, d4, out, blocks, reps),
               run<2, 2>(p, n, d4, out, blocks, reps), run<2, 4>(p, n, d4, out, blocks, reps),
               run<3, 1>(p, n, d4, out, blocks, reps));
        fflush(stdout);
    }
    return 0;
}
