// mfma_timing.hip -- per-segment cycle breakdown of the MFMA batch kernel (debug build of
// flat_mfma_kernel.h with RAG_MFMA_TIMING; not part of librag_mi355x.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRAG_MFMA_TIMING -o benchmarks/mfma_timing \
//       benchmarks/mfma_timing.hip rag_system_amd/csrc/lib.hip rag_system_amd/csrc/flat_mfma.hip
//   ./benchmarks/mfma_timing [bf16=1] [nq=1024] [d=768] [k=10] [shape=0] [stages=0] [n=10000000]
#include "../rag_system_amd/csrc/flat_mfma_kernel.h"

#include <cstdlib>
#include <vector>

namespace rag {
int flat_init_scratch() { return 0; }  // lib.hip's rag_init asks flat.hip for its scratch address; this harness has no flat.hip
namespace mfma {
int dispatch_bf16(const MfmaPlan& p, const Args& a) { return dispatch<true>(p, a); }
int dispatch_f32(const MfmaPlan& p, const Args& a) { return dispatch<false>(p, a); }
}  // namespace mfma
}  // namespace rag

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void fill_kernel(unsigned int* p, size_t nwords, unsigned int seed, bool bf) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        unsigned int h = (unsigned int)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // small random values around +-1/16 (bf16: two per word; fp32: one)
        if (bf) p[i] = 0x3D803D80u ^ ((h & 0x8000007Fu)) ^ ((h >> 9) & 0x007F0000u) ^ ((h << 3) & 0x80000000u >> 16 << 16 >> 16);
        else p[i] = 0x3D800000u ^ (h & 0x807FFFFFu);
    }
}

int main(int argc, char** argv) {
    const bool bf = argc > 1 ? atoi(argv[1]) != 0 : true;
    const int nq = argc > 2 ? atoi(argv[2]) : 1024;
    const int d = argc > 3 ? atoi(argv[3]) : 768;
    const int k = argc > 4 ? atoi(argv[4]) : 10;
    const int shape = argc > 5 ? atoi(argv[5]) : 0;
    const int stages = argc > 6 ? atoi(argv[6]) : 0;
    const long long n = argc > 7 ? atoll(argv[7]) : 10000000LL;
    const bool zeros = argc > 8 ? atoi(argv[8]) != 0 : false;  // all-zero operands: no data toggling (DVFS check)
    if (rag_init(0) != 0) { fprintf(stderr, "%s\n", rag_last_error()); return 1; }
    rag_set_option("mfma_shape", shape);
    rag_set_option("mfma_stages", stages);
    const int ldw = bf ? d / 2 : d;  // words per row (d must be a multiple of 8)
    unsigned int *X, *Q;
    float *ck, *zero; int* ci;
    CK(hipMalloc(&X, (size_t)n * ldw * 4));
    CK(hipMalloc(&Q, (size_t)(nq + 256) * ldw * 4));
    CK(hipMalloc(&zero, 64)); CK(hipMemset(zero, 0, 64));
    if (zeros) {
        CK(hipMemset(X, 0, (size_t)n * ldw * 4));
        CK(hipMemset(Q, 0, (size_t)(nq + 256) * ldw * 4));
    } else {
        fill_kernel<<<4096, 256>>>(X, (size_t)n * ldw, 1u, bf);
        fill_kernel<<<256, 256>>>(Q, (size_t)(nq + 256) * ldw, 7u, bf);
    }
    rag::MfmaPlan p;
    if (!rag::mfma_plan(n, nq, k, bf, &p)) { fprintf(stderr, "no plan\n"); return 1; }
    CK(hipMalloc(&ck, (size_t)nq * p.nlists * k * 4)); CK(hipMalloc(&ci, (size_t)nq * p.nlists * k * 4));
    CK(hipDeviceSynchronize());
    hipStream_t st = rag::lib_stream();
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long z[8] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(rag::mfma::g_mfma_timing), z, sizeof z));
        CK(hipEventRecord(a, st));
        if (rag::mfma_launch(p, (const float*)X, n, ldw, (const float*)Q, nq, k, 0, ck, ci, nullptr, nullptr, zero, st)) {
            fprintf(stderr, "%s\n", rag_last_error()); return 1;
        }
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long t[8];
        CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(rag::mfma::g_mfma_timing), sizeof t));
        const double steps = (double)t[5];
        printf("shape %d (WM %d WN %d MT %d) NS %d S %d col_tiles %d: %.3f ms; per K step of wave 0, cycles (100 MHz ticks if constant clock): "
               "vmcnt-wait %.1f | barrier %.1f | issue %.1f | lds+mfma %.1f | epilogue %.1f | steps/block %.0f\n",
               p.shape + 1, p.WM, p.WN, p.MT, p.NS, p.S, p.col_tiles, ms, t[0] / steps, t[1] / steps, t[2] / steps,
               t[3] / steps, t[4] / steps, steps / (double)t[6]);
    }
    return 0;
}
