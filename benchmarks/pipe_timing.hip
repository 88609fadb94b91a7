// pipe_timing.hip -- the software-pipelined bf16 tile kernel (flat_mfma_pipe.h) alone, with the
// ablation switches RAG_PIPE_EXP (1: no tile loads after the first row tile, 2: no fragment
// reads, 4: no MFMA -- faulted the GPU with the 16x16x32 tiles of r03, do not run --, 8: no barrier in
// the K loop; results are then meaningless, only the time is read).  Not part of librag_mi355x.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRAG_PIPE_EXP=n] -Iinclude -o benchmarks/pipe_timing[_n] \
//       benchmarks/pipe_timing.hip rag_system_amd/csrc/lib.hip
//   ./benchmarks/pipe_timing [nq=1024] [d=768] [n=10000000] [stages=4] [zeros=0]
#include "../rag_system_amd/csrc/flat_mfma_kernel.h"

#include <cstdlib>

namespace rag {
int flat_init_scratch() { return 0; }  // lib.hip's rag_init calls it; flat.hip is not linked here
}  // namespace rag

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void fill_kernel(unsigned int* p, size_t nwords, unsigned int seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        unsigned int h = (unsigned int)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = 0x3D803D80u ^ (h & 0x807F807Fu);   // two bf16 around +-1/16
    }
}
__global__ void fill_f(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int main(int argc, char** argv) {
    const int nq = argc > 1 ? atoi(argv[1]) : 1024;
    const int d = argc > 2 ? atoi(argv[2]) : 768;
    const long long n = argc > 3 ? atoll(argv[3]) : 10000000LL;
    const int stages = argc > 4 ? atoi(argv[4]) : 4;
    const bool zeros = argc > 5 ? atoi(argv[5]) != 0 : false;
    if (rag_init(0) != 0) { fprintf(stderr, "%s\n", rag_last_error()); return 1; }
    const int ldw = d / 2;
    unsigned int *X, *Q, *counts;
    float *ck, *zero, *thr; int* ci;
    const int cap = 1024;
    CK(hipMalloc(&X, (size_t)n * ldw * 4));
    CK(hipMalloc(&Q, (size_t)(nq + 256) * ldw * 4));
    CK(hipMalloc(&zero, 64)); CK(hipMemset(zero, 0, 64));
    CK(hipMalloc(&thr, (size_t)nq * 4)); CK(hipMalloc(&counts, (size_t)nq * 4)); CK(hipMemset(counts, 0, (size_t)nq * 4));
    CK(hipMalloc(&ck, (size_t)nq * cap * 4)); CK(hipMalloc(&ci, (size_t)nq * cap * 4));
    if (zeros) {
        CK(hipMemset(X, 0, (size_t)n * ldw * 4));
        CK(hipMemset(Q, 0, (size_t)(nq + 256) * ldw * 4));
    } else {
        fill_kernel<<<4096, 256>>>(X, (size_t)n * ldw, 1u);
        fill_kernel<<<256, 256>>>(Q, (size_t)(nq + 256) * ldw, 7u);
    }
    fill_f<<<(nq + 255) / 256, 256>>>(thr, nq, 1e30f);   // nothing passes: the K loop alone
    rag::MfmaPlan p;
    p.shape = 9; p.WM = 2; p.WN = 4; p.MT = 4; p.NS = stages == 3 ? 3 : 4; p.KS = 1; p.bf16 = true; p.filt = true;
    p.tile_step = 1; p.cap = cap; p.col_tiles = (nq + 255) / 256; p.nlists = 1;
    p.S = ((rag::num_cus() + p.col_tiles - 1) / p.col_tiles + 7) & ~7;
    CK(hipDeviceSynchronize());
    hipStream_t st = rag::lib_stream();
    float* Qt;  // the queries as the kernel's LDS image (flat_mfma_pipe.h)
    CK(hipMalloc(&Qt, (size_t)rag::mfma::pipe_tiled_words(p.col_tiles, ldw) * 4));
    if (rag::mfma::pipe_tile_queries((const float*)Q, nq, ldw, p.col_tiles, Qt, st)) { fprintf(stderr, "%s\n", rag_last_error()); return 1; }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, st));
        const rag::mfma::Args args{(const float*)X, n, ldw, (const float*)Qt, nq, 100, 0, ck, ci, nullptr, nullptr, zero, st,
                                   nullptr, thr, counts};
        if (rag::mfma::launch_pipe<true>(p, args)) {
            fprintf(stderr, "%s\n", rag_last_error()); return 1;
        }
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
#ifdef RAG_PIPE_CHECK
        unsigned long long chk[8];
        CK(hipMemcpyFromSymbol(chk, HIP_SYMBOL(rag::mfma::g_pipe_check), sizeof chk));
        printf("address check: %llu out of range; first: offset %lld tag %llu block %llu thread %llu\n", chk[0], (long long)chk[1],
               chk[2], chk[3] >> 32, chk[3] & 0xffffffffULL);
#endif
#ifdef RAG_PIPE_STAMPS
        unsigned long long stp[8], zero8[8] = {0};
        CK(hipMemcpyFromSymbol(stp, HIP_SYMBOL(rag::mfma::g_pipe_stamps), sizeof stp));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(rag::mfma::g_pipe_stamps), zero8, sizeof zero8));
        if (stp[4]) printf("stamps per wave and stage: vmcnt wait %.0f cycles, barrier %.0f, stage %.0f (%llu waves)\n", (double)stp[0] / stp[4],
                           (double)stp[1] / stp[4], (double)stp[2] / stp[4], stp[3]);
#endif
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double tf = 2.0 * nq * (double)n * d / (ms * 1e-3) / 1e12;
        printf("exp %d NS %d S %d col_tiles %d %s: %.3f ms  %.0f TFLOP/s (%.1f %% of 2500)\n",
#ifdef RAG_PIPE_EXP
               RAG_PIPE_EXP,
#else
               0,
#endif
               p.NS, p.S, p.col_tiles, zeros ? "zeros" : "random", ms, tf, tf / 25.0);
    }
    return 0;
}
