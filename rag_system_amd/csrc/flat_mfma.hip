// flat_mfma.hip -- host side of the MFMA batch path: block-shape plan, row norms, dispatch.
// The kernel lives in flat_mfma_kernel.h and is instantiated per element type in
// flat_mfma_f32.hip / flat_mfma_bf16.hip.
#include "common.h"
#include "flat_mfma.h"
#include "flat_mfma_kernel.h"

namespace {

using namespace rag;

// ||x||^2 per row (fp32, same fmaf order on every call) for the L2 decomposition.
template <bool BF>
__global__ __launch_bounds__(256) void row_norms_kernel(const void* __restrict__ Xv, int ld, long long first,
                                                        long long n, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += W) {
        float ss = 0.f;
        for (int cidx = lane; cidx < ld; cidx += 64) {
            float v;
            if (BF) v = __uint_as_float((unsigned int)reinterpret_cast<const unsigned short*>(Xv)[(first + r) * ld + cidx] << 16);
            else v = reinterpret_cast<const float*>(Xv)[(first + r) * ld + cidx];
            ss = fmaf(v, v, ss);
        }
        for (int dist = 32; dist >= 1; dist >>= 1) ss += __shfl_xor(ss, dist);
        if (lane == 0) out[first + r] = ss;
    }
}

}  // namespace

namespace rag {

using mfma::SHAPES;
using mfma::NSHAPES;
using mfma::Shape;

bool mfma_plan(long long n, long long nq, int k, bool bf16, MfmaPlan* p) {
    if (k > 128 || nq < 1 || n < 1) return false;
    const int KS = k > 64 ? 2 : 1;
    int shape = nq > 128 ? 3 : (nq > 64 ? 2 : (nq > 32 ? 1 : 0));  // measured per batch size (benchmarks/ab/run_nq64_sweep.sh)
    const long long force = options().mfma_shape;
    if (force >= 1 && force <= 9) shape = (int)force - 1;  // 9: 64-query wave tiles
    if (KS == 2 && SHAPES[shape].WM * SHAPES[shape].WN > 4) shape = SHAPES[shape].BK == 32 ? 6 : 2;
    int NS = 3;
    if (options().mfma_stages >= 2 && options().mfma_stages <= 3) NS = (int)options().mfma_stages;
    const int NS0 = NS;
    int bpc = SHAPES[shape].bpc;
    if (options().mfma_bpc > 0) bpc = (int)options().mfma_bpc;
    // fit LDS: shrink the ring first, then fall back to a 4-wave shape (fewer per-wave lists)
    for (;;) {
        const Shape& sh = SHAPES[shape];
        if (mfma::smem_need(sh.WM, sh.WN, sh.MT, sh.NT, sh.BK, NS, k) * bpc <= 160 * 1024) break;
        if (NS > 2) { --NS; continue; }
        if (bpc > 1) { --bpc; NS = NS0; continue; }
        if (shape != 2) { shape = 2; bpc = 1; NS = 3; continue; }
        return false;
    }
    const Shape& sh = SHAPES[shape];
    p->shape = shape;
    p->WM = sh.WM;
    p->WN = sh.WN;
    p->MT = sh.MT;
    p->NS = NS;
    p->KS = KS;
    p->bf16 = bf16;
    p->filt = false;
    p->tile_step = 1;
    p->cap = 0;
    const int BN = 32 * sh.NT * sh.WN, BM = 32 * sh.MT * sh.WM;
    p->col_tiles = (int)((nq + BN - 1) / BN);
    const long long ntiles = (n + BM - 1) / BM;
    long long S = ((long long)num_cus() * bpc + p->col_tiles - 1) / p->col_tiles;
    if (S >= 8) S = (S + 7) & ~7LL;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    p->S = (int)S;
    p->nlists = p->S * sh.WM;
    return true;
}

bool mfma_plan_filter(long long n, long long nq, bool bf16, int cap, MfmaPlan* p, int ld_words) {
    if (nq < 1 || n < 1 || cap < 1) return false;
    int shape = nq > 128 ? 3 : (nq > 64 ? 2 : (nq > 32 ? 1 : 0));
    // more than 64 queries over bfloat16 rows of whole 128-byte lines: the pipelined kernel of
    // flat_mfma_pipe.h (config C3: 17.x ms per batch against 19.1 for shape 4; 65-128 queries, where its
    // 256-query tile is half empty: 10M x 768, k 100: 3.87-3.99 ms against 4.13-4.31 for the 128-query tile
    // kernel, 10M x 384: 2.18-2.23 against 2.52-2.61; benchmarks/ab/run_pipe_mid.sh)
    // fp32 rows, more than 128 queries (up to 192 go to the register-fed kernel of flat_wide.hip on large shards):
    // 10M x 384, nq 1024: filter pass 56.2 ms against 62.1 ms for the list-mode pass of shape 4 (r03)
    if (nq > (bf16 ? 64 : 128) && ld_words > 0 && ld_words % 32 == 0) shape = 9;
    const long long force = options().mfma_shape;
    if ((force >= 1 && force <= 4) || (force >= 7 && force <= 10)) shape = (int)force - 1;
    int NS = 3;
    if (options().mfma_stages >= 2 && options().mfma_stages <= 3) NS = (int)options().mfma_stages;
    if (shape == 9) {  // pipelined kernel: fixed rings (3 row stages, 2 query stages), whole 128-byte lines
        if (ld_words % 32 != 0) shape = 3;
        else NS = 3;
    }
    const Shape& sh = SHAPES[shape];
    int bpc = sh.bpc;
    if (shape != 9)
    while (mfma::smem_need(sh.WM, sh.WN, sh.MT, sh.NT, sh.BK, NS, 0) * bpc > 160 * 1024) {
        if (NS > 2) --NS;
        else if (bpc > 1) --bpc;
        else return false;
    }
    p->shape = shape;
    p->WM = sh.WM;
    p->WN = sh.WN;
    p->MT = sh.MT;
    p->NS = NS;
    p->KS = 1;
    p->bf16 = bf16;
    p->filt = true;
    p->tile_step = 1;
    p->cap = cap;
    const int BN = 32 * sh.NT * sh.WN, BM = 32 * sh.MT * sh.WM;
    p->col_tiles = (int)((nq + BN - 1) / BN);
    const long long ntiles = (n + BM - 1) / BM;
    long long S = ((long long)num_cus() * bpc + p->col_tiles - 1) / p->col_tiles;
    if (S >= 8) S = (S + 7) & ~7LL;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    p->S = (int)S;
    p->nlists = 1;
    return true;
}

int mfma_block_rows(const MfmaPlan& p) { return 32 * p.MT * p.WM; }

long long mfma_tiled_query_words(const MfmaPlan& p, int ld_words) {
    return p.shape == 9 ? mfma::pipe_tiled_words(p.col_tiles, ld_words) : 0;
}
int mfma_tile_queries(const MfmaPlan& p, const float* Qw, int nq, int ld_words, float* out, hipStream_t st) {
    return mfma::pipe_tile_queries(Qw, nq, ld_words, p.col_tiles, out, st);
}

int mfma_row_norms(const void* X, int ld, bool bf16, long long first, long long n, float* out, hipStream_t st) {
    if (n <= 0) return RAG_OK;
    long long nb = (n + 3) / 4;
    if (nb > (long long)num_cus() * 8) nb = (long long)num_cus() * 8;
    if (bf16) row_norms_kernel<true><<<(unsigned)nb, 256, 0, st>>>(X, ld, first, n, out);
    else row_norms_kernel<false><<<(unsigned)nb, 256, 0, st>>>(X, ld, first, n, out);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

int mfma_launch(const MfmaPlan& p, const float* X, long long n, int ld, const float* Qp, int nq, int k, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, const float* zero16,
                hipStream_t st, const int* run_flag, const float* thr_q, unsigned int* counts) {
    const mfma::Args a{X, n, ld, Qp, nq, k, metric, cand_key, cand_id, xnorm, qnorm, zero16, st, run_flag, thr_q, counts};
    return p.bf16 ? mfma::dispatch_bf16(p, a) : mfma::dispatch_f32(p, a);
}

}  // namespace rag
