// flat_mfma.h -- internal interface of the MFMA batch kernel (flat_mfma.hip) used by flat.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace rag {

struct MfmaPlan {
    int shape;      // index into the built block shapes (flat_mfma.hip SHAPES)
    int WM, WN;     // waves per block along rows / queries (block covers 32*WN queries)
    int MT;         // 32-row accumulator tiles per wave (block covers 32*MT*WM rows)
    int NS;         // LDS-DMA ring stages
    int KS;         // 64-entry register slots per top-k list (1: k <= 64, 2: k <= 128)
    bool bf16;      // rows and queries are bfloat16 (two per 4-byte word)
    bool filt;      // filter mode: fixed per-query thresholds + append buffers instead of lists
    int tile_step;  // visit only row tiles t*tile_step (sample pass of the filter mode; else 1)
    int cap;        // filter mode: entries per query in the append buffer
    int S;          // row splits
    int col_tiles;  // query tiles
    int nlists;     // candidate lists written per query = S * WM
};

// false if the MFMA path does not serve this shape (k > 128): caller falls back to the scan kernel.
bool mfma_plan(long long n, long long nq, int k, bool bf16, MfmaPlan* p);
// Filter-mode plan (no LDS lists, any k): block shape by batch size among the filter-mode shapes.
// ld_words: 4-byte words per row (block shape 10 needs a multiple of 16).
bool mfma_plan_filter(long long n, long long nq, bool bf16, int cap, MfmaPlan* p, int ld_words = 0);
// Block shape 10 (flat_mfma_pipe.h) reads its queries from a tiled image: words it needs, and the kernel
// that fills it from the nq x ld_words query array; the image is then passed to mfma_launch as Qp.
long long mfma_tiled_query_words(const MfmaPlan& p, int ld_words);
int mfma_tile_queries(const MfmaPlan& p, const float* Qw, int nq, int ld_words, float* out, hipStream_t st);
// Rows per block tile of a plan (for sizing sample passes).
int mfma_block_rows(const MfmaPlan& p);
// out[first + i] = ||X[first + i]||^2 for i in [0, n); ld in elements (fp32 or bfloat16 rows)
int mfma_row_norms(const void* X, int ld, bool bf16, long long first, long long n, float* out, hipStream_t st);
// X / Qp are addressed in 4-byte words (ld = words per row; bfloat16: two elements per word).
// Qp: queries zero-padded to (>= nq) x ld.  cand_*: [nq][nlists][k].  xnorm/qnorm only for L2.
// zero16: >= 16 bytes of zeros in HBM (source of out-of-range K columns / queries).
int mfma_launch(const MfmaPlan& p, const float* X, long long n, int ld, const float* Qp, int nq, int k, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, const float* zero16,
                hipStream_t st, const int* run_flag = nullptr, const float* thr_q = nullptr,
                unsigned int* counts = nullptr);

}  // namespace rag
