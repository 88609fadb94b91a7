// flat_mfma.h -- internal interface of the MFMA batch kernel (flat_mfma.hip) used by flat.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace rag {

struct MfmaPlan {
    int NW;         // waves per block (block covers 32*NW queries)
    int MT;         // 32-row accumulator tiles per wave (block covers 32*MT rows)
    int S;          // row splits = per-query candidate lists written
    int col_tiles;  // query tiles
};

// false if the MFMA path does not serve this shape (k > 64): caller falls back to the scan kernel.
bool mfma_plan(long long n, long long nq, int k, MfmaPlan* p);
// out[first + i] = ||X[first + i]||^2 for i in [0, n)
int mfma_row_norms(const float* X, int ld, long long first, long long n, float* out, hipStream_t st);
// Qp: queries zero-padded to (>= nq) x ld.  cand_*: [nq][S][k].  xnorm/qnorm only for L2.
int mfma_launch(const MfmaPlan& p, const float* X, long long n, int ld, const float* Qp, int nq, int k, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st);

}  // namespace rag
