// flat_stream.h -- internal interface of the register-fed small-batch kernel (flat_stream.hip)
// used by flat.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace rag {

struct StreamPlan {
    int G;          // query groups of 4 (the kernel serves nq <= 4*G queries)
    int RG;         // row groups of 4 per wave tile
    int KS;         // 64-entry register slots per top-k list (list mode)
    int NB;         // segment buffers per wave (ring depth)
    bool bf16;      // rows and queries are bfloat16
    bool filt;      // filter mode: fixed per-query thresholds + append buffers instead of lists
    int cap;        // filter mode: entries per query in the append buffer
    int tile_step;  // visit only wave tiles t*tile_step (sample pass; else 1)
    int blocks;     // grid size = candidate lists written per query in list mode
    int nlists;     // = blocks
    size_t smem;
};

// false if this kernel does not serve the shape (nq > 16, k > 128, rows too long for the
// LDS-resident queries): the caller uses the MFMA tile kernel or the scan kernel instead.
bool stream_plan(long long n, long long nq, int k, int dch, bool bf16, bool filt, int cap, int tile_step,
                 StreamPlan* p);
// X / Q: rows of dch 16-byte chunks (fp32: 4 values, bfloat16: 8 values per chunk; Q has nq rows
// in the element type of X).  cand_*: list mode [nq][nlists][k], filter mode [nq][cap].
// xnorm / qnorm only for L2.
int stream_launch(const StreamPlan& p, const void* X, long long n, int dch, const void* Q, int nq, int k, int metric,
                  float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st,
                  const int* run_flag = nullptr, const float* thr_q = nullptr, unsigned int* counts = nullptr,
                  unsigned int* ticket = nullptr, float* D = nullptr, long long* I = nullptr, long long id_base = 0);
// List mode with 1-2 queries and k <= 16: with D / I (nq x k, device-writable) and a zeroed ticket word the block that
// finishes last merges the per-block lists itself and writes the final result (no flat_merge_kernel launch).
bool stream_can_fuse(const StreamPlan& p, long long nq, int k);

// ---- flat_wide.hip: 9..64 queries per pass, 16x16 MFMA fed from registers, sample + filter passes only
struct WidePlan {
    int NT;           // query tiles of 16
    bool bf16;
    int cap;          // filter pass: entries per query in the append buffer
    int sample_step;  // sample pass: every sample_step-th 16-row tile
    int blocks, wpb;  // grid, waves per block
    int nb;           // ring depth (step buffers per wave): 3 or 4
    int wpad;         // sample pass output entries per query (= blocks)
    size_t smem;
};
// false if the kernel does not serve the shape (nq > 64, fewer than 16 rows, queries beyond the LDS).
bool wide_plan(long long n, long long nq, int dch, bool bf16, int j, int cap, int sample_step, WidePlan* p);
// cand_*: [nq][wpad] (best score of every block over its sampled tiles; -FLT_MAX = none)
int wide_sample(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st);
// cand_*: [nq][cap]; counts zeroed by the caller
int wide_filter(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, const float* thr_q,
                unsigned int* counts, hipStream_t st);

}  // namespace rag
