// common.h -- shared host-side plumbing of librag_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdarg>
#include <mutex>
#include <string>

#include "../../include/rag_mi355x.h"

namespace rag {

void set_error(const char* fmt, ...);
int  fail(int code, const char* fmt, ...);
bool initialized();
int  device_id();
hipStream_t lib_stream();
int  num_cus();

// Process-wide tuning knobs (rag_set_option).
struct Options {
    int64_t scan_blocks = 0;   // 0 = auto (CUs * blocks_per_cu)
    int64_t scan_nt = 1;       // non-temporal loads for the streamed matrix
    int64_t scan_rows = 0;     // rows per wave tile (0 = auto)
    int64_t mfma_min_nq = 0;   // nq >= this -> MFMA batch kernel (0 = auto: 3 for fp32 rows, 2 for bfloat16)
    int64_t mfma_shape = 0;    // force MFMA block shape 1..8 (0 = by batch size)
    int64_t mfma_stages = 0;   // force LDS ring depth 2..3 (0 = auto)
    int64_t mfma_bpc = 0;      // force MFMA blocks per CU used to size the row split (0 = auto)
    int64_t mfma_filter_k = 0;             // k >= this -> MFMA filter mode (0 = auto: 16 up to 128 queries, else 33)
    int64_t mfma_filter_min_rows = 1 << 19;  // ... on shards of at least this many rows
    int64_t mfma_filter_step = 0;            // sample every N-th row tile (0 = auto: 32, 64 for 256+ queries on 4M+ rows)
    int64_t stream_max_nq = 0;  // batches up to this size take the register-fed stream kernel (0 = auto: 16; -1 = never)
    int64_t stream_rg = 0;      // row groups of 4 per wave tile, 2 or 4 (0 = auto)
    int64_t stream_bpc = 0;     // stream kernel blocks per CU (0 = auto)
    int64_t stream_nb = 0;      // stream kernel ring depth, 2 or 3 (0 = auto)
    int64_t mfma_pipe_sample = 0;  // -1: the pipelined tile kernel's filter flow samples with the list-mode tile kernel
    int64_t sparse_rank_one = 0;    // -1: rag_tfidf_query_end ranks with the three rank kernels instead of the one-launch kernel
    int64_t wide_nb = 0;            // 3 / 4: ring depth of the wide kernel for up to two query tiles (0: by padding)
    int64_t prefilter_kp = 0;       // candidates per query of the bfloat16 prefilter (rag_flat_prefilter); 0: 64 (<= 8 queries, k <= 16) / 104
    int64_t prefilter_min_rows = 32768;  // smaller shards are searched directly
    int64_t stream_fuse = 0;    // -1: 1-2 query list launches leave the merge of the block lists to flat_merge_kernel
    int64_t wide_max_nq = 0;    // batches up to this size take the 16x16 register-fed kernel (0 = auto: 64; -1 = never)
    int64_t wide_min_nq = 0;    // ... from this size (0 = auto: 9; smaller batches and small shards stay with the stream kernel)
    int64_t wide_wpb = 0;       // waves per block of that kernel (0 = auto by LDS use)
    int64_t sparse_topk_lists = 0;  // 1: BM25 / TF-IDF top-k always by the list kernels (no threshold path)
    int64_t sparse_fused = 1;       // 0: BM25 / TF-IDF scoring by one launch per query term (the pre-fusion path)
};
Options& options();
// flat.hip: read-only view of a flat index' arena after flushing its staged rows (rows of `ld` elements, fp32 or
// bfloat16).  `use` runs WITH THE HANDLE'S MUTEX HELD and must have ENQUEUED every kernel that reads the arena
// before it returns: a concurrent rag_flat_add may then grow the arena, but its hipFree of the old one waits for
// the kernels already enqueued (the reference's UI adds documents beside its query thread, UI_manager.py:753,883).
struct FlatView {
    const void* arena;
    int ld, d, bf16;
    int64_t n_rows;
};
int flat_with_view(rag_flat_t* h, int (*use)(const FlatView&, void*), void* ctx);
// sparse.hip: a begun TF-IDF query (rag_tfidf_query_begin) seen by rag_hybrid_rrf_mmr (cand.hip)
struct CscBegunView {
    hipStream_t st;                   // the handle's own stream: scoring + selection are enqueued there
    int begun_k;                      // entries of the selected head
    int64_t n_rows;
    const double* head_D;             // pinned host memory, written by the selection kernel: the begun_k best scores
    const long long* head_I;          //   and their rows, in ranking order
    const int* overflow_flag;         // pinned: != 0 -> the head is not valid (selection buffer overflow)
    const unsigned long long* ranks;  // HBM: ranks of the rows given to csc_with_begun, in their order
    const long long* doc_row;         // HBM tables of rag_csc_set_doc_tables, or nullptr
    const long long* doc_bytes;
};
int csc_with_begun(rag_csc_t* h, int n_query, const int* rows, int (*use)(const CscBegunView&, void*), void* ctx,
                   bool* consumed);
int flat_init_scratch();  // flat.hip: address of the no-allocation entry point's scratch on the current device

}  // namespace rag

#define RAG_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return rag::fail(_e == hipErrorOutOfMemory ? RAG_ERR_OOM : RAG_ERR_HIP,          \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                      \
    } while (0)

#define RAG_REQUIRE_INIT()                                                             \
    do {                                                                               \
        if (!rag::initialized())                                                       \
            return rag::fail(RAG_ERR_NOT_INIT, "rag_init() has not succeeded in this process"); \
    } while (0)
