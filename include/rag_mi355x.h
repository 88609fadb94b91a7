/*
 * rag_mi355x.h -- C ABI of librag_mi355x.so, the MI355X (gfx950) engine behind
 * RAG-system's search path.
 *
 * Every entry point replaces one call the reference makes into a third-party
 * numeric library (faiss / rank_bm25 / scikit-learn).  The citation on each
 * declaration is the reference call site (paths relative to the reference
 * checkout) whose arithmetic the entry point performs on the GPU.
 *
 * Conventions
 *   - All functions return 0 on success and a negative RAG_ERR_* code on
 *     failure; rag_last_error() returns a thread-local message for the last
 *     failure on the calling thread.
 *   - The caller owns every host buffer; the library owns device memory.
 *   - Pointers named *_dev are device pointers (HBM) on the device selected by
 *     rag_init(); everything else is host memory.
 *   - `stream` arguments are a hipStream_t passed as void*; NULL is HIP's null
 *     stream, as in every HIP API (rag_lib_stream() returns the library's own
 *     non-blocking stream, the one the host entry points run on).  *_dev entry
 *     points are asynchronous on `stream` and ordered with the caller's other
 *     work on that stream only.
 *   - One mutex per handle: add/search on one handle may be called from
 *     different host threads (the reference UI does, UI_manager.py:753,883).
 *     A handle's search workspaces are shared by its searches: a search on a
 *     different stream than the handle's previous one first waits (on the
 *     device) for everything enqueued on that previous stream, which must
 *     therefore still exist.  Searches on one handle never overlap on the GPU;
 *     use one handle per stream for concurrency.
 *   - There is NO CPU fallback: without a usable GPU rag_init() fails with
 *     RAG_ERR_NO_DEVICE and every other call fails with RAG_ERR_NOT_INIT.
 */
#ifndef RAG_MI355X_H
#define RAG_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAG_OK                 0
#define RAG_ERR_INVALID_ARG   -1
#define RAG_ERR_NO_DEVICE     -2
#define RAG_ERR_NOT_INIT      -3
#define RAG_ERR_HIP           -4
#define RAG_ERR_OOM           -5
#define RAG_ERR_UNSUPPORTED   -6

#define RAG_METRIC_IP 0   /* faiss.METRIC_INNER_PRODUCT, IndexFlatIP  */
#define RAG_METRIC_L2 1   /* faiss.METRIC_L2 (squared), IndexFlatL2   */

#define RAG_DTYPE_F32  0  /* rows stored as float32 (faiss IndexFlat)      */
#define RAG_DTYPE_BF16 1  /* rows stored as bfloat16 (round-to-nearest-even); queries are rounded
                             the same way at search time; exact products, fp32 accumulate.  No
                             counterpart in the reference (BASELINE.json config 3).            */

/* Largest k one fused scan+select pass serves; larger k (up to RAG_MAX_K_TOTAL) is served in
 * ceil(k/256) passes over the matrix.  (faiss-gpu's own limit is 2048; the reference issues
 * k in {top_k, 3*top_k, 9*top_k} = {10,30,90} at its default top_k.)  rag_topk_merge_dev and
 * the sparse top-k accept k <= RAG_MAX_K. */
#define RAG_MAX_K 256
#define RAG_MAX_K_TOTAL 16384

typedef struct rag_flat rag_flat_t;
typedef struct rag_csc  rag_csc_t;

/* ---- library ---------------------------------------------------------- */

/* Select HIP device `device_id` for this process (one process per GPU) and
 * create the library stream.  Idempotent for the same device. */
int rag_init(int device_id);
/* Thread-local text of the last error on this thread ("" if none). */
const char* rag_last_error(void);
/* Number of HIP devices visible (0 if none / no driver). */
int rag_device_count(void);
/* ABI version of this header (bumped on incompatible change; 2: a NULL stream is HIP's null
 * stream, no longer the library's own). */
int rag_abi_version(void);
/* The library's own stream (hipStream_t, created non-blocking by rag_init): host entry points
 * such as rag_flat_search run and synchronise on it.  No counterpart in the reference. */
int rag_lib_stream(void** out);

/* ---- dense flat index -------------------------------------------------- */

/* faiss.IndexFlatIP(d) / IndexFlatL2(d): vector_storage.py:79,146,151. */
int rag_flat_create(int d, int metric, int storage_dtype, rag_flat_t** out);
/* Destroy the index and free its HBM. */
void rag_flat_free(rag_flat_t* h);
/* Pre-size the HBM arena for `n_capacity` rows (optional). */
int rag_flat_reserve(rag_flat_t* h, int64_t n_capacity);
/* index.add(x): vector_storage.py:105 (one row per call) and :152 (bulk).
 * x is row-major n*d float32 on the host.  Rows are staged and flushed to HBM
 * in large copies; ids are assigned sequentially from ntotal. */
int rag_flat_add(rag_flat_t* h, int64_t n, const float* x);
/* Same as rag_flat_add but the rows are already in HBM (n*d float32). */
int rag_flat_add_dev(rag_flat_t* h, int64_t n, const float* x_dev, void* stream);
/* Append n synthetic L2-normalised rows generated on the device by the
 * counter-based generator of oracle/flat_ref.c:synth_row() (bit-identical),
 * keyed by (seed, first_global_row + i).  Benchmark / test input only; no
 * counterpart in the reference. */
int rag_flat_add_synthetic(rag_flat_t* h, int64_t n, uint64_t seed, int64_t first_global_row);
/* index.ntotal: vector_storage.py:193. */
int64_t rag_flat_ntotal(const rag_flat_t* h);
/* Text naming the kernels and launch shape that served the handle's last search ("stream list G1 RG4
 * KS1 NB2 blocks256", "wide filter ...", "tile list ...", "scan"): lets tests and benches assert the
 * dispatch instead of assuming it.  Valid until the next search on the handle.  No counterpart in
 * the reference. */
const char* rag_flat_last_kernel(const rag_flat_t* h);
/* index.d */
int rag_flat_dim(const rag_flat_t* h);
/* index.reset() (rebuild_faiss_index creates a fresh index, :146,151). */
int rag_flat_reset(rag_flat_t* h);
/* D, I = index.search(q, k): vector_storage.py:196.
 * q: nq*d float32; D: nq*k float32; I: nq*k int64, best first.
 * IP: descending inner product; L2: ascending squared distance.
 * Ties: lower id first.  k > ntotal: tail padded with I=-1 and
 * D=-FLT_MAX (IP) / +FLT_MAX (L2).  1 <= k <= RAG_MAX_K_TOTAL. */
int rag_flat_search(rag_flat_t* h, int64_t nq, const float* q, int k, float* D, int64_t* I);
/* Opt-in bfloat16 prefilter of an fp32 index.  No counterpart in the reference (vector_storage.py:196 is a
 * plain faiss IndexFlat search); the results stay those of the fp32 rows.  enable != 0: the index keeps a
 * bfloat16 copy of its rows (+50 % HBM, filled by the next search and kept up to date by later ones).
 * Searches with k <= 26 on shards of "prefilter_min_rows" rows or more then stream the copy for 64 (up to
 * 8 queries, k <= 16) or 104 candidates per query, score the candidates again from the fp32 rows (float64
 * accumulation, one rounding to fp32) and check per query that no other row can reach the top k:
 * |bf16 score - exact score| <= (2^-7 + 2^-16 + (d+16) 2^-23) ||x|| ||q||.  A query whose check does not
 * close is searched again on the fp32 rows (device side, same call).  Scores may differ from the plain
 * fp32 search in the last bits (another summation order); ids only between rows whose fp32 scores tie.
 * enable == 0: the copy is freed.  bfloat16-stored indexes: RAG_ERR_UNSUPPORTED. */
int rag_flat_prefilter(rag_flat_t* h, int enable);
/* How many queries of the last search on this handle the prefilter could not prove and searched again on
 * the fp32 rows (0 when that search did not use the prefilter).  Waits for the device.  Diagnostic: a
 * corpus whose neighbours lie closer together than a bfloat16 rounding pays for both passes. */
int rag_flat_prefilter_redone(rag_flat_t* h, int64_t* n_queries);
/* rag_flat_search in two halves, for callers whose next step does not depend on the result yet:
 * _begin copies the queries, enqueues the search on the handle's own stream and returns; _end waits
 * for it and delivers D / I (nq x k, as rag_flat_search).  search_engine.py:266-267 (and :230-231,
 * :170-173) ask for the dense ranking and then the BM25 ranking of the same query, one after the
 * other; behind _begin the BM25 / TF-IDF scoring runs beside the dense pass.  One begun search per
 * handle; other searches and adds on the handle are refused / ordered behind it until _end.
 * nq * (d + 3k) * 4 bytes must stay under 4 MiB (the results are written straight into a pinned
 * staging buffer by the search's last kernel). */
int rag_flat_search_begin(rag_flat_t* h, int64_t nq, const float* q, int k);
int rag_flat_search_end(rag_flat_t* h, float* D, int64_t* I);
/* Same, device-resident query and outputs, asynchronous on `stream`;
 * returned ids are id_base + local row (id_base = first global row of this
 * shard when the corpus is row-sharded over ranks). */
int rag_flat_search_dev(rag_flat_t* h, int64_t nq, const float* q_dev, int k,
                        float* D_dev, int64_t* I_dev, int64_t id_base, void* stream);
/* rag_flat_search_dev for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same inputs, outputs and
 * results as rag_flat_search_dev (same kernels), but its workspaces -- padded queries, query norms, the
 * per-block candidate lists -- live in a static 8.4 MiB device array of the library instead of buffers
 * grown on demand, and it calls no hipMalloc / hipFree / allocating routine.  Limits, rejected with
 * RAG_ERR_INVALID_ARG / RAG_ERR_UNSUPPORTED: nq <= 8, k <= 128, at least 16 rows, rows short enough for the
 * LDS-resident queries (any d <= 2048 qualifies), no rows still staged on the host by rag_flat_add (search
 * once with the allocating entry point, or add in bulk), L2 only once the row norms exist (any earlier
 * allocating search computes them).  Every call on a GPU shares that GPU's array: calls that overlap on
 * one GPU (two streams, two handles) race; issue them on one stream or serialise them. */
int rag_flat_search_dev_noalloc(rag_flat_t* h, int64_t nq, const float* q_dev, int k,
                                float* D_dev, int64_t* I_dev, int64_t id_base, void* stream);
/* Rows [first, first+n) back to the host as float32 (faiss.write_index,
 * vector_storage.py:231, needs the raw vectors). */
int rag_flat_copy_to_host(rag_flat_t* h, int64_t first, int64_t n, float* out);
/* Merge `nlists` per-shard results (layout [nlists][nq][k], each list sorted
 * best-first with pads last) into the global top-k.  Runs after the RCCL
 * all-gather of per-rank candidates; metric decides the order. */
int rag_topk_merge_dev(int metric, int nlists, int64_t nq, int k,
                       const float* D_lists_dev, const int64_t* I_lists_dev,
                       float* D_dev, int64_t* I_dev, void* stream);
/* Same merge with the lists `stride_D` floats / `stride_I` int64 apart (>= nq*k): lets the
 * per-rank scores and ids travel in ONE packed all-gather buffer (rag_system_amd/sharded.py). */
int rag_topk_merge_strided_dev(int metric, int nlists, int64_t nq, int k,
                               const float* D_lists_dev, const int64_t* I_lists_dev,
                               int64_t stride_D, int64_t stride_I,
                               float* D_dev, int64_t* I_dev, void* stream);
/* Kernel-time accounting for the dense scan kernel (the dominant kernel):
 * when enabled, every search brackets the scan kernel with HIP events on the
 * stream it is launched on. */
int rag_flat_profiling(rag_flat_t* h, int enable);
/* Synchronises, then returns the number of timed scan launches since the last
 * call, the sum of their durations (ms) and the algorithmic bytes per launch
 * (ntotal*d*sizeof(elem)). */
int rag_flat_timing(rag_flat_t* h, int64_t* n_launches, double* total_ms, double* bytes_per_launch);
/* Tuning knobs for A/B runs in one process: "scan_blocks", "scan_nt",
 * "scan_rows", "mfma_min_nq", "mfma_shape" (1..9), "mfma_bpc", "mfma_stages" (2..3),
 * "mfma_filter_k", "mfma_filter_min_rows", "mfma_filter_step" (when the batch kernels use thresholds +
 * append buffers, and how densely the rows are sampled for the thresholds),
 * "stream_max_nq" (-1: never), "stream_rg", "stream_bpc", "stream_nb", "stream_fuse" (-1: separate merge launch), "mfma_pipe_sample" (-1: list-mode sample pass) (register-fed kernel for up to 16 queries),
 * "wide_max_nq" (-1: never), "wide_min_nq", "wide_wpb" (register-fed kernel for 9..64 queries),
 * "sparse_topk_lists" (1: BM25 / TF-IDF top-k by the list kernels only, no threshold path),
 * "sparse_fused" (0: BM25 / TF-IDF scoring by one launch per query term instead of the fused kernel).
 * Unknown name -> RAG_ERR_INVALID_ARG.  The selected ids never depend on them; scores are
 * bit-identical within one kernel and equal up to the fp32 summation order across kernels
 * (tests/test_flat_gpu.py, test_flat_stream_gpu.py, test_flat_wide_gpu.py). */
int rag_set_option(const char* name, int64_t value);

/* ---- sparse scoring (BM25Okapi, TF-IDF cosine) ------------------------- */

/* Term-major (CSC) postings in HBM: for term t, postings
 * [indptr[t], indptr[t+1]) hold (row[p], val[p]) = (chunk/doc index, weight).
 * BM25: val = term frequency of t in the chunk (BM25Okapi state built at
 * vector_storage.py:170,305).  TF-IDF: val = the L2-normalised tf-idf weight
 * (TfidfVectorizer.fit_transform, vector_storage.py:216-217). */
int rag_csc_create(int64_t n_terms, int64_t n_rows, const int64_t* indptr,
                   const int32_t* row, const double* val, rag_csc_t** out);
/* The postings of a tokenised corpus, built on the device (sort + segment) instead of the
 * per-document dictionary loops of rank_bm25's BM25Okapi.__init__ (called at
 * vector_storage.py:170,305 on the corpus tokenised at :162).  Document i =
 * words[doc_indptr[i] .. doc_indptr[i+1]), words in [0, n_words).  Outputs (host, caller-
 * allocated): *n_terms distinct words; term_word[t] (capacity n_words) = the word of term t,
 * terms numbered by first appearance (the order of rank_bm25's dictionaries); indptr
 * (capacity n_words + 1); row / tf (capacity n_tokens, *nnz used): the postings of term t,
 * ascending by document, tf = term frequency.  Feed them to rag_csc_create. */
int rag_postings_build(int64_t n_docs, const int64_t* doc_indptr, int64_t n_tokens, const int32_t* words,
                       int64_t n_words, int64_t* n_terms, int32_t* term_word, int64_t* indptr, int32_t* row,
                       double* tf, int64_t* nnz);
void rag_csc_free(rag_csc_t* h);
/* BM25Okapi parameters: per-term idf (epsilon floor already applied), per-row
 * token count, avgdl, k1, b (rank_bm25 BM25Okapi.__init__/_calc_idf). */
int rag_bm25_set(rag_csc_t* h, const double* idf, const int32_t* doc_len,
                 double avgdl, double k1, double b);
/* scores = bm25.get_scores(tokens): vector_storage.py:178, then the stable
 * descending sort + [:top_k] of :180.  term_ids are the query tokens in query
 * order (repeats allowed, out-of-vocabulary tokens dropped by the caller:
 * they add 0).  scores_or_null (n_rows float64) receives the dense score
 * vector if non-NULL; D/I (k entries, k may exceed n_rows -> padded with
 * I=-1, D=-DBL_MAX) receive the top-k (score desc, index asc).  k may be 0. */
int rag_bm25_scores(rag_csc_t* h, int n_qterms, const int32_t* term_ids,
                    double* scores_or_null, int k, double* D, int64_t* I);
/* cosine_similarity(q, M)[0]: search_engine.py:284-285,330-331.  M's rows
 * are unit-norm (val above), q_weights is the L2-normalised query row;
 * term_ids ascending as scipy stores them.  Same outputs as rag_bm25_scores. */
int rag_tfidf_scores(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights,
                     double* scores_or_null, int k, double* D, int64_t* I);
/* Rank (0-based position in the stable descending sort of the LAST score
 * vector computed on this handle) of each of n_query rows: search_engine.py
 * :332-336 sorts ALL documents; RRF needs 1/(60+rank) only for candidates. */
int rag_csc_ranks(rag_csc_t* h, int n_query, const int64_t* rows, int64_t* ranks);
/* Scores of n_query rows out of the LAST score vector computed on this handle:
 * search_engine.py:286 builds a dict over ALL documents, of which the fusion
 * (:301-311) only reads the candidates' entries. */
int rag_csc_gather(rag_csc_t* h, int n_query, const int64_t* rows, double* out_scores);
/* counts[i] = number of rows j of the LAST score vector with (score_j > keys[i]) or (score_j == keys[i] and
 * j < ids[i]): how many rows of this index stand ahead of an arbitrary (score, id) pair in the stable descending
 * order of vector_storage.py:180 / search_engine.py:332-336.  ids are in this index' row numbering and may lie
 * outside [0, n_rows) -- rows of ANOTHER shard (id - this shard's first row): a document's rank in a row-sharded
 * index is the sum of its count over the shards ("sparse indices shard the same way by document
 * rows").  With a pair that is one of this index' own rows this is rag_csc_ranks. */
int rag_csc_count_ahead(rag_csc_t* h, int n, const double* keys, const int64_t* ids, int64_t* counts);
/* rag_tfidf_scores (top-k only) + rag_csc_ranks + rag_csc_gather in ONE call and one
 * synchronisation: what a hybrid query needs from the TF-IDF ranking (search_engine.py:284-286 scores
 * and :332-336 ranks ALL documents; the fusions at :301-311 / :338-345 read the head of the ranking and
 * the candidates' own entries).  n_query <= 1024; ranks / row_scores may be NULL.  Identical outputs
 * to the three calls. */
int rag_tfidf_query(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights, int k, double* D,
                    int64_t* I, int n_query, const int64_t* rows, int64_t* ranks, double* row_scores);
/* rag_tfidf_query in two halves, for the caller that knows the query before it knows the rows it wants ranked:
 * search_engine.py:330-336 scores the query against the TF-IDF matrix and sorts ALL documents after the dense
 * and the BM25 rankings are in; the scoring does not depend on them.  _begin enqueues scoring and the top-k
 * selection (k = an upper bound) on the handle's own stream and returns; _end waits for it, delivers the first
 * k (<= the bound) entries of the ranking and then ranks / gathers `rows` from the score vector left in HBM
 * (three short launches, one synchronisation).  One begun query per handle; other calls on the handle are
 * refused until _end.  RAG_ERR_UNSUPPORTED when the one-launch scoring path does not serve the index or k. */
int rag_tfidf_query_begin(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights, int k);
int rag_tfidf_query_end(rag_csc_t* h, int k, double* D, int64_t* I, int n_query, const int64_t* rows, int64_t* ranks,
                        double* row_scores);

/* ---- candidate-set kernels -------------------------------------------- */

/* reciprocal_rank_fusion(*rankings, k=60): search_engine.py:352-359.
 * rankings are concatenated in `ids` (already in ranking order, best first),
 * lens[r] entries each; every entry adds 1/(k_const+rank) to its id's score
 * in ranking order.  Output: distinct ids in first-appearance order with their
 * fused float64 scores; *n_out = number of distinct ids (<= sum lens). */
int rag_rrf(int n_rankings, const int64_t* ids, const int64_t* lens, double k_const,
            int64_t* out_ids, double* out_scores, int64_t* n_out);
/* mmr_selection: search_engine.py:396-468.  cand: C*d float32 candidate
 * embeddings (un-normalised), q: d float32 query embedding; byte_len: UTF-8
 * length of each candidate text; max_bytes < 0 = no byte budget.
 * out_order receives the selected candidate indices in selection order. */
int rag_mmr(int C, int d, const float* cand, const float* q, double lambda, int top_k,
            const int64_t* byte_len, int64_t max_bytes, int32_t* out_order, int* out_n);

/* mmr_selection with the candidates' embeddings taken from the flat index itself: candidate i is row
 * rows[i] of `h` (a chunk the index holds) instead of a re-encoding of its text by the embedding model
 * (search_engine.py:418 calls model.encode on every candidate).  Same selection rule and outputs as
 * rag_mmr; identical results whenever the stored rows equal the re-encodings. */
int rag_flat_mmr(rag_flat_t* h, int C, const int64_t* rows, const float* q, double lambda, int top_k,
                 const int64_t* byte_len, int64_t max_bytes, int32_t* out_order, int* out_n);

/* The tail of a strategy=rrf query in ONE call (search_engine.py:340-347 reciprocal_rank_fusion over the three
 * rankings + sort + cut, :121-129 knee cutoff and threshold of search_with_strategy, :131-133 / :396-468 MMR):
 * one rank query, one single-workgroup kernel, one synchronisation -- the candidates never visit the host between
 * the stages (rag_rrf, rag_tfidf_query_end and rag_flat_mmr are three synchronous calls for the same work; they
 * stay, and tests compare this call with them bit for bit).
 *   tfidf        a handle with a query BEGUN by rag_tfidf_query_begin with k >= top_k + (distinct documents of the
 *                two rankings); this call ends it (on RAG_OK).  The full TF-IDF ranking of :332-336 enters as the
 *                ranks of the candidates plus the head of the ranking, as rag_system_amd/search_engine.py derives.
 *   dense_docs   the dense document ranking (best first, :321), each document as its ROW of the TF-IDF index, or
 *                as a negative code of the caller's choice when it has none; bm25_docs the same for :322 (documents
 *                may repeat; every entry counts, :356-358).  At most 128 entries each; top_k <= 64.
 *   use_knee / threshold   as search_with_strategy's use_knee_detection / threshold.
 *   flat, q, lambda, max_bytes   MMR over the entries that pass, with each document's embedding = the row of
 *                `flat` named by rag_csc_set_doc_tables (mmr_selection(..., reuse_index_vectors=True)); flat NULL =
 *                no MMR.  *mmr_done = 0 when some document has no such row (tables missing, -1, rows too long for
 *                the kernel's LDS): the caller then runs mmr_selection on the kept entries itself.
 * Outputs: fused_docs / fused_scores [top_k] with *n_fused entries, best first (codes as given, TF-IDF rows for
 * documents of the head); *n_kept of them pass knee + threshold (a subsequence in the same order: recomputed by
 * the caller from the scores); order [top_k] = *n_sel indices into the kept entries in MMR selection order.
 * RAG_ERR_UNSUPPORTED (limits, too short a head, selection overflow) leaves the query begun: end it with
 * rag_tfidf_query_end and take the three-call route. */
int rag_hybrid_rrf_mmr(rag_csc_t* tfidf, rag_flat_t* flat, int n_dense, const int64_t* dense_docs, int n_bm25,
                       const int64_t* bm25_docs, double k_const, int top_k, double threshold, int use_knee,
                       const float* q, double lambda, int64_t max_bytes, int* n_fused, int64_t* fused_docs,
                       double* fused_scores, int* n_kept, int* mmr_done, int* n_sel, int32_t* order);
/* Side tables for rag_hybrid_rrf_mmr's MMR stage, one entry per row of the (TF-IDF) index: flat_row = the row of
 * the flat index whose vector is the document's embedding (a document that is exactly one stored chunk:
 * search_engine.py:415,418 would re-encode that very text), -1 = none; text_bytes = UTF-8 length of the text MMR
 * budgets with (:437,:458).  Both NULL drops the tables.  Copied to HBM; refused while a query is begun. */
int rag_csc_set_doc_tables(rag_csc_t* h, const int64_t* flat_row, const int64_t* text_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RAG_MI355X_H */
