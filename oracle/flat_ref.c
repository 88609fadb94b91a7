/*
 * oracle/flat_ref.c -- CPU restatement of the dense hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (rag_system_amd/) never does.
 *
 * What it restates
 *   vector_storage.py:196   distances, indices = self.index.search(query_vector, top_k)
 *   vector_storage.py:79    faiss.IndexFlatIP(d)  (and IndexFlatL2, named by BASELINE.json)
 * The arithmetic itself lives in faiss-cpu, an UNPINNED, un-vendored dependency
 * (requirements.txt:1) that is absent from this container, and the reference holds no
 * test, fixture or golden vector for it.  The only outputs of the real library available
 * offline are the ones its own documentation publishes: the "Getting started" IndexFlatL2
 * example of the faiss wiki (60 ids, 20 distances), which this file reproduces exactly
 * (tests/golden/published_answers.json, tests/test_published_answers.py; round 3).  Beyond
 * that example: PARITY UNPINNED.  The published algorithm of
 * faiss IndexFlat (exhaustive scan, fp32 inner product / squared L2 per row, k best kept
 * in a heap, results sorted best-first, tail padded with id -1 and -FLT_MAX (IP) /
 * +FLT_MAX (L2) when k > ntotal) is restated here.  Tie rule (equal score -> lower id
 * first) is this repo's deterministic choice; faiss only guarantees it for equal scores
 * in recent versions (CMin/CMax cmp2).
 *
 * flat_search_ref(..., accum=1) accumulates in double and is the ground truth the GPU
 * path is compared with (1e-5 relative, BASELINE.json north_star); accum=0 is the
 * sequential fp32 scan faiss performs for nq < 20 (summation order differs from faiss'
 * SIMD order, which is itself not specified).
 *
 * synth_row() is the counter-based synthetic-embedding generator of BASELINE.md section 2;
 * rag_flat_add_synthetic() in the HIP library produces bit-identical rows.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define METRIC_IP 0
#define METRIC_L2 1

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* Integer "bell" variate in [-131070, 131070]: sum of four 16-bit uniforms, centred. */
static inline int32_t synth_int(uint64_t row_key, int col) {
    uint64_t h = splitmix64(row_key + (uint64_t)col);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) +
                (int32_t)((h >> 32) & 0xFFFF) + (int32_t)((h >> 48) & 0xFFFF);
    return s - 131070;
}

static inline uint64_t synth_row_key(uint64_t seed, int64_t global_row) {
    return splitmix64(seed ^ ((uint64_t)global_row * 0xD1B54A32D192ED03ULL));
}

/* One L2-normalised row: x[c] = (float)( v[c] / sqrt(sum v^2) ), all in IEEE double
 * (integer sum of squares is exact), rounded once to float. */
void synth_row(uint64_t seed, int64_t global_row, int d, float* out) {
    uint64_t key = synth_row_key(seed, global_row);
    uint64_t ss = 0;
    for (int c = 0; c < d; ++c) {
        int64_t v = synth_int(key, c);
        ss += (uint64_t)(v * v);
    }
    double nrm = sqrt((double)ss);
    for (int c = 0; c < d; ++c) {
        double v = (double)synth_int(key, c);
        out[c] = (float)(v / nrm);
    }
}

/* float -> bfloat16 -> float, round-to-nearest-even on the upper 16 bits (NaN stays quiet):
 * the storage rounding of a RAG_DTYPE_BF16 index and of the queries searched against it.
 * Not in the reference (faiss IndexFlat is fp32); BASELINE.json config 3 names bf16 rows. */
static inline float bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) u = (u | 0x00400000u) & 0xFFFF0000u;
    else u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&f, &u, 4);
    return f;
}

void bf16_round_array(const float* in, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = bf16_rne(in[i]);
}

void synth_rows(uint64_t seed, int64_t first_global_row, int64_t n, int d, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) synth_row(seed, first_global_row + i, d, out + i * (int64_t)d);
}

/* ---- top-k list: key descending, id ascending ------------------------------------- */

typedef struct {
    int k;
    int cnt;
    double* key; /* larger is better */
    int64_t* id;
} topk_t;

static void topk_init(topk_t* t, int k) {
    t->k = k;
    t->cnt = 0;
    t->key = (double*)malloc(sizeof(double) * (size_t)k);
    t->id = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
}
static void topk_free(topk_t* t) {
    free(t->key);
    free(t->id);
}
static inline int ranks_ahead(double ka, int64_t ia, double kb, int64_t ib) {
    return (ka > kb) || (ka == kb && ia < ib);
}
static inline void topk_push(topk_t* t, double key, int64_t id) {
    if (t->cnt == t->k && !ranks_ahead(key, id, t->key[t->k - 1], t->id[t->k - 1])) return;
    int p = (t->cnt < t->k) ? t->cnt : t->k - 1;
    while (p > 0 && ranks_ahead(key, id, t->key[p - 1], t->id[p - 1])) {
        t->key[p] = t->key[p - 1];
        t->id[p] = t->id[p - 1];
        --p;
    }
    t->key[p] = key;
    t->id[p] = id;
    if (t->cnt < t->k) t->cnt++;
}

static inline double score_row(const float* x, const float* q, int d, int metric, int accum) {
    if (accum) {
        double s = 0.0;
        if (metric == METRIC_IP)
            for (int c = 0; c < d; ++c) s += (double)x[c] * (double)q[c];
        else
            for (int c = 0; c < d; ++c) {
                double t = (double)x[c] - (double)q[c];
                s += t * t;
            }
        return s;
    } else {
        float s = 0.0f;
        if (metric == METRIC_IP)
            for (int c = 0; c < d; ++c) s += x[c] * q[c];
        else
            for (int c = 0; c < d; ++c) {
                float t = x[c] - q[c];
                s += t * t;
            }
        return (double)s;
    }
}

static void emit(const topk_t* t, int metric, float* D, int64_t* I, double* D64) {
    for (int j = 0; j < t->k; ++j) {
        if (j < t->cnt) {
            double v = (metric == METRIC_IP) ? t->key[j] : -t->key[j];
            D[j] = (float)v;
            if (D64) D64[j] = v;
            I[j] = t->id[j];
        } else {
            D[j] = (metric == METRIC_IP) ? -FLT_MAX : FLT_MAX;
            if (D64) D64[j] = D[j];
            I[j] = -1;
        }
    }
}

/* D, I = IndexFlat{IP,L2}.search(Q, k) over stored rows X (n*d, row-major).
 * D64 (optional, nq*k) receives the un-rounded double scores when accum=1. */
void flat_search_ref(const float* X, int64_t n, int d, const float* Q, int64_t nq, int k,
                     int metric, int accum, float* D, int64_t* I, double* D64) {
#pragma omp parallel for schedule(dynamic, 1) if (nq > 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
        topk_t t;
        topk_init(&t, k);
        const float* q = Q + qi * (int64_t)d;
        for (int64_t r = 0; r < n; ++r) {
            double s = score_row(X + r * (int64_t)d, q, d, metric, accum);
            topk_push(&t, metric == METRIC_IP ? s : -s, r);
        }
        emit(&t, metric, D + qi * (int64_t)k, I + qi * (int64_t)k, D64 ? D64 + qi * (int64_t)k : NULL);
        topk_free(&t);
    }
}

/* Same search over the synthetic corpus rows [first_global_row, first_global_row+n),
 * generated on the fly (nothing stored), parallel over row blocks.  Ids are local
 * (0..n-1).  Used for full-size checks (10M x 384) without 15 GB of host memory. */
void flat_search_synth_ref2(uint64_t seed, int64_t first_global_row, int64_t n, int d, const float* Q,
                            int64_t nq, int k, int metric, int bf16_rows, float* D, int64_t* I, double* D64);

void flat_search_synth_ref(uint64_t seed, int64_t first_global_row, int64_t n, int d, const float* Q,
                           int64_t nq, int k, int metric, float* D, int64_t* I, double* D64) {
    flat_search_synth_ref2(seed, first_global_row, n, d, Q, nq, k, metric, 0, D, I, D64);
}

/* bf16_rows != 0: every generated row is rounded to bfloat16 before scoring (the caller
 * rounds Q the same way). */
void flat_search_synth_ref2(uint64_t seed, int64_t first_global_row, int64_t n, int d, const float* Q,
                            int64_t nq, int k, int metric, int bf16_rows, float* D, int64_t* I, double* D64) {
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    topk_t* lists = (topk_t*)malloc(sizeof(topk_t) * (size_t)nthreads * (size_t)nq);
    for (int64_t i = 0; i < (int64_t)nthreads * nq; ++i) topk_init(&lists[i], k);
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        float* row = (float*)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            synth_row(seed, first_global_row + r, d, row);
            if (bf16_rows)
                for (int c = 0; c < d; ++c) row[c] = bf16_rne(row[c]);
            for (int64_t qi = 0; qi < nq; ++qi) {
                double s = score_row(row, Q + qi * (int64_t)d, d, metric, 1);
                topk_push(&lists[(int64_t)tid * nq + qi], metric == METRIC_IP ? s : -s, r);
            }
        }
        free(row);
    }
    for (int64_t qi = 0; qi < nq; ++qi) {
        topk_t t;
        topk_init(&t, k);
        for (int th = 0; th < nthreads; ++th) {
            topk_t* s = &lists[(int64_t)th * nq + qi];
            for (int j = 0; j < s->cnt; ++j) topk_push(&t, s->key[j], s->id[j]);
        }
        emit(&t, metric, D + qi * (int64_t)k, I + qi * (int64_t)k, D64 ? D64 + qi * (int64_t)k : NULL);
        topk_free(&t);
    }
    for (int64_t i = 0; i < (int64_t)nthreads * nq; ++i) topk_free(&lists[i]);
    free(lists);
}

/* Merge per-shard lists [nlists][nq][k] (pads: id -1) into the global top-k: the CPU
 * restatement of the post-all-gather merge. */
void topk_merge_ref(int metric, int nlists, int64_t nq, int k, const float* Dl, const int64_t* Il,
                    float* D, int64_t* I) {
    for (int64_t qi = 0; qi < nq; ++qi) {
        topk_t t;
        topk_init(&t, k);
        for (int l = 0; l < nlists; ++l)
            for (int j = 0; j < k; ++j) {
                int64_t o = ((int64_t)l * nq + qi) * k + j;
                if (Il[o] < 0) continue;
                double s = Dl[o];
                topk_push(&t, metric == METRIC_IP ? s : -s, Il[o]);
            }
        emit(&t, metric, D + qi * (int64_t)k, I + qi * (int64_t)k, NULL);
        topk_free(&t);
    }
}

/* ---- CPU baseline (timed by bench.py): faiss-cpu-style flat scan ------------------- */

/* fp32 inner products of one query against n rows, 8 independent partial sums per row so
 * the compiler vectorises the way faiss' fvec_inner_product does; single thread per query
 * exactly like faiss for nq < 20 (its OpenMP loop is over queries). */
__attribute__((target_clones("avx2,fma", "default")))
static void scan_ip_1q(const float* X, int64_t n, int d, const float* q, topk_t* t) {
    for (int64_t r = 0; r < n; ++r) {
        const float* x = X + r * (int64_t)d;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int c = 0;
        for (; c + 8 <= d; c += 8)
            for (int u = 0; u < 8; ++u) acc[u] += x[c + u] * q[c + u];
        float s = ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
        for (; c < d; ++c) s += x[c] * q[c];
        if (t->cnt < t->k || s > (float)t->key[t->k - 1]) topk_push(t, (double)s, r);
    }
}

/* nthreads <= 1: faiss' batch-1 behaviour (one core).  nthreads > 1: rows split over
 * threads and merged (what a multi-threaded CPU scan could do; NOT what faiss does). */
void flat_search_cpu_baseline(const float* X, int64_t n, int d, const float* Q, int64_t nq, int k,
                              int nthreads, float* D, int64_t* I) {
    if (nthreads < 1) nthreads = 1;
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float* q = Q + qi * (int64_t)d;
        topk_t* parts = (topk_t*)malloc(sizeof(topk_t) * (size_t)nthreads);
        for (int th = 0; th < nthreads; ++th) topk_init(&parts[th], k);
        int64_t per = (n + nthreads - 1) / nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1) if (nthreads > 1)
        for (int th = 0; th < nthreads; ++th) {
            int64_t lo = th * per, hi = lo + per > n ? n : lo + per;
            if (lo < hi) {
                topk_t local;
                topk_init(&local, k);
                scan_ip_1q(X + lo * (int64_t)d, hi - lo, d, q, &local);
                for (int j = 0; j < local.cnt; ++j) topk_push(&parts[th], local.key[j], local.id[j] + lo);
                topk_free(&local);
            }
        }
        topk_t t;
        topk_init(&t, k);
        for (int th = 0; th < nthreads; ++th) {
            for (int j = 0; j < parts[th].cnt; ++j) topk_push(&t, parts[th].key[j], parts[th].id[j]);
            topk_free(&parts[th]);
        }
        free(parts);
        emit(&t, METRIC_IP, D + qi * (int64_t)k, I + qi * (int64_t)k, NULL);
        topk_free(&t);
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
