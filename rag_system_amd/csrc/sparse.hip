// sparse.hip -- BM25Okapi and TF-IDF cosine scoring over term-major postings (float64).
//
// Replaces rank_bm25's BM25Okapi.get_scores (vector_storage.py:178) + the stable descending
// sort of :180, and sklearn's cosine_similarity(q, M)[0] (search_engine.py:285,331).
//
// Compiled with -ffp-contract=off: every score is the same sequence of IEEE float64
// operations numpy / scipy perform on the host (no fused multiply-add), and query terms are
// applied one kernel launch after another on one stream, so each row's additions happen in
// query-term order exactly as `score += ...` does in the reference loop.  Within one term a
// row appears at most once, so the read-modify-write needs no atomics and is deterministic.
//
// HBM layout: term-major postings (CSC of the rows x terms matrix): indptr[n_terms+1] int64,
// row[nnz] int32, val[nnz] float64; dense float64 score vector of n_rows.
#include "common.h"
#include "topk.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

using namespace rag;

constexpr int TK_THREADS = 256;
constexpr int TK_WAVES = TK_THREADS / 64;
constexpr int MERGE_THREADS = 1024;
constexpr int MERGE_WAVES = MERGE_THREADS / 64;

// score[row] += idf * ( tf*(k1+1) / ( tf + k1*((1-b) + (b*len)/avgdl) ) )   -- rank_bm25 order
__global__ void bm25_term_kernel(const int* __restrict__ row, const double* __restrict__ val, long long begin,
                                 long long end, double idf, const int* __restrict__ doc_len, double avgdl,
                                 double k1, double b, double k1p1, double one_minus_b,
                                 double* __restrict__ score) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = begin + (long long)blockIdx.x * blockDim.x + threadIdx.x; p < end; p += stride) {
        const int r = row[p];
        const double tf = val[p];
        const double num = tf * k1p1;
        const double t1 = b * (double)doc_len[r];
        const double t2 = t1 / avgdl;
        const double t3 = one_minus_b + t2;
        const double t4 = k1 * t3;
        const double den = tf + t4;
        const double frac = num / den;
        const double c = idf * frac;
        score[r] = score[r] + c;
    }
}

// score[row] += w * val   -- scipy csr_matmat order (one query term at a time)
__global__ void tfidf_term_kernel(const int* __restrict__ row, const double* __restrict__ val, long long begin,
                                  long long end, double w, double* __restrict__ score) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = begin + (long long)blockIdx.x * blockDim.x + threadIdx.x; p < end; p += stride) {
        const int r = row[p];
        const double prod = w * val[p];
        score[r] = score[r] + prod;
    }
}

// Per-block top-k of a dense float64 score vector (key desc, index asc).
template <int KS>
__global__ __launch_bounds__(TK_THREADS) void dense_topk_kernel(const double* __restrict__ score, long long n,
                                                                int k, double* __restrict__ cand_key,
                                                                int* __restrict__ cand_id,
                                                                const int* __restrict__ run_flag) {
    if (run_flag != nullptr && *run_flag == 0) return;  // threshold path succeeded (uniform)
    __shared__ double mk[(TK_WAVES - 1) * KS * 64];
    __shared__ int mi[(TK_WAVES - 1) * KS * 64];
    // contiguous span per block so that in-wave ids ascend
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = lo + per < n ? lo + per : n;
    const long long C = hi > lo ? hi - lo : 0;
    WaveTopK<double, int, KS> tk;
    block_merge<double, int, KS, TK_WAVES>(
        [&](long long c, double& key, int& id) {
            key = score[lo + c];
            id = (int)(lo + c);
            return true;
        },
        C, k, tk, mk, mi);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const long long base = (long long)blockIdx.x * k;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int p = s * 64 + lane;
            if (p < k) {
                cand_key[base + p] = tk.key[s];
                cand_id[base + p] = tk.id[s];
            }
        }
    }
}

template <int KS>
__global__ __launch_bounds__(MERGE_THREADS) void dense_topk_merge_kernel(const double* __restrict__ cand_key,
                                                                         const int* __restrict__ cand_id,
                                                                         int nlists, int k, double* __restrict__ D,
                                                                         long long* __restrict__ I,
                                                                         const int* __restrict__ run_flag) {
    if (run_flag != nullptr && *run_flag == 0) return;
    __shared__ double mk[(MERGE_WAVES - 1) * KS * 64];
    __shared__ int mi[(MERGE_WAVES - 1) * KS * 64];
    WaveTopK<double, int, KS> tk;
    if (KS == 1 && k <= 32) {
        block_merge<double, int, KS, MERGE_WAVES>(
            [&](long long c, double& key, int& id) {
                key = cand_key[c];
                id = cand_id[c];
                return id != IdPad<int>::v();
            },
            (long long)nlists * k, k, tk, mk, mi);
    } else {
        block_merge_lists<double, int, KS, MERGE_WAVES>(
            [&](int l, int p, double& key, int& id) {
                const long long o = (long long)l * k + p;
                key = cand_key[o];
                id = cand_id[o];
                return id != IdPad<int>::v();
            },
            nlists, k, tk, mk, mi);
    }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int p = s * 64 + lane;
            if (p < k) {
                const bool pad = tk.id[s] == IdPad<int>::v();
                D[p] = pad ? -DBL_MAX : tk.key[s];
                I[p] = pad ? -1LL : (long long)tk.id[s];
            }
        }
    }
}

// ---- threshold path of the dense top-k (large score vectors) -----------------------------------
// Sorted inserts cost ~0.3-0.5 us each and every wave of dense_topk_kernel has to fill its own
// k-list first, so that path takes 80-160 us for a million scores although reading them takes
// 5 us.  Here: (1) SPANS = 1024 waves take the maximum of one contiguous span each; (2) the k-th
// largest span maximum is a threshold T with at least k scores >= T by construction (k spans
// hold one each) and, for scores in no particular order, few more than k; (3) every score > T is
// appended to a buffer; (4) one block takes the exact best min(k, count) of the buffer and, when
// fewer than k scores are strictly greater, completes the list with the lowest-numbered rows
// scoring exactly T (ties rank by ascending row: the result is the stable descending sort the
// reference takes, vector_storage.py:180).  A buffer overflow (scores sorted by row number, say)
// sets a flag and the list kernels above run instead; they return at once when the flag is clear.
constexpr int SPANS = 1024;
constexpr size_t PIN_BYTES = 64 << 10;
// rag_tfidf_query keeps the top-k at [0, 2K) / [32K, 34K) like every top-k call and adds:
constexpr size_t PIN_Q_ROWS = 4 << 10, PIN_Q_RANKS = 36 << 10, PIN_Q_SCORES = 44 << 10;  // 8 KiB each (RANK_MAXQ x 8)
constexpr int THR_CAP = 8192;

__global__ __launch_bounds__(256) void span_max_kernel(const double* __restrict__ score, long long n,
                                                       double* __restrict__ wmax) {
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long per = (n + SPANS - 1) / SPANS;
    const long long lo = w * per, hi = lo + per < n ? lo + per : n;
    double m = -DBL_MAX;
    for (long long i = lo + lane; i < hi; i += 64) {
        const double v = score[i];
        m = v > m ? v : m;  // NaN never wins
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if (lane == 0) wmax[w] = m;
}

// One block of SPANS threads: bitonic sort (descending) of the span maxima in LDS, T = k-th largest.
__global__ __launch_bounds__(SPANS) void kth_span_max_kernel(const double* __restrict__ wmax, int k,
                                                             double* __restrict__ thr, int* __restrict__ counters) {
    __shared__ double v[SPANS];
    const int t = threadIdx.x;
    v[t] = wmax[t];
    if (t < 3) counters[t] = 0;  // [0] scores > T appended, [1] overflow flag, [2] rows scoring exactly T
    __syncthreads();
    for (int size = 2; size <= SPANS; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int p = t ^ stride;
            if (p > t) {
                const bool desc = (t & size) == 0;
                const double a = v[t], b = v[p];
                if (desc ? a < b : a > b) {
                    v[t] = b;
                    v[p] = a;
                }
            }
            __syncthreads();
        }
    }
    if (t == 0) thr[0] = v[k - 1];
}

__global__ __launch_bounds__(256) void compact_gt_kernel(const double* __restrict__ score, long long n,
                                                         const double* __restrict__ thr, double* __restrict__ buf_key,
                                                         int* __restrict__ buf_id, int* __restrict__ eq_id,
                                                         int* __restrict__ counters) {
    const double T = thr[0];
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // whole waves iterate together (the ballot below needs every lane)
    const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long rounds = (n + stride - 1) / stride;
    for (long long r = 0; r < rounds; ++r) {
        const long long i = first + r * stride;
        const double v = i < n ? score[i] : -DBL_MAX;
        const bool take = i < n && v > T;
        const unsigned long long m = __ballot(take);
        if (m != 0ULL) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[0], __popcll(m));
            base = __shfl(base, 0, 64);
            if (take) {
                const int slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
                if (slot < THR_CAP) {
                    buf_key[slot] = v;
                    buf_id[slot] = (int)i;
                }
            }
        }
        // rows scoring exactly T: remembered while they are few (when they are many, the first
        // k of them are found by a short scan from row 0)
        const bool eq = i < n && v == T;
        const unsigned long long me = __ballot(eq);
        if (me != 0ULL) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[2], __popcll(me));
            base = __shfl(base, 0, 64);
            if (eq) {
                const int slot = base + __popcll(me & ((1ULL << lane) - 1ULL));
                if (slot < THR_CAP) eq_id[slot] = (int)i;
            }
        }
    }
}

template <int KS>
__global__ __launch_bounds__(MERGE_THREADS) void select_fill_kernel(const double* __restrict__ score, long long n,
                                                                    const double* __restrict__ thr,
                                                                    const double* __restrict__ buf_key,
                                                                    const int* __restrict__ buf_id,
                                                                    const int* __restrict__ eq_id,
                                                                    int* __restrict__ counters, int k,
                                                                    double* __restrict__ D, long long* __restrict__ I) {
    __shared__ double mk[(MERGE_WAVES - 1) * KS * 64];
    __shared__ int mi[(MERGE_WAVES - 1) * KS * 64];
    __shared__ int wcount[MERGE_WAVES];
    __shared__ int filled;
    const int cnt = counters[0];
    if (cnt > THR_CAP) {  // uniform: the list kernels take over
        if (threadIdx.x == 0) counters[1] = 1;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    WaveTopK<double, int, KS> tk;
    block_merge<double, int, KS, MERGE_WAVES>(
        [&](long long c, double& key, int& id) {
            key = buf_key[c];
            id = buf_id[c];
            return true;
        },
        (long long)cnt, k, tk, mk, mi);
    const int have = cnt < k ? cnt : k;
    if (tid < 64) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int p = s * 64 + lane;
            if (p < have) {
                D[p] = tk.key[s];
                I[p] = (long long)tk.id[s];
            }
        }
    }
    if (have == k) return;
    // complete with the lowest-numbered rows scoring exactly T (at least k - have of them exist)
    const double T = thr[0];
    const int ceq = counters[2];
    if (ceq <= THR_CAP) {  // few such rows: all remembered; the smallest row numbers = the best keys -row
        const int need = k - have;
        __syncthreads();   // the scratch of the first selection is free
        WaveTopK<double, int, KS> te;
        block_merge<double, int, KS, MERGE_WAVES>(
            [&](long long c, double& key, int& id) {
                id = eq_id[c];
                key = -(double)id;
                return true;
            },
            (long long)ceq, need, te, mk, mi);
        if (tid < 64) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s * 64 + lane;
                if (p < need && p < ceq) {
                    D[have + p] = T;
                    I[have + p] = (long long)te.id[s];
                }
            }
        }
        return;
    }
    if (tid == 0) filled = have;
    __syncthreads();
    for (long long base = 0; base < n; base += MERGE_THREADS) {
        const long long i = base + tid;
        const bool eq = i < n && score[i] == T;
        const unsigned long long m = __ballot(eq);
        if (lane == 0) wcount[wid] = __popcll(m);
        __syncthreads();
        int before = filled;
        for (int w = 0; w < wid; ++w) before += wcount[w];
        int total = 0;
        for (int w = 0; w < MERGE_WAVES; ++w) total += wcount[w];
        if (eq) {
            const int pos = before + __popcll(m & ((1ULL << lane) - 1ULL));
            if (pos < k) {
                D[pos] = T;
                I[pos] = i;
            }
        }
        __syncthreads();
        if (tid == 0) filled += total;
        __syncthreads();
        if (filled >= k) break;
    }
}

// rank[i] = #{ j : s_j > s_q(i)  or  (s_j == s_q(i) and j < q(i)) }  for nq queried rows, i.e.
// the position of row q(i) in the stable descending sort of all n scores.
//   rank_prep_kernel   sorts the nq queried (score, row) pairs best-first (rank by counting).
//   rank_hist_kernel   every row j finds b(j) = number of queried pairs ranked ahead of (s_j, j)
//                      by binary search and bumps histogram bin b(j): row j precedes exactly
//                      the queried pairs at sorted positions >= b(j) (except itself).
//   rank_final_kernel  rank of sorted position p = hist[0] + ... + hist[p] - 1 (minus itself).
// One pass over the score vector whatever nq is.
constexpr int RANK_MAXQ = 1024;

__device__ __forceinline__ bool pair_ahead(double ka, long long ia, double kb, long long ib) {
    return (ka > kb) || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(RANK_MAXQ) void rank_prep_kernel(const double* __restrict__ score, int nq,
                                                              const long long* __restrict__ qrow,
                                                              double* __restrict__ skey, long long* __restrict__ srow,
                                                              int* __restrict__ pos_of, unsigned long long* __restrict__ hist) {
    __shared__ double ks[RANK_MAXQ];
    __shared__ long long rs[RANK_MAXQ];
    const int i = threadIdx.x;
    if (i < nq) {
        rs[i] = qrow[i];
        ks[i] = score[qrow[i]];
    }
    if (i < nq) hist[i] = 0;
    if (i == 0) hist[nq] = 0;
    __syncthreads();
    if (i < nq) {
        const double k = ks[i];
        const long long r = rs[i];
        int pos = 0;
        for (int j = 0; j < nq; ++j)  // a row queried twice: the earlier query sorts first
            pos += (pair_ahead(ks[j], rs[j], k, r) || (ks[j] == k && rs[j] == r && j < i)) ? 1 : 0;
        skey[pos] = k;
        srow[pos] = r;
        pos_of[i] = pos;
    }
}

__global__ __launch_bounds__(256) void rank_hist_kernel(const double* __restrict__ score, long long n, int nq,
                                                        const double* __restrict__ skey,
                                                        const long long* __restrict__ srow,
                                                        unsigned long long* __restrict__ hist) {
    __shared__ double ks[RANK_MAXQ];
    __shared__ long long rs[RANK_MAXQ];
    __shared__ unsigned int h[RANK_MAXQ + 1];
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
        ks[i] = skey[i];
        rs[i] = srow[i];
    }
    for (int i = threadIdx.x; i <= nq; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    int run_bin = -1;
    unsigned int run_cnt = 0;  // consecutive rows of a thread mostly fall into one bin
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const double v = score[j];
        int lo = 0, hi = nq;  // first sorted position whose pair is NOT ahead of (v, j)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pair_ahead(ks[mid], rs[mid], v, j)) lo = mid + 1;
            else hi = mid;
        }
        if (lo == run_bin) {
            ++run_cnt;
        } else {
            if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
            run_bin = lo;
            run_cnt = 1;
        }
    }
    if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
    __syncthreads();
    for (int i = threadIdx.x; i <= nq; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ __launch_bounds__(RANK_MAXQ) void rank_final_kernel(const unsigned long long* __restrict__ hist, int nq,
                                                               const int* __restrict__ pos_of,
                                                               unsigned long long* __restrict__ rank) {
    __shared__ unsigned long long pre[RANK_MAXQ];
    const int i = threadIdx.x;
    if (i == 0) {
        unsigned long long run = 0;
        for (int p = 0; p < nq; ++p) {
            run += hist[p];
            pre[p] = run;
        }
    }
    __syncthreads();
    if (i < nq) rank[i] = pre[pos_of[i]] - 1ULL;
}

__global__ void gather_kernel(const double* __restrict__ score, int nq, const long long* __restrict__ qrow,
                              double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) out[i] = score[qrow[i]];
}

}  // namespace

struct rag_csc {
    int64_t n_terms = 0, n_rows = 0, nnz = 0;
    std::vector<int64_t> indptr;  // host copy: launch bounds per term
    int* row = nullptr;
    double* val = nullptr;
    double* score = nullptr;
    bool have_scores = false;
    // bm25
    bool bm25 = false;
    std::vector<double> idf;
    int* doc_len = nullptr;
    double avgdl = 0, k1 = 0, b = 0;
    // top-k workspace
    double* cand_key = nullptr; int* cand_id = nullptr; int64_t cand_cap = 0;
    double* out_D = nullptr; long long* out_I = nullptr; int64_t out_cap = 0;
    double* thr_ws = nullptr;   // [SPANS] span maxima + [1] threshold (threshold path of the top-k)
    int* thr_cnt = nullptr;     // [2] appended scores, overflow flag
    char* pin = nullptr;        // 64 KiB of pinned host memory: results and rank queries of small calls
    double* rk_gather = nullptr;  // rag_tfidf_query: scores of the queried rows
    long long* rk_rows = nullptr; unsigned long long* rk_out = nullptr;
    double* rk_skey = nullptr; long long* rk_srow = nullptr; int* rk_pos = nullptr; unsigned long long* rk_hist = nullptr;
    std::mutex mu;
};

namespace {

int ks_for(int k) { return k <= 64 ? 1 : (k <= 128 ? 2 : 4); }

int term_grid(long long len) {
    long long nb = (len + 255) / 256;
    const long long cap = (long long)num_cus() * 8;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// Dense top-k of h->score into host D/I (k entries; pads when k > n_rows).
int topk_scores(rag_csc* h, int k, double* D, int64_t* I, hipStream_t st) {
    if (k <= 0) return RAG_OK;
    if (k > RAG_MAX_K) return fail(RAG_ERR_INVALID_ARG, "k=%d out of range [0,%d]", k, RAG_MAX_K);
    const int KS = ks_for(k);
    long long nb = (h->n_rows + 4095) / 4096;  // >= 4096 rows per block
    nb = std::max<long long>(1, std::min<long long>(nb, (long long)num_cus() * 4));
    // threshold path for large score vectors (see span_max_kernel); the list kernels stay enqueued
    // behind it and return at once unless its buffer overflowed
    const bool thr_path = h->n_rows >= 65536 && k <= SPANS / 4 && options().sparse_topk_lists == 0;
    if (thr_path && h->thr_ws == nullptr) {
        RAG_HIP(hipMalloc(&h->thr_ws, (size_t)(SPANS + 1) * sizeof(double)));
        RAG_HIP(hipMalloc(&h->thr_cnt, 4 * sizeof(int)));
    }
    if (std::max<long long>(nb * k, thr_path ? 2 * THR_CAP : 0) > h->cand_cap) {
        if (h->cand_key) RAG_HIP(hipFree(h->cand_key));
        if (h->cand_id) RAG_HIP(hipFree(h->cand_id));
        h->cand_key = nullptr; h->cand_id = nullptr; h->cand_cap = 0;
        const int64_t want = std::max<int64_t>(nb * k, 2 * THR_CAP);
        RAG_HIP(hipMalloc(&h->cand_key, (size_t)want * sizeof(double)));
        RAG_HIP(hipMalloc(&h->cand_id, (size_t)want * sizeof(int)));
        h->cand_cap = want;
    }
    if (k > h->out_cap) {
        if (h->out_D) RAG_HIP(hipFree(h->out_D));
        if (h->out_I) RAG_HIP(hipFree(h->out_I));
        h->out_D = nullptr; h->out_I = nullptr; h->out_cap = 0;
        RAG_HIP(hipMalloc(&h->out_D, (size_t)RAG_MAX_K * sizeof(double)));
        RAG_HIP(hipMalloc(&h->out_I, (size_t)RAG_MAX_K * sizeof(long long)));
        h->out_cap = RAG_MAX_K;
    }
    const int* flag = nullptr;
    if (thr_path) {
        double* wmax = h->thr_ws;
        double* thr = h->thr_ws + SPANS;
        flag = h->thr_cnt + 1;
        span_max_kernel<<<SPANS / 4, 256, 0, st>>>(h->score, (long long)h->n_rows, wmax);
        kth_span_max_kernel<<<1, SPANS, 0, st>>>(wmax, k, thr, h->thr_cnt);
        const long long cb = std::min<long long>((h->n_rows + 255) / 256, (long long)num_cus() * 8);
        int* eq_id = h->cand_id + THR_CAP;
        compact_gt_kernel<<<(unsigned)cb, 256, 0, st>>>(h->score, (long long)h->n_rows, thr, h->cand_key, h->cand_id,
                                                        eq_id, h->thr_cnt);
        if (KS == 1)
            select_fill_kernel<1><<<1, MERGE_THREADS, 0, st>>>(h->score, (long long)h->n_rows, thr, h->cand_key,
                                                               h->cand_id, eq_id, h->thr_cnt, k, h->out_D, h->out_I);
        else if (KS == 2)
            select_fill_kernel<2><<<1, MERGE_THREADS, 0, st>>>(h->score, (long long)h->n_rows, thr, h->cand_key,
                                                               h->cand_id, eq_id, h->thr_cnt, k, h->out_D, h->out_I);
        else
            select_fill_kernel<4><<<1, MERGE_THREADS, 0, st>>>(h->score, (long long)h->n_rows, thr, h->cand_key,
                                                               h->cand_id, eq_id, h->thr_cnt, k, h->out_D, h->out_I);
        RAG_HIP(hipGetLastError());
    }
#define LAUNCH_TOPK(KSV)                                                                                   \
    do {                                                                                                   \
        dense_topk_kernel<KSV><<<(unsigned)nb, TK_THREADS, 0, st>>>(h->score, (long long)h->n_rows, k,     \
                                                                    h->cand_key, h->cand_id, flag);       \
        dense_topk_merge_kernel<KSV><<<1, MERGE_THREADS, 0, st>>>(h->cand_key, h->cand_id, (int)nb, k,     \
                                                                  h->out_D, h->out_I, flag);               \
    } while (0)
    if (KS == 1) LAUNCH_TOPK(1);
    else if (KS == 2) LAUNCH_TOPK(2);
    else LAUNCH_TOPK(4);
#undef LAUNCH_TOPK
    RAG_HIP(hipGetLastError());
    // into pinned memory (true asynchronous copies); topk_finish() hands them over after the caller's
    // synchronisation
    if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
    RAG_HIP(hipMemcpyAsync(h->pin, h->out_D, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, st));
    RAG_HIP(hipMemcpyAsync(h->pin + PIN_BYTES / 2, h->out_I, (size_t)k * sizeof(long long), hipMemcpyDeviceToHost, st));
    return RAG_OK;
}

void topk_finish(rag_csc* h, int k, double* D, int64_t* I) {
    if (k <= 0) return;
    std::memcpy(D, h->pin, (size_t)k * sizeof(double));
    std::memcpy(I, h->pin + PIN_BYTES / 2, (size_t)k * sizeof(long long));
}

}  // namespace

extern "C" {

int rag_csc_create(int64_t n_terms, int64_t n_rows, const int64_t* indptr, const int32_t* row, const double* val,
                   rag_csc_t** out) {
    RAG_REQUIRE_INIT();
    if (!out) return fail(RAG_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n_terms < 0 || n_rows < 0 || !indptr) return fail(RAG_ERR_INVALID_ARG, "bad sizes or indptr");
    if (n_rows > (int64_t)INT_MAX - 8) return fail(RAG_ERR_UNSUPPORTED, "more than 2^31-9 rows");
    if (indptr[0] != 0) return fail(RAG_ERR_INVALID_ARG, "indptr[0] must be 0");
    for (int64_t t = 0; t < n_terms; ++t)
        if (indptr[t + 1] < indptr[t]) return fail(RAG_ERR_INVALID_ARG, "indptr not monotone at term %lld", (long long)t);
    const int64_t nnz = indptr[n_terms];
    if (nnz > 0 && (!row || !val)) return fail(RAG_ERR_INVALID_ARG, "row/val NULL with nnz > 0");
    for (int64_t p = 0; p < nnz; ++p)
        if (row[p] < 0 || row[p] >= n_rows)
            return fail(RAG_ERR_INVALID_ARG, "posting %lld has row %d outside [0,%lld)", (long long)p, row[p],
                        (long long)n_rows);
    rag_csc* h = new rag_csc();
    h->n_terms = n_terms;
    h->n_rows = n_rows;
    h->nnz = nnz;
    h->indptr.assign(indptr, indptr + n_terms + 1);
    hipError_t e = hipSuccess;
    if (nnz > 0) {
        e = hipMalloc(&h->row, (size_t)nnz * sizeof(int));
        if (e == hipSuccess) e = hipMalloc(&h->val, (size_t)nnz * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(h->row, row, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&h->score, (size_t)std::max<int64_t>(n_rows, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->rk_rows, RANK_MAXQ * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&h->rk_out, RANK_MAXQ * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&h->rk_skey, RANK_MAXQ * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->rk_srow, RANK_MAXQ * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&h->rk_pos, RANK_MAXQ * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->rk_hist, (RANK_MAXQ + 1) * sizeof(unsigned long long));
    if (e != hipSuccess) {
        rag_csc_free(h);
        return fail(e == hipErrorOutOfMemory ? RAG_ERR_OOM : RAG_ERR_HIP, "rag_csc_create: %s", hipGetErrorString(e));
    }
    *out = h;
    return RAG_OK;
}

void rag_csc_free(rag_csc_t* h) {
    if (!h) return;
    if (lib_stream()) (void)hipStreamSynchronize(lib_stream());
    (void)hipFree(h->row);
    (void)hipFree(h->val);
    (void)hipFree(h->score);
    (void)hipFree(h->doc_len);
    (void)hipFree(h->cand_key);
    (void)hipFree(h->cand_id);
    (void)hipFree(h->out_D);
    (void)hipFree(h->out_I);
    (void)hipFree(h->thr_ws);
    (void)hipFree(h->thr_cnt);
    if (h->pin) (void)hipHostFree(h->pin);
    (void)hipFree(h->rk_rows);
    (void)hipFree(h->rk_out);
    (void)hipFree(h->rk_skey);
    (void)hipFree(h->rk_srow);
    (void)hipFree(h->rk_pos);
    (void)hipFree(h->rk_hist);
    (void)hipFree(h->rk_gather);
    delete h;
}

int rag_bm25_set(rag_csc_t* h, const double* idf, const int32_t* doc_len, double avgdl, double k1, double b) {
    RAG_REQUIRE_INIT();
    if (!h || (h->n_terms > 0 && !idf) || (h->n_rows > 0 && !doc_len)) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    h->idf.assign(idf, idf + h->n_terms);
    if (!h->doc_len) RAG_HIP(hipMalloc(&h->doc_len, (size_t)std::max<int64_t>(h->n_rows, 1) * sizeof(int)));
    if (h->n_rows > 0)
        RAG_HIP(hipMemcpy(h->doc_len, doc_len, (size_t)h->n_rows * sizeof(int), hipMemcpyHostToDevice));
    h->avgdl = avgdl;
    h->k1 = k1;
    h->b = b;
    h->bm25 = true;
    return RAG_OK;
}

int rag_bm25_scores(rag_csc_t* h, int n_qterms, const int32_t* term_ids, double* scores_or_null, int k, double* D,
                    int64_t* I) {
    RAG_REQUIRE_INIT();
    if (!h || n_qterms < 0 || (n_qterms > 0 && !term_ids) || k < 0 || (k > 0 && (!D || !I)))
        return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->bm25) return fail(RAG_ERR_INVALID_ARG, "rag_bm25_set has not been called on this handle");
    for (int i = 0; i < n_qterms; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    hipStream_t st = lib_stream();
    if (h->n_rows > 0) RAG_HIP(hipMemsetAsync(h->score, 0, (size_t)h->n_rows * sizeof(double), st));
    // Python evaluates (self.k1 + 1) and (1 - self.b) once per term as float64 scalars.
    const double k1p1 = h->k1 + 1.0, omb = 1.0 - h->b;
    for (int i = 0; i < n_qterms; ++i) {
        const int t = term_ids[i];
        const long long b0 = h->indptr[t], b1 = h->indptr[t + 1];
        if (b1 <= b0) continue;
        bm25_term_kernel<<<term_grid(b1 - b0), 256, 0, st>>>(h->row, h->val, b0, b1, h->idf[t], h->doc_len, h->avgdl,
                                                            h->k1, h->b, k1p1, omb, h->score);
    }
    RAG_HIP(hipGetLastError());
    h->have_scores = true;
    if (scores_or_null && h->n_rows > 0)
        RAG_HIP(hipMemcpyAsync(scores_or_null, h->score, (size_t)h->n_rows * sizeof(double), hipMemcpyDeviceToHost, st));
    int rc = topk_scores(h, k, D, I, st);
    if (rc) return rc;
    RAG_HIP(hipStreamSynchronize(st));
    topk_finish(h, k, D, I);
    return RAG_OK;
}

int rag_tfidf_scores(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights,
                     double* scores_or_null, int k, double* D, int64_t* I) {
    RAG_REQUIRE_INIT();
    if (!h || nnz_q < 0 || (nnz_q > 0 && (!term_ids || !q_weights)) || k < 0 || (k > 0 && (!D || !I)))
        return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < nnz_q; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    hipStream_t st = lib_stream();
    if (h->n_rows > 0) RAG_HIP(hipMemsetAsync(h->score, 0, (size_t)h->n_rows * sizeof(double), st));
    for (int i = 0; i < nnz_q; ++i) {
        const int t = term_ids[i];
        const long long b0 = h->indptr[t], b1 = h->indptr[t + 1];
        if (b1 <= b0) continue;
        tfidf_term_kernel<<<term_grid(b1 - b0), 256, 0, st>>>(h->row, h->val, b0, b1, q_weights[i], h->score);
    }
    RAG_HIP(hipGetLastError());
    h->have_scores = true;
    if (scores_or_null && h->n_rows > 0)
        RAG_HIP(hipMemcpyAsync(scores_or_null, h->score, (size_t)h->n_rows * sizeof(double), hipMemcpyDeviceToHost, st));
    int rc = topk_scores(h, k, D, I, st);
    if (rc) return rc;
    RAG_HIP(hipStreamSynchronize(st));
    topk_finish(h, k, D, I);
    return RAG_OK;
}

int rag_tfidf_query(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights, int k, double* D,
                    int64_t* I, int n_query, const int64_t* rows, int64_t* ranks, double* row_scores) {
    RAG_REQUIRE_INIT();
    if (!h || nnz_q < 0 || (nnz_q > 0 && (!term_ids || !q_weights)) || k < 0 || (k > 0 && (!D || !I)) || n_query < 0 ||
        (n_query > 0 && !rows))
        return fail(RAG_ERR_INVALID_ARG, "bad argument");
    if (n_query > RANK_MAXQ)
        return fail(RAG_ERR_INVALID_ARG, "rag_tfidf_query: n_query=%d beyond %d (use rag_csc_ranks / rag_csc_gather)",
                    n_query, RANK_MAXQ);
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < nnz_q; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    for (int i = 0; i < n_query; ++i)
        if (rows[i] < 0 || rows[i] >= h->n_rows)
            return fail(RAG_ERR_INVALID_ARG, "row %lld outside [0,%lld)", (long long)rows[i], (long long)h->n_rows);
    hipStream_t st = lib_stream();
    if (h->n_rows > 0) RAG_HIP(hipMemsetAsync(h->score, 0, (size_t)h->n_rows * sizeof(double), st));
    for (int i = 0; i < nnz_q; ++i) {
        const int t = term_ids[i];
        const long long b0 = h->indptr[t], b1 = h->indptr[t + 1];
        if (b1 <= b0) continue;
        tfidf_term_kernel<<<term_grid(b1 - b0), 256, 0, st>>>(h->row, h->val, b0, b1, q_weights[i], h->score);
    }
    RAG_HIP(hipGetLastError());
    h->have_scores = true;
    int rc = topk_scores(h, k, D, I, st);
    if (rc) return rc;
    const bool want_ranks = n_query > 0 && ranks != nullptr, want_scores = n_query > 0 && row_scores != nullptr;
    if (want_ranks || want_scores) {
        if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
        std::memcpy(h->pin + PIN_Q_ROWS, rows, (size_t)n_query * sizeof(long long));
        RAG_HIP(hipMemcpyAsync(h->rk_rows, h->pin + PIN_Q_ROWS, (size_t)n_query * sizeof(long long), hipMemcpyHostToDevice, st));
    }
    if (want_ranks) {
        long long nb = (h->n_rows + 2047) / 2048;
        nb = std::max<long long>(1, std::min<long long>(nb, (long long)num_cus() * 4));
        rank_prep_kernel<<<1, RANK_MAXQ, 0, st>>>(h->score, n_query, h->rk_rows, h->rk_skey, h->rk_srow, h->rk_pos, h->rk_hist);
        rank_hist_kernel<<<(unsigned)nb, 256, 0, st>>>(h->score, (long long)h->n_rows, n_query, h->rk_skey, h->rk_srow,
                                                      h->rk_hist);
        rank_final_kernel<<<1, RANK_MAXQ, 0, st>>>(h->rk_hist, n_query, h->rk_pos, h->rk_out);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipMemcpyAsync(h->pin + PIN_Q_RANKS, h->rk_out, (size_t)n_query * sizeof(long long), hipMemcpyDeviceToHost, st));
    }
    if (want_scores) {
        if (!h->rk_gather) RAG_HIP(hipMalloc(&h->rk_gather, RANK_MAXQ * sizeof(double)));
        gather_kernel<<<(n_query + 255) / 256, 256, 0, st>>>(h->score, n_query, h->rk_rows, h->rk_gather);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipMemcpyAsync(h->pin + PIN_Q_SCORES, h->rk_gather, (size_t)n_query * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    RAG_HIP(hipStreamSynchronize(st));
    topk_finish(h, k, D, I);
    if (want_ranks) std::memcpy(ranks, h->pin + PIN_Q_RANKS, (size_t)n_query * sizeof(long long));
    if (want_scores) std::memcpy(row_scores, h->pin + PIN_Q_SCORES, (size_t)n_query * sizeof(double));
    return RAG_OK;
}

int rag_csc_ranks(rag_csc_t* h, int n_query, const int64_t* rows, int64_t* ranks) {
    RAG_REQUIRE_INIT();
    if (!h || n_query < 0 || (n_query > 0 && (!rows || !ranks))) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->have_scores) return fail(RAG_ERR_INVALID_ARG, "no score vector has been computed on this handle");
    for (int i = 0; i < n_query; ++i)
        if (rows[i] < 0 || rows[i] >= h->n_rows)
            return fail(RAG_ERR_INVALID_ARG, "row %lld outside [0,%lld)", (long long)rows[i], (long long)h->n_rows);
    hipStream_t st = lib_stream();
    for (int q0 = 0; q0 < n_query; q0 += RANK_MAXQ) {
        const int nq = std::min(RANK_MAXQ, n_query - q0);
        if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
        std::memcpy(h->pin, rows + q0, (size_t)nq * sizeof(long long));
        RAG_HIP(hipMemcpyAsync(h->rk_rows, h->pin, (size_t)nq * sizeof(long long), hipMemcpyHostToDevice, st));
        long long nb = (h->n_rows + 2047) / 2048;
        nb = std::max<long long>(1, std::min<long long>(nb, (long long)num_cus() * 4));
        rank_prep_kernel<<<1, RANK_MAXQ, 0, st>>>(h->score, nq, h->rk_rows, h->rk_skey, h->rk_srow, h->rk_pos, h->rk_hist);
        rank_hist_kernel<<<(unsigned)nb, 256, 0, st>>>(h->score, (long long)h->n_rows, nq, h->rk_skey, h->rk_srow,
                                                      h->rk_hist);
        rank_final_kernel<<<1, RANK_MAXQ, 0, st>>>(h->rk_hist, nq, h->rk_pos, h->rk_out);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipMemcpyAsync(h->pin + PIN_BYTES / 2, h->rk_out, (size_t)nq * sizeof(long long), hipMemcpyDeviceToHost, st));
        RAG_HIP(hipStreamSynchronize(st));
        std::memcpy(ranks + q0, h->pin + PIN_BYTES / 2, (size_t)nq * sizeof(long long));
    }
    return RAG_OK;
}

int rag_csc_gather(rag_csc_t* h, int n_query, const int64_t* rows, double* out_scores) {
    RAG_REQUIRE_INIT();
    if (!h || n_query < 0 || (n_query > 0 && (!rows || !out_scores))) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->have_scores) return fail(RAG_ERR_INVALID_ARG, "no score vector has been computed on this handle");
    for (int i = 0; i < n_query; ++i)
        if (rows[i] < 0 || rows[i] >= h->n_rows)
            return fail(RAG_ERR_INVALID_ARG, "row %lld outside [0,%lld)", (long long)rows[i], (long long)h->n_rows);
    hipStream_t st = lib_stream();
    for (int q0 = 0; q0 < n_query; q0 += RANK_MAXQ) {
        const int nq = std::min(RANK_MAXQ, n_query - q0);
        double* dout = reinterpret_cast<double*>(h->rk_out);  // same size: RANK_MAXQ * 8 bytes
        if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
        std::memcpy(h->pin, rows + q0, (size_t)nq * sizeof(long long));
        RAG_HIP(hipMemcpyAsync(h->rk_rows, h->pin, (size_t)nq * sizeof(long long), hipMemcpyHostToDevice, st));
        gather_kernel<<<(nq + 255) / 256, 256, 0, st>>>(h->score, nq, h->rk_rows, dout);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipMemcpyAsync(out_scores + q0, dout, (size_t)nq * sizeof(double), hipMemcpyDeviceToHost, st));
        RAG_HIP(hipStreamSynchronize(st));
    }
    return RAG_OK;
}

}  // extern "C"
