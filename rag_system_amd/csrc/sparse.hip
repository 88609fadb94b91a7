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
constexpr size_t PIN_FLAG = 60 << 10;  // fused selection: 1 = its buffer overflowed, rerun with the list kernels
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
                                                              int* __restrict__ pos_of, unsigned long long* __restrict__ hist,
                                                              double* __restrict__ gather_out = nullptr,
                                                              const double* __restrict__ pair_key = nullptr,
                                                              long long n = 0) {
    __shared__ double ks[RANK_MAXQ];
    __shared__ long long rs[RANK_MAXQ];
    const int i = threadIdx.x;
    bool foreign = false;
    if (i < nq) {
        rs[i] = qrow[i];
        if (pair_key != nullptr) {
            // rag_csc_count_ahead: explicit (score, id) pairs, ids in this vector's row numbering but possibly
            // outside [0, n) -- another shard's rows.  Only a pair that IS one of this vector's rows counts itself
            // in the histogram below and has 1 subtracted by rank_final_kernel.
            ks[i] = pair_key[i];
            foreign = !(rs[i] >= 0 && rs[i] < n && score[rs[i]] == ks[i]);
        } else {
            ks[i] = score[qrow[i]];
        }
        if (gather_out != nullptr) gather_out[i] = ks[i];  // rag_tfidf_query_end: the queried rows' scores (pinned memory)
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
        pos_of[i] = pos | (foreign ? (1 << 30) : 0);
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
    if (i < nq) {
        const int p = pos_of[i];
        rank[i] = pre[p & ((1 << 30) - 1)] - ((p >> 30) & 1 ? 0ULL : 1ULL);
    }
}

// The three kernels above in ONE launch, for up to RANK1_MAXQ queried rows (rag_tfidf_query_end: a hybrid query
// asks for the ranks of its 30-60 fused candidates): every workgroup sorts the queried pairs itself (LDS, rank by
// counting), bins its share of the rows, adds its bins to `hist` with agent-scope atomics and draws a ticket;
// the workgroup that arrives last sums the prefix, writes ranks (and block 0 the gathered scores) -- to pinned
// host memory -- and puts `hist` and the ticket back to zero for the next call (both start at zero: hipMemset
// when they are allocated).
constexpr int RANK1_MAXQ = 256;
constexpr int RANK1_T = 256;
constexpr int RANK1_RB = 8;  // rows per thread and chunk
constexpr int RANK1_HREP = 16;  // copies of the histogram (workgroup b adds to copy b % 16): atomics on one line serialise
struct Rank1Rows {
    int r[RANK1_MAXQ];  // the queried rows travel as a kernel argument (1 KiB): 256-512 workgroups reading them from
};                      // pinned host memory took ~10 us of the kernel's 23
__global__ __launch_bounds__(RANK1_T) void rank_one_kernel(const double* __restrict__ score, long long n, int nq,
                                                           const Rank1Rows qrow,
                                                           unsigned long long* __restrict__ hist, int* __restrict__ ticket,
                                                           unsigned long long* __restrict__ out_rank,
                                                           double* __restrict__ out_score) {
    __shared__ double uk[RANK1_MAXQ], ks[RANK1_MAXQ];
    __shared__ long long ur[RANK1_MAXQ], rs[RANK1_MAXQ];
    __shared__ int po[RANK1_MAXQ];
    __shared__ unsigned int hh[RANK1_MAXQ + 1];
    __shared__ unsigned long long pre[RANK1_MAXQ];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    if (tid < nq) {
        const long long r = qrow.r[tid];
        const double v = score[r];
        ur[tid] = r;
        uk[tid] = v;
        if (out_score != nullptr && blockIdx.x == 0) out_score[tid] = v;
    }
    for (int i = tid; i <= nq; i += RANK1_T) hh[i] = 0;
    __syncthreads();
    if (tid < nq) {
        const double k = uk[tid];
        const long long r = ur[tid];
        int pos = 0;
        for (int j = 0; j < nq; ++j)  // a row queried twice: the earlier query sorts first
            pos += (pair_ahead(uk[j], ur[j], k, r) || (uk[j] == k && ur[j] == r && j < tid)) ? 1 : 0;
        po[tid] = pos;
        ks[pos] = k;
        rs[pos] = r;
    }
    __syncthreads();
    // chunks of RANK1_RB * 256 consecutive rows; a thread's RANK1_RB searches run side by side (independent LDS
    // reads per step: one search after the other was 28 us for 1M rows with one wave per SIMD)
    const int lane = tid & 63;
    const long long nchunks = (n + (long long)RANK1_T * RANK1_RB - 1) / ((long long)RANK1_T * RANK1_RB);
    for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const long long j0 = ch * RANK1_T * RANK1_RB + tid;
        double v[RANK1_RB];
        int lo[RANK1_RB], hi[RANK1_RB];
#pragma unroll
        for (int u = 0; u < RANK1_RB; ++u) {
            const long long j = j0 + (long long)u * RANK1_T;
            v[u] = j < n ? score[j] : 0.0;
            lo[u] = 0;
            hi[u] = j < n ? nq : 0;
        }
        for (int span = nq; span > 0; span >>= 1) {
#pragma unroll
            for (int u = 0; u < RANK1_RB; ++u) {
                if (lo[u] < hi[u]) {  // first sorted position whose pair is NOT ahead of (v, j)
                    const int mid = (lo[u] + hi[u]) >> 1;
                    if (pair_ahead(ks[mid], rs[mid], v[u], j0 + (long long)u * RANK1_T)) lo[u] = mid + 1;
                    else hi[u] = mid;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RANK1_RB; ++u) {
            const int bin = j0 + (long long)u * RANK1_T < n ? lo[u] : -1;
            const int b0 = __shfl(bin, 0, 64);  // most waves fall into one bin whole (the rows behind every queried pair)
            if (__all(bin == b0)) {
                if (lane == 0 && b0 >= 0) atomicAdd(&hh[b0], 64u);
            } else if (bin >= 0) {
                atomicAdd(&hh[bin], 1u);
            }
        }
    }
    __syncthreads();
    // (the last bin -- rows behind every queried pair -- is never read)
    if (tid < nq && hh[tid])
        __hip_atomic_fetch_add(&hist[(blockIdx.x % RANK1_HREP) * RANK1_MAXQ + tid], (unsigned long long)hh[tid],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (int)gridDim.x - 1;
        if (last) {
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < nq) {
        unsigned long long a = 0;
#pragma unroll
        for (int r = 0; r < RANK1_HREP; ++r) {
            a += __hip_atomic_load(&hist[r * RANK1_MAXQ + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&hist[r * RANK1_MAXQ + tid], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pre[tid] = a;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int p = 0; p < nq; ++p) {
            run += pre[p];
            pre[p] = run;
        }
    }
    __syncthreads();
    if (tid < nq) out_rank[tid] = pre[po[tid]] - 1ULL;
}

__global__ void gather_kernel(const double* __restrict__ score, int nq, const long long* __restrict__ qrow,
                              double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) out[i] = score[qrow[i]];
}


// ================================================================ fused path (one launch per scoring)
// The per-term launches above are launch-bound (a query of 3-8 terms is 5 us per term for ~1 us of
// traffic each).  Fused path: the rows are cut into blocks of SP_R, one workgroup owns one block, keeps
// its SP_R scores in LDS and walks the query terms IN ORDER over its slice of each term's postings
// (postings ascend by row within a term; the slice bounds of frequent terms come from a skip table built
// with the index, rare terms are scanned whole by every workgroup).  A row's additions therefore happen
// in query-term order with exactly the float64 operations of the per-term kernels: bit-identical scores,
// no atomics, no read-modify-write traffic to HBM -- 12-16 bytes per posting instead of 28-32.
// The selection kernel behind it fuses what used to be six launches (span maxima, threshold, compaction,
// exact selection, rank histogram, rank prefix): every workgroup derives the threshold from the span
// maxima, compacts its own rows and bins them for the rank query; the workgroup that finishes last
// (one ticket, agent-scope release / acquire) selects, completes ties and writes the results -- straight
// into pinned host memory, so a call is two launches and one synchronisation.
constexpr int SP_R = 1024;            // rows owned by one workgroup
constexpr int SP_T = 256;             // threads per workgroup
constexpr int SP_SUB = SP_R / 64;     // 64-row sub-spans per workgroup; their maxima feed the threshold
constexpr int SP_MAXT = 64;           // query terms per launch (longer queries: several launches, scores carried in HBM)
constexpr long long SP_FREQ_MIN = 1024;  // terms with at least this many postings get a row of the skip table
// counters: [0] scores appended, [1] unused, [2] rows scoring exactly T, [3] second-level ticket of the selection
// kernel; first-level tickets on cache lines of their own: one word takes ~88 returning atomics per
// microsecond, a thousand workgroups on one word were 11 us.  The scoring kernel's own tickets follow.
constexpr int SP_TGRP = 32;
constexpr int SP_TICKET0 = 32;                               // selection kernel: SP_TGRP words, stride 32 ints
constexpr int SP_TICKET1 = SP_TICKET0 + SP_TGRP * 32;         // scoring kernel: SP_TGRP + 1 words, stride 32 ints
constexpr int SP_NCOUNTERS = SP_TICKET1 + (SP_TGRP + 1) * 32;
constexpr int SP_ZERO_PER_CALL = SP_TICKET1;                  // what the scoring kernel clears for the selection kernel
constexpr int SP_HREP = 32;                                   // replicas of the rank histogram (workgroup % SP_HREP)

struct SpTerms {
    int n;
    int freq[SP_MAXT];          // row of the skip table, or -1: rare term
    long long begin[SP_MAXT];   // postings [begin, end) of the term
    long long end[SP_MAXT];
    double w[SP_MAXT];          // BM25: idf of the term; TF-IDF: the query weight
};

template <bool BM25>
__global__ __launch_bounds__(SP_T) void sparse_score_kernel(
    const SpTerms qt, const int* __restrict__ row, const double* __restrict__ val, const int* __restrict__ skip,
    long long skip_ld, const int* __restrict__ doc_len, double avgdl, double k1, double b, double k1p1, double omb,
    long long n_rows, double* __restrict__ score, double* __restrict__ wmax, double* __restrict__ bmax, int accumulate,
    int* __restrict__ counters, int n_counters, unsigned long long* __restrict__ hist, int n_hist,
    const long long* __restrict__ qrow_host, long long* __restrict__ qrow_dev, int n_qrow, int thr_G, int thr_k,
    int thr_per_block, double* __restrict__ thr_out) {
    __shared__ double s[SP_R];
    __shared__ int flag[2];
    __shared__ double t4s[BM25 ? SP_R : 1];
    __shared__ long long rng[SP_MAXT][2];
    __shared__ double sub[SP_SUB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long lo = (long long)blockIdx.x * SP_R;
    const int cnt = (int)(n_rows - lo < SP_R ? n_rows - lo : SP_R);
    if (blockIdx.x == 0) {  // per-call set-up of what the selection kernel consumes (it runs behind this launch)
        for (int i = tid; i < n_counters; i += SP_T) counters[i] = 0;
        for (int i = tid; i < n_hist; i += SP_T) hist[i] = 0ULL;
        for (int i = tid; i < n_qrow; i += SP_T) qrow_dev[i] = qrow_host[i];
    }
    for (int r = tid; r < SP_R; r += SP_T) {
        s[r] = (accumulate && r < cnt) ? score[lo + r] : 0.0;
        if (BM25) {
            double t4 = 0.0;
            if (r < cnt) {  // the row's part of the denominator, the same operations as bm25_term_kernel
                const double t1 = b * (double)doc_len[lo + r];
                const double t2 = t1 / avgdl;
                const double t3 = omb + t2;
                t4 = k1 * t3;
            }
            t4s[r] = t4;
        }
    }
    if (tid < qt.n) {
        const int f = qt.freq[tid];
        long long pb = qt.begin[tid], pe = qt.end[tid];
        if (f >= 0) {
            const int* sk = skip + (long long)f * skip_ld + blockIdx.x;
            pe = pb + sk[1];
            pb = pb + sk[0];
        }
        rng[tid][0] = pb;
        rng[tid][1] = pe;
    }
    __syncthreads();
    for (int i = 0; i < qt.n; ++i) {
        const long long pb = rng[i][0], pe = rng[i][1];
        const double w = qt.w[i];
        const bool sliced = qt.freq[i] >= 0;
        for (long long p = pb + tid; p < pe; p += SP_T) {
            const int r = row[p] - (int)lo;
            if (sliced || (r >= 0 && r < cnt)) {
                const double v = val[p];
                if (BM25) {
                    const double num = v * k1p1;
                    const double den = v + t4s[r];
                    const double frac = num / den;
                    const double c = w * frac;
                    s[r] = s[r] + c;
                } else {
                    const double prod = w * v;
                    s[r] = s[r] + prod;
                }
            }
        }
        __syncthreads();  // a row's additions happen in query-term order
    }
#pragma unroll
    for (int j = 0; j < SP_R / SP_T; ++j) {
        const int r = j * SP_T + tid;
        const double v = s[r];
        if (r < cnt) score[lo + r] = v;
        double m = r < cnt ? v : -DBL_MAX;
        if (!(m > -DBL_MAX)) m = -DBL_MAX;  // NaN never wins
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double o = __shfl_xor(m, d, 64);
            m = o > m ? o : m;
        }
        if (lane == 0) {  // write-through (sc1): read by the last workgroup of this launch, see below
            __hip_atomic_store(&wmax[(long long)blockIdx.x * SP_SUB + j * (SP_T / 64) + wid], m, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            sub[j * (SP_T / 64) + wid] = m;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double m = -DBL_MAX;
        for (int i = 0; i < SP_SUB; ++i) m = sub[i] > m ? sub[i] : m;
        __hip_atomic_store(&bmax[blockIdx.x], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (thr_G <= 0) return;  // no selection behind this launch (or not its last launch)

    // ---- threshold of the selection kernel, computed ONCE by the workgroup that finishes last (every
    // workgroup of the selection kernel doing it for itself cost 10-58 us): T = the thr_k-th largest of
    // thr_G group maxima.  k groups hold a score >= T each, so at least k rows score >= T; for scores in no
    // particular order few more do.  Hand-off as in sparse_select_kernel (sc1 stores, drained, one ticket).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int nb = (int)gridDim.x, g = (int)blockIdx.x % SP_TGRP;
        const int gsize = (nb - g + SP_TGRP - 1) / SP_TGRP;
        int last = 0;
        const int t = __hip_atomic_fetch_add(&counters[SP_TICKET1 + g * 32], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gsize - 1) {
            const int groups = nb < SP_TGRP ? nb : SP_TGRP;
            const int t2 = __hip_atomic_fetch_add(&counters[SP_TICKET1 + SP_TGRP * 32], 1, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
            last = (t2 == groups - 1) ? 1 : 0;
        }
        flag[0] = last;
    }
    __syncthreads();
    if (flag[0] == 0) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // everyone has arrived: the ticket words go back to zero for the next launch
    if (tid <= SP_TGRP) __hip_atomic_store(&counters[SP_TICKET1 + tid * 32], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    double* gv = s;  // thr_G <= SP_R doubles
    const double* gmax = thr_per_block ? bmax : wmax;
    const long long n_g = thr_per_block ? (long long)gridDim.x : (long long)gridDim.x * SP_SUB;
    const long long gs = (n_g + thr_G - 1) / thr_G;
    for (int t = tid; t < thr_G; t += SP_T) {
        double m = -DBL_MAX;
        const long long s0 = (long long)t * gs;
        const long long s1 = s0 + gs < n_g ? s0 + gs : n_g;
        for (long long sp = s0; sp < s1; ++sp) {
            const double v = gmax[sp];
            m = v > m ? v : m;
        }
        gv[t] = m;
    }
    __syncthreads();
    // rank by counting (broadcast LDS reads, eight in flight; no barrier per step)
    for (int t = tid; t < thr_G; t += SP_T) {
        const double v = gv[t];
        int ahead = 0;
        for (int j = 0; j < thr_G; j += 8) {
            double o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = gv[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) ahead += (o[u] > v || (o[u] == v && j + u < t)) ? 1 : 0;
        }
        if (ahead == thr_k - 1) thr_out[0] = v;
    }
}

// Dynamic LDS of sparse_select_kernel (bytes): [gv: G doubles][uk, ks: nq doubles each][ur, rs: nq int64 each]
// [po: nq ints][hh: nq+1 u32][hl: nq+1 u64][sk: SP_SORT doubles][si: SP_SORT ints][ahead_of: 512 ints][mk: 3*KS*64 doubles]
// [mi: 3*KS*64 ints][misc: 16 ints]
constexpr int SP_SORT = 1024;  // candidates the last workgroup sorts in LDS (more: list merge)
constexpr int SP_COUNT = 384;
  // ... and up to this many it ranks by counting instead
__host__ __device__ inline size_t sp_select_smem(int G, int nq, int KS) {
    size_t o = (size_t)G * 8 + (size_t)nq * 32;
    o += (((size_t)nq * 4 + 7) & ~(size_t)7) + ((((size_t)nq + 1) * 4 + 7) & ~(size_t)7) + ((size_t)nq + 1) * 8;
    o += (size_t)SP_SORT * 12 + 512 * 4;
    o += (size_t)3 * KS * 64 * 12 + 64;
    return o;
}

// Best-first (key descending, id ascending) bitonic sort of M = 2^x (key, id) pairs in LDS, M <= SP_SORT.
__device__ __forceinline__ void lds_sort_pairs(double* sk, int* si, int M, int tid) {
    for (int size = 2; size <= M; size <<= 1) {
        for (int st = size >> 1; st > 0; st >>= 1) {
            for (int p = tid; p < M / 2; p += SP_T) {
                const int i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
                const int j = i + st;
                const bool best_first = (i & size) == 0;
                const double ka = sk[i], kb = sk[j];
                const int ia = si[i], ib = si[j];
                const bool b_ahead = (kb > ka) || (kb == ka && ib < ia);
                const bool a_ahead = (ka > kb) || (ka == kb && ia < ib);
                if (best_first ? b_ahead : a_ahead) {
                    sk[i] = kb; sk[j] = ka;
                    si[i] = ib; si[j] = ia;
                }
            }
            __syncthreads();
        }
    }
}

// thr[0]: the threshold left by the scoring launch.  G = 0 (no LDS for group maxima here).
template <int KS>
__global__ __launch_bounds__(SP_T) void sparse_select_kernel(
    const double* __restrict__ score, long long n_rows, const double* __restrict__ thr, int G, int k,
    double* __restrict__ buf_key, int* __restrict__ buf_id, int* __restrict__ eq_id, int* __restrict__ counters,
    int nq, const long long* __restrict__ qrow, unsigned long long* __restrict__ hist, double* __restrict__ out_D,
    long long* __restrict__ out_I, long long* __restrict__ out_rank, double* __restrict__ out_rowscore,
    int* __restrict__ out_flag) {
    extern __shared__ __align__(16) unsigned char sm[];
    double* gv = reinterpret_cast<double*>(sm);
    double* uk = gv + G;
    double* ks = uk + nq;
    long long* ur = reinterpret_cast<long long*>(ks + nq);
    long long* rs = ur + nq;
    int* po = reinterpret_cast<int*>(rs + nq);
    unsigned int* hh = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(po) + (((size_t)nq * 4 + 7) & ~(size_t)7));
    unsigned long long* hl = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(hh) +
                                                                   ((((size_t)nq + 1) * 4 + 7) & ~(size_t)7));
    double* sk = reinterpret_cast<double*>(hl + nq + 1);
    int* si = reinterpret_cast<int*>(sk + SP_SORT);
    int* ahead_of = si + SP_SORT;  // 512 ints: SP_COUNT rounded up to a power of two
    double* mk = reinterpret_cast<double*>(ahead_of + 512);
    int* mi = reinterpret_cast<int*>(mk + 3 * KS * 64);
    int* misc = mi + 3 * KS * 64;  // [0] last-workgroup flag, [1] filled, [2..5] per-wave counts
    int* stage_i = mi;             // k <= 64*KS ints fit the 3*KS*64
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#ifdef RAG_SP_TIMING
    unsigned long long ts[16];
    int nts = 0;
#define SP_STAMP() ts[nts++] = __builtin_readcyclecounter() * 0 + wall_clock64()
#else
#define SP_STAMP()
#endif
    SP_STAMP();

    // every independent load first: this workgroup's scores, the queried rows and their scores, the maxima
    double myv[SP_R / SP_T];
#pragma unroll
    for (int j = 0; j < SP_R / SP_T; ++j) {
        const long long i = (long long)blockIdx.x * SP_R + j * SP_T + tid;
        myv[j] = i < n_rows ? score[i] : -DBL_MAX;
    }
    for (int i = tid; i < nq; i += SP_T) {
        const long long r = qrow[i];
        ur[i] = r;
        uk[i] = score[r];
    }
    for (int i = tid; i <= nq; i += SP_T) hh[i] = 0u;
    const double T = thr[0];  // computed by the last workgroup of the scoring launch
    __syncthreads();
    SP_STAMP();

    // ---- b. this workgroup's rows: scores > T appended, rows scoring exactly T remembered while they are few
#pragma unroll
    for (int j = 0; j < SP_R / SP_T; ++j) {
        const long long i = (long long)blockIdx.x * SP_R + j * SP_T + tid;
        const double v = myv[j];
        const bool take = i < n_rows && v > T;
        const unsigned long long m = __ballot(take);
        if (m != 0ULL) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[0], __popcll(m));
            base = __shfl(base, 0, 64);
            if (take) {
                const int slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
                if (slot < THR_CAP) {  // write-through (sc1) stores: the hand-off below needs no L2 write-back
                    __hip_atomic_store(&buf_key[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&buf_id[slot], (int)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        const bool eq = i < n_rows && v == T;
        const unsigned long long me = __ballot(eq);
        if (me != 0ULL) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[2], __popcll(me));
            base = __shfl(base, 0, 64);
            if (eq) {
                const int slot = base + __popcll(me & ((1ULL << lane) - 1ULL));
                if (slot < THR_CAP) __hip_atomic_store(&eq_id[slot], (int)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    SP_STAMP();
    // ---- c. rank query: sort the queried (score, row) pairs best-first (every workgroup, in LDS), bin this
    // workgroup's rows among them (rank_hist_kernel's rule)
    if (nq > 0) {
        // position of pair i = number of pairs ahead of it; the count is split over `parts` threads per pair so
        // that all 256 threads work when there are few pairs (the loop is ~100 cycles per comparison)
        int np2 = 1;
        while (np2 < nq) np2 <<= 1;
        const int parts = np2 >= SP_T ? 1 : SP_T / np2;
        for (int i = tid; i < nq; i += SP_T) po[i] = 0;
        __syncthreads();
        for (int w0 = 0; w0 < nq; w0 += SP_T / parts) {
            const int i = w0 + tid / parts, part = tid % parts;
            if (i < nq) {
                const double kk = uk[i];
                const long long r = ur[i];
                int pos = 0;
                for (int j = part; j < nq; j += parts)  // a row queried twice: the earlier query sorts first
                    pos += (pair_ahead(uk[j], ur[j], kk, r) || (uk[j] == kk && ur[j] == r && j < i)) ? 1 : 0;
                if (pos) atomicAdd(&po[i], pos);
            }
        }
        __syncthreads();
        for (int i = tid; i < nq; i += SP_T) {
            const int pos = po[i];
            ks[pos] = uk[i];
            rs[pos] = ur[i];
        }
        __syncthreads();
#ifdef RAG_SP_TIMING
        const unsigned long long t_prep = wall_clock64();
#endif
        // first sorted position whose pair is NOT ahead of (v, i): the searches of a thread's rows run side by
        // side (independent LDS reads per step)
        int blo[SP_R / SP_T], bhi[SP_R / SP_T];
#pragma unroll
        for (int j = 0; j < SP_R / SP_T; ++j) {
            blo[j] = 0;
            bhi[j] = nq;
        }
        for (int span = nq; span > 0; span >>= 1) {
#pragma unroll
            for (int j = 0; j < SP_R / SP_T; ++j) {
                if (blo[j] < bhi[j]) {
                    const long long i = (long long)blockIdx.x * SP_R + j * SP_T + tid;
                    const int mid = (blo[j] + bhi[j]) >> 1;
                    if (pair_ahead(ks[mid], rs[mid], myv[j], i)) blo[j] = mid + 1;
                    else bhi[j] = mid;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SP_R / SP_T; ++j) {
            const long long i = (long long)blockIdx.x * SP_R + j * SP_T + tid;
            int bin = -1;
            if (i < n_rows) bin = blo[j];
            // most rows of a wave fall into one bin (the rows behind every queried pair)
            const int b0 = __shfl(bin, 0, 64);
            if (__all(bin == b0)) {
                if (lane == 0 && b0 >= 0) atomicAdd(&hh[b0], 64u);
            } else if (bin >= 0) {
                atomicAdd(&hh[bin], 1u);
            }
        }
        __syncthreads();
#ifdef RAG_SP_TIMING
        if (tid == 0 && blockIdx.x == 0) printf("rank phase: prep %.2f us, bin %.2f us\n", (t_prep - ts[2]) * 0.01, (wall_clock64() - t_prep) * 0.01);
#endif
        // (the last bin -- rows behind every queried pair, usually nearly all rows -- is never read)
        unsigned long long* hrep = hist + (size_t)(blockIdx.x % SP_HREP) * (size_t)(nq + 1);
        for (int i = tid; i < nq; i += SP_T)
            if (hh[i]) atomicAdd(&hrep[i], (unsigned long long)hh[i]);
    }

    SP_STAMP();
    // ---- d. the workgroup that finishes last does the rest.  Hand-off (Guideline 16, counter form): everything
    // this workgroup published went out by write-through (sc1) stores or by agent-scope atomics, every wave
    // drains its stores, then ONE lane draws the ticket behind the workgroup barrier; the last arriver
    // acquires at agent scope before it reads.  (A release fence in each of ~1000 workgroups cost 12 us here.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        // two-level ticket: one word takes ~88 returning atomics per microsecond, a thousand workgroups on
        // one word were 11 us; SP_TGRP words on cache lines of their own, then one
        const int nb = (int)gridDim.x, g = (int)blockIdx.x % SP_TGRP;
        const int gsize = (nb - g + SP_TGRP - 1) / SP_TGRP;
        int last = 0;
        const int t = __hip_atomic_fetch_add(&counters[SP_TICKET0 + g * 32], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gsize - 1) {
            const int groups = nb < SP_TGRP ? nb : SP_TGRP;
            const int t2 = __hip_atomic_fetch_add(&counters[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t2 == groups - 1) ? 1 : 0;
        }
        misc[0] = last;
    }
    __syncthreads();
    if (misc[0] == 0) return;
    SP_STAMP();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    SP_STAMP();

    // ---- e. the counters, the rank histogram and the first SP_SORT candidates in one round trip
    const int cnt = __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ceq = __hip_atomic_load(&counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < nq; i += SP_T) {
        unsigned long long a = 0;
#pragma unroll 8
        for (int r = 0; r < SP_HREP; ++r)
            a += __hip_atomic_load(&hist[(size_t)r * (size_t)(nq + 1) + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hl[i] = a;
    }
    const bool small = cnt <= SP_SORT;
    int M = 64;
    if (small) {
        while (M < cnt) M <<= 1;
        for (int i = tid; i < M; i += SP_T) {
            sk[i] = i < cnt ? buf_key[i] : -DBL_MAX;
            si[i] = i < cnt ? buf_id[i] : INT_MAX;
        }
    }
    __syncthreads();
    if (nq > 0) {
        for (int i = tid; i < nq; i += SP_T) {
            unsigned long long run = 0;
            const int pos = po[i];
            for (int p = 0; p <= pos; ++p) run += hl[p];
            if (out_rank) out_rank[i] = (long long)(run - 1ULL);
            if (out_rowscore) out_rowscore[i] = uk[i];
        }
    }

    SP_STAMP();
    // ---- f. exact selection among the appended scores, ties completed by ascending row (select_fill_kernel)
    if (cnt > THR_CAP) {  // uniform: the caller reruns the selection with the list kernels
        if (tid == 0) *out_flag = 1;
        return;
    }
    if (tid == 0) *out_flag = 0;
    const int have = cnt < k ? cnt : k;
    if (cnt <= SP_COUNT) {  // rank by counting: position = number of candidates ahead; `parts` threads per candidate
        const int parts = M >= SP_T ? 1 : SP_T / M;
        __syncthreads();
        for (int i = tid; i < M; i += SP_T) ahead_of[i] = 0;
        __syncthreads();
        for (int w0 = 0; w0 < cnt; w0 += SP_T / parts) {
            const int i = w0 + tid / parts, part = tid % parts;
            if (i < cnt) {
                const double kk = sk[i];
                const int id = si[i];
                int ahead = 0;
                for (int j = part; j < cnt; j += parts) {
                    const double ok = sk[j];
                    const int oi = si[j];
                    ahead += (ok > kk || (ok == kk && oi < id)) ? 1 : 0;
                }
                if (ahead) atomicAdd(&ahead_of[i], ahead);
            }
        }
        __syncthreads();
        for (int i = tid; i < cnt; i += SP_T) {
            const int ahead = ahead_of[i];
            if (ahead < k) {  // staged: scattered 8-byte stores to host memory are one PCIe write each
                mk[ahead] = sk[i];
                stage_i[ahead] = si[i];
            }
        }
        __syncthreads();
#ifdef RAG_SP_TIMING
        const unsigned long long t_cnt = wall_clock64();
        if (tid == 0) printf("count loop %.2f us\n", (t_cnt - ts[6]) * 0.01);
#endif
        for (int p = tid; p < have; p += SP_T) {
            out_D[p] = mk[p];
            out_I[p] = (long long)stage_i[p];
        }
#ifdef RAG_SP_TIMING
        SP_STAMP();
        if (tid == 0)
            printf("select k=%d G=%d nq=%d cnt=%d ceq=%d M=%d | us: thr %.2f compact %.2f rank %.2f ticket %.2f acquire %.2f loads+ranks %.2f count-select %.2f\n",
                   k, G, nq, cnt, ceq, M, (ts[1] - ts[0]) * 0.01, (ts[2] - ts[1]) * 0.01, (ts[3] - ts[2]) * 0.01,
                   (ts[4] - ts[3]) * 0.01, (ts[5] - ts[4]) * 0.01, (ts[6] - ts[5]) * 0.01, (ts[7] - ts[6]) * 0.01);
#endif
    } else if (small) {
        lds_sort_pairs(sk, si, M, tid);
#ifdef RAG_SP_TIMING
        SP_STAMP();
        if (tid == 0)
            printf("select k=%d G=%d nq=%d cnt=%d ceq=%d M=%d | us: thr %.2f compact %.2f rank %.2f ticket %.2f acquire %.2f loads+ranks %.2f sort %.2f\n",
                   k, G, nq, cnt, ceq, M, (ts[1] - ts[0]) * 0.01, (ts[2] - ts[1]) * 0.01, (ts[3] - ts[2]) * 0.01,
                   (ts[4] - ts[3]) * 0.01, (ts[5] - ts[4]) * 0.01, (ts[6] - ts[5]) * 0.01, (ts[7] - ts[6]) * 0.01);
#endif
        for (int p = tid; p < have; p += SP_T) {
            out_D[p] = sk[p];
            out_I[p] = (long long)si[p];
        }
    } else {
        WaveTopK<double, int, KS> tk;
        block_merge<double, int, KS, SP_T / 64>(
            [&](long long c, double& key, int& id) {
                key = buf_key[c];
                id = buf_id[c];
                return true;
            },
            (long long)cnt, k, tk, mk, mi);
        if (tid < 64) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s * 64 + lane;
                if (p < have) {
                    out_D[p] = tk.key[s];
                    out_I[p] = (long long)tk.id[s];
                }
            }
        }
    }
    if (have == k) return;
    const int need = k - have;
    if (ceq <= THR_CAP) {  // few rows score exactly T: all remembered; the lowest row numbers complete the list
        __syncthreads();
        if (ceq <= SP_SORT) {
            int M2 = 64;
            while (M2 < ceq) M2 <<= 1;
            for (int i = tid; i < M2; i += SP_T) {
                const int id = i < ceq ? eq_id[i] : INT_MAX;
                sk[i] = i < ceq ? -(double)id : -DBL_MAX;
                si[i] = id;
            }
            __syncthreads();
            lds_sort_pairs(sk, si, M2, tid);
            for (int p = tid; p < need && p < ceq; p += SP_T) {
                out_D[have + p] = T;
                out_I[have + p] = (long long)si[p];
            }
            return;
        }
        WaveTopK<double, int, KS> te;
        block_merge<double, int, KS, SP_T / 64>(
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
                    out_D[have + p] = T;
                    out_I[have + p] = (long long)te.id[s];
                }
            }
        }
        return;
    }
    // many rows score exactly T (the zeros of a BM25 vector): the first k - have of them from row 0
    if (tid == 0) misc[1] = have;
    __syncthreads();
    for (long long base = 0; base < n_rows; base += SP_T) {
        const long long i = base + tid;
        const bool eq = i < n_rows && score[i] == T;
        const unsigned long long m = __ballot(eq);
        if (lane == 0) misc[2 + wid] = __popcll(m);
        __syncthreads();
        int before = misc[1];
        for (int w = 0; w < wid; ++w) before += misc[2 + w];
        int total = 0;
        for (int w = 0; w < SP_T / 64; ++w) total += misc[2 + w];
        if (eq) {
            const int pos = before + __popcll(m & ((1ULL << lane) - 1ULL));
            if (pos < k) {
                out_D[pos] = T;
                out_I[pos] = i;
            }
        }
        __syncthreads();
        if (tid == 0) misc[1] += total;
        __syncthreads();
        if (misc[1] >= k) break;
    }
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
    // fused path: row blocks, skip table of the frequent terms, span maxima
    bool fused = false;             // postings ascend strictly by row within every term and the skip table is built
    int64_t nblk = 0;               // workgroups = ceil(n_rows / SP_R)
    int64_t n_freq = 0;             // rows of the skip table
    int* skip = nullptr;            // [n_freq][nblk + 1] posting offsets (relative to the term's first posting)
    std::vector<int> ft_of_term;    // term -> row of the skip table, -1: rare
    double* wmax = nullptr;         // [nblk * SP_SUB] maxima of the 64-row spans
    double* bmax = nullptr;         // [nblk] maxima of the row blocks
    // rag_tfidf_query_begin / _end: the handle's own stream, the k of the begun query (0 = none begun)
    hipStream_t own_st = nullptr;
    int begun_k = 0;
    unsigned long long* rk1_hist = nullptr;  // rank_one_kernel: [RANK1_HREP][RANK1_MAXQ] bins + ticket word behind them, zero between calls
    // rag_csc_set_doc_tables (rag_hybrid_rrf_mmr): per row of this index, the row of a flat index that holds the
    // document's embedding (-1: none) and the byte length of its text
    long long* doc_row = nullptr;
    long long* doc_bytes = nullptr;
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

int ensure_topk_ws(rag_csc* h, long long nb, int k, bool thr_path);

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
    {
        const int rcw = ensure_topk_ws(h, nb, k, thr_path);
        if (rcw) return rcw;
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
    RAG_HIP(hipMemcpyAsync(h->pin, h->out_D, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, st));
    RAG_HIP(hipMemcpyAsync(h->pin + PIN_BYTES / 2, h->out_I, (size_t)k * sizeof(long long), hipMemcpyDeviceToHost, st));
    return RAG_OK;
}

void topk_finish(rag_csc* h, int k, double* D, int64_t* I) {
    if (k <= 0) return;
    std::memcpy(D, h->pin, (size_t)k * sizeof(double));
    std::memcpy(I, h->pin + PIN_BYTES / 2, (size_t)k * sizeof(long long));
}


// Workspaces of the top-k (both paths).
int ensure_topk_ws(rag_csc* h, long long nb, int k, bool thr_path) {
    if (thr_path && h->thr_ws == nullptr) {
        RAG_HIP(hipMalloc(&h->thr_ws, (size_t)(SPANS + 1) * sizeof(double)));
        RAG_HIP(hipMalloc(&h->thr_cnt, SP_NCOUNTERS * sizeof(int)));
        RAG_HIP(hipMemset(h->thr_cnt, 0, SP_NCOUNTERS * sizeof(int)));
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
    if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
    return RAG_OK;
}

// Fused scoring: one launch per SP_MAXT query terms (usually one).  weights: idf per TERM ID (BM25) or per
// QUERY POSITION (TF-IDF).  n_qrow > 0: the rows of a rank query wait in pinned memory at PIN_Q_ROWS and are
// brought to rk_rows by workgroup 0 of the (last) launch, which also clears the selection kernel's counters.
int launch_fused_scores(rag_csc* h, bool bm25, int nt, const int32_t* tids, const double* qw, int n_qrow,
                        int thr_G, int thr_k, hipStream_t st) {
    if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
    if (h->thr_cnt == nullptr) {
        RAG_HIP(hipMalloc(&h->thr_ws, (size_t)(SPANS + 1) * sizeof(double)));
        RAG_HIP(hipMalloc(&h->thr_cnt, SP_NCOUNTERS * sizeof(int)));
        RAG_HIP(hipMemset(h->thr_cnt, 0, SP_NCOUNTERS * sizeof(int)));
    }
    const double k1p1 = h->k1 + 1.0, omb = 1.0 - h->b;  // Python evaluates these once per term as float64 scalars
    SpTerms qt;
    int done = 0;
    bool first = true;
    do {
        qt.n = 0;
        while (done < nt && qt.n < SP_MAXT) {
            const int t = tids[done];
            const long long b0 = h->indptr[t], b1 = h->indptr[t + 1];
            if (b1 > b0) {
                qt.freq[qt.n] = h->ft_of_term[t];
                qt.begin[qt.n] = b0;
                qt.end[qt.n] = b1;
                qt.w[qt.n] = bm25 ? h->idf[t] : qw[done];
                ++qt.n;
            }
            ++done;
        }
        const bool last = done >= nt;
        const int nqr = last ? n_qrow : 0;
        const int tG = last ? thr_G : 0;   // the threshold comes from the final scores
        const int per_block = h->nblk >= thr_G ? 1 : 0;
        double* thr_out = h->thr_ws + SPANS;
        const long long* qh = reinterpret_cast<const long long*>(h->pin + PIN_Q_ROWS);
        if (bm25)
            sparse_score_kernel<true><<<(unsigned)h->nblk, SP_T, 0, st>>>(
                qt, h->row, h->val, h->skip, (long long)(h->nblk + 1), h->doc_len, h->avgdl, h->k1, h->b, k1p1, omb,
                (long long)h->n_rows, h->score, h->wmax, h->bmax, first ? 0 : 1, h->thr_cnt, SP_ZERO_PER_CALL, h->rk_hist,
                nqr ? SP_HREP * (nqr + 1) : 0, qh, h->rk_rows, nqr, tG, thr_k, per_block, thr_out);
        else
            sparse_score_kernel<false><<<(unsigned)h->nblk, SP_T, 0, st>>>(
                qt, h->row, h->val, h->skip, (long long)(h->nblk + 1), nullptr, 0.0, 0.0, 0.0, 0.0, 0.0,
                (long long)h->n_rows, h->score, h->wmax, h->bmax, first ? 0 : 1, h->thr_cnt, SP_ZERO_PER_CALL, h->rk_hist,
                nqr ? SP_HREP * (nqr + 1) : 0, qh, h->rk_rows, nqr, tG, thr_k, per_block, thr_out);
        RAG_HIP(hipGetLastError());
        first = false;
    } while (done < nt);
    h->have_scores = true;
    return RAG_OK;
}

// Can the fused selection kernel serve (k, n_rows)?  It needs k <= the number of real span groups (else its
// threshold is -DBL_MAX and every row is a candidate, which only a small vector survives).
bool fused_select_ok(const rag_csc* h, int k, int* G_out) {
    if (!h->fused || k < 1 || k > RAG_MAX_K || k > h->n_rows || options().sparse_topk_lists != 0) return false;
    // T = the k-th largest of ng group maxima; for scores in no particular order about -ng*ln(1 - k/ng) rows
    // exceed it (82 for k = 70, ng = 256; 960 for k = 250): 256 groups serve k <= 200, 1024 the rest
    const int G = k <= 200 ? 256 : 1024;
    const long long n_g = h->nblk >= G ? h->nblk : h->nblk * SP_SUB;  // maxima per workgroup, or per 64-row span
    const long long gs = (n_g + G - 1) / G;
    const long long ng = (n_g + gs - 1) / gs;
    *G_out = G;
    return 4 * ng >= 5LL * k || h->n_rows <= THR_CAP;
}

// Selection (+ rank query) behind launch_fused_scores on the same stream; results land in pinned memory.
int launch_fused_select(rag_csc* h, int k, int G, int n_qrow, bool want_ranks, bool want_scores, hipStream_t st) {
    const int KS = ks_for(k);
    int rc = ensure_topk_ws(h, 1, k, true);
    if (rc) return rc;
    (void)G;
    const size_t smem = sp_select_smem(0, n_qrow, KS);
    double* oD = reinterpret_cast<double*>(h->pin);
    long long* oI = reinterpret_cast<long long*>(h->pin + PIN_BYTES / 2);
    long long* oR = want_ranks ? reinterpret_cast<long long*>(h->pin + PIN_Q_RANKS) : nullptr;
    double* oS = want_scores ? reinterpret_cast<double*>(h->pin + PIN_Q_SCORES) : nullptr;
    int* oF = reinterpret_cast<int*>(h->pin + PIN_FLAG);
    int* eq_id = h->cand_id + THR_CAP;
#define RAG_SP_SELECT(KSV)                                                                                          \
    do {                                                                                                            \
        auto fn = sparse_select_kernel<KSV>;                                                                        \
        if (smem > 48 * 1024)                                                                                       \
            RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)smem));                                                                \
        fn<<<(unsigned)h->nblk, SP_T, smem, st>>>(h->score, (long long)h->n_rows, h->thr_ws + SPANS, 0, k, h->cand_key, \
                                                  h->cand_id, eq_id,                                                 \
                                                  h->thr_cnt, n_qrow, h->rk_rows, h->rk_hist, oD, oI, oR, oS, oF);   \
    } while (0)
    if (KS == 1) RAG_SP_SELECT(1);
    else if (KS == 2) RAG_SP_SELECT(2);
    else RAG_SP_SELECT(4);
#undef RAG_SP_SELECT
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

// Skip table + row blocks of the fused path (host side of rag_csc_create).  false: the postings of some
// term do not ascend strictly by row, or the table would be unreasonably large -> per-term kernels.
bool build_fused_index(rag_csc* h, const int64_t* indptr, const int32_t* row, std::vector<int>* skip_host) {
    if (h->n_rows <= 0 || h->n_terms <= 0) return false;
    for (int64_t t = 0; t < h->n_terms; ++t)
        for (int64_t p = indptr[t] + 1; p < indptr[t + 1]; ++p)
            if (row[p] <= row[p - 1]) return false;
    h->nblk = (h->n_rows + SP_R - 1) / SP_R;
    h->ft_of_term.assign((size_t)h->n_terms, -1);
    int64_t nf = 0;
    for (int64_t t = 0; t < h->n_terms; ++t)
        if (indptr[t + 1] - indptr[t] >= SP_FREQ_MIN) h->ft_of_term[(size_t)t] = (int)nf++;
    if ((double)nf * (double)(h->nblk + 1) * 4.0 > 4e9) return false;  // > 4 GB of skip table
    h->n_freq = nf;
    skip_host->assign((size_t)nf * (size_t)(h->nblk + 1), 0);
    for (int64_t t = 0; t < h->n_terms; ++t) {
        const int f = h->ft_of_term[(size_t)t];
        if (f < 0) continue;
        const int32_t* r0 = row + indptr[t];
        const int64_t df = indptr[t + 1] - indptr[t];
        int* sk = skip_host->data() + (size_t)f * (size_t)(h->nblk + 1);
        int64_t p = 0;
        for (int64_t b = 0; b <= h->nblk; ++b) {  // first posting with row >= b * SP_R (one forward walk)
            const int64_t bound = b * SP_R;
            if (b == h->nblk) p = df;
            else if (p < df && r0[p] < bound) p = std::lower_bound(r0 + p, r0 + df, (int32_t)std::min<int64_t>(bound, INT_MAX)) - r0;
            sk[b] = (int)p;
        }
    }
    return true;
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
    {
        std::vector<int> skip_host;
        if (nnz > 0 && build_fused_index(h, indptr, row, &skip_host)) {
            if (e == hipSuccess) e = hipMalloc(&h->skip, std::max<size_t>(skip_host.size(), 1) * sizeof(int));
            if (e == hipSuccess && !skip_host.empty())
                e = hipMemcpy(h->skip, skip_host.data(), skip_host.size() * sizeof(int), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc(&h->wmax, (size_t)h->nblk * SP_SUB * sizeof(double));
            if (e == hipSuccess) e = hipMalloc(&h->bmax, (size_t)h->nblk * sizeof(double));
            h->fused = e == hipSuccess;
        }
    }
    if (e == hipSuccess) e = hipMalloc(&h->score, (size_t)std::max<int64_t>(n_rows, 1) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->rk_rows, RANK_MAXQ * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&h->rk_out, RANK_MAXQ * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&h->rk_skey, RANK_MAXQ * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->rk_srow, RANK_MAXQ * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&h->rk_pos, RANK_MAXQ * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->rk_hist, (size_t)SP_HREP * (RANK_MAXQ + 1) * sizeof(unsigned long long));
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
    if (h->own_st) {
        (void)hipStreamSynchronize(h->own_st);
        (void)hipStreamDestroy(h->own_st);
    }
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
    (void)hipFree(h->rk1_hist);
    (void)hipFree(h->doc_row);
    (void)hipFree(h->doc_bytes);
    (void)hipFree(h->skip);
    (void)hipFree(h->wmax);
    (void)hipFree(h->bmax);
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

// Per-term launches (postings not sorted by row, or no skip table): scores accumulate in HBM.
static int launch_term_scores(rag_csc* h, bool bm25, int nt, const int32_t* tids, const double* qw, hipStream_t st) {
    if (h->n_rows > 0) RAG_HIP(hipMemsetAsync(h->score, 0, (size_t)h->n_rows * sizeof(double), st));
    // Python evaluates (self.k1 + 1) and (1 - self.b) once per term as float64 scalars.
    const double k1p1 = h->k1 + 1.0, omb = 1.0 - h->b;
    for (int i = 0; i < nt; ++i) {
        const int t = tids[i];
        const long long b0 = h->indptr[t], b1 = h->indptr[t + 1];
        if (b1 <= b0) continue;
        if (bm25)
            bm25_term_kernel<<<term_grid(b1 - b0), 256, 0, st>>>(h->row, h->val, b0, b1, h->idf[t], h->doc_len, h->avgdl,
                                                                h->k1, h->b, k1p1, omb, h->score);
        else
            tfidf_term_kernel<<<term_grid(b1 - b0), 256, 0, st>>>(h->row, h->val, b0, b1, qw[i], h->score);
    }
    RAG_HIP(hipGetLastError());
    h->have_scores = true;
    return RAG_OK;
}

// Scores (+ optional copy of the whole vector to the host) + top-k, one synchronisation.
static int score_and_topk(rag_csc* h, bool bm25, int nt, const int32_t* tids, const double* qw, double* scores_or_null,
                          int k, double* D, int64_t* I) {
    hipStream_t st = lib_stream();
    const bool fused = h->fused && options().sparse_fused != 0;
    int G = 0;
    const bool fsel = fused && k > 0 && fused_select_ok(h, k, &G);
    int rc = fused ? launch_fused_scores(h, bm25, nt, tids, qw, 0, fsel ? G : 0, k, st)
                   : launch_term_scores(h, bm25, nt, tids, qw, st);
    if (rc) return rc;
    if (scores_or_null && h->n_rows > 0)
        RAG_HIP(hipMemcpyAsync(scores_or_null, h->score, (size_t)h->n_rows * sizeof(double), hipMemcpyDeviceToHost, st));
    if (fsel) {
        rc = launch_fused_select(h, k, G, 0, false, false, st);
        if (rc) return rc;
        RAG_HIP(hipStreamSynchronize(st));
        if (*reinterpret_cast<volatile int*>(h->pin + PIN_FLAG) == 0) {
            topk_finish(h, k, D, I);
            return RAG_OK;
        }  // buffer overflow (scores sorted by row number, say): the list kernels take over
    }
    rc = topk_scores(h, k, D, I, st);
    if (rc) return rc;
    RAG_HIP(hipStreamSynchronize(st));
    topk_finish(h, k, D, I);
    return RAG_OK;
}

int rag_bm25_scores(rag_csc_t* h, int n_qterms, const int32_t* term_ids, double* scores_or_null, int k, double* D,
                    int64_t* I) {
    RAG_REQUIRE_INIT();
    if (!h || n_qterms < 0 || (n_qterms > 0 && !term_ids) || k < 0 || (k > 0 && (!D || !I)))
        return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first (rag_tfidf_query_end)");
    if (!h->bm25) return fail(RAG_ERR_INVALID_ARG, "rag_bm25_set has not been called on this handle");
    for (int i = 0; i < n_qterms; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    return score_and_topk(h, true, n_qterms, term_ids, nullptr, scores_or_null, k, D, I);
}

int rag_tfidf_scores(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights,
                     double* scores_or_null, int k, double* D, int64_t* I) {
    RAG_REQUIRE_INIT();
    if (!h || nnz_q < 0 || (nnz_q > 0 && (!term_ids || !q_weights)) || k < 0 || (k > 0 && (!D || !I)))
        return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first (rag_tfidf_query_end)");
    for (int i = 0; i < nnz_q; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    return score_and_topk(h, false, nnz_q, term_ids, q_weights, scores_or_null, k, D, I);
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
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first (rag_tfidf_query_end)");
    for (int i = 0; i < nnz_q; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    for (int i = 0; i < n_query; ++i)
        if (rows[i] < 0 || rows[i] >= h->n_rows)
            return fail(RAG_ERR_INVALID_ARG, "row %lld outside [0,%lld)", (long long)rows[i], (long long)h->n_rows);
    hipStream_t st = lib_stream();
    const bool want_ranks = n_query > 0 && ranks != nullptr, want_scores = n_query > 0 && row_scores != nullptr;
    const int nqr = (want_ranks || want_scores) ? n_query : 0;
    if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
    if (nqr) std::memcpy(h->pin + PIN_Q_ROWS, rows, (size_t)n_query * sizeof(long long));
    const bool fused = h->fused && options().sparse_fused != 0;
    int G = 0;
    const bool fsel = fused && k > 0 && fused_select_ok(h, k, &G);
    // fused: scores, then selection + rank query in one more launch; the rows travel with the first launch
    int rc = fused ? launch_fused_scores(h, false, nnz_q, term_ids, q_weights, fsel ? nqr : 0, fsel ? G : 0, k, st)
                   : launch_term_scores(h, false, nnz_q, term_ids, q_weights, st);
    if (rc) return rc;
    if (fsel) {
        rc = launch_fused_select(h, k, G, nqr, want_ranks, want_scores, st);
        if (rc) return rc;
        RAG_HIP(hipStreamSynchronize(st));
        if (*reinterpret_cast<volatile int*>(h->pin + PIN_FLAG) != 0) {  // selection buffer overflow: list kernels
            // (the rank query's results are already in pinned memory: save them from the top-k's staging area)
            std::vector<long long> rsave;
            std::vector<double> ssave;
            if (want_ranks) rsave.assign(reinterpret_cast<long long*>(h->pin + PIN_Q_RANKS),
                                         reinterpret_cast<long long*>(h->pin + PIN_Q_RANKS) + n_query);
            if (want_scores) ssave.assign(reinterpret_cast<double*>(h->pin + PIN_Q_SCORES),
                                          reinterpret_cast<double*>(h->pin + PIN_Q_SCORES) + n_query);
            rc = topk_scores(h, k, D, I, st);
            if (rc) return rc;
            RAG_HIP(hipStreamSynchronize(st));
            topk_finish(h, k, D, I);
            if (want_ranks) std::memcpy(ranks, rsave.data(), (size_t)n_query * sizeof(long long));
            if (want_scores) std::memcpy(row_scores, ssave.data(), (size_t)n_query * sizeof(double));
            return RAG_OK;
        }
        topk_finish(h, k, D, I);
        if (want_ranks) std::memcpy(ranks, h->pin + PIN_Q_RANKS, (size_t)n_query * sizeof(long long));
        if (want_scores) std::memcpy(row_scores, h->pin + PIN_Q_SCORES, (size_t)n_query * sizeof(double));
        return RAG_OK;
    }
    rc = topk_scores(h, k, D, I, st);
    if (rc) return rc;
    if (want_ranks || want_scores)
        RAG_HIP(hipMemcpyAsync(h->rk_rows, h->pin + PIN_Q_ROWS, (size_t)n_query * sizeof(long long), hipMemcpyHostToDevice, st));
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

int rag_tfidf_query_begin(rag_csc_t* h, int nnz_q, const int32_t* term_ids, const double* q_weights, int k) {
    RAG_REQUIRE_INIT();
    if (!h || nnz_q < 0 || (nnz_q > 0 && (!term_ids || !q_weights)) || k < 1) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "rag_tfidf_query_begin: a query is already begun on this handle");
    for (int i = 0; i < nnz_q; ++i)
        if (term_ids[i] < 0 || term_ids[i] >= h->n_terms)
            return fail(RAG_ERR_INVALID_ARG, "query term id %d outside [0,%lld)", term_ids[i], (long long)h->n_terms);
    int G = 0;
    if (!(h->fused && options().sparse_fused != 0) || !fused_select_ok(h, k, &G))
        return fail(RAG_ERR_UNSUPPORTED, "rag_tfidf_query_begin: the one-launch scoring path does not serve this index or k=%d "
                                         "(use rag_tfidf_query)", k);
    if (!h->own_st) RAG_HIP(hipStreamCreateWithFlags(&h->own_st, hipStreamNonBlocking));
    hipStream_t st = h->own_st;
    // earlier calls on this handle ran on the library stream and were synchronised there before they returned
    int rc = launch_fused_scores(h, false, nnz_q, term_ids, q_weights, 0, G, k, st);
    if (rc) return rc;
    rc = launch_fused_select(h, k, G, 0, false, false, st);
    if (rc) return rc;
    h->begun_k = k;
    return RAG_OK;
}

int rag_tfidf_query_end(rag_csc_t* h, int k, double* D, int64_t* I, int n_query, const int64_t* rows, int64_t* ranks,
                        double* row_scores) {
    RAG_REQUIRE_INIT();
    if (!h || k < 0 || (k > 0 && (!D || !I)) || n_query < 0 || (n_query > 0 && !rows))
        return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->begun_k) return fail(RAG_ERR_INVALID_ARG, "rag_tfidf_query_end: no query begun on this handle");
    const int kb = h->begun_k;
    hipStream_t st = h->own_st;
    RAG_HIP(hipStreamSynchronize(st));
    h->begun_k = 0;
    if (k > kb) return fail(RAG_ERR_INVALID_ARG, "rag_tfidf_query_end: k=%d beyond the %d the query was begun with", k, kb);
    if (n_query > RANK_MAXQ)
        return fail(RAG_ERR_INVALID_ARG, "rag_tfidf_query_end: n_query=%d beyond %d", n_query, RANK_MAXQ);
    for (int i = 0; i < n_query; ++i)
        if (rows[i] < 0 || rows[i] >= h->n_rows)
            return fail(RAG_ERR_INVALID_ARG, "row %lld outside [0,%lld)", (long long)rows[i], (long long)h->n_rows);
    if (*reinterpret_cast<volatile int*>(h->pin + PIN_FLAG) != 0) {  // selection buffer overflow: list kernels
        const int rc = topk_scores(h, kb, D, I, st);
        if (rc) return rc;
        RAG_HIP(hipStreamSynchronize(st));
    }
    topk_finish(h, k, D, I);  // the first k of the kb best: a prefix of the same stable ranking
    const bool want_ranks = n_query > 0 && ranks != nullptr, want_scores = n_query > 0 && row_scores != nullptr;
    if (!want_ranks && !want_scores) return RAG_OK;
    // rank query over the score vector the begun launch left in HBM: rows read from, ranks and scores written to
    // pinned memory by the kernels themselves (three launches, one synchronisation, no copies)
    std::memcpy(h->pin + PIN_Q_ROWS, rows, (size_t)n_query * sizeof(long long));
    const long long* qrows = reinterpret_cast<const long long*>(h->pin + PIN_Q_ROWS);
    long long nb = (h->n_rows + 2047) / 2048;
    nb = std::max<long long>(1, std::min<long long>(nb, (long long)num_cus() * 4));
    if (want_ranks && n_query <= RANK1_MAXQ && options().sparse_rank_one >= 0) {
        if (!h->rk1_hist) {
            RAG_HIP(hipMalloc(&h->rk1_hist, (RANK1_HREP * RANK1_MAXQ + 8) * sizeof(unsigned long long)));
            RAG_HIP(hipMemset(h->rk1_hist, 0, (RANK1_HREP * RANK1_MAXQ + 8) * sizeof(unsigned long long)));
        }
        const long long nb1 = std::max<long long>(1, std::min<long long>((h->n_rows + RANK1_T * RANK1_RB - 1) / (RANK1_T * RANK1_RB),
                                                                         (long long)num_cus()));
        Rank1Rows qr;
        for (int i = 0; i < n_query; ++i) qr.r[i] = (int)rows[i];
        rank_one_kernel<<<(unsigned)nb1, RANK1_T, 0, st>>>(h->score, (long long)h->n_rows, n_query, qr, h->rk1_hist,
                                                          reinterpret_cast<int*>(h->rk1_hist + RANK1_HREP * RANK1_MAXQ),
                                                          reinterpret_cast<unsigned long long*>(h->pin + PIN_Q_RANKS),
                                                          want_scores ? reinterpret_cast<double*>(h->pin + PIN_Q_SCORES) : nullptr);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipStreamSynchronize(st));
        std::memcpy(ranks, h->pin + PIN_Q_RANKS, (size_t)n_query * sizeof(long long));
        if (want_scores) std::memcpy(row_scores, h->pin + PIN_Q_SCORES, (size_t)n_query * sizeof(double));
        return RAG_OK;
    }
    rank_prep_kernel<<<1, RANK_MAXQ, 0, st>>>(h->score, n_query, qrows, h->rk_skey, h->rk_srow, h->rk_pos, h->rk_hist,
                                              reinterpret_cast<double*>(h->pin + PIN_Q_SCORES));
    if (want_ranks) {
        rank_hist_kernel<<<(unsigned)nb, 256, 0, st>>>(h->score, (long long)h->n_rows, n_query, h->rk_skey, h->rk_srow, h->rk_hist);
        rank_final_kernel<<<1, RANK_MAXQ, 0, st>>>(h->rk_hist, n_query, h->rk_pos,
                                                   reinterpret_cast<unsigned long long*>(h->pin + PIN_Q_RANKS));
    }
    RAG_HIP(hipGetLastError());
    RAG_HIP(hipStreamSynchronize(st));
    if (want_ranks) std::memcpy(ranks, h->pin + PIN_Q_RANKS, (size_t)n_query * sizeof(long long));
    if (want_scores) std::memcpy(row_scores, h->pin + PIN_Q_SCORES, (size_t)n_query * sizeof(double));
    return RAG_OK;
}

int rag_csc_set_doc_tables(rag_csc_t* h, const int64_t* flat_row, const int64_t* text_bytes) {
    RAG_REQUIRE_INIT();
    if (!h) return fail(RAG_ERR_INVALID_ARG, "bad handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first");
    if (h->own_st) RAG_HIP(hipStreamSynchronize(h->own_st));
    if (!flat_row || !text_bytes) {  // drop the tables
        if (h->doc_row) RAG_HIP(hipFree(h->doc_row));
        if (h->doc_bytes) RAG_HIP(hipFree(h->doc_bytes));
        h->doc_row = h->doc_bytes = nullptr;
        return RAG_OK;
    }
    const size_t b = (size_t)std::max<int64_t>(h->n_rows, 1) * sizeof(long long);
    if (!h->doc_row) RAG_HIP(hipMalloc(&h->doc_row, b));
    if (!h->doc_bytes) RAG_HIP(hipMalloc(&h->doc_bytes, b));
    if (h->n_rows > 0) {
        RAG_HIP(hipMemcpy(h->doc_row, flat_row, (size_t)h->n_rows * sizeof(long long), hipMemcpyHostToDevice));
        RAG_HIP(hipMemcpy(h->doc_bytes, text_bytes, (size_t)h->n_rows * sizeof(long long), hipMemcpyHostToDevice));
    }
    return RAG_OK;
}

int rag_csc_ranks(rag_csc_t* h, int n_query, const int64_t* rows, int64_t* ranks) {
    RAG_REQUIRE_INIT();
    if (!h || n_query < 0 || (n_query > 0 && (!rows || !ranks))) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first (rag_tfidf_query_end)");
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

int rag_csc_count_ahead(rag_csc_t* h, int n, const double* keys, const int64_t* ids, int64_t* counts) {
    RAG_REQUIRE_INIT();
    if (!h || n < 0 || (n > 0 && (!keys || !ids || !counts))) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first (rag_tfidf_query_end)");
    if (!h->have_scores) return fail(RAG_ERR_INVALID_ARG, "no score vector has been computed on this handle");
    hipStream_t st = lib_stream();
    if (!h->pin) RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), PIN_BYTES, hipHostMallocDefault));
    if (!h->rk_gather) RAG_HIP(hipMalloc(&h->rk_gather, RANK_MAXQ * sizeof(double)));
    for (int q0 = 0; q0 < n; q0 += RANK_MAXQ) {
        const int nq = std::min(RANK_MAXQ, n - q0);
        std::memcpy(h->pin, ids + q0, (size_t)nq * sizeof(long long));
        std::memcpy(h->pin + PIN_Q_SCORES, keys + q0, (size_t)nq * sizeof(double));
        RAG_HIP(hipMemcpyAsync(h->rk_rows, h->pin, (size_t)nq * sizeof(long long), hipMemcpyHostToDevice, st));
        RAG_HIP(hipMemcpyAsync(h->rk_gather, h->pin + PIN_Q_SCORES, (size_t)nq * sizeof(double), hipMemcpyHostToDevice, st));
        long long nb = (h->n_rows + 2047) / 2048;
        nb = std::max<long long>(1, std::min<long long>(nb, (long long)num_cus() * 4));
        rank_prep_kernel<<<1, RANK_MAXQ, 0, st>>>(h->score, nq, h->rk_rows, h->rk_skey, h->rk_srow, h->rk_pos, h->rk_hist,
                                                  nullptr, h->rk_gather, (long long)h->n_rows);
        rank_hist_kernel<<<(unsigned)nb, 256, 0, st>>>(h->score, (long long)h->n_rows, nq, h->rk_skey, h->rk_srow,
                                                      h->rk_hist);
        rank_final_kernel<<<1, RANK_MAXQ, 0, st>>>(h->rk_hist, nq, h->rk_pos, h->rk_out);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipMemcpyAsync(h->pin + PIN_BYTES / 2, h->rk_out, (size_t)nq * sizeof(long long), hipMemcpyDeviceToHost, st));
        RAG_HIP(hipStreamSynchronize(st));
        std::memcpy(counts + q0, h->pin + PIN_BYTES / 2, (size_t)nq * sizeof(long long));
    }
    return RAG_OK;
}

int rag_csc_gather(rag_csc_t* h, int n_query, const int64_t* rows, double* out_scores) {
    RAG_REQUIRE_INIT();
    if (!h || n_query < 0 || (n_query > 0 && (!rows || !out_scores))) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun_k) return fail(RAG_ERR_INVALID_ARG, "a TF-IDF query is begun on this handle: end it first (rag_tfidf_query_end)");
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

namespace rag {
// cand.hip (rag_hybrid_rrf_mmr): the begun TF-IDF query of `h` seen from outside.  Runs with the handle's mutex
// held: enqueues the rank query of `rows` (rank_one_kernel, ranks left in HBM) on the handle's own stream behind the
// begun scoring + selection, then calls `use`, which enqueues its own kernel on view.st, synchronises and reads its
// results.  *consumed (set by `use`) = the query is over; otherwise it stays begun (rag_tfidf_query_end ends it).
int csc_with_begun(rag_csc_t* h, int n_query, const int* rows, int (*use)(const CscBegunView&, void*), void* ctx,
                   bool* consumed) {
    *consumed = false;
    if (!h || n_query < 0 || n_query > RANK1_MAXQ || (n_query > 0 && !rows)) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->begun_k) return fail(RAG_ERR_INVALID_ARG, "no TF-IDF query begun on this handle (rag_tfidf_query_begin)");
    for (int i = 0; i < n_query; ++i)
        if (rows[i] < 0 || rows[i] >= h->n_rows)
            return fail(RAG_ERR_INVALID_ARG, "row %d outside [0,%lld)", rows[i], (long long)h->n_rows);
    hipStream_t st = h->own_st;
    if (n_query > 0) {
        if (!h->rk1_hist) {
            RAG_HIP(hipMalloc(&h->rk1_hist, (RANK1_HREP * RANK1_MAXQ + 8) * sizeof(unsigned long long)));
            RAG_HIP(hipMemset(h->rk1_hist, 0, (RANK1_HREP * RANK1_MAXQ + 8) * sizeof(unsigned long long)));
        }
        const long long nb1 = std::max<long long>(1, std::min<long long>((h->n_rows + RANK1_T * RANK1_RB - 1) / (RANK1_T * RANK1_RB),
                                                                         (long long)num_cus()));
        Rank1Rows qr;
        for (int i = 0; i < n_query; ++i) qr.r[i] = rows[i];
        rank_one_kernel<<<(unsigned)nb1, RANK1_T, 0, st>>>(h->score, (long long)h->n_rows, n_query, qr, h->rk1_hist,
                                                          reinterpret_cast<int*>(h->rk1_hist + RANK1_HREP * RANK1_MAXQ),
                                                          h->rk_out, nullptr);
        RAG_HIP(hipGetLastError());
    }
    CscBegunView v;
    v.st = st;
    v.begun_k = h->begun_k;
    v.n_rows = h->n_rows;
    v.head_D = reinterpret_cast<const double*>(h->pin);
    v.head_I = reinterpret_cast<const long long*>(h->pin + PIN_BYTES / 2);
    v.overflow_flag = reinterpret_cast<const int*>(h->pin + PIN_FLAG);
    v.ranks = h->rk_out;
    v.doc_row = h->doc_row;
    v.doc_bytes = h->doc_bytes;
    const int rc = use(v, ctx);
    // whatever `use` did, nothing of this call is left running on the stream
    (void)hipStreamSynchronize(st);
    if (rc == RAG_OK && *consumed) h->begun_k = 0;
    return rc;
}
}  // namespace rag
