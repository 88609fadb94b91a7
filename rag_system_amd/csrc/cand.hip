// cand.hip -- candidate-set kernels: Reciprocal Rank Fusion and MMR re-ranking.
//
// Replaces SearchEngine.reciprocal_rank_fusion (search_engine.py:352-359) and the numeric
// part of SearchEngine.mmr_selection (search_engine.py:419-465).  Candidate sets are tiny
// (tens to a few thousand entries), so each call is ONE workgroup; the work is latency-bound
// and reported in microseconds, not against a roofline.  Compiled with -ffp-contract=off so
// float64 sums match Python's.
#include "common.h"

#include <cfloat>
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

using namespace rag;

constexpr int CAND_THREADS = 1024;
constexpr int RRF_MAX_ENTRIES = 16384;
constexpr int MMR_MAX_C = 2048;
constexpr int MMR_SMALL_C = 64;

// ids: all rankings concatenated (entry e belongs to ranking r with rank e - start[r]).
// first[e] = smallest e' <= e with the same id.  An id's score is accumulated by the thread
// owning its first entry, over its entries in order (= the order Python adds them).
// Up to RRF_LDS_T entries (every call the search engine makes: three rankings of <= 9*top_k entries) the
// ids, ranks and first-occurrence table live in LDS: the O(T^2) comparisons never leave the CU, and the
// inputs may then sit in pinned HOST memory -- they are read exactly once, coalesced -- so that a call
// is one launch and one synchronisation with no copies.  Larger T: rrf_sort_kernel below.
constexpr int RRF_LDS_T = 2048;
__global__ __launch_bounds__(CAND_THREADS) void rrf_kernel(const long long* __restrict__ ids_g,
                                                           const int* __restrict__ rank_g, int T, double k_const,
                                                           long long* __restrict__ out_ids,
                                                           double* __restrict__ out_scores, int* __restrict__ n_out) {
    __shared__ int s_count[CAND_THREADS / 64];
    __shared__ long long s_ids[RRF_LDS_T];
    __shared__ int s_rank[RRF_LDS_T];
    __shared__ int s_first[RRF_LDS_T];
    const int tid = threadIdx.x;
    for (int e = tid; e < T; e += CAND_THREADS) {
        s_ids[e] = ids_g[e];
        s_rank[e] = rank_g[e];
    }
    __syncthreads();
    const long long* ids = s_ids;
    const int* rank_of = s_rank;
    int* first = s_first;
    for (int e = tid; e < T; e += CAND_THREADS) {
        const long long id = ids[e];
        int f = e;
        for (int j = 0; j < e; ++j)
            if (ids[j] == id) {
                f = j;
                break;
            }
        first[e] = f;
    }
    __syncthreads();
    // contiguous span per thread -> ordered compaction of distinct ids
    const int per = (T + CAND_THREADS - 1) / CAND_THREADS;
    const int lo = tid * per, hi = min(T, lo + per);
    int cnt = 0;
    for (int e = lo; e < hi; ++e) cnt += first[e] == e;
    // exclusive prefix of the per-thread counts: a scan inside each wave, then one over the 16 wave totals
    // (thread 0 walking all 1024 counts alone was most of this kernel's 20 us)
    const int lane = tid & 63, wid = tid >> 6;
    int incl = cnt;
    for (int dd = 1; dd < 64; dd <<= 1) {
        const int y = __shfl_up(incl, dd);
        if (lane >= dd) incl += y;
    }
    if (lane == 63) s_count[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        constexpr int NW = CAND_THREADS / 64;
        const int w = lane < NW ? s_count[lane] : 0;
        int ws = w;
        for (int dd = 1; dd < NW; dd <<= 1) {
            const int y = __shfl_up(ws, dd);
            if (lane >= dd) ws += y;
        }
        if (lane < NW) s_count[lane] = ws - w;
        if (lane == NW - 1) *n_out = ws;
    }
    __syncthreads();
    int o = s_count[wid] + incl - cnt;
    for (int e = lo; e < hi; ++e) {
        if (first[e] != e) continue;
        const long long id = ids[e];
        double s = 0.0;
        for (int j = e; j < T; ++j)
            if (ids[j] == id) s = s + 1.0 / (k_const + (double)rank_of[j]);
        out_ids[o] = id;
        out_scores[o] = s;
        ++o;
    }
}

// 2048 < T <= 16384 entries (top_k in the hundreds): the quadratic table above would take a millisecond.
// Here every id gets the entry number of its first occurrence through a hash table in LDS (open addressing;
// a slot belongs to the id that claimed it, its value is the smallest entry number seen for that id), the keys
// (first occurrence << 14 | entry) are sorted in LDS, and a sorted run of equal first occurrences is one
// distinct id with its entries in input order: the same sums in the same order as Python's dictionary
// (search_engine.py:352-359), output in first-appearance order.  O(T log^2 T / 1024) per thread, and a run is
// walked by one thread (all entries the same id: T additions, as before).
constexpr int RRF_HASH_SLOTS = 32768;  // >= 2 * RRF_MAX_ENTRIES; 128 KiB of LDS, reused for the <= 16384 sort keys
__global__ __launch_bounds__(CAND_THREADS) void rrf_sort_kernel(const long long* __restrict__ ids,
                                                                const int* __restrict__ rank_of, int T, double k_const,
                                                                long long* __restrict__ out_ids,
                                                                double* __restrict__ out_scores, int* __restrict__ n_out,
                                                                int* __restrict__ slot_of) {
    extern __shared__ unsigned int sm[];
    __shared__ int s_count[CAND_THREADS / 64];
    const int tid = threadIdx.x;
    int* table = reinterpret_cast<int*>(sm);
    for (int i = tid; i < RRF_HASH_SLOTS; i += CAND_THREADS) table[i] = INT_MAX;
    __syncthreads();
    for (int e = tid; e < T; e += CAND_THREADS) {
        const long long id = ids[e];
        unsigned int slot = (unsigned int)(((unsigned long long)id * 0x9E3779B97F4A7C15ULL) >> 49);  // 15 bits
        for (;;) {
            int cur = table[slot];
            if (cur == INT_MAX) {
                cur = atomicCAS(&table[slot], INT_MAX, e);
                if (cur == INT_MAX) break;  // claimed for this id
            }
            if (ids[cur] == id) {           // (whichever entry of the owner is recorded right now: the same id)
                atomicMin(&table[slot], e);
                break;
            }
            slot = (slot + 1) & (RRF_HASH_SLOTS - 1);
        }
        slot_of[e] = (int)slot;
    }
    __syncthreads();
    unsigned int mykey[RRF_MAX_ENTRIES / CAND_THREADS];
#pragma unroll
    for (int u = 0; u < RRF_MAX_ENTRIES / CAND_THREADS; ++u) {
        const int e = tid + u * CAND_THREADS;
        mykey[u] = e < T ? ((unsigned int)table[slot_of[e]] << 14) | (unsigned int)e : 0xFFFFFFFFu;
    }
    __syncthreads();  // the table is dead: its LDS becomes the key array
    int N = 2048;
    while (N < T) N <<= 1;
#pragma unroll
    for (int u = 0; u < RRF_MAX_ENTRIES / CAND_THREADS; ++u) {
        const int e = tid + u * CAND_THREADS;
        if (e < N) sm[e] = mykey[u];
    }
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1) {
        for (int st = size >> 1; st > 0; st >>= 1) {
            for (int p = tid; p < N / 2; p += CAND_THREADS) {
                const int i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
                const int jx = i + st;
                const bool asc = (i & size) == 0;
                const unsigned int a = sm[i], b = sm[jx];
                if (asc ? a > b : a < b) {
                    sm[i] = b;
                    sm[jx] = a;
                }
            }
            __syncthreads();
        }
    }
    // a run starts where entry == first occurrence; contiguous span of sorted positions per thread
    const int per = (T + CAND_THREADS - 1) / CAND_THREADS;
    const int lo = tid * per, hi = min(T, lo + per);
    int cnt = 0;
    for (int p = lo; p < hi; ++p) cnt += (sm[p] >> 14) == (sm[p] & 16383u);
    const int lane = tid & 63, wid = tid >> 6;
    int incl = cnt;
    for (int dd = 1; dd < 64; dd <<= 1) {
        const int y = __shfl_up(incl, dd);
        if (lane >= dd) incl += y;
    }
    if (lane == 63) s_count[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        constexpr int NW = CAND_THREADS / 64;
        const int w = lane < NW ? s_count[lane] : 0;
        int ws = w;
        for (int dd = 1; dd < NW; dd <<= 1) {
            const int y = __shfl_up(ws, dd);
            if (lane >= dd) ws += y;
        }
        if (lane < NW) s_count[lane] = ws - w;
        if (lane == NW - 1) *n_out = ws;
    }
    __syncthreads();
    int o = s_count[wid] + incl - cnt;
    for (int p = lo; p < hi; ++p) {
        const unsigned int f = sm[p] >> 14;
        if (f != (sm[p] & 16383u)) continue;
        double s = 0.0;
        for (int pp = p; pp < T && (sm[pp] >> 14) == f; ++pp)
            s = s + 1.0 / (k_const + (double)rank_of[sm[pp] & 16383u]);
        out_ids[o] = ids[f];
        out_scores[o] = s;
        ++o;
    }
}

// MMR.  cand: C x d (un-normalised fp32), q: d.  Work arrays in HBM: nrm (C x d normalised),
// G (C x C cosine), simq (C).  One workgroup; selection order written to out_order.
// STORED: candidate i is row rows[i] of a flat index' arena (fp32 or bfloat16 rows of `ld` elements) instead of
// row i of `cand` -- the embeddings the index already holds, with no re-encoding and no upload.
// LROWS (up to 64 candidates whose rows fit 96 KiB of LDS -- every call the search engine makes): the query and the
// candidate rows are read from (pinned / HBM) memory ONCE into LDS and every later phase works there; the same
// float operations in the same order as the path through `nrm` / `simq` in HBM, four memory round trips fewer.
constexpr size_t MMR_LDS_BYTES = 96 << 10;
// (a __device__ body so that hybrid_tail_kernel below runs the very same selection behind its fusion; pointers
// may address LDS: they are generic)
template <bool STORED, bool LROWS>
__device__ __forceinline__ void mmr_body(const float* cand, const float* q, int C, int d, double lambda, int top_k,
                                         const long long* byte_len, long long max_bytes, float* nrm, float* G,
                                         float* simq, int* out_order, int* out_n, const long long* rows,
                                         const void* arena, int ld, int bf16) {
    auto cval = [&](int i, int c) -> float {
        if (!STORED) return cand[(size_t)i * d + c];
        const size_t o = (size_t)rows[i] * ld + c;
        if (bf16) return __uint_as_float((unsigned int)reinterpret_cast<const unsigned short*>(arena)[o] << 16);
        return reinterpret_cast<const float*>(arena)[o];
    };
    __shared__ float s_inv[MMR_MAX_C];     // 1/||cand_i|| (fp32, zero norm -> 1e-9 as :422)
    __shared__ float s_maxsim[MMR_MAX_C];  // max cosine to the selected set
    __shared__ unsigned char s_state[MMR_MAX_C];  // 0 = candidate, 1 = selected/removed
    __shared__ double s_best[CAND_THREADS / 64];
    __shared__ int s_besti[CAND_THREADS / 64];
    __shared__ float s_qinv;
    __shared__ int s_pick;
    __shared__ long long s_bytes[MMR_MAX_C];  // read once: the inputs may sit in pinned host memory
    __shared__ float s_G[MMR_SMALL_C * MMR_SMALL_C];  // C <= 64: the Gram matrix never leaves the CU
    extern __shared__ __align__(16) float mmr_dyn[];  // LROWS: [d] query, [C][d] rows
    __shared__ float s_simq[MMR_SMALL_C];
    const bool small = C <= MMR_SMALL_C;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NW = CAND_THREADS / 64;
    for (int i = tid; i < C; i += CAND_THREADS) s_bytes[i] = byte_len[i];
    if (LROWS) {
        for (int c = tid; c < d; c += CAND_THREADS) mmr_dyn[c] = q[c];
        for (size_t e = tid; e < (size_t)C * d; e += CAND_THREADS) mmr_dyn[d + e] = cval((int)(e / d), (int)(e % d));
        __syncthreads();
        q = mmr_dyn;
        nrm = mmr_dyn + d;
        simq = s_simq;
    }
    auto rval = [&](int i, int c) -> float { return LROWS ? nrm[(size_t)i * d + c] : cval(i, c); };

    // row norms (one wave per row), query norm
    for (int i = wid; i <= C; i += NW) {
        float ss = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float x = i < C ? rval(i, c) : q[c];
            ss = fmaf(x, x, ss);
        }
        for (int dist = 32; dist >= 1; dist >>= 1) ss += __shfl_xor(ss, dist);
        float n = sqrtf(ss);
        if (i < C) {
            if (n == 0.f) n = 1e-9f;
            if (lane == 0) s_inv[i] = n;
        } else if (lane == 0) {
            s_qinv = n == 0.f ? 1.f : n;
        }
    }
    __syncthreads();
    for (size_t e = tid; e < (size_t)C * d; e += CAND_THREADS) nrm[e] = rval((int)(e / d), (int)(e % d)) / s_inv[e / d];
    __syncthreads();
    // cosine to the query and the Gram matrix (one wave per dot product)
    for (int i = wid; i < C; i += NW) {
        const float* a = nrm + (size_t)i * d;
        float s = 0.f, na = 0.f;
        for (int c = lane; c < d; c += 64) {
            s = fmaf(a[c], q[c], s);
            na = fmaf(a[c], a[c], na);
        }
        for (int dist = 32; dist >= 1; dist >>= 1) {
            s += __shfl_xor(s, dist);
            na += __shfl_xor(na, dist);
        }
        // sklearn's cosine_similarity re-normalises both sides
        na = sqrtf(na);
        if (na == 0.f) na = 1.f;
        if (lane == 0) {
            simq[i] = s / (na * s_qinv);
            s_inv[i] = na;  // now holds the norm of the normalised row (~1)
        }
    }
    __syncthreads();
    for (long long pr = wid; pr < (long long)C * C; pr += NW) {
        const int i = (int)(pr / C), j = (int)(pr % C);
        if (j < i) continue;
        const float* a = nrm + (size_t)i * d;
        const float* b = nrm + (size_t)j * d;
        float s = 0.f;
        for (int c = lane; c < d; c += 64) s = fmaf(a[c], b[c], s);
        for (int dist = 32; dist >= 1; dist >>= 1) s += __shfl_xor(s, dist);
        if (lane == 0) {
            const float v = s / (s_inv[i] * s_inv[j]);
            if (small) {
                s_G[i * MMR_SMALL_C + j] = v;
                s_G[j * MMR_SMALL_C + i] = v;
            } else {
                G[(size_t)i * C + j] = v;
                G[(size_t)j * C + i] = v;
            }
        }
    }
    for (int i = tid; i < C; i += CAND_THREADS) {
        s_state[i] = 0;
        s_maxsim[i] = -FLT_MAX;
    }
    __syncthreads();

    if (small) {
        // Up to 64 candidates (every call the search engine makes: top_k .. a few times top_k): ONE wave runs
        // the greedy loop, candidate i in lane i -- same scores, same comparisons, same order as the block-wide
        // loop below (first index wins ties, like np.argmax), without two barriers and an LDS hand-off per pick.
        if (wid != 0) return;
        const bool live0 = lane < C;
        const double sq = live0 ? (double)simq[lane] : 0.0;
        const long long bytes = live0 ? s_bytes[lane] : 0;
        float maxsim = -FLT_MAX;
        bool dead = !live0;
        int n_sel = 0, remaining = C, last = -1;
        long long total_bytes = 0;
        bool first_pick = true;
        while (remaining > 0 && n_sel < top_k) {
            if (last >= 0 && !dead) maxsim = fmaxf(maxsim, s_G[lane * MMR_SMALL_C + last]);
            double best = dead ? -DBL_MAX : (first_pick ? sq : lambda * sq - (1.0 - lambda) * (double)maxsim);
            int besti = dead ? INT_MAX : lane;
            if (!(best > -DBL_MAX)) {  // NaN (or dead): never picked, and must not poison the butterfly below
                best = -DBL_MAX;
                besti = INT_MAX;
            }
            for (int dist = 32; dist >= 1; dist >>= 1) {
                const double ob = __shfl_xor(best, dist);
                const int oi = __shfl_xor(besti, dist);
                if (ob > best || (ob == best && oi < besti)) {
                    best = ob;
                    besti = oi;
                }
            }
            const int pick = besti;
            if (pick >= C) break;  // only NaN scores left
            last = -1;
            const long long pb = __shfl(bytes, pick);
            bool take = true;
            if (!first_pick && max_bytes >= 0 && total_bytes + pb > max_bytes) take = false;
            if (take) {
                total_bytes += pb;
                if (lane == 0) out_order[n_sel] = pick;
                ++n_sel;
                last = pick;
            }
            if (lane == pick) dead = true;
            --remaining;
            first_pick = false;
        }
        if (lane == 0) *out_n = n_sel;
        return;
    }

    int n_sel = 0;
    long long total_bytes = 0;
    int last = -1;
    bool first_pick = true;
    int remaining = C;
    while (remaining > 0 && n_sel < top_k) {
        // score every live candidate; block argmax, first index wins ties (np.argmax)
        double best = -DBL_MAX;
        int besti = INT_MAX;
        for (int i = tid; i < C; i += CAND_THREADS) {
            if (s_state[i]) continue;
            if (last >= 0) s_maxsim[i] = fmaxf(s_maxsim[i], G[(size_t)i * C + last]);
            const double sc = first_pick ? (double)simq[i]
                                         : lambda * (double)simq[i] - (1.0 - lambda) * (double)s_maxsim[i];
            if (sc > best || (sc == best && i < besti)) {
                best = sc;
                besti = i;
            }
        }
        for (int dist = 32; dist >= 1; dist >>= 1) {
            const double ob = __shfl_xor(best, dist);
            const int oi = __shfl_xor(besti, dist);
            if (ob > best || (ob == best && oi < besti)) {
                best = ob;
                besti = oi;
            }
        }
        if (lane == 0) {
            s_best[wid] = best;
            s_besti[wid] = besti;
        }
        __syncthreads();
        if (tid == 0) {
            double b = s_best[0];
            int bi = s_besti[0];
            for (int w = 1; w < NW; ++w)
                if (s_best[w] > b || (s_best[w] == b && s_besti[w] < bi)) {
                    b = s_best[w];
                    bi = s_besti[w];
                }
            s_pick = bi;
        }
        __syncthreads();
        const int pick = s_pick;
        if (pick >= C) break;  // only NaN scores left: nothing selectable (block-uniform)
        last = -1;
        // byte budget (search_engine.py:457-463): the first pick is never checked (:437)
        bool take = true;
        if (!first_pick && max_bytes >= 0 && total_bytes + s_bytes[pick] > max_bytes) take = false;
        if (take) {
            total_bytes += s_bytes[pick];
            if (tid == 0) out_order[n_sel] = pick;
            ++n_sel;
            last = pick;
        }
        if (tid == 0) s_state[pick] = 1;
        --remaining;
        first_pick = false;
        __syncthreads();
    }
    if (tid == 0) *out_n = n_sel;
}

template <bool STORED, bool LROWS>
__global__ __launch_bounds__(CAND_THREADS) void mmr_kernel(const float* __restrict__ cand, const float* __restrict__ q,
                                                           int C, int d, double lambda, int top_k,
                                                           const long long* __restrict__ byte_len, long long max_bytes,
                                                           float* __restrict__ nrm, float* __restrict__ G,
                                                           float* __restrict__ simq, int* __restrict__ out_order,
                                                           int* __restrict__ out_n, const long long* __restrict__ rows,
                                                           const void* __restrict__ arena, int ld, int bf16) {
    mmr_body<STORED, LROWS>(cand, q, C, d, lambda, top_k, byte_len, max_bytes, nrm, G, simq, out_order, out_n, rows, arena,
                            ld, bf16);
}

// ---------------------------------------------------------------- hybrid tail (rag_hybrid_rrf_mmr)
// Everything search_rrf does behind its three rankings, in ONE workgroup and one launch (search_engine.py:340-347,
// :121-133, :396-468): RRF over the dense ranking, the BM25 ranking and the FULL TF-IDF ranking -- of which only the
// ranks of the dense / BM25 candidates A (rank_one_kernel, enqueued just before) and the head (the begun query's
// selection: the first top_k + |A| documents) can reach the result, rag_system_amd/search_engine.py -- the stable
// descending sort, the cut to top_k, the knee cutoff and the threshold of search_with_strategy, and MMR over what
// is left, with the documents' stored rows as embeddings (tables of rag_csc_set_doc_tables).  float64 sums in
// Python's order, ties by dictionary insertion order: bit-identical to rag_rrf + host fusion + rag_flat_mmr.
constexpr int HYB_MAX_LIST = 128;   // entries of the dense / of the BM25 ranking
constexpr int HYB_MAX_A = 256;      // distinct candidates of the two
constexpr int HYB_MAX_HEAD = 320;   // TF-IDF head entries (top_k + |A|)
constexpr int HYB_MAX_TOPK = 64;
struct HybArgs {
    int n_dense, n_bm25, nA, n_head, top_k, use_knee, do_mmr;
    int d, ld, bf16;
    double k_const, threshold, lambda;
    long long max_bytes, n_arena;
    unsigned short dense_a[HYB_MAX_LIST], bm25_a[HYB_MAX_LIST];  // the rankings as indices into A
    int a_row[HYB_MAX_A];   // TF-IDF row of candidate a, -1: the document has none
    short a_rk[HYB_MAX_A];  // position of its rank in `ranks`, -1: none
};
struct HybOut {
    int status;    // 0 ok, 1 = the head is not valid (selection overflow): nothing else was written
    int n_fused;   // entries of the fused list (<= top_k), best first
    int n_kept;    // of which pass the knee cutoff and the threshold (a subsequence, same order)
    int mmr_done;  // 1: order / n_sel hold the MMR selection over the kept entries
    int n_sel;
    int pad[3];
    int fa[HYB_MAX_TOPK];          // index into A, or -1: a document of the TF-IDF head
    long long frow[HYB_MAX_TOPK];  // its TF-IDF row (-1: none)
    double fscore[HYB_MAX_TOPK];
    int order[HYB_MAX_TOPK];       // MMR: indices into the kept entries, in selection order
};
// (the arguments come through pinned memory into LDS: as a by-value kernel argument the struct's arrays, indexed
// per lane, were spilled to scratch by every one of the 1024 lanes)
__global__ __launch_bounds__(CAND_THREADS) void hybrid_tail_kernel(
    const HybArgs* __restrict__ ap, const long long* __restrict__ head_I, const int* __restrict__ overflow_flag,
    const unsigned long long* __restrict__ ranks, const long long* __restrict__ doc_row,
    const long long* __restrict__ doc_bytes, const float* __restrict__ q, const void* __restrict__ arena,
    HybOut* __restrict__ out) {
    constexpr int NF = HYB_MAX_A + HYB_MAX_HEAD;
    constexpr int NW = CAND_THREADS / 64;
    extern __shared__ __align__(16) float hyb_dyn[];  // the MMR stage's LDS image; [0, d) = the query (mmr_body)
    __shared__ double s_score[NF];
    __shared__ double s_inv[HYB_MAX_HEAD];     // 1 / (k + rank) of the list positions
    __shared__ long long s_row[NF];
    __shared__ unsigned long long s_rank[HYB_MAX_A];
    __shared__ unsigned char s_live[NF];
    __shared__ int s_topf[HYB_MAX_TOPK];       // fused position -> entry
    __shared__ long long s_mrow[HYB_MAX_TOPK], s_mbytes[HYB_MAX_TOPK];
    __shared__ int s_nkept, s_mmr_ok, s_status;
    __shared__ int s_kept[HYB_MAX_TOPK];
    __shared__ HybArgs a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // Everything that comes from pinned host memory or waits in HBM is requested at once: the arguments, the
    // overflow flag and the head of the begun selection, the candidates' ranks, and the query for the MMR stage
    // (one trip over the host link costs ~2.6 us here; the phases below are latency-bound LDS work, laid out so
    // that a wave works on one entry at a time with ballots instead of every lane walking a list alone: the
    // lane-serial version of this kernel spent 24 us in the two loops)
    static_assert(sizeof(HybArgs) % 4 == 0, "copied word by word");
    for (int w = tid; w < (int)(sizeof(HybArgs) / 4); w += CAND_THREADS)
        reinterpret_cast<int*>(&a)[w] = reinterpret_cast<const int*>(ap)[w];
    if (tid == 0) s_status = *overflow_flag != 0 ? 1 : 0;
    long long head_v = -1;     // (n_head <= 320 < CAND_THREADS: one entry per lane)
    if (tid < HYB_MAX_HEAD) head_v = head_I[tid];
    if (tid < HYB_MAX_A) s_rank[tid] = ranks[tid];   // (the buffer holds 1024 entries; those beyond the query are stale, unused)
    __syncthreads();
    const int nF = a.nA + a.n_head;
    if (a.do_mmr)
        for (int c = tid; c < a.d; c += CAND_THREADS) hyb_dyn[c] = q[c];
    if (tid < a.n_head) s_row[a.nA + tid] = head_v;
    for (int i = tid; i < a.nA; i += CAND_THREADS) s_row[i] = a.a_row[i];
    if (tid < HYB_MAX_HEAD) s_inv[tid] = 1.0 / (a.k_const + (double)tid);
    __syncthreads();
    if (s_status) {
        if (tid == 0) out->status = 1;
        return;
    }
    // scores: a candidate's terms in Python's order (dense ranking, BM25 ranking entry by entry, TF-IDF)
    for (int f = wid; f < a.nA; f += NW) {
        double s = 0.0;
        for (int base = 0; base < a.n_dense; base += 64) {
            const int jj = base + lane;
            unsigned long long m = __ballot(jj < a.n_dense && a.dense_a[jj] == f);
            while (m) {
                s = s + s_inv[base + __builtin_ctzll(m)];
                m &= m - 1;
            }
        }
        for (int base = 0; base < a.n_bm25; base += 64) {
            const int jj = base + lane;
            unsigned long long m = __ballot(jj < a.n_bm25 && a.bm25_a[jj] == f);
            while (m) {
                s = s + s_inv[base + __builtin_ctzll(m)];
                m &= m - 1;
            }
        }
        if (a.a_rk[f] >= 0) s = s + 1.0 / (a.k_const + (double)s_rank[a.a_rk[f]]);
        if (lane == 0) {
            s_score[f] = s;
            s_live[f] = 1;
        }
    }
    // a head document that is no candidate has its TF-IDF term only; one that is, is already counted
    for (int h = wid; h < a.n_head; h += NW) {
        const long long r = s_row[a.nA + h];
        bool found = false;
        for (int base = 0; base < a.nA; base += 64) {
            const int i2 = base + lane;
            found = found || __any(i2 < a.nA && s_row[i2] == r);
        }
        if (lane == 0) {
            s_score[a.nA + h] = 0.0 + s_inv[h];
            s_live[a.nA + h] = found ? 0 : 1;
        }
    }
    __syncthreads();
    // stable descending order by counting (ties: dictionary insertion order = entry number); cut to top_k
    const int nlive = __syncthreads_count(tid < nF && s_live[tid]);   // (nF <= 576 < CAND_THREADS)
    const int nfused = nlive < a.top_k ? nlive : a.top_k;
    for (int f = wid; f < nF; f += NW) {
        if (!s_live[f]) continue;
        const double s = s_score[f];
        int pos = 0;
        for (int base = 0; base < nF; base += 64) {
            const int g = base + lane;
            bool ahead = false;
            if (g < nF && s_live[g]) {
                const double t = s_score[g];
                ahead = t > s || (t == s && g < f);
            }
            pos += __popcll(__ballot(ahead));
        }
        if (lane == 0 && pos < a.top_k) s_topf[pos] = f;
    }
    __syncthreads();
    if (tid < nfused) {
        const int f = s_topf[tid];
        out->fa[tid] = f < a.nA ? f : -1;
        out->frow[tid] = s_row[f];
        out->fscore[tid] = s_score[f];
    }
    if (tid == 0) {
        // search_with_strategy :121-129: (already sorted) knee cutoff with strict >, then threshold with >=
        int nk = 0;
        double cutoff = 0.0;
        bool knee = a.use_knee != 0 && nfused > 0;
        if (knee) {
            if (nfused < 2) cutoff = s_score[s_topf[0]];
            else {
                int best = 0;
                double bd = s_score[s_topf[0]] - s_score[s_topf[1]];
                for (int i = 1; i < nfused - 1; ++i) {
                    const double dd = s_score[s_topf[i]] - s_score[s_topf[i + 1]];
                    if (dd > bd) {
                        bd = dd;
                        best = i;
                    }
                }
                cutoff = s_score[s_topf[best + 1]];
            }
        }
        for (int i = 0; i < nfused; ++i) {
            const double s = s_score[s_topf[i]];
            if (knee && !(s > cutoff)) continue;
            if (!(s >= a.threshold)) continue;
            s_kept[nk++] = i;
        }
        s_nkept = nk;
        s_mmr_ok = a.do_mmr != 0 && nk > 0 && doc_row != nullptr && doc_bytes != nullptr;
        out->status = 0;
        out->n_fused = nfused;
        out->n_kept = nk;
    }
    __syncthreads();
    // the embeddings' rows and the byte lengths of the kept documents: one lane each, all loads in flight at once
    bool bad = false;
    if (s_mmr_ok && tid < s_nkept) {
        const long long r = s_row[s_topf[s_kept[tid]]];
        long long ar = -1, nb = 0;
        if (r >= 0) {
            ar = doc_row[r];
            nb = doc_bytes[r];
        }
        bad = ar < 0 || ar >= a.n_arena;
        s_mrow[tid] = bad ? 0 : ar;
        s_mbytes[tid] = nb;
    }
    const int nbad = __syncthreads_count(bad);
    if (tid == 0) {
        const int ok = s_mmr_ok && nbad == 0;
        out->mmr_done = ok;
        if (!ok) out->n_sel = 0;
    }
    if (!s_mmr_ok || nbad != 0) return;
    // (the query is in the LDS image already: mmr_body copies it onto itself)
    mmr_body<true, true>(nullptr, hyb_dyn, s_nkept, a.d, a.lambda, a.top_k, s_mbytes, a.max_bytes, nullptr, nullptr, nullptr,
                         out->order, &out->n_sel, s_mrow, arena, a.ld, a.bf16);
}

inline size_t al8(size_t b) { return (b + 7) & ~(size_t)7; }

// Process-wide workspace of the candidate-set calls (they are serialised by its mutex).
struct CandWs {
    std::mutex mu;
    char* dev = nullptr;
    size_t dev_cap = 0;
    char* pin = nullptr;
    size_t pin_cap = 0;
    int reserve(size_t dev_bytes, size_t pin_bytes) {
        if (dev_bytes > dev_cap) {
            if (dev) RAG_HIP(hipFree(dev));
            dev = nullptr; dev_cap = 0;
            const size_t want = std::max<size_t>(dev_bytes * 2, (size_t)1 << 20);
            RAG_HIP(hipMalloc(reinterpret_cast<void**>(&dev), want));
            dev_cap = want;
        }
        if (pin_bytes > pin_cap) {
            if (pin) RAG_HIP(hipHostFree(pin));
            pin = nullptr; pin_cap = 0;
            const size_t want = std::max<size_t>(pin_bytes * 2, (size_t)256 << 10);
            RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin), want, hipHostMallocDefault));
            pin_cap = want;
        }
        return RAG_OK;
    }
};
CandWs& cand_ws() {
    static CandWs* w = new CandWs();  // never destroyed: no HIP calls at process exit
    return *w;
}

}  // namespace

extern "C" {

int rag_rrf(int n_rankings, const int64_t* ids, const int64_t* lens, double k_const, int64_t* out_ids,
            double* out_scores, int64_t* n_out) {
    RAG_REQUIRE_INIT();
    if (n_rankings < 0 || (n_rankings > 0 && !lens) || !n_out) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    int64_t T = 0;
    for (int r = 0; r < n_rankings; ++r) {
        if (lens[r] < 0) return fail(RAG_ERR_INVALID_ARG, "negative ranking length");
        T += lens[r];
    }
    *n_out = 0;
    if (T == 0) return RAG_OK;
    if (!ids || !out_ids || !out_scores) return fail(RAG_ERR_INVALID_ARG, "NULL buffer");
    if (T > RRF_MAX_ENTRIES)
        return fail(RAG_ERR_UNSUPPORTED, "%lld ranking entries exceed the candidate-set limit %d", (long long)T,
                    RRF_MAX_ENTRIES);
    std::vector<int> rank_of((size_t)T);
    {
        int64_t e = 0;
        for (int r = 0; r < n_rankings; ++r)
            for (int64_t j = 0; j < lens[r]; ++j) rank_of[(size_t)e++] = (int)j;
    }
    // One device arena and one pinned host arena per process, kept across calls: a call is ONE
    // host-to-device copy, the kernel, ONE copy back and ONE synchronisation (the per-call
    // hipMalloc / hipFree of eight small buffers cost more than the kernel).
    hipStream_t st = lib_stream();
    CandWs& ws = cand_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    const size_t ids_b = (size_t)T * sizeof(long long), rank_b = al8((size_t)T * sizeof(int));
    const size_t in_b = ids_b + rank_b;
    const size_t out_b = 8 + (size_t)T * (sizeof(long long) + sizeof(double));
    const size_t first_b = al8((size_t)T * sizeof(int));
    int rc = ws.reserve(in_b + out_b + first_b, in_b + out_b);
    if (rc) return rc;
    char* hin = ws.pin;
    char* hout = ws.pin + in_b;
    std::memcpy(hin, ids, ids_b);
    std::memcpy(hin + ids_b, rank_of.data(), (size_t)T * sizeof(int));
    char* din = ws.dev;
    char* dout = ws.dev + in_b;
    char* dfirst = dout + out_b;
    if (T <= RRF_LDS_T) {
        // the kernel reads the pinned inputs once into LDS and writes the results to pinned memory: no copies
        rrf_kernel<<<1, CAND_THREADS, 0, st>>>(reinterpret_cast<long long*>(hin), reinterpret_cast<int*>(hin + ids_b),
                                               (int)T, k_const, reinterpret_cast<long long*>(hout + 8),
                                               reinterpret_cast<double*>(hout + 8 + ids_b), reinterpret_cast<int*>(hout));
        RAG_HIP(hipGetLastError());
    } else {
        RAG_HIP(hipMemcpyAsync(din, hin, in_b, hipMemcpyHostToDevice, st));
        static bool attr_set = false;  // under ws.mu
        if (!attr_set) {
            RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rrf_sort_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, RRF_HASH_SLOTS * 4));
            attr_set = true;
        }
        rrf_sort_kernel<<<1, CAND_THREADS, RRF_HASH_SLOTS * 4, st>>>(
            reinterpret_cast<long long*>(din), reinterpret_cast<int*>(din + ids_b), (int)T, k_const,
            reinterpret_cast<long long*>(dout + 8), reinterpret_cast<double*>(dout + 8 + ids_b), reinterpret_cast<int*>(dout),
            reinterpret_cast<int*>(dfirst));
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipMemcpyAsync(hout, dout, out_b, hipMemcpyDeviceToHost, st));
    }
    RAG_HIP(hipStreamSynchronize(st));
    const int n = *reinterpret_cast<const int*>(hout);
    std::memcpy(out_ids, hout + 8, (size_t)n * sizeof(long long));
    std::memcpy(out_scores, hout + 8 + ids_b, (size_t)n * sizeof(double));
    *n_out = n;
    return RAG_OK;
}

// Shared by rag_mmr (cand: C x d host rows) and rag_flat_mmr (rows of a flat index' arena).
static int mmr_call(int C, int d, const float* cand, const long long* rows, const void* arena, int ld, int bf16,
                    const float* q, double lambda, int top_k, const int64_t* byte_len, int64_t max_bytes,
                    int32_t* out_order, int* out_n) {
    hipStream_t st = lib_stream();
    CandWs& ws = cand_ws();
    std::lock_guard<std::mutex> lk(ws.mu);
    const bool stored = rows != nullptr;
    const size_t cand_b = stored ? al8((size_t)C * sizeof(long long)) : al8((size_t)C * d * sizeof(float));
    const size_t q_b = al8((size_t)d * sizeof(float));
    const size_t bytes_b = (size_t)C * sizeof(long long);
    const size_t in_b = cand_b + q_b + bytes_b;
    const size_t out_b = 8 + al8((size_t)C * sizeof(int));
    const size_t nrm_b = al8((size_t)C * d * sizeof(float)), G_b = al8((size_t)C * C * sizeof(float)),
                 simq_b = al8((size_t)C * sizeof(float));
    int rc = ws.reserve(in_b + out_b + nrm_b + G_b + simq_b, in_b + out_b);
    if (rc) return rc;
    char* hin = ws.pin;
    char* hout = ws.pin + in_b;
    if (stored) std::memcpy(hin, rows, (size_t)C * sizeof(long long));
    else std::memcpy(hin, cand, (size_t)C * d * sizeof(float));
    std::memcpy(hin + cand_b, q, (size_t)d * sizeof(float));
    std::memcpy(hin + cand_b + q_b, byte_len, bytes_b);
    char* din = ws.dev;
    char* dout = din + in_b;
    char* dnrm = dout + out_b;
    char* dG = dnrm + nrm_b;
    char* dsimq = dG + G_b;
    // small inputs (tens of candidates: every call the search engine makes) are read by the kernel straight
    // from pinned memory -- each value once or twice, coalesced -- and the selection goes back the same way
    const bool zero_copy = in_b <= ((size_t)256 << 10);
    char* kin = zero_copy ? hin : din;
    char* kout = zero_copy ? hout : dout;
    if (!zero_copy) RAG_HIP(hipMemcpyAsync(din, hin, in_b, hipMemcpyHostToDevice, st));
    const size_t lds_b = ((size_t)C * d + d) * sizeof(float);
    const bool lrows = C <= MMR_SMALL_C && lds_b <= MMR_LDS_BYTES;
    auto launch = [&](auto fn, size_t smem) {
        if (smem > 0) {
            static bool attr_done[4] = {false, false, false, false};  // under ws.mu
            const int slot = (stored ? 2 : 0) + 1;
            if (!attr_done[slot]) {
                RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)MMR_LDS_BYTES));
                attr_done[slot] = true;
            }
        }
        fn<<<1, CAND_THREADS, smem, st>>>(stored ? nullptr : reinterpret_cast<float*>(kin), reinterpret_cast<float*>(kin + cand_b), C, d,
                                          lambda, top_k, reinterpret_cast<long long*>(kin + cand_b + q_b), (long long)max_bytes,
                                          reinterpret_cast<float*>(dnrm), reinterpret_cast<float*>(dG),
                                          reinterpret_cast<float*>(dsimq), reinterpret_cast<int*>(kout + 8),
                                          reinterpret_cast<int*>(kout), stored ? reinterpret_cast<long long*>(kin) : nullptr,
                                          stored ? arena : nullptr, stored ? ld : 0, stored ? bf16 : 0);
        return RAG_OK;
    };
    if (stored) rc = lrows ? launch(mmr_kernel<true, true>, lds_b) : launch(mmr_kernel<true, false>, 0);
    else rc = lrows ? launch(mmr_kernel<false, true>, lds_b) : launch(mmr_kernel<false, false>, 0);
    if (rc) return rc;
    RAG_HIP(hipGetLastError());
    if (!zero_copy) RAG_HIP(hipMemcpyAsync(hout, dout, out_b, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipStreamSynchronize(st));
    const int n = *reinterpret_cast<const int*>(hout);
    if (n > 0) std::memcpy(out_order, hout + 8, (size_t)n * sizeof(int));
    *out_n = n;
    return RAG_OK;
}

int rag_mmr(int C, int d, const float* cand, const float* q, double lambda, int top_k, const int64_t* byte_len,
            int64_t max_bytes, int32_t* out_order, int* out_n) {
    RAG_REQUIRE_INIT();
    if (C < 0 || d <= 0 || !out_n || top_k < 0) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    *out_n = 0;
    if (C == 0 || top_k == 0) return RAG_OK;
    if (!cand || !q || !byte_len || !out_order) return fail(RAG_ERR_INVALID_ARG, "NULL buffer");
    if (C > MMR_MAX_C)
        return fail(RAG_ERR_UNSUPPORTED, "%d candidates exceed the candidate-set limit %d", C, MMR_MAX_C);
    return mmr_call(C, d, cand, nullptr, nullptr, 0, 0, q, lambda, top_k, byte_len, max_bytes, out_order, out_n);
}

int rag_flat_mmr(rag_flat_t* h, int C, const int64_t* rows, const float* q, double lambda, int top_k,
                 const int64_t* byte_len, int64_t max_bytes, int32_t* out_order, int* out_n) {
    RAG_REQUIRE_INIT();
    if (!h || C < 0 || !out_n || top_k < 0) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    *out_n = 0;
    if (C == 0 || top_k == 0) return RAG_OK;
    if (!rows || !q || !byte_len || !out_order) return fail(RAG_ERR_INVALID_ARG, "NULL buffer");
    if (C > MMR_MAX_C)
        return fail(RAG_ERR_UNSUPPORTED, "%d candidates exceed the candidate-set limit %d", C, MMR_MAX_C);
    // the whole selection runs under the index' mutex (rag::flat_with_view): the kernel gathers the candidate
    // rows from the arena, which a concurrent rag_flat_add could otherwise reallocate between the view and the
    // launch
    struct Ctx {
        int C;
        const int64_t* rows;
        const float* q;
        double lambda;
        int top_k;
        const int64_t* byte_len;
        int64_t max_bytes;
        int32_t* out_order;
        int* out_n;
    } ctx{C, rows, q, lambda, top_k, byte_len, max_bytes, out_order, out_n};
    return flat_with_view(h, [](const FlatView& v, void* p) -> int {
        const Ctx& c = *static_cast<const Ctx*>(p);
        for (int i = 0; i < c.C; ++i)
            if (c.rows[i] < 0 || c.rows[i] >= v.n_rows)
                return fail(RAG_ERR_INVALID_ARG, "row %lld outside [0,%lld)", (long long)c.rows[i], (long long)v.n_rows);
        return mmr_call(c.C, v.d, nullptr, reinterpret_cast<const long long*>(c.rows), v.arena, v.ld, v.bf16, c.q,
                        c.lambda, c.top_k, c.byte_len, c.max_bytes, c.out_order, c.out_n);
    }, &ctx);
}


int rag_hybrid_rrf_mmr(rag_csc_t* tfidf, rag_flat_t* flat, int n_dense, const int64_t* dense_docs, int n_bm25,
                       const int64_t* bm25_docs, double k_const, int top_k, double threshold, int use_knee,
                       const float* q, double lambda, int64_t max_bytes, int* n_fused, int64_t* fused_docs,
                       double* fused_scores, int* n_kept, int* mmr_done, int* n_sel, int32_t* order) {
    RAG_REQUIRE_INIT();
    if (!tfidf || n_dense < 0 || n_bm25 < 0 || (n_dense > 0 && !dense_docs) || (n_bm25 > 0 && !bm25_docs) || top_k < 1 ||
        !n_fused || !fused_docs || !fused_scores || !n_kept || !mmr_done || !n_sel || !order || (flat && !q))
        return fail(RAG_ERR_INVALID_ARG, "rag_hybrid_rrf_mmr: bad argument");
    *n_fused = *n_kept = *mmr_done = *n_sel = 0;
    if (n_dense > HYB_MAX_LIST || n_bm25 > HYB_MAX_LIST || top_k > HYB_MAX_TOPK)
        return fail(RAG_ERR_UNSUPPORTED, "rag_hybrid_rrf_mmr: rankings of %d / %d entries or top_k=%d beyond %d / %d / %d",
                    n_dense, n_bm25, top_k, HYB_MAX_LIST, HYB_MAX_LIST, HYB_MAX_TOPK);
    struct Ctx {
        HybArgs a;
        std::vector<long long> a_doc;  // the caller's code of candidate a
        int rows[HYB_MAX_A];
        int n_rows_q;
        rag_flat_t* flat;
        const float* q;
        const CscBegunView* v;
        bool* consumed;
        int* n_fused; int64_t* fused_docs; double* fused_scores; int* n_kept; int* mmr_done; int* n_sel; int32_t* order;
    } c;
    std::memset(&c.a, 0, sizeof c.a);
    c.n_rows_q = 0;
    // A: distinct documents in first-appearance order (dense ranking, then BM25 ranking)
    auto index_of = [&](long long doc) -> int {
        for (size_t i = 0; i < c.a_doc.size(); ++i)
            if (c.a_doc[i] == doc) return (int)i;
        c.a_doc.push_back(doc);
        return (int)c.a_doc.size() - 1;
    };
    for (int j = 0; j < n_dense; ++j) c.a.dense_a[j] = (unsigned short)index_of(dense_docs[j]);
    for (int j = 0; j < n_bm25; ++j) c.a.bm25_a[j] = (unsigned short)index_of(bm25_docs[j]);
    c.a.nA = (int)c.a_doc.size();
    for (int i = 0; i < c.a.nA; ++i) {
        const long long doc = c.a_doc[(size_t)i];
        if (doc >= 0) {
            if (doc > INT_MAX) return fail(RAG_ERR_INVALID_ARG, "rag_hybrid_rrf_mmr: document row %lld", doc);
            c.a.a_row[i] = (int)doc;
            c.a.a_rk[i] = (short)c.n_rows_q;
            c.rows[c.n_rows_q++] = (int)doc;
        } else {
            c.a.a_row[i] = -1;
            c.a.a_rk[i] = -1;
        }
    }
    c.a.n_dense = n_dense;
    c.a.n_bm25 = n_bm25;
    c.a.top_k = top_k;
    c.a.use_knee = use_knee;
    c.a.k_const = k_const;
    c.a.threshold = threshold;
    c.a.lambda = lambda;
    c.a.max_bytes = max_bytes;
    c.flat = flat;
    c.q = q;
    c.n_fused = n_fused; c.fused_docs = fused_docs; c.fused_scores = fused_scores; c.n_kept = n_kept;
    c.mmr_done = mmr_done; c.n_sel = n_sel; c.order = order;
    bool consumed = false;
    c.consumed = &consumed;

    // innermost step: launch on the TF-IDF handle's stream, synchronise, hand the results over
    auto run = [](Ctx& c, const void* arena) -> int {
        const CscBegunView& v = *c.v;
        CandWs& ws = cand_ws();
        std::lock_guard<std::mutex> lk(ws.mu);
        const size_t q_b = al8((size_t)std::max(c.a.d, 1) * sizeof(float));
        int rc = ws.reserve(64, q_b + al8(sizeof(HybOut)) + sizeof(HybArgs));
        if (rc) return rc;
        float* hq = reinterpret_cast<float*>(ws.pin);
        HybOut* ho = reinterpret_cast<HybOut*>(ws.pin + q_b);
        HybArgs* ha = reinterpret_cast<HybArgs*>(ws.pin + q_b + al8(sizeof(HybOut)));
        if (c.a.do_mmr) std::memcpy(hq, c.q, (size_t)c.a.d * sizeof(float));
        ho->status = -1;
        size_t smem = c.a.do_mmr ? ((size_t)c.a.top_k * c.a.d + c.a.d) * sizeof(float) : 0;
        static size_t dyn_max = 0;  // under ws.mu: what the CU's 160 KiB of LDS leave beside the kernel's static arrays
        if (smem > 0 && dyn_max == 0) {
            hipFuncAttributes fa;
            RAG_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(hybrid_tail_kernel)));
            dyn_max = std::min<size_t>(MMR_LDS_BYTES, ((size_t)160 << 10) - fa.sharedSizeBytes) & ~(size_t)255;
            RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hybrid_tail_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_max));
        }
        if (smem > dyn_max) {       // rows too long for the LDS image: fusion only, MMR left to the caller
            c.a.do_mmr = 0;
            smem = 0;
        }
        *ha = c.a;
        hybrid_tail_kernel<<<1, CAND_THREADS, smem, v.st>>>(ha, v.head_I, v.overflow_flag, v.ranks, v.doc_row, v.doc_bytes,
                                                           hq, arena, ho);
        RAG_HIP(hipGetLastError());
        RAG_HIP(hipStreamSynchronize(v.st));
        if (ho->status == 1)
            return fail(RAG_ERR_UNSUPPORTED, "rag_hybrid_rrf_mmr: the TF-IDF selection overflowed (the query stays begun: "
                                             "rag_tfidf_query_end serves it)");
        if (ho->status != 0) return fail(RAG_ERR_HIP, "rag_hybrid_rrf_mmr: the kernel left no result");
        *c.n_fused = ho->n_fused;
        for (int i = 0; i < ho->n_fused; ++i) {
            c.fused_docs[i] = ho->fa[i] >= 0 ? c.a_doc[(size_t)ho->fa[i]] : ho->frow[i];
            c.fused_scores[i] = ho->fscore[i];
        }
        *c.n_kept = ho->n_kept;
        *c.mmr_done = ho->mmr_done;
        *c.n_sel = ho->mmr_done ? ho->n_sel : 0;
        for (int i = 0; i < *c.n_sel; ++i) c.order[i] = ho->order[i];
        *c.consumed = true;
        return RAG_OK;
    };
    static int (*run_fn)(Ctx&, const void*) = run;

    return csc_with_begun(tfidf, c.n_rows_q, c.rows, [](const CscBegunView& v, void* p) -> int {
        Ctx& c = *static_cast<Ctx*>(p);
        c.v = &v;
        c.a.n_head = (int)std::min<long long>(std::min<long long>(v.begun_k, v.n_rows),
                                              std::min<long long>((long long)c.a.top_k + c.a.nA, HYB_MAX_HEAD));
        // the head must hold the first top_k + |A| documents of the ranking (or all of them)
        if (c.a.n_head < std::min<long long>((long long)c.a.top_k + c.a.nA, v.n_rows))
            return fail(RAG_ERR_UNSUPPORTED, "rag_hybrid_rrf_mmr: the begun query selected %d documents, %d are needed "
                                             "(top_k + candidates)", v.begun_k, c.a.top_k + c.a.nA);
        if (!c.flat) {
            c.a.do_mmr = 0;
            return run_fn(c, nullptr);
        }
        return flat_with_view(c.flat, [](const FlatView& fv, void* p2) -> int {
            Ctx& c = *static_cast<Ctx*>(p2);
            c.a.d = fv.d;
            c.a.ld = fv.ld;
            c.a.bf16 = fv.bf16;
            c.a.n_arena = fv.n_rows;
            // the rows of top_k candidates and the query must fit the kernel's LDS image; else fusion only
            c.a.do_mmr = ((size_t)c.a.top_k * fv.d + fv.d) * sizeof(float) <= MMR_LDS_BYTES ? 1 : 0;
            return run_fn(c, fv.arena);
        }, &c);
    }, &c, &consumed);
}

}  // extern "C"
