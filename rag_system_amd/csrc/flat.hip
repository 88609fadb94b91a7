// flat.hip -- exact dense retrieval: IndexFlatIP / IndexFlatL2 scan fused with top-k.
//
// Replaces faiss' IndexFlat.search (vector_storage.py:196) and .add (:105,:152).
//
// HBM layout: one arena per index, row-major, row stride `ld` elements (d rounded up to a
// multiple of 4 floats / 8 bfloat16 so every row starts 16-byte aligned; pad columns are zero).
// RAG_DTYPE_BF16: rows are rounded to bfloat16 (round-to-nearest-even) when added and the
// QUERY is rounded the same way at search time, in every kernel; products of two bfloat16
// are exact in fp32 and accumulation is fp32, so the scores are those of the rounded
// operands to fp32 accuracy whatever the batch size.
//
// Small query batches (nq <= 2 per pass): flat_scan_kernel -- a bandwidth-bound fused GEMV.
//   * every wave streams tiles of R consecutive rows with 16-byte coalesced loads
//     (lane l reads float4 l, l+64, ... of the row: 1 KiB contiguous per wave instruction),
//   * the query sits in LDS, partial dot products stay in registers,
//   * R*NQ partial sums are reduced across the wave at once (wave_multi_reduce),
//   * rows that beat the wave's current k-th key are inserted into a register-resident
//     sorted list (WaveTopK); the score matrix is never written,
//   * the four waves of a block merge through LDS and write ONE k-list per block,
//   * flat_merge_kernel (one block per query) merges the per-block lists.
#include "common.h"
#include "flat_mfma.h"
#include "flat_stream.h"
#include "topk.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

// ---------------------------------------------------------------- scratch of the no-allocation entry point
// rag_flat_search_dev_noalloc (callers that cannot allocate: HIP graph capture, tight serving loops)
// keeps its padded queries, query norms and per-block candidate lists here instead of in workspaces
// grown with hipMalloc.  One array per GPU the library runs on, held until the process ends.  The name
// below is a host-side stand-in, not a device address: host code only ever uses the pointer that
// hipGetSymbolAddress returned for the device (g_scratch_ptr, set by rag::flat_init_scratch()).
#define RAG_NOALLOC_MAX_NQ 8
#define RAG_NOALLOC_MAX_K 128
namespace {
constexpr size_t NA_Q_BYTES = (size_t)RAG_NOALLOC_MAX_NQ * 8192 * 4;  // fp32 queries, padded rows (d <= 8192)
constexpr size_t NA_QB_BYTES = NA_Q_BYTES / 2;                        // the same as bfloat16
constexpr size_t NA_QN_BYTES = 256;                                   // ||q||^2 (L2)
constexpr size_t NA_LIST_ENTRIES = (size_t)1 << 20;                   // nq * blocks * k <= 8 * 1024 * 128
constexpr size_t NA_OFF_QB = NA_Q_BYTES;
constexpr size_t NA_OFF_QN = NA_OFF_QB + NA_QB_BYTES;
constexpr size_t NA_OFF_KEY = NA_OFF_QN + NA_QN_BYTES;
constexpr size_t NA_OFF_ID = NA_OFF_KEY + NA_LIST_ENTRIES * 4;
constexpr size_t NA_SCRATCH_BYTES = NA_OFF_ID + NA_LIST_ENTRIES * 4;  // 8.4 MiB
}  // namespace
static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];

namespace {

using namespace rag;

// per device: the address hipGetSymbolAddress gave for `scratch` with that device current
int g_scratch_dev = -1;
unsigned char* g_scratch_ptr = nullptr;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// float -> bfloat16 bits, round-to-nearest-even (NaN stays a quiet NaN); oracle/flat_ref.c
// bf16_rne() is the same function.
__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xFFFF0000u); }

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / 64;
constexpr int MERGE_THREADS = 1024;
constexpr int MERGE_WAVES = MERGE_THREADS / 64;
constexpr int SELECT_SORT_MAX = 2048;  // filter_select_kernel sorts up to this many candidates in LDS

// ---------------------------------------------------------------- scan kernel

template <int METRIC>
__device__ __forceinline__ float dot4_acc(f32x4 x, f32x4 q, float acc) {
    if (METRIC == RAG_METRIC_IP) {
        acc = fmaf(x.x, q.x, acc);
        acc = fmaf(x.y, q.y, acc);
        acc = fmaf(x.z, q.z, acc);
        acc = fmaf(x.w, q.w, acc);
    } else {
        float t;
        t = x.x - q.x; acc = fmaf(t, t, acc);
        t = x.y - q.y; acc = fmaf(t, t, acc);
        t = x.z - q.z; acc = fmaf(t, t, acc);
        t = x.w - q.w; acc = fmaf(t, t, acc);
    }
    return acc;
}

// 8 bfloat16 (one 16-byte chunk) against 8 fp32 query values, in element order.
template <int METRIC>
__device__ __forceinline__ float dot8_acc(u32x4 x, f32x4 qa, f32x4 qb, float acc) {
    f32x4 lo, hi;
    lo.x = bf16_lo(x.x); lo.y = bf16_hi(x.x); lo.z = bf16_lo(x.y); lo.w = bf16_hi(x.y);
    hi.x = bf16_lo(x.z); hi.y = bf16_hi(x.z); hi.z = bf16_lo(x.w); hi.w = bf16_hi(x.w);
    acc = dot4_acc<METRIC>(lo, qa, acc);
    return dot4_acc<METRIC>(hi, qb, acc);
}

// d4 = 16-byte chunks per stored row (ld/4 for fp32 rows, ld/8 for bfloat16 rows).  Q4: NQ
// query rows of ld fp32 values (already rounded to bfloat16 values for a bfloat16 index).
template <int R, int NQ, int KS, int METRIC, bool NT, bool BF>
__global__ __launch_bounds__(SCAN_THREADS) void flat_scan_kernel(
    const f32x4* __restrict__ X4, long long n, int d4, const f32x4* __restrict__ Q4, int k,
    float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ floor_key,
    const int* __restrict__ floor_id) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    f32x4* q_lds = reinterpret_cast<f32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    if (!BF) {
        for (int i = tid; i < NQ * d4; i += SCAN_THREADS) q_lds[i] = Q4[i];
    } else {
        // a 16-byte chunk of 8 bfloat16 meets two float4 of the query: keep the even float4s
        // of a query row in [0, d4) and the odd ones in [d4, 2*d4) so both reads are lane-linear
        for (int i = tid; i < NQ * 2 * d4; i += SCAN_THREADS) {
            const int q = i / (2 * d4), j = i - q * 2 * d4;
            q_lds[q * 2 * d4 + (j & 1) * d4 + (j >> 1)] = Q4[i];
        }
    }
    __syncthreads();

    using TK = WaveTopK<float, int, KS>;
    TK tk[NQ];
    float thr_k[NQ];
    int thr_i[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        tk[q].init();
        thr_k[q] = Lowest<float>::v();
        thr_i[q] = IdPad<int>::v();
    }
    // k > 256 is served in passes: a later pass only admits rows that rank strictly behind the
    // last entry of the previous pass (the "floor") of the same query.
    const bool has_floor = floor_key != nullptr;
    float fl_k[NQ];
    int fl_i[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        fl_k[q] = has_floor ? floor_key[q] : 0.f;
        fl_i[q] = has_floor ? floor_id[q] : 0;
    }

    constexpr int N = R * NQ;
    constexpr int LOGN = Log2<N>::v;
    constexpr int REP = 64 / N;
    const long long W = (long long)gridDim.x * SCAN_WAVES;
    const long long gw = (long long)blockIdx.x * SCAN_WAVES + wid;
    const long long ntiles = (n + R - 1) / R;
    const int nchunk = (d4 + 63) >> 6;

    for (long long t = gw; t < ntiles; t += W) {
        const long long row0 = t * R;
        float acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.0f;

        for (int c = 0; c < nchunk; ++c) {
            const int idx = c * 64 + lane;
            if (idx < d4) {
                f32x4 xv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    long long row = row0 + r;
                    row = row < n ? row : n - 1;
                    const f32x4* p = X4 + row * d4 + idx;
                    xv[r] = NT ? __builtin_nontemporal_load(p) : *p;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!BF) {
                        const f32x4 qv = q_lds[q * d4 + idx];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[q * R + r] = dot4_acc<METRIC>(xv[r], qv, acc[q * R + r]);
                    } else {
                        const f32x4 qa = q_lds[q * 2 * d4 + idx];
                        const f32x4 qb = q_lds[q * 2 * d4 + d4 + idx];
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[q * R + r] = dot8_acc<METRIC>(__builtin_bit_cast(u32x4, xv[r]), qa, qb, acc[q * R + r]);
                    }
                }
            }
        }

        const float tot = wave_multi_reduce<N>(acc, lane);
        const int item = (lane >> (6 - LOGN)) & (N - 1);
        const bool rep = (lane & (REP - 1)) == 0;
        const int r_of = item % R, q_of = item / R;
        const long long row = row0 + r_of;
        const float key = (METRIC == RAG_METRIC_IP) ? tot : -tot;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            bool ok = rep && q_of == q && row < n;
            if (has_floor) ok = ok && TK::ahead(fl_k[q], fl_i[q], key, (int)row);
            wave_offer<float, int, KS>(tk[q], k, thr_k[q], thr_i[q], key, (int)row, ok, lane);
        }
    }

    // ---- merge the block's four wave lists through LDS, wave 0 writes one list per query
    const int qbytes = NQ * d4 * 16 * (BF ? 2 : 1);
    float* mk = reinterpret_cast<float*>(smem_raw + qbytes);
    int* mi = reinterpret_cast<int*>(mk + (SCAN_WAVES - 1) * NQ * KS * 64);
    if (wid > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int o = (((wid - 1) * NQ + q) * KS + s) * 64 + lane;
                mk[o] = tk[q].key[s];
                mi[o] = tk[q].id[s];
            }
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            for (int w = 0; w < SCAN_WAVES - 1; ++w)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int o = ((w * NQ + q) * KS + s) * 64 + lane;
                    wave_offer<float, int, KS>(tk[q], k, thr_k[q], thr_i[q], mk[o], mi[o], true, lane);
                }
            const long long base = ((long long)q * gridDim.x + blockIdx.x) * k;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s * 64 + lane;
                if (p < k) {
                    cand_key[base + p] = tk[q].key[s];
                    cand_id[base + p] = tk[q].id[s];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- merge kernels

// Internal candidates [nq][nlists][k] (key = larger-is-better, id = local row, pads INT_MAX).
template <int KS, int METRIC>
__global__ __launch_bounds__(MERGE_THREADS) void flat_merge_kernel(
    const float* __restrict__ cand_key, const int* __restrict__ cand_id, int nlists, int k,
    float* __restrict__ D, long long* __restrict__ I, long long id_base, int out_stride, int out_off,
    const int* __restrict__ only_if) {
    __shared__ float mk[(MERGE_WAVES - 1) * KS * 64];
    __shared__ int mi[(MERGE_WAVES - 1) * KS * 64];
    const long long q = blockIdx.x;
    if (only_if != nullptr && only_if[q] == 0) return;  // redo pass: only the queries that failed
    const float* ck = cand_key + q * (long long)nlists * k;
    const int* ci = cand_id + q * (long long)nlists * k;
    // Fast path (up to 4096 sorted lists): the k-th largest list HEAD T bounds the result from below
    // -- k heads are >= T, so every entry of the global top-k is >= T -- and the entries >= T are a
    // prefix of the few lists whose head is >= T: typically k..3k candidates out of nlists*k.  Both
    // steps are LDS bitonic sorts (heads: 55-78 compare-exchange steps; candidates: a few), ~9 us for
    // 1024 lists whatever k is, against 24 us (k = 10) to 195 us (k = 100) of list merging below,
    // which remains the path for more lists or more than 2048 candidates (long tie runs).
    if (nlists <= 4096) {
        __shared__ unsigned long long sv[SELECT_SORT_MAX];
        __shared__ int n_cand;
        float* hv = reinterpret_cast<float*>(sv);  // 4096 floats = the same 16 KiB
        int N = 64;
        while (N < nlists) N <<= 1;
        for (int i = threadIdx.x; i < N; i += MERGE_THREADS) hv[i] = i < nlists ? ck[(long long)i * k] : -FLT_MAX;
        if (threadIdx.x == 0) n_cand = 0;
        __syncthreads();
        for (int size = 2; size <= N; size <<= 1) {
            for (int st = size >> 1; st > 0; st >>= 1) {
                for (int p = threadIdx.x; p < N / 2; p += MERGE_THREADS) {
                    const int i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
                    const int j = i + st;
                    const bool desc = (i & size) == 0;
                    const float a = hv[i], b = hv[j];
                    if (desc ? a < b : a > b) {
                        hv[i] = b;
                        hv[j] = a;
                    }
                }
                __syncthreads();
            }
        }
        const float T = k <= N ? hv[k - 1] : -FLT_MAX;
        __syncthreads();  // everyone has T before the buffer is reused
        for (int l = threadIdx.x; l < nlists; l += MERGE_THREADS) {
            const float* lk = ck + (long long)l * k;
            const int* li = ci + (long long)l * k;
            // four entries per round trip (independent loads): the walk down a list is a chain of
            // dependent L2 reads otherwise, and the deepest list sets the kernel's time
            bool more = true;
            for (int p0 = 0; p0 < k && more; p0 += 4) {
                float key4[4];
                int id4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int p = p0 + e < k ? p0 + e : k - 1;
                    key4[e] = lk[p];
                    id4[e] = li[p];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!more || p0 + e >= k) break;
                    const float key = key4[e];
                    const int id = id4[e];
                    if (!(key >= T) || id == IdPad<int>::v()) {  // sorted: the rest is below T (or padding)
                        more = false;
                        break;
                    }
                    const int slot = atomicAdd(&n_cand, 1);
                    if (slot < SELECT_SORT_MAX) {
                        unsigned int u = __float_as_uint(key);
                        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                        sv[slot] = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)id);
                    }
                }
            }
        }
        __syncthreads();
        const int cnt = n_cand;
        if (cnt <= SELECT_SORT_MAX) {  // uniform
            int M = 64;
            while (M < cnt) M <<= 1;
            for (int i = cnt + threadIdx.x; i < M; i += MERGE_THREADS) sv[i] = 0ULL;
            __syncthreads();
            for (int size = 2; size <= M; size <<= 1) {
                for (int st = size >> 1; st > 0; st >>= 1) {
                    for (int p = threadIdx.x; p < M / 2; p += MERGE_THREADS) {
                        const int i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
                        const int j = i + st;
                        const bool desc = (i & size) == 0;
                        const unsigned long long a = sv[i], b = sv[j];
                        if (desc ? a < b : a > b) {
                            sv[i] = b;
                            sv[j] = a;
                        }
                    }
                    __syncthreads();
                }
            }
            for (int p = threadIdx.x; p < k; p += MERGE_THREADS) {
                const long long o = q * out_stride + out_off + p;
                if (p < cnt) {
                    const unsigned long long c = sv[p];
                    unsigned int u = (unsigned int)(c >> 32);
                    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
                    const float key = __uint_as_float(u);
                    D[o] = METRIC == RAG_METRIC_IP ? key : -key;
                    I[o] = id_base + (long long)(0xFFFFFFFFu - (unsigned int)(c & 0xFFFFFFFFULL));
                } else {
                    D[o] = METRIC == RAG_METRIC_IP ? -FLT_MAX : FLT_MAX;
                    I[o] = -1LL;
                }
            }
            return;
        }
        __syncthreads();
    }
    WaveTopK<float, int, KS> tk;
    if (KS == 1 && k <= 32) {
        // short lists: several lists share a 64-candidate chunk (flat walk, sorted inserts)
        block_merge<float, int, KS, MERGE_WAVES>(
            [&](long long c, float& key, int& id) {
                key = ck[c];
                id = ci[c];
                return id != IdPad<int>::v();
            },
            (long long)nlists * k, k, tk, mk, mi);
    } else {
        // a whole sorted list per chunk: bitonic merge or a few inserts, never a sort
        block_merge_lists<float, int, KS, MERGE_WAVES>(
            [&](int l, int p, float& key, int& id) {
                const long long o = (long long)l * k + p;
                key = ck[o];
                id = ci[o];
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
                const float key = tk.key[s];
                const long long o = q * out_stride + out_off + p;
                D[o] = pad ? (METRIC == RAG_METRIC_IP ? -FLT_MAX : FLT_MAX) : (METRIC == RAG_METRIC_IP ? key : -key);
                I[o] = pad ? -1LL : id_base + (long long)tk.id[s];
            }
        }
    }
}

// External per-shard results [nlists][nq][k] (faiss-format D / int64 I, pads I = -1).
template <int KS>
__global__ __launch_bounds__(MERGE_THREADS) void ext_merge_kernel(
    int metric, const float* __restrict__ Dl, const long long* __restrict__ Il, int nlists, long long nq,
    int k, float* __restrict__ D, long long* __restrict__ I, long long stride_D, long long stride_I) {
    __shared__ float mk[(MERGE_WAVES - 1) * KS * 64];
    __shared__ long long mi[(MERGE_WAVES - 1) * KS * 64];
    const long long q = blockIdx.x;
    const bool ip = metric == RAG_METRIC_IP;
    WaveTopK<float, long long, KS> tk;
    if (KS == 1 && k <= 32) {
        block_merge<float, long long, KS, MERGE_WAVES>(
            [&](long long c, float& key, long long& id) {
                const long long l = c / k, j = c - l * k;
                id = Il[l * stride_I + q * k + j];
                const float v = Dl[l * stride_D + q * k + j];
                key = ip ? v : -v;
                return id >= 0;
            },
            (long long)nlists * k, k, tk, mk, mi);
    } else {
        block_merge_lists<float, long long, KS, MERGE_WAVES>(
            [&](int l, int p, float& key, long long& id) {
                id = Il[(long long)l * stride_I + q * k + p];
                const float v = Dl[(long long)l * stride_D + q * k + p];
                key = ip ? v : -v;
                return id >= 0;
            },
            nlists, k, tk, mk, mi);
    }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int p = s * 64 + lane;
            if (p < k) {
                const bool pad = tk.id[s] == IdPad<long long>::v();
                const float key = tk.key[s];
                D[q * k + p] = pad ? (ip ? -FLT_MAX : FLT_MAX) : (ip ? key : -key);
                I[q * k + p] = pad ? -1LL : tk.id[s];
            }
        }
    }
}

// Filter mode of the MFMA path: top-k of one query's append buffer (unsorted; every row scoring
// >= the query's threshold).  count in [k, cap] -> the buffer holds the exact global top-k;
// otherwise the query is flagged for the list-mode redo pass.
template <int KS, int METRIC>
__global__ __launch_bounds__(MERGE_THREADS) void filter_select_kernel(
    const float* __restrict__ cand_key, const int* __restrict__ cand_id, const unsigned int* __restrict__ counts,
    int cap, int k, float* __restrict__ D, long long* __restrict__ I, long long id_base, int* __restrict__ nfail,
    int* __restrict__ fail) {
    __shared__ float mk[(MERGE_WAVES - 1) * KS * 64];
    __shared__ int mi[(MERGE_WAVES - 1) * KS * 64];
    const long long q = blockIdx.x;
    const unsigned int cnt = counts[q];
    const bool ok = cnt >= (unsigned int)k && cnt <= (unsigned int)cap;
    if (threadIdx.x == 0) {
        fail[q] = ok ? 0 : 1;
        if (!ok) atomicAdd(nfail, 1);
    }
    if (!ok) return;
    const float* ck = cand_key + q * cap;
    const int* ci = cand_id + q * cap;
    if (cnt <= (unsigned int)SELECT_SORT_MAX) {
        // Up to 2048 candidates (the usual few hundred): bitonic sort in LDS of 64-bit composites
        // (order-preserving key bits, then ~id, so that descending = key descending, id ascending)
        // -- 45-66 compare-exchange steps, ~8 us, against 15-45 us of sorted inserts.
        __shared__ unsigned long long sv[SELECT_SORT_MAX];
        const int N = cnt <= 1024u ? 1024 : SELECT_SORT_MAX;
        for (int i = threadIdx.x; i < N; i += MERGE_THREADS) {
            unsigned long long c = 0ULL;
            if (i < (int)cnt) {
                unsigned int u = __float_as_uint(ck[i]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                c = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)ci[i]);
            }
            sv[i] = c;
        }
        __syncthreads();
        for (int size = 2; size <= N; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int p = threadIdx.x; p < N / 2; p += MERGE_THREADS) {
                    const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                    const int j = i + stride;
                    const bool desc = (i & size) == 0;
                    const unsigned long long a = sv[i], b = sv[j];
                    if (desc ? a < b : a > b) {
                        sv[i] = b;
                        sv[j] = a;
                    }
                }
                __syncthreads();
            }
        }
        for (int p = threadIdx.x; p < k; p += MERGE_THREADS) {
            const unsigned long long c = sv[p];
            unsigned int u = (unsigned int)(c >> 32);
            u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
            const float key = __uint_as_float(u);
            D[q * k + p] = METRIC == RAG_METRIC_IP ? key : -key;
            I[q * k + p] = id_base + (long long)(0xFFFFFFFFu - (unsigned int)(c & 0xFFFFFFFFULL));
        }
        return;
    }
    WaveTopK<float, int, KS> tk;
    block_merge<float, int, KS, MERGE_WAVES>(
        [&](long long c, float& key, int& id) {
            key = ck[c];
            id = ci[c];
            return true;
        },
        (long long)cnt, k, tk, mk, mi);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int p = s * 64 + lane;
            if (p < k) {
                const float key = tk.key[s];
                D[q * k + p] = METRIC == RAG_METRIC_IP ? key : -key;
                I[q * k + p] = id_base + (long long)tk.id[s];
            }
        }
    }
}

// Threshold of the filter flows whose sample pass leaves ONE best score per wave or block and
// query (vals[q*stride + i], i < count <= 4096; -FLT_MAX = nothing seen): thr[q] = the j-th largest,
// by a bitonic sort in LDS.  Fewer than j real values -> -FLT_MAX (every row passes, the query
// overflows its buffer and is redone in list mode).
__global__ __launch_bounds__(MERGE_THREADS) void kth_largest_kernel(const float* __restrict__ vals, int stride,
                                                                    int count, int j, float* __restrict__ thr) {
    __shared__ float v[4096];
    const float* src = vals + (long long)blockIdx.x * stride;
    count = count < 4096 ? count : 4096;  // more values than fit: the rest could only raise the threshold
    int N = 64;
    while (N < count) N <<= 1;
    for (int i = threadIdx.x; i < N; i += MERGE_THREADS) v[i] = i < count ? src[i] : -FLT_MAX;
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1) {
        for (int st = size >> 1; st > 0; st >>= 1) {
            for (int p = threadIdx.x; p < N / 2; p += MERGE_THREADS) {
                const int i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
                const int jx = i + st;
                const bool desc = (i & size) == 0;
                const float a = v[i], b = v[jx];
                if (desc ? a < b : a > b) {
                    v[i] = b;
                    v[jx] = a;
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) thr[blockIdx.x] = j <= N ? v[j - 1] : -FLT_MAX;
}

// floor of pass p+1 = last entry of pass p (position `pos` of each query's output row).
__global__ void make_floor_kernel(const float* __restrict__ D, const long long* __restrict__ I, long long nq,
                                  int stride, int pos, int metric, long long id_base, float* __restrict__ fk,
                                  int* __restrict__ fi) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) {
        const long long id = I[q * stride + pos];
        const float d = D[q * stride + pos];
        fk[q] = id < 0 ? Lowest<float>::v() : (metric == RAG_METRIC_IP ? d : -d);
        fi[q] = id < 0 ? IdPad<int>::v() : (int)(id - id_base);
    }
}

__global__ void fill_pad_kernel(float* D, long long* I, long long total, float dpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        D[i] = dpad;
        I[i] = -1;
    }
}

// Queries (nq x d) -> zero-padded (nq x ld).  bfloat16 index (qb != nullptr): the fp32 copy
// holds the ROUNDED values and qb the same values as bfloat16 (MFMA operand).
__global__ void pad_queries_kernel(const float* __restrict__ q, float* __restrict__ qpad,
                                   unsigned short* __restrict__ qb, long long nq, int d, int ld) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq * ld) {
        const long long r = i / ld;
        const int c = (int)(i - r * ld);
        float v = c < d ? q[r * d + c] : 0.0f;
        if (qb) {
            const unsigned short b = bf16_rne(v);
            qb[i] = b;
            v = __uint_as_float((unsigned int)b << 16);
        }
        qpad[i] = v;
    }
}

// fp32 rows (n x d, dense) -> bfloat16 arena rows (n x ld, pad columns zero).
__global__ void rows_to_bf16_kernel(const float* __restrict__ src, int d, unsigned short* __restrict__ dst, int ld,
                                    long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * ld) {
        const long long r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = c < d ? bf16_rne(src[r * d + c]) : (unsigned short)0;
    }
}

// bfloat16 arena rows -> fp32 rows (n x d, dense); exact widening.
__global__ void rows_from_bf16_kernel(const unsigned short* __restrict__ src, int ld, float* __restrict__ dst, int d,
                                      long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * d) {
        const long long r = i / d;
        const int c = (int)(i - r * d);
        dst[i] = __uint_as_float((unsigned int)src[r * ld + c] << 16);
    }
}

// ---------------------------------------------------------------- bfloat16 prefilter (rag_flat_prefilter)
//
// An fp32 index may keep a bfloat16 copy of its rows (the "shadow", a second arena of half the size).  A
// search then streams the copy -- half the HBM bytes, and for batches the bf16 matrix cores instead of the
// fp32 ones -- for kp > k candidates per query, scores those candidates again from the fp32 rows, and
// PROVES that nothing outside them can belong to the top k:
//   |<bf16(x), bf16(q)>_computed - <x, q>| <= eps * ||x|| * ||q||,  eps = 2u + u^2 + (d + 16) * 2^-23,  u = 2^-8
//   (round-to-nearest of both operands, exact products, d fp32 additions; Cauchy-Schwarz on sum |x_i q_i|),
// so every row outside the candidates scores at most (worst candidate's approximate score) + eps *
// max||x|| * ||q||.  If the k-th best EXACT score among the candidates is above that bound, the candidates'
// exact top k is the index' exact top k; otherwise the query is flagged and redone on the fp32 rows by
// the list-mode tile kernel (the redo path of the filter flow).  The result never depends on bfloat16
// arithmetic, only the amount of work does.

// fp32 arena rows -> shadow rows (round to nearest even), ||x||^2 per row (L2) and the largest of them.
__global__ __launch_bounds__(256) void shadow_rows_kernel(const float* __restrict__ X, int ld, int d, long long n,
                                                          unsigned short* __restrict__ S, int sld,
                                                          float* __restrict__ xn_out, unsigned int* __restrict__ max_bits) {
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * 4;
    float mx = 0.f;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += W) {
        const float* row = X + r * ld;
        unsigned short* out = S + r * sld;
        double ss = 0.0;
        for (int c = lane; c < sld; c += 64) {
            const float v = c < d ? row[c] : 0.0f;
            ss += (double)v * (double)v;
            out[c] = bf16_rne(v);
        }
        for (int dist = 32; dist >= 1; dist >>= 1) ss += __shfl_xor(ss, dist);
        float nn = (float)ss;
        if (!(nn >= 0.f)) nn = __builtin_inff();  // NaN in the row: no bound holds, every search is redone exactly
        if (xn_out != nullptr && lane == 0) xn_out[r] = (float)ss;
        mx = fmaxf(mx, nn);
    }
    if (lane == 0 && mx > 0.f) atomicMax(max_bits, __float_as_uint(mx));
}

constexpr int PF_MAX_KP = 128;
constexpr int PF_THREADS = 1024;  // 16 waves x 4 candidates: one round for 64 candidates

// One workgroup per query.  pD / pI: the shadow's kp best rows (approximate scores, local ids, -1 = none).
// Exact scores: float64 accumulation over the fp32 row, rounded once to fp32 (L2: sum of (x - q)^2).
// Writes the exact top k (ties: lower id first) to D / I and flags the query (failq = 1, ++nfail) when
// the bound above does not close.
template <int METRIC>
__global__ __launch_bounds__(PF_THREADS) void prefilter_rescore_kernel(
    const float* __restrict__ X, long long n, int ld, int d, const float* __restrict__ Q,
    const float* __restrict__ pD, const long long* __restrict__ pI, int kp, int k,
    const unsigned int* __restrict__ max_bits, double eps_rel, double slack, float* __restrict__ D,
    long long* __restrict__ I, long long id_base, int* __restrict__ nfail, int* __restrict__ failq) {
    extern __shared__ __align__(16) float s_q[];  // ld floats, pad columns zero
    __shared__ double s_a[PF_MAX_KP];              // approximate score, larger = better (L2: minus the distance)
    __shared__ float s_e[PF_MAX_KP];               // exact score, same sense
    __shared__ int s_id[PF_MAX_KP];
    __shared__ unsigned char s_need[PF_MAX_KP];
    __shared__ double s_red[1][PF_THREADS / 64];
    __shared__ double s_L, s_B;
    __shared__ float s_Ek;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* qrow = Q + (long long)q * d;
    double qq = 0.0;
    for (int c = tid; c < ld; c += PF_THREADS) {
        const float v = c < d ? qrow[c] : 0.0f;
        s_q[c] = v;
        qq += (double)v * (double)v;
    }
    for (int dist = 32; dist >= 1; dist >>= 1) qq += __shfl_xor(qq, dist);
    if (lane == 0) s_red[0][wid] = qq;
    if (tid == 0) {
        s_L = __builtin_nan("");
        s_B = __builtin_inf();
        s_Ek = __builtin_nanf("");
    }
    __syncthreads();
    qq = 0.0;
    for (int w = 0; w < PF_THREADS / 64; ++w) qq += s_red[0][w];
    const double mx2 = (double)__uint_as_float(*max_bits);
    double eps = eps_rel * sqrt(mx2 * qq);
    if (METRIC == RAG_METRIC_L2) eps = 2.0 * eps + slack * (mx2 + qq);  // + the fp32 roundings of both norms and of the key
    eps = eps * (1.0 + 1e-6) + 1e-30 * (1.0 + sqrt(qq));

    bool valid = false;
    double a = -__builtin_inf();
    if (tid < kp) {
        const long long id = pI[(long long)q * kp + tid];
        valid = id >= 0 && id < n;
        const double v = (double)pD[(long long)q * kp + tid];
        // L2: the shadow's distance is ||x||^2 + ||q||^2 - 2<bf16(x), bf16(q)> with BOTH norms taken from the fp32
        // operands (sync_shadow copies the fp32 rows' norms, prepare_l2_norms squares the fp32 query), so only
        // the cross term carries the rounding and eps (twice the inner-product bound) covers it; no shift
        if (valid) a = METRIC == RAG_METRIC_IP ? v : -v;
        s_a[tid] = a;
        s_id[tid] = valid ? (int)id : -1;
    }
    const int nv = __syncthreads_count(valid);
    const int kk = k < nv ? k : nv;
    if (valid) {
        int ra = 0;
        for (int j = 0; j < kp; ++j) {
            const double b = s_a[j];
            ra += (b > a || (b == a && j < tid)) && s_id[j] >= 0;
        }
        if (ra == kk - 1) s_L = a - eps;
        // rows outside the candidates exist only if the shadow filled all kp places
        if (ra == nv - 1) s_B = (nv == kp && n > (long long)kp) ? a + eps : -__builtin_inf();
    }
    __syncthreads();
    const bool need = valid && (a + eps >= s_L);
    if (tid < kp) s_need[tid] = need;
    const int m = __syncthreads_count(need);
    // sixteen lanes per candidate, four candidates per wave, all 64 of a round at once: every load of a row is
    // independent of the others, so a round costs one trip to HBM (one wave per candidate, one after the other: four)
    const int sub = lane >> 4, l16 = lane & 15;
    for (int i = wid * 4 + sub; i < kp; i += PF_THREADS / 16) {
        double acc = 0.0;
        if (s_need[i]) {
            const f32x4* row = reinterpret_cast<const f32x4*>(X + (long long)s_id[i] * ld);
            const f32x4* qv = reinterpret_cast<const f32x4*>(s_q);
            for (int c = l16; c < ld / 4; c += 16) {
                const f32x4 x = row[c], y = qv[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (METRIC == RAG_METRIC_IP) acc += (double)x[e] * (double)y[e];
                    else {
                        const double df = (double)x[e] - (double)y[e];
                        acc += df * df;
                    }
                }
            }
        }
        for (int dist = 8; dist >= 1; dist >>= 1) acc += __shfl_xor(acc, dist);
        if (l16 == 0 && s_need[i]) s_e[i] = METRIC == RAG_METRIC_IP ? (float)acc : -(float)acc;
    }
    __syncthreads();
    if (need) {
        const float e = s_e[tid];
        const int id = s_id[tid];
        int r = 0;
        for (int j = 0; j < kp; ++j) {
            if (!s_need[j]) continue;
            const float b = s_e[j];
            r += b > e || (b == e && s_id[j] < id);
        }
        if (r < k) {
            D[(long long)q * k + r] = METRIC == RAG_METRIC_IP ? e : -e;
            I[(long long)q * k + r] = (long long)id + id_base;
        }
        if (r == kk - 1) s_Ek = e;
    }
    for (int pos = (m < k ? m : k) + tid; pos < k; pos += PF_THREADS) {
        D[(long long)q * k + pos] = METRIC == RAG_METRIC_IP ? -FLT_MAX : FLT_MAX;
        I[(long long)q * k + pos] = -1;
    }
    __syncthreads();
    if (tid == 0) {
        // NaN anywhere (scores, norms, query) makes one of these comparisons false: the query is redone
        const bool ok = m >= kk && kk >= 1 && (double)s_Ek > s_B;
        failq[q] = ok ? 0 : 1;
        if (!ok) atomicAdd(nfail, 1);
    }
}

// ---------------------------------------------------------------- synthetic rows

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ int synth_int(unsigned long long row_key, int col) {
    const unsigned long long h = splitmix64(row_key + (unsigned long long)col);
    const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) +
                  (int)((h >> 48) & 0xFFFF);
    return s - 131070;
}

// One wave per row; bit-identical to oracle/flat_ref.c:synth_row (integer sum of squares is
// exact, sqrt and the division are IEEE double, one rounding to float).
template <bool BF>
__global__ __launch_bounds__(256) void synth_rows_kernel(void* __restrict__ Xv, int ld, int d, long long n,
                                                         unsigned long long seed, long long first_global) {
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += W) {
        const unsigned long long key =
            splitmix64(seed ^ ((unsigned long long)(first_global + r) * 0xD1B54A32D192ED03ULL));
        unsigned long long ss = 0;
        for (int c = lane; c < d; c += 64) {
            const long long v = synth_int(key, c);
            ss += (unsigned long long)(v * v);
        }
        for (int dist = 32; dist >= 1; dist >>= 1) ss += __shfl_xor(ss, dist);
        const double nrm = sqrt((double)ss);
        for (int c = lane; c < ld; c += 64) {
            const float v = c < d ? (float)((double)synth_int(key, c) / nrm) : 0.0f;
            if (BF) reinterpret_cast<unsigned short*>(Xv)[r * ld + c] = bf16_rne(v);
            else reinterpret_cast<float*>(Xv)[r * ld + c] = v;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------- handle

struct rag_flat {
    int d = 0, ld = 0, metric = 0, dtype = 0;
    int esz = 4;             // bytes per stored element (4 fp32, 2 bfloat16)
    int64_t ntotal = 0;      // rows visible to search (device rows + staged rows)
    int64_t ndev = 0;        // rows resident in HBM
    int64_t cap = 0;         // arena capacity in rows
    char* x = nullptr;       // HBM arena, cap * ld elements of esz bytes
    float* ws_cvt = nullptr; int64_t ws_cvt_cap = 0;   // fp32 staging for bfloat16 conversion
    unsigned short* ws_qb = nullptr; int64_t ws_qb_cap = 0;  // bfloat16 queries (MFMA operand)
    float* ws_qt = nullptr; int64_t ws_qt_cap = 0;           // queries as block shape 10's LDS image
    hipStream_t async_st = nullptr;                           // rag_flat_search_begin / _end: the handle's own stream
    bool begun = false;                                       // ... a begun search owns the pinned buffer until it is ended
    size_t begun_doff = 0, begun_ioff = 0, begun_db = 0, begun_ib = 0;
    unsigned int* ws_ticket = nullptr;                        // arrival counter of the stream kernel's fused merge (zero between launches)
    std::vector<float> staged;  // host rows not yet flushed (d floats each)
    // search workspace (HBM)
    float* ws_q = nullptr;   int64_t ws_q_cap = 0;     // padded queries
    float* ws_key = nullptr; int64_t ws_key_cap = 0;
    int* ws_id = nullptr;    int64_t ws_id_cap = 0;
    float* ws_fk = nullptr;  int64_t ws_fk_cap = 0;    // per-query floors for k > 256 passes
    int* ws_fi = nullptr;    int64_t ws_fi_cap = 0;
    float* ws_qn = nullptr;  int64_t ws_qn_cap = 0;    // ||q||^2 (L2, MFMA path)
    // MFMA filter mode: sample results, thresholds, counters [nq] + nfail [1] + fail flags [nq]
    float* ws_sD = nullptr;  int64_t ws_sD_cap = 0;
    long long* ws_sI = nullptr; int64_t ws_sI_cap = 0;
    float* ws_thr = nullptr; int64_t ws_thr_cap = 0;
    int* ws_cnt = nullptr;   int64_t ws_cnt_cap = 0;
    float* zero16 = nullptr;                           // 64 B of zeros (MFMA tile padding source)
    float* xnorm = nullptr;  int64_t xnorm_cap = 0;    // ||x||^2 per row (L2, MFMA path), lazily filled
    int64_t norm_valid = 0;
    float* ws_hq = nullptr;  int64_t ws_hq_cap = 0;    // host-variant device copies
    float* ws_D = nullptr; long long* ws_I = nullptr; int64_t ws_out_cap = 0;
    char* pin = nullptr; size_t pin_cap = 0;            // pinned staging of the host variant (small batches)
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
    // stream of the last asynchronous search on this handle: the workspaces above are shared by every
    // search, so a search on ANOTHER stream first waits for everything enqueued on this one
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t order_ev = nullptr;
    // rag_flat_prefilter: bfloat16 copy of the rows ("shadow": its own handle, arena and workspaces), the
    // largest ||x||^2 among the copied rows (device word, float bits), the candidates of a search
    rag_flat* shadow = nullptr;
    unsigned int* sh_max = nullptr;
    int* last_nfail = nullptr;     // device word of the last prefiltered search: queries it had to redo (rag_flat_prefilter_redone)
    float* ws_pD = nullptr; int64_t ws_pD_cap = 0;
    long long* ws_pI = nullptr; int64_t ws_pI_cap = 0;
    char last_kernel[128] = "";   // which kernels served the last search (rag_flat_last_kernel)
    std::mutex mu;
};

namespace {

constexpr int64_t STAGE_FLUSH_FLOATS = 16 << 20;  // 64 MiB of staged rows

int ensure_capacity(rag_flat* h, int64_t rows) {
    if (rows <= h->cap) return RAG_OK;
    int64_t want = std::max<int64_t>(rows, std::max<int64_t>(h->cap * 2, 1024));
    char* nx = nullptr;
    const size_t rb = (size_t)h->ld * h->esz;
    hipError_t e = hipMalloc(&nx, (size_t)want * rb);
    if (e != hipSuccess && want > rows) {
        (void)hipGetLastError();
        want = rows;
        e = hipMalloc(&nx, (size_t)want * rb);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(RAG_ERR_OOM, "hipMalloc of %lld rows x %zu bytes failed: %s", (long long)want, rb,
                    hipGetErrorString(e));
    }
    hipStream_t st = lib_stream();
    if (h->ld != h->d)  // pad columns must read as zero
        RAG_HIP(hipMemsetAsync(nx, 0, (size_t)want * rb, st));
    if (h->ndev > 0)
        RAG_HIP(hipMemcpyAsync(nx, h->x, (size_t)h->ndev * rb, hipMemcpyDeviceToDevice, st));
    RAG_HIP(hipStreamSynchronize(st));
    if (h->x) RAG_HIP(hipFree(h->x));
    h->x = nx;
    h->cap = want;
    return RAG_OK;
}

template <typename T>
int grow(T*& p, int64_t& cap, int64_t need);

constexpr int64_t CVT_CHUNK_FLOATS = 32 << 20;  // 128 MiB of fp32 per conversion chunk

// fp32 device rows (n x d dense) -> bfloat16 arena rows starting at row `first`.
int convert_rows_dev(rag_flat* h, const float* src_dev, int64_t first, int64_t n, hipStream_t st) {
    const long long total = (long long)n * h->ld;
    unsigned short* dst = reinterpret_cast<unsigned short*>(h->x) + (size_t)first * h->ld;
    rows_to_bf16_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(src_dev, h->d, dst, h->ld, (long long)n);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

// Copy n host rows (d floats each) to arena rows [ndev, ndev+n).
int upload_rows(rag_flat* h, const float* src, int64_t n) {
    if (n <= 0) return RAG_OK;
    int rc = ensure_capacity(h, h->ndev + n);
    if (rc) return rc;
    if (h->dtype == RAG_DTYPE_BF16) {
        const int64_t rows_per = std::max<int64_t>(1, CVT_CHUNK_FLOATS / h->d);
        hipStream_t st = lib_stream();
        rc = grow(h->ws_cvt, h->ws_cvt_cap, std::min(n, rows_per) * h->d);
        if (rc) return rc;
        for (int64_t r0 = 0; r0 < n; r0 += rows_per) {
            const int64_t m = std::min(rows_per, n - r0);
            RAG_HIP(hipMemcpyAsync(h->ws_cvt, src + r0 * h->d, (size_t)m * h->d * sizeof(float), hipMemcpyHostToDevice, st));
            rc = convert_rows_dev(h, h->ws_cvt, h->ndev + r0, m, st);
            if (rc) return rc;
            RAG_HIP(hipStreamSynchronize(st));
        }
        h->ndev += n;
        return RAG_OK;
    }
    float* dst = reinterpret_cast<float*>(h->x) + (size_t)h->ndev * h->ld;
    if (h->ld == h->d)
        RAG_HIP(hipMemcpy(dst, src, (size_t)n * h->d * sizeof(float), hipMemcpyHostToDevice));
    else
        RAG_HIP(hipMemcpy2D(dst, (size_t)h->ld * sizeof(float), src, (size_t)h->d * sizeof(float),
                            (size_t)h->d * sizeof(float), (size_t)n, hipMemcpyHostToDevice));
    h->ndev += n;
    return RAG_OK;
}

int flush_staged(rag_flat* h) {
    if (h->staged.empty()) return RAG_OK;
    const int64_t n = (int64_t)h->staged.size() / h->d;
    int rc = upload_rows(h, h->staged.data(), n);
    if (rc) return rc;
    h->staged.clear();
    h->staged.shrink_to_fit();
    return RAG_OK;
}

// hipFree waits for all work on the device before it releases memory, so replacing a workspace is
// safe even while an earlier asynchronous search (on any stream) still reads the old one; it costs
// a device-wide wait, once per growth.
template <typename T>
int grow(T*& p, int64_t& cap, int64_t need) {
    if (need <= cap) return RAG_OK;
    if (p) RAG_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    const int64_t want = std::max<int64_t>(need, 1024);
    RAG_HIP(hipMalloc(&p, (size_t)want * sizeof(T)));
    cap = want;
    return RAG_OK;
}

// Event pair around the dominant kernel of a search (rag_flat_profiling).  A pair that was taken
// but never completed -- an early error return between begin() and end() -- goes back to the free
// list instead of leaking.
struct ProfPair {
    rag_flat* h;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool done = false;
    explicit ProfPair(rag_flat* hh) : h(hh) {}
    int begin(hipStream_t st) {
        if (!h->prof) return RAG_OK;
        if (!h->ev_free.empty()) {
            e0 = h->ev_free.back().first;
            e1 = h->ev_free.back().second;
            h->ev_free.pop_back();
        } else {
            RAG_HIP(hipEventCreate(&e0));
            hipError_t e = hipEventCreate(&e1);
            if (e != hipSuccess) {
                (void)hipEventDestroy(e0);
                e0 = nullptr;
                return fail(RAG_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
            }
        }
        RAG_HIP(hipEventRecord(e0, st));
        return RAG_OK;
    }
    int end(hipStream_t st) {
        if (!e0) return RAG_OK;
        RAG_HIP(hipEventRecord(e1, st));
        h->ev_used.emplace_back(e0, e1);
        done = true;
        return RAG_OK;
    }
    ~ProfPair() {
        if (e0 && !done) h->ev_free.emplace_back(e0, e1);
    }
};

struct ScanCfg {
    int R, NQ, KS, metric;
    bool nt, bf;
};

typedef void (*scan_fn_t)(const f32x4*, long long, int, const f32x4*, int, float*, int*, const float*, const int*);

template <int R, int NQ, int KS, bool BF>
scan_fn_t pick_scan3(int metric, bool nt) {
    if (metric == RAG_METRIC_IP)
        return nt ? flat_scan_kernel<R, NQ, KS, RAG_METRIC_IP, true, BF>
                  : flat_scan_kernel<R, NQ, KS, RAG_METRIC_IP, false, BF>;
    return nt ? flat_scan_kernel<R, NQ, KS, RAG_METRIC_L2, true, BF>
              : flat_scan_kernel<R, NQ, KS, RAG_METRIC_L2, false, BF>;
}
template <int R, int NQ, int KS>
scan_fn_t pick_scan2(int metric, bool nt, bool bf) {
    return bf ? pick_scan3<R, NQ, KS, true>(metric, nt) : pick_scan3<R, NQ, KS, false>(metric, nt);
}
template <int R, int NQ>
scan_fn_t pick_scan1(int KS, int metric, bool nt, bool bf) {
    switch (KS) {
        case 1: return pick_scan2<R, NQ, 1>(metric, nt, bf);
        case 2: return pick_scan2<R, NQ, 2>(metric, nt, bf);
        default: return pick_scan2<R, NQ, 4>(metric, nt, bf);
    }
}
scan_fn_t pick_scan(const ScanCfg& c) {
    if (c.NQ == 1) {
        if (c.R == 2) return pick_scan1<2, 1>(c.KS, c.metric, c.nt, c.bf);
        if (c.R == 8) return pick_scan1<8, 1>(c.KS, c.metric, c.nt, c.bf);
        return pick_scan1<4, 1>(c.KS, c.metric, c.nt, c.bf);
    }
    return pick_scan1<4, 2>(c.KS, c.metric, c.nt, c.bf);
}

typedef void (*merge_fn_t)(const float*, const int*, int, int, float*, long long*, long long, int, int, const int*);
merge_fn_t pick_merge(int KS, int metric) {
    if (metric == RAG_METRIC_IP) {
        if (KS == 1) return flat_merge_kernel<1, RAG_METRIC_IP>;
        if (KS == 2) return flat_merge_kernel<2, RAG_METRIC_IP>;
        return flat_merge_kernel<4, RAG_METRIC_IP>;
    }
    if (KS == 1) return flat_merge_kernel<1, RAG_METRIC_L2>;
    if (KS == 2) return flat_merge_kernel<2, RAG_METRIC_L2>;
    return flat_merge_kernel<4, RAG_METRIC_L2>;
}

int ks_for(int k) { return k <= 64 ? 1 : (k <= 128 ? 2 : 4); }

// L2 through the matrix-core kernels is ||x||^2 + ||q||^2 - 2<x,q>: row norms are computed once per
// added row (kept next to the arena), query norms per call.
int ensure_norm_array(rag_flat* h, hipStream_t st);

int prepare_l2_norms(rag_flat* h, const float* qf, int64_t nq, hipStream_t st) {
    if (h->metric != RAG_METRIC_L2) return RAG_OK;
    const int64_t n = h->ndev;
    const int ld = h->ld;
    const bool bf = h->dtype == RAG_DTYPE_BF16;
    int rc = RAG_OK;
    rc = ensure_norm_array(h, st);  // follow the arena
    if (rc) return rc;
    if (h->norm_valid < n) {
        rc = mfma_row_norms(h->x, ld, bf, h->norm_valid, n - h->norm_valid, h->xnorm, st);
        if (rc) return rc;
        h->norm_valid = n;
    }
    rc = grow(h->ws_qn, h->ws_qn_cap, nq);
    if (rc) return rc;
    return mfma_row_norms(qf, ld, false, 0, nq, h->ws_qn, st);
}

// The search workspaces (candidate lists, thresholds, padded queries, counters) are per handle,
// and the mutex only covers the host side of an asynchronous call.  When a search arrives on a
// different stream than the previous one, everything enqueued so far on the previous stream is
// made a dependency of the new stream (event recorded there now, waited for here), so two searches
// on one handle never overlap on the device.  Same stream: nothing to do (stream order).
int order_after_previous(rag_flat* h, hipStream_t st) {
    if (h->last_stream_valid && h->last_stream != st) {
        if (!h->order_ev) RAG_HIP(hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));
        RAG_HIP(hipEventRecord(h->order_ev, h->last_stream));
        RAG_HIP(hipStreamWaitEvent(st, h->order_ev, 0));
    }
    h->last_stream = st;
    h->last_stream_valid = true;
    return RAG_OK;
}

int search_device(rag_flat* h, int64_t nq, const float* q_dev, int k, float* D_dev, long long* I_dev,
                  int64_t id_base, hipStream_t st);

// The row-norm array follows the arena's capacity (entries below norm_valid are kept).
int ensure_norm_array(rag_flat* h, hipStream_t st) {
    if (h->xnorm_cap >= h->cap) return RAG_OK;
    float* nn = nullptr;
    RAG_HIP(hipMalloc(&nn, (size_t)h->cap * sizeof(float)));
    if (h->norm_valid > 0)
        RAG_HIP(hipMemcpyAsync(nn, h->xnorm, (size_t)h->norm_valid * sizeof(float), hipMemcpyDeviceToDevice, st));
    RAG_HIP(hipStreamSynchronize(st));
    if (h->xnorm) RAG_HIP(hipFree(h->xnorm));
    h->xnorm = nn;
    h->xnorm_cap = h->cap;
    return RAG_OK;
}

// Bring the bfloat16 shadow of an fp32 index up to its ndev rows (rows are only ever appended; reset empties
// both).  L2: the shadow's row norms are those of the fp32 rows, so that only the cross term carries the
// rounding of the copy.  Synchronises `st` when rows were copied (once per add, not per search).
int sync_shadow(rag_flat* h, hipStream_t st) {
    rag_flat* sh = h->shadow;
    if (sh->ndev == h->ndev) return RAG_OK;
    if (!h->sh_max) RAG_HIP(hipMalloc(&h->sh_max, 256));
    if (sh->ndev > h->ndev || sh->ndev == 0) {  // after a reset: start again
        sh->ndev = sh->ntotal = 0;
        sh->norm_valid = 0;
        RAG_HIP(hipMemsetAsync(h->sh_max, 0, 256, st));
    }
    int rc = ensure_capacity(sh, h->ndev);
    if (rc) return rc;
    float* xn = nullptr;
    if (h->metric == RAG_METRIC_L2) {
        rc = ensure_norm_array(sh, st);
        if (rc) return rc;
        xn = sh->xnorm + sh->ndev;
    }
    const int64_t first = sh->ndev, m = h->ndev - first;
    const int64_t nb = std::min<int64_t>((m + 3) / 4, (int64_t)num_cus() * 16);
    shadow_rows_kernel<<<(unsigned)nb, 256, 0, st>>>(reinterpret_cast<const float*>(h->x) + (size_t)first * h->ld, h->ld,
                                                    h->d, (long long)m,
                                                    reinterpret_cast<unsigned short*>(sh->x) + (size_t)first * sh->ld,
                                                    sh->ld, xn, h->sh_max);
    RAG_HIP(hipGetLastError());
    RAG_HIP(hipStreamSynchronize(st));
    sh->ndev = sh->ntotal = h->ndev;
    if (h->metric == RAG_METRIC_L2) sh->norm_valid = h->ndev;
    return RAG_OK;
}

// The whole device-side search: q_dev (nq x d) -> D_dev/I_dev (nq x k) on `st`.
int search_device(rag_flat* h, int64_t nq, const float* q_dev, int k, float* D_dev, long long* I_dev,
                  int64_t id_base, hipStream_t st) {
    const int64_t n = h->ndev;
    if (n == 0) {
        const long long total = nq * k;
        fill_pad_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
            D_dev, I_dev, total, h->metric == RAG_METRIC_IP ? -FLT_MAX : FLT_MAX);
        RAG_HIP(hipGetLastError());
        return RAG_OK;
    }
    if (n > (int64_t)INT_MAX - 8)
        return fail(RAG_ERR_UNSUPPORTED, "more than 2^31-9 rows in one shard (%lld)", (long long)n);
    const bool bf = h->dtype == RAG_DTYPE_BF16;
    const int ld = h->ld, q4 = ld / 4;      // float4 per fp32 query row
    const int dch = ld * h->esz / 16;       // 16-byte chunks per stored row
    const Options& opt = options();
    const int KS = ks_for(k > RAG_MAX_K ? RAG_MAX_K : k);

    int rc = RAG_OK;
    // 65..192 queries on a large shard: two or three passes of the 16x16 register-fed kernel
    // (flat_wide.hip, <= 64 queries per pass) beat the LDS tile kernel, whose 128-query block shapes
    // are half empty there (10M x 384, benchmarks/ab/run_mid_nq.sh: 65 / 96 / 128 / 192 queries
    // 8.9 / 9.0 / 9.4 / 16.2 ms on the tile kernel against 3.65-4.5 ms per pass).
    // fp32 rows only: bfloat16 passes are bound by re-reading the rows, and the tile kernel needs
    // one pass (10M x 768 bf16, 65 / 128 queries: 4.2 / 4.4 ms against 5.2 / 5.8 ms in chunks).
    // (not with the bfloat16 prefilter on: its copy takes the whole batch in one pass of the bf16 tile kernel)
    const bool prefiltered = h->shadow != nullptr && !bf && n >= opt.prefilter_min_rows && 4 * k <= 104 && opt.prefilter_kp == 0;
    if (!bf && !prefiltered && nq > 64 && nq <= 192 && k <= 128 && n >= opt.mfma_filter_min_rows && opt.mfma_filter_k <= 1 &&
        (opt.wide_max_nq == 0 || opt.wide_max_nq >= 64) && opt.mfma_min_nq <= 64) {
        WidePlan probe;
        if (wide_plan(n, 64, dch, bf, 1, 1024, 32, &probe)) {
            const int64_t nch = (nq + 63) / 64;
            const int64_t per = ((nq + nch - 1) / nch + 15) & ~(int64_t)15;
            for (int64_t c0 = 0; c0 < nq; c0 += per) {
                const int64_t cn = std::min<int64_t>(per, nq - c0);
                rc = search_device(h, cn, q_dev + c0 * h->d, k, D_dev + c0 * k, I_dev + c0 * k, id_base, st);
                if (rc) return rc;
            }
            return RAG_OK;
        }
    }
    // fp32 rows whose length needs no padding: the kernels read the caller's queries directly
    const bool q_direct = !bf && ld == h->d && (reinterpret_cast<uintptr_t>(q_dev) & 15) == 0;
    const float* qf = q_dev;  // nq x ld fp32 queries as the kernels see them
    if (!q_direct) {
        rc = grow(h->ws_q, h->ws_q_cap, nq * ld);
        if (rc) return rc;
        if (bf) {
            rc = grow(h->ws_qb, h->ws_qb_cap, nq * ld);
            if (rc) return rc;
        }
        const long long total = nq * ld;
        pad_queries_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(q_dev, h->ws_q, bf ? h->ws_qb : nullptr, nq,
                                                                           h->d, ld);
        RAG_HIP(hipGetLastError());
        qf = h->ws_q;
    }

    MfmaPlan plan;
    // batches from 3 queries (fp32 rows; 2 for bfloat16 rows) take the MFMA kernel: measured on
    // 10M rows, fp32 nq=2 scan 2.35 ms vs MFMA 3.1 ms, nq=3 4.6 vs 3.2; bf16 nq=2 2.61 vs 2.54
    const int64_t mfma_min_nq = opt.mfma_min_nq > 0 ? opt.mfma_min_nq : (bf ? 2 : 3);
    // One or two queries, k <= 128: the stream kernel (flat_stream.hip) where it applies -- it reads
    // the rows a little faster than the scan kernel below (10M x 384 fp32, nq = 1: 2.171 against
    // 2.19-2.26 ms per pass; 10M x 768 bf16: 2.164 against 2.28; k = 100 through filter mode 2.35
    // against 2.41-2.46 ms per step; benchmarks/ab/run_nq1_stream.sh).  An explicit "mfma_min_nq"
    // keeps such batches on the scan kernel.
    StreamPlan small_probe;
    const bool small_stream = opt.mfma_min_nq <= 0 && nq < mfma_min_nq && k <= 128 &&
                              stream_plan(n, nq, 1, dch, bf, false, 0, 1, &small_probe);
    // The LDS-resident lists of the tile kernel stop fitting at k ~ 122 (flat_mfma_kernel.h smem_need);
    // the stream kernel keeps its lists in registers and serves k <= 128 for up to 8 queries per pass, so
    // such batches go through it in chunks of 8 instead of falling to the scan kernel's query pairs.
    const bool plan_ok = nq <= INT_MAX && mfma_plan(n, nq, k, bf, &plan);
    if (!plan_ok && k <= 128 && nq > 8 && opt.mfma_min_nq <= 0 && nq >= mfma_min_nq) {
        StreamPlan probe8;
        if (stream_plan(n, 8, k, dch, bf, false, 0, 1, &probe8)) {
            for (int64_t c0 = 0; c0 < nq; c0 += 8) {
                const int64_t cn = std::min<int64_t>(8, nq - c0);
                rc = search_device(h, cn, q_dev + c0 * h->d, k, D_dev + c0 * k, I_dev + c0 * k, id_base, st);
                if (rc) return rc;
            }
            return RAG_OK;
        }
    }
    // bfloat16 prefilter (rag_flat_prefilter): kp candidates per query from the shadow, exact scores from the
    // fp32 rows, a proof per query that nothing else can rank among the top k, and the list-mode tile kernel
    // for the queries whose proof does not close (both of its kernels return at once when there are none)
    if (h->shadow != nullptr && !bf && plan_ok && n >= opt.prefilter_min_rows) {
        // four candidates per wanted row: with fewer the gap between the k-th and the kp-th score is often
        // inside the rounding margin (70k x 768 unit rows, k = 64 of 128 candidates: 23 of 33 queries redone)
        // (104, not 128: the list-mode tile kernel -- the redo of the copy's own filter flow -- keeps at most
        // 109 entries per list in LDS for its 128-query block shape)
        int kp = opt.prefilter_kp > 0 ? (int)std::min<int64_t>(opt.prefilter_kp, PF_MAX_KP)
                                      : ((nq <= 8 && k <= 16) ? 64 : 104);
        if (kp >= 4 * k || (opt.prefilter_kp > 0 && kp >= k)) {
            rc = sync_shadow(h, st);
            if (rc) return rc;
            rc = grow(h->ws_pD, h->ws_pD_cap, nq * kp);
            if (rc) return rc;
            rc = grow(h->ws_pI, h->ws_pI_cap, nq * kp);
            if (rc) return rc;
            rc = grow(h->ws_key, h->ws_key_cap, nq * plan.nlists * k);
            if (rc) return rc;
            rc = grow(h->ws_id, h->ws_id_cap, nq * plan.nlists * k);
            if (rc) return rc;
            rc = grow(h->ws_cnt, h->ws_cnt_cap, 2 * nq + 1);
            if (rc) return rc;
            rc = prepare_l2_norms(h, qf, nq, st);
            if (rc) return rc;
            if (!h->zero16) {
                RAG_HIP(hipMalloc(&h->zero16, 64));
                RAG_HIP(hipMemsetAsync(h->zero16, 0, 64, st));
            }
            h->shadow->prof = h->prof;
            rc = search_device(h->shadow, nq, q_dev, kp, h->ws_pD, h->ws_pI, 0, st);
            if (rc) return rc;
            int* nfail = h->ws_cnt + nq;
            int* failq = h->ws_cnt + nq + 1;
            RAG_HIP(hipMemsetAsync(nfail, 0, sizeof(int), st));
            const double u = 1.0 / 256.0;
            const double eps_rel = 2.0 * u + u * u + (double)(h->d + 16) / 8388608.0;
            const double slack = (double)(h->d + 32) / 8388608.0;
            const size_t smem = (size_t)ld * sizeof(float);
            const float* Xf = reinterpret_cast<const float*>(h->x);
            if (h->metric == RAG_METRIC_IP)
                prefilter_rescore_kernel<RAG_METRIC_IP><<<(unsigned)nq, PF_THREADS, smem, st>>>(
                    Xf, (long long)n, ld, h->d, q_dev, h->ws_pD, h->ws_pI, kp, k, h->sh_max, eps_rel, slack, D_dev, I_dev,
                    (long long)id_base, nfail, failq);
            else
                prefilter_rescore_kernel<RAG_METRIC_L2><<<(unsigned)nq, PF_THREADS, smem, st>>>(
                    Xf, (long long)n, ld, h->d, q_dev, h->ws_pD, h->ws_pI, kp, k, h->sh_max, eps_rel, slack, D_dev, I_dev,
                    (long long)id_base, nfail, failq);
            RAG_HIP(hipGetLastError());
            rc = mfma_launch(plan, Xf, (long long)n, ld, qf, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm, h->ws_qn,
                             h->zero16, st, nfail);
            if (rc) return rc;
            pick_merge(KS, h->metric)<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, h->ws_id, plan.nlists, k, D_dev, I_dev,
                                                                            (long long)id_base, k, 0, failq);
            RAG_HIP(hipGetLastError());
            h->last_nfail = nfail;
            snprintf(h->last_kernel, sizeof h->last_kernel, "prefilter kp%d [%.90s]", kp, h->shadow->last_kernel);
            return RAG_OK;
        }
    }
    StreamPlan sp_only;
    const bool stream_list_ok = !plan_ok && k <= 128 && nq <= 8 && opt.mfma_min_nq <= 0 &&
                                stream_plan(n, nq, k, dch, bf, false, 0, 1, &sp_only);
    if ((nq >= mfma_min_nq || small_stream) && (plan_ok || stream_list_ok)) {
        // Filter mode for larger k on large shards: a 1/FILTER_STEP sample of the row tiles gives
        // each query a threshold (its j-th best sample score), one pass appends every row at or
        // above it, and the top-k of the few hundred appended rows is the exact result.  With
        // E[count] = j*step and sd ~ step*sqrt(j), j = (2 + sqrt(4 + k/step))^2 puts k four
        // deviations below the mean; queries that still end outside [k, cap] (ties, skewed row
        // order) are redone by the list-mode kernel, which otherwise returns at once.
        // sample stride: 1/32 of the row tiles; 1/64 for batches of 256+ queries on 4M+ rows, where the
        // list-mode sample pass of the tile kernel is a tenth of the whole search ("mfma_filter_step")
        const int FILTER_STEP = opt.mfma_filter_step > 0 ? (int)std::min<int64_t>(opt.mfma_filter_step, 1024)
                                : (nq >= 256 && n >= (4LL << 20) ? 64 : 32);
        const double root = 2.0 + std::sqrt(4.0 + (double)k / FILTER_STEP);
        const int j = std::min(64, (int)std::ceil(root * root));
        int cap = 1024;
        while (cap < j * FILTER_STEP + 8 * FILTER_STEP * (int)std::ceil(std::sqrt((double)j))) cap *= 2;
        MfmaPlan ps, pf;
        // measured crossover (benchmarks/ab/run_filter_k.sh): nq = 64: filter mode wins from k = 16;
        // nq = 1024 (one list per query and block): from k = 32
        // stream kernel (nq <= 16, lists in registers; benchmarks/ab/run_stream_k.sh): 1..3 queries
        // list mode for every k <= 128 (the merge of the block lists costs ~10 us whatever k is:
        // nq 1, k 100: 2.23 ms against 2.28 in filter mode), 4..8 queries up to k = 64 (nq 8, k 100:
        // 3.0 ms against 2.48), 9..16 queries up to k ~ 16 (nq 16, k 32: 3.24 ms against 2.85)
        StreamPlan probe;
        const bool stream_ok = stream_plan(n, nq, 1, dch, bf, false, 0, 1, &probe);
        // 17..64 queries on a large shard: the 16x16 register-fed kernel (flat_wide.hip), which only
        // has sample and filter passes -> filter mode for every k
        WidePlan wp;
        // from 9 queries: 10M x 384, nq 16: 2.57-2.67 ms against 2.76 for the stream kernel (k = 100: 2.67 / 3.26)
        const int64_t wide_min_nq = opt.wide_min_nq > 0 ? opt.wide_min_nq : 9;
        // (filter mode and the wide kernel, which only has sample and filter passes, need the list-mode
        // tile kernel as their redo path)
        const bool wide = plan_ok && nq >= wide_min_nq && opt.mfma_filter_k <= 1 && n >= opt.mfma_filter_min_rows &&
                          wide_plan(n, nq, dch, bf, j, cap, FILTER_STEP, &wp);
        // more than 128 queries over rows of whole 128-byte stages: the pipelined tile kernel (flat_mfma_pipe.h)
        // only has sample and filter passes and beats the list-mode tile kernel at every k (10M x 384 fp32,
        // nq 1024, k 10: 57.2 ms for the whole flow against 62.1; benchmarks/ab/run_r03_f32_pipe.sh)
        const bool pipe_any_k = !stream_ok && nq > 128 && (ld * h->esz / 4) % 32 == 0 && opt.mfma_shape == 0;
        const int64_t filter_k = opt.mfma_filter_k > 0 ? opt.mfma_filter_k
                                 : wide                ? 1
                                 : stream_ok           ? (nq > 8 ? 24 : (nq >= 4 ? 65 : 129))
                                 : pipe_any_k          ? 1
                                                       : (nq <= 128 ? 16 : 33);
        const bool filter = plan_ok && k >= filter_k && n >= opt.mfma_filter_min_rows &&
                            mfma_plan(std::max<long long>(1, n / FILTER_STEP), nq, j, bf, &ps) &&
                            mfma_plan_filter(n, nq, bf, cap, &pf, (int)(ld * h->esz / 4));
        // batches of up to 16 queries: the register-fed stream kernel (flat_stream.hip) takes the pass
        // over all rows, in list mode or as the filter pass; sample pass and redo stay with the tile kernel
        StreamPlan sp, sps;
        const bool stream = !wide && stream_plan(n, nq, k, dch, bf, filter, cap, 1, &sp) &&
                            (!filter || stream_plan(n, nq, 1, dch, bf, false, 0, FILTER_STEP, &sps));
        if (!plan_ok && !stream) return fail(RAG_ERR_HIP, "dispatch: neither the tile nor the stream kernel serves this shape");
        int64_t need = plan_ok ? nq * plan.nlists * k : 0;
        if (stream && !filter) need = std::max<int64_t>(need, nq * (int64_t)sp.nlists * k);
        if (stream && filter) need = std::max<int64_t>(need, nq * (int64_t)sps.nlists);
        if (wide && filter) need = std::max<int64_t>(need, nq * (int64_t)wp.wpad);
        if (filter) need = std::max<int64_t>(need, std::max<int64_t>(nq * ps.nlists * j, nq * (int64_t)cap));
        // tile-kernel filter flow with the pipelined kernel: its sample mode leaves two values per visited row
        // tile and query, of which kth_largest_kernel sorts up to 4096
        const long long pipe_ntv = ((n + 255) / 256 + FILTER_STEP - 1) / FILTER_STEP;
        const bool pipe_sample = filter && !wide && !stream && mfma_tiled_query_words(pf, ld * h->esz / 4) > 0 &&
                                 2 * pipe_ntv <= 4096 && opt.mfma_pipe_sample >= 0;
        if (pipe_sample) need = std::max<int64_t>(need, nq * 2 * pipe_ntv);
        rc = grow(h->ws_key, h->ws_key_cap, need);
        if (rc) return rc;
        rc = grow(h->ws_id, h->ws_id_cap, need);
        if (rc) return rc;
        rc = prepare_l2_norms(h, qf, nq, st);
        if (rc) return rc;
        if (filter) {
            rc = grow(h->ws_sD, h->ws_sD_cap, nq * j);
            if (rc) return rc;
            rc = grow(h->ws_sI, h->ws_sI_cap, nq * j);
            if (rc) return rc;
            rc = grow(h->ws_thr, h->ws_thr_cap, nq);
            if (rc) return rc;
            rc = grow(h->ws_fi, h->ws_fi_cap, nq);
            if (rc) return rc;
            rc = grow(h->ws_cnt, h->ws_cnt_cap, 2 * nq + 1);
            if (rc) return rc;
            if (!wide && !stream) {
                rc = grow(h->ws_qt, h->ws_qt_cap, mfma_tiled_query_words(pf, ld * h->esz / 4));
                if (rc) return rc;
            }
        }
        ProfPair prof(h);
        rc = prof.begin(st);
        if (rc) return rc;
        if (!h->zero16) {
            RAG_HIP(hipMalloc(&h->zero16, 64));
            RAG_HIP(hipMemsetAsync(h->zero16, 0, 64, st));
        }
        // the kernel moves 16-byte chunks: both operands are addressed in 4-byte words
        const float* Xw = reinterpret_cast<const float*>(h->x);
        const float* Qw = bf ? reinterpret_cast<const float*>(h->ws_qb) : qf;
        const int ldw = ld * h->esz / 4;
        merge_fn_t mf = pick_merge(KS, h->metric);
        if (!filter) {
            snprintf(h->last_kernel, sizeof h->last_kernel, "%s list G%d RG%d KS%d NB%d blocks%d",
                     stream ? "stream" : "tile", stream ? sp.G : 0, stream ? sp.RG : 0, stream ? sp.KS : plan.KS,
                     stream ? sp.NB : plan.NS, stream ? sp.blocks : plan.nlists);
            // 1-2 queries: the stream kernel's last block merges the block lists itself (one launch per search)
            const bool fuse = stream && stream_can_fuse(sp, nq, k);
            if (fuse && !h->ws_ticket) {
                RAG_HIP(hipMalloc(&h->ws_ticket, 256));
                RAG_HIP(hipMemsetAsync(h->ws_ticket, 0, 256, st));
            }
            if (stream)
                rc = stream_launch(sp, h->x, (long long)n, dch, Qw, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm,
                                   h->ws_qn, st, nullptr, nullptr, nullptr, fuse ? h->ws_ticket : nullptr,
                                   fuse ? D_dev : nullptr, fuse ? I_dev : nullptr, (long long)id_base);
            else
                rc = mfma_launch(plan, Xw, (long long)n, ldw, Qw, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm,
                                 h->ws_qn, h->zero16, st);
            if (rc) return rc;
            rc = prof.end(st);
            if (rc) return rc;
            if (fuse) {
                strncat(h->last_kernel, " fused-merge", sizeof h->last_kernel - strlen(h->last_kernel) - 1);
                return RAG_OK;
            }
            mf<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, h->ws_id, stream ? sp.nlists : plan.nlists, k, D_dev,
                                                       I_dev, (long long)id_base, k, 0, nullptr);
            RAG_HIP(hipGetLastError());
            return RAG_OK;
        }
        // 1. sample pass over every FILTER_STEP-th row tile -> threshold per query.  The register-fed
        // kernels leave one best score per wave (wide) or block (stream, list mode with k = 1) and the
        // j-th largest of those is the threshold: it equals the j-th best sampled row unless two of
        // the best j fell to one wave / block, which only lowers it.  The tile kernel keeps j per list.
        ps.tile_step = FILTER_STEP;
        snprintf(h->last_kernel, sizeof h->last_kernel, "%s filter step%d j%d cap%d", wide ? "wide" : (stream ? "stream" : "tile"),
                 FILTER_STEP, j, cap);
        if (wide) {
            rc = wide_sample(wp, h->x, (long long)n, dch, Qw, (int)nq, h->metric, h->ws_key, h->ws_id, h->xnorm, h->ws_qn,
                             st);
            if (rc) return rc;
            kth_largest_kernel<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, wp.wpad, wp.blocks, j,
                                                                       h->ws_thr);
        } else if (stream) {
            rc = stream_launch(sps, h->x, (long long)n, dch, Qw, (int)nq, 1, h->metric, h->ws_key, h->ws_id, h->xnorm,
                               h->ws_qn, st);
            if (rc) return rc;
            kth_largest_kernel<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, sps.nlists, sps.nlists, j, h->ws_thr);
        } else if (pipe_sample) {
            // the pipelined kernel's own sample mode: the best score per wave tile and query, no lists
            rc = mfma_tile_queries(pf, Qw, (int)nq, ldw, h->ws_qt, st);
            if (rc) return rc;
            MfmaPlan psm = pf;
            psm.tile_step = FILTER_STEP;
            psm.S = (int)std::min<long long>(pf.S, pipe_ntv);
            psm.cap = (int)(2 * pipe_ntv);
            rc = mfma_launch(psm, Xw, (long long)n, ldw, h->ws_qt, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm,
                             h->ws_qn, h->zero16, st);
            if (rc) return rc;
            kth_largest_kernel<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, psm.cap, psm.cap, j, h->ws_thr);
        } else {
            rc = mfma_launch(ps, Xw, (long long)n, ldw, Qw, (int)nq, j, h->metric, h->ws_key, h->ws_id, h->xnorm,
                             h->ws_qn, h->zero16, st);
            if (rc) return rc;
            pick_merge(1, h->metric)<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, h->ws_id, ps.nlists, j, h->ws_sD,
                                                                            h->ws_sI, 0LL, j, 0, nullptr);
            make_floor_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(h->ws_sD, h->ws_sI, nq, j, j - 1, h->metric,
                                                                           0LL, h->ws_thr, h->ws_fi);
        }
        RAG_HIP(hipGetLastError());
        // 2. filter pass over all rows
        unsigned int* counts = reinterpret_cast<unsigned int*>(h->ws_cnt);
        int* nfail = h->ws_cnt + nq;
        int* failq = h->ws_cnt + nq + 1;
        RAG_HIP(hipMemsetAsync(h->ws_cnt, 0, (size_t)(nq + 1) * sizeof(int), st));
        if (wide)
            rc = wide_filter(wp, h->x, (long long)n, dch, Qw, (int)nq, h->metric, h->ws_key, h->ws_id, h->xnorm, h->ws_qn,
                             h->ws_thr, counts, st);
        else if (stream)
            rc = stream_launch(sp, h->x, (long long)n, dch, Qw, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm,
                               h->ws_qn, st, nullptr, h->ws_thr, counts);
        else {
            const float* Qf = Qw;
            if (mfma_tiled_query_words(pf, ldw) > 0) {  // the pipelined kernel reads a tiled image of the queries
                if (!pipe_sample) {                      // (already made for its sample pass)
                    rc = mfma_tile_queries(pf, Qw, (int)nq, ldw, h->ws_qt, st);
                    if (rc) return rc;
                }
                Qf = h->ws_qt;
            }
            rc = mfma_launch(pf, Xw, (long long)n, ldw, Qf, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm,
                             h->ws_qn, h->zero16, st, nullptr, h->ws_thr, counts);
        }
        if (rc) return rc;
        rc = prof.end(st);
        if (rc) return rc;
        // 3. exact top-k of each append buffer
#define RAG_SELECT(KSV)                                                                                             \
    do {                                                                                                            \
        if (h->metric == RAG_METRIC_IP)                                                                             \
            filter_select_kernel<KSV, RAG_METRIC_IP><<<(unsigned)nq, MERGE_THREADS, 0, st>>>(                       \
                h->ws_key, h->ws_id, counts, cap, k, D_dev, I_dev, (long long)id_base, nfail, failq);               \
        else                                                                                                        \
            filter_select_kernel<KSV, RAG_METRIC_L2><<<(unsigned)nq, MERGE_THREADS, 0, st>>>(                       \
                h->ws_key, h->ws_id, counts, cap, k, D_dev, I_dev, (long long)id_base, nfail, failq);               \
    } while (0)
        if (KS == 1) RAG_SELECT(1);
        else if (KS == 2) RAG_SELECT(2);
        else RAG_SELECT(4);
#undef RAG_SELECT
        RAG_HIP(hipGetLastError());
        // 4. list-mode redo of the queries that failed (both kernels return at once if none did)
        rc = mfma_launch(plan, Xw, (long long)n, ldw, Qw, (int)nq, k, h->metric, h->ws_key, h->ws_id, h->xnorm, h->ws_qn,
                         h->zero16, st, nfail);
        if (rc) return rc;
        mf<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(h->ws_key, h->ws_id, plan.nlists, k, D_dev, I_dev, (long long)id_base,
                                                   k, 0, failq);
        RAG_HIP(hipGetLastError());
        return RAG_OK;
    }

    // 128 < k <= 256 (document mode asks for 9 * top_k): the stream kernel with 256-entry lists, four
    // queries per pass (10M x 384: 2.4 ms per pass against 2.4-2.6 ms per PAIR on the scan kernel)
    if (k > 128 && k <= RAG_MAX_K && opt.mfma_min_nq <= 0 && nq <= INT_MAX) {
        StreamPlan sp4;
        if (stream_plan(n, std::min<int64_t>(nq, 4), k, dch, bf, false, 0, 1, &sp4)) {
            rc = prepare_l2_norms(h, qf, nq, st);
            if (rc) return rc;
            const char* Qb = bf ? reinterpret_cast<const char*>(h->ws_qb) : reinterpret_cast<const char*>(qf);
            const size_t qrow = (size_t)ld * h->esz;
            merge_fn_t mf4 = pick_merge(4, h->metric);
            snprintf(h->last_kernel, sizeof h->last_kernel, "stream list KS4 (four queries per pass)");
            for (int64_t q0 = 0; q0 < nq; q0 += 4) {
                const int cn = (int)std::min<int64_t>(4, nq - q0);
                if (!stream_plan(n, cn, k, dch, bf, false, 0, 1, &sp4)) return fail(RAG_ERR_HIP, "stream plan changed");
                rc = grow(h->ws_key, h->ws_key_cap, (int64_t)cn * sp4.nlists * k);
                if (rc) return rc;
                rc = grow(h->ws_id, h->ws_id_cap, (int64_t)cn * sp4.nlists * k);
                if (rc) return rc;
                ProfPair prof(h);
                rc = prof.begin(st);
                if (rc) return rc;
                rc = stream_launch(sp4, h->x, (long long)n, dch, Qb + (size_t)q0 * qrow, cn, k, h->metric, h->ws_key,
                                   h->ws_id, h->xnorm, h->ws_qn ? h->ws_qn + q0 : nullptr, st);
                if (rc) return rc;
                rc = prof.end(st);
                if (rc) return rc;
                mf4<<<(unsigned)cn, MERGE_THREADS, 0, st>>>(h->ws_key, h->ws_id, sp4.nlists, k, D_dev + q0 * k,
                                                            I_dev + q0 * k, (long long)id_base, k, 0, nullptr);
                RAG_HIP(hipGetLastError());
            }
            return RAG_OK;
        }
    }

    // waves per CU (benchmarks/ab/run_scan_threads_ab.sh): fp32 rows stream fastest with 12 (10M x 384:
    // 2.19 ms against 2.215 ms with 16; 14 and 10 are slower), bfloat16 rows need 16 to cover the
    // unpacking (2.28 ms against 2.63 ms with 12)
    snprintf(h->last_kernel, sizeof h->last_kernel, "scan");
    int64_t blocks = opt.scan_blocks > 0 ? opt.scan_blocks : (int64_t)num_cus() * (bf ? 4 : 3);
    int Rsel = (opt.scan_rows == 2 || opt.scan_rows == 8) ? (int)opt.scan_rows : 4;

    // k <= 256: one pass.  Larger k: passes of <= 256, each bounded by the previous pass' last entry.
    for (int done = 0; done < k; done += RAG_MAX_K) {
        const int kp = std::min(RAG_MAX_K, k - done);
        const int KSp = ks_for(kp);
        const float* fl_k = nullptr;
        const int* fl_i = nullptr;
        if (done > 0) {
            rc = grow(h->ws_fk, h->ws_fk_cap, nq);
            if (rc) return rc;
            rc = grow(h->ws_fi, h->ws_fi_cap, nq);
            if (rc) return rc;
            make_floor_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(D_dev, I_dev, nq, k, done - 1, h->metric,
                                                                           (long long)id_base, h->ws_fk, h->ws_fi);
            RAG_HIP(hipGetLastError());
            fl_k = h->ws_fk;
            fl_i = h->ws_fi;
        }
        for (int64_t q0 = 0; q0 < nq;) {
            const int64_t left = nq - q0;
            // pairs at most: a 4-query tile (16 partial sums per wave reduction) measured 2.9x the
            // time of a pair (10M x 384: 6.8 ms vs 2.35 ms)
            const int NQ = left >= 2 ? 2 : 1;
            const int R = NQ == 1 ? Rsel : 4;
            const int64_t ntiles = (n + R - 1) / R;
            int64_t nb = std::min<int64_t>(blocks, (ntiles + SCAN_WAVES - 1) / SCAN_WAVES);
            nb = std::max<int64_t>(nb, 1);
            rc = grow(h->ws_key, h->ws_key_cap, (int64_t)NQ * nb * kp);
            if (rc) return rc;
            rc = grow(h->ws_id, h->ws_id_cap, (int64_t)NQ * nb * kp);
            if (rc) return rc;
            const size_t qbytes = (size_t)NQ * q4 * 16;
            const size_t smem = qbytes + (size_t)(SCAN_WAVES - 1) * NQ * KSp * 64 * 8;
            if (smem > 160 * 1024)
                return fail(RAG_ERR_UNSUPPORTED, "d=%d too large for the LDS-resident query (%zu B)", h->d, smem);
            ScanCfg cfg{R, NQ, KSp, h->metric, opt.scan_nt != 0, bf};
            scan_fn_t fn = pick_scan(cfg);
            if (smem > 48 * 1024)
                RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            ProfPair prof(h);
            rc = prof.begin(st);
            if (rc) return rc;
            fn<<<(unsigned)nb, SCAN_THREADS, smem, st>>>(reinterpret_cast<const f32x4*>(h->x), (long long)n, dch,
                                                        reinterpret_cast<const f32x4*>(qf + q0 * ld), kp,
                                                        h->ws_key, h->ws_id, fl_k ? fl_k + q0 : nullptr,
                                                        fl_i ? fl_i + q0 : nullptr);
            RAG_HIP(hipGetLastError());
            rc = prof.end(st);
            if (rc) return rc;
            merge_fn_t mf = pick_merge(KSp, h->metric);
            mf<<<(unsigned)NQ, MERGE_THREADS, 0, st>>>(h->ws_key, h->ws_id, (int)nb, kp, D_dev + q0 * k,
                                                       I_dev + q0 * k, (long long)id_base, k, done, nullptr);
            RAG_HIP(hipGetLastError());
            q0 += NQ;
        }
    }
    return RAG_OK;
}

// The no-allocation search: the stream kernel in list mode (<= 8 queries, k <= 128) + the merge of its
// block lists, workspaces inside `ws` (this device's copy of the static scratch array).  Nothing here
// allocates or frees; every byte of scratch a kernel reads was written earlier in the same call
// (queries by pad_queries_kernel, norms by the norm kernel, every block writes all k entries of each
// of its lists), so nothing has to start at zero and a call does not depend on the previous one.
int search_device_noalloc(rag_flat* h, int64_t nq, const float* q_dev, int k, float* D_dev, long long* I_dev,
                          int64_t id_base, hipStream_t st, unsigned char* ws) {
    const int64_t n = h->ndev;
    if (n == 0) {
        const long long total = nq * k;
        fill_pad_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
            D_dev, I_dev, total, h->metric == RAG_METRIC_IP ? -FLT_MAX : FLT_MAX);
        RAG_HIP(hipGetLastError());
        return RAG_OK;
    }
    if (n > (int64_t)INT_MAX - 8)
        return fail(RAG_ERR_UNSUPPORTED, "more than 2^31-9 rows in one shard (%lld)", (long long)n);
    const bool bf = h->dtype == RAG_DTYPE_BF16;
    const int ld = h->ld;
    const int dch = ld * h->esz / 16;
    if (h->metric == RAG_METRIC_L2 && h->norm_valid < n)
        return fail(RAG_ERR_UNSUPPORTED,
                    "rag_flat_search_dev_noalloc: the L2 row norms of %lld rows are not computed yet (they live in "
                    "allocated memory): run one rag_flat_search[_dev] after adding rows",
                    (long long)(n - h->norm_valid));
    StreamPlan sp;
    if (!stream_plan(n, nq, k, dch, bf, false, 0, 1, &sp))
        return fail(RAG_ERR_UNSUPPORTED,
                    "rag_flat_search_dev_noalloc: shape not served by the stream kernel (n=%lld rows [>= 16], d=%d)",
                    (long long)n, h->d);
    if ((size_t)nq * (size_t)sp.nlists * (size_t)k > NA_LIST_ENTRIES)
        return fail(RAG_ERR_UNSUPPORTED, "rag_flat_search_dev_noalloc: %lld x %d lists of %d entries exceed the scratch",
                    (long long)nq, sp.nlists, k);
    float* ws_q = reinterpret_cast<float*>(ws);
    unsigned short* ws_qb = reinterpret_cast<unsigned short*>(ws + NA_OFF_QB);
    float* ws_qn = reinterpret_cast<float*>(ws + NA_OFF_QN);
    float* ws_key = reinterpret_cast<float*>(ws + NA_OFF_KEY);
    int* ws_id = reinterpret_cast<int*>(ws + NA_OFF_ID);
    const bool q_direct = !bf && ld == h->d && (reinterpret_cast<uintptr_t>(q_dev) & 15) == 0;
    const float* qf = q_dev;
    if (!q_direct) {
        const long long total = nq * ld;
        pad_queries_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(q_dev, ws_q, bf ? ws_qb : nullptr, nq, h->d, ld);
        RAG_HIP(hipGetLastError());
        qf = ws_q;
    }
    if (h->metric == RAG_METRIC_L2) {
        int rc = mfma_row_norms(qf, ld, false, 0, nq, ws_qn, st);
        if (rc) return rc;
    }
    const void* Qw = bf ? static_cast<const void*>(ws_qb) : static_cast<const void*>(qf);
    snprintf(h->last_kernel, sizeof h->last_kernel, "stream list G%d RG%d KS%d NB%d blocks%d (noalloc)", sp.G, sp.RG, sp.KS,
             sp.NB, sp.blocks);
    int rc = stream_launch(sp, h->x, (long long)n, dch, Qw, (int)nq, k, h->metric, ws_key, ws_id, h->xnorm, ws_qn, st);
    if (rc) return rc;
    pick_merge(ks_for(k), h->metric)<<<(unsigned)nq, MERGE_THREADS, 0, st>>>(ws_key, ws_id, sp.nlists, k, D_dev, I_dev,
                                                                            (long long)id_base, k, 0, nullptr);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

}  // namespace

namespace rag {
int flat_with_view(rag_flat_t* h, int (*use)(const FlatView&, void*), void* ctx) {
    std::lock_guard<std::mutex> lk(h->mu);
    const int rc = flush_staged(h);
    if (rc) return rc;
    RAG_HIP(hipStreamSynchronize(lib_stream()));
    const FlatView v{h->x, h->ld, h->d, h->dtype == RAG_DTYPE_BF16 ? 1 : 0, h->ndev};
    return use(v, ctx);
}
// Called by rag_init() with the selected device current, once per process (one process per GPU).
int flat_init_scratch() {
    int dev = -1;
    RAG_HIP(hipGetDevice(&dev));
    if (g_scratch_ptr != nullptr && g_scratch_dev == dev) return RAG_OK;
    unsigned char* p = nullptr;
    RAG_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
    if (p == nullptr) return fail(RAG_ERR_HIP, "hipGetSymbolAddress(scratch) returned NULL");
    g_scratch_ptr = p;
    g_scratch_dev = dev;
    return RAG_OK;
}
}  // namespace rag

// ---------------------------------------------------------------- C ABI

extern "C" {

int rag_flat_create(int d, int metric, int storage_dtype, rag_flat_t** out) {
    RAG_REQUIRE_INIT();
    if (!out) return fail(RAG_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (d <= 0 || d > 8192) return fail(RAG_ERR_INVALID_ARG, "d=%d out of range [1,8192]", d);
    if (metric != RAG_METRIC_IP && metric != RAG_METRIC_L2)
        return fail(RAG_ERR_INVALID_ARG, "metric %d is neither IP(0) nor L2(1)", metric);
    if (storage_dtype != RAG_DTYPE_F32 && storage_dtype != RAG_DTYPE_BF16)
        return fail(RAG_ERR_INVALID_ARG, "storage dtype %d is neither fp32(0) nor bf16(1)", storage_dtype);
    rag_flat* h = new rag_flat();
    h->d = d;
    h->esz = storage_dtype == RAG_DTYPE_BF16 ? 2 : 4;
    h->ld = storage_dtype == RAG_DTYPE_BF16 ? ((d + 7) & ~7) : ((d + 3) & ~3);
    h->metric = metric;
    h->dtype = storage_dtype;
    // the ordering event exists from the start, so that order_after_previous never creates anything
    // (rag_flat_search_dev_noalloc calls it too)
    if (hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return fail(RAG_ERR_HIP, "hipEventCreateWithFlags failed");
    }
    *out = h;
    return RAG_OK;
}

void rag_flat_free(rag_flat_t* h) {
    if (!h) return;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        (void)hipDeviceSynchronize();  // searches may still be running on caller streams
        (void)hipFree(h->x);
        (void)hipFree(h->ws_q);
        (void)hipFree(h->ws_qb);
        (void)hipFree(h->ws_qt);
        (void)hipFree(h->ws_ticket);
        (void)hipFree(h->sh_max);
        (void)hipFree(h->ws_pD);
        (void)hipFree(h->ws_pI);
        if (h->async_st) (void)hipStreamDestroy(h->async_st);
        (void)hipFree(h->ws_cvt);
        (void)hipFree(h->ws_key);
        (void)hipFree(h->ws_id);
        (void)hipFree(h->ws_hq);
        (void)hipFree(h->ws_qn);
        (void)hipFree(h->ws_sD);
        (void)hipFree(h->ws_sI);
        (void)hipFree(h->ws_thr);
        (void)hipFree(h->ws_cnt);
        (void)hipFree(h->zero16);
        (void)hipFree(h->ws_fk);
        (void)hipFree(h->ws_fi);
        (void)hipFree(h->xnorm);
        (void)hipFree(h->ws_D);
        (void)hipFree(h->ws_I);
        if (h->pin) (void)hipHostFree(h->pin);
        if (h->order_ev) (void)hipEventDestroy(h->order_ev);
        for (auto& p : h->ev_used) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        for (auto& p : h->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    }
    if (h->shadow) rag_flat_free(h->shadow);
    delete h;
}

int rag_flat_prefilter_redone(rag_flat_t* h, int64_t* n_queries) {
    RAG_REQUIRE_INIT();
    if (!h || !n_queries) return fail(RAG_ERR_INVALID_ARG, "bad handle or NULL output");
    std::lock_guard<std::mutex> lk(h->mu);
    *n_queries = 0;
    if (!h->last_nfail || std::strncmp(h->last_kernel, "prefilter", 9) != 0) return RAG_OK;
    int v = 0;
    RAG_HIP(hipDeviceSynchronize());  // the search may still run on a caller's stream
    RAG_HIP(hipMemcpy(&v, h->last_nfail, sizeof v, hipMemcpyDeviceToHost));
    *n_queries = v;
    return RAG_OK;
}

int rag_flat_prefilter(rag_flat_t* h, int enable) {
    RAG_REQUIRE_INIT();
    if (!h) return fail(RAG_ERR_INVALID_ARG, "bad handle");
    rag_flat* old = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (enable) {
            if (h->dtype != RAG_DTYPE_F32)
                return fail(RAG_ERR_UNSUPPORTED, "rag_flat_prefilter: the index already stores bfloat16 rows");
            if (h->shadow) return RAG_OK;
            rag_flat_t* sh = nullptr;
            const int rc = rag_flat_create(h->d, h->metric, RAG_DTYPE_BF16, &sh);
            if (rc) return rc;
            h->shadow = sh;  // rows are copied by the next search (sync_shadow)
            return RAG_OK;
        }
        old = h->shadow;
        h->shadow = nullptr;
    }
    if (old) rag_flat_free(old);  // waits for the device
    return RAG_OK;
}

int rag_flat_reserve(rag_flat_t* h, int64_t n_capacity) {
    RAG_REQUIRE_INIT();
    if (!h || n_capacity < 0) return fail(RAG_ERR_INVALID_ARG, "bad handle or capacity");
    std::lock_guard<std::mutex> lk(h->mu);
    return ensure_capacity(h, n_capacity);
}

int rag_flat_add(rag_flat_t* h, int64_t n, const float* x) {
    RAG_REQUIRE_INIT();
    if (!h || n < 0 || (n > 0 && !x)) return fail(RAG_ERR_INVALID_ARG, "bad handle, n or x");
    if (n == 0) return RAG_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    const int64_t nf = n * h->d;
    if (nf >= STAGE_FLUSH_FLOATS) {  // bulk add: straight to HBM
        int rc = flush_staged(h);
        if (rc) return rc;
        rc = upload_rows(h, x, n);
        if (rc) return rc;
    } else {
        h->staged.insert(h->staged.end(), x, x + nf);
        if ((int64_t)h->staged.size() >= STAGE_FLUSH_FLOATS) {
            int rc = flush_staged(h);
            if (rc) return rc;
        }
    }
    h->ntotal += n;
    return RAG_OK;
}

int rag_flat_add_dev(rag_flat_t* h, int64_t n, const float* x_dev, void* stream) {
    RAG_REQUIRE_INIT();
    if (!h || n < 0 || (n > 0 && !x_dev)) return fail(RAG_ERR_INVALID_ARG, "bad handle, n or x_dev");
    if (n == 0) return RAG_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = flush_staged(h);
    if (rc) return rc;
    rc = ensure_capacity(h, h->ndev + n);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;  // NULL = HIP's null stream
    float* dst = reinterpret_cast<float*>(h->x) + (size_t)h->ndev * h->ld;
    if (h->dtype == RAG_DTYPE_BF16) {
        rc = convert_rows_dev(h, x_dev, h->ndev, n, st);
        if (rc) return rc;
    } else if (h->ld == h->d)
        RAG_HIP(hipMemcpyAsync(dst, x_dev, (size_t)n * h->d * sizeof(float), hipMemcpyDeviceToDevice, st));
    else
        RAG_HIP(hipMemcpy2DAsync(dst, (size_t)h->ld * sizeof(float), x_dev, (size_t)h->d * sizeof(float),
                                 (size_t)h->d * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, st));
    RAG_HIP(hipStreamSynchronize(st));
    h->ndev += n;
    h->ntotal += n;
    return RAG_OK;
}

int rag_flat_add_synthetic(rag_flat_t* h, int64_t n, uint64_t seed, int64_t first_global_row) {
    RAG_REQUIRE_INIT();
    if (!h || n < 0) return fail(RAG_ERR_INVALID_ARG, "bad handle or n");
    if (n == 0) return RAG_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = flush_staged(h);
    if (rc) return rc;
    rc = ensure_capacity(h, h->ndev + n);
    if (rc) return rc;
    hipStream_t st = lib_stream();
    const int64_t nb = std::min<int64_t>((n + 3) / 4, (int64_t)num_cus() * 8);
    void* dst = h->x + (size_t)h->ndev * h->ld * h->esz;
    if (h->dtype == RAG_DTYPE_BF16)
        synth_rows_kernel<true><<<(unsigned)nb, 256, 0, st>>>(dst, h->ld, h->d, (long long)n, (unsigned long long)seed,
                                                             (long long)first_global_row);
    else
        synth_rows_kernel<false><<<(unsigned)nb, 256, 0, st>>>(dst, h->ld, h->d, (long long)n, (unsigned long long)seed,
                                                              (long long)first_global_row);
    RAG_HIP(hipGetLastError());
    RAG_HIP(hipStreamSynchronize(st));
    h->ndev += n;
    h->ntotal += n;
    return RAG_OK;
}

int64_t rag_flat_ntotal(const rag_flat_t* h) { return h ? h->ntotal : -1; }
const char* rag_flat_last_kernel(const rag_flat_t* h) { return h ? h->last_kernel : ""; }
int rag_flat_dim(const rag_flat_t* h) { return h ? h->d : -1; }

int rag_flat_reset(rag_flat_t* h) {
    RAG_REQUIRE_INIT();
    if (!h) return fail(RAG_ERR_INVALID_ARG, "bad handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->staged.clear();
    h->ntotal = 0;
    h->ndev = 0;
    h->norm_valid = 0;
    if (h->shadow) {  // refilled from row 0 by the next search (sync_shadow)
        h->shadow->ntotal = h->shadow->ndev = 0;
        h->shadow->norm_valid = 0;
    }
    return RAG_OK;
}

int rag_flat_search_dev(rag_flat_t* h, int64_t nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev,
                        int64_t id_base, void* stream) {
    RAG_REQUIRE_INIT();
    if (!h || nq < 0 || (nq > 0 && (!q_dev || !D_dev || !I_dev)))
        return fail(RAG_ERR_INVALID_ARG, "bad handle, nq or NULL buffer");
    if (k < 1 || k > RAG_MAX_K_TOTAL) return fail(RAG_ERR_INVALID_ARG, "k=%d out of range [1,%d]", k, RAG_MAX_K_TOTAL);
    if (nq == 0) return RAG_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    int rc = flush_staged(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;  // NULL = HIP's null stream, as in any HIP API
    rc = order_after_previous(h, st);
    if (rc) return rc;
    return search_device(h, nq, q_dev, k, D_dev, reinterpret_cast<long long*>(I_dev), id_base, st);
}

int rag_flat_search_dev_noalloc(rag_flat_t* h, int64_t nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev,
                                int64_t id_base, void* stream) {
    RAG_REQUIRE_INIT();
    if (!h || nq < 0 || (nq > 0 && (!q_dev || !D_dev || !I_dev)))
        return fail(RAG_ERR_INVALID_ARG, "bad handle, nq or NULL buffer");
    if (nq > RAG_NOALLOC_MAX_NQ)
        return fail(RAG_ERR_INVALID_ARG, "rag_flat_search_dev_noalloc: nq=%lld beyond its maximum %d", (long long)nq,
                    RAG_NOALLOC_MAX_NQ);
    if (k < 1 || k > RAG_NOALLOC_MAX_K)
        return fail(RAG_ERR_INVALID_ARG, "rag_flat_search_dev_noalloc: k=%d out of range [1,%d]", k, RAG_NOALLOC_MAX_K);
    if (nq == 0) return RAG_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->staged.empty())
        return fail(RAG_ERR_UNSUPPORTED,
                    "rag_flat_search_dev_noalloc: %lld added rows are still staged on the host (flushing them "
                    "allocates): call rag_flat_search[_dev] once, or add rows in bulk",
                    (long long)(h->staged.size() / h->d));
    int dev = -1;
    RAG_HIP(hipGetDevice(&dev));
    if (g_scratch_ptr == nullptr || dev != g_scratch_dev)
        return fail(RAG_ERR_NOT_INIT, "rag_flat_search_dev_noalloc: no scratch address for device %d (rag_init ran on %d)",
                    dev, g_scratch_dev);
    // ordered behind the handle's previous search when that ran on another stream (it may still be writing the
    // L2 row norms this one reads): an event record + wait, nothing is created or allocated
    const int rc = order_after_previous(h, (hipStream_t)stream);
    if (rc) return rc;
    return search_device_noalloc(h, nq, q_dev, k, D_dev, reinterpret_cast<long long*>(I_dev), id_base,
                                 (hipStream_t)stream, g_scratch_ptr);
}

namespace {

// Host-buffer search, enqueue part: queries to the device, search_device on `st`, results on their way back.
// staged (small batches, the reference's one query per call): queries and results go through the handle's
// pinned buffer, so the transfers are asynchronous DMA behind ONE synchronisation; direct: the search's last
// kernel writes D / I straight into that pinned buffer (no copy back at all).  Caller holds h->mu.
struct HostSearch {
    bool staged = false;
    size_t ioff = 0, doff = 0, db = 0, ib = 0;
};

int host_search_enqueue(rag_flat* h, int64_t nq, const float* q, int k, float* D, int64_t* I, hipStream_t st,
                        bool direct, HostSearch* hs) {
    int rc = flush_staged(h);
    if (rc) return rc;
    rc = order_after_previous(h, st);
    if (rc) return rc;
    rc = grow(h->ws_hq, h->ws_hq_cap, nq * h->d);
    if (rc) return rc;
    const size_t qb = (size_t)nq * h->d * sizeof(float);
    hs->db = (size_t)nq * k * sizeof(float);
    hs->ib = (size_t)nq * k * sizeof(long long);
    const size_t qoff = 0;
    hs->ioff = (qb + 15) & ~(size_t)15;
    hs->doff = hs->ioff + ((hs->ib + 15) & ~(size_t)15);
    const size_t need = hs->doff + hs->db;
    hs->staged = need <= ((size_t)4 << 20);
    if (direct && !hs->staged)
        return fail(RAG_ERR_UNSUPPORTED, "rag_flat_search_begin: nq * (d + 3k) * 4 bytes must stay under 4 MiB (nq=%lld, k=%d)",
                    (long long)nq, k);
    if (!direct && nq * k > h->ws_out_cap) {
        if (h->ws_D) RAG_HIP(hipFree(h->ws_D));
        if (h->ws_I) RAG_HIP(hipFree(h->ws_I));
        h->ws_D = nullptr; h->ws_I = nullptr; h->ws_out_cap = 0;
        const int64_t want = std::max<int64_t>(nq * k, 1024);
        RAG_HIP(hipMalloc(&h->ws_D, (size_t)want * sizeof(float)));
        RAG_HIP(hipMalloc(&h->ws_I, (size_t)want * sizeof(long long)));
        h->ws_out_cap = want;
    }
    if (hs->staged && need > h->pin_cap) {
        if (h->pin) RAG_HIP(hipHostFree(h->pin));
        h->pin = nullptr; h->pin_cap = 0;
        const size_t want = std::max<size_t>(need, (size_t)64 << 10);
        RAG_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pin), want, hipHostMallocDefault));
        h->pin_cap = want;
    }
    if (hs->staged) {
        std::memcpy(h->pin + qoff, q, qb);
        RAG_HIP(hipMemcpyAsync(h->ws_hq, h->pin + qoff, qb, hipMemcpyHostToDevice, st));
    } else {
        RAG_HIP(hipMemcpyAsync(h->ws_hq, q, qb, hipMemcpyHostToDevice, st));
    }
    if (direct)
        return search_device(h, nq, h->ws_hq, k, reinterpret_cast<float*>(h->pin + hs->doff),
                             reinterpret_cast<long long*>(h->pin + hs->ioff), 0, st);
    rc = search_device(h, nq, h->ws_hq, k, h->ws_D, h->ws_I, 0, st);
    if (rc) return rc;
    RAG_HIP(hipMemcpyAsync(hs->staged ? (void*)(h->pin + hs->doff) : (void*)D, h->ws_D, hs->db, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipMemcpyAsync(hs->staged ? (void*)(h->pin + hs->ioff) : (void*)I, h->ws_I, hs->ib, hipMemcpyDeviceToHost, st));
    return RAG_OK;
}

}  // namespace

int rag_flat_search(rag_flat_t* h, int64_t nq, const float* q, int k, float* D, int64_t* I) {
    RAG_REQUIRE_INIT();
    if (!h || nq < 0 || (nq > 0 && (!q || !D || !I)))
        return fail(RAG_ERR_INVALID_ARG, "bad handle, nq or NULL buffer");
    if (k < 1 || k > RAG_MAX_K_TOTAL) return fail(RAG_ERR_INVALID_ARG, "k=%d out of range [1,%d]", k, RAG_MAX_K_TOTAL);
    if (nq == 0) return RAG_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun) return fail(RAG_ERR_INVALID_ARG, "rag_flat_search: a search begun with rag_flat_search_begin has not been ended");
    hipStream_t st = lib_stream();
    HostSearch hs;
    int rc = host_search_enqueue(h, nq, q, k, D, I, st, false, &hs);
    if (rc) return rc;
    RAG_HIP(hipStreamSynchronize(st));
    if (hs.staged) {
        std::memcpy(D, h->pin + hs.doff, hs.db);
        std::memcpy(I, h->pin + hs.ioff, hs.ib);
    }
    return RAG_OK;
}

int rag_flat_search_begin(rag_flat_t* h, int64_t nq, const float* q, int k) {
    RAG_REQUIRE_INIT();
    if (!h || nq < 1 || !q) return fail(RAG_ERR_INVALID_ARG, "bad handle, nq or NULL buffer");
    if (k < 1 || k > RAG_MAX_K_TOTAL) return fail(RAG_ERR_INVALID_ARG, "k=%d out of range [1,%d]", k, RAG_MAX_K_TOTAL);
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->begun) return fail(RAG_ERR_INVALID_ARG, "rag_flat_search_begin: the previous begun search has not been ended");
    if (!h->async_st) RAG_HIP(hipStreamCreateWithFlags(&h->async_st, hipStreamNonBlocking));
    HostSearch hs;
    const int rc = host_search_enqueue(h, nq, q, k, nullptr, nullptr, h->async_st, true, &hs);
    if (rc) return rc;
    h->begun = true;
    h->begun_doff = hs.doff;
    h->begun_ioff = hs.ioff;
    h->begun_db = hs.db;
    h->begun_ib = hs.ib;
    return RAG_OK;
}

int rag_flat_search_end(rag_flat_t* h, float* D, int64_t* I) {
    RAG_REQUIRE_INIT();
    if (!h || !D || !I) return fail(RAG_ERR_INVALID_ARG, "bad handle or NULL buffer");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->begun) return fail(RAG_ERR_INVALID_ARG, "rag_flat_search_end: no search has been begun on this handle");
    h->begun = false;
    RAG_HIP(hipStreamSynchronize(h->async_st));
    std::memcpy(D, h->pin + h->begun_doff, h->begun_db);
    std::memcpy(I, h->pin + h->begun_ioff, h->begun_ib);
    return RAG_OK;
}

int rag_flat_copy_to_host(rag_flat_t* h, int64_t first, int64_t n, float* out) {
    RAG_REQUIRE_INIT();
    if (!h || first < 0 || n < 0 || (n > 0 && !out)) return fail(RAG_ERR_INVALID_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (first + n > h->ntotal)
        return fail(RAG_ERR_INVALID_ARG, "rows [%lld,%lld) exceed ntotal=%lld", (long long)first,
                    (long long)(first + n), (long long)h->ntotal);
    if (n == 0) return RAG_OK;
    int rc = flush_staged(h);
    if (rc) return rc;
    RAG_HIP(hipStreamSynchronize(lib_stream()));
    if (h->dtype == RAG_DTYPE_BF16) {  // exact widening of the stored (rounded) values
        hipStream_t st = lib_stream();
        const int64_t rows_per = std::max<int64_t>(1, CVT_CHUNK_FLOATS / h->d);
        rc = grow(h->ws_cvt, h->ws_cvt_cap, std::min(n, rows_per) * h->d);
        if (rc) return rc;
        for (int64_t r0 = 0; r0 < n; r0 += rows_per) {
            const int64_t m = std::min(rows_per, n - r0);
            const long long total = (long long)m * h->d;
            rows_from_bf16_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
                reinterpret_cast<const unsigned short*>(h->x) + (size_t)(first + r0) * h->ld, h->ld, h->ws_cvt, h->d,
                (long long)m);
            RAG_HIP(hipGetLastError());
            RAG_HIP(hipMemcpyAsync(out + r0 * h->d, h->ws_cvt, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, st));
            RAG_HIP(hipStreamSynchronize(st));
        }
        return RAG_OK;
    }
    const float* src = reinterpret_cast<const float*>(h->x) + (size_t)first * h->ld;
    if (h->ld == h->d)
        RAG_HIP(hipMemcpy(out, src, (size_t)n * h->d * sizeof(float), hipMemcpyDeviceToHost));
    else
        RAG_HIP(hipMemcpy2D(out, (size_t)h->d * sizeof(float), src, (size_t)h->ld * sizeof(float),
                            (size_t)h->d * sizeof(float), (size_t)n, hipMemcpyDeviceToHost));
    return RAG_OK;
}

int rag_topk_merge_strided_dev(int metric, int nlists, int64_t nq, int k, const float* Dl, const int64_t* Il,
                               int64_t stride_D, int64_t stride_I, float* D, int64_t* I, void* stream) {
    RAG_REQUIRE_INIT();
    if (nlists < 1 || nq < 0 || k < 1 || k > RAG_MAX_K || !Dl || !Il || !D || !I)
        return fail(RAG_ERR_INVALID_ARG, "bad argument to rag_topk_merge_dev");
    if (stride_D < nq * k || stride_I < nq * k) return fail(RAG_ERR_INVALID_ARG, "list stride smaller than nq*k");
    if (metric != RAG_METRIC_IP && metric != RAG_METRIC_L2) return fail(RAG_ERR_INVALID_ARG, "bad metric");
    if (nq == 0) return RAG_OK;
    hipStream_t st = (hipStream_t)stream;  // NULL = HIP's null stream
    const int KS = ks_for(k);
    const long long* Ill = reinterpret_cast<const long long*>(Il);
    long long* Io = reinterpret_cast<long long*>(I);
    if (KS == 1)
        ext_merge_kernel<1><<<(unsigned)nq, MERGE_THREADS, 0, st>>>(metric, Dl, Ill, nlists, nq, k, D, Io, stride_D, stride_I);
    else if (KS == 2)
        ext_merge_kernel<2><<<(unsigned)nq, MERGE_THREADS, 0, st>>>(metric, Dl, Ill, nlists, nq, k, D, Io, stride_D, stride_I);
    else
        ext_merge_kernel<4><<<(unsigned)nq, MERGE_THREADS, 0, st>>>(metric, Dl, Ill, nlists, nq, k, D, Io, stride_D, stride_I);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

int rag_topk_merge_dev(int metric, int nlists, int64_t nq, int k, const float* Dl, const int64_t* Il,
                       float* D, int64_t* I, void* stream) {
    return rag_topk_merge_strided_dev(metric, nlists, nq, k, Dl, Il, nq * (int64_t)k, nq * (int64_t)k, D, I, stream);
}

int rag_flat_profiling(rag_flat_t* h, int enable) {
    RAG_REQUIRE_INIT();
    if (!h) return fail(RAG_ERR_INVALID_ARG, "bad handle");
    std::lock_guard<std::mutex> lk(h->mu);
    h->prof = enable != 0;
    return RAG_OK;
}

int rag_flat_timing(rag_flat_t* h, int64_t* n_launches, double* total_ms, double* bytes_per_launch) {
    RAG_REQUIRE_INIT();
    if (!h) return fail(RAG_ERR_INVALID_ARG, "bad handle");
    std::lock_guard<std::mutex> lk(h->mu);
    int64_t cnt = 0;
    double sum = 0.0;
    for (auto& p : h->ev_used) {
        RAG_HIP(hipEventSynchronize(p.second));
        float ms = 0.f;
        RAG_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        sum += ms;
        ++cnt;
        h->ev_free.push_back(p);
    }
    h->ev_used.clear();
    double bytes = (double)h->ndev * h->d * h->esz;
    if (h->shadow && !h->shadow->ev_used.empty()) {  // prefiltered searches: the pass over the bfloat16 copy
        rag_flat* sh = h->shadow;
        for (auto& p : sh->ev_used) {
            RAG_HIP(hipEventSynchronize(p.second));
            float ms = 0.f;
            RAG_HIP(hipEventElapsedTime(&ms, p.first, p.second));
            sum += ms;
            ++cnt;
            sh->ev_free.push_back(p);
        }
        sh->ev_used.clear();
        bytes = (double)sh->ndev * sh->d * sh->esz;
    }
    if (n_launches) *n_launches = cnt;
    if (total_ms) *total_ms = sum;
    if (bytes_per_launch) *bytes_per_launch = bytes;
    return RAG_OK;
}

}  // extern "C"
