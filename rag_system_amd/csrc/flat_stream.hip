// flat_stream.hip -- exact top-k of a small query batch (nq <= 16) over a row-major matrix, rows
// streamed from HBM straight into MFMA operands (no LDS staging of the matrix).
//
// Replaces, for 1..16 queries per call, the same reference call as flat.hip: index.search(q, k),
// vector_storage.py:196 (faiss IndexFlatIP built at :79 / :146 / :151), one call per query in the
// reference; SURVEY.md section 8 row (a3), batched by search_engine.search_faiss_batch.
//
// Why a third kernel: with <= 16 queries every matrix element is used by exactly one matrix
// instruction, so the LDS tile ring of flat_mfma_kernel.h only adds latency (its load, LDS and MFMA
// phases run in lock step: 3.1-3.6 ms per pass over 10M x 384 against the 2.2 ms the bytes cost),
// while the scan kernel's per-query VALU work does not scale past two queries.  Here a wave loads
// 4 rows x 256 B per instruction in exactly the operand layout of v_mfma_f32_4x4x1_16b_f32
// (lane 4b+i = row i, 16-byte chunk b; benchmarks/membw_quad.hip: this pattern streams at
// 7.0-7.1 TB/s like any other) and feeds each loaded register to the matrix core once per group of
// 4 queries: block b of the instruction accumulates D_b[i][j] += x[row i][k_b] * q[query j][k_b],
// 16 independent k slices at a time, 64 flops per cycle and SIMD -- 0.2 ms per 4 queries for
// 10M x 384, hidden under the stream.  bfloat16 rows use v_mfma_f32_4x4x4_16b_bf16 the same way
// (a lane's 8 values = 2 instructions).  A tile ends with a fold of the 16 k-slice partials over
// lane bits 2..5 (DPP / permlane, no LDS): every lane then holds one (row, query) score.
//
// Numerics: fp32 products and accumulation inside the matrix core (each slice an fmaf chain in k
// order, slices added in a fixed tree): within the 1e-5 bar of the float64 oracle, independent of
// the launch shape.  Selection is exact (same lists / filter buffers as the other kernels).
#include "common.h"
#include "flat_stream.h"
#include "topk.h"

#include <algorithm>

#ifndef RAG_STREAM_NB
#define RAG_STREAM_NB 3  // segment buffers per wave: NB-1 segments in flight ahead of the one being multiplied
#endif
// The light shapes (up to 4 queries, k <= 64, or filter mode) are also built with a ring of 2 and run
// ONE block per CU: with little work per segment a wave only waits for its loads, and 4 waves x one
// or two 4 KiB segments in flight per CU is where HBM delivers most -- 10M x 384, nq = 1: 7.23 TB/s
// against 7.10 with 16 waves x 3 buffers; 8 or 16 KiB more in flight per wave only queue in the
// memory system (benchmarks/ab/run_stream_variants.sh, run_nb2_matrix.sh).  Shapes with more work per
// row (8+ queries, long lists) need the other waves of a SIMD to cover it and keep the deep ring.

namespace rag {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef RAG_STREAM_THREADS
#define RAG_STREAM_THREADS 256
#endif
constexpr int ST_THREADS = RAG_STREAM_THREADS;
constexpr int ST_WAVES = ST_THREADS / 64;
constexpr int NB_DEFAULT = RAG_STREAM_NB;
constexpr int NB_LIGHT = 2;

// Sum N = 4*RG per-lane values over lane bits 5..2 (the 16 k slices), halving the value count
// at each step while it lasts: afterwards v[0] of lane l is the total of item (l >> (6-log2 N)).
// Distances 32 and 16 fold a pair of registers with ONE v_permlane{32,16}_swap (the halves that
// have to change lanes trade places) and one add -- no selects.
template <int N, int HALF, int DIST>
__device__ __forceinline__ void fold_slices(float (&v)[N], int lane) {
    if constexpr (DIST >= 4) {
        if constexpr (HALF >= 1) {
            if constexpr (DIST >= 16) {
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const int a = __float_as_int(v[i]), c = __float_as_int(v[i + HALF]);
                    if constexpr (DIST == 32) {
                        const auto p = __builtin_amdgcn_permlane32_swap(a, c, false, false);
                        v[i] = __int_as_float(p[0]) + __int_as_float(p[1]);
                    } else {
                        const auto p = __builtin_amdgcn_permlane16_swap(a, c, false, false);
                        v[i] = __int_as_float(p[0]) + __int_as_float(p[1]);
                    }
                }
            } else {
                const bool upper = (lane & DIST) != 0;
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const float keep = upper ? v[i + HALF] : v[i];
                    const float send = upper ? v[i] : v[i + HALF];
                    v[i] = keep + xor_lane<DIST>(send);
                }
            }
            fold_slices<N, HALF / 2, DIST / 2>(v, lane);
        } else {
            v[0] += xor_lane<DIST>(v[0]);
            fold_slices<N, 0, DIST / 2>(v, lane);
        }
    }
}

// ---- fused merge of the per-block lists (1-2 queries, k <= 16, <= 512 blocks): the block that finishes
// LAST merges cand[q][0..nlists)[k] and writes D / I itself -- the same result as flat_merge_kernel
// (flat.hip), bit for bit, without a second launch (a launch gap of ~6 us plus ~12 us of a 1024-thread
// kernel behind a 2.1 ms pass).  Hand-off: every list entry is stored write-through (sc1: relaxed
// agent-scope atomic stores, so no block needs a release fence), every storing wave drains its stores,
// the block's barrier, one lane draws a ticket; the last arriver makes ONE agent-scope acquire and reads
// the lists with plain loads.
// The merge itself stays in registers and LDS: thread t loads the whole lists t and t + 256 in ONE
// round trip (2k independent loads in flight), the k-th largest head T of the first 64 lists bounds the
// result from below (k entries are >= T) and is found by rank counting over those heads in LDS (no
// sorting network, two barriers); the entries >= T -- prefixes of the lists whose head is >= T, a few
// times k of them -- are appended to an LDS buffer and placed by rank counting again.  More than FUSE_CAP such entries
// (long tie runs): an exact k-way merge over the registers instead, one block-wide maximum per output.
constexpr int FUSE_LPT = 2;                       // lists per thread
constexpr int FUSE_MAX_LISTS = FUSE_LPT * ST_THREADS;
constexpr int FUSE_MAX_K = 16;
constexpr int FUSE_CAP = 512;
constexpr size_t FUSE_SMEM = FUSE_MAX_LISTS * 4 + FUSE_CAP * 8 + 64;

__device__ __forceinline__ unsigned long long fuse_pack(float key, int id) {
    unsigned int u = __float_as_uint(key);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)id);
}

// KMAX: list entries held per list (k <= KMAX <= FUSE_MAX_K).  The merge runs once per launch, in one block, on cold
// code; every entry slot is unrolled code (loads, compares, candidate stores), and ten slots instead of sixteen make the
// kernel 1.0-1.2 us shorter at k = 10 (benchmarks/ab/run_r03_fuse_code_size.sh: 1M rows 0.2351 -> 0.2342 ms, 20k rows
// 28.4 -> 27.2 us), so the reference's default top_k gets its own instance.
template <int METRIC, int KMAX>
__device__ __forceinline__ void stream_fused_merge(unsigned char* smem, const float* cand_key, const int* cand_id,
                                                   int nlists, int nq, int k, float* __restrict__ D,
                                                   long long* __restrict__ I, long long id_base) {
    float* hv = reinterpret_cast<float*>(smem);                                                       // [FUSE_MAX_LISTS]
    unsigned long long* sv = reinterpret_cast<unsigned long long*>(smem + FUSE_MAX_LISTS * 4);        // [FUSE_CAP]
    unsigned long long* smax = reinterpret_cast<unsigned long long*>(smem + FUSE_MAX_LISTS * 4 + FUSE_CAP * 8);
    int* n_cand = reinterpret_cast<int*>(smax + 1);
    float* T_s = reinterpret_cast<float*>(smax + 2);
    const int tid = threadIdx.x;
    auto emit = [&](int q, int p, unsigned long long c) {
        const long long o = (long long)q * k + p;
        if (c != 0ULL) {
            unsigned int u = (unsigned int)(c >> 32);
            u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
            const float key = __uint_as_float(u);
            D[o] = METRIC == RAG_METRIC_IP ? key : -key;
            I[o] = id_base + (long long)(0xFFFFFFFFu - (unsigned int)(c & 0xFFFFFFFFULL));
        } else {
            D[o] = METRIC == RAG_METRIC_IP ? -FLT_MAX : FLT_MAX;
            I[o] = -1LL;
        }
    };
    const int N4 = (nlists + 3) & ~3;
    for (int q = 0; q < nq; ++q) {
        const float* ck = cand_key + (long long)q * nlists * k;
        const int* ci = cand_id + (long long)q * nlists * k;
        // this thread's lists, whole, in one round trip; entries past k and lists past nlists are padding
        float key[FUSE_LPT][KMAX];
        int id[FUSE_LPT][KMAX];
#pragma unroll
        for (int u = 0; u < FUSE_LPT; ++u) {
            const int l = tid + u * ST_THREADS;
#pragma unroll
            for (int e = 0; e < KMAX; ++e) {
                key[u][e] = -FLT_MAX;
                id[u][e] = IdPad<int>::v();
                if (l < nlists && e < k) {
                    key[u][e] = ck[(long long)l * k + e];
                    id[u][e] = ci[(long long)l * k + e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FUSE_LPT; ++u) {
            const int l = tid + u * ST_THREADS;
            if (l < N4) hv[l] = key[u][0];
        }
        if (tid == 0) {
            *n_cand = 0;
            *T_s = -FLT_MAX;  // fewer than k lists: everything is a candidate
        }
        __syncthreads();
        // T: the k-th largest head among the FIRST 64 lists (rank counting by one wave, 64 LDS broadcasts).
        // Any k entries bound the result from below; a quarter of the heads instead of all of them makes T
        // a little lower and the candidate set a few times k larger, and saves the other 3/4 of a loop that
        // one thread per list walks alone (all heads: 18 us of a 22 us tail; this way ~1 us).
        const int ns = nlists < 64 ? nlists : 64;
        if (tid < ns) {
            const float h = key[0][0];
            int rank = 0;
            for (int j0 = 0; j0 < ns; j0 += 4) {
                const f32x4 h4 = *reinterpret_cast<const f32x4*>(hv + j0);  // same address in every lane: a broadcast
#pragma unroll
                for (int e = 0; e < 4; ++e) rank += (j0 + e < ns && (h4[e] > h || (h4[e] == h && j0 + e < tid))) ? 1 : 0;
            }
            if (rank == k - 1) *T_s = h;
        }
        __syncthreads();
        const float T = *T_s;
#pragma unroll
        for (int u = 0; u < FUSE_LPT; ++u) {
            // sorted: the entries >= T are a prefix; ONE counter update per list that has any
            int c = 0;
            bool more = true;
#pragma unroll
            for (int e = 0; e < KMAX; ++e) {
                more = more && key[u][e] >= T && id[u][e] != IdPad<int>::v();
                c += more ? 1 : 0;
            }
            if (c > 0) {
                const int slot = atomicAdd(n_cand, c);
#pragma unroll
                for (int e = 0; e < KMAX; ++e)
                    if (e < c && slot + e < FUSE_CAP) sv[slot + e] = fuse_pack(key[u][e], id[u][e]);
            }
        }
        __syncthreads();
        const int cnt = *n_cand;
        if (cnt <= FUSE_CAP) {  // uniform
            // place by rank: the composites are distinct (ids are), so the ranks are a permutation
            for (int i = tid; i < cnt; i += ST_THREADS) {
                const unsigned long long c = sv[i];
                int r = 0;
                for (int j = 0; j < cnt; ++j) r += sv[j] > c ? 1 : 0;
                if (r < k) emit(q, r, c);
            }
            for (int p = cnt + tid; p < k; p += ST_THREADS) emit(q, p, 0ULL);
            __syncthreads();  // sv / hv are reused by the next query
            continue;
        }
        // long tie runs: exact k-way merge over the registers
        int pos[FUSE_LPT];
#pragma unroll
        for (int u = 0; u < FUSE_LPT; ++u) pos[u] = 0;
        for (int p = 0; p < k; ++p) {
            if (tid == 0) *smax = 0ULL;
            __syncthreads();
            unsigned long long cur[FUSE_LPT], best = 0ULL;
#pragma unroll
            for (int u = 0; u < FUSE_LPT; ++u) {
                cur[u] = 0ULL;
#pragma unroll
                for (int e = 0; e < KMAX; ++e)
                    if (e == pos[u] && id[u][e] != IdPad<int>::v()) cur[u] = fuse_pack(key[u][e], id[u][e]);
                best = cur[u] > best ? cur[u] : best;
            }
            if (best != 0ULL) atomicMax(smax, best);
            __syncthreads();
            const unsigned long long w = *smax;
            if (tid == 0) emit(q, p, w);
#pragma unroll
            for (int u = 0; u < FUSE_LPT; ++u)
                if (w != 0ULL && cur[u] == w) ++pos[u];  // ids are unique: exactly one list of one thread
            __syncthreads();
        }
    }
}

// n >= 4*RG (the host plans the tile kernel for smaller shards): the last, partial wave tile is
// the window [n - 4*RG, n) with the rows below the tile's own first row masked out.
#ifndef RAG_STREAM_MINB
#define RAG_STREAM_MINB (G * RG >= 8 ? 3 : 4)  // blocks per CU the register budget must admit
#endif
#ifdef RAG_STREAM_STAMPS
// diagnostic build (benchmarks/build_variant.sh stamps flat_stream.hip -DRAG_STREAM_STAMPS; benchmarks/stream_stamps.py):
// lane 0 of wave 0 of every block records the constant 100 MHz clock at [0] kernel entry, [1] queries staged (first
// row loads in flight), [2] end of the row loop, [3] block list stored, [4] ticket drawn, [5] fused merge done (last
// block only).  Never in the product build.
__device__ unsigned long long g_stream_stamps[1024 * 8];
#define RAG_STAMP(i)                                                                              \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_stream_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define RAG_STAMP(i)
#endif
template <int G, int RG, int KS, int METRIC, bool BF, bool FILT, int NB>
__global__ __launch_bounds__(ST_THREADS, RAG_STREAM_MINB) void flat_stream_kernel(
    const u32x4* __restrict__ X, long long n, int dch, const u32x4* __restrict__ Q, int nq, int k,
    float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm, long long tile_step, const int* __restrict__ run_flag,
    const float* __restrict__ thr_q, unsigned int* __restrict__ counts, int cap, unsigned int* __restrict__ ticket,
    float* __restrict__ D, long long* __restrict__ I, long long id_base) {
    if (run_flag != nullptr && *run_flag == 0) return;  // uniform, before any barrier
    RAG_STAMP(0);
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u32x4* q_lds = reinterpret_cast<u32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the address math below
    const int b = lane >> 2, j = lane & 3;   // k slice (16-byte chunk within a 256-byte segment), query in group
    const int dchp = ((dch + 16 * NB - 1) / (16 * NB)) * (16 * NB);  // whole segments, a multiple of NB of them

    constexpr int NL = FILT ? 1 : 4 * G;
    using TK = WaveTopK<float, int, KS>;
    TK tk[NL];
    float thr_k[NL];
    int thr_i[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        tk[q].init();
        thr_k[q] = Lowest<float>::v();
        thr_i[q] = IdPad<int>::v();
    }
    float thr_lane[G], qn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int q = 4 * g + j;
        thr_lane[g] = FILT ? (q < nq ? thr_q[q] : FLT_MAX) : Lowest<float>::v();
        qn[g] = (METRIC == RAG_METRIC_L2 && q < nq) ? qnorm[q] : 0.f;
    }

    constexpr int N = 4 * RG;
    constexpr int LOGN = Log2<N>::v;
    const int item = (lane >> (6 - LOGN)) & (N - 1);
    const bool rep = (lane & ((1 << (6 - LOGN)) - 1) & ~3) == 0;
    constexpr long long TR = 4 * RG;
    const long long W = (long long)gridDim.x * ST_WAVES;
    const long long gw = (long long)blockIdx.x * ST_WAVES + wid;
    const long long ntiles = ((n + TR - 1) / TR + tile_step - 1) / tile_step;
    const long long trows = TR * tile_step;

    // Addresses: wave-uniform 64-bit base (tile, segment, row group) + a per-lane 32-bit offset
    // that never changes -> no vector address arithmetic in the loop.
    const char* Xb = reinterpret_cast<const char*>(X);
    const unsigned int voff = ((unsigned int)j * (unsigned int)dch + (unsigned int)b) * 16u;
    const long long rg_bytes = 64LL * dch;  // 4 rows
    const int nseg = dchp >> 4;             // 256-byte segments per row, padded to a multiple of NB

    // Register ring of NB segment buffers with STATIC indices (a copy of a register that is still
    // being loaded would wait for the load): step s multiplies buf[s % NB] after issuing the loads
    // of step s + NB - 1 into the buffer step s - 1 left free.  Every tile takes nseg (a multiple
    // of NB) steps, so a tile always starts at buf[0]; segments past the row end are zeros.
    // EVERY path through issue() issues exactly RG loads (past the last tile: the last tile again;
    // past the row end: the row's last chunk, zeroed when it is used), so the compiler can count
    // outstanding loads (s_waitcnt vmcnt(N)) instead of draining them.
    long long ti = gw;         // issue side: next (tile, segment) to load
    int si = 0;
    auto tile_base = [&](long long t) {
        t = t < ntiles ? t : ntiles - 1;
        long long r0 = t * trows;
        r0 = r0 < n - TR ? r0 : n - TR;
        return Xb + r0 * dch * 16;
    };
    const char* isb = tile_base(ti);  // base of tile ti
    auto issue = [&](u32x4(&dst)[RG]) {
        const int ci = si << 4;
        if (ci + 16 <= dch) {
            const char* sb = isb + (long long)ci * 16;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
                dst[rg] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sb + rg * rg_bytes + voff));
        } else {  // the segment the rows end in, or padding
            const int c = ci + b < dch ? ci + b : dch - 1;
            const unsigned int vo = ((unsigned int)j * (unsigned int)dch + (unsigned int)c) * 16u;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
                dst[rg] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(isb + rg * rg_bytes + vo));
        }
        if (++si == nseg) {
            si = 0;
            ti += W;
            isb = tile_base(ti);
        }
    };
    u32x4 buf[NB][RG];
#pragma unroll
    for (int p = 0; p < NB; ++p)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) buf[p][rg] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p + 1 < NB; ++p) issue(buf[p]);
    // queries (staged BEHIND the first row loads, so that the two round trips to memory overlap instead of
    // following each other: ~1.5 us per launch): chunk c of query 4g+jj at q_lds[(g*dchp + c)*4 + jj] -> one
    // lane-linear b128 read per (segment, group); missing queries and the chunks past the row end are zeros
    for (int i = tid; i < G * dchp * 4; i += ST_THREADS) {
        const int jj = i & 3, c = (i >> 2) % dchp, g = (i >> 2) / dchp;
        const int q = 4 * g + jj;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < nq && c < dch) v = Q[(long long)q * dch + c];
        q_lds[i] = v;
    }
    __syncthreads();
    RAG_STAMP(1);

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto bf_half = [](const u32x4& v, int e) {
        const unsigned int lo = v[2 * e], hi = v[2 * e + 1];
        return s16x4{(short)(lo & 0xFFFFu), (short)(lo >> 16), (short)(hi & 0xFFFFu), (short)(hi >> 16)};
    };
    for (long long t = gw; t < ntiles; t += W) {
        const long long row0 = t * trows;
        const long long row0w = row0 < n - TR ? row0 : n - TR;
        const long long row = row0w + item;
        float xn = 0.f;
        if (METRIC == RAG_METRIC_L2) xn = xnorm[row];  // in flight behind the tile's loads, used after them
        f32x4 acc[G][RG];
        for (int s0 = 0; s0 < nseg; s0 += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                issue(buf[(u + NB - 1) % NB]);
                u32x4 xc[RG];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) xc[rg] = buf[u][rg];
                if (((s0 + u) << 4) + 16 > dch) {  // uniform: the segment the rows end in, or padding
                    if (((s0 + u) << 4) + b >= dch) {  // chunks past the row end count as zeros
#pragma unroll
                        for (int rg = 0; rg < RG; ++rg) xc[rg] = u32x4{0u, 0u, 0u, 0u};
                    }
                }
                u32x4 qv[G];
#pragma unroll
                for (int g = 0; g < G; ++g) qv[g] = q_lds[(g * dchp + (s0 + u) * 16) * 4 + lane];
                // the first instruction of a tile starts from C = 0: the accumulators are never cleared
                const bool first = u == 0 && s0 == 0;
                if (!BF) {
                    if (first) {
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int rg = 0; rg < RG; ++rg)
                                acc[g][rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                    __uint_as_float(xc[rg][0]), __uint_as_float(qv[g][0]), zero4, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int rg = 0; rg < RG; ++rg)
                                acc[g][rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                    __uint_as_float(xc[rg][0]), __uint_as_float(qv[g][0]), acc[g][rg], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 1; e < 4; ++e)
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int rg = 0; rg < RG; ++rg)
                                acc[g][rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                    __uint_as_float(xc[rg][e]), __uint_as_float(qv[g][e]), acc[g][rg], 0, 0, 0);
                } else {
                    if (first) {
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int rg = 0; rg < RG; ++rg)
                                acc[g][rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(bf_half(xc[rg], 0), bf_half(qv[g], 0),
                                                                                  zero4, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int rg = 0; rg < RG; ++rg)
                                acc[g][rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(bf_half(xc[rg], 0), bf_half(qv[g], 0),
                                                                                  acc[g][rg], 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int rg = 0; rg < RG; ++rg)
                            acc[g][rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(bf_half(xc[rg], 1), bf_half(qv[g], 1),
                                                                              acc[g][rg], 0, 0, 0);
                }
            }
        }

        // ---- tile epilogue: fold the k slices, one (row, query) score per lane and group
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float v[N];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[rg * 4 + i] = acc[g][rg][i];
            fold_slices<N, N / 2, 32>(v, lane);
            const float key = METRIC == RAG_METRIC_IP ? v[0] : (2.f * v[0] - xn) - qn[g];
            const bool ok = rep && row >= row0 && 4 * g + j < nq && key >= thr_lane[g];
            const unsigned long long m = __ballot(ok);
            if (m != 0ULL) {
                if (FILT) {
                    if (ok) {
                        const int q = 4 * g + j;
                        const unsigned int slot = atomicAdd(&counts[q], 1u);
                        if (slot < (unsigned int)cap) {
                            const long long o = (long long)q * cap + slot;
                            cand_key[o] = key;
                            cand_id[o] = (int)row;
                        }
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if (m & (0x1111111111111111ULL << jj)) {
                            const int q = FILT ? 0 : 4 * g + jj;  // static after unrolling
                            wave_offer<float, int, KS>(tk[q], k, thr_k[q], thr_i[q], key, (int)row, ok && j == jj, lane);
                            if (j == jj) thr_lane[g] = thr_k[q];
                        }
                    }
                }
            }
        }
    }

    RAG_STAMP(2);
    if (FILT) return;
    // ---- merge the block's four wave lists through LDS (over the query area): every wave posts its
    // lists, wave w merges the queries q = w (mod 4) -- the tail of the kernel is a chain of sorted
    // inserts (~0.13 us each), four times shorter when the waves share it -- and writes one list
    // per real query: cand[q][block][k]
    __syncthreads();
    float* mk = reinterpret_cast<float*>(smem_raw);
    int* mi = reinterpret_cast<int*>(mk + ST_WAVES * NL * KS * 64);
#pragma unroll
    for (int q = 0; q < NL; ++q)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int o = ((wid * NL + q) * KS + s) * 64 + lane;
            mk[o] = tk[q].key[s];
            mi[o] = tk[q].id[s];
        }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        if ((q & (ST_WAVES - 1)) == wid && q < nq) {
            for (int dw = 1; dw < ST_WAVES; ++dw) {
                const int w = (wid + dw) & (ST_WAVES - 1);
                float ck[KS];
                int ci[KS];
                bool cv[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int o = ((w * NL + q) * KS + s) * 64 + lane;
                    ck[s] = mk[o];
                    ci[s] = mi[o];
                    cv[s] = s * 64 + lane < k;  // entries past a list's k-th cannot be among the best k of the union
                }
                // a sorted list: a handful of inserts, or one bitonic merge when many entries qualify (larger k)
                wave_offer_chunk<float, int, KS>(tk[q], k, thr_k[q], thr_i[q], ck, ci, cv, true, lane);
            }
            const long long base = ((long long)q * gridDim.x + blockIdx.x) * k;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s * 64 + lane;
                if (p < k) {
                    if (D != nullptr) {  // read by this launch's last block: write-through (sc1) stores
                        __hip_atomic_store(cand_key + base + p, tk[q].key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(cand_id + base + p, tk[q].id[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        cand_key[base + p] = tk[q].key[s];
                        cand_id[base + p] = tk[q].id[s];
                    }
                }
            }
        }
    }
    RAG_STAMP(3);
    if constexpr (G != 1 || KS != 1) return;  // the fused merge exists for 1-2 queries with k <= 16 only
    if (D == nullptr) return;  // uniform: the caller merges the lists with flat_merge_kernel
    // ---- fused merge: every storing wave drains its (write-through) stores, the block's barrier, then ONE
    // lane draws the ticket; the last arriver acquires and reads every block's lists
#ifndef RAG_FUSE_STRICT
    // Relaxed ticket, write-through (sc1) list stores drained by every wave before the block's barrier, ONE explicit
    // acquire by the last arriver, plain loads behind it (sc1 loads and no acquire: 18 us -- thousands of uncached
    // 4-byte reads of a 25 KB region against 200 cache lines).  This leans on what sc1 stores and s_waitcnt do on
    // gfx950 rather than on the HIP memory model; the model-conformant variant below (-DRAG_FUSE_STRICT) was
    // measured 5-9 us per launch slower (r03, benchmarks/ab/run_r03_fuse.sh: 2.1398 -> 2.1486 ms at 10M rows,
    // 0.2468 -> 0.2515 ms at 1M: the release half of every block's ticket draw writes the XCD's L2 back), so this
    // one stays, guarded by tests/test_flat_stream_gpu.py::test_fused_merge_never_reads_a_stale_list.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* last = reinterpret_cast<int*>(smem_raw + FUSE_SMEM - 16);
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool is_last = t == gridDim.x - 1;
        if (is_last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *last = is_last ? 1 : 0;
    }
#else
    // Inside the HIP memory model (build with -DRAG_FUSE_STRICT): the block's barrier orders every lane's list stores before lane 0's ticket
    // draw, which is an acq_rel read-modify-write at agent scope -- its release half publishes what happened
    // before it in this block (cumulativity across the barrier), its acquire half, in the block that draws the
    // last ticket, makes every earlier block's publication visible; the barrier behind it extends that to the
    // other lanes of the last block, which then read the lists with plain loads.
    __syncthreads();
    int* last = reinterpret_cast<int*>(smem_raw + FUSE_SMEM - 16);
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const bool is_last = t == gridDim.x - 1;
        if (is_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
        *last = is_last ? 1 : 0;
    }
#endif
    __syncthreads();
    RAG_STAMP(4);
    if (*last == 0) return;
    __syncthreads();  // *last is read by everyone before the merge reuses the LDS
    if (k <= 10) stream_fused_merge<METRIC, 10>(smem_raw, cand_key, cand_id, (int)gridDim.x, nq, k, D, I, id_base);
    else stream_fused_merge<METRIC, FUSE_MAX_K>(smem_raw, cand_key, cand_id, (int)gridDim.x, nq, k, D, I, id_base);
    RAG_STAMP(5);
}

struct Args {
    const void *X, *Q;
    long long n;
    int dch, nq, k, metric;
    float* cand_key;
    int* cand_id;
    const float *xnorm, *qnorm;
    const int* run_flag;
    const float* thr_q;
    unsigned int* counts;
    hipStream_t st;
    unsigned int* ticket;  // fused merge (list mode, D != nullptr): zero between launches
    float* D;
    long long* I;
    long long id_base;
};

template <int G, int RG, int KS, bool BF, bool FILT, int NB>
int launch_nb(const StreamPlan& p, const Args& a) {
    auto fn = a.metric == RAG_METRIC_IP ? flat_stream_kernel<G, RG, KS, RAG_METRIC_IP, BF, FILT, NB>
                                        : flat_stream_kernel<G, RG, KS, RAG_METRIC_L2, BF, FILT, NB>;
    if (p.smem > 48 * 1024)
        RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)p.smem));
    fn<<<(unsigned)p.blocks, ST_THREADS, p.smem, a.st>>>(
        reinterpret_cast<const u32x4*>(a.X), a.n, a.dch, reinterpret_cast<const u32x4*>(a.Q), a.nq, a.k, a.cand_key,
        a.cand_id, a.xnorm, a.qnorm, (long long)p.tile_step, a.run_flag, a.thr_q, a.counts, p.cap, a.ticket, a.D, a.I,
        a.id_base);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

template <int G, int RG, int KS, bool BF, bool FILT>
int launch(const StreamPlan& p, const Args& a) {
    if constexpr (G == 1 && RG == 4 && KS == 1) {  // the shapes stream_plan() may give the short ring
        if (p.NB == NB_LIGHT) return launch_nb<G, RG, KS, BF, FILT, NB_LIGHT>(p, a);
    }
    return launch_nb<G, RG, KS, BF, FILT, NB_DEFAULT>(p, a);
}

template <int G, int RG, bool BF>
int launch_mode(const StreamPlan& p, const Args& a) {
    if (p.filt) return launch<G, RG, 1, BF, true>(p, a);
    if constexpr (G == 1 && RG == 4) {  // 128 < k <= 256: four 256-entry lists per wave, one query group only
        if (p.KS == 4) return launch<G, RG, 4, BF, false>(p, a);
    }
    return p.KS == 1 ? launch<G, RG, 1, BF, false>(p, a) : launch<G, RG, 2, BF, false>(p, a);
}

template <int G, bool BF>
int launch_rg(const StreamPlan& p, const Args& a) {
    if constexpr (G < 4) {
        if (p.RG == 4) return launch_mode<G, 4, BF>(p, a);
    }
    return launch_mode<G, 2, BF>(p, a);  // 4 query groups x 4 row groups of accumulators do not fit 3 waves per SIMD
}

template <bool BF>
int launch_g(const StreamPlan& p, const Args& a) {
    switch (p.G) {
        case 1: return launch_rg<1, BF>(p, a);
        case 2: return launch_rg<2, BF>(p, a);
        default: return launch_rg<4, BF>(p, a);
    }
}

}  // namespace

bool stream_plan(long long n, long long nq, int k, int dch, bool bf16, bool filt, int cap, int tile_step,
                 StreamPlan* p) {
    const Options& opt = options();
    const long long max_nq = opt.stream_max_nq > 0 ? std::min<long long>(opt.stream_max_nq, 16) : 16;
    if (opt.stream_max_nq < 0 || nq < 1 || nq > max_nq || n < 1) return false;
    if (!filt && (k < 1 || k > 256 || (k > 128 && nq > 4))) return false;
    if (n < 16) return false;  // a wave tile is up to 16 rows; tiny shards go to the tile kernel
    p->G = nq <= 4 ? 1 : (nq <= 8 ? 2 : 4);
    p->KS = k <= 64 ? 1 : (k <= 128 ? 2 : 4);
    p->bf16 = bf16;
    p->filt = filt;
    p->cap = cap;
    p->tile_step = tile_step < 1 ? 1 : tile_step;
    p->RG = opt.stream_rg == 2 || opt.stream_rg == 4 ? (int)opt.stream_rg : (p->G == 4 ? 2 : 4);
    if (p->G == 4) p->RG = 2;
    if (p->KS == 4) p->RG = 4;  // the only shape built with 256-entry lists
    // ring depth: see NB_LIGHT above ("stream_nb": 2 / 3 force one where both are built)
    const bool light_built = p->G == 1 && p->RG == 4 && (filt || p->KS == 1);  // filter mode is built with KS = 1
    // measured (benchmarks/ab/run_nb_k.sh, 10M x 384): nq 1: k 10 / 32 / 64 = 2.142 / 2.157 / 2.175 ms against
    // 2.191-2.196; nq 4: k 10 2.173 against 2.197, k 20 a tie, k 32 and nq 3 k 64 lose 1.5-3 % (the one
    // wave of a SIMD also carries every list insert); filter pass nq 4 k 100: 2.318 against 2.354.
    // 1M rows: nq 1 k 10 0.240 against 0.246 ms; nq 4 k 30 0.307 against 0.265 (list warm-up)
    const long long list_load = nq * (long long)k;
    // rows of up to three 256-byte segments (384 bfloat16 values: the copy of rag_flat_prefilter): a tile is over
    // after three steps and the epilogue's fold + inserts stall a ring of two (10M x 384 bf16, nq 1, k 64:
    // 1.354 ms with the light shape, 1.086 with the deep ring = 7.07 TB/s; benchmarks/ab/run_prefilter_opts.sh)
    const bool short_rows = dch <= 48;
    const bool light_auto = !short_rows && (filt || (list_load <= 64 && (n >= (4LL << 20) || list_load <= 16)));
    p->NB = light_built && (opt.stream_nb == NB_LIGHT || (opt.stream_nb == 0 && light_auto)) ? NB_LIGHT : NB_DEFAULT;
    const int NB = p->NB;
    const int dchp = ((dch + 16 * NB - 1) / (16 * NB)) * (16 * NB);
    const size_t qbytes = (size_t)p->G * dchp * 64;
    const size_t mbytes = filt ? 0 : (size_t)ST_WAVES * 4 * p->G * p->KS * 64 * 8;
    p->smem = std::max(qbytes, mbytes);
    if (!filt && nq <= 2 && k <= FUSE_MAX_K) p->smem = std::max(p->smem, FUSE_SMEM);  // room for the fused merge of the block lists
    if (p->smem > 64 * 1024) return false;  // long rows: the tile kernel streams its queries instead
    const long long TR = 4 * p->RG;
    const long long ntiles = ((n + TR - 1) / TR + p->tile_step - 1) / p->tile_step;
    const long long bpc = opt.stream_bpc > 0 ? opt.stream_bpc
                          : p->NB == NB_LIGHT ? 1
                          // long lists for one or two queries: two blocks per CU (nq 1, k 100: 2.195 against
                          // 2.207 ms with four; 1M rows, k 90: 0.252 against 0.263; benchmarks/ab/run_less_inflight.sh)
                          : (p->G == 1 && nq <= 2) ? 2
                                                   : (p->G * p->RG >= 8 ? 3 : 4);  // = the kernel's launch bound
    long long blocks = std::min<long long>((long long)num_cus() * bpc, (ntiles + ST_WAVES - 1) / ST_WAVES);
    p->blocks = (int)std::max<long long>(blocks, 1);
    p->nlists = p->blocks;
    return true;
}

bool stream_can_fuse(const StreamPlan& p, long long nq, int k) {
    return !p.filt && nq <= 2 && k <= FUSE_MAX_K && p.blocks <= FUSE_MAX_LISTS && p.smem >= FUSE_SMEM &&
           options().stream_fuse >= 0;
}

int stream_launch(const StreamPlan& p, const void* X, long long n, int dch, const void* Q, int nq, int k, int metric,
                  float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st,
                  const int* run_flag, const float* thr_q, unsigned int* counts, unsigned int* ticket, float* D,
                  long long* I, long long id_base) {
    if (D != nullptr && (p.filt || nq > 2 || k > FUSE_MAX_K || p.blocks > FUSE_MAX_LISTS || ticket == nullptr ||
                         I == nullptr || p.smem < FUSE_SMEM || run_flag != nullptr))
        return fail(RAG_ERR_INVALID_ARG, "stream kernel: the fused merge serves list mode with 1-2 queries, k <= %d and <= %d blocks",
                    FUSE_MAX_K, FUSE_MAX_LISTS);
    if (nq < 1 || nq > 4 * p.G) return fail(RAG_ERR_INVALID_ARG, "stream kernel: nq=%d beyond the planned %d", nq, 4 * p.G);
    if (!p.filt && (k < 1 || k > 64 * p.KS)) return fail(RAG_ERR_INVALID_ARG, "stream kernel: k=%d beyond %d", k, 64 * p.KS);
    if (p.filt && (thr_q == nullptr || counts == nullptr || p.cap < 1))
        return fail(RAG_ERR_INVALID_ARG, "stream kernel: filter mode needs thresholds, counters and a capacity");
    if (metric == RAG_METRIC_L2 && (xnorm == nullptr || qnorm == nullptr))
        return fail(RAG_ERR_INVALID_ARG, "stream kernel: L2 needs row and query norms");
    const Args a{X, Q, n, dch, nq, k, metric, cand_key, cand_id, xnorm, qnorm, run_flag, thr_q, counts, st,
                 ticket, D, I, id_base};
    return p.bf16 ? launch_g<true>(p, a) : launch_g<false>(p, a);
}

}  // namespace rag

#ifdef RAG_STREAM_STAMPS
extern "C" int rag_debug_stream_stamps(unsigned long long* out, int n_words) {
    if (n_words > 1024 * 8) n_words = 1024 * 8;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rag::g_stream_stamps), (size_t)n_words * 8) != hipSuccess) return -1;
    return 0;
}
#endif
