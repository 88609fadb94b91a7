// flat_wide.hip -- exact top-k of a medium query batch (9..64 queries per pass) over a row-major matrix:
// rows streamed from HBM straight into 16x16 MFMA operands, no LDS staging of the matrix and no
// barrier after the queries are in LDS -- every wave runs its own load -> multiply pipeline.
//
// Same reference call as flat.hip / flat_stream.hip (index.search(q, k), vector_storage.py:196, on the
// faiss IndexFlatIP built at :79 / :146 / :151), for the batch sizes where neither the
// 4x4 stream kernel (query operands re-read from LDS per 4 rows) nor the LDS tile kernel (load,
// LDS and MFMA phases in lock step: 5.8 ms for 64 queries over 10M x 384 against 3.1 ms of matrix
// math) is near a roofline.
//
// Operand layout.  v_mfma_f32_16x16x4_f32 wants lane l to hold row l%16 at k index l/16; a load of
// that shape (16 rows x 64 B per instruction) reaches only 5.8 TB/s, whole cache lines (8 rows x
// 128 B per instruction) reach 7.1 TB/s (benchmarks/membw_quad.hip).  So a step loads two such
// blocks -- LA: lane l -> row l%8, chunk l/8;  LB: lane l -> row 8 + l%8, chunk (l/8)^1 -- and a
// per-lane select on lane bit 3 gives two operands without any lane exchange:
//     in1 = bit3 ? LB : LA  = row l%16,        chunk 2*(l/16)      (even chunks of the step)
//     in2 = bit3 ? LA : LB  = row (l%16+8)%16, chunk 2*(l/16) + 1  (odd chunks, rows rotated by 8)
// in1 accumulates into acc1 and in2 into acc2 against the matching query chunks; the rotation only
// means that acc2 holds row p where acc1 holds row (p+8)%16, i.e. in lane l^32, and the tile
// epilogue adds acc1 + xor32(acc2).  The four values of a lane's 16-byte chunk go to four
// instructions (k = 4*chunk + e for e = 0..3); bfloat16 rows feed v_mfma_f32_16x16x32_bf16 with
// the whole chunk (8 values) at once.  The K reduction happens inside the matrix core: no fold.
//
// Selection: this kernel keeps no lists.  SAMPLE mode visits every 32nd 16-row tile and records
// each block's best score per query; the j-th largest of those block maxima is the query's
// threshold (equal to the j-th best sampled row unless two of the best j fall to one block, which
// only lowers the threshold).  FILTER mode appends every row scoring >= the threshold to the
// query's buffer; flat.hip selects the exact top-k from the buffer and redoes a query whose
// count left [k, cap] with the list-mode tile kernel.
#include "common.h"
#include "flat_stream.h"
#include "topk.h"

#include <algorithm>


namespace rag {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WIDE_MAX_THREADS = 1024;
#ifndef RAG_WIDE_NB_HI
#define RAG_WIDE_NB_HI 0  // A/B builds: ring depth for three and four query tiles (0 = default)
#endif

// Step buffers per wave: NB-1 steps (of 2 KiB) are in flight ahead of the one being multiplied.
// Up to two query tiles the registers allow a deeper ring, and the kernel needs it: with few
// matrix instructions per step the waves mostly wait for memory (measured 2.7 us under load).
__host__ __device__ constexpr int wide_nb(int NT) { return RAG_WIDE_NB_HI > 0 && NT >= 3 ? RAG_WIDE_NB_HI : (NT <= 2 ? 4 : 3); }
// 128-byte steps per row, a multiple of NB
__host__ __device__ inline int wide_nstep(int dch, int nb) { return ((dch + 8 * nb - 1) / (8 * nb)) * nb; }

// n >= 16.  cand_*: SAMPLE [nq][wpad] (one entry per block), FILTER [nq][cap].
template <int NT, int METRIC, bool BF, bool SAMPLE, int NB>
__global__ __launch_bounds__(NT >= 3 ? 768 : WIDE_MAX_THREADS) void flat_wide_kernel(
    const u32x4* __restrict__ X, long long n, int dch, const u32x4* __restrict__ Q, int nq,
    float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm, long long tile_step, const float* __restrict__ thr_q,
    unsigned int* __restrict__ counts, int cap, int wpad) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u32x4* q_lds = reinterpret_cast<u32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwv = blockDim.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const bool hi8 = (lane & 8) != 0;
    const int nstep = wide_nstep(dch, NB);
    const int npair = nstep * 4;            // chunk pairs per (padded) row
    const int plane = NT * npair * 16;      // u32x4 per plane (even / odd chunks)

    // queries: chunk 2*pi + P of query 16t + j at q_lds[P*plane + (t*npair + pi)*16 + j]: a step's
    // read of one plane is lane-linear.  Missing queries and chunks past the row end are zeros.
    for (int i = tid; i < 2 * plane; i += blockDim.x) {
        const int P = i / plane, rem = i - P * plane;
        const int jq = rem & 15, pi = (rem >> 4) % npair, t = (rem >> 4) / npair;
        const int q = 16 * t + jq, c = 2 * pi + P;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < nq && c < dch) v = Q[(long long)q * dch + c];
        q_lds[i] = v;
    }
    __syncthreads();

    float thr[NT], qn[NT], mx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = 16 * t + r;
        thr[t] = (!SAMPLE && q < nq) ? thr_q[q] : FLT_MAX;
#if defined(RAG_WIDE_EXP) && (RAG_WIDE_EXP & 1)
        thr[t] = FLT_MAX;  // ablation build: the scores are meaningless, append nothing
#endif
        qn[t] = (METRIC == RAG_METRIC_L2 && q < nq) ? qnorm[q] : 0.f;
        mx[t] = -FLT_MAX;
    }

    constexpr long long TR = 16;
    const long long W = (long long)gridDim.x * nwv;
    const long long gw = (long long)blockIdx.x * nwv + wid;
    const long long ntiles = ((n + TR - 1) / TR + tile_step - 1) / tile_step;
    const long long trows = TR * tile_step;

    // wave-uniform base + constant per-lane offsets (see flat_stream.hip)
    const char* Xb = reinterpret_cast<const char*>(X);
    const int cA = lane >> 3, cB = (lane >> 3) ^ 1;
    const unsigned int voffA = ((unsigned int)(lane & 7) * (unsigned int)dch + (unsigned int)cA) * 16u;
    const unsigned int voffB = ((unsigned int)(8 + (lane & 7)) * (unsigned int)dch + (unsigned int)cB) * 16u;

    // Issue side: next (tile, step) to load.  EVERY path issues exactly two loads (past the last
    // tile: the last tile again; past the row end: the row's last chunk, zeroed when it is used),
    // so the compiler can count outstanding loads (s_waitcnt vmcnt(N)) instead of draining them.
    long long ti = gw;
    int si = 0;
    auto tile_base = [&](long long t) {
        t = t < ntiles ? t : ntiles - 1;
        long long r0 = t * trows;
        r0 = r0 < n - TR ? r0 : n - TR;
        return Xb + r0 * dch * 16;
    };
    const char* isb = tile_base(ti);
    const unsigned int rowA = (unsigned int)(lane & 7) * (unsigned int)dch, rowB = rowA + 8u * (unsigned int)dch;
    auto issue = [&](u32x4& a, u32x4& b) {
        const int c0 = si << 3;
        if (c0 + 8 <= dch) {
            const char* sb = isb + (long long)c0 * 16;
            a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sb + voffA));
            b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sb + voffB));
        } else {  // the step the rows end in, or padding
            const int ca = c0 + cA < dch ? c0 + cA : dch - 1, cb = c0 + cB < dch ? c0 + cB : dch - 1;
            a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(isb + (rowA + (unsigned int)ca) * 16u));
            b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(isb + (rowB + (unsigned int)cb) * 16u));
        }
        if (++si == nstep) {
            si = 0;
            ti += W;
            isb = tile_base(ti);
        }
    };
    u32x4 bufA[NB], bufB[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        bufA[p] = u32x4{0u, 0u, 0u, 0u};
        bufB[p] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int p = 0; p + 1 < NB; ++p) issue(bufA[p], bufB[p]);

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (long long t0 = gw; t0 < ntiles; t0 += W) {
        const long long row0 = t0 * trows;
        const long long row0w = row0 < n - TR ? row0 : n - TR;
        const long long rowl = row0w + 4 * g;  // first of this lane's four rows
        f32x4 xn4 = zero4;
        if (METRIC == RAG_METRIC_L2) {
#pragma unroll
            for (int v = 0; v < 4; ++v) xn4[v] = xnorm[rowl + v];  // in flight behind the tile's loads
        }
        f32x4 acc1[NT], acc2[NT];
        for (int s0 = 0; s0 < nstep; s0 += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                issue(bufA[(u + NB - 1) % NB], bufB[(u + NB - 1) % NB]);
                u32x4 xa = bufA[u], xb = bufB[u];
                const int c0 = (s0 + u) << 3;
                if (c0 + 8 > dch) {  // chunks past the row end count as zeros (their loads were clamped)
                    if (c0 + cA >= dch) xa = u32x4{0u, 0u, 0u, 0u};
                    if (c0 + cB >= dch) xb = u32x4{0u, 0u, 0u, 0u};
                }
                u32x4 in1, in2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    in1[e] = hi8 ? xb[e] : xa[e];
                    in2[e] = hi8 ? xa[e] : xb[e];
                }
                u32x4 q1[NT], q2[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int o = t * npair * 16 + (s0 + u) * 64 + lane;
                    q1[t] = q_lds[o];
                    q2[t] = q_lds[plane + o];
                }
                const bool first = u == 0 && s0 == 0;  // C = 0: the accumulators are never cleared
#if defined(RAG_WIDE_EXP) && (RAG_WIDE_EXP & 1)
                // ablation build (benchmarks/build_variant.sh): loads and LDS reads only, no matrix instructions
                asm volatile("" ::"v"(in1), "v"(in2));
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    asm volatile("" ::"v"(q1[t]), "v"(q2[t]));
                    if (first) acc1[t] = acc2[t] = zero4;
                }
                if (false)
#endif
                if (!BF) {
                    if (first) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(in1[0]), __uint_as_float(q1[t][0]), zero4, 0, 0, 0);
                            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(in2[0]), __uint_as_float(q2[t][0]), zero4, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(in1[0]), __uint_as_float(q1[t][0]), acc1[t], 0, 0, 0);
                            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(in2[0]), __uint_as_float(q2[t][0]), acc2[t], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int e = 1; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(in1[e]), __uint_as_float(q1[t][e]), acc1[t], 0, 0, 0);
                            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(in2[e]), __uint_as_float(q2[t][e]), acc2[t], 0, 0, 0);
                        }
                } else {
                    const bf16x8 a1 = __builtin_bit_cast(bf16x8, in1), a2 = __builtin_bit_cast(bf16x8, in2);
                    if (first) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, q1[t]), zero4, 0, 0, 0);
                            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, q2[t]), zero4, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, q1[t]), acc1[t], 0, 0, 0);
                            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, q2[t]), acc2[t], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ---- tile epilogue: lane l holds rows rowl..rowl+3 of query 16t + l%16.  Only the
        // window-shifted last tile has rows that belong to another tile (uniform branch).
        const bool shifted = row0w != row0;
        bool any = false;
        float key[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float tot = acc1[t][v] + xor_lane<32>(acc2[t][v]);
                key[t][v] = METRIC == RAG_METRIC_IP ? tot : (2.f * tot - xn4[v]) - qn[t];
            }
            float best;
            if (!shifted) {
                best = fmaxf(fmaxf(key[t][0], key[t][1]), fmaxf(key[t][2], key[t][3]));
            } else {
                best = -FLT_MAX;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (rowl + v >= row0) best = fmaxf(best, key[t][v]);
            }
            if (SAMPLE) {
                if (16 * t + r < nq) mx[t] = fmaxf(mx[t], best);
            } else {
                any = any || best >= thr[t];  // thr = FLT_MAX for a missing query
            }
        }
        if (!SAMPLE) {
            if (__ballot(any) != 0ULL) {  // rare: append every passing row of this lane's queries
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int q = 16 * t + r;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        if (q < nq && rowl + v >= row0 && key[t][v] >= thr[t]) {
                            const unsigned int slot = atomicAdd(&counts[q], 1u);
                            if (slot < (unsigned int)cap) {
                                const long long o = (long long)q * cap + slot;
                                cand_key[o] = key[t][v];
                                cand_id[o] = (int)(rowl + v);
                            }
                        }
                    }
                }
            }
        }
    }

    if (SAMPLE) {
        // one best score per BLOCK and query (the waves' maxima meet in LDS, over the query area):
        // 256 values per query keep the threshold kernel at a few microseconds (3072 per-wave values
        // cost it 36 us); blocks that saw no row write -FLT_MAX
        __syncthreads();  // every wave is done with the queries
        float* smax = reinterpret_cast<float*>(smem_raw);  // [nwv][NT*16]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float m = mx[t];
            m = fmaxf(m, xor_lane<16>(m));
            m = fmaxf(m, xor_lane<32>(m));
            if (lane < 16) smax[wid * (NT * 16) + 16 * t + lane] = m;
        }
        __syncthreads();
        for (int q = tid; q < NT * 16 && q < nq; q += blockDim.x) {
            float m = -FLT_MAX;
            for (int w = 0; w < nwv; ++w) m = fmaxf(m, smax[w * (NT * 16) + q]);
            const long long o = (long long)q * wpad + blockIdx.x;
            cand_key[o] = m;
            cand_id[o] = m > -FLT_MAX ? (int)blockIdx.x : IdPad<int>::v();
        }
    }
}

template <int NT, bool BF, bool SAMPLE, int NB>
int launch_nb(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric, float* cand_key,
              int* cand_id, const float* xnorm, const float* qnorm, const float* thr_q, unsigned int* counts, hipStream_t st) {
    auto fn = metric == RAG_METRIC_IP ? flat_wide_kernel<NT, RAG_METRIC_IP, BF, SAMPLE, NB>
                                      : flat_wide_kernel<NT, RAG_METRIC_L2, BF, SAMPLE, NB>;
    if (p.smem > 48 * 1024)
        RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)p.smem));
    fn<<<(unsigned)p.blocks, (unsigned)(p.wpb * 64), p.smem, st>>>(
        reinterpret_cast<const u32x4*>(X), n, dch, reinterpret_cast<const u32x4*>(Q), nq, cand_key, cand_id, xnorm, qnorm,
        (long long)(SAMPLE ? p.sample_step : 1), thr_q, counts, p.cap, p.wpad);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

// the ring of the plan: the default depth of the tile count, or 3 where up to two query tiles ask for it
template <int NT, bool BF, bool SAMPLE>
int launch(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric, float* cand_key,
           int* cand_id, const float* xnorm, const float* qnorm, const float* thr_q, unsigned int* counts, hipStream_t st) {
    if constexpr (NT <= 2 && wide_nb(NT) != 3) {
        if (p.nb == 3)
            return launch_nb<NT, BF, SAMPLE, 3>(p, X, n, dch, Q, nq, metric, cand_key, cand_id, xnorm, qnorm, thr_q, counts, st);
    }
    return launch_nb<NT, BF, SAMPLE, wide_nb(NT)>(p, X, n, dch, Q, nq, metric, cand_key, cand_id, xnorm, qnorm, thr_q, counts, st);
}

template <bool SAMPLE>
int launch_any(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric,
               float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, const float* thr_q,
               unsigned int* counts, hipStream_t st) {
    if (nq < 1 || nq > 16 * p.NT) return fail(RAG_ERR_INVALID_ARG, "wide kernel: nq=%d beyond the planned %d", nq, 16 * p.NT);
    if (n < 16) return fail(RAG_ERR_INVALID_ARG, "wide kernel: needs at least 16 rows");
    if (metric == RAG_METRIC_L2 && (xnorm == nullptr || qnorm == nullptr))
        return fail(RAG_ERR_INVALID_ARG, "wide kernel: L2 needs row and query norms");
#define RAG_WIDE_CASE(NTV)                                                                                          \
    case NTV:                                                                                                       \
        return p.bf16 ? launch<NTV, true, SAMPLE>(p, X, n, dch, Q, nq, metric, cand_key, cand_id, xnorm, qnorm,     \
                                                  thr_q, counts, st)                                                \
                      : launch<NTV, false, SAMPLE>(p, X, n, dch, Q, nq, metric, cand_key, cand_id, xnorm, qnorm,    \
                                                   thr_q, counts, st);
    switch (p.NT) {
        RAG_WIDE_CASE(1)
        RAG_WIDE_CASE(2)
        RAG_WIDE_CASE(3)
        RAG_WIDE_CASE(4)
    }
#undef RAG_WIDE_CASE
    return fail(RAG_ERR_INVALID_ARG, "wide kernel: %d query tiles not built", p.NT);
}

}  // namespace

bool wide_plan(long long n, long long nq, int dch, bool bf16, int j, int cap, int sample_step, WidePlan* p) {
    const Options& opt = options();
    const long long max_nq = opt.wide_max_nq > 0 ? std::min<long long>(opt.wide_max_nq, 64) : 64;
    if (opt.wide_max_nq < 0 || nq < 1 || nq > max_nq || n < 16 || j < 1 || j > 32) return false;
    p->NT = (int)((nq + 15) / 16);
    p->bf16 = bf16;
    p->cap = cap;
    p->sample_step = sample_step < 1 ? 1 : sample_step;
    // Ring depth: rows are padded to a multiple of NB 128-byte steps.  Up to two query tiles the default is 4;
    // rows of 6 steps (384 bfloat16 values -- the copy of rag_flat_prefilter) then run 8.  The padding steps are
    // cheap (their loads re-read a cached chunk), so the shorter ring only pays with two query tiles (10M x 384
    // bf16: 32 queries 1.319 ms against 1.381, 16 queries 1.439 against 1.399; benchmarks/ab/run_prefilter_mid.sh).
    p->nb = wide_nb(p->NT);
    if (p->NT == 2 && wide_nstep(dch, 3) < wide_nstep(dch, p->nb)) p->nb = 3;
    if (opt.wide_nb == 3 && p->NT <= 2) p->nb = 3;
    if (opt.wide_nb == 4 && p->NT <= 2) p->nb = 4;
    const int nstep = wide_nstep(dch, p->nb);
    p->smem = (size_t)2 * p->NT * nstep * 4 * 16 * 16;
    if (p->smem > 160 * 1024) return false;  // long rows: the tile kernel streams its queries instead
    // Waves per CU the register budget admits (launch bounds of the kernel): 24 for one query tile,
    // 16 for two, 12 above; in blocks of a multiple of 4 waves (every SIMD the same load), as many
    // blocks as the LDS-resident queries allow.  Three and four tiles: one block of 12 waves.
    const int lds_bpc = std::max(1, (int)((160 * 1024) / p->smem));
    int bpc;
    if (p->NT >= 3) {
        bpc = 1;
        p->wpb = 12;
    } else if (p->NT == 2) {
        bpc = lds_bpc >= 4 ? 4 : (lds_bpc >= 2 ? 2 : 1);
        p->wpb = 16 / bpc;
    } else {
        bpc = lds_bpc >= 6 ? 6 : (lds_bpc >= 3 ? 3 : (lds_bpc >= 2 ? 2 : 1));
        p->wpb = bpc == 6 ? 4 : (bpc == 3 ? 8 : (bpc == 2 ? 12 : 16));
    }
    if (opt.wide_wpb > 0) p->wpb = (int)std::min<int64_t>(p->NT >= 3 ? 12 : 16, opt.wide_wpb);
    const long long ntiles = (n + 15) / 16;
    long long blocks = std::min<long long>((long long)num_cus() * bpc, (ntiles + p->wpb - 1) / p->wpb);
    p->blocks = (int)std::max<long long>(blocks, 1);
    p->wpad = p->blocks;  // sample pass: one value per block and query
    return true;
}

int wide_sample(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st) {
    return launch_any<true>(p, X, n, dch, Q, nq, metric, cand_key, cand_id, xnorm, qnorm, nullptr, nullptr, st);
}

int wide_filter(const WidePlan& p, const void* X, long long n, int dch, const void* Q, int nq, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, const float* thr_q,
                unsigned int* counts, hipStream_t st) {
    if (thr_q == nullptr || counts == nullptr || p.cap < 1)
        return fail(RAG_ERR_INVALID_ARG, "wide kernel: filter pass needs thresholds, counters and a capacity");
    return launch_any<false>(p, X, n, dch, Q, nq, metric, cand_key, cand_id, xnorm, qnorm, thr_q, counts, st);
}

}  // namespace rag
