// flat_mfma.hip -- large query batches: exact-f32 MFMA GEMM tile fused with per-query top-k.
//
// The one place the search path is a real dense contraction (BASELINE.json north_star):
// S[row, q] = sum_k X[row, k] * Q[q, k] for nq >= 5 queries, followed by top-k over rows.
// Uses v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain, no
// TF32/xf32 on gfx950), peak 157 TFLOP/s = the bound for nq >~ 40 (ridge ~20 flop/B).
//
// Tiling (one workgroup = NW waves, NW in {1,2,4,8}):
//   block tile   BM = 32*MT rows  x  BN = 32*NW queries,  K step BK = 16 floats
//   wave w       ALL BM rows x queries [32w, 32w+32): MT accumulator tiles of 32x32
//                -> every query column is owned by exactly one wave: its top-k list and
//                   threshold need no cross-wave synchronisation.
//   operands     A = X tile (row = lane&31, k = lane>>5), B = Q^T tile (query = lane&31);
//                C layout: column (lane&31) = query, 16 registers = rows
//                (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> a lane compares its 16 rows against
//                ITS query's threshold with plain v_cmp, no shuffles.
//   LDS          X and Q tiles double-buffered, rows padded to 20 floats (conflict-free
//                ds_read_b128 over 16 distinct rows); per-query sorted k-lists; 32-float
//                scratch per wave for the column transposition on a threshold hit.
//   pipeline     register-staged: global loads of step e+1 are issued before the MFMAs of
//                step e and written to the other LDS buffer after them (one barrier per step);
//                the K loop runs on across row tiles so there is no bubble between tiles.
//   grid         row_splits x query_tiles persistent blocks; block b works on row tiles
//                s, s+S, ...; blocks that share a row split (different query tiles) get
//                consecutive ids with equal b%8 so they share an XCD's L2 (speed only).
// Epilogue per row tile: ballot the lanes with a score above their query's current k-th
// key (rare once warm); for each such query dump its column to scratch, offer the 32 rows
// to the LDS-resident list through the wave-cooperative WaveTopK insert, refresh the
// threshold.  The score matrix is never written to HBM.
#include "common.h"
#include "flat_mfma.h"
#include "topk.h"

namespace {

using namespace rag;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;       // floats per K step
constexpr int LDS_LD = 20;   // padded row stride (floats) of the X / Q tiles in LDS

template <int NW, int MT, int METRIC>
__global__ __launch_bounds__(64 * NW) void flat_mfma_kernel(
    const float* __restrict__ X, long long n, int ld, const float* __restrict__ Qp, int nq, int k, int S,
    int col_tiles, float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm) {
    constexpr int T = 64 * NW;
    constexpr int BM = 32 * MT;
    constexpr int BN = 32 * NW;
    constexpr int XQ = (BM * 4) / T;  // 16-byte chunks of the X tile per thread per step
    constexpr int QQ = (BN * 4) / T;  // = 2
    static_assert((BM * 4) % T == 0 && QQ == 2, "tile / thread-count mismatch");

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);           // [2][BM][LDS_LD]
    float* Qs = Xs + 2 * BM * LDS_LD;                         // [2][BN][LDS_LD]
    float* lkey = Qs + 2 * BN * LDS_LD;                       // [BN][k]
    int* lid = reinterpret_cast<int*>(lkey + BN * k);         // [BN][k]
    float* scratch = reinterpret_cast<float*>(lid + BN * k);  // [NW][32]
    float* xn_s = scratch + NW * 32;                          // [BM] row norms (L2 only)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    int s, c;
    {
        const int b = blockIdx.x;
        if ((S & 7) == 0) {
            const int xcd = b & 7, t_ = b >> 3;
            c = t_ % col_tiles;
            s = (t_ / col_tiles) * 8 + xcd;
        } else {
            c = b % col_tiles;
            s = b / col_tiles;
        }
    }
    const int q0 = c * BN;
    const long long ntiles = (n + BM - 1) / BM;
    const long long my_tiles = s < ntiles ? (ntiles - s + S - 1) / S : 0;
    const int nk = (ld + BK - 1) / BK;
    const long long total = my_tiles * nk;

    // lists start as pads
    for (int i = tid; i < BN * k; i += T) {
        lkey[i] = Lowest<float>::v();
        lid[i] = IdPad<int>::v();
    }
    float thr = Lowest<float>::v();  // current k-th key of this lane's query
    float qn = 0.f;
    if (METRIC == RAG_METRIC_L2) {
        const int qi = q0 + 32 * wid + li;
        qn = qi < nq ? qnorm[qi] : 0.f;
    }

    f32x4 xr[XQ], qr[QQ];
    auto load_global = [&](long long e) {
        const long long ti = e / nk;
        const int ks = (int)(e - ti * nk);
        const long long row0 = (s + ti * S) * (long long)BM;
        const int k0 = ks * BK;
#pragma unroll
        for (int it = 0; it < XQ; ++it) {
            const int id = it * T + tid;
            const int r = id >> 2, cc = id & 3;
            long long row = row0 + r;
            row = row < n ? row : n - 1;
            const int kcol = k0 + 4 * cc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kcol < ld) {
                const f32x4* gp = reinterpret_cast<const f32x4*>(X + row * ld + kcol);
                // a row tile is re-read by the other query tiles of its row split through L2:
                // stream it non-temporally only when this block is its sole reader
                v = col_tiles == 1 ? __builtin_nontemporal_load(gp) : *gp;
            }
            xr[it] = v;
        }
#pragma unroll
        for (int it = 0; it < QQ; ++it) {
            const int id = it * T + tid;
            const int r = id >> 2, cc = id & 3;
            const int kcol = k0 + 4 * cc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kcol < ld && q0 + r < nq) v = *reinterpret_cast<const f32x4*>(Qp + (long long)(q0 + r) * ld + kcol);
            qr[it] = v;
        }
    };
    auto store_lds = [&](int buf) {
        float* xs = Xs + buf * BM * LDS_LD;
        float* qs = Qs + buf * BN * LDS_LD;
#pragma unroll
        for (int it = 0; it < XQ; ++it) {
            const int id = it * T + tid;
            *reinterpret_cast<f32x4*>(xs + (id >> 2) * LDS_LD + 4 * (id & 3)) = xr[it];
        }
#pragma unroll
        for (int it = 0; it < QQ; ++it) {
            const int id = it * T + tid;
            *reinterpret_cast<f32x4*>(qs + (id >> 2) * LDS_LD + 4 * (id & 3)) = qr[it];
        }
    };

    f32x16 acc[MT];
    using TK = WaveTopK<float, int, 1>;

    if (total > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();

    for (long long e = 0; e < total; ++e) {
        const int cur = (int)(e & 1);
        const long long ti = e / nk;
        const int ks = (int)(e - ti * nk);
        if (e + 1 < total) load_global(e + 1);
        if (ks == 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
        {
            const float* xs = Xs + cur * BM * LDS_LD;
            const float* qs = Qs + cur * BN * LDS_LD;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(qs + (32 * wid + li) * LDS_LD + kk * 8 + 4 * lh);
                f32x4 a[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    a[t] = *reinterpret_cast<const f32x4*>(xs + (32 * t + li) * LDS_LD + kk * 8 + 4 * lh);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][cc], bq[cc], acc[t], 0, 0, 0);
            }
        }
        if (ks == nk - 1) {
            // ---- epilogue of one row tile: threshold filter + list update
            const long long row0 = (s + ti * S) * (long long)BM;
            const bool partial = row0 + BM > n;
            if (METRIC == RAG_METRIC_L2) {
                // every wave stages the tile's row norms for itself (wave-private use below)
                __syncthreads();
                for (int i = tid; i < BM; i += T) {
                    const long long row = row0 + i;
                    xn_s[i] = row < n ? xnorm[row] : 0.f;
                }
                __syncthreads();
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const long long rowbase = row0 + 32 * t;
                float key[16];
                bool any = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[t][r];
                    if (METRIC == RAG_METRIC_L2) v = (2.f * v - xn_s[32 * t + ro]) - qn;
                    key[r] = v;
                    bool ok = v > thr;
                    if (partial) ok = ok && (rowbase + ro < n);
                    any = any || ok;
                }
                const unsigned long long m = __ballot(any);
                unsigned qm = (unsigned)(m | (m >> 32));
                while (qm) {
                    const int j = __ffs(qm) - 1;
                    qm &= qm - 1;
                    float* sc = scratch + wid * 32;
                    if (li == j) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[(r & 3) + 8 * (r >> 2) + 4 * lh] = key[r];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const float cs = sc[li];
                    const long long crow = rowbase + li;
                    TK tk;
                    const int lo = (32 * wid + j) * k;
                    tk.key[0] = lane < k ? lkey[lo + lane] : Lowest<float>::v();
                    tk.id[0] = lane < k ? lid[lo + lane] : IdPad<int>::v();
                    float thr_k;
                    int thr_i;
                    tk.kth(k, thr_k, thr_i);
                    wave_offer<float, int, 1>(tk, k, thr_k, thr_i, cs, (int)crow, lane < 32 && crow < n, lane);
                    if (lane < k) {
                        lkey[lo + lane] = tk.key[0];
                        lid[lo + lane] = tk.id[0];
                    }
                    if (li == j) thr = thr_k;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        if (e + 1 < total) store_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- write this block's list of every real query: cand[q][s][k]
    __syncthreads();
    for (int i = tid; i < BN * k; i += T) {
        const int ql = i / k, p = i - ql * k;
        const int qi = q0 + ql;
        if (qi < nq) {
            const long long o = ((long long)qi * S + s) * k + p;
            cand_key[o] = lkey[i];
            cand_id[o] = lid[i];
        }
    }
}

// ||x||^2 per row (fp32, same fmaf order on every call) for the L2 decomposition.
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ X, int ld, long long first,
                                                        long long n, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long W = (long long)gridDim.x * 4;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += W) {
        const float* row = X + (first + r) * ld;
        float ss = 0.f;
        for (int cidx = lane; cidx < ld; cidx += 64) ss = fmaf(row[cidx], row[cidx], ss);
        for (int dist = 32; dist >= 1; dist >>= 1) ss += __shfl_xor(ss, dist);
        if (lane == 0) out[first + r] = ss;
    }
}

template <int NW, int MT>
size_t smem_bytes(int k) {
    const size_t BM = 32 * MT, BN = 32 * NW;
    return (2 * BM * LDS_LD + 2 * BN * LDS_LD) * 4 + BN * (size_t)k * 8 + NW * 32 * 4 + BM * 4;
}

template <int NW, int MT>
int launch(const MfmaPlan& p, const float* X, long long n, int ld, const float* Qp, int nq, int k, int metric,
           float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st) {
    const size_t smem = smem_bytes<NW, MT>(k);
    auto fn = metric == RAG_METRIC_IP ? flat_mfma_kernel<NW, MT, RAG_METRIC_IP> : flat_mfma_kernel<NW, MT, RAG_METRIC_L2>;
    if (smem > 48 * 1024)
        RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
    fn<<<(unsigned)(p.S * p.col_tiles), 64 * NW, smem, st>>>(X, n, ld, Qp, nq, k, p.S, p.col_tiles, cand_key, cand_id,
                                                            xnorm, qnorm);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

}  // namespace

namespace rag {

bool mfma_plan(long long n, long long nq, int k, MfmaPlan* p) {
    if (k > 64 || nq < 1 || n < 1) return false;
    int NW = nq > 128 ? 8 : (nq > 64 ? 4 : (nq > 32 ? 2 : 1));
    if (k > 32 && NW > 4) NW = 4;  // LDS: BN * k * 8 bytes of lists beside the tiles
    p->NW = NW;
    p->MT = NW == 1 ? 4 : 8;
    const int BN = 32 * NW, BM = 32 * p->MT;
    p->col_tiles = (int)((nq + BN - 1) / BN);
    const long long ntiles = (n + BM - 1) / BM;
    const int bpc = NW == 8 ? 1 : (NW == 4 ? 2 : (NW == 2 ? 2 : 4));
    long long S = ((long long)num_cus() * bpc + p->col_tiles - 1) / p->col_tiles;
    if (S >= 8) S = (S + 7) & ~7LL;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    p->S = (int)S;
    return true;
}

int mfma_row_norms(const float* X, int ld, long long first, long long n, float* out, hipStream_t st) {
    if (n <= 0) return RAG_OK;
    long long nb = (n + 3) / 4;
    if (nb > (long long)num_cus() * 8) nb = (long long)num_cus() * 8;
    row_norms_kernel<<<(unsigned)nb, 256, 0, st>>>(X, ld, first, n, out);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

int mfma_launch(const MfmaPlan& p, const float* X, long long n, int ld, const float* Qp, int nq, int k, int metric,
                float* cand_key, int* cand_id, const float* xnorm, const float* qnorm, hipStream_t st) {
    switch (p.NW) {
        case 8: return launch<8, 8>(p, X, n, ld, Qp, nq, k, metric, cand_key, cand_id, xnorm, qnorm, st);
        case 4: return launch<4, 8>(p, X, n, ld, Qp, nq, k, metric, cand_key, cand_id, xnorm, qnorm, st);
        case 2: return launch<2, 8>(p, X, n, ld, Qp, nq, k, metric, cand_key, cand_id, xnorm, qnorm, st);
        default: return launch<1, 4>(p, X, n, ld, Qp, nq, k, metric, cand_key, cand_id, xnorm, qnorm, st);
    }
}

}  // namespace rag
