// topk.h -- wave64 register-resident sorted top-k list and helpers (gfx950).
//
// One wave keeps the KS*64 best (key, id) pairs seen so far, sorted best-first and spread
// over the lanes: list position p lives in lane p%64 of register slot p/64.  "Best" is
// key descending, id ascending -- the order Python's stable sorted(..., reverse=True)
// produces (vector_storage.py:180) and the tie rule this engine uses for faiss results.
// Insertions are rare once the list is warm (a streamed row has to beat the current k-th
// key), so they are done cooperatively by the whole wave: one ballot to find the position,
// one lane shift per slot.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <climits>

namespace rag {

template <typename T> struct Lowest;
template <> struct Lowest<float>  { static __device__ __forceinline__ float  v() { return -FLT_MAX; } };
template <> struct Lowest<double> { static __device__ __forceinline__ double v() { return -DBL_MAX; } };
template <typename T> struct IdPad;
template <> struct IdPad<int>       { static __device__ __forceinline__ int       v() { return INT_MAX; } };
template <> struct IdPad<long long> { static __device__ __forceinline__ long long v() { return LLONG_MAX; } };

// Cross-lane moves without the LDS crossbar: v_readlane_b32 for a wave-uniform source lane,
// v_mov_b32_dpp wave_shr:1 for "lane i takes lane i-1" (lane 0 keeps its own value).  A
// ds_bpermute round trip costs ~100 cycles; the list insert below is a chain of such moves.
__device__ __forceinline__ int bcast_lane(int v, int uniform_lane) { return __builtin_amdgcn_readlane(v, uniform_lane); }
__device__ __forceinline__ float bcast_lane(float v, int uniform_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), uniform_lane));
}
__device__ __forceinline__ long long bcast_lane(long long v, int uniform_lane) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFLL), uniform_lane);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), uniform_lane);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double bcast_lane(double v, int uniform_lane) {
    return __longlong_as_double(bcast_lane(__double_as_longlong(v), uniform_lane));
}
__device__ __forceinline__ int shift_up1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ float shift_up1(float v) { return __int_as_float(shift_up1(__float_as_int(v))); }
__device__ __forceinline__ long long shift_up1(long long v) {
    const int lo = shift_up1((int)(v & 0xFFFFFFFFLL));
    const int hi = shift_up1((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double shift_up1(double v) {
    return __longlong_as_double(shift_up1(__double_as_longlong(v)));
}

// lane i takes the value of lane i^D without the LDS crossbar: quad_perm DPP (D = 1, 2), a pair
// of bank-masked row shifts (D = 4, 8), v_permlane16_swap / v_permlane32_swap (gfx950) + select.
// Checked on the device by benchmarks/xor_lane_check.hip.
template <int D>
__device__ __forceinline__ int xor_lane(int v) {
    static_assert(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32, "lane distance");
    if constexpr (D == 1) {
        return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);   // quad_perm:[1,0,3,2]
    } else if constexpr (D == 2) {
        return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
    } else if constexpr (D == 4) {
        const int r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);  // row_shl:4 -> banks 0,2
        return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);         // row_shr:4 -> banks 1,3
    } else if constexpr (D == 8) {
        const int r = __builtin_amdgcn_update_dpp(v, v, 0x108, 0xF, 0x3, false);  // row_shl:8 -> banks 0,1
        return __builtin_amdgcn_update_dpp(r, v, 0x118, 0xF, 0xC, false);         // row_shr:8 -> banks 2,3
    } else if constexpr (D == 16) {
        const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? p[0] : p[1];
    } else {
        const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? p[0] : p[1];
    }
}
template <int D> __device__ __forceinline__ unsigned int xor_lane(unsigned int v) { return (unsigned int)xor_lane<D>((int)v); }
template <int D> __device__ __forceinline__ float xor_lane(float v) { return __int_as_float(xor_lane<D>(__float_as_int(v))); }
template <int D> __device__ __forceinline__ long long xor_lane(long long v) {
    const int lo = xor_lane<D>((int)(v & 0xFFFFFFFFLL));
    const int hi = xor_lane<D>((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
template <int D> __device__ __forceinline__ double xor_lane(double v) {
    return __longlong_as_double(xor_lane<D>(__double_as_longlong(v)));
}
// lane i takes the value of lane 63-i: row_mirror DPP (reverse within 16) + the two swaps
__device__ __forceinline__ int reverse_lanes(int v) {
    const int r = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false);
    return xor_lane<32>(xor_lane<16>(r));
}
__device__ __forceinline__ float reverse_lanes(float v) { return __int_as_float(reverse_lanes(__float_as_int(v))); }
__device__ __forceinline__ long long reverse_lanes(long long v) {
    const int lo = reverse_lanes((int)(v & 0xFFFFFFFFLL));
    const int hi = reverse_lanes((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double reverse_lanes(double v) {
    return __longlong_as_double(reverse_lanes(__double_as_longlong(v)));
}

template <typename KeyT, typename IdT, int KS>
struct WaveTopK {
    KeyT key[KS];
    IdT id[KS];

    static __device__ __forceinline__ bool ahead(KeyT ka, IdT ia, KeyT kb, IdT ib) {
        return (ka > kb) || (ka == kb && ia < ib);
    }

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            key[s] = Lowest<KeyT>::v();
            id[s] = IdPad<IdT>::v();
        }
    }

    // Wave-uniform (nk, ni).  All 64 lanes must call.
    __device__ __forceinline__ void insert(KeyT nk, IdT ni, int lane) {
        int pos = 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pos += __popcll(__ballot(ahead(key[s], id[s], nk, ni)));
#pragma unroll
        for (int s = KS - 1; s >= 0; --s) {
            KeyT uk = shift_up1(key[s]);
            IdT ui = shift_up1(id[s]);
            if (s > 0) {
                KeyT ck = bcast_lane(key[s - 1], 63);
                IdT ci = bcast_lane(id[s - 1], 63);
                if (lane == 0) {
                    uk = ck;
                    ui = ci;
                }
            }
            const int p = s * 64 + lane;
            if (p > pos) {
                key[s] = uk;
                id[s] = ui;
            } else if (p == pos) {
                key[s] = nk;
                id[s] = ni;
            }
        }
    }

    // (key, id) at list position k-1, broadcast to every lane.
    __device__ __forceinline__ void kth(int k, KeyT& tk, IdT& ti) const {
        const int p = k - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (s == (p >> 6)) {
                tk = bcast_lane(key[s], p & 63);
                ti = bcast_lane(id[s], p & 63);
            }
    }
};

// Each lane offers one candidate (ck, ci) (valid = false: none).  Candidates that rank
// ahead of the current k-th entry are inserted in lane order.  thr_* track the k-th entry.
template <typename KeyT, typename IdT, int KS>
__device__ __forceinline__ void wave_offer(WaveTopK<KeyT, IdT, KS>& tk, int k, KeyT& thr_k, IdT& thr_i,
                                           KeyT ck, IdT ci, bool valid, int lane) {
    using TK = WaveTopK<KeyT, IdT, KS>;
    unsigned long long m = __ballot(valid && TK::ahead(ck, ci, thr_k, thr_i));
    while (m) {
        const int l = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const KeyT nk = bcast_lane(ck, l);
        const IdT ni = bcast_lane(ci, l);
        if (TK::ahead(nk, ni, thr_k, thr_i)) {
            tk.insert(nk, ni, lane);
            tk.kth(k, thr_k, thr_i);
        }
    }
}

// ---- batch updates: bitonic sort / merge of 64*KS (key, id) pairs held one per (slot, lane) ----
// Element index e = slot*64 + lane.  A sorted insert costs ~0.3 us (a chain of ~80 dependent
// instructions); when more than a handful of candidates of a 64*KS chunk beat the threshold it is
// cheaper to sort the chunk (log^2 steps) and merge it with the list (log steps) at once.
template <typename KeyT, typename IdT, int KS>
struct WaveBitonic {
    using TK = WaveTopK<KeyT, IdT, KS>;
    static constexpr int N = 64 * KS;

    // one compare-exchange step at distance D inside blocks of SZ (SZ >= N: whole array best-first)
    template <int D, int SZ>
    static __device__ __forceinline__ void step(KeyT (&key)[KS], IdT (&id)[KS], int lane) {
        if constexpr (D >= 64) {
            constexpr int DS = D >> 6;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ps = s ^ DS;
                if (ps > s) {  // s is the lower index of the pair
                    const bool dir = SZ >= N ? true : (((s * 64) & SZ) == 0);
                    const bool swap = dir ? TK::ahead(key[ps], id[ps], key[s], id[s])
                                          : TK::ahead(key[s], id[s], key[ps], id[ps]);
                    if (swap) {
                        const KeyT tk_ = key[s]; key[s] = key[ps]; key[ps] = tk_;
                        const IdT ti_ = id[s]; id[s] = id[ps]; id[ps] = ti_;
                    }
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const KeyT ok = xor_lane<D>(key[s]);
                const IdT oi = xor_lane<D>(id[s]);
                const int e = s * 64 + lane;
                const bool lower = (lane & D) == 0;
                const bool dir = SZ >= N ? true : ((e & SZ) == 0);
                const bool keep_better = lower == dir;
                const bool take = keep_better ? TK::ahead(ok, oi, key[s], id[s]) : TK::ahead(key[s], id[s], ok, oi);
                if (take) {
                    key[s] = ok;
                    id[s] = oi;
                }
            }
        }
    }
    template <int D, int SZ>
    static __device__ __forceinline__ void steps_down(KeyT (&key)[KS], IdT (&id)[KS], int lane) {
        step<D, SZ>(key, id, lane);
        if constexpr (D > 1) steps_down<D / 2, SZ>(key, id, lane);
    }
    template <int SZ>
    static __device__ __forceinline__ void sort_from(KeyT (&key)[KS], IdT (&id)[KS], int lane) {
        steps_down<SZ / 2, SZ>(key, id, lane);
        if constexpr (SZ < N) sort_from<SZ * 2>(key, id, lane);
    }
    // full bitonic sort, best first
    static __device__ __forceinline__ void sort(KeyT (&key)[KS], IdT (&id)[KS], int lane) { sort_from<2>(key, id, lane); }

    // list (sorted best-first) <- best N of list U chunk (sorted best-first), sorted best-first
    static __device__ __forceinline__ void merge(TK& tk, const KeyT (&ck)[KS], const IdT (&ci)[KS], int lane) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {  // against the reversed chunk: the result is bitonic
            const KeyT rk = reverse_lanes(ck[KS - 1 - s]);
            const IdT ri = reverse_lanes(ci[KS - 1 - s]);
            if (TK::ahead(rk, ri, tk.key[s], tk.id[s])) {
                tk.key[s] = rk;
                tk.id[s] = ri;
            }
        }
        steps_down<N / 2, N>(tk.key, tk.id, lane);
    }
};

// Offer a chunk of 64*KS candidates (one per slot and lane).  Few beat the threshold: sorted
// inserts.  Many: filter, sort (unless `sorted`: the chunk is already best-first with pads last)
// and merge.  The result is the same list either way (the order is total).
template <typename KeyT, typename IdT, int KS>
__device__ __forceinline__ void wave_offer_chunk(WaveTopK<KeyT, IdT, KS>& tk, int k, KeyT& thr_k, IdT& thr_i,
                                                 KeyT (&ck)[KS], IdT (&ci)[KS], const bool (&valid)[KS], bool sorted,
                                                 int lane) {
    using TK = WaveTopK<KeyT, IdT, KS>;
    bool acc[KS];
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        acc[s] = valid[s] && TK::ahead(ck[s], ci[s], thr_k, thr_i);
        cnt += __popcll(__ballot(acc[s]));
    }
    if (cnt == 0) return;
    // measured on one wave (benchmarks/topk_cost.hip): insert 310/500/960 cycles (KS 1/2/4), sort
    // 6.6k/17.7k/44k, merge 2.0k/4.2k/8.6k: sorting pays from ~28/44/55 accepted candidates, a
    // merge alone from ~7/9/9
    if (cnt <= (sorted ? 8 : 24 + 10 * KS)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) wave_offer<KeyT, IdT, KS>(tk, k, thr_k, thr_i, ck[s], ci[s], acc[s], lane);
        return;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
        if (!acc[s]) {
            ck[s] = Lowest<KeyT>::v();
            ci[s] = IdPad<IdT>::v();
        }
    if (!sorted) WaveBitonic<KeyT, IdT, KS>::sort(ck, ci, lane);
    WaveBitonic<KeyT, IdT, KS>::merge(tk, ck, ci, lane);
    tk.kth(k, thr_k, thr_i);
}

// Sum N = 2^LOGN per-lane values across the wave at once.  Returns, in every lane, the
// wave-wide total of item (lane >> (6-LOGN)); costs N-1 select/shuffle/add steps plus
// 6-LOGN plain butterfly steps instead of 6*N.  Summation order is fixed (deterministic).
template <int N, int HALF, int DIST>
__device__ __forceinline__ void wave_multi_reduce_fold(float (&v)[N], int lane) {
    if constexpr (HALF >= 1) {
        const bool upper = (lane & DIST) != 0;
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const float keep = upper ? v[i + HALF] : v[i];
            const float send = upper ? v[i] : v[i + HALF];
            v[i] = keep + xor_lane<DIST>(send);
        }
        wave_multi_reduce_fold<N, HALF / 2, DIST / 2>(v, lane);
    } else if constexpr (DIST >= 1) {
        v[0] += xor_lane<DIST>(v[0]);
        wave_multi_reduce_fold<N, 0, DIST / 2>(v, lane);
    }
}
template <int N>
__device__ __forceinline__ float wave_multi_reduce(float (&v)[N], int lane) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two <= 64");
    wave_multi_reduce_fold<N, N / 2, 32>(v, lane);
    return v[0];
}

// Block-wide merge of C candidate (key, id) pairs into one sorted k-list held by wave 0 of a
// block of NWAVES waves.  `load(c, key, id)` returns the validity of candidate c in [0, C).
// mk / mi: LDS scratch of (NWAVES-1)*KS*64 entries each.  Every thread of the block must call.
template <typename KeyT, typename IdT, int KS, int NWAVES, class Loader>
__device__ __forceinline__ void block_merge(Loader load, long long C, int k, WaveTopK<KeyT, IdT, KS>& tk,
                                            KeyT* mk, IdT* mi) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    KeyT thr_k = Lowest<KeyT>::v();
    IdT thr_i = IdPad<IdT>::v();
    tk.init();
    // U independent candidate loads are in flight before the first is offered: one round trip to
    // memory per U*64 candidates instead of one per 64 (the loop is latency-bound otherwise).
    constexpr int U = 8;
    static_assert(U % KS == 0, "chunks of KS groups");
    for (long long base = (long long)wid * 64 * U; base < C; base += (long long)NWAVES * 64 * U) {
        KeyT ck[U];
        IdT ci[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long c = base + u * 64 + lane;
            ck[u] = Lowest<KeyT>::v();
            ci[u] = IdPad<IdT>::v();
            valid[u] = c < C ? load(c, ck[u], ci[u]) : false;
        }
#pragma unroll
        for (int u = 0; u < U; u += KS) {
            KeyT k2[KS];
            IdT i2[KS];
            bool v2[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                k2[s] = ck[u + s];
                i2[s] = ci[u + s];
                v2[s] = valid[u + s];
            }
            wave_offer_chunk<KeyT, IdT, KS>(tk, k, thr_k, thr_i, k2, i2, v2, false, lane);
        }
    }
    if (wid > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int o = ((wid - 1) * KS + s) * 64 + lane;
            mk[o] = tk.key[s];
            mi[o] = tk.id[s];
        }
    }
    __syncthreads();
    if (wid == 0) {
        for (int w = 0; w < NWAVES - 1; ++w) {  // the other waves' lists are sorted: merge only
            KeyT k2[KS];
            IdT i2[KS];
            bool v2[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int o = (w * KS + s) * 64 + lane;
                k2[s] = mk[o];
                i2[s] = mi[o];
                v2[s] = true;
            }
            wave_offer_chunk<KeyT, IdT, KS>(tk, k, thr_k, thr_i, k2, i2, v2, true, lane);
        }
    }
}

// Block-wide merge of `nlists` SORTED k-lists (best first, pads last) into one sorted k-list held
// by wave 0.  `load(l, p, key, id)` returns the validity of entry p of list l.  A whole list is one
// chunk (entry p in slot p/64, lane p%64), already sorted: it is merged by the bitonic network, or
// by a few sorted inserts when only a few of its entries beat the wave's threshold -- no sort and
// no per-candidate insert chain as in block_merge.  U lists are in flight per wave.
template <typename KeyT, typename IdT, int KS, int NWAVES, class Loader>
__device__ __forceinline__ void block_merge_lists(Loader load, int nlists, int k, WaveTopK<KeyT, IdT, KS>& tk,
                                                  KeyT* mk, IdT* mi) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    KeyT thr_k = Lowest<KeyT>::v();
    IdT thr_i = IdPad<IdT>::v();
    tk.init();
    constexpr int U = 4;
    for (int l0 = wid; l0 < nlists; l0 += NWAVES * U) {
        KeyT ck[U][KS];
        IdT ci[U][KS];
        bool valid[U][KS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = l0 + u * NWAVES;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s * 64 + lane;
                ck[u][s] = Lowest<KeyT>::v();
                ci[u][s] = IdPad<IdT>::v();
                valid[u][s] = (l < nlists && p < k) ? load(l, p, ck[u][s], ci[u][s]) : false;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            wave_offer_chunk<KeyT, IdT, KS>(tk, k, thr_k, thr_i, ck[u], ci[u], valid[u], true, lane);
    }
    if (wid > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int o = ((wid - 1) * KS + s) * 64 + lane;
            mk[o] = tk.key[s];
            mi[o] = tk.id[s];
        }
    }
    __syncthreads();
    if (wid == 0) {
        for (int w = 0; w < NWAVES - 1; ++w) {
            KeyT k2[KS];
            IdT i2[KS];
            bool v2[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int o = (w * KS + s) * 64 + lane;
                k2[s] = mk[o];
                i2[s] = mi[o];
                v2[s] = true;
            }
            wave_offer_chunk<KeyT, IdT, KS>(tk, k, thr_k, thr_i, k2, i2, v2, true, lane);
        }
    }
}

template <int N> struct Log2 { static constexpr int v = 1 + Log2<N / 2>::v; };
template <> struct Log2<1> { static constexpr int v = 0; };

}  // namespace rag
