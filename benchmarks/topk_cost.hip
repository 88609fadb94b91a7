// topk_cost.hip -- cycles per primitive of csrc/topk.h on one wave (s_memtime around loops).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o benchmarks/topk_cost benchmarks/topk_cost.hip
#include "../rag_system_amd/csrc/topk.h"
#include <cstdio>
using namespace rag;

template <int KS>
__global__ void cost(const float* in, long long* out, int k, int reps) {
    const int lane = threadIdx.x & 63;
    WaveTopK<float, int, KS> tk;
    tk.init();
    float thr_k = Lowest<float>::v();
    int thr_i = IdPad<int>::v();
    // (a) inserts of increasing keys (always accepted, always at position 0)
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        tk.insert(in[r & 1023] + (float)r, r, lane);
        tk.kth(k, thr_k, thr_i);
    }
    long long t1 = __builtin_readcyclecounter();
    // (b) bitonic sort of a chunk
    float ck[KS]; int ci[KS];
    float acc = 0.f;
    for (int r = 0; r < reps / 8; ++r) {
#pragma unroll
        for (int s = 0; s < KS; ++s) { ck[s] = in[(r * 64 * KS + s * 64 + lane) & 1023]; ci[s] = r * 64 * KS + s * 64 + lane; }
        WaveBitonic<float, int, KS>::sort(ck, ci, lane);
        acc += ck[0];
    }
    long long t2 = __builtin_readcyclecounter();
    // (c) merge of a sorted chunk into the list
    for (int r = 0; r < reps / 8; ++r) {
        WaveBitonic<float, int, KS>::merge(tk, ck, ci, lane);
        ck[0] += 1.f;
    }
    long long t3 = __builtin_readcyclecounter();
    // (d) wave_offer of 64 candidates of which exactly one is accepted
    for (int r = 0; r < reps; ++r) {
        tk.kth(k, thr_k, thr_i);
        const float c = lane == (r & 63) ? thr_k + 1.f : Lowest<float>::v();
        wave_offer<float, int, KS>(tk, k, thr_k, thr_i, c, 100000 + r, true, lane);
    }
    long long t4 = __builtin_readcyclecounter();
    // (e) rejected group: ballot only
    for (int r = 0; r < reps; ++r) {
        const float c = in[(r + lane) & 1023] - 1e30f;
        wave_offer<float, int, KS>(tk, k, thr_k, thr_i, c, r, true, lane);
    }
    long long t5 = __builtin_readcyclecounter();
    if (lane == 0) {
        out[0] = (t1 - t0) / reps; out[1] = (t2 - t1) / (reps / 8); out[2] = (t3 - t2) / (reps / 8);
        out[3] = (t4 - t3) / reps; out[4] = (t5 - t4) / reps;
        out[5] = (long long)(acc + tk.key[0] + thr_k);
    }
}

int main() {
    float h[1024], *d; long long *o, ho[6];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
    hipMalloc(&d, 4096); hipMalloc(&o, 64); hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        cost<1><<<1, 64>>>(d, o, 64, 4096); hipMemcpy(ho, o, 48, hipMemcpyDeviceToHost);
        if (pass) printf("KS=1: insert+kth %lld | sort64 %lld | merge %lld | offer(1 hit) %lld | offer(0 hits) %lld cycles\n", ho[0], ho[1], ho[2], ho[3], ho[4]);
        cost<2><<<1, 64>>>(d, o, 128, 4096); hipMemcpy(ho, o, 48, hipMemcpyDeviceToHost);
        if (pass) printf("KS=2: insert+kth %lld | sort128 %lld | merge %lld | offer(1 hit) %lld | offer(0 hits) %lld cycles\n", ho[0], ho[1], ho[2], ho[3], ho[4]);
        cost<4><<<1, 64>>>(d, o, 256, 4096); hipMemcpy(ho, o, 48, hipMemcpyDeviceToHost);
        if (pass) printf("KS=4: insert+kth %lld | sort256 %lld | merge %lld | offer(1 hit) %lld | offer(0 hits) %lld cycles\n", ho[0], ho[1], ho[2], ho[3], ho[4]);
    }
    return 0;
}
