// build.hip -- term-major postings of a tokenised corpus, built on the device.
//
// Replaces the per-document Python loops in which rank_bm25's BM25Okapi.__init__ counts term
// frequencies (called at vector_storage.py:170 and :305, corpus tokenised at :162): tokens are
// mapped to integer words on the host, everything after that -- vocabulary in first-appearance order,
// (term, document) pairs, term frequencies, document frequencies -- is a sort + segment on the GPU:
//   1. first position of every word            atomicMin per token
//   2. term id = rank of that position          host, over the vocabulary only (small)
//   3. key = term << 32 | document              document by binary search in doc_indptr
//   4. radix sort of the keys                   rocPRIM (a plain library sort, like a library GEMM)
//   5. run-length encode                        rocPRIM: one run per (term, document) = one posting,
//                                               run length = term frequency
//   6. indptr from the term boundaries of the sorted postings
// The result is exactly what the host loops produce: postings of a term ascend by document, term
// ids follow first appearance (the order of rank_bm25's dictionaries, which fixes the float64
// summation order of average_idf).  Integer work throughout: bit-exact by construction, checked
// against the numpy restatement in tests/test_build_gpu.py.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace {

using namespace rag;

constexpr long long NO_POS = 0x7fffffffffffffffLL;

__global__ void fill_i64_kernel(long long* p, long long n, long long v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// first_pos[w] = min position of word w
__global__ void first_pos_kernel(const int* __restrict__ words, long long n_tokens, long long n_words,
                                 long long* __restrict__ first_pos, int* __restrict__ bad) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_tokens; i += stride) {
        const int w = words[i];
        if (w < 0 || w >= n_words) {
            *bad = 1;
            continue;
        }
        if (first_pos[w] > i) atomicMin(&first_pos[w], i);  // the plain read only skips atomics that cannot win
    }
}

// key = term(word) << 32 | document of the token (doc_indptr[d] <= i < doc_indptr[d+1])
__global__ void make_keys_kernel(const int* __restrict__ words, long long n_tokens, const int* __restrict__ word_to_term,
                                 const long long* __restrict__ doc_indptr, long long n_docs,
                                 unsigned long long* __restrict__ keys) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_tokens; i += stride) {
        long long lo = 0, hi = n_docs;  // largest d with doc_indptr[d] <= i (empty documents are skipped over)
        while (hi - lo > 1) {
            const long long mid = (lo + hi) >> 1;
            if (doc_indptr[mid] <= i) lo = mid;
            else hi = mid;
        }
        keys[i] = ((unsigned long long)(unsigned int)word_to_term[words[i]] << 32) | (unsigned long long)lo;
    }
}

// postings from the runs: rows[j], tf[j]; indptr[t] = first posting of term t (terms without postings
// inherit the next term's start), indptr[n_terms] = nnz
__global__ void postings_kernel(const unsigned long long* __restrict__ ukeys, const unsigned int* __restrict__ counts,
                                long long nnz, long long n_terms, int* __restrict__ rows, double* __restrict__ tf,
                                long long* __restrict__ indptr) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) {
        const unsigned long long key = ukeys[j];
        rows[j] = (int)(key & 0xffffffffULL);
        tf[j] = (double)counts[j];
        const long long t = (long long)(key >> 32);
        const long long tprev = j == 0 ? -1 : (long long)(ukeys[j - 1] >> 32);
        for (long long u = tprev + 1; u <= t; ++u) indptr[u] = j;
        if (j == nnz - 1)
            for (long long u = t + 1; u <= n_terms; ++u) indptr[u] = nnz;
    }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        RAG_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return RAG_OK;
    }
    template <typename T>
    T* as() { return static_cast<T*>(p); }
};

}  // namespace

extern "C" {

int rag_postings_build(int64_t n_docs, const int64_t* doc_indptr, int64_t n_tokens, const int32_t* words, int64_t n_words,
                       int64_t* n_terms_out, int32_t* term_word, int64_t* indptr, int32_t* rows, double* tf, int64_t* nnz_out) {
    if (!doc_indptr || !n_terms_out || !term_word || !indptr || !nnz_out || n_docs < 1 || n_tokens < 0 || n_words < 0 ||
        (n_tokens > 0 && (!words || !rows || !tf)))
        return fail(RAG_ERR_INVALID_ARG, "rag_postings_build: null pointer or negative size");
    if (n_docs > 0x7fffffffLL || n_words > 0x7fffffffLL)
        return fail(RAG_ERR_UNSUPPORTED, "rag_postings_build: more than 2^31-1 documents or words");
    if (doc_indptr[0] != 0 || doc_indptr[n_docs] != n_tokens)
        return fail(RAG_ERR_INVALID_ARG, "rag_postings_build: doc_indptr must run from 0 to n_tokens");
    RAG_REQUIRE_INIT();
    if (n_tokens > 0xffffffffLL) return fail(RAG_ERR_UNSUPPORTED, "rag_postings_build: more than 2^32-1 tokens");
    int rc = RAG_OK;
    hipStream_t st = lib_stream();
    *n_terms_out = 0;
    *nnz_out = 0;
    indptr[0] = 0;
    if (n_tokens == 0) return RAG_OK;
    if (n_words == 0) return fail(RAG_ERR_INVALID_ARG, "rag_postings_build: a word id outside [0, n_words) (n_words = 0)");
    // host targets / sources of asynchronous copies, declared before `drain` so that every return path --
    // the error returns included -- waits for the stream before they go out of scope
    std::vector<long long> first((size_t)n_words);
    std::vector<int> w2t((size_t)n_words, -1);
    int bad = 0;
    unsigned long long nruns = 0;
    struct Drain {
        hipStream_t st;
        ~Drain() { (void)hipStreamSynchronize(st); }
    } drain{st};

    DevBuf d_words, d_indptr, d_first, d_bad, d_w2t, d_keys, d_keys2, d_ukeys, d_counts, d_nruns, d_tmp, d_rows, d_tf, d_pind;
    if ((rc = d_words.alloc((size_t)n_tokens * 4)) || (rc = d_indptr.alloc((size_t)(n_docs + 1) * 8)) ||
        (rc = d_first.alloc((size_t)n_words * 8)) || (rc = d_bad.alloc(4)) || (rc = d_w2t.alloc((size_t)n_words * 4)))
        return rc;
    RAG_HIP(hipMemcpyAsync(d_words.p, words, (size_t)n_tokens * 4, hipMemcpyHostToDevice, st));
    RAG_HIP(hipMemcpyAsync(d_indptr.p, doc_indptr, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, st));
    RAG_HIP(hipMemsetAsync(d_bad.p, 0, 4, st));
    const unsigned nb = (unsigned)std::min<long long>((n_tokens + 255) / 256, (long long)num_cus() * 16);
    fill_i64_kernel<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>(d_first.as<long long>(), n_words, NO_POS);
    first_pos_kernel<<<nb, 256, 0, st>>>(d_words.as<int>(), n_tokens, n_words, d_first.as<long long>(), d_bad.as<int>());
    RAG_HIP(hipGetLastError());
    RAG_HIP(hipMemcpyAsync(first.data(), d_first.p, (size_t)n_words * 8, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipStreamSynchronize(st));
    if (bad) return fail(RAG_ERR_INVALID_ARG, "rag_postings_build: a word id outside [0, n_words)");

    // term ids in first-appearance order (vocabulary-sized host step)
    std::vector<int> seen;
    seen.reserve((size_t)n_words);
    for (long long w = 0; w < n_words; ++w)
        if (first[(size_t)w] != NO_POS) seen.push_back((int)w);
    std::sort(seen.begin(), seen.end(), [&](int a, int b) { return first[(size_t)a] < first[(size_t)b]; });
    const long long n_terms = (long long)seen.size();
    for (long long t = 0; t < n_terms; ++t) {
        term_word[t] = seen[(size_t)t];
        w2t[(size_t)seen[(size_t)t]] = (int)t;
    }
    RAG_HIP(hipMemcpyAsync(d_w2t.p, w2t.data(), (size_t)n_words * 4, hipMemcpyHostToDevice, st));

    if ((rc = d_keys.alloc((size_t)n_tokens * 8)) || (rc = d_keys2.alloc((size_t)n_tokens * 8))) return rc;
    make_keys_kernel<<<nb, 256, 0, st>>>(d_words.as<int>(), n_tokens, d_w2t.as<int>(), d_indptr.as<long long>(), n_docs,
                                         d_keys.as<unsigned long long>());
    RAG_HIP(hipGetLastError());
    int term_bits = 1;
    while ((1LL << term_bits) < n_terms) ++term_bits;
    const unsigned end_bit = 32u + (unsigned)term_bits;
    size_t tmp_bytes = 0;
    RAG_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                     (size_t)n_tokens, 0u, end_bit, st));
    if ((rc = d_tmp.alloc(tmp_bytes))) return rc;
    RAG_HIP(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(),
                                     (size_t)n_tokens, 0u, end_bit, st));
    // runs of equal keys = postings; d_keys is free again and receives the unique keys
    if ((rc = d_counts.alloc((size_t)n_tokens * 4)) || (rc = d_nruns.alloc(8))) return rc;
    size_t tmp2 = 0;
    RAG_HIP(rocprim::run_length_encode(nullptr, tmp2, d_keys2.as<unsigned long long>(), (unsigned int)n_tokens,
                                       d_keys.as<unsigned long long>(), d_counts.as<unsigned int>(),
                                       d_nruns.as<unsigned long long>(), st));
    DevBuf d_tmp2;
    if ((rc = d_tmp2.alloc(tmp2))) return rc;
    RAG_HIP(rocprim::run_length_encode(d_tmp2.p, tmp2, d_keys2.as<unsigned long long>(), (unsigned int)n_tokens,
                                       d_keys.as<unsigned long long>(), d_counts.as<unsigned int>(),
                                       d_nruns.as<unsigned long long>(), st));
    RAG_HIP(hipMemcpyAsync(&nruns, d_nruns.p, 8, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipStreamSynchronize(st));
    const long long nnz = (long long)nruns;
    if ((rc = d_rows.alloc((size_t)nnz * 4)) || (rc = d_tf.alloc((size_t)nnz * 8)) || (rc = d_pind.alloc((size_t)(n_terms + 1) * 8)))
        return rc;
    postings_kernel<<<(unsigned)std::min<long long>((nnz + 255) / 256, (long long)num_cus() * 16), 256, 0, st>>>(
        d_keys.as<unsigned long long>(), d_counts.as<unsigned int>(), nnz, n_terms, d_rows.as<int>(), d_tf.as<double>(),
        d_pind.as<long long>());
    RAG_HIP(hipGetLastError());
    RAG_HIP(hipMemcpyAsync(rows, d_rows.p, (size_t)nnz * 4, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipMemcpyAsync(tf, d_tf.p, (size_t)nnz * 8, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipMemcpyAsync(indptr, d_pind.p, (size_t)(n_terms + 1) * 8, hipMemcpyDeviceToHost, st));
    RAG_HIP(hipStreamSynchronize(st));
    *n_terms_out = n_terms;
    *nnz_out = nnz;
    return RAG_OK;
}

}  // extern "C"
