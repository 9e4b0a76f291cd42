// faiss::ProductQuantizer as far as the reference uses it (IndexIVF_HNSW.h:56-57, IndexIVF_HNSW.cpp:14-18):
// M sub-quantizers of ksub = 2^nbits code words over dsub = d / M dims, code words stored
// [M][ksub][dsub].  Only nbits = 8 is supported (the reference always passes 8).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace faiss {

struct ProductQuantizer {
    size_t d;      ///< input dimension
    size_t M;      ///< number of sub-quantizers
    size_t nbits;  ///< bits per sub-index
    size_t dsub;   ///< d / M
    size_t byte_per_idx;
    size_t code_size; ///< bytes per code = M * byte_per_idx
    size_t ksub;   ///< code words per sub-quantizer
    bool verbose;
    std::vector<float> centroids; ///< [M][ksub][dsub]

    ProductQuantizer(size_t d, size_t M, size_t nbits);
    ProductQuantizer();

    float *get_centroids(size_t m, size_t i) { return &centroids[(m * ksub + i) * dsub]; }
    const float *get_centroids(size_t m, size_t i) const { return &centroids[(m * ksub + i) * dsub]; }

    /// Lloyd k-means per sub-space (construction side; not on the search path)
    void train(int n, const float *x);
    /// the same with a chosen number of iterations; warm = keep the current code words as the start
    void train_iters(int n, const float *x, int niter, bool warm);
    void compute_code(const float *x, uint8_t *code) const;
    void compute_codes(const float *x, uint8_t *codes, size_t n) const;
    void decode(const uint8_t *code, float *x) const;
    void decode(const uint8_t *code, float *x, size_t n) const;
    /// dis_table[m * ksub + c] = <x_m, centroid(m, c)>
    void compute_inner_prod_table(const float *x, float *dis_table) const;
};

} // namespace faiss
