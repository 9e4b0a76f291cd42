// faiss::VectorTransform / LinearTransform / OPQMatrix as far as the reference uses them
// (IndexIVF_HNSW.h:58, IndexIVF_HNSW.cpp:93,108,240,546-559,798).
#pragma once
#include <cstddef>
#include <vector>

namespace faiss {

struct VectorTransform {
    int d_in, d_out;
    bool is_trained;
    explicit VectorTransform(int d_in = 0, int d_out = 0) : d_in(d_in), d_out(d_out), is_trained(true) {}
    virtual ~VectorTransform() {}
    virtual void train(long n, const float *x);
    /// returns a new[]-allocated array of n * d_out floats (the reference frees it with delete)
    float *apply(long n, const float *x) const;
    virtual void apply_noalloc(long n, const float *x, float *xt) const = 0;
};

struct LinearTransform : VectorTransform {
    bool have_bias;
    std::vector<float> A; ///< [d_out][d_in] row major
    std::vector<float> b; ///< [d_out] when have_bias
    bool verbose;
    explicit LinearTransform(int d_in = 0, int d_out = 0, bool have_bias = false);
    /// xt = A x (+ b)
    void apply_noalloc(long n, const float *x, float *xt) const override;
    /// x = A^T (y - b): the inverse when A is orthonormal
    void transform_transpose(long n, const float *y, float *x) const;
};

struct OPQMatrix : LinearTransform {
    int M;
    int niter;
    int niter_pq;
    int niter_pq_0;
    size_t max_train_points;
    OPQMatrix(int d = 0, int M = 1, int d2 = -1);
    /// non-parametric OPQ (alternate PQ training and an orthogonal Procrustes step); construction side, host only
    void train(long n, const float *x) override;
};

} // namespace faiss
