// The handful of faiss/utils.h helpers the reference links against.
#pragma once
#include <cstddef>

namespace faiss {

float fvec_L2sqr(const float *x, const float *y, size_t d);
float fvec_inner_product(const float *x, const float *y, size_t d);
float fvec_norm_L2sqr(const float *x, size_t d);
/// nr[i] = ||x_i||^2 for nx vectors of dimension d
void fvec_norms_L2sqr(float *nr, const float *x, size_t d, size_t nx);
/// c = a + bf * b
void fvec_madd(size_t n, const float *a, float bf, const float *b, float *c);
/// a random permutation of 0..n-1, reproducible from the seed
void rand_perm(int *perm, size_t n, long seed);

} // namespace faiss
