// The PQ / OPQ file readers and writers the reference's drivers call
// (tests/test_ivfhnsw_sift1b.cpp:54-90).  Layouts follow faiss 1.x index_io.cpp:
//   ProductQuantizer : size_t d, size_t M, size_t nbits, size_t count, count floats (count == d * ksub)
//   LinearTransform  : uint32 fourcc "LTra", bool have_bias, vec<f32> A, vec<f32> b (size_t count + data),
//                      int d_in, int d_out, bool is_trained
// Readers are strict and throw std::runtime_error on anything else.
#pragma once
#include "ProductQuantizer.h"
#include "VectorTransform.h"

namespace faiss {

void write_ProductQuantizer(const ProductQuantizer *pq, const char *fname);
ProductQuantizer *read_ProductQuantizer(const char *fname);
void write_VectorTransform(const VectorTransform *vt, const char *fname);
VectorTransform *read_VectorTransform(const char *fname);

} // namespace faiss
