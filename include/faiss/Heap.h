// The three max-heap entry points of faiss/Heap.h that the reference's search loop calls
// (IndexIVF_HNSW.cpp:265,286-287; IndexIVF_HNSW_Grouping.cpp:268,330-331), on parallel (value, id)
// arrays.  faiss is an un-vendored submodule of the reference; this is an independent implementation of
// the published algorithm (1-based binary max-heap, comparison on values only).
#pragma once
#include <cfloat>
#include <cstddef>

namespace faiss {

/// Fill a k-slot heap with the neutral element (FLT_MAX, -1), or load k0 <= k given pairs.
void maxheap_heapify(size_t k, float *vals, long *ids, const float *x = nullptr, const long *ids_in = nullptr,
                     size_t k0 = 0);
/// Remove the largest value (slot 0); the heap then holds k - 1 entries in slots [0, k-1).
void maxheap_pop(size_t k, float *vals, long *ids);
/// Insert into a heap that currently holds k - 1 entries.
void maxheap_push(size_t k, float *vals, long *ids, float v, long id);

} // namespace faiss
