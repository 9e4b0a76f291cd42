// Host-side mirror of the reference's coarse quantizer class (hnswlib/hnswalg.h:47-102): a single-layer
// navigable small-world graph over the nc centroids.  Same public members, node storage and file formats, so the
// reference's drivers compile against it unchanged; the search hot path itself runs on the device
// (ivf-hnsw_amd/csrc/kernels_hnsw.hip) -- this class is the container the index classes load, rotate, save and
// upload from, and serves construction-side callers (assign, add_group).
//
// Node record (hnswalg.cpp:25-27): [uint8 link count][maxM x uint32 links][d x float32 vector].
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <queue>
#include <string>
#include <utility>

#include <faiss/Heap.h>

#include "visited_list_pool.h"

namespace hnswlib {

typedef uint32_t idx_t;

struct HierarchicalNSW {
    // ---- search-time knob the drivers set directly
    size_t efSearch;

    // ---- node storage: maxelements_ records of size_data_per_element bytes
    char *data_level0_memory_;
    size_t size_data_per_element; ///< size_links_level0 + data_size_
    size_t size_links_level0;     ///< 1 + maxM_ * sizeof(idx_t)
    size_t offset_data;           ///< where the vector starts inside a record (= size_links_level0)
    size_t data_size_;            ///< d_ * sizeof(float)
    size_t d_;

    // ---- graph parameters and state
    size_t M_;    ///< links chosen for a new node
    size_t maxM_; ///< link slots per node
    size_t efConstruction_;
    size_t maxelements_;
    size_t cur_element_count;
    idx_t enterpoint_node;
    std::mutex cur_element_count_guard_;
    VisitedListPool *visitedlistpool;
    size_t dist_calc; ///< distance evaluations so far (diagnostic)

    /// empty graph for maxelements vectors of dimension d
    HierarchicalNSW(size_t d, size_t maxelements, size_t M, size_t maxM, size_t efConstruction = 500);
    /// load a saved graph: info (parameters), data (.fvecs centroids), edges
    HierarchicalNSW(const std::string &infoLocation, const std::string &dataLocation, const std::string &edgeLocation);
    ~HierarchicalNSW();
    HierarchicalNSW(const HierarchicalNSW &) = delete;
    HierarchicalNSW &operator=(const HierarchicalNSW &) = delete;

    inline uint8_t *get_linklist0(idx_t internal_id) const
    {
        return reinterpret_cast<uint8_t *>(data_level0_memory_ + internal_id * size_data_per_element);
    }
    inline float *getDataByInternalId(idx_t internal_id) const
    {
        return reinterpret_cast<float *>(get_linklist0(internal_id) + offset_data);
    }

    /// the k closest vertices found by a beam search of width efSearch, farthest on top
    std::priority_queue<std::pair<float, idx_t>> searchKnn(const float *query_data, size_t k);
    std::priority_queue<std::pair<float, idx_t>> searchBaseLayer(const float *x, size_t ef);
    float fstdistfunc(const float *x, const float *y);

    // construction (serial, so that internal ids equal insertion order)
    void addPoint(const float *point);
    void mutuallyConnectNewElement(const float *x, idx_t id, std::priority_queue<std::pair<float, idx_t>> topResults);
    void getNeighborsByHeuristic(std::priority_queue<std::pair<float, idx_t>> &topResults, size_t NN);

    // files
    void LoadInfo(const std::string &location);
    void LoadData(const std::string &location);
    void LoadEdges(const std::string &location);
    void SaveInfo(const std::string &location);
    void SaveEdges(const std::string &location);
};

} // namespace hnswlib
