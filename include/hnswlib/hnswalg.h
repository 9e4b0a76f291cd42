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
    size_t maxelements_;
    size_t cur_element_count;
    size_t efConstruction_;

    VisitedListPool *visitedlistpool;

    std::mutex cur_element_count_guard_;
    idx_t enterpoint_node;

    size_t dist_calc;

    char *data_level0_memory_;

    size_t d_;
    size_t data_size_;
    size_t offset_data;
    size_t size_data_per_element;
    size_t M_;
    size_t maxM_;
    size_t size_links_level0;
    size_t efSearch;

    /// load a saved graph: info (parameters), data (.fvecs centroids), edges
    HierarchicalNSW(const std::string &infoLocation, const std::string &dataLocation, const std::string &edgeLocation);
    /// empty graph for maxelements vectors of dimension d
    HierarchicalNSW(size_t d, size_t maxelements, size_t M, size_t maxM, size_t efConstruction = 500);
    ~HierarchicalNSW();
    HierarchicalNSW(const HierarchicalNSW &) = delete;
    HierarchicalNSW &operator=(const HierarchicalNSW &) = delete;

    inline float *getDataByInternalId(idx_t internal_id) const
    {
        return reinterpret_cast<float *>(data_level0_memory_ + internal_id * size_data_per_element + offset_data);
    }
    inline uint8_t *get_linklist0(idx_t internal_id) const
    {
        return reinterpret_cast<uint8_t *>(data_level0_memory_ + internal_id * size_data_per_element);
    }

    std::priority_queue<std::pair<float, idx_t>> searchBaseLayer(const float *x, size_t ef);
    void getNeighborsByHeuristic(std::priority_queue<std::pair<float, idx_t>> &topResults, size_t NN);
    void mutuallyConnectNewElement(const float *x, idx_t id, std::priority_queue<std::pair<float, idx_t>> topResults);
    void addPoint(const float *point);
    std::priority_queue<std::pair<float, idx_t>> searchKnn(const float *query_data, size_t k);

    void SaveInfo(const std::string &location);
    void SaveEdges(const std::string &location);
    void LoadInfo(const std::string &location);
    void LoadData(const std::string &location);
    void LoadEdges(const std::string &location);

    float fstdistfunc(const float *x, const float *y);
};

} // namespace hnswlib
