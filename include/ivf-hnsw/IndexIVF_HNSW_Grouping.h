// ivfhnsw::IndexIVF_HNSW_Grouping -- the reference's Grouping(+Pruning) index (IndexIVF_HNSW_Grouping.h:15-75)
// with its search on the device.  Each list is stored sub-group by sub-group (nsubc sub-centroids on the
// segments towards the nsubc nearest centroids); search scores only the sub-groups the pruning rule keeps.
#ifndef IVFHNSW_AMD_INDEX_IVF_HNSW_GROUPING_H
#define IVFHNSW_AMD_INDEX_IVF_HNSW_GROUPING_H

#include "IndexIVF_HNSW.h"

namespace ivfhnsw {

extern int centriodTraceSetup();
extern void centriodTraceClose();

struct IndexIVF_HNSW_Grouping : IndexIVF_HNSW {
    size_t nsubc;    ///< sub-centroids per group
    bool do_pruning; ///< score only sub-groups closer than the mean sub-centroid distance

    std::vector<std::vector<idx_t>> nn_centroid_idxs; ///< per centroid: its nsubc nearest centroids
    std::vector<std::vector<idx_t>> subgroup_sizes;   ///< per centroid: codes per sub-group (empty if no codes)
    std::vector<float> alphas;                        ///< per centroid: position of the sub-centroids

public:
    IndexIVF_HNSW_Grouping(size_t dim, size_t ncentroids, size_t bytes_per_code, size_t nbits_per_idx,
                           size_t nsubcentroids);

    void add_group(size_t group_idx, size_t group_size, const float *x, const idx_t *ids);

    void search(size_t k, const float *x, float *distances, long *labels);
    void search_batch(size_t nq, size_t k, const float *x, float *distances, long *labels) override;
    void searchDisk(size_t k, const float *query, float *distances, long *labels, const char *path_base);

    void write(const char *path_index);
    void read(const char *path_index);
    void write(const char *path_index, bool do_trunc);

    void train_pq(size_t n, const float *x);

    void compute_inter_centroid_dists();
    void dump_inter_centroid_dists(char *path);

    void sync_to_device() override;

protected:
    bool shards_need_graph() const override { return true; } // sub-centroid distances (Grouping.cpp:248,314) on every shard

public:

protected:
    std::vector<float> query_centroid_dists;
    std::vector<std::vector<float>> inter_centroid_dists;

private:
    void compute_residuals(size_t n, const float *x, float *residuals, const float *subcentroids, const idx_t *keys);
    void reconstruct(size_t n, float *x, const float *decoded_residuals, const float *subcentroids, const idx_t *keys);
    void compute_subcentroid_idxs(idx_t *subcentroid_idxs, const float *subcentroids, const float *points,
                                  size_t group_size);
    float compute_alpha(const float *centroid_vectors, const float *points, const float *centroid,
                        const float *centroid_vector_norms_L2sqr, size_t group_size);
};

} // namespace ivfhnsw
#endif
