// ivfhnsw::IndexIVF_HNSW -- the reference's IVFADC index class (IndexIVF_HNSW.h:46-190) with its search path on
// an MI355X.  Same public data members, constructor and method signatures and on-disk format, so the
// reference's drivers (tests/*.cpp) compile against this header unchanged.  What differs is behind search():
// the lists are mirrored to HBM as CSR arrays the first time they are needed and every search runs through the
// C ABI of <ivfhnsw_hip.h> (HNSW walk, PQ table, ADC scan and top-k all on the device).  There is no CPU search
// path: without a gfx950 device search() throws.
//
// Extensions (not in the reference): search_batch(), sync_to_device(), invalidate_device().
#ifndef IVFHNSW_AMD_INDEX_IVF_HNSW_H
#define IVFHNSW_AMD_INDEX_IVF_HNSW_H

#include <cstdio>
#include <fstream>
#include <iostream>
#include <unordered_map>
#include <vector>

#include <faiss/FaissAssert.h>
#include <faiss/Heap.h>
#include <faiss/ProductQuantizer.h>
#include <faiss/VectorTransform.h>
#include <faiss/index_io.h>
#include <faiss/utils.h>

#include <hnswlib/hnswalg.h>

#include "orcv.h"
#include "utils.h"

#define TRACE_CENTROIDS

struct ivfhnsw_gpu; // opaque device handle of <ivfhnsw_hip.h>

namespace ivfhnsw {

struct IndexIVF_HNSW {
    typedef uint32_t idx_t;

    size_t d;         ///< vector dimension
    size_t nc;        ///< number of coarse centroids (= inverted lists)
    size_t code_size; ///< bytes per PQ code

    hnswlib::HierarchicalNSW *quantizer; ///< coarse quantizer graph over the centroids (owned)

    faiss::ProductQuantizer *pq;        ///< residual code books (owned; drivers replace it)
    faiss::ProductQuantizer *norm_pq;   ///< 1 x 256 quantizer of reconstructed norms (owned)
    faiss::LinearTransform *opq_matrix; ///< OPQ rotation (owned) when do_opq
    bool do_opq;

    size_t nprobe;    ///< lists visited per query
    size_t max_codes; ///< stop after the list that brings the scanned codes to this many

    std::vector<std::vector<idx_t>> ids;          ///< per list: vector ids
    std::vector<std::vector<uint8_t>> codes;      ///< per list: PQ codes, code_size bytes each
    std::vector<std::vector<uint8_t>> norm_codes; ///< per list: norm code

    orcvhdr_t hdr_idx;

    std::vector<float> &get_centroid_norms() { return centroid_norms; }

    // per-query coarse trace kept for source compatibility with the debug drivers; filled by search()
    std::vector<float> trace_query_centroid_dists;
    std::vector<idx_t> trace_centroid_idxs;
    void trace_centroids(size_t idx_q, bool missed);

protected:
    std::vector<float> norms;
    std::vector<float> centroid_norms; ///< ||centroid||^2, saved with the index
    int copy_file(const char *file_src, const char *file_dst);
    size_t M;
    float dmatch = 4444.0;
    float dnear = 8888.0;

public:
    explicit IndexIVF_HNSW(size_t dim, size_t ncentroids, size_t bytes_per_code, size_t nbits_per_idx,
                           size_t max_group_size = 65536);
    virtual ~IndexIVF_HNSW();

    /// Load the graph when both files exist, else build it serially from the centroid file and save it.
    void build_quantizer(const char *path_data, const char *path_info, const char *path_edges, size_t M = 16,
                         size_t efConstruction = 500);

    /// labels[i*k .. ] = the k graph vertices nearest to x_i (device HNSW walk at quantizer->efSearch)
    void assign(size_t n, const float *x, idx_t *labels, size_t k = 1);

    /// One query.  distances / labels hold k entries; unfilled slots are FLT_MAX / -1.
    virtual void search(size_t k, const float *x, float *distances, long *labels);
    virtual void search_debug(size_t k, const float *x, float *distances, long *labels);
    virtual idx_t search_enn(const float *x, float *distances, long *labels);

    /// Extension: nq queries in one device pass (what the hardware is for).
    virtual void search_batch(size_t nq, size_t k, const float *x, float *distances, long *labels);

    virtual void add_batch(size_t n, const float *x, const idx_t *xids, const idx_t *precomputed_idx = nullptr);
    virtual void add_batch2(size_t n, const float *x, const idx_t *xids, const idx_t *idx, uint64_t *eids, char *obuf);
    virtual void train_pq(size_t n, const float *x);

    virtual void write(const char *path_index);
    virtual void write(const char *path_index, bool do_trunc);
    virtual void write2(const char *home_dir, size_t n_vecs, bool do_opq, const char *path_edge);
    virtual void read(const char *path);

    void compute_centroid_norms();
    void rotate_quantizer();

    /// coarse stage supplied by the caller (nearest first), as the reference's search2
    void search2(size_t k, const float *x, float *distances, long *labels, float *query_centroid_dists,
                 idx_t *centroid_idxs);
    void search2m(size_t k, const float *x, float *distances[], long *labels[], float *query_centroid_dists,
                  idx_t *centroid_idxs);

    /// Extension: mirror lists, tables and graph to the device now (search() does it lazily).
    virtual void sync_to_device();
    /// Extension: call after mutating ids/codes/norm_codes/pq/... behind the class's back.
    void invalidate_device() { device_dirty_ = true; }

protected:
    std::vector<float> precomputed_table;
    float pq_L2sqr(const uint8_t *code);

    ivfhnsw_gpu *gpu_;
    bool device_dirty_;
    void ensure_device();
    void device_upload_common();
    /// construction side: graph (once per quantizer state) and code books (every call) for ivfhnsw_gpu_encode
    void ensure_encoder();
    void upload_graph();
    bool graph_dirty_;
    const void *graph_uploaded_for_;
    /// one query per call (what search() is): the latency form of the coarse walk, prepared at the first such call
    /// (ivfhnsw_gpu_prepare_latency; IVFHNSW_LATENCY=0 keeps the throughput walk)
    void ensure_latency_walk();
    const void *latency_for_;
    /// IVFHNSW_SHARDS=N: the lists are split list-wise (c % N) over N device handles -- one per GPU of the node, round
    /// robin when there are fewer -- gpu_ being shard 0; every search is the shard step of SURVEY 8e with the merge done
    /// on the host (keys are nq * k * 8 bytes per shard).  shards_ holds handles 1 .. N-1.
    std::vector<ivfhnsw_gpu *> shards_;
    ivfhnsw_gpu *shard(size_t r) const { return r == 0 ? gpu_ : shards_[r - 1]; }
    size_t nshards() const { return shards_.size() + 1; }
    /// every search of the class ends here: one handle, or the sharded step
    void device_search(size_t nq, size_t k, const float *x, const idx_t *coarse_ids, const float *coarse_dists,
                       size_t nprobe_, size_t max_codes_, bool pruning, float *distances, long *labels);
    void upload_graph_to(ivfhnsw_gpu *handle);
    virtual bool shards_need_graph() const { return false; } ///< Grouping: sub-centroid distances on every shard

private:
    void reconstruct(size_t n, float *x, const float *decoded_residuals, const idx_t *keys);
    void compute_residuals(size_t n, const float *x, float *residuals, const idx_t *keys);
    // cheap fingerprint of what was uploaded, to catch drivers swapping public members
    const void *up_pq_, *up_norm_pq_, *up_opq_, *up_quantizer_;
    size_t up_total_;
    bool up_do_opq_;
};

} // namespace ivfhnsw
#endif
