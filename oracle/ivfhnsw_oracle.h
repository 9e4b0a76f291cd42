/*
 * ivfhnsw_oracle.h -- CPU restatement of the uniio/ivf-hnsw IVFADC search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The shipped path is ivf-hnsw_amd/csrc (HIP) behind include/ivfhnsw_hip.h.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, fixtures or known-answer tests for
 * this path and cannot be built here (every reference translation unit
 * includes <faiss/...> headers of an empty, un-vendored submodule).  This file
 * restates the reference's algorithm from its sources; each function cites the file:line it
 * follows.  Leaf arithmetic that lives in faiss (third-party, version not recoverable, flat-header
 * era <= 1.5) is restated from faiss's published algorithms and marked "faiss spec".
 *
 * All citations are relative to /root/reference.
 */
#ifndef IVFHNSW_ORACLE_H
#define IVFHNSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- leaf arithmetic ---------------------------------------------------------------------- */

/* hnswlib/hnswalg.cpp:326-357 == utils.cpp:22-52 (AVX branch): 8 lane accumulators over blocks
 * of 16 floats, unfused multiply then add, lanes summed left to right.  Dims beyond the last
 * multiple of 16 are ignored, as in the reference. */
float orc_l2sqr(const float *x, const float *y, size_t d);

/* faiss spec: ProductQuantizer::compute_inner_prod_table (call site IndexIVF_HNSW.cpp:262).
 * tab[m*256 + c] = <x[m*dsub..], centroid[m][c]>, evaluated in the order of faiss's SSE
 * fvec_inner_product: 4 lane sums over blocks of 4 (zero padded tail), then (s0+s1)+(s2+s3). */
void orc_inner_prod_table(const float *x, const float *pq_centroids, size_t d, size_t M, float *tab);

/* faiss spec: LinearTransform::apply for an OPQ matrix without bias (call site
 * IndexIVF_HNSW.cpp:240).  y[i] = sum_k A[i][k]*x[k] as an fmaf chain in k order (BLAS order is
 * unspecified in the reference; this order is what the gfx950 f32 MFMA reproduces bit for bit). */
void orc_opq_apply(const float *A, const float *x, size_t d, float *y);

/* faiss spec: Heap.h maxheap_heapify / maxheap_pop / maxheap_push (CMax<float,long>), call sites
 * IndexIVF_HNSW.cpp:265,286-287. */
void orc_maxheap_heapify(size_t k, float *val, long *ids);
void orc_maxheap_pop(size_t k, float *val, long *ids);
void orc_maxheap_push(size_t k, float *val, long *ids, float v, long id);

/* ---- HNSW coarse quantizer (hnswlib/hnswalg.{h,cpp}) ---------------------------------------- */

typedef struct orc_hnsw orc_hnsw;

orc_hnsw *orc_hnsw_new(size_t d, size_t maxelements, size_t M, size_t maxM, size_t efConstruction);
void orc_hnsw_free(orc_hnsw *g);
/* hnswalg.cpp:212-225 addPoint (serial, reference-identical graph). */
int orc_hnsw_add_point(orc_hnsw *g, const float *point);
/* Adopt a graph built elsewhere: counts[n], links[n*maxM], vectors[n*d] are copied. */
orc_hnsw *orc_hnsw_from_arrays(size_t d, size_t n, size_t M, size_t maxM, uint32_t enterpoint,
                               const uint8_t *counts, const uint32_t *links, const float *vectors);
/* hnswalg.cpp:227-234 searchKnn: returns the number of results r <= k; results are unloaded
 * nearest first (the order IndexIVF_HNSW.cpp:249-259 leaves them in). */
size_t orc_hnsw_search_knn(orc_hnsw *g, const float *query, size_t ef, size_t k, uint32_t *out_ids,
                           float *out_dists);
size_t orc_hnsw_n(const orc_hnsw *g);
size_t orc_hnsw_d(const orc_hnsw *g);
size_t orc_hnsw_maxM(const orc_hnsw *g);
uint32_t orc_hnsw_enterpoint(const orc_hnsw *g);
const uint8_t *orc_hnsw_counts(const orc_hnsw *g);   /* [n] */
const uint32_t *orc_hnsw_links(const orc_hnsw *g);   /* [n*maxM] */
float *orc_hnsw_vectors(orc_hnsw *g);                /* [n*d], mutable for rotate_quantizer */
unsigned long long orc_hnsw_dist_calc(const orc_hnsw *g); /* hnswalg.cpp:57,91 dist_calc */
/* hnswalg.cpp:236-324 Save/Load Info, Edges, Data (fvecs). */
int orc_hnsw_save(const orc_hnsw *g, const char *path_info, const char *path_edges);
orc_hnsw *orc_hnsw_load(const char *path_info, const char *path_data, const char *path_edges);

/* ---- index ------------------------------------------------------------------------------------ */

typedef struct orc_index {
    size_t d, nc, code_size; /* IndexIVF_HNSW.h:50-52 */
    orc_hnsw *quantizer;     /* :54 (borrowed) */
    float *pq_centroids;     /* faiss pq->centroids [M][256][dsub] */
    float norm_table[256];   /* norm_pq->centroids */
    float *opq_A;            /* opq_matrix->A [d][d] row major, or NULL */
    int do_opq;              /* :59 */
    size_t nprobe, max_codes, efSearch; /* :61-62, hnswalg.h:69 */
    uint64_t *offsets;       /* CSR form of ids/codes/norm_codes (:64-66): list c = [offsets[c], offsets[c+1]) */
    uint32_t *ids;
    uint8_t *codes;
    uint8_t *norm_codes;
    float *centroid_norms;   /* :81 */
    /* grouping (IndexIVF_HNSW_Grouping.h:17-22,61); nsubc == 0 for plain IVFADC */
    size_t nsubc;
    int do_pruning;
    float *alphas;           /* [nc] */
    uint32_t *nn_centroid_idxs; /* [nc*nsubc] */
    uint32_t *subgroup_sizes;   /* [nc*nsubc]; all zero for an empty group */
    float *inter_centroid_dists; /* [nc*nsubc] */
} orc_index;

/* Per-query accounting the bench and the tests use (the reference's `ncode`, IndexIVF_HNSW.cpp:290). */
typedef struct orc_stats {
    unsigned long long ncode;      /* codes scored */
    unsigned long long nseg;       /* (sub)lists scored */
    unsigned long long dist_evals; /* coarse + sub-centroid distance evaluations */
} orc_stats;

/* IndexIVF_HNSW.cpp:234-296.  distances/labels are k-sized, heap-array order, FLT_MAX/-1 padded. */
void orc_search_ivf(const orc_index *ix, size_t k, const float *x, float *distances, long *labels,
                    orc_stats *st);
/* IndexIVF_HNSW.cpp:453-492 search2: coarse ids/dists supplied by the caller (nearest first). */
void orc_search_ivf_coarse(const orc_index *ix, size_t k, const float *x, const uint32_t *centroid_idxs,
                           const float *query_centroid_dists, float *distances, long *labels, orc_stats *st);
/* IndexIVF_HNSW_Grouping.cpp:188-363 (TRACE_NEIGHBOUR logging compiled out). */
void orc_search_grouping(const orc_index *ix, size_t k, const float *x, float *distances, long *labels,
                         orc_stats *st);
/* Same with the coarse stage supplied by the caller. */
void orc_search_grouping_coarse(const orc_index *ix, size_t k, const float *x, const uint32_t *centroid_idxs,
                                const float *coarse_dists, float *distances, long *labels, orc_stats *st);

/* Serial loop of the above over nq queries (tests/test_ivfhnsw_sift1b.cpp:193-208), or the same
 * loop under OpenMP with per-thread scratch when nthreads > 1 (an extension: the reference has no
 * parallel search path).  Also returns the coarse stage when out_coarse_* are non-NULL
 * ([nq*nprobe], nearest first, padded with 0xffffffff / 0 when fewer than nprobe are found). */
void orc_search_batch(const orc_index *ix, size_t nq, size_t k, const float *x, float *distances,
                      long *labels, uint32_t *out_coarse_ids, float *out_coarse_dists, orc_stats *st_sum,
                      int nthreads);

/* Construction side, IndexIVF_HNSW.cpp:75-121 (add_batch up to the append loop): for n base vectors
 * idx = precomputed_idx or assign() (:68-72, searchKnn(x, 1) with ix->efSearch), residual (:258-262 fvec_madd
 * with -1), [OPQ apply], pq->compute_codes, pq->decode, [OPQ transform_transpose], reconstruct (fvec_madd
 * with +1), fvec_norms_L2sqr, norm_pq->compute_codes.  Uses ix->quantizer, pq_centroids, norm_table, opq_A /
 * do_opq, d, code_size.  out_idx may be NULL; out_norms (the float norms before coding) may be NULL. */
void orc_add_batch_encode(const orc_index *ix, size_t n, const float *x, const uint32_t *precomputed_idx,
                          uint32_t *out_idx, uint8_t *out_codes, uint8_t *out_norm_codes, float *out_norms);

/* Code book training (IndexIVF_HNSW.cpp:536-593 train_pq -> faiss ProductQuantizer::train; faiss spec, parity
 * unpinned): niter Lloyd iterations on n points x [n][d], centroids [M][256][d/M] in/out; assignment = first nearest
 * code word in fvec_L2sqr's SSE order, update = mean with the sum taken in point order in float, empty clusters keep
 * their code word.  out_assign (nullable) [n][M]: the last iteration's assignments. */
void orc_pq_lloyd(size_t n, size_t d, size_t M, const float *x, size_t niter, float *centroids,
                  uint8_t *out_assign);
/* C[a][b] = sum_i X[i][a] * Y[i][b] in the order of the device's MFMA kernel: fmaf chains over chunks of `chunk`
 * points, partial products added in chunk order (OPQ's Procrustes product). */
void orc_xty(size_t n, size_t d, const float *X, const float *Y, size_t chunk, float *C);

/* Grouping construction, IndexIVF_HNSW_Grouping.cpp:43-157 (add_group up to the distribution loops) for ONE
 * group: neighbour centroids = searchKnn(centroid, nsubc+1) with ix->efSearch minus the nearest (:47-62),
 * alpha (compute_alpha, :691-733), sub-centroids (:83-88), the sub-centroid of every point
 * (compute_subcentroid_idxs, :673-689), then residual / [OPQ] / codes / decode / [OPQ back] / reconstruct /
 * norm / norm code against the sub-centroids (:93-125).  Outputs: nn_centroid_idxs [nsubc], *alpha,
 * subcentroid_idxs [group_size], codes [group_size*code_size], norm_codes [group_size].  The caller lays the
 * list out sub-group by sub-group in arrival order (:127-155).  Returns 0, or -1 when the walk finds fewer than
 * nsubc+1 centroids (the reference then leaves zero entries behind; defined here as an error).
 * A neighbour at distance 0 (duplicate centroids) makes alpha candidates NaN; the reference's heap order with
 * NaN keys is whatever std::priority_queue happens to do -- here a NaN candidate loses to every other one. */
int orc_add_group_encode(const orc_index *ix, size_t nsubc, uint32_t centroid_idx, size_t group_size,
                         const float *data, uint32_t *nn_centroid_idxs, float *alpha, uint32_t *subcentroid_idxs,
                         uint8_t *codes, uint8_t *norm_codes);

/* IndexIVF_HNSW.cpp:781-787 / Grouping.cpp:620-631. */
void orc_compute_centroid_norms(const orc_hnsw *g, float *centroid_norms);
void orc_compute_inter_centroid_dists(const orc_hnsw *g, size_t nsubc, const uint32_t *nn_idx, float *out);
/* IndexIVF_HNSW.cpp:789-800 rotate_quantizer (in place). */
void orc_rotate_quantizer(orc_hnsw *g, const float *A);

/* ---- on-disk formats ---------------------------------------------------------------------------- */

/* IndexIVF_HNSW.cpp:637-663,758-779 and Grouping.cpp:397-483.  grouping != 0 selects the
 * Grouping layout.  The writer takes CSR arrays; the reader allocates them (free with
 * orc_index_free_lists). */
int orc_index_write(const orc_index *ix, const char *path, int grouping);
int orc_index_read(orc_index *ix, const char *path, int grouping);
void orc_index_free_lists(orc_index *ix);

/* exact k-nearest-neighbour tables: the contract of ivfhnsw_gpu_knn (see the .c file) */
void orc_knn(size_t nq, size_t nx, size_t d, const float *queries, const float *base, size_t k, int mode,
             uint32_t *ids, float *dists);
/* the serial insertion loop ivfhnsw_gpu_build_graph unrolls (exact candidates, the reference's connect step) */
orc_hnsw *orc_hnsw_build_exact(size_t d, size_t n, size_t M, size_t maxM, size_t ncand, const float *vectors);

#ifdef __cplusplus
}
#endif
#endif
