/*
 * ivfhnsw_hip.h -- C ABI of the MI355X (gfx950) IVFADC search path.
 *
 * This is the drop-in boundary: the reference (uniio/ivf-hnsw, C++11, CPU only) has no FFI of its
 * own, so every entry point below cites the reference interface it replaces (file:line relative to
 * the reference tree).  The host-side mirror of the reference classes (the headers under include/ivf-hnsw/,
 * ivf-hnsw_amd/csrc/host/) calls nothing but these functions; INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions: plain C, opaque handle, int status (0 = ok, negative = error, message through
 * ivfhnsw_gpu_last_error()), no exceptions cross the boundary, host pointers passed to upload_*
 * are copied and never retained.  There is no CPU fallback: every call fails with
 * IVFHNSW_ERR_HIP when no gfx950 device is usable.
 */
#ifndef IVFHNSW_HIP_H
#define IVFHNSW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVFHNSW_OK 0
#define IVFHNSW_ERR_INVALID (-1) /* bad argument / shape mismatch */
#define IVFHNSW_ERR_HIP (-2)     /* HIP runtime error, or no device */
#define IVFHNSW_ERR_STATE (-3)   /* call order: something required was not uploaded */
#define IVFHNSW_ERR_NOMEM (-4)

typedef struct ivfhnsw_gpu ivfhnsw_gpu;

/* Thread-local message of the last failing call on this thread. */
const char *ivfhnsw_gpu_last_error(void);

/* ABI version of this header (bumped on any signature change). */
int ivfhnsw_gpu_abi_version(void);

/* IndexIVF_HNSW::IndexIVF_HNSW / ~IndexIVF_HNSW (IndexIVF_HNSW.cpp:8-32): device-side state of one
 * index.  `device` is the HIP device ordinal. */
int ivfhnsw_gpu_create(int device, ivfhnsw_gpu **out);
/* HIP devices visible to the process (a sharded index puts one handle on each). */
int ivfhnsw_gpu_device_count(int *count);
int ivfhnsw_gpu_destroy(ivfhnsw_gpu *h);

/* A second search context on the SAME device tables: own stream, own per-batch workspace, nothing copied.
 * Batches submitted to the parent and to its views run concurrently (the walk is ALU-bound, the scan HBM-bound:
 * two batches in flight overlap them and fill each other's tails).  The reference's nearest notion is one
 * IndexIVF_HNSW searched from several OpenMP threads, which its member scratch forbids (SURVEY 8b "Threading").
 * The parent must outlive its views and must not upload again while they exist; uploads on a view fail
 * (IVFHNSW_ERR_STATE).  Destroy with ivfhnsw_gpu_destroy. */
int ivfhnsw_gpu_create_view(ivfhnsw_gpu *parent, ivfhnsw_gpu **out);

/* Optional: run on a caller-owned hipStream_t (passed as void*) instead of the handle's own stream. */
int ivfhnsw_gpu_set_stream(ivfhnsw_gpu *h, void *hip_stream);
/* Block until everything queued on the handle's stream has finished.  Also reports (as
 * IVFHNSW_ERR_STATE) a condition a kernel could not represent, e.g. more than 64 exact distance ties
 * at the efSearch boundary of the HNSW walk; the host-pointer entry points check this themselves. */
int ivfhnsw_gpu_sync(ivfhnsw_gpu *h);

/* The inverted lists and the quantizer tables: the data members of IndexIVF_HNSW
 * (IndexIVF_HNSW.h:50-66,81) after read() (IndexIVF_HNSW.cpp:758-779), flattened to CSR.
 * list c occupies [offsets[c], offsets[c+1]) of ids / norm_codes and code_size times that of codes,
 * in list order (the scan order decides ties, IndexIVF_HNSW.cpp:285).
 *
 * Sharding: with shard_world > 1 the arrays ids/codes/norm_codes hold only the lists this
 * shard owns, concatenated in increasing c; offsets is always the global table, so every shard derives
 * the same scan plan.  List c is owned by rank list_owner[c]; with list_owner == NULL by rank
 * c % shard_world.  The owner table lets the caller keep the lists a query probes together on few ranks
 * (a balanced spatial partition of the centroids), so that a query's table is built and staged on those
 * ranks only. */
typedef struct ivfhnsw_ivf_desc {
    size_t d;                    /* IndexIVF_HNSW.h:50 */
    size_t nc;                   /* :51 */
    size_t code_size;            /* :52; multiple of 4 (IndexIVF_HNSW.cpp:805) */
    const uint64_t *offsets;     /* [nc+1] */
    const uint32_t *ids;         /* :64 */
    const uint8_t *codes;        /* :65 */
    const uint8_t *norm_codes;   /* :66 */
    const float *centroid_norms; /* :81, [nc] */
    const float *pq_centroids;   /* pq->centroids, [code_size][256][d/code_size] (:56) */
    const float *norm_table;     /* norm_pq->centroids, [256] (:57) */
    const float *opq_A;          /* opq_matrix->A, [d][d] row major, or NULL when !do_opq (:58-59) */
    uint32_t shard_rank, shard_world; /* 0, 1 for a single GPU */
    const uint32_t *list_owner;  /* [nc] owning rank of every list (< shard_world), or NULL = c % shard_world */
} ivfhnsw_ivf_desc;
int ivfhnsw_gpu_upload_ivf(ivfhnsw_gpu *h, const ivfhnsw_ivf_desc *desc);

/* Same tables, but codes / norm codes are generated on the device (uniform bytes from `seed`) and
 * ids are the running index: the SIFT1B-shaped synthetic corpus of SURVEY 8d for sizes that never
 * exist on the host.  desc->ids/codes/norm_codes are ignored.  With shard_world > 1 the shard receives
 * exactly the bytes and ids its lists have in the unsharded corpus. */
int ivfhnsw_gpu_upload_ivf_synthetic(ivfhnsw_gpu *h, const ivfhnsw_ivf_desc *desc, uint64_t seed);

/* The extra members of IndexIVF_HNSW_Grouping (IndexIVF_HNSW_Grouping.h:17-22,61) after read()
 * (IndexIVF_HNSW_Grouping.cpp:445-483).  All [nc*nsubc] row major; subgroup_sizes rows of empty
 * groups are zero.  Requires upload_ivf first and upload_quantizer before searching. */
int ivfhnsw_gpu_upload_grouping(ivfhnsw_gpu *h, size_t nsubc, const float *alphas,
                                const uint32_t *nn_centroid_idxs, const uint32_t *subgroup_sizes,
                                const float *inter_centroid_dists);

/* The coarse quantizer: hnswlib::HierarchicalNSW node storage (hnswlib/hnswalg.h:47-82; link count,
 * maxM link slots and d floats per node, hnswalg.cpp:25-27) as three arrays.  `vectors` are the
 * centroids as the graph holds them at search time, i.e. after rotate_quantizer()
 * (IndexIVF_HNSW.cpp:789-800) when OPQ is on.  Needed for Grouping (sub-centroid distances,
 * IndexIVF_HNSW_Grouping.cpp:248,314) and for the on-device coarse walk. */
int ivfhnsw_gpu_upload_quantizer(ivfhnsw_gpu *h, size_t n, size_t d, size_t maxM, uint32_t enterpoint,
                                 const uint8_t *link_counts, const uint32_t *links, const float *vectors);

/* One query per call is how the reference's drivers search (tests/test_ivfhnsw_sift1b.cpp:193-208).  prepare_latency
 * builds, once per uploaded quantizer, the "fat" copy of the graph the latency form of the walk reads -- every node
 * with the float rows of its own neighbours, n * 32 * d * 4 bytes (16 GB at 993 127 centroids, d = 128): a whole
 * workgroup then walks one query with one memory round trip per expansion.  After it, calls with at most 256 queries
 * take that form (same results: tests/test_latency_gpu.py).  Needs d = 128 or 96, maxM <= 32, at most 2^20 nodes,
 * efSearch <= 256; otherwise IVFHNSW_ERR_INVALID and nothing changes.  upload_quantizer discards the copy. */
int ivfhnsw_gpu_prepare_latency(ivfhnsw_gpu *h);

/* Large batches as two uneven parts on two streams inside ivfhnsw_gpu_search[_dev] (a batched extension; the reference
 * searches one query per call, IndexIVF_HNSW.cpp:232-293).  permille = share of the batch in the first part, 1..999;
 * 1000, the default since ABI 9, = chosen per call from its own parameters (the walk's time against table + plan + scan:
 * 0.79 at (32, 10000, 80), 1.81 -> 1.67 ms per 10 k queries at the 1B shape; 0.63 at (64, 30000, 100); 0.49 for Grouping);
 * 0 = one part (the environment variable IVFHNSW_SPLIT sets the same at ivfhnsw_gpu_create).  The second part is a whole
 * number of 2048-query rounds, at most half the batch, and runs on an internal view of the handle;
 * its walk fills the tail of the first part's, the first part's table + scan run beside it.  Results, ordering behind
 * the handle's stream and error reporting are those of the unsplit call.  Applies to calls of >= 8192 queries without
 * given coarse results, out_keys or heap-order k > 1. */
int ivfhnsw_gpu_set_batch_split(ivfhnsw_gpu *h, int permille);
/* The queries in the two parts of the last search[_dev] call (second = 0: it ran in one part). */
int ivfhnsw_gpu_last_batch_parts(ivfhnsw_gpu *h, uint64_t *first, uint64_t *second);

/* Options of this library (nothing of the reference's: its knobs are public members, below).  Unknown keys are refused.
 *   "scan_pipe"  -1 (default) the library chooses, 0 never, 1 wherever the shape allows: table + scan of a list shard as
 *                ONE software-pipelined kernel (kernels_scan3.hip) instead of two kernels.  A caller that runs a sharded
 *                step as two overlapping parts (ShardedSearcher) turns it off: the pipelined form holds most of a CU's
 *                LDS and cannot run beside the other part's walk. */
int ivfhnsw_gpu_set_option(ivfhnsw_gpu *h, const char *key, long value);

/* The search-time knobs the drivers set as public members (IndexIVF_HNSW.h:61-62, hnswalg.h:69,
 * IndexIVF_HNSW_Grouping.h:18). */
typedef struct ivfhnsw_search_params {
    size_t nprobe;
    size_t max_codes;
    size_t efSearch;
    int do_pruning;
    int heap_order; /* k > 1 only: 0 = results ascending by (distance, scan position); 1 = exactly the array
                       faiss's max-heap leaves behind (IndexIVF_HNSW.cpp:265,285-288), slot 0 = current worst */
} ivfhnsw_search_params;

/* IndexIVF_HNSW::search / IndexIVF_HNSW_Grouping::search (IndexIVF_HNSW.cpp:234-296,
 * IndexIVF_HNSW_Grouping.cpp:188-363) for nq queries at once.  All pointers are HOST memory.
 *
 * coarse_ids / coarse_dists ([nq*nprobe], nearest first) are the result of the coarse stage when the
 * caller ran it, exactly IndexIVF_HNSW::search2 (IndexIVF_HNSW.cpp:453-492); coarse slots holding
 * 0xffffffff are skipped.  Pass NULL for both to run the HNSW walk (hnswalg.cpp:48-109,227-234) on
 * the device; that needs upload_quantizer and efSearch >= nprobe.
 *
 * Results: distances[nq*k], labels[nq*k] (int64, the reference's `long`); unfilled slots hold FLT_MAX / -1 as
 * after faiss::maxheap_heapify.  For k = 1 (every preset of the reference) this is exactly the reference's
 * output.  For k > 1 the same set is returned either ascending by (distance, scan position)
 * (params->heap_order = 0) or, with heap_order = 1, in exactly the heap-array order the reference leaves: the
 * device replays faiss's pop/push over a superset of the admitted codes in scan order, which yields the same
 * heap because a code that fails `dist < distances[0]` leaves the heap untouched. */
int ivfhnsw_gpu_search(ivfhnsw_gpu *h, size_t nq, size_t k, const float *queries, const uint32_t *coarse_ids,
                       const float *coarse_dists, const ivfhnsw_search_params *params, float *distances,
                       int64_t *labels);

/* Same, with every buffer already resident in HBM (device pointers), asynchronous on the handle's
 * stream.  out_keys (nullable, [nq*k] int64) receives the packed (orderable distance << 32 | scan
 * position) keys, sign-flipped so that a plain signed int64 MIN over the shards (RCCL all-reduce) picks
 * the reference's winner: smallest distance, earliest scan position on ties. */
int ivfhnsw_gpu_search_dev(ivfhnsw_gpu *h, size_t nq, size_t k, const float *d_queries,
                           const uint32_t *d_coarse_ids, const float *d_coarse_dists,
                           const ivfhnsw_search_params *params, float *d_distances, int64_t *d_labels,
                           int64_t *d_out_keys);

/* Multi-GPU merge helper: given keys already MIN-reduced over the shards, resolve the
 * labels this shard owns ([nq*k]; -1 where the winner lives on another shard) from the scan plan of
 * the last search_dev call, and decode the distances.  The caller then MAX-reduces the labels. */
int ivfhnsw_gpu_resolve_keys_dev(ivfhnsw_gpu *h, size_t nq, size_t k, const int64_t *d_keys,
                                 float *d_distances, int64_t *d_labels);

/* k > 1 across shards.  Ascending order: all-gather the out_keys of every shard and keep the k smallest
 * per query, then resolve.  The reference's heap-array order (IndexIVF_HNSW.cpp:265,285-288) needs the sequence of
 * admitted codes in scan order: search_dev with heap_order = 1 AND out_keys leaves each shard's candidate stream -- a
 * superset of the codes faiss's heap admits, in this shard's scan order, as (orderable distance << 32 | scan position)
 * keys -- copied out by last_stream_dev: d_len ([nq], nullable) receives the stream lengths, d_keys ([nq][len_cap],
 * nullable) the first len_cap keys of every query's stream, *stream_cap (nullable) the library's capacity; a length
 * above that capacity means the stream overflowed: use ascending order.  The caller merges the shards' streams by
 * scan position and hands the merged stream to replay_stream_dev, which replays faiss's pop/push over it and writes
 * the heap ARRAY as signed keys ([nq*k], unfilled slots = the FLT_MAX key); resolve_keys_dev + MAX-reduce then give
 * labels.  Replaying a superset in scan order is exact: a code failing `dist < distances[0]` leaves the heap as it is. */
int ivfhnsw_gpu_last_stream_dev(ivfhnsw_gpu *h, size_t nq, size_t len_cap, uint64_t *d_keys, uint32_t *d_len,
                                uint32_t *stream_cap);
int ivfhnsw_gpu_replay_stream_dev(ivfhnsw_gpu *h, size_t nq, size_t k, const uint64_t *d_stream, const uint32_t *d_len,
                                  uint32_t cap, int64_t *d_out_keys);

/* Host-pointer forms of the shard step, for a caller that merges the shards itself in one process (the bundled classes
 * with IVFHNSW_SHARDS=N: N handles, one per GPU of the node).  search_keys = search_dev with out_keys on host buffers
 * (the coarse stage must be supplied: it is computed once for all shards); resolve_keys = resolve_keys_dev;
 * last_stream = last_stream_dev.  All synchronous. */
int ivfhnsw_gpu_search_keys(ivfhnsw_gpu *h, size_t nq, size_t k, const float *queries, const uint32_t *coarse_ids,
                            const float *coarse_dists, const ivfhnsw_search_params *params, int64_t *keys);
int ivfhnsw_gpu_resolve_keys(ivfhnsw_gpu *h, size_t nq, size_t k, const int64_t *keys, float *distances,
                             int64_t *labels);
int ivfhnsw_gpu_last_stream(ivfhnsw_gpu *h, size_t nq, size_t len_cap, uint64_t *keys, uint32_t *lens,
                            uint32_t *stream_cap);

/* The whole shard step for a caller that holds the N shard handles of one index in ONE process (north_star: "per-shard
 * top-k merged over RCCL/xGMI" under C++ host code): every shard scans its lists for the batch (the coarse stage is
 * supplied: it is computed once, IndexIVF_HNSW::search2's split, IndexIVF_HNSW.cpp:453-492), the packed keys are
 * MIN-merged across the shards' devices with one RCCL all-reduce (ncclInt64 / ncclMin; RCCL is loaded on first use), each
 * shard resolves the labels it owns, one more all-reduce (ncclMax) merges them, and shard 0's device returns distances
 * and labels.  k = 1, or k > 1 ascending (heap_order = 0).  Shards that share a device (a one-GPU box) or k > 1: the same
 * step with the merge on the host.  Host pointers; at most 131 072 queries per call. */
int ivfhnsw_gpu_search_sharded(ivfhnsw_gpu *const *shards, size_t nshards, size_t nq, size_t k, const float *queries,
                               const uint32_t *coarse_ids, const float *coarse_dists, const ivfhnsw_search_params *params,
                               float *distances, int64_t *labels);

/* The coarse stage alone (HierarchicalNSW::searchKnn, hnswalg.cpp:227-234, plus the unload loop of
 * IndexIVF_HNSW.cpp:249-259): device pointers, [nq*nprobe] outputs, nearest first.  Queries must
 * already be rotated when OPQ is on. */
int ivfhnsw_gpu_coarse_dev(ivfhnsw_gpu *h, size_t nq, const float *d_queries, size_t nprobe, size_t efSearch,
                           uint32_t *d_coarse_ids, float *d_coarse_dists);

/* opq_matrix->apply (IndexIVF_HNSW.cpp:240) alone, device pointers: d_out[q] = A * d_queries[q].  A plain copy
 * when the index holds no OPQ matrix.  For callers that run the coarse stage themselves (sharded search). */
int ivfhnsw_gpu_rotate_dev(ivfhnsw_gpu *h, size_t nq, const float *d_queries, float *d_out);

/* Host-pointer form of the coarse stage; with k = 1 this is IndexIVF_HNSW::assign
 * (IndexIVF_HNSW.cpp:68-72) for n vectors.  The OPQ rotation is NOT applied (assign() takes vectors in
 * the graph's space).  Slots beyond the number of nodes found hold 0xffffffff / 0. */
int ivfhnsw_gpu_coarse(ivfhnsw_gpu *h, size_t nq, const float *queries, size_t k, size_t efSearch,
                       uint32_t *ids, float *dists);

/* ---- measurement ------------------------------------------------------------------------------ */

/* ---- construction side: what IndexIVF_HNSW::add_batch computes before it appends --------
 *
 * ivfhnsw_gpu_upload_codebooks: the residual code book (faiss::ProductQuantizer, [M][256][d/M] floats), the norm
 * code book (ProductQuantizer(1,1,8): 256 floats) and the OPQ matrix (row major [d][d], NULL = no OPQ).
 * Independent of upload_ivf: an index under construction has no lists yet.  upload_quantizer must have
 * been called as well (the centroid rows are the graph's vectors).
 *
 * ivfhnsw_gpu_encode replaces IndexIVF_HNSW.cpp:75-121 for n base vectors (host pointers):
 *   idx      = precomputed_idx, or assign(n, x) = searchKnn(x, 1) with efSearch (:68-72) when NULL
 *   residual = x - centroid[idx]                     (fvec_madd, :258-262)
 *   codes    = pq->compute_codes([A] residual)       (:92-93)
 *   norm     = || centroid[idx] + [A^T] pq->decode(codes) ||^2,  norm_codes = norm_pq->compute_codes(norm)
 * out_idx may be NULL; out_codes [n*code_size]; out_norm_codes [n].  The caller appends them to its lists
 * (:122-131).  Bytes out: compared bit for bit with the CPU restatement in tests/test_gpu_encode.py. */
int ivfhnsw_gpu_upload_codebooks(ivfhnsw_gpu *h, size_t d, size_t code_size, const float *pq_centroids,
                                 const float *norm_table, const float *opq_A);
int ivfhnsw_gpu_encode(ivfhnsw_gpu *h, size_t n, const float *x, const uint32_t *precomputed_idx, size_t efSearch,
                       uint32_t *out_idx, uint8_t *out_codes, uint8_t *out_norm_codes);

/* ivfhnsw_gpu_encode_groups replaces IndexIVF_HNSW_Grouping::add_group (IndexIVF_HNSW_Grouping.cpp:43-125, up to
 * the distribution loops) for ngroups groups at once; group g is centroid centroid_idx[g] with the points
 * x[offsets[g] .. offsets[g+1]) (host pointers, offsets[0] = 0).  Per group:
 *   nn_centroid_idxs = searchKnn(centroid, nsubc + 1) with efSearch, minus the nearest          (:47-62)
 *   alpha            = compute_alpha over the group's points                                    (:691-733)
 *   sub-centroid s   = centroid + alpha * (neighbour_s - centroid)                              (:70-87)
 *   subcentroid_idx  = first nearest sub-centroid of every point                                (:673-689)
 *   codes, norm_codes as in ivfhnsw_gpu_encode with the sub-centroid in place of the centroid   (:93-125)
 * out_nn_centroid_idxs [ngroups*nsubc]; out_alphas [ngroups] (written only for groups with points: an empty
 * group keeps the caller's value, :63-64); out_subcentroid_idxs [n]; out_codes [n*code_size]; out_norm_codes [n].
 * The caller lays each list out sub-group by sub-group in arrival order and records the sub-group sizes
 * (:127-155).  Needs efSearch >= nsubc + 1; a walk that finds fewer centroids is an error. */
int ivfhnsw_gpu_encode_groups(ivfhnsw_gpu *h, size_t ngroups, size_t nsubc, const uint32_t *centroid_idx,
                              const uint64_t *offsets, const float *x, size_t efSearch, uint32_t *out_nn_centroid_idxs,
                              float *out_alphas, uint32_t *out_subcentroid_idxs, uint8_t *out_codes,
                              uint8_t *out_norm_codes);

/* ---- code-book training -------------------------------------------------------------------
 *
 * ivfhnsw_gpu_pq_train: the Lloyd iterations behind faiss::ProductQuantizer::train, which IndexIVF_HNSW::train_pq
 * (IndexIVF_HNSW.cpp:536-593) and IndexIVF_HNSW_Grouping::train_pq (IndexIVF_HNSW_Grouping.cpp:486-560) call on
 * residuals: niter iterations on n points x [n][d] (host), centroids [M][256][d/M] in and out (host).  Assignment =
 * pq->compute_codes with the current code book (first nearest code word, faiss's SSE order), update = mean of the
 * assigned sub-vectors with the sum taken in point order in float, code words nothing was assigned to stay.
 * out_assign (nullable) [n][M]: the last iteration's assignments.  Needs no upload.  faiss's own clustering (its
 * sampling, its random stream, its empty-cluster splits) is not reproduced: spec-level, parity unpinned.
 *
 * ivfhnsw_gpu_xty: C[a][b] = sum_i X[i][a] * Y[i][b] ([n][d] each, host; C [d][d]): the product behind the orthogonal
 * Procrustes step of faiss::OPQMatrix::train, on the matrix cores (v_mfma_f32_32x32x2_f32).  Order: fmaf chains over
 * chunks of IVFHNSW_XTY_CHUNK points, the chunks' partial products added in chunk order. */
#define IVFHNSW_XTY_CHUNK 2048
int ivfhnsw_gpu_pq_train(ivfhnsw_gpu *h, size_t n, size_t d, size_t M, const float *x, size_t niter, float *centroids,
                         uint8_t *out_assign);
int ivfhnsw_gpu_xty(ivfhnsw_gpu *h, size_t n, size_t d, const float *X, const float *Y, float *C);

/* ---- exact nearest-neighbour tables and graph construction -------------------------------------
 *
 * ivfhnsw_gpu_knn: for each of nq query rows the k nearest of nx base rows by brute force on the matrix cores -- the exact
 * form of what the reference's construction side approximates with graph searches (a new node's link candidates,
 * hnswlib/hnswalg.cpp:212-225 -> :48-109; the nsubc + 1 nearest centroids of a centroid,
 * IndexIVF_HNSW_Grouping.cpp:47-62) and of the ground-truth files its drivers score Recall@1 against
 * (tests/test_ivfhnsw_sift1b.cpp:173-215).  Host pointers; queries == NULL means the base rows themselves (nq = nx).
 * mode: IVFHNSW_KNN_ALL every base row is a candidate; IVFHNSW_KNN_NOT_SELF row i of the base is no candidate of query
 * i (a neighbour table); IVFHNSW_KNN_EARLIER only base rows j < i are candidates of query i (the table an incremental
 * construction sees: node i against the nodes inserted before it).  d a multiple of 4, at most 128; k <= 80.
 * Arithmetic: dist = (norm(q) + norm(x)) - 2 * dot(q, x) with norm and dot as fmaf chains over k = 0..d-1 (what
 * v_mfma_f32_32x32x2_f32 computes); results ascending by (dist, id), slots beyond the candidates that exist hold
 * 0xffffffff / FLT_MAX.  out_dists may be NULL.
 * ivfhnsw_gpu_knn_dev: the same on device pointers, asynchronous on the handle's stream (d_queries may equal d_base). */
#define IVFHNSW_KNN_ALL 0
#define IVFHNSW_KNN_NOT_SELF 1
#define IVFHNSW_KNN_EARLIER 2
int ivfhnsw_gpu_knn(ivfhnsw_gpu *h, size_t nq, size_t nx, size_t d, const float *queries, const float *base, size_t k,
                    int mode, uint32_t *out_ids, float *out_dists);
int ivfhnsw_gpu_knn_dev(ivfhnsw_gpu *h, size_t nq, size_t nx, size_t d, const float *d_queries, const float *d_base,
                        size_t k, int mode, uint32_t *d_out_ids, float *d_out_dists);

/* ivfhnsw_gpu_build_graph: hnswlib::HierarchicalNSW::addPoint for ALL n nodes at once (hnswlib/hnswalg.cpp:212-225 as
 * IndexIVF_HNSW::build_quantizer loops it, IndexIVF_HNSW.cpp:34-66), with ONE deviation: a node's link candidates are
 * its EXACT ncand nearest among the nodes inserted before it (ivfhnsw_gpu_knn, IVFHNSW_KNN_EARLIER, on the matrix cores)
 * instead of the efConstruction results of a greedy search of the graph built so far (hnswalg.cpp:221 -> :48-109).
 * Everything after the candidates is the reference's: getNeighborsByHeuristic down to M links (:110-146, distances by
 * fstdistfunc, :326-357), the links stored farthest first (:153-170), and mutuallyConnectNewElement's reverse links in
 * insertion order -- appended while the neighbour has room, else the neighbour's maxM + 1 candidates shrunk by the same
 * heuristic (:171-209).  Because the candidates no longer depend on the graph, node t's final list is a fold over the
 * later nodes that chose t, in their order: every node is processed independently (host threads), and the result is
 * exactly what the serial loop would leave.  vectors [n][d] host; out_counts [n] (the 1-byte link count of
 * hnswalg.cpp:25); out_links [n][maxM].  M <= maxM <= 64, ncand <= 80, n < 2^32. */
int ivfhnsw_gpu_build_graph(ivfhnsw_gpu *h, size_t n, size_t d, const float *vectors, size_t M, size_t maxM, size_t ncand,
                            uint8_t *out_counts, uint32_t *out_links);

enum ivfhnsw_stage {
    IVFHNSW_STAGE_OPQ = 0,    /* opq_matrix->apply, IndexIVF_HNSW.cpp:240 */
    IVFHNSW_STAGE_COARSE = 1, /* quantizer->searchKnn, :248 */
    IVFHNSW_STAGE_LUT = 2,    /* pq->compute_inner_prod_table, :262 */
    IVFHNSW_STAGE_PLAN = 3,   /* probe order + max_codes rule, :267-292 / Grouping.cpp:222-262 */
    IVFHNSW_STAGE_SCAN = 4,   /* the ADC loop, :282-289 */
    IVFHNSW_STAGE_SELECT = 5, /* label resolution */
    IVFHNSW_STAGE_COUNT = 6
};
/* With profiling on, every search brackets each stage with hipEvents on the launch stream: enabled = 1 every stage,
 * 2 only the scan, the kernel the roofline is about (an event pair costs about 7 us of stream time; six of them are
 * 2 % of a 10 k-query step), 0 off. */
int ivfhnsw_gpu_set_profiling(ivfhnsw_gpu *h, int enabled);
/* Accumulated since the last reset: milliseconds and number of launches of one stage. */
int ivfhnsw_gpu_get_stage_ms(ivfhnsw_gpu *h, int stage, double *ms_total, uint64_t *launches);
int ivfhnsw_gpu_reset_stage_ms(ivfhnsw_gpu *h);
/* Accounting of the last search on this handle: codes scored by this shard (the reference's `ncode`,
 * IndexIVF_HNSW.cpp:290, summed over the batch) and (sub)lists scored.  Synchronises. */
int ivfhnsw_gpu_last_scan_counts(ivfhnsw_gpu *h, uint64_t *ncodes, uint64_t *nsegments);
/* Name of the scan kernel the last search on this handle launched (the dominant kernel of the path: bench.py
 * reports its roofline under this name).  Never NULL; empty before the first search. */
const char *ivfhnsw_gpu_last_scan_kernel(ivfhnsw_gpu *h);
/* Bytes of HBM currently held by the handle. */
int ivfhnsw_gpu_memory_bytes(ivfhnsw_gpu *h, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif
