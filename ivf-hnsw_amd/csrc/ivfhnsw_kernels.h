// Device-side data layout and kernel launchers of the gfx950 IVFADC search path.
// Host code (capi.cpp) sees only the launch_* functions; kernels live in kernels_*.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>

namespace ivfhnsw_gpu_impl {

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that drives several GPUs
// (the bundled classes with IVFHNSW_SHARDS=N, one host thread per shard) must raise it on each of them, and the
// bookkeeping of "already raised to ..." must be per device and safe against the shard threads.
struct DynLdsState {
    std::mutex m;
    size_t raised[64] = {};
};
inline hipError_t raise_dyn_lds(const void *kern, size_t bytes, DynLdsState &st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    if (dev < 0 || dev >= 64)
        return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    std::lock_guard<std::mutex> lk(st.m);
    if (bytes > st.raised[dev]) {
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess)
            return e;
        st.raised[dev] = bytes;
    }
    return hipSuccess;
}


// One scored (sub)list of one query: the unit of the scan plan.
// start  : index of the first code inside this shard's flat code array
// len    : number of codes
// vpos   : position of the first code in the query's global scan order (tie-break, and label lookup)
// cterm  : the per-list constant of the distance: term1 (IndexIVF_HNSW.cpp:277) or term1+term2
//          (IndexIVF_HNSW_Grouping.cpp:290,318)
struct __attribute__((aligned(16))) Seg {
    uint32_t start;
    uint32_t len;
    uint32_t vpos;
    float cterm;
};

// Per-query plan header: nseg segments owned by this shard, `total` codes in them.
struct __attribute__((aligned(8))) PlanHdr {
    uint32_t nseg;
    uint32_t total;
};

// Packed result key: (orderable(dist) << 32) | vpos.  Unsigned order of the key is the order of
// (dist, scan position): strict '<' of IndexIVF_HNSW.cpp:285 == "smallest key wins".
constexpr uint32_t kOrdFltMax = 0x7f7fffffu | 0x80000000u; // orderable(FLT_MAX)
constexpr uint64_t kKeyInit = (uint64_t)kOrdFltMax << 32;  // accept iff key < kKeyInit
constexpr uint64_t kSignFlip = 0x8000000000000000ull;      // keys leave the library as signed-orderable int64

constexpr uint32_t kNotOwned = 0xffffffffu;
// dynamic LDS the run-time-code-size scan kernels may ask for (1 KB per code byte; the rest of the 160 KB holds
// the plan chunk, the norm table and the top-k buffers)
constexpr size_t kScanDynLdsMax = 128 * 1024;

struct IvfTables {
    int d, M, dsub;             // M == code_size
    uint32_t nc;
    const uint64_t *goff;       // [nc+1] global list offsets
    const uint32_t *loff;       // [nc] first local code of list c; kNotOwned when the list lives on another shard
    const float *centroid_norms;
    const float *pq_centroids;  // [M][256][dsub]
    const float *norm_table;    // [256]
    const float *opq_At;        // [d][d] TRANSPOSED: At[k*d+i] = A[i][k], or null
    const uint8_t *codes;
    const uint8_t *norm_codes;
    const uint32_t *ids;
    uint32_t shard_rank, shard_world;
};

struct GroupTables {
    int nsubc;
    const float *alphas;
    const uint32_t *nn_idx;
    const uint32_t *sub_sizes;
    const float *inter_dists;
    int dedupe; // the plan evaluates every distinct neighbour centroid of a query once (kernels_grouping.hip DEDUPE): set at
                // upload when the neighbour lists of nearby groups overlap enough to pay for the hash set
};

struct GraphTables {
    uint32_t n;
    int d, maxM;
    uint32_t enterpoint;
    const uint8_t *counts;
    const uint32_t *links;   // [n][maxM]
    const float *vectors;    // [n][d]
    // Exact rejection filter of the walk (DESIGN.md "coarse walk"): every row once more as d bytes,
    // x ~ q_lo + q_step * byte, with q_errc >= max_row ||x - (q_lo + q_step*byte)|| / q_step.  NULL = filter off.
    const uint8_t *qrows;    // [n][d]
    float q_lo, q_step, q_errc;
    // The same bytes once more, laid out for the walk (d <= 128): node i carries the byte rows of its own
    // neighbours, [n][nb_rows][128] with nb_rows = maxM rounded up to 32, zero padded.  One contiguous
    // nb_rows*128-byte read per expansion, issued together with the link list.  NULL = gather from qrows.
    const uint8_t *nbrows;
    const uint32_t *nbnorms; // [n][nb_rows] sum of bytes^2 of every neighbour row (the filter's ||c||^2 term)
    const uint32_t *links_c; // [n][maxM] id | (link count of id) << 24; goes with nbrows: the walk's keys carry the count
    int nb_rows;
    // Latency form of the walk (kernels_hnsw_lat.hip): node i carries the FLOAT rows of its own neighbours,
    // per node 32 rows of d floats (zero beyond the link count) and a 256-byte trailer with its links and link count;
    // NULL until ivfhnsw_gpu_prepare_latency builds it.
    const float *fat;
    int visited_clean;    // walk (set by its launcher): the global visited bitmaps are zero on entry and left zero
    int skip_padding;     // walk: the filter skips the arithmetic of rows beyond the link count (always on since round 3)
    int merge_admissions; // walk: insert a pass's admitted rows in one step (A/B knob IVFHNSW_WALK_MERGE=0)
    int links_unique;     // no id twice in a link list: survivors of the filter may enter the visited set late
};

// y[q][i] = fmaf chain over k of A[i][k] * x[q][k]  (IndexIVF_HNSW.cpp:240)
hipError_t launch_opq(hipStream_t s, const float *At, const float *x, float *y, int nq, int d);
// tab[q][m][c] = <x_m, centroid[m][c]>  (IndexIVF_HNSW.cpp:262)
// hdr (optional): queries whose plan is empty on this shard get no table
hipError_t launch_lut(hipStream_t s, const IvfTables &t, const float *xq, float *luts, int nq, const PlanHdr *hdr = nullptr);
// probe order + max_codes rule (IndexIVF_HNSW.cpp:267-292); also resets keys[nq*k] to kKeyInit
hipError_t launch_plan_ivf(hipStream_t s, const IvfTables &t, const uint32_t *coarse_ids,
                           const float *coarse_dists, int nq, int nprobe, uint64_t max_codes, Seg *segs,
                           uint32_t *lpos, PlanHdr *hdr, int max_seg, uint64_t *keys, int k);
// plan_ivf + lut in one launch (one GPU; hipErrorInvalidValue for a dsub without an instantiation)
hipError_t launch_plan_lut(hipStream_t s, const IvfTables &t, const float *xq, const uint32_t *coarse_ids,
                           const float *coarse_dists, int nq, int nprobe, uint64_t max_codes, Seg *segs, uint32_t *lpos,
                           PlanHdr *hdr, int max_seg, uint64_t *keys, int k, float *luts);
// Grouping: sub-centroid distances, pruning threshold, pass-2 plan (IndexIVF_HNSW_Grouping.cpp:222-353)
hipError_t launch_plan_grouping(hipStream_t s, const IvfTables &t, const GroupTables &g, const GraphTables &gr,
                                const float *xq, const uint32_t *coarse_ids, const float *coarse_dists, int nq,
                                int nprobe, uint64_t max_codes, int do_pruning, Seg *segs, uint32_t *lpos,
                                PlanHdr *hdr, int max_seg, uint64_t *keys, int k, float *qsd_scratch);
// the ADC loop (IndexIVF_HNSW.cpp:282-289 / IndexIVF_HNSW_Grouping.cpp:321-333)
hipError_t launch_scan(hipStream_t s, const IvfTables &t, const float *luts, const Seg *segs, const uint32_t *lpos,
                       const PlanHdr *hdr, int max_seg, int nq, int k, int nsplit, uint64_t *keys,
                       uint64_t *stream = nullptr, uint32_t *stream_len = nullptr, uint32_t stream_cap = 0,
                       int seg_len_hint = 0,
                       // k = 1: let the scan resolve the winner's label itself where one workgroup sees the whole query;
                       // *did_select tells whether it did (else launch_select has to run)
                       float *sel_dist = nullptr, int64_t *sel_labels = nullptr, bool *did_select = nullptr); // expected codes per plan segment (0 = unknown): picks the scan form
// table + scan pipelined over queries, for list shards (kernels_scan3.hip)
bool scan_pipe_supported(const IvfTables &t, int max_seg, int nq, int nsplit, bool has_codes, bool forced = false);
hipError_t launch_scan_pipe(hipStream_t s, const IvfTables &t, const float *xq, const Seg *segs, const uint32_t *lpos,
                            const PlanHdr *hdr, int max_seg, int nq, uint64_t *keys);
// plan + table + scan + select of a small IVFADC batch in one launch (kernels_tail.hip); keys_inv [nq] and done [nq] zeroed
bool ivf_tail_supported(const IvfTables &t, int nprobe, int k);
hipError_t launch_ivf_tail(hipStream_t s, const IvfTables &t, const float *xq, const uint32_t *cid, const float *cd,
                           int nq, int nprobe, uint64_t max_codes, int nsplit, uint64_t *keys_inv, uint32_t *done,
                           PlanHdr *hdr, float *dist, int64_t *labels, const uint32_t *status_word, uint32_t *status_out);
const char *last_scan_kernel_name(); // the kernel the calling thread's last launch_scan chose
// k > 1 in faiss heap-array order: sequential replay of the top-k kernel's candidate stream
hipError_t launch_heap_replay(hipStream_t s, const IvfTables &t, const Seg *segs, const PlanHdr *hdr, int max_seg,
                              const uint64_t *stream, const uint32_t *stream_len, uint32_t stream_cap, int nq, int k,
                              float *dist, int64_t *labels, uint32_t *status,
                              int64_t *out_keys = nullptr); // non-null: the heap array as signed keys, no labels (sharded)
// keys -> (distance, label) through the plan; also emits signed-orderable keys when out_keys != null
hipError_t launch_select(hipStream_t s, const IvfTables &t, const Seg *segs, const PlanHdr *hdr, int max_seg,
                         const uint64_t *keys, int nq, int k, float *dist, int64_t *labels, int64_t *out_keys);
// same, reading signed-orderable keys (after a cross-shard MIN)
hipError_t launch_resolve(hipStream_t s, const IvfTables &t, const Seg *segs, const PlanHdr *hdr, int max_seg,
                          const int64_t *skeys, int nq, int k, float *dist, int64_t *labels);
// HNSW walk, one wavefront per query (hnswalg.cpp:48-109,227-234)
hipError_t launch_coarse(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                         uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                         size_t visited_words_per_slot, int nslots, uint32_t *status, uint32_t *next_query,
                         size_t visited_bytes = 0, bool *visited_zero = nullptr, // in/out: the whole scratch is zero
                         // queries the fast form cannot finish (> 64 exact ties at the efSearch boundary) are listed in
                         // redo_list [nq] and walked again by launch_coarse_redo, which launch_coarse ends with;
                         // next_query must be followed by three more words (the list's header: length, the redo
                         // launch's counter, its exit count); tail_bitmaps [tail_slots][words] zero on entry and on
                         // exit (walk_set.h TailSpill).  counters_clean (in/out): the four words are zero -- the redo
                         // launch's last wavefront leaves them so, and the launcher then skips its memset
                         uint32_t *redo_list = nullptr, uint32_t *tail_bitmaps = nullptr, int tail_slots = 0,
                         bool *counters_clean = nullptr);
hipError_t launch_coarse_redo(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                              uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                              size_t visited_words_per_slot, uint32_t *status, uint32_t *redo_hdr, uint32_t *redo_list,
                              uint32_t *tail_bitmaps, int tail_slots);
int coarse_slots_for(int ef);
// one workgroup per query on the fat graph (small batches: the reference's one-query-per-call drivers)
bool coarse_latency_supported(const GraphTables &g, int ef);
size_t coarse_latency_fat_bytes(const GraphTables &g);
hipError_t launch_build_fat(hipStream_t s, const GraphTables &g, float *fat);
hipError_t launch_coarse_latency(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                                 uint32_t *coarse_ids, float *coarse_dists, uint32_t *status,
                                 uint64_t *zero_keys = nullptr, uint32_t *zero_done = nullptr, // [nq] words to clear
                                 uint32_t *redo_hdr = nullptr, uint32_t *redo_list = nullptr); // zeroed header: see launch_coarse
// bits of the device status word
constexpr uint32_t kStatusHnswTieOverflow = 1u; // latency walk on the synchronous host-pointer path only: the call repeats itself (capi.cpp)
constexpr uint32_t kStatusTopkStreamOverflow = 2u;
// synthetic corpus: uniform bytes from a counter hash; ids = running index
// construction side (kernels_encode.hip): IndexIVF_HNSW.cpp:75-121
hipError_t launch_madd_rows(hipStream_t s, const float *a, float bf, const float *table, const uint32_t *idx, float *c,
                            size_t n, int d);
hipError_t launch_pq_encode(hipStream_t s, const float *r, const float *cb, uint8_t *codes, size_t n, int d, int M);
hipError_t launch_pq_decode(hipStream_t s, const uint8_t *codes, const float *cb, float *dec, size_t n, int d, int M);
hipError_t launch_norm_codes(hipStream_t s, const float *rec, const float *ntab, uint8_t *norm_codes, float *norms_out,
                             size_t n, int d);
// code-book training (kernels_train.hip): the update half of a Lloyd iteration (the assignment half is
// launch_pq_encode), and OPQ's X^T Y on the matrix cores (chunks of kXtyChunk points, partials [nchunks][d][d])
constexpr int kXtyChunk = 2048;
hipError_t launch_lloyd_update(hipStream_t s, const float *x, const uint8_t *assign, float *cb, size_t n, int d, int M);
hipError_t launch_xty(hipStream_t s, const float *X, const float *Y, float *partials, float *C, size_t n, int d);
// exact k-nearest-neighbour tables on the matrix cores (kernels_knn.hip); part = [nsplit][nq][k] u64 workspace
int knn_splits_for(size_t nq, size_t nx);
hipError_t launch_knn_norms(hipStream_t s, const float *x, float *out, size_t n, int d);
hipError_t launch_knn(hipStream_t s, const float *Q, const float *X, const float *qn, const float *xn, size_t nq, size_t nx,
                      int d, int k, int mode, int nsplit, unsigned long long *part, uint32_t *ids, float *dists); // mode: IVFHNSW_KNN_*
// Grouping construction (IndexIVF_HNSW_Grouping.cpp:43-157)
hipError_t launch_group_table(hipStream_t s, int mode, const float *vectors, const uint32_t *centroid_idx,
                              const uint32_t *nn, const float *alphas, const float *cv_in, float *out, size_t ngroups,
                              int nsubc, int d);
size_t group_points_lds_bytes(int nsubc, int d);
hipError_t launch_group_points(hipStream_t s, int mode, const float *vectors, const uint32_t *centroid_idx,
                               const float *table, const float *cv_norms, const unsigned long long *offsets,
                               const float *x, float *out_num, float *out_den, uint32_t *out_sub, size_t ngroups,
                               int nsubc, int d);
hipError_t launch_group_alpha(hipStream_t s, const unsigned long long *offsets, const float *num, const float *den,
                              float *alphas, size_t ngroups);
hipError_t launch_group_rows(hipStream_t s, const unsigned long long *offsets, const uint32_t *sub, uint32_t *rows,
                             size_t ngroups, int nsubc);
// nbrows[i][j] = qrows[links[i][j]] for j < counts[i], zero otherwise (GraphTables::nbrows)
hipError_t launch_build_nbrows(hipStream_t s, const GraphTables &g, uint8_t *nbrows, uint32_t *nbnorms, int nb_rows,
                               uint32_t *links_c);
hipError_t launch_fill_bytes(hipStream_t s, uint8_t *dst, size_t nbytes, uint64_t seed);
hipError_t launch_fill_iota(hipStream_t s, uint32_t *dst, size_t n, uint32_t first);
hipError_t launch_fill_lists(hipStream_t s, const IvfTables &t, uint8_t *codes, uint8_t *norm_codes, uint32_t *ids,
                             uint64_t seed_codes, uint64_t seed_norms);
// sum of PlanHdr.total / nseg over the batch into out[0], out[1]
hipError_t launch_plan_totals(hipStream_t s, const PlanHdr *hdr, int nq, unsigned long long *out);

} // namespace ivfhnsw_gpu_impl
