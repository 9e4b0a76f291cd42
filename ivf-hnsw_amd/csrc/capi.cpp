// C ABI of the gfx950 IVFADC search path (include/ivfhnsw_hip.h): device state, uploads, the
// batched search pipeline (rotate -> coarse -> table -> plan -> scan -> select) and its measurement.
#include "../../include/ivfhnsw_hip.h"
#include "ivfhnsw_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h> // types and enum values only: the library itself is dlopen'ed on first use (no link dependency)
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace ivfhnsw_gpu_impl;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(e_ == hipErrorOutOfMemory ? IVFHNSW_ERR_NOMEM : IVFHNSW_ERR_HIP, "%s: %s (%s:%d)",   \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                   \
    } while (0)

// A device allocation that only ever grows.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes)
            return IVFHNSW_OK;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipMalloc(&p, need ? need : 1));
        bytes = need;
        return IVFHNSW_OK;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Pinned, device-visible host memory that only ever grows (the small-batch entry point reads queries and writes
// results through it: no staging copies on the latency path).
struct HostBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes)
            return IVFHNSW_OK;
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipHostMalloc(&p, need ? need : 1, hipHostMallocDefault));
        bytes = need;
        return IVFHNSW_OK;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct StageEvent {
    int stage;
    hipEvent_t a, b;
};

} // namespace

struct ivfhnsw_gpu {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool is_view = false; // ivfhnsw_gpu_create_view: tables belong to the parent, workspace and stream are its own

    // index tables
    DevBuf goff, loff, cnorm, pqc, ntab, opq_at, codes, ncodes, ids;
    IvfTables t{};
    bool has_ivf = false;
    uint64_t n_local = 0;
    DevBuf g_alpha, g_nn, g_sizes, g_inter;
    GroupTables g{};
    bool has_group = false;
    DevBuf q_counts, q_links, q_vectors, q_qrows, q_nbrows, q_nbnorms, q_fat, q_links_c;
    GraphTables gr{};
    bool has_graph = false;
    // construction side: code books for ivfhnsw_gpu_encode and its workspace
    DevBuf e_pqc, e_ntab, e_a, e_at, e_x, e_idx, e_dist, e_res, e_tmp, e_codes, e_ncodes;
    DevBuf t_x, t_y, t_cb, t_assign, t_part, t_c; // training (pq_train, xty)
    DevBuf k_q, k_x, k_qn, k_xn, k_part, k_ids, k_dists; // exact neighbour tables (ivfhnsw_gpu_knn)
    DevBuf cg_q, cg_cidx, cg_ids, cg_dists, gc_nn, cg_cvn, cg_tab, cg_tab2, cg_off, cg_alpha2, cg_sub; // add_group
    size_t e_d = 0, e_M = 0;
    bool e_opq = false, has_codebooks = false;

    // per-batch workspace
    // one large batch as two uneven parts on two streams (ivfhnsw_gpu_search_dev): the second part runs on this view
    ivfhnsw_gpu *split_view = nullptr;
    hipEvent_t split_fork = nullptr, split_join = nullptr;
    bool last_split = false;
    uint32_t *status_shared = nullptr; // the internal split view raises its status bits in the PARENT's word (no merge launch)
    size_t last_parts[2] = {0, 0}; // queries in the two parts of the last search_dev call (second 0 = one part)
    int split_pm = 0; // permille of a large batch in its first part; 0 = one part (ivfhnsw_gpu_set_batch_split)
    bool walk_counters_clean = true;  // w_status[1..4] are zero (ivfhnsw_gpu_create clears them, the redo launch's last wavefront restores it)
    bool visited_zero = false;        // every byte of w_visited is zero (the walk's overflow bitmaps, kernels_hnsw.hip)
    void *visited_zero_ptr = nullptr; // ... of this allocation
    size_t visited_zero_bytes = 0;
    DevBuf w_xq, w_luts, w_segs, w_lpos, w_hdr, w_keys, w_cid, w_cd, w_qsd, w_totals, w_visited, w_status, w_stream,
        w_slen, w_counter, w_tail, w_redo;
    int opt_scan_pipe = -1;      // ivfhnsw_gpu_set_option "scan_pipe"
    bool lat_defer_redo = false; // host-pointer small batches: the latency walk flags a tie overflow, the call repeats itself
    bool latency_off = false;    // ... on the throughput walk
    // staging for the host-pointer entry point
    DevBuf s_q, s_cid, s_cd, s_dist, s_lab, s_keys, s_len;
    HostBuf p_in, p_out; // pinned: small batches

    int last_nq = 0, last_max_seg = 0;
    const char *last_scan_kernel = "";
    uint32_t *tail_status_out = nullptr; // pinned word the tail kernel copies the status into (host-pointer path)
    bool tail_wrote_status = false;
    uint64_t *walk_zero_keys = nullptr; // the tail kernel's meeting words, cleared by the latency walk when it runs
    uint32_t *walk_zero_done = nullptr;
    bool walk_zeroed = false;
    bool last_stream = false; // the last search left a candidate stream (k > 1, heap_order)

    int profiling = 0; // 0 off, 1 every stage, 2 only the scan (an event pair costs ~7 us of stream time)
    std::vector<StageEvent> pending;
    std::vector<hipEvent_t> pool;
    double stage_ms[IVFHNSW_STAGE_COUNT] = {0};
    uint64_t stage_n[IVFHNSW_STAGE_COUNT] = {0};
};

namespace {

int bind(ivfhnsw_gpu *h)
{
    if (!h)
        return fail(IVFHNSW_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return IVFHNSW_OK;
}

// the word the kernels of this handle raise status bits in
inline uint32_t *status_word(ivfhnsw_gpu *h) { return h->status_shared ? h->status_shared : h->w_status.as<uint32_t>(); }

int upload(DevBuf &b, const void *src, size_t bytes)
{
    int rc = b.ensure(bytes);
    if (rc)
        return rc;
    if (bytes)
        HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return IVFHNSW_OK;
}

hipEvent_t take_event(ivfhnsw_gpu *h)
{
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct StageScope {
    ivfhnsw_gpu *h;
    StageEvent ev{};
    bool on;
    StageScope(ivfhnsw_gpu *h_, int stage)
        : h(h_), on(h_->profiling == 1 ||
                    (h_->profiling == 2 && stage == IVFHNSW_STAGE_SCAN))
    {
        if (!on)
            return;
        ev.stage = stage;
        ev.a = take_event(h);
        ev.b = take_event(h);
        (void)hipEventRecord(ev.a, h->stream);
    }
    ~StageScope()
    {
        if (!on)
            return;
        (void)hipEventRecord(ev.b, h->stream);
        h->pending.push_back(ev);
    }
};

int drain_events(ivfhnsw_gpu *h)
{
    for (auto &ev : h->pending) {
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        h->stage_ms[ev.stage] += ms;
        h->stage_n[ev.stage] += 1;
        h->pool.push_back(ev.a);
        h->pool.push_back(ev.b);
    }
    h->pending.clear();
    return IVFHNSW_OK;
}

// candidate-stream entries per query kept for the heap-order replay (k > 1)
static const uint32_t kHeapStreamCap = 8192;

// After a stream sync: did any kernel flag something it could not represent?
int check_status(ivfhnsw_gpu *h)
{
    uint32_t st = 0;
    HIP_TRY(hipMemcpy(&st, h->w_status.p, sizeof(st), hipMemcpyDeviceToHost));
    if (!st)
        return IVFHNSW_OK;
    HIP_TRY(hipMemset(h->w_status.p, 0, sizeof(st)));
    if (st & kStatusHnswTieOverflow)
        return fail(IVFHNSW_ERR_STATE, "HNSW walk: more than 64 candidates tie exactly with the efSearch-th "
                                       "distance; results of this batch are invalid");
    if (st & kStatusTopkStreamOverflow)
        return fail(IVFHNSW_ERR_STATE, "heap-order top-k: candidate stream of a query exceeded %u entries; use "
                                       "heap_order = 0 for this k / max_codes", kHeapStreamCap);
    return fail(IVFHNSW_ERR_STATE, "device status 0x%x", st);
}

int check_desc(const ivfhnsw_ivf_desc *d, bool need_lists)
{
    if (!d)
        return fail(IVFHNSW_ERR_INVALID, "null descriptor");
    if (d->d == 0 || d->nc == 0 || d->code_size == 0)
        return fail(IVFHNSW_ERR_INVALID, "d, nc and code_size must be positive");
    if (d->code_size % 4)
        return fail(IVFHNSW_ERR_INVALID, "code_size %zu is not a multiple of 4 (IndexIVF_HNSW.cpp:805)", d->code_size);
    if (d->code_size * 1024 > kScanDynLdsMax)
        return fail(IVFHNSW_ERR_INVALID, "code_size %zu: the query's table (1 KB per code byte) must fit %zu KB of LDS",
                    d->code_size, kScanDynLdsMax / 1024);
    if (d->d % d->code_size)
        return fail(IVFHNSW_ERR_INVALID, "d %zu is not a multiple of code_size %zu", d->d, d->code_size);
    if (d->d / d->code_size > 64)
        return fail(IVFHNSW_ERR_INVALID, "sub-vector dimension %zu > 64 unsupported", d->d / d->code_size);
    if (d->nc >= 0xffffffffull)
        return fail(IVFHNSW_ERR_INVALID, "nc too large");
    if (!d->offsets || !d->centroid_norms || !d->pq_centroids || !d->norm_table)
        return fail(IVFHNSW_ERR_INVALID, "offsets, centroid_norms, pq_centroids and norm_table are required");
    if (need_lists && d->offsets[d->nc] != 0 && (!d->ids || !d->codes || !d->norm_codes))
        return fail(IVFHNSW_ERR_INVALID, "ids, codes and norm_codes are required");
    if (d->shard_world == 0 || d->shard_rank >= d->shard_world)
        return fail(IVFHNSW_ERR_INVALID, "bad shard %u of %u", d->shard_rank, d->shard_world);
    if (d->offsets[0] != 0)
        return fail(IVFHNSW_ERR_INVALID, "offsets[0] must be 0");
    for (size_t c = 0; c < d->nc; c++)
        if (d->offsets[c + 1] < d->offsets[c])
            return fail(IVFHNSW_ERR_INVALID, "offsets not monotone at list %zu", c);
    return IVFHNSW_OK;
}

// Tables shared by upload_ivf and upload_ivf_synthetic; fills h->t except codes/norm_codes/ids.
int upload_tables(ivfhnsw_gpu *h, const ivfhnsw_ivf_desc *d, std::vector<uint32_t> &loff, uint64_t &n_local)
{
    loff.assign(d->nc, kNotOwned);
    n_local = 0;
    for (size_t c = 0; c < d->nc; c++) {
        const uint32_t owner = d->list_owner ? d->list_owner[c] : (uint32_t)(c % d->shard_world);
        if (owner >= d->shard_world)
            return fail(IVFHNSW_ERR_INVALID, "list_owner[%zu] = %u is not a rank of %u", c, owner, d->shard_world);
        if (owner != d->shard_rank)
            continue;
        if (n_local >= 0xffffffffull)
            return fail(IVFHNSW_ERR_INVALID, "2^32 - 1 or more codes on one shard");
        loff[c] = (uint32_t)n_local;
        n_local += d->offsets[c + 1] - d->offsets[c];
    }
    if (n_local >= 0xffffffffull)
        return fail(IVFHNSW_ERR_INVALID, "2^32 - 1 or more codes on one shard");
    int rc;
    if ((rc = upload(h->goff, d->offsets, (d->nc + 1) * sizeof(uint64_t))))
        return rc;
    if ((rc = upload(h->loff, loff.data(), d->nc * sizeof(uint32_t))))
        return rc;
    if ((rc = upload(h->cnorm, d->centroid_norms, d->nc * sizeof(float))))
        return rc;
    if ((rc = upload(h->pqc, d->pq_centroids, 256 * d->d * sizeof(float))))
        return rc;
    if ((rc = upload(h->ntab, d->norm_table, 256 * sizeof(float))))
        return rc;
    if (d->opq_A) {
        std::vector<float> at(d->d * d->d);
        for (size_t i = 0; i < d->d; i++)
            for (size_t k = 0; k < d->d; k++)
                at[k * d->d + i] = d->opq_A[i * d->d + k];
        if ((rc = upload(h->opq_at, at.data(), at.size() * sizeof(float))))
            return rc;
    } else {
        h->opq_at.release();
    }
    IvfTables &t = h->t;
    t.d = (int)d->d;
    t.M = (int)d->code_size;
    t.dsub = (int)(d->d / d->code_size);
    t.nc = (uint32_t)d->nc;
    t.goff = h->goff.as<uint64_t>();
    t.loff = h->loff.as<uint32_t>();
    t.centroid_norms = h->cnorm.as<float>();
    t.pq_centroids = h->pqc.as<float>();
    t.norm_table = h->ntab.as<float>();
    t.opq_At = d->opq_A ? h->opq_at.as<float>() : nullptr;
    t.shard_rank = d->shard_rank;
    t.shard_world = d->shard_world;
    return IVFHNSW_OK;
}

} // namespace

// for the library's other translation units (graph_build.cpp): record a failure the way every entry point does
int ivfhnsw_gpu_fail_msg(int code, const char *msg) { return fail(code, "%s", msg); }

extern "C" {

const char *ivfhnsw_gpu_last_error(void) { return g_last_error.c_str(); }

int ivfhnsw_gpu_abi_version(void) { return 9; }

int ivfhnsw_gpu_device_count(int *count)
{
    if (!count)
        return fail(IVFHNSW_ERR_INVALID, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return fail(IVFHNSW_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    *count = n;
    return IVFHNSW_OK;
}

static const int kSplitAuto = 1000; // the first part's share follows the call's walk : table + plan + scan estimate

static int split_permille_env()
{
    static const int v = [] {
        // on by default since round 3; IVFHNSW_SPLIT=0 = one part, 1..999 = that share in the first part, unset = by estimate
        const char *e = getenv("IVFHNSW_SPLIT");
        const int x = (e && *e) ? atoi(e) : kSplitAuto;
        return (x > 0 && x <= kSplitAuto) ? x : 0;
    }();
    return v;
}

int ivfhnsw_gpu_create(int device, ivfhnsw_gpu **out)
{
    if (!out)
        return fail(IVFHNSW_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(IVFHNSW_ERR_HIP, "no HIP device available (%s); there is no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(IVFHNSW_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(IVFHNSW_ERR_HIP, "device %d is %s; this library holds gfx950 code only", device, prop.gcnArchName);
    ivfhnsw_gpu *h = new ivfhnsw_gpu();
    h->device = device;
    hipError_t se = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete h;
        return fail(IVFHNSW_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se));
    }
    h->own_stream = true;
    h->split_pm = split_permille_env();
    // [0] status bits, [1] the walk's query counter, [2] queries on the redo list, [3] the redo launch's counter, [4] its exit count
    if (h->w_status.ensure(8 * sizeof(uint32_t)) || hipMemset(h->w_status.p, 0, 8 * sizeof(uint32_t)) != hipSuccess) {
        ivfhnsw_gpu_destroy(h);
        return fail(IVFHNSW_ERR_HIP, "cannot allocate the device status word");
    }
    *out = h;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_destroy(ivfhnsw_gpu *h)
{
    if (!h)
        return IVFHNSW_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->split_view) {
        ivfhnsw_gpu_destroy(h->split_view);
        h->split_view = nullptr;
        (void)hipSetDevice(h->device);
    }
    if (h->split_fork)
        (void)hipEventDestroy(h->split_fork);
    if (h->split_join)
        (void)hipEventDestroy(h->split_join);
    for (auto &ev : h->pending) {
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    for (auto e : h->pool)
        (void)hipEventDestroy(e);
    DevBuf *all[] = {&h->goff, &h->loff, &h->cnorm, &h->pqc, &h->ntab, &h->opq_at, &h->codes, &h->ncodes, &h->ids,
                     &h->g_alpha, &h->g_nn, &h->g_sizes, &h->g_inter, &h->q_counts, &h->q_links, &h->q_vectors, &h->q_qrows, &h->q_nbrows, &h->q_nbnorms, &h->q_fat, &h->q_links_c, &h->e_pqc, &h->e_ntab, &h->e_a, &h->e_at, &h->e_x, &h->e_idx, &h->e_dist, &h->e_res, &h->e_tmp, &h->e_codes, &h->e_ncodes, &h->cg_q, &h->cg_cidx, &h->cg_ids, &h->cg_dists, &h->gc_nn, &h->cg_cvn, &h->cg_tab, &h->cg_tab2, &h->cg_off, &h->cg_alpha2, &h->cg_sub,
                     &h->w_xq, &h->w_luts, &h->w_segs, &h->w_lpos, &h->w_hdr, &h->w_keys, &h->w_cid, &h->w_cd,
                     &h->w_qsd, &h->w_totals, &h->w_visited, &h->w_status, &h->w_stream, &h->w_slen, &h->w_counter, &h->w_tail, &h->w_redo, &h->k_q, &h->k_x, &h->k_qn, &h->k_xn, &h->k_part, &h->k_ids, &h->k_dists, &h->t_x, &h->t_y, &h->t_cb, &h->t_assign, &h->t_part, &h->t_c, &h->s_q, &h->s_cid, &h->s_cd, &h->s_dist, &h->s_lab, &h->s_keys, &h->s_len};
    for (auto *b : all)
        b->release();
    h->p_in.release();
    h->p_out.release();
    if (h->own_stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_create_view(ivfhnsw_gpu *parent, ivfhnsw_gpu **out)
{
    if (!out)
        return fail(IVFHNSW_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int rc = bind(parent);
    if (rc)
        return rc;
    if (parent->is_view)
        return fail(IVFHNSW_ERR_INVALID, "a view of a view: create it from the handle that holds the tables");
    // the parent's uploads (and the neighbour-row build) run on its stream: finished before anyone reads them
    HIP_TRY(hipStreamSynchronize(parent->stream));
    ivfhnsw_gpu *h = nullptr;
    if ((rc = ivfhnsw_gpu_create(parent->device, &h)))
        return rc;
    h->is_view = true;
    h->split_pm = 0;
    h->t = parent->t;
    h->has_ivf = parent->has_ivf;
    h->n_local = parent->n_local;
    h->g = parent->g;
    h->has_group = parent->has_group;
    h->gr = parent->gr;
    h->has_graph = parent->has_graph;
    *out = h;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_set_stream(ivfhnsw_gpu *h, void *hip_stream)
{
    int rc = bind(h);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream)
        (void)hipStreamDestroy(h->stream);
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_sync(ivfhnsw_gpu *h)
{
    int rc = bind(h);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return check_status(h);
}

int ivfhnsw_gpu_upload_ivf(ivfhnsw_gpu *h, const ivfhnsw_ivf_desc *d)
try {
    if (h && h->is_view)
        return fail(IVFHNSW_ERR_STATE, "uploads go to the handle that holds the tables, not to a view of it");
    int rc = bind(h);
    if (rc)
        return rc;
    if ((rc = check_desc(d, true)))
        return rc;
    h->has_ivf = false;
    std::vector<uint32_t> loff;
    uint64_t n_local = 0;
    if ((rc = upload_tables(h, d, loff, n_local)))
        return rc;
    if ((rc = upload(h->codes, d->codes, n_local * d->code_size)))
        return rc;
    if ((rc = upload(h->ncodes, d->norm_codes, n_local)))
        return rc;
    if ((rc = upload(h->ids, d->ids, n_local * sizeof(uint32_t))))
        return rc;
    h->t.codes = h->codes.as<uint8_t>();
    h->t.norm_codes = h->ncodes.as<uint8_t>();
    h->t.ids = h->ids.as<uint32_t>();
    h->n_local = n_local;
    h->has_ivf = true;
    h->has_group = false;
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_upload_ivf: host allocation failed");
}

int ivfhnsw_gpu_upload_ivf_synthetic(ivfhnsw_gpu *h, const ivfhnsw_ivf_desc *d, uint64_t seed)
try {
    if (h && h->is_view)
        return fail(IVFHNSW_ERR_STATE, "uploads go to the handle that holds the tables, not to a view of it");
    int rc = bind(h);
    if (rc)
        return rc;
    if ((rc = check_desc(d, false)))
        return rc;
    h->has_ivf = false;
    std::vector<uint32_t> loff;
    uint64_t n_local = 0;
    if ((rc = upload_tables(h, d, loff, n_local)))
        return rc;
    if ((rc = h->codes.ensure(n_local * d->code_size)))
        return rc;
    if ((rc = h->ncodes.ensure(n_local)))
        return rc;
    if ((rc = h->ids.ensure(n_local * sizeof(uint32_t))))
        return rc;
    if (d->offsets[d->nc] > 0xffffffffull)
        return fail(IVFHNSW_ERR_INVALID, "synthetic corpus: more than 2^32 vectors (ids are uint32)");
    HIP_TRY(launch_fill_lists(h->stream, h->t, h->codes.as<uint8_t>(), h->ncodes.as<uint8_t>(), h->ids.as<uint32_t>(),
                              seed, seed ^ 0x6e6f726d6e6f726dull));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->t.codes = h->codes.as<uint8_t>();
    h->t.norm_codes = h->ncodes.as<uint8_t>();
    h->t.ids = h->ids.as<uint32_t>();
    h->n_local = n_local;
    h->has_ivf = true;
    h->has_group = false;
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_upload_ivf_synthetic: host allocation failed");
}

int ivfhnsw_gpu_upload_grouping(ivfhnsw_gpu *h, size_t nsubc, const float *alphas, const uint32_t *nn_centroid_idxs,
                                const uint32_t *subgroup_sizes, const float *inter_centroid_dists)
try {
    if (h && h->is_view)
        return fail(IVFHNSW_ERR_STATE, "uploads go to the handle that holds the tables, not to a view of it");
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf)
        return fail(IVFHNSW_ERR_STATE, "upload_grouping before upload_ivf");
    if (nsubc == 0 || nsubc > 4096 || !alphas || !nn_centroid_idxs || !subgroup_sizes || !inter_centroid_dists)
        return fail(IVFHNSW_ERR_INVALID, "bad grouping tables (nsubc %zu)", nsubc);
    const size_t nc = h->t.nc;
    // every group's sub-group sizes must add up to the list size, and neighbours must be valid ids
    {
        std::vector<uint64_t> goff(nc + 1);
        HIP_TRY(hipMemcpy(goff.data(), h->goff.p, (nc + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (size_t c = 0; c < nc; c++) {
            uint64_t s = 0;
            for (size_t j = 0; j < nsubc; j++) {
                s += subgroup_sizes[c * nsubc + j];
                if (subgroup_sizes[c * nsubc + j] && nn_centroid_idxs[c * nsubc + j] >= nc)
                    return fail(IVFHNSW_ERR_INVALID, "nn_centroid_idxs[%zu][%zu] out of range", c, j);
            }
            if (s != goff[c + 1] - goff[c])
                return fail(IVFHNSW_ERR_INVALID, "subgroup sizes of list %zu sum to %llu, list holds %llu", c,
                            (unsigned long long)s, (unsigned long long)(goff[c + 1] - goff[c]));
        }
    }
    if ((rc = upload(h->g_alpha, alphas, nc * sizeof(float))))
        return rc;
    if ((rc = upload(h->g_nn, nn_centroid_idxs, nc * nsubc * sizeof(uint32_t))))
        return rc;
    if ((rc = upload(h->g_sizes, subgroup_sizes, nc * nsubc * sizeof(uint32_t))))
        return rc;
    if ((rc = upload(h->g_inter, inter_centroid_dists, nc * nsubc * sizeof(float))))
        return rc;
    h->g.nsubc = (int)nsubc;
    h->g.alphas = h->g_alpha.as<float>();
    h->g.nn_idx = h->g_nn.as<uint32_t>();
    h->g.sub_sizes = h->g_sizes.as<uint32_t>();
    h->g.inter_dists = h->g_inter.as<float>();
    // How much do the neighbour lists of groups a query probes together overlap?  Sampled: a group and its 15 nearest
    // neighbour groups stand for a query's probes; the share of DISTINCT ids in their 16 lists.  Clustered centroids
    // (k-means of real descriptors): ~0.1-0.3, the plan's hash set saves most row gathers; iid synthetic: ~0.7, it costs
    // more than it saves (measured).
    {
        const size_t take = std::min<size_t>(15, nsubc), step = std::max<size_t>(1, nc / 512);
        double distinct = 0, total = 0;
        std::vector<uint32_t> ids;
        for (size_t c = 0; c < nc; c += step) {
            ids.clear();
            auto add_list = [&](size_t cc) {
                for (size_t j = 0; j < nsubc; j++)
                    if (subgroup_sizes[cc * nsubc + j])
                        ids.push_back(nn_centroid_idxs[cc * nsubc + j]);
            };
            add_list(c);
            for (size_t j = 0; j < take; j++)
                if (nn_centroid_idxs[c * nsubc + j] < nc)
                    add_list(nn_centroid_idxs[c * nsubc + j]);
            total += (double)ids.size();
            std::sort(ids.begin(), ids.end());
            distinct += (double)(std::unique(ids.begin(), ids.end()) - ids.begin());
        }
        h->g.dedupe = (total > 0 && distinct / total < 0.55) ? 1 : 0;
    }
    h->has_group = true;
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_upload_grouping: host allocation failed");
}

// The walk's exact rejection filter (kernels_hnsw.hip): one byte per component, x ~ lo + step * byte with one
// (lo, step) for the whole table, and the largest row error ||x - (lo + step*byte)|| in units of step, rounded
// up.  Tables with non-finite values, a single value or d > 2048 run without the filter (always exact, only
// slower); IVFHNSW_WALK_PREFILTER=0 turns it off for A/B runs.
static int build_byte_rows(ivfhnsw_gpu *h, size_t n, size_t d, const float *vectors)
{
    static const bool off = [] {
        const char *e = getenv("IVFHNSW_WALK_PREFILTER");
        return e && atoi(e) == 0;
    }();
    if (off || d > 2048)
        return IVFHNSW_OK;
    float lo = vectors[0], hi = vectors[0];
    bool finite = true;
    for (size_t i = 0; i < n * d; i++) {
        const float v = vectors[i];
        finite &= std::isfinite(v);
        lo = std::min(lo, v);
        hi = std::max(hi, v);
    }
    const float step = (float)(((double)hi - (double)lo) / 255.0);
    if (!finite || !(step > 0.f) || !std::isfinite(step) || !std::isfinite(1.f / step))
        return IVFHNSW_OK;
    std::vector<uint8_t> rows(n * d);
    double worst = 0.0;
    for (size_t r = 0; r < n; r++) {
        double e2 = 0.0;
        for (size_t j = 0; j < d; j++) {
            const double x = vectors[r * d + j];
            double c = std::nearbyint((x - (double)lo) / (double)step);
            c = std::min(255.0, std::max(0.0, c));
            rows[r * d + j] = (uint8_t)c;
            const double e = x - ((double)lo + (double)step * c);
            e2 += e * e;
        }
        worst = std::max(worst, e2);
    }
    const double errc = std::sqrt(worst) / (double)step * (1.0 + 1e-6) + 1e-6;
    float errc_f = (float)errc;
    if ((double)errc_f < errc)
        errc_f = std::nextafter(errc_f, INFINITY);
    int rc = upload(h->q_qrows, rows.data(), n * d);
    if (rc)
        return rc;
    h->gr.qrows = h->q_qrows.as<uint8_t>();
    h->gr.q_lo = lo;
    h->gr.q_step = step;
    h->gr.q_errc = errc_f;
    return IVFHNSW_OK;
}

// Second copy of the byte rows in walk order (GraphTables::nbrows): 128 B per link, so n * maxM * 128 bytes --
// 4 GiB for the reference's 993127-centroid, maxM 32 quantizer.  Skipped (the walk then gathers from qrows)
// for d > 128, above IVFHNSW_WALK_NBROWS_GIB (default 48) and with IVFHNSW_WALK_PREFILTER=1.
static int build_neighbour_rows(ivfhnsw_gpu *h)
{
    static const int mode = [] {
        const char *e = getenv("IVFHNSW_WALK_PREFILTER");
        return e ? atoi(e) : 2;
    }();
    static const double cap_gib = [] {
        const char *e = getenv("IVFHNSW_WALK_NBROWS_GIB");
        return e ? atof(e) : 48.0;
    }();
    if (!h->gr.qrows || mode != 2 || h->gr.d > 128 || h->gr.n >= (1u << 24) || h->gr.maxM > 64)
        return IVFHNSW_OK;
    const int nb_rows = (h->gr.maxM + 31) & ~31;
    const size_t bytes = (size_t)h->gr.n * nb_rows * 128;
    if ((double)bytes > cap_gib * 1073741824.0)
        return IVFHNSW_OK;
    int rc = h->q_nbrows.ensure(bytes);
    if (rc)
        return rc;
    if ((rc = h->q_nbnorms.ensure((size_t)h->gr.n * nb_rows * sizeof(uint32_t))))
        return rc;
    if ((rc = h->q_links_c.ensure((size_t)h->gr.n * h->gr.maxM * sizeof(uint32_t))))
        return rc;
    HIP_TRY(launch_build_nbrows(h->stream, h->gr, h->q_nbrows.as<uint8_t>(), h->q_nbnorms.as<uint32_t>(), nb_rows,
                                h->q_links_c.as<uint32_t>()));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->gr.nbrows = h->q_nbrows.as<uint8_t>();
    h->gr.nbnorms = h->q_nbnorms.as<uint32_t>();
    h->gr.links_c = h->q_links_c.as<uint32_t>();
    h->gr.nb_rows = nb_rows;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_upload_quantizer(ivfhnsw_gpu *h, size_t n, size_t d, size_t maxM, uint32_t enterpoint,
                                 const uint8_t *link_counts, const uint32_t *links, const float *vectors)
try {
    if (h && h->is_view)
        return fail(IVFHNSW_ERR_STATE, "uploads go to the handle that holds the tables, not to a view of it");
    int rc = bind(h);
    if (rc)
        return rc;
    if (n == 0 || d == 0 || maxM == 0 || maxM > 255 || n >= 0xffffffffull || enterpoint >= n || !link_counts ||
        !links || !vectors)
        return fail(IVFHNSW_ERR_INVALID, "bad quantizer arrays (n %zu, d %zu, maxM %zu)", n, d, maxM);
    if (d % 16)
        return fail(IVFHNSW_ERR_INVALID, "d %zu: the reference distance ignores dims beyond a multiple of 16 "
                                         "(hnswalg.cpp:330); only multiples of 16 are supported here", d);
    // Link lists without repeated ids (every graph the reference builds) is what the walk's visited set assumes
    // (kernels_hnsw.hip, VisFields).  A repeated id has no effect in the reference -- the second occurrence is
    // skipped as visited (hnswalg.cpp:80-82) -- so lists that have them are uploaded without the repeats.
    bool unique = true;
    uint32_t tmp[256];
    for (size_t i = 0; i < n; i++) {
        if (link_counts[i] > maxM)
            return fail(IVFHNSW_ERR_INVALID, "node %zu has %u links > maxM %zu", i, link_counts[i], maxM);
        for (size_t j = 0; j < link_counts[i]; j++)
            if (links[i * maxM + j] >= n)
                return fail(IVFHNSW_ERR_INVALID, "node %zu link %zu out of range", i, j);
        if (unique && link_counts[i] > 1) {
            const size_t c = link_counts[i];
            std::copy(links + i * maxM, links + i * maxM + c, tmp);
            std::sort(tmp, tmp + c);
            unique = std::adjacent_find(tmp, tmp + c) == tmp + c;
        }
    }
    std::vector<uint8_t> counts_u;
    std::vector<uint32_t> links_u;
    if (!unique) {
        counts_u.assign(link_counts, link_counts + n);
        links_u.assign(links, links + n * maxM);
        for (size_t i = 0; i < n; i++) {
            uint32_t *row = links_u.data() + i * maxM;
            size_t c = 0;
            for (size_t j = 0; j < link_counts[i]; j++) {
                bool seen = false;
                for (size_t k2 = 0; k2 < c && !seen; k2++)
                    seen = row[k2] == row[j];
                if (!seen)
                    row[c++] = row[j];
            }
            for (size_t j = c; j < link_counts[i]; j++)
                row[j] = 0;
            counts_u[i] = (uint8_t)c;
        }
        link_counts = counts_u.data();
        links = links_u.data();
        unique = true;
    }
    if ((rc = upload(h->q_counts, link_counts, n)))
        return rc;
    if ((rc = upload(h->q_links, links, n * maxM * sizeof(uint32_t))))
        return rc;
    if ((rc = upload(h->q_vectors, vectors, n * d * sizeof(float))))
        return rc;
    h->gr.n = (uint32_t)n;
    h->gr.d = (int)d;
    h->gr.maxM = (int)maxM;
    h->gr.enterpoint = enterpoint;
    h->gr.counts = h->q_counts.as<uint8_t>();
    h->gr.links = h->q_links.as<uint32_t>();
    h->gr.vectors = h->q_vectors.as<float>();
    {
        static const int merge = [] {
            const char *e = getenv("IVFHNSW_WALK_MERGE");
            return (e && atoi(e) == 0) ? 0 : 1;
        }();
        h->gr.merge_admissions = merge;
        h->gr.skip_padding = 1;
    }
    {
        // IVFHNSW_WALK_LATE_VISIT: 1 always, 0 never, unset = where it pays -- graphs whose ids need more than 8 tag
        // bits (beyond 255 * 1008 nodes), where the visited set of a query would otherwise run 40 % full and
        // overflow into global atomics (1.60 -> 1.50 ms per 10 k queries at 993 127 nodes; at 2^17 nodes the extra
        // LDS pass costs 1 %: 1.19 -> 1.205 ms)
        static const int late_knob = [] {
            const char *e = getenv("IVFHNSW_WALK_LATE_VISIT");
            return e ? (atoi(e) != 0 ? 1 : 0) : -1;
        }();
        const bool late = late_knob < 0 ? n > 255u * 1008u : late_knob == 1;
        h->gr.links_unique = (unique && late) ? 1 : 0;
    }
    h->gr.fat = nullptr; // the latency form's copy belongs to the previous graph
    h->gr.qrows = nullptr;
    h->gr.nbrows = nullptr;
    h->gr.nbnorms = nullptr;
    h->gr.links_c = nullptr;
    h->gr.nb_rows = 0;
    h->gr.q_lo = 0.f;
    h->gr.q_step = 1.f;
    h->gr.q_errc = 0.f;
    if ((rc = build_byte_rows(h, n, d, vectors)))
        return rc;
    if ((rc = build_neighbour_rows(h)))
        return rc;
    h->has_graph = true;
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_upload_quantizer: host allocation failed");
}

int ivfhnsw_gpu_prepare_latency(ivfhnsw_gpu *h)
{
    if (h && h->is_view)
        return fail(IVFHNSW_ERR_STATE, "prepare_latency goes to the handle that holds the tables, not to a view of it");
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_graph)
        return fail(IVFHNSW_ERR_STATE, "prepare_latency needs upload_quantizer");
    if (h->gr.fat)
        return IVFHNSW_OK;
    GraphTables probe = h->gr;
    probe.fat = reinterpret_cast<const float *>(h); // any non-null value: shape check only
    if (!coarse_latency_supported(probe, 1))
        return fail(IVFHNSW_ERR_INVALID, "the latency walk needs d = 128 or 96, maxM <= 32 and at most 2^20 nodes "
                                         "(have d %d, maxM %d, %u nodes); small batches keep the throughput walk",
                    h->gr.d, h->gr.maxM, h->gr.n);
    if ((rc = h->q_fat.ensure(coarse_latency_fat_bytes(h->gr))))
        return rc;
    HIP_TRY(launch_build_fat(h->stream, h->gr, h->q_fat.as<float>()));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->gr.fat = h->q_fat.as<float>();
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_coarse_dev(ivfhnsw_gpu *h, size_t nq, const float *d_queries, size_t nprobe, size_t efSearch,
                           uint32_t *d_coarse_ids, float *d_coarse_dists)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_graph)
        return fail(IVFHNSW_ERR_STATE, "coarse search needs upload_quantizer");
    if (nprobe == 0 || efSearch < nprobe)
        return fail(IVFHNSW_ERR_INVALID, "efSearch %zu < nprobe %zu (precondition of IndexIVF_HNSW.cpp:249-258)",
                    efSearch, nprobe);
    if (efSearch > 1024)
        return fail(IVFHNSW_ERR_INVALID, "efSearch %zu > 1024 unsupported on the device", efSearch);
    if (nq == 0)
        return IVFHNSW_OK;
    if (nq > 0x7fffffffull)
        return fail(IVFHNSW_ERR_INVALID, "nq too large");
    // one visited bitmap per query in flight
    const size_t words = ((((size_t)h->gr.n + 31) / 32) + 3) & ~(size_t)3;
    StageScope sc(h, IVFHNSW_STAGE_COARSE);
    // few queries (the reference's drivers: one per call): a workgroup per query on the fat graph, when it was prepared
    static const size_t lat_max_nq = [] {
        const char *e = getenv("IVFHNSW_LATENCY_MAX_NQ");
        return (e && *e) ? (size_t)atol(e) : (size_t)256;
    }();
    // scratch of the walk: first half the visited bitmaps (the LDS set's overflow store), second half the tail bitmaps
    // of the redo form (walk_set.h TailSpill) -- always at the middle of the ALLOCATION, so that no launch's visited area
    // ever overlaps them; both halves zero between launches
    auto walk_scratch = [&](size_t nslots, uint32_t **tails) -> int {
        int r = h->w_visited.ensure(words * sizeof(uint32_t) * std::max<size_t>(nslots, 64) * 2);
        if (r)
            return r;
        if (h->w_visited.p != h->visited_zero_ptr || h->w_visited.bytes != h->visited_zero_bytes) {
            HIP_TRY(hipMemsetAsync(h->w_visited.p, 0, h->w_visited.bytes, h->stream)); // (re)allocated: contents unknown
            h->visited_zero = true;
            h->visited_zero_ptr = h->w_visited.p;
            h->visited_zero_bytes = h->w_visited.bytes;
        }
        *tails = reinterpret_cast<uint32_t *>(h->w_visited.as<char>() + h->w_visited.bytes / 2);
        return h->w_redo.ensure(nq * sizeof(uint32_t));
    };
    uint32_t *tails = nullptr;
    if (!h->latency_off && nq <= lat_max_nq && coarse_latency_supported(h->gr, (int)efSearch)) {
        uint32_t *hdr = h->w_status.as<uint32_t>() + 2;
        const bool defer = h->lat_defer_redo; // the caller synchronises and reads the status word itself
        if (!defer) {
            if ((rc = walk_scratch(64, &tails)))
                return rc;
            HIP_TRY(hipMemsetAsync(hdr, 0, 3 * sizeof(uint32_t), h->stream)); // length, counter, exit count
        }
        HIP_TRY(launch_coarse_latency(h->stream, h->gr, d_queries, (int)nq, (int)nprobe, (int)efSearch, d_coarse_ids,
                                      d_coarse_dists, status_word(h), h->walk_zero_keys, h->walk_zero_done,
                                      defer ? nullptr : hdr, defer ? nullptr : h->w_redo.as<uint32_t>()));
        h->walk_zeroed = h->walk_zero_keys != nullptr;
        h->walk_zero_keys = nullptr; // consumed: set by search_dev right before the call, never carried over
        h->walk_zero_done = nullptr;
        if (!defer)
            HIP_TRY(launch_coarse_redo(h->stream, h->gr, d_queries, (int)nq, (int)nprobe, (int)efSearch, d_coarse_ids,
                                       d_coarse_dists, h->w_visited.as<uint32_t>(), words, status_word(h), hdr,
                                       h->w_redo.as<uint32_t>(), tails, 64));
        return IVFHNSW_OK;
    }
    {
        const int nslots = (int)std::min<size_t>(nq, (size_t)coarse_slots_for((int)efSearch));
        if ((rc = walk_scratch((size_t)nslots, &tails)))
            return rc;
        HIP_TRY(launch_coarse(h->stream, h->gr, d_queries, (int)nq, (int)nprobe, (int)efSearch, d_coarse_ids,
                              d_coarse_dists, h->w_visited.as<uint32_t>(), words, nslots, status_word(h),
                              h->w_status.as<uint32_t>() + 1, h->w_visited.bytes / 2, &h->visited_zero,
                              h->w_redo.as<uint32_t>(), tails, 64, &h->walk_counters_clean));
    }
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_rotate_dev(ivfhnsw_gpu *h, size_t nq, const float *d_queries, float *d_out)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf)
        return fail(IVFHNSW_ERR_STATE, "rotate needs upload_ivf");
    if (nq == 0)
        return IVFHNSW_OK;
    if (!d_queries || !d_out || nq > 0x7fffffffull)
        return fail(IVFHNSW_ERR_INVALID, "null buffer or nq too large");
    if (h->t.opq_At) {
        StageScope sc(h, IVFHNSW_STAGE_OPQ);
        HIP_TRY(launch_opq(h->stream, h->t.opq_At, d_queries, d_out, (int)nq, h->t.d));
    } else if (d_out != d_queries) {
        HIP_TRY(hipMemcpyAsync(d_out, d_queries, nq * (size_t)h->t.d * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_coarse(ivfhnsw_gpu *h, size_t nq, const float *queries, size_t k, size_t efSearch, uint32_t *ids,
                       float *dists)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_graph)
        return fail(IVFHNSW_ERR_STATE, "coarse search needs upload_quantizer");
    if (nq == 0)
        return IVFHNSW_OK;
    if (!queries || !ids || !dists || k == 0)
        return fail(IVFHNSW_ERR_INVALID, "null buffer or k == 0");
    const size_t d = h->gr.d;
    if ((rc = h->s_q.ensure(nq * d * sizeof(float))))
        return rc;
    if ((rc = h->s_cid.ensure(nq * k * sizeof(uint32_t))))
        return rc;
    if ((rc = h->s_cd.ensure(nq * k * sizeof(float))))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->s_q.p, queries, nq * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if ((rc = ivfhnsw_gpu_coarse_dev(h, nq, h->s_q.as<float>(), k, efSearch, h->s_cid.as<uint32_t>(),
                                     h->s_cd.as<float>())))
        return rc;
    HIP_TRY(hipMemcpyAsync(ids, h->s_cid.p, nq * k * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(dists, h->s_cd.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return check_status(h);
}

// ---------------------------------------------------------------------------------------------------------------
// construction side
int ivfhnsw_gpu_upload_codebooks(ivfhnsw_gpu *h, size_t d, size_t code_size, const float *pq_centroids,
                                 const float *norm_table, const float *opq_A)
try {
    if (h && h->is_view)
        return fail(IVFHNSW_ERR_STATE, "uploads go to the handle that holds the tables, not to a view of it");
    int rc = bind(h);
    if (rc)
        return rc;
    if (d == 0 || code_size == 0 || d % code_size || d / code_size > 64 || !pq_centroids || !norm_table)
        return fail(IVFHNSW_ERR_INVALID, "bad code books (d %zu, code_size %zu)", d, code_size);
    h->has_codebooks = false;
    if ((rc = upload(h->e_pqc, pq_centroids, 256 * d * sizeof(float))))
        return rc;
    if ((rc = upload(h->e_ntab, norm_table, 256 * sizeof(float))))
        return rc;
    h->e_opq = opq_A != nullptr;
    if (opq_A) {
        // both orientations: apply reads A transposed, transform_transpose reads it as it is (launch_opq takes
        // the matrix of y = B x stored as B^T)
        std::vector<float> at(d * d);
        for (size_t i = 0; i < d; i++)
            for (size_t k = 0; k < d; k++)
                at[k * d + i] = opq_A[i * d + k];
        if ((rc = upload(h->e_at, at.data(), d * d * sizeof(float))))
            return rc;
        if ((rc = upload(h->e_a, opq_A, d * d * sizeof(float))))
            return rc;
    }
    h->e_d = d;
    h->e_M = code_size;
    h->has_codebooks = true;
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_upload_codebooks: host allocation failed");
}

// residual against table[rows[i]] -> [OPQ] -> codes -> decode -> [OPQ back] -> + table row -> squared norm -> norm
// code, for m vectors in dx (clobbered).  Leaves the bytes in e_codes / e_ncodes.  The table is the centroid
// table (add_batch) or a batch's sub-centroid table (add_group).
static int encode_rows(ivfhnsw_gpu *h, size_t m, float *dx, const float *table, const uint32_t *rows)
{
    const size_t d = h->e_d, M = h->e_M;
    float *res = h->e_res.as<float>(), *tmp = h->e_tmp.as<float>();
    HIP_TRY(launch_madd_rows(h->stream, dx, -1.f, table, rows, res, m, (int)d));
    const float *enc_in = res;
    if (h->e_opq) {
        HIP_TRY(launch_opq(h->stream, h->e_at.as<float>(), res, tmp, (int)m, (int)d));
        enc_in = tmp;
    }
    HIP_TRY(launch_pq_encode(h->stream, enc_in, h->e_pqc.as<float>(), h->e_codes.as<uint8_t>(), m, (int)d, (int)M));
    float *dec = h->e_opq ? res : tmp; // the buffer the encoder did not read
    HIP_TRY(launch_pq_decode(h->stream, h->e_codes.as<uint8_t>(), h->e_pqc.as<float>(), dec, m, (int)d, (int)M));
    float *back = dec;
    if (h->e_opq) {
        HIP_TRY(launch_opq(h->stream, h->e_a.as<float>(), dec, tmp, (int)m, (int)d));
        back = tmp;
    }
    HIP_TRY(launch_madd_rows(h->stream, back, 1.f, table, rows, dx, m, (int)d)); // x is spent: reuse
    HIP_TRY(launch_norm_codes(h->stream, dx, h->e_ntab.as<float>(), h->e_ncodes.as<uint8_t>(), nullptr, m, (int)d));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_encode(ivfhnsw_gpu *h, size_t n, const float *x, const uint32_t *precomputed_idx, size_t efSearch,
                       uint32_t *out_idx, uint8_t *out_codes, uint8_t *out_norm_codes)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_codebooks || !h->has_graph)
        return fail(IVFHNSW_ERR_STATE, "encode needs upload_codebooks and upload_quantizer");
    if ((size_t)h->gr.d != h->e_d)
        return fail(IVFHNSW_ERR_STATE, "code books are for d = %zu, the quantizer holds d = %d", h->e_d, h->gr.d);
    if (n == 0)
        return IVFHNSW_OK;
    if (!x || !out_codes || !out_norm_codes)
        return fail(IVFHNSW_ERR_INVALID, "null buffer");
    if (!precomputed_idx && efSearch == 0)
        return fail(IVFHNSW_ERR_INVALID, "efSearch 0 (assign runs searchKnn(x, 1))");
    const size_t d = h->e_d, M = h->e_M;
    const size_t kChunk = (size_t)1 << 18; // 128 MB of vectors at d = 128 per buffer
    for (size_t i0 = 0; i0 < n; i0 += kChunk) {
        const size_t m = std::min(kChunk, n - i0);
        if ((rc = h->e_x.ensure(m * d * sizeof(float))) || (rc = h->e_res.ensure(m * d * sizeof(float))) ||
            (rc = h->e_tmp.ensure(m * d * sizeof(float))) || (rc = h->e_idx.ensure(m * sizeof(uint32_t))) ||
            (rc = h->e_dist.ensure(m * sizeof(float))) || (rc = h->e_codes.ensure(m * M)) ||
            (rc = h->e_ncodes.ensure(m)))
            return rc;
        float *dx = h->e_x.as<float>();
        uint32_t *idx = h->e_idx.as<uint32_t>();
        HIP_TRY(hipMemcpyAsync(dx, x + i0 * d, m * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if (precomputed_idx) {
            for (size_t i = 0; i < m; i++)
                if (precomputed_idx[i0 + i] >= h->gr.n)
                    return fail(IVFHNSW_ERR_INVALID, "precomputed_idx[%zu] = %u out of range", i0 + i,
                                precomputed_idx[i0 + i]);
            HIP_TRY(hipMemcpyAsync(idx, precomputed_idx + i0, m * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        } else if ((rc = ivfhnsw_gpu_coarse_dev(h, m, dx, 1, efSearch, idx, h->e_dist.as<float>()))) {
            return rc;
        }
        if ((rc = encode_rows(h, m, dx, h->gr.vectors, idx)))
            return rc;
        HIP_TRY(hipMemcpyAsync(out_codes + i0 * M, h->e_codes.p, m * M, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(out_norm_codes + i0, h->e_ncodes.p, m, hipMemcpyDeviceToHost, h->stream));
        if (out_idx)
            HIP_TRY(hipMemcpyAsync(out_idx + i0, idx, m * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if ((rc = check_status(h)))
            return rc;
    }
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_encode_groups(ivfhnsw_gpu *h, size_t ngroups, size_t nsubc, const uint32_t *centroid_idx,
                              const uint64_t *offsets, const float *x, size_t efSearch, uint32_t *out_nn_centroid_idxs,
                              float *out_alphas, uint32_t *out_subcentroid_idxs, uint8_t *out_codes,
                              uint8_t *out_norm_codes)
try {
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_codebooks || !h->has_graph)
        return fail(IVFHNSW_ERR_STATE, "encode_groups needs upload_codebooks and upload_quantizer");
    if ((size_t)h->gr.d != h->e_d)
        return fail(IVFHNSW_ERR_STATE, "code books are for d = %zu, the quantizer holds d = %d", h->e_d, h->gr.d);
    if (ngroups == 0)
        return IVFHNSW_OK;
    if (!centroid_idx || !offsets || !out_nn_centroid_idxs || !out_alphas)
        return fail(IVFHNSW_ERR_INVALID, "null buffer");
    const size_t d = h->e_d, M = h->e_M, k = nsubc + 1;
    if (nsubc == 0 || nsubc > 4096 || efSearch < k || k > h->gr.n)
        return fail(IVFHNSW_ERR_INVALID, "nsubc %zu: need 1 <= nsubc, nsubc + 1 <= efSearch (%zu) and <= %u centroids",
                    nsubc, efSearch, h->gr.n);
    if (group_points_lds_bytes((int)nsubc, (int)d) > 160 * 1024)
        return fail(IVFHNSW_ERR_INVALID, "nsubc %zu x d %zu does not fit the 160 KB of LDS of one workgroup", nsubc, d);
    if (offsets[0] != 0)
        return fail(IVFHNSW_ERR_INVALID, "offsets[0] must be 0");
    const uint64_t n_total = offsets[ngroups];
    for (size_t g = 0; g < ngroups; g++) {
        if (offsets[g + 1] < offsets[g])
            return fail(IVFHNSW_ERR_INVALID, "offsets not monotone at group %zu", g);
        if (centroid_idx[g] >= h->gr.n)
            return fail(IVFHNSW_ERR_INVALID, "centroid_idx[%zu] = %u out of range", g, centroid_idx[g]);
    }
    if (n_total && (!x || !out_subcentroid_idxs || !out_codes || !out_norm_codes))
        return fail(IVFHNSW_ERR_INVALID, "null buffer");
    // chunks of whole groups: at most 2^18 points (one oversized group goes alone) and 4096 groups
    const size_t kMaxPoints = (size_t)1 << 18, kMaxGroups = 4096;
    std::vector<uint32_t> ids, nn;
    std::vector<float> dists, cvn;
    std::vector<unsigned long long> off;
    for (size_t g0 = 0; g0 < ngroups;) {
        size_t g1 = g0 + 1;
        while (g1 < ngroups && g1 - g0 < kMaxGroups && offsets[g1 + 1] - offsets[g0] <= kMaxPoints)
            g1++;
        const size_t G = g1 - g0, p0 = offsets[g0], m = offsets[g1] - p0;
        // neighbour centroids: searchKnn(centroid, nsubc + 1) for every group of the chunk (Grouping.cpp:47-62)
        if ((rc = h->cg_q.ensure(G * d * sizeof(float))) || (rc = h->cg_cidx.ensure(G * sizeof(uint32_t))) ||
            (rc = h->cg_ids.ensure(G * k * sizeof(uint32_t))) || (rc = h->cg_dists.ensure(G * k * sizeof(float))) ||
            (rc = h->gc_nn.ensure(G * nsubc * sizeof(uint32_t))) || (rc = h->cg_cvn.ensure(G * nsubc * sizeof(float))) ||
            (rc = h->cg_tab.ensure(G * nsubc * d * sizeof(float))) || (rc = h->cg_tab2.ensure(G * nsubc * d * sizeof(float))) ||
            (rc = h->cg_off.ensure((G + 1) * sizeof(unsigned long long))) || (rc = h->cg_alpha2.ensure(G * sizeof(float))))
            return rc;
        HIP_TRY(hipMemcpyAsync(h->cg_cidx.p, centroid_idx + g0, G * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(h->cg_q.p, 0, G * d * sizeof(float), h->stream));
        HIP_TRY(launch_madd_rows(h->stream, h->cg_q.as<float>(), 1.f, h->gr.vectors, h->cg_cidx.as<uint32_t>(),
                                 h->cg_q.as<float>(), G, (int)d)); // 0 + 1 * row: the centroid rows as queries
        if ((rc = ivfhnsw_gpu_coarse_dev(h, G, h->cg_q.as<float>(), k, efSearch, h->cg_ids.as<uint32_t>(),
                                         h->cg_dists.as<float>())))
            return rc;
        ids.resize(G * k);
        dists.resize(G * k);
        HIP_TRY(hipMemcpyAsync(ids.data(), h->cg_ids.p, G * k * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(dists.data(), h->cg_dists.p, G * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if ((rc = check_status(h)))
            return rc;
        nn.resize(G * nsubc);
        cvn.resize(G * nsubc);
        for (size_t g = 0; g < G; g++)
            for (size_t s = 0; s < nsubc; s++) { // the nearest one (the centroid itself) is dropped
                if (ids[g * k + s + 1] == 0xffffffffu)
                    return fail(IVFHNSW_ERR_INVALID, "group %zu: the walk found fewer than nsubc + 1 = %zu centroids "
                                                     "(the reference leaves zero entries behind here)", g0 + g, k);
                nn[g * nsubc + s] = ids[g * k + s + 1];
                cvn[g * nsubc + s] = dists[g * k + s + 1];
            }
        std::memcpy(out_nn_centroid_idxs + g0 * nsubc, nn.data(), G * nsubc * sizeof(uint32_t));
        if (m == 0) { // only empty groups: alpha stays what the caller has (Grouping.cpp:63-64)
            g0 = g1;
            continue;
        }
        off.resize(G + 1);
        for (size_t g = 0; g <= G; g++)
            off[g] = offsets[g0 + g] - p0;
        if ((rc = h->e_x.ensure(m * d * sizeof(float))) || (rc = h->e_res.ensure(m * d * sizeof(float))) ||
            (rc = h->e_tmp.ensure(m * d * sizeof(float))) || (rc = h->e_idx.ensure(m * sizeof(uint32_t))) ||
            (rc = h->e_dist.ensure(2 * m * sizeof(float))) || (rc = h->e_codes.ensure(m * M)) ||
            (rc = h->e_ncodes.ensure(m)) || (rc = h->cg_sub.ensure(m * sizeof(uint32_t))))
            return rc;
        float *dx = h->e_x.as<float>();
        HIP_TRY(hipMemcpyAsync(dx, x + p0 * d, m * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->gc_nn.p, nn.data(), G * nsubc * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->cg_cvn.p, cvn.data(), G * nsubc * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->cg_off.p, off.data(), (G + 1) * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
        const uint32_t *cidx = h->cg_cidx.as<uint32_t>();
        const unsigned long long *doff = h->cg_off.as<unsigned long long>();
        float *cv = h->cg_tab.as<float>(), *sub = h->cg_tab2.as<float>(), *num = h->e_dist.as<float>(), *den = num + m;
        HIP_TRY(launch_group_table(h->stream, 0, h->gr.vectors, cidx, (const uint32_t *)h->gc_nn.p, nullptr, nullptr, cv, G,
                                   (int)nsubc, (int)d));
        HIP_TRY(launch_group_points(h->stream, 0, h->gr.vectors, cidx, cv, h->cg_cvn.as<float>(), doff, dx, num, den,
                                    nullptr, G, (int)nsubc, (int)d));
        HIP_TRY(launch_group_alpha(h->stream, doff, num, den, h->cg_alpha2.as<float>(), G));
        HIP_TRY(launch_group_table(h->stream, 1, h->gr.vectors, cidx, nullptr, h->cg_alpha2.as<float>(), cv, sub, G,
                                   (int)nsubc, (int)d));
        HIP_TRY(launch_group_points(h->stream, 1, h->gr.vectors, cidx, sub, nullptr, doff, dx, nullptr, nullptr,
                                    h->cg_sub.as<uint32_t>(), G, (int)nsubc, (int)d));
        HIP_TRY(launch_group_rows(h->stream, doff, h->cg_sub.as<uint32_t>(), h->e_idx.as<uint32_t>(), G, (int)nsubc));
        if ((rc = encode_rows(h, m, dx, sub, h->e_idx.as<uint32_t>())))
            return rc;
        // alphas: only groups with points are written (an empty group keeps the caller's value)
        std::vector<float> al(G);
        HIP_TRY(hipMemcpyAsync(al.data(), h->cg_alpha2.p, G * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(out_subcentroid_idxs + p0, h->cg_sub.p, m * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(out_codes + p0 * M, h->e_codes.p, m * M, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(out_norm_codes + p0, h->e_ncodes.p, m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t g = 0; g < G; g++)
            if (off[g + 1] > off[g])
                out_alphas[g0 + g] = al[g];
        g0 = g1;
    }
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_encode_groups: host allocation failed");
}

int ivfhnsw_gpu_pq_train(ivfhnsw_gpu *h, size_t n, size_t d, size_t M, const float *x, size_t niter, float *centroids,
                         uint8_t *out_assign)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (n == 0 || d == 0 || M == 0 || d % M || d / M > 64 || !x || !centroids)
        return fail(IVFHNSW_ERR_INVALID, "bad training arguments (n %zu, d %zu, M %zu)", n, d, M);
    if ((rc = h->t_x.ensure(n * d * sizeof(float))) || (rc = h->t_cb.ensure(256 * d * sizeof(float))) ||
        (rc = h->t_assign.ensure(n * M)))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->t_x.p, x, n * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->t_cb.p, centroids, 256 * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    for (size_t it = 0; it < niter; it++) {
        // assignment = pq->compute_codes with the current code book; update = means in point order
        HIP_TRY(launch_pq_encode(h->stream, h->t_x.as<float>(), h->t_cb.as<float>(), h->t_assign.as<uint8_t>(), n, (int)d,
                                 (int)M));
        HIP_TRY(launch_lloyd_update(h->stream, h->t_x.as<float>(), h->t_assign.as<uint8_t>(), h->t_cb.as<float>(), n,
                                    (int)d, (int)M));
    }
    HIP_TRY(hipMemcpyAsync(centroids, h->t_cb.p, 256 * d * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (out_assign && niter)
        HIP_TRY(hipMemcpyAsync(out_assign, h->t_assign.p, n * M, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_xty(ivfhnsw_gpu *h, size_t n, size_t d, const float *X, const float *Y, float *C)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (n == 0 || d == 0 || d > 4096 || !X || !Y || !C)
        return fail(IVFHNSW_ERR_INVALID, "bad xty arguments (n %zu, d %zu)", n, d);
    const size_t nchunks = (n + kXtyChunk - 1) / kXtyChunk;
    if (nchunks > 65535)
        return fail(IVFHNSW_ERR_INVALID, "xty: more than %d points", 65535 * kXtyChunk);
    if ((rc = h->t_x.ensure(n * d * sizeof(float))) || (rc = h->t_y.ensure(n * d * sizeof(float))) ||
        (rc = h->t_part.ensure(nchunks * d * d * sizeof(float))) || (rc = h->t_c.ensure(d * d * sizeof(float))))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->t_x.p, X, n * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->t_y.p, Y, n * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(launch_xty(h->stream, h->t_x.as<float>(), h->t_y.as<float>(), h->t_part.as<float>(), h->t_c.as<float>(), n,
                       (int)d));
    HIP_TRY(hipMemcpyAsync(C, h->t_c.p, d * d * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_knn_dev(ivfhnsw_gpu *h, size_t nq, size_t nx, size_t d, const float *d_queries, const float *d_base,
                        size_t k, int mode, uint32_t *d_out_ids, float *d_out_dists)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (d < 4 || d > 128 || (d & 3))
        return fail(IVFHNSW_ERR_INVALID, "knn: d %zu must be a multiple of 4, at most 128", d);
    if (k == 0 || k > 80)
        return fail(IVFHNSW_ERR_INVALID, "knn: k %zu outside 1..80", k);
    if (nq > 0x7fffffffull || nx > 0xffffffffull)
        return fail(IVFHNSW_ERR_INVALID, "knn: too many rows");
    if (mode < IVFHNSW_KNN_ALL || mode > IVFHNSW_KNN_EARLIER)
        return fail(IVFHNSW_ERR_INVALID, "knn: unknown mode %d", mode);
    if (nq == 0)
        return IVFHNSW_OK;
    if (!d_queries || !d_base || !d_out_ids)
        return fail(IVFHNSW_ERR_INVALID, "knn: null buffer");
    // (the triangular table is one sweep per row block: its column range depends on the block)
    const int nsplit = mode == IVFHNSW_KNN_EARLIER ? 1 : knn_splits_for(nq, nx ? nx : 1);
    if ((rc = h->k_qn.ensure(nq * sizeof(float))) || (rc = h->k_xn.ensure((nx ? nx : 1) * sizeof(float))) ||
        (rc = h->k_part.ensure((size_t)nsplit * nq * k * sizeof(uint64_t))))
        return rc;
    float *dd = d_out_dists;
    if (!dd) {
        if ((rc = h->k_dists.ensure(nq * k * sizeof(float))))
            return rc;
        dd = h->k_dists.as<float>();
    }
    HIP_TRY(launch_knn_norms(h->stream, d_queries, h->k_qn.as<float>(), nq, (int)d));
    HIP_TRY(launch_knn_norms(h->stream, d_base, h->k_xn.as<float>(), nx, (int)d));
    HIP_TRY(launch_knn(h->stream, d_queries, d_base, h->k_qn.as<float>(), h->k_xn.as<float>(), nq, nx, (int)d, (int)k,
                       mode, nsplit, h->k_part.as<unsigned long long>(), d_out_ids, dd));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_knn(ivfhnsw_gpu *h, size_t nq, size_t nx, size_t d, const float *queries, const float *base, size_t k,
                    int mode, uint32_t *out_ids, float *out_dists)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!base || !out_ids)
        return fail(IVFHNSW_ERR_INVALID, "knn: null buffer");
    const bool self = queries == nullptr;
    if (self)
        nq = nx;
    if (nq == 0)
        return IVFHNSW_OK;
    if ((rc = upload(h->k_x, base, nx * d * sizeof(float))))
        return rc;
    if (!self && (rc = upload(h->k_q, queries, nq * d * sizeof(float))))
        return rc;
    if ((rc = h->k_ids.ensure(nq * k * sizeof(uint32_t))) || (rc = h->k_dists.ensure(nq * k * sizeof(float))))
        return rc;
    // (k_dists doubles as the output buffer here: knn_dev is handed it explicitly, so it does not allocate its own)
    if ((rc = ivfhnsw_gpu_knn_dev(h, nq, nx, d, self ? h->k_x.as<float>() : h->k_q.as<float>(), h->k_x.as<float>(), k,
                                  mode, h->k_ids.as<uint32_t>(), h->k_dists.as<float>())))
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out_ids, h->k_ids.p, nq * k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (out_dists)
        HIP_TRY(hipMemcpy(out_dists, h->k_dists.p, nq * k * sizeof(float), hipMemcpyDeviceToHost));
    return IVFHNSW_OK;
}

static int search_dev_chunk(ivfhnsw_gpu *h, size_t nq, size_t k, const float *d_queries, const uint32_t *d_coarse_ids,
                            const float *d_coarse_dists, const ivfhnsw_search_params *p, float *d_distances,
                            int64_t *d_labels, int64_t *d_out_keys);

// Batches beyond kMaxBatch queries are processed in slices so that the per-batch workspace (16 KB of table per
// query at PQ16, plus the plan) stays bounded; the multi-GPU resolve step needs the whole plan, so it is limited
// to one slice.
static const size_t kMaxBatchAll = 1 << 17;

// One large batch as TWO uneven parts on two streams.  The walk's resident wavefronts pull queries from a counter, so a
// launch ends with a tail of partly idle CUs (10 000 queries on 4096 slots: 2.44 "rounds"), and the scan can only start
// when the last query has been walked.  Here the first ~78 % of the batch run on the handle's stream and the rest on an
// internal view (own workspace and stream, same tables): the second part's walk moves into the slots the first part's
// last round frees, and the first part's table + scan (LDS-bound) run beside it (HBM-bound).  Fork and join are events,
// so the call keeps its contract: everything is ordered behind the caller's stream and complete when that stream gets
// there.  Measured (tools/split_probe.py, 1B corpus, 10 k queries): 1.81 -> 1.67 ms per batch, the first part's scan at
// 4.8 instead of 5.0 TB/s; three parts give no more, four lose.  ON by default since round 3 (the first part's share by
// estimate, auto_split_permille; ivfhnsw_gpu_set_batch_split(h, 0) or IVFHNSW_SPLIT=0 = one part): a plain search_dev call should deliver the
// fastest exact form.  The price is in the scan's accounting: two launches per step, the first slowed a little by the walk
// beside it, the second a single round of workgroups -- 17 B x all codes over the summed launch time is 0.58-0.59 of the
// HBM peak where the one-launch form reads 0.62 (bench.py reports both).  Not for sharded calls (their resolve step
// needs one plan), heap-order k > 1 (one candidate stream), given coarse results (no walk to overlap) or batches below
// two rounds of the walk.
static const size_t kSplitMinNq = 8192;

int ivfhnsw_gpu_set_batch_split(ivfhnsw_gpu *h, int permille)
{
    if (!h)
        return fail(IVFHNSW_ERR_INVALID, "null handle");
    if (permille < 0 || permille > kSplitAuto)
        return fail(IVFHNSW_ERR_INVALID, "batch split %d outside 0..999 permille (1000 = by estimate)", permille);
    if (h->is_view && permille)
        return fail(IVFHNSW_ERR_INVALID, "a view cannot split its batches (it is what the second part runs on)");
    h->split_pm = permille;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_set_option(ivfhnsw_gpu *h, const char *key, long value)
{
    if (!h || !key)
        return fail(IVFHNSW_ERR_INVALID, "set_option: null argument");
    if (!strcmp(key, "scan_pipe")) {
        if (value < -1 || value > 1)
            return fail(IVFHNSW_ERR_INVALID, "set_option scan_pipe: %ld outside -1..1", value);
        h->opt_scan_pipe = (int)value;
        return IVFHNSW_OK;
    }
    return fail(IVFHNSW_ERR_INVALID, "set_option: unknown key '%s'", key);
}

// The first part's share when nobody fixed it.  The second part's walk fills the tail of the first part's, the first part's
// table + plan + scan run beside the second part's walk, the second part's scan runs alone: the step is shortest where the
// second walk and the first scan take equally long, share = W / (W + S).  W and S per query from the call's own parameters,
// with rates measured at the 1B shapes on one MI355X: the walk 1.7 ns per unit of efSearch (1.37 / 1.63 /
// 2.56 ms per 10 k queries at 80 / 100 / 130), the scan (M + 1) bytes per code at 5 TB/s over the codes the max_codes rule
// lets through, the table 5 ns, the Grouping plan ~90 ns and its scan at 0.8 of the rate.  Measured against fixed shares:
// (32, 10000, 80) 0.79 -> the 2048-query second part that measured best; (64, 30000, 100) 0.63 -> 4096, 3.97 -> 4.18 M
// queries/s; Grouping + pruning 0.49 -> 4096, 3.35 -> 3.44 M.  The estimate only has to land on the right multiple of 2048.
static int auto_split_permille(const ivfhnsw_gpu *h, const ivfhnsw_search_params *p)
{
    const double walk = 1.7 * (double)p->efSearch;
    const double nc = (double)std::max<uint32_t>(h->t.nc, 1u);
    const double list = (double)h->n_local / nc;
    const double codes = std::min((double)p->nprobe * list, (double)p->max_codes + 0.5 * list);
    double scan = codes * (double)(h->t.M + 1) / 5000.0 + 5.0;
    if (h->has_group)
        scan = scan / 0.8 + 90.0;
    const double share = walk / (walk + scan);
    return (int)std::min(900.0, std::max(400.0, share * 1000.0 + 0.5));
}

static int search_dev_split(ivfhnsw_gpu *h, size_t nq, size_t k, const float *d_queries, const ivfhnsw_search_params *p,
                            float *d_distances, int64_t *d_labels)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->split_view) {
        if ((rc = ivfhnsw_gpu_create_view(h, &h->split_view)))
            return rc;
        HIP_TRY(hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->split_join, hipEventDisableTiming));
    }
    ivfhnsw_gpu *v = h->split_view;
    // the view follows the handle's tables (uploads since its creation included)
    v->t = h->t;
    v->has_ivf = h->has_ivf;
    v->n_local = h->n_local;
    v->g = h->g;
    v->has_group = h->has_group;
    v->gr = h->gr;
    v->has_graph = h->has_graph;
    v->profiling = h->profiling;
    v->status_shared = h->w_status.as<uint32_t>();
    // the second part: ~22 % of the batch, in whole "rounds" of the scan's resident workgroups (8 per CU x 256 CUs): its
    // scan runs alone at the end of the step, and 2200 workgroups on 2048 slots would take two rounds for one
    const size_t round_wgs = 2048;
    const int pm = h->split_pm == kSplitAuto ? auto_split_permille(h, p) : h->split_pm;
    size_t n2 = ((nq * (size_t)(1000 - pm) / 1000 + round_wgs / 2) / round_wgs) * round_wgs;
    n2 = std::max(round_wgs, std::min(n2, nq / 2));
    const size_t n1 = nq - n2;
    const size_t d = (size_t)h->t.d;
    HIP_TRY(hipEventRecord(h->split_fork, h->stream));
    HIP_TRY(hipStreamWaitEvent(v->stream, h->split_fork, 0));
    rc = search_dev_chunk(h, n1, k, d_queries, nullptr, nullptr, p, d_distances, d_labels, nullptr);
    int rc2 = rc ? rc
                 : search_dev_chunk(v, nq - n1, k, d_queries + n1 * d, nullptr, nullptr, p, d_distances + n1 * k,
                                    d_labels + n1 * k, nullptr);
    // (whatever the second part flags it raises in the handle's own status word: v->status_shared)
    // the join itself, always (the fork was recorded)
    (void)hipSetDevice(h->device);
    HIP_TRY(hipEventRecord(h->split_join, v->stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->split_join, 0));
    h->last_split = rc2 == 0;
    h->last_parts[0] = n1;
    h->last_parts[1] = n2;
    return rc2;
}

static int search_dev_part(ivfhnsw_gpu *h, size_t nq, size_t k, const float *d_queries, const uint32_t *d_coarse_ids,
                           const float *d_coarse_dists, const ivfhnsw_search_params *p, float *d_distances,
                           int64_t *d_labels, int64_t *d_out_keys)
{
    if (h) {
        h->last_split = false;
        h->last_parts[0] = nq;
        h->last_parts[1] = 0;
    }
    const bool split = h && p && !h->is_view && h->split_pm > 0 && nq >= kSplitMinNq && !d_coarse_ids &&
                       !d_out_keys && !(p->heap_order && k > 1) && h->has_ivf && h->has_graph && p->nprobe > 0 && k > 0 &&
                       k <= 1024 && d_queries && d_distances && d_labels;
    if (split)
        return search_dev_split(h, nq, k, d_queries, p, d_distances, d_labels);
    return search_dev_chunk(h, nq, k, d_queries, d_coarse_ids, d_coarse_dists, p, d_distances, d_labels, d_out_keys);
}

int ivfhnsw_gpu_search_dev(ivfhnsw_gpu *h, size_t nq, size_t k, const float *d_queries, const uint32_t *d_coarse_ids,
                           const float *d_coarse_dists, const ivfhnsw_search_params *p, float *d_distances,
                           int64_t *d_labels, int64_t *d_out_keys)
{
    const size_t kMaxBatch = (p && p->heap_order && k > 1) ? kMaxBatchAll / 8 : kMaxBatchAll;
    if (nq <= kMaxBatch || !h || !p)
        return search_dev_part(h, nq, k, d_queries, d_coarse_ids, d_coarse_dists, p, d_distances, d_labels, d_out_keys);
    if (d_out_keys)
        return fail(IVFHNSW_ERR_INVALID, "sharded search (out_keys) is limited to %zu queries per call", kMaxBatch);
    const size_t d = (size_t)h->t.d;
    for (size_t q0 = 0; q0 < nq; q0 += kMaxBatch) {
        const size_t n = std::min(kMaxBatch, nq - q0);
        int rc = search_dev_part(h, n, k, d_queries + q0 * d, d_coarse_ids ? d_coarse_ids + q0 * p->nprobe : nullptr,
                                 d_coarse_dists ? d_coarse_dists + q0 * p->nprobe : nullptr, p, d_distances + q0 * k,
                                 d_labels + q0 * k, nullptr);
        if (rc)
            return rc;
    }
    return IVFHNSW_OK;
}

static int search_dev_chunk(ivfhnsw_gpu *h, size_t nq, size_t k, const float *d_queries, const uint32_t *d_coarse_ids,
                            const float *d_coarse_dists, const ivfhnsw_search_params *p, float *d_distances,
                            int64_t *d_labels, int64_t *d_out_keys)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf)
        return fail(IVFHNSW_ERR_STATE, "search before upload_ivf");
    if (!p || p->nprobe == 0 || k == 0)
        return fail(IVFHNSW_ERR_INVALID, "nprobe and k must be positive");
    if ((d_coarse_ids == nullptr) != (d_coarse_dists == nullptr))
        return fail(IVFHNSW_ERR_INVALID, "coarse_ids and coarse_dists must both be given or both be NULL");
    if (nq > 0 && (!d_queries || !d_distances || !d_labels))
        return fail(IVFHNSW_ERR_INVALID, "null query/result buffer");
    if (k > 1024)
        return fail(IVFHNSW_ERR_INVALID, "k %zu > 1024 unsupported", k);
    if (nq > 0x7fffffffull / (k > p->nprobe ? k : p->nprobe))
        return fail(IVFHNSW_ERR_INVALID, "nq too large");
    if (h->has_group && !h->has_graph)
        return fail(IVFHNSW_ERR_STATE, "Grouping search needs upload_quantizer (sub-centroid distances)");
    if (h->has_graph && (h->gr.d != h->t.d || h->gr.n != h->t.nc))
        return fail(IVFHNSW_ERR_STATE, "quantizer (%u x %d) does not match the index (%u x %d)", h->gr.n, h->gr.d,
                    h->t.nc, h->t.d);
    h->last_nq = 0;
    if (nq == 0)
        return IVFHNSW_OK;

    const int d = h->t.d, M = h->t.M, nprobe = (int)p->nprobe;
    const int max_seg = h->has_group ? nprobe * h->g.nsubc : nprobe;
    if ((rc = h->w_segs.ensure(nq * (size_t)max_seg * sizeof(Seg))))
        return rc;
    if ((rc = h->w_lpos.ensure(nq * (size_t)max_seg * sizeof(uint32_t))))
        return rc;
    if ((rc = h->w_hdr.ensure(nq * sizeof(PlanHdr))))
        return rc;
    if ((rc = h->w_keys.ensure(nq * k * sizeof(uint64_t))))
        return rc;
    if ((rc = h->w_totals.ensure(2 * sizeof(unsigned long long))))
        return rc;

    // 1. rotate (IndexIVF_HNSW.cpp:240)
    const float *xq = d_queries;
    if (h->t.opq_At) {
        if ((rc = h->w_xq.ensure(nq * (size_t)d * sizeof(float))))
            return rc;
        StageScope sc(h, IVFHNSW_STAGE_OPQ);
        HIP_TRY(launch_opq(h->stream, h->t.opq_At, d_queries, h->w_xq.as<float>(), (int)nq, d));
        xq = h->w_xq.as<float>();
    }
    // small IVFADC batches: everything behind the coarse stage in one launch (kernels_tail.hip); its per-query
    // meeting words are cleared by the latency walk when that runs, by a memset otherwise
    static const size_t tail_max_nq = [] {
        const char *e = getenv("IVFHNSW_TAIL_MAX_NQ");
        return (e && *e) ? (size_t)atol(e) : (size_t)8;
    }();
    const bool use_tail =
        !h->has_group && nq <= tail_max_nq && !d_out_keys && ivf_tail_supported(h->t, nprobe, (int)k);
    const size_t tail_kbytes = nq * sizeof(uint64_t);
    h->walk_zeroed = false;
    h->walk_zero_keys = nullptr;
    h->walk_zero_done = nullptr;
    if (use_tail) {
        if ((rc = h->w_tail.ensure(tail_kbytes + nq * sizeof(uint32_t))))
            return rc;
        h->walk_zero_keys = h->w_tail.as<uint64_t>();
        h->walk_zero_done = reinterpret_cast<uint32_t *>(h->w_tail.as<char>() + tail_kbytes);
    }
    // 2. coarse (IndexIVF_HNSW.cpp:248-259)
    const uint32_t *cid = d_coarse_ids;
    const float *cd = d_coarse_dists;
    if (!cid) {
        if ((rc = h->w_cid.ensure(nq * (size_t)nprobe * sizeof(uint32_t))))
            return rc;
        if ((rc = h->w_cd.ensure(nq * (size_t)nprobe * sizeof(float))))
            return rc;
        if ((rc = ivfhnsw_gpu_coarse_dev(h, nq, xq, p->nprobe, p->efSearch, h->w_cid.as<uint32_t>(),
                                         h->w_cd.as<float>())))
            return rc;
        cid = h->w_cid.as<uint32_t>();
        cd = h->w_cd.as<float>();
    }
    h->tail_wrote_status = false;
    h->walk_zero_keys = nullptr;
    h->walk_zero_done = nullptr;
    if (use_tail) {
        const int nsplit = (int)std::min<size_t>(32, (2048 + nq - 1) / nq);
        StageScope sc(h, IVFHNSW_STAGE_SCAN);
        if (!h->walk_zeroed)
            HIP_TRY(hipMemsetAsync(h->w_tail.p, 0, tail_kbytes + nq * sizeof(uint32_t), h->stream));
        HIP_TRY(launch_ivf_tail(h->stream, h->t, xq, cid, cd, (int)nq, nprobe, p->max_codes, nsplit,
                                h->w_tail.as<uint64_t>(), reinterpret_cast<uint32_t *>(h->w_tail.as<char>() + tail_kbytes),
                                h->w_hdr.as<PlanHdr>(), d_distances, d_labels, status_word(h),
                                h->tail_status_out));
        h->tail_wrote_status = h->tail_status_out != nullptr;
        h->last_scan_kernel = "ivf_tail_kernel";
        h->last_nq = (int)nq;
        h->last_max_seg = max_seg;
        h->last_stream = false;
        return IVFHNSW_OK;
    }
    // a plan segment is a list (IVFADC) or a sub-group (Grouping): the mean length decides the scan form
    const uint64_t nseg_all = (uint64_t)h->t.nc * (h->has_group ? (uint64_t)h->g.nsubc : 1);
    const int seg_hint = (int)std::min<uint64_t>(1u << 20, nseg_all ? (h->n_local * h->t.shard_world) / nseg_all : 0);
    const bool heap = p->heap_order && k > 1;
    // small batches: split each query over several workgroups so the chip still fills
    int nsplit = 1;
    if (k == 1 && nq < 1024)
        nsplit = (int)std::min<size_t>(32, (2048 + nq - 1) / nq);
    // list shards: table and scan in one software-pipelined kernel, the table never leaves the chip (kernels_scan3.hip)
    const bool pipe = k == 1 && !h->has_group && !heap && h->opt_scan_pipe != 0 &&
                      scan_pipe_supported(h->t, max_seg, (int)nq, nsplit, h->n_local > 0, h->opt_scan_pipe == 1);
    // one GPU, IVFADC: plan and tables are independent of each other and go in ONE launch (kernels_search.hip
    // plan_lut_kernel; IVFHNSW_PLAN_LUT=0 keeps them apart)
    static const bool plan_lut_on = [] {
        const char *e = getenv("IVFHNSW_PLAN_LUT");
        return !(e && *e && atoi(e) == 0);
    }();
    const int ds = h->t.dsub;
    const bool plan_lut = plan_lut_on && !h->has_group && !pipe && h->t.shard_world == 1 &&
                          (ds == 4 || ds == 6 || ds == 8 || ds == 12 || ds == 16);
    if (!pipe && (rc = h->w_luts.ensure(nq * (size_t)M * 256 * sizeof(float))))
        return rc;
    // 3. plan (IndexIVF_HNSW.cpp:267-292 / IndexIVF_HNSW_Grouping.cpp:222-353)
    if (plan_lut) {
        StageScope sc(h, IVFHNSW_STAGE_LUT); // plan + tables: one kernel, accounted as the table stage
        HIP_TRY(launch_plan_lut(h->stream, h->t, xq, cid, cd, (int)nq, nprobe, p->max_codes, h->w_segs.as<Seg>(),
                                h->w_lpos.as<uint32_t>(), h->w_hdr.as<PlanHdr>(), max_seg, h->w_keys.as<uint64_t>(),
                                (int)k, h->w_luts.as<float>()));
    } else {
        StageScope sc(h, IVFHNSW_STAGE_PLAN);
        if (h->has_group) {
            // per query: pass-1 values and sub-centroid distances of every (row, sub-group) the plan touches
            if ((rc = h->w_qsd.ensure(nq * (size_t)max_seg * 2 * sizeof(float))))
                return rc;
            HIP_TRY(launch_plan_grouping(h->stream, h->t, h->g, h->gr, xq, cid, cd, (int)nq, nprobe, p->max_codes,
                                         p->do_pruning, h->w_segs.as<Seg>(), h->w_lpos.as<uint32_t>(),
                                         h->w_hdr.as<PlanHdr>(), max_seg, h->w_keys.as<uint64_t>(), (int)k,
                                         h->w_qsd.as<float>()));
        } else {
            HIP_TRY(launch_plan_ivf(h->stream, h->t, cid, cd, (int)nq, nprobe, p->max_codes, h->w_segs.as<Seg>(),
                                    h->w_lpos.as<uint32_t>(), h->w_hdr.as<PlanHdr>(), max_seg,
                                    h->w_keys.as<uint64_t>(), (int)k));
        }
    }
    // 4. table (IndexIVF_HNSW.cpp:262)
    if (!pipe && !plan_lut) {
        StageScope sc(h, IVFHNSW_STAGE_LUT);
        HIP_TRY(launch_lut(h->stream, h->t, xq, h->w_luts.as<float>(), (int)nq, h->w_hdr.as<PlanHdr>()));
    }
    // 5. scan (IndexIVF_HNSW.cpp:282-289)
    bool scan_selected = false;
    if (heap) {
        // with out_keys (sharded search) the replay is the caller's: it merges the shards' candidate streams in scan
        // order first (ivfhnsw_gpu_last_stream_dev, ivfhnsw_gpu_replay_stream_dev)
        if ((rc = h->w_stream.ensure(nq * (size_t)kHeapStreamCap * sizeof(uint64_t))))
            return rc;
        if ((rc = h->w_slen.ensure(nq * sizeof(uint32_t))))
            return rc;
    }
    {
        StageScope sc(h, IVFHNSW_STAGE_SCAN);
        if (pipe) {
            HIP_TRY(launch_scan_pipe(h->stream, h->t, xq, h->w_segs.as<Seg>(), h->w_lpos.as<uint32_t>(),
                                     h->w_hdr.as<PlanHdr>(), max_seg, (int)nq, h->w_keys.as<uint64_t>()));
            h->last_scan_kernel = "scan_pipe_kernel";
        } else {
            // k = 1 without out_keys: the scan writes distance and label itself where it can (no select launch)
            const bool want_sel = k == 1 && !d_out_keys && !heap;
            HIP_TRY(launch_scan(h->stream, h->t, h->w_luts.as<float>(), h->w_segs.as<Seg>(), h->w_lpos.as<uint32_t>(),
                                h->w_hdr.as<PlanHdr>(), max_seg, (int)nq, (int)k, nsplit, h->w_keys.as<uint64_t>(),
                                heap ? h->w_stream.as<uint64_t>() : nullptr, heap ? h->w_slen.as<uint32_t>() : nullptr,
                                heap ? kHeapStreamCap : 0, seg_hint, want_sel ? d_distances : nullptr,
                                want_sel ? d_labels : nullptr, &scan_selected));
            h->last_scan_kernel = last_scan_kernel_name();
        }
    }
    // 6. select
    if (!scan_selected) {
        StageScope sc(h, IVFHNSW_STAGE_SELECT);
        if (heap && !d_out_keys)
            HIP_TRY(launch_heap_replay(h->stream, h->t, h->w_segs.as<Seg>(), h->w_hdr.as<PlanHdr>(), max_seg,
                                       h->w_stream.as<uint64_t>(), h->w_slen.as<uint32_t>(), kHeapStreamCap, (int)nq,
                                       (int)k, d_distances, d_labels, status_word(h), nullptr));
        else
            HIP_TRY(launch_select(h->stream, h->t, h->w_segs.as<Seg>(), h->w_hdr.as<PlanHdr>(), max_seg,
                                  h->w_keys.as<uint64_t>(), (int)nq, (int)k, d_distances, d_labels, d_out_keys));
    }
    h->last_nq = (int)nq;
    h->last_max_seg = max_seg;
    h->last_stream = heap;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_resolve_keys_dev(ivfhnsw_gpu *h, size_t nq, size_t k, const int64_t *d_keys, float *d_distances,
                                 int64_t *d_labels)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf || h->last_nq == 0 || (size_t)h->last_nq != nq)
        return fail(IVFHNSW_ERR_STATE, "resolve_keys needs the plan of a preceding search_dev with the same nq");
    StageScope sc(h, IVFHNSW_STAGE_SELECT);
    HIP_TRY(launch_resolve(h->stream, h->t, h->w_segs.as<Seg>(), h->w_hdr.as<PlanHdr>(), h->last_max_seg, d_keys,
                           (int)nq, (int)k, d_distances, d_labels));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_last_stream_dev(ivfhnsw_gpu *h, size_t nq, size_t len_cap, uint64_t *d_keys, uint32_t *d_len,
                                uint32_t *stream_cap)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->last_nq || !h->last_stream || (size_t)h->last_nq != nq)
        return fail(IVFHNSW_ERR_STATE, "last_stream needs a preceding search_dev of the same nq with k > 1 and heap_order = 1");
    if (stream_cap)
        *stream_cap = kHeapStreamCap;
    if (d_len)
        HIP_TRY(hipMemcpyAsync(d_len, h->w_slen.p, nq * sizeof(uint32_t), hipMemcpyDeviceToDevice, h->stream));
    if (d_keys) {
        if (len_cap == 0 || len_cap > kHeapStreamCap)
            return fail(IVFHNSW_ERR_INVALID, "len_cap %zu outside 1..%u", len_cap, kHeapStreamCap);
        HIP_TRY(hipMemcpy2DAsync(d_keys, len_cap * sizeof(uint64_t), h->w_stream.p, (size_t)kHeapStreamCap * sizeof(uint64_t),
                                 len_cap * sizeof(uint64_t), nq, hipMemcpyDeviceToDevice, h->stream));
    }
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_replay_stream_dev(ivfhnsw_gpu *h, size_t nq, size_t k, const uint64_t *d_stream, const uint32_t *d_len,
                                  uint32_t cap, int64_t *d_out_keys)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf)
        return fail(IVFHNSW_ERR_STATE, "replay_stream before upload_ivf");
    if (nq == 0)
        return IVFHNSW_OK;
    if (!d_stream || !d_len || !d_out_keys || k == 0 || k > 1024 || nq > 0x7fffffffull / k)
        return fail(IVFHNSW_ERR_INVALID, "bad replay_stream arguments (k %zu)", k);
    StageScope sc(h, IVFHNSW_STAGE_SELECT);
    HIP_TRY(launch_heap_replay(h->stream, h->t, nullptr, nullptr, 0, d_stream, d_len, cap, (int)nq, (int)k, nullptr,
                               nullptr, status_word(h), d_out_keys));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_search(ivfhnsw_gpu *h, size_t nq, size_t k, const float *queries, const uint32_t *coarse_ids,
                       const float *coarse_dists, const ivfhnsw_search_params *p, float *distances, int64_t *labels)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf)
        return fail(IVFHNSW_ERR_STATE, "search before upload_ivf");
    if (!p || p->nprobe == 0 || k == 0)
        return fail(IVFHNSW_ERR_INVALID, "nprobe and k must be positive");
    if (nq == 0)
        return IVFHNSW_OK;
    if (!queries || !distances || !labels)
        return fail(IVFHNSW_ERR_INVALID, "null query/result buffer");
    if ((coarse_ids == nullptr) != (coarse_dists == nullptr))
        return fail(IVFHNSW_ERR_INVALID, "coarse_ids and coarse_dists must both be given or both be NULL");
    const size_t d = h->t.d;
    // Small batches -- the reference's drivers pass ONE query per call (tests/test_ivfhnsw_sift1b.cpp:193-208) -- go
    // through pinned host memory the kernels read and write directly: no staging copies, one synchronisation.  Layout
    // of the two blocks: in = queries | coarse ids | coarse dists; out = distances | labels | status word.
    static const size_t pinned_max_nq = [] {
        const char *e = getenv("IVFHNSW_PINNED_MAX_NQ");
        return (e && *e) ? (size_t)atol(e) : (size_t)256;
    }();
    if (nq <= pinned_max_nq) {
        const size_t np = p->nprobe;
        const size_t in_q = nq * d * sizeof(float), in_c = coarse_ids ? nq * np * sizeof(uint32_t) : 0;
        const size_t out_d = (nq * k * sizeof(float) + 7) & ~(size_t)7, out_l = nq * k * sizeof(int64_t);
        if ((rc = h->p_in.ensure(in_q + 2 * in_c)) || (rc = h->p_out.ensure(out_d + out_l + 8)))
            return rc;
        char *pin = h->p_in.as<char>(), *pout = h->p_out.as<char>();
        memcpy(pin, queries, in_q);
        if (coarse_ids) {
            memcpy(pin + in_q, coarse_ids, in_c);
            memcpy(pin + in_q + in_c, coarse_dists, in_c);
        }
        uint32_t *pst = reinterpret_cast<uint32_t *>(pout + out_d + out_l);
        // The latency walk keeps at most 64 exact ties at the efSearch boundary.  This call synchronises anyway, so instead
        // of a redo launch behind every one-query call the walk only raises a status bit, and the call repeats itself once
        // on the throughput walk, whose redo form has no such limit (hnswalg.cpp:67-68,93 has none either).
        for (int attempt = 0; attempt < 2; attempt++) {
            h->tail_status_out = pst;
            h->lat_defer_redo = attempt == 0;
            h->latency_off = attempt == 1;
            rc = ivfhnsw_gpu_search_dev(h, nq, k, reinterpret_cast<const float *>(pin),
                                        coarse_ids ? reinterpret_cast<const uint32_t *>(pin + in_q) : nullptr,
                                        coarse_ids ? reinterpret_cast<const float *>(pin + in_q + in_c) : nullptr, p,
                                        reinterpret_cast<float *>(pout), reinterpret_cast<int64_t *>(pout + out_d), nullptr);
            h->tail_status_out = nullptr;
            h->lat_defer_redo = false;
            h->latency_off = false;
            if (rc)
                return rc;
            if (!h->tail_wrote_status)
                HIP_TRY(hipMemcpyAsync(pst, h->w_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (attempt == 0 && (*pst & kStatusHnswTieOverflow)) {
                *pst &= ~kStatusHnswTieOverflow;
                HIP_TRY(hipMemcpy(h->w_status.p, pst, sizeof(uint32_t), hipMemcpyHostToDevice)); // bit consumed
                continue;
            }
            break;
        }
        memcpy(distances, pout, nq * k * sizeof(float));
        memcpy(labels, pout + out_d, out_l);
        return *pst ? check_status(h) : IVFHNSW_OK;
    }
    if ((rc = h->s_q.ensure(nq * d * sizeof(float))))
        return rc;
    if ((rc = h->s_dist.ensure(nq * k * sizeof(float))))
        return rc;
    if ((rc = h->s_lab.ensure(nq * k * sizeof(int64_t))))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->s_q.p, queries, nq * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (coarse_ids) {
        if ((rc = h->s_cid.ensure(nq * p->nprobe * sizeof(uint32_t))))
            return rc;
        if ((rc = h->s_cd.ensure(nq * p->nprobe * sizeof(float))))
            return rc;
        HIP_TRY(hipMemcpyAsync(h->s_cid.p, coarse_ids, nq * p->nprobe * sizeof(uint32_t), hipMemcpyHostToDevice,
                               h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_cd.p, coarse_dists, nq * p->nprobe * sizeof(float), hipMemcpyHostToDevice,
                               h->stream));
    }
    rc = ivfhnsw_gpu_search_dev(h, nq, k, h->s_q.as<float>(), coarse_ids ? h->s_cid.as<uint32_t>() : nullptr,
                                coarse_ids ? h->s_cd.as<float>() : nullptr, p, h->s_dist.as<float>(),
                                h->s_lab.as<int64_t>(), nullptr);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(distances, h->s_dist.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(labels, h->s_lab.p, nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return check_status(h);
}

int ivfhnsw_gpu_search_keys(ivfhnsw_gpu *h, size_t nq, size_t k, const float *queries, const uint32_t *coarse_ids,
                            const float *coarse_dists, const ivfhnsw_search_params *p, int64_t *keys)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (!h->has_ivf)
        return fail(IVFHNSW_ERR_STATE, "search before upload_ivf");
    if (!p || p->nprobe == 0 || k == 0)
        return fail(IVFHNSW_ERR_INVALID, "nprobe and k must be positive");
    if (nq == 0)
        return IVFHNSW_OK;
    if (!queries || !keys || !coarse_ids || !coarse_dists)
        return fail(IVFHNSW_ERR_INVALID, "null buffer (a shard is searched with the coarse stage supplied)");
    const size_t d = h->t.d, np = p->nprobe;
    if ((rc = h->s_q.ensure(nq * d * sizeof(float))) || (rc = h->s_dist.ensure(nq * k * sizeof(float))) ||
        (rc = h->s_lab.ensure(nq * k * sizeof(int64_t))) || (rc = h->s_keys.ensure(nq * k * sizeof(int64_t))) ||
        (rc = h->s_cid.ensure(nq * np * sizeof(uint32_t))) || (rc = h->s_cd.ensure(nq * np * sizeof(float))))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->s_q.p, queries, nq * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_cid.p, coarse_ids, nq * np * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_cd.p, coarse_dists, nq * np * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if ((rc = ivfhnsw_gpu_search_dev(h, nq, k, h->s_q.as<float>(), h->s_cid.as<uint32_t>(), h->s_cd.as<float>(), p,
                                     h->s_dist.as<float>(), h->s_lab.as<int64_t>(), h->s_keys.as<int64_t>())))
        return rc;
    HIP_TRY(hipMemcpyAsync(keys, h->s_keys.p, nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return check_status(h);
}

// ---------------------------------------------------------------------------------------------------------------------
// The shard step for a caller that holds all N shard handles in ONE process (the bundled classes with IVFHNSW_SHARDS=N):
// every shard scans its lists for the whole batch, the packed keys are MIN-merged ACROSS THE DEVICES over RCCL (xGMI) and
// the owner's labels MAX-merged, without the keys ever visiting the host.  RCCL is loaded on first use (dlopen: the
// library has no link-time dependency on it) and one communicator per device list is kept for the life of the process
// (ncclCommInitAll).  Shards that share a device -- a one-GPU box -- cannot form a communicator (RCCL refuses two ranks
// on one device): the same step then merges on the host, which is what the classes did before round 3.
// ---------------------------------------------------------------------------------------------------------------------
extern "C++" {
namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib)
                break;
        }
        if (x.lib) {
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.lib, "ncclCommInitAll"));
            x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.lib, "ncclAllReduce"));
            x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.lib, "ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.lib, "ncclGroupEnd"));
            x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.lib, "ncclGetErrorString"));
            x.ok = x.CommInitAll && x.AllReduce && x.GroupStart && x.GroupEnd;
        }
        return x;
    }();
    return r;
}

// one communicator set per device list, created once (ncclCommInitAll is collective over the devices of this process)
std::vector<ncclComm_t> *rccl_comms(const std::vector<int> &devs)
{
    static std::map<std::vector<int>, std::vector<ncclComm_t>> cache;
    static std::mutex mu;   // handles are one-thread-at-a-time objects, this table is process-wide
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(devs);
    if (it != cache.end())
        return it->second.empty() ? nullptr : &it->second;
    std::vector<ncclComm_t> comms(devs.size(), nullptr);
    Rccl &r = rccl();
    if (!r.ok || r.CommInitAll(comms.data(), (int)devs.size(), devs.data()) != ncclSuccess)
        comms.clear();
    auto &slot = cache[devs] = comms;
    return slot.empty() ? nullptr : &slot;
}

} // namespace
} // extern "C++"

int ivfhnsw_gpu_search_sharded(ivfhnsw_gpu *const *shards, size_t nshards, size_t nq, size_t k, const float *queries,
                               const uint32_t *coarse_ids, const float *coarse_dists, const ivfhnsw_search_params *p,
                               float *distances, int64_t *labels)
try {
    if (!shards || nshards == 0 || nshards > 64)
        return fail(IVFHNSW_ERR_INVALID, "search_sharded: 1..64 shard handles");
    for (size_t r = 0; r < nshards; r++)
        if (!shards[r] || !shards[r]->has_ivf)
            return fail(IVFHNSW_ERR_STATE, "search_sharded: shard %zu has no index", r);
    if (!p || p->nprobe == 0 || k == 0)
        return fail(IVFHNSW_ERR_INVALID, "nprobe and k must be positive");
    if (p->heap_order && k > 1)
        return fail(IVFHNSW_ERR_INVALID, "search_sharded merges keys (k = 1, or k > 1 ascending); the heap-array order of "
                                         "k > 1 needs the candidate streams (ivfhnsw_gpu_last_stream)");
    if (nq == 0)
        return IVFHNSW_OK;
    if (!queries || !coarse_ids || !coarse_dists || !distances || !labels)
        return fail(IVFHNSW_ERR_INVALID, "null buffer (the coarse stage is computed once and supplied)");
    if (nq > kMaxBatchAll)
        return fail(IVFHNSW_ERR_INVALID, "search_sharded is limited to %zu queries per call", kMaxBatchAll);
    const size_t d = shards[0]->t.d, np = p->nprobe, nk = nq * k;
    int rc;
    // 1. every shard: inputs to its device, scan of its lists, keys left on the device (asynchronous, one stream each)
    for (size_t r = 0; r < nshards; r++) {
        ivfhnsw_gpu *h = shards[r];
        if ((rc = bind(h)))
            return rc;
        if ((rc = h->s_q.ensure(nq * d * sizeof(float))) || (rc = h->s_dist.ensure(nk * sizeof(float))) ||
            (rc = h->s_lab.ensure(nk * sizeof(int64_t))) || (rc = h->s_keys.ensure(nk * sizeof(int64_t))) ||
            (rc = h->s_cid.ensure(nq * np * sizeof(uint32_t))) || (rc = h->s_cd.ensure(nq * np * sizeof(float))))
            return rc;
        HIP_TRY(hipMemcpyAsync(h->s_q.p, queries, nq * d * sizeof(float), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_cid.p, coarse_ids, nq * np * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_cd.p, coarse_dists, nq * np * sizeof(float), hipMemcpyHostToDevice, h->stream));
        if ((rc = ivfhnsw_gpu_search_dev(h, nq, k, h->s_q.as<float>(), h->s_cid.as<uint32_t>(), h->s_cd.as<float>(), p,
                                         h->s_dist.as<float>(), h->s_lab.as<int64_t>(), h->s_keys.as<int64_t>())))
            return rc;
    }
    // 2. the merge.  Over RCCL when every shard has a device of its own (or, for the single shard of a test, when
    // IVFHNSW_SHARDS_RCCL=1 asks for the calls to be made anyway); k = 1: one int64 MIN of the keys, one MAX of the labels
    std::vector<int> devs(nshards);
    bool distinct = true;
    for (size_t r = 0; r < nshards; r++) {
        devs[r] = shards[r]->device;
        for (size_t q = 0; q < r; q++)
            distinct = distinct && devs[q] != devs[r];
    }
    static const int force = [] {
        const char *e = getenv("IVFHNSW_SHARDS_RCCL");
        return (e && *e) ? atoi(e) : -1;
    }();
    std::vector<ncclComm_t> *comms = nullptr;
    if (k == 1 && distinct && force != 0 && (nshards > 1 || force == 1))
        comms = rccl_comms(devs);
    if (comms) {
        Rccl &R = rccl();
        auto all_reduce = [&](bool labels_pass) -> int {
            ncclResult_t e = R.GroupStart();
            for (size_t r = 0; r < nshards && e == ncclSuccess; r++) {
                ivfhnsw_gpu *h = shards[r];
                (void)hipSetDevice(h->device);
                void *buf = labels_pass ? h->s_lab.p : h->s_keys.p;
                e = R.AllReduce(buf, buf, nk, ncclInt64, labels_pass ? ncclMax : ncclMin, (*comms)[r], h->stream);
            }
            const ncclResult_t e2 = R.GroupEnd();
            return (int)(e != ncclSuccess ? e : e2);
        };
        int e = all_reduce(false);
        if (e)
            return fail(IVFHNSW_ERR_HIP, "RCCL all-reduce (MIN of the keys) failed: %s", R.GetErrorString ? R.GetErrorString((ncclResult_t)e) : "?");
        for (size_t r = 0; r < nshards; r++) {
            ivfhnsw_gpu *h = shards[r];
            if ((rc = ivfhnsw_gpu_resolve_keys_dev(h, nq, k, h->s_keys.as<int64_t>(), h->s_dist.as<float>(), h->s_lab.as<int64_t>())))
                return rc;
        }
        if ((e = all_reduce(true)))
            return fail(IVFHNSW_ERR_HIP, "RCCL all-reduce (MAX of the labels) failed: %s", R.GetErrorString ? R.GetErrorString((ncclResult_t)e) : "?");
        ivfhnsw_gpu *h0 = shards[0];
        if ((rc = bind(h0)))
            return rc;
        HIP_TRY(hipMemcpyAsync(distances, h0->s_dist.p, nk * sizeof(float), hipMemcpyDeviceToHost, h0->stream));
        HIP_TRY(hipMemcpyAsync(labels, h0->s_lab.p, nk * sizeof(int64_t), hipMemcpyDeviceToHost, h0->stream));
        for (size_t r = 0; r < nshards; r++) {
            if ((rc = bind(shards[r])))
                return rc;
            HIP_TRY(hipStreamSynchronize(shards[r]->stream));
            if ((rc = check_status(shards[r])))
                return rc;
        }
        return IVFHNSW_OK;
    }
    // ... on the host otherwise (shards sharing a device, k > 1): the k smallest keys per query over the shards
    std::vector<std::vector<int64_t>> keys(nshards, std::vector<int64_t>(nk));
    for (size_t r = 0; r < nshards; r++) {
        ivfhnsw_gpu *h = shards[r];
        if ((rc = bind(h)))
            return rc;
        HIP_TRY(hipMemcpyAsync(keys[r].data(), h->s_keys.p, nk * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    }
    for (size_t r = 0; r < nshards; r++) {
        if ((rc = bind(shards[r])))
            return rc;
        HIP_TRY(hipStreamSynchronize(shards[r]->stream));
        if ((rc = check_status(shards[r])))
            return rc;
    }
    std::vector<int64_t> merged(nk);
    std::vector<int64_t> pool(nshards * k);
    for (size_t i = 0; i < nq; i++) {
        if (k == 1) {
            int64_t m = keys[0][i];
            for (size_t r = 1; r < nshards; r++)
                m = keys[r][i] < m ? keys[r][i] : m;
            merged[i] = m;
        } else {
            for (size_t r = 0; r < nshards; r++)
                std::copy(keys[r].begin() + i * k, keys[r].begin() + (i + 1) * k, pool.begin() + r * k);
            std::partial_sort(pool.begin(), pool.begin() + k, pool.end());
            std::copy(pool.begin(), pool.begin() + k, merged.begin() + i * k);
        }
    }
    std::vector<int64_t> lab(nk);
    for (size_t r = 0; r < nshards; r++) {
        ivfhnsw_gpu *h = shards[r];
        if ((rc = bind(h)))
            return rc;
        HIP_TRY(hipMemcpyAsync(h->s_keys.p, merged.data(), nk * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        if ((rc = ivfhnsw_gpu_resolve_keys_dev(h, nq, k, h->s_keys.as<int64_t>(), h->s_dist.as<float>(), h->s_lab.as<int64_t>())))
            return rc;
    }
    for (size_t r = 0; r < nshards; r++) {
        ivfhnsw_gpu *h = shards[r];
        if ((rc = bind(h)))
            return rc;
        HIP_TRY(hipMemcpyAsync(lab.data(), h->s_lab.p, nk * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        if (r == 0)
            HIP_TRY(hipMemcpyAsync(distances, h->s_dist.p, nk * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < nk; i++)
            labels[i] = (r == 0 || lab[i] > labels[i]) ? lab[i] : labels[i];
    }
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return fail(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_search_sharded: host allocation failed");
}

int ivfhnsw_gpu_resolve_keys(ivfhnsw_gpu *h, size_t nq, size_t k, const int64_t *keys, float *distances, int64_t *labels)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (nq == 0)
        return IVFHNSW_OK;
    if (!keys || !distances || !labels)
        return fail(IVFHNSW_ERR_INVALID, "null buffer");
    if ((rc = h->s_keys.ensure(nq * k * sizeof(int64_t))) || (rc = h->s_dist.ensure(nq * k * sizeof(float))) ||
        (rc = h->s_lab.ensure(nq * k * sizeof(int64_t))))
        return rc;
    HIP_TRY(hipMemcpyAsync(h->s_keys.p, keys, nq * k * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    if ((rc = ivfhnsw_gpu_resolve_keys_dev(h, nq, k, h->s_keys.as<int64_t>(), h->s_dist.as<float>(), h->s_lab.as<int64_t>())))
        return rc;
    HIP_TRY(hipMemcpyAsync(distances, h->s_dist.p, nq * k * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(labels, h->s_lab.p, nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_last_stream(ivfhnsw_gpu *h, size_t nq, size_t len_cap, uint64_t *keys, uint32_t *lens, uint32_t *stream_cap)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (lens) {
        if ((rc = h->s_len.ensure(nq * sizeof(uint32_t))))
            return rc;
        if ((rc = ivfhnsw_gpu_last_stream_dev(h, nq, 0, nullptr, h->s_len.as<uint32_t>(), stream_cap)))
            return rc;
        HIP_TRY(hipMemcpyAsync(lens, h->s_len.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    }
    if (keys) {
        if ((rc = h->s_keys.ensure(nq * len_cap * sizeof(uint64_t))))
            return rc;
        if ((rc = ivfhnsw_gpu_last_stream_dev(h, nq, len_cap, h->s_keys.as<uint64_t>(), nullptr, stream_cap)))
            return rc;
        HIP_TRY(hipMemcpyAsync(keys, h->s_keys.p, nq * len_cap * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_set_profiling(ivfhnsw_gpu *h, int enabled)
{
    int rc = bind(h);
    if (rc)
        return rc;
    h->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    if (h->split_view)
        h->split_view->profiling = h->profiling;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_get_stage_ms(ivfhnsw_gpu *h, int stage, double *ms_total, uint64_t *launches)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if (stage < 0 || stage >= IVFHNSW_STAGE_COUNT)
        return fail(IVFHNSW_ERR_INVALID, "bad stage %d", stage);
    if ((rc = drain_events(h)))
        return rc;
    double ms = h->stage_ms[stage];
    uint64_t n = h->stage_n[stage];
    if (h->split_view) { // the second part of split batches: its launches and their time join the handle's
        if ((rc = drain_events(h->split_view)))
            return rc;
        ms += h->split_view->stage_ms[stage];
        n += h->split_view->stage_n[stage];
    }
    if (ms_total)
        *ms_total = ms;
    if (launches)
        *launches = n;
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_reset_stage_ms(ivfhnsw_gpu *h)
{
    int rc = bind(h);
    if (rc)
        return rc;
    if ((rc = drain_events(h)))
        return rc;
    for (int i = 0; i < IVFHNSW_STAGE_COUNT; i++) {
        h->stage_ms[i] = 0;
        h->stage_n[i] = 0;
    }
    if (h->split_view)
        return ivfhnsw_gpu_reset_stage_ms(h->split_view);
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_last_scan_counts(ivfhnsw_gpu *h, uint64_t *ncodes, uint64_t *nsegments)
{
    int rc = bind(h);
    if (rc)
        return rc;
    unsigned long long out[2] = {0, 0};
    if (h->last_nq) {
        HIP_TRY(launch_plan_totals(h->stream, h->w_hdr.as<PlanHdr>(), h->last_nq,
                                   h->w_totals.as<unsigned long long>()));
        HIP_TRY(hipMemcpyAsync(out, h->w_totals.p, sizeof(out), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->last_split && h->split_view) {
        uint64_t c2 = 0, s2 = 0;
        if ((rc = ivfhnsw_gpu_last_scan_counts(h->split_view, &c2, &s2)))
            return rc;
        out[0] += c2;
        out[1] += s2;
        (void)hipSetDevice(h->device);
    }
    if (ncodes)
        *ncodes = out[0];
    if (nsegments)
        *nsegments = out[1];
    return IVFHNSW_OK;
}

const char *ivfhnsw_gpu_last_scan_kernel(ivfhnsw_gpu *h) { return h ? h->last_scan_kernel : ""; }

int ivfhnsw_gpu_last_batch_parts(ivfhnsw_gpu *h, uint64_t *first, uint64_t *second)
{
    if (!h)
        return fail(IVFHNSW_ERR_INVALID, "null handle");
    if (first)
        *first = h->last_parts[0];
    if (second)
        *second = h->last_parts[1];
    return IVFHNSW_OK;
}

int ivfhnsw_gpu_memory_bytes(ivfhnsw_gpu *h, uint64_t *bytes)
{
    if (!h || !bytes)
        return fail(IVFHNSW_ERR_INVALID, "null argument");
    const DevBuf *all[] = {&h->goff, &h->loff, &h->cnorm, &h->pqc, &h->ntab, &h->opq_at, &h->codes, &h->ncodes,
                           &h->ids, &h->g_alpha, &h->g_nn, &h->g_sizes, &h->g_inter, &h->q_counts, &h->q_links,
                           &h->q_vectors, &h->q_qrows, &h->q_nbrows, &h->q_nbnorms, &h->q_fat, &h->q_links_c, &h->e_pqc, &h->e_ntab, &h->e_a, &h->e_at, &h->e_x, &h->e_idx, &h->e_dist, &h->e_res, &h->e_tmp, &h->e_codes, &h->e_ncodes, &h->cg_q, &h->cg_cidx, &h->cg_ids, &h->cg_dists, &h->gc_nn, &h->cg_cvn, &h->cg_tab, &h->cg_tab2, &h->cg_off, &h->cg_alpha2, &h->cg_sub, &h->w_xq, &h->w_luts, &h->w_segs, &h->w_lpos, &h->w_hdr, &h->w_keys,
                           &h->w_cid, &h->w_cd, &h->w_qsd, &h->w_totals, &h->w_visited, &h->w_status, &h->w_stream, &h->w_slen, &h->w_counter, &h->w_tail, &h->w_redo, &h->k_q, &h->k_x, &h->k_qn, &h->k_xn, &h->k_part, &h->k_ids, &h->k_dists, &h->t_x, &h->t_y, &h->t_cb, &h->t_assign, &h->t_part, &h->t_c, &h->s_q, &h->s_cid, &h->s_cd,
                           &h->s_dist, &h->s_lab};
    uint64_t s = 0;
    for (auto *b : all)
        s += b->bytes;
    *bytes = s;
    return IVFHNSW_OK;
}

} // extern "C"
