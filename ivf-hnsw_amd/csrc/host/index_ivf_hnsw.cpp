// ivfhnsw::IndexIVF_HNSW over the MI355X C ABI (include/ivf-hnsw/IndexIVF_HNSW.h).
//
// search / search2 / assign / search_batch run on the device; read / write keep the reference's .index format
// (IndexIVF_HNSW.cpp:637-663,758-779); the construction side (add_batch, train_pq) is a plain host
// implementation kept only so that the drivers' cold-start path works at small scale.
#include <ivf-hnsw/IndexIVF_HNSW.h>

#include <ivfhnsw_hip.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace ivfhnsw {

namespace {

[[noreturn]] void gpu_fail(const char *what)
{
    throw std::runtime_error(std::string(what) + ": " + ivfhnsw_gpu_last_error());
}

} // namespace

IndexIVF_HNSW::IndexIVF_HNSW(size_t dim, size_t ncentroids, size_t bytes_per_code, size_t nbits_per_idx,
                             size_t max_group_size)
    : d(dim), nc(ncentroids), code_size(0), quantizer(nullptr), pq(nullptr), norm_pq(nullptr), opq_matrix(nullptr),
      do_opq(false), nprobe(1), max_codes(0), M(16), gpu_(nullptr), device_dirty_(true), graph_dirty_(true), graph_uploaded_for_(nullptr), latency_for_(nullptr), up_pq_(nullptr),
      up_norm_pq_(nullptr), up_opq_(nullptr), up_quantizer_(nullptr), up_total_(0), up_do_opq_(false)
{
    std::memset(&hdr_idx, 0, sizeof(hdr_idx));
    pq = new faiss::ProductQuantizer(d, bytes_per_code, nbits_per_idx);
    norm_pq = new faiss::ProductQuantizer(1, 1, nbits_per_idx);
    code_size = pq->code_size;
    norms.resize(max_group_size);
    precomputed_table.resize(pq->ksub * pq->M);
    codes.resize(nc);
    norm_codes.resize(nc);
    ids.resize(nc);
    centroid_norms.resize(nc);
}

IndexIVF_HNSW::~IndexIVF_HNSW()
{
    for (ivfhnsw_gpu *sh : shards_)
        ivfhnsw_gpu_destroy(sh);
    if (gpu_)
        ivfhnsw_gpu_destroy(gpu_);
    delete quantizer;
    delete pq;
    delete norm_pq;
    delete opq_matrix;
}

void IndexIVF_HNSW::build_quantizer(const char *path_data, const char *path_info, const char *path_edges, size_t M_,
                                    size_t efConstruction)
{
    hdr_idx.efConstruction = (uint32_t)efConstruction;
    M = M_;
    device_dirty_ = true;
    graph_dirty_ = true;
    if (exists(path_info) && exists(path_edges)) {
        quantizer = new hnswlib::HierarchicalNSW(path_info, path_data, path_edges);
        quantizer->efSearch = efConstruction;
        return;
    }
    quantizer = new hnswlib::HierarchicalNSW(d, nc, M_, 2 * M_, efConstruction);
    std::cout << "Constructing quantizer\n";
    std::ifstream input(path_data, std::ios::binary);
    // IVFHNSW_BUILD=device: the insertion loop for all centroids at once on the device (ivfhnsw_gpu_build_graph: exact
    // candidates on the matrix cores, then this very heuristic and connect step per node) -- seconds for a million
    // centroids where the serial loop below takes hours (README.md:65 of the reference); the default keeps the serial
    // loop, whose graph equals the reference's link for link.
    static const bool on_device = [] {
        const char *e = getenv("IVFHNSW_BUILD");
        return e && std::string(e) == "device";
    }();
    if (on_device && nc > 1 && d % 4 == 0 && d <= 128 && 2 * M_ <= 64) {
        std::vector<float> all(nc * d);
        for (size_t i = 0; i < nc; i++)
            readXvec<float>(input, all.data() + i * d, d);
        std::vector<uint8_t> cnt(nc);
        std::vector<idx_t> lk(nc * 2 * M_);
        if (!gpu_ && ivfhnsw_gpu_create(0, &gpu_))
            gpu_fail("ivfhnsw_gpu_create");
        const size_t ncand = std::min<size_t>(80, std::max<size_t>(4 * M_, 2 * M_));
        if (ivfhnsw_gpu_build_graph(gpu_, nc, d, all.data(), M_, 2 * M_, ncand, cnt.data(), lk.data()))
            gpu_fail("ivfhnsw_gpu_build_graph");
        for (size_t i = 0; i < nc; i++) {
            uint8_t *rec = quantizer->get_linklist0((idx_t)i);
            std::memset(rec, 0, quantizer->size_data_per_element);
            rec[0] = cnt[i];
            std::memcpy(rec + 1, lk.data() + i * 2 * M_, (size_t)cnt[i] * sizeof(idx_t));
            std::memcpy(quantizer->getDataByInternalId((idx_t)i), all.data() + i * d, d * sizeof(float));
        }
        quantizer->cur_element_count = nc;
        quantizer->enterpoint_node = 0;
    } else {
        std::vector<float> v(d);
        for (size_t i = 0; i < nc; i++) {
            readXvec<float>(input, v.data(), d);
            if (i % 100000 == 0)
                std::cout << i / (0.01 * nc) << " %\n";
            quantizer->addPoint(v.data());
        }
    }
    quantizer->SaveInfo(path_info);
    quantizer->SaveEdges(path_edges);
}

// ------------------------------------------------------------------------------------------ device mirror
void IndexIVF_HNSW::ensure_latency_walk()
{
    // once per uploaded graph; shapes the latency form does not take (IVFHNSW_ERR_INVALID) keep the throughput walk
    static const bool off = [] {
        const char *e = getenv("IVFHNSW_LATENCY");
        return e && *e && atoi(e) == 0;
    }();
    if (off || latency_for_ == graph_uploaded_for_ || !graph_uploaded_for_)
        return;
    latency_for_ = graph_uploaded_for_;
    const int rc = ivfhnsw_gpu_prepare_latency(gpu_);
    if (rc != IVFHNSW_OK && rc != IVFHNSW_ERR_INVALID)
        gpu_fail("ivfhnsw_gpu_prepare_latency");
}

void IndexIVF_HNSW::ensure_device()
{
    // The list total is the fingerprint that catches a driver appending to the public lists behind the class's back;
    // summing a million list sizes per call would cost more than the search itself (one query per call at 993 127
    // centroids), so beyond 2^16 lists the class relies on its own dirty flags (add_batch, read, invalidate_device()).
    size_t total = up_total_;
    if (nc <= (1u << 16) || !gpu_) {
        total = 0;
        for (size_t c = 0; c < nc; c++)
            total += ids[c].size();
    }
    if (!gpu_ || device_dirty_ || up_pq_ != pq || up_norm_pq_ != norm_pq || up_opq_ != opq_matrix ||
        up_quantizer_ != quantizer || up_total_ != total || up_do_opq_ != do_opq)
        sync_to_device();
}

void IndexIVF_HNSW::device_upload_common()
{
    if (!quantizer)
        throw std::runtime_error("IndexIVF_HNSW: no quantizer (call build_quantizer first)");
    if (!gpu_ && ivfhnsw_gpu_create(0, &gpu_))
        gpu_fail("ivfhnsw_gpu_create");
    if (do_opq && !opq_matrix)
        throw std::runtime_error("IndexIVF_HNSW: do_opq is set but opq_matrix is null");
    if (pq->centroids.size() != 256 * d || norm_pq->centroids.size() != 256)
        throw std::runtime_error("IndexIVF_HNSW: pq / norm_pq have unexpected shapes");
    // IVFHNSW_SHARDS=N: N handles, shard r on device r % (devices of the node), lists c % N == r
    static const size_t want_shards = [] {
        const char *e = getenv("IVFHNSW_SHARDS");
        const long v = (e && *e) ? atol(e) : 1;
        return (size_t)(v < 1 ? 1 : v > 64 ? 64 : v);
    }();
    if (nshards() != want_shards) {
        for (ivfhnsw_gpu *sh : shards_)
            ivfhnsw_gpu_destroy(sh);
        shards_.clear();
        int ndev = 1;
        if (ivfhnsw_gpu_device_count(&ndev) || ndev < 1)
            gpu_fail("ivfhnsw_gpu_device_count");
        for (size_t r = 1; r < want_shards; r++) {
            ivfhnsw_gpu *sh = nullptr;
            if (ivfhnsw_gpu_create((int)(r % (size_t)ndev), &sh))
                gpu_fail("ivfhnsw_gpu_create");
            shards_.push_back(sh);
        }
    }
    const size_t world = nshards();

    std::vector<uint64_t> off(nc + 1, 0);
    for (size_t c = 0; c < nc; c++) {
        if (codes[c].size() != ids[c].size() * code_size || norm_codes[c].size() != ids[c].size())
            throw std::runtime_error("IndexIVF_HNSW: list " + std::to_string(c) + " has inconsistent sizes");
        off[c + 1] = off[c] + ids[c].size();
    }
    const size_t total = off[nc];
    for (size_t r = 0; r < world; r++) {
        size_t mine = 0;
        for (size_t c = r; c < nc; c += world)
            mine += ids[c].size();
        std::vector<idx_t> fid(mine ? mine : 1);
        std::vector<uint8_t> fcode(mine ? mine * code_size : 1), fnorm(mine ? mine : 1);
        size_t at = 0;
        for (size_t c = r; c < nc; c += world) { // list order is scan order (ties): keep it
            std::copy(ids[c].begin(), ids[c].end(), fid.begin() + at);
            std::copy(codes[c].begin(), codes[c].end(), fcode.begin() + at * code_size);
            std::copy(norm_codes[c].begin(), norm_codes[c].end(), fnorm.begin() + at);
            at += ids[c].size();
        }
        ivfhnsw_ivf_desc desc;
        std::memset(&desc, 0, sizeof(desc));
        desc.d = d;
        desc.nc = nc;
        desc.code_size = code_size;
        desc.offsets = off.data();
        desc.ids = fid.data();
        desc.codes = fcode.data();
        desc.norm_codes = fnorm.data();
        desc.centroid_norms = centroid_norms.data();
        desc.pq_centroids = pq->centroids.data();
        desc.norm_table = norm_pq->centroids.data();
        desc.opq_A = do_opq ? opq_matrix->A.data() : nullptr;
        desc.shard_rank = (uint32_t)r;
        desc.shard_world = (uint32_t)world;
        desc.list_owner = nullptr;
        if (ivfhnsw_gpu_upload_ivf(shard(r), &desc))
            gpu_fail("ivfhnsw_gpu_upload_ivf");
    }

    upload_graph();
    if (shards_need_graph())
        for (ivfhnsw_gpu *sh : shards_)
            upload_graph_to(sh);

    up_pq_ = pq;
    up_norm_pq_ = norm_pq;
    up_opq_ = opq_matrix;
    up_quantizer_ = quantizer;
    up_total_ = total;
    up_do_opq_ = do_opq;
}

void IndexIVF_HNSW::upload_graph()
{
    upload_graph_to(gpu_);
    graph_dirty_ = false;
    graph_uploaded_for_ = quantizer;
    // ivfhnsw_gpu_upload_quantizer discards the fat copy the latency walk reads: the next one-query search() must
    // prepare it again even though the quantizer POINTER is the one it was prepared for (add_batch, read and
    // rotate_quantizer all come through here)
    latency_for_ = nullptr;
}

void IndexIVF_HNSW::upload_graph_to(ivfhnsw_gpu *handle)
{
    // node records [count][maxM links][d floats] -> three arrays
    const size_t maxM = quantizer->maxM_, n = quantizer->maxelements_;
    std::vector<uint8_t> cnt(n);
    std::vector<uint32_t> lnk(n * maxM, 0);
    std::vector<float> vec(n * d);
    for (size_t i = 0; i < n; i++) {
        const uint8_t *rec = quantizer->get_linklist0((idx_t)i);
        cnt[i] = rec[0];
        std::memcpy(&lnk[i * maxM], rec + 1, (size_t)rec[0] * sizeof(uint32_t));
        std::memcpy(&vec[i * d], quantizer->getDataByInternalId((idx_t)i), d * sizeof(float));
    }
    if (ivfhnsw_gpu_upload_quantizer(handle, n, d, maxM, quantizer->enterpoint_node, cnt.data(), lnk.data(), vec.data()))
        gpu_fail("ivfhnsw_gpu_upload_quantizer");
}

void IndexIVF_HNSW::ensure_encoder()
{
    if (!quantizer)
        throw std::runtime_error("IndexIVF_HNSW: no quantizer (call build_quantizer first)");
    if (!pq || !norm_pq)
        throw std::runtime_error("IndexIVF_HNSW::add_batch: pq / norm_pq are not set (train_pq or read them first)");
    if (do_opq && !opq_matrix)
        throw std::runtime_error("IndexIVF_HNSW: do_opq is set but opq_matrix is null");
    if (pq->centroids.size() != 256 * d || norm_pq->centroids.size() != 256)
        throw std::runtime_error("IndexIVF_HNSW: pq / norm_pq have unexpected shapes");
    if (!gpu_ && ivfhnsw_gpu_create(0, &gpu_))
        gpu_fail("ivfhnsw_gpu_create");
    if (graph_dirty_ || graph_uploaded_for_ != quantizer)
        upload_graph();
    // 192 KB: sent with every batch, so a driver that retrains or swaps pq / opq_matrix in place is always seen
    if (ivfhnsw_gpu_upload_codebooks(gpu_, d, code_size, pq->centroids.data(), norm_pq->centroids.data(),
                                     do_opq ? opq_matrix->A.data() : nullptr))
        gpu_fail("ivfhnsw_gpu_upload_codebooks");
}

void IndexIVF_HNSW::sync_to_device()
{
    device_upload_common();
    device_dirty_ = false;
}

// ------------------------------------------------------------------------------------------ search side
void IndexIVF_HNSW::assign(size_t n, const float *x, idx_t *labels, size_t k)
{
    ensure_device();
    std::vector<float> dist(n * k);
    if (ivfhnsw_gpu_coarse(gpu_, n, x, k, quantizer->efSearch < k ? k : quantizer->efSearch, labels, dist.data()))
        gpu_fail("ivfhnsw_gpu_coarse");
}

// Every search of the class: one handle, or -- IVFHNSW_SHARDS=N -- the shard step of SURVEY 8e merged on the host.
// Keys are the library's packed (orderable distance << 32 | scan position) words, sign-flipped so that a signed
// minimum picks the reference's winner (smallest distance, earliest scan position on ties).
namespace {
const uint64_t kSignFlipH = 0x8000000000000000ull;
const uint64_t kKeyInitH = (uint64_t)(0x7f7fffffu | 0x80000000u) << 32;
inline float key_dist(uint64_t ukey)
{
    const uint32_t o = (uint32_t)(ukey >> 32);
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &u, sizeof(f));
    return f;
}
inline uint64_t key_pack(float dist, uint32_t vpos)
{
    uint32_t u;
    std::memcpy(&u, &dist, sizeof(u));
    const uint32_t o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)o << 32) | vpos;
}
} // namespace

void IndexIVF_HNSW::device_search(size_t nq, size_t k, const float *x, const idx_t *coarse_ids, const float *coarse_dists,
                                  size_t nprobe_, size_t max_codes_, bool pruning, float *distances, long *labels)
{
    static_assert(sizeof(long) == sizeof(int64_t), "LP64 expected");
    ivfhnsw_search_params p;
    p.nprobe = nprobe_;
    p.max_codes = max_codes_;
    p.efSearch = quantizer->efSearch;
    p.do_pruning = pruning ? 1 : 0;
    p.heap_order = 1; // k > 1: the array faiss's max-heap leaves, as the reference returns it
    if (shards_.empty()) {
        if (ivfhnsw_gpu_search(gpu_, nq, k, x, coarse_ids, coarse_dists, &p, distances, reinterpret_cast<int64_t *>(labels)))
            gpu_fail("ivfhnsw_gpu_search");
        return;
    }
    if (nq == 0)
        return;
    // the shard step holds one plan per call: at most 2^17 queries (2^14 in heap order, k > 1), like the unsharded
    // entry point, which slices larger batches itself
    const size_t chunk = k > 1 ? ((size_t)1 << 14) : ((size_t)1 << 17);
    if (nq > chunk) {
        for (size_t q0 = 0; q0 < nq; q0 += chunk)
            device_search(std::min(chunk, nq - q0), k, x + q0 * d, coarse_ids ? coarse_ids + q0 * nprobe_ : nullptr,
                          coarse_dists ? coarse_dists + q0 * nprobe_ : nullptr, nprobe_, max_codes_, pruning,
                          distances + q0 * k, labels + q0 * k);
        return;
    }
    const size_t world = nshards();
    // the coarse stage once, on shard 0 (the rotated query walks the rotated graph, IndexIVF_HNSW.cpp:240,248)
    std::vector<idx_t> cid_own;
    std::vector<float> cd_own;
    if (!coarse_ids) {
        std::vector<float> xr;
        const float *q = x;
        if (do_opq) {
            xr.resize(nq * d);
            opq_matrix->apply_noalloc((long)nq, x, xr.data());
            q = xr.data();
        }
        cid_own.resize(nq * nprobe_);
        cd_own.resize(nq * nprobe_);
        if (ivfhnsw_gpu_coarse(gpu_, nq, q, nprobe_, quantizer->efSearch, cid_own.data(), cd_own.data()))
            gpu_fail("ivfhnsw_gpu_coarse");
        coarse_ids = cid_own.data();
        coarse_dists = cd_own.data();
    }
    const bool heap = k > 1;
    p.heap_order = heap ? 1 : 0;
    if (!heap) {
        // k = 1 (every preset of the reference): the whole shard step below the C ABI -- scans on the shards' devices, the
        // keys MIN-merged and the labels MAX-merged over RCCL between the devices (on the host when shards share a device)
        std::vector<ivfhnsw_gpu *> hs(world);
        for (size_t r = 0; r < world; r++)
            hs[r] = shard(r);
        if (ivfhnsw_gpu_search_sharded(hs.data(), world, nq, k, x, coarse_ids, coarse_dists, &p, distances,
                                       reinterpret_cast<int64_t *>(labels)))
            gpu_fail("ivfhnsw_gpu_search_sharded");
        return;
    }
    std::vector<std::vector<int64_t>> keys(world, std::vector<int64_t>(nq * k));
    std::vector<std::string> err(world);
    // one host thread per shard: every handle has its own device and stream
#pragma omp parallel for schedule(static, 1)
    for (long r = 0; r < (long)world; r++)
        if (ivfhnsw_gpu_search_keys(shard((size_t)r), nq, k, x, coarse_ids, coarse_dists, &p, keys[(size_t)r].data()))
            err[(size_t)r] = ivfhnsw_gpu_last_error();
    for (const auto &e : err)
        if (!e.empty())
            throw std::runtime_error("ivfhnsw_gpu_search_keys: " + e);
    std::vector<int64_t> merged(nq * k);
    if (!heap) {
        // k = 1: the smallest key over the shards
        for (size_t i = 0; i < nq; i++) {
            int64_t m = keys[0][i];
            for (size_t r = 1; r < world; r++)
                m = keys[r][i] < m ? keys[r][i] : m;
            merged[i] = m;
        }
    } else {
        // k > 1, the reference's heap ARRAY (IndexIVF_HNSW.cpp:265,285-288): the shards' candidate streams -- supersets
        // of what faiss's heap admits, each in its scan order -- merged by scan position and replayed through the very
        // maxheap_pop / maxheap_push the reference calls; a code failing `dist < distances[0]` leaves the heap untouched
        std::vector<std::vector<uint32_t>> lens(world, std::vector<uint32_t>(nq));
        uint32_t cap = 0, lmax = 1;
        for (size_t r = 0; r < world; r++) {
            if (ivfhnsw_gpu_last_stream(shard(r), nq, 0, nullptr, lens[r].data(), &cap))
                gpu_fail("ivfhnsw_gpu_last_stream");
            for (uint32_t l : lens[r]) {
                if (l > cap)
                    throw std::runtime_error("IndexIVF_HNSW: candidate stream of a query exceeded the device buffer "
                                             "(k / max_codes too large for heap-array order across shards)");
                lmax = l > lmax ? l : lmax;
            }
        }
        std::vector<std::vector<uint64_t>> streams(world, std::vector<uint64_t>(nq * (size_t)lmax));
        for (size_t r = 0; r < world; r++)
            if (ivfhnsw_gpu_last_stream(shard(r), nq, lmax, streams[r].data(), nullptr, &cap))
                gpu_fail("ivfhnsw_gpu_last_stream");
        std::vector<uint64_t> all;
        std::vector<float> hv(k);
        std::vector<long> hp(k);
        for (size_t i = 0; i < nq; i++) {
            all.clear();
            for (size_t r = 0; r < world; r++)
                all.insert(all.end(), streams[r].begin() + i * lmax, streams[r].begin() + i * lmax + lens[r][i]);
            std::sort(all.begin(), all.end(),
                      [](uint64_t a, uint64_t b) { return (uint32_t)a < (uint32_t)b; }); // scan order
            faiss::maxheap_heapify(k, hv.data(), hp.data());
            for (uint64_t ukey : all) {
                const float dj = key_dist(ukey);
                if (dj < hv[0]) {
                    faiss::maxheap_pop(k, hv.data(), hp.data());
                    faiss::maxheap_push(k, hv.data(), hp.data(), dj, (long)(uint32_t)ukey);
                }
            }
            for (size_t j = 0; j < k; j++)
                merged[i * k + j] =
                    (int64_t)((hp[j] < 0 ? kKeyInitH : key_pack(hv[j], (uint32_t)hp[j])) ^ kSignFlipH);
        }
    }
    // every shard resolves the labels it owns (-1 elsewhere): the maximum is the label
    std::vector<std::vector<long>> lab(world, std::vector<long>(nq * k));
    std::vector<std::vector<float>> dis(world, std::vector<float>(nq * k));
#pragma omp parallel for schedule(static, 1)
    for (long r = 0; r < (long)world; r++)
        if (ivfhnsw_gpu_resolve_keys(shard((size_t)r), nq, k, merged.data(), dis[(size_t)r].data(),
                                     reinterpret_cast<int64_t *>(lab[(size_t)r].data())))
            err[(size_t)r] = ivfhnsw_gpu_last_error();
    for (const auto &e : err)
        if (!e.empty())
            throw std::runtime_error("ivfhnsw_gpu_resolve_keys: " + e);
    for (size_t i = 0; i < nq * k; i++) {
        long l = lab[0][i];
        for (size_t r = 1; r < world; r++)
            l = lab[r][i] > l ? lab[r][i] : l;
        labels[i] = l;
        distances[i] = dis[0][i];
    }
}

void IndexIVF_HNSW::search_batch(size_t nq, size_t k, const float *x, float *distances, long *labels)
{
    ensure_device();
    device_search(nq, k, x, nullptr, nullptr, nprobe, max_codes, false, distances, labels);
}

void IndexIVF_HNSW::search(size_t k, const float *x, float *distances, long *labels)
{
#ifdef TRACE_CENTROIDS
    trace_query_centroid_dists.clear();
    trace_centroid_idxs.clear();
#endif
    ensure_device();
    ensure_latency_walk();
    search_batch(1, k, x, distances, labels);
}

void IndexIVF_HNSW::search_debug(size_t k, const float *x, float *distances, long *labels)
{
    // search() plus the coarse-stage report of IndexIVF_HNSW.cpp:343-353: the walk runs once, its result is printed
    // (farthest probe first, as the reference loops) and handed to the scan
    ensure_device();
    std::vector<float> xr(d);
    const float *q = x;
    if (do_opq) {
        opq_matrix->apply_noalloc(1, x, xr.data());
        q = xr.data();
    }
    std::vector<idx_t> cid(nprobe);
    std::vector<float> cd(nprobe);
    if (ivfhnsw_gpu_coarse(gpu_, 1, q, nprobe, quantizer->efSearch, cid.data(), cd.data()))
        gpu_fail("ivfhnsw_gpu_coarse");
    std::cout << "coarse centroids info:" << std::endl;
    for (size_t i = nprobe; i-- > 0;) {
        if (cid[i] >= nc)
            continue; // the walk found fewer than nprobe centroids (undefined in the reference)
        std::cout << "centroid " << cid[i] << " with query distance of " << cd[i] << std::endl;
        std::cout << "group size: " << norm_codes[cid[i]].size() << std::endl;
    }
    device_search(1, k, x, cid.data(), cd.data(), nprobe, max_codes, false, distances, labels);
}

IndexIVF_HNSW::idx_t IndexIVF_HNSW::search_enn(const float *x, float *distances, long *labels)
{
    // nprobe = 1, k = 1, independent of the members (IndexIVF_HNSW.cpp:393-451)
    ensure_device();
    std::vector<float> xr(d);
    const float *q = x;
    if (do_opq) {
        opq_matrix->apply_noalloc(1, x, xr.data());
        q = xr.data();
    }
    idx_t cid = 0;
    float cd = 0.f;
    if (ivfhnsw_gpu_coarse(gpu_, 1, q, 1, quantizer->efSearch ? quantizer->efSearch : 1, &cid, &cd))
        gpu_fail("ivfhnsw_gpu_coarse");
    std::cout << "Get centroid in ENN: " << cid << std::endl;
    device_search(1, 1, x, &cid, &cd, 1, max_codes, false, distances, labels);
    return cid;
}

void IndexIVF_HNSW::search2(size_t k, const float *x, float *distances, long *labels, float *query_centroid_dists,
                            idx_t *centroid_idxs)
{
    ensure_device();
    device_search(1, k, x, centroid_idxs, query_centroid_dists, nprobe, max_codes, false, distances, labels);
}

void IndexIVF_HNSW::search2m(size_t k, const float *x, float *distances[], long *labels[],
                             float *query_centroid_dists, idx_t *centroid_idxs)
{
    // one result heap per probe (the reference's variant is racy; this one is well defined): probe i alone
    ensure_device();
    for (size_t i = 0; i < nprobe; i++)
        device_search(1, k, x, centroid_idxs + i, query_centroid_dists + i, 1, (size_t)-1, false, distances[i], labels[i]);
}

void IndexIVF_HNSW::trace_centroids(size_t idx_q, bool missed)
{
    // the coarse trace lives on the device; fetch it for this query on demand is not wired up
    (void)idx_q;
    (void)missed;
    std::cout << "centroids number " << trace_centroid_idxs.size() << std::endl;
}

float IndexIVF_HNSW::pq_L2sqr(const uint8_t *code)
{
    float result = 0.f;
    for (size_t m = 0; m < code_size; m++)
        result += precomputed_table[pq->ksub * m + code[m]];
    return result;
}

// ------------------------------------------------------------------------------------------ construction side
void IndexIVF_HNSW::compute_residuals(size_t n, const float *x, float *residuals, const idx_t *keys)
{
    for (size_t i = 0; i < n; i++)
        faiss::fvec_madd(d, x + i * d, -1.f, quantizer->getDataByInternalId(keys[i]), residuals + i * d);
}

void IndexIVF_HNSW::reconstruct(size_t n, float *x, const float *decoded_residuals, const idx_t *keys)
{
    for (size_t i = 0; i < n; i++)
        faiss::fvec_madd(d, decoded_residuals + i * d, 1.f, quantizer->getDataByInternalId(keys[i]), x + i * d);
}

void IndexIVF_HNSW::add_batch(size_t n, const float *x, const idx_t *xids, const idx_t *precomputed_idx)
{
    // assign -> residual -> (rotate) -> encode -> decode -> (rotate back) -> reconstruct -> norm -> norm code
    // (the reference's IndexIVF_HNSW.cpp:75-121) is one call on the device; the append loop (:122-131) stays here
    ensure_encoder();
    std::vector<idx_t> idx(n);
    std::vector<uint8_t> xcodes(n * code_size), ncodes(n);
    if (ivfhnsw_gpu_encode(gpu_, n, x, precomputed_idx, quantizer->efSearch, idx.data(), xcodes.data(), ncodes.data()))
        gpu_fail("ivfhnsw_gpu_encode");
    for (size_t i = 0; i < n; i++) {
        const idx_t key = idx[i];
        ids[key].push_back(xids[i]);
        codes[key].insert(codes[key].end(), xcodes.begin() + i * code_size, xcodes.begin() + (i + 1) * code_size);
        norm_codes[key].push_back(ncodes[i]);
    }
    device_dirty_ = true;
}

void IndexIVF_HNSW::add_batch2(size_t, const float *, const idx_t *, const idx_t *, uint64_t *, char *)
{
    throw std::runtime_error("IndexIVF_HNSW::add_batch2: the ORCV vendor format is out of scope (SURVEY.md 2, row 9)");
}

void IndexIVF_HNSW::train_pq(size_t n, const float *x)
{
    std::vector<idx_t> assigned(n);
    assign(n, x, assigned.data());
    std::vector<float> res(n * d), tmp;
    compute_residuals(n, x, res.data(), assigned.data());
    if (do_opq) {
        faiss::OPQMatrix *matrix = new faiss::OPQMatrix((int)d, (int)pq->M);
        matrix->verbose = true;
        matrix->max_train_points = n;
        matrix->niter = 70;
        try {
            matrix->train((long)n, res.data());
        } catch (...) {
            delete matrix;
            throw;
        }
        opq_matrix = matrix;
        tmp = res;
        opq_matrix->apply_noalloc((long)n, tmp.data(), res.data());
    }
    printf("Training %zdx%zd product quantizer on %zd vectors in %zdD\n", pq->M, pq->ksub, n, d);
    pq->verbose = true;
    pq->train((int)n, res.data());
    std::vector<uint8_t> xcodes(n * code_size);
    pq->compute_codes(res.data(), xcodes.data(), n);
    std::vector<float> dec(n * d);
    pq->decode(xcodes.data(), dec.data(), n);
    if (do_opq) {
        tmp = dec;
        opq_matrix->transform_transpose((long)n, tmp.data(), dec.data());
    }
    std::vector<float> rec(n * d), nrm(n);
    reconstruct(n, rec.data(), dec.data(), assigned.data());
    faiss::fvec_norms_L2sqr(nrm.data(), rec.data(), d, n);
    printf("Training %zdx%zd product quantizer on %zd vectors in %zdD\n", norm_pq->M, norm_pq->ksub, n, (size_t)1);
    norm_pq->verbose = true;
    norm_pq->train((int)n, nrm.data());
    device_dirty_ = true;
}

int IndexIVF_HNSW::copy_file(const char *file_src, const char *file_dst)
{
    std::ifstream in(file_src, std::ios::binary);
    std::ofstream out(file_dst, std::ios::binary | std::ios::trunc);
    if (!in || !out)
        return -1;
    out << in.rdbuf();
    return out ? 0 : -1;
}

// ------------------------------------------------------------------------------------------ .index files
void IndexIVF_HNSW::write(const char *path_index, bool do_trunc)
{
    std::ofstream out(path_index, do_trunc ? (std::ios::binary | std::ios::trunc) : std::ios::binary);
    write_variable(out, d);
    write_variable(out, nc);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, ids[c]);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, codes[c]);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, norm_codes[c]);
    write_vector(out, centroid_norms);
}

void IndexIVF_HNSW::write(const char *path_index) { this->write(path_index, false); }

void IndexIVF_HNSW::write2(const char *, size_t, bool, const char *)
{
    throw std::runtime_error("IndexIVF_HNSW::write2: the ORCV vendor format is out of scope (SURVEY.md 2, row 9)");
}

void IndexIVF_HNSW::read(const char *path_index)
{
    std::ifstream in(path_index, std::ios::binary);
    if (!in)
        throw std::runtime_error(std::string("cannot open ") + path_index);
    read_variable(in, d);
    read_variable(in, nc);
    ids.resize(nc);
    codes.resize(nc);
    norm_codes.resize(nc);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, ids[c]);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, codes[c]);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, norm_codes[c]);
    read_vector(in, centroid_norms);
    if (!in)
        throw std::runtime_error(std::string("truncated index file ") + path_index);
    device_dirty_ = true;
}

void IndexIVF_HNSW::compute_centroid_norms()
{
    for (size_t i = 0; i < nc; i++)
        centroid_norms[i] = faiss::fvec_norm_L2sqr(quantizer->getDataByInternalId((idx_t)i), d);
    device_dirty_ = true;
}

void IndexIVF_HNSW::rotate_quantizer()
{
    if (!do_opq) {
        printf("OPQ encoding is turned off\n");
        abort();
    }
    std::vector<float> tmp(d);
    for (size_t i = 0; i < nc; i++) {
        float *c = quantizer->getDataByInternalId((idx_t)i);
        std::memcpy(tmp.data(), c, d * sizeof(float));
        opq_matrix->apply_noalloc(1, tmp.data(), c);
    }
    device_dirty_ = true;
    graph_dirty_ = true;
}

} // namespace ivfhnsw
