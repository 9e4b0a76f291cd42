// ivfhnsw::IndexIVF_HNSW over the MI355X C ABI (include/ivf-hnsw/IndexIVF_HNSW.h).
//
// search / search2 / assign / search_batch run on the device; read / write keep the reference's .index format
// (IndexIVF_HNSW.cpp:637-663,758-779); the construction side (add_batch, train_pq) is a plain host
// implementation kept only so that the drivers' cold-start path works at small scale.
#include <ivf-hnsw/IndexIVF_HNSW.h>

#include <ivfhnsw_hip.h>

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
    std::vector<float> v(d);
    for (size_t i = 0; i < nc; i++) {
        readXvec<float>(input, v.data(), d);
        if (i % 100000 == 0)
            std::cout << i / (0.01 * nc) << " %\n";
        quantizer->addPoint(v.data());
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

    std::vector<uint64_t> off(nc + 1, 0);
    for (size_t c = 0; c < nc; c++) {
        if (codes[c].size() != ids[c].size() * code_size || norm_codes[c].size() != ids[c].size())
            throw std::runtime_error("IndexIVF_HNSW: list " + std::to_string(c) + " has inconsistent sizes");
        off[c + 1] = off[c] + ids[c].size();
    }
    const size_t total = off[nc];
    std::vector<idx_t> fid(total ? total : 1);
    std::vector<uint8_t> fcode(total ? total * code_size : 1), fnorm(total ? total : 1);
    for (size_t c = 0; c < nc; c++) { // list order is scan order (ties): keep it
        std::copy(ids[c].begin(), ids[c].end(), fid.begin() + off[c]);
        std::copy(codes[c].begin(), codes[c].end(), fcode.begin() + off[c] * code_size);
        std::copy(norm_codes[c].begin(), norm_codes[c].end(), fnorm.begin() + off[c]);
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
    desc.shard_rank = 0;
    desc.shard_world = 1;
    desc.list_owner = nullptr;
    if (pq->centroids.size() != 256 * d || norm_pq->centroids.size() != 256)
        throw std::runtime_error("IndexIVF_HNSW: pq / norm_pq have unexpected shapes");
    if (ivfhnsw_gpu_upload_ivf(gpu_, &desc))
        gpu_fail("ivfhnsw_gpu_upload_ivf");

    upload_graph();

    up_pq_ = pq;
    up_norm_pq_ = norm_pq;
    up_opq_ = opq_matrix;
    up_quantizer_ = quantizer;
    up_total_ = total;
    up_do_opq_ = do_opq;
}

void IndexIVF_HNSW::upload_graph()
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
    if (ivfhnsw_gpu_upload_quantizer(gpu_, n, d, maxM, quantizer->enterpoint_node, cnt.data(), lnk.data(), vec.data()))
        gpu_fail("ivfhnsw_gpu_upload_quantizer");
    graph_dirty_ = false;
    graph_uploaded_for_ = quantizer;
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

void IndexIVF_HNSW::search_batch(size_t nq, size_t k, const float *x, float *distances, long *labels)
{
    static_assert(sizeof(long) == sizeof(int64_t), "LP64 expected");
    ensure_device();
    ivfhnsw_search_params p;
    p.nprobe = nprobe;
    p.max_codes = max_codes;
    p.efSearch = quantizer->efSearch;
    p.do_pruning = 0;
    p.heap_order = 1; // k > 1: the array faiss's max-heap leaves, as the reference returns it
    if (ivfhnsw_gpu_search(gpu_, nq, k, x, nullptr, nullptr, &p, distances, reinterpret_cast<int64_t *>(labels)))
        gpu_fail("ivfhnsw_gpu_search");
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
    ivfhnsw_search_params p = {nprobe, max_codes, quantizer->efSearch, 0, 1};
    if (ivfhnsw_gpu_search(gpu_, 1, k, x, cid.data(), cd.data(), &p, distances, reinterpret_cast<int64_t *>(labels)))
        gpu_fail("ivfhnsw_gpu_search");
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
    ivfhnsw_search_params p = {1, max_codes, quantizer->efSearch, 0, 1};
    if (ivfhnsw_gpu_search(gpu_, 1, 1, x, &cid, &cd, &p, distances, reinterpret_cast<int64_t *>(labels)))
        gpu_fail("ivfhnsw_gpu_search");
    return cid;
}

void IndexIVF_HNSW::search2(size_t k, const float *x, float *distances, long *labels, float *query_centroid_dists,
                            idx_t *centroid_idxs)
{
    ensure_device();
    ivfhnsw_search_params p = {nprobe, max_codes, quantizer->efSearch, 0, 1};
    if (ivfhnsw_gpu_search(gpu_, 1, k, x, centroid_idxs, query_centroid_dists, &p, distances,
                           reinterpret_cast<int64_t *>(labels)))
        gpu_fail("ivfhnsw_gpu_search");
}

void IndexIVF_HNSW::search2m(size_t k, const float *x, float *distances[], long *labels[],
                             float *query_centroid_dists, idx_t *centroid_idxs)
{
    // one result heap per probe (the reference's variant is racy; this one is well defined): probe i alone
    ensure_device();
    for (size_t i = 0; i < nprobe; i++) {
        ivfhnsw_search_params p = {1, (size_t)-1, quantizer->efSearch, 0, 1};
        if (ivfhnsw_gpu_search(gpu_, 1, k, x, centroid_idxs + i, query_centroid_dists + i, &p, distances[i],
                               reinterpret_cast<int64_t *>(labels[i])))
            gpu_fail("ivfhnsw_gpu_search");
    }
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
