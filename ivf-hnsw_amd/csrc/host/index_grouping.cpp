// ivfhnsw::IndexIVF_HNSW_Grouping over the MI355X C ABI (include/ivf-hnsw/IndexIVF_HNSW_Grouping.h).
// search runs on the device (plan_grouping_kernel + the ADC scan); read/write keep the reference's Grouping
// .index layout (IndexIVF_HNSW_Grouping.cpp:397-483).  add_group / train_pq (index construction) are outside
// the search path and not implemented yet.
#include <ivf-hnsw/IndexIVF_HNSW_Grouping.h>

#include <ivfhnsw_hip.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

namespace ivfhnsw {

static FILE *g_centroid_trace = nullptr;

int centriodTraceSetup()
{
    g_centroid_trace = fopen("centriod.log", "w");
    return g_centroid_trace ? 0 : -1;
}

void centriodTraceClose()
{
    if (g_centroid_trace)
        fclose(g_centroid_trace);
    g_centroid_trace = nullptr;
}

IndexIVF_HNSW_Grouping::IndexIVF_HNSW_Grouping(size_t dim, size_t ncentroids, size_t bytes_per_code,
                                               size_t nbits_per_idx, size_t nsubcentroids)
    : IndexIVF_HNSW(dim, ncentroids, bytes_per_code, nbits_per_idx), nsubc(nsubcentroids), do_pruning(false)
{
    alphas.resize(nc);
    nn_centroid_idxs.resize(nc);
    subgroup_sizes.resize(nc);
    query_centroid_dists.assign(nc, 0.f);
    inter_centroid_dists.resize(nc);
}

void IndexIVF_HNSW_Grouping::sync_to_device()
{
    device_upload_common();
    // [nc][nsubc] row-major tables; groups without codes have empty vectors on the host -> zero rows
    std::vector<float> icd(nc * nsubc, 0.f);
    std::vector<uint32_t> nn(nc * nsubc, 0), sz(nc * nsubc, 0);
    for (size_t c = 0; c < nc; c++) {
        if (!subgroup_sizes[c].empty() && subgroup_sizes[c].size() != nsubc)
            throw std::runtime_error("IndexIVF_HNSW_Grouping: subgroup_sizes[" + std::to_string(c) + "] has wrong length");
        std::copy(subgroup_sizes[c].begin(), subgroup_sizes[c].end(), sz.begin() + c * nsubc);
        if (nn_centroid_idxs[c].size() == nsubc)
            std::copy(nn_centroid_idxs[c].begin(), nn_centroid_idxs[c].end(), nn.begin() + c * nsubc);
        if (inter_centroid_dists[c].size() == nsubc)
            std::copy(inter_centroid_dists[c].begin(), inter_centroid_dists[c].end(), icd.begin() + c * nsubc);
    }
    if (ivfhnsw_gpu_upload_grouping(gpu_, nsubc, alphas.data(), nn.data(), sz.data(), icd.data()))
        throw std::runtime_error(std::string("ivfhnsw_gpu_upload_grouping: ") + ivfhnsw_gpu_last_error());
    device_dirty_ = false;
}

void IndexIVF_HNSW_Grouping::search_batch(size_t nq, size_t k, const float *x, float *distances, long *labels)
{
    ensure_device();
    ivfhnsw_search_params p;
    p.nprobe = nprobe;
    p.max_codes = max_codes;
    p.efSearch = quantizer->efSearch;
    p.do_pruning = do_pruning ? 1 : 0;
    p.heap_order = 1;
    if (ivfhnsw_gpu_search(gpu_, nq, k, x, nullptr, nullptr, &p, distances, reinterpret_cast<int64_t *>(labels)))
        throw std::runtime_error(std::string("ivfhnsw_gpu_search: ") + ivfhnsw_gpu_last_error());
}

void IndexIVF_HNSW_Grouping::search(size_t k, const float *x, float *distances, long *labels)
{
#ifdef TRACE_CENTROIDS
    trace_query_centroid_dists.clear();
    trace_centroid_idxs.clear();
#endif
    search_batch(1, k, x, distances, labels);
}

void IndexIVF_HNSW_Grouping::searchDisk(size_t k, const float *query, float *distances, long *labels,
                                        const char *path_base)
{
    // ANN search, then exact re-ranking of its k results against the base file (the reference reads 2k results
    // out of a k-sized search, IndexIVF_HNSW_Grouping.cpp:368-383; this re-ranks the k that exist)
    std::vector<float> d0(k);
    std::vector<long> l0(k);
    search(k, query, d0.data(), l0.data());
    std::vector<SearchInfo_t> ranked;
    for (size_t i = 0; i < k; i++)
        if (l0[i] >= 0) {
            SearchInfo_t s;
            s.label = l0[i];
            s.distance = getL2Distance(query, path_base, d, l0[i], base_vec);
            ranked.push_back(s);
        }
    std::sort(ranked.begin(), ranked.end(), cmp);
    for (size_t i = 0; i < k; i++) {
        distances[i] = i < ranked.size() ? ranked[i].distance : FLT_MAX;
        labels[i] = i < ranked.size() ? ranked[i].label : -1;
    }
}

void IndexIVF_HNSW_Grouping::write(const char *path_index, bool do_trunc)
{
    std::ofstream out(path_index, do_trunc ? (std::ios::binary | std::ios::trunc) : std::ios::binary);
    write_variable(out, d);
    write_variable(out, nc);
    write_variable(out, nsubc);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, ids[c]);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, codes[c]);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, norm_codes[c]);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, nn_centroid_idxs[c]);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, subgroup_sizes[c]);
    write_vector(out, alphas);
    write_vector(out, centroid_norms);
    for (size_t c = 0; c < nc; c++)
        write_vector(out, inter_centroid_dists[c]);
}

void IndexIVF_HNSW_Grouping::write(const char *path_index) { this->write(path_index, false); }

void IndexIVF_HNSW_Grouping::read(const char *path_index)
{
    std::ifstream in(path_index, std::ios::binary);
    if (!in)
        throw std::runtime_error(std::string("cannot open ") + path_index);
    read_variable(in, d);
    read_variable(in, nc);
    read_variable(in, nsubc);
    ids.resize(nc);
    codes.resize(nc);
    norm_codes.resize(nc);
    nn_centroid_idxs.resize(nc);
    subgroup_sizes.resize(nc);
    inter_centroid_dists.resize(nc);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, ids[c]);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, codes[c]);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, norm_codes[c]);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, nn_centroid_idxs[c]);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, subgroup_sizes[c]);
    read_vector(in, alphas);
    read_vector(in, centroid_norms);
    for (size_t c = 0; c < nc; c++)
        read_vector(in, inter_centroid_dists[c]);
    if (!in)
        throw std::runtime_error(std::string("truncated index file ") + path_index);
    query_centroid_dists.assign(nc, 0.f);
    device_dirty_ = true;
}

void IndexIVF_HNSW_Grouping::compute_inter_centroid_dists()
{
    for (size_t i = 0; i < nc; i++) {
        inter_centroid_dists[i].resize(nsubc);
        const float *c = quantizer->getDataByInternalId((idx_t)i);
        for (size_t s = 0; s < nsubc; s++)
            inter_centroid_dists[i][s] = fvec_L2sqr(quantizer->getDataByInternalId(nn_centroid_idxs[i][s]), c, d);
    }
    device_dirty_ = true;
}

void IndexIVF_HNSW_Grouping::dump_inter_centroid_dists(char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp) {
        std::cout << "Failed to open file: " << path << std::endl;
        return;
    }
    for (size_t i = 0; i < nc; i++)
        for (size_t s = 0; s < inter_centroid_dists[i].size(); s++)
            fprintf(fp, "centroid %zu sub %zu (%u): %f\n", i, s, nn_centroid_idxs[i][s], inter_centroid_dists[i][s]);
    fclose(fp);
}

// ---- index construction: not part of the search path ---------------------------------
static void not_built(const char *what)
{
    throw std::runtime_error(std::string("IndexIVF_HNSW_Grouping::") + what +
                             ": Grouping index construction is outside the MI355X search path and not implemented "
                             "yet (SURVEY.md 8f); load a built index with read()");
}

void IndexIVF_HNSW_Grouping::add_group(size_t, size_t, const float *, const idx_t *) { not_built("add_group"); }
void IndexIVF_HNSW_Grouping::train_pq(size_t, const float *) { not_built("train_pq"); }
void IndexIVF_HNSW_Grouping::compute_residuals(size_t, const float *, float *, const float *, const idx_t *)
{
    not_built("compute_residuals");
}
void IndexIVF_HNSW_Grouping::reconstruct(size_t, float *, const float *, const float *, const idx_t *)
{
    not_built("reconstruct");
}
void IndexIVF_HNSW_Grouping::compute_subcentroid_idxs(idx_t *, const float *, const float *, size_t)
{
    not_built("compute_subcentroid_idxs");
}
float IndexIVF_HNSW_Grouping::compute_alpha(const float *, const float *, const float *, const float *, size_t)
{
    not_built("compute_alpha");
    return 0.f;
}

} // namespace ivfhnsw
