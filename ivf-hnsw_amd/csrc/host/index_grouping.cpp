// ivfhnsw::IndexIVF_HNSW_Grouping over the MI355X C ABI (include/ivf-hnsw/IndexIVF_HNSW_Grouping.h).
// search runs on the device (plan_grouping_kernel + the ADC scan); read/write keep the reference's Grouping
// .index layout (IndexIVF_HNSW_Grouping.cpp:397-483).  add_group / train_pq (index construction) are outside
// the search path; add_group runs on the device, train_pq (code book training) on the host.
#include <ivf-hnsw/IndexIVF_HNSW_Grouping.h>

#include <ivfhnsw_hip.h>

#include <algorithm>
#include <iostream>
#include <map>
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
    for (size_t r = 0; r < nshards(); r++)
        if (ivfhnsw_gpu_upload_grouping(shard(r), nsubc, alphas.data(), nn.data(), sz.data(), icd.data()))
            throw std::runtime_error(std::string("ivfhnsw_gpu_upload_grouping: ") + ivfhnsw_gpu_last_error());
    device_dirty_ = false;
}

void IndexIVF_HNSW_Grouping::search_batch(size_t nq, size_t k, const float *x, float *distances, long *labels)
{
    ensure_device();
    device_search(nq, k, x, nullptr, nullptr, nprobe, max_codes, do_pruning, distances, labels);
}

void IndexIVF_HNSW_Grouping::search(size_t k, const float *x, float *distances, long *labels)
{
#ifdef TRACE_CENTROIDS
    trace_query_centroid_dists.clear();
    trace_centroid_idxs.clear();
#endif
    ensure_device();
    ensure_latency_walk();
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

// ---- index construction -------------------------------------------------------------------
// add_group: everything up to the distribution loops (Grouping.cpp:43-125) is one device call; the sub-group
// layout of the list (:127-155) stays here.
void IndexIVF_HNSW_Grouping::add_group(size_t centroid_idx, size_t group_size, const float *data, const idx_t *idxs)
{
    ensure_encoder();
    const uint32_t cidx = (uint32_t)centroid_idx;
    const uint64_t off[2] = {0, (uint64_t)group_size};
    nn_centroid_idxs[centroid_idx].resize(nsubc);
    std::vector<idx_t> sub(group_size);
    std::vector<uint8_t> xcodes(group_size * code_size), xnorm(group_size);
    float alpha = alphas[centroid_idx];
    if (ivfhnsw_gpu_encode_groups(gpu_, 1, nsubc, &cidx, off, data, quantizer->efSearch,
                                  nn_centroid_idxs[centroid_idx].data(), &alpha, sub.data(), xcodes.data(), xnorm.data()))
        throw std::runtime_error(std::string("ivfhnsw_gpu_encode_groups: ") + ivfhnsw_gpu_last_error());
    if (group_size == 0)
        return; // :63-64: neighbours recorded, nothing else
    alphas[centroid_idx] = alpha;
    std::vector<std::vector<size_t>> members(nsubc); // arrival order inside every sub-group
    for (size_t i = 0; i < group_size; i++)
        members[sub[i]].push_back(i);
    for (size_t s = 0; s < nsubc; s++) {
        subgroup_sizes[centroid_idx].push_back((idx_t)members[s].size());
        for (size_t i : members[s]) {
            ids[centroid_idx].push_back(idxs[i]);
            codes[centroid_idx].insert(codes[centroid_idx].end(), xcodes.begin() + i * code_size,
                                       xcodes.begin() + (i + 1) * code_size);
            norm_codes[centroid_idx].push_back(xnorm[i]);
        }
    }
    device_dirty_ = true;
}

// train_pq (Grouping.cpp:486-618): residuals against the SUB-centroids of a training sample, then the two code
// books.  Host code (code book training is not on the device); the assignment of the sample runs on the device.
void IndexIVF_HNSW_Grouping::train_pq(size_t n, const float *x)
{
    std::vector<idx_t> assigned(n);
    assign(n, x, assigned.data());
    std::map<idx_t, std::vector<size_t>> groups; // ordered, so that a run is reproducible (the reference's is not)
    for (size_t i = 0; i < n; i++)
        groups[assigned[i]].push_back(i);

    std::vector<float> train_subcentroids, train_residuals;
    train_subcentroids.reserve(n * d);
    train_residuals.reserve(n * d);
    std::vector<size_t> group_sizes;
    std::cout << "Training Residual PQ codebook " << std::endl;
    for (const auto &grp : groups) {
        const idx_t centroid_idx = grp.first;
        const float *centroid = quantizer->getDataByInternalId(centroid_idx);
        const size_t group_size = grp.second.size();
        std::vector<float> data(group_size * d);
        for (size_t i = 0; i < group_size; i++)
            std::memcpy(&data[i * d], x + grp.second[i] * d, d * sizeof(float));
        // :507-517 neighbour centroids, nearest dropped
        std::vector<idx_t> nn(nsubc);
        std::vector<float> cv_norms(nsubc);
        auto raw = quantizer->searchKnn(centroid, nsubc + 1);
        while (raw.size() > 1) {
            cv_norms[raw.size() - 2] = raw.top().first;
            nn[raw.size() - 2] = raw.top().second;
            raw.pop();
        }
        std::vector<float> cvs(nsubc * d), subc(nsubc * d);
        for (size_t s2 = 0; s2 < nsubc; s2++)
            faiss::fvec_madd(d, quantizer->getDataByInternalId(nn[s2]), -1.f, centroid, &cvs[s2 * d]);
        const float alpha = compute_alpha(cvs.data(), data.data(), centroid, cv_norms.data(), group_size);
        for (size_t s2 = 0; s2 < nsubc; s2++)
            faiss::fvec_madd(d, centroid, alpha, &cvs[s2 * d], &subc[s2 * d]);
        std::vector<idx_t> sidx(group_size);
        compute_subcentroid_idxs(sidx.data(), subc.data(), data.data(), group_size);
        std::vector<float> res(group_size * d);
        compute_residuals(group_size, data.data(), res.data(), subc.data(), sidx.data());
        for (size_t i = 0; i < group_size; i++) {
            train_subcentroids.insert(train_subcentroids.end(), &subc[(size_t)sidx[i] * d], &subc[(size_t)sidx[i] * d] + d);
            train_residuals.insert(train_residuals.end(), &res[i * d], &res[i * d] + d);
        }
        group_sizes.push_back(group_size);
    }
    if (do_opq) { // :555-569
        faiss::OPQMatrix *matrix = new faiss::OPQMatrix((int)d, (int)pq->M);
        std::cout << "Training OPQ Matrix" << std::endl;
        matrix->verbose = true;
        matrix->max_train_points = n;
        matrix->niter = 100;
        try {
            matrix->train((long)n, train_residuals.data());
        } catch (...) {
            delete matrix;
            throw;
        }
        opq_matrix = matrix;
        std::vector<float> copy(train_residuals);
        opq_matrix->apply_noalloc((long)n, copy.data(), train_residuals.data());
    }
    printf("Training %zdx%zd PQ on %zd vectors in %zdD\n", pq->M, pq->ksub, train_residuals.size() / d, d);
    pq->verbose = true;
    pq->train((int)n, train_residuals.data());

    // :577-617 norm code book from the reconstructed training points
    std::cout << "Training Norm PQ codebook " << std::endl;
    std::vector<float> train_norms;
    train_norms.reserve(n);
    size_t off = 0;
    for (const size_t group_size : group_sizes) {
        const float *res = train_residuals.data() + off * d, *sub = train_subcentroids.data() + off * d;
        std::vector<uint8_t> xcodes(group_size * code_size);
        pq->compute_codes(res, xcodes.data(), group_size);
        std::vector<float> dec(group_size * d);
        pq->decode(xcodes.data(), dec.data(), group_size);
        if (do_opq) {
            std::vector<float> copy(dec);
            opq_matrix->transform_transpose((long)group_size, copy.data(), dec.data());
        }
        std::vector<float> rec(group_size * d), norms(group_size);
        for (size_t i = 0; i < group_size; i++)
            faiss::fvec_madd(d, &dec[i * d], 1.f, sub + i * d, &rec[i * d]);
        faiss::fvec_norms_L2sqr(norms.data(), rec.data(), d, group_size);
        train_norms.insert(train_norms.end(), norms.begin(), norms.end());
        off += group_size;
    }
    printf("Training %zdx%zd PQ on %zd vectors in 1D\n", norm_pq->M, norm_pq->ksub, train_norms.size());
    norm_pq->verbose = true;
    norm_pq->train((int)n, train_norms.data());
    device_dirty_ = true;
}

// The per-group helpers of the reference (:655-733) as host functions, for callers that use them directly; add_group
// does not go through them.
void IndexIVF_HNSW_Grouping::compute_residuals(size_t n, const float *x, float *residuals, const float *subcentroids,
                                               const idx_t *keys)
{
    for (size_t i = 0; i < n; i++)
        faiss::fvec_madd(d, x + i * d, -1.f, subcentroids + (size_t)keys[i] * d, residuals + i * d);
}

void IndexIVF_HNSW_Grouping::reconstruct(size_t n, float *x, const float *decoded_residuals, const float *subcentroids,
                                         const idx_t *keys)
{
    for (size_t i = 0; i < n; i++)
        faiss::fvec_madd(d, decoded_residuals + i * d, 1.f, subcentroids + (size_t)keys[i] * d, x + i * d);
}

void IndexIVF_HNSW_Grouping::compute_subcentroid_idxs(idx_t *subcentroid_idxs, const float *subcentroids,
                                                      const float *x, size_t group_size)
{
    for (size_t i = 0; i < group_size; i++) {
        float min_dist = 0.f;
        long min_idx = -1;
        for (size_t s = 0; s < nsubc; s++) {
            const float dist = fvec_L2sqr(subcentroids + s * d, x + i * d, d);
            if (min_idx == -1 || dist < min_dist) {
                min_dist = dist;
                min_idx = (long)s;
            }
        }
        subcentroid_idxs[i] = (idx_t)min_idx;
    }
}

float IndexIVF_HNSW_Grouping::compute_alpha(const float *centroid_vectors, const float *points, const float *centroid,
                                            const float *centroid_vector_norms_L2sqr, size_t group_size)
{
    float group_numerator = 0.f, group_denominator = 0.f;
    std::vector<float> pv(d), sub(d);
    for (size_t i = 0; i < group_size; i++) {
        const float *point = points + i * d;
        faiss::fvec_madd(d, point, -1.f, centroid, pv.data());
        bool have = false;
        float bneg = 0.f, bnum = 0.f, bden = 0.f;
        for (size_t s = 0; s < nsubc; s++) {
            const float *cv = centroid_vectors + s * d;
            float numerator = faiss::fvec_inner_product(cv, pv.data(), d);
            numerator = numerator > 0 ? numerator : 0.f;
            const float denominator = centroid_vector_norms_L2sqr[s];
            faiss::fvec_madd(d, centroid, numerator / denominator, cv, sub.data());
            const float neg = -fvec_L2sqr(point, sub.data(), d);
            // top of the reference's max-heap of pair<-dist, pair<numerator, denominator>>; a NaN never wins
            bool better;
            if (!have)
                better = true;
            else if (neg != neg)
                better = false;
            else if (bneg != bneg)
                better = true;
            else
                better = bneg < neg || (!(neg < bneg) && (bnum < numerator || (!(numerator < bnum) && bden < denominator)));
            if (better) {
                have = true;
                bneg = neg;
                bnum = numerator;
                bden = denominator;
            }
        }
        group_numerator += bnum;
        group_denominator += bden;
    }
    return group_denominator > 0 ? group_numerator / group_denominator : 0.f;
}

} // namespace ivfhnsw
