// Test driver for the host-side class surface (include/ivf-hnsw/*.h, libivfhnsw.so).  Sub-commands are
// exercised by tests/test_host_library.py; the file-format ones need no GPU.
//
//   index_roundtrip ivf|grouping d nc code_size nsubc in.index out.index
//   hnsw_roundtrip info data edges out_info out_edges
//   hnsw_build data.fvecs n d M efConstruction out_info out_edges
//   hnsw_search info data edges queries.fvecs nq ef k out.bin        (host walk: ids u32[nq*k], dists f32[nq*k])
//   pq_roundtrip in out | vt_roundtrip in out
//   search ivf|grouping d nc code_size nsubc centroids info edges pq norm_pq opq|- index queries.fvecs nq k nprobe
//          max_codes efSearch pruning out.bin                        (GPU; single-query loop AND batch)
//   siblings <same arguments>    (GPU; search_debug, search_enn, search2, search2m of IndexIVF_HNSW.cpp:328-534)
#include <ivf-hnsw/IndexIVF_HNSW_Grouping.h>
#include <ivf-hnsw/hnswalg.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

using namespace ivfhnsw;

static void dump(const char *path, const void *a, size_t na, const void *b, size_t nb)
{
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(a, 1, na, f) != na || fwrite(b, 1, nb, f) != nb)
        throw std::runtime_error(std::string("cannot write ") + path);
    fclose(f);
}

static int run(int argc, char **argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "index_roundtrip" && argc == 9) {
        const bool grp = !strcmp(argv[2], "grouping");
        const size_t d = atol(argv[3]), nc = atol(argv[4]), cs = atol(argv[5]), nsubc = atol(argv[6]);
        if (grp) {
            IndexIVF_HNSW_Grouping ix(d, nc, cs, 8, nsubc);
            ix.read(argv[7]);
            ix.write(argv[8]);
        } else {
            IndexIVF_HNSW ix(d, nc, cs, 8);
            ix.read(argv[7]);
            ix.write(argv[8]);
        }
        return 0;
    }
    if (cmd == "hnsw_roundtrip" && argc == 7) {
        hnswlib::HierarchicalNSW g(argv[2], argv[3], argv[4]);
        g.SaveInfo(argv[5]);
        g.SaveEdges(argv[6]);
        return 0;
    }
    if (cmd == "hnsw_build" && argc == 9) {
        const size_t n = atol(argv[3]), d = atol(argv[4]), M = atol(argv[5]), efc = atol(argv[6]);
        hnswlib::HierarchicalNSW g(d, n, M, 2 * M, efc);
        std::ifstream in(argv[2], std::ios::binary);
        std::vector<float> v(d);
        for (size_t i = 0; i < n; i++) {
            readXvec<float>(in, v.data(), d);
            g.addPoint(v.data());
        }
        g.SaveInfo(argv[7]);
        g.SaveEdges(argv[8]);
        return 0;
    }
    if (cmd == "build_quantizer" && argc == 9) {
        // IndexIVF_HNSW::build_quantizer (IndexIVF_HNSW.cpp:34-66) as the drivers call it: centroids file -> info + edges files
        const size_t n = atol(argv[3]), d = atol(argv[4]), M = atol(argv[5]), efc = atol(argv[6]);
        ivfhnsw::IndexIVF_HNSW index(d, n, 16, 8);
        index.build_quantizer(argv[2], argv[7], argv[8], M, efc);
        return 0;
    }
    if (cmd == "hnsw_search" && argc == 10) {
        hnswlib::HierarchicalNSW g(argv[2], argv[3], argv[4]);
        const size_t nq = atol(argv[6]), ef = atol(argv[7]), k = atol(argv[8]);
        g.efSearch = ef;
        std::vector<float> q(nq * g.d_);
        std::ifstream in(argv[5], std::ios::binary);
        readXvec<float>(in, q.data(), g.d_, nq);
        std::vector<uint32_t> ids(nq * k, 0xffffffffu);
        std::vector<float> dist(nq * k, 0.f);
        for (size_t i = 0; i < nq; i++) {
            auto res = g.searchKnn(q.data() + i * g.d_, k);
            for (size_t j = res.size(); j-- > 0;) {
                ids[i * k + j] = res.top().second;
                dist[i * k + j] = res.top().first;
                res.pop();
            }
        }
        dump(argv[9], ids.data(), ids.size() * 4, dist.data(), dist.size() * 4);
        return 0;
    }
    if (cmd == "pq_roundtrip" && argc == 4) {
        faiss::ProductQuantizer *pq = faiss::read_ProductQuantizer(argv[2]);
        faiss::write_ProductQuantizer(pq, argv[3]);
        delete pq;
        return 0;
    }
    if (cmd == "vt_roundtrip" && argc == 4) {
        faiss::VectorTransform *vt = faiss::read_VectorTransform(argv[2]);
        faiss::write_VectorTransform(vt, argv[3]);
        delete vt;
        return 0;
    }
    if (cmd == "add_batch" && argc == 16) {
        // the construction sequence of the reference's drivers (tests/test_ivfhnsw_sift1b.cpp:47-67,100-160):
        // build/load the quantizer, load pq / norm_pq / opq, add the base vectors batch by batch, write the index
        const size_t d = atol(argv[2]), nc = atol(argv[3]), cs = atol(argv[4]);
        const char *centroids = argv[5], *info = argv[6], *edges = argv[7], *ppq = argv[8], *pnorm = argv[9],
                   *popq = argv[10], *pbase = argv[11], *pidx = argv[12], *pindex = argv[15];
        const size_t n = atol(argv[13]), batch = atol(argv[14]);
        IndexIVF_HNSW *index = new IndexIVF_HNSW(d, nc, cs, 8);
        index->build_quantizer(centroids, info, edges, 16, 500);
        index->do_opq = strcmp(popq, "-") != 0;
        delete index->pq;
        index->pq = faiss::read_ProductQuantizer(ppq);
        if (index->do_opq)
            index->opq_matrix = dynamic_cast<faiss::LinearTransform *>(faiss::read_VectorTransform(popq));
        delete index->norm_pq;
        index->norm_pq = faiss::read_ProductQuantizer(pnorm);
        index->quantizer->efSearch = 40;
        std::vector<float> x(n * d);
        {
            std::ifstream in(pbase, std::ios::binary);
            readXvec<float>(in, x.data(), d, n);
        }
        std::vector<uint32_t> pre;
        if (strcmp(pidx, "-") != 0) { // precomputed assignments, one uint32 per vector
            pre.resize(n);
            std::ifstream in(pidx, std::ios::binary);
            in.read((char *)pre.data(), n * sizeof(uint32_t));
        }
        std::vector<uint32_t> xids(n);
        for (size_t i = 0; i < n; i++)
            xids[i] = (uint32_t)(1000 + i);
        for (size_t i0 = 0; i0 < n; i0 += batch) {
            const size_t m = std::min(batch, n - i0);
            index->add_batch(m, x.data() + i0 * d, xids.data() + i0, pre.empty() ? nullptr : pre.data() + i0);
        }
        index->compute_centroid_norms();
        index->write(pindex);
        delete index;
        return 0;
    }
    if (cmd == "add_group" && argc == 16) {
        // Grouping construction as the reference's driver does it (tests/test_ivfhnsw_grouping_sift1b.cpp:
        // one add_group call per centroid with that centroid's points, then the two table passes and write)
        const size_t d = atol(argv[2]), nc = atol(argv[3]), cs = atol(argv[4]), nsubc = atol(argv[5]);
        const char *centroids = argv[6], *info = argv[7], *edges = argv[8], *ppq = argv[9], *pnorm = argv[10],
                   *popq = argv[11], *pbase = argv[12], *pidx = argv[13], *pindex = argv[15];
        const size_t n = atol(argv[14]);
        IndexIVF_HNSW_Grouping *index = new IndexIVF_HNSW_Grouping(d, nc, cs, 8, nsubc);
        index->build_quantizer(centroids, info, edges, 16, 500);
        index->do_opq = strcmp(popq, "-") != 0;
        delete index->pq;
        index->pq = faiss::read_ProductQuantizer(ppq);
        if (index->do_opq)
            index->opq_matrix = dynamic_cast<faiss::LinearTransform *>(faiss::read_VectorTransform(popq));
        delete index->norm_pq;
        index->norm_pq = faiss::read_ProductQuantizer(pnorm);
        index->quantizer->efSearch = 40;
        std::vector<float> x(n * d);
        {
            std::ifstream in(pbase, std::ios::binary);
            readXvec<float>(in, x.data(), d, n);
        }
        std::vector<uint32_t> pre(n);
        {
            std::ifstream in(pidx, std::ios::binary);
            in.read((char *)pre.data(), n * sizeof(uint32_t));
        }
        for (size_t c = 0; c < nc; c++) {
            std::vector<float> data;
            std::vector<uint32_t> gids;
            for (size_t i = 0; i < n; i++)
                if (pre[i] == c) {
                    data.insert(data.end(), x.begin() + i * d, x.begin() + (i + 1) * d);
                    gids.push_back((uint32_t)(1000 + i));
                }
            index->add_group(c, gids.size(), data.data(), gids.data());
        }
        index->compute_centroid_norms();
        index->compute_inter_centroid_dists();
        index->write(pindex);
        delete index;
        return 0;
    }
    if (cmd == "opq_train" && argc == 8) {
        // OPQMatrix::train on the rows of an .fvecs file; prints the quantisation error of a PQ trained on the raw
        // points and of one trained on the rotated points, writes the matrix
        const size_t d = atol(argv[2]), M = atol(argv[3]), n = atol(argv[4]);
        const int niter = atoi(argv[6]);
        std::vector<float> x(n * d);
        {
            std::ifstream in(argv[5], std::ios::binary);
            readXvec<float>(in, x.data(), d, n);
        }
        auto pq_error = [&](const float *pts) {
            faiss::ProductQuantizer pq(d, M, 8);
            pq.train((int)n, pts);
            std::vector<uint8_t> codes(n * pq.code_size);
            pq.compute_codes(pts, codes.data(), n);
            std::vector<float> dec(n * d);
            pq.decode(codes.data(), dec.data(), n);
            double e = 0;
            for (size_t i = 0; i < n * d; i++)
                e += (double)(pts[i] - dec[i]) * (pts[i] - dec[i]);
            return e / (double)n;
        };
        faiss::OPQMatrix opq((int)d, (int)M);
        opq.niter = niter;
        opq.niter_pq_0 = 10;
        opq.niter_pq = 3;
        opq.max_train_points = n;
        opq.train((long)n, x.data());
        std::vector<float> xr(n * d);
        opq.apply_noalloc((long)n, x.data(), xr.data());
        printf("err_plain %.9g\nerr_opq %.9g\n", pq_error(x.data()), pq_error(xr.data()));
        faiss::write_VectorTransform(&opq, argv[7]);
        return 0;
    }
    if (cmd == "grouping_train" && argc == 13) {
        // IndexIVF_HNSW_Grouping::train_pq on a training sample, code books written in faiss's formats
        const size_t d = atol(argv[2]), nc = atol(argv[3]), cs = atol(argv[4]), nsubc = atol(argv[5]);
        const char *centroids = argv[6], *info = argv[7], *edges = argv[8], *plearn = argv[9];
        const size_t n = atol(argv[10]);
        IndexIVF_HNSW_Grouping *index = new IndexIVF_HNSW_Grouping(d, nc, cs, 8, nsubc);
        index->build_quantizer(centroids, info, edges, 16, 500);
        index->quantizer->efSearch = 40;
        index->do_opq = false;
        std::vector<float> x(n * d);
        {
            std::ifstream in(plearn, std::ios::binary);
            readXvec<float>(in, x.data(), d, n);
        }
        index->train_pq(n, x.data());
        faiss::write_ProductQuantizer(index->pq, argv[11]);
        faiss::write_ProductQuantizer(index->norm_pq, argv[12]);
        delete index;
        return 0;
    }
    if ((cmd == "search" || cmd == "siblings") && argc == 22) {
        const bool grp = !strcmp(argv[2], "grouping");
        const size_t d = atol(argv[3]), nc = atol(argv[4]), cs = atol(argv[5]), nsubc = atol(argv[6]);
        const char *centroids = argv[7], *info = argv[8], *edges = argv[9], *ppq = argv[10], *pnorm = argv[11],
                   *popq = argv[12], *pindex = argv[13], *pq_ = argv[14];
        const size_t nq = atol(argv[15]), k = atol(argv[16]), nprobe = atol(argv[17]), max_codes = atol(argv[18]),
                     ef = atol(argv[19]);
        const bool pruning = atoi(argv[20]) != 0;
        // the load sequence of the reference's drivers (tests/test_ivfhnsw_sift1b.cpp:47-67,125-129,164-183)
        IndexIVF_HNSW_Grouping *gix = grp ? new IndexIVF_HNSW_Grouping(d, nc, cs, 8, nsubc) : nullptr;
        IndexIVF_HNSW *index = grp ? gix : new IndexIVF_HNSW(d, nc, cs, 8);
        index->build_quantizer(centroids, info, edges, 16, 500);
        index->do_opq = strcmp(popq, "-") != 0;
        delete index->pq;
        index->pq = faiss::read_ProductQuantizer(ppq);
        if (index->do_opq)
            index->opq_matrix = dynamic_cast<faiss::LinearTransform *>(faiss::read_VectorTransform(popq));
        delete index->norm_pq;
        index->norm_pq = faiss::read_ProductQuantizer(pnorm);
        index->read(pindex);
        if (index->do_opq)
            index->rotate_quantizer();
        index->nprobe = nprobe;
        index->max_codes = max_codes;
        index->quantizer->efSearch = ef;
        if (gix)
            gix->do_pruning = pruning;
        std::vector<float> q(nq * d);
        {
            std::ifstream in(pq_, std::ios::binary);
            readXvec<float>(in, q.data(), d, nq);
        }
        if (cmd == "search") {
            // (1) one query per call, as the drivers do; (2) the batched extension.  Both are written out.
            std::vector<float> dist(2 * nq * k);
            std::vector<long> lab(2 * nq * k);
            for (size_t i = 0; i < nq; i++)
                index->search(k, q.data() + i * d, dist.data() + i * k, lab.data() + i * k);
            index->search_batch(nq, k, q.data(), dist.data() + nq * k, lab.data() + nq * k);
            dump(argv[21], lab.data(), lab.size() * sizeof(long), dist.data(), dist.size() * sizeof(float));
        } else {
            // The sibling entry points of IndexIVF_HNSW (IndexIVF_HNSW.cpp:328-534), one query per call as a driver
            // would use them.  Output: labels [debug nq*k | enn nq | search2 nq*k | search2m nq*nprobe*k], then the
            // distances in the same layout, then the centroid search_enn returned (u32 [nq]) and the coarse stage
            // handed to search2 / search2m (ids u32 [nq*nprobe], dists f32 [nq*nprobe]).
            const size_t n_lab = nq * k + nq + nq * k + nq * nprobe * k;
            std::vector<long> lab(n_lab, -7);
            std::vector<float> dist(n_lab, -7.f);
            std::vector<uint32_t> enn_c(nq), cids(nq * nprobe);
            std::vector<float> cds(nq * nprobe);
            long *l_dbg = lab.data(), *l_enn = l_dbg + nq * k, *l_s2 = l_enn + nq, *l_s2m = l_s2 + nq * k;
            float *d_dbg = dist.data(), *d_enn = d_dbg + nq * k, *d_s2 = d_enn + nq, *d_s2m = d_s2 + nq * k;
            for (size_t i = 0; i < nq; i++) {
                const float *x = q.data() + i * d;
                index->search_debug(k, x, d_dbg + i * k, l_dbg + i * k);
                enn_c[i] = index->search_enn(x, d_enn + i, l_enn + i);
                // the caller's coarse stage of search2 / search2m: the host walk on the rotated query
                // (IndexIVF_HNSW.cpp:240-259 is what a driver repeats before calling search2)
                const float *xr = index->do_opq ? index->opq_matrix->apply(1, x) : x;
                auto coarse = index->quantizer->searchKnn(xr, nprobe);
                if (coarse.size() != nprobe)
                    throw std::runtime_error("host walk returned fewer than nprobe centroids");
                for (size_t j = nprobe; j-- > 0;) {
                    cds[i * nprobe + j] = coarse.top().first;
                    cids[i * nprobe + j] = coarse.top().second;
                    coarse.pop();
                }
                if (index->do_opq)
                    delete[] const_cast<float *>(xr);
                index->search2(k, x, d_s2 + i * k, l_s2 + i * k, cds.data() + i * nprobe, cids.data() + i * nprobe);
                std::vector<float *> dp(nprobe);
                std::vector<long *> lp(nprobe);
                for (size_t j = 0; j < nprobe; j++) {
                    dp[j] = d_s2m + (i * nprobe + j) * k;
                    lp[j] = l_s2m + (i * nprobe + j) * k;
                }
                index->search2m(k, x, dp.data(), lp.data(), cds.data() + i * nprobe, cids.data() + i * nprobe);
            }
            FILE *f = fopen(argv[21], "wb");
            if (!f)
                throw std::runtime_error("cannot write the result file");
            fwrite(lab.data(), sizeof(long), lab.size(), f);
            fwrite(dist.data(), sizeof(float), dist.size(), f);
            fwrite(enn_c.data(), 4, enn_c.size(), f);
            fwrite(cids.data(), 4, cids.size(), f);
            fwrite(cds.data(), 4, cds.size(), f);
            fclose(f);
        }
        delete index;
        return 0;
    }
    fprintf(stderr, "usage: see the header of tests/cpp/hostlib_tool.cpp (got %d args for '%s')\n", argc, cmd.c_str());
    return 2;
}

int main(int argc, char **argv)
{
    try {
        return run(argc, argv);
    } catch (const std::exception &e) {
        fprintf(stderr, "hostlib_tool: %s\n", e.what());
        return 1;
    }
}
