// ivfhnsw_gpu_build_graph: hnswlib's addPoint loop (hnswlib/hnswalg.cpp:212-225) for all nodes at once.
//
// The one deviation from the reference: a new node's link candidates are its EXACT ncand nearest among the nodes
// inserted before it -- one triangular sweep of the MFMA neighbour-table kernel (kernels_knn.hip, IVFHNSW_KNN_EARLIER)
// -- instead of the efConstruction results of a greedy search of the graph built so far (hnswalg.cpp:221).  With that,
// a node's forward links no longer depend on the state of the graph, so the serial insertion loop unrolls exactly:
//   A. forward links of every node c, independently: getNeighborsByHeuristic over its candidates down to M
//      (hnswalg.cpp:110-146; distances by fstdistfunc's order, :326-357), stored farthest first (:153-170);
//   B. reverse lists: for every node t the later nodes c that chose t, ascending c (= the order the serial loop meets them);
//   C. the fold of mutuallyConnectNewElement's second half over t's reverse list, independently per t: append while t has
//      room, else shrink t's maxM + 1 candidates with the same heuristic (:171-209).
// The result is what the serial loop leaves, link for link (tests/test_gpu_graph_build.py against the oracle's serial
// restatement).  A and C run on host threads (they are short dependent chains of 512-byte row reads: 99.9 % of the
// arithmetic is the neighbour table, on the device).
#include "../../include/ivfhnsw_hip.h"

#include <algorithm>
#include <new>
#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

int ivfhnsw_gpu_fail_msg(int code, const char *msg); // capi.cpp

namespace {

// hnswalg.cpp:326-357 / utils.cpp:22-52: eight accumulators over blocks of 16 floats, unfused sub / mul / add, the eight
// sums added left to right; dimensions beyond a multiple of 16 are ignored (as the reference does)
inline float l2_ref(const float *x, const float *y, size_t d)
{
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t nblk = d >> 4;
    for (size_t b = 0; b < nblk; b++)
        for (int h = 0; h < 2; h++)
            for (int l = 0; l < 8; l++) {
                const float df = x[b * 16 + h * 8 + l] - y[b * 16 + h * 8 + l];
                const float sq = df * df;
                s[l] = s[l] + sq;
            }
    float r = s[0] + s[1];
    r = r + s[2];
    r = r + s[3];
    r = r + s[4];
    r = r + s[5];
    r = r + s[6];
    r = r + s[7];
    return r;
}

struct Cand {
    float dist;
    uint32_t id;
};

// getNeighborsByHeuristic (hnswalg.cpp:110-146) on `c` (any order in); out: the kept elements in the order
// mutuallyConnectNewElement stores them (pops of a max-heap of (dist, id): farthest first, larger id first among equals)
void heuristic(const float *vec, size_t d, std::vector<Cand> &c, size_t NN)
{
    if (c.size() >= NN) {
        // resultSet pops by (-dist, id) descending: smallest distance first, larger id first among equal distances
        std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.dist < b.dist || (a.dist == b.dist && a.id > b.id); });
        size_t nk = 0;
        for (size_t i = 0; i < c.size() && nk < NN; i++) {
            bool good = true;
            for (size_t j = 0; j < nk; j++)
                if (l2_ref(vec + (size_t)c[j].id * d, vec + (size_t)c[i].id * d, d) < c[i].dist) {
                    good = false;
                    break;
                }
            if (good)
                c[nk++] = c[i];
        }
        c.resize(nk);
    }
    std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.dist > b.dist || (a.dist == b.dist && a.id > b.id); });
}

template <class F> void parallel_for(size_t n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : nt > 64 ? 64 : nt;
    if (n < 4096)
        nt = 1;
    std::vector<std::thread> th;
    const size_t chunk = 256;
    std::vector<size_t> next(1, 0);
    static_assert(sizeof(size_t) == 8, "LP64");
    auto *counter = reinterpret_cast<unsigned long long *>(next.data());
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            (void)t;
            for (;;) {
                const size_t b = __atomic_fetch_add(counter, (unsigned long long)chunk, __ATOMIC_RELAXED);
                if (b >= n)
                    break;
                const size_t e = b + chunk < n ? b + chunk : n;
                for (size_t i = b; i < e; i++)
                    f(i);
            }
        });
    for (auto &x : th)
        x.join();
}

} // namespace

extern "C" int ivfhnsw_gpu_build_graph(ivfhnsw_gpu *h, size_t n, size_t d, const float *vectors, size_t M, size_t maxM,
                                       size_t ncand, uint8_t *out_counts, uint32_t *out_links)
try {
    if (!h || !vectors || !out_counts || !out_links)
        return ivfhnsw_gpu_fail_msg(IVFHNSW_ERR_INVALID, "build_graph: null argument");
    if (M < 1 || M > maxM || maxM > 64 || ncand < M || ncand > 80 || n >= 0xffffffffull)
        return ivfhnsw_gpu_fail_msg(IVFHNSW_ERR_INVALID, "build_graph: need 1 <= M <= maxM <= 64, M <= ncand <= 80, n < 2^32");
    if (d % 16 != 0)
        return ivfhnsw_gpu_fail_msg(IVFHNSW_ERR_INVALID, "build_graph: d must be a multiple of 16 (the reference's distance "
                                                         "ignores the dims beyond one, hnswalg.cpp:330; so does upload_quantizer)");
    std::memset(out_counts, 0, n);
    std::memset(out_links, 0, n * maxM * sizeof(uint32_t));
    if (n <= 1)
        return IVFHNSW_OK;
    // the exact candidates: row c against rows 0..c-1
    std::vector<uint32_t> table(n * ncand);
    int rc = ivfhnsw_gpu_knn(h, 0, n, d, nullptr, vectors, ncand, IVFHNSW_KNN_EARLIER, table.data(), nullptr);
    if (rc)
        return rc;
    // A. forward links
    std::vector<uint8_t> fcnt(n, 0);
    std::vector<uint32_t> fwd(n * M);
    parallel_for(n, [&](size_t c) {
        if (c == 0)
            return;
        std::vector<Cand> cand;
        cand.reserve(ncand);
        const uint32_t *row = table.data() + c * ncand;
        for (size_t i = 0; i < ncand && row[i] != 0xffffffffu; i++)
            cand.push_back({l2_ref(vectors + c * d, vectors + (size_t)row[i] * d, d), row[i]});
        heuristic(vectors, d, cand, M);
        // (fewer than M candidates: the reference keeps them all, unpruned -- hnswalg.cpp:112-113)
        fcnt[c] = (uint8_t)cand.size();
        for (size_t i = 0; i < cand.size(); i++)
            fwd[c * M + i] = cand[i].id;
    });
    // B. reverse lists, ascending in the choosing node
    std::vector<uint64_t> roff(n + 1, 0);
    for (size_t c = 1; c < n; c++)
        for (size_t i = 0; i < fcnt[c]; i++)
            roff[fwd[c * M + i] + 1]++;
    for (size_t t = 0; t < n; t++)
        roff[t + 1] += roff[t];
    std::vector<uint32_t> rev(roff[n]);
    {
        std::vector<uint64_t> at(roff.begin(), roff.end() - 1);
        for (size_t c = 1; c < n; c++)
            for (size_t i = 0; i < fcnt[c]; i++)
                rev[at[fwd[c * M + i]]++] = (uint32_t)c;
    }
    // C. every node's own insertion, then the later nodes that chose it, in their order
    parallel_for(n, [&](size_t t) {
        uint32_t *data = out_links + t * maxM;
        size_t cnt = fcnt[t];
        for (size_t i = 0; i < cnt; i++)
            data[i] = fwd[t * M + i];
        std::vector<Cand> cand;
        for (uint64_t r = roff[t]; r < roff[t + 1]; r++) {
            const uint32_t c = rev[r];
            if (cnt < maxM) {
                data[cnt++] = c;
                continue;
            }
            cand.clear();
            cand.push_back({l2_ref(vectors + (size_t)c * d, vectors + t * d, d), c});
            for (size_t j = 0; j < cnt; j++)
                cand.push_back({l2_ref(vectors + (size_t)data[j] * d, vectors + t * d, d), data[j]});
            heuristic(vectors, d, cand, maxM);
            cnt = cand.size();
            for (size_t j = 0; j < cnt; j++)
                data[j] = cand[j].id;
        }
        for (size_t j = cnt; j < maxM; j++)
            data[j] = 0;
        out_counts[t] = (uint8_t)cnt;
    });
    return IVFHNSW_OK;
} catch (const std::bad_alloc &) {
    return ivfhnsw_gpu_fail_msg(IVFHNSW_ERR_NOMEM, "ivfhnsw_gpu_build_graph: host allocation failed");
}
