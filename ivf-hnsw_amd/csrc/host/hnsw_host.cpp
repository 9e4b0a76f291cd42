// Host-side coarse quantizer graph (include/hnswlib/hnswalg.h): storage, file formats, serial
// reference-order construction and a host walk for construction-side callers.  The search hot path walks the
// same graph on the device (kernels_hnsw.hip); results of both are identical by construction (same (dist, id)
// ordering, same float order), which tests/test_host_library.py checks.
//
// File formats (reference hnswlib/hnswalg.cpp:236-324):
//   info : size_t maxelements, uint32 enterpoint, size_t data_size, offset_data, size_data_per_element, M, maxM,
//          size_links_level0            (60 bytes, no padding)
//   edges: per node uint32 n, n x uint32
//   data : .fvecs (uint32 dim, dim floats per record)
#include <hnswlib/hnswalg.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <vector>

namespace hnswlib {

namespace {

typedef std::pair<float, idx_t> Cand;
typedef std::priority_queue<Cand> MaxQ;

template <typename T> void put(std::ostream &o, const T &v) { o.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
template <typename T> void get(std::istream &i, T &v) { i.read(reinterpret_cast<char *>(&v), sizeof(T)); }

} // namespace

HierarchicalNSW::HierarchicalNSW(size_t d, size_t maxelements, size_t M, size_t maxM, size_t efConstruction)
{
    efSearch = efConstruction;
    data_level0_memory_ = nullptr;
    d_ = d;
    data_size_ = d * sizeof(float);
    M_ = M;
    maxM_ = maxM;
    efConstruction_ = efConstruction;
    maxelements_ = maxelements;
    cur_element_count = 0;
    enterpoint_node = 0;
    visitedlistpool = nullptr;
    dist_calc = 0;
    size_links_level0 = maxM_ * sizeof(idx_t) + sizeof(uint8_t);
    offset_data = size_links_level0;
    size_data_per_element = size_links_level0 + data_size_;
    data_level0_memory_ = static_cast<char *>(calloc(maxelements_ ? maxelements_ : 1, size_data_per_element));
    if (!data_level0_memory_)
        throw std::bad_alloc();
    std::cout << "Size Mb: " << (maxelements_ * size_data_per_element) / (1000 * 1000) << std::endl;
    visitedlistpool = new VisitedListPool(1, maxelements_);
}

HierarchicalNSW::HierarchicalNSW(const std::string &infoLocation, const std::string &dataLocation,
                                 const std::string &edgeLocation)
{
    efSearch = 0;
    data_level0_memory_ = nullptr;
    size_data_per_element = size_links_level0 = offset_data = data_size_ = d_ = 0;
    M_ = maxM_ = efConstruction_ = maxelements_ = cur_element_count = 0;
    enterpoint_node = 0;
    visitedlistpool = nullptr;
    dist_calc = 0;
    LoadInfo(infoLocation);
    LoadData(dataLocation);
    LoadEdges(edgeLocation);
}

HierarchicalNSW::~HierarchicalNSW()
{
    free(data_level0_memory_);
    delete visitedlistpool;
}

// 8 accumulators over blocks of 16 floats, unfused multiply then add, accumulators summed left to right --
// the float order of the reference's AVX distance (hnswalg.cpp:326-357).  Dims past the last multiple of 16
// are ignored, as there.
float HierarchicalNSW::fstdistfunc(const float *x, const float *y)
{
    typedef float v8f __attribute__((vector_size(32), aligned(4)));
    v8f acc = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t nblk = d_ >> 4;
    for (size_t b = 0; b < 2 * nblk; b++) {
        const v8f diff = *reinterpret_cast<const v8f *>(x + 8 * b) - *reinterpret_cast<const v8f *>(y + 8 * b);
        acc = acc + diff * diff;
    }
    float r = acc[0] + acc[1];
    for (int l = 2; l < 8; l++)
        r = r + acc[l];
    return r;
}

MaxQ HierarchicalNSW::searchBaseLayer(const float *point, size_t ef)
{
    VisitedList *vl = visitedlistpool->getFreeVisitedList();
    vl_type *seen = vl->mass;
    const vl_type tag = vl->curV;

    MaxQ best;      // the ef closest so far, farthest on top
    MaxQ frontier;  // nodes to expand, keyed by negated distance
    const float d0 = fstdistfunc(point, getDataByInternalId(enterpoint_node));
    dist_calc++;
    best.emplace(d0, enterpoint_node);
    frontier.emplace(-d0, enterpoint_node);
    seen[enterpoint_node] = tag;
    float bound = d0;

    while (!frontier.empty()) {
        const Cand cur = frontier.top();
        if (-cur.first > bound)
            break;
        frontier.pop();
        const uint8_t *rec = get_linklist0(cur.second);
        const size_t n = rec[0];
        const idx_t *nb = reinterpret_cast<const idx_t *>(rec + 1);
        for (size_t j = 0; j < n; j++) {
            const idx_t t = nb[j];
            if (seen[t] == tag)
                continue;
            seen[t] = tag;
            const float dt = fstdistfunc(point, getDataByInternalId(t));
            dist_calc++;
            if (best.top().first > dt || best.size() < ef) {
                frontier.emplace(-dt, t);
                best.emplace(dt, t);
                if (best.size() > ef)
                    best.pop();
                bound = best.top().first;
            }
        }
    }
    visitedlistpool->releaseVisitedList(vl);
    return best;
}

MaxQ HierarchicalNSW::searchKnn(const float *query, size_t k)
{
    MaxQ res = searchBaseLayer(query, efSearch);
    while (res.size() > k)
        res.pop();
    return res;
}

// Keep at most NN candidates, closest first, dropping any that is closer to an already kept one than to the
// query (the diversification rule of hnswalg.cpp:112-146).
void HierarchicalNSW::getNeighborsByHeuristic(MaxQ &topResults, size_t NN)
{
    if (topResults.size() < NN)
        return;
    std::vector<Cand> byDist;
    byDist.reserve(topResults.size());
    while (!topResults.empty()) {
        byDist.push_back(topResults.top());
        topResults.pop();
    }
    // closest first; among equal distances the larger id first (pop order of a max-heap of (-dist, id))
    std::sort(byDist.begin(), byDist.end(), [](const Cand &a, const Cand &b) {
        return a.first < b.first || (a.first == b.first && a.second > b.second);
    });
    std::vector<Cand> kept;
    for (const Cand &c : byDist) {
        if (kept.size() >= NN)
            break;
        bool ok = true;
        for (const Cand &k : kept)
            if (fstdistfunc(getDataByInternalId(k.second), getDataByInternalId(c.second)) < c.first) {
                ok = false;
                break;
            }
        if (ok)
            kept.push_back(c);
    }
    for (const Cand &k : kept)
        topResults.emplace(k.first, k.second);
}

void HierarchicalNSW::mutuallyConnectNewElement(const float *, idx_t cur, MaxQ topResults)
{
    getNeighborsByHeuristic(topResults, M_);
    std::vector<idx_t> chosen;
    while (!topResults.empty()) {
        chosen.push_back(topResults.top().second);
        topResults.pop();
    }
    uint8_t *rec = get_linklist0(cur);
    rec[0] = (uint8_t)chosen.size();
    std::memcpy(rec + 1, chosen.data(), chosen.size() * sizeof(idx_t));

    for (idx_t o : chosen) {
        if (o == cur)
            throw std::runtime_error("Connection to the same element");
        uint8_t *orec = get_linklist0(o);
        idx_t *olinks = reinterpret_cast<idx_t *>(orec + 1);
        const size_t n = orec[0];
        if (n > maxM_)
            throw std::runtime_error("Bad sz_link_list_other");
        if (n < maxM_) {
            olinks[n] = cur;
            orec[0] = (uint8_t)(n + 1);
            continue;
        }
        // full: re-select among the old links plus the new node
        MaxQ cand;
        cand.emplace(fstdistfunc(getDataByInternalId(cur), getDataByInternalId(o)), cur);
        for (size_t j = 0; j < n; j++)
            cand.emplace(fstdistfunc(getDataByInternalId(olinks[j]), getDataByInternalId(o)), olinks[j]);
        getNeighborsByHeuristic(cand, maxM_);
        size_t w = 0;
        while (!cand.empty()) {
            olinks[w++] = cand.top().second;
            cand.pop();
        }
        orec[0] = (uint8_t)w;
    }
}

void HierarchicalNSW::addPoint(const float *point)
{
    if (cur_element_count >= maxelements_) {
        std::cout << "The number of elements exceeds the specified limit\n";
        throw std::runtime_error("The number of elements exceeds the specified limit");
    }
    const idx_t cur = (idx_t)cur_element_count++;
    std::memset(get_linklist0(cur), 0, size_data_per_element);
    std::memcpy(getDataByInternalId(cur), point, data_size_);
    if (cur == 0)
        return;
    mutuallyConnectNewElement(point, cur, searchBaseLayer(point, efConstruction_));
}

void HierarchicalNSW::SaveInfo(const std::string &location)
{
    std::cout << "Saving info to " << location << std::endl;
    std::ofstream out(location, std::ios::binary);
    put(out, maxelements_);
    put(out, enterpoint_node);
    put(out, data_size_);
    put(out, offset_data);
    put(out, size_data_per_element);
    put(out, M_);
    put(out, maxM_);
    put(out, size_links_level0);
}

void HierarchicalNSW::SaveEdges(const std::string &location)
{
    std::cout << "Saving edges to " << location << std::endl;
    std::ofstream out(location, std::ios::binary);
    for (size_t i = 0; i < maxelements_; i++) {
        const uint8_t *rec = get_linklist0((idx_t)i);
        const uint32_t n = rec[0];
        put(out, n);
        out.write(reinterpret_cast<const char *>(rec + 1), (std::streamsize)n * sizeof(idx_t));
    }
}

void HierarchicalNSW::LoadInfo(const std::string &location)
{
    std::cout << "Loading info from " << location << std::endl;
    std::ifstream in(location, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + location);
    get(in, maxelements_);
    get(in, enterpoint_node);
    get(in, data_size_);
    get(in, offset_data);
    get(in, size_data_per_element);
    get(in, M_);
    get(in, maxM_);
    get(in, size_links_level0);
    if (!in || size_links_level0 != maxM_ * sizeof(idx_t) + 1 || offset_data != size_links_level0 ||
        size_data_per_element != size_links_level0 + data_size_ || data_size_ % sizeof(float) || maxM_ > 255 ||
        enterpoint_node >= maxelements_)
        throw std::runtime_error("inconsistent HNSW info file " + location);
    d_ = data_size_ / sizeof(float);
    free(data_level0_memory_);
    data_level0_memory_ = static_cast<char *>(calloc(maxelements_, size_data_per_element));
    if (!data_level0_memory_)
        throw std::bad_alloc();
    efConstruction_ = 0;
    cur_element_count = maxelements_;
    delete visitedlistpool;
    visitedlistpool = new VisitedListPool(1, maxelements_);
}

void HierarchicalNSW::LoadData(const std::string &location)
{
    std::cout << "Loading data from " << location << std::endl;
    std::ifstream in(location, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + location);
    for (size_t i = 0; i < maxelements_; i++) {
        uint32_t dim = 0;
        get(in, dim);
        if (!in || dim != d_) {
            std::cout << "Wront data dim" << std::endl;
            exit(1);
        }
        in.read(reinterpret_cast<char *>(getDataByInternalId((idx_t)i)), (std::streamsize)data_size_);
    }
}

void HierarchicalNSW::LoadEdges(const std::string &location)
{
    std::cout << "Loading edges from " << location << std::endl;
    std::ifstream in(location, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + location);
    for (size_t i = 0; i < maxelements_; i++) {
        uint32_t n = 0;
        get(in, n);
        if (!in || n > maxM_)
            throw std::runtime_error("bad edge record in " + location);
        uint8_t *rec = get_linklist0((idx_t)i);
        rec[0] = (uint8_t)n;
        in.read(reinterpret_cast<char *>(rec + 1), (std::streamsize)n * sizeof(idx_t));
    }
}

} // namespace hnswlib
