// Per-thread "visited" marks for the host-side graph walk: an epoch array (one uint16 per node, cleared only
// when the epoch wraps) handed out from a mutex-guarded free list, so concurrent searchKnn calls are safe.
// Same role as the reference's hnswlib/visited_list_pool.h.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace hnswlib {

typedef uint16_t vl_type;

class VisitedList {
public:
    vl_type curV;
    vl_type *mass;
    size_t numelements;

    explicit VisitedList(size_t n) : curV((vl_type)-1), mass(new vl_type[n]), numelements(n) {}
    ~VisitedList() { delete[] mass; }
    VisitedList(const VisitedList &) = delete;
    VisitedList &operator=(const VisitedList &) = delete;

    /// start a new search: bump the epoch, wiping the array when it wraps to 0
    void reset()
    {
        if (++curV == 0) {
            std::memset(mass, 0, sizeof(vl_type) * numelements);
            curV = 1;
        }
    }
};

class VisitedListPool {
    std::vector<std::unique_ptr<VisitedList>> free_;
    std::mutex guard_;
    size_t numelements_;

public:
    VisitedListPool(size_t initial, size_t numelements) : numelements_(numelements)
    {
        for (size_t i = 0; i < initial; i++)
            free_.emplace_back(new VisitedList(numelements));
    }
    VisitedList *getFreeVisitedList()
    {
        VisitedList *vl = nullptr;
        {
            std::lock_guard<std::mutex> lock(guard_);
            if (!free_.empty()) {
                vl = free_.back().release();
                free_.pop_back();
            }
        }
        if (!vl)
            vl = new VisitedList(numelements_);
        vl->reset();
        return vl;
    }
    void releaseVisitedList(VisitedList *vl)
    {
        std::lock_guard<std::mutex> lock(guard_);
        free_.emplace_back(vl);
    }
};

} // namespace hnswlib
