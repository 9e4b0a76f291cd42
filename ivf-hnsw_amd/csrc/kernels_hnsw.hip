// On-device HNSW coarse walk: hnswlib::HierarchicalNSW::searchKnn / searchBaseLayer
// (hnswlib/hnswalg.cpp:227-234, 48-109) plus the unload loop of IndexIVF_HNSW.cpp:249-259,
// one wavefront per query.
//
// The reference walks with two std::priority_queue<pair<float,idx_t>>: `topResults` (max-heap, <= ef
// entries) and `candidateSet` (min-heap by negated distance).  Both are replaced here by ONE array R
// sorted ascending by (dist, id), held in LDS, each entry carrying an "expanded" bit:
//   * topResults            = R itself (its maximum is R[n-1]);
//   * candidateSet's top    = the not-yet-expanded entry of smallest distance (largest id among equal
//                             distances, as pair<-dist,id> orders them).  A candidate that was evicted
//                             from topResults can only still be popped if its distance EQUALS the
//                             current lowerBound (hnswalg.cpp:67 breaks on '>'), so evicted entries are
//                             kept in a small `tail` while that equality holds and dropped otherwise.
// Results are therefore identical to the reference's for any input, ties included.
//
// Per expansion: lanes read the node's <= maxM links (coalesced), test-and-set the visited bitmap
// with one returning atomicOr each, lane j evaluates the exact 8-accumulator L2 distance
// (hnswalg.cpp:326-357) of neighbour j, then admissions (hnswalg.cpp:93-103) are applied in link order
// with wave-parallel sorted insertion.
// Bound: latency of dependent HBM/L2 round trips (links -> bitmap -> vectors), hidden by running one
// query per resident wavefront slot.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>

namespace ivfhnsw_gpu_impl {

namespace {

constexpr int kTailCap = 64;

// key = dist bits (non-negative float: bit order == value order) : id : expanded flag
__device__ __forceinline__ unsigned long long mk_key(float dist, uint32_t id)
{
    return ((unsigned long long)__float_as_uint(dist) << 32) | ((unsigned long long)id << 1);
}
__device__ __forceinline__ uint32_t key_dist_bits(unsigned long long k) { return (uint32_t)(k >> 32); }
__device__ __forceinline__ uint32_t key_id(unsigned long long k) { return (uint32_t)(k & 0xffffffffu) >> 1; }

} // namespace

// One wavefront (64-thread block) per slot; slot s walks queries s, s + nslots, ...
// dynamic LDS: float query[d] | u64 R[efc + 1] | u64 tail[kTailCap]
__global__ __launch_bounds__(64) void hnsw_walk_kernel(GraphTables g, const float *__restrict__ xq, int nq, int nprobe,
                                                       int ef, int efc, uint32_t *__restrict__ coarse_ids,
                                                       float *__restrict__ coarse_dists,
                                                       uint32_t *__restrict__ visited, size_t vwords,
                                                       uint32_t *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_q = reinterpret_cast<float *>(smem);
    unsigned long long *R = reinterpret_cast<unsigned long long *>(smem + (size_t)g.d * sizeof(float));
    unsigned long long *tail = R + efc + 1;

    const int lane = threadIdx.x;
    const int nchunk = efc / 64;
    uint32_t *bm = visited + (size_t)blockIdx.x * vwords;

    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        // reset the visited bitmap (visited_list_pool.h:25-32 does it by epoch) and stage the query
        for (size_t w = lane; w < vwords; w += 64)
            bm[w] = 0u;
        for (int i = lane; i < g.d; i += 64)
            s_q[i] = xq[(size_t)q * g.d + i];
        __syncthreads();

        int n = 0;     // entries in R (== topResults.size())
        int ntail = 0; // evicted entries whose distance still equals the lower bound
        {
            // hnswalg.cpp:56-62: seed with the enter point
            const float d0 = l2_ref_order(g.vectors + (size_t)g.enterpoint * g.d, s_q, g.d);
            if (lane == 0) {
                R[0] = mk_key(d0, g.enterpoint);
                bm[g.enterpoint >> 5] = 1u << (g.enterpoint & 31);
            }
            n = 1;
        }
        __syncthreads();

        for (;;) {
            // ---- candidateSet.top(): first unexpanded entry of R, ties -> largest id; tail joins at dist == max
            int first = -1;
            for (int c = 0; c < nchunk && first < 0; c++) {
                const int i = c * 64 + lane;
                const bool un = i < n && !(R[i] & 1ull);
                const unsigned long long m = __ballot(un);
                if (m)
                    first = c * 64 + (__ffsll((long long)m) - 1);
            }
            const uint32_t maxbits = key_dist_bits(R[n - 1]);
            int pick = -1;          // index in R, or
            int pick_tail = -1;     // index in tail
            uint32_t pick_id = 0;
            if (first >= 0) {
                const uint32_t db = key_dist_bits(R[first]);
                // last unexpanded entry with the same distance (the run is contiguous in R)
                int last = first;
                for (int c = first / 64; c < nchunk; c++) {
                    const int i = c * 64 + lane;
                    const bool hit = i < n && i >= first && key_dist_bits(R[i]) == db && !(R[i] & 1ull);
                    const unsigned long long m = __ballot(hit);
                    if (m)
                        last = c * 64 + (63 - __clzll((long long)m));
                    const bool beyond = i < n && key_dist_bits(R[i]) > db;
                    if (__ballot(beyond))
                        break;
                }
                pick = last;
                pick_id = key_id(R[last]);
                if (db == maxbits && ntail > 0) {
                    // tail entries share this distance; the larger id pops first
                    for (int t = 0; t < ntail; t++)
                        if (key_id(tail[t]) > pick_id) {
                            pick_id = key_id(tail[t]);
                            pick_tail = t;
                        }
                    if (pick_tail >= 0)
                        pick = -1;
                }
            } else if (ntail > 0) {
                pick_tail = 0;
                pick_id = key_id(tail[0]);
                for (int t = 1; t < ntail; t++)
                    if (key_id(tail[t]) > pick_id) {
                        pick_id = key_id(tail[t]);
                        pick_tail = t;
                    }
            } else {
                break; // candidateSet exhausted (hnswalg.cpp:64) or only entries beyond lowerBound left (:67)
            }
            __syncthreads();
            if (lane == 0) {
                if (pick >= 0)
                    R[pick] |= 1ull;
                else
                    tail[pick_tail] = tail[ntail - 1];
            }
            if (pick < 0)
                ntail--;
            __syncthreads();

            // ---- expand: links, visited test-and-set, distances (hnswalg.cpp:72-91)
            const uint32_t node = pick_id;
            const int cnt = g.counts[node];
            uint32_t nb = 0;
            bool fresh = false;
            float dist = 0.f;
            if (lane < cnt) {
                nb = g.links[(size_t)node * g.maxM + lane];
                const uint32_t bit = 1u << (nb & 31);
                const uint32_t old = atomicOr(&bm[nb >> 5], bit);
                fresh = !(old & bit);
            }
            if (fresh)
                dist = l2_ref_order(g.vectors + (size_t)nb * g.d, s_q, g.d);
            unsigned long long todo = __ballot(fresh);

            // ---- admissions in link order (hnswalg.cpp:93-103)
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const float dj = __shfl(dist, j, 64);
                const uint32_t idj = __shfl(nb, j, 64);
                const unsigned long long topk = R[n - 1];
                const float topd = __uint_as_float(key_dist_bits(topk));
                if (!(topd > dj || n < ef))
                    continue;
                const unsigned long long K = mk_key(dj, idj);
                // position = number of entries with (dist,id) below K (flag bit masked out)
                int pos = 0;
                unsigned long long v[16];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    if (c < nchunk) {
                        const int i = c * 64 + lane;
                        v[c] = i < n ? R[i] : ~0ull;
                        pos += __popcll(__ballot(i < n && (v[c] & ~1ull) < K));
                    }
                }
                __syncthreads();
                const bool full = n == ef;
                const unsigned long long evicted = R[n - 1];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    if (c < nchunk) {
                        const int i = c * 64 + lane;
                        if (i < n && i >= pos && i + 1 < ef)
                            R[i + 1] = v[c];
                    }
                }
                if (lane == 0)
                    R[pos] = K;
                __syncthreads();
                if (!full)
                    n++;
                // bookkeeping of candidates that left topResults but may still be popped
                const uint32_t newmax = key_dist_bits(R[n - 1]);
                if (ntail > 0 && key_dist_bits(tail[0]) != newmax)
                    ntail = 0; // lower bound moved below them: dead for good
                if (full && !(evicted & 1ull) && key_dist_bits(evicted) == newmax) {
                    if (ntail < kTailCap) {
                        if (lane == 0)
                            tail[ntail] = evicted;
                        ntail++;
                    } else {
                        // more than kTailCap exact distance ties at the boundary: cannot be represented
                        if (lane == 0)
                            atomicOr(status, kStatusHnswTieOverflow);
                        ntail = -1;
                    }
                    __syncthreads();
                }
                if (ntail < 0)
                    break;
            }
            if (ntail < 0)
                break;
        }

        // searchKnn pops down to nprobe (hnswalg.cpp:229-233); IndexIVF_HNSW.cpp:249-259 unloads nearest first
        {
            for (int i = lane; i < nprobe; i += 64) {
                const bool have = ntail >= 0 && i < n;
                coarse_ids[(size_t)q * nprobe + i] = have ? key_id(R[i]) : 0xffffffffu;
                coarse_dists[(size_t)q * nprobe + i] = have ? __uint_as_float(key_dist_bits(R[i])) : 0.f;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_coarse(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                         uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                         size_t visited_words_per_slot, int nslots, uint32_t *status)
{
    if (nq == 0)
        return hipSuccess;
    if (ef > 1024 || ef < 1 || g.maxM > 64 || g.n >= 0x80000000u)
        return hipErrorInvalidValue;
    const int efc = ((ef + 63) / 64) * 64;
    const size_t shm = (size_t)g.d * sizeof(float) + (size_t)(efc + 1 + kTailCap) * sizeof(unsigned long long);
    hipLaunchKernelGGL(hnsw_walk_kernel, dim3(nslots), dim3(64), shm, s, g, xq, nq, nprobe, ef, efc, coarse_ids,
                       coarse_dists, visited_scratch, visited_words_per_slot, status);
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
