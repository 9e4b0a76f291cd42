// On-device HNSW coarse walk: hnswlib::HierarchicalNSW::searchKnn / searchBaseLayer
// (hnswlib/hnswalg.cpp:227-234, 48-109) plus the unload loop of IndexIVF_HNSW.cpp:249-259,
// one wavefront per query.
//
// The reference walks with two std::priority_queue<pair<float,idx_t>>: `topResults` (max-heap, <= ef
// entries) and `candidateSet` (min-heap by negated distance).  Both are replaced here by ONE array R
// sorted ascending by (dist, id), each entry carrying an "expanded" bit, DISTRIBUTED OVER THE LANES'
// REGISTERS (entry i lives in lane i % 64, register i / 64), so that selecting the next candidate,
// testing admission and inserting are ballots, readlanes and one-lane DPP shifts -- no memory at all:
//   * topResults            = R itself (its maximum is R[n-1]);
//   * candidateSet's top    = the not-yet-expanded entry of smallest distance (largest id among equal
//                             distances, as pair<-dist,id> orders them).  A candidate that was evicted
//                             from topResults can only still be popped if its distance EQUALS the
//                             current lowerBound (hnswalg.cpp:67 breaks on '>'), so evicted entries are
//                             kept in a small LDS `tail` while that equality holds and dropped otherwise.
// Results are therefore identical to the reference's for any input, ties included.
//
// Per expansion: lanes read the node's link count and <= maxM links (coalesced), test-and-set the
// visited bitmap with one returning atomicOr each; the not-yet-visited neighbours are compacted and a
// quad of lanes evaluates the exact 8-accumulator L2 distance (hnswalg.cpp:326-357) of each, 16 rows per
// pass with all of a row's loads in flight at once; then admissions (hnswalg.cpp:93-103) are applied in
// link order.
// Bound: latency of the dependent round trips (links -> bitmap -> vectors) and the bandwidth of the
// gathered 4*d-byte rows; hidden by keeping one query per resident wavefront (queries are handed out
// through an atomic counter).
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>
#include <stdlib.h>

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

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// value of the lane below (lane 0 gets its own value back)
__device__ __forceinline__ unsigned long long lane_below_u64(unsigned long long v)
{
    // DPP wave_shr:1 (0x138): every lane reads lane-1 across the whole wavefront
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
    const int hi =
        __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}

template <int NCH> struct RSet {
    unsigned long long r[NCH]; // entry i: lane i & 63, register i >> 6

    __device__ __forceinline__ unsigned long long get(int idx) const // idx wave-uniform
    {
        const int c = idx >> 6, l = __builtin_amdgcn_readfirstlane(idx & 63);
        unsigned long long v = 0;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++)
            if (cc == c)
                v = readlane_u64(r[cc], l);
        return v;
    }
    __device__ __forceinline__ void mark_expanded(int idx, int lane)
    {
#pragma unroll
        for (int cc = 0; cc < NCH; cc++)
            if (cc == (idx >> 6) && lane == (idx & 63))
                r[cc] |= 1ull;
    }
    // first not-yet-expanded entry among the first n, or -1
    __device__ __forceinline__ int first_unexpanded(int n, int lane) const
    {
        int first = -1;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++) {
            const unsigned long long m = __ballot(cc * 64 + lane < n && !(r[cc] & 1ull));
            if (m && first < 0)
                first = cc * 64 + (__ffsll((long long)m) - 1);
        }
        return first;
    }
    // last not-yet-expanded entry with distance bits db at index >= first
    __device__ __forceinline__ int last_unexpanded_with(uint32_t db, int first, int n, int lane) const
    {
        int last = first;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++) {
            const int i = cc * 64 + lane;
            const unsigned long long m =
                __ballot(i < n && i >= first && key_dist_bits(r[cc]) == db && !(r[cc] & 1ull));
            if (m)
                last = cc * 64 + (63 - __clzll((long long)m));
        }
        return last;
    }
    // number of entries among the first n whose (dist, id) is below K
    __device__ __forceinline__ int rank_of(unsigned long long K, int n, int lane) const
    {
        int pos = 0;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++)
            pos += __popcll(__ballot(cc * 64 + lane < n && (r[cc] & ~1ull) < K));
        return pos;
    }
    // insert K at sorted position pos, shifting the entries above it up by one (the last one falls off
    // when the set is full: the caller read it first)
    __device__ __forceinline__ void insert_at(unsigned long long K, int pos, int lane)
    {
#pragma unroll
        for (int cc = NCH - 1; cc >= 0; cc--) {
            unsigned long long below = lane_below_u64(r[cc]);
            if (cc > 0) {
                const unsigned long long carry = readlane_u64(r[cc - 1], 63);
                if (lane == 0)
                    below = carry;
            }
            const int i = cc * 64 + lane;
            r[cc] = i < pos ? r[cc] : (i == pos ? K : below);
        }
    }
};

} // namespace

// One wavefront (64-thread block) per resident slot; queries are taken from an atomic counter.
// dynamic LDS: float query[d] | u64 tail[kTailCap]
// Visited set (visited_list_pool.h:8-33 in the reference).  LDSVIS: an exact hash set in LDS -- kVisBuckets
// buckets of four 16-bit tags (tag = id / kVisBuckets + 1, bucket = id % kVisBuckets), inserted with a 64-bit
// compare-and-swap on the bucket; an id whose bucket is already full is recorded in the wave's global bitmap
// instead (a few percent of the ids), and is looked up there from then on because a full bucket never empties.
// Why: one returning global atomicOr per neighbour on bitmaps that do not fit L2 was the walk's largest cost
// (19.6 M scattered atomics per 10 k queries, ~1 ms; MI355X guide: scattered atomics run ~17x below the
// coalesced rate).
constexpr int kVisBuckets = 1024;

// TAGW = bits per tag: 8 (8 tags per bucket, graphs up to 255 * 1024 nodes), 12 (5 per bucket, up to 4095 * 1024),
// 16 (4 per bucket, up to 65535 * 1024); 0 = no LDS set, global bitmap only.  Fields are scanned SWAR-style:
// haszero(v) = (v - ones) & ~v & highs flags the lowest zero field exactly (and is non-zero iff some field is zero).
template <int TAGW> struct VisFields {
    static constexpr int slots = 64 / TAGW;
    static constexpr unsigned long long ones()
    {
        unsigned long long o = 0;
        for (int i = 0; i < slots; i++)
            o |= 1ull << (TAGW * i);
        return o;
    }
};

template <int TAGW>
__device__ __forceinline__ bool visit_test_and_set(uint32_t id, unsigned long long *vt, uint32_t *bm, bool &used_bitmap)
{
    if constexpr (TAGW != 0) {
        constexpr unsigned long long kOnes = VisFields<TAGW>::ones();
        constexpr unsigned long long kHighs = kOnes << (TAGW - 1);
        const uint32_t b = id & (kVisBuckets - 1);
        const unsigned long long tag = (unsigned long long)((id / kVisBuckets) + 1);
        for (;;) {
            const unsigned long long old = vt[b];
            const unsigned long long x = old ^ (tag * kOnes);
            if ((x - kOnes) & ~x & kHighs)
                return false; // some field equals the tag: seen before
            const unsigned long long z = (old - kOnes) & ~old & kHighs; // lowest set bit = first empty field
            if (!z)
                break; // bucket full: this id lives in the bitmap
            const int shift = (__ffsll((long long)z) - 1) - (TAGW - 1);
            const unsigned long long want = old | (tag << shift);
            if (atomicCAS(&vt[b], old, want) == old)
                return true;
            // another lane changed the bucket in the meantime: look again
        }
        used_bitmap = true;
    }
    const uint32_t bit = 1u << (id & 31);
    return !(atomicOr(&bm[id >> 5], bit) & bit);
}

template <int NCH, int MINW, int TAGW>
__global__ __launch_bounds__(64, MINW) void hnsw_walk_kernel(GraphTables g, const float *__restrict__ xq, int nq, int nprobe,
                                                       int ef, uint32_t *__restrict__ coarse_ids,
                                                       float *__restrict__ coarse_dists,
                                                       uint32_t *__restrict__ visited, size_t vwords,
                                                       uint32_t *__restrict__ status, uint32_t *__restrict__ next_query)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_q = reinterpret_cast<float *>(smem);
    unsigned long long *tail = reinterpret_cast<unsigned long long *>(smem + (size_t)g.d * sizeof(float));
    unsigned long long *vt = tail + kTailCap; // [kVisBuckets] when TAGW != 0
    constexpr bool LDSVIS = TAGW != 0;

    const int lane = threadIdx.x;
    uint32_t *bm = visited + (size_t)blockIdx.x * vwords;
    bool bitmap_dirty = true; // the bitmap must be wiped before its first use and after any query that used it

    for (;;) {
        int q = 0;
        if (lane == 0)
            q = (int)atomicAdd(next_query, 1u);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= nq)
            break;

        // reset the visited set (visited_list_pool.h:25-32 does it by epoch) and stage the query
        if (!LDSVIS || bitmap_dirty) {
            uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
            const size_t v4 = vwords / 4; // vwords is padded to a multiple of 4
            for (size_t w = lane; w < v4; w += 64)
                bm4[w] = make_uint4(0u, 0u, 0u, 0u);
            bitmap_dirty = false;
        }
        if (LDSVIS)
            for (int w = lane; w < kVisBuckets; w += 64)
                vt[w] = 0ull;
        __syncthreads(); // the previous query's readers of s_q are done
        for (int i = lane; i < g.d; i += 64)
            s_q[i] = xq[(size_t)q * g.d + i];
        __syncthreads(); // also orders the bitmap reset before the atomics below

        RSet<NCH> R;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++)
            R.r[cc] = ~0ull;
        int n = 1;     // entries in R (== topResults.size())
        int ntail = 0; // evicted entries whose distance still equals the lower bound
        {
            // hnswalg.cpp:56-62: seed with the enter point
            const float d0 = l2_ref_order_quad(g.vectors + (size_t)g.enterpoint * g.d, s_q, g.d, lane & 3);
            bool ub = false;
            if (lane == 0) {
                R.r[0] = mk_key(d0, g.enterpoint);
                (void)visit_test_and_set<TAGW>(g.enterpoint, vt, bm, ub);
            }
            __syncthreads();
        }
        bool used_bitmap = false;

        for (;;) {
            // ---- candidateSet.top(): first unexpanded entry of R, ties -> largest id; tail joins at dist == max
            const int first = R.first_unexpanded(n, lane);
            const uint32_t maxbits = key_dist_bits(R.get(n - 1));
            int pick = -1;      // index in R, or
            int pick_tail = -1; // index in tail
            uint32_t pick_id = 0;
            if (first >= 0) {
                const uint32_t db = key_dist_bits(R.get(first));
                pick = R.last_unexpanded_with(db, first, n, lane);
                pick_id = key_id(R.get(pick));
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
            if (pick >= 0) {
                R.mark_expanded(pick, lane);
            } else {
                __syncthreads();
                if (lane == 0)
                    tail[pick_tail] = tail[ntail - 1];
                ntail--;
                __syncthreads();
            }

            // ---- expand: links, visited test-and-set, distances (hnswalg.cpp:72-91)
            const uint32_t node = pick_id;
            const int cnt = g.counts[node];
            uint32_t nb = 0;
            if (lane < g.maxM)
                nb = g.links[(size_t)node * g.maxM + lane]; // issued together with the count
            bool fresh = false;
            if (lane < cnt)
                fresh = visit_test_and_set<TAGW>(nb, vt, bm, used_bitmap);
            const unsigned long long mask = __ballot(fresh);
            const int nfresh = __popcll(mask);

            // ---- distances, 16 rows per pass (a quad of lanes per row), then admissions in link order
            for (int base = 0; base < nfresh && ntail >= 0; base += 16) {
                const int r = base + (lane >> 2);
                const bool active = r < nfresh;
                const int src = active ? nth_set_bit(mask, r) : 0;
                const uint32_t nbq = (uint32_t)__shfl((int)nb, src, 64);
                float dq = 0.f;
                if (active)
                    dq = l2_ref_order_quad(g.vectors + (size_t)nbq * g.d, s_q, g.d, lane & 3);
                // Rows that cannot be admitted are dropped wave-wide before the sequential part: once the
                // set is full its maximum only decreases, so a row failing 'top > dist' (hnswalg.cpp:93)
                // against the current maximum fails against every later one too.
                const float top0 = __uint_as_float(key_dist_bits(R.get(n - 1)));
                unsigned long long cand = __ballot(active && (lane & 3) == 0 && (n < ef || top0 > dq));
                while (cand) { // hnswalg.cpp:93-103, in link order
                    const int b = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const float dj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dq), b));
                    const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)nbq, b);
                    const unsigned long long topk = R.get(n - 1);
                    const float topd = __uint_as_float(key_dist_bits(topk));
                    if (!(topd > dj || n < ef))
                        continue;
                    const unsigned long long K = mk_key(dj, idj);
                    const int pos = R.rank_of(K, n, lane);
                    const bool full = n == ef;
                    R.insert_at(K, pos, lane);
                    if (!full)
                        n++;
                    // bookkeeping of candidates that left topResults but may still be popped
                    const uint32_t newmax = key_dist_bits(R.get(n - 1));
                    if (ntail > 0 && key_dist_bits(tail[0]) != newmax)
                        ntail = 0; // lower bound moved below them: dead for good
                    if (full && !(topk & 1ull) && key_dist_bits(topk) == newmax) {
                        if (ntail < kTailCap) {
                            __syncthreads();
                            if (lane == 0)
                                tail[ntail] = topk;
                            ntail++;
                            __syncthreads();
                        } else {
                            // more than kTailCap exact distance ties at the boundary: cannot be represented
                            if (lane == 0)
                                atomicOr(status, kStatusHnswTieOverflow);
                            ntail = -1;
                            break;
                        }
                    }
                }
            }
            if (ntail < 0)
                break;
        }

        if (LDSVIS && __ballot(used_bitmap))
            bitmap_dirty = true;
        // searchKnn pops down to nprobe (hnswalg.cpp:229-233); IndexIVF_HNSW.cpp:249-259 unloads nearest first
#pragma unroll
        for (int cc = 0; cc < NCH; cc++) {
            const int i = cc * 64 + lane;
            if (i < nprobe) {
                const bool have = ntail >= 0 && i < n;
                coarse_ids[(size_t)q * nprobe + i] = have ? key_id(R.r[cc]) : 0xffffffffu;
                coarse_dists[(size_t)q * nprobe + i] = have ? __uint_as_float(key_dist_bits(R.r[cc])) : 0.f;
            }
        }
    }
}

// how many wavefront slots the walk keeps resident for a given ef (sizes the visited bitmaps)
int coarse_slots_for(int ef)
{
    const int nch = (ef + 63) / 64;
    const int waves_per_simd = nch <= 4 ? 8 : (nch <= 8 ? 4 : 3); // upper bound; extra blocks just find the queue empty
    return 256 * 4 * waves_per_simd;
}

hipError_t launch_coarse(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                         uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                         size_t visited_words_per_slot, int nslots, uint32_t *status, uint32_t *next_query)
{
    if (nq == 0)
        return hipSuccess;
    if (ef > 1024 || ef < 1 || nprobe > ef || g.maxM > 64 || g.n >= 0x80000000u || (visited_words_per_slot & 3))
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(next_query, 0, sizeof(uint32_t), s);
    if (e != hipSuccess)
        return e;
    // the LDS visited set needs 16-bit tags (n <= 2^26); IVFHNSW_WALK_VIS=bitmap forces the old form (A/B runs)
    static const bool force_bitmap = [] {
        const char *e = getenv("IVFHNSW_WALK_VIS");
        return e && e[0] == 'b';
    }();
    const int tagw = force_bitmap ? 0
                     : g.n <= 255u * kVisBuckets ? 8
                     : g.n <= 4095u * kVisBuckets ? 12
                     : g.n <= 65535u * kVisBuckets ? 16 : 0;
    const size_t shm = (size_t)g.d * sizeof(float) + (size_t)kTailCap * sizeof(unsigned long long) +
                       (tagw ? (size_t)kVisBuckets * sizeof(unsigned long long) : 0);
    const int nch = (ef + 63) / 64;
#define IVFHNSW_WALK_T(N, W, T)                                                                                     \
    hipLaunchKernelGGL((hnsw_walk_kernel<N, W, T>), dim3(nslots), dim3(64), shm, s, g, xq, nq, nprobe, ef, coarse_ids, \
                       coarse_dists, visited_scratch, visited_words_per_slot, status, next_query)
#define IVFHNSW_WALK(N, W)             \
    do {                               \
        if (tagw == 8)                 \
            IVFHNSW_WALK_T(N, W, 8);   \
        else if (tagw == 12)           \
            IVFHNSW_WALK_T(N, W, 12);  \
        else if (tagw == 16)           \
            IVFHNSW_WALK_T(N, W, 16);  \
        else                           \
            IVFHNSW_WALK_T(N, W, 0);   \
    } while (0)
    // tuning knob (A/B on the device).  Measured on MI355X (100M / 2^17-centroid workload, ef 80): the
    // walk is bound by per-wave instruction issue, and 4 waves/SIMD with ~110 VGPRs (2.05 ms per 10 k
    // queries) beats 8 waves/SIMD with ~55 (2.6 ms), so 4 is the default.
    static const int occ = [] {
        const char *e = getenv("IVFHNSW_WALK_OCC");
        const int v = e ? atoi(e) : 4;
        return (v == 5 || v == 6 || v == 8) ? v : 4;
    }();
#define IVFHNSW_WALK_N(N)              \
    do {                               \
        if (occ == 8)                  \
            IVFHNSW_WALK(N, 8);        \
        else if (occ == 6)             \
            IVFHNSW_WALK(N, 6);        \
        else if (occ == 5)             \
            IVFHNSW_WALK(N, 5);        \
        else                           \
            IVFHNSW_WALK(N, 4);        \
    } while (0)
    if (nch <= 1) {
        IVFHNSW_WALK_N(1);
    } else if (nch <= 2) {
        IVFHNSW_WALK_N(2);
    } else if (nch <= 4) {
        IVFHNSW_WALK_N(4);
    } else if (nch <= 8) {
        IVFHNSW_WALK(8, 4);
    } else {
        IVFHNSW_WALK(16, 4);
    }
#undef IVFHNSW_WALK_N
#undef IVFHNSW_WALK
#undef IVFHNSW_WALK_T
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
