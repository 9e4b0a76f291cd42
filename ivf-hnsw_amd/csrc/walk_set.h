// The walk's result set and its keys, shared by the throughput walk (kernels_hnsw.hip: one wavefront per query) and the
// latency walk (kernels_hnsw_lat.hip: one workgroup per query).  See kernels_hnsw.hip for how the reference's two
// priority queues (hnswalg.cpp:48-109) map onto ONE sorted array distributed over a wavefront's registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ivfhnsw_gpu_impl {

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

// lanes of register cc that hold entries below index n (wave-uniform, scalar unit): ANDed with a ballot it replaces
// a per-lane range compare inside the ballot's operand, which hipcc materialises through a select and a compare
__device__ __forceinline__ unsigned long long lanes_below(int n, int cc)
{
    const int r = n - cc * 64;
    return r >= 64 ? ~0ull : (r > 0 ? (1ull << r) - 1ull : 0ull);
}

// ---- the tail beyond kTailCap entries --------------------------------------------------------------------------------
// Evicted candidates wait in the tail only while their distance EQUALS the set's maximum, so every tail entry has the
// same distance and the tail is a SET OF IDS.  The reference has no limit on it (hnswalg.cpp:67-68,93: the candidate
// heap is a std::priority_queue); up to kTailCap entries live in LDS, and when one more arrives the whole tail moves into
// a per-wavefront global bitmap (n bits, zero between uses) and stays there until it dies or drains.  Rare path (65 exact
// distance ties at the efSearch boundary): every access is a returning atomic, so no stale L1 line is ever read.
struct TailSpill {
    uint32_t *bm; // [words] of this wavefront, all zero outside spill mode
    int hi;       // highest word that may hold a bit (-1: none); wave-uniform
    int count;    // ids in the bitmap; > 0 = spill mode; wave-uniform

    __device__ __forceinline__ void add(uint32_t id, int lane)
    {
        if (lane == 0)
            atomicOr(&bm[id >> 5], 1u << (id & 31));
        const int w = (int)(id >> 5);
        hi = w > hi ? w : hi;
        count++;
    }
    // largest id in the bitmap (count > 0); lowers `hi` to its word; the bit stays
    __device__ __noinline__ uint32_t peek_max(int lane)
    {
        for (int w0 = hi; w0 >= 0; w0 -= 64) {
            const int w = w0 - lane;
            const uint32_t v = w >= 0 ? atomicOr(&bm[w], 0u) : 0u;
            const unsigned long long m = __ballot(v != 0u);
            if (m) {
                const int first = __ffsll((long long)m) - 1; // lowest lane = highest word
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)v, first);
                hi = w0 - first;
                return (uint32_t)hi * 32u + (31u - (uint32_t)__clz((int)wv));
            }
        }
        hi = -1;
        return 0xffffffffu;
    }
    __device__ __forceinline__ void remove(uint32_t id, int lane)
    {
        if (lane == 0)
            atomicAnd(&bm[id >> 5], ~(1u << (id & 31)));
        count--;
    }
    __device__ __noinline__ void clear(int lane)
    {
        for (int w = lane; w <= hi; w += 64)
            atomicAnd(&bm[w], 0u);
        hi = -1;
        count = 0;
    }
};

template <int NCH> struct RSet {
    unsigned long long r[NCH]; // entry i: lane i & 63, register i >> 6

    __device__ __forceinline__ unsigned long long get(int idx) const // idx wave-uniform
    {
        const int c = idx >> 6, l = __builtin_amdgcn_readfirstlane(idx & 63);
        if constexpr (NCH <= 4) {
            // both registers read, one kept: four readlanes and two scalar selects, no branch
            unsigned long long v = readlane_u64(r[0], l);
#pragma unroll
            for (int cc = 1; cc < NCH; cc++) {
                const unsigned long long t = readlane_u64(r[cc], l);
                v = c == cc ? t : v;
            }
            return v;
        }
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
            const unsigned long long m = __ballot(!(r[cc] & 1ull)) & lanes_below(n, cc);
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
    // Sorted insertion without a position: every lane decides from its own entry and the one below it
    // (keep it, become K, or take the one below) -- two DPP shifts and two compares per register, no ballot,
    // no scalar round trip.  Entries at and beyond n are all-ones or left-overs, never below K's successor:
    // they shift like real ones and are never read.  K is not in the set (ids are visited once).
    __device__ __forceinline__ void insert_sorted(unsigned long long K, int lane)
    {
#pragma unroll
        for (int cc = NCH - 1; cc >= 0; cc--) {
            unsigned long long below = lane_below_u64(r[cc]);
            bool below_lt = (below & ~1ull) < K;
            if (cc > 0) {
                const unsigned long long carry = readlane_u64(r[cc - 1], 63);
                if (lane == 0) {
                    below = carry;
                    below_lt = (carry & ~1ull) < K;
                }
            } else if (lane == 0) {
                below_lt = true; // nothing below entry 0
            }
            r[cc] = (r[cc] & ~1ull) < K ? r[cc] : (below_lt ? K : below);
        }
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


} // namespace ivfhnsw_gpu_impl
