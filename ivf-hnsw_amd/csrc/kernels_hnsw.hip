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
#include "walk_set.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

// One wavefront (64-thread block) per resident slot; queries are taken from an atomic counter.
// dynamic LDS: float query[d] | u64 tail[kTailCap]
// Visited set (visited_list_pool.h:8-33 in the reference).  LDSVIS: an exact hash set in LDS -- kVisBuckets
// 64-bit buckets of tags plus a claim count (VisFields below; tag = id / buckets + 1, bucket = id % buckets), an id
// entered with one returning add and one or; an id whose bucket is already full is recorded in the wave's global
// bitmap instead (a few percent of the ids), and is looked up there from then on because a full bucket never empties.
// Why: one returning global atomicOr per neighbour on bitmaps that do not fit L2 was the walk's largest cost
// (19.6 M scattered atomics per 10 k queries, ~1 ms; MI355X guide: scattered atomics run ~17x below the
// coalesced rate).
// The bucket count follows the occupancy the kernel is built for (MINW waves per SIMD, 4 * MINW per CU sharing
// 160 KB of LDS): 1008 buckets = 7.9 KB at 4 (with query, planes, tail and merge buffer 9.8 KB per wave at efSearch 80:
// sixteen waves fill 157 of a CU's 160 KB), 6 KB at 5, 5 KB at 6.
constexpr int vis_buckets(int minw, int tagw = 8)
{
    // 10-bit tags (six per bucket) exist for graphs of about a million nodes -- the reference's 993 127 centroids,
    // the 2^20 of the 8-GPU bench: with 12-bit tags (five per bucket) 896 buckets ran 56 % full and every
    // expansion paid a global atomic for an overflowing neighbour.  1040 buckets x 6 = 6240 slots, 8320 bytes.
    return minw <= 4 ? (tagw == 10 ? 1040 : 1008) : minw == 5 ? 768 : minw <= 7 ? 640 : 384;
}

// TAGW = bits per tag; a 64-bit bucket holds `slots` tags and, in its top bits, the number of ids that claimed a
// field:  8 -> 7 tags + 8-bit count (graphs up to 255 * buckets nodes), 10 -> 5 + 14, 12 -> 4 + 16, 16 -> 3 + 16;
// 0 = no LDS set, global bitmap only.  Tags are scanned SWAR-style: haszero(v) = (v - ones) & ~v & highs is non-zero
// iff some field of v is zero.
// Insertion needs no compare-and-swap loop: an id not found in the bucket claims a field index with ONE returning
// add on the count and ORs its tag into that (still zero) field -- two LDS instructions, no retry, where the CAS form
// re-read the bucket whenever two lanes of a link list hashed to it.  This relies on the ids of one link list being
// distinct (two equal ids in one instruction would both be entered and both reported new): the upload drops
// repeated ids, which the reference skips as visited anyway (capi.cpp, upload_quantizer).
template <int TAGW> struct VisFields {
    static constexpr int cntw = TAGW == 8 ? 8 : TAGW == 10 ? 14 : 16;
    static constexpr int cshift = 64 - cntw;
    static constexpr int slots = cshift / TAGW;
    static constexpr unsigned long long ones()
    {
        unsigned long long o = 0;
        for (int i = 0; i < slots; i++)
            o |= 1ull << (TAGW * i);
        return o;
    }
};

template <int TAGW, int NB>
__device__ __forceinline__ bool visit_test_and_set(uint32_t id, unsigned long long *vt, uint32_t *bm, bool &used_bitmap)
{
    if constexpr (TAGW != 0) {
        using F = VisFields<TAGW>;
        constexpr unsigned long long kOnes = F::ones();
        constexpr unsigned long long kHighs = kOnes << (TAGW - 1);
        const uint32_t b = id % (uint32_t)NB; // NB is a constant: multiply and shift
        const unsigned long long tag = (unsigned long long)((id / (uint32_t)NB) + 1);
        const unsigned long long old = vt[b];
        const unsigned long long x = old ^ (tag * kOnes);
        if ((x - kOnes) & ~x & kHighs)
            return false; // some field equals the tag: seen before
        // a full bucket never empties, and no lane adds to a count it has seen full: the count stays below
        // slots + 64, far from wrapping
        if ((uint32_t)(old >> F::cshift) < (uint32_t)F::slots) {
            const unsigned long long got = atomicAdd(&vt[b], 1ull << F::cshift);
            const uint32_t idx = (uint32_t)(got >> F::cshift);
            if (idx < (uint32_t)F::slots) {
                atomicOr(&vt[b], tag << (TAGW * idx));
                return true;
            }
        }
        used_bitmap = true; // bucket full: this id lives in the bitmap
    }
    const uint32_t bit = 1u << (id & 31);
    return !(atomicOr(&bm[id >> 5], bit) & bit);
}

// Insertion of an id the caller KNOWS to be absent (visit_lookup said so and nothing was entered since): no read,
// just the claim and the or.  `bucket_full` is what that lookup saw: a full bucket is not added to any more, which
// keeps every count below slots + 64 (one instruction's worth of lanes), far from wrapping.
template <int TAGW, int NB>
__device__ __forceinline__ void visit_insert_absent(uint32_t id, bool bucket_full, unsigned long long *vt, uint32_t *bm,
                                                    bool &used_bitmap)
{
    static_assert(TAGW != 0, "LDS set only");
    using F = VisFields<TAGW>;
    if (!bucket_full) {
        const uint32_t b = id % (uint32_t)NB;
        const unsigned long long tag = (unsigned long long)((id / (uint32_t)NB) + 1);
        const unsigned long long got = atomicAdd(&vt[b], 1ull << F::cshift);
        const uint32_t idx = (uint32_t)(got >> F::cshift);
        if (idx < (uint32_t)F::slots) {
            atomicOr(&vt[b], tag << (TAGW * idx));
            return;
        }
    }
    used_bitmap = true;
    atomicOr(&bm[id >> 5], 1u << (id & 31));
}

// Read-only membership test of the LDS set (TAGW != 0).  An id whose bucket is full may live in the wave's bitmap:
// read through a returning atomic, which sees the atomicOr of earlier insertions (a plain load could hit a stale
// L1 line).
template <int TAGW, int NB>
__device__ __forceinline__ bool visit_lookup(uint32_t id, const unsigned long long *vt, uint32_t *bm, bool &bucket_full)
{
    static_assert(TAGW != 0, "LDS set only");
    constexpr unsigned long long kOnes = VisFields<TAGW>::ones();
    constexpr unsigned long long kHighs = kOnes << (TAGW - 1);
    const uint32_t b = id % (uint32_t)NB;
    const unsigned long long tag = (unsigned long long)((id / (uint32_t)NB) + 1);
    const unsigned long long old = vt[b];
    const unsigned long long x = old ^ (tag * kOnes);
    bucket_full = false;
    if ((x - kOnes) & ~x & kHighs)
        return true;
    bucket_full = (uint32_t)(old >> VisFields<TAGW>::cshift) >= (uint32_t)VisFields<TAGW>::slots;
    if (!bucket_full)
        return false; // fields left: nothing that hashed here has gone to the bitmap
    return (atomicOr(&bm[id >> 5], 0u) >> (id & 31)) & 1u;
}

// sum over half of a byte row (16-byte chunks h, h+2, ...) of (q'[i] - byte[i])^2; the pair of lanes adds up
__device__ __forceinline__ float byte_row_dist_half(const uint8_t *row, const float *sqp, int d, int h)
{
    const uint4 *r4 = reinterpret_cast<const uint4 *>(row);
    const int nchunk = d >> 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j0 = h; j0 < nchunk; j0 += 8) {
        // four chunks per lane in flight (a 128-byte row in one round trip), then the arithmetic
        uint4 w[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            w[i] = j0 + 2 * i < nchunk ? r4[j0 + 2 * i] : make_uint4(0u, 0u, 0u, 0u);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (j0 + 2 * i < nchunk) {
                const float4 *qq = reinterpret_cast<const float4 *>(sqp + 16 * (j0 + 2 * i));
                const uint32_t ws[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float4 qv = qq[c];
                    const float t0 = qv.x - (float)(ws[c] & 0xffu);
                    const float t1 = qv.y - (float)((ws[c] >> 8) & 0xffu);
                    const float t2 = qv.z - (float)((ws[c] >> 16) & 0xffu);
                    const float t3 = qv.w - (float)(ws[c] >> 24);
                    a0 = fmaf(t0, t0, a0);
                    a1 = fmaf(t1, t1, a1);
                    a2 = fmaf(t2, t2, a2);
                    a3 = fmaf(t3, t3, a3);
                }
            }
        }
    }
    return (a0 + a1) + (a2 + a3);
}

// STAMPS: diagnostic build only (IVFHNSW_WALK_STAMPS=1): s_memtime at segment boundaries, per-segment cycle sums
// added to stamp_out[0..5] (selection, links+visited, distances, admissions, per-query prologue, expansions).
__device__ __forceinline__ unsigned long long walk_stamp()
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): s_memtime returns through the scalar cache
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// FMODE: the exact rejection filter -- 0 off, 1 gather from GraphTables::qrows, 2 neighbour rows (nbrows),
// 3 neighbour rows with the survivors entered into the visited set late (see filter_first)
// SPILL: the "redo" instantiation.  The fast forms keep evicted candidates that tie with the lower bound in a kTailCap-entry
// LDS tail; a query that needs more (65 exact distance ties at the efSearch boundary -- the reference has no limit,
// hnswalg.cpp:67-68,93) is appended to the redo list instead of being finished, and walked again by this form, whose tail
// overflows into a per-wavefront global bitmap (walk_set.h TailSpill).  Keeping that code out of the fast forms matters:
// compiled into them it cost 17 % of the walk (1.365 -> 1.60 ms per 10 k queries, same box) in registers and scratch.
// redo_hdr: [0] number of listed queries, [1] the redo launch's own query counter (both zeroed with next_query); redo_list:
// the queries.
template <int NCH, int MINW, int TAGW, int FMODE, bool STAMPS = false, bool SPILL = false>
__global__ __launch_bounds__(64, MINW) void hnsw_walk_kernel(GraphTables g, const float *__restrict__ xq, int nq, int nprobe,
                                                       int ef, uint32_t *__restrict__ coarse_ids,
                                                       float *__restrict__ coarse_dists,
                                                       uint32_t *__restrict__ visited, size_t vwords,
                                                       uint32_t *__restrict__ status, uint32_t *__restrict__ next_query,
                                                       uint32_t *__restrict__ redo_hdr, uint32_t *__restrict__ redo_list,
                                                       uint32_t *__restrict__ tail_bitmaps = nullptr, // SPILL: [grid][vwords]
                                                       unsigned long long *__restrict__ stamp_out = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_q = reinterpret_cast<float *>(smem);
    // gather form of the filter only: the query in byte-row units, (q - q_lo) / q_step
    float *s_qp = s_q + g.d;
    const int dp = FMODE == 1 ? g.d : 0;
    // neighbour-row form: the same in 1/256 steps, four byte planes of 32 words (hi, lo, XL, XH)
    uint32_t *s_q8 = reinterpret_cast<uint32_t *>(s_qp + dp);
    unsigned long long *tail = reinterpret_cast<unsigned long long *>(smem + (size_t)(g.d + dp) * sizeof(float) + 512);
    constexpr int NB = vis_buckets(MINW, TAGW);
    unsigned long long st_acc[9] = {0, 0, 0, 0, 0, 0}, st_t = 0;
    // the tail doubles as the merge buffer of a pass's admissions (ef + 8 entries; only used while the tail is empty)
    const int merge_extra = (NCH <= 4 && ef + 8 > kTailCap) ? ef + 8 - kTailCap : 0;
    unsigned long long *vt = tail + kTailCap + merge_extra; // [NB] when TAGW != 0
    constexpr bool LDSVIS = TAGW != 0;

    const int lane = threadIdx.x;
    // the row of a 32-row batch whose bound this lane evaluates in the filter (oct_sum4's layout)
    const int judged_row = 8 * (((lane >> 2) & 1) * 2 + (lane & 1)) + (lane >> 3);
    const bool merge_on = g.merge_admissions != 0;
    constexpr bool prefilter = FMODE != 0;
    constexpr bool inline_rows = FMODE >= 2;
    // The rows' squared norms from a table built at upload (one more dword load per expansion) instead of sixteen
    // v_dot4 of each row with itself: measured on one box, 1.438 -> 1.384 ms per 10 k queries at 993 127 nodes, but
    // 1.085 -> 1.103 ms at 2^17 nodes, where the load's wait costs more than the instructions -- so it goes with
    // the form large graphs take.
    constexpr bool ROWNORMS = FMODE == 3;
    // Neighbour-row forms: a node's LINK COUNT travels with its id.  A node's record is nb_rows byte rows but the mean
    // degree is ~21 of 32: a third of the 4-KB read is zero padding, and the walk moves ~63 % of the HBM peak, so those
    // bytes are time.  The count only used to arrive together with the record.  Here the "id" of every key is
    // id << 7 | count (the count is a function of the id, so the (dist, id) order is the reference's; ids stay below 2^24)
    // and the link table carries id | count << 24 (GraphTables::links_c, built with the rows): when a node is popped its
    // count is already in a scalar register, and only the rows below it are fetched.
    constexpr bool CK = FMODE >= 2;
    auto enc_id = [](uint32_t id, uint32_t c) { return CK ? (id << 7) | c : id; };
    auto dec_id = [](uint32_t e) { return CK ? e >> 7 : e; };
    uint32_t *bm = visited + (size_t)blockIdx.x * vwords;
    TailSpill spill;
    spill.bm = SPILL ? tail_bitmaps + (size_t)blockIdx.x * vwords : nullptr;
    spill.hi = -1;
    spill.count = 0;
    // The bitmap must be clean before a query uses it.  With the LDS set it is only the overflow store (rarely touched):
    // the launcher hands it over zeroed ONCE (g.visited_clean) and a block that did touch it wipes it before it leaves,
    // so launches do not start by clearing n / 8 bytes per resident wavefront (508 MB per launch at 993 127 nodes --
    // it was ALL of the walk's WRITE_SIZE, 8 % of its traffic).
    bool bitmap_dirty = !(LDSVIS && g.visited_clean);
    bool redo_ran = false; // SPILL: this wavefront walked a query (an empty redo list costs a launch and nothing else)

    for (;;) {
        int q = 0;
        if constexpr (SPILL) {
            // the queries the fast forms could not finish
            if (lane == 0) {
                const uint32_t i = atomicAdd(&redo_hdr[1], 1u);
                q = i < redo_hdr[0] ? (int)redo_list[i] : nq;
            }
        } else {
            if (lane == 0)
                q = (int)atomicAdd(next_query, 1u);
        }
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= nq)
            break;
        redo_ran = true;
        if (STAMPS)
            st_t = walk_stamp();

        // reset the visited set (visited_list_pool.h:25-32 does it by epoch) and stage the query
        if (!LDSVIS || bitmap_dirty) {
            uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
            const size_t v4 = vwords / 4; // vwords is padded to a multiple of 4
            for (size_t w = lane; w < v4; w += 64)
                bm4[w] = make_uint4(0u, 0u, 0u, 0u);
            bitmap_dirty = false;
        }
        if (LDSVIS)
            for (int w = lane; w < NB; w += 64)
                vt[w] = 0ull;
        __syncthreads(); // the previous query's readers of s_q are done
        float qp_sq = 0.f;
        for (int i = lane; i < g.d; i += 64) {
            const float v = xq[(size_t)q * g.d + i];
            s_q[i] = v;
            if (prefilter) {
                const float vp = __fdiv_rn(__fsub_rn(v, g.q_lo), g.q_step);
                if (!inline_rows)
                    s_qp[i] = vp;
                qp_sq = fmaf(vp, vp, qp_sq);
            }
        }
        __syncthreads(); // also orders the bitmap reset before the atomics below
        // d = 128 (SIFT) and 96 (DEEP): the lane's share of the query for the 8-lanes-per-row distance stays in
        // registers (one LDS round trip less in every expansion's dependent chain)
        // (builds for more than 4 waves per SIMD have no registers to spare for it)
        constexpr bool QREG = MINW <= 4;
        float q_reg[QREG ? 16 : 1];
        if constexpr (QREG) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                q_reg[i] = (g.d == 128 || g.d == 96) && 8 * i < g.d ? s_q[8 * i + (lane & 7)] : 0.f;
        }
        // slack of the rejection test in byte-row units: quantisation error of the worst row, plus what the
        // rounding of s_qp can move a distance by (<= 2^-22 ||q'||; 2^-18 leaves a factor 16)
        float pf_slack = 0.f, pf_slack_q = 0.f, pf_bonus = 0.f;
        int q16_w = 0; // 128 * ||Q / 256||^2 of the two-byte query Q below (floor)
        int x_const = 0;    // 255 * sum XH
        bool x_any = false; // some component of this query lies outside the byte range
        bool x_hi = false;  // ... above it (the XH plane is not all zero)
        uint4 qp_h = make_uint4(0u, 0u, 0u, 0u), qp_l = qp_h, qp_x = qp_h;
        if (prefilter) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                qp_sq += __shfl_xor(qp_sq, off, 64);
            pf_slack = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(g.q_errc + 3.9e-6f * sqrtf(qp_sq))));
            if (inline_rows) {
                // The integer form of the filter.  q' = q_in + q_out: q_in is q' clamped to the byte range and
                // rounded to 1/256 steps, Q = 256 * hi + lo (two byte planes in LDS); q_out is what the clamp cut
                // off (zero for most components).  For a byte row c
                //   ||q' - c||^2 = ||q_in - c||^2 + 2 q_out.(q_in - c) + ||q_out||^2
                // where  ||q_in - c|| >= ||Q/256 - c|| - ||q_in - Q/256||  (the rounding, measured here: pf_slack_q),
                //        2 q_out.(q_in - c) >= XL.c + XH.(255 - c)  with XL = floor(2 |q_out|) below the range and
                //        XH the same above it (two more byte planes; every term of the sum is >= 0),
                //        ||q_out||^2 is a per-query constant (pf_bonus).
                // All dot products are exact in integers.
                float dq_sq = 0.f, bonus = 0.f;
                int s_hh = 0, s_hl = 0, s_ll = 0, s_xh = 0, s_xl = 0;
                if (lane < 32) {
                    uint32_t wh = 0, wl = 0, wxl = 0, wxh = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int comp = 4 * lane + b; // rows are zero beyond d: so is the query
                        const float vp = comp < g.d ? __fdiv_rn(__fsub_rn(s_q[comp], g.q_lo), g.q_step) : 0.f;
                        const float fh = fminf(255.f, fmaxf(0.f, floorf(vp))); // NaN -> 0, and the slack turns NaN: no rejections
                        const float fl = fminf(255.f, fmaxf(0.f, rintf((vp - fh) * 256.f)));
                        const float qin = fh + fl * 0.00390625f;
                        const bool below = vp < 0.f, above = vp >= 256.f;
                        const float e = (below || above) ? 0.f : vp - qin;
                        dq_sq = fmaf(e, e, dq_sq);
                        const float out = (below || above) ? fabsf(vp - qin) : 0.f; // |q_out|
                        bonus = fmaf(out, out, bonus);
                        const int x = (int)fminf(255.f, floorf(2.f * out));
                        const int hi = (int)fh, lo = (int)fl;
                        s_hh += hi * hi;
                        s_hl += hi * lo;
                        s_ll += lo * lo;
                        wh |= (uint32_t)hi << (8 * b);
                        wl |= (uint32_t)lo << (8 * b);
                        if (below)
                            wxl |= (uint32_t)x << (8 * b);
                        if (above)
                            wxh |= (uint32_t)x << (8 * b);
                        s_xl += below ? x : 0;
                        s_xh += above ? x : 0;
                    }
                    s_q8[lane] = wh;
                    s_q8[32 + lane] = wl;
                    s_q8[64 + lane] = wxl;
                    s_q8[96 + lane] = wxh;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    dq_sq += __shfl_xor(dq_sq, off, 64);
                    bonus += __shfl_xor(bonus, off, 64);
                    s_hh += __shfl_xor(s_hh, off, 64);
                    s_hl += __shfl_xor(s_hl, off, 64);
                    s_ll += __shfl_xor(s_ll, off, 64);
                    s_xl += __shfl_xor(s_xl, off, 64);
                    s_xh += __shfl_xor(s_xh, off, 64);
                }
                // 128 * sum Q^2 / 65536 = 128 hh + hl + ll / 512  (< 2^31; the floor only lowers the bound)
                // (wave-uniform after the butterflies: scalars, and scalar branches on x_any / x_hi below)
                s_xl = __builtin_amdgcn_readfirstlane(s_xl);
                s_xh = __builtin_amdgcn_readfirstlane(s_xh);
                q16_w = __builtin_amdgcn_readfirstlane(128 * s_hh + s_hl + (s_ll >> 9));
                x_any = (s_xl | s_xh) != 0;
                x_hi = s_xh != 0;
                x_const = 255 * s_xh;
                pf_slack_q = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.001f * sqrtf(dq_sq))));
                pf_bonus = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(0.999f * bonus)));
                __syncthreads();
                qp_h = *reinterpret_cast<const uint4 *>(s_q8 + 4 * (lane & 7));
                qp_l = *reinterpret_cast<const uint4 *>(s_q8 + 32 + 4 * (lane & 7));
                qp_x = *reinterpret_cast<const uint4 *>(s_q8 + 64 + 4 * (lane & 7));
            }
        }

        RSet<NCH> R;
#pragma unroll
        for (int cc = 0; cc < NCH; cc++)
            R.r[cc] = ~0ull;
        int n = 1;     // entries in R (== topResults.size())
        int ntail = 0; // evicted entries whose distance still equals the lower bound (SPILL: in LDS, or ALL in the bitmap)
        uint32_t tail_db = 0; // SPILL: that distance's bits
        {
            // hnswalg.cpp:56-62: seed with the enter point
            const float d0 = l2_ref_order_quad(g.vectors + (size_t)g.enterpoint * g.d, s_q, g.d, lane & 3);
            bool ub = false;
            if (lane == 0) {
                R.r[0] = mk_key(d0, enc_id(g.enterpoint, CK ? (uint32_t)g.counts[g.enterpoint] : 0u));
                (void)visit_test_and_set<TAGW, NB>(g.enterpoint, vt, bm, ub);
            }
            __syncthreads();
        }
        bool used_bitmap = false;
        // survivors of the last pass, entered into the visited set at the top of the next expansion
        bool pend = false, pend_any = false, pend_full = false;
        uint32_t pend_id = 0;
        if (STAMPS) {
            const unsigned long long t = walk_stamp();
            st_acc[4] += t - st_t;
            st_t = t;
        }

        for (;;) {
            // ---- candidateSet.top(): first unexpanded entry of R, ties -> largest id; tail joins at dist == max
            const int first = R.first_unexpanded(n, lane);
            int pick = -1;      // index in R, or
            int pick_tail = -1; // index in tail
            uint32_t pick_id = 0;
            if (first >= 0) {
                const unsigned long long kfirst = R.get(first);
                const uint32_t db = key_dist_bits(kfirst);
                // equal distances sit next to each other: only if the following entry shares this one's distance
                // can a larger id with the same distance exist (the full search is the rare path)
                pick = first;
                pick_id = key_id(kfirst);
                if (first + 1 < n && key_dist_bits(R.get(first + 1)) == db) {
                    pick = R.last_unexpanded_with(db, first, n, lane);
                    pick_id = key_id(R.get(pick));
                }
                if (ntail > 0 && db == key_dist_bits(R.get(n - 1))) {
                    // tail entries share this distance; the larger id pops first
                    if (SPILL && spill.count > 0) {
                        const uint32_t sid = spill.peek_max(lane);
                        const uint32_t senc = enc_id(sid, CK ? (uint32_t)g.counts[sid] : 0u);
                        if (senc > pick_id) {
                            pick_id = senc;
                            pick_tail = kTailCap; // "in the bitmap"
                        }
                    } else {
                        for (int t = 0; t < ntail; t++)
                            if (key_id(tail[t]) > pick_id) {
                                pick_id = key_id(tail[t]);
                                pick_tail = t;
                            }
                    }
                    if (pick_tail >= 0)
                        pick = -1;
                }
            } else if (SPILL && spill.count > 0) {
                const uint32_t sid = spill.peek_max(lane);
                pick_id = enc_id(sid, CK ? (uint32_t)g.counts[sid] : 0u);
                pick_tail = kTailCap;
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
            } else if (SPILL && pick_tail == kTailCap) {
                spill.remove(dec_id(pick_id), lane);
                ntail--;
            } else {
                __syncthreads();
                if (lane == 0)
                    tail[pick_tail] = tail[ntail - 1];
                ntail--;
                __syncthreads();
            }

            if (STAMPS) {
                const unsigned long long t = walk_stamp();
                st_acc[0] += t - st_t;
                st_t = t;
                st_acc[5] += 1;
            }
            // ---- expand: links, visited test-and-set, distances (hnswalg.cpp:72-91)
            const uint32_t node = dec_id(pick_id);
            int cnt;
            uint32_t nb = 0, nb_enc = 0; // neighbour id, and the "id" its key will carry
            if constexpr (CK) {
                cnt = (int)(pick_id & 127u); // scalar: known before anything of the record is read
                if (lane < g.maxM) {
                    const uint32_t raw = g.links_c[(size_t)node * g.maxM + lane];
                    nb = raw & 0xffffffu;
                    nb_enc = (nb << 7) | (raw >> 24);
                }
            } else {
                cnt = g.counts[node];
                if (lane < g.maxM)
                    nb = g.links[(size_t)node * g.maxM + lane]; // issued together with the count
                nb_enc = nb;
            }
            // neighbour byte rows of this node (exact rejection filter, below): rows 8i + (lane >> 3), 16-byte
            // chunk lane & 7 -- four 1 KiB wave reads, in flight together with the link list
            const bool filter_now = prefilter && n == ef;
            uint4 nw[4] = {};
            uint32_t row_rr = 0; // sum of bytes^2 of the row this lane will judge (row 8 * (2 bit2 + bit0) + group)
            if (filter_now && inline_rows) {
                if constexpr (CK) {
                    // rows below the count only: a lane whose row is padding re-reads the last real row's chunk (same
                    // cache lines, no branch; nobody reads its verdict), an 8-row group wholly beyond it is skipped
                    const uint4 *nbr = reinterpret_cast<const uint4 *>(g.nbrows + (size_t)node * g.nb_rows * 128) + (lane & 7);
                    const int last = max(cnt - 1, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (8 * i < cnt)
                            nw[i] = nbr[min(8 * i + (lane >> 3), last) * 8];
                } else {
                    const uint4 *nbr = reinterpret_cast<const uint4 *>(g.nbrows + (size_t)node * g.nb_rows * 128) + lane;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        nw[i] = nbr[i * 64];
                }
                if constexpr (ROWNORMS)
                    row_rr = g.nbnorms[(size_t)node * g.nb_rows + judged_row];
            }
            // With the neighbour-row filter on (and the LDS set), only the rows the filter lets through are ENTERED
            // into the visited set: a row it rejects has dist >= bound > max(topResults), the maximum never grows
            // and the bound is a function of (row, query) alone, so the row is rejected again whenever it comes
            // back -- marking it visited (hnswalg.cpp:80-82) changes nothing.  The set then holds ~5 ids per
            // expansion instead of ~27 (no bucket overflows), the membership
            // test is a plain LDS read, and the insertions of a pass's survivors wait until the NEXT expansion's
            // loads are in flight (just above the test, which must see them) -- off the dependent chain.  Needs
            // link lists without repeated ids (checked at upload): a repeated survivor would be entered twice.
            constexpr bool LATE = FMODE == 3 && LDSVIS; // the launcher picks it only for GraphTables::links_unique
            const bool filter_first = LATE && filter_now;
            bool fresh = false, seen = false, full_now = false;
            if constexpr (LATE) {
                if (pend_any) {
                    if (pend)
                        visit_insert_absent<TAGW, NB>(pend_id, pend_full, vt, bm, used_bitmap);
                    pend_any = false;
                }
                if (lane < cnt) {
                    if (filter_first)
                        seen = visit_lookup<TAGW, NB>(nb, vt, bm, full_now);
                    else
                        fresh = visit_test_and_set<TAGW, NB>(nb, vt, bm, used_bitmap);
                }
            } else {
                if (lane < cnt)
                    fresh = visit_test_and_set<TAGW, NB>(nb, vt, bm, used_bitmap);
            }
            unsigned long long mask = __ballot(fresh);
            int nfresh = __popcll(mask);
            // the set's last entry: read once per expansion, for the filter and for the first pass below (the set
            // does not change in between)
            const unsigned long long last_now = R.get(n - 1);
            if (filter_now && inline_rows) {
                if (filter_first)
                    fresh = lane < cnt && !seen;
                // (see the gather form below for the bound)  Link lane L is row L of the node's block.
                const float worst = __uint_as_float(key_dist_bits(last_now));
                // hi / lo / XL planes: this lane's 16 bytes of each live in registers for the whole query (qp_*, set
                // where the planes are built); XH, rarely non-zero, is read when needed
                const uint32_t qh[4] = {qp_h.x, qp_h.y, qp_h.z, qp_h.w}, ql[4] = {qp_l.x, qp_l.y, qp_l.z, qp_l.w};
                const uint32_t xl[4] = {qp_x.x, qp_x.y, qp_x.z, qp_x.w};
                uint32_t xh[4] = {0u, 0u, 0u, 0u};
                if (x_hi) {
                    const uint4 b = *reinterpret_cast<const uint4 *>(s_q8 + 96 + 4 * (lane & 7));
                    xh[0] = b.x, xh[1] = b.y, xh[2] = b.z, xh[3] = b.w;
                }
                const int cnt_s = __builtin_amdgcn_readfirstlane(cnt);
                for (int rb = 0;;) {
                    int S[4], X[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        // rows at and beyond the link count are zero padding whose verdicts nobody reads (mean degree
                        // 21 of 32 slots): a scalar branch over their arithmetic (IVFHNSW_WALK_SKIPPAD, A/B knob)
                        if ((CK || g.skip_padding) && rb + 8 * i >= cnt_s) {
                            S[i] = 0;
                            X[i] = 0;
                            continue;
                        }
                        // 128 * (row.row - 2 (Q/256).row), exact in integers; 128 * Q.Q / 65536 joins below
                        const uint32_t ws[4] = {nw[i].x, nw[i].y, nw[i].z, nw[i].w};
                        uint32_t hr = 0, rr = 0;
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            hr = __builtin_amdgcn_udot4(qh[c], ws[c], hr, false);
                            if constexpr (!ROWNORMS)
                                rr = __builtin_amdgcn_udot4(ws[c], ws[c], rr, false);
                        }
                        uint32_t lr = hr << 8; // the lo plane's sum accumulates on top of 256 * hr
#pragma unroll
                        for (int c = 0; c < 4; c++)
                            lr = __builtin_amdgcn_udot4(ql[c], ws[c], lr, false);
                        // this lane's 16 bytes of row rb + 8i + (lane >> 3); with ROWNORMS 128 * row.row
                        // (GraphTables::nbnorms, computed at upload) joins after the group sum instead
                        S[i] = (int)(128u * rr - lr);
                        X[i] = 0;
                        if (x_any) { // wave-uniform (scalar): SIFT-like queries only ever leave the range downwards
                            uint32_t xlr = 0, xhr = 0;
#pragma unroll
                            for (int c = 0; c < 4; c++)
                                xlr = __builtin_amdgcn_udot4(xl[c], ws[c], xlr, false);
                            if (x_hi) {
                                asm volatile("" ::); // a real (scalar) branch: hipcc otherwise computes and selects
#pragma unroll
                                for (int c = 0; c < 4; c++)
                                    xhr = __builtin_amdgcn_udot4(xh[c], ws[c], xhr, false);
                            }
                            X[i] = (int)(xlr - xhr);
                        }
                    }
                    // Group G = lane >> 3 holds rows rb + 8i + G (i = 0..3), 16 bytes per lane.  oct_sum4 leaves in
                    // lane 8G + ii the complete sum of row i = 2 * bit2(ii) + bit0(ii) (a reduce-scatter: 4 DPP adds
                    // instead of 12 for four full group sums); that lane evaluates the bound, one ballot collects
                    // the verdicts of the lanes with bit1(ii) = 0, and link lane L (row L - rb = 8i + G) reads bit
                    // 8G + 4 (i >> 1) + (i & 1) of it -- no cross-lane traffic through the LDS crossbar.
                    int mine, minex = 0;
                    if (x_any)
                        oct_sum4x2(S, X, lane, mine, minex);
                    else
                        mine = oct_sum4(S, lane);
                    // v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence: the factors 1 - 2^-10 cover it
                    mine += (int)(128u * row_rr);
                    const float m1 = fmaxf(
                        0.f, __builtin_amdgcn_sqrtf((float)(mine + q16_w) * 0.0078125f) * 0.9990234375f - pf_slack_q);
                    const float m2 = fmaf(m1, m1, fmaf((float)(minex + x_const), 0.9990234375f, pf_bonus));
                    const float m = fmaxf(0.f, __builtin_amdgcn_sqrtf(m2) * 0.9990234375f - pf_slack) * g.q_step;
                    const float lb = m * m * 0.9990234375f;
                    const unsigned long long dropm = __ballot(lb > worst) & 0x3333333333333333ull; // lanes with bit 1 clear
                    const int rel = lane - rb; // this link lane's row within the batch
                    const int ri = rel >> 3;
                    if (rel >= 0 && rel < 32 && ((dropm >> (8 * (rel & 7) + 4 * (ri >> 1) + (ri & 1))) & 1ull))
                        fresh = false;
                    rb += 32;
                    if (rb >= cnt)
                        break;
                    if constexpr (CK) {
                        const uint4 *nbr =
                            reinterpret_cast<const uint4 *>(g.nbrows + (size_t)node * g.nb_rows * 128) + (lane & 7);
                        const int last = max(cnt - 1, 0);
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (rb + 8 * i < cnt)
                                nw[i] = nbr[min(rb + 8 * i + (lane >> 3), last) * 8];
                    } else {
                        const uint4 *nbr =
                            reinterpret_cast<const uint4 *>(g.nbrows + ((size_t)node * g.nb_rows + rb) * 128) + lane;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            nw[i] = nbr[i * 64];
                    }
                    if constexpr (ROWNORMS)
                        row_rr = g.nbnorms[(size_t)node * g.nb_rows + rb + judged_row];
                }
                mask = __ballot(fresh);
                nfresh = __popcll(mask);
                if (filter_first) {
                    pend = fresh;
                    pend_full = full_now;
                    pend_id = nb;
                    pend_any = nfresh > 0;
                }
            } else if (filter_now && nfresh > 0) {
                // ---- exact rejection filter.  Once the set is full a row is admitted only if its distance is
                // below the current maximum (hnswalg.cpp:93), and the maximum never grows.  A lower bound of
                // the row's float distance that already exceeds it settles the row without reading its d floats:
                //   ||q - x|| >= q_step * (||q' - bytes|| - errc),
                // shrunk by 2^-10 twice, which covers every rounding on either side (d <= 2048: the float sums
                // are within (d+2) * 2^-24 of the real ones).  Rejected rows stay marked visited, as in the
                // reference; everything else takes the float path below unchanged.
                const float worst = __uint_as_float(key_dist_bits(last_now));
                const int myrow = __popcll(mask & ((1ull << lane) - 1ull)); // row index of this link lane
                for (int base = 0; base < nfresh; base += 32) {
                    const int r = base + (lane >> 1);
                    const bool active = r < nfresh;
                    const int src = active ? nth_set_bit(mask, r) : 0;
                    const uint32_t nbq = (uint32_t)__shfl((int)nb, src, 64);
                    float S = 0.f;
                    if (active)
                        S = byte_row_dist_half(g.qrows + (size_t)nbq * g.d, s_qp, g.d, lane & 1);
                    S += __shfl_xor(S, 1, 64);
                    const float m = fmaxf(0.f, __builtin_amdgcn_sqrtf(S) * 0.9990234375f - pf_slack) * g.q_step;
                    const float lb = m * m * 0.9990234375f;
                    const unsigned long long drop = __ballot(active && lb > worst);
                    const int rel = myrow - base;
                    if (fresh && rel >= 0 && rel < 32 && ((drop >> (2 * rel)) & 1ull))
                        fresh = false;
                }
                mask = __ballot(fresh);
                nfresh = __popcll(mask);
            }
            if (STAMPS) {
                const unsigned long long t = walk_stamp();
                st_acc[1] += t - st_t;
                st_t = t;
            }

            // ---- distances, 8 rows per pass (eight lanes per row), then admissions in link order
            unsigned long long rest = mask; // link lanes not yet dealt to a pass
            for (int base = 0; base < nfresh && ntail >= 0; base += 8) {
                const int r = base + (lane >> 3);
                const bool active = r < nfresh;
                const int src = take8_set_bits(rest, lane); // the r-th fresh link lane (0 beyond the last)
                const uint32_t nbe = (uint32_t)__shfl((int)nb_enc, src, 64); // what the key carries (id with its count)
                const uint32_t nbq = dec_id(nbe);
                float dq = 0.f;
                if (active) {
                    if constexpr (QREG)
                        dq = g.d == 128  ? l2_ref_order_oct_regs<16>(g.vectors + (size_t)nbq * 128, q_reg, lane & 7)
                             : g.d == 96 ? l2_ref_order_oct_regs<12>(g.vectors + (size_t)nbq * 96, q_reg, lane & 7)
                                         : l2_ref_order_oct(g.vectors + (size_t)nbq * g.d, s_q, g.d, lane & 7);
                    else
                        dq = l2_ref_order_oct(g.vectors + (size_t)nbq * g.d, s_q, g.d, lane & 7);
                }
                if (STAMPS) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long t = walk_stamp();
                    st_acc[2] += t - st_t;
                    st_t = t;
                }
                // Rows that cannot be admitted are dropped wave-wide before the sequential part: once the
                // set is full its maximum only decreases, so a row failing 'top > dist' (hnswalg.cpp:93)
                // against the current maximum fails against every later one too.
                unsigned long long topk = base == 0 ? last_now : R.get(n - 1); // the set's last entry, carried through the loop
                const float top0 = __uint_as_float(key_dist_bits(topk));
                // (first lane of each row's group, rows that exist: constant and scalar masks instead of per-lane tests)
                unsigned long long cand = (n < ef ? ~0ull : __ballot(top0 > dq)) & 0x0101010101010101ull &
                                          lanes_below(8 * (nfresh - base), 0);
                if (STAMPS) {
                    st_acc[6] += (unsigned long long)__popcll(__ballot(active && (lane & 7) == 0));
                    st_acc[7] += (unsigned long long)__popcll(cand);
                }
                if (NCH <= 4 && merge_on && n == ef && ntail == 0 && cand) {
                    // ---- all admissions of the pass in one step.  With the set full, inserting the pass's
                    // candidates one by one in link order (below) leaves the ef smallest keys of set + candidates,
                    // PROVIDED the ef-th and (ef+1)-th of the merged order differ in distance: then every loser has
                    // a distance above the final maximum, so no strict-'<' tie decided anything and nothing can
                    // wait in the tail.  So: rank every candidate among the set and among the candidates, every
                    // set entry among the candidates, scatter through LDS, look at the boundary; a tie there
                    // leaves the registers untouched and takes the sequential path.
                    const bool is_cand = (cand >> lane) & 1ull;
                    const unsigned long long Kc = mk_key(dq, nbe);
                    int dn[NCH]; // candidates ABOVE this lane's set entries (an add-with-carry per compare)
#pragma unroll
                    for (int cc = 0; cc < NCH; cc++)
                        dn[cc] = 0;
                    const int ncand = __popcll(cand);
                    int cand_below = 0, set_below = 0;
                    for (unsigned long long cm = cand; cm; cm &= cm - 1) {
                        const int b = __ffsll((long long)cm) - 1;
                        const unsigned long long Kj = readlane_u64(Kc, b); // lane b's key, built once per lane above
                        int below = 0;
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++) {
                            // (entries at and beyond ef are left-overs: their dn is never used, the mask keeps
                            // them out of the count)
                            const bool lt = (R.r[cc] & ~1ull) < Kj;
                            below += __popcll(__ballot(lt) & lanes_below(ef, cc));
                            dn[cc] += lt ? 1 : 0;
                        }
                        cand_below += Kj < Kc ? 1 : 0;
                        set_below = lane == b ? below : set_below;
                    }
                    unsigned long long *mb = tail;
#pragma unroll
                    for (int cc = 0; cc < NCH; cc++)
                        if (cc * 64 + lane < ef)
                            mb[cc * 64 + lane + (ncand - dn[cc])] = R.r[cc];
                    if (is_cand)
                        mb[set_below + cand_below] = Kc;
                    __syncthreads();
                    // boundary and merged entries read together: one LDS round trip, the verdict picks afterwards
                    const unsigned long long b_lo = mb[ef - 1], b_hi = mb[ef];
                    unsigned long long merged[NCH];
#pragma unroll
                    for (int cc = 0; cc < NCH; cc++)
                        merged[cc] = cc * 64 + lane < ef ? mb[cc * 64 + lane] : R.r[cc];
                    const bool clean = key_dist_bits(b_lo) != key_dist_bits(b_hi);
                    if (clean) {
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++)
                            R.r[cc] = merged[cc];
                        if (STAMPS)
                            st_acc[8] += (unsigned long long)__popcll(__ballot(is_cand && set_below + cand_below < ef));
                        cand = 0;
                    }
                    __syncthreads();
                }
                while (cand) { // hnswalg.cpp:93-103, in link order
                    const int b = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const float dj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dq), b));
                    const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)nbe, b);
                    const unsigned long long oldtop = topk;
                    if (!(__uint_as_float(key_dist_bits(oldtop)) > dj || n < ef))
                        continue;
                    const unsigned long long K = mk_key(dj, idj);
                    const bool full = n == ef;
                    if (STAMPS)
                        st_acc[8] += 1;
                    R.insert_sorted(K, lane);
                    if (!full)
                        n++;
                    topk = R.get(n - 1);
                    // bookkeeping of candidates that left topResults but may still be popped
                    const uint32_t newmax = key_dist_bits(topk);
                    if (ntail > 0 && (SPILL ? tail_db : key_dist_bits(tail[0])) != newmax) {
                        ntail = 0; // lower bound moved below them: dead for good
                        if (SPILL && spill.count > 0)
                            spill.clear(lane);
                    }
                    if (full && !(oldtop & 1ull) && key_dist_bits(oldtop) == newmax) {
                        if (SPILL)
                            tail_db = newmax;
                        if ((!SPILL || spill.count == 0) && ntail < kTailCap) {
                            __syncthreads();
                            if (lane == 0)
                                tail[ntail] = oldtop;
                            ntail++;
                            __syncthreads();
                        } else if constexpr (SPILL) {
                            // the whole tail moves into the wavefront's global bitmap and stays there until it dies or drains
                            if (spill.count == 0) {
                                __syncthreads();
                                for (int t = 0; t < ntail; t++)
                                    spill.add(dec_id(key_id(tail[t])), lane);
                                __syncthreads();
                            }
                            spill.add(dec_id(key_id(oldtop)), lane);
                            ntail++;
                        } else {
                            // more than kTailCap exact distance ties at the boundary: this form cannot finish the query --
                            // it goes on the redo list (the SPILL form walks it again, results written there)
                            if (lane == 0)
                                redo_list[atomicAdd(&redo_hdr[0], 1u)] = (uint32_t)q;
                            ntail = -1;
                            break;
                        }
                    }
                }
                if (STAMPS) {
                    const unsigned long long t = walk_stamp();
                    st_acc[3] += t - st_t;
                    st_t = t;
                }
            }
            if (ntail < 0)
                break;
        }

        if (SPILL && spill.count > 0)
            spill.clear(lane); // the tail bitmap is handed back all zero
        if (LDSVIS && __ballot(used_bitmap))
            bitmap_dirty = true;
        // searchKnn pops down to nprobe (hnswalg.cpp:229-233); IndexIVF_HNSW.cpp:249-259 unloads nearest first
        if (ntail >= 0) { // (ntail < 0: on the redo list, the SPILL form writes this query's rows)
#pragma unroll
            for (int cc = 0; cc < NCH; cc++) {
                const int i = cc * 64 + lane;
                if (i < nprobe) {
                    const bool have = i < n;
                    coarse_ids[(size_t)q * nprobe + i] = have ? dec_id(key_id(R.r[cc])) : 0xffffffffu;
                    coarse_dists[(size_t)q * nprobe + i] = have ? __uint_as_float(key_dist_bits(R.r[cc])) : 0.f;
                }
            }
        }
    }
    if constexpr (SPILL) {
        // The last wavefront out leaves the walk's counters as the next launch expects them -- the fast form's query
        // counter (the word before the header), the list's length, this launch's own counter and the exit count itself --
        // so that launches need no memset between them (it was ~8 us of every step's critical path).
        if (lane == 0) {
            __threadfence();
            if (atomicAdd(&redo_hdr[2], 1u) == gridDim.x - 1) {
                redo_hdr[-1] = 0;
                redo_hdr[0] = 0;
                redo_hdr[1] = 0;
                redo_hdr[2] = 0;
            }
        }
    }
    if (SPILL && redo_ran) { // the redo form runs on bitmap-only visited sets inside scratch the fast forms expect to find zero
        uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
        for (size_t w = lane; w < vwords / 4; w += 64)
            bm4[w] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (LDSVIS && bitmap_dirty) { // leave the overflow store as it was found
        uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
        for (size_t w = lane; w < vwords / 4; w += 64)
            bm4[w] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (STAMPS && stamp_out && lane == 0)
        for (int i = 0; i < 9; i++)
            atomicAdd(&stamp_out[i], st_acc[i]);
}

// one 64-thread block per (node, 8 rows): lane -> row (lane >> 3), 16-byte chunk (lane & 7)
// ... and the rows' squared norms (sum of bytes^2, u32 [n][nb_rows]): the walk's filter needs them per row, and a
// dword load per row is cheaper there than sixteen v_dot4 of the row with itself
__global__ __launch_bounds__(64) void build_nbrows_kernel(GraphTables g, uint8_t *__restrict__ nbrows,
                                                          uint32_t *__restrict__ nbnorms, int nb_rows,
                                                          uint32_t *__restrict__ links_c)
{
    const size_t node = blockIdx.x;
    const int cnt = g.counts[node];
    // ... and the link table whose entries carry the link count of the node they name (GraphTables::links_c)
    if ((int)threadIdx.x < g.maxM) {
        uint32_t e = 0;
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = g.links[node * g.maxM + threadIdx.x];
            e = id | ((uint32_t)g.counts[id] << 24);
        }
        links_c[node * g.maxM + threadIdx.x] = e;
    }
    const int cpr = g.d >> 4; // 16-byte chunks per source row (<= 8)
    for (int r0 = 0; r0 < nb_rows; r0 += 8) {
        const int r = r0 + (threadIdx.x >> 3), c = threadIdx.x & 7;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r < cnt && c < cpr)
            v = reinterpret_cast<const uint4 *>(g.qrows + (size_t)g.links[node * g.maxM + r] * g.d)[c];
        reinterpret_cast<uint4 *>(nbrows + (node * nb_rows + r) * 128)[c] = v;
        uint32_t rr = 0;
        rr = __builtin_amdgcn_udot4(v.x, v.x, rr, false);
        rr = __builtin_amdgcn_udot4(v.y, v.y, rr, false);
        rr = __builtin_amdgcn_udot4(v.z, v.z, rr, false);
        rr = __builtin_amdgcn_udot4(v.w, v.w, rr, false);
        rr = (uint32_t)oct_sum((int)rr);
        if (c == 0)
            nbnorms[node * nb_rows + r] = rr;
    }
}

hipError_t launch_build_nbrows(hipStream_t s, const GraphTables &g, uint8_t *nbrows, uint32_t *nbnorms, int nb_rows,
                               uint32_t *links_c)
{
    if (!g.qrows || g.d > 128 || (g.d & 15) || nb_rows < g.maxM || (nb_rows & 31) || g.n >= (1u << 24) || g.maxM > 64)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(build_nbrows_kernel, dim3(g.n), dim3(64), 0, s, g, nbrows, nbnorms, nb_rows, links_c);
    return hipGetLastError();
}

// how many wavefront slots the walk keeps resident for a given ef (sizes the visited bitmaps)
int coarse_slots_for(int ef)
{
    const int nch = (ef + 63) / 64;
    // Exactly the wavefronts that can be resident (256 CUs x 4 SIMDs x the waves per SIMD the kernel is built for;
    // its registers allow no more): blocks beyond that only start when a slot frees up, find the queue empty and
    // cost their prologue -- 1.205 vs 1.189 ms per 10 k queries with 8192 instead of 4096 blocks.
    // (builds for 5 and 6 waves per SIMD were measured -- 1.52 / 1.50 ms against 1.50 at 2^17 nodes, 1.51 / 1.68 against
    // 1.37 at 993 127 -- and removed in round 3 with their IVFHNSW_WALK_OCC knob)
    const int waves_per_simd = nch <= 8 ? 4 : 3;
    return 256 * 4 * waves_per_simd;
}

hipError_t launch_coarse(hipStream_t s, const GraphTables &g_in, const float *xq, int nq, int nprobe, int ef,
                         uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                         size_t visited_words_per_slot, int nslots, uint32_t *status, uint32_t *next_query,
                         size_t visited_bytes, bool *visited_zero, uint32_t *redo_list, uint32_t *tail_bitmaps,
                         int tail_slots, bool *counters_clean)
{
    if (nq == 0)
        return hipSuccess;
    GraphTables g = g_in;
    if (ef > 1024 || ef < 1 || nprobe > ef || g.maxM > 64 || g.n >= 0x80000000u || (visited_words_per_slot & 3) ||
        !redo_list || !tail_bitmaps || tail_slots < 1)
        return hipErrorInvalidValue;
    // next_query, and behind it the redo list's header (number of listed queries, the redo launch's counter, the redo
    // launch's exit count).  All four are zero between launches: the redo launch's last wavefront clears them; only a
    // handle that cannot vouch for that (first use, a failed launch) pays a memset.
    uint32_t *redo_hdr = next_query + 1;
    hipError_t e = hipSuccess;
    if (!counters_clean || !*counters_clean) {
        e = hipMemsetAsync(next_query, 0, 4 * sizeof(uint32_t), s);
        if (e != hipSuccess)
            return e;
    }
    if (counters_clean)
        *counters_clean = false;
    // Waves per SIMD the kernel is built for (register budget and visited-set size follow from it).  Measured
    // on MI355X (100M / 2^17-centroid workload, ef 80) with the LDS padded to hold 2, 3, 4 waves per SIMD
    // resident: 2.61, 1.90, 1.54 ms per 10 k queries = 0.48 + 4.26 / waves -- the walk is latency bound per wave.
    constexpr int occ = 4;
    const int nch = (ef + 63) / 64;
    const int occ_eff = 4;
    uint32_t nbk = (uint32_t)vis_buckets(occ_eff);
    // the LDS visited set needs 16-bit tags at most; IVFHNSW_WALK_VIS=bitmap forces the global bitmap (A/B runs)
    static const bool force_bitmap = [] {
        const char *e = getenv("IVFHNSW_WALK_VIS");
        return e && e[0] == 'b';
    }();
    int tagw = force_bitmap ? 0 : g.n <= 255u * nbk ? 8 : g.n <= 4095u * nbk ? 12 : g.n <= 65535u * nbk ? 16 : 0;
    // IVFHNSW_WALK_TAGW=10|12|16 forces a wider tag than the graph needs (tests: small graphs reach every form)
    static const int force_tagw = [] {
        const char *e = getenv("IVFHNSW_WALK_TAGW");
        const int v = e ? atoi(e) : 0;
        return (v == 10 || v == 12 || v == 16) ? v : 0;
    }();
    if (tagw && force_tagw > tagw)
        tagw = force_tagw;
    if ((tagw == 12 || tagw == 10) && occ_eff == 4 && g.n <= 1023u * (uint32_t)vis_buckets(4, 10) && force_tagw != 12) {
        tagw = 10;
        nbk = (uint32_t)vis_buckets(4, 10);
    } else if (tagw == 10) {
        tagw = 12; // the 10-bit form exists for 4 waves per SIMD only
    }
    const size_t shm = (size_t)(g.d + ((g.qrows && !g.nbrows) ? g.d : 0)) * sizeof(float) + 512 +
                       (size_t)(kTailCap + ((ef <= 256 && ef + 8 > kTailCap) ? ef + 8 - kTailCap : 0)) *
                           sizeof(unsigned long long) +
                       (tagw ? (size_t)nbk * sizeof(unsigned long long) : 0);
    const int fmode = g.nbrows ? (g.links_unique && tagw ? 3 : 2) : g.qrows ? 1 : 0;
    // the global bitmaps: with the LDS set (tagw != 0) they are the overflow store, zero between launches -- cleared
    // here once if the caller cannot vouch for them; the bitmap-only forms wipe per query and leave them used
    g.visited_clean = 0;
    if (tagw && visited_zero) {
        if (!*visited_zero) {
            e = hipMemsetAsync(visited_scratch, 0, visited_bytes, s);
            if (e != hipSuccess)
                return e;
        }
        g.visited_clean = 1;
        *visited_zero = true;
    } else if (visited_zero) {
        *visited_zero = false;
    }
#define IVFHNSW_WALK_F(N, W, T, F)                                                                                    \
    hipLaunchKernelGGL((hnsw_walk_kernel<N, W, T, F>), dim3(nslots), dim3(64), shm, s, g, xq, nq, nprobe, ef,         \
                       coarse_ids, coarse_dists, visited_scratch, visited_words_per_slot, status, next_query, redo_hdr, redo_list)
#define IVFHNSW_WALK_T(N, W, T)          \
    do {                                 \
        if (fmode == 3)                  \
            IVFHNSW_WALK_F(N, W, T, 3);  \
        else if (fmode == 2)             \
            IVFHNSW_WALK_F(N, W, T, 2);  \
        else if (fmode == 1)             \
            IVFHNSW_WALK_F(N, 4, T, 1);  \
        else                             \
            IVFHNSW_WALK_F(N, 4, T, 0);  \
    } while (0)
#define IVFHNSW_WALK(N, W)             \
    do {                               \
        if (tagw == 8)                 \
            IVFHNSW_WALK_T(N, W, 8);   \
        else if (tagw == 10)           \
            IVFHNSW_WALK_T(N, 4, 10);  \
        else if (tagw == 12)           \
            IVFHNSW_WALK_T(N, W, 12);  \
        else if (tagw == 16)           \
            IVFHNSW_WALK_T(N, W, 16);  \
        else                           \
            IVFHNSW_WALK_T(N, W, 0);   \
    } while (0)
#define IVFHNSW_WALK_N(N) IVFHNSW_WALK(N, 4)
    static const bool stamps = [] {
        const char *e = getenv("IVFHNSW_WALK_STAMPS");
        return e && atoi(e) == 1;
    }();
    if (stamps && nch == 2 && ((tagw == 8 && fmode == 2) || (tagw == 10 && fmode == 3)) && occ == 4) {
        // diagnostic build: per-segment cycle sums, printed when the process exits (never used for timing claims)
        static unsigned long long *d_st = nullptr;
        if (!d_st) {
            (void)hipMalloc(&d_st, 9 * sizeof(unsigned long long));
            (void)hipMemset(d_st, 0, 9 * sizeof(unsigned long long));
            atexit([] {
                unsigned long long h[9];
                if (hipMemcpy(h, d_st, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                    const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
                    fprintf(stderr, "[walk stamps] expansions %llu; cycles/expansion: select %.0f, links+visited %.0f, "
                                    "distances %.0f, admissions %.0f; prologue share %.1f%%; total wave-cycles %.3g\n",
                            h[5], (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5],
                            100.0 * h[4] / tot, tot);
                    fprintf(stderr, "[walk stamps] rows evaluated/expansion %.2f, passing the wave-wide test %.2f, admitted %.2f\n",
                            (double)h[6] / h[5], (double)h[7] / h[5], (double)h[8] / h[5]);
                }
            });
        }
        if (tagw == 8)
            hipLaunchKernelGGL((hnsw_walk_kernel<2, 4, 8, 2, true>), dim3(nslots), dim3(64), shm, s, g, xq, nq, nprobe, ef,
                               coarse_ids, coarse_dists, visited_scratch, visited_words_per_slot, status, next_query, redo_hdr,
                               redo_list, nullptr, d_st);
        else // the form graphs of about a million nodes take (10-bit tags, late visited entry): round 3
            hipLaunchKernelGGL((hnsw_walk_kernel<2, 4, 10, 3, true>), dim3(nslots), dim3(64), shm, s, g, xq, nq, nprobe, ef,
                               coarse_ids, coarse_dists, visited_scratch, visited_words_per_slot, status, next_query, redo_hdr,
                               redo_list, nullptr, d_st);
        return hipGetLastError();
    }
    if (nch <= 1) {
        IVFHNSW_WALK_N(1);
    } else if (nch <= 2) {
        IVFHNSW_WALK_N(2);
    } else if (nch == 3) { // efSearch 129..192 (the reference's DEEP1B presets use 130)
        IVFHNSW_WALK_N(3);
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
#undef IVFHNSW_WALK_F
    e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    e = launch_coarse_redo(s, g, xq, nq, nprobe, ef, coarse_ids, coarse_dists, visited_scratch, visited_words_per_slot,
                           status, redo_hdr, redo_list, tail_bitmaps, tail_slots);
    if (e == hipSuccess && counters_clean)
        *counters_clean = true;
    return e;
}

// The queries a fast form of the walk (this file's, or the latency form of kernels_hnsw_lat.hip) could not finish: more
// than kTailCap exact ties at the efSearch boundary.  The plainest exact form -- no rejection filter, global visited
// bitmaps -- with the tail's overflow in global bitmaps (SPILL).  Always launched (the host cannot know the list's
// length without a round trip): a few wavefronts that find the list empty and leave, ~4 us of stream time.
hipError_t launch_coarse_redo(hipStream_t s, const GraphTables &g_in, const float *xq, int nq, int nprobe, int ef,
                              uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                              size_t visited_words_per_slot, uint32_t *status, uint32_t *redo_hdr, uint32_t *redo_list,
                              uint32_t *tail_bitmaps, int tail_slots)
{
    GraphTables g = g_in;
    g.visited_clean = 0;
    const int nch = (ef + 63) / 64;
    const int slots = std::min(std::min(nq, tail_slots), 64);
    const size_t shm = (size_t)g.d * sizeof(float) + 512 +
                       (size_t)(kTailCap + ((ef <= 256 && ef + 8 > kTailCap) ? ef + 8 - kTailCap : 0)) * sizeof(unsigned long long);
#define IVFHNSW_REDO(N)                                                                                               \
    hipLaunchKernelGGL((hnsw_walk_kernel<N, 4, 0, 0, false, true>), dim3(slots), dim3(64), shm, s, g, xq, nq, nprobe, ef, \
                       coarse_ids, coarse_dists, visited_scratch, visited_words_per_slot, status, redo_hdr + 1, redo_hdr, \
                       redo_list, tail_bitmaps)
    if (nch <= 1)
        IVFHNSW_REDO(1);
    else if (nch <= 2)
        IVFHNSW_REDO(2);
    else if (nch == 3)
        IVFHNSW_REDO(3);
    else if (nch <= 4)
        IVFHNSW_REDO(4);
    else if (nch <= 8)
        IVFHNSW_REDO(8);
    else
        IVFHNSW_REDO(16);
#undef IVFHNSW_REDO
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
