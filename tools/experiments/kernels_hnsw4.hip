// On-device HNSW coarse walk, FOUR queries per wavefront: same algorithm and the same exact results as
// kernels_hnsw.hip (hnswlib/hnswalg.cpp:48-109,227-234 + IndexIVF_HNSW.cpp:249-259), re-mapped so that each query
// owns one DPP row of 16 lanes instead of a whole wavefront.
//
// Why: counters of the one-query-per-wave kernel (profiles/r01_summary.md) show it bound by per-wave instruction
// issue (~73 k instructions per query, most of them set bookkeeping executed by 64 lanes for an 80-entry set) and
// by three dependent round trips per expansion.  With a row per query every bookkeeping instruction serves four
// searches, and four searches' link rows, visited bits and vector rows are in flight per wave.
//
// Per query (row r = lane / 16): the result set R is sorted by (dist, id) with an "expanded" bit, entry i in lane
// i % 16 of the row, register i / 16; selection / admission / insertion are ballots (each row reads its own 16
// bits of the mask), ds_bpermute reads inside the row and DPP row_shr / row_ror shifts.  Distances of all four
// queries' unvisited neighbours are pooled: a lane quad per vector row, 16 rows per pass, whichever query they
// belong to.  The four rows run in lockstep; a row that finishes takes the next query from the atomic counter
// while the others keep walking.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

namespace {

constexpr int kTail4 = 16;   // evicted-but-still-poppable candidates kept per query (exact distance ties only)
constexpr int kMaxLinks = 64;

__device__ __forceinline__ unsigned long long mk_key4(float dist, uint32_t id)
{
    return ((unsigned long long)__float_as_uint(dist) << 32) | ((unsigned long long)id << 1);
}
__device__ __forceinline__ uint32_t kdist(unsigned long long k) { return (uint32_t)(k >> 32); }
__device__ __forceinline__ uint32_t kid(unsigned long long k) { return (uint32_t)(k & 0xffffffffu) >> 1; }

// the 16 bits of a wave ballot that belong to this lane's row
__device__ __forceinline__ uint32_t row_bits(bool pred, int rowbase)
{
    return (uint32_t)(__ballot(pred) >> rowbase) & 0xffffu;
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src)
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// DPP inside the 16-lane row: value of the lane below (lane 0 of the row keeps its own), and rotate right by one
__device__ __forceinline__ unsigned long long row_shr1(unsigned long long v)
{
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x111, 0xf, 0xf, false);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ unsigned long long row_ror1(unsigned long long v)
{
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x121, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x121, 0xf, 0xf, false);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}

template <int NREG> struct RowSet {
    unsigned long long r[NREG]; // entry i of this lane's query: lane i & 15 of the row, register i >> 4

    // entry idx (idx uniform inside the row, may differ between rows)
    __device__ __forceinline__ unsigned long long get(int idx, int rowbase) const
    {
        // mask arithmetic, not a select chain: hipcc turns a chain of selects on (idx >> 4) == c into a stack copy
        // of r[] with a dynamically indexed scratch load (a memory round trip per access)
        unsigned long long v = 0;
        const int reg = idx >> 4;
#pragma unroll
        for (int c = 0; c < NREG; c++)
            v |= r[c] & (0ull - (unsigned long long)(reg == c));
        return shfl_u64(v, rowbase + (idx & 15));
    }
    __device__ __forceinline__ void mark_expanded(int idx, int li, bool act)
    {
        const int reg = idx >> 4;
#pragma unroll
        for (int c = 0; c < NREG; c++)
            r[c] |= (unsigned long long)(act && reg == c && li == (idx & 15));
    }
    __device__ __forceinline__ int first_unexpanded(int n, int li, int rowbase) const
    {
        int first = -1;
#pragma unroll
        for (int c = 0; c < NREG; c++) {
            const uint32_t m = row_bits(c * 16 + li < n && !(r[c] & 1ull), rowbase);
            if (m && first < 0)
                first = c * 16 + (__ffs((int)m) - 1);
        }
        return first;
    }
    __device__ __forceinline__ int last_unexpanded_with(uint32_t db, int first, int n, int li, int rowbase) const
    {
        int last = first;
#pragma unroll
        for (int c = 0; c < NREG; c++) {
            const int i = c * 16 + li;
            const uint32_t m = row_bits(i < n && i >= first && kdist(r[c]) == db && !(r[c] & 1ull), rowbase);
            if (m)
                last = c * 16 + (31 - __clz((int)m));
        }
        return last;
    }
    __device__ __forceinline__ int rank_of(unsigned long long K, int n, int li, int rowbase) const
    {
        int pos = 0;
#pragma unroll
        for (int c = 0; c < NREG; c++)
            pos += __popc(row_bits(c * 16 + li < n && (r[c] & ~1ull) < K, rowbase));
        return pos;
    }
    // insert K at sorted position pos in the rows where act holds (others unchanged)
    __device__ __forceinline__ void insert_at(unsigned long long K, int pos, int li, bool act)
    {
#pragma unroll
        for (int c = NREG - 1; c >= 0; c--) {
            unsigned long long below = row_shr1(r[c]);
            if (c > 0) {
                const unsigned long long carry = row_ror1(r[c - 1]); // lane 0 receives lane 15 of the register below
                if (li == 0)
                    below = carry;
            }
            const int i = c * 16 + li;
            const unsigned long long nv = i < pos ? r[c] : (i == pos ? K : below);
            r[c] = act ? nv : r[c];
        }
    }
};

} // namespace

// One wavefront per block; it serves four queries at a time (row = lane / 16).
// dynamic LDS: float q[4][d] | u32 fid[4][64] | f32 fd[4][64] | u64 tail[4][kTail4] | u32 nf[4]
template <int NREG, int LPL /* links per lane = ceil(maxM / 16) */>
__global__ __launch_bounds__(64, 4) void hnsw_walk4_kernel(GraphTables g, const float *__restrict__ xq, int nq, int nprobe,
                                                          int ef, uint32_t *__restrict__ coarse_ids,
                                                          float *__restrict__ coarse_dists,
                                                          uint32_t *__restrict__ visited, size_t vwords,
                                                          uint32_t *__restrict__ status, uint32_t *__restrict__ next_query)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_q = reinterpret_cast<float *>(smem);
    uint32_t *s_fid = reinterpret_cast<uint32_t *>(s_q + 4 * g.d);
    float *s_fd = reinterpret_cast<float *>(s_fid + 4 * kMaxLinks);
    unsigned long long *s_tail = reinterpret_cast<unsigned long long *>(s_fd + 4 * kMaxLinks);
    uint32_t *s_nf = reinterpret_cast<uint32_t *>(s_tail + 4 * kTail4);

    const int lane = threadIdx.x;
    const int row = lane >> 4, li = lane & 15, rowbase = lane & 48;
    uint32_t *bm = visited + ((size_t)blockIdx.x * 4 + row) * vwords;
    float *my_q = s_q + row * g.d;
    unsigned long long *my_tail = s_tail + row * kTail4;

    RowSet<NREG> R;
#pragma unroll
    for (int c = 0; c < NREG; c++)
        R.r[c] = ~0ull;
    int q = -1;          // query of this row, -1 = idle
    bool drained = false; // the queue had nothing left for this row
    int n = 0, ntail = 0;
    bool bad = false; // more exact ties at the boundary than the tail can hold

    for (;;) {
        // ---------------------------------------------------------------- (0) idle rows take the next query
        const bool want = q < 0 && !drained;
        if (__ballot(want)) {
            int nqry = 0;
            if (want && li == 0)
                nqry = (int)atomicAdd(next_query, 1u);
            nqry = __shfl(nqry, rowbase, 64);
            const bool fresh_q = want && nqry < nq;
            if (want && !fresh_q)
                drained = true;
            if (fresh_q) {
                q = nqry;
                uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
                for (size_t w = li; w < vwords / 4; w += 16)
                    bm4[w] = make_uint4(0u, 0u, 0u, 0u);
                for (int i = li; i < g.d; i += 16)
                    my_q[i] = xq[(size_t)q * g.d + i];
#pragma unroll
                for (int c = 0; c < NREG; c++)
                    R.r[c] = ~0ull;
                n = 1;
                ntail = 0;
                bad = false;
            }
            __syncthreads(); // queries staged, bitmaps cleared before the atomics below
            // hnswalg.cpp:56-62: seed with the enter point (the first quad of the row evaluates the distance)
            float d0 = 0.f;
            if (fresh_q && li < 4)
                d0 = l2_ref_order_quad(g.vectors + (size_t)g.enterpoint * g.d, my_q, g.d, li);
            if (fresh_q && li == 0) {
                R.r[0] = mk_key4(d0, g.enterpoint);
                atomicOr(&bm[g.enterpoint >> 5], 1u << (g.enterpoint & 31));
            }
        }
        if (!__ballot(q >= 0))
            break;

        // ---------------------------------------------------------------- (1) candidateSet.top() per row
        bool act = q >= 0;
        int pick = -1, pick_tail = -1;
        uint32_t pick_id = 0;
        {
            const int first = R.first_unexpanded(n, li, rowbase);
            const uint32_t maxbits = kdist(R.get(act ? n - 1 : 0, rowbase));
            const uint32_t db = kdist(R.get(first >= 0 ? first : 0, rowbase));
            const int last = R.last_unexpanded_with(db, first >= 0 ? first : 0, first >= 0 ? n : 0, li, rowbase);
            const uint32_t last_id = kid(R.get(last >= 0 ? last : 0, rowbase));
            if (act && first >= 0) {
                pick = last;
                pick_id = last_id;
                if (db == maxbits && ntail > 0) { // tail entries share this distance; the larger id pops first
                    for (int t = 0; t < ntail; t++)
                        if (kid(my_tail[t]) > pick_id) {
                            pick_id = kid(my_tail[t]);
                            pick_tail = t;
                        }
                    if (pick_tail >= 0)
                        pick = -1;
                }
            } else if (act && ntail > 0) {
                pick_tail = 0;
                pick_id = kid(my_tail[0]);
                for (int t = 1; t < ntail; t++)
                    if (kid(my_tail[t]) > pick_id) {
                        pick_id = kid(my_tail[t]);
                        pick_tail = t;
                    }
            }
        }
        const bool finished = act && pick < 0 && pick_tail < 0; // candidateSet exhausted (hnswalg.cpp:64,67)
        if (finished) {
            // searchKnn pops down to nprobe (hnswalg.cpp:229-233); IndexIVF_HNSW.cpp:249-259 unloads nearest first
#pragma unroll
            for (int c = 0; c < NREG; c++) {
                const int i = c * 16 + li;
                if (i < nprobe) {
                    const bool have = i < n;
                    coarse_ids[(size_t)q * nprobe + i] = have ? kid(R.r[c]) : 0xffffffffu;
                    coarse_dists[(size_t)q * nprobe + i] = have ? __uint_as_float(kdist(R.r[c])) : 0.f;
                }
            }
            q = -1;
            act = false;
        }
        R.mark_expanded(pick >= 0 ? pick : 0, li, act && pick >= 0);
        __syncthreads();
        if (act && pick < 0 && li == 0)
            my_tail[pick_tail] = my_tail[ntail - 1];
        if (act && pick < 0)
            ntail--;

        // ---------------------------------------------------------------- (2) expand: links, visited test-and-set
        uint32_t nb[LPL];
        bool fr[LPL];
        {
            const int cnt = act ? (int)g.counts[pick_id] : 0;
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                const int j = k * 16 + li;
                nb[k] = 0;
                fr[k] = false;
                if (act && j < g.maxM)
                    nb[k] = g.links[(size_t)pick_id * g.maxM + j];
                if (j < cnt) {
                    const uint32_t bit = 1u << (nb[k] & 31);
                    const uint32_t old = atomicOr(&bm[nb[k] >> 5], bit);
                    fr[k] = !(old & bit);
                }
            }
        }
        // unvisited neighbours of each query, compacted in link order
        int nfresh = 0;
#pragma unroll
        for (int k = 0; k < LPL; k++) {
            const uint32_t m = row_bits(fr[k], rowbase);
            if (fr[k])
                s_fid[row * kMaxLinks + nfresh + __popc(m & ((1u << li) - 1u))] = nb[k];
            nfresh += __popc(m);
        }
        if (li == 0)
            s_nf[row] = (uint32_t)nfresh;
        __syncthreads();

        // ---------------------------------------------------------------- (3) distances, pooled over the 4 queries
        {
            const int f0 = (int)s_nf[0], f1 = f0 + (int)s_nf[1], f2 = f1 + (int)s_nf[2], f3 = f2 + (int)s_nf[3];
            for (int base = 0; base < f3; base += 16) {
                const int gi = base + (lane >> 2);
                if (gi < f3) {
                    const int qr = (gi >= f0) + (gi >= f1) + (gi >= f2);
                    const int slot = gi - (qr == 0 ? 0 : (qr == 1 ? f0 : (qr == 2 ? f1 : f2)));
                    const uint32_t nbq = s_fid[qr * kMaxLinks + slot];
                    const float dq = l2_ref_order_quad(g.vectors + (size_t)nbq * g.d, s_q + qr * g.d, g.d, lane & 3);
                    if ((lane & 3) == 0)
                        s_fd[qr * kMaxLinks + slot] = dq;
                }
            }
        }
        __syncthreads();

        // ---------------------------------------------------------------- (4) admissions in link order (hnswalg.cpp:93-103)
        // Pre-filter per row: once the set is full its maximum only decreases, so a neighbour failing
        // 'top > dist' against the current maximum fails against every later one too.
        unsigned long long cand = 0; // bit s = slot s of this row may still be admitted
        {
            const float top0 = __uint_as_float(kdist(R.get(act ? n - 1 : 0, rowbase)));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int s = k * 16 + li;
                const bool c = act && s < nfresh && (n < ef || top0 > s_fd[row * kMaxLinks + s]);
                cand |= (unsigned long long)row_bits(c, rowbase) << (16 * k);
            }
        }
        while (__ballot(cand != 0)) {
            const bool a = cand != 0;
            const int b = a ? __ffsll((long long)cand) - 1 : 0;
            cand &= cand - 1;
            const float dj = s_fd[row * kMaxLinks + b];
            const uint32_t idj = s_fid[row * kMaxLinks + b];
            const unsigned long long topk = R.get(a ? n - 1 : 0, rowbase);
            const float topd = __uint_as_float(kdist(topk));
            const bool adm = a && (topd > dj || n < ef);
            const unsigned long long K = mk_key4(dj, idj);
            const int pos = R.rank_of(K, adm ? n : 0, li, rowbase);
            const bool full = n == ef;
            R.insert_at(K, pos, li, adm);
            if (adm && !full)
                n++;
            // candidates that left topResults but may still be popped (distance equal to the new lower bound)
            const uint32_t newmax = kdist(R.get(adm ? n - 1 : 0, rowbase));
            __syncthreads();
            if (adm && ntail > 0 && kdist(my_tail[0]) != newmax)
                ntail = 0;
            if (adm && !bad && full && !(topk & 1ull) && kdist(topk) == newmax) {
                if (ntail < kTail4) {
                    if (li == 0)
                        my_tail[ntail] = topk;
                    ntail++;
                } else {
                    if (li == 0)
                        atomicOr(status, kStatusHnswTieOverflow);
                    bad = true;
                }
            }
            __syncthreads();
        }
        if (act && bad) {
            // tie overflow: give up on this query (status word is set, the host reports the batch invalid)
#pragma unroll
            for (int c = 0; c < NREG; c++) {
                const int i = c * 16 + li;
                if (i < nprobe) {
                    coarse_ids[(size_t)q * nprobe + i] = 0xffffffffu;
                    coarse_dists[(size_t)q * nprobe + i] = 0.f;
                }
            }
            q = -1;
            ntail = 0;
            bad = false;
        }
    }
}

int coarse4_waves_resident() { return 256 * 4 * 4; }

hipError_t launch_coarse4(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                          uint32_t *coarse_ids, float *coarse_dists, uint32_t *visited_scratch,
                          size_t visited_words_per_slot, int nwaves, uint32_t *status, uint32_t *next_query)
{
    if (nq == 0)
        return hipSuccess;
    if (ef > 256 || ef < 1 || nprobe > ef || g.maxM > kMaxLinks || g.n >= 0x80000000u || (visited_words_per_slot & 3))
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(next_query, 0, sizeof(uint32_t), s);
    if (e != hipSuccess)
        return e;
    const size_t shm = (size_t)4 * g.d * sizeof(float) + 4 * kMaxLinks * 8 + 4 * kTail4 * 8 + 4 * sizeof(uint32_t);
    const int nreg = (ef + 15) / 16;
    const int lpl = (g.maxM + 15) / 16;
#define IVFHNSW_W4(NR, LP)                                                                                       \
    hipLaunchKernelGGL((hnsw_walk4_kernel<NR, LP>), dim3(nwaves), dim3(64), shm, s, g, xq, nq, nprobe, ef,        \
                       coarse_ids, coarse_dists, visited_scratch, visited_words_per_slot, status, next_query)
#define IVFHNSW_W4_L(NR)       \
    do {                       \
        if (lpl <= 1)          \
            IVFHNSW_W4(NR, 1); \
        else if (lpl <= 2)     \
            IVFHNSW_W4(NR, 2); \
        else                   \
            IVFHNSW_W4(NR, 4); \
    } while (0)
    if (nreg <= 2)
        IVFHNSW_W4_L(2);
    else if (nreg <= 4)
        IVFHNSW_W4_L(4);
    else if (nreg <= 6)
        IVFHNSW_W4_L(6);
    else if (nreg <= 8)
        IVFHNSW_W4_L(8);
    else
        IVFHNSW_W4_L(16);
#undef IVFHNSW_W4_L
#undef IVFHNSW_W4
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
