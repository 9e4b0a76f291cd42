// The coarse walk for ONE query at a time (hnswlib searchKnn / searchBaseLayer, hnswalg.cpp:227-234, 48-109): the
// latency form.  The reference's drivers call search() once per query (tests/test_ivfhnsw_sift1b.cpp:193-208); the
// throughput walk (kernels_hnsw.hip) gives such a call one wavefront on an otherwise idle chip, and one wavefront
// issues an instruction every ~8 cycles: ~95 expansions x (two dependent memory round trips + ~700 instructions) =
// 0.29 ms, slower than one host core.  Here a WORKGROUP walks the query:
//
//   * wave 0 keeps the result set (walk_set.h, the very structure of the throughput walk), the visited set -- a plain
//     bitmap in LDS, 128 KB for up to 2^20 nodes: the CU is ours alone -- and runs admissions and selection;
//   * waves 1..4 are loaders: eight lanes per row, each wave eight of a node's <= 32 neighbour rows, the exact
//     8-accumulator distance in the reference's order (l2_ref_order_oct_regs, query components in registers);
//   * the rows come from a "fat" copy of the graph (GraphTables::fat): node i carries the FLOAT rows of its own
//     neighbours, maxM x d floats = 16 KB at d = 128 -- one round trip per expansion instead of links -> rows, bought
//     with HBM capacity (16 GB at the reference's 993 127 centroids; built on request, ivfhnsw_gpu_prepare_latency);
//   * the next node is predicted as soon as the distances of the current one are known -- the nearer of the set's
//     first unexpanded entry and the nearest admissible new candidate -- and the loaders fetch ITS rows while wave 0
//     is still inserting: a wrong guess (a tie, an eviction) costs one more round trip and never changes a result.
//
// Decisions are those of the reference: every neighbour is tested and marked visited (hnswalg.cpp:80-82), admitted
// in link order iff topResults is not full or dist < top (hnswalg.cpp:93-103), the candidate popped next is the
// smallest (dist, -id) not yet expanded, the walk ends when that distance exceeds the lower bound (hnswalg.cpp:67).
#include "ivfhnsw_kernels.h"
#include "device_common.h"
#include "walk_set.h"

#include <float.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

namespace {

constexpr int LAT_THREADS = 320; // wave 0 + four loader waves
constexpr uint32_t LAT_CMD_NONE = 0xffffffffu, LAT_CMD_EXIT = 0xfffffffeu;

__device__ __forceinline__ unsigned long long wave_min_u64_l(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

struct LatShared {
    float dist[2][32];     // distances of the node being fetched, double buffered
    uint32_t link[2][32];  // its link list
    uint32_t cnt[2];       // its link count
    uint32_t cmd;          // node the loaders fetch next (LAT_CMD_NONE: nothing, LAT_CMD_EXIT: done)
    uint32_t buf;          // buffer they fill
    unsigned long long tail[kTailCap];
};

// NJ = d / 8 (16: d = 128, 12: d = 96); NCH = registers of the result set (ef <= 64 * NCH)
template <int NCH, int NJ>
__global__ __launch_bounds__(LAT_THREADS) void hnsw_walk_lat_kernel(GraphTables g, const float *__restrict__ xq, int nq,
                                                                    int nprobe, int ef, uint32_t *__restrict__ coarse_ids,
                                                                    float *__restrict__ coarse_dists,
                                                                    uint32_t *__restrict__ status)
{
    constexpr int D = NJ * 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_lat[];
    LatShared *sh = reinterpret_cast<LatShared *>(smem_lat);
    uint32_t *bm = reinterpret_cast<uint32_t *>(smem_lat + ((sizeof(LatShared) + 15) & ~(size_t)15));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = blockIdx.x;
    const uint32_t bm_words4 = (g.n + 127u) / 128u; // uint4 words of the bitmap

    // ---- per query: wipe the visited bitmap, stage the query into registers (every lane: its column of eight)
    {
        uint4 *b4 = reinterpret_cast<uint4 *>(bm);
        for (uint32_t w = tid; w < bm_words4; w += LAT_THREADS)
            b4[w] = make_uint4(0u, 0u, 0u, 0u);
    }
    float q_reg[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        q_reg[i] = i < NJ ? xq[(size_t)q * D + 8 * i + (lane & 7)] : 0.f;
    if (tid == 0) {
        sh->cmd = LAT_CMD_NONE;
        sh->buf = 0;
    }
    __syncthreads();

    if (wave != 0) {
        // ---- loaders: B1 -> read the command -> fetch -> B2
        const int r = 8 * (wave - 1) + (lane >> 3); // the row this group of eight lanes evaluates
        for (;;) {
            __syncthreads(); // B1
            const uint32_t node = sh->cmd;
            const uint32_t b = sh->buf;
            if (node == LAT_CMD_EXIT)
                break;
            if (node != LAT_CMD_NONE) {
                const float *row = g.fat + ((size_t)node * 32 + r) * D;
                const float dq = l2_ref_order_oct_regs<NJ>(row, q_reg, lane & 7);
                if ((lane & 7) == 0)
                    sh->dist[b][r] = dq;
                if (wave == 1) {
                    if (lane < 32)
                        sh->link[b][lane] = lane < g.maxM ? g.links[(size_t)node * g.maxM + lane] : 0u;
                    if (lane == 32)
                        sh->cnt[b] = g.counts[node];
                }
            }
            __syncthreads(); // B2
        }
        return;
    }

    // ---- wave 0: the walk
    RSet<NCH> R;
#pragma unroll
    for (int cc = 0; cc < NCH; cc++)
        R.r[cc] = ~0ull;
    int n = 1, ntail = 0;
    unsigned long long *tail = sh->tail;
    {
        // hnswalg.cpp:56-62: seed with the enter point
        const float d0 = l2_ref_order_oct_regs<NJ>(g.vectors + (size_t)g.enterpoint * D, q_reg, lane & 7);
        const float d00 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(d0)));
        if (lane == 0) {
            R.r[0] = mk_key(d00, g.enterpoint);
            bm[g.enterpoint >> 5] |= 1u << (g.enterpoint & 31);
        }
    }
    uint32_t have_node = LAT_CMD_NONE; // node whose rows the loaders have fetched (into buffer cur)
    uint32_t cur = 0;
    bool overflow = false;

    for (;;) {
        // ---- candidateSet.top(): first unexpanded entry of R, ties -> largest id; tail joins at dist == max
        const int first = R.first_unexpanded(n, lane);
        int pick = -1, pick_tail = -1;
        uint32_t pick_id = 0;
        if (first >= 0) {
            const unsigned long long kfirst = R.get(first);
            const uint32_t db = key_dist_bits(kfirst);
            pick = first;
            pick_id = key_id(kfirst);
            if (first + 1 < n && key_dist_bits(R.get(first + 1)) == db) {
                pick = R.last_unexpanded_with(db, first, n, lane);
                pick_id = key_id(R.get(pick));
            }
            if (ntail > 0 && db == key_dist_bits(R.get(n - 1))) {
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
            __builtin_amdgcn_wave_barrier();
            if (lane == 0)
                tail[pick_tail] = tail[ntail - 1];
            ntail--;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t node = pick_id;

        // ---- the node's rows: already fetched if the prediction held, else one more round
        if (have_node != node) {
            cur ^= 1u;
            if (lane == 0) {
                sh->cmd = node;
                sh->buf = cur;
            }
            __syncthreads(); // B1
            __syncthreads(); // B2
        }
        const int cnt = (int)sh->cnt[cur];
        const uint32_t nb = lane < 32 ? sh->link[cur][lane] : 0u;
        const float dq = lane < 32 ? sh->dist[cur][lane] : 0.f;

        // ---- visited test-and-set (hnswalg.cpp:78-82): the bitmap is this wave's alone
        bool fresh = false;
        if (lane < cnt) {
            const uint32_t bit = 1u << (nb & 31);
            const uint32_t old = atomicOr(&bm[nb >> 5], bit); // returning LDS atomic: two links never alias a word unseen
            fresh = !(old & bit);
        }
        unsigned long long topk = R.get(n - 1);
        const float top0 = __uint_as_float(key_dist_bits(topk));
        unsigned long long cand = __ballot(fresh && (n < ef || top0 > dq));

        // ---- prediction of the node after this one: the nearer of R's next unexpanded entry and the best new candidate
        {
            const unsigned long long kc = ((cand >> lane) & 1ull) ? mk_key(dq, nb) : ~0ull;
            const unsigned long long best_new = wave_min_u64_l(kc);
            const int f2 = R.first_unexpanded(n, lane);
            const unsigned long long best_old = f2 >= 0 ? (R.get(f2) & ~1ull) : ~0ull;
            const unsigned long long best = best_new < best_old ? best_new : best_old;
            const uint32_t pred = best == ~0ull ? LAT_CMD_NONE : key_id(best);
            cur ^= 1u;
            if (lane == 0) {
                sh->cmd = pred;
                sh->buf = cur;
            }
            have_node = pred;
            __syncthreads(); // B1: the loaders start on the predicted node
        }

        // ---- admissions in link order (hnswalg.cpp:93-103)
        while (cand) {
            const int b = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const float dj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dq), b));
            const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)nb, b);
            const unsigned long long oldtop = topk;
            if (!(__uint_as_float(key_dist_bits(oldtop)) > dj || n < ef))
                continue;
            const unsigned long long K = mk_key(dj, idj);
            const bool full = n == ef;
            R.insert_sorted(K, lane);
            if (!full)
                n++;
            topk = R.get(n - 1);
            // candidates that left topResults but may still be popped (distance equal to the lower bound)
            const uint32_t newmax = key_dist_bits(topk);
            if (ntail > 0 && key_dist_bits(tail[0]) != newmax)
                ntail = 0;
            if (full && !(oldtop & 1ull) && key_dist_bits(oldtop) == newmax) {
                if (ntail < kTailCap) {
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0)
                        tail[ntail] = oldtop;
                    ntail++;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    if (lane == 0)
                        atomicOr(status, kStatusHnswTieOverflow);
                    overflow = true;
                    break;
                }
            }
        }
        __syncthreads(); // B2: the predicted node's rows are in buffer cur
        if (overflow)
            break;
    }
    if (lane == 0)
        sh->cmd = LAT_CMD_EXIT;
    __syncthreads(); // B1 of the loaders' last round

    // searchKnn pops down to nprobe (hnswalg.cpp:229-233); IndexIVF_HNSW.cpp:249-259 unloads nearest first
#pragma unroll
    for (int cc = 0; cc < NCH; cc++) {
        const int i = cc * 64 + lane;
        if (i < nprobe) {
            const bool have = !overflow && i < n;
            coarse_ids[(size_t)q * nprobe + i] = have ? key_id(R.r[cc]) : 0xffffffffu;
            coarse_dists[(size_t)q * nprobe + i] = have ? __uint_as_float(key_dist_bits(R.r[cc])) : 0.f;
        }
    }
}

// fat[i][r][:] = vectors[links[i][r]] for r < counts[i], zero beyond: one workgroup per node
__global__ __launch_bounds__(256) void build_fat_kernel(GraphTables g, float *__restrict__ fat)
{
    const size_t node = blockIdx.x;
    const int cnt = g.counts[node];
    const int d4 = g.d >> 2;
    float4 *dst = reinterpret_cast<float4 *>(fat + node * 32 * (size_t)g.d);
    for (int e = threadIdx.x; e < 32 * d4; e += 256) {
        const int r = e / d4, c = e - r * d4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < cnt)
            v = reinterpret_cast<const float4 *>(g.vectors + (size_t)g.links[node * g.maxM + r] * g.d)[c];
        dst[e] = v;
    }
}

} // namespace

bool coarse_latency_supported(const GraphTables &g, int ef)
{
    return g.fat && (g.d == 128 || g.d == 96) && g.maxM <= 32 && g.n <= (1u << 20) && ef <= 256;
}

hipError_t launch_build_fat(hipStream_t s, const GraphTables &g, float *fat)
{
    if (g.maxM > 32 || (g.d & 3))
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(build_fat_kernel, dim3(g.n), dim3(256), 0, s, g, fat);
    return hipGetLastError();
}

hipError_t launch_coarse_latency(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                                 uint32_t *coarse_ids, float *coarse_dists, uint32_t *status)
{
    if (nq == 0)
        return hipSuccess;
    if (!coarse_latency_supported(g, ef) || nprobe > ef)
        return hipErrorInvalidValue;
    const size_t shm = ((sizeof(LatShared) + 15) & ~(size_t)15) + (size_t)((g.n + 127u) / 128u) * 16;
    const int nch = (ef + 63) / 64;
#define IVFHNSW_LAT(N, J)                                                                                             \
    do {                                                                                                              \
        auto *kern = hnsw_walk_lat_kernel<N, J>;                                                                      \
        static size_t attr = 0;                                                                                       \
        if (shm > attr) {                                                                                             \
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
            if (e != hipSuccess)                                                                                      \
                return e;                                                                                             \
            attr = shm;                                                                                               \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3(nq), dim3(LAT_THREADS), shm, s, g, xq, nq, nprobe, ef, coarse_ids, coarse_dists, \
                           status);                                                                                   \
    } while (0)
#define IVFHNSW_LAT_J(N)       \
    do {                       \
        if (g.d == 128)        \
            IVFHNSW_LAT(N, 16); \
        else                   \
            IVFHNSW_LAT(N, 12); \
    } while (0)
    if (nch <= 1)
        IVFHNSW_LAT_J(1);
    else if (nch == 2)
        IVFHNSW_LAT_J(2);
    else if (nch == 3)
        IVFHNSW_LAT_J(3);
    else
        IVFHNSW_LAT_J(4);
#undef IVFHNSW_LAT_J
#undef IVFHNSW_LAT
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
