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
#include <stdio.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

namespace {

constexpr int LAT_THREADS = 384; // wave 0, loader waves 1, 2, 3 and 5; wave 4 would share wave 0's SIMD: it leaves
constexpr uint32_t LAT_CMD_NONE = 0xffffffffu, LAT_CMD_EXIT = 0xfffffffeu;
// a fat record: 32 rows of d floats, then a 256-byte trailer -- the node's link list (32 words), its link count, padding --
// so that ONE touched region carries everything an expansion reads
constexpr int LAT_TRAILER = 64; // floats

struct LatShared {
    float dist[2][32];     // distances of the node being fetched, double buffered
    uint32_t link[2][32];  // its link list
    uint32_t cnt[2];       // its link count
    uint32_t cmd;          // node the loaders fetch next (LAT_CMD_NONE: nothing, LAT_CMD_EXIT: done)
    uint32_t buf;          // buffer they fill
    uint32_t touch[2];     // the two candidates after it: their rows are only touched (pulled into L2 and the TLBs)
    unsigned long long tail[kTailCap];
    unsigned long long mb[256 + 32]; // merge buffer: the full set and the candidates of one expansion
};

// NJ = d / 8 (16: d = 128, 12: d = 96); NCH = registers of the result set (ef <= 64 * NCH)
__device__ __forceinline__ unsigned long long lat_stamp()
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define LAT_STAMP(i)                          \
    if (STAMPS) {                             \
        const unsigned long long t_ = lat_stamp(); \
        st_acc[i] += t_ - st_t;               \
        st_t = t_;                            \
    }

// STAMPS: diagnostic instantiation (IVFHNSW_LAT_STAMPS=1), never timed: per-phase s_memtime sums printed per query
template <int NCH, int NJ, bool STAMPS>
__global__ __launch_bounds__(LAT_THREADS) void hnsw_walk_lat_kernel(GraphTables g, const float *__restrict__ xq, int nq,
                                                                    int nprobe, int ef, uint32_t *__restrict__ coarse_ids,
                                                                    float *__restrict__ coarse_dists,
                                                                    uint32_t *__restrict__ status, int diag,
                                                                    unsigned long long *__restrict__ zero_keys,
                                                                    uint32_t *__restrict__ zero_done,
                                                                    uint32_t *__restrict__ redo_hdr,
                                                                    uint32_t *__restrict__ redo_list)
{
    // diag (STAMPS builds only, IVFHNSW_LAT_DIAG=1): the loaders skip their fetch -- results are garbage, what is read
    // off the stamps is the cost of the two barriers alone
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
        sh->touch[0] = sh->touch[1] = LAT_CMD_NONE;
        if (zero_keys) { // the tail kernel's per-query meeting words (kernels_tail.hip): one launch less per call
            zero_keys[q] = 0ull;
            zero_done[q] = 0u;
        }
    }
    __syncthreads();

    if (wave == 4)
        return; // waves go to SIMD (wave % 4): nothing shares the SIMD of the serial wave (stamps: its partner delayed
                // both -- 1300 cycles of wave 0 waiting at B2 per expansion)
    if (wave != 0) {
        // ---- loaders: B1 -> read the command -> fetch -> B2
        // A fetch from HBM through cold TLBs costs ~5000 cycles here (stamps: wave 0 waited 2300 cycles at B2 AFTER
        // 2600 cycles of insertions), an L2 hit a few hundred.  So besides the predicted node's rows every round
        // TOUCHES the rows of the two candidates behind it -- one dword per 128-byte line, 256 lines, one per loader
        // lane -- which are the nodes most likely to be expanded a round or two later.
        const int li = wave == 5 ? 3 : wave - 1;    // loader index 0..3
        const int r = 8 * li + (lane >> 3);         // the row this group of eight lanes evaluates
        const int tix = li * 64 + lane;             // 0..255: touched node tix >> 7, line tix & 127 of its 16 KB
        float sink = 0.f;
        unsigned long long ld_acc[4] = {0, 0, 0, 0}, ld_t = 0;
        int ld_n = 0;
        for (;;) {
            __syncthreads(); // B1
            if (STAMPS)
                ld_t = lat_stamp();
            const uint32_t node = sh->cmd;
            const uint32_t b = sh->buf;
            const uint32_t tnode = sh->touch[tix >> 7];
            if (node == LAT_CMD_EXIT)
                break;
            float tv_a = 0.f, tv_b = 0.f;
            if (node != LAT_CMD_NONE && !(STAMPS && diag)) {
                const float *rec = g.fat + (size_t)node * (32 * D + LAT_TRAILER);
                // a fat row is stored TRANSPOSED for this read: lane t's NJ components (x[8j + t], j = 0..NJ-1: accumulator t
                // of the reference's eight) are contiguous, NJ / 4 sixteen-byte loads instead of NJ dword loads
                const float4 *r4 = reinterpret_cast<const float4 *>(rec + r * D + (lane & 7) * NJ);
                float y[NJ];
#pragma unroll
                for (int i = 0; i < NJ / 4; i++) {
                    const float4 v = r4[i];
                    y[4 * i] = v.x, y[4 * i + 1] = v.y, y[4 * i + 2] = v.z, y[4 * i + 3] = v.w;
                }
                // Everything below is issued by EVERY lane without a branch in between: with loads under control flow
                // hipcc counts only the sixteen row loads and waits vmcnt(15..0) for them -- and vmcnt(0) before the
                // last row means waiting for the touches behind it, a whole HBM round trip on the critical path (stamps:
                // 1300 cycles of every wave standing at B2).  Unconditional, they are the youngest three loads and the
                // rows are waited for with vmcnt(18..3).
                uint32_t lk = reinterpret_cast<const uint32_t *>(rec + 32 * D)[lane < 32 ? lane : 32]; // links | count
                uint32_t ct = lk;
                // touches: line tix & 127 of one of the two nodes behind the predicted one, and (cheap: the same line
                // for most lanes) the trailer's two lines; an absent node re-touches the record being read anyway
                const float *trec = tnode < g.n ? g.fat + (size_t)tnode * (32 * D + LAT_TRAILER) : rec;
                const float tv0 = trec[(size_t)(tix & 127) * (D * 32 / 128)];
                const float tv1 = trec[(tix & 127) < 2 ? 32 * D + 32 * (tix & 127) : (size_t)(tix & 127) * (D * 32 / 128)];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (STAMPS) {
                    unsigned long long t_ = lat_stamp();
                    ld_acc[0] += t_ - ld_t; // command read, addresses, loads issued
                    ld_t = t_;
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // rows and trailer (the two touches are the youngest loads)
                    t_ = lat_stamp();
                    ld_acc[1] += t_ - ld_t; // rows arrived
                    ld_t = t_;
                    ld_n++;
                }
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NJ; i++) {
                    const float d0 = __fsub_rn(q_reg[i], y[i]);
                    acc = __fadd_rn(acc, __fmul_rn(d0, d0));
                }
                const float mir = __uint_as_float(
                    (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(acc), 0x141, 0xf, 0xf, true));
                float dq = __fadd_rn(quad_bcast<0>(acc), quad_bcast<1>(acc));
                dq = __fadd_rn(dq, quad_bcast<2>(acc));
                dq = __fadd_rn(dq, quad_bcast<3>(acc));
                dq = __fadd_rn(dq, quad_bcast<3>(mir));
                dq = __fadd_rn(dq, quad_bcast<2>(mir));
                dq = __fadd_rn(dq, quad_bcast<1>(mir));
                dq = __fadd_rn(dq, quad_bcast<0>(mir));
                if ((lane & 7) == 0)
                    sh->dist[b][r] = dq;
                if (wave == 1) {
                    if (lane < 32)
                        sh->link[b][lane] = lk;
                    if (lane == 32)
                        sh->cnt[b] = ct;
                }
                tv_a = tv0;
                tv_b = tv1;
            }
            if (STAMPS) {
                const unsigned long long t_ = lat_stamp();
                ld_acc[2] += t_ - ld_t; // arithmetic, reduction, LDS writes
                ld_t = t_;
            }
            // B2 by hand: __syncthreads() drains vmcnt before its s_barrier, i.e. it would make every loader wait for its
            // TOUCH load (a whole HBM round trip nobody needs yet: 1300 cycles of wave 0 standing at B2 per expansion,
            // stamps).  What wave 0 reads after B2 are the LDS writes above: lgkmcnt(0) is all the barrier has to cover.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(sink) : : "memory");
            sink += tv_a + tv_b; // the touches are waited for here (the asm's output pins the add behind the barrier)
            if (STAMPS) {
                const unsigned long long t_ = lat_stamp();
                ld_acc[3] += t_ - ld_t; // B2 and the touch
            }
        }
        if (STAMPS && lane == 0 && ld_n)
            printf("[lat stamps] loader wave %d, %d fetches; cycles/fetch: issue %llu, rows arrive %llu, arithmetic %llu, B2 + touch %llu\n", wave,
                   ld_n, ld_acc[0] / ld_n, ld_acc[1] / ld_n, ld_acc[2] / ld_n, ld_acc[3] / ld_n);
        if (sink == 1.2345678e30f) // never: keeps the touches alive
            status[1] = 0u;
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
    bool preselected = false; // the end of the previous expansion already popped the next node (fast path below)
    uint32_t node = 0;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = STAMPS ? lat_stamp() : 0ull;
    int st_exp = 0, st_miss = 0, st_fast = 0, st_adm = 0;

    for (;;) {
      if (STAMPS)
          st_fast += preselected ? 1 : 0;
      if (!preselected) {
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
        node = pick_id;
      }
        preselected = false;
        LAT_STAMP(0) // selection
        if (STAMPS) {
            st_exp++;
            st_miss += have_node != node ? 1 : 0;
        }

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
        LAT_STAMP(1) // waiting for a node that was not predicted
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

        LAT_STAMP(2) // staged rows, visited test, candidate mask
        // ---- prediction of the node after this one: the nearer of R's next unexpanded entry and the best new candidate
        unsigned long long pred_key = ~0ull;
        {
            // (a handful of candidates: their minimum on the scalar unit, two v_readlane each -- a wave-wide 64-bit
            // minimum is six dependent cross-lane round trips, which a lone wavefront waits out one by one)
            unsigned long long best_new = ~0ull;
            for (unsigned long long cm = cand; cm; cm &= cm - 1) {
                const int b = __ffsll((long long)cm) - 1;
                const uint32_t dbj = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(dq), b);
                const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)nb, b);
                const unsigned long long kj = ((unsigned long long)dbj << 32) | ((unsigned long long)idj << 1);
                best_new = kj < best_new ? kj : best_new;
            }
            // R's unexpanded entries: the first is the old candidate of the prediction, the next two are what the
            // loaders touch (from the same ballots: scalar bit picking and two v_readlane each)
            int f2 = -1;
            uint32_t t0 = LAT_CMD_NONE, t1 = LAT_CMD_NONE;
            {
                int seen = 0;
#pragma unroll
                for (int cc = 0; cc < NCH; cc++) {
                    unsigned long long m = __ballot(!(R.r[cc] & 1ull)) & lanes_below(n, cc);
                    if (m && seen < 3) {
                        const int l0 = __ffsll((long long)m) - 1;
                        if (seen == 0)
                            f2 = cc * 64 + l0;
                        else if (seen == 1)
                            t0 = key_id(readlane_u64(R.r[cc], l0));
                        else
                            t1 = key_id(readlane_u64(R.r[cc], l0));
                        seen++;
                        m &= m - 1;
                        if (m && seen < 3) {
                            const int l1 = __ffsll((long long)m) - 1;
                            if (seen == 1)
                                t0 = key_id(readlane_u64(R.r[cc], l1));
                            else
                                t1 = key_id(readlane_u64(R.r[cc], l1));
                            seen++;
                            m &= m - 1;
                            if (m && seen < 3) {
                                t1 = key_id(readlane_u64(R.r[cc], __ffsll((long long)m) - 1));
                                seen++;
                            }
                        }
                    }
                }
            }
            const unsigned long long best_old = f2 >= 0 ? (R.get(f2) & ~1ull) : ~0ull;
            const unsigned long long best = best_new < best_old ? best_new : best_old;
            pred_key = best;
            const uint32_t pred = best == ~0ull ? LAT_CMD_NONE : key_id(best);
            cur ^= 1u;
            if (lane == 0) {
                sh->cmd = pred;
                sh->buf = cur;
                sh->touch[0] = t0;
                sh->touch[1] = t1;
            }
            have_node = pred;
            __syncthreads(); // B1: the loaders start on the predicted node
        }

        LAT_STAMP(3) // prediction + B1
        // ---- all admissions of the expansion in one step (the merge of kernels_hnsw.hip, here over up to 32 candidates).
        // With the set full, inserting the candidates one by one in link order (below) leaves the ef smallest keys of
        // set + candidates PROVIDED the ef-th and (ef+1)-th of the merged order differ in distance: every loser then lies
        // above the final maximum, no strict-'<' tie decided anything, nothing can wait in the tail.  Every candidate is
        // ranked among the set and the candidates, every set entry among the candidates, the merged order is scattered
        // through LDS and its boundary inspected; a tie there leaves the registers untouched for the sequential path.
        if (NCH <= 4 && n == ef && ntail == 0 && cand) {
            const bool is_cand = (cand >> lane) & 1ull;
            const unsigned long long Kc = mk_key(dq, nb);
            int dn[NCH];
#pragma unroll
            for (int cc = 0; cc < NCH; cc++)
                dn[cc] = 0;
            const int ncand = __popcll(cand);
            int cand_below = 0, set_below = 0;
            for (unsigned long long cm = cand; cm; cm &= cm - 1) {
                const int b = __ffsll((long long)cm) - 1;
                const unsigned long long Kj = readlane_u64(Kc, b);
                int below = 0;
#pragma unroll
                for (int cc = 0; cc < NCH; cc++) {
                    const bool lt = (R.r[cc] & ~1ull) < Kj;
                    below += __popcll(__ballot(lt) & lanes_below(ef, cc));
                    dn[cc] += lt ? 1 : 0;
                }
                cand_below += Kj < Kc ? 1 : 0;
                set_below = lane == b ? below : set_below;
            }
            unsigned long long *mb = sh->mb;
#pragma unroll
            for (int cc = 0; cc < NCH; cc++)
                if (cc * 64 + lane < ef)
                    mb[cc * 64 + lane + (ncand - dn[cc])] = R.r[cc];
            if (is_cand)
                mb[set_below + cand_below] = Kc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long b_lo = mb[ef - 1], b_hi = mb[ef];
            unsigned long long merged[NCH];
#pragma unroll
            for (int cc = 0; cc < NCH; cc++)
                merged[cc] = cc * 64 + lane < ef ? mb[cc * 64 + lane] : R.r[cc];
            if (key_dist_bits(b_lo) != key_dist_bits(b_hi)) {
#pragma unroll
                for (int cc = 0; cc < NCH; cc++)
                    R.r[cc] = merged[cc];
                if (STAMPS)
                    st_adm += ncand;
                cand = 0;
            }
            __builtin_amdgcn_wave_barrier();
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
            if (STAMPS)
                st_adm++;
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
                    // more than kTailCap exact ties at the efSearch boundary: this form cannot finish the query.  Async
                    // callers get it walked again by the redo form (launch_coarse_redo, kernels_hnsw.hip: the list);
                    // the synchronous host-pointer call asks for the status bit instead and repeats itself on the
                    // throughput walk (capi.cpp: no redo launch on the one-query-per-call path)
                    if (lane == 0) {
                        if (redo_hdr)
                            redo_list[atomicAdd(&redo_hdr[0], 1u)] = (uint32_t)q;
                        else
                            atomicOr(status, kStatusHnswTieOverflow);
                    }
                    overflow = true;
                    break;
                }
            }
        }
        LAT_STAMP(4) // admissions
        __syncthreads(); // B2: the predicted node's rows are in buffer cur
        LAT_STAMP(5) // waiting for the loaders at B2
        if (overflow)
            break;
        // ---- fast pop: the predicted key is the smallest of all unexpanded entries and all candidates of this
        // expansion, so after the insertions it IS the first unexpanded entry -- unless it was not admitted or evicted
        // (then it is absent) or another unexpanded entry shares its distance (the reference pops the LARGEST id of
        // equal distances: the full selection decides).  One compare per register instead of the search above.
        if (ntail == 0 && pred_key != ~0ull) {
            const uint32_t pd = key_dist_bits(pred_key);
            int found = 0, same = 0;
#pragma unroll
            for (int cc = 0; cc < NCH; cc++) {
                const bool live = !(R.r[cc] & 1ull);
                const unsigned long long in = lanes_below(n, cc);
                found += __popcll(__ballot(live && (R.r[cc] & ~1ull) == pred_key) & in);
                same += __popcll(__ballot(live && key_dist_bits(R.r[cc]) == pd) & in);
            }
            if (found == 1 && same == 1) {
#pragma unroll
                for (int cc = 0; cc < NCH; cc++)
                    if ((R.r[cc] & ~1ull) == pred_key && cc * 64 + lane < n)
                        R.r[cc] |= 1ull;
                node = key_id(pred_key);
                preselected = true;
            }
        }
        LAT_STAMP(6) // fast pop
    }
    if (STAMPS && lane == 0)
        printf("[lat stamps] query %d: %d expansions (%d not predicted, %d fast pops, %d admissions); cycles/expansion: "
               "select %llu, unpredicted wait %llu, stage+visited %llu, predict+B1 %llu, admissions %llu, B2 wait %llu, "
               "fast pop %llu\n", q, st_exp, st_miss, st_fast, st_adm, st_acc[0] / st_exp, st_acc[1] / st_exp,
               st_acc[2] / st_exp, st_acc[3] / st_exp, st_acc[4] / st_exp, st_acc[5] / st_exp, st_acc[6] / st_exp);
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

// fat[i]: rows r < counts[i] = vectors[links[i][r]], zero beyond; then the trailer.  One workgroup per node.
__global__ __launch_bounds__(256) void build_fat_kernel(GraphTables g, float *__restrict__ fat)
{
    const size_t node = blockIdx.x;
    const int cnt = g.counts[node];
    const int d4 = g.d >> 2;
    float *rec = fat + node * (32 * (size_t)g.d + LAT_TRAILER);
    if (threadIdx.x < LAT_TRAILER) {
        uint32_t w = 0;
        if ((int)threadIdx.x < cnt)
            w = g.links[node * g.maxM + threadIdx.x];
        else if (threadIdx.x == 32)
            w = (uint32_t)cnt;
        reinterpret_cast<uint32_t *>(rec + 32 * (size_t)g.d)[threadIdx.x] = w;
    }
    // rows transposed for the loaders: element t * (d / 8) + j of a row is component 8j + t of the neighbour
    const int nj = g.d >> 3;
    (void)d4;
    for (int e = threadIdx.x; e < 32 * g.d; e += 256) {
        const int r = e / g.d, o = e - r * g.d;
        const int t = o / nj, j = o - t * nj;
        float v = 0.f;
        if (r < cnt)
            v = g.vectors[(size_t)g.links[node * g.maxM + r] * g.d + 8 * j + t];
        rec[e] = v;
    }
}

} // namespace

bool coarse_latency_supported(const GraphTables &g, int ef)
{
    return g.fat && (g.d == 128 || g.d == 96) && g.maxM <= 32 && g.n <= (1u << 20) && ef <= 256;
}

size_t coarse_latency_fat_bytes(const GraphTables &g) { return (size_t)g.n * (32 * (size_t)g.d + LAT_TRAILER) * sizeof(float); }

hipError_t launch_build_fat(hipStream_t s, const GraphTables &g, float *fat)
{
    if (g.maxM > 32 || (g.d & 3))
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(build_fat_kernel, dim3(g.n), dim3(256), 0, s, g, fat);
    return hipGetLastError();
}

hipError_t launch_coarse_latency(hipStream_t s, const GraphTables &g, const float *xq, int nq, int nprobe, int ef,
                                 uint32_t *coarse_ids, float *coarse_dists, uint32_t *status, uint64_t *zero_keys_u64,
                                 uint32_t *zero_done, uint32_t *redo_hdr, uint32_t *redo_list)
{
    unsigned long long *zero_keys = reinterpret_cast<unsigned long long *>(zero_keys_u64);
    if (nq == 0)
        return hipSuccess;
    if (!coarse_latency_supported(g, ef) || nprobe > ef)
        return hipErrorInvalidValue;
    const size_t shm = ((sizeof(LatShared) + 15) & ~(size_t)15) + (size_t)((g.n + 127u) / 128u) * 16;
    const int nch = (ef + 63) / 64;
    static const bool stamps = [] {
        const char *e = getenv("IVFHNSW_LAT_STAMPS");
        return e && atoi(e) == 1;
    }();
    static const int diag = [] {
        const char *e = getenv("IVFHNSW_LAT_DIAG");
        return (e && atoi(e) == 1) ? 1 : 0;
    }();
#define IVFHNSW_LAT(N, J)                                                                                             \
    do {                                                                                                              \
        auto *kern = stamps ? hnsw_walk_lat_kernel<N, J, true> : hnsw_walk_lat_kernel<N, J, false>;                   \
        static DynLdsState attr[2];                                                                                   \
        if (hipError_t e = raise_dyn_lds((const void *)kern, shm, attr[stamps ? 1 : 0]); e != hipSuccess)             \
            return e;                                                                                                 \
        hipLaunchKernelGGL(kern, dim3(nq), dim3(LAT_THREADS), shm, s, g, xq, nq, nprobe, ef, coarse_ids, coarse_dists, \
                           status, diag, zero_keys, zero_done, redo_hdr, redo_list);                                                                                   \
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
