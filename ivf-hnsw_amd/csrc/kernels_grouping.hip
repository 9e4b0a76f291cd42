// Scan plan of IndexIVF_HNSW_Grouping::search (IndexIVF_HNSW_Grouping.cpp:222-353): sub-centroid
// distances, pruning threshold (pass 1) and the list of sub-groups to score with their constant
// term1 + term2 (pass 2).  One wavefront per query; lanes run over the nsubc sub-groups of a probe.
//
// The reference evaluates ||x - y_N||^2 lazily through a per-query cache (Grouping.cpp:244-250,
// 311-316); the cached value is a pure function of (query, centroid) -- the HNSW walk and fvec_L2sqr are
// the same arithmetic (utils.cpp:22-52 == hnswalg.cpp:326-357) -- so it is simply evaluated where needed.
// Sequential float sums (the threshold, Grouping.cpp:253) keep the reference's (probe, sub-group) order.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <stdio.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

// scratch per query: qsd[max_seg] (pass-1 values, row-major [row][subc]) | qn[max_seg] (distances)
__global__ __launch_bounds__(64) void plan_grouping_kernel(IvfTables t, GroupTables g, GraphTables gr,
                                                           const float *__restrict__ xq,
                                                           const uint32_t *__restrict__ cid,
                                                           const float *__restrict__ cd, int nq, int nprobe,
                                                           unsigned long long max_codes, int do_pruning,
                                                           Seg *__restrict__ segs, uint32_t *__restrict__ lpos,
                                                           PlanHdr *__restrict__ hdr, int max_seg,
                                                           unsigned long long *__restrict__ keys, int k,
                                                           float *__restrict__ scratch)
{
    extern __shared__ __attribute__((aligned(16))) float s_q[]; // query[d] | dist[64]
    const int lane = threadIdx.x;
    const int q = blockIdx.x;
    float *s_dist = s_q + t.d;
    const int nsubc = g.nsubc;
    for (int j = lane; j < k; j += 64)
        keys[(size_t)q * k + j] = kKeyInit;
    for (int i = lane; i < t.d; i += 64)
        s_q[i] = xq[(size_t)q * t.d + i];
    __syncthreads();

    float *qsd = scratch + (size_t)q * 2 * max_seg;
    float *qnv = qsd + max_seg;
    const uint32_t *qc = cid + (size_t)q * nprobe;
    const float *qd = cd + (size_t)q * nprobe;

    // ---- pass 1 (Grouping.cpp:222-262)
    float threshold = 0.0f;
    int p1_rows = 0;
    if (do_pruning) {
        unsigned long long ncode = 0;
        unsigned long long nsubgroups = 0;
        int row = 0;
        for (int i = 0; i < nprobe; i++) {
            const uint32_t c = qc[i];
            if (c >= t.nc)
                continue;
            const unsigned long long gs = t.goff[c + 1] - t.goff[c];
            if (gs == 0)
                continue;
            const float alpha = g.alphas[c];
            const float oma = __fsub_rn(1.0f, alpha);
            const float term1 = __fmul_rn(oma, qd[i]);
            for (int s0 = 0; s0 < nsubc; s0 += 64) {
                const int subc = s0 + lane;
                bool active = false;
                float v = 0.f, qn = 0.f;
                uint32_t nn = 0;
                if (subc < nsubc && g.sub_sizes[(size_t)c * nsubc + subc] != 0) {
                    active = true;
                    nn = g.nn_idx[(size_t)c * nsubc + subc];
                }
                // distances of the active sub-groups' neighbour centroids: a quad of lanes per row, 16 rows per
                // pass, results handed back to the owning lane through LDS
                {
                    const unsigned long long am = __ballot(active);
                    const int na = __popcll(am);
                    for (int base = 0; base < na; base += 16) {
                        const int r = base + (lane >> 2);
                        const int src = r < na ? nth_set_bit(am, r) : 0;
                        const uint32_t nnq = (uint32_t)__shfl((int)nn, src, 64);
                        float dq = 0.f;
                        if (r < na)
                            dq = l2_ref_order_quad(gr.vectors + (size_t)nnq * t.d, s_q, t.d, lane & 3);
                        if (r < na && (lane & 3) == 0)
                            s_dist[src] = dq;
                    }
                    __syncthreads();
                    if (active)
                        qn = s_dist[lane];
                    __syncthreads();
                }
                if (active) {
                    const float a = __fmul_rn(oma, g.inter_dists[(size_t)c * nsubc + subc]);
                    const float b = __fsub_rn(a, qn);
                    v = __fsub_rn(term1, __fmul_rn(alpha, b)); // Grouping.cpp:251-252
                }
                if (subc < nsubc) {
                    qsd[(size_t)row * nsubc + subc] = v; // value-initialised 0.0 where inactive (:228)
                    qnv[(size_t)row * nsubc + subc] = qn;
                }
                unsigned long long m = __ballot(active);
                nsubgroups += __popcll(m);
                while (m) { // threshold += qsd[subc], in sub-group order (:253)
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    threshold = __fadd_rn(threshold, __shfl(v, j, 64));
                }
            }
            ncode += gs;
            row++;
            if (ncode >= 2 * max_codes)
                break;
        }
        p1_rows = row;
        threshold = __fdiv_rn(threshold, (float)nsubgroups); // :261, 0/0 = NaN when nothing was seen
    }
    __syncthreads();

    // ---- pass 2 (Grouping.cpp:283-353)
    unsigned long long ncode = 0; // codes scored so far == scan position of the next one
    uint32_t ns = 0, nl = 0;
    int row = 0;
    Seg *sq = segs + (size_t)q * max_seg;
    uint32_t *lq = lpos + (size_t)q * max_seg;
    for (int i = 0; i < nprobe; i++) {
        const uint32_t c = qc[i];
        if (c >= t.nc)
            continue;
        const unsigned long long gs = t.goff[c + 1] - t.goff[c];
        if (gs == 0)
            continue;
        const float alpha = g.alphas[c];
        const float oma = __fsub_rn(1.0f, alpha);
        const float term1 = __fmul_rn(oma, __fsub_rn(qd[i], t.centroid_norms[c]));
        const uint32_t lo_c = t.loff[c];
        const bool owned = lo_c != kNotOwned;
        uint32_t list_off = 0; // codes of this list before the current chunk of sub-groups
        for (int s0 = 0; s0 < nsubc; s0 += 64) {
            const int subc = s0 + lane;
            uint32_t sz = 0;
            bool scanned = false;
            float cterm = 0.f;
            if (subc < nsubc)
                sz = g.sub_sizes[(size_t)c * nsubc + subc];
            if (sz != 0) {
                float qs = 0.0f;
                if (do_pruning && row < p1_rows)
                    qs = qsd[(size_t)row * nsubc + subc];
                scanned = !do_pruning || qs < threshold; // :308
            }
            // term2 needs ||x - y_N||^2: stored by pass 1 for its rows, evaluated now (quads) for the others
            float qn2 = 0.f;
            const bool need = scanned && !(do_pruning && row < p1_rows);
            if (scanned && !need)
                qn2 = qnv[(size_t)row * nsubc + subc];
            {
                const uint32_t nn = scanned ? g.nn_idx[(size_t)c * nsubc + subc] : 0u;
                const unsigned long long am = __ballot(need);
                const int na = __popcll(am);
                if (na) {
                    for (int base = 0; base < na; base += 16) {
                        const int r = base + (lane >> 2);
                        const int src = r < na ? nth_set_bit(am, r) : 0;
                        const uint32_t nnq = (uint32_t)__shfl((int)nn, src, 64);
                        float dq = 0.f;
                        if (r < na)
                            dq = l2_ref_order_quad(gr.vectors + (size_t)nnq * t.d, s_q, t.d, lane & 3);
                        if (r < na && (lane & 3) == 0)
                            s_dist[src] = dq;
                    }
                    __syncthreads();
                    if (need)
                        qn2 = s_dist[lane];
                    __syncthreads();
                }
                if (scanned) {
                    const float term2 = __fmul_rn(alpha, __fsub_rn(qn2, t.centroid_norms[nn])); // :318
                    cterm = __fadd_rn(term1, term2);
                }
            }
            const uint32_t in_sz = wave_incl_scan(sz, lane);
            const uint32_t in_sc = wave_incl_scan(scanned ? sz : 0u, lane);
            const unsigned long long m = __ballot(scanned);
            const uint32_t rank_sc = __popcll(m & ((1ull << lane) - 1ull));
            if (scanned && owned) {
                Seg sg;
                sg.start = lo_c + list_off + (in_sz - sz);
                sg.len = sz;
                sg.vpos = (uint32_t)ncode + (in_sc - sz);
                sg.cterm = cterm;
                sq[ns + rank_sc] = sg;
                lq[ns + rank_sc] = nl + (in_sc - sz);
            }
            const uint32_t tot_sz = __shfl(in_sz, 63, 64), tot_sc = __shfl(in_sc, 63, 64);
            list_off += tot_sz;
            ncode += tot_sc;
            if (owned) {
                ns += (uint32_t)__popcll(m);
                nl += tot_sc;
            }
        }
        if (ncode >= max_codes)
            break;
        if (do_pruning)
            row++;
    }
    if (lane == 0) {
        PlanHdr h;
        h.nseg = ns;
        h.total = nl;
        hdr[q] = h;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same plan by FOUR wavefronts per query.  plan_grouping_kernel is one wavefront walking a long dependent chain:
// ~26 probed groups x (4 gather passes of 16 rows, each a memory round trip, + the sequential threshold sum): ~0.5 ms
// per query, 10 000 queries on the ~4096-8192 wavefronts that fit -- two "rounds" of one query latency, the second half
// empty, at the SAME 1.3 ms whether the centroid table sits in the memory-side cache (100M shape) or in HBM (1B): it is
// bound by that chain, not by bytes.  Which (probe, sub-group) distances a query needs is known from list sizes alone
// (pass 1 runs until 2 x max_codes codes have been seen, Grouping.cpp:256-258; without pruning pass 2 runs until
// max_codes), so the four wavefronts gather them for different probes at once into the per-query scratch (phase A); the
// threshold is then summed by wavefront 0 in the reference's (probe, sub-group) order from that scratch (phase B), and
// pass 2 -- bookkeeping over the probes in order -- runs on wavefront 0 as before, reading the distances (phase C).
// Same arithmetic on the same operands in the same order wherever order matters: same bits.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_order()
{
    // LDS operations of one wavefront execute in order; this only stops hipcc from moving them across each other
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ unsigned long long incl_scan_u64(unsigned long long v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(v, off, 64);
        if (lane >= off)
            v += o;
    }
    return v;
}

// DEDUPE (round 3): phase A evaluates every DISTINCT neighbour centroid of the query once.  The reference caches
// ||x - y_N||^2 per query (query_centroid_dists, Grouping.cpp:244-250, 311-316) because the probed groups' neighbour lists
// overlap: on clustered centroids -- a query's probes are each other's neighbours -- the ~1350 (row, sub-group) pairs of a
// query name only a few hundred distinct centroids, and the 512-byte rows gathered for them are what phase A is made of.
// The four wavefronts enter the pairs' centroid ids into an LDS hash set (compare-and-swap, linear probing), compact the
// occupied slots, evaluate those rows by lane quads, and every pair reads its value back: the same function of (query,
// centroid), the same bits.  Chosen per index at upload from the overlap of its neighbour lists (GroupTables::dedupe).
constexpr int GH_SLOTS = 2048;
constexpr int GH_PROBES = 48;
constexpr uint32_t GH_EMPTY = 0xffffffffu;

__device__ __forceinline__ uint32_t gh_hash(uint32_t id) { return (id * 2654435761u) >> 21; } // 11 bits

template <bool DEDUPE>
__global__ __launch_bounds__(256) void plan_grouping4_kernel(IvfTables t, GroupTables g, GraphTables gr,
                                                            const float *__restrict__ xq,
                                                            const uint32_t *__restrict__ cid,
                                                            const float *__restrict__ cd, int nq, int nprobe,
                                                            unsigned long long max_codes, int do_pruning,
                                                            Seg *__restrict__ segs, uint32_t *__restrict__ lpos,
                                                            PlanHdr *__restrict__ hdr, int max_seg,
                                                            unsigned long long *__restrict__ keys, int k,
                                                            float *__restrict__ scratch,
                                                            unsigned long long *__restrict__ stamps)
{
    // stamps (diagnostic, IVFHNSW_PLAN_STAMPS=1, never timed): s_memtime sums of wavefront 0 per phase
    unsigned long long st_t = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    auto stamp = [&](int i) {
        if (stamps) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0)
                atomicAdd(&stamps[i], now - st_t);
            st_t = now;
        }
    };
    extern __shared__ __attribute__((aligned(16))) float s_q[]; // query[d] | dist[4][64] | probe of row[nprobe]
    __shared__ int s_p1, s_ra, s_nrows;
    __shared__ uint32_t h_key[DEDUPE ? GH_SLOTS : 1];
    __shared__ float h_val[DEDUPE ? GH_SLOTS : 1];
    __shared__ uint16_t h_list[DEDUPE ? GH_SLOTS : 2];
    __shared__ int h_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    float *s_dist = s_q + t.d + wave * 64;
    int *s_rowp = reinterpret_cast<int *>(s_q + t.d + 256);
    // per row (a non-empty probed group, in probe order), staged once so that wavefront 0's serial passes read LDS
    // instead of walking a chain of dependent global loads per row: centroid, list size, alpha, ||c||^2, first local
    // code, coarse distance, active sub-groups
    uint32_t *s_rc = reinterpret_cast<uint32_t *>(s_rowp + nprobe);
    uint32_t *s_rgs = s_rc + nprobe;
    float *s_ral = reinterpret_cast<float *>(s_rgs + nprobe);
    float *s_rcn = s_ral + nprobe;
    uint32_t *s_rlo = reinterpret_cast<uint32_t *>(s_rcn + nprobe);
    float *s_rqd = reinterpret_cast<float *>(s_rlo + nprobe);
    uint32_t *s_rna = reinterpret_cast<uint32_t *>(s_rqd + nprobe);
    const int nsubc = g.nsubc;
    for (int j = tid; j < k; j += 256)
        keys[(size_t)q * k + j] = kKeyInit;
    for (int i = tid; i < t.d; i += 256)
        s_q[i] = xq[(size_t)q * t.d + i];
    if constexpr (DEDUPE) {
        for (int i = tid; i < GH_SLOTS; i += 256)
            h_key[i] = GH_EMPTY;
        if (tid == 0)
            h_n = 0;
    }

    float *qsd = scratch + (size_t)q * 2 * max_seg;
    float *qnv = qsd + max_seg;
    const uint32_t *qc = cid + (size_t)q * nprobe;
    const float *qd = cd + (size_t)q * nprobe;

    // ---- phase 0 (wavefront 0): the rows (non-empty probed groups, in probe order) and how many of them need distances
    if (wave == 0) {
        unsigned long long cum = 0;
        int nrows = 0, p1 = -1, r2 = -1;
        for (int base = 0; base < nprobe; base += 64) {
            const int i = base + lane;
            const uint32_t c = i < nprobe ? qc[i] : 0xffffffffu;
            unsigned long long gs = 0;
            if (c < t.nc)
                gs = t.goff[c + 1] - t.goff[c];
            const bool ne = gs != 0;
            const unsigned long long m = __ballot(ne);
            const unsigned long long incl = incl_scan_u64(gs, lane) + cum;
            const int upto = nrows + __popcll(m & ((2ull << lane) - 1ull)); // rows up to and including this lane's
            if (ne) {
                s_rowp[upto - 1] = i;
                s_rc[upto - 1] = c;
                s_rgs[upto - 1] = (uint32_t)gs;
            }
            const unsigned long long h1 = __ballot(ne && incl >= 2 * max_codes);
            const unsigned long long h2 = __ballot(ne && incl >= max_codes);
            if (p1 < 0 && h1)
                p1 = __shfl(upto, __ffsll((long long)h1) - 1, 64);
            if (r2 < 0 && h2)
                r2 = __shfl(upto, __ffsll((long long)h2) - 1, 64);
            nrows += __popcll(m);
            cum = __shfl(incl, 63, 64);
        }
        if (lane == 0) {
            s_p1 = do_pruning ? (p1 >= 0 ? p1 : nrows) : 0;                      // Grouping.cpp:256-258
            s_ra = do_pruning ? (p1 >= 0 ? p1 : nrows) : (r2 >= 0 ? r2 : nrows); // no pruning: what pass 2 will visit
            s_nrows = nrows;
        }
        wave_lds_order();
        for (int r = lane; r < nrows; r += 64) { // lanes over rows: independent loads, one round trip
            const uint32_t c = s_rc[r];
            s_ral[r] = g.alphas[c];
            s_rcn[r] = t.centroid_norms[c];
            s_rlo[r] = t.loff[c];
            s_rqd[r] = qd[s_rowp[r]];
            s_rna[r] = 0;
        }
    }
    __syncthreads();
    const int p1_rows = s_p1, ra = s_ra, nrows = s_nrows;
    stamp(0);

    if constexpr (DEDUPE) {
        // A1: the pairs' centroid ids into the hash set (a pair that finds no slot within GH_PROBES is evaluated directly)
        for (int r = wave; r < ra; r += 4) {
            const uint32_t c = s_rc[r];
            for (int s0 = 0; s0 < nsubc; s0 += 64) {
                const int subc = s0 + lane;
                if (subc < nsubc && g.sub_sizes[(size_t)c * nsubc + subc] != 0) {
                    const uint32_t nn = g.nn_idx[(size_t)c * nsubc + subc];
                    uint32_t slot = gh_hash(nn);
                    for (int pr = 0; pr < GH_PROBES; pr++) {
                        const uint32_t old = atomicCAS(&h_key[slot], GH_EMPTY, nn);
                        if (old == GH_EMPTY || old == nn)
                            break;
                        slot = (slot + 1) & (GH_SLOTS - 1);
                    }
                }
            }
        }
        __syncthreads();
        // A2: the occupied slots, compacted (any order)
        for (int i = tid; i < GH_SLOTS; i += 256)
            if (h_key[i] != GH_EMPTY)
                h_list[atomicAdd(&h_n, 1)] = (uint16_t)i;
        __syncthreads();
        // A3: one distance per distinct centroid, a quad of lanes per row
        const int nd = h_n;
        for (int base = wave * 16; base < nd; base += 64) {
            const int rr = base + (lane >> 2);
            if (rr < nd) {
                const int slot = h_list[rr];
                const float dq = l2_ref_order_quad_batched(gr.vectors + (size_t)h_key[slot] * t.d, s_q, t.d, lane & 3);
                if ((lane & 3) == 0)
                    h_val[slot] = dq;
            }
        }
        __syncthreads();
    }

    // ---- phase A (all four wavefronts, rows dealt round robin): sub-centroid distances (and pass-1 values) to scratch
    for (int r = wave; r < ra; r += 4) {
        const int i = s_rowp[r];
        const uint32_t c = qc[i];
        const float alpha = g.alphas[c];
        const float oma = __fsub_rn(1.0f, alpha);
        const float term1 = __fmul_rn(oma, qd[i]);
        for (int s0 = 0; s0 < nsubc; s0 += 64) {
            const int subc = s0 + lane;
            bool active = false;
            float v = 0.f, qn = 0.f;
            uint32_t nn = 0;
            if (subc < nsubc && g.sub_sizes[(size_t)c * nsubc + subc] != 0) {
                active = true;
                nn = g.nn_idx[(size_t)c * nsubc + subc];
            }
            bool direct = active; // this pair's row is gathered here (always, without DEDUPE)
            if constexpr (DEDUPE) {
                if (active) {
                    uint32_t slot = gh_hash(nn);
                    for (int pr = 0; pr < GH_PROBES; pr++) {
                        const uint32_t kk = h_key[slot];
                        if (kk == nn) {
                            qn = h_val[slot];
                            direct = false;
                            break;
                        }
                        if (kk == GH_EMPTY)
                            break;
                        slot = (slot + 1) & (GH_SLOTS - 1);
                    }
                }
            }
            {
                const unsigned long long am = __ballot(direct);
                const int na = __popcll(am);
                for (int base = 0; base < na; base += 16) {
                    const int rr = base + (lane >> 2);
                    const int src = rr < na ? nth_set_bit(am, rr) : 0;
                    const uint32_t nnq = (uint32_t)__shfl((int)nn, src, 64);
                    float dq = 0.f;
                    if (rr < na)
                        dq = l2_ref_order_quad_batched(gr.vectors + (size_t)nnq * t.d, s_q, t.d, lane & 3);
                    if (rr < na && (lane & 3) == 0)
                        s_dist[src] = dq;
                }
                if (na) {
                    wave_lds_order();
                    if (direct)
                        qn = s_dist[lane];
                    wave_lds_order();
                }
            }
            if (active && do_pruning) {
                const float a = __fmul_rn(oma, g.inter_dists[(size_t)c * nsubc + subc]);
                const float b = __fsub_rn(a, qn);
                v = __fsub_rn(term1, __fmul_rn(alpha, b)); // Grouping.cpp:251-252
            }
            if (subc < nsubc) {
                if (do_pruning)
                    qsd[(size_t)r * nsubc + subc] = v; // value-initialised 0.0 where inactive (:228)
                qnv[(size_t)r * nsubc + subc] = qn;
            }
            const int nact = __popcll(__ballot(active));
            if (lane == 0)
                s_rna[r] += (uint32_t)nact;
        }
    }
    __syncthreads(); // the scratch of this query is complete (workgroup-scope fence)
    if (wave != 0)
        return;
    stamp(1);

    // ---- phase B (wavefront 0): the threshold, summed in (probe, sub-group) order (Grouping.cpp:253, 261)
    float threshold = 0.0f;
    if (do_pruning && nsubc <= 64) {
        // Inactive sub-groups hold 0.0 (value-initialised, :228) and x + 0.0 == x, so the sum over the ACTIVE sub-groups in
        // (row, sub-group) order is the plain sum over all of them in that order: one readlane + one add per element, every
        // row's values loaded a row ahead.  (The loop below -- ballot, find-first, shuffle per active element -- cost 160
        // cycles per element: 217 k cycles of a query's 550 k, stamps.)
        unsigned long long nsubgroups = 0;
        float v = (lane < nsubc && p1_rows > 0) ? qsd[lane] : 0.f;
        for (int r = 0; r < p1_rows; r++) {
            const float vn = (lane < nsubc && r + 1 < p1_rows) ? qsd[(size_t)(r + 1) * nsubc + lane] : 0.f;
            nsubgroups += s_rna[r];
            if (nsubc == 64) { // the reference's preset: 64 readlane + add pairs, straight line
#pragma unroll
                for (int j = 0; j < 64; j++)
                    threshold = __fadd_rn(threshold, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j)));
            } else {
                for (int j = 0; j < nsubc; j++)
                    threshold = __fadd_rn(threshold, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j)));
            }
            v = vn;
        }
        threshold = __fdiv_rn(threshold, (float)nsubgroups); // :261, 0/0 = NaN when nothing was seen
    } else if (do_pruning) {
        unsigned long long nsubgroups = 0;
        for (int r = 0; r < p1_rows; r++) {
            const uint32_t c = qc[s_rowp[r]];
            for (int s0 = 0; s0 < nsubc; s0 += 64) {
                const int subc = s0 + lane;
                const bool active = subc < nsubc && g.sub_sizes[(size_t)c * nsubc + subc] != 0;
                const float v = active ? qsd[(size_t)r * nsubc + subc] : 0.f;
                unsigned long long m = __ballot(active);
                nsubgroups += __popcll(m);
                while (m) {
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    threshold = __fadd_rn(threshold, __shfl(v, j, 64));
                }
            }
        }
        threshold = __fdiv_rn(threshold, (float)nsubgroups); // :261, 0/0 = NaN when nothing was seen
    }
    stamp(2);

    // ---- phase C, one chunk of sub-groups per row (nsubc <= 64, the reference's preset): the same pass 2 with every
    // row's loads issued two rows ahead (sizes, pass-1 value, distance, neighbour id) and its gather of neighbour norms
    // one row ahead.  Written per row in the order of the loop further down, which it replaces: there, each row walked
    // five dependent global round trips -- 224 k cycles of a query's 550 k (stamps).
    if (nsubc <= 64) {
        unsigned long long ncode = 0;
        uint32_t ns = 0, nl = 0;
        Seg *sq = segs + (size_t)q * max_seg;
        uint32_t *lq = lpos + (size_t)q * max_seg;
        const bool inl = lane < nsubc;
        const int lcl = inl ? lane : 0;
        struct RowLoads {
            uint32_t sz, nn;
            float qs, qn;
        };
        auto load_row = [&](int r) {
            RowLoads x;
            const int rr = r < nrows ? r : nrows - 1;
            const size_t o = (size_t)s_rc[rr] * nsubc + lcl, so = (size_t)rr * nsubc + lcl;
            x.sz = g.sub_sizes[o];
            x.nn = g.nn_idx[o];
            x.qs = qsd[so];
            x.qn = qnv[so];
            return x;
        };
        if (nrows > 0) {
            RowLoads cur = load_row(0), nxt = load_row(1);
            float cn_cur = t.centroid_norms[(inl && cur.sz != 0) ? cur.nn : 0u];
            for (int r = 0; r < nrows; r++) {
                const RowLoads nx2 = load_row(r + 2);
                const float cn_nxt = t.centroid_norms[(inl && nxt.sz != 0) ? nxt.nn : 0u];
                const float alpha = s_ral[r];
                const float oma = __fsub_rn(1.0f, alpha);
                const float term1 = __fmul_rn(oma, __fsub_rn(s_rqd[r], s_rcn[r]));
                const uint32_t lo_c = s_rlo[r];
                const bool owned = lo_c != kNotOwned;
                const bool have = r < ra; // this row's distances are in the scratch
                const uint32_t sz = inl ? cur.sz : 0u;
                bool scanned = false;
                if (sz != 0) {
                    const float qs = (do_pruning && r < p1_rows) ? cur.qs : 0.0f;
                    scanned = !do_pruning || qs < threshold; // :308
                }
                float qn2 = (scanned && have) ? cur.qn : 0.f;
                const bool need = scanned && !have;
                const unsigned long long am = __ballot(need);
                if (am) { // rows beyond what the sizes promised (pass 2 ran further than pass 1): evaluated here
                    const int na = __popcll(am);
                    for (int base = 0; base < na; base += 16) {
                        const int rr = base + (lane >> 2);
                        const int src = rr < na ? nth_set_bit(am, rr) : 0;
                        const uint32_t nnq = (uint32_t)__shfl((int)cur.nn, src, 64);
                        float dq = 0.f;
                        if (rr < na)
                            dq = l2_ref_order_quad_batched(gr.vectors + (size_t)nnq * t.d, s_q, t.d, lane & 3);
                        if (rr < na && (lane & 3) == 0)
                            s_dist[src] = dq;
                    }
                    wave_lds_order();
                    if (need)
                        qn2 = s_dist[lane];
                    wave_lds_order();
                }
                float cterm = 0.f;
                if (scanned) {
                    const float term2 = __fmul_rn(alpha, __fsub_rn(qn2, cn_cur)); // :318
                    cterm = __fadd_rn(term1, term2);
                }
                const uint32_t in_sz = wave_incl_scan(sz, lane);
                const uint32_t in_sc = wave_incl_scan(scanned ? sz : 0u, lane);
                const unsigned long long m = __ballot(scanned);
                const uint32_t rank_sc = __popcll(m & ((1ull << lane) - 1ull));
                if (scanned && owned) {
                    Seg sg;
                    sg.start = lo_c + (in_sz - sz);
                    sg.len = sz;
                    sg.vpos = (uint32_t)ncode + (in_sc - sz);
                    sg.cterm = cterm;
                    sq[ns + rank_sc] = sg;
                    lq[ns + rank_sc] = nl + (in_sc - sz);
                }
                const uint32_t tot_sc = __shfl(in_sc, 63, 64);
                ncode += tot_sc;
                if (owned) {
                    ns += (uint32_t)__popcll(m);
                    nl += tot_sc;
                }
                if (ncode >= max_codes)
                    break;
                cur = nxt;
                cn_cur = cn_nxt;
                nxt = nx2;
            }
        }
        if (lane == 0) {
            PlanHdr h;
            h.nseg = ns;
            h.total = nl;
            hdr[q] = h;
        }
        stamp(3);
        if (stamps && threadIdx.x == 0)
            atomicAdd(&stamps[4], 1ull);
        return;
    }

    // ---- phase C (wavefront 0): pass 2 (Grouping.cpp:283-353), distances from the scratch for the rows phase A covered
    unsigned long long ncode = 0; // codes scored so far == scan position of the next one
    uint32_t ns = 0, nl = 0;
    int row = 0;
    Seg *sq = segs + (size_t)q * max_seg;
    uint32_t *lq = lpos + (size_t)q * max_seg;
    for (int i = 0; i < nprobe; i++) {
        const uint32_t c = qc[i];
        if (c >= t.nc)
            continue;
        const unsigned long long gs = t.goff[c + 1] - t.goff[c];
        if (gs == 0)
            continue;
        const float alpha = g.alphas[c];
        const float oma = __fsub_rn(1.0f, alpha);
        const float term1 = __fmul_rn(oma, __fsub_rn(qd[i], t.centroid_norms[c]));
        const uint32_t lo_c = t.loff[c];
        const bool owned = lo_c != kNotOwned;
        const bool have = row < ra; // this row's distances are in the scratch
        uint32_t list_off = 0;      // codes of this list before the current chunk of sub-groups
        for (int s0 = 0; s0 < nsubc; s0 += 64) {
            const int subc = s0 + lane;
            uint32_t sz = 0;
            bool scanned = false;
            float cterm = 0.f;
            if (subc < nsubc)
                sz = g.sub_sizes[(size_t)c * nsubc + subc];
            if (sz != 0) {
                float qs = 0.0f;
                if (do_pruning && row < p1_rows)
                    qs = qsd[(size_t)row * nsubc + subc];
                scanned = !do_pruning || qs < threshold; // :308
            }
            float qn2 = 0.f;
            const bool need = scanned && !have;
            if (scanned && have)
                qn2 = qnv[(size_t)row * nsubc + subc];
            {
                const uint32_t nn = scanned ? g.nn_idx[(size_t)c * nsubc + subc] : 0u;
                const unsigned long long am = __ballot(need);
                const int na = __popcll(am);
                if (na) { // rows beyond what the sizes promised (pass 2 ran further than pass 1): evaluated here
                    for (int base = 0; base < na; base += 16) {
                        const int rr = base + (lane >> 2);
                        const int src = rr < na ? nth_set_bit(am, rr) : 0;
                        const uint32_t nnq = (uint32_t)__shfl((int)nn, src, 64);
                        float dq = 0.f;
                        if (rr < na)
                            dq = l2_ref_order_quad_batched(gr.vectors + (size_t)nnq * t.d, s_q, t.d, lane & 3);
                        if (rr < na && (lane & 3) == 0)
                            s_dist[src] = dq;
                    }
                    wave_lds_order();
                    if (need)
                        qn2 = s_dist[lane];
                    wave_lds_order();
                }
                if (scanned) {
                    const float term2 = __fmul_rn(alpha, __fsub_rn(qn2, t.centroid_norms[nn])); // :318
                    cterm = __fadd_rn(term1, term2);
                }
            }
            const uint32_t in_sz = wave_incl_scan(sz, lane);
            const uint32_t in_sc = wave_incl_scan(scanned ? sz : 0u, lane);
            const unsigned long long m = __ballot(scanned);
            const uint32_t rank_sc = __popcll(m & ((1ull << lane) - 1ull));
            if (scanned && owned) {
                Seg sg;
                sg.start = lo_c + list_off + (in_sz - sz);
                sg.len = sz;
                sg.vpos = (uint32_t)ncode + (in_sc - sz);
                sg.cterm = cterm;
                sq[ns + rank_sc] = sg;
                lq[ns + rank_sc] = nl + (in_sc - sz);
            }
            const uint32_t tot_sz = __shfl(in_sz, 63, 64), tot_sc = __shfl(in_sc, 63, 64);
            list_off += tot_sz;
            ncode += tot_sc;
            if (owned) {
                ns += (uint32_t)__popcll(m);
                nl += tot_sc;
            }
        }
        if (ncode >= max_codes)
            break;
        row++;
    }
    if (lane == 0) {
        PlanHdr h;
        h.nseg = ns;
        h.total = nl;
        hdr[q] = h;
    }
    stamp(3);
    if (stamps && threadIdx.x == 0)
        atomicAdd(&stamps[4], 1ull);
}

hipError_t launch_plan_grouping(hipStream_t s, const IvfTables &t, const GroupTables &g, const GraphTables &gr,
                                const float *xq, const uint32_t *coarse_ids, const float *coarse_dists, int nq,
                                int nprobe, uint64_t max_codes, int do_pruning, Seg *segs, uint32_t *lpos,
                                PlanHdr *hdr, int max_seg, uint64_t *keys, int k, float *scratch)
{
    if (nq == 0)
        return hipSuccess;
    // IVFHNSW_PLAN_GROUP4=0: the one-wavefront-per-query form (A/B runs)
    static const bool four = [] {
        const char *e = getenv("IVFHNSW_PLAN_GROUP4");
        return !(e && *e && atoi(e) == 0);
    }();
    static unsigned long long *d_st = [] {
        const char *e = getenv("IVFHNSW_PLAN_STAMPS");
        unsigned long long *p = nullptr;
        if (e && atoi(e) == 1 && hipMalloc(&p, 8 * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemset(p, 0, 8 * sizeof(unsigned long long));
            static unsigned long long *keep = p;
            atexit([] {
                unsigned long long h[8];
                if (hipMemcpy(h, keep, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[4])
                    fprintf(stderr, "[plan stamps] %llu queries; cycles per query (100 MHz s_memtime ticks x 1): rows %llu, "
                                    "phase A (4 wavefronts) %llu, phase B (threshold) %llu, phase C (pass 2) %llu\n",
                            h[4], h[0] / h[4], h[1] / h[4], h[2] / h[4], h[3] / h[4]);
            });
        }
        return p;
    }();
    // IVFHNSW_PLAN_DEDUPE=0 / 1 overrides what upload_grouping measured on the index's neighbour lists (g.dedupe)
    static const int dedupe_knob = [] {
        const char *e = getenv("IVFHNSW_PLAN_DEDUPE");
        return (e && *e) ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    const bool dedupe = dedupe_knob < 0 ? g.dedupe != 0 : dedupe_knob == 1;
    const size_t shm4 = (size_t)(t.d + 256) * sizeof(float) + (size_t)nprobe * 8 * sizeof(int);
    if (four && dedupe)
        hipLaunchKernelGGL(plan_grouping4_kernel<true>, dim3(nq), dim3(256), shm4, s, t, g, gr, xq, coarse_ids,
                           coarse_dists, nq, nprobe, (unsigned long long)max_codes, do_pruning, segs, lpos, hdr, max_seg,
                           reinterpret_cast<unsigned long long *>(keys), k, scratch, d_st);
    else if (four)
        hipLaunchKernelGGL(plan_grouping4_kernel<false>, dim3(nq), dim3(256), shm4, s, t, g, gr, xq, coarse_ids,
                           coarse_dists, nq, nprobe, (unsigned long long)max_codes, do_pruning, segs, lpos, hdr, max_seg,
                           reinterpret_cast<unsigned long long *>(keys), k, scratch, d_st);
    else
        hipLaunchKernelGGL(plan_grouping_kernel, dim3(nq), dim3(64), (t.d + 64) * sizeof(float), s, t, g, gr, xq,
                           coarse_ids, coarse_dists, nq, nprobe, (unsigned long long)max_codes, do_pruning, segs, lpos, hdr,
                           max_seg, reinterpret_cast<unsigned long long *>(keys), k, scratch);
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
