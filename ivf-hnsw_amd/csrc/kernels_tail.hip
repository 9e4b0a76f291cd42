// Everything behind the coarse stage in ONE launch, for small batches (IndexIVF_HNSW.cpp:262-293 for a handful of
// queries): the scan plan, the query's inner-product table, the ADC scan of its share of the plan, and -- by the last
// workgroup of a query to finish -- the label resolution and the result write.
//
// Why: one query per call is how the reference's drivers search (tests/test_ivfhnsw_sift1b.cpp:193-208).  plan_ivf_kernel,
// lut_kernel, scan_k1_kernel and select_kernel are four dependent launches of a few microseconds of work each; on an idle
// chip each costs ~7 us of dispatch and dependency latency (30 us of a 215 us call).  Here nq * nsplit workgroups each
// derive the (tiny) plan themselves, build the table from the L2-resident code book, scan their slice, and meet in one
// atomic per query.  Same arithmetic in the same order as the four kernels (plan rule of plan_ivf_kernel, ip_sse_order,
// the m-sequential ADC sum, packed (distance, scan position) keys), so the same bits.
//
// keys_inv[q] holds the bitwise complement of the best key (atomicMax; 0 = nothing yet, so a plain memset initialises
// it), done[q] counts finished workgroups of query q.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>

namespace ivfhnsw_gpu_impl {

namespace {

__device__ __forceinline__ unsigned long long tail_wave_min(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long tail_wave_scan(unsigned long long v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(v, off, 64);
        if (lane >= off)
            v += o;
    }
    return v;
}

template <int CS, int DSUB, int U>
__global__ __launch_bounds__(256) void ivf_tail_kernel(IvfTables t, const float *__restrict__ xq,
                                                       const uint32_t *__restrict__ cid, const float *__restrict__ cd,
                                                       int nq, int nprobe, unsigned long long max_codes, int nsplit,
                                                       unsigned long long *__restrict__ keys_inv,
                                                       uint32_t *__restrict__ done, PlanHdr *__restrict__ hdr_out,
                                                       float *__restrict__ dist_out, long long *__restrict__ labels_out,
                                                       const uint32_t *__restrict__ status_word,
                                                       uint32_t *__restrict__ status_out)
{
    constexpr int D = CS * DSUB;
    __shared__ __attribute__((aligned(16))) float s_lut[CS * 256];
    __shared__ float s_norm[256];
    __shared__ __attribute__((aligned(16))) float s_x[D];
    __shared__ __attribute__((aligned(16))) Seg s_seg[64];
    __shared__ uint32_t s_lpos[65];
    __shared__ uint32_t s_nseg, s_last;
    __shared__ unsigned long long s_red[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x / nsplit;
    const int split = blockIdx.x - q * nsplit;

    // ---- stage the query and the norm table; wave 0 derives the plan (plan_ivf_kernel for nprobe <= 64)
    if (tid < D)
        s_x[tid] = xq[(size_t)q * D + tid];
    s_norm[tid] = t.norm_table[tid];
    if (wave == 0) {
        uint32_t c = 0xffffffffu;
        if (lane < nprobe)
            c = cid[(size_t)q * nprobe + lane];
        const bool ok = c < t.nc;
        unsigned long long n = 0;
        if (ok)
            n = t.goff[c + 1] - t.goff[c];
        const unsigned long long incl = tail_wave_scan(n, lane);
        const unsigned long long excl = incl - n;
        // the check follows the scoring (IndexIVF_HNSW.cpp:290-292): the first non-empty list is always scored
        const bool take = n != 0 && (excl < max_codes || excl == 0);
        const uint32_t lo_c = take ? t.loff[c] : kNotOwned;
        const bool owned = lo_c != kNotOwned;
        const unsigned long long om = __ballot(owned);
        const unsigned long long own_incl = tail_wave_scan(owned ? n : 0ull, lane);
        if (owned) {
            const uint32_t r = (uint32_t)__popcll(om & ((1ull << lane) - 1ull));
            Seg sg;
            sg.start = lo_c;
            sg.len = (uint32_t)n;
            sg.vpos = (uint32_t)excl;
            sg.cterm = __fsub_rn(cd[(size_t)q * nprobe + lane], t.centroid_norms[c]);
            s_seg[r] = sg;
            s_lpos[r] = (uint32_t)(own_incl - n);
        }
        const uint32_t ns = (uint32_t)__popcll(om);
        const uint32_t nl = (uint32_t)__shfl(own_incl, 63, 64);
        if (lane == 0) {
            s_nseg = ns;
            s_lpos[ns] = nl;
            if (split == 0 && hdr_out) { // the accounting of ivfhnsw_gpu_last_scan_counts
                PlanHdr h;
                h.nseg = ns;
                h.total = nl;
                hdr_out[q] = h;
            }
        }
    }
    __syncthreads();
    const uint32_t nseg = s_nseg;
    const uint32_t total = s_lpos[nseg];

    unsigned long long best = kKeyInit;
    uint32_t per = (total + nsplit - 1) / nsplit;
    per = (per + 255u) & ~255u;
    const uint32_t lo = min((uint32_t)split * per, total);
    const uint32_t hi = min(lo + per, total);
    if (lo < hi) { // block-uniform
        // ---- table: thread c builds tab[m][c] for every m (lut_kernel's arithmetic; the code book comes from L2)
        {
            const int c = tid;
#pragma unroll 4
            for (int m = 0; m < CS; m++) {
                const float *src = t.pq_centroids + ((size_t)m * 256 + c) * DSUB;
                float row[DSUB];
                if constexpr (DSUB % 4 == 0) {
#pragma unroll
                    for (int i = 0; i < DSUB; i += 4) {
                        const float4 v = *reinterpret_cast<const float4 *>(src + i);
                        row[i] = v.x, row[i + 1] = v.y, row[i + 2] = v.z, row[i + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < DSUB; i += 2) {
                        const float2 v = *reinterpret_cast<const float2 *>(src + i);
                        row[i] = v.x, row[i + 1] = v.y;
                    }
                }
                s_lut[m * 256 + c] = ip_sse_order<DSUB>(s_x + m * DSUB, row, DSUB);
            }
        }
        __syncthreads();
        // ---- scan of [lo, hi) (scan_k1_kernel's loop, one plan chunk)
        uint32_t s = 0, seg_lo = 0, seg_hi = 0, seg_start = 0, seg_vpos = 0;
        float seg_ct = 0.f;
        for (uint32_t base = lo; base < hi; base += 256 * U) {
            uint32_t w[U][CS / 4];
            uint32_t nb[U], vp[U];
            float ct[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t p = base + u * 256 + tid;
                ok[u] = p < hi;
                if (ok[u]) {
                    if (p >= seg_hi) {
                        while (p >= s_lpos[s + 1])
                            s++;
                        const Seg sg = s_seg[s];
                        seg_lo = s_lpos[s];
                        seg_hi = seg_lo + sg.len;
                        seg_start = sg.start;
                        seg_vpos = sg.vpos;
                        seg_ct = sg.cterm;
                    }
                    const uint32_t off = p - seg_lo;
                    const uint32_t gi = seg_start + off;
                    load_code_words<CS>(t.codes, gi, w[u]);
                    nb[u] = t.norm_codes[gi];
                    vp[u] = seg_vpos + off;
                    ct[u] = seg_ct;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (ok[u]) {
                    const float sum = adc_sum<CS>(s_lut, w[u]);
                    const float tt = __fadd_rn(ct[u], s_norm[nb[u]]);
                    const float dist = __fsub_rn(tt, __fmul_rn(2.0f, sum));
                    if (dist < FLT_MAX) {
                        const unsigned long long key =
                            ((unsigned long long)f32_orderable(__fadd_rn(dist, 0.0f)) << 32) | vp[u];
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
    best = tail_wave_min(best);
    if (lane == 0)
        s_red[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = s_red[0];
#pragma unroll
        for (int i = 1; i < 4; i++)
            b = s_red[i] < b ? s_red[i] : b;
        if (b < kKeyInit)
            atomicMax(&keys_inv[q], ~b);
        __threadfence();
        s_last = atomicAdd(&done[q], 1u) == (uint32_t)nsplit - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || tid != 0)
        return;
    // ---- the last workgroup of the query: the winner's scan position -> label (the plan is still in LDS)
    __threadfence();
    const unsigned long long inv = atomicMax(&keys_inv[q], 0ull); // a read that sees the other workgroups' maxima
    const unsigned long long key = ~inv;
    float dv = FLT_MAX;
    long long lb = -1;
    if (inv != 0ull && key < kKeyInit) {
        dv = orderable_f32((uint32_t)(key >> 32));
        const uint32_t vpos = (uint32_t)key;
        for (uint32_t i = 0; i < nseg; i++) {
            const Seg sg = s_seg[i];
            if (vpos >= sg.vpos && vpos - sg.vpos < sg.len) {
                lb = (long long)t.ids[sg.start + (vpos - sg.vpos)];
                break;
            }
        }
    }
    dist_out[q] = dv;
    labels_out[q] = lb;
    if (q == 0 && status_out)
        *status_out = *status_word;
}

} // namespace

bool ivf_tail_supported(const IvfTables &t, int nprobe, int k)
{
    // IVFHNSW_TAIL=0 keeps the four launches (A/B runs)
    static const bool off = [] {
        const char *e = getenv("IVFHNSW_TAIL");
        return e && *e && atoi(e) == 0;
    }();
    if (off || k != 1 || nprobe > 64 || t.shard_world != 1)
        return false;
    return (t.M == 16 && (t.dsub == 8 || t.dsub == 6)) || (t.M == 8 && (t.dsub == 16 || t.dsub == 12));
}

hipError_t launch_ivf_tail(hipStream_t s, const IvfTables &t, const float *xq, const uint32_t *cid, const float *cd,
                           int nq, int nprobe, uint64_t max_codes, int nsplit, uint64_t *keys_inv, uint32_t *done,
                           PlanHdr *hdr, float *dist, int64_t *labels, const uint32_t *status_word, uint32_t *status_out)
{
    if (nq == 0)
        return hipSuccess;
    const dim3 grid((unsigned)nq * nsplit), block(256);
#define IVFHNSW_TAIL(CS, DSUB)                                                                                        \
    hipLaunchKernelGGL((ivf_tail_kernel<CS, DSUB, 4>), grid, block, 0, s, t, xq, cid, cd, nq, nprobe,                 \
                       (unsigned long long)max_codes, nsplit, reinterpret_cast<unsigned long long *>(keys_inv), done, \
                       hdr, dist, reinterpret_cast<long long *>(labels), status_word, status_out)
    if (t.M == 16 && t.dsub == 8)
        IVFHNSW_TAIL(16, 8);
    else if (t.M == 16 && t.dsub == 6)
        IVFHNSW_TAIL(16, 6);
    else if (t.M == 8 && t.dsub == 16)
        IVFHNSW_TAIL(8, 16);
    else if (t.M == 8 && t.dsub == 12)
        IVFHNSW_TAIL(8, 12);
    else
        return hipErrorInvalidValue;
#undef IVFHNSW_TAIL
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
