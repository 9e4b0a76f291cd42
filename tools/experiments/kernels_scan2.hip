// Fused table + scan, k = 1 (IndexIVF_HNSW.cpp:262 and :282-289 / IndexIVF_HNSW_Grouping.cpp:265, 321-333 in
// one kernel): the query's inner-product table is built straight into LDS by the workgroup that scans with it,
// from a code book that never leaves the register file.
//
// Why: lut_kernel + scan_k1_kernel move every query's table through HBM twice (M KB written, M KB staged into
// LDS), and re-read the 128-KB code book from L2 for every four queries.  On one GPU that is 9 % of the scan's
// bytes; on N list-wise shards every rank needs the table of (nearly) every query for 1/N of its codes, so the
// tables grow to the size of the code stream itself.  Here a persistent 512-thread workgroup keeps
// the WHOLE code book distributed over its registers -- thread (h, c) holds centroid[m][c][:] for the m of half h:
// d/2 floats = 64 VGPRs at d = 128 -- and loops over queries: 8 table entries per thread from scalar-loaded query
// components (faiss's SSE order, ip_sse_order: bit-identical to lut_kernel), written conflict-free to LDS; then the
// same eight wavefronts scan the query's plan exactly as scan_k1_kernel does.  Nothing of the table touches memory.
//
// Occupancy: ~110 VGPRs -> 4 waves per SIMD = 2 workgroups per CU (the plain scan runs 8 per SIMD); the scan keeps
// U = 4 sixteen-byte loads per lane in flight, 64 KB per CU.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

namespace {

__device__ __forceinline__ uint64_t wave_min_u64_f(uint64_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint64_t o = __shfl_xor((unsigned long long)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

constexpr int FT = 512; // threads per workgroup: two halves of 256 code words

// SHORT = the plan is made of short segments (Grouping: a sub-group holds ~16 codes at the reference's nsubc 64).
// Positions are still dealt to lanes one code each -- every lane busy, the LDS gathers at full width -- and the
// segment of a position comes from a bitmap instead of a per-lane search: one bit per position of the chunk, set
// where a segment starts, plus the number of starts before every 64-position word.  A wavefront's 64 positions
// share a word, so   segment = pref[word] + popcount(mask[word] & lanes_up_to_mine) - 1   costs one broadcast LDS
// read and two v_mbcnt.  (scan_k1_short_kernel gave a lane GROUP to every segment: half the lanes idle at this
// segment length, and the kernel is bound by LDS instruction issue.)
constexpr int SPANCAP = 32768;     // positions of one plan chunk the bitmap covers
constexpr int SPANW = SPANCAP / 64;

template <int CS, int DSUB, int SEGCAP, int U, bool SHORT>
__global__ __launch_bounds__(FT) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_fused_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ norm_codes, const float *__restrict__ xq,
    const float *__restrict__ cb, const float *__restrict__ norm_table, const Seg *__restrict__ segs,
    const uint32_t *__restrict__ lpos, const PlanHdr *__restrict__ hdr, int max_seg, int nq, int nsplit,
    unsigned long long *__restrict__ keys, uint32_t *__restrict__ counter)
{
    constexpr int D = CS * DSUB;
    constexpr int MH = CS / 2; // sub-quantizers per half
    __shared__ __attribute__((aligned(16))) float s_lut[CS * 256];
    __shared__ float s_norm[256];
    __shared__ __attribute__((aligned(16))) Seg s_seg[SEGCAP];
    __shared__ uint32_t s_lpos[SEGCAP + 1];
    __shared__ unsigned long long s_red[FT / 64];
    __shared__ uint32_t s_item;
    __shared__ unsigned long long s_mask[SHORT ? SPANW : 1];
    __shared__ uint32_t s_pref[SHORT ? SPANW : 1];
    __shared__ uint32_t s_wtot[FT / 64];

    const int tid = threadIdx.x;
    const int c = tid & 255;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);

    // this thread's share of the code book, resident for the whole launch
    float row[MH][DSUB];
#pragma unroll
    for (int m = 0; m < MH; m++) {
        const float *src = cb + ((size_t)(half * MH + m) * 256 + c) * DSUB;
        if constexpr (DSUB % 4 == 0) {
#pragma unroll
            for (int i = 0; i < DSUB; i += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src + i);
                row[m][i] = v.x, row[m][i + 1] = v.y, row[m][i + 2] = v.z, row[m][i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < DSUB; i += 2) {
                const float2 v = *reinterpret_cast<const float2 *>(src + i);
                row[m][i] = v.x, row[m][i + 1] = v.y;
            }
        }
    }
    if (tid < 256)
        s_norm[tid] = norm_table[tid];

    const uint32_t nitems = (uint32_t)nq * (uint32_t)nsplit;
    for (;;) {
        __syncthreads(); // the previous item's table, plan chunk and reduction slots are free again
        if (tid == 0)
            s_item = atomicAdd(counter, 1u);
        __syncthreads();
        const uint32_t item = __builtin_amdgcn_readfirstlane(s_item);
        if (item >= nitems)
            break;
        const int q = (int)(item / (uint32_t)nsplit);
        const int split = (int)(item - (uint32_t)q * (uint32_t)nsplit);
        const PlanHdr h = hdr[q];
        if (h.total == 0)
            continue; // nothing of this query lives on this shard: no table either
        // this split's slice of the virtual code array, in multiples of the block width
        uint32_t per = (h.total + nsplit - 1) / nsplit;
        per = (per + (FT - 1)) & ~(uint32_t)(FT - 1);
        const uint32_t lo = min((uint32_t)split * per, h.total);
        const uint32_t hi = min(lo + per, h.total);
        if (lo >= hi)
            continue;

        // ---- table: tab[m][c] = <x_m, centroid[m][c]> for this thread's m (IndexIVF_HNSW.cpp:262)
        {
            const float *x = xq + (size_t)q * D + half * (MH * DSUB); // wave-uniform: scalar loads
#pragma unroll
            for (int m = 0; m < MH; m++) {
                float xs[DSUB];
#pragma unroll
                for (int i = 0; i < DSUB; i++)
                    xs[i] = x[m * DSUB + i];
                s_lut[(half * MH + m) * 256 + c] = ip_sse_order<DSUB>(xs, row[m], DSUB);
            }
        }

        // ---- scan (scan_k1_kernel's loop with a 512-wide block)
        const Seg *sq = segs + (size_t)q * max_seg;
        const uint32_t *lq = lpos + (size_t)q * max_seg;
        unsigned long long best = kKeyInit;
        if constexpr (SHORT) {
            const int lane = tid & 63, wave = tid >> 6;
            for (uint32_t cs = 0; cs < h.nseg;) {
                // as many segments as fit the plan buffer AND the bitmap's span; one oversized segment goes alone
                const uint32_t cl = lq[cs];
                uint32_t cn = min((uint32_t)SEGCAP, h.nseg - cs);
                {
                    const uint32_t end_all = (cs + cn == h.nseg) ? h.total : lq[cs + cn];
                    if (end_all - cl > (uint32_t)SPANCAP) {
                        uint32_t a = 1, b = cn > 1 ? cn - 1 : 1; // largest count in [1, cn) whose end stays inside the span (or 1)
                        while (a < b) {
                            const uint32_t mid = (a + b + 1) >> 1;
                            if (lq[cs + mid] - cl <= (uint32_t)SPANCAP)
                                a = mid;
                            else
                                b = mid - 1;
                        }
                        cn = a;
                    }
                }
                const uint32_t ch = (cs + cn == h.nseg) ? h.total : lq[cs + cn];
                const bool single = ch - cl > (uint32_t)SPANCAP; // cn == 1: every position is segment 0
                const uint32_t nwords = single ? 0u : (ch - cl + 63) >> 6;
                __syncthreads(); // previous chunk fully consumed (and the table complete, first time)
                for (uint32_t i = tid; i < cn; i += FT) {
                    s_seg[i] = sq[cs + i];
                    s_lpos[i] = lq[cs + i];
                }
                for (uint32_t i = tid; i < nwords; i += FT)
                    s_mask[i] = 0ull;
                __syncthreads();
                if (!single) {
                    uint32_t *m32 = reinterpret_cast<uint32_t *>(s_mask);
                    for (uint32_t i = tid; i < cn; i += FT) {
                        const uint32_t r = s_lpos[i] - cl;
                        atomicOr(&m32[r >> 5], 1u << (r & 31));
                    }
                    __syncthreads();
                    // starts before every word: a wave scan per 64 words, then the waves' totals
                    uint32_t run = 0;
                    for (uint32_t w0 = 0; w0 < nwords; w0 += FT) {
                        const uint32_t w = w0 + tid;
                        const uint32_t cnt = w < nwords ? (uint32_t)__popcll(s_mask[w]) : 0u;
                        const uint32_t inc = wave_incl_scan(cnt, lane);
                        if (lane == 63)
                            s_wtot[wave] = inc;
                        __syncthreads();
                        uint32_t before = run;
#pragma unroll
                        for (int j = 0; j < FT / 64; j++) {
                            const uint32_t tj = s_wtot[j];
                            before += j < wave ? tj : 0u;
                            run += tj;
                        }
                        if (w < nwords)
                            s_pref[w] = before + inc - cnt;
                        __syncthreads();
                    }
                }
                const uint32_t b0 = max(cl, lo), b1 = min(ch, hi);
                if (b0 < b1) {
                    for (uint32_t rbase = (b0 - cl) & ~63u; rbase < b1 - cl; rbase += FT * U) {
                        uint32_t w[U][CS / 4];
                        uint32_t nb[U], vp[U];
                        float ct[U];
                        bool ok[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const uint32_t r = rbase + u * FT + tid;
                            const uint32_t p = cl + r;
                            ok[u] = p >= b0 && p < b1;
                            const uint32_t wi = __builtin_amdgcn_readfirstlane(r >> 6);
                            if (wi * 64u < b1 - cl) { // wave-uniform
                                uint32_t sgi = 0;
                                if (!single) {
                                    const unsigned long long mw = s_mask[wi];
                                    const uint32_t below = __builtin_amdgcn_mbcnt_hi(
                                        (uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, 0u));
                                    sgi = s_pref[wi] + below + (uint32_t)((mw >> lane) & 1ull) - 1u;
                                }
                                if (ok[u]) {
                                    const Seg sg = s_seg[sgi];
                                    const uint32_t off = p - s_lpos[sgi];
                                    const uint32_t gi = sg.start + off;
                                    load_code_words<CS>(codes, gi, w[u]);
                                    nb[u] = norm_codes[gi];
                                    vp[u] = sg.vpos + off;
                                    ct[u] = sg.cterm;
                                }
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
                cs += cn;
            }
        } else {
            for (uint32_t cs = 0; cs < h.nseg; cs += SEGCAP) {
                const uint32_t cn = min((uint32_t)SEGCAP, h.nseg - cs);
                __syncthreads(); // previous chunk fully consumed (and the table complete, first time)
                for (uint32_t i = tid; i < cn; i += FT) {
                    s_seg[i] = sq[cs + i];
                    s_lpos[i] = lq[cs + i];
                }
                const uint32_t ch = (cs + cn == h.nseg) ? h.total : lq[cs + cn];
                if (tid == 0)
                    s_lpos[cn] = ch;
                __syncthreads();
                const uint32_t cl = s_lpos[0];
                const uint32_t b0 = max(cl, lo), b1 = min(ch, hi);
                uint32_t s = 0;
                uint32_t seg_lo = 0, seg_hi = 0, seg_start = 0, seg_vpos = 0;
                float seg_ct = 0.f;
                for (uint32_t base = b0; base < b1; base += FT * U) {
                    uint32_t w[U][CS / 4];
                    uint32_t nb[U], vp[U];
                    float ct[U];
                    bool ok[U];
    #pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t p = base + u * FT + tid;
                        ok[u] = p < b1;
                        if (ok[u]) {
                            if (p >= seg_hi) {
                                if constexpr (SEGCAP <= 64) {
                                    while (p >= s_lpos[s + 1])
                                        s++;
                                } else {
                                    uint32_t a = s, b = cn - 1;
                                    if (a < b && s_lpos[a + 1] <= p) {
                                        a++;
                                        if (a < b && s_lpos[a + 1] <= p)
                                            a++;
                                        else
                                            b = a;
                                    } else {
                                        b = a;
                                    }
                                    while (a < b) {
                                        const uint32_t mid = (a + b) >> 1;
                                        if (s_lpos[mid + 1] > p)
                                            b = mid;
                                        else
                                            a = mid + 1;
                                    }
                                    s = a;
                                }
                                const Seg sg = s_seg[s];
                                seg_lo = s_lpos[s];
                                seg_hi = seg_lo + sg.len;
                                seg_start = sg.start;
                                seg_vpos = sg.vpos;
                                seg_ct = sg.cterm;
                            }
                            const uint32_t off = p - seg_lo;
                            const uint32_t gi = seg_start + off;
                            load_code_words<CS>(codes, gi, w[u]);
                            nb[u] = norm_codes[gi];
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
                            if (dist < FLT_MAX) { // also rejects NaN, as 'dist < distances[0]' does
                                const unsigned long long key =
                                    ((unsigned long long)f32_orderable(__fadd_rn(dist, 0.0f)) << 32) | vp[u];
                                best = key < best ? key : best;
                            }
                        }
                    }
                }
            }

        }

        best = wave_min_u64_f(best);
        if ((tid & 63) == 0)
            s_red[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = s_red[0];
#pragma unroll
            for (int i = 1; i < FT / 64; i++)
                b = s_red[i] < b ? s_red[i] : b;
            if (nsplit == 1)
                keys[q] = b;
            else if (b < kKeyInit)
                atomicMin(&keys[q], b);
        }
    }
}

} // namespace

// Where the fused form is used (measured on MI355X, 1B codes, profiles/r02_fused_ab.md): NOWHERE by default.
//   * one GPU, whole lists: 0.506 ms against 0.052 (tables) + 0.366 (scan_k1_kernel);
//   * one rank of eight list shards, 80 k queries: 0.971 ms against 0.262 + 0.476 -- and a rank holding HALF the codes
//     of another takes the same time: the kernel is bound by its per-query dependent chain (queue -> plan header ->
//     query -> table -> barrier -> plan -> one scan iteration -> reduction, ~6 us) with only two workgroups per CU to
//     overlap it, where the plain scan keeps eight independent workgroups per CU in flight;
//   * sub-group plans (Grouping): 0.561 ms against 0.053 + 0.579 for scan_k1_short_kernel -- a gain that came from the
//     bitmap segment lookup, not from the fusion: scan_k1_bitmap_kernel (kernels_search.hip) has the lookup without it.
// Kept as an exact, tested form behind IVFHNSW_SCAN_FUSED=1 (the differential suite passes with it forced on): what
// would make it pay is a software pipeline over queries (next query's header, vector and plan in flight during the
// current scan).
bool scan_fused_supported(const IvfTables &t, bool short_segments)
{
    static const int knob = [] {
        const char *e = getenv("IVFHNSW_SCAN_FUSED");
        return (e && *e) ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    if (knob == 0)
        return false;
    const bool shape = (t.M == 16 && (t.dsub == 8 || t.dsub == 6)) || (t.M == 8 && (t.dsub == 16 || t.dsub == 12));
    if (!shape)
        return false;
    (void)short_segments;
    return knob == 1;
}

hipError_t launch_scan_fused(hipStream_t s, const IvfTables &t, const float *xq, const Seg *segs, const uint32_t *lpos,
                             const PlanHdr *hdr, int max_seg, int nq, int nsplit, uint64_t *keys, uint32_t *counter,
                             bool short_segments)
{
    if (nq == 0)
        return hipSuccess;
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint32_t), s);
    if (e != hipSuccess)
        return e;
    static const int resident = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 2 * cus; // two 512-thread workgroups per CU at 4 waves per SIMD
    }();
    const long long items = (long long)nq * nsplit;
    const dim3 grid((unsigned)(items < resident ? items : resident)), block(FT);
    auto *k64 = reinterpret_cast<unsigned long long *>(keys);
#define IVFHNSW_FUSED(CS, DSUB, SEGCAP, SH)                                                                            \
    hipLaunchKernelGGL((scan_fused_kernel<CS, DSUB, SEGCAP, 4, SH>), grid, block, 0, s, t.codes, t.norm_codes, xq,       \
                       t.pq_centroids, t.norm_table, segs, lpos, hdr, max_seg, nq, nsplit, k64, counter)
#define IVFHNSW_FUSED_SEG(CS, DSUB)                 \
    do {                                            \
        if (short_segments)                         \
            IVFHNSW_FUSED(CS, DSUB, 1024, true);    \
        else if (max_seg <= 64)                     \
            IVFHNSW_FUSED(CS, DSUB, 64, false);     \
        else if (max_seg <= 256)                    \
            IVFHNSW_FUSED(CS, DSUB, 256, false);    \
        else                                        \
            IVFHNSW_FUSED(CS, DSUB, 1024, false);   \
    } while (0)
    if (t.M == 16 && t.dsub == 8)
        IVFHNSW_FUSED_SEG(16, 8);
    else if (t.M == 16 && t.dsub == 6)
        IVFHNSW_FUSED_SEG(16, 6);
    else if (t.M == 8 && t.dsub == 16)
        IVFHNSW_FUSED_SEG(8, 16);
    else if (t.M == 8 && t.dsub == 12)
        IVFHNSW_FUSED_SEG(8, 12);
    else
        return hipErrorInvalidValue;
#undef IVFHNSW_FUSED_SEG
#undef IVFHNSW_FUSED
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
