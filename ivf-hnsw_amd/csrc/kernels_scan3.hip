// Table + scan for list SHARDS, software-pipelined over queries (k = 1; IndexIVF_HNSW.cpp:262 and :282-289).
//
// Why: on N list-wise shards every rank needs the inner-product table of (nearly) every query of the N-fold batch for
// 1/N of its codes: at N = 8 the 16-KB table of a query is built, written to HBM and staged back for
// ~1.3 k codes (22 KB) -- lut_kernel + scan_k1_kernel spend 0.26 + 0.48 ms per rank where one GPU spends 0.05 + 0.36
// on the same number of codes.  scan_fused_kernel (kernels_scan2.hip) removed the table traffic and lost: one query
// costs it a CHAIN of dependent round trips (queue atomic -> plan header -> query -> table -> barrier -> plan ->
// codes -> reduction -> barrier, ~6 us) with two workgroups per CU to overlap it.
//
// Here the chain is cut into a pipeline.  A workgroup's queries are fixed up front (q = block + j * grid: no queue),
// and while item j is scored from codes ALREADY in registers and a table ALREADY in LDS,
//   * the table of item j+1 is built into the other table buffer (code book resident in registers, as in the fused form),
//   * the codes of item j+1 are requested (its plan has been in LDS since the previous iteration),
//   * plan and query of item j+2 move from registers to LDS, and those of item j+3 are requested.
// Every global load is consumed one full iteration after it was issued, there is ONE barrier per query (hand-written:
// __syncthreads() would drain the loads in flight, guide 5.3), and the per-wave minima meet in LDS, one wavefront
// writing the query's key an iteration later.  Same arithmetic in the same order as lut_kernel /
// scan_k1_kernel (ip_sse_order, adc_sum, the packed (distance, scan position) key), so the same bits.
//
// Buffers: tables x2, plan and query x3 (item j's plan is still read by the slower wavefronts' extra passes while a
// faster one already stores item j+2's).  ~37 KB of LDS and ~120 VGPRs per 512-thread workgroup: two per CU.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>

namespace ivfhnsw_gpu_impl {

namespace {

constexpr int PT = 512;  // threads per workgroup: two halves of 256 code words
constexpr int PSEG = 64; // plan segments per query (nprobe <= 64)

__device__ __forceinline__ void pipe_barrier()
{
    // LDS traffic drained, global loads left in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ unsigned long long pipe_wave_min(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

template <int CS, int U>
struct PipeRegs {
    uint32_t w[U][CS / 4];
    uint32_t nb[U]; // norm codes (separate registers: packing them would wait for each load where it is issued)
    uint32_t sg;    // byte u: plan segment of slot u
    uint32_t ok; // bit u: slot u holds a code
};
static_assert(PSEG == 64, "segment starts one per lane; the index travels in a byte");

// request the codes at plan positions base + u * PT + tid (u < U) of one query.  The segment of a position is the
// number of segment starts at or below it: the starts sit one per lane in a register (one LDS read), and a scalar
// loop over them costs a v_readlane and U compares -- no chain of dependent LDS reads per slot.
template <int CS, int U>
__device__ __forceinline__ void pipe_request(PipeRegs<CS, U> &r, const uint8_t *__restrict__ codes,
                                             const uint8_t *__restrict__ norm_codes, const Seg *s_seg,
                                             const uint32_t *s_lpos, uint32_t nseg, uint32_t total, uint32_t base,
                                             int tid)
{
    static_assert(U <= 4, "one byte per slot");
    r.sg = 0;
    const uint32_t starts = s_lpos[tid & 63]; // PSEG == 64: lane l holds the start of segment l
    uint32_t s[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        s[u] = 0;
    for (uint32_t l = 1; l < nseg; l++) {
        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)starts, (int)l);
#pragma unroll
        for (int u = 0; u < U; u++)
            s[u] += base + u * PT + tid >= st ? 1u : 0u;
    }
    // Every slot loads, unconditionally (idle lanes and idle slots read code 0; the shard holds at least one): a load
    // under a branch leaves its byte behind a phi, hipcc zero-extends it right behind the join -- and waits for it there.
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t p = base + u * PT + tid;
        const uint32_t su = min(s[u], nseg > 0 ? nseg - 1u : 0u); // (positions beyond the plan counted every start)
        const uint32_t gi = p < total ? s_seg[su].start + (p - s_lpos[su]) : 0u;
        load_code_words<CS>(codes, gi, r.w[u]);
        r.nb[u] = norm_codes[gi];
        r.sg |= su << (8 * u);
    }
}

// "These registers are needed NOW": an empty asm that names them makes hipcc place the wait for their loads here, at the
// top of the iteration, on every path.  Without it the wait sits inside the conditional scoring code, the path around
// that code reaches the next requests with the loads formally pending, and every temporary that reuses one of their
// registers gets a s_waitcnt in front -- which serialised the four slots' loads and stalled wave 0 on the plan it had
// just requested.
template <int CS, int U>
__device__ __forceinline__ void pipe_touch(const PipeRegs<CS, U> &r)
{
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
        for (int i = 0; i < CS / 4; i++)
            asm volatile("" ::"v"(r.w[u][i]));
        asm volatile("" ::"v"(r.nb[u]));
    }
}

// (scan position and centroid term come from the plan again: the item's plan buffer outlives its scoring)
template <int CS, int U>
__device__ __forceinline__ unsigned long long pipe_score(const PipeRegs<CS, U> &r, const float *s_lut,
                                                         const float *s_norm, const Seg *s_seg, const uint32_t *s_lpos,
                                                         uint32_t total, uint32_t base, int tid,
                                                         unsigned long long best)
{
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (base + u * PT + tid < total) { // the slots pipe_request filled
            const uint32_t s = (r.sg >> (8 * u)) & 0xffu;
            const Seg sg = s_seg[s];
            const uint32_t vp = sg.vpos + (base + u * PT + tid - s_lpos[s]);
            const float sum = adc_sum<CS>(s_lut, r.w[u]);
            const float tt = __fadd_rn(sg.cterm, s_norm[r.nb[u]]);
            const float dist = __fsub_rn(tt, __fmul_rn(2.0f, sum));
            if (dist < FLT_MAX) { // also rejects NaN, as 'dist < distances[0]' does
                const unsigned long long key = ((unsigned long long)f32_orderable(__fadd_rn(dist, 0.0f)) << 32) | vp;
                best = key < best ? key : best;
            }
        }
    }
    return best;
}

// what a thread carries of a query's inputs between "requested" and "in LDS": wave 0 a plan segment (as plain
// words -- a struct member here gets promoted to an LDS stack slot, with a wait right behind the load), its position and
// the following wavefronts a query component each; the header travels in scalar registers
struct PipeIn {
    uint4 sg;
    uint32_t lp; // wave 0: plan position; waves 1..: a query component's bits
    uint2 h;
};

template <int D>
__device__ __forceinline__ void pipe_fetch(PipeIn &in, const float *__restrict__ xq, const Seg *__restrict__ segs,
                                           const uint32_t *__restrict__ lpos, const PlanHdr *__restrict__ hdr,
                                           int max_seg, int q, int tid)
{
    in.h = *reinterpret_cast<const uint2 *>(hdr + q); // uniform address: a scalar load
    if (tid < 64) { // wave 0: the plan
        const size_t i = (size_t)q * max_seg + min(tid, max_seg - 1);
        in.sg = *reinterpret_cast<const uint4 *>(segs + i);
        in.lp = lpos[i];
    } else if (tid < 64 + D) { // waves 1..: the query
        in.lp = __float_as_uint((xq + (size_t)q * D)[(uint32_t)(tid - 64)]);
    }
}

template <int D>
__device__ __forceinline__ void pipe_store(const PipeIn &in, bool live, Seg *s_seg, uint32_t *s_lpos, uint32_t *s_nseg,
                                           float *s_x, int tid)
{
    if (tid < 64) {
        const uint32_t ns = live ? min(in.h.x, (uint32_t)PSEG) : 0u; // PlanHdr: nseg, total
        const uint32_t tot = live ? in.h.y : 0u;
        *reinterpret_cast<uint4 *>(s_seg + tid) = in.sg;
        s_lpos[tid] = (uint32_t)tid == ns ? tot : in.lp; // [nseg] = the end of the last segment (0 for an empty plan)
        if (tid == 0) {
            *s_nseg = ns;
            if (ns == PSEG)
                s_lpos[PSEG] = tot;
        }
    } else if (tid < 64 + D) {
        s_x[tid - 64] = __uint_as_float(in.lp);
    }
}

// ML = how many of a thread's CS / 2 code-book rows live in LDS instead of registers: at d = 128 the whole share
// (64 registers) leaves the allocator no room beside the 21 of the requested codes -- it spills into the loop --
// and a quarter of the book (32 KB per workgroup, two workgroups per CU still fit) costs 8 conflict-free LDS reads per item.
// STAMPS: a diagnostic instantiation (IVFHNSW_PIPE_STAMPS=1, never timed): wave 1 sums s_memtime differences per phase
// -- 0 wait for the codes, 1 scoring, 2 inputs, 3 request, 4 table, 5 barrier -- into stamps[0..5], items in [6].
template <int CS, int DSUB, int U, int ML, bool STAMPS>
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_pipe_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ norm_codes, const float *__restrict__ xq,
    const float *__restrict__ cb, const float *__restrict__ norm_table, const Seg *__restrict__ segs,
    const uint32_t *__restrict__ lpos, const PlanHdr *__restrict__ hdr, int max_seg, int nq,
    unsigned long long *__restrict__ keys, unsigned long long *__restrict__ stamps)
{
    constexpr int D = CS * DSUB;
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = 0;
#define STAMP(i)                                                    \
    if (STAMPS) {                                                   \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += t_ - st_t;                                     \
        st_t = t_;                                                  \
    }
    constexpr int MH = CS / 2; // sub-quantizers per half
    constexpr int MR = MH - ML; // ... of which in registers
    constexpr int D2 = DSUB / 2;
    __shared__ float2 s_cb[ML > 0 ? 2 * ML * D2 * 256 : 1]; // [half][ml][i / 2][c]: lane c reads consecutive float2
    __shared__ __attribute__((aligned(16))) float s_lut[2][CS * 256];
    __shared__ float s_norm[256];
    __shared__ __attribute__((aligned(16))) Seg s_seg[3][PSEG];
    __shared__ uint32_t s_lpos[3][PSEG + 1];
    __shared__ uint32_t s_nseg[3];
    __shared__ __attribute__((aligned(16))) float s_x[3][D];
    __shared__ unsigned long long s_red[2][PT / 64];

    const int tid = threadIdx.x;
    const int c = tid & 255;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int G = (int)gridDim.x;
    const int g = (int)blockIdx.x;
    const int n = (nq - g + G - 1) / G; // this workgroup's items: queries g, g + G, ...
    auto q_of = [&](int j) { return min(g + j * G, nq - 1); }; // items beyond the last re-read the last query, dead

    // this thread's share of the code book, resident for the whole launch
    float row[MR][DSUB];
#pragma unroll
    for (int m = 0; m < MR; m++) {
        const float *src = cb + ((size_t)(half * MH + m) * 256 + c) * DSUB;
#pragma unroll
        for (int i = 0; i < DSUB; i += 2) {
            const float2 v = *reinterpret_cast<const float2 *>(src + i);
            row[m][i] = v.x, row[m][i + 1] = v.y;
        }
    }
#pragma unroll
    for (int ml = 0; ml < ML; ml++) {
        const float *src = cb + ((size_t)(half * MH + MR + ml) * 256 + c) * DSUB;
#pragma unroll
        for (int i2 = 0; i2 < D2; i2++)
            s_cb[((half * ML + ml) * D2 + i2) * 256 + c] = *reinterpret_cast<const float2 *>(src + 2 * i2);
    }
    if (tid < 256)
        s_norm[tid] = norm_table[tid];

    // tab[m][c] = <x_m, centroid[m][c]> for this thread's m, into table buffer tb (IndexIVF_HNSW.cpp:262)
    auto build_table = [&](int tb, int xb) {
        const float *x = s_x[xb] + half * (MH * DSUB);
#pragma unroll
        for (int m = 0; m < MH; m++) {
            float xs[DSUB];
#pragma unroll
            for (int i = 0; i < DSUB; i += 2) {
                const float2 v = *reinterpret_cast<const float2 *>(x + m * DSUB + i);
                xs[i] = v.x, xs[i + 1] = v.y;
            }
            float r;
            if (m < MR) {
                r = ip_sse_order<DSUB>(xs, row[m < MR ? m : 0], DSUB);
            } else {
                float rl[DSUB];
#pragma unroll
                for (int i2 = 0; i2 < D2; i2++) {
                    const float2 v = s_cb[((half * ML + (m - MR)) * D2 + i2) * 256 + c];
                    rl[2 * i2] = v.x, rl[2 * i2 + 1] = v.y;
                }
                r = ip_sse_order<DSUB>(xs, rl, DSUB);
            }
            s_lut[tb][(half * MH + m) * 256 + c] = r;
        }
    };

    auto flush_key = [&](int jj) { // by one wavefront, after the barrier that follows item jj's scoring
        const int lane = tid & 63;
        unsigned long long v = lane < PT / 64 ? s_red[jj & 1][lane] : ~0ull;
#pragma unroll
        for (int off = PT / 128; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(v, off, 64);
            v = o < v ? o : v;
        }
        if (lane == 0)
            keys[g + jj * G] = v;
    };

    // ---- prologue: items 0 and 1 in LDS, table and codes of item 0, inputs of item 2 requested
    PipeIn in;
    PipeRegs<CS, U> regs;
    pipe_fetch<D>(in, xq, segs, lpos, hdr, max_seg, q_of(0), tid);
    pipe_store<D>(in, 0 < n, s_seg[0], s_lpos[0], &s_nseg[0], s_x[0], tid);
    pipe_fetch<D>(in, xq, segs, lpos, hdr, max_seg, q_of(1), tid);
    pipe_store<D>(in, 1 < n, s_seg[1], s_lpos[1], &s_nseg[1], s_x[1], tid);
    __syncthreads();
    {
        const uint32_t ns0 = s_nseg[0], tot0 = s_lpos[0][ns0];
        if (tot0 > 0)
            build_table(0, 0);
        pipe_request<CS, U>(regs, codes, norm_codes, s_seg[0], s_lpos[0], ns0, tot0, 0u, tid);
    }
    pipe_fetch<D>(in, xq, segs, lpos, hdr, max_seg, q_of(2), tid);
    __syncthreads();

    int b0 = 0, b1 = 1, b2 = 2; // plan / query buffers of items j, j + 1, j + 2
    if (STAMPS)
        st_t = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n; j++) {
        pipe_touch<CS, U>(regs);
        asm volatile("" ::"v"(in.sg.x), "v"(in.sg.y), "v"(in.sg.z), "v"(in.sg.w), "v"(in.lp));
        STAMP(0);
        // ---- 1. score item j: first pass from the registers, the rest of a long plan in place
        const uint32_t ns = s_nseg[b0], total = s_lpos[b0][ns];
        unsigned long long best = kKeyInit;
        if (total > 0) { // block-uniform
            best =
                pipe_score<CS, U>(regs, s_lut[j & 1], s_norm, s_seg[b0], s_lpos[b0], total, 0u, tid, kKeyInit);
            for (uint32_t base = U * PT; base < total; base += U * PT) {
                pipe_request<CS, U>(regs, codes, norm_codes, s_seg[b0], s_lpos[b0], ns, total, base, tid);
                pipe_touch<CS, U>(regs);
                best = pipe_score<CS, U>(regs, s_lut[j & 1], s_norm, s_seg[b0], s_lpos[b0], total, base, tid, best);
            }
            best = pipe_wave_min(best);
        }
        // the wavefronts' minima meet in LDS; wave 3 writes the query's key an iteration later (no atomic, and no
        // store whose completion a later wait of THIS iteration would sit on)
        if ((tid & 63) == 0)
            s_red[j & 1][tid >> 6] = best;
        if (j > 0 && (tid >> 6) == 3)
            flush_key(j - 1);
        STAMP(1);
        // ---- 2. inputs of item j + 2 into LDS (requested an iteration ago), those of item j + 3 requested.  (Before
        // the codes: both branches of pipe_fetch write in.lp, hipcc guards the second with a full wait, and wave 0
        // must have nothing in flight when it gets there.)
        pipe_store<D>(in, j + 2 < n, s_seg[b2], s_lpos[b2], &s_nseg[b2], s_x[b2], tid);
        pipe_fetch<D>(in, xq, segs, lpos, hdr, max_seg, q_of(j + 3), tid);
        STAMP(2);
        // ---- 3. codes of item j + 1 requested: in flight behind the table build and the barrier
        const uint32_t ns1 = s_nseg[b1], tot1 = s_lpos[b1][ns1];
        pipe_request<CS, U>(regs, codes, norm_codes, s_seg[b1], s_lpos[b1], ns1, tot1, 0u, tid);
        STAMP(3);
        // ---- 4. table of item j + 1
        if (tot1 > 0)
            build_table((j + 1) & 1, b1);
        STAMP(4);
        pipe_barrier();
        STAMP(5);
        const int t = b0;
        b0 = b1, b1 = b2, b2 = t;
    }
    if ((tid >> 6) == 3)
        flush_key(n - 1);
    if (STAMPS && tid == 64) {
        for (int i = 0; i < 6; i++)
            atomicAdd(&stamps[i], st_acc[i]);
        atomicAdd(&stamps[6], (unsigned long long)n);
    }
#undef STAMP
}

} // namespace

// Used for list shards (shard_world >= 8) on the common shapes, whole batches, plans of at most 64 segments;
// IVFHNSW_SCAN_PIPE = 1 forces it for a single shard too, 0 turns it off.
bool scan_pipe_supported(const IvfTables &t, int max_seg, int nq, int nsplit, bool has_codes, bool forced)
{
    static const int knob = [] {
        const char *e = getenv("IVFHNSW_SCAN_PIPE");
        return (e && *e) ? (atoi(e) != 0 ? 1 : 0) : -1;
    }();
    (void)nsplit; // a workgroup takes whole queries: small batches simply launch fewer workgroups
    if (knob == 0 || max_seg > PSEG || !has_codes || (knob != 1 && !forced && nq < 1024))
        return false;
    const bool shape = (t.M == 16 && (t.dsub == 8 || t.dsub == 6)) || (t.M == 8 && (t.dsub == 16 || t.dsub == 12));
    if (!shape)
        return false;
    // measured per rank (tools/rank_emulation.py, 1B corpus): at 2 and 4 shards lut_kernel + scan_k1_kernel win (0.48 vs 0.55,
    // 0.59 vs 0.63 ms: their eight workgroups per CU keep the LDS busier than two pipelined ones), at 8 the table traffic
    // has grown to where this form draws level (0.735 vs 0.745) and saves the 1.3-GB table buffer
    return knob == 1 || forced || t.shard_world >= 8;
}

hipError_t launch_scan_pipe(hipStream_t s, const IvfTables &t, const float *xq, const Seg *segs, const uint32_t *lpos,
                            const PlanHdr *hdr, int max_seg, int nq, uint64_t *keys)
{
    if (nq == 0)
        return hipSuccess;
    static const int resident = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 2 * cus; // two 512-thread workgroups per CU at 4 waves per SIMD
    }();
    const dim3 grid((unsigned)(nq < resident ? nq : resident)), block(PT);
    auto *k64 = reinterpret_cast<unsigned long long *>(keys);
    static const bool want_stamps = [] {
        const char *e = getenv("IVFHNSW_PIPE_STAMPS");
        return e && *e && atoi(e) != 0;
    }();
    static unsigned long long *d_stamps = nullptr;
    if (want_stamps) {
        if (!d_stamps && hipMalloc(&d_stamps, 8 * sizeof(unsigned long long)) != hipSuccess)
            return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_stamps, 0, 8 * sizeof(unsigned long long), s);
    }
#define IVFHNSW_PIPE(CS, DSUB, ML)                                                                               \
    do {                                                                                                         \
        if (want_stamps)                                                                                         \
            hipLaunchKernelGGL((scan_pipe_kernel<CS, DSUB, 4, ML, true>), grid, block, 0, s, t.codes,            \
                               t.norm_codes, xq, t.pq_centroids, t.norm_table, segs, lpos, hdr, max_seg, nq, k64, \
                               d_stamps);                                                                        \
        else                                                                                                     \
            hipLaunchKernelGGL((scan_pipe_kernel<CS, DSUB, 4, ML, false>), grid, block, 0, s, t.codes,           \
                               t.norm_codes, xq, t.pq_centroids, t.norm_table, segs, lpos, hdr, max_seg, nq, k64, \
                               nullptr);                                                                         \
    } while (0)
    if (t.M == 16 && t.dsub == 8)
        IVFHNSW_PIPE(16, 8, 2);
    else if (t.M == 16 && t.dsub == 6)
        IVFHNSW_PIPE(16, 6, 0);
    else if (t.M == 8 && t.dsub == 16)
        IVFHNSW_PIPE(8, 16, 1);
    else if (t.M == 8 && t.dsub == 12)
        IVFHNSW_PIPE(8, 12, 0);
    else
        return hipErrorInvalidValue;
#undef IVFHNSW_PIPE
    if (want_stamps) {
        unsigned long long hst[8] = {};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hst, d_stamps, sizeof(hst), hipMemcpyDeviceToHost);
        const double it = hst[6] ? (double)hst[6] : 1.0;
        fprintf(stderr, "[pipe stamps] items/wg-sum %llu; per item (100 MHz ticks): wait %.1f score %.1f inputs %.1f request %.1f table %.1f barrier %.1f\n",
                hst[6], hst[0] / it, hst[1] / it, hst[2] / it, hst[3] / it, hst[4] / it, hst[5] / it);
    }
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
