// ADC list scan for k > 1: same arithmetic as scan_k1_kernel (kernels_search.hip), but the k smallest
// (distance, scan position) keys are kept per query.
//
// faiss's max-heap admits a code iff dist < current k-th best (IndexIVF_HNSW.cpp:285-288); the final
// CONTENT of that heap is the k smallest keys with ties at the boundary resolved for the earlier scanned
// code, which is exactly "the k smallest (dist, vpos) keys".  (The heap-array ORDER the reference leaves
// them in is not reproduced: results come out ascending.  k = 1, the only value the reference's presets
// use, has no such freedom.)
//
// One 256-thread workgroup per query.  Threads push keys below the running threshold T into an LDS
// buffer; when it fills, the buffer and the current top are bitonic-sorted together and T tightens, so
// after the first few hundred codes almost nothing passes the filter.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>

namespace ivfhnsw_gpu_impl {

constexpr int TK_N = 2048;      // sort buffer entries (top | candidates)
constexpr int TK_KCAP = 1024;   // max k
constexpr int TK_SEGCAP = 512;
constexpr int TK_U = 2;

__device__ __forceinline__ void bitonic_sort_2048(unsigned long long *buf, int tid)
{
    for (int size = 2; size <= TK_N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < TK_N / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = buf[lo], b = buf[hi];
                if ((a > b) == up) {
                    buf[lo] = b;
                    buf[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

template <int CS>
__global__ __launch_bounds__(256) void scan_topk_kernel(const uint8_t *__restrict__ codes,
                                                        const uint8_t *__restrict__ norm_codes,
                                                        const float *__restrict__ luts,
                                                        const float *__restrict__ norm_table,
                                                        const Seg *__restrict__ segs, const uint32_t *__restrict__ lpos,
                                                        const PlanHdr *__restrict__ hdr, int max_seg, int k,
                                                        unsigned long long *__restrict__ keys,
                                                        unsigned long long *__restrict__ stream,
                                                        uint32_t *__restrict__ stream_len, uint32_t stream_cap)
{
    __shared__ __attribute__((aligned(16))) float s_lut[CS * 256];
    __shared__ float s_norm[256];
    __shared__ __attribute__((aligned(16))) Seg s_seg[TK_SEGCAP];
    __shared__ uint32_t s_lpos[TK_SEGCAP + 1];
    __shared__ unsigned long long s_buf[TK_N];
    __shared__ uint32_t s_ncand;
    __shared__ unsigned long long s_T;
    __shared__ uint32_t s_wcnt[TK_U][4]; // candidates per (unroll step, wave) of the current iteration
    __shared__ uint32_t s_slen;          // length of this query's candidate stream so far

    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const PlanHdr h = hdr[q];
    if (h.total == 0) {
        if (stream_len && tid == 0)
            stream_len[q] = 0;
        return; // keys were reset by the plan kernel
    }
    {
        const float4 *src = reinterpret_cast<const float4 *>(luts + (size_t)q * CS * 256);
        float4 *dst = reinterpret_cast<float4 *>(s_lut);
#pragma unroll
        for (int i = 0; i < CS * 64 / 256; i++)
            dst[i * 256 + tid] = src[i * 256 + tid];
        s_norm[tid] = norm_table[tid];
        for (int i = tid; i < TK_N; i += 256)
            s_buf[i] = ~0ull;
        if (tid == 0) {
            s_ncand = 0;
            s_slen = 0;
            s_T = kKeyInit;
        }
    }
    const Seg *sq = segs + (size_t)q * max_seg;
    const uint32_t *lq = lpos + (size_t)q * max_seg;

    auto flush = [&]() {
        // all candidates are in s_buf[TK_KCAP .. TK_KCAP + ncand); sort everything, keep the k smallest
        __syncthreads();
        bitonic_sort_2048(s_buf, tid);
        for (int i = tid; i < TK_N; i += 256)
            if (i >= k)
                s_buf[i] = ~0ull;
        if (tid == 0) {
            s_ncand = 0;
            const unsigned long long kth = s_buf[k - 1];
            s_T = kth < kKeyInit ? kth : kKeyInit;
        }
        __syncthreads();
    };

    for (uint32_t cs = 0; cs < h.nseg; cs += TK_SEGCAP) {
        const uint32_t cn = min((uint32_t)TK_SEGCAP, h.nseg - cs);
        __syncthreads();
        for (uint32_t i = tid; i < cn; i += 256) {
            s_seg[i] = sq[cs + i];
            s_lpos[i] = lq[cs + i];
        }
        const uint32_t ch = (cs + cn == h.nseg) ? h.total : lq[cs + cn];
        if (tid == 0)
            s_lpos[cn] = ch;
        __syncthreads();
        const uint32_t cl = s_lpos[0];
        uint32_t s = 0;
        for (uint32_t base = cl; base < ch; base += 256 * TK_U) {
            // read the fill level before anyone can add to it again, so the decision is uniform
            const uint32_t fill = s_ncand;
            __syncthreads();
            if (fill > TK_N - TK_KCAP - 256 * TK_U)
                flush();
            const unsigned long long T = s_T;
            unsigned long long key[TK_U];
            bool pass[TK_U];
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                const uint32_t p = base + u * 256 + tid;
                pass[u] = false;
                key[u] = 0;
                if (p < ch) {
                    uint32_t a = s, b = cn - 1;
                    while (a < b) {
                        const uint32_t mid = (a + b) >> 1;
                        if (s_lpos[mid + 1] > p)
                            b = mid;
                        else
                            a = mid + 1;
                    }
                    s = a;
                    const Seg sg = s_seg[s];
                    const uint32_t off = p - s_lpos[s];
                    const uint32_t gi = sg.start + off;
                    uint32_t w[CS / 4];
                    load_code_words<CS>(codes, gi, w);
                    const uint32_t nb = norm_codes[gi];
                    const float sum = adc_sum<CS>(s_lut, w);
                    const float tt = __fadd_rn(sg.cterm, s_norm[nb]);
                    const float dist = __fsub_rn(tt, __fmul_rn(2.0f, sum));
                    if (dist < FLT_MAX) {
                        key[u] = ((unsigned long long)f32_orderable(__fadd_rn(dist, 0.0f)) << 32) | (sg.vpos + off);
                        pass[u] = key[u] < T;
                    }
                }
            }
            // Append the candidates IN SCAN ORDER (position = unroll step, then wave, then lane): the buffer is
            // then a stream a sequential consumer can replay (heap-order output, below).
            unsigned long long bal[TK_U];
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                bal[u] = __ballot(pass[u]);
                if ((tid & 63) == 0)
                    s_wcnt[u][tid >> 6] = (uint32_t)__popcll(bal[u]);
            }
            __syncthreads();
            {
                uint32_t before = 0, total = 0;
#pragma unroll
                for (int u = 0; u < TK_U; u++)
#pragma unroll
                    for (int w2 = 0; w2 < 4; w2++) {
                        const uint32_t c = s_wcnt[u][w2];
                        total += c;
                        (void)before;
                    }
                const uint32_t nc0 = s_ncand, sl0 = s_slen;
#pragma unroll
                for (int u = 0; u < TK_U; u++) {
                    uint32_t off = 0;
#pragma unroll
                    for (int u2 = 0; u2 < TK_U; u2++)
#pragma unroll
                        for (int w2 = 0; w2 < 4; w2++)
                            if (u2 < u || (u2 == u && w2 < (tid >> 6)))
                                off += s_wcnt[u2][w2];
                    if (pass[u]) {
                        const uint32_t pos = off + (uint32_t)__popcll(bal[u] & ((1ull << (tid & 63)) - 1ull));
                        s_buf[TK_KCAP + nc0 + pos] = key[u];
                        if (stream && sl0 + pos < stream_cap)
                            stream[(size_t)q * stream_cap + sl0 + pos] = key[u];
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    s_ncand = nc0 + total;
                    s_slen = sl0 + total;
                }
            }
            __syncthreads();
        }
    }
    flush();
    for (int j = tid; j < k; j += 256) {
        const unsigned long long v = s_buf[j];
        keys[(size_t)q * k + j] = v < kKeyInit ? v : kKeyInit;
    }
    if (stream_len && tid == 0)
        stream_len[q] = s_slen; // > stream_cap: the stream was truncated (the consumer reports it)
}

// ---------------------------------------------------------------------------------------------
// faiss heap-array order for k > 1 (IndexIVF_HNSW.cpp:265,285-288).  The reference pushes a code iff
// dist < distances[0] at that moment; a code that fails the test leaves the heap untouched.  Replaying the same
// pop/push over ANY superset of the admitted codes, in scan order, therefore reproduces the heap exactly -- and
// the candidate stream above is such a superset (its filter threshold is never below the heap's current
// maximum).  One thread per query; faiss Heap.h semantics (1-based binary max-heap on values only).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void heap_replace_top(int k, float *val, long long *ids, float nv, long long nid)
{
    // maxheap_pop followed by maxheap_push, as the reference calls them
    float *v = val - 1;
    long long *id = ids - 1;
    {
        const float last = v[k];
        int hole = 1;
        for (;;) {
            const int l = hole * 2, r = l + 1;
            if (l > k)
                break;
            const int big = (r == k + 1 || v[l] > v[r]) ? l : r;
            if (last > v[big])
                break;
            v[hole] = v[big];
            id[hole] = id[big];
            hole = big;
        }
        v[hole] = v[k];
        id[hole] = id[k];
    }
    {
        int hole = k;
        while (hole > 1) {
            const int parent = hole / 2;
            if (!(nv > v[parent]))
                break;
            v[hole] = v[parent];
            id[hole] = id[parent];
            hole = parent;
        }
        v[hole] = nv;
        id[hole] = nid;
    }
}

__global__ void heap_replay_kernel(IvfTables t, const Seg *__restrict__ segs, const PlanHdr *__restrict__ hdr, int max_seg,
                                   const unsigned long long *__restrict__ stream, const uint32_t *__restrict__ stream_len,
                                   uint32_t stream_cap, int nq, int k, float *__restrict__ dist,
                                   long long *__restrict__ labels, uint32_t *__restrict__ status)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq)
        return;
    float *val = dist + (size_t)q * k;
    long long *ids = labels + (size_t)q * k;
    for (int j = 0; j < k; j++) { // maxheap_heapify
        val[j] = FLT_MAX;
        ids[j] = -1;
    }
    const uint32_t len = stream_len[q];
    if (len > stream_cap) {
        atomicOr(status, kStatusTopkStreamOverflow);
        return;
    }
    const Seg *sq = segs + (size_t)q * max_seg;
    const uint32_t nseg = hdr[q].nseg;
    const unsigned long long *st = stream + (size_t)q * stream_cap;
    for (uint32_t i = 0; i < len; i++) {
        const unsigned long long key = st[i];
        const float d = orderable_f32((uint32_t)(key >> 32));
        if (!(d < val[0]))
            continue;
        // label of scan position vpos (segments ascend in vpos)
        const uint32_t vpos = (uint32_t)key;
        uint32_t a = 0, b = nseg - 1;
        while (a < b) {
            const uint32_t mid = (a + b + 1) >> 1;
            if (sq[mid].vpos <= vpos)
                a = mid;
            else
                b = mid - 1;
        }
        const Seg sg = sq[a];
        heap_replace_top(k, val, ids, d, (long long)t.ids[sg.start + (vpos - sg.vpos)]);
    }
}

hipError_t launch_heap_replay(hipStream_t s, const IvfTables &t, const Seg *segs, const PlanHdr *hdr, int max_seg,
                              const uint64_t *stream, const uint32_t *stream_len, uint32_t stream_cap, int nq, int k,
                              float *dist, int64_t *labels, uint32_t *status)
{
    if (nq == 0)
        return hipSuccess;
    hipLaunchKernelGGL(heap_replay_kernel, dim3((nq + 63) / 64), dim3(64), 0, s, t, segs, hdr, max_seg,
                       reinterpret_cast<const unsigned long long *>(stream), stream_len, stream_cap, nq, k, dist,
                       reinterpret_cast<long long *>(labels), status);
    return hipGetLastError();
}

hipError_t launch_scan_topk(hipStream_t s, const IvfTables &t, const float *luts, const Seg *segs,
                            const uint32_t *lpos, const PlanHdr *hdr, int max_seg, int nq, int k, uint64_t *keys,
                            uint64_t *stream, uint32_t *stream_len, uint32_t stream_cap)
{
    if (k < 1 || k > TK_KCAP)
        return hipErrorInvalidValue;
    dim3 grid((unsigned)nq), block(256);
    auto *k64 = reinterpret_cast<unsigned long long *>(keys);
#define IVFHNSW_TOPK(CS)                                                                                              \
    hipLaunchKernelGGL((scan_topk_kernel<CS>), grid, block, 0, s, t.codes, t.norm_codes, luts, t.norm_table, segs, lpos, \
                       hdr, max_seg, k, k64, reinterpret_cast<unsigned long long *>(stream), stream_len, stream_cap)
    switch (t.M) {
    case 4: IVFHNSW_TOPK(4); break;
    case 8: IVFHNSW_TOPK(8); break;
    case 16: IVFHNSW_TOPK(16); break;
    case 32: IVFHNSW_TOPK(32); break;
    default: return hipErrorInvalidValue;
    }
#undef IVFHNSW_TOPK
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
