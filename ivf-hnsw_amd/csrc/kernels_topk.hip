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
                                                        unsigned long long *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float s_lut[CS * 256];
    __shared__ float s_norm[256];
    __shared__ __attribute__((aligned(16))) Seg s_seg[TK_SEGCAP];
    __shared__ uint32_t s_lpos[TK_SEGCAP + 1];
    __shared__ unsigned long long s_buf[TK_N];
    __shared__ uint32_t s_ncand;
    __shared__ unsigned long long s_T;

    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const PlanHdr h = hdr[q];
    if (h.total == 0)
        return; // keys were reset by the plan kernel
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
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                const uint32_t p = base + u * 256 + tid;
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
                        const unsigned long long key =
                            ((unsigned long long)f32_orderable(__fadd_rn(dist, 0.0f)) << 32) | (sg.vpos + off);
                        if (key < T) {
                            const uint32_t slot = atomicAdd(&s_ncand, 1u);
                            s_buf[TK_KCAP + slot] = key;
                        }
                    }
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
}

hipError_t launch_scan_topk(hipStream_t s, const IvfTables &t, const float *luts, const Seg *segs,
                            const uint32_t *lpos, const PlanHdr *hdr, int max_seg, int nq, int k, uint64_t *keys)
{
    if (k < 1 || k > TK_KCAP)
        return hipErrorInvalidValue;
    dim3 grid((unsigned)nq), block(256);
    auto *k64 = reinterpret_cast<unsigned long long *>(keys);
#define IVFHNSW_TOPK(CS)                                                                                              \
    hipLaunchKernelGGL((scan_topk_kernel<CS>), grid, block, 0, s, t.codes, t.norm_codes, luts, t.norm_table, segs, lpos, \
                       hdr, max_seg, k, k64)
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
