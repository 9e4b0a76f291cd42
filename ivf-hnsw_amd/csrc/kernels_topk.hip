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
constexpr int TK_U = 4;

// ascending bitonic sort of buf[0 .. n), n a power of two <= TK_N (entries beyond the live ones hold ~0)
__device__ __forceinline__ void bitonic_sort_n(unsigned long long *buf, int n, int tid)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n / 2; i += 256) {
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
                                                        uint32_t *__restrict__ stream_len, uint32_t stream_cap,
                                                        int cs_rt)
{
    // CS == 0: run-time code size cs_rt, table in dynamic LDS (see scan_k1_kernel)
    __shared__ __attribute__((aligned(16))) float s_lut_fixed[(CS > 0 ? CS : 1) * 256];
    extern __shared__ __attribute__((aligned(16))) float s_lut_dyn[];
    float *s_lut = CS > 0 ? s_lut_fixed : s_lut_dyn;
    const int csz = CS > 0 ? CS : cs_rt;
    __shared__ float s_norm[256];
    __shared__ __attribute__((aligned(16))) Seg s_seg[TK_SEGCAP];
    __shared__ uint32_t s_lpos[TK_SEGCAP + 1];
    __shared__ unsigned long long s_buf[TK_N];
    __shared__ unsigned long long s_T;
    __shared__ uint32_t s_wcnt[2][TK_U][4]; // candidates per (unroll step, wave), double buffered by iteration

    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const PlanHdr h = hdr[q];
    if (h.total == 0) {
        if (stream_len && tid == 0)
            stream_len[q] = 0;
        return; // keys were reset by the plan kernel
    }
    {
        const float4 *src = reinterpret_cast<const float4 *>(luts + (size_t)q * csz * 256);
        float4 *dst = reinterpret_cast<float4 *>(s_lut);
        if constexpr (CS > 0) {
#pragma unroll
            for (int i = 0; i < CS * 64 / 256; i++)
                dst[i * 256 + tid] = src[i * 256 + tid];
        } else {
            for (int i = tid; i < csz * 64; i += 256)
                dst[i] = src[i];
        }
        s_norm[tid] = norm_table[tid];
        for (int i = tid; i < TK_N; i += 256)
            s_buf[i] = ~0ull;
        if (tid == 0)
            s_T = kKeyInit;
    }
    // fill level of the sort buffer and length of the candidate stream: every thread keeps the same copy (they
    // all add the same per-wave counts), so neither needs a broadcast
    uint32_t ncand = 0, slen = 0;
    unsigned long long T = kKeyInit;
    uint32_t iter = 0;
    const Seg *sq = segs + (size_t)q * max_seg;
    const uint32_t *lq = lpos + (size_t)q * max_seg;

    auto flush = [&]() {
        // all candidates are in s_buf[k .. k + ncand); sort everything, keep the k smallest
        __syncthreads();
        // only the live prefix is sorted: the first flush holds a whole iteration of candidates, later ones a few
        // dozen -- the 2048-entry sort was three quarters of this kernel's instructions
        int n_sort = 64;
        while (n_sort < k + (int)ncand)
            n_sort <<= 1;
        bitonic_sort_n(s_buf, n_sort, tid);
        for (int i = tid; i < n_sort; i += 256)
            if (i >= k)
                s_buf[i] = ~0ull;
        if (tid == 0) {
            const unsigned long long kth = s_buf[k - 1];
            s_T = kth < kKeyInit ? kth : kKeyInit;
        }
        __syncthreads();
        ncand = 0;
        T = s_T;
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
        // the segment this lane is inside, kept in registers (positions only grow)
        uint32_t s = 0, seg_lo = 0, seg_hi = 0, seg_start = 0, seg_vpos = 0;
        float seg_ct = 0.f;
        for (uint32_t base = cl; base < ch; iter++) {
            // no threshold yet (every code passes): take a quarter iteration and sort 512 entries instead of 2048
            if ((ncand && T == kKeyInit) || ncand + k + 256 * TK_U > TK_N)
                flush();
            const int u_now = T == kKeyInit ? 1 : TK_U;
            unsigned long long key[TK_U];
            bool pass[TK_U];
            CodeRegs<CS> w[TK_U];
            uint32_t nbv[TK_U], vp[TK_U];
            float ct[TK_U];
            bool ok[TK_U];
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                const uint32_t p = base + u * 256 + tid;
                ok[u] = u < u_now && p < ch;
                if (ok[u]) {
                    if (p >= seg_hi) {
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
                        seg_lo = s_lpos[s];
                        seg_hi = seg_lo + sg.len;
                        seg_start = sg.start;
                        seg_vpos = sg.vpos;
                        seg_ct = sg.cterm;
                    }
                    const uint32_t off = p - seg_lo;
                    const uint32_t gi = seg_start + off;
                    code_fetch<CS>(codes, gi, cs_rt, s_lut, w[u]);
                    nbv[u] = norm_codes[gi];
                    vp[u] = seg_vpos + off;
                    ct[u] = seg_ct;
                }
            }
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                pass[u] = false;
                key[u] = 0;
                if (ok[u]) {
                    const float sum = code_sum<CS>(s_lut, w[u]);
                    const float tt = __fadd_rn(ct[u], s_norm[nbv[u]]);
                    const float dist = __fsub_rn(tt, __fmul_rn(2.0f, sum));
                    if (dist < FLT_MAX) {
                        key[u] = ((unsigned long long)f32_orderable(__fadd_rn(dist, 0.0f)) << 32) | vp[u];
                        pass[u] = key[u] < T;
                    }
                }
            }
            // Append the candidates IN SCAN ORDER (position = unroll step, then wave, then lane): the buffer is
            // then a stream a sequential consumer can replay (heap-order output, below).  One barrier per
            // iteration: the per-wave counts are double buffered, the running totals live in registers.
            unsigned long long bal[TK_U];
            uint32_t(*wc)[4] = s_wcnt[iter & 1];
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                bal[u] = __ballot(pass[u]);
                if ((tid & 63) == 0)
                    wc[u][tid >> 6] = (uint32_t)__popcll(bal[u]);
            }
            __syncthreads();
            uint32_t total = 0, mine[TK_U];
#pragma unroll
            for (int u = 0; u < TK_U; u++) {
                mine[u] = total; // candidates of earlier unroll steps and earlier waves of this step
#pragma unroll
                for (int w2 = 0; w2 < 4; w2++) {
                    const uint32_t c = wc[u][w2];
                    if (w2 < (tid >> 6))
                        mine[u] += c;
                    total += c;
                }
            }
            if (total) {
#pragma unroll
                for (int u = 0; u < TK_U; u++)
                    if (pass[u]) {
                        const uint32_t pos = mine[u] + (uint32_t)__popcll(bal[u] & ((1ull << (tid & 63)) - 1ull));
                        s_buf[k + ncand + pos] = key[u];
                        if (stream && slen + pos < stream_cap)
                            stream[(size_t)q * stream_cap + slen + pos] = key[u];
                    }
                ncand += total;
                slen += total;
            }
            base += 256 * u_now;
        }
    }
    flush();
    for (int j = tid; j < k; j += 256) {
        const unsigned long long v = s_buf[j];
        keys[(size_t)q * k + j] = v < kKeyInit ? v : kKeyInit;
    }
    if (stream_len && tid == 0)
        stream_len[q] = slen; // > stream_cap: the stream was truncated (the consumer reports it)
}

// ---------------------------------------------------------------------------------------------
// faiss heap-array order for k > 1 (IndexIVF_HNSW.cpp:265,285-288).  The reference pushes a code iff
// dist < distances[0] at that moment; a code that fails the test leaves the heap untouched.  Replaying the same
// pop/push over ANY superset of the admitted codes, in scan order, therefore reproduces the heap exactly -- and
// the candidate stream above is such a superset (its filter threshold is never below the heap's current
// maximum).  One thread per query; faiss Heap.h semantics (1-based binary max-heap on values only).
// ---------------------------------------------------------------------------------------------
template <typename ID>
__device__ __forceinline__ void heap_replace_top(int k, float *val, ID *ids, float nv, ID nid)
{
    // maxheap_pop followed by maxheap_push, as the reference calls them
    float *v = val - 1;
    ID *id = ids - 1;
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

// One wavefront per query (four per workgroup).  The heap lives in LDS as (value, scan position); the stream is
// read 64 keys at a time by all lanes, keys that cannot pass (not below the heap's maximum at the start of the
// batch -- it only falls) are dropped with one ballot, the rest are replayed in order by lane 0; labels are
// resolved for the k survivors only, in parallel, at the end.
__global__ __launch_bounds__(256) void heap_replay_kernel(IvfTables t, const Seg *__restrict__ segs,
                                                          const PlanHdr *__restrict__ hdr, int max_seg,
                                                          const unsigned long long *__restrict__ stream,
                                                          const uint32_t *__restrict__ stream_len, uint32_t stream_cap,
                                                          int nq, int k, float *__restrict__ dist,
                                                          long long *__restrict__ labels, uint32_t *__restrict__ status,
                                                          long long *__restrict__ out_keys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nq)
        return; // whole wavefront; no workgroup barrier below
    float *val = reinterpret_cast<float *>(smem_h) + (size_t)wave * 2 * k;
    uint32_t *pos = reinterpret_cast<uint32_t *>(val + k);
    for (int j = lane; j < k; j += 64) { // maxheap_heapify
        val[j] = FLT_MAX;
        pos[j] = 0xffffffffu;
    }
    const uint32_t len = stream_len[q];
    float *out_d = dist + (size_t)q * k;
    long long *out_l = labels + (size_t)q * k;
    long long *out_k = out_keys ? out_keys + (size_t)q * k : nullptr;
    if (len > stream_cap) {
        if (lane == 0)
            atomicOr(status, kStatusTopkStreamOverflow);
        for (int j = lane; j < k; j += 64) {
            if (out_k) {
                out_k[j] = (long long)(kKeyInit ^ kSignFlip);
            } else {
                out_d[j] = FLT_MAX;
                out_l[j] = -1;
            }
        }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long *st = stream + (size_t)q * stream_cap;
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t i = base + lane;
        unsigned long long key = ~0ull;
        if (i < len)
            key = st[i];
        const float d = orderable_f32((uint32_t)(key >> 32));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // lane 0's heap writes of the previous batch
        const float top0 = val[0];
        unsigned long long m = __ballot(i < len && d < top0);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float dj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), b));
            const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, b);
            if (lane == 0 && dj < val[0])
                heap_replace_top<uint32_t>(k, val, pos, dj, pj);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (out_k) {
        // sharded search: the heap array as packed (distance, scan position) keys; the owner of each position resolves
        // its label afterwards (ivfhnsw_gpu_resolve_keys_dev)
        for (int j = lane; j < k; j += 64) {
            const unsigned long long key =
                pos[j] == 0xffffffffu ? kKeyInit : (((unsigned long long)f32_orderable(val[j]) << 32) | pos[j]);
            out_k[j] = (long long)(key ^ kSignFlip);
        }
        return;
    }
    // labels of the survivors (segments ascend in vpos)
    const Seg *sq = segs + (size_t)q * max_seg;
    const uint32_t nseg = hdr[q].nseg;
    for (int j = lane; j < k; j += 64) {
        const uint32_t vpos = pos[j];
        long long lab = -1;
        if (vpos != 0xffffffffu && nseg) {
            uint32_t a = 0, b = nseg - 1;
            while (a < b) {
                const uint32_t mid = (a + b + 1) >> 1;
                if (sq[mid].vpos <= vpos)
                    a = mid;
                else
                    b = mid - 1;
            }
            const Seg sg = sq[a];
            lab = (long long)t.ids[sg.start + (vpos - sg.vpos)];
        }
        out_d[j] = val[j];
        out_l[j] = lab;
    }
}

hipError_t launch_heap_replay(hipStream_t s, const IvfTables &t, const Seg *segs, const PlanHdr *hdr, int max_seg,
                              const uint64_t *stream, const uint32_t *stream_len, uint32_t stream_cap, int nq, int k,
                              float *dist, int64_t *labels, uint32_t *status, int64_t *out_keys)
{
    if (nq == 0)
        return hipSuccess;
    hipLaunchKernelGGL(heap_replay_kernel, dim3((nq + 3) / 4), dim3(256), (size_t)4 * 2 * k * sizeof(float), s, t, segs, hdr, max_seg,
                       reinterpret_cast<const unsigned long long *>(stream), stream_len, stream_cap, nq, k, dist,
                       reinterpret_cast<long long *>(labels), status, reinterpret_cast<long long *>(out_keys));
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
                       hdr, max_seg, k, k64, reinterpret_cast<unsigned long long *>(stream), stream_len, stream_cap, t.M)
    switch (t.M) {
    case 4: IVFHNSW_TOPK(4); break;
    case 8: IVFHNSW_TOPK(8); break;
    case 16: IVFHNSW_TOPK(16); break;
    case 32: IVFHNSW_TOPK(32); break;
    default: {
        const size_t shm = (size_t)t.M * 1024;
        if (t.M % 4 || shm > kScanDynLdsMax)
            return hipErrorInvalidValue;
        auto *kern = scan_topk_kernel<0>;
        static DynLdsState attr_set;
        if (hipError_t e = raise_dyn_lds((const void *)kern, shm, attr_set); e != hipSuccess)
            return e;
        hipLaunchKernelGGL(kern, grid, block, shm, s, t.codes, t.norm_codes, luts, t.norm_table, segs, lpos, hdr, max_seg, k,
                           k64, reinterpret_cast<unsigned long long *>(stream), stream_len, stream_cap, t.M);
        break;
    }
    }
#undef IVFHNSW_TOPK
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
