// Exact k-nearest-neighbour tables by brute force on the matrix cores (the open half of row f4).
//
// What it stands for in the reference: every construction step that asks "which stored vectors are nearest to this one"
// -- hnswlib's addPoint finds a new node's link candidates with an efConstruction-wide search of the graph built so far
// (hnswlib/hnswalg.cpp:112-225), Grouping's add_group asks the graph for the nsubc + 1 nearest centroids of a centroid
// (IndexIVF_HNSW_Grouping.cpp:47-62), and the drivers score Recall@1 against ground-truth files that are exact brute
// force (tests/test_ivfhnsw_sift1b.cpp:173-215).  All three are approximations of, or ARE, the exact table this kernel
// computes; the bench and the tests use it to build million-node neighbour graphs, Grouping neighbour tables and ground
// truth without leaving the library.
//
// Arithmetic (the contract the oracle restates, orc_knn):
//     norm(v)    = fmaf chain over k = 0..d-1 of v[k] * v[k], starting from 0
//     dot(q, x)  = fmaf chain over k = 0..d-1 of q[k] * x[k]      -- v_mfma_f32_32x32x2_f32 is exactly that chain
//     dist(q, x) = (norm(q) + norm(x)) - 2 * dot(q, x)            -- two rounded operations, 2 * dot is exact
//     result     = the k smallest (dist, id) pairs, ascending
// This IS a dense contraction (nq x d times d x nx), so it belongs on MFMA; it is not the reference's search-time
// distance (the walk keeps hnswalg.cpp:326-357's order, kernels_hnsw.hip).
//
// Shape: a 256-thread workgroup owns 128 query rows, wave w the 32-row strip [32w, 32w + 32).  The strip (operand A:
// lane l feeds q[row l % 32][k = 2j + l / 32]) stays in d/2 registers for the whole sweep; the workgroup streams x in
// tiles of 32 columns through LDS, stored k-major ([k][32 columns]: the staging writes and the operand-B reads of a
// half-wave both touch 32 consecutive words, no bank conflicts); the next tile's global loads are in flight in
// registers behind the current tile's d/2 MFMAs (one LDS tile, so that with k <= 16 two workgroups fit a CU and fill
// each other's barriers).  Selection: every lane holds the running k-th distance of its 16 rows;
// a candidate below it (rare after the first tiles) is appended to the row's LDS buffer, placed by a ballot; a row whose
// buffer could overflow in the next tile is compacted by its wave (rank of every key among the row's keys: keys are
// unique, id in the low word).  Columns arrive in increasing id order, so among equal distances the earlier id wins by
// the strict '<' alone.  Few query rows (ground truth for 10 k queries): the columns are split over blockIdx.y and the
// partial tables merged by knn_merge_kernel.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>

namespace ivfhnsw_gpu_impl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr unsigned long long kKnnKeyNone = ~0ull;

__global__ __launch_bounds__(256) void knn_norms_kernel(const float *__restrict__ x, float *__restrict__ out, size_t n, int d,
                                                        int ld)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float *r = x + i * (size_t)ld;
    float acc = 0.f;
    for (int k = 0; k < d; k++)
        acc = __fmaf_rn(r[k], r[k], acc);
    out[i] = acc;
}

// D2 = padded d / 2 (rows are read as `d` floats, the rest of the chain multiplies zeros: fmaf(0, 0, acc) = acc)
// CAP = entries of a row's candidate buffer; a row is compacted when it holds more than CAP - 32
template <int D2, int CAP>
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(const float *__restrict__ Q, const float *__restrict__ X,
                                                          const float *__restrict__ qn, const float *__restrict__ xn,
                                                          int nq, size_t nx, int d, int k, int mode,
                                                          size_t cols_per_split,
                                                          unsigned long long *__restrict__ out_keys)
{
    constexpr int DP = 2 * D2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_x = reinterpret_cast<float *>(smem);                                       // [DP][32]
    float *s_xn = s_x + DP * 32;                                                        // [32]
    unsigned long long *s_buf = reinterpret_cast<unsigned long long *>(s_xn + 32);      // [4][32][CAP]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_buf + 4 * 32 * CAP);               // [4][32]
    float *s_thr = reinterpret_cast<float *>(s_cnt + 128);                              // [4][32]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m = lane & 31, kk = lane >> 5;
    const int r0 = blockIdx.x * 128 + wave * 32;
    const size_t c_begin = (size_t)blockIdx.y * cols_per_split;
    size_t c_end = c_begin + cols_per_split < nx ? c_begin + cols_per_split : nx;
    // mode 2 (only EARLIER rows are candidates: the neighbour table of an incremental construction, row i against
    // rows 0..i-1): columns at and beyond the block's last row are never candidates
    if (mode == 2 && c_end > (size_t)(blockIdx.x + 1) * 128)
        c_end = (size_t)(blockIdx.x + 1) * 128;
    unsigned long long *buf = s_buf + (size_t)wave * 32 * CAP;
    uint32_t *cnt = s_cnt + wave * 32;
    float *thr_s = s_thr + wave * 32;

    // operand A: the wave's 32 query rows, d/2 registers per lane
    float a[D2];
    {
        const int row = min(r0 + m, nq - 1);
        const float *qr = Q + (size_t)row * d;
#pragma unroll
        for (int j = 0; j < D2; j++) {
            const int kx = 2 * j + kk;
            a[j] = kx < d ? qr[kx] : 0.f;
        }
    }
    // the 16 rows whose dot products this lane holds (C/D layout of the 32x32 MFMA)
    float qn_r[16], thr[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        qn_r[r] = qn[min(r0 + row, nq - 1)];
        thr[r] = FLT_MAX;
    }
    if (lane < 32) {
        cnt[lane] = 0;
        thr_s[lane] = FLT_MAX;
    }

    // staging: thread -> (column tid % 32, 16 floats starting at (tid / 32) * 16) of the tile
    const int st_col = tid & 31, st_k0 = (tid >> 5) * 16;
    float4 pre[4];
    float pre_n = 0.f;
    auto fetch = [&](size_t c0) {
        const size_t c = c0 + st_col;
        const bool ok = c < c_end;
        const float *xr = X + (ok ? c : 0) * (size_t)d;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int kx = st_k0 + 4 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && kx + 3 < d)
                v = *reinterpret_cast<const float4 *>(xr + kx);
            else if (ok && kx < d) {
                v.x = xr[kx];
                v.y = kx + 1 < d ? xr[kx + 1] : 0.f;
                v.z = kx + 2 < d ? xr[kx + 2] : 0.f;
            }
            pre[i] = v;
        }
        if (tid < 32)
            pre_n = ok ? xn[c] : 0.f;
    };
    auto stash = [&]() {
        float *dst = s_x;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int kx = st_k0 + 4 * i;
            if (kx < DP) {
                dst[(kx + 0) * 32 + st_col] = pre[i].x;
                dst[(kx + 1) * 32 + st_col] = pre[i].y;
                dst[(kx + 2) * 32 + st_col] = pre[i].z;
                dst[(kx + 3) * 32 + st_col] = pre[i].w;
            }
        }
        if (tid < 32)
            s_xn[tid] = pre_n;
    };

    fetch(c_begin);
    for (size_t c0 = c_begin; c0 < c_end; c0 += 32) {
        __syncthreads(); // the previous tile has been consumed by every wave
        stash();
        __syncthreads();
        if (c0 + 32 < c_end)
            fetch(c0 + 32); // in flight behind this tile's MFMAs
        const float *xt = s_x;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < D2; j++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], xt[(2 * j + kk) * 32 + m], acc, 0, 0, 0);
        const float xnc = s_xn[m];
        const size_t col = c0 + m;
        const bool col_ok = col < c_end;
        bool any_new = false;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float dist = __fsub_rn(__fadd_rn(qn_r[r], xnc), __fmul_rn(2.0f, acc[r]));
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const bool pass = col_ok && dist < thr[r] && (mode == 0 || (mode == 1 ? (long long)col != (long long)(r0 + row)
                                                                                   : (long long)col < (long long)(r0 + row)));
            const unsigned long long mask = __ballot(pass);
            if (mask) { // wave-uniform: rare once the thresholds have settled
                const uint32_t half = kk ? (uint32_t)(mask >> 32) : (uint32_t)mask;
                const uint32_t base = cnt[row];
                if (pass)
                    buf[(size_t)row * CAP + base + __popc(half & ((1u << m) - 1u))] =
                        ((unsigned long long)f32_orderable(dist) << 32) | (uint32_t)col;
                if (m == 0 && half)
                    cnt[row] = base + __popc(half);
                any_new = true;
            }
        }
        if (any_new) {
            // rows that could overflow in the next tile (32 more entries at most): keep their k smallest
            __builtin_amdgcn_wave_barrier();
            unsigned long long need = __ballot(lane < 32 && cnt[lane] > (uint32_t)(CAP - 32));
            while (need) {
                const int row = __ffsll((long long)need) - 1;
                need &= need - 1;
                const int n = (int)cnt[row];
                unsigned long long *rb = buf + (size_t)row * CAP;
                // two keys per lane; rank = number of smaller keys (all distinct)
                const unsigned long long k0 = lane < n ? rb[lane] : kKnnKeyNone;
                const unsigned long long k1 = lane + 64 < n ? rb[lane + 64] : kKnnKeyNone;
                int rk0 = 0, rk1 = 0;
                for (int i = 0; i < n; i++) {
                    const unsigned long long v = rb[i];
                    rk0 += v < k0 ? 1 : 0;
                    rk1 += v < k1 ? 1 : 0;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < n && rk0 < k)
                    rb[rk0] = k0;
                if (lane + 64 < n && rk1 < k)
                    rb[rk1] = k1;
                if (lane == 0)
                    cnt[row] = (uint32_t)min(n, k);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0 && n >= k)
                    thr_s[row] = orderable_f32((uint32_t)(rb[k - 1] >> 32));
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; r++)
                thr[r] = thr_s[(r & 3) + 8 * (r >> 2) + 4 * kk];
        }
    }

    // final order of every row, then out: [split][nq][k] keys ascending, kKnnKeyNone beyond what was found
    for (int row = 0; row < 32; row++) {
        const int n = (int)cnt[row];
        unsigned long long *rb = buf + (size_t)row * CAP;
        const unsigned long long k0 = lane < n ? rb[lane] : kKnnKeyNone;
        const unsigned long long k1 = lane + 64 < n ? rb[lane + 64] : kKnnKeyNone;
        int rk0 = 0, rk1 = 0;
        for (int i = 0; i < n; i++) {
            const unsigned long long v = rb[i];
            rk0 += v < k0 ? 1 : 0;
            rk1 += v < k1 ? 1 : 0;
        }
        if (r0 + row < nq) {
            unsigned long long *o = out_keys + ((size_t)blockIdx.y * nq + (r0 + row)) * (size_t)k;
            if (lane < n && rk0 < k)
                o[rk0] = k0;
            if (lane + 64 < n && rk1 < k)
                o[rk1] = k1;
            for (int i = n + lane; i < k; i += 64)
                o[i] = kKnnKeyNone;
        }
    }
}

// one wavefront per query: the k smallest of the splits' sorted partial tables (keys distinct: ids differ)
__global__ __launch_bounds__(64) void knn_merge_kernel(const unsigned long long *__restrict__ part, int nsplit, int nq,
                                                       int k, uint32_t *__restrict__ ids, float *__restrict__ dists)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    const int total = nsplit * k;
    for (int i = lane; i < total; i += 64) {
        const int s = i / k, j = i - s * k;
        const unsigned long long key = part[((size_t)s * nq + q) * k + j];
        if (key == kKnnKeyNone)
            continue;
        int rank = 0;
        for (int t = 0; t < nsplit; t++) {
            // keys of split t below `key`: the lists are sorted, binary search
            const unsigned long long *p = part + ((size_t)t * nq + q) * k;
            int lo = 0, hi = k;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (p[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            ids[(size_t)q * k + rank] = (uint32_t)key;
            dists[(size_t)q * k + rank] = orderable_f32((uint32_t)(key >> 32));
        }
    }
    // slots nothing reached: fewer than k candidates exist
    int found = 0;
    for (int t = 0; t < nsplit; t++) {
        const unsigned long long *p = part + ((size_t)t * nq + q) * k;
        int lo = 0, hi = k;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (p[mid] != kKnnKeyNone)
                lo = mid + 1;
            else
                hi = mid;
        }
        found += lo;
    }
    for (int i = min(found, k) + lane; i < k; i += 64) {
        ids[(size_t)q * k + i] = 0xffffffffu;
        dists[(size_t)q * k + i] = FLT_MAX;
    }
}

template <int D2, int CAP>
hipError_t launch_knn_t(hipStream_t s, const float *Q, const float *X, const float *qn, const float *xn, int nq, size_t nx,
                        int d, int k, int mode, int nsplit, size_t cols_per_split, unsigned long long *part)
{
    const size_t shm = (size_t)(2 * D2 * 32 + 32) * sizeof(float) + (size_t)4 * 32 * CAP * sizeof(unsigned long long) +
                       128 * sizeof(uint32_t) + 128 * sizeof(float);
    auto *kern = knn_mfma_kernel<D2, CAP>;
    static DynLdsState attr_set;
    if (hipError_t e = raise_dyn_lds((const void *)kern, shm, attr_set); e != hipSuccess)
        return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((nq + 127) / 128), (unsigned)nsplit), dim3(256), shm, s, Q, X, qn, xn, nq, nx, d,
                       k, mode, cols_per_split, part);
    return hipGetLastError();
}

} // namespace

int knn_splits_for(size_t nq, size_t nx)
{
    // enough workgroups for two per CU; a split never shorter than 4096 columns
    const size_t row_blocks = (nq + 127) / 128;
    size_t s = (512 + row_blocks - 1) / row_blocks;
    const size_t max_s = (nx + 4095) / 4096;
    s = s < 1 ? 1 : s;
    s = s > max_s ? max_s : s;
    return (int)(s > 64 ? 64 : s);
}

hipError_t launch_knn_norms(hipStream_t s, const float *x, float *out, size_t n, int d)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, n, d, d);
    return hipGetLastError();
}

// part: [nsplit][nq][k] u64 workspace
hipError_t launch_knn(hipStream_t s, const float *Q, const float *X, const float *qn, const float *xn, size_t nq, size_t nx,
                      int d, int k, int mode, int nsplit, unsigned long long *part, uint32_t *ids, float *dists)
{
    if (nq == 0 || k == 0)
        return hipSuccess;
    if (d < 4 || d > 128 || (d & 3) || k > 80 || nq > 0x7fffffffull || nx > 0xffffffffull || nsplit < 1 || mode < 0 || mode > 2)
        return hipErrorInvalidValue;
    size_t cps = (nx + nsplit - 1) / nsplit;
    cps = (cps + 31) & ~(size_t)31;
    hipError_t e;
    const int d2 = (d + 1) / 2;
#define IVFHNSW_KNN(D2)                                                                                                \
    (k <= 16   ? launch_knn_t<D2, 48>(s, Q, X, qn, xn, (int)nq, nx, d, k, mode, nsplit, cps, part)              \
     : k <= 32 ? launch_knn_t<D2, 64>(s, Q, X, qn, xn, (int)nq, nx, d, k, mode, nsplit, cps, part)              \
               : launch_knn_t<D2, 112>(s, Q, X, qn, xn, (int)nq, nx, d, k, mode, nsplit, cps, part))
    if (d2 <= 16)
        e = IVFHNSW_KNN(16);
    else if (d2 <= 32)
        e = IVFHNSW_KNN(32);
    else if (d2 <= 48)
        e = IVFHNSW_KNN(48);
    else
        e = IVFHNSW_KNN(64);
#undef IVFHNSW_KNN
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)nq), dim3(64), 0, s, part, nsplit, (int)nq, k, ids, dists);
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
