// Construction side of the IVFADC path: what IndexIVF_HNSW::add_batch (IndexIVF_HNSW.cpp:75-121) computes for a
// batch of base vectors before it appends them to the lists --
//   residual (fvec_madd with -1, :258-262 via compute_residuals) -> [OPQ apply] -> PQ codes (pq->compute_codes)
//   -> decode -> [OPQ transform_transpose] -> reconstruct (fvec_madd with +1) -> squared norm
//   (fvec_norms_L2sqr) -> norm code (norm_pq->compute_codes).
// The centroid assignment in front of it is the coarse walk with k = 1 (IndexIVF_HNSW.cpp:68-72), the OPQ
// products are launch_opq (kernels_search.hip).  Everything here is byte output, so the float orders are the
// contract: faiss's SSE kernels (4 partial sums over blocks of 4, zero-padded tail, (s0+s1)+(s2+s3)) for
// fvec_L2sqr and fvec_norm_L2sqr, unfused mul/add for fvec_madd, first minimum wins in the arg-min loops
// (strict '<').  faiss itself is absent from the reference tree (empty submodule): these orders are the
// published behaviour of its SSE build, not something that could be checked here.
#include "ivfhnsw_kernels.h"
#include "device_common.h"

#include <float.h>

namespace ivfhnsw_gpu_impl {

namespace {

// c = a + bf * b per element (faiss::fvec_madd), bf = -1: residual, bf = +1: reconstruction.
// b is row idx[i] of the centroid table.
__global__ __launch_bounds__(256) void madd_rows_kernel(const float *a, float bf, const float *__restrict__ table,
                                                        const uint32_t *__restrict__ idx, float *c, size_t n, int d)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * (size_t)d)
        return;
    const size_t i = e / (size_t)d;
    const int j = (int)(e - i * (size_t)d);
    c[e] = __fadd_rn(a[e], __fmul_rn(bf, table[(size_t)idx[i] * d + j]));
}

// faiss fvec_L2sqr, SSE build
template <int DSUB> __device__ __forceinline__ float l2_sse_order(const float *x, const float *y, int dsub_rt)
{
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
#pragma unroll
    for (; i + 4 <= dsub; i += 4) {
        const float t0 = __fsub_rn(x[i], y[i]), t1 = __fsub_rn(x[i + 1], y[i + 1]);
        const float t2 = __fsub_rn(x[i + 2], y[i + 2]), t3 = __fsub_rn(x[i + 3], y[i + 3]);
        s0 = __fadd_rn(s0, __fmul_rn(t0, t0));
        s1 = __fadd_rn(s1, __fmul_rn(t1, t1));
        s2 = __fadd_rn(s2, __fmul_rn(t2, t2));
        s3 = __fadd_rn(s3, __fmul_rn(t3, t3));
    }
    if (i < dsub) {
        const float t = __fsub_rn(x[i], y[i]);
        s0 = __fadd_rn(s0, __fmul_rn(t, t));
    }
    if (i + 1 < dsub) {
        const float t = __fsub_rn(x[i + 1], y[i + 1]);
        s1 = __fadd_rn(s1, __fmul_rn(t, t));
    }
    if (i + 2 < dsub) {
        const float t = __fsub_rn(x[i + 2], y[i + 2]);
        s2 = __fadd_rn(s2, __fmul_rn(t, t));
    }
    return __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
}

// pq->compute_codes: code[i][m] = first arg-min over c of ||r[i][m] - centroid[m][c]||^2.
// One block = 256 vectors x ONE sub-quantizer: its 256 code words sit in LDS (8 KB at dsub 8) and every lane of
// a wavefront reads the same word at the same time (broadcast, no bank conflicts); the sub-vector stays in
// registers.  blockIdx.y = m.
template <int DSUB>
__global__ __launch_bounds__(256) void pq_encode_kernel(const float *__restrict__ r, const float *__restrict__ cb,
                                                        uint8_t *__restrict__ codes, size_t n, int d, int M, int dsub_rt)
{
    extern __shared__ __attribute__((aligned(16))) float s_cb[]; // [256][dsub]
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    const int m = blockIdx.y;
    for (int e = threadIdx.x; e < 256 * dsub; e += 256)
        s_cb[e] = cb[(size_t)m * 256 * dsub + e];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    constexpr int RMAX = DSUB > 0 ? DSUB : 64;
    float x[RMAX];
#pragma unroll
    for (int j = 0; j < RMAX; j++)
        x[j] = j < dsub ? r[i * (size_t)d + (size_t)m * dsub + j] : 0.f;
    float best = 1e20f; // faiss starts its search at 1e20, not at infinity
    int arg = -1;
    for (int c = 0; c < 256; c++) {
        const float dist = l2_sse_order<DSUB>(x, s_cb + c * dsub, dsub);
        if (dist < best) {
            best = dist;
            arg = c;
        }
    }
    codes[i * (size_t)M + m] = (uint8_t)arg; // arg stays -1 (-> 255) only if every distance is >= 1e20 or NaN
}

// pq->decode: dec[i][m*dsub + j] = centroid[m][code[i][m]][j]
__global__ __launch_bounds__(256) void pq_decode_kernel(const uint8_t *__restrict__ codes, const float *__restrict__ cb,
                                                        float *__restrict__ dec, size_t n, int d, int M, int dsub)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * (size_t)d)
        return;
    const size_t i = e / (size_t)d;
    const int j = (int)(e - i * (size_t)d);
    const int m = j / dsub;
    dec[e] = cb[((size_t)m * 256 + codes[i * (size_t)M + m]) * dsub + (j - m * dsub)];
}

// fvec_norms_L2sqr + norm_pq->compute_codes (ProductQuantizer(1, 1, 8)): a quad of lanes per vector, lane l owns
// SSE partial sum l (elements 4b + l), the quad adds (s0+s1)+(s2+s3); then each lane searches 64 of the 256
// norm code words and the quad keeps the smallest (distance, index) -- the first minimum of the serial loop.
__global__ __launch_bounds__(256) void norm_code_kernel(const float *__restrict__ rec, const float *__restrict__ ntab,
                                                        uint8_t *__restrict__ norm_codes, float *__restrict__ norms_out,
                                                        size_t n, int d)
{
    __shared__ float s_nt[256];
    s_nt[threadIdx.x] = ntab[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    const int l = threadIdx.x & 3;
    float s = 0.f;
    if (i < n) {
        const float *row = rec + i * (size_t)d;
        for (int b = 0; b + 4 <= d; b += 4) {
            const float v = row[b + l];
            s = __fadd_rn(s, __fmul_rn(v, v));
        }
        const int tail = d & 3, b = d & ~3;
        if (l < tail) {
            const float v = row[b + l];
            s = __fadd_rn(s, __fmul_rn(v, v));
        }
    }
    const float nrm = __fadd_rn(__fadd_rn(quad_bcast<0>(s), quad_bcast<1>(s)), __fadd_rn(quad_bcast<2>(s), quad_bcast<3>(s)));
    float best = 1e20f;
    int arg = -1;
    for (int c = l * 64; c < l * 64 + 64; c++) {
        const float t = __fsub_rn(nrm, s_nt[c]);
        const float dist = __fmul_rn(t, t); // one element: s0 = 0 + t*t, (s0+0)+(0+0)
        if (dist < best) {
            best = dist;
            arg = c;
        }
    }
    // lexicographic (dist, index) minimum over the quad; a lane that found nothing (arg -1) never wins
    unsigned long long key = arg < 0 ? ~0ull : ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)arg;
#pragma unroll
    for (int k = 1; k < 4; k <<= 1) {
        const unsigned long long o = __shfl_xor(key, k, 64);
        key = o < key ? o : key;
    }
    if (i < n && l == 0) {
        norm_codes[i] = key == ~0ull ? (uint8_t)255 : (uint8_t)(key & 0xffu);
        if (norms_out)
            norms_out[i] = nrm;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Grouping construction (IndexIVF_HNSW_Grouping.cpp:43-157).  Per group: cv[s] = neighbour_s - centroid, alpha,
// sub-centroids centroid + alpha * cv[s], every point's nearest sub-centroid; the code bytes then come from the
// same kernels as above with the sub-centroid table in place of the centroid table.

// cv[g][s][:] = vectors[nn[g][s]] + (-1 * centroid_g)   (fvec_madd, :70-74)      MODE 0
// sub[g][s][:] = centroid_g + alpha_g * cv[g][s][:]      (fvec_madd, :82-87)      MODE 1
template <int MODE>
__global__ __launch_bounds__(256) void group_table_kernel(const float *__restrict__ vectors,
                                                          const uint32_t *__restrict__ centroid_idx,
                                                          const uint32_t *__restrict__ nn, const float *__restrict__ alphas,
                                                          const float *__restrict__ cv_in, float *__restrict__ out,
                                                          size_t ngroups, int nsubc, int d)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)nsubc * d;
    if (e >= ngroups * per)
        return;
    const size_t g = e / per;
    const int j = (int)(e % (size_t)d);
    const float c = vectors[(size_t)centroid_idx[g] * d + j];
    if (MODE == 0) {
        const int sidx = (int)((e - g * per) / (size_t)d);
        out[e] = __fadd_rn(vectors[(size_t)nn[g * nsubc + sidx] * d + j], __fmul_rn(-1.f, c));
    } else {
        out[e] = __fadd_rn(c, __fmul_rn(alphas[g], cv_in[e]));
    }
}

// ivfhnsw::fvec_L2sqr (utils.cpp:22-52) between x[] (LDS, row p) and y[] computed on the fly: lane-free form,
// eight accumulators, element j goes to accumulator j % 8 in increasing j, dims beyond a multiple of 16 ignored,
// accumulators summed left to right.
struct L2Avx8 {
    float a[8];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < 8; k++)
            a[k] = 0.f;
    }
    __device__ __forceinline__ float sum() const
    {
        float r = __fadd_rn(a[0], a[1]);
#pragma unroll
        for (int k = 2; k < 8; k++)
            r = __fadd_rn(r, a[k]);
        return r;
    }
};

// Does candidate (neg, num, den) displace the best so far?  The reference keeps the top of a max-heap of
// pair<-dist, pair<numerator, denominator>> (std::pair order); a NaN distance never displaces anything and is
// displaced by anything.  Written without branches on purpose: the nested if/else form of this test inside
// the candidate loop was miscompiled by hipcc 7.2 for gfx950 (the distance of a share's second candidate was
// dropped while its numerator was kept; adding a printf to the loop made it correct), see DESIGN.md 3.4.
__device__ __forceinline__ bool alpha_cand_better(bool have, float bneg, float bnum, float bden, float neg, float num,
                                                  float den)
{
    const bool ok = neg == neg, bok = bneg == bneg;
    const bool lt = bneg < neg;
    const bool eq = !(neg < bneg) & !lt;
    const bool second = (bnum < num) | (!(num < bnum) & (bden < den));
    return !have | (ok & (!bok | lt | (eq & second)));
}

// One block per group, 64 points per tile, thread = (point, quarter of the sub-centroids).
//  MODE 0 (compute_alpha, :691-733): table = cv; per point the (numerator, denominator) of the candidate
//          sub-centroid centroid + (max(<cv,pv>,0)/norm) * cv closest to the point (ties: larger numerator,
//          then denominator -- std::pair order of the reference's max-heap).
//  MODE 1 (compute_subcentroid_idxs, :673-689): table = sub-centroids; first minimum of the distance.
template <int MODE>
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ vectors,
                                                           const uint32_t *__restrict__ centroid_idx,
                                                           const float *__restrict__ table, // [G][nsubc][d]
                                                           const float *__restrict__ cv_norms, // [G][nsubc] (MODE 0)
                                                           const unsigned long long *__restrict__ offsets,
                                                           const float *__restrict__ x, float *__restrict__ out_num,
                                                           float *__restrict__ out_den, uint32_t *__restrict__ out_sub,
                                                           int nsubc, int d)
{
    extern __shared__ __attribute__((aligned(16))) float smem_g[];
    const int ld = d + 1;
    float *s_tab = smem_g;                     // [nsubc][ld]
    float *s_x = s_tab + (size_t)nsubc * ld;   // [64][ld]
    float *s_c = s_x + (size_t)64 * ld;        // [d]
    float *s_n = s_c + d;                      // [nsubc]
    float *s_r0 = s_n + nsubc;                 // [256] reduction scratch
    float *s_r1 = s_r0 + 256;
    float *s_r2 = s_r1 + 256;
    const size_t g = blockIdx.x;
    const size_t p0 = offsets[g], p1 = offsets[g + 1];
    if (p0 == p1)
        return;
    for (int e = threadIdx.x; e < nsubc * d; e += 256)
        s_tab[(e / d) * ld + (e % d)] = table[g * (size_t)nsubc * d + e];
    for (int e = threadIdx.x; e < d; e += 256)
        s_c[e] = vectors[(size_t)centroid_idx[g] * d + e];
    if (MODE == 0)
        for (int e = threadIdx.x; e < nsubc; e += 256)
            s_n[e] = cv_norms[g * (size_t)nsubc + e];
    const int p = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const int s_lo = (nsubc * qtr) / 4, s_hi = (nsubc * (qtr + 1)) / 4;
    const int d16 = d & ~15;
    for (size_t t0 = p0; t0 < p1; t0 += 64) {
        __syncthreads();
        const int np = (int)((p1 - t0) < 64 ? (p1 - t0) : 64);
        for (int e = threadIdx.x; e < np * d; e += 256)
            s_x[(e / d) * ld + (e % d)] = x[t0 * (size_t)d + e];
        __syncthreads();
        // per-thread best over its share of the sub-centroids
        bool have = false;
        float bneg = 0.f, bnum = 0.f, bden = 0.f; // MODE 0
        float bdist = 0.f;                        // MODE 1
        int bs = 0;
        if (p < np) {
            const float *xr = s_x + p * ld;
            for (int sc = s_lo; sc < s_hi; sc++) {
                const float *tr = s_tab + sc * ld;
                if (MODE == 0) {
                    // numerator = faiss::fvec_inner_product(cv, pv) with pv = x + (-1 * centroid): SSE order
                    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
                    int j = 0;
                    for (; j + 4 <= d; j += 4) {
                        q0 = __fadd_rn(q0, __fmul_rn(tr[j], __fadd_rn(xr[j], __fmul_rn(-1.f, s_c[j]))));
                        q1 = __fadd_rn(q1, __fmul_rn(tr[j + 1], __fadd_rn(xr[j + 1], __fmul_rn(-1.f, s_c[j + 1]))));
                        q2 = __fadd_rn(q2, __fmul_rn(tr[j + 2], __fadd_rn(xr[j + 2], __fmul_rn(-1.f, s_c[j + 2]))));
                        q3 = __fadd_rn(q3, __fmul_rn(tr[j + 3], __fadd_rn(xr[j + 3], __fmul_rn(-1.f, s_c[j + 3]))));
                    }
                    if (j < d)
                        q0 = __fadd_rn(q0, __fmul_rn(tr[j], __fadd_rn(xr[j], __fmul_rn(-1.f, s_c[j]))));
                    if (j + 1 < d)
                        q1 = __fadd_rn(q1, __fmul_rn(tr[j + 1], __fadd_rn(xr[j + 1], __fmul_rn(-1.f, s_c[j + 1]))));
                    if (j + 2 < d)
                        q2 = __fadd_rn(q2, __fmul_rn(tr[j + 2], __fadd_rn(xr[j + 2], __fmul_rn(-1.f, s_c[j + 2]))));
                    float num = __fadd_rn(__fadd_rn(q0, q1), __fadd_rn(q2, q3));
                    num = num > 0.f ? num : 0.f;
                    const float den = s_n[sc];
                    const float al = num / den; // IEEE division: correctly rounded (no fast-math)
                    L2Avx8 acc;
                    acc.init();
                    for (int b = 0; b < d16; b += 8) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const float sub = __fadd_rn(s_c[b + k], __fmul_rn(al, tr[b + k]));
                            const float df = __fsub_rn(xr[b + k], sub);
                            acc.a[k] = __fadd_rn(acc.a[k], __fmul_rn(df, df));
                        }
                    }
                    const float neg = -acc.sum();
                    const bool better = alpha_cand_better(have, bneg, bnum, bden, neg, num, den);
                    have = true;
                    bneg = better ? neg : bneg;
                    bnum = better ? num : bnum;
                    bden = better ? den : bden;
                } else {
                    L2Avx8 acc;
                    acc.init();
                    for (int b = 0; b < d16; b += 8) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const float df = __fsub_rn(tr[b + k], xr[b + k]);
                            acc.a[k] = __fadd_rn(acc.a[k], __fmul_rn(df, df));
                        }
                    }
                    const float dist = acc.sum();
                    const bool take = !have | (dist < bdist); // first minimum inside this share (increasing sc)
                    have = true;
                    bdist = take ? dist : bdist;
                    bs = take ? sc : bs;
                }
            }
        }
        // fold the four shares of a point in increasing order of sub-centroid index (= the serial loop's order)
        s_r0[threadIdx.x] = MODE == 0 ? bneg : bdist;
        s_r1[threadIdx.x] = MODE == 0 ? bnum : __int_as_float(bs);
        s_r2[threadIdx.x] = MODE == 0 ? bden : (have ? 1.f : 0.f);
        __syncthreads();
        if (qtr == 0 && p < np) {
            if (MODE == 0) {
                float fneg = bneg, fnum = bnum, fden = bden;
                bool fhave = have;
                for (int k = 1; k < 4; k++) {
                    if (((nsubc * (k + 1)) / 4) == ((nsubc * k) / 4))
                        continue; // empty share
                    const float neg = s_r0[k * 64 + p], num = s_r1[k * 64 + p], den = s_r2[k * 64 + p];
                    const bool better = alpha_cand_better(fhave, fneg, fnum, fden, neg, num, den);
                    fhave = true;
                    fneg = better ? neg : fneg;
                    fnum = better ? num : fnum;
                    fden = better ? den : fden;
                }
                out_num[t0 + p] = fnum;
                out_den[t0 + p] = fden;
            } else {
                float fd = bdist;
                int fs = bs;
                bool fhave = have;
                for (int k = 1; k < 4; k++) {
                    if (s_r2[k * 64 + p] == 0.f)
                        continue;
                    const float dist = s_r0[k * 64 + p];
                    const bool take = !fhave | (dist < fd);
                    fhave = true;
                    fs = take ? __float_as_int(s_r1[k * 64 + p]) : fs;
                    fd = take ? dist : fd;
                }
                out_sub[t0 + p] = (uint32_t)fs;
            }
        }
    }
}

// alpha_g = sum of the points' numerators / sum of their denominators, both added in point order (:724-727)
__global__ void group_alpha_kernel(const unsigned long long *__restrict__ offsets, const float *__restrict__ num,
                                   const float *__restrict__ den, float *__restrict__ alphas, size_t ngroups)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups)
        return;
    float a = 0.f, b = 0.f;
    for (size_t i = offsets[g]; i < offsets[g + 1]; i++) {
        a = __fadd_rn(a, num[i]);
        b = __fadd_rn(b, den[i]);
    }
    alphas[g] = b > 0.f ? __fdiv_rn(a, b) : 0.f;
}

// table row of every point: g * nsubc + sub-centroid index
__global__ void group_rows_kernel(const unsigned long long *__restrict__ offsets, const uint32_t *__restrict__ sub,
                                  uint32_t *__restrict__ rows, size_t ngroups, int nsubc)
{
    const size_t g = blockIdx.x;
    for (size_t i = offsets[g] + threadIdx.x; i < offsets[g + 1]; i += blockDim.x)
        rows[i] = (uint32_t)(g * (size_t)nsubc + sub[i]);
}

} // namespace

hipError_t launch_madd_rows(hipStream_t s, const float *a, float bf, const float *table, const uint32_t *idx, float *c,
                            size_t n, int d)
{
    if (n == 0)
        return hipSuccess;
    const size_t blocks = (n * (size_t)d + 255) / 256;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(madd_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, bf, table, idx, c, n, d);
    return hipGetLastError();
}

hipError_t launch_pq_encode(hipStream_t s, const float *r, const float *cb, uint8_t *codes, size_t n, int d, int M)
{
    if (n == 0)
        return hipSuccess;
    const int dsub = d / M;
    if (dsub < 1 || dsub > 64 || M > 65535)
        return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)M), block(256);
    const size_t shm = (size_t)256 * dsub * sizeof(float);
    switch (dsub) {
    case 4: hipLaunchKernelGGL(pq_encode_kernel<4>, grid, block, shm, s, r, cb, codes, n, d, M, dsub); break;
    case 6: hipLaunchKernelGGL(pq_encode_kernel<6>, grid, block, shm, s, r, cb, codes, n, d, M, dsub); break;
    case 8: hipLaunchKernelGGL(pq_encode_kernel<8>, grid, block, shm, s, r, cb, codes, n, d, M, dsub); break;
    case 12: hipLaunchKernelGGL(pq_encode_kernel<12>, grid, block, shm, s, r, cb, codes, n, d, M, dsub); break;
    case 16: hipLaunchKernelGGL(pq_encode_kernel<16>, grid, block, shm, s, r, cb, codes, n, d, M, dsub); break;
    default: hipLaunchKernelGGL(pq_encode_kernel<0>, grid, block, shm, s, r, cb, codes, n, d, M, dsub); break;
    }
    return hipGetLastError();
}

hipError_t launch_pq_decode(hipStream_t s, const uint8_t *codes, const float *cb, float *dec, size_t n, int d, int M)
{
    if (n == 0)
        return hipSuccess;
    const size_t blocks = (n * (size_t)d + 255) / 256;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(pq_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, cb, dec, n, d, M, d / M);
    return hipGetLastError();
}

hipError_t launch_norm_codes(hipStream_t s, const float *rec, const float *ntab, uint8_t *norm_codes, float *norms_out,
                             size_t n, int d)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(norm_code_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, rec, ntab, norm_codes,
                       norms_out, n, d);
    return hipGetLastError();
}

hipError_t launch_group_table(hipStream_t s, int mode, const float *vectors, const uint32_t *centroid_idx,
                              const uint32_t *nn, const float *alphas, const float *cv_in, float *out, size_t ngroups,
                              int nsubc, int d)
{
    if (ngroups == 0)
        return hipSuccess;
    const size_t blocks = (ngroups * (size_t)nsubc * d + 255) / 256;
    if (blocks > 0x7fffffffull)
        return hipErrorInvalidValue;
    if (mode == 0)
        hipLaunchKernelGGL(group_table_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, vectors, centroid_idx, nn,
                           alphas, cv_in, out, ngroups, nsubc, d);
    else
        hipLaunchKernelGGL(group_table_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, vectors, centroid_idx, nn,
                           alphas, cv_in, out, ngroups, nsubc, d);
    return hipGetLastError();
}

size_t group_points_lds_bytes(int nsubc, int d)
{
    return ((size_t)(nsubc + 64) * (d + 1) + d + nsubc + 3 * 256) * sizeof(float);
}

hipError_t launch_group_points(hipStream_t s, int mode, const float *vectors, const uint32_t *centroid_idx,
                               const float *table, const float *cv_norms, const unsigned long long *offsets,
                               const float *x, float *out_num, float *out_den, uint32_t *out_sub, size_t ngroups,
                               int nsubc, int d)
{
    if (ngroups == 0)
        return hipSuccess;
    const size_t shm = group_points_lds_bytes(nsubc, d);
    if (shm > 160 * 1024 || ngroups > 0x7fffffffull)
        return hipErrorInvalidValue;
    if (mode == 0) {
        (void)hipFuncSetAttribute((const void *)group_points_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(group_points_kernel<0>, dim3((unsigned)ngroups), dim3(256), shm, s, vectors, centroid_idx,
                           table, cv_norms, offsets, x, out_num, out_den, out_sub, nsubc, d);
    } else {
        (void)hipFuncSetAttribute((const void *)group_points_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(group_points_kernel<1>, dim3((unsigned)ngroups), dim3(256), shm, s, vectors, centroid_idx,
                           table, cv_norms, offsets, x, out_num, out_den, out_sub, nsubc, d);
    }
    return hipGetLastError();
}

hipError_t launch_group_alpha(hipStream_t s, const unsigned long long *offsets, const float *num, const float *den,
                              float *alphas, size_t ngroups)
{
    if (ngroups == 0)
        return hipSuccess;
    hipLaunchKernelGGL(group_alpha_kernel, dim3((unsigned)((ngroups + 63) / 64)), dim3(64), 0, s, offsets, num, den,
                       alphas, ngroups);
    return hipGetLastError();
}

hipError_t launch_group_rows(hipStream_t s, const unsigned long long *offsets, const uint32_t *sub, uint32_t *rows,
                             size_t ngroups, int nsubc)
{
    if (ngroups == 0)
        return hipSuccess;
    hipLaunchKernelGGL(group_rows_kernel, dim3((unsigned)ngroups), dim3(256), 0, s, offsets, sub, rows, ngroups, nsubc);
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
