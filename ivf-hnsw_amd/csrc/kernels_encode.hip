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
__global__ __launch_bounds__(256) void madd_rows_kernel(const float *__restrict__ a, float bf,
                                                        const float *__restrict__ table,
                                                        const uint32_t *__restrict__ idx, float *__restrict__ c,
                                                        size_t n, int d)
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

} // namespace ivfhnsw_gpu_impl
