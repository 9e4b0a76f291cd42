// Code-book training on the device: the Lloyd update of ProductQuantizer::train
// (IndexIVF_HNSW.cpp:536-593 hands residuals to faiss's clustering) and the d x d product X^T Y of OPQ's
// orthogonal Procrustes step (faiss OPQMatrix::train) on the matrix cores.
//
// The assignment step of a Lloyd iteration IS pq->compute_codes with the current code book: launch_pq_encode
// (kernels_encode.hip), the exact direct form in faiss's SSE order.  It is deliberately not reshaped into a
// ||x||^2 - 2 x.c + ||c||^2 GEMM for MFMA: that changes roundings, hence assignments near ties, hence every later
// iteration -- and the direct kernel already runs at the rate the training points arrive over PCIe.
// X^T Y contracts over the n training points (n = 65 536 at faiss's default): that one is a true dense
// contraction and runs on v_mfma_f32_32x32x2_f32, whose accumulation is a k-ordered fmaf chain -- a defined order
// the oracle restates (orc_xty).
#include "ivfhnsw_kernels.h"
#include "device_common.h"

namespace ivfhnsw_gpu_impl {

namespace {

// One 64-thread block per (sub-quantizer m, code word c): thread j < dsub owns component j of the new code word.
// The sum runs over the points in index order, in float, exactly as the host loop and the oracle take it
// (orc_pq_lloyd): the result is a function of the order, so the order is the contract.
__global__ __launch_bounds__(64) void lloyd_update_kernel(const float *__restrict__ x, const uint8_t *__restrict__ assign,
                                                          float *__restrict__ cb, size_t n, int d, int M, int dsub)
{
    const int m = blockIdx.x, c = blockIdx.y, j = threadIdx.x;
    float sum = 0.f;
    unsigned long long cnt = 0;
    const float *xj = x + (size_t)m * dsub + (j < dsub ? j : 0);
    // the assignment bytes are wave-uniform: 16 of them per pass, then the (rare: 1 in 256) matching points
    for (size_t i0 = 0; i0 < n; i0 += 16) {
        uint8_t a[16];
#pragma unroll
        for (int u = 0; u < 16; u++)
            a[u] = i0 + u < n ? assign[(i0 + u) * M + m] : (uint8_t)(c + 1);
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (a[u] == (uint8_t)c && i0 + u < n) {
                cnt++;
                sum = __fadd_rn(sum, xj[(i0 + u) * d]);
            }
    }
    if (cnt && j < dsub)
        cb[((size_t)m * 256 + c) * dsub + j] = __fdiv_rn(sum, (float)cnt);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// P[chunk][a][b] = fmaf chain over the chunk's points of X[i][a] * Y[i][b].  One wavefront per 32 x 32 tile and chunk:
// MFMA operand A = X^T (row a, k = point), operand B = Y (k = point, column b); lane l feeds X[i + l/32][a0 + l%32] and
// Y[i + l/32][b0 + l%32] -- both coalesced 128-byte row pieces.  Points beyond n contribute fmaf(0, 0, acc) = acc.
__global__ __launch_bounds__(64) void xty_mfma_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                      float *__restrict__ P, size_t n, int d, int chunk)
{
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const size_t i0 = (size_t)blockIdx.z * chunk;
    const int lane = threadIdx.x, m = lane & 31, kk = lane >> 5;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < chunk; s += 8) {
        float xa[4], yb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { // four MFMA steps' operands in flight
            const size_t i = i0 + s + 2 * u + kk;
            xa[u] = i < n ? X[i * d + a0 + m] : 0.f;
            yb[u] = i < n ? Y[i * d + b0 + m] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u], yb[u], acc, 0, 0, 0);
    }
    float *p = P + (size_t)blockIdx.z * d * d;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk; // C/D layout of the 32x32 MFMA
        p[(size_t)(a0 + row) * d + b0 + m] = acc[r];
    }
}

// any d (not a multiple of 32): one thread per output element, the same order
__global__ __launch_bounds__(256) void xty_scalar_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                         float *__restrict__ P, size_t n, int d, int chunk)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d * d)
        return;
    const int a = e / d, b = e - a * d;
    const size_t i0 = (size_t)blockIdx.y * chunk;
    const size_t i1 = i0 + chunk < n ? i0 + chunk : n;
    float acc = 0.f;
    for (size_t i = i0; i < i1; i++)
        acc = __fmaf_rn(X[i * d + a], Y[i * d + b], acc);
    P[(size_t)blockIdx.y * d * d + e] = acc;
}

// C = ((P[0] + P[1]) + P[2]) + ...
__global__ __launch_bounds__(256) void xty_reduce_kernel(const float *__restrict__ P, float *__restrict__ C, int dd,
                                                         int nchunks)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= dd)
        return;
    float t = P[e];
    for (int c = 1; c < nchunks; c++)
        t = __fadd_rn(t, P[(size_t)c * dd + e]);
    C[e] = t;
}

} // namespace

hipError_t launch_lloyd_update(hipStream_t s, const float *x, const uint8_t *assign, float *cb, size_t n, int d, int M)
{
    const int dsub = d / M;
    if (n == 0)
        return hipSuccess;
    if (dsub < 1 || dsub > 64 || M > 65535)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(lloyd_update_kernel, dim3((unsigned)M, 256), dim3(64), 0, s, x, assign, cb, n, d, M, dsub);
    return hipGetLastError();
}

hipError_t launch_xty(hipStream_t s, const float *X, const float *Y, float *partials, float *C, size_t n, int d)
{
    const int chunk = kXtyChunk;
    const int nchunks = (int)((n + chunk - 1) / chunk);
    if (n == 0 || nchunks > 65535)
        return hipErrorInvalidValue;
    if (d % 32 == 0)
        hipLaunchKernelGGL(xty_mfma_kernel, dim3(d / 32, d / 32, nchunks), dim3(64), 0, s, X, Y, partials, n, d, chunk);
    else
        hipLaunchKernelGGL(xty_scalar_kernel, dim3((d * d + 255) / 256, nchunks), dim3(256), 0, s, X, Y, partials, n, d,
                           chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(xty_reduce_kernel, dim3((d * d + 255) / 256), dim3(256), 0, s, partials, C, d * d, nchunks);
    return hipGetLastError();
}

} // namespace ivfhnsw_gpu_impl
