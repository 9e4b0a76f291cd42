// Device helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ivfhnsw_gpu_impl {

// Exact reference distance (hnswlib/hnswalg.cpp:326-357 == utils.cpp:22-52, AVX branch): 8 accumulators
// over blocks of 16 floats, unfused multiply then add, accumulators summed left to right.
// `row` is this lane's vector in global memory, `sq` the query in LDS (uniform address -> broadcast).
__device__ __forceinline__ float l2_ref_order(const float *__restrict__ row, const float *sq, int d)
{
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
    const float4 *q4 = reinterpret_cast<const float4 *>(sq);
    for (int b = 0; b < d / 16; b++) {
        const float4 y0 = r4[4 * b], y1 = r4[4 * b + 1], y2 = r4[4 * b + 2], y3 = r4[4 * b + 3];
        const float4 x0 = q4[4 * b], x1 = q4[4 * b + 1], x2 = q4[4 * b + 2], x3 = q4[4 * b + 3];
        float t;
#define IVFHNSW_ACC(a, xv, yv) \
    t = __fsub_rn(xv, yv);     \
    a = __fadd_rn(a, __fmul_rn(t, t));
        IVFHNSW_ACC(a0, x0.x, y0.x) IVFHNSW_ACC(a1, x0.y, y0.y) IVFHNSW_ACC(a2, x0.z, y0.z) IVFHNSW_ACC(a3, x0.w, y0.w)
        IVFHNSW_ACC(a4, x1.x, y1.x) IVFHNSW_ACC(a5, x1.y, y1.y) IVFHNSW_ACC(a6, x1.z, y1.z) IVFHNSW_ACC(a7, x1.w, y1.w)
        IVFHNSW_ACC(a0, x2.x, y2.x) IVFHNSW_ACC(a1, x2.y, y2.y) IVFHNSW_ACC(a2, x2.z, y2.z) IVFHNSW_ACC(a3, x2.w, y2.w)
        IVFHNSW_ACC(a4, x3.x, y3.x) IVFHNSW_ACC(a5, x3.y, y3.y) IVFHNSW_ACC(a6, x3.z, y3.z) IVFHNSW_ACC(a7, x3.w, y3.w)
#undef IVFHNSW_ACC
    }
    float r = __fadd_rn(a0, a1);
    r = __fadd_rn(r, a2);
    r = __fadd_rn(r, a3);
    r = __fadd_rn(r, a4);
    r = __fadd_rn(r, a5);
    r = __fadd_rn(r, a6);
    r = __fadd_rn(r, a7);
    return r;
}

// position of the r-th (0-based) set bit of mask; r < popcount(mask)
__device__ __forceinline__ int nth_set_bit(unsigned long long mask, int r)
{
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int c = __popcll(mask & (((1ull << w) - 1ull) << pos));
        if (c <= r) {
            r -= c;
            pos += w;
        }
    }
    return pos;
}

// The next eight set bits of a WAVE-UNIFORM mask, consumed from it: lane group g = lane >> 3 gets the position of
// the g-th of them (0 where the mask ran out).  All on the scalar unit (s_ff1 / s_lshl / s_or per bit) plus one
// 64-bit shift per lane, where nth_set_bit costs ~54 vector instructions for a per-lane rank.
__device__ __forceinline__ int take8_set_bits(unsigned long long &m, int lane)
{
    // (a sentinel at bit 63 keeps the count-trailing-zeros defined: an exhausted mask yields position 63, which
    // only lanes without a row ever see)
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t p = (uint32_t)__builtin_ctzll(m | (1ull << 63));
        lo |= p << (8 * j);
        m &= ~(1ull << p);
    }
    if (m) { // wave-uniform: a pass usually has fewer than five rows
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t p = (uint32_t)__builtin_ctzll(m | (1ull << 63));
            hi |= p << (8 * j);
            m &= ~(1ull << p);
        }
    }
    const uint32_t w = (lane & 32) ? hi : lo;
    return (int)((w >> (8 * ((lane >> 3) & 3))) & 0xffu);
}

// broadcast lane k of each quad to the whole quad (DPP quad_perm, no LDS)
template <int K> __device__ __forceinline__ float quad_bcast(float v)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), K * 0x55, 0xf, 0xf, true));
}

// sum over each aligned group of 8 lanes, left in all 8 (DPP: quad_perm swaps, then row_half_mirror; no LDS)
__device__ __forceinline__ int oct_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); // row_half_mirror: lane i <-> 7 - i
    return v;
}

// Four values per lane, summed over each aligned group of 8 lanes as a reduce-scatter: lane 8G + ii returns the
// group's sum of value 2 * bit2(ii) + bit0(ii) (lanes ii and ii ^ 2 hold the same one).  4 DPP adds + 6 selects.
__device__ __forceinline__ int oct_sum4(const int (&S)[4], int lane)
{
    const bool b2 = (lane & 4) != 0, b0 = (lane & 1) != 0;
    // partner 7 - i has the opposite bit 2: keep the pair {2 b2, 2 b2 + 1}, hand over the other
    const int k0 = b2 ? S[2] : S[0], k1 = b2 ? S[3] : S[1];
    const int h0 = b2 ? S[0] : S[2], h1 = b2 ? S[1] : S[3];
    const int a0 = k0 + __builtin_amdgcn_update_dpp(0, h0, 0x141, 0xf, 0xf, true); // row_half_mirror
    const int a1 = k1 + __builtin_amdgcn_update_dpp(0, h1, 0x141, 0xf, 0xf, true);
    // partner i ^ 1: keep the one bit 0 names
    const int k = b0 ? a1 : a0, h = b0 ? a0 : a1;
    int c = k + __builtin_amdgcn_update_dpp(0, h, 0xB1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
    c += __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xf, 0xf, true);        // quad_perm [2,3,0,1]: same value index there
    return c;
}

// oct_sum4 of two value sets at once, step by step in turn: each DPP add of one chain fills the wait states the
// other chain's DPP operand needs (the compiler pads a lone chain with s_nop).
__device__ __forceinline__ void oct_sum4x2(const int (&S)[4], const int (&X)[4], int lane, int &s_out, int &x_out)
{
    const bool b2 = (lane & 4) != 0, b0 = (lane & 1) != 0;
    const int sk0 = b2 ? S[2] : S[0], sk1 = b2 ? S[3] : S[1], sh0 = b2 ? S[0] : S[2], sh1 = b2 ? S[1] : S[3];
    const int xk0 = b2 ? X[2] : X[0], xk1 = b2 ? X[3] : X[1], xh0 = b2 ? X[0] : X[2], xh1 = b2 ? X[1] : X[3];
    const int sa0 = sk0 + __builtin_amdgcn_update_dpp(0, sh0, 0x141, 0xf, 0xf, true);
    const int xa0 = xk0 + __builtin_amdgcn_update_dpp(0, xh0, 0x141, 0xf, 0xf, true);
    const int sa1 = sk1 + __builtin_amdgcn_update_dpp(0, sh1, 0x141, 0xf, 0xf, true);
    const int xa1 = xk1 + __builtin_amdgcn_update_dpp(0, xh1, 0x141, 0xf, 0xf, true);
    const int sk = b0 ? sa1 : sa0, sh = b0 ? sa0 : sa1;
    const int xk = b0 ? xa1 : xa0, xh = b0 ? xa0 : xa1;
    int sc = sk + __builtin_amdgcn_update_dpp(0, sh, 0xB1, 0xf, 0xf, true);
    int xc = xk + __builtin_amdgcn_update_dpp(0, xh, 0xB1, 0xf, 0xf, true);
    sc += __builtin_amdgcn_update_dpp(0, sc, 0x4E, 0xf, 0xf, true);
    xc += __builtin_amdgcn_update_dpp(0, xc, 0x4E, 0xf, 0xf, true);
    s_out = sc;
    x_out = xc;
}

// Exact reference distance (hnswalg.cpp:326-357) of one row evaluated by a QUAD of lanes: lane t of the quad
// owns accumulators 2t and 2t+1 of the reference's eight, i.e. dims 8j+2t, 8j+2t+1 for j = 0..d/8-1 in
// increasing order -- exactly the order each __m256 lane accumulates in.  All d/8 8-byte loads of a lane are
// independent (one round trip per row); the quad then sums the eight accumulators left to right.
// Every lane of the quad returns the distance.
__device__ __forceinline__ float l2_ref_order_quad(const float *__restrict__ row, const float *sq, int d, int t)
{
    const float2 *r2 = reinterpret_cast<const float2 *>(row) + t;
    const float2 *q2 = reinterpret_cast<const float2 *>(sq) + t;
    float alo = 0.f, ahi = 0.f;
    const int nj = d >> 3;
#pragma unroll 16
    for (int j = 0; j < nj; j++) {
        const float2 y = r2[4 * j];
        const float2 x = q2[4 * j];
        const float d0 = __fsub_rn(x.x, y.x), d1 = __fsub_rn(x.y, y.y);
        alo = __fadd_rn(alo, __fmul_rn(d0, d0));
        ahi = __fadd_rn(ahi, __fmul_rn(d1, d1));
    }
    float r = __fadd_rn(quad_bcast<0>(alo), quad_bcast<0>(ahi));
    r = __fadd_rn(r, quad_bcast<1>(alo));
    r = __fadd_rn(r, quad_bcast<1>(ahi));
    r = __fadd_rn(r, quad_bcast<2>(alo));
    r = __fadd_rn(r, quad_bcast<2>(ahi));
    r = __fadd_rn(r, quad_bcast<3>(alo));
    r = __fadd_rn(r, quad_bcast<3>(ahi));
    return r;
}

// NB consecutive 8-float steps of the same sum with all NB row loads issued before the first use: the
// compiler barrier pins the order "loads, then arithmetic", which the scheduler otherwise gives up under
// register pressure (measured: the walk went from 1.7 to 2.7 ms when it serialised the loads of a row).
template <int NB>
__device__ __forceinline__ void l2_quad_block(const float2 *r2, const float2 *q2, int j0, float &alo, float &ahi)
{
    float2 y[NB];
#pragma unroll
    for (int i = 0; i < NB; i++)
        y[i] = r2[4 * (j0 + i)];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const float2 x = q2[4 * (j0 + i)];
        const float d0 = __fsub_rn(x.x, y[i].x), d1 = __fsub_rn(x.y, y[i].y);
        alo = __fadd_rn(alo, __fmul_rn(d0, d0));
        ahi = __fadd_rn(ahi, __fmul_rn(d1, d1));
    }
}

// l2_ref_order_quad with the loads of up to 16 steps (a 128-float row) in flight together
__device__ __forceinline__ float l2_ref_order_quad_batched(const float *row, const float *sq, int d, int t)
{
    const float2 *r2 = reinterpret_cast<const float2 *>(row) + t;
    const float2 *q2 = reinterpret_cast<const float2 *>(sq) + t;
    float alo = 0.f, ahi = 0.f;
    const int nj = d >> 3; // even: d is a multiple of 16
    int j0 = 0;
    for (; j0 + 16 <= nj; j0 += 16)
        l2_quad_block<16>(r2, q2, j0, alo, ahi);
    if (j0 + 8 <= nj) {
        l2_quad_block<8>(r2, q2, j0, alo, ahi);
        j0 += 8;
    }
    if (j0 + 4 <= nj) {
        l2_quad_block<4>(r2, q2, j0, alo, ahi);
        j0 += 4;
    }
    if (j0 + 2 <= nj)
        l2_quad_block<2>(r2, q2, j0, alo, ahi);
    float r = __fadd_rn(quad_bcast<0>(alo), quad_bcast<0>(ahi));
    r = __fadd_rn(r, quad_bcast<1>(alo));
    r = __fadd_rn(r, quad_bcast<1>(ahi));
    r = __fadd_rn(r, quad_bcast<2>(alo));
    r = __fadd_rn(r, quad_bcast<2>(ahi));
    r = __fadd_rn(r, quad_bcast<3>(alo));
    r = __fadd_rn(r, quad_bcast<3>(ahi));
    return r;
}

// The same distance by EIGHT lanes per row: lane t owns accumulator t of the reference's eight (dims 8j + t in
// increasing j).  Half the registers and half the arithmetic instructions of the quad form per pass, for 8
// rows per pass instead of 16 -- the walk's filter leaves ~5 rows per expansion.  The result is valid in the
// first four lanes of each group of eight.
template <int NB>
__device__ __forceinline__ void l2_oct_block(const float *r1, const float *q1, int j0, float &acc)
{
    float y[NB];
#pragma unroll
    for (int i = 0; i < NB; i++)
        y[i] = r1[8 * (j0 + i)];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const float d0 = __fsub_rn(q1[8 * (j0 + i)], y[i]);
        acc = __fadd_rn(acc, __fmul_rn(d0, d0));
    }
}

__device__ __forceinline__ float l2_ref_order_oct(const float *row, const float *sq, int d, int t)
{
    const float *r1 = row + t;
    const float *q1 = sq + t;
    float acc = 0.f;
    const int nj = d >> 3; // even: d is a multiple of 16
    int j0 = 0;
    for (; j0 + 16 <= nj; j0 += 16)
        l2_oct_block<16>(r1, q1, j0, acc);
    if (j0 + 8 <= nj) {
        l2_oct_block<8>(r1, q1, j0, acc);
        j0 += 8;
    }
    if (j0 + 4 <= nj) {
        l2_oct_block<4>(r1, q1, j0, acc);
        j0 += 4;
    }
    if (j0 + 2 <= nj)
        l2_oct_block<2>(r1, q1, j0, acc);
    // lanes 0-3 of the group: a0..a3 are in the own quad, a4..a7 arrive mirrored (lane i <-> 7 - i)
    const float mir = __uint_as_float(
        (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(acc), 0x141, 0xf, 0xf, true));
    float r = __fadd_rn(quad_bcast<0>(acc), quad_bcast<1>(acc));
    r = __fadd_rn(r, quad_bcast<2>(acc));
    r = __fadd_rn(r, quad_bcast<3>(acc));
    r = __fadd_rn(r, quad_bcast<3>(mir));
    r = __fadd_rn(r, quad_bcast<2>(mir));
    r = __fadd_rn(r, quad_bcast<1>(mir));
    r = __fadd_rn(r, quad_bcast<0>(mir));
    return r;
}

// d = 8 * NJ (128: SIFT, 96: DEEP) with the lane's NJ query components (q[8j + t]) held in registers for the whole
// query instead of being read from LDS for every pass: same loads, same order, same sum.
template <int NJ>
__device__ __forceinline__ float l2_ref_order_oct_regs(const float *row, const float (&qr)[16], int t)
{
    const float *r1 = row + t;
    float y[NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++)
        y[i] = r1[8 * i];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; i++) {
        const float d0 = __fsub_rn(qr[i], y[i]);
        acc = __fadd_rn(acc, __fmul_rn(d0, d0));
    }
    const float mir = __uint_as_float(
        (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(acc), 0x141, 0xf, 0xf, true));
    float r = __fadd_rn(quad_bcast<0>(acc), quad_bcast<1>(acc));
    r = __fadd_rn(r, quad_bcast<2>(acc));
    r = __fadd_rn(r, quad_bcast<3>(acc));
    r = __fadd_rn(r, quad_bcast<3>(mir));
    r = __fadd_rn(r, quad_bcast<2>(mir));
    r = __fadd_rn(r, quad_bcast<1>(mir));
    r = __fadd_rn(r, quad_bcast<0>(mir));
    return r;
}

// the reference's left-to-right sum of its eight accumulators, held two per lane by a quad (lane t: 2t, 2t+1)
__device__ __forceinline__ float quad_sum8(float alo, float ahi)
{
    float r = __fadd_rn(quad_bcast<0>(alo), quad_bcast<0>(ahi));
    r = __fadd_rn(r, quad_bcast<1>(alo));
    r = __fadd_rn(r, quad_bcast<1>(ahi));
    r = __fadd_rn(r, quad_bcast<2>(alo));
    r = __fadd_rn(r, quad_bcast<2>(ahi));
    r = __fadd_rn(r, quad_bcast<3>(alo));
    r = __fadd_rn(r, quad_bcast<3>(ahi));
    return r;
}

// inclusive prefix sum over the 64 lanes of a wavefront: four row_shr steps inside every row of 16 lanes, then
// row_bcast:15 / row_bcast:31 carry the rows' totals across (six DPP adds; the __shfl_up form was six ds_bpermute round
// trips through the LDS crossbar -- ~100 cycles each for a lone wavefront, and the Grouping plan's serial pass takes two
// scans per row)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    (void)lane;
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
    return (uint32_t)x;
}

// order-preserving map f32 -> u32 (and back) for packed (distance, scan position) keys
__device__ __forceinline__ uint32_t f32_orderable(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}


template <int CS>
__device__ __forceinline__ void load_code_words(const uint8_t *__restrict__ codes, uint32_t gi, uint32_t (&w)[CS / 4])
{
    const uint8_t *p = codes + (size_t)gi * CS;
    if constexpr (CS % 16 == 0) {
#pragma unroll
        for (int i = 0; i < CS / 16; i++) {
            uint4 v = reinterpret_cast<const uint4 *>(p)[i];
            w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
        }
    } else if constexpr (CS % 8 == 0) {
#pragma unroll
        for (int i = 0; i < CS / 8; i++) {
            uint2 v = reinterpret_cast<const uint2 *>(p)[i];
            w[2 * i] = v.x, w[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CS / 4; i++)
            w[i] = reinterpret_cast<const uint32_t *>(p)[i];
    }
}

// IndexIVF_HNSW.cpp:802-814: result starts at 0 and adds table entries for m = 0..CS-1 in order.
template <int CS>
__device__ __forceinline__ float adc_sum(const float *s_lut, const uint32_t (&w)[CS / 4])
{
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < CS; m++) {
        const uint32_t b = (w[m >> 2] >> ((m & 3) * 8)) & 0xffu;
        sum = __fadd_rn(sum, s_lut[m * 256 + b]);
    }
    return sum;
}

// faiss's SSE fvec_inner_product (the order of pq->compute_inner_prod_table, IndexIVF_HNSW.cpp:262): 4 partial sums
// over blocks of 4, zero-padded tail, then (s0+s1)+(s2+s3).
template <int DSUB>
__device__ __forceinline__ float ip_sse_order(const float *x, const float *y, int dsub_rt)
{
    const int dsub = DSUB > 0 ? DSUB : dsub_rt;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
#pragma unroll
    for (; i + 4 <= dsub; i += 4) {
        s0 = __fadd_rn(s0, __fmul_rn(x[i], y[i]));
        s1 = __fadd_rn(s1, __fmul_rn(x[i + 1], y[i + 1]));
        s2 = __fadd_rn(s2, __fmul_rn(x[i + 2], y[i + 2]));
        s3 = __fadd_rn(s3, __fmul_rn(x[i + 3], y[i + 3]));
    }
    if (i < dsub)
        s0 = __fadd_rn(s0, __fmul_rn(x[i], y[i]));
    if (i + 1 < dsub)
        s1 = __fadd_rn(s1, __fmul_rn(x[i + 1], y[i + 1]));
    if (i + 2 < dsub)
        s2 = __fadd_rn(s2, __fmul_rn(x[i + 2], y[i + 2]));
    return __fadd_rn(__fadd_rn(s0, s1), __fadd_rn(s2, s3));
}

// One code's bytes between its load and its use.  CS > 0: the code words stay in registers (all loads of an unrolled
// step are issued before the first table lookup).  CS == 0 is the run-time form for code sizes without an
// instantiation of their own (any multiple of 4, IndexIVF_HNSW.cpp:805): the sum is taken word by word as the code
// is read, in the same m order.
template <int CS> struct CodeRegs {
    uint32_t w[CS / 4];
};
template <> struct CodeRegs<0> {
    float sum;
};

template <int CS>
__device__ __forceinline__ void code_fetch(const uint8_t *__restrict__ codes, uint32_t gi, int cs_rt, const float *s_lut,
                                           CodeRegs<CS> &r)
{
    if constexpr (CS > 0) {
        load_code_words<CS>(codes, gi, r.w);
    } else {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(codes + (size_t)gi * cs_rt);
        float sum = 0.0f;
        for (int j = 0; j < cs_rt / 4; j++) {
            const uint32_t w = p[j];
            const float *t = s_lut + j * 1024;
            sum = __fadd_rn(sum, t[w & 0xffu]);
            sum = __fadd_rn(sum, t[256 + ((w >> 8) & 0xffu)]);
            sum = __fadd_rn(sum, t[512 + ((w >> 16) & 0xffu)]);
            sum = __fadd_rn(sum, t[768 + (w >> 24)]);
        }
        r.sum = sum;
    }
}

template <int CS> __device__ __forceinline__ float code_sum(const float *s_lut, const CodeRegs<CS> &r)
{
    if constexpr (CS > 0)
        return adc_sum<CS>(s_lut, r.w);
    else
        return r.sum;
}

} // namespace ivfhnsw_gpu_impl
