// Calibration of rocprofv3's FETCH_SIZE for the access shapes of the HNSW walk (MI355X guide, HBM section: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Four kernels read RANDOM rows of
// a buffer far larger than the Infinity Cache, every row once, in exactly the shapes hnsw_walk_kernel uses:
//   links    one 128-byte row per wave instruction: 32 lanes x one dword            (links_c, nbnorms rows)
//   bytes    `cnt` x 128-byte rows in one piece: 64 lanes x 16 bytes per instruction (neighbour byte rows, cnt = 22)
//   floats   eight 512-byte rows per pass: 8 lanes per row, lane t reads x[8j + t]    (float rows of the survivors)
//   stream   64 lanes x 16 bytes, consecutive (the guide's calibrated case: FETCH_SIZE = bytes / 2)
// Run each under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; the program prints the bytes every kernel really read.
// usage: fetch_probe.bin [GiB of buffer, default 8]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// a bijection on [0, 2^bits): multiply by an odd constant, so that every row is read exactly once
__device__ __forceinline__ uint64_t perm(uint64_t i, int bits) { return (i * 0x9E3779B97F4A7C15ull + 0x1234567ull) & ((1ull << bits) - 1ull); }

__global__ __launch_bounds__(64) void probe_links(const uint32_t *buf, int bits, uint64_t rows_per_wave, uint32_t *sink)
{
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t r = 0; r < rows_per_wave; r++) {
        const uint64_t row = perm((uint64_t)blockIdx.x * rows_per_wave + r, bits); // 128-byte rows
        if (lane < 32)
            acc += buf[row * 32 + lane];
    }
    if (acc == 0x12345u)
        sink[0] = acc;
}

__global__ __launch_bounds__(64) void probe_bytes(const uint4 *buf, int bits, uint64_t recs_per_wave, int cnt, uint32_t *sink)
{
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t r = 0; r < recs_per_wave; r++) {
        const uint64_t rec = perm((uint64_t)blockIdx.x * recs_per_wave + r, bits); // 4-KB records (32 rows of 128 B)
        const uint4 *p = buf + rec * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = i * 8 + (lane >> 3);
            if (row < cnt) {
                const uint4 v = p[row * 8 + (lane & 7)];
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345u)
        sink[0] = acc;
}

__global__ __launch_bounds__(64) void probe_floats(const float *buf, int bits, uint64_t passes_per_wave, uint32_t *sink)
{
    const int lane = threadIdx.x;
    float acc = 0.f;
    for (uint64_t r = 0; r < passes_per_wave; r++) {
        // eight 512-byte rows per pass, each random
        const uint64_t row = perm(((uint64_t)blockIdx.x * passes_per_wave + r) * 8 + (lane >> 3), bits);
        const float *p = buf + row * 128 + (lane & 7);
        float y[16];
#pragma unroll
        for (int j = 0; j < 16; j++)
            y[j] = p[8 * j];
#pragma unroll
        for (int j = 0; j < 16; j++)
            acc += y[j];
    }
    if (acc == 12345.678f)
        sink[0] = 1;
}

__global__ __launch_bounds__(256) void probe_stream(const uint4 *buf, uint64_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u)
        sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int gib = argc > 1 ? atoi(argv[1]) : 8;
    int bits_bytes = 0;
    while ((1ull << (bits_bytes + 1)) <= (uint64_t)gib << 30)
        bits_bytes++;
    const uint64_t bytes = 1ull << bits_bytes;
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    const int waves = 256 * 16;
    // every kernel reads a QUARTER of its rows (a power of two of them, each once), spread over the whole buffer
    {
        const int bits = bits_bytes - 7; // 128-byte rows
        const uint64_t per = ((1ull << bits) / 4) / waves;
        hipLaunchKernelGGL(probe_links, dim3(waves), dim3(64), 0, 0, (const uint32_t *)buf, bits, per, sink);
        CK(hipDeviceSynchronize());
        printf("probe_links   read %.6f GB (%llu rows of 128 B, 32 lanes x dword)\n", per * waves * 128 / 1e9, (unsigned long long)(per * waves));
    }
    {
        const int bits = bits_bytes - 12; // 4-KB records
        const uint64_t per = ((1ull << bits) / 4) / waves;
        const int cnt = 22;
        hipLaunchKernelGGL(probe_bytes, dim3(waves), dim3(64), 0, 0, (const uint4 *)buf, bits, per, cnt, sink);
        CK(hipDeviceSynchronize());
        printf("probe_bytes   read %.6f GB (%llu records x %d rows of 128 B, 64 lanes x 16 B)\n", per * waves * cnt * 128 / 1e9, (unsigned long long)(per * waves), cnt);
    }
    {
        const int bits = bits_bytes - 9; // 512-byte rows
        const uint64_t per = ((1ull << bits) / 4) / 8 / waves;
        hipLaunchKernelGGL(probe_floats, dim3(waves), dim3(64), 0, 0, (const float *)buf, bits, per, sink);
        CK(hipDeviceSynchronize());
        printf("probe_floats  read %.6f GB (%llu rows of 512 B, 8 lanes per row, dword loads)\n", per * waves * 8 * 512 / 1e9, (unsigned long long)(per * waves * 8));
    }
    {
        const uint64_t n16 = bytes / 4 / 16;
        hipLaunchKernelGGL(probe_stream, dim3(256 * 8), dim3(256), 0, 0, (const uint4 *)buf, n16, sink);
        CK(hipDeviceSynchronize());
        printf("probe_stream  read %.6f GB (consecutive, 16 B per lane)\n", n16 * 16 / 1e9);
    }
    return 0;
}
