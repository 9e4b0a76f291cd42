// LDS bank-behaviour probe for gfx950 (exploration tool, not part of the product): times ds_read_b32 gathers
// whose per-lane dword index follows a chosen pattern, one wavefront per CU-resident block.
//   build: hipcc -O3 --offload-arch=gfx950 tools/lds_probe.hip -o tools/lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void probe(const int *__restrict__ idx, int iters, float *out, int nwaves_active)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x)
        lds[i] = (float)(i & 255);
    __syncthreads();
    if ((int)(threadIdx.x >> 6) >= nwaves_active)
        return;
    int a = idx[threadIdx.x & 63];
    float acc = 0.f;
    volatile float *vl = lds; // every read is a real ds_read_b32
    for (int it = 0; it < iters; it++) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++)
            v[u] = vl[(a + u * 1024) & 16383]; // + multiples of 1024 dwords: same bank for every lane
#pragma unroll
        for (int u = 0; u < 16; u++)
            acc += v[u];
    }
    if (acc == -1.f)
        out[0] = acc;
}

static double run(const std::vector<int> &h, int iters, int nwaves)
{
    int *d;
    float *o;
    (void)hipMalloc(&d, 64 * sizeof(int));
    (void)hipMalloc(&o, 4);
    hipMemcpy(d, h.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<<<256, 1024>>>(d, 10, o, nwaves);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<<<256, 1024>>>(d, iters, o, nwaves);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    hipFree(o);
    // ns per wave-level ds_read instruction per CU (nwaves waves issue 16*iters each)
    return ms * 1e6 / ((double)iters * 16 * nwaves);
}

int main()
{
    const int iters = 20000;
    srand(1);
    auto pat = [&](const char *name, auto f) {
        std::vector<int> h(64);
        for (int l = 0; l < 64; l++)
            h[l] = f(l);
        printf("%-58s 4 waves: %5.2f   16 waves: %5.2f cycles/instr/CU (at 2.4 GHz)\n", name, run(h, iters, 4) * 2.4, run(h, iters, 16) * 2.4);
    };
    pat("all lanes same address (broadcast)", [](int) { return 5; });
    pat("lane l -> dword l (conflict free)", [](int l) { return l; });
    for (int s : {2, 4, 8, 16, 32, 64})
    {
        char nm[64];
        snprintf(nm, sizeof nm, "stride %d dwords", s);
        pat(nm, [s](int l) { return l * s; });
    }
    pat("l and l+32 same bank, else distinct (l%32 + 32*(l/32)*8)", [](int l) { return (l % 32) + 256 * (l / 32); });
    pat("l and l+16 same bank (l%16 + 16*... stride 32 rows)", [](int l) { return (l % 16) + 32 * (l / 16); });
    pat("l and l+8 same bank", [](int l) { return (l % 8) + 32 * (l / 8); });
    pat("l and l+4 same bank", [](int l) { return (l % 4) + 32 * (l / 4); });
    pat("pairs (2l, 2l+1) same bank", [](int l) { return (l / 2) + 32 * (l % 2); });
    pat("random dword in 256 (the PQ gather)", [](int) { return rand() % 256; });
    pat("random, lanes confined to 8-bank groups by (l%32)/8", [](int l) { int r = (l % 32) / 8; return (rand() % 32) * 32 + r * 8 + rand() % 8; });
    pat("random, lanes confined to 8-bank groups by l/8 (64 banks)", [](int l) { int r = l / 8; return (rand() % 32) * 64 + r * 8 + rand() % 8; });
    pat("random, lanes confined to 16-bank groups by l/16 (64 banks)", [](int l) { int r = l / 16; return (rand() % 32) * 64 + r * 16 + rand() % 16; });
    pat("random, 4-bank groups by l/4 (64 banks)", [](int l) { int r = l / 4; return (rand() % 32) * 64 + r * 4 + rand() % 4; });
    pat("random over 64 banks (dword in 64-wide rows)", [](int) { return (rand() % 32) * 64 + rand() % 64; });
    // the lane IS the sub-quantizer (systolic ADC): table m = l % 16 pinned to banks {m, m + 16} by the code's low bit
    pat("table per lane: 16*c + l%16, c random (2 lanes per bank pair)", [](int l) { return 16 * (rand() % 256) + (l % 16); });
    // ... and with a second query's tables in banks 16..31 (lanes 16..31 of each half): one lane per bank
    pat("table per lane: 32*c + l%32, c random (conflict free)", [](int l) { return 32 * (rand() % 256) + (l % 32); });
    return 0;
}
