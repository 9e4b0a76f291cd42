// Does hipStreamWaitValue32 release a stream on a value a KERNEL of another stream stores into signal memory, and how soon?
// (round 3: one walk launch that tells the first part's scan when its queries are done.)  Every wait here is released by a
// write that is already enqueued, so nothing can stay blocked.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(uint32_t *sig, uint32_t value, unsigned long long spin, unsigned long long *t_store)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(sig, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    *t_store = wall_clock64();
    // ... and keeps running, as the walk's tail does
    const unsigned long long t1 = wall_clock64();
    while (wall_clock64() - t1 < spin) { }
}
__global__ void consumer(unsigned long long *t_start) { *t_start = wall_clock64(); }

int main()
{
    int ok = 0;
    CK(hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", ok);
    if (!ok)
        return 0;
    uint32_t *sig = nullptr;
    CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    unsigned long long *ts = nullptr;
    CK(hipMalloc(&ts, 16));
    CK(hipMemset(ts, 0, 16));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    CK(hipStreamWriteValue32(a, sig, 0, 0));
    CK(hipStreamSynchronize(a));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    for (uint32_t epoch = 1; epoch <= 5; epoch++) {
        CK(hipStreamWaitValue32(b, sig, epoch, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, ts + 1);
        hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, a, sig, epoch, (unsigned long long)rate_khz * 1 /* 1 ms */, ts);
        CK(hipStreamWriteValue32(a, sig, epoch, 0)); // the fallback release behind the kernel
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        unsigned long long h[2];
        CK(hipMemcpy(h, ts, 16, hipMemcpyDeviceToHost));
        printf("epoch %u: consumer started %.1f us after the kernel's store (the producer ran on for 1 ms)\n", epoch,
               ((double)h[1] - (double)h[0]) / rate_khz * 1e3);
    }
    printf("ok\n");
    return 0;
}
