// placeholder: general-k scan (written next)
#include "ivfhnsw_kernels.h"
namespace ivfhnsw_gpu_impl {
hipError_t launch_scan_topk(hipStream_t, const IvfTables &, const float *, const Seg *, const uint32_t *,
                            const PlanHdr *, int, int, int, uint64_t *) { return hipErrorNotSupported; }
}
