// placeholder: Grouping plan kernel (written next)
#include "ivfhnsw_kernels.h"
namespace ivfhnsw_gpu_impl {
hipError_t launch_plan_grouping(hipStream_t, const IvfTables &, const GroupTables &, const GraphTables &,
                                const float *, const uint32_t *, const float *, int, int, uint64_t, int, Seg *,
                                uint32_t *, PlanHdr *, int, uint64_t *, int, float *) { return hipErrorNotSupported; }
}
