// placeholder: on-device HNSW walk (written next)
#include "ivfhnsw_kernels.h"
namespace ivfhnsw_gpu_impl {
hipError_t launch_coarse(hipStream_t, const GraphTables &, const float *, int, int, int, uint32_t *, float *,
                         uint32_t *, size_t, int) { return hipErrorNotSupported; }
}
