// monodetr_amd/csrc/colsum.h -- internal launcher declarations (see colsum.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/monodetr_amd.h"

namespace mdetr {

// dtype: 0 = f32, 2 = bf16 (MDETR_F32 / MDETR_BF16 of include/monodetr_amd.h)
bool colsum_supported(int dtype, int cols, int64_t ld, const void *x);
int64_t colsum_workspace_bytes(int64_t rows, int cols);
// out_dtype: 0 = fp32 `out`, 2 = bf16 `out` (the fp32 sum rounded once)
hipError_t colsum_launch(int dtype, const void *x, void *out, void *workspace, int64_t rows, int cols, int64_t ld,
                         hipStream_t st, int out_dtype = 0);

// grouped chunk sums (mdetr_chunk_sums): message or nullptr; any number of jobs (launched kChunkJobs at a time)
const char *chunk_sums_check(const mdetr_chunk_job *jobs, int njobs);
hipError_t chunk_sums_launch(const mdetr_chunk_job *jobs, int njobs, hipStream_t st);

}  // namespace mdetr
