// monodetr_amd/csrc/kitti_prep.h -- internal launcher declaration (see kitti_prep.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/monodetr_amd.h"

namespace mdetr {

struct KittiNorm { float mean[3], stdv[3]; };

// out_dtype: 0 = f32, 2 = bf16
hipError_t kitti_prep_launch(const uint8_t *pixels, const MdetrKittiImage *images, int n_images, void *out,
                             int out_dtype, int out_h, int out_w, KittiNorm norm, int channels_last, hipStream_t st);

}  // namespace mdetr
