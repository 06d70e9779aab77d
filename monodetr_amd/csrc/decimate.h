// monodetr_amd/csrc/decimate.h -- internal launcher declarations (see decimate.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// pixel_bytes = C * element size, a multiple of 16; x [B, H, W, C] channels-last, y [B, (H + 1) / 2, (W + 1) / 2, C]
bool decimate2_supported(int64_t pixel_bytes, const void *x, const void *y);
hipError_t decimate2_forward_launch(const void *x, void *y, int B, int H, int W, int64_t pixel_bytes, hipStream_t st);
// dx [B, H, W, C]: dy at the even pixels, zero elsewhere (every byte of dx is written)
hipError_t decimate2_backward_launch(const void *dy, void *dx, int B, int H, int W, int64_t pixel_bytes, hipStream_t st);

// many dense tensors -> one flat buffer: block b copies bytes [blk_start[b], + chunk_bytes) of tensor blk_tensor[b]; the block tables and
// dst_off / nbytes live on the device, the source pointers and tensor_block_begin (first block of each tensor, ntensors + 1 entries) on the host
hipError_t gather_flat_launch(const void *const *src_host, int ntensors, const int *tensor_block_begin_host, void *dst, const int64_t *dst_off,
                              const int64_t *nbytes, const int *blk_tensor, const int64_t *blk_start, int chunk_bytes, hipStream_t st);

// 3x3 / stride 2 / pad 1 max pooling, bf16 channels-last, C a multiple of 8: x [B, H, W, C] -> y [B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, C]
hipError_t maxpool3x3s2_bf16_launch(const void *x, void *y, int B, int H, int W, int C, hipStream_t st);

}  // namespace mdetr
