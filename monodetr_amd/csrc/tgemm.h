// monodetr_amd/csrc/tgemm.h -- internal launcher declarations (see tgemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// flag bits (the C ABI's MDETR_TGEMM_* in include/monodetr_amd.h)
constexpr int kTgemmRelu = 1, kTgemmNN = 2, kTgemmBiasF32 = 4, kTgemmOutF32 = 8;

struct TgemmProblem {
    const void *a;            // bf16 [T, K], row stride lda
    const void *w;            // bf16: NT [N, K] (row stride ldw) -- y = a w^T;  NN [K, N] (row stride ldw) -- y = a w
    const void *bias;         // [N] bf16 (fp32 with kTgemmBiasF32) or null
    const void *res;          // bf16 [T, N] (row stride ldr) added before the ReLU, or null; may alias y (beta = 1 accumulation)
    void *y;                  // bf16 (fp32 with kTgemmOutF32) [T, N], row stride ldy
    int64_t T;
    int N, K;
    int64_t lda, ldw, ldr, ldy;
    int flags;
    float dropout_p;          // > 0: y = dropout(relu(.)) with the stateless hash of add_ln_math.h on the element index t * N + n
    uint64_t seed;
    const uint64_t *seed_dev; // added to seed when not null (a replayed graph's seed lives on the device)
    const void *mask = nullptr;   // bf16 [T, N] (row stride ldm) or null: y = mask <= 0 ? 0 : a w + res (NN form, bf16 output, no other tail)
    int64_t ldm = 0;
};

bool tgemm_supported(const TgemmProblem &p);
hipError_t tgemm_launch(const TgemmProblem &p, hipStream_t st);

}  // namespace mdetr
