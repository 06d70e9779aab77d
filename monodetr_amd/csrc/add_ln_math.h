// monodetr_amd/csrc/add_ln_math.h -- per-element pieces of y = LayerNorm(a + dropout(b)) shared by add_ln.hip and
// the tests: the stateless dropout decision for element (row, col) of a launch seeded with `seed`.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#define MDETR_HD inline
#endif

namespace mdetr {

// 32-bit mix of the element's flat index and the launch seed (two multiply-xorshift rounds: the kernel is
// memory-bound, the extra multiply is free); keep the element iff the hash is >= thresh = p * 2^32.
MDETR_HD uint32_t ln_hash(uint64_t seed, uint64_t index)
{
    uint64_t z = index * 0x9E3779B97F4A7C15ull + seed;
    uint32_t x = static_cast<uint32_t>(z) ^ static_cast<uint32_t>(z >> 32);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

MDETR_HD uint32_t ln_threshold(float p)
{
    const double t = static_cast<double>(p) * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : static_cast<uint32_t>(t);
}

}  // namespace mdetr
