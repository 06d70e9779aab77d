// tests/native/hipshim/hip/hip_bf16.h -- TEST INFRASTRUCTURE (see hip_runtime.h): bf16 storage type with
// round-to-nearest-even conversion, as __float2bfloat16 / __bfloat162float.
#pragma once
#include <stdint.h>
#include <string.h>

struct __hip_bfloat16 {
    uint16_t bits;
    __hip_bfloat16() = default;
    __hip_bfloat16(float f);                // round to nearest even, as the device type's converting constructor
    operator float() const;
};

inline __hip_bfloat16 __float2bfloat16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    __hip_bfloat16 h;
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) { h.bits = static_cast<uint16_t>((u >> 16) | 0x40); return h; }
    u += 0x7FFFu + ((u >> 16) & 1u);
    h.bits = static_cast<uint16_t>(u >> 16);
    return h;
}
inline float __bfloat162float(__hip_bfloat16 h)
{
    const uint32_t u = static_cast<uint32_t>(h.bits) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline __hip_bfloat16::__hip_bfloat16(float f) { bits = __float2bfloat16(f).bits; }
inline __hip_bfloat16::operator float() const { return __bfloat162float(*this); }
