// tests/native/hipshim/mdetr_wave.h -- TEST INFRASTRUCTURE: CPU stand-in for monodetr_amd/csrc/mdetr_wave.h (found
// first on the include path of the emulation build).  Fragment vectors are small structs; the matrix instruction is
// emulated as a wave-collective exchange with the operand / accumulator layout documented (and GPU-validated through
// attn.hip) in the real header: products of bf16 values are exact in fp32, accumulation is fp32 in k order.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#define __bf16 hipshim_bf16
struct hipshim_bf16 {
    uint16_t bits;
    hipshim_bf16() = default;
    explicit hipshim_bf16(float f) : bits(__float2bfloat16(f).bits) {}
    explicit operator float() const { __hip_bfloat16 h; h.bits = bits; return __bfloat162float(h); }
};

template <typename T, int N> struct alignas(sizeof(T) * N) hipshim_vec {
    T v[N];
    T &operator[](int i) { return v[i]; }
    const T &operator[](int i) const { return v[i]; }
};
typedef hipshim_vec<__bf16, 8> bf16x8;
typedef hipshim_vec<__bf16, 4> bf16x4;
struct f32x16 {
    float v[16];
    float &operator[](int i) { return v[i]; }
    const float &operator[](int i) const { return v[i]; }
};

namespace hipshim {
extern float mfma_a[1024][8], mfma_b[1024][8];
unsigned char *dynamic_lds();
}  // namespace hipshim

inline f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c)
{
    const int me = threadIdx.x, base = hipshim::lane_base(), lane = me - base;
    for (int i = 0; i < 8; ++i) {
        hipshim::mfma_a[me][i] = static_cast<float>(a[i]);
        hipshim::mfma_b[me][i] = static_cast<float>(b[i]);
    }
    hipshim::sync_wave();
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc += hipshim::mfma_a[base + row + 32 * (k >> 3)][k & 7] * hipshim::mfma_b[base + col + 32 * (k >> 3)][k & 7];
        c[r] = acc;
    }
    hipshim::sync_wave();
    return c;
}

// v_mfma_f32_32x32x2_f32: lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31]; D = the k-ordered fmaf chain (exact fp32)
inline f32x16 mfma_f32(float a, float b, f32x16 c)
{
    const int me = threadIdx.x, base = hipshim::lane_base(), lane = me - base;
    hipshim::mfma_a[me][0] = a;
    hipshim::mfma_b[me][0] = b;
    hipshim::sync_wave();
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k)
            acc = fmaf(hipshim::mfma_a[base + row + 32 * k][0], hipshim::mfma_b[base + col + 32 * k][0], acc);
        c[r] = acc;
    }
    hipshim::sync_wave();
    return c;
}

// address-space statements are for the GPU compiler only
#define MDETR_GLOBAL
template <typename T> inline const T *as_global(const void *p) { return static_cast<const T *>(p); }
template <typename T> inline T *as_global_rw(void *p) { return static_cast<T *>(p); }

// v_readfirstlane of a value that is uniform by construction: the value itself
inline int wave_uniform(int v) { return v; }
inline int64_t wave_uniform64(int64_t v) { return v; }

// two packed fp32 lanes (the real header: ext_vector_type(2) -> v_pk_fma_f32 / v_pk_mul_f32)
struct f32x2 { float x, y; };
struct alignas(16) f32x4 { float x, y, z, w; };
struct alignas(16) u32x4 { unsigned x, y, z, w; };
struct alignas(8) u32x2 { unsigned x, y; };
inline f32x2 make_f32x2(float x, float y) { return {x, y}; }
inline f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
inline f32x2 mul2(f32x2 a, f32x2 b) { return {a.x * b.x, a.y * b.y}; }
inline f32x2 add2(f32x2 a, f32x2 b) { return {a.x + b.x, a.y + b.y}; }
inline f32x2 sub2(f32x2 a, f32x2 b) { return {a.x - b.x, a.y - b.y}; }

// v_dot2c_f32_bf16 (products of bf16 values are exact in fp32; the order of the two additions is the emulation's choice)
inline float dot2_bf16(unsigned a, unsigned b, float c)
{
    return fmaf(__uint_as_float(a & 0xFFFF0000u), __uint_as_float(b & 0xFFFF0000u), fmaf(__uint_as_float(a << 16), __uint_as_float(b << 16), c));
}

// buffer-resource loads (the real header: raw buffer loads that return zero beyond the resource's size; the range check covers
// the per-lane offset only)
struct mdetr_rsrc { const unsigned char *base; unsigned bytes; };
constexpr unsigned kRsrcOob = 0xfffffff0u;
inline mdetr_rsrc make_rsrc(const void *base, unsigned bytes) { return {static_cast<const unsigned char *>(base), bytes}; }
inline bf16x8 rsrc_load_bf16x8(mdetr_rsrc r, unsigned lane_offset, unsigned scalar_offset)
{
    bf16x8 v;
    if (static_cast<unsigned long long>(lane_offset) + 16 > r.bytes) {
        for (int i = 0; i < 8; ++i) v[i].bits = 0;
        return v;
    }
    if (static_cast<unsigned long long>(lane_offset) + scalar_offset + 16 > r.bytes) abort();     // a valid lane must address the tensor
    memcpy(&v, r.base + lane_offset + scalar_offset, 16);
    return v;
}

inline unsigned short rsrc_load_u16(mdetr_rsrc r, unsigned lane_offset, unsigned scalar_offset)
{
    unsigned short v = 0;
    if (static_cast<unsigned long long>(lane_offset) + 2 > r.bytes) return 0;
    if (static_cast<unsigned long long>(lane_offset) + scalar_offset + 2 > r.bytes) abort();
    memcpy(&v, r.base + lane_offset + scalar_offset, 2);
    return v;
}

// buffer-resource stores: a lane whose offset lies beyond the resource's size stores nothing
template <typename V> inline void rsrc_store16(mdetr_rsrc r, const V &v, unsigned lane_offset, unsigned scalar_offset)
{
    static_assert(sizeof(V) == 16, "16-byte store");
    if (static_cast<unsigned long long>(lane_offset) + 16 > r.bytes) return;
    if (static_cast<unsigned long long>(lane_offset) + scalar_offset + 16 > r.bytes) abort();      // a valid lane must address the tensor
    memcpy(const_cast<unsigned char *>(r.base) + lane_offset + scalar_offset, &v, 16);
}
inline void rsrc_store_bf16x8(mdetr_rsrc r, bf16x8 v, unsigned lane_offset, unsigned scalar_offset) { rsrc_store16(r, v, lane_offset, scalar_offset); }
inline void rsrc_store_f32x4(mdetr_rsrc r, f32x4 v, unsigned lane_offset, unsigned scalar_offset) { rsrc_store16(r, v, lane_offset, scalar_offset); }

// ds_read_b64_tr_b16 (the real header: the transposing LDS read, semantics measured on gfx950 with scripts/exp/ds_read_tr16_probe.hip):
// within a 16-lane group, lane i receives element i & 3 of the 4-element vectors that lanes (i >> 2) + 4 j, j = 0 .. 3, point at
inline bf16x4 lds_read_tr4(const __bf16 *p)
{
    const int me = threadIdx.x, base = hipshim::lane_base(), lane = me - base;
    static_assert(sizeof(bf16x4) == 8, "one exchange slot");
    if ((reinterpret_cast<uintptr_t>(p) & 7) != 0) abort();       // (misaligned: the hardware would read the aligned address's data)
    memcpy(&hipshim::exchange[me], p, 8);
    hipshim::sync_wave();
    bf16x4 r;
    const int grp = base + (lane & ~15), i = lane & 15;
    for (int j = 0; j < 4; ++j) {
        bf16x4 src;
        memcpy(&src, &hipshim::exchange[grp + 4 * j + (i >> 2)], 8);
        r[j] = src[i & 3];
    }
    hipshim::sync_wave();
    return r;
}

inline void wave_sync() { hipshim::sync_wave(); }

#define MDETR_DYNAMIC_LDS(type, name) type *name = reinterpret_cast<type *>(hipshim::dynamic_lds())
