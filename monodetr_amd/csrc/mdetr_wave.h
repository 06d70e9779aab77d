// monodetr_amd/csrc/mdetr_wave.h -- the wave-level vocabulary of the MFMA kernels written after round 1's GPU budget:
// bf16 / fp32 fragment vector types, the 32x32x16 bf16 matrix instruction, the wave-private LDS hand-over barrier
// and the dynamic-LDS declaration.  Included as <mdetr_wave.h> so that the CPU emulation build
// (tests/native/hipshim, tests/native_emul.py) can substitute its own implementation of exactly these pieces and
// run the unmodified kernel source on the host.
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// D = A B + C on one wave: lane l supplies A[l & 31][8 (l >> 5) + 0..7] and B[8 (l >> 5) + 0..7][l & 31];
// accumulator register r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]  (validated in attn.hip)
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// D = A B + C on one wave with fp32 operands (v_mfma_f32_32x32x2_f32): lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31] -- ONE
// value each, a contraction of depth 2 per instruction; the accumulator layout is mfma_bf16's.  Exact fp32: the result is the
// k-ordered fmaf chain (MI355X_MICROARCH.md, "Matrix cores"), at the fp32 vector rate (1/16 of the bf16 instruction).
__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// A pointer that went through LDS or a register shuffle has lost its address space: the compiler then emits FLAT loads, which count
// on both memory counters and can only be waited for all at once -- a software pipeline of such loads degenerates into
// load-wait-use.  `as_global<T>(p)` states that p points into global memory (global_load / global_store, vmcnt only).
#define MDETR_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const MDETR_GLOBAL T *as_global(const void *p) { return (const MDETR_GLOBAL T *)(p); }
template <typename T> __device__ __forceinline__ MDETR_GLOBAL T *as_global_rw(void *p) { return (MDETR_GLOBAL T *)(p); }

// a value every lane of the wave holds identically, moved to a scalar register (v_readfirstlane): what was read from LDS or memory
// per lane but is uniform by construction (descriptors, strides) then costs no vector register and feeds scalar address arithmetic
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t wave_uniform64(int64_t v)
{
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)));
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

// two packed fp32 lanes: element-wise fma / mul on this type select v_pk_fma_f32 / v_pk_mul_f32 (one issue slot for two
// channels; the plain-float spelling compiles to two scalar FMAs)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;      // 16 bytes as one register quad (arrays of it stay in registers)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ f32x2 make_f32x2(float x, float y) { f32x2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 mul2(f32x2 a, f32x2 b) { return a * b; }
__device__ __forceinline__ f32x2 add2(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 sub2(f32x2 a, f32x2 b) { return a - b; }

// c + a.lo * b.lo + a.hi * b.hi on two PACKED bf16 pairs (the 32-bit words as they sit in memory), fp32 accumulate:
// v_dot2c_f32_bf16 -- a dot product over bf16 data without widening either operand first
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}

// ds_read_b64_tr_b16: the transposing LDS read.  Every lane passes the (8-byte aligned) address of 4 consecutive bf16; within a
// 16-lane group, lane i receives element i & 3 of the vectors that lanes (i >> 2), 4 + (i >> 2), 8 + (i >> 2), 12 + (i >> 2) point at
// (measured: scripts/exp/ds_read_tr16_probe.hip).  With lane s pointing at row s >> 2, columns 4 (s & 3) .. + 3 of a row-major tile,
// lane i gets column i of rows 0 .. 3: four consecutive CONTRACTION values of its own column -- half an MFMA operand.
__device__ __forceinline__ bf16x4 lds_read_tr4(const __bf16 *p)
{
    typedef bf16x4 __attribute__((address_space(3))) lds_bf16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p));
}

// orders a wave's LDS writes before its subsequent LDS reads (wave-private buffers: no workgroup barrier needed)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Buffer-resource loads: a load whose per-lane offset lies beyond the resource's size returns ZERO instead of faulting -- the
// image borders (convolution padding) and ragged tiles of the convolution kernels cost a select on the offset, not a branch.
// The range check covers the per-lane offset only (the scalar offset is added afterwards): an invalid lane passes kRsrcOob.
typedef __amdgpu_buffer_rsrc_t mdetr_rsrc;
constexpr unsigned kRsrcOob = 0xfffffff0u;
__device__ __forceinline__ mdetr_rsrc make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);      // raw buffer, 32-bit data format
}
__device__ __forceinline__ bf16x8 rsrc_load_bf16x8(mdetr_rsrc r, unsigned lane_offset, unsigned scalar_offset)
{
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_offset, scalar_offset, 0));
}
__device__ __forceinline__ unsigned short rsrc_load_u16(mdetr_rsrc r, unsigned lane_offset, unsigned scalar_offset)
{
    return __builtin_amdgcn_raw_buffer_load_b16(r, lane_offset, scalar_offset, 0);
}

// ... and stores: a lane whose offset lies beyond the resource's size stores nothing (ragged tiles without a branch)
__device__ __forceinline__ void rsrc_store_bf16x8(mdetr_rsrc r, bf16x8 v, unsigned lane_offset, unsigned scalar_offset)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_offset, scalar_offset, 0);
}
__device__ __forceinline__ void rsrc_store_f32x4(mdetr_rsrc r, f32x4 v, unsigned lane_offset, unsigned scalar_offset)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_offset, scalar_offset, 0);
}

#define MDETR_DYNAMIC_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
