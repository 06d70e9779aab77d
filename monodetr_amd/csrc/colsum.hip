// monodetr_amd/csrc/colsum.hip -- column sums of a tall row-major matrix (fp32 accumulation).
//
// The bias gradient of a token-wise linear layer is db[j] = sum_t dY[t, j] with T = 81 600 token rows
// and 256 .. 1024 columns (27 such reductions per MonoDETR training step).  The framework's generic
// reduction runs them at 0.2 - 0.6 TB/s (up to 260 us each); this is a pure HBM stream:
//   stage 1: a block owns kRowsPerBlock consecutive rows; a thread owns one 16-byte column vector and
//            every (256 / vectors-per-row)-th row, keeps VEC fp32 partial sums in registers, the row
//            lanes are combined through LDS, one fp32 partial row per block goes to the workspace;
//   stage 2: the partial rows are added in a fixed order (deterministic, no atomics).
// Algorithmic bytes = rows * cols * sizeof(T) (read once).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "colsum.h"

namespace mdetr {
namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 256;

template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void add(float *acc, const uint4 &v)
    {
        acc[0] += __uint_as_float(v.x); acc[1] += __uint_as_float(v.y);
        acc[2] += __uint_as_float(v.z); acc[3] += __uint_as_float(v.w);
    }
};
template <> struct Vec16<__hip_bfloat16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void add(float *acc, const uint4 &v)
    {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // bf16 -> fp32 is a 16-bit shift
            acc[2 * i] += __uint_as_float(w[i] << 16);
            acc[2 * i + 1] += __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
};

// CT = column vectors handled per block row pass (power of two <= 256); row lanes = 256 / CT
// PT = float: one partial row per block into the workspace; PT = bf16: only for a single row block, whose partial row IS
// the result (rounded once)
template <typename T, typename PT = float>
__global__ __launch_bounds__(kThreads)
void colsum_partial_kernel(const T *__restrict__ x, PT *__restrict__ partial, int64_t rows, int cols, int64_t ld, int CT)
{
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[kThreads * VEC];
    const int cvs = cols / VEC;
    const int lane_c = threadIdx.x % CT, lane_r = threadIdx.x / CT, RL = kThreads / CT;
    const int cv = blockIdx.y * CT + lane_c;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock;
    const int64_t r1 = r0 + kRowsPerBlock < rows ? r0 + kRowsPerBlock : rows;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (cv < cvs) {
        const T *p = x + static_cast<int64_t>(cv) * VEC;
        int64_t r = r0 + lane_r;
        for (; r + 3 * RL < r1; r += 4 * RL) {             // four independent loads in flight
            const uint4 a = *reinterpret_cast<const uint4 *>(p + r * ld);
            const uint4 b = *reinterpret_cast<const uint4 *>(p + (r + RL) * ld);
            const uint4 c = *reinterpret_cast<const uint4 *>(p + (r + 2 * RL) * ld);
            const uint4 d = *reinterpret_cast<const uint4 *>(p + (r + 3 * RL) * ld);
            Vec16<T>::add(acc, a); Vec16<T>::add(acc, b); Vec16<T>::add(acc, c); Vec16<T>::add(acc, d);
        }
        for (; r < r1; r += RL) Vec16<T>::add(acc, *reinterpret_cast<const uint4 *>(p + r * ld));
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(lane_r * CT + lane_c) * VEC + i] = acc[i];
    __syncthreads();
    // thread t < CT * VEC sums scalar column (t) of this tile over the row lanes, in lane order
    for (int t = threadIdx.x; t < CT * VEC; t += kThreads) {
        float s = 0.f;
        for (int l = 0; l < RL; ++l) s += red[l * CT * VEC + t];
        const int col = blockIdx.y * CT * VEC + t;
        if (col < cols) partial[static_cast<int64_t>(blockIdx.x) * cols + col] = static_cast<PT>(s);
    }
}

// 16 columns x 16 row slices per block: slice s adds partial rows s, s+16, ... (four loads in flight),
// the slices are combined through LDS in slice order
template <typename OT>
__global__ __launch_bounds__(kThreads)
void colsum_final_kernel(const float *__restrict__ partial, OT *__restrict__ out, int nblk, int cols)
{
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < cols) {
        int b = sl;
        for (; b + 48 < nblk; b += 64) {
            s0 += partial[static_cast<int64_t>(b) * cols + col];
            s1 += partial[static_cast<int64_t>(b + 16) * cols + col];
            s2 += partial[static_cast<int64_t>(b + 32) * cols + col];
            s3 += partial[static_cast<int64_t>(b + 48) * cols + col];
        }
        for (; b < nblk; b += 16) s0 += partial[static_cast<int64_t>(b) * cols + col];
    }
    red[sl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && col < cols) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) s += red[l][c];
        out[col] = static_cast<OT>(s);                      // one rounding of the fp32 sum (a bf16 gradient needs no cast launch)
    }
}

// ---- grouped chunk sums ------------------------------------------------------------------------------------------------------
// Every split weight-gradient kernel (twgrad.hip, conv_wgrad.hip) leaves fp32 partials [chunks][cols] that have to be added in
// a fixed order.  One launch of colsum_partial_kernel per weight gradient was 159 launches of ~6 us in the round-5 step, nearly
// all of it fill and drain.  Here ONE launch adds the partials of up to kChunkJobs gradients: the jobs' tables travel as the
// kernel's argument (copied to LDS once per workgroup), a workgroup owns 1 024 consecutive columns of one job, a thread four
// of them; the chunks are added in order with four loads in flight.  Same sums, same order, same single rounding.
constexpr int kChunkJobs = 48, kChunkCols = 4 * kThreads;
struct ChunkJob {
    const float *part;
    void *out;
    int64_t cols;
    int chunks, out_bf16;
    int block0, pad_;
};
struct ChunkArgs {
    ChunkJob j[kChunkJobs];
    int njobs, pad_;
};

__global__ __launch_bounds__(kThreads)
void chunk_sums_kernel(const ChunkArgs a)
{
    __shared__ __attribute__((aligned(16))) ChunkArgs la;
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(&a);
        unsigned *dst = reinterpret_cast<unsigned *>(&la);
        for (unsigned i = threadIdx.x; i < sizeof(ChunkArgs) / 4; i += kThreads) dst[i] = src[i];
        __syncthreads();
    }
    int ji = 0;
    for (int i = 1; i < la.njobs; ++i)
        if (static_cast<int>(blockIdx.x) >= la.j[i].block0) ji = i;
    const ChunkJob &J = la.j[ji];
    const int64_t cols = J.cols, c = static_cast<int64_t>(static_cast<int>(blockIdx.x) - J.block0) * kChunkCols + 4 * threadIdx.x;
    if (c >= cols) return;
    const float *p = J.part + c;
    const int chunks = J.chunks;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 3 < chunks; k += 4) {
        const float4 v0 = *reinterpret_cast<const float4 *>(p + static_cast<int64_t>(k) * cols);
        const float4 v1 = *reinterpret_cast<const float4 *>(p + static_cast<int64_t>(k + 1) * cols);
        const float4 v2 = *reinterpret_cast<const float4 *>(p + static_cast<int64_t>(k + 2) * cols);
        const float4 v3 = *reinterpret_cast<const float4 *>(p + static_cast<int64_t>(k + 3) * cols);
        s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x; s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
        s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z; s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; k < chunks; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(p + static_cast<int64_t>(k) * cols);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (J.out_bf16) {
        __hip_bfloat16 *o = static_cast<__hip_bfloat16 *>(J.out) + c;
        o[0] = __float2bfloat16(s.x); o[1] = __float2bfloat16(s.y); o[2] = __float2bfloat16(s.z); o[3] = __float2bfloat16(s.w);
    } else {
        *reinterpret_cast<float4 *>(static_cast<float *>(J.out) + c) = s;
    }
}

}  // namespace

const char *chunk_sums_check(const mdetr_chunk_job *jobs, int njobs)
{
    if (!jobs || njobs <= 0) return "no jobs";
    for (int i = 0; i < njobs; ++i) {
        const mdetr_chunk_job &q = jobs[i];
        if (!q.part || !q.out || q.cols <= 0 || q.chunks <= 0) return "null pointer or empty job";
        if (q.cols % 4 != 0 || (reinterpret_cast<uintptr_t>(q.part) & 15) != 0) return "cols must be a multiple of 4 and the partials 16-byte aligned";
        if (q.out_dtype != 0 && q.out_dtype != 2) return "out_dtype must be MDETR_F32 or MDETR_BF16";
        if ((reinterpret_cast<uintptr_t>(q.out) & (q.out_dtype == 2 ? 7 : 15)) != 0) return "result not aligned (16 bytes fp32, 8 bytes bf16)";
        if ((q.cols + kChunkCols - 1) / kChunkCols > (1 << 24)) return "job too wide";
    }
    return nullptr;
}

hipError_t chunk_sums_launch(const mdetr_chunk_job *jobs, int njobs, hipStream_t st)
{
    for (int first = 0; first < njobs; first += kChunkJobs) {
        ChunkArgs a;
        const int n = njobs - first < kChunkJobs ? njobs - first : kChunkJobs;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const mdetr_chunk_job &q = jobs[first + i];
            ChunkJob &J = a.j[i];
            J.part = q.part; J.out = q.out; J.cols = q.cols; J.chunks = q.chunks; J.out_bf16 = q.out_dtype == 2; J.block0 = blocks; J.pad_ = 0;
            blocks += static_cast<int>((q.cols + kChunkCols - 1) / kChunkCols);
        }
        a.njobs = n; a.pad_ = 0;
        hipLaunchKernelGGL(chunk_sums_kernel, dim3(blocks), dim3(kThreads), 0, st, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int64_t colsum_workspace_bytes(int64_t rows, int cols)
{
    const int64_t nblk = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    return nblk * cols * static_cast<int64_t>(sizeof(float));
}

bool colsum_supported(int dtype, int cols, int64_t ld, const void *x)
{
    const int vec = dtype == 2 ? 8 : 4, esz = dtype == 2 ? 2 : 4;
    return (dtype == 0 || dtype == 2) && cols > 0 && cols % vec == 0 && (ld * esz) % 16 == 0 &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

hipError_t colsum_launch(int dtype, const void *x, void *out, void *workspace, int64_t rows, int cols, int64_t ld,
                         hipStream_t st, int out_dtype)
{
    if (cols == 0) return hipSuccess;
    const int64_t nblk = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    // a single row block: its partial row IS the result, in either output type
    float *partial = nblk == 1 ? static_cast<float *>(out) : static_cast<float *>(workspace);
    if (nblk > 0) {
        const int vec = dtype == 2 ? 8 : 4;
        const int cvs = cols / vec;
        int CT = 1;
        while (CT * 2 <= cvs && CT * 2 <= kThreads) CT *= 2;
        // few row blocks and many columns (the chunk sums of a split-K weight gradient: 16 - 69 rows x 65 536 columns, ONE row
        // block): 256 column vectors per workgroup would leave 32 - 65 workgroups on 256 CUs, each lane walking all rows one
        // load after the other.  Trade column vectors for row lanes until ~512 workgroups exist (every row lane keeps a row).
        while (CT > 8 && nblk * ((cvs + CT - 1) / CT) < 512 && kThreads / (CT / 2) <= (rows < kRowsPerBlock ? rows : kRowsPerBlock)) CT /= 2;
        const dim3 grid(static_cast<unsigned>(nblk), static_cast<unsigned>((cvs + CT - 1) / CT));
        if (nblk == 1 && out_dtype == 2) {
            __hip_bfloat16 *o = static_cast<__hip_bfloat16 *>(out);
            if (dtype == 2)
                hipLaunchKernelGGL((colsum_partial_kernel<__hip_bfloat16, __hip_bfloat16>), grid, dim3(kThreads), 0, st,
                                   static_cast<const __hip_bfloat16 *>(x), o, rows, cols, ld, CT);
            else
                hipLaunchKernelGGL((colsum_partial_kernel<float, __hip_bfloat16>), grid, dim3(kThreads), 0, st,
                                   static_cast<const float *>(x), o, rows, cols, ld, CT);
        } else if (dtype == 2) {
            hipLaunchKernelGGL((colsum_partial_kernel<__hip_bfloat16, float>), grid, dim3(kThreads), 0, st,
                               static_cast<const __hip_bfloat16 *>(x), partial, rows, cols, ld, CT);
        } else {
            hipLaunchKernelGGL((colsum_partial_kernel<float, float>), grid, dim3(kThreads), 0, st,
                               static_cast<const float *>(x), partial, rows, cols, ld, CT);
        }
    }
    if (nblk == 1) return hipGetLastError();
    if (out_dtype == 2)
        hipLaunchKernelGGL(colsum_final_kernel<__hip_bfloat16>, dim3((cols + 15) / 16), dim3(kThreads), 0, st,
                           partial, static_cast<__hip_bfloat16 *>(out), static_cast<int>(nblk), cols);
    else
        hipLaunchKernelGGL(colsum_final_kernel<float>, dim3((cols + 15) / 16), dim3(kThreads), 0, st,
                           partial, static_cast<float *>(out), static_cast<int>(nblk), cols);
    return hipGetLastError();
}

}  // namespace mdetr
