// tests/native/hipshim/runtime.cpp -- TEST INFRASTRUCTURE: the block executor behind hip/hip_runtime.h (fibers via
// ucontext, see that header for the model) and no-op stand-ins for the few HIP runtime calls capi.hip makes.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipshim {

unsigned long long exchange[1024];
float mfma_a[1024][8], mfma_b[1024][8];
static std::vector<unsigned char> lds_buffer;
unsigned char *dynamic_lds() { return reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(lds_buffer.data()) + 63) & ~uintptr_t(63)); }

namespace {

enum State { READY, WAIT_BLOCK, WAIT_WAVE, DONE };
constexpr size_t kStack = 256 * 1024;

struct Fiber { ucontext_t ctx; State state; };
std::vector<Fiber> fibers;
ucontext_t scheduler;
char *stacks = nullptr;
size_t stacks_for = 0;
int current = -1;
const std::function<void()> *body_fn = nullptr;

void entry()
{
    (*body_fn)();
    fibers[current].state = DONE;
    swapcontext(&fibers[current].ctx, &scheduler);
}

void yield(State s)
{
    const int me = current;
    fibers[me].state = s;
    swapcontext(&fibers[me].ctx, &scheduler);
    threadIdx = dim3(static_cast<unsigned>(me));       // restored by the scheduler too; kept for clarity
}

void run_block(int nthreads)
{
    if (stacks_for < static_cast<size_t>(nthreads)) {
        if (stacks) munmap(stacks, stacks_for * kStack);
        stacks = static_cast<char *>(mmap(nullptr, nthreads * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (stacks == MAP_FAILED) { perror("hipshim: mmap"); abort(); }
        stacks_for = nthreads;
    }
    fibers.assign(nthreads, Fiber());
    for (int t = 0; t < nthreads; ++t) {
        getcontext(&fibers[t].ctx);
        fibers[t].ctx.uc_stack.ss_sp = stacks + t * kStack;
        fibers[t].ctx.uc_stack.ss_size = kStack;
        fibers[t].ctx.uc_link = &scheduler;
        makecontext(&fibers[t].ctx, entry, 0);
        fibers[t].state = READY;
    }
    // Lane order between synchronisation points: ascending by default; HIPSHIM_ORDER=reverse or =shuffle runs the lanes
    // in another order, which exposes a missing barrier (a consumer running before its producer) as a wrong result.
    std::vector<int> order(nthreads);
    for (int t = 0; t < nthreads; ++t) order[t] = t;
    static const char *mode = getenv("HIPSHIM_ORDER");
    static unsigned lcg = 12345u;
    for (;;) {
        if (mode && mode[0] == 'r') for (int t = 0; t < nthreads; ++t) order[t] = nthreads - 1 - t;
        if (mode && mode[0] == 's')
            for (int t = nthreads - 1; t > 0; --t) {
                lcg = lcg * 1664525u + 1013904223u;
                std::swap(order[t], order[(lcg >> 8) % (t + 1)]);
            }
        bool ran = false;
        for (int i = 0; i < nthreads; ++i) {
            const int t = order[i];
            if (fibers[t].state == READY) {
                current = t;
                threadIdx = dim3(static_cast<unsigned>(t));
                swapcontext(&scheduler, &fibers[t].ctx);
                ran = true;
            }
        }
        bool released = false, all_done = true, all_block = true;
        for (int w = 0; w * 64 < nthreads; ++w) {                      // wave-level release
            int live = 0, waiting = 0;
            for (int t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t) {
                live += fibers[t].state != DONE;
                waiting += fibers[t].state == WAIT_WAVE;
            }
            if (live && waiting == live) {
                for (int t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t)
                    if (fibers[t].state == WAIT_WAVE) fibers[t].state = READY;
                released = true;
            }
        }
        for (int t = 0; t < nthreads; ++t) {
            all_done &= fibers[t].state == DONE;
            all_block &= fibers[t].state == DONE || fibers[t].state == WAIT_BLOCK;
        }
        if (all_done) break;
        if (!released && all_block) {
            for (int t = 0; t < nthreads; ++t)
                if (fibers[t].state == WAIT_BLOCK) fibers[t].state = READY;
            released = true;
        }
        if (!released && !ran) {
            fprintf(stderr, "hipshim: deadlock in block (%u,%u,%u): threads wait at different barriers\n", blockIdx.x, blockIdx.y, blockIdx.z);
            abort();
        }
    }
    current = -1;
}

}  // namespace

void sync_block() { yield(WAIT_BLOCK); }
void sync_wave() { yield(WAIT_WAVE); }
int lane_base() { return static_cast<int>(threadIdx.x) / 64 * 64; }
int live_lanes() { return std::min<int>(64, static_cast<int>(blockDim.x) - lane_base()); }

void run(dim3 grid, dim3 block, size_t lds, const std::function<void()> &body)
{
    if (block.y != 1 || block.z != 1) { fprintf(stderr, "hipshim: 1-D blocks only\n"); abort(); }
    if (lds > 160 * 1024) { fprintf(stderr, "hipshim: %zu bytes of dynamic LDS exceed the 160 KiB of a CU\n", lds); abort(); }
    lds_buffer.assign(lds + 64, 0xCD);                     // poisoned: reads of unwritten LDS show up as garbage
    body_fn = &body;
    gridDim = grid;
    blockDim = block;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                blockIdx = dim3(x, y, z);
                run_block(static_cast<int>(block.x));
            }
    body_fn = nullptr;
}

}  // namespace hipshim

hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorNotSupported ? "hipErrorNotSupported" : "hipError"); }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
