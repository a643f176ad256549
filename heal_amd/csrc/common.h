// Shared host/device helpers for libheal_amd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>

#define HEAL_WAVE 64

namespace heal {

// thread-local last error text, exposed through heal_last_error()
char* err_buf();
int set_error(const char* fmt, ...);

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over the caller-provided workspace.
struct Arena {
    char* base; size_t cap; size_t off;
    Arena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
    template <typename T> T* take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= cap && (base != nullptr || off == 0); }
};

#define HEAL_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return heal::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                   __FILE__, __LINE__);                                     \
    } while (0)

#define HEAL_REQUIRE(cond, ...)                       \
    do {                                              \
        if (!(cond)) return heal::set_error(__VA_ARGS__); \
    } while (0)

#define HEAL_LAUNCH_CHECK() HEAL_HIP(hipGetLastError())

// Debug switches that SKIP WORK (HEAL_K4_DBG, HEAL_K5_DBG, HEAL_SP_DBG: timing anatomy only, outputs invalid) are read ONCE per process and
// announced on stderr when set, so that a stray variable cannot corrupt results silently and costs nothing per launch.
int debug_env_once(const char* name);
#define HEAL_DEBUG_ENV(name) ([]() -> int { static const int v__ = heal::debug_env_once(name); return v__; }())

// fill_bytes: the library's ONLY way of initialising device memory (a kernel, prims.hip).  hipMemsetAsync is not used anywhere: captured
// into a HIP graph it becomes a memset NODE, and in round 4 K1's 0xFF table fill came back from such a node with byte 0 of every
// 16 B cleared (workspace dump in profiles/r05_k1_memset_node_dump.txt; DESIGN 5 "memory fault") -- the runtime's fill goes through
// a pattern buffer that is not private to the node.  A kernel node carries its pattern in its own arguments.
// dst and bytes must be multiples of 4; `byte` is replicated.  Returns 0 / sets the error text.
int fill_bytes(void* dst, int byte, size_t bytes, hipStream_t s);
// 4-byte device-to-device copy as a kernel (dst <- *src, or 0 when src is null); same reason
int copy_word(int* dst, const int* src, hipStream_t s);
#define HEAL_FILL(dst, byte, bytes, s)                      \
    do {                                                    \
        if (heal::fill_bytes((dst), (byte), (bytes), (s))) return 1; \
    } while (0)

// Measurement hook (heal_next_launch_events): a thread-local pair of events armed by the caller and consumed by the next launch
// that supports it (HEAL_LAUNCH_EV).  hipExtLaunchKernelGGL stamps the events with the kernel's OWN begin / end, the interval a
// rocprofv3 kernel trace reports; an event pair recorded around a launch also holds the dispatch and marker latencies (3-5 us,
// a quarter of a 13-us kernel).
struct LaunchEvents { hipEvent_t start, stop; };
LaunchEvents take_launch_events();
#define HEAL_LAUNCH_EV(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                             \
        const heal::LaunchEvents ev__ = heal::take_launch_events();                                                  \
        if (ev__.start && ev__.stop)                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, ev__.start, ev__.stop, 0, __VA_ARGS__); \
        else                                                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, __VA_ARGS__);                      \
    } while (0)

// the same with explicit events (either may be null): the first kernel of a two-kernel operator takes `start`, the last `stop`
#define HEAL_LAUNCH_EV2(kernel, grid, block, lds, stream, ev_start, ev_stop, ...)                                    \
    do {                                                                                                             \
        if ((ev_start) || (ev_stop))                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, ev_start, ev_stop, 0, __VA_ARGS__); \
        else                                                                                                         \
            hipLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, __VA_ARGS__);                      \
    } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}


// Workgroup barrier for data exchanged through LDS only.  `__syncthreads()` compiles to `s_waitcnt vmcnt(0) lgkmcnt(0);
// s_barrier` (its workgroup-scope fence covers global memory too), which drains every global load in flight and so exposes
// one L2/HBM round trip per barrier in a software-pipelined loop.  This form waits for the wave's LDS operations only; global
// loads issued before it stay in flight across the barrier (the compiler still waits for them before their first use).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- XCD-aware block index -----------------------------------------------------------------------------
// The dispatcher is observed to place block b (x fastest) on XCD b % 8, each XCD with a private 4 MiB L2.
// Tiled kernels whose neighbouring tiles share input (conv halos, rotated bilinear footprints) remap the
// dispatch id so that every XCD works on one CONTIGUOUS range of logical tiles; the shared lines are then
// fetched into one L2 instead of up to eight.  Bijective for any grid size.  A speed choice only.
struct Block3 { unsigned x, y, z; };
__device__ inline Block3 xcd_block() {
    const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
    unsigned b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    Block3 o;
    o.x = b % gx;
    b /= gx;
    o.y = b % gy;
    o.z = b / gy;
    return o;
}

}  // namespace heal
