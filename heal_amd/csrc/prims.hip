// Device-wide scan and stable radix sort (see prims.h).  gfx950, wave = 64.
#include <stdlib.h>
#include "prims.h"
#include "../../include/heal_amd.h"

namespace heal {

// ------------------------------------------------------------------------------------------------
// error plumbing (one definition for the whole library)
// ------------------------------------------------------------------------------------------------
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_reduce_sum_1024(int v, int* lds /*>=16 ints*/) {
    v = wave_sum_i(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    int t = (threadIdx.x < (blockDim.x >> 6)) ? lds[threadIdx.x] : 0;
    if (threadIdx.x < 64) {
        t = wave_sum_i(t);
        if (threadIdx.x == 0) lds[0] = t;
    }
    __syncthreads();
    int r = lds[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int* __restrict__ in, int n,
                                                                 int* __restrict__ block_sums) {
    __shared__ int lds[32];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int i = base + k;
        if (i < n) s += in[i];
    }
    int tot = block_reduce_sum_1024(s, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const int* in, int* out, int n,
                                                            const int* __restrict__ block_sums,
                                                            int nblocks, int* total) {
    __shared__ int lds[32];
    // offset of this tile = sum of the sums of all earlier tiles
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += block_sums[b];
    const int tile_offset = block_reduce_sum_1024(part, lds);

    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int i = base + k;
        v[k] = (i < n) ? in[i] : 0;
        s += v[k];
    }
    // exclusive scan of the per-thread sums across the block
    int incl = wave_incl_scan(s);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 63) lds[w] = incl;
    __syncthreads();
    if (threadIdx.x < 64) {
        int t = (threadIdx.x < (SCAN_THREADS >> 6)) ? lds[threadIdx.x] : 0;
        int ti = wave_incl_scan(t);
        if (threadIdx.x < (SCAN_THREADS >> 6)) lds[threadIdx.x] = ti - t;  // exclusive wave offsets
        if (threadIdx.x == (SCAN_THREADS >> 6) - 1) lds[16] = ti;          // tile total
    }
    __syncthreads();
    int run = tile_offset + lds[w] + (incl - s);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (total != nullptr && (int)blockIdx.x == nblocks - 1 && threadIdx.x == 0)
        *total = tile_offset + lds[16];
}

// ------------------------------------------------------------------------------------------------
// fill / one-word copy (see common.h: no hipMemsetAsync in this library)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_words(uint32_t* __restrict__ dst, uint32_t v, size_t words) {
    // 16-B stores over the aligned body, words one by one at the ragged ends
    const size_t head = (size_t)((16 - ((uintptr_t)dst & 15)) & 15) / 4;          // words in front of the first 16-B boundary
    const size_t h = head < words ? head : words;
    const size_t body = (words - h) / 4;
    uint4* d4 = reinterpret_cast<uint4*>(dst + h);
    const uint4 v4 = make_uint4(v, v, v, v);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < body; i += stride) d4[i] = v4;
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) dst[threadIdx.x] = v;
        const size_t done = h + body * 4;
        if (threadIdx.x < words - done) dst[done + threadIdx.x] = v;
    }
}

__global__ void k_copy_word(int* __restrict__ dst, const int* __restrict__ src) { *dst = src ? *src : 0; }

int debug_env_once(const char* name) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    if (v != 0)
        fprintf(stderr, "[heal_amd] WARNING: %s=%d is set -- a DEBUG switch that skips parts of the kernel's work; "
                        "outputs of the affected operator are INVALID in this process\n", name, v);
    return v;
}

int fill_bytes(void* dst, int byte, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (dst == nullptr || (((uintptr_t)dst | bytes) & 3) != 0)
        return set_error("fill_bytes: destination / size must be non-null multiples of 4 (%p, %zu)", dst, bytes);
    const uint32_t b = (uint32_t)(byte & 0xFF), v = b | (b << 8) | (b << 16) | (b << 24);
    const size_t words = bytes / 4;
    size_t blocks = (words / 4 + 255) / 256;                       // one 16-B store per thread up to 2048 blocks, then grid-stride
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    k_fill_words<<<(unsigned)blocks, 256, 0, s>>>(reinterpret_cast<uint32_t*>(dst), v, words);
    HEAL_LAUNCH_CHECK();
    return 0;
}

int copy_word(int* dst, const int* src, hipStream_t s) {
    if (dst == nullptr) return set_error("copy_word: null destination");
    k_copy_word<<<1, 1, 0, s>>>(dst, src);
    HEAL_LAUNCH_CHECK();
    return 0;
}

int scan_exclusive(const int* in, int* out, int n, int* total, int* scratch, hipStream_t s) {
    if (n <= 0) {
        if (total) HEAL_FILL(total, 0, sizeof(int), s);
        return 0;
    }
    const int nb = ceil_div(n, SCAN_TILE);
    k_scan_block_sums<<<nb, SCAN_THREADS, 0, s>>>(in, n, scratch);
    k_scan_final<<<nb, SCAN_THREADS, 0, s>>>(in, out, n, scratch, nb, total);
    HEAL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort, 9-bit digits
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint32_t* __restrict__ keys, int n,
                                                           int shift, int ntiles,
                                                           int* __restrict__ hist /*[BINS][ntiles]*/) {
    __shared__ int h[SORT_BINS];
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        int e = base + it * SORT_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & (SORT_BINS - 1)], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS) hist[d * ntiles + blockIdx.x] = h[d];
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n, int shift, int ntiles,
    const int* __restrict__ gbase /*[BINS][ntiles], exclusive-scanned*/) {
    constexpr int WAVES = SORT_THREADS / 64;
    // cnt[it][wave][digit]: elements with that digit held by (iteration, wave); turned into the
    // exclusive prefix over (it, wave) order, which is the stable order inside the tile.
    __shared__ int cnt[SORT_ITEMS * WAVES * SORT_BINS];
    for (int i = threadIdx.x; i < SORT_ITEMS * WAVES * SORT_BINS; i += SORT_THREADS) cnt[i] = 0;
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const unsigned long long lt = lanemask_lt();
    const int base = blockIdx.x * SORT_TILE;
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS];
    int digit[SORT_ITEMS], rank[SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int e = base + it * SORT_THREADS + threadIdx.x;
        const bool valid = e < n;
        key[it] = valid ? keys_in[e] : 0u;
        val[it] = valid ? vals_in[e] : 0u;
        const int d = (key[it] >> shift) & (SORT_BINS - 1);
        digit[it] = d;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < SORT_RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        rank[it] = __popcll(peers & lt);
        if (valid && rank[it] == 0) cnt[(it * WAVES + wave) * SORT_BINS + d] = __popcll(peers);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS) {
        int run = gbase[d * ntiles + blockIdx.x];
#pragma unroll 4
        for (int j = 0; j < SORT_ITEMS * WAVES; ++j) {
            int c = cnt[j * SORT_BINS + d];
            cnt[j * SORT_BINS + d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int e = base + it * SORT_THREADS + threadIdx.x;
        if (e < n) {
            const int pos = cnt[(it * WAVES + wave) * SORT_BINS + digit[it]] + rank[it];
            keys_out[pos] = key[it];
            vals_out[pos] = val[it];
        }
    }
}

int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int n, int key_bits, int* result_buf,
                     int* scratch, hipStream_t s) {
    *result_buf = 0;
    if (n <= 1) return 0;
    const int ntiles = ceil_div(n, SORT_TILE);
    const int npass = sort_num_passes(key_bits);
    int* hist = scratch;
    int* scan_scratch = scratch + (size_t)ntiles * SORT_BINS;
    int cur = 0;
    for (int p = 0; p < npass; ++p) {
        const int shift = p * SORT_RADIX_BITS;
        k_sort_hist<<<ntiles, SORT_THREADS, 0, s>>>(keys[cur], n, shift, ntiles, hist);
        if (scan_exclusive(hist, hist, ntiles * SORT_BINS, nullptr, scan_scratch, s)) return 1;
        k_sort_scatter<<<ntiles, SORT_THREADS, 0, s>>>(keys[cur], vals[cur], keys[cur ^ 1],
                                                       vals[cur ^ 1], n, shift, ntiles, hist);
        cur ^= 1;
    }
    HEAL_LAUNCH_CHECK();
    *result_buf = cur;
    return 0;
}

static thread_local LaunchEvents g_launch_events = {nullptr, nullptr};
LaunchEvents take_launch_events() {
    const LaunchEvents e = g_launch_events;
    g_launch_events = {nullptr, nullptr};
    return e;
}

}  // namespace heal

extern "C" {
int heal_next_launch_events(void* start_event, void* stop_event) {
    heal::g_launch_events = {(hipEvent_t)start_event, (hipEvent_t)stop_event};
    return 0;
}
int heal_abi_version(void) { return HEAL_AMD_ABI_VERSION; }
const char* heal_last_error(void) { return heal::err_buf(); }
int heal_fill_bytes(void* dst, int byte, size_t bytes, void* stream) { return heal::fill_bytes(dst, byte, bytes, (hipStream_t)stream); }
}
