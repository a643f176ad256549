// Device-wide index-building primitives used by the voxeliser, the BEV pool and the NMS:
// exclusive scan (int32) and a stable LSD radix sort of (key,value) pairs.
// Built from wave ballots / prefix sums (wave = 64); no library code.
#pragma once
#include "common.h"

namespace heal {

constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// number of int32 scratch words scan_exclusive needs for n elements
inline size_t scan_scratch_words(int64_t n) { return (size_t)ceil_div64(n, SCAN_TILE) + 1; }

// out[i] = sum(in[0..i)); if total != nullptr, *total = sum(in[0..n)).  in may alias out.
int scan_exclusive(const int* in, int* out, int n, int* total, int* scratch, hipStream_t s);

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
constexpr int SORT_RADIX_BITS = 9;
constexpr int SORT_BINS = 1 << SORT_RADIX_BITS;

inline int sort_num_passes(int key_bits) { return ceil_div(key_bits < 1 ? 1 : key_bits, SORT_RADIX_BITS); }
// int32 scratch words for sorting n pairs (histogram + scan scratch)
inline size_t sort_scratch_words(int64_t n) {
    int64_t tiles = ceil_div64(n < 1 ? 1 : n, SORT_TILE);
    return (size_t)(tiles * SORT_BINS) + scan_scratch_words(tiles * SORT_BINS) + 64;
}

// Stable ascending sort on the low `key_bits` bits of the keys.  Buffers [0] hold the input;
// the function ping-pongs between [0] and [1] and returns (through *result_buf) which of the two
// holds the sorted output.
int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int n, int key_bits, int* result_buf,
                     int* scratch, hipStream_t s);

}  // namespace heal
