// K1 -- deterministic hard voxelisation on gfx950.
//
// Semantics restated from spconv's CPU point->voxel generator as called at
// opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:46-68:
// scan points in input order; cell = floor((p - range_min) / voxel_size) per axis in fp32, drop if
// outside [0,grid); a new voxel id is handed out on the first touch of a cell (ids in order of
// first appearance, none beyond max_voxels); a point is appended to its voxel while the voxel holds
// fewer than max_points.
//
// Round 6: THREE kernels and no fill -- the per-cell table is one 16-B record {key, min point index, counter, segment}, SELF-CLEANING (the
// last kernel resets every record it used, so a workspace that entered clean leaves clean: `tables_clean`), and the index-list fill is
// folded into k_vox_assign (a point waits for its cell's segment start, which a tile at or below its own publishes).  The description
// below is the round-4 chain these two changes start from.
// GPU formulation (order-independent, so bit-identical to the sequential scan), ONE fill + FOUR kernels for any number of
// agents (round 4; round 3: five; rounds 1-2 sorted (voxel id, point index) pairs with a 3-pass radix sort: 24 launches):
//   0. one fill(0xFF) (a KERNEL, heal::fill_bytes -- not hipMemsetAsync, see common.h): hash keys | per-cell minimum point index | per-cell point counter | the tiles' publication words
//   1. k_voxb_insert: hash-grid insert, cell -> min point index, ticket   (atomicCAS claim + atomicMin + atomicAdd, each behind a
//      plain read that rules most of the first two out)
//   2 + 3. k_vox_assign: per 1024-point tile the "i is the first point of its cell" flags are counted and PUBLISHED (one 64-bit word
//      per tile), the tile looks back over the published words of the tiles below it (chained scan: tiles publish before they wait
//      and are dispatched in order), ranks its first points -> voxel id in first-appearance order per agent, row in the collated
//      output, coordinates; both caps applied;
//      and a segment [seg, seg + points of the cell) of one index list (prefix sums of the cells' point counts: deterministic)
//   4. k_vox_fill: every point drops its index at seg + ticket (the ticket it drew from its cell's counter in step 1: an
//      arbitrary but collision-free position)
//   5. k_vox_select_write: a group of G >= P lanes per voxel sorts the segment with a bitonic network on lane shuffles (longer
//      segments are merged in G-element chunks), keeps the P smallest point indices = the first P points in input order,
//      whatever the execution order, gathers them and writes the row (zeros beyond): coalesced 16-B stores, no memset.
//   (max_points > 64: steps 4-5 are an atomicMin insertion cascade into P candidate slots per voxel + a gather.  An earlier
//   version of this round used the cascade for every P: 339 us on three 64-line sweeps -- the 32 slots of a dense pillar are one
//   cache line and a few hundred points serialise on it -- against 12 us for fill + select.)
#include "prims.h"
#include "../../include/heal_amd.h"

namespace heal {

constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct VoxGrid {
    float rmin[3];
    float vsize[3];
    int grid[3];  // x,y,z
};

__device__ __forceinline__ bool point_cell(const float4 p, const VoxGrid& g, int& cx, int& cy, int& cz) {
    // fp32, one subtraction and one IEEE division per axis, then floor (no contraction: the
    // library is built with -ffp-contract=off)
    const float fx = floorf((p.x - g.rmin[0]) / g.vsize[0]);
    const float fy = floorf((p.y - g.rmin[1]) / g.vsize[1]);
    const float fz = floorf((p.z - g.rmin[2]) / g.vsize[2]);
    // NaN coordinates fail every comparison below and are dropped
    if (!(fx >= 0.f && fx < (float)g.grid[0])) return false;
    if (!(fy >= 0.f && fy < (float)g.grid[1])) return false;
    if (!(fz >= 0.f && fz < (float)g.grid[2])) return false;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    return true;
}

constexpr int VOX_MAX_BATCH = 16;
constexpr int VOX_TILE = 1024;          // points per scan tile (4 per thread)
constexpr int VOX_MAX_TILES = 4096;     // block-local scan of the tile counts: clouds up to 4 M points per call
constexpr uint32_t VOX_DROPPED = 0xFFFFFFFFu;
struct VoxBatch {
    int B;
    int pt_off[VOX_MAX_BATCH + 1];  // points of agent b: [pt_off[b], pt_off[b+1])
    int label0;                     // coords[:, 0] of agent b = label0 + b (single-cloud form: its batch_idx)
};

__device__ __forceinline__ int vox_agent(const VoxBatch& vb, int i) {
    int a = 0;
#pragma unroll 1
    while (a + 1 < vb.B && i >= vb.pt_off[a + 1]) ++a;
    return a;
}

// 1. insert: table_key[slot] = (agent, cell), table_min[slot] = min point index, ticket; slot_of[i] = slot or -1.
// (Grouping the lanes of a wave by cell first -- ballot matching, one set of atomics per group leader -- was measured on three
// 64-line sweeps: 35 us against 26 us for this form; the matching loop costs more than the same-address atomics it saves.)
// DENSE (round 4, pillar grids): when agents x cells fits the table the cell IS the slot -- no key array, no probe, no CAS claim
// (131 072 cells x 3 agents against 524 288 slots in scene 5); what follows only uses slots as ids, so the output is unchanged.
// per-cell words, one ARRAY each: tkey = hash key (the cell; no array for the dense map), tmin = minimum point index, tcnt = point counter
// (starts at 0xFFFFFFFF: the ticket of the first arrival is 0), tseg = start of the cell's segment in the index list (bit 31: the cell's voxel
// was dropped by max_voxels); all-ones = free.  (Measured and dropped in round 6: ONE 16-B record per cell -- a point's atomicMin and atomicAdd
// then hit the same line and SERIALISE at the memory side: k_voxb_insert 21.6 -> 32.7 us for the dense map, 27.3 for the hash grid.)
constexpr uint32_t VOX_DROP_BIT = 0x80000000u;
// meta words (ints): 0 rows written, 1 base row, 2 generation of the last finished chain, 3 generation of the chain in flight,
// 4 generation of the last chain that dropped a voxel
constexpr int VM_ROWS = 0, VM_BASE = 1, VM_GEN = 2, VM_GEN_NEXT = 3, VM_DROPGEN = 4;

template <bool DENSE>
__global__ __launch_bounds__(256) void k_voxb_insert(const float4* __restrict__ pts, VoxBatch vb, VoxGrid g,
                                                    uint32_t cells, uint32_t* __restrict__ tkey, uint32_t* __restrict__ tmin,
                                                    uint32_t* __restrict__ tcnt, uint32_t mask, int2* __restrict__ st,
                                                    int* __restrict__ meta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) meta[VM_GEN_NEXT] = meta[VM_GEN] + 1;      // (nobody reads VM_GEN_NEXT in this kernel, nobody writes VM_GEN)
    if (i >= vb.pt_off[vb.B]) return;
    int cx, cy, cz;
    if (!point_cell(pts[i], g, cx, cy, cz)) { st[i] = make_int2(-1, 0); return; }
    const uint32_t cell = (uint32_t)vox_agent(vb, i) * cells +
                          (((uint32_t)cz * (uint32_t)g.grid[1] + (uint32_t)cy) * (uint32_t)g.grid[0] + (uint32_t)cx);
    // Scattered device-scope atomics run at a fixed rate (~27 G operations/s chip-wide): every one that a plain read can rule out is
    // time saved.  A plain read may be stale, but only in the harmless direction: a key seen as the cell's own is final (keys are
    // written once), a key seen as EMPTY / a minimum seen too large just falls through to the atomic.
    uint32_t slot = cell;
    if constexpr (!DENSE) {
        slot = hash_u32(cell) & mask;
        for (;;) {
            const uint32_t seen = tkey[slot];
            if (seen == cell) break;                       // claimed earlier by another point of this cell: no CAS
            if (seen == HASH_EMPTY) {
                const uint32_t prev = atomicCAS(&tkey[slot], HASH_EMPTY, cell);
                if (prev == HASH_EMPTY || prev == cell) break;
            }
            slot = (slot + 1) & mask;
        }
    }
    if (tmin[slot] > (uint32_t)i) atomicMin(&tmin[slot], (uint32_t)i);   // (a stale value is >= the true one: skipping is safe)
    const uint32_t tick = atomicAdd(&tcnt[slot], 1u) + 1u;     // the counter starts at 0xFFFFFFFF: tickets 0, 1, ...; final value = points - 1
    st[i] = make_int2((int)slot, (int)tick);
}

// first-point flags of a tile: point j * 256 + t of tile `tile` (j = 0..3); bit j of the result
// (also returns the points' slots / tickets and, for first points, their cells' point counts)
__device__ __forceinline__ unsigned tile_flags(const int2* __restrict__ st, const uint32_t* __restrict__ tmin,
                                               const uint32_t* __restrict__ tcnt, int tile, int n, int2 (&my)[4], int (&cnt)[4]) {
    unsigned f = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tile * VOX_TILE + j * 256 + (int)threadIdx.x;
        my[j] = make_int2(-1, 0);
        cnt[j] = 0;
        if (i < n) {
            my[j] = st[i];
            if (my[j].x >= 0) {
                if (tmin[my[j].x] == (uint32_t)i) { f |= 1u << j; cnt[j] = (int)(tcnt[my[j].x] + 1u); }
            }
        }
    }
    return f;
}

__device__ __forceinline__ int block_sum_256(int v, int* s_red) {   // all threads get the sum; s_red: 4 ints, reused
    v = wave_sum_i(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// 2 + 3. ranks -> rows and segments, ONE launch (round 4; rounds 1-3: a tile-sums launch, then every block re-scanned all tile sums).
// Every 1024-point tile counts its first points and the points of their cells, PUBLISHES the pair as one 64-bit word
// (tile_pub[tile] = first points << 32 | cell points; the array is part of the 0xFF fill, bit 63 set = not yet published) and then
// looks back: thread t waits for tiles t, t + 256, ... below its own and adds them up -- tiles are dispatched in index order and
// publish before they wait, so the chain always makes progress.  A tile that holds the first point of agent b also publishes the
// first points in front of that boundary (part_pub[b]).  meta[0] = rows written (M), meta[1] = base row (0 without row_offset):
// written by the LAST tile, which has seen every other.
__device__ __forceinline__ unsigned long long vox_wait_pub(const unsigned long long* p) {
    unsigned long long v;
    do { v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (v >> 63);
    return v;
}

__global__ __launch_bounds__(256) void k_vox_assign(const float4* __restrict__ pts, VoxBatch vb, VoxGrid g,
                                                   const int2* __restrict__ st, const uint32_t* __restrict__ tmin,
                                                   const uint32_t* __restrict__ tcnt, uint32_t* __restrict__ tseg,
                                                   unsigned long long* __restrict__ tile_pub,
                                                   unsigned long long* __restrict__ part_pub,
                                                   int n_tiles, int max_voxels, const int* __restrict__ row_offset,
                                                   uint32_t* __restrict__ tvid /* P > 64 only, else null */, int* __restrict__ row_seg,
                                                   int* __restrict__ row_cnt, int* __restrict__ row_slot, int* __restrict__ coords,
                                                   int* __restrict__ offsets_out, int* __restrict__ n_voxels_out,
                                                   int* __restrict__ row_offset_next, int* __restrict__ meta,
                                                   uint32_t* __restrict__ seg /* null: no index list (P > 64) */) {
    __shared__ int s_red[4], s_w[4][4], s_wc[4][4], s_vbase[VOX_MAX_BATCH + 1], s_obase[VOX_MAX_BATCH + 1], s_run[2];
    __shared__ int s_part[VOX_MAX_BATCH + 1], s_vsum[VOX_MAX_BATCH + 1];
    const int n = vb.pt_off[vb.B], tile = blockIdx.x, t = threadIdx.x;
    int2 my[4];
    int cnt[4];
    const unsigned f = tile_flags(st, tmin, tcnt, tile, n, my, cnt);
    const int wave = t >> 6;
    const unsigned long long lt = lanemask_lt();
    int before[4], cbefore[4];     // first points (and their cells' points) of this wave that precede point (j, t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool first = (f >> j) & 1u;
        const unsigned long long bal = __ballot(first);
        before[j] = __popcll(bal & lt);
        const int incl = wave_incl_scan(cnt[j]);
        cbefore[j] = incl - cnt[j];
        if ((t & 63) == 0) s_w[j][wave] = __popcll(bal);
        if ((t & 63) == 63) s_wc[j][wave] = incl;
    }
    __syncthreads();
    // ---- publish this tile's sums (before any waiting) ---------------------------------------------------------------------------------
    if (t == 0) {
        int tot = 0, totc = 0;
        for (int j = 0; j < 4; ++j)
            for (int w = 0; w < 4; ++w) { tot += s_w[j][w]; totc += s_wc[j][w]; }
        s_run[0] = tot; s_run[1] = totc;
        __hip_atomic_store(&tile_pub[tile], ((unsigned long long)(unsigned)tot << 32) | (unsigned)totc, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    const int last_i = min((tile + 1) * VOX_TILE, n) - 1;
    const bool last_tile = tile == n_tiles - 1;
    const int a_max = last_tile ? vb.B : vox_agent(vb, last_i);      // agents whose bases this tile needs (the last tile: all, for the totals)
    for (int b = 1; b <= min(a_max, vb.B - 1); ++b) {                 // block-uniform: a boundary inside this tile is rare
        const int bnd = vb.pt_off[b];
        if (bnd < tile * VOX_TILE || bnd >= (tile + 1) * VOX_TILE || bnd >= n) continue;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((f >> j) & 1u) && tile * VOX_TILE + j * 256 + t < bnd) ++c;
        c = block_sum_256(c, s_red);
        if (t == 0) {
            s_part[b] = c;
            __hip_atomic_store(&part_pub[b], (unsigned long long)(unsigned)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- look back: sums of the tiles below this one; per agent boundary the sums of the tiles below the boundary's tile ----------------
    int P = 0, C = 0;
    int vs[VOX_MAX_BATCH];
#pragma unroll
    for (int b = 0; b < VOX_MAX_BATCH; ++b) vs[b] = 0;
    for (int q = t; q < tile; q += 256) {
        const unsigned long long v = vox_wait_pub(&tile_pub[q]);
        const int fp = (int)(v >> 32), cp = (int)(unsigned)v;
        P += fp; C += cp;
#pragma unroll
        for (int b = 1; b < VOX_MAX_BATCH; ++b)
            if (b <= a_max && b < vb.B && q < vb.pt_off[b] / VOX_TILE) vs[b] += fp;
    }
    P = block_sum_256(P, s_red);
    C = block_sum_256(C, s_red);
#pragma unroll
    for (int b = 1; b < VOX_MAX_BATCH; ++b) {
        if (b > a_max || b >= vb.B) break;                            // block-uniform
        const int v = block_sum_256(vs[b], s_red);
        if (t == 0) s_vsum[b] = v;
    }
    if (t == 0) {
        const int total = P + s_run[0];                               // meaningful in the last tile only
        int acc = 0;
        for (int b = 0; b <= a_max; ++b) {
            const int bnd = vb.pt_off[b];
            int v;
            if (b == 0) v = 0;
            else if (b == vb.B || bnd >= n) v = total;                // (reached by the last tile only)
            else {
                const int bt = bnd / VOX_TILE;
                const int part = bt == tile ? s_part[b] : (int)(unsigned)vox_wait_pub(&part_pub[b]);
                v = s_vsum[b] + part;
            }
            s_vbase[b] = v;
            if (b > 0) acc += min(v - s_vbase[b - 1], min(max_voxels, vb.pt_off[b] - vb.pt_off[b - 1]));
            s_obase[b] = acc;
        }
        if (last_tile) {
            const int base = row_offset ? *row_offset : 0;
            meta[VM_ROWS] = acc; meta[VM_BASE] = base;
            if (offsets_out) for (int b = 0; b <= vb.B; ++b) offsets_out[b] = s_obase[b];
            if (n_voxels_out) *n_voxels_out = acc;
            if (row_offset_next) *row_offset_next = base + acc;
        }
    }
    __syncthreads();
    const int base = row_offset ? *row_offset : 0;
    int run = P, runc = C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = run + before[j], sg = runc + cbefore[j];
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { r += s_w[j][w]; sg += s_wc[j][w]; }
            run += s_w[j][w];
            runc += s_wc[j][w];
        }
        if (!((f >> j) & 1u)) continue;
        const int i = tile * VOX_TILE + j * 256 + t;
        const int a = vox_agent(vb, i);
        const int vl = r - s_vbase[a];                 // first-appearance rank inside the agent
        const int s = my[j].x;
        uint32_t segw = (uint32_t)sg;
        if (vl < max_voxels) {
            const int row = s_obase[a] + vl;           // row in this call's outputs (the caller's base row is added on write)
            if (tvid) tvid[s] = (uint32_t)row;
            row_seg[row] = sg;
            row_cnt[row] = cnt[j];
            row_slot[row] = s;
            int cx, cy, cz;
            point_cell(pts[i], g, cx, cy, cz);
            reinterpret_cast<int4*>(coords)[base + row] = make_int4(vb.label0 + a, cz, cy, cx);
        } else {
            if (tvid) tvid[s] = VOX_DROPPED;           // voxels past max_voxels are dropped
            segw |= VOX_DROP_BIT;
            meta[VM_DROPGEN] = meta[VM_GEN_NEXT];      // (every writer stores the same value; read by the last kernel of the chain)
        }
        // publish the segment start: every point of the cell is waiting for it (below, in this or a later tile)
        __hip_atomic_store(&tseg[s], segw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!seg) return;
    // ---- index list (round 6: was its own launch): the points of a cell occupy [segment, segment + count) in ticket order.  A point's
    // cell is ranked by the cell's FIRST point, which lies in this tile or in a lower one -- tiles are dispatched in index order and a
    // tile publishes its segments without waiting for anybody above it, so the wait always ends.
    __syncthreads();                                   // the segments this tile itself published: visible before its own points look
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (my[j].x < 0) continue;
        uint32_t w;
        do { w = __hip_atomic_load(&tseg[my[j].x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (w == 0xFFFFFFFFu);
        if (!(w & VOX_DROP_BIT)) seg[w + (uint32_t)my[j].y] = (uint32_t)(tile * VOX_TILE + j * 256 + t);
    }
}

// Table hygiene, by the LAST kernel of a chain: every record the call used goes back to all-ones (rows: through row_slot; dropped cells,
// which have no row: their first point finds them, only on calls that dropped one), the tiles' publication words too.  All threads of
// the grid call it with their global index.
struct VoxTab { uint32_t *tkey /* null: dense map */, *tmin, *tcnt, *tseg; };

__device__ __forceinline__ void vox_free_slot(const VoxTab& tb, int s) {
    if (tb.tkey) tb.tkey[s] = 0xFFFFFFFFu;
    tb.tmin[s] = 0xFFFFFFFFu; tb.tcnt[s] = 0xFFFFFFFFu; tb.tseg[s] = 0xFFFFFFFFu;
}

__device__ __forceinline__ void vox_clean(long long gid, long long gsize, VoxTab tb, const int2* __restrict__ st,
                                          const int* __restrict__ row_slot, unsigned long long* __restrict__ tile_pub,
                                          unsigned long long* __restrict__ part_pub, int n_tiles, int n, int* __restrict__ meta) {
    const int M = meta[VM_ROWS], gen = meta[VM_GEN_NEXT];
    for (long long r = gid; r < M; r += gsize) vox_free_slot(tb, row_slot[r]);
    for (long long q = gid; q < n_tiles + 1 + VOX_MAX_BATCH + 1; q += gsize) {
        if (q <= n_tiles) tile_pub[q] = ~0ull;
        else part_pub[q - n_tiles - 1] = ~0ull;
    }
    if (meta[VM_DROPGEN] == gen) {
        for (long long i = gid; i < n; i += gsize) {
            const int s = st[i].x;
            if (s < 0) continue;
            const uint32_t sg = tb.tseg[s];
            if (tb.tmin[s] == (uint32_t)i && (sg & VOX_DROP_BIT) && sg != 0xFFFFFFFFu) vox_free_slot(tb, s);
        }
    }
    if (gid == 0) meta[VM_GEN] = gen;                  // (read by the NEXT chain's first kernel only)
}

__global__ __launch_bounds__(256) void k_vox_clean(VoxTab tb, const int2* __restrict__ st, const int* __restrict__ row_slot,
                                                  unsigned long long* __restrict__ tile_pub, unsigned long long* __restrict__ part_pub,
                                                  int n_tiles, int n, int* __restrict__ meta) {
    vox_clean((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, tb, st, row_slot, tile_pub, part_pub, n_tiles, n, meta);
}

// bitonic network over the G lanes of a group (G a power of two <= 64, groups aligned in the wave): ascending
template <int G>
__device__ __forceinline__ uint32_t group_sort(uint32_t v, int l) {
#pragma unroll
    for (int k = 2; k <= G; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t o = __shfl_xor(v, j, 64);
            const bool up = (l & k) == 0 || k == G, lower = (l & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_merge(uint32_t v, int l) {   // v bitonic -> ascending
#pragma unroll
    for (int j = G >> 1; j > 0; j >>= 1) {
        const uint32_t o = __shfl_xor(v, j, 64);
        v = ((l & j) == 0) ? min(v, o) : max(v, o);
    }
    return v;
}

// 5. G lanes per voxel: the P smallest point indices of its segment, ascending -> gather -> row
template <int G>
__global__ __launch_bounds__(256) void k_vox_select_write(const float4* __restrict__ pts, const uint32_t* __restrict__ seg,
                                                         const int* __restrict__ row_seg, const int* __restrict__ row_cnt,
                                                         int* __restrict__ meta, int P, float4* __restrict__ voxels,
                                                         int* __restrict__ num_points, int n_pts, VoxTab tb,
                                                         const int2* __restrict__ slot_tick, const int* __restrict__ row_slot,
                                                         unsigned long long* __restrict__ tile_pub,
                                                         unsigned long long* __restrict__ part_pub, int n_tiles) {
    // table hygiene first (nothing below reads the records): this is the chain's last kernel
    vox_clean((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, tb, slot_tick, row_slot, tile_pub, part_pub, n_tiles, n_pts, meta);
    const int row = blockIdx.x * (256 / G) + (int)threadIdx.x / G, l = (int)threadIdx.x & (G - 1);
    const int M = meta[VM_ROWS];
    if ((int)blockIdx.x * (256 / G) >= M) return;     // block-uniform: the grid is sized by capacity
    // (a group past the last row idles through the shuffles with an empty segment: whole waves take the same path)
    const bool live = row < M;
    const int cnt = live ? row_cnt[row] : 0, st = live ? row_seg[row] : 0;
    // the longest segment of the wave decides the trip count (wave-uniform loop: the shuffles need every lane)
    int longest = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o, 64));
    uint32_t best = group_sort<G>(l < cnt ? seg[st + l] : HASH_EMPTY, l);
    for (int pos = G; pos < longest; pos += G) {
        const uint32_t c = group_sort<G>(pos + l < cnt ? seg[st + pos + l] : HASH_EMPTY, l);
        const uint32_t r = __shfl(c, ((int)threadIdx.x & 63 & ~(G - 1)) | (G - 1 - l), 64);   // the chunk, descending
        best = group_merge<G>(min(best, r), l);                                              // the G smallest of both
    }
    if (!live || l >= P) return;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (best != HASH_EMPTY) {
#ifdef HEAL_VOX_RECORDER
        // DEBUG BUILDS ONLY (-DHEAL_VOX_RECORDER; scripts/ring_dbg.py zeroes and prints meta[8..14]): an index that is no point of this call can
        // only come from a corrupted table (round 4: the runtime's memset node, see common.h fill_bytes); it is recorded -- meta[8] = hits,
        // meta[9..14] = index, row, count, segment, M, lane of the first -- instead of being dereferenced.  The production kernel has NO such
        // guard (round 6): it would turn a memory fault into silently wrong voxels with no host-visible signal; the cause it was
        // written for is removed by construction (no runtime memset / copy node anywhere in the library), and a corrupted table must fault loudly.
        if (best >= (uint32_t)n_pts) {
            if (atomicAdd(&meta[8], 1) == 0) { meta[9] = (int)best; meta[10] = row; meta[11] = cnt; meta[12] = st; meta[13] = M; meta[14] = l; }
        } else
#endif
        val = pts[best];
    }
    const int base = meta[VM_BASE];
    voxels[(size_t)(base + row) * P + l] = val;
    if (l == 0) num_points[base + row] = min(cnt, P);
}

// 4'. (max_points > 64) candidate slots: cand[row][0..P) ascending, 0xFFFFFFFF = empty
__global__ __launch_bounds__(256) void k_vox_candidates(const int2* __restrict__ st, const uint32_t* __restrict__ tvid,
                                                       int n, int P, uint32_t* __restrict__ cand) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = st[i].x;
    if (s < 0) return;
    const uint32_t row = tvid[s];
    if (row == VOX_DROPPED) return;
    uint32_t* c = cand + (size_t)row * P;
    // P smaller indices already sit in the slots (values only decrease; the occupant of the last slot came past P - 1 smaller
    // ones): this point is not among the first P of its cell
    if (__builtin_nontemporal_load(c + (P - 1)) < (uint32_t)i) return;
    uint32_t v = (uint32_t)i;
    for (int k = 0; k < P; ++k) {
        if (__builtin_nontemporal_load(c + k) < v) continue;      // a (possibly stale, hence larger) smaller value: not our slot
        const uint32_t old = atomicMin(c + k, v);
        if (old == HASH_EMPTY) return;                            // took a free slot
        if (old > v) v = old;                                     // displaced a larger index: it moves on
    }
}

// 5'. (max_points > 64) thread (row, slot): the point or zeros; the number of points = position of the first empty slot
__global__ __launch_bounds__(256) void k_vox_write(const float4* __restrict__ pts, const uint32_t* __restrict__ cand,
                                                  const int* __restrict__ meta, int P, float4* __restrict__ voxels,
                                                  int* __restrict__ num_points) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int row = (int)(e / P), p = (int)(e - (long long)row * P);
    if (row >= meta[VM_ROWS]) return;
    const int base = meta[VM_BASE];
    const uint32_t idx = cand[e];
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx != HASH_EMPTY) val = pts[idx];
    voxels[(size_t)(base + row) * P + p] = val;
    if (idx == HASH_EMPTY) {
        if (p == 0 || cand[e - 1] != HASH_EMPTY) num_points[base + row] = p;
    } else if (p == P - 1) {
        num_points[base + row] = P;
    }
}

struct VoxWs {
    uint32_t *tkey, *tmin, *tcnt, *tseg;         // [tcap] per-cell words (tkey: hash grid only)
    uint32_t *cand, *tvid, *seg;
    int2* st;                                    // [n] (slot, ticket) of every point
    int *row_seg, *row_cnt, *row_slot, *meta;
    unsigned long long *tile_pub, *part_pub;     // published tile sums / agent-boundary partial counts (all-ones: unpublished)
    uint32_t tcap;
    size_t clean_bytes;     // tmin | tcnt | tseg | tkey | tile_pub | part_pub: all-ones between calls (self-cleaning); meta follows (zero-initialised)
};

static uint32_t table_cap(int n) {
    uint32_t c = 1024;
    while (c < 2u * (uint32_t)(n < 1 ? 1 : n)) c <<= 1;
    return c;
}

// Table slots: pillar grids whose agents x cells stay below 2^21 use the CELL as the slot (no key, no probe, no CAS): with the self-cleaning
// records a large sparse table costs nothing per call (nothing is filled, only used records are reset), so the table is simply made as
// large as the grid -- 786 432 cells for three agents at 512 x 512, 12.6 MB -- instead of 2 n hash slots; anything bigger hashes.
constexpr long long VOX_DENSE_MAX = 1ll << 21;
static bool table_dense(long long agents_x_cells) {
    static const bool dense_ok = [] { const char* e = getenv("HEAL_VOX_DENSE"); return !(e && e[0] == '0'); }();
    return dense_ok && agents_x_cells > 0 && agents_x_cells <= VOX_DENSE_MAX;
}
static uint32_t table_slots(int n, long long agents_x_cells) {
    return table_dense(agents_x_cells) ? (uint32_t)agents_x_cells : table_cap(n);
}

static bool carve(Arena& a, int n, int cap, int P, long long agents_x_cells, VoxWs& w) {
    w.tcap = table_slots(n, agents_x_cells);
    w.tmin = a.take<uint32_t>(w.tcap);
    w.tcnt = a.take<uint32_t>(w.tcap);
    w.tseg = a.take<uint32_t>(w.tcap);
    w.tkey = a.take<uint32_t>(table_dense(agents_x_cells) ? 1 : w.tcap);
    w.tile_pub = a.take<unsigned long long>(ceil_div(n, VOX_TILE) + 1);
    w.part_pub = a.take<unsigned long long>(VOX_MAX_BATCH + 1);
    w.meta = a.take<int>(64);
    w.clean_bytes = (size_t)((char*)w.meta - (char*)w.tmin);
    w.cand = a.take<uint32_t>(P > 64 ? (size_t)cap * P : 1);
    w.tvid = a.take<uint32_t>(P > 64 ? w.tcap : 1);
    w.st = a.take<int2>(n);
    w.seg = a.take<uint32_t>(n);
    w.row_seg = a.take<int>(cap);
    w.row_cnt = a.take<int>(cap);
    w.row_slot = a.take<int>(cap);
    return a.ok();
}

// the whole chain; outputs as documented in include/heal_amd.h
static int voxelize_chain(const float4* pts, const VoxBatch& vb, const VoxGrid& g, uint32_t cells, int P, int max_voxels, int cap,
                          const int32_t* row_offset, float* voxels, int32_t* coords, int32_t* num_points, int32_t* offsets_out,
                          int32_t* n_voxels_out, int32_t* row_offset_next, void* ws, size_t ws_bytes, int tables_clean, hipStream_t s,
                          const char* who) {
    const int n = vb.pt_off[vb.B];
    const int n_tiles = ceil_div(n, VOX_TILE);
    HEAL_REQUIRE(n_tiles <= VOX_MAX_TILES, "%s: at most %d points per call (got %d)", who, VOX_MAX_TILES * VOX_TILE, n);
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "%s: workspace must be 256-B aligned", who);
    Arena a(ws, ws_bytes);
    VoxWs w;
    // the cell map if the grid qualifies AND the caller sized the workspace for it (agents_x_cells of the size query), else the hash grid
    long long axc = (long long)vb.B * cells;
    if (table_dense(axc) && !carve(a, n, cap, P, axc, w)) { axc = 0; a = Arena(ws, ws_bytes); }
    if (!table_dense(axc)) {
        axc = 0;
        HEAL_REQUIRE(carve(a, n, cap, P, 0, w), "%s: workspace too small (%zu < %zu)", who, ws_bytes, a.off);
    }
    const int nb = ceil_div(n, 256);
    if (!tables_clean) {       // a workspace of unknown contents: records / publication words all-ones, meta zero (one fill launch each)
        HEAL_FILL(w.tmin, 0xFF, w.clean_bytes, s);
        HEAL_FILL(w.meta, 0, 64 * sizeof(int), s);
    }
    if (table_dense(axc))
        k_voxb_insert<true><<<nb, 256, 0, s>>>(pts, vb, g, cells, w.tkey, w.tmin, w.tcnt, w.tcap - 1, w.st, w.meta);
    else
        k_voxb_insert<false><<<nb, 256, 0, s>>>(pts, vb, g, cells, w.tkey, w.tmin, w.tcnt, w.tcap - 1, w.st, w.meta);
    const VoxTab tb{table_dense(axc) ? nullptr : w.tkey, w.tmin, w.tcnt, w.tseg};
    k_vox_assign<<<n_tiles, 256, 0, s>>>(pts, vb, g, w.st, w.tmin, w.tcnt, w.tseg, w.tile_pub, w.part_pub, n_tiles, max_voxels, row_offset,
                                         P > 64 ? w.tvid : nullptr, w.row_seg, w.row_cnt, w.row_slot, coords, offsets_out, n_voxels_out,
                                         row_offset_next, w.meta, P > 64 ? nullptr : w.seg);
    float4* vox4 = reinterpret_cast<float4*>(voxels);
    if (P > 64) {
        HEAL_FILL(w.cand, 0xFF, (size_t)cap * P * sizeof(uint32_t), s);
        k_vox_candidates<<<nb, 256, 0, s>>>(w.st, w.tvid, n, P, w.cand);
        k_vox_write<<<(unsigned)(((long long)cap * P + 255) / 256), 256, 0, s>>>(pts, w.cand, w.meta, P, vox4, num_points);
        k_vox_clean<<<ceil_div(n, 256), 256, 0, s>>>(tb, w.st, w.row_slot, w.tile_pub, w.part_pub, n_tiles, n, w.meta);
    } else {
#define HEAL_VSW(G_) k_vox_select_write<G_><<<ceil_div(cap, 256 / G_), 256, 0, s>>>(pts, w.seg, w.row_seg, w.row_cnt, w.meta, P, vox4, num_points, n, \
                                                                                  tb, w.st, w.row_slot, w.tile_pub, w.part_pub, n_tiles)
        if (P > 32) HEAL_VSW(64);
        else if (P > 16) HEAL_VSW(32);
        else if (P > 8) HEAL_VSW(16);
        else if (P > 4) HEAL_VSW(8);
        else if (P > 2) HEAL_VSW(4);
        else if (P > 1) HEAL_VSW(2);
        else HEAL_VSW(1);
#undef HEAL_VSW
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}

static bool vox_grid(const float* range_host, const float* voxel_size_host, VoxGrid& g, int64_t& cells) {
    cells = 1;
    for (int j = 0; j < 3; ++j) {
        g.rmin[j] = range_host[j];
        g.vsize[j] = voxel_size_host[j];
        // grid = round((max-min)/size), computed like numpy does on the python floats (fp64)
        const double gs = ((double)range_host[3 + j] - (double)range_host[j]) / (double)voxel_size_host[j];
        g.grid[j] = (int)__builtin_rint(gs);
        if (g.grid[j] < 1) return false;
        cells *= g.grid[j];
    }
    return true;
}

}  // namespace heal

using namespace heal;

extern "C" size_t heal_voxelize_workspace(int n_points, int max_points, int max_voxels, long long agents_x_cells) {
    if (n_points < 1) n_points = 1;
    int cap = n_points < max_voxels ? n_points : max_voxels;
    if (cap < 1) cap = 1;
    Arena a(nullptr, 0);
    VoxWs w;
    carve(a, n_points, cap, max_points < 1 ? 1 : max_points, agents_x_cells, w);
    return a.off + 256;
}

namespace heal {
struct PointMask {
    float lo[3], hi[3];
    int use_range, use_ego;
};

__global__ __launch_bounds__(256) void k_mask_points(const float4* pts, int n, PointMask m, float4* out) {  // may alias
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    bool keep = true;
    if (m.use_range)  // pcd_utils.py:58-63 (a NaN coordinate fails every comparison: dropped, like numpy)
        keep = p.x > m.lo[0] && p.x < m.hi[0] && p.y > m.lo[1] && p.y < m.hi[1] && p.z > m.lo[2] && p.z < m.hi[2];
    if (m.use_ego) {  // pcd_utils.py:84-86
        const bool body = p.x >= -1.95f && p.x <= 2.95f && p.y >= -1.1f && p.y <= 1.1f;
        keep = keep && !body;
    }
    const float nan = __int_as_float(0x7fc00000);
    out[i] = keep ? p : float4{nan, nan, nan, nan};
}
}  // namespace heal

extern "C" int heal_mask_points(const float* points, int n_points, const float* range_host, int mask_ego, float* out,
                                void* stream) {
    HEAL_REQUIRE(n_points >= 0, "mask_points: bad size");
    if (n_points == 0) return 0;
    HEAL_REQUIRE(points && out, "mask_points: null pointer");
    HEAL_REQUIRE((((uintptr_t)points | (uintptr_t)out) & 15) == 0, "mask_points: points / out must be 16-byte aligned");
    heal::PointMask m;
    m.use_range = range_host != nullptr;
    m.use_ego = mask_ego != 0;
    for (int k = 0; k < 3; ++k) {
        m.lo[k] = range_host ? range_host[k] : 0.f;
        m.hi[k] = range_host ? range_host[3 + k] : 0.f;
    }
    heal::k_mask_points<<<heal::ceil_div(n_points, 256), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(points), n_points, m, reinterpret_cast<float4*>(out));
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_voxelize(const float* points, int n_points, const float* range_host,
                             const float* voxel_size_host, int max_points, int max_voxels,
                             int batch_idx, float* voxels, int32_t* coords, int32_t* num_points,
                             int32_t* n_voxels, const int32_t* row_offset, int32_t* row_offset_next, void* ws,
                             size_t ws_bytes, int tables_clean, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_points >= 0 && max_points >= 1 && max_voxels >= 1, "voxelize: bad sizes");
    HEAL_REQUIRE(n_voxels != nullptr, "voxelize: n_voxels is NULL");
    if (n_points == 0) {
        HEAL_FILL(n_voxels, 0, sizeof(int), s);
        if (row_offset_next) {
            if (row_offset) { if (heal::copy_word(row_offset_next, row_offset, s)) return 1; }
            else HEAL_FILL(row_offset_next, 0, sizeof(int), s);
        }
        return 0;
    }
    VoxGrid g;
    int64_t cells;
    HEAL_REQUIRE(vox_grid(range_host, voxel_size_host, g, cells), "voxelize: empty grid");
    HEAL_REQUIRE(cells < 0xFFFFFFFFll, "voxelize: grid has too many cells (%lld)", (long long)cells);
    VoxBatch vb;
    vb.B = 1; vb.pt_off[0] = 0; vb.pt_off[1] = n_points; vb.label0 = batch_idx;
    for (int b = 2; b <= VOX_MAX_BATCH; ++b) vb.pt_off[b] = n_points;
    const int cap = n_points < max_voxels ? n_points : max_voxels;
    return voxelize_chain(reinterpret_cast<const float4*>(points), vb, g, (uint32_t)cells, max_points, max_voxels, cap, row_offset,
                          voxels, coords, num_points, nullptr, n_voxels, row_offset_next, ws, ws_bytes, tables_clean, s, "voxelize");
}

extern "C" size_t heal_voxelize_batch_workspace(int n_points_total, int n_agents, int max_points, int max_voxels,
                                                long long agents_x_cells) {
    if (n_points_total < 1) n_points_total = 1;
    (void)n_agents;
    Arena a(nullptr, 0);
    VoxWs w;
    // rows: sum_b min(n_b, max_voxels) <= min(n_total, n_agents * max_voxels)
    long long cap = (long long)(n_agents < 1 ? 1 : n_agents) * (max_voxels < 1 ? 1 : max_voxels);
    if (cap > n_points_total) cap = n_points_total;
    carve(a, n_points_total, (int)cap, max_points < 1 ? 1 : max_points, agents_x_cells, w);
    return a.off + 256;
}

extern "C" int heal_voxelize_batch(const float* points, const int32_t* point_offsets_host, int n_agents,
                                   const float* range_host, const float* voxel_size_host, int max_points,
                                   int max_voxels, float* voxels, int32_t* coords, int32_t* num_points,
                                   int32_t* row_offsets, void* ws, size_t ws_bytes, int tables_clean, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_agents >= 1 && n_agents <= VOX_MAX_BATCH, "voxelize_batch: 1..%d agents per call", VOX_MAX_BATCH);
    HEAL_REQUIRE(max_points >= 1 && max_voxels >= 1 && row_offsets != nullptr, "voxelize_batch: bad arguments");
    VoxBatch vb;
    vb.B = n_agents; vb.label0 = 0;
    for (int b = 0; b <= VOX_MAX_BATCH; ++b) {
        vb.pt_off[b] = point_offsets_host[b <= n_agents ? b : n_agents];
        HEAL_REQUIRE(b == 0 ? vb.pt_off[0] == 0 : vb.pt_off[b] >= vb.pt_off[b - 1], "voxelize_batch: offsets must ascend from 0");
    }
    const int n = vb.pt_off[n_agents];
    if (n == 0) {
        HEAL_FILL(row_offsets, 0, sizeof(int) * (size_t)(n_agents + 1), s);
        return 0;
    }
    VoxGrid g;
    int64_t cells;
    HEAL_REQUIRE(vox_grid(range_host, voxel_size_host, g, cells), "voxelize_batch: empty grid");
    HEAL_REQUIRE(cells * n_agents < 0xFFFFFFFFll, "voxelize_batch: agents x cells exceeds 32-bit keys");
    int cap = 0;  // rows of the collated outputs: sum of min(n_b, max_voxels)
    for (int b = 0; b < n_agents; ++b) cap += (vb.pt_off[b + 1] - vb.pt_off[b]) < max_voxels ? (vb.pt_off[b + 1] - vb.pt_off[b]) : max_voxels;
    return voxelize_chain(reinterpret_cast<const float4*>(points), vb, g, (uint32_t)cells, max_points, max_voxels, cap, nullptr,
                          voxels, coords, num_points, row_offsets, nullptr, nullptr, ws, ws_bytes, tables_clean, s, "voxelize_batch");
}
