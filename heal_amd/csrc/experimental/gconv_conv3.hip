// Back half of a ResNeXt bottleneck in ONE kernel: the 32-group 3x3 convolution (conv2 + bn2 + relu) and the pointwise
// convolution that follows it (conv3 + bn3 + identity + relu) -- opencood/models/sub_modules/resblock.py:110-121, groups = 32, stride 1,
// BatchNorms folded into weights / biases by the caller.
//
// Why: at the PyramidFusion widths of the large maps (128 channels at 256 x 256 x 5 agents, 256 at 128 x 128) both kernels are HBM-bound
// and the 2C-wide intermediate between them is written once and read once -- 336 of the 590 MB the two launches move at level 1.
// Here it never leaves the CU:
//   * block = 8 x 32 output pixels, 4 waves; the block walks the 16-channel SUPER-GROUPS of the intermediate one after the other;
//   * per super-group the 10 x 34 input patch of its 16 channels is staged exactly as in k_gconv_small (csrc/gconv_small.hip: channel-
//     interleaved [row][pixel][16 channels], 16-B global loads, zero padding at the LDS store) and the grouped convolution runs on the
//     16-block v_mfma_f32_4x4x1_16b_f32 (lane l: output channel l % 16, pixels (l / 16) * 4 + {0..3} of a 16-pixel segment);
//   * the wave adds bias + ReLU and parks its 16 channels x 64 pixels in its OWN LDS slice [16][64 + 16] -- the pixels a wave produces
//     (its two rows of the tile) are exactly the pixels it consumes next, so this hand-over needs no block barrier;
//   * conv3 partial product on v_mfma_f32_16x16x4_f32: acc3[Cout x 64 pixels] += W3[:, super-group] x mid, A = W3 pre-laid in fragment
//     order (ops.mfma_a_fragments: one coalesced 256-B load per fragment, L2-resident), B = the parked tile (row stride 80 words:
//     the four k-rows of a fragment read fall in disjoint bank groups); the accumulators (Cout / 16 x 4 tiles per wave) stay in
//     registers across the super-groups;
//   * epilogue: + b3 + identity, ReLU, stores (64-B runs per channel); the identity values are requested before the last MFMAs.
// Traffic at level 1 per launch: 168 MB read (+ halo from L2) + 84 MB identity + 84 MB written -- what the grouped convolution ALONE
// moved before.
#include <stdlib.h>
#include "../common.h"
#include "../../../include/heal_amd.h"
#include "../../../include/heal_amd_experimental.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CG, int COUT>
__global__ __launch_bounds__(512) void k_gconv_conv3(const float* __restrict__ x /*[n, WIDTH, H, W]*/, const float* __restrict__ wq,
                                                    const float* __restrict__ b2, const float* __restrict__ w3f,
                                                    const float* __restrict__ b3, const float* __restrict__ res /*[n, COUT, H, W]*/,
                                                    int WIDTH, int H, int W, int tiles_x, int relu, float* __restrict__ y,
                                                    long long* __restrict__ dbg) {
    static_assert(CG == 4 || CG == 8, "4 or 8 channels per group");
    static_assert(COUT % 16 == 0 && COUT <= 128, "output channels");
    constexpr int TH = 8, TW = 32, NSEG = 2, RW = 2;         // 8 x 32 output pixels, wave pair w owns rows 2w, 2w + 1
    constexpr int PR = TH + 2, PC = TW + 2;                  // patch rows / columns
    constexpr int NSTEP = 9 * CG, NB = CG / 4;
    constexpr int MT = COUT / 16, NT = RW * NSEG;            // conv3 tiles per consumer wave: MT x 4
    constexpr int MS = RW * TW + 16;                         // row stride of the parked tile (words): 80 = 16 (mod 64)
    __shared__ float4 sP[2][PR * PC * 4];                    // double-buffered: [row][pixel][unit = channel / 4]
    __shared__ float4 sW[2][NSTEP * 16 / 4];                 // [tap][ci][output channel of the super-group]
    __shared__ float sM[2][4][16 * MS];                      // double-buffered, per wave pair: [channel of the super-group][its 64 pixels]
    const Block3 bk = xcd_block();                           // x: tile, y: image
    const int n = bk.y;
    const int ty = bk.x / tiles_x, tx = bk.x - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int gy0 = oy0 - 1, gx0 = ox0;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave8 < 4;                         // waves 0..3: staging + grouped conv; waves 4..7: conv3 + epilogue
    const int wave = wave8 & 3, t = threadIdx.x & 255, l = threadIdx.x & 63;
    const int lk = l >> 4, ln = l & 15;
    const size_t HW = (size_t)H * W;
    const int nsg = WIDTH / 16;
    int dbg_i = 0;
#define GC3_STAMP() do { if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && l == 0 && wave == 0) dbg[(producer ? 0 : 64) + dbg_i++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

    constexpr int N_IN = PR * 8 * 4, IT_IN = (N_IN + 255) / 256;   // interior quads: 4 channels x 4 pixels per item
    constexpr int N_HA = PR * 2 * 4;                               // halo columns (left, right)
    static_assert(N_HA <= 256, "staging shape");
    constexpr int N_W4 = NSTEP * 16 / 4, IT_W = (N_W4 + 255) / 256;

    // WAVE SPECIALISATION.  Iteration i: the producer waves run the grouped convolution of super-group i out of patch buffer i & 1 and
    // park its 16 channels in sM[i & 1], store the patch of super-group i + 1 (requested one iteration earlier) into the other patch
    // buffer and request the one after; the consumer waves multiply the tile parked in iteration i - 1 with their slice of W3.  One
    // block barrier per iteration; a SIMD holds one producer and one consumer wave, whose LDS-heavy and MFMA-heavy phases overlap.
    if (producer) {
        const int p = lk * 4 + (l & 3);                                // gconv: output pixel within a 16-pixel segment
        const int unit0 = (((l >> 2) & 3) * 4 / CG) * NB;              // first 16-B unit of the lane's group within a pixel
        // TWO sets of staging registers that alternate: a patch is requested two iterations before it is stored (one iteration of
        // compute does not cover the load latency of a loaded memory system: 3 us per iteration with one set)
        struct Stage { float4 vin[IT_IN][4], vha, vw[IT_W]; unsigned ok_bits; };
        Stage stA, stB;
        auto request = [&](int sg, Stage& S) {
            float4 (&vin)[IT_IN][4] = S.vin; float4& vha = S.vha; float4 (&vw)[IT_W] = S.vw; unsigned& ok_bits = S.ok_bits;
            const float* __restrict__ xin = x + ((size_t)n * WIDTH + (size_t)sg * 16) * HW;
            ok_bits = 0;
#pragma unroll
            for (int it = 0; it < IT_IN; ++it) {
                const int u = min(t + 256 * it, N_IN - 1);
                const int gq = u & 3, quad = (u >> 2) & 7, row = u >> 5;
                const int gy = gy0 + row, gx = gx0 + quad * 4;
                const bool ok = gy >= 0 && gy < H && gx < W;               // W % 4 == 0: a quad is all-in or all-out
                ok_bits |= ok ? (1u << it) : 0u;
                const float* src = xin + (size_t)(gq * 4) * HW + (ok ? (size_t)gy * W + gx : 0);
                vin[it][0] = *reinterpret_cast<const float4*>(src);
                vin[it][1] = *reinterpret_cast<const float4*>(src + HW);
                vin[it][2] = *reinterpret_cast<const float4*>(src + 2 * HW);
                vin[it][3] = *reinterpret_cast<const float4*>(src + 3 * HW);
            }
            {
                const int u = min(t, N_HA - 1);
                const int gq = u & 3, side = (u >> 2) & 1, row = u >> 3;
                const int gy = gy0 + row, gx = side ? gx0 + 32 : gx0 - 1;
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
                ok_bits |= ok ? 256u : 0u;
                const float* src = xin + (size_t)(gq * 4) * HW + (ok ? (size_t)gy * W + gx : 0);
                vha = make_float4(src[0], src[HW], src[2 * HW], src[3 * HW]);
            }
#pragma unroll
            for (int it = 0; it < IT_W; ++it)
                vw[it] = reinterpret_cast<const float4*>(wq + (size_t)sg * NSTEP * 16)[min(t + 256 * it, N_W4 - 1)];
        };
        auto deposit = [&](int buf, const Stage& S) {
            const float4 (&vin)[IT_IN][4] = S.vin; const float4& vha = S.vha; const float4 (&vw)[IT_W] = S.vw;
            const unsigned ok_bits = S.ok_bits;
            float4* __restrict__ P = sP[buf];
#pragma unroll
            for (int it = 0; it < IT_W; ++it) sW[buf][min(t + 256 * it, N_W4 - 1)] = vw[it];
#pragma unroll
            for (int it = 0; it < IT_IN; ++it) {
                const int u = min(t + 256 * it, N_IN - 1);
                const int gq = u & 3, quad = (u >> 2) & 7, row = u >> 5;
                const bool ok = (ok_bits >> it) & 1u;
                float4* d = P + (row * PC + 1 + quad * 4) * 4 + gq;
                const float4 a = vin[it][0], b = vin[it][1], c = vin[it][2], e = vin[it][3];
                // (component-wise selects: a select between two float4 VALUES goes through private memory with this compiler)
                d[0] = make_float4(ok ? a.x : 0.f, ok ? b.x : 0.f, ok ? c.x : 0.f, ok ? e.x : 0.f);
                d[4] = make_float4(ok ? a.y : 0.f, ok ? b.y : 0.f, ok ? c.y : 0.f, ok ? e.y : 0.f);
                d[8] = make_float4(ok ? a.z : 0.f, ok ? b.z : 0.f, ok ? c.z : 0.f, ok ? e.z : 0.f);
                d[12] = make_float4(ok ? a.w : 0.f, ok ? b.w : 0.f, ok ? c.w : 0.f, ok ? e.w : 0.f);
            }
            {
                const int u = min(t, N_HA - 1);
                const int gq = u & 3, side = (u >> 2) & 1, row = u >> 3;
                const bool okh = ok_bits & 256u;
                P[(row * PC + (side ? PC - 1 : 0)) * 4 + gq] =
                    make_float4(okh ? vha.x : 0.f, okh ? vha.y : 0.f, okh ? vha.z : 0.f, okh ? vha.w : 0.f);
            }
        };
        request(0, stA);
        deposit(0, stA);
        request(min(1, nsg - 1), stB);       // (past the end: the last super-group again, never stored)
        request(min(2, nsg - 1), stA);
        __syncthreads();
        for (int sg2 = 0; sg2 <= nsg; sg2 += 2) {
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            const int sg = sg2 + par;
            if (sg > nsg) break;
            Stage& HAVE = par == 0 ? stB : stA;      // holds super-group sg + 1
            GC3_STAMP();
            if (sg < nsg) {
                const int buf = sg & 1;
                const float4* __restrict__ abase = sP[buf] + ((wave * RW) * PC + p) * 4 + unit0;
                const float bv2 = b2[sg * 16 + ln];
                // TWO accumulators per (row, segment), even / odd input channels: with RW * NSEG = 4 chains a dependent 4x4x1 MFMA came
                // every 2-4 instructions and the pipe waited for its own results (s_nop padding); summed when the tile is parked
                f32x4 acc[RW][NSEG], accb[RW][NSEG];
#pragma unroll
                for (int o = 0; o < RW; ++o)
#pragma unroll
                    for (int s = 0; s < NSEG; ++s) { acc[o][s] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[o][s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                float wreg[9 * CG];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int ci = 0; ci < CG; ++ci)
                        wreg[tap * CG + ci] = reinterpret_cast<const float*>(sW[buf])[(tap * CG + ci) * 16 + ln];
                // operand rows double-buffered in registers: the reads of step (ry, dx) + 1 are issued before the MFMAs of step (ry, dx)
                float4 av[2][NSEG][NB];
#pragma unroll
                for (int s = 0; s < NSEG; ++s)
#pragma unroll
                    for (int h = 0; h < NB; ++h) av[0][s][h] = abase[(s * 16) * 4 + h];
#pragma unroll
                for (int st = 0; st < 3 * (RW + 2); ++st) {
                    const int ry = st / 3, dx = st - ry * 3;
                    if (st + 1 < 3 * (RW + 2)) {
                        const int ry1 = (st + 1) / 3, dx1 = (st + 1) - ry1 * 3;
#pragma unroll
                        for (int s = 0; s < NSEG; ++s)
#pragma unroll
                            for (int h = 0; h < NB; ++h) av[(st + 1) & 1][s][h] = abase[(ry1 * PC + s * 16 + dx1) * 4 + h];
                    }
                    __builtin_amdgcn_sched_barrier(0);     // keep the prefetch above this step's MFMAs
#pragma unroll
                    for (int ci = 0; ci < CG; ++ci) {
#pragma unroll
                        for (int o = 0; o < RW; ++o) {
                            const int dy = ry - o;
                            if (dy < 0 || dy > 2) continue;       // compile-time after unrolling
#pragma unroll
                            for (int s = 0; s < NSEG; ++s) {
                                const float4 q = av[st & 1][s][ci >> 2];
                                const float a = (ci & 3) == 0 ? q.x : (ci & 3) == 1 ? q.y : (ci & 3) == 2 ? q.z : q.w;
                                if (ci & 1) accb[o][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, wreg[(dy * 3 + dx) * CG + ci], accb[o][s], 0, 0, 0);
                                else acc[o][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, wreg[(dy * 3 + dx) * CG + ci], acc[o][s], 0, 0, 0);
                            }
                        }
                    }
                }
                // the next super-group's patch: registers -> the other buffer (its last readers passed the previous barrier), then the
                // request for the one after -- HERE, while the matrix pipe still works through this super-group's MFMAs (nothing
                // below needs their results before the tile is parked)
                GC3_STAMP();
                if (sg + 1 < nsg) deposit(buf ^ 1, HAVE);
                request(min(sg + 3, nsg - 1), HAVE);     // unconditional (clamped): the load queue stays countable
                GC3_STAMP();
#pragma unroll
                for (int o = 0; o < RW; ++o)
#pragma unroll
                    for (int s = 0; s < NSEG; ++s) acc[o][s] += accb[o][s];
                // bias + ReLU, park 16 channels x 64 pixels: lane = (channel ln, pixels o * 32 + s * 16 + lk * 4 + {0..3})
                float* __restrict__ sm = sM[buf][wave];
#pragma unroll
                for (int o = 0; o < RW; ++o)
#pragma unroll
                    for (int s = 0; s < NSEG; ++s) {
                        float4 v = make_float4(fmaxf(acc[o][s][0] + bv2, 0.f), fmaxf(acc[o][s][1] + bv2, 0.f),
                                               fmaxf(acc[o][s][2] + bv2, 0.f), fmaxf(acc[o][s][3] + bv2, 0.f));
                        *reinterpret_cast<float4*>(&sm[ln * MS + o * TW + s * 16 + lk * 4]) = v;
                    }
            }
            __syncthreads();
          }
        }
        return;
    }

    // ---- consumer waves: acc3[m][q] += W3[m-tile, k-steps of the super-group] x parked tile; epilogue ---------------------------------
    f32x4 acc3[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc3[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a3[MT][4], a3n[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a3[m][ks] = w3f[((size_t)m * (WIDTH / 4) + ks) * 64 + l];
    __syncthreads();                                              // (matches the producers' first barrier)
    for (int sg = 0; sg <= nsg; ++sg) {
        GC3_STAMP();
        if (sg >= 1) {
            const int k = sg - 1;                                    // the super-group parked in the previous iteration
            const float* __restrict__ sm = sM[k & 1][wave];
            const int kn = min(k + 1, nsg - 1);                      // next fragments: requested before the MFMAs, consumed after
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a3n[m][ks] = w3f[((size_t)m * (WIDTH / 4) + kn * 4 + ks) * 64 + l];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float bf[NT];
#pragma unroll
                for (int q = 0; q < NT; ++q) bf[q] = sm[(ks * 4 + lk) * MS + q * 16 + ln];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc3[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[m][ks], bf[q], acc3[m][q], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a3[m][ks] = a3n[m][ks];
            GC3_STAMP();
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = lk * 4 + r][col = ln] of (m, q): channel m * 16 + lk * 4 + r, pixel q of the wave pair ---------------------
    float* __restrict__ yout = y + (size_t)n * COUT * HW;
    const float* __restrict__ rin = res ? res + (size_t)n * COUT * HW : nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float rr[NT][4];
        if (rin) {
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int oy = min(oy0 + wave * RW + (q >> 1), H - 1), ox = min(ox0 + (q & 1) * 16 + ln, W - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[q][r] = rin[(size_t)(m * 16 + lk * 4 + r) * HW + (size_t)oy * W + ox];
            }
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int oy = oy0 + wave * RW + (q >> 1), ox = ox0 + (q & 1) * 16 + ln;
            if (oy >= H || ox >= W) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m * 16 + lk * 4 + r;
                float v = acc3[m][q][r] + b3[co];
                if (rin) v += rr[q][r];
                if (relu) v = fmaxf(v, 0.f);
                yout[(size_t)co * HW + (size_t)oy * W + ox] = v;
            }
        }
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_gconv_conv3_supported(int width, int group_channels, int cout, int H, int W) {
    return (group_channels == 4 || group_channels == 8) && width % 16 == 0 && (cout == 64 || cout == 128) && W % 4 == 0 && H >= 1 &&
           W >= 4;
}

// y = act( W3 . relu( gconv3x3(x; 32 groups) + b2 ) + b3 (+ residual) ).  x [n, width, H, W]; weight_q: the grouped weights in the
// layout of heal_grouped_small_conv3x3 ([super-group][tap][ci][16]); w3_frag: W3 [cout, width] in MFMA A-fragment order
// (frag[mt][ks][lane] = W3[mt * 16 + (lane & 15)][ks * 4 + (lane >> 4)]); residual [n, cout, H, W] or NULL; y [n, cout, H, W].
extern "C" int heal_gconv_conv3(const float* x, const float* weight_q, const float* b2, const float* w3_frag, const float* b3,
                                const float* residual, int n, int width, int group_channels, int cout, int H, int W, int relu,
                                float* y, void* stream) {
    HEAL_REQUIRE(heal_gconv_conv3_supported(width, group_channels, cout, H, W),
                 "gconv_conv3: needs 4 | 8 channels per group, width %% 16 == 0, Cout 64 | 128, W %% 4 == 0 (got width=%d cg=%d cout=%d W=%d)",
                 width, group_channels, cout, W);
    HEAL_REQUIRE(x && weight_q && b2 && w3_frag && b3 && y && ((uintptr_t)x & 15) == 0, "gconv_conv3: bad pointer");
    HEAL_REQUIRE(n >= 1 && n <= 65535, "gconv_conv3: grid limit");
    const int tiles_x = ceil_div(W, 32), tiles_y = ceil_div(H, 8);
    const dim3 grid(tiles_x * tiles_y, n);
    hipStream_t s = (hipStream_t)stream;
#define HEAL_GC3(CG_, CO_)                                                                                             \
    if (group_channels == CG_ && cout == CO_)                                                                          \
        HEAL_LAUNCH_EV((k_gconv_conv3<CG_, CO_>), grid, dim3(512), 0, s, x, weight_q, b2, w3_frag, b3, residual, width, H, W,    \
                       tiles_x, relu, y, dbg);
    static long long* dbg = nullptr;
    static const bool want_dbg = getenv("HEAL_GC3_DBG") != nullptr;
    if (want_dbg && !dbg) { HEAL_HIP(hipMalloc(&dbg, 128 * 8)); }
    if (want_dbg) HEAL_FILL(dbg, 0, 128 * 8, s);
    HEAL_GC3(4, 64) HEAL_GC3(8, 128) HEAL_GC3(4, 128) HEAL_GC3(8, 64)
#undef HEAL_GC3
    HEAL_LAUNCH_CHECK();
    if (want_dbg) {
        long long h[128];
        HEAL_HIP(hipStreamSynchronize(s));
        HEAL_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[gc3] producer stamps (s_memtime ticks = shader cycles):");
        for (int i = 0; i < 40 && h[i]; ++i) fprintf(stderr, " %lld", h[i] - h[0]);
        fprintf(stderr, "\n[gc3] consumer stamps:");
        for (int i = 0; i < 24 && h[64 + i]; ++i) fprintf(stderr, " %lld", h[64 + i] - h[0]);
        fprintf(stderr, "\n");
    }
    return 0;
}
