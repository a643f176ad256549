// 32-group 3x3 convolution of the ResNeXt bottlenecks for SMALL groups (4 or 8 channels per group) on the matrix cores.
// Reference: opencood/models/sub_modules/resblock.py:90-98,110-112 (conv2 + bn2 + relu, groups = 32) at the PyramidFusion
// widths 128 (4 channels per group, 256 x 256 maps) and 256 (8 per group, 128 x 128 maps).
//
// A 16 x 16 MFMA m-tile would spend 3/4 (1/2) of its multiplies on the zeros of a block-diagonal weight.  gfx950 still has
// the CDNA multi-block form v_mfma_f32_4x4x1_16b_f32: SIXTEEN independent 4x4 outer products per instruction at the full
// fp32 matrix rate (512 FLOP / 8 cycles / SIMD) -- a 4-output-channel x 4-pixel x 1-input-channel product per block, which is
// exactly a group of 4 (or half a group of 8).  Operand layout (measured, scripts/probes/mfma4_probe.hip):
//     D[lane l][reg r] += A[lane 4*(l/4) + r] * B[lane l]            block = l / 4, i = r, j = l % 4.
// Mapping used here, for a 16-channel super-group sg and a run of 16 output pixels:
//     block b = pq * 4 + hq   (pq = pixel quad 0..3, hq = which 4 output channels of the 16)
//     A = activations (row i of a block = pixel pq * 4 + i), B = weights (column j = output channel hq * 4 + j):
//     -> lane l supplies pixel p = (l / 16) * 4 + l % 4 and the weight of output channel l % 16 (36 or 72 registers: every
//        (tap, input channel) of its group) and receives pixels (l / 16) * 4 + {0..3} of output channel l % 16: one 16-B store;
//     -> its A operand is input channel ci of ITS group at pixel p (+ tap offset): the patch is staged in LDS channel-interleaved,
//        [row][pixel][16 channels], so one ds_read_b128 returns the B operands of four k-steps, and the 64 lanes of a read
//        cover 16 pixels x 64 B = 1 KiB contiguous (conflict-free for 4 channels per group; 2-way for 8, where the two halves of a
//        group read the same 16 bytes).  A patch row read serves the three output rows it touches.
// One block = one super-group x a 16 x 32 output tile; wave w owns rows 4w .. 4w+3 (8 accumulators).  The patch (18 x 34 pixels
// x 16 channels, 38 KiB) is staged with 16-B global loads along the rows (4 channels x 4 pixels per thread, transposed in
// registers into four 16-B LDS stores), unconditional and clamped with the zero padding applied at the store.
// Bytes: every input element is read once per tile (+ halo 1.2x, from L2), every output written once: HBM-bound at these
// widths (0.33 GB per call at 128 channels x 256^2 x 5 agents).
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CG, int TH, int STRIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_gconv_small(
    const float* __restrict__ x, const float* __restrict__ wq, const float* __restrict__ bias, int C, int H, int W, int Ho,
    int Wo, int tiles_x, int relu, float* __restrict__ y) {
    static_assert(CG == 4 || CG == 8 || CG == 16, "4, 8 or 16 channels per group");
    static_assert(TH == 8 || TH == 16, "tile height");
    static_assert(STRIDE == 1 || STRIDE == 2, "stride");
    // output tile TH x TW; input patch rows STRIDE*oy0 - 1 .., columns STRIDE*ox0 - 1 .. (stride 2 needs no right / bottom halo)
    constexpr int TW = STRIDE == 1 ? 32 : 16, NSEG = TW / 16, RW = TH / 4;   // RW output rows per wave
    constexpr int PR = STRIDE * TH + 3 - STRIDE, PC = STRIDE * TW + 3 - STRIDE;
    constexpr int NSIDE = 3 - STRIDE;        // halo columns: left and right | left only
    constexpr int NSTEP = 9 * CG;            // (tap, ci) k-steps of one group
    constexpr int KH = CG == 16 ? 2 : 1;     // passes over the patch: 16 channels per group run as two halves of 8 input channels
    constexpr int CGH = CG / KH;             //   (the weight operands of a pass stay in registers: 9 * CGH of them)
    constexpr int NB = CG / 4, NBH = CGH / 4;   // 16-B units per pixel of a group | read per (row, segment, dx) and pass
    __shared__ float4 sP[PR * PC * 4];       // [row][pixel][unit = channel / 4]
    __shared__ float4 sW[NSTEP * 16 / 4];    // [tap][ci][output channel of the super-group]
    const Block3 bk = xcd_block();           // x: tile, y: super-group, z: image
    const int sg = bk.y, n = bk.z;
    const int ty = bk.x / tiles_x, tx = bk.x - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int gy0 = oy0 * STRIDE - 1, gx0 = ox0 * STRIDE;          // patch row 0 | first interior column (16-B aligned)
    const int t = threadIdx.x, wave = t >> 6, l = t & 63;
    const size_t HW = (size_t)H * W, HWo = (size_t)Ho * Wo;
    const float* __restrict__ xin = x + ((size_t)n * C + (size_t)sg * 16) * HW;

    // ---- stage the patch: interior quads (16-B loads, 4 channels x 4 pixels per item: 8 quads per row) + the halo column(s);
    // unconditional clamped loads, all in flight together; the zero padding is applied at the LDS store -------------------
    constexpr int N_IN = PR * 8 * 4, IT_IN = (N_IN + 255) / 256;
    constexpr int N_HA = PR * NSIDE * 4;
    static_assert(N_HA <= 256 && STRIDE * TW == 32, "staging shape");
    float4 vin[IT_IN][4];
    unsigned ok_bits = 0;                                          // bit it: interior item valid; bit 8: halo item valid
#pragma unroll
    for (int it = 0; it < IT_IN; ++it) {
        const int u = min(t + 256 * it, N_IN - 1);                 // surplus threads repeat the last item (same data)
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
    float4 vha;
    {
        const int u = min(t, N_HA - 1);
        const int gq = u & 3, side = NSIDE == 2 ? (u >> 2) & 1 : 0, row = u / (4 * NSIDE);
        const int gy = gy0 + row, gx = side ? gx0 + 32 : gx0 - 1;
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        ok_bits |= ok ? 256u : 0u;
        const float* src = xin + (size_t)(gq * 4) * HW + (ok ? (size_t)gy * W + gx : 0);
        vha = make_float4(src[0], src[HW], src[2 * HW], src[3 * HW]);
    }
    // the super-group's weights [tap][ci][16]: one coalesced pass into LDS, from where every lane picks its column
    constexpr int N_W4 = NSTEP * 16 / 4, IT_W = (N_W4 + 255) / 256;
#pragma unroll
    for (int it = 0; it < IT_W; ++it) {
        const int i = min(t + 256 * it, N_W4 - 1);
        sW[i] = reinterpret_cast<const float4*>(wq + (size_t)sg * NSTEP * 16)[i];
    }
#pragma unroll
    for (int it = 0; it < IT_IN; ++it) {
        const int u = min(t + 256 * it, N_IN - 1);
        const int gq = u & 3, quad = (u >> 2) & 7, row = u >> 5;
        const bool ok = (ok_bits >> it) & 1u;
        float4* d = sP + (row * PC + 1 + quad * 4) * 4 + gq;
        const float4 a = vin[it][0], b = vin[it][1], c = vin[it][2], e = vin[it][3];
        // (component-wise selects: a select between two float4 VALUES goes through private memory with this compiler)
        d[0] = make_float4(ok ? a.x : 0.f, ok ? b.x : 0.f, ok ? c.x : 0.f, ok ? e.x : 0.f);
        d[4] = make_float4(ok ? a.y : 0.f, ok ? b.y : 0.f, ok ? c.y : 0.f, ok ? e.y : 0.f);
        d[8] = make_float4(ok ? a.z : 0.f, ok ? b.z : 0.f, ok ? c.z : 0.f, ok ? e.z : 0.f);
        d[12] = make_float4(ok ? a.w : 0.f, ok ? b.w : 0.f, ok ? c.w : 0.f, ok ? e.w : 0.f);
    }
    {
        const int u = min(t, N_HA - 1);
        const int gq = u & 3, side = NSIDE == 2 ? (u >> 2) & 1 : 0, row = u / (4 * NSIDE);
        const bool okh = ok_bits & 256u;
        sP[(row * PC + (side ? PC - 1 : 0)) * 4 + gq] =
            make_float4(okh ? vha.x : 0.f, okh ? vha.y : 0.f, okh ? vha.z : 0.f, okh ? vha.w : 0.f);
    }
    __syncthreads();
    const float bv0 = bias ? bias[sg * 16 + (l & 15)] : 0.f;
    const int p = (l >> 4) * 4 + (l & 3);                 // output pixel within a 16-pixel segment
    const int unit0 = (((l >> 2) & 3) * 4 / CG) * NB;     // first 16-B unit of the lane's group within a pixel
    const float4* __restrict__ abase = sP + ((wave * RW * STRIDE) * PC + p * STRIDE) * 4 + unit0;

    // ---- RW output rows x NSEG segments per wave; a patch-row read feeds every output row it touches ---------------------
    f32x4 acc[RW][NSEG];
#pragma unroll
    for (int o = 0; o < RW; ++o)
#pragma unroll
        for (int s = 0; s < NSEG; ++s) acc[o][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < KH; ++half) {
        // B operand: the weight of output channel l % 16 for every (tap, ci) of this pass, in registers
        float wreg[9 * CGH];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ci = 0; ci < CGH; ++ci)
                wreg[tap * CGH + ci] = reinterpret_cast<const float*>(sW)[(tap * CG + half * CGH + ci) * 16 + (l & 15)];
#pragma unroll
        for (int ry = 0; ry < STRIDE * (RW - 1) + 3; ++ry) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float4 av[NSEG][NBH];
#pragma unroll
                for (int s = 0; s < NSEG; ++s)
#pragma unroll
                    for (int h = 0; h < NBH; ++h) av[s][h] = abase[(ry * PC + s * 16 * STRIDE + dx) * 4 + half * NBH + h];
#pragma unroll
                for (int ci = 0; ci < CGH; ++ci) {
#pragma unroll
                    for (int o = 0; o < RW; ++o) {
                        const int dy = ry - STRIDE * o;
                        if (dy < 0 || dy > 2) continue;       // compile-time after unrolling
#pragma unroll
                        for (int s = 0; s < NSEG; ++s) {
                            const float4 q = av[s][ci >> 2];
                            const float a = (ci & 3) == 0 ? q.x : (ci & 3) == 1 ? q.y : (ci & 3) == 2 ? q.z : q.w;
                            acc[o][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, wreg[(dy * 3 + dx) * CGH + ci], acc[o][s], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- epilogue: D[l][r] = pixel (l / 16) * 4 + r of the segment, output channel l % 16 -> one 16-B store per tile ------
    float* __restrict__ yout = y + ((size_t)n * C + (size_t)sg * 16 + (l & 15)) * HWo;
#pragma unroll
    for (int o = 0; o < RW; ++o) {
        const int oy = oy0 + wave * RW + o;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int ox = ox0 + s * 16 + (l >> 4) * 4;
            if (oy >= Ho || ox >= Wo) continue;            // Wo % 4 == 0: a quad is all-in or all-out
            float4 v = make_float4(acc[o][s][0] + bv0, acc[o][s][1] + bv0, acc[o][s][2] + bv0, acc[o][s][3] + bv0);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(yout + (size_t)oy * Wo + ox) = v;
        }
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_grouped_small_conv3x3(const float* x, const float* weight_q, const float* bias, int n, int channels,
                                          int group_channels, int H, int W, int stride, int relu, float* y, void* stream) {
    HEAL_REQUIRE(group_channels == 4 || group_channels == 8 || group_channels == 16,
                 "grouped_small_conv3x3: 4, 8 or 16 channels per group (got %d)", group_channels);
    HEAL_REQUIRE(stride == 1 || stride == 2, "grouped_small_conv3x3: stride must be 1 or 2 (got %d)", stride);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    HEAL_REQUIRE(n >= 1 && channels >= 16 && channels % 16 == 0 && H >= 1 && W >= 4 && W % 4 == 0 && Wo % 4 == 0,
                 "grouped_small_conv3x3: needs channels %% 16 == 0, W %% 4 == 0 and an output width %% 4 == 0 (got C=%d W=%d)",
                 channels, W);
    HEAL_REQUIRE(x && weight_q && y && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
                 "grouped_small_conv3x3: bad pointer (x, y must be 16-B aligned)");
    // tile height: 16 rows (4 channels per group), 8 rows (8 per group: 72 weight registers; stride 2); HEAL_GS_TH overrides
    int th = (group_channels == 4 && stride == 1) ? 16 : 8;
    if (const char* e = getenv("HEAL_GS_TH")) th = atoi(e) == 8 ? 8 : atoi(e) == 16 ? 16 : th;
    if (stride == 2 || group_channels == 16) th = 8;   // (a 16-row stride-2 patch is 72 KB)
    const int tw = stride == 1 ? 32 : 16;
    const int tiles_x = ceil_div(Wo, tw), tiles_y = ceil_div(Ho, th);
    HEAL_REQUIRE(channels / 16 <= 65535 && n <= 65535, "grouped_small_conv3x3: grid limit");
    const dim3 grid(tiles_x * tiles_y, channels / 16, n);
    hipStream_t s_ = (hipStream_t)stream;
#define HEAL_GS(CG_, TH_, ST_)                                                                                        \
    if (group_channels == CG_ && th == TH_ && stride == ST_)                                                          \
        HEAL_LAUNCH_EV((k_gconv_small<CG_, TH_, ST_>), grid, dim3(256), 0, s_, x, weight_q, bias, channels, H, W, Ho, Wo, tiles_x, relu, y);
    HEAL_GS(4, 16, 1) HEAL_GS(4, 8, 1) HEAL_GS(8, 16, 1) HEAL_GS(8, 8, 1)
    HEAL_GS(4, 8, 2) HEAL_GS(8, 8, 2) HEAL_GS(16, 8, 1) HEAL_GS(16, 8, 2)
#undef HEAL_GS
    HEAL_LAUNCH_CHECK();
    return 0;
}
