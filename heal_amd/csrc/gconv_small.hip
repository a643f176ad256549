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
//     -> lane l computes pixel p = (l / 16) * 4 + l % 4 for the output channels hq * 4 + {0..3}, hq = (l / 4) % 4, and holds the
//        weight of output channel l % 16 as its A operand (36 or 72 registers: every (tap, input channel) of its group);
//     -> its B operand is input channel ci of ITS group at pixel p (+ tap offset): the patch is staged in LDS channel-interleaved,
//        [row][pixel][16 channels], so one ds_read_b128 returns the B operands of four k-steps, and the 64 lanes of a read
//        cover 16 pixels x 64 B = 1 KiB contiguous (conflict-free for 4 channels per group; 2-way for 8, where the two halves of a
//        group read the same 16 bytes).  A patch row read serves the three output rows it touches.
// One block = one super-group x a 16 x 32 output tile; wave w owns rows 4w .. 4w+3 (8 accumulators).  The patch (18 x 34 pixels
// x 16 channels, 38 KiB) is staged with 16-B global loads along the rows (4 channels x 4 pixels per thread, transposed in
// registers into four 16-B LDS stores), unconditional and clamped with the zero padding applied at the store.
// Bytes: every input element is read once per tile (+ halo 1.2x, from L2), every output written once: HBM-bound at these
// widths (0.33 GB per call at 128 channels x 256^2 x 5 agents).
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_gconv_small(const float* __restrict__ x, const float* __restrict__ wq,
                                                    const float* __restrict__ bias, int C, int H, int W, int tiles_x,
                                                    int relu, float* __restrict__ y) {
    static_assert(CG == 4 || CG == 8, "4 or 8 channels per group");
    constexpr int TH = 16, TW = 32, PR = TH + 2, PC = TW + 2;
    constexpr int NSTEP = 9 * CG;            // (tap, ci) k-steps of one group
    constexpr int NB = CG / 4;               // float4 B reads per (row, segment, dx)
    __shared__ float4 sP[PR * PC * 4];       // [row][pixel][unit = channel / 4]
    const Block3 bk = xcd_block();           // x: tile, y: super-group, z: image
    const int ty = bk.x / tiles_x, tx = bk.x - ty * tiles_x;
    const int sg = bk.y, n = bk.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int t = threadIdx.x, wave = t >> 6, l = t & 63;
    const size_t HW = (size_t)H * W;
    const float* __restrict__ xin = x + ((size_t)n * C + (size_t)sg * 16) * HW;

    // ---- stage the patch: interior quads (16-B loads, 4 channels x 4 pixels per item) + the two halo columns ---------------
    constexpr int N_IN = PR * 8 * 4, IT_IN = (N_IN + 255) / 256;   // 576 items -> 3 per thread
    constexpr int N_HA = PR * 2 * 4;                               // 144 halo items -> threads 0..143
    float4 vin[IT_IN][4];
    unsigned ok_in = 0;
#pragma unroll
    for (int it = 0; it < IT_IN; ++it) {
        const int u = min(t + 256 * it, N_IN - 1);                 // surplus threads repeat the last item (same data)
        const int gq = u & 3, quad = (u >> 2) & 7, row = u >> 5;
        const int gy = oy0 - 1 + row, gx = ox0 + quad * 4;
        const bool ok = gy >= 0 && gy < H && gx < W;               // W % 4 == 0: a quad is all-in or all-out
        ok_in |= ok ? (1u << it) : 0u;
        const size_t off = ok ? (size_t)gy * W + gx : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) vin[it][c] = *reinterpret_cast<const float4*>(xin + (size_t)(gq * 4 + c) * HW + off);
    }
    float4 vha;
    bool ok_ha;
    {
        const int u = min(t, N_HA - 1);
        const int gq = u & 3, side = (u >> 2) & 1, row = u >> 3;
        const int gy = oy0 - 1 + row, gx = side ? ox0 + TW : ox0 - 1;
        ok_ha = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t off = ok_ha ? (size_t)gy * W + gx : 0;
        const float* s = xin + (size_t)(gq * 4) * HW + off;
        vha = make_float4(s[0], s[HW], s[2 * HW], s[3 * HW]);
    }
    // A operands: the weights of output channel l % 16, every (tap, ci) of its group (L2-resident, 64 B per wave-load)
    float wreg[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wreg[s] = wq[((size_t)sg * NSTEP + s) * 16 + (l & 15)];

#pragma unroll
    for (int it = 0; it < IT_IN; ++it) {
        const int u = min(t + 256 * it, N_IN - 1);
        const int gq = u & 3, quad = (u >> 2) & 7, row = u >> 5;
        const bool ok = (ok_in >> it) & 1u;
        float4* d = sP + (row * PC + 1 + quad * 4) * 4 + gq;
        const float4 a = vin[it][0], b = vin[it][1], c = vin[it][2], e = vin[it][3];
        d[0] = ok ? make_float4(a.x, b.x, c.x, e.x) : make_float4(0.f, 0.f, 0.f, 0.f);
        d[4] = ok ? make_float4(a.y, b.y, c.y, e.y) : make_float4(0.f, 0.f, 0.f, 0.f);
        d[8] = ok ? make_float4(a.z, b.z, c.z, e.z) : make_float4(0.f, 0.f, 0.f, 0.f);
        d[12] = ok ? make_float4(a.w, b.w, c.w, e.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const int u = min(t, N_HA - 1);
        const int gq = u & 3, side = (u >> 2) & 1, row = u >> 3;
        sP[(row * PC + (side ? PC - 1 : 0)) * 4 + gq] = ok_ha ? vha : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // ---- 4 output rows x 2 segments per wave; a patch-row read feeds the (up to) three output rows it touches ------------
    f32x4 acc[4][2];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[o][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int p = (l >> 4) * 4 + (l & 3);                 // pixel within a 16-pixel segment
    const int unit0 = (((l >> 2) & 3) * 4 / CG) * NB;     // first 16-B unit of the lane's group within a pixel
    const float4* __restrict__ bbase = sP + ((wave * 4) * PC + p) * 4 + unit0;
#pragma unroll
    for (int ry = 0; ry < 6; ++ry) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float4 bv[2][NB];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int h = 0; h < NB; ++h) bv[s][h] = bbase[(ry * PC + s * 16 + dx) * 4 + h];
#pragma unroll
            for (int ci = 0; ci < CG; ++ci) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int dy = ry - o;
                    if (dy < 0 || dy > 2) continue;       // compile-time after unrolling
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float4 q = bv[s][ci >> 2];
                        const float b = (ci & 3) == 0 ? q.x : (ci & 3) == 1 ? q.y : (ci & 3) == 2 ? q.z : q.w;
                        acc[o][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[(dy * 3 + dx) * CG + ci], b, acc[o][s], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: D[l][r] = output channel ((l / 4) % 4) * 4 + r of the super-group at pixel p --------------------------
    float* __restrict__ yout = y + ((size_t)n * C + (size_t)sg * 16) * HW;
    const int cbase = ((l >> 2) & 3) * 4;
    float bvv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bvv[r] = bias ? bias[sg * 16 + cbase + r] : 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int oy = oy0 + wave * 4 + o;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ox = ox0 + s * 16 + p;
            if (oy >= H || ox >= W) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[o][s][r] + bvv[r];
                if (relu) v = fmaxf(v, 0.f);
                yout[(size_t)(cbase + r) * HW + (size_t)oy * W + ox] = v;
            }
        }
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_grouped_small_conv3x3(const float* x, const float* weight_q, const float* bias, int n, int channels,
                                          int group_channels, int H, int W, int relu, float* y, void* stream) {
    HEAL_REQUIRE(group_channels == 4 || group_channels == 8, "grouped_small_conv3x3: 4 or 8 channels per group (got %d)",
                 group_channels);
    HEAL_REQUIRE(n >= 1 && channels >= 16 && channels % 16 == 0 && H >= 1 && W >= 4 && W % 4 == 0,
                 "grouped_small_conv3x3: needs channels %% 16 == 0 and W %% 4 == 0 (got C=%d W=%d)", channels, W);
    HEAL_REQUIRE(x && weight_q && y && ((uintptr_t)x & 15) == 0, "grouped_small_conv3x3: bad pointer (x must be 16-B aligned)");
    const int tiles_x = ceil_div(W, 32), tiles_y = ceil_div(H, 16);
    HEAL_REQUIRE(channels / 16 <= 65535 && n <= 65535, "grouped_small_conv3x3: grid limit");
    const dim3 grid(tiles_x * tiles_y, channels / 16, n);
    if (group_channels == 4)
        k_gconv_small<4><<<grid, 256, 0, (hipStream_t)stream>>>(x, weight_q, bias, channels, H, W, tiles_x, relu, y);
    else
        k_gconv_small<8><<<grid, 256, 0, (hipStream_t)stream>>>(x, weight_q, bias, channels, H, W, tiles_x, relu, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}
