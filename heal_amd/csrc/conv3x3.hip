// Dense 3x3 convolution (padding 1, stride 1 | 2) on the fp32 matrix cores with the epilogue fused:
//     y = act(W * x + b (+ residual)),  NCHW fp32 in and out.
// Replaces the library (MIOpen Winograd / implicit-GEMM + layout transposes) convolutions and the separate bias / residual /
// ReLU pass of the dense BEV stacks:
//   opencood/models/sub_modules/downsample_conv.py:7-27   DoubleConv (shrink header: 384->256, 256->256 @ 256x256)
//   opencood/models/sub_modules/resblock.py:18-64          BasicBlock conv1 (stride 1 | 2) / conv2 + BN + (identity) + ReLU
//   opencood/models/sub_modules/base_bev_backbone.py:6-124 plain Conv-BN-ReLU stacks
//   opencood/models/sub_modules/lss_submodule.py:17-36     Up (432->512, 552->512, 512->512 at the image feature resolution)
//   torchvision Bottleneck.conv2 of the ResNet101 stem (lss_submodule.py:153-161)
//   opencood/models/sub_modules/naive_compress.py:5-31
//
// Implicit GEMM per image:  Y[co, p] = sum_{tap, ci} W[co][ci][tap] * X[ci][p + off(tap)]  on v_mfma_f32_16x16x4_f32
// (fp32 in, fp32 accumulate: bitwise an fmaf chain, no reduced precision anywhere).
//   * block = 4 waves = 64 output channels x (TH x 16) output pixels; wave w owns all 64 channels (4 m-tiles) of the
//     tile's rows [w*TH/4, (w+1)*TH/4) (NT = TH/4 n-tiles of 16 pixels): 16 | 8 independent accumulators per wave;
//   * K loop over chunks of KC = 8 input channels.  A chunk's input patch (tile + halo, zero padding resolved at staging
//     time) and its 9 x 8 x 64 weights go through LDS: one staging pass feeds 9 taps x 2 k-steps x MT x NT MFMAs per wave
//     (288 | 144), i.e. every staged activation is used 9 x 64 times -- the ratio that makes a 3x3 convolution easier
//     to keep on the matrix cores than the 1x1 (heal_conv1x1: every staged activation is used 64 times);
//   * B fragment (activations): lane (k = l >> 4, pixel = l & 15) reads patch[k][row + dy][col*S + dx]; the per-channel
//     stride of the patch is padded to 16 mod 32 words (stride 1) or to an odd number of words (stride 2) so that the two
//     k-rows a 32-lane LDS phase touches fall in disjoint banks -> conflict-free ds_read_b32;
//   * A fragment (weights): pre-laid by the host in fragment order [Cout/64][Cin/8][tap][k-step][m-tile][lane]
//     (ops.conv3x3_fragments), so a chunk's weights are ONE contiguous 18-KB run in memory (coalesced 16-B loads, L2
//     resident) and a fragment is 64 consecutive LDS words;
//   * the next chunk's global loads are issued before the current chunk's MFMAs (register staging), the LDS is single
//     buffered (28.8 KB | 36 KB per block: 4-5 blocks per CU hide the two barriers per chunk);
//   * epilogue in registers: + bias[co] (+ residual) -> ReLU -> 64-B row segments; XCD-contiguous tile order.
// Roofline: fp32 MFMA (157.3 TFLOP/s); HBM traffic is input (x Cout/64 re-reads, mostly L2 hits) + output.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int C3_KC = 8;                 // input channels per chunk
constexpr int C3_KS = C3_KC / 4;         // k-steps per chunk
constexpr int C3_WCHUNK = 9 * C3_KS * 4 * 64;  // floats of one chunk's weights (fragment order)

template <int STRIDE, int TH>
struct C3Geom {
    static constexpr int TW = 16;
    static constexpr int NT = TH / 4;                        // n-tiles (rows of 16 pixels) per wave
    static constexpr int PH = (TH - 1) * STRIDE + 3;         // patch rows
    static constexpr int PW = (TW - 1) * STRIDE + 3;         // patch columns
    static constexpr int PE = PH * PW;                       // elements per channel
    // per-channel LDS stride: == 16 (mod 32) for stride 1, odd for stride 2 (see header)
    static constexpr int CS = STRIDE == 1 ? ((PE + 15) / 32 * 32 + 16) : (PE | 1);
    static constexpr int NP = (C3_KC * PE + 255) / 256;      // patch elements staged per thread
    static constexpr int NW = (C3_WCHUNK / 4 + 255) / 256;   // weight float4 staged per thread
};

template <int STRIDE, int TH>
__global__ __launch_bounds__(256) void k_conv3x3(const float* __restrict__ x, const float* __restrict__ wfrag,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                int Cin, int nchunks, int Cout, int H, int W, int Ho, int Wo,
                                                int tiles_x, int tiles_y, int relu, int pad_t, int pad_l,
                                                float* __restrict__ y) {
    // relu: 0 none | 1 ReLU | 2 SiLU.  pad_t / pad_l: zero rows / columns in front of the map (1 = the symmetric padding of
    // the BEV stacks; 0 = TensorFlow-style "same" padding of a stride-2 convolution on an even map, which pads only behind:
    // the EfficientNet stem); what lies behind the map is zero through the bounds test either way.
    using G = C3Geom<STRIDE, TH>;
    constexpr int NT = G::NT, PH = G::PH, PW = G::PW, PE = G::PE, CS = G::CS, NP = G::NP, NW = G::NW;
    static_assert(CS >= PE, "bad patch stride");
    __shared__ __attribute__((aligned(16))) float sW[C3_WCHUNK];
    __shared__ float sP[C3_KC * CS];

    const Block3 bk = xcd_block();       // x: Cout block (fastest: blocks sharing a patch are neighbours), y: tile, z: image
    const int mb = bk.x, n = bk.z;
    const int ty = bk.y / tiles_x, tx = bk.y - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * G::TW;
    const int iy0 = oy0 * STRIDE - pad_t, ix0 = ox0 * STRIDE - pad_l;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int lk = l >> 4, ln = l & 15;
    const size_t HWin = (size_t)H * W;
    const float* __restrict__ xin = x + (size_t)n * Cin * HWin;
    const float4* __restrict__ wsrc = reinterpret_cast<const float4*>(wfrag + (size_t)mb * nchunks * C3_WCHUNK);

    // staging plan of this thread (the same patch positions for every chunk): element e = tid + 256 j of the [KC][PH][PW] patch
    int p_off[NP];       // offset inside the image plane, or -1 (outside the image / beyond the patch)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int ci = e / PE, rem = e - ci * PE, py = rem / PW, px = rem - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool ok = e < C3_KC * PE && gy >= 0 && gy < H && gx >= 0 && gx < W;
        p_off[j] = ok ? gy * W + gx : -1;
    }

    float pst[NP];
    float4 wst[NW];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int ch = c * C3_KC + (threadIdx.x + 256 * j) / PE;
            pst[j] = (p_off[j] >= 0 && ch < Cin) ? xin[(size_t)ch * HWin + p_off[j]] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = threadIdx.x + 256 * j;
            wst[j] = i < C3_WCHUNK / 4 ? wsrc[(size_t)c * (C3_WCHUNK / 4) + i] : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int e = threadIdx.x + 256 * j, ci = e / PE;
            if (e < C3_KC * PE) sP[ci * CS + (e - ci * PE)] = pst[j];
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < C3_WCHUNK / 4) reinterpret_cast<float4*>(sW)[i] = wst[j];
        }
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // B-fragment base of this lane: patch[lk][(wave*NT + nt)*S + dy][ln*S + dx]
    const float* __restrict__ bbase = sP + lk * CS + (wave * NT * STRIDE) * PW + ln * STRIDE;

    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);   // global loads in flight under this chunk's MFMAs
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int ks = 0; ks < C3_KS; ++ks) {
                float a[4], b[NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = sW[((tap * C3_KS + ks) * 4 + mt) * 64 + l];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = bbase[ks * 4 * CS + (nt * STRIDE + dy) * PW + dx];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
            __syncthreads();      // every wave is done reading this chunk
            store_chunk();
            __syncthreads();
        }
    }

    // Epilogue.  MFMA leaves D[row = lk*4 + r][col = ln]: lane (lk, ln) holds 4 consecutive output channels of pixel
    // (row = wave*NT + nt, col = ln) per (mt, nt); a store instruction writes 4 channel planes x 64 B.
    const size_t HWo = (size_t)Ho * Wo;
    float* __restrict__ yout = y + (size_t)n * Cout * HWo;
    const float* __restrict__ rin = res ? res + (size_t)n * Cout * HWo : nullptr;
    const int ox = ox0 + ln;
    // residual values first, all of them in flight together (unconditional loads on clamped addresses): a load inside the
    // predicated store loop sits in its own basic block and costs one exposed round trip each
    float rr[4][NT][4];
    if (rin) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const size_t pix = (size_t)min(oy0 + wave * NT + nt, Ho - 1) * Wo + min(ox, Wo - 1);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rr[mt][nt][r] = rin[(size_t)min(mb * 64 + mt * 16 + lk * 4 + r, Cout - 1) * HWo + pix];
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + wave * NT + nt;
        if (oy >= Ho || ox >= Wo) continue;
        const size_t pix = (size_t)oy * Wo + ox;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mb * 64 + mt * 16 + lk * 4 + r;
                if (co >= Cout) continue;
                float v = acc[mt][nt][r] + (bias ? bias[co] : 0.f);
                const size_t o = (size_t)co * HWo + pix;
                if (rin) v += rr[mt][nt][r];
                if (relu == 1) v = fmaxf(v, 0.f);
                else if (relu == 2) v = v / (1.f + expf(-v));
                yout[o] = v;
            }
        }
    }
}


// ---- Winograd F(2x2, 3x3) on the matrix cores (stride 1) ----------------------------------------------------------------
// The direct kernel above sustains 116-124 TFLOP/s on the shrink header -- level with the library's Winograd kernel, which
// does 2.25x fewer multiplications on the vector ALUs.  The same minimal-filtering transform on the MATRIX cores:
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A,   d: 4x4 input window (stride 2), g: 3x3 filter, Y: 2x2 outputs
// turns the convolution into 16 independent GEMMs (one per position xi of the 4x4 transform domain)
//     M[xi][co, tile] = sum_ci U[xi][co, ci] V[xi][ci, tile]
// with 16/36 of the direct MFMA count.  Fused, nothing of the transform domain touches HBM:
//   * block = 8 waves = 64 output channels x 16x16 output pixels (64 tiles of 2x2); wave w owns the transform positions
//     xi = 2w, 2w+1: two 64x64 GEMM accumulators = 128 registers per lane;
//   * per chunk of 8 input channels: the 18x18 input patch goes global -> registers (prefetched one chunk ahead) -> LDS;
//     thread (ci, tile) forms V = B^T d B (32 additions) and writes its 16 values to sV[xi][ci][tile]; each wave then issues
//     2 xi x 2 k-steps x 16 MFMAs with B fragments from sV (row stride 80 words: conflict-free) and A fragments (U) read
//     straight from L2 in a lane-major pre-laid order (four 16-B loads per lane per chunk, prefetched one chunk ahead);
//   * epilogue: four passes of 16 output channels through LDS (sM[xi][co][tile], aliasing the staging buffers), thread
//     (co, tile) gathers its 16 transform-domain sums, applies A^T . A, bias (+ residual) (+ ReLU) and stores 2x2 pixels.
// Arithmetic: fp32 throughout; the transform changes the rounding sequence (like the library's F(2,3) kernel): ~1e-6
// relative to the direct evaluation, inside the 1e-3 feature tolerance and tested at 1e-4.
constexpr int WG_KC = 8;                         // input channels per chunk (default); KC = 16: round 5, half the barriers per MFMA
constexpr int WG_PROW = 24;                      // patch row stride (words): 2*24 = 48 -> the 4 tile rows of a 32-lane ds_read_b64 phase fall in disjoint bank ranges
constexpr int WG_PDUMMY = 512;                   // landing zone of the staging slots beyond the patch (keeps the stores unconditional)

// NW = waves per block: 8 -> 16x16 output pixels (64 tiles), two transform positions per wave, one block per CU;
//                       4 -> 8x16 output pixels (32 tiles), four transform positions per wave, TWO independent blocks per CU
//                            (same 8 waves per CU), so one block's transform phase overlaps the other's MFMA phase.
template <int NW, int KC = WG_KC>
struct WgGeom {
    static constexpr int TR = 2 * NW;                    // output rows of the block's tile
    static constexpr int NTL = 8 * NW;                   // 2x2 tiles per block (TR/2 rows x 8)
    static constexpr int XW = 16 / NW;                   // transform positions per wave
    static constexpr int NTN = NTL / 16;                 // MFMA n-tiles per transform position
    static constexpr int PR = TR + 2;                    // patch rows
    static constexpr int PE = PR * 18;                   // patch elements per channel
    static constexpr int PCI = PR * WG_PROW;             // LDS words per patch channel
    static constexpr int VROW = NTL + 16;                // sV row stride: k-rows lk, lk+1 of a B fragment hit disjoint banks
    static constexpr int MROW = NTL + 4;                 // sM row stride: 4*MROW = 16 (mod 32) -> conflict-free accumulator dump
    static constexpr int STAGE = KC * PCI + WG_PDUMMY + 16 * KC * VROW;
    static constexpr int SMEM = STAGE > 16 * 16 * MROW ? STAGE : 16 * 16 * MROW;   // staging buffers alias the epilogue buffer
    static constexpr int NP = (KC * PE + 64 * NW - 1) / (64 * NW);                 // patch elements staged per thread
    static constexpr int KS = KC / 4;                    // MFMA k-steps per chunk
    static constexpr int UQ = XW * KS;                   // float4 of U per lane per chunk (XW xi x KS k-steps x 4 m-tiles floats)
};

// EXACT (Cin % 8 == 0: every shape of the two BASELINE scenes): the addresses of the loop are a UNIFORM base that advances by one
// chunk plus per-thread byte offsets fixed before the loop -- `global_load v, v_off, s[base]` with no vector ALU work per load.
// Round 3 recomputed, per chunk and element, the clamped channel, a 64-bit `channel * HW + offset` and the in-range predicate of
// the store: ~60 of the 150 non-MFMA instructions of an iteration, all in the transform phase the matrix pipe waits for.
// The zero padding is written ONCE (slots outside the image keep their zero: their loads land in the dummy zone).
template <int NW, bool EXACT, int KC = WG_KC>
__global__ __launch_bounds__(64 * NW) void k_conv3x3_wino(const float* __restrict__ x, const float4* __restrict__ ufrag,
                                                         const float* __restrict__ bias, const float* __restrict__ res,
                                                         int Cin, int nchunks, int Cout, int H, int W, int tiles_x, int relu,
                                                         float* __restrict__ y, int n_img, int ksplit) {
    using G = WgGeom<NW, KC>;
    constexpr int NT = 64 * NW, TR = G::TR, NTL = G::NTL, XW = G::XW, NTN = G::NTN, PE = G::PE, PCI = G::PCI;
    constexpr int VROW = G::VROW, MROW = G::MROW, NP = G::NP, UQ = G::UQ, KS = G::KS;
    constexpr int TCH = KC / (NT / NTL);          // channels a thread transforms per chunk (NT / NTL = 8 per sweep)
    static_assert(KC == 8 || KC == 16, "chunk of 8 or 16 input channels");
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
    float* sP = smem;
    float* sV = smem + KC * PCI + WG_PDUMMY;
    float* sM = smem;

    // Block order: tile fastest, image next, Cout block SLOWEST -- the Winograd weights are the big stream (16 x Cin x 64
    // floats per Cout block: 1.6 MB at Cin = 384, re-read by every tile) and have to stay in the XCDs' 4 MB L2s, so the
    // blocks in flight at any time share one Cout block; the input map is then read once per Cout block.
    // ksplit > 1 (small maps with a deep reduction: 192 blocks of 64 chunks each for the 512 -> 512 layers of the camera trunk at
    // 4 x 24 x 32 pixels): grid.y = ksplit * n_img, block y reduces chunk range kpart of image n and writes its PARTIAL output --
    // the output transform is linear, so partial sums may leave the transform domain -- to y[kpart][n][Cout][HW] without bias /
    // residual / ReLU; k_conv1x1_splitk_reduce adds the partials in split order (deterministic) and applies them.
    const Block3 bk = xcd_block();
    const int mb = bk.z, n = (int)bk.y % n_img, kpart = (int)bk.y / n_img;
    const int cps = (nchunks + ksplit - 1) / ksplit, cbeg = kpart * cps, cend = min(cbeg + cps, nchunks);
    const int tyb = bk.x / tiles_x, txb = bk.x - tyb * tiles_x;
    const int oy0 = tyb * TR, ox0 = txb * 16;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int lk = l >> 4, ln = l & 15;
    const size_t HW = (size_t)H * W;
    const float* __restrict__ xin = x + (size_t)n * Cin * HW;

    // patch staging plan: element e = tid + NT j of the [8][PR][18] patch (same positions for every chunk).  Loads are
    // UNCONDITIONAL (clamped in-bounds address; the zero padding / channel tail is applied when the value goes to LDS):
    // predicated loads put every load in its own basic block, and the compiler's waitcnt insertion then falls back to
    // `s_waitcnt vmcnt(0)` at each of them -- which drains the U loads issued just before and serialises one L2 round trip per
    // chunk (visible in the ISA of the first version of this loop; scripts/wg_dbg.py: 125 + 76 us of 654 us).
    int p_off[NP];      // offset inside the image plane (clamped to 0 when outside); EXACT: BYTE offset from the chunk's base
    int p_lds[NP];      // LDS word (slots beyond the patch land in the dummy zone: no branch around the store -- a conditional
                        // store lets the compiler sink the LOAD into the branch, right in front of its wait)
    unsigned p_ok = 0;  // bit j: element j lies inside the image and the patch
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int e = threadIdx.x + NT * j;
        const int ci = e / PE, rem = e - ci * PE, py = rem / 18, px = rem - py * 18;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        const bool ok = e < KC * PE && gy >= 0 && gy < H && gx >= 0 && gx < W;
        p_off[j] = ok ? gy * W + gx : 0;
        p_ok |= ok ? (1u << j) : 0u;
        p_lds[j] = e < KC * PE ? ci * PCI + py * WG_PROW + px : KC * PCI + (e - KC * PE) % WG_PDUMMY;
        if constexpr (EXACT) {
            if (!ok && e < KC * PE) {      // padding: zero once, then send this slot's (clamped) loads to the dummy zone
                sP[p_lds[j]] = 0.f;
                p_lds[j] = KC * PCI + e % WG_PDUMMY;
            }
            p_off[j] = (int)(((size_t)min(ci, KC - 1) * HW + (size_t)p_off[j]) * sizeof(float));   // < 8 HW 4 B: fits (host check)
        }
    }
    float pst[NP];
    const size_t chunk_bytes = (size_t)KC * HW * sizeof(float);
    auto load_patch = [&](int c) {
        if constexpr (EXACT) {
            const char* __restrict__ cb = reinterpret_cast<const char*>(xin) + (size_t)c * chunk_bytes;   // uniform
#pragma unroll
            for (int j = 0; j < NP; ++j) pst[j] = *reinterpret_cast<const float*>(cb + (unsigned)p_off[j]);
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int ch = min(c * KC + (int)(threadIdx.x + NT * j) / PE, Cin - 1);
                pst[j] = xin[(size_t)ch * HW + p_off[j]];
            }
        }
    };
    auto store_patch = [&](int c) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if constexpr (EXACT) {
                sP[p_lds[j]] = pst[j];
            } else {
                const int ci = (int)(threadIdx.x + NT * j) / PE;
                const bool ok = ((p_ok >> j) & 1u) && c * KC + ci < Cin;
                sP[p_lds[j]] = ok ? pst[j] : 0.f;
            }
        }
    };
    // U fragments of this wave: XW*8 floats per lane per chunk, lane-major: [mb][chunk][wave][lane][(xi_i*2 + ks)*4 + mt]
    // (uniform chunk base + a per-lane byte offset: no vector address arithmetic per load)
    const char* __restrict__ ublock = reinterpret_cast<const char*>(ufrag + (size_t)mb * nchunks * NW * 64 * UQ);
    const unsigned u_off = (unsigned)((wave * 64 + l) * UQ * sizeof(float4));
    float4 ua[UQ];
    auto load_u = [&](int c) {
        const char* __restrict__ ub = ublock + (size_t)c * (NW * 64 * UQ * sizeof(float4));
#pragma unroll
        for (int q = 0; q < UQ; ++q) ua[q] = *reinterpret_cast<const float4*>(ub + u_off + q * sizeof(float4));
    };

    f32x4 acc[XW][4][NTN];
#pragma unroll
    for (int a = 0; a < XW; ++a)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[a][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transform role of this thread: channel tci of the chunk, tile (tty, ttx) of the (TR/2) x 8 tile grid
    const int tci = threadIdx.x / NTL, ttile = threadIdx.x - tci * NTL, tty = ttile >> 3, ttx = ttile & 7;
    const float* __restrict__ dsrc = sP + tci * PCI + (2 * tty) * WG_PROW + 2 * ttx;
    float* __restrict__ vdst = sV + tci * VROW + ttile;

    // Software pipeline: the global loads of chunk c+1 (input patch -> registers) are issued at the TOP of iteration c and
    // consumed at its BOTTOM (-> LDS); the U fragments of chunk c+1 are requested right after the MFMAs of chunk c have
    // been issued and land during the next barrier + transform.  This requires barriers that do not drain VMEM
    // (lds_barrier, common.h) and a loop body that is ONE basic block (no predicated loads; the last iteration reloads
    // chunk nchunks-1 instead of branching).
    load_patch(cbeg);
    load_u(cbeg);
    store_patch(cbeg);
    for (int c = cbeg; c < cend; ++c) {
        const int cn = min(c + 1, cend - 1);
        lds_barrier();                            // patch(c) is in LDS; every wave is done with sV of chunk c-1
        load_patch(cn);
#pragma unroll
        for (int h = 0; h < TCH; ++h) {   // V = B^T d B for (tci + 8 h, ttile)
            const float* __restrict__ ds_ = dsrc + h * (NT / NTL) * PCI;
            float* __restrict__ vd_ = vdst + h * (NT / NTL) * VROW;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 lo = *reinterpret_cast<const float2*>(ds_ + i * WG_PROW);
                const float2 hi = *reinterpret_cast<const float2*>(ds_ + i * WG_PROW + 2);
                d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
            }
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j];
                t[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vd_[((4 * i + 0) * KC) * VROW] = t[i][0] - t[i][2];
                vd_[((4 * i + 1) * KC) * VROW] = t[i][1] + t[i][2];
                vd_[((4 * i + 2) * KC) * VROW] = t[i][2] - t[i][1];
                vd_[((4 * i + 3) * KC) * VROW] = t[i][1] - t[i][3];
            }
        }
        lds_barrier();                            // sV(c) complete; sP free (global loads stay in flight)
        {
            float a_[UQ * 4];
#pragma unroll
            for (int q = 0; q < UQ; ++q) { a_[4 * q] = ua[q].x; a_[4 * q + 1] = ua[q].y; a_[4 * q + 2] = ua[q].z; a_[4 * q + 3] = ua[q].w; }
#pragma unroll
            for (int xi_i = 0; xi_i < XW; ++xi_i) {
                // column ln of n-tile nt is TILE NTN * ln + nt (GEMM columns can be numbered freely): a lane's B operands of all its
                // n-tiles are consecutive words of sV -- one ds_read_b128 (b64 for the 4-wave block) instead of NTN b32 reads
                const float* __restrict__ vb = sV + ((XW * wave + xi_i) * KC + lk) * VROW + NTN * ln;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    float b[NTN];
                    if constexpr (NTN == 4) {
                        const float4 q4 = *reinterpret_cast<const float4*>(vb + ks * 4 * VROW);
                        b[0] = q4.x; b[1] = q4.y; b[2] = q4.z; b[3] = q4.w;
                    } else {
                        const float2 q2 = *reinterpret_cast<const float2*>(vb + ks * 4 * VROW);
                        b[0] = q2.x; b[NTN - 1] = q2.y;
                    }
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[xi_i][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[(xi_i * KS + ks) * 4 + mt], b[nt],
                                                                                     acc[xi_i][mt][nt], 0, 0, 0);
                }
            }
        }
        load_u(cn);                               // next chunk's U: lands during the barrier + transform that follow
        store_patch(cn);                          // sP is free since the second barrier (last iteration: rewrites the last chunk, unused)
    }

    // Epilogue: four passes of 16 output channels (m-tile p) through sM[xi][co][tile]
    const size_t HWo = HW;                        // stride 1, padding 1: same map size
    float* __restrict__ yout = y + ((size_t)kpart * n_img + n) * Cout * HWo;
    const float* __restrict__ rin = res ? res + (size_t)n * Cout * HWo : nullptr;
    constexpr int CPP = NT / NTL;                 // channels handled per sweep of the block (8)
    const int ecg = threadIdx.x / NTL, etile = threadIdx.x - ecg * NTL, ety = etile >> 3, etx = etile & 7;
    const int oy = oy0 + 2 * ety, ox = ox0 + 2 * etx;
    const bool even_w = (W & 1) == 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // the residual values of this pass are requested BEFORE the accumulators go through LDS (unconditional loads on clamped
        // addresses): they land under the dump, the barrier and the transform instead of costing a round trip per (channel, row)
        // right in front of the stores
        float rr0[16 / CPP][2], rr1[16 / CPP][2];
        if (rin) {
#pragma unroll
            for (int h = 0; h < 16 / CPP; ++h) {
                const int co = min(mb * 64 + p * 16 + ecg + CPP * h, Cout - 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int yy = min(oy + i, H - 1), xx = min(ox, W - (even_w ? 2 : 1));
                    const size_t o0 = (size_t)co * HWo + (size_t)yy * W + xx;
                    if (even_w) {
                        const float2 q = *reinterpret_cast<const float2*>(rin + o0);
                        rr0[h][i] = q.x; rr1[h][i] = q.y;
                    } else {
                        rr0[h][i] = rin[o0];
                        rr1[h][i] = rin[o0 + (xx + 1 < W ? 1 : 0)];
                    }
                }
            }
        }
        lds_barrier();                            // staging buffers (first pass) / previous pass no longer read
#pragma unroll
        for (int xi_i = 0; xi_i < XW; ++xi_i)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sM[((XW * wave + xi_i) * 16 + lk * 4 + r) * MROW + nt * 16 + ln] = acc[xi_i][p][nt][r];
        lds_barrier();
#pragma unroll
        for (int h = 0; h < 16 / CPP; ++h) {
            const int col = ecg + CPP * h;         // channel inside the 16-channel pass
            const int co = mb * 64 + p * 16 + col;
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = sM[(xi * 16 + col) * MROW + (etile % NTN) * 16 + etile / NTN];   // GEMM column of tile etile
            if (co >= Cout || oy >= H || ox >= W) continue;
            // Y = A^T M A,  A^T = [[1,1,1,0],[0,1,-1,-1]]
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = (m[j] + m[4 + j]) + m[8 + j];
                t1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
            }
            float o[2][2] = {{(t0[0] + t0[1]) + t0[2], (t0[1] - t0[2]) - t0[3]},
                             {(t1[0] + t1[1]) + t1[2], (t1[1] - t1[2]) - t1[3]}};
            const float bv = bias ? bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (oy + i >= H) continue;
                const size_t o0 = (size_t)co * HWo + (size_t)(oy + i) * W + ox;
                float v0 = o[i][0] + bv, v1 = o[i][1] + bv;
                const bool two = ox + 1 < W;
                if (rin) { v0 += rr0[h][i]; v1 += rr1[h][i]; }
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                if (two && even_w) *reinterpret_cast<float2*>(yout + o0) = make_float2(v0, v1);
                else { yout[o0] = v0; if (two) yout[o0 + 1] = v1; }
            }
        }
    }
}

// ---- 32-group 3x3 convolution of the ResNeXt bottlenecks on the matrix cores ------------------------------------------------
// heal_grouped_conv3x3 (bev_conv.hip) is a vector-ALU stencil: 21-38 TFLOP/s, 4.6x the HBM floor at 16 channels per group.
// With 16 channels per group a group IS a small dense convolution -- [16 co x 144] x [144 x pixels] -- exactly one MFMA
// m-tile wide; groups of 8 channels are paired into 16-channel super-groups with block-diagonal (zero-padded) weights (half
// of the executed MFMAs multiply zeros, still 2x faster than the stencil).  One block = one super-group x one 16x16-pixel
// tile: the 18x18 patch of its 16 input channels and its 9 x 16 x 16 weights (fragment order, pre-laid by the host) are staged
// once, wave w owns the tile rows 4w..4w+3 (4 accumulators), 9 taps x 4 k-steps x 4 MFMAs per wave, no K loop.
// Stride 1, padding 1.  Reference: opencood/models/sub_modules/resblock.py:90-98,110-112 (conv2 + bn2 + relu, groups = 32).
__global__ __launch_bounds__(256) void k_grouped16_conv3x3(const float* __restrict__ x, const float4* __restrict__ wfrag,
                                                          const float* __restrict__ bias, int C, int H, int W, int tiles_x,
                                                          int relu, float* __restrict__ y) {
    constexpr int PE = 324, CS = 336;       // 18 x 18 patch; channel stride == 16 (mod 32): conflict-free B fragments
    constexpr int NPT = (16 * PE + 255) / 256;   // 21 patch elements per thread
    constexpr int WF = 9 * 4 * 64;          // 2304 weight floats per super-group
    __shared__ __attribute__((aligned(16))) float sW[WF];
    __shared__ float sP[16 * CS + 256];     // + landing zone of the staging slots beyond the patch
    const Block3 bk = xcd_block();          // x: tile, y: super-group, z: image
    const int ty = bk.x / tiles_x, tx = bk.x - ty * tiles_x;
    const int sg = bk.y, n = bk.z;
    const int oy0 = ty * 16, ox0 = tx * 16;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, lk = l >> 4, ln = l & 15;
    const size_t HW = (size_t)H * W;
    const float* __restrict__ xin = x + ((size_t)n * C + (size_t)sg * 16) * HW;

    // stage: unconditional clamped loads (all in flight together), zero padding applied at the LDS store.  A patch row is
    // 16 interior pixels (64-B aligned: four 16-B loads, W % 4 == 0) plus one halo pixel on either side: 288 rows ->
    // 1152 float4 + 576 scalars = 7 loads per thread instead of 21 scalar ones.
    float4 ist[5];
    float hst[3];
    unsigned ok_i = 0, ok_h = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int u = threadIdx.x + 256 * j, row = min(u >> 2, 287), q = u & 3;
        const int ci = row / 18, py = row - ci * 18;
        const int gy = oy0 - 1 + py, gx = ox0 + 4 * q;
        const bool ok = u < 1152 && gy >= 0 && gy < H && gx < W;
        ok_i |= ok ? (1u << j) : 0u;
        ist[j] = *reinterpret_cast<const float4*>(xin + (size_t)ci * HW + (ok ? (size_t)gy * W + gx : 0));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int h = threadIdx.x + 256 * j, row = min(h >> 1, 287), side = h & 1;
        const int ci = row / 18, py = row - ci * 18;
        const int gy = oy0 - 1 + py, gx = side ? ox0 + 16 : ox0 - 1;
        const bool ok = h < 576 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        ok_h |= ok ? (1u << j) : 0u;
        hst[j] = xin[(size_t)ci * HW + (ok ? (size_t)gy * W + gx : 0)];
    }
    float4 wst[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = min((int)threadIdx.x + 256 * j, WF / 4 - 1);
        wst[j] = wfrag[(size_t)sg * (WF / 4) + i];
    }
    (void)NPT;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int u = threadIdx.x + 256 * j, row = u >> 2, q = u & 3;
        const int ci = row / 18, py = row - ci * 18;
        const bool ok = (ok_i >> j) & 1u;
        float* d = u < 1152 ? sP + ci * CS + py * 18 + 1 + 4 * q : sP + 16 * CS + (u - 1152) * 4 % 256;
        d[0] = ok ? ist[j].x : 0.f; d[1] = ok ? ist[j].y : 0.f; d[2] = ok ? ist[j].z : 0.f; d[3] = ok ? ist[j].w : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int h = threadIdx.x + 256 * j, row = h >> 1, side = h & 1;
        const int ci = row / 18, py = row - ci * 18;
        sP[h < 576 ? ci * CS + py * 18 + (side ? 17 : 0) : 16 * CS + (h - 576)] = ((ok_h >> j) & 1u) ? hst[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < WF / 4) reinterpret_cast<float4*>(sW)[i] = wst[j];
    }
    __syncthreads();

    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ bbase = sP + lk * CS + (wave * 4) * 18 + ln;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float a = sW[(tap * 4 + ks) * 64 + l];
            float b[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) b[nt] = bbase[ks * 4 * CS + (nt + dy) * 18 + dx];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[nt], 0, 0, 0);
        }
    }
    // D[row = lk*4 + r][col = ln]: lane holds 4 consecutive output channels of pixel (4*wave + nt, ln)
    float* __restrict__ yout = y + ((size_t)n * C + (size_t)sg * 16) * HW;
    const int ox = ox0 + ln;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int oy = oy0 + wave * 4 + nt;
        if (oy >= H || ox >= W) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = lk * 4 + r;
            float v = acc[nt][r] + (bias ? bias[sg * 16 + co] : 0.f);
            if (relu) v = fmaxf(v, 0.f);
            yout[(size_t)co * HW + (size_t)oy * W + ox] = v;
        }
    }
}

}  // namespace heal

using namespace heal;

static int conv3x3_launch(const float* x, const float* weight_frag, const float* bias, const float* residual, int n,
                          int cin, int cout, int H, int W, int stride, int pad_t, int pad_l, int Ho, int Wo, int relu,
                          float* y, void* stream) {
    HEAL_REQUIRE(n >= 1 && H >= 1 && W >= 1 && cin >= 1 && cout >= 1, "conv3x3: bad shape");
    HEAL_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride must be 1 or 2 (got %d)", stride);
    HEAL_REQUIRE(x && weight_frag && y, "conv3x3: null pointer");
    HEAL_REQUIRE(((uintptr_t)weight_frag & 15) == 0, "conv3x3: weight fragments must be 16-B aligned");
    const int nchunks = (cin + C3_KC - 1) / C3_KC, mblocks = (cout + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    // Tile height (measured, scripts/conv3x3_bench.py): the kernel is barrier-bound at 2 waves per SIMD, so more, smaller
    // blocks win -- 4-row tiles everywhere except for deep reductions (Cin >= 256: the 18-KB weight chunk per block-chunk
    // then dominates the staging traffic and taller tiles amortise it): 8 rows, 16 when that still leaves >= 1024 blocks.
    const long long per_row_tiles = (long long)n * mblocks * ((Wo + 15) / 16);
    int th = 4;
    if (cin >= 256) {
        th = 8;
        if (stride == 1 && per_row_tiles * ((Ho + 15) / 16) >= 1024) th = 16;
    }
    if (const char* e = getenv("HEAL_C3_TH")) { const int v = atoi(e); if ((v == 4 || v == 8 || (v == 16 && stride == 1))) th = v; }
    const int tiles_x = ceil_div(Wo, 16), tiles_y = ceil_div(Ho, th);
    HEAL_REQUIRE((long long)tiles_x * tiles_y <= 65535 && n <= 65535, "conv3x3: map too large for the launch grid");
    const dim3 grid(mblocks, tiles_x * tiles_y, n);
#define HEAL_C3(ST_, TH_)                                                                                          \
    HEAL_LAUNCH_EV((k_conv3x3<ST_, TH_>), grid, dim3(256), 0, s, x, weight_frag, bias, residual, cin, nchunks, cout, H, W, Ho, Wo, \
                   tiles_x, tiles_y, relu, pad_t, pad_l, y)
    if (stride == 1 && th == 16) HEAL_C3(1, 16);
    else if (stride == 1 && th == 8) HEAL_C3(1, 8);
    else if (stride == 1) HEAL_C3(1, 4);
    else if (th == 8) HEAL_C3(2, 8);
    else HEAL_C3(2, 4);
#undef HEAL_C3
    HEAL_LAUNCH_CHECK();
    return 0;
}


extern "C" int heal_conv3x3(const float* x, const float* weight_frag, const float* bias, const float* residual, int n,
                            int cin, int cout, int H, int W, int stride, int relu, float* y, void* stream) {
    HEAL_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride must be 1 or 2 (got %d)", stride);
    return conv3x3_launch(x, weight_frag, bias, residual, n, cin, cout, H, W, stride, 1, 1, (H + 2 - 3) / stride + 1,
                          (W + 2 - 3) / stride + 1, relu ? 1 : 0, y, stream);
}

extern "C" int heal_conv3x3_same(const float* x, const float* weight_frag, const float* bias, int n, int cin, int cout, int H,
                                 int W, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, float* y, void* stream) {
    HEAL_REQUIRE((pad_t == 0 || pad_t == 1) && (pad_l == 0 || pad_l == 1), "conv3x3_same: leading padding must be 0 or 1");
    HEAL_REQUIRE(act >= 0 && act <= 2, "conv3x3_same: act must be 0 (none), 1 (ReLU) or 2 (SiLU)");
    HEAL_REQUIRE(Ho >= 1 && Wo >= 1 && (Ho - 1) * stride + 3 - pad_t <= H + 1 && (Wo - 1) * stride + 3 - pad_l <= W + 1,
                 "conv3x3_same: output %dx%d needs more than one trailing padding row / column", Ho, Wo);
    return conv3x3_launch(x, weight_frag, bias, nullptr, n, cin, cout, H, W, stride, pad_t, pad_l, Ho, Wo, act, y, stream);
}

extern "C" int heal_conv3x3_winograd(const float* x, const float* u_frag, const float* bias, const float* residual, int n,
                                     int cin, int cout, int H, int W, int relu, int waves, float* y, void* stream) {
    return heal_conv3x3_winograd_kc(x, u_frag, bias, residual, n, cin, cout, H, W, relu, waves, WG_KC, y, stream);
}

namespace heal {
// conv1x1.hip: out = act(sum_ks part[ks] + bias (+ residual)), the partials [ksplit][n][Cout][HW] added in split order
int splitk_reduce_launch(const float* partials, const float* bias, const float* residual, int ksplit, int n, int cout, int HW,
                         int act, float* y, hipStream_t s, hipEvent_t ev_stop);
}

static int winograd_launch(const float* x, const float* u_frag, const float* bias, const float* residual, int n, int cin, int cout,
                           int H, int W, int relu, int waves, int kc, int ksplit, float* partials, float* y, void* stream) {
    HEAL_REQUIRE(n >= 1 && H >= 1 && W >= 1 && cin >= 1 && cout >= 1, "conv3x3_winograd: bad shape");
    HEAL_REQUIRE(x && u_frag && y, "conv3x3_winograd: null pointer");
    HEAL_REQUIRE(((uintptr_t)u_frag & 15) == 0, "conv3x3_winograd: weight fragments must be 16-B aligned");
    HEAL_REQUIRE(kc == 8 || (kc == 16 && waves == 8 && cin % 16 == 0),
                 "conv3x3_winograd: chunks of 16 input channels need 8 waves per block and cin %% 16 == 0 (kc=%d waves=%d cin=%d)", kc, waves, cin);
    const int nchunks = (cin + kc - 1) / kc, mblocks = (cout + 63) / 64;
    HEAL_REQUIRE(waves == 8 || waves == 4, "conv3x3_winograd: waves per block must be 8 (16x16-pixel tiles) or 4 (8x16)");
    const int tiles_x = ceil_div(W, 16), tiles_y = ceil_div(H, 2 * waves);
    HEAL_REQUIRE((long long)tiles_x * tiles_y <= 2147483647ll / 4 && (long long)n * ksplit <= 65535 && mblocks <= 65535,
                 "conv3x3_winograd: map too large for the launch grid");
    const dim3 grid(tiles_x * tiles_y, n * ksplit, mblocks);
    const float4* uf = reinterpret_cast<const float4*>(u_frag);
    // uniform-base addressing (see the kernel): whole chunks and byte offsets inside a chunk that fit 32 bits
    const bool exact = cin % kc == 0 && (long long)kc * H * W * 4 < 2147483647ll;
    // split K: the blocks write partial outputs, the reduce launch applies bias / residual / ReLU
    const float* kb = ksplit > 1 ? nullptr : bias;
    const float* kr = ksplit > 1 ? nullptr : residual;
    const int krelu = ksplit > 1 ? 0 : relu;
    float* ky = ksplit > 1 ? partials : y;
    LaunchEvents ev = take_launch_events();
    hipEvent_t ev_mid = ksplit > 1 ? (hipEvent_t) nullptr : ev.stop;
#define HEAL_WINO_LAUNCH(NW_, EX_, KC_)                                                                                              \
    HEAL_LAUNCH_EV2((k_conv3x3_wino<NW_, EX_, KC_>), grid, dim3(64 * NW_), 0, (hipStream_t)stream, ev.start, ev_mid, x, uf, kb, kr, cin, \
                    nchunks, cout, H, W, tiles_x, krelu, ky, n, ksplit)
    if (kc == 16) {
#ifdef HEAL_BUILD_EXPERIMENTAL
        HEAL_REQUIRE(exact, "conv3x3_winograd: kc = 16 needs a map of less than 2^31 / 64 bytes per channel");
        HEAL_WINO_LAUNCH(8, true, 16);     // (111.6 KB of STATIC LDS: no dynamic-LDS attribute to raise)
#else
        HEAL_REQUIRE(false, "conv3x3_winograd: kc = 16 is a measured-negative variant, built only with HEAL_BUILD_EXPERIMENTAL=1");
#endif
    } else if (waves == 8) { if (exact) HEAL_WINO_LAUNCH(8, true, 8); else HEAL_WINO_LAUNCH(8, false, 8); }
    else { if (exact) HEAL_WINO_LAUNCH(4, true, 8); else HEAL_WINO_LAUNCH(4, false, 8); }
#undef HEAL_WINO_LAUNCH
    HEAL_LAUNCH_CHECK();
    if (ksplit > 1) return splitk_reduce_launch(partials, bias, residual, ksplit, n, cout, H * W, relu ? 1 : 0, y, (hipStream_t)stream, ev.stop);
    return 0;
}

extern "C" int heal_conv3x3_winograd_kc(const float* x, const float* u_frag, const float* bias, const float* residual, int n,
                                        int cin, int cout, int H, int W, int relu, int waves, int kc, float* y, void* stream) {
    return winograd_launch(x, u_frag, bias, residual, n, cin, cout, H, W, relu, waves, kc, 1, nullptr, y, stream);
}

extern "C" size_t heal_conv3x3_winograd_splitk_workspace(int n, int cout, int H, int W, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * n * cout * H * W * sizeof(float) : 0;
}

extern "C" int heal_conv3x3_winograd_splitk(const float* x, const float* u_frag, const float* bias, const float* residual, int n,
                                            int cin, int cout, int H, int W, int relu, int waves, int ksplit, float* y, void* ws,
                                            size_t ws_bytes, void* stream) {
    const int nchunks = (cin + WG_KC - 1) / WG_KC;
    HEAL_REQUIRE(ksplit >= 2 && ksplit <= nchunks, "conv3x3_winograd_splitk: ksplit must be in [2, %d] (got %d)", nchunks, ksplit);
    HEAL_REQUIRE((ksplit - 1) * ceil_div(nchunks, ksplit) < nchunks,
                 "conv3x3_winograd_splitk: %d splits of %d chunks leave an empty split (use ceil(chunks / ceil(chunks / ksplit)))",
                 ksplit, nchunks);
    HEAL_REQUIRE((H * W) % 4 == 0, "conv3x3_winograd_splitk: H*W must be a multiple of 4");
    HEAL_REQUIRE(ws && ws_bytes >= heal_conv3x3_winograd_splitk_workspace(n, cout, H, W, ksplit) && ((uintptr_t)ws & 15) == 0,
                 "conv3x3_winograd_splitk: workspace too small or misaligned");
    return winograd_launch(x, u_frag, bias, residual, n, cin, cout, H, W, relu, waves, WG_KC, ksplit, (float*)ws, y, stream);
}


extern "C" int heal_grouped16_conv3x3(const float* x, const float* weight_frag, const float* bias, int n, int channels,
                                      int H, int W, int relu, float* y, void* stream) {
    HEAL_REQUIRE(n >= 1 && channels >= 16 && channels % 16 == 0 && H >= 1 && W >= 4 && W % 4 == 0,
                 "grouped16_conv3x3: needs channels %% 16 == 0 and W %% 4 == 0 (got C=%d W=%d)", channels, W);
    HEAL_REQUIRE(((uintptr_t)x & 15) == 0, "grouped16_conv3x3: x must be 16-B aligned");
    HEAL_REQUIRE(x && weight_frag && y && ((uintptr_t)weight_frag & 15) == 0, "grouped16_conv3x3: bad pointer");
    const int tiles_x = ceil_div(W, 16), tiles_y = ceil_div(H, 16);
    HEAL_REQUIRE(channels / 16 <= 65535 && n <= 65535, "grouped16_conv3x3: grid limit");
    k_grouped16_conv3x3<<<dim3(tiles_x * tiles_y, channels / 16, n), 256, 0, (hipStream_t)stream>>>(
        x, reinterpret_cast<const float4*>(weight_frag), bias, channels, H, W, tiles_x, relu, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}
