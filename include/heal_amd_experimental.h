/* heal_amd_experimental.h -- entry points of MEASURED-NEGATIVE kernels.
 *
 * These kernels are correct (each has a parity test) but lose against the production path at every shape of the BASELINE scenes
 * (DESIGN.md section 3 / 8 records the measurements), so they are NOT part of the shipped library or of the C ABI a reference
 * maintainer binds: they are compiled only with `HEAL_BUILD_EXPERIMENTAL=1 python -m heal_amd.build` (sources under
 * heal_amd/csrc/experimental/, plus the 16-channel-chunk instantiation of the Winograd kernel in csrc/conv3x3.hip).
 * Same conventions as include/heal_amd.h (device pointers, a hipStream_t, non-zero return + heal_last_error()).
 */
#ifndef HEAL_AMD_EXPERIMENTAL_H
#define HEAL_AMD_EXPERIMENTAL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* heal_gconv_conv3 (round 4): the BACK HALF of a ResNeXt bottleneck in one kernel -- the 32-group 3x3 convolution (conv2 + bn2 + relu)
 *   and the pointwise convolution behind it (conv3 + bn3 + identity + relu; opencood/models/sub_modules/resblock.py:110-121, stride 1,
 *   BatchNorms folded by the caller): y = act(W3 . relu(gconv3x3(x) + b2) + b3 (+ residual)).  The 2C-wide intermediate stays in LDS.
 *   x [n, width, H, W]; weight_q: the grouped weights in the layout heal_grouped_small_conv3x3 takes; w3_frag: W3 [cout, width] in MFMA
 *   A-fragment order (frag[mt][ks][lane] = W3[mt*16 + (lane & 15)][ks*4 + (lane >> 4)]); residual [n, cout, H, W] or NULL.
 *   Supported (heal_gconv_conv3_supported): 4 | 8 channels per group, width % 16 == 0, cout 64 | 128, W % 4 == 0.                        */
int heal_gconv_conv3_supported(int width, int group_channels, int cout, int H, int W);
int heal_gconv_conv3(const float* x, const float* weight_q, const float* b2, const float* w3_frag, const float* b3,
                     const float* residual, int n, int width, int group_channels, int cout, int H, int W, int relu, float* y,
                     void* stream);

/* heal_resnext_bottleneck: one fused kernel for a stride-1 ResNeXt bottleneck without downsample
 *   (opencood/models/sub_modules/resblock.py:100-122; 32 groups, width = 2*C, expansion 1):
 *   y = relu(conv3(relu(gconv2(relu(conv1(x)+b1))+b2))+b3+x), BatchNorms folded by the caller.
 *   x,y [n,C,H,W] (C = 64|128|256); w2 [2C, 2C/32, 3, 3]; b1,b2 [2C]; b3 [C];
 *   w1_frag / w3_frag: the 1x1 weights W1 [2C,C], W3 [C,2C] re-laid in MFMA A-fragment order
 *   frag[mt][ks][lane] = Wm[mt*16 + (lane & 15)][ks*4 + (lane >> 4)].                                */
int heal_resnext_bottleneck(const float* x, const float* w1_frag, const float* b1, const float* w2,
                            const float* b2, const float* w3_frag, const float* b3, int n, int channels, int H,
                            int W, float* y, void* stream);


/* heal_conv1x1_tiled: the same pointwise convolution on the 128 x 128 x 32 core (v_mfma_f32_32x32x2_f32, both operands through
 *   LDS) for its MFMA-bound shapes: weight is the PLAIN [Cout, Cin] row-major matrix (nn.Conv2d's storage, no fragment
 *   pre-layout); Cin % 32 == 0, Cout % 64 == 0, stride 1, no input gate.  d2s_k = 0: y [n, Cout, H, W]; d2s_k = k >= 1: the
 *   depth-to-space + channel-offset write of heal_conv1x1_d2s into y [n, dst_channels, H k, W k].  bias / residual may be NULL.
 *   heal_conv1x1_tiled_supported: 1 if the shape fills the chip with this tiling (>= 256 blocks), else 0.                    */
int heal_conv1x1_tiled_supported(int n, int cin, int cout, int H, int W);
int heal_conv1x1_tiled(const float* x, const float* weight, const float* bias, const float* residual, int n, int cin, int cout,
                       int H, int W, int act, int d2s_k, int dst_channels, int dst_channel_offset, float* y, void* stream);

/* heal_conv3x3_winograd4: the same operator with the F(4x4,3x3) transform (36 positions, 6x6 input windows, 4x4 output tiles:
 *   1/4 of the direct multiplications; csrc/conv3x3_wino4.hip).  u_frag: U = G g G^T in the lane-major order
 *   [ceil(Cout/32)][ceil(Cin/16)][wave 8][lane 64][xi_i 9][ks 4] (heal_amd.ops.conv3x3_winograd4_fragments).  fp32; ~2e-5 of
 *   the output scale against float64 in the worst case measured (F(2x2,3x3): ~1e-6).                                          */
int heal_conv3x3_winograd4(const float* x, const float* u_frag, const float* bias, const float* residual, int n, int cin,
                           int cout, int H, int W, int relu, float* y, void* stream);


/* heal_bev_pool_scatter_multi (round 6): heal_bev_pool_scatter for up to 4 independent problems in ONE launch (measured: the launch reaches 0.41 of the HBM roof, the stream join it needs costs the step 1.2 ms: profiles/r06_k4_shared_launch.json) -- the camera agents of every
 *   camera modality of a scene (different heads, image sizes, frustums, workspaces; the same ceil(D / 16)).  Arrays of n_problems entries;
 *   dx_host / bx_host / nx_host hold 3 values per problem; every problem has its own workspace (heal_bev_pool_pm_workspace) and its own
 *   consumer (heal_bev_stem_block / heal_bev_pool_emit) afterwards.  Same arithmetic per problem as the single launch.                  */
int heal_bev_pool_scatter_multi(int n_problems, const float* const* heads, const int32_t* head_strides, const float* const* frustums,
                                const float* const* cam_mats, const int32_t* n_agents, const int32_t* n_cams, const int32_t* D,
                                const int32_t* fH, const int32_t* fW, const int32_t* channels, const float* dx_host,
                                const float* bx_host, const int32_t* nx_host, void* const* ws, const size_t* ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
