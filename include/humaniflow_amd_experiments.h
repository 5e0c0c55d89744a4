/* Timing-experiment entry points.  They are NOT in libhumaniflow_amd.so: `python -m humaniflow_amd.build --experiments` compiles the same
 * sources with -DHMF_EXPERIMENTS into tools/libhumaniflow_amd_experiments.so, which only the scripts under tools/ load.  They launch ablated
 * builds of product kernels to attribute a launch's time to its phases; most variants compute WRONG results.  In that build
 * hmf_conv2d_nhwc additionally accepts tile = {1,2,3,4,7,8}064064, hmf_conv_pair variant != 0 and hmf_smpl_lbs impl = 16 * ABL. */
#ifndef HUMANIFLOW_AMD_EXPERIMENTS_H
#define HUMANIFLOW_AMD_EXPERIMENTS_H

#ifdef __cplusplus
extern "C" {
#endif

/* Timing experiments on the same kernel (tools/winograd_ablate.py): variant 0 = hmf_conv_winograd; 10 = three waves per SIMD; 20 = the input
 * transform of k-group g+1 issued with the MFMAs of group g; 1 / 2 / 3 / 13 drop the patch fetch and / or the output transform and compute
 * WRONG results.  Not used by the product path. */
int hmf_conv_winograd_variant(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                              int Cin, int Cout, int relu, int variant, void* stream);

/* csrc/conv_winograd_nw.hip (round 5, measured slower than hmf_conv_winograd and not shipped): the same convolution, bitwise the same result, with a
 * work-group of 8 x 16 pixels x 128 output channels and a wave that owns all sixteen transform positions of one 32-channel tile.
 * HMF_ERR_UNSUPPORTED unless Cin % 64 == 0, Cout % 128 == 0 and W >= 16.  hmf_debug_set_wino_nw_mode: timing ablations (& 1 no next-slab patch DMA,
 * & 2 no operand stream, & 4 no window reads; wrong results when != 0). */
int hmf_conv_winograd_nw(const float* x, const float* w_packed, const float* bias, const float* residual, float* y, int B, int H, int W,
                         int Cin, int Cout, int relu, void* stream);
int hmf_debug_set_wino_nw_mode(int mode);

/* Diagnostic: device buffer of 23 x 8 int64 that k_flow_chain's tile 0 fills with shader-clock stamps per part phase
 * (tools/flow_timeline.py); NULL (default) disables it. */
int hmf_debug_set_timeline(void* dev_ptr);
/* Diagnostic: device buffer of >= 256 int64; every k_lbs_mfma block stores its shader-clock span (effective clock =
 * span / event time, tools/lbs_tune.py); NULL disables it. */
int hmf_debug_set_lbs_cycles(void* dev_ptr);
/* k_lbs_mfma16 (tools/lbs16_tune.py): shader-clock cycles summed over all waves since the last call -- [0] staging + waits at the segment barriers, [1] tile start ->
 * first k-group, [2] main loop, [3] skinning epilogue, [4] number of wave tiles; [5..7] unused.  Reads and clears. */
int hmf_debug_lbs16_phases(unsigned long long* out_host);

/* Diagnostic (tools/wino_probe.py): stamps of the last k_conv_winograd launch, 8 int64 per work-group: shader clock at start, first patch in LDS,
 * main loop done, end; cycles between a slab's top and its second barrier summed over the slabs; the 100 MHz counter at start and end; unused. */
int hmf_debug_wino_stamps(long long* out_host, int blocks);

/* tools/wgw_probe.py: number of work-groups hmf_conv_wgrad_winograd aims for when it splits the tiles into slices (default 512). */
int hmf_debug_set_wgw_target_wgs(int n);
int hmf_debug_set_wgw_single_stage(int on);     /* 1: the single-staged 32-tile-chunk form of k_wgw (csrc/conv_wgrad_wino.hip) */

/* Diagnostic (tools/stem_probe.py): stamps of the last k_stem_conv<POOL> launch, 6 int64 per work-group (shader clock at start, patch in LDS, main
 * loop done, end; the 100 MHz counter at start and end). */
int hmf_debug_stem_stamps(long long* out_host, int blocks);
/* the same for the polyphase stem (csrc/stem_wino.hip): 12 values per work-group -- shader clock at start, -, main loop done, tile in LDS, end; 100 MHz counter at
 * start / end; wave 1 main loop done; HW_ID; XCC_ID */
int hmf_debug_stem_wino_stamps(long long* out_host, int blocks);

/* Diagnostic (tools/gemm1x1_probe.py --timeline): stamps of the last k_conv_gemm launch, 6 int64 per work-group: shader clock at start, first
 * chunk visible, main loop done, stores retired; the constant 100 MHz counter at start and end.  Work-groups 0 .. blocks-1 (<= 4096). */
int hmf_debug_gemm_stamps(long long* out_host, int blocks);

/* Round-6 what-if builds of the 1x1 family (tools/whatif_1x1.py; timing only, WRONG results).  hmf_debug_set_gemm_whatif: & 1 = k_conv_gemm's epilogue without
 * its transposition through LDS, & 2 = the activation panel of the K <= 256 launches fetched for the prologue chunks only.  hmf_debug_set_pair_whatif:
 * & 32 = the layer1 seams without their W3 / W1 operand streams. */
int hmf_debug_set_gemm_whatif(int mask);
int hmf_debug_set_pair_whatif(int mask);
/* & 1 = hmf_flow_forward does not launch the kinematic-loop kernel at all (the step's ceiling for any faster k_flow_chain). */
int hmf_debug_set_flow_whatif(int mask);

#ifdef __cplusplus
}
#endif

#endif
