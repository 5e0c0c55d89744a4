/* Timing-experiment entry points of libhumaniflow_amd.so (tools/ only; NOT part of the drop-in boundary of humaniflow_amd.h).
 * They launch ablated builds of product kernels to attribute a launch's time to its phases; most variants compute wrong results. */
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

#ifdef __cplusplus
}
#endif

#endif
