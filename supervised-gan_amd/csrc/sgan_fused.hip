// One launch for a layer's backward-data and backward-weight (split-bf16 mode).
//
// Both read the layer's output gradient and neither feeds the other; apart they are two latency-bound launches that each leave most
// of the chip idle (a few hundred workgroups whose serial k-loops set the launch time: DESIGN.md R2.5), and a hipGraph replays
// kernel nodes strictly one after the other, so the only way to run them side by side is to make them ONE grid: the first
// `ndg` workgroups run the backward-data body (sg_igemm3p_body / sg_igemm3_body), the rest the backward-weight body
// (sg_wgrad3_body), each with the workgroup coordinates it would have had in its own launch.  Nothing else changes: same
// parameter blocks, same results (the two bodies write disjoint tensors).
#define SG_KERNELS_ONLY
#include "sgan_igemm.hip"
#include "sgan_igemm3.hip"
#include "sgan_wgrad.h"
#include "sgan_wgrad3.hip"

// DV: 1 = patch kernel with <= 128 patch pixels, 2 = patch kernel up to 256, 3 = sg_igemm3 64 x 64 (two k-tiles per barrier),
//     5 = patch kernel, stride-2 gather (parity planes; ConvTranspose2d stride 2 backward-data);
//     6 = sg_igemm3 128 x 32 (<= 32 result channels: backward-data into the first PatchGAN layer).  Round 2's variant 4 (the exact-fp32
//         sg_igemm 128 x 32 tile for backward-data into a layer without a normalisation) is retired: that launch now runs variant 6 on
//         fp16 planes scaled by the gradient's published maximum -- an fp32-equivalent product at the split kernels' speed;
// WV: 1 = backward-weight 64 x 64 tiles, 2 = 32 x 128
// F16: the backward-data half on fp16 planes (every job brought the maximum of its gradient tensor); the backward-weight half is bf16
template <int DV, int WV, bool WPRO, bool F16>
__global__ __launch_bounds__(256) void sg_bwd_fused_kernel(const SgIgemmParams G, const SgWgradParams W, int ndg, int wx, int wy, int wmode, int dks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_warm_kernargs<(int)(sizeof(SgIgemmParams) + sizeof(SgWgradParams))>();
    const int b = blockIdx.x;
    if (b < ndg) {
        if constexpr (DV == 1) sg_igemm3p_body<64, 2, false, F16>(G, smem, b, ndg);
        else if constexpr (DV == 2) sg_igemm3p_body<64, 4, false, F16>(G, smem, b, ndg);
        else if constexpr (DV == 3) sg_igemm3_body<64, 64, 2, 2, false, F16, true>(G, smem, b % (ndg / dks), ndg / dks, b / (ndg / dks));      // (tile, k split)
        else if constexpr (DV == 5) sg_igemm3p_body<64, 6, false, F16, true>(G, smem, b, ndg);
        else sg_igemm3_body<128, 32, 4, 1, false, F16, false>(G, smem, b, ndg, 0);
    } else {
        // wmode 1 (SGAN_FUSED_WXCD=1, off by default): the backward-weight workgroups start at a multiple of 8 and every XCD takes a
        // CONTIGUOUS range of the pixel-range-major order, so that an L2 holds a few pixel ranges of dOut / x instead of all of them.
        // Measured stand-alone (tools/probe_bwd_split.py): 32 -> 64, six problems (36 ranges) 70.6 -> 64.8 us, but 128 -> 256 (8 ranges
        // of unequal work on eight XCDs) 130 -> 140 us and, three problems (4 ranges), 65 -> 85 us; inside the training step the
        // 32 -> 64 launches moved by 0-3 us.  Equal-work ranges would need the pixel split cut across problem boundaries.
        const int ndp = wmode ? (ndg + 7) & ~7 : ndg;
        if (b < ndp) return;
        int w = b - ndp;
        if (wmode) w = sg_xcd_remap(w, (int)gridDim.x - ndp);
        const int bx = w % wx, by = (w / wx) % wy, bz = w / (wx * wy);
        if constexpr (WV == 1) sg_wgrad3_body<64, 64, 2, 2, WPRO, false>(W, smem, bx, by, bz);
        else sg_wgrad3_body<32, 128, 1, 4, WPRO, false>(W, smem, bx, by, bz);
    }
}

template <int DV, int WV>
static void sg_fused_launch(const SgIgemmParams& P, const SgWgradParams& W, const SgFusePlan& pd, const SgFusePlan& pw, hipStream_t st) {
    static const int wmode = getenv("SGAN_FUSED_WXCD") ? (atoi(getenv("SGAN_FUSED_WXCD")) != 0) : 0;      // tuning knob (see the kernel)
    const dim3 grid((wmode ? ((pd.nblocks + 7) & ~7) : pd.nblocks) + pw.nblocks);
    const size_t lds = pd.lds > pw.lds ? pd.lds : pw.lds;
    if constexpr (DV == 6) {       // fp16 planes are asked for where they matter: backward-data into a layer without a normalisation
        if (P.planes_f16) {
            if (pw.pro) hipLaunchKernelGGL((sg_bwd_fused_kernel<DV, WV, true, true>), grid, dim3(256), lds, st, P, W, pd.nblocks, pw.gx, pw.gy, wmode, pd.ks);
            else hipLaunchKernelGGL((sg_bwd_fused_kernel<DV, WV, false, true>), grid, dim3(256), lds, st, P, W, pd.nblocks, pw.gx, pw.gy, wmode, pd.ks);
            return;
        }
    }
    if (pw.pro) hipLaunchKernelGGL((sg_bwd_fused_kernel<DV, WV, true, false>), grid, dim3(256), lds, st, P, W, pd.nblocks, pw.gx, pw.gy, wmode, pd.ks);
    else hipLaunchKernelGGL((sg_bwd_fused_kernel<DV, WV, false, false>), grid, dim3(256), lds, st, P, W, pd.nblocks, pw.gx, pw.gy, wmode, pd.ks);
}

// 0: launched; 1: this pair is not covered (launch sgan_conv_dgrad_grouped and sgan_conv_wgrad_grouped instead); < 0: error.
// workspace_bytes == -1: query -- KiB of workspace the pair wants (0: none), nothing is launched.
static int sg_conv_bwd_fused_impl(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw,
                                  int32_t dgrad_math, void* workspace, int64_t workspace_bytes, void* stream);
extern "C" int sgan_conv_bwd_fused(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw,
                                   int32_t dgrad_math, void* stream) {
    return sg_conv_bwd_fused_impl(djobs, nd, wjobs, nw, dgrad_math, nullptr, 0, stream);
}
extern "C" int sgan_conv_bwd_fused_ws(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw,
                                      int32_t dgrad_math, void* workspace, int64_t workspace_bytes, void* stream) {
    return sg_conv_bwd_fused_impl(djobs, nd, wjobs, nw, dgrad_math, workspace, workspace_bytes, stream);
}
static int sg_conv_bwd_fused_impl(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw,
                                  int32_t dgrad_math, void* workspace, int64_t workspace_bytes, void* stream) {
    const bool query = workspace_bytes == -1;
    static const int off = getenv("SGAN_NO_BWD_FUSION") ? 1 : 0;
    if (off) return query ? 0 : 1;
    SgIgemmParams P;
    SgWgradParams W;
    int rc = sg_build_dgrad_params(djobs, nd, P, true);      // fp16 planes when every job brought its gradient's maximum
    if (rc) return rc;
    rc = sg_build_wgrad_params(wjobs, nw, W, false);         // the backward-weight half of a fused launch stays on bf16 planes
    if (rc) return rc;
    if (wjobs[0].d->math != SGAN_MATH_BF16X3 || sg_dgrad_is_skinny(P)) return query ? 0 : 1;
    if (dgrad_math >= 0) P.math = dgrad_math;      // the two job lists may share descriptors: the backward-data mode comes apart
    const int e3 = sg_igemm3_eligible(P);
    if (e3 < 0) return e3;
    SgFusePlan pd, pw;
    if (e3 == 0) return query ? 0 : 1;      // exact-fp32 backward-data (tiny maps, SGAN_MATH_F32): the two grouped calls
    sg_igemm3_fuse_plan(P, &pd);
    if (pd.variant != 6 && P.planes_f16) {      // only variant 6 is instantiated with fp16 planes: the others run this pair on bf16 planes
        rc = sg_build_dgrad_params(djobs, nd, P, false);
        if (rc) return rc;
        if (dgrad_math >= 0) P.math = dgrad_math;
        sg_igemm3_fuse_plan(P, &pd);
    }
    if (pd.variant == 0 || pd.nblocks == 0) return query ? 0 : 1;
    sg_wgrad3_fuse_plan(W, &pw);
    if (pw.variant == 0 || pw.nblocks == 0) return query ? 0 : 1;
    if ((pd.lds > pw.lds ? pd.lds : pw.lds) > 160 * 1024) return query ? 0 : 1;     // the CU's LDS (the split kernels with two k-tiles per barrier take a little over 64 KB)
    const int64_t slab = (int64_t)P.q[0].Hout * P.q[0].Wout * P.N;
    const int64_t need = pd.ks > 1 ? (int64_t)pd.ks * slab * 4 : 0;
    if (query) return (int)((need + 1023) >> 10);
    if (need > 0 && (!workspace || workspace_bytes < need)) return 1;      // a split backward-data half needs its slabs: the caller runs the two grouped calls
    P.ksplit = pd.ks;
    P.slab = pd.ks > 1 ? (float*)workspace : nullptr;
    P.slab_stride = slab;
    hipStream_t st = (hipStream_t)stream;
    sg_prof_begin(st);
    switch (pd.variant * 10 + pw.variant) {
        case 11: sg_fused_launch<1, 1>(P, W, pd, pw, st); break;
        case 12: sg_fused_launch<1, 2>(P, W, pd, pw, st); break;
        case 21: sg_fused_launch<2, 1>(P, W, pd, pw, st); break;
        case 22: sg_fused_launch<2, 2>(P, W, pd, pw, st); break;
        case 31: sg_fused_launch<3, 1>(P, W, pd, pw, st); break;
        case 32: sg_fused_launch<3, 2>(P, W, pd, pw, st); break;
        case 51: sg_fused_launch<5, 1>(P, W, pd, pw, st); break;
        case 52: sg_fused_launch<5, 2>(P, W, pd, pw, st); break;
        case 61: sg_fused_launch<6, 1>(P, W, pd, pw, st); break;
        default: sg_fused_launch<6, 2>(P, W, pd, pw, st); break;
    }
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_bwd_fused_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    if (pd.ks > 1) return sg_launch_splitk_epilogue(P, st);      // sum of the slabs + the backward-data epilogue (activation derivative, norm-backward sums)
    return SGAN_OK;
}
