// Shared by the backward-weight kernels (sgan_wgrad.hip: exact fp32 MFMA; sgan_wgrad3.hip: split-bf16 MFMA).
#pragma once
#include "sgan_common.h"

#define SGW_MAX_PROB 8

struct SgWgradProb {
    const float* in;
    const float* dout;
    float* dw;
    float* dbias;
    const double* pro_stats;
    const float* pro_gamma;
    const float* pro_beta;
    const float* amax;   // fp16 planes: device scalar max|dout| (sgan_conv_wgrad_job.dout_amax), or null
    int32_t Hin, Win, in_ld;
    int32_t Hout, Wout, dout_ld;
    int32_t pro_count, pro_sq, pro_rep;
    int32_t nsplit;  // pixel-range splits of this problem
    int32_t z0;      // first blockIdx.z of this problem (z = z0 + phase * nsplit + split)
    int32_t Hp[SGAN_MAX_PHASES], Wp[SGAN_MAX_PHASES];
};

struct SgWgradParams {   // kernel argument: common layer description + up to 8 problems (see sgan_igemm.hip)
    int32_t Cin, Cout;
    int32_t is, os;
    int32_t w_ns;
    int32_t nphase, nprob;
    int32_t pro_act;
    int32_t planes_f16;  // split kernel: fp16 planes (every job brought dout_amax) instead of bf16
    int32_t thin_real;   // thin kernels: channels of the 4-wide side that carry data (Cin_logical / Cout_logical hint, else 4)
    float pro_slope, pro_eps;
    int32_t oa[SGAN_MAX_PHASES], ob[SGAN_MAX_PHASES], ntaps[SGAN_MAX_PHASES], ktot[SGAN_MAX_PHASES];
    SgTap taps[SGAN_MAX_TAPS];          // every phase's taps back to back (k * k in all)
    int32_t tap0[SGAN_MAX_PHASES];     // first tap of a phase
    SgWgradProb q[SGW_MAX_PROB];
};

struct SgWgradLocal {
    const float* in; const float* dout; float* dw; float* dbias;
    int32_t Hin, Win, Cin, in_ld, Hout, Wout, Cout, dout_ld, is, os, w_ns, nsplit;
    SgNorm pro;
};


// split-bf16 tiled kernel (sgan_wgrad3.hip): 1 = launched, 0 = layer not covered (caller runs the fp32 kernel), < 0 = error
int sg_launch_wgrad3(SgWgradParams& P, hipStream_t st);

// ---- one launch for a layer's backward-data and backward-weight (sgan_fused.hip) ----
int sg_wgrad3_fuse_plan(SgWgradParams& P, SgFusePlan* out);
int sg_build_wgrad_params(const sgan_conv_wgrad_job* jobs, int32_t n, SgWgradParams& P, bool allow_f16 = true);   // sgan_wgrad.hip: checks + parameter block of sgan_conv_wgrad_grouped
