// Shared by the implicit-GEMM kernels (sgan_igemm.hip: exact fp32 MFMA; sgan_igemm3.hip: split-bf16 MFMA): the kernel
// argument, the per-problem view, the XCD-aware work-item order and the host-side tiling helpers.
#pragma once
#include "sgan_common.h"

#define SG_MAX_PROB 8

// One launch serves up to SG_MAX_PROB independent problems of the SAME layer type (same kind / k / stride /
// pad / channels, hence the same taps) but their own tensors and spatial sizes -- e.g. the matching layer of
// the three discriminators on the fake and on the real batch.  blockIdx.x walks the concatenated M tiles of
// every (problem, phase); `tile0` is the prefix table.
struct SgProb {
    const float* in;    // gathered tensor
    float* out;         // result tensor
    const float* w;     // master weight
    const void* wp;     // split-bf16 copy of the same weights (sgan_pack_weights), or null
    const float* bias;  // [N] or null
    const float* xref;  // dact epilogue: forward tensor at the output positions, or null
    double* stats;      // [2N]: fwd (sum, sumsq) of the result, or bwd sums (s1, s2); or null
    const double* pro_stats;  // prologue norm of the gathered tensor (or null)
    const float* pro_gamma;
    const float* pro_beta;
    const double* xn_stats;   // norm the forward consumer applied to xref (or null)
    const float* xn_gamma;
    const float* xn_beta;
    const float* amax;  // split kernels with fp16 planes: device scalar max|gathered tensor| (backward-data: sgan_conv_dgrad_job.dout_amax) or null
    int32_t Hin, Win, in_ld;     // gathered tensor geometry
    int32_t Hout, Wout, out_ld;  // result tensor geometry
    int32_t xref_ld, pro_count, xn_count;
    int32_t pro_sq, xn_sq;   // sum -> sumsq distance of the two norm statistics (0 = channel count)
    int32_t stats_sq;        // same for the statistics this launch accumulates (0 = N)
    int32_t pro_rep, xn_rep, stats_rep;   // replica strides (sgan_norm_desc.rep_stride) of the two norms read / the statistics written
    int32_t accum;           // out += result (backward-data into a tensor with two forward consumers)
    int32_t Hp[SGAN_MAX_PHASES], Wp[SGAN_MAX_PHASES];
    int32_t tile0[SGAN_MAX_PHASES];  // first blockIdx.x of (this problem, phase)
};

struct SgIgemmParams {   // the kernel argument (~2.3 KB)
    int32_t Ck, N;        // GEMM-K channels (gathered tensor), GEMM-N channels (result tensor)
    int32_t is, os;
    int32_t w_ns, w_ks;   // element strides of B[k-channel][n] inside a tap slab
    int32_t out_act;
    int32_t nphase, nprob;
    int32_t ksplit;       // > 1 (single problem only): split-K, raw partial tiles go to `slab`
    int32_t n_real;       // result channels that carry data (<= N; the rest is zero padding), small-N kernel only
    int32_t pro_act, xn_act;
    int32_t math;         // SGAN_MATH_*
    int32_t planes_f16;   // split kernels: operand planes are fp16 (forward) instead of bf16 (backward-data)
    float pro_slope, xn_slope, pro_eps, xn_eps;
    int32_t oa[SGAN_MAX_PHASES], ob[SGAN_MAX_PHASES], ntaps[SGAN_MAX_PHASES], ktot[SGAN_MAX_PHASES];
    SgTap taps[SGAN_MAX_TAPS];          // every phase's taps back to back (k * k in all)
    int32_t tap0[SGAN_MAX_PHASES];     // first tap of a phase
    // patch-stationary split kernel (sgan_igemm3.hip: sg_igemm3p_kernel): per phase, the first tap offset and the patch extent
    // (an 8 x 8 block of result pixels reads patch rows [pdy0, pdy0 + pph) x columns [pdx0, pdx0 + ppw) relative to its corner)
    int32_t pdy0[SGAN_MAX_PHASES], pdx0[SGAN_MAX_PHASES], pph[SGAN_MAX_PHASES], ppw[SGAN_MAX_PHASES];
    float* slab;          // [ksplit][Hout*Wout][N] fp32 partials (caller workspace)
    int64_t slab_stride;  // Hout*Wout*N
    SgProb q[SG_MAX_PROB];
};

// The view of ONE problem the kernel bodies work with (scalarised by the compiler).
struct SgLocal {
    const float* in; float* out; const float* w; const float* bias; const float* xref; double* stats;
    int32_t Hin, Win, Ck, in_ld, Hout, Wout, N, out_ld, xref_ld, is, os, w_ns, w_ks, out_act, ksplit;
    float* slab; int64_t slab_stride;
    int32_t stats_sq, accum, stats_rep;
    float a_scale, out_scale;   // fp16 planes: the gathered operand is staged times a_scale (a power of two), the accumulator leaves times out_scale
    SgNorm pro, xn;
};


__device__ __forceinline__ SgLocal sg_local(const SgIgemmParams& G, int g) {
    const SgProb& Q = G.q[g];
    SgLocal P;
    P.in = Q.in; P.out = Q.out; P.w = Q.w; P.bias = Q.bias; P.xref = Q.xref; P.stats = Q.stats;
    P.Hin = Q.Hin; P.Win = Q.Win; P.Ck = G.Ck; P.in_ld = Q.in_ld; P.Hout = Q.Hout; P.Wout = Q.Wout; P.N = G.N;
    P.out_ld = Q.out_ld; P.xref_ld = Q.xref_ld; P.is = G.is; P.os = G.os; P.w_ns = G.w_ns; P.w_ks = G.w_ks;
    P.out_act = G.out_act; P.ksplit = G.ksplit; P.slab = G.slab; P.slab_stride = G.slab_stride;
    P.pro.stats = Q.pro_stats; P.pro.gamma = Q.pro_gamma; P.pro.beta = Q.pro_beta; P.pro.count = Q.pro_count;
    P.pro.eps = G.pro_eps; P.pro.act = G.pro_act; P.pro.slope = G.pro_slope; P.pro.sq_stride = Q.pro_sq; P.pro.rep_stride = Q.pro_rep;
    P.stats_sq = Q.stats_sq ? Q.stats_sq : G.N; P.accum = Q.accum; P.stats_rep = Q.stats_rep;
    P.xn.stats = Q.xn_stats; P.xn.gamma = Q.xn_gamma; P.xn.beta = Q.xn_beta; P.xn.count = Q.xn_count;
    P.xn.eps = G.xn_eps; P.xn.act = G.xn_act; P.xn.slope = G.xn_slope; P.xn.sq_stride = Q.xn_sq; P.xn.rep_stride = Q.xn_rep;
    // fp16 planes: the packed weights hold w * 2^SGAN_F16_WEIGHT_SHIFT; a gathered gradient is scaled up by its own power of two
    const int sh = (G.planes_f16 && Q.amax) ? sg_f16_shift(*Q.amax) : 0;
    P.a_scale = sg_pow2(sh);
    P.out_scale = G.planes_f16 ? sg_pow2(-SGAN_F16_WEIGHT_SHIFT - sh) : 1.f;
    return P;
}

// XCD-aware work-item order (MI355X: 8 XCDs, each with a private 4 MB L2; workgroups are dealt round-robin
// over the XCDs in launch order, so launch id b lands on the XCD "b % 8").  Give every XCD one CONTIGUOUS
// range of work items: tiles that share operand rows (same M tile, neighbouring M tiles, all N tiles) then
// hit the same L2 instead of every L2 having to hold the whole activation tensor plus the weights.
// Bijective for any count (placement is a speed matter only, never correctness).
__device__ __forceinline__ int sg_xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// blockIdx.x -> (problem, phase, M tile) through the prefix table
__device__ __forceinline__ void sg_decode_tile(const SgIgemmParams& G, int bx, int& g, int& phz, int& mtile) {
    g = 0;
    phz = 0;
    for (int gi = 0; gi < G.nprob; ++gi)
        for (int ph = 0; ph < G.nphase; ++ph)
            if (bx >= G.q[gi].tile0[ph]) { g = gi; phz = ph; }
    mtile = bx - G.q[g].tile0[phz];
}


// ---- host helpers (defined in sgan_igemm.hip) ----
int sg_fill_tiles(SgIgemmParams& P, int rows_per_tile);   // blockIdx prefix table over (problem, phase); returns total M tiles
int sg_max_k(const SgIgemmParams& P);
long sg_total_tiles(const SgIgemmParams& P, int BM);
int sg_plan_ksplit(const SgIgemmParams& P, int BM, int BN);
int sg_launch_splitk_epilogue(const SgIgemmParams& P, hipStream_t st);
// split-bf16 path (sgan_igemm3.hip): 1 = the launch runs there, 0 = not covered, < 0 = error (packed weights missing)
int sg_igemm3_eligible(const SgIgemmParams& P);
int sg_launch_igemm3(SgIgemmParams& P, hipStream_t st, float* ws, int64_t ws_bytes);
int64_t sg_igemm3_workspace_need(const SgIgemmParams& P);

int sg_launch_head2(SgIgemmParams& P, hipStream_t st);      // sgan_head.hip: one-channel stride-1 head forward; 1 = not covered

// ---- one launch for a layer's backward-data and backward-weight (sgan_fused.hip) ----
int sg_igemm3_fuse_plan(SgIgemmParams& P, SgFusePlan* out);
int sg_build_dgrad_params(const sgan_conv_dgrad_job* jobs, int32_t n, SgIgemmParams& P, bool allow_f16 = true);   // sgan_igemm.hip: the argument checks + parameter block of sgan_conv_dgrad_grouped
bool sg_dgrad_is_skinny(const SgIgemmParams& P);                                           // routed to the direct small-N kernels
