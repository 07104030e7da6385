// Backward pass of the PatchGAN logits head (Conv2d(C -> 1, k4, s1, p2), models/networks.py:832-835) in ONE launch: the gradient of its
// input (with the LeakyReLU derivative of the normalised forward tensor and the two norm-backward sums, as the backward-data epilogues
// of sgan_igemm*.hip do) AND its weight / bias gradient.  Round 2 ran them as two launches of two generic kernels
// (sg_wgrad_thin_kernel<4,cout4> 20 us + sg_conv_c4_kernel 13-18 us per discriminator pass) that both stream the same 256-channel map.
//
// The layer has ONE result channel, so both products are rank-16 per pixel: with G[q][tap] = dz[q - tap + pad] (the 16 logits
// gradients that input pixel q reaches)
//     dX[q][c]    = sum_tap G[q][tap] * W[tap][c]                       (then * act'(norm(x[q][c])), sums s1 += v, s2 += v * xhat)
//     dW[tap][c] += sum_q   G[q][tap] * act(norm(x[q][c]))              (summed over INPUT pixels: no halo of x is needed)
// A thread owns one CHANNEL: its K * K weights, its mean / rstd and its dW / s1 / s2 accumulators live in registers.  A workgroup is
// 4 waves x 64 channels on an 8-row x 16-column tile of input pixels: wave w takes rows 2w, 2w+1 as half rows of 8 pixels, reads
// x[q][c] once (64 consecutive floats per pixel and wave: coalesced; the next half row's 8 loads in flight while this one is
// computed), the tile's (8+K-1) x (16+K-1) patch of dz sits in LDS and is read as broadcasts; the waves' dW / sums meet in LDS, so a
// workgroup issues K*K + 2 wave-atomics.  32 fp32 FMA per (pixel, channel); HBM: x once, dX once.  What bounds it is the length of
// one wave's chain (setup + 4 half rows), not bandwidth: the first version (a wave = 64 pixels of an 8 x 8 tile, loads under
// branches, every wave its own atomics) took 61 us, this one 19 us for the six problems of an fcgan D step (tools/abl_head.sh).
// Exact fp32 (the thin kernels it replaces were exact fp32 too).  Algorithmic bytes per launch: 8 B per (pixel, channel).
//
// Reference ops replaced: convolution_backward (input, weight, bias) of the last nn.Conv2d of NLayerDiscriminator
// (models/networks.py:832-835) + the LeakyReLU backward in front of it.
#include "sgan_igemm.h"
#include "sgan_wgrad.h"

#ifndef SGH_TY
#define SGH_TY 8      // tile rows: 2 per wave (measured: 16 rows 24.6 us, 8 rows 19.7 us for the six problems of an fcgan D step)
#endif
#define SGH_TX 16         // tile columns: two half rows of 8 pixels (one half row's loads of x are in flight together)
#define SGH_CH 64         // channels per workgroup = lanes of a wave; the 4 waves take 4 row groups of the tile
#define SGH_MAXK 4
#ifndef SGH_ABL
#define SGH_ABL 0     // timing-only ablations (tools/abl_head.sh): 1 no dW atomics, 2 no sum atomics, 4 no dX store, 8 no x loads
#endif

struct SgHeadProb {
    const float* x; float* din; const float* dz; const float* w; float* dw; float* dbias; double* sums;
    const double* xn_stats; const float* xn_gamma; const float* xn_beta;
    int32_t H, W, x_ld, din_ld, Ho, Wo, dz_ld, xn_count, xn_sq, xn_rep, sums_sq, sums_rep, tile0, tiles_x;
};
struct SgHeadParams {
    int32_t C, k, pad, nprob, w_transposed, xn_act;
    float xn_slope, xn_eps;
    SgHeadProb q[SG_MAX_PROB];
};

template <int K>
__global__ __launch_bounds__(256) void sg_head_bwd_kernel(const SgHeadParams G) {
    sg_warm_kernargs<(int)sizeof(SgHeadParams)>();
    constexpr int PS = SGH_TX + K - 1, PR = SGH_TY + K - 1;      // patch of dz (columns, rows): input pixel q reaches outputs q - tap + pad
    constexpr int RW = SGH_TY / 4;                               // rows per wave
    __shared__ float dzs[PR * PS];
    __shared__ float bred[4];
    __shared__ float red[4][K * K][SGH_CH];
    __shared__ double reds[4][2][SGH_CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int g = 0;
    for (int gi = 1; gi < G.nprob; ++gi)
        if ((int)blockIdx.x >= G.q[gi].tile0) g = gi;
    const SgHeadProb& Q = G.q[g];
    const int tile = blockIdx.x - Q.tile0;
    const int ty0 = (tile / Q.tiles_x) * SGH_TY, tx0 = (tile % Q.tiles_x) * SGH_TX;
    const int c = blockIdx.y * SGH_CH + lane, C = G.C, pad = G.pad;
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;

    // the tile's patch of dz (zero outside the logits map): patch (r, s) = dz[ty0 + r - (K - 1 - pad)][tx0 + s - (K - 1 - pad)]
    for (int i = tid; i < PR * PS; i += 256) {
        const int r = i / PS, s = i - r * PS;
        const int oy = ty0 + r - (K - 1 - pad), ox = tx0 + s - (K - 1 - pad);
        const bool in = (unsigned)oy < (unsigned)Q.Ho && (unsigned)ox < (unsigned)Q.Wo;
        const float v = Q.dz[((int64_t)(in ? oy : 0) * Q.Wo + (in ? ox : 0)) * Q.dz_ld];
        dzs[i] = in ? v : 0.f;
    }
    // bias gradient: the first workgroup of a problem sums every logit gradient (a few thousand values)
    const bool bias_wg = tile == 0 && blockIdx.y == 0 && Q.dbias;
    if (bias_wg) {
        float s = 0.f;
        for (int i = tid; i < Q.Ho * Q.Wo; i += 256) s += Q.dz[(int64_t)i * Q.dz_ld];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) bred[wave] = s;
    }
    // this thread's channel: its K * K weights, the normalisation of the forward tensor.  Unconditional loads from clamped addresses,
    // masked afterwards: a load under a branch makes the compiler wait for it at the join (a chain of memory round trips)
    float wv[K * K], mean = 0.f, rstd = 1.f, gam = 1.f, bet = 0.f;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float wl = G.w_transposed ? Q.w[((int64_t)t * C + cc) * 4] : Q.w[(int64_t)t * 4 * C + cc];
        wv[t] = cok ? wl : 0.f;
    }
    const bool xnorm = Q.xn_stats != nullptr;
    if (xnorm) {
        SgNorm n;
        n.stats = Q.xn_stats; n.gamma = Q.xn_gamma; n.beta = Q.xn_beta; n.count = Q.xn_count; n.eps = G.xn_eps; n.act = G.xn_act;
        n.slope = G.xn_slope; n.sq_stride = Q.xn_sq; n.rep_stride = Q.xn_rep;
        sg_mean_rstd(n, C, cc, mean, rstd);
        gam = Q.xn_gamma ? Q.xn_gamma[cc] : 1.f;
        bet = Q.xn_beta ? Q.xn_beta[cc] : 0.f;
    }
    const float neg = G.xn_act == SGAN_ACT_NONE ? 1.f : (G.xn_act == SGAN_ACT_RELU ? 0.f : G.xn_slope);
    SG_SYNC();
    if (bias_wg && tid == 0) atomicAdd(Q.dbias, (bred[0] + bred[1]) + (bred[2] + bred[3]));

    float dwa[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) dwa[t] = 0.f;
    double s1 = 0.0, s2 = 0.0;
    // this wave: rows [RW * wave, RW * wave + RW) of the tile, each as two half rows of 8 pixels; the NEXT half row's eight loads of x
    // are in flight while this one is computed
    const int py0 = wave * RW;
    const int rows = min(RW, Q.H - ty0 - py0);      // <= 0: this wave's rows lie below the map
    float xs[8], xn[8];
    auto load_half = [&](int it, float (&dst)[8]) {
        const int qy = min(ty0 + py0 + (it >> 1), Q.H - 1);
#pragma unroll
        for (int px = 0; px < 8; ++px) {
            const int qx = min(tx0 + (it & 1) * 8 + px, Q.W - 1);
            dst[px] = (SGH_ABL & 8) ? 0.f : Q.x[((int64_t)qy * Q.W + qx) * Q.x_ld + cc];
        }
    };
    if (rows > 0) load_half(0, xs);
    for (int it = 0; it < 2 * rows; ++it) {
        const int py = py0 + (it >> 1), hx = (it & 1) * 8, qy = ty0 + py;
        load_half(it + 1, xn);      // (a clamped re-read after the last half row: never used)
        // the K patch rows this half row reaches, once (broadcast reads): every register index below is a compile-time constant
        float prow[K][8 + K - 1], h1 = 0.f, h2 = 0.f;      // fp32 over the 8 pixels of a half row, fp64 across half rows / tiles
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int sx = 0; sx < 8 + K - 1; ++sx) prow[ky][sx] = dzs[(py + K - 1 - ky) * PS + hx + sx];
#pragma unroll
        for (int px = 0; px < 8; ++px) {
            const bool pok = tx0 + hx + px < Q.W;      // columns past the map: masked out of dW and the sums, nothing stored
            const float xhat = (xs[px] - mean) * rstd;
            const float y = xnorm ? gam * xhat + bet : xs[px];
            const float a = pok ? (y > 0.f ? y : y * neg) : 0.f;            // act(norm(x)): the backward-weight operand
            float acc = 0.f;
            // G[q][tap (ky, kx)] = dz[q - (ky, kx) + pad] = patch[(py + K - 1 - ky), (px + K - 1 - kx)]
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float gq = prow[ky][px + K - 1 - kx];
                    acc = fmaf(gq, wv[ky * K + kx], acc);
                    dwa[ky * K + kx] = fmaf(gq, a, dwa[ky * K + kx]);
                }
            const float v = pok ? acc * (y > 0.f ? 1.f : neg) : 0.f;
            h1 += v;
            h2 = fmaf(v, xhat, h2);
            if (cok && pok && !(SGH_ABL & 4)) Q.din[((int64_t)qy * Q.W + tx0 + hx + px) * Q.din_ld + c] = v;
        }
        s1 += (double)h1;
        s2 += (double)h2;
#pragma unroll
        for (int px = 0; px < 8; ++px) xs[px] = xn[px];
    }
    // the four waves hold partial sums of the SAME 64 channels: one pass through LDS, then a quarter of the atomics (fp32 atomics on the
    // weight gradient execute memory-side, ~75 ns per cache-line operation and serial per line: their count is what this launch's tail costs)
#pragma unroll
    for (int t = 0; t < K * K; ++t) red[wave][t][lane] = dwa[t];
    reds[wave][0][lane] = s1;
    reds[wave][1][lane] = s2;
    SG_SYNC();
    if (!cok) return;
    if (Q.dw && !(SGH_ABL & 1)) {
#pragma unroll
        for (int j = 0; j < (K * K + 3) / 4; ++j) {
            const int t = wave + 4 * j;      // master layout [tap][4][C], row 0: 64 consecutive channels of one tap per wave instruction
            if (t < K * K) atomicAdd(Q.dw + (int64_t)t * 4 * C + c, (red[0][t][lane] + red[1][t][lane]) + (red[2][t][lane] + red[3][t][lane]));
        }
    }
    if (Q.sums && !(SGH_ABL & 2) && wave < 2) {
        double* st = sg_stat_replica(Q.sums, Q.sums_rep, blockIdx.x);
        const double v = (reds[0][wave][lane] + reds[1][wave][lane]) + (reds[2][wave][lane] + reds[3][wave][lane]);
        atomicAdd(&st[(wave ? (Q.sums_sq ? Q.sums_sq : C) : 0) + c], v);
    }
}

// 0: launched; 1: not this layer type (the caller issues the generic calls); < 0: error
extern "C" int sgan_conv_head_bwd(const sgan_conv_dgrad_job* djobs, const sgan_conv_wgrad_job* wjobs, int32_t n, void* stream) {
    static const int off = getenv("SGAN_NO_HEAD_BWD") ? 1 : 0;
    if (off) return 1;
    SGAN_CHECK(djobs && n >= 1 && n <= SG_MAX_PROB, "1..%d jobs", SG_MAX_PROB);
    const sgan_conv_desc* d0 = djobs[0].d;
    if (!d0) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    if (d0->kind != SGAN_CONV || d0->stride != 1 || (d0->k != 3 && d0->k != 4) || d0->Cout != 4 || d0->Cout_logical != 1 || d0->Cin < 64 ||
        2 * d0->pad > d0->k)
        return 1;
    SgHeadParams P;
    memset(&P, 0, sizeof(P));
    P.C = d0->Cin; P.k = d0->k; P.pad = d0->pad; P.nprob = n; P.w_transposed = djobs[0].w_transposed ? 1 : 0;
    const sgan_norm_desc* x0 = djobs[0].x ? djobs[0].x_norm : nullptr;
    P.xn_act = x0 ? x0->act : SGAN_ACT_NONE; P.xn_slope = x0 ? x0->slope : 0.f; P.xn_eps = x0 ? x0->eps : 0.f;
    int tiles = 0;
    for (int g = 0; g < n; ++g) {
        const sgan_conv_dgrad_job& J = djobs[g];
        const sgan_conv_desc* d = J.d;
        if (!d || d->kind != d0->kind || d->k != d0->k || d->stride != d0->stride || d->pad != d0->pad || d->Cin != d0->Cin || d->Cout != d0->Cout)
            return sgan_fail(SGAN_ERR_INVALID, "grouped problems must be the same layer type");
        if (d->Hout != d->Hin + 2 * d->pad - d->k + 1 || d->Wout != d->Win + 2 * d->pad - d->k + 1) return sgan_fail(SGAN_ERR_INVALID, "conv geometry mismatch");
        SGAN_CHECK(J.dout && J.w && J.din && J.x, "null tensor in job %d", g);
        if (J.accumulate || (J.w_transposed != 0) != (djobs[0].w_transposed != 0)) return 1;
        const sgan_norm_desc* xn = J.x_norm;
        if ((xn ? xn->act : SGAN_ACT_NONE) != P.xn_act) return sgan_fail(SGAN_ERR_INVALID, "grouped jobs must share the activation");
        if (wjobs && (wjobs[g].dout != J.dout || wjobs[g].in != J.x || wjobs[g].d != J.d)) return 1;      // the two job lists must describe the same pass
        SgHeadProb& Q = P.q[g];
        Q.x = J.x; Q.din = J.din; Q.dz = J.dout; Q.w = J.w; Q.dw = wjobs ? wjobs[g].dw : nullptr; Q.dbias = wjobs ? wjobs[g].dbias : nullptr;
        Q.sums = J.bwd_sums;
        Q.xn_stats = xn ? xn->stats : nullptr; Q.xn_gamma = xn ? xn->gamma : nullptr; Q.xn_beta = xn ? xn->beta : nullptr;
        Q.xn_count = xn ? xn->count : 1; Q.xn_sq = xn ? xn->sq_stride : 0; Q.xn_rep = xn ? xn->rep_stride : 0;
        Q.H = d->Hin; Q.W = d->Win; Q.x_ld = J.x_ld; Q.din_ld = J.din_ld; Q.Ho = d->Hout; Q.Wo = d->Wout; Q.dz_ld = J.dout_ld;
        Q.sums_sq = J.bwd_sums_sq_stride; Q.sums_rep = J.bwd_sums_rep_stride;
        Q.tiles_x = (d->Win + SGH_TX - 1) / SGH_TX;
        Q.tile0 = tiles;
        tiles += Q.tiles_x * ((d->Hin + SGH_TY - 1) / SGH_TY);
    }
    if (tiles == 0) return SGAN_OK;
    // Every tile adds its dW with same-address fp32 atomics (~75 ns each, serial per cache line): fine for the PatchGAN heads (2 x 45
    // tiles per weight tensor: 19 us against 36 for the two generic launches), break-even at ~150 tiles (131 x 131 map: 42 us against
    // 44), hopeless on a full-size map (CRN output conv, 64 -> 1 on 512 x 512 = 2048 tiles: 300 us against ~70).  Past the limit the
    // caller runs the generic pair.
    static const int max_tiles = getenv("SGAN_HEAD_BWD_MAX_TILES") ? atoi(getenv("SGAN_HEAD_BWD_MAX_TILES")) : 128;
    for (int g = 0; g < n && wjobs; ++g) {
        int per_dw = 0;
        for (int h = 0; h < n; ++h)
            if (P.q[h].dw == P.q[g].dw) per_dw += P.q[h].tiles_x * ((P.q[h].H + SGH_TY - 1) / SGH_TY);
        if (P.q[g].dw && per_dw > max_tiles) return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    sg_prof_begin(st);
    const dim3 grid(tiles, (P.C + SGH_CH - 1) / SGH_CH);
    if (P.k == 4) hipLaunchKernelGGL(sg_head_bwd_kernel<4>, grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL(sg_head_bwd_kernel<3>, grid, dim3(256), 0, st, P);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_head_bwd_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}


// ------------------------------------------------------------------------------------------
// Forward of the same head (and of any one-channel stride-1 Conv2d with k 3 or 4 on >= 64 channels: the CRN output conv) in the
// shape of the backward kernel above.  sg_conv_head_kernel (sgan_igemm.hip) walks the channels in blocks of 32 through one LDS
// patch with two barriers per block -- a chain of 16 barrier-separated phases that a launch of ~230 workgroups cannot overlap
// (22 us for 0.1 GFLOP).  Here a thread owns one CHANNEL and the input is the stationary operand: a wave walks the
// (R + K - 1) x (16 + K - 1) input patch of an R x 16 tile of result pixels for its 64 channels, every loaded value is normalised /
// activated once and feeds the up to K * K results it reaches (R * 16 accumulators in registers, every index a compile-time constant),
// the next patch row's loads are in flight while this row is computed, and there is no barrier until the end: the 64 lanes'
// accumulators are folded with a butterfly (31 or 63 shuffles), the waves' sums meet in LDS atomics, one thread per pixel writes.
// A workgroup = 4 waves = all channel chunks of one tile (C = 256: one chunk each) or, with fewer chunks than waves, several
// row groups.  Exact fp32.  Reference op: the last nn.Conv2d of NLayerDiscriminator (models/networks.py:832-835) forward.
// ------------------------------------------------------------------------------------------
#define SGF_TW 16
template <int K, int R>
__global__ __launch_bounds__(256) void sg_conv_head2_kernel(const SgIgemmParams G, const int pad, const int nrg) {
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();
    constexpr int PC = SGF_TW + K - 1, PR = R + K - 1, NV = R * SGF_TW;
    __shared__ float osum[4 * NV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int g = 0;
    for (int gi = 1; gi < G.nprob; ++gi)
        if ((int)blockIdx.x >= G.q[gi].tile0[0]) g = gi;
    const SgLocal P = sg_local(G, g);
    const int tile = blockIdx.x - G.q[g].tile0[0], tiles_x = G.q[g].tile0[1];
    const int TY = R * nrg;                                   // result rows per workgroup
    const int oy0 = (tile / tiles_x) * TY, ox0 = (tile % tiles_x) * SGF_TW;
    const int Ck = P.Ck, wpr = 4 / nrg;                       // waves per row group = channel chunks in flight
    const int rg = wave / wpr, ch0 = wave % wpr;
    const int nch = (Ck + 63) >> 6;
    for (int i = tid; i < 4 * NV; i += 256) osum[i] = 0.f;
    const bool pnorm = P.pro.stats != nullptr;
    const float neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    const int ry0 = oy0 + rg * R;                             // first result row of this wave
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    if (ry0 < P.Hout) {
        for (int ch = ch0; ch < nch; ch += wpr) {
            const int c = ch * 64 + lane;
            const bool cok = c < Ck;
            const int cc = cok ? c : Ck - 1;
            // unconditional loads from clamped addresses, masked afterwards (a load under a branch is waited for at the join)
            float wv[K * K];
#pragma unroll
            for (int t = 0; t < K * K; ++t) {
                const float wl = P.w[G.taps[G.tap0[0] + t].w_off + cc];
                wv[t] = cok ? wl : 0.f;
            }
            float sc = 1.f, sh = 0.f;
            if (pnorm) {
                float mean, rstd;
                sg_mean_rstd(P.pro, Ck, cc, mean, rstd);
                const float gm = P.pro.gamma ? P.pro.gamma[cc] : 1.f;
                const float bt = P.pro.beta ? P.pro.beta[cc] : 0.f;
                sc = gm * rstd;
                sh = bt - mean * sc;
            }
            // buffer loads: per patch column one 32-bit lane offset (fixed for the tile), per patch row one scalar offset
            float xs[PC], xn[PC];
            int voff[PC];
#pragma unroll
            for (int pc = 0; pc < PC; ++pc) voff[pc] = (min(max(ox0 - pad + pc, 0), P.Win - 1) * P.in_ld + cc) << 2;
            auto load_row = [&](int pr, float (&dst)[PC]) {
                const int iy = min(max(ry0 - pad + pr, 0), P.Hin - 1);
                const int soff = __builtin_amdgcn_readfirstlane(iy * P.Win * P.in_ld * 4);
#pragma unroll
                for (int pc = 0; pc < PC; ++pc) dst[pc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff[pc], soff, 0));
            };
            load_row(0, xs);
#pragma unroll
            for (int pr = 0; pr < PR; ++pr) {
                if (pr + 1 < PR) load_row(pr + 1, xn);
                const int iy = ry0 - pad + pr;
                const bool rok = cok && (unsigned)iy < (unsigned)P.Hin;
#pragma unroll
                for (int pc = 0; pc < PC; ++pc) {
                    const int ix = ox0 - pad + pc;
                    const float y = xs[pc] * sc + sh;
                    float a = y > 0.f ? y : y * neg;
                    a = (rok && (unsigned)ix < (unsigned)P.Win) ? a : 0.f;      // zero padding applies AFTER norm + activation
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) {
                            const int r = pr - ky, q = pc - kx;      // result (row, column) this (input pixel, tap) feeds
                            if (r >= 0 && r < R && q >= 0 && q < SGF_TW) acc[r * SGF_TW + q] = fmaf(a, wv[ky * K + kx], acc[r * SGF_TW + q]);
                        }
                }
                if (pr + 1 < PR) {
#pragma unroll
                    for (int pc = 0; pc < PC; ++pc) xs[pc] = xn[pc];
                }
            }
        }
    }
    // fold the 64 lanes: at each step a lane keeps half of its values and adds the partner's copies of them; after log2(NV) steps
    // every lane holds ONE result (value index = the lane bits used so far), summed over the lanes that differ in the remaining bits
    int idx = 0;
    {
        int n = NV;
        float* v = acc;
#pragma unroll
        for (int off = 32; off >= 1 && n > 1; off >>= 1) {
            const bool hi = lane & off;
            n >>= 1;
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const float send = hi ? v[i] : v[i + n];
                const float keep = hi ? v[i + n] : v[i];
                v[i] = keep + __shfl_xor(send, off);
            }
            idx = idx * 2 + (hi ? 1 : 0);
        }
    }
    float tot = acc[0];
    if constexpr (NV == 32) tot += __shfl_xor(tot, 1);        // 32 values on 64 lanes: lanes l and l ^ 1 hold halves of the same result
    // value index: the first step split [0, NV/2) | [NV/2, NV) by its lane bit, the next split each half, ... : idx read as a binary number
    SG_SYNC();      // osum zeroed
    if (NV == 64 || !(lane & 1)) atomicAdd(&osum[rg * NV + idx], tot);
    SG_SYNC();
    for (int i = tid; i < TY * SGF_TW; i += 256) {
        const int oy = oy0 + i / SGF_TW, ox = ox0 + (i % SGF_TW);
        if (oy >= P.Hout || ox >= P.Wout) continue;
        float v = osum[i] + (P.bias ? P.bias[0] : 0.f);
        if (P.out_act == SGAN_ACT_TANH) v = tanhf(v);
        f32x4 o4 = (f32x4){v, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 1; n < 4; ++n) {       // padding channels carry the bias like the generic kernels (zero in practice)
            float b = P.bias ? P.bias[n] : 0.f;
            if (P.out_act == SGAN_ACT_TANH) b = tanhf(b);
            o4[n] = b;
        }
        *reinterpret_cast<f32x4*>(P.out + ((int64_t)oy * P.Wout + ox) * P.out_ld) = o4;
    }
}

// 0: launched; 1: not covered (the caller goes on to sg_conv_head_kernel); called after sg_head_plan() accepted the launch
int sg_launch_head2(SgIgemmParams& P, hipStream_t st) {
    static const int off = getenv("SGAN_NO_HEAD2") ? 1 : 0;
    if (off || P.Ck < 64 || P.nphase != 1 || P.is != 1 || P.os != 1 || P.w_ks != 1) return 1;
    const int nt = P.ntaps[0];
    const int K = nt == 16 ? 4 : nt == 9 ? 3 : 0;
    if (!K) return 1;
    const int dy0 = P.taps[P.tap0[0]].dy, dx0 = P.taps[P.tap0[0]].dx;
    if (dy0 != dx0 || dy0 > 0 || -dy0 >= K) return 1;
    for (int t = 0; t < nt; ++t) {      // row-major k x k taps
        const SgTap& tp = P.taps[P.tap0[0] + t];
        if (tp.dy != dy0 + t / K || tp.dx != dx0 + t % K) return 1;
    }
    const int pad = -dy0;
    const int nch = (P.Ck + 63) / 64;
    const int nrg = nch >= 3 ? 1 : nch == 2 ? 2 : 4;      // row groups per workgroup: waves that have no channel chunk of their own take rows
    static const int r_env = getenv("SGAN_HEAD2_R") ? atoi(getenv("SGAN_HEAD2_R")) : 0;      // tuning knob: 2 or 4 result rows per wave
    const int R = r_env == 4 ? 4 : 2;
    int t = 0;
    for (int g = 0; g < P.nprob; ++g) {
        const int tx = (P.q[g].Wout + SGF_TW - 1) / SGF_TW, ty = (P.q[g].Hout + R * nrg - 1) / (R * nrg);
        P.q[g].tile0[0] = t;
        P.q[g].tile0[1] = tx;
        t += tx * ty;
    }
    if (t == 0) return SGAN_OK;
    sg_prof_begin(st);
    if (K == 4 && R == 2) hipLaunchKernelGGL((sg_conv_head2_kernel<4, 2>), dim3(t), dim3(256), 0, st, P, pad, nrg);
    else if (K == 4) hipLaunchKernelGGL((sg_conv_head2_kernel<4, 4>), dim3(t), dim3(256), 0, st, P, pad, nrg);
    else if (R == 2) hipLaunchKernelGGL((sg_conv_head2_kernel<3, 2>), dim3(t), dim3(256), 0, st, P, pad, nrg);
    else hipLaunchKernelGGL((sg_conv_head2_kernel<3, 4>), dim3(t), dim3(256), 0, st, P, pad, nrg);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_conv_head2_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}
