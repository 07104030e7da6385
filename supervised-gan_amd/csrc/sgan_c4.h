// The two 4-channel forward / backward-data kernels (scatter form for stride-2 transposed layers with a 4-channel result, one-MFMA-per-tap
// form for a 4-channel gathered tensor) and their launchers.  A header because two translation units launch them: sgan_igemm.hip (on
// their own) and sgan_wgrad.hip (side by side with the thin backward-weight kernel in one grid, sg_bwd_thin_pair_kernel).
#pragma once
#include "sgan_igemm.h"

// ------------------------------------------------------------------------------------------
// Scatter form for 4-channel results of stride-2 transposed layers (generator output ConvT 32 -> 2, image gradient of the
// first discriminator conv): every input pixel feeds 16 (phase, tap) outputs, so the gather kernel above reads each
// activation 4 times through L1 and is bound by exactly that.  Here a workgroup takes an 8 x 32 tile of INPUT pixels,
// computes Z[pixel][(phase, tap, co)] = sum_c x[pixel][c] * W[(phase, tap)][co][c] as a dense 256 x 64 x Ck MFMA GEMM (A read
// from global memory once, normalise-on-load applied in registers, B = the whole weight tensor), parks Z in LDS, and
// then every output pixel of the 6 x 30 interior adds its 4 (neighbour, tap) entries: out = bias + sum_t Z[p + d_t][t].
// Requires 4 phases x 4 taps with |dy|, |dx| <= 1 (k4 s2), Ck % 16 == 0, Ck <= 512.
// ------------------------------------------------------------------------------------------
#define SG_SC_TH 8
#define SG_SC_TW 32
#define SG_SC_LDZ 68
// (body: the launch's workgroup index comes in as `bx`, so that another kernel can run it on a slice of its own grid)
__device__ __forceinline__ void sg_conv_scatter4_body(const SgIgemmParams& G, char* smem, const int bx) {
    float* Zs = reinterpret_cast<float*>(smem);                    // [256][SG_SC_LDZ]
    float* pscale = Zs + 256 * SG_SC_LDZ;                          // [Ck]
    float* pshift = pscale + G.Ck;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int g = 0;
    for (int gi = 1; gi < G.nprob; ++gi)
        if (bx >= G.q[gi].tile0[0]) g = gi;
    const SgLocal P = sg_local(G, g);
    const int tloc = bx - G.q[g].tile0[0];
    const int tiles_x = G.q[g].tile0[1];
    const int ty = tloc / tiles_x, tx = tloc - ty * tiles_x;
    const int y0 = ty * (SG_SC_TH - 2), x0 = tx * (SG_SC_TW - 2);     // first interior (phase-grid) pixel of this tile
    const int Ck = P.Ck;
    const bool has_pro = (P.pro.stats != nullptr) || (P.pro.act != SGAN_ACT_NONE);
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    for (int c = tid; c < Ck; c += 256) {
        float sc = 1.f, sh = 0.f;
        if (P.pro.stats) {
            float mean, rstd;
            sg_mean_rstd(P.pro, Ck, c, mean, rstd);
            const float gm = P.pro.gamma ? P.pro.gamma[c] : 1.f;
            const float bt = P.pro.beta ? P.pro.beta[c] : 0.f;
            sc = gm * rstd;
            sh = bt - mean * sc;
        }
        pscale[c] = sc;
        pshift[c] = sh;
    }
    SG_SYNC();

    // ---- Z = X W^T: wave w owns tile rows 2w, 2w+1 (64 pixels = 4 MFMA row blocks), all 64 columns ----
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.w), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;
    int a_off[4], b_off[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = wid * 64 + i * 16 + fr;
        const int iy = y0 - 1 + m / SG_SC_TW, ix = x0 - 1 + m % SG_SC_TW;
        a_ok[i] = ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
        a_off[i] = a_ok[i] ? ((iy * P.Win + ix) * P.in_ld + fq * 4) << 2 : OOB;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {      // column n = j*16 + fr = (phase j, tap fr >> 2, co fr & 3)
        const int t = fr >> 2, co = fr & 3;
        b_off[j] = (G.taps[G.tap0[j] + t].w_off + co * P.w_ns + fq * 4) << 2;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < Ck; kk += 16) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off[i] + kk * 4, 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[j] + kk * 4, 0, 0));
        if (has_pro) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(pscale + kk + fq * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(pshift + kk + fq * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float okf = a_ok[i] ? 1.f : 0.f, okn = okf * pro_neg;     // zero padding applies after norm + activation
                const f32x4 y = a[i] * sc + sh;
                const f32x4 yp = y * okf, yn = y * okn;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[i][e] = fmaxf(yp[e], yn[e]);
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s4], b[j][s4], acc[i][j], 0, 0, 0);
    }
    // acc[i][j][r] = Z[m = wid*64 + i*16 + fq*4 + r][n = j*16 + fr]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Zs[(wid * 64 + i * 16 + fq * 4 + r) * SG_SC_LDZ + j * 16 + fr] = acc[i][j][r];
    SG_SYNC();

    // ---- overlap-add: the (TH-2) x (TW-2) interior in all 4 phases, row-major over output pixels (coalesced 16-byte stores) ----
    constexpr int OH = 2 * (SG_SC_TH - 2), OW = 2 * (SG_SC_TW - 2);
    f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (P.bias) bias = *reinterpret_cast<const f32x4*>(P.bias);
    for (int idx = tid; idx < OH * OW; idx += 256) {
        const int ry = idx / OW, rx = idx - ry * OW;
        const int a_ = ry & 1, b_ = rx & 1, ph = a_ * 2 + b_;        // os = 2: output (2 py + a, 2 px + b)
        const int py = y0 + (ry >> 1), px = x0 + (rx >> 1);
        if (py >= G.q[g].Hp[ph] || px >= G.q[g].Wp[ph]) continue;
        f32x4 v = bias;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int my = py + G.taps[G.tap0[ph] + t].dy - (y0 - 1), mx = px + G.taps[G.tap0[ph] + t].dx - (x0 - 1);
            v += *reinterpret_cast<const f32x4*>(Zs + (my * SG_SC_TW + mx) * SG_SC_LDZ + ph * 16 + t * 4);
        }
        if (P.out_act == SGAN_ACT_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
        }
        const int64_t pix = (int64_t)(py * 2 + a_) * P.Wout + (px * 2 + b_);
        *reinterpret_cast<f32x4*>(P.out + pix * P.out_ld) = v;
    }
}

static __global__ __launch_bounds__(256) void sg_conv_scatter4_kernel(const SgIgemmParams G) {      // (static: this header is in two translation units)
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_conv_scatter4_body(G, smem, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// Forward of a conv whose gathered tensor has 4 stored channels (the first PatchGAN / U-Net conv on the 2- or 3-channel image):
// K = 4 channels x <= 16 taps.  The generic kernel stages two 32-deep k-tiles through LDS for that, with its barriers, prologue
// and ring set-up -- all fixed cost.  Here ONE v_mfma_f32_16x16x4_f32 is one tap: its k dimension is the pixel's four channels.
// A wave keeps ALL weights of its N columns in registers (W[tap][n][k]: 16 taps x N/16 VGPRs), walks 16-pixel row blocks, loads
// one float per lane and tap straight from the image (L1 serves the overlap of neighbouring taps and pixels) and stores the tile
// from the accumulators.  No LDS, no barrier.  Exact fp32, same products and tap order as the generic kernel.
// ------------------------------------------------------------------------------------------
// EPI: the full epilogue of the generic kernel (backward-data: times act'(norm(x)) of the forward tensor, the two norm-backward sums,
// accumulate; forward: the statistics of the result).  Those launches tile N in 16 NB columns (grid.y) -- the image gradient that
// reaches the logits head (1 -> 256 channels) and the generator's last layer (2 -> 32) are K = 64 problems of the same shape.
template <int NB, int RB, bool EPI>     // 16-column blocks per workgroup; 16-pixel row blocks per wave (64 RB result pixels per workgroup)
__device__ __forceinline__ void sg_conv_c4_body(const SgIgemmParams& G, const int bx, const int by) {
    __shared__ int4 ttab[16];
    __shared__ __attribute__((aligned(16))) float Ws[16 * NB * 16 * 4];      // [tap][n][4 channels], zero for taps / columns that do not exist
    __shared__ __attribute__((aligned(16))) float cf[EPI ? 4 * NB * 16 : 4];  // EPI: mean | rstd | gamma | beta of the forward tensor's norm
    __shared__ double red[EPI ? 2 * NB * 16 : 2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int g, phz, mtile;
    sg_decode_tile(G, bx, g, phz, mtile);
    const int n0 = EPI ? by * NB * 16 : 0;
    const SgLocal P = sg_local(G, g);
    const int Hp = G.q[g].Hp[phz], Wp = G.q[g].Wp[phz];
    const int M = Hp * Wp;
    const int ntaps = G.ntaps[phz], t0 = G.tap0[phz];
    const int oa = G.oa[phz], ob = G.ob[phz];
    constexpr int OOB = (int)0x80000000u;
    constexpr int NW = NB * 16;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    if (tid < 16) {
        const SgTap tp = G.taps[t0 + (tid < ntaps ? tid : 0)];
        ttab[tid] = make_int4((int)tp.dy, (int)tp.dx, tp.w_off, 0);
    }
    SG_SYNC();
    for (int e = tid; e < 16 * NW; e += 256) {      // one (tap, column) per item: its four channel weights
        const int t = e / NW, n = n0 + e - t * NW;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t < ntaps && n < P.N) v = *reinterpret_cast<const f32x4*>(P.w + ttab[t].z + n * P.w_ns);
        *reinterpret_cast<f32x4*>(Ws + e * 4) = v;
    }
    const bool dact = EPI && P.xref != nullptr, want_stats = EPI && P.stats != nullptr;
    const bool xnorm = dact && P.xn.stats != nullptr;
    if constexpr (EPI) {
        if (tid < NW) {
            const int n = n0 + tid;
            float mean = 0.f, rstd = 1.f, gm = 1.f, bt = 0.f;
            if (xnorm && n < P.N) {
                sg_mean_rstd(P.xn, P.N, n, mean, rstd);
                gm = P.xn.gamma ? P.xn.gamma[n] : 1.f;
                bt = P.xn.beta ? P.xn.beta[n] : 0.f;
            }
            cf[tid] = mean; cf[NW + tid] = rstd; cf[2 * NW + tid] = gm; cf[3 * NW + tid] = bt;
        }
        if (tid < 2 * NW) red[tid] = 0.0;
    }
    // The MFMA runs "transposed": rows = result channels (weights as the A operand), columns = the 16 pixels of a row block (the
    // gathered image as B).  k of one MFMA = four TAPS of one channel: lane (fr, fq) loads the whole 16-byte pixel of tap 4 T + fq
    // once and feeds its four channels to four MFMAs; the accumulator of lane (fr, fq) is then result pixel fr, channels
    // 16 j + 4 fq .. + 3 -- one 16-byte store.  All gathers of the wave are in flight before the first MFMA.
    const int m_wave = mtile * (64 * RB) + wid * (16 * RB);
    f32x4 xv[RB][4];
    int opix[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int m = m_wave + rb * 16 + fr;
        const bool valid = m < M;
        const int py = m / Wp, px = m - py * Wp;
        opix[rb] = valid ? (py * P.os + oa) * P.Wout + (px * P.os + ob) : -1;
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const int4 tp = ttab[4 * T + fq];
            const int iy = py * P.is + tp.x, ix = px * P.is + tp.y;
            const bool ok = valid & (4 * T + fq < ntaps) & ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
            xv[rb][T] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? ((iy * P.Win + ix) * P.in_ld) << 2 : OOB, 0, 0));
        }
    }
    SG_SYNC();      // weights staged
    f32x4 wreg[4][NB];    // [T][j]: the four channel weights of (tap 4 T + fq, column 16 j + fr)
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int j = 0; j < NB; ++j) wreg[T][j] = *reinterpret_cast<const f32x4*>(Ws + ((4 * T + fq) * NW + j * 16 + fr) * 4);
    f32x4 bias4[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        bias4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (P.bias)
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[j][r] = (n0 + j * 16 + 4 * fq + r < P.N) ? P.bias[n0 + j * 16 + 4 * fq + r] : 0.f;
    }
    const float xn_neg = P.xn.act == SGAN_ACT_NONE ? 1.f : (P.xn.act == SGAN_ACT_RELU ? 0.f : P.xn.slope);
    double s1[EPI ? NB : 1][4], s2[EPI ? NB : 1][4];
    if constexpr (EPI) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[j][r] = 0.0; s2[j][r] = 0.0; }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        f32x4 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[T][j][c], xv[rb][T][c], acc[j], 0, 0, 0);
        if (opix[rb] < 0) continue;
        float* o = P.out + (int64_t)opix[rb] * P.out_ld;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int nl = j * 16 + 4 * fq, n = n0 + nl;
            if (n >= P.N) continue;      // N is a multiple of 4: whole 16-byte groups
            f32x4 v = acc[j] + bias4[j];
            if constexpr (EPI) {
                if (dact) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(P.xref + (int64_t)opix[rb] * P.xref_ld + n);
                    const f32x4 mean = *reinterpret_cast<const f32x4*>(cf + nl), rstd = *reinterpret_cast<const f32x4*>(cf + NW + nl);
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(cf + 2 * NW + nl), bt = *reinterpret_cast<const f32x4*>(cf + 3 * NW + nl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xhat = (x[r] - mean[r]) * rstd[r];
                        const float y = xnorm ? (gm[r] * xhat + bt[r]) : x[r];
                        v[r] *= (y > 0.f ? 1.f : xn_neg);
                        s1[j][r] += (double)v[r];
                        s2[j][r] += (double)(v[r] * xhat);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[j][r] += (double)v[r];
                        s2[j][r] += (double)v[r] * (double)v[r];
                    }
                }
            }
            if (!dact && P.out_act == SGAN_ACT_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
            }
            if constexpr (EPI) {
                if (P.accum) v += *reinterpret_cast<const f32x4*>(o + n);
            }
            *reinterpret_cast<f32x4*>(o + n) = v;
        }
    }
    if constexpr (EPI) {
        if (want_stats) {      // lanes of one fq hold the same channels for 16 pixels: fold them, then the waves meet in LDS
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double a1 = s1[j][r], a2 = s2[j][r];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
                    if (fr == 0) {
                        atomicAdd(&red[j * 16 + 4 * fq + r], a1);
                        atomicAdd(&red[NW + j * 16 + 4 * fq + r], a2);
                    }
                }
            SG_SYNC();
            if (tid < NW && n0 + tid < P.N) {
                double* st = sg_stat_replica(P.stats, P.stats_rep, bx);
                atomicAdd(&st[n0 + tid], red[tid]);
                atomicAdd(&st[P.stats_sq + n0 + tid], red[NW + tid]);
            }
        }
    }
}

template <int NB, int RB, bool EPI>
__global__ __launch_bounds__(256) void sg_conv_c4_kernel(const SgIgemmParams G) {
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    sg_conv_c4_body<NB, RB, EPI>(G, (int)blockIdx.x, (int)blockIdx.y);
}

// 4 gathered channels, k-contiguous weights, <= 16 taps per phase, N > 16.  Plain launches (bias / tanh only): N <= 64 in one
// workgroup column.  Launches with the full epilogue (forward-tensor derivative, statistics, accumulate): any N, tiled by 32.
static bool sg_c4_needs_epi(const SgIgemmParams& P) {
    for (int g = 0; g < P.nprob; ++g)
        if (P.q[g].stats || P.q[g].xref || P.q[g].accum) return true;
    return false;
}

static bool sg_use_c4(const SgIgemmParams& P) {
    static const int off = getenv("SGAN_NO_C4") ? atoi(getenv("SGAN_NO_C4")) : 0;      // 1: never; 2: not for the full-epilogue launches
    if (off == 1 || P.Ck != 4 || P.w_ks != 1 || (P.w_ns & 3) || P.N <= 16 || (P.N & 3) || P.pro_act != SGAN_ACT_NONE) return false;
    for (int ph = 0; ph < P.nphase; ++ph)
        if (P.ntaps[ph] > 16) return false;
    const bool epi = sg_c4_needs_epi(P);
    if (epi ? off == 2 : P.N > 64) return false;
    for (int g = 0; g < P.nprob; ++g) {
        const SgProb& Q = P.q[g];
        if (Q.pro_stats || (Q.out_ld & 3) || (Q.in_ld & 3) || (Q.xref && (Q.xref_ld & 3))) return false;
        if (epi && (!Q.xref != !P.q[0].xref || !Q.stats != !P.q[0].stats)) return false;      // one epilogue shape per launch
    }
    return true;
}

template <int RB>
static void sg_launch_c4_rb(SgIgemmParams& P, int tiles, hipStream_t st) {
    if (sg_c4_needs_epi(P)) {
        hipLaunchKernelGGL((sg_conv_c4_kernel<2, RB, true>), dim3(tiles, (P.N + 31) / 32), dim3(256), 0, st, P);
        return;
    }
    if (P.N <= 32) hipLaunchKernelGGL((sg_conv_c4_kernel<2, RB, false>), dim3(tiles), dim3(256), 0, st, P);
    else if (P.N <= 48) hipLaunchKernelGGL((sg_conv_c4_kernel<3, RB, false>), dim3(tiles), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((sg_conv_c4_kernel<4, RB, false>), dim3(tiles), dim3(256), 0, st, P);
}

// row blocks per wave of a c4 launch (see sg_launch_c4)
static int sg_c4_pick_rb(SgIgemmParams& P) {
    static const int rb = getenv("SGAN_C4_RB") ? atoi(getenv("SGAN_C4_RB")) : 0;
    int RB = (rb == 1 || rb == 2 || rb == 4) ? rb : 4;
    if (!rb && sg_fill_tiles(P, 64 * 4) <= 400) RB = 2;
    return RB;
}

static int sg_launch_c4(SgIgemmParams& P, hipStream_t st) {
    static const int rb = getenv("SGAN_C4_RB") ? atoi(getenv("SGAN_C4_RB")) : 0;      // tuning knob: 1, 2 or 4 row blocks per wave (0: by grid size)
    // four row blocks per wave unless that leaves the chip with under ~1.5 workgroups per CU (three-problem first PatchGAN conv: 340
    // workgroups, 11.4 us; with two row blocks 680 workgroups, 9.8 us; the six-problem launch keeps four: 16.7 us either way)
    int RB = (rb == 1 || rb == 2 || rb == 4) ? rb : 4;
    if (!rb && sg_fill_tiles(P, 64 * 4) <= 400) RB = 2;
    const int tiles = sg_fill_tiles(P, 64 * RB);
    if (tiles == 0) return SGAN_OK;
    sg_prof_begin(st);
    if (RB == 1) sg_launch_c4_rb<1>(P, tiles, st);
    else if (RB == 2) sg_launch_c4_rb<2>(P, tiles, st);
    else sg_launch_c4_rb<4>(P, tiles, st);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_conv_c4_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

static bool sg_use_scatter4(const SgIgemmParams& P) {
    static const int off = getenv("SGAN_NO_SCATTER4") ? 1 : 0;
    if (off || P.N != 4 || P.nphase != 4 || P.os != 2 || P.is != 1 || P.w_ks != 1 || (P.Ck & 15) || P.Ck > 512 || P.Ck < 16) return false;
    for (int ph = 0; ph < 4; ++ph) {
        if (P.ntaps[ph] != 4 || P.oa[ph] != (ph >> 1) || P.ob[ph] != (ph & 1)) return false;
        for (int t = 0; t < 4; ++t)
            if (P.taps[P.tap0[ph] + t].dy < -1 || P.taps[P.tap0[ph] + t].dy > 1 || P.taps[P.tap0[ph] + t].dx < -1 || P.taps[P.tap0[ph] + t].dx > 1) return false;
    }
    return true;
}

// tile table of a scatter4 launch (first workgroup and tiles per row of every problem); returns the workgroup count
static int sg_scatter4_plan(SgIgemmParams& P, size_t* lds) {
    int t = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int hp = 0, wp = 0;
        for (int ph = 0; ph < 4; ++ph) { hp = max(hp, P.q[g].Hp[ph]); wp = max(wp, P.q[g].Wp[ph]); }
        const int tx = (wp + SG_SC_TW - 3) / (SG_SC_TW - 2), ty = (hp + SG_SC_TH - 3) / (SG_SC_TH - 2);
        P.q[g].tile0[0] = t;
        P.q[g].tile0[1] = tx;
        t += tx * ty;
    }
    *lds = (size_t)256 * SG_SC_LDZ * 4 + (size_t)2 * P.Ck * 4;
    return t;
}

static int sg_launch_scatter4(SgIgemmParams& P, hipStream_t st) {
    size_t lds;
    const int t = sg_scatter4_plan(P, &lds);
    if (t == 0) return SGAN_OK;
    sg_prof_begin(st);
    hipLaunchKernelGGL(sg_conv_scatter4_kernel, dim3(t), dim3(256), lds, st, P);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_conv_scatter4_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

