// Implicit-GEMM Conv2d / ConvTranspose2d forward and backward-data for gfx950 (CDNA4), fp32.
//
// One kernel template covers the four conv-like ops of the hot path (see SgPhase in
// sgan_common.h): out[m][n] = sum_k A[m][k] * B[k][n], where m runs over the pixels of an output
// phase, k = (tap, channel) and A is gathered on the fly from the NHWC source -- no im2col, no
// zero insertion for the transposed conv (4 sub-pixel phases with 2x2 taps each).
//
//  * arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD = the fp32 matrix peak);
//  * a 256-thread workgroup (4 wave64) owns a BM x BN tile; each wave MB x NB 16x16 accumulators;
//  * operands are staged HBM -> VGPR -> LDS as [row][32 k] 128-byte rows, 16-byte chunks
//    XOR-swizzled with (row>>1)&7 so both the ds_write_b128 and the ds_read_b128 fragment reads are
//    bank-conflict free; LDS is double buffered and the next tile's global loads are issued before
//    the MFMA block of the current one (one barrier per k-tile);
//  * "normalise-on-load": InstanceNorm / BatchNorm(+affine) and ReLU / LeakyReLU of the PRODUCER
//    layer are applied to A while it is staged (per-channel scale/shift derived in-kernel from the
//    producer's (sum, sumsq) statistics), so no normalised tensor is ever written to HBM;
//  * epilogue: +bias, per-channel (sum, sumsq) statistics of the result for the NEXT layer's
//    normalisation (wave shuffle -> LDS -> one fp64 atomic per channel per workgroup), tanh;
//  * backward-data epilogue: multiply by act'(norm(x)) of the forward tensor and accumulate the two
//    norm-backward sums (sum dY, sum dY*xhat).
//
// Reference ops replaced: nn.Conv2d / nn.ConvTranspose2d forward + convolution_backward(input)
// as instantiated at models/networks.py:502-529 (FCGANGenerator) and :815-835 (NLayerDiscriminator).
#include "sgan_common.h"

struct SgIgemmParams {
    const float* in;    // gathered tensor
    float* out;         // result tensor
    const float* w;     // master weight
    const float* bias;  // [N] or null
    const float* xref;  // dact epilogue: forward tensor at the output positions, or null
    double* stats;      // [2N]: fwd (sum, sumsq) of the result, or bwd sums (s1, s2); or null
    int32_t Hin, Win, Ck, in_ld;    // gathered tensor geometry, Ck = its channels (GEMM-K channels)
    int32_t Hout, Wout, N, out_ld;  // result tensor geometry, N = its channels
    int32_t xref_ld;
    int32_t is, os;
    int32_t w_ns, w_ks;  // element strides of B[k-channel][n] inside a tap slab
    int32_t out_act;
    int32_t nphase;
    SgNorm pro;  // prologue on the gathered tensor
    SgNorm xn;   // how the forward consumer read xref (dact)
    SgPhase phase[SGAN_MAX_PHASES];
};

__device__ __forceinline__ int sg_swz(int row, int kslot) { return (kslot ^ ((row >> 1) & 7)) << 2; }

// BKC: B operand is k-contiguous in memory (forward: W[tap][n][k]); otherwise n-contiguous (backward-data)
template <int BM, int BN, int WGM, int WGN, bool BKC>
__global__ __launch_bounds__(256) void sg_igemm_kernel(const SgIgemmParams P) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MB = WTM / 16, NB = WTN / 16;
    constexpr int A_IT = BM * 8 / 256;
    constexpr int B_IT = (BN * 8 + 255) / 256;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(BM * 8 % 256 == 0, "A tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);  // [2][BM*32]
    float* Bs = As + 2 * BM * 32;                // [2][BN*32]
    float* red = Bs + 2 * BN * 32;               // [2*BN]
    int* tdy = reinterpret_cast<int*>(red + 2 * BN);  // [16]
    int* tdx = tdy + SGAN_MAX_TAPS;
    int* two = tdx + SGAN_MAX_TAPS;
    float* pscale = reinterpret_cast<float*>(two + SGAN_MAX_TAPS);  // [Ck]
    float* pshift = pscale + P.Ck;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int phz = blockIdx.z;
    const int Hp = P.phase[phz].Hp, Wp = P.phase[phz].Wp;
    const int M = Hp * Wp;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M) return;  // phases of an odd-sized grid differ in size (block-uniform exit)
    const int oa = P.phase[phz].oa, ob = P.phase[phz].ob;
    const int ktot = P.phase[phz].ktot;
    const int Ck = P.Ck, N = P.N;
    const bool has_pro = (P.pro.stats != nullptr) || (P.pro.act != SGAN_ACT_NONE);

    // ---- one-time setup: tap table, prologue scale/shift, reduction scratch ----
    if (tid < SGAN_MAX_TAPS) {
        const bool v = tid < P.phase[phz].ntaps;
        tdy[tid] = v ? (int)P.phase[phz].taps[tid].dy : 0;
        tdx[tid] = v ? (int)P.phase[phz].taps[tid].dx : 0;
        two[tid] = v ? P.phase[phz].taps[tid].w_off : 0;
    }
    for (int i = tid; i < 2 * BN; i += 256) red[i] = 0.f;
    if (has_pro) {
        for (int c = tid; c < Ck; c += 256) {
            float sc = 1.f, sh = 0.f;
            if (P.pro.stats) {
                float mean, rstd;
                sg_mean_rstd(P.pro, Ck, c, mean, rstd);
                const float g = P.pro.gamma ? P.pro.gamma[c] : 1.f;
                const float b = P.pro.beta ? P.pro.beta[c] : 0.f;
                sc = g * rstd;
                sh = b - mean * sc;
            }
            pscale[c] = sc;
            pshift[c] = sh;
        }
    }

    // ---- per-thread staging state ----
    // A: this thread owns kslot a_ks of rows (tid>>3) + 32*it.  Its (tap, channel) position walks forward
    // by 32 k per tile: 32 = adv_tap * Ck + adv_c, so one conditional wrap per tile and no division.
    const int adv_tap = 32 / Ck, adv_c = 32 - adv_tap * Ck;
    const int ntaps = P.phase[phz].ntaps;
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    int a_iy[A_IT], a_ix[A_IT], a_dst[A_IT];
    bool a_rowok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int e = tid + it * 256;
        const int row = e >> 3, ks = e & 7;
        const int m = m0 + row;
        a_rowok[it] = m < M;
        const int py = m / Wp, px = m - py * Wp;
        a_iy[it] = py * P.is;
        a_ix[it] = px * P.is;
        a_dst[it] = row * 32 + sg_swz(row, ks);
    }
    const int a_ks = tid & 7;  // same for every it (256 % 8 == 0)
    int a_tap = (a_ks * 4) / Ck;
    int a_c = a_ks * 4 - a_tap * Ck;

    f32x4 a_reg[A_IT];
    bool a_ok[A_IT];
    int a_cs = 0;  // channel of the staged A registers (for the prologue transform)
    f32x4 b_reg[B_IT];
    bool b_ok[B_IT];
    constexpr bool b_kcontig = BKC;
    // B, n-contiguous form (backward-data): element (k = e / NQ, n4 = e % NQ); its own (tap, channel) walk
    constexpr int NQ = BN / 4;
    int b_tap[B_IT], b_c[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int k = (tid + it * 256) / NQ;
        b_tap[it] = k / Ck;
        b_c[it] = k - b_tap[it] * Ck;
    }

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // tap table visible

    // All loads are unconditional (invalid elements read offset 0 of the tensor and are zeroed when the
    // registers are written to LDS), so hipcc issues them back to back and waits only at first use.
    auto load_tile = [&]() {
        {
            const bool kok = a_tap < ntaps;
            const int tap = kok ? a_tap : 0;
            const int dy = tdy[tap], dx = tdx[tap];
            a_cs = kok ? a_c : 0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int iy = a_iy[it] + dy, ix = a_ix[it] + dx;
                const bool ok = a_rowok[it] && kok && (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
                const int64_t off = ok ? ((int64_t)iy * P.Win + ix) * P.in_ld + a_c : 0;
                a_reg[it] = *reinterpret_cast<const f32x4*>(P.in + off);
                a_ok[it] = ok;
            }
            if constexpr (b_kcontig) {
                const int wtap = two[tap];
#pragma unroll
                for (int it = 0; it < B_IT; ++it) {
                    const int e = tid + it * 256;
                    const int n = e >> 3;
                    const bool ok = (B_IT * 256 == BN * 8 || e < BN * 8) && kok && n0 + n < N;
                    const int64_t off = ok ? wtap + (int64_t)(n0 + n) * P.w_ns + a_c : 0;
                    b_reg[it] = *reinterpret_cast<const f32x4*>(P.w + off);
                    b_ok[it] = ok;
                }
            }
            a_tap += adv_tap;
            a_c += adv_c;
            if (a_c >= Ck) { a_c -= Ck; ++a_tap; }
        }
        if constexpr (!b_kcontig) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int e = tid + it * 256;
                const int k = e / NQ, n4 = e % NQ;
                const bool tok = b_tap[it] < ntaps;
                const int wtap = two[tok ? b_tap[it] : 0];
                const bool ok = (B_IT * 256 <= 32 * NQ || k < 32) && tok && n0 + n4 * 4 < N;
                const int64_t off = ok ? wtap + (int64_t)b_c[it] * P.w_ks + n0 + n4 * 4 : 0;
                b_reg[it] = *reinterpret_cast<const f32x4*>(P.w + off);
                b_ok[it] = ok;
                b_tap[it] += adv_tap;
                b_c[it] += adv_c;
                if (b_c[it] >= Ck) { b_c[it] -= Ck; ++b_tap[it]; }
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* Ab = As + buf * BM * 32;
        float* Bb = Bs + buf * BN * 32;
        f32x4 sc = (f32x4){1.f, 1.f, 1.f, 1.f}, sh = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_pro) {
            sc = *reinterpret_cast<const f32x4*>(pscale + a_cs);
            sh = *reinterpret_cast<const f32x4*>(pshift + a_cs);
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            f32x4 v = a_reg[it];
            if (has_pro) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y = v[j] * sc[j] + sh[j];
                    v[j] = y > 0.f ? y : y * pro_neg;
                }
            }
            if (!a_ok[it]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(Ab + a_dst[it]) = v;
        }
        if constexpr (b_kcontig) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int e = tid + it * 256;
                const int n = e >> 3, ks = e & 7;
                f32x4 v = b_reg[it];
                if (!b_ok[it]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (B_IT * 256 == BN * 8 || e < BN * 8) *reinterpret_cast<f32x4*>(Bb + n * 32 + sg_swz(n, ks)) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int e = tid + it * 256;
                const int k = e / NQ, n4 = e % NQ;
                f32x4 v = b_reg[it];
                if (!b_ok[it]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (B_IT * 256 <= 32 * NQ || k < 32) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = n4 * 4 + j;
                        Bb[row * 32 + sg_swz(row, k >> 2) + (k & 3)] = v[j];
                    }
                }
            }
        }
    };

    const int nkt = (ktot + 31) >> 5;
    const int fr = lane & 15, fq = lane >> 4;

    load_tile();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        store_tile(buf);
        __syncthreads();
        if (kt + 1 < nkt) load_tile();
        const float* Ab = As + buf * BM * 32;
        const float* Bb = Bs + buf * BN * 32;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            f32x4 af[MB], bf[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int row = wm * WTM + i * 16 + fr;
                af[i] = *reinterpret_cast<const f32x4*>(Ab + row * 32 + sg_swz(row, kh * 4 + fq));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int row = wn * WTN + j * 16 + fr;
                bf[j] = *reinterpret_cast<const f32x4*>(Bb + row * 32 + sg_swz(row, kh * 4 + fq));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        // no second barrier: the next iteration writes the OTHER buffer, and the barrier at its top
        // orders those writes against this iteration's reads of `buf` two iterations later.
    }

    // ---- epilogue ----
    const bool dact = P.xref != nullptr;
    const bool want_stats = P.stats != nullptr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int nl = wn * WTN + j * 16 + fr;
        const int n = n0 + nl;
        const bool nvalid = n < N;
        const float bias_v = (P.bias && nvalid) ? P.bias[n] : 0.f;
        float x_mean = 0.f, x_rstd = 1.f, x_g = 1.f, x_b = 0.f;
        const bool xnorm = dact && P.xn.stats != nullptr;
        if (xnorm && nvalid) {
            sg_mean_rstd(P.xn, N, n, x_mean, x_rstd);
            x_g = P.xn.gamma ? P.xn.gamma[n] : 1.f;
            x_b = P.xn.beta ? P.xn.beta[n] : 0.f;
        }
        const float xn_neg = P.xn.act == SGAN_ACT_NONE ? 1.f : (P.xn.act == SGAN_ACT_RELU ? 0.f : P.xn.slope);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WTM + i * 16 + fq * 4 + r;
                if (m < M && nvalid) {
                    const int py = m / Wp, px = m - py * Wp;
                    const int64_t pix = (int64_t)(py * P.os + oa) * P.Wout + (px * P.os + ob);
                    float v = acc[i][j][r] + bias_v;
                    if (dact) {
                        const float x = P.xref[pix * P.xref_ld + n];
                        const float xhat = (x - x_mean) * x_rstd;
                        const float y = xnorm ? (x_g * xhat + x_b) : x;
                        v *= (y > 0.f ? 1.f : xn_neg);
                        s1 += v;
                        s2 += v * xhat;
                    } else {
                        s1 += v;
                        s2 += v * v;
                        if (P.out_act == SGAN_ACT_TANH) v = tanhf(v);
                    }
                    P.out[pix * P.out_ld + n] = v;
                }
            }
        }
        if (want_stats) {
            s1 += __shfl_xor(s1, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16);
            s2 += __shfl_xor(s2, 32);
            if (fq == 0 && nvalid) {
                atomicAdd(&red[nl], s1);
                atomicAdd(&red[BN + nl], s2);
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            atomicAdd(&P.stats[n0 + tid], (double)red[tid]);
            atomicAdd(&P.stats[N + n0 + tid], (double)red[BN + tid]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline int sg_cdiv(int a, int b) { return (a + b - 1) / b; }

int sg_build_phases(const sgan_conv_desc* d, bool dgrad, SgPhase* ph, int* nphase, int* is, int* os) {
    const int k = d->k, s = d->stride, p = d->pad;
    if (k * k > SGAN_MAX_TAPS) return sgan_fail(SGAN_ERR_UNSUPPORTED, "kernel size %d not supported", k);
    if (d->kind == SGAN_CONV) {
        if (d->Hout != (d->Hin + 2 * p - k) / s + 1 || d->Wout != (d->Win + 2 * p - k) / s + 1)
            return sgan_fail(SGAN_ERR_INVALID, "conv geometry mismatch");
    } else if (d->kind == SGAN_CONVT) {
        if (d->Hout != (d->Hin - 1) * s - 2 * p + k || d->Wout != (d->Win - 1) * s - 2 * p + k)
            return sgan_fail(SGAN_ERR_INVALID, "convT geometry mismatch");
    } else {
        return sgan_fail(SGAN_ERR_INVALID, "bad conv kind %d", d->kind);
    }
    const int slab = d->Cout * d->Cin;
    // conv-form (gather at grid*stride + k - pad): Conv fwd/wgrad, ConvT dgrad
    const bool conv_form = (d->kind == SGAN_CONV) != dgrad;
    // the tensor the phases tile:  fwd/wgrad -> forward output ; dgrad -> forward input
    const int Hg = dgrad ? d->Hin : d->Hout, Wg = dgrad ? d->Win : d->Wout;
    // channels of the gathered tensor: fwd -> Cin ; dgrad -> Cout
    const int Ck = dgrad ? d->Cout : d->Cin;
    if (conv_form) {
        *nphase = 1;
        *is = s;
        *os = 1;
        SgPhase& q = ph[0];
        q.oa = q.ob = 0;
        q.Hp = Hg;
        q.Wp = Wg;
        q.ntaps = k * k;
        q.ktot = q.ntaps * Ck;
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                SgTap& t = q.taps[ky * k + kx];
                t.dy = (int16_t)(ky - p);
                t.dx = (int16_t)(kx - p);
                t.w_off = (ky * k + kx) * slab;
            }
    } else {
        if (s * s > SGAN_MAX_PHASES) return sgan_fail(SGAN_ERR_UNSUPPORTED, "stride %d not supported", s);
        *nphase = s * s;
        *is = 1;
        *os = s;
        for (int a = 0; a < s; ++a)
            for (int b = 0; b < s; ++b) {
                SgPhase& q = ph[a * s + b];
                q.oa = a;
                q.ob = b;
                q.Hp = Hg > a ? sg_cdiv(Hg - a, s) : 0;
                q.Wp = Wg > b ? sg_cdiv(Wg - b, s) : 0;
                int nt = 0;
                for (int ky = 0; ky < k; ++ky) {
                    const int ry = a + p - ky;
                    if (((ry % s) + s) % s != 0) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        const int rx = b + p - kx;
                        if (((rx % s) + s) % s != 0) continue;
                        SgTap& t = q.taps[nt++];
                        t.dy = (int16_t)(ry / s);
                        t.dx = (int16_t)(rx / s);
                        t.w_off = (ky * k + kx) * slab;
                    }
                }
                q.ntaps = nt;
                q.ktot = nt * Ck;
            }
    }
    return SGAN_OK;
}

template <int BM, int BN, int WGM, int WGN>
static int sg_launch_igemm(const SgIgemmParams& P, hipStream_t st) {
    const bool bkc = P.w_ks == 1;
    int maxM = 0;
    for (int i = 0; i < P.nphase; ++i) {
        const int M = P.phase[i].Hp * P.phase[i].Wp;
        if (M > maxM) maxM = M;
    }
    if (maxM == 0) return SGAN_OK;
    dim3 grid(sg_cdiv(maxM, BM), sg_cdiv(P.N, BN), P.nphase);
    const size_t lds = (size_t)(2 * BM * 32 + 2 * BN * 32 + 2 * BN) * 4 + 3 * SGAN_MAX_TAPS * 4 + (size_t)2 * P.Ck * 4;
    if (lds > 160 * 1024) return sgan_fail(SGAN_ERR_UNSUPPORTED, "LDS %zu too large", lds);
    if (bkc) hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, false>), grid, dim3(256), lds, st, P);
    SGAN_LAUNCH_CHECK();
    if (bkc)
        g_sgan_last_kernel = BM == 64 ? "sg_igemm_kernel<64,64,2,2,true>" : BN == 64 ? "sg_igemm_kernel<128,64,2,2,true>"
                             : BN == 32 ? "sg_igemm_kernel<128,32,4,1,true>" : "sg_igemm_kernel<128,16,4,1,true>";
    else
        g_sgan_last_kernel = BM == 64 ? "sg_igemm_kernel<64,64,2,2,false>" : BN == 64 ? "sg_igemm_kernel<128,64,2,2,false>"
                             : BN == 32 ? "sg_igemm_kernel<128,32,4,1,false>" : "sg_igemm_kernel<128,16,4,1,false>";
    return SGAN_OK;
}

static int sg_dispatch_igemm(const SgIgemmParams& P, hipStream_t st) {
    int maxM = 0;
    for (int i = 0; i < P.nphase; ++i) {
        const int M = P.phase[i].Hp * P.phase[i].Wp;
        if (M > maxM) maxM = M;
    }
    if (P.N <= 16) return sg_launch_igemm<128, 16, 4, 1>(P, st);
    if (P.N <= 32) return sg_launch_igemm<128, 32, 4, 1>(P, st);
    // 128x64 tiles only when they still fill the chip (256 CUs, 2 workgroups each)
    const long blocks128 = (long)sg_cdiv(maxM, 128) * sg_cdiv(P.N, 64) * P.nphase;
    if (blocks128 >= 512) return sg_launch_igemm<128, 64, 2, 2>(P, st);
    return sg_launch_igemm<64, 64, 2, 2>(P, st);
}

static int sg_check_common(const sgan_conv_desc* d) {
    if (!d) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    if ((d->Cin & 3) || (d->Cout & 3)) return sgan_fail(SGAN_ERR_INVALID, "stored channels must be multiples of 4 (Cin %d Cout %d)", d->Cin, d->Cout);
    if (d->Hin <= 0 || d->Win <= 0 || d->Hout <= 0 || d->Wout <= 0) return sgan_fail(SGAN_ERR_INVALID, "empty tensor");
    return SGAN_OK;
}

extern "C" int sgan_conv_fwd(const sgan_conv_desc* d, const float* in, int32_t in_ld, const sgan_norm_desc* in_norm,
                             const float* w, const float* bias, float* out, int32_t out_ld, int32_t out_act,
                             double* out_stats, void* stream) {
    int rc = sg_check_common(d);
    if (rc) return rc;
    SGAN_CHECK(in && w && out, "null tensor");
    SGAN_CHECK(in_ld >= d->Cin && out_ld >= d->Cout && (in_ld & 3) == 0, "bad leading dims");
    SGAN_CHECK(out_act == SGAN_ACT_NONE || out_act == SGAN_ACT_TANH, "out_act must be none or tanh");
    SgIgemmParams P;
    memset(&P, 0, sizeof(P));
    rc = sg_build_phases(d, false, P.phase, &P.nphase, &P.is, &P.os);
    if (rc) return rc;
    P.in = in; P.out = out; P.w = w; P.bias = bias; P.xref = nullptr; P.stats = out_stats;
    P.Hin = d->Hin; P.Win = d->Win; P.Ck = d->Cin; P.in_ld = in_ld;
    P.Hout = d->Hout; P.Wout = d->Wout; P.N = d->Cout; P.out_ld = out_ld; P.xref_ld = 0;
    P.w_ns = d->Cin; P.w_ks = 1;  // B[k=ci][n=co] = W[tap][co][ci]
    P.out_act = out_act;
    P.pro = sg_norm_from(in_norm);
    P.xn = sg_norm_from(nullptr);
    return sg_dispatch_igemm(P, (hipStream_t)stream);
}

extern "C" int sgan_conv_dgrad(const sgan_conv_desc* d, const float* dout, int32_t dout_ld, const float* w,
                               float* din, int32_t din_ld, const float* x, int32_t x_ld, const sgan_norm_desc* x_norm,
                               double* bwd_sums, void* stream) {
    int rc = sg_check_common(d);
    if (rc) return rc;
    SGAN_CHECK(dout && w && din, "null tensor");
    SGAN_CHECK(dout_ld >= d->Cout && din_ld >= d->Cin && (dout_ld & 3) == 0, "bad leading dims");
    SGAN_CHECK(!x || x_ld >= d->Cin, "bad x_ld");
    SGAN_CHECK(!(bwd_sums && !x), "bwd_sums needs x");
    SgIgemmParams P;
    memset(&P, 0, sizeof(P));
    rc = sg_build_phases(d, true, P.phase, &P.nphase, &P.is, &P.os);
    if (rc) return rc;
    P.in = dout; P.out = din; P.w = w; P.bias = nullptr; P.xref = x; P.stats = bwd_sums;
    P.Hin = d->Hout; P.Win = d->Wout; P.Ck = d->Cout; P.in_ld = dout_ld;
    P.Hout = d->Hin; P.Wout = d->Win; P.N = d->Cin; P.out_ld = din_ld; P.xref_ld = x_ld;
    P.w_ns = 1; P.w_ks = d->Cin;  // B[k=co][n=ci] = W[tap][co][ci]
    P.out_act = SGAN_ACT_NONE;
    P.pro = sg_norm_from(nullptr);
    P.xn = sg_norm_from(x ? x_norm : nullptr);
    return sg_dispatch_igemm(P, (hipStream_t)stream);
}
