// Conv2d / ConvTranspose2d backward-weight (+ bias) for gfx950 (CDNA4), fp32.
//
//   dW[co][kcol] += sum_pixels dOut[pixel][co] * Acol[pixel][kcol],  kcol = (tap, ci)
//
// Acol is the same on-the-fly im2col gather (with the producer's norm + activation applied while
// staging) that the forward kernel uses, so (tap, ci) is one flat GEMM-N dimension and even the
// 2-channel image layers fill a 64-wide tile.  The reduction runs over pixels:
//  * both operands arrive in their natural NHWC form [pixel][channel] and are written to LDS
//    untransposed ([32 pixels][C + pad] rows, pad chosen so the two 16-lane halves of a
//    ds_read_b32 fragment read hit disjoint banks);
//  * v_mfma_f32_16x16x4_f32 with A = dOut^T (rows = co), B = Acol (cols = kcol), k = pixel;
//  * split-K over pixel ranges; partial tiles are combined with fp32 atomics straight into the
//    gradient buffer (which the optimizer's zero_grad memsets), the bias gradient is the column
//    sum of the dOut tile taken by the kcol-tile-0 workgroups.
//
// Reference ops replaced: convolution_backward (weight, bias) of every nn.Conv2d /
// nn.ConvTranspose2d on the path (models/networks.py:502-529, :815-835).
#include <type_traits>

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
    int32_t Hin, Win, in_ld;
    int32_t Hout, Wout, dout_ld;
    int32_t pro_count, pro_sq;
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
    float pro_slope, pro_eps;
    int32_t oa[SGAN_MAX_PHASES], ob[SGAN_MAX_PHASES], ntaps[SGAN_MAX_PHASES], ktot[SGAN_MAX_PHASES];
    SgTap taps[SGAN_MAX_PHASES][SGAN_MAX_TAPS];
    SgWgradProb q[SGW_MAX_PROB];
};

struct SgWgradLocal {
    const float* in; const float* dout; float* dw; float* dbias;
    int32_t Hin, Win, Cin, in_ld, Hout, Wout, Cout, dout_ld, is, os, w_ns, nsplit;
    SgNorm pro;
};

// PRO: the forward input gets the producer's norm + activation applied while it is staged
template <int BCO, int BKC, int WGC, int WGK, bool PRO>
__global__ __launch_bounds__(256) void sg_wgrad_kernel(const SgWgradParams G) {
    constexpr int BP = 32;
    constexpr int WTC = BCO / WGC, WTK = BKC / WGK, MB = WTC / 16, NB = WTK / 16;
    constexpr int LDD = (BCO % 32 == 0) ? BCO + 16 : BCO;
    constexpr int LDA = (BKC % 32 == 0) ? BKC + 16 : BKC;
    constexpr int DQ = BCO / 4, AQ = BKC / 4;  // float4 per pixel row
    constexpr int D_IT = (BP * DQ + 255) / 256, A_IT = (BP * AQ + 255) / 256;
    static_assert(WGC * WGK == 4, "4 waves");
    static_assert(256 % AQ == 0 && 256 % DQ == 0, "row mapping");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ds = reinterpret_cast<float*>(smem);  // [2][BP*LDD]
    float* As = Ds + 2 * BP * LDD;               // [2][BP*LDA]
    float* pscale = As + 2 * BP * LDA;           // [Cin]
    float* pshift = pscale + G.Cin;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid / WGK, wk = wid % WGK;
    const int fr = lane & 15, fq = lane >> 4;
    int g = 0;
    for (int gi = 1; gi < G.nprob; ++gi)
        if ((int)blockIdx.z >= G.q[gi].z0) g = gi;
    const SgWgradProb& Q = G.q[g];
    SgWgradLocal P;
    P.in = Q.in; P.dout = Q.dout; P.dw = Q.dw; P.dbias = Q.dbias;
    P.Hin = Q.Hin; P.Win = Q.Win; P.Cin = G.Cin; P.in_ld = Q.in_ld; P.Hout = Q.Hout; P.Wout = Q.Wout; P.Cout = G.Cout;
    P.dout_ld = Q.dout_ld; P.is = G.is; P.os = G.os; P.w_ns = G.w_ns; P.nsplit = Q.nsplit;
    P.pro.stats = Q.pro_stats; P.pro.gamma = Q.pro_gamma; P.pro.beta = Q.pro_beta; P.pro.count = Q.pro_count;
    P.pro.eps = G.pro_eps; P.pro.act = G.pro_act; P.pro.slope = G.pro_slope; P.pro.sq_stride = Q.pro_sq;
    const int zl = blockIdx.z - Q.z0;
    const int phz = zl / P.nsplit, split = zl % P.nsplit;
    const int Hp = Q.Hp[phz], Wp = Q.Wp[phz], M = Hp * Wp, ktot = G.ktot[phz];
    const int ph_oa = G.oa[phz], ph_ob = G.ob[phz];
    const int kc0 = blockIdx.x * BKC, co0 = blockIdx.y * BCO;
    if (kc0 >= ktot || M == 0) return;
    const int nchunk_total = (M + BP - 1) / BP;
    const int per = (nchunk_total + P.nsplit - 1) / P.nsplit;
    const int ch_begin = split * per, ch_end = min(nchunk_total, ch_begin + per);
    if (ch_begin >= ch_end) return;
    const int Cin = P.Cin, Cout = P.Cout;
    if constexpr (PRO) {
        for (int c = tid; c < Cin; c += 256) {
            float sc = 1.f, sh = 0.f;
            if (P.pro.stats) {
                float mean, rstd;
                sg_mean_rstd(P.pro, Cin, c, mean, rstd);
                const float g = P.pro.gamma ? P.pro.gamma[c] : 1.f;
                const float b = P.pro.beta ? P.pro.beta[c] : 0.f;
                sc = g * rstd;
                sh = b - mean * sc;
            }
            pscale[c] = sc;
            pshift[c] = sh;
        }
    }

    // A (im2col) staging: this thread's column group is fixed for the whole kernel
    const int a_c4 = tid % AQ, a_row0 = tid / AQ;
    constexpr int A_ROWS_PER_IT = 256 / AQ;
    const int a_kcol = kc0 + a_c4 * 4;
    const bool a_kok = a_kcol < ktot;
    const int a_tap = a_kok ? a_kcol / Cin : 0;
    const int a_c = a_kcol - a_tap * Cin;
    const int a_dy = G.taps[phz][a_tap].dy, a_dx = G.taps[phz][a_tap].dx;
    // D staging
    const int d_c4 = tid % DQ, d_row0 = tid / DQ;
    constexpr int D_ROWS_PER_IT = 256 / DQ;
    const bool d_cok = co0 + d_c4 * 4 < Cout;

    // Register ring of NS chunk sets (see sgan_igemm.hip: same pipeline).  Buffer loads: 32-bit byte offsets,
    // an out-of-range offset returns zeros in hardware, so rows past the image / past this split's pixel range
    // and padding taps need no select and no flag (except under PRO, where zero padding applies AFTER the
    // producer's norm + activation).
    constexpr int NS = 3;
    f32x4 a_reg[NS][A_IT], d_reg[NS][D_IT];
    bool a_val[NS][A_IT];
    int a_off_n[A_IT], d_off_n[D_IT];
    bool a_ok_n[A_IT];
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    const int oa = ph_oa, ob = ph_ob;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.dout), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;

    // pixel walk: row r of chunk ch is pixel m = ch*32 + r; (py, px) advance by 32 pixels per chunk with one
    // conditional wrap (32 = adv_y * Wp + adv_x) -- no division inside the loop
    const int adv_y = BP / Wp, adv_x = BP - adv_y * Wp;
    int a_py[A_IT], a_px[A_IT], d_py[D_IT], d_px[D_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = ch_begin * BP + a_row0 + it * A_ROWS_PER_IT;
        a_py[it] = m / Wp;
        a_px[it] = m - a_py[it] * Wp;
    }
#pragma unroll
    for (int it = 0; it < D_IT; ++it) {
        const int m = ch_begin * BP + d_row0 + it * D_ROWS_PER_IT;
        d_py[it] = m / Wp;
        d_px[it] = m - d_py[it] * Wp;
    }
    int ch_next = ch_begin;   // chunk whose offsets next_addrs computes next

    auto next_addrs = [&]() {
        const bool chok = ch_next < ch_end;   // chunks past this split's range belong to another workgroup
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int prow = a_row0 + it * A_ROWS_PER_IT;
            const int iy = a_py[it] * P.is + a_dy, ix = a_px[it] * P.is + a_dx;
            const bool ok = chok && (A_IT * A_ROWS_PER_IT == BP || prow < BP) && a_py[it] < Hp && a_kok &&
                            (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
            a_off_n[it] = ok ? ((iy * P.Win + ix) * P.in_ld + a_c) << 2 : OOB;
            if constexpr (PRO) a_ok_n[it] = ok;
            a_py[it] += adv_y;
            a_px[it] += adv_x;
            if (a_px[it] >= Wp) { a_px[it] -= Wp; ++a_py[it]; }
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            const int prow = d_row0 + it * D_ROWS_PER_IT;
            const bool ok = chok && (D_IT * D_ROWS_PER_IT == BP || prow < BP) && d_py[it] < Hp && d_cok;
            const int pix = (d_py[it] * P.os + oa) * P.Wout + (d_px[it] * P.os + ob);
            d_off_n[it] = ok ? (pix * P.dout_ld + co0 + d_c4 * 4) << 2 : OOB;
            d_py[it] += adv_y;
            d_px[it] += adv_x;
            if (d_px[it] >= Wp) { d_px[it] -= Wp; ++d_py[it]; }
        }
        ++ch_next;
    };

    auto issue_loads = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            a_reg[S][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off_n[it], 0, 0));
            if constexpr (PRO) a_val[S][it] = a_ok_n[it];
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it)
            d_reg[S][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, d_off_n[it], 0, 0));
    };

    auto store_chunk = [&](auto S_, int buf) {
        constexpr int S = decltype(S_)::value;
        float* Ab = As + buf * BP * LDA;
        float* Db = Ds + buf * BP * LDD;
        if constexpr (PRO) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(pscale + a_c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(pshift + a_c);
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int prow = a_row0 + it * A_ROWS_PER_IT;
                f32x4 v = a_reg[S][it];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y = v[j] * sc[j] + sh[j];
                    v[j] = y > 0.f ? y : y * pro_neg;
                }
                if (!a_val[S][it]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (A_IT * A_ROWS_PER_IT == BP || prow < BP) *reinterpret_cast<f32x4*>(Ab + prow * LDA + a_c4 * 4) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int prow = a_row0 + it * A_ROWS_PER_IT;
                if (A_IT * A_ROWS_PER_IT == BP || prow < BP) *reinterpret_cast<f32x4*>(Ab + prow * LDA + a_c4 * 4) = a_reg[S][it];
            }
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            const int prow = d_row0 + it * D_ROWS_PER_IT;
            if (D_IT * D_ROWS_PER_IT == BP || prow < BP) *reinterpret_cast<f32x4*>(Db + prow * LDD + d_c4 * 4) = d_reg[S][it];
        }
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool do_bias = (P.dbias != nullptr) && (blockIdx.x == 0);

    __syncthreads();  // pscale/pshift visible
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    int it_no = 0;   // chunk index relative to ch_begin (LDS buffer = it_no & 1)
    auto iteration = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        issue_loads(std::integral_constant<int, S>{});   // chunk it_no + NS
        __builtin_amdgcn_sched_barrier(0);
        const int buf = it_no & 1;
        const float* Ab = As + buf * BP * LDA;
        const float* Db = Ds + buf * BP * LDD;
#pragma unroll
        for (int k4 = 0; k4 < BP / 4; ++k4) {
            float af[MB], bf[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = Db[(k4 * 4 + fq) * LDD + wc * WTC + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < NB; ++j) bf[j] = Ab[(k4 * 4 + fq) * LDA + wk * WTK + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias && tid < BCO) {
            float sb = 0.f;
#pragma unroll 8
            for (int p = 0; p < BP; ++p) sb += Db[p * LDD + tid];
            bsum += sb;
        }
        store_chunk(std::integral_constant<int, (S + 1) % NS>{}, buf ^ 1);   // chunk it_no + 1
        next_addrs();
        ++it_no;
        __syncthreads();
    };
    next_addrs();
    issue_loads(J0{});
    next_addrs();
    issue_loads(J1{});
    next_addrs();
    issue_loads(J2{});
    next_addrs();
    store_chunk(J0{}, 0);
    __syncthreads();
    {
        const int n_it = ch_end - ch_begin;
        int i = 0;
        for (; i + 2 < n_it; i += 3) {
            iteration(J0{});
            iteration(J1{});
            iteration(J2{});
        }
        if (i < n_it) iteration(J0{});
        if (i + 1 < n_it) iteration(J1{});
    }

    // ---- combine: fp32 atomics into the gradient buffer ----
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int kcol = kc0 + wk * WTK + j * 16 + fr;
        if (kcol >= ktot) continue;
        const int tap = kcol / Cin, ci = kcol - tap * Cin;
        float* base = P.dw + G.taps[phz][tap].w_off + ci;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wc * WTC + i * 16 + fq * 4 + r;
                if (co < Cout) atomicAdd(base + (int64_t)co * P.w_ns, acc[i][j][r]);
            }
        }
    }
    if (do_bias && tid < BCO && co0 + tid < Cout) atomicAdd(P.dbias + co0 + tid, bsum);
}

static inline int sgw_cdiv(int a, int b) { return (a + b - 1) / b; }

template <int BCO, int BKC, int WGC, int WGK>
static int sg_launch_wgrad(SgWgradParams& P, hipStream_t st) {
    constexpr int LDD = (BCO % 32 == 0) ? BCO + 16 : BCO;
    constexpr int LDA = (BKC % 32 == 0) ? BKC + 16 : BKC;
    int maxK = 0;
    for (int i = 0; i < P.nphase; ++i) maxK = max(maxK, P.ktot[i]);
    const int tiles = sgw_cdiv(maxK, BKC) * sgw_cdiv(P.Cout, BCO);
    // pixel-range split per problem: aim at ~1024 workgroups over the whole launch, >= 4 chunks of 32 pixels each
    long chunks_total = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        chunks_total += (long)sgw_cdiv(maxM, 32) * P.nphase;
    }
    if (chunks_total == 0) return SGAN_OK;
    const double want_z = 1024.0 / tiles;   // total z extent we would like
    int z = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        const int nchunk = sgw_cdiv(maxM, 32);
        int nsplit = (int)(want_z * ((double)nchunk * P.nphase / (double)chunks_total) / P.nphase + 0.5);
        if (nsplit > nchunk / 4) nsplit = nchunk / 4;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 512) nsplit = 512;
        P.q[g].nsplit = nsplit;
        P.q[g].z0 = z;
        z += P.nphase * nsplit;
    }
    dim3 grid(sgw_cdiv(maxK, BKC), sgw_cdiv(P.Cout, BCO), z);
    const size_t lds = (size_t)(2 * 32 * LDD + 2 * 32 * LDA + 2 * P.Cin) * 4;
    bool pro = P.pro_act != SGAN_ACT_NONE;
    for (int g = 0; g < P.nprob; ++g) pro = pro || P.q[g].pro_stats != nullptr;
    sg_prof_begin(st);
    if (pro) hipLaunchKernelGGL((sg_wgrad_kernel<BCO, BKC, WGC, WGK, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((sg_wgrad_kernel<BCO, BKC, WGC, WGK, false>), grid, dim3(256), lds, st, P);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = BCO == 64 ? "sg_wgrad_kernel<64,64,2,2>" : BCO == 32 ? "sg_wgrad_kernel<32,64,1,4>"
                                                                            : "sg_wgrad_kernel<16,128,1,4>";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

extern "C" int sgan_conv_wgrad_grouped(const sgan_conv_wgrad_job* jobs, int32_t n, void* stream) {
    SGAN_CHECK(jobs && n >= 1 && n <= SGW_MAX_PROB, "1..%d jobs", SGW_MAX_PROB);
    SgWgradParams P;
    memset(&P, 0, sizeof(P));
    P.nprob = n;
    const sgan_conv_desc* d0 = jobs[0].d;
    if (!d0) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    const sgan_norm_desc* n0 = jobs[0].in_norm;
    P.pro_act = n0 ? n0->act : SGAN_ACT_NONE; P.pro_slope = n0 ? n0->slope : 0.f; P.pro_eps = n0 ? n0->eps : 0.f;
    for (int g = 0; g < n; ++g) {
        const sgan_conv_wgrad_job& J = jobs[g];
        const sgan_conv_desc* d = J.d;
        if (!d) return sgan_fail(SGAN_ERR_INVALID, "null desc");
        SGAN_CHECK((d->Cin & 3) == 0 && (d->Cout & 3) == 0, "stored channels must be multiples of 4");
        SGAN_CHECK(J.in && J.dout && J.dw, "null tensor in job %d", g);
        SGAN_CHECK(J.in_ld >= d->Cin && J.dout_ld >= d->Cout && (J.in_ld & 3) == 0 && (J.dout_ld & 3) == 0, "bad leading dims in job %d", g);
        SGAN_CHECK(d->kind == d0->kind && d->k == d0->k && d->stride == d0->stride && d->pad == d0->pad && d->Cin == d0->Cin &&
                       d->Cout == d0->Cout, "grouped problems must be the same layer type");
        SGAN_CHECK((J.in_norm ? J.in_norm->act : SGAN_ACT_NONE) == P.pro_act, "grouped jobs must share the prologue activation");
        SgPhase ph[SGAN_MAX_PHASES];
        int nphase, is, os;
        int rc = sg_build_phases(d, false, ph, &nphase, &is, &os);
        if (rc) return rc;
        if (g == 0) {
            P.nphase = nphase; P.is = is; P.os = os;
            for (int i = 0; i < nphase; ++i) {
                P.oa[i] = ph[i].oa; P.ob[i] = ph[i].ob; P.ntaps[i] = ph[i].ntaps; P.ktot[i] = ph[i].ktot;
                for (int t = 0; t < ph[i].ntaps; ++t) P.taps[i][t] = ph[i].taps[t];
            }
        }
        SgWgradProb& Q = P.q[g];
        for (int i = 0; i < nphase; ++i) { Q.Hp[i] = ph[i].Hp; Q.Wp[i] = ph[i].Wp; }
        Q.in = J.in; Q.dout = J.dout; Q.dw = J.dw; Q.dbias = J.dbias;
        Q.Hin = d->Hin; Q.Win = d->Win; Q.in_ld = J.in_ld; Q.Hout = d->Hout; Q.Wout = d->Wout; Q.dout_ld = J.dout_ld;
        Q.pro_stats = J.in_norm ? J.in_norm->stats : nullptr;
        Q.pro_gamma = J.in_norm ? J.in_norm->gamma : nullptr;
        Q.pro_beta = J.in_norm ? J.in_norm->beta : nullptr;
        Q.pro_count = J.in_norm ? J.in_norm->count : 1;
        Q.pro_sq = J.in_norm ? J.in_norm->sq_stride : 0;
    }
    P.Cin = d0->Cin; P.Cout = d0->Cout; P.w_ns = d0->Cin;
    hipStream_t st = (hipStream_t)stream;
    if (d0->Cout <= 16) return sg_launch_wgrad<16, 128, 1, 4>(P, st);
    if (d0->Cout <= 32) return sg_launch_wgrad<32, 64, 1, 4>(P, st);
    return sg_launch_wgrad<64, 64, 2, 2>(P, st);
}

extern "C" int sgan_conv_wgrad(const sgan_conv_desc* d, const float* in, int32_t in_ld, const sgan_norm_desc* in_norm,
                               const float* dout, int32_t dout_ld, float* dw, float* dbias, void* stream) {
    sgan_conv_wgrad_job j = {d, in, in_ld, in_norm, dout, dout_ld, dw, dbias};
    return sgan_conv_wgrad_grouped(&j, 1, stream);
}
