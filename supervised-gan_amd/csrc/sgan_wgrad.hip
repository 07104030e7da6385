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

#include "sgan_wgrad.h"
#include "sgan_c4.h"      // the 4-channel backward-data bodies the thin-pair launch runs beside the thin backward-weight body

// PRO: the forward input gets the producer's norm + activation applied while it is staged
template <int BCO, int BKC, int WGC, int WGK, bool PRO>
__global__ __launch_bounds__(256) void sg_wgrad_kernel(const SgWgradParams G) {
    sg_warm_kernargs<(int)sizeof(SgWgradParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
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
    P.pro.eps = G.pro_eps; P.pro.act = G.pro_act; P.pro.slope = G.pro_slope; P.pro.sq_stride = Q.pro_sq; P.pro.rep_stride = Q.pro_rep;
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
    const int a_dy = G.taps[G.tap0[phz] + a_tap].dy, a_dx = G.taps[G.tap0[phz] + a_tap].dx;
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
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.dout), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;

    // pixel walk: row r of chunk ch is pixel m = ch*32 + r; (py, px) advance by 32 pixels per chunk with one
    // conditional wrap (32 = adv_y * Wp + adv_x).  Input coordinates and the two linear element offsets advance with
    // them by constants (a second constant on the wrap) -- no division and no multiply inside the loop.
    const int adv_y = BP / Wp, adv_x = BP - adv_y * Wp;
    int a_py[A_IT], a_px[A_IT], a_iy[A_IT], a_ix[A_IT], a_lin[A_IT], d_py[D_IT], d_px[D_IT], d_lin[D_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = ch_begin * BP + a_row0 + it * A_ROWS_PER_IT;
        a_py[it] = m / Wp;
        a_px[it] = m - a_py[it] * Wp;
        a_iy[it] = a_py[it] * P.is + a_dy;
        a_ix[it] = a_px[it] * P.is + a_dx;
        a_lin[it] = (a_iy[it] * P.Win + a_ix[it]) * P.in_ld + a_c;
    }
#pragma unroll
    for (int it = 0; it < D_IT; ++it) {
        const int m = ch_begin * BP + d_row0 + it * D_ROWS_PER_IT;
        d_py[it] = m / Wp;
        d_px[it] = m - d_py[it] * Wp;
        d_lin[it] = ((d_py[it] * P.os + ph_oa) * P.Wout + (d_px[it] * P.os + ph_ob)) * P.dout_ld + co0 + d_c4 * 4;
    }
    const int a_iy_step = adv_y * P.is, a_ix_step = adv_x * P.is, a_ix_wrap = -Wp * P.is;
    const int a_lin_step = (a_iy_step * P.Win + a_ix_step) * P.in_ld, a_lin_wrap = (P.is * P.Win + a_ix_wrap) * P.in_ld;
    const int d_lin_step = (adv_y * P.os * P.Wout + adv_x * P.os) * P.dout_ld, d_lin_wrap = (P.os * P.Wout - Wp * P.os) * P.dout_ld;
    int ch_next = ch_begin;   // chunk whose offsets next_addrs computes next

    auto next_addrs = [&]() {
        const bool chok = ch_next < ch_end;   // chunks past this split's range belong to another workgroup
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int prow = a_row0 + it * A_ROWS_PER_IT;
            // bitwise on purpose (see sgan_igemm.hip): no branches inside the region shared with the MFMA block
            const bool ok = chok & (A_IT * A_ROWS_PER_IT == BP || prow < BP) & (a_py[it] < Hp) & a_kok &
                            ((unsigned)a_iy[it] < (unsigned)P.Hin) & ((unsigned)a_ix[it] < (unsigned)P.Win);
            a_off_n[it] = ok ? a_lin[it] << 2 : OOB;
            if constexpr (PRO) a_ok_n[it] = ok;
            a_px[it] += adv_x;
            const bool wrap = a_px[it] >= Wp;
            a_px[it] -= wrap ? Wp : 0;
            a_py[it] += adv_y + (wrap ? 1 : 0);
            a_iy[it] += a_iy_step + (wrap ? P.is : 0);
            a_ix[it] += a_ix_step + (wrap ? a_ix_wrap : 0);
            a_lin[it] += a_lin_step + (wrap ? a_lin_wrap : 0);
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            const int prow = d_row0 + it * D_ROWS_PER_IT;
            const bool ok = chok & (D_IT * D_ROWS_PER_IT == BP || prow < BP) & (d_py[it] < Hp) & d_cok;
            d_off_n[it] = ok ? d_lin[it] << 2 : OOB;
            d_px[it] += adv_x;
            const bool wrap = d_px[it] >= Wp;
            d_px[it] -= wrap ? Wp : 0;
            d_py[it] += adv_y + (wrap ? 1 : 0);
            d_lin[it] += d_lin_step + (wrap ? d_lin_wrap : 0);
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

    f32x4 bacc = (f32x4){0.f, 0.f, 0.f, 0.f};   // this thread's channel quad d_c4 summed over the pixel rows it stages
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
                // okf * act(y) = max(okf * y, okf * neg * y), neg <= 1: packed fp32 math (see sgan_igemm.hip)
                const float okf = a_val[S][it] ? 1.f : 0.f;
                const float okn = okf * pro_neg;
                const f32x4 y = a_reg[S][it] * sc + sh;
                const f32x4 yp = y * okf, yn = y * okn;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(yp[j], yn[j]);
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
            if (D_IT * D_ROWS_PER_IT == BP || prow < BP) {
                *reinterpret_cast<f32x4*>(Db + prow * LDD + d_c4 * 4) = d_reg[S][it];
                bacc += d_reg[S][it];   // bias gradient = column sums of dOut: every chunk passes through here exactly once
            }
        }
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (P.dbias != nullptr) && (blockIdx.x == 0);

    SG_SYNC();  // pscale/pshift visible
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
        // Fragment reads run half a chunk ahead of the MFMAs that consume them (a read issued right before its MFMA
        // leaves the matrix pipe idle for the LDS latency, 8 times per chunk); the second half of the MFMAs shares its
        // scheduling region with the transform + LDS store of the next chunk and the address arithmetic.
        constexpr int KH = BP / 8;
        float af[2 * KH][MB], bf[2 * KH][NB];
        auto read_frags = [&](int k0) {
#pragma unroll
            for (int k4 = k0; k4 < k0 + KH; ++k4) {
#pragma unroll
                for (int i = 0; i < MB; ++i) af[k4][i] = Db[(k4 * 4 + fq) * LDD + wc * WTC + i * 16 + fr];
#pragma unroll
                for (int j = 0; j < NB; ++j) bf[k4][j] = Ab[(k4 * 4 + fq) * LDA + wk * WTK + j * 16 + fr];
            }
        };
        auto mfmas = [&](int k0) {
#pragma unroll
            for (int k4 = k0; k4 < k0 + KH; ++k4)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k4][i], bf[k4][j], acc[i][j], 0, 0, 0);
        };
        read_frags(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(KH);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(KH);
        store_chunk(std::integral_constant<int, (S + 1) % NS>{}, buf ^ 1);   // chunk it_no + 1
        next_addrs();
        ++it_no;
        SG_SYNC();
    };
    next_addrs();
    issue_loads(J0{});
    next_addrs();
    issue_loads(J1{});
    next_addrs();
    issue_loads(J2{});
    next_addrs();
    store_chunk(J0{}, 0);
    SG_SYNC();
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
        float* base = P.dw + G.taps[G.tap0[phz] + tap].w_off + ci;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wc * WTC + i * 16 + fq * 4 + r;
                if (co < Cout) atomicAdd(base + (int64_t)co * P.w_ns, acc[i][j][r]);
            }
        }
    }
    if (do_bias) {   // (uniform) combine the per-thread column sums: threads tid = d_c4 + DQ * row share a channel quad
        f32x4* red = reinterpret_cast<f32x4*>(smem);   // the staging buffers are dead: every wave is past the last barrier
        red[tid] = bacc;
        SG_SYNC();
        if (tid < BCO && co0 + tid < Cout) {
            float sb = 0.f;
#pragma unroll
            for (int r = 0; r < 256 / DQ; ++r) sb += reinterpret_cast<const float*>(red + (tid >> 2) + DQ * r)[tid & 3];
            atomicAdd(P.dbias + co0 + tid, sb);
        }
    }
}

static inline int sgw_cdiv(int a, int b) { return (a + b - 1) / b; }

template <int BCO, int BKC, int WGC, int WGK>
static int sg_launch_wgrad(SgWgradParams& P, hipStream_t st) {
    constexpr int LDD = (BCO % 32 == 0) ? BCO + 16 : BCO;
    constexpr int LDA = (BKC % 32 == 0) ? BKC + 16 : BKC;
    int maxK = 0;
    for (int i = 0; i < P.nphase; ++i) maxK = max(maxK, P.ktot[i]);
    const int tiles = sgw_cdiv(maxK, BKC) * sgw_cdiv(P.Cout, BCO);
    // pixel-range split per problem: ~512 workgroups over the whole launch with the SAME number of 32-pixel chunks each
    // (the longest workgroup is the critical path), >= 4 chunks per workgroup
    long chunks_total = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        chunks_total += (long)sgw_cdiv(maxM, 32) * P.nphase;
    }
    if (chunks_total == 0) return SGAN_OK;
    // 512 workgroups (two per CU): with the fragment-prefetching main loop more splits only add atomics (sweep on the fcgan step)
    const double want = getenv("SGAN_WGRAD_WANT") ? atof(getenv("SGAN_WGRAD_WANT")) : 512.0;
    int per = (int)((double)chunks_total * tiles / want + 0.999);
    if (per < 4) per = 4;
    int z = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        const int nchunk = sgw_cdiv(maxM, 32);
        int nsplit = sgw_cdiv(nchunk, per);
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

// ------------------------------------------------------------------------------------------
// Thin layers: one side of the conv has only 4 stored channels (image-side first convs, image / logits
// heads).  dW is then a [rows][16 taps x 4] matrix reduced over up to 10^6 pixels -- bandwidth-bound
// streaming, not a GEMM worth tiling.  Both cases are one kernel:
//   SWAP = false (stored Cin == 4):  rows = co,  dense operand = dOut[p][co],
//                                    gathered operand = x[p*s + tap][0..3]          (walk over output pixels p)
//   SWAP = true  (stored Cout == 4): rows = ci,  dense operand = act(norm(x[q][ci])),
//                                    gathered operand = dOut[q (+) tap][0..3]       (walk over INPUT pixels q)
// so the MFMA N dimension is always the 64 (tap, thin channel) pairs and no MFMA row is wasted on padding.
// Every wave owns a pixel range, feeds the MFMA straight from global memory (no LDS staging, no barriers in
// the loop) and keeps its whole 16*MB x 64 partial in accumulators:
//   * B operand, lane (fr, fq): tap fr of pixel p + fq, its 4 channels = one 16-byte load;
//   * A operand, lane (fr, fq): rows r0 + fr*MB + i (i < MB) of the same pixel = consecutive floats;
//   * the waves of a workgroup combine their partials through LDS and the workgroup adds its tile into dW (see below).
// (The row permutation relative to the wide kernel is only a relabelling of MFMA tiles.)
// ------------------------------------------------------------------------------------------
#ifndef SG_THIN_U
#define SG_THIN_U 4      // steps per load group, two groups in flight (U = 8 measured slower: 20.3 vs 17.5 us on the six-problem first-layer launch;
#endif                   // in-kernel stamps: ~1000 cycles per 4-pixel step, one memory round trip per group turn)

#ifndef SG_THIN_ABL
#define SG_THIN_ABL 0     // diagnostics builds only (wrong results): 1 no final atomics, 2 no main loop, 4 no norm prologue
#endif
#ifdef SGTHIN_STAMP      // diagnostics build: per-workgroup s_memtime stamps (tools/stamp_thin.py reads them through sgan_debug_stamps_thin)
__device__ unsigned long long sgthin_stamps[8 * 4096];
#define SGT_MARK(i)                                                                                        \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (threadIdx.x == 0 && blockIdx.z * gridDim.y + blockIdx.y < 4096) {                              \
            unsigned long long t_;                                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            sgthin_stamps[(blockIdx.z * gridDim.y + blockIdx.y) * 8 + (i)] = (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : t_; \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)
extern "C" int sgan_debug_stamps_thin(void* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sgthin_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#else
#define SGT_MARK(i)
#endif
#define SG_THIN_NW 8      // waves per workgroup: two per SIMD, so a SIMD has a second instruction stream while one waits on its loads

// Round 3: ONE pass.  Round 2 ran ~1024 four-wave workgroups of ~170 pixels each (11 steps per wave), wrote 8.5 MB of partial tiles
// and summed them in a second launch (8 launches per fcgan step, 8-9 us each plus the boundary).  Now ~256 eight-wave workgroups (one
// per CU, 2 waves per SIMD: the exact-fp32 MFMAs of this kernel are 4.5 us of matrix-pipe time spread over every SIMD of the chip),
// 2 x SG_THIN_U steps in flight per wave, the eight partials of a workgroup meet in LDS and the workgroup adds its tile to dW with
// fp32 atomics whose 64 lanes cover 256 CONTIGUOUS bytes (lane -> (row, thin channel) of one tap: the full-rate form of
// MI355X_MICROARCH.md "Global float atomics"; round 2's direct-atomic experiment scattered 16-byte pieces 512 B apart).
// (body: workgroup coordinates come in as `by`, `bz` so that sg_bwd_thin_pair_kernel can run it on a slice of its grid; NW = 4 there --
// a 256-thread workgroup like its partner's, whose waves are the SIMDs' second instruction stream)
template <int MB, bool PRO, bool SWAP, int NW>
__device__ __forceinline__ void sg_wgrad_thin_body(const SgWgradParams& G, const int nbias_z0, char* smem, const int by, const int bz) {
    constexpr int ROWS = 16 * MB, COLS = 64, NB = 4, NT = 64 * NW;
    static_assert(MB == 1 || MB == 2 || MB == 4, "operand widths");
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    float* red = reinterpret_cast<float*>(smem);   // [4 slots][ROWS][COLS]: waves w and w + 4 share slot w
    float* redb = red + 4 * ROWS * COLS;           // [4 slots][ROWS]
    float* pss = redb + 4 * ROWS;                  // [2 * Cin] prologue scale | shift
    int* woff = reinterpret_cast<int*>(pss + 2 * G.Cin);   // [16] weight slab offset of a tap
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    SGT_MARK(0); SGT_MARK(6);
    if constexpr (SWAP) {
        if (bz >= nbias_z0) {   // bias gradient of a thin-Cout layer: sum of dOut over its pixels (<= 4 channels), few workgroups
            const SgWgradProb& Q = G.q[(bz - nbias_z0) >> 2];
            if (!Q.dbias || by != 0) return;
            const int npix = Q.Hout * Q.Wout, part = (bz - nbias_z0) & 3;
            f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int p = part * NT + tid; p < npix; p += 4 * NT) s4 += *reinterpret_cast<const f32x4*>(Q.dout + (int64_t)p * Q.dout_ld);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = s4[c];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) red[wid * 4 + c] = v;
            }
            SG_SYNC();
            if (tid < 4 && tid < G.Cout) {
                float v = 0.f;
                for (int w = 0; w < NW; ++w) v += red[w * 4 + tid];
                atomicAdd(Q.dbias + tid, v);
            }
            return;
        }
    }
    int g = 0;
    for (int gi = 1; gi < G.nprob; ++gi)
        if (bz >= G.q[gi].z0) g = gi;
    const SgWgradProb& Q = G.q[g];
    const int zl = bz - Q.z0;
    const int phz = zl / Q.nsplit, split = zl % Q.nsplit;
    const int Hp = Q.Hp[phz], Wp = Q.Wp[phz], M = Hp * Wp;
    if (M == 0) return;
    const int rows_total = SWAP ? G.Cin : G.Cout;
    const int r0 = by * ROWS;
    // this wave's pixel range: the workgroup's share of M, cut into NW (multiples of 4 pixels)
    const int per_wg = ((M + Q.nsplit - 1) / Q.nsplit + 4 * NW - 1) / (4 * NW) * (4 * NW);
    const int per_wave = per_wg / NW;
    const int m_begin = split * per_wg + wid * per_wave;
    const int m_end = min(M, m_begin + per_wave);

    // lane constants: column group = tap fr, rows r0 + fr*MB + i
    const bool kok = fr < G.ntaps[phz];
    const int tdy = kok ? (int)G.taps[G.tap0[phz] + fr].dy : 0, tdx = kok ? (int)G.taps[G.tap0[phz] + fr].dx : 0;
    if (tid < 16) woff[tid] = tid < G.ntaps[phz] ? G.taps[G.tap0[phz] + tid].w_off : -1;
    const int row_l = r0 + fr * MB;
    const bool rok = row_l < rows_total;
    const int gs = G.is, os = G.os, oa = G.oa[phz], ob = G.ob[phz];
    const float* gat = SWAP ? Q.dout : Q.in;
    const float* den = SWAP ? Q.in : Q.dout;
    const int gat_H = SWAP ? Q.Hout : Q.Hin, gat_W = SWAP ? Q.Wout : Q.Win, gat_ld = SWAP ? Q.dout_ld : Q.in_ld;
    const int den_W = SWAP ? Q.Win : Q.Wout, den_ld = SWAP ? Q.in_ld : Q.dout_ld;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gat), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(den), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;
    constexpr int NP = SWAP ? MB : NB;      // channels of x this lane touches
    float psc[NP], psh[NP];
    float pro_neg = 1.f;
    if constexpr (PRO) {
        // per-channel scale / shift once per workgroup (fp64 divide + sqrt per channel), then each lane picks its own
        SgNorm pn;
        pn.stats = Q.pro_stats; pn.gamma = Q.pro_gamma; pn.beta = Q.pro_beta; pn.count = Q.pro_count;
        pn.eps = G.pro_eps; pn.act = G.pro_act; pn.slope = G.pro_slope; pn.sq_stride = Q.pro_sq; pn.rep_stride = Q.pro_rep;
        pro_neg = G.pro_act == SGAN_ACT_NONE ? 1.f : (G.pro_act == SGAN_ACT_RELU ? 0.f : G.pro_slope);
        for (int c = tid; c < G.Cin; c += NT) {
            float sc = 1.f, sh = 0.f;
            if (pn.stats && !(SG_THIN_ABL & 4)) {
                float mean, rstd;
                sg_mean_rstd(pn, G.Cin, c, mean, rstd);
                const float gm = pn.gamma ? pn.gamma[c] : 1.f;
                const float bt = pn.beta ? pn.beta[c] : 0.f;
                sc = gm * rstd;
                sh = bt - mean * sc;
            }
            pss[c] = sc;
            pss[G.Cin + c] = sh;
        }
        SG_SYNC();
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int c = SWAP ? row_l + j : j;
            const bool cv = c < G.Cin;
            psc[j] = cv ? pss[c] : 1.f;
            psh[j] = cv ? pss[G.Cin + c] : 0.f;
        }
    }
    const bool do_bias = !SWAP && Q.dbias != nullptr;
    // thin channels that carry data (2 of the 4 stored for the fcgan image, 1 for the logits): the MFMAs of the padding are skipped
    const int nthin = G.thin_real;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) bsum[i] = 0.f;

    // pixel walk of this lane: m = m_begin + fq, advancing by 4 per step (no division in the loop)
    int m = m_begin + fq;
    int py = m / Wp, px = m - py * Wp;
    const int adv_y = 4 / Wp, adv_x = 4 - adv_y * Wp;

    constexpr int U = SG_THIN_U;     // steps per group; group k+1 is in flight while group k feeds the MFMAs
    f32x4 xb[2][U];
    float da[2][U][MB];
    bool gok[2][U], dok[2][U];
    auto load_group = [&](int slot) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = m < m_end;
            const int gy = py * gs + tdy, gx = px * gs + tdx;
            const bool ok = valid && kok && (unsigned)gy < (unsigned)gat_H && (unsigned)gx < (unsigned)gat_W;
            xb[slot][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, ok ? ((gy * gat_W + gx) * gat_ld) << 2 : OOB, 0, 0));
            gok[slot][u] = ok;
            const int pix = (py * os + oa) * den_W + (px * os + ob);
            const int dof = (valid && rok) ? (pix * den_ld + row_l) << 2 : OOB;
            dok[slot][u] = valid && rok;
            if constexpr (MB == 1) {
                da[slot][u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, dof, 0, 0));
            } else if constexpr (MB == 2) {
                da[slot][u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, dof, 0, 0));
                da[slot][u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, dof, 4, 0));
            } else {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, dof, 0, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) da[slot][u][i] = v[i];
            }
            m += 4;
            py += adv_y;
            px += adv_x;
            while (px >= Wp) { px -= Wp; ++py; }
        }
    };
    auto compute_group = [&](int slot) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float b[NB], a[MB];
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = xb[slot][u][e];
#pragma unroll
            for (int i = 0; i < MB; ++i) a[i] = da[slot][u][i];
            if constexpr (PRO && !SWAP) {   // x is the gathered operand: zero padding applies after norm + activation
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float y = b[j] * psc[j] + psh[j];
                    b[j] = gok[slot][u] ? (y > 0.f ? y : y * pro_neg) : 0.f;
                }
            }
            if constexpr (PRO && SWAP) {    // x is the dense operand
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const float y = a[i] * psc[i] + psh[i];
                    a[i] = dok[slot][u] ? (y > 0.f ? y : y * pro_neg) : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                bsum[i] += a[i];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nthin) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);      // wave-uniform skip
            }
        }
    };
    const int nsteps = (SG_THIN_ABL & 2) ? 0 : (max(m_end - m_begin, 0) + 3) >> 2;
    const int ngroups = (nsteps + U - 1) / U;
    // branch-free body (loads past the range carry the out-of-range offset and return zeros): a conditional load
    // would make the compiler wait for ALL outstanding loads at the join, serialising the two slots
    SGT_MARK(1);
    load_group(0);
    for (int gk = 0; gk < ngroups; gk += 2) {
        load_group(1);
        compute_group(0);
        load_group(0);
        compute_group(1);
    }

    // ---- combine the eight waves through LDS: waves 4..7 store, waves 0..3 add their own on top (same lane -> element map),
    //      then every thread sums the four slots of its elements ----
    auto stash = [&](bool add) {
        float* mine = red + (wid & 3) * ROWS * COLS;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][e][r];
                f32x4* dst = reinterpret_cast<f32x4*>(mine + ((fq * 4 + r) * MB + i) * COLS + fr * NB);
                if (add) v += *dst;
                *dst = v;
            }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            float b = bsum[i];
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            if (fq == 0) redb[(wid & 3) * ROWS + fr * MB + i] = add ? redb[(wid & 3) * ROWS + fr * MB + i] + b : b;
        }
    };
    SGT_MARK(2);
    if constexpr (NW == 8) {
        if (wid >= 4) stash(false);
        SG_SYNC();
        if (wid < 4) stash(true);
    } else {
        stash(false);
    }
    SG_SYNC();
    SGT_MARK(3);
    // fp32 atomics, 64 consecutive lanes -> 64 consecutive floats of dW:
    //   cin4 (SWAP = false): dW[tap][co][c4]  -> e' = tap * (ROWS * 4) + rl * 4 + c4
    //   cout4 (SWAP = true): dW[tap][c4][ci]  -> e' = (tap * 4 + c4) * ROWS + rl
    for (int e2 = tid; e2 < ROWS * COLS; e2 += NT) {
        int rl, kl;
        if constexpr (SWAP) { kl = e2 / ROWS; rl = e2 - kl * ROWS; }
        else { const int t = e2 / (ROWS * 4), r = e2 - t * (ROWS * 4); rl = r >> 2; kl = t * 4 + (r & 3); }
        const int e = rl * COLS + kl;
        const float v = (red[e] + red[ROWS * COLS + e]) + (red[2 * ROWS * COLS + e] + red[3 * ROWS * COLS + e]);
        const int t = kl >> 2, c4 = kl & 3, row = r0 + rl;
        const int wo = woff[t];
        if (row < rows_total && wo >= 0 && c4 < (SWAP ? G.Cout : G.Cin) && (!(SG_THIN_ABL & 1) || v == 123.f))
            atomicAdd(Q.dw + wo + (SWAP ? (int64_t)c4 * G.w_ns + row : (int64_t)row * G.w_ns + c4), v);
    }
    if (tid < ROWS && do_bias && r0 + tid < rows_total)
        atomicAdd(Q.dbias + r0 + tid, (redb[tid] + redb[ROWS + tid]) + (redb[2 * ROWS + tid] + redb[3 * ROWS + tid]));
    SGT_MARK(4); SGT_MARK(7);
}

template <int MB, bool PRO, bool SWAP>
__global__ __launch_bounds__(64 * SG_THIN_NW) void sg_wgrad_thin_kernel(const SgWgradParams G, int nbias_z0) {
    sg_warm_kernargs<(int)sizeof(SgWgradParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_wgrad_thin_body<MB, PRO, SWAP, SG_THIN_NW>(G, nbias_z0, smem, (int)blockIdx.y, (int)blockIdx.z);
}

struct SgThinPlan { int nrb, z, zgrid; size_t lds; bool pro; };

// pixel split per problem (nsplit, z0) for ~`want_wgs` workgroups over the launch; false: nothing to do
template <int MB, bool SWAP>
static bool sg_thin_plan(SgWgradParams& P, double want_wgs, SgThinPlan* out) {
    constexpr int ROWS = 16 * MB, COLS = 64, PS = ROWS * COLS + ROWS;
    const int nrb = sgw_cdiv(SWAP ? P.Cin : P.Cout, ROWS);
    long pix_total = 0;
    for (int g = 0; g < P.nprob; ++g)
        for (int i = 0; i < P.nphase; ++i) pix_total += (long)P.q[g].Hp[i] * P.q[g].Wp[i];
    if (pix_total == 0) return false;
    const int min_pix = getenv("SGAN_THIN_MINPIX") ? atoi(getenv("SGAN_THIN_MINPIX")) : 256;
    const double want = want_wgs / (double)nrb;
    int z = 0;
    for (int g = 0; g < P.nprob; ++g) {
        long pg = 0;
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) {
            pg += (long)P.q[g].Hp[i] * P.q[g].Wp[i];
            maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        }
        int nsplit = (int)(want * ((double)pg / (double)pix_total) / P.nphase + 0.5);
        if (nsplit > maxM / min_pix) nsplit = maxM / min_pix;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 1024) nsplit = 1024;
        P.q[g].nsplit = nsplit;
        P.q[g].z0 = z;
        z += P.nphase * nsplit;
    }
    out->nrb = nrb;
    out->z = z;
    out->zgrid = z + (SWAP ? 4 * P.nprob : 0);      // SWAP: four bias workgroups per problem behind the tiles
    out->lds = (size_t)(4 * PS + 2 * P.Cin + 16) * 4;
    out->pro = P.pro_act != SGAN_ACT_NONE;
    for (int g = 0; g < P.nprob; ++g) out->pro = out->pro || P.q[g].pro_stats != nullptr;
    return true;
}

template <int MB, bool SWAP>
static int sg_launch_wgrad_thin(SgWgradParams& P, hipStream_t st, const char* name, void* workspace, int64_t workspace_bytes) {
    (void)workspace;
    if (workspace_bytes == -1) return 0;      // single pass since round 3: no workspace
    // one eight-wave workgroup per CU over the whole launch, >= 256 pixels per workgroup
    const char* want_env = getenv("SGAN_THIN_WANT");        // tuning knob
    SgThinPlan pl;
    if (!sg_thin_plan<MB, SWAP>(P, want_env ? atof(want_env) : 256.0, &pl)) return SGAN_OK;
    dim3 grid(1, pl.nrb, pl.zgrid);
    sg_prof_begin(st);
    if (pl.pro) hipLaunchKernelGGL((sg_wgrad_thin_kernel<MB, true, SWAP>), grid, dim3(64 * SG_THIN_NW), pl.lds, st, P, pl.z);
    else hipLaunchKernelGGL((sg_wgrad_thin_kernel<MB, false, SWAP>), grid, dim3(64 * SG_THIN_NW), pl.lds, st, P, pl.z);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = name;
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

// thin-Cout layers walk the INPUT pixels: one "phase", the k*k taps gather dOut at q*gs + (dy, dx)
static void sg_thin_swap_geometry(SgWgradParams& P, const sgan_conv_desc* d0, const sgan_conv_wgrad_job* jobs, int n) {
    const int k = d0->k, p = d0->pad;
    const bool tr = d0->kind == SGAN_CONVT;
    P.nphase = 1;
    P.tap0[0] = 0;
    P.is = tr ? d0->stride : 1;
    P.os = 1;
    P.oa[0] = P.ob[0] = 0;
    P.ntaps[0] = k * k;
    P.ktot[0] = 64;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            SgTap& t = P.taps[ky * k + kx];
            t.dy = (int16_t)(tr ? ky - p : p - ky);
            t.dx = (int16_t)(tr ? kx - p : p - kx);
            t.w_off = (ky * k + kx) * d0->Cout * d0->Cin;
        }
    for (int g = 0; g < n; ++g) {
        P.q[g].Hp[0] = jobs[g].d->Hin;
        P.q[g].Wp[0] = jobs[g].d->Win;
    }
}

int sg_build_wgrad_params(const sgan_conv_wgrad_job* jobs, int32_t n, SgWgradParams& P, bool allow_f16) {
    SGAN_CHECK(jobs && n >= 1 && n <= SGW_MAX_PROB, "1..%d jobs", SGW_MAX_PROB);
    memset(&P, 0, sizeof(P));
    P.nprob = n;
    const sgan_conv_desc* d0 = jobs[0].d;
    if (!d0) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    const sgan_norm_desc* n0 = jobs[0].in_norm;
    P.pro_act = n0 ? n0->act : SGAN_ACT_NONE; P.pro_slope = n0 ? n0->slope : 0.f; P.pro_eps = n0 ? n0->eps : 0.f;
    SGAN_CHECK(P.pro_act != SGAN_ACT_LRELU || P.pro_slope <= 1.f, "LeakyReLU slope must be <= 1 (the kernels evaluate max(y, slope * y))");
    for (int g = 0; g < n; ++g) {
        const sgan_conv_wgrad_job& J = jobs[g];
        const sgan_conv_desc* d = J.d;
        if (!d) return sgan_fail(SGAN_ERR_INVALID, "null desc");
        SGAN_CHECK((d->Cin & 3) == 0 && (d->Cout & 3) == 0, "stored channels must be multiples of 4");
        SGAN_CHECK(J.in && J.dout && J.dw, "null tensor in job %d", g);
        SGAN_CHECK(J.in_ld >= d->Cin && J.dout_ld >= d->Cout && (J.in_ld & 3) == 0 && (J.dout_ld & 3) == 0, "bad leading dims in job %d", g);
        SGAN_CHECK(d->kind == d0->kind && d->k == d0->k && d->stride == d0->stride && d->pad == d0->pad && d->Cin == d0->Cin &&
                       d->Cout == d0->Cout && d->math == d0->math, "grouped problems must be the same layer type and math mode");
        SGAN_CHECK((J.in_norm ? J.in_norm->act : SGAN_ACT_NONE) == P.pro_act, "grouped jobs must share the prologue activation");
        SgPhase ph[SGAN_MAX_PHASES];
        int nphase, is, os;
        int rc = sg_build_phases(d, false, ph, &nphase, &is, &os);
        if (rc) return rc;
        if (g == 0) {
            P.nphase = nphase; P.is = is; P.os = os;
            for (int i = 0; i < nphase; ++i) {
                P.oa[i] = ph[i].oa; P.ob[i] = ph[i].ob; P.ntaps[i] = ph[i].ntaps; P.ktot[i] = ph[i].ktot;
                P.tap0[i] = i == 0 ? 0 : P.tap0[i - 1] + ph[i - 1].ntaps;
                for (int t = 0; t < ph[i].ntaps; ++t) P.taps[P.tap0[i] + t] = ph[i].taps[t];
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
        Q.pro_rep = J.in_norm ? J.in_norm->rep_stride : 0;
        Q.amax = J.dout_amax;
    }
    P.Cin = d0->Cin; P.Cout = d0->Cout; P.w_ns = d0->Cin;
    bool f16 = allow_f16;       // fp16 planes when every job brings its gradient's maximum (sgan_wgrad3.hip); else bf16 planes
    for (int g = 0; g < n; ++g) f16 = f16 && jobs[g].dout_amax;
    static const int no_f16 = getenv("SGAN_NO_F16_BWD") ? atoi(getenv("SGAN_NO_F16_BWD")) : 0;
    P.planes_f16 = (f16 && !no_f16) ? 1 : 0;
    return SGAN_OK;
}

extern "C" int sgan_conv_wgrad_grouped(const sgan_conv_wgrad_job* jobs, int32_t n, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
    SgWgradParams P;
    int rcb = sg_build_wgrad_params(jobs, n, P);
    if (rcb) return rcb;
    const sgan_conv_desc* d0 = jobs[0].d;
    hipStream_t st = (hipStream_t)stream;
    if (!getenv("SGAN_NO_THIN_WGRAD") && d0->k * d0->k <= 16) {
        if (d0->Cin == 4 && P.nphase == 1) {      // conv-form gather of a 4-channel image
            P.thin_real = (d0->Cin_logical > 0 && d0->Cin_logical < 4) ? d0->Cin_logical : 4;
            if (d0->Cout <= 16) return sg_launch_wgrad_thin<1, false>(P, st, "sg_wgrad_thin_kernel<1,cin4>", workspace, workspace_bytes);
            if (d0->Cout <= 32) return sg_launch_wgrad_thin<2, false>(P, st, "sg_wgrad_thin_kernel<2,cin4>", workspace, workspace_bytes);
            return sg_launch_wgrad_thin<4, false>(P, st, "sg_wgrad_thin_kernel<4,cin4>", workspace, workspace_bytes);
        }
        if (d0->Cout == 4 && d0->Cin >= 16 && (d0->kind == SGAN_CONVT || d0->stride == 1)) {
            SgWgradParams S = P;
            sg_thin_swap_geometry(S, d0, jobs, n);
            S.thin_real = (d0->Cout_logical > 0 && d0->Cout_logical < 4) ? d0->Cout_logical : 4;
            int rc;
            if (d0->Cin <= 32) rc = sg_launch_wgrad_thin<2, true>(S, st, "sg_wgrad_thin_kernel<2,cout4>", workspace, workspace_bytes);
            else rc = sg_launch_wgrad_thin<4, true>(S, st, "sg_wgrad_thin_kernel<4,cout4>", workspace, workspace_bytes);
            if (rc != 1 || workspace_bytes == -1) return rc;
        }
    }
    if (workspace_bytes == -1) return 0;   // the tiled kernels combine their splits with atomics: no workspace
    if (d0->math == SGAN_MATH_BF16X3) {      // split-bf16 MFMA (sgan_wgrad3.hip) where it covers the layer
        const int r3 = sg_launch_wgrad3(P, st);
        if (r3 != 0) return r3 < 0 ? r3 : SGAN_OK;
    }
    if (d0->Cout <= 16) return sg_launch_wgrad<16, 128, 1, 4>(P, st);
    if (d0->Cout <= 32) return sg_launch_wgrad<32, 64, 1, 4>(P, st);
    return sg_launch_wgrad<64, 64, 2, 2>(P, st);
}

// ------------------------------------------------------------------------------------------
// The two backward launches of a layer with a 4-channel side in ONE grid: the generator's output layer (ConvT 32 -> 2: backward-data
// on sg_conv_c4_body, backward-weight on the thin kernel, 15 + 19 us apart) and the first PatchGAN conv in a generator update
// (4 -> 32: sg_conv_scatter4_body + the thin kernel, 19 + 16 us).  Neither feeds the other, each leaves most of the chip waiting on
// its own memory round trips, and a hipGraph replays kernel nodes one after the other -- the same argument as sg_bwd_fused_kernel.
// The first `ndg` workgroups run the backward-data body, the rest the thin backward-weight body with four waves per workgroup
// (a 256-thread workgroup like its partner's).  DK: 0 = c4 (full epilogue, <= 32 result channels), 1 = scatter4.
// ------------------------------------------------------------------------------------------
template <int DK, int RB, int MB, bool WPRO, bool SWAP>
__global__ __launch_bounds__(256) void sg_bwd_thin_pair_kernel(const SgIgemmParams G, const SgWgradParams W, int ndg, int dgx, int wy, int nbias_z0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_warm_kernargs<(int)(sizeof(SgIgemmParams) + sizeof(SgWgradParams))>();
    const int b = blockIdx.x;
    if (b < ndg) {
        if constexpr (DK == 0) sg_conv_c4_body<2, RB, true>(G, b % dgx, b / dgx);
        else sg_conv_scatter4_body(G, smem, b);
    } else {
        const int w = b - ndg;
        sg_wgrad_thin_body<MB, WPRO, SWAP, 4>(W, nbias_z0, smem, w % wy, w / wy);
    }
}

// 0: launched; 1: not such a pair (the caller issues the two grouped calls); < 0: error
extern "C" int sgan_conv_bwd_thin_pair(const sgan_conv_dgrad_job* djobs, int32_t nd, const sgan_conv_wgrad_job* wjobs, int32_t nw, void* stream) {
    static const int off = getenv("SGAN_NO_THIN_PAIR") ? 1 : 0;
    if (off || getenv("SGAN_NO_THIN_WGRAD")) return 1;
    SgIgemmParams P;
    SgWgradParams W;
    int rc = sg_build_dgrad_params(djobs, nd, P, false);
    if (rc) return rc;
    rc = sg_build_wgrad_params(wjobs, nw, W, false);
    if (rc) return rc;
    const sgan_conv_desc* d0 = wjobs[0].d;
    if (d0->k * d0->k > 16 || djobs[0].d->kind != d0->kind || djobs[0].d->Cin != d0->Cin || djobs[0].d->Cout != d0->Cout) return 1;
    bool swap;
    if (d0->Cin == 4 && W.nphase == 1 && d0->Cout > 16 && d0->Cout <= 32) {      // conv-form gather of a 4-channel image, MB = 2
        swap = false;
        W.thin_real = (d0->Cin_logical > 0 && d0->Cin_logical < 4) ? d0->Cin_logical : 4;
    } else if (d0->Cout == 4 && d0->Cin >= 16 && d0->Cin <= 32 && (d0->kind == SGAN_CONVT || d0->stride == 1)) {
        swap = true;
        sg_thin_swap_geometry(W, d0, wjobs, nw);
        W.thin_real = (d0->Cout_logical > 0 && d0->Cout_logical < 4) ? d0->Cout_logical : 4;
    } else {
        return 1;
    }
    // the backward-data half: where sg_dispatch_igemm (sgan_igemm.hip) would have sent it
    P.ksplit = 1;
    P.slab = nullptr;
    P.slab_stride = 0;
    bool skinny = P.N == 4;
    for (int g = 0; g < P.nprob; ++g)
        if (P.q[g].xref || P.q[g].stats || (P.q[g].out_ld & 3) || P.q[g].accum) skinny = false;
    int dk, ndg, dgx = 1, RB = 4;
    size_t lds_d = 0;
    if (skinny && sg_use_scatter4(P)) {
        dk = 1;
        ndg = sg_scatter4_plan(P, &lds_d);
    } else if (P.N > 4 && P.N <= 32 && sg_igemm3_eligible(P) == 0 && sg_use_c4(P) && sg_c4_needs_epi(P)) {
        dk = 0;
        RB = sg_c4_pick_rb(P);
        if (RB == 1) RB = 2;
        dgx = sg_fill_tiles(P, 64 * RB);
        ndg = dgx * ((P.N + 31) / 32);
    } else {
        return 1;
    }
    if (ndg == 0 || (dk == 1) != !swap) return 1;      // (the two pairs of the fcgan step: scatter4 + cin4, c4 + cout4)
    static const double want = getenv("SGAN_THIN_PAIR_WANT") ? atof(getenv("SGAN_THIN_PAIR_WANT")) : 512.0;      // four-wave workgroups
    SgThinPlan pl;
    const bool ok = swap ? sg_thin_plan<2, true>(W, want, &pl) : sg_thin_plan<2, false>(W, want, &pl);
    if (!ok) return 1;
    const size_t lds = lds_d > pl.lds ? lds_d : pl.lds;
    if (lds > 64 * 1024 && lds_d <= 64 * 1024 && pl.lds <= 64 * 1024) return 1;
    const dim3 grid(ndg + pl.nrb * pl.zgrid);
    hipStream_t st = (hipStream_t)stream;
    sg_prof_begin(st);
    if (dk == 1) {
        if (pl.pro) hipLaunchKernelGGL((sg_bwd_thin_pair_kernel<1, 4, 2, true, false>), grid, dim3(256), lds, st, P, W, ndg, dgx, pl.nrb, pl.z);
        else hipLaunchKernelGGL((sg_bwd_thin_pair_kernel<1, 4, 2, false, false>), grid, dim3(256), lds, st, P, W, ndg, dgx, pl.nrb, pl.z);
    } else if (RB == 2) {
        if (pl.pro) hipLaunchKernelGGL((sg_bwd_thin_pair_kernel<0, 2, 2, true, true>), grid, dim3(256), lds, st, P, W, ndg, dgx, pl.nrb, pl.z);
        else hipLaunchKernelGGL((sg_bwd_thin_pair_kernel<0, 2, 2, false, true>), grid, dim3(256), lds, st, P, W, ndg, dgx, pl.nrb, pl.z);
    } else {
        if (pl.pro) hipLaunchKernelGGL((sg_bwd_thin_pair_kernel<0, 4, 2, true, true>), grid, dim3(256), lds, st, P, W, ndg, dgx, pl.nrb, pl.z);
        else hipLaunchKernelGGL((sg_bwd_thin_pair_kernel<0, 4, 2, false, true>), grid, dim3(256), lds, st, P, W, ndg, dgx, pl.nrb, pl.z);
    }
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_bwd_thin_pair_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

extern "C" int sgan_conv_wgrad(const sgan_conv_desc* d, const float* in, int32_t in_ld, const sgan_norm_desc* in_norm,
                               const float* dout, int32_t dout_ld, float* dw, float* dbias, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    sgan_conv_wgrad_job j = {d, in, in_ld, in_norm, dout, dout_ld, dw, dbias, nullptr};
    return sgan_conv_wgrad_grouped(&j, 1, workspace, workspace_bytes, stream);
}
