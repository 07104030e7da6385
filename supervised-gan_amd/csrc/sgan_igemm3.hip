// Implicit-GEMM Conv2d / ConvTranspose2d forward and backward-data for gfx950 (CDNA4), split-bf16 arithmetic
// (SGAN_MATH_BF16X3): fp32 tensors in HBM, fp32 accumulation, every product evaluated as
//     a * b  ~=  a_hi * b_hi + a_hi * b_lo + a_lo * b_hi,      x_hi = bf16(x),  x_lo = bf16(x - x_hi)
// on v_mfma_f32_32x32x16_bf16 (16x the per-clock rate of the fp32 MFMA, so three of them still retire 16/3 of the fp32
// rate).  What is dropped is a_lo * b_lo (<= 2^-16 |a b|) and the rounding of the lo parts (<= 2^-17): the result is an
// fp32-equivalent product, measured 1e-6 .. 1e-5 of the output scale per layer against the exact-fp32 kernel.
//
// Same GEMM view, gather, prologue ("normalise-on-load") and epilogue as sg_igemm_kernel (sgan_igemm.hip); what differs:
//  * weights come pre-split from the packed copy (sgan_pack_weights): rows [n][k/8]{8 hi | 8 lo}, so the B operand is
//    staged with plain 16-byte copies; activations are split while they are staged (after the prologue transform):
//    6 VALU per 2 elements (cvt_pk, shift, and, 2 sub, cvt_pk);
//  * LDS tile rows are 128 bytes = one 32-deep k-tile of one row: 8 chunks {8 bf16} = (k-group 0..3) x (hi, lo) at
//    slot 2 * kgroup + plane, chunk position XOR-swizzled with (row >> 1) & 7 -- conflict free for the ds_read_b128
//    fragment reads (a lane group reads one slot of 16 rows) and for the ds_write_b128 of both operands (A: eight lanes
//    write four k-groups of rows r and r + 2; B: eight lanes write the eight slots of one row);
//  * MFMA 32x32x16: lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8h + j] / B[k = 8h + j][col r], j < 8, i.e.
//    exactly one LDS chunk per (k16 step, plane); the 32x32 shape leaves 24 of every 32 cycles of VALU issue beside an
//    MFMA (16x16x32: 8 of 16), which the staging arithmetic needs;
//  * tiles: 128x128 with eight waves (wave tile 64x32), 128x64 / 64x64 / 128x32 with four.  LDS traffic, not the MFMA
//    pipe, bounds the small tiles (a 64x64 tile moves 256 LDS-array cycles per k-tile against 192 MFMA cycles; 128x128:
//    640 against 768), so the dispatcher takes the largest tile that still fills the chip.
//
// Reference ops replaced: nn.Conv2d / nn.ConvTranspose2d forward + convolution_backward(input) as instantiated at
// models/networks.py:502-529 (FCGANGenerator), :815-835 (NLayerDiscriminator), :356-398 (U-Net), :686-774 (CRN).
#include <type_traits>

#include "sgan_igemm.h"

#ifndef SG3_ABL
#define SG3_ABL 0   // diagnostics builds only (wrong results): 1 no global loads, 2 no transform / split, 4 no LDS stores, 8 no address arithmetic, 16 no MFMA, 32 no fragment reads
#endif

typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sg_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sg_f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 fp32 -> {8 x 16-bit hi, 8 x 16-bit lo}.  hi = RNE(x); lo = RNE(x - hi) (the subtraction is exact).  F16: fp16 planes
// (11 significant bits each: x = hi + lo to 2^-23 |x| as long as lo stays a normal fp16, i.e. |x| >= 0.25; below, to 3e-8 absolute --
// the forward operands are post-normalisation activations of order 1 and weights scaled by 2^10); else bf16 planes (8 bits each,
// 2^-17 |x|, the fp32 exponent range: gradients).
template <bool F16>
__device__ __forceinline__ void sg_split8(const f32x4 v0, const f32x4 v1, u32x4& hi, u32x4& lo) {
    float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 p = {x[2 * i], x[2 * i + 1]};
        if constexpr (F16) {
            const sg_f16x2 h = __builtin_convertvector(p, sg_f16x2);                     // v_cvt_pk_f16_f32
            const f32x2 r = p - __builtin_convertvector(h, f32x2);
            hi[i] = __builtin_bit_cast(unsigned, h);
            lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sg_f16x2));
        } else {
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(p, sg_bf16x2));   // v_cvt_pk_bf16_f32
            const f32x2 r = {x[2 * i] - __builtin_bit_cast(float, h << 16), x[2 * i + 1] - __builtin_bit_cast(float, h & 0xffff0000u)};
            hi[i] = h;
            lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sg_bf16x2));
        }
    }
}

template <bool F16>
__device__ __forceinline__ f32x16 sg3_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sg_f16x8, a), __builtin_bit_cast(sg_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sg_bf16x8, a), __builtin_bit_cast(sg_bf16x8, b), c, 0, 0, 0);
}

// byte offset of 16-byte chunk `slot` of tile row `row` (128-byte rows)
__device__ __forceinline__ int sg3_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// ------------------------------------------------------------------------------------------
// Epilogue shared by the split kernels: acc[i][j][r] is tile row wm*WTM + i*32 + (r & 3) + 8 (r >> 2) + 4 fh, column
// n0 + wn*WTN + j*32 + fr; `rowpix(row)` gives the pixel index of a tile row in the result tensor (< 0: outside the problem).
// Bias, fp64 statistics of the result (forward) or act'(norm(x)) and the two norm-backward sums (backward-data), tanh, accumulate;
// or the raw partial tile to the split-K slab.  `red` is [2 BN] fp64 of LDS, zeroed before the main loop.
// ------------------------------------------------------------------------------------------
// Per-column constants of the epilogue (bias; mean / rstd / affine of the norm the forward consumer applied), fetched BEFORE the
// main loop: read in the epilogue itself they put two dependent memory round trips in front of the first store.
template <int NB>
struct Sg3EpiConst { float bias_v[NB], x_mean[NB], x_rstd[NB], x_g[NB], x_b[NB]; };

template <int WTN, int NB>
__device__ __forceinline__ Sg3EpiConst<NB> sg3_epilogue_constants(const SgLocal& P, int n0, int wn, int tid) {
    Sg3EpiConst<NB> E;
    const int fr = tid & 31;
    const bool xnorm = P.xref != nullptr && P.xn.stats != nullptr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * WTN + j * 32 + fr;
        const bool nv = n < P.N;
        E.bias_v[j] = (P.bias && nv) ? P.bias[n] : 0.f;
        E.x_mean[j] = 0.f; E.x_rstd[j] = 1.f; E.x_g[j] = 1.f; E.x_b[j] = 0.f;
        if (xnorm && nv) {
            sg_mean_rstd(P.xn, P.N, n, E.x_mean[j], E.x_rstd[j]);
            E.x_g[j] = P.xn.gamma ? P.xn.gamma[n] : 1.f;
            E.x_b[j] = P.xn.beta ? P.xn.beta[n] : 0.f;
        }
    }
    return E;
}

template <int BN, int WTM, int WTN, int MB, int NB, bool F16, typename RowPix>
__device__ __forceinline__ void sg3_epilogue(const SgLocal& P, f32x16 (&acc)[MB][NB], double* red, int split, int n0, int wm, int wn,
                                             int tid, unsigned bid, const Sg3EpiConst<NB>& EC, RowPix rowpix) {
    const int lane = tid & 63, fr = lane & 31, fh = lane >> 5;
    const int N = P.N;
    if constexpr (F16) {     // the fp16 weight planes hold w * 2^SGAN_F16_WEIGHT_SHIFT, a gathered gradient came in times 2^s: exact powers of two
        const float os = P.out_scale;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] *= os;
    }
    const bool want_stats = P.stats != nullptr;
    if (P.ksplit > 1) {   // split-K: raw partial tile to this split's slab; sg_splitk_epilogue_kernel finishes
        float* sl = P.slab + (int64_t)split * P.slab_stride;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t pix = rowpix(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh);
                if (pix >= 0) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int n = n0 + wn * WTN + j * 32 + fr;
                        if (n < N) sl[pix * N + n] = acc[i][j][r];
                    }
                }
            }
        }
        return;
    }
    const bool dact = P.xref != nullptr;
    const bool xnorm = dact && P.xn.stats != nullptr;
    const float xn_neg = P.xn.act == SGAN_ACT_NONE ? 1.f : (P.xn.act == SGAN_ACT_RELU ? 0.f : P.xn.slope);
    const float (&bias_v)[NB] = EC.bias_v, (&x_mean)[NB] = EC.x_mean, (&x_rstd)[NB] = EC.x_rstd, (&x_g)[NB] = EC.x_g, (&x_b)[NB] = EC.x_b;
    bool nvalid[NB];
    double s1[NB], s2[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        nvalid[j] = n0 + wn * WTN + j * 32 + fr < N;
        s1[j] = 0.0;
        s2[j] = 0.0;
    }
    // the forward tensor at the result positions, all loads in flight before the first store (the stores below may alias them as far
    // as the compiler knows: read row by row they would each wait out a full memory round trip)
    float xv[MB][16][NB];
    if (dact) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t pix = rowpix(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    xv[i][r][j] = (pix >= 0 && nvalid[j]) ? P.xref[pix * P.xref_ld + n0 + wn * WTN + j * 32 + fr] : 0.f;
            }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t pix = rowpix(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh);
            if (pix >= 0) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int n = n0 + wn * WTN + j * 32 + fr;
                    if (nvalid[j]) {
                        float v = acc[i][j][r] + bias_v[j];
                        if (dact) {
                            const float x = xv[i][r][j];
                            const float xhat = (x - x_mean[j]) * x_rstd[j];
                            const float y = xnorm ? (x_g[j] * xhat + x_b[j]) : x;
                            v *= (y > 0.f ? 1.f : xn_neg);
                            s1[j] += (double)v;
                            s2[j] += (double)(v * xhat);
                        } else {
                            s1[j] += (double)v;
                            s2[j] += (double)v * (double)v;
                            if (P.out_act == SGAN_ACT_TANH) v = tanhf(v);
                        }
                        if (P.accum) v += P.out[pix * P.out_ld + n];
                        P.out[pix * P.out_ld + n] = v;
                    }
                }
            }
        }
    }
    if (want_stats) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int nl = wn * WTN + j * 32 + fr;
            double a = s1[j], b = s2[j];
            a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 32);
            if (fh == 0 && nvalid[j]) {
                atomicAdd(&red[nl], a);
                atomicAdd(&red[BN + nl], b);
            }
        }
        SG_SYNC();
        if (tid < BN && n0 + tid < N) {
#ifndef SG_NO_STAT_ATOMICS      // diagnostics build: what the same-address fp64 atomics cost
            double* st = sg_stat_replica(P.stats, P.stats_rep, bid);
            atomicAdd(&st[n0 + tid], red[tid]);
            atomicAdd(&st[P.stats_sq + n0 + tid], red[BN + tid]);
#endif
        }
    }
}

// F16: operand planes are fp16 (forward pass: fp32-equivalent products) instead of bf16 (backward-data)
// KB2: two k-tiles per barrier (four LDS buffers): with 16-bit MFMAs a 32-deep k-tile is only 192 MFMA cycles per wave, less than
// what a barrier interval costs in waits and bookkeeping; two tiles per interval halve that overhead per flop.
// The kernel body takes its workgroup id, grid size and split index as arguments so that sg_bwd_fused_kernel (sgan_fused.hip) can run
// it on a slice of a launch it shares with the backward-weight body.
template <int BM, int BN, int WGM, int WGN, bool PRO, bool F16, bool KB2>
__device__ __forceinline__ void sg_igemm3_body(const SgIgemmParams& G, char* smem, const int bid, const int nblocks, const int split) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MB = WTM / 32, NB = WTN / 32;
    constexpr int A_IT = BM * 4 / NT;              // tasks of 8 consecutive k of one row (two 16-byte loads) per thread
    constexpr int B_IT = (BN * 8 + NT - 1) / NT;   // 16-byte chunks of the packed weight rows per thread
    static_assert(BM * 4 % NT == 0 && A_IT >= 1, "A tile");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");

    constexpr int NBUF = KB2 ? 4 : 2;
    char* As = smem;                            // [NBUF][BM * 128]
    char* Bs = smem + NBUF * BM * 128;          // [NBUF][BN * 128]
    double* red = reinterpret_cast<double*>(smem + NBUF * (BM + BN) * 128);   // [2 * BN]
    int4* ttab = reinterpret_cast<int4*>(red + 2 * BN);                     // [16] {dy, dx, gather offset, weight slab offset}
    float* pscale = reinterpret_cast<float*>(ttab + SGAN_MAX_TAPS);         // [Ck]
    float* pshift = pscale + G.Ck;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int ntn = (G.N + BN - 1) / BN;
    const int item = sg_xcd_remap(bid, nblocks);
    int g, phz, mtile;
    sg_decode_tile(G, item / ntn, g, phz, mtile);
    const SgLocal P = sg_local(G, g);
    const int Hp = G.q[g].Hp[phz], Wp = G.q[g].Wp[phz];
    const int M = Hp * Wp;
    const int m0 = mtile * BM, n0 = (item % ntn) * BN;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(G.q[g].wp), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;
    const int oa = G.oa[phz], ob = G.ob[phz];
    const int ktot = G.ktot[phz];
    const int Ck = P.Ck, N = P.N;
    const int nkt_total = (ktot + 31) >> 5;
    const int kt_per = (nkt_total + P.ksplit - 1) / P.ksplit;
    const int kt0 = split * kt_per;
    const int nkt = max(min(nkt_total, kt0 + kt_per) - kt0, 0);   // may be 0 for a trailing split: writes zeros

    // ---- one-time setup: tap table, prologue scale/shift, reduction scratch ----
    if (tid < SGAN_MAX_TAPS) {
        const bool v = tid < G.ntaps[phz];
        const SgTap tp = G.taps[v ? G.tap0[phz] + tid : 0];
        const int dy = v ? (int)tp.dy : 0, dx = v ? (int)tp.dx : 0;
        ttab[tid] = make_int4(dy, dx, (dy * P.Win + dx) * P.in_ld, v ? tp.w_off : 0);
    }
    for (int i = tid; i < 2 * BN; i += NT) red[i] = 0.0;
    if constexpr (PRO) {
        for (int c = tid; c < Ck; c += NT) {
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
    }

    // ---- per-thread staging state ----
    // Every thread works on k-group kg = tid & 3 (8 consecutive k = 8 consecutive channels of one tap, Ck % 8 == 0) of its A
    // rows and of its B rows, so one (tap, channel) walk serves both: + 32 k per tile = adv_tap taps + adv_c channels.
    const int kg = tid & 3;
    const int adv_tap = 32 / Ck, adv_c = 32 - adv_tap * Ck;
    const int ntaps = G.ntaps[phz];
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    int a_iy[A_IT], a_ix[A_IT], a_base[A_IT], a_dst[A_IT];
    bool a_rowok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int e = tid + it * NT;
        const int j = e >> 3, u = (e >> 2) & 1;
        const int row = 4 * (j >> 1) + (j & 1) + 2 * u;     // eight consecutive lanes: rows r and r + 2 (bank-conflict-free stores)
        const int m = m0 + row;
        a_rowok[it] = m < M;
        const int py = m / Wp, px = m - py * Wp;
        a_iy[it] = py * P.is;
        a_ix[it] = px * P.is;
        a_base[it] = (a_iy[it] * P.Win + a_ix[it]) * P.in_ld;
        a_dst[it] = sg3_off(row, 2 * kg);        // hi chunk; the lo chunk is slot + 1: byte offset ^ 16
    }
    int a_tap = (kt0 * 32 + kg * 8) / Ck;
    int a_c = kt0 * 32 + kg * 8 - a_tap * Ck;
    int b_base[B_IT], b_dst[B_IT];
    bool b_rowok[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int e = tid + it * NT;
        const int n = e >> 3, t = e & 7;             // t & 3 == kg; plane = t >> 2
        b_rowok[it] = (B_IT * NT == BN * 8 || e < BN * 8) && n0 + n < N;
        b_base[it] = (n0 + n) * P.w_ns + 4 * (t >> 2);
        b_dst[it] = sg3_off(n, 2 * (t & 3) + (t >> 2));
    }

    // Register ring: tile kt+1 being written to LDS, tiles kt+2 .. kt+NSET in flight.  The kernel is bound by the latency x
    // bandwidth of the L2 -> CU path (16 KB per 64x64 k-tile): the more tiles in flight, the more bytes per CU cover it.
#ifndef SG3_NSET
#define SG3_NSET 0
#endif
    constexpr int NSET = KB2 ? 4 : (SG3_NSET ? SG3_NSET : ((2 * A_IT + B_IT <= 4) ? 6 : 4));      // KB2: the LDS buffer of a tile is its slot
    f32x4 a_reg[NSET][A_IT][2];
    bool a_ok[NSET][A_IT];
    int a_cs[NSET];
#pragma unroll
    for (int i = 0; i < NSET; ++i) a_cs[i] = 0;
    u32x4 b_reg[NSET][B_IT];
    int a_off_n[A_IT], b_off_n[B_IT], a_cs_n = 0;
    bool a_ok_n[A_IT];

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    SG_SYNC();  // tap table visible
    const Sg3EpiConst<NB> epi_const = sg3_epilogue_constants<WTN, NB>(P, n0, wn, tid);

    int ld_left = nkt;
    auto next_addrs = [&]() {
        const bool in_range = ld_left > 0;
        --ld_left;
        const bool kok = (a_tap < ntaps) & in_range;
        const int4 t = ttab[kok ? a_tap : 0];
        a_cs_n = kok ? a_c : 0;
        const int toff = t.z + a_c;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int iy = a_iy[it] + t.x, ix = a_ix[it] + t.y;
            // bitwise on purpose: a short-circuit becomes a branch and ends the scheduling region shared with the MFMA block
            const bool ok = a_rowok[it] & kok & ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
            a_off_n[it] = ok ? (a_base[it] + toff) << 2 : OOB;
            if constexpr (PRO) a_ok_n[it] = ok;
        }
        const int woff = t.w + a_c;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) b_off_n[it] = (b_rowok[it] & kok) ? (b_base[it] + woff) << 2 : OOB;
        a_tap += adv_tap;
        a_c += adv_c;
        if (a_c >= Ck) { a_c -= Ck; ++a_tap; }
    };

    auto issue_loads = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        if constexpr (SG3_ABL & 1) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) { a_reg[S][it][0] = (f32x4){1.f, 2.f, 3.f, 4.f}; a_reg[S][it][1] = (f32x4){1.f, 2.f, 3.f, 4.f}; if constexpr (PRO) a_ok[S][it] = a_ok_n[it]; }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) b_reg[S][it] = (u32x4){1u, 2u, 3u, 4u};
            if constexpr (PRO) a_cs[S] = a_cs_n;
            return;
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            a_reg[S][it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off_n[it], 0, 0));
            a_reg[S][it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off_n[it] + 16, 0, 0));
            if constexpr (PRO) a_ok[S][it] = a_ok_n[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            b_reg[S][it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off_n[it], 0, 0));
        if constexpr (PRO) a_cs[S] = a_cs_n;
    };

    // prologue transform (norm + activation of the producer layer), split, write register set S to LDS buffer S & 1.  The
    // scale / shift chunks of the set's channels are read from LDS by load_scales() ahead of the fragment reads, so the
    // counted lgkmcnt wait in front of the transform does not cover the fragments.
    f32x4 sc0, sc1, sh0, sh1;
    auto load_scales = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        if constexpr (PRO) {
            sc0 = *reinterpret_cast<const f32x4*>(pscale + a_cs[S]);
            sc1 = *reinterpret_cast<const f32x4*>(pscale + a_cs[S] + 4);
            sh0 = *reinterpret_cast<const f32x4*>(pshift + a_cs[S]);
            sh1 = *reinterpret_cast<const f32x4*>(pshift + a_cs[S] + 4);
        }
    };
    auto store_tile = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        char* Ab = As + (S & (NBUF - 1)) * BM * 128;
        char* Bb = Bs + (S & (NBUF - 1)) * BN * 128;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            f32x4 v0 = a_reg[S][it][0], v1 = a_reg[S][it][1];
            if constexpr (PRO && !(SG3_ABL & 2)) {
                // okf * act(y) = max(okf * y, okf * neg * y), neg <= 1 (host-checked): zero padding applies AFTER norm + activation
                const float okf = a_ok[S][it] ? 1.f : 0.f;
                const float okn = okf * pro_neg;
                const f32x4 y0 = v0 * sc0 + sh0, y1 = v1 * sc1 + sh1;
                const f32x4 p0 = y0 * okf, q0 = y0 * okn, p1 = y1 * okf, q1 = y1 * okn;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(p0[j], q0[j]); v1[j] = fmaxf(p1[j], q1[j]); }
            }
            if constexpr (F16 && !PRO) { v0 *= P.a_scale; v1 *= P.a_scale; }     // backward-data on fp16 planes: the gradient times 2^s
            u32x4 hi, lo;
            if constexpr (SG3_ABL & 2) { hi = __builtin_bit_cast(u32x4, v0); lo = __builtin_bit_cast(u32x4, v1); }
            else sg_split8<F16>(v0, v1, hi, lo);
            if (!(SG3_ABL & 4) || G.nprob > 100) {
                *reinterpret_cast<u32x4*>(Ab + a_dst[it]) = hi;
                *reinterpret_cast<u32x4*>(Ab + (a_dst[it] ^ 16)) = lo;
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int e = tid + it * NT;
            if ((!(SG3_ABL & 4) || G.nprob > 100) && (B_IT * NT == BN * 8 || e < BN * 8)) *reinterpret_cast<u32x4*>(Bb + b_dst[it]) = b_reg[S][it];
        }
    };

    // fragment addresses: lane (r = lane & 31, h = lane >> 5) reads chunk slot 2 * (2 s + h) + plane of its row
    const int fr = lane & 31, fh = lane >> 5;
    const int fswz = (fr >> 1) & 7;
    int f_off[2][2];   // [k16 step][plane]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) f_off[s][p] = ((2 * (2 * s + fh) + p) ^ fswz) << 4;
    const int fa_row = (wm * WTM + fr) * 128, fb_row = (wn * WTN + fr) * 128;

    // One k-tile per iteration, one barrier.  A wave issues in order, so the order below IS the schedule (pinned with
    // sched_barrier; inside the last phase sched_group_barrier deals the address arithmetic into the MFMA shadows):
    //   (1) global loads of tile kt + NSET, LDS reads of the scale / shift chunks, then every fragment read of tile kt;
    //   (2) transform + split + LDS store of tile kt + 1 (its global loads were issued NSET - 1 iterations ago): VALU work that
    //       needs no fragment -- it covers the LDS latency of (1);
    //   (3) the MFMAs of tile kt, each followed by its share of the address arithmetic of tile kt + NSET + 1.
    // Measured alternatives (D 128 -> 256 @65^2 x 6 problems, 12.2 GFLOP): compiler-scheduled 67 us; these phases 62 us; the two
    // wave halves of an 8-wave workgroup in opposite phase order 73 us; fully software-pipelined (fragments of tile kt + 1 and
    // the store of tile kt + 2 dealt between the MFMAs of tile kt, 40 more VGPRs) 62 us.  The instruction schedule is not what
    // bounds this kernel: it moves 16 KB (64x64 tile) per k-tile through L2 -> L1 -> VGPR -> LDS and the L2s deliver ~12 TB/s
    // chip-wide (TCC_HIT * 128 B / time), which is 190 TFLOP/s at 0.125 B per MAC -- see DESIGN.md for what lifts that.
    auto iteration = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        constexpr int SN = (S + (KB2 ? 2 : 1)) % NSET;      // the tile this iteration moves from registers to LDS
        const char* Ab = As + (S & (NBUF - 1)) * BM * 128 + fa_row;
        const char* Bb = Bs + (S & (NBUF - 1)) * BN * 128 + fb_row;
        issue_loads(std::integral_constant<int, S>{});
        load_scales(std::integral_constant<int, SN>{});
        u32x4 ah[2][MB], al[2][MB], bh[2][NB], bl[2][NB];
        if constexpr (SG3_ABL & 32) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < MB; ++i) { ah[s][i] = __builtin_bit_cast(u32x4, a_reg[0][0][0]); al[s][i] = __builtin_bit_cast(u32x4, a_reg[1][0][0]); }
#pragma unroll
                for (int j = 0; j < NB; ++j) { bh[s][j] = b_reg[0][0]; bl[s][j] = b_reg[1][0]; }
            }
        } else
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                ah[s][i] = *reinterpret_cast<const u32x4*>(Ab + i * 32 * 128 + f_off[s][0]);
                al[s][i] = *reinterpret_cast<const u32x4*>(Ab + i * 32 * 128 + f_off[s][1]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bh[s][j] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * 128 + f_off[s][0]);
                bl[s][j] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * 128 + f_off[s][1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_tile(std::integral_constant<int, SN>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if constexpr (SG3_ABL & 16) { acc[i][j][0] += __builtin_bit_cast(f32x4, al[s][i])[0] + __builtin_bit_cast(f32x4, ah[s][i])[1] + __builtin_bit_cast(f32x4, bl[s][j])[2] + __builtin_bit_cast(f32x4, bh[s][j])[3]; continue; }
                    acc[i][j] = sg3_mfma<F16>(al[s][i], bh[s][j], acc[i][j]);
                    acc[i][j] = sg3_mfma<F16>(ah[s][i], bl[s][j], acc[i][j]);
                    acc[i][j] = sg3_mfma<F16>(ah[s][i], bh[s][j], acc[i][j]);
                }
        if constexpr (!(SG3_ABL & 8)) next_addrs();
        constexpr int NMFMA = 6 * MB * NB;
        constexpr int PER = (40 + 6 * A_IT + 2 * B_IT + NMFMA - 1) / NMFMA;   // next_addrs is ~40 + 6 A_IT + 2 B_IT VALU / SALU
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x006, PER, 0);    // VALU / SALU in its shadow
        }
        if constexpr (!KB2 || (S & 1)) SG_SYNC();
    };
    auto prefetch = [&](auto K_) { next_addrs(); issue_loads(K_); };
    auto maybe = [&](auto K_, int kt) { if (kt + decltype(K_)::value < nkt) iteration(K_); };
#define SG3_FOR_SETS(F, ...)                                                                                              \
    do {                                                                                                                  \
        F(std::integral_constant<int, 0>{}, ##__VA_ARGS__); F(std::integral_constant<int, 1>{}, ##__VA_ARGS__);           \
        F(std::integral_constant<int, 2>{}, ##__VA_ARGS__); F(std::integral_constant<int, 3>{}, ##__VA_ARGS__);           \
        if constexpr (NSET > 4) { F(std::integral_constant<int, 4 % NSET>{}, ##__VA_ARGS__); F(std::integral_constant<int, 5 % NSET>{}, ##__VA_ARGS__); } \
        if constexpr (NSET > 6) { F(std::integral_constant<int, 6 % NSET>{}, ##__VA_ARGS__); F(std::integral_constant<int, 7 % NSET>{}, ##__VA_ARGS__); } \
    } while (0)
    static_assert(NSET == 4 || NSET == 6 || NSET == 8, "ring depth");
    SG3_FOR_SETS(prefetch);
    next_addrs();
    load_scales(std::integral_constant<int, 0>{});
    store_tile(std::integral_constant<int, 0>{});
    if constexpr (KB2) {
        load_scales(std::integral_constant<int, 1>{});
        store_tile(std::integral_constant<int, 1>{});
    }
    SG_SYNC();
    {
        int kt = 0;
        for (; kt + NSET - 1 < nkt; kt += NSET) SG3_FOR_SETS(iteration);
        SG3_FOR_SETS(maybe, kt);     // the last one is never taken (kt + NSET - 1 >= nkt here)
    }
#undef SG3_FOR_SETS

    sg3_epilogue<BN, WTM, WTN, MB, NB, F16>(P, acc, red, split, n0, wm, wn, tid, (unsigned)bid, epi_const, [&](int row) -> int64_t {
        const int m = m0 + row;
        if (m >= M) return -1;
        const int py = m / Wp, px = m - py * Wp;
        return (int64_t)(py * P.os + oa) * P.Wout + (px * P.os + ob);
    });
}

template <int BM, int BN, int WGM, int WGN, bool PRO, bool F16, bool KB2 = false>
__global__ __launch_bounds__(64 * WGM * WGN) void sg_igemm3_kernel(const SgIgemmParams G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();
    sg_igemm3_body<BM, BN, WGM, WGN, PRO, F16, KB2>(G, smem, blockIdx.x, gridDim.x, blockIdx.z);
}


// ------------------------------------------------------------------------------------------
// Patch-stationary variant for unit-stride gathers (Conv2d stride 1 forward and backward-data, every phase of a
// ConvTranspose2d forward / strided-Conv2d backward-data).  sg_igemm3_kernel walks K as (tap, channel): a result tile re-loads,
// re-normalises and re-splits each input element once per tap that touches it -- 16 times for a 4x4 kernel -- and that staging
// (VALU split, L2 -> L1 -> VGPR -> LDS traffic), not the MFMA pipe, is what bounds it.  Here K is walked as (32-channel block, tap):
//  * the M tile is an 8 x 8 block of result pixels; per channel block its input PATCH ((8 + span - 1)^2 pixels, 11 x 11 for
//    4x4 taps) is loaded, transformed and split ONCE and stays in LDS: pixel stride 144 B (32 channels x {hi, lo} + 16 pad),
//    row stride = 128 mod 256 B, so the 16 lanes of a ds_read_b128 group (two tile rows of eight pixels) hit 16 distinct
//    16-byte slots at any tap offset;
//  * per tap the A fragments are read straight out of the patch at a per-tap byte offset (no copy); only the weight tile
//    [BN][32 k] of (tap, channel block) is staged per step, a plain 16-byte copy of the packed rows through a register ring;
//  * one barrier per tap, the LDS reads of the next tap issued between the two MFMA halves of the current one; at a channel-
//    block boundary one more barrier behind the patch store (its loads were issued a block ahead).
// Same prologue / epilogue / packed weights / problem grouping as sg_igemm3_kernel; rectangular tiles waste the ragged edge
// (66 x 66 result: 81 tiles for 68 tiles' worth of pixels), which the dispatcher prices in.
// ------------------------------------------------------------------------------------------
#define SG3P_PS 144
#ifndef SG3P_ABL
#define SG3P_ABL 0   // diagnostics builds only (wrong results, tools/abl3p.sh): 1 no weight loads, 2 no weight LDS stores, 4 no statistics -> scale / shift in the set-up
#endif
__host__ __device__ __forceinline__ int sg3p_row_stride(int ppw) { return ((ppw * SG3P_PS + 127) & ~255) + 128; }

#ifdef SG3P_STAMP      // diagnostics build: per-workgroup s_memtime stamps (tools/stamp3p.py reads them through sgan_debug_stamps)
__device__ unsigned long long sg3p_stamps[8 * 4096];
#define SG3P_MARK(i)                                                                                       \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                       \
            unsigned long long t_;                                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            sg3p_stamps[blockIdx.x * 8 + (i)] = t_;                                                        \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)
extern "C" int sgan_debug_stamps(void* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sg3p_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#else
#define SG3P_MARK(i)
#endif

// S2: stride-2 gather (Conv2d stride 2 forward, ConvTranspose2d stride 2 backward-data; one phase).  The patch of an 8 x 8 result tile is
// then (14 + span)^2 pixels, kept in LDS as FOUR parity planes [row & 1][col & 1][row >> 1][col >> 1], each laid out like a stride-1
// patch: tap (ty, tx) of result pixel (py, px) reads patch pixel (2 py + ty, 2 px + tx) = plane (ty & 1, tx & 1) at (py + (ty >> 1),
// px + (tx >> 1)) -- the same lane addressing as stride 1 plus a per-tap constant, so the fragment reads stay conflict free and the
// loop body does not change.  Each input element is staged (18 / 8)^2 = 5 times for a 4 x 4 kernel instead of 16 (sg_igemm3_kernel).
// KW: wave groups per workgroup (256 threads each).  KW = 2: the two groups take the two halves of the channel blocks of the SAME tile,
// each with its own patch and weight buffers, in lockstep (every barrier is the whole workgroup's); group 1 hands its accumulators to
// group 0 through LDS at the end.  For launches of 64 - 256 workgroups (the generator at batch 1): the serial chain of a tile
// halves and the CU runs two waves per SIMD instead of one.
template <int BN, int A_IT, bool PRO, bool F16, bool S2 = false, int KW = 1>
__device__ __forceinline__ void sg_igemm3p_body(const SgIgemmParams& G, char* smem, const int bid, const int nblocks) {
    constexpr int NT = 256, WGN = 2, WTM = 32, WTN = BN / WGN, MB = 1, NB = WTN / 32;
    static_assert(KW == 1 || KW == 2, "wave groups");
    constexpr int B_IT = BN * 8 / NT;
    // weight-tile register ring (even: the LDS buffer of a step is its slot's parity).  BN = 128 (wave tile 32 x 64, ring of 2)
    // compiles but measured slower on every layer tried (D 128 -> 256 @65^2 x 6: dgrad 73 us against 63.5): the halved workgroup
    // count costs more than the better MFMA : LDS ratio returns, so the dispatcher only uses BN = 64.
    constexpr int NSET = BN == 64 ? 4 : 2;
    static_assert(BN % 64 == 0 && B_IT >= 1, "N tile");

    SG3P_MARK(0);
#ifdef SG3P_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) sg3p_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    const int kgp = KW > 1 ? (int)threadIdx.x >> 8 : 0;               // wave group
    const int tid = KW > 1 ? (int)threadIdx.x & 255 : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;      // index inside the group
    const int wm = wid / WGN, wn = wid % WGN;
    const int ntn = (G.N + BN - 1) / BN;
    const int item = sg_xcd_remap(bid, nblocks);
    int g, phz, mtile;
    sg_decode_tile(G, item / ntn, g, phz, mtile);
    const SgLocal P = sg_local(G, g);
    const int Hp = G.q[g].Hp[phz], Wp = G.q[g].Wp[phz];
    const int tiles_x = (Wp + 7) >> 3;
    const int ty0 = (mtile / tiles_x) * 8, tx0 = (mtile % tiles_x) * 8;
    const int n0 = (item % ntn) * BN;
    const int PH = G.pph[phz], PW = G.ppw[phz];
    const int RS = sg3p_row_stride(S2 ? (PW + 1) >> 1 : PW);
    const int PLANE = S2 ? ((PH + 1) >> 1) * RS : 0;         // bytes of one parity plane
    const int npix = PH * PW;

    const int patch_bytes = ((S2 ? 4 * PLANE : PH * RS) + 255) & ~255;
    char* Ap = smem + kgp * patch_bytes;                     // [PH][RS] patch of the current channel block (S2: four parity planes), one per wave group
    char* Bs = smem + KW * patch_bytes + kgp * (2 * BN * 128);              // [2][BN * 128], one per wave group
    double* red = reinterpret_cast<double*>(smem + KW * patch_bytes + KW * (2 * BN * 128));            // [2 * BN]
    int4* ttab = reinterpret_cast<int4*>(red + 2 * BN);                    // per tap {patch byte offset, -, -, weight slab offset}
    float* pscale = reinterpret_cast<float*>(ttab + SGAN_MAX_TAPS);        // [Ck]
    float* pshift = pscale + G.Ck;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(G.q[g].wp), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;
    const int oa = G.oa[phz], ob = G.ob[phz];
    const int Ck = P.Ck, N = P.N;
    const int ntaps = G.ntaps[phz];
    const int ncb = (Ck >> 5) / KW;          // channel blocks of THIS wave group (the launcher made the count a multiple of KW) ...
    const int cb_base = kgp * ncb;           // ... starting here
    const int nunits = ntaps * ncb;
    const int dy0 = G.pdy0[phz], dx0 = G.pdx0[phz];

    SG3P_MARK(5);
    if (tid < SGAN_MAX_TAPS) {
        const bool v = tid < ntaps;
        const SgTap tp = G.taps[v ? G.tap0[phz] + tid : 0];
        const int ty = (int)tp.dy - dy0, tx = (int)tp.dx - dx0;
        const int poff = S2 ? ((ty & 1) * 2 + (tx & 1)) * PLANE + (ty >> 1) * RS + (tx >> 1) * SG3P_PS : ty * RS + tx * SG3P_PS;
        ttab[tid] = make_int4(v ? poff : 0, 0, 0, v ? tp.w_off : 0);
    }
    for (int i = tid; i < 2 * BN; i += NT) red[i] = 0.0;
    if constexpr (PRO) {
        for (int c = tid; c < Ck; c += NT) {
            float sc = 1.f, sh = 0.f;
            if (P.pro.stats && !(SG3P_ABL & 4)) {
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
    }

    // ---- patch staging: item e = 4 * patch pixel + k-group (8 channels = two 16-byte loads -> one hi and one lo chunk) ----
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    const int kg = tid & 3;
    int a_goff[A_IT], a_dst[A_IT];     // a_goff == OOB: outside the input (zero after the transform); a_dst < 0: no such patch pixel
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int p = (tid + it * NT) >> 2;
        const int pr = p / PW, pc = p - pr * PW;
        const int iy = ty0 * P.is + dy0 + pr, ix = tx0 * P.is + dx0 + pc;
        const bool ok = (p < npix) & ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
        a_goff[it] = ok ? ((iy * P.Win + ix) * P.in_ld + kg * 8) << 2 : OOB;
        const int ldst = S2 ? ((pr & 1) * 2 + (pc & 1)) * PLANE + (pr >> 1) * RS + (pc >> 1) * SG3P_PS : pr * RS + pc * SG3P_PS;
        a_dst[it] = p < npix ? ldst + kg * 32 : -1;
    }
    f32x4 a_reg[A_IT][2];
    auto issue_a = [&](int cb) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int o = ((a_goff[it] != OOB) & (cb < ncb)) ? a_goff[it] + (cb + cb_base) * 128 : OOB;     // past the last block: zeros, never used
            a_reg[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, o, 0, 0));
            a_reg[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, o + 16, 0, 0));
        }
    };
    auto store_a = [&](int cb) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            f32x4 v0 = a_reg[it][0], v1 = a_reg[it][1];
            if constexpr (PRO) {   // zero padding applies AFTER norm + activation (see sg_igemm3_kernel)
                const int c = (cb + cb_base) * 32 + kg * 8;
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(pscale + c), sc1 = *reinterpret_cast<const f32x4*>(pscale + c + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(pshift + c), sh1 = *reinterpret_cast<const f32x4*>(pshift + c + 4);
                const float okf = a_goff[it] != OOB ? 1.f : 0.f;
                const float okn = okf * pro_neg;
                const f32x4 y0 = v0 * sc0 + sh0, y1 = v1 * sc1 + sh1;
                const f32x4 p0 = y0 * okf, q0 = y0 * okn, p1 = y1 * okf, q1 = y1 * okn;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(p0[j], q0[j]); v1[j] = fmaxf(p1[j], q1[j]); }
            }
            if constexpr (F16 && !PRO) { v0 *= P.a_scale; v1 *= P.a_scale; }     // backward-data on fp16 planes: the gradient times 2^s
            u32x4 hi, lo;
            sg_split8<F16>(v0, v1, hi, lo);
            if (a_dst[it] >= 0) {
                *reinterpret_cast<u32x4*>(Ap + a_dst[it]) = hi;
                *reinterpret_cast<u32x4*>(Ap + a_dst[it] + 16) = lo;
            }
        }
    };

    // ---- weight-tile staging (as sg_igemm3_kernel): thread -> (row n, 16-byte chunk t of the 128-byte k-row) ----
    int b_base[B_IT], b_dst[B_IT];
    bool b_rowok[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int e = tid + it * NT;
        const int n = e >> 3, t = e & 7;
        b_rowok[it] = n0 + n < N;
        b_base[it] = (n0 + n) * P.w_ns + 8 * (t & 3) + 4 * (t >> 2);
        b_dst[it] = sg3_off(n, 2 * (t & 3) + (t >> 2));
    }
    u32x4 b_reg[NSET][B_IT];
    int b_off_n[B_IT];
    int ld_tap = 0, ld_cb = 0;
    // The tap-table entries (LDS) are fetched ONE CALL AHEAD of their use: read and used in the same step, each lookup puts an LDS
    // round trip into the in-order instruction stream of every wave (two per step: this one and the fragment offset below).
    int w_pref = 0;            // set behind the barrier that publishes the tap table
    auto next_b_addrs = [&]() {
        const bool in_range = ld_cb < ncb;
        const int woff = w_pref + (ld_cb + cb_base) * 32;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) b_off_n[it] = (b_rowok[it] & in_range) ? (b_base[it] + woff) << 2 : OOB;
        ++ld_tap;
        if (ld_tap == ntaps) { ld_tap = 0; ++ld_cb; }
        w_pref = ttab[ld_tap].w;
    };
    auto issue_b = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            if constexpr (SG3P_ABL & 1) b_reg[S][it] = (u32x4){(unsigned)b_off_n[it], 2u, 3u, 4u};
            else b_reg[S][it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off_n[it], 0, 0));
        }
    };
    auto store_b = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        char* Bb = Bs + (S & 1) * BN * 128;
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            if (!(SG3P_ABL & 2) || G.nprob > 100) *reinterpret_cast<u32x4*>(Bb + b_dst[it]) = b_reg[S][it];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // fragment addresses.  A: lane (r = lane & 31, h = lane >> 5) is tile pixel (4 wm + (r >> 3), r & 7), chunk 2 (2 s + h) + plane
    const int fr = lane & 31, fh = lane >> 5;
    const int fa_base = (4 * wm + (fr >> 3)) * RS + (fr & 7) * SG3P_PS + fh * 32;
    const int fswz = (fr >> 1) & 7;
    int f_off[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) f_off[s][p] = ((2 * (2 * s + fh) + p) ^ fswz) << 4;
    const int fb_row = (wn * WTN + fr) * 128;

    SG_SYNC();   // tap table, scale / shift visible
    const Sg3EpiConst<NB> epi_const = sg3_epilogue_constants<WTN, NB>(P, n0, wn, tid);
    SG3P_MARK(1);

    // Step u = (channel block cur_cb, tap cur_tap); its weight tile lives in LDS buffer u & 1 and register slot u % 4.  Order
    // inside a step (a wave issues in order; pinned with sched_barrier), software-pipelined by k16 halves so that the LDS reads of
    // step u + 1 run under the MFMAs of step u and the first MFMA after the barrier never waits:
    //   MFMA half 0 of u | [last tap of a block: store the next patch, barrier] | fragment reads half 0 of u + 1 |
    //   MFMA half 1 of u | fragment reads half 1 of u + 1 | weight tile of u + 2: registers -> LDS buffer u & 1 (step u's fragments
    //   left it one step ago) | global loads of the tile of u + 6 into the slot just stored | barrier.
    // In-kernel stamps of the unpipelined order (reads, then MFMAs, per step): ~370 cycles of fragment reads, ~190 of MFMA
    // issue and ~350 at the barrier per 64x64x32 step, whether the workgroup shares its CU or not.
    int cur_tap = 0, cur_cb = 0;
    w_pref = ttab[0].w;
    int tapx_pref = ttab[ntaps > 1 ? 1 : 0].x;      // patch offset of the NEXT step's tap, fetched a step ahead
    u32x4 ah[2], al[2], bh[2][NB], bl[2][NB];
    auto read_half = [&](auto H_, const char* Ab, const char* Bb) {
        constexpr int s = decltype(H_)::value;
        ah[s] = *reinterpret_cast<const u32x4*>(Ab + s * 64);
        al[s] = *reinterpret_cast<const u32x4*>(Ab + s * 64 + 16);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bh[s][j] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * 128 + f_off[s][0]);
            bl[s][j] = *reinterpret_cast<const u32x4*>(Bb + j * 32 * 128 + f_off[s][1]);
        }
    };
    auto mfma_half = [&](auto H_) {
        constexpr int s = decltype(H_)::value;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            acc[0][j] = sg3_mfma<F16>(al[s], bh[s][j], acc[0][j]);
            acc[0][j] = sg3_mfma<F16>(ah[s], bl[s][j], acc[0][j]);
            acc[0][j] = sg3_mfma<F16>(ah[s], bh[s][j], acc[0][j]);
        }
    };
    auto iteration = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        mfma_half(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        ++cur_tap;
        if (cur_tap == ntaps) {
            cur_tap = 0;
            ++cur_cb;
            if (cur_cb < ncb) {       // nobody reads the old patch any more (the reads of this step were issued a step ago)
                store_a(cur_cb);
                issue_a(cur_cb + 1);
                SG_SYNC();
            }
        }
        const char* Ab = Ap + fa_base + tapx_pref;
        tapx_pref = ttab[cur_tap + 1 == ntaps ? 0 : cur_tap + 1].x;
        const char* Bb = Bs + ((S + 1) & 1) * BN * 128 + fb_row;
        read_half(std::integral_constant<int, 0>{}, Ab, Bb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        read_half(std::integral_constant<int, 1>{}, Ab, Bb);
        store_b(std::integral_constant<int, (S + 2) % NSET>{});
        issue_b(std::integral_constant<int, (S + 2) % NSET>{});
        next_b_addrs();
        SG_SYNC();
    };
    auto prefetch = [&](auto K_) { next_b_addrs(); issue_b(K_); };
    auto maybe = [&](auto K_, int u) { if (u + decltype(K_)::value < nunits) iteration(K_); };

    issue_a(0);
    prefetch(std::integral_constant<int, 0>{}); prefetch(std::integral_constant<int, 1>{});
    if constexpr (NSET == 4) { prefetch(std::integral_constant<int, 2>{}); prefetch(std::integral_constant<int, 3>{}); }
    store_a(0);
    issue_a(1);
    store_b(std::integral_constant<int, 0>{});      // step 0 -> buffer 0, step 1 -> buffer 1; their slots take steps NSET, NSET + 1
    prefetch(std::integral_constant<int, 0>{});
    store_b(std::integral_constant<int, 1>{});
    prefetch(std::integral_constant<int, 1>{});
    next_b_addrs();
    SG_SYNC();
#ifdef SG3P_DELAY_WAVE   // tools/race_stress.sh: hold one wave back here, the way a cold instruction cache would, only for longer
    if ((threadIdx.x >> 6) == SG3P_DELAY_WAVE) { for (int i = 0; i < 4; ++i) asm volatile("s_sleep 127" ::: "memory"); }
    __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads behind the sleep
#endif
    {
        const char* Ab = Ap + fa_base + ttab[0].x;
        read_half(std::integral_constant<int, 0>{}, Ab, Bs + fb_row);
        read_half(std::integral_constant<int, 1>{}, Ab, Bs + fb_row);
    }
    // Step 0's fragments are read HERE, not inside an iteration: iteration 0 ends by storing step 2's weight tile into the same LDS
    // buffer, and nothing but this barrier keeps a wave that runs ahead from doing so while a delayed wave (a cold instruction
    // cache, a crowded SIMD) is still reading.  Without it the full-size twostage parity run came out a few percent wrong about
    // once in a few dozen runs; every later step is covered by the barrier that ends the iteration before it.
    // (tools/race_stress.sh builds the kernel with one wave held back, with and without this barrier.)
#ifndef SG3P_NO_STEP0_BARRIER
    SG_SYNC();
#endif
    SG3P_MARK(2);
    {
        int u = 0;
        for (; u + NSET - 1 < nunits; u += NSET) {
            iteration(std::integral_constant<int, 0>{}); iteration(std::integral_constant<int, 1>{});
            if constexpr (NSET == 4) { iteration(std::integral_constant<int, 2>{}); iteration(std::integral_constant<int, 3>{}); }
        }
        maybe(std::integral_constant<int, 0>{}, u);
        if constexpr (NSET == 4) { maybe(std::integral_constant<int, 1>{}, u); maybe(std::integral_constant<int, 2>{}, u); }
    }
    SG3P_MARK(3);
    if constexpr (KW > 1) {      // the last iteration ended on a barrier: every patch / weight buffer is free; group 1 -> LDS -> group 0
        float* xch = reinterpret_cast<float*>(smem);       // [NB * 16][256]
        if (kgp == 1) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 256 + tid] = acc[0][j][r];
        }
        SG_SYNC();
        if (kgp == 1) {
            if (P.stats != nullptr) SG_SYNC();      // the barrier inside group 0's epilogue
            return;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] += xch[(j * 16 + r) * 256 + tid];
    }

    sg3_epilogue<BN, WTM, WTN, MB, NB, F16>(P, acc, red, 0, n0, wm, wn, tid, (unsigned)bid, epi_const, [&](int row) -> int64_t {
        const int py = ty0 + (row >> 3), px = tx0 + (row & 7);
        if (py >= Hp || px >= Wp) return -1;
        return (int64_t)(py * P.os + oa) * P.Wout + (px * P.os + ob);
    });
    SG3P_MARK(4);
#ifdef SG3P_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) sg3p_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int BN, int A_IT, bool PRO, bool F16, bool S2 = false, int KW = 1>
__global__ __launch_bounds__(256 * KW) void sg_igemm3p_kernel(const SgIgemmParams G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();
    sg_igemm3p_body<BN, A_IT, PRO, F16, S2, KW>(G, smem, blockIdx.x, gridDim.x);
}

#ifndef SG_KERNELS_ONLY      // sgan_fused.hip includes this file for the kernel bodies only
// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline int sg3_cdiv(int a, int b) { return (a + b - 1) / b; }

// 1: runs on the split-bf16 kernels; 0: not covered (the fp32 kernels serve it); < 0: asked for, covered, but a job lacks its packed weights
int sg_igemm3_eligible(const SgIgemmParams& P) {
    if (P.math != SGAN_MATH_BF16X3 || P.w_ks != 1 || (P.Ck & 7) || P.Ck < 16 || P.N < 16 || P.w_ns != P.Ck) return 0;
    // Tiny maps stay on the exact-fp32 kernels: they cost nothing (latency bound), and an InstanceNorm over a 2x2 .. 8x8 map
    // (the inner U-Net levels) divides by the standard deviation of a handful of values -- it amplifies the 5e-6 of the
    // split products by orders of magnitude where it leaves the 3e-7 of the fp32 chain inside the 1e-3 contract.
    for (int g = 0; g < P.nprob; ++g)
        if (P.q[g].Hin * P.q[g].Win < SGAN_BF16X3_MIN_PIXELS || P.q[g].Hout * P.q[g].Wout < SGAN_BF16X3_MIN_PIXELS) return 0;
    for (int g = 0; g < P.nprob; ++g)
        if (!P.q[g].wp)
            return sgan_fail(SGAN_ERR_INVALID, "SGAN_MATH_BF16X3: job %d has no w_packed copy of its weights (sgan_pack_weights)", g);
    return 1;
}

struct Sg3Tile { int BM, BN; };

// Largest tile that still gives the chip about two workgroups of four waves per CU (LDS traffic per flop falls with the
// tile: see the file header); SGAN_TILE3 = "BMxBN" overrides (tuning).
static Sg3Tile sg3_pick_tile(const SgIgemmParams& P) {
    const char* force = getenv("SGAN_TILE3");    // read per call: tests walk the tile shapes in one process
    if (force) {
        int bm = 0, bn = 0;
        if (sscanf(force, "%dx%d", &bm, &bn) == 2) {
            if (bm == 128 && bn == 128 && P.N > 64) return {128, 128};
            if (bm == 128 && bn == 64 && P.N > 32) return {128, 64};
            if (bm == 64 && bn == 64 && P.N > 32) return {64, 64};
        }
    }
    if (P.N <= 32) return {128, 32};
    static const long want = getenv("SGAN_TILE3_WANT") ? atol(getenv("SGAN_TILE3_WANT")) : 384;
    if (P.N > 64 && sg_total_tiles(P, 128) * sg3_cdiv(P.N, 128) >= want / 2) return {128, 128};   // eight waves per workgroup
    if (sg_total_tiles(P, 128) * sg3_cdiv(P.N, 64) >= want) return {128, 64};
    return {64, 64};
}

template <int BM, int BN, int WGM, int WGN, bool KB2 = false>
static int sg3_launch(SgIgemmParams& P, hipStream_t st, float* ws, int64_t ws_bytes, const char* name) {
    constexpr int NT = 64 * WGM * WGN;
    const int tiles = sg_fill_tiles(P, BM);
    if (tiles == 0) return SGAN_OK;
    int ks = sg_plan_ksplit(P, BM, BN);
    const int64_t slab = (int64_t)P.q[0].Hout * P.q[0].Wout * P.N;
    if (ks > 1 && (!ws || ws_bytes < (int64_t)ks * slab * 4)) ks = 1;   // no workspace: unsplit (still correct)
    P.ksplit = ks;
    P.slab = ks > 1 ? ws : nullptr;
    P.slab_stride = slab;
    dim3 grid(tiles * sg3_cdiv(P.N, BN), 1, ks);
    const size_t lds = (size_t)(KB2 ? 4 : 2) * (BM + BN) * 128 + (size_t)4 * BN * 4 + SGAN_MAX_TAPS * 16 + (size_t)2 * P.Ck * 4;
    if (lds > 160 * 1024) return sgan_fail(SGAN_ERR_UNSUPPORTED, "LDS %zu too large", lds);
    bool pro = P.pro_act != SGAN_ACT_NONE;
    for (int g = 0; g < P.nprob; ++g) pro = pro || P.q[g].pro_stats != nullptr;
    sg_prof_begin(st);
    if (P.planes_f16) {
        if (pro) hipLaunchKernelGGL((sg_igemm3_kernel<BM, BN, WGM, WGN, true, true, KB2>), grid, dim3(NT), lds, st, P);
        else hipLaunchKernelGGL((sg_igemm3_kernel<BM, BN, WGM, WGN, false, true, KB2>), grid, dim3(NT), lds, st, P);
    } else {
        if (pro) hipLaunchKernelGGL((sg_igemm3_kernel<BM, BN, WGM, WGN, true, false, KB2>), grid, dim3(NT), lds, st, P);
        else hipLaunchKernelGGL((sg_igemm3_kernel<BM, BN, WGM, WGN, false, false, KB2>), grid, dim3(NT), lds, st, P);
    }
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = name;
    sg_prof_end(st, g_sgan_last_kernel);
    if (ks > 1) return sg_launch_splitk_epilogue(P, st);
    return SGAN_OK;
}

// ---- patch-stationary kernel: plan (patch extents per phase) and launch ----
struct Sg3pPlan { bool ok; int a_it; long tiles; double waste; int min_taps; bool s2; long wgs; };

static Sg3pPlan sg3p_plan(SgIgemmParams& P) {
    Sg3pPlan pl = {false, 0, 0, 1.0, 1 << 30, false, 0};
    static const int n32 = getenv("SGAN_PATCH_N32") ? atoi(getenv("SGAN_PATCH_N32")) : 0;      // tuning knob: 32 result channels on the 64-wide tile
    if ((P.is != 1 && !(P.is == 2 && P.nphase == 1)) || (P.Ck & 31) || P.N < 32 || (P.N == 32 && !n32)) return pl;
    static const int no_s2 = getenv("SGAN_NO_PATCH_S2") ? atoi(getenv("SGAN_NO_PATCH_S2")) : 0;      // tuning knob
    if (P.is == 2 && no_s2) return pl;
    int maxpix = 0;
    for (int ph = 0; ph < P.nphase; ++ph) {
        int dy0 = 1 << 30, dy1 = -(1 << 30), dx0 = 1 << 30, dx1 = -(1 << 30);
        if (P.ntaps[ph] <= 0) return pl;
        for (int t = 0; t < P.ntaps[ph]; ++t) {
            const SgTap& tp = P.taps[P.tap0[ph] + t];
            dy0 = min(dy0, (int)tp.dy); dy1 = max(dy1, (int)tp.dy);
            dx0 = min(dx0, (int)tp.dx); dx1 = max(dx1, (int)tp.dx);
        }
        P.pdy0[ph] = dy0; P.pdx0[ph] = dx0;
        P.pph[ph] = 7 * P.is + 1 + dy1 - dy0; P.ppw[ph] = 7 * P.is + 1 + dx1 - dx0;
        maxpix = max(maxpix, P.pph[ph] * P.ppw[ph]);
        pl.min_taps = min(pl.min_taps, P.ntaps[ph]);
    }
    if (maxpix > (P.is == 2 ? 384 : 256)) return pl;
    pl.a_it = P.is == 2 ? 6 : (maxpix <= 128 ? 2 : 4);
    long padded = 0, real = 0;
    for (int g = 0; g < P.nprob; ++g)
        for (int ph = 0; ph < P.nphase; ++ph) {
            const long t = (long)sg3_cdiv(P.q[g].Hp[ph], 8) * sg3_cdiv(P.q[g].Wp[ph], 8);
            pl.tiles += t;
            padded += t * 64;
            real += (long)P.q[g].Hp[ph] * P.q[g].Wp[ph];
        }
    if (real == 0) return pl;
    pl.waste = (double)padded / (double)real;
    pl.s2 = P.is == 2;
    pl.wgs = pl.tiles * sg3_cdiv(P.N, 64);
    pl.ok = true;
    return pl;
}

// LDS bytes of the patch image of phase ph (stride 2: four parity planes)
static int sg3p_patch_lds(const SgIgemmParams& P, int ph) {
    if (P.is == 2) return (4 * ((P.pph[ph] + 1) / 2) * sg3p_row_stride((P.ppw[ph] + 1) / 2) + 255) & ~255;
    return (P.pph[ph] * sg3p_row_stride(P.ppw[ph]) + 255) & ~255;
}

// The patch kernel stages each input element once per channel block instead of once per tap: it wins where a result pixel has many
// taps (4x4 stride 1: 16) and the 8 x 8 tiles do not waste much of the map; SGAN_IGEMM3P = 0 / 1 forces the choice (tuning, tests).
static bool sg3p_wanted(const Sg3pPlan& pl) {
    if (!pl.ok) return false;
    const char* force = getenv("SGAN_IGEMM3P");
    if (force) return atoi(force) != 0;
    // stride 2: only where the grid alone fills a good part of the chip -- the deep, narrow generator layers (16 .. 64 workgroups of
    // 64 .. 128 steps) stay on sg_igemm3_kernel, which can split K across workgroups
    static const long s2_min = getenv("SGAN_PATCH_S2_MIN") ? atol(getenv("SGAN_PATCH_S2_MIN")) : 100;
    if (pl.s2 && pl.wgs < s2_min) return false;
    if (pl.min_taps >= 9) return pl.waste < 1.6;
    return pl.min_taps >= 4 && pl.waste < 1.3;
}

template <int BN, int A_IT, bool S2 = false>
static int sg3p_launch(SgIgemmParams& P, hipStream_t st, const char* name) {   // BN: 64, or 128 (wave tile 32 x 64) for wide results
    int t = 0, maxlds = 0;
    for (int g = 0; g < P.nprob; ++g)
        for (int ph = 0; ph < P.nphase; ++ph) {
            P.q[g].tile0[ph] = t;
            t += sg3_cdiv(P.q[g].Hp[ph], 8) * sg3_cdiv(P.q[g].Wp[ph], 8);
        }
    for (int g = 0; g < P.nprob; ++g)
        for (int ph = P.nphase; ph < SGAN_MAX_PHASES; ++ph) P.q[g].tile0[ph] = 1 << 30;
    if (t == 0) return SGAN_OK;
    for (int ph = 0; ph < P.nphase; ++ph) maxlds = max(maxlds, sg3p_patch_lds(P, ph));
    P.ksplit = 1;
    P.slab = nullptr;
    P.slab_stride = 0;
    dim3 grid(t * sg3_cdiv(P.N, BN), 1, 1);
    // two wave groups per workgroup (the channel blocks split between them) where the grid leaves most SIMDs with one wave or none
    static const int kw_max_wgs = getenv("SGAN_PATCH_KW_MAX") ? atoi(getenv("SGAN_PATCH_KW_MAX")) : 256;      // tuning knob; 0 = never
    const int ncb = P.Ck >> 5;
    const bool kw2 = !S2 && A_IT == 2 && BN == 64 && (long)grid.x <= kw_max_wgs && (ncb & 1) == 0 && ncb >= 2;
    const int KWr = kw2 ? 2 : 1;
    const size_t lds = (size_t)KWr * ((size_t)maxlds + (size_t)2 * BN * 128) + (size_t)4 * BN * 4 + SGAN_MAX_TAPS * 16 + (size_t)2 * P.Ck * 4;
    if (lds > 160 * 1024) return sgan_fail(SGAN_ERR_UNSUPPORTED, "LDS %zu too large", lds);
    bool pro = P.pro_act != SGAN_ACT_NONE;
    for (int g = 0; g < P.nprob; ++g) pro = pro || P.q[g].pro_stats != nullptr;
    sg_prof_begin(st);
    if constexpr (!S2 && A_IT == 2 && BN == 64) {
        if (kw2) {
            if (P.planes_f16) {
                if (pro) hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, true, true, false, 2>), grid, dim3(512), lds, st, P);
                else hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, false, true, false, 2>), grid, dim3(512), lds, st, P);
            } else {
                if (pro) hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, true, false, false, 2>), grid, dim3(512), lds, st, P);
                else hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, false, false, false, 2>), grid, dim3(512), lds, st, P);
            }
        }
    }
    if (kw2) { /* launched above */ }
    else if (P.planes_f16) {
        if (pro) hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, true, true, S2>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, false, true, S2>), grid, dim3(256), lds, st, P);
    } else {
        if (pro) hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, true, false, S2>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((sg_igemm3p_kernel<BN, A_IT, false, false, S2>), grid, dim3(256), lds, st, P);
    }
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = name;
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

// Plan of a backward-data launch for sg_bwd_fused_kernel (sgan_fused.hip): fills the tile tables of P exactly as the launchers
// below do and says which body runs it.  variant 0: not one of the bodies the fused kernel carries (the caller launches separately).
int sg_igemm3_fuse_plan(SgIgemmParams& P, SgFusePlan* out) {
    out->variant = 0;
    out->ks = 1;
    if (P.pro_act != SGAN_ACT_NONE) return 0;
    for (int g = 0; g < P.nprob; ++g)
        if (P.q[g].pro_stats) return 0;
    P.ksplit = 1;
    P.slab = nullptr;
    P.slab_stride = 0;
    const Sg3pPlan pl = sg3p_plan(P);
    if (sg3p_wanted(pl)) {
        int t = 0, maxlds = 0;
        for (int g = 0; g < P.nprob; ++g)
            for (int ph = 0; ph < P.nphase; ++ph) {
                P.q[g].tile0[ph] = t;
                t += sg3_cdiv(P.q[g].Hp[ph], 8) * sg3_cdiv(P.q[g].Wp[ph], 8);
            }
        for (int g = 0; g < P.nprob; ++g)
            for (int ph = P.nphase; ph < SGAN_MAX_PHASES; ++ph) P.q[g].tile0[ph] = 1 << 30;
        for (int ph = 0; ph < P.nphase; ++ph) maxlds = max(maxlds, sg3p_patch_lds(P, ph));
        out->variant = pl.a_it == 2 ? 1 : (pl.a_it == 4 ? 2 : 5);
        out->nblocks = t * sg3_cdiv(P.N, 64);
        out->lds = (size_t)maxlds + (size_t)2 * 64 * 128 + (size_t)4 * 64 * 4 + SGAN_MAX_TAPS * 16 + (size_t)2 * P.Ck * 4;
        out->name = pl.a_it == 6 ? "sg_igemm3p_kernel<64,s2>" : "sg_igemm3p_kernel<64>";
        return 0;
    }
    const Sg3Tile tl = sg3_pick_tile(P);
    if (tl.BM == 128 && tl.BN == 32 && sg_plan_ksplit(P, 128, 32) == 1) {      // <= 32 result channels (the first PatchGAN layer's input gradient)
        const int tiles = sg_fill_tiles(P, 128);
        out->variant = 6;
        out->nblocks = tiles * sg3_cdiv(P.N, 32);
        out->lds = (size_t)2 * (128 + 32) * 128 + (size_t)4 * 32 * 4 + SGAN_MAX_TAPS * 16 + (size_t)2 * P.Ck * 4;
        out->name = "sg_igemm3_kernel<128,32,4,1>";
        return 0;
    }
    // a launch that would have been split-K on its own runs unsplit here when the split is shallow: the backward-weight
    // workgroups of the same grid fill the CUs the split was there to fill (SGAN_FUSE_MAX_KS: tuning knob)
    // A launch that would have been split-K on its own: the split stays (its slabs go to the caller's workspace and
    // sg_splitk_epilogue_kernel finishes them after the fused launch) -- the deep, narrow layers (generator 256 -> 128 at 32 x 32,
    // the inner U-Net levels) are exactly the ones whose two halves leave the chip idle one after the other.  SGAN_FUSE_MAX_KS caps the
    // split a fused launch accepts (0: only unsplit launches fuse, as in round 2).
    static const int max_ks = getenv("SGAN_FUSE_MAX_KS") ? atoi(getenv("SGAN_FUSE_MAX_KS")) : 64;
    if (tl.BM != 64 || tl.BN != 64) return 0;
    int ks = sg_plan_ksplit(P, 64, 64);
    if (ks > 1 && ks > max_ks) return 0;
    const int tiles = sg_fill_tiles(P, 64);
    if (ks > 1) {      // beside the backward-weight workgroups ~128 backward-data workgroups are enough (measured on the generator's two deep
        // layers: 256 -> 128, 64 tiles: split 2 28.6 us, split 8 34 us; 256 -> 256, 16 tiles: split 8 23.6 us, split 32 31.5 us, split 2 34.9 us)
        static const int want = getenv("SGAN_FUSE_KS_WGS") ? atoi(getenv("SGAN_FUSE_KS_WGS")) : 128;
        const int cap = max(1, want / max(1, tiles * sg3_cdiv(P.N, 64)));
        if (ks > cap) {
            const int nkt = sg3_cdiv(sg_max_k(P), 32);
            const int per = sg3_cdiv(nkt, cap);
            ks = sg3_cdiv(nkt, per);      // every split non-empty
        }
    }
    out->variant = 3;
    out->ks = ks;
    out->nblocks = tiles * sg3_cdiv(P.N, 64) * ks;
    out->lds = (size_t)4 * (64 + 64) * 128 + (size_t)4 * 64 * 4 + SGAN_MAX_TAPS * 16 + (size_t)2 * P.Ck * 4;
    out->name = "sg_igemm3_kernel<64,64,2,2>";
    return 0;
}

int sg_launch_igemm3(SgIgemmParams& P, hipStream_t st, float* ws, int64_t ws_bytes) {
    const Sg3pPlan pl = sg3p_plan(P);
    if (sg3p_wanted(pl)) {
        if (pl.a_it == 2) return sg3p_launch<64, 2>(P, st, "sg_igemm3p_kernel<64>");
        if (pl.a_it == 6) return sg3p_launch<64, 6, true>(P, st, "sg_igemm3p_kernel<64,s2>");
        return sg3p_launch<64, 4>(P, st, "sg_igemm3p_kernel<64>");
    }
    const Sg3Tile t = sg3_pick_tile(P);
    if (t.BN == 32) return sg3_launch<128, 32, 4, 1>(P, st, ws, ws_bytes, "sg_igemm3_kernel<128,32,4,1>");
    if (t.BM == 128 && t.BN == 128) return sg3_launch<128, 128, 2, 4>(P, st, ws, ws_bytes, "sg_igemm3_kernel<128,128,2,4>");
    if (t.BM == 128) return sg3_launch<128, 64, 2, 2>(P, st, ws, ws_bytes, "sg_igemm3_kernel<128,64,2,2>");
    static const int kb2 = getenv("SGAN_KB2") ? atoi(getenv("SGAN_KB2")) : 1;      // tuning knob
    if (kb2) return sg3_launch<64, 64, 2, 2, true>(P, st, ws, ws_bytes, "sg_igemm3_kernel<64,64,2,2>");
    return sg3_launch<64, 64, 2, 2>(P, st, ws, ws_bytes, "sg_igemm3_kernel<64,64,2,2>");
}

int64_t sg_igemm3_workspace_need(const SgIgemmParams& P) {
    SgIgemmParams Q = P;
    if (sg3p_wanted(sg3p_plan(Q))) return 0;
    const Sg3Tile t = sg3_pick_tile(P);
    const int ks = sg_plan_ksplit(P, t.BM, t.BN);
    return ks > 1 ? (int64_t)ks * P.q[0].Hout * P.q[0].Wout * P.N * 4 : 0;
}
#endif      // SG_KERNELS_ONLY
