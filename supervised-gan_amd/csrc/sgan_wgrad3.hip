// Conv2d / ConvTranspose2d backward-weight (+ bias) for gfx950 (CDNA4), split-bf16 arithmetic (SGAN_MATH_BF16X3).
//
//   dW[co][kcol] += sum_pixels dOut[pixel][co] * Acol[pixel][kcol],  kcol = (tap, ci)
//
// Same decomposition as sg_wgrad_kernel (sgan_wgrad.hip: on-the-fly im2col with the producer's norm + activation applied while
// staging, reduction over 32-pixel chunks, split over pixel ranges, fp32 atomics into the gradient buffer); the arithmetic is
// that of sgan_igemm3.hip: every fp32 operand is cut into hi = bf16(x), lo = bf16(x - hi) while it is staged and a product is
// d_hi * a_hi + d_hi * a_lo + d_lo * a_hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// The reduction index is the PIXEL, and both operands arrive pixel-major ([pixel][channel]), while an MFMA lane wants 8
// consecutive k of one row: the tiles are therefore stored untransposed, as [32-channel block][32 pixels][32 channels] bf16
// (64-byte rows, one image per plane), and read with ds_read_b64_tr_b16 -- the LDS hardware hands every lane its column of a
// 4 x 16 block, i.e. the transpose happens in the read (two reads per 8-element fragment; conflict free: the four rows of a
// block are the four 64-byte quarters of a 256-byte bank row).  Stores are 16 bytes (8 channels of one pixel, one plane) with
// eight consecutive lanes covering two pixel rows of one channel block = 128 contiguous bytes.
//
// Reference ops replaced: convolution_backward (weight, bias) of every nn.Conv2d / nn.ConvTranspose2d on the path
// (models/networks.py:502-529, :815-835).
#include <type_traits>

#include "sgan_wgrad.h"

typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sg_bf16x2 __attribute__((ext_vector_type(2)));
typedef short sg_s16x4 __attribute__((ext_vector_type(4)));
typedef short sg_s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 sgw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sgw_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ void sgw_split8(const f32x4 v0, const f32x4 v1, u32x4& hi, u32x4& lo) {   // as sg_split8 (sgan_igemm3.hip)
    float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 p = {x[2 * i], x[2 * i + 1]};
        if constexpr (F16) {
            const sgw_f16x2 h = __builtin_convertvector(p, sgw_f16x2);
            const f32x2 r = p - __builtin_convertvector(h, f32x2);
            hi[i] = __builtin_bit_cast(unsigned, h);
            lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sgw_f16x2));
        } else {
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(p, sg_bf16x2));
            const f32x2 r = {x[2 * i] - __builtin_bit_cast(float, h << 16), x[2 * i + 1] - __builtin_bit_cast(float, h & 0xffff0000u)};
            hi[i] = h;
            lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, sg_bf16x2));
        }
    }
}
template <bool F16>
__device__ __forceinline__ f32x16 sgw_mfma(const sg_bf16x8 a, const sg_bf16x8 b, const f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sgw_f16x8, a), __builtin_bit_cast(sgw_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 8 consecutive k (pixels) of one column: two transposed LDS reads of 4 each
__device__ __forceinline__ sg_bf16x8 sgw_tr8(const char* p) {
    typedef __attribute__((address_space(3))) sg_s16x4 lds_s16x4;
    const sg_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const sg_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 64));
    const sg_s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(sg_bf16x8, v);
}

// BCO x BKC tile of dW per workgroup, WGC x WGK waves of (BCO / WGC) x (BKC / WGK) each (multiples of 32)
// body with explicit workgroup coordinates: sg_bwd_fused_kernel (sgan_fused.hip) runs it beside the backward-data body in one launch
// F16: fp16 planes (dout times 2^s, s from its published maximum; x is a post-normalisation activation): 11 + 11 significant bits, an
// fp32-equivalent product; else bf16 planes (8 + 8)
template <int BCO, int BKC, int WGC, int WGK, bool PRO, bool F16 = false>
__device__ __forceinline__ void sg_wgrad3_body(const SgWgradParams& G, char* smem, const int bx, const int by, const int bz) {
    constexpr int BP = 32;
    constexpr int WTC = BCO / WGC, WTK = BKC / WGK, MB = WTC / 32, NB = WTK / 32;
    // staging tasks (8 channels of one pixel) per thread: the 256 threads cover two 32-channel blocks of the 32 pixel rows per pass
    constexpr int D_IT = (BCO / 32 + 1) / 2, A_IT = (BKC / 32 + 1) / 2;
    constexpr int D_PLANE = BCO / 32 * 2048, A_PLANE = BKC / 32 * 2048;     // bytes of one plane of one operand
    static_assert(WGC * WGK == 4 && WTC % 32 == 0 && WTK % 32 == 0, "4 waves of 32 x 32 blocks");

    char* Ds = smem;                                    // [2 buffers][2 planes][D_PLANE]
    char* As = smem + 4 * D_PLANE;                      // [2 buffers][2 planes][A_PLANE]
    float* pscale = reinterpret_cast<float*>(smem + 4 * D_PLANE + 4 * A_PLANE);   // [Cin]
    float* pshift = pscale + G.Cin;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wc = wid / WGK, wk = wid % WGK;
    int g = 0;
    for (int gi = 1; gi < G.nprob; ++gi)
        if (bz >= G.q[gi].z0) g = gi;
    const SgWgradProb& Q = G.q[g];
    SgWgradLocal P;
    P.in = Q.in; P.dout = Q.dout; P.dw = Q.dw; P.dbias = Q.dbias;
    P.Hin = Q.Hin; P.Win = Q.Win; P.Cin = G.Cin; P.in_ld = Q.in_ld; P.Hout = Q.Hout; P.Wout = Q.Wout; P.Cout = G.Cout;
    P.dout_ld = Q.dout_ld; P.is = G.is; P.os = G.os; P.w_ns = G.w_ns; P.nsplit = Q.nsplit;
    P.pro.stats = Q.pro_stats; P.pro.gamma = Q.pro_gamma; P.pro.beta = Q.pro_beta; P.pro.count = Q.pro_count;
    P.pro.eps = G.pro_eps; P.pro.act = G.pro_act; P.pro.slope = G.pro_slope; P.pro.sq_stride = Q.pro_sq; P.pro.rep_stride = Q.pro_rep;
    const int zl = bz - Q.z0;
    const int phz = zl / P.nsplit, split = zl % P.nsplit;
    const int Hp = Q.Hp[phz], Wp = Q.Wp[phz], M = Hp * Wp, ktot = G.ktot[phz];
    const int ph_oa = G.oa[phz], ph_ob = G.ob[phz];
    const int kc0 = bx * BKC, co0 = by * BCO;
    if (kc0 >= ktot || M == 0) return;
    const int nchunk_total = (M + BP - 1) / BP;
    const int per = (nchunk_total + P.nsplit - 1) / P.nsplit;
    const int ch_begin = split * per, ch_end = min(nchunk_total, ch_begin + per);
    if (ch_begin >= ch_end) return;
    const int Cin = P.Cin, Cout = P.Cout;
    const int f16_shift = (F16 && Q.amax) ? sg_f16_shift(*Q.amax) : 0;
    const float d_scale = sg_pow2(f16_shift), out_scale = sg_pow2(-f16_shift);
    if constexpr (PRO) {
        for (int c = tid; c < Cin; c += 256) {
            float sc = 1.f, sh = 0.f;
            if (P.pro.stats) {
                float mean, rstd;
                sg_mean_rstd(P.pro, Cin, c, mean, rstd);
                const float gm = P.pro.gamma ? P.pro.gamma[c] : 1.f;
                const float bt = P.pro.beta ? P.pro.beta[c] : 0.f;
                sc = gm * rstd;
                sh = bt - mean * sc;
            }
            pscale[c] = sc;
            pshift[c] = sh;
        }
    }

    // Staging task e -> (channel block e >> 7, pixel (e >> 2) & 31, 16-byte chunk e & 3 of the pixel's 64-byte row): eight
    // consecutive lanes store two pixel rows of one channel block = 128 contiguous bytes.
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.dout), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;
    const int adv_y = BP / Wp, adv_x = BP - adv_y * Wp;
    const int s_pix = (tid >> 2) & 31, s_q = tid & 3;
    const int s_dst = s_pix * 64 + s_q * 16;
    // D side
    int d_blk[D_IT];
    bool d_cok[D_IT];
    int d_py, d_px, d_lin;
    {
        const int m = ch_begin * BP + s_pix;
        d_py = m / Wp;
        d_px = m - d_py * Wp;
        d_lin = ((d_py * P.os + ph_oa) * P.Wout + (d_px * P.os + ph_ob)) * P.dout_ld + co0 + s_q * 8;
    }
#pragma unroll
    for (int it = 0; it < D_IT; ++it) {
        d_blk[it] = (tid >> 7) + 2 * it;
        d_cok[it] = (d_blk[it] < BCO / 32) && co0 + d_blk[it] * 32 + s_q * 8 < Cout;
    }
    // A side: every task of this thread has its own (tap, channel) column group, fixed for the whole kernel
    int a_py, a_px;
    int a_blk[A_IT], a_c[A_IT], a_dy[A_IT], a_dx[A_IT], a_iy[A_IT], a_ix[A_IT], a_lin[A_IT];
    bool a_kok[A_IT];
    {
        const int m = ch_begin * BP + s_pix;
        a_py = m / Wp;
        a_px = m - a_py * Wp;
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        a_blk[it] = (tid >> 7) + 2 * it;
        const int kcol = kc0 + a_blk[it] * 32 + s_q * 8;
        a_kok[it] = (a_blk[it] < BKC / 32) && kcol < ktot;
        const int tap = a_kok[it] ? kcol / Cin : 0;
        a_c[it] = a_kok[it] ? kcol - tap * Cin : 0;
        a_dy[it] = G.taps[G.tap0[phz] + tap].dy;
        a_dx[it] = G.taps[G.tap0[phz] + tap].dx;
        a_iy[it] = a_py * P.is + a_dy[it];
        a_ix[it] = a_px * P.is + a_dx[it];
        a_lin[it] = (a_iy[it] * P.Win + a_ix[it]) * P.in_ld + a_c[it];
    }
    const int a_iy_step = adv_y * P.is, a_ix_step = adv_x * P.is, a_ix_wrap = -Wp * P.is;
    const int a_lin_step = (a_iy_step * P.Win + a_ix_step) * P.in_ld, a_lin_wrap = (P.is * P.Win + a_ix_wrap) * P.in_ld;
    const int d_lin_step = (adv_y * P.os * P.Wout + adv_x * P.os) * P.dout_ld, d_lin_wrap = (P.os * P.Wout - Wp * P.os) * P.dout_ld;
    int ch_next = ch_begin;

    constexpr int NS = 3;     // register ring (see sgan_wgrad.hip)
    f32x4 a_reg[NS][A_IT][2], d_reg[NS][D_IT][2];
    bool a_val[NS][A_IT];
    int a_off_n[A_IT], d_off_n[D_IT];
    bool a_ok_n[A_IT];

    auto next_addrs = [&]() {
        const bool chok = ch_next < ch_end;   // chunks past this split's range belong to another workgroup
        const bool rok = chok & (a_py < Hp);  // pixel row of this thread inside the phase grid (same pixel for D and A)
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool ok = rok & a_kok[it] & ((unsigned)a_iy[it] < (unsigned)P.Hin) & ((unsigned)a_ix[it] < (unsigned)P.Win);
            a_off_n[it] = ok ? a_lin[it] << 2 : OOB;
            if constexpr (PRO) a_ok_n[it] = ok;
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) d_off_n[it] = (rok & d_cok[it]) ? (d_lin + d_blk[it] * 32) << 2 : OOB;
        a_px += adv_x;
        const bool wrap = a_px >= Wp;
        a_px -= wrap ? Wp : 0;
        a_py += adv_y + (wrap ? 1 : 0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            a_iy[it] += a_iy_step + (wrap ? P.is : 0);
            a_ix[it] += a_ix_step + (wrap ? a_ix_wrap : 0);
            a_lin[it] += a_lin_step + (wrap ? a_lin_wrap : 0);
        }
        d_lin += d_lin_step + (wrap ? d_lin_wrap : 0);
        ++ch_next;
    };

    auto issue_loads = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            a_reg[S][it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off_n[it], 0, 0));
            a_reg[S][it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off_n[it] + 16, 0, 0));
            if constexpr (PRO) a_val[S][it] = a_ok_n[it];
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            d_reg[S][it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, d_off_n[it], 0, 0));
            d_reg[S][it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, d_off_n[it] + 16, 0, 0));
        }
    };

    f32x4 bacc[D_IT][2];     // bias gradient: this thread's 8 channels summed over the pixel rows it stages (fp32, before the split)
#pragma unroll
    for (int it = 0; it < D_IT; ++it) bacc[it][0] = bacc[it][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto store_chunk = [&](auto S_, int buf) {
        constexpr int S = decltype(S_)::value;
        char* Ab = As + buf * 2 * A_PLANE;
        char* Db = Ds + buf * 2 * D_PLANE;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            f32x4 v0 = a_reg[S][it][0], v1 = a_reg[S][it][1];
            if constexpr (PRO) {
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(pscale + a_c[it]), sc1 = *reinterpret_cast<const f32x4*>(pscale + a_c[it] + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(pshift + a_c[it]), sh1 = *reinterpret_cast<const f32x4*>(pshift + a_c[it] + 4);
                const float okf = a_val[S][it] ? 1.f : 0.f;     // zero padding applies AFTER norm + activation (see sgan_igemm.hip)
                const float okn = okf * pro_neg;
                const f32x4 y0 = v0 * sc0 + sh0, y1 = v1 * sc1 + sh1;
                const f32x4 p0 = y0 * okf, q0 = y0 * okn, p1 = y1 * okf, q1 = y1 * okn;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(p0[j], q0[j]); v1[j] = fmaxf(p1[j], q1[j]); }
            }
            u32x4 hi, lo;
            sgw_split8<F16>(v0, v1, hi, lo);
            if (BKC / 32 % 2 == 0 || a_blk[it] < BKC / 32) {
                *reinterpret_cast<u32x4*>(Ab + a_blk[it] * 2048 + s_dst) = hi;
                *reinterpret_cast<u32x4*>(Ab + A_PLANE + a_blk[it] * 2048 + s_dst) = lo;
            }
        }
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            bacc[it][0] += d_reg[S][it][0];
            bacc[it][1] += d_reg[S][it][1];
            u32x4 hi, lo;
            if constexpr (F16) sgw_split8<true>(d_reg[S][it][0] * d_scale, d_reg[S][it][1] * d_scale, hi, lo);
            else sgw_split8<false>(d_reg[S][it][0], d_reg[S][it][1], hi, lo);
            if (BCO / 32 % 2 == 0 || d_blk[it] < BCO / 32) {
                *reinterpret_cast<u32x4*>(Db + d_blk[it] * 2048 + s_dst) = hi;
                *reinterpret_cast<u32x4*>(Db + D_PLANE + d_blk[it] * 2048 + s_dst) = lo;
            }
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_bias = (P.dbias != nullptr) && (bx == 0);

    // transposed fragment reads: lane L = 16 G + 4 q + p4 supplies the address of pixel row 8 (G >> 1) + q (+ 4 for the second
    // read, + 16 for the second k16 step), columns 16 (G & 1) + 4 p4 .. + 3 of its 32-channel block
    const int fG = lane >> 4, fq = (lane >> 2) & 3, fp4 = lane & 3;
    const int f_lane = (8 * (fG >> 1) + fq) * 64 + (16 * (fG & 1) + 4 * fp4) * 2;

    SG_SYNC();  // pscale / pshift visible
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    int it_no = 0;
    auto iteration = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        issue_loads(std::integral_constant<int, S>{});   // chunk it_no + NS
        const int buf = it_no & 1;
        const char* Ab = As + buf * 2 * A_PLANE + wk * (WTK / 32) * 2048 + f_lane;
        const char* Db = Ds + buf * 2 * D_PLANE + wc * (WTC / 32) * 2048 + f_lane;
        sg_bf16x8 dh[2][MB], dl[2][MB], ah[2][NB], al[2][NB];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                dh[s][i] = sgw_tr8(Db + i * 2048 + s * 16 * 64);
                dl[s][i] = sgw_tr8(Db + D_PLANE + i * 2048 + s * 16 * 64);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                ah[s][j] = sgw_tr8(Ab + j * 2048 + s * 16 * 64);
                al[s][j] = sgw_tr8(Ab + A_PLANE + j * 2048 + s * 16 * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(std::integral_constant<int, (S + 1) % NS>{}, buf ^ 1);   // chunk it_no + 1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i][j] = sgw_mfma<F16>(dl[s][i], ah[s][j], acc[i][j]);
                    acc[i][j] = sgw_mfma<F16>(dh[s][i], al[s][j], acc[i][j]);
                    acc[i][j] = sgw_mfma<F16>(dh[s][i], ah[s][j], acc[i][j]);
                }
        next_addrs();
        constexpr int NMFMA = 6 * MB * NB;
        constexpr int PER = (24 + 8 * A_IT + 2 * D_IT + NMFMA - 1) / NMFMA;
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, PER, 0);
        }
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

    // ---- combine: fp32 atomics into the gradient buffer.  acc[i][j][r] = dW[co = co0 + wc*WTC + i*32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)]
    //      [kcol = kc0 + wk*WTK + j*32 + (lane & 31)]: a wave instruction adds to 32 consecutive ci of two weight rows ----
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int kcol = kc0 + wk * WTK + j * 32 + fr;
        if (kcol >= ktot) continue;
        const int tap = kcol / Cin, ci = kcol - tap * Cin;
        float* base = P.dw + G.taps[G.tap0[phz] + tap].w_off + ci;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wc * WTC + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#ifndef SGW3_NO_ATOMICS      // diagnostics build: what the gradient atomics cost
                if (co < Cout) atomicAdd(base + (int64_t)co * P.w_ns, F16 ? acc[i][j][r] * out_scale : acc[i][j][r]);
#else
                if (co < Cout && acc[i][j][r] == 12345.678f) base[(int64_t)co * P.w_ns] = 0.f;
#endif
            }
        }
    }
    if (do_bias) {   // (uniform) threads with equal (channel block, chunk) -- tid & 0x83 -- share 8 channels: sum over their 32 pixel rows
        SG_SYNC();   // every wave is past its last fragment read: the staging buffers are free
        f32x4* red = reinterpret_cast<f32x4*>(smem);   // [D_IT][2][256]
#pragma unroll
        for (int it = 0; it < D_IT; ++it) {
            red[(it * 2 + 0) * 256 + tid] = bacc[it][0];
            red[(it * 2 + 1) * 256 + tid] = bacc[it][1];
        }
        SG_SYNC();
        if (tid < BCO) {
            const int blk = tid >> 5, cq = (tid & 31) >> 3, half = (tid >> 2) & 1, e = tid & 3;   // channel tid of the tile
            const int it = blk >> 1, t0 = ((blk & 1) << 7) + cq;
            float sb = 0.f;
#pragma unroll 8
            for (int p = 0; p < 32; ++p) sb += reinterpret_cast<const float*>(red + (it * 2 + half) * 256 + t0 + 4 * p)[e];
            if (co0 + tid < Cout) atomicAdd(P.dbias + co0 + tid, sb);
        }
    }
}

template <int BCO, int BKC, int WGC, int WGK, bool PRO, bool F16 = false>
__global__ __launch_bounds__(256) void sg_wgrad3_kernel(const SgWgradParams G) {
    sg_warm_kernargs<(int)sizeof(SgWgradParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_wgrad3_body<BCO, BKC, WGC, WGK, PRO, F16>(G, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

#ifndef SG_KERNELS_ONLY      // sgan_fused.hip includes this file for the kernel bodies only
static inline int sgw3_cdiv(int a, int b) { return (a + b - 1) / b; }

// pixel splits, z offsets, grid and LDS bytes of a launch with BCO x BKC tiles; false: nothing to do
static bool sgw3_prepare(SgWgradParams& P, int BCO, int BKC, dim3* grid_out, size_t* lds_out, bool* pro_out) {
    int maxK = 0;
    for (int i = 0; i < P.nphase; ++i) maxK = max(maxK, P.ktot[i]);
    const int tiles = sgw3_cdiv(maxK, BKC) * sgw3_cdiv(P.Cout, BCO);
    long chunks_total = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        chunks_total += (long)sgw3_cdiv(maxM, 32) * P.nphase;
    }
    if (chunks_total == 0) return false;
    // pixel-range split per problem: ~512 workgroups over the launch (sweep 96 .. 2304 on the fcgan launches: 512 is best or within 3 % of best everywhere; 768 cost the small generator layers 10-25 %) with the same number of 32-pixel chunks each (>= 4)
    const double want = getenv("SGAN_WGRAD3_WANT") ? atof(getenv("SGAN_WGRAD3_WANT")) : 512.0;
    int per = (int)((double)chunks_total * tiles / want + 0.999);
    if (per < 4) per = 4;
    int z = 0;
    for (int g = 0; g < P.nprob; ++g) {
        int maxM = 0;
        for (int i = 0; i < P.nphase; ++i) maxM = max(maxM, P.q[g].Hp[i] * P.q[g].Wp[i]);
        const int nchunk = sgw3_cdiv(maxM, 32);
        int nsplit = sgw3_cdiv(nchunk, per);
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 512) nsplit = 512;
        P.q[g].nsplit = nsplit;
        P.q[g].z0 = z;
        z += P.nphase * nsplit;
    }
    *grid_out = dim3(sgw3_cdiv(maxK, BKC), sgw3_cdiv(P.Cout, BCO), z);
    *lds_out = (size_t)4 * (BCO / 32 * 2048) + (size_t)4 * (BKC / 32 * 2048) + (size_t)2 * P.Cin * 4;
    bool pro = P.pro_act != SGAN_ACT_NONE;
    for (int g = 0; g < P.nprob; ++g) pro = pro || P.q[g].pro_stats != nullptr;
    *pro_out = pro;
    return true;
}

template <int BCO, int BKC, int WGC, int WGK>
static int sgw3_launch(SgWgradParams& P, hipStream_t st, const char* name) {
    dim3 grid;
    size_t lds;
    bool pro;
    if (!sgw3_prepare(P, BCO, BKC, &grid, &lds, &pro)) return 1;
    sg_prof_begin(st);
    if (P.planes_f16) {
        if (pro) hipLaunchKernelGGL((sg_wgrad3_kernel<BCO, BKC, WGC, WGK, true, true>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((sg_wgrad3_kernel<BCO, BKC, WGC, WGK, false, true>), grid, dim3(256), lds, st, P);
    } else {
        if (pro) hipLaunchKernelGGL((sg_wgrad3_kernel<BCO, BKC, WGC, WGK, true, false>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((sg_wgrad3_kernel<BCO, BKC, WGC, WGK, false, false>), grid, dim3(256), lds, st, P);
    }
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return sgan_fail(SGAN_ERR_HIP, "%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_));
    g_sgan_last_kernel = name;
    sg_prof_end(st, g_sgan_last_kernel);
    return 1;
}

static bool sgw3_covers(const SgWgradParams& P) {
    if ((P.Cin & 7) || (P.Cout & 7) || P.Cin < 16 || P.Cout < 32) return false;
    for (int g = 0; g < P.nprob; ++g)     // tiny maps stay exact fp32 (see sg_igemm3_eligible)
        if (P.q[g].Hin * P.q[g].Win < SGAN_BF16X3_MIN_PIXELS || P.q[g].Hout * P.q[g].Wout < SGAN_BF16X3_MIN_PIXELS) return false;
    return true;
}

// Plan of a backward-weight launch for sg_bwd_fused_kernel: variant 1 = 64 x 64 tiles, 2 = 32 x 128, 0 = not covered / nothing to do
int sg_wgrad3_fuse_plan(SgWgradParams& P, SgFusePlan* out) {
    out->variant = 0;
    if (!sgw3_covers(P)) return 0;
    dim3 grid;
    const bool narrow = P.Cout < 64;
    if (!sgw3_prepare(P, narrow ? 32 : 64, narrow ? 128 : 64, &grid, &out->lds, &out->pro)) return 0;
    out->variant = narrow ? 2 : 1;
    out->gx = grid.x; out->gy = grid.y; out->gz = grid.z;
    out->nblocks = grid.x * grid.y * grid.z;
    out->name = narrow ? "sg_wgrad3_kernel<32,128,1,4>" : "sg_wgrad3_kernel<64,64,2,2>";
    return 0;
}

int sg_launch_wgrad3(SgWgradParams& P, hipStream_t st) {
    if (!sgw3_covers(P)) return 0;
    if (P.Cout < 64) return sgw3_launch<32, 128, 1, 4>(P, st, "sg_wgrad3_kernel<32,128,1,4>");
    return sgw3_launch<64, 64, 2, 2>(P, st, "sg_wgrad3_kernel<64,64,2,2>");
}
#endif      // SG_KERNELS_ONLY
