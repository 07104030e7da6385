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
#include <type_traits>

#include "sgan_igemm.h"

#ifndef SG_ABLATE
#define SG_ABLATE 0   // diagnostics build only: 1 = skip MFMAs, 2 = skip global loads, 4 = skip LDS stores
#endif

__device__ __forceinline__ int sg_swz(int row, int kslot) { return (kslot ^ ((row >> 1) & 7)) << 2; }

// BKC: B operand is k-contiguous in memory (forward: W[tap][n][k]); otherwise n-contiguous (backward-data)
// PRO: the gathered tensor gets the producer's norm + activation applied while it is staged
// KW:  wave groups per workgroup.  KW = 2 (512 threads): the two groups of four waves take one half each of the
//      workgroup's k-tiles through their own LDS buffers, in step (shared barriers), and group 1 hands its accumulators
//      to group 0 through LDS before the epilogue.  A grid of fewer than ~2 tiles per CU is latency bound -- one wave
//      per SIMD waits out every barrier and LDS round trip alone -- and this doubles the waves without the slab round
//      trip and second kernel of the global split-K.
template <int BM, int BN, int WGM, int WGN, bool BKC, bool PRO, int KW = 1>
__device__ __forceinline__ void sg_igemm_body(const SgIgemmParams& G, char* smem, const int bid, const int nblocks, const int split) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN, MB = WTM / 16, NB = WTN / 16;
    constexpr int A_IT = BM * 8 / 256;
    constexpr int B_IT = (BN * 8 + 255) / 256;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(BM * 8 % 256 == 0, "A tile");

    constexpr int GROUP_FLOATS = 2 * BM * 32 + 2 * BN * 32;
    const int kg = KW == 1 ? 0 : (int)(threadIdx.x >> 8);   // wave group
    float* As = reinterpret_cast<float*>(smem) + kg * GROUP_FLOATS;  // [2][BM*32]
    float* Bs = As + 2 * BM * 32;                                    // [2][BN*32]
    // statistics are accumulated in fp64 from the first add on: var = E[x^2] - mean^2 cancels catastrophically in fp32
    // partial sums as soon as |mean| >> std (conv of a near-constant map plus its bias, e.g. the CRN label branch)
    double* red = reinterpret_cast<double*>(reinterpret_cast<float*>(smem) + KW * GROUP_FLOATS);   // [2*BN]
    int4* ttab = reinterpret_cast<int4*>(red + 2 * BN);  // [16] {dy, dx, gather offset, weight slab offset}
    float* pscale = reinterpret_cast<float*>(ttab + SGAN_MAX_TAPS);  // [Ck]
    float* pshift = pscale + G.Ck;

    const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    // grid.x enumerates (M tile, N tile) pairs, N fastest, in XCD-contiguous order; grid.z = K split
    const int ntn = (G.N + BN - 1) / BN;
    const int item = sg_xcd_remap(bid, nblocks);
    int g, phz, mtile;
    sg_decode_tile(G, item / ntn, g, phz, mtile);
    const SgLocal P = sg_local(G, g);
    const int Hp = G.q[g].Hp[phz], Wp = G.q[g].Wp[phz];
    const int M = Hp * Wp;
    const int m0 = mtile * BM, n0 = (item % ntn) * BN;
    // Buffer loads: 32-bit byte offsets against a wave-uniform descriptor, and an out-of-range offset returns
    // zeros in hardware -- padding taps, rows past M, channels past N and k past the end need no select, no
    // validity flag and no 64-bit address arithmetic (the loop is VALU-issue bound, not latency bound).
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.w), 0, 0x7FFFFFFF, 0x00020000);
    constexpr int OOB = (int)0x80000000u;
    const int oa = G.oa[phz], ob = G.ob[phz];
    const int ktot = G.ktot[phz];
    const int Ck = P.Ck, N = P.N;
    // this workgroup's k-tile range (split-K)
    const int nkt_total = (ktot + 31) >> 5;
    const int kt_per = (nkt_total + P.ksplit - 1) / P.ksplit;
    const int wg_kt0 = split * kt_per;
    const int wg_nkt = max(min(nkt_total, wg_kt0 + kt_per) - wg_kt0, 0);   // may be 0 for a trailing split: writes zeros
    // this wave group's share; every group runs `nkt` iterations (shared barriers), tiles past `my_nkt` load as zeros
    const int nkt = (wg_nkt + KW - 1) / KW;
    const int kt0 = wg_kt0 + kg * nkt;
    const int my_nkt = min(max(wg_nkt - kg * nkt, 0), nkt);

    // ---- one-time setup: tap table, prologue scale/shift, reduction scratch ----
    if (threadIdx.x < SGAN_MAX_TAPS) {
        const bool v = tid < G.ntaps[phz];
        const SgTap tp = G.taps[v ? G.tap0[phz] + tid : 0];
        const int dy = v ? (int)tp.dy : 0, dx = v ? (int)tp.dx : 0;
        ttab[tid] = make_int4(dy, dx, (dy * P.Win + dx) * P.in_ld, v ? tp.w_off : 0);
    }
    for (int i = threadIdx.x; i < 2 * BN; i += 256 * KW) red[i] = 0.0;
    {   // always present (identity when there is no prologue) so the main loop is branch-free
        for (int c = threadIdx.x; c < Ck; c += 256 * KW) {
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
    // Element offset of a gathered chunk = a_base[row] + ttab[tap].z + channel: two adds per tile.
    const int adv_tap = 32 / Ck, adv_c = 32 - adv_tap * Ck;
    const int ntaps = G.ntaps[phz];
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    int a_iy[A_IT], a_ix[A_IT], a_base[A_IT], a_dst[A_IT];
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
        a_base[it] = (a_iy[it] * P.Win + a_ix[it]) * P.in_ld;
        a_dst[it] = row * 32 + sg_swz(row, ks);
    }
    const int a_ks = tid & 7;  // same for every it (256 % 8 == 0)
    int a_tap = (kt0 * 32 + a_ks * 4) / Ck;
    int a_c = kt0 * 32 + a_ks * 4 - a_tap * Ck;

    // ring of NSET register sets: tile kt+1 (landed, being written to LDS) and tiles kt+2 .. kt+NSET in flight.
    // Measured load-to-use latency under load is ~1 us (~2.4 k-tile iterations): three tiles must be in flight.
    constexpr int NSET = 4;
    f32x4 a_reg[NSET][A_IT];
    bool a_ok[NSET][A_IT];
    int a_cs[NSET] = {0, 0, 0, 0};  // channel of the staged A registers (for the prologue transform)
    f32x4 b_reg[NSET][B_IT];
    constexpr bool b_kcontig = BKC;
    // B, n-contiguous form (backward-data): element (k = e / NQ, n4 = e % NQ); its own (tap, channel) walk
    constexpr int NQ = BN / 4;
    int b_tap[B_IT], b_c[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int k = kt0 * 32 + (tid + it * 256) / NQ;
        b_tap[it] = k / Ck;
        b_c[it] = k - b_tap[it] * Ck;
    }
    int b_base[B_IT];     // k-contiguous form: row n of the weight slab
    bool b_rowok[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int e = tid + it * 256;
        const int n = e >> 3;
        b_rowok[it] = (B_IT * 256 == BN * 8 || e < BN * 8) && n0 + n < N;
        b_base[it] = (n0 + n) * P.w_ns;
    }
    // element offsets / validity of the NEXT tile to load (filled by next_addrs one iteration ahead)
    int a_off_n[A_IT], b_off_n[B_IT], a_cs_n = 0;
    bool a_ok_n[A_IT];

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    SG_SYNC();  // tap table visible

    // Pure address arithmetic for the next tile (no memory traffic except the LDS tap table): scheduled
    // inside the MFMA block of the previous tile.  Invalid elements get offset 0 and are zeroed when the
    // registers are written to LDS, so the loads themselves are unconditional.
    int ld_left = my_nkt;   // tiles of this group's range not yet addressed
    auto next_addrs = [&]() {
        const bool in_range = ld_left > 0;
        --ld_left;
        {
            const bool kok = (a_tap < ntaps) & in_range;
            const int4 t = ttab[kok ? a_tap : 0];
            a_cs_n = kok ? a_c : 0;
            const int toff = t.z + a_c;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int iy = a_iy[it] + t.x, ix = a_ix[it] + t.y;
                // bitwise on purpose: short-circuit evaluation turns into branches, and a branch ends the scheduling region
                // this arithmetic is meant to share with the MFMA block
                const bool ok = a_rowok[it] & kok & ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
                a_off_n[it] = ok ? (a_base[it] + toff) << 2 : OOB;
                if constexpr (PRO) a_ok_n[it] = ok;
            }
            if constexpr (b_kcontig) {
                const int woff = t.w + a_c;
#pragma unroll
                for (int it = 0; it < B_IT; ++it) b_off_n[it] = (b_rowok[it] & kok) ? (b_base[it] + woff) << 2 : OOB;
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
                const bool tok = (b_tap[it] < ntaps) & in_range;
                const int wtap = ttab[tok ? b_tap[it] : 0].w;
                const bool ok = (B_IT * 256 <= 32 * NQ || k < 32) & tok & (n0 + n4 * 4 < N);
                b_off_n[it] = ok ? (wtap + b_c[it] * P.w_ks + n0 + n4 * 4) << 2 : OOB;
                b_tap[it] += adv_tap;
                b_c[it] += adv_c;
                if (b_c[it] >= Ck) { b_c[it] -= Ck; ++b_tap[it]; }
            }
        }
    };

    // Issue the global loads of the tile whose offsets next_addrs prepared, into register set S.
    auto issue_loads = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        if constexpr (SG_ABLATE & 2) return;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            a_reg[S][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_off_n[it], 0, 0));
            if constexpr (PRO) a_ok[S][it] = a_ok_n[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            b_reg[S][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off_n[it], 0, 0));
        if constexpr (PRO) a_cs[S] = a_cs_n;
    };

    // Transform (norm + activation of the producer layer), mask and write register set S to LDS buffer S & 1.
    auto store_tile = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        float* Ab = As + (S & 1) * BM * 32;
        float* Bb = Bs + (S & 1) * BN * 32;
        if constexpr (PRO) {
#if SG_ABLATE & 8
            const f32x4 sc = (f32x4){1.f, 1.f, 1.f, 1.f} * pro_neg, sh = (f32x4){0.f, 0.f, 0.f, 0.f} * pro_neg;
#else
            const f32x4 sc = *reinterpret_cast<const f32x4*>(pscale + a_cs[S]);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(pshift + a_cs[S]);
#endif
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                // act(y) = max(y, neg * y) for neg <= 1 (checked on the host); the conv zero padding applies AFTER norm +
                // activation, so the validity flag rides along as a factor: okf * max(y, neg * y) = max(okf * y, okf * neg * y).
                // Written on whole vectors so that it maps to packed fp32 math (2 pk_fma + 4 pk_mul + 4 max per 16 bytes).
                const float okf = a_ok[S][it] ? 1.f : 0.f;
                const float okn = okf * pro_neg;
                const f32x4 y = a_reg[S][it] * sc + sh;
                const f32x4 yp = y * okf, yn = y * okn;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(yp[j], yn[j]);
                *reinterpret_cast<f32x4*>(Ab + a_dst[it]) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) *reinterpret_cast<f32x4*>(Ab + a_dst[it]) = a_reg[S][it];
        }
        if constexpr (b_kcontig) {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int e = tid + it * 256;
                const int n = e >> 3, ks = e & 7;
                if (B_IT * 256 == BN * 8 || e < BN * 8) *reinterpret_cast<f32x4*>(Bb + n * 32 + sg_swz(n, ks)) = b_reg[S][it];
            }
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int e = tid + it * 256;
                const int k = e / NQ, n4 = e % NQ;
                if (B_IT * 256 <= 32 * NQ || k < 32) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = n4 * 4 + j;
                        Bb[row * 32 + sg_swz(row, k >> 2) + (k & 3)] = b_reg[S][it][j];
                    }
                }
            }
        }
    };

    const int fr = lane & 15, fq = lane >> 4;

    // Software pipeline, one barrier per k-tile, global loads NSET tiles ahead.  Iteration kt (S = kt % NSET):
    //   (1) issue the loads of tile kt+NSET into register set S (tile kt left it for LDS one iteration ago);
    //   (2) one scheduling region: MFMA block on tile kt (LDS buffer kt & 1)  ||  transform + LDS store of
    //       tile kt+1 (register set S+1, issued NSET-1 iterations ago: its counted vmcnt wait is free) into the
    //       other LDS buffer  ||  address arithmetic of tile kt+NSET+1;
    //   (3) barrier.
    // Tiles past the end of K read offset 0 and land as zeros in a buffer that is never consumed.
    auto mfma_tile = [&](int buf) {
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
            if constexpr (!(SG_ABLATE & 1)) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < MB; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
                for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(bf[j]));
            }
        }
    };
    auto iteration = [&](auto S_) {
        constexpr int S = decltype(S_)::value;
        issue_loads(std::integral_constant<int, S>{});           // tile kt+NSET
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile(S & 1);                                         // tile kt
        if constexpr (!(SG_ABLATE & 4)) store_tile(std::integral_constant<int, (S + 1) % NSET>{});   // tile kt+1
        next_addrs();                                             // tile kt+NSET+1
        SG_SYNC();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    next_addrs();
    issue_loads(I0{});   // tile 0
    next_addrs();
    issue_loads(I1{});   // tile 1
    next_addrs();
    issue_loads(I2{});   // tile 2
    next_addrs();
    issue_loads(I3{});   // tile 3
    next_addrs();        // offsets of tile 4
    store_tile(I0{});
    SG_SYNC();
    {
        int kt = 0;
        for (; kt + 3 < nkt; kt += 4) {
            iteration(I0{});
            iteration(I1{});
            iteration(I2{});
            iteration(I3{});
        }
        if (kt < nkt) iteration(I0{});
        if (kt + 1 < nkt) iteration(I1{});
        if (kt + 2 < nkt) iteration(I2{});
    }

    const bool want_stats = P.stats != nullptr;
    if constexpr (KW == 2) {   // group 1 -> group 0 (the tile buffers are dead: every wave is past the loop's last barrier)
        f32x4* xch = reinterpret_cast<f32x4*>(smem);
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) xch[(i * NB + j) * 256 + tid] = acc[i][j];
        }
        SG_SYNC();
        if (kg == 1) {
            if (P.ksplit <= 1 && want_stats) SG_SYNC();   // keep the barrier count of group 0's epilogue
            return;
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] += xch[(i * NB + j) * 256 + tid];
    }

    // ---- epilogue ----
    if (P.ksplit > 1) {   // split-K: raw partial tile to this split's slab; sg_splitk_epilogue_kernel finishes
        float* sl = P.slab + (int64_t)split * P.slab_stride;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wn * WTN + j * 16 + fr;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + fq * 4 + r;
                    if (m < M && n < N) {
                        const int py = m / Wp, px = m - py * Wp;
                        const int64_t pix = (int64_t)(py * P.os + oa) * P.Wout + (px * P.os + ob);
                        sl[pix * N + n] = acc[i][j][r];
                    }
                }
            }
        }
        return;
    }
    const bool dact = P.xref != nullptr;
    // the forward tensor at the result positions, every load in flight before the first store (which may alias it as far as the
    // compiler knows: element by element each load would wait out a memory round trip behind the previous store)
    float xv[NB][MB][4];
    if (dact) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + fq * 4 + r;
                    const int n = n0 + wn * WTN + j * 16 + fr;
                    float x = 0.f;
                    if (m < M && n < N) {
                        const int py = m / Wp, px = m - py * Wp;
                        x = P.xref[((int64_t)(py * P.os + oa) * P.Wout + (px * P.os + ob)) * P.xref_ld + n];
                    }
                    xv[j][i][r] = x;
                }
    }
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
        double s1 = 0.0, s2 = 0.0;
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
                        const float x = xv[j][i][r];
                        const float xhat = (x - x_mean) * x_rstd;
                        const float y = xnorm ? (x_g * xhat + x_b) : x;
                        v *= (y > 0.f ? 1.f : xn_neg);
                        s1 += (double)v;
                        s2 += (double)(v * xhat);
                    } else {
                        s1 += (double)v;
                        s2 += (double)v * (double)v;
                        if (P.out_act == SGAN_ACT_TANH) v = tanhf(v);
                    }
                    if (P.accum) v += P.out[pix * P.out_ld + n];
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

template <int BM, int BN, int WGM, int WGN, bool BKC, bool PRO, int KW = 1>
__global__ __launch_bounds__(256 * KW) void sg_igemm_kernel(const SgIgemmParams G) {
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sg_igemm_body<BM, BN, WGM, WGN, BKC, PRO, KW>(G, smem, blockIdx.x, gridDim.x, blockIdx.z);
}

#ifndef SG_KERNELS_ONLY      // sgan_fused.hip includes this file for sg_igemm_body only
// ------------------------------------------------------------------------------------------
// Direct kernel for results with <= 4 stored channels (PatchGAN logits head Cout = 1, generator output
// Cout = 2, image gradient of the first discriminator conv Cin = 2).  Such a GEMM has N = 4: an MFMA
// tile would be >= 75 % padding and, worse, one workgroup would walk the whole K = 4096 reduction
// serially.  Here LPP lanes share one output pixel: each lane takes every LPP-th 16-byte chunk of the
// (tap, channel) reduction (coalesced 16 B/lane loads of the gathered NHWC tensor and of the weights,
// which stay L1/L2 resident), keeps 4 accumulators, and a shuffle tree combines the lanes.
// Same prologue (norm + activation on load) and bias / tanh epilogue as the MFMA kernel.
// ------------------------------------------------------------------------------------------
template <int LPP, int R, int U, int NR, bool BKC>
__global__ __launch_bounds__(256) void sg_conv_small_n_kernel(const SgIgemmParams G) {
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* tdy = reinterpret_cast<int*>(smem);
    int* tdx = tdy + SGAN_MAX_TAPS;
    int* two = tdx + SGAN_MAX_TAPS;
    float* pscale = reinterpret_cast<float*>(two + SGAN_MAX_TAPS);
    float* pshift = pscale + G.Ck;
    const int tid = threadIdx.x;
    int g, phz, mtile;
    sg_decode_tile(G, blockIdx.x, g, phz, mtile);
    const SgLocal P = sg_local(G, g);
    const int Wp = G.q[g].Wp[phz], M = G.q[g].Hp[phz] * Wp;
    const int ntaps = G.ntaps[phz], Ck = P.Ck;
    const bool has_pro = (P.pro.stats != nullptr) || (P.pro.act != SGAN_ACT_NONE);
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    if (tid < SGAN_MAX_TAPS) {
        const bool v = tid < ntaps;
        const SgTap tp = G.taps[v ? G.tap0[phz] + tid : 0];
        tdy[tid] = v ? (int)tp.dy : 0;
        tdx[tid] = v ? (int)tp.dx : 0;
        two[tid] = v ? tp.w_off : 0;
    }
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
    SG_SYNC();
    // LPP lanes share a pixel; every lane works on R pixels at once (PPB apart), so one set of weight chunks and one
    // scale/shift chunk serve R gathered chunks, and a workgroup's setup is spread over PPB * R pixels.  The loop is
    // load-latency bound: the loads of U chunks (U * (R + NW) 16-byte loads per lane) are issued before any of them is
    // consumed.
    constexpr int PPB = 256 / LPP;
    constexpr int NW = BKC ? NR : 4;   // weight chunks per k chunk
    const int sub = tid % LPP;
    const int mbase = mtile * (PPB * R) + tid / LPP;
    int iy0[R], ix0[R], pyv[R], pxv[R];
    bool mok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = mbase + r * PPB;
        mok[r] = m < M;
        pyv[r] = mok[r] ? m / Wp : 0;
        pxv[r] = mok[r] ? m - pyv[r] * Wp : 0;
        iy0[r] = pyv[r] * P.is;
        ix0[r] = pxv[r] * P.is;
    }
    // (tap, channel) walk of this lane: chunk q = sub, sub + LPP, ...  (4 k per chunk)
    const int adv_tap = (4 * LPP) / Ck, adv_c = 4 * LPP - adv_tap * Ck;
    int tap = (4 * sub) / Ck;
    int c = 4 * sub - tap * Ck;
    float acc[R][NR];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < NR; ++n) acc[r][n] = 0.f;
    const int nq = (G.ktot[phz] + 4 * LPP - 1) / (4 * LPP);
    for (int q0 = 0; q0 < nq; q0 += U) {
        f32x4 a[U][R], w[U][NW];
        bool ok[U][R];
        int ccs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool kok = tap < ntaps;     // chunks past the end of K (q0 + u >= nq included) load offset 0 and are masked
            const int t = kok ? tap : 0;
            const int dy = tdy[t], dx = tdx[t];
            const int cc = kok ? c : 0;
            ccs[u] = cc;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int iy = iy0[r] + dy, ix = ix0[r] + dx;
                ok[u][r] = mok[r] && kok && (unsigned)iy < (unsigned)P.Hin && (unsigned)ix < (unsigned)P.Win;
                const int off = ok[u][r] ? (iy * P.Win + ix) * P.in_ld + cc : 0;
                a[u][r] = *reinterpret_cast<const f32x4*>(P.in + off);
            }
            const float* wb = P.w + two[t];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if constexpr (BKC) w[u][j] = *reinterpret_cast<const f32x4*>(wb + j * P.w_ns + cc);   // W[tap][n = j][k chunk]
                else w[u][j] = *reinterpret_cast<const f32x4*>(wb + (int64_t)(cc + j) * P.w_ks);      // W[tap][k = cc + j][n 0..3]
            }
            tap += adv_tap;
            c += adv_c;
            if (c >= Ck) { c -= Ck; ++tap; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 sc = (f32x4){1.f, 1.f, 1.f, 1.f}, sh = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (has_pro) {
                sc = *reinterpret_cast<const f32x4*>(pscale + ccs[u]);
                sh = *reinterpret_cast<const f32x4*>(pshift + ccs[u]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f32x4 v = a[u][r];
                if (has_pro) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float y = v[j] * sc[j] + sh[j];
                        v[j] = y > 0.f ? y : y * pro_neg;
                    }
                }
                if (!ok[u][r]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    if constexpr (BKC) acc[r][n] += v[0] * w[u][n][0] + v[1] * w[u][n][1] + v[2] * w[u][n][2] + v[3] * w[u][n][3];
                    else acc[r][n] += v[0] * w[u][0][n] + v[1] * w[u][1][n] + v[2] * w[u][2][n] + v[3] * w[u][3][n];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) {
#pragma unroll
            for (int n = 0; n < NR; ++n) acc[r][n] += __shfl_xor(acc[r][n], o);
        }
        if (sub == 0 && mok[r]) {
            const int64_t pix = (int64_t)(pyv[r] * P.os + G.oa[phz]) * P.Wout + (pxv[r] * P.os + G.ob[phz]);
            f32x4 o4;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float v = (n < NR ? acc[r][n < NR ? n : 0] : 0.f) + (P.bias ? P.bias[n] : 0.f);
                if (P.out_act == SGAN_ACT_TANH) v = tanhf(v);
                o4[n] = v;
            }
            *reinterpret_cast<f32x4*>(P.out + pix * P.out_ld) = o4;
        }
    }
}


// ------------------------------------------------------------------------------------------
// One-channel head of a stride-1 Conv2d (PatchGAN logits: 256 -> 1, k4 s1 p2 on 66 x 66): every output pixel is a 4096-term dot
// product, and the gather kernel above re-fetches each input element once per tap (16x) from L1/L2.  Here a workgroup owns an
// 8 x 8 block of output pixels: per 32-channel block the (8 + span - 1)^2 input patch is loaded once, normalised / activated and
// kept in LDS; lane (pixel, channel octet) walks the taps reading its 8 channels of the shifted pixel and the matching 8 weights
// (all weights of the layer sit in LDS, read as broadcasts), and a 4-lane shuffle adds the octets.  Exact fp32 (VALU FMA), the
// same norm-on-load prologue and bias / tanh epilogue as the other forward kernels; no output statistics (heads have no norm).
// ------------------------------------------------------------------------------------------
#define SGH_PS 144     // bytes per patch pixel (32 fp32 channels + 16 pad: the four octets of neighbouring pixels spread over the banks)
// TH x 8 result pixels per workgroup; the 256 threads are (pixel, channel group): 256 / (8 TH) groups of 32 / groups channels each.
// Short tiles (TH 4, 2) give a small launch enough workgroups to overlap their load latencies (the kernel is latency-, not
// bandwidth-bound: one 8 x 8 tile per CU leaves one wave per SIMD).
template <int TH>
__global__ __launch_bounds__(256) void sg_conv_head_kernel(const SgIgemmParams G) {
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    constexpr int NG = 256 / (8 * TH), CPG = 32 / NG;       // channel groups per pixel, channels per group: (4, 8) (8, 4) (16, 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int g, phz, mtile;
    sg_decode_tile(G, blockIdx.x, g, phz, mtile);
    const SgLocal P = sg_local(G, g);
    const int Hp = G.q[g].Hp[phz], Wp = G.q[g].Wp[phz];
    const int tiles_x = (Wp + 7) >> 3;
    const int ty0 = (mtile / tiles_x) * TH, tx0 = (mtile % tiles_x) * 8;
    const int ntaps = G.ntaps[phz], Ck = P.Ck, ncb = Ck >> 5;
    const int PH = G.pph[phz], PW = G.ppw[phz], dy0 = G.pdy0[phz], dx0 = G.pdx0[phz];
    const int RS = PW * SGH_PS;
    const int npix = PH * PW;
    char* Ap = smem;                                                                  // [PH][PW] pixels x 32 channels
    float* Ws = reinterpret_cast<float*>(smem + ((PH * RS + 255) & ~255));            // [ntaps][Ck]
    int* toff = reinterpret_cast<int*>(Ws + ntaps * Ck);                              // [ntaps] patch byte offset of the tap
    float* pscale = reinterpret_cast<float*>(toff + SGAN_MAX_TAPS);                   // [Ck]
    float* pshift = pscale + Ck;
    const bool has_pro = (P.pro.stats != nullptr) || (P.pro.act != SGAN_ACT_NONE);
    const float pro_neg = P.pro.act == SGAN_ACT_NONE ? 1.f : (P.pro.act == SGAN_ACT_RELU ? 0.f : P.pro.slope);
    if (tid < ntaps) {
        const SgTap tp = G.taps[G.tap0[phz] + tid];
        toff[tid] = ((int)tp.dy - dy0) * RS + ((int)tp.dx - dx0) * SGH_PS;
    }
    for (int i = tid * 4; i < ntaps * Ck; i += 1024) {     // W[tap][n = 0][k]: one contiguous Ck row per tap
        const int t = i / Ck, k = i - t * Ck;
        *reinterpret_cast<f32x4*>(Ws + i) = *reinterpret_cast<const f32x4*>(P.w + G.taps[G.tap0[phz] + t].w_off + k);
    }
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
    // staging: item e = 8 * patch pixel + channel quad; S_IT items per thread cover 32 S_IT patch pixels.  The loads of D channel
    // blocks are in flight at once (register ring with static slots: the block loop is unrolled by D).  Measured on the fcgan
    // heads (256 channels, 230 tall tiles): D = 6..8 is no faster than D = 2 -- the launch is not bound by this latency
    constexpr int S_IT = TH, D = 2;
    const int cq = tid & 7;
    int s_goff[S_IT], s_dst[S_IT];
#pragma unroll
    for (int it = 0; it < S_IT; ++it) {
        const int p = (tid + it * 256) >> 3;
        const int pr = p / PW, pc = p - pr * PW;
        const int iy = ty0 + pr + dy0, ix = tx0 + pc + dx0;
        const bool ok = (p < npix) & ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
        s_goff[it] = ok ? (iy * P.Win + ix) * P.in_ld + cq * 4 : -1;
        s_dst[it] = p < npix ? pr * RS + pc * SGH_PS + cq * 16 : -1;
    }
    f32x4 s_regs[D][S_IT];
    auto issue = [&](int cb, f32x4* s_reg) {
#pragma unroll
        for (int it = 0; it < S_IT; ++it)
            if (s_goff[it] >= 0 && cb < ncb) s_reg[it] = *reinterpret_cast<const f32x4*>(P.in + s_goff[it] + cb * 32);
    };
    auto store = [&](int cb, const f32x4* s_reg) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(pscale + cb * 32 + cq * 4), sh = *reinterpret_cast<const f32x4*>(pshift + cb * 32 + cq * 4);
#pragma unroll
        for (int it = 0; it < S_IT; ++it) {
            if (s_dst[it] < 0) continue;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};      // zero padding applies AFTER norm + activation
            if (s_goff[it] >= 0) {
                v = s_reg[it];
                if (has_pro) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float y = v[j] * sc[j] + sh[j];
                        v[j] = y > 0.f ? y : y * pro_neg;
                    }
                }
            }
            *reinterpret_cast<f32x4*>(Ap + s_dst[it]) = v;
        }
    };
    const int px = tid / NG, co = tid % NG;      // output pixel of the tile, channel group of the block
    const int f_base = ((px >> 3) * RS + (px & 7) * SGH_PS) + co * (CPG * 4);
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, s_regs[d]);
    SG_SYNC();      // tap offsets, weights, scale / shift visible
    for (int cb0 = 0; cb0 < ncb; cb0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int cb = cb0 + d;
        if (cb >= ncb) break;
        store(cb, s_regs[d]);
        issue(cb + D, s_regs[d]);
        SG_SYNC();
        const float* wrow = Ws + cb * 32 + co * CPG;
#pragma unroll 4
        for (int t = 0; t < ntaps; ++t) {
            const char* a = Ap + f_base + toff[t];
            if constexpr (CPG == 8) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 16);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + t * Ck), w1 = *reinterpret_cast<const f32x4*>(wrow + t * Ck + 4);
                acc += (a0[0] * w0[0] + a0[1] * w0[1]) + (a0[2] * w0[2] + a0[3] * w0[3]) + (a1[0] * w1[0] + a1[1] * w1[1]) + (a1[2] * w1[2] + a1[3] * w1[3]);
            } else if constexpr (CPG == 4) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(a);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + t * Ck);
                acc += (a0[0] * w0[0] + a0[1] * w0[1]) + (a0[2] * w0[2] + a0[3] * w0[3]);
            } else {
                const f32x2 a0 = *reinterpret_cast<const f32x2*>(a);
                const f32x2 w0 = *reinterpret_cast<const f32x2*>(wrow + t * Ck);
                acc += a0[0] * w0[0] + a0[1] * w0[1];
            }
        }
        SG_SYNC();   // everyone is done with the patch before the next block overwrites it
      }
    }
#pragma unroll
    for (int o = 1; o < NG; o <<= 1) acc += __shfl_xor(acc, o);
    const int py = ty0 + (px >> 3), pxx = tx0 + (px & 7);
    if (co == 0 && py < Hp && pxx < Wp) {
        const int64_t pix = (int64_t)(py * P.os + G.oa[phz]) * P.Wout + (pxx * P.os + G.ob[phz]);
        float v = acc + (P.bias ? P.bias[0] : 0.f);
        if (P.out_act == SGAN_ACT_TANH) v = tanhf(v);
        f32x4 o4 = (f32x4){v, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 1; n < 4; ++n) {       // padding channels carry the bias like the generic kernel (zero in practice)
            float b = P.bias ? P.bias[n] : 0.f;
            if (P.out_act == SGAN_ACT_TANH) b = tanhf(b);
            o4[n] = b;
        }
        *reinterpret_cast<f32x4*>(P.out + pix * P.out_ld) = o4;
    }
}

// eligibility + patch extents (shared fields with the split patch kernel); 8 x 8 tiles, one phase, unit stride, one real channel
static bool sg_head_plan(SgIgemmParams& P) {
    if (P.is != 1 || P.os != 1 || P.nphase != 1 || P.n_real > 1 || P.w_ks != 1 || (P.Ck & 31) || P.Ck > 1024) return false;
    if (getenv("SGAN_NO_HEAD_KERNEL")) return false;
    for (int g = 0; g < P.nprob; ++g)
        if (P.q[g].stats || P.q[g].xref || P.q[g].accum || (P.q[g].out_ld & 3) || P.q[g].out_ld < 4) return false;
    int dy0 = 1 << 30, dy1 = -(1 << 30), dx0 = 1 << 30, dx1 = -(1 << 30);
    if (P.ntaps[0] < 4 || P.ntaps[0] > SGAN_MAX_TAPS) return false;
    for (int t = 0; t < P.ntaps[0]; ++t) {
        const SgTap& tp = P.taps[P.tap0[0] + t];
        dy0 = min(dy0, (int)tp.dy); dy1 = max(dy1, (int)tp.dy);
        dx0 = min(dx0, (int)tp.dx); dx1 = max(dx1, (int)tp.dx);
    }
    P.pdy0[0] = dy0; P.pdx0[0] = dx0;
    // tile height: 8 rows when that alone gives every CU a few workgroups, else 4 (SGAN_HEAD_TH = 8 / 4 / 2 overrides: tuning knob)
    long t8 = 0;
    for (int g = 0; g < P.nprob; ++g) t8 += (long)((P.q[g].Hp[0] + 7) / 8) * ((P.q[g].Wp[0] + 7) / 8);
    static const int th_env = getenv("SGAN_HEAD_TH") ? atoi(getenv("SGAN_HEAD_TH")) : 0;
    int th = t8 >= 1024 ? 8 : 4;      // fcgan heads (230 tall tiles): 27.6 us with 8 rows, 21.6 with 4, 22.5 with 2
    if (th_env == 8 || th_env == 4 || th_env == 2) th = th_env;
    P.pph[0] = th + dy1 - dy0; P.ppw[0] = 8 + dx1 - dx0;
    P.pph[1] = th;      // (phase 1 is unused by this kernel: carries the tile height to the launcher)
    if (P.pph[0] * P.ppw[0] > 32 * th) {      // the staging of a TH tile covers 32 TH patch pixels: big kernels keep the tall tile
        th = 8;
        P.pph[0] = th + dy1 - dy0;
        P.pph[1] = th;
    }
    if (P.pph[0] * P.ppw[0] > 256) return false;
    const size_t lds = (size_t)((P.pph[0] * P.ppw[0] * SGH_PS + 255) & ~255) + (size_t)P.ntaps[0] * P.Ck * 4 + SGAN_MAX_TAPS * 4 + (size_t)2 * P.Ck * 4;
    return lds <= 64 * 1024;
}

static int sg_launch_head(SgIgemmParams& P, hipStream_t st) {
    int t = 0;
    const int th = P.pph[1];
    for (int g = 0; g < P.nprob; ++g) {
        P.q[g].tile0[0] = t;
        t += ((P.q[g].Hp[0] + th - 1) / th) * ((P.q[g].Wp[0] + 7) / 8);
        for (int ph = 1; ph < SGAN_MAX_PHASES; ++ph) P.q[g].tile0[ph] = 1 << 30;
    }
    if (t == 0) return SGAN_OK;
    const size_t lds = (size_t)((P.pph[0] * P.ppw[0] * SGH_PS + 255) & ~255) + (size_t)P.ntaps[0] * P.Ck * 4 + SGAN_MAX_TAPS * 4 + (size_t)2 * P.Ck * 4;
    sg_prof_begin(st);
    if (th == 8) hipLaunchKernelGGL(sg_conv_head_kernel<8>, dim3(t), dim3(256), lds, st, P);
    else if (th == 4) hipLaunchKernelGGL(sg_conv_head_kernel<4>, dim3(t), dim3(256), lds, st, P);
    else hipLaunchKernelGGL(sg_conv_head_kernel<2>, dim3(t), dim3(256), lds, st, P);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = "sg_conv_head_kernel";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

template <int LPP, int R, int U, int NR>
static int sg_launch_small_n_nr(SgIgemmParams& P, hipStream_t st) {
    constexpr int PPB = 256 / LPP;
    const int tiles = sg_fill_tiles(P, PPB * R);
    if (tiles == 0) return SGAN_OK;
    dim3 grid(tiles, 1, 1);
    const size_t lds = 3 * SGAN_MAX_TAPS * 4 + (size_t)2 * P.Ck * 4;
    sg_prof_begin(st);
    if (P.w_ks == 1) hipLaunchKernelGGL((sg_conv_small_n_kernel<LPP, R, U, NR, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((sg_conv_small_n_kernel<LPP, R, U, NR, false>), grid, dim3(256), lds, st, P);
    SGAN_LAUNCH_CHECK();
    g_sgan_last_kernel = LPP == 64 ? "sg_conv_small_n_kernel<64>" : LPP == 16 ? "sg_conv_small_n_kernel<16>" : "sg_conv_small_n_kernel<8>";
    sg_prof_end(st, g_sgan_last_kernel);
    return SGAN_OK;
}

template <int LPP, int R, int U>
static int sg_launch_small_n(SgIgemmParams& P, hipStream_t st) {
    if (P.n_real <= 1) return sg_launch_small_n_nr<LPP, R, U, 1>(P, st);
    if (P.n_real == 2) return sg_launch_small_n_nr<LPP, R, U, 2>(P, st);
    return sg_launch_small_n_nr<LPP, R, U, 4>(P, st);
}

#include "sgan_c4.h"      // sg_conv_scatter4 / sg_conv_c4 kernels and their launchers (shared with the thin-pair launch of sgan_wgrad.hip)

// ------------------------------------------------------------------------------------------
// Split-K finish: out = epilogue( sum_s slab[s] ) -- the same epilogue as the unsplit kernel (bias,
// per-channel statistics, tanh; or act'(norm(x)) and the two norm-backward sums), as one 16-byte-
// per-lane streaming pass.  Thread t always works on channel group t % (N/4), so the statistics are
// accumulated in registers and reach memory as one LDS atomic + one fp64 atomic per channel per block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_splitk_epilogue_kernel(const SgIgemmParams G) {
    sg_warm_kernargs<(int)sizeof(SgIgemmParams)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SgLocal P = sg_local(G, 0);   // split-K is single-problem
    const int N = P.N, NQ = N >> 2;
    double* red = reinterpret_cast<double*>(smem);               // [2N], fp64 (see sg_igemm_kernel)
    float* cMean = reinterpret_cast<float*>(red + 2 * N);        // [N] (dact with norm)
    float* cRstd = cMean + N;
    float* cG = cRstd + N;
    float* cB = cG + N;
    const bool dact = P.xref != nullptr;
    const bool xnorm = dact && P.xn.stats != nullptr;
    const bool want_stats = P.stats != nullptr;
    for (int i = threadIdx.x; i < 2 * N; i += 256) red[i] = 0.0;
    for (int c = threadIdx.x; c < N; c += 256) {
        float mean = 0.f, rstd = 1.f;
        if (xnorm) sg_mean_rstd(P.xn, N, c, mean, rstd);
        cMean[c] = mean;
        cRstd[c] = rstd;
        cG[c] = (xnorm && P.xn.gamma) ? P.xn.gamma[c] : 1.f;
        cB[c] = (xnorm && P.xn.beta) ? P.xn.beta[c] : 0.f;
    }
    SG_SYNC();
    const float xn_neg = P.xn.act == SGAN_ACT_NONE ? 1.f : (P.xn.act == SGAN_ACT_RELU ? 0.f : P.xn.slope);
    const int64_t total = (int64_t)P.Hout * P.Wout * NQ;
    const int64_t stride = (int64_t)gridDim.x * 256;   // host makes this a multiple of NQ
    const int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(e0 % NQ) * 4;
    f32x4 bias = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (P.bias) bias = *reinterpret_cast<const f32x4*>(P.bias + n);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t e = e0; e < total; e += stride) {
        const int64_t pix = e / NQ;
        f32x4 v = bias;
        const float* sp0 = P.slab + pix * N + n;
        int sp = 0;
        for (; sp + 4 <= P.ksplit; sp += 4) {   // four independent 16-byte loads in flight
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(sp0 + (sp + 0) * P.slab_stride);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(sp0 + (sp + 1) * P.slab_stride);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(sp0 + (sp + 2) * P.slab_stride);
            const f32x4 t3 = *reinterpret_cast<const f32x4*>(sp0 + (sp + 3) * P.slab_stride);
            v += (t0 + t1) + (t2 + t3);
        }
        for (; sp < P.ksplit; ++sp) v += *reinterpret_cast<const f32x4*>(sp0 + sp * P.slab_stride);
        if (dact) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(P.xref + pix * P.xref_ld + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xhat = (x[j] - cMean[n + j]) * cRstd[n + j];
                const float y = xnorm ? (cG[n + j] * xhat + cB[n + j]) : x[j];
                v[j] *= (y > 0.f ? 1.f : xn_neg);
                s1[j] += (double)v[j];
                s2[j] += (double)(v[j] * xhat);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s1[j] += (double)v[j];
                s2[j] += (double)v[j] * (double)v[j];
                if (P.out_act == SGAN_ACT_TANH) v[j] = tanhf(v[j]);
            }
        }
        if (P.accum) v += *reinterpret_cast<const f32x4*>(P.out + pix * P.out_ld + n);
        *reinterpret_cast<f32x4*>(P.out + pix * P.out_ld + n) = v;
    }
    if (want_stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&red[n + j], s1[j]);
            atomicAdd(&red[N + n + j], s2[j]);
        }
        SG_SYNC();
        for (int c = threadIdx.x; c < N; c += 256) {
#ifndef SG_NO_STAT_ATOMICS      // diagnostics build: what the same-address fp64 atomics cost
            double* st = sg_stat_replica(P.stats, P.stats_rep, blockIdx.x);
            atomicAdd(&st[c], red[c]);
            atomicAdd(&st[P.stats_sq + c], red[N + c]);
#endif
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
        // output_padding (nn.ConvTranspose2d(k3, s2, p1, output_padding=1) of the resnet generators, models/networks.py:250-252): up to
        // s - 1 extra rows / columns; they are ordinary outputs of the taps that still reach an input pixel
        const int h0 = (d->Hin - 1) * s - 2 * p + k, w0 = (d->Win - 1) * s - 2 * p + k;
        if (d->Hout < h0 || d->Hout >= h0 + s || d->Wout < w0 || d->Wout >= w0 + s)
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

int sg_fill_tiles(SgIgemmParams& P, int rows_per_tile) {
    int t = 0;
    for (int g = 0; g < P.nprob; ++g)
        for (int ph = 0; ph < P.nphase; ++ph) {
            P.q[g].tile0[ph] = t;
            t += sg_cdiv(P.q[g].Hp[ph] * P.q[g].Wp[ph], rows_per_tile);
        }
    for (int g = 0; g < P.nprob; ++g)
        for (int ph = P.nphase; ph < SGAN_MAX_PHASES; ++ph) P.q[g].tile0[ph] = 1 << 30;
    return t;
}

int sg_max_k(const SgIgemmParams& P) {
    int k = 0;
    for (int i = 0; i < P.nphase; ++i) k = max(k, P.ktot[i]);
    return k;
}

long sg_total_tiles(const SgIgemmParams& P, int BM) {
    long t = 0;
    for (int g = 0; g < P.nprob; ++g)
        for (int ph = 0; ph < P.nphase; ++ph) t += sg_cdiv(P.q[g].Hp[ph] * P.q[g].Wp[ph], BM);
    return t;
}

// Rough duration (us) of `wgs` 64x64 workgroups walking `nkt_wg` k-tiles each, fitted to the per-call timings of the
// cgan step: a workgroup alone on its CU needs ~1.1 us per k-tile (latency bound, ~40 % of the CU's MFMA rate); from two
// co-resident workgroups on the CU is saturated at ~0.69 us per k-tile per workgroup, and the chip waits for the fullest CU.
static double sg_time_model(long wgs, int nkt_wg) {
    const double per_cu = (double)((wgs + 255) / 256);
    return nkt_wg * fmax(1.1, per_cu * 0.69);
}

// Split-K plan (single problem only): deep reductions on small grids (a 17x17 layer is 24 workgroups walking
// 64-128 k-tiles one after the other) are cut so that ~2 workgroups land on every CU.  Needs N % 4 == 0 channels
// with 256 % (N/4) == 0 for the epilogue.
int sg_plan_ksplit(const SgIgemmParams& P, int BM, int BN) {
    if (P.nprob != 1) return 1;
    const SgProb& Q = P.q[0];
    const int NQ = P.N >> 2;
    if (NQ <= 0 || 256 % NQ != 0 || (Q.out_ld & 3) || (Q.xref && (Q.xref_ld & 3))) return 1;
    const long blocks = sg_total_tiles(P, BM) * sg_cdiv(P.N, BN);
    if (blocks == 0) return 1;
    const int nkt = sg_cdiv(sg_max_k(P), 32);
    // measured on MI355X (tools/bench_layers.py): the slab round trip only pays when the grid is well under
    // one workgroup per CU, and the deeper the reduction the larger the grid it still pays for
    const int min_nkt = blocks <= 32 ? 16 : blocks <= 96 ? 32 : blocks <= 192 ? 64 : 1 << 30;
    // Grids a little over one workgroup per CU with a deep reduction (discriminator 256 -> 512 @65x65: 268 tiles x 256
    // k-tiles): twelve CUs get two tiles and the chip waits for them while the rest run one workgroup each, well under a CU's
    // MFMA rate.  A split both fills the CUs and evens them out; pick it with sg_time_model().
    static const int mid = getenv("SGAN_NO_MID_SPLIT") ? 0 : 1;
    static const int mid_force = getenv("SGAN_MID_KS") ? atoi(getenv("SGAN_MID_KS")) : 0;    // tuning knob
    static const int mid_nkt = getenv("SGAN_MID_NKT") ? atoi(getenv("SGAN_MID_NKT")) : 128;          // tuning knob
    if (mid && blocks > 192 && blocks <= 768 && nkt >= mid_nkt) {
        const double slab_us = (double)Q.Hout * Q.Wout * P.N * 8.0 / 4e6;     // write + read of one slab at ~4 TB/s
        int best = 1;
        double best_cost = sg_time_model(blocks, nkt);
        for (int c : {2, 3, 4, 6, 8}) {
            if (nkt / c < 16) break;
            const double cost = sg_time_model(blocks * c, sg_cdiv(nkt, c)) + c * slab_us + 12.0;
            if (cost < 0.95 * best_cost) { best = c; best_cost = cost; }
        }
        if (mid_force) best = mid_force;
        if (best > 1) {
            const int per = sg_cdiv(nkt, best);
            return sg_cdiv(nkt, per);
        }
        return 1;
    }
    if (nkt < min_nkt) return 1;
    int ks = (int)sg_cdiv(512, (int)blocks);
    ks = min(ks, nkt / 4);
    static const int ks_max = getenv("SGAN_KS_MAX") ? atoi(getenv("SGAN_KS_MAX")) : 32;      // tuning / test knob
    ks = min(ks, ks_max);
    if (ks < 2) return 1;
    const int per = sg_cdiv(nkt, ks);
    return sg_cdiv(nkt, per);   // every split non-empty
}

int sg_launch_splitk_epilogue(const SgIgemmParams& P, hipStream_t st) {
    const int NQ = P.N >> 2;
    const int64_t total = (int64_t)P.q[0].Hout * P.q[0].Wout * NQ;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;   // 256 % NQ == 0, so blocks * 256 is a multiple of NQ
    const size_t elds = (size_t)8 * P.N * 4;
    hipLaunchKernelGGL(sg_splitk_epilogue_kernel, dim3(blocks), dim3(256), elds, st, P);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

template <int BM, int BN, int WGM, int WGN>
static int sg_launch_igemm(SgIgemmParams& P, hipStream_t st, float* ws, int64_t ws_bytes) {
    const bool bkc = P.w_ks == 1;
    const int tiles = sg_fill_tiles(P, BM);
    if (tiles == 0) return SGAN_OK;
    int ks = sg_plan_ksplit(P, BM, BN);
    const int64_t slab = (int64_t)P.q[0].Hout * P.q[0].Wout * P.N;
    if (ks > 1 && (!ws || ws_bytes < (int64_t)ks * slab * 4)) ks = 1;   // no workspace: unsplit (still correct)
    P.ksplit = ks;
    P.slab = ks > 1 ? ws : nullptr;
    P.slab_stride = slab;
    dim3 grid(tiles * sg_cdiv(P.N, BN), 1, ks);
    // two wave groups per workgroup when the grid leaves most SIMDs with a single wave (see the kernel header)
    static const long kw2_max = getenv("SGAN_KW2_MAX") ? atol(getenv("SGAN_KW2_MAX")) : 512;
    const long wgs = (long)grid.x * ks;
    const int nkt_wg = sg_cdiv(sg_cdiv(sg_max_k(P), 32), ks);
    const bool kw2 = BM == 64 && bkc && wgs <= kw2_max && nkt_wg >= 8;
    const int kw = kw2 ? 2 : 1;
    const size_t lds = (size_t)(kw * (2 * BM * 32 + 2 * BN * 32) + 4 * BN) * 4 + SGAN_MAX_TAPS * 16 + (size_t)2 * P.Ck * 4;
    if (lds > 160 * 1024) return sgan_fail(SGAN_ERR_UNSUPPORTED, "LDS %zu too large", lds);
    sg_prof_begin(st);
    bool pro = P.pro_act != SGAN_ACT_NONE;
    for (int g = 0; g < P.nprob; ++g) pro = pro || P.q[g].pro_stats != nullptr;
    if (!bkc && pro) return sgan_fail(SGAN_ERR_UNSUPPORTED, "backward-data has no prologue");
    if constexpr (BM == 64) {
        if (kw2 && pro) hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, true, true, 2>), grid, dim3(512), lds, st, P);
        else if (kw2) hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, true, false, 2>), grid, dim3(512), lds, st, P);
    }
    if (kw2) { /* launched above */ }
    else if (!bkc) hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, false, false>), grid, dim3(256), lds, st, P);
    else if (pro) hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, true, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((sg_igemm_kernel<BM, BN, WGM, WGN, true, false>), grid, dim3(256), lds, st, P);
    SGAN_LAUNCH_CHECK();
    if (bkc)
        g_sgan_last_kernel = (BM == 64 && BN == 32) ? "sg_igemm_kernel<64,32,2,2,true>" : BM == 64 ? "sg_igemm_kernel<64,64,2,2,true>" : BN == 64 ? "sg_igemm_kernel<128,64,2,2,true>"
                             : BN == 32 ? "sg_igemm_kernel<128,32,4,1,true>" : "sg_igemm_kernel<128,16,4,1,true>";
    else
        g_sgan_last_kernel = (BM == 64 && BN == 32) ? "sg_igemm_kernel<64,32,2,2,false>" : BM == 64 ? "sg_igemm_kernel<64,64,2,2,false>" : BN == 64 ? "sg_igemm_kernel<128,64,2,2,false>"
                             : BN == 32 ? "sg_igemm_kernel<128,32,4,1,false>" : "sg_igemm_kernel<128,16,4,1,false>";
    sg_prof_end(st, g_sgan_last_kernel);
    if (ks > 1) return sg_launch_splitk_epilogue(P, st);
    return SGAN_OK;
}

static void sg_pick_tile(const SgIgemmParams& P, int* BM, int* BN) {
    if (P.N <= 16) { *BM = 128; *BN = 16; return; }
    if (P.N <= 32) { *BM = 128; *BN = 32; return; }
    // 128x64 tiles lose to 64x64 at every grid size measured (cgan step 19.4 -> 18.8 ms without them: the smaller tile
    // keeps three workgroups per CU); kept selectable for tuning only
    const long blocks128 = sg_total_tiles(P, 128) * sg_cdiv(P.N, 64);
    static const long min128 = getenv("SGAN_MIN128") ? atol(getenv("SGAN_MIN128")) : (1L << 60);
    *BM = blocks128 >= min128 ? 128 : 64;
    *BN = 64;
}

static bool sg_use_small_n(const SgIgemmParams& P) {
    if (P.N != 4) return false;
    for (int g = 0; g < P.nprob; ++g)
        if (P.q[g].xref || P.q[g].stats || (P.q[g].out_ld & 3) || P.q[g].accum) return false;
    return true;
}

static int sg_dispatch_igemm(SgIgemmParams& P, hipStream_t st, float* ws, int64_t ws_bytes) {
    P.ksplit = 1;
    P.slab = nullptr;
    P.slab_stride = 0;
    if (sg_use_small_n(P)) {   // skinny result: direct kernels
        if (sg_use_scatter4(P)) return sg_launch_scatter4(P, st);
        if (sg_head_plan(P)) {
            const int r2 = sg_launch_head2(P, st);      // sgan_head.hip: channel-per-thread form (k 3 / 4, >= 64 channels)
            return r2 != 1 ? r2 : sg_launch_head(P, st);
        }
        const int ktot = sg_max_k(P);
        if (ktot >= 2048) return sg_launch_small_n<64, 1, 8>(P, st);
        if (ktot >= 256) return sg_launch_small_n<16, 2, 4>(P, st);
        return sg_launch_small_n<8, 4, 4>(P, st);
    }
    const int e3 = sg_igemm3_eligible(P);   // split-bf16 MFMA (sgan_igemm3.hip)
    if (e3 < 0) return e3;
    if (e3) return sg_launch_igemm3(P, st, ws, ws_bytes);
    if (sg_use_c4(P)) return sg_launch_c4(P, st);     // first conv on the image: one MFMA per tap, weights in registers
    int BM, BN;
    sg_pick_tile(P, &BM, &BN);
    if (BN == 16) return sg_launch_igemm<128, 16, 4, 1>(P, st, ws, ws_bytes);
    if (BN == 32) return sg_launch_igemm<128, 32, 4, 1>(P, st, ws, ws_bytes);
    if (BM == 128) return sg_launch_igemm<128, 64, 2, 2>(P, st, ws, ws_bytes);
    // short reductions on small grids: half-width tiles double the workgroups per CU, so one's pipeline fill / epilogue
    // overlaps another's MFMA block (measured on the fcgan step: 3.70 -> 3.51 ms; thresholds from a sweep, env-tunable)
    static const int half = getenv("SGAN_HALF_TILES") ? atoi(getenv("SGAN_HALF_TILES")) : 1024;
    if (half && BM == 64) {
        const long blocks = sg_total_tiles(P, 64) * sg_cdiv(P.N, 64);
        const int nkt = sg_cdiv(sg_max_k(P), 32);
        static const int half_nkt = getenv("SGAN_HALF_NKT") ? atoi(getenv("SGAN_HALF_NKT")) : 128;
        // ... unless the full-width tiles happen to fill the chip in one go: 3 workgroups of 64x64 fit a CU (768 slots); a
        // grid of 80..100 % of that runs as one balanced wave (measured on the grouped 128->256 discriminator layer:
        // 157 -> 137 us), where 2x as many half tiles would need one and a half waves of their 1024 slots
        static const int full_lo = getenv("SGAN_FULL_LO") ? atoi(getenv("SGAN_FULL_LO")) : 620;
        const bool one_wave = blocks >= full_lo && blocks <= 768;
        if (!one_wave && blocks <= half && nkt <= half_nkt && sg_plan_ksplit(P, 64, 64) == 1) return sg_launch_igemm<64, 32, 2, 2>(P, st, ws, ws_bytes);
    }
    return sg_launch_igemm<64, 64, 2, 2>(P, st, ws, ws_bytes);
}


static int64_t sg_workspace_need(const SgIgemmParams& P) {
    if (sg_use_small_n(P)) return 0;
    if (sg_igemm3_eligible(P) > 0) return sg_igemm3_workspace_need(P);
    if (sg_use_c4(P)) return 0;
    int BM, BN;
    sg_pick_tile(P, &BM, &BN);
    const int ks = sg_plan_ksplit(P, BM, BN);
    return ks > 1 ? (int64_t)ks * P.q[0].Hout * P.q[0].Wout * P.N * 4 : 0;
}

static int sg_check_common(const sgan_conv_desc* d) {
    if (!d) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    if ((int64_t)d->Hin * d->Win * d->Cin >= (1ll << 28) || (int64_t)d->Hout * d->Wout * d->Cout >= (1ll << 28) ||
        (int64_t)d->k * d->k * d->Cin * d->Cout >= (1ll << 28))
        return sgan_fail(SGAN_ERR_UNSUPPORTED, "tensor too large for 31-bit byte offsets");
    if ((d->Cin & 3) || (d->Cout & 3)) return sgan_fail(SGAN_ERR_INVALID, "stored channels must be multiples of 4 (Cin %d Cout %d)", d->Cin, d->Cout);
    if (d->Hin <= 0 || d->Win <= 0 || d->Hout <= 0 || d->Wout <= 0) return sgan_fail(SGAN_ERR_INVALID, "empty tensor");
    return SGAN_OK;
}

static bool sg_same_layer(const sgan_conv_desc* a, const sgan_conv_desc* b) {
    return a->kind == b->kind && a->k == b->k && a->stride == b->stride && a->pad == b->pad && a->Cin == b->Cin && a->Cout == b->Cout &&
           a->Cin_logical == b->Cin_logical && a->Cout_logical == b->Cout_logical;
}

// common part of a group from its first descriptor; per-problem phase sizes from each descriptor
static int sg_group_geometry(SgIgemmParams& P, const sgan_conv_desc* const* descs, int n, bool dgrad) {
    if (n < 1 || n > SG_MAX_PROB) return sgan_fail(SGAN_ERR_INVALID, "1..%d problems per grouped launch", SG_MAX_PROB);
    memset(&P, 0, sizeof(P));
    P.nprob = n;
    {
        const int nl = dgrad ? descs[0]->Cin_logical : descs[0]->Cout_logical, ns = dgrad ? descs[0]->Cin : descs[0]->Cout;
        P.n_real = (nl > 0 && nl <= ns) ? nl : ns;
    }
    for (int g = 0; g < n; ++g) {
        int rc = sg_check_common(descs[g]);
        if (rc) return rc;
        if (!sg_same_layer(descs[0], descs[g])) return sgan_fail(SGAN_ERR_INVALID, "grouped problems must be the same layer type");
        SgPhase ph[SGAN_MAX_PHASES];
        int nphase, is, os;
        rc = sg_build_phases(descs[g], dgrad, ph, &nphase, &is, &os);
        if (rc) return rc;
        if (g == 0) {
            P.nphase = nphase; P.is = is; P.os = os;
            for (int i = 0; i < nphase; ++i) {
                P.oa[i] = ph[i].oa; P.ob[i] = ph[i].ob; P.ntaps[i] = ph[i].ntaps; P.ktot[i] = ph[i].ktot;
                P.tap0[i] = i == 0 ? 0 : P.tap0[i - 1] + ph[i - 1].ntaps;
                for (int t = 0; t < ph[i].ntaps; ++t) P.taps[P.tap0[i] + t] = ph[i].taps[t];
            }
        }
        for (int i = 0; i < nphase; ++i) { P.q[g].Hp[i] = ph[i].Hp; P.q[g].Wp[i] = ph[i].Wp; }
    }
    return SGAN_OK;
}

static void sg_set_norm(const sgan_norm_desc* d, const double** stats, const float** gamma, const float** beta, int32_t* count,
                        int32_t* sq, int32_t* rep) {
    *stats = d ? d->stats : nullptr;
    *gamma = d ? d->gamma : nullptr;
    *beta = d ? d->beta : nullptr;
    *count = d ? d->count : 1;
    *sq = d ? d->sq_stride : 0;
    *rep = d ? d->rep_stride : 0;
}

extern "C" int sgan_conv_fwd_grouped(const sgan_conv_fwd_job* jobs, int32_t n, int32_t out_act, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    SGAN_CHECK(jobs && n >= 1 && n <= SG_MAX_PROB, "1..%d jobs", SG_MAX_PROB);
    SGAN_CHECK(out_act == SGAN_ACT_NONE || out_act == SGAN_ACT_TANH, "out_act must be none or tanh");
    const sgan_conv_desc* descs[SG_MAX_PROB];
    for (int g = 0; g < n; ++g) descs[g] = jobs[g].d;
    SgIgemmParams P;
    int rc = sg_group_geometry(P, descs, n, false);
    if (rc) return rc;
    const sgan_conv_desc* d0 = jobs[0].d;
    P.Ck = d0->Cin; P.N = d0->Cout;
    P.w_ns = d0->Cin; P.w_ks = 1;  // B[k=ci][n=co] = W[tap][co][ci]
    P.math = d0->math;
    P.planes_f16 = 1;      // forward: post-normalisation activations and 2^10-scaled weights fit fp16's range: fp32-equivalent products
    P.out_act = out_act;
    const sgan_norm_desc* n0 = jobs[0].in_norm;
    P.pro_act = n0 ? n0->act : SGAN_ACT_NONE; P.pro_slope = n0 ? n0->slope : 0.f; P.pro_eps = n0 ? n0->eps : 0.f;
    SGAN_CHECK(P.pro_act != SGAN_ACT_LRELU || P.pro_slope <= 1.f, "LeakyReLU slope must be <= 1 (the kernels evaluate max(y, slope * y))");
    P.xn_act = SGAN_ACT_NONE;
    for (int g = 0; g < n; ++g) {
        const sgan_conv_fwd_job& J = jobs[g];
        SGAN_CHECK(J.in && J.w && J.out, "null tensor in job %d", g);
        SGAN_CHECK(J.in_ld >= J.d->Cin && J.out_ld >= J.d->Cout && (J.in_ld & 3) == 0, "bad leading dims in job %d", g);
        SGAN_CHECK((J.in_norm ? J.in_norm->act : SGAN_ACT_NONE) == P.pro_act, "grouped jobs must share the prologue activation");
        SgProb& Q = P.q[g];
        SGAN_CHECK(J.d->math == d0->math, "grouped jobs must share the math mode");
        Q.in = J.in; Q.out = J.out; Q.w = J.w; Q.wp = J.w_packed; Q.bias = J.bias; Q.xref = nullptr; Q.stats = J.out_stats;
        Q.amax = nullptr;
        Q.Hin = J.d->Hin; Q.Win = J.d->Win; Q.in_ld = J.in_ld; Q.Hout = J.d->Hout; Q.Wout = J.d->Wout; Q.out_ld = J.out_ld;
        sg_set_norm(J.in_norm, &Q.pro_stats, &Q.pro_gamma, &Q.pro_beta, &Q.pro_count, &Q.pro_sq, &Q.pro_rep);
        Q.xn_count = 1;
        Q.stats_sq = J.out_stats_sq_stride;
        Q.stats_rep = J.out_stats_rep_stride;
    }
    if (workspace_bytes == -1) return (int)(sg_workspace_need(P) >> 10) + (sg_workspace_need(P) ? 1 : 0);   // query (KiB)
    return sg_dispatch_igemm(P, (hipStream_t)stream, (float*)workspace, workspace_bytes);
}

bool sg_dgrad_is_skinny(const SgIgemmParams& P) { return sg_use_small_n(P); }

int sg_build_dgrad_params(const sgan_conv_dgrad_job* jobs, int32_t n, SgIgemmParams& P, bool allow_f16) {
    SGAN_CHECK(jobs && n >= 1 && n <= SG_MAX_PROB, "1..%d jobs", SG_MAX_PROB);
    const sgan_conv_desc* descs[SG_MAX_PROB];
    for (int g = 0; g < n; ++g) descs[g] = jobs[g].d;
    int rc = sg_group_geometry(P, descs, n, true);
    if (rc) return rc;
    const sgan_conv_desc* d0 = jobs[0].d;
    P.Ck = d0->Cout; P.N = d0->Cin;
    if (jobs[0].w_transposed) { P.w_ns = d0->Cout; P.w_ks = 1; }   // B[k=co][n=ci] = Wt[tap][ci][co]: k contiguous
    else { P.w_ns = 1; P.w_ks = d0->Cin; }                          // B[k=co][n=ci] = W[tap][co][ci]: n contiguous
    P.math = d0->math;
    P.out_act = SGAN_ACT_NONE;
    P.pro_act = SGAN_ACT_NONE;
    const sgan_norm_desc* x0 = jobs[0].x ? jobs[0].x_norm : nullptr;
    P.xn_act = x0 ? x0->act : SGAN_ACT_NONE; P.xn_slope = x0 ? x0->slope : 0.f; P.xn_eps = x0 ? x0->eps : 0.f;
    // fp16 planes (an fp32-equivalent product, as in the forward pass) when every job brings its gradient's maximum and the fp16 copy
    // of its weights; else bf16 planes
    bool f16 = allow_f16;
    for (int g = 0; g < n; ++g) f16 = f16 && jobs[g].dout_amax && jobs[g].w_packed_f16 && jobs[g].w_transposed;
    static const int no_f16 = getenv("SGAN_NO_F16_BWD") ? atoi(getenv("SGAN_NO_F16_BWD")) : 0;      // diagnostics: bf16 planes as in round 2
    if (no_f16) f16 = false;
    P.planes_f16 = f16 ? 1 : 0;
    for (int g = 0; g < n; ++g) {
        const sgan_conv_dgrad_job& J = jobs[g];
        SGAN_CHECK(J.dout && J.w && J.din, "null tensor in job %d", g);
        SGAN_CHECK(J.dout_ld >= J.d->Cout && J.din_ld >= J.d->Cin && (J.dout_ld & 3) == 0, "bad leading dims in job %d", g);
        SGAN_CHECK(!J.x || J.x_ld >= J.d->Cin, "bad x_ld in job %d", g);
        SGAN_CHECK(!(J.bwd_sums && !J.x), "bwd_sums needs x (job %d)", g);
        SGAN_CHECK((J.x != nullptr) == (jobs[0].x != nullptr), "grouped jobs must all have / all lack the forward tensor");
        SGAN_CHECK((J.w_transposed != 0) == (jobs[0].w_transposed != 0), "grouped jobs must use the same weight layout");
        const sgan_norm_desc* xn = J.x ? J.x_norm : nullptr;
        SGAN_CHECK((xn ? xn->act : SGAN_ACT_NONE) == P.xn_act, "grouped jobs must share the activation");
        SgProb& Q = P.q[g];
        SGAN_CHECK(J.d->math == d0->math, "grouped jobs must share the math mode");
        Q.in = J.dout; Q.out = J.din; Q.w = J.w; Q.wp = J.w_transposed ? (f16 ? J.w_packed_f16 : J.w_packed) : nullptr; Q.bias = nullptr; Q.xref = J.x; Q.stats = J.bwd_sums;
        Q.amax = f16 ? J.dout_amax : nullptr;
        Q.Hin = J.d->Hout; Q.Win = J.d->Wout; Q.in_ld = J.dout_ld; Q.Hout = J.d->Hin; Q.Wout = J.d->Win; Q.out_ld = J.din_ld;
        Q.xref_ld = J.x_ld;
        Q.pro_count = 1;
        sg_set_norm(xn, &Q.xn_stats, &Q.xn_gamma, &Q.xn_beta, &Q.xn_count, &Q.xn_sq, &Q.xn_rep);
        Q.stats_sq = J.bwd_sums_sq_stride;
        Q.stats_rep = J.bwd_sums_rep_stride;
        Q.accum = J.accumulate;
    }
    return SGAN_OK;
}

extern "C" int sgan_conv_dgrad_grouped(const sgan_conv_dgrad_job* jobs, int32_t n, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
    SgIgemmParams P;
    int rc = sg_build_dgrad_params(jobs, n, P);
    if (rc) return rc;
    if (workspace_bytes == -1) return (int)(sg_workspace_need(P) >> 10) + (sg_workspace_need(P) ? 1 : 0);   // query (KiB)
    return sg_dispatch_igemm(P, (hipStream_t)stream, (float*)workspace, workspace_bytes);
}

extern "C" int sgan_conv_fwd(const sgan_conv_desc* d, const float* in, int32_t in_ld, const sgan_norm_desc* in_norm,
                             const float* w, const float* bias, float* out, int32_t out_ld, int32_t out_act,
                             double* out_stats, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!d) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    sgan_conv_fwd_job j = {d, in, in_ld, in_norm, w, bias, out, out_ld, out_stats, 0, nullptr};   // no packed copy: fp32 kernels
    return sgan_conv_fwd_grouped(&j, 1, out_act, workspace, workspace_bytes, stream);
}

extern "C" int sgan_conv_dgrad(const sgan_conv_desc* d, const float* dout, int32_t dout_ld, const float* w,
                               float* din, int32_t din_ld, const float* x, int32_t x_ld, const sgan_norm_desc* x_norm,
                               double* bwd_sums, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!d) return sgan_fail(SGAN_ERR_INVALID, "null desc");
    sgan_conv_dgrad_job j = {d, dout, dout_ld, w, din, din_ld, x, x_ld, x_norm, bwd_sums, 0, 0, 0, nullptr, 0, nullptr, nullptr};
    return sgan_conv_dgrad_grouped(&j, 1, workspace, workspace_bytes, stream);
}
#endif      // SG_KERNELS_ONLY
