// HBM-bound kernels of the hot path for gfx950: norm backward, BatchNorm running statistics,
// Gaussian pre-filter (strided, diagonal only), GAN loss on the logits map, tanh backward, layout
// boundary copy, multi-segment Adam, Philox normal fill.  All are 16-byte-per-lane streaming
// kernels (or tiny single-workgroup reductions); none allocates or synchronises.
#include "sgan_common.h"

thread_local char g_sgan_err[512] = {0};
thread_local const char* g_sgan_last_kernel = "";

int sgan_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_sgan_err, sizeof(g_sgan_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* sgan_version(void) { return "sgan_hip 0.2 (gfx950; fp32 MFMA 16x16x4 + split-bf16 MFMA 32x32x16)"; }
extern "C" const char* sgan_last_error(void) { return g_sgan_err; }
extern "C" const char* sgan_last_kernel(void) { return g_sgan_last_kernel; }
extern "C" int sgan_stat_replicas(void) { return SGAN_STAT_REPLICAS; }      // what this build was compiled with (callers size their arenas by it)
// The explicit device of the boundary: kernels go to the device the stream belongs to, and HIP wants the calling thread's
// current device to be that one.  A host thread that has not chosen a device itself (an autograd worker of a non-torch host, a
// thread pool) calls this before its first entry point.
extern "C" int sgan_set_device(int32_t device_id) {
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // HIP keeps the error sticky: the next launch check of this thread would report it again
        return sgan_fail(SGAN_ERR_HIP, "hipSetDevice(%d): %s", (int)device_id, hipGetErrorString(e));
    }
    return SGAN_OK;
}
extern "C" int sgan_stream_device(void* stream, int32_t* device_id) {      // which device a stream's launches will run on
    if (!device_id) return sgan_fail(SGAN_ERR_INVALID, "null device_id");
    hipDevice_t dev;
    hipError_t e = hipStreamGetDevice((hipStream_t)stream, &dev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return sgan_fail(SGAN_ERR_HIP, "hipStreamGetDevice: %s", hipGetErrorString(e));
    }
    *device_id = (int32_t)dev;
    return SGAN_OK;
}

// ---- optional per-launch timing -------------------------------------------------------------
#define SG_PROF_MAX 8192
static bool g_prof_on = false;
static int g_prof_n = 0;
static hipEvent_t g_prof_e0[SG_PROF_MAX], g_prof_e1[SG_PROF_MAX];
static const char* g_prof_name[SG_PROF_MAX];
static bool g_prof_open = false;

void sg_prof_begin(hipStream_t st) {
    if (!g_prof_on || g_prof_n >= SG_PROF_MAX) return;
    if (hipEventCreate(&g_prof_e0[g_prof_n]) != hipSuccess || hipEventCreate(&g_prof_e1[g_prof_n]) != hipSuccess) return;
    (void)hipEventRecord(g_prof_e0[g_prof_n], st);
    g_prof_open = true;
}

void sg_prof_end(hipStream_t st, const char* name) {
    if (!g_prof_open) return;
    (void)hipEventRecord(g_prof_e1[g_prof_n], st);
    g_prof_name[g_prof_n] = name;
    ++g_prof_n;
    g_prof_open = false;
}

extern "C" int sgan_profile_enable(int on) {   // single-threaded diagnostic facility; resets the record list
    for (int i = 0; i < g_prof_n; ++i) {
        (void)hipEventDestroy(g_prof_e0[i]);
        (void)hipEventDestroy(g_prof_e1[i]);
    }
    g_prof_n = 0;
    g_prof_on = on != 0;
    return SGAN_OK;
}

extern "C" int sgan_profile_count(void) { return g_prof_n; }

// an empty bracket ("null"): the fixed cost of the two event records themselves, for calibration
extern "C" int sgan_profile_mark(void* stream) {
    sg_prof_begin((hipStream_t)stream);
    sg_prof_end((hipStream_t)stream, "null");
    return SGAN_OK;
}

extern "C" int sgan_profile_read(int i, const char** name, float* ms) {   // caller synchronised the device first
    if (i < 0 || i >= g_prof_n || !name || !ms) return sgan_fail(SGAN_ERR_INVALID, "bad profile record index");
    *name = g_prof_name[i];
    if (hipEventElapsedTime(ms, g_prof_e0[i], g_prof_e1[i]) != hipSuccess) return sgan_fail(SGAN_ERR_HIP, "hipEventElapsedTime failed");
    return SGAN_OK;
}

static inline int ew_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// InstanceNorm / BatchNorm(batch 1) backward, in place on dy
// ------------------------------------------------------------------------------------------
struct SgNormBwdJob {
    float* dy; const float* x; const double* sums; float* dgamma; float* dbeta; unsigned* amax;
    SgNorm xn;
    int32_t dy_ld, x_ld, npix, C, sums_sq, sums_rep, blocks;
};
struct SgNormBwdTable { SgNormBwdJob j[8]; };

// blockIdx.y = job; every job strides over its own elements with its own block count (<= gridDim.x)
__global__ __launch_bounds__(256) void sg_norm_bwd_apply_kernel(const SgNormBwdTable T) {
    sg_warm_kernargs<(int)sizeof(SgNormBwdTable)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    const SgNormBwdJob& J = T.j[blockIdx.y];
    if ((int)blockIdx.x >= J.blocks) return;
    const int C = J.C;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cA = reinterpret_cast<float*>(smem);  // gamma * rstd
    float* cMean = cA + C;
    float* cRstd = cMean + C;
    float* cS1 = cRstd + C;  // s1 / M
    float* cS2 = cS1 + C;    // s2 / M
    const double invM = 1.0 / (double)J.npix;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mean, rstd;
        sg_mean_rstd(J.xn, C, c, mean, rstd);
        const float g = J.xn.gamma ? J.xn.gamma[c] : 1.f;
        cA[c] = g * rstd;
        cMean[c] = mean;
        cRstd[c] = rstd;
        const double s1 = sg_stat_sum(J.sums, c, J.sums_rep), s2 = sg_stat_sum(J.sums, J.sums_sq + c, J.sums_rep);
        cS1[c] = (float)(s1 * invM);
        cS2[c] = (float)(s2 * invM);
        if (blockIdx.x == 0) {
            if (J.dgamma) atomicAdd(&J.dgamma[c], (float)s2);   // concurrent chains may share the buffer
            if (J.dbeta) atomicAdd(&J.dbeta[c], (float)s1);
        }
    }
    SG_SYNC();
    const int CQ = C >> 2;
    const int64_t total = (int64_t)J.npix * CQ;
    float* dy = J.dy;
    const float* x = J.x;
    // four chunks per trip, all eight loads in flight before the first store: dy is updated in place, so the compiler keeps every
    // load behind the previous trip's store and a one-chunk loop waits out a memory round trip per chunk
    const int64_t stride = (int64_t)J.blocks * 256;
    float amax = 0.f;      // max |result| of this thread (published below: the fp16-plane scale of the kernels that read dy next)
    for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += 4 * stride) {
        f32x4 d[4], xv[4];
        int64_t od[4];
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = e0 + u * stride;
            const bool ok = e < total;
            const int p = ok ? (int)(e / CQ) : 0;
            cc[u] = ok ? (int)(e - (int64_t)p * CQ) * 4 : 0;
            od[u] = ok ? (int64_t)p * J.dy_ld + cc[u] : -1;
            d[u] = *reinterpret_cast<const f32x4*>(dy + (ok ? od[u] : 0));
            xv[u] = *reinterpret_cast<const f32x4*>(x + (ok ? (int64_t)p * J.x_ld + cc[u] : 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (od[u] < 0) continue;
            const int c = cc[u];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xhat = (xv[u][j] - cMean[c + j]) * cRstd[c + j];
                d[u][j] = cA[c + j] * (d[u][j] - cS1[c + j] - xhat * cS2[c + j]);
                amax = fmaxf(amax, fabsf(d[u][j]));
            }
            *reinterpret_cast<f32x4*>(dy + od[u]) = d[u];
        }
    }
    if (J.amax) {      // non-negative floats order like their bit patterns: an integer atomic max -- one per WORKGROUP, and only when the
        // published value (read past the L1) is still below this workgroup's: thousands of same-address atomics cost 12 ns each
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        SG_SYNC();      // the coefficient arrays are free
        if ((threadIdx.x & 63) == 0) cA[threadIdx.x >> 6] = amax;
        SG_SYNC();
        if (threadIdx.x == 0) {
            const unsigned mine = __builtin_bit_cast(unsigned, fmaxf(fmaxf(cA[0], cA[1]), fmaxf(cA[2], cA[3])));
            if (mine > __hip_atomic_load(J.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(J.amax, mine);
        }
    }
}

extern "C" int sgan_norm_bwd_apply_multi(const sgan_norm_bwd_job* jobs, int32_t n, void* stream) {
    SGAN_CHECK(jobs && n >= 1 && n <= 8, "1..8 jobs");
    SgNormBwdTable T;
    int maxb = 1, maxC = 4;
    for (int i = 0; i < n; ++i) {
        const sgan_norm_bwd_job& S = jobs[i];
        SGAN_CHECK(S.dy && S.x && S.x_norm && S.x_norm->stats && S.bwd_sums, "null argument in job %d", i);
        SGAN_CHECK((S.C & 3) == 0 && S.C > 0 && S.C <= 4096 && S.dy_ld >= S.C && S.x_ld >= S.C && (S.dy_ld & 3) == 0 &&
                       (S.x_ld & 3) == 0 && S.npix > 0, "bad dims in job %d", i);
        SgNormBwdJob& J = T.j[i];
        J.dy = S.dy; J.x = S.x; J.sums = S.bwd_sums; J.dgamma = S.dgamma; J.dbeta = S.dbeta;
        J.amax = reinterpret_cast<unsigned*>(S.amax_out);
        J.xn = sg_norm_from(S.x_norm);
        J.dy_ld = S.dy_ld; J.x_ld = S.x_ld; J.npix = S.npix; J.C = S.C;
        J.sums_sq = S.bwd_sums_sq_stride ? S.bwd_sums_sq_stride : S.C;
        J.sums_rep = S.bwd_sums_rep_stride;
        const int64_t total = (int64_t)S.npix * (S.C >> 2);
        // every workgroup first turns the replica sums of ALL channels into coefficients (32 fp64 loads + a divide and a root per
        // channel): few, fat workgroups amortise that (SGAN_NBA_CHUNK 16-byte chunks per thread: tuning knob)
        static const int chunk = getenv("SGAN_NBA_CHUNK") ? atoi(getenv("SGAN_NBA_CHUNK")) : 4;
        int blocks = ew_cdiv(total, 256 * (chunk > 0 ? chunk : 4));
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        J.blocks = blocks;
        if (blocks > maxb) maxb = blocks;
        if (S.C > maxC) maxC = S.C;
    }
    hipLaunchKernelGGL(sg_norm_bwd_apply_kernel, dim3(maxb, n), dim3(256), (size_t)5 * maxC * 4, (hipStream_t)stream, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_norm_bwd_apply(float* dy, int32_t dy_ld, const float* x, int32_t x_ld, int32_t npix, int32_t C,
                                   const sgan_norm_desc* x_norm, const double* bwd_sums, int32_t bwd_sums_sq_stride,
                                   float* dgamma, float* dbeta, void* stream) {
    sgan_norm_bwd_job j = {dy, dy_ld, x, x_ld, npix, C, x_norm, bwd_sums, bwd_sums_sq_stride, dgamma, dbeta, 0, nullptr};
    return sgan_norm_bwd_apply_multi(&j, 1, stream);
}

// ------------------------------------------------------------------------------------------
// U-Net up path: normalise (+ dropout) (+ noise) into a concat slice; and the first backward pass
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_norm_apply_fwd_kernel(const float* u, int u_ld, SgNorm un, const float* mask,
                                                                const float* noise, float sigma, float* t, int t_ld, int npix,
                                                                int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cMean = reinterpret_cast<float*>(smem);
    float* cA = cMean + C;         // gamma * rstd     (gamma, beta: the BatchNorm affine; absent for InstanceNorm)
    float* cB = cA + C;            // beta
    for (int c = threadIdx.x; c < C; c += 256) {
        float mean = 0.f, rstd = 1.f;
        if (un.stats) sg_mean_rstd(un, C, c, mean, rstd);
        cMean[c] = mean;
        cA[c] = (un.gamma ? un.gamma[c] : 1.f) * rstd;
        cB[c] = un.beta ? un.beta[c] : 0.f;
    }
    SG_SYNC();
    const int CQ = C >> 2;
    const int64_t total = (int64_t)npix * CQ;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int p = (int)(e / CQ), c = (int)(e - (int64_t)p * CQ) * 4;
        const f32x4 x = *reinterpret_cast<const f32x4*>(u + (int64_t)p * u_ld + c);
        f32x4 m = (f32x4){1.f, 1.f, 1.f, 1.f}, nz = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (mask) m = *reinterpret_cast<const f32x4*>(mask + (int64_t)p * C + c);
        if (noise) nz = *reinterpret_cast<const f32x4*>(noise + (int64_t)p * C + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = ((x[j] - cMean[c + j]) * cA[c + j] + cB[c + j]) * m[j] + sigma * nz[j];
        *reinterpret_cast<f32x4*>(t + (int64_t)p * t_ld + c) = o;
    }
}

// thread t always works on channel group t % (C/4) (host: 256 % (C/4) == 0 or the generic path below)
__global__ __launch_bounds__(256) void sg_norm_apply_bwd_sums_kernel(float* dt, int dt_ld, const float* mask, const float* u,
                                                                     int u_ld, SgNorm un, double* sums, int npix, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cMean = reinterpret_cast<float*>(smem);
    float* cRstd = cMean + C;
    float* red = cRstd + C;  // [2C]
    for (int c = threadIdx.x; c < C; c += 256) {
        float mean = 0.f, rstd = 1.f;
        if (un.stats) sg_mean_rstd(un, C, c, mean, rstd);
        cMean[c] = mean;
        cRstd[c] = rstd;
    }
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
    SG_SYNC();
    const int CQ = C >> 2;
    const int64_t total = (int64_t)npix * CQ;
    if (256 % CQ == 0) {
        // the element stride (gridDim.x * 256) is a multiple of CQ: this thread always sees channel group tid % CQ, so the
        // two sums live in registers and reach LDS once per thread (an LDS atomic per element was the whole cost of this kernel)
        const int c = (threadIdx.x % CQ) * 4;
        const f32x4 mean = *reinterpret_cast<const f32x4*>(cMean + c), rstd = *reinterpret_cast<const f32x4*>(cRstd + c);
        f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int64_t p = e / CQ;
            f32x4 d = *reinterpret_cast<const f32x4*>(dt + p * dt_ld + c);
            const f32x4 x = *reinterpret_cast<const f32x4*>(u + p * u_ld + c);
            if (mask) {
                d *= *reinterpret_cast<const f32x4*>(mask + p * C + c);
                *reinterpret_cast<f32x4*>(dt + p * dt_ld + c) = d;
            }
            s1 += d;
            s2 += d * ((x - mean) * rstd);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&red[c + j], s1[j]);
            atomicAdd(&red[C + c + j], s2[j]);
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int p = (int)(e / CQ), c = (int)(e - (int64_t)p * CQ) * 4;
            f32x4 d = *reinterpret_cast<const f32x4*>(dt + (int64_t)p * dt_ld + c);
            const f32x4 x = *reinterpret_cast<const f32x4*>(u + (int64_t)p * u_ld + c);
            if (mask) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(mask + (int64_t)p * C + c);
                d *= m;
                *reinterpret_cast<f32x4*>(dt + (int64_t)p * dt_ld + c) = d;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xhat = (x[j] - cMean[c + j]) * cRstd[c + j];
                atomicAdd(&red[c + j], d[j]);
                atomicAdd(&red[C + c + j], d[j] * xhat);
            }
        }
    }
    SG_SYNC();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(&sums[c], (double)red[c]);
        atomicAdd(&sums[C + c], (double)red[C + c]);
    }
}

extern "C" int sgan_norm_apply_fwd(const float* u, int32_t u_ld, const sgan_norm_desc* u_norm, const float* mask,
                                   const float* noise, float sigma, float* t, int32_t t_ld, int32_t npix, int32_t C, void* stream) {
    SGAN_CHECK(u && t && npix > 0 && C > 0 && (C & 3) == 0 && u_ld >= C && t_ld >= C && (u_ld & 3) == 0 && (t_ld & 3) == 0, "bad argument");
    const int64_t total = (int64_t)npix * (C >> 2);
    int blocks = ew_cdiv(total, 256 * 2);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sg_norm_apply_fwd_kernel, dim3(blocks), dim3(256), (size_t)3 * C * 4, (hipStream_t)stream, u, u_ld,
                       sg_norm_from(u_norm), mask, noise, sigma, t, t_ld, npix, C);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_norm_apply_bwd_sums(float* dt, int32_t dt_ld, const float* mask, const float* u, int32_t u_ld,
                                        const sgan_norm_desc* u_norm, double* bwd_sums, int32_t npix, int32_t C, void* stream) {
    SGAN_CHECK(dt && u && bwd_sums && npix > 0 && C > 0 && (C & 3) == 0 && u_ld >= C && dt_ld >= C, "bad argument");
    const int64_t total = (int64_t)npix * (C >> 2);
    int blocks = ew_cdiv(total, 256 * 8);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sg_norm_apply_bwd_sums_kernel, dim3(blocks), dim3(256), (size_t)4 * C * 4, (hipStream_t)stream, dt, dt_ld,
                       mask, u, u_ld, sg_norm_from(u_norm), bwd_sums, npix, C);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// nn.ReflectionPad2d in front of a conv (resnet generators, models/networks.py:232,258,282-300): the padded tensor is
// materialised together with everything the reference applies before the padding --
//     out[py][px][c] = mask * act(norm(x[refl(py - pad)][refl(px - pad)][c])),    refl(i) = |i| for i < 0, 2 (n - 1) - i for i >= n
// (pad = 0: plain materialisation of act(norm(x)), e.g. the first resnet block's input).  The conv that follows runs with pad 0
// and no prologue.  Backward: the gradient of the padded tensor is folded back (every interior pixel collects the <= 4 padded
// positions that mirror it), multiplied by mask and act'(norm(x)), and the two norm-backward sums are accumulated (as the
// backward-data epilogue of the conv kernels does); sgan_norm_bwd_apply finishes.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int sg_refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

__global__ __launch_bounds__(256) void sg_pad_reflect_fwd_kernel(const float* x, int x_ld, int H, int W, int C, SgNorm xn, const float* mask,
                                                                 int pad, float* out, int out_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cSc = reinterpret_cast<float*>(smem);
    float* cSh = cSc + C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float sc = 1.f, sh = 0.f;
        if (xn.stats) {
            float mean, rstd;
            sg_mean_rstd(xn, C, c, mean, rstd);
            const float gm = xn.gamma ? xn.gamma[c] : 1.f, bt = xn.beta ? xn.beta[c] : 0.f;
            sc = gm * rstd;
            sh = bt - mean * sc;
        }
        cSc[c] = sc;
        cSh[c] = sh;
    }
    SG_SYNC();
    const int CQ = C >> 2, Wp = W + 2 * pad, Hp = H + 2 * pad;
    const int64_t total = (int64_t)Hp * Wp * CQ;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % CQ) * 4;
        const int64_t pp = e / CQ;
        const int px = (int)(pp % Wp), py = (int)(pp / Wp);
        const int64_t src = (int64_t)sg_refl(py - pad, H) * W + sg_refl(px - pad, W);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + src * x_ld + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sg_act(v[j] * cSc[c + j] + cSh[c + j], xn.act, xn.slope);
        if (mask) v *= *reinterpret_cast<const f32x4*>(mask + src * C + c);
        *reinterpret_cast<f32x4*>(out + pp * out_ld + c) = v;
    }
}

extern "C" int sgan_pad_reflect_fwd(const float* x, int32_t x_ld, int32_t H, int32_t W, int32_t C, const sgan_norm_desc* x_norm,
                                    const float* mask, int32_t pad, float* out, int32_t out_ld, void* stream) {
    SGAN_CHECK(x && out && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (x_ld & 3) == 0 && (out_ld & 3) == 0 && x_ld >= C && out_ld >= C,
               "bad argument");
    SGAN_CHECK(pad >= 0 && pad < H && pad < W, "reflection padding must be smaller than the image");
    const int64_t total = (int64_t)(H + 2 * pad) * (W + 2 * pad) * (C >> 2);
    int blocks = ew_cdiv(total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sg_pad_reflect_fwd_kernel, dim3(blocks), dim3(256), (size_t)2 * C * 4, (hipStream_t)stream, x, x_ld, H, W, C,
                       sg_norm_from(x_norm), mask, pad, out, out_ld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

__global__ __launch_bounds__(256) void sg_pad_reflect_bwd_kernel(const float* dout, int dout_ld, int H, int W, int C, int pad, const float* x,
                                                                 int x_ld, SgNorm xn, const float* mask, float* din, int din_ld,
                                                                 double* sums, int sums_sq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cMean = reinterpret_cast<float*>(smem);
    float* cRstd = cMean + C;
    float* cG = cRstd + C;
    float* cB = cG + C;
    double* red = reinterpret_cast<double*>(cB + C);   // [2C]
    for (int c = threadIdx.x; c < C; c += 256) {
        float mean = 0.f, rstd = 1.f;
        if (xn.stats) sg_mean_rstd(xn, C, c, mean, rstd);
        cMean[c] = mean;
        cRstd[c] = rstd;
        cG[c] = (xn.stats && xn.gamma) ? xn.gamma[c] : 1.f;
        cB[c] = (xn.stats && xn.beta) ? xn.beta[c] : 0.f;
    }
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.0;
    SG_SYNC();
    const int CQ = C >> 2, Wp = W + 2 * pad;
    const int64_t total = (int64_t)H * W * CQ;
    const bool has_x = x != nullptr;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % CQ) * 4;
        const int64_t pix = e / CQ;
        const int xx = (int)(pix % W), yy = (int)(pix / W);
        // padded rows / columns that mirror this pixel: itself, and one reflection at most on each side (pad < H / 2 checked on the host)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = yy + pad;
        if (yy >= 1 && yy <= pad) ys[ny++] = pad - yy;
        if (yy <= H - 2 && yy >= H - 1 - pad) ys[ny++] = pad + 2 * (H - 1) - yy;
        xs[nx++] = xx + pad;
        if (xx >= 1 && xx <= pad) xs[nx++] = pad - xx;
        if (xx <= W - 2 && xx >= W - 1 - pad) xs[nx++] = pad + 2 * (W - 1) - xx;
        f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) d += *reinterpret_cast<const f32x4*>(dout + ((int64_t)ys[a] * Wp + xs[b]) * dout_ld + c);
        if (mask) d *= *reinterpret_cast<const f32x4*>(mask + pix * C + c);
        if (has_x) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * x_ld + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xhat = (xv[j] - cMean[c + j]) * cRstd[c + j];
                const float y = xn.stats ? cG[c + j] * xhat + cB[c + j] : xv[j];
                d[j] *= sg_act_grad(y, xn.act, xn.slope);
                if (sums) {
                    atomicAdd(&red[c + j], (double)d[j]);
                    atomicAdd(&red[C + c + j], (double)(d[j] * xhat));
                }
            }
        }
        *reinterpret_cast<f32x4*>(din + pix * din_ld + c) = d;
    }
    if (sums) {
        SG_SYNC();
        for (int c = threadIdx.x; c < C; c += 256) {
            atomicAdd(&sums[c], red[c]);
            atomicAdd(&sums[(sums_sq ? sums_sq : C) + c], red[C + c]);
        }
    }
}

extern "C" int sgan_pad_reflect_bwd(const float* dout, int32_t dout_ld, int32_t H, int32_t W, int32_t C, int32_t pad, const float* x,
                                    int32_t x_ld, const sgan_norm_desc* x_norm, const float* mask, float* din, int32_t din_ld,
                                    double* bwd_sums, int32_t bwd_sums_sq_stride, void* stream) {
    SGAN_CHECK(dout && din && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (dout_ld & 3) == 0 && (din_ld & 3) == 0 && dout_ld >= C && din_ld >= C,
               "bad argument");
    SGAN_CHECK(pad >= 0 && 2 * pad < H && 2 * pad < W, "reflection padding must be smaller than half the image");
    SGAN_CHECK(!(bwd_sums && !x), "bwd_sums needs x");
    SGAN_CHECK(!x || ((x_ld & 3) == 0 && x_ld >= C), "bad x_ld");
    const int64_t total = (int64_t)H * W * (C >> 2);
    int blocks = ew_cdiv(total, 256 * 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sg_pad_reflect_bwd_kernel, dim3(blocks), dim3(256), (size_t)4 * C * 4 + (size_t)2 * C * 8, (hipStream_t)stream, dout,
                       dout_ld, H, W, C, pad, x, x_ld, sg_norm_from(x ? x_norm : nullptr), mask, din, din_ld, bwd_sums, bwd_sums_sq_stride);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// weighted L1 / BCE on rescaled maps: image-sized losses.  One pixel per thread over up to SG_IMGLOSS_BLOCKS workgroups
// (a single workgroup walking 512x512 pixels took 400 us); block partial sums (fp64) go to the caller's scratch and a
// one-wave kernel finishes the mean -- a kernel boundary instead of atomics or a zero-initialised accumulator.
// ------------------------------------------------------------------------------------------
#define SG_IMGLOSS_BLOCKS 256
__device__ __forceinline__ void sg_block_partial(double acc, double* part) {
    __shared__ double wsum[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    SG_SYNC();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(64) void sg_imgloss_fin_kernel(const double* part, int nparts, double scale, float* loss_out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) loss_out[0] = (float)(acc * scale);
}

__global__ __launch_bounds__(256) void sg_l1w_fwd_kernel(const float* x, int x_ld, const float* y, int y_ld, int npix, int C,
                                                         const float* a, int a_ld, const float* wts, int nw, float lambda,
                                                         double* part, float* g, int g_ld) {
    const double inv = 1.0 / ((double)npix * (double)C);
    double acc = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        float w = 1.f;
        if (a) {
            if (nw == 0) w = a[(int64_t)p * a_ld];   // explicit per-pixel weight map
            for (int i = 0; i < nw; ++i) w += (a[(int64_t)p * a_ld + i] + 1.f) * 0.5f * (wts[i] - 1.f);
        }
        for (int c = 0; c < g_ld; ++c) {
            float gv = 0.f;
            if (c < C) {
                const float d = x[(int64_t)p * x_ld + c] - y[(int64_t)p * y_ld + c];
                acc += (double)(fabsf(d) * w);
                gv = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w * lambda * (float)inv;
            }
            g[(int64_t)p * g_ld + c] = gv;
        }
    }
    sg_block_partial(acc, part);
}

__global__ __launch_bounds__(256) void sg_scale_kernel(const float* gout, const float* g, float* dx, int64_t n4) {
    const float s = gout[0];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(g)[e];
        v *= s;
        reinterpret_cast<f32x4*>(dx)[e] = v;
    }
}

// BCELoss((x + 1) / 2, (t + 1) / 2), mean over npix * C (torch's -100 log clamp); g = dloss/dx for a unit upstream
// gradient (torch: dL/dp = (p - t) / max(p (1 - p), 1e-12), then dp/dx = 1/2)
__global__ __launch_bounds__(256) void sg_bce01_fwd_kernel(const float* x, int x_ld, const float* t, int t_ld, int npix, int C,
                                                           double* part, float* g, int g_ld) {
    const double inv = 1.0 / ((double)npix * (double)C);
    double acc = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        for (int c = 0; c < g_ld; ++c) {
            float gv = 0.f;
            if (c < C) {
                const float pr = (x[(int64_t)p * x_ld + c] + 1.f) * 0.5f, tg = (t[(int64_t)p * t_ld + c] + 1.f) * 0.5f;
                const float lp = fmaxf(logf(pr), -100.f), lq = fmaxf(log1pf(-pr), -100.f);
                acc += (double)(-(tg * lp + (1.f - tg) * lq));
                gv = (pr - tg) / fmaxf(pr * (1.f - pr), 1e-12f) * 0.5f * (float)inv;
            }
            g[(int64_t)p * g_ld + c] = gv;
        }
    }
    sg_block_partial(acc, part);
}

static int sg_imgloss_blocks(int npix) {
    int b = ew_cdiv(npix, 256);
    return b > SG_IMGLOSS_BLOCKS ? SG_IMGLOSS_BLOCKS : b;
}

extern "C" int sgan_bce01_fwd(const float* x, int32_t x_ld, const float* t, int32_t t_ld, int32_t npix, int32_t C, float* loss_out,
                              float* g, int32_t g_ld, void* workspace, int64_t workspace_bytes, void* stream) {
    SGAN_CHECK(x && t && loss_out && g && npix > 0 && C > 0 && g_ld >= C && x_ld >= C && t_ld >= C, "bad argument");
    SGAN_CHECK(workspace && workspace_bytes >= SGAN_IMAGE_LOSS_WS_BYTES && ((uintptr_t)workspace & 7) == 0,
               "workspace of SGAN_IMAGE_LOSS_WS_BYTES (8-byte aligned) required");
    static_assert(SG_IMGLOSS_BLOCKS * sizeof(double) <= SGAN_IMAGE_LOSS_WS_BYTES, "workspace size");
    const int blocks = sg_imgloss_blocks(npix);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(sg_bce01_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, t, t_ld, npix, C, part, g, g_ld);
    SGAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sg_imgloss_fin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, blocks, 1.0 / ((double)npix * (double)C), loss_out);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_l1w_fwd(const float* x, int32_t x_ld, const float* y, int32_t y_ld, int32_t npix, int32_t C, const float* a,
                            int32_t a_ld, const float* weights_dev, int32_t nweights, float lambda, float* loss_out, float* g,
                            int32_t g_ld, void* workspace, int64_t workspace_bytes, void* stream) {
    SGAN_CHECK(x && y && loss_out && g && npix > 0 && C > 0 && g_ld >= C && x_ld >= C && y_ld >= C, "bad argument");
    SGAN_CHECK(!a || nweights == 0 || (weights_dev && nweights > 0 && a_ld >= nweights), "weights need the label image");
    SGAN_CHECK(workspace && workspace_bytes >= SGAN_IMAGE_LOSS_WS_BYTES && ((uintptr_t)workspace & 7) == 0,
               "workspace of SGAN_IMAGE_LOSS_WS_BYTES (8-byte aligned) required");
    const int blocks = sg_imgloss_blocks(npix);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(sg_l1w_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_ld, y, y_ld, npix, C, a, a_ld,
                       weights_dev, nweights, lambda, part, g, g_ld);
    SGAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sg_imgloss_fin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, blocks,
                       (double)lambda / ((double)npix * (double)C), loss_out);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_scale(const float* gout, const float* g, float* dx, int64_t n, void* stream) {
    SGAN_CHECK(gout && g && dx && n > 0 && (n & 3) == 0, "bad argument");
    int blocks = ew_cdiv(n / 4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sg_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gout, g, dx, n / 4);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Class-weighted cross-entropy on logits and softmax over the channel dimension of an NHWC map (round 3; the (f)-row trainers ran
// these on aten kernels): GANLossMultiClass (models/networks.py:188-202: CrossEntropyLoss of every pixel's class scores against one
// class), CrossEntropyLoss2d of the segmentation trainers (models/loss.py:6-12: NLLLoss2d(log_softmax), class weights), and the
// softmax that turns the U-Net's logits into the "fake" label map (models/segm_model.py:155-160).  C <= 16 classes per pixel.
//   loss = sum_p w[y_p] * (lse_p - z_p[y_p]) / sum_p w[y_p]        (torch's weighted mean)
// forward: per-workgroup partials (fp64) of the two sums, the last workgroup (ticket) finishes; backward: recomputes the softmax and
// writes d loss / d z = gout * w[y_p] * (softmax_p - onehot(y_p)) / sum w -- the denominator is read from the forward's sums.
// ------------------------------------------------------------------------------------------
#define SG_CE_MAXC 16
__device__ __forceinline__ float sg_ce_lse(const float* z, int C, float* zs) {
    float m = -3.4e38f;
    for (int c = 0; c < C; ++c) { zs[c] = z[c]; m = fmaxf(m, zs[c]); }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(zs[c] - m);
    return m + logf(sum);
}

__global__ __launch_bounds__(256) void sg_ce_fwd_kernel(const float* logits, int ld, int npix, int C, const int64_t* label, int const_label,
                                                        const float* class_w, double* acc, unsigned* ticket, float* loss_out) {
    double s_num = 0.0, s_den = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        float zs[SG_CE_MAXC];
        const float lse = sg_ce_lse(logits + (int64_t)p * ld, C, zs);
        int y = label ? (int)label[p] : const_label;
        if (y < 0 || y >= C) continue;      // torch's ignore_index (-100) and anything out of range: no contribution
        const float w = class_w ? class_w[y] : 1.f;
        float zy = 0.f;
        for (int c = 0; c < C; ++c) zy = c == y ? zs[c] : zy;
        s_num += (double)(w * (lse - zy));
        s_den += (double)w;
    }
    __shared__ double red[2][4];
    for (int o = 32; o > 0; o >>= 1) { s_num += __shfl_xor(s_num, o); s_den += __shfl_xor(s_den, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_num; red[1][threadIdx.x >> 6] = s_den; }
    SG_SYNC();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&acc[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {      // every workgroup's sums are in: finish
            const double num = __hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double den = __hip_atomic_load(&acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            loss_out[0] = den > 0.0 ? (float)(num / den) : 0.f;
            ticket[0] = 0u;
        }
    }
}

__global__ __launch_bounds__(256) void sg_ce_bwd_kernel(const float* logits, int ld, int npix, int C, const int64_t* label, int const_label,
                                                        const float* class_w, const double* acc, const float* gout, float* dlogits, int dld) {
    const double den = acc[1];
    const float scale = den > 0.0 ? (float)((double)gout[0] / den) : 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        float zs[SG_CE_MAXC];
        const float lse = sg_ce_lse(logits + (int64_t)p * ld, C, zs);
        const int y = label ? (int)label[p] : const_label;
        const bool ok = y >= 0 && y < C;
        const float w = ok ? (class_w ? class_w[y] : 1.f) * scale : 0.f;
        for (int c = 0; c < dld; ++c) dlogits[(int64_t)p * dld + c] = c < C ? w * (expf(zs[c] - lse) - (c == y ? 1.f : 0.f)) : 0.f;
    }
}

extern "C" int sgan_ce_fwd(const float* logits, int32_t ld, int32_t npix, int32_t C, const int64_t* label, int32_t const_label,
                           const float* class_w, double* acc, uint32_t* ticket, float* loss_out, void* stream) {
    SGAN_CHECK(logits && acc && ticket && loss_out && npix > 0 && C >= 1 && C <= SG_CE_MAXC && ld >= C, "bad argument (1..%d classes)", SG_CE_MAXC);
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(sg_ce_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, npix, C, label, const_label, class_w, acc,
                       ticket, loss_out);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_ce_bwd(const float* logits, int32_t ld, int32_t npix, int32_t C, const int64_t* label, int32_t const_label,
                           const float* class_w, const double* acc, const float* gout, float* dlogits, int32_t dld, void* stream) {
    SGAN_CHECK(logits && acc && gout && dlogits && npix > 0 && C >= 1 && C <= SG_CE_MAXC && ld >= C && dld >= C, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sg_ce_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, npix, C, label, const_label, class_w, acc,
                       gout, dlogits, dld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// softmax over the C logical channels of every pixel (padding channels of the result are written as zeros); backward:
// dz = p * (dp - sum_c dp_c p_c)
__global__ __launch_bounds__(256) void sg_softmax_fwd_kernel(const float* z, int ld, int npix, int C, float* p_out, int pld) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        float zs[SG_CE_MAXC];
        const float lse = sg_ce_lse(z + (int64_t)p * ld, C, zs);
        for (int c = 0; c < pld; ++c) p_out[(int64_t)p * pld + c] = c < C ? expf(zs[c] - lse) : 0.f;
    }
}
__global__ __launch_bounds__(256) void sg_softmax_bwd_kernel(const float* dp, int dpld, const float* pr, int pld, int npix, int C, float* dz, int dzld) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += dp[(int64_t)p * dpld + c] * pr[(int64_t)p * pld + c];
        for (int c = 0; c < dzld; ++c) dz[(int64_t)p * dzld + c] = c < C ? pr[(int64_t)p * pld + c] * (dp[(int64_t)p * dpld + c] - dot) : 0.f;
    }
}
extern "C" int sgan_softmax_fwd(const float* z, int32_t ld, int32_t npix, int32_t C, float* p, int32_t pld, void* stream) {
    SGAN_CHECK(z && p && npix > 0 && C >= 1 && C <= SG_CE_MAXC && ld >= C && pld >= C, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sg_softmax_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, ld, npix, C, p, pld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}
extern "C" int sgan_softmax_bwd(const float* dp, int32_t dpld, const float* p, int32_t pld, int32_t npix, int32_t C, float* dz, int32_t dzld,
                                void* stream) {
    SGAN_CHECK(dp && p && dz && npix > 0 && C >= 1 && C <= SG_CE_MAXC && dpld >= C && pld >= C && dzld >= C, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sg_softmax_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dp, dpld, p, pld, npix, C, dz, dzld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// transposed master copy of the conv weights (backward-data wants the reduction channel contiguous)
// ------------------------------------------------------------------------------------------
struct SgWtTable { sgan_wt_seg s[64]; int64_t first[65]; int32_t n; };   // first[i]: first 32x32 tile of segment i

__global__ __launch_bounds__(256) void sg_transpose_weights_kernel(const float* flat, float* flat_t, const SgWtTable T) {
    __shared__ float tile[32][33];
    int i = 0;
    for (int k = 1; k < T.n; ++k)
        if ((int64_t)blockIdx.x >= T.first[k]) i = k;
    const sgan_wt_seg S = T.s[i];
    int64_t t = blockIdx.x - T.first[i];
    const int tc = (S.cin + 31) / 32, tr = (S.cout + 31) / 32;
    const int bx = (int)(t % tc); t /= tc;
    const int by = (int)(t % tr); t /= tr;      // t = tap
    const float* src = flat + S.off + t * (int64_t)S.cout * S.cin;
    float* dst = flat_t + S.off + t * (int64_t)S.cout * S.cin;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int co = by * 32 + r, ci = bx * 32 + lx;
        tile[r][lx] = (co < S.cout && ci < S.cin) ? src[(int64_t)co * S.cin + ci] : 0.f;
    }
    SG_SYNC();
    for (int r = ly; r < 32; r += 8) {
        const int ci = bx * 32 + r, co = by * 32 + lx;
        if (ci < S.cin && co < S.cout) dst[(int64_t)ci * S.cout + co] = tile[lx][r];
    }
}

extern "C" int sgan_transpose_weights(const float* flat, float* flat_t, const sgan_wt_seg* segs, int32_t n, void* stream) {
    SGAN_CHECK(flat && flat_t && segs && n >= 1 && n <= 64, "1..64 segments");
    SgWtTable T;
    T.n = n;
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        SGAN_CHECK(segs[i].taps > 0 && segs[i].cout > 0 && segs[i].cin > 0, "bad segment %d", i);
        T.s[i] = segs[i];
        T.first[i] = tiles;
        tiles += (int64_t)segs[i].taps * ((segs[i].cout + 31) / 32) * ((segs[i].cin + 31) / 32);
    }
    T.first[n] = tiles;
    hipLaunchKernelGGL(sg_transpose_weights_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, flat, flat_t, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Every derived weight copy in one launch (sgan_pack_weights): the fp32 transposed copy, the split-fp16 forward copy and the
// split-bf16 backward copy.
// One workgroup per 32 x 32 (co, ci) tile of a tap: the tile is read once into LDS; thread (row r = tid >> 3, group
// q = (tid >> 1) & 3, plane p = tid & 1) then writes one 16-byte chunk {8 x bf16} of the forward copy (row = co, the 8
// consecutive ci of group q) and one of the backward copy (row = ci, 8 consecutive co).
// ------------------------------------------------------------------------------------------
typedef unsigned sg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned sg_bf16_rne(float x) {   // bits of bf16(x), round to nearest even (NaN stays NaN)
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x);
}

// plane 0: hi = bf16(x); plane 1: lo = bf16(x - hi)
__device__ __forceinline__ sg_u32x4 sg_split8(const float* v, int plane) {
    unsigned h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned hi = sg_bf16_rne(v[e]);
        h[e] = plane == 0 ? hi : sg_bf16_rne(v[e] - __builtin_bit_cast(float, hi << 16));
    }
    return (sg_u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
}

// fp16 planes of x * 2^SGAN_F16_WEIGHT_SHIFT: hi = fp16(x s), lo = fp16(x s - hi)  (round to nearest even)
__device__ __forceinline__ sg_u32x4 sg_split8_f16(const float* v, int plane) {
    unsigned h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * (float)(1 << SGAN_F16_WEIGHT_SHIFT);
        const _Float16 hi = (_Float16)x;
        const _Float16 r = plane == 0 ? hi : (_Float16)(x - (float)hi);
        h[e] = (unsigned)__builtin_bit_cast(unsigned short, r);
    }
    return (sg_u32x4){h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
}

__global__ __launch_bounds__(256) void sg_pack_weights_kernel(const float* flat, float* flat_t, float* pk_f, float* pk_b, float* pk_bh, const SgWtTable T) {
    __shared__ float tile[32][33];
    int i = 0;
    for (int k = 1; k < T.n; ++k)
        if ((int64_t)blockIdx.x >= T.first[k]) i = k;
    const sgan_wt_seg S = T.s[i];
    int64_t t = blockIdx.x - T.first[i];
    const int tc = (S.cin + 31) / 32, tr = (S.cout + 31) / 32;
    const int bx = (int)(t % tc); t /= tc;
    const int by = (int)(t % tr); t /= tr;      // t = tap
    const int64_t slab = S.off + t * (int64_t)S.cout * S.cin;
    const float* src = flat + slab;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int co = by * 32 + r, ci = bx * 32 + lx;
        tile[r][lx] = (co < S.cout && ci < S.cin) ? src[(int64_t)co * S.cin + ci] : 0.f;
    }
    SG_SYNC();
    if (flat_t) {
        float* dst = flat_t + slab;
        for (int r = ly; r < 32; r += 8) {
            const int ci = bx * 32 + r, co = by * 32 + lx;
            if (ci < S.cin && co < S.cout) dst[(int64_t)ci * S.cout + co] = tile[lx][r];
        }
    }
    const int r = threadIdx.x >> 3, q = (threadIdx.x >> 1) & 3, p = threadIdx.x & 1;
    if (pk_f && (S.cin & 7) == 0) {
        const int co = by * 32 + r, ci = bx * 32 + q * 8;
        if (co < S.cout && ci < S.cin) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[r][q * 8 + e];
            *reinterpret_cast<sg_u32x4*>(pk_f + slab + (int64_t)co * S.cin + ci + 4 * p) = sg_split8_f16(v, p);
        }
    }
    if (pk_b && (S.cout & 7) == 0) {
        const int ci = bx * 32 + r, co = by * 32 + q * 8;
        if (ci < S.cin && co < S.cout) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[q * 8 + e][r];
            *reinterpret_cast<sg_u32x4*>(pk_b + slab + (int64_t)ci * S.cout + co + 4 * p) = sg_split8(v, p);
            if (pk_bh) *reinterpret_cast<sg_u32x4*>(pk_bh + slab + (int64_t)ci * S.cout + co + 4 * p) = sg_split8_f16(v, p);
        }
    }
}

extern "C" int sgan_pack_weights(const float* flat, float* flat_t, void* packed_fwd, void* packed_bwd, void* packed_bwd_f16,
                                 const sgan_wt_seg* segs, int32_t n, void* stream) {
    SGAN_CHECK(flat && segs && n >= 1 && n <= 64, "1..64 segments");
    SGAN_CHECK(flat_t || packed_fwd || packed_bwd, "no destination");
    SgWtTable T;
    T.n = n;
    int64_t tiles = 0;
    for (int i = 0; i < n; ++i) {
        SGAN_CHECK(segs[i].taps > 0 && segs[i].cout > 0 && segs[i].cin > 0 && (segs[i].off & 3) == 0, "bad segment %d", i);
        T.s[i] = segs[i];
        T.first[i] = tiles;
        tiles += (int64_t)segs[i].taps * ((segs[i].cout + 31) / 32) * ((segs[i].cin + 31) / 32);
    }
    T.first[n] = tiles;
    SGAN_CHECK(!packed_bwd_f16 || packed_bwd, "packed_bwd_f16 rides on the packed_bwd pass");
    hipLaunchKernelGGL(sg_pack_weights_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, flat, flat_t,
                       (float*)packed_fwd, (float*)packed_bwd, (float*)packed_bwd_f16, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// BatchNorm running statistics for up to 16 layers in one launch
// ------------------------------------------------------------------------------------------
struct SgBnRunTable {
    sgan_bn_running_desc l[16];
    int n;
    float momentum;
};

__global__ __launch_bounds__(256) void sg_bn_running_kernel(SgBnRunTable T) {
    sg_warm_kernargs<(int)sizeof(SgBnRunTable)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    const sgan_bn_running_desc& L = T.l[blockIdx.x];
    const double inv = 1.0 / (double)L.count;
    const double unb = L.count > 1 ? (double)L.count / (double)(L.count - 1) : 1.0;
    if (threadIdx.x == 0 && L.num_batches_tracked) L.num_batches_tracked[0] += 1;
    for (int c = threadIdx.x; c < L.C; c += 256) {
        const double m = sg_stat_sum(L.stats, c, L.rep_stride) * inv;
        double var = sg_stat_sum(L.stats, (L.sq_stride ? L.sq_stride : L.C) + c, L.rep_stride) * inv - m * m;
        if (var < 0.0) var = 0.0;
        L.running_mean[c] = (1.f - T.momentum) * L.running_mean[c] + T.momentum * (float)m;
        L.running_var[c] = (1.f - T.momentum) * L.running_var[c] + T.momentum * (float)(var * unb);
    }
}

extern "C" int sgan_bn_running_update(const sgan_bn_running_desc* layers, int32_t n, float momentum, void* stream) {
    SGAN_CHECK(layers && n > 0 && n <= 16, "1..16 layers per launch");
    SgBnRunTable T;
    for (int i = 0; i < n; ++i) T.l[i] = layers[i];
    T.n = n;
    T.momentum = momentum;
    hipLaunchKernelGGL(sg_bn_running_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Input pipeline tail: crop -> horizontal flip -> rotate by 90 deg * rot -> ToTensor -> Normalize(0.5, 0.5), one thread per
// output pixel: a 4-byte store per selected channel, the decoded RGB bytes gathered from the crop window.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_image_prep_kernel(const unsigned char* img, int H0, int W0, int x0, int y0, int n, int flip,
                                                            int rot, float* dst, int dst_ld, int Cstore) {
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % n), i = (int)(e / n);
        // PIL's transpose(ROTATE_90 * rot) is counter-clockwise: R[i][j] = F[j][n-1-i] (rot 1), F[n-1-i][n-1-j] (2), F[n-1-j][i] (3)
        int a = i, b = j;
        if (rot == 1) { a = j; b = n - 1 - i; }
        else if (rot == 2) { a = n - 1 - i; b = n - 1 - j; }
        else if (rot == 3) { a = n - 1 - j; b = i; }
        if (flip) b = n - 1 - b;
        const unsigned char* px = img + ((int64_t)(y0 + a) * W0 + (x0 + b)) * 3;
        float* o = dst + e * dst_ld;
        for (int c = 0; c < Cstore; ++c) {
            float v = 0.f;
            if (c < 3) {
                v = (float)px[c] / 255.f;         // ToTensor
                v = (v - 0.5f) / 0.5f;            // Normalize((.5,.5,.5), (.5,.5,.5)), same operation order: bit-exact
            }
            o[c] = v;
        }
    }
}

extern "C" int sgan_image_prep(const unsigned char* img, int32_t H0, int32_t W0, int32_t x0, int32_t y0, int32_t n, int32_t flip,
                               int32_t rot, float* dst, int32_t dst_ld, int32_t Cstore, void* stream) {
    SGAN_CHECK(img && dst && H0 > 0 && W0 > 0 && n > 0, "bad argument");
    SGAN_CHECK(x0 >= 0 && y0 >= 0 && x0 + n <= W0 && y0 + n <= H0, "crop window %d+%d x %d+%d outside the %d x %d image", x0, n, y0, n, W0, H0);
    SGAN_CHECK(rot >= 0 && rot <= 3 && Cstore >= 3 && dst_ld >= Cstore, "bad rot / channel count");
    int blocks = ew_cdiv((int64_t)n * n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_image_prep_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, H0, W0, x0, y0, n, flip != 0, rot, dst,
                       dst_ld, Cstore);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Image.resize(size, BILINEAR | BICUBIC) of an 8-bit interleaved image, bit-exact with Pillow's two-pass resampler
// (src/libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc + ImagingResampleHorizontal/Vertical_8bpc; called from
// data/base_dataset.py:19-21,43-50 and data/aligned_dataset.py:25).  Per axis: the filter stretched by the down-scale factor, taps
// normalised in double and rounded to 22-bit fixed point (one thread per output coordinate; fp contraction off so every double
// operation rounds as the C source's does), then an integer gather-MAC per output byte, rounded and clipped to 8 bits after each pass.
// ------------------------------------------------------------------------------------------
#define SG_RESAMPLE_BITS (32 - 8 - 2)

__device__ inline double sg_resample_filter(int bicubic, double x) {
#pragma clang fp contract(off)
    if (x < 0.0) x = -x;
    if (!bicubic) return x < 1.0 ? 1.0 - x : 0.0;
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

__global__ __launch_bounds__(256) void sg_resample_coeffs_kernel(int in_size, int out_size, int bicubic, int ksize, int* bounds, int* kk) {
#pragma clang fp contract(off)
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = (bicubic ? 2.0 : 1.0) * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += sg_resample_filter(bicubic, (x + xmin - center + 0.5) * ss);
    int* k = kk + (int64_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = sg_resample_filter(bicubic, (x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
        }
        k[x] = w < 0 ? (int)(-0.5 + w * (1 << SG_RESAMPLE_BITS)) : (int)(0.5 + w * (1 << SG_RESAMPLE_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

// one thread per output byte (consecutive threads -> consecutive bytes of a row)
template <bool HORIZ>
__global__ __launch_bounds__(256) void sg_resample_pass_kernel(const unsigned char* src, int Ws, int C, unsigned char* dst, int Hd, int Wd,
                                                               const int* bounds, const int* kk, int ksize) {
    const int64_t total = (int64_t)Hd * Wd * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const int64_t pix = e / C;
        const int x = (int)(pix % Wd), y = (int)(pix / Wd);
        const int xx = HORIZ ? x : y;
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int* k = kk + (int64_t)xx * ksize;
        const unsigned char* p = HORIZ ? src + ((int64_t)y * Ws + x0) * C + c : src + ((int64_t)x0 * Ws + x) * C + c;
        const int64_t step = HORIZ ? C : (int64_t)Ws * C;
        int acc = 1 << (SG_RESAMPLE_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (int)p[t * step] * k[t];
        acc >>= SG_RESAMPLE_BITS;
        dst[e] = (unsigned char)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
    }
}

static int sg_resample_ksize(int in_size, int out_size, int filter) {
    double scale = (double)in_size / out_size;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil((filter == SGAN_RESAMPLE_BICUBIC ? 2.0 : 1.0) * scale) * 2 + 1;
}

static inline int64_t sg_up16(int64_t v) { return (v + 15) & ~(int64_t)15; }

extern "C" int64_t sgan_image_resize_workspace(int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, int32_t filter) {
    if (H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0) return 0;
    return sg_up16((int64_t)H * Wo * C) + sg_up16((int64_t)Wo * (2 + sg_resample_ksize(W, Wo, filter)) * 4) +
           sg_up16((int64_t)Ho * (2 + sg_resample_ksize(H, Ho, filter)) * 4);
}

extern "C" int sgan_image_resize(const unsigned char* src, int32_t H, int32_t W, int32_t C, unsigned char* dst, int32_t Ho, int32_t Wo,
                                 int32_t filter, void* ws, int64_t ws_bytes, void* stream) {
    SGAN_CHECK(src && dst && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C <= 4, "bad argument");
    SGAN_CHECK(filter == SGAN_RESAMPLE_BILINEAR || filter == SGAN_RESAMPLE_BICUBIC, "resample filter %d not supported (2 bilinear, 3 bicubic)", filter);
    SGAN_CHECK(H < (1 << 24) && W < (1 << 24) && Ho < (1 << 24) && Wo < (1 << 24), "image too large");
    hipStream_t st = (hipStream_t)stream;
    if (Ho == H && Wo == W) {      // Image.resize returns a copy
        hipError_t e = hipMemcpyAsync(dst, src, (size_t)H * W * C, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return sgan_fail(SGAN_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
        return SGAN_OK;
    }
    SGAN_CHECK(ws && ws_bytes >= sgan_image_resize_workspace(H, W, C, Ho, Wo, filter), "workspace too small (sgan_image_resize_workspace)");
    const int bic = filter == SGAN_RESAMPLE_BICUBIC;
    unsigned char* tmp = (unsigned char*)ws;
    int* bh = (int*)(tmp + sg_up16((int64_t)H * Wo * C));
    const int ksh = sg_resample_ksize(W, Wo, filter), ksv = sg_resample_ksize(H, Ho, filter);
    int* kh = bh + 2 * Wo;
    int* bv = (int*)((char*)bh + sg_up16((int64_t)Wo * (2 + ksh) * 4));
    int* kv = bv + 2 * Ho;
    const unsigned char* cur = src;
    if (Wo != W) {                 // horizontal pass first (ImagingResample)
        unsigned char* out = Ho != H ? tmp : dst;
        hipLaunchKernelGGL(sg_resample_coeffs_kernel, dim3(ew_cdiv(Wo, 256)), dim3(256), 0, st, W, Wo, bic, ksh, bh, kh);
        int blocks = ew_cdiv((int64_t)H * Wo * C, 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((sg_resample_pass_kernel<true>), dim3(blocks), dim3(256), 0, st, cur, W, C, out, H, Wo, bh, kh, ksh);
        cur = out;
    }
    if (Ho != H) {
        hipLaunchKernelGGL(sg_resample_coeffs_kernel, dim3(ew_cdiv(Ho, 256)), dim3(256), 0, st, H, Ho, bic, ksv, bv, kv);
        int blocks = ew_cdiv((int64_t)Ho * Wo * C, 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((sg_resample_pass_kernel<false>), dim3(blocks), dim3(256), 0, st, cur, Wo, C, dst, Ho, Wo, bv, kv, ksv);
    }
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Gaussian pre-filter: depthwise (diagonal of the dense reference weight), strided outputs only
// ------------------------------------------------------------------------------------------
// one thread per (pixel, channel quad): 16-byte loads, the k x k taps of the channel's Gaussian from LDS
__global__ __launch_bounds__(256) void sg_gauss_fwd_kernel(const float* in, int in_ld, int H, int W, int C, int Creal,
                                                           const float* g, int gcs, int k, int pad, int s, float* out,
                                                           int out_ld, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gs = reinterpret_cast<float*>(smem);   // [k*k][C] (zero for padding channels)
    for (int i = threadIdx.x; i < k * k * C; i += 256) {
        const int c = i % C, t = i / C;
        gs[i] = c < Creal ? g[(int64_t)c * gcs + t] : 0.f;
    }
    SG_SYNC();
    const int CQ = C >> 2;
    const int total = Ho * Wo * CQ;      // < 2^31 (host-checked): 32-bit index arithmetic (the 64-bit divisions were most of this kernel's time)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int pix = e / CQ, c = (e - pix * CQ) * 4;
        const int oy = pix / Wo, ox = pix - oy * Wo;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ky0 = max(0, pad - oy * s), ky1 = min(k, H + pad - oy * s);
        const int kx0 = max(0, pad - ox * s), kx1 = min(k, W + pad - ox * s);
        for (int ky = ky0; ky < ky1; ++ky) {
            const float* row = in + ((int64_t)(oy * s + ky - pad) * W + (ox * s - pad)) * in_ld + c;
            for (int kx = kx0; kx < kx1; ++kx)
                acc += *reinterpret_cast<const f32x4*>(gs + (ky * k + kx) * C + c) * *reinterpret_cast<const f32x4*>(row + (int64_t)kx * in_ld);
        }
        *reinterpret_cast<f32x4*>(out + (int64_t)pix * out_ld + c) = acc;
    }
}

// transpose of the above: only the taps with (iy + pad - ky) % s == 0 reach an output row (ceil(k/s) per axis)
__global__ __launch_bounds__(256) void sg_gauss_bwd_kernel(const float* dout, int dout_ld, int Ho, int Wo, int C, int Creal,
                                                           const float* g, int gcs, int k, int pad, int s, float* din,
                                                           int din_ld, int H, int W, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gs = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < k * k * C; i += 256) {
        const int c = i % C, t = i / C;
        gs[i] = c < Creal ? g[(int64_t)c * gcs + t] : 0.f;
    }
    SG_SYNC();
    const int CQ = C >> 2;
    const int64_t total = (int64_t)H * W * CQ;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % CQ) * 4;
        const int64_t pix = e / CQ;
        const int ix = (int)(pix % W), iy = (int)(pix / W);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ky = (iy + pad) % s; ky < k; ky += s) {
            const int oy = (iy + pad - ky) / s;          // exact; decreasing in ky
            if (iy + pad - ky < 0) break;
            if (oy >= Ho) continue;
            for (int kx = (ix + pad) % s; kx < k; kx += s) {
                const int ox = (ix + pad - kx) / s;
                if (ix + pad - kx < 0) break;
                if (ox >= Wo) continue;
                acc += *reinterpret_cast<const f32x4*>(gs + (ky * k + kx) * C + c) *
                       *reinterpret_cast<const f32x4*>(dout + ((int64_t)oy * Wo + ox) * dout_ld + c);
            }
        }
        f32x4* o = reinterpret_cast<f32x4*>(din + pix * din_ld + c);
        *o = accumulate ? *o + acc : acc;
    }
}

// ---- several pre-filters in one launch: the multi-scale discriminators filter the same image at scale 2 and 4 --------
struct SgGaussJob {
    const float* src; float* dst; const float* g;
    int32_t src_ld, dst_ld, Hs, Ws, Hd, Wd, gcs, k, pad, s;   // src/dst: fwd = (image, downsampled), bwd = (d downsampled, d image)
};
struct SgGaussTable { SgGaussJob j[4]; int32_t n, C, Creal, accumulate; };

__global__ __launch_bounds__(256) void sg_gauss_multi_fwd_kernel(const SgGaussTable T) {
    sg_warm_kernargs<(int)sizeof(SgGaussTable)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gs = reinterpret_cast<float*>(smem);   // [k*k][C] (zero for padding channels)
    const SgGaussJob& J = T.j[blockIdx.y];
    const int C = T.C, k = J.k, pad = J.pad, s = J.s, H = J.Hs, W = J.Ws, Ho = J.Hd, Wo = J.Wd;
    for (int i = threadIdx.x; i < k * k * C; i += 256) {
        const int c = i % C, t = i / C;
        gs[i] = c < T.Creal ? J.g[(int64_t)c * J.gcs + t] : 0.f;
    }
    SG_SYNC();
    const int CQ = C >> 2;
    const int total = Ho * Wo * CQ;      // < 2^31 (host-checked): 32-bit index arithmetic (the 64-bit divisions were most of this kernel's time)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int pix = e / CQ, c = (e - pix * CQ) * 4;
        const int oy = pix / Wo, ox = pix - oy * Wo;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int ky0 = max(0, pad - oy * s), ky1 = min(k, H + pad - oy * s);
        const int kx0 = max(0, pad - ox * s), kx1 = min(k, W + pad - ox * s);
        // a row of taps at a time, all its loads in flight before the first is used (round 2 walked the taps one dependent load after
        // the other behind run-time loop bounds: 81 memory round trips per output of the scale-4 filter); taps outside the image or
        // beyond k read a clamped address and meet a zero weight
        for (int ky = ky0; ky < ky1; ++ky) {
            const float* row = J.src + ((int64_t)(oy * s + ky - pad) * W + (ox * s - pad)) * J.src_ld + c;
            const float* grow = gs + ky * k * C + c;
            for (int kb = 0; kb < k; kb += 9) {
                f32x4 v[9], g4[9];
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int kx = kb + u;
                    const bool ok = (kx >= kx0) & (kx < kx1);
                    v[u] = *reinterpret_cast<const f32x4*>(row + (int64_t)(ok ? kx : kx0) * J.src_ld);
                    g4[u] = ok ? *reinterpret_cast<const f32x4*>(grow + kx * C) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 9; ++u) acc += g4[u] * v[u];
            }
        }
        *reinterpret_cast<f32x4*>(J.dst + (int64_t)pix * J.dst_ld + c) = acc;
    }
}

// backward of several pre-filters of ONE image: every thread owns an image (pixel, channel quad) and sums the jobs
__global__ __launch_bounds__(256) void sg_gauss_multi_bwd_kernel(const SgGaussTable T) {
    sg_warm_kernargs<(int)sizeof(SgGaussTable)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gs = reinterpret_cast<float*>(smem);   // job after job: [k*k][C]
    const int C = T.C;
    int goff[4];
    {
        int o = 0;
        for (int j = 0; j < T.n; ++j) {
            goff[j] = o;
            const int kk = T.j[j].k * T.j[j].k;
            for (int i = threadIdx.x; i < kk * C; i += 256) {
                const int c = i % C, t = i / C;
                gs[o + i] = c < T.Creal ? T.j[j].g[(int64_t)c * T.j[j].gcs + t] : 0.f;
            }
            o += kk * C;
        }
    }
    SG_SYNC();
    const int H = T.j[0].Hd, W = T.j[0].Wd, CQ = C >> 2;
    const int total = H * W * CQ;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int pix = e / CQ, c = (e - pix * CQ) * 4;
        const int iy = pix / W, ix = pix - iy * W;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < T.n; ++j) {
            const SgGaussJob& J = T.j[j];
            const int k = J.k, pad = J.pad, s = J.s, Ho = J.Hs, Wo = J.Ws;
            const float* gj = gs + goff[j];
            // at most ceil(k / s)^2 <= 3 x 3 taps reach this image pixel (k = 4 sigma + 1, s = 2 sigma): all nine loads leave before the
            // first is used, invalid ones read a clamped address against a zero weight (host-checked: k <= 3 s)
            const int kyb = (iy + pad) % s, kxb = (ix + pad) % s;
            f32x4 v[9], g4[9];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int ky = kyb + a * s, ny = iy + pad - ky;
                const int oy = ny / s;          // exact when ny >= 0
                const bool yok = (ky < k) & (ny >= 0) & (oy < Ho);
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int kx = kxb + b * s, nx = ix + pad - kx;
                    const int ox = nx / s;
                    const bool ok = yok & (kx < k) & (nx >= 0) & (ox < Wo);
                    v[a * 3 + b] = *reinterpret_cast<const f32x4*>(J.src + (ok ? ((int64_t)oy * Wo + ox) * J.src_ld : 0) + c);
                    g4[a * 3 + b] = ok ? *reinterpret_cast<const f32x4*>(gj + (ky * k + kx) * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < 9; ++u) acc += g4[u] * v[u];
        }
        f32x4* o = reinterpret_cast<f32x4*>(T.j[0].dst + (int64_t)pix * T.j[0].dst_ld + c);
        *o = T.accumulate ? *o + acc : acc;
    }
}

static int sg_fill_gauss(SgGaussTable& T, const sgan_gauss_job* jobs, int n, int C, int Creal, bool bwd) {
    if (!jobs || n < 1 || n > 4 || (C & 3) || Creal > C || C <= 0) return sgan_fail(SGAN_ERR_INVALID, "1..4 gauss jobs, C % 4 == 0");
    memset(&T, 0, sizeof(T));
    T.n = n; T.C = C; T.Creal = Creal;
    int lds = 0;
    for (int i = 0; i < n; ++i) {
        const sgan_gauss_job& J = jobs[i];
        if (!J.image || !J.down || !J.g || J.k <= 0 || J.s <= 0 || (J.image_ld & 3) || (J.down_ld & 3) || J.image_ld < C || J.down_ld < C)
            return sgan_fail(SGAN_ERR_INVALID, "bad gauss job %d", i);
        if (J.Ho != (J.H + 2 * J.pad - J.k) / J.s + 1 || J.Wo != (J.W + 2 * J.pad - J.k) / J.s + 1)
            return sgan_fail(SGAN_ERR_INVALID, "gauss geometry mismatch in job %d", i);
        SgGaussJob& D = T.j[i];
        D.g = J.g; D.gcs = J.g_chan_stride; D.k = J.k; D.pad = J.pad; D.s = J.s;
        if (!bwd) { D.src = J.image; D.src_ld = J.image_ld; D.Hs = J.H; D.Ws = J.W; D.dst = J.down; D.dst_ld = J.down_ld; D.Hd = J.Ho; D.Wd = J.Wo; }
        else { D.src = J.down; D.src_ld = J.down_ld; D.Hs = J.Ho; D.Ws = J.Wo; D.dst = J.image; D.dst_ld = J.image_ld; D.Hd = J.H; D.Wd = J.W; }
        lds += J.k * J.k * C * 4;
        if (bwd && (J.image != jobs[0].image || J.H != jobs[0].H || J.W != jobs[0].W || J.image_ld != jobs[0].image_ld))
            return sgan_fail(SGAN_ERR_INVALID, "backward jobs must share the image gradient");
    }
    if (lds > 60000) return sgan_fail(SGAN_ERR_UNSUPPORTED, "gauss taps do not fit LDS");
    for (int i = 0; i < n; ++i) {
        if ((int64_t)jobs[i].H * jobs[i].W * (C >> 2) >= (1ll << 31)) return sgan_fail(SGAN_ERR_UNSUPPORTED, "gauss image too large (job %d)", i);
        if (bwd && jobs[i].k > 3 * jobs[i].s) return sgan_fail(SGAN_ERR_UNSUPPORTED, "gauss backward: k <= 3 * stride (job %d)", i);
    }
    return SGAN_OK;
}

extern "C" int sgan_gauss_down_multi_fwd(const sgan_gauss_job* jobs, int32_t n, int32_t C, int32_t Creal, void* stream) {
    SgGaussTable T;
    int rc = sg_fill_gauss(T, jobs, n, C, Creal, false);
    if (rc) return rc;
    int64_t maxt = 0;
    int maxk = 0;
    for (int i = 0; i < n; ++i) { maxt = max(maxt, (int64_t)jobs[i].Ho * jobs[i].Wo * (C >> 2)); maxk = max(maxk, jobs[i].k); }
    int blocks = ew_cdiv(maxt, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sg_gauss_multi_fwd_kernel, dim3(blocks, n), dim3(256), (size_t)maxk * maxk * C * 4, (hipStream_t)stream, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_gauss_down_multi_bwd(const sgan_gauss_job* jobs, int32_t n, int32_t C, int32_t Creal, int32_t accumulate, void* stream) {
    SgGaussTable T;
    int rc = sg_fill_gauss(T, jobs, n, C, Creal, true);
    if (rc) return rc;
    T.accumulate = accumulate;
    size_t lds = 0;
    for (int i = 0; i < n; ++i) lds += (size_t)jobs[i].k * jobs[i].k * C * 4;
    const int64_t total = (int64_t)jobs[0].H * jobs[0].W * (C >> 2);
    int blocks = ew_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_gauss_multi_bwd_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_gauss_down_fwd(const float* in, int32_t in_ld, int32_t H, int32_t W, int32_t C, int32_t Creal,
                                   const float* g, int32_t g_chan_stride, int32_t k, int32_t pad, int32_t s, float* out,
                                   int32_t out_ld, int32_t Ho, int32_t Wo, void* stream) {
    SGAN_CHECK(in && g && out && k > 0 && s > 0 && Creal <= C, "bad argument");
    SGAN_CHECK((C & 3) == 0 && (in_ld & 3) == 0 && (out_ld & 3) == 0 && in_ld >= C && out_ld >= C && k * k * C * 4 <= 60000, "bad channel layout");
    SGAN_CHECK(Ho == (H + 2 * pad - k) / s + 1 && Wo == (W + 2 * pad - k) / s + 1, "gauss geometry mismatch");
    const int64_t total = (int64_t)Ho * Wo * (C >> 2);
    int blocks = ew_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_gauss_fwd_kernel, dim3(blocks), dim3(256), (size_t)k * k * C * 4, (hipStream_t)stream, in, in_ld, H, W, C, Creal, g,
                       g_chan_stride, k, pad, s, out, out_ld, Ho, Wo);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_gauss_down_bwd(const float* dout, int32_t dout_ld, int32_t Ho, int32_t Wo, int32_t C, int32_t Creal,
                                   const float* g, int32_t g_chan_stride, int32_t k, int32_t pad, int32_t s, float* din,
                                   int32_t din_ld, int32_t H, int32_t W, int32_t accumulate, void* stream) {
    SGAN_CHECK(dout && g && din && k > 0 && s > 0 && Creal <= C, "bad argument");
    SGAN_CHECK((C & 3) == 0 && (din_ld & 3) == 0 && (dout_ld & 3) == 0 && din_ld >= C && dout_ld >= C && k * k * C * 4 <= 60000, "bad channel layout");
    SGAN_CHECK(Ho == (H + 2 * pad - k) / s + 1 && Wo == (W + 2 * pad - k) / s + 1, "gauss geometry mismatch");
    const int64_t total = (int64_t)H * W * (C >> 2);
    int blocks = ew_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_gauss_bwd_kernel, dim3(blocks), dim3(256), (size_t)k * k * C * 4, (hipStream_t)stream, dout, dout_ld, Ho, Wo, C, Creal,
                       g, g_chan_stride, k, pad, s, din, din_ld, H, W, accumulate);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// CRN: bilinear x2 upsampling (align_corners = False) with the statistics of its result, its adjoint,
// and the label pyramid AvgPool2d(2^(s+1)), s = 0..5
// ------------------------------------------------------------------------------------------
// out[2i + a] = 0.25 * in[clamp(i - 1 + 2a)] + 0.75 * in[i]   (a = 0: neighbour i-1, a = 1: neighbour i+1), separable.
// Thread t always works on channel quad t % (C/4) (host: 256 % (C/4) == 0), so the per-channel sums stay in registers.
__global__ __launch_bounds__(256) void sg_bilinear_up2_fwd_kernel(const float* in, int in_ld, int H, int W, int C, float* out,
                                                                  int out_ld, double* stats, int stats_sq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);   // [2C], fp64 from the first add (see sg_igemm_kernel)
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.0;
    SG_SYNC();
    const int CQ = C >> 2, Wo = 2 * W;
    const int64_t total = (int64_t)4 * H * W * CQ;
    const int c = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % CQ) * 4;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / CQ;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        const int iy = oy >> 1, ix = ox >> 1;
        const int ny = min(max(iy - 1 + 2 * (oy & 1), 0), H - 1), nx = min(max(ix - 1 + 2 * (ox & 1), 0), W - 1);
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + ((int64_t)iy * W + ix) * in_ld + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(in + ((int64_t)iy * W + nx) * in_ld + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(in + ((int64_t)ny * W + ix) * in_ld + c);
        const f32x4 f = *reinterpret_cast<const f32x4*>(in + ((int64_t)ny * W + nx) * in_ld + c);
        const f32x4 v = 0.75f * (0.75f * a + 0.25f * b) + 0.25f * (0.75f * d + 0.25f * f);
        *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s1[j] += (double)v[j];
            s2[j] += (double)v[j] * (double)v[j];
        }
    }
    if (stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(&red[c + j], s1[j]);
            atomicAdd(&red[C + c + j], s2[j]);
        }
        SG_SYNC();
        for (int i = threadIdx.x; i < C; i += 256) {
            atomicAdd(&stats[i], red[i]);
            atomicAdd(&stats[stats_sq + i], red[C + i]);
        }
    }
}

// din[i] = sum over output rows {2i-1, 2i, 2i+1, 2i+2} (clamped into the image) with weights {.25, .75, .75, .25}, separable
__global__ __launch_bounds__(256) void sg_bilinear_up2_bwd_kernel(const float* dout, int dout_ld, int H, int W, int C, float* din,
                                                                  int din_ld) {
    const int CQ = C >> 2, Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)H * W * CQ;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % CQ) * 4;
        const int64_t pix = e / CQ;
        const int ix = (int)(pix % W), iy = (int)(pix / W);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = min(max(2 * iy - 1 + a, 0), Ho - 1);
            const float wy = (a == 0 || a == 3) ? 0.25f : 0.75f;
            f32x4 row = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = min(max(2 * ix - 1 + b, 0), Wo - 1);
                const float wx = (b == 0 || b == 3) ? 0.25f : 0.75f;
                row += wx * *reinterpret_cast<const f32x4*>(dout + ((int64_t)oy * Wo + ox) * dout_ld + c);
            }
            acc += wy * row;
        }
        *reinterpret_cast<f32x4*>(din + pix * din_ld + c) = acc;
    }
}

extern "C" int sgan_bilinear_up2_fwd(const float* in, int32_t in_ld, int32_t H, int32_t W, int32_t C, float* out, int32_t out_ld,
                                     double* out_stats, int32_t out_stats_sq_stride, void* stream) {
    SGAN_CHECK(in && out && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && in_ld >= C && out_ld >= C && (in_ld & 3) == 0 &&
                   (out_ld & 3) == 0, "bad argument");
    SGAN_CHECK(256 % (C >> 2) == 0, "channel count must be 4 * a divisor of 256");
    const int64_t total = (int64_t)4 * H * W * (C >> 2);
    int blocks = ew_cdiv(total, 256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sg_bilinear_up2_fwd_kernel, dim3(blocks), dim3(256), (size_t)2 * C * 8, (hipStream_t)stream, in, in_ld, H, W,
                       C, out, out_ld, out_stats, out_stats_sq_stride ? out_stats_sq_stride : C);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_bilinear_up2_bwd(const float* dout, int32_t dout_ld, int32_t H, int32_t W, int32_t C, float* din,
                                     int32_t din_ld, void* stream) {
    SGAN_CHECK(dout && din && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && din_ld >= C && dout_ld >= C && (din_ld & 3) == 0 &&
                   (dout_ld & 3) == 0, "bad argument");
    const int64_t total = (int64_t)H * W * (C >> 2);
    int blocks = ew_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_bilinear_up2_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, dout_ld, H, W, C, din,
                       din_ld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

struct SgPyramid { float* l[6]; int32_t ld[6]; };

// one workgroup per 64x64 tile of the label image: level s = mean over 2^(s+1) x 2^(s+1) pixels, built level by level in LDS
__global__ __launch_bounds__(256) void sg_avgpool_pyramid_fwd_kernel(const float* label, int ld, int H, int W, SgPyramid P) {
    __shared__ f32x4 lv[32 * 32];
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int y = i >> 5, x = i & 31;
        const float* p = label + ((int64_t)(ty0 + 2 * y) * W + (tx0 + 2 * x)) * ld;
        const f32x4 v = 0.25f * ((*reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + ld)) +
                                 (*reinterpret_cast<const f32x4*>(p + (int64_t)W * ld) + *reinterpret_cast<const f32x4*>(p + (int64_t)W * ld + ld)));
        lv[i] = v;
        *reinterpret_cast<f32x4*>(P.l[0] + ((int64_t)(ty0 / 2 + y) * (W / 2) + (tx0 / 2 + x)) * P.ld[0]) = v;
    }
    int n = 32;   // side of the level held in lv[] (row pitch stays 32)
    for (int s = 1; s < 6; ++s) {
        SG_SYNC();
        const int m = n >> 1;
        f32x4 v[4];
        int cnt = 0;
        for (int i = threadIdx.x; i < m * m; i += 256, ++cnt) {
            const int y = i / m, x = i % m;
            v[cnt] = 0.25f * ((lv[(2 * y) * 32 + 2 * x] + lv[(2 * y) * 32 + 2 * x + 1]) + (lv[(2 * y + 1) * 32 + 2 * x] + lv[(2 * y + 1) * 32 + 2 * x + 1]));
        }
        SG_SYNC();
        cnt = 0;
        for (int i = threadIdx.x; i < m * m; i += 256, ++cnt) {
            const int y = i / m, x = i % m;
            lv[y * 32 + x] = v[cnt];
            const int Ws = W >> (s + 1);
            *reinterpret_cast<f32x4*>(P.l[s] + ((int64_t)((ty0 >> (s + 1)) + y) * Ws + ((tx0 >> (s + 1)) + x)) * P.ld[s]) = v[cnt];
        }
        n = m;
    }
}

// dlabel[p] (+)= sum_s dl_s[p >> (s+1)] / 4^(s+1)
__global__ __launch_bounds__(256) void sg_avgpool_pyramid_bwd_kernel(SgPyramid D, int H, int W, float* dlabel, int ld, int accumulate) {
    const int64_t total = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
        const int x = (int)(p % W), y = (int)(p / W);
        f32x4 acc = accumulate ? *reinterpret_cast<const f32x4*>(dlabel + p * ld) : (f32x4){0.f, 0.f, 0.f, 0.f};
        float sc = 0.25f;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            if (D.l[s]) acc += sc * *reinterpret_cast<const f32x4*>(D.l[s] + ((int64_t)(y >> (s + 1)) * (W >> (s + 1)) + (x >> (s + 1))) * D.ld[s]);
            sc *= 0.25f;
        }
        *reinterpret_cast<f32x4*>(dlabel + p * ld) = acc;
    }
}

extern "C" int sgan_avgpool_pyramid_fwd(const float* label, int32_t ld, int32_t H, int32_t W, float* const* levels,
                                        const int32_t* level_ld, void* stream) {
    SGAN_CHECK(label && levels && level_ld && H > 0 && W > 0 && H % 64 == 0 && W % 64 == 0 && ld >= 4 && (ld & 3) == 0,
               "label must be a 4-channel (stored) image with sides divisible by 64");
    SgPyramid P;
    for (int s = 0; s < 6; ++s) {
        SGAN_CHECK(levels[s] && level_ld[s] >= 4 && (level_ld[s] & 3) == 0, "bad level %d", s);
        P.l[s] = levels[s];
        P.ld[s] = level_ld[s];
    }
    hipLaunchKernelGGL(sg_avgpool_pyramid_fwd_kernel, dim3(W / 64, H / 64), dim3(256), 0, (hipStream_t)stream, label, ld, H, W, P);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_avgpool_pyramid_bwd(const float* const* dlevels, const int32_t* level_ld, int32_t H, int32_t W, float* dlabel,
                                        int32_t ld, int32_t accumulate, void* stream) {
    SGAN_CHECK(dlevels && level_ld && dlabel && H % 64 == 0 && W % 64 == 0 && H > 0 && W > 0 && ld >= 4 && (ld & 3) == 0, "bad argument");
    SgPyramid P;
    for (int s = 0; s < 6; ++s) {
        P.l[s] = const_cast<float*>(dlevels[s]);
        P.ld[s] = level_ld[s];
    }
    int blocks = ew_cdiv((int64_t)H * W, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_avgpool_pyramid_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, H, W, dlabel, ld, accumulate);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// GAN loss on the logits map (one workgroup; the maps are <= 67x67)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(1024) void sg_gan_loss_fwd_kernel(const float* logits, int ld, int npix, float target, int mode,
                                                               float* loss_out, float* p_out) {
    __shared__ double wsum[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npix; i += 1024) {
        const float x = logits[(int64_t)i * ld];
        float l;
        if (mode == 0) {
            const float p = sg_sigmoid(x);
            if (p_out) p_out[(int64_t)i * ld] = p;
            const float lp = fmaxf(logf(p), -100.f);
            const float lq = fmaxf(log1pf(-p), -100.f);
            l = -(target * lp + (1.f - target) * lq);
        } else {
            const float d = x - target;
            l = d * d;
        }
        acc += (double)l;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    SG_SYNC();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += wsum[i];
        loss_out[0] = (float)(t / (double)npix);
    }
}

__global__ __launch_bounds__(256) void sg_gan_loss_bwd_kernel(const float* logits, int ld, int npix, float target, int mode,
                                                              const float* gout, float* dlogits, int dld) {
    const float go = gout[0] / (float)npix;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const float x = logits[(int64_t)i * ld];
        float d;
        if (mode == 0) {
            // torch: dL/dp = (p - t) / max((1 - p) * p, 1e-12), then sigmoid' = p (1 - p).  Identical to
            // (p - t) except where p(1-p) underflows below the clamp (|x| > ~27), where it decays to 0.
            const float p = sg_sigmoid(x);
            const float pq = (1.f - p) * p;
            d = (p - target) / fmaxf(pq, 1e-12f) * go * pq;
        } else {
            d = 2.f * (x - target) * go;
        }
        float* o = dlogits + (int64_t)i * dld;
        o[0] = d;
        for (int c = 1; c < dld; ++c) o[c] = 0.f;
    }
}

extern "C" int sgan_gan_loss_fwd(const float* logits, int32_t ld, int32_t npix, float target, int32_t mode, float* loss_out,
                                 float* p_out, void* stream) {
    SGAN_CHECK(logits && loss_out && npix > 0 && ld >= 1 && (mode == 0 || mode == 1), "bad argument");
    hipLaunchKernelGGL(sg_gan_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, ld, npix, target, mode,
                       loss_out, p_out);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_gan_loss_bwd(const float* logits, int32_t ld, int32_t npix, float target, int32_t mode, const float* gout,
                                 float* dlogits, int32_t dld, void* stream) {
    SGAN_CHECK(logits && gout && dlogits && npix > 0 && dld >= 1 && (mode == 0 || mode == 1), "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(sg_gan_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, ld, npix, target,
                       mode, gout, dlogits, dld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// All GAN-loss terms of one backward pass in one launch: total = sum_i weight_i * loss_i over up to 8
// logits maps (the three discriminators on the fake and the real batch), each term as in sg_gan_loss_*.
// Replaces the per-term Sigmoid+BCELoss modules AND the scalar adds/muls the trainers build around them
// ((fake + real) * 0.5, * lambda_D: models/fcgan_model.py:150-176).
// ------------------------------------------------------------------------------------------
struct SgLossMulti {
    const float* logits[8];
    float* dlogits[8];
    int32_t ld[8], dld[8], npix[8];
    float target[8], weight[8];
    int32_t n, mode;
};

// The log / exp arithmetic of a few 67x67 maps keeps a single CU busy for ~20 us, so the terms are spread over
// SG_LOSS_BLOCKS workgroups each: block (b, j) leaves the fp64 partial sum of its slice of term j in `part` and, where the
// job carries a gradient buffer, writes d total / d logits of its slice (for an upstream gradient of 1: the total is the scalar
// the trainers call backward() on).  The workgroup that arrives last at the counter behind `part` turns the partials into
// each[] and the weighted total and leaves the counter at zero for the next call: one launch, no zero fill per call.
#define SG_LOSS_BLOCKS 16
__global__ __launch_bounds__(256) void sg_gan_loss_multi_fwd_kernel(SgLossMulti J, double* part, unsigned* counter, float* each, float* total) {
    sg_warm_kernargs<(int)sizeof(SgLossMulti)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    __shared__ double wsum[4];
    __shared__ int last;
    const int j = blockIdx.y, b = blockIdx.x;
    const float tg = J.target[j];
    const int np = J.npix[j], ld = J.ld[j];
    const float* lg = J.logits[j];
    float* dl = J.dlogits[j];
    const int dld = J.dld[j];
    const float go = J.weight[j] / (float)np;
    double acc = 0.0;
    for (int i0 = b * 256 + threadIdx.x; i0 < np; i0 += SG_LOSS_BLOCKS * 256 * 4) {
        float xs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SG_LOSS_BLOCKS * 256;
            xs[u] = i < np ? lg[(int64_t)i * ld] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * SG_LOSS_BLOCKS * 256;
            if (i >= np) break;
            const float x = xs[u];
            float l, d;
            if (J.mode == 0) {
                const float p = sg_sigmoid(x);
                const float lp = fmaxf(logf(p), -100.f);
                const float lq = fmaxf(log1pf(-p), -100.f);
                l = -(tg * lp + (1.f - tg) * lq);
                const float pq = (1.f - p) * p;
                d = (p - tg) / fmaxf(pq, 1e-12f) * go * pq;     // the same expression as sg_gan_loss_multi_bwd_kernel
            } else {
                const float df = x - tg;
                l = df * df;
                d = 2.f * df * go;
            }
            acc += (double)l;
            if (dl) {
                float* o = dl + (int64_t)i * dld;
                o[0] = d;
                for (int c = 1; c < dld; ++c) o[c] = 0.f;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    SG_SYNC();
    if (threadIdx.x == 0) {
        part[j * SG_LOSS_BLOCKS + b] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        __threadfence();                                     // the partial is out before the ticket is taken
        last = atomicAdd(counter, 1u) == (unsigned)(SG_LOSS_BLOCKS * J.n - 1);
    }
    SG_SYNC();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();                                         // every other workgroup's partial is visible from here on
    const int t = threadIdx.x;
    double w = 0.0;
    if (t < J.n) {
        double sum = 0.0;
        for (int bb = 0; bb < SG_LOSS_BLOCKS; ++bb) sum += __builtin_nontemporal_load(&part[t * SG_LOSS_BLOCKS + bb]);
        const float m = (float)(sum / (double)J.npix[t]);
        each[t] = m;
        w = (double)J.weight[t] * (double)m;
    }
    for (int off = 4; off > 0; off >>= 1) w += __shfl_xor(w, off);   // n <= 8 terms sit in lanes 0..7
    if (t == 0) {
        total[0] = (float)w;
        counter[0] = 0u;
    }
}

__global__ __launch_bounds__(256) void sg_gan_loss_multi_bwd_kernel(SgLossMulti J, const float* gout) {
    sg_warm_kernargs<(int)sizeof(SgLossMulti)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    const int j = blockIdx.y;
    const float go = gout[0] * J.weight[j] / (float)J.npix[j];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < J.npix[j]; i += gridDim.x * 256) {
        const float x = J.logits[j][(int64_t)i * J.ld[j]];
        float d;
        if (J.mode == 0) {
            const float p = sg_sigmoid(x);
            const float pq = (1.f - p) * p;
            d = (p - J.target[j]) / fmaxf(pq, 1e-12f) * go * pq;
        } else {
            d = 2.f * (x - J.target[j]) * go;
        }
        float* o = J.dlogits[j] + (int64_t)i * J.dld[j];
        o[0] = d;
        for (int c = 1; c < J.dld[j]; ++c) o[c] = 0.f;
    }
}

static int sg_fill_loss(SgLossMulti& J, const sgan_gan_loss_job* jobs, int n, int mode, bool bwd) {
    if (!jobs || n < 1 || n > 8 || (mode != 0 && mode != 1)) return sgan_fail(SGAN_ERR_INVALID, "1..8 loss jobs, mode 0/1");
    memset(&J, 0, sizeof(J));
    J.n = n;
    J.mode = mode;
    for (int i = 0; i < n; ++i) {
        if (!jobs[i].logits || jobs[i].npix <= 0 || jobs[i].ld < 1 || (bwd && (!jobs[i].dlogits || jobs[i].dld < 1)))
            return sgan_fail(SGAN_ERR_INVALID, "bad loss job %d", i);
        J.logits[i] = jobs[i].logits; J.dlogits[i] = jobs[i].dlogits; J.ld[i] = jobs[i].ld; J.dld[i] = jobs[i].dld;
        J.npix[i] = jobs[i].npix; J.target[i] = jobs[i].target; J.weight[i] = jobs[i].weight;
    }
    return SGAN_OK;
}

extern "C" int sgan_gan_loss_multi_fwd(const sgan_gan_loss_job* jobs, int32_t n, int32_t mode, float* each_out,
                                       float* total_out, void* workspace, int64_t workspace_bytes, void* stream) {
    SgLossMulti J;
    int rc = sg_fill_loss(J, jobs, n, mode, false);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) SGAN_CHECK(!jobs[i].dlogits || jobs[i].dld >= 1, "bad gradient leading dimension of loss job %d", i);
    SGAN_CHECK(each_out && total_out, "null output");
    SGAN_CHECK(workspace && workspace_bytes >= SGAN_GAN_LOSS_WS_BYTES && ((uintptr_t)workspace & 7) == 0,
               "workspace of SGAN_GAN_LOSS_WS_BYTES (8-byte aligned) required");
    static_assert((8 * SG_LOSS_BLOCKS + 1) * sizeof(double) <= SGAN_GAN_LOSS_WS_BYTES, "workspace size");
    double* part = static_cast<double*>(workspace);
    unsigned* counter = reinterpret_cast<unsigned*>(part + 8 * SG_LOSS_BLOCKS);
    hipLaunchKernelGGL(sg_gan_loss_multi_fwd_kernel, dim3(SG_LOSS_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, J, part, counter, each_out,
                       total_out);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_gan_loss_multi_bwd(const sgan_gan_loss_job* jobs, int32_t n, int32_t mode, const float* gout,
                                       void* stream) {
    SgLossMulti J;
    int rc = sg_fill_loss(J, jobs, n, mode, true);
    if (rc) return rc;
    SGAN_CHECK(gout, "null gout");
    int maxp = 0;
    for (int i = 0; i < n; ++i) maxp = max(maxp, J.npix[i]);
    hipLaunchKernelGGL(sg_gan_loss_multi_bwd_kernel, dim3((maxp + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, J, gout);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// standalone sigmoid on channel 0 of a logits map (only used when the caller wants probabilities;
// the training path feeds the logits straight to sgan_gan_loss_*)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_sigmoid_fwd_kernel(const float* x, int ld, int npix, float* p, int pld) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        float* o = p + (int64_t)i * pld;
        o[0] = sg_sigmoid(x[(int64_t)i * ld]);
        for (int c = 1; c < pld; ++c) o[c] = 0.f;
    }
}

__global__ __launch_bounds__(256) void sg_sigmoid_bwd_kernel(const float* dp, int dpld, const float* p, int pld, int npix,
                                                             float* dx, int dxld) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const float pv = p[(int64_t)i * pld];
        float* o = dx + (int64_t)i * dxld;
        o[0] = dp[(int64_t)i * dpld] * pv * (1.f - pv);
        for (int c = 1; c < dxld; ++c) o[c] = 0.f;
    }
}

extern "C" int sgan_sigmoid_fwd(const float* x, int32_t ld, int32_t npix, float* p, int32_t pld, void* stream) {
    SGAN_CHECK(x && p && npix > 0, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(sg_sigmoid_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, npix, p, pld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_sigmoid_bwd(const float* dp, int32_t dpld, const float* p, int32_t pld, int32_t npix, float* dx,
                                int32_t dxld, void* stream) {
    SGAN_CHECK(dp && p && dx && npix > 0, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(sg_sigmoid_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dp, dpld, p, pld, npix, dx, dxld);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// tanh backward, layout boundary copy
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_tanh_bwd_kernel(const float* dy, const float* y, float* dx, int64_t n4) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const f32x4 d = reinterpret_cast<const f32x4*>(dy)[e];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[e];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = d[j] * (1.f - v[j] * v[j]);
        reinterpret_cast<f32x4*>(dx)[e] = o;
    }
}

extern "C" int sgan_tanh_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    SGAN_CHECK(dy && y && dx && n > 0 && (n & 3) == 0, "bad argument");
    int blocks = ew_cdiv(n / 4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sg_tanh_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n / 4);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// out = tanh(a + b) (or a + b): the `--use_residual` tail of ResnetGenerator / UnetGenerator (models/networks.py:268, :367); the
// backward is sgan_tanh_bwd of `out`, one gradient for both addends
template <bool TANH>
__global__ __launch_bounds__(256) void sg_add_act_kernel(const float* a, const float* b, float* out, int64_t n4) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const f32x4 u = reinterpret_cast<const f32x4*>(a)[e];
        const f32x4 v = reinterpret_cast<const f32x4*>(b)[e];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = TANH ? tanhf(u[j] + v[j]) : u[j] + v[j];
        reinterpret_cast<f32x4*>(out)[e] = o;
    }
}

extern "C" int sgan_add_act_fwd(const float* a, const float* b, float* out, int64_t n, int32_t act, void* stream) {
    SGAN_CHECK(a && b && out && n > 0 && (n & 3) == 0, "bad argument");
    SGAN_CHECK(act == SGAN_ACT_NONE || act == SGAN_ACT_TANH, "act must be SGAN_ACT_NONE or SGAN_ACT_TANH");
    int blocks = ew_cdiv(n / 4, 256);
    if (blocks > 2048) blocks = 2048;
    if (act == SGAN_ACT_TANH) hipLaunchKernelGGL(sg_add_act_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);
    else hipLaunchKernelGGL(sg_add_act_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

__global__ __launch_bounds__(256) void sg_to_nhwc_kernel(const float* src, int64_t sc, int64_t sh, int64_t sw, int H, int W,
                                                         int Creal, float* dst, int dst_ld, int Cstore) {
    const int64_t total = (int64_t)H * W * Cstore;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % Cstore);
        const int64_t pix = e / Cstore;
        const int x = (int)(pix % W), y = (int)(pix / W);
        dst[pix * dst_ld + c] = c < Creal ? src[c * sc + y * sh + x * sw] : 0.f;
    }
}

// planar source with unit pixel stride and <= 4 channels (an NCHW image batch): one thread per pixel reads each plane
// coalesced and writes one 16-byte NHWC pixel.  `src` may be PINNED HOST memory (it is mapped into the device's address
// space): the kernel then IS the host-to-device copy of the batch, on the compute queue of the step's stream.
__global__ __launch_bounds__(256) void sg_planes_to_nhwc4_kernel(const float* src, int64_t sc, int64_t sh, int H, int W, int Creal,
                                                                 float* dst, int dst_ld) {
    // four consecutive pixels of a row per thread: one 16-byte load per plane (W % 4 == 0 and 16-byte aligned rows: the fast path;
    // round 2 walked single pixels with 64-bit divisions -- 16 us for a 512 x 512 batch already in HBM), four 16-byte stores
    const int W4 = W >> 2;
    const bool fast = (W & 3) == 0 && (sh & 3) == 0 && (sc & 3) == 0 && ((uintptr_t)src & 15) == 0;
    const int total = fast ? H * W4 : H * W;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        if (fast) {
            const int y = t / W4, x = (t - y * W4) * 4;
            f32x4 pl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                pl[c] = c < Creal ? *reinterpret_cast<const f32x4*>(src + c * sc + (int64_t)y * sh + x) : (f32x4){0.f, 0.f, 0.f, 0.f};
            float* o = dst + ((int64_t)y * W + x) * dst_ld;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(o + (int64_t)i * dst_ld) = (f32x4){pl[0][i], pl[1][i], pl[2][i], pl[3][i]};
        } else {
            const int y = t / W, x = t - y * W;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < Creal) v[c] = src[c * sc + (int64_t)y * sh + x];
            *reinterpret_cast<f32x4*>(dst + (int64_t)t * dst_ld) = v;
        }
    }
}

extern "C" int sgan_to_nhwc(const float* src, int64_t sc, int64_t sh, int64_t sw, int32_t H, int32_t W, int32_t Creal,
                            float* dst, int32_t dst_ld, int32_t Cstore, void* stream) {
    SGAN_CHECK(src && dst && H > 0 && W > 0 && Creal <= Cstore && dst_ld >= Cstore, "bad argument");
    if (sw == 1 && Cstore == 4 && (dst_ld & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        int blocks = ew_cdiv((int64_t)H * W, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(sg_planes_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, sc, sh, H, W, Creal, dst, dst_ld);
        SGAN_LAUNCH_CHECK();
        return SGAN_OK;
    }
    const int64_t total = (int64_t)H * W * Cstore;
    int blocks = ew_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_to_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, sc, sh, sw, H, W, Creal, dst,
                       dst_ld, Cstore);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// (label, image) pair of the conditional discriminators: torch.cat((A, B), 1) of two NHWC buffers as ONE pass that writes the
// padded NHWC buffer the discriminator reads, and its backward (the channel slice of the pair's gradient that belongs to one
// member).  One thread per pixel, whole padded pixels in and out.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_concat_nhwc_kernel(const float* a, int a_ld, int Ca, const float* b, int b_ld, int Cb,
                                                             int64_t npix, float* dst, int dst_ld, int Cstore) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const float* pa = a + p * a_ld;
        const float* pb = b + p * b_ld;
        float* o = dst + p * dst_ld;
        for (int c = 0; c < Cstore; ++c) o[c] = c < Ca ? pa[c] : (c < Ca + Cb ? pb[c - Ca] : 0.f);
    }
}

__global__ __launch_bounds__(256) void sg_slice_nhwc_kernel(const float* src, int src_ld, int c0, int C, int64_t npix, float* dst,
                                                            int dst_ld, int Cstore) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const float* ps = src + p * src_ld + c0;
        float* o = dst + p * dst_ld;
        for (int c = 0; c < Cstore; ++c) o[c] = c < C ? ps[c] : 0.f;
    }
}

extern "C" int sgan_concat_nhwc(const float* a, int32_t a_ld, int32_t Ca, const float* b, int32_t b_ld, int32_t Cb, int64_t npix,
                                float* dst, int32_t dst_ld, int32_t Cstore, void* stream) {
    SGAN_CHECK(a && b && dst && npix > 0 && Ca > 0 && Cb > 0 && a_ld >= Ca && b_ld >= Cb && Cstore >= Ca + Cb && dst_ld >= Cstore, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_concat_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, a_ld, Ca, b, b_ld, Cb, npix, dst, dst_ld, Cstore);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_slice_nhwc(const float* src, int32_t src_ld, int32_t c0, int32_t C, int64_t npix, float* dst, int32_t dst_ld,
                               int32_t Cstore, void* stream) {
    SGAN_CHECK(src && dst && npix > 0 && c0 >= 0 && C > 0 && src_ld >= c0 + C && Cstore >= C && dst_ld >= Cstore, "bad argument");
    int blocks = ew_cdiv(npix, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sg_slice_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src_ld, c0, C, npix, dst, dst_ld, Cstore);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Adam: one prep launch (step counter + bias corrections, fp64) and one streaming launch
// ------------------------------------------------------------------------------------------
struct SgAdamTable {
    sgan_adam_seg s[64];
    int nseg;
};

__global__ void sg_adam_prep_kernel(int32_t* state, const float* lr, float b1, float b2) {
    const int t = state[0] + 1;
    state[0] = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    reinterpret_cast<float*>(state)[1] = (float)((double)lr[0] / bc1);
    reinterpret_cast<float*>(state)[2] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void sg_adam_kernel(SgAdamTable T, const int32_t* state, float b1, float b2, float eps) {
    const sgan_adam_seg& S = T.s[blockIdx.y];
    const float step_size = reinterpret_cast<const float*>(state)[1];
    const float inv_sqrt_bc2 = reinterpret_cast<const float*>(state)[2];
    const int64_t n4 = S.n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        f32x4 p = reinterpret_cast<f32x4*>(S.p)[e];
        const f32x4 g = reinterpret_cast<const f32x4*>(S.g)[e];
        f32x4 m = reinterpret_cast<f32x4*>(S.m)[e];
        f32x4 v = reinterpret_cast<f32x4*>(S.v)[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = b1 * m[j] + (1.f - b1) * g[j];
            v[j] = b2 * v[j] + (1.f - b2) * g[j] * g[j];
            const float denom = sqrtf(v[j]) * inv_sqrt_bc2 + eps;
            p[j] -= step_size * (m[j] / denom);
        }
        reinterpret_cast<f32x4*>(S.p)[e] = p;
        reinterpret_cast<f32x4*>(S.m)[e] = m;
        reinterpret_cast<f32x4*>(S.v)[e] = v;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (S.n & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        const float g = S.g[e];
        const float m = b1 * S.m[e] + (1.f - b1) * g;
        const float v = b2 * S.v[e] + (1.f - b2) * g * g;
        S.m[e] = m;
        S.v[e] = v;
        S.p[e] -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    }
}

extern "C" int sgan_adam_multi(const sgan_adam_seg* segs, int32_t nseg, const float* lr_dev, float beta1, float beta2,
                               float eps, int32_t* state_dev, void* stream) {
    SGAN_CHECK(segs && nseg > 0 && nseg <= 64 && lr_dev && state_dev, "bad argument");
    SgAdamTable T;
    int64_t maxn = 0;
    for (int i = 0; i < nseg; ++i) {
        T.s[i] = segs[i];
        SGAN_CHECK(segs[i].p && segs[i].g && segs[i].m && segs[i].v && segs[i].n > 0, "bad segment %d", i);
        SGAN_CHECK(((uintptr_t)segs[i].p & 15) == 0 && ((uintptr_t)segs[i].g & 15) == 0 && ((uintptr_t)segs[i].m & 15) == 0 &&
                       ((uintptr_t)segs[i].v & 15) == 0, "segment %d not 16-byte aligned", i);
        if (segs[i].n > maxn) maxn = segs[i].n;
    }
    T.nseg = nseg;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sg_adam_prep_kernel, dim3(1), dim3(1), 0, st, state_dev, lr_dev, beta1, beta2);
    SGAN_LAUNCH_CHECK();
    int bx = ew_cdiv(maxn / 4 + 1, 256 * 2);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(sg_adam_kernel, dim3(bx, nseg), dim3(256), 0, st, T, state_dev, beta1, beta2, eps);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// sgan_adam_pack: the whole optimizer step of one flat arena segment in ONE launch (round 2: adam_prep + adam + pack_weights,
// nine launches per fcgan step).  Workgroups [0, ntiles) own one 32 x 32 (co, ci) tile of one tap of a conv weight range: Adam on the
// tile, the updated tile stays in LDS and leaves as the master value, the fp32 transposed copy and the two split 16-bit copies
// (exactly sg_pack_weights_kernel's writes).  Workgroups behind them run plain Adam over the ranges between the conv weights
// (biases, BatchNorm affines), 1024 elements each.  The step counter: every workgroup reads t, derives the two bias corrections
// in fp64 (thread 0, broadcast through LDS), then takes a ticket; whoever draws the last ticket knows that every other workgroup
// has read t already and stores t + 1 (and clears the ticket) -- a captured hipGraph replays with the right t.
// zero_grads: the consumed gradient is overwritten with zeros (the caller's next zero_grad() is then a no-op).
// ------------------------------------------------------------------------------------------
struct SgAdamPackTable {
    sgan_wt_seg s[64];
    int32_t first[65];          // first tile of conv range i
    int64_t gap0[65], gapn[65]; // ranges outside every conv weight
    int32_t gfirst[66];         // first 1024-element chunk of gap i
    int32_t n, ngap, ntiles;
};

__device__ __forceinline__ void sg_adam1(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size, float inv_sqrt_bc2) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
}

__global__ __launch_bounds__(256) void sg_adam_pack_kernel(float* P, float* Gr, float* M, float* V, float* flat_t, float* pk_f, float* pk_b,
                                                           float* pk_bh, const SgAdamPackTable T, int32_t* state, const float* lr, float b1, float b2,
                                                           float eps, int zero_grads, int nitems) {
    sg_warm_kernargs<(int)sizeof(SgAdamPackTable)>();      // sgan_common.h: the scalar-cache misses of the parameter block, taken together
    __shared__ float tile[32][33];
    __shared__ float bc[2];
    int t_step = 0;
    if (threadIdx.x == 0) {
        t_step = state[0] + 1;
        const double bc1 = 1.0 - pow((double)b1, (double)t_step);
        const double bc2 = 1.0 - pow((double)b2, (double)t_step);
        bc[0] = (float)((double)lr[0] / bc1);
        bc[1] = (float)(1.0 / sqrt(bc2));
    }
    SG_SYNC();
    const float step_size = bc[0], inv_sqrt_bc2 = bc[1];
    // a few hundred workgroups walk the work items: the ticket below is one same-address atomic per WORKGROUP (~12 ns each at the
    // memory side; one workgroup per tile was 3700 of them per launch -- 40 us of serialised atomics, measured as a slower step)
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        if (item >= T.ntiles) {       // plain ranges
            const int c = item - T.ntiles;
            int i = 0;
            for (int k = 1; k < T.ngap; ++k)
                if (c >= T.gfirst[k]) i = k;
            const int64_t e0 = T.gap0[i] + (int64_t)(c - T.gfirst[i]) * 1024, e1 = T.gap0[i] + T.gapn[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t e = e0 + u * 256 + threadIdx.x;
                if (e < e1) {
                    float p = P[e], m = M[e], v = V[e];
                    sg_adam1(p, Gr[e], m, v, b1, b2, eps, step_size, inv_sqrt_bc2);
                    P[e] = p; M[e] = m; V[e] = v;
                    if (zero_grads) Gr[e] = 0.f;
                }
            }
            continue;
        }
        int i = 0;
        for (int k = 1; k < T.n; ++k)
            if (item >= T.first[k]) i = k;
        const sgan_wt_seg S = T.s[i];
        int t = item - T.first[i];
        const int tc = (S.cin + 31) / 32, tr = (S.cout + 31) / 32;
        const int bx = t % tc; t /= tc;
        const int by = t % tr; t /= tr;      // t = tap
        const int64_t slab = S.off + t * (int64_t)S.cout * S.cin;
        const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
        for (int r = ly; r < 32; r += 8) {
            const int co = by * 32 + r, ci = bx * 32 + lx;
            float p = 0.f;
            if (co < S.cout && ci < S.cin) {
                const int64_t e = slab + (int64_t)co * S.cin + ci;
                float m = M[e], v = V[e];
                p = P[e];
                sg_adam1(p, Gr[e], m, v, b1, b2, eps, step_size, inv_sqrt_bc2);
                P[e] = p; M[e] = m; V[e] = v;
                if (zero_grads) Gr[e] = 0.f;
            }
            tile[r][lx] = p;
        }
        SG_SYNC();
        if (flat_t) {
            float* dst = flat_t + slab;
            for (int r = ly; r < 32; r += 8) {
                const int ci = bx * 32 + r, co = by * 32 + lx;
                if (ci < S.cin && co < S.cout) dst[(int64_t)ci * S.cout + co] = tile[lx][r];
            }
        }
        const int r = threadIdx.x >> 3, q = (threadIdx.x >> 1) & 3, pl = threadIdx.x & 1;
        if (pk_f && (S.cin & 7) == 0) {
            const int co = by * 32 + r, ci = bx * 32 + q * 8;
            if (co < S.cout && ci < S.cin) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[r][q * 8 + e];
                *reinterpret_cast<sg_u32x4*>(pk_f + slab + (int64_t)co * S.cin + ci + 4 * pl) = sg_split8_f16(v, pl);
            }
        }
        if (pk_b && (S.cout & 7) == 0) {
            const int ci = bx * 32 + r, co = by * 32 + q * 8;
            if (ci < S.cin && co < S.cout) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tile[q * 8 + e][r];
                *reinterpret_cast<sg_u32x4*>(pk_b + slab + (int64_t)ci * S.cout + co + 4 * pl) = sg_split8(v, pl);
                if (pk_bh) *reinterpret_cast<sg_u32x4*>(pk_bh + slab + (int64_t)ci * S.cout + co + 4 * pl) = sg_split8_f16(v, pl);
            }
        }
        SG_SYNC();      // the tile is rewritten by the next item
    }
    if (threadIdx.x == 0) {
        // this workgroup read state[0] long ago (its value went into the bias corrections): it may be counted.  Whoever draws the
        // last ticket knows that every workgroup has read t, and moves the counter on
        const unsigned ticket = atomicAdd(reinterpret_cast<unsigned*>(state) + 3, 1u);
        if (ticket == gridDim.x - 1) {
            state[0] = t_step;
            reinterpret_cast<float*>(state)[1] = step_size;
            reinterpret_cast<float*>(state)[2] = inv_sqrt_bc2;
            reinterpret_cast<unsigned*>(state)[3] = 0u;
        }
    }
}

extern "C" int sgan_adam_pack(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2, float eps,
                              int32_t* state_dev, float* flat_t, void* packed_fwd, void* packed_bwd, void* packed_bwd_f16,
                              const sgan_wt_seg* segs, int32_t nseg, int32_t zero_grads, void* stream) {
    SGAN_CHECK(p && g && m && v && n > 0 && lr_dev && state_dev && nseg >= 0 && nseg <= 64 && (nseg == 0 || segs), "bad argument");
    SgAdamPackTable T;
    memset(&T, 0, sizeof(T));
    int64_t cur = 0;
    int tiles = 0, chunks = 0;
    for (int i = 0; i < nseg; ++i) {
        const int64_t len = (int64_t)segs[i].taps * segs[i].cout * segs[i].cin;
        SGAN_CHECK(segs[i].taps > 0 && segs[i].cout > 0 && segs[i].cin > 0 && (segs[i].off & 3) == 0, "bad conv range %d", i);
        SGAN_CHECK(segs[i].off >= cur && segs[i].off + len <= n, "conv ranges must be sorted, disjoint and inside the segment (range %d)", i);
        if (segs[i].off > cur) {
            T.gap0[T.ngap] = cur; T.gapn[T.ngap] = segs[i].off - cur; T.gfirst[T.ngap] = chunks;
            chunks += (int)((segs[i].off - cur + 1023) / 1024);
            ++T.ngap;
        }
        T.s[i] = segs[i];
        T.first[i] = tiles;
        tiles += segs[i].taps * ((segs[i].cout + 31) / 32) * ((segs[i].cin + 31) / 32);
        cur = segs[i].off + len;
    }
    if (cur < n) {
        T.gap0[T.ngap] = cur; T.gapn[T.ngap] = n - cur; T.gfirst[T.ngap] = chunks;
        chunks += (int)((n - cur + 1023) / 1024);
        ++T.ngap;
    }
    T.n = nseg; T.ntiles = tiles; T.first[nseg] = tiles; T.gfirst[T.ngap] = chunks;
    static const int max_wg = getenv("SGAN_ADAM_WGS") ? atoi(getenv("SGAN_ADAM_WGS")) : 1024;      // tuning knob
    const int nitems = tiles + chunks;
    hipLaunchKernelGGL(sg_adam_pack_kernel, dim3((unsigned)(nitems < max_wg ? nitems : max_wg)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       flat_t, (float*)packed_fwd, (float*)packed_bwd, (float*)packed_bwd_f16, T, state_dev, lr_dev, beta1, beta2, eps, (int)zero_grads, nitems);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// Zero up to 64 buffers in one launch (statistics arenas of a training step: ops.begin_step).  Sizes in bytes, multiples of 16.
struct SgZeroTable { void* p[64]; int64_t n16[64]; int32_t n; };
__global__ __launch_bounds__(256) void sg_zero_multi_kernel(const SgZeroTable T) {
    f32x4* dst = reinterpret_cast<f32x4*>(T.p[blockIdx.y]);
    const int64_t n = T.n16[blockIdx.y];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) dst[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
extern "C" int sgan_zero_multi(void* const* ptrs, const int64_t* bytes, int32_t n, void* stream) {
    SGAN_CHECK(ptrs && bytes && n >= 1 && n <= 64, "1..64 buffers");
    SgZeroTable T;
    int64_t maxn = 0;
    for (int i = 0; i < n; ++i) {
        SGAN_CHECK(ptrs[i] && bytes[i] > 0 && (bytes[i] & 15) == 0 && ((uintptr_t)ptrs[i] & 15) == 0, "buffer %d: 16-byte aligned pointer and size", i);
        T.p[i] = ptrs[i]; T.n16[i] = bytes[i] >> 4;
        if (T.n16[i] > maxn) maxn = T.n16[i];
    }
    T.n = n;
    int bx = (int)((maxn + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(sg_zero_multi_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, T);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// SGD (torch.optim.SGD form: buf = mu * buf + g, p -= lr * buf; dampening 0, no Nesterov, no weight decay) over the same
// segment table as Adam; `m` is the momentum buffer (NULL or momentum == 0: plain p -= lr * g), `v` is unused.  The first
// step of torch's SGD sets buf = g, which the zero-initialised buffer reproduces.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_sgd_kernel(SgAdamTable T, const float* lr_dev, float mu) {
    const sgan_adam_seg& S = T.s[blockIdx.y];
    const float lr = lr_dev[0];
    const bool mom = mu != 0.f && S.m != nullptr;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < S.n; e += (int64_t)gridDim.x * 256) {
        float g = S.g[e];
        if (mom) {
            g = mu * S.m[e] + g;
            S.m[e] = g;
        }
        S.p[e] -= lr * g;
    }
}

extern "C" int sgan_sgd_multi(const sgan_adam_seg* segs, int32_t nseg, const float* lr_dev, float momentum, void* stream) {
    SGAN_CHECK(segs && nseg > 0 && nseg <= 64 && lr_dev, "bad argument");
    SgAdamTable T;
    int64_t maxn = 0;
    for (int i = 0; i < nseg; ++i) {
        T.s[i] = segs[i];
        SGAN_CHECK(segs[i].p && segs[i].g && segs[i].n > 0, "bad segment %d", i);
        if (segs[i].n > maxn) maxn = segs[i].n;
    }
    T.nseg = nseg;
    int bx = ew_cdiv(maxn, 256 * 4);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(sg_sgd_kernel, dim3(bx, nseg), dim3(256), 0, (hipStream_t)stream, T, lr_dev, momentum);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

// ------------------------------------------------------------------------------------------
// N(0,1) fill: Philox4x32-10, Box-Muller
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sg_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                          uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// advance_by != 0 (single-block launches only): the block moves the stream offset itself once every thread has read it
// cs != 0: element i of the logical [C][H][W] tensor goes to the NHWC buffer position (i % hw) * cs + i / hw
__global__ __launch_bounds__(256) void sg_normal_fill_kernel(float* dst, int64_t n, uint64_t seed, uint64_t* offset, uint64_t advance_by,
                                                             int hw, int cs) {
    const uint64_t off = offset ? offset[0] : 0;
    if (advance_by) {
        SG_SYNC();
        if (threadIdx.x == 0) offset[0] = off + advance_by;
    }
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        const uint64_t ctr = off + (uint64_t)q;
        uint32_t r[4];
        sg_philox((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
            const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);       // [0,1)
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.28318530717958647692f * u2, &sn, &cs);
            z[2 * h] = rad * cs;
            z[2 * h + 1] = rad * sn;
        }
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < n) dst[cs ? (i % hw) * cs + i / hw : i] = z[j];
        }
    }
}

__global__ void sg_rng_advance_kernel(uint64_t* offset, uint64_t by) { offset[0] += by; }

// Two small latents drawn back to back (dst_a first: the values two sg_normal_fill_kernel launches give) by ONE block, which also
// zeroes `zero` (zero_n16 x 16 bytes: the statistics arena of the generator pass that reads the latents) and advances the stream.
__global__ __launch_bounds__(256) void sg_normal_fill_pair_kernel(float* dst_a, float* dst_b, int64_t n, uint64_t seed, uint64_t* offset,
                                                                  int hw, int cs, float4* zero, int64_t zero_n16) {
    const uint64_t off = offset[0];
    SG_SYNC();
    const int64_t nq = (n + 3) >> 2;
    if (threadIdx.x == 0) offset[0] = off + 2 * (uint64_t)nq;
    for (int64_t i = threadIdx.x; i < zero_n16; i += 256) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q2 = threadIdx.x; q2 < 2 * nq; q2 += 256) {
        const uint64_t ctr = off + (uint64_t)q2;
        float* dst = q2 < nq ? dst_a : dst_b;
        const int64_t q = q2 < nq ? q2 : q2 - nq;
        uint32_t r[4];
        sg_philox((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1]
            const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);       // [0,1)
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, csn;
            sincosf(6.28318530717958647692f * u2, &sn, &csn);
            z[2 * h] = rad * csn;
            z[2 * h + 1] = rad * sn;
        }
        for (int j = 0; j < 4; ++j) {
            const int64_t i = q * 4 + j;
            if (i < n) dst[cs ? (i % hw) * cs + i / hw : i] = z[j];
        }
    }
}

__global__ __launch_bounds__(256) void sg_dropout_mask_kernel(float* mask, int64_t n, float p, float keep_scale, uint64_t seed,
                                                              const uint64_t* offset) {
    const uint64_t off = offset ? offset[0] : 0;
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
        const uint64_t ctr = off + (uint64_t)q;
        uint32_t r[4];
        sg_philox((uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) mask[q * 4 + j] = ((float)(r[j] >> 8) * (1.0f / 16777216.0f) < p) ? 0.f : keep_scale;
    }
}

extern "C" int sgan_rng_advance(uint64_t* offset_dev, uint64_t by, void* stream) {
    SGAN_CHECK(offset_dev, "null offset");
    hipLaunchKernelGGL(sg_rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, offset_dev, by);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, uint64_t* offset_dev, int32_t advance, void* stream) {
    SGAN_CHECK(mask && n > 0 && p >= 0.f && p < 1.f, "bad argument");
    const int64_t nq = (n + 3) >> 2;
    int blocks = ew_cdiv(nq, 256);
    if (blocks > 1024) blocks = 1024;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sg_dropout_mask_kernel, dim3(blocks), dim3(256), 0, st, mask, n, p, 1.f / (1.f - p), seed, offset_dev);
    SGAN_LAUNCH_CHECK();
    if (offset_dev && advance) {
        hipLaunchKernelGGL(sg_rng_advance_kernel, dim3(1), dim3(1), 0, st, offset_dev, (uint64_t)nq);
        SGAN_LAUNCH_CHECK();
    }
    return SGAN_OK;
}

static int sg_normal_fill_launch(float* dst, int64_t n, int hw, int cs, uint64_t seed, uint64_t* offset_dev, int32_t advance, void* stream) {
    const int64_t nq = (n + 3) >> 2;
    int blocks = ew_cdiv(nq, 256);
    if (blocks > 1024) blocks = 1024;
    hipStream_t st = (hipStream_t)stream;
    if (!advance) {
        hipLaunchKernelGGL(sg_normal_fill_kernel, dim3(blocks), dim3(256), 0, st, dst, n, seed, offset_dev, (uint64_t)0, hw, cs);
        SGAN_LAUNCH_CHECK();
        return SGAN_OK;
    }
    const bool self_advance = offset_dev && blocks == 1;
    hipLaunchKernelGGL(sg_normal_fill_kernel, dim3(blocks), dim3(256), 0, st, dst, n, seed, offset_dev, self_advance ? (uint64_t)nq : 0, hw, cs);
    SGAN_LAUNCH_CHECK();
    if (offset_dev && !self_advance) {
        hipLaunchKernelGGL(sg_rng_advance_kernel, dim3(1), dim3(1), 0, st, offset_dev, (uint64_t)nq);
        SGAN_LAUNCH_CHECK();
    }
    return SGAN_OK;
}

extern "C" int sgan_normal_fill(float* dst, int64_t n, uint64_t seed, uint64_t* offset_dev, int32_t advance, void* stream) {
    SGAN_CHECK(dst && n > 0, "bad argument");
    return sg_normal_fill_launch(dst, n, 1, 0, seed, offset_dev, advance, stream);
}

extern "C" int sgan_normal_fill_nhwc_pair(float* dst_a, float* dst_b, int32_t C, int32_t H, int32_t W, int32_t Cs, uint64_t seed,
                                          uint64_t* offset_dev, void* zero, int64_t zero_bytes, void* stream) {
    SGAN_CHECK(dst_a && dst_b && dst_a != dst_b && offset_dev && C > 0 && H > 0 && W > 0 && Cs >= C, "bad argument");
    const int64_t n = (int64_t)C * H * W;
    SGAN_CHECK(n <= 65536 && zero_bytes >= 0 && zero_bytes <= (1 << 22) && (zero_bytes % 16) == 0 && (zero || zero_bytes == 0),
               "one block draws both latents: <= 65536 values each, <= 4 MiB (multiple of 16 bytes) to zero");
    hipLaunchKernelGGL(sg_normal_fill_pair_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dst_a, dst_b, n, seed, offset_dev, H * W, Cs,
                       (float4*)zero, zero_bytes / 16);
    SGAN_LAUNCH_CHECK();
    return SGAN_OK;
}

extern "C" int sgan_normal_fill_nhwc(float* dst, int32_t C, int32_t H, int32_t W, int32_t Cs, uint64_t seed, uint64_t* offset_dev,
                                     int32_t advance, void* stream) {
    SGAN_CHECK(dst && C > 0 && H > 0 && W > 0 && Cs >= C, "bad argument");
    return sg_normal_fill_launch(dst, (int64_t)C * H * W, H * W, Cs, seed, offset_dev, advance, stream);
}
