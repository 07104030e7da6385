// Internal definitions shared by the gfx950 kernels of libsgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/sgan_hip.h"

// Every workgroup barrier of the library goes through SG_SYNC().  A stress build (tools/race_stress_all.sh, -DSG_STRESS_DELAY=<wave>)
// holds one wave of every workgroup back for ~6400 cycles after each barrier, so that the other waves run as far ahead as the
// barriers let them: an LDS buffer that is overwritten without a barrier after its last read then gives wrong results every time
// instead of once in a thousand runs.  The product build is the plain barrier.
#ifdef SG_STRESS_DELAY
#define SG_SYNC()                                                                                   \
    do {                                                                                            \
        __syncthreads();                                                                            \
        if ((int)(threadIdx.x >> 6) == SG_STRESS_DELAY) asm volatile("s_sleep 100" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#else
#define SG_SYNC() __syncthreads()
#endif

// The parameter blocks of the conv kernels are 1-2.5 KB of kernel arguments read through the scalar cache, and the set-up walks them
// in three or four DEPENDENT steps (which problem is this workgroup's -> that problem's fields -> the phase's extents -> ...): in-kernel
// stamps put 4000-5000 cycles (2 us) between a workgroup's first instruction and its tap table, nearly all of it scalar-cache misses
// taken one after the other.  Touching one dword of every 64-byte line up front takes the misses TOGETHER (one round trip); the
// dependent reads that follow hit the cache.
// One asm statement: every load AND the wait (the compiler does not know that an asm output arrives late: with the wait outside it
// could hand the destination register to another value while the load is still in flight).  All loads share one scratch register.
// The base comes from __builtin_amdgcn_kernarg_segment_ptr(): taking the address of the by-value parameter itself makes hipcc copy the
// whole block to scratch memory first.  Called from the __global__ wrappers with the total size of their parameters.
#ifndef SG_NO_KERNARG_WARM
#define SG_WARM_8(o) "s_load_dword %0, %1, " #o "+0x0\n s_load_dword %0, %1, " #o "+0x40\n s_load_dword %0, %1, " #o "+0x80\n s_load_dword %0, %1, " #o "+0xc0\n" \
                     "s_load_dword %0, %1, " #o "+0x100\n s_load_dword %0, %1, " #o "+0x140\n s_load_dword %0, %1, " #o "+0x180\n s_load_dword %0, %1, " #o "+0x1c0\n"
template <int BYTES>      // touches the first (BYTES rounded down to a multiple of 512, at most 4096) bytes of the kernel arguments
__device__ __forceinline__ void sg_warm_kernargs() {
    const void* k = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
    int scratch;
    constexpr int NB = BYTES / 512 > 8 ? 8 : BYTES / 512;
    if constexpr (NB == 8)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) SG_WARM_8(0x400) SG_WARM_8(0x600) SG_WARM_8(0x800) SG_WARM_8(0xa00) SG_WARM_8(0xc00) SG_WARM_8(0xe00) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 7)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) SG_WARM_8(0x400) SG_WARM_8(0x600) SG_WARM_8(0x800) SG_WARM_8(0xa00) SG_WARM_8(0xc00) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 6)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) SG_WARM_8(0x400) SG_WARM_8(0x600) SG_WARM_8(0x800) SG_WARM_8(0xa00) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 5)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) SG_WARM_8(0x400) SG_WARM_8(0x600) SG_WARM_8(0x800) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 4)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) SG_WARM_8(0x400) SG_WARM_8(0x600) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 3)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) SG_WARM_8(0x400) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 2)
        asm volatile(SG_WARM_8(0x0) SG_WARM_8(0x200) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
    else if constexpr (NB == 1)
        asm volatile(SG_WARM_8(0x0) "s_waitcnt lgkmcnt(0)" : "=&s"(scratch) : "s"(k) : "memory");
}
#else
template <int BYTES>
__device__ __forceinline__ void sg_warm_kernargs() {}
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// thread-local last-error string (sgan_last_error()).
extern thread_local char g_sgan_err[512];
int sgan_fail(int code, const char* fmt, ...);
extern thread_local const char* g_sgan_last_kernel;  // name of the kernel the last conv entry point launched

// Optional per-launch timing (sgan_profile_*): when enabled, every main conv kernel launch is bracketed by two
// HIP events recorded on the launch stream from inside the library (back to back with the launch, so a busy
// queue yields kernel-only durations comparable with rocprofv3 --kernel-trace).
void sg_prof_begin(hipStream_t st);
void sg_prof_end(hipStream_t st, const char* name);

#define SGAN_CHECK(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return sgan_fail(SGAN_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define SGAN_LAUNCH_CHECK()                                                                   \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) return sgan_fail(SGAN_ERR_HIP, "%s:%d: %s", __FILE__, __LINE__, \
                                               hipGetErrorString(e_));                        \
    } while (0)

// ------------------------------------------------------------------------------------------
// Tap / phase description shared by the implicit-GEMM forward, backward-data and backward-weight
// kernels.  A conv-like op is a set of <= 4 output "phases"; phase (oa, ob) covers output pixels
// (py*os + oa, px*os + ob), py < Hp, px < Wp, and each of its taps reads the gathered tensor at
// (py*is + dy, px*is + dx) and the weight slab at w_off.
//   Conv2d fwd / ConvT dgrad : 1 phase, k*k taps, is = stride, os = 1, dy = ky - pad
//   ConvT fwd / Conv dgrad   : stride^2 phases, (k/stride)^2 taps each, is = 1, os = stride
// ------------------------------------------------------------------------------------------
#define SGAN_F16_WEIGHT_SHIFT 10        // forward (fp16-plane) weight copy holds w * 2^10: |w| down to 2^-24 / 2^10 resolved, |w| < 64 representable
#define SGAN_BF16X3_MIN_PIXELS 256   // smaller maps always run the exact-fp32 MFMA kernels (see sg_igemm3_eligible)
#define SGAN_MAX_TAPS 49              // k <= 7 (the resnet generators' k7 layers, models/networks.py:234,260)
#define SGAN_MAX_PHASES 4

struct SgTap {
    int16_t dy, dx;
    int32_t w_off;  // element offset of this tap's [Cout][Cin] slab in the master weight
};

struct SgPhase {
    int32_t oa, ob;  // output phase offset
    int32_t Hp, Wp;  // phase grid
    int32_t ntaps;
    int32_t ktot;    // ntaps * (gathered channels)
    SgTap taps[SGAN_MAX_TAPS];
};

// what one half of a fused backward launch needs to know about the other (sgan_fused.hip)
struct SgFusePlan { int variant; int nblocks; int gx, gy, gz; size_t lds; bool pro; const char* name; int ks; };      // ks: split-K of the backward-data half (variant 3), 1 = none

struct SgNorm {  // device-side copy of sgan_norm_desc
    const double* stats;
    const float* gamma;
    const float* beta;
    int32_t count;
    float eps;
    int32_t act;
    float slope;
    int32_t sq_stride;  // 0 = C
    int32_t rep_stride; // 0 = one copy; else SGAN_STAT_REPLICAS copies this many doubles apart, summed on read
};

static inline SgNorm sg_norm_from(const sgan_norm_desc* d) {
    SgNorm n;
    if (d) {
        n.stats = d->stats; n.gamma = d->gamma; n.beta = d->beta;
        n.count = d->count; n.eps = d->eps; n.act = d->act; n.slope = d->slope; n.sq_stride = d->sq_stride;
        n.rep_stride = d->rep_stride;
    } else {
        n.stats = nullptr; n.gamma = nullptr; n.beta = nullptr;
        n.count = 1; n.eps = 0.f; n.act = SGAN_ACT_NONE; n.slope = 0.f; n.sq_stride = 0; n.rep_stride = 0;
    }
    return n;
}

// Build the phase tables.  `gather_*` is the tensor the taps read (forward input for fwd/wgrad,
// dout for dgrad); `grid_*` is the tensor the phases tile (forward output for fwd/wgrad, din for
// dgrad).  `transposed_access` selects the scatter form (ConvT fwd, Conv dgrad).
int sg_build_phases(const sgan_conv_desc* d, bool dgrad, SgPhase* ph, int* nphase, int* is, int* os);

// mean / rstd of channel c from accumulated (sum, sumsq) statistics (biased variance)
// a statistic kept in SGAN_STAT_REPLICAS copies `rep` doubles apart (0: one copy): the copies summed
__device__ __forceinline__ double sg_stat_sum(const double* base, int idx, int rep) {
    double v = base[idx];
    if (rep) {
#pragma unroll
        for (int r = 1; r < SGAN_STAT_REPLICAS; ++r) v += base[idx + (int64_t)r * rep];
    }
    return v;
}

// the copy workgroup `b` adds its partial statistics to
__device__ __forceinline__ double* sg_stat_replica(double* base, int rep, unsigned b) {
    return base + (int64_t)(rep ? (b & (SGAN_STAT_REPLICAS - 1)) : 0) * rep;
}

__device__ __forceinline__ void sg_mean_rstd(const SgNorm& n, int C, int c, float& mean, float& rstd) {
    double s = sg_stat_sum(n.stats, c, n.rep_stride), q = sg_stat_sum(n.stats, (n.sq_stride ? n.sq_stride : C) + c, n.rep_stride);
    double inv = 1.0 / (double)n.count;
    double m = s * inv;
    double var = q * inv - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)n.eps));
}

// fp16 planes: scale 2^s that brings a tensor of maximum magnitude `amax` under 2^15 (s = 14 - floor(log2 amax)); 0 / denormal -> 1
__device__ __forceinline__ int sg_f16_shift(float amax) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u);
    int s = e ? 141 - e : 0;
    return s > 100 ? 100 : (s < -100 ? -100 : s);
}
__device__ __forceinline__ float sg_pow2(int s) { return __builtin_bit_cast(float, (unsigned)(s + 127) << 23); }

__device__ __forceinline__ float sg_act(float y, int act, float slope) {
    if (act == SGAN_ACT_RELU) return y > 0.f ? y : 0.f;
    if (act == SGAN_ACT_LRELU) return y > 0.f ? y : y * slope;
    return y;
}

__device__ __forceinline__ float sg_act_grad(float y, int act, float slope) {
    if (act == SGAN_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == SGAN_ACT_LRELU) return y > 0.f ? 1.f : slope;
    return 1.f;
}
