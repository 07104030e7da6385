// Stand-alone host for the C ABI of libsgan_hip.so (include/sgan_hip.h): no Python, no torch -- what a C/C++/cgo/JNI
// caller does.  Runs the discriminator's second conv (Conv2d 32 -> 64, k4 s2 p2, LeakyReLU(0.2) read on load) forward,
// backward-data and backward-weight on a 33 x 41 image and checks all three against plain loops on the host.
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/c_abi_conv.cpp -Lsupervised-gan_amd/csrc -lsgan_hip \
//         -Wl,-rpath,$PWD/supervised-gan_amd/csrc -o /tmp/c_abi_conv && /tmp/c_abi_conv
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sgan_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SGAN_OK_(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, sgan_last_error()); return 3; } } while (0)

static float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

int main() {
    const int H = 33, W = 41, Ci = 32, Co = 64, k = 4, s = 2, p = 2;
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    std::vector<float> x((size_t)H * W * Ci), w((size_t)k * k * Co * Ci), b(Co), r((size_t)Ho * Wo * Co);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 32768.f - 1.f; };
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = 0.05f * rnd();      // master layout [kh*kw][Cout][Cin]
    for (auto& v : b) v = 0.1f * rnd();
    for (auto& v : r) v = rnd();              // upstream gradient

    // ---- host reference ----
    std::vector<float> y((size_t)Ho * Wo * Co, 0.f), dx((size_t)H * W * Ci, 0.f), dw(w.size(), 0.f), db(Co, 0.f);
    for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox)
            for (int co = 0; co < Co; ++co) {
                double acc = b[co];
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const int iy = oy * s + ky - p, ix = ox * s + kx - p;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        for (int ci = 0; ci < Ci; ++ci)
                            acc += (double)lrelu(x[((size_t)iy * W + ix) * Ci + ci]) * w[((size_t)(ky * k + kx) * Co + co) * Ci + ci];
                    }
                y[((size_t)oy * Wo + ox) * Co + co] = (float)acc;
            }
    for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox)
            for (int co = 0; co < Co; ++co) {
                const float g = r[((size_t)oy * Wo + ox) * Co + co];
                db[co] += g;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const int iy = oy * s + ky - p, ix = ox * s + kx - p;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        for (int ci = 0; ci < Ci; ++ci) {
                            const size_t xi = ((size_t)iy * W + ix) * Ci + ci, wi = ((size_t)(ky * k + kx) * Co + co) * Ci + ci;
                            dx[xi] += g * w[wi] * (x[xi] > 0.f ? 1.f : 0.2f);      // through the LeakyReLU the conv read
                            dw[wi] += g * lrelu(x[xi]);
                        }
                    }
            }

    // ---- device ----
    float *dX, *dW, *dB, *dY, *dR, *dDX, *dDW, *dDB;
    HIP_OK(hipMalloc(&dX, x.size() * 4)); HIP_OK(hipMalloc(&dW, w.size() * 4)); HIP_OK(hipMalloc(&dB, b.size() * 4));
    HIP_OK(hipMalloc(&dY, y.size() * 4)); HIP_OK(hipMalloc(&dR, r.size() * 4)); HIP_OK(hipMalloc(&dDX, x.size() * 4));
    HIP_OK(hipMalloc(&dDW, w.size() * 4)); HIP_OK(hipMalloc(&dDB, b.size() * 4));
    HIP_OK(hipMemcpy(dX, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dR, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dDW, 0, w.size() * 4)); HIP_OK(hipMemset(dDB, 0, b.size() * 4));     // backward-weight accumulates

    sgan_conv_desc d = {SGAN_CONV, k, s, p, H, W, Ci, Ho, Wo, Co, 0, 0};
    sgan_norm_desc act = {nullptr, nullptr, nullptr, 1, 0.f, SGAN_ACT_LRELU, 0.2f, 0};      // no norm, LeakyReLU(0.2) on load
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const int64_t ws_kib = sgan_conv_fwd(&d, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, -1, nullptr);
    void* ws = nullptr;
    if (ws_kib > 0) HIP_OK(hipMalloc(&ws, (size_t)ws_kib * 1024));
    SGAN_OK_(sgan_conv_fwd(&d, dX, Ci, &act, dW, dB, dY, Co, SGAN_ACT_NONE, nullptr, ws, ws_kib * 1024, st));
    SGAN_OK_(sgan_conv_dgrad(&d, dR, Co, dW, dDX, Ci, dX, Ci, &act, nullptr, ws, ws_kib * 1024, st));
    SGAN_OK_(sgan_conv_wgrad(&d, dX, Ci, &act, dR, Co, dDW, dDB, nullptr, 0, st));
    HIP_OK(hipStreamSynchronize(st));

    auto check = [](const char* what, const float* dev, const std::vector<float>& ref) {
        std::vector<float> got(ref.size());
        if (hipMemcpy(got.data(), dev, ref.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1.0;
        double mx = 0, err = 0;
        for (size_t i = 0; i < ref.size(); ++i) { mx = std::fmax(mx, std::fabs(ref[i])); err = std::fmax(err, std::fabs(got[i] - ref[i])); }
        std::printf("%-16s max|err| / max|ref| = %.3e\n", what, err / mx);
        return err / mx;
    };
    double worst = 0;
    worst = std::fmax(worst, check("forward", dY, y));
    worst = std::fmax(worst, check("backward-data", dDX, dx));
    worst = std::fmax(worst, check("backward-weight", dDW, dw));
    worst = std::fmax(worst, check("backward-bias", dDB, db));
    std::printf("%s -- %s\n", sgan_version(), worst < 1e-4 ? "OK" : "MISMATCH");
    return worst < 1e-4 ? 0 : 1;
}
