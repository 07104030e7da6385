#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int v[640]; };   // 2560 B
struct Big2 { int v[512]; };  // 2048 B
__global__ void k(Big a, Big2 b, int* out) { if (threadIdx.x == 0) out[blockIdx.x] = a.v[639] + b.v[511] + a.v[0]; }
int main() {
    Big a; Big2 b; for (int i = 0; i < 640; ++i) a.v[i] = i; for (int i = 0; i < 512; ++i) b.v[i] = 2 * i;
    int* d; hipMalloc(&d, 16); 
    hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, a, b, d);
    hipError_t e = hipDeviceSynchronize();
    int h[2] = {0, 0}; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("err=%d out=%d %d (expect %d)\n", (int)e, h[0], h[1], 639 + 1022 + 0);
    // graph capture as well
    hipStream_t s; hipStreamCreate(&s); hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    a.v[0] = 5;
    hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, s, a, b, d);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); e = hipStreamSynchronize(s);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("graph err=%d out=%d (expect %d)\n", (int)e, h[0], 639 + 1022 + 5);
    return 0;
}
