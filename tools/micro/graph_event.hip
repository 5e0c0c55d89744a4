// Does hipEventRecordWithFlags(hipEventRecordExternal) work under stream capture on this ROCm?  (tools/micro)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, int n) { for (int i = 0; i < n; ++i) p[threadIdx.x] += 1.f; }
#define CK(x) do { hipError_t e = (x); printf("%-60s %s\n", #x, hipGetErrorString(e)); } while (0)
int main() {
    float* p; hipMalloc(&p, 1024);
    hipStream_t s; hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CK(hipEventRecordWithFlags(a, s, hipEventRecordExternal));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, s, p, 100000);
    CK(hipGetLastError());
    CK(hipEventRecordWithFlags(b, s, hipEventRecordExternal));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventSynchronize(b));
    float ms = -1; CK(hipEventElapsedTime(&ms, a, b));
    printf("elapsed %.3f ms\n", ms);
    return 0;
}
