// Per-kernel cost of back-to-back small kernels: stream-ordered launches vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
    float* buf; int n = 1 << 18; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int K = 2000;
    for (int blocks : {1, 256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < K; ++i) tiny<<<blocks, 256, 0, s>>>(buf, n);
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("stream  blocks=%4d: %.2f us/kernel\n", blocks, ms * 1e3 / K);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; ++i) tiny<<<blocks, 256, 0, s>>>(buf, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("graph   blocks=%4d: %.2f us/kernel\n", blocks, ms * 1e3 / K);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
