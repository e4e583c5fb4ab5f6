// How fast is v_mfma_f32_16x16x32_bf16 really, and what does s_memtime count?  256 workgroups x 8 waves, every wave runs a chain-free
// stream of MFMAs on 8 accumulators.  Prints: TFLOP/s by HIP events, s_memtime ticks per MFMA per SIMD (2 waves share a SIMD), tick rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3F80 + threadIdx.x % 3); b[e] = (short)(0x3F80 + threadIdx.x % 5); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    const int blocks = 256, iters = 20000;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 8; waves >= 4; waves -= 4) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, out, ticks, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, out, ticks, iters); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        double tk = 0; for (int i = 0; i < blocks; ++i) tk += h[i]; tk /= blocks;
        const double mfma = (double)blocks * waves * iters * 8, flops = mfma * 16384.0;
        printf("%d waves/WG: %.3f ms, %.1f TFLOP/s; s_memtime: %.0f ticks per kernel = %.2f GHz tick rate; ticks per MFMA per SIMD: %.2f\n", waves, ms,
               flops / ms / 1e9, tk, tk / ms / 1e6, tk / ((double)iters * 8 * waves / 4));
    }
    return 0;
}
