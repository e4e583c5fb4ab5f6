// Do VALU instructions hide behind v_mfma_f32_16x16x32_bf16 (16 cycles each)?  Every wave runs NV independent VALU ops per MFMA.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3F80 + threadIdx.x % 3); b[e] = (short)(0x3F80 + threadIdx.x % 5); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 7 + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(v[(i + q) & 7]) : "v"(v[(i + q + 3) & 7]));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (float)v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int NV> void run(float* out, unsigned long long* ticks) {
    const int blocks = 256, iters = 20000;
    for (int waves = 8; waves >= 4; waves -= 4) {
        hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(64 * waves), 0, 0, out, ticks, 100);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(64 * waves), 0, 0, out, ticks, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        double tk = 0; for (int i = 0; i < blocks; ++i) tk += h[i]; tk /= blocks;
        printf("NV=%d VALU per MFMA, %d waves/SIMD: %.2f ticks per MFMA per SIMD; events: %.3f ms = %.0f TFLOP/s, tick rate %.2f GHz\n", NV, waves / 4,
               tk / ((double)iters * 8 * waves / 4), ms, (double)blocks * waves * iters * 8 * 16384.0 / ms / 1e9, tk / ms / 1e6);
    }
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 256 * 8);
    run<0>(out, ticks); run<1>(out, ticks); run<2>(out, ticks); run<3>(out, ticks); run<4>(out, ticks); run<6>(out, ticks);
    return 0;
}
