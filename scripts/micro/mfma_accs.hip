// MFMA rate against the number of accumulator tiles and operand registers in play (the nine-tap weight-gradient tile holds 36 tiles).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_accs.hip -o /tmp/mfma_accs && /tmp/mfma_accs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int NA, int NB>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a[NA], b[NB];
    for (int j = 0; j < NA; ++j) for (int e = 0; e < 8; ++e) a[j][e] = (short)(0x3F80 + (threadIdx.x + j) % 3);
    for (int j = 0; j < NB; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (short)(0x3F80 + (threadIdx.x + j) % 5);
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i / NB) % NA]), "v"(b[i % NB]));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int NA, int NB> void run(float* out) {
    const int blocks = 256, iters = 40000 / NACC * 8;
    for (int waves = 8; waves >= 4; waves -= 4) {
        hipLaunchKernelGGL((k<NACC, NA, NB>), dim3(blocks), dim3(64 * waves), 0, 0, out, 10);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, NA, NB>), dim3(blocks), dim3(64 * waves), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%2d accumulator tiles, %d x %d operand fragments, %d waves/SIMD: %.0f TFLOP/s\n", NACC, NA, NB, waves / 4,
               (double)blocks * waves * iters * NACC * 16384.0 / ms / 1e9);
    }
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<8, 1, 1>(out); run<16, 4, 4>(out); run<36, 9, 4>(out); run<36, 3, 4>(out); run<48, 12, 4>(out);
    return 0;
}
