// Time-embedding path (unet6.py:18-34, 395-399, 350, 359): sinusoidal embedding -> Linear -> SiLU -> Linear -> SiLU ->
// the per-block projections, and its backward.  fp32 throughout; the row count is the BATCH (32 at cfg2, sample_num in
// the sampler), so these are weight-streaming contractions with 16-128 rows: 30 MFLOP that took 150 us of a 4.3 ms step
// as eleven generic launches (64x64-tiled kernels give 8 workgroups for a [32,512]x[512,512] layer; the VALU skinny
// kernel re-read the parked activations from LDS once per thread).
//
// One shape of kernel, two operand orders.  A workgroup owns a strip of 16 NB output columns and all rows; its 4 waves
// split the reduction range, every wave runs v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) on
// fragments loaded STRAIGHT from global memory into registers -- a lane's float4 of 4 consecutive k feeds 4 MFMAs
// (lane group g supplies k = 4 g + j to MFMA j, the same permutation on both operands) -- and the 4 partial tiles meet
// in LDS once.  No LDS staging of operands: every fragment is used by exactly one wave.
//   skinny_nt:  Y[M][N] = X[M][K] W[N][K]^T + b     (+ optional second output silu(Y); X may be the sinusoidal
//               embedding of t, generated in registers and stored once by workgroup 0)
//   skinny_nn:  dX[M][N] = dY[M][K] W[K][N]         (x silu'(pre) when unsplit; split over workgroups into fp32 slabs
//               that silu_bwd_sum adds while it applies silu')
#include "common.h"

namespace mdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef MDM_STAMP
__device__ unsigned long long g_tstamp_buf[1024 * 8];
__device__ __forceinline__ unsigned long long tstamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MDM_TT(...) __VA_ARGS__
#else
#define MDM_TT(...)
#endif
__device__ __forceinline__ float silu_grad(float v) {
    const float s = 1.f / (1.f + expf(-v));
    return s * (1.f + v * (1.f - s));
}

// element k of the embedding of timestep tv: flip = 0, shift = 1: unet6.py:18-34 ([sin | cos], exponent / (half - 1));
// flip = 1, shift = 0: diffusers' Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0) -- same arithmetic as temb_kernel
__device__ __forceinline__ float temb_elem(float tv, int k, int half, int flip, float shift) {
    const int j = k < half ? k : k - half;
    const float a = tv * expf(-(float)j * (logf(10000.f) / ((float)half - shift)));
    return ((k < half) != (flip != 0)) ? sinf(a) : cosf(a);
}

// the 4 partial accumulator tiles of a workgroup -> wave 0 (waves 1..3 park theirs in LDS)
template <int MB, int NB>
__device__ __forceinline__ void meet_in_wave0(f32x4 (&acc)[MB][NB], float* red, int wave, int lane) {
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                *reinterpret_cast<f32x4*>(red + ((((wave - 1) * MB + i) * NB + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] += *reinterpret_cast<const f32x4*>(red + (((w * MB + i) * NB + j) * 64 + lane) * 4);
    }
}

template <int MB, int NB, bool TEMB>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ t, int flip,
                                                        float shift, float* __restrict__ emb_out, const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ bias, int M, int N, int K, float* __restrict__ y, int ldy,
                                                        float* __restrict__ act_out) {
    __shared__ __attribute__((aligned(16))) float red[3 * MB * NB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16 * NB, kq = K >> 2, kbeg = wave * kq;
    const int r16 = lane & 15, kg = 4 * (lane >> 4);
    MDM_TT(const unsigned long long ts0 = tstamp_now(); unsigned long long ts1 = 0, ts2 = 0;)
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tv[MB];
    if (TEMB) {
#pragma unroll
        for (int i = 0; i < MB; ++i) tv[i] = (i * 16 + r16 < M) ? t[i * 16 + r16] : 0.f;
    }
    // DEPTH 16-wide k-chunks are loaded before the first of them is multiplied: the wave's whole range is in flight at once
    // (a [32,512] x [512,512] layer has 32 workgroups -- nothing else hides the latency)
    constexpr int DEPTH = MB <= 2 ? 8 : 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = kbeg; kb < kbeg + kq; kb += 16 * DEPTH) {
        float4 a[DEPTH][MB], b[DEPTH][NB];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int k = kb + 16 * u + kg;
            const bool live = kb + 16 * u < kbeg + kq;
            // (loads from a CLAMPED address, the value zeroed afterwards: `cond ? *p : zero4` made hipcc select between the global
            // pointer and a private zero constant, i.e. 4 flat_load_dword per float4 with a scratch store in front -- the load phase of
            // these kernels was 24 000 - 33 000 cycles of a 13 - 18 us launch, scripts/stamp_temb.py)
            const int ks = live ? k : kbeg + kg;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int n = n0 + j * 16 + r16;
                const float4 w = *reinterpret_cast<const float4*>(W + (int64_t)(n < N ? n : N - 1) * ldw + ks);
                const bool ok = live && n < N;
                b[u][j] = make_float4(ok ? w.x : 0.f, ok ? w.y : 0.f, ok ? w.z : 0.f, ok ? w.w : 0.f);
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int m = i * 16 + r16;
                if (TEMB) {
                    const int half = K >> 1;
                    a[u][i] = zero4;
                    if (live && m < M) {
                        a[u][i] = make_float4(temb_elem(tv[i], k, half, flip, shift), temb_elem(tv[i], k + 1, half, flip, shift),
                                              temb_elem(tv[i], k + 2, half, flip, shift), temb_elem(tv[i], k + 3, half, flip, shift));
                        if (emb_out && blockIdx.x == 0) *reinterpret_cast<float4*>(emb_out + (int64_t)m * K + k) = a[u][i];
                    }
                } else {
                    const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(m < M ? m : M - 1) * ldx + ks);
                    const bool ok = live && m < M;
                    a[u][i] = make_float4(ok ? xv.x : 0.f, ok ? xv.y : 0.f, ok ? xv.z : 0.f, ok ? xv.w : 0.f);
                }
            }
        }
        MDM_TT(ts1 = tstamp_now();)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    // operands swapped as in gemm_f32_mfma_kernel: the accumulator holds D[m = lane & 15][n = 4 (lane >> 4) + reg]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].x, a[u][i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].y, a[u][i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].z, a[u][i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].w, a[u][i].w, acc[i][j], 0, 0, 0);
                }
    }
    MDM_TT(ts2 = tstamp_now();)
    meet_in_wave0<MB, NB>(acc, red, wave, lane);
    if (wave != 0) return;
    MDM_TT(const unsigned long long ts3 = tstamp_now();
           if (lane == 0 && blockIdx.x < 1024) { unsigned long long* r = g_tstamp_buf + blockIdx.x * 8; r[0] = 1; r[1] = ts1 - ts0; r[2] = ts2 - ts1; r[3] = ts3 - ts2; r[4] = ts0; r[5] = ts3; })
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = i * 16 + r16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + j * 16 + kg;
            if (n >= N) continue;               // N % 4 == 0: a float4 is inside or outside
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (bias) { const float4 bb = *reinterpret_cast<const float4*>(bias + n); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
            *reinterpret_cast<float4*>(y + (int64_t)m * ldy + n) = v;
            if (act_out) {
                const float4 s = make_float4(v.x / (1.f + expf(-v.x)), v.y / (1.f + expf(-v.y)), v.z / (1.f + expf(-v.z)),
                                             v.w / (1.f + expf(-v.w)));
                *reinterpret_cast<float4*>(act_out + (int64_t)m * ldy + n) = s;
            }
        }
    }
}

// grid = (N / (16 NB), splits); workgroup (strip, s) reduces k in [s K / splits, (s + 1) K / splits)
template <int MB, int NB>
__global__ __launch_bounds__(256) void skinny_nn_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ W, int ldw, int M,
                                                        int N, int K, const float* __restrict__ pre, float* __restrict__ dx,
                                                        float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float red[3 * MB * NB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16 * NB, splits = gridDim.y, ks = K / splits, kq = ks >> 2, kbeg = blockIdx.y * ks + wave * kq;
    const int r16 = lane & 15, kg = 4 * (lane >> 4);
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int DEPTH = MB <= 2 ? 8 : 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = kbeg; kb < kbeg + kq; kb += 16 * DEPTH) {
        float4 a[DEPTH][MB], b[DEPTH][NB];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int k = kb + 16 * u + kg;
            const bool live = kb + 16 * u < kbeg + kq;
            const int kc = live ? k : kbeg + kg;    // (clamped addresses, values zeroed afterwards: see skinny_nt_kernel)
#pragma unroll
            for (int j = 0; j < NB; ++j) {          // W is [K][N]: 16 lanes read 64 contiguous bytes of each of 4 k-rows
                const int n = n0 + j * 16 + r16;
                const float* p = W + (int64_t)kc * ldw + (n < N ? n : N - 1);
                const float w0 = p[0], w1 = p[ldw], w2 = p[2 * (int64_t)ldw], w3 = p[3 * (int64_t)ldw];
                const bool ok = live && n < N;
                b[u][j] = make_float4(ok ? w0 : 0.f, ok ? w1 : 0.f, ok ? w2 : 0.f, ok ? w3 : 0.f);
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int m = i * 16 + r16;
                const float4 dv = *reinterpret_cast<const float4*>(dy + (int64_t)(m < M ? m : M - 1) * lddy + kc);
                const bool ok = live && m < M;
                a[u][i] = make_float4(ok ? dv.x : 0.f, ok ? dv.y : 0.f, ok ? dv.z : 0.f, ok ? dv.w : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].x, a[u][i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].y, a[u][i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].z, a[u][i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][j].w, a[u][i].w, acc[i][j], 0, 0, 0);
                }
    }
    meet_in_wave0<MB, NB>(acc, red, wave, lane);
    if (wave != 0) return;
    float* out = splits > 1 ? slabs + (int64_t)blockIdx.y * M * N : dx;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = i * 16 + r16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + j * 16 + kg;
            if (n >= N) continue;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (splits == 1 && pre) {
                const float4 p = *reinterpret_cast<const float4*>(pre + (int64_t)m * N + n);
                v.x *= silu_grad(p.x); v.y *= silu_grad(p.y); v.z *= silu_grad(p.z); v.w *= silu_grad(p.w);
            }
            *reinterpret_cast<float4*>(out + (int64_t)m * N + n) = v;
        }
    }
}

__global__ void silu_bwd_sum_kernel(const float* __restrict__ pre, const float* __restrict__ slabs, int nslab, int64_t n4,
                                    float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4*>(slabs)[i];
    for (int q = 1; q < nslab; ++q) {
        const float4 v = reinterpret_cast<const float4*>(slabs)[(int64_t)q * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (pre) {
        const float4 p = reinterpret_cast<const float4*>(pre)[i];
        s.x *= silu_grad(p.x); s.y *= silu_grad(p.y); s.z *= silu_grad(p.z); s.w *= silu_grad(p.w);
    }
    reinterpret_cast<float4*>(dx)[i] = s;
}

}  // namespace mdm

using namespace mdm;

extern "C" int mdm_skinny_supported(int M, int N, int K, int splits) {
    return (M >= 1 && M <= 128 && N >= 16 && N % 16 == 0 && K >= 64 && splits >= 1 && K % (64 * splits) == 0) ? 1 : 0;
}

extern "C" int mdm_skinny_linear_fwd(const float* x, int ldx, const float* t, int flip_sin_to_cos, float freq_shift, float* emb_out,
                                     const float* W, int ldw, const float* bias, int M, int N, int K, float* y, int ldy, float* act_out,
                                     void* stream) {
    MDM_REQUIRE(mdm_skinny_supported(M, N, K, 1), "skinny_linear_fwd: unsupported shape M=%d N=%d K=%d", M, N, K);
    MDM_REQUIRE((x != nullptr) != (t != nullptr), "skinny_linear_fwd: give the input matrix OR the timesteps");
    MDM_REQUIRE(W && y && ldw % 4 == 0 && ldy % 4 == 0 && (!x || ldx % 4 == 0), "skinny_linear_fwd: bad pointers / pitches");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (N >= 2048 && N % 32 == 0) ? 2 : 1;          // wide layers: two column blocks per workgroup halve the re-reads of X
    dim3 grid((unsigned)(N / (16 * nb)));
#define MDM_NT(MB, NB, TE) hipLaunchKernelGGL((skinny_nt_kernel<MB, NB, TE>), grid, dim3(256), 0, s, x, ldx, t, flip_sin_to_cos, freq_shift, \
                                              emb_out, W, ldw, bias, M, N, K, y, ldy, act_out)
    if (t) { if (M <= 32) MDM_NT(2, 1, true); else MDM_NT(8, 1, true); }
    else if (nb == 2) { if (M <= 32) MDM_NT(2, 2, false); else MDM_NT(8, 2, false); }
    else { if (M <= 32) MDM_NT(2, 1, false); else MDM_NT(8, 1, false); }
#undef MDM_NT
    return launch_status("skinny_linear_fwd");
}

extern "C" int mdm_skinny_linear_bwd(const float* dy, int lddy, const float* W, int ldw, int M, int N, int K, int splits, const float* pre,
                                     float* dx, float* slabs, void* stream) {
    MDM_REQUIRE(mdm_skinny_supported(M, N, K, splits), "skinny_linear_bwd: unsupported shape M=%d N=%d K=%d splits=%d", M, N, K, splits);
    MDM_REQUIRE(dy && W && lddy % 4 == 0 && (splits > 1 ? slabs != nullptr : dx != nullptr), "skinny_linear_bwd: bad pointers / pitches");
    dim3 grid((unsigned)(N / 16), (unsigned)splits);
    if (M <= 32) hipLaunchKernelGGL((skinny_nn_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, dy, lddy, W, ldw, M, N, K, pre, dx, slabs);
    else hipLaunchKernelGGL((skinny_nn_kernel<8, 1>), grid, dim3(256), 0, (hipStream_t)stream, dy, lddy, W, ldw, M, N, K, pre, dx, slabs);
    return launch_status("skinny_linear_bwd");
}

extern "C" int mdm_silu_bwd_sum(const float* pre, const float* slabs, int nslab, int64_t n, float* dx, void* stream) {
    MDM_REQUIRE(slabs && dx && nslab >= 1 && n % 4 == 0, "silu_bwd_sum: bad arguments");
    hipLaunchKernelGGL(silu_bwd_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, slabs, nslab, n / 4, dx);
    return launch_status("silu_bwd_sum");
}

#ifdef MDM_STAMP
extern "C" int mdm_debug_stamps_temb(unsigned long long* out, int reset) {      // out: 1024 * 8 entries
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdm::g_tstamp_buf), 1024 * 8 * 8) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mdm::g_tstamp_buf)) != hipSuccess || hipMemset(p, 0, 1024 * 8 * 8) != hipSuccess) return -1;
    }
    return 0;
}
#endif
