// Optimizer pass over flat fp32 buffers: global grad-norm, clip, AdamW, EMA and the
// bf16 weight shadow in one read-modify-write sweep (HBM-bound: 16 B read + 12..18 B
// written per parameter).
//
// Replaces accelerator.clip_grad_norm_ + torch.optim.AdamW.step + diffusers EMAModel.step
// (reference trainer_masked_mean_shift.py:163-172, main_train_masked.py:116-141).
#include "common.h"

namespace mdm {

// Deterministic two-stage reduction: data-parallel replicas must derive the SAME clip coefficient from the
// same (all-reduced) gradient, bit for bit, or they drift apart -- so no float atomics here.
constexpr int SQN_BLOCKS = 1024;
__device__ float g_sqnorm_partials[SQN_BLOCKS];

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, int64_t n4, int64_t n) {
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(g)[i];
        a = fmaf(v.x, v.x, a); a = fmaf(v.y, v.y, a); a = fmaf(v.z, v.z, a); a = fmaf(v.w, v.w, a);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) a = fmaf(g[i], g[i], a);
    a = wave_sum(a);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) g_sqnorm_partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(float* out, int nblk) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += g_sqnorm_partials[i];
    a = wave_sum(a);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) *out = (part[0] + part[1]) + (part[2] + part[3]);      // stored: no zeroing launch in front
}

// hp: lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, ema_decay
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, float* ema, bf16_t* shadow,
                                                    int64_t n, const float* hp, const float* sqnorm, float max_norm, float gmul) {
    const float lr = hp[0], b1 = hp[1], b2 = hp[2], eps = hp[3], wd = hp[4], bc1 = hp[5], bc2 = hp[6], ed = hp[7];
    float coef = gmul;
    if (max_norm > 0.f) {
        float norm = sqrtf(*sqnorm) * gmul;
        float c = max_norm / (norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_
        if (c < 1.f) coef *= c;
    }
    const float step = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * wd);            // decoupled weight decay first (torch AdamW)
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        pi -= step * mi / denom;
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema) { float e = ema[i]; ema[i] = e - (1.f - ed) * (e - pi); }
        if (shadow) shadow[i] = f2bf(pi);
    }
}

// Transposed bf16 shadow of the conv filters: bf16 [tap][Cout][Cin] (written by the optimizer kernel) -> bf16 [tap][Cin][Cout],
// so the data-gradient contraction reads its weights k-contiguous like the forward does.  One workgroup per 64 x 64 tile of
// one tap of one layer, 16-byte loads and stores on both sides (Cout, Cin are multiples of 8);
// `tiles` = {element offset of the tap matrix, Cout, Cin, row0, col0} x ntiles.
// (The first version moved 2 bytes per lane on 32 x 32 tiles: 62 us for 143 MB; this one is bound by the bytes.)
__global__ __launch_bounds__(256) void transpose_shadow_bf16_kernel(const bf16_t* Pb, bf16_t* PT, const int64_t* tiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
    const int64_t* e = tiles + (int64_t)blockIdx.x * 5;
    const int64_t off = e[0];
    const int Cout = (int)e[1], Cin = (int)e[2], r0 = (int)e[3], c0 = (int)e[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, row = q >> 3, cc = (q & 7) * 8;
        const int r = r0 + row, c = c0 + cc;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < Cout && c < Cin) v = *reinterpret_cast<const uint4*>(Pb + off + (int64_t)r * Cin + c);
        *reinterpret_cast<uint4*>(&tile[row][cc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, col = q >> 3, rr = (q & 7) * 8;
        const int c = c0 + col, r = r0 + rr;
        if (c < Cin && r < Cout) {                                   // Cout % 8 == 0: the 8 rows are inside or outside together
            uint4 v;
            v.x = (uint32_t)tile[rr + 0][col] | ((uint32_t)tile[rr + 1][col] << 16);
            v.y = (uint32_t)tile[rr + 2][col] | ((uint32_t)tile[rr + 3][col] << 16);
            v.z = (uint32_t)tile[rr + 4][col] | ((uint32_t)tile[rr + 5][col] << 16);
            v.w = (uint32_t)tile[rr + 6][col] | ((uint32_t)tile[rr + 7][col] << 16);
            *reinterpret_cast<uint4*>(PT + off + (int64_t)c * Cout + r) = v;
        }
    }
}

// The split shadow of fp32 filters (mdm_gemm_desc.B_split; conv_halo_body<..., SPLIT>): per 32-element block, chunk g = 8 bf16 hi
// halves of elements {4g..4g+3, 16+4g..16+4g+3} -- dword k = (element 4g+k, element 16+4g+k) --, chunk 4+g = their lo halves in the same
// order.  One thread per (block, g): two 16-byte loads, two stores.
// segs[i] = {element offset, length}; blockIdx.y = segment.
__global__ __launch_bounds__(256) void split_shadow_kernel(const float* P, float* Ps, const int64_t* segs) {
    const int64_t off = segs[2 * blockIdx.y], len = segs[2 * blockIdx.y + 1];
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < len / 8; id += (int64_t)gridDim.x * 256) {
        const int64_t base = off + (id >> 2) * 32 + (id & 3) * 4;
        const float4 a = *reinterpret_cast<const float4*>(P + base), b = *reinterpret_cast<const float4*>(P + base + 16);
        const float xa[4] = {a.x, a.y, a.z, a.w}, xb[4] = {b.x, b.y, b.z, b.w};
        uint32_t h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {            // dword k: element k of the first chunk (low half) | element k of the second (high half)
            const bf16_t h0 = f2bf(xa[k]), h1 = f2bf(xb[k]);
            h[k] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            l[k] = (uint32_t)f2bf(xa[k] - bf2f(h0)) | ((uint32_t)f2bf(xb[k] - bf2f(h1)) << 16);
        }
        *reinterpret_cast<uint4*>(Ps + base) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(Ps + base + 16) = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

__global__ void cast_bf16_kernel(const float* src, bf16_t* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}
__global__ void fill_kernel(float* p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// segs[i] = {element offset, length}; lengths are multiples of 4 and <= 4096, offsets 16-byte aligned: one workgroup each
__global__ __launch_bounds__(256) void fill_segments_kernel(float* base, const int64_t* segs, float v) {
    const int64_t off = segs[2 * blockIdx.x], len = segs[2 * blockIdx.x + 1];
    const float4 v4 = make_float4(v, v, v, v);
    for (int64_t i = 4 * (int64_t)threadIdx.x; i < len; i += 1024) *reinterpret_cast<float4*>(base + off + i) = v4;
}
static inline int ogrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace mdm
using namespace mdm;

extern "C" int mdm_sqnorm(const float* g, int64_t n, float* out, void* stream) {
    MDM_REQUIRE(g && out && n > 0, "sqnorm: bad arguments");
    MDM_REQUIRE(((uintptr_t)g & 15) == 0, "sqnorm: buffer must be 16-byte aligned");
    int nb = ogrid(n / 4);
    if (nb > SQN_BLOCKS) nb = SQN_BLOCKS;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n / 4, n);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, nb);
    return launch_status("sqnorm");
}
extern "C" int mdm_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t n,
                             const float* hp, const float* sqnorm, float max_norm, float gmul, void* stream) {
    MDM_REQUIRE(p && g && m && v && hp && n > 0, "adamw: bad arguments");
    MDM_REQUIRE(max_norm <= 0.f || sqnorm, "adamw: clipping needs the squared norm");
    hipLaunchKernelGGL(adamw_kernel, dim3(ogrid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16_t*)shadow_bf16, n, hp,
                       sqnorm, max_norm, gmul);
    return launch_status("adamw");
}
extern "C" int mdm_transpose_shadow_bf16(const void* Pb, void* PT, const int64_t* tiles, int ntiles, void* stream) {
    MDM_REQUIRE(Pb && PT && tiles && ntiles > 0, "transpose_shadow_bf16: bad arguments");
    hipLaunchKernelGGL(transpose_shadow_bf16_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Pb, (bf16_t*)PT, tiles, ntiles);
    return launch_status("transpose_shadow_bf16");
}
extern "C" int mdm_split_shadow(const float* P, float* Ps, const int64_t* segs, int nseg, void* stream) {
    MDM_REQUIRE(P && Ps && segs && nseg > 0 && nseg <= 65535, "split_shadow: bad arguments");
    MDM_REQUIRE((((uintptr_t)P | (uintptr_t)Ps) & 15) == 0, "split_shadow: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(split_shadow_kernel, dim3(64, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, P, Ps, segs);
    return launch_status("split_shadow");
}
extern "C" int mdm_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ogrid(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return launch_status("cast_bf16");
}
extern "C" int mdm_fill_segments_f32(float* base, const int64_t* segs, int nseg, float v, void* stream) {
    MDM_REQUIRE(base && segs && nseg > 0, "fill_segments: bad arguments");
    hipLaunchKernelGGL(fill_segments_kernel, dim3((unsigned)nseg), dim3(256), 0, (hipStream_t)stream, base, segs, v);
    return launch_status("fill_segments");
}
extern "C" int mdm_fill_f32(float* p, float v, int64_t n, void* stream) {
    hipLaunchKernelGGL(fill_kernel, dim3(ogrid(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
    return launch_status("fill");
}
