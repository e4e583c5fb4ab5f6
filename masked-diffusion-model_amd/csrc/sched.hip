// Scheduler / loss / sampler kernels: the forward degradation (mask draw + fill), the
// time-scaled shift, the fused x0-space MSE with its gradient, and the cold-diffusion
// reverse update.  NCHW fp32 like the reference's tensors; pure HBM streaming.
//
// Replaces reference scheduler.py:286-323, 438-477, 572-598 (degrade), :612-732, 757-777
// (shift), trainer_masked_mean_shift.py:142-159 (loss), sampler.py:146-152, 199-216.
#include "common.h"

namespace mdm {

// Philox stream ids so the draws of one step never overlap
// (id 0: timesteps, 1: training mask, 2: shift, 3/4: sampler masks t / t-1)
__device__ __forceinline__ uint4 philox_at(const uint64_t* rng, int stream_id, uint64_t idx) {
    Philox ph(rng[0]);
    return ph(idx, rng[1] * 8 + (uint64_t)stream_id);
}

__global__ void draw_timesteps_kernel(const uint64_t* rng, const int32_t* used, int n_used, const double* table,
                                      const float* wtab, int N, float* t_out, double* amount_out, float* weight_out,
                                      int32_t* idx_out, const double* table2, double* out2, long long* zero_out) {
    if (zero_out && blockIdx.x == 0 && threadIdx.x == 0) { zero_out[0] = 0; zero_out[1] = 0; }   // the step's loss accumulator (mdm_loss_fwd_bwd adds)
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    uint4 r = philox_at(rng, 0, (uint64_t)n);
    int idx = (int)(((uint64_t)r.x * (uint64_t)n_used) >> 32);     // uniform in [0, n_used)
    int t = used[idx];
    if (t_out) t_out[n] = (float)t;
    if (amount_out) amount_out[n] = table[t - 1];
    if (weight_out) weight_out[n] = wtab ? wtab[idx] : 1.f;
    if (idx_out) idx_out[n] = idx;
    if (out2) out2[n] = table2[t - 1];
}

// One workgroup per image.  Pass A (only for data-dependent fills): masked sums per channel.
// Pass B: write x_t / mask.  The mask value is recomputed in pass B from the same uniform.
__global__ __launch_bounds__(256) void degrade_kernel(const float* x0, const float* u, const float* mask_in,
                                                      const double* amount, int amount_stride, const uint64_t* rng,
                                                      int rng_stream, int C, int HW, int Cm, int fill_mode, float fill_const,
                                                      float* x_t, float* mask, float* mean_pixel, int keep_lds) {
    const int img = blockIdx.x, t = threadIdx.x;
    const double thr = amount ? amount[(int64_t)img * amount_stride] : 0.0;
    __shared__ float s_sum[8], s_cnt[8], s_fill[8], s_wsum[4][8], s_wcnt[4][8];
    // The keep flags of the image, drawn ONCE into LDS (keep_lds != 0: Cm * HW bytes fit and HW % 4 == 0): one Philox call
    // yields the uniforms of 4 consecutive elements, and both passes and all C channels of a 1-channel mask read the same flag
    // (computing it at every use cost 24 Philox calls per thread at cfg2).  Same uniforms, same fp64 comparison.
    extern __shared__ unsigned char s_keep[];
    if (keep_lds && !mask_in) {
        for (int q = t; q < (Cm * HW) >> 2; q += 256) {
            const int64_t e0 = (int64_t)img * Cm * HW + 4 * q;
            float uv[4];
            if (u) { uv[0] = u[e0]; uv[1] = u[e0 + 1]; uv[2] = u[e0 + 2]; uv[3] = u[e0 + 3]; }
            else {
                const uint4 r = philox_at(rng, rng_stream, (uint64_t)e0 >> 2);
                uv[0] = u01(r.x); uv[1] = u01(r.y); uv[2] = u01(r.z); uv[3] = u01(r.w);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s_keep[4 * q + j] = ((double)uv[j] > thr) ? 1 : 0;
        }
        __syncthreads();
    }
    auto keep_at = [&](int c, int p) -> float {
        if (mask_in) return mask_in[((int64_t)img * C + c) * HW + p];
        int cm = Cm == 1 ? 0 : c;
        if (keep_lds) return s_keep[cm * HW + p] ? 1.f : 0.f;
        float uv;
        if (u) uv = u[(int64_t)img * Cm * HW + (int64_t)cm * HW + p];
        else {
            uint64_t e = ((uint64_t)img * Cm + cm) * HW + p;
            uint4 r = philox_at(rng, rng_stream, e >> 2);
            uint32_t w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
            uv = u01(w);
        }
        return ((double)uv > thr) ? 1.f : 0.f;      // reference compares fp32 uniforms with the fp64 ratio
    };
    if (t < 8) { s_sum[t] = 0.f; s_cnt[t] = 0.f; s_fill[t] = fill_const; }
    __syncthreads();
    if (fill_mode != 0) {
        for (int c = 0; c < C; ++c) {
            float a = 0.f, n = 0.f;
            for (int p = t; p < HW; p += 256) {
                float k = keep_at(c, p);
                float xv = x0[((int64_t)img * C + c) * HW + p];
                if (fill_mode == 3) { a += xv * k; n += 1.f - k; }
                else { a += xv * (1.f - k); n += 1.f - k; }
            }
            a = wave_sum(a); n = wave_sum(n);
            if ((t & 63) == 0) { s_wsum[t >> 6][c] = a; s_wcnt[t >> 6][c] = n; }      // per-wave slots, summed in wave order below
        }
        __syncthreads();
        if (t < C) {
            s_sum[t] = (s_wsum[0][t] + s_wsum[1][t]) + (s_wsum[2][t] + s_wsum[3][t]);
            s_cnt[t] = (s_wcnt[0][t] + s_wcnt[1][t]) + (s_wcnt[2][t] + s_wcnt[3][t]);
        }
        __syncthreads();
        if (t == 0) {
            if (fill_mode == 1) {            // degraded_area, image-wise: one value for all channels
                float a = 0.f, n = 0.f;
                for (int c = 0; c < C; ++c) { a += s_sum[c]; n += s_cnt[c]; }
                for (int c = 0; c < C; ++c) s_fill[c] = a / n;
            } else if (fill_mode == 2) {     // degraded_area, channel-wise
                for (int c = 0; c < C; ++c) s_fill[c] = s_sum[c] / s_cnt[c];
            } else {                         // non_degraded_area: -(sum kept)/(count degraded), NaN -> 0
                for (int c = 0; c < C; ++c) {
                    float v = s_sum[c] / s_cnt[c] * -1.f;
                    s_fill[c] = (v != v) ? 0.f : v;
                }
            }
        }
        __syncthreads();
    }
    if (mean_pixel && t < C) mean_pixel[(int64_t)img * C + t] = s_fill[t];
    const float inv_hw = rcp_small(HW);
    for (int i = t; i < C * HW; i += 256) {
        int c = (C * HW < (1 << 20) && HW <= 1024) ? div_small(i, inv_hw) : i / HW, p = i - c * HW;
        float k = keep_at(c, p);
        int64_t o = ((int64_t)img * C + c) * HW + p;
        if (x_t) x_t[o] = (1.f - k) * s_fill[c] + k * x0[o];
        if (mask) mask[o] = k;
    }
}

// 'indexing' mode on device: keep-mask with EXACTLY count[n] zeros per image, chosen as the
// count smallest of HW i.i.d. uniform keys (same distribution as randperm(HW)[:count]).
__global__ __launch_bounds__(256) void index_mask_kernel(const double* count, int count_stride, const uint64_t* rng,
                                                         int rng_stream, int C, int HW, float* mask) {
    extern __shared__ uint32_t keys[];
    const int img = blockIdx.x, t = threadIdx.x;
    for (int p = t; p < HW; p += 256) {
        uint64_t e = (uint64_t)img * HW + p;
        uint4 r = philox_at(rng, rng_stream, e);
        keys[p] = r.x;
    }
    __syncthreads();
    const int cnt = (int)count[(int64_t)img * count_stride];
    for (int p = t; p < HW; p += 256) {
        uint32_t kp = keys[p];
        int rank = 0;
        for (int q = 0; q < HW; ++q) rank += (keys[q] < kp) || (keys[q] == kp && q < p);
        float k = rank < cnt ? 0.f : 1.f;
        for (int c = 0; c < C; ++c) mask[((int64_t)img * C + c) * HW + p] = k;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void shift_kernel(const float* x_t, const float* z, const double* ratio, const uint64_t* rng,
                                                    int rng_stream, int kind, float noise_mean, int per_column, int N, int C,
                                                    int H, int W, float* s_out, float* x_in, T* x_nhwc, int Cp) {
    const int HW = H * W;
    const int64_t total = (int64_t)N * C * HW;
    const int zc = (kind == 2 || kind == 4 || kind == 5) ? 3 : 1;
    const int zhw = (kind == 3 || kind == 4 || kind == 5) ? HW : 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int p = (int)(i % HW);
        int64_t r = i / HW;
        int c = (int)(r % C), n = (int)(r / C);
        float sv = 0.f;
        if (kind != 0) {
            int czi = zc == 1 ? 0 : c;
            int64_t zi = ((int64_t)n * zc + czi) * zhw + (zhw == 1 ? 0 : p);
            float zv;
            if (z) zv = z[zi];
            else if (kind == 5) {     // noise_std_reduction: N(noise_mean, ratio_n) (scheduler.py:691-694)
                uint4 q = philox_at(rng, rng_stream, (uint64_t)zi >> 1);
                float2 g = box_muller(q.x, q.y);
                zv = (float)((double)noise_mean + (double)((zi & 1) ? g.y : g.x) * ratio[n]);
            } else {
                uint4 q = philox_at(rng, rng_stream, (uint64_t)zi >> 1);
                if (kind == 1 || kind == 2) zv = u01((zi & 1) ? q.z : q.x) * 2.f - 1.f;
                else { float2 g = box_muller(q.x, q.y); zv = ((zi & 1) ? g.y : g.x) + noise_mean; }
            }
            // fp32 draw times fp64 ratio, rounded once to fp32 (scheduler.py:625-626, 680, 713, 725)
            if (kind == 5) sv = zv;         // the draw already carries mean and std
            else {
                double rt = (per_column && (kind == 3 || kind == 4)) ? ratio[p % W] : ratio[n];
                sv = (float)((double)zv * rt);
            }
        }
        float xv = x_t[i] + sv;
        if (s_out) s_out[i] = sv;
        if (x_in) x_in[i] = xv;
        if (x_nhwc) Elem<T>::st(x_nhwc + ((int64_t)n * HW + p) * Cp + c, xv);
    }
}
template <typename T>
__global__ void zero_pad_channels_kernel(T* x_nhwc, int C, int Cp, int64_t npix) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int padc = Cp - C;
    if (i >= npix * padc) return;
    int64_t pix = i / padc;
    int c = C + (int)(i - pix * padc);
    Elem<T>::st(x_nhwc + pix * Cp + c, 0.f);
}

// One thread per PIXEL (n, p): its Cp padded channels of pred / dpred are contiguous (NHWC), the fp32 operands are NCHW planes
// that neighbouring threads read at neighbouring addresses, and every load of a thread is independent of the others (the
// element-per-thread version walked 8 dependent iterations of three scattered loads: 15 us for 100 k elements).
template <typename T>
__global__ __launch_bounds__(256) void loss_kernel(const T* pred, const float* x_in, const float* s, const float* x0,
                                                   const float* w, int N, int C, int HW, int Cp, float gscale, T* dpred,
                                                   unsigned long long* loss_q40) {
    const int64_t npix = (int64_t)N * HW;
    const float inv_numel = 1.f / ((float)N * (float)C * (float)HW);
    float local = 0.f;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(pix / HW), p = (int)(pix - (int64_t)n * HW);
        const float wn = w ? w[n] : 1.f;
        for (int c0 = 0; c0 < Cp; c0 += 8) {
            float pv[8], g[8];
            if (Elem<T>::VEC == 8) {
                const float8 v = load8(pred + pix * Cp + c0);
                pv[0] = v.lo.x; pv[1] = v.lo.y; pv[2] = v.lo.z; pv[3] = v.lo.w; pv[4] = v.hi.x; pv[5] = v.hi.y; pv[6] = v.hi.z; pv[7] = v.hi.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = Elem<T>::ld(pred + pix * Cp + c0 + e);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c0 + e;
                g[e] = 0.f;
                if (c < C) {
                    const int64_t o = ((int64_t)n * C + c) * HW + p;
                    // fp32 registers, reference order: (x_in + pred) - s - x0
                    float r = x_in[o] + pv[e];
                    if (s) r -= s[o];
                    r -= x0[o];
                    local = fmaf(wn * r, r, local);
                    g[e] = 2.f * wn * r * inv_numel * gscale;
                }
            }
            if (dpred) {
#pragma unroll
                for (int e = 0; e < 8; ++e) Elem<T>::st(dpred + pix * Cp + c0 + e, g[e]);
            }
        }
    }
    local = wave_sum(local);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    // The workgroups meet in a FIXED-POINT accumulator (Q23.40, two's complement): integer adds commute, so the sum does not
    // depend on the order the workgroups finish in (a float atomic does: the fp32 loss was not reproducible run to run).
    // 2^-40 resolves every partial >= 2^-16 exactly; a partial that is not finite or >= 2^22 bumps the flag word instead.
    if (threadIdx.x == 0) {
        const float v = ((part[0] + part[1]) + (part[2] + part[3])) * inv_numel;
        if (fabsf(v) < 4194304.f) atomicAdd(loss_q40, (unsigned long long)__float2ll_rn(v * 1099511627776.f));
        else atomicAdd(loss_q40 + 1, 1ull);
    }
}

template <typename T>
__global__ void sampler_x0_kernel(const T* pred, int Cp, const float* x_in, const float* s, int C, int HW, int64_t total,
                                  float* pred_nchw, float* shifted0, float* x0_hat) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;    // NCHW index
    if (i >= total) return;
    int p = (int)(i % HW);
    int64_t r = i / HW;
    int c = (int)(r % C);
    int64_t n = r / C;
    float pv = Elem<T>::ld(pred + (n * HW + p) * Cp + c);
    float sh0 = x_in[i] + pv;
    if (pred_nchw) pred_nchw[i] = pv;
    if (shifted0) shifted0[i] = sh0;
    x0_hat[i] = s ? sh0 - s[i] : sh0;
}
__global__ void sampler_update_kernel(const float* d_t, const float* d_next, float* x_t, float* diff, int momentum, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float df = d_next[i] - d_t[i];
    if (diff) diff[i] = df;
    x_t[i] = momentum ? x_t[i] + df : d_next[i];
}

// The Philox offset lives in device memory and is bumped ON the device, in stream order: a host counter copied over
// asynchronously from pinned memory is read when the copy executes, by which time a host that runs ahead of the GPU
// (it always does under hipGraph replay) may have bumped it again -- steps would skip or repeat offsets.
__global__ void rng_advance_kernel(unsigned long long* rng) { rng[1] += 1ull; }

// Per-step parameters of the reverse sampler, produced ON the device so that a whole reverse step can be one hipGraph
// replayed T times with no host work in between (sampler.py:137-170): step counter -> timestep t of this step,
// t_next = t - 1 (t on the last step), the shift ratio and the two degrade amounts from the schedule tables, and the
// Philox offset bump the host path does with DeviceRng.advance().
__global__ void sampler_step_kernel(const int* timesteps, int T, int* step_ctr, const double* ratio_tab, const double* amount_tab,
                                    int n, float* time_out, double* ratio_out, double* amt_t, double* amt_next,
                                    unsigned long long* rng) {
    const int step = *step_ctr;
    const int i = T - 1 - step;
    const int t = timesteps[i < 0 ? 0 : i];
    const int tn = i > 0 ? t - 1 : t;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        time_out[j] = (float)t;
        if (ratio_out) ratio_out[j] = ratio_tab[t - 1];
        amt_t[j] = amount_tab[t - 1];
        amt_next[j] = amount_tab[tn - 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) { *step_ctr = step + 1; rng[1] += 1ull; }
}

static inline int sgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace mdm
using namespace mdm;

extern "C" int mdm_draw_timesteps(const uint64_t* rng, const int32_t* used, int n_used, const double* table, const float* wtab,
                                  int N, float* t_out, double* amount_out, float* weight_out, int32_t* idx_out, const double* table2,
                                  double* out2, int64_t* zero_out, void* stream) {
    MDM_REQUIRE(rng && used && table && n_used > 0 && N > 0 && (!out2 || table2), "draw_timesteps: bad arguments");
    hipLaunchKernelGGL(draw_timesteps_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rng, used, n_used, table,
                       wtab, N, t_out, amount_out, weight_out, idx_out, table2, out2, reinterpret_cast<long long*>(zero_out));
    return launch_status("draw_timesteps");
}

extern "C" int mdm_degrade(const float* x0, const float* u, const float* mask_in, const double* amount, int amount_stride,
                           const uint64_t* rng, int rng_stream, int N, int C, int HW, int Cm, int fill_mode, float fill_const,
                           float* x_t, float* mask, float* mean_pixel, void* stream) {
    MDM_REQUIRE(x0 && N > 0 && C > 0 && C <= 8 && HW > 0, "degrade: bad shape (C <= 8)");
    MDM_REQUIRE(Cm == 1 || Cm == C, "degrade: Cm must be 1 or C");
    MDM_REQUIRE(fill_mode >= 0 && fill_mode <= 3, "degrade: bad fill_mode");
    MDM_REQUIRE(mask_in || (amount && (u || rng)), "degrade: need mask_in, or amount with u / rng");
    const int keep_lds = (!mask_in && HW % 4 == 0 && Cm * HW <= 48 * 1024) ? 1 : 0;
    hipLaunchKernelGGL(degrade_kernel, dim3(N), dim3(256), keep_lds ? Cm * HW : 0, (hipStream_t)stream, x0, u, mask_in, amount,
                       amount_stride, rng, rng_stream, C, HW, Cm, fill_mode, fill_const, x_t, mask, mean_pixel, keep_lds);
    return launch_status("degrade");
}

extern "C" int mdm_index_mask(const double* count, int count_stride, const uint64_t* rng, int rng_stream, int N, int C, int HW,
                              float* mask, void* stream) {
    MDM_REQUIRE(count && rng && mask && HW * 4 <= 64 * 1024, "index_mask: bad arguments (HW <= 16384)");
    hipLaunchKernelGGL(index_mask_kernel, dim3(N), dim3(256), HW * sizeof(uint32_t), (hipStream_t)stream, count, count_stride,
                       rng, rng_stream, C, HW, mask);
    return launch_status("index_mask");
}

extern "C" int mdm_shift(const float* x_t, const float* z, const double* ratio, const uint64_t* rng, int rng_stream, int kind,
                         float noise_mean, int per_column, int N, int C, int H, int W, float* s, float* x_in, int dtype,
                         void* x_in_nhwc, int Cp, void* stream) {
    MDM_REQUIRE(x_t && N > 0 && C > 0 && H > 0 && W > 0, "shift: bad shape");
    MDM_REQUIRE(kind >= 0 && kind <= 5, "shift: bad kind %d", kind);
    MDM_REQUIRE(kind == 0 || (ratio && (z || rng)), "shift: need ratio and z / rng");
    MDM_REQUIRE(!(kind == 2 || kind == 4 || kind == 5) || C == 3, "shift: kinds 2/4 are hard-wired to 3 channels upstream (D9)");
    MDM_REQUIRE(!per_column || N == W, "shift: per_column needs N == W");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)N * C * H * W;
    if (x_in_nhwc) MDM_REQUIRE(Cp >= C && Cp % 8 == 0, "shift: bad Cp");
    if (dtype == MDM_BF16) {
        hipLaunchKernelGGL((shift_kernel<bf16_t>), dim3(sgrid(total)), dim3(256), 0, st, x_t, z, ratio, rng, rng_stream, kind,
                           noise_mean, per_column, N, C, H, W, s, x_in, (bf16_t*)x_in_nhwc, Cp);
    } else {
        hipLaunchKernelGGL((shift_kernel<float>), dim3(sgrid(total)), dim3(256), 0, st, x_t, z, ratio, rng, rng_stream, kind,
                           noise_mean, per_column, N, C, H, W, s, x_in, (float*)x_in_nhwc, Cp);
    }
    return launch_status("shift");
}

extern "C" int mdm_zero_pad_channels(int dtype, void* x_nhwc, int64_t npix, int C, int Cp, void* stream) {
    MDM_REQUIRE(x_nhwc && npix > 0 && C > 0 && Cp >= C && Cp % 8 == 0, "zero_pad_channels: bad arguments");
    MDM_REQUIRE(dtype == MDM_F32 || dtype == MDM_BF16, "zero_pad_channels: bad dtype %d", dtype);
    if (Cp == C) return 0;
    const int64_t n = npix * (Cp - C);
    if (dtype == MDM_BF16)
        hipLaunchKernelGGL((zero_pad_channels_kernel<bf16_t>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x_nhwc, C, Cp, npix);
    else
        hipLaunchKernelGGL((zero_pad_channels_kernel<float>), dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (float*)x_nhwc, C, Cp, npix);
    return launch_status("zero_pad_channels");
}

extern "C" int mdm_loss_fwd_bwd(int dtype, const void* pred, const float* x_in, const float* s, const float* x0, const float* w,
                                int N, int C, int H, int W, int Cp, float gscale, void* dpred, int64_t* loss_q40, void* stream) {
    MDM_REQUIRE(pred && x_in && x0 && loss_q40 && Cp >= C && Cp % 8 == 0, "loss: bad arguments (Cp must be a multiple of 8)");
    unsigned long long* loss_accum = reinterpret_cast<unsigned long long*>(loss_q40);
    int grid = sgrid((int64_t)N * H * W);
    if (grid > 256) grid = 256;            // one same-address atomic per workgroup at the end: keep them few
    if (dtype == MDM_BF16)
        hipLaunchKernelGGL((loss_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, x_in, s, x0, w,
                           N, C, H * W, Cp, gscale, (bf16_t*)dpred, loss_accum);
    else
        hipLaunchKernelGGL((loss_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)pred, x_in, s, x0, w, N,
                           C, H * W, Cp, gscale, (float*)dpred, loss_accum);
    return launch_status("loss");
}

extern "C" int mdm_sampler_x0(int dtype, const void* pred_nhwc, int Cp, const float* x_in, const float* s, int N, int C, int H,
                              int W, float* pred_nchw, float* shifted0, float* x0_hat, void* stream) {
    MDM_REQUIRE(pred_nhwc && x_in && x0_hat, "sampler_x0: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    if (dtype == MDM_BF16)
        hipLaunchKernelGGL((sampler_x0_kernel<bf16_t>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)pred_nhwc, Cp, x_in, s, C, H * W, total, pred_nchw, shifted0, x0_hat);
    else
        hipLaunchKernelGGL((sampler_x0_kernel<float>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)pred_nhwc, Cp, x_in, s, C, H * W, total, pred_nchw, shifted0, x0_hat);
    return launch_status("sampler_x0");
}

extern "C" int mdm_rng_advance(uint64_t* rng, void* stream) {
    MDM_REQUIRE(rng, "rng_advance: null state");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(rng));
    return launch_status("rng_advance");
}

extern "C" int mdm_sampler_step_params(const int32_t* timesteps, int T, int32_t* step_ctr, const double* ratio_tab,
                                       const double* amount_tab, int n, float* time_out, double* ratio_out, double* amt_t,
                                       double* amt_next, uint64_t* rng, void* stream) {
    MDM_REQUIRE(timesteps && step_ctr && amount_tab && time_out && amt_t && amt_next && rng && T > 0 && n > 0,
                "sampler_step_params: bad arguments");
    MDM_REQUIRE(!ratio_out || ratio_tab, "sampler_step_params: ratio_out needs the ratio table");
    hipLaunchKernelGGL(sampler_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, timesteps, T, step_ctr, ratio_tab, amount_tab,
                       n, time_out, ratio_out, amt_t, amt_next, reinterpret_cast<unsigned long long*>(rng));
    return launch_status("sampler_step_params");
}

extern "C" int mdm_sampler_update(const float* d_t, const float* d_next, float* x_t, float* diff, int momentum, int64_t n,
                                  void* stream) {
    MDM_REQUIRE(d_t && d_next && x_t, "sampler_update: bad arguments");
    hipLaunchKernelGGL(sampler_update_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, d_t, d_next, x_t, diff, momentum, n);
    return launch_status("sampler_update");
}
