// GroupNorm(32)+SiLU forward/backward, row softmax, column sums, 2x2 sum-pool,
// NCHW<->NHWC converters.  All HBM-bound streaming kernels over NHWC with 16-byte
// vectors (8 channels) per lane; reductions are wave-shuffle + LDS + fp32 atomics.
//
// Replaces nn.GroupNorm / nn.SiLU (reference unet6.py:288-293, 358-360), torch.softmax
// (unet6.py:320-322), nn.Upsample backward (unet6.py:472), and the bias / time-embedding
// broadcast backward sums (unet6.py:233, 359).
#include "common.h"

namespace mdm {

// Thread mapping used by the four GroupNorm kernels: a workgroup owns one image and a
// slab of pixels; thread t always handles the same 8-channel vector (t % VPP), so its
// per-channel partial sums live in registers for the whole slab.
struct GnGeom {
    int C0, C1, C, VPP, lanes_used, pix_per_pass, cpg;
};
__device__ __forceinline__ GnGeom gn_geom(int C0, int C1, int G) {
    GnGeom g;
    g.C0 = C0; g.C1 = C1; g.C = C0 + C1;
    g.VPP = g.C / 8;
    g.pix_per_pass = 256 / g.VPP;
    if (g.pix_per_pass < 1) g.pix_per_pass = 1;
    g.lanes_used = g.VPP * g.pix_per_pass;
    g.cpg = g.C / G;
    return g;
}

template <typename T>
__device__ __forceinline__ const T* src_ptr(const T* s0, const T* s1, int C0, int C1, int64_t pix, int c) {
    return c < C0 ? s0 + pix * C0 + c : s1 + pix * C1 + (c - C0);
}

constexpr int GN_MAX_SLABS = 32;   // workgroups per image at most (bounds the partial-sum scratch)

// ---- forward pass 1: per (image, group) SHIFTED sums  sum(x-K), sum((x-K)^2) -> ws[N][G][2]
// K = the group's first element of that image.  Shifting removes the cancellation of
// E[x^2]-mean^2 and makes constant feature maps (a fully degraded, all-zero input image gives
// them) come out with variance exactly 0 and mean exactly K, as the reference's two-pass
// group_norm does.
template <typename T>
__device__ __forceinline__ float gn_pivot(const T* s0, const T* s1, int C0, int C1, int64_t img_pix0, int grp, int cpg) {
    return Elem<T>::ld(src_ptr(s0, s1, C0, C1, img_pix0, grp * cpg));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* s0, int C0, const T* s1, int C1, int P, int G, float* ws) {
    const GnGeom g = gn_geom(C0, C1, G);
    __shared__ float red[2 * 64];      // G <= 64
    const int t = threadIdx.x, img = blockIdx.y;
    if (t < 2 * G) red[t] = 0.f;
    __syncthreads();
    if (t < g.lanes_used) {
        const int v = t % g.VPP, c = v * 8;
        float s[8] = {}, q[8] = {}, K[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) K[e] = gn_pivot(s0, s1, C0, C1, (int64_t)img * P, (c + e) / g.cpg, g.cpg);
        const int span = (P + gridDim.x - 1) / gridDim.x;     // pixels per workgroup
        const int p_beg = blockIdx.x * span;
        const int p_end = min(P, p_beg + span);
        auto add = [&](const float8& x) {
            float xv[8] = {x.lo.x, x.lo.y, x.lo.z, x.lo.w, x.hi.x, x.hi.y, x.hi.z, x.hi.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) { float dlt = xv[e] - K[e]; s[e] += dlt; q[e] = fmaf(dlt, dlt, q[e]); }
        };
        const int st = g.pix_per_pass;
        int p = p_beg + t / g.VPP;
        const int64_t base = (int64_t)img * P;
        for (; p + 3 * st < p_end; p += 4 * st) {          // four independent 16-byte loads in flight per lane
            float8 x0 = load8(src_ptr(s0, s1, C0, C1, base + p, c));
            float8 x1 = load8(src_ptr(s0, s1, C0, C1, base + p + st, c));
            float8 x2 = load8(src_ptr(s0, s1, C0, C1, base + p + 2 * st, c));
            float8 x3 = load8(src_ptr(s0, s1, C0, C1, base + p + 3 * st, c));
            add(x0); add(x1); add(x2); add(x3);
        }
        for (; p < p_end; p += st) add(load8(src_ptr(s0, s1, C0, C1, base + p, c)));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int grp = (c + e) / g.cpg;
            atomicAdd(&red[2 * grp], s[e]);
            atomicAdd(&red[2 * grp + 1], q[e]);
        }
    }
    __syncthreads();
    if (t < 2 * G) ws[((int64_t)img * gridDim.x + blockIdx.x) * 2 * G + t] = red[t];     // partial of this slab
}

// per (image, channel): sum the slab partials of the channel's group, finish mean / rstd (group
// leader also stores them in `stats` for the backward), emit {mean, rstd*gamma} so the streaming
// pass does two vector loads instead of per-element look-ups.  y = (x-mean)*a + beta keeps x-mean exact.
template <typename T>
__global__ void gn_coef_kernel(const T* s0, int C0, const T* s1, int C1, int P, int G, const float* ws, int nblk,
                               const float* gamma, float inv_cnt, float eps, float* stats, float2* coef, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;      // over N*C
    if (i >= n) return;
    const int C = C0 + C1, cpg = C / G;
    int img = i / C, c = i - img * C, grp = c / cpg;
    float sd = 0.f, sq = 0.f;
    for (int b = 0; b < nblk; ++b) {
        const float* p = ws + ((int64_t)img * nblk + b) * 2 * G + 2 * grp;
        sd += p[0]; sq += p[1];
    }
    float K = gn_pivot(s0, s1, C0, C1, (int64_t)img * P, grp, cpg);
    float md = sd * inv_cnt;
    float var = fmaxf(sq * inv_cnt - md * md, 0.f);
    float mean = K + md, rstd = rsqrtf(var + eps);
    if (c == grp * cpg) { stats[((int64_t)img * G + grp) * 2] = mean; stats[((int64_t)img * G + grp) * 2 + 1] = rstd; }
    coef[i] = make_float2(mean, rstd * gamma[c]);
}
// backward: {mean, rstd, rstd*s1/cnt, rstd*s2/cnt}
__global__ void gn_bwd_coef_kernel(const float* stats, const float* ws, int nblk, int C, int G, float inv_cnt, float4* coef, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int img = i / C, c = i - img * C, grp = c / (C / G);
    float a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < nblk; ++b) {
        const float* p = ws + ((int64_t)img * nblk + b) * 2 * G + 2 * grp;
        a1 += p[0]; a2 += p[1];
    }
    int64_t si = ((int64_t)img * G + grp) * 2;
    float mean = stats[si], rstd = stats[si + 1];
    coef[i] = make_float4(mean, rstd, rstd * a1 * inv_cnt, rstd * a2 * inv_cnt);
}

// ---- forward pass 2: y = act((x - mean) * (rstd*gamma) + beta)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* s0, int C0, const T* s1, int C1, int P,
                                                       const float2* coef, const float* beta, int silu, T* y, int64_t total_vec) {
    const int C = C0 + C1, VPP = C / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix = i / VPP;
        int c = (int)(i - pix * VPP) * 8;
        int img = (int)(pix / P);
        float8 x = load8(src_ptr(s0, s1, C0, C1, pix, c));
        const float4* cf = reinterpret_cast<const float4*>(coef + (int64_t)img * C + c);     // 8 x {mean, a}
        float4 c01 = cf[0], c23 = cf[1], c45 = cf[2], c67 = cf[3];
        float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
        float o[8];
        o[0] = fmaf(x.lo.x - c01.x, c01.y, b0.x); o[1] = fmaf(x.lo.y - c01.z, c01.w, b0.y);
        o[2] = fmaf(x.lo.z - c23.x, c23.y, b0.z); o[3] = fmaf(x.lo.w - c23.z, c23.w, b0.w);
        o[4] = fmaf(x.hi.x - c45.x, c45.y, b1.x); o[5] = fmaf(x.hi.y - c45.z, c45.w, b1.y);
        o[6] = fmaf(x.hi.z - c67.x, c67.y, b1.z); o[7] = fmaf(x.hi.w - c67.z, c67.w, b1.w);
        if (silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = silu_f(o[e]);
        }
        float8 r = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
        store8(y + pix * C + c, r);
    }
}

// ---- backward pass 1: per (image, group) s1 = sum g*gamma, s2 = sum g*gamma*xhat; dgamma/dbeta
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* s0, int C0, const T* s1, int C1, int P, int G,
                                                           const float* gamma, const float* beta, int silu,
                                                           const T* dy, const float* stats, float* ws,
                                                           float* dgamma, float* dbeta) {
    const GnGeom g = gn_geom(C0, C1, G);
    __shared__ float red[2 * 64];
    extern __shared__ float chan[];          // [2][C]: per-channel dgamma / dbeta partials of this workgroup
    const int t = threadIdx.x, img = blockIdx.y;
    if (t < 2 * G) red[t] = 0.f;
    for (int i = t; i < 2 * g.C; i += 256) chan[i] = 0.f;
    __syncthreads();
    if (t < g.lanes_used) {
        const int v = t % g.VPP, c = v * 8;
        float ga[8], be[8], mean[8], rstd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int grp = (c + e) / g.cpg;
            ga[e] = gamma[c + e]; be[e] = beta[c + e];
            mean[e] = stats[((int64_t)img * G + grp) * 2]; rstd[e] = stats[((int64_t)img * G + grp) * 2 + 1];
        }
        float a1[8] = {}, a2[8] = {}, dg[8] = {}, db[8] = {};
        const int span = (P + gridDim.x - 1) / gridDim.x;
        const int p_beg = blockIdx.x * span;
        const int p_end = min(P, p_beg + span);
        auto add = [&](const float8& x, const float8& d) {
            float xv[8] = {x.lo.x, x.lo.y, x.lo.z, x.lo.w, x.hi.x, x.hi.y, x.hi.z, x.hi.w};
            float dv[8] = {d.lo.x, d.lo.y, d.lo.z, d.lo.w, d.hi.x, d.hi.y, d.hi.z, d.hi.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xh = (xv[e] - mean[e]) * rstd[e];
                float gz = dv[e];
                if (silu) gz *= silu_grad_f(xh * ga[e] + be[e]);
                dg[e] = fmaf(gz, xh, dg[e]); db[e] += gz;
                float gg = gz * ga[e];
                a1[e] += gg; a2[e] = fmaf(gg, xh, a2[e]);
            }
        };
        const int st = g.pix_per_pass;
        int p = p_beg + t / g.VPP;
        const int64_t base = (int64_t)img * P;
        for (; p + st < p_end; p += 2 * st) {              // two pixel rows (four 16-byte loads) in flight per lane
            float8 x0 = load8(src_ptr(s0, s1, C0, C1, base + p, c));
            float8 d0 = load8(dy + (base + p) * g.C + c);
            float8 x1 = load8(src_ptr(s0, s1, C0, C1, base + p + st, c));
            float8 d1 = load8(dy + (base + p + st) * g.C + c);
            add(x0, d0); add(x1, d1);
        }
        for (; p < p_end; p += st) add(load8(src_ptr(s0, s1, C0, C1, base + p, c)), load8(dy + (base + p) * g.C + c));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int grp = (c + e) / g.cpg;
            atomicAdd(&red[2 * grp], a1[e]);
            atomicAdd(&red[2 * grp + 1], a2[e]);
            atomicAdd(&chan[c + e], dg[e]);                 // LDS atomics: the pixel lanes of one channel meet here
            atomicAdd(&chan[g.C + c + e], db[e]);
        }
    }
    __syncthreads();
    if (t < 2 * G) ws[((int64_t)img * gridDim.x + blockIdx.x) * 2 * G + t] = red[t];
    for (int i = t; i < g.C; i += 256) {                    // one global atomic per channel per workgroup
        atomicAdd(&dgamma[i], chan[i]);
        atomicAdd(&dbeta[i], chan[g.C + i]);
    }
}

// ---- backward pass 2: dx = rstd*gamma*g - (rstd*s1/cnt + xhat*rstd*s2/cnt),  g = dy * act'(xhat*gamma + beta)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* s0, int C0, const T* s1, int C1, int P,
                                                           const float* gamma, const float* beta, int silu,
                                                           const T* dy, const float4* coef,
                                                           T* d0, int acc0, T* d1, int acc1, int64_t total_vec) {
    const int C = C0 + C1, VPP = C / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix = i / VPP;
        int c = (int)(i - pix * VPP) * 8;
        int img = (int)(pix / P);
        float8 x = load8(src_ptr(s0, s1, C0, C1, pix, c));
        float8 d = load8(dy + pix * C + c);
        float xv[8] = {x.lo.x, x.lo.y, x.lo.z, x.lo.w, x.hi.x, x.hi.y, x.hi.z, x.hi.w};
        float dv[8] = {d.lo.x, d.lo.y, d.lo.z, d.lo.w, d.hi.x, d.hi.y, d.hi.z, d.hi.w};
        const float4* cf = coef + (int64_t)img * C + c;
        float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
        float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
        float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float4 k = cf[e];                          // mean, rstd, rstd*s1/cnt, rstd*s2/cnt
            float xh = (xv[e] - k.x) * k.y;
            float gz = dv[e];
            if (silu) gz *= silu_grad_f(fmaf(xh, gv[e], bv[e]));
            o[e] = k.y * gv[e] * gz - fmaf(xh, k.w, k.z);
        }
        T* dst; int acc;
        if (c < C0) { dst = d0 + pix * C0 + c; acc = acc0; } else { dst = d1 + pix * C1 + (c - C0); acc = acc1; }
        if (acc) {
            float8 old = load8(dst);
            o[0] += old.lo.x; o[1] += old.lo.y; o[2] += old.lo.z; o[3] += old.lo.w;
            o[4] += old.hi.x; o[5] += old.hi.y; o[6] += old.hi.z; o[7] += old.hi.w;
        }
        float8 r = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
        store8(dst, r);
    }
}

// ---- row softmax: one wave per row
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(T* S, int rows, int L) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    T* p = S + (int64_t)row * L;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, Elem<T>::ld(p + j));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) sum += __expf(Elem<T>::ld(p + j) - mx);
    sum = wave_sum(sum);
    float inv = 1.f / sum;
    for (int j = lane; j < L; j += 64) Elem<T>::st(p + j, __expf(Elem<T>::ld(p + j) - mx) * inv);
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* Pm, T* dP, int rows, int L) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* p = Pm + (int64_t)row * L;
    T* g = dP + (int64_t)row * L;
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) dot = fmaf(Elem<T>::ld(p + j), Elem<T>::ld(g + j), dot);
    dot = wave_sum(dot);
    for (int j = lane; j < L; j += 64) Elem<T>::st(g + j, Elem<T>::ld(p + j) * (Elem<T>::ld(g + j) - dot));
}

// ---- column sums of dY[N][P][C]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* dY, int P, int C, float* per_img, int ld, int acc_img, float* dbias) {
    // grid (C/8 vector groups of 32.., N): thread t -> vector v = blockIdx.x*32 + t%32? keep it simple:
    // each workgroup owns one image and 8 vectors (64 channels); 32 pixel lanes per vector.
    __shared__ float red[32][65];
    const int img = blockIdx.y, t = threadIdx.x;
    const int v = blockIdx.x * 8 + (t & 7), lanep = t >> 3;      // 8 vectors x 32 pixel lanes
    float s[8] = {};
    if (v * 8 < C) {
        for (int p = lanep; p < P; p += 32) {
            float8 x = load8(dY + ((int64_t)img * P + p) * C + v * 8);
            s[0] += x.lo.x; s[1] += x.lo.y; s[2] += x.lo.z; s[3] += x.lo.w;
            s[4] += x.hi.x; s[5] += x.hi.y; s[6] += x.hi.z; s[7] += x.hi.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[lanep][(t & 7) * 8 + e] = s[e];
    __syncthreads();
    if (t < 64) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) a += red[r][t];
        int c = blockIdx.x * 64 + t;
        if (c < C) {
            if (per_img) {
                float* q = per_img + (int64_t)img * ld + c;
                *q = acc_img ? *q + a : a;
            }
            if (dbias) atomicAdd(&dbias[c], a);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sumpool2_kernel(const T* g, T* dst, int acc, int H, int W, int C, int64_t total_vec) {
    const int VPP = C / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix = i / VPP;
        int c = (int)(i - pix * VPP) * 8;
        int x = (int)(pix % W);
        int64_t r = pix / W;
        int y = (int)(r % H);
        int64_t img = r / H;
        const T* base = g + (((img * 2 * H + 2 * y) * 2 * W) + 2 * x) * (int64_t)C + c;
        float8 a = load8(base), b = load8(base + C), cc = load8(base + (int64_t)2 * W * C), dd = load8(base + (int64_t)2 * W * C + C);
        float8 o;
        o.lo = make_float4(a.lo.x + b.lo.x + cc.lo.x + dd.lo.x, a.lo.y + b.lo.y + cc.lo.y + dd.lo.y,
                           a.lo.z + b.lo.z + cc.lo.z + dd.lo.z, a.lo.w + b.lo.w + cc.lo.w + dd.lo.w);
        o.hi = make_float4(a.hi.x + b.hi.x + cc.hi.x + dd.hi.x, a.hi.y + b.hi.y + cc.hi.y + dd.hi.y,
                           a.hi.z + b.hi.z + cc.hi.z + dd.hi.z, a.hi.w + b.hi.w + cc.hi.w + dd.hi.w);
        T* q = dst + pix * C + c;
        if (acc) {
            float8 old = load8(q);
            o.lo.x += old.lo.x; o.lo.y += old.lo.y; o.lo.z += old.lo.z; o.lo.w += old.lo.w;
            o.hi.x += old.hi.x; o.hi.y += old.hi.y; o.hi.z += old.hi.z; o.hi.w += old.hi.w;
        }
        store8(q, o);
    }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int C, int HW, int Cp, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over N*HW*Cp
    if (i >= total) return;
    int c = (int)(i % Cp);
    int64_t pix = i / Cp;
    int64_t img = pix / HW;
    int p = (int)(pix - img * HW);
    float v = c < C ? x[(img * C + c) * HW + p] : 0.f;
    Elem<T>::st(y + i, v);
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int C, int HW, int Cp, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over N*C*HW
    if (i >= total) return;
    int p = (int)(i % HW);
    int64_t r = i / HW;
    int c = (int)(r % C);
    int64_t img = r / C;
    y[i] = Elem<T>::ld(x + (img * HW + p) * Cp + c);
}

template <typename T>
__global__ void add_kernel(T* dst, const T* src, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float8 a = load8(dst + i * 8), b = load8(src + i * 8);
        a.lo.x += b.lo.x; a.lo.y += b.lo.y; a.lo.z += b.lo.z; a.lo.w += b.lo.w;
        a.hi.x += b.hi.x; a.hi.y += b.hi.y; a.hi.z += b.hi.z; a.hi.w += b.hi.w;
        store8(dst + i * 8, a);
    }
}

__global__ void temb_kernel(const float* t, int N, int dim, float* y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int half = dim / 2;
    if (i >= N * half) return;
    int n = i / half, j = i - n * half;
    float f = expf(-(float)j * (logf(10000.f) / (float)(half - 1)));
    float a = t[n] * f;
    y[n * dim + j] = sinf(a);
    y[n * dim + half + j] = cosf(a);
    if ((dim & 1) && j == 0) y[n * dim + dim - 1] = 0.f;
}
__global__ void silu_fwd_kernel(const float* x, float* y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = x[i]; y[i] = v / (1.f + expf(-v)); }
}
__global__ void silu_bwd_kernel(const float* x, const float* dy, float* dx, int acc, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = x[i], s = 1.f / (1.f + expf(-v));
        float g = dy[i] * s * (1.f + v * (1.f - s));
        dx[i] = acc ? dx[i] + g : g;
    }
}

__global__ void zero_f32_kernel(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

static inline int stream_grid(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace mdm

using namespace mdm;

#define DISPATCH_T(dtype, ...)                                      \
    do {                                                            \
        if ((dtype) == MDM_F32) { typedef float T; __VA_ARGS__; }   \
        else if ((dtype) == MDM_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else { set_error("bad dtype %d", (dtype)); return -1; }     \
    } while (0)

static int gn_check(int C0, int C1, int G, int N, int P) {
    MDM_REQUIRE(C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0, "groupnorm: channel counts must be multiples of 8 (C0=%d C1=%d)", C0, C1);
    MDM_REQUIRE(G > 0 && G <= 64 && (C0 + C1) % G == 0, "groupnorm: C=%d not divisible by G=%d", C0 + C1, G);
    MDM_REQUIRE((C0 + C1) / 8 <= 256, "groupnorm: C=%d too large", C0 + C1);
    MDM_REQUIRE(N > 0 && P > 0, "groupnorm: bad N/P");
    return 0;
}

// ws layout: [N][nblk][G][2] slab partials (nblk <= GN_MAX_SLABS), then [N][C] float4 coefficient slots
static inline int gn_slabs(int P, int ppp) {
    int nb = cdiv(P, 8 * ppp);                    // ~8 passes per workgroup (measured: 2 passes = 4x the workgroups is slower)
    return nb < 1 ? 1 : (nb > GN_MAX_SLABS ? GN_MAX_SLABS : nb);
}
extern "C" int mdm_groupnorm_fwd(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int P, int G,
                                 float eps, const float* gamma, const float* beta, int silu, void* y, float* stats,
                                 float* ws, void* stream) {
    if (int rc = gn_check(C0, C1, G, N, P)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int C = C0 + C1, ppp = (256 / (C / 8)) < 1 ? 1 : 256 / (C / 8);
    const int nblk = gn_slabs(P, ppp);
    dim3 g1(nblk, N);
    const int64_t tv = (int64_t)N * P * (C / 8);
    float2* coef = reinterpret_cast<float2*>(ws + (int64_t)N * GN_MAX_SLABS * 2 * G);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((gn_stats_kernel<T>), g1, dim3(256), 0, s, (const T*)src0, C0, (const T*)src1, C1, P, G, ws);
        hipLaunchKernelGGL((gn_coef_kernel<T>), dim3(cdiv(N * C, 256)), dim3(256), 0, s, (const T*)src0, C0, (const T*)src1, C1, P, G,
                           ws, nblk, gamma, 1.f / ((float)(C / G) * (float)P), eps, stats, coef, N * C);
        hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(stream_grid(tv)), dim3(256), 0, s, (const T*)src0, C0, (const T*)src1, C1,
                           P, coef, beta, silu, (T*)y, tv);
    });
    return launch_status("groupnorm_fwd");
}

extern "C" int mdm_groupnorm_bwd(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int P, int G,
                                 const float* gamma, const float* beta, int silu, const void* dy, const float* stats,
                                 void* dst0, int acc0, void* dst1, int acc1, float* dgamma, float* dbeta, float* ws,
                                 void* stream) {
    if (int rc = gn_check(C0, C1, G, N, P)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int C = C0 + C1, ppp = (256 / (C / 8)) < 1 ? 1 : 256 / (C / 8);
    const int nblk = gn_slabs(P, ppp);
    dim3 g1(nblk, N);
    const int64_t tv = (int64_t)N * P * (C / 8);
    float4* coef = reinterpret_cast<float4*>(ws + (int64_t)N * GN_MAX_SLABS * 2 * G);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL((gn_bwd_stats_kernel<T>), g1, dim3(256), 2 * C * sizeof(float), s, (const T*)src0, C0, (const T*)src1, C1, P, G, gamma,
                           beta, silu, (const T*)dy, stats, ws, dgamma, dbeta);
        hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, s, stats, ws, nblk, C, G,
                           1.f / ((float)(C / G) * (float)P), coef, N * C);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(stream_grid(tv)), dim3(256), 0, s, (const T*)src0, C0, (const T*)src1,
                           C1, P, gamma, beta, silu, (const T*)dy, coef, (T*)dst0, acc0, (T*)dst1, acc1, tv);
    });
    return launch_status("groupnorm_bwd");
}

extern "C" int mdm_softmax_fwd(int dtype, void* S, int rows, int L, void* stream) {
    MDM_REQUIRE(rows > 0 && L > 0, "softmax: bad shape");
    DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_fwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (T*)S, rows, L));
    return launch_status("softmax_fwd");
}
extern "C" int mdm_softmax_bwd(int dtype, const void* P, void* dP, int rows, int L, void* stream) {
    MDM_REQUIRE(rows > 0 && L > 0, "softmax: bad shape");
    DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)P, (T*)dP, rows, L));
    return launch_status("softmax_bwd");
}

extern "C" int mdm_timestep_embedding(const float* t, int N, int dim, float* y, void* stream) {
    MDM_REQUIRE(N > 0 && dim >= 4, "timestep_embedding: bad shape");
    hipLaunchKernelGGL(temb_kernel, dim3(cdiv(N * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, N, dim, y);
    return launch_status("timestep_embedding");
}
extern "C" int mdm_silu_fwd(const float* x, float* y, int64_t n, void* stream) {
    hipLaunchKernelGGL(silu_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return launch_status("silu_fwd");
}
extern "C" int mdm_silu_bwd(const float* x, const float* dy, float* dx, int acc, int64_t n, void* stream) {
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, acc, n);
    return launch_status("silu_bwd");
}

extern "C" int mdm_colsum(int dtype, const void* dY, int N, int P, int C, float* per_img, int ld, int acc_img, float* dbias,
                          void* stream) {
    MDM_REQUIRE(C % 8 == 0 && N > 0 && P > 0, "colsum: bad shape");
    dim3 grid(cdiv(C, 64), N);
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, pick_stream(stream), (const T*)dY, P, C, per_img, ld, acc_img, dbias));
    return launch_status("colsum");
}

extern "C" int mdm_sumpool2(int dtype, const void* g, void* dst, int acc, int N, int H, int W, int C, void* stream) {
    MDM_REQUIRE(C % 8 == 0, "sumpool2: C must be a multiple of 8");
    const int64_t tv = (int64_t)N * H * W * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((sumpool2_kernel<T>), dim3(stream_grid(tv)), dim3(256), 0, (hipStream_t)stream, (const T*)g, (T*)dst, acc, H, W, C, tv));
    return launch_status("sumpool2");
}

extern "C" int mdm_add(int dtype, void* dst, const void* src, int64_t n, void* stream) {
    MDM_REQUIRE(n % 8 == 0 && dst && src, "add: n must be a multiple of 8");
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (T*)dst, (const T*)src, n / 8));
    return launch_status("add");
}

extern "C" int mdm_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, int Cp, void* stream) {
    MDM_REQUIRE(Cp >= C && Cp % 8 == 0, "nchw_to_nhwc: bad Cp");
    const int64_t total = (int64_t)N * H * W * Cp;
    DISPATCH_T(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (T*)y, C, H * W, Cp, total));
    return launch_status("nchw_to_nhwc");
}
extern "C" int mdm_nhwc_to_nchw(int dtype, const void* x, float* y, int N, int C, int H, int W, int Cp, void* stream) {
    MDM_REQUIRE(Cp >= C, "nhwc_to_nchw: bad Cp");
    const int64_t total = (int64_t)N * C * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, y, C, H * W, Cp, total));
    return launch_status("nhwc_to_nchw");
}
