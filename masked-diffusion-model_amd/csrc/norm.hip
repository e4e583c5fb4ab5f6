// GroupNorm(32)+SiLU forward/backward, row softmax, column sums, 2x2 sum-pool,
// NCHW<->NHWC converters.  All HBM-bound streaming kernels over NHWC with 16-byte
// vectors (8 channels) per lane.  Reductions inside a workgroup have a FIXED order (LDS column sums, one thread per
// group); across workgroups the fp32 path goes through per-image partials + a second stage (bit-reproducible), the
// bf16 path through one float atomic per channel per image.
//
// Replaces nn.GroupNorm / nn.SiLU (reference unet6.py:288-293, 358-360), torch.softmax
// (unet6.py:320-322), nn.Upsample backward (unet6.py:472), and the bias / time-embedding
// broadcast backward sums (unet6.py:233, 359).
#include <stdlib.h>

#include "common.h"

namespace mdm {

// GroupNorm is ONE kernel per direction.  A workgroup owns one image and a block of CBLK channels
// made of whole groups (CBLK = multiple of lcm(C/G, 8), >= 32): pass 1 streams its [P][CBLK] slice
// once for the statistics (lane = fixed 8-channel vector, so per-channel partials stay in registers;
// pixel lanes meet in LDS), pass 2 re-reads the slice (64 KB for the largest unet6 tensor: L2-resident)
// and applies.  No scratch buffers, no zeroing, no atomics on the statistics.
template <typename T>
__device__ __forceinline__ const T* src_ptr(const T* s0, const T* s1, int C0, int C1, int64_t pix, int c) {
    return c < C0 ? s0 + pix * C0 + c : s1 + pix * C1 + (c - C0);
}
// K = the group's first element of that image.  Shifting the sums by it removes the cancellation of
// E[x^2]-mean^2 and makes constant feature maps (a fully degraded, all-zero input image gives them)
// come out with variance exactly 0 and mean exactly K, as the reference's two-pass group_norm does.
template <typename T>
__device__ __forceinline__ float gn_pivot(const T* s0, const T* s1, int C0, int C1, int64_t img_pix0, int grp, int cpg) {
    return Elem<T>::ld(src_ptr(s0, s1, C0, C1, img_pix0, grp * cpg));
}
#define F8_TO_ARR(v) {v.lo.x, v.lo.y, v.lo.z, v.lo.w, v.hi.x, v.hi.y, v.hi.z, v.hi.w}

// Sum over the lanes of a wave that own the same channel vector (lane ids congruent mod VB, VB a power of two):
// butterfly over the offsets VB, 2VB, ... 32.  Afterwards lanes 0..VB-1 hold the wave's totals, and only those
// touch LDS -- 64 lanes adding to one LDS address with ds_add_f32 would serialise completely.
// ---- workgroup column sums through LDS.  Every thread holds K partial values; thread t belongs to column
// v = t % VB (its 8-channel vector) and the sums run over the 256/VB pixel lanes of each column.
// (The first version reduced with __shfl_xor: 4 ds_bpermute round trips per value, 128 per thread in the
// backward -- 20k cycles of a 36k-cycle kernel.)  scratch: K * CS_PITCH floats; out[k * VB + v].
#ifndef MDM_GN_COLSUM_SHFL
#define MDM_GN_COLSUM_SHFL 1        // 0: every column sum through LDS (A/B builds)
#endif
// Column sums inside the wave by RECURSIVE HALVING (VB a power of two: the lanes of a column are those with equal low log2(VB) bits):
// at the step over lane bit b a lane keeps one half of the quantities it still holds (the upper half where its bit is set), sends the
// other half to its partner and adds what the partner sends -- K/2 + K/4 + ... <= K - 1 ds_bpermute per lane for ALL K quantities
// (the plain butterfly of the first version needed log2(lanes) per quantity: 128 in the backward), after which lane l holds quantity
// q0(l) (+ r, r < K >> steps) summed over its column's lanes of this wave.  Those K * VB values per wave go to LDS ([wave][k][v]) and
// K * VB threads add the waves up in ascending order.  LDS traffic: NT/64 * K * VB floats instead of K * NT.
template <int K, int S>
__device__ __forceinline__ void colsum_halving_step(float (&cur)[K], int& q0, int l, int vsh) {
    const int bitpos = vsh + S;
    if (bitpos < 6) {                                                 // uniform
        const int mask = 1 << bitpos;
        const bool up = (l >> bitpos) & 1;
        constexpr int n = (K >> S) > 1 ? (K >> S) : 1;                // quantities a lane holds in front of step S
        if constexpr (n > 1) {
            constexpr int half = n / 2;
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const float send = up ? cur[j] : cur[half + j];
                const float keep = up ? cur[half + j] : cur[j];
                cur[j] = keep + __shfl_xor(send, mask, 64);
            }
            q0 += up ? half : 0;
        } else {
            cur[0] += __shfl_xor(cur[0], mask, 64);
        }
    }
}
template <int K, int NT = 256>
__device__ __forceinline__ void block_colsum_shfl(const float (&val)[K], float* scratch, float* out, int t, int VB) {
    static_assert((K & (K - 1)) == 0 && K >= 8, "block_colsum_shfl: K a power of two");
    const int l = t & 63, w = t >> 6, v = l & (VB - 1);
    const int vsh = __builtin_ctz(VB);                               // uniform
    float cur[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = val[k];
    int q0 = 0;
    colsum_halving_step<K, 0>(cur, q0, l, vsh); colsum_halving_step<K, 1>(cur, q0, l, vsh); colsum_halving_step<K, 2>(cur, q0, l, vsh);
    colsum_halving_step<K, 3>(cur, q0, l, vsh); colsum_halving_step<K, 4>(cur, q0, l, vsh); colsum_halving_step<K, 5>(cur, q0, l, vsh);
    const int steps = 6 - vsh;
    const int R = (K >> steps) > 1 ? (K >> steps) : 1;               // quantities left per lane (K >> steps == 0: duplicates, same value)
    constexpr int RMAX = (K >> 3) > 1 ? (K >> 3) : 1;                // VB <= 8
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        if (r < R) scratch[(w * K + q0 + r) * VB + v] = cur[r];
    __syncthreads();
    if (t < K * VB) {
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < NT / 64; ++ww) sum += scratch[ww * K * VB + t];
        out[t] = sum;
    }
    __syncthreads();
}
// SHFL: the bf16 kernels (measured: step 3.528 -> 3.502 ms); the fp32-storage forward of the reverse sampler keeps the LDS sums (its
// lanes hold 8 NP fp32 registers of the slice: with the halving's K more the sampler read 5.36 - 5.47 against 5.33 ms per reverse step)
template <int K, int NT = 256, bool SHFL = true>
__device__ __forceinline__ void block_colsum(const float (&val)[K], float* scratch, float* out, int t, int VB, int PL) {
    if (MDM_GN_COLSUM_SHFL && SHFL && (VB & (VB - 1)) == 0 && VB <= 8) {     // uniform
        block_colsum_shfl<K, NT>(val, scratch, out, t, VB);
        return;
    }
    constexpr int CS_PITCH = NT + 4;
#pragma unroll
    for (int k = 0; k < K; ++k) scratch[k * CS_PITCH + t] = val[k];
    __syncthreads();
    const int cols = K * VB;                       // <= 256
    int tpc = 1, tpc_sh = 0;                       // threads per column (power of two)
    while (tpc * 2 * cols <= NT) { tpc *= 2; ++tpc_sh; }
    const int col = t >> tpc_sh, part = t - (col << tpc_sh);
    float sum = 0.f;
    if (col < cols) {
        const int k = div_small(col, rcp_small(VB)), v = col - k * VB;
        // the tpc threads of a column take the pixel lanes INTERLEAVED (l = part, part + tpc, ...): neighbouring lanes then
        // read neighbouring LDS words.  (Contiguous shares put the threads of a column a multiple of 64 words apart --
        // 8-way bank conflicts, 62 % of the GroupNorm kernels' LDS cycles by SQ_LDS_BANK_CONFLICT.)
        const float* src = scratch + k * CS_PITCH + v;
        for (int l = part; l < PL; l += tpc) sum += src[l * VB];
    }
    for (int o = 1; o < tpc; o <<= 1) sum += __shfl_xor(sum, o, 64);     // <= 3 steps on ONE value
    if (col < cols && part == 0) out[col] = sum;
    __syncthreads();
}

// out[w] = sum over the `cpg` consecutive channels lc0 .. lc0+cpg-1 (local to the workgroup's channel block) of quantity w of
// block_colsum's output: quantity w of channel lc sits at csum[(w * qstride + (lc & 7)) * VB + (lc >> 3)].
// One thread, ascending channel order.
template <int NQ>
__device__ __forceinline__ void group_sums(const float* csum, int VB, int lc0, int cpg, int qstride, float* out) {
    float a[NQ];
#pragma unroll
    for (int w = 0; w < NQ; ++w) a[w] = 0.f;
    for (int j = 0; j < cpg; ++j) {
        const int lc = lc0 + j, vv = lc >> 3, e = lc & 7;
#pragma unroll
        for (int w = 0; w < NQ; ++w) a[w] += csum[(w * qstride + e) * VB + vv];
    }
#pragma unroll
    for (int w = 0; w < NQ; ++w) out[w] = a[w];
}

template <typename T>
__global__ __launch_bounds__(256) void gn_fwd_kernel(const T* s0, int C0, const T* s1, int C1, int P, int G, int CBLK,
                                                     float eps, const float* gamma, const float* beta, int silu, T* y,
                                                     float* stats) {
    const int C = C0 + C1, cpg = div_small(C, rcp_small(G));
    const float inv_cpg = rcp_small(cpg);
    const int VB = CBLK >> 3, PL = div_small(256, rcp_small(VB));
    const int img = blockIdx.x, cb = blockIdx.y * CBLK;     // image fastest: the channel blocks of one image (they share 128-B lines) land on one XCD
    const int t = threadIdx.x, lane = div_small(t, rcp_small(VB)), v = t - lane * VB, c = cb + v * 8;
    const int ng = div_small(CBLK, inv_cpg), g0 = div_small(cb, inv_cpg);
    const bool on = t < VB * PL && c < C;
    __shared__ float gsum[2 * 64], gmean[64], grstd[64];
    __shared__ float scratch[16 * (256 + 4)];
    __shared__ float csum[16 * 8];
    const int64_t base = (int64_t)img * P;
    float part[16];                       // [0, 8): sums of (x - K), [8, 16): sums of (x - K)^2, per channel of this lane's vector
#pragma unroll
    for (int k = 0; k < 16; ++k) part[k] = 0.f;
    if (on) {
        float K[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) K[e] = gn_pivot(s0, s1, C0, C1, base, div_small(c + e, inv_cpg), cpg);
        auto add = [&](const float8& x) {
            float xv[8] = F8_TO_ARR(x);
#pragma unroll
            for (int e = 0; e < 8; ++e) { float dlt = xv[e] - K[e]; part[e] += dlt; part[8 + e] = fmaf(dlt, dlt, part[8 + e]); }
        };
        int p = lane;
        for (; p + 3 * PL < P; p += 4 * PL) {          // four independent 16-byte loads in flight per lane
            float8 x0 = load8(src_ptr(s0, s1, C0, C1, base + p, c));
            float8 x1 = load8(src_ptr(s0, s1, C0, C1, base + p + PL, c));
            float8 x2 = load8(src_ptr(s0, s1, C0, C1, base + p + 2 * PL, c));
            float8 x3 = load8(src_ptr(s0, s1, C0, C1, base + p + 3 * PL, c));
            add(x0); add(x1); add(x2); add(x3);
        }
        for (; p < P; p += PL) add(load8(src_ptr(s0, s1, C0, C1, base + p, c)));
    }
    // FIXED summation order (no float atomics: the same input gives the same bits on every box): pixel lanes per channel
    // through block_colsum, then one thread per group walks its channels in order
    block_colsum<16, 256>(part, scratch, csum, t, VB, PL);          // csum[(q*8+e)*VB + v]
    if (t < ng) group_sums<2>(csum, VB, t * cpg, cpg, 8, &gsum[2 * t]);
    __syncthreads();
    if (t < ng && (g0 + t) < G) {
        const float inv_cnt = 1.f / ((float)cpg * (float)P);
        float K = gn_pivot(s0, s1, C0, C1, base, g0 + t, cpg);
        float md = gsum[2 * t] * inv_cnt;
        float var = fmaxf(gsum[2 * t + 1] * inv_cnt - md * md, 0.f);
        float mean = K + md, rstd = rsqrtf(var + eps);
        gmean[t] = mean; grstd[t] = rstd;
        stats[((int64_t)img * G + g0 + t) * 2] = mean;
        stats[((int64_t)img * G + g0 + t) * 2 + 1] = rstd;
    }
    __syncthreads();
    if (on) {
        float m[8], a[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int gl = div_small(c + e, inv_cpg) - g0;
            m[e] = gmean[gl]; a[e] = grstd[gl] * gamma[c + e]; bt[e] = beta[c + e];
        }
        auto put = [&](int p, const float8& x) {
            float xv[8] = F8_TO_ARR(x);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = fmaf(xv[e] - m[e], a[e], bt[e]);     // (x - mean) stays exact
                if (silu) o[e] = silu_f(o[e]);
            }
            float8 r = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
            store8(y + (base + p) * C + c, r);
        };
        int p = lane;
        for (; p + 3 * PL < P; p += 4 * PL) {
            float8 x0 = load8(src_ptr(s0, s1, C0, C1, base + p, c));
            float8 x1 = load8(src_ptr(s0, s1, C0, C1, base + p + PL, c));
            float8 x2 = load8(src_ptr(s0, s1, C0, C1, base + p + 2 * PL, c));
            float8 x3 = load8(src_ptr(s0, s1, C0, C1, base + p + 3 * PL, c));
            put(p, x0); put(p + PL, x1); put(p + 2 * PL, x2); put(p + 3 * PL, x3);
        }
        for (; p < P; p += PL) put(p, load8(src_ptr(s0, s1, C0, C1, base + p, c)));
    }
}

// backward: dx = rstd*gamma*g - rstd*(s1 + xhat*s2)/cnt with g = dy * act'(xhat*gamma + beta),
// s1 = sum(g*gamma), s2 = sum(g*gamma*xhat) per (image, group); dgamma += sum g*xhat, dbeta += sum g.
// Every reduction has a FIXED order.  Inside the workgroup: block_colsum + one thread per group.  Across the images:
// `part` != nullptr (the fp32 path) -> this workgroup's per-channel sums go to part[img][{dgamma, dbeta}][C] (plain stores)
// and gn_param_reduce_kernel adds them up image by image behind this launch (it also forms sum_all from sum_img);
// part == nullptr (bf16 large maps) -> one float atomic per channel per image, as the register-cached kernels do.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const T* s0, int C0, const T* s1, int C1, int P, int G, int CBLK,
                                                     const float* gamma, const float* beta, int silu, const T* dy,
                                                     const float* stats, T* d0, const T* add0, T* d1, const T* add1, const T* add0b,
                                                     float* dgamma, float* dbeta, float* sum_img, int sum_ld, float* sum_all,
                                                     float* part) {
    const int C = C0 + C1, cpg = div_small(C, rcp_small(G));
    const float inv_cpg = rcp_small(cpg);
    const int VB = CBLK >> 3, PL = div_small(256, rcp_small(VB));
    const int img = blockIdx.x, cb = blockIdx.y * CBLK;     // image fastest: the channel blocks of one image (they share 128-B lines) land on one XCD
    const int t = threadIdx.x, lane = div_small(t, rcp_small(VB)), v = t - lane * VB, c = cb + v * 8;
    const int ng = div_small(CBLK, inv_cpg), g0 = div_small(cb, inv_cpg);
    const bool on = t < VB * PL && c < C;
    __shared__ float gsum[2 * 64];
    __shared__ float scratch[32 * (256 + 4)];
    __shared__ float csum[32 * 8];
    const int64_t base = (int64_t)img * P;
    float ga[8], be[8], mean[8], rstd[8];
    float acc[32];                        // per channel of this lane's vector: a1, a2, dgamma, dbeta
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int grp = div_small(c + e, inv_cpg);
            ga[e] = gamma[c + e]; be[e] = beta[c + e];
            mean[e] = stats[((int64_t)img * G + grp) * 2]; rstd[e] = stats[((int64_t)img * G + grp) * 2 + 1];
        }
        auto add = [&](const float8& x, const float8& d) {
            float xv[8] = F8_TO_ARR(x);
            float dv[8] = F8_TO_ARR(d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xh = (xv[e] - mean[e]) * rstd[e];
                float gz = dv[e];
                if (silu) gz *= silu_grad_f(fmaf(xh, ga[e], be[e]));
                acc[16 + e] = fmaf(gz, xh, acc[16 + e]); acc[24 + e] += gz;
                float gg = gz * ga[e];
                acc[e] += gg; acc[8 + e] = fmaf(gg, xh, acc[8 + e]);
            }
        };
        int p = lane;
        for (; p + PL < P; p += 2 * PL) {
            float8 x0 = load8(src_ptr(s0, s1, C0, C1, base + p, c));
            float8 e0 = load8(dy + (base + p) * C + c);
            float8 x1 = load8(src_ptr(s0, s1, C0, C1, base + p + PL, c));
            float8 e1 = load8(dy + (base + p + PL) * C + c);
            add(x0, e0); add(x1, e1);
        }
        for (; p < P; p += PL) add(load8(src_ptr(s0, s1, C0, C1, base + p, c)), load8(dy + (base + p) * C + c));
    }
    block_colsum<32, 256>(acc, scratch, csum, t, VB, PL);          // csum[(q*8+e)*VB + v], q = {a1, a2, dgamma, dbeta}
    if (t < ng) group_sums<2>(csum, VB, t * cpg, cpg, 8, &gsum[2 * t]);
    if (t < CBLK && cb + t < C) {            // this workgroup is the only one that holds (image, channel)
        const float dgv = csum[(16 + (t & 7)) * VB + (t >> 3)], dbv = csum[(24 + (t & 7)) * VB + (t >> 3)];
        if (part) {
            part[((int64_t)img * 3) * C + cb + t] = dgv;
            part[((int64_t)img * 3 + 1) * C + cb + t] = dbv;
        } else {
            atomicAdd(&dgamma[cb + t], dgv);
            atomicAdd(&dbeta[cb + t], dbv);
        }
    }
    __syncthreads();
    float k1[8], k2[8], ag[8];
    if (on) {
        const float inv_cnt = 1.f / ((float)cpg * (float)P);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int gl = div_small(c + e, inv_cpg) - g0;
            k1[e] = rstd[e] * gsum[2 * gl] * inv_cnt;
            k2[e] = rstd[e] * gsum[2 * gl + 1] * inv_cnt;
            ag[e] = rstd[e] * ga[e];
        }
    }
    float sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of dx (bias / time-embedding gradient of the producer conv)
    if (on) {
        T* dst; const T* addp; int cc, CS;           // addp: a tensor laid out like dst whose values are added (dst itself = accumulate)
        if (c < C0) { dst = d0; addp = add0; cc = c; CS = C0; } else { dst = d1; addp = add1; cc = c - C0; CS = C1; }
        const T* addq = c < C0 ? add0b : nullptr;    // a second addend for source 0 (accumulate AND a residual-branch gradient)
        auto put = [&](int p, const float8& x, const float8& d) {
            float xv[8] = F8_TO_ARR(x);
            float dv[8] = F8_TO_ARR(d);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xh = (xv[e] - mean[e]) * rstd[e];
                float gz = dv[e];
                if (silu) gz *= silu_grad_f(fmaf(xh, ga[e], be[e]));
                o[e] = ag[e] * gz - fmaf(xh, k2[e], k1[e]);
                sx[e] += o[e];
            }
            T* q = dst + (base + p) * CS + cc;
            if (addp) {
                float8 old = load8(addp + (base + p) * CS + cc);
                float ov[8] = F8_TO_ARR(old);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += ov[e];
            }
            if (addq) {
                float8 old2 = load8(addq + (base + p) * CS + cc);
                float ov2[8] = F8_TO_ARR(old2);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += ov2[e];
            }
            float8 r = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
            store8(q, r);
        };
        int p = lane;
        for (; p + PL < P; p += 2 * PL) {
            float8 x0 = load8(src_ptr(s0, s1, C0, C1, base + p, c));
            float8 e0 = load8(dy + (base + p) * C + c);
            float8 x1 = load8(src_ptr(s0, s1, C0, C1, base + p + PL, c));
            float8 e1 = load8(dy + (base + p + PL) * C + c);
            put(p, x0, e0); put(p + PL, x1, e1);
        }
        for (; p < P; p += PL) put(p, load8(src_ptr(s0, s1, C0, C1, base + p, c)), load8(dy + (base + p) * C + c));
    }
    if (sum_img || sum_all) {               // uniform
        block_colsum<8, 256>(sx, scratch, csum, t, VB, PL);
        if (t < CBLK && cb + t < C) {
            const float r = csum[(t & 7) * VB + (t >> 3)];
            if (sum_img) sum_img[(int64_t)img * sum_ld + cb + t] = r;              // one workgroup owns (image, channel)
            if (part) part[((int64_t)img * 3 + 2) * C + cb + t] = r;              // fixed-order mode: the reduce kernel adds the images up
            else if (sum_all) atomicAdd(&sum_all[cb + t], r);
        }
    }
}

// second stage of the fixed-order GroupNorm backward over part[img][{dgamma, dbeta, colsum(dx)}][C]: dgamma[c] += sum_img
// part[img][0][c], dbeta[c] += sum_img part[img][1][c], sum_all[c] += sum_img part[img][2][c], images in ascending order
// (one thread per channel)
__global__ __launch_bounds__(256) void gn_param_reduce_kernel(const float* part, int N, int C, float* dgamma, float* dbeta, float* sum_all) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f, s = 0.f;
    for (int n = 0; n < N; ++n) {
        a += part[((int64_t)n * 3) * C + c];
        b += part[((int64_t)n * 3 + 1) * C + c];
        if (sum_all) s += part[((int64_t)n * 3 + 2) * C + c];
    }
    dgamma[c] += a; dbeta[c] += b;
    if (sum_all) sum_all[c] += s;
}

// ---- register-cached variants (bf16): a lane's share of the slice is at most NP 16-byte vectors per tensor,
// so it is loaded ONCE (all NP loads in flight together), kept in registers across the statistics and the
// apply pass, and global memory sees a single read.  Same arithmetic as the streaming kernels above.
__device__ __forceinline__ float8 unpack8(const uint4& r) {
    float8 o;
    o.lo = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    o.hi = make_float4(__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
                       __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u));
    return o;
}

#ifdef MDM_STAMP
__device__ unsigned long long g_nstamp_buf[4096 * 16];
__device__ __forceinline__ unsigned long long nstamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MDM_T(...) __VA_ARGS__
#else
#define MDM_T(...)
#endif

// MODE 0: the whole GroupNorm in one workgroup per (image, channel block).  Large maps leave too few such
// workgroups to pull HBM bandwidth (128 for a 32x32x128 map: 14/24 us fwd/bwd against ~6/9 us of traffic),
// so there the pixels are cut into gridDim.z chunks: MODE 1 = statistics of one chunk -> partials in ws
// [image][chunk][group][2] (plain stores, nothing to zero, fixed summation order), MODE 2 = apply.
// NT = threads per workgroup: 512 on the larger slices (the kernels are VALU- and latency-bound at one wave per SIMD:
// twice the waves per workgroup halve each thread's serial share of exp / rcp work)
// (T = float, round 3: the fp32 forward -- the reverse sampler -- ran on the streaming kernel, two reads and a write per element;
//  a lane's share of 8 channels x NP pixels is NP float8, 8 NP registers.)
template <typename T> struct RegVec;
template <> struct RegVec<bf16_t> {
    typedef uint4 type;
    static __device__ __forceinline__ uint4 zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ uint4 ld(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ float8 unpack(const uint4& r) { return unpack8(r); }
};
template <> struct RegVec<float> {
    typedef float8 type;
    static __device__ __forceinline__ float8 zero() { return {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}; }
    static __device__ __forceinline__ float8 ld(const float* p) { return load8(p); }
    static __device__ __forceinline__ float8 unpack(const float8& r) { return r; }
};
template <int NP, int MODE, int NT = 256, typename T = bf16_t>
__global__ __launch_bounds__(NT) void gn_fwd_reg_kernel(const T* s0, int C0, const T* s1, int C1, int P, int G, int CBLK,
                                                         float eps, const float* gamma, const float* beta, int silu, T* y,
                                                         float* stats, float* ws) {
    constexpr int CS_PITCH = NT + 4;
    const int C = C0 + C1, cpg = div_small(C, rcp_small(G));
    const float inv_cpg = rcp_small(cpg);
    const int VB = CBLK >> 3, PL = div_small(NT, rcp_small(VB));
    const int img = blockIdx.x, cb = blockIdx.y * CBLK;     // image fastest: the channel blocks of one image (they share 128-B lines) land on one XCD
    const int t = threadIdx.x, lane = div_small(t, rcp_small(VB)), v = t - lane * VB, c = cb + v * 8;
    const int ng = div_small(CBLK, inv_cpg), g0 = div_small(cb, inv_cpg);
    const bool on = t < VB * PL && c < C;
    const int chunks = gridDim.z, chunk = blockIdx.z;
    const int plen = chunks == 1 ? P : (P + chunks - 1) / chunks, pbeg = chunk * plen, pend = min(P, pbeg + plen);
    __shared__ float scratch[16 * CS_PITCH];
    __shared__ float csum[16 * 8];
    __shared__ float gsum[2 * 64], gmean[64], grstd[64], gpiv[64];
    const int64_t base = (int64_t)img * P;
    if (t < 2 * ng) {
        float a = 0.f;
        if (MODE == 2)
            for (int ch = 0; ch < chunks; ++ch) a += ws[(((int64_t)img * chunks + ch) * G + g0 + (t >> 1)) * 2 + (t & 1)];
        gsum[t] = a;
    }
    // one pivot load per group (in flight together with the slice loads below), shared through LDS: every thread
    // loading its 8 pivots itself and the statistics thread loading its pivot AGAIN after the reduction put a second
    // memory round trip on the critical path of a ~3 us kernel
    if (t < ng && g0 + t < G) gpiv[t] = gn_pivot(s0, s1, C0, C1, base, g0 + t, cpg);
    typename RegVec<T>::type cx[NP];
    float part[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) part[k] = 0.f;
    if (on) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int p = pbeg + lane + i * PL;
            cx[i] = p < pend ? RegVec<T>::ld(src_ptr(s0, s1, C0, C1, base + p, c)) : RegVec<T>::zero();
        }
    }
    __syncthreads();
    if (MODE != 2 && on) {
        float K[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) K[e] = gpiv[div_small(c + e, inv_cpg) - g0];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (pbeg + lane + i * PL < pend) {
                float8 x = RegVec<T>::unpack(cx[i]);
                float xv[8] = F8_TO_ARR(x);
#pragma unroll
                for (int e = 0; e < 8; ++e) { float dlt = xv[e] - K[e]; part[e] += dlt; part[8 + e] = fmaf(dlt, dlt, part[8 + e]); }
            }
        }
    }
    if (MODE != 2) {
        block_colsum<16, NT, sizeof(T) == 2>(part, scratch, csum, t, VB, PL);      // csum[(q*8+e)*VB + v]
        // one thread per group walks its channels in order (it was an LDS float atomic per channel: arrival order)
        if (t < ng) {
            float gv[2];
            group_sums<2>(csum, VB, t * cpg, cpg, 8, gv);
            gsum[2 * t] += gv[0]; gsum[2 * t + 1] += gv[1];
        }
    }
    __syncthreads();
    if (MODE == 1) {
        if (t < 2 * ng && g0 + (t >> 1) < G) ws[(((int64_t)img * chunks + chunk) * G + g0 + (t >> 1)) * 2 + (t & 1)] = gsum[t];
        return;
    }
    if (t < ng && (g0 + t) < G) {
        const float inv_cnt = 1.f / ((float)cpg * (float)P);
        float K = gpiv[t];
        float md = gsum[2 * t] * inv_cnt;
        float var = fmaxf(gsum[2 * t + 1] * inv_cnt - md * md, 0.f);
        float mean = K + md, rstd = rsqrtf(var + eps);
        gmean[t] = mean; grstd[t] = rstd;
        if (chunk == 0) {
            stats[((int64_t)img * G + g0 + t) * 2] = mean;
            stats[((int64_t)img * G + g0 + t) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    if (on) {
        float m[8], a[8], bt[8];
        const float4 g_lo = *reinterpret_cast<const float4*>(gamma + c), g_hi = *reinterpret_cast<const float4*>(gamma + c + 4);
        const float4 b_lo = *reinterpret_cast<const float4*>(beta + c), b_hi = *reinterpret_cast<const float4*>(beta + c + 4);
        const float gv[8] = {g_lo.x, g_lo.y, g_lo.z, g_lo.w, g_hi.x, g_hi.y, g_hi.z, g_hi.w};
        const float bv[8] = {b_lo.x, b_lo.y, b_lo.z, b_lo.w, b_hi.x, b_hi.y, b_hi.z, b_hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int gl = div_small(c + e, inv_cpg) - g0;
            m[e] = gmean[gl]; a[e] = grstd[gl] * gv[e]; bt[e] = bv[e];
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int p = pbeg + lane + i * PL;
            if (p < pend) {
                float8 x = RegVec<T>::unpack(cx[i]);
                float xv[8] = F8_TO_ARR(x);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = fmaf(xv[e] - m[e], a[e], bt[e]);
                    if (silu) o[e] = silu_f(o[e]);
                }
                float8 r = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
                store8(y + (base + p) * C + c, r);
            }
        }
    }
}

template <int NP, int MODE, int NT = 256>
__global__ __launch_bounds__(NT) void gn_bwd_reg_kernel(const bf16_t* s0, int C0, const bf16_t* s1, int C1, int P, int G, int CBLK,
                                                         const float* gamma, const float* beta, int silu, const bf16_t* dy,
                                                         const float* stats, bf16_t* d0, const bf16_t* add0, bf16_t* d1, const bf16_t* add1, const bf16_t* add0b,
                                                         float* dgamma, float* dbeta, float* sum_img, int sum_ld, float* sum_all,
                                                         float* ws) {
    constexpr int CS_PITCH = NT + 4;
    const int C = C0 + C1, cpg = div_small(C, rcp_small(G));
    const float inv_cpg = rcp_small(cpg);
    const int VB = CBLK >> 3, PL = div_small(NT, rcp_small(VB));
    const int img = blockIdx.x, cb = blockIdx.y * CBLK;     // image fastest: the channel blocks of one image (they share 128-B lines) land on one XCD
    const int t = threadIdx.x, lane = div_small(t, rcp_small(VB)), v = t - lane * VB, c = cb + v * 8;
    const int ng = div_small(CBLK, inv_cpg), g0 = div_small(cb, inv_cpg);
    const bool on = t < VB * PL && c < C;
    const int chunks = gridDim.z, chunk = blockIdx.z;
    const int plen = chunks == 1 ? P : (P + chunks - 1) / chunks, pbeg = chunk * plen, pend = min(P, pbeg + plen);
    __shared__ float scratch[16 * CS_PITCH];
    __shared__ float csum[32 * 8];
    __shared__ float gsum[2 * 64], sgam[64];
    MDM_T(const unsigned long long ts0 = nstamp_now();)
    if (t >= 128 && t < 128 + CBLK && cb + t - 128 < C) sgam[t - 128] = gamma[cb + t - 128];     // for the group sums behind the column sums
    if (t < 2 * ng) {
        float a = 0.f;
        if (MODE == 2)
            for (int ch = 0; ch < chunks; ++ch) a += ws[(((int64_t)img * chunks + ch) * G + g0 + (t >> 1)) * 2 + (t & 1)];
        gsum[t] = a;
    }
    const int64_t base = (int64_t)img * P;
    uint4 cx[NP], cd[NP];
    // the tensors ADDED to dx (accumulated gradient / residual branch) are fetched with x and dy, not behind the reduction:
    // a second exposed memory round trip on a kernel that is one round trip + a reduction long
    constexpr bool PRE_ADD = NP <= 4;
    uint4 cadd[PRE_ADD ? NP : 1], caddq[PRE_ADD ? NP : 1];
    // single-launch mode with a small slice: dy * silu'(..) is kept in fp32 registers for the apply pass instead of being
    // recomputed (exp + rcp per element: the large-map kernels are VALU-bound, 1 wave per SIMD)
    constexpr bool CACHE = MODE == 0 && NP <= 8;
    float gzc[CACHE ? NP : 1][8];
    float ga[8], be[8], mean[8], rstd[8], nmr[8], za[8], zb[8];
    float part[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) part[k] = 0.f;
    MDM_T(unsigned long long ts1 = 0;)
    if (on) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int p = pbeg + lane + i * PL;
            bool ok = p < pend;
            cx[i] = ok ? *reinterpret_cast<const uint4*>(src_ptr(s0, s1, C0, C1, base + p, c)) : make_uint4(0, 0, 0, 0);
            cd[i] = ok ? *reinterpret_cast<const uint4*>(dy + (base + p) * C + c) : make_uint4(0, 0, 0, 0);
            if (PRE_ADD) {
                const bf16_t* ap = c < C0 ? add0 : add1;
                const int cc2 = c < C0 ? c : c - C0, CS2 = c < C0 ? C0 : C1;
                cadd[i] = (ok && ap) ? *reinterpret_cast<const uint4*>(ap + (base + p) * CS2 + cc2) : make_uint4(0, 0, 0, 0);
                caddq[i] = (ok && add0b && c < C0) ? *reinterpret_cast<const uint4*>(add0b + (base + p) * C0 + c) : make_uint4(0, 0, 0, 0);
            }
        }
        {
            const float4 g_lo = *reinterpret_cast<const float4*>(gamma + c), g_hi = *reinterpret_cast<const float4*>(gamma + c + 4);
            const float4 b_lo = *reinterpret_cast<const float4*>(beta + c), b_hi = *reinterpret_cast<const float4*>(beta + c + 4);
            ga[0] = g_lo.x; ga[1] = g_lo.y; ga[2] = g_lo.z; ga[3] = g_lo.w; ga[4] = g_hi.x; ga[5] = g_hi.y; ga[6] = g_hi.z; ga[7] = g_hi.w;
            be[0] = b_lo.x; be[1] = b_lo.y; be[2] = b_lo.z; be[3] = b_lo.w; be[4] = b_hi.x; be[5] = b_hi.y; be[6] = b_hi.z; be[7] = b_hi.w;
            // 8 consecutive channels touch at most 8/cpg + 1 groups; load each group's pair once
            const int gA = div_small(c, inv_cpg);
            float2 st_prev = *reinterpret_cast<const float2*>(stats + ((int64_t)img * G + gA) * 2);
            int g_prev = gA;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int grp = div_small(c + e, inv_cpg);
                if (grp != g_prev) { st_prev = *reinterpret_cast<const float2*>(stats + ((int64_t)img * G + grp) * 2); g_prev = grp; }
                mean[e] = st_prev.x; rstd[e] = st_prev.y;
            }
        }
        MDM_T(ts1 = nstamp_now();)
        // (round 4: these kernels are VALU-bound -- ~60 vector instructions per element at two waves per SIMD, finding 48.  The
        // normalisation and the affine map are one fma each from per-channel constants, and only TWO sums are kept per channel:
        // sum(gz xh) = dgamma and sum(gz) = dbeta; the group sums of gz gamma and gz gamma xh are gamma-weighted sums of those
        // two over the group's channels, taken once behind the column sums: 16 quantities through block_colsum instead of 32.)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            nmr[e] = -mean[e] * rstd[e];                     // xh = fma(x, rstd, nmr)
            za[e] = rstd[e] * ga[e]; zb[e] = fmaf(nmr[e], ga[e], be[e]);      // gamma xh + beta = fma(x, za, zb)
        }
        if (MODE != 2) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (pbeg + lane + i * PL < pend) {
                float8 x = unpack8(cx[i]), d = unpack8(cd[i]);
                float xv[8] = F8_TO_ARR(x);
                float dv[8] = F8_TO_ARR(d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xh = fmaf(xv[e], rstd[e], nmr[e]);
                    float gz = dv[e];
                    if (silu) gz *= silu_grad_f(fmaf(xv[e], za[e], zb[e]));
                    if (CACHE) gzc[CACHE ? i : 0][e] = gz;
                    part[e] = fmaf(gz, xh, part[e]); part[8 + e] += gz;                      // dgamma, dbeta
                }
            }
        }
        }
    }
    if (MODE != 2) {
        block_colsum<16, NT>(part, scratch, csum, t, VB, PL);      // csum[(q*8+e)*VB + v], q = {dgamma, dbeta}
        if (t < CBLK && cb + t < C) {            // across images: one float atomic per (image, channel) (bf16 path)
            const int vv = t >> 3, e = t & 7;
            atomicAdd(&dgamma[cb + t], csum[e * VB + vv]);
            atomicAdd(&dbeta[cb + t], csum[(8 + e) * VB + vv]);
        }
        if (t >= 64 && t < 64 + ng) {            // inside the workgroup: fixed order (a second wave, next to the atomics above)
            const int gi = t - 64;
            float a1 = 0.f, a2 = 0.f;            // sum over the group's channels of gamma dbeta / gamma dgamma, in channel order
            for (int lc = gi * cpg; lc < (gi + 1) * cpg; ++lc) {
                const float gm = sgam[lc];
                a1 = fmaf(gm, csum[(8 + (lc & 7)) * VB + (lc >> 3)], a1);
                a2 = fmaf(gm, csum[(lc & 7) * VB + (lc >> 3)], a2);
            }
            gsum[2 * gi] += a1; gsum[2 * gi + 1] += a2;
        }
    }
    MDM_T(const unsigned long long ts2 = nstamp_now();)
    __syncthreads();
    if (MODE == 1) {
        if (t < 2 * ng && g0 + (t >> 1) < G) ws[(((int64_t)img * chunks + chunk) * G + g0 + (t >> 1)) * 2 + (t & 1)] = gsum[t];
        if (sum_img && chunk == 0 && t < CBLK && cb + t < C) sum_img[(int64_t)img * sum_ld + cb + t] = 0.f;
        return;
    }
    MDM_T(const unsigned long long ts3 = nstamp_now(); const unsigned long long ts4 = ts3;)
    float k1[8], k2[8], ag[8];              // dx = ag gz - (k2 xh + k1) = fma(ag, gz, -fma(x, k2 rstd, k2 nmr + k1))
    if (on) {
        const float inv_cnt = 1.f / ((float)cpg * (float)P);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int gl = div_small(c + e, inv_cpg) - g0;
            const float q1 = rstd[e] * gsum[2 * gl] * inv_cnt, q2 = rstd[e] * gsum[2 * gl + 1] * inv_cnt;
            k1[e] = fmaf(q2, nmr[e], q1);
            k2[e] = q2 * rstd[e];
            ag[e] = za[e];
        }
    }
    float sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) {
        bf16_t* dst; const bf16_t* addp; int cc, CS;
        if (c < C0) { dst = d0; addp = add0; cc = c; CS = C0; } else { dst = d1; addp = add1; cc = c - C0; CS = C1; }
        const bf16_t* addq = c < C0 ? add0b : nullptr;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int p = pbeg + lane + i * PL;
            if (p < pend) {
                float8 x = unpack8(cx[i]), d = unpack8(cd[i]);
                float xv[8] = F8_TO_ARR(x);
                float dv[8] = F8_TO_ARR(d);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float gz;
                    if (CACHE) gz = gzc[CACHE ? i : 0][e];
                    else { gz = dv[e]; if (silu) gz *= silu_grad_f(fmaf(xv[e], za[e], zb[e])); }
                    o[e] = fmaf(ag[e], gz, -fmaf(xv[e], k2[e], k1[e]));
                    sx[e] += o[e];
                }
                bf16_t* q = dst + (base + p) * CS + cc;
                if (addp) {
                    float8 old = PRE_ADD ? unpack8(cadd[PRE_ADD ? i : 0]) : load8(addp + (base + p) * CS + cc);
                    float ov[8] = F8_TO_ARR(old);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += ov[e];
                }
                if (addq) {
                    float8 old2 = PRE_ADD ? unpack8(caddq[PRE_ADD ? i : 0]) : load8(addq + (base + p) * CS + cc);
                    float ov2[8] = F8_TO_ARR(old2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += ov2[e];
                }
                float8 r = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
                store8(q, r);
            }
        }
    }
    if (sum_img || sum_all) {            // uniform
        block_colsum<8, NT>(sx, scratch, csum, t, VB, PL);
        if (t < CBLK && cb + t < C) {
            const float r = csum[(t & 7) * VB + (t >> 3)];
            if (sum_img) {
                if (MODE == 0) sum_img[(int64_t)img * sum_ld + cb + t] = r;         // single writer
                else atomicAdd(&sum_img[(int64_t)img * sum_ld + cb + t], r);        // zeroed by the MODE 1 launch
            }
            if (sum_all) atomicAdd(&sum_all[cb + t], r);
        }
    }
#ifdef MDM_STAMP
    {
        const unsigned long long ts5 = nstamp_now();
        const unsigned widx = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (t >> 6);
        if ((t & 63) == 0 && widx < 4096) {
            unsigned long long* r = g_nstamp_buf + widx * 16;
            r[0] = 1; r[1] = ts1 - ts0; r[2] = ts2 - ts1; r[3] = ts3 - ts2; r[4] = ts4 - ts3; r[5] = ts5 - ts4; r[6] = ts0; r[7] = ts5;
        }
    }
#endif
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

// ---- column sums of dY[N][P][C]: per_img[n][c] (=|+=) sum_p dY[n][p][c], dbias[c] += sum_n sum_p dY[n][p][c].
// A workgroup owns 64 channels (8 vectors x 32 pixel lanes) of `imgs` consecutive images and walks them in order, so that
// with imgs = N (the fp32 path) dbias has ONE writer per channel and a fixed summation order; with imgs = 1 (bf16: more
// workgroups) the images meet through one float atomic per channel per image.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* dY, int P, int C, float* per_img, int ld, int acc_img, float* dbias,
                                                     int imgs, int N) {
    __shared__ float red[32][65];
    const int t = threadIdx.x;
    const int v = blockIdx.x * 8 + (t & 7), lanep = t >> 3;      // 8 vectors x 32 pixel lanes
    float total = 0.f;
    for (int img = blockIdx.y * imgs; img < min(N, (int)(blockIdx.y + 1) * imgs); ++img) {
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
            const int c = blockIdx.x * 64 + t;
            if (c < C && per_img) {
                float* q = per_img + (int64_t)img * ld + c;
                *q = acc_img ? *q + a : a;
            }
            total += a;
        }
        __syncthreads();
    }
    if (t < 64) {
        const int c = blockIdx.x * 64 + t;
        if (c < C && dbias) {
            if (imgs >= N) dbias[c] += total;            // single writer
            else atomicAdd(&dbias[c], total);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sumpool2_kernel(const T* g, T* dst, int acc, int H, int W, int C, int64_t total_vec) {
    const int VPP = C / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t pix, img;
        int c, x, y;
        if (total_vec < (1ll << 31)) {          // 32-bit quotients (a 64-bit division is ~100 instructions)
            const unsigned iu = (unsigned)i, pu = iu / (unsigned)VPP, ru = pu / (unsigned)W, mu = ru / (unsigned)H;
            pix = pu; c = (int)(iu - pu * (unsigned)VPP) * 8; x = (int)(pu - ru * (unsigned)W); y = (int)(ru - mu * (unsigned)H); img = mu;
        } else {
            pix = i / VPP;
            c = (int)(i - pix * VPP) * 8;
            x = (int)(pix % W);
            const int64_t r = pix / W;
            y = (int)(r % H);
            img = r / H;
        }
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
__global__ void add_kernel(T* dst, const T* x, const T* y, int64_t nvec) {       // dst = x + y (y == nullptr: copy)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float8 a = load8(x + i * 8);
        if (y) {
            float8 b = load8(y + i * 8);
            a.lo.x += b.lo.x; a.lo.y += b.lo.y; a.lo.z += b.lo.z; a.lo.w += b.lo.w;
            a.hi.x += b.hi.x; a.hi.y += b.hi.y; a.hi.z += b.hi.z; a.hi.w += b.hi.w;
        }
        store8(dst + i * 8, a);
    }
}

// flip = 0, shift = 1: unet6.py:18-34 ([sin | cos], exponent / (half - 1)); flip = 1, shift = 0: diffusers' `Timesteps(
// flip_sin_to_cos=True, downscale_freq_shift=0)` as UNet2DModel builds it ([cos | sin], exponent / half)
__global__ void temb_kernel(const float* t, int N, int dim, int flip, float shift, float* y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int half = dim / 2;
    if (i >= N * half) return;
    int n = i / half, j = i - n * half;
    float f = expf(-(float)j * (logf(10000.f) / ((float)half - shift)));
    float a = t[n] * f;
    y[n * dim + (flip ? half : 0) + j] = sinf(a);
    y[n * dim + (flip ? 0 : half) + j] = cosf(a);
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

// channels per workgroup: whole groups, whole 16-byte vectors, at least 32 channels
#ifndef MDM_GN_F32_CBLK32
#define MDM_GN_F32_CBLK32 1
#endif
static int gn_cblk(int C, int G, int N, int P) {
    int cpg = C / G, l = cpg;
    while (l % 8) l += cpg;            // lcm(cpg, 8)
    // 32 channels per workgroup; 16 on large maps (more workgroups pulling HBM: 12.3 -> 10.1 us forward,
    // 21.0 -> 15.8 us backward on 32x32x128; slower on the small maps, where the launch floor dominates;
    // re-measured at the end of round 2 on the whole step: 8 channels 3.989, 16 channels 3.958, 32 channels 4.008 ms)
    int want = P > 256 ? 16 : 32;
    int cb = l;
    while (cb < want) cb += l;
    return cb > C ? C : cb;
}
// (A pixel-chunked statistics + apply pair of launches -- MODE 1 / 2 of the register kernels -- was measured no faster
// than the single launch at cfg2: 13.4 vs 13.7 us forward, 25.3 vs 24.1 us backward on 32x32x128; it is not dispatched.)
extern "C" int mdm_groupnorm_fwd(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int P, int G,
                                 float eps, const float* gamma, const float* beta, int silu, void* y, float* stats,
                                 float* ws, void* stream) {
    if (int rc = gn_check(C0, C1, G, N, P)) return rc;
    const int C = C0 + C1;
    int cblk = gn_cblk(C, G, N, P);
    // fp32 forward on large maps: 32 channels per workgroup = whole 128-byte lines per pixel (16 fp32 channels are 64-byte pieces:
    // 3.0 TB/s on the sampler's 32x32x128 maps), as long as the slice still fits the register-cached kernels
    if (dtype == MDM_F32 && P > 256 && MDM_GN_F32_CBLK32) {
        const int wide = gn_cblk(C, G, N, 256);
        if (cdiv(P, 256 / (wide / 8)) <= 16) cblk = wide;
    }
    MDM_REQUIRE(cblk <= 64 && cblk / (C / G) <= 64, "groupnorm: unsupported channel/group combination C=%d G=%d", C, G);
    dim3 grid(N, cdiv(C, cblk));
    const int np = cdiv(P, 256 / (cblk / 8));          // 16-byte vectors per lane of a 256-thread workgroup
    if (dtype == MDM_BF16 && np <= 16) {
#define GN_FWD_REG(NPV, NT) hipLaunchKernelGGL((gn_fwd_reg_kernel<NPV, 0, NT>), grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)src0, C0, \
                                               (const bf16_t*)src1, C1, P, G, cblk, eps, gamma, beta, silu, (bf16_t*)y, stats, ws)
        if (np <= 1) GN_FWD_REG(1, 256); else if (np <= 2) GN_FWD_REG(2, 256); else if (np <= 4) GN_FWD_REG(2, 512);
        else if (np <= 8) GN_FWD_REG(4, 512); else GN_FWD_REG(8, 512);
#undef GN_FWD_REG
        return launch_status("groupnorm_fwd");
    }
    if (dtype == MDM_F32 && np <= 16) {            // the same kernels on fp32 storage: one read of the slice instead of two
#define GN_FWD_REG(NPV, NT) hipLaunchKernelGGL((gn_fwd_reg_kernel<NPV, 0, NT, float>), grid, dim3(NT), 0, (hipStream_t)stream, (const float*)src0, C0, \
                                               (const float*)src1, C1, P, G, cblk, eps, gamma, beta, silu, (float*)y, stats, ws)
        if (np <= 1) GN_FWD_REG(1, 256); else if (np <= 2) GN_FWD_REG(2, 256); else if (np <= 4) GN_FWD_REG(2, 512);
        else if (np <= 8) GN_FWD_REG(4, 512); else GN_FWD_REG(8, 512);
#undef GN_FWD_REG
        return launch_status("groupnorm_fwd");
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((gn_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)src0, C0,
                                         (const T*)src1, C1, P, G, cblk, eps, gamma, beta, silu, (T*)y, stats));
    return launch_status("groupnorm_fwd");
}

// floats of `ws` the backward entry points write for this problem (0: none needed).  The C ABI carries no buffer sizes, so a caller
// sizes its workspace with this query instead of with a constant from the header's prose.
extern "C" int64_t mdm_groupnorm_bwd_ws_floats(int dtype, int N, int C) {
    return dtype == MDM_F32 ? (int64_t)3 * N * C : 0;
}

extern "C" int mdm_groupnorm_bwd_add(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int P, int G,
                                     const float* gamma, const float* beta, int silu, const void* dy, const float* stats,
                                     void* dst0, const void* add0, void* dst1, const void* add1, float* dgamma, float* dbeta,
                                     float* sum_img, int sum_ld, float* sum_all, float* ws, const void* add0b, void* stream) {
    if (int rc = gn_check(C0, C1, G, N, P)) return rc;
    const int C = C0 + C1, cblk = gn_cblk(C, G, N, P);
    MDM_REQUIRE(cblk <= 64 && cblk / (C / G) <= 64, "groupnorm: unsupported channel/group combination C=%d G=%d", C, G);
    MDM_REQUIRE(!(sum_img || sum_all) || C1 == 0, "groupnorm_bwd: column sums need a single-source dx");
    dim3 grid(N, cdiv(C, cblk));
    const int np = cdiv(P, 256 / (cblk / 8));
    if (dtype == MDM_BF16 && np <= 16) {
#define GN_BWD_REG(NPV, NT) hipLaunchKernelGGL((gn_bwd_reg_kernel<NPV, 0, NT>), grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)src0, C0, \
                                               (const bf16_t*)src1, C1, P, G, cblk, gamma, beta, silu, (const bf16_t*)dy, stats,            \
                                               (bf16_t*)dst0, (const bf16_t*)add0, (bf16_t*)dst1, (const bf16_t*)add1, (const bf16_t*)add0b, dgamma, dbeta, \
                                               sum_img, sum_ld, sum_all, ws)
        if (np <= 1) GN_BWD_REG(1, 256); else if (np <= 2) GN_BWD_REG(2, 256); else if (np <= 4) GN_BWD_REG(2, 512);
        else if (np <= 8) GN_BWD_REG(4, 512); else GN_BWD_REG(8, 512);
#undef GN_BWD_REG
        return launch_status("groupnorm_bwd");
    }
    // fp32 path: fixed summation order across the images too (per-image partials in ws + a second stage); bf16 large maps: atomics
    float* part = dtype == MDM_F32 ? ws : nullptr;
    MDM_REQUIRE(dtype != MDM_F32 || ws, "groupnorm_bwd: the fp32 path needs ws (>= 3 * N * C floats) for its fixed-order sums");
    DISPATCH_T(dtype, hipLaunchKernelGGL((gn_bwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)src0, C0,
                                         (const T*)src1, C1, P, G, cblk, gamma, beta, silu, (const T*)dy, stats, (T*)dst0,
                                         (const T*)add0, (T*)dst1, (const T*)add1, (const T*)add0b, dgamma, dbeta, sum_img, sum_ld, sum_all, part));
    if (part)
        hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, part, N, C, dgamma, dbeta, sum_all);
    return launch_status("groupnorm_bwd");
}

extern "C" int mdm_groupnorm_bwd_sums(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int P, int G,
                                      const float* gamma, const float* beta, int silu, const void* dy, const float* stats,
                                      void* dst0, int acc0, void* dst1, int acc1, float* dgamma, float* dbeta,
                                      float* sum_img, int sum_ld, float* sum_all, float* ws, void* stream) {
    MDM_REQUIRE(!(sum_img || sum_all) || acc0 == 0, "groupnorm_bwd_sums: column sums need a plain (non-accumulating) dx");
    return mdm_groupnorm_bwd_add(dtype, src0, C0, src1, C1, N, P, G, gamma, beta, silu, dy, stats, dst0, acc0 ? dst0 : nullptr,
                                 dst1, acc1 ? dst1 : nullptr, dgamma, dbeta, sum_img, sum_ld, sum_all, ws, nullptr, stream);
}

extern "C" int mdm_groupnorm_bwd(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int P, int G,
                                 const float* gamma, const float* beta, int silu, const void* dy, const float* stats,
                                 void* dst0, int acc0, void* dst1, int acc1, float* dgamma, float* dbeta, float* ws,
                                 void* stream) {
    return mdm_groupnorm_bwd_sums(dtype, src0, C0, src1, C1, N, P, G, gamma, beta, silu, dy, stats, dst0, acc0, dst1, acc1,
                                  dgamma, dbeta, nullptr, 0, nullptr, ws, stream);
}

#ifdef MDM_STAMP
extern "C" int mdm_debug_stamps_norm(unsigned long long* out, int reset) {      // out: 4096 * 16 entries
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdm::g_nstamp_buf), 4096 * 16 * 8) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mdm::g_nstamp_buf)) != hipSuccess || hipMemset(p, 0, 4096 * 16 * 8) != hipSuccess) return -1;
    }
    return 0;
}
#endif

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
    hipLaunchKernelGGL(temb_kernel, dim3(cdiv(N * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, N, dim, 0, 1.f, y);
    return launch_status("timestep_embedding");
}
extern "C" int mdm_timestep_embedding2(const float* t, int N, int dim, int flip_sin_to_cos, float freq_shift, float* y, void* stream) {
    MDM_REQUIRE(N > 0 && dim >= 4, "timestep_embedding: bad shape");
    hipLaunchKernelGGL(temb_kernel, dim3(cdiv(N * (dim / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, N, dim, flip_sin_to_cos,
                       freq_shift, y);
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
    const int imgs = dtype == MDM_F32 ? N : 1;          // fp32 path: fixed summation order over the images (see colsum_kernel)
    dim3 grid(cdiv(C, 64), cdiv(N, imgs));
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, pick_stream(stream), (const T*)dY, P, C, per_img, ld, acc_img, dbias, imgs, N));
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
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (T*)dst, (const T*)dst, (const T*)src, n / 8));
    return launch_status("add");
}
extern "C" int mdm_add3(int dtype, void* dst, const void* x, const void* y, int64_t n, void* stream) {
    MDM_REQUIRE(n % 8 == 0 && dst && x, "add3: n must be a multiple of 8");
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (T*)dst, (const T*)x, (const T*)y, n / 8));
    return launch_status("add3");
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
