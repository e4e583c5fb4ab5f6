// Contraction kernels for gfx950: one descriptor (mdm_gemm_desc), three operand
// layouts (NT / NN / TN), optional implicit-im2col gather over NHWC sources.
//
//   bf16 path : 256-thread workgroups, 4 waves as 2x2, v_mfma_f32_16x16x32_bf16,
//               tiles 128x128x64 or 64x64x64 staged through LDS (register-staged
//               global loads so the gather can zero-fill and the rows can be padded
//               against bank conflicts); k-strided operands are kept in their natural
//               [k][col] image and read with ds_read_b64_tr_b16 (hardware transpose).
//   fp32 path : exact-fp32 contraction on v_mfma_f32_16x16x4_f32 (no reduced-precision step),
//               the path of the parity tests and of the reverse sampler that meets 1e-3;
//               shares every index function with the bf16 path.
//
// Replaces F.conv2d fwd/dgrad/wgrad, F.linear and the attention einsums
// (reference unet6.py:170-171, 232-235, 316-324).
#include <stdlib.h>

#include "common.h"
#include <algorithm>
#include <type_traits>
#include <vector>

namespace mdm {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4_t;

#ifndef MDM_USE_TR_READ
#define MDM_USE_TR_READ 1
#endif

// ----------------------------------------------------------------------------
// index helpers shared by both paths
// ----------------------------------------------------------------------------
struct RowPix { int img, oy, ox; };

__device__ __forceinline__ RowPix decode_row(const mdm_gemm_desc& d, int row) {
    RowPix r;
    int per = d.OH * d.OW;
    if (((per & (per - 1)) | (d.OW & (d.OW - 1))) == 0) {      // power-of-two extents (every unet6 level): shifts, no division
        int sp = 31 - __clz(per), sw = 31 - __clz(d.OW);
        r.img = row >> sp;
        int rem = row & (per - 1);
        r.oy = rem >> sw;
        r.ox = rem & (d.OW - 1);
        return r;
    }
    r.img = row / per;
    int rem = row - r.img * per;
    r.oy = rem / d.OW;
    r.ox = rem - r.oy * d.OW;
    return r;
}

// source-pixel index that (row pixel, tap ty/tx) reads, or -1 for zero padding
__device__ __forceinline__ int gather_pix(const mdm_gemm_desc& d, const RowPix& r, int ty, int tx) {
    int iy, ix;
    if (!d.transposed) {
        iy = r.oy * d.stride + ty - d.pad_t;
        ix = r.ox * d.stride + tx - d.pad_l;
        if ((unsigned)iy >= (unsigned)d.IH || (unsigned)ix >= (unsigned)d.IW) return -1;
    } else {
        iy = r.oy + d.pad_t - ty;
        ix = r.ox + d.pad_l - tx;
        if (iy < 0 || ix < 0) return -1;
        if (d.stride == 2) {
            if ((iy | ix) & 1) return -1;
            iy >>= 1; ix >>= 1;
        }
        if (iy >= d.IH || ix >= d.IW) return -1;
    }
    int sh = d.IH >> d.ups, sw = d.IW >> d.ups;
    iy >>= d.ups; ix >>= d.ups;
    return (r.img * sh + iy) * sw + ix;
}
template <typename T>
__device__ __forceinline__ const T* pix_chan_ptr(const mdm_gemm_desc& d, int spix, int c) {
    if (spix < 0) return nullptr;
    if (c < d.C0) return reinterpret_cast<const T*>(d.src0) + (int64_t)spix * d.ld0 + c;
    return reinterpret_cast<const T*>(d.src1) + (int64_t)spix * d.ld1 + (c - d.C0);
}

// pointer to channel c of the source pixel that (row pixel, tap) reads; nullptr = zero padding
template <typename T>
__device__ __forceinline__ const T* gather_ptr(const mdm_gemm_desc& d, const RowPix& r, int tap, int c) {
    int ty = tap / d.KW, tx = tap - ty * d.KW;
    int iy, ix;
    if (!d.transposed) {
        iy = r.oy * d.stride + ty - d.pad_t;
        ix = r.ox * d.stride + tx - d.pad_l;
        if ((unsigned)iy >= (unsigned)d.IH || (unsigned)ix >= (unsigned)d.IW) return nullptr;
    } else {
        iy = r.oy + d.pad_t - ty;
        ix = r.ox + d.pad_l - tx;
        if (iy < 0 || ix < 0) return nullptr;
        if (d.stride == 2) {
            if ((iy | ix) & 1) return nullptr;
            iy >>= 1; ix >>= 1;
        }
        if (iy >= d.IH || ix >= d.IW) return nullptr;
    }
    int sh = d.IH >> d.ups, sw = d.IW >> d.ups;
    iy >>= d.ups; ix >>= d.ups;
    int64_t spix = ((int64_t)r.img * sh + iy) * sw + ix;
    if (c < d.C0) return reinterpret_cast<const T*>(d.src0) + spix * d.ld0 + c;
    return reinterpret_cast<const T*>(d.src1) + spix * d.ld1 + (c - d.C0);
}

struct ZInfo { int batch, tap, kbeg, kend, ks, outer, nouter; };   // nouter: taps / batches the split-K partial slabs are laid out over

// Workgroups are dealt to the 8 XCDs round-robin by linear id; each XCD has its own L2.  This bijection hands
// XCD x the x-th CONTIGUOUS eighth of the work items instead, so items that read the same operand slices
// (the filter taps and output tiles of one k-range; neighbouring pixel tiles and the column tiles of one
// pixel tile) meet in one L2.
// x / dv for a WAVE-UNIFORM x >= 0 and dv in [1, 2^10]: a float reciprocal estimate (one v_rcp, off by at most one for x < 2^22)
// and one correction step, instead of the ~25 dependent scalar instructions of an integer division
__device__ __forceinline__ int udiv_small(int x, int dv) {
    int q = (int)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)dv));
    const int r = x - q * dv;
    q += r >= dv ? 1 : (r < 0 ? -1 : 0);
    return q;
}
// m / rows for a row count that is a power of two wherever a convolution supplies it (rows per image = OH * OW)
__device__ __forceinline__ int div_rows(int m, int rows) {
    return (rows & (rows - 1)) == 0 ? m >> __builtin_ctz(rows) : m / rows;
}

__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7, x = lin & 7, w = lin >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + w;
}

__device__ __forceinline__ ZInfo decode_z(const mdm_gemm_desc& d, int BK) {
    ZInfo z;
    z.batch = 0; z.tap = 0; z.kbeg = 0; z.kend = d.K;
    int zi = blockIdx.z;
    int sk = d.splitk < 1 ? 1 : d.splitk;
    int outer = zi / sk, ks = zi - outer * sk;
    z.ks = ks; z.outer = outer; z.nouter = gridDim.z / sk;
    if (d.layout == 2 && d.conv) z.tap = outer; else z.batch = d.conv ? 0 : outer;
    if (sk > 1) {
        int chunk = ((d.K + sk - 1) / sk + BK - 1) / BK * BK;
        z.kbeg = ks * chunk;
        z.kend = min(d.K, z.kbeg + chunk);
    }
    return z;
}

template <typename T>
__device__ __forceinline__ void epilogue4(const mdm_gemm_desc& d, const ZInfo& z, int m, int n, float4 v) {
    if (d.splitk > 1 && d.ws && d.layout != 2) {      // raw partial; splitk_epilogue_kernel finishes the job
        float* p = reinterpret_cast<float*>(d.ws) + (int64_t)z.ks * ((int64_t)d.M * d.N) + (int64_t)m * d.N + n;
        store4(p, v);
        return;
    }
    v.x *= d.alpha; v.y *= d.alpha; v.z *= d.alpha; v.w *= d.alpha;
    if (d.bias) {
        float4 b = *reinterpret_cast<const float4*>(d.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (d.rowvec) {
        float4 b = *reinterpret_cast<const float4*>(d.rowvec + (int64_t)div_rows(m, d.rows_per_img) * d.rv_ld + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (d.resid) {
        float4 b = load4(reinterpret_cast<const T*>(d.resid) + z.batch * d.sR + (int64_t)m * d.ldr + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    void* base; int ld, col, acc;
    if (n < d.N0) { base = d.D0; ld = d.ldd0; col = n; acc = d.acc0; }
    else          { base = d.D1; ld = d.ldd1; col = n - d.N0; acc = d.acc1; }
    int64_t off = z.batch * d.sD + z.tap * d.dtap + (int64_t)m * ld + col;
    if (d.splitk > 1 && d.ws) {          // partial slab [split][tap|batch][M][N], plain stores; summed by splitk_reduce_kernel
        float* p = reinterpret_cast<float*>(d.ws) + ((int64_t)z.ks * z.nouter + z.outer) * ((int64_t)d.M * d.N) + (int64_t)m * d.N + n;
        store4(p, v);
    } else if (d.out_f32) {        // (a split reduction never lands here: without a workspace resolve() does not split)
        float* p = reinterpret_cast<float*>(base) + off;
        if (acc) { float4 o = load4(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        store4(p, v);
    } else {
        T* p = reinterpret_cast<T*>(base) + off;
        if (acc) { float4 o = load4(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        store4(p, v);
    }
}

// address of VEC consecutive k-elements of A-row `gm` starting at reduction index k (layouts 0/1)
template <typename T>
__device__ __forceinline__ const T* a_row_ptr(const mdm_gemm_desc& d, const ZInfo& z, int gm, const RowPix& rp, int k) {
    if (gm >= d.M || k >= z.kend) return nullptr;
    if (d.conv) {
        int tap = k / d.Ck, c = k - tap * d.Ck;
        return gather_ptr<T>(d, rp, tap, c);
    }
    return reinterpret_cast<const T*>(d.A) + z.batch * d.sA + (int64_t)gm * d.lda + k;
}
// layout 0: VEC consecutive k of B-row gn
template <typename T>
__device__ __forceinline__ const T* b_row_ptr(const mdm_gemm_desc& d, const ZInfo& z, int gn, int k) {
    if (gn >= d.N || k >= z.kend) return nullptr;
    const T* B = reinterpret_cast<const T*>(d.B) + z.batch * d.sB;
    if (d.conv) {
        int tap = k / d.Ck, c = k - tap * d.Ck;
        return B + tap * d.wtap + (int64_t)gn * d.ldb + c;
    }
    return B + (int64_t)gn * d.ldb + k;
}
// layouts 1/2: VEC consecutive n of B k-row `k`
template <typename T>
__device__ __forceinline__ const T* b_col_ptr(const mdm_gemm_desc& d, const ZInfo& z, int k, int gn) {
    if (gn >= d.N || k >= z.kend) return nullptr;
    if (d.layout == 1) {
        const T* B = reinterpret_cast<const T*>(d.B) + z.batch * d.sB;
        if (d.conv) {
            int tap = k / d.Ck, c = k - tap * d.Ck;
            return B + tap * d.wtap + (int64_t)c * d.ldb + gn;
        }
        return B + (int64_t)k * d.ldb + gn;
    }
    if (d.conv) {
        RowPix rp = decode_row(d, k);
        return gather_ptr<T>(d, rp, z.tap, gn);
    }
    return reinterpret_cast<const T*>(d.B) + z.batch * d.sB + (int64_t)k * d.ldb + gn;
}
// layout 2: VEC consecutive m of A k-row `k`
template <typename T>
__device__ __forceinline__ const T* a_col_ptr(const mdm_gemm_desc& d, const ZInfo& z, int k, int gm) {
    if (gm >= d.M || k >= z.kend) return nullptr;
    return reinterpret_cast<const T*>(d.A) + z.batch * d.sA + (int64_t)k * d.lda + gm;
}

// sums the split-K partial slabs into the dense fp32 destination (D = or += sum_s ws[s])
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int splitk, int64_t total4, float* D, int acc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(ws)[i];
        for (int s = 1; s < splitk; ++s) {
            float4 b = reinterpret_cast<const float4*>(ws)[(int64_t)s * total4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float4* q = reinterpret_cast<float4*>(D) + i;
        if (acc) { float4 o = *q; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        *q = a;
    }
}

// The same sum for MANY contractions in one launch (mdm_splitk_reduce_pending): the table travels by value in
// the kernel arguments (a hipGraph keeps its own copy), a workgroup finds its segment by a scan of the
// prefix of workgroup counts.
struct ReduceSeg { const float* ws; float* D; long long total4; int splitk, acc; };
constexpr int REDUCE_MAX_SEGS = 96;
struct ReduceTable { int n; int first_block[REDUCE_MAX_SEGS + 1]; ReduceSeg seg[REDUCE_MAX_SEGS]; };
constexpr int REDUCE_VEC_PER_BLOCK = 1024;          // float4 per workgroup
__global__ __launch_bounds__(256) void splitk_reduce_batched_kernel(ReduceTable tab) {
    int si = 0;
    while (si + 1 < tab.n && (int)blockIdx.x >= tab.first_block[si + 1]) ++si;
    const ReduceSeg sg = tab.seg[si];
    const long long base = (long long)(blockIdx.x - tab.first_block[si]) * REDUCE_VEC_PER_BLOCK;
#pragma unroll
    for (int r = 0; r < REDUCE_VEC_PER_BLOCK / 256; ++r) {
        const long long i = base + r * 256 + threadIdx.x;
        if (i >= sg.total4) break;
        const float4* w = reinterpret_cast<const float4*>(sg.ws) + i;
        float4 a = w[0];
        int s = 1;
        for (; s + 3 < sg.splitk; s += 4) {              // four independent loads in flight (fixed summation order)
            const float4 b0 = w[(long long)s * sg.total4], b1 = w[(long long)(s + 1) * sg.total4];
            const float4 b2 = w[(long long)(s + 2) * sg.total4], b3 = w[(long long)(s + 3) * sg.total4];
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        }
        for (; s < sg.splitk; ++s) {
            const float4 b = w[(long long)s * sg.total4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float4* q = reinterpret_cast<float4*>(sg.D) + i;
        if (sg.acc) { float4 o = *q; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        *q = a;
    }
}

// split-K over the filter taps for the small-M forward / data-gradient convolutions: sums the slabs and
// applies the whole epilogue (scale, bias, time-embedding row, residual, accumulate, channel split, store)
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(mdm_gemm_desc d) {
    const int n4 = d.N / 4;
    const int64_t total4 = (int64_t)d.M * n4;
    ZInfo z; z.batch = 0; z.tap = 0; z.kbeg = 0; z.kend = d.K; z.ks = 0; z.outer = 0; z.nouter = 1;
    mdm_gemm_desc e = d;
    e.splitk = 1;                       // epilogue4 must take its plain-store branch
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4* w = reinterpret_cast<const float4*>(d.ws) + i;
        float4 a = w[0];
        for (int s = 1; s < d.splitk; ++s) {
            float4 b = w[(int64_t)s * total4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        int m = total4 < (1ll << 31) ? (int)((unsigned)i / (unsigned)n4) : (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
        epilogue4<T>(e, z, m, n, a);
    }
}

// ----------------------------------------------------------------------------
// fp32 MFMA path: the same contraction on v_mfma_f32_16x16x4_f32 -- fp32 operands, fp32 accumulate, bit-for-bit a
// k-ordered fmaf chain (no reduced-precision step anywhere), at the matrix pipe's fp32 rate (= the VALU peak, but
// with one VGPR per operand per lane and the VALU free; MI355X_MICROARCH.md: 4096^3 at 122 TF against 52 TF on
// v_pk_fma_f32).  This is the path whose reverse sampler meets north_star's 1e-3 (bf16 storage does not: measured
// 1e-2 after 20 steps, 0.5 after 250), so it must not be slower than it has to be.
//   256 threads = 4 waves as 2 x 2, tile BM x BN x 32, register-staged double-buffered LDS images [row][32 + 4]
//   (k-strided operands are transposed on the way into LDS), fragments by ds_read_b128: a lane's 4 consecutive k
//   feed 4 MFMAs (lane group g supplies k = 4 g + j to MFMA j -- the same permutation on both operands).
// Shares every index function (gather, split-K ranges, epilogue) with the bf16 kernels.  (It replaced a 64x64x16
// VALU kernel -- 4x4 fmaf per thread -- that ran the 1000-step fp32 sampler in 28.2 s; this one takes 17.5 s.)
// ----------------------------------------------------------------------------
// SPLIT (layout 0; mdm_gemm_desc.f32_split): the operands pass through registers on their way into LDS, so each thread splits its own
// float4 there -- x = hi + lo, hi = bf16(x), lo = bf16(x - hi) -- and stores 4 hi halves at byte 2 k and 4 lo halves at byte 64 + 2 k of
// the same 128-byte row; a lane group's fragment is then 8 consecutive k of hi (or lo) and a 32-deep slab is THREE
// v_mfma_f32_16x16x32_bf16 (hi*lo, lo*hi, hi*hi; fp32 accumulate) instead of eight v_mfma_f32_16x16x4_f32: 48 matrix cycles against 256.
template <int BM, int BN, int LAYOUT, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(mdm_gemm_desc d) {
    static_assert(!SPLIT || LAYOUT == 0, "gemm_f32_mfma: split products need both operands k-contiguous");
    constexpr int BK = 32, LD = BK + 4;
    constexpr bool A_ROWS = (LAYOUT != 2), B_ROWS = (LAYOUT == 0);
    constexpr int A_EL = BM * LD, B_EL = BN * LD, STAGE = A_EL + B_EL;
    constexpr int NVA = BM * BK / 4 / 256, NVB = BN * BK / 4 / 256;        // float4 per thread per slab
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    extern __shared__ __attribute__((aligned(16))) float fsm[];              // [2][STAGE]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const ZInfo z = decode_z(d, 16);
    // row-operand mapping: 8 float4 per 32-wide row, 32 rows per pass
    const int rr = t >> 3, rk = (t & 7) * 4;
    // col-operand mapping ([k][cols] in memory): AVR float4 per k-row
    constexpr int AVR = BM / 4, BVR = BN / 4, AKP = 256 / AVR, BKP = 256 / BVR;
    const int ack = t / AVR, acc_ = (t % AVR) * 4;
    const int bck = t / BVR, bcc = (t % BVR) * 4;

    RowPix arow[NVA];
    if (A_ROWS && d.conv) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int gm = m0 + rr + 32 * i;
            arow[i] = decode_row(d, gm < d.M ? gm : 0);
        }
    }
    const int wg_ty = (LAYOUT == 2 && d.conv) ? z.tap / d.KW : 0;
    const int wg_tx = (LAYOUT == 2 && d.conv) ? z.tap - wg_ty * d.KW : 0;
    // tap-major fast path (layouts 0/1, gathered A, unsplit): every 32-slab lies inside one filter tap, so the source pixel
    // of each of this thread's rows is computed once per tap and the slab loop has no integer division
    const bool tapmajor = A_ROWS && d.conv && (d.Ck % BK == 0) && d.splitk <= 1;
    int apix[NVA];
    int cur_tap = -1, nxt_tap = 0, nxt_c = 0;
    float4 ra[NVA], rb[NVB];
    auto ld4 = [](const float* p) { return p ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f); };
    auto load_tiles = [&](int k0) {
        if (tapmajor) {
            const bool live = nxt_tap < d.KH * d.KW;
            if (live && nxt_tap != cur_tap) {
                const int ty = nxt_tap / d.KW, tx = nxt_tap - ty * d.KW;
#pragma unroll
                for (int i = 0; i < NVA; ++i) apix[i] = (m0 + rr + 32 * i < d.M) ? gather_pix(d, arow[i], ty, tx) : -1;
                cur_tap = nxt_tap;
            }
            const int cc = nxt_c + rk;
#pragma unroll
            for (int i = 0; i < NVA; ++i) ra[i] = ld4(live ? pix_chan_ptr<float>(d, apix[i], cc) : nullptr);
            const float* Bt = reinterpret_cast<const float*>(d.B) + (int64_t)nxt_tap * d.wtap;
            if (B_ROWS) {
#pragma unroll
                for (int i = 0; i < NVB; ++i) {
                    const int gn = n0 + rr + 32 * i;
                    rb[i] = ld4((live && gn < d.N) ? Bt + (int64_t)gn * d.ldb + cc : nullptr);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NVB; ++i) {
                    const int gn = n0 + bcc;
                    rb[i] = ld4((live && gn < d.N) ? Bt + (int64_t)(nxt_c + bck + BKP * i) * d.ldb + gn : nullptr);
                }
            }
            nxt_c += BK;
            if (nxt_c >= d.Ck) { nxt_c = 0; ++nxt_tap; }
            return;
        }
        if (A_ROWS) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) ra[i] = ld4(a_row_ptr<float>(d, z, m0 + rr + 32 * i, arow[i], k0 + rk));
        } else {
#pragma unroll
            for (int i = 0; i < NVA; ++i) ra[i] = ld4(a_col_ptr<float>(d, z, k0 + ack + AKP * i, m0 + acc_));
        }
        if (B_ROWS) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) rb[i] = ld4(b_row_ptr<float>(d, z, n0 + rr + 32 * i, k0 + rk));
        } else if (LAYOUT == 2 && d.conv) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int k = k0 + bck + BKP * i, gn = n0 + bcc;
                const float* p = nullptr;
                if (k < z.kend && gn < d.N) p = pix_chan_ptr<float>(d, gather_pix(d, decode_row(d, k), wg_ty, wg_tx), gn);
                rb[i] = ld4(p);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NVB; ++i) rb[i] = ld4(b_col_ptr<float>(d, z, k0 + bck + BKP * i, n0 + bcc));
        }
    };
    auto store_tiles = [&](int buf) {
        float* As = fsm + buf * STAGE;
        float* Bs = As + A_EL;
        if constexpr (SPLIT) {
            auto put = [&](float* row, const float4& v) {          // 4 consecutive k -> 4 hi halves | 4 lo halves
                const bf16_t h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
                uint2 hi, lo;
                hi.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
                hi.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                lo.x = (uint32_t)f2bf(v.x - bf2f(h0)) | ((uint32_t)f2bf(v.y - bf2f(h1)) << 16);
                lo.y = (uint32_t)f2bf(v.z - bf2f(h2)) | ((uint32_t)f2bf(v.w - bf2f(h3)) << 16);
                char* p = reinterpret_cast<char*>(row) + rk * 2;
                *reinterpret_cast<uint2*>(p) = hi;
                *reinterpret_cast<uint2*>(p + 64) = lo;
            };
#pragma unroll
            for (int i = 0; i < NVA; ++i) put(&As[(rr + 32 * i) * LD], ra[i]);
#pragma unroll
            for (int i = 0; i < NVB; ++i) put(&Bs[(rr + 32 * i) * LD], rb[i]);
            return;
        }
        if (A_ROWS) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) *reinterpret_cast<float4*>(&As[(rr + 32 * i) * LD + rk]) = ra[i];
        } else {            // [k][m] in memory -> [m][k] image
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int k = ack + AKP * i;
                As[(acc_ + 0) * LD + k] = ra[i].x; As[(acc_ + 1) * LD + k] = ra[i].y;
                As[(acc_ + 2) * LD + k] = ra[i].z; As[(acc_ + 3) * LD + k] = ra[i].w;
            }
        }
        if (B_ROWS) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) *reinterpret_cast<float4*>(&Bs[(rr + 32 * i) * LD + rk]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const int k = bck + BKP * i;
                Bs[(bcc + 0) * LD + k] = rb[i].x; Bs[(bcc + 1) * LD + k] = rb[i].y;
                Bs[(bcc + 2) * LD + k] = rb[i].z; Bs[(bcc + 3) * LD + k] = rb[i].w;
            }
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = tapmajor ? d.KH * d.KW * (d.Ck / BK) : (z.kend - z.kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_tiles(z.kbeg);
        store_tiles(0);
        if (nk > 1) load_tiles(z.kbeg + BK);
    }
    __syncthreads();
    const int frow = lane & 15, fk = 4 * (lane >> 4);
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < nk) store_tiles(cur ^ 1);
        if (it + 2 < nk) load_tiles(z.kbeg + (it + 2) * BK);
        const float* As = fsm + cur * STAGE;
        const float* Bs = As + A_EL;
        if constexpr (SPLIT) {
            bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const char* p = reinterpret_cast<const char*>(&As[(wr * WM + i * 16 + frow) * LD]) + fk * 4;     // 16 (lane >> 4) bytes
                ah[i] = *reinterpret_cast<const bf16x8*>(p);
                al[i] = *reinterpret_cast<const bf16x8*>(p + 64);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const char* p = reinterpret_cast<const char*>(&Bs[(wc * WN + j * 16 + frow) * LD]) + fk * 4;
                bh[j] = *reinterpret_cast<const bf16x8*>(p);
                bl[j] = *reinterpret_cast<const bf16x8*>(p + 64);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            float4 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(&As[(wr * WM + i * 16 + frow) * LD + kk * 16 + fk]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const float4*>(&Bs[(wc * WN + j * 16 + frow) * LD + kk * 16 + fk]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    // operands swapped like the bf16 kernels: the accumulator holds D[m = lane&15][n = 4*(lane>>4) + reg]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j].x, af[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j].y, af[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j].z, af[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j].w, af[i].w, acc[i][j], 0, 0, 0);
                }
        }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wr * WM + i * 16 + (lane & 15);
        if (m >= d.M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
            if (n < d.N) epilogue4<float>(d, z, m, n, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
        }
    }
    // bias gradient of a linear layer next to its weight gradient (layout 2, no gather): dbias[m] += sum_k A[k][m] over the
    // WHOLE reduction, by the workgroups of the first column tile of the first k-range -- one writer per element and a fixed
    // order (the fp32 path is bit-reproducible); the k-rows are few (the batch) and just went through L2
    if (LAYOUT == 2 && d.dbias && !d.conv && n0 == 0 && blockIdx.z == 0 && t < BM && m0 + t < d.M) {
        float sum = 0.f;
        const float* p = reinterpret_cast<const float*>(d.A) + m0 + t;
        for (int k = 0; k < d.K; ++k) sum += p[(int64_t)k * d.lda];
        d.dbias[m0 + t] += sum;
    }
}

// ----------------------------------------------------------------------------
// fp32 weight gradient of a linear layer whose reduction is the BATCH (layout 2, K <= 128 rows): D[m][n] (=|+=) alpha sum_k A[k][m] B[k][n],
// dbias[m] += sum_k A[k][m].  The time-embedding path's three weight gradients (unet6.py:395-399, 350: [4992|512|512] x [512|512|128] from 32
// rows) took 16 / 11 / 11 us on the 64x64x32 MFMA tiles -- a 32-deep reduction never fills that pipeline; they are 10 / 1 / 0.25 MB of
// output to write.  Here: both operand panels (K x 64 each) parked in LDS once, 4 x 4 outputs per thread, k-ordered fmaf chain (fixed
// order, bit-reproducible), 16-byte stores.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tn_skinny_f32_kernel(mdm_gemm_desc d) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];             // As[K][64] | Bs[K][64]
    const int t = threadIdx.x, K = d.K;
    const int tiles_n = (d.N + 63) >> 6;
    const int m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
    const float* A = reinterpret_cast<const float*>(d.A);
    const float* B = reinterpret_cast<const float*>(d.B);
    float* As = tsm;
    float* Bs = tsm + K * 64;
    for (int i = t; i < K * 16; i += 256) {
        const int k = i >> 4, c4 = (i & 15) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (m0 + c4 < d.M) a = *reinterpret_cast<const float4*>(A + (int64_t)k * d.lda + m0 + c4);
        if (n0 + c4 < d.N) b = *reinterpret_cast<const float4*>(B + (int64_t)k * d.ldb + n0 + c4);
        *reinterpret_cast<float4*>(As + k * 64 + c4) = a;
        *reinterpret_cast<float4*>(Bs + k * 64 + c4) = b;
    }
    __syncthreads();
    const int ty = t >> 4, tx = t & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(As + k * 64 + 4 * ty);
        const float4 b = *reinterpret_cast<const float4*>(Bs + k * 64 + 4 * tx);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    const int n = n0 + 4 * tx;
    if (n < d.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 4 * ty + i;
            if (m >= d.M) continue;
            float4 v = make_float4(acc[i][0] * d.alpha, acc[i][1] * d.alpha, acc[i][2] * d.alpha, acc[i][3] * d.alpha);
            float* q = reinterpret_cast<float*>(d.D0) + (int64_t)m * d.ldd0 + n;
            if (d.acc0) { const float4 o = *reinterpret_cast<const float4*>(q); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(q) = v;
        }
    }
    if (d.dbias && n0 == 0 && t < 64 && m0 + t < d.M) {          // one writer per element, k order
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += As[k * 64 + t];
        d.dbias[m0 + t] += sum;
    }
}

// ----------------------------------------------------------------------------
// Skinny fp32 linear layer D[m][n] = alpha * sum_k A[m][k] B[n][k] + bias[n] with M <= 32 rows (the batch):
// the time-embedding MLP and the per-block projections (unet6.py:395-399, 350).  A 64x64-tiled kernel gives
// 8 workgroups for a [32,512]x[512,512] layer, each pulling 128 KB of weights alone (36 us, and three such
// layers sit in a row at the very start of the forward).  Here a workgroup owns 16 output columns: the
// activation matrix (<= 64 KB) is parked in LDS once, thread (column, k-part) streams its share of the weight
// row in 16-byte pieces, 256 B contiguous per 16 lanes, the 16 k-parts meet in LDS at the end.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_skinny_f32_kernel(mdm_gemm_desc d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // A [32][K] then the partials [32][16][17]
    const int t = threadIdx.x, kp = t & 15, nl = t >> 4;
    const int n0 = blockIdx.x * 16, K = d.K, M = d.M;
    const float* A = reinterpret_cast<const float*>(d.A);
    for (int i = t * 4; i < 32 * K; i += 1024) {
        const int m = i / K, k = i - m * K;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M) v = *reinterpret_cast<const float4*>(A + (int64_t)m * d.lda + k);
        *reinterpret_cast<float4*>(sm + i) = v;
    }
    __syncthreads();
    float acc[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m] = 0.f;
    const int n = n0 + nl;
    const float* Brow = reinterpret_cast<const float*>(d.B) + (int64_t)(n < d.N ? n : 0) * d.ldb;
    for (int k = kp * 4; k < K; k += 64) {
        const float4 b = n < d.N ? *reinterpret_cast<const float4*>(Brow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const float4 a = *reinterpret_cast<const float4*>(sm + m * K + k);
            acc[m] = fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, acc[m]))));
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 32; ++m) sm[(m * 16 + nl) * 17 + kp] = acc[m];
    __syncthreads();
    for (int o = t; o < 32 * 16; o += 256) {
        const int m = o >> 4, c = o & 15;
        if (m >= M || n0 + c >= d.N) continue;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += sm[o * 17 + q];
        v *= d.alpha;
        if (d.bias) v += d.bias[n0 + c];
        reinterpret_cast<float*>(d.D0)[(int64_t)m * d.ldd0 + n0 + c] = v;
    }
}

// ----------------------------------------------------------------------------
// bf16 MFMA path
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint4 ldg16(const bf16_t* p) {
    return p ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
}

// fragment from a k-contiguous image [rows][BK+8]: lane -> row (l&15), k = 8*(l>>4)..+7
template <int LDR>
__device__ __forceinline__ bf16x8 frag_rows(const bf16_t* tile, int row0, int ks, int lane) {
    const bf16_t* p = tile + (row0 + (lane & 15)) * LDR + ks * 32 + 8 * (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(p);
}
// fragment from a natural [k][cols+8] image: element j of lane l is tile[k = ks*32+8*(l>>4)+j][col0 + (l&15)]
template <int LDC>
__device__ __forceinline__ bf16x8 frag_cols(const bf16_t* tile, int col0, int ks, int lane) {
    bf16x8 f;
#if MDM_USE_TR_READ
    const int i = lane & 15;
    const int kb = ks * 32 + 8 * (lane >> 4);
    const bf16_t* p0 = tile + (kb + (i >> 2)) * LDC + col0 + 4 * (i & 3);
    const bf16_t* p1 = p0 + 4 * LDC;
    bf4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(p0));
    bf4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(p1));
    bf16x4 l4 = *reinterpret_cast<bf16x4*>(&lo), h4 = *reinterpret_cast<bf16x4*>(&hi);
    f[0] = l4[0]; f[1] = l4[1]; f[2] = l4[2]; f[3] = l4[3];
    f[4] = h4[0]; f[5] = h4[1]; f[6] = h4[2]; f[7] = h4[3];
#else
    const bf16_t* p = tile + (ks * 32 + 8 * (lane >> 4)) * LDC + col0 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)p[j * LDC];
#endif
    return f;
}

template <int BM, int BN, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(mdm_gemm_desc d) {
    constexpr int BK = 64;
    constexpr bool A_ROWS = (LAYOUT != 2), B_ROWS = (LAYOUT == 0);
    constexpr int LDA = A_ROWS ? BK + 8 : BM + 8;     // row pitch of the A image in elements
    constexpr int LDB = B_ROWS ? BK + 8 : BN + 8;
    constexpr int A_ELEMS = A_ROWS ? BM * LDA : BK * LDA;
    constexpr int B_ELEMS = B_ROWS ? BN * LDB : BK * LDB;
    constexpr int NVA = BM * BK / 8 / 256, NVB = BN * BK / 8 / 256;   // 16-byte vectors per thread
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];      // double-buffered LDS image

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const ZInfo z = decode_z(d, BK);

    // row-operand mapping: 8 vectors per row, 32 rows per pass
    const int rr = t >> 3, rk = (t & 7) * 8;
    // col-operand mapping
    constexpr int AVR = BM / 8, BVR = BN / 8;          // vectors per k-row
    constexpr int AKP = 256 / AVR, BKP = 256 / BVR;    // k-rows per pass
    const int ack = t / AVR, acc_ = (t % AVR) * 8;
    const int bck = t / BVR, bcc = (t % BVR) * 8;

    RowPix arow[NVA];
    if (A_ROWS && d.conv) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            int gm = m0 + rr + 32 * i;
            arow[i] = decode_row(d, gm < d.M ? gm : 0);
        }
    }
    // layout 2 + conv: the filter tap is fixed per workgroup (grid z) -> decode it once
    const int wg_ty = (LAYOUT == 2 && d.conv) ? z.tap / d.KW : 0;
    const int wg_tx = (LAYOUT == 2 && d.conv) ? z.tap - wg_ty * d.KW : 0;
    // Tap-major fast path (layouts 0/1, gathered A): when every BK-slab of the reduction lies inside
    // one filter tap, the source pixel of each of this thread's rows is computed once per tap and the
    // hot loop has no integer division at all.
    const bool tapmajor = A_ROWS && d.conv && (d.Ck % BK == 0);
    int apix[NVA];
    int cur_tap = -1;
    auto set_tap = [&](int tap) {
        int ty = tap / d.KW, tx = tap - ty * d.KW;         // wave-uniform, once per tap
#pragma unroll
        for (int i = 0; i < NVA; ++i) apix[i] = (m0 + rr + 32 * i < d.M) ? gather_pix(d, arow[i], ty, tx) : -1;
        cur_tap = tap;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[NVA], rb[NVB];
    int nxt_tap = 0, nxt_c = 0;       // (tap, channel offset) of the slab load_tiles() fetches next
    auto load_tiles = [&](int k0) {
        if (tapmajor) {
            if (nxt_tap != cur_tap && nxt_tap < d.KH * d.KW) set_tap(nxt_tap);
            const bool live = nxt_tap < d.KH * d.KW;
            const int cc = nxt_c + rk;
#pragma unroll
            for (int i = 0; i < NVA; ++i) ra[i] = ldg16(live ? pix_chan_ptr<bf16_t>(d, apix[i], cc) : nullptr);
            const bf16_t* Bt = reinterpret_cast<const bf16_t*>(d.B) + (int64_t)nxt_tap * d.wtap;
            if (B_ROWS) {
#pragma unroll
                for (int i = 0; i < NVB; ++i) {
                    int gn = n0 + rr + 32 * i;
                    rb[i] = ldg16((live && gn < d.N) ? Bt + (int64_t)gn * d.ldb + cc : nullptr);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NVB; ++i) {
                    int gn = n0 + bcc;
                    rb[i] = ldg16((live && gn < d.N) ? Bt + (int64_t)(nxt_c + bck + BKP * i) * d.ldb + gn : nullptr);
                }
            }
            nxt_c += BK;
            if (nxt_c >= d.Ck) { nxt_c = 0; ++nxt_tap; }
            return;
        }
        if (A_ROWS) {
#pragma unroll
            for (int i = 0; i < NVA; ++i)
                ra[i] = ldg16(a_row_ptr<bf16_t>(d, z, m0 + rr + 32 * i, arow[i], k0 + rk));
        } else {
#pragma unroll
            for (int i = 0; i < NVA; ++i)
                ra[i] = ldg16(a_col_ptr<bf16_t>(d, z, k0 + ack + AKP * i, m0 + acc_));
        }
        if (B_ROWS) {
#pragma unroll
            for (int i = 0; i < NVB; ++i)
                rb[i] = ldg16(b_row_ptr<bf16_t>(d, z, n0 + rr + 32 * i, k0 + rk));
        } else if (LAYOUT == 2 && d.conv) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                int k = k0 + bck + BKP * i, gn = n0 + bcc;
                const bf16_t* p = nullptr;
                if (k < z.kend && gn < d.N) p = pix_chan_ptr<bf16_t>(d, gather_pix(d, decode_row(d, k), wg_ty, wg_tx), gn);
                rb[i] = ldg16(p);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NVB; ++i)
                rb[i] = ldg16(b_col_ptr<bf16_t>(d, z, k0 + bck + BKP * i, n0 + bcc));
        }
    };
    auto store_tiles = [&](int buf) {
        bf16_t* As = smem + buf * STAGE;
        bf16_t* Bs = As + A_ELEMS;
        if (A_ROWS) {
#pragma unroll
            for (int i = 0; i < NVA; ++i) *reinterpret_cast<uint4*>(&As[(rr + 32 * i) * LDA + rk]) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < NVA; ++i) *reinterpret_cast<uint4*>(&As[(ack + AKP * i) * LDA + acc_]) = ra[i];
        }
        if (B_ROWS) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) *reinterpret_cast<uint4*>(&Bs[(rr + 32 * i) * LDB + rk]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < NVB; ++i) *reinterpret_cast<uint4*>(&Bs[(bck + BKP * i) * LDB + bcc]) = rb[i];
        }
    };

    // Software pipeline over k-slabs: registers hold slab it+1 while slab it is read from LDS stage
    // it&1; one barrier per slab.  (store waits for loads issued a whole slab ago.)
    const int nk = tapmajor ? d.KH * d.KW * (d.Ck / BK) : (z.kend - z.kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_tiles(z.kbeg);
        store_tiles(0);
        if (nk > 1) load_tiles(z.kbeg + BK);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < nk) store_tiles(cur ^ 1);
        if (it + 2 < nk) load_tiles(z.kbeg + (it + 2) * BK);
        const bf16_t* As = smem + cur * STAGE;
        const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = A_ROWS ? frag_rows<LDA>(As, wr * WM + i * 16, ks, lane)
                               : frag_cols<LDA>(As, wr * WM + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bfr[j] = B_ROWS ? frag_rows<LDB>(Bs, wc * WN + j * 16, ks, lane)
                                : frag_cols<LDB>(Bs, wc * WN + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    // operands swapped: the accumulator holds D[m = lane&15][n = 4*(lane>>4) + reg],
                    // i.e. four consecutive output channels per lane -> 8/16-byte NHWC stores
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = m0 + wr * WM + i * 16 + (lane & 15);
        if (m >= d.M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
            if (n < d.N) epilogue4<bf16_t>(d, z, m, n, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
        }
    }
}

// ----------------------------------------------------------------------------
// bf16 MFMA path, pipelined: NSTAGE-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4).
//
// The register-staged kernel above keeps one k-slab in flight per workgroup and is bound by
// the memory round trip per slab.  Here every wave issues its share of slab it+NSTAGE-1 straight
// into LDS while slab `it` feeds the MFMAs, waits with a COUNTED vmcnt and meets the other waves
// at ONE raw s_barrier per slab (hipcc would drain vmcnt(0) at a __syncthreads()).
//   * LDS-DMA writes 64 lanes x 16 B contiguously, so images are unpadded and bank conflicts
//     are handled by an XOR swizzle of the 16-byte chunk index applied to the per-lane SOURCE
//     address and again on the fragment read (same involution on both sides);
//   * padding taps, row/col tails and k tails read a zero page instead of being predicated.
// Eligible: every 64-wide k-slab lies in one filter tap (Ck % 64 == 0), or no gather and
// K % 64 == 0, or layout 2 (k = rows).  Everything else runs on the kernel above.
// ----------------------------------------------------------------------------
__device__ uint4 g_zero_page[1024];        // 16 KiB of zeros, the source of every padded 16-byte chunk (conv_lin2 walks inside it)

__device__ __forceinline__ int swz_rows(int row) { return row & 7; }                       // [rows][64] image, 8 chunks/row
template <int CPR> __device__ __forceinline__ int swz_cols(int row) {                      // [64][cols] image
    return CPR == 16 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row ^ (row >> 3)) & 7);
}

__device__ __forceinline__ void lds_dma16(const void* src, char* lds_piece_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_piece_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 ring_frag_rows(const char* tile, int row0, int ks, int lane) {
    int row = row0 + (lane & 15);
    int ch = (ks * 4 + (lane >> 4)) ^ swz_rows(row);
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (ch << 4));
}
template <int CPR>
__device__ __forceinline__ bf16x8 ring_frag_cols(const char* tile, int col0, int ks, int lane) {
    const int i = lane & 15;
    const int kb = ks * 32 + 8 * (lane >> 4) + (i >> 2);
    const int col = col0 + 4 * (i & 3);
    const int ch = col >> 3, half = (col >> 2) & 1;
    const char* p0 = tile + kb * (CPR * 16) + ((ch ^ swz_cols<CPR>(kb)) << 4) + half * 8;
    const char* p1 = tile + (kb + 4) * (CPR * 16) + ((ch ^ swz_cols<CPR>(kb + 4)) << 4) + half * 8;
    bf4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(p0));
    bf4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(p1));
    bf16x4 l4 = *reinterpret_cast<bf16x4*>(&lo), h4 = *reinterpret_cast<bf16x4*>(&hi);
    bf16x8 f;
    f[0] = l4[0]; f[1] = l4[1]; f[2] = l4[2]; f[3] = l4[3];
    f[4] = h4[0]; f[5] = h4[1]; f[6] = h4[2]; f[7] = h4[3];
    return f;
}


// ds_read_b64_tr_b16 through inline asm.  hipcc orders the builtin form behind EVERY in-flight LDS-DMA of the
// wave (it cannot prove the read and the global_load_lds destinations apart and emits s_waitcnt vmcnt(0) in
// front of the first read: the whole prefetch ring collapses to "issue, then wait for it" -- 2100 cycles per
// slab in the weight-gradient kernel against 1380 in the forward kernel that reads with plain ds_read_b128).
// asm loads are invisible to the compiler's counters: the caller retires them with tr_wait() before use.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
struct TrFrag { v2u_t lo, hi; };
template <int CPR>
__device__ __forceinline__ void ring_frag_cols_issue(const char* tile, int col0, int ks, int lane, TrFrag& f) {
    const int i = lane & 15;
    const int kb = ks * 32 + 8 * (lane >> 4) + (i >> 2);
    const int col = col0 + 4 * (i & 3);
    const int ch = col >> 3, half = (col >> 2) & 1;
    const char* p0 = tile + kb * (CPR * 16) + ((ch ^ swz_cols<CPR>(kb)) << 4) + half * 8;
    const char* p1 = tile + (kb + 4) * (CPR * 16) + ((ch ^ swz_cols<CPR>(kb + 4)) << 4) + half * 8;
    const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p0;
    const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p1;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1) : "memory");
}
__device__ __forceinline__ void tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);          // nothing that consumes the fragments may move above the wait
}
__device__ __forceinline__ bf16x8 tr_value(const TrFrag& f) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    v4u_t v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// select without a branch: p if ok else the zero page
__device__ __forceinline__ const void* or_zero(bool ok, const void* p, const void* zero) { return ok ? p : zero; }

#ifndef MDM_RING_PARITY
#define MDM_RING_PARITY 1          // 0: stride-2 data gradients as plain nine-tap gathers (A/B builds)
#endif
template <int BM, int BN, int LAYOUT, int NSTAGE, bool CONV, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_ring_kernel(mdm_gemm_desc d) {
    constexpr int BK = 64;
    constexpr bool A_ROWS = (LAYOUT != 2), B_ROWS = (LAYOUT == 0);
    constexpr bool TAPMAJOR = A_ROWS && CONV;          // reduction index = (tap, channel), slab inside one tap
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int GA = A_BYTES / 1024 / NW, GB = B_BYTES / 1024 / NW, G = GA + GB;   // LDS-DMA pieces per wave per slab
    constexpr int ACPR = BM / 8, BCPR = BN / 8;                                     // chunks per row of a [64][cols] image
    constexpr int A_RPP = 64 / ACPR, B_RPP = 64 / BCPR;                             // k-rows per 1-KiB piece of such an image
    constexpr int WROWS = NW / 2;                                                   // waves as WROWS x 2
    constexpr int WM = BM / WROWS, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    extern __shared__ __attribute__((aligned(1024))) char ring[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const ZInfo z = decode_z(d, BK);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(d.A) + z.batch * d.sA;
    const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(d.B) + z.batch * d.sB;
    const bf16_t* S0 = reinterpret_cast<const bf16_t*>(d.src0);
    const bf16_t* S1 = reinterpret_cast<const bf16_t*>(d.src1);
    // The data gradient of a stride-2 convolution (transposed gather): output pixel (oy, ox) takes tap (ty, tx) only where
    // oy + pad_t - ty and ox + pad_l - tx are EVEN -- one, two, two or four of the nine taps, by the pixel's parity class.  As a plain
    // gather three of four (pixel, tap) pairs read the zero page (12 + 25 + 25 us per step at cfg2 for a quarter of that work).
    // `par`: the rows of the problem are taken in a PERMUTED order, class by class inside every image (class = the top two bits of the
    // in-image index), so that a tile is one class, walks only that class's taps, and the epilogue stores row m at the pixel it stands for.
    const int per_ = d.OH * d.OW;
    const bool par = MDM_RING_PARITY && TAPMAJOR && d.transposed && d.stride == 2 && d.KH == 3 && d.KW == 3 && d.splitk <= 1 &&
                     (per_ & (per_ - 1)) == 0 && (d.OW & (d.OW - 1)) == 0 && d.OW >= 2 && (per_ >> 2) % BM == 0 && d.M % BM == 0;
    const int par_sp = par ? 31 - __clz(per_) : 2, par_c = par_sp - 2, par_h = par ? 30 - __clz(d.OW) : 0;   // log2: pixels, class size, OW / 2
    auto par_pix = [&](int row, RowPix& rp) {                       // permuted row -> (image, oy, ox)
        const int rem = row & (per_ - 1), cls = rem >> par_c, idx = rem & ((per_ >> 2) - 1);
        rp.img = row >> par_sp;
        rp.oy = 2 * (idx >> par_h) + (cls >> 1);
        rp.ox = 2 * (idx & ((d.OW >> 1) - 1)) + (cls & 1);
    };
    const int par_cls = par ? ((m0 & (per_ - 1)) >> par_c) : 0;
    // taps of this tile's class: ty in {ty0, ty0 + 2, ..} (ty = oy + pad_t mod 2), likewise tx
    const int par_ty0 = ((par_cls >> 1) + d.pad_t) & 1, par_tx0 = ((par_cls & 1) + d.pad_l) & 1;
    const int par_nty = par_ty0 ? 1 : 2, par_ntx = par_tx0 ? 1 : 2;

    // per-lane geometry of the pieces this wave fills (fixed for the whole k loop)
    const int r_sub = lane >> 3;                          // rows image: piece p covers rows 8p..8p+7
    const int r_lch = (lane & 7) ^ r_sub;                 // logical chunk (swizzle depends on row&7 = lane>>3 only)
    const int ac_row = lane / ACPR, ac_pch = lane % ACPR; // cols image: lane -> (k-row in piece, physical chunk)
    const int bc_row = lane / BCPR, bc_pch = lane % BCPR;

    // ---- hoisted address generation --------------------------------------------------------
    // Every LDS-DMA piece of this lane has a base pointer that is fixed for a whole segment of the
    // k loop (the whole loop for weights / plain operands, one (tap, source) pair for gathered
    // pixels) plus a WAVE-UNIFORM element offset that advances per slab.  Padding / tails point the
    // base at the zero page and freeze the offset (step 0), so the hot loop is a 64-bit add per piece.
    const char* a_base[GA];
    int a_step[GA];                     // 1: follows the slab offset, 0: zero page
    const char* b_base[GB];
    int b_step[GB];
    int a_img[GA], a_oy[GA], a_ox[GA];
    bool a_ok[GA];
    const int wg_ty = (LAYOUT == 2 && CONV) ? z.tap / d.KW : 0;
    const int wg_tx = (LAYOUT == 2 && CONV) ? z.tap - wg_ty * d.KW : 0;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        a_img[j] = a_oy[j] = a_ox[j] = 0;
        a_base[j] = zero; a_step[j] = 0; a_ok[j] = false;
        const int piece = wave * GA + j;
        if (A_ROWS) {
            const int gm = m0 + 8 * piece + r_sub;
            a_ok[j] = gm < d.M;
            if (TAPMAJOR) {
                RowPix rp = decode_row(d, a_ok[j] ? gm : 0);
                if (par) par_pix(a_ok[j] ? gm : 0, rp);
                a_img[j] = rp.img; a_oy[j] = rp.oy; a_ox[j] = rp.ox;
            } else if (a_ok[j]) {
                a_base[j] = reinterpret_cast<const char*>(Abase + (int64_t)gm * d.lda + 8 * r_lch);
                a_step[j] = 1;
            }
        } else {
            const int kl = piece * A_RPP + ac_row;
            const int gm = m0 + 8 * (ac_pch ^ swz_cols<ACPR>(kl));
            a_ok[j] = gm < d.M;
            a_base[j] = reinterpret_cast<const char*>(Abase + (int64_t)kl * d.lda + gm);   // + k*lda per slab
        }
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        b_base[j] = zero; b_step[j] = 0;
        const int piece = wave * GB + j;
        if (B_ROWS) {
            const int gn = n0 + 8 * piece + r_sub;
            if (gn < d.N) { b_base[j] = reinterpret_cast<const char*>(Bbase + (int64_t)gn * d.ldb + 8 * r_lch); b_step[j] = 1; }
        } else {
            const int kl = piece * B_RPP + bc_row;
            const int gn = n0 + 8 * (bc_pch ^ swz_cols<BCPR>(kl));
            if (gn < d.N) {
                b_step[j] = 1;
                if (LAYOUT == 1) b_base[j] = reinterpret_cast<const char*>(Bbase + (int64_t)kl * d.ldb + gn);
                else if (!CONV) b_base[j] = reinterpret_cast<const char*>(Bbase + (int64_t)kl * d.ldb + gn);
            }
        }
    }
    // slab cursor.  TAPMAJOR: segment = (tap, source); c runs over that source's channels.
    // TAPMAJOR split-K: this workgroup owns the taps [z.ks * tps, (z.ks + 1) * tps)
    const int tps = TAPMAJOR ? (d.KH * d.KW) / (d.splitk < 1 ? 1 : d.splitk) : 0;
    int seg_tap = TAPMAJOR ? z.ks * tps : 0, seg_src = 0, seg_c = 0;
    int tap_ty = TAPMAJOR ? seg_tap / d.KW : 0, tap_tx = TAPMAJOR ? seg_tap - (seg_tap / d.KW) * d.KW : 0;
    if (par) { tap_ty = par_ty0; tap_tx = par_tx0; seg_tap = tap_ty * d.KW + tap_tx; }
    bool seg_dirty = true;
    int nxt_k = z.kbeg;
    const int nsrc = (TAPMAJOR && d.C1 > 0) ? 2 : 1;

#define MDM_RING_ISSUE(SLAB)                                                                                          \
    do {                                                                                                              \
        char* stage = ring + ((SLAB) % NSTAGE) * STAGE_BYTES;                                                         \
        const bool live = (SLAB) < nk;     /* over-issued tail slabs read the zero page only */                      \
        /* ------------------------------------------------ A */                                                      \
        if (TAPMAJOR) {                                                                                               \
            if (seg_dirty) {              /* new (tap, source): re-aim the gathered pixel pointers */                \
                const bf16_t* S = seg_src ? S1 : S0;                                                                  \
                const int ld = seg_src ? d.ld1 : d.ld0;                                                               \
                _Pragma("unroll") for (int j = 0; j < GA; ++j) {                                                      \
                    RowPix rp = {a_img[j], a_oy[j], a_ox[j]};                                                         \
                    const int spix = (a_ok[j] && live) ? gather_pix(d, rp, tap_ty, tap_tx) : -1;                      \
                    a_step[j] = spix >= 0;                                                                            \
                    a_base[j] = spix >= 0 ? reinterpret_cast<const char*>(S + (int64_t)spix * ld + 8 * r_lch) : zero; \
                }                                                                                                     \
                seg_dirty = false;                                                                                    \
            }                                                                                                         \
            const int64_t aoff = (int64_t)seg_c * 2;                                                                  \
            _Pragma("unroll") for (int j = 0; j < GA; ++j)                                                            \
                lds_dma16(a_base[j] + (a_step[j] ? aoff : 0), stage + (wave * GA + j) * 1024);                        \
        } else if (A_ROWS) {                                                                                          \
            const int64_t aoff = live ? (int64_t)nxt_k * 2 : 0;                                                       \
            _Pragma("unroll") for (int j = 0; j < GA; ++j)                                                            \
                lds_dma16((live && a_step[j]) ? a_base[j] + aoff : zero, stage + (wave * GA + j) * 1024);             \
        } else {                                                                                                      \
            const int64_t aoff = (int64_t)nxt_k * d.lda * 2;                                                          \
            _Pragma("unroll") for (int j = 0; j < GA; ++j) {                                                          \
                const int k = nxt_k + (wave * GA + j) * A_RPP + ac_row;                                               \
                lds_dma16((a_ok[j] && k < z.kend) ? a_base[j] + aoff : zero, stage + (wave * GA + j) * 1024);          \
            }                                                                                                         \
        }                                                                                                             \
        /* ------------------------------------------------ B */                                                      \
        char* bst = stage + A_BYTES;                                                                                  \
        if (B_ROWS || LAYOUT == 1) {                                                                                  \
            int64_t boff;                                                                                             \
            if (TAPMAJOR) {                                                                                           \
                const int cg = seg_c + (seg_src ? d.C0 : 0);         /* channel index inside the tap */              \
                boff = B_ROWS ? ((int64_t)seg_tap * d.wtap + cg) * 2 : ((int64_t)seg_tap * d.wtap + (int64_t)cg * d.ldb) * 2; \
            } else {                                                                                                  \
                boff = B_ROWS ? (int64_t)nxt_k * 2 : (int64_t)nxt_k * d.ldb * 2;                                      \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < GB; ++j)                                                            \
                lds_dma16((live && b_step[j]) ? b_base[j] + boff : zero, bst + (wave * GB + j) * 1024);               \
        } else if (CONV) {                /* layout 2: rows of B are gathered pixels of this slab */                 \
            _Pragma("unroll") for (int j = 0; j < GB; ++j) {                                                          \
                const int kl = (wave * GB + j) * B_RPP + bc_row;                                                      \
                const int k = nxt_k + kl;                                                                             \
                const int gn = n0 + 8 * (bc_pch ^ swz_cols<BCPR>(kl));                                                \
                const int spix = gather_pix(d, decode_row(d, k < z.kend ? k : 0), wg_ty, wg_tx);                      \
                const int px = spix < 0 ? 0 : spix;                                                                   \
                const bf16_t* q = gn < d.C0 ? S0 + (int64_t)px * d.ld0 + gn : S1 + (int64_t)px * d.ld1 + (gn - d.C0); \
                lds_dma16(or_zero(b_step[j] && spix >= 0 && k < z.kend, q, zero), bst + (wave * GB + j) * 1024);      \
            }                                                                                                         \
        } else {                                                                                                      \
            const int64_t boff = (int64_t)nxt_k * d.ldb * 2;                                                          \
            _Pragma("unroll") for (int j = 0; j < GB; ++j) {                                                          \
                const int k = nxt_k + (wave * GB + j) * B_RPP + bc_row;                                               \
                lds_dma16((b_step[j] && k < z.kend) ? b_base[j] + boff : zero, bst + (wave * GB + j) * 1024);          \
            }                                                                                                         \
        }                                                                                                             \
        /* ------------------------------------------------ advance */                                                \
        if (TAPMAJOR) {                                                                                               \
            seg_c += BK;                                                                                              \
            if (seg_c >= (seg_src ? d.C1 : d.C0)) {                                                                   \
                seg_c = 0; seg_dirty = true;                                                                          \
                if (++seg_src == nsrc) {                                                                              \
                    seg_src = 0;                                                                                      \
                    if (par) {                 /* the next tap of this parity class */                               \
                        tap_tx += 2;                                                                                  \
                        if (tap_tx >= d.KW) { tap_tx = par_tx0; tap_ty += 2; }                                        \
                        seg_tap = tap_ty * d.KW + tap_tx;                                                             \
                    } else {                                                                                          \
                        ++seg_tap;                                                                                    \
                        if (++tap_tx == d.KW) { tap_tx = 0; ++tap_ty; }                                               \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
        } else {                                                                                                      \
            nxt_k += BK;                                                                                              \
        }                                                                                                             \
    } while (0)

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient fused into the weight gradient: column sums of the A tile (= dY) as one more MFMA
    const bool do_bias = LAYOUT == 2 && d.dbias != nullptr && z.outer == 0 && n0 == 0 && wc == 0;
    f32x4 accb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;      // bf16 1.0

    const int nk = TAPMAJOR ? (par ? par_nty * par_ntx : tps) * (d.Ck / BK) : (z.kend - z.kbeg + BK - 1) / BK;
    // prologue: NSTAGE-1 slabs in flight.  Slabs past the end are issued too (all-zero or harmless
    // re-reads) so that the vmcnt bookkeeping below is the same for every trip count.
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) MDM_RING_ISSUE(s);

    for (int it = 0; it < nk; ++it) {
        wait_vmcnt<(NSTAGE - 2) * G>();        // slab `it` of THIS wave has landed ...
        __builtin_amdgcn_s_barrier();          // ... and of every wave; slab it-1 is fully consumed
        MDM_RING_ISSUE(it + NSTAGE - 1);       // refill the stage that slab it-1 occupied
        const char* As = ring + (it % NSTAGE) * STAGE_BYTES;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[MI], bfr[NI];
            TrFrag ta[MI], tb[NI];          // k-strided operands: transposing reads through asm (see ring_frag_cols_issue)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (A_ROWS) af[i] = ring_frag_rows(As, wr * WM + i * 16, ks, lane);
                else ring_frag_cols_issue<ACPR>(As, wr * WM + i * 16, ks, lane, ta[i]);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (B_ROWS) bfr[j] = ring_frag_rows(Bs, wc * WN + j * 16, ks, lane);
                else ring_frag_cols_issue<BCPR>(Bs, wc * WN + j * 16, ks, lane, tb[j]);
            }
            if (!A_ROWS || !B_ROWS) {
                tr_wait();
#pragma unroll
                for (int i = 0; i < MI; ++i) if (!A_ROWS) af[i] = tr_value(ta[i]);
#pragma unroll
                for (int j = 0; j < NI; ++j) if (!B_ROWS) bfr[j] = tr_value(tb[j]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if (LAYOUT == 2 && do_bias) {
#pragma unroll
                for (int i = 0; i < MI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);
            }
        }
    }
    wait_vmcnt<0>();                           // the over-issued tail slabs must land before the workgroup retires
    if (LAYOUT == 2 && do_bias && (lane >> 4) == 0) {          // every column of ones x A^T holds sum_k A[k][m]
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + wr * WM + i * 16 + (lane & 15);
            if (m < d.M) atomicAdd(&d.dbias[m], accb[i][0]);
        }
    }
#undef MDM_RING_ISSUE
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = m0 + wr * WM + i * 16 + (lane & 15);
        if (m >= d.M) continue;
        if (par) { RowPix rp; par_pix(m, rp); m = (rp.img * d.OH + rp.oy) * d.OW + rp.ox; }      // the pixel this row stands for
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
            if (n < d.N) epilogue4<bf16_t>(d, z, m, n, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
        }
    }
}

// ----------------------------------------------------------------------------
// conv_lin2: ring kernel for the bulk of the convolutions -- layout 0 (both operands k-contiguous), stride 1 (or the
// stride-2 forward), no folded upsample: every 3x3 / 1x1 forward and (through the transposed weight shadow) every such
// data gradient.  There the source pixel is LINEAR in the filter tap, spix(tap) = base0 + sgn * (ty * IW + tx), so a lane
// keeps one row pointer per source and a 9-bit validity mask per row.  The ISSUE side is reduced to its minimum:
// the first version of this kernel spent ~120 scalar + ~70 vector instructions per k-slab per wave around
// 16 MFMAs (tap decode, selects against the zero page, M0 through readfirstlane); with both waves of a
// SIMD in lockstep behind the slab barrier that instruction stream, not MFMA/LDS/L2, sets the slab time.
// Here every LDS-DMA piece owns ONE 64-bit pointer that is re-aimed only when the (tap, source) segment
// changes and otherwise just advances by 128 bytes per slab; invalid taps / tails point into a 16-KiB
// zero page and walk inside it, so the slab loop has no select at all.  The wave index is made provably
// uniform (readfirstlane once) so LDS destinations live in SGPRs.  Waves are laid out WR x WC x WK:
// WK = 2 splits the two 32-wide k-steps of a slab between wave pairs (64x64 wave tiles: 1.5x fewer LDS
// fragment reads than 32x64), the pair's accumulators meet in LDS once at the end.
// ----------------------------------------------------------------------------

// ----------------------------------------------------------------------------
// Tile epilogue through LDS (bf16 destinations).  The MFMA accumulator layout gives a lane 4 channels
// of 16 different pixel rows: stored directly that is 8-byte pieces scattered over 16 rows per
// instruction (measured: 4.7 us of a 21 us kernel for a 128x128 tile).  Here the fp32 tile (after scale,
// bias and time-embedding row) is parked in the idle ring, then every thread owns 8 consecutive channels
// of one pixel: residual, accumulate and the store are 16-byte accesses, 256 B contiguous per 16 lanes.
// Rounding happens once, after every fp32 term is in -- same arithmetic as epilogue4.
// ----------------------------------------------------------------------------
// Tile stores of a convolution that runs as a phase of a CHAIN (chain_kernel instantiates the bodies with WT): WRITE-THROUGH (sc1), so
// that the consumer workgroup of the next phase finds the bytes in memory once this workgroup's waves have drained their stores --
// no agent-scope release (an L2 write-back: 6-8 us per hand-off with plain stores, MI355X_MICROARCH.md "publish-large").  hipcc does
// not count an asm store (the chain drains with an explicit vmcnt(0)); `s_nop 1` keeps the data registers until the store has read them.
typedef unsigned st4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(void* p, const st4u_t& r) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
template <bool WT>
__device__ __forceinline__ void store8_pub(bf16_t* p, const float8& v) {
    if constexpr (WT) {
        st4u_t r;
        r[0] = (uint32_t)f2bf(v.lo.x) | ((uint32_t)f2bf(v.lo.y) << 16);
        r[1] = (uint32_t)f2bf(v.lo.z) | ((uint32_t)f2bf(v.lo.w) << 16);
        r[2] = (uint32_t)f2bf(v.hi.x) | ((uint32_t)f2bf(v.hi.y) << 16);
        r[3] = (uint32_t)f2bf(v.hi.z) | ((uint32_t)f2bf(v.hi.w) << 16);
        store16_wt(p, r);
    } else store8(p, v);
}
template <bool WT>
__device__ __forceinline__ void store8_pub(float* p, const float8& v) {
    if constexpr (WT) {
        store16_wt(p, __builtin_bit_cast(st4u_t, v.lo));
        store16_wt(p + 4, __builtin_bit_cast(st4u_t, v.hi));
    } else store8(p, v);
}

template <int BM, int BN, int NW, int MI, int NI, typename T = bf16_t, bool WT = false>
__device__ __forceinline__ void epilogue_tile(const mdm_gemm_desc& d, char* lds, int m0, int n0, int row_w, int col_w,
                                              int lane, int t, f32x4 (&acc)[MI][NI]) {
    constexpr int PITCH = BN * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = row_w + i * 16 + (lane & 15), m = m0 + ml;
        const float* rv = (d.rowvec && m < d.M) ? d.rowvec + (int64_t)div_rows(m, d.rows_per_img) * d.rv_ld : nullptr;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = col_w + j * 16 + 4 * (lane >> 4), n = n0 + nl;
            float4 v = make_float4(acc[i][j][0] * d.alpha, acc[i][j][1] * d.alpha, acc[i][j][2] * d.alpha, acc[i][j][3] * d.alpha);
            if (n < d.N) {
                if (d.bias) { float4 b = *reinterpret_cast<const float4*>(d.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (rv) { float4 b = *reinterpret_cast<const float4*>(rv + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            }
            *reinterpret_cast<float4*>(lds + ml * PITCH + (((nl >> 2) ^ (ml & 7)) << 4)) = v;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll 2
    for (int idx = t; idx < BM * CPR; idx += 64 * NW) {
        const int r = idx / CPR, q = idx - r * CPR;
        const int m = m0 + r, n = n0 + q * 8;
        if (m >= d.M || n >= d.N) continue;
        const float4 lo = *reinterpret_cast<const float4*>(lds + r * PITCH + (((2 * q) ^ (r & 7)) << 4));
        const float4 hi = *reinterpret_cast<const float4*>(lds + r * PITCH + (((2 * q + 1) ^ (r & 7)) << 4));
        float8 v = {lo, hi};
        if (d.resid) {
            float8 b = load8(reinterpret_cast<const T*>(d.resid) + (int64_t)m * d.ldr + n);
            v.lo.x += b.lo.x; v.lo.y += b.lo.y; v.lo.z += b.lo.z; v.lo.w += b.lo.w;
            v.hi.x += b.hi.x; v.hi.y += b.hi.y; v.hi.z += b.hi.z; v.hi.w += b.hi.w;
        }
        T* p; int accf;
        if (n < d.N0) { p = reinterpret_cast<T*>(d.D0) + (int64_t)m * d.ldd0 + n; accf = d.acc0; }
        else          { p = reinterpret_cast<T*>(d.D1) + (int64_t)m * d.ldd1 + (n - d.N0); accf = d.acc1; }
        if (accf) {
            float8 o = load8(p);
            v.lo.x += o.lo.x; v.lo.y += o.lo.y; v.lo.z += o.lo.z; v.lo.w += o.lo.w;
            v.hi.x += o.hi.x; v.hi.y += o.hi.y; v.hi.z += o.hi.z; v.hi.w += o.hi.w;
        }
        store8_pub<WT>(p, v);
    }
}


// fp32 partial tile of a split-K contraction -> its slab, through LDS: 16-byte stores, 512 B contiguous per 32 lanes
// (the MFMA layout stores 64-B pieces over 16 rows per instruction)
template <int BM, int BN, int NW, int MI, int NI>
__device__ __forceinline__ void epilogue_tile_slab(const mdm_gemm_desc& d, float* slab, char* lds, int m0, int n0,
                                                   int row_w, int col_w, int lane, int t, f32x4 (&acc)[MI][NI]) {
    constexpr int PITCH = BN * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = row_w + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = col_w + j * 16 + 4 * (lane >> 4);
            *reinterpret_cast<float4*>(lds + ml * PITCH + (((nl >> 2) ^ (ml & 7)) << 4)) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 4;
#pragma unroll 4
    for (int idx = t; idx < BM * CPR; idx += 64 * NW) {
        const int r = idx / CPR, q = idx - r * CPR;
        const int m = m0 + r, n = n0 + q * 4;
        if (m >= d.M || n >= d.N) continue;
        const float4 v = *reinterpret_cast<const float4*>(lds + r * PITCH + ((q ^ (r & 7)) << 4));
        *reinterpret_cast<float4*>(slab + (int64_t)m * d.N + n) = v;
    }
}


// ----------------------------------------------------------------------------
// GroupNorm backward fused into the data-gradient epilogue of the conv that consumes the normalised tensor, for the
// 64-pixel x BN-channel halo tiles of whole images (4x4 / 8x8 maps; BN = 64 or 32): the tile IS complete (image, group)
// blocks, so the group sums stay inside the workgroup and d(z) never goes to memory.  64 pixels x NCH = BN / 8 channel
// chunks = 512 or 256 active threads.  Same arithmetic as gn_bwd_reg_kernel (norm.hip); d(z) enters in fp32 instead of bf16.
// LDS: [0, 16K) the fp32 tile, [16K, 80K) column-sum scratch [32 quantities][512], then small arrays.
// ----------------------------------------------------------------------------
template <int MI, int NI, int BN, bool WT = false>
__device__ __forceinline__ void epilogue_tile_gnb(const mdm_gemm_desc& d, char* lds, int m0, int n0, int row_w, int col_w,
                                                  int lane, int t, f32x4 (&acc)[MI][NI]) {
    constexpr int PITCH = BN * 4, NCH = BN / 8, NCH_SH = NCH == 8 ? 3 : 2, ACTIVE = 64 * NCH;
    static_assert(BN == 64 || BN == 32, "fused GroupNorm epilogues: 64- or 32-channel tiles");
    float* scratch = reinterpret_cast<float*>(lds + 16384);               // [32][512]
    float* psum = reinterpret_cast<float*>(lds + 16384 + 65536);          // [32 NCH columns][4 parts]
    float* gs = psum + 1024;                                              // [4 images][16 groups][2]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = row_w + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = col_w + j * 16 + 4 * (lane >> 4);
            *reinterpret_cast<float4*>(lds + ml * PITCH + (((nl >> 2) ^ (ml & 7)) << 4)) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
    if (t < 128) gs[t] = 0.f;
    __syncthreads();
    // P = 16 or 64 pixels per image, cpg = 4 .. BN channels per group: powers of two (mdm_gemm_can_fuse_gn_bwd), so the
    // per-element group index is a shift (it was 16 integer divisions per thread)
    const int C = d.N, G = d.gnb_G, P = d.OH * d.OW, p_sh = __builtin_ctz(P), cpg_sh = __builtin_ctz(C) - __builtin_ctz(G), cpg = 1 << cpg_sh;
    const bool on = t < ACTIVE;
    const int r = (t >> NCH_SH) & 63, q = t & (NCH - 1), il = r >> p_sh;
    const int m = m0 + r, n = n0 + q * 8, img = (m0 >> p_sh) + il;
    float rstd[8], xh[8], gz[8], ga[8];
    if (on) {
        const float4 lo = *reinterpret_cast<const float4*>(lds + r * PITCH + (((2 * q) ^ (r & 7)) << 4));
        const float4 hi = *reinterpret_cast<const float4*>(lds + r * PITCH + (((2 * q + 1) ^ (r & 7)) << 4));
        const float dz[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float8 x8 = load8(reinterpret_cast<const bf16_t*>(d.gnb_x) + (int64_t)m * C + n);
        const float xv[8] = {x8.lo.x, x8.lo.y, x8.lo.z, x8.lo.w, x8.hi.x, x8.hi.y, x8.hi.z, x8.hi.w};
        const float4 g_lo = *reinterpret_cast<const float4*>(d.gnb_gamma + n), g_hi = *reinterpret_cast<const float4*>(d.gnb_gamma + n + 4);
        const float4 b_lo = *reinterpret_cast<const float4*>(d.gnb_beta + n), b_hi = *reinterpret_cast<const float4*>(d.gnb_beta + n + 4);
        const float gav[8] = {g_lo.x, g_lo.y, g_lo.z, g_lo.w, g_hi.x, g_hi.y, g_hi.z, g_hi.w};
        const float be[8] = {b_lo.x, b_lo.y, b_lo.z, b_lo.w, b_hi.x, b_hi.y, b_hi.z, b_hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float2 st = *reinterpret_cast<const float2*>(d.gnb_stats + ((int64_t)img * G + ((n + e) >> cpg_sh)) * 2);
            ga[e] = gav[e];
            rstd[e] = st.y;
            xh[e] = (xv[e] - st.x) * st.y;
            gz[e] = dz[e];
            if (d.gnb_silu) gz[e] *= silu_grad_f(fmaf(xh[e], ga[e], be[e]));
            scratch[e * 512 + t] = gz[e] * ga[e];                             // a1: sum over the group of dy*gamma
            scratch[(8 + e) * 512 + t] = gz[e] * ga[e] * xh[e];               // a2
            scratch[(16 + e) * 512 + t] = gz[e] * xh[e];                      // dgamma
            scratch[(24 + e) * 512 + t] = gz[e];                              // dbeta
        }
    }
    __syncthreads();
    // column sums over the pixels, in 4 parts of 16 rows (a part never straddles an image: P is 16 or 64);
    // column = quantity k (0..31) x chunk: the values of one column sit NCH words apart in row k of the scratch
#pragma unroll
    for (int it2 = 0; it2 < (32 * NCH * 4) / 512; ++it2) {
        const int id = t + 512 * it2, col = id >> 2, part = id & 3;       // col = k * NCH + chunk
        const float* src = scratch + (col >> NCH_SH) * 512 + (part * 16) * NCH + (col & (NCH - 1));
        float sum = 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sum += src[rr * NCH];
        psum[id] = sum;
    }
    __syncthreads();
    if (t < 2 * BN) {                    // dgamma / dbeta: one atomic per channel per workgroup
        const int which = t >= BN ? 1 : 0, c = t - which * BN, col = (16 + 8 * which + (c & 7)) * NCH + (c >> 3);
        const float v = psum[col * 4] + psum[col * 4 + 1] + psum[col * 4 + 2] + psum[col * 4 + 3];
        atomicAdd(which ? d.gnb_dbeta + n0 + c : d.gnb_dgamma + n0 + c, v);
    }
    const int ppi = P >> 4, nimg = 64 >> p_sh;          // parts per image (1 or 4), images per tile
    {                                    // group sums per image: thread (image il2, group gl, w) walks its channels in order
        const int ngt_sh = (BN == 64 ? 6 : 5) - cpg_sh, ngt = 1 << ngt_sh;          // groups inside this BN-channel tile
        if (t < nimg * ngt * 2) {        // (it was an LDS float atomic per channel: sums in arrival order are not reproducible)
            const int w = t & 1, gl = (t >> 1) & (ngt - 1), il2 = t >> (1 + ngt_sh);
            float v = 0.f;
            for (int j = 0; j < cpg; ++j) {
                const int c = (gl << cpg_sh) + j, col = (8 * w + (c & 7)) * NCH + (c >> 3);
                for (int pp = 0; pp < ppi; ++pp) v += psum[col * 4 + il2 * ppi + pp];
            }
            gs[(il2 * 16 + gl) * 2 + w] = v;
        }
    }
    __syncthreads();
    const float inv_cnt = 1.f / ((float)cpg * (float)P);
    float o[8];
    if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int gl = (q * 8 + e) >> cpg_sh;
            const float k1 = rstd[e] * gs[(il * 16 + gl) * 2] * inv_cnt, k2 = rstd[e] * gs[(il * 16 + gl) * 2 + 1] * inv_cnt;
            o[e] = rstd[e] * ga[e] * gz[e] - fmaf(xh[e], k2, k1);
        }
        if (d.gnb_sum_img || d.gnb_sum_all) {                                  // uniform: column sums of dx (before accumulation)
#pragma unroll
            for (int e = 0; e < 8; ++e) scratch[e * 512 + t] = o[e];
        }
        bf16_t* p = reinterpret_cast<bf16_t*>(d.D0) + (int64_t)m * d.ldd0 + n;
        float8 v = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
        if (d.acc0) {                      // accumulate in place
            const float8 old = load8(p);
            v.lo.x += old.lo.x; v.lo.y += old.lo.y; v.lo.z += old.lo.z; v.lo.w += old.lo.w;
            v.hi.x += old.hi.x; v.hi.y += old.hi.y; v.hi.z += old.hi.z; v.hi.w += old.hi.w;
        }
        if (d.gnb_add) {                   // and / or add a tensor laid out like D0 (the residual branch's gradient)
            const float8 old = load8(reinterpret_cast<const bf16_t*>(d.gnb_add) + (int64_t)m * d.ldd0 + n);
            v.lo.x += old.lo.x; v.lo.y += old.lo.y; v.lo.z += old.lo.z; v.lo.w += old.lo.w;
            v.hi.x += old.hi.x; v.hi.y += old.hi.y; v.hi.z += old.hi.z; v.hi.w += old.hi.w;
        }
        store8_pub<WT>(p, v);
    }
    if (d.gnb_sum_img || d.gnb_sum_all) {
        __syncthreads();
        if (t < 8 * NCH * 4) {
            const int col = t >> 2, part = t & 3;                          // col = e * NCH + chunk
            const float* src = scratch + (col >> NCH_SH) * 512 + (part * 16) * NCH + (col & (NCH - 1));
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sum += src[rr * NCH];
            psum[t] = sum;
        }
        __syncthreads();
        if (t < nimg * BN) {
            const int il2 = t / BN, c = t - il2 * BN, col = (c & 7) * NCH + (c >> 3);
            float v = 0.f;
            for (int pp = 0; pp < ppi; ++pp) v += psum[col * 4 + il2 * ppi + pp];
            if (d.gnb_sum_img) d.gnb_sum_img[(int64_t)((m0 >> p_sh) + il2) * d.gnb_sum_ld + n0 + c] = v;      // this workgroup owns (image, channel)
            if (d.gnb_sum_all) atomicAdd(d.gnb_sum_all + n0 + c, v);
        }
    }
}


// ----------------------------------------------------------------------------
// GroupNorm forward fused into the epilogue of the conv that PRODUCES the tensor (same 64-pixel x BN-channel tiles of
// whole images): the usual epilogue (scale, bias, time-embedding row, residual, bf16 rounding, store y), then
// statistics over the ROUNDED values exactly as a separate GroupNorm launch would see them -- two passes (mean, then
// centred squares: exact for constant maps like the pivot-shifted sums of norm.hip) -- and z = silu?(y_hat*gamma+beta).
// ----------------------------------------------------------------------------
template <int MI, int NI, int BN, bool WT = false>
__device__ __forceinline__ void epilogue_tile_gnf(const mdm_gemm_desc& d, char* lds, int m0, int n0, int row_w, int col_w,
                                                  int lane, int t, f32x4 (&acc)[MI][NI]) {
    constexpr int PITCH = BN * 4, NCH = BN / 8, NCH_SH = NCH == 8 ? 3 : 2, ACTIVE = 64 * NCH;
    static_assert(BN == 64 || BN == 32, "fused GroupNorm epilogues: 64- or 32-channel tiles");
    float* scratch = reinterpret_cast<float*>(lds + 16384);               // [8][512]
    float* psum = reinterpret_cast<float*>(lds + 16384 + 16384);          // [8 NCH columns][4 parts]
    float* gs = psum + 256;                                               // [4 images][16 groups]: sum, then centred squares
    float* gm = gs + 64;                                                  // mean
    float* gr = gm + 64;                                                  // rstd
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = row_w + i * 16 + (lane & 15), mrow = m0 + ml;
        const float* rv = (d.rowvec && mrow < d.M) ? d.rowvec + (int64_t)div_rows(mrow, d.rows_per_img) * d.rv_ld : nullptr;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = col_w + j * 16 + 4 * (lane >> 4), ncol = n0 + nl;
            float4 v = make_float4(acc[i][j][0] * d.alpha, acc[i][j][1] * d.alpha, acc[i][j][2] * d.alpha, acc[i][j][3] * d.alpha);
            if (d.bias) { float4 b = *reinterpret_cast<const float4*>(d.bias + ncol); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (rv) { float4 b = *reinterpret_cast<const float4*>(rv + ncol); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            *reinterpret_cast<float4*>(lds + ml * PITCH + (((nl >> 2) ^ (ml & 7)) << 4)) = v;
        }
    }
    if (t < 64) gs[t] = 0.f;
    __syncthreads();
    const int C = d.N, G = d.gnf_G, P = d.OH * d.OW, p_sh = __builtin_ctz(P), cpg_sh = __builtin_ctz(C) - __builtin_ctz(G), cpg = 1 << cpg_sh;
    const bool on = t < ACTIVE;
    const int r = (t >> NCH_SH) & 63, q = t & (NCH - 1), il = r >> p_sh;     // powers of two (mdm_gemm_can_fuse_gn_fwd): shifts, not divisions
    const int m = m0 + r, n = n0 + q * 8;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = 0.f;
    if (on) {
        const float4 lo = *reinterpret_cast<const float4*>(lds + r * PITCH + (((2 * q) ^ (r & 7)) << 4));
        const float4 hi = *reinterpret_cast<const float4*>(lds + r * PITCH + (((2 * q + 1) ^ (r & 7)) << 4));
        float8 v = {lo, hi};
        if (d.resid) {
            const float8 b = load8(reinterpret_cast<const bf16_t*>(d.resid) + (int64_t)m * d.ldr + n);
            v.lo.x += b.lo.x; v.lo.y += b.lo.y; v.lo.z += b.lo.z; v.lo.w += b.lo.w;
            v.hi.x += b.hi.x; v.hi.y += b.hi.y; v.hi.z += b.hi.z; v.hi.w += b.hi.w;
        }
        bf16_t* yp = reinterpret_cast<bf16_t*>(d.D0) + (int64_t)m * d.ldd0 + n;
        store8_pub<WT>(yp, v);
        // the values as GroupNorm reads them back: rounded to bf16
        const float vv[8] = {v.lo.x, v.lo.y, v.lo.z, v.lo.w, v.hi.x, v.hi.y, v.hi.z, v.hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = bf2f(f2bf(vv[e]));
    }
    const int ppi = P >> 4, nimg = 64 >> p_sh;
    const float inv_cnt = 1.f / ((float)cpg * (float)P);
    float mean[8], rstd[8];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (on) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c0 = pass ? y[e] - mean[e] : y[e]; scratch[e * 512 + t] = pass ? c0 * c0 : c0; }
        }
        __syncthreads();
        if (t < 8 * NCH * 4) {
            const int col = t >> 2, part = t & 3;                          // col = e * NCH + chunk
            const float* src = scratch + (col >> NCH_SH) * 512 + (part * 16) * NCH + (col & (NCH - 1));
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sum += src[rr * NCH];
            psum[t] = sum;
        }
        __syncthreads();
        {                                // thread (image il2, group gl) walks its channels in order (fixed summation order)
            const int ngt_sh = (BN == 64 ? 6 : 5) - cpg_sh, ngt2 = 1 << ngt_sh;
            if (t < nimg * ngt2) {
                const int gl = t & (ngt2 - 1), il2 = t >> ngt_sh;
                float s2 = 0.f;
                for (int j = 0; j < cpg; ++j) {
                    const int c = (gl << cpg_sh) + j, col = (c & 7) * NCH + (c >> 3);
                    for (int pp = 0; pp < ppi; ++pp) s2 += psum[col * 4 + il2 * ppi + pp];
                }
                gs[il2 * 16 + gl] = s2;
            }
        }
        __syncthreads();
        if (t < 64) {
            if (pass == 0) gm[t] = gs[t] * inv_cnt; else gr[t] = rsqrtf(gs[t] * inv_cnt + d.gnf_eps);
        }
        __syncthreads();
        if (pass == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mean[e] = gm[il * 16 + ((q * 8 + e) >> cpg_sh)];
            if (t < 64) gs[t] = 0.f;
            __syncthreads();
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) rstd[e] = gr[il * 16 + ((q * 8 + e) >> cpg_sh)];
        }
    }
    if (on) {
        const float4 g_lo = *reinterpret_cast<const float4*>(d.gnf_gamma + n), g_hi = *reinterpret_cast<const float4*>(d.gnf_gamma + n + 4);
        const float4 b_lo = *reinterpret_cast<const float4*>(d.gnf_beta + n), b_hi = *reinterpret_cast<const float4*>(d.gnf_beta + n + 4);
        const float ga[8] = {g_lo.x, g_lo.y, g_lo.z, g_lo.w, g_hi.x, g_hi.y, g_hi.z, g_hi.w};
        const float be[8] = {b_lo.x, b_lo.y, b_lo.z, b_lo.w, b_hi.x, b_hi.y, b_hi.z, b_hi.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = fmaf((y[e] - mean[e]) * rstd[e], ga[e], be[e]);
            if (d.gnf_silu) o[e] = silu_f(o[e]);
        }
        const float8 zo = {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
        store8_pub<WT>(reinterpret_cast<bf16_t*>(d.gnf_out) + (int64_t)m * C + n, zo);
    }
    const int ngt = BN >> cpg_sh;                                          // groups inside this BN-channel tile
    if (t < nimg * ngt) {
        const int il2 = t / ngt, gl = t - il2 * ngt;
        float* sp = d.gnf_stats + ((int64_t)((m0 >> p_sh) + il2) * G + (n0 >> cpg_sh) + gl) * 2;
        sp[0] = gm[il2 * 16 + gl]; sp[1] = gr[il2 * 16 + gl];
    }
}

// ----------------------------------------------------------------------------
// epilogue_rows: the epilogue of conv_small_body in REGISTERS.  The waves' partial tiles pass through LDS once anyway (the k-split
// reduction), so the reader is free to pick its own layout: wave g (of four) takes channels 8g .. 8g+7 of the 32-channel tile = ONE
// GroupNorm group (C / G = 8), lane r takes pixel r of the 64-pixel tile -- a lane's eight values are exactly one 16-byte NHWC vector
// (x, the residual, the destination and the normalised output are each one load / store per lane), and every GroupNorm sum is a sum
// over the lanes of an image: wave shuffles, no LDS passes, no workgroup barriers.  The LDS epilogues this replaces on these tiles
// (epilogue_tile_gnf / _gnb: ~12 barriers, two dependent rounds of column sums with 16-way bank conflicts) took 11 000 cycles behind a
// loop of 11 000 (in-kernel stamps, profiles/r04_small_conv_stamps.txt).  Same arithmetic: statistics in two passes over the
// bf16-rounded values (forward), the gn_bwd_reg_kernel formulas (backward); sums in a fixed (butterfly) order.
// Requires C / G == 8 where a GroupNorm is fused, N0 % 8 == 0.  P16: 4x4 maps (an image = 16 lanes), else 8x8 (64 lanes).
// (Measured and NOT kept, three ways of fetching the epilogue's vectors -- x / residual, the accumulate tensors -- early: plain loads at
//  KERNEL ENTRY (loads return in order: the loop's first counted wait then waits for HBM: 3.535 -> 3.594 ms/step); touching the lines
//  through the dummy DMA slots of the last TWO groups (3.703 -> 3.738) or of the LAST group only (3.523 -> 3.553); and ALL of the
//  epilogue's global loads issued in one burst in front of the barrier and the LDS gather (3.553 -> 3.565): the epilogue is not waiting
//  for memory.  Stamps (32 x 16 tiles, 4x4 maps): partials parked + barrier 330 cycles, gather 900, epilogue_rows 4 700 (forward
//  GroupNorm) / 7 300 (backward), drain 900.)
// ----------------------------------------------------------------------------
template <bool P16>
__device__ __forceinline__ float img_sum(float v) {                // sum over the lanes of one image, result in every lane of it
#pragma unroll
    for (int o = 1; o < (P16 ? 16 : 64); o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <bool P16>
__device__ __forceinline__ float tile_sum_from_img(float v) {      // image sums -> sum over all 64 pixels of the tile
    if constexpr (P16) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); }
    return v;
}
__device__ __forceinline__ void unpack8(const float8& a, float (&o)[8]) {
    o[0] = a.lo.x; o[1] = a.lo.y; o[2] = a.lo.z; o[3] = a.lo.w; o[4] = a.hi.x; o[5] = a.hi.y; o[6] = a.hi.z; o[7] = a.hi.w;
}
__device__ __forceinline__ float8 pack8(const float (&o)[8]) {
    return {make_float4(o[0], o[1], o[2], o[3]), make_float4(o[4], o[5], o[6], o[7])};
}
__device__ __forceinline__ void load8f(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// conv_small_body's parked partials -> pixel r, channels 8g .. 8g+7 of its 64 x 32 tile: two accumulator quads of each of the four
// k-step partials, added in ascending k (fixed order)
template <int MI, int NI>          // a wave's partial = MI x NI accumulator tiles; pixel half ph = r / (16 MI) (r < 32 MI)
__device__ __forceinline__ void rows_gather_small(const f32x4* red, const int g, const int r, float (&v)[8]) {
    constexpr int T = MI * NI;
    const int ph = r / (16 * MI), ti = ((r >> 4) % MI) * NI + (g >> 1), l0 = (r & 15) + 32 * (g & 1);
    {
        const f32x4 a = red[((ph * 4) * T + ti) * 64 + l0], b = red[((ph * 4) * T + ti) * 64 + l0 + 16];
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
        const f32x4 a = red[((ph * 4 + ks) * T + ti) * 64 + l0], b = red[((ph * 4 + ks) * T + ti) * 64 + l0 + 16];
        v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
}
// `act`: this lane's pixel exists (tiles of fewer than 64 pixels leave lanes over: they load valid addresses, contribute zeros to
// every sum and store nothing)
template <bool WT, bool P16>
__device__ __forceinline__ void epilogue_rows(const mdm_gemm_desc& d, float (&v)[8], const int m0, const int n0, const int g, const int r_lane,
                                              const bool act = true) {
    constexpr int P = P16 ? 16 : 64, p_sh = P16 ? 4 : 6;
    const int r = act ? r_lane : 0;
    if (!act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    const int64_t m = (int64_t)m0 + r;
    const int n = n0 + 8 * g, C = d.N;
    const int img = (m0 >> p_sh) + (P16 ? (r >> 4) : 0);
    const bool first = act && (r & (P - 1)) == 0;                   // the lane that writes per-image results
    const float inv_cnt = 1.f / (8.f * (float)P);
    if (d.gnb_x) {
        // ---- GroupNorm backward of the tensor this convolution's input came from (mdm_gemm_can_fuse_gn_bwd: no bias / row / residual)
        const int G = d.gnb_G;
        float x[8], ga[8], be[8], gz[8], xh[8], o[8];
        unpack8(load8(reinterpret_cast<const bf16_t*>(d.gnb_x) + m * C + n), x);
        const float2 st = *reinterpret_cast<const float2*>(d.gnb_stats + ((int64_t)img * G + (n >> 3)) * 2);
        load8f(d.gnb_gamma + n, ga);
        load8f(d.gnb_beta + n, be);
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh[e] = (x[e] - st.x) * st.y;
            gz[e] = v[e];
            if (d.gnb_silu) gz[e] *= silu_grad_f(fmaf(xh[e], ga[e], be[e]));
            a1 += gz[e] * ga[e];
            a2 += gz[e] * ga[e] * xh[e];
        }
        a1 = img_sum<P16>(a1);
        a2 = img_sum<P16>(a2);
        const float k1 = st.y * a1 * inv_cnt, k2 = st.y * a2 * inv_cnt;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = act ? st.y * ga[e] * gz[e] - fmaf(xh[e], k2, k1) : 0.f;
        // dgamma / dbeta: sums over every pixel of the tile; lane e adds dgamma[n + e], lane 8 + e dbeta[n + e] (one atomic per
        // channel per workgroup, as before)
        float mine = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dg = tile_sum_from_img<P16>(img_sum<P16>(gz[e] * xh[e])), db = tile_sum_from_img<P16>(img_sum<P16>(gz[e]));
            mine = r_lane == e ? dg : (r_lane == 8 + e ? db : mine);
        }
        if (r_lane < 16) atomicAdd((r_lane < 8 ? d.gnb_dgamma : d.gnb_dbeta) + n + (r_lane & 7), mine);
        if (d.gnb_sum_img || d.gnb_sum_all) {                       // uniform: column sums of dx (before any accumulation)
            float cs[8], tot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] = img_sum<P16>(o[e]);
                const float tt = tile_sum_from_img<P16>(cs[e]);
                tot = r_lane == e ? tt : tot;
            }
            if (d.gnb_sum_img && first) {                           // this workgroup owns (image, channel)
                float* sp = d.gnb_sum_img + (int64_t)img * d.gnb_sum_ld + n;
                *reinterpret_cast<float4*>(sp) = make_float4(cs[0], cs[1], cs[2], cs[3]);
                *reinterpret_cast<float4*>(sp + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
            }
            if (d.gnb_sum_all && r_lane < 8) atomicAdd(d.gnb_sum_all + n + r_lane, tot);
        }
        bf16_t* p = reinterpret_cast<bf16_t*>(d.D0) + m * d.ldd0 + n;
        if (d.acc0) {
            float old[8];
            unpack8(load8(p), old);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += old[e];
        }
        if (d.gnb_add) {
            float old[8];
            unpack8(load8(reinterpret_cast<const bf16_t*>(d.gnb_add) + m * d.ldd0 + n), old);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += old[e];
        }
        if (act) store8_pub<WT>(p, pack8(o));
        return;
    }
    // ---- forward-type epilogue: scale, bias, time-embedding row, residual, accumulate; optionally the GroupNorm of the result
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= d.alpha;
    if (d.bias) {
        float b[8];
        load8f(d.bias + n, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b[e];
    }
    if (d.rowvec) {
        float b[8];
        load8f(d.rowvec + (int64_t)div_rows((int)m, d.rows_per_img) * d.rv_ld + n, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b[e];
    }
    if (d.resid) {
        float b[8];
        unpack8(load8(reinterpret_cast<const bf16_t*>(d.resid) + m * d.ldr + n), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b[e];
    }
    bf16_t* p; int accf;
    if (n < d.N0) { p = reinterpret_cast<bf16_t*>(d.D0) + m * d.ldd0 + n; accf = d.acc0; }
    else          { p = reinterpret_cast<bf16_t*>(d.D1) + m * d.ldd1 + (n - d.N0); accf = d.acc1; }
    if (accf) {
        float old[8];
        unpack8(load8(p), old);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += old[e];
    }
    if (act) store8_pub<WT>(p, pack8(v));
    if (d.gnf_out) {
        const int G = d.gnf_G;
        float y[8], ga[8], be[8], o[8];
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { y[e] = bf2f(f2bf(v[e])); s1 += y[e]; }       // the values as a GroupNorm launch would read them back
        const float mean = img_sum<P16>(s1) * inv_cnt;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float c0 = y[e] - mean; s2 += c0 * c0; }
        const float rstd = rsqrtf(img_sum<P16>(s2) * inv_cnt + d.gnf_eps);
        load8f(d.gnf_gamma + n, ga);
        load8f(d.gnf_beta + n, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = fmaf((y[e] - mean) * rstd, ga[e], be[e]);
            if (d.gnf_silu) o[e] = silu_f(o[e]);
        }
        if (act) store8_pub<WT>(reinterpret_cast<bf16_t*>(d.gnf_out) + m * C + n, pack8(o));
        if (first) {
            float* sp = d.gnf_stats + ((int64_t)img * G + (n >> 3)) * 2;
            sp[0] = mean; sp[1] = rstd;
        }
    }
}

#ifdef MDM_STAMP
// debug build only (make EXTRA=-DMDM_STAMP): cycles per phase of the slab loop, summed over waves
#define MDM_STAMP_RECS 32768
__device__ unsigned long long g_stamp_buf[MDM_STAMP_RECS * 32];     // one 32-entry record per wave (weight gradients: per workgroup), plain stores
__device__ __forceinline__ unsigned long long stamp_hw_id() {      // which CU this wave runs on: HW_ID (cu / sh / se fields) | XCC_ID << 32
    unsigned a, b;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(a));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(b));
    return ((unsigned long long)b << 32) | a;
}
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MDM_T(...) __VA_ARGS__
#else
#define MDM_T(...)
#endif

// (device body: the kernel proper below, and one role of conv_pair_kernel.  bx / gx / bz stand for blockIdx.x / gridDim.x /
// blockIdx.z of a launch of its own)
template <int BM, int BN, int NSTAGE, int WR, int WC, int WK, bool PIPE = false, bool STAG = false, bool WT = false>
__device__ __forceinline__ void conv_lin2_body(const mdm_gemm_desc& d, char* ring, const int bx, const int gx, const int bz) {
    constexpr int BK = 64, NW = WR * WC * WK;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int GA = A_BYTES / 1024 / NW, GB = B_BYTES / 1024 / NW, G = GA + GB;
    constexpr int WM = BM / WR, WN = BN / WC, MI = WM / 16, NI = WN / 16, KSN = 2 / WK;
    static_assert(GA >= 1 && GB >= 1 && MI >= 1 && NI >= 1 && KSN >= 1, "tile too small for this wave layout");
    MDM_T(const unsigned long long t_entry = stamp_now();)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wk = wave / (WR * WC), wrc = wave % (WR * WC), wr = wrc / WC, wc = wrc % WC;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int tile_i = xcd_remap(bx, gx);                              // XCD-aware tile order: see conv_halo_kernel
    const int tile_m = udiv_small(tile_i, tiles_n);                    // (scalar integer divisions cost ~200 cycles each here)
    const int m0 = tile_m * BM, n0 = (tile_i - tile_m * tiles_n) * BN;
    const char* zlane = reinterpret_cast<const char*>(g_zero_page) + lane * 16;
    const int sk = d.splitk < 1 ? 1 : d.splitk;
    ZInfo z; z.batch = 0; z.tap = 0; z.kbeg = 0; z.kend = d.K; z.outer = 0; z.nouter = 1; z.ks = bz;

    const int r_sub = lane >> 3, r_lch = (lane & 7) ^ r_sub;
    const int sgn = d.transposed ? -1 : 1;
    const int ntaps = d.KH * d.KW;

    const char* a_row0[GA];
    const char* a_row1[GA];
    unsigned a_vmask[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int gm = m0 + 8 * (wave * GA + j) + r_sub;
        RowPix rp = decode_row(d, gm < d.M ? gm : 0);
        const int by = d.transposed ? rp.oy + d.pad_t : rp.oy * d.stride - d.pad_t;     // stride 2: forward only (host rule)
        const int bx = d.transposed ? rp.ox + d.pad_l : rp.ox * d.stride - d.pad_l;
        unsigned vm = 0;
        for (int ty = 0; ty < d.KH; ++ty)
            for (int tx = 0; tx < d.KW; ++tx) {
                int iy = by + sgn * ty, ix = bx + sgn * tx;
                bool ok = gm < d.M && (unsigned)iy < (unsigned)d.IH && (unsigned)ix < (unsigned)d.IW;
                vm |= (ok ? 1u : 0u) << (ty * d.KW + tx);
            }
        a_vmask[j] = vm;
        const int64_t base0 = ((int64_t)rp.img * d.IH + by) * d.IW + bx;
        a_row0[j] = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.src0) + base0 * d.ld0 + 8 * r_lch);
        a_row1[j] = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.src1) + base0 * d.ld1 + 8 * r_lch);
    }
    const char* b_row[GB];
    bool b_ok[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int gn = n0 + 8 * (wave * GB + j) + r_sub;
        b_ok[j] = gn < d.N;
        b_row[j] = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.B) + (int64_t)(gn < d.N ? gn : 0) * d.ldb + 8 * r_lch);
    }

    // ---- issue cursor (all wave-uniform): filter tap, source, channel; one live pointer per piece
    const int tps = udiv_small(ntaps, sk);
    const int tap_beg = z.ks * tps;
    int i_tap = tap_beg, i_ty = udiv_small(tap_beg, d.KW), i_tx = tap_beg - i_ty * d.KW, i_src = 0, i_c = 0;
    const int nk = tps * (d.Ck / BK);
    int issued = 0;
    int i_stage = 0;                    // byte offset of the stage the next issue fills
    const char* pa[GA];
    const char* pb[GB];
#pragma unroll
    for (int j = 0; j < GA; ++j) pa[j] = zlane;
#pragma unroll
    for (int j = 0; j < GB; ++j) pb[j] = zlane;

    auto issue_prepare = [&]() {
        if (i_c == 0) {                                          // new (tap, source) segment: re-aim
            if (issued < nk) {
                const int ld = i_src ? d.ld1 : d.ld0;
                const int64_t aoff = (int64_t)(sgn * (i_ty * d.IW + i_tx)) * ld * 2;
                const int64_t boff = ((int64_t)i_tap * d.wtap + (i_src ? d.C0 : 0)) * 2;
#pragma unroll
                for (int j = 0; j < GA; ++j)
                    pa[j] = ((a_vmask[j] >> i_tap) & 1u) ? (i_src ? a_row1[j] : a_row0[j]) + aoff : zlane;
#pragma unroll
                for (int j = 0; j < GB; ++j) pb[j] = b_ok[j] ? b_row[j] + boff : zlane;
            } else {                                             // over-issued tail slabs: zeros
#pragma unroll
                for (int j = 0; j < GA; ++j) pa[j] = zlane;
#pragma unroll
                for (int j = 0; j < GB; ++j) pb[j] = zlane;
            }
        }
    };
    auto issue_piece = [&](int g) {                              // g is a compile-time constant after unrolling
        char* stage = ring + i_stage;
#pragma unroll
        for (int j = 0; j < GA; ++j)
            if (g == j) { lds_dma16(pa[j], stage + (wave * GA + j) * 1024); pa[j] += 128; }
#pragma unroll
        for (int j = 0; j < GB; ++j)
            if (g == GA + j) { lds_dma16(pb[j], stage + A_BYTES + (wave * GB + j) * 1024); pb[j] += 128; }
    };
    auto issue_finish = [&]() {
        i_stage += STAGE_BYTES;
        if (i_stage == NSTAGE * STAGE_BYTES) i_stage = 0;
        if (++issued <= nk) {
            i_c += BK;
            if (i_c >= (i_src ? d.C1 : d.C0)) {
                i_c = 0;
                if (++i_src == (d.C1 > 0 ? 2 : 1)) {
                    i_src = 0; ++i_tap;
                    if (++i_tx == d.KW) { i_tx = 0; ++i_ty; }
                }
            }
            if (issued == nk) i_c = 0;                           // next call re-aims at the zero page
        }
    };
    auto issue = [&]() {
        issue_prepare();
#pragma unroll
        for (int g = 0; g < G; ++g) issue_piece(g);
        issue_finish();
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets of this lane inside a stage (the swizzle term depends on lane and k-step only)
    int a_off[KSN], b_off[KSN];
#pragma unroll
    for (int q = 0; q < KSN; ++q) {
        const int ks = wk + q * WK;
        const int ch = ((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
        a_off[q] = (wr * WM + (lane & 15)) * 128 + ch;
        b_off[q] = A_BYTES + (wc * WN + (lane & 15)) * 128 + ch;
    }

    int c_stage = 0;
    MDM_T(unsigned long long tw = 0, ti = 0, tc = 0, tb = 0; unsigned long long tis[12] = {}, tbs[12] = {}; unsigned long long tstart = 0;)
    if constexpr (PIPE) {
        // Software-pipelined slab loop.  The fragments of k-step 1 are read while k-step 0 multiplies, the
        // barrier sits in the MIDDLE of a slab (after it every wave has read slab `it` completely, so its
        // stage takes slab it+NSTAGE at once), the fragments of the next slab's k-step 0 are read while
        // k-step 1 multiplies, and the LDS-DMA pieces of the refill are dealt out between those MFMAs
        // (all waves issuing their pieces in one burst right behind the barrier queued ~500 cycles on the
        // CU's address unit per slab with the matrix pipe idle).
        static_assert(!PIPE || KSN == 2, "pipelined loop: both k-steps in one wave");
        // STAG: waves NW/2.. run half a slab BEHIND waves 0..NW/2-1 (the barrier sits before their k-step 0
        // instead of after it), so on every SIMD one wave multiplies (phase A) while its partner sits in the
        // DMA-issue-heavy phase B.  The late waves still read slab k while the early ones already refill, so
        // the refill goes one stage further back: slab k+3 replaces slab k-1 and the ring has 4 stages.
        static_assert(!STAG || NSTAGE == 4, "staggered loop: refill distance 3 needs 4 stages");
        constexpr int AHEAD = STAG ? 3 : NSTAGE;                 // slabs issued before the loop
        const bool late = STAG && wave >= NW / 2;
#pragma unroll
        for (int s2 = 0; s2 < AHEAD; ++s2) issue();
        wait_vmcnt<(AHEAD - 1) * G>();
        __builtin_amdgcn_s_barrier();
        bf16x8 af0[MI], bf0[NI], af1[MI], bf1[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(ring + a_off[0] + i * 2048);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf0[j] = *reinterpret_cast<const bf16x8*>(ring + b_off[0] + j * 2048);
        MDM_T(tstart = stamp_now();)
        auto phase_a = [&]() {
            const char* st = ring + c_stage;
#pragma unroll
            for (int i = 0; i < MI; ++i) af1[i] = *reinterpret_cast<const bf16x8*>(st + a_off[1] + i * 2048);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf1[j] = *reinterpret_cast<const bf16x8*>(st + b_off[1] + j * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j], af0[i], acc[i][j], 0, 0, 0);
        };
        for (int it = 0; it < nk; ++it) {
            MDM_T(const unsigned long long t0 = stamp_now();)
            if (!late) phase_a();
            __builtin_amdgcn_sched_barrier(0);
            MDM_T(const unsigned long long t0b = stamp_now();)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read this wave has issued has returned
            wait_vmcnt<(AHEAD - 2) * G>();                       // slab it+1 of this wave has landed
            MDM_T(const unsigned long long t1 = stamp_now();)
            __builtin_amdgcn_s_barrier();
            MDM_T(const unsigned long long t2 = stamp_now();)
            __builtin_amdgcn_sched_barrier(0);
            if (late) phase_a();
            c_stage += STAGE_BYTES;
            if (c_stage == NSTAGE * STAGE_BYTES) c_stage = 0;
            const char* sn = ring + c_stage;
#pragma unroll
            for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(sn + a_off[0] + i * 2048);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf0[j] = *reinterpret_cast<const bf16x8*>(sn + b_off[0] + j * 2048);
            issue_prepare();
            constexpr int NM = MI * NI, EVERY = NM / G > 0 ? NM / G : 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j], af1[i], acc[i][j], 0, 0, 0);
                    const int m = i * NI + j;
                    if (m % EVERY == EVERY - 1 && m / EVERY < G) issue_piece(m / EVERY);
                }
#pragma unroll
            for (int g = NM / EVERY; g < G; ++g) issue_piece(g);
            issue_finish();
            MDM_T(const unsigned long long t3 = stamp_now(); tw += t0b - t0; tb += t2 - t1; ti += t1 - t0b; tc += t3 - t2;)
        }
    } else {
#pragma unroll
    for (int s2 = 0; s2 < NSTAGE - 1; ++s2) issue();
    MDM_T(tstart = stamp_now();)
    for (int it = 0; it < nk; ++it) {
        MDM_T(const unsigned long long t0 = stamp_now();)
        wait_vmcnt<(NSTAGE - 2) * G>();
        MDM_T(const unsigned long long t0b = stamp_now();)
        __builtin_amdgcn_s_barrier();
        MDM_T(const unsigned long long t1 = stamp_now();)
        issue();
        MDM_T(const unsigned long long t2 = stamp_now();)
        const char* st = ring + c_stage;
#pragma unroll
        for (int q = 0; q < KSN; ++q) {
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[q] + i * 2048);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_off[q] + j * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        c_stage += STAGE_BYTES;
        if (c_stage == NSTAGE * STAGE_BYTES) c_stage = 0;
        MDM_T(const unsigned long long t3 = stamp_now(); tw += t0b - t0; tb += t1 - t0b; ti += t2 - t1; tc += t3 - t2;
              _Pragma("unroll") for (int q = 0; q < 12; ++q) if (it == q) { tis[q] = t2 - t1; tbs[q] = t1 - t0b; })
    }
    }
    wait_vmcnt<0>();
    MDM_T(const unsigned long long t_loop_end = stamp_now();)
    if (WK == 2) {                      // the k-halves of a wave pair meet in LDS (the ring is free now)
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(ring) + (wrc * MI * NI) * 64 + lane;
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) red[(i * NI + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (wk == 1) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                f32x4 o = red[(i * NI + j) * 64];
                acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
            }
    }
    const bool tile_epi = WK == 1 && !(d.splitk > 1 && d.ws) && !d.out_f32 && (d.N & 7) == 0 && (d.N0 & 7) == 0;
    if (tile_epi) {
        __syncthreads();                // every wave is done with the ring (tail DMA landed: vmcnt(0) above)
        if constexpr (BM == 64 && BN == 64 && WR == 4 && WC == 2 && WK == 1) {
            // 1x1 convolutions on 4x4 / 8x8 maps (the attention block's projections): a 64-row tile is whole images, like the
            // 64-pixel halo tiles -- the same fused GroupNorm epilogues apply (lin2_gn_tile)
            // (the register form of these epilogues -- epilogue_rows with all eight waves, one GroupNorm group each -- was built for
            //  these tiles too and measured level: 3.558 vs 3.562 ms/step; not kept)
            if (d.gnb_x) epilogue_tile_gnb<MI, NI, 64, WT>(d, ring, m0, n0, wr * WM, wc * WN, lane, t, acc);
            else if (d.gnf_out) epilogue_tile_gnf<MI, NI, 64, WT>(d, ring, m0, n0, wr * WM, wc * WN, lane, t, acc);
            else epilogue_tile<BM, BN, NW, MI, NI, bf16_t, WT>(d, ring, m0, n0, wr * WM, wc * WN, lane, t, acc);
        } else
        epilogue_tile<BM, BN, NW, MI, NI, bf16_t, WT>(d, ring, m0, n0, wr * WM, wc * WN, lane, t, acc);
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + wr * WM + i * 16 + (lane & 15);
            if (m >= d.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
                if (n < d.N) epilogue4<bf16_t>(d, z, m, n, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
            }
        }
    }
#ifdef MDM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const unsigned widx = (unsigned)(bz * gx + bx) * NW + wave;
        if (widx < 4096) {
            unsigned long long* r = g_stamp_buf + widx * 32;
            r[0] = tw; r[1] = tb; r[2] = ti; r[3] = tc; r[4] = nk; r[5] = 1; r[6] = t_loop_end - tstart; r[7] = tstart;
            r[8] = tstart - t_entry; r[9] = stamp_now() - t_loop_end; r[10] = t_entry;
        }
    }
#endif
}


template <int BM, int BN, int NSTAGE, int WR, int WC, int WK, bool PIPE = false, bool STAG = false>
__global__ __launch_bounds__(64 * WR * WC * WK) void conv_lin2_kernel(mdm_gemm_desc d) {
    extern __shared__ __attribute__((aligned(1024))) char ring[];
    conv_lin2_body<BM, BN, NSTAGE, WR, WC, WK, PIPE, STAG>(d, ring, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z);
}

// ----------------------------------------------------------------------------
// wgrad_lin: weight gradient of a stride-1 "same" convolution (layout 2: k = output pixel, rows of both
// operands are pixels), the counterpart of conv_lin2 for the backward-weights pass.  gemm_ring_kernel decodes
// (image, y, x) and gathers the source pixel per lane per piece per slab; here the input pixel of tap
// (ty, tx) is LINEAR in the reduction index (p_in = p_out + (ty - pad_t) * W + tx - pad_l), a slab is 64
// consecutive pixels = 64 / W whole image rows, so a lane's x never changes and its y advances by a constant:
// one live pointer per piece (+= a constant per slab) and ~6 VALU of validity per gathered piece.
// Requires: stride 1, no upsample, IH == OH, IW == OW, 64 % OW == 0, OH a power of two, K % 64 == 0.
// ----------------------------------------------------------------------------
template <int BM, int BN, int NSTAGE, int NW>
__device__ __forceinline__ void wgrad_lin_body(const mdm_gemm_desc& d, int item, const int tiles_x) {
    constexpr int BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int GA = A_BYTES / 1024 / NW, GB = B_BYTES / 1024 / NW, G = GA + GB;
    constexpr int ACPR = BM / 8, BCPR = BN / 8, A_RPP = 64 / ACPR, B_RPP = 64 / BCPR;
    constexpr int WROWS = NW / 2;
    constexpr int WM = BM / WROWS, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    extern __shared__ __attribute__((aligned(1024))) char ring[];
    MDM_T(const unsigned long long t_entry = stamp_now();)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN;
    // work item order: k-range major, then filter tap, then output tile -- a contiguous run per XCD (xcd_remap)
    const int sk = d.splitk < 1 ? 1 : d.splitk, ntap = d.KH * d.KW, per_k = ntap * tiles_x;
    // (wave-uniform quotients through udiv_small and shifts: seven scalar integer divisions cost ~1700 cycles per work item)
    const int ks_i = udiv_small(item, per_k), rem_i = item - ks_i * per_k, tap_i = udiv_small(rem_i, tiles_x), tile_i = rem_i - tap_i * tiles_x;
    const int tile_m = udiv_small(tile_i, tiles_n);
    const int m0 = tile_m * BM, n0 = (tile_i - tile_m * tiles_n) * BN;
    ZInfo z; z.batch = 0; z.tap = tap_i; z.outer = tap_i; z.nouter = ntap; z.ks = ks_i; z.kbeg = 0; z.kend = d.K;
    if (sk > 1) {
        const int chunk = (udiv_small(d.K + sk - 1, sk) + BK - 1) / BK * BK;
        z.kbeg = ks_i * chunk; z.kend = min(d.K, z.kbeg + chunk);
    }
    const char* zlane = reinterpret_cast<const char*>(g_zero_page) + lane * 16;
    const int ty = udiv_small(z.tap, d.KW), tx = z.tap - ty * d.KW;
    const int dyy = ty - d.pad_t, dxx = tx - d.pad_l;
    const int ow_sh = __builtin_ctz(d.OW), oh_sh = __builtin_ctz(d.OH);       // powers of two (wgrad_lin_eligible)
    const int rows_per_slab = BK >> ow_sh;
    const int nk = (z.kend - z.kbeg) / BK;

    // ---- A = dY[k][m]: [64][BM] image, piece = A_RPP k-rows
    const char* pa[GA];
    const int64_t a_step = (int64_t)BK * d.lda * 2;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int kl = (wave * GA + j) * A_RPP + lane / ACPR;
        const int gm = m0 + 8 * ((lane % ACPR) ^ swz_cols<ACPR>(kl));
        pa[j] = (gm < d.M && nk > 0) ? reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.A) + (int64_t)(z.kbeg + kl) * d.lda + gm) : zlane;
        if (!(gm < d.M)) pa[j] = zlane;
    }
    bool a_live[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) a_live[j] = pa[j] != zlane;
    // ---- B = gathered input pixels [k][n]
    const char* pb[GB];         // pointer to the SHIFTED source pixel of the current slab (may be out of the image: see b_val)
    int b_y[GB];                // y of this lane's output pixel in the current slab
    int b_step[GB];             // bytes per slab (depends on the source the lane's channels come from)
    bool b_xok[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int kl = (wave * GB + j) * B_RPP + lane / BCPR;
        const int gn = n0 + 8 * ((lane % BCPR) ^ swz_cols<BCPR>(kl));
        const int k = z.kbeg + kl;
        const int x = k & (d.OW - 1), y = (k >> ow_sh) & (d.OH - 1), img_i = k >> (ow_sh + oh_sh);
        b_y[j] = y;
        // source pixel of output pixel (y, x) under tap (ty, tx): virtual (y s + dy, x s + dx), physical = virtual >> ups
        // (stride 2: unet6.py:257-272; folded nearest x2 upsample: unet6.py:472).  A slab is rows_per_slab whole output
        // rows, so the physical row advances by (rows_per_slab * s) >> ups per slab -- also across image boundaries.
        const int vx = x * d.stride + dxx, vy = y * d.stride + dyy;
        b_xok[j] = gn < d.N && (unsigned)vx < (unsigned)d.IW;
        const bool s1 = gn >= d.C0;
        const int ld = s1 ? d.ld1 : d.ld0;
        const bf16_t* S = reinterpret_cast<const bf16_t*>(s1 ? d.src1 : d.src0);
        const int PH = d.IH >> d.ups, PW = d.IW >> d.ups;
        b_step[j] = ((rows_per_slab * d.stride) >> d.ups) * PW * ld * 2;
        pb[j] = reinterpret_cast<const char*>(S + (((int64_t)img_i * PH + (vy >> d.ups)) * PW + (vx >> d.ups)) * ld + (s1 ? gn - d.C0 : gn));
    }

    int issued = 0, i_stage = 0;
    auto issue = [&]() {
        char* stage = ring + i_stage;
        const bool live = issued < nk;
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            lds_dma16((live && a_live[j]) ? pa[j] : zlane, stage + (wave * GA + j) * 1024);
            pa[j] += a_step;
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const bool ok = live && b_xok[j] && (unsigned)(b_y[j] * d.stride + dyy) < (unsigned)d.IH;
            lds_dma16(ok ? pb[j] : zlane, stage + A_BYTES + (wave * GB + j) * 1024);
            pb[j] += b_step[j];
            b_y[j] = (b_y[j] + rows_per_slab) & (d.OH - 1);
        }
        ++issued;
        i_stage += STAGE_BYTES;
        if (i_stage == NSTAGE * STAGE_BYTES) i_stage = 0;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = d.dbias != nullptr && z.outer == 0 && n0 == 0 && wc == 0;
    f32x4 accb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

#pragma unroll
    for (int s2 = 0; s2 < NSTAGE - 1; ++s2) issue();
    int c_stage = 0;
    MDM_T(const unsigned long long tstart = stamp_now();)
    for (int it = 0; it < nk; ++it) {
        wait_vmcnt<(NSTAGE - 2) * G>();
        __builtin_amdgcn_s_barrier();
        issue();
        const char* As = ring + c_stage;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            TrFrag ta[MI], tb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) ring_frag_cols_issue<ACPR>(As, wr * WM + i * 16, ks, lane, ta[i]);
#pragma unroll
            for (int j = 0; j < NI; ++j) ring_frag_cols_issue<BCPR>(Bs, wc * WN + j * 16, ks, lane, tb[j]);
            tr_wait();
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = tr_value(ta[i]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[j] = tr_value(tb[j]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < MI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], accb[i], 0, 0, 0);
            }
        }
        c_stage += STAGE_BYTES;
        if (c_stage == NSTAGE * STAGE_BYTES) c_stage = 0;
    }
    wait_vmcnt<0>();
    MDM_T(const unsigned long long t_loop_end = stamp_now();)
    if (do_bias && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + wr * WM + i * 16 + (lane & 15);
            if (m < d.M) atomicAdd(&d.dbias[m], accb[i][0]);
        }
    }
    if (d.splitk > 1 && d.ws && (d.N & 3) == 0) {          // uniform: partial slab, wide stores through LDS
        __syncthreads();                // every wave is done with the ring (tail DMA landed: vmcnt(0) above)
        float* slab = reinterpret_cast<float*>(d.ws) + ((int64_t)z.ks * z.nouter + z.outer) * ((int64_t)d.M * d.N);
        epilogue_tile_slab<BM, BN, NW, MI, NI>(d, slab, ring, m0, n0, wr * WM, wc * WN, lane, t, acc);
    } else if (d.splitk <= 1 && d.out_f32 && !d.acc0 && d.ldd0 == d.N && d.N0 == d.N && (d.N & 3) == 0 && d.alpha == 1.0f) {
        // unsplit, overwriting, dense fp32 destination (the weight gradient of a small map): the gradient IS the slab
        __syncthreads();
        float* slab = reinterpret_cast<float*>(d.D0) + z.tap * d.dtap;
        epilogue_tile_slab<BM, BN, NW, MI, NI>(d, slab, ring, m0, n0, wr * WM, wc * WN, lane, t, acc);
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int m = m0 + wr * WM + i * 16 + (lane & 15);
            if (m >= d.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int n = n0 + wc * WN + j * 16 + 4 * (lane >> 4);
                if (n < d.N) epilogue4<bf16_t>(d, z, m, n, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
            }
        }
    }
#ifdef MDM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wave == 0) {          // one record per WORKGROUP (a group launch has thousands of them)
        const unsigned widx = (unsigned)blockIdx.x;
        if (widx < MDM_STAMP_RECS) {
            unsigned long long* r = g_stamp_buf + widx * 32;
            const unsigned long long t_end = stamp_now();
            r[0] = 0; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = nk > 0 ? nk : 0; r[5] = 1; r[6] = t_loop_end - tstart; r[7] = tstart;
            r[8] = tstart - t_entry; r[9] = t_end - t_loop_end; r[10] = t_entry;
            r[11] = BM; r[12] = d.M; r[13] = d.N; r[14] = d.K; r[15] = stamp_hw_id(); r[16] = t_end; r[17] = d.KH * d.KW; r[18] = d.splitk;
        }
    }
#endif
}


template <int BM, int BN, int NSTAGE, int NW>
__global__ __launch_bounds__(64 * NW) void wgrad_lin_kernel(mdm_gemm_desc d, int tiles_x) {
    wgrad_lin_body<BM, BN, NSTAGE, NW>(d, (int)blockIdx.x, tiles_x);
}

// ----------------------------------------------------------------------------
// wgrad_taps: ALL NINE TAPS of a 3x3 stride-1 weight gradient from one pass over dY and the layer input.
// wgrad_lin runs every filter tap as its own contraction and every wave reads its MFMA fragments with the transposing
// ds_read_b64_tr_b16 (both operands are pixel-major in memory, the contraction runs over pixels).  Stamps inside the grouped
// launch: 2 500 cycles per 64-pixel slab of a 256x128 tile against 1 024 of MFMA -- and the same ~10 cycles per transposing
// read instruction per CU on every tile shape: the loop is bound by the LDS transposing reads, which the waves of a tile repeat
// for each other (4x on dY) and the nine taps repeat on the same input pixels.
// Here a 128 (output channels) x 64 (input channels) tile
//   * lands dY slabs and input blocks pixel-major by LDS-DMA as before, then transposes each of them ONCE (one transposing
//     read + one ds_write_b128 per 8 pixels x 16 channels, shared out over the 8 waves) into channel-major arrays At[128][64 px]
//     and Xt[64][256 px ring]; the fragments of the loop are then plain 16-byte reads (8 consecutive pixels of one channel);
//   * takes the nine taps from the SAME ring: tap (ty, tx) reads at pixel offset (ty - 1) W + (tx - 1).  The row offset is an
//     address; the +-1 pixel of tx is made in registers from the centre fragment and its two neighbour dwords (v_alignbit);
//   * masks what lies outside the image in registers: x neighbours by a lane-constant AND on one register of the fragment (a
//     fragment is 8 consecutive pixels of one image row, W >= 8), y neighbours by zeroing fragments of the first / last row;
//   * a folded nearest x2 upsample (unet6.py:472) only changes the address the input DMA reads a (virtual) pixel from.
// 24 KB of operands per slab for 9.4 MFLOP (per-tap tiles: 48 KB for 4.2), 36 fp32 accumulator tiles (144 registers) per wave.
// Work item = (tile, range of slabs [k0, k1)): the host cuts the (tile, slab) space of a whole group into one contiguous
// share per CU (mdm_wgrad_group_create), so a tile that is cut writes its partial sums to a `slot` ([9][128][64] fp32) and
// tile_parts_reduce_kernel adds the slots; an uncut tile goes straight to the gradient.
// Requires: 3x3, stride 1, pad 1, OW in {8, 16, 32}, OH a power of two, OH OW >= 64, bf16; M a multiple of 128 or one partly filled
// tile (M < 128, 8 | M), N a multiple of 64 or one partly filled tile: the two 8-channel ends of the U-Net ride along on zero operands.
// ----------------------------------------------------------------------------
constexpr int TAPS_BM = 128, TAPS_BN = 64, TAPS_SLOT_FLOATS = 9 * TAPS_BM * TAPS_BN;
constexpr int TAPS_XL = 0, TAPS_XT = 3 * 8192, TAPS_XT_PITCH = 528, TAPS_AL = TAPS_XT + 64 * TAPS_XT_PITCH, TAPS_AT = TAPS_AL + 3 * 16384,
              TAPS_AT_PITCH = 144, TAPS_LDS_BYTES = TAPS_AT + 2 * TAPS_BM * TAPS_AT_PITCH;
static_assert(TAPS_LDS_BYTES <= 3 * (256 + 128) * 64 * 2 && 2 * TAPS_BM * TAPS_BN * 4 <= TAPS_LDS_BYTES, "nine-tap LDS map");

__device__ __forceinline__ int swz2(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
// LDS traffic of the loop through asm: the compiler cannot tell these addresses from the LDS-DMA destinations in flight and
// would otherwise wait for vmcnt(0) in front of the first read (see ring_frag_cols_issue)
__device__ __forceinline__ void lds_read16(unsigned addr, v4u_t& v) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void lds_read4(unsigned addr, unsigned& v) { asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void lds_write16(unsigned addr, const v4u_t& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void tr_issue_at(unsigned a0, unsigned a1, TrFrag& f) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1) : "memory");
}

template <bool BIAS>
__device__ __forceinline__ void wgrad_taps_body(const mdm_gemm_desc& d, const int tile_i, const int k0, const int k1, float* slot) {
    constexpr int BM = TAPS_BM, BN = TAPS_BN, NW = 8, MI = 4;
    extern __shared__ __attribute__((aligned(1024))) char ring[];
    MDM_T(const unsigned long long t_entry = stamp_now();)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int tile_m = udiv_small(tile_i, tiles_n);
    const int m0 = tile_m * BM, n0 = (tile_i - tile_m * tiles_n) * BN;
    const int W = d.OW, H = d.OH, lw = __builtin_ctz(d.OW), lh = __builtin_ctz(d.OH);
    const int nk = k1 - k0, total_slabs = d.K >> 6;
    const int rows = min(BM, d.M - m0), cols = min(BN, d.N - n0);        // an 8-channel end of the net is ONE partly filled tile (zero operands)
    const char* zlane = reinterpret_cast<const char*>(g_zero_page) + lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)ring;
    const int i16 = lane & 15, g4 = lane >> 4;

    // ---- landing buffers.  dY: [64 pixels][128 channels] per stage, 16 pieces of 4 pixel rows, two per wave (as wgrad_lin)
    const char* pa[2];
    const int64_t a_step = (int64_t)64 * d.lda * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kl = (wave * 2 + j) * 4 + (lane >> 4);
        const int gm = m0 + 8 * ((lane & 15) ^ swz_cols<16>(kl));
        pa[j] = gm < d.M ? reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.A) + ((int64_t)k0 * 64 + kl) * d.lda + gm) : nullptr;
    }
    // input pixels: [64 pixels][64 channels] per block, one piece (8 pixel rows) per wave
    const int xrow = 8 * wave + (lane >> 3);
    const int xgn = n0 + 8 * ((lane & 7) ^ swz2(xrow));
    const bool xs1 = xgn >= d.C0;
    const int xld = xs1 ? d.ld1 : d.ld0;
    const bf16_t* const xS = reinterpret_cast<const bf16_t*>(xs1 ? d.src1 : d.src0) + (xs1 ? xgn - d.C0 : xgn);
    const int ups = d.ups;
    // (`d` lives in device memory and the loop is full of asm with memory clobbers: a `d.N` inside x_src was RE-LOADED in every iteration,
    //  and the s_waitcnt vmcnt(0) the compiler puts behind such a load also waits for every LDS-DMA in flight -- the prefetch ring of the
    //  loop was drained once per slab.  Everything the loop needs from the descriptor is a local.)
    const bool x_live = xgn < d.N;
    auto x_src = [&](int blk) -> const char* {                    // where block `blk` (pixels 64 blk ..) comes from
        if (blk < 0 || blk >= total_slabs || blk > k1 || !x_live) return zlane;
        const int p = blk * 64 + xrow;
        int phys = p;
        if (ups) {
            const int x = p & (W - 1), y = (p >> lw) & (H - 1), img = p >> (lw + lh);
            phys = ((img * (H >> 1) + (y >> 1)) << (lw - 1)) + (x >> 1);
        }
        return reinterpret_cast<const char*>(xS + (int64_t)phys * xld);
    };
    auto stage3 = [](int v) { return v - 3 * (int)(((unsigned)(v + 3000) * 43691u >> 17) - 1000); };      // v mod 3 for v >= -3000
    auto issue_a = [&](int slab) {                                // dY slab -> landing stage slab mod 3
        const bool live = slab < k1;
        char* dst = ring + TAPS_AL + stage3(slab) * 16384;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds_dma16(live && pa[j] ? pa[j] : zlane, dst + (wave * 2 + j) * 1024);
            if (pa[j]) pa[j] += a_step;
        }
    };
    auto issue_x = [&](int blk) { lds_dma16(x_src(blk), ring + TAPS_XL + stage3(blk) * 8192 + wave * 1024); };

    // ---- transposes: landing (pixel-major) -> At / Xt (channel-major).  X: unit = wave: 16 channels x 32 pixels
    const int xcg = wave & 3, xks = wave >> 2;
    unsigned xt_src[2];                                           // the transposing read of this lane inside a landing block
    {
        const int col = xcg * 16 + 4 * (i16 & 3), ch = col >> 3, half = (col >> 2) & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = xks * 32 + 8 * g4 + (i16 >> 2) + 4 * h;
            xt_src[h] = (unsigned)(q * 128 + ((ch ^ swz2(q)) << 4) + half * 8);
        }
    }
    const unsigned xt_dst = lds0 + TAPS_XT + (unsigned)((xcg * 16 + i16) * TAPS_XT_PITCH);          // + 2 * ((64 blk + 32 ks + 8 g) & 255)
    auto transpose_x = [&](int blk, TrFrag& f) {
        const unsigned src = lds0 + TAPS_XL + (unsigned)(stage3(blk) * 8192);
        tr_issue_at(src + xt_src[0], src + xt_src[1], f);
    };
    auto store_x = [&](int blk, const TrFrag& f) {
        const v4u_t v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
        lds_write16(xt_dst + 2u * (unsigned)((blk * 64 + xks * 32 + 8 * g4) & 255), v);
    };
    // dY: 16 units of 16 channels x 32 pixels, two per wave
    auto transpose_a = [&](int slab, TrFrag (&f)[2]) {
        const char* src = ring + TAPS_AL + stage3(slab) * 16384;
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int u = wave * 2 + j; ring_frag_cols_issue<16>(src, (u & 7) * 16, u >> 3, lane, f[j]); }
    };
    auto store_a = [&](int slab, const TrFrag (&f)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = wave * 2 + j;
            const v4u_t v = {f[j].lo[0], f[j].lo[1], f[j].hi[0], f[j].hi[1]};
            lds_write16(lds0 + TAPS_AT + (unsigned)((slab & 1) * (BM * TAPS_AT_PITCH) + ((u & 7) * 16 + i16) * TAPS_AT_PITCH + ((u >> 3) * 32 + 8 * g4) * 2), v);
        }
    };

    // ---- fragment addresses of the loop (lane constants)
    const unsigned a_frag = lds0 + TAPS_AT + (unsigned)((wr * 64 + i16) * TAPS_AT_PITCH + 16 * g4);      // + buf, + i * 16 rows, + 64 ks
    const unsigned b_row = lds0 + TAPS_XT + (unsigned)((wc * 16 + i16) * TAPS_XT_PITCH);                 // + 2 * (pixel & 255)
    // x masks (lane constants): element 0 of a fragment is pixel x = (32 ks + 8 g) mod W, element 7 is x + 7
    unsigned mlo[2], mhi[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int x0 = (ks * 32 + 8 * g4) & (W - 1);
        mlo[ks] = x0 == 0 ? 0xFFFF0000u : 0xFFFFFFFFu;
        mhi[ks] = x0 + 7 == W - 1 ? 0x0000FFFFu : 0xFFFFFFFFu;
    }

    f32x4 acc[9][MI][1];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[tp][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient (column sums of dY) on the first input-channel tile of a layer: on the VALU, one fp32 per fragment and lane
    // (an MFMA against ones would cost 16 more accumulator registers next to the 144 of the taps)
    const bool do_bias = BIAS && wc == 0;
    float bsum[BIAS ? MI : 1];
#pragma unroll
    for (int i = 0; i < (BIAS ? MI : 1); ++i) bsum[i] = 0.f;

    // ---- prologue: blocks k0-1 .. k0+1 and slab k0 transposed, block k0+2 / slab k0+1 landed, block k0+3 / slab k0+2 in flight
    issue_x(k0 - 1); issue_x(k0); issue_x(k0 + 1); issue_a(k0);
    wait_vmcnt<0>();
    __syncthreads();
    {
        TrFrag fx[3], fa[2];
        transpose_x(k0 - 1, fx[0]); transpose_x(k0, fx[1]); transpose_x(k0 + 1, fx[2]); transpose_a(k0, fa);
        wait_lgkmcnt<0>();
        store_x(k0 - 1, fx[0]); store_x(k0, fx[1]); store_x(k0 + 1, fx[2]); store_a(k0, fa);
        wait_lgkmcnt<0>();              // (asm stores: the compiler's own wait in front of the barrier does not know them)
    }
    __syncthreads();                    // (the landing stages of k0-1 .. k0+1 / k0 are read: they may be refilled)
    issue_x(k0 + 2); issue_a(k0 + 1);
    issue_x(k0 + 3); issue_a(k0 + 2);
    MDM_T(const unsigned long long tstart = stamp_now(); unsigned long long tw = 0, tt = 0, tc = 0;)
    for (int it = 0; it < nk; ++it) {
        const int s = k0 + it;
        MDM_T(const unsigned long long q0 = stamp_now();)
        wait_vmcnt<3>();                // block s+2 and slab s+1 have landed
        __builtin_amdgcn_s_barrier();   // ... for every wave; and everybody is done with slab s-1 (Xt slot s-2, At[(s+1)&1], landing s+1 / s)
        MDM_T(const unsigned long long q1 = stamp_now();)
        // Order inside a slab: the fragment reads of the first k-step go out first; the transposing reads of the NEXT slab's operands and
        // the DMA of the ones after are issued behind the first MFMA burst and stored behind the third -- their LDS round trip and address
        // arithmetic sit under MFMAs instead of in front of them (every wave leaves the barrier at the same time: nothing else would run).
        TrFrag fx, fa[2];
        MDM_T(const unsigned long long q2 = stamp_now();)
        const unsigned a_buf = a_frag + (unsigned)((s & 1) * (BM * TAPS_AT_PITCH));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4u_t av[MI], bc[2];
            unsigned bm[2], bp[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) lds_read16(a_buf + (unsigned)(i * 16 * TAPS_AT_PITCH + ks * 64), av[i]);
            const int pbase = s * 64 + ks * 32 + 8 * g4;
            auto issue_row = [&](int dy, int b) {                 // the centre fragment of filter row dy and its two neighbour dwords
                const int p = pbase + (dy - 1) * W;
                lds_read16(b_row + 2u * (unsigned)(p & 255), bc[b]);
                lds_read4(b_row + 2u * (unsigned)((p - 2) & 255), bm[b]);
                lds_read4(b_row + 2u * (unsigned)((p + 8) & 255), bp[b]);
            };
            issue_row(0, 0);
            if (ks == 1) {              // in flight: the 6 transposing reads (older), 4 + 3 fragment reads: store the transposed copies
                wait_lgkmcnt<7>();
                store_x(s + 2, fx);
                store_a(s + 1, fa);
            }
            issue_row(1, 1);
            // y masks of this k-step's pixel row(s)
            const int yv = (pbase >> lw) & (H - 1);
            const unsigned m_up = yv != 0 ? 0xFFFFFFFFu : 0u, m_dn = yv != H - 1 ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int b = dy & 1;
                // ks 0: rows 1 / 2 fly, behind them (from dy = 1 on) the 6 transposing reads.  ks 1: the 3 stores sit between row 0 and row 1.
                if (ks == 0) { if (dy == 0) wait_lgkmcnt<3>(); else if (dy == 1) wait_lgkmcnt<9>(); else wait_lgkmcnt<6>(); }
                else { if (dy == 0) wait_lgkmcnt<6>(); else if (dy == 1) wait_lgkmcnt<3>(); else wait_lgkmcnt<0>(); }
                if (BIAS && do_bias && dy == 0) {
#pragma unroll
                    for (int i = 0; i < (BIAS ? MI : 1); ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) bsum[i] += __uint_as_float(av[i][e] << 16) + __uint_as_float(av[i][e] & 0xFFFF0000u);
                }
                const unsigned my = dy == 0 ? m_up : dy == 2 ? m_dn : 0xFFFFFFFFu;
                v4u_t c = bc[b];
                unsigned dm = bm[b], dp = bp[b];
                if (dy == 0) issue_row(2, 0);                     // (slot 0 is copied out: refill it behind the copy)
                if (dy != 1) { c[0] &= my; c[1] &= my; c[2] &= my; c[3] &= my; dm &= my; dp &= my; }
                const unsigned p1 = __builtin_amdgcn_alignbit(c[1], c[0], 16), p2 = __builtin_amdgcn_alignbit(c[2], c[1], 16),
                               p3 = __builtin_amdgcn_alignbit(c[3], c[2], 16);
                const v4u_t fl = {__builtin_amdgcn_alignbit(c[0], dm, 16) & mlo[ks], p1, p2, p3};           // pixels x-1 .. x+6
                const v4u_t fr = {p1, p2, p3, __builtin_amdgcn_alignbit(dp, c[3], 16) & mhi[ks]};           // pixels x+1 .. x+8
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, dx == 0 ? fl : dx == 1 ? c : fr);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[dy * 3 + dx][i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, __builtin_bit_cast(bf16x8, av[i]), acc[dy * 3 + dx][i][0], 0, 0, 0);
                }
                if (ks == 0 && dy == 0) {
                    transpose_x(s + 2, fx);
                    transpose_a(s + 1, fa);
                    issue_x(s + 4);     // into the landing stage block s+1 left (transposed during slab s-1)
                    issue_a(s + 3);     // ... slab s left
                }
            }
        }
        MDM_T(const unsigned long long q3 = stamp_now(); tw += q1 - q0; tt += q2 - q1; tc += q3 - q2;)
    }
    wait_vmcnt<0>();
    MDM_T(const unsigned long long t_loop_end = stamp_now();)
    if (BIAS && do_bias) {              // lane (g, c) holds the sum over its k-group's pixels of channel c: add the four groups
#pragma unroll
        for (int i = 0; i < (BIAS ? MI : 1); ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if ((lane >> 4) == 0 && wr * 64 + i * 16 + (lane & 15) < rows) atomicAdd(&d.dbias[m0 + wr * 64 + i * 16 + (lane & 15)], v);
        }
    }
    // ---- nine fp32 tiles -> the gradient (uncut tile) or this item's slot, through LDS for 16-byte stores of whole rows
    float* obase; int64_t opitch, otap;
    if (slot) { obase = slot; opitch = BN; otap = BM * BN; }
    else { obase = reinterpret_cast<float*>(d.D0) + (int64_t)m0 * d.N + n0; opitch = d.N; otap = d.dtap; }
    __syncthreads();                    // every wave is done with the rings (tail DMA landed: vmcnt(0) above)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        char* buf = ring + (tp & 1) * (BM * BN * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wr * 64 + i * 16 + (lane & 15), nl = wc * 16 + 4 * (lane >> 4);
            *reinterpret_cast<float4*>(buf + ml * (BN * 4) + (((nl >> 2) ^ (ml & 7)) << 4)) =
                make_float4(acc[tp][i][0][0], acc[tp][i][0][1], acc[tp][i][0][2], acc[tp][i][0][3]);
        }
        __syncthreads();
        float* o = obase + tp * otap;
#pragma unroll
        for (int idx = t; idx < BM * (BN / 4); idx += 64 * NW) {
            const int r = idx >> 4, q = idx & 15;
            if (slot || (r < rows && q * 4 < cols))
                *reinterpret_cast<float4*>(o + r * opitch + q * 4) = *reinterpret_cast<const float4*>(buf + r * (BN * 4) + ((q ^ (r & 7)) << 4));
        }
    }
#ifdef MDM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wave == 4) {
        const unsigned widx = 16384u + (unsigned)blockIdx.x + (unsigned)gridDim.x * (unsigned)(k0 & 31);
        if (widx < MDM_STAMP_RECS) { unsigned long long* r = g_stamp_buf + widx * 32; r[19] = tw; r[20] = tt; r[21] = tc; }
    }
    if (lane == 0 && wave == 0) {
        const unsigned widx = 16384u + (unsigned)blockIdx.x + (unsigned)gridDim.x * (unsigned)(k0 & 31);      // several items per workgroup: spread the records
        if (widx < MDM_STAMP_RECS) {
            unsigned long long* r = g_stamp_buf + widx * 32;
            const unsigned long long t_end = stamp_now();
            r[0] = 0; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = nk; r[5] = 1; r[6] = t_loop_end - tstart; r[7] = tstart;
            r[8] = tstart - t_entry; r[9] = t_end - t_loop_end; r[10] = t_entry;
            r[11] = 999; r[12] = d.M; r[13] = d.N; r[14] = d.K; r[15] = stamp_hw_id(); r[16] = t_end; r[17] = 9; r[18] = slot ? 2 : 1;
            r[0] = tw; r[1] = tt; r[2] = tc;
        }
    }
#endif
}

// sums the partial slots of the tiles that mdm_wgrad_group_create cut (fixed order: slot index = position in the tile's slab range)
struct PartTile { float* dst; long long dtap; int N, m0, n0, first_slot, parts, rows, cols, pad; };
__global__ __launch_bounds__(256) void tile_parts_reduce_kernel(const PartTile* __restrict__ tab, const float* __restrict__ slots) {
    constexpr int PER_TILE4 = TAPS_SLOT_FLOATS / 4, PIECES = PER_TILE4 / 1024;
    const PartTile pt = tab[blockIdx.x / PIECES];
    const int piece = blockIdx.x % PIECES;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i4 = piece * 1024 + k * 256 + threadIdx.x;
        const float4* w = reinterpret_cast<const float4*>(slots + (int64_t)pt.first_slot * TAPS_SLOT_FLOATS) + i4;
        float4 a = w[0];
        for (int s2 = 1; s2 < pt.parts; ++s2) {
            const float4 b = w[(int64_t)s2 * PER_TILE4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int tp = i4 / (TAPS_BM * TAPS_BN / 4), rem = i4 - tp * (TAPS_BM * TAPS_BN / 4), r = rem >> 4, q = rem & 15;
        if (r < pt.rows && q * 4 < pt.cols) *reinterpret_cast<float4*>(pt.dst + tp * pt.dtap + (int64_t)(pt.m0 + r) * pt.N + pt.n0 + q * 4) = a;
    }
}

// A GROUP of weight gradients in one launch (mdm_wgrad_group_*): the weight gradients are leaves of the backward
// pass -- they only read dY and the layer input, both of which stay in memory -- so they need not run where autograd
// would put them.  The host collects the descriptors of a whole stretch of the backward, cuts every contraction
// into work items of similar length (output tile x filter tap x k-range), sorts them longest first and launches
// them as ONE flat grid: the chip is filled by the group, not by each layer on its own, so a layer needs only as
// many k-splits as balance asks for (fewer fp32 partial slabs: the per-layer launches wrote and re-read ~0.9 GB
// of them per step at cfg2), and ~70 launch gaps / prologues / drain tails per step disappear.
// items[i] = {descriptor index, item index inside it, tiles per (tap, k-range), tile: 2 = 256x128, 1 = 128x128, 0 = 64x64}.
// (256 output channels x 128 input channels per workgroup feed 48 KB of operands per 64-pixel slab for twice the MFMA
// work of the 128x128 tile's 32 KB: the loop is bound by the bytes a CU takes in, so the wider tile runs ~1.3x faster.)
__global__ __launch_bounds__(512) void wgrad_group_kernel(const mdm_gemm_desc* descs, const int4* items, int n_items) {
    // grid == n_items: one item per workgroup; a smaller grid walks the (longest-first) list with stride gridDim.x
    for (int i = blockIdx.x; i < n_items; i += gridDim.x) {
        const int4 it = items[i];
        if (it.x < 0) continue;             // padding of the per-XCD queues (uniform for the workgroup)
        const mdm_gemm_desc d = descs[it.x];
        if (it.w == 2) wgrad_lin_body<256, 128, 3, 8>(d, it.y, it.z);
        else if (it.w == 1) wgrad_lin_body<128, 128, 3, 8>(d, it.y, it.z);
        else wgrad_lin_body<64, 64, 4, 8>(d, it.y, it.z);
        __syncthreads();            // the next item refills the LDS ring
    }
}
// The nine-tap layers of a group: grid = CUs, workgroup q walks column q of table[round][queue] (one share per CU, equal by
// construction: mdm_wgrad_group_create).  item = {descriptor, tile | (slot + 1) << 12, k0 | k1 << 16, 3}.
__global__ __launch_bounds__(512) void wgrad_taps_group_kernel(const mdm_gemm_desc* descs, const int4* items, int n_items, float* slots) {
    for (int i = blockIdx.x; i < n_items; i += gridDim.x) {
        int4 it = items[i];
        it.x = __builtin_amdgcn_readfirstlane(it.x); it.y = __builtin_amdgcn_readfirstlane(it.y); it.z = __builtin_amdgcn_readfirstlane(it.z);
        it.w = __builtin_amdgcn_readfirstlane(it.w);
        if (it.x < 0) continue;
        const mdm_gemm_desc& d = descs[it.x];       // (fields are read where they are used: a copy would sit in registers next to 144 accumulators)
        float* slot = (it.y >> 12) ? slots + (int64_t)((it.y >> 12) - 1) * TAPS_SLOT_FLOATS : nullptr;
        const int tile = it.y & 4095, k0 = it.z & 0xFFFF, k1 = (int)((unsigned)it.z >> 16);
        if (it.w != 3) {                    // a per-tap item that rides in this CU's queue (1x1 projections, 8-channel / stride-2 / 4x4 layers)
            const mdm_gemm_desc dc = d;
            if (it.w == 2) wgrad_lin_body<256, 128, 3, 8>(dc, it.y, it.z);
            else if (it.w == 1) wgrad_lin_body<128, 128, 3, 8>(dc, it.y, it.z);
            else wgrad_lin_body<64, 64, 4, 8>(dc, it.y, it.z);
        } else if (d.dbias != nullptr && tile % ((d.N + TAPS_BN - 1) / TAPS_BN) == 0) wgrad_taps_body<true>(d, tile, k0, k1, slot);
        else wgrad_taps_body<false>(d, tile, k0, k1, slot);
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------
// conv_halo: 3x3 stride-1 "same" convolution (forward, and the data gradient through the transposed shadow)
// with the INPUT TILE staged once per 64-channel slab and reused by all 9 filter taps.
// Stamps on conv_lin2 (scripts/stamp_lin2.py) put a 128x128 slab at ~1400 cycles whatever the schedule
// (pipelined, staggered wave groups): 32 KB of operands per slab against a per-CU LDS-DMA intake of
// ~65 GB/s (27 B/clk) IS 1200 cycles -- the loop is bound by bytes fed to the CU, not by MFMA, LDS or L2.
// An im2col gather feeds every input pixel 9 times.  Here a workgroup owns BM = R whole image rows x 64
// output channels: the (R+2) x (W+2) halo of one 64-channel slab is DMA'd once (43 KB for 8 rows of 32)
// and the 9 taps read it at shifted rows; only the 8-KB filter tile changes per tap.  Per tap-slab a wave
// issues ~1.7 DMA pieces instead of 4 (13 KB per 2.1 MFLOP instead of 32 KB).
//   LDS: 2 halo buffers (next slab lands while this one multiplies) + 4-stage ring of filter tiles.
//   A fragment rows = halo rows hb(pixel) + tap shift: a per-lane table of 9 x MI LDS offsets (the
//   XOR swizzle depends on the shifted row) built once.
// Requires: KH = KW = 3, pads 1, stride 1, no upsample, IH = OH, IW = OW in {16, 32, 64}, BM % OW == 0,
// (BM / OW) | OH, C0 % 64 == C1 % 64 == 0, N % 64 == 0.
// ----------------------------------------------------------------------------
// A "group" = TG consecutive filter taps behind one barrier (NG = 9 / TG groups per channel slab).
template <int NPW, int APT, int NG> constexpr int halo_a_count(int g) {          // halo pieces a wave issues in group g (mod NG)
    const int t = ((g % NG) + NG) % NG;
    return t * APT >= NPW ? 0 : (NPW - t * APT < APT ? NPW - t * APT : APT);
}
template <int NPW, int APT, int D, int TG, int GBW = 1> constexpr int halo_vmcnt(int g) {      // DMA operations younger than group g's filter tiles
    int n = (D - 1) * TG * GBW;                                 // (GBW filter pieces per wave and tap: 2 on 128-channel tiles)
    for (int j = 1; j <= D; ++j) n += halo_a_count<NPW, APT, 9 / TG>(g - j);
    return n;
}
// NSB stages of TG filter tiles each, refilled D groups ahead.  TG = 3 (one filter row per barrier) cuts the barriers
// and counted waits to a third: on the small maps a tap is 64 MFMA cycles per wave and the loop ran at ~450 cycles
// per tap, the price of the barrier + wait + the filter-fragment round trip.  The halo pieces of the next channel
// slab are dealt out APT per group over the first groups, so that they are all older than the filter tiles the
// first group of that slab waits for.
// acc[i][j] += B-fragment j x A-fragment i over one k-step (32 bf16 k, or 16 fp32 k as four 16x16x4 MFMAs with the k index
// OUTERMOST: consecutive MFMAs then write different accumulators, none waits for the one in front of it)
template <int MI, int NI>
__device__ __forceinline__ void halo_mma_tile(f32x4 (&acc)[MI][NI], const bf16x8 (&b)[NI], const bf16x8 (&a)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
}
template <int MI, int NI>
__device__ __forceinline__ void halo_mma_tile(f32x4 (&acc)[MI][NI], const f32x4 (&b)[NI], const f32x4 (&a)[MI]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float bv = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
                const float av = c == 0 ? a[i].x : c == 1 ? a[i].y : c == 2 ? a[i].z : a[i].w;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
            }
}
// fp32 operands as bf16 pairs (SPLIT, below): x = hi + lo, a fragment register holds 8 hi (or 8 lo) halves of the same 8 k
__device__ __forceinline__ bf16x8 as_bf16x8(const f32x4& v) { return __builtin_bit_cast(bf16x8, v); }
template <int MI, int NI>
__device__ __forceinline__ void halo_mma_split(f32x4 (&acc)[MI][NI], const f32x4 (&b)[NI], const f32x4 (&a)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b[j]), as_bf16x8(a[i]), acc[i][j], 0, 0, 0);
}
// One 16-byte chunk pair (8 fp32 values: chunks c and c ^ 4 of a 128-byte row) -> 8 bf16 hi halves in the first chunk, 8 bf16 lo halves
// in the second, both round-to-nearest: hi = bf16(x), lo = bf16(x - hi) (the difference is exact in fp32), |x - hi - lo| <= 2^-16 |x|.
// Dword e of the hi (lo) chunk holds element e of the FIRST chunk in its low half and element e of the SECOND chunk in its high half: the
// order of the 8 k inside a fragment is free as long as both operands agree (mdm_split_shadow writes the filters the same way), and this
// one is what v_cvt_pk_bf16_f32 produces -- 6 VALU operations per two elements, no re-interleaving (the element-order version compiled
// to 46 per chunk pair, twice the time of the whole split pass).
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ void split_bf16_pair(f32x4& a, f32x4& b) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = cvt_pk_bf16(a[e], b[e]);
        l[e] = cvt_pk_bf16(a[e] - __uint_as_float(h[e] << 16), b[e] - __uint_as_float(h[e] & 0xffff0000u));
    }
    a = (f32x4){__uint_as_float(h[0]), __uint_as_float(h[1]), __uint_as_float(h[2]), __uint_as_float(h[3])};
    b = (f32x4){__uint_as_float(l[0]), __uint_as_float(l[1]), __uint_as_float(l[2]), __uint_as_float(l[3])};
}
// T = bf16_t: v_mfma_f32_16x16x32_bf16 on 64-channel slabs.  T = float (the exact-fp32 path: the reverse sampler of record): the SAME
// tile, halo, ring and offset tables in bytes -- a 128-byte halo row is then 32 channels, a lane's 16-byte fragment is 4 consecutive
// k that feed four v_mfma_f32_16x16x4_f32 (lane group g supplies k = 4 g + j to MFMA j on both operands, as in gemm_f32_mfma_kernel).
// At a sixteenth of the bf16 matrix rate the loop, not the prologue / epilogue / LDS, sets the time: the kernel is MFMA-bound.
// SPLIT (T = float only; the descriptor's B_split): fp32 storage, products on the bf16 matrix pipe.  Every operand is x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi), and a product is hi*hi + hi*lo + lo*hi (the lo*lo term, <= 2^-16 of the product, is dropped):
// three v_mfma_f32_16x16x32_bf16 (48 matrix cycles per 32 k) instead of eight v_mfma_f32_16x16x4_f32 (256), fp32 accumulation.
// An earlier build split every fragment in registers per wave and tap and was VALU-bound (DESIGN finding 28); here the halo of a
// channel slab is split ONCE, IN PLACE in LDS by all 512 threads during the last filter row of the slab in front of it (chunk g of a
// 128-byte row becomes the 8 hi halves, chunk g ^ 4 the 8 lo halves of the same 8 channels: the fragment addresses do not change),
// and the filter tiles arrive already split from the B_split shadow (mdm_split_shadow: same bytes, same arrangement).
template <int BM, int NPW, int BN = 64, int NSB = 4, int TG = 1, typename T = bf16_t, bool SPLIT = false, bool WT = false>
__device__ __forceinline__ void conv_halo_body(const mdm_gemm_desc& d, char* lds, const int bx, const int gx, const int m_base = 0) {
    static_assert(!SPLIT || sizeof(T) == 4, "conv_halo: SPLIT is the fp32-storage variant");
    constexpr int NW = 8, WR = 4, WC = 2, WM = BM / WR, WN = BN / WC, MI = WM / 16, NI = WN / 16;
    constexpr int NG = 9 / TG;                                    // groups per channel slab
    constexpr int D = TG == 1 ? (NSB == 4 ? 3 : NSB - 2) : NSB - 1;  // refill distance (groups)
    constexpr int APT = (NPW + (NG - D) - 1) / (NG - D);          // halo pieces issued per group
    static_assert(9 % TG == 0 && D >= 1 && D < NG && APT * (NG - D) >= NPW, "conv_halo: halo pieces do not fit in front of the refill distance");
    static_assert(MI >= 1 && NI >= 1, "conv_halo: tile too small for 4 x 2 waves");
    constexpr int B_BYTES = BN * 128, STAGE_B = TG * B_BYTES;
    constexpr int GBW = BN > 64 ? BN / 64 : 1;                    // filter pieces (8 rows of 128 B) per wave and tap
    static_assert(BN <= 64 || BN % 64 == 0, "conv_halo: channel tiles above 64 in steps of 64");
    // DMA operations a wave issues behind the last halo piece of a slab up to the barrier of the slab's last group: the filter
    // tiles of the groups in between (the pieces go out in groups 0 .. (NPW - 1) / APT, behind that group's own filter tiles)
    constexpr int SPLIT_HALO = (NG - 2 - (NPW - 1) / APT) * TG * GBW;
    static_assert(!SPLIT || SPLIT_HALO >= 0, "conv_halo: split schedule");
    constexpr int LAST_NORMAL = halo_vmcnt<NPW, APT, D, TG, GBW>(NG - 1);      // that barrier's own count (its filter tiles have landed)
    constexpr int SPLIT_WAIT = SPLIT_HALO < LAST_NORMAL ? SPLIT_HALO : LAST_NORMAL;
    constexpr int KC = 128 / (int)sizeof(T);                    // channels per slab (one 128-byte halo row)
    constexpr bool F32 = sizeof(T) == 4;
    typedef typename std::conditional<F32, f32x4, bf16x8>::type Frag;
    MDM_T(const unsigned long long t_entry = stamp_now();)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int OW = d.OW, OH = d.OH, HW2 = OW + 2;
    // the tile is R image rows of one image, or (maps of <= BM pixels) IMGS whole images with one halo block each
    // OW and OH are powers of two (halo_tile): every division by them is a shift.  The prologue of this kernel was ~480
    // instructions before its first DMA, most of them eight scalar integer divisions in a dependent chain (~3000 cycles).
    const int ow_sh = __builtin_ctz(OW), p_sh = ow_sh + __builtin_ctz(OH);
    const int IMGS = BM > (1 << p_sh) ? BM >> p_sh : 1, R = IMGS > 1 ? OH : BM >> ow_sh;
    const int HRI = (R + 2) * HW2, HR = IMGS * HRI, NPA = (HR + 7) >> 3, ABUF = NPA * 1024;
    // per-lane index arithmetic below divides by HW2 and HRI (not powers of two): (x + 0.5) * rcp(d) truncated is exact while
    // the 1-ulp error of v_rcp times x / d stays under 0.5 / d: here x < 800, d <= 400
    const float inv_hw2 = __builtin_amdgcn_rcpf((float)HW2), inv_hri = __builtin_amdgcn_rcpf((float)HRI);
    const int rows_sh = __builtin_ctz(R) + ow_sh;                               // R * OW is a power of two
    char* const bring = lds + 2 * ABUF;
    char* const dummy = bring + NSB * STAGE_B;
    const int tiles_n = (d.N + BN - 1) / BN;                     // (N < BN: the last convolution's 8 output channels as one partly filled tile)
    // XCD-aware tile order (xcd_remap): the output-channel tiles of one pixel tile read the same halo and neighbouring
    // pixel tiles the same filter slabs -- handing each XCD a CONTIGUOUS eighth of the tile list lets them meet in ONE L2
    // instead of eight.  No effect on time at cfg2 (the loop is bound by the CU's intake, not by L2 misses); it is there
    // for the L2-side traffic (FETCH_SIZE), which counts every XCD's own fetch of the same line.
    const int bid = xcd_remap(bx, gx);
    const int mt = udiv_small(bid, tiles_n), n0 = (bid - mt * tiles_n) * BN, m0 = m_base + mt * BM;   // (m_base: conv_halo_mixed_kernel)
    const int img = m0 >> p_sh, y0 = (m0 >> ow_sh) & (OH - 1);
    const char* zlane = reinterpret_cast<const char*>(g_zero_page) + lane * 16;
    const int NCS = d.Ck / KC;                       // channel slabs over both sources
    const int sgn = d.transposed ? -1 : 1;

    // ---- halo pieces of this wave: piece p = wave + 8 k holds halo rows 8p .. 8p+7
    int apix[NPW];
    const int lch16 = ((lane & 7) ^ (lane >> 3)) << 4;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int hr = (wave + 8 * k) * 8 + (lane >> 3);
        const int il = (int)(((float)hr + 0.5f) * inv_hri), hrem = hr - il * HRI;
        const int hy = (int)(((float)hrem + 0.5f) * inv_hw2), hx = hrem - hy * HW2;
        const int y = y0 - 1 + hy, x = hx - 1;
        const bool ok = hr < HR && (unsigned)y < (unsigned)OH && (unsigned)x < (unsigned)OW && m0 < d.M;
        // folded nearest x2 upsample (unet6.py:472): the map the conv sees is virtual, pixel (y, x) lives at (y>>1, x>>1)
        apix[k] = ok ? (((img + il) * (OH >> d.ups) + (y >> d.ups)) * (OW >> d.ups) + (x >> d.ups)) : -1;
    }
    auto issue_a = [&](int k, int cs, char* abuf) {                 // k compile-time after unrolling
        const int p = wave + 8 * k;
        const int c = cs * KC;
        const bool s1 = c >= d.C0;
        const T* S = reinterpret_cast<const T*>(s1 ? d.src1 : d.src0);
        const int ld = s1 ? d.ld1 : d.ld0, cc = s1 ? c - d.C0 : c;
        const char* src = reinterpret_cast<const char*>(S + (int64_t)apix[k] * ld + cc) + lch16;
        lds_dma16((apix[k] >= 0 && cs < NCS) ? src : zlane, p < NPA ? abuf + p * 1024 : dummy);
    };
    // ---- filter tile of (tap, slab): BN rows (output channels) x 128 B, one piece per wave (BN = 32: waves 4..7
    // issue into the dummy page so that every wave counts the same number of DMA operations)
    const bool b_wave = wave * 8 < BN;
    const int bn = n0 + (b_wave ? wave * 8 : 0) + (lane >> 3);
    const char* b_row = reinterpret_cast<const char*>(reinterpret_cast<const T*>(SPLIT ? d.B_split : d.B) + (int64_t)bn * d.ldb) + lch16;
    const bool b_live = b_wave && bn < d.N;                          // filter rows beyond N (a partly filled channel tile) are zeros
    auto issue_b = [&](int tap, int cs, int lds_off) {               // lds_off: byte offset of the tile inside the ring
        const int64_t off = ((int64_t)tap * d.wtap + (int64_t)cs * KC) * (int64_t)sizeof(T);
        lds_dma16((cs < NCS && b_live) ? b_row + off : zlane, b_wave ? bring + lds_off + wave * 1024 : dummy);
#pragma unroll
        for (int q = 1; q < GBW; ++q)                                // 128-channel tiles: rows 64 q + 8 wave .. of the tile (N % BN == 0 there)
            lds_dma16(cs < NCS ? b_row + off + (int64_t)q * 64 * d.ldb * (int64_t)sizeof(T) : zlane, bring + lds_off + (wave + 8 * q) * 1024);
    };

    // ---- prologue: halo of slab 0, filter tiles of tap-slabs 0..D-1 (in flight while the offset tables below are built)
#pragma unroll
    for (int k = 0; k < NPW; ++k) issue_a(k, 0, lds);
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
        for (int k = 0; k < TG; ++k) issue_b((u % NG) * TG + k, u / NG, u * STAGE_B + k * B_BYTES);

    // ---- per-lane fragment offsets: a table of 9 x MI addresses -- or, on the tiles whose accumulators and fragments leave no room for
    // it (128 channels: 16 accumulator tiles per wave; the table cost 36 registers and the kernel spilled), the MI halo rows of the
    // centre tap only, the tap's row shift and the swizzle being four VALU operations per address under 48 MFMAs per tap
    constexpr bool A_TABLE = MI * NI < 16;
    int a_addr[A_TABLE ? 9 : 1][MI];
    int a_hb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wr * WM + i * 16 + (lane & 15);
        const int il = ml >> rows_sh, mrem = ml - (il << rows_sh);
        const int r = mrem >> ow_sh, x = mrem - (r << ow_sh);
        const int hb = il * HRI + (r + 1) * HW2 + x + 1;
        a_hb[i] = hb;
#pragma unroll
        for (int tp = 0; tp < (A_TABLE ? 9 : 1); ++tp) {
            const int ty = tp / 3, tx = tp - ty * 3;
            const int hr = hb + sgn * (ty - 1) * HW2 + sgn * (tx - 1);
            a_addr[tp][i] = hr * 128 + (((lane >> 4) ^ (hr & 7)) << 4);
        }
    }
    const int lane_g = lane >> 4;
    auto a_at = [&](int tp, int i) -> int {                         // tp is a constant after unrolling
        if constexpr (A_TABLE) return a_addr[tp][i];
        else {
            const int ty = tp / 3, tx = tp - ty * 3;
            const int hr = a_hb[i] + sgn * ((ty - 1) * HW2 + (tx - 1));
            return hr * 128 + ((lane_g ^ (hr & 7)) << 4);
        }
    };
    int b_off[2][NI];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = wc * WN + j * 16 + (lane & 15);
            b_off[ks][j] = nl * 128 + (((ks * 4 + (lane >> 4)) ^ (nl & 7)) << 4);
        }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // SPLIT: halo rows [0, HR) of `abuf` from fp32 to (hi, lo) bf16 halves in place; item = (row, chunk pair g / g ^ 4), one thread each
    auto split_halo = [&](char* abuf) {
        auto slot = [&](int id, bool second) {
            const int hr = id >> 2, pg = ((id & 3) ^ hr) & 7;          // chunk c of row hr sits at position c ^ (hr & 7)
            return reinterpret_cast<f32x4*>(abuf + hr * 128 + ((second ? pg ^ 4 : pg) << 4));   // chunk g: k = 4 g ..; chunk g + 4: k = 16 + 4 g ..
        };
        int id = t;
        for (; id + NW * 64 < HR * 4; id += 2 * NW * 64) {             // two items per thread with their four reads in flight together
            f32x4 *pa0 = slot(id, false), *pb0 = slot(id, true), *pa1 = slot(id + NW * 64, false), *pb1 = slot(id + NW * 64, true);
            f32x4 va0 = *pa0, vb0 = *pb0, va1 = *pa1, vb1 = *pb1;
            split_bf16_pair(va0, vb0);
            *pa0 = va0; *pb0 = vb0;                                    // 8 hi halves (what the k-step-0 fragment address reads), 8 lo halves
            split_bf16_pair(va1, vb1);
            *pa1 = va1; *pb1 = vb1;
        }
        if (id < HR * 4) {
            f32x4 *pa = slot(id, false), *pb = slot(id, true);
            f32x4 va = *pa, vb = *pb;
            split_bf16_pair(va, vb);
            *pa = va; *pb = vb;
        }
    };
    if constexpr (SPLIT) {                 // slab 0: its pieces are older than the D * TG filter tiles of the prologue
        wait_vmcnt<D * TG * GBW>();
        __builtin_amdgcn_s_barrier();
        split_halo(lds);
    }

    int a_cur = 0;                         // byte offset of the halo buffer being multiplied
    int b_stage = 0;                       // ring stage of the current tap-slab
    const int a_flip = ABUF;               // the two halo buffers sit at offsets 0 and ABUF: a_cur toggles between them
#define MDM_HALO_TAP(T)                                                                                          \
    {                                                                                                            \
        if ((T) % TG == 0) {                                                                                     \
            /* the filter tiles of this group have landed (issued D groups ago), and everything older */         \
            /* SPLIT: behind the barrier of the LAST group the whole halo of the next slab must have landed too   \
               (SPLIT_WAIT: what was issued behind its last piece), so that it can be split during this group;    \
               the split's LDS writes are complete before the barrier that opens the next slab */                 \
            MDM_T(const unsigned long long tw0 = stamp_now();)                                                   \
            if (SPLIT && (T) / TG == NG - 1) wait_vmcnt<SPLIT_WAIT>();                                           \
            else wait_vmcnt<halo_vmcnt<NPW, APT, D, TG, GBW>((T) / TG)>();                                       \
            if (SPLIT && (T) == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
            MDM_T(const unsigned long long tw1 = stamp_now(); t_wait += tw1 - tw0;)                              \
            __builtin_amdgcn_s_barrier();                                                                        \
            MDM_T(t_bar += stamp_now() - tw1;)                                                                   \
            const int rs = b_stage + D >= NSB ? b_stage + D - NSB : b_stage + D;                                 \
            /* (dealing these DMA pieces out between the MFMAs of the tap, as conv_lin2 does, changed neither the  \
               bf16 step -- 3.904 vs 3.908 ms -- nor the fp32 sampler -- 14.38 vs 14.37 s: measured, not kept) */  \
            _Pragma("unroll") for (int k = 0; k < TG; ++k)                                                       \
                issue_b((((T) / TG + D) % NG) * TG + k, cs + ((T) / TG + D) / NG, rs * STAGE_B + k * B_BYTES);   \
            _Pragma("unroll") for (int q = 0; q < APT; ++q)                                                      \
                if (((T) / TG) * APT + q < NPW) issue_a(((T) / TG) * APT + q, cs + 1, lds + (a_cur ^ a_flip));   \
            if constexpr (SPLIT) {                                                                               \
                MDM_T(const unsigned long long ts0 = stamp_now();)                                               \
                if ((T) / TG == NG - 1 && cs + 1 < NCS) split_halo(lds + (a_cur ^ a_flip));                      \
                MDM_T(t_split += stamp_now() - ts0;)                                                             \
            }                                                                                                    \
        }                                                                                                        \
        const char* As = lds + a_cur;                                                                            \
        const char* Bs = bring + b_stage * STAGE_B + ((T) % TG) * B_BYTES;                                       \
        Frag bfr[2][NI];                                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                       \
                bfr[ks][j] = *reinterpret_cast<const Frag*>(Bs + b_off[ks][j]);                                  \
        if ((T) == 0 || !A_TABLE) { /* a new halo buffer: its fragments can only be read behind this barrier (no table = no   \
                                       second fragment set either: this tap's fragments are fetched at its top) */       \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
                    afr[A_TABLE ? 0 : AF(T)][ks][i] = *reinterpret_cast<const Frag*>(As + (a_at((T), i) ^ (ks << 6))); \
        }                                                                                                        \
        if constexpr (SPLIT) halo_mma_split<MI, NI>(acc, bfr[0], afr[AF(T)][1]);        /* b.hi x a.lo */       \
        else halo_mma_tile<MI, NI>(acc, bfr[0], afr[AF(T)][0]);                                                  \
        if ((T) < 8 && A_TABLE) {  /* the next tap reads the SAME halo buffer at shifted rows: fetch it now */    \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                   \
                    afr[AF((T) + 1)][ks][i] = *reinterpret_cast<const Frag*>(As + (a_at(((T) + 1) % 9, i) ^ (ks << 6))); \
        }                                                                                                        \
        if constexpr (SPLIT) {                                                                                   \
            halo_mma_split<MI, NI>(acc, bfr[1], afr[AF(T)][0]);                         /* b.lo x a.hi */       \
            halo_mma_split<MI, NI>(acc, bfr[0], afr[AF(T)][0]);                         /* b.hi x a.hi */       \
        } else halo_mma_tile<MI, NI>(acc, bfr[1], afr[AF(T)][1]);                                                \
        if ((T) % TG == TG - 1) b_stage = b_stage + 1 == NSB ? 0 : b_stage + 1;                                  \
    }
#define AF(T) (A_TABLE ? ((T) & 1) : 0)
    Frag afr[A_TABLE ? 2 : 1][2][MI];      // [tap parity][k-step][fragment]: tap T multiplies set T&1 while set (T+1)&1 is fetched
    MDM_T(unsigned long long t_wait = 0, t_bar = 0, t_split = 0;)
    MDM_T(const unsigned long long tstart = stamp_now();)
    for (int cs = 0; cs < NCS; ++cs) {
        MDM_HALO_TAP(0) MDM_HALO_TAP(1) MDM_HALO_TAP(2) MDM_HALO_TAP(3) MDM_HALO_TAP(4)
        MDM_HALO_TAP(5) MDM_HALO_TAP(6) MDM_HALO_TAP(7) MDM_HALO_TAP(8)
        a_cur = a_cur == 0 ? ABUF : 0;
    }
#undef MDM_HALO_TAP
#undef AF
    wait_vmcnt<0>();
    MDM_T(const unsigned long long t_loop_end = stamp_now();)
    __syncthreads();
    if constexpr (!F32 && BM == 64 && (BN == 64 || BN == 32)) {
        if (d.gnb_x) epilogue_tile_gnb<MI, NI, BN, WT>(d, lds, m0, n0, wr * WM, wc * WN, lane, t, acc);      // uniform
        else if (d.gnf_out) epilogue_tile_gnf<MI, NI, BN, WT>(d, lds, m0, n0, wr * WM, wc * WN, lane, t, acc);
        else epilogue_tile<BM, BN, NW, MI, NI, bf16_t, WT>(d, lds, m0, n0, wr * WM, wc * WN, lane, t, acc);
    } else {
        epilogue_tile<BM, BN, NW, MI, NI, T, WT>(d, lds, m0, n0, wr * WM, wc * WN, lane, t, acc);
    }
#ifdef MDM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const unsigned widx = (unsigned)bx * NW + wave;
        if (widx < 4096) {
            unsigned long long* r = g_stamp_buf + widx * 32;
            r[0] = t_wait; r[1] = t_bar; r[2] = t_split; r[3] = 0; r[4] = NCS * 9; r[5] = 1; r[6] = t_loop_end - tstart; r[7] = tstart;
            r[8] = tstart - t_entry; r[9] = stamp_now() - t_loop_end; r[10] = t_entry;
        }
    }
#endif
}

template <int BM, int NPW, int BN = 64, int NSB = 4, int TG = 1, typename T = bf16_t, bool SPLIT = false>
__global__ __launch_bounds__(512) void conv_halo_kernel(mdm_gemm_desc d) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    conv_halo_body<BM, NPW, BN, NSB, TG, T, SPLIT>(d, lds, (int)blockIdx.x, (int)gridDim.x);
}



// ----------------------------------------------------------------------------
// conv_small: the 3x3 stride-1 "same" convolutions of the 4x4 / 8x8 maps (forward incl. the folded x2 upsample, and the data gradient
// through the transposed shadow) with the REDUCTION split over the waves.  On conv_halo_body's 64 x 32 tiles of these maps every wave
// owns ONE 16 x 16 accumulator: per filter tap it reads four fragments for two MFMAs that wait for each other, and the in-kernel
// stamps put the loop at 430 cycles per tap (15 500 cycles for a 256 -> 256 layer: 14 B/clk of operands against the CU's ~27 B/clk
// intake, 7 % of the matrix pipe; profiles/r04_chain_stamps.txt).  Here the tile is the same (64 pixels = whole images x 32 output
// channels, so the fused GroupNorm epilogues apply unchanged) but a channel "superslab" is 128 channels = four 32-deep k-steps:
//   wave = (ks, ph): k-step ks of the superslab, pixel half ph -> 2 x 2 accumulator tiles per wave, four INDEPENDENT MFMAs per tap
//   from four fragments, 12 MFMAs per barrier (3 taps); the 8 waves' partial sums meet in LDS once, behind the loop, in a fixed order.
//   LDS: two halo buffers (the (OH+2) x (OW+2) halo of a 128-channel superslab, 256-byte pixel rows, 16-byte chunks XOR-swizzled by
//   the row), a three-stage ring of filter stages (3 taps x 32 rows x 256 B = 24 KiB each), refilled two groups ahead; the halo of
//   superslab s+1 arrives during the first two groups of superslab s.  Every wave issues the same number of LDS-DMA operations per
//   group (missing pieces go to a dummy page), so the waits are compile-time vmcnt counts.
// Requires what halo_tile() == 64 requires, and C0 % 128 == C1 % 128 == 0, N % 32 == 0.
// ----------------------------------------------------------------------------
#ifndef MDM_SMALL_EPI
#define MDM_SMALL_EPI 1             // 0: conv_small keeps the LDS epilogues of the halo tiles (A/B builds)
#endif
// BM x BN: 64 x 32, or 32 x 16 on the 4x4 maps (two whole images x two GroupNorm groups: 256 workgroups instead of 64, each streaming a
// quarter of the bytes -- the loop is bound by what ONE CU takes in)
template <int NPW, bool WT = false, int BM = 64, int BN = 32>
__device__ __forceinline__ void conv_small_body(const mdm_gemm_desc& d, char* lds, const int bx, const int gx) {
    constexpr int CS = 128, NSB = 3, MI = BM / 32, NI = BN / 16, BPT = BN / 4;     // BPT: 1-KiB pieces of one tap's filter tile
    static_assert((BM == 64 && BN == 32) || (BM == 32 && BN == 16), "conv_small: 64 x 32 or 32 x 16 tiles");
    constexpr int ROWB = CS * 2, B_TAP = BN * ROWB, STAGE_B = 3 * B_TAP;
    constexpr int HA = (NPW + 1) / 2, HB = NPW - HA;                // halo pieces a wave issues in group 0 / group 1 of a superslab
    MDM_T(const unsigned long long t_entry = stamp_now();)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ks = wave & 3, ph = wave >> 2;
    const int OW = d.OW, OH = d.OH, HW2 = OW + 2;
    const int ow_sh = __builtin_ctz(OW), p_sh = ow_sh + __builtin_ctz(OH);
    const int IMGS = BM >> p_sh;                                     // whole images per tile: 4 / 2 (4x4) or 1 (8x8)
    const int HRI = (OH + 2) * HW2, HR = IMGS * HRI, NPA = (HR + 3) >> 2, ABUF = NPA * 1024;
    const float inv_hw2 = __builtin_amdgcn_rcpf((float)HW2), inv_hri = __builtin_amdgcn_rcpf((float)HRI);
    char* const bring = lds + 2 * ABUF;
    char* const dummy = bring + NSB * STAGE_B;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int bid = xcd_remap(bx, gx);
    const int mt = udiv_small(bid, tiles_n), n0 = (bid - mt * tiles_n) * BN, m0 = mt * BM;
    const int img0 = m0 >> p_sh;
    const char* zlane = reinterpret_cast<const char*>(g_zero_page) + lane * 16;
    const int NSS = d.Ck / CS;
    const int sgn = d.transposed ? -1 : 1;

    // ---- halo pieces of this wave: piece p = wave + 8 k holds halo rows 4p .. 4p+3 (256 B each); lane -> (row, chunk position)
    int apix[NPW], a_lch[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int hr = (wave + 8 * k) * 4 + (lane >> 4);
        const int il = (int)(((float)hr + 0.5f) * inv_hri), hrem = hr - il * HRI;
        const int hy = (int)(((float)hrem + 0.5f) * inv_hw2), hx = hrem - hy * HW2;
        const int y = hy - 1, x = hx - 1;
        const bool ok = hr < HR && (unsigned)y < (unsigned)OH && (unsigned)x < (unsigned)OW && m0 < d.M;
        apix[k] = ok ? (((img0 + il) * (OH >> d.ups) + (y >> d.ups)) * (OW >> d.ups) + (x >> d.ups)) : -1;
        a_lch[k] = ((lane & 15) ^ (hr & 15)) << 4;                  // position c of row hr holds source chunk c ^ (hr & 15)
    }
    auto issue_a = [&](int k, int ss, char* abuf) {                 // k compile-time after unrolling
        const int p = wave + 8 * k;
        const int c = ss * CS;
        const bool s1 = c >= d.C0;
        const bf16_t* S = reinterpret_cast<const bf16_t*>(s1 ? d.src1 : d.src0);
        const int ld = s1 ? d.ld1 : d.ld0, cc = s1 ? c - d.C0 : c;
        const char* src = reinterpret_cast<const char*>(S + (int64_t)apix[k] * ld + cc) + a_lch[k];
        lds_dma16((apix[k] >= 0 && ss < NSS) ? src : zlane, p < NPA ? abuf + p * 1024 : dummy);
    };
    // ---- filter tile of (tap, superslab): 32 rows (output channels) x 256 B = 8 pieces, one per wave
    const bool b_wave = wave < BPT;                                 // (16-channel tiles: waves 4 .. 7 issue into the dummy page, the counts stay uniform)
    const int bn_l = b_wave ? 4 * wave + (lane >> 4) : 0;
    const bool b_live = b_wave && n0 + bn_l < d.N;
    const char* b_row = reinterpret_cast<const char*>(reinterpret_cast<const bf16_t*>(d.B) + (int64_t)(n0 + bn_l) * d.ldb) +
                        (((lane & 15) ^ (bn_l & 15)) << 4);
    auto issue_b = [&](int tap, int ss, int lds_off) {
        const int64_t off = ((int64_t)tap * d.wtap + (int64_t)ss * CS) * 2;
        const bool live = ss < NSS && b_live;
        lds_dma16(live ? b_row + off : zlane, (ss < NSS && b_wave) ? bring + lds_off + wave * 1024 : dummy);
    };

    // ---- prologue: halo of superslab 0, filter groups 0 and 1
#pragma unroll
    for (int k = 0; k < NPW; ++k) issue_a(k, 0, lds);
#pragma unroll
    for (int k = 0; k < 3; ++k) issue_b(k, 0, k * B_TAP);
#pragma unroll
    for (int k = 0; k < 3; ++k) issue_b(3 + k, 0, STAGE_B + k * B_TAP);

    // ---- fragment addresses: centre-tap halo row of this lane's pixel in each of its two 16-row blocks; filter rows
    const int kch = ks * 4 + (lane >> 4);                           // 16-byte chunk of this lane inside a 256-byte row
    int a_hb[MI], b_off[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = ph * (BM / 2) + i * 16 + (lane & 15);
        const int il = ml >> p_sh, mrem = ml - (il << p_sh);
        const int r = mrem >> ow_sh, x = mrem - (r << ow_sh);
        a_hb[i] = il * HRI + (r + 1) * HW2 + x + 1;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nl = j * 16 + (lane & 15);
        b_off[j] = nl * ROWB + ((kch ^ (nl & 15)) << 4);
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int stage = 0;
    MDM_T(unsigned long long t_wait = 0, t_bar = 0;)
    MDM_T(const unsigned long long tstart = stamp_now();)
#define MDM_SMALL_GROUP(G3)                                                                                        \
    {                                                                                                              \
        MDM_T(const unsigned long long tw0 = stamp_now();)                                                         \
        wait_vmcnt<(G3) == 0 ? 3 : ((G3) == 1 ? 3 + HA : 3 + NPW)>();                                              \
        MDM_T(const unsigned long long tw1 = stamp_now(); t_wait += tw1 - tw0;)                                    \
        __builtin_amdgcn_s_barrier();                                                                              \
        MDM_T(t_bar += stamp_now() - tw1;)                                                                         \
        const int rs = stage + 2 >= NSB ? stage + 2 - NSB : stage + 2;                                             \
        _Pragma("unroll") for (int k = 0; k < 3; ++k)                                                              \
            issue_b((((G3) + 2) % 3) * 3 + k, ss + ((G3) + 2) / 3, rs * STAGE_B + k * B_TAP);                      \
        if ((G3) == 0) { _Pragma("unroll") for (int q = 0; q < HA; ++q) issue_a(q, ss + 1, anext); }               \
        if ((G3) == 1) { _Pragma("unroll") for (int q = 0; q < HB; ++q) issue_a(HA + q, ss + 1, anext); }          \
        const char* Bs = bring + stage * STAGE_B;                                                                  \
        bf16x8 af[3][MI], bfr[3][NI];                                                                              \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                            \
            const int tx = k, ty = (G3);                                                                           \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                         \
                bfr[k][j] = *reinterpret_cast<const bf16x8*>(Bs + k * B_TAP + b_off[j]);                           \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                       \
                const int hr = a_hb[i] + sgn * ((ty - 1) * HW2 + (tx - 1));                                        \
                af[k][i] = *reinterpret_cast<const bf16x8*>(acur + hr * ROWB + ((kch ^ (hr & 15)) << 4));          \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 3; ++k)                                                              \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[k][j], af[k][i], acc[i][j], 0, 0, 0);  \
        stage = stage + 1 == NSB ? 0 : stage + 1;                                                                  \
    }
    for (int ss = 0; ss < NSS; ++ss) {
        const char* const acur = lds + (ss & 1) * ABUF;
        char* const anext = lds + ((ss + 1) & 1) * ABUF;
        MDM_SMALL_GROUP(0) MDM_SMALL_GROUP(1) MDM_SMALL_GROUP(2)
    }
#undef MDM_SMALL_GROUP
    wait_vmcnt<0>();
    MDM_T(const unsigned long long t_loop_end = stamp_now(); unsigned long long t_e1 = 0, t_e2 = 0, t_e3 = 0;)
    __syncthreads();
    // ---- the waves' partial sums meet in LDS: wave (ks, ph) parks its 2 x 2 tiles, wave (wr, wc) of the epilogue layout adds the four
    // k-steps of ITS 16 x 16 tile in ascending order (fixed summation order); the scratch sits behind the epilogue's fp32 tile
    {
        f32x4* red = reinterpret_cast<f32x4*>(lds + 16384);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) red[(wave * (MI * NI) + i * NI + j) * 64 + lane] = acc[i][j];
        __syncthreads();
        MDM_T(t_e1 = stamp_now();)
        // C / G == 8 (or no fused GroupNorm): the register epilogue, four waves; anything else: the LDS epilogues of the halo tiles
        const int cpg = d.gnb_x ? d.N / d.gnb_G : (d.gnf_out ? d.N / d.gnf_G : 8);
        if ((BM != 64 || MDM_SMALL_EPI) && cpg == 8 && (d.N0 & 7) == 0) {                                           // uniform
            if (wave < BN / 8) {                                   // one wave per group of eight channels; lane = pixel of the tile
                float v8[8];
                const bool act = lane < BM;
                rows_gather_small<MI, NI>(red, wave, act ? lane : 0, v8);
                MDM_T(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_e2 = stamp_now();)
                if (p_sh == 4) epilogue_rows<WT, true>(d, v8, m0, n0, wave, lane, act);
                else epilogue_rows<WT, false>(d, v8, m0, n0, wave, lane, act);
                MDM_T(t_e3 = stamp_now();)
            }
        } else if constexpr (BM == 64 && BN == 32) {
        const int wr = wave >> 1, wc = wave & 1;
        const int src = (wr >> 1) * 4, tile = (wr & 1) * 2 + wc;
        f32x4 v = red[((src + 0) * 4 + tile) * 64 + lane];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const f32x4 o = red[((src + q) * 4 + tile) * 64 + lane];
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
        f32x4 one[1][1];
        one[0][0] = v;
        if (d.gnb_x) epilogue_tile_gnb<1, 1, BN, WT>(d, lds, m0, n0, wr * 16, wc * 16, lane, t, one);          // uniform
        else if (d.gnf_out) epilogue_tile_gnf<1, 1, BN, WT>(d, lds, m0, n0, wr * 16, wc * 16, lane, t, one);
        else epilogue_tile<64, BN, 8, 1, 1, bf16_t, WT>(d, lds, m0, n0, wr * 16, wc * 16, lane, t, one);
        }
    }
#ifdef MDM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const unsigned widx = (unsigned)bx * 8 + wave;
        if (widx < 4096) {
            unsigned long long* r = g_stamp_buf + widx * 32;
            r[0] = t_wait; r[1] = t_bar; r[2] = 0; r[3] = 0; r[4] = NSS * 9; r[5] = 1; r[6] = t_loop_end - tstart; r[7] = tstart;
            r[8] = tstart - t_entry; r[9] = stamp_now() - t_loop_end; r[10] = t_entry;
            r[11] = t_e1 - t_loop_end; r[12] = t_e2 ? t_e2 - t_e1 : 0; r[13] = t_e3 ? t_e3 - t_e2 : 0; r[14] = t_e3 ? stamp_now() - t_e3 : 0;
        }
    }
#endif
}

template <int NPW, int BM = 64, int BN = 32>
__global__ __launch_bounds__(512) void conv_small_kernel(mdm_gemm_desc d) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    conv_small_body<NPW, false, BM, BN>(d, lds, (int)blockIdx.x, (int)gridDim.x);
}
// conv_small + a 1x1 projection in one launch (the pairs of mdm_gemm_pair on the 4x4 / 8x8 maps)
template <int NPW, int BM = 64, int BN = 32>
__global__ __launch_bounds__(512) void conv_pair_small_kernel(mdm_gemm_desc a, mdm_gemm_desc b, int na) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    if ((int)blockIdx.x < na) conv_small_body<NPW, false, BM, BN>(a, lds, (int)blockIdx.x, na);
    else conv_lin2_body<64, 64, 4, 4, 2, 1, true, false>(b, lds, (int)blockIdx.x - na, (int)gridDim.x - na, 0);
}

// Two tile sizes in one launch (split products, sample_num = 100): a 32x32 layer is 800 tiles of 256 pixels on 256 CUs -- 3.1 rounds of
// work in 4 rounds of workgroups, the last one on 32 CUs.  Here the first `n_big` workgroups take whole rounds of 256-pixel tiles and
// the rest of the pixels go out as 128-pixel tiles (0.6 of a big tile's time) behind them: 3 rounds + a short one.
template <int NPW_A, int NSB_A, int NPW_B, int NSB_B>
__global__ __launch_bounds__(512) void conv_halo_mixed_kernel(mdm_gemm_desc d, int n_big, int m_split) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    if ((int)blockIdx.x < n_big) conv_halo_body<256, NPW_A, 64, NSB_A, 3, float, true>(d, lds, (int)blockIdx.x, n_big);
    else conv_halo_body<128, NPW_B, 64, NSB_B, 3, float, true>(d, lds, (int)blockIdx.x - n_big, (int)gridDim.x - n_big, m_split);
}

// ----------------------------------------------------------------------------
// lin_split: 1x1 stride-1 convolution over one or two fp32 NHWC sources (a ResidualBlock's skip projection over the concatenated
// decoder input, the attention block's project_in / project_out: unet6.py:296-333, 350-362) with SPLIT products -- the companion of
// conv_halo_body<..., SPLIT> for the layers that have no halo to reuse.  On the register-staged fp32 kernel these launches were a
// load -> store -> barrier chain per 32-deep slab (66 TFLOP/s at sample_num = 100, three times their HBM time).
//   tile 128 pixels x BN channels, 8 waves as 4 x 2; per 32-channel slab both operand tiles arrive by LDS-DMA in a four-stage ring
//   (slab s in stage s & 3): while slab s multiplies, slab s+1 (landed) is split in place -- 128 rows x 4 chunk pairs = one item per
//   thread --, slab s+2 is in flight and slab s+3 is issued into the stage slab s-1 has left.  ONE barrier per slab.  Filters come
//   already split from B_split.  Same swizzle, chunk pairing (g, g ^ 4) and product order as the halo kernel; epilogue_tile.
// Also the stride-2 3x3 convolutions (SamePad2d + stride 2, unet6.py:257-272): a filter tap is a constant source-pixel offset and a
// per-row validity bit, the slab loop runs tap-major over taps x channel slabs (on the register-staged kernel those three launches
// took 355 us of a 6.3 ms reverse step).  Rows beyond M read the zero page and are skipped by the epilogue.
// Requires: KH = KW in {1, 3}, stride 1 (1x1) or 2, no upsample, C0 % 32 == C1 % 32 == 0, N % BN == 0.
// ----------------------------------------------------------------------------
template <int BN, int NS>
__global__ __launch_bounds__(512) void lin_split_kernel(mdm_gemm_desc d) {
    constexpr int BM = 128, NW = 8, WR = 4, WC = 2, WM = BM / WR, WN = BN / WC, MI = WM / 16, NI = WN / 16;
    // NS ring stages (slab s in stage s % NS): slab s multiplies, s+1 is split, s+2 .. s+NS-2 are in flight, s+NS-1 is issued into the
    // stage slab s-1 left.  Four stages on both tiles (a deeper ring on the 64-channel tile -- six stages of 24 KB -- was measured
    // slower: its layers pay the longer prologue, not the latency of the slabs in flight)
    constexpr int A_ST = BM * 128, B_ST = BN * 128, GA = BM / 64, GB = BN / 64, G = GA + GB;
    static_assert(NS >= 4, "lin_split: read / split / in flight / issue need four stages");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* const aring = lds;
    char* const bring = lds + NS * A_ST;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int tiles_n = d.N / BN;
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int mt = udiv_small(bid, tiles_n), n0 = (bid - mt * tiles_n) * BN, m0 = mt * BM;
    const int NCS = d.Ck >> 5, C0 = d.C0, KW = d.KW, NSL = d.KH * d.KW * NCS;     // slabs: filter tap major, 32-channel slab minor
    const char* zlane = reinterpret_cast<const char*>(g_zero_page) + lane * 16;
    const int lch16 = ((lane & 7) ^ (lane >> 3)) << 4;
    // row m = output pixel (img, oy, ox); filter tap (ty, tx) reads source pixel (oy * stride - pad_t + ty, ox * stride - pad_l + tx):
    // one pointer per row for tap (0, 0), a constant pixel offset per tap, and a 9-bit mask of the taps that fall inside the image
    const char* a0[GA];
    const char* a1[GA];
    unsigned tapmask[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int m = m0 + 8 * (wave * GA + j) + (lane >> 3);
        const RowPix rp = decode_row(d, m < d.M ? m : 0);
        const int iy0 = rp.oy * d.stride - d.pad_t, ix0 = rp.ox * d.stride - d.pad_l;
        const int64_t pix0 = ((int64_t)rp.img * d.IH + iy0) * d.IW + ix0;
        unsigned mk = 0;
        for (int tp = 0; tp < d.KH * KW; ++tp) {
            const int ty = tp / KW, tx = tp - ty * KW;
            if (m < d.M && (unsigned)(iy0 + ty) < (unsigned)d.IH && (unsigned)(ix0 + tx) < (unsigned)d.IW) mk |= 1u << tp;
        }
        tapmask[j] = mk;
        a0[j] = reinterpret_cast<const char*>(reinterpret_cast<const float*>(d.src0) + pix0 * d.ld0) + lch16;
        a1[j] = d.C1 ? reinterpret_cast<const char*>(reinterpret_cast<const float*>(d.src1) + pix0 * d.ld1) + lch16 : zlane;
    }
    const int ld0b = d.ld0 * 4, ld1b = d.ld1 * 4, IWs = d.IW;
    const int64_t wtapb = d.wtap * 4;
    const char* b0[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int64_t n = n0 + 8 * (wave * GB + j) + (lane >> 3);
        b0[j] = reinterpret_cast<const char*>(reinterpret_cast<const float*>(d.B_split) + n * d.ldb) + lch16;
    }
    int is_tap = 0, is_cs = 0, is_ty = 0, is_tx = 0;                // (tap, slab) of the next slab to issue: issue() is called in slab order
    auto issue = [&](int sl) {                                      // every wave issues G operations per slab, live or not
        const int st = sl % NS, c = is_cs << 5;
        const bool live = sl < NSL, s1 = c >= C0;
        const int poff = is_ty * IWs + is_tx;                       // source pixels between tap (0, 0) and this tap
#pragma unroll
        for (int j = 0; j < GA; ++j) {
            const bool ok = live && ((tapmask[j] >> is_tap) & 1u);
            const char* src = s1 ? a1[j] + (int64_t)poff * ld1b + (int64_t)(c - C0) * 4 : a0[j] + (int64_t)poff * ld0b + (int64_t)c * 4;
            lds_dma16(ok ? src : zlane, aring + st * A_ST + (wave * GA + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) lds_dma16(live ? b0[j] + is_tap * wtapb + (int64_t)c * 4 : zlane, bring + st * B_ST + (wave * GB + j) * 1024);
        if (++is_cs == NCS) { is_cs = 0; ++is_tap; if (++is_tx == KW) { is_tx = 0; ++is_ty; } }
    };
    auto split_a = [&](int cs) {                                    // rows x chunk pairs of stage cs & 3: one item per thread
        char* abuf = aring + (cs % NS) * A_ST;
        const int hr = t >> 2, pg = ((t & 3) ^ hr) & 7;
        f32x4* pa = reinterpret_cast<f32x4*>(abuf + hr * 128 + (pg << 4));
        f32x4* pb = reinterpret_cast<f32x4*>(abuf + hr * 128 + ((pg ^ 4) << 4));
        f32x4 va = *pa, vb = *pb;
        split_bf16_pair(va, vb);
        *pa = va; *pb = vb;
    };
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) issue(u);
    int a_off[MI], b_off[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wr * WM + i * 16 + (lane & 15);
        a_off[i] = ml * 128 + (((lane >> 4) ^ (ml & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nl = wc * WN + j * 16 + (lane & 15);
        b_off[j] = nl * 128 + (((lane >> 4) ^ (nl & 7)) << 4);
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wait_vmcnt<(NS - 2) * G>();                                     // slab 0 has landed
    __builtin_amdgcn_s_barrier();
    split_a(0);
    for (int s = 0; s < NSL; ++s) {
        wait_vmcnt<(NS - 3) * G>();                                 // slabs <= s+1 have landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the split of slab s is written
        __builtin_amdgcn_s_barrier();                               // ... for every wave; slab s-1 is no longer read
        issue(s + NS - 1);
        if (s + 1 < NSL) split_a(s + 1);
        const char* As = aring + (s % NS) * A_ST;
        const char* Bs = bring + (s % NS) * B_ST;
        f32x4 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) { bh[j] = *reinterpret_cast<const f32x4*>(Bs + b_off[j]); bl[j] = *reinterpret_cast<const f32x4*>(Bs + (b_off[j] ^ 64)); }
#pragma unroll
        for (int i = 0; i < MI; ++i) { ah[i] = *reinterpret_cast<const f32x4*>(As + a_off[i]); al[i] = *reinterpret_cast<const f32x4*>(As + (a_off[i] ^ 64)); }
        halo_mma_split<MI, NI>(acc, bh, al);
        halo_mma_split<MI, NI>(acc, bl, ah);
        halo_mma_split<MI, NI>(acc, bh, ah);
    }
    wait_vmcnt<0>();
    __syncthreads();
    epilogue_tile<BM, BN, NW, MI, NI, float>(d, lds, m0, n0, wr * WM, wc * WN, lane, t, acc);
}

// ----------------------------------------------------------------------------
// The two ENDS of the U-Net (unet6.py:403-404, 505): 3 -> 128 channels in, 128 -> 3 out, with the 3 padded to 8 -- one 16-byte chunk
// per pixel.  Through the general kernels these ran ~20 us each for 0.6 GFLOP (k-slabs that are not 64-aligned fall back to the
// register-staged kernel: two 64-deep slabs of per-element gather arithmetic and an 8-byte scattered epilogue).  They are memory-bound
// streaming problems, and an MFMA fragment of 8 consecutive k IS one tap's 8 channels of one pixel:
//   conv_thin_k (8 input channels: the first convolution, and the data gradient of the last through the transposed shadow): the reduction
//     is 9 taps x 8 = 72 -> three k-steps of 4 taps (3 of 12 zero); a lane loads its tap's pixel chunk STRAIGHT from global memory into the
//     A fragment (no LDS), the filter fragments (3 x N/16, from an 18-KB array) stay in registers while the wave walks its 16-pixel tiles;
//   (The mirror image for the LAST convolution -- 8 output channels, 1152-deep reduction, pixel fragments straight from global memory,
//    filter fragments in LDS -- was written and measured at 18.8 us against the general kernel's 16.6: without a halo in LDS the 3x3
//    re-reads of a 128-channel pixel go to L2 nine times.  Not kept.)
// Same operand order as everywhere: acc holds D[m = lane & 15][n = 4 (lane >> 4) + reg].
// ----------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void conv_thin_k_kernel(mdm_gemm_desc d) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, r = lane & 15;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(d.B);
    const bf16_t* S = reinterpret_cast<const bf16_t*>(d.src0);
    bf16x8 bf[3][NJ];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
        const int tap = 4 * s2 + g;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = j * 16 + r;
            bf[s2][j] = (tap < 9 && n < d.N) ? *reinterpret_cast<const bf16x8*>(B + (int64_t)tap * d.wtap + (int64_t)n * d.ldb) : zero8;
        }
    }
    const int ntile = (d.M + 15) >> 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntile; tile += gridDim.x * 4) {
        const int m = tile * 16 + r;
        const RowPix rp = decode_row(d, m < d.M ? m : 0);
        bf16x8 af[3];
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            const int tap = 4 * s2 + g, ty = tap / 3, tx = tap - ty * 3;
            const int spix = (tap < 9 && m < d.M) ? gather_pix(d, rp, ty, tx) : -1;
            af[s2] = spix >= 0 ? *reinterpret_cast<const bf16x8*>(S + (int64_t)spix * d.ld0) : zero8;
        }
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s2][j], af[s2], acc[j], 0, 0, 0);
        if (m < d.M) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = j * 16 + 4 * g;
                if (n >= d.N) continue;
                float4 v = make_float4(acc[j][0] * d.alpha, acc[j][1] * d.alpha, acc[j][2] * d.alpha, acc[j][3] * d.alpha);
                if (d.bias) { const float4 b = *reinterpret_cast<const float4*>(d.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                bf16_t* q = reinterpret_cast<bf16_t*>(d.D0) + (int64_t)m * d.ldd0 + n;
                if (d.acc0) { const float4 o = load4(q); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                store4(q, v);
            }
        }
    }
}

// TWO independent convolutions in one launch (mdm_gemm_pair): workgroups [0, na) run the 3x3 halo convolution `a`, the rest
// the 1x1 convolution `b` (conv_lin2).  The pairs are the ResidualBlock's conv1 next to its skip projection in the forward
// (both only need the block input / its norm) and conv2's data gradient next to the skip projection's in the backward (both
// read the block's dY): the 1x1 launch -- 7-15 us, mostly launch, prologue and drain -- disappears into the tail of its
// neighbour, whose workgroups are dispatched first.  Both bodies are 512-thread workgroups on dynamic LDS.
template <int BM, int NPW, int BN, int NSB, int LBM, int LBN, int LNS, int LWR, int LWC>
__global__ __launch_bounds__(512) void conv_pair_kernel(mdm_gemm_desc a, mdm_gemm_desc b, int na) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    static_assert(LWR * LWC == 8, "conv_pair: both roles are 8-wave workgroups");
    if ((int)blockIdx.x < na) conv_halo_body<BM, NPW, BN, NSB, 3>(a, lds, (int)blockIdx.x, na);
    else conv_lin2_body<LBM, LBN, LNS, LWR, LWC, 1, true, false>(b, lds, (int)blockIdx.x - na, (int)gridDim.x - na, 0);
}


// ----------------------------------------------------------------------------
// CHAIN: consecutive convolutions of the U-Net's small-map trunk (4x4 / 8x8: unet6.py:336-362, 296-333, 478-506) as ONE persistent
// launch (mdm_chain_*).  At 32 images per GPU each of those layers is a 9-16 us launch for 1-4 us of loop (DESIGN findings 23, 26): the
// fixed parts of a launch -- boundary, prologue, drain -- are what the trunk's time is made of.  Here a grid of resident workgroups (one
// per CU) walks the layers ("phases") in order; a phase is one mdm_gemm or one mdm_gemm_pair, its virtual blocks are dealt out with
// stride gridDim.x and run the SAME device bodies (conv_halo_body / conv_lin2_body) as the per-layer launches -- the results are
// bit-identical to them.
//   Dependencies.  Every tile of these phases is 64 output pixels = whole images (one 8x8 image or four 4x4 images) and a
//   convolution reads only the pixels of its own images, so a block of phase p needs exactly the blocks of phase p-1 that cover ITS
//   images -- not a grid-wide barrier.  cnt[p][image] counts the finished blocks of phase p that cover that image; a block of phase p
//   waits until cnt[p-1][img] == target[p-1] for each of its images.  Because the wait is on ALL blocks of p-1 for those images and
//   each of them had waited for all of p-2's, everything any earlier phase wrote for these images is complete too (residuals, skip
//   tensors, GroupNorm statistics), and a later phase cannot overwrite what an earlier one still reads.
//   Visibility (cdna_hip_programming.md Guideline 16): producer = every wave drains its stores (vmcnt(0)), workgroup barrier, one
//   lane releases at agent scope, then the counter adds; consumer = one wave polls (relaxed, agent scope), ONE agent-scope acquire,
//   vmcnt(0), workgroup barrier, then plain / LDS-DMA loads.  Nothing depends on which XCD a workgroup lands on.
//   Progress.  Every workgroup walks (phase, block) in the same global order and waits only on earlier phases; with all gridDim.x
//   workgroups resident (grid <= CUs: one workgroup per CU by its LDS request) the earliest unfinished block is never blocked.  The
//   spin is bounded all the same: on a timeout the error word is set and the launch ends (with wrong results) instead of hanging.
// ----------------------------------------------------------------------------
#ifndef MDM_CHAIN_RELEASE_FENCE
#define MDM_CHAIN_RELEASE_FENCE 1           // 1: an agent-scope release (L2 write-back) per block before its arrivals
#endif
#ifndef MDM_CHAIN_WT
#define MDM_CHAIN_WT 0                      // 1: the tile stores of a chain's phases write through (sc1); without the release an
                                            //    intermittent mismatch was observed (profiles/r04_chain_findings.md): not the default
#endif
struct ChainPhase {
    int kind[2];            // role 0 / role 1 (mdm_gemm_pair): 0 / 1 = halo 64x32 (2 / 3 pieces per wave), 2 = lin2 64x64, 3 / 4 = conv_small 64x32,
                            // 5 = conv_small 32x16, -1 = none
    int desc[2];
    int nblk[2];
    int tiles_n[2];         // channel tiles per 64-pixel row tile
    int img_sh_role[2];     // log2(images per row tile) of each role: 0 on 8x8 maps, 2 (64-pixel tiles) or 1 (32-pixel tiles) on 4x4
    unsigned target;        // finished blocks per image that complete this phase
    unsigned prev_target;   // ... the phase in front of it (0: nothing to wait for)
};
#ifdef MDM_STAMP
#define MDM_CHAIN_STAMP_BASE 16384          // records [16384, 32768) of g_stamp_buf: one per (phase, virtual block)
#endif

__global__ __launch_bounds__(512) void chain_zero_kernel(unsigned* cnt, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) cnt[i] = 0u;
}

__global__ __launch_bounds__(512) void chain_kernel(const mdm_gemm_desc* __restrict__ descs, const ChainPhase* __restrict__ phases,
                                                    const int n_ph, unsigned* cnt, const int n_img, unsigned* err) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x;
    for (int p = 0; p < n_ph; ++p) {
        const ChainPhase P = phases[p];
        const int total = P.nblk[0] + P.nblk[1];
        unsigned* const mine = cnt + (int64_t)p * n_img;
        const unsigned* const prev = cnt + (int64_t)(p - 1) * n_img;
        for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
            const int role = vb >= P.nblk[0] ? 1 : 0;
            const int bx = role ? vb - P.nblk[0] : vb, gx = P.nblk[role];
            const int rt = udiv_small(xcd_remap(bx, gx), P.tiles_n[role]);     // the row tile the body will compute (same arithmetic)
            const int nimg = 1 << P.img_sh_role[role], img0 = rt << P.img_sh_role[role];
            MDM_T(const unsigned long long t_w0 = stamp_now();)
            if (P.prev_target != 0u) {
                if (t < 64) {                                   // wave 0: lane i polls the counter of image img0 + i
                    bool ok = true;
                    if (t < nimg) {
                        const unsigned* c = prev + img0 + t;
                        unsigned spins = 0;
                        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P.prev_target) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1u << 20)) { ok = false; break; }
                        }
                    }
                    if (!ok) __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
            }
            MDM_T(const unsigned long long t_w1 = stamp_now();)
            {
                const mdm_gemm_desc d = descs[P.desc[role]];
                const int kind = P.kind[role];
                constexpr bool WT = MDM_CHAIN_WT != 0;      // tile stores write through (store8_pub)
                if (kind == 0) conv_halo_body<64, 2, 32, 3, 3, bf16_t, false, WT>(d, lds, bx, gx);
                else if (kind == 1) conv_halo_body<64, 3, 32, 3, 3, bf16_t, false, WT>(d, lds, bx, gx);
                else if (kind == 3) conv_small_body<4, WT>(d, lds, bx, gx);
                else if (kind == 4) conv_small_body<5, WT>(d, lds, bx, gx);
                else if (kind == 5) conv_small_body<3, WT, 32, 16>(d, lds, bx, gx);
                else conv_lin2_body<64, 64, 4, 4, 2, 1, true, false, WT>(d, lds, bx, gx, 0);
            }
            MDM_T(const unsigned long long t_b1 = stamp_now();)
            // publish: every wave's stores have been acknowledged, then one release for the workgroup, then the arrivals
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                    // (also: the next block may refill the LDS)
            if (t < 64) {
#if MDM_CHAIN_RELEASE_FENCE
                if (t == 0) {                                   // (plain tile stores + L2 write-back: measured +9 us per hand-off)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#endif
                if (t < nimg) __hip_atomic_fetch_add(mine + img0 + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef MDM_STAMP
            if (t == 0) {
                const unsigned rec = MDM_CHAIN_STAMP_BASE + (unsigned)p * 512u + (unsigned)vb;
                if (vb < 512 && rec < MDM_STAMP_RECS) {
                    unsigned long long* r = g_stamp_buf + (size_t)rec * 32;
                    r[0] = t_w0; r[1] = t_w1; r[2] = t_b1; r[3] = stamp_now(); r[4] = stamp_hw_id(); r[5] = (unsigned long long)blockIdx.x; r[6] = 1;
                }
            }
#endif
        }
    }
}

// ----------------------------------------------------------------------------
// host launch
// ----------------------------------------------------------------------------
static int validate(const mdm_gemm_desc& d) {
    MDM_REQUIRE(d.dtype == MDM_F32 || d.dtype == MDM_BF16, "gemm: bad dtype %d", d.dtype);
    MDM_REQUIRE(d.layout >= 0 && d.layout <= 2, "gemm: bad layout %d", d.layout);
    MDM_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch >= 1, "gemm: bad dims M=%d N=%d K=%d batch=%d", d.M, d.N, d.K, d.batch);
    const int vec = d.dtype == MDM_BF16 ? 8 : 4;
    MDM_REQUIRE(d.N % vec == 0, "gemm: N=%d must be a multiple of %d", d.N, vec);
    MDM_REQUIRE(d.N0 % 4 == 0 && d.N0 > 0 && d.N0 <= d.N, "gemm: bad N0=%d (N=%d)", d.N0, d.N);
    MDM_REQUIRE(d.D0 != nullptr && (d.N0 == d.N || d.D1 != nullptr), "gemm: missing destination");
    MDM_REQUIRE(d.ldd0 % 4 == 0 && (d.D1 == nullptr || d.ldd1 % 4 == 0), "gemm: destination pitch must be a multiple of 4");
    if (d.layout != 2) MDM_REQUIRE(d.K % vec == 0, "gemm: K=%d must be a multiple of %d in layouts 0/1", d.K, vec);
    if (d.layout == 2) MDM_REQUIRE(d.M % vec == 0, "gemm: M=%d must be a multiple of %d in layout 2", d.M, vec);
    if (d.conv) {
        MDM_REQUIRE(d.batch == 1, "gemm: conv gather does not take a batch");
        MDM_REQUIRE(d.src0 != nullptr && d.C0 > 0 && d.C0 % vec == 0 && d.C1 % vec == 0 && (d.C1 == 0 || d.src1 != nullptr),
                    "gemm: bad sources C0=%d C1=%d", d.C0, d.C1);
        MDM_REQUIRE(d.ld0 % vec == 0 && (d.C1 == 0 || d.ld1 % vec == 0), "gemm: source pitch must be a multiple of %d", vec);
        MDM_REQUIRE(d.stride == 1 || d.stride == 2, "gemm: stride %d", d.stride);
        MDM_REQUIRE(d.ups == 0 || d.ups == 1, "gemm: ups %d", d.ups);
        MDM_REQUIRE(d.KH > 0 && d.KW > 0 && d.OH > 0 && d.OW > 0 && d.IH > 0 && d.IW > 0, "gemm: bad conv geometry");
        MDM_REQUIRE(!d.ups || (d.IH % 2 == 0 && d.IW % 2 == 0), "gemm: upsampled extent must be even");
        if (d.layout != 2) {
            MDM_REQUIRE(d.Ck > 0 && d.Ck % vec == 0 && d.K == d.KH * d.KW * d.Ck, "gemm: K=%d != taps*Ck (Ck=%d)", d.K, d.Ck);
            MDM_REQUIRE(d.B != nullptr, "gemm: missing weights");
        } else {
            MDM_REQUIRE(d.A != nullptr && d.N == d.C0 + d.C1, "gemm: wgrad N=%d must equal C0+C1", d.N);
        }
    } else {
        MDM_REQUIRE(d.A != nullptr && d.B != nullptr, "gemm: missing operand");
        MDM_REQUIRE(d.lda % vec == 0 && d.ldb % vec == 0, "gemm: operand pitch must be a multiple of %d", vec);
    }
    if (d.rowvec) MDM_REQUIRE(d.rows_per_img > 0 && d.rv_ld % 4 == 0, "gemm: bad rowvec params");
    if (d.dbias) MDM_REQUIRE(d.layout == 2 && (d.dtype == MDM_BF16 || (!d.conv && d.batch == 1)),
                             "gemm: dbias is fused into the bf16 layout-2 kernels and the fp32 layout-2 kernel without a gather");
    if (d.resid) MDM_REQUIRE(d.ldr % 4 == 0, "gemm: bad resid pitch");
    if (d.splitk > 1) MDM_REQUIRE(d.out_f32 || d.dtype == MDM_F32, "gemm: split-K needs an fp32 destination");
    if (d.splitk > 1 && d.layout != 2 && !d.conv)
        MDM_REQUIRE(!d.bias && !d.rowvec && !d.resid && d.alpha == 1.0f,
                    "gemm: split-K of a plain contraction in layouts 0/1 takes no epilogue");
    return 0;
}

// ---- one dispatch rule, no run-time switches (measured alternatives are named in the comments of the kernels)
constexpr int kBigMinTiles = 200;      // a tile shape is used when it still yields about one workgroup per CU
constexpr int kWgradBlocks = 256;      // per-layer weight-gradient launch (mdm_gemm): aim at this many workgroups
constexpr int kWgradMinSlabs = 4;      // ... of at least this many 64-deep k-slabs each

template <int BM, int BN, int LAYOUT, int NSTAGE, bool CONV>
static int launch_ring_one(const mdm_gemm_desc& d, dim3 grid, hipStream_t s) {
    constexpr int bytes = NSTAGE * (BM + BN) * 64 * 2;
    static bool configured = false;
    if (!configured) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<BM, BN, LAYOUT, NSTAGE, CONV, 8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, LAYOUT, NSTAGE, CONV, 8>), grid, dim3(512), bytes, s, d);
    return 0;
}
template <int BM, int BN, int NSTAGE>
static int launch_ring(const mdm_gemm_desc& d, dim3 grid, hipStream_t s) {
    const bool c = d.conv != 0;
    switch (d.layout) {
        case 0: return c ? launch_ring_one<BM, BN, 0, NSTAGE, true>(d, grid, s) : launch_ring_one<BM, BN, 0, NSTAGE, false>(d, grid, s);
        case 1: return c ? launch_ring_one<BM, BN, 1, NSTAGE, true>(d, grid, s) : launch_ring_one<BM, BN, 1, NSTAGE, false>(d, grid, s);
        default: return c ? launch_ring_one<BM, BN, 2, NSTAGE, true>(d, grid, s) : launch_ring_one<BM, BN, 2, NSTAGE, false>(d, grid, s);
    }
}

template <int BM, int BN, int NSTAGE, int WR, int WC>
static int launch_lin2(const mdm_gemm_desc& d, dim3 grid, hipStream_t s) {         // the software-pipelined variant
    int bytes = NSTAGE * (BM + BN) * 64 * 2;
    if (d.gnb_x && bytes < 16384 + 65536 + 4096 + 512) bytes = 16384 + 65536 + 4096 + 512;      // fused GroupNorm backward (64 x 64 tiles)
    static int configured = 0;
    if (configured < bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lin2_kernel<BM, BN, NSTAGE, WR, WC, 1, true, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = bytes;
    }
    hipLaunchKernelGGL((conv_lin2_kernel<BM, BN, NSTAGE, WR, WC, 1, true, false>), grid, dim3(64 * WR * WC), bytes, s, d);
    return 0;
}

template <int BM, int BN, int NSTAGE, int NW>
static int launch_wgrad_lin(const mdm_gemm_desc& d, int tiles_x, int items, hipStream_t s) {
    constexpr int bytes = NSTAGE * (BM + BN) * 64 * 2;
    static bool configured = false;
    if (!configured) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lin_kernel<BM, BN, NSTAGE, NW>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = true;
    }
    hipLaunchKernelGGL((wgrad_lin_kernel<BM, BN, NSTAGE, NW>), dim3((unsigned)items), dim3(64 * NW), bytes, s, d, tiles_x);
    return 0;
}
static bool wgrad_lin_eligible(const mdm_gemm_desc& d) {
    if (!(d.dtype == MDM_BF16 && d.layout == 2 && d.conv && d.OW > 0 && 64 % d.OW == 0 && (d.OH & (d.OH - 1)) == 0 &&
          d.K % 64 == 0 && d.C0 % 8 == 0 && d.C1 % 8 == 0 && d.IW == d.OW * d.stride && d.IH == d.OH * d.stride))
        return false;
    if (d.ups == 0) return d.stride == 1 || d.stride == 2;                          // "same" conv / SamePad2d + stride 2
    return d.stride == 1 && d.ups == 1 && ((64 / d.OW) & 1) == 0;                   // folded nearest x2 upsample
}

// all nine taps in one pass (wgrad_taps_body); MDM_WGRAD_TAPS=0 keeps every layer on the per-tap kernel (A/B runs)
static bool wgrad_taps_eligible(const mdm_gemm_desc& d) {
    static const bool off = [] { const char* e = getenv("MDM_WGRAD_TAPS"); return e && atoi(e) == 0; }();
    if (off) return false;
    return d.dtype == MDM_BF16 && d.layout == 2 && d.conv && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad_t == 1 && d.pad_l == 1 &&
           (d.OW == 8 || d.OW == 16 || d.OW == 32 || d.OW == 64) && d.OH >= 2 && (d.OH & (d.OH - 1)) == 0 && d.OH * d.OW >= 64 && d.IH == d.OH && d.IW == d.OW &&
           (d.ups == 0 || d.ups == 1) && (d.M % TAPS_BM == 0 || (d.M < TAPS_BM && d.M % 8 == 0)) && (d.N % TAPS_BN == 0 || (d.N < TAPS_BN && d.N % 8 == 0)) &&
           cdiv(d.N, TAPS_BN) * cdiv(d.M, TAPS_BM) <= 4095 && d.K % 64 == 0 &&
           d.K / 64 < 65535 && d.C0 % 8 == 0 && d.C1 % 8 == 0 && d.N == d.C0 + d.C1 && d.out_f32 && !d.acc0 && d.ldd0 == d.N && d.N0 == d.N &&
           d.alpha == 1.0f && d.dtap == (int64_t)d.M * d.N;
}

static int halo_pieces(int bm, int OH, int OW) {           // 1-KiB pieces of one halo buffer
    const int imgs = bm > OH * OW ? bm / (OH * OW) : 1, R = imgs > 1 ? OH : bm / OW;
    return (imgs * (R + 2) * (OW + 2) + 7) / 8;
}
template <int BM, int NPW, int NSB, int BN = 64, typename T = bf16_t, bool SPLIT = false, int TG = 3>
static int launch_halo(const mdm_gemm_desc& d, hipStream_t s) {      // TG = 3: one filter row (3 taps) per barrier
    const int NPA = halo_pieces(BM, d.OH, d.OW);
    int bytes = 2 * NPA * 1024 + NSB * TG * BN * 128 + 1024;
    if (bytes < BM * BN * 4) bytes = BM * BN * 4;                 // the tile epilogue parks the fp32 tile there
    if (BM == 64 && d.gnb_x && bytes < 16384 + 65536 + 4096 + 512) bytes = 16384 + 65536 + 4096 + 512;   // fused GroupNorm backward
    MDM_REQUIRE(NPA <= 8 * NPW && bytes <= 160 * 1024, "conv_halo: tile does not fit (NPA=%d, %d bytes)", NPA, bytes);
    static int configured = 0;
    if (configured < bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<BM, NPW, BN, NSB, TG, T, SPLIT>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = bytes;
    }
    dim3 grid((unsigned)((int64_t)(d.M / BM) * cdiv(d.N, BN)));
    hipLaunchKernelGGL((conv_halo_kernel<BM, NPW, BN, NSB, TG, T, SPLIT>), grid, dim3(512), bytes, s, d);
    return 0;
}
static int small_pieces(const mdm_gemm_desc& d, int bm = 64) {           // 1-KiB pieces of one conv_small halo buffer (256-byte pixel rows)
    const int imgs = bm / (d.OH * d.OW);
    return (imgs * (d.OH + 2) * (d.OW + 2) + 3) / 4;
}
static int small_lds_bytes(const mdm_gemm_desc& d, int bm = 64, int bn = 32) {
    const int bytes = 2 * small_pieces(d, bm) * 1024 + 3 * 3 * bn * 256 + 1024;
    return bytes < 16384 + 8 * 1024 ? 16384 + 8 * 1024 : bytes;          // (the parked partials: 8 waves x <= 1 KiB behind the first 16 KiB)
}
template <int NPW, int BM = 64, int BN = 32>
static int launch_small(const mdm_gemm_desc& d, hipStream_t s) {
    const int bytes = small_lds_bytes(d, BM, BN);
    MDM_REQUIRE(small_pieces(d, BM) <= 8 * NPW && bytes <= 160 * 1024 && (BM != 64 || bytes >= 16384 + 65536 + 4096 + 512),
                "conv_small: tile does not fit (%d bytes)", bytes);
    static int configured = 0;
    if (configured < bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_kernel<NPW, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = bytes;
    }
    hipLaunchKernelGGL((conv_small_kernel<NPW, BM, BN>), dim3((unsigned)((int64_t)(d.M / BM) * cdiv(d.N, BN))), dim3(512), bytes, s, d);
    return 0;
}
// 0: not eligible, else the pixel tile (64, 128 or 256)
static int halo_tile(const mdm_gemm_desc& d) {
    if (!(d.dtype == MDM_BF16 && d.layout == 0 && d.conv && d.KH == 3 && d.KW == 3 && d.stride == 1 && (d.ups == 0 || d.ups == 1) &&
          !(d.ups && (d.transposed || d.C1)) && d.pad_t == 1 && d.pad_l == 1 && d.IH == d.OH && d.IW == d.OW &&
          d.C0 % 64 == 0 && d.C1 % 64 == 0 && d.Ck == d.C0 + d.C1 && d.N % 64 == 0 && d.N0 % 8 == 0 && !d.out_f32 &&
          (d.OH & (d.OH - 1)) == 0))          // the kernel shifts by log2(OW), log2(OH)
        return 0;
    if (d.OW == 16 || d.OW == 32 || d.OW == 64) {
        for (int bm : {256, 128}) {
            if (bm % d.OW || d.OH % (bm / d.OW) || d.M % bm) continue;
            if (halo_pieces(bm, d.OH, d.OW) > 48) continue;
            if ((int64_t)(d.M / bm) * (d.N / 64) >= kBigMinTiles) return bm;
        }
        return 0;
    }
    // small maps (4x4, 8x8): 64-pixel tiles of whole images.  Too few workgroups to fill the chip, but a workgroup's
    // time is set by the filter bytes it streams (64 channels x 9 C: the same for every tile size), one launch
    // replaces the tap-split conv + its epilogue launch, and the tile holds whole images (GroupNorm-fusable)
    // (4x4 maps with > 256 input channels: level with the tap-split conv + its epilogue launch since the tiles are 32 channels
    //  wide -- 3.977 vs 3.976 ms/step -- and six launches fewer)
    if ((d.OW == 4 || d.OW == 8) && d.OH == d.OW && 64 % (d.OH * d.OW) == 0 && d.M % 64 == 0) return 64;
    return 0;
}

// The exact-fp32 path on the halo kernel (conv_halo_body<..., float>): 0 = not eligible, else the pixel tile.  Same geometry
// rules as halo_tile; channel counts in 32-channel slabs.  The fp32 loop is MFMA-bound, so the tile is picked for the fewest
// idle slots in the last round of workgroups (one workgroup per CU): at sample_num = 100 a 32x32 layer is 800 tiles of 256 pixels
// (3.1 rounds: 78 % of the slots busy) or 1600 of 128 (6.25 rounds: 89 %).
static int halo_tile_f32(const mdm_gemm_desc& d) {
    if (!(d.dtype == MDM_F32 && d.layout == 0 && d.conv && d.KH == 3 && d.KW == 3 && d.stride == 1 && (d.ups == 0 || d.ups == 1) &&
          !(d.ups && (d.transposed || d.C1)) && d.pad_t == 1 && d.pad_l == 1 && d.IH == d.OH && d.IW == d.OW &&
          d.C0 % 32 == 0 && d.C1 % 32 == 0 && d.Ck == d.C0 + d.C1 && d.N % 64 == 0 && d.N0 % 8 == 0 && d.splitk <= 1 &&
          (d.OH & (d.OH - 1)) == 0 && (d.OW & (d.OW - 1)) == 0))
        return 0;
    if (d.OW == 16 || d.OW == 32 || d.OW == 64) {
        int best = 0;
        double best_eff = 0.0;
        for (int bm : {256, 128}) {
            if (bm % d.OW || d.OH % (bm / d.OW) || d.M % bm) continue;
            if (halo_pieces(bm, d.OH, d.OW) > 48) continue;
            const int64_t tiles = (int64_t)(d.M / bm) * (d.N / 64);
            const double eff = (double)tiles / (double)(((tiles + 255) / 256) * 256);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = bm; }
        }
        return best;
    }
    if ((d.OW == 4 || d.OW == 8) && d.OH == d.OW && 64 % (d.OH * d.OW) == 0 && d.M % 64 == 0) return 64;
    return 0;
}

// conv_halo_mixed_kernel: whole rounds of 256-pixel tiles, the remainder as 128-pixel tiles -- taken when the remainder is at most one
// round of small tiles (otherwise two short rounds cost more than the one long round they replace).  0 = plain launch.
#ifndef MDM_SPLIT_BN128
#define MDM_SPLIT_BN128 160         // 128-channel split tiles when the layer has at least this many of them (0: never)
#endif
#ifndef MDM_SPLIT_MIXED
#define MDM_SPLIT_MIXED 1
#endif
static int launch_halo_mixed(const mdm_gemm_desc& d, hipStream_t s) {
    const int tn = d.N / 64;
    const int64_t tiles = (int64_t)(d.M / 256) * tn;
    const int64_t big_m = (tiles / 256) * 256 / tn;                // M tiles of 256 pixels in whole rounds (all their channel tiles)
    const int64_t rest = (d.M / 256 - big_m) * 2 * tn;             // 128-pixel tiles behind them
    if (!MDM_SPLIT_MIXED || d.OW != 32 || d.OH % 8 || big_m < 1 || rest < 1 || rest > 256 || tiles % 256 == 0) return -2;
    const int npa = halo_pieces(256, d.OH, d.OW), npb = halo_pieces(128, d.OH, d.OW);
    if (npa > 48 || npb > 32) return -2;
    int bytes = std::max(2 * npa * 1024 + 2 * 3 * 64 * 128 + 1024, 2 * npb * 1024 + 3 * 3 * 64 * 128 + 1024);
    bytes = std::max(bytes, 256 * 64 * 4);
    MDM_REQUIRE(bytes <= 160 * 1024, "conv_halo_mixed: tile does not fit (%d bytes)", bytes);
    static int configured = 0;
    if (configured < bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_mixed_kernel<6, 2, 4, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = bytes;
    }
    hipLaunchKernelGGL((conv_halo_mixed_kernel<6, 2, 4, 3>), dim3((unsigned)(big_m * tn + rest)), dim3(512), bytes, s, d, (int)(big_m * tn),
                       (int)(big_m * 256));
    return 0;
}

// The LAST convolution of the net (128 -> 3, padded to 8 output channels; unet6.py:505) with split products: on the register-staged
// kernel its 64-wide tiles were 8 real channels each and it took 172 us of the reverse step -- more than a 128 -> 128 layer.  On the
// 256-pixel halo tiles with a 32-channel tile (one 16-wide MFMA column per wave, rows beyond N zero): 0 = not eligible, else 256.
static int halo_small_n_split(const mdm_gemm_desc& d) {
    if (!(d.dtype == MDM_F32 && d.B_split != nullptr && d.layout == 0 && d.conv && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.ups == 0 &&
          !d.transposed && d.pad_t == 1 && d.pad_l == 1 && d.IH == d.OH && d.IW == d.OW && d.C0 % 32 == 0 && d.C1 % 32 == 0 &&
          d.Ck == d.C0 + d.C1 && d.N % 8 == 0 && d.N <= 32 && d.N0 % 8 == 0 && d.splitk <= 1 && (d.OH & (d.OH - 1)) == 0 &&
          (d.OW == 16 || d.OW == 32) && d.OH % (256 / d.OW) == 0 && d.M % 256 == 0 && halo_pieces(256, d.OH, d.OW) <= 48 && d.M / 256 >= 128))
        return 0;
    return 256;
}
#ifndef MDM_LIN_SPLIT_NS64
#define MDM_LIN_SPLIT_NS64 4        // ring stages of lin_split_kernel's 64-channel tile (six, 144 KB, measured 22.5 against 21.7 us per launch)
#endif
// lin_split_kernel: 0 = not eligible, else the channel tile (128, or 64 when 128 would leave the chip short of workgroups)
static int lin_split_tile(const mdm_gemm_desc& d) {
    const bool k1 = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_t == 0 && d.pad_l == 0 && d.IH == d.OH && d.IW == d.OW;
    const bool s2 = d.KH == 3 && d.KW == 3 && d.stride == 2 && d.pad_t >= 0 && d.pad_l >= 0 && d.pad_t <= 1 && d.pad_l <= 1;
    if (!(d.dtype == MDM_F32 && d.B_split != nullptr && d.layout == 0 && d.conv && (k1 || s2) && d.ups == 0 && !d.transposed &&
          d.C0 % 32 == 0 && d.C1 % 32 == 0 && d.C0 > 0 && d.Ck == d.C0 + d.C1 && d.K == d.KH * d.KW * d.Ck && d.N % 64 == 0 && d.N0 % 8 == 0 &&
          d.splitk <= 1 && d.batch <= 1 && d.ldb % 4 == 0 && d.ld0 % 4 == 0 && d.ld1 % 4 == 0 && d.M >= 64 &&
          (int64_t)d.M * d.ld0 < (1ll << 40)))
        return 0;
    if (d.N % 128 == 0 && (int64_t)cdiv(d.M, 128) * (d.N / 128) >= kBigMinTiles) return 128;
    return 64;
}
template <int BN>
static int launch_lin_split(const mdm_gemm_desc& d, hipStream_t s) {
    constexpr int NS = BN == 64 ? MDM_LIN_SPLIT_NS64 : 4;
    constexpr int bytes = NS * (128 * 128 + BN * 128);
    static_assert(bytes <= 160 * 1024 && bytes >= 128 * BN * 4, "lin_split: ring / epilogue tile do not fit");
    static bool configured = false;
    if (!configured) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_split_kernel<BN, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = true;
    }
    hipLaunchKernelGGL((lin_split_kernel<BN, NS>), dim3((unsigned)((int64_t)cdiv(d.M, 128) * (d.N / BN))), dim3(512), bytes, s, d);
    return 0;
}
#ifndef MDM_NSB256
#define MDM_NSB256 2                // filter stages of the 256-pixel bf16 halo tiles
#endif
#ifndef MDM_SPLIT_MIN128
#define MDM_SPLIT_MIN128 160
#endif
#ifndef MDM_SPLIT_NSB256
#define MDM_SPLIT_NSB256 2          // filter stages of the 256-pixel split tiles
#endif
// The split-products variant is not MFMA-bound: per flop a 256-pixel tile reads 2/3 of the LDS fragment bytes and streams half the
// filter bytes of a 128-pixel one (stamps: 893 cycles per tap on 128 pixels; the 16x16 layers on 256-pixel tiles run the same flops in
// 2/3 of the time of the 32x32 layers on 128-pixel tiles).  Estimated time = rounds of workgroups x tile cost (256: 1.46 x 128).
// (Pixel tile of the 64-channel split tiles; layers with enough 128-channel tiles take those instead, see gemm_launch.)
static int halo_tile_f32_split(const mdm_gemm_desc& d, int exact_choice) {
    // 8x8 maps: two whole images per tile when that still gives every CU a tile (sample_num = 100, 256 channels: 400 tiles of 64
    // pixels = 1.56 rounds at 16 % of the matrix pipe inside the loop; 200 tiles of 128 pixels = one round)
    if (exact_choice == 64 && d.OW == 8 && d.OH == 8 && d.M % 128 == 0 && (int64_t)(d.M / 128) * (d.N / 64) >= MDM_SPLIT_MIN128) return 128;
    if (exact_choice == 64 || !(d.OW == 16 || d.OW == 32 || d.OW == 64)) return exact_choice;
    int best = exact_choice;
    double best_t = 1e30;
    for (int bm : {256, 128}) {
        if (bm % d.OW || d.OH % (bm / d.OW) || d.M % bm) continue;
        if (halo_pieces(bm, d.OH, d.OW) > 48) continue;
        const int64_t tiles = (int64_t)(d.M / bm) * (d.N / 64);
        const double tm = (double)((tiles + 255) / 256) * (bm == 256 ? 1.46 : 1.0);
        if (tm < best_t - 1e-9) { best_t = tm; best = bm; }
    }
    return best;
}

// conv_thin_k: 3x3 stride-1 "same" bf16 convolution with 8 (padded) input channels, layout 0 (forward, or the data gradient of an
// 8-output-channel convolution through the transposed shadow), one source, plain epilogue (scale, bias, accumulate)
static bool thin_conv(const mdm_gemm_desc& d) {
    return d.dtype == MDM_BF16 && d.layout == 0 && d.conv && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.ups == 0 && d.pad_t == 1 &&
           d.pad_l == 1 && d.IH == d.OH && d.IW == d.OW && d.C1 == 0 && d.N0 == d.N && !d.D1 && !d.out_f32 && !d.rowvec && !d.resid &&
           d.splitk <= 1 && !d.gnb_x && !d.gnf_out && d.N % 4 == 0 && d.ldd0 % 4 == 0 &&
           d.C0 == 8 && d.Ck == 8 && d.ldb == 8 && d.ld0 % 8 == 0 && d.N <= 128 && d.N >= 16;
}

static bool ring_eligible(const mdm_gemm_desc& d) {
    if (d.dtype != MDM_BF16) return false;
    if (d.layout == 2) return true;
    if (d.splitk > 1 && !d.conv) return false;
    if (d.conv) return d.Ck % 64 == 0 && d.C0 % 64 == 0 && d.C1 % 64 == 0;
    return d.K % 64 == 0;
}

template <int BM, int BN, int LAYOUT, bool SPLIT = false>
static int launch_f32_mfma_one(const mdm_gemm_desc& d, dim3 grid, hipStream_t s) {
    constexpr int bytes = 2 * (BM + BN) * 36 * 4;
    static bool configured = false;
    if (!configured) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_mfma_kernel<BM, BN, LAYOUT, SPLIT>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = true;
    }
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, LAYOUT, SPLIT>), grid, dim3(256), bytes, s, d);
    return 0;
}
template <int BM, int BN>
static int launch_f32_mfma(const mdm_gemm_desc& d, dim3 grid, hipStream_t s) {
    switch (d.layout) {
        // (split products pay on the 128 x 128 tiles only: 77.6 -> 72.5 us per launch; on the 64 x 64 tiles the converting stores lengthen
        //  a loop that is a load -> store -> barrier chain, 33 -> 51 us: measured, not taken)
        case 0: return (d.f32_split && BM == 128) ? launch_f32_mfma_one<BM, BN, 0, BM == 128>(d, grid, s) : launch_f32_mfma_one<BM, BN, 0>(d, grid, s);
        case 1: return launch_f32_mfma_one<BM, BN, 1>(d, grid, s);
        default: return launch_f32_mfma_one<BM, BN, 2>(d, grid, s);
    }
}

template <int BM, int BN>
static void launch_bf16(const mdm_gemm_desc& d, dim3 grid, hipStream_t s) {
    switch (d.layout) {
        case 0: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 0>), grid, dim3(256), 0, s, d); break;
        case 1: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 1>), grid, dim3(256), 0, s, d); break;
        default: hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, 2>), grid, dim3(256), 0, s, d); break;
    }
}

}  // namespace mdm
extern "C" int mdm_gemm_can_fuse_gn_bwd(const mdm_gemm_desc* desc_host, int G);
extern "C" int mdm_gemm_can_fuse_gn_fwd(const mdm_gemm_desc* desc_host, int G);
namespace mdm {

// What mdm_gemm decides before it launches: the tile family, the reduction split, where split-K partials go.
struct Resolved {
    mdm_gemm_desc d;
    bool big, tap_split;
    int zouter;
    int64_t tiles, slab;        // output tiles of the chosen shape; bytes of one dense fp32 copy of the output
};
static int resolve(const mdm_gemm_desc* dh, bool planning, Resolved& r) {
    MDM_REQUIRE(dh != nullptr, "gemm: null descriptor");
    mdm_gemm_desc& d = r.d;
    d = *dh;
    if (planning) { d.ws = reinterpret_cast<void*>(16); d.ws_bytes = (int64_t)1 << 60; }   // "unlimited": never dereferenced
    if (d.N0 == 0) d.N0 = d.N;
    if (int rc = validate(d)) return rc;
    r.zouter = d.batch;
    if (d.layout == 2 && d.conv) r.zouter = d.KH * d.KW;
    // tile choice: 128x128 when that still yields enough workgroups; weight gradients (layout 2: small
    // output, huge reduction, split-K supplies the parallelism) take the big tile whenever it fits.
    r.big = d.dtype == MDM_BF16 && d.N >= 128 && d.M >= 128 &&
            (d.layout == 2 ? d.K >= 2048 : (int64_t)cdiv(d.M, 128) * cdiv(d.N, 128) * r.zouter >= kBigMinTiles);
    const int BM = d.dtype == MDM_F32 ? 64 : (r.big ? 128 : 64), BN = BM;
    const int BK = d.dtype == MDM_F32 ? 16 : 64;
    r.tiles = (int64_t)cdiv(d.M, BM) * cdiv(d.N, BN);
    if (d.layout == 2) {
        if (d.splitk <= 0) {     // auto: aim at ~kWgradBlocks workgroups, at least kWgradMinSlabs k-steps each
            int64_t want = (r.big ? kWgradBlocks : 4 * kWgradBlocks) / (r.tiles * r.zouter);
            int64_t cap = d.K / (kWgradMinSlabs * BK);
            d.splitk = (int)(want < 1 ? 1 : (want > cap ? (cap < 1 ? 1 : cap) : want));
        }
        if (d.splitk > 1 && !(d.out_f32 || d.dtype == MDM_F32)) d.splitk = 1;
    } else if (d.conv) {
        // small-M forward / data gradient: too few tiles to fill the chip -> split the filter taps over grid z
        d.splitk = 1;
        const int taps = d.KH * d.KW;
        if (halo_tile(d) == 64) {
            /* single launch on the halo kernel: no tap split */
        } else if (d.ws && d.dtype == MDM_BF16 && ring_eligible(d) && taps >= 9 && taps % 3 == 0 && r.tiles <= 160) {
            int sk = r.tiles <= 48 ? taps : 3;
            if (taps % sk) sk = 3;
            if (d.ws_bytes >= (int64_t)sk * d.M * d.N * 4) d.splitk = sk;
        }
    } else if (d.splitk < 1) {
        d.splitk = 1;
    }
    // slab mode needs a dense fp32 [tap|batch][M][N] destination and room for every split
    r.tap_split = d.conv && d.layout != 2 && d.splitk > 1;
    const bool dense = (d.out_f32 || d.dtype == MDM_F32) && d.N0 == d.N && d.ldd0 == d.N &&
                       (d.layout == 2 && d.conv ? d.dtap == (int64_t)d.M * d.N : (r.zouter == 1 || d.sD == (int64_t)d.M * d.N));
    r.slab = (int64_t)r.zouter * d.M * d.N * 4;
    if (r.tap_split) {
        /* workspace size checked above */
    } else if (d.splitk > 1 && d.ws && dense && d.ws_bytes >= r.slab * 2) {
        if (d.ws_bytes < r.slab * d.splitk) d.splitk = (int)(d.ws_bytes / r.slab);
    } else {
        // no (usable) workspace: the reduction is not split.  (Round 1-2 fell back to float atomics on D here; summation in
        // arrival order is not reproducible, and nothing on the product path needs it: every caller passes a workspace.)
        d.ws = nullptr;
        d.splitk = 1;
    }
    return 0;
}

// Which kernel of the conv_halo / conv_lin2 family a bf16 forward / data-gradient convolution takes (0: none of them).
// Tile choice (measured per shape): the largest tile that still gives the chip ~one workgroup per CU -- whole-row halo tiles
// of 256 / 128 pixels on the 32x32 / 16x16 maps; 64-pixel whole-image tiles on 4x4 / 8x8 (the loop there is the filter stream
// of ONE workgroup and these maps give only 32-128 of them, so 32 output channels per workgroup: half the stream, twice the
// workgroups; a fused GroupNorm epilogue needs whole groups inside the tile, i.e. groups of <= 32 channels); for everything
// else conv_lin2 at 128x128 / 64x128 (the 16x16 maps: +20 % over 64x64) / 64x64.
// (128-pixel halo tiles run three filter stages -- 3.951 vs 3.960 ms/step with two; the 256-pixel tiles have room for two.)
// 1x1 stride-1 convolution on a 4x4 / 8x8 map whose 64 x 64 conv_lin2 tiles hold whole images and whole channel groups: may
// carry the fused GroupNorm epilogues of the 64-pixel halo tiles
static bool lin2_gn_tile(const mdm_gemm_desc& d) {
    return d.dtype == MDM_BF16 && d.layout == 0 && d.conv && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.ups == 0 &&
           d.IH == d.OH && d.IW == d.OW && d.OH == d.OW && (d.OW == 4 || d.OW == 8) && d.M % 64 == 0 && d.N % 64 == 0 &&
           d.C0 % 64 == 0 && d.C1 % 64 == 0 && d.Ck == d.C0 + d.C1 && d.N0 % 8 == 0 && !d.out_f32 && d.splitk <= 1;
}
enum ConvVar { CV_NONE = 0, CV_H256_4, CV_H256_6, CV_H128_3, CV_H128_4, CV_H128_6, CV_H64_2_32, CV_H64_3_32, CV_H64_2_64, CV_H64_3_64,
               CV_L128, CV_L64x128, CV_L64, CV_S64_4, CV_S64_5, CV_S32_3 };
#ifndef MDM_SMALL_32X16
#define MDM_SMALL_32X16 1           // 0: the 4x4 maps stay on conv_small's 64 x 32 tiles (A/B builds)
#endif
#ifndef MDM_SMALL_CONV
#define MDM_SMALL_CONV 1            // 0: the 4x4 / 8x8 maps stay on conv_halo_body's 64 x 32 tiles (A/B builds)
#endif

static ConvVar conv_variant(const mdm_gemm_desc& d, const Resolved& r, unsigned grid_z) {
    if (!(d.dtype == MDM_BF16 && ring_eligible(d) && d.layout == 0 && d.conv &&
          (d.stride == 1 || (d.stride == 2 && !d.transposed && d.ups == 0)) && d.C0 <= 4096 && d.C1 <= 4096 &&
          (d.ups == 0 || (d.splitk <= 1 && halo_tile(d) != 0)) && d.KH * d.KW <= 9 && (d.KH * d.KW) % d.splitk == 0))
        return CV_NONE;
    const int64_t t_mid = (int64_t)cdiv(d.M, 64) * cdiv(d.N, 128) * grid_z;
    const int hb = d.splitk <= 1 ? halo_tile(d) : 0;
    if (hb) {
        const int npw = (halo_pieces(hb, d.OH, d.OW) + 7) / 8;      // halo pieces per wave
        if (hb == 256) return npw <= 4 ? CV_H256_4 : CV_H256_6;
        if (hb == 128) return npw <= 3 ? CV_H128_3 : npw <= 4 ? CV_H128_4 : CV_H128_6;
        if (d.N % 32 == 0 && (!(d.gnb_x || d.gnf_out) || d.N / (d.gnb_x ? d.gnb_G : d.gnf_G) <= 32)) {
            // 128-channel superslabs with the reduction split over the waves (conv_small_body) where the channel counts allow
            const int spw = (small_pieces(d) + 7) / 8;
            if (MDM_SMALL_CONV && d.Ck % 128 == 0 && d.C0 % 128 == 0 && d.C1 % 128 == 0 && spw <= 5) {
                // 4x4 maps: 32 pixels (two whole images) x 16 channels (two GroupNorm groups of eight) -- four times the workgroups, each
                // streaming a quarter of the filter bytes; its epilogue is the register one only (C / G == 8 where a GroupNorm is fused)
                const int cpg = d.gnb_x ? d.N / d.gnb_G : (d.gnf_out ? d.N / d.gnf_G : 8);
                if (MDM_SMALL_32X16 && d.OH * d.OW == 16 && d.M % 32 == 0 && d.N % 16 == 0 && cpg == 8 && (d.N0 & 7) == 0 &&
                    (int64_t)(d.M / 64) * (d.N / 32) < kBigMinTiles && (small_pieces(d, 32) + 7) / 8 <= 3)
                    return CV_S32_3;
                return spw <= 4 ? CV_S64_4 : CV_S64_5;
            }
            return npw <= 2 ? CV_H64_2_32 : CV_H64_3_32;
        }
        return npw <= 2 ? CV_H64_2_64 : CV_H64_3_64;
    }
    if ((d.gnb_x || d.gnf_out) && lin2_gn_tile(d)) return CV_L64;      // the fused epilogues live on the 64 x 64 tile
    if (r.big) return CV_L128;
    // (the qkv projection of the 8x8 maps is 192 such tiles -- 384 tiles of 64 x 64 = 1.5 rounds today; taking them from 180 on was level:
    //  3.582 vs 3.586 ms/step)
    if (d.N >= 128 && t_mid >= kBigMinTiles) return CV_L64x128;
    return CV_L64;
}

static int check_fused_gn(const mdm_gemm_desc& d) {
    if (d.gnb_x)
        MDM_REQUIRE(mdm_gemm_can_fuse_gn_bwd(&d, d.gnb_G) == 1 && d.gnb_stats && d.gnb_gamma && d.gnb_beta && d.gnb_dgamma && d.gnb_dbeta,
                    "gemm: gnb_* epilogue on a descriptor that does not qualify (mdm_gemm_can_fuse_gn_bwd)");
    if (d.gnf_out)
        MDM_REQUIRE(mdm_gemm_can_fuse_gn_fwd(&d, d.gnf_G) == 1 && d.gnf_gamma && d.gnf_beta && d.gnf_stats,
                    "gemm: gnf_* epilogue on a descriptor that does not qualify (mdm_gemm_can_fuse_gn_fwd)");
    return 0;
}

static int gemm_launch(const mdm_gemm_desc* dh, hipStream_t s) {
    Resolved r;
    if (int rc = resolve(dh, false, r)) return rc;
    const mdm_gemm_desc& d = r.d;
    const bool big = r.big;
    if (int rc0 = check_fused_gn(d)) return rc0;
    MDM_REQUIRE(r.tiles < (1ll << 31), "gemm: grid too large");
    dim3 grid((unsigned)r.tiles, 1, (unsigned)(r.zouter * d.splitk));
    MDM_REQUIRE(grid.z <= 65535, "gemm: grid.z=%u too large", grid.z);
    int rc = 0;
    if (d.dtype == MDM_F32 && d.layout == 0 && !d.conv && d.M <= 32 && d.K % 64 == 0 && d.K <= 512 && d.splitk <= 1 &&
        d.batch == 1 && !d.rowvec && !d.resid && !d.acc0 && d.N0 == d.N && d.lda % 4 == 0 && d.ldb % 4 == 0) {
        static bool configured = false;
        const int bytes = 32 * (d.K > 16 * 17 ? d.K : 16 * 17) * 4;
        if (!configured) {
            MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_skinny_f32_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 512 * 4));
            configured = true;
        }
        hipLaunchKernelGGL(linear_skinny_f32_kernel, dim3((unsigned)cdiv(d.N, 16)), dim3(256), bytes, s, d);
    } else if (d.dtype == MDM_F32 && d.layout == 2 && !d.conv && d.batch == 1 && d.K <= 128 && d.splitk <= 1 && d.N0 == d.N && !d.D1 &&
               d.M % 4 == 0 && d.N % 4 == 0 && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.ldd0 % 4 == 0 && !d.bias && !d.rowvec && !d.resid) {
        // weight gradient of a linear layer over a batch-sized reduction (the time-embedding path)
        const int bytes = d.K * 128 * 4;
        static int configured = 0;
        if (configured < bytes) {
            MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tn_skinny_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            configured = bytes;
        }
        hipLaunchKernelGGL(tn_skinny_f32_kernel, dim3((unsigned)((int64_t)cdiv(d.M, 64) * cdiv(d.N, 64))), dim3(256), bytes, s, d);
    } else if (const int lsb = lin_split_tile(d)) {
        rc = lsb == 128 ? launch_lin_split<128>(d, s) : launch_lin_split<64>(d, s);
    } else if (halo_small_n_split(d)) {
        rc = (halo_pieces(256, d.OH, d.OW) + 7) / 8 <= 4 ? launch_halo<256, 4, 2, 32, float, true>(d, s) : launch_halo<256, 6, 2, 32, float, true>(d, s);
    } else if (const int hb32_exact = halo_tile_f32(d)) {
        // exact-fp32 3x3 convolutions on the halo kernel (forward, folded upsample, transposed shadow): MFMA-bound
        const int hb32 = hb32_exact;
        const int npw = (halo_pieces(hb32, d.OH, d.OW) + 7) / 8;
        if (d.B_split != nullptr && !d.transposed) {
            // fp32 storage, products as bf16 hi / lo pairs on the bf16 matrix pipe (conv_halo_body<..., SPLIT>)
            const int hb32 = halo_tile_f32_split(d, hb32_exact);
            const int npw = (halo_pieces(hb32, d.OH, d.OW) + 7) / 8;
            // 256 pixels x 128 channels (one tap per barrier, four filter stages): the halo is staged and split once for twice the
            // channels and a wave multiplies 4 x 4 fragments per tap -- 16 fragment reads for 48 MFMAs where the 64-channel tile reads
            // 12 for 24; half as many tiles, so a 16x16 layer at sample_num = 100 is ONE round of workgroups instead of 1.56 in two
            if (hb32 == 256 && MDM_SPLIT_BN128 && d.N % 128 == 0 && (d.OW == 16 || d.OW == 32) &&
                (int64_t)(d.M / 256) * (d.N / 128) >= MDM_SPLIT_BN128) {
                rc = npw <= 4 ? launch_halo<256, 4, 4, 128, float, true, 1>(d, s) : launch_halo<256, 6, 4, 128, float, true, 1>(d, s);
            } else
            if (hb32 == 256 && (rc = launch_halo_mixed(d, s)) != -2) { /* whole rounds of 256-pixel tiles + a short round of 128-pixel ones */ }
            else if (hb32 == 256) rc = npw <= 4 ? launch_halo<256, 4, MDM_SPLIT_NSB256, 64, float, true>(d, s) : launch_halo<256, 6, MDM_SPLIT_NSB256, 64, float, true>(d, s);
            else if (hb32 == 128) rc = npw <= 3 ? launch_halo<128, 3, 3, 64, float, true>(d, s) : npw <= 4 ? launch_halo<128, 4, 3, 64, float, true>(d, s)
                                                                                             : launch_halo<128, 6, 3, 64, float, true>(d, s);
            else if ((int64_t)(d.M / 64) * (d.N / 64) < kBigMinTiles && d.N % 32 == 0)
                rc = npw <= 2 ? launch_halo<64, 2, 3, 32, float, true>(d, s) : launch_halo<64, 3, 3, 32, float, true>(d, s);
            else rc = npw <= 2 ? launch_halo<64, 2, 3, 64, float, true>(d, s) : launch_halo<64, 3, 3, 64, float, true>(d, s);
        } else
        if (hb32 == 256) rc = npw <= 4 ? launch_halo<256, 4, 2, 64, float>(d, s) : launch_halo<256, 6, 2, 64, float>(d, s);
        else if (hb32 == 128) rc = npw <= 3 ? launch_halo<128, 3, 3, 64, float>(d, s) : npw <= 4 ? launch_halo<128, 4, 3, 64, float>(d, s)
                                                                                         : launch_halo<128, 6, 3, 64, float>(d, s);
        else if ((int64_t)(d.M / 64) * (d.N / 64) < kBigMinTiles && d.N % 32 == 0)       // 4x4 maps: 32-channel tiles, twice the workgroups
            rc = npw <= 2 ? launch_halo<64, 2, 3, 32, float>(d, s) : launch_halo<64, 3, 3, 32, float>(d, s);
        else rc = npw <= 2 ? launch_halo<64, 2, 3, 64, float>(d, s) : launch_halo<64, 3, 3, 64, float>(d, s);
    } else if (d.dtype == MDM_F32) {
        // exact-fp32 MFMA kernel; 128 x 128 tiles when that still gives about one workgroup per CU
        const bool big32 = d.M >= 128 && d.N >= 128 && (int64_t)cdiv(d.M, 128) * cdiv(d.N, 128) * grid.z >= kBigMinTiles;
        rc = big32 ? launch_f32_mfma<128, 128>(d, dim3((unsigned)((int64_t)cdiv(d.M, 128) * cdiv(d.N, 128)), 1, grid.z), s)
                   : launch_f32_mfma<64, 64>(d, grid, s);
    } else if (thin_conv(d)) {
        // the 8-channel ends of the U-Net: the first convolution and the data gradient of the last
        const unsigned nb = (unsigned)std::min(cdiv(cdiv(d.M, 16), 4), 1024);
        switch (cdiv(d.N, 16)) {
            case 1: hipLaunchKernelGGL((conv_thin_k_kernel<1>), dim3(nb), dim3(256), 0, s, d); break;
            case 2: hipLaunchKernelGGL((conv_thin_k_kernel<2>), dim3(nb), dim3(256), 0, s, d); break;
            case 3: case 4: hipLaunchKernelGGL((conv_thin_k_kernel<4>), dim3(nb), dim3(256), 0, s, d); break;
            default: hipLaunchKernelGGL((conv_thin_k_kernel<8>), dim3(nb), dim3(256), 0, s, d); break;
        }
    } else if (const ConvVar cv = conv_variant(d, r, grid.z)) {
        const dim3 g2((unsigned)((int64_t)cdiv(d.M, 64) * cdiv(d.N, 128)), 1, grid.z);
        switch (cv) {
            case CV_H256_4: rc = launch_halo<256, 4, MDM_NSB256>(d, s); break;
            case CV_H256_6: rc = launch_halo<256, 6, MDM_NSB256>(d, s); break;
            case CV_H128_3: rc = launch_halo<128, 3, 3>(d, s); break;
            case CV_H128_4: rc = launch_halo<128, 4, 3>(d, s); break;
            case CV_H128_6: rc = launch_halo<128, 6, 3>(d, s); break;
            case CV_H64_2_32: rc = launch_halo<64, 2, 3, 32>(d, s); break;
            case CV_H64_3_32: rc = launch_halo<64, 3, 3, 32>(d, s); break;
            case CV_H64_2_64: rc = launch_halo<64, 2, 3>(d, s); break;
            case CV_H64_3_64: rc = launch_halo<64, 3, 3>(d, s); break;
            case CV_S64_4: rc = launch_small<4>(d, s); break;
            case CV_S64_5: rc = launch_small<5>(d, s); break;
            case CV_S32_3: rc = launch_small<3, 32, 16>(d, s); break;
            case CV_L128: rc = launch_lin2<128, 128, 3, 4, 2>(d, grid, s); break;
            case CV_L64x128: rc = launch_lin2<64, 128, 3, 2, 4>(d, g2, s); break;
            default: rc = launch_lin2<64, 64, 4, 4, 2>(d, grid, s); break;
        }
    } else if (wgrad_lin_eligible(d)) {
        MDM_REQUIRE((int64_t)grid.x * grid.z < (1ll << 30), "gemm: grid too large");
        rc = big ? launch_wgrad_lin<128, 128, 3, 8>(d, (int)grid.x, (int)(grid.x * grid.z), s)
                 : launch_wgrad_lin<64, 64, 4, 8>(d, (int)grid.x, (int)(grid.x * grid.z), s);
    } else if (ring_eligible(d)) {
        rc = big ? launch_ring<128, 128, 3>(d, grid, s) : launch_ring<64, 64, 4>(d, grid, s);
    } else if (big) {
        launch_bf16<128, 128>(d, grid, s);
    } else {
        launch_bf16<64, 64>(d, grid, s);
    }
    if (rc) return rc;
    if (r.tap_split) {
        const int64_t total4 = (int64_t)d.M * d.N / 4;
        int64_t nb = (total4 + 255) / 256;
        hipLaunchKernelGGL((splitk_epilogue_kernel<bf16_t>), dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, s, d);
    } else if (d.splitk > 1 && d.ws) {
        const int64_t total4 = (int64_t)r.zouter * d.M * d.N / 4;
        int64_t nb = (total4 + 255) / 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, s,
                           reinterpret_cast<const float*>(d.ws), d.splitk, total4, reinterpret_cast<float*>(d.D0), d.acc0);
    }
    return launch_status("gemm launch");
}

// ---- grouped weight gradients ---------------------------------------------------------------------------------
struct WgradGroup {
    const mdm_gemm_desc* descs_dev = nullptr;
    const int4* items_dev = nullptr;
    int n_items = 0, max_blocks = 0;
    std::vector<ReduceTable> reduces;       // the split-K sums of the group's split layers: passed by value at launch
    std::vector<int> reduce_blocks;
    const int4* taps_items_dev = nullptr;   // the nine-tap layers: table[round][CU] of the persistent launch
    int n_taps_items = 0, taps_blocks = 0;
    float* slots_dev = nullptr;             // nine-tap tiles that were cut: partial sums [slot][9][128][64]
    const PartTile* parts_dev = nullptr;
    int n_part_tiles = 0;
};
struct GroupItem { int desc, item, tiles_x, big, cost; };
struct TapsTile { int desc, tile, slabs; };


// ---- chains of small-map convolutions in one persistent launch (chain_kernel) ------------------------------------
struct Chain {
    const mdm_gemm_desc* descs_dev = nullptr;
    const ChainPhase* phases_dev = nullptr;
    unsigned* cnt_dev = nullptr;            // [n_ph][n_img] arrival counters + 4 words (error, spare)
    int n_ph = 0, n_img = 0, grid = 0, lds_bytes = 0;
};
// 0 = halo 64x32 with 2 pieces per wave, 1 = with 3, 2 = lin2 64x64, 3 / 4 = conv_small with 4 / 5 pieces, -1 = not a chain link
static int chain_kind(const mdm_gemm_desc* dh, Resolved& r, int* lds_bytes) {
    if (resolve(dh, false, r)) return -1;
    const mdm_gemm_desc& d = r.d;
    if (check_fused_gn(d)) return -1;
    const unsigned z = (unsigned)(r.zouter * d.splitk);
    if (z != 1 || r.tap_split || d.splitk > 1 || !d.conv || d.layout != 0 || d.dtype != MDM_BF16) return -1;
    const int P = d.OH * d.OW;
    if (!(P == 16 || P == 64) || d.M % 64 != 0 || d.M % P != 0) return -1;
    const ConvVar cv = conv_variant(d, r, z);
    if (cv == CV_H64_2_32 || cv == CV_H64_3_32) {
        const int NPA = halo_pieces(64, d.OH, d.OW);
        int bytes = 2 * NPA * 1024 + 3 * 3 * 32 * 128 + 1024;                       // as launch_halo<64, NPW, 3, 32>
        if (bytes < 64 * 32 * 4) bytes = 64 * 32 * 4;
        if (d.gnb_x && bytes < 16384 + 65536 + 4096 + 512) bytes = 16384 + 65536 + 4096 + 512;
        if (d.gnf_out && bytes < 16384 + 16384 + 4096) bytes = 16384 + 16384 + 4096;
        if (NPA > 8 * (cv == CV_H64_2_32 ? 2 : 3)) return -1;
        *lds_bytes = bytes;
        return cv == CV_H64_2_32 ? 0 : 1;
    }
    if (cv == CV_S64_4 || cv == CV_S64_5) {
        *lds_bytes = small_lds_bytes(d);
        return cv == CV_S64_4 ? 3 : 4;
    }
    if (cv == CV_S32_3) {
        *lds_bytes = small_lds_bytes(d, 32, 16);
        return 5;
    }
    if (cv == CV_L64 && r.tiles < (1ll << 20)) {
        int bytes = 4 * (64 + 64) * 64 * 2;                                        // as launch_lin2<64, 64, 4, 4, 2>
        if (d.gnb_x && bytes < 16384 + 65536 + 4096 + 512) bytes = 16384 + 65536 + 4096 + 512;
        *lds_bytes = bytes;
        return 2;
    }
    return -1;
}

}  // namespace mdm
using namespace mdm;

extern "C" int mdm_gemm(const mdm_gemm_desc* desc_host, void* stream) {
    return gemm_launch(desc_host, pick_stream(stream));
}

// ---- two independent convolutions in one launch (conv_pair_kernel)
template <int BM, int NPW, int BN, int NSB, int LBM, int LBN, int LNS, int LWR, int LWC>
static int launch_pair(const mdm_gemm_desc& a, const mdm_gemm_desc& b, int nb, hipStream_t s) {
    const int NPA = halo_pieces(BM, a.OH, a.OW);
    int bytes = 2 * NPA * 1024 + NSB * 3 * BN * 128 + 1024;                    // as launch_halo
    if (bytes < BM * BN * 4) bytes = BM * BN * 4;
    if (BM == 64 && a.gnb_x && bytes < 16384 + 65536 + 4096 + 512) bytes = 16384 + 65536 + 4096 + 512;
    constexpr int lin_bytes = LNS * (LBM + LBN) * 64 * 2;                      // as launch_lin2
    if (bytes < lin_bytes) bytes = lin_bytes;
    MDM_REQUIRE(NPA <= 8 * NPW && bytes <= 160 * 1024, "conv_pair: tile does not fit (NPA=%d, %d bytes)", NPA, bytes);
    static int configured = 0;
    if (configured < bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<BM, NPW, BN, NSB, LBM, LBN, LNS, LWR, LWC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = bytes;
    }
    const int na = (int)((int64_t)(a.M / BM) * (a.N / BN));
    hipLaunchKernelGGL((conv_pair_kernel<BM, NPW, BN, NSB, LBM, LBN, LNS, LWR, LWC>), dim3((unsigned)(na + nb)), dim3(512), bytes, s, a, b, na);
    return 0;
}

template <int NPW, int BM = 64, int BN = 32>
static int launch_pair_small(const mdm_gemm_desc& a, const mdm_gemm_desc& b, int nb, hipStream_t s) {
    int bytes = small_lds_bytes(a, BM, BN);
    constexpr int lin_bytes = 4 * (64 + 64) * 64 * 2;
    if (bytes < lin_bytes) bytes = lin_bytes;
    MDM_REQUIRE(small_pieces(a, BM) <= 8 * NPW && bytes <= 160 * 1024, "conv_pair_small: tile does not fit (%d bytes)", bytes);
    static int configured = 0;
    if (configured < bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_small_kernel<NPW, BM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = bytes;
    }
    const int na = (int)((int64_t)(a.M / BM) * (a.N / BN));
    hipLaunchKernelGGL((conv_pair_small_kernel<NPW, BM, BN>), dim3((unsigned)(na + nb)), dim3(512), bytes, s, a, b, na);
    return 0;
}

extern "C" int mdm_gemm_pair(const mdm_gemm_desc* a_host, const mdm_gemm_desc* b_host, void* stream) {
    hipStream_t s = pick_stream(stream);
    Resolved ra, rb;
    if (int rc = resolve(a_host, false, ra)) return rc;
    if (int rc = resolve(b_host, false, rb)) return rc;
    const mdm_gemm_desc &a = ra.d, &b = rb.d;
    if (int rc = check_fused_gn(a)) return rc;
    if (int rc = check_fused_gn(b)) return rc;
    const unsigned za = (unsigned)(ra.zouter * a.splitk), zb = (unsigned)(rb.zouter * b.splitk);
    const ConvVar va = conv_variant(a, ra, za), vb = conv_variant(b, rb, zb);
    // `b` must be a conv_lin2 launch of its own with no second stage (no split reduction), `a` a halo launch; the four pairs
    // below are the ones a unet6 step produces (4x4, 8x8, 16x16, 32x32 maps).  Anything else: two launches, same results.
    const bool b_plain = zb == 1 && !rb.tap_split && !(b.splitk > 1) && rb.tiles < (1ll << 20) && !b.gnb_x && !b.gnf_out;
    int rc = -2;
    if (b_plain && za == 1) {
        const int nb64 = (int)rb.tiles, nb64x128 = (int)((int64_t)cdiv(b.M, 64) * cdiv(b.N, 128));
        if (va == CV_S64_4 && vb == CV_L64) rc = launch_pair_small<4>(a, b, nb64, s);
        else if (va == CV_S64_5 && vb == CV_L64) rc = launch_pair_small<5>(a, b, nb64, s);
        else if (va == CV_S32_3 && vb == CV_L64) rc = launch_pair_small<3, 32, 16>(a, b, nb64, s);
        else if (va == CV_H64_2_32 && vb == CV_L64) rc = launch_pair<64, 2, 32, 3, 64, 64, 4, 4, 2>(a, b, nb64, s);
        else if (va == CV_H64_3_32 && vb == CV_L64) rc = launch_pair<64, 3, 32, 3, 64, 64, 4, 4, 2>(a, b, nb64, s);
        else if (va == CV_H128_3 && vb == CV_L64x128) rc = launch_pair<128, 3, 64, 3, 64, 128, 3, 2, 4>(a, b, nb64x128, s);
        else if (va == CV_H128_3 && vb == CV_L64) rc = launch_pair<128, 3, 64, 3, 64, 64, 4, 4, 2>(a, b, nb64, s);
        else if (va == CV_H256_6 && vb == CV_L128) rc = launch_pair<256, 6, 64, MDM_NSB256, 128, 128, 3, 4, 2>(a, b, nb64, s);
    }
    if (rc == -2) {
        rc = gemm_launch(a_host, s);
        if (rc == 0) rc = gemm_launch(b_host, s);
        return rc;
    }
    if (rc) return rc;
    return launch_status("gemm pair launch");
}

// ---- mdm_chain_*: see chain_kernel ----------------------------------------------------------------------------------
extern "C" int mdm_chain_accepts(const mdm_gemm_desc* a_host, const mdm_gemm_desc* b_host) {
    if (!a_host) return 0;
    Resolved ra, rb;
    int la = 0, lb = 0;
    const int ka = chain_kind(a_host, ra, &la);
    if (ka < 0) return 0;
    if (!b_host) return 1;
    const int kb = chain_kind(b_host, rb, &lb);
    // a pair = a halo convolution + a 1x1 projection without a fused GroupNorm epilogue (what mdm_gemm_pair fuses), same batch and map
    return (ka != 2 && kb == 2 && !rb.d.gnb_x && !rb.d.gnf_out && ra.d.M == rb.d.M && ra.d.OH == rb.d.OH && ra.d.OW == rb.d.OW) ? 1 : 0;
}
extern "C" int mdm_chain_create(const mdm_gemm_desc* descs_host, const int* roles, int n_phases, void* dev_buf, int64_t dev_bytes,
                                int64_t* need_bytes_out, void** handle_out) {
    MDM_REQUIRE(descs_host && roles && n_phases > 0 && need_bytes_out && handle_out, "chain_create: bad arguments");
    *handle_out = nullptr;
    static const int n_cu = [] {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) return 256;
        return pr.multiProcessorCount;
    }();
    std::vector<mdm_gemm_desc> ds;
    std::vector<ChainPhase> ph((size_t)n_phases);
    int n_img = 0, lds_max = 0, max_blocks = 0, di = 0;
    unsigned prev_target = 0;
    for (int p = 0; p < n_phases; ++p) {
        MDM_REQUIRE(roles[p] == 1 || roles[p] == 2, "chain_create: phase %d has %d descriptors (1 or 2)", p, roles[p]);
        ChainPhase& P = ph[(size_t)p];
        P.kind[1] = -1; P.desc[1] = 0; P.nblk[1] = 0; P.tiles_n[1] = 1; P.img_sh_role[0] = P.img_sh_role[1] = 0;
        MDM_REQUIRE(mdm_chain_accepts(descs_host + di, roles[p] == 2 ? descs_host + di + 1 : nullptr) == 1,
                    "chain_create: phase %d is not a chain link (mdm_chain_accepts)", p);
        unsigned target = 0;
        for (int q = 0; q < roles[p]; ++q) {
            Resolved r;
            int lb = 0;
            const int k = chain_kind(descs_host + di + q, r, &lb);
            const mdm_gemm_desc& d = r.d;
            const int Pix = d.OH * d.OW, imgs = d.M / Pix, bn = k == 2 ? 64 : (k == 5 ? 16 : 32), bm = k == 5 ? 32 : 64;
            MDM_REQUIRE(n_img == 0 || n_img == imgs, "chain_create: phase %d works on %d images, the chain on %d", p, imgs, n_img);
            n_img = imgs;
            P.kind[q] = k; P.desc[q] = (int)ds.size(); P.tiles_n[q] = cdiv(d.N, bn); P.nblk[q] = (d.M / bm) * P.tiles_n[q];
            P.img_sh_role[q] = Pix == 64 ? 0 : (bm == 32 ? 1 : 2);
            target += (unsigned)P.tiles_n[q];
            lds_max = lb > lds_max ? lb : lds_max;
            ds.push_back(d);
        }
        P.target = target; P.prev_target = prev_target;
        prev_target = target;
        max_blocks = std::max(max_blocks, P.nblk[0] + P.nblk[1]);
        di += roles[p];
    }
    MDM_REQUIRE(lds_max <= 160 * 1024, "chain_create: %d bytes of LDS", lds_max);
    auto pad256 = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t desc_bytes = pad256((int64_t)ds.size() * (int64_t)sizeof(mdm_gemm_desc)), ph_bytes = pad256((int64_t)n_phases * (int64_t)sizeof(ChainPhase));
    const int64_t cnt_bytes = pad256(((int64_t)n_phases * n_img + 4) * 4);
    *need_bytes_out = desc_bytes + ph_bytes + cnt_bytes;
    if (!dev_buf || dev_bytes < *need_bytes_out) return 0;          // size query
    char* base = reinterpret_cast<char*>(dev_buf);
    hipError_t e = hipMemcpy(base, ds.data(), ds.size() * sizeof(mdm_gemm_desc), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + desc_bytes, ph.data(), ph.size() * sizeof(ChainPhase), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(base + desc_bytes + ph_bytes, 0, (size_t)cnt_bytes);
    if (e != hipSuccess) return hip_fail(e, "chain_create: hipMemcpy");
    Chain* c = new Chain();
    c->descs_dev = reinterpret_cast<const mdm_gemm_desc*>(base);
    c->phases_dev = reinterpret_cast<const ChainPhase*>(base + desc_bytes);
    c->cnt_dev = reinterpret_cast<unsigned*>(base + desc_bytes + ph_bytes);
    c->n_ph = n_phases; c->n_img = n_img;
    c->grid = std::min(n_cu, max_blocks);              // every workgroup resident: one per CU (a workgroup asks for > 80 KiB of LDS or the grid is <= CUs anyway)
    c->lds_bytes = std::max(lds_max, 81 * 1024);       // > half of a CU's LDS: never two of these workgroups on one CU
    *handle_out = c;
    return 0;
}
extern "C" int mdm_chain_launch(void* handle, void* stream) {
    MDM_REQUIRE(handle, "chain_launch: null handle");
    const Chain* c = reinterpret_cast<const Chain*>(handle);
    hipStream_t s = pick_stream(stream);
    static int configured = 0;
    if (configured < c->lds_bytes) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_bytes));
        configured = c->lds_bytes;
    }
    const int n_cnt = c->n_ph * c->n_img;
    hipLaunchKernelGGL(chain_zero_kernel, dim3((unsigned)cdiv(n_cnt, 512)), dim3(512), 0, s, c->cnt_dev, n_cnt);
    hipLaunchKernelGGL(chain_kernel, dim3((unsigned)c->grid), dim3(512), (size_t)c->lds_bytes, s, c->descs_dev, c->phases_dev, c->n_ph, c->cnt_dev,
                       c->n_img, c->cnt_dev + n_cnt);
    return launch_status("chain");
}
// the error word of the chain's last launches (0 = every wait was satisfied); synchronises with the device
extern "C" int mdm_chain_status(void* handle, unsigned* err_out) {
    MDM_REQUIRE(handle && err_out, "chain_status: bad arguments");
    const Chain* c = reinterpret_cast<const Chain*>(handle);
    MDM_CHECK_HIP(hipMemcpy(err_out, c->cnt_dev + c->n_ph * c->n_img, 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int mdm_chain_destroy(void* handle) {
    delete reinterpret_cast<Chain*>(handle);
    return 0;
}

extern "C" int mdm_gemm_can_fuse_gn_bwd(const mdm_gemm_desc* desc_host, int G) {
    if (!desc_host || G <= 0) return 0;
    mdm_gemm_desc d = *desc_host;
    if (d.N0 == 0) d.N0 = d.N;
    if (!(d.transposed && d.N % G == 0 && d.N0 == d.N && d.C1 == 0 && !d.D1 && !d.bias && !d.rowvec && !d.resid && d.alpha == 1.0f)) return 0;
    const int cpg = d.N / G;
    if (!(cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32 || cpg == 64)) return 0;
    return (halo_tile(d) == 64 || lin2_gn_tile(d)) ? 1 : 0;
}
extern "C" int mdm_gemm_can_fuse_gn_fwd(const mdm_gemm_desc* desc_host, int G) {
    if (!desc_host || G <= 0) return 0;
    mdm_gemm_desc d = *desc_host;
    if (d.N0 == 0) d.N0 = d.N;
    if (!(!d.transposed && d.N % G == 0 && d.N0 == d.N && !d.D1 && !d.out_f32 && !d.acc0 && d.ldd0 == d.N)) return 0;
    const int cpg = d.N / G;
    if (!(cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32 || cpg == 64)) return 0;
    return (halo_tile(d) == 64 || lin2_gn_tile(d)) ? 1 : 0;
}
extern "C" int mdm_gemm_plan(const mdm_gemm_desc* desc_host, int* splitk_out, int64_t* ws_bytes_out) {
    if (!splitk_out || !ws_bytes_out) { set_error("gemm_plan: null output"); return -1; }
    Resolved r;
    if (int rc = resolve(desc_host, true, r)) return rc;
    *splitk_out = r.d.splitk;
    *ws_bytes_out = (r.d.ws && r.d.splitk > 1) ? (r.tap_split ? (int64_t)r.d.splitk * r.d.M * r.d.N * 4 : r.slab * r.d.splitk) : 0;
    return 0;
}

extern "C" int mdm_wgrad_group_accepts(const mdm_gemm_desc* desc_host) {
    if (!desc_host) return 0;
    mdm_gemm_desc d = *desc_host;
    if (d.N0 == 0) d.N0 = d.N;
    return wgrad_lin_eligible(d) && d.out_f32 && d.N0 == d.N && d.ldd0 == d.N && d.dtap == (int64_t)d.M * d.N ? 1 : 0;
}
// Schedule of a group's nine-tap layers: ONE queue per CU, walked by one persistent workgroup (grid = CUs, table[round][queue]).
// The (tile, slab) space is poured over the queues in order, every queue up to the common level -- a tile is cut where a queue
// is full, so the shares are equal by construction and a tile has only as many partial slots as CUs that worked on it.
// Costs in shader cycles per slab / per item, from in-kernel stamps (scripts/stamp_group.py).
static void build_taps_schedule(const std::vector<mdm_gemm_desc>& ds, const std::vector<TapsTile>& tt, const std::vector<GroupItem>& legacy, int NQ,
                                std::vector<int4>& table, std::vector<PartTile>& parts, int& nslots) {
    static const double CF = [] { const char* e = getenv("MDM_TAPS_SLAB_COST"); return e ? atof(e) : 4350.0; }();
    static const double FF = [] { const char* e = getenv("MDM_TAPS_ITEM_COST"); return e ? atof(e) : 25000.0; }();
    const int MINPART = 8;
    std::vector<std::vector<int4>> fq((size_t)NQ), lq((size_t)NQ);
    std::vector<double> load((size_t)NQ, 0.0);
    double total = 0.0;
    {   // the per-tap items of the group (1x1 projections, 8-channel / stride-2 / 4x4 layers) ride in the same queues: longest first
        // onto the least loaded one; the nine-tap shares then fill every queue up to the common level
        auto cost = [&](const GroupItem& gi) -> double {
            const mdm_gemm_desc& d = ds[(size_t)gi.desc];
            const int sk = d.splitk < 1 ? 1 : d.splitk;
            const int chunk = ((d.K + sk - 1) / sk + 63) / 64 * 64;
            const int ks = gi.item / (gi.tiles_x * d.KH * d.KW);
            int len = d.K - ks * chunk; if (len > chunk) len = chunk;
            const double slabs = len / 64.0;
            return gi.big == 2 ? slabs * 2500 + 11000 : gi.big == 1 ? slabs * 1510 + 5700 : slabs * 1050 + 3000;
        };
        std::vector<std::pair<double, int>> order;
        for (size_t a = 0; a < legacy.size(); ++a) order.push_back({cost(legacy[a]), (int)a});
        std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
        for (const auto& o : order) {
            int qq = 0;
            for (int x = 1; x < NQ; ++x) if (load[(size_t)x] < load[(size_t)qq]) qq = x;
            const GroupItem& gi = legacy[(size_t)o.second];
            lq[(size_t)qq].push_back(make_int4(gi.desc, gi.item, gi.tiles_x, gi.big));
            load[(size_t)qq] += o.first; total += o.first;
        }
    }
    double slabs_total = 0.0;
    for (const auto& t : tt) slabs_total += t.slabs;
    double level = (total + slabs_total * CF + ((double)tt.size() + NQ) * FF) / NQ;
    const std::vector<double> base_load = load;
    // pour the nine-tap tiles; a part is never shorter than MINPART slabs.  What does not fit (every queue keeps up to MINPART slabs
    // of room) would all land in the last queue: raise the level until the last queue is no higher than the others.
    for (int attempt = 0; attempt < 40; ++attempt) {
        load = base_load; parts.clear(); nslots = 0;
        for (auto& v : fq) v.clear();
        int q = 0;
        for (const auto& t : tt) {
            const mdm_gemm_desc& d = ds[(size_t)t.desc];
            struct Part { int q, k0, k1; };
            std::vector<Part> ps;
            int k = 0;
            while (k < t.slabs) {
                const int rem = t.slabs - k;
                int room = (int)((level - load[(size_t)q] - FF) / CF);
                if (q == NQ - 1) room = rem;
                if (room < MINPART && q < NQ - 1) { ++q; continue; }
                int take = room < rem ? room : rem;
                if (rem - take > 0 && rem - take < MINPART) take = rem;
                ps.push_back(Part{q, k, k + take});
                load[(size_t)q] += take * CF + FF;
                k += take;
            }
            const int tiles_n = cdiv(d.N, TAPS_BN), tm = t.tile / tiles_n, tn = t.tile - tm * tiles_n;
            if (ps.size() > 1) {
                PartTile pt;
                pt.dst = reinterpret_cast<float*>(d.D0); pt.dtap = d.dtap; pt.N = d.N; pt.m0 = tm * TAPS_BM; pt.n0 = tn * TAPS_BN;
                pt.first_slot = nslots; pt.parts = (int)ps.size();
                pt.rows = d.M - pt.m0 < TAPS_BM ? d.M - pt.m0 : TAPS_BM; pt.cols = d.N - pt.n0 < TAPS_BN ? d.N - pt.n0 : TAPS_BN;
                parts.push_back(pt);
            }
            for (size_t a2 = 0; a2 < ps.size(); ++a2) {
                const int slot1 = ps.size() > 1 ? nslots + (int)a2 + 1 : 0;
                fq[(size_t)ps[a2].q].push_back(make_int4(t.desc, t.tile | (slot1 << 12), ps[a2].k0 | (ps[a2].k1 << 16), 3));
            }
            if (ps.size() > 1) nslots += (int)ps.size();
        }
        if (load[(size_t)NQ - 1] <= level * 1.01) break;
        level *= 1.015;
    }
    size_t rounds = 0;
    for (int x = 0; x < NQ; ++x) {
        fq[(size_t)x].insert(fq[(size_t)x].end(), lq[(size_t)x].begin(), lq[(size_t)x].end());
        rounds = fq[(size_t)x].size() > rounds ? fq[(size_t)x].size() : rounds;
    }
    table.assign(rounds * (size_t)NQ, make_int4(-1, 0, 0, 0));
    for (int x = 0; x < NQ; ++x)
        for (size_t w = 0; w < fq[(size_t)x].size(); ++w) table[w * (size_t)NQ + (size_t)x] = fq[(size_t)x][w];
    if (getenv("MDM_TAPS_DEBUG")) {
        double lo = 1e30, hi = 0;
        for (int x = 0; x < NQ; ++x) { lo = load[(size_t)x] < lo ? load[(size_t)x] : lo; hi = load[(size_t)x] > hi ? load[(size_t)x] : hi; }
        fprintf(stderr, "[mdm] wgrad group: %zu nine-tap tiles (%.0f slabs) + %zu per-tap items, %d queues x %zu rounds, level %.0f, load %.0f..%.0f, %zu cut tiles, %d slots\n",
                tt.size(), slabs_total, legacy.size(), NQ, rounds, level, lo, hi, parts.size(), nslots);
    }
}

extern "C" int mdm_wgrad_group_create(const mdm_gemm_desc* descs_host, int n, void* dev_buf, int64_t dev_bytes,
                                      int64_t* need_bytes_out, void** handle_out) {
    MDM_REQUIRE(descs_host && n > 0 && need_bytes_out && handle_out, "wgrad_group_create: bad arguments");
    *handle_out = nullptr;
    std::vector<mdm_gemm_desc> ds((size_t)n);
    std::vector<GroupItem> items;
    std::vector<TapsTile> taps_tiles;
    WgradGroup* g = new WgradGroup();
    ReduceTable tab;
    tab.n = 0;
    int blocks = 0;
    auto close_table = [&]() {
        if (tab.n > 0) { tab.first_block[tab.n] = blocks; g->reduces.push_back(tab); g->reduce_blocks.push_back(blocks); }
        tab.n = 0; blocks = 0;
    };
    static const int n_cu_dev = [] {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) return 256;
        return pr.multiProcessorCount;
    }();
    // The nine-tap kernel pays ~30 000 cycles per work item around its loop (two DMA round trips + transposes in front, nine fp32 tiles
    // behind) and a launch that sums the cut tiles' slots; its loop is ~1.3x faster per flop than the per-tap tiles'.  Per group
    // (profiles/r03_wgrad_taps_stamps.txt): the decoder's 32x32 / 16x16 layers (94 slabs per CU) 392 -> 330 us, the encoder's (34 slabs per
    // CU) and the 8x8 maps' (10 per CU) came out 10-40 % slower.  So: only where a CU's share is long.
    const char* ms_env = getenv("MDM_TAPS_MIN_SHARE");          // (read per group: the kernel test forces the path on a small group)
    const int min_share = ms_env ? atoi(ms_env) : 48;
    // MDM_WGRAD_RESERVE_CUS=r (data-parallel runs): the persistent nine-tap launch is built for CUs - r workgroups, leaving r CUs
    // to whatever else wants to run beside it -- RCCL's all-reduce kernels of the previous bucket (bench.py --reserve-cus)
    const char* rs_env = getenv("MDM_WGRAD_RESERVE_CUS");
    const int n_cu = std::max(8, n_cu_dev - (rs_env ? std::max(0, atoi(rs_env)) : 0));
    long long taps_slabs = 0;
    for (int i = 0; i < n; ++i) {
        Resolved r;
        if (resolve(descs_host + i, false, r) == 0 && wgrad_taps_eligible(r.d)) taps_slabs += (long long)cdiv(r.d.M, TAPS_BM) * cdiv(r.d.N, TAPS_BN) * (r.d.K / 64);
    }
    const bool use_taps = taps_slabs >= (long long)min_share * n_cu;
    for (int i = 0; i < n; ++i) {
        Resolved r;
        if (int rc = resolve(descs_host + i, false, r)) { delete g; return rc; }
        if (!mdm_wgrad_group_accepts(&r.d)) { delete g; set_error("wgrad_group_create: descriptor %d is not a groupable weight gradient", i); return -1; }
        if (r.d.splitk > 1 && !r.d.ws) { delete g; set_error("wgrad_group_create: descriptor %d is split %d ways but has no workspace of its own", i, r.d.splitk); return -1; }
        if (use_taps && wgrad_taps_eligible(r.d)) {             // all nine taps per work item: cut into per-CU shares below, no split-K slabs
            r.d.splitk = 1;
            ds[(size_t)i] = r.d;
            const int tiles = cdiv(r.d.M, TAPS_BM) * cdiv(r.d.N, TAPS_BN);
            for (int tl = 0; tl < tiles; ++tl) taps_tiles.push_back(TapsTile{i, tl, r.d.K / 64});
            continue;
        }
        ds[(size_t)i] = r.d;
        const mdm_gemm_desc& d = r.d;
        const int BK = 64, sk = d.splitk < 1 ? 1 : d.splitk;
        const int chunk = ((d.K + sk - 1) / sk + BK - 1) / BK * BK;
        // tile: 256 x 128 where the filter has >= 256 output channels (a multiple of 256) and the reduction is long
        // (short reductions -- the 4x4 maps: 8 slabs -- take 64x64 tiles when a layer has to fill the chip on its own; in the static
        //  queues of a nine-tap group the big tiles' 2.7x fewer cycles per flop count instead)
        const bool big = r.big || (use_taps && d.M >= 128 && d.N >= 128 && sk == 1);
        const int tile_kind = !big ? 0 : (d.M % 256 == 0 && d.N >= 128 ? 2 : 1);
        const int tiles_i = tile_kind == 2 ? (d.M / 256) * cdiv(d.N, 128) : tile_kind == 1 ? cdiv(d.M, 128) * cdiv(d.N, 128) : (int)r.tiles;
        const int n_local = tiles_i * r.zouter * sk;
        for (int it = 0; it < n_local; ++it) {
            const int ks = it / (tiles_i * r.zouter);
            int len = d.K - ks * chunk;
            if (len > chunk) len = chunk;
            items.push_back(GroupItem{i, it, tiles_i, tile_kind, (len / BK) * (tile_kind == 2 ? 6 : tile_kind == 1 ? 4 : 1) + 2});
        }
        if (sk > 1) {
            const long long total4 = (long long)((int64_t)r.zouter * d.M * d.N / 4);
            const long long nb = (total4 + REDUCE_VEC_PER_BLOCK - 1) / REDUCE_VEC_PER_BLOCK;
            if (tab.n == REDUCE_MAX_SEGS || blocks + nb > (1ll << 30)) close_table();
            tab.first_block[tab.n] = blocks;
            tab.seg[tab.n++] = ReduceSeg{reinterpret_cast<const float*>(d.ws), reinterpret_cast<float*>(d.D0), total4, sk, d.acc0};
            blocks += (int)nb;
        }
    }
    close_table();
    // Order.  The items of one (layer, k-range) -- every filter tap x output tile -- read the same dY and input slabs,
    // so they should meet in ONE XCD's L2 (each of the 8 XCDs otherwise fetches the slabs for itself: this kernel was
    // 2.8 GB of the step's 6.3 GB of L2-side traffic).  Workgroup b runs on XCD b % 8 (observed dispatch rule, used for
    // speed only): the table is laid out [8 queues][maxlen], workgroup b takes entry (b % 8, b / 8).  Bundles go longest
    // first to the queue with the least work so far (short ones fill the tail); unused entries are no-ops (desc -1).
    std::vector<std::pair<long long, std::pair<int, int>>> bundles;            // (cost of one item, [first, last) in `items`)
    for (size_t a = 0; a < items.size();) {
        size_t b = a;
        const int per = items[a].tiles_x * (int)(ds[(size_t)items[a].desc].KH * ds[(size_t)items[a].desc].KW);
        while (b < items.size() && items[b].desc == items[a].desc && items[b].item / per == items[a].item / per) ++b;
        // (at most 12 items per bundle -- a few neighbouring taps x the output tiles: whole 18..72-item bundles balance
        // the 32 CUs of an XCD too coarsely, measured +1 % on the step)
        for (size_t c = a; c < b; c += 12) bundles.push_back({items[a].cost, {(int)c, (int)(c + 12 < b ? c + 12 : b)}});
        a = b;
    }
    std::stable_sort(bundles.begin(), bundles.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
    std::vector<std::vector<int>> queue(8);
    long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const auto& bd : bundles) {
        int q = 0;
        for (int x = 1; x < 8; ++x) if (load[x] < load[q]) q = x;
        for (int i = bd.second.first; i < bd.second.second; ++i) { queue[(size_t)q].push_back(i); load[q] += items[(size_t)i].cost; }
    }
    size_t maxlen = 0;
    for (const auto& qv : queue) maxlen = qv.size() > maxlen ? qv.size() : maxlen;
    const size_t n_slots_lin = 8 * maxlen;
    // the nine-tap layers: their own table, launch and partial slots
    std::vector<int4> taps_table;
    std::vector<PartTile> parts;
    int nslots = 0;
    const bool merged = !taps_tiles.empty();        // a group with nine-tap layers runs as ONE persistent launch, its per-tap items in the same queues
    if (merged) build_taps_schedule(ds, taps_tiles, items, n_cu, taps_table, parts, nslots);
    const size_t n_slots = merged ? 0 : n_slots_lin;
    auto pad256 = [](int64_t v) { return (v + 255) / 256 * 256; };
    const int64_t desc_bytes = pad256((int64_t)n * (int64_t)sizeof(mdm_gemm_desc));
    const int64_t item_bytes = pad256((int64_t)n_slots * 16), titem_bytes = pad256((int64_t)taps_table.size() * 16);
    const int64_t part_bytes = pad256((int64_t)parts.size() * (int64_t)sizeof(PartTile));
    const int64_t need = desc_bytes + item_bytes + titem_bytes + part_bytes + (int64_t)nslots * TAPS_SLOT_FLOATS * 4;
    *need_bytes_out = need;
    if (!dev_buf || dev_bytes < need) { delete g; return 0; }      // size query
    std::vector<int4> it4(n_slots, make_int4(-1, 0, 0, 0));
    for (int x = 0; x < 8 && !merged; ++x)
        for (size_t w = 0; w < queue[(size_t)x].size(); ++w) {
            const GroupItem& gi = items[(size_t)queue[(size_t)x][w]];
            it4[w * 8 + (size_t)x] = make_int4(gi.desc, gi.item, gi.tiles_x, gi.big);      // slot of workgroup b = w * 8 + x
        }
    char* base = reinterpret_cast<char*>(dev_buf);
    hipError_t e = hipMemcpy(base, ds.data(), (size_t)n * sizeof(mdm_gemm_desc), hipMemcpyHostToDevice);
    if (e == hipSuccess && !it4.empty()) e = hipMemcpy(base + desc_bytes, it4.data(), it4.size() * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && !taps_table.empty()) e = hipMemcpy(base + desc_bytes + item_bytes, taps_table.data(), taps_table.size() * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && !parts.empty()) e = hipMemcpy(base + desc_bytes + item_bytes + titem_bytes, parts.data(), parts.size() * sizeof(PartTile), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete g; return hip_fail(e, "wgrad_group_create: hipMemcpy"); }
    g->descs_dev = reinterpret_cast<const mdm_gemm_desc*>(base);
    g->items_dev = reinterpret_cast<const int4*>(base + desc_bytes);
    g->n_items = (int)n_slots;
    g->taps_items_dev = reinterpret_cast<const int4*>(base + desc_bytes + item_bytes);
    g->n_taps_items = (int)taps_table.size();
    g->taps_blocks = n_cu;
    g->parts_dev = reinterpret_cast<const PartTile*>(base + desc_bytes + item_bytes + titem_bytes);
    g->slots_dev = reinterpret_cast<float*>(base + desc_bytes + item_bytes + titem_bytes + part_bytes);
    g->n_part_tiles = (int)parts.size();
    *handle_out = g;
    return 0;
}
extern "C" int mdm_wgrad_group_launch(void* handle, void* stream) {
    MDM_REQUIRE(handle, "wgrad_group_launch: null handle");
    const WgradGroup* g = reinterpret_cast<const WgradGroup*>(handle);
    hipStream_t s = pick_stream(stream);
    constexpr int bytes = 3 * (256 + 128) * 64 * 2;        // the largest of the three rings (256x128: 3 stages of 48 KiB)
    static bool configured = false;
    if (!configured) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_taps_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    configured = true;
    if (g->n_taps_items > 0) {
        const int tb = g->taps_blocks < g->n_taps_items ? g->taps_blocks : g->n_taps_items;
        hipLaunchKernelGGL(wgrad_taps_group_kernel, dim3((unsigned)tb), dim3(512), bytes, s, g->descs_dev, g->taps_items_dev, g->n_taps_items, g->slots_dev);
    }
    if (g->n_items > 0) {
        const int nb = (g->max_blocks > 0 && g->max_blocks < g->n_items) ? g->max_blocks : g->n_items;
        hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)nb), dim3(512), bytes, s, g->descs_dev, g->items_dev, g->n_items);
    }
    for (size_t i = 0; i < g->reduces.size(); ++i)
        hipLaunchKernelGGL(splitk_reduce_batched_kernel, dim3((unsigned)g->reduce_blocks[i]), dim3(256), 0, s, g->reduces[i]);
    if (g->n_part_tiles > 0)
        hipLaunchKernelGGL(tile_parts_reduce_kernel, dim3((unsigned)(g->n_part_tiles * (TAPS_SLOT_FLOATS / 4 / 1024))), dim3(256), 0, s, g->parts_dev, g->slots_dev);
    return launch_status("wgrad group");
}
extern "C" int mdm_wgrad_group_destroy(void* handle) {
    delete reinterpret_cast<WgradGroup*>(handle);
    return 0;
}

#ifdef MDM_STAMP
extern "C" int mdm_debug_stamps_n(unsigned long long* out, int n_records, int reset) {      // out: n_records * 32 entries
    if (n_records < 1 || n_records > MDM_STAMP_RECS) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mdm::g_stamp_buf), (size_t)n_records * 32 * 8) != hipSuccess) return -1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mdm::g_stamp_buf)) != hipSuccess || hipMemset(p, 0, (size_t)n_records * 32 * 8) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int mdm_debug_stamps(unsigned long long* out, int reset) { return mdm_debug_stamps_n(out, 4096, reset); }
#endif
