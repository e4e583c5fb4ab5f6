// Fused single-head self-attention for the AttentionBlock of unet6 (reference unet6.py:316-333):
//     w = softmax(q k^T / sqrt(C)) ;  o = w v          over L = H*W tokens, one head of width C,
// on qkv = project_in(norm(x)) stored NHWC as [N][L][3C] (q | k | v), bf16.
//
// The unfused path (two batched contractions + a softmax launch, S[N][L][L] materialised in HBM; five more
// launches and dP[N][L][L] in the backward) is replaced by ONE forward kernel and ONE backward launch (dQ and dK/dV workgroups side by side) that
// never write the scores: per 64-query tile the keys are walked in tiles of 64 with an online softmax
// (running max m, running sum l per query), the backward recomputes P from q, k and the saved
// log-sum-exp.  QK^T, PV and the four backward products run on v_mfma_f32_16x16x32_bf16.
//
// Orientation.  Every product is computed TRANSPOSED (operands swapped) so that the query (dQ kernel, forward)
// or the key (dK/dV kernel) sits on the MFMA lane (lane & 15) and the other index on the accumulator registers:
//   * row statistics (max, sum, LSE, delta) are lane-local -- two __shfl_xor (16, 32) join the four lane groups;
//   * the accumulator of the first product IS the B operand of the second one (cvt to bf16, no LDS round trip):
//     registers r of blocks 2s and 2s+1 of lane group g form k-step s with the key order
//     k = 8 g + j  <->  key 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4); the A operand of the second product is
//     read from LDS in the SAME order with ds_read_b64_tr_b16 (4 consecutive rows per read), which also does
//     the transposition (V^T, K^T, dO^T, Q^T) for free.
// LDS tiles are [64 rows][C] bf16 with the 32-byte pair index XOR-ed by (row & 7): the transposing reads are
// conflict-free, the ds_read_b128 row reads 2-way.
//
// 256 threads = 4 waves, one per SIMD (the accumulators of the C = 256 backward need the whole register file);
// wave w owns 16 of the tile's 64 queries (keys in the dK/dV kernel).  L must be a multiple of 16.
#include "common.h"

namespace mdm {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4_t;

template <int D>
__device__ __forceinline__ int tile_off(int row, int chunk) {          // byte offset of 16-byte chunk `chunk` of tile row `row`
    constexpr int PPR = D / 16, MASK = (PPR - 1) < 7 ? (PPR - 1) : 7;
    return row * (2 * D) + ((((chunk >> 1) ^ (row & MASK)) << 5) | ((chunk & 1) << 4));
}

// 64 rows x D of src (row pitch ld elements; rows beyond `nrows` repeat the last valid one: they are masked later)
template <int D>
__device__ __forceinline__ void load_tile(char* lds, const bf16_t* src, int row0, int nrows, int ld, int t) {
    constexpr int CPR = D / 8;
#pragma unroll
    for (int idx = t; idx < 64 * CPR; idx += 256) {
        const int row = idx / CPR, c = idx - row * CPR;
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)gr * ld + c * 8);
        *reinterpret_cast<uint4*>(lds + tile_off<D>(row, c)) = v;
    }
}
// row-read fragment: lane -> row r0 + (lane & 15), elements d = 32 ks + 8 (lane >> 4) .. + 7
template <int D>
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int r0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + tile_off<D>(r0 + (lane & 15), ks * 4 + (lane >> 4)));
}
// transposed fragment: element j of lane l = tile[row(j)][d0 + (l & 15)], row(j) = 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4), g = l >> 4
template <int D>
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int s, int d0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int r0 = 32 * s + 4 * g + q, r1 = r0 + 16;
    const int chunk = (d0 >> 3) + (p >> 1), sub = (p & 1) * 8;
    const char* p0 = tile + tile_off<D>(r0, chunk) + sub;
    const char* p1 = tile + tile_off<D>(r1, chunk) + sub;
    bf4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(p0));
    bf4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf4_t __attribute__((address_space(3)))*)(p1));
    bf16x4 l4 = *reinterpret_cast<bf16x4*>(&lo), h4 = *reinterpret_cast<bf16x4*>(&hi);
    bf16x8 f;
    f[0] = l4[0]; f[1] = l4[1]; f[2] = l4[2]; f[3] = l4[3];
    f[4] = h4[0]; f[5] = h4[1]; f[6] = h4[2]; f[7] = h4[3];
    return f;
}
// two accumulator blocks (rows 4g+r of 16-row blocks 2s, 2s+1) -> the bf16 B fragment of k-step s
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
    bf16x8 f;
    f[0] = (short)f2bf(a[0]); f[1] = (short)f2bf(a[1]); f[2] = (short)f2bf(a[2]); f[3] = (short)f2bf(a[3]);
    f[4] = (short)f2bf(b[0]); f[5] = (short)f2bf(b[1]); f[6] = (short)f2bf(b[2]); f[7] = (short)f2bf(b[3]);
    return f;
}
__device__ __forceinline__ float group_max(float v) {       // over the 4 lane groups that share (lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ void store4bf(bf16_t* p, const f32x4& v) {
    uint2 r;
    r.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    r.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = r;
}
constexpr float NEG_BIG = -1.0e30f;

// ------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                       int L, float scale) {
    constexpr int KS = D / 32, DB = D / 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Kt = lds;
    char* Vt = lds + 64 * 2 * D;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
    const int n = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * D;
    const bf16_t* base = qkv + (int64_t)n * L * ld;
    int qrow = q0 + (lane & 15);
    const bool q_ok = qrow < L;
    qrow = q_ok ? qrow : L - 1;
    bf16x8 Qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qrow * ld + ks * 32 + 8 * g);
    f32x4 O[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) O[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = NEG_BIG, lsum = 0.f;
    for (int k0 = 0; k0 < L; k0 += 64) {
        __syncthreads();
        load_tile<D>(Kt, base + D, k0, L, ld, t);
        load_tile<D>(Vt, base + 2 * D, k0, L, ld, t);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            s[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Kt, 16 * jb, ks, lane), Qf[ks], s[jb], 0, 0, 0);
        }
        float tmax = NEG_BIG;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = k0 + 16 * jb + 4 * g + r < L;
                s[jb][r] = ok ? s[jb][r] * scale : NEG_BIG;
                tmax = fmaxf(tmax, s[jb][r]);
            }
        tmax = group_max(tmax);
        const float m_new = fmaxf(m, tmax), alpha = __expf(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[jb][r] - m_new);
                s[jb][r] = p;
                rs += p;
            }
        rs = group_sum(rs);
        lsum = lsum * alpha + rs;
        m = m_new;
#pragma unroll
        for (int db = 0; db < DB; ++db) { O[db][0] *= alpha; O[db][1] *= alpha; O[db][2] *= alpha; O[db][3] *= alpha; }
        const bf16x8 pf0 = pack_pair(s[0], s[1]), pf1 = pack_pair(s[2], s[3]);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            O[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Vt, 0, 16 * db, lane), pf0, O[db], 0, 0, 0);
            O[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Vt, 1, 16 * db, lane), pf1, O[db], 0, 0, 0);
        }
    }
    if (q_ok) {
        const float inv = 1.f / lsum;
        bf16_t* orow = o + ((int64_t)n * L + qrow) * D;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            f32x4 v = O[db];
            v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
            store4bf(orow + 16 * db + 4 * g, v);
        }
        if (g == 0) lse[(int64_t)n * L + qrow] = m + __logf(lsum);
    }
}


// ------------------------------------------------------------------------------------------------ forward, K / V by LDS-DMA
// L >= 256 (cfg4: L = 1024, cfg3: L = 256): attn_fwd_kernel stages every 64-key tile through registers between two barriers --
// load, store, barrier, multiply -- and ran at 170 TFLOP/s at L = 1024 (8.6 GFLOP in 50 us; profiles/r03_cfg4_kernel_stats.csv):
// 16 dependent round trips to L2 per workgroup.  Here the K and V tiles of key tile j+1 (j+2 where three stages fit: D <= 128)
// arrive by LDS-DMA (global_load_lds_dwordx4) while tile j multiplies; counted vmcnt, one barrier per tile.  Same tile layout
// (tile_off: the DMA's per-lane SOURCE address carries the swizzle), same arithmetic, same orientation as attn_fwd_kernel; the
// transposing V reads go through inline asm (the builtin form makes hipcc drain every DMA in flight, gemm.hip finding 32).
// Requires L % 64 == 0.
__device__ __forceinline__ void attn_dma16(const void* src, char* lds_piece_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_piece_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef unsigned av2u_t __attribute__((ext_vector_type(2)));
typedef unsigned av4u_t __attribute__((ext_vector_type(4)));
struct ATrFrag { av2u_t lo, hi; };
template <int D>
__device__ __forceinline__ void frag_tr_issue(const char* tile, int s, int d0, int lane, ATrFrag& f) {    // frag_tr's two reads, not waited for
    const int i = lane & 15, g = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int r0 = 32 * s + 4 * g + q, r1 = r0 + 16;
    const int chunk = (d0 >> 3) + (p >> 1), sub = (p & 1) * 8;
    const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tile + tile_off<D>(r0, chunk) + sub);
    const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tile + tile_off<D>(r1, chunk) + sub);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1) : "memory");
}
__device__ __forceinline__ bf16x8 atr_value(const ATrFrag& f) {
    const av4u_t v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_dma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                           int L, float scale) {
    constexpr int KS = D / 32, DB = D / 16, TILE = 64 * 2 * D, NST = D <= 128 ? 3 : 2;
    constexpr int PPT = TILE / 1024, PW = PPT / 4;                 // 1-KiB pieces per K (or V) tile; per wave
    constexpr int CPR = D / 8, RPP = 1024 / (2 * D);               // 16-byte chunks per tile row; tile rows per piece
    constexpr int PPR = D / 16, MASK = (PPR - 1) < 7 ? (PPR - 1) : 7;
    static_assert(PW >= 1 && RPP >= 1, "attn_fwd_dma: D in {64, 128, 256}");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int t = threadIdx.x, lane = t & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * D;
    const bf16_t* base = qkv + (int64_t)n * L * ld;
    const int qrow = q0 + (lane & 15);                             // < L: L % 64 == 0
    const int ntiles = L >> 6;
    // this lane's slot in each of its pieces: tile row, and which source chunk the tile layout wants there
    int64_t src_off[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        const int row = (wave * PW + k) * RPP + lane / CPR, pos = lane % CPR;
        const int c = (((pos >> 1) ^ (row & MASK)) << 1) | (pos & 1);
        src_off[k] = ((int64_t)row * ld + c * 8) * 2;
    }
    const char* const kbase = reinterpret_cast<const char*>(base + D);
    auto issue_tile = [&](int j, int stage) {
        const int jt = j < ntiles ? j : ntiles - 1;                // (beyond the end: the last tile again, into a stage nobody reads)
        const char* src = kbase + (int64_t)jt * 64 * ld * 2;
        char* dst = lds + stage * 2 * TILE + wave * PW * 1024;
#pragma unroll
        for (int k = 0; k < PW; ++k) attn_dma16(src + src_off[k], dst + k * 1024);
#pragma unroll
        for (int k = 0; k < PW; ++k) attn_dma16(src + src_off[k] + D * 2, dst + TILE + k * 1024);
    };
    issue_tile(0, 0);
    if (NST == 3) issue_tile(1, 1);
    bf16x8 Qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qrow * ld + ks * 32 + 8 * g);
    f32x4 O[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) O[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = NEG_BIG, lsum = 0.f;
    int stage = 0;
    for (int j = 0; j < ntiles; ++j) {
        if (NST == 3) attn_wait_vmcnt<2 * PW>(); else attn_wait_vmcnt<0>();       // tile j of this wave has landed (tile j+1 may be in flight)
        __builtin_amdgcn_s_barrier();                              // ... of every wave; and everybody is done with tile j-1
        const int ns = stage + NST - 1 >= NST ? stage - 1 : stage + NST - 1;
        issue_tile(j + NST - 1, ns);
        const char* Kt = lds + stage * 2 * TILE;
        const char* Vt = Kt + TILE;
        f32x4 s[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            s[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Kt, 16 * jb, ks, lane), Qf[ks], s[jb], 0, 0, 0);
        }
        float tmax = NEG_BIG;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[jb][r] *= scale;
                tmax = fmaxf(tmax, s[jb][r]);
            }
        tmax = group_max(tmax);
        const float m_new = fmaxf(m, tmax), alpha = __expf(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(s[jb][r] - m_new);
                s[jb][r] = pv;
                rs += pv;
            }
        rs = group_sum(rs);
        lsum = lsum * alpha + rs;
        m = m_new;
#pragma unroll
        for (int db = 0; db < DB; ++db) { O[db][0] *= alpha; O[db][1] *= alpha; O[db][2] *= alpha; O[db][3] *= alpha; }
        const bf16x8 pf0 = pack_pair(s[0], s[1]), pf1 = pack_pair(s[2], s[3]);
        constexpr int CH = DB < 4 ? DB : 4;                        // V fragments of four 16-channel blocks in flight at a time
#pragma unroll
        for (int d0 = 0; d0 < DB; d0 += CH) {
            ATrFrag vf[CH][2];
#pragma unroll
            for (int c = 0; c < CH; ++c) { frag_tr_issue<D>(Vt, 0, 16 * (d0 + c), lane, vf[c][0]); frag_tr_issue<D>(Vt, 1, 16 * (d0 + c), lane, vf[c][1]); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                O[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(vf[c][0]), pf0, O[d0 + c], 0, 0, 0);
                O[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(vf[c][1]), pf1, O[d0 + c], 0, 0, 0);
            }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    attn_wait_vmcnt<0>();
    const float inv = 1.f / lsum;
    bf16_t* orow = o + ((int64_t)n * L + qrow) * D;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        f32x4 v = O[db];
        v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
        store4bf(orow + 16 * db + 4 * g, v);
    }
    if (g == 0) lse[(int64_t)n * L + qrow] = m + __logf(lsum);
}

// ------------------------------------------------------------------------------------------------ backward: dQ (and delta)
template <int D>
__device__ __forceinline__ void attn_bwd_dq_body(char* lds, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                 const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                 float* __restrict__ delta, bf16_t* __restrict__ dqkv, int L, float scale) {
    constexpr int KS = D / 32, DB = D / 16;
    char* Kt = lds;
    char* Vt = lds + 64 * 2 * D;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
    const int n = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * D;
    const bf16_t* base = qkv + (int64_t)n * L * ld;
    int qrow = q0 + (lane & 15);
    const bool q_ok = qrow < L;
    qrow = q_ok ? qrow : L - 1;
    bf16x8 Qf[KS], dOf[KS];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qrow * ld + ks * 32 + 8 * g);
        dOf[ks] = *reinterpret_cast<const bf16x8*>(d_o + ((int64_t)n * L + qrow) * D + ks * 32 + 8 * g);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + ((int64_t)n * L + qrow) * D + ks * 32 + 8 * g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl = fmaf(bf2f((bf16_t)dOf[ks][e]), bf2f((bf16_t)of[e]), dl);
    }
    dl = group_sum(dl);                               // delta[q] = sum_d dO[q][d] O[q][d]
    const float ls = lse[(int64_t)n * L + qrow];
    if (q_ok && g == 0) delta[(int64_t)n * L + qrow] = dl;
    f32x4 dQ[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) dQ[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += 64) {
        __syncthreads();
        load_tile<D>(Kt, base + D, k0, L, ld, t);
        load_tile<D>(Vt, base + 2 * D, k0, L, ld, t);
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            s[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Kt, 16 * jb, ks, lane), Qf[ks], s[jb], 0, 0, 0);
                dp[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Vt, 16 * jb, ks, lane), dOf[ks], dp[jb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = k0 + 16 * jb + 4 * g + r < L;
                const float p = ok ? __expf(s[jb][r] * scale - ls) : 0.f;
                s[jb][r] = p * (dp[jb][r] - dl) * scale;                   // dS
            }
        const bf16x8 f0 = pack_pair(s[0], s[1]), f1 = pack_pair(s[2], s[3]);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            dQ[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Kt, 0, 16 * db, lane), f0, dQ[db], 0, 0, 0);
            dQ[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Kt, 1, 16 * db, lane), f1, dQ[db], 0, 0, 0);
        }
    }
    if (q_ok) {
        bf16_t* row = dqkv + ((int64_t)n * L + qrow) * ld;
#pragma unroll
        for (int db = 0; db < DB; ++db) store4bf(row + 16 * db + 4 * g, dQ[db]);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// delta[q] = sum_d dO[q][d] O[q][d] is recomputed here for the image's L queries (4 threads per query, into LDS) rather than
// read from the dQ workgroups: the two halves of the backward then have no dependency and run as ONE launch
template <int D>
__device__ __forceinline__ void attn_bwd_dkv_body(char* lds, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                  const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                  bf16_t* __restrict__ dqkv, int L, float scale) {
    constexpr int KS = D / 32, DB = D / 16;
    char* Qt = lds;
    char* Gt = lds + 64 * 2 * D;           // dO tile
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4;
    const int n = blockIdx.y, kk0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * D;
    const bf16_t* base = qkv + (int64_t)n * L * ld;
    int krow = kk0 + (lane & 15);
    const bool k_ok = krow < L;
    krow = k_ok ? krow : L - 1;
    bf16x8 Kf[KS], Vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Kf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)krow * ld + D + ks * 32 + 8 * g);
        Vf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)krow * ld + 2 * D + ks * 32 + 8 * g);
    }
    f32x4 dK[DB], dV[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { dK[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float* lse_n = lse + (int64_t)n * L;
    float* del_n = reinterpret_cast<float*>(lds + 2 * 64 * 2 * D);
    for (int q = t >> 2; q < L; q += 64) {
        const int part = t & 3;
        const bf16_t* po = o + ((int64_t)n * L + q) * D + part * (D / 4);
        const bf16_t* pg = d_o + ((int64_t)n * L + q) * D + part * (D / 4);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < D / 4; e += 8) {
            const float8 a = load8(po + e), b = load8(pg + e);
            sum = fmaf(a.lo.x, b.lo.x, sum); sum = fmaf(a.lo.y, b.lo.y, sum); sum = fmaf(a.lo.z, b.lo.z, sum); sum = fmaf(a.lo.w, b.lo.w, sum);
            sum = fmaf(a.hi.x, b.hi.x, sum); sum = fmaf(a.hi.y, b.hi.y, sum); sum = fmaf(a.hi.z, b.hi.z, sum); sum = fmaf(a.hi.w, b.hi.w, sum);
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (part == 0) del_n[q] = sum;
    }
    // (the first __syncthreads of the loop below orders these writes before any read)
    for (int q0 = 0; q0 < L; q0 += 64) {
        __syncthreads();
        load_tile<D>(Qt, base, q0, L, ld, t);
        load_tile<D>(Gt, d_o + (int64_t)n * L * D, q0, L, D, t);
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            s[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Qt, 16 * qb, ks, lane), Kf[ks], s[qb], 0, 0, 0);
                dp[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Gt, 16 * qb, ks, lane), Vf[ks], dp[qb], 0, 0, 0);
            }
        }
        // lane: key (lane & 15), queries q0 + 16 qb + 4 g + r
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int qa = q0 + 16 * qb + 4 * g;          // 4 consecutive queries; L % 16 == 0 -> all valid or all invalid
            const bool ok = qa < L;
            const float4 l4 = ok ? *reinterpret_cast<const float4*>(lse_n + qa) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 d4 = ok ? *reinterpret_cast<const float4*>(del_n + qa) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = ok ? __expf(s[qb][r] * scale - lq[r]) : 0.f;
                s[qb][r] = p;                                            // P
                dp[qb][r] = p * (dp[qb][r] - dq[r]) * scale;             // dS
            }
        }
        const bf16x8 p0 = pack_pair(s[0], s[1]), p1 = pack_pair(s[2], s[3]);
        const bf16x8 e0 = pack_pair(dp[0], dp[1]), e1 = pack_pair(dp[2], dp[3]);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            dV[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Gt, 0, 16 * db, lane), p0, dV[db], 0, 0, 0);
            dV[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Gt, 1, 16 * db, lane), p1, dV[db], 0, 0, 0);
            dK[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Qt, 0, 16 * db, lane), e0, dK[db], 0, 0, 0);
            dK[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(Qt, 1, 16 * db, lane), e1, dK[db], 0, 0, 0);
        }
    }
    if (k_ok) {
        bf16_t* row = dqkv + ((int64_t)n * L + krow) * ld;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            store4bf(row + D + 16 * db + 4 * g, dK[db]);
            store4bf(row + 2 * D + 16 * db + 4 * g, dV[db]);
        }
    }
}


// ------------------------------------------------------------------------------------------------ backward, tiles by LDS-DMA
// The two backward bodies with their 64-row tiles (K and V for dQ; Q and dO for dK / dV) prefetched by LDS-DMA like
// attn_fwd_dma_kernel: same tile layout, arithmetic and orientation as attn_bwd_dq_body / attn_bwd_dkv_body.  L % 64 == 0.
template <int D>
struct AttnPipe {              // two operand tiles per step, NST stages; every wave issues 2 * PW pieces per step
    static constexpr int TILE = 64 * 2 * D, NST = D <= 128 ? 3 : 2, PPT = TILE / 1024, PW = PPT / 4;
    static constexpr int CPR = D / 8, RPP = 1024 / (2 * D), PPR = D / 16, MASK = (PPR - 1) < 7 ? (PPR - 1) : 7;
    int64_t off_a[PW], off_b[PW];
    const char *src_a, *src_b;
    int64_t step_a, step_b;
    int wave, ntiles;
    __device__ __forceinline__ void init(int lane, int wave_, const void* a, int ld_a, const void* b, int ld_b, int ntiles_) {
        wave = wave_; ntiles = ntiles_;
        src_a = reinterpret_cast<const char*>(a); src_b = reinterpret_cast<const char*>(b);
        step_a = (int64_t)64 * ld_a * 2; step_b = (int64_t)64 * ld_b * 2;
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int row = (wave * PW + k) * RPP + lane / CPR, pos = lane % CPR;
            const int c = (((pos >> 1) ^ (row & MASK)) << 1) | (pos & 1);
            off_a[k] = ((int64_t)row * ld_a + c * 8) * 2;
            off_b[k] = ((int64_t)row * ld_b + c * 8) * 2;
        }
    }
    __device__ __forceinline__ void issue(char* lds, int j, int stage) const {
        const int jt = j < ntiles ? j : ntiles - 1;
        char* dst = lds + stage * 2 * TILE + wave * PW * 1024;
#pragma unroll
        for (int k = 0; k < PW; ++k) attn_dma16(src_a + jt * step_a + off_a[k], dst + k * 1024);
#pragma unroll
        for (int k = 0; k < PW; ++k) attn_dma16(src_b + jt * step_b + off_b[k], dst + TILE + k * 1024);
    }
    // top of step j: this wave's tiles j have landed (NST == 3: those of j+1 may fly), every wave is done with j-1, then prefetch
    __device__ __forceinline__ void step(char* lds, int j, int& stage_next) const {
        if (NST == 3) attn_wait_vmcnt<2 * PW>(); else attn_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        issue(lds, j + NST - 1, stage_next);
    }
};
template <int D, int NF>       // NF transposed fragments of `tile` (k-step s, 16-row blocks d0 ..) -> registers, waited for
__device__ __forceinline__ void frag_tr_burst(const char* tile, int db0, int lane, ATrFrag (&f)[NF][2]) {
#pragma unroll
    for (int c = 0; c < NF; ++c) { frag_tr_issue<D>(tile, 0, 16 * (db0 + c), lane, f[c][0]); frag_tr_issue<D>(tile, 1, 16 * (db0 + c), lane, f[c][1]); }
}
__device__ __forceinline__ void atr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int D>
__device__ __forceinline__ void attn_bwd_dq_dma_body(char* lds, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                     const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                     float* __restrict__ delta, bf16_t* __restrict__ dqkv, int L, float scale) {
    constexpr int KS = D / 32, DB = D / 16, TILE = 64 * 2 * D, NST = AttnPipe<D>::NST;
    const int t = threadIdx.x, lane = t & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * D;
    const bf16_t* base = qkv + (int64_t)n * L * ld;
    const int qrow = q0 + (lane & 15);
    AttnPipe<D> pipe;
    pipe.init(lane, wave, base + D, ld, base + 2 * D, ld, L >> 6);
    pipe.issue(lds, 0, 0);
    if (NST == 3) pipe.issue(lds, 1, 1);
    bf16x8 Qf[KS], dOf[KS];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Qf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qrow * ld + ks * 32 + 8 * g);
        dOf[ks] = *reinterpret_cast<const bf16x8*>(d_o + ((int64_t)n * L + qrow) * D + ks * 32 + 8 * g);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + ((int64_t)n * L + qrow) * D + ks * 32 + 8 * g);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl = fmaf(bf2f((bf16_t)dOf[ks][e]), bf2f((bf16_t)of[e]), dl);
    }
    dl = group_sum(dl);
    const float ls = lse[(int64_t)n * L + qrow];
    if (g == 0) delta[(int64_t)n * L + qrow] = dl;
    f32x4 dQ[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) dQ[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int stage = 0;
    const int ntiles = L >> 6;
    for (int j = 0; j < ntiles; ++j) {
        int ns = stage + NST - 1 >= NST ? stage - 1 : stage + NST - 1;
        pipe.step(lds, j, ns);
        const char* Kt = lds + stage * 2 * TILE;
        const char* Vt = Kt + TILE;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            s[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Kt, 16 * jb, ks, lane), Qf[ks], s[jb], 0, 0, 0);
                dp[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Vt, 16 * jb, ks, lane), dOf[ks], dp[jb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[jb][r] * scale - ls);
                s[jb][r] = p * (dp[jb][r] - dl) * scale;                   // dS
            }
        const bf16x8 f0 = pack_pair(s[0], s[1]), f1 = pack_pair(s[2], s[3]);
        constexpr int CH = DB < 4 ? DB : 4;
#pragma unroll
        for (int d0 = 0; d0 < DB; d0 += CH) {
            ATrFrag kf[CH][2];
            frag_tr_burst<D, CH>(Kt, d0, lane, kf);
            atr_wait();
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                dQ[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(kf[c][0]), f0, dQ[d0 + c], 0, 0, 0);
                dQ[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(kf[c][1]), f1, dQ[d0 + c], 0, 0, 0);
            }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    attn_wait_vmcnt<0>();
    bf16_t* row = dqkv + ((int64_t)n * L + qrow) * ld;
#pragma unroll
    for (int db = 0; db < DB; ++db) store4bf(row + 16 * db + 4 * g, dQ[db]);
}

template <int D>
__device__ __forceinline__ void attn_bwd_dkv_dma_body(char* lds, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                      const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                      bf16_t* __restrict__ dqkv, int L, float scale) {
    constexpr int KS = D / 32, DB = D / 16, TILE = 64 * 2 * D, NST = AttnPipe<D>::NST;
    const int t = threadIdx.x, lane = t & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = blockIdx.y, kk0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * D;
    const bf16_t* base = qkv + (int64_t)n * L * ld;
    const int krow = kk0 + (lane & 15);
    AttnPipe<D> pipe;
    pipe.init(lane, wave, base, ld, d_o + (int64_t)n * L * D, D, L >> 6);
    pipe.issue(lds, 0, 0);
    if (NST == 3) pipe.issue(lds, 1, 1);
    bf16x8 Kf[KS], Vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Kf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)krow * ld + D + ks * 32 + 8 * g);
        Vf[ks] = *reinterpret_cast<const bf16x8*>(base + (int64_t)krow * ld + 2 * D + ks * 32 + 8 * g);
    }
    f32x4 dK[DB], dV[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { dK[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float* lse_n = lse + (int64_t)n * L;
    float* del_n = reinterpret_cast<float*>(lds + NST * 2 * TILE);
    float* lse_s = del_n + L;                        // the log-sum-exps too: a global load consumed inside the loop would drain the DMA ring
    for (int q = t >> 2; q < L; q += 64) {           // delta of the image's queries into LDS (see attn_bwd_dkv_body)
        const int part = t & 3;
        const bf16_t* po = o + ((int64_t)n * L + q) * D + part * (D / 4);
        const bf16_t* pg = d_o + ((int64_t)n * L + q) * D + part * (D / 4);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < D / 4; e += 8) {
            const float8 a = load8(po + e), b = load8(pg + e);
            sum = fmaf(a.lo.x, b.lo.x, sum); sum = fmaf(a.lo.y, b.lo.y, sum); sum = fmaf(a.lo.z, b.lo.z, sum); sum = fmaf(a.lo.w, b.lo.w, sum);
            sum = fmaf(a.hi.x, b.hi.x, sum); sum = fmaf(a.hi.y, b.hi.y, sum); sum = fmaf(a.hi.z, b.hi.z, sum); sum = fmaf(a.hi.w, b.hi.w, sum);
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (part == 0) { del_n[q] = sum; lse_s[q] = lse_n[q]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the LDS writes above; the loop's first barrier orders them before any read)
    int stage = 0;
    const int ntiles = L >> 6;
    for (int j = 0; j < ntiles; ++j) {
        int ns = stage + NST - 1 >= NST ? stage - 1 : stage + NST - 1;
        pipe.step(lds, j, ns);
        const char* Qt = lds + stage * 2 * TILE;
        const char* Gt = Qt + TILE;
        const int q0 = j * 64;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            s[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Qt, 16 * qb, ks, lane), Kf[ks], s[qb], 0, 0, 0);
                dp[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(Gt, 16 * qb, ks, lane), Vf[ks], dp[qb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int qa = q0 + 16 * qb + 4 * g;
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qa);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_n + qa);
            const float lq[4] = {l4[0], l4[1], l4[2], l4[3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[qb][r] * scale - lq[r]);
                s[qb][r] = p;                                            // P
                dp[qb][r] = p * (dp[qb][r] - d4[r]) * scale;             // dS
            }
        }
        const bf16x8 p0 = pack_pair(s[0], s[1]), p1 = pack_pair(s[2], s[3]);
        const bf16x8 e0 = pack_pair(dp[0], dp[1]), e1 = pack_pair(dp[2], dp[3]);
        constexpr int CH = DB < 2 ? DB : 2;
#pragma unroll
        for (int d0 = 0; d0 < DB; d0 += CH) {
            ATrFrag gf[CH][2], qf[CH][2];
            frag_tr_burst<D, CH>(Gt, d0, lane, gf);
            frag_tr_burst<D, CH>(Qt, d0, lane, qf);
            atr_wait();
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                dV[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(gf[c][0]), p0, dV[d0 + c], 0, 0, 0);
                dV[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(gf[c][1]), p1, dV[d0 + c], 0, 0, 0);
                dK[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(qf[c][0]), e0, dK[d0 + c], 0, 0, 0);
                dK[d0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(atr_value(qf[c][1]), e1, dK[d0 + c], 0, 0, 0);
            }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    attn_wait_vmcnt<0>();
    bf16_t* row = dqkv + ((int64_t)n * L + krow) * ld;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        store4bf(row + D + 16 * db + 4 * g, dK[db]);
        store4bf(row + 2 * D + 16 * db + 4 * g, dV[db]);
    }
}
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                           const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16_t* __restrict__ dqkv, int L, float scale) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    if (blockIdx.z == 0) attn_bwd_dq_dma_body<D>(lds, qkv, o, d_o, lse, delta, dqkv, L, scale);
    else attn_bwd_dkv_dma_body<D>(lds, qkv, o, d_o, lse, dqkv, L, scale);
}

// blockIdx.z = 0: dQ (and delta, kept as an output), 1: dK and dV
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                       const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                       float* __restrict__ delta, bf16_t* __restrict__ dqkv, int L, float scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (blockIdx.z == 0) attn_bwd_dq_body<D>(lds, qkv, o, d_o, lse, delta, dqkv, L, scale);
    else attn_bwd_dkv_body<D>(lds, qkv, o, d_o, lse, dqkv, L, scale);
}

#ifndef MDM_ATTN_DMA
#define MDM_ATTN_DMA 1
#endif
template <int D>
static int attn_launch(int which, const bf16_t* qkv, bf16_t* o, const bf16_t* d_o, float* lse, float* delta, bf16_t* dqkv,
                       int N, int L, float scale, hipStream_t s) {
    constexpr int bytes = 2 * 64 * 2 * D;
    static bool configured = false;
    if (!configured) {
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes + 4 * 4096));
        configured = true;
    }
    dim3 grid((unsigned)cdiv(L, 64), (unsigned)N);
    if (which == 0) {
        if constexpr (D >= 64) {
            if (MDM_ATTN_DMA && L % 64 == 0 && L >= 256) {         // several key tiles: K / V prefetched by LDS-DMA
                constexpr int dma_bytes = (D <= 128 ? 3 : 2) * 2 * 64 * 2 * D;
                static bool dma_configured = false;
                if (!dma_configured) {
                    MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_dma_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_bytes));
                    dma_configured = true;
                }
                hipLaunchKernelGGL((attn_fwd_dma_kernel<D>), grid, dim3(256), dma_bytes, s, qkv, o, lse, L, scale);
                return launch_status("attention");
            }
        }
        hipLaunchKernelGGL((attn_fwd_kernel<D>), grid, dim3(256), bytes, s, qkv, o, lse, L, scale);
    } else {
        MDM_REQUIRE(L <= 4096, "attention backward: L=%d > 4096", L);
        if constexpr (D >= 64) {
            if (MDM_ATTN_DMA && L % 64 == 0 && L >= 256) {
                constexpr int dma_bytes = (D <= 128 ? 3 : 2) * 2 * 64 * 2 * D;
                static bool dma_configured = false;
                if (!dma_configured) {
                    MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dma_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      dma_bytes + 8 * 4096));
                    dma_configured = true;
                }
                hipLaunchKernelGGL((attn_bwd_dma_kernel<D>), dim3(grid.x, grid.y, 2), dim3(256), dma_bytes + 8 * L, s, qkv, (const bf16_t*)o, d_o,
                                   (const float*)lse, delta, dqkv, L, scale);
                return launch_status("attention");
            }
        }
        hipLaunchKernelGGL((attn_bwd_kernel<D>), dim3(grid.x, grid.y, 2), dim3(256), bytes + 4 * L, s, qkv, (const bf16_t*)o, d_o,
                           (const float*)lse, delta, dqkv, L, scale);
    }
    return launch_status("attention");
}
static int attn_dispatch(int which, int C, const bf16_t* qkv, bf16_t* o, const bf16_t* d_o, float* lse, float* delta, bf16_t* dqkv,
                         int N, int L, float scale, hipStream_t s) {
    switch (C) {
        case 32: return attn_launch<32>(which, qkv, o, d_o, lse, delta, dqkv, N, L, scale, s);
        case 64: return attn_launch<64>(which, qkv, o, d_o, lse, delta, dqkv, N, L, scale, s);
        case 128: return attn_launch<128>(which, qkv, o, d_o, lse, delta, dqkv, N, L, scale, s);
        case 256: return attn_launch<256>(which, qkv, o, d_o, lse, delta, dqkv, N, L, scale, s);
    }
    set_error("attention: head width C=%d is not one of 32, 64, 128, 256", C);
    return -1;
}


// ------------------------------------------------------------------------------------------------ many small heads
// diffusers' attention blocks (UNet2DModel: attention_head_dim = 8 -> C / 8 heads of width 8) on separate q, k, v, o
// tensors [N][L][C], head h = channels [h D, (h + 1) D).  A width-8 head wastes 3/4 of a 16x16x32 MFMA step and L is
// small (<= 256) where these blocks sit, so this is a VALU kernel: one workgroup per (image, head, 256-query block), a
// thread per query (forward, dQ) or per key (dK / dV), the other side staged through LDS in chunks of 256 rows, online
// softmax in the forward, P recomputed from the saved log-sum-exp in the backward.  Any L, D in {8, 16, 32}.
template <typename T, int D>
__device__ __forceinline__ void ld_row(const T* p, float (&r)[D]) {
#pragma unroll
    for (int e = 0; e < D; e += Elem<T>::VEC == 8 ? 8 : 4) {
        if (Elem<T>::VEC == 8) { float8 v = load8(p + e); r[e] = v.lo.x; r[e + 1] = v.lo.y; r[e + 2] = v.lo.z; r[e + 3] = v.lo.w; r[e + 4] = v.hi.x; r[e + 5] = v.hi.y; r[e + 6] = v.hi.z; r[e + 7] = v.hi.w; }
        else { float4 v = load4(p + e); r[e] = v.x; r[e + 1] = v.y; r[e + 2] = v.z; r[e + 3] = v.w; }
    }
}
template <typename T, int D>
__device__ __forceinline__ void st_row(T* p, const float (&r)[D]) {
#pragma unroll
    for (int e = 0; e < D; e += 4) store4(p + e, make_float4(r[e], r[e + 1], r[e + 2], r[e + 3]));
}

// which: 0 forward (o, lse), 1 dQ (+ delta), 2 dK and dV
template <typename T, int D, int WHICH>
__global__ __launch_bounds__(256) void attn_mh_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                      T* __restrict__ o, const T* __restrict__ d_o, float* __restrict__ lse,
                                                      float* __restrict__ delta, T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv,
                                                      int L, int C, int H, float scale) {
    __shared__ float sa[256 * D], sb[256 * D], sl[256], sd[256];
    const int t = threadIdx.x, nh = blockIdx.y, n = nh / H, h = nh - n * H;
    const int row = blockIdx.x * 256 + t;
    const bool ok = row < L;
    const int64_t base = (int64_t)n * L * C + h * D;          // + row * C
    const int64_t sbase = ((int64_t)n * H + h) * L;           // lse / delta: [N][H][L]
    const int64_t ro = base + (int64_t)(ok ? row : 0) * C;
    float a[D], g[D], acc0[D], acc1[D];
#pragma unroll
    for (int e = 0; e < D; ++e) { a[e] = 0.f; g[e] = 0.f; acc0[e] = 0.f; acc1[e] = 0.f; }
    float m = NEG_BIG, lsum = 0.f, ls = 0.f, dl = 0.f;
    if (WHICH == 0) ld_row<T, D>(q + ro, a);
    if (WHICH == 1) {
        float ov[D];
        ld_row<T, D>(q + ro, a); ld_row<T, D>(d_o + ro, g); ld_row<T, D>(o + ro, ov);
#pragma unroll
        for (int e = 0; e < D; ++e) dl = fmaf(g[e], ov[e], dl);
        ls = lse[sbase + (ok ? row : 0)];
        if (ok) delta[sbase + row] = dl;
    }
    if (WHICH == 2) { ld_row<T, D>(k + ro, a); ld_row<T, D>(v + ro, g); }
    for (int c0 = 0; c0 < L; c0 += 256) {
        const int cn = min(256, L - c0);
        __syncthreads();
        if (t < cn) {
            const int64_t co = base + (int64_t)(c0 + t) * C;
            float r0[D], r1[D];
            if (WHICH == 2) { ld_row<T, D>(q + co, r0); ld_row<T, D>(d_o + co, r1); sl[t] = lse[sbase + c0 + t]; sd[t] = delta[sbase + c0 + t]; }
            else { ld_row<T, D>(k + co, r0); ld_row<T, D>(v + co, r1); }
#pragma unroll
            for (int e = 0; e < D; ++e) { sa[t * D + e] = r0[e]; sb[t * D + e] = r1[e]; }
        }
        __syncthreads();
        for (int j = 0; j < cn; ++j) {
            const float* pa = sa + j * D;       // forward / dQ: key j and its value; dK/dV: query j and its dO
            const float* pb = sb + j * D;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < D; ++e) s = fmaf(a[e], pa[e], s);
            s *= scale;
            if (WHICH == 0) {
                const float mn = fmaxf(m, s), al = __expf(m - mn), p = __expf(s - mn);
                lsum = lsum * al + p; m = mn;
#pragma unroll
                for (int e = 0; e < D; ++e) acc0[e] = fmaf(p, pb[e], acc0[e] * al);
            } else if (WHICH == 1) {
                const float p = __expf(s - ls);
                float dp = 0.f;
#pragma unroll
                for (int e = 0; e < D; ++e) dp = fmaf(g[e], pb[e], dp);
                const float ds = p * (dp - dl) * scale;
#pragma unroll
                for (int e = 0; e < D; ++e) acc0[e] = fmaf(ds, pa[e], acc0[e]);
            } else {
                const float p = __expf(s - sl[j]);      // a = this key, pa = query j: s is the same score
                float dp = 0.f;
#pragma unroll
                for (int e = 0; e < D; ++e) dp = fmaf(g[e], pb[e], dp);      // v . dO_j
                const float ds = p * (dp - sd[j]) * scale;
#pragma unroll
                for (int e = 0; e < D; ++e) { acc0[e] = fmaf(ds, pa[e], acc0[e]); acc1[e] = fmaf(p, pb[e], acc1[e]); }
            }
        }
    }
    if (!ok) return;
    if (WHICH == 0) {
        const float inv = 1.f / lsum;
#pragma unroll
        for (int e = 0; e < D; ++e) acc0[e] *= inv;
        st_row<T, D>(o + ro, acc0);
        lse[sbase + row] = m + __logf(lsum);
    } else if (WHICH == 1) {
        st_row<T, D>(dq + ro, acc0);
    } else {
        st_row<T, D>(dk + ro, acc0);
        st_row<T, D>(dv + ro, acc1);
    }
}

template <typename T, int D>
static int attn_mh_launch(int which, const void* q, const void* k, const void* v, void* o, const void* d_o, float* lse, float* delta,
                          void* dq, void* dk, void* dv, int N, int L, int C, int H, float scale, hipStream_t s) {
    dim3 grid((unsigned)cdiv(L, 256), (unsigned)(N * H));
#define MDM_MH(W) hipLaunchKernelGGL((attn_mh_kernel<T, D, W>), grid, dim3(256), 0, s, (const T*)q, (const T*)k, (const T*)v, (T*)o, \
                                     (const T*)d_o, lse, delta, (T*)dq, (T*)dk, (T*)dv, L, C, H, scale)
    if (which == 0) MDM_MH(0);
    else { MDM_MH(1); MDM_MH(2); }
#undef MDM_MH
    return launch_status("attention (multi-head)");
}
template <typename T>
static int attn_mh_dispatch(int which, int D, const void* q, const void* k, const void* v, void* o, const void* d_o, float* lse,
                            float* delta, void* dq, void* dk, void* dv, int N, int L, int C, int H, float scale, hipStream_t s) {
    switch (D) {
        case 8: return attn_mh_launch<T, 8>(which, q, k, v, o, d_o, lse, delta, dq, dk, dv, N, L, C, H, scale, s);
        case 16: return attn_mh_launch<T, 16>(which, q, k, v, o, d_o, lse, delta, dq, dk, dv, N, L, C, H, scale, s);
        case 32: return attn_mh_launch<T, 32>(which, q, k, v, o, d_o, lse, delta, dq, dk, dv, N, L, C, H, scale, s);
    }
    set_error("attention (multi-head): head width %d is not one of 8, 16, 32", D);
    return -1;
}

// ----------------------------------------------------------------------------
// Exact-fp32 attention FORWARD for the short sequences of the fp32 path (L <= 64: the 8x8 and 4x4 attention blocks of unet6 in the reverse
// sampler and in fp32 training): o = softmax(q k^T * scale) v on v_mfma_f32_16x16x4_f32, one workgroup per image, one launch instead
// of two batched contractions and a softmax launch (15 + 6.5 + 13 us per block at sample_num = 100).  The probabilities ARE written
// (S[N][L][L], 16 KB per image at L = 64): the fp32 backward of the unfused path reads them.
//   Orientation as in the bf16 kernels: S^T[key][query] = K Q^T -- wave w owns queries 16 w .. 16 w + 15 (the MFMA column, lane & 15) and
//   all keys, so a query's softmax is lane-local plus two __shfl_xor over the four lane groups; the accumulator of that product
//   (lane group g, register r = key 16 kt + 4 g + r) IS the B operand of o^T[c][query] = V^T P^T with the same "a lane's four
//   consecutive k feed four MFMAs" convention as gemm_f32_mfma_kernel; V is staged once per image in LDS ([L][C + 4]: the four
//   lane groups read four different rows, the pad puts them 16 banks apart) and read as scalars (A[i = c][k = key] is strided in memory).
//   Fragments of Q (kept in registers: C / 16 float4 per lane) and K come straight from global memory.
template <int C>
__global__ __launch_bounds__(256) void attn_f32_small_kernel(const float* __restrict__ qkv, float* __restrict__ o, float* __restrict__ S,
                                                             int L, float scale) {
    constexpr int NS = C / 16, VP = C + 4;
    extern __shared__ __attribute__((aligned(16))) float vs[];             // [L][VP]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, g = lane >> 4;
    const int n = blockIdx.x, KT = L >> 4;
    const float* base = qkv + (int64_t)n * L * 3 * C;
    for (int idx = t; idx < L * (C / 4); idx += 256) {                       // V -> LDS, coalesced 16-byte loads
        const int row = idx / (C / 4), c4 = idx - row * (C / 4);
        *reinterpret_cast<float4*>(vs + row * VP + 4 * c4) = *reinterpret_cast<const float4*>(base + (int64_t)row * 3 * C + 2 * C + 4 * c4);
    }
    __syncthreads();
    if (wave >= KT) return;                                                  // (L = 16: one query tile; no barrier behind this point)
    const int query = 16 * wave + j;
    float4 qf[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) qf[s2] = *reinterpret_cast<const float4*>(base + (int64_t)query * 3 * C + 16 * s2 + 4 * g);
    f32x4 st[4];                                                             // S^T tiles: keys 16 kt + 4 g + r, this lane's query
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kt < KT) {
            const float* kr = base + (int64_t)(16 * kt + j) * 3 * C + C + 4 * g;      // A[i = key][k = c]: this lane's key row
            f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f};             // two accumulation chains (even / odd k-steps): an MFMA waits for the one it adds to
#pragma unroll
            for (int s2 = 0; s2 < NS; s2 += 2) {
                const float4 kf = *reinterpret_cast<const float4*>(kr + 16 * s2), kg = *reinterpret_cast<const float4*>(kr + 16 * s2 + 16);
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[s2].x, st[kt], 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kg.x, qf[s2 + 1].x, s1, 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[s2].y, st[kt], 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kg.y, qf[s2 + 1].y, s1, 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[s2].z, st[kt], 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kg.z, qf[s2 + 1].z, s1, 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[s2].w, st[kt], 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kg.w, qf[s2 + 1].w, s1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kt][r] += s1[r];
        }
    }
    // softmax over the keys of this lane's query: 4 registers x KT tiles here, the other lane groups through two shuffles
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
        if (kt < KT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[kt][r] *= scale; mx = fmaxf(mx, st[kt][r]); }
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
        if (kt < KT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[kt][r] = __expf(st[kt][r] - mx); sum += st[kt][r]; }
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float* Srow = S + ((int64_t)n * L + query) * L;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
        if (kt < KT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kt][r] *= inv;
            *reinterpret_cast<float4*>(Srow + 16 * kt + 4 * g) = make_float4(st[kt][0], st[kt][1], st[kt][2], st[kt][3]);
        }
    // o^T[c][query] = sum_key V^T[c][key] P^T[key][query]
    float* orow = o + ((int64_t)n * L + query) * C;
#pragma unroll 4
    for (int ct = 0; ct < NS; ++ct) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};       // two chains: key tiles 0, 2 / 1, 3
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            if (kt < KT) {
                const float* vp = vs + (16 * kt + 4 * g) * VP + 16 * ct + j;           // A[i = c = 16 ct + j][k = key 16 kt + 4 g + m]
                f32x4& a = (kt & 1) ? acc1 : acc;
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[0], st[kt][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[VP], st[kt][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[2 * VP], st[kt][2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[3 * VP], st[kt][3], a, 0, 0, 0);
            }
        *reinterpret_cast<float4*>(orow + 16 * ct + 4 * g) = make_float4(acc[0] + acc1[0], acc[1] + acc1[1], acc[2] + acc1[2], acc[3] + acc1[3]);   // c = 16 ct + 4 g + r
    }
}

}  // namespace mdm
using namespace mdm;

extern "C" int mdm_attn_f32_small_supported(int L, int C) {
    return (L == 16 || L == 32 || L == 48 || L == 64) && (C == 64 || C == 128 || C == 256) ? 1 : 0;
}
extern "C" int mdm_attn_f32_small_fwd(const float* qkv, float* o, float* S, int N, int L, int C, float scale, void* stream) {
    MDM_REQUIRE(mdm_attn_f32_small_supported(L, C), "attn_f32_small_fwd: unsupported L %d / C %d (use the contraction path)", L, C);
    MDM_REQUIRE(qkv && o && S && N > 0, "attn_f32_small_fwd: bad arguments");
    const int bytes = L * (C + 4) * 4;
    hipStream_t s = pick_stream(stream);
#define MDM_ATTN_F32(CC)                                                                                                               \
    {                                                                                                                                  \
        static int configured = 0;                                                                                                     \
        if (configured < bytes) {                                                                                                      \
            MDM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_small_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)); \
            configured = bytes;                                                                                                        \
        }                                                                                                                              \
        hipLaunchKernelGGL((attn_f32_small_kernel<CC>), dim3((unsigned)N), dim3(256), bytes, s, qkv, o, S, L, scale);                  \
    }
    if (C == 64) MDM_ATTN_F32(64) else if (C == 128) MDM_ATTN_F32(128) else MDM_ATTN_F32(256)
#undef MDM_ATTN_F32
    return launch_status("attn_f32_small_fwd");
}

extern "C" int mdm_attn_supported(int dtype, int L, int C) {
    return dtype == MDM_BF16 && L > 0 && L % 16 == 0 && (C == 32 || C == 64 || C == 128 || C == 256) ? 1 : 0;
}
extern "C" int mdm_attn_fwd(int dtype, const void* qkv, void* o, float* lse, int N, int L, int C, float scale, void* stream) {
    MDM_REQUIRE(mdm_attn_supported(dtype, L, C), "attn_fwd: unsupported dtype %d / L %d / C %d (use the contraction path)", dtype, L, C);
    MDM_REQUIRE(qkv && o && lse && N > 0, "attn_fwd: bad arguments");
    return attn_dispatch(0, C, (const bf16_t*)qkv, (bf16_t*)o, nullptr, lse, nullptr, nullptr, N, L, scale, pick_stream(stream));
}
extern "C" int mdm_attn_bwd(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                            int N, int L, int C, float scale, void* stream) {
    MDM_REQUIRE(mdm_attn_supported(dtype, L, C), "attn_bwd: unsupported dtype %d / L %d / C %d (use the contraction path)", dtype, L, C);
    MDM_REQUIRE(qkv && o && d_o && lse && delta && dqkv && N > 0, "attn_bwd: bad arguments");
    return attn_dispatch(1, C, (const bf16_t*)qkv, (bf16_t*)const_cast<void*>(o), (const bf16_t*)d_o, const_cast<float*>(lse), delta,
                         (bf16_t*)dqkv, N, L, scale, pick_stream(stream));
}

extern "C" int mdm_attn_mh_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int N, int L, int C,
                               int heads, float scale, void* stream) {
    MDM_REQUIRE(q && k && v && o && lse && N > 0 && L > 0 && heads > 0 && C % heads == 0, "attn_mh_fwd: bad arguments");
    hipStream_t s = pick_stream(stream);
    if (dtype == MDM_BF16) return attn_mh_dispatch<bf16_t>(0, C / heads, q, k, v, o, nullptr, lse, nullptr, nullptr, nullptr, nullptr, N, L, C, heads, scale, s);
    if (dtype == MDM_F32) return attn_mh_dispatch<float>(0, C / heads, q, k, v, o, nullptr, lse, nullptr, nullptr, nullptr, nullptr, N, L, C, heads, scale, s);
    set_error("attn_mh_fwd: bad dtype %d", dtype);
    return -1;
}
extern "C" int mdm_attn_mh_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                               float* delta, void* dq, void* dk, void* dv, int N, int L, int C, int heads, float scale, void* stream) {
    MDM_REQUIRE(q && k && v && o && d_o && lse && delta && dq && dk && dv && N > 0 && L > 0 && heads > 0 && C % heads == 0, "attn_mh_bwd: bad arguments");
    hipStream_t s = pick_stream(stream);
    if (dtype == MDM_BF16) return attn_mh_dispatch<bf16_t>(1, C / heads, q, k, v, const_cast<void*>(o), d_o, const_cast<float*>(lse), delta, dq, dk, dv, N, L, C, heads, scale, s);
    if (dtype == MDM_F32) return attn_mh_dispatch<float>(1, C / heads, q, k, v, const_cast<void*>(o), d_o, const_cast<float*>(lse), delta, dq, dk, dv, N, L, C, heads, scale, s);
    set_error("attn_mh_bwd: bad dtype %d", dtype);
    return -1;
}
