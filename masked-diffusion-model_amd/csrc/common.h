// Shared device/host helpers for libmdm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mdm_hip.h"

namespace mdm {

// ------------------------------------------------------------------ errors
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define MDM_CHECK_HIP(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ::mdm::hip_fail(_e, #expr); \
    } while (0)

#define MDM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::mdm::set_error(__VA_ARGS__);     \
            return -1;                         \
        }                                      \
    } while (0)

inline hipStream_t pick_stream(void* stream) { return reinterpret_cast<hipStream_t>(stream); }

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return 0;
}

// ------------------------------------------------------------------ small integer division
// x / d through a float reciprocal: (x + 0.5) * rcp(d) truncated is EXACT for 0 <= x < 2^20 and 1 <= d <= 2^10 (the 1-ulp error
// of v_rcp_f32 and of the product moves the value by less than 0.5 / d).  Three VALU instructions per quotient instead of the
// ~35 of an integer division: the index arithmetic of the GroupNorm kernels (16 group indices per thread) and the prologues of
// the convolution kernels were a measurable part of their few microseconds.
__device__ __forceinline__ float rcp_small(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ int div_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// ------------------------------------------------------------------ bf16
typedef unsigned short bf16_t;   // raw bits

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-nearest-even, NaN stays NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;                        // elements per 16-byte vector
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 4 consecutive elements <-> float4
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    uint2 r;
    r.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    r.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = r;
}

// 8 consecutive elements (two float4)
struct float8 { float4 lo, hi; };
__device__ __forceinline__ float8 load8(const float* p) { return {load4(p), load4(p + 4)}; }
__device__ __forceinline__ float8 load8(const bf16_t* p) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    float8 o;
    o.lo = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    o.hi = make_float4(__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
                       __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u));
    return o;
}
__device__ __forceinline__ void store8(float* p, const float8& v) { store4(p, v.lo); store4(p + 4, v.hi); }
__device__ __forceinline__ void store8(bf16_t* p, const float8& v) {
    uint4 r;
    r.x = (uint32_t)f2bf(v.lo.x) | ((uint32_t)f2bf(v.lo.y) << 16);
    r.y = (uint32_t)f2bf(v.lo.z) | ((uint32_t)f2bf(v.lo.w) << 16);
    r.z = (uint32_t)f2bf(v.hi.x) | ((uint32_t)f2bf(v.hi.y) << 16);
    r.w = (uint32_t)f2bf(v.hi.z) | ((uint32_t)f2bf(v.hi.w) << 16);
    *reinterpret_cast<uint4*>(p) = r;
}

// ------------------------------------------------------------------ reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: these sit in the VALU-bound GroupNorm loops
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}

// ------------------------------------------------------------------ Philox4x32-10
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    static __device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
        uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    __device__ uint4 operator()(uint64_t ctr_lo, uint64_t ctr_hi) const {
        uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t h0, l0, h1, l1;
            mulhilo(0xD2511F53u, c0, h0, l0);
            mulhilo(0xCD9E8D57u, c2, h1, l1);
            uint32_t n0 = h1 ^ c1 ^ a, n1 = l1, n2 = h0 ^ c3 ^ b, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
// [0,1) with 24 bits, like torch's float uniform
__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
    float u1 = ((a >> 8) + 1) * (1.0f / 16777216.0f);   // (0,1]
    float u2 = u01(b);
    float r = sqrtf(-2.f * __logf(u1));
    float s, c;
    __sincosf(6.28318530717958647692f * u2, &s, &c);
    return make_float2(r * c, r * s);
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace mdm
