// Evaluation-side kernels (the step after the sampler): per-image min-max normalisation, unit-length rows and the
// column arg-max of a similarity matrix.  The similarity matrix itself is one fp32 contraction (mdm_gemm, layout 0).
//
// Replaces reference utils/datautils.py:211-222 (normalize01), tester.py:140-145 / sampler.py:520-526
// (_compute_similarity: cosine similarity of every source image with every data image) and the
// `score.max(dim=0)` of tester.py:185-201 / sampler.py:487-518 (nearest neighbour).
#include "common.h"

namespace mdm {

// one workgroup per image: y = (x - min) / (max - min), NaN (constant image) -> 0
__global__ __launch_bounds__(256) void normalize01_kernel(const float* x, float* y, int E) {
    const int img = blockIdx.x, t = threadIdx.x;
    const float* xi = x + (int64_t)img * E;
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int i = t; i < E; i += 256) { const float v = xi[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    hi = wave_max(hi);
    lo = -wave_max(-lo);
    __shared__ float s_lo[4], s_hi[4];
    if ((t & 63) == 0) { s_lo[t >> 6] = lo; s_hi[t >> 6] = hi; }
    __syncthreads();
    lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    const float range = hi - lo;
    for (int i = t; i < E; i += 256) {
        const float v = (xi[i] - lo) / range;
        y[(int64_t)img * E + i] = (v != v) ? 0.f : v;
    }
}

// one workgroup per row: y = x / max(||x||, eps)   (torch.nn.functional.cosine_similarity's normalisation)
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* x, float* y, int D, float eps) {
    const int r = blockIdx.x, t = threadIdx.x;
    const float* xr = x + (int64_t)r * D;
    float a = 0.f;
    for (int i = t; i < D; i += 256) a = fmaf(xr[i], xr[i], a);
    a = wave_sum(a);
    __shared__ float part[4];
    if ((t & 63) == 0) part[t >> 6] = a;
    __syncthreads();
    const float inv = 1.f / fmaxf(sqrtf((part[0] + part[1]) + (part[2] + part[3])), eps);
    for (int i = t; i < D; i += 256) y[(int64_t)r * D + i] = xr[i] * inv;
}

// S[M][B] -> for every column b: the largest value and the FIRST row that holds it (torch.max(dim=0) tie rule)
__global__ __launch_bounds__(256) void col_argmax_kernel(const float* S, int M, int B, float* val, int64_t* idx) {
    const int b = blockIdx.x, t = threadIdx.x;
    float best = -3.0e38f;
    int bi = 0x7fffffff;
    for (int m = t; m < M; m += 256) {
        const float v = S[(int64_t)m * B + b];
        if (v > best || (v == best && m < bi)) { best = v; bi = m; }
    }
    __shared__ float sv[256];
    __shared__ int si[256];
    sv[t] = best; si[t] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            const float v = sv[t + o];
            const int i = si[t + o];
            if (v > sv[t] || (v == sv[t] && i < si[t])) { sv[t] = v; si[t] = i; }
        }
        __syncthreads();
    }
    if (t == 0) { if (val) val[b] = sv[0]; idx[b] = si[0]; }
}

}  // namespace mdm
using namespace mdm;

extern "C" int mdm_normalize01(const float* x, float* y, int N, int E, void* stream) {
    MDM_REQUIRE(x && y && N > 0 && E > 0, "normalize01: bad arguments");
    hipLaunchKernelGGL(normalize01_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, y, E);
    return launch_status("normalize01");
}
extern "C" int mdm_unit_rows(const float* x, float* y, int R, int D, float eps, void* stream) {
    MDM_REQUIRE(x && y && R > 0 && D > 0, "unit_rows: bad arguments");
    hipLaunchKernelGGL(unit_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, y, D, eps);
    return launch_status("unit_rows");
}
extern "C" int mdm_col_argmax(const float* S, int M, int B, float* val, int64_t* idx, void* stream) {
    MDM_REQUIRE(S && idx && M > 0 && B > 0, "col_argmax: bad arguments");
    hipLaunchKernelGGL(col_argmax_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, S, M, B, val, idx);
    return launch_status("col_argmax");
}
