// Shared device helpers for the DiG MI355X (gfx950 / CDNA4) kernels.
// Wave = 64 lanes.  All MFMA kernels are templated on the storage type T in {bf16, float};
// accumulation, statistics and softmax are always fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>

#include "../../include/dig_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define DIG_WAVE 64

// ---- last-error plumbing (host) ---------------------------------------------------------
void dig_set_error(const char* fmt, ...);
#define DIG_CHECK(cond, code, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            dig_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)
#define DIG_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            dig_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return DIG_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- an 8-element K-fragment: what one lane feeds to a 16-deep MFMA step ------------------
// bf16: 8 x bf16 = 16 B, consumed by ONE v_mfma_f32_32x32x16_bf16 (lane l: row/col l&31, k = 8*(l>>5)+j).
// f32 : 8 x f32  = 32 B, consumed by EIGHT v_mfma_f32_32x32x2_f32: step j multiplies element j of
//       both halves (the k-slot assignment is arbitrary as long as A and B agree), exact fp32.
template <typename T> struct Frag;
template <> struct Frag<bf16> {
    bf16x8 v;
};
template <> struct Frag<float> {
    float v[8];
};

template <typename T> __device__ __forceinline__ void frag_zero(Frag<T>& f);
template <> __device__ __forceinline__ void frag_zero<bf16>(Frag<bf16>& f) {
    f.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
}
template <> __device__ __forceinline__ void frag_zero<float>(Frag<float>& f) {
#pragma unroll
    for (int i = 0; i < 8; i++) f.v[i] = 0.f;
}

// load 8 contiguous elements (16-B aligned for bf16, 16-B aligned halves for f32) from LDS or global
template <typename T> __device__ __forceinline__ void frag_load(Frag<T>& f, const T* p);
template <> __device__ __forceinline__ void frag_load<bf16>(Frag<bf16>& f, const bf16* p) {
    f.v = *reinterpret_cast<const bf16x8*>(p);
}
template <> __device__ __forceinline__ void frag_load<float>(Frag<float>& f, const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
template <typename T> __device__ __forceinline__ void frag_store(const Frag<T>& f, T* p);
template <> __device__ __forceinline__ void frag_store<bf16>(const Frag<bf16>& f, bf16* p) {
    *reinterpret_cast<bf16x8*>(p) = f.v;
}
template <> __device__ __forceinline__ void frag_store<float>(const Frag<float>& f, float* p) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f.v[0], f.v[1], f.v[2], f.v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f.v[4], f.v[5], f.v[6], f.v[7]};
}
template <typename T> __device__ __forceinline__ float frag_get(const Frag<T>& f, int i);
template <> __device__ __forceinline__ float frag_get<bf16>(const Frag<bf16>& f, int i) { return (float)f.v[i]; }
template <> __device__ __forceinline__ float frag_get<float>(const Frag<float>& f, int i) { return f.v[i]; }
template <typename T> __device__ __forceinline__ void frag_set(Frag<T>& f, int i, float x);
template <> __device__ __forceinline__ void frag_set<bf16>(Frag<bf16>& f, int i, float x) { f.v[i] = (bf16)x; }
template <> __device__ __forceinline__ void frag_set<float>(Frag<float>& f, int i, float x) { f.v[i] = x; }

// D(32x32) += A(32x16) * B(16x32): lane l supplies A[row l&31][k-slots of half l>>5] and
// B[k-slots][col l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
template <typename T> __device__ __forceinline__ void mma16(f32x16& acc, const Frag<T>& a, const Frag<T>& b);
template <> __device__ __forceinline__ void mma16<bf16>(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- scalar conversions -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// q_sample / predict_start_from_v element (ddp.py:857-860, 630-634): two rounded products and a rounded sum, exactly the
// reference's tensor expression.  The files are built with -ffp-contract=fast, under which the backend fuses a product
// into the sum wherever it likes (a `#pragma clang fp contract(off)` does not stop it): every inlined copy could fuse a
// different product, and a halo pixel computed by one tile would round differently from the same pixel computed by the
// tile that owns it.  The empty asm statements make both products opaque, so nothing can be fused.
__device__ __forceinline__ float axpby_rn(float ca, float x, float cs, float y) {
    float p = ca * x;
    float q = cs * y;
    asm volatile("" : "+v"(p), "+v"(q));
    return p + q;
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp): no IEEE division sequence in the conv loader's prologue
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// SiLU of an affine, with the affine PRE-SCALED by -log2(e): for (ap, bp) = -log2e * (a, b) and u = ap x + bp = -log2e v,
//   silu(v) = v / (1 + 2^u) = u / (-log2e (1 + 2^u)) = u * rcp(fma(2^u, -log2e, -log2e))
// - one fma, v_exp, one fma, v_rcp, one mul: a full-rate instruction less per element than silu_f(a x + b).
constexpr float SILU_PRE = -1.4426950408889634f;
__device__ __forceinline__ float silu_pre(float ap, float bp, float x) {
    const float u = fmaf(ap, x, bp);
    return u * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(u), SILU_PRE, SILU_PRE));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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
// block-wide reductions (blockDim.x multiple of 64, <= 1024); `red` = LDS scratch of >= 32 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; i++) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; i++) t = fmaxf(t, red[i]);
    return t;
}
