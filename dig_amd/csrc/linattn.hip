// LinearAttention context (ddp.py:225,229) on MFMA, gfx950.
//   ctxT[e][d] = scale * sum_n v[e][n] * softmax_n(k)[d][n]
// k and v are channel-major [32][n] slabs per (image, head) - produced by running the qkv projection as
// W*x^T - so both MFMA operands are K(=n)-contiguous and are loaded straight from HBM into fragments:
// each lane reads 4 x 16 B = 32 consecutive n of its row per 64-n chunk (the k-slot order inside an MFMA is
// arbitrary as long as A and B agree).  exp(k - kmax) is fused into the B operand; the softmax denominator
// is accumulated alongside and divided out in the finalize kernel, which also sums the n-splits in a fixed
// order (deterministic, no atomics).
#include "common.hpp"

namespace {

template <typename T> __global__ __launch_bounds__(256) void la_context_kernel(dig_la_context_args a) {
    __shared__ float red[4][1024];
    __shared__ float ksum_red[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int split = blockIdx.x, bh = blockIdx.y;
    const int b = bh / a.heads, h = bh % a.heads;
    const T* krow = (const T*)a.k + (size_t)b * a.b_stride_k + (size_t)h * a.h_stride_k + (size_t)r * a.n;
    const T* vrow = (const T*)a.v + (size_t)b * a.b_stride_v + (size_t)h * a.h_stride_v + (size_t)r * a.n;
    const float kmax = a.kmax[(size_t)bh * 32 + r];
    // n range of this block, in 64-element chunks dealt round-robin to the 4 waves
    const int chunks = (a.n + 63) / 64;
    const int cps = (chunks + a.splits - 1) / a.splits;
    const int c_begin = split * cps, c_end = min(chunks, c_begin + cps);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    float ks = 0.f;
    for (int c = c_begin + wave; c < c_end; c += 4) {
        const int nb = c * 64 + hf * 32;
        Frag<T> fv[4], fk[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int n = nb + s * 8;
            if (n < a.n) {  // n % 8 == 0 is guaranteed by the host, so a vector is all-in or all-out
                frag_load<T>(fv[s], vrow + n);
                frag_load<T>(fk[s], krow + n);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float ev = __expf(frag_get<T>(fk[s], j) - kmax);
                    frag_set<T>(fk[s], j, ev);
                    ks += frag_get<T>(fk[s], j);  // sum what the MFMA will actually see (rounded to T)
                }
            } else {
                frag_zero<T>(fv[s]);
                frag_zero<T>(fk[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) mma16<T>(acc, fv[s], fk[s]);
    }
    ks += __shfl_xor(ks, 32, 64);
#pragma unroll
    for (int e = 0; e < 16; e++) red[wave][acc_row(e, lane) * 32 + r] = acc[e];
    if (hf == 0) ksum_red[wave][r] = ks;
    __syncthreads();
    float* part = a.part + ((size_t)bh * a.splits + split) * 1024;
    for (int i = tid; i < 1024; i += 256) part[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (tid < 32) a.ksum[((size_t)bh * a.splits + split) * 32 + tid] = (ksum_red[0][tid] + ksum_red[1][tid]) + (ksum_red[2][tid] + ksum_red[3][tid]);
}

template <typename T> __global__ void la_finalize_kernel(dig_la_finalize_args a) {
    const int bh = blockIdx.x;
    T* out = (T*)a.ctxT + (size_t)bh * 1024;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        const int d = i & 31;
        float s = 0.f, ks = 0.f;
        for (int sp = 0; sp < a.splits; sp++) {
            s += a.part[((size_t)bh * a.splits + sp) * 1024 + i];
            ks += a.ksum[((size_t)bh * a.splits + sp) * 32 + d];
        }
        out[a.transposed ? (d * 32 + (i >> 5)) : i] = from_f32<T>(a.scale * s / ks);
    }
}

}  // namespace

extern "C" int dig_la_context(const dig_la_context_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->k && a->v && a->kmax && a->part && a->ksum, DIG_EINVAL, "la_context: null pointer");
    DIG_CHECK(a->B > 0 && a->heads > 0 && a->n > 0 && a->n % 8 == 0 && a->splits > 0, DIG_EINVAL, "la_context: bad shape (n must be a multiple of 8)");
    DIG_CHECK(a->B * a->heads <= 65535, DIG_EINVAL, "la_context: too many (image, head) pairs");
    const dim3 grid(a->splits, a->B * a->heads);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(la_context_kernel<bf16>, grid, dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(la_context_kernel<float>, grid, dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("la_context");
    return DIG_OK;
}

extern "C" int dig_la_finalize(const dig_la_finalize_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->part && a->ksum && a->ctxT && a->BH > 0 && a->splits > 0, DIG_EINVAL, "la_finalize: bad argument");
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(la_finalize_kernel<bf16>, dim3(a->BH), dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(la_finalize_kernel<float>, dim3(a->BH), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("la_finalize");
    return DIG_OK;
}
