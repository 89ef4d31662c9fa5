// Fused LinearAttention (ddp.py:196-233) for gfx950: the input tensor is read twice and the output written once;
// q, k, v, the softmaxes, the context and the to_out projection never touch HBM.
//
//   pass A  la_kv_ctx : per (image, pixel split) and head (one wave per head):
//                       k,v = W' x (RMSNorm folded: g into W', 1/||x|| as a row scale) on MFMA with pixels as ROWS of
//                       the accumulator, online softmax of k over pixels (running max per column d), and
//                       ctxT[e][d] += v^T kexp  where BOTH operands are the just-computed accumulators re-used
//                       as MFMA operands in registers (CDNA "accumulator tile as the next MFMA's operand": the
//                       contraction runs over the accumulator ROW index, so no lane movement and no LDS).
//   merge   la_merge  : log-sum-exp merge of the pixel splits -> ctx[h][d][e]; then M_b = Wout . blockdiag(ctx^T)
//                       (tiny GEMM, gemm.hip) folds the second einsum and the to_out conv into one [dim][128] matrix.
//   pass B  la_out    : per 32-pixel sub-tile (one wave): q = Wq' x with pixels as COLUMNS, softmax over d in
//                       registers, y = M_b qs again taking the accumulator as the B operand, + bias, RMSNorm over
//                       channels, * g2, + x (residual), written back through the wave's own rows of the LDS tile.
// Storage type T in {bf16, float}; all statistics fp32.
#include "common.hpp"

namespace {

template <typename T> __device__ __forceinline__ void acc_to_frag(Frag<T>& f, const f32x16& a, int s) {
#pragma unroll
    for (int j = 0; j < 8; j++) frag_set<T>(f, j, a[8 * s + j]);
}
// k index (row of the accumulator tile) that element j of lane-half hf carries in k-step s of an accumulator-derived operand
__device__ __forceinline__ int acc_k(int s, int j, int hf) { return 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3); }

template <typename T> __device__ __forceinline__ void load_4_4(Frag<T>& f, const T* lo, const T* hi);
template <> __device__ __forceinline__ void load_4_4<bf16>(Frag<bf16>& f, const bf16* lo, const bf16* hi) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(lo), b = *reinterpret_cast<const bf16x4*>(hi);
    f.v = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <> __device__ __forceinline__ void load_4_4<float>(Frag<float>& f, const float* lo, const float* hi) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(lo), b = *reinterpret_cast<const f32x4*>(hi);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}

// ------------------------------------------------------------------------------------------------------------
// pass A
template <typename T, int KS, bool WREG>
__global__ __launch_bounds__(256) void la_kv_ctx_kernel(dig_la_kvctx_args a) {
    constexpr int C = KS * 16, CP = C + 8, TP = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);                       // [TP][CP]
    float* winv = reinterpret_cast<float*>(xs + TP * CP);     // [4 waves][32]
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int split = blockIdx.x, b = blockIdx.y;
    const T* x = (const T*)a.x + (size_t)b * a.n * C;
    const T* wk = (const T*)a.wkv + (size_t)(h * 32 + r) * C;
    const T* wv = (const T*)a.wkv + (size_t)(a.hid + h * 32 + r) * C;
    const int tiles = (a.n + TP - 1) / TP, tps = (tiles + a.splits - 1) / a.splits;
    const int t_begin = split * tps, t_end = min(tiles, t_begin + tps);
    Frag<T> fk[WREG ? KS : 1], fv[WREG ? KS : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            frag_load<T>(fk[ks], wk + ks * 16 + hf * 8);
            frag_load<T>(fv[ks], wv + ks * 16 + hf * 8);
        }
    }
    f32x16 ctx;
#pragma unroll
    for (int e = 0; e < 16; e++) ctx[e] = 0.f;
    float m_run = -INFINITY, ksum = 0.f;
    const float sqrtC = sqrtf((float)C);
    for (int t = t_begin; t < t_end; t++) {
        const int p0 = t * TP;
        __syncthreads();
        for (int i = tid; i < TP * (C / 8); i += 256) {
            const int row = i / (C / 8), v = i - row * (C / 8);
            Frag<T> f;
            if (p0 + row < a.n) frag_load<T>(f, x + (size_t)(p0 + row) * C + v * 8);
            else frag_zero<T>(f);
            frag_store<T>(f, xs + row * CP + v * 8);
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < TP / 32; sub++) {
            f32x16 ak, av;
#pragma unroll
            for (int e = 0; e < 16; e++) ak[e] = av[e] = 0.f;
            float ssq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                Frag<T> fa;
                frag_load<T>(fa, xs + (sub * 32 + r) * CP + ks * 16 + hf * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float xv = frag_get<T>(fa, j);
                    ssq += xv * xv;
                }
                if constexpr (WREG) {
                    mma16<T>(ak, fa, fk[ks]);
                    mma16<T>(av, fa, fv[ks]);
                } else {
                    Frag<T> gk, gv;
                    frag_load<T>(gk, wk + ks * 16 + hf * 8);
                    frag_load<T>(gv, wv + ks * 16 + hf * 8);
                    mma16<T>(ak, fa, gk);
                    mma16<T>(av, fa, gv);
                }
            }
            ssq += __shfl_xor(ssq, 32, 64);
            if (hf == 0) winv[h * 32 + r] = sqrtC / fmaxf(sqrtf(ssq), 1e-12f);  // RMSNorm row scale of pixel r
            __builtin_amdgcn_wave_barrier();
            float mt = -INFINITY;
            float kv[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = acc_row(e, lane);
                const float inv = winv[h * 32 + row];
                const bool ok = p0 + sub * 32 + row < a.n;
                kv[e] = ok ? ak[e] * inv : -INFINITY;
                av[e] = ok ? av[e] * inv : 0.f;
                mt = fmaxf(mt, kv[e]);
            }
            __builtin_amdgcn_wave_barrier();
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            const float f = __expf(m_run - m_new);  // 0 on the first tile (m_run = -inf)
#pragma unroll
            for (int e = 0; e < 16; e++) ctx[e] *= f;
            ksum *= f;
            m_run = m_new;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                Frag<T> fA, fB;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    frag_set<T>(fB, j, __expf(kv[8 * s + j] - m_new));  // exp(-inf) = 0 for masked rows
                    ksum += frag_get<T>(fB, j);                         // what the MFMA sees (rounded to T)
                    frag_set<T>(fA, j, av[8 * s + j]);
                }
                mma16<T>(ctx, fA, fB);  // ctxT[e][d] += sum_p v[p][e] * kexp[p][d]
            }
        }
    }
    ksum += __shfl_xor(ksum, 32, 64);
    const size_t slot = ((size_t)b * a.splits + split) * 4 + h;
#pragma unroll
    for (int e = 0; e < 16; e++) a.part[slot * 1024 + acc_row(e, lane) * 32 + r] = ctx[e];
    if (hf == 0) {
        a.pmax[slot * 32 + r] = m_run;
        a.psum[slot * 32 + r] = ksum;
    }
}

// merge of the splits: ctx[b][h][d][e] = sum_s w_s part_s[e][d] / sum_s w_s ksum_s[d],  w_s = exp(m_s - max_s m_s)
template <typename T> __global__ void la_merge_kernel(dig_la_merge_args a) {
    const int bh = blockIdx.x, b = bh / 4, h = bh % 4;
    T* out = (T*)a.ctx + (size_t)bh * 1024;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        const int e = i >> 5, d = i & 31;
        float M = -INFINITY;
        for (int s = 0; s < a.splits; s++) M = fmaxf(M, a.pmax[(((size_t)b * a.splits + s) * 4 + h) * 32 + d]);
        float num = 0.f, den = 0.f;
        for (int s = 0; s < a.splits; s++) {
            const size_t slot = ((size_t)b * a.splits + s) * 4 + h;
            const float w = __expf(a.pmax[slot * 32 + d] - M);
            num += w * a.part[slot * 1024 + i];
            den += w * a.psum[slot * 32 + d];
        }
        out[d * 32 + e] = from_f32<T>(num / den);
    }
}

// ------------------------------------------------------------------------------------------------------------
// pass B
template <typename T, int KS, int DT>
__global__ __launch_bounds__(256) void la_out_kernel(dig_la_out_args a) {
    constexpr int C = KS * 16, CP = C + 8, TP = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);  // [TP][CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const long long P = (long long)a.B * a.n;
    const long long p0 = (long long)blockIdx.x * TP;
    const T* x = (const T*)a.x;
    for (int i = tid; i < TP * (C / 8); i += 256) {
        const int row = i / (C / 8), v = i - row * (C / 8);
        Frag<T> f;
        if (p0 + row < P) frag_load<T>(f, x + (size_t)(p0 + row) * C + v * 8);
        else frag_zero<T>(f);
        frag_store<T>(f, xs + row * CP + v * 8);
    }
    __syncthreads();
    const long long pw = p0 + wave * 32;  // first pixel of this wave's sub-tile (n % 32 == 0: one image per sub-tile)
    if (pw >= P) return;
    const int b = (int)(pw / a.n);
    T* xw = xs + wave * 32 * CP;
    // ---- q[n'][p] = Wq'[n'] . x[p]: head h = M tile h, pixel = column (lane)
    f32x16 aq[4];
#pragma unroll
    for (int hh = 0; hh < 4; hh++)
#pragma unroll
        for (int e = 0; e < 16; e++) aq[hh][e] = 0.f;
    float ssq = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        Frag<T> fb;
        frag_load<T>(fb, xw + r * CP + ks * 16 + hf * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float xv = frag_get<T>(fb, j);
            ssq += xv * xv;
        }
#pragma unroll
        for (int hh = 0; hh < 4; hh++) {
            Frag<T> fa;
            frag_load<T>(fa, (const T*)a.wq + (size_t)(hh * 32 + r) * C + ks * 16 + hf * 8);
            mma16<T>(aq[hh], fa, fb);
        }
    }
    ssq += __shfl_xor(ssq, 32, 64);
    const float inv = sqrtf((float)C) / fmaxf(sqrtf(ssq), 1e-12f);
    // ---- softmax over d (the 32 rows of each head tile: 16 registers x 2 lane halves), * scale; keep as B fragments
    Frag<T> qf[4][2];
#pragma unroll
    for (int hh = 0; hh < 4; hh++) {
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            aq[hh][e] *= inv;
            mx = fmaxf(mx, aq[hh][e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            aq[hh][e] = __expf(aq[hh][e] - mx);
            sm += aq[hh][e];
        }
        sm += __shfl_xor(sm, 32, 64);
        const float sc = a.scale / sm;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int j = 0; j < 8; j++) frag_set<T>(qf[hh][s], j, aq[hh][8 * s + j] * sc);
    }
    // ---- y[c][p] = sum_n' Mb[c][n'] qs[n'][p]  (+ bias), contraction over the accumulator rows of q
    const T* Mb = (const T*)a.mb + (size_t)b * a.dim * 128;
    f32x16 ay[DT];
    float ss2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < DT; ct++) {
#pragma unroll
        for (int e = 0; e < 16; e++) ay[ct][e] = 0.f;
        const int c = ct * 32 + r;
        const bool cv = c < a.dim;
#pragma unroll
        for (int hh = 0; hh < 4; hh++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                Frag<T> fa;
                if (cv) {
                    const T* mp = Mb + (size_t)c * 128 + hh * 32 + 16 * s + 4 * hf;  // elements j<4 at +0..3, j>=4 at +8..11 (acc_k)
                    load_4_4<T>(fa, mp, mp + 8);
                } else {
                    frag_zero<T>(fa);
                }
                mma16<T>(ay[ct], fa, qf[hh][s]);
            }
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int cc = ct * 32 + acc_row(e, lane);
            const float v = cc < a.dim ? ay[ct][e] + a.bias[cc] : 0.f;
            ay[ct][e] = v;
            ss2 += v * v;
        }
    }
    ss2 += __shfl_xor(ss2, 32, 64);
    const float sc2 = sqrtf((float)a.dim) / fmaxf(sqrtf(ss2), 1e-12f);
    // ---- RMSNorm * g2 + residual, written into this wave's own rows of the LDS tile (dim == C for LinearAttention)
#pragma unroll
    for (int ct = 0; ct < DT; ct++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int cc = ct * 32 + acc_row(e, lane);
            if (cc < a.dim) {
                T* px = xw + r * CP + cc;
                *px = from_f32<T>(ay[ct][e] * sc2 * a.g2[cc] + to_f32<T>(*px));
            }
        }
    __builtin_amdgcn_wave_barrier();
    T* out = (T*)a.out;
    for (int i = lane; i < 32 * (C / 8); i += 64) {
        const int row = i / (C / 8), v = i - row * (C / 8);
        if (pw + row < P) {
            Frag<T> f;
            frag_load<T>(f, xw + row * CP + v * 8);
            frag_store<T>(f, out + (size_t)(pw + row) * C + v * 8);
        }
    }
}

template <typename T, int KS> int launch_kvctx(const dig_la_kvctx_args* a, hipStream_t st) {
    constexpr int C = KS * 16;
    const size_t lds = (size_t)64 * (C + 8) * sizeof(T) + 4 * 32 * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)la_kv_ctx_kernel<T, KS, (KS <= 8)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((la_kv_ctx_kernel<T, KS, (KS <= 8)>), dim3(a->splits, a->B), dim3(256), lds, st, *a);
    return 0;
}
template <typename T, int KS> int launch_out(const dig_la_out_args* a, hipStream_t st) {
    constexpr int C = KS * 16, DT = (C + 31) / 32;
    const size_t lds = (size_t)128 * (C + 8) * sizeof(T);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)la_out_kernel<T, KS, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long P = (long long)a->B * a->n;
    hipLaunchKernelGGL((la_out_kernel<T, KS, DT>), dim3((unsigned)((P + 127) / 128)), dim3(256), lds, st, *a);
    return 0;
}

}  // namespace

#define DISPATCH_KS(T, FN, ARG, ST)                                   \
    switch (C / 16) {                                                 \
        case 1: FN<T, 1>(ARG, ST); break;                             \
        case 2: FN<T, 2>(ARG, ST); break;                             \
        case 4: FN<T, 4>(ARG, ST); break;                             \
        case 8: FN<T, 8>(ARG, ST); break;                             \
        case 16: FN<T, 16>(ARG, ST); break;                           \
        default: DIG_CHECK(false, DIG_EINVAL, "fused LinearAttention supports C in {16,32,64,128,256}, got %d", C); \
    }

extern "C" int dig_la_kvctx(const dig_la_kvctx_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->x && a->wkv && a->part && a->pmax && a->psum, DIG_EINVAL, "la_kvctx: null pointer");
    DIG_CHECK(a->B > 0 && a->n > 0 && a->splits > 0 && a->hid == 128 && a->B <= 65535, DIG_EINVAL, "la_kvctx: bad shape (needs 4 heads x 32)");
    const int C = a->C;
    if (a->dtype == DIG_BF16) { DISPATCH_KS(bf16, launch_kvctx, a, st) } else { DISPATCH_KS(float, launch_kvctx, a, st) }
    DIG_LAUNCH_CHECK("la_kvctx");
    return DIG_OK;
}

extern "C" int dig_la_merge(const dig_la_merge_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->part && a->pmax && a->psum && a->ctx && a->B > 0 && a->splits > 0, DIG_EINVAL, "la_merge: bad argument");
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(la_merge_kernel<bf16>, dim3(a->B * 4), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(la_merge_kernel<float>, dim3(a->B * 4), dim3(256), 0, st, *a);
    DIG_LAUNCH_CHECK("la_merge");
    return DIG_OK;
}

extern "C" int dig_la_out(const dig_la_out_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->x && a->wq && a->mb && a->bias && a->g2 && a->out, DIG_EINVAL, "la_out: null pointer");
    DIG_CHECK(a->B > 0 && a->n > 0 && a->n % 32 == 0 && a->dim == a->C, DIG_EINVAL, "la_out: needs n %% 32 == 0 and dim == C");
    const int C = a->C;
    if (a->dtype == DIG_BF16) { DISPATCH_KS(bf16, launch_out, a, st) } else { DISPATCH_KS(float, launch_out, a, st) }
    DIG_LAUNCH_CHECK("la_out");
    return DIG_OK;
}
