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
#include <cstdlib>

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

template <typename T> __device__ __forceinline__ void load4f(float (&o)[4], const T* p);
template <> __device__ __forceinline__ void load4f<bf16>(float (&o)[4], const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}
template <> __device__ __forceinline__ void load4f<float>(float (&o)[4], const float* p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
template <typename T> __device__ __forceinline__ void store4f(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void store4f<bf16>(bf16* p, const float (&o)[4]) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
}
template <> __device__ __forceinline__ void store4f<float>(float* p, const float (&o)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
}
template <typename T> __device__ __forceinline__ void copy4(T* d, const T* s);
template <> __device__ __forceinline__ void copy4<bf16>(bf16* d, const bf16* s) { *reinterpret_cast<bf16x4*>(d) = *reinterpret_cast<const bf16x4*>(s); }
template <> __device__ __forceinline__ void copy4<float>(float* d, const float* s) { *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(s); }

constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32; x <= 0 here

// ------------------------------------------------------------------------------------------------------------
// pass A.  The pixel tile is RMS-normalised once while it is staged into LDS (F.normalize(x, dim=1) * sqrt(C);
// g is folded into the weights), so k and v leave the MFMA ready to use; the next tile's rows are prefetched into
// registers under the current tile's MFMAs.  The online-softmax rescale happens once per 64-pixel tile.
template <typename T, int KS, bool WREG, bool MASK>
__device__ __forceinline__ void kv_tile(const T* xs, const Frag<T>* fk, const Frag<T>* fv, const T* wk, const T* wv, f32x16& ctx,
                                        float& m_run, float& ksum, int valid_rows, int lane) {
    constexpr int C = KS * 16, CP = C + 8;
    const int r = lane & 31, hf = lane >> 5;
    // Wide instantiations: one 32-pixel half of the tile at a time (online-softmax rescale per half): half the accumulator
    // registers - the C = 128 kernel no longer spills, 84 -> 73 us at 128x128; at C = 64 the 64-pixel form is faster (161 vs 167 us).
    if constexpr (KS * sizeof(T) >= 16) {
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
        f32x16 ak, av;
#pragma unroll
        for (int e = 0; e < 16; e++) ak[e] = av[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frag<T> fa;
            frag_load<T>(fa, xs + (sub * 32 + r) * CP + ks * 16 + hf * 8);
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
        if constexpr (MASK) {
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (sub * 32 + acc_row(e, lane) >= valid_rows) ak[e] = -INFINITY;
        }
        float mt = ak[0];
#pragma unroll
        for (int e = 1; e < 16; e++) mt = fmaxf(mt, ak[e]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        if (m_new == -INFINITY) continue;  // nothing valid yet for this d (only possible in a masked half)
        const float f = exp2_fast((m_run - m_new) * LOG2E);
#pragma unroll
        for (int e = 0; e < 16; e++) ctx[e] *= f;
        ksum *= f;
        m_run = m_new;
        const float mc = m_new * LOG2E;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            Frag<T> fA, fB;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float pe = exp2_fast(fmaf(ak[8 * s + j], LOG2E, -mc));
                ksum += pe;
                frag_set<T>(fB, j, pe);
                frag_set<T>(fA, j, av[8 * s + j]);
            }
            mma16<T>(ctx, fA, fB);
        }
    }
    return;
    }
    f32x16 ak[2], av[2];
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
#pragma unroll
        for (int e = 0; e < 16; e++) ak[sub][e] = av[sub][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frag<T> fa;
            frag_load<T>(fa, xs + (sub * 32 + r) * CP + ks * 16 + hf * 8);
            if constexpr (WREG) {
                mma16<T>(ak[sub], fa, fk[ks]);
                mma16<T>(av[sub], fa, fv[ks]);
            } else {
                Frag<T> gk, gv;
                frag_load<T>(gk, wk + ks * 16 + hf * 8);
                frag_load<T>(gv, wv + ks * 16 + hf * 8);
                mma16<T>(ak[sub], fa, gk);
                mma16<T>(av[sub], fa, gv);
            }
        }
    }
    if constexpr (MASK) {  // rows past the end of the image: k = -inf (v is already 0 there: the staged rows are zero)
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (sub * 32 + acc_row(e, lane) >= valid_rows) ak[sub][e] = -INFINITY;
    }
    float mt = fmaxf(ak[0][0], ak[1][0]);
#pragma unroll
    for (int e = 1; e < 16; e++) mt = fmaxf(mt, fmaxf(ak[0][e], ak[1][e]));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float f = exp2_fast((m_run - m_new) * LOG2E);  // 0 on the first tile (m_run = -inf)
#pragma unroll
    for (int e = 0; e < 16; e++) ctx[e] *= f;
    ksum *= f;
    m_run = m_new;
    const float mc = m_new * LOG2E;
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            Frag<T> fA, fB;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float pe = exp2_fast(fmaf(ak[sub][8 * s + j], LOG2E, -mc));  // exp2(-inf) = 0 for masked rows
                ksum += pe;
                frag_set<T>(fB, j, pe);
                frag_set<T>(fA, j, av[sub][8 * s + j]);
            }
            mma16<T>(ctx, fA, fB);  // ctxT[e][d] += sum_p v[p][e] * kexp[p][d]
        }
}

#ifndef LA_KV_WAVES
#define LA_KV_WAVES 2
#endif
template <typename T, int KS, bool WREG>
__global__ __launch_bounds__(256, (KS * (int)sizeof(T) <= 8 ? LA_KV_WAVES : (KS * (int)sizeof(T) <= 16 ? 2 : 1))) void la_kv_ctx_kernel(dig_la_kvctx_args a) {
    constexpr int C = KS * 16, CP = C + 8, TP = 64;
    constexpr int LPR = C / 8;                       // lanes that share one pixel row while staging
    constexpr int NV = (TP * LPR + 255) / 256;       // 16-byte vectors per thread per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);  // [TP][CP], rows already normalised
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
    // prefetch depth in tiles (register sets are named: static indexing); the wide instantiations have no registers to spare
#ifndef LA_KV_PD
#define LA_KV_PD 3
#endif
    constexpr int PD = (KS * (int)sizeof(T) <= 8) ? LA_KV_PD : 1;
    Frag<T> pre0[NV], pre1[NV], pre2[NV];
    auto gload = [&](Frag<T>(&pre)[NV], int t) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = tid + i * 256, row = idx / LPR, v = idx - row * LPR;
            if (idx < TP * LPR && t * TP + row < a.n) frag_load<T>(pre[i], x + (size_t)(t * TP + row) * C + v * 8);
            else frag_zero<T>(pre[i]);
        }
    };
    auto lstore = [&](Frag<T>(&pre)[NV], T* xs) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = tid + i * 256, row = idx / LPR, v = idx - row * LPR;
            float xv[8], ssq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xv[j] = frag_get<T>(pre[i], j);
                ssq += xv[j] * xv[j];
            }
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) ssq += __shfl_xor(ssq, o, 64);  // the LPR lanes of a row are adjacent
            const float inv = sqrtC / fmaxf(sqrtf(ssq), 1e-12f);              // F.normalize eps (ddp.py:150)
#pragma unroll
            for (int j = 0; j < 8; j++) frag_set<T>(pre[i], j, xv[j] * inv);
            if (idx < TP * LPR) frag_store<T>(pre[i], xs + row * CP + v * 8);
        }
    };
    // Software pipeline, PD tiles deep: a block has PD x 8..16 KB of rows in flight (two blocks per CU), enough to cover the
    // HBM latency at the CU's share of the bandwidth - with one tile ahead the kernel ran at a third of it.  The LDS tile is
    // double-buffered, so one barrier per tile suffices (the tile staged at step t is read at step t and rewritten at t + 2).
    auto step = [&](Frag<T>(&pre)[NV], int t) {
        if (t < t_end) {  // block-uniform
            T* xb = xs + ((t - t_begin) & 1) * (TP * CP);
            lstore(pre, xb);
            __syncthreads();
            if (t + PD < t_end) gload(pre, t + PD);
            const int valid = a.n - t * TP;
            if (valid >= TP) kv_tile<T, KS, WREG, false>(xb, fk, fv, wk, wv, ctx, m_run, ksum, valid, lane);
            else kv_tile<T, KS, WREG, true>(xb, fk, fv, wk, wv, ctx, m_run, ksum, valid, lane);
        }
    };
    if (t_begin < t_end) gload(pre0, t_begin);
    if constexpr (PD > 1) {
        if (t_begin + 1 < t_end) gload(pre1, t_begin + 1);
    }
    if constexpr (PD > 2) {
        if (t_begin + 2 < t_end) gload(pre2, t_begin + 2);
    }
    for (int t = t_begin; t < t_end; t += PD) {
        step(pre0, t);
        if constexpr (PD > 1) step(pre1, t + 1);
        if constexpr (PD > 2) step(pre2, t + 2);
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

// ------------------------------------------------------------------------------------------------------------
// pass A, wave-private form (bf16, C = 64, large images of a whole number of 32-pixel tiles): one WAVE owns a run of
// 32-pixel tiles and computes all four heads for them - nothing is shared between the waves but the read-only weights in
// LDS, so there is no block barrier in the loop and every wave prefetches, stages (RMS-normalises) and multiplies on its own
// schedule.  Both forms are bound by VALU issue (the exponentials, the bf16 packing, the staging), not by HBM: the
// block-cooperative kernel above holds 1.8 TB/s on the 256x256 level.  What this form drops is the tail masking (the
// dispatch only sends whole tiles here), the LDS-routed shuffles (DPP instead) and the barrier phases.  The NWV waves of a
// block merge their online-softmax states in LDS at the end (fixed wave order), so a block is one `split` of the merge.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 8 lanes, the same value (bitwise) in all 8: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ float sum8_dpp(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}
template <int KS, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void la_kv_ctx_wave_kernel(dig_la_kvctx_args a) {
    constexpr int C = KS * 16, CP = C + 8, LPR = C / 8, NV = 32 * LPR / 64;
    static_assert(LPR == 8, "sum8_dpp reduces the 8 lanes of a 64-channel row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* wl = reinterpret_cast<bf16*>(smem);  // [256][CP]: k rows (head h: 32h..), then v rows (128 + 32h..), RMSNorm gain folded in
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16* xs = wl + 256 * CP + wave * 32 * CP;  // this wave's 32 staged rows
    const int b = blockIdx.y;
    for (int i = tid; i < 256 * LPR; i += NWV * 64) {
        const int row = i / LPR, v = i - row * LPR;
        *reinterpret_cast<bf16x8*>(wl + row * CP + v * 8) = *reinterpret_cast<const bf16x8*>((const bf16*)a.wkv + (size_t)row * C + v * 8);
    }
    __syncthreads();
    const bf16* x = (const bf16*)a.x + (size_t)b * a.n * C;
    // the host checked n % 32 == 0 and n / 32 >= splits * NWV: every wave owns at least one whole tile
    const int tiles = a.n / 32, wsplits = a.splits * NWV, tps = (tiles + wsplits - 1) / wsplits;
    const int t_begin = min(tiles - 1, (blockIdx.x * NWV + wave) * tps), t_end = min(tiles, t_begin + tps);
    f32x16 ctx[4];
    float m_run[4], ksum[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
#pragma unroll
        for (int e = 0; e < 16; e++) ctx[h][e] = 0.f;
        m_run[h] = -INFINITY;
        ksum[h] = 0.f;
    }
    const float sqrtC = sqrtf((float)C);
    Frag<bf16> pre[NV];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = lane + i * 64, row = idx / LPR, v = idx - row * LPR;
            frag_load<bf16>(pre[i], x + (size_t)(t * 32 + row) * C + v * 8);
        }
    };
    gload(t_begin);
    for (int t = t_begin; t < t_end; t++) {
        __builtin_amdgcn_wave_barrier();  // the previous tile's fragment reads are done (in-order LDS, same wave)
#pragma unroll
        for (int i = 0; i < NV; i++) {  // RMS-normalise while staging: F.normalize(x, dim=1) * sqrt(C) (ddp.py:108-114), g is in the weights
            const int idx = lane + i * 64, row = idx / LPR, v = idx - row * LPR;
            float xv[8], ssq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xv[j] = frag_get<bf16>(pre[i], j);
                ssq += xv[j] * xv[j];
            }
            ssq = sum8_dpp(ssq);  // the 8 lanes of a row are adjacent
            const float inv = sqrtC / fmaxf(sqrtf(ssq), 1e-12f);
            Frag<bf16> f;
#pragma unroll
            for (int j = 0; j < 8; j++) frag_set<bf16>(f, j, xv[j] * inv);
            frag_store<bf16>(f, xs + row * CP + v * 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t + 1 < t_end) gload(t + 1);
        Frag<bf16> fa[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) frag_load<bf16>(fa[ks], xs + r * CP + ks * 16 + hf * 8);
#pragma unroll
        for (int h = 0; h < 4; h++) {
            f32x16 ak, av;
#pragma unroll
            for (int e = 0; e < 16; e++) ak[e] = av[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                Frag<bf16> gk, gv;
                frag_load<bf16>(gk, wl + (h * 32 + r) * CP + ks * 16 + hf * 8);
                frag_load<bf16>(gv, wl + (128 + h * 32 + r) * CP + ks * 16 + hf * 8);
                mma16<bf16>(ak, fa[ks], gk);
                mma16<bf16>(av, fa[ks], gv);
            }
            float mt = ak[0];
#pragma unroll
            for (int e = 1; e < 16; e++) mt = fmaxf(mt, ak[e]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run[h], mt);
            const float f = exp2_fast((m_run[h] - m_new) * LOG2E);  // 0 on the first tile (m_run = -inf)
#pragma unroll
            for (int e = 0; e < 16; e++) ctx[h][e] *= f;
            ksum[h] *= f;
            m_run[h] = m_new;
            const float mc = m_new * LOG2E;
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                Frag<bf16> fA, fB;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float pe = exp2_fast(fmaf(ak[8 * s2 + j], LOG2E, -mc));
                    ksum[h] += pe;
                    frag_set<bf16>(fB, j, pe);
                    frag_set<bf16>(fA, j, av[8 * s2 + j]);
                }
                mma16<bf16>(ctx[h], fA, fB);  // ctxT[e][d] += sum_p v[p][e] * kexp[p][d]
            }
        }
    }
    // ---- the block's waves merge (every wave has a finite running maximum): state -> LDS, M = max over the waves, every
    // wave rescales its own part, head h's four parts are summed by wave h % NWV in wave order 0..NWV-1
    __syncthreads();  // the weights and the staged rows are dead: the LDS is reused
    float* ml = reinterpret_cast<float*>(smem);                  // [NWV][4][32] running maxima
    float* sl = ml + NWV * 128;                                  // [NWV][4][32] scaled sums (written after M is known)
    float* cl = sl + NWV * 128;                                  // [NWV][3 heads in flight][16][64] scaled contexts
    static_assert(NWV == 4, "the merge deals head h to wave h");
    static_assert((size_t)(2 * NWV * 128 + NWV * 3 * 1024) * 4 <= (size_t)(256 + NWV * 32) * CP * 2, "merge scratch fits the staging LDS");
#pragma unroll
    for (int h = 0; h < 4; h++) {
        ksum[h] += __shfl_xor(ksum[h], 32, 64);
        if (hf == 0) ml[(wave * 4 + h) * 32 + r] = m_run[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; h++) {
        float M = ml[h * 32 + r];
#pragma unroll
        for (int w = 1; w < NWV; w++) M = fmaxf(M, ml[(w * 4 + h) * 32 + r]);
        const float f = exp2_fast((m_run[h] - M) * LOG2E);
        m_run[h] = M;
        if (hf == 0) sl[(wave * 4 + h) * 32 + r] = ksum[h] * f;
        if (h != wave) {
            const int slot = h - (h > wave ? 1 : 0);
#pragma unroll
            for (int e = 0; e < 16; e++) cl[((wave * 3 + slot) * 16 + e) * 64 + lane] = ctx[h][e] * f;
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) ctx[h][e] *= f;
        }
    }
    __syncthreads();
    const size_t slot = ((size_t)b * a.splits + blockIdx.x) * 4 + wave;  // head `wave`
    f32x16 own;
    float mo = 0.f;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        if (h == wave) {
            own = ctx[h];
            mo = m_run[h];
        }
    }
    f32x16 sum;
#pragma unroll
    for (int e = 0; e < 16; e++) sum[e] = 0.f;
    float ks = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
        ks += sl[(w * 4 + wave) * 32 + r];
        if (w == wave) {
#pragma unroll
            for (int e = 0; e < 16; e++) sum[e] += own[e];
        } else {
            const int sw = wave - (wave > w ? 1 : 0);
#pragma unroll
            for (int e = 0; e < 16; e++) sum[e] += cl[((w * 3 + sw) * 16 + e) * 64 + lane];
        }
    }
#pragma unroll
    for (int e = 0; e < 16; e++) a.part[slot * 1024 + acc_row(e, lane) * 32 + r] = sum[e];
    if (hf == 0) {
        a.pmax[slot * 32 + r] = mo;
        a.psum[slot * 32 + r] = ks;
    }
}

// merge of the splits: ctx[b][h][d][e] = sum_s w_s part_s[e][d] / sum_s w_s ksum_s[d],  w_s = exp(m_s - max_s m_s)
template <typename T> __global__ __launch_bounds__(256) void la_merge_kernel(dig_la_merge_args a) {
    const int bh = blockIdx.x, b = bh / 4, h = bh % 4;
    T* out = (T*)a.ctx + (size_t)bh * 1024;
    const int i = blockIdx.y * 256 + threadIdx.x;
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

// merge + M_b in one launch (one 1024-thread block per (image, head)): the merged context stays in LDS (rounded through the
// storage type, as the stand-alone merge stores it) and M_b[c][h*32+d] = sum_e Wout[c][h*32+e] ctx[d][e] (ddp.py:231 folded
// into to_out) is formed right there - two launches (la_merge 19 us + a 128-problem batched GEMM 10 us) become one of a few us.
#define LA_MAX_SPLITS 16
template <typename T> __global__ __launch_bounds__(1024) void la_merge_mb_kernel(dig_la_merge_args a) {
    __shared__ float cs[32 * 33];
    const int bh = blockIdx.x, b = bh / 4, h = bh % 4;
    const int i = threadIdx.x, e = i >> 5, d = i & 31;
    // the loads of LA_MAX_SPLITS splits at a time are issued before anything is used (a loop over the splits is one L2 round
    // trip per split - for the maxima as well as for the sums); same order and arithmetic as la_merge_kernel: the two give
    // identical contexts
    float pm[LA_MAX_SPLITS], pt[LA_MAX_SPLITS], ps[LA_MAX_SPLITS];
    auto load_chunk = [&](int s0) {
#pragma unroll
        for (int s = 0; s < LA_MAX_SPLITS; s++) {
            const size_t slot = ((size_t)b * a.splits + min(s0 + s, a.splits - 1)) * 4 + h;
            pm[s] = a.pmax[slot * 32 + d];
            pt[s] = a.part[slot * 1024 + i];
            ps[s] = a.psum[slot * 32 + d];
        }
    };
    load_chunk(0);
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < LA_MAX_SPLITS; s++) M = fmaxf(M, pm[s]);  // (slots past the end repeat the last split)
    for (int s0 = LA_MAX_SPLITS; s0 < a.splits; s0 += LA_MAX_SPLITS) {
        float q[LA_MAX_SPLITS];
#pragma unroll
        for (int s = 0; s < LA_MAX_SPLITS; s++) q[s] = a.pmax[(((size_t)b * a.splits + min(s0 + s, a.splits - 1)) * 4 + h) * 32 + d];
#pragma unroll
        for (int s = 0; s < LA_MAX_SPLITS; s++) M = fmaxf(M, q[s]);
    }
    float num = 0.f, den = 0.f;
    for (int s0 = 0; s0 < a.splits; s0 += LA_MAX_SPLITS) {
        if (s0) load_chunk(s0);
#pragma unroll
        for (int s = 0; s < LA_MAX_SPLITS; s++) {
            if (s0 + s < a.splits) {
                const float w = __expf(pm[s] - M);
                num += w * pt[s];
                den += w * ps[s];
            }
        }
    }
    const T v = from_f32<T>(num / den);
    if (a.ctx) ((T*)a.ctx)[(size_t)bh * 1024 + d * 32 + e] = v;
    cs[d * 33 + e] = to_f32<T>(v);
    __syncthreads();
    const T* wo = (const T*)a.wo;
    T* mb = (T*)a.mb + (size_t)b * a.dim * 128;
    for (int idx = i; idx < a.dim * 32; idx += 1024) {
        const int c = idx >> 5, dd = idx & 31;
        float acc = 0.f;
#pragma unroll
        for (int v8 = 0; v8 < 4; v8++) {
            Frag<T> f;
            frag_load<T>(f, wo + (size_t)c * 128 + h * 32 + v8 * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) acc = fmaf(frag_get<T>(f, j), cs[dd * 33 + v8 * 8 + j], acc);
        }
        mb[(size_t)c * 128 + h * 32 + dd] = from_f32<T>(acc);
    }
}

// ------------------------------------------------------------------------------------------------------------
// pass B.  One wave owns 32 pixels at a time (its own 32 rows of the LDS tile - no block barriers in the loop) and
// walks `iters` such sub-tiles of one image.  Where they fit (WQL / MBL, see launch_out) the q weights and this image's
// M_b sit in LDS for the block's lifetime, M_b with the columns of every 16-group permuted so that an accumulator-ordered k-fragment
// is one 16-byte read.
template <typename T, int KS, int DT, int NW, bool WQL, bool MBL>
__global__ __launch_bounds__(NW * 64, (KS <= 8 ? 2 : 1)) void la_out_kernel(dig_la_out_args a, int iters) {
    constexpr int C = KS * 16, CP = C + 8, MP = 128 + 8, TP = NW * 32, NT = NW * 64;
    constexpr int QV = C >= 32 ? 4 : C / 8;  // valid 8-row groups of an output tile (C = 16: 2; otherwise C % 32 == 0)
    constexpr int NV = (32 * (C / 8) + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* xs = reinterpret_cast<T*>(smem);                       // [TP][CP]
    float* bg = reinterpret_cast<float*>(xs + TP * CP);       // bias[DT*32] | g2[DT*32]
    T* wqs = reinterpret_cast<T*>(bg + 2 * DT * 32);          // [128][CP]            (WQL)
    T* mbs = wqs + (WQL ? 128 * CP : 0);                      // [DT*32][MP] permuted (MBL)
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const T* x = (const T*)a.x + (size_t)b * a.n * C;
    T* out = (T*)a.out + (size_t)b * a.n * C;
    const T* Mb = (const T*)a.mb + (size_t)b * a.dim * 128;
    for (int i = tid; i < DT * 32; i += NT) {
        bg[i] = i < C ? a.bias[i] : 0.f;
        bg[DT * 32 + i] = i < C ? a.g2[i] : 0.f;
    }
    if constexpr (WQL) {
        for (int i = tid; i < 128 * (C / 8); i += NT) {
            const int row = i / (C / 8), v = i - row * (C / 8);
            Frag<T> f;
            frag_load<T>(f, (const T*)a.wq + (size_t)row * C + v * 8);
            frag_store<T>(f, wqs + row * CP + v * 8);
        }
    }
    if constexpr (MBL) {
        for (int i = tid; i < DT * 32 * 32; i += NT) {
            const int row = i >> 5, pc = i & 31, g = pc >> 2, q = pc & 3;
            const int nq = (q == 1) ? 2 : (q == 2) ? 1 : q;  // [0-3 | 8-11 | 4-7 | 12-15]: lane half hf reads 8 contiguous
            if (row < C) copy4<T>(mbs + row * MP + g * 16 + nq * 4, Mb + (size_t)row * 128 + pc * 4);
            else {
                const float z[4] = {0.f, 0.f, 0.f, 0.f};
                store4f<T>(mbs + row * MP + g * 16 + nq * 4, z);
            }
        }
    }
    __syncthreads();
    T* xw = xs + wave * 32 * CP;
    const int nsub = a.n / 32;
    const int sub0 = blockIdx.x * iters * NW + wave;  // this wave's sub-tiles: sub0, sub0 + NW, ...
    Frag<T> pre[NV];
    auto gload = [&](int sub) {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = lane + i * 64, row = idx / (C / 8), v = idx - row * (C / 8);
            if (idx < 32 * (C / 8)) frag_load<T>(pre[i], x + ((size_t)sub * 32 + row) * C + v * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = lane + i * 64, row = idx / (C / 8), v = idx - row * (C / 8);
            if (idx < 32 * (C / 8)) frag_store<T>(pre[i], xw + row * CP + v * 8);
        }
    };
    constexpr bool PFX = sizeof(T) == 2 || KS <= 4;  // next sub-tile's rows wait in registers under this one's MFMAs
    if (PFX && sub0 < nsub) gload(sub0);
    for (int it = 0; it < iters; it++) {
        const int sub = sub0 + it * NW;
        if (sub >= nsub) break;
        if (!PFX) gload(sub);
        __builtin_amdgcn_wave_barrier();
        lstore();
        __builtin_amdgcn_wave_barrier();
        if (PFX && it + 1 < iters && sub + NW < nsub) gload(sub + NW);
        // ---- q[n'][p] = Wq'[n'] . x[p] per head (M tile = head, pixel = column), softmax over d, * scale -> B fragments
        Frag<T> qf[4][2];
        float c2 = 0.f;
#pragma unroll
        for (int hh = 0; hh < 4; hh++) {
            f32x16 aq;
#pragma unroll
            for (int e = 0; e < 16; e++) aq[e] = 0.f;
            float ssq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                Frag<T> fb, fa;
                frag_load<T>(fb, xw + r * CP + ks * 16 + hf * 8);
                if (hh == 0) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float xv = frag_get<T>(fb, j);
                        ssq += xv * xv;
                    }
                }
                if constexpr (WQL) frag_load<T>(fa, wqs + (hh * 32 + r) * CP + ks * 16 + hf * 8);
                else frag_load<T>(fa, (const T*)a.wq + (size_t)(hh * 32 + r) * C + ks * 16 + hf * 8);
                mma16<T>(aq, fa, fb);
            }
            if (hh == 0) {
                ssq += __shfl_xor(ssq, 32, 64);
                c2 = LOG2E * sqrtf((float)C) / fmaxf(sqrtf(ssq), 1e-12f);  // RMSNorm scale of this pixel (g is in Wq'), in log2 units
            }
            float mx = aq[0];
#pragma unroll
            for (int e = 1; e < 16; e++) mx = fmaxf(mx, aq[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mc = mx * c2;
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                aq[e] = exp2_fast(fmaf(aq[e], c2, -mc));
                sm += aq[e];
            }
            sm += __shfl_xor(sm, 32, 64);
            const float sc = a.scale / sm;
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) frag_set<T>(qf[hh][s], j, aq[8 * s + j] * sc);
        }
        // ---- y[c][p] = bias[c] + sum_n' Mb[c][n'] qs[n'][p]: contraction over the accumulator rows of q
        f32x16 ay[DT];
        float ss2 = 0.f;
#pragma unroll
        for (int ct = 0; ct < DT; ct++) {
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bg + ct * 32 + 8 * q4 + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; j++) ay[ct][4 * q4 + j] = bv[j];
            }
#pragma unroll
            for (int hh = 0; hh < 4; hh++)
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    Frag<T> fa;
                    if constexpr (MBL) {
                        frag_load<T>(fa, mbs + (ct * 32 + r) * MP + hh * 32 + 16 * s + 8 * hf);
                    } else {
                        const T* mp = Mb + (size_t)(ct * 32 + r) * 128 + hh * 32 + 16 * s + 4 * hf;  // j<4 at +0..3, j>=4 at +8..11 (acc_k)
                        load_4_4<T>(fa, mp, mp + 8);
                    }
                    mma16<T>(ay[ct], fa, qf[hh][s]);
                }
#pragma unroll
            for (int e = 0; e < 4 * QV; e++) ss2 += ay[ct][e] * ay[ct][e];
        }
        ss2 += __shfl_xor(ss2, 32, 64);
        const float sc2 = sqrtf((float)a.dim) / fmaxf(sqrtf(ss2), 1e-12f);
        // ---- RMSNorm * g2 + residual, in place in this wave's own rows of the LDS tile (dim == C)
#pragma unroll
        for (int ct = 0; ct < DT; ct++) {
#pragma unroll
            for (int q4 = 0; q4 < QV; q4++) {
                const int c0 = ct * 32 + 8 * q4 + 4 * hf;
                const f32x4 gv = *reinterpret_cast<const f32x4*>(bg + DT * 32 + c0);
                float xv[4];
                load4f<T>(xv, xw + r * CP + c0);
#pragma unroll
                for (int j = 0; j < 4; j++) xv[j] = fmaf(ay[ct][4 * q4 + j] * sc2, gv[j], xv[j]);
                store4f<T>(xw + r * CP + c0, xv);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int idx = lane + i * 64, row = idx / (C / 8), v = idx - row * (C / 8);
            if (idx < 32 * (C / 8)) {
                Frag<T> f;
                frag_load<T>(f, xw + row * CP + v * 8);
                frag_store<T>(f, out + ((size_t)sub * 32 + row) * C + v * 8);
            }
        }
    }
}

#ifndef LA_WAVE_NWV
#define LA_WAVE_NWV 4
#endif
// the wave-private kernel: bf16, C = 64, images of at least 8192 pixels in whole 32-pixel tiles (a function of the shape
// only, never of the batch); a split is a block of LA_WAVE_NWV waves there
template <typename T, int KS> constexpr bool la_wave_shape() { return sizeof(T) == 2 && KS == 4; }
template <typename T, int KS> int launch_kvctx(const dig_la_kvctx_args* a, hipStream_t st) {
    constexpr int C = KS * 16;
    if constexpr (la_wave_shape<T, KS>()) {
        if (a->n >= 8192 && a->n % 32 == 0 && (long long)a->splits * LA_WAVE_NWV * 32 <= a->n) {
            constexpr int NWV = LA_WAVE_NWV;
            const size_t lds = (size_t)(256 + NWV * 32) * (C + 8) * sizeof(T);
            static std::atomic<bool> attr_set[64];
            int dev = 0;
            (void)hipGetDevice(&dev);
            dev &= 63;
            if (!attr_set[dev].load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute((const void*)la_kv_ctx_wave_kernel<KS, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "la_kvctx: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
                attr_set[dev].store(true, std::memory_order_release);
            }
            hipLaunchKernelGGL((la_kv_ctx_wave_kernel<KS, NWV>), dim3(a->splits, a->B), dim3(NWV * 64), lds, st, *a);
            return 0;
        }
    }
    const size_t lds = (size_t)2 * 64 * (C + 8) * sizeof(T);  // double-buffered pixel tile
    constexpr bool WREG = sizeof(T) == 2 || KS <= 4;  // k/v weight fragments live in registers (fp32 fragments are twice as large)
    if (lds > 48 * 1024) {  // once per device and instantiation (not per call: the launchers must stay capture-safe after a warm-up)
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void*)la_kv_ctx_kernel<T, KS, WREG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "la_kvctx: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
            attr_set[dev].store(true, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((la_kv_ctx_kernel<T, KS, WREG>), dim3(a->splits, a->B), dim3(256), lds, st, *a);
    return 0;
}
template <typename T, int KS> int launch_out(const dig_la_out_args* a, hipStream_t st) {
    constexpr int C = KS * 16, DT = (C + 31) / 32;
    // LDS residency (bf16): C <= 64: Wq' and M_b, 4 waves, two blocks per CU; C = 128: both, 8 waves, one block per CU (140 KB);
    // C = 256: M_b only (138 KB with the 4 waves' pixel rows), Wq' from L2.  fp32 storage keeps both in L2 above C = 64.
    constexpr bool BF = sizeof(T) == 2;
    constexpr bool WQL = KS <= 4 || (BF && KS == 8), MBL = KS <= 4 || (BF && KS >= 8);
#ifndef LA_OUT_NW64
#define LA_OUT_NW64 8  // C = 64: 8 waves per block, two blocks per CU = 4 waves per SIMD (120 VGPRs): the per-wave chain LDS -> MFMA ->
#endif                 // softmax -> MFMA -> RMSNorm -> store is latency bound; 165 -> 140 us at 256x256 against 4-wave blocks
    constexpr int NW = (BF && KS == 8) ? 8 : ((BF && KS == 4) ? LA_OUT_NW64 : 4);
    size_t lds = (size_t)NW * 32 * (C + 8) * sizeof(T) + 2 * DT * 32 * sizeof(float);
    if (WQL) lds += (size_t)128 * (C + 8) * sizeof(T);
    if (MBL) lds += (size_t)DT * 32 * 136 * sizeof(T);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)la_out_kernel<T, KS, DT, NW, WQL, MBL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "la_out: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
    }
    const int tiles = (a->n + NW * 32 - 1) / (NW * 32);  // pixel tiles (one sub-tile per wave) per image
    // sub-tiles per wave: the blocks that stage 70-100 KB of weights (one per CU) get as many as still leave a block per CU
    int iters = (int)(((long long)tiles * a->B) / ((BF && KS >= 8) ? 256 : 2048));
    iters = iters < 1 ? 1 : (iters > 8 ? 8 : iters);
#ifdef LA_LAB
    if (const char* e = getenv("DIG_LAOUT_ITERS")) iters = atoi(e);
#endif
    hipLaunchKernelGGL((la_out_kernel<T, KS, DT, NW, WQL, MBL>), dim3((unsigned)((tiles + iters - 1) / iters), a->B), dim3(NW * 64), lds, st, *a, iters);
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
    DIG_CHECK(a && a->part && a->pmax && a->psum && (a->ctx || a->mb) && a->B > 0 && a->splits > 0, DIG_EINVAL, "la_merge: bad argument");
    if (a->mb) {
        DIG_CHECK(a->wo && a->dim > 0, DIG_EINVAL, "la_merge: M_b output needs wo and dim");
        DIG_CHECK(((uintptr_t)a->wo | (uintptr_t)a->mb) % 16 == 0, DIG_EALIGN, "la_merge: wo / mb must be 16-byte aligned");
        if (a->dtype == DIG_BF16) hipLaunchKernelGGL(la_merge_mb_kernel<bf16>, dim3(a->B * 4), dim3(1024), 0, st, *a);
        else hipLaunchKernelGGL(la_merge_mb_kernel<float>, dim3(a->B * 4), dim3(1024), 0, st, *a);
        DIG_LAUNCH_CHECK("la_merge(mb)");
        return DIG_OK;
    }
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(la_merge_kernel<bf16>, dim3(a->B * 4, 4), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(la_merge_kernel<float>, dim3(a->B * 4, 4), dim3(256), 0, st, *a);
    DIG_LAUNCH_CHECK("la_merge");
    return DIG_OK;
}

// ---- allocation-free variant of the context computation (dig_la_kvctx + dig_la_merge) -----------------------------------
// The split partials (part / pmax / psum) that dig_la_kvctx takes as a caller-allocated workspace live here in a file-scope
// device array, so a caller that cannot allocate (HIP graph capture, a serving loop with a frozen memory pool) needs
// nothing but its inputs and outputs.  In host code the array's NAME is only a stand-in: its device address is taken once
// per device with hipGetSymbolAddress and only that pointer is ever handed to kernels.
#define LA_NOALLOC_MAX_SLOTS 1024                                   /* B * splits */
#define LA_NOALLOC_PART_BYTES ((size_t)LA_NOALLOC_MAX_SLOTS * 4 * 1024 * sizeof(float))
#define LA_NOALLOC_VEC_BYTES ((size_t)LA_NOALLOC_MAX_SLOTS * 4 * 32 * sizeof(float))
static __device__ __align__(256) unsigned char la_scratch[LA_NOALLOC_PART_BYTES + 2 * LA_NOALLOC_VEC_BYTES];  // 17.3 MB per GPU

#include <mutex>
static std::mutex la_scratch_mu;
static std::atomic<unsigned char*> la_scratch_ptr[64];  // per device: each GPU holds its own copy of the array at its own address

extern "C" int dig_la_ctx_noalloc_max_slots(void) { return LA_NOALLOC_MAX_SLOTS; }

extern "C" int dig_la_ctx_noalloc(const dig_la_ctx_noalloc_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->x && a->wkv && a->ctx, DIG_EINVAL, "la_ctx_noalloc: null pointer");
    DIG_CHECK(a->B > 0 && a->n > 0 && a->splits > 0 && a->hid == 128 && a->B <= 65535, DIG_EINVAL, "la_ctx_noalloc: bad shape (needs 4 heads x 32)");
    DIG_CHECK(a->dtype == DIG_BF16 || a->dtype == DIG_F32, DIG_EINVAL, "la_ctx_noalloc: unknown dtype %d", a->dtype);
    DIG_CHECK((long long)a->B * a->splits <= LA_NOALLOC_MAX_SLOTS, DIG_EINVAL,
              "la_ctx_noalloc: B * splits = %lld exceeds the static scratch (%d slots); use dig_la_kvctx + dig_la_merge with a caller workspace",
              (long long)a->B * a->splits, LA_NOALLOC_MAX_SLOTS);
    int dev = 0;
    DIG_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, DIG_ELAUNCH, "la_ctx_noalloc: cannot query the current device");
    // per device, once: the array's device address (each GPU holds its own copy).  Published with release / read with acquire,
    // so the steady state takes no lock; only the first calls on a device serialise on the mutex.
    unsigned char* base = la_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!base) {
        std::lock_guard<std::mutex> lock(la_scratch_mu);
        base = la_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (!base) {
            void* ptr = nullptr;
            hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(la_scratch));
            DIG_CHECK(e == hipSuccess && ptr, DIG_ELAUNCH, "la_ctx_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
            base = (unsigned char*)ptr;
            la_scratch_ptr[dev].store(base, std::memory_order_release);
        }
    }
    dig_la_kvctx_args k;
    k.x = a->x; k.wkv = a->wkv;
    k.part = (float*)base;
    k.pmax = (float*)(base + LA_NOALLOC_PART_BYTES);
    k.psum = (float*)(base + LA_NOALLOC_PART_BYTES + LA_NOALLOC_VEC_BYTES);
    k.B = a->B; k.n = a->n; k.C = a->C; k.hid = a->hid; k.splits = a->splits; k.dtype = a->dtype;
    int rc = dig_la_kvctx(&k, stream_);  // every slot < B * splits is fully rewritten: nothing has to start at zero
    if (rc != DIG_OK) return rc;
    dig_la_merge_args m;
    m.part = k.part; m.pmax = k.pmax; m.psum = k.psum; m.ctx = a->ctx; m.B = a->B; m.splits = a->splits; m.dtype = a->dtype;
    m.wo = nullptr; m.mb = nullptr; m.dim = 0;
    (void)st;
    return dig_la_merge(&m, stream_);
}

extern "C" int dig_la_out(const dig_la_out_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->x && a->wq && a->mb && a->bias && a->g2 && a->out, DIG_EINVAL, "la_out: null pointer");
    DIG_CHECK(a->B > 0 && a->B <= 65535 && a->n > 0 && a->n % 32 == 0 && a->dim == a->C, DIG_EINVAL, "la_out: needs n %% 32 == 0 and dim == C");
    const int C = a->C;
    if (a->dtype == DIG_BF16) { DISPATCH_KS(bf16, launch_out, a, st) } else { DISPATCH_KS(float, launch_out, a, st) }
    DIG_LAUNCH_CHECK("la_out");
    return DIG_OK;
}
