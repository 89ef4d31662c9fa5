// Full softmax attention, dim_head 32 (Attend, attend.py:98-113; ddp.py:235-264), flash style for gfx950: the
// [n][n] logits never leave the registers.
//
// Per (image, head) a 256-thread block owns 128 queries (32 per wave) and streams the keys through LDS in tiles of
// 128.  Everything is computed transposed so that a query is a COLUMN of the MFMA tiles:
//     S^T[j][i] = k_j . q_i                (A = K tile rows, B = q fragments held in registers)
//     softmax over j = over the accumulator rows: in-lane over the 16 registers + one xor-32 shuffle
//     O^T[e][i] += sum_j v[j][e] P^T[j][i] (A = V^T from LDS, B = P^T taken straight from the accumulator registers)
// V arrives channel-major ([B][hid][n], written that way by the projection GEMM); its LDS image has the keys of
// every 16-group permuted to the accumulator's row order so that an A fragment is one 16-byte read.
// The online-softmax rescale runs once per 128-key tile.  Storage type T in {bf16, float}; statistics fp32.
#include "common.hpp"

namespace {

constexpr int KT = 128;  // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> __device__ __forceinline__ void store_half(T* dst, const Frag<T>& f, int half);
template <> __device__ __forceinline__ void store_half<bf16>(bf16* dst, const Frag<bf16>& f, int half) {
    *reinterpret_cast<bf16x4*>(dst) = half ? bf16x4{f.v[4], f.v[5], f.v[6], f.v[7]} : bf16x4{f.v[0], f.v[1], f.v[2], f.v[3]};
}
template <> __device__ __forceinline__ void store_half<float>(float* dst, const Frag<float>& f, int half) {
    *reinterpret_cast<f32x4*>(dst) = half ? f32x4{f.v[4], f.v[5], f.v[6], f.v[7]} : f32x4{f.v[0], f.v[1], f.v[2], f.v[3]};
}
template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)a, (bf16)b, (bf16)c, (bf16)d};
}
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}

template <typename T> struct AttnCfg {
    static constexpr int KRS = 32 + 16 / (int)sizeof(T);  // K tile row stride (elements): 32 d + one 16-byte slot
    static constexpr int VRS = KT + 16 / (int)sizeof(T);  // V^T tile row stride
    static constexpr size_t LDS = (size_t)(KT * KRS + 32 * VRS) * sizeof(T);
};

template <typename T, bool MASK>
__device__ __forceinline__ void attn_tile(const T* ks, const T* vs, const Frag<T> (&fq)[2], f32x16& o, float& m_run, float& l_run,
                                          float c, int valid_keys, int lane) {
    using Cfg = AttnCfg<T>;
    const int r = lane & 31, hf = lane >> 5;
    f32x16 s[KT / 32];
#pragma unroll
    for (int kt = 0; kt < KT / 32; kt++) {
#pragma unroll
        for (int e = 0; e < 16; e++) s[kt][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            Frag<T> fa;
            frag_load<T>(fa, ks + (kt * 32 + r) * Cfg::KRS + kk * 16 + hf * 8);
            mma16<T>(s[kt], fa, fq[kk]);
        }
    }
    if constexpr (MASK) {
#pragma unroll
        for (int kt = 0; kt < KT / 32; kt++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (kt * 32 + acc_row(e, lane) >= valid_keys) s[kt][e] = -INFINITY;
    }
    float mt = s[0][0];
#pragma unroll
    for (int kt = 0; kt < KT / 32; kt++)
#pragma unroll
        for (int e = 0; e < 16; e++) mt = fmaxf(mt, s[kt][e]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);          // raw logits; the softmax scale (> 0) is applied inside exp2
    const float f = __builtin_amdgcn_exp2f((m_run - m_new) * c);  // 0 on the first tile
#pragma unroll
    for (int e = 0; e < 16; e++) o[e] *= f;
    l_run *= f;
    m_run = m_new;
    const float mc = m_new * c;
#pragma unroll
    for (int kt = 0; kt < KT / 32; kt++)
#pragma unroll
        for (int sx = 0; sx < 2; sx++) {
            Frag<T> fp, fv;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][8 * sx + j], c, -mc));
                l_run += p;
                frag_set<T>(fp, j, p);
            }
            frag_load<T>(fv, vs + r * Cfg::VRS + kt * 32 + 16 * sx + 8 * hf);
            mma16<T>(o, fv, fp);
        }
}

#ifndef FA_BLOCKS
#define FA_BLOCKS 2
#endif
template <typename T> __global__ __launch_bounds__(256, FA_BLOCKS) void flash_attn_kernel(dig_flash_attn_args a) {
    using Cfg = AttnCfg<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* ks = reinterpret_cast<T*>(smem);   // [KT][KRS]
    T* vs = ks + KT * Cfg::KRS;           // [32][VRS], keys permuted inside every 16-group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh % a.heads;
    const int hid = a.heads * 32;
    const T* qk = (const T*)a.qk + (size_t)b * a.n * a.ldqk;
    const T* vT = (const T*)a.vT + ((size_t)b * hid + h * 32) * a.n;
    const int i0 = blockIdx.x * 128 + wave * 32;  // this wave's first query
    Frag<T> fq[2];
    {
        const int i = i0 + r;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            if (i < a.n) frag_load<T>(fq[kk], qk + (size_t)i * a.ldqk + h * 32 + kk * 16 + hf * 8);
            else frag_zero<T>(fq[kk]);
        }
    }
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; e++) o[e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale * LOG2E;
    // staging: K tile = KT rows x 4 vectors, V^T tile = 32 rows x KT/8 vectors; 2 + 2 vectors per thread
    constexpr int NK = KT * 4 / 256, NVV = 32 * (KT / 8) / 256;
    Frag<T> pk[NK], pv[NVV];
    auto gload = [&](int j0) {
#pragma unroll
        for (int i = 0; i < NK; i++) {
            const int idx = tid + i * 256, row = idx >> 2, v = idx & 3;
            if (j0 + row < a.n) frag_load<T>(pk[i], qk + (size_t)(j0 + row) * a.ldqk + hid + h * 32 + v * 8);
            else frag_zero<T>(pk[i]);
        }
#pragma unroll
        for (int i = 0; i < NVV; i++) {
            const int idx = tid + i * 256, row = idx / (KT / 8), v = idx % (KT / 8);
            if (j0 + v * 8 < a.n) frag_load<T>(pv[i], vT + (size_t)row * a.n + j0 + v * 8);  // n % 8 == 0: whole vector valid
            else frag_zero<T>(pv[i]);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NK; i++) {
            const int idx = tid + i * 256, row = idx >> 2, v = idx & 3;
            frag_store<T>(pk[i], ks + row * Cfg::KRS + v * 8);
        }
#pragma unroll
        for (int i = 0; i < NVV; i++) {
            const int idx = tid + i * 256, row = idx / (KT / 8), v = idx % (KT / 8);
            // keys 8v..8v+7 = pieces (2(v&1), 2(v&1)+1) of 16-group v>>1; piece order in LDS is [0, 2, 1, 3]
            T* dst = vs + row * Cfg::VRS + (v >> 1) * 16 + (v & 1) * 4;
            store_half<T>(dst, pv[i], 0);
            store_half<T>(dst + 8, pv[i], 1);
        }
    };
    gload(0);
    for (int j0 = 0; j0 < a.n; j0 += KT) {
        __syncthreads();  // previous tile's reads are done
        lstore();
        __syncthreads();
        if (j0 + KT < a.n) gload(j0 + KT);
        const int valid = a.n - j0;
        if (valid >= KT) attn_tile<T, false>(ks, vs, fq, o, m_run, l_run, c, valid, lane);
        else attn_tile<T, true>(ks, vs, fq, o, m_run, l_run, c, valid, lane);
    }
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_run;
    const int i = i0 + r;
    if (i < a.n) {
        T* dst = (T*)a.out + ((size_t)b * a.n + i) * hid + h * 32;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++)
            store4<T>(dst + 8 * q4 + 4 * hf, o[4 * q4] * inv, o[4 * q4 + 1] * inv, o[4 * q4 + 2] * inv, o[4 * q4 + 3] * inv);
    }
}

template <typename T> int launch(const dig_flash_attn_args* a, hipStream_t st) {
    hipLaunchKernelGGL(flash_attn_kernel<T>, dim3((a->n + 127) / 128, a->B * a->heads), dim3(256), AttnCfg<T>::LDS, st, *a);
    return 0;
}

}  // namespace

extern "C" int dig_flash_attn(const dig_flash_attn_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->qk && a->vT && a->out, DIG_EINVAL, "flash_attn: null pointer");
    DIG_CHECK(a->B > 0 && a->heads > 0 && a->n > 0 && a->n % 8 == 0 && a->ldqk >= 2 * a->heads * 32 && a->ldqk % 8 == 0, DIG_EINVAL,
              "flash_attn: bad shape (n=%d must be a multiple of 8, ldqk=%d)", a->n, a->ldqk);
    DIG_CHECK((long long)a->B * a->heads <= 65535, DIG_EINVAL, "flash_attn: B*heads too large");
    DIG_CHECK(((uintptr_t)a->qk | (uintptr_t)a->vT | (uintptr_t)a->out) % 16 == 0, DIG_EALIGN, "flash_attn: pointers must be 16-byte aligned");
    if (a->dtype == DIG_BF16) launch<bf16>(a, st);
    else if (a->dtype == DIG_F32) launch<float>(a, st);
    else DIG_CHECK(false, DIG_EINVAL, "flash_attn: unknown dtype %d", a->dtype);
    DIG_LAUNCH_CHECK("flash_attn");
    return DIG_OK;
}
