// Batched NT GEMM on MFMA (gfx950): C[z][m][n] = epi(alpha * sum_k A[z][m][k] * B[z][n][k]).
// Both operands K-contiguous, staged global -> registers -> LDS (rows padded by 8 elements: conflict-free
// ds_read_b128 fragment reads) with the next K-tile's global loads in flight under the current tile's MFMAs.
// The A loader can gather a row from two tensors (fused channel concat) or from the 2x2 pixel-unshuffle of an
// NHWC tensor (Downsample); the epilogue fuses bias, GroupNorm-apply+SiLU of a second tensor, GELU, a residual
// add and an NCHW/fp32 store.  256 threads = 2x2 waves, each wave a (BM/2)x(BN/2) sub-tile of 32x32 MFMA tiles.
#include "common.hpp"

namespace {

constexpr int NTHREADS = 256;
constexpr int BM = 128;

template <typename T> struct GP {
    const T* a0; const T* a1; const T* b; void* c;
    const float* bias; const T* res; const T* gsrc; const float* gaff;
    long long sa_o, sa_i, sa1_o, sa1_i, sb_o, sb_i, sc_o, sc_i, sr_o, sr_i;
    int M, N, K, K0, lda0, lda1, ldb, ldc, ldr, ldg;
    int z_inner, a_mode, Ho, Wo, Cu, c_mode, hw, rows_per_img, gelu, out_f32;
    float alpha, gsm_scale;
};

template <typename T, int BN, int BK>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(GP<T> p) {
    constexpr int BKP = BK + 8;
    constexpr int V = BK / 8;                 // vectors per row per K-tile
    constexpr int RPP = NTHREADS / V;         // rows covered per pass
    constexpr int APASS = BM / RPP, BPASS = (BN + RPP - 1) / RPP;
    static_assert(BM % RPP == 0, "tile/loader mismatch");
    constexpr int MR = 2, NR = BN / 64;
    constexpr int OS = BN + 4;  // fp32 staging row stride of the epilogue
    constexpr size_t MAIN_BYTES = (size_t)(BM + BN) * BKP * sizeof(T), EPI_BYTES = (size_t)64 * OS * sizeof(float);
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
    T* As = reinterpret_cast<T*>(smem_raw);
    T* Bs = As + BM * BKP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int z = blockIdx.z, zo = z / p.z_inner, zi = z - zo * p.z_inner;

    const T* A0 = p.a0 + zo * p.sa_o + zi * p.sa_i;
    const T* A1 = p.a1 ? p.a1 + zo * p.sa1_o + zi * p.sa1_i : nullptr;
    const T* Bm = p.b + zo * p.sb_o + zi * p.sb_i;

    const int kv = tid % V, lrow = tid / V;
    // per-thread row descriptors (fixed across K tiles)
    const T* arow0[APASS];
    const T* arow1[APASS];
    bool aok[APASS];
#pragma unroll
    for (int i = 0; i < APASS; i++) {
        const int m = m0 + lrow + i * RPP;
        aok[i] = m < p.M;
        const int mm = aok[i] ? m : 0;
        if (p.a_mode == 0) {
            arow0[i] = A0 + (size_t)mm * p.lda0;
            arow1[i] = A1 ? A1 + (size_t)mm * p.lda1 : nullptr;
        } else {  // 2x2 pixel-unshuffle gather
            const int hw = p.Ho * p.Wo;
            const int bi = mm / hw, rem = mm - bi * hw, y = rem / p.Wo, x = rem - y * p.Wo;
            arow0[i] = A0 + (((size_t)bi * 2 * p.Ho + 2 * y) * 2 * p.Wo + 2 * x) * p.Cu;
            arow1[i] = nullptr;
        }
    }
    const T* brow[BPASS];
    bool bok[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; i++) {
        const int n = n0 + lrow + i * RPP;
        bok[i] = n < p.N && (lrow + i * RPP) < BN;
        brow[i] = Bm + (size_t)(bok[i] ? n : 0) * p.ldb;
    }

    Frag<T> ra[APASS], rb[BPASS];
    auto gload = [&](int kt) {
        const int k = kt * BK + kv * 8;
        const bool kok = k < p.K;
#pragma unroll
        for (int i = 0; i < APASS; i++) {
            if (aok[i] && kok) {
                const T* src;
                if (p.a_mode == 0) {
                    src = (k < p.K0) ? arow0[i] + k : arow1[i] + (k - p.K0);
                } else {
                    const int seg = k / p.Cu, c = k - seg * p.Cu;
                    src = arow0[i] + ((size_t)(seg >> 1) * 2 * p.Wo + (seg & 1)) * p.Cu + c;
                }
                frag_load<T>(ra[i], src);
            } else {
                frag_zero<T>(ra[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < BPASS; i++) {
            if (bok[i] && kok) frag_load<T>(rb[i], brow[i] + k);
            else frag_zero<T>(rb[i]);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < APASS; i++) frag_store<T>(ra[i], As + (lrow + i * RPP) * BKP + kv * 8);
#pragma unroll
        for (int i = 0; i < BPASS; i++)
            if (lrow + i * RPP < BN) frag_store<T>(rb[i], Bs + (lrow + i * RPP) * BKP + kv * 8);
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < NR; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 16; kk++) {
            Frag<T> fa[MR], fb[NR];
#pragma unroll
            for (int mr = 0; mr < MR; mr++)
                frag_load<T>(fa[mr], As + (wm * 64 + mr * 32 + r) * BKP + kk * 16 + hf * 8);
#pragma unroll
            for (int nr = 0; nr < NR; nr++)
                frag_load<T>(fb[nr], Bs + (wn * (BN / 2) + nr * 32 + r) * BKP + kk * 16 + hf * 8);
#pragma unroll
            for (int mr = 0; mr < MR; mr++)
#pragma unroll
                for (int nr = 0; nr < NR; nr++) mma16<T>(acc[mr][nr], fa[mr], fb[nr]);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            lstore();
            __syncthreads();
        }
    }

    // ---- epilogue: the fp32 tile goes through LDS (64 rows at a time) so that every global access of the epilogue
    // (bias, GroupNorm source, residual, output) is a 16-byte vector of 8 consecutive columns.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA retired before the accumulators are read
    const long long coff = zo * p.sc_o + zi * p.sc_i;
    const T* R = p.res ? p.res + zo * p.sr_o + zi * p.sr_i : nullptr;
    float* ot = reinterpret_cast<float*>(smem_raw);
    const bool vec = p.c_mode == 0 && (p.ldc % 8) == 0 && (coff % 8) == 0 && (!R || (p.ldr % 8) == 0) && (!p.gsrc || (p.ldg % 8) == 0) &&
                     (p.N % 8) == 0;
#pragma unroll
    for (int mr = 0; mr < MR; mr++) {
#pragma unroll
        for (int nr = 0; nr < NR; nr++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ot[(wm * 32 + acc_row(e, lane)) * OS + wn * (BN / 2) + nr * 32 + r] = acc[mr][nr][e] * p.alpha;
        __syncthreads();
        for (int idx = tid; idx < 64 * (BN / 8); idx += NTHREADS) {
            const int rl = idx / (BN / 8), v8 = idx - rl * (BN / 8);
            const int m = m0 + (rl >> 5) * 64 + mr * 32 + (rl & 31);
            const int n = n0 + v8 * 8;
            if (m >= p.M || n >= p.N) continue;
            const float* sp = ot + rl * OS + v8 * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = sp[j];
            const int nj = p.N - n < 8 ? p.N - n : 8;
            if (p.bias)
                for (int j = 0; j < nj; j++) v[j] += p.bias[n + j];
            if (p.gsrc) {
                const int img = m / p.rows_per_img;
                const float* ga = p.gaff + ((size_t)img * p.N + n) * 2;
                if (vec) {
                    Frag<T> g;
                    frag_load<T>(g, p.gsrc + (size_t)m * p.ldg + n);
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] += silu_f(ga[2 * j] * frag_get<T>(g, j) + ga[2 * j + 1]);
                } else {
                    for (int j = 0; j < nj; j++) v[j] += silu_f(ga[2 * j] * to_f32<T>(p.gsrc[(size_t)m * p.ldg + n + j]) + ga[2 * j + 1]);
                }
            }
            if (p.gelu) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = gelu_erf_f(v[j]);
            }
            if (p.gsm_scale != 0.f) {  // softmax over each aligned group of 32 columns (4 adjacent lanes x 8), times scale
                float mx = v[0];
#pragma unroll
                for (int j = 1; j < 8; j++) mx = fmaxf(mx, v[j]);
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                float sm = 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    v[j] = __expf(v[j] - mx);
                    sm += v[j];
                }
                sm += __shfl_xor(sm, 1, 64);
                sm += __shfl_xor(sm, 2, 64);
                const float inv = p.gsm_scale / sm;
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] *= inv;
            }
            if (R) {
                if (vec) {
                    Frag<T> g;
                    frag_load<T>(g, R + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] += frag_get<T>(g, j);
                } else {
                    for (int j = 0; j < nj; j++) v[j] += to_f32<T>(R[(size_t)m * p.ldr + n + j]);
                }
            }
            if (vec) {
                const size_t idxo = coff + (size_t)m * p.ldc + n;
                if (p.out_f32) {
                    float* d = (float*)p.c + idxo;
                    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    Frag<T> f;
#pragma unroll
                    for (int j = 0; j < 8; j++) frag_set<T>(f, j, v[j]);
                    frag_store<T>(f, (T*)p.c + idxo);
                }
            } else {
                for (int j = 0; j < nj; j++) {
                    size_t idxo;
                    if (p.c_mode == 0) idxo = (size_t)m * p.ldc + n + j;
                    else {
                        const int bi = m / p.hw, pix = m - bi * p.hw;
                        idxo = ((size_t)bi * p.N + n + j) * p.hw + pix;
                    }
                    if (p.out_f32) ((float*)p.c)[coff + idxo] = v[j];
                    else ((T*)p.c)[coff + idxo] = from_f32<T>(v[j]);
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int BN, int BK> int launch(const dig_gemm_args* a, hipStream_t stream) {
    GP<T> p;
    p.a0 = (const T*)a->a0; p.a1 = (const T*)a->a1; p.b = (const T*)a->b; p.c = a->c;
    p.bias = a->bias; p.res = (const T*)a->res; p.gsrc = (const T*)a->gsrc; p.gaff = a->gaff;
    p.sa_o = a->sa_o; p.sa_i = a->sa_i; p.sa1_o = a->sa1_o; p.sa1_i = a->sa1_i;
    p.sb_o = a->sb_o; p.sb_i = a->sb_i; p.sc_o = a->sc_o; p.sc_i = a->sc_i; p.sr_o = a->sr_o; p.sr_i = a->sr_i;
    p.M = a->M; p.N = a->N; p.K = a->K; p.K0 = a->a1 ? a->K0 : a->K;
    p.lda0 = a->lda0; p.lda1 = a->lda1; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldg = a->ldg;
    p.z_inner = a->z_inner > 0 ? a->z_inner : 1;
    p.a_mode = a->a_mode; p.Ho = a->unsh_Ho; p.Wo = a->unsh_Wo; p.Cu = a->unsh_C;
    p.c_mode = a->c_mode; p.hw = a->hw > 0 ? a->hw : 1; p.rows_per_img = a->rows_per_img > 0 ? a->rows_per_img : 1;
    p.gelu = a->gelu; p.out_f32 = a->out_f32; p.alpha = a->alpha; p.gsm_scale = a->gsm_scale;
    dim3 grid(cdiv(a->M, BM), cdiv(a->N, BN), a->batch > 0 ? a->batch : 1);
    hipLaunchKernelGGL((gemm_nt_kernel<T, BN, BK>), grid, dim3(NTHREADS), 0, stream, p);
    DIG_LAUNCH_CHECK("gemm_nt");
    return DIG_OK;
}

}  // namespace

extern "C" int dig_gemm_nt(const dig_gemm_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->a0 && a->b && a->c, DIG_EINVAL, "gemm_nt: null pointer");
    DIG_CHECK(a->M > 0 && a->N > 0 && a->K > 0 && a->K % 8 == 0, DIG_EINVAL, "gemm_nt: bad M=%d N=%d K=%d (K must be a multiple of 8)", a->M, a->N, a->K);
    const int esz = a->dtype == DIG_F32 ? 4 : 2;
    const int al = 16 / esz;  // element alignment of every row start
    DIG_CHECK(a->ldb % al == 0 && ((uintptr_t)a->b % 16) == 0 && a->sb_o % al == 0 && a->sb_i % al == 0, DIG_EALIGN, "gemm_nt: B operand not 16-byte aligned");
    if (a->a_mode == 0) {
        DIG_CHECK(a->lda0 % al == 0 && ((uintptr_t)a->a0 % 16) == 0 && a->sa_o % al == 0 && a->sa_i % al == 0, DIG_EALIGN, "gemm_nt: A operand not 16-byte aligned");
        if (a->a1) {
            DIG_CHECK(a->K0 > 0 && a->K0 < a->K && a->K0 % 8 == 0 && a->lda1 % al == 0 && ((uintptr_t)a->a1 % 16) == 0, DIG_EALIGN, "gemm_nt: bad concat split K0=%d", a->K0);
        }
    } else {
        DIG_CHECK(a->a_mode == 1 && a->unsh_C % 8 == 0 && a->K == 4 * a->unsh_C && a->unsh_Ho > 0 && a->unsh_Wo > 0 && a->a1 == nullptr &&
                      a->M % (a->unsh_Ho * a->unsh_Wo) == 0, DIG_EINVAL, "gemm_nt: bad unshuffle geometry");
    }
    DIG_CHECK(a->c_mode == 0 || (a->c_mode == 1 && a->hw > 0 && a->M % a->hw == 0), DIG_EINVAL, "gemm_nt: bad NCHW store geometry");
    DIG_CHECK(!a->gsrc || (a->gaff && a->rows_per_img > 0), DIG_EINVAL, "gemm_nt: gsrc needs gaff and rows_per_img");
    DIG_CHECK(a->batch <= 65535, DIG_EINVAL, "gemm_nt: batch too large");
    DIG_CHECK(a->gsm_scale == 0.f || (a->N % 32 == 0 && a->c_mode == 0), DIG_EINVAL, "gemm_nt: group softmax needs N %% 32 == 0");
    if (a->dtype == DIG_BF16) {
        if (a->N > 64) return launch<bf16, 128, 32>(a, stream);
        return launch<bf16, 64, 32>(a, stream);
    } else if (a->dtype == DIG_F32) {
        if (a->N > 64) return launch<float, 128, 16>(a, stream);
        return launch<float, 64, 16>(a, stream);
    }
    DIG_CHECK(false, DIG_EINVAL, "gemm_nt: unknown dtype %d", a->dtype);
}
