// Batched NT GEMM on MFMA (gfx950): C[z][m][n] = epi(alpha * sum_k A[z][m][k] * B[z][n][k]).
// Both operands K-contiguous, staged global -> registers -> LDS (rows padded by 8 elements: conflict-free
// ds_read_b128 fragment reads) with the next K-tile's global loads in flight under the current tile's MFMAs.
// The A loader can gather a row from two tensors (fused channel concat) or from the 2x2 pixel-unshuffle of an
// NHWC tensor (Downsample); the epilogue fuses bias, GroupNorm-apply+SiLU of a second tensor, GELU, a residual
// add and an NCHW/fp32 store.  256 threads = 2x2 waves, each wave a (BM/2)x(BN/2) sub-tile of 32x32 MFMA tiles.
#include <stdlib.h>

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


// One epilogue pass over a staged fp32 tile of 64 rows x BN columns (`ot`, row stride OS floats; staged row rl belongs to
// output row m0 + (rl >> 5) * 64 + mr * 32 + (rl & 31)): bias, + SiLU(GroupNorm-apply(gsrc)), GELU, group softmax, residual,
// store - every global access a 16-byte vector of 8 consecutive columns when `vec` holds.
// The GroupNorm source (or, without one, the residual) vectors of the items are loaded by epilogue_prefetch() at kernel start,
// so that their memory latency is spent under the main loop instead of twice per block behind the staging barriers.
template <typename T, int BN>
__device__ __forceinline__ void epilogue_prefetch(const GP<T>& p, int tid, int m0, int n0, const T* R, bool vec, Frag<T> (&pre)[2][BN / 32]) {
    const T* src = p.gsrc ? p.gsrc : R;
    const int ld = p.gsrc ? p.ldg : p.ldr;
#pragma unroll
    for (int mr = 0; mr < 2; mr++)
#pragma unroll
        for (int it = 0; it < BN / 32; it++) {
            const int idx = tid + it * NTHREADS;
            const int rl = idx / (BN / 8), v8 = idx - rl * (BN / 8);
            const int m = m0 + (rl >> 5) * 64 + mr * 32 + (rl & 31);
            const int n = n0 + v8 * 8;
            if (src && vec && m < p.M && n < p.N) frag_load<T>(pre[mr][it], src + (size_t)m * ld + n);
            else frag_zero<T>(pre[mr][it]);
        }
}

template <typename T, int BN>
__device__ __forceinline__ void epilogue_rows(const GP<T>& p, const float* ot, int OS, int tid, int m0, int n0, int mr, long long coff,
                                              const T* R, bool vec, const Frag<T> (&pre)[BN / 32]) {
#pragma unroll
        for (int it = 0; it < BN / 32; it++) {
            const int idx = tid + it * NTHREADS;
            const int rl = idx / (BN / 8), v8 = idx - rl * (BN / 8);
            const int m = m0 + (rl >> 5) * 64 + mr * 32 + (rl & 31);
            const int n = n0 + v8 * 8;
            if (m >= p.M || n >= p.N) continue;
            const float* sp = ot + rl * OS + v8 * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = sp[j];
            const int nj = p.N - n < 8 ? p.N - n : 8;
            if (p.bias) {
                if (vec) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        v[j] += b0[j];
                        v[4 + j] += b1[j];
                    }
                } else {
                    for (int j = 0; j < nj; j++) v[j] += p.bias[n + j];
                }
            }
            if (p.gsrc) {
                const int img = m / p.rows_per_img;
                const float* ga = p.gaff + ((size_t)img * p.N + n) * 2;
                if (vec) {
                    const Frag<T>& g = pre[it];
#pragma unroll
                    for (int j2 = 0; j2 < 4; j2++) {
                        const f32x4 ab = *reinterpret_cast<const f32x4*>(ga + 4 * j2);
                        v[2 * j2] += silu_f(ab[0] * frag_get<T>(g, 2 * j2) + ab[1]);
                        v[2 * j2 + 1] += silu_f(ab[2] * frag_get<T>(g, 2 * j2 + 1) + ab[3]);
                    }
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
                    if (p.gsrc) frag_load<T>(g, R + (size_t)m * p.ldr + n);
                    else g = pre[it];
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
}

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

    const long long coff = zo * p.sc_o + zi * p.sc_i;
    const T* R = p.res ? p.res + zo * p.sr_o + zi * p.sr_i : nullptr;
    const bool vec = p.c_mode == 0 && (p.ldc % 8) == 0 && (coff % 8) == 0 && (!R || (p.ldr % 8) == 0) && (!p.gsrc || (p.ldg % 8) == 0) &&
                     (p.N % 8) == 0 && (((uintptr_t)p.bias | (uintptr_t)p.gaff) % 16) == 0;
    Frag<T> pre[2][BN / 32];
    epilogue_prefetch<T, BN>(p, tid, m0, n0, R, vec, pre);

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
    float* ot = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int mr = 0; mr < MR; mr++) {
#pragma unroll
        for (int nr = 0; nr < NR; nr++)
#pragma unroll
            for (int e = 0; e < 16; e++)
                ot[(wm * 32 + acc_row(e, lane)) * OS + wn * (BN / 2) + nr * 32 + r] = acc[mr][nr][e] * p.alpha;
        __syncthreads();
        epilogue_rows<T, BN>(p, ot, OS, tid, m0, n0, mr, coff, R, vec, pre[mr]);
        __syncthreads();
    }
}


// ---- bf16, K % 32 == 0: multi-stage LDS-DMA pipeline -----------------------------------------------------------------------
// The 1x1 convolutions of the Unet have K = 64..768: few K steps per tile, so the kernel above (one tile of global loads in
// flight per block, two barriers per step) spends its time waiting on memory latency.  Here the operand tiles go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS store instructions), DMA_STAGES-1 K tiles
// ahead, with counted vmcnt waits and ONE barrier per K step.  LDS rows are unpadded 64-byte K slices (the DMA writes a
// wave's 64 x 16 bytes contiguously); 16-byte slot s of row r holds K piece s ^ ((r >> 2) & 3), which makes the fragment
// ds_read_b128s conflict-free.  The MFMA takes the weight rows as its A operand, so a lane's accumulators are 4 consecutive
// output columns of one output row and the epilogue staging is written as 16-byte vectors.
// Block index -> tile: the N tiles of one M tile run back to back on ONE XCD (id % 8), so the A tile is re-read from that
// XCD's L2.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
// 3 stages x 16 KB leave room for three blocks per CU (144 VGPRs allow them): with 2-24 K steps per tile these GEMMs are bound
// by pipeline fill and epilogue latency, not by prefetch depth - measured 790 vs 817 us over a step's 26 launches against 4 stages / 2 blocks
#ifndef GEMM_DMA_STAGES
#define GEMM_DMA_STAGES 3
#endif
#ifndef GEMM_DMA_BLOCKS
#define GEMM_DMA_BLOCKS 3
#endif
constexpr int DMA_STAGES = GEMM_DMA_STAGES;

template <int N> __device__ __forceinline__ void dma_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BN>
__global__ __launch_bounds__(NTHREADS, GEMM_DMA_BLOCKS) void gemm_nt_dma_kernel(GP<bf16> p, int mtiles, int ntiles) {
    constexpr int S = DMA_STAGES;
    constexpr int MR = 2, NR = BN / 64;
    constexpr int A_BYTES = BM * 64, ST_BYTES = (BM + BN) * 64;
    constexpr int NCH = (BM + BN) / 64;  // 1-KiB DMA chunks per wave per stage: 2 of A, NCH - 2 of B
    constexpr int OS = BN + 4;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_dma[];

    // XCD-aware tile order (see above)
    int mt, nt;
    if (mtiles >= 16) {
        const int id = blockIdx.x, grp = id / (8 * ntiles), within = id - grp * 8 * ntiles;
        mt = grp * 8 + (within & 7);
        nt = within >> 3;
        if (mt >= mtiles) return;
    } else {  // few rows (time-embedding MLPs, class tokens): spread the N tiles over the XCDs
        mt = blockIdx.x / ntiles;
        nt = blockIdx.x - mt * ntiles;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z, zo = z / p.z_inner, zi = z - zo * p.z_inner;
    const bf16* A0 = p.a0 + zo * p.sa_o + zi * p.sa_i;
    const bf16* A1 = p.a1 ? p.a1 + zo * p.sa1_o + zi * p.sa1_i : nullptr;
    const bf16* Bm = p.b + zo * p.sb_o + zi * p.sb_i;

    // this lane's DMA pieces: chunk c = wave + 4 j covers tile rows 16 c + lane / 4 (A for j < 2, then B), LDS slot lane & 3
    const int prow = lane >> 2;
    const int kpiece = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const bf16* arow0[2];
    const bf16* arow1[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int m = m0 + (wave + 4 * j) * 16 + prow;
        m = m < p.M ? m : p.M - 1;  // rows past the end: re-read the last one, never stored
        if (p.a_mode == 0) {
            arow0[j] = A0 + (size_t)m * p.lda0;
            arow1[j] = A1 ? A1 + (size_t)m * p.lda1 : nullptr;
        } else {
            const int hw = p.Ho * p.Wo;
            const int bi = m / hw, rem = m - bi * hw, y = rem / p.Wo, x = rem - y * p.Wo;
            arow0[j] = A0 + (((size_t)bi * 2 * p.Ho + 2 * y) * 2 * p.Wo + 2 * x) * p.Cu;
            arow1[j] = nullptr;
        }
    }
    const bf16* brow[NCH - 2];
#pragma unroll
    for (int j = 0; j < NCH - 2; j++) {
        int n = n0 + (wave + 4 * j) * 16 + prow;
        n = n < p.N ? n : p.N - 1;
        brow[j] = Bm + (size_t)n * p.ldb;
    }
    auto issue = [&](int kt, int st) {
        unsigned char* sb = smem_dma + st * ST_BYTES;
        const int k = kt * 32 + kpiece;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const bf16* src;
            if (p.a_mode == 0) {
                src = (k < p.K0) ? arow0[j] + k : arow1[j] + (k - p.K0);
            } else {
                const int seg = k / p.Cu, c = k - seg * p.Cu;
                src = arow0[j] + ((size_t)(seg >> 1) * 2 * p.Wo + (seg & 1)) * p.Cu + c;
            }
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(sb + (wave + 4 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NCH - 2; j++)
            __builtin_amdgcn_global_load_lds((glb_void*)(brow[j] + k), (lds_void*)(sb + A_BYTES + (wave + 4 * j) * 1024), 16, 0, 0);
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < NR; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // epilogue operands first: loads return in order, so the counted waits below (on the newest DMA pieces) cover them too
    const long long coff = zo * p.sc_o + zi * p.sc_i;
    const bf16* R = p.res ? p.res + zo * p.sr_o + zi * p.sr_i : nullptr;
    const bool vec = p.c_mode == 0 && (p.ldc % 8) == 0 && (coff % 8) == 0 && (!R || (p.ldr % 8) == 0) && (!p.gsrc || (p.ldg % 8) == 0) &&
                     (p.N % 8) == 0 && (((uintptr_t)p.bias | (uintptr_t)p.gaff) % 16) == 0;
    Frag<bf16> pre[2][BN / 32];
    epilogue_prefetch<bf16, BN>(p, tid, m0, n0, R, vec, pre);

    const int nk = p.K / 32;
#pragma unroll
    for (int s = 0; s < S - 1; s++)
        if (s < nk) issue(s, s);
    const int fsw = (r >> 2) & 3;
    for (int kt = 0; kt < nk; kt++) {
        // tile kt has landed (every wave waits for its own pieces, then the barrier), tile kt - 1's buffer is free
        const int ahead = nk - 1 - kt;
        if (ahead >= S - 2) dma_wait_barrier<(S - 2) * NCH>();
        else if (ahead == 1) dma_wait_barrier<NCH>();
        else dma_wait_barrier<0>();
        if (kt + S - 1 < nk) issue(kt + S - 1, (kt + S - 1) % S);
        const unsigned char* sa = smem_dma + (kt % S) * ST_BYTES;
        const unsigned char* sbb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int slot = ((kk * 2 + hf) ^ fsw) * 16;
            Frag<bf16> fa[MR], fb[NR];
#pragma unroll
            for (int mr = 0; mr < MR; mr++) fa[mr].v = *reinterpret_cast<const bf16x8*>(sa + (wm * 64 + mr * 32 + r) * 64 + slot);
#pragma unroll
            for (int nr = 0; nr < NR; nr++) fb[nr].v = *reinterpret_cast<const bf16x8*>(sbb + (wn * (BN / 2) + nr * 32 + r) * 64 + slot);
#pragma unroll
            for (int mr = 0; mr < MR; mr++)
#pragma unroll
                for (int nr = 0; nr < NR; nr++) mma16<bf16>(acc[mr][nr], fb[nr], fa[mr]);  // rows = output columns n
        }
    }
    static_assert(S >= 2 && S <= 4, "the wait ladder above covers 2 to 4 stages");

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float* ot = reinterpret_cast<float*>(smem_dma);
    __syncthreads();  // every wave is done with the operand tiles before they become the staging tile
#pragma unroll
    for (int mr = 0; mr < MR; mr++) {
#pragma unroll
        for (int nr = 0; nr < NR; nr++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++)
                *reinterpret_cast<f32x4*>(ot + (wm * 32 + r) * OS + wn * (BN / 2) + nr * 32 + 8 * q4 + 4 * hf) =
                    f32x4{acc[mr][nr][4 * q4] * p.alpha, acc[mr][nr][4 * q4 + 1] * p.alpha, acc[mr][nr][4 * q4 + 2] * p.alpha,
                          acc[mr][nr][4 * q4 + 3] * p.alpha};
        __syncthreads();
        epilogue_rows<bf16, BN>(p, ot, OS, tid, m0, n0, mr, coff, R, vec, pre[mr]);
        __syncthreads();
    }
}

template <int BN> constexpr size_t dma_lds() {
    return (size_t)DMA_STAGES * (BM + BN) * 64 > (size_t)64 * (BN + 4) * 4 ? (size_t)DMA_STAGES * (BM + BN) * 64 : (size_t)64 * (BN + 4) * 4;
}


// ---- final_conv (ddp.py:377: Conv2d(dim, channels, 1)): [M][64] bf16 rows -> N <= 4 fp32 NCHW planes ------------------------
// 128 B in, 4 N bytes out per pixel: a stream, not a GEMM tile.  8 lanes share a pixel (one 16-byte piece each, so a wave
// load covers whole 128-byte rows), a lane group walks 8 consecutive pixels, the partial dot products are summed across the
// 8 lanes with xor-shuffles, and lane i of the group keeps pixel i: every plane store is 64 consecutive floats per wave.
template <int N>
__global__ __launch_bounds__(256) void conv1x1_k64_nchw_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, long long M, int hw) {
    const int lane = threadIdx.x & 63, g = lane >> 3, i = lane & 7;
    const long long pix0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;  // this wave's 64 pixels (same image: hw % 64 == 0)
    if (pix0 >= M) return;
    float wf[N][8];
#pragma unroll
    for (int n = 0; n < N; n++) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + n * 64 + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) wf[n][j] = (float)wv[j];
    }
    bf16x8 xv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) xv[j] = *reinterpret_cast<const bf16x8*>(x + (size_t)(pix0 + g * 8 + j) * 64 + i * 8);
    float keep[N];
#pragma unroll
    for (int n = 0; n < N; n++) keep[n] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int n = 0; n < N; n++) {
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) sacc += (float)xv[j][e] * wf[n][e];
            sacc += __shfl_xor(sacc, 1, 64);
            sacc += __shfl_xor(sacc, 2, 64);
            sacc += __shfl_xor(sacc, 4, 64);
            keep[n] = (i == j) ? sacc : keep[n];
        }
    }
    const long long img = pix0 / hw, pin = pix0 - img * hw;
#pragma unroll
    for (int n = 0; n < N; n++) out[((size_t)img * N + n) * hw + pin + lane] = keep[n] + (bias ? bias[n] : 0.f);
}

bool conv1x1_small_eligible(const dig_gemm_args* a) {
    return a->dtype == DIG_BF16 && a->c_mode == 1 && a->out_f32 && a->N >= 1 && a->N <= 4 && a->K == 64 && a->a_mode == 0 && !a->a1 && !a->res &&
           !a->gsrc && !a->gelu && a->gsm_scale == 0.f && a->alpha == 1.f && a->batch <= 1 && a->lda0 == 64 && a->ldb == 64 && a->hw % 64 == 0 &&
           a->M % 64 == 0;
}

int conv1x1_small_launch(const dig_gemm_args* a, hipStream_t stream) {
    const dim3 grid((unsigned)((a->M / 64 + 3) / 4)), block(256);
    const bf16* x = (const bf16*)a->a0;
    const bf16* w = (const bf16*)a->b;
    float* o = (float*)a->c;
    switch (a->N) {
        case 1: hipLaunchKernelGGL(conv1x1_k64_nchw_kernel<1>, grid, block, 0, stream, x, w, a->bias, o, (long long)a->M, a->hw); break;
        case 2: hipLaunchKernelGGL(conv1x1_k64_nchw_kernel<2>, grid, block, 0, stream, x, w, a->bias, o, (long long)a->M, a->hw); break;
        case 3: hipLaunchKernelGGL(conv1x1_k64_nchw_kernel<3>, grid, block, 0, stream, x, w, a->bias, o, (long long)a->M, a->hw); break;
        default: hipLaunchKernelGGL(conv1x1_k64_nchw_kernel<4>, grid, block, 0, stream, x, w, a->bias, o, (long long)a->M, a->hw); break;
    }
    DIG_LAUNCH_CHECK("gemm_nt(conv1x1 small)");
    return DIG_OK;
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
    if constexpr (sizeof(T) == 2) {
        static const bool force_old = getenv("DIG_GEMM_NO_DMA") != nullptr;  // test hook: keep the register-staged kernel
        if (a->K % 32 == 0 && !force_old) {
            constexpr size_t lds = dma_lds<BN>();
            static std::atomic<bool> attr_set[64];
            int dev = 0;
            (void)hipGetDevice(&dev);
            dev &= 63;
            if (!attr_set[dev].load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "gemm_nt: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
                attr_set[dev].store(true, std::memory_order_release);
            }
            const int mtiles = cdiv(a->M, BM), ntiles = cdiv(a->N, BN);
            dim3 grid((mtiles >= 16 ? cdiv(mtiles, 8) * 8 : mtiles) * ntiles, 1, a->batch > 0 ? a->batch : 1);
            hipLaunchKernelGGL((gemm_nt_dma_kernel<BN>), grid, dim3(NTHREADS), lds, stream, p, mtiles, ntiles);
            DIG_LAUNCH_CHECK("gemm_nt");
            return DIG_OK;
        }
    }
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
    if (conv1x1_small_eligible(a)) return conv1x1_small_launch(a, stream);
    if (a->dtype == DIG_BF16) {
        if (a->N > 64) return launch<bf16, 128, 32>(a, stream);
        return launch<bf16, 64, 32>(a, stream);
    } else if (a->dtype == DIG_F32) {
        if (a->N > 64) return launch<float, 128, 16>(a, stream);
        return launch<float, 64, 16>(a, stream);
    }
    DIG_CHECK(false, DIG_EINVAL, "gemm_nt: unknown dtype %d", a->dtype);
}
