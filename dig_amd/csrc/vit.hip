// MCTformer class-token attention -> CAM path (gfx950), fp32 throughout so that the argmax label maps match the
// reference: patch im2row, LayerNorm, multi-head attention with on-the-fly head-mean / layer-sum rollout (the
// 12 x [B,6,N,N] probability tensors of the reference are never materialised), bicubic pos-embed resize,
// CAM aggregation, flip-TTA + bilinear upsample + max_norm, and the argmax label map.
// The linears of the ViT run on the f32 MFMA GEMM (gemm.hip); the pieces here are small and HBM/LDS bound.
#include <stdlib.h>

#include "common.hpp"

namespace {

// ---- patch im2row: one thread per (patch row, c, ky) copies ps contiguous pixels --------------------------
__global__ void patchify_kernel(dig_patchify_args a) {
    const int hp = a.H / a.ps, wp = a.W / a.ps, K = 3 * a.ps * a.ps;
    const size_t total = (size_t)a.B * hp * wp * 3 * a.ps * a.ps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const size_t row = i / K;
        const int kx = k % a.ps, ky = (k / a.ps) % a.ps, c = k / (a.ps * a.ps);
        const int px = (int)(row % wp), py = (int)((row / wp) % hp), b = (int)(row / ((size_t)wp * hp));
        const bool mirror = a.flip_from > 0 && b >= a.flip_from;
        const int sb = mirror ? b - a.flip_from : b, sx = mirror ? a.W - 1 - (px * a.ps + kx) : px * a.ps + kx;
        a.rows[i] = a.img[(((size_t)sb * 3 + c) * a.H + py * a.ps + ky) * a.W + sx];
    }
}

// ---- LayerNorm: one wave per row (two-pass mean/variance in fp32) ------------------------------------------
__global__ void layernorm_kernel(dig_layernorm_args a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const float* x = a.x + row * a.D;
    float s = 0.f;
    for (int i = lane; i < a.D; i += 64) s += x[i];
    const float mean = wave_sum(s) / (float)a.D;
    float v = 0.f;
    for (int i = lane; i < a.D; i += 64) {
        const float d = x[i] - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)a.D + a.eps);
    float* o = a.out + row * a.D;
    for (int i = lane; i < a.D; i += 64) o[i] = (x[i] - mean) * rstd * a.gamma[i] + a.beta[i];
}

// ---- attention with rollout: block = (query tile of TQ rows, image); loops over heads ----------------------
constexpr int TQ = 8;
template <int DH> __global__ __launch_bounds__(256) void vit_attn_kernel(dig_vit_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = a.N, Np = (N + 3) & ~3;
    float* Ps = sm;                 // [TQ][Np] probabilities of the current head
    float* Pacc = Ps + TQ * Np;     // [TQ][Np] sum over heads
    float* Qs = Pacc + TQ * Np;     // [TQ][DH]
    const int tid = threadIdx.x;
    const int b = blockIdx.y, q0 = blockIdx.x * TQ;
    const int D3 = 3 * a.heads * DH, Dm = a.heads * DH;
    const float* qkv = a.qkv + (size_t)b * N * D3;
    const float scale = rsqrtf((float)DH);
    for (int i = tid; i < TQ * Np; i += 256) Pacc[i] = 0.f;
    for (int h = 0; h < a.heads; h++) {
        __syncthreads();
        for (int i = tid; i < TQ * DH; i += 256) {
            const int qi = i / DH, d = i - qi * DH;
            Qs[i] = (q0 + qi < N) ? qkv[(size_t)(q0 + qi) * D3 + h * DH + d] : 0.f;
        }
        __syncthreads();
        // scores: each thread owns keys j = tid, tid+256, ...
        for (int j = tid; j < N; j += 256) {
            const f32x4* kp = reinterpret_cast<const f32x4*>(qkv + (size_t)j * D3 + Dm + h * DH);
            float accq[TQ];
#pragma unroll
            for (int i = 0; i < TQ; i++) accq[i] = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < DH / 4; d4++) {
                const f32x4 kv = kp[d4];
#pragma unroll
                for (int i = 0; i < TQ; i++) {
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(Qs + i * DH + d4 * 4);
                    accq[i] += qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
                }
            }
#pragma unroll
            for (int i = 0; i < TQ; i++) Ps[i * Np + j] = accq[i] * scale;
        }
        __syncthreads();
        // softmax: 32 lanes per query row
        {
            const int qi = tid >> 5, l = tid & 31;
            float* row = Ps + qi * Np;
            float m = -INFINITY;
            for (int j = l; j < N; j += 32) m = fmaxf(m, row[j]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            float s = 0.f;
            for (int j = l; j < N; j += 32) {
                const float e = expf(row[j] - m);
                row[j] = e;
                s += e;
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float inv = 1.0f / s;
            float* arow = Pacc + qi * Np;
            for (int j = l; j < N; j += 32) {
                const float pv = row[j] * inv;
                row[j] = pv;
                arow[j] += pv;
            }
        }
        __syncthreads();
        // O = P V : thread = (query row, 2 output dims)
        {
            constexpr int TPR = DH / 2;            // threads per row
            const int qi = tid / TPR, d2 = (tid % TPR) * 2;
            if (qi < TQ && q0 + qi < N) {
                const float* prow = Ps + qi * Np;
                const float* vp = qkv + 2 * Dm + h * DH + d2;
                float o0 = 0.f, o1 = 0.f;
                for (int j = 0; j < N; j++) {
                    const f32x2 vv = *reinterpret_cast<const f32x2*>(vp + (size_t)j * D3);
                    o0 += prow[j] * vv[0];
                    o1 += prow[j] * vv[1];
                }
                float* op = a.out + ((size_t)b * N + q0 + qi) * Dm + h * DH + d2;
                op[0] = o0;
                op[1] = o1;
            }
        }
    }
    __syncthreads();
    const float nh = (float)a.heads;
    for (int i = tid; i < TQ * N; i += 256) {
        const int qi = i / N, j = i - qi * N;
        const int q = q0 + qi;
        if (q >= N) continue;
        const float mean = Pacc[qi * Np + j] / nh;  // torch.mean over heads (mctformer.py:303)
        a.acc_all[((size_t)b * N + q) * N + j] += mean;
        if (a.acc_cls && q < a.ncls) a.acc_cls[((size_t)b * a.ncls + q) * N + j] += mean;
    }
}


// ---- the same on the fp32 matrix cores (N <= 320 tokens: 224x224 with up to 124 class tokens, or smaller images) ----------
// Block = (32-query tile, image), loops over the heads.  Per head: V of the head is staged into LDS; S = Q K^T as 32x32 tiles (one key tile per wave at a
// time, Q and K fragments straight from global/L2 rows); the scaled scores go to LDS, softmax with 8 lanes per query row
// (each lane keeps its share of the head-sum in registers: no second LDS tile); O = P V with both operands from LDS, the
// keys split over the waves.  The next head's V rows and all of a head's Q / K fragments are requested ahead of their use.
// Exact fp32 products and fp32 accumulation (v_mfma_f32_32x32x2_f32), so the rolled-up probabilities keep the accuracy
// the bit-exact argmax labels need.
constexpr int TQM = 32, ATT_MAXE = 40, ATT_MAXKT = 3, ATT_MAXV = 20;
template <int DH> __global__ __launch_bounds__(256, 1) void vit_attn_mfma_kernel(dig_vit_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = a.N, nkt = (N + 31) / 32, NPk = nkt * 32, NP = NPk + 4;
    float* Ps = sm;              // [32][NP] scores, then probabilities, of the current head
    float* Vs = Ps + TQM * NP;   // [NPk][DH + 1] V of the current head (zero rows for keys >= N); also the P V partial-sum exchange
    constexpr int VP = DH + 1;   // odd row pitch: the k-strided fragment reads of P V hit consecutive banks
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, q0 = blockIdx.x * TQM;
    const int D3 = 3 * a.heads * DH, Dm = a.heads * DH;
    const float* qkv = a.qkv + (size_t)b * N * D3;
    const float scale = rsqrtf((float)DH);
    const int srow = tid >> 3, sl = tid & 7, ne = NPk / 8;  // softmax: 8 lanes per query row, elements j = sl + 8 i
    float pacc[ATT_MAXE];
#pragma unroll
    for (int i = 0; i < ATT_MAXE; i++) pacc[i] = 0.f;
    const int qrow = min(q0 + r, N - 1);  // rows past the end repeat the last query; they are never stored
    // V rows of head h as 16-byte pieces, one head ahead in registers (the block is alone on its CU: nothing else hides HBM/L2 latency)
    const int nv = NPk * (DH / 4);
    f32x4 vreg[ATT_MAXV];
    auto vload = [&](int h) {
#pragma unroll
        for (int i = 0; i < ATT_MAXV; i++) {
            const int idx = tid + i * 256, key = idx / (DH / 4), d4 = idx - key * (DH / 4);
            vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < nv && key < N) vreg[i] = *reinterpret_cast<const f32x4*>(qkv + (size_t)key * D3 + 2 * Dm + h * DH + d4 * 4);
        }
    };
    vload(0);
    for (int h = 0; h < a.heads; h++) {
        // this head's Q and K fragments: all requested up front so that their latencies overlap
        Frag<float> fq[DH / 16], fk[ATT_MAXKT][DH / 16];
#pragma unroll
        for (int ks = 0; ks < DH / 16; ks++) frag_load<float>(fq[ks], qkv + (size_t)qrow * D3 + h * DH + ks * 16 + hf * 8);
#pragma unroll
        for (int t = 0; t < ATT_MAXKT; t++) {
            const int kt = wave + 4 * t;
            if (kt < nkt) {
                const int key = min(kt * 32 + r, N - 1);
#pragma unroll
                for (int ks = 0; ks < DH / 16; ks++) frag_load<float>(fk[t][ks], qkv + (size_t)key * D3 + Dm + h * DH + ks * 16 + hf * 8);
            }
        }
        __syncthreads();  // previous head's P V reads are done
#pragma unroll
        for (int i = 0; i < ATT_MAXV; i++) {
            const int idx = tid + i * 256, key = idx / (DH / 4), d4 = idx - key * (DH / 4);
            if (idx < nv) {
#pragma unroll
                for (int j = 0; j < 4; j++) Vs[key * VP + d4 * 4 + j] = vreg[i][j];
            }
        }
        if (h + 1 < a.heads) vload(h + 1);
#pragma unroll
        for (int t = 0; t < ATT_MAXKT; t++) {
            const int kt = wave + 4 * t;
            if (kt < nkt) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < DH / 16; ks++) mma16<float>(acc, fq[ks], fk[t][ks]);
                const bool kval = kt * 32 + r < N;
#pragma unroll
                for (int e = 0; e < 16; e++) Ps[acc_row(e, lane) * NP + kt * 32 + r] = kval ? acc[e] * scale : -INFINITY;
            }
        }
        __syncthreads();
        {
            float* row = Ps + srow * NP;
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < ATT_MAXE; i++)
                if (i < ne) m = fmaxf(m, row[sl + 8 * i]);
            m = fmaxf(m, __shfl_xor(m, 1, 64));
            m = fmaxf(m, __shfl_xor(m, 2, 64));
            m = fmaxf(m, __shfl_xor(m, 4, 64));
            float ev[ATT_MAXE], ssum = 0.f;
#pragma unroll
            for (int i = 0; i < ATT_MAXE; i++)
                if (i < ne) {
                    ev[i] = expf(row[sl + 8 * i] - m);
                    ssum += ev[i];
                }
            ssum += __shfl_xor(ssum, 1, 64);
            ssum += __shfl_xor(ssum, 2, 64);
            ssum += __shfl_xor(ssum, 4, 64);
            const float inv = 1.0f / ssum;
#pragma unroll
            for (int i = 0; i < ATT_MAXE; i++)
                if (i < ne) {
                    const float pv = ev[i] * inv;
                    row[sl + 8 * i] = pv;
                    pacc[i] += pv;
                }
        }
        __syncthreads();
        // O = P V: wave = (32-channel tile, key half) for DH = 64, (the tile, key quarter) for DH = 32; the partial sums meet in LDS
        constexpr int DT = DH / 32, KSPLIT = 4 / DT;
        const int dt = wave % DT, kpart = wave / DT;
        const int nks = NPk / 16, ks0 = kpart * nks / KSPLIT, ks1 = (kpart + 1) * nks / KSPLIT;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        for (int ks = ks0; ks < ks1; ks++) {
            Frag<float> fp, fv;
            frag_load<float>(fp, Ps + r * NP + ks * 16 + hf * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) fv.v[j] = Vs[(ks * 16 + hf * 8 + j) * VP + dt * 32 + r];
            mma16<float>(acc, fp, fv);
        }
        __syncthreads();  // every wave is done reading V: its rows become the exchange buffer [kpart][dt][32 q][33]
        float* xb = Vs + ((kpart * DT + dt) * 32) * 33;
        if (kpart > 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) xb[acc_row(e, lane) * 33 + r] = acc[e];
        }
        __syncthreads();
        if (kpart == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                float v = acc[e];
#pragma unroll
                for (int kp = 1; kp < KSPLIT; kp++) v += Vs[((kp * DT + dt) * 32 + acc_row(e, lane)) * 33 + r];
                const int q = q0 + acc_row(e, lane);
                if (q < N) a.out[((size_t)b * N + q) * Dm + h * DH + dt * 32 + r] = v;
            }
        }
    }
    const float nh = (float)a.heads;
    const int q = q0 + srow;
    if (q < N) {
#pragma unroll
        for (int i = 0; i < ATT_MAXE; i++) {
            const int j = sl + 8 * i;
            if (i < ne && j < N) {
                const float mean = pacc[i] / nh;  // torch.mean over heads (mctformer.py:303)
                a.acc_all[((size_t)b * N + q) * N + j] += mean;
                if (a.acc_cls && q < a.ncls) a.acc_cls[((size_t)b * a.ncls + q) * N + j] += mean;
            }
        }
    }
}

// ---- the matrix-core path for long token sequences (320 < N <= 928: the reference's VOC-sized inputs, e.g. 368x496 -> 713 patches
// + 20 class tokens = 733 tokens, infer_DiG.py:91-118) -----------------------------------------------------------------------------
// Same block = (32-query tile, image) and head loop, but nothing of size N lives in registers and neither the scores nor V of a
// head are staged in LDS.  Two passes over the 32-key tiles of a head, the tiles dealt round-robin to the four waves:
//   pass A  S^T = K Q^T (keys down the accumulator registers, queries across the lanes, so a query's running max / sum of
//           exponentials is lane-local), merged over the 2 lane halves x 4 waves through LDS;
//   pass B  S^T again, p = exp(s - m) / l in the accumulator registers.  Those registers ARE the A operand of O += P V on
//           v_mfma_f32_32x32x2_f32 (lane (q, half) supplies P[q][key(e, half)]), the matching V row comes straight from
//           global/L2 as a coalesced 128-byte read, and p is added into the head-sum [32][N | 1] in LDS (odd pitch: the
//           32 query rows of one key hit 32 banks; every (q, key) is owned by exactly one lane, so no atomics).
// The waves' partial O tiles meet in LDS once per head.  fp32 products and accumulation throughout, as in the kernel above.
// GACC (N > 928: the head-sum row [32][N] no longer fits LDS - 368x496 at scale 1.5 / 2.0 of VOC12DatasetMS is 1573 / 2872
// tokens, voc12/data.py:794-809): p / heads is added straight into acc_all (and acc_cls) once per head.  Every (q, key) is
// still owned by exactly one lane of one block, so the read-modify-write needs no atomics; a lane's 16 keys of a tile are four
// 16-byte pieces and the two lane halves of a query cover the tile's whole 128-byte line.  The old values are loaded at the
// top of the tile, beside V, so their latency sits under the score MFMAs.
constexpr int ATT_T_MAXN = 928;
template <int DH, bool GACC> __global__ __launch_bounds__(256, 1) void vit_attn_tiled_kernel(dig_vit_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int DT = DH / 32, NKS = DH / 16;
    const int N = a.N, nkt = (N + 31) / 32, PP = N | 1;
    float* Pacc = sm;                       // [32][PP] sum over the heads of the probabilities (not with GACC)
    float* xb = Pacc + (GACC ? 0 : 32 * PP);  // [4 waves][DT][32 q][33] partial O tiles
    float* ml = xb + 4 * DT * 32 * 33;      // [8 = wave x half][32 q][2] partial (max, sum)
    float* mfin = ml + 8 * 32 * 2;          // [32 q][2] (max, 1 / sum) of the current head
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, q0 = blockIdx.x * TQM;
    const int D3 = 3 * a.heads * DH, Dm = a.heads * DH;
    const float* qkv = a.qkv + (size_t)b * N * D3;
    const float scale = rsqrtf((float)DH);
    const int qrow = min(q0 + r, N - 1);  // rows past the end repeat the last query; they are never stored
    if constexpr (!GACC)
        for (int i = tid; i < 32 * PP; i += 256) Pacc[i] = 0.f;
    const float inv_nh = 1.0f / (float)a.heads;
    const bool qvalid = q0 + r < N, cvalid = a.acc_cls != nullptr && q0 + r < a.ncls;
    float* arow = a.acc_all + ((size_t)b * N + qrow) * N;
    float* crow = a.acc_cls ? a.acc_cls + ((size_t)b * a.ncls + min(q0 + r, a.ncls - 1)) * N : nullptr;
    auto kload = [&](Frag<float>(&fk)[NKS], int kt, int h) {
        const int key = min(kt * 32 + r, N - 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) frag_load<float>(fk[ks], qkv + (size_t)key * D3 + Dm + h * DH + ks * 16 + hf * 8);
    };
    // scaled scores of key tile kt for this lane's query: s[e] belongs to key kt*32 + acc_row(e, lane); -inf past the end
    auto scores = [&](f32x16& s, const Frag<float>(&fk)[NKS], const Frag<float>(&fq)[NKS], int kt) {
#pragma unroll
        for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) mma16<float>(s, fk[ks], fq[ks]);
#pragma unroll
        for (int e = 0; e < 16; e++) s[e] = (kt * 32 + acc_row(e, lane) < N) ? s[e] * scale : -INFINITY;
    };
    for (int h = 0; h < a.heads; h++) {
        Frag<float> fq[NKS], fk[NKS], fkn[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) frag_load<float>(fq[ks], qkv + (size_t)qrow * D3 + h * DH + ks * 16 + hf * 8);
        // ---- pass A: per-lane online max / sum over this wave's key tiles
        float m_run = -INFINITY, l_run = 0.f;
        if (wave < nkt) kload(fk, wave, h);
        for (int kt = wave; kt < nkt; kt += 4) {
            if (kt + 4 < nkt) kload(fkn, kt + 4, h);
            f32x16 s;
            scores(s, fk, fq, kt);
            float tm = s[0];
#pragma unroll
            for (int e = 1; e < 16; e++) tm = fmaxf(tm, s[e]);
            if (tm > m_run) {
                l_run *= expf(m_run - tm);  // exp(-inf) = 0 on the first tile
                m_run = tm;
            }
            if (m_run > -INFINITY) {
#pragma unroll
                for (int e = 0; e < 16; e++) l_run += expf(s[e] - m_run);
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) fk[ks] = fkn[ks];
        }
        ml[((wave * 2 + hf) * 32 + r) * 2 + 0] = m_run;
        ml[((wave * 2 + hf) * 32 + r) * 2 + 1] = l_run;
        __syncthreads();
        if (tid < 32) {
            float m = -INFINITY;
#pragma unroll
            for (int p = 0; p < 8; p++) m = fmaxf(m, ml[(p * 32 + tid) * 2]);
            float l = 0.f;
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const float mp = ml[(p * 32 + tid) * 2];
                if (mp > -INFINITY) l += ml[(p * 32 + tid) * 2 + 1] * expf(mp - m);
            }
            mfin[tid * 2] = m;
            mfin[tid * 2 + 1] = 1.0f / l;
        }
        __syncthreads();
        const float m = mfin[r * 2], linv = mfin[r * 2 + 1];
        // ---- pass B: probabilities, head-sum, O += P V
        f32x16 acc_o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc_o[dt][e] = 0.f;
        if (wave < nkt) kload(fk, wave, h);
        for (int kt = wave; kt < nkt; kt += 4) {
            if (kt + 4 < nkt) kload(fkn, kt + 4, h);
            float vv[16][DT];  // V[key(e, hf)][dt*32 + r]: in flight under the score MFMAs
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int key = min(kt * 32 + acc_row(e, lane), N - 1);
#pragma unroll
                for (int dt = 0; dt < DT; dt++) vv[e][dt] = qkv[(size_t)key * D3 + 2 * Dm + h * DH + dt * 32 + r];
            }
            float old[16];  // GACC: acc_all[q][key(e, hf)] before this head
            if constexpr (GACC) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int key = kt * 32 + acc_row(e, lane);
                    old[e] = (qvalid && key < N) ? arow[key] : 0.f;
                }
            }
            f32x16 s;
            scores(s, fk, fq, kt);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int key = kt * 32 + acc_row(e, lane);
                const float p = expf(s[e] - m) * linv;  // exp(-inf) = 0 for the keys past the end
                s[e] = p;
                if constexpr (GACC) {
                    if (qvalid && key < N) {
                        arow[key] = old[e] + p * inv_nh;
                        if (cvalid) crow[key] += p * inv_nh;
                    }
                } else {
                    if (key < N) Pacc[r * PP + key] += p;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
#pragma unroll
                for (int dt = 0; dt < DT; dt++) acc_o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[e], vv[e][dt], acc_o[dt], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) fk[ks] = fkn[ks];
        }
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
#pragma unroll
            for (int e = 0; e < 16; e++) xb[((wave * DT + dt) * 32 + acc_row(e, lane)) * 33 + r] = acc_o[dt][e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 32 * DH / 256; i++) {
            const int idx = tid + i * 256, q = idx / DH, d = idx - q * DH, dt = d >> 5, dd = d & 31;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) v += xb[((w * DT + dt) * 32 + q) * 33 + dd];
            if (q0 + q < N) a.out[((size_t)b * N + q0 + q) * Dm + h * DH + d] = v;
        }
        // the next head writes ml before its first barrier and xb after its second: no barrier needed here
    }
    if constexpr (GACC) return;
    __syncthreads();
    const float nh = (float)a.heads;
    for (int i = tid; i < 32 * N; i += 256) {
        const int qi = i / N, j = i - qi * N, q = q0 + qi;
        if (q >= N) break;
        const float mean = Pacc[qi * PP + j] / nh;  // torch.mean over heads (mctformer.py:303)
        a.acc_all[((size_t)b * N + q) * N + j] += mean;
        if (a.acc_cls && q < a.ncls) a.acc_cls[((size_t)b * a.ncls + q) * N + j] += mean;
    }
}

// ---- bicubic resize of the pos-embed grid -----------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ void bicubic_pos_kernel(dig_bicubic_pos_args a) {
    const float A = -0.75f;
    const size_t total = (size_t)a.ho * a.wo * a.D;
    const float rh = a.rscale_h, rw = a.rscale_w;  // area_pixel_compute_scale with a given scale factor
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % a.D);
        const int ox = (int)((i / a.D) % a.wo), oy = (int)(i / ((size_t)a.D * a.wo));
        const float sy = rh * (oy + 0.5f) - 0.5f, sx = rw * (ox + 0.5f) - 0.5f;
        const int iy = (int)floorf(sy), ix = (int)floorf(sx);
        const float ty = sy - iy, tx = sx - ix;
        float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int yy = min(max(iy - 1 + m, 0), a.g - 1);
            float rowv = 0.f;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const int xx = min(max(ix - 1 + n, 0), a.g - 1);
                rowv += a.in[((size_t)yy * a.g + xx) * a.D + d] * wx[n];
            }
            acc += rowv * wy[m];
        }
        a.out[i] = acc;
    }
}

// ---- CAM aggregation: one block per (b, c) -----------------------------------------------------------------
__global__ void cam_fuse_kernel(dig_cam_fuse_args a) {
    extern __shared__ __attribute__((aligned(16))) float camsm[];  // [P]
    const int b = blockIdx.x / a.C, c = blockIdx.x % a.C;
    const int N = a.C + a.P;
    for (int p = threadIdx.x; p < a.P; p += blockDim.x) {
        const float mt = a.acc_cls[((size_t)b * a.C + c) * N + a.C + p];
        const float hv = fmaxf(a.head[((size_t)b * a.P + p) * a.C + c], 0.f);
        const float cm = mt * hv;
        camsm[p] = cm;
        a.mtatt[((size_t)b * a.C + c) * a.P + p] = mt;
        a.cams[((size_t)b * a.C + c) * a.P + p] = cm;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < a.P; p += blockDim.x) {
        const float* arow = a.acc_all + ((size_t)b * N + a.C + p) * N + a.C;
        float s = 0.f;
        for (int q = 0; q < a.P; q++) s += arow[q] * camsm[q];
        a.rcams[((size_t)b * a.C + c) * a.P + p] = s;
    }
}

// ---- flip-TTA + bilinear (align_corners=False) + max_norm: one block per (b, c) ----------------------------
__device__ __forceinline__ float bilin(const float* m, int hp, int wp, float sh, float sw, int y, int x) {
    float fy = sh * (y + 0.5f) - 0.5f, fx = sw * (x + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hp - 1 ? 1 : 0), x1 = x0 + (x0 < wp - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * m[y0 * wp + x0] + lx * m[y0 * wp + x1]) + ly * (hx * m[y1 * wp + x0] + lx * m[y1 * wp + x1]);
}
__global__ __launch_bounds__(256) void cam_tta_norm_kernel(dig_cam_tta_norm_args a) {
    extern __shared__ __attribute__((aligned(16))) float mm[];  // [2][hp*wp] + reduction scratch
    __shared__ float red[32];
    const int bc = blockIdx.x, b = bc / a.C, c = bc % a.C;
    const int P = a.hp * a.wp;
    float* m0 = mm;
    float* m1 = mm + P;
    for (int i = threadIdx.x; i < P; i += 256) {
        m0[i] = a.rc[(size_t)bc * P + i];
        m1[i] = a.rc_flip[(size_t)bc * P + i];
    }
    __syncthreads();
    const float lab = a.label[(size_t)b * a.C + c];
    const float sh = (float)a.hp / (float)a.H, sw = (float)a.wp / (float)a.W;
    const int HW = a.H * a.W;
    float mx = -INFINITY, mn = INFINITY;
    float* out = a.out + (size_t)bc * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int y = i / a.W, x = i - y * a.W;
        const float s = bilin(m0, a.hp, a.wp, sh, sw, y, x) * lab + bilin(m1, a.hp, a.wp, sh, sw, y, a.W - 1 - x) * lab;
        const float rl = fmaxf(s, 0.f);
        out[i] = rl;
        mx = fmaxf(mx, rl);
        mn = fminf(mn, rl);
    }
    mx = block_max(mx, red);
    mn = -block_max(-mn, red);
    const float den = mx - mn + 1e-5f;
    for (int i = threadIdx.x; i < HW; i += 256) out[i] = fmaxf(out[i] - mn - 1e-5f, 0.f) / den;  // same thread wrote out[i]
}

__global__ void cam_argmax_kernel(dig_cam_argmax_args a) {
    const size_t total = (size_t)a.B * a.HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.HW), pix = (int)(i % a.HW);
        float best = a.threshold;
        int arg = 0;
        for (int c = 0; c < a.C; c++) {
            const float v = a.label[(size_t)b * a.C + c] > 1e-5f ? a.cam[((size_t)b * a.C + c) * a.HW + pix] : 0.f;
            if (v > best) {  // strict: first maximum wins, like np.argmax
                best = v;
                arg = c + 1;
            }
        }
        a.out[i] = (unsigned char)arg;
    }
}

inline int grid_for(size_t work, int block, int cap = 256 * 8) {
    size_t g = (work + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace

#define STREAM hipStream_t stream = (hipStream_t)stream_

extern "C" int dig_patchify(const dig_patchify_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->img && a->rows && a->ps > 0 && a->H % a->ps == 0 && a->W % a->ps == 0, DIG_EINVAL, "patchify: bad argument");
    DIG_CHECK(a->B > 0 && (a->flip_from == 0 || a->B == 2 * a->flip_from), DIG_EINVAL, "patchify: flip_from=%d needs B = 2*flip_from, got B=%d", a->flip_from, a->B);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((size_t)a->B * 3 * a->H * a->W, 256)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("patchify");
    return DIG_OK;
}

extern "C" int dig_layernorm(const dig_layernorm_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->gamma && a->beta && a->out && a->rows > 0 && a->D > 0, DIG_EINVAL, "layernorm: bad argument");
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((a->rows + 3) / 4)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("layernorm");
    return DIG_OK;
}

extern "C" int dig_vit_attn(const dig_vit_attn_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->qkv && a->out && a->acc_all, DIG_EINVAL, "vit_attn: null pointer");
    DIG_CHECK(a->B > 0 && a->N > 0 && a->heads > 0 && (a->dh == 64 || a->dh == 32), DIG_EINVAL, "vit_attn: bad shape (head dim must be 32 or 64)");
    const bool no_mfma = getenv("DIG_VIT_ATTN_NO_MFMA") != nullptr;      // test hooks, read per call: keep the VALU kernel,
    const bool force_tiled = getenv("DIG_VIT_ATTN_TILED") != nullptr || getenv("DIG_VIT_ATTN_GACC") != nullptr;  // or the long-sequence kernel at any N
    if (a->N <= 8 * ATT_MAXE && !no_mfma && !force_tiled) {
        const int NP = (a->N + 31) / 32 * 32 + 4;
        size_t vsz = (size_t)(NP - 4) * (a->dh + 1);  // V rows; reused as the 4 x [32][33] partial-sum exchange of P V
        if (vsz < 4 * 32 * 33) vsz = 4 * 32 * 33;
        const size_t ldsm = ((size_t)TQM * NP + vsz) * sizeof(float);
        const dim3 gridm(cdiv(a->N, TQM), a->B);
        hipError_t e = hipSuccess;
        if (a->dh == 64) {
            e = hipFuncSetAttribute((const void*)vit_attn_mfma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);
            if (e == hipSuccess) hipLaunchKernelGGL(vit_attn_mfma_kernel<64>, gridm, dim3(256), ldsm, stream, *a);
        } else {
            e = hipFuncSetAttribute((const void*)vit_attn_mfma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);
            if (e == hipSuccess) hipLaunchKernelGGL(vit_attn_mfma_kernel<32>, gridm, dim3(256), ldsm, stream, *a);
        }
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "vit_attn: cannot raise dynamic LDS to %zu: %s", ldsm, hipGetErrorString(e));
        DIG_LAUNCH_CHECK("vit_attn");
        return DIG_OK;
    }
    if (!no_mfma) {  // key-tiled matrix-core kernel; above ATT_T_MAXN tokens the head-sum goes straight to acc_all (GACC)
        const int DT = a->dh / 32;
        const bool gacc = a->N > ATT_T_MAXN || getenv("DIG_VIT_ATTN_GACC") != nullptr;  // (test hook: GACC at any N)
        const size_t ldst = ((gacc ? 0 : (size_t)32 * (a->N | 1)) + 4 * DT * 32 * 33 + 8 * 32 * 2 + 32 * 2) * sizeof(float);
        const dim3 gridt(cdiv(a->N, TQM), a->B);
        hipError_t e = hipSuccess;
#define VIT_TILED_LAUNCH(DHV, G)                                                                                                          \
    do {                                                                                                                                 \
        e = hipFuncSetAttribute((const void*)vit_attn_tiled_kernel<DHV, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst);       \
        if (e == hipSuccess) hipLaunchKernelGGL((vit_attn_tiled_kernel<DHV, G>), gridt, dim3(256), ldst, stream, *a);                    \
    } while (0)
        if (a->dh == 64 && gacc) VIT_TILED_LAUNCH(64, true);
        else if (a->dh == 64) VIT_TILED_LAUNCH(64, false);
        else if (gacc) VIT_TILED_LAUNCH(32, true);
        else VIT_TILED_LAUNCH(32, false);
#undef VIT_TILED_LAUNCH
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "vit_attn: cannot raise dynamic LDS to %zu: %s", ldst, hipGetErrorString(e));
        DIG_LAUNCH_CHECK("vit_attn");
        return DIG_OK;
    }
    const int Np = (a->N + 3) & ~3;
    const size_t lds = (size_t)(2 * TQ * Np + TQ * a->dh) * sizeof(float);
    DIG_CHECK(lds <= 150 * 1024, DIG_EINVAL, "vit_attn: %d tokens need %zu B of LDS (max 150 KiB)", a->N, lds);
    const dim3 grid(cdiv(a->N, TQ), a->B);
    if (a->dh == 64) {
        (void)hipFuncSetAttribute((const void*)vit_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(vit_attn_kernel<64>, grid, dim3(256), lds, stream, *a);
    } else {
        (void)hipFuncSetAttribute((const void*)vit_attn_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(vit_attn_kernel<32>, grid, dim3(256), lds, stream, *a);
    }
    DIG_LAUNCH_CHECK("vit_attn");
    return DIG_OK;
}

extern "C" int dig_bicubic_pos(const dig_bicubic_pos_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out && a->g > 0 && a->ho > 0 && a->wo > 0 && a->rscale_h > 0 && a->rscale_w > 0, DIG_EINVAL, "bicubic_pos: bad argument");
    hipLaunchKernelGGL(bicubic_pos_kernel, dim3(grid_for((size_t)a->ho * a->wo * a->D, 256)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("bicubic_pos");
    return DIG_OK;
}

extern "C" int dig_cam_fuse(const dig_cam_fuse_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->acc_all && a->acc_cls && a->head && a->mtatt && a->cams && a->rcams, DIG_EINVAL, "cam_fuse: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->P > 0 && a->P * 4 <= 64 * 1024, DIG_EINVAL, "cam_fuse: bad shape");
    hipLaunchKernelGGL(cam_fuse_kernel, dim3(a->B * a->C), dim3(256), (size_t)a->P * 4, stream, *a);
    DIG_LAUNCH_CHECK("cam_fuse");
    return DIG_OK;
}

extern "C" int dig_cam_tta_norm(const dig_cam_tta_norm_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->rc && a->rc_flip && a->label && a->out, DIG_EINVAL, "cam_tta_norm: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->hp > 0 && a->wp > 0 && a->H > 0 && a->W > 0 && (size_t)a->hp * a->wp * 8 <= 60 * 1024, DIG_EINVAL, "cam_tta_norm: bad shape");
    hipLaunchKernelGGL(cam_tta_norm_kernel, dim3(a->B * a->C), dim3(256), (size_t)a->hp * a->wp * 8, stream, *a);
    DIG_LAUNCH_CHECK("cam_tta_norm");
    return DIG_OK;
}

extern "C" int dig_cam_argmax(const dig_cam_argmax_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->cam && a->label && a->out && a->B > 0 && a->C > 0 && a->C < 255 && a->HW > 0, DIG_EINVAL, "cam_argmax: bad argument");
    hipLaunchKernelGGL(cam_argmax_kernel, dim3(grid_for((size_t)a->B * a->HW, 256)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("cam_argmax");
    return DIG_OK;
}
