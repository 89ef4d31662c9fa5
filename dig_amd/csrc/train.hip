// Backward pass and optimiser step of the MCTformer classification losses (models/model_ECCV24_DiG.py:337-341,406-410,
// 424-425: loss_trm.backward(); opt_trm.step()), gfx950, fp32 throughout (the CAM path's arithmetic type).
//
// The reference has no backward code of its own - autograd derives it from the forward modules (vision_transformer.py:54-103,
// 177-196; mctformer.py:227-300).  What is here are the adjoints of those forward operators, composed by dig_amd/train.py:
//   gemm_f32        C = alpha * A B (+ bias) (+ C) with ARBITRARY element strides on A and B and two-level batching: every
//                   product of the backward pass (dY W, dY^T X, P^T dO, dO V^T, dS K, dS^T Q) is one launch of it on the
//                   tensors where they lie - nothing is transposed in memory.  fp32 MFMA (v_mfma_f32_32x32x2_f32), 128x64
//                   tile, k-major LDS tiles, the next tile's global loads in flight under the MFMAs.
//   layernorm_bwd   dx (+ the residual path's gradient) and per-row-block partial sums of dgamma / dbeta
//   colsum          deterministic two-stage column sums (bias gradients, the LayerNorm partials, pos_embed over the batch)
//   gelu_bwd, softmax_bwd, msm_grad, head_pool_bwd   adjoints of the erf-GELU, the row softmax (scaled logits), the
//                   multilabel soft-margin loss and the top-3 pooled patch logits
//   adamw           torch.optim.AdamW's update (decoupled decay, bias-corrected moments), one launch per parameter
// No atomics: every sum has a fixed order, so a step is bitwise reproducible.
#include "common.hpp"

namespace {

#define STREAM hipStream_t stream = (hipStream_t)stream_

// ------------------------------------------------------------------------------------------------------------------------
// strided fp32 GEMM.  Block = 256 threads = 4 waves, tile 128 (M) x 64 (N), BK = 16; wave w owns rows 32w..32w+31 and both
// 32-column halves (2 accumulator tiles).  LDS tiles are k-major (As[k][m], Bs[k][n]) so that an MFMA operand read - lane l
// wants element (row or column l & 31, k = l >> 5) - is 32 consecutive floats per half wave.
constexpr int GBM = 128, GBN = 64, GBK = 16, GPA = GBM + 4, GPB = GBN + 4;

__global__ __launch_bounds__(256) void gemm_f32_kernel(dig_gemm_f32_args a) {
    __shared__ float As[2][GBK][GPA];
    __shared__ float Bs[2][GBK][GPB];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // split-K (ksplit > 1: the weight gradients - a small output summed over every token): K slice zk writes its own partial
    // output at c + zk * sc_k; the caller adds the slices with dig_colsum (fixed order)
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    const int zk = blockIdx.z % ks, z = blockIdx.z / ks, zo = z / a.z_inner, zi = z - zo * a.z_inner;
    const int kc = ((a.K + ks - 1) / ks + GBK - 1) / GBK * GBK;
    const int k_begin = zk * kc, k_end = min(a.K, k_begin + kc);
    const float* A = a.a + (size_t)zo * a.sa_o + (size_t)zi * a.sa_i;
    const float* B = a.b + (size_t)zo * a.sb_o + (size_t)zi * a.sb_i;
    float* C = a.c + (size_t)zo * a.sc_o + (size_t)zi * a.sc_i + (size_t)zk * a.sc_k;
    const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
    // global -> register staging: 8 A elements and 4 B elements per thread and K tile.  The thread -> element map follows
    // the contiguous direction of each operand (uniform branch), so a wave's loads are whole 256-byte runs either way.
    const bool a_kfast = a.sak == 1, b_kfast = a.sbk == 1 && a.sbn != 1;
    float ra[8], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int mm, kk;
            if (a_kfast) { kk = tid & 15; mm = (tid >> 4) + 16 * i; }
            else { mm = (tid & 127); kk = (tid >> 7) + 2 * i; }
            const int m = m0 + mm, k = k0 + kk;
            ra[i] = (m < a.M && k < k_end) ? A[(size_t)m * a.sam + (size_t)k * a.sak] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int nn, kk;
            if (b_kfast) { kk = tid & 15; nn = (tid >> 4) + 16 * i; }
            else { nn = tid & 63; kk = (tid >> 6) + 4 * i; }
            const int n = n0 + nn, k = k0 + kk;
            rb[i] = (n < a.N && k < k_end) ? B[(size_t)k * a.sbk + (size_t)n * a.sbn] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int mm, kk;
            if (a_kfast) { kk = tid & 15; mm = (tid >> 4) + 16 * i; }
            else { mm = (tid & 127); kk = (tid >> 7) + 2 * i; }
            As[buf][kk][mm] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int nn, kk;
            if (b_kfast) { kk = tid & 15; nn = (tid >> 4) + 16 * i; }
            else { nn = tid & 63; kk = (tid >> 6) + 4 * i; }
            Bs[buf][kk][nn] = rb[i];
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
    const int nk = k_end > k_begin ? (k_end - k_begin + GBK - 1) / GBK : 0;  // an empty slice writes zeros
    gload(k_begin);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(k_begin + (kt + 1) * GBK);
#pragma unroll
        for (int k2 = 0; k2 < GBK / 2; k2++) {
            const float av = As[buf][2 * k2 + hf][32 * wave + r];
            const float b0 = Bs[buf][2 * k2 + hf][r], b1 = Bs[buf][2 * k2 + hf][32 + r];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);  // the other buffer: its last readers passed the barrier of the previous iteration
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = n0 + 32 * j + r;
        if (n >= a.N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int m = m0 + 32 * wave + acc_row(e, lane);
            if (m < a.M) {
                float* p = C + (size_t)m * a.ldc + n;
                float v = a.alpha * acc[j][e] + bv;
                if (a.accumulate) v += *p;
                *p = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, 8 rows per 256-thread block (32 rows left 216 blocks for 256 CUs at B = 32: 51 us).  y = (x - mu) rstd g + b:
//   dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)) [+ add],  dgamma = sum_rows dy xhat,  dbeta = sum_rows dy
constexpr int LN_ROWS = 8, LN_MAXV = 16;  // D <= 64 * LN_MAXV
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(dig_layernorm_bwd_args a) {
    __shared__ float red[4][2][256];  // one 256-column slice of the four waves' partial sums at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = (a.D + 63) / 64;
    float dg[LN_MAXV], db[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) dg[i] = db[i] = 0.f;
    const long long row0 = (long long)blockIdx.x * LN_ROWS;
    for (int rr = wave; rr < LN_ROWS; rr += 4) {  // fixed row order per wave
        const long long row = row0 + rr;
        if (row >= a.rows) break;
        const float* x = a.x + (size_t)row * a.D;
        const float* dy = a.dy + (size_t)row * a.D;
        float xv[LN_MAXV], gv[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            xv[i] = (i < nv && c < a.D) ? x[c] : 0.f;
            s += xv[i];
        }
        const float mu = wave_sum(s) / (float)a.D;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            const float d = (i < nv && c < a.D) ? xv[i] - mu : 0.f;
            v += d * d;
        }
        const float rstd = rsqrtf(wave_sum(v) / (float)a.D + a.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            const bool ok = i < nv && c < a.D;
            const float xh = ok ? (xv[i] - mu) * rstd : 0.f;
            const float d = ok ? dy[c] : 0.f;
            const float g = ok ? a.gamma[c] * d : 0.f;
            dg[i] += d * xh;
            db[i] += d;
            s1 += g;
            s2 += g * xh;
            xv[i] = xh;
            gv[i] = g;
        }
        s1 = wave_sum(s1) / (float)a.D;
        s2 = wave_sum(s2) / (float)a.D;
        float* dx = a.dx + (size_t)row * a.D;
        const float* add = a.add ? a.add + (size_t)row * a.D : nullptr;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            if (i < nv && c < a.D) {
                float o = rstd * (gv[i] - s1 - xv[i] * s2);
                if (add) o += add[c];
                dx[c] = o;
            }
        }
    }
    // the four waves' partial sums, added in wave order; red is re-used per 256-column slice
    float* part = a.part + (size_t)blockIdx.x * 2 * a.D;
#pragma unroll
    for (int q = 0; q < LN_MAXV / 4; q++) {
        if (q * 4 >= nv) break;  // uniform
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            red[wave][0][lane + 64 * i] = dg[4 * q + i];
            red[wave][1][lane + 64 * i] = db[4 * q + i];
        }
        __syncthreads();
        const int c = q * 256 + threadIdx.x;
        if (c < a.D) {
            part[c] = ((red[0][0][threadIdx.x] + red[1][0][threadIdx.x]) + red[2][0][threadIdx.x]) + red[3][0][threadIdx.x];
            part[a.D + c] = ((red[0][1][threadIdx.x] + red[1][1][threadIdx.x]) + red[2][1][threadIdx.x]) + red[3][1][threadIdx.x];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// column sums: stage 1 sums chunks of CS_ROWS rows (4 row lanes x 64 columns per block, fixed order), stage 2 the chunks
constexpr int CS_ROWS = 128;
// (direct: a single chunk - the split-K reduction, the LayerNorm partials of a small batch - goes straight to `out`)
__global__ __launch_bounds__(256) void colsum1_kernel(const float* x, float* ws, long long M, int N, long long ld, float* out, float scale, int accumulate) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    float s = 0.f;
    if (col < N)
        for (long long rw = r0 + rl; rw < r1; rw += 4) s += x[(size_t)rw * ld + col];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < N) {
        const float t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        if (out) out[col] = accumulate ? out[col] + t * scale : t * scale;
        else ws[(size_t)blockIdx.y * N + col] = t;
    }
}
__global__ __launch_bounds__(256) void colsum2_kernel(const float* ws, float* out, int chunks, int N, float scale, int accumulate) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    float s = 0.f;
    for (int c = 0; c < chunks; c++) s += ws[(size_t)c * N + col];
    s *= scale;
    out[col] = accumulate ? out[col] + s : s;
}

// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* x, const float* dy, float* dx, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        dx[i] = dy[i] * (cdf + v * pdf);
    }
}

// softmax backward on rows of n: ds = scale * p * (dp - sum_j p_j dp_j), in place over dp; one wave per row.  g (optional,
// [B][N][n]): the gradient that reaches EVERY head's probabilities through the head-averaged roll-up (mctformer.py:302-315),
// added to dp first; row = (b * heads + h) * N + i reads g row b * N + i.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(dig_softmax_bwd_args a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int n = a.n;
    const float* pr = a.p + (size_t)row * n;
    float* dr = a.dp + (size_t)row * n;
    const float* gr = nullptr;
    if (a.g) {
        const long long per_img = (long long)a.heads * a.N;
        gr = a.g + (size_t)((row / per_img) * a.N + row % a.N) * n;
    }
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += pr[j] * (gr ? dr[j] + gr[j] : dr[j]);
    s = wave_sum(s);
    for (int j = lane; j < n; j += 64) dr[j] = a.scale * pr[j] * ((gr ? dr[j] + gr[j] : dr[j]) - s);
}

// out[b][i] (+)= scale * sum_h p[b][h][i]: the head mean / layer sum of the attention probabilities (mctformer.py:302-303,310,315)
__global__ __launch_bounds__(256) void head_sum_kernel(dig_head_sum_args a) {
    const long long total = (long long)a.B * a.n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / a.n, r = i - b * a.n;
        const float* p = a.p + (size_t)b * a.H * a.n + r;
        float s = 0.f;
        for (int h = 0; h < a.H; h++) s += p[(size_t)h * a.n];
        s *= a.scale;
        a.out[i] = a.accumulate ? a.out[i] + s : s;
    }
}

// adjoint of  | (max_norm(a) * (a_times_label ? label : 1) - max_norm(b)) * label |  summed over the map, w.r.t. b (same map
// size on both sides).  max_norm(x) = relu(relu(x) - min - 1e-5) / (max - min + 1e-5), min / max over the map of relu(x)
// (model.py:572-578): the gradient also reaches the positions that ARE the maximum and the minimum, as autograd's does.
// One block per (image, class); db is written in full.
__global__ __launch_bounds__(256) void maxnorm_l1_bwd_kernel(dig_maxnorm_l1_bwd_args a) {
    extern __shared__ float lsm[];
    __shared__ float red[32];
    __shared__ int redi[8];
    const int P = a.P, bc = blockIdx.x, tid = threadIdx.x;
    float* am = lsm;       // relu(a)
    float* bm = am + P;    // relu(b)
    float* gm = bm + P;    // d loss / d max_norm(b)
    const float* ap = a.a + (size_t)bc * P;
    const float* bp = a.b + (size_t)bc * P;
    float amax = -INFINITY, amin = INFINITY, bmax = -INFINITY, bmin = INFINITY;
    for (int i = tid; i < P; i += 256) {
        const float av = fmaxf(ap[i], 0.f), bv = fmaxf(bp[i], 0.f);
        am[i] = av;
        bm[i] = bv;
        amax = fmaxf(amax, av);
        amin = fminf(amin, av);
        bmax = fmaxf(bmax, bv);
        bmin = fminf(bmin, bv);
    }
    amax = block_max(amax, red);
    amin = -block_max(-amin, red);
    bmax = block_max(bmax, red);
    bmin = -block_max(-bmin, red);
    // first position of the maximum / minimum (torch.max / torch.min over a dimension return one index each)
    int imax = P, imin = P;
    for (int i = tid; i < P; i += 256) {
        if (bm[i] == bmax) imax = min(imax, i);
        if (bm[i] == bmin) imin = min(imin, i);
    }
    auto block_min_i = [&](int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
        __syncthreads();
        if ((tid & 63) == 0) redi[tid >> 6] = v;
        __syncthreads();
        return min(min(redi[0], redi[1]), min(redi[2], redi[3]));
    };
    imax = block_min_i(imax);
    imin = block_min_i(imin);
    const float lab = a.label[bc], la = a.a_times_label ? lab : 1.f;
    const float aden = amax - amin + 1e-5f, den = bmax - bmin + 1e-5f;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < P; i += 256) {
        const float an = fmaxf(am[i] - amin - 1e-5f, 0.f) / aden;
        const float u = bm[i] - bmin - 1e-5f;
        const float bn = fmaxf(u, 0.f) / den;
        const float d = (an * la - bn) * lab;
        const float g = -(d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * lab * a.scale;
        const float du = u > 0.f ? g / den : 0.f;
        gm[i] = du;
        s1 += g * fmaxf(u, 0.f);
        s2 += du;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float dden = -s1 / (den * den);
    const float dmx = dden, dmn = -dden - s2;
    float* db = a.db + (size_t)bc * P;
    for (int i = tid; i < P; i += 256) {
        float g = gm[i];
        if (i == imax) g += dmx;
        if (i == imin) g += dmn;
        db[i] = bp[i] > 0.f ? g : 0.f;
    }
}

// adjoint of the CAM aggregation's elementwise part (mctformer.py:306-312: cams = mtatt * relu(head)), and of cams = relu(head)
// alone (mtatt == NULL: outputs['cams'], the CAM-consistency term): for dcams [B][C][P]
//   G[b][c][C + j]  = inv_heads * dcams * relu(head)          (mtatt's share: class-token rows of the roll-up gradient; G optional)
//   dhead[b][j][c] += dcams * (mtatt or 1) * [head > 0]       (NHWC, padded to Cp channels)
__global__ __launch_bounds__(256) void cam_fuse_bwd_kernel(dig_cam_fuse_bwd_args a) {
    const long long total = (long long)a.B * a.C * a.P;
    const int N = a.C + a.P;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i % a.P), c = (int)((i / a.P) % a.C), b = (int)(i / ((long long)a.P * a.C));
        const float d = a.dcams[i];
        const float hv = a.head[((size_t)b * a.P + j) * a.C + c];
        if (a.G) a.G[((size_t)b * N + c) * N + a.C + j] = a.inv_heads * d * fmaxf(hv, 0.f);
        if (hv > 0.f) a.dhead[((size_t)b * a.P + j) * a.Cp + c] += d * (a.mtatt ? a.mtatt[i] : 1.f);
    }
}

// d/dx of scale * F.multilabel_soft_margin_loss(x, y): scale * (sigmoid(x) - y) / (B C)
__global__ __launch_bounds__(256) void msm_grad_kernel(dig_msm_grad_args a) {
    const int n = a.B * a.C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = a.x[i];
    const float sg = x >= 0.f ? 1.0f / (1.0f + expf(-x)) : expf(x) / (1.0f + expf(x));
    a.dx[i] = a.scale * (sg - a.y[i]) / (float)n;
}

// adjoint of head_pool: dpooled[b][c] / k goes to the k largest positive and the k largest negative-magnitude positions of
// channel c (mctformer.py:300: topk(relu(x)) - topk(relu(-x)); ReLU passes no gradient at 0).  out [B][P][Cp] is written
// in full: zero elsewhere and in the padding channels C..Cp-1.  One wave per (b, c), c < Cp.
struct Top3i {
    float v[3];
    int i[3];
    __device__ __forceinline__ void init() {
        v[0] = v[1] = v[2] = -INFINITY;
        i[0] = i[1] = i[2] = -1;
    }
    __device__ __forceinline__ void push(float x, int ix) {
        if (x > v[0]) { v[2] = v[1]; i[2] = i[1]; v[1] = v[0]; i[1] = i[0]; v[0] = x; i[0] = ix; }
        else if (x > v[1]) { v[2] = v[1]; i[2] = i[1]; v[1] = x; i[1] = ix; }
        else if (x > v[2]) { v[2] = x; i[2] = ix; }
    }
    __device__ __forceinline__ void pop() { v[0] = v[1]; i[0] = i[1]; v[1] = v[2]; i[1] = i[2]; v[2] = -INFINITY; i[2] = -1; }
};
__global__ __launch_bounds__(256) void head_pool_bwd_kernel(dig_head_pool_bwd_args a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)a.B * a.Cp) return;
    const int b = (int)(row / a.Cp), c = (int)(row % a.Cp);
    float* out = a.dhead + (size_t)b * a.P * a.Cp + c;
    for (int p = lane; p < a.P; p += 64) out[(size_t)p * a.Cp] = 0.f;
    if (c >= a.C) return;
    const float* in = a.head + (size_t)b * a.P * a.C + c;
    Top3i pos, neg;
    pos.init();
    neg.init();
    for (int p = lane; p < a.P; p += 64) {
        const float v = in[(size_t)p * a.C];
        pos.push(fmaxf(v, 0.f), p);
        neg.push(fmaxf(-v, 0.f), p);
    }
    const int k = min(3, a.P);
    const float g = a.dpooled[(size_t)b * a.C + c] / (float)k;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the zero fill above precedes the scattered stores below
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int side = 0; side < 2; side++) {
        Top3i& t = side ? neg : pos;
        for (int i = 0; i < k; i++) {
            const float m = wave_max(t.v[0]);
            const unsigned long long owners = __ballot(t.v[0] == m);
            if (lane == (int)__builtin_ctzll(owners)) {
                if (m > 0.f) out[(size_t)t.i[0] * a.Cp] = g;  // a position is positive or negative, never both: no overlap between the sides
                t.pop();
            }
        }
    }
}

// torch.optim.AdamW, one parameter tensor: p *= 1 - lr wd; m, v updated; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(dig_adamw_args a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        const float g = a.g[i];
        float p = a.p[i] * (1.0f - a.lr * a.wd);
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const float denom = sqrtf(v) / a.sqrt_bc2 + a.eps;
        a.p[i] = p - a.step_size * (m / denom);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// adjoints used by the LFCA branch (mctformer.py:56-166; resnet101.py:31-38,195-222)
// GroupNorm(1, C) backward on NHWC x [B][HW][C]: one 1024-thread block per image (the statistics span the whole image):
//   dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)),  part[b][0][c] = sum_hw dy xhat,  part[b][1][c] = sum_hw dy
__global__ __launch_bounds__(1024) void gn1_bwd_kernel(dig_gn1_bwd_args a) {
    __shared__ float red[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * a.HW * a.C;
    const float* x = a.x + base;
    const float* dy = a.dy + base;
    const float n = (float)a.HW * (float)a.C;
    float s = 0.f;
    for (int c = tid; c < a.C; c += 1024)
        for (int p = 0; p < a.HW; p++) s += x[(size_t)p * a.C + c];
    const float mu = block_sum(s, red) / n;
    float v = 0.f;
    for (int c = tid; c < a.C; c += 1024)
        for (int p = 0; p < a.HW; p++) {
            const float d = x[(size_t)p * a.C + c] - mu;
            v += d * d;
        }
    const float rstd = rsqrtf(block_sum(v, red) / n + a.eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = tid; c < a.C; c += 1024) {
        const float g = a.gamma[c];
        float dg = 0.f, db = 0.f;
        for (int p = 0; p < a.HW; p++) {
            const float d = dy[(size_t)p * a.C + c], xh = (x[(size_t)p * a.C + c] - mu) * rstd;
            dg += d * xh;
            db += d;
        }
        a.part[((size_t)b * 2) * a.C + c] = dg;
        a.part[((size_t)b * 2 + 1) * a.C + c] = db;
        s1 += g * db;
        s2 += g * dg;
    }
    s1 = block_sum(s1, red) / n;
    s2 = block_sum(s2, red) / n;
    float* dx = a.dx + base;
    for (int c = tid; c < a.C; c += 1024) {
        const float g = a.gamma[c];
        for (int p = 0; p < a.HW; p++) {
            const size_t i = (size_t)p * a.C + c;
            dx[i] = rstd * (g * dy[i] - s1 - (x[i] - mu) * rstd * s2);
        }
    }
}

// dx = dy * [y > 0] (y = the ReLU's output or input: same mask); dx may alias dy
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* y, const float* dy, float* dx, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// d/d(student) of scale * KLDivLoss(batchmean, log_target)(log cat(S, 1-S), log cat(Tt, 1-Tt)), S = softmax(student / T),
// Tt = softmax(teacher / T) (model.py:347-361; the teacher is detached): with a_c = -Tt_c / S_c + (1 - Tt_c) / (1 - S_c),
//   dz_j = scale / (B T) * S_j (a_j - sum_c S_c a_c).  One wave per row.
__global__ __launch_bounds__(256) void kd_bin_grad_kernel(dig_kd_bin_grad_args a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const float* s = a.student + (size_t)b * a.C;
    const float* t = a.teacher + (size_t)b * a.C;
    float ms = -INFINITY, mt = -INFINITY;
    for (int c = lane; c < a.C; c += 64) {
        ms = fmaxf(ms, s[c] / a.T);
        mt = fmaxf(mt, t[c] / a.T);
    }
    ms = wave_max(ms);
    mt = wave_max(mt);
    float zs = 0.f, zt = 0.f;
    for (int c = lane; c < a.C; c += 64) {
        zs += expf(s[c] / a.T - ms);
        zt += expf(t[c] / a.T - mt);
    }
    zs = wave_sum(zs);
    zt = wave_sum(zt);
    float dot = 0.f;
    for (int c = lane; c < a.C; c += 64) {
        const float ps = expf(s[c] / a.T - ms) / zs, pt = expf(t[c] / a.T - mt) / zt;
        dot += ps * (-pt / ps + (1.f - pt) / (1.f - ps));
    }
    dot = wave_sum(dot);
    for (int c = lane; c < a.C; c += 64) {
        const float ps = expf(s[c] / a.T - ms) / zs, pt = expf(t[c] / a.T - mt) / zt;
        const float g = a.scale / ((float)a.B * a.T) * ps * ((-pt / ps + (1.f - pt) / (1.f - ps)) - dot);
        a.dx[(size_t)b * a.C + c] = a.accumulate ? a.dx[(size_t)b * a.C + c] + g : g;
    }
}

}  // namespace

extern "C" int dig_gemm_f32(const dig_gemm_f32_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->a && a->b && a->c, DIG_EINVAL, "gemm_f32: null pointer");
    DIG_CHECK(a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0 && a->z_inner > 0 && a->batch % a->z_inner == 0 && a->ldc >= a->N, DIG_EINVAL,
              "gemm_f32: bad shape M=%d N=%d K=%d batch=%d z_inner=%d ldc=%d", a->M, a->N, a->K, a->batch, a->z_inner, a->ldc);
    const int ks = a->ksplit > 1 ? a->ksplit : 1;
    DIG_CHECK((long long)a->batch * ks <= 65535, DIG_EINVAL, "gemm_f32: batch %d x ksplit %d > 65535", a->batch, ks);
    DIG_CHECK(ks == 1 || (!a->bias && !a->accumulate && a->sc_k >= (long long)a->M * a->ldc), DIG_EINVAL,
              "gemm_f32: split-K writes plain partial outputs (no bias / accumulate) sc_k >= M*ldc elements apart");
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(a->N, GBN), cdiv(a->M, GBM), a->batch * ks), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("gemm_f32");
    return DIG_OK;
}

extern "C" int dig_layernorm_bwd_blocks(long long rows) { return (int)((rows + LN_ROWS - 1) / LN_ROWS); }

extern "C" int dig_layernorm_bwd(const dig_layernorm_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->gamma && a->dy && a->dx && a->part, DIG_EINVAL, "layernorm_bwd: null pointer");
    DIG_CHECK(a->rows > 0 && a->D > 0 && a->D <= 64 * LN_MAXV, DIG_EINVAL, "layernorm_bwd: bad shape rows=%lld D=%d (D <= %d)", a->rows, a->D, 64 * LN_MAXV);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)dig_layernorm_bwd_blocks(a->rows)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("layernorm_bwd");
    return DIG_OK;
}

extern "C" int dig_colsum_chunks(long long M) { return (int)((M + CS_ROWS - 1) / CS_ROWS); }

extern "C" int dig_colsum(const dig_colsum_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->ws && a->out, DIG_EINVAL, "colsum: null pointer");
    DIG_CHECK(a->M > 0 && a->N > 0 && a->ld >= a->N, DIG_EINVAL, "colsum: bad shape M=%lld N=%d ld=%lld", a->M, a->N, a->ld);
    const int chunks = dig_colsum_chunks(a->M);
    DIG_CHECK(chunks <= 65535, DIG_EINVAL, "colsum: %d row chunks > 65535", chunks);
    if (chunks == 1) {
        hipLaunchKernelGGL(colsum1_kernel, dim3(cdiv(a->N, 64), 1), dim3(256), 0, stream, a->x, a->ws, a->M, a->N, a->ld, a->out, a->scale, a->accumulate);
    } else {
        hipLaunchKernelGGL(colsum1_kernel, dim3(cdiv(a->N, 64), chunks), dim3(256), 0, stream, a->x, a->ws, a->M, a->N, a->ld, (float*)nullptr, 1.0f, 0);
        hipLaunchKernelGGL(colsum2_kernel, dim3(cdiv(a->N, 256)), dim3(256), 0, stream, (const float*)a->ws, a->out, chunks, a->N, a->scale, a->accumulate);
    }
    DIG_LAUNCH_CHECK("colsum");
    return DIG_OK;
}

extern "C" int dig_gelu_bwd(const dig_gelu_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->dy && a->dx && a->n > 0, DIG_EINVAL, "gelu_bwd: bad argument");
    const long long blocks = (a->n + 255) / 256;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, a->x, a->dy, a->dx, a->n);
    DIG_LAUNCH_CHECK("gelu_bwd");
    return DIG_OK;
}

extern "C" int dig_softmax_bwd(const dig_softmax_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->p && a->dp && a->rows > 0 && a->n > 0, DIG_EINVAL, "softmax_bwd: bad argument");
    const long long blocks = (a->rows + 3) / 4;
    DIG_CHECK(blocks < (1ll << 31), DIG_EINVAL, "softmax_bwd: too many rows");
    DIG_CHECK(!a->g || (a->heads > 0 && a->N > 0 && a->rows % ((long long)a->heads * a->N) == 0), DIG_EINVAL, "softmax_bwd: the roll-up gradient needs heads, N with rows = B * heads * N");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("softmax_bwd");
    return DIG_OK;
}

extern "C" int dig_msm_grad(const dig_msm_grad_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->y && a->dx && a->B > 0 && a->C > 0, DIG_EINVAL, "msm_grad: bad argument");
    hipLaunchKernelGGL(msm_grad_kernel, dim3(cdiv(a->B * a->C, 256)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("msm_grad");
    return DIG_OK;
}

extern "C" int dig_head_pool_bwd(const dig_head_pool_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->head && a->dpooled && a->dhead, DIG_EINVAL, "head_pool_bwd: null pointer");
    DIG_CHECK(a->B > 0 && a->P > 0 && a->C > 0 && a->Cp >= a->C, DIG_EINVAL, "head_pool_bwd: bad shape B=%d P=%d C=%d Cp=%d", a->B, a->P, a->C, a->Cp);
    hipLaunchKernelGGL(head_pool_bwd_kernel, dim3((unsigned)(((long long)a->B * a->Cp + 3) / 4)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("head_pool_bwd");
    return DIG_OK;
}

extern "C" int dig_adamw(const dig_adamw_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->p && a->g && a->m && a->v && a->n > 0, DIG_EINVAL, "adamw: bad argument");
    DIG_CHECK(a->sqrt_bc2 > 0.f && a->eps >= 0.f, DIG_EINVAL, "adamw: bad bias correction / eps");
    const long long blocks = (a->n + 255) / 256;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("adamw");
    return DIG_OK;
}

extern "C" int dig_head_sum(const dig_head_sum_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->p && a->out && a->B > 0 && a->H > 0 && a->n > 0, DIG_EINVAL, "head_sum: bad argument");
    const long long blocks = ((long long)a->B * a->n + 255) / 256;
    hipLaunchKernelGGL(head_sum_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("head_sum");
    return DIG_OK;
}

extern "C" int dig_maxnorm_l1_bwd(const dig_maxnorm_l1_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->a && a->b && a->label && a->db, DIG_EINVAL, "maxnorm_l1_bwd: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->P > 0 && (size_t)a->P * 12 <= 60 * 1024, DIG_EINVAL, "maxnorm_l1_bwd: bad shape B=%d C=%d P=%d (P <= 5120)", a->B, a->C, a->P);
    hipLaunchKernelGGL(maxnorm_l1_bwd_kernel, dim3((unsigned)(a->B * a->C)), dim3(256), (size_t)a->P * 12, stream, *a);
    DIG_LAUNCH_CHECK("maxnorm_l1_bwd");
    return DIG_OK;
}

extern "C" int dig_cam_fuse_bwd(const dig_cam_fuse_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->dcams && a->head && a->dhead, DIG_EINVAL, "cam_fuse_bwd: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->P > 0 && a->Cp >= a->C && (!a->G || a->inv_heads > 0.f), DIG_EINVAL, "cam_fuse_bwd: bad argument");
    const long long blocks = ((long long)a->B * a->C * a->P + 255) / 256;
    hipLaunchKernelGGL(cam_fuse_bwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("cam_fuse_bwd");
    return DIG_OK;
}

extern "C" int dig_gn1_bwd(const dig_gn1_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->gamma && a->dy && a->dx && a->part, DIG_EINVAL, "gn1_bwd: null pointer");
    DIG_CHECK(a->B > 0 && a->HW > 0 && a->C > 0 && a->eps > 0.f, DIG_EINVAL, "gn1_bwd: bad shape B=%d HW=%d C=%d", a->B, a->HW, a->C);
    hipLaunchKernelGGL(gn1_bwd_kernel, dim3(a->B), dim3(1024), 0, stream, *a);
    DIG_LAUNCH_CHECK("gn1_bwd");
    return DIG_OK;
}

extern "C" int dig_relu_bwd(const dig_relu_bwd_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->y && a->dy && a->dx && a->n > 0, DIG_EINVAL, "relu_bwd: bad argument");
    const long long blocks = (a->n + 255) / 256;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, a->y, a->dy, a->dx, a->n);
    DIG_LAUNCH_CHECK("relu_bwd");
    return DIG_OK;
}

extern "C" int dig_kd_bin_grad(const dig_kd_bin_grad_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->student && a->teacher && a->dx && a->B > 0 && a->C > 0 && a->T > 0.f, DIG_EINVAL, "kd_bin_grad: bad argument");
    hipLaunchKernelGGL(kd_bin_grad_kernel, dim3((a->B + 3) / 4), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("kd_bin_grad");
    return DIG_OK;
}
