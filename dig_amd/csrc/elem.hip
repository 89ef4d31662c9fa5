// HBM-bound helper kernels of the U-Net path (gfx950): GroupNorm finalize / apply, RMSNorm, softmaxes,
// schedule arithmetic, sinusoidal embedding, activations, bilinear(align_corners=True)+concat, layout converters.
// All global accesses on NHWC tensors are 16-byte vectors (8 x bf16 / 2 x 4 x f32), coalesced along C.
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// GroupNorm finalize: one 64-thread block per (image, group).  Partials are summed in a fixed order in
// fp64 so the result is deterministic and free of E[x^2]-mean^2 cancellation at fp32.
// One wave finalizes group g of image b; dst receives (a, b) pairs at dst[2 * c] for the channels c of the group (c counted
// from 0 of the image: the separate launch points dst at the image's row of the table, the fused kernel at its LDS copy).
__device__ __forceinline__ void gn_finalize_group(const dig_gn_finalize_args& a, int b, int g, int lane, float* dst) {
    const int cpg = a.C / a.groups;
    const float* st = a.stats + ((size_t)b * a.groups + g) * a.tiles * 2;
    // the first channel's parameters are fetched beside the partials, not after the reduction (the kernel is one chain of
    // dependent L2 round trips: partials -> sums -> parameters -> store)
    const int c0 = g * cpg + min(lane, cpg - 1);
    const float ga0 = a.gamma[c0], be0 = a.beta[c0];
    float sc0 = 1.0f, sh0 = 0.0f;
    if (a.ss) {
        const float* row = a.ss + (size_t)b * a.ss_ld + a.ss_off;
        sc0 = row[c0] + 1.0f;
        sh0 = row[a.C + c0];
    }
    double s = 0.0, ss = 0.0;
    // 8 partials per round trip (all loads issued before the first add; the adds keep their order, so the sums are the same
    // bits as the plain loop's): at 256x256 the weight-stationary kernel leaves 1024 partials per (image, group)
    for (int i0 = lane; i0 < a.tiles; i0 += 64 * 8) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + 64 * u;
            v[u] = i < a.tiles ? *reinterpret_cast<const f32x2*>(st + 2 * i) : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (i0 + 64 * u < a.tiles) {
                s += (double)v[u][0];
                ss += (double)v[u][1];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        ss += __shfl_xor(ss, o, 64);
    }
    const double n = (double)cpg * (double)a.hw;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)mean;
    for (int i = lane; i < cpg; i += 64) {
        const int c = g * cpg + i;
        const bool first = i == lane;
        float ga = (first ? ga0 : a.gamma[c]) * rstd;
        float be = (first ? be0 : a.beta[c]) - meanf * ga;
        if (a.ss) {
            const float* row = a.ss + (size_t)b * a.ss_ld + a.ss_off;
            const float sc = first ? sc0 : row[c] + 1.0f, sh = first ? sh0 : row[a.C + c];
            ga *= sc;
            be = be * sc + sh;
        }
        dst[2 * c] = ga;
        dst[2 * c + 1] = be;
    }
}

__global__ void gn_finalize_kernel(dig_gn_finalize_args a) {
    const int b = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
    gn_finalize_group(a, b, g, threadIdx.x, a.aff + (size_t)b * a.C * 2);
}

// finalize + apply in one launch: block = (image, chunk of its HW * C / 8 vectors).  The four waves finalize the image's groups
// into LDS (gn_finalize_group: the table's bits are those of the separate launch), then the block streams its chunk.
constexpr int GN_FIN_MAXC = 1024, GN_FIN_MAX_TILES = 64;
template <typename T> __global__ __launch_bounds__(256) void gn_fin_silu_add_kernel(dig_gn_fin_silu_add_args a, int chunks) {
    __shared__ float saff[2 * GN_FIN_MAXC];
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    dig_gn_finalize_args fa;
    fa.stats = a.stats; fa.gamma = a.gamma; fa.beta = a.beta; fa.ss = a.ss; fa.aff = nullptr;
    fa.B = a.B; fa.C = a.C; fa.groups = a.groups; fa.tiles = a.tiles; fa.hw = a.hw; fa.ss_ld = a.ss_ld; fa.ss_off = a.ss_off; fa.eps = a.eps;
    for (int g = threadIdx.x >> 6; g < a.groups; g += 4) gn_finalize_group(fa, b, g, threadIdx.x & 63, saff);
    __syncthreads();
    const int cv = a.C / 8;
    const size_t per_img = (size_t)a.HW * cv;
    const size_t per = (per_img + chunks - 1) / chunks;
    const size_t v0 = (size_t)ch * per, v1 = v0 + per < per_img ? v0 + per : per_img;
    const T* x = (const T*)a.x + (size_t)b * per_img * 8;
    const T* res = a.res ? (const T*)a.res + (size_t)b * per_img * 8 : nullptr;
    T* out = (T*)a.out + (size_t)b * per_img * 8;
    for (size_t v = v0 + threadIdx.x; v < v1; v += 256) {
        const int c = (int)(v % cv) * 8;
        const float* af = saff + 2 * c;
        Frag<T> f, r;
        frag_load<T>(f, x + v * 8);
        if (res) frag_load<T>(r, res + v * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float y = silu_f(af[2 * j] * frag_get<T>(f, j) + af[2 * j + 1]);
            if (res) y += frag_get<T>(r, j);
            frag_set<T>(f, j, y);
        }
        frag_store<T>(f, out + v * 8);
    }
}

// out = silu(a*x+b) [+ res], 8 elements per thread-iteration
template <typename T> __global__ void gn_silu_add_kernel(dig_gn_silu_add_args a) {
    const int cv = a.C / 8;
    const size_t per_img = (size_t)a.HW * cv, total = per_img * a.B;
    const T* x = (const T*)a.x;
    const T* res = (const T*)a.res;
    T* out = (T*)a.out;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(v / per_img);
        const int c = (int)(v % cv) * 8;
        const float* af = a.aff + ((size_t)b * a.C + c) * 2;
        Frag<T> f, r;
        frag_load<T>(f, x + v * 8);
        if (res) frag_load<T>(r, res + v * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float y = silu_f(af[2 * j] * frag_get<T>(f, j) + af[2 * j + 1]);
            if (res) y += frag_get<T>(r, j);
            frag_set<T>(f, j, y);
        }
        frag_store<T>(f, out + v * 8);
    }
}

// RMSNorm: G = lanes cooperating on one row (C/8 vectors per row handled as ceil(C/8/G) vectors per lane)
template <typename T, int G> __global__ void rmsnorm_kernel(dig_rmsnorm_args a) {
    const int cv = a.C / 8;
    const int rows_per_block = blockDim.x / G;
    const int sub = threadIdx.x % G;
    const T* x = (const T*)a.x;
    const T* res = (const T*)a.res;
    T* out = (T*)a.out;
    const float mult = sqrtf((float)a.C);
    // all G lanes of a group share `row`, so a group enters/leaves the loop together and the shuffles below
    // only ever exchange data between lanes that are active.
    for (long long row = (long long)blockIdx.x * rows_per_block + threadIdx.x / G; row < a.rows;
         row += (long long)gridDim.x * rows_per_block) {
        float ssq = 0.f;
        for (int v = sub; v < cv; v += G) {
            Frag<T> f;
            frag_load<T>(f, x + (size_t)row * a.C + v * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float t = frag_get<T>(f, j);
                ssq += t * t;
            }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) ssq += __shfl_xor(ssq, o, 64);
        const float inv = mult / fmaxf(sqrtf(ssq), 1e-12f);
        for (int v = sub; v < cv; v += G) {
            Frag<T> f, r;
            frag_load<T>(f, x + (size_t)row * a.C + v * 8);
            if (res) frag_load<T>(r, res + (size_t)row * a.C + v * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float y = frag_get<T>(f, j) * inv * a.g[v * 8 + j];
                if (res) y += frag_get<T>(r, j);
                frag_set<T>(f, j, y);
            }
            frag_store<T>(f, out + (size_t)row * a.C + v * 8);
        }
    }
}

// softmax over contiguous rows; one block per row.  Rows of up to 256*RPT elements are held in registers, so the
// row is read from HBM exactly once (fp32 logits in, T probabilities out); longer rows fall back to three passes.
template <typename TI, typename TO, int RPT> __global__ __launch_bounds__(256) void row_softmax_kernel(const TI* in, TO* out, int n) {
    __shared__ float red[32];
    const TI* x = in + (size_t)blockIdx.x * n;
    TO* y = out + (size_t)blockIdx.x * n;
    if (n <= 256 * RPT) {
        float v[RPT];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = threadIdx.x + k * 256;
            v[k] = i < n ? (float)x[i] : -INFINITY;
            m = fmaxf(m, v[k]);
        }
        m = block_max(m, red);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            v[k] = __expf(v[k] - m);
            s += v[k];
        }
        s = block_sum(s, red);
        const float inv = 1.0f / s;
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = threadIdx.x + k * 256;
            if (i < n) y[i] = (TO)(v[k] * inv);
        }
        return;
    }
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, (float)x[i]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += __expf((float)x[i] - m);
    s = block_sum(s, red);
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] = (TO)(__expf((float)x[i] - m) * inv);
}

// q softmax over each 32-channel head slice, times scale; one thread per (row, head)
template <typename T> __global__ void group_softmax32_kernel(dig_group_softmax32_args a) {
    T* q = (T*)a.q;
    const long long total = a.rows * a.groups;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        T* p = q + i * 32;
        Frag<T> f[4];
        float m = -INFINITY;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            frag_load<T>(f[v], p + v * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) m = fmaxf(m, frag_get<T>(f[v], j));
        }
        float e[32], s = 0.f;
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                e[v * 8 + j] = __expf(frag_get<T>(f[v], j) - m);
                s += e[v * 8 + j];
            }
        const float inv = a.scale / s;
#pragma unroll
        for (int v = 0; v < 4; v++) {
#pragma unroll
            for (int j = 0; j < 8; j++) frag_set<T>(f[v], j, e[v * 8 + j] * inv);
            frag_store<T>(f[v], p + v * 8);
        }
    }
}

template <typename T> __global__ void row_max_kernel(dig_row_max_args a) {
    __shared__ float red[32];
    const long long grp = blockIdx.x / a.group_rows, sub = blockIdx.x % a.group_rows;
    const T* x = (const T*)a.in + grp * a.group_stride + sub * (long long)a.n;
    float m = -INFINITY;
    const int nv = a.n / 8;
    for (int v = threadIdx.x; v < nv; v += blockDim.x) {
        Frag<T> f;
        frag_load<T>(f, x + v * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) m = fmaxf(m, frag_get<T>(f, j));
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) a.out[blockIdx.x] = m;
}

// out = post_scale[c]*(ca[t[b]]*x + sign*cs[t[b]]*y) + post_shift[c]   (NCHW fp32)
__global__ void sched_axpby_kernel(dig_sched_axpby_args a) {
    const size_t total = (size_t)a.B * 3 * a.HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / ((size_t)3 * a.HW));
        const int c = (int)((i / a.HW) % 3);
        const long long t = a.t[b];
        // two rounded products and a rounded sum, as the reference's tensor expression (ddp.py:857-860, 630-634) and the stem kernels
        const float v = axpby_rn(a.sqrt_ac[t], a.x[i], a.sign * a.sqrt_1mac[t], a.y[i]);
        a.out[i] = a.post_scale[c] * v + a.post_shift[c];
    }
}

// one DDIM update for objective pred_v (see dig_hip.h); mul_rn keeps each product a separately rounded fp32 value
__device__ __forceinline__ float mul_rn(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__global__ void ddim_step_kernel(dig_ddim_step_args a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)a.n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = a.x[i];
        float x0 = axpby_rn(a.sqrt_ac, x, -a.sqrt_1mac, a.v[i]);
        x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float o = x0;
        if (!a.last) {
            const float eps = (mul_rn(a.sqrt_recip, x) - x0) / a.sqrt_recipm1;
            float s = mul_rn(x0, a.sqrt_alpha_next) + mul_rn(a.c, eps);
            asm volatile("" : "+v"(s));
            o = a.noise ? s + mul_rn(a.sigma, a.noise[i]) : s;
        }
        a.out[i] = o;
        if (a.out_unnorm) a.out_unnorm[i] = (o + 1.f) * 0.5f;
    }
}

template <typename T> __global__ void sinusoid_kernel(dig_sinusoid_args a) {
    const int half = a.dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.dim) return;
    const int b = i / a.dim, j = i % a.dim;
    const int fi = j < half ? j : j - half;
    // same op order as the reference: exp(arange * -(ln(1e4)/(half-1))) then t * freq (ddp.py:126-128)
    const float freq = expf((float)fi * -(logf(10000.0f) / (float)(half - 1)));
    const float ang = (float)a.t[b] * freq;
    ((T*)a.out)[i] = from_f32<T>(j < half ? sinf(ang) : cosf(ang));
}

// Fused head of the time embedding (ddp.py:118-130,316-321): sinusoid -> Linear(dim, tdim) -> GELU(erf) -> Linear(tdim, tdim),
// one block per image; writes temb and SiLU(temb) (the input of every block's `mlp`, ddp.py:172-175).  fp32 throughout.
__global__ __launch_bounds__(1024) void time_embed_kernel(dig_time_embed_args a) {
    extern __shared__ float te_sm[];
    float* emb = te_sm;          // [dim]
    float* e1 = te_sm + a.dim;   // [tdim]
    const int b = blockIdx.x, tid = threadIdx.x, half = a.dim / 2;
    for (int j = tid; j < a.dim; j += 1024) {
        const int fi = j < half ? j : j - half;
        const float freq = expf((float)fi * -(logf(10000.0f) / (float)(half - 1)));  // as sinusoid_kernel
        const float ang = (float)a.t[b] * freq;
        emb[j] = j < half ? sinf(ang) : cosf(ang);
    }
    __syncthreads();
    // rows of W are split over G adjacent lanes (16-byte loads, whole rows coalesced), 64 / G rows per wave instruction, and
    // the loads of 8 row groups (x 16 waves: half the rows of a 256-wide layer) are issued before the first is used (a block is one dependent chain: L2 latency is all it waits for)
    const int lane = tid & 63, wave = tid >> 6;
    auto matvec = [&](const float* __restrict__ W, const float* __restrict__ bias, const float* x, int K, auto&& emit) {
        const int G = K / 4 < 64 ? K / 4 : 64, rp = 64 / G, gl = lane % G, gr = lane / G;
        for (int n0 = wave * rp; n0 < a.tdim; n0 += 16 * rp * 8) {
            float acc[8];
#pragma unroll
            for (int u = 0; u < 8; u++) acc[u] = 0.f;
            for (int k = gl * 4; k < K; k += G * 4) {
                f32x4 w4[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w4[u] = *reinterpret_cast<const f32x4*>(W + (size_t)min(n0 + u * 16 * rp + gr, a.tdim - 1) * K + k);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    acc[u] = fmaf(w4[u][0], x[k], acc[u]); acc[u] = fmaf(w4[u][1], x[k + 1], acc[u]);
                    acc[u] = fmaf(w4[u][2], x[k + 2], acc[u]); acc[u] = fmaf(w4[u][3], x[k + 3], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                for (int o = G >> 1; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, 64);
                const int n = n0 + u * 16 * rp + gr;
                if (n < a.tdim && gl == 0) emit(n, acc[u] + bias[n]);
            }
        }
    };
    matvec(a.w1, a.b1, emb, a.dim, [&](int n, float v) { e1[n] = gelu_erf_f(v); });
    __syncthreads();
    float* __restrict__ temb = a.temb + (size_t)b * a.tdim;
    float* __restrict__ st = a.st + (size_t)b * a.tdim;
    matvec(a.w3, a.b3, e1, a.tdim, [&](int n, float v) {
        temb[n] = v;
        st[n] = silu_f(v);
    });
}

// All per-block time projections in one skinny GEMM (ddp.py:172-175,186-188 for the 19 ResnetBlocks at once):
// out[b][n] = bias[n] + sum_k st[b][k] W[n][k], B <= a few hundred rows, N = sum 2*C_out (8064), K = tdim (256).
// One wave per 32 rows x 32 columns, exact fp32 on v_mfma_f32_32x32x2_f32, operands straight from L2 (32-byte runs per lane).
__global__ __launch_bounds__(64) void time_mlp_kernel(dig_time_mlp_args a) {
    const int lane = threadIdx.x, r = lane & 31, hf = lane >> 5;
    const int n0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int row = b0 + r, col = n0 + r;
    const bool rv = row < a.B, cv = col < a.N;
    const float* ap = a.st + (size_t)(rv ? row : 0) * a.K + hf * 8;
    const float* bp = a.w + (size_t)(cv ? col : 0) * a.K + hf * 8;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += 64) {
        Frag<float> fa[4], fb[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            frag_load<float>(fa[s], ap + k0 + s * 16);
            frag_load<float>(fb[s], bp + k0 + s * 16);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (!rv) frag_zero<float>(fa[s]);
            if (!cv) frag_zero<float>(fb[s]);
            mma16<float>(acc, fa[s], fb[s]);
        }
    }
    if (cv) {
        const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int rr = b0 + acc_row(e, lane);
            if (rr < a.B) a.out[(size_t)rr * a.N + col] = acc[e] + bias;
        }
    }
}

template <typename TI, typename TO> __global__ void act_kernel(const TI* in, TO* out, long long n, int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = (float)in[i];
        out[i] = (TO)(act == 0 ? silu_f(x) : gelu_erf_f(x));
    }
}

// bilinear align_corners=True from NHWC T to an NCHW fp32 slice, transposed through LDS: a block owns (image, 64 channels, 128
// consecutive output pixels).  Phase 1: lane = channel, so the four corner reads of a pixel are 128-byte runs (bf16); phase 2:
// lane = output pixel, so the NCHW stores are 256-byte runs per wave.  (The thread-per-element kernel below stored 64 lanes to 64
// different lines: 33 us for the 4.2 M outputs of the 512-channel map.)  Same expression per element as that kernel.
constexpr int BL_C = 64, BL_P = 128, BL_LD = BL_P + 1;
template <typename T> __global__ __launch_bounds__(256) void bilinear_ac_tiled_kernel(dig_bilinear_ac_args a) {
    __shared__ float tile[BL_C * BL_LD];
    const int hw = a.h * a.w;
    const int pchunks = (hw + BL_P - 1) / BL_P, cchunks = (a.C + BL_C - 1) / BL_C;
    int id = blockIdx.x;
    const int pc = id % pchunks; id /= pchunks;
    const int cc = id % cchunks;
    const int b = id / cchunks;
    const int p0 = pc * BL_P, c0 = cc * BL_C;
    const int np = min(BL_P, hw - p0), nc = min(BL_C, a.C - c0);
    const T* in = (const T*)a.in + (size_t)b * a.H * a.W * a.C + c0;
    const float sy = a.h > 1 ? (float)(a.H - 1) / (float)(a.h - 1) : 0.f;
    const float sx = a.w > 1 ? (float)(a.W - 1) / (float)(a.w - 1) : 0.f;
    const int c = threadIdx.x & 63, pg = threadIdx.x >> 6;
    if (c < nc) {
        constexpr int PB = 8;  // pixels per batch: 32 corner loads in flight per lane before the first one is used
        for (int pb = pg; pb < np; pb += 4 * PB) {
            T v[PB][4];
            float wy[PB], wx[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const int p = min(pb + 4 * j, np - 1);  // past the chunk: re-read its last pixel, never stored
                const int op = p0 + p, oy = op / a.w, ox = op - oy * a.w;
                const float fy = sy * oy, fx = sx * ox;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
                wy[j] = fy - y0;
                wx[j] = fx - x0;
                const T* base = in + c;
                v[j][0] = base[((size_t)y0 * a.W + x0) * a.C];
                v[j][1] = base[((size_t)y0 * a.W + x1) * a.C];
                v[j][2] = base[((size_t)y1 * a.W + x0) * a.C];
                v[j][3] = base[((size_t)y1 * a.W + x1) * a.C];
            }
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const int p = pb + 4 * j;
                const float ly = wy[j], lx = wx[j], hy = 1.f - ly, hx = 1.f - lx;
                const float p00 = to_f32<T>(v[j][0]), p01 = to_f32<T>(v[j][1]), p10 = to_f32<T>(v[j][2]), p11 = to_f32<T>(v[j][3]);
                if (p < np) tile[c * BL_LD + p] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
            }
        }
    }
    __syncthreads();
    float* out = a.out + ((size_t)b * a.Ctot + a.c_off + c0) * hw + p0;
    for (int i = threadIdx.x; i < nc * BL_P; i += 256) {
        const int ch = i / BL_P, p = i - ch * BL_P;
        if (p < np) out[(size_t)ch * hw + p] = tile[ch * BL_LD + p];
    }
}

// the same, thread = (b, oy, ox, c), c fastest (kept for DIG_BILINEAR_FLAT, the A/B and bit-equality reference of the tiled kernel)
template <typename T> __global__ void bilinear_ac_kernel(dig_bilinear_ac_args a) {
    const size_t total = (size_t)a.B * a.h * a.w * a.C;
    const T* in = (const T*)a.in;
    const float sy = a.h > 1 ? (float)(a.H - 1) / (float)(a.h - 1) : 0.f;
    const float sx = a.w > 1 ? (float)(a.W - 1) / (float)(a.w - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        size_t rest = i / a.C;
        const int ox = (int)(rest % a.w);
        rest /= a.w;
        const int oy = (int)(rest % a.h);
        const int b = (int)(rest / a.h);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const T* base = in + (size_t)b * a.H * a.W * a.C + c;
        const float p00 = to_f32<T>(base[((size_t)y0 * a.W + x0) * a.C]);
        const float p01 = to_f32<T>(base[((size_t)y0 * a.W + x1) * a.C]);
        const float p10 = to_f32<T>(base[((size_t)y1 * a.W + x0) * a.C]);
        const float p11 = to_f32<T>(base[((size_t)y1 * a.W + x1) * a.C]);
        a.out[(((size_t)b * a.Ctot + a.c_off + c) * a.h + oy) * a.w + ox] = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    }
}

template <typename T> __global__ void nchw_to_nhwc_kernel(const float* in, T* out, int B, int C, int HW) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t rest = i / C;
        const int pix = (int)(rest % HW);
        const int b = (int)(rest / HW);
        out[i] = from_f32<T>(in[((size_t)b * C + c) * HW + pix]);
    }
}
template <typename T> __global__ void nhwc_to_nchw_kernel(const T* in, float* out, int B, int C, int HW, int ld) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const size_t rest = i / HW;
        const int c = (int)(rest % C);
        const int b = (int)(rest / C);
        out[rest * ld + pix] = to_f32<T>(in[((size_t)b * HW + pix) * C + c]);
    }
}

inline int grid_for(size_t work, int block, int cap = 256 * 8) {
    size_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}

}  // namespace

#define STREAM hipStream_t stream = (hipStream_t)stream_

extern "C" int dig_gn_finalize(const dig_gn_finalize_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->stats && a->gamma && a->beta && a->aff, DIG_EINVAL, "gn_finalize: null pointer");
    DIG_CHECK(a->B > 0 && a->groups > 0 && a->C % a->groups == 0 && a->tiles > 0 && a->hw > 0, DIG_EINVAL, "gn_finalize: bad shape");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a->B * a->groups), dim3(64), 0, stream, *a);
    DIG_LAUNCH_CHECK("gn_finalize");
    return DIG_OK;
}

extern "C" int dig_gn_fin_max_tiles(void) { return GN_FIN_MAX_TILES; }

extern "C" int dig_gn_fin_silu_add(const dig_gn_fin_silu_add_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->out && a->stats && a->gamma && a->beta, DIG_EINVAL, "gn_fin_silu_add: null pointer");
    DIG_CHECK(a->C % 8 == 0 && a->C <= GN_FIN_MAXC && a->B > 0 && a->HW > 0 && a->hw > 0, DIG_EINVAL, "gn_fin_silu_add: bad shape B=%d HW=%d C=%d", a->B, a->HW, a->C);
    DIG_CHECK(a->groups > 0 && a->C % a->groups == 0 && a->tiles > 0 && a->tiles <= GN_FIN_MAX_TILES, DIG_EINVAL,
              "gn_fin_silu_add: groups=%d tiles=%d (at most %d partials per image and group: use dig_gn_finalize + dig_gn_silu_add)", a->groups, a->tiles, GN_FIN_MAX_TILES);
    // chunks per image: about four blocks per CU over the batch, at least 2048 vectors (32 KB of bf16) per block
    const long long per_img = (long long)a->HW * (a->C / 8);
    long long chunks = (1024 + a->B - 1) / a->B;
    if (chunks > per_img / 2048) chunks = per_img / 2048;
    if (chunks < 1) chunks = 1;
    DIG_CHECK((long long)a->B * chunks < (1ll << 31), DIG_EINVAL, "gn_fin_silu_add: too many blocks");
    const dim3 grid((unsigned)(a->B * chunks));
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(gn_fin_silu_add_kernel<bf16>, grid, dim3(256), 0, stream, *a, (int)chunks);
    else hipLaunchKernelGGL(gn_fin_silu_add_kernel<float>, grid, dim3(256), 0, stream, *a, (int)chunks);
    DIG_LAUNCH_CHECK("gn_fin_silu_add");
    return DIG_OK;
}

extern "C" int dig_gn_silu_add(const dig_gn_silu_add_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->aff && a->out, DIG_EINVAL, "gn_silu_add: null pointer");
    DIG_CHECK(a->C % 8 == 0 && a->B > 0 && a->HW > 0, DIG_EINVAL, "gn_silu_add: bad shape");
    const size_t work = (size_t)a->B * a->HW * (a->C / 8);
    const int g = grid_for(work, 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(gn_silu_add_kernel<bf16>, dim3(g), dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(gn_silu_add_kernel<float>, dim3(g), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("gn_silu_add");
    return DIG_OK;
}

template <typename T> static void launch_rms(const dig_rmsnorm_args* a, hipStream_t stream) {
    const int cv = a->C / 8;
    int G = 64;
    while (G > 1 && G / 2 >= cv) G /= 2;  // smallest power of two >= cv, capped at 64
    const int rows_per_block = 256 / G;
    const int g = grid_for((size_t)((a->rows + rows_per_block - 1) / rows_per_block), 1);
    switch (G) {
#define CASE(GG) case GG: hipLaunchKernelGGL((rmsnorm_kernel<T, GG>), dim3(g), dim3(256), 0, stream, *a); break;
        CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    }
}
extern "C" int dig_rmsnorm(const dig_rmsnorm_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->g && a->out, DIG_EINVAL, "rmsnorm: null pointer");
    DIG_CHECK(a->C % 8 == 0 && a->rows > 0, DIG_EINVAL, "rmsnorm: bad shape");
    if (a->dtype == DIG_BF16) launch_rms<bf16>(a, stream);
    else launch_rms<float>(a, stream);
    DIG_LAUNCH_CHECK("rmsnorm");
    return DIG_OK;
}

extern "C" int dig_row_softmax(const dig_row_softmax_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out && a->rows > 0 && a->n > 0, DIG_EINVAL, "row_softmax: bad argument");
    DIG_CHECK(a->rows < 2147483647LL, DIG_EINVAL, "row_softmax: too many rows");
    const dim3 g((unsigned)a->rows), b(256);
    if (a->dtype == DIG_BF16) {
        if (a->in_f32) hipLaunchKernelGGL((row_softmax_kernel<float, bf16, 8>), g, b, 0, stream, (const float*)a->in, (bf16*)a->out, a->n);
        else hipLaunchKernelGGL((row_softmax_kernel<bf16, bf16, 8>), g, b, 0, stream, (const bf16*)a->in, (bf16*)a->out, a->n);
    } else {
        hipLaunchKernelGGL((row_softmax_kernel<float, float, 8>), g, b, 0, stream, (const float*)a->in, (float*)a->out, a->n);
    }
    DIG_LAUNCH_CHECK("row_softmax");
    return DIG_OK;
}

extern "C" int dig_group_softmax32(const dig_group_softmax32_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->q && a->rows > 0 && a->groups > 0, DIG_EINVAL, "group_softmax32: bad argument");
    const int g = grid_for((size_t)a->rows * a->groups, 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(group_softmax32_kernel<bf16>, dim3(g), dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(group_softmax32_kernel<float>, dim3(g), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("group_softmax32");
    return DIG_OK;
}

extern "C" int dig_row_max(const dig_row_max_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out && a->rows > 0 && a->n > 0 && a->n % 8 == 0 && a->group_rows > 0, DIG_EINVAL, "row_max: bad argument");
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(row_max_kernel<bf16>, dim3((unsigned)a->rows), dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(row_max_kernel<float>, dim3((unsigned)a->rows), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("row_max");
    return DIG_OK;
}

extern "C" int dig_ddim_step(const dig_ddim_step_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->v && a->out, DIG_EINVAL, "ddim_step: null pointer");
    DIG_CHECK(a->n > 0, DIG_EINVAL, "ddim_step: bad size n=%lld", a->n);
    DIG_CHECK(a->last || a->noise || a->sigma == 0.f, DIG_EINVAL, "ddim_step: sigma=%g needs a noise tensor", (double)a->sigma);
    DIG_CHECK(a->last || a->sqrt_recipm1 > 0.f, DIG_EINVAL, "ddim_step: sqrt_recipm1_alphas_cumprod must be positive");
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for((size_t)a->n, 256)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("ddim_step");
    return DIG_OK;
}

extern "C" int dig_sched_axpby(const dig_sched_axpby_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->y && a->t && a->sqrt_ac && a->sqrt_1mac && a->out, DIG_EINVAL, "sched_axpby: null pointer");
    DIG_CHECK(a->B > 0 && a->HW > 0, DIG_EINVAL, "sched_axpby: bad shape B=%d HW=%d", a->B, a->HW);
    hipLaunchKernelGGL(sched_axpby_kernel, dim3(grid_for((size_t)a->B * 3 * a->HW, 256)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("sched_axpby");
    return DIG_OK;
}

extern "C" int dig_sinusoid(const dig_sinusoid_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->t && a->out && a->B > 0 && a->dim >= 4 && a->dim % 2 == 0, DIG_EINVAL, "sinusoid: bad argument");
    const int g = cdiv(a->B * a->dim, 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(sinusoid_kernel<bf16>, dim3(g), dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(sinusoid_kernel<float>, dim3(g), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("sinusoid");
    return DIG_OK;
}

extern "C" int dig_time_embed(const dig_time_embed_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->t && a->w1 && a->b1 && a->w3 && a->b3 && a->temb && a->st, DIG_EINVAL, "time_embed: null pointer");
    auto okw = [](int k) { return k >= 4 && k % 4 == 0 && (k % 256 == 0 || (k <= 256 && ((k / 4) & (k / 4 - 1)) == 0)); };
    DIG_CHECK(a->B > 0 && okw(a->dim) && okw(a->tdim) && (a->dim + a->tdim) * 4 <= 48 * 1024, DIG_EINVAL,
              "time_embed: needs B > 0 and dim, tdim each a power of two in [4, 256] or a multiple of 256 (got B=%d dim=%d tdim=%d)", a->B, a->dim, a->tdim);
    DIG_CHECK(((uintptr_t)a->w1 | (uintptr_t)a->w3) % 16 == 0, DIG_EALIGN, "time_embed: weights must be 16-byte aligned");
    hipLaunchKernelGGL(time_embed_kernel, dim3(a->B), dim3(1024), (size_t)(a->dim + a->tdim) * 4, stream, *a);
    DIG_LAUNCH_CHECK("time_embed");
    return DIG_OK;
}

extern "C" int dig_time_mlp(const dig_time_mlp_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->st && a->w && a->out, DIG_EINVAL, "time_mlp: null pointer");
    DIG_CHECK(a->B > 0 && a->N > 0 && a->K > 0 && a->K % 64 == 0, DIG_EINVAL, "time_mlp: needs K %% 64 == 0 (got B=%d N=%d K=%d)", a->B, a->N, a->K);
    DIG_CHECK(((uintptr_t)a->st | (uintptr_t)a->w) % 16 == 0, DIG_EALIGN, "time_mlp: st / w must be 16-byte aligned");
    hipLaunchKernelGGL(time_mlp_kernel, dim3(cdiv(a->N, 32), cdiv(a->B, 32)), dim3(64), 0, stream, *a);
    DIG_LAUNCH_CHECK("time_mlp");
    return DIG_OK;
}

extern "C" int dig_act(const dig_act_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out && a->n > 0, DIG_EINVAL, "act: bad argument");
    const dim3 g(grid_for((size_t)a->n, 256)), b(256);
    const bool bf = a->dtype == DIG_BF16;
    const bool fi = a->in_f32 || !bf, fo = a->out_f32 || !bf;
    if (fi && fo) hipLaunchKernelGGL((act_kernel<float, float>), g, b, 0, stream, (const float*)a->in, (float*)a->out, a->n, a->act);
    else if (fi) hipLaunchKernelGGL((act_kernel<float, bf16>), g, b, 0, stream, (const float*)a->in, (bf16*)a->out, a->n, a->act);
    else if (fo) hipLaunchKernelGGL((act_kernel<bf16, float>), g, b, 0, stream, (const bf16*)a->in, (float*)a->out, a->n, a->act);
    else hipLaunchKernelGGL((act_kernel<bf16, bf16>), g, b, 0, stream, (const bf16*)a->in, (bf16*)a->out, a->n, a->act);
    DIG_LAUNCH_CHECK("act");
    return DIG_OK;
}

extern "C" int dig_bilinear_ac_concat(const dig_bilinear_ac_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out, DIG_EINVAL, "bilinear_ac: null pointer");
    DIG_CHECK(a->h > 0 && a->w > 0 && a->c_off >= 0 && a->c_off + a->C <= a->Ctot, DIG_EINVAL, "bilinear_ac: bad geometry");
    static const bool flat = getenv("DIG_BILINEAR_FLAT") != nullptr;  // test hook: the thread-per-element kernel
    if (!flat) {
        const long long blocks = (long long)a->B * ((a->C + BL_C - 1) / BL_C) * ((a->h * a->w + BL_P - 1) / BL_P);
        DIG_CHECK(blocks < (1ll << 31), DIG_EINVAL, "bilinear_ac: too many tiles");
        if (a->dtype == DIG_BF16) hipLaunchKernelGGL(bilinear_ac_tiled_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, *a);
        else hipLaunchKernelGGL(bilinear_ac_tiled_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, *a);
        DIG_LAUNCH_CHECK("bilinear_ac");
        return DIG_OK;
    }
    const int g = grid_for((size_t)a->B * a->h * a->w * a->C, 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(bilinear_ac_kernel<bf16>, dim3(g), dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(bilinear_ac_kernel<float>, dim3(g), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("bilinear_ac");
    return DIG_OK;
}

extern "C" int dig_nchw_f32_to_nhwc(const dig_layout_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out, DIG_EINVAL, "layout: null pointer");
    const int g = grid_for((size_t)a->B * a->C * a->HW, 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const float*)a->in, (bf16*)a->out, a->B, a->C, a->HW);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)a->in, (float*)a->out, a->B, a->C, a->HW);
    DIG_LAUNCH_CHECK("nchw_to_nhwc");
    return DIG_OK;
}
extern "C" int dig_nhwc_to_nchw_f32(const dig_layout_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out, DIG_EINVAL, "layout: null pointer");
    DIG_CHECK(a->out_ld == 0 || a->out_ld >= a->HW, DIG_EINVAL, "layout: out_ld=%d < HW=%d", a->out_ld, a->HW);
    const int g = grid_for((size_t)a->B * a->C * a->HW, 256), ld = a->out_ld ? a->out_ld : a->HW;
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(g), dim3(256), 0, stream, (const bf16*)a->in, (float*)a->out, a->B, a->C, a->HW, ld);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)a->in, (float*)a->out, a->B, a->C, a->HW, ld);
    DIG_LAUNCH_CHECK("nhwc_to_nchw");
    return DIG_OK;
}
