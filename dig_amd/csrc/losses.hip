// Heads and forward losses around the MCTformer passes of model_WSSS.update (models/model_ECCV24_DiG.py:324-421), gfx950.
// Everything here is a few KB to a few hundred KB per batch: one wave or one block per (image, class) row, fp32,
// deterministic summation orders (no atomics), results as device scalars so that the host never synchronises for a loss.
//   head_pool     NHWC head map -> ReLU'd NCHW map + the top-k pooled patch logits   (mctformer.py:296-300, 336-341)
//   row_mean      class-token logits: mean over the embedding of (x0 [+ x1] / 2)      (mctformer.py:295, 287-288)
//   msm_loss      F.multilabel_soft_margin_loss                                        (model.py:337-340, 358-359, 406-409)
//   kd_bin_loss   KLDivLoss(batchmean, log_target) of log cat(softmax(s/T), 1 - softmax(s/T)) (model.py:347-361)
//   maxnorm_l1    |max_norm(a) [* label] - max_norm(resize(b))| * label, per (image, class) partial sums (model.py:368-390, 414-418)
//   sum_scale     out = scale * sum(partials)                                          (the .mean() of the two lines above, the total)
#include "common.hpp"

namespace {

#define STREAM hipStream_t stream = (hipStream_t)stream_

// ---- head_pool: one wave per (b, c) ------------------------------------------------------------------------------------
// pooled[b][c] = mean(topk(relu(v), k)) - mean(topk(relu(-v), k)) over the P positions of channel c, k = min(3, P)
struct Top3 {
    float t0, t1, t2;  // descending
    __device__ __forceinline__ void init() { t0 = t1 = t2 = -INFINITY; }
    __device__ __forceinline__ void push(float v) {
        if (v > t0) { t2 = t1; t1 = t0; t0 = v; }
        else if (v > t1) { t2 = t1; t1 = v; }
        else if (v > t2) t2 = v;
    }
    __device__ __forceinline__ void pop() { t0 = t1; t1 = t2; t2 = -INFINITY; }
};
// the k largest values held anywhere in the wave, summed largest first (the order torch.topk(...)[0].mean(-1) adds them in)
__device__ __forceinline__ float wave_topk_sum(Top3 t, int k, int lane) {
    float s = 0.f;
    for (int i = 0; i < k; i++) {
        const float m = wave_max(t.t0);
        const unsigned long long owners = __ballot(t.t0 == m);
        if (lane == (int)__builtin_ctzll(owners)) t.pop();
        s += m;
    }
    return s;
}
__global__ __launch_bounds__(256) void head_pool_kernel(dig_head_pool_args a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)a.B * a.C) return;
    const int b = (int)(row / a.C), c = (int)(row % a.C);
    const float* in = a.in + (size_t)b * a.P * a.C + c;
    Top3 pos, neg;
    pos.init();
    neg.init();
    for (int p = lane; p < a.P; p += 64) {
        const float v = in[(size_t)p * a.C];
        const float rp = fmaxf(v, 0.f), rn = fmaxf(-v, 0.f);
        if (a.relu_nchw) a.relu_nchw[((size_t)b * a.C + c) * a.P + p] = rp;
        pos.push(rp);
        neg.push(rn);
    }
    const int k = min(3, a.P);
    const float sp = wave_topk_sum(pos, k, lane), sn = wave_topk_sum(neg, k, lane);
    if (lane == 0 && a.pooled) a.pooled[row] = sp / (float)k - sn / (float)k;
}

// ---- row_mean: one wave per (b, c) row of D floats ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_mean_kernel(dig_row_mean_args a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)a.B * a.C) return;
    const int b = (int)(row / a.C), c = (int)(row % a.C);
    const float* x0 = a.x0 + (size_t)b * a.bstride0 + (size_t)c * a.D;
    const float* x1 = a.x1 ? a.x1 + (size_t)b * a.bstride1 + (size_t)c * a.D : nullptr;
    float s = 0.f;
    for (int d = lane; d < a.D; d += 64) s += x1 ? (x0[d] + x1[d]) * 0.5f : x0[d];
    s = wave_sum(s);
    if (lane == 0) a.out[row] = s / (float)a.D;
}

// ---- multilabel soft margin: out = scale * mean_b( -mean_c( y*logsigmoid(x) + (1-y)*logsigmoid(-x) ) ) -----------------------
__device__ __forceinline__ float logsigmoid_f(float x) {  // min(x, 0) - log1p(exp(-|x|)), torch's formulation
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__global__ __launch_bounds__(256) void msm_loss_kernel(dig_msm_loss_args a) {
    __shared__ float red[32];
    const int n = a.B * a.C;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = a.x[i], y = a.y[i];
        s += y * logsigmoid_f(x) + (1.f - y) * logsigmoid_f(-x);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float v = a.scale * (-s / (float)n);
        a.out[0] = a.accumulate ? a.out[0] + v : v;
    }
}

// ---- binary-split distillation term: one wave per row, block = 4 rows at a time, a single block ---------------------------
__global__ __launch_bounds__(256) void kd_bin_loss_kernel(dig_kd_bin_loss_args a) {
    __shared__ float red[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float tot = 0.f;
    for (int b = wave; b < a.B; b += 4) {
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
        for (int c = lane; c < a.C; c += 64) {
            const float ps = expf(s[c] / a.T - ms) / zs, pt = expf(t[c] / a.T - mt) / zt;
            const float ls = logf(ps), lt = logf(pt), ls1 = logf(1.f - ps), lt1 = logf(1.f - pt);
            tot += expf(lt) * (lt - ls) + expf(lt1) * (lt1 - ls1);  // KLDivLoss(log_target=True) pointwise
        }
    }
    tot = block_sum(tot, red);
    if (threadIdx.x == 0) {
        const float v = a.scale * (tot / (float)a.B);  // reduction='batchmean'
        a.out[0] = a.accumulate ? a.out[0] + v : v;
    }
}

// ---- max_norm'd L1 between two CAM stacks: one block per (b, c) ---------------------------------------------------------
__device__ __forceinline__ float bilin_at(const float* m, int hi, int wi, int ho, int wo, int oy, int ox) {
    if (hi == ho && wi == wo) return m[oy * wi + ox];
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    const float fy = fmaxf((oy + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * sw - 0.5f, 0.f);
    const int y0 = min((int)fy, hi - 1), x0 = min((int)fx, wi - 1);
    const int y1 = min(y0 + 1, hi - 1), x1 = min(x0 + 1, wi - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    return (1.f - ly) * ((1.f - lx) * m[y0 * wi + x0] + lx * m[y0 * wi + x1]) + ly * ((1.f - lx) * m[y1 * wi + x0] + lx * m[y1 * wi + x1]);
}
__global__ __launch_bounds__(256) void maxnorm_l1_kernel(dig_maxnorm_l1_args a) {
    extern __shared__ float lsm[];
    __shared__ float red[32];
    const int HW = a.ha * a.wa, HWb = a.hb * a.wb;
    float* am = lsm;            // [HW] relu(a)
    float* bm = am + HW;        // [HW] relu(resize(b))
    float* braw = bm + HW;      // [HWb]
    const int bc = blockIdx.x, tid = threadIdx.x;
    const float* ap = a.a + (size_t)bc * HW;
    const float* bp = a.b + (size_t)bc * HWb;
    for (int i = tid; i < HWb; i += 256) braw[i] = bp[i];
    __syncthreads();
    float amax = -INFINITY, amin = INFINITY, bmax = -INFINITY, bmin = INFINITY;
    for (int i = tid; i < HW; i += 256) {
        const float av = fmaxf(ap[i], 0.f);
        const float bv = fmaxf(bilin_at(braw, a.hb, a.wb, a.ha, a.wa, i / a.wa, i % a.wa), 0.f);
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
    const float lab = a.label[bc];
    const float la = a.a_times_label ? lab : 1.f;
    float s = 0.f;
    for (int i = tid; i < HW; i += 256) {
        const float an = fmaxf(am[i] - amin - 1e-5f, 0.f) / (amax - amin + 1e-5f);
        const float bn = fmaxf(bm[i] - bmin - 1e-5f, 0.f) / (bmax - bmin + 1e-5f);
        s += fabsf((an * la - bn) * lab);
    }
    s = block_sum(s, red);
    if (tid == 0) a.partial[bc] = s;
}

__global__ __launch_bounds__(256) void sum_scale_kernel(dig_sum_scale_args a) {
    __shared__ float red[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < a.n; i += 256) s += a.in[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float v = a.scale * s;
        a.out[0] = a.accumulate ? a.out[0] + v : v;
    }
}

}  // namespace

extern "C" int dig_head_pool(const dig_head_pool_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && (a->relu_nchw || a->pooled), DIG_EINVAL, "head_pool: null pointer");
    DIG_CHECK(a->B > 0 && a->P > 0 && a->C > 0, DIG_EINVAL, "head_pool: bad shape B=%d P=%d C=%d", a->B, a->P, a->C);
    hipLaunchKernelGGL(head_pool_kernel, dim3((unsigned)(((long long)a->B * a->C + 3) / 4)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("head_pool");
    return DIG_OK;
}

extern "C" int dig_row_mean(const dig_row_mean_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x0 && a->out, DIG_EINVAL, "row_mean: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->D > 0 && a->bstride0 >= (long long)a->C * a->D && (!a->x1 || a->bstride1 >= (long long)a->C * a->D),
              DIG_EINVAL, "row_mean: bad shape B=%d C=%d D=%d", a->B, a->C, a->D);
    hipLaunchKernelGGL(row_mean_kernel, dim3((unsigned)(((long long)a->B * a->C + 3) / 4)), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("row_mean");
    return DIG_OK;
}

extern "C" int dig_msm_loss(const dig_msm_loss_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->x && a->y && a->out, DIG_EINVAL, "msm_loss: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0, DIG_EINVAL, "msm_loss: bad shape B=%d C=%d", a->B, a->C);
    hipLaunchKernelGGL(msm_loss_kernel, dim3(1), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("msm_loss");
    return DIG_OK;
}

extern "C" int dig_kd_bin_loss(const dig_kd_bin_loss_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->student && a->teacher && a->out, DIG_EINVAL, "kd_bin_loss: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->T > 0.f, DIG_EINVAL, "kd_bin_loss: bad argument B=%d C=%d T=%g", a->B, a->C, (double)a->T);
    hipLaunchKernelGGL(kd_bin_loss_kernel, dim3(1), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("kd_bin_loss");
    return DIG_OK;
}

extern "C" int dig_maxnorm_l1(const dig_maxnorm_l1_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->a && a->b && a->label && a->partial, DIG_EINVAL, "maxnorm_l1: null pointer");
    DIG_CHECK(a->B > 0 && a->C > 0 && a->ha > 0 && a->wa > 0 && a->hb > 0 && a->wb > 0, DIG_EINVAL, "maxnorm_l1: bad shape");
    const size_t lds = ((size_t)2 * a->ha * a->wa + (size_t)a->hb * a->wb) * sizeof(float);
    DIG_CHECK(lds <= 150 * 1024, DIG_EINVAL, "maxnorm_l1: maps of %dx%d / %dx%d need %zu B of LDS (max 150 KiB)", a->ha, a->wa, a->hb, a->wb, lds);
    hipError_t e = hipFuncSetAttribute((const void*)maxnorm_l1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "maxnorm_l1: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
    hipLaunchKernelGGL(maxnorm_l1_kernel, dim3((unsigned)(a->B * a->C)), dim3(256), lds, stream, *a);
    DIG_LAUNCH_CHECK("maxnorm_l1");
    return DIG_OK;
}

extern "C" int dig_sum_scale(const dig_sum_scale_args* a, void* stream_) {
    STREAM;
    DIG_CHECK(a && a->in && a->out && a->n > 0, DIG_EINVAL, "sum_scale: bad argument");
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, *a);
    DIG_LAUNCH_CHECK("sum_scale");
    return DIG_OK;
}
