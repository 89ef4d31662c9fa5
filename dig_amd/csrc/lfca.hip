// Element-wise / reduction helpers of the LFCA consumer of the guidance maps (networks/mctformer.py:56-166,
// networks/resnet101.py:195-222): everything that is not a convolution, a GEMM, a LayerNorm or a softmax there.
//   affine_act_add : out = post(pre(xs * (a[b][c] * x + b[b][c])) + rs * res)      (GroupNorm-apply / ReLU / averages)
//   gn_stats       : per-(image, group) sum and sum of squares of an NHWC tensor, as deterministic per-slab partials
//                    in the layout dig_gn_finalize reads (GroupNorm on a tensor that is not a conv3x3 output)
//   bilinear_resize: F.interpolate(mode='bilinear', align_corners=False) on NHWC, optionally + res
// All three are HBM streams: 16-byte vectors of 8 channels per lane.
#include "common.hpp"

namespace {

__device__ __forceinline__ float act_f(float x, int act) {
    switch (act) {
        case 1: return silu_f(x);
        case 2: return fmaxf(x, 0.f);
        case 3: return gelu_erf_f(x);
        default: return x;
    }
}

template <typename T> __global__ __launch_bounds__(256) void affine_act_add_kernel(dig_affine_act_add_args a) {
    const int cv = a.C / 8;
    const size_t per_img = (size_t)a.HW * cv, total = per_img * a.B;
    const T* x = (const T*)a.x;
    const T* res = (const T*)a.res;
    T* out = (T*)a.out;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(v / per_img);
        const int c = (int)(v % cv) * 8;
        Frag<T> f, r;
        frag_load<T>(f, x + v * 8);
        if (res) frag_load<T>(r, res + v * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float y = frag_get<T>(f, j);
            if (a.aff) {
                const float* af = a.aff + ((size_t)b * a.C + c + j) * 2;
                y = af[0] * y + af[1];
            }
            y = act_f(a.x_scale * y, a.pre_act);
            if (res) y += a.res_scale * frag_get<T>(r, j);
            frag_set<T>(f, j, act_f(y, a.post_act));
        }
        frag_store<T>(f, out + v * 8);
    }
}

// grid (tiles, B); block 256.  Thread t owns channel vector t % cv_eff of the pixels p = t / cv_eff + k * ppb of its slab, so
// its partial sums belong to ONE group; the block combines them in a fixed order (deterministic).
template <typename T> __global__ __launch_bounds__(256) void gn_stats_kernel(dig_gn_stats_args a) {
    __shared__ float ps[256], pss[256];
    const int cv = a.C / 8, cpg = a.C / a.groups;
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ppt = (a.HW + a.tiles - 1) / a.tiles;
    const int p0 = tile * ppt, p1 = min(a.HW, p0 + ppt);
    const T* x = (const T*)a.x + (size_t)b * a.HW * a.C;
    float gs = 0.f, gss = 0.f;  // only meaningful for tid < groups
    const int slots = cv < 256 ? cv : 256;        // vectors handled side by side
    const int ppb = 256 / slots;                  // pixels handled side by side
    const int vs = tid % slots, pr = tid / slots;
    for (int v0 = 0; v0 < cv; v0 += slots) {
        const int v = v0 + vs;
        float s = 0.f, ss = 0.f;
        if (v < cv && pr < ppb)
            for (int p = p0 + pr; p < p1; p += ppb) {
                Frag<T> f;
                frag_load<T>(f, x + (size_t)p * a.C + v * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float t = frag_get<T>(f, j);
                    s += t;
                    ss += t * t;
                }
            }
        __syncthreads();
        ps[tid] = s;
        pss[tid] = ss;
        __syncthreads();
        if (tid < a.groups) {  // fixed-order sum of the threads whose vector lies in group `tid` (cpg % 8 == 0)
            for (int t = 0; t < 256; t++) {
                const int tv = v0 + t % slots;
                if (tv < cv && t / slots < ppb && (tv * 8) / cpg == tid) {
                    gs += ps[t];
                    gss += pss[t];
                }
            }
        }
    }
    if (tid < a.groups) {
        float* d = a.stats + (((size_t)b * a.groups + tid) * a.tiles + tile) * 2;
        d[0] = gs;
        d[1] = gss;
    }
}

// thread = (b, oy, ox, channel vector); PyTorch's align_corners=False source index: max(0, (o + 0.5) * in/out - 0.5)
template <typename T> __global__ __launch_bounds__(256) void bilinear_resize_kernel(dig_bilinear_resize_args a) {
    const int cv = a.C / 8;
    const size_t total = (size_t)a.B * a.ho * a.wo * cv;
    const T* in = (const T*)a.in;
    const T* res = (const T*)a.res;
    T* out = (T*)a.out;
    const float sh = (float)a.hi / (float)a.ho, sw = (float)a.wi / (float)a.wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % cv);
        size_t p = i / cv;
        const int ox = (int)(p % a.wo);
        p /= a.wo;
        const int oy = (int)(p % a.ho), b = (int)(p / a.ho);
        const float fy = fmaxf((oy + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * sw - 0.5f, 0.f);
        const int y0 = min((int)fy, a.hi - 1), x0 = min((int)fx, a.wi - 1);
        const int y1 = min(y0 + 1, a.hi - 1), x1 = min(x0 + 1, a.wi - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const T* base = in + (size_t)b * a.hi * a.wi * a.C + v * 8;
        Frag<T> f00, f01, f10, f11, r, o;
        frag_load<T>(f00, base + ((size_t)y0 * a.wi + x0) * a.C);
        frag_load<T>(f01, base + ((size_t)y0 * a.wi + x1) * a.C);
        frag_load<T>(f10, base + ((size_t)y1 * a.wi + x0) * a.C);
        frag_load<T>(f11, base + ((size_t)y1 * a.wi + x1) * a.C);
        if (res) frag_load<T>(r, res + i * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // same association as ATen's upsample_bilinear2d: w(1-ly)(.. (1-lx) + .. lx) + ly(..)
            float y = (1.f - ly) * ((1.f - lx) * frag_get<T>(f00, j) + lx * frag_get<T>(f01, j)) +
                      ly * ((1.f - lx) * frag_get<T>(f10, j) + lx * frag_get<T>(f11, j));
            if (res) y += frag_get<T>(r, j);
            frag_set<T>(o, j, y);
        }
        frag_store<T>(o, out + i * 8);
    }
}

inline int grid_for(size_t work, int block, int cap = 256 * 8) {
    size_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}

}  // namespace

namespace {
template <typename TI, typename TO> __global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long long nvec) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        Frag<TI> f;
        Frag<TO> o;
        frag_load<TI>(f, in + v * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) frag_set<TO>(o, j, frag_get<TI>(f, j));
        frag_store<TO>(o, out + v * 8);
    }
}
}  // namespace

extern "C" int dig_cast(const dig_cast_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->in && a->out && a->n > 0 && a->n % 8 == 0, DIG_EINVAL, "cast: n must be a positive multiple of 8");
    DIG_CHECK(((uintptr_t)a->in | (uintptr_t)a->out) % 16 == 0, DIG_EALIGN, "cast: pointers must be 16-byte aligned");
    const int g = grid_for((size_t)(a->n / 8), 256);
    if (a->to_f32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(g), dim3(256), 0, stream, (const bf16*)a->in, (float*)a->out, a->n / 8);
    else hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(g), dim3(256), 0, stream, (const float*)a->in, (bf16*)a->out, a->n / 8);
    DIG_LAUNCH_CHECK("cast");
    return DIG_OK;
}

extern "C" int dig_affine_act_add(const dig_affine_act_add_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->x && a->out, DIG_EINVAL, "affine_act_add: null pointer");
    DIG_CHECK(a->B > 0 && a->HW > 0 && a->C > 0 && a->C % 8 == 0, DIG_EINVAL, "affine_act_add: bad shape (C %% 8)");
    DIG_CHECK(a->pre_act >= 0 && a->pre_act <= 3 && a->post_act >= 0 && a->post_act <= 3, DIG_EINVAL, "affine_act_add: unknown activation");
    DIG_CHECK(((uintptr_t)a->x | (uintptr_t)a->res | (uintptr_t)a->out) % 16 == 0, DIG_EALIGN, "affine_act_add: pointers must be 16-byte aligned");
    const int g = grid_for((size_t)a->B * a->HW * (a->C / 8), 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(affine_act_add_kernel<bf16>, dim3(g), dim3(256), 0, stream, *a);
    else if (a->dtype == DIG_F32) hipLaunchKernelGGL(affine_act_add_kernel<float>, dim3(g), dim3(256), 0, stream, *a);
    else DIG_CHECK(false, DIG_EINVAL, "affine_act_add: unknown dtype %d", a->dtype);
    DIG_LAUNCH_CHECK("affine_act_add");
    return DIG_OK;
}

extern "C" int dig_gn_stats(const dig_gn_stats_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->x && a->stats, DIG_EINVAL, "gn_stats: null pointer");
    DIG_CHECK(a->B > 0 && a->B <= 65535 && a->HW > 0 && a->tiles > 0 && a->tiles <= a->HW, DIG_EINVAL, "gn_stats: bad shape");
    DIG_CHECK(a->groups > 0 && a->groups <= 256 && a->C % a->groups == 0 && (a->C / a->groups) % 8 == 0, DIG_EINVAL,
              "gn_stats: group width must be a multiple of 8 channels");
    DIG_CHECK((uintptr_t)a->x % 16 == 0, DIG_EALIGN, "gn_stats: x must be 16-byte aligned");
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16>, dim3(a->tiles, a->B), dim3(256), 0, stream, *a);
    else if (a->dtype == DIG_F32) hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(a->tiles, a->B), dim3(256), 0, stream, *a);
    else DIG_CHECK(false, DIG_EINVAL, "gn_stats: unknown dtype %d", a->dtype);
    DIG_LAUNCH_CHECK("gn_stats");
    return DIG_OK;
}

extern "C" int dig_bilinear_resize(const dig_bilinear_resize_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->in && a->out, DIG_EINVAL, "bilinear_resize: null pointer");
    DIG_CHECK(a->B > 0 && a->hi > 0 && a->wi > 0 && a->ho > 0 && a->wo > 0 && a->C > 0 && a->C % 8 == 0, DIG_EINVAL, "bilinear_resize: bad shape (C %% 8)");
    DIG_CHECK(((uintptr_t)a->in | (uintptr_t)a->res | (uintptr_t)a->out) % 16 == 0, DIG_EALIGN, "bilinear_resize: pointers must be 16-byte aligned");
    const int g = grid_for((size_t)a->B * a->ho * a->wo * (a->C / 8), 256);
    if (a->dtype == DIG_BF16) hipLaunchKernelGGL(bilinear_resize_kernel<bf16>, dim3(g), dim3(256), 0, stream, *a);
    else if (a->dtype == DIG_F32) hipLaunchKernelGGL(bilinear_resize_kernel<float>, dim3(g), dim3(256), 0, stream, *a);
    else DIG_CHECK(false, DIG_EINVAL, "bilinear_resize: unknown dtype %d", a->dtype);
    DIG_LAUNCH_CHECK("bilinear_resize");
    return DIG_OK;
}

// ---- mIoU bookkeeping of evaluation.py:72-88 on the device ---------------------------------------------------------------
// counts[0][i] += #(predict == i and gt < 255)   (P)
// counts[1][i] += #(gt == i and gt < 255)        (T)
// counts[2][i] += #(gt == i and predict == gt and gt < 255)   (TP)
// Integer counters: block-local LDS histogram, then one 64-bit atomic per non-empty bin (order-independent, exact).
namespace {
__global__ __launch_bounds__(256) void seg_hist_kernel(const unsigned char* __restrict__ predict, const unsigned char* __restrict__ gt,
                                                       unsigned long long* __restrict__ counts, long long n, int num_cls) {
    __shared__ unsigned int h[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) h[i] = 0u;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int p = predict[i], g = gt[i];
        if (g < 255) {
            if (p < num_cls) atomicAdd(&h[p], 1u);
            if (g < num_cls) {
                atomicAdd(&h[256 + g], 1u);
                if (p == g) atomicAdd(&h[512 + g], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 256; i += 256) {
        const int k = i >> 8, c = i & 255;
        if (c < num_cls && h[i] != 0u) atomicAdd(&counts[(size_t)k * num_cls + c], (unsigned long long)h[i]);
    }
}
}  // namespace

extern "C" int dig_seg_hist(const dig_seg_hist_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->predict && a->gt && a->counts, DIG_EINVAL, "seg_hist: null pointer");
    DIG_CHECK(a->n > 0 && a->num_cls > 0 && a->num_cls <= 255, DIG_EINVAL, "seg_hist: bad shape (num_cls <= 255: 255 is the ignore label)");
    long long blocks = (a->n + 256LL * 16 - 1) / (256LL * 16);  // 16 pixels per thread: each block's 2^32-safe LDS counters see <= 4096 pixels per pass of the grid
    if (blocks > 2048) blocks = 2048;
    DIG_CHECK(a->n / blocks < 4000000000LL, DIG_EINVAL, "seg_hist: too many pixels per call for the 32-bit block counters");
    hipLaunchKernelGGL(seg_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a->predict, a->gt, a->counts, a->n, a->num_cls);
    DIG_LAUNCH_CHECK("seg_hist");
    return DIG_OK;
}
