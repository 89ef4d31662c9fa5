// Stem: input normalisation + q_sample + conv7x7 (3 -> Cout, pad 3) in one kernel (gfx950).
// The noised image x_t is formed in registers from the NCHW fp32 image and noise while the 22x22x3 halo tile
// is staged in LDS, so normalisation and q_sample never touch HBM on their own.  K = 147 is too ragged for a
// clean MFMA mapping and the layer is 0.5 % of the network's FLOPs, so this is an fp32 VALU direct convolution:
// one thread per output pixel, 16 output channels per pass, weights broadcast from LDS as float4.
#include "common.hpp"

namespace {

constexpr int ST = 16, SH = ST + 6;  // 16x16 tile, 22x22 halo

template <typename T> __global__ __launch_bounds__(256) void stem_kernel(dig_stem_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* halo = sm;                 // [3][22][23]
    float* wl = sm + 3 * SH * (SH + 1);  // [147][Cout]
    const int tid = threadIdx.x;
    const int tiles_x = (a.W + ST - 1) / ST, tiles_y = (a.H + ST - 1) / ST;
    const int b = blockIdx.x / (tiles_x * tiles_y), tile = blockIdx.x % (tiles_x * tiles_y);
    const int ty0 = (tile / tiles_x) * ST, tx0 = (tile % tiles_x) * ST;
    const long long t = a.t[b];
    const float ca = a.sqrt_ac[t], cs = a.sqrt_1mac[t];
    const size_t HW = (size_t)a.H * a.W;
    for (int i = tid; i < 3 * SH * SH; i += 256) {
        const int c = i / (SH * SH), rem = i % (SH * SH), hy = rem / SH, hx = rem % SH;
        const int Y = ty0 + hy - 3, X = tx0 + hx - 3;
        float v = 0.f;
        if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
            const size_t idx = ((size_t)b * 3 + c) * HW + (size_t)Y * a.W + X;
            const float x0 = a.pre_scale[c] * a.img[idx] + a.pre_shift[c];
            v = ca * x0 + cs * a.noise[idx];
            if (a.x_t && hy >= 3 && hy < 3 + ST && hx >= 3 && hx < 3 + ST) a.x_t[idx] = v;
        }
        halo[(c * SH + hy) * (SH + 1) + hx] = v;
    }
    for (int i = tid; i < 147 * a.Cout; i += 256) wl[i] = a.w[i];
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    const int Y = ty0 + py, X = tx0 + px;
    const bool ok = Y < a.H && X < a.W;
    T* out = (T*)a.out + (((size_t)b * a.H + Y) * a.W + X) * a.Cout;
    for (int n0 = 0; n0 < a.Cout; n0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] = a.bias[n0 + j];
        for (int ky = 0; ky < 7; ky++)
            for (int kx = 0; kx < 7; kx++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float x = halo[(c * SH + py + ky) * (SH + 1) + px + kx];
                    const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + ((ky * 7 + kx) * 3 + c) * a.Cout + n0);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f32x4 w = w4[q];
                        acc[q * 4 + 0] += x * w[0];
                        acc[q * 4 + 1] += x * w[1];
                        acc[q * 4 + 2] += x * w[2];
                        acc[q * 4 + 3] += x * w[3];
                    }
                }
        if (ok) {
            Frag<T> f0, f1;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                frag_set<T>(f0, j, acc[j]);
                frag_set<T>(f1, j, acc[8 + j]);
            }
            frag_store<T>(f0, out + n0);
            frag_store<T>(f1, out + n0 + 8);
        }
    }
}

}  // namespace

extern "C" int dig_qsample_stem(const dig_stem_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->img && a->noise && a->t && a->sqrt_ac && a->sqrt_1mac && a->w && a->bias && a->out, DIG_EINVAL, "qsample_stem: null pointer");
    DIG_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->Cout % 16 == 0 && a->Cout <= 128, DIG_EINVAL, "qsample_stem: bad shape (Cout=%d)", a->Cout);
    const size_t lds = (size_t)(3 * SH * (SH + 1) + 147 * a->Cout) * sizeof(float);
    const int tiles = cdiv(a->W, ST) * cdiv(a->H, ST);
    if (a->dtype == DIG_BF16) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)stem_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(stem_kernel<bf16>, dim3(a->B * tiles), dim3(256), lds, stream, *a);
    } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)stem_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(stem_kernel<float>, dim3(a->B * tiles), dim3(256), lds, stream, *a);
    }
    DIG_LAUNCH_CHECK("qsample_stem");
    return DIG_OK;
}
