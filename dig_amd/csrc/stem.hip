// Stem: input normalisation + q_sample + conv7x7 (3 -> Cout, pad 3) in one kernel (gfx950).
//
// The noised image x_t = ca[t]*pre(img) + cs[t]*noise is formed in registers from the NCHW fp32 inputs while the
// (8+6)x(16+6)x3 halo tile is staged in LDS, so normalisation and q_sample never touch HBM on their own (x_t is
// also written out once, NCHW fp32, because p_losses returns it).  The convolution is an implicit GEMM on MFMA:
// each 256-thread block owns an 8x16-pixel tile, builds the [128 pixels][K] im2col panel in LDS with
// K = 7*7*3 = 147 zero-padded to 160 (k = (ky*7+kx)*3 + c), keeps the [Cout][K] weight panel beside it, and runs
// 10 16-deep MFMA steps per 32x32 output tile.  Output NHWC in the storage type T (bf16 or f32).
#include "common.hpp"

namespace {

constexpr int TH = 8, TW = 16;            // output tile
constexpr int HH = TH + 6, HW_ = TW + 6;  // halo 14 x 22
constexpr int KPAD = 160, KROW = KPAD + 8;

template <typename T, int HALF> __device__ __forceinline__ void build_rows(const float* hb, T* dst) {
#pragma unroll
    for (int v = 0; v < 10; v++) {
        Frag<T> f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int k = (HALF * 10 + v) * 8 + j;  // compile-time after unrolling
            float x = 0.f;
            if (k < 147) {
                const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
                x = hb[(c * HH + ky) * (HW_ + 1) + kx];
            }
            frag_set<T>(f, j, x);
        }
        frag_store<T>(f, dst + (HALF * 10 + v) * 8);
    }
}

template <typename T> __global__ __launch_bounds__(256) void stem_kernel(dig_stem_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* halo = reinterpret_cast<float*>(smem);                    // [3][14][23] fp32
    T* Ap = reinterpret_cast<T*>(smem + 3 * HH * (HW_ + 1) * 4 + 8); // [128][KROW]  (offset keeps 16-B alignment: 3864+8)
    T* Wp = Ap + 128 * KROW;                                         // [Cout][KROW]
    float* ot = reinterpret_cast<float*>(Ap);                        // epilogue staging [128][36] fp32, aliases Ap
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.B * tiles_x * tiles_y;
    const size_t HW = (size_t)a.H * a.W;
    // weight panel, once per (persistent) block: w is [Cout][160] in T, copied as 16-byte vectors
    for (int i = tid; i < a.Cout * (KPAD / 8); i += 256) {
        const int n = i / (KPAD / 8), v = i - n * (KPAD / 8);
        Frag<T> f;
        frag_load<T>(f, (const T*)a.w + (size_t)n * KPAD + v * 8);
        frag_store<T>(f, Wp + n * KROW + v * 8);
    }
    for (int tidx = blockIdx.x; tidx < ntiles; tidx += gridDim.x) {
        const int b = tidx / (tiles_x * tiles_y), tile = tidx % (tiles_x * tiles_y);
        const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
        const long long t = a.t[b];
        const float ca = a.sqrt_ac[t], cs = a.sqrt_1mac[t];
        __syncthreads();  // previous tile's staging reads are done
        for (int i = tid; i < 3 * HH * HW_; i += 256) {
            const int c = i / (HH * HW_), rem = i % (HH * HW_), hy = rem / HW_, hx = rem % HW_;
            const int Y = ty0 + hy - 3, X = tx0 + hx - 3;
            float v = 0.f;
            if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
                const size_t idx = ((size_t)b * 3 + c) * HW + (size_t)Y * a.W + X;
                const float x0 = a.pre_scale[c] * a.img[idx] + a.pre_shift[c];
                v = ca * x0 + cs * a.noise[idx];
                if (a.x_t && hy >= 3 && hy < 3 + TH && hx >= 3 && hx < 3 + TW) a.x_t[idx] = v;
            }
            halo[(c * HH + hy) * (HW_ + 1) + hx] = v;
        }
        __syncthreads();
        // im2col panel: 2 threads per pixel, each builds 10 of the 20 8-element vectors of its row.  The K range of a
        // thread is a compile-time constant (template on the half) so tap/channel decoding costs no integer division.
        {
            const int pix = tid >> 1;
            const int py = pix / TW, px = pix % TW;
            const float* hb = halo + py * (HW_ + 1) + px;
            T* dst = Ap + pix * KROW;
            if (tid & 1) build_rows<T, 1>(hb, dst);
            else build_rows<T, 0>(hb, dst);
        }
        __syncthreads();
        // MFMA: wave w owns pixels 32w..32w+31 (tile rows 2w, 2w+1); A fragments are read once and kept in registers
        Frag<T> fa[KPAD / 16];
#pragma unroll
        for (int kk = 0; kk < KPAD / 16; kk++) frag_load<T>(fa[kk], Ap + (wave * 32 + r) * KROW + kk * 16 + hf * 8);
        __syncthreads();  // Ap may now be overwritten by the output staging
        for (int n0 = 0; n0 < a.Cout; n0 += 32) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            const bool nv = n0 + r < a.Cout;
#pragma unroll
            for (int kk = 0; kk < KPAD / 16; kk++) {
                Frag<T> fb;
                if (nv) frag_load<T>(fb, Wp + (n0 + r) * KROW + kk * 16 + hf * 8);
                else frag_zero<T>(fb);
                mma16<T>(acc, fa[kk], fb);
            }
            const float bias = nv ? a.bias[n0 + r] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) ot[(wave * 32 + acc_row(e, lane)) * 36 + r] = acc[e] + bias;
            __syncthreads();
            for (int idx = tid; idx < 128 * 4; idx += 256) {
                const int pix = idx >> 2, v8 = idx & 3;
                const int Y = ty0 + pix / TW, X = tx0 + pix % TW;
                const int n = n0 + v8 * 8;
                if (Y < a.H && X < a.W && n < a.Cout) {
                    const float* sp = ot + pix * 36 + v8 * 8;
                    Frag<T> f;
#pragma unroll
                    for (int j = 0; j < 8; j++) frag_set<T>(f, j, sp[j]);
                    frag_store<T>(f, (T*)a.out + (((size_t)b * a.H + Y) * a.W + X) * a.Cout + n);
                }
            }
            __syncthreads();
        }
    }
}

template <typename T> size_t stem_lds(int Cout) { return 3 * HH * (HW_ + 1) * 4 + 8 + (size_t)(128 + Cout) * KROW * sizeof(T); }

template <typename T> int launch(const dig_stem_args* a, hipStream_t stream) {
    const size_t lds = stem_lds<T>(a->Cout);
    DIG_CHECK(lds <= 160 * 1024, DIG_EINVAL, "qsample_stem: Cout=%d needs %zu B of LDS", a->Cout, lds);
    hipError_t e = hipFuncSetAttribute((const void*)stem_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "qsample_stem: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    const int tiles = cdiv(a->W, TW) * cdiv(a->H, TH);
    const int total = a->B * tiles;
    hipLaunchKernelGGL(stem_kernel<T>, dim3(total < 1024 ? total : 1024), dim3(256), lds, stream, *a);
    DIG_LAUNCH_CHECK("qsample_stem");
    return DIG_OK;
}

}  // namespace

extern "C" int dig_qsample_stem(const dig_stem_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->img && a->noise && a->t && a->sqrt_ac && a->sqrt_1mac && a->w && a->bias && a->out, DIG_EINVAL, "qsample_stem: null pointer");
    DIG_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && a->Cout % 8 == 0, DIG_EINVAL, "qsample_stem: bad shape (Cout=%d)", a->Cout);
    if (a->dtype == DIG_BF16) return launch<bf16>(a, stream);
    if (a->dtype == DIG_F32) return launch<float>(a, stream);
    DIG_CHECK(false, DIG_EINVAL, "qsample_stem: unknown dtype %d", a->dtype);
}
