// ResnetBlock tail with a channel-changing shortcut (ddp.py:190-194), streaming form for the high-resolution levels:
//     out[p][:] = res_conv([x0 | x1][p][:]) + SiLU(a . h2[p][:] + b)          (a, b = GroupNorm-apply of block2, per image)
// At 256x256 / 128x128 this is four tensor passes around a tiny GEMM (K <= 192, N <= 128): HBM-bound.  The general GEMM
// (gemm.hip) pays a block-wide pipeline fill and two barriers per 32-deep K step for it; here the 1x1 weights stay in
// LDS for the block's lifetime and every wave streams 32-pixel sub-tiles on its own - no block barrier in the loop:
//   rows of x0 | x1 and of h2 -> registers (prefetched one sub-tile ahead) -> the wave's own LDS rows,
//   y[c][p] on MFMA with pixels as COLUMNS (A = weights, B = pixel rows), accumulators start from the bias,
//   + SiLU(GroupNorm(h2)) read back from the LDS rows in the accumulator's own layout (4 consecutive channels = 8 bytes),
//   result written in place over the h2 rows and copied out as 16-byte NHWC vectors.
// bf16 storage only (the fp32 parity mode keeps the general GEMM).
#include <cstdlib>

#include "common.hpp"

namespace {

// FIN: the block is the Unet's final_res_block and the 1x1 final_conv (ddp.py:373,443-448: N -> f_no <= 4 channels, fp32 NCHW) is
// applied to the block output while its rows sit in the wave's LDS tile: the output never goes to HBM (at 256x256 a 268 MB
// write + read and one launch).
template <int KS, int NT, bool FIN>
__global__ __launch_bounds__(256, (NT <= 2 ? 2 : 1)) void pw_res_kernel(dig_pw_res_args a, int iters) {
    constexpr int K = KS * 16, N = NT * 32, KP = K + 8, NP = N + 8;
    constexpr int NVX = K / 16, NVG = N / 16;  // 16-byte vectors per lane per sub-tile: x rows, h2 rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* ws = reinterpret_cast<bf16*>(smem);                       // [N][KP]
    float* bias_l = reinterpret_cast<float*>(ws + N * KP);          // [N]
    float* aff_l = bias_l + N;                                      // [4 waves][N][2]
    bf16* xs_all = reinterpret_cast<bf16*>(aff_l + 4 * N * 2);      // [4 waves][32][KP]
    bf16* gs_all = xs_all + 4 * 32 * KP;                            // [4 waves][32][NP]
    float* fw_l = reinterpret_cast<float*>(gs_all + 4 * 32 * NP);   // FIN: [4][N] final_conv weights (rows >= f_no zero)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const bf16* x0 = (const bf16*)a.x0;
    const bf16* x1 = (const bf16*)a.x1;
    const bf16* gsrc = (const bf16*)a.gsrc;
    bf16* out = (bf16*)a.out;
    const int K0 = a.K0, K1 = K - K0;

    for (int i = tid; i < N * (K / 8); i += 256) {
        const int n = i / (K / 8), v = i - n * (K / 8);
        *reinterpret_cast<bf16x8*>(ws + n * KP + v * 8) = *reinterpret_cast<const bf16x8*>((const bf16*)a.w + (size_t)n * K + v * 8);
    }
    for (int i = tid; i < N; i += 256) bias_l[i] = a.bias ? a.bias[i] : 0.f;
    if constexpr (FIN)
        for (int i = tid; i < 4 * N; i += 256) fw_l[i] = i < a.f_no * N ? a.fw[i] : 0.f;
    __syncthreads();

    bf16* xs = xs_all + wave * 32 * KP;
    bf16* gs = gs_all + wave * 32 * NP;
    float* aff = aff_l + wave * N * 2;
    const long long nsub = a.M / 32;
    // Sub-tiles are dealt round-robin over the blocks (iteration `it` of block b takes group it * gridDim + b): at any moment the
    // resident blocks sweep ONE window of a few MB per stream instead of 500 separate 256 KB runs - the HBM pages stay open
    // (tools/lab/hbm_bw: 3R+1W streams reach 5.0-5.8 TB/s swept as windows; block-contiguous runs held this kernel at 4.0).
    const long long sub0 = (long long)blockIdx.x * 4 + wave, sub_step = (long long)gridDim.x * 4;

    // register sets of the prefetch (PD2: two sub-tiles ahead instead of one)
    bf16x8 pxa[NVX], pga[NVG], pxb[NVX], pgb[NVG];
    auto gload = [&](long long sub, bf16x8(&px)[NVX], bf16x8(&pg)[NVG]) {
        const long long row0 = sub * 32;
#pragma unroll
        for (int i = 0; i < NVX; i++) {
            const int idx = lane + i * 64, row = idx / (K / 8), v = idx - row * (K / 8);
            const int k = v * 8;
            const bf16* src = k < K0 ? x0 + (size_t)(row0 + row) * K0 + k : x1 + (size_t)(row0 + row) * K1 + (k - K0);
            px[i] = *reinterpret_cast<const bf16x8*>(src);
        }
#pragma unroll
        for (int i = 0; i < NVG; i++) {
            const int idx = lane + i * 64, row = idx / (N / 8), v = idx - row * (N / 8);
            pg[i] = *reinterpret_cast<const bf16x8*>(gsrc + (size_t)(row0 + row) * N + v * 8);
        }
    };
    int cur_img = -1;
    if (sub0 < nsub) gload(sub0, pxa, pga);
    // Two sub-tiles ahead (PD2) was measured and is off: 219 vs 216 us at N = 64, +12 % at K = 192, +4 % on the fused-final-conv
    // instantiation; removing the SiLU and the MFMAs altogether changes the N = 64 launches by 2 % (ablation builds): at
    // 256x256 the kernel runs at the rate a plain 3-read + 1-write stream reaches (tools/lab/hbm_bw: 5.0-5.8 TB/s).
    constexpr bool PD2 = false;
    if (PD2 && iters > 1 && sub0 + sub_step < nsub) gload(sub0 + sub_step, pxb, pgb);
    auto step = [&](int it, bf16x8(&px)[NVX], bf16x8(&pg)[NVG]) {
        const long long sub = sub0 + (long long)it * sub_step;
        if (it >= iters || sub >= nsub) return;
        const int img = ((int)sub * 32) / a.rows_per_img;
        __builtin_amdgcn_wave_barrier();
        if (img != cur_img) {  // wave-uniform: refresh this wave's GroupNorm affine table
            for (int i = lane; i < N * 2 / 4; i += 64)
                *reinterpret_cast<f32x4*>(aff + i * 4) = *reinterpret_cast<const f32x4*>(a.gaff + (size_t)img * N * 2 + i * 4);
            cur_img = img;
        }
#pragma unroll
        for (int i = 0; i < NVX; i++) {
            const int idx = lane + i * 64, row = idx / (K / 8), v = idx - row * (K / 8);
            *reinterpret_cast<bf16x8*>(xs + row * KP + v * 8) = px[i];
        }
#pragma unroll
        for (int i = 0; i < NVG; i++) {
            const int idx = lane + i * 64, row = idx / (N / 8), v = idx - row * (N / 8);
            *reinterpret_cast<bf16x8*>(gs + row * NP + v * 8) = pg[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (PD2) {
            if (it + 2 < iters && sub + 2 * sub_step < nsub) gload(sub + 2 * sub_step, px, pg);
        } else {
            if (it + 1 < iters && sub + sub_step < nsub) gload(sub + sub_step, px, pg);
        }

        const bf16* wsp = ws;
        if constexpr (FIN) {  // keep the 64 registers of weight fragments from being hoisted out of the loop (FIN has none to spare)
            int opq = 0;
            asm volatile("" : "+s"(opq));
            wsp = ws + opq;
        }
        f32x16 acc[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + ct * 32 + 8 * q4 + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; j++) acc[ct][4 * q4 + j] = bv[j];
            }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            Frag<bf16> fb;
            frag_load<bf16>(fb, xs + r * KP + ks * 16 + hf * 8);
#pragma unroll
            for (int ct = 0; ct < NT; ct++) {
                Frag<bf16> fa;
                frag_load<bf16>(fa, wsp + (ct * 32 + r) * KP + ks * 16 + hf * 8);
                mma16<bf16>(acc[ct], fa, fb);
            }
        }
        // + SiLU(a h2 + b), in place over the h2 rows (pixel r of this lane, channels ct*32 + 8 q4 + 4 hf + 0..3)
#pragma unroll
        for (int ct = 0; ct < NT; ct++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const int c0 = ct * 32 + 8 * q4 + 4 * hf;
                bf16* gp = gs + r * NP + c0;
                const bf16x4 g = *reinterpret_cast<const bf16x4*>(gp);
                const f32x4 ab0 = *reinterpret_cast<const f32x4*>(aff + c0 * 2), ab1 = *reinterpret_cast<const f32x4*>(aff + c0 * 2 + 4);
                const float v0 = acc[ct][4 * q4 + 0] + silu_f(ab0[0] * (float)g[0] + ab0[1]);
                const float v1 = acc[ct][4 * q4 + 1] + silu_f(ab0[2] * (float)g[1] + ab0[3]);
                const float v2 = acc[ct][4 * q4 + 2] + silu_f(ab1[0] * (float)g[2] + ab1[1]);
                const float v3 = acc[ct][4 * q4 + 3] + silu_f(ab1[2] * (float)g[3] + ab1[3]);
                *reinterpret_cast<bf16x4*>(gp) = bf16x4{(bf16)v0, (bf16)v1, (bf16)v2, (bf16)v3};
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (FIN) {
            // final_conv on the block output as stored (bf16 rows in LDS, the accumulators are dead by now): lane (pixel r, half hf)
            // takes channels hf*N/2 .. +N/2-1, one xor-32 shuffle completes the dot products; fout[img][o][pixel], lanes of half hf
            // store planes o = hf and hf + 2 (32 consecutive pixels = 128 bytes each)
            float fd[4] = {0.f, 0.f, 0.f, 0.f};
            // (an opaque zero ties the weight reads to the iteration: left loop-invariant, all 128 of them are hoisted into
            // registers and the kernel spills)
            int opq = 0;
            asm volatile("" : "+s"(opq));
            const float* fwp = fw_l + opq;
#pragma unroll
            for (int v8 = 0; v8 < N / 16; v8++) {
                const int c0 = hf * (N / 2) + v8 * 8;
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(gs + r * NP + c0);
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const f32x4 wa = *reinterpret_cast<const f32x4*>(fwp + o * N + c0), wb = *reinterpret_cast<const f32x4*>(fwp + o * N + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) fd[o] = fmaf(wa[j], (float)g[j], fd[o]);
#pragma unroll
                    for (int j = 0; j < 4; j++) fd[o] = fmaf(wb[j], (float)g[4 + j], fd[o]);
                }
            }
            const int row = (int)sub * 32 + r;  // M < 2^31 (checked on the host)
            const int im = row / a.rows_per_img, pix = row - im * a.rows_per_img;
#pragma unroll
            for (int o = 0; o < 4; o++) fd[o] += __shfl_xor(fd[o], 32, 64);
#pragma unroll
            for (int oo = 0; oo < 2; oo++) {
                const int o = hf + 2 * oo;
                if (o < a.f_no) a.fout[((size_t)im * a.f_no + o) * a.rows_per_img + pix] = (hf ? fd[1 + 2 * oo] : fd[2 * oo]) + a.fb[o];
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < NVG; i++) {
            const int idx = lane + i * 64, row = idx / (N / 8), v = idx - row * (N / 8);
            *reinterpret_cast<bf16x8*>(out + (size_t)(sub * 32 + row) * N + v * 8) = *reinterpret_cast<const bf16x8*>(gs + row * NP + v * 8);
        }
    };
    if constexpr (PD2) {
        for (int it = 0; it < iters; it += 2) {
            step(it, pxa, pga);
            step(it + 1, pxb, pgb);
        }
    } else {
        for (int it = 0; it < iters; it++) step(it, pxa, pga);
    }
}

template <int KS, int NT> size_t pw_lds() {
    constexpr int K = KS * 16, N = NT * 32;
    return (size_t)N * (K + 8) * 2 + (size_t)N * 4 + (size_t)4 * N * 2 * 4 + (size_t)4 * 32 * (K + 8) * 2 + (size_t)4 * 32 * (N + 8) * 2 + (size_t)4 * N * 4;
}

template <int KS, int NT, bool FIN = false> int launch(const dig_pw_res_args* a, hipStream_t st) {
    const size_t lds = pw_lds<KS, NT>();
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)pw_res_kernel<KS, NT, FIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "pw_res: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
        attr_set[dev].store(true, std::memory_order_release);
    }
    const long long nsub = a->M / 32;
    const long long groups = (nsub + 3) / 4;  // 4 sub-tiles (one per wave) per block-iteration
    // sub-tile groups per block: the N = 128 instantiation (one block per CU, 51 KB of weights to stage per block) wants two
    // rounds of blocks, not four (measured at M = 524288: 135 -> 117 us; the N = 64 one is flat from 4 to 16 iterations)
    int iters = (int)(groups / (NT > 2 ? 512 : 1024));
    iters = iters < 1 ? 1 : (iters > 16 ? 16 : iters);
#ifdef PW_LAB
    if (const char* e = getenv("DIG_PW_ITERS")) iters = atoi(e);
#endif
    const long long grid = (groups + iters - 1) / iters;
    hipLaunchKernelGGL((pw_res_kernel<KS, NT, FIN>), dim3((unsigned)grid), dim3(256), lds, st, *a, iters);
    DIG_LAUNCH_CHECK("pw_res");
    return DIG_OK;
}

}  // namespace

extern "C" int dig_pw_res_supported(int K0, int K1, int N, int dtype) {
    if (dtype != DIG_BF16 || K0 <= 0 || K0 % 8 != 0 || K1 < 0 || K1 % 8 != 0) return 0;
    const int K = K0 + K1;
    return (K == 128 && N == 64) || (K == 192 && N == 128) ? 1 : 0;
}

extern "C" int dig_pw_res(const dig_pw_res_args* a, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    DIG_CHECK(a && a->x0 && a->w && a->gsrc && a->gaff && (a->out || a->fout), DIG_EINVAL, "pw_res: null pointer");
    if (a->fout) {
        DIG_CHECK(a->fw && a->fb && a->f_no >= 1 && a->f_no <= 4, DIG_EINVAL, "pw_res: the fused final conv needs fw, fb and 1 <= f_no <= 4 (got %d)", a->f_no);
        DIG_CHECK(a->K0 + a->K1 == 128 && a->N == 64, DIG_EINVAL, "pw_res: the fused final conv is built for K0+K1 = 128, N = 64");
        DIG_CHECK(((uintptr_t)a->fw | (uintptr_t)a->fb | (uintptr_t)a->fout) % 4 == 0, DIG_EALIGN, "pw_res: fw / fb / fout must be 4-byte aligned");
    }
    DIG_CHECK(a->K1 == 0 || a->x1, DIG_EINVAL, "pw_res: K1 > 0 needs x1");
    DIG_CHECK(dig_pw_res_supported(a->K0, a->K1, a->N, a->dtype), DIG_EINVAL, "pw_res: unsupported shape K0=%d K1=%d N=%d dtype=%d", a->K0, a->K1, a->N, a->dtype);
    DIG_CHECK(a->M > 0 && a->M < (1ll << 31) && a->M % 32 == 0 && a->rows_per_img > 0 && a->rows_per_img % 32 == 0, DIG_EINVAL, "pw_res: M and rows_per_img must be multiples of 32");
    DIG_CHECK(((uintptr_t)a->x0 | (uintptr_t)a->x1 | (uintptr_t)a->w | (uintptr_t)a->gsrc | (uintptr_t)a->gaff | (uintptr_t)a->out) % 16 == 0, DIG_EALIGN,
              "pw_res: pointers must be 16-byte aligned");
    if (a->fout) return launch<8, 2, true>(a, st);
    if (a->K0 + a->K1 == 128) return launch<8, 2>(a, st);
    return launch<12, 4>(a, st);
}
