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
                const float x0 = __builtin_fmaf(a.pre_scale[c], a.img[idx], a.pre_shift[c]);
                v = axpby_rn(ca, x0, cs, a.noise[idx]);  // as stem64_kernel and dig_sched_axpby: x_t is bit-identical everywhere
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


// ---- bf16, Cout == 64 (the DiG Unet): no im2col panel ----------------------------------------------------------------
// The panel kernel above spends its time building [128][160] im2col rows with scalar LDS traffic behind five block
// barriers per 128 pixels.  Here the halo tile is kept as [row][col][4] bf16 (c = 3 is zero): with K ordered
// k = ky*32 + kx*4 + c (kx = 7 and c = 3 carry zero weights, K = 224 = 14 MFMA steps) the 8 k-values a lane feeds to one
// step are the 16 contiguous bytes of two neighbouring halo pixels - the fragment is read straight from the halo tile
// (8-byte aligned: two ds_read_b64), nothing is materialised.  16x32-pixel tile per 256-thread block, wave w owns tile
// rows 4w..4w+3 (four 32-pixel accumulator tiles x two 32-channel tiles); the weights are the MFMA A operand, so a lane
// ends up with 4 consecutive channels of ONE pixel and the epilogue goes through the wave's own LDS rows (no block
// barrier) into 16-byte NHWC stores.  Persistent blocks; the next tile's pixels are prefetched into registers under the
// current tile's MFMAs and stored into the other halo buffer: one block barrier per tile.
constexpr int S_TH = 16, S_TW = 32, S_HH = S_TH + 6, S_HC = 40;  // halo 22 rows x 40 columns (38 real + kx = 7 overhang)
constexpr int S_K = 224, S_WROW = S_K + 8;                       // weight row stride 464 B = 29 x 16 B (odd: conflict-free)
constexpr int S_NPX = (S_HH * S_HC + 255) / 256;                 // halo pixels per thread: 4
constexpr int S_OROW = 64 + 8;                                   // epilogue staging row (bf16 elements)
constexpr int S_LDS = 64 * S_WROW * 2 + 2 * S_HH * S_HC * 8 + 4 * 32 * S_OROW * 2;

__global__ __launch_bounds__(256, 2) void stem64_kernel(dig_stem_args a, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* wl = reinterpret_cast<bf16*>(smem);                                     // [64][S_WROW]
    unsigned char* hbuf = smem + 64 * S_WROW * 2;                                 // 2 x [22][40][4] bf16
    bf16* ostage = reinterpret_cast<bf16*>(hbuf + 2 * S_HH * S_HC * 8);           // [4 waves][32][S_OROW]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = a.B * tiles_x * tiles_y;
    const size_t HW = (size_t)a.H * a.W;

    // weight panel: [n][ky][kx 0..7][c 0..3] from the packed [n][(ky*7+kx)*3+c] rows, zero in the padding slots
    for (int i = tid; i < 64 * 7 * 8; i += 256) {
        const int n = i / 56, rem = i - n * 56, ky = rem >> 3, kx = rem & 7;
        bf16x4 v = bf16x4{0, 0, 0, 0};
        if (kx < 7) {
            const bf16* src = (const bf16*)a.w + (size_t)n * 160 + (ky * 7 + kx) * 3;
            v = bf16x4{src[0], src[1], src[2], (bf16)0.f};
        }
        *reinterpret_cast<bf16x4*>(wl + n * S_WROW + ky * 32 + kx * 4) = v;
    }

    float pimg[S_NPX][3], pnz[S_NPX][3];
    auto gload = [&](int tidx) {
        const int b = tidx / (tiles_x * tiles_y), tile = tidx - b * (tiles_x * tiles_y);
        const int ty0 = (tile / tiles_x) * S_TH, tx0 = (tile % tiles_x) * S_TW;
#pragma unroll
        for (int i = 0; i < S_NPX; i++) {
            const int idx = tid + i * 256, hy = idx / S_HC, hx = idx - hy * S_HC;
            const int Y = ty0 + hy - 3, X = tx0 + hx - 3;
            const bool ok = idx < S_HH * S_HC && Y >= 0 && Y < a.H && X >= 0 && X < a.W;
            const size_t o = (size_t)b * 3 * HW + (size_t)(ok ? Y : 0) * a.W + (ok ? X : 0);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pimg[i][c] = ok ? a.img[o + c * HW] : 0.f;
                pnz[i][c] = ok ? a.noise[o + c * HW] : 0.f;
            }
        }
    };
    // q_sample in registers, x_t out for the tile's own pixels, halo pixel -> one 8-byte LDS store
    auto hstore = [&](int tidx, unsigned char* hb) {
        const int b = tidx / (tiles_x * tiles_y), tile = tidx - b * (tiles_x * tiles_y);
        const int ty0 = (tile / tiles_x) * S_TH, tx0 = (tile % tiles_x) * S_TW;
        const long long t = a.t[b];
        const float ca = a.sqrt_ac[t], cs = a.sqrt_1mac[t];
#pragma unroll
        for (int i = 0; i < S_NPX; i++) {
            const int idx = tid + i * 256, hy = idx / S_HC, hx = idx - hy * S_HC;
            if (idx >= S_HH * S_HC) continue;
            const int Y = ty0 + hy - 3, X = tx0 + hx - 3;
            const bool ok = Y >= 0 && Y < a.H && X >= 0 && X < a.W;
            float v[3];
#pragma unroll
            // explicitly rounded operations (ddp.py:857-860 is two multiplications and an addition): this lambda is inlined at
            // two call sites (first tile / steady state) and under -ffp-contract=fast the two copies would otherwise contract
            // differently, i.e. round a halo pixel differently from the tile that owns it
            for (int c = 0; c < 3; c++)
                v[c] = ok ? axpby_rn(ca, __builtin_fmaf(a.pre_scale[c], pimg[i][c], a.pre_shift[c]), cs, pnz[i][c]) : 0.f;
            if (a.x_t && ok && hy >= 3 && hy < 3 + S_TH && hx >= 3 && hx < 3 + S_TW) {
                const size_t o = (size_t)b * 3 * HW + (size_t)Y * a.W + X;
#pragma unroll
                for (int c = 0; c < 3; c++) a.x_t[o + c * HW] = v[c];
            }
            *reinterpret_cast<bf16x4*>(hb + (size_t)idx * 8) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)0.f};
        }
    };

    int tidx = blockIdx.x;
    if (tidx < ntiles) {
        gload(tidx);
        hstore(tidx, hbuf);
    }
    __syncthreads();
    bf16* ow = ostage + wave * 32 * S_OROW;
    for (int it = 0; tidx < ntiles; tidx += gridDim.x, it++) {
        const unsigned char* hb = hbuf + (it & 1) * (S_HH * S_HC * 8);
        unsigned char* hn = hbuf + ((it & 1) ^ 1) * (S_HH * S_HC * 8);
        const int nxt = tidx + gridDim.x;
        if (nxt < ntiles) gload(nxt);
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 14; s++) {
            const int ky = s >> 1, half = s & 1;
            Frag<bf16> fw[2];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) frag_load<bf16>(fw[nt], wl + (nt * 32 + r) * S_WROW + s * 16 + hf * 8);
#pragma unroll
            for (int mr = 0; mr < 4; mr++) {
                const unsigned char* hp = hb + ((wave * 4 + mr + ky) * S_HC + r + half * 4 + hf * 2) * 8;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(hp), hi = *reinterpret_cast<const bf16x4*>(hp + 8);
                Frag<bf16> fx;
                fx.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int nt = 0; nt < 2; nt++) mma16<bf16>(acc[mr][nt], fw[nt], fx);
            }
        }
        if (nxt < ntiles) hstore(nxt, hn);
        // epilogue, per wave: + bias, this wave's four pixel rows through its own LDS rows, 16-byte NHWC stores
        const int b = tidx / (tiles_x * tiles_y), tile = tidx - b * (tiles_x * tiles_y);
        const int ty0 = (tile / tiles_x) * S_TH, tx0 = (tile % tiles_x) * S_TW;
#pragma unroll
        for (int mr = 0; mr < 4; mr++) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const int c0 = nt * 32 + 8 * q4 + 4 * hf;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c0);
                    *reinterpret_cast<bf16x4*>(ow + r * S_OROW + c0) =
                        bf16x4{(bf16)(acc[mr][nt][4 * q4] + bv[0]), (bf16)(acc[mr][nt][4 * q4 + 1] + bv[1]),
                               (bf16)(acc[mr][nt][4 * q4 + 2] + bv[2]), (bf16)(acc[mr][nt][4 * q4 + 3] + bv[3])};
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int Y = ty0 + wave * 4 + mr;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int idx = lane + i * 64, px = idx >> 3, v8 = idx & 7;
                const int X = tx0 + px;
                if (Y < a.H && X < a.W)
                    *reinterpret_cast<bf16x8*>((bf16*)a.out + (((size_t)b * a.H + Y) * a.W + X) * 64 + v8 * 8) =
                        *reinterpret_cast<const bf16x8*>(ow + px * S_OROW + v8 * 8);
            }
        }
        __syncthreads();  // the other halo buffer is complete; this one may be overwritten next time round
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
    if (a->dtype == DIG_BF16 && a->Cout == 64 && ((uintptr_t)a->bias % 16) == 0) {
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev &= 63;
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute((const void*)stem64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);
            DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "qsample_stem: cannot raise dynamic LDS: %s", hipGetErrorString(e));
            attr_set[dev].store(true, std::memory_order_release);
        }
        const int tx = cdiv(a->W, S_TW), ty = cdiv(a->H, S_TH);
        const int total = a->B * tx * ty;
        hipLaunchKernelGGL(stem64_kernel, dim3(total < 512 ? total : 512), dim3(256), S_LDS, stream, *a, tx, ty);
        DIG_LAUNCH_CHECK("qsample_stem");
        return DIG_OK;
    }
    if (a->dtype == DIG_BF16) return launch<bf16>(a, stream);
    if (a->dtype == DIG_F32) return launch<float>(a, stream);
    DIG_CHECK(false, DIG_EINVAL, "qsample_stem: unknown dtype %d", a->dtype);
}
