// conv3x3 (pad 1) as an implicit GEMM on MFMA, NHWC, gfx950.
//
// One 256-thread block (4 waves) computes a 16x16-pixel tile x BN output channels.  Per K-chunk of KC
// input channels it stages ONE 18x18 halo tile of the input in LDS (each input element is fetched once and
// reused by all 9 taps - no im2col traffic) plus the [9][BN][KC] weight slab, then runs 9*KC/16 MFMA steps
// whose A operands are plain shifted reads of the halo tile.  The loader can (a) read a channel concat of two
// tensors, (b) read the input through nearest-x2 addressing and (c) apply the previous Block's
// GroupNorm-apply + SiLU on the fly, so none of those ever makes its own HBM pass.  The epilogue adds the
// bias, writes the raw conv output and emits per-tile (sum, sumsq) partials for the next GroupNorm.
//
// LDS pixel stride is KC+8 elements and the halo row stride is padded to a multiple of 256 B so that the
// 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots (conflict-free A and B fragment reads).
#include "common.hpp"
#include <cstdlib>

// weight-stationary persistent variant for 64 -> 64 bf16 layers (conv3x3_ws.hip)
bool conv3x3_ws64_eligible(int H, int W, int C0, int C1, int Cout, int groups, int dtype);
int conv3x3_ws64_launch(const dig_conv3x3_args* a, hipStream_t stream);
// double-buffered 8-wave variant for the wide bf16 layers (conv3x3_db.hip): 32x16-pixel tiles
bool conv3x3_rs128_eligible(int H, int W, int C0, int C1, int Cout, int groups, int dtype, bool has_aff);
int conv3x3_rs128_launch(const dig_conv3x3_args* a, hipStream_t stream);
bool conv3x3_db_eligible(int C0, int C1, int Cout, int dtype);
int conv3x3_db_launch(const dig_conv3x3_args* a, hipStream_t stream);

namespace {

constexpr int TILE_W = 16;          // tile width; tile height = 4 rows per pixel-group wave
constexpr int HALO_W = TILE_W + 2;  // 18
#ifndef CONV_PF64
#define CONV_PF64 true
#endif
#ifndef CONV_PF128
#define CONV_PF128 false
#endif
#ifndef CONV_WNG128
#define CONV_WNG128 1
#endif
#ifndef CONV_WPG64
#define CONV_WPG64 4
#endif
#ifndef CONV_KC64
#define CONV_KC64 16
#endif

// WPG = waves along pixels (each owns 4 tile rows = 64 pixels), WNG = waves along output channels
template <typename T, int KC, int WPG> struct ConvCfg {
    static constexpr int KCP = KC + 8;  // padded per-pixel channel stride (elements)
    static constexpr int TILE_H = 4 * WPG, HALO_H = TILE_H + 2;
    static constexpr int HROW = ((HALO_W * KCP * (int)sizeof(T) + 255) / 256) * 256 / (int)sizeof(T);
    static constexpr int HALO_ELEMS = HALO_H * HROW;
};

template <typename T, int BN, int KC, int WPG> constexpr size_t conv_lds_bytes() {
    using C = ConvCfg<T, KC, WPG>;
    size_t main = (size_t)(C::HALO_ELEMS + 9 * BN * C::KCP) * sizeof(T);
    size_t epi = (size_t)(WPG * 32) * (BN + 4) * sizeof(float);          // output staging, half of the tile at a time
    size_t st = (size_t)(WPG * 2 * BN * 2 + BN * 2) * sizeof(float);     // GroupNorm partials
    size_t m = main > epi ? main : epi;
    return m > st ? m : st;
}

template <typename T> struct Params {
    const T* in[2];
    int C[2];
    const float* aff0;
    const T* w;  // packed [n_tile][chunk][9][BN][KC]
    const float* bias;
    T* out;
    float* stats;
    long long bstride0;
    int B, H, W, Cout, groups, up2, tiles_x, tiles_y;
};

template <typename T, int BN, int KC, bool PF, int WPG, int WNG>
#ifndef CONV_MINW
#define CONV_MINW(WPG, WNG, BN) ((WPG * WNG) / 2)
#endif
__global__ __launch_bounds__(64 * WPG * WNG, CONV_MINW(WPG, WNG, BN)) void conv3x3_kernel(Params<T> p) {
    using C = ConvCfg<T, KC, WPG>;
    constexpr int NTHREADS = 64 * WPG * WNG;
    constexpr int KCP = C::KCP, HROW = C::HROW, TILE_H = C::TILE_H, HALO_H = C::HALO_H;
    constexpr int NR = BN / 32 / WNG, MR = 2;
    constexpr int V = KC / 8;  // 8-element vectors per pixel per chunk
    constexpr int NH = (HALO_H * HALO_W * V + NTHREADS - 1) / NTHREADS;  // halo vectors per thread per chunk
    constexpr int NW = (9 * BN * V + NTHREADS - 1) / NTHREADS;       // weight vectors per thread per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* halo = reinterpret_cast<T*>(smem_raw);
    T* wl = halo + C::HALO_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pg = wave % WPG, ng = wave / WPG;  // pixel group / channel group of this wave
    const int r = lane & 31, hf = lane >> 5;
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / p.tiles_x) * TILE_H, tx0 = (tile % p.tiles_x) * TILE_W;
    const int n0 = blockIdx.y * BN;
    const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;  // source dims
    const int nch0 = (p.C[0] + KC - 1) / KC, nch1 = (p.C[1] + KC - 1) / KC, nch = nch0 + nch1;
    const T* wbase = p.w + (size_t)blockIdx.y * nch * (9 * BN * KC);

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < NR; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int c8 = tid % V;  // this thread's vector slot inside a chunk (constant: 256 % V == 0)
    Frag<T> hreg[NH], wreg[NW];
    unsigned hok = 0;  // bit i: hreg[i] holds real (in-bounds) data

    // global -> registers for chunk ci (weights: one contiguous pre-packed slab, perfectly coalesced)
    auto gload = [&](int ci) {
        const int piece = ci >= nch0 ? 1 : 0;
        const int cb = (ci - (piece ? nch0 : 0)) * KC;
        const int Cp = p.C[piece];
        const T* src = p.in[piece];
        const size_t bs = piece ? (size_t)Hs * Ws * Cp : (size_t)p.bstride0;
        const int ch = cb + c8 * 8;
        const bool chv = ch < Cp;
        hok = 0;
#pragma unroll
        for (int i = 0; i < NH; i++) {
            const int idx = tid + i * NTHREADS;
            const int pix = idx / V;
            const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
            const int Y = ty0 + hy - 1, X = tx0 + hx - 1;
            const bool ok = (idx < HALO_H * HALO_W * V) && chv && Y >= 0 && Y < p.H && X >= 0 && X < p.W;
#ifdef ABL_NO_GLOAD
            if (ok && p.B == 12345) {
#else
            if (ok) {
#endif
                const int sy = p.up2 ? Y >> 1 : Y, sx = p.up2 ? X >> 1 : X;
                frag_load<T>(hreg[i], src + (size_t)b * bs + ((size_t)sy * Ws + sx) * Cp + ch);
                hok |= 1u << i;
            } else {
                frag_zero<T>(hreg[i]);
            }
        }
        const T* slab = wbase + (size_t)ci * (9 * BN * KC);
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int idx = tid + i * NTHREADS;
#ifdef ABL_NO_GLOAD
            if (idx < 9 * BN * V && p.B == 12345) frag_load<T>(wreg[i], slab + (size_t)idx * 8);
#elif defined(ABL_NO_WLOAD)
            if (idx < 9 * BN * V && ci == 0) frag_load<T>(wreg[i], slab + (size_t)idx * 8);
#else
            if (idx < 9 * BN * V) frag_load<T>(wreg[i], slab + (size_t)idx * 8);
#endif
        }
    };
    // registers -> LDS (GroupNorm-apply + SiLU prologue applied here, once per input element)
    auto lstore = [&](int ci) {
        const bool pro = (ci < nch0) && (p.aff0 != nullptr);
        if (pro) {
            const int ch = ci * KC + c8 * 8;
            if (ch < p.C[0]) {
                const float* af = p.aff0 + ((size_t)b * p.C[0] + ch) * 2;
                float pa[8], pb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    pa[j] = af[2 * j];
                    pb[j] = af[2 * j + 1];
                }
#pragma unroll
                for (int i = 0; i < NH; i++)
                    if (hok & (1u << i)) {
#pragma unroll
                        for (int j = 0; j < 8; j++) frag_set<T>(hreg[i], j, silu_f(pa[j] * frag_get<T>(hreg[i], j) + pb[j]));
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < NH; i++) {
            const int idx = tid + i * NTHREADS;
            if (idx < HALO_H * HALO_W * V) {
                const int pix = idx / V;
                const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
                frag_store<T>(hreg[i], halo + hy * HROW + hx * KCP + c8 * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int idx = tid + i * NTHREADS;
            if (idx < 9 * BN * V) frag_store<T>(wreg[i], wl + (idx / V) * KCP + c8 * 8);
        }
    };

    // One flat loop over the K-chunks of both concat pieces (a nested piece/chunk loop made hipcc shuttle an
    // accumulator register through a VGPR copy at the inner loop's exit, read before the last MFMA had retired).
    if (PF) gload(0);
    for (int ci = 0; ci < nch; ci++) {
        if (!PF) gload(ci);
        __syncthreads();  // previous chunk's MFMA reads are done
        lstore(ci);
        __syncthreads();
        if (PF && ci + 1 < nch) gload(ci + 1);  // next chunk's HBM/L2 latency hides under this chunk's MFMAs
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int kk = 0; kk < KC / 16; kk++) {
                Frag<T> fa[MR], fb[NR];
#pragma unroll
                for (int mr = 0; mr < MR; mr++) {
                    const int py = pg * 4 + mr * 2 + (r >> 4), px = r & 15;
                    frag_load<T>(fa[mr], halo + (py + ky) * HROW + (px + kx) * KCP + kk * 16 + hf * 8);
                }
#pragma unroll
                for (int nr = 0; nr < NR; nr++)
                    frag_load<T>(fb[nr], wl + (tap * BN + (ng * NR + nr) * 32 + r) * KCP + kk * 16 + hf * 8);
#pragma unroll
                for (int mr = 0; mr < MR; mr++)
#pragma unroll
                    for (int nr = 0; nr < NR; nr++) {
#ifndef ABL_NO_MFMA
                        mma16<T>(acc[mr][nr], fa[mr], fb[nr]);
#else
                        acc[mr][nr][0] += frag_get<T>(fa[mr], 0) * frag_get<T>(fb[nr], 0);
#endif
                    }
            }
        }
    }
    // Let the last MFMA retire before anything reads the accumulators (do not rely on the hazard recognizer across
    // the loop exit), and fence LDS for the epilogue.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    // ---- epilogue: +bias, GroupNorm partials from the fp32 accumulators, and the output tile transposed through
    // LDS (fp32, one 128-pixel half at a time) so that HBM sees 16-byte stores of 8 consecutive channels.
    float* ot = reinterpret_cast<float*>(smem_raw);
    constexpr int OS = BN + 4;
    float cs[NR], css[NR];
#pragma unroll
    for (int nr = 0; nr < NR; nr++) cs[nr] = css[nr] = 0.f;
    const bool vec_ok = (p.Cout % 8) == 0;
#pragma unroll
    for (int mr = 0; mr < MR; mr++) {
#pragma unroll
        for (int nr = 0; nr < NR; nr++) {
            const int nl = (ng * NR + nr) * 32 + r;
            const int n = n0 + nl;
            const bool nv = n < p.Cout;
            const float bias = nv ? p.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = acc_row(e, lane);
                const int Y = ty0 + pg * 4 + mr * 2 + (row >> 4), X = tx0 + (row & 15);
                const float v = acc[mr][nr][e] + bias;
                ot[(pg * 32 + row) * OS + nl] = v;
                if (nv && Y < p.H && X < p.W) {
                    cs[nr] += v;
                    css[nr] += v * v;
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < WPG * 32 * (BN / 8); idx += NTHREADS) {
            const int pl = idx / (BN / 8), v8 = idx - pl * (BN / 8);
            const int wv = pl >> 5, row = pl & 31;
            const int Y = ty0 + wv * 4 + mr * 2 + (row >> 4), X = tx0 + (row & 15);
            const int n = n0 + v8 * 8;
            if (Y < p.H && X < p.W && n < p.Cout) {
                const float* sp = ot + pl * OS + v8 * 8;
                T* dst = p.out + (((size_t)b * p.H + Y) * p.W + X) * p.Cout + n;
                if (vec_ok) {
                    Frag<T> f;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 4);
                    frag_set<T>(f, 0, lo[0]); frag_set<T>(f, 1, lo[1]); frag_set<T>(f, 2, lo[2]); frag_set<T>(f, 3, lo[3]);
                    frag_set<T>(f, 4, hi[0]); frag_set<T>(f, 5, hi[1]); frag_set<T>(f, 6, hi[2]); frag_set<T>(f, 7, hi[3]);
#ifndef ABL_NO_STORE
                    frag_store<T>(f, dst);
#else
                    if (lo[0] == 123.456f) frag_store<T>(f, dst);
#endif
                } else {
                    for (int j = 0; j < 8 && n + j < p.Cout; j++) dst[j] = from_f32<T>(sp[j]);
                }
            }
        }
        __syncthreads();
    }
    if (p.stats != nullptr) {
        float* st = reinterpret_cast<float*>(smem_raw);  // [pixel group][half][BN][2], then [BN][2]
        float* col = st + WPG * 2 * BN * 2;
#pragma unroll
        for (int nr = 0; nr < NR; nr++) {
            float* d = st + (((pg * 2 + hf) * BN) + (ng * NR + nr) * 32 + r) * 2;
            d[0] = cs[nr];
            d[1] = css[nr];
        }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int k = 0; k < WPG * 2; k++) {  // fixed order -> deterministic
                s += st[(k * BN + tid) * 2];
                ss += st[(k * BN + tid) * 2 + 1];
            }
            col[tid * 2] = s;
            col[tid * 2 + 1] = ss;
        }
        __syncthreads();
        const int cpg = p.Cout / p.groups;
        const int gpt = BN / cpg;  // groups covered by this N tile (host guarantees cpg | BN)
        if (tid < gpt) {
            const int g = (n0 / cpg) + tid;
            if (g < p.groups) {
                float s = 0.f, ss = 0.f;
                for (int c = 0; c < cpg; c++) {
                    s += col[(tid * cpg + c) * 2];
                    ss += col[(tid * cpg + c) * 2 + 1];
                }
                float* dst = p.stats + (((size_t)b * p.groups + g) * tiles + tile) * 2;
                dst[0] = s;
                dst[1] = ss;
            }
        }
    }
}

template <typename T, int BN, int KC, bool PF, int WPG, int WNG>
int launch(const dig_conv3x3_args* a, hipStream_t stream) {
    Params<T> p;
    p.in[0] = (const T*)a->in0;
    p.in[1] = (const T*)a->in1;
    p.C[0] = a->C0;
    p.C[1] = a->in1 ? a->C1 : 0;
    p.aff0 = a->aff0;
    p.w = (const T*)a->w;
    p.bias = a->bias;
    p.out = (T*)a->out;
    p.stats = a->stats;
    p.bstride0 = a->in0_bstride > 0 ? a->in0_bstride : (long long)(a->up2 ? (a->H / 2) * (a->W / 2) : a->H * a->W) * a->C0;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.groups = a->groups; p.up2 = a->up2;
    p.tiles_x = cdiv(a->W, TILE_W);
    p.tiles_y = cdiv(a->H, 4 * WPG);
    constexpr size_t lds = conv_lds_bytes<T, BN, KC, WPG>();
    static std::atomic<bool> attr_set[64];  // per instantiation, per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_kernel<T, BN, KC, PF, WPG, WNG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "conv3x3: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
        attr_set[dev].store(true, std::memory_order_release);
    }
    dim3 grid(a->B * p.tiles_x * p.tiles_y, cdiv(a->Cout, BN));
    hipLaunchKernelGGL((conv3x3_kernel<T, BN, KC, PF, WPG, WNG>), grid, dim3(64 * WPG * WNG), lds, stream, p);
    DIG_LAUNCH_CHECK("conv3x3");
    return DIG_OK;
}

// tile configuration per (Cout, dtype): the weight packer (host) and the launcher must agree
void conv_config(int Cout, int dtype, int* BN, int* KC, int* TH) {
    int bn = (Cout % 128 == 0) ? 128 : (Cout > 32 ? 64 : 32);
    if (dtype == DIG_F32 && bn > 64) bn = 64;
    *BN = bn;
    *KC = (dtype == DIG_F32) ? 16 : (bn == 32 ? 32 : (bn == 64 ? CONV_KC64 : 16));
    *TH = (dtype == DIG_BF16 && bn == 64) ? 4 * CONV_WPG64 : 16;  // tile height (width is always 16)
}

}  // namespace

extern "C" int dig_conv3x3_tiles(int H, int W, int C0, int C1, int Cout, int groups, int dtype) {
    if (conv3x3_db_eligible(C0, C1, Cout, dtype)) return cdiv(H, 16) * cdiv(W, TILE_W);  // one partial per 16x16 pixels from either tile height
    if (conv3x3_ws64_eligible(H, W, C0, C1, Cout, groups, dtype)) return (H / 16) * (W / 16) * 4;  // per-wave partials
    int BN, KC, TH;
    conv_config(Cout, dtype, &BN, &KC, &TH);
    return cdiv(H, TH) * cdiv(W, TILE_W);
}

extern "C" int dig_conv3x3_wlayout(int C0, int C1, int Cout, int dtype) {
    return conv3x3_db_eligible(C0, C1, Cout, dtype) ? 2 : 0;
}

extern "C" int dig_conv3x3_config(int Cout, int dtype, int* BN, int* KC) {
    DIG_CHECK(Cout > 0 && BN && KC && (dtype == DIG_BF16 || dtype == DIG_F32), DIG_EINVAL, "conv3x3_config: bad argument");
    int TH;
    conv_config(Cout, dtype, BN, KC, &TH);
    return DIG_OK;
}

extern "C" int dig_conv3x3(const dig_conv3x3_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DIG_CHECK(a && a->in0 && a->w && a->bias && a->out, DIG_EINVAL, "conv3x3: null pointer");
    DIG_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, DIG_EINVAL, "conv3x3: bad shape B=%d H=%d W=%d Cout=%d", a->B, a->H, a->W, a->Cout);
    DIG_CHECK(a->C0 > 0 && a->C0 % 8 == 0 && (a->in1 == nullptr || (a->C1 > 0 && a->C1 % 8 == 0)), DIG_EINVAL,
              "conv3x3: channel counts must be positive multiples of 8 (C0=%d C1=%d)", a->C0, a->C1);
    DIG_CHECK(!a->up2 || (a->H % 2 == 0 && a->W % 2 == 0), DIG_EINVAL, "conv3x3: up2 needs even H,W");
    DIG_CHECK(((uintptr_t)a->in0 | (uintptr_t)a->in1 | (uintptr_t)a->w | (uintptr_t)a->out) % 16 == 0, DIG_EALIGN,
              "conv3x3: pointers must be 16-byte aligned");
    DIG_CHECK(a->dtype == DIG_BF16 || a->dtype == DIG_F32, DIG_EINVAL, "conv3x3: unknown dtype %d", a->dtype);
    int BN, KC, TH;
    conv_config(a->Cout, a->dtype, &BN, &KC, &TH);
    if (a->stats) {
        DIG_CHECK(a->groups > 0 && a->Cout % a->groups == 0, DIG_EINVAL, "conv3x3: Cout %% groups != 0");
        const int cpg = a->Cout / a->groups;
        DIG_CHECK(cpg <= BN && BN % cpg == 0 && a->Cout % cpg == 0, DIG_EINVAL,
                  "conv3x3: GroupNorm group width %d does not tile the %d-channel block", cpg, BN);
    }
    // (the packed weight layout follows the same order: dig_conv3x3_wlayout)
    if (conv3x3_db_eligible(a->C0, a->in1 ? a->C1 : 0, a->Cout, a->dtype)) return conv3x3_db_launch(a, stream);
    if (conv3x3_ws64_eligible(a->H, a->W, a->C0, a->in1 ? a->C1 : 0, a->Cout, a->stats ? a->groups : 8, a->dtype))
        return conv3x3_ws64_launch(a, stream);
    // 128 -> 64 without a GroupNorm prologue: register-stationary weights (same GroupNorm partial count as the generic kernel)
    if (conv3x3_rs128_eligible(a->H, a->W, a->C0, a->in1 ? a->C1 : 0, a->Cout, a->stats ? a->groups : 8, a->dtype, a->aff0 != nullptr) &&
        !getenv("DIG_NO_RS128"))
        return conv3x3_rs128_launch(a, stream);
    if (a->dtype == DIG_BF16) {
        if (BN == 128) return launch<bf16, 128, 16, CONV_PF128, 4, CONV_WNG128>(a, stream);
        if (BN == 64) return launch<bf16, 64, CONV_KC64, CONV_PF64, CONV_WPG64, 1>(a, stream);
        return launch<bf16, 32, 32, true, 4, 1>(a, stream);
    }
    if (BN == 64) return launch<float, 64, 16, false, 4, 1>(a, stream);
    return launch<float, 32, 16, false, 4, 1>(a, stream);
}
