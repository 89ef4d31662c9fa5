// conv3x3 (pad 1) for the wide layers (Cout % 128 == 0, bf16): double-buffered implicit GEMM on v_mfma_f32_16x16x32_bf16, gfx950.
//
// Tiling as conv3x3.hip (one halo tile of KC = 16 input channels serves all 9 taps), rebuilt around the data movement and
// the clock that bound those layers:
//   * 32x16-pixel x 128-channel tile per block, 512 threads = 8 waves (8 pixel groups of 4 rows: 64 px x 128 ch each,
//     32 accumulator tiles of 16x16), ONE block per CU; a 16-row / 4-wave variant for launches that would leave CUs idle;
//   * the matrix instruction is the 16x16x32 form: on this part a loop of them holds a 12-14 % higher clock than the same
//     flops issued as 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7; tools/lab/mfma_peak: 2.15 vs 1.88 PFLOP/s),
//     with the same LDS fragment traffic for the same 64 x 128 register tile (12 ds_read_b128 per 32 MFMAs);
//   * K = 32 of one instruction = TWO TAPS x 16 channels: taps (0,1), (2,3), (4,5), (6,7) of a chunk are four K steps, and
//     tap 8 is paired ACROSS chunks - tap 8 of chunks 2P and 2P+1 form one K step, issued in the odd period.  For that the
//     halo is a ring of THREE chunk buffers (chunk 2P is still resident in period 2P+1) and the tap-8 weights of a chunk
//     pair are a slab of their own;
//   * weights are the A operand (accumulator rows = channels): a lane ends up with 8 consecutive output channels of one
//     pixel (two row-permuted 16-channel tiles), so the epilogue converts to bf16 in registers and passes 16-byte vectors
//     through the wave's own LDS rows - no fp32 staging, no epilogue barriers except the one of the GroupNorm statistics;
//   * everything goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, no registers, no LDS store instructions): the weight
//     slabs and the halo chunk (concat / nearest-x2 are source addressing).  A GroupNorm-apply + SiLU prologue (PRO) is applied
//     IN PLACE in LDS: each lane transforms exactly the 16-byte slots its own DMA instructions wrote (so its own vmcnt tells
//     it they have landed), inside the MFMA sequence of the period before the chunk is used; one barrier per chunk.
// Weight slabs are packed on the host exactly as they sit in LDS (dig_conv3x3_wlayout == 2, see ops.ConvWeight):
//   per 128-channel N tile, per chunk pair P:  main(2P) | main(2P+1) | w8(P)
//   main(c): [4 K steps][128 rows][4 slots x 16 B]  slot s of K step p = tap 2p + (s >> 1), channel half s & 1 of chunk c
//   w8(P)  : [128 rows][4 slots x 16 B]             slot s = tap 8 of chunk 2P + (s >> 1), channel half s & 1
//   row rho = mt*16 + rr holds output channel 32*(mt >> 1) + 8*(rr >> 2) + 4*(mt & 1) + (rr & 3) of the N tile, and its
//   four 16-byte slots are stored at position s ^ f[(rho >> 2) & 3], f = {0, 2, 3, 1}: with 64-byte rows that makes the
//   16-lane groups of ds_read_b128 hit 16 different 16-byte bank groups (MI355X_MICROARCH.md, LDS).
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 16, HALO_W = 18;

// NW = waves per block = 4-row pixel groups per tile: 8 (32x16-pixel tile, the default) or 4 (16x16-pixel tile, for launches
// whose 32-row tiles would leave CUs idle: e.g. 256->256 at 32x32 with B = 32 is 128 blocks of 8 waves or 256 of 4).
// The halo LDS image is the plain [halo pixel][16 channels] array (32 B per pixel, no padding); a DMA instruction writes 64
// consecutive 16-byte slots, so the bank swizzle is applied on the SOURCE side: slot (pixel, h) holds channel half
// h ^ (halo row & 1).  A fragment is 16 pixels of ONE row x 2 halves x 2 taps; the 16 lanes that ds_read_b128 serves together
// read 8 consecutive pixels' one half and the other 8 pixels' other half: all 64 banks.
// BN = output channels per block: 128, or 64 for the 64-channel layers (Cout == 64: half the weight slabs, 16 accumulator tiles
// per wave, two 4-wave blocks per CU at 16-row tiles).
template <int NW, int BN = 128> struct DbCfg {
    static constexpr int WMAIN_BYTES = 4 * BN * 64;                // BN=128: 32768: taps 0..7 of one chunk = 4 K steps
    static constexpr int W8_BYTES = BN * 64;                       // 8192: tap 8 of a chunk pair = 1 K step
    static constexpr int PAIR_BYTES = 2 * WMAIN_BYTES + W8_BYTES;  // 73728 = 2 chunks x 9 taps x 128 x 16 x 2 B
    static constexpr int TILE_H = 4 * NW, HALO_H = TILE_H + 2, NTHREADS = 64 * NW;
    static constexpr int HSLOTS = HALO_H * HALO_W * 2, HPIECES = (HSLOTS + 63) / 64;  // 16-byte slots, 1 KB DMA pieces
    static constexpr int HALO_BYTES = HPIECES * 1024;                                 // NW=8: 20480, NW=4: 11264
    static constexpr int W_OFF = 3 * HALO_BYTES;                                      // ring of three halo chunks, then the weights
    static constexpr int AFF_OFF = W_OFF + 2 * WMAIN_BYTES + 2 * W8_BYTES;            // NW=8: 143360, NW=4: 115712
    static constexpr int AFF_MAX_C = BN == 128 ? 2048 : 768;                          // PRO: [C0][2] fp32 GroupNorm affine of this image (BN = 64, 4 waves: two blocks per CU still fit)
    static constexpr int EPI_BYTES = NW * 64 * (2 * BN + 16) + NW * (BN / 8) * 8;     // epilogue: staged bf16 rows + GroupNorm sums
    static constexpr int DB_LDS = AFF_OFF > EPI_BYTES ? AFF_OFF : EPI_BYTES;
    static constexpr int DB_LDS_PRO = DB_LDS + AFF_MAX_C * 8;                         // NW=8, BN=128: 159744 of 163840
};

struct DbParams {
    const bf16* in[2];
    int C[2];
    const float* aff0;
    const bf16* w;
    const float* bias;
    bf16* out;
    float* stats;
    long long bstride0;
    int B, H, W, Cout, groups, up2, tiles_x, tiles_y;
};

// (PRO instantiations run without the fences: with the prologue's registers on top they spill, measured 8 % slower at 16-row tiles)
#define DB_FENCE                                                  \
    do {                                                          \
        if constexpr (!PRO) __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
__device__ __forceinline__ f32x4 mma32(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// WP = the N-tile width the weights were PACKED for: BN, or 128 with BN = 64 - then the block takes the lower / upper 64 rows of a
// 128-row slab (whole 4 KB runs per K step; the row -> channel map and the slot swizzle of rows 64..127 repeat those of 0..63).
template <int NW, bool PRO, int BN, int WP = BN> __global__ __launch_bounds__(64 * NW, 2) void conv3x3_db_kernel(DbParams p) {
    using Cfg = DbCfg<NW, BN>;
    constexpr int WMAIN_BYTES = Cfg::WMAIN_BYTES, W8_BYTES = Cfg::W8_BYTES, PAIR_BYTES = Cfg::PAIR_BYTES;
    constexpr int MT = BN / 16, NG = MT / 2, NJ = BN / 32;  // 16-channel accumulator tiles, fragment groups per K step, 32-channel tile pairs
    constexpr int TILE_H = Cfg::TILE_H, HALO_H = Cfg::HALO_H, NTHREADS = Cfg::NTHREADS;
    constexpr int HALO_BYTES = Cfg::HALO_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave;  // pixel group: tile rows 4pg..4pg+3; every wave computes all 128 channels
    const int col = lane & 15, g = lane >> 4;  // fragment row (weights) / pixel column; K group: 8 of the 32 K values
    const int tiles = p.tiles_x * p.tiles_y;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / p.tiles_x) * TILE_H, tx0 = (tile % p.tiles_x) * 16;
    const int n0 = blockIdx.y * BN;
    const int Hs = p.up2 ? p.H >> 1 : p.H, Ws = p.up2 ? p.W >> 1 : p.W;
    const int nch0 = p.C[0] / KC, nch1 = p.C[1] / KC, nch = nch0 + nch1;
    static_assert(WP == BN || (WP == 128 && BN == 64), "weights packed for BN, or for 128 rows and split in halves");
    constexpr int SRC_MAIN = 4 * WP * 64, SRC_PAIR = 2 * SRC_MAIN + WP * 64;  // the packed slabs' sizes
    constexpr int NSPLIT = WP / BN;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + (size_t)(blockIdx.y / NSPLIT) * ((nch + 1) >> 1) * SRC_PAIR +
                                 (size_t)(blockIdx.y % NSPLIT) * (BN * 64);

    f32x4 acc[MT][4];  // [channel tile mt][pixel row mr]
#pragma unroll
    for (int i = 0; i < MT * 4; i++) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addresses (bytes).  Weights: row `col` of a 16-row tile, slot g swizzled (see the header).
    const int woff = col * 64 + ((g ^ ((0x1320 >> (4 * (col >> 2))) & 3)) << 4);
    // Pixels: K step q covers taps 2q and 2q+1; this lane's tap is 2q + (g >> 1), its channel half g & 1.  Halo row of pixel
    // row mr = 4pg + mr + ky (4pg is even): even mr read from hbe[q] (+ 2 rows for mr = 2), odd mr from hbo[q].
    // The odd-row address is the even-row one a halo row further with the slot bit flipped: (hbe + 576) ^ 16 (576 has no bits
    // below bit 6, so the add cannot carry into bit 4) - formed at the use, which keeps five registers free.
    int hbe[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int tap = 2 * q + (g >> 1), ky = tap / 3, kx = tap - 3 * ky;
        hbe[q] = ((4 * pg + ky) * HALO_W + col + kx) * 32 + (((g & 1) ^ (ky & 1)) << 4);
    }
    const int hbe8 = ((4 * pg + 2) * HALO_W + col + 2) * 32 + ((g & 1) << 4);  // tap 8 (ky = kx = 2)

    // ---- halo piece q (slots 64q..64q+63) by wave q % NW; per piece the lane's source pixel, channel half and validity
    constexpr int HPW = (Cfg::HPIECES + NW - 1) / NW;  // pieces per wave (3 for both tile heights)
    int dpix[HPW];
    bool dval[HPW];
    int dhalf[HPW];
    {
#pragma unroll
        for (int i = 0; i < HPW; i++) {
            const int slot = (wave + i * NW) * 64 + lane;
            const int pix = min(slot >> 1, HALO_H * HALO_W - 1), h = slot & 1;
            const int hy = pix / HALO_W, hx = pix - hy * HALO_W;
            const int Y = ty0 + hy - 1, X = tx0 + hx - 1;
            dval[i] = (wave + i * NW) < Cfg::HPIECES && slot < Cfg::HSLOTS && Y >= 0 && Y < p.H && X >= 0 && X < p.W;
            const int Yc = min(max(Y, 0), p.H - 1), Xc = min(max(X, 0), p.W - 1);
            const int sy = p.up2 ? Yc >> 1 : Yc, sx = p.up2 ? Xc >> 1 : Xc;
            dpix[i] = sy * Ws + sx;
            dhalf[i] = h ^ (hy & 1);
            // slots that no DMA ever writes (outside the image, or the tail of the last piece) are zeroed once, in all three buffers
            if (!dval[i] && (wave + i * NW) < Cfg::HPIECES) {
#pragma unroll
                for (int k = 0; k < 3; k++) *reinterpret_cast<i32x4*>(smem + k * HALO_BYTES + (wave + i * NW) * 1024 + lane * 16) = i32x4{0, 0, 0, 0};
            }
        }
    }
    auto hdma = [&](int ci, unsigned char* buf) {
        const int piece = ci >= nch0 ? 1 : 0;
        const int cb = (ci - (piece ? nch0 : 0)) * KC;
        const int Cp = p.C[piece];
        const bf16* src = p.in[piece] + (size_t)b * (piece ? (size_t)Hs * Ws * Cp : (size_t)p.bstride0) + cb;
#pragma unroll
        for (int i = 0; i < HPW; i++) {
            if ((wave + i * NW) < Cfg::HPIECES) {  // scalar
                if (dval[i])                       // lanes of out-of-image pixels stay out of the DMA (EXEC)
                    __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)dpix[i] * Cp + dhalf[i] * 8), (lds_void*)(buf + (wave + i * NW) * 1024), 16, 0, 0);
            }
        }
    };
    // ---- PRO: GroupNorm-apply + SiLU (ddp.py:160-166) of chunk ci, in place on the slots this lane's DMA wrote (out-of-image
    // slots stay zero: the convolution pads the ACTIVATED tensor).  afl: this image's [C0][2] table in LDS, (a, b) pre-scaled
    // by -log2(e) (silu_pre, common.hpp).
    float* const afl = reinterpret_cast<float*>(smem + Cfg::AFF_OFF);
    auto hpro = [&](int ci, unsigned char* buf, int i) {
        if ((wave + i * NW) < Cfg::HPIECES && dval[i]) {
            bf16x8* sp = reinterpret_cast<bf16x8*>(buf + (wave + i * NW) * 1024 + lane * 16);
            bf16x8 v = *sp;
            const float* af = afl + (ci * KC + dhalf[i] * 8) * 2;
#pragma unroll
            for (int j4 = 0; j4 < 4; j4++) {
                const f32x4 ab = *reinterpret_cast<const f32x4*>(af + j4 * 4);
                v[2 * j4] = (bf16)silu_pre(ab[0], ab[1], (float)v[2 * j4]);
                v[2 * j4 + 1] = (bf16)silu_pre(ab[2], ab[3], (float)v[2 * j4 + 1]);
            }
            *sp = v;
        }
    };
    // ---- weights of chunk ci: main slab (32 pieces of 1 KB) and, with every even chunk, the tap-8 slab of its pair (8 pieces);
    // global -> LDS directly, piece q by wave q % NW
    unsigned char* const wl = smem + Cfg::W_OFF;
    auto wload = [&](int ci) {
        const unsigned char* pair = wbase + (size_t)(ci >> 1) * SRC_PAIR;
        const unsigned char* slab = pair + (ci & 1) * SRC_MAIN;
        unsigned char* dst = wl + (ci & 1) * WMAIN_BYTES;
        constexpr int PPS = BN * 64 / 1024;  // 1 KB pieces per K step of this block's rows
#ifdef DB_DMA_UNROLL1
#pragma unroll 1
#endif
        for (int q = wave; q < WMAIN_BYTES / 1024; q += NW)
            __builtin_amdgcn_global_load_lds((glb_void*)(slab + (q / PPS) * (WP * 64) + (q % PPS) * 1024 + lane * 16), (lds_void*)(dst + q * 1024), 16, 0, 0);
        if (!(ci & 1)) {
            const unsigned char* s8 = pair + 2 * SRC_MAIN;
            unsigned char* d8 = wl + 2 * WMAIN_BYTES + ((ci >> 1) & 1) * W8_BYTES;
#ifdef DB_DMA_UNROLL1
#pragma unroll 1
#endif
            for (int q = wave; q < W8_BYTES / 1024; q += NW)
                __builtin_amdgcn_global_load_lds((glb_void*)(s8 + q * 1024 + lane * 16), (lds_void*)(d8 + q * 1024), 16, 0, 0);
        }
    };

    wload(0);
    hdma(0, smem);
    if constexpr (PRO) {
        const float* asrc = p.aff0 + (size_t)b * p.C[0] * 2;
        for (int i = tid; i < p.C[0] / 2; i += NTHREADS) *reinterpret_cast<f32x4*>(afl + i * 4) = *reinterpret_cast<const f32x4*>(asrc + i * 4) * SILU_PRE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
        __syncthreads();                                   // the affine table is complete
#pragma unroll
        for (int i = 0; i < HPW; i++) hpro(0, smem, i);
    }
    __syncthreads();  // (drains the LDS-DMA as well: vmcnt(0))
    int hcur = 0;     // ring slot of the current chunk's halo
    // Fragment pipeline, explicit: a K step is 4 groups of (2 weight fragments x 4 pixel fragments = 8 MFMAs); the fragments of
    // group i+1 (and, at the end of a K step, the next step's pixel fragments) are read BEFORE the MFMAs of group i are issued.
    // sched_barrier pins that order - left alone, hipcc sinks every LDS read down to its first use and the wave sits out a
    // full LDS round trip every 8 MFMAs.
    bf16x8 fa[2][2], fb[2][4];
    auto LB = [&](bf16x8(&d)[4], const unsigned char* hb, int be) {
        const int bo = (be + HALO_W * 32) ^ 16;
#pragma unroll
        for (int mr = 0; mr < 4; mr++) d[mr] = *reinterpret_cast<const bf16x8*>(hb + ((mr & 1) ? bo : be) + (mr >> 1) * (2 * HALO_W * 32));
    };
    auto LA = [&](bf16x8(&d)[2], const unsigned char* wp, int gq) {
        d[0] = *reinterpret_cast<const bf16x8*>(wp + (2 * gq) * 1024);
        d[1] = *reinterpret_cast<const bf16x8*>(wp + (2 * gq + 1) * 1024);
    };
    auto MM = [&](const bf16x8(&a2)[2], const bf16x8(&b4)[4], int gq) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mr = 0; mr < 4; mr++) acc[2 * gq + t][mr] = mma32(a2[t], b4[mr], acc[2 * gq + t][mr]);
    };
    for (int ci = 0; ci < nch; ci++) {
        const int hnxt = hcur == 2 ? 0 : hcur + 1, hprv = hcur == 0 ? 2 : hcur - 1;
        const unsigned char* halo = smem + hcur * HALO_BYTES;
        unsigned char* hnext = smem + hnxt * HALO_BYTES;
        const unsigned char* wm = wl + (ci & 1) * WMAIN_BYTES + woff;
        const bool more = ci + 1 < nch;
        // tap 8 of the chunk pair (2P, 2P+1) is ONE K step, in the odd period: K groups 0,1 read chunk 2P's halo (still in the
        // ring), K groups 2,3 this chunk's.  A lone last chunk pairs with zeros (its slab's upper slots are zero too).
        const bool has8 = (ci & 1) || !more, lone = !(ci & 1);
        const unsigned char* w8 = wl + 2 * WMAIN_BYTES + ((ci >> 1) & 1) * W8_BYTES + woff;
        const unsigned char* h8 = smem + (lone ? hcur : ((g >> 1) ? hcur : hprv)) * HALO_BYTES;
        LB(fb[0], halo, hbe[0]);
        LA(fa[0], wm, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int gq = 0; gq < NG; gq++) {
                if (gq < NG - 1) {
                    LA(fa[(gq + 1) & 1], wm + q * (BN * 64), gq + 1);
                } else if (q < 3) {
                    LA(fa[0], wm + (q + 1) * (BN * 64), 0);
                    LB(fb[(q + 1) & 1], halo, hbe[q + 1]);
                } else if (has8) {
                    LA(fa[0], w8, 0);
                    LB(fb[0], h8, hbe8);
                }
                DB_FENCE;
                MM(fa[gq & 1], fb[q & 1], gq);
                DB_FENCE;
                if (q == 0 && gq == 0 && more) {  // next chunk's loads go out once the MFMA pipe has work queued behind the barrier
                    wload(ci + 1);
                    hdma(ci + 1, hnext);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the prologue of the NEXT chunk, one DMA piece per MFMA group, late in the period (the pieces were issued at its
                // start); the two wave halves do it in different K steps so that they do not all leave the matrix pipe at once
                if constexpr (PRO) {
                    if (q == (wave < NW / 2 ? 3 : 2) && more && ci + 1 < nch0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (gq == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int i = gq; i < HPW; i += NG) hpro(ci + 1, hnext, i);  // the pieces dealt out over the step's groups
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (has8) {
            if (lone && (g >> 1)) {
#pragma unroll
                for (int mr = 0; mr < 4; mr++) fb[0][mr] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int gq = 0; gq < NG; gq++) {
                if (gq < NG - 1) LA(fa[(gq + 1) & 1], w8, gq + 1);
                DB_FENCE;
                MM(fa[gq & 1], fb[0], gq);
                DB_FENCE;
            }
        }
        __syncthreads();
        hcur = hnxt;
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA retired before the accumulators are read (see conv3x3.hip)
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue from the accumulators: lane (pixel column col, row group g) holds, for every tile pair j, the 8 consecutive
    // channels n0 + 32j + 8g .. +7 of its pixel in each of the wave's 4 rows: + bias, GroupNorm sums of the fp32 values, bf16.
    // Stored straight from these registers a wave instruction would write 64 scattered 64-byte pieces (measured: 18 us per
    // launch, three times the LDS-staged epilogue of the 32x32x16 kernel), so the 16-byte vectors take a turn through the wave's
    // OWN 64 x 272-byte LDS rows (no block barrier: every wave is past the loop's last one) and leave as whole 256-byte pixel rows.
    constexpr int EROW = 2 * BN + 16;  // bytes per staged pixel (+16: the 8 lanes that ds_write_b128 serves together hit 8 bank groups)
    unsigned char* const est = smem + wave * (64 * EROW);
    float cs[NJ], css[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        cs[j] = css[j] = 0.f;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 32 * j + 8 * g), b1 = *reinterpret_cast<const f32x4*>(p.bias + n0 + 32 * j + 8 * g + 4);
#pragma unroll
        for (int mr = 0; mr < 4; mr++) {
            const int Y = ty0 + 4 * pg + mr, X = tx0 + col;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] = acc[2 * j][mr][e] + b0[e];
                v[4 + e] = acc[2 * j + 1][mr][e] + b1[e];
            }
            *reinterpret_cast<bf16x8*>(est + (mr * 16 + col) * EROW + j * 64 + g * 16) =
                bf16x8{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
            if (Y < p.H && X < p.W) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    cs[j] += v[e];
                    css[j] += v[e] * v[e];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < BN / 8; i++) {  // (512 / BN) pixels x 2*BN bytes per wave instruction
        constexpr int VPP = BN / 8, PPI = 64 / VPP;  // 16-byte vectors per pixel, pixels per instruction
        const int pix = i * PPI + lane / VPP, v16 = lane % VPP;
        const int Y = ty0 + 4 * pg + (pix >> 4), X = tx0 + (pix & 15);
        if (Y < p.H && X < p.W)
            *reinterpret_cast<bf16x8*>(p.out + (((size_t)b * p.H + Y) * p.W + X) * p.Cout + n0 + v16 * 8) = *reinterpret_cast<const bf16x8*>(est + pix * EROW + v16 * 16);
    }
    if (p.stats != nullptr) {
        // per wave and 8-channel vector (j, g): sums over the wave's 4 x 16 pixels (xor tree over the 16 pixel columns), then
        // - fixed order, so deterministic - over the 4 waves of a 16-row sub-tile and the vectors of a GroupNorm group.  One
        // partial per 16x16 pixels from either tile height, summed the same way: the launch heuristic cannot change a bit.
        constexpr int NV = BN / 8;  // 8-channel vectors of the N tile
        float* st = reinterpret_cast<float*>(smem + NW * (64 * EROW));  // [wave][NV vectors][2], behind the staging rows
#pragma unroll
        for (int j = 0; j < NJ; j++) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                cs[j] += __shfl_xor(cs[j], o, 64);
                css[j] += __shfl_xor(css[j], o, 64);
            }
            if (col == 0) {
                st[((pg * NV) + j * 4 + g) * 2] = cs[j];
                st[((pg * NV) + j * 4 + g) * 2 + 1] = css[j];
            }
        }
        __syncthreads();
        constexpr int NS = NW / 4;
        const int cpg = p.Cout / p.groups;
        const int gpt = BN / cpg;  // groups covered by this N tile (host guarantees cpg | BN and 8 | cpg)
        const int vpg = cpg / 8;   // 8-channel vectors per group
        if (tid < gpt * NS) {
            const int ns = tid / gpt, gi = tid - ns * gpt;
            const int grp = (n0 / cpg) + gi;
            const int ty16 = (tile / p.tiles_x) * NS + ns, rows16 = (p.H + 15) / 16;
            if (grp < p.groups && ty16 < rows16) {
                float s = 0.f, ss = 0.f;
                for (int k = 0; k < 4; k++)
                    for (int v = 0; v < vpg; v++) {
                        s += st[(((ns * 4 + k) * NV) + gi * vpg + v) * 2];
                        ss += st[(((ns * 4 + k) * NV) + gi * vpg + v) * 2 + 1];
                    }
                float* dst = p.stats + (((size_t)b * p.groups + grp) * (rows16 * p.tiles_x) + ty16 * p.tiles_x + (tile % p.tiles_x)) * 2;
                dst[0] = s;
                dst[1] = ss;
            }
        }
    }
}

}  // namespace

// eligibility: bf16, 16-channel input chunks, and either whole 128-channel output tiles or the 64-channel layers this kernel
// is the faster one for (db64_wanted)
// Measured at B = 32, 256x256 (tools/lab/ab_env.sh): the BN = 64 instantiation loses to both special kernels of the 64-channel
// layers - 128 -> 64: 340 us against 318 (conv3x3_rs128), 64 -> 64: 222 / 280 us (plain / prologue) against 160 / 206
// (conv3x3_ws64): with 4-8 chunks per tile the per-tile weight re-fetch and fill are not amortised.  It therefore stays OFF by
// default and exists for Cout = 64 shapes the special kernels do not take.  DIG_DB64 (read per call: the weight packer asks
// dig_conv3x3_wlayout with the same setting) = "all": every Cout == 64 layer, "k128": the 128 -> 64 layers, default none.
static int db64_mode() {
    const char* e = getenv("DIG_DB64");
    if (!e) return 0;
    return e[0] == 'a' ? 2 : (e[0] == 'k' ? 1 : 0);
}
static bool db64_wanted(int C0, int C1) {
    const int m = db64_mode();
    return m == 2 || (m == 1 && C0 + C1 == 128);
}
bool conv3x3_db_eligible(int C0, int C1, int Cout, int dtype) {
    if (dtype != DIG_BF16 || C0 % 16 != 0 || C1 % 16 != 0) return false;
    return Cout % 128 == 0 || (Cout == 64 && db64_wanted(C0, C1));
}

// BN = channels per block, WP = the N-tile width the weights are packed for (conv_config: 128 when Cout % 128 == 0, else 64)
template <int BN, int WP> static int db_launch(const dig_conv3x3_args* a, hipStream_t stream, bool small, bool pro) {
    DbParams p;
    p.in[0] = (const bf16*)a->in0;
    p.in[1] = (const bf16*)a->in1;
    p.C[0] = a->C0;
    p.C[1] = a->in1 ? a->C1 : 0;
    p.aff0 = a->aff0;
    p.w = (const bf16*)a->w;
    p.bias = a->bias;
    p.out = (bf16*)a->out;
    p.stats = a->stats;
    p.bstride0 = a->in0_bstride > 0 ? a->in0_bstride : (long long)(a->up2 ? (a->H / 2) * (a->W / 2) : a->H * a->W) * a->C0;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.groups = a->groups; p.up2 = a->up2;
    p.tiles_x = cdiv(a->W, 16);
    if (a->stats) {
        const int cpg = a->Cout / a->groups;
        DIG_CHECK(cpg % 8 == 0 && BN % cpg == 0, DIG_EINVAL, "conv3x3(db): GroupNorm group width %d must be a multiple of 8 that divides %d", cpg, BN);
    }
    static std::atomic<bool> attr_set[64];  // per instantiation (BN, WP) and device; racing first calls all do the same
    int device = 0;
    (void)hipGetDevice(&device);
    const int dev = device & 63;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_db_kernel<8, false, BN, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, DbCfg<8, BN>::DB_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_db_kernel<4, false, BN, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, DbCfg<4, BN>::DB_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_db_kernel<8, true, BN, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, DbCfg<8, BN>::DB_LDS_PRO);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_db_kernel<4, true, BN, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, DbCfg<4, BN>::DB_LDS_PRO);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "conv3x3(db): cannot raise dynamic LDS to %d: %s", DbCfg<8, BN>::DB_LDS_PRO, hipGetErrorString(e));
        attr_set[dev].store(true, std::memory_order_release);
    }
    DIG_CHECK(!pro || (a->C0 <= DbCfg<8, BN>::AFF_MAX_C && ((uintptr_t)a->aff0) % 16 == 0), DIG_EINVAL,
              "conv3x3(db): a prologue needs C0 <= %d and a 16-byte aligned aff0 (C0 = %d)", DbCfg<8, BN>::AFF_MAX_C, a->C0);
    p.tiles_y = cdiv(a->H, small ? 16 : 32);
    const dim3 grid(a->B * p.tiles_x * p.tiles_y, a->Cout / BN);
    constexpr int lds4 = DbCfg<4, BN>::DB_LDS, lds4p = DbCfg<4, BN>::DB_LDS_PRO, lds8 = DbCfg<8, BN>::DB_LDS, lds8p = DbCfg<8, BN>::DB_LDS_PRO;
    if (small && pro) hipLaunchKernelGGL((conv3x3_db_kernel<4, true, BN, WP>), grid, dim3(256), lds4p, stream, p);
    else if (small) hipLaunchKernelGGL((conv3x3_db_kernel<4, false, BN, WP>), grid, dim3(256), lds4, stream, p);
    else if (pro) hipLaunchKernelGGL((conv3x3_db_kernel<8, true, BN, WP>), grid, dim3(512), lds8p, stream, p);
    else hipLaunchKernelGGL((conv3x3_db_kernel<8, false, BN, WP>), grid, dim3(512), lds8, stream, p);
    DIG_LAUNCH_CHECK("conv3x3(db)");
    return DIG_OK;
}

int conv3x3_db_launch(const dig_conv3x3_args* a, hipStream_t stream) {
    DIG_CHECK(((uintptr_t)a->bias) % 16 == 0, DIG_EALIGN, "conv3x3(db): bias must be 16-byte aligned");
    static std::atomic<int> ncu[64];  // zero-initialised; racing first calls all store the same value
    int device = 0;
    (void)hipGetDevice(&device);
    int ncus = ncu[device & 63].load(std::memory_order_acquire);
    if (!ncus) {
        hipError_t e = hipDeviceGetAttribute(&ncus, hipDeviceAttributeMultiprocessorCount, device);
        DIG_CHECK(e == hipSuccess && ncus > 0, DIG_ELAUNCH, "conv3x3(db): cannot query the CU count");
        ncu[device & 63].store(ncus, std::memory_order_release);
    }
    const bool pro = a->aff0 != nullptr;  // GroupNorm-apply + SiLU prologue on the channels of in0
    const int tx = cdiv(a->W, 16);
    if (a->Cout % 128 != 0) {  // weights packed per 64 channels: 16-row tiles, two 4-wave blocks per CU (a tile has few chunks,
                               // and the second block's main loop covers the first one's fill and epilogue)
        const char* force = getenv("DIG_DB_TILE");
        return db_launch<64, 64>(a, stream, force ? atoi(force) == 16 : true, pro);
    }
    // Block shape (rows x output channels) by a cost model, in units of one 32-row x 128-channel block on a CU of its own:
    // rounds of blocks x cost per block, the costs measured at B = 32 / 20, 256x256 (tools/lab/ab_shapes.sh; per unit of work a
    // 16 x 128 block costs 1.25 (1.36 with the prologue, whose SiLU every block repeats on its own halo), a 16 x 64 block 1.12
    // (1.32): two of those share a CU (75 KB of LDS each), so a full round of 2 x #CUs blocks costs 0.56 (0.66) and a last
    // round that leaves at most one block per CU 0.38 (0.45)).  The small shapes win where they remove a mostly empty last
    // round: 20 images at 64x64 with 256 output channels are 320 blocks of 32 x 128 (2 rounds) or 1280 of 16 x 64
    // (2 x 0.56 + 0.38: measured 165 against 206 us).  EVERY shape gives the same bits - outputs and GroupNorm partials (one
    // per 16x16 pixels and group, summed in the same order) - so the choice may depend on the batch without breaking the
    // shard-equals-batch invariance (tests/test_gpu_kernels.py::test_conv3x3_db_tile_heights_bitwise).
    const long long g32 = (long long)a->B * tx * cdiv(a->H, 32) * (a->Cout / 128), g16 = (long long)a->B * tx * cdiv(a->H, 16) * (a->Cout / 128);
    const double c32 = (double)((g32 + ncus - 1) / ncus);
    const double c16 = (pro ? 0.68 : 0.625) * (double)((g16 + ncus - 1) / ncus);
    const long long gq = 2 * g16, full = gq / (2 * ncus), rem = gq - full * 2 * ncus;
    const double cq = (pro ? 0.66 : 0.56) * (double)full + (rem == 0 ? 0.0 : (rem <= ncus ? (pro ? 0.45 : 0.38) : (pro ? 0.66 : 0.56)));
    int shape = c16 < 0.97 * c32 ? 1 : 0;                // 0: 32 x 128, 1: 16 x 128, 2: 16 x 64, 3: 32 x 64 (test hook only)
    if (cq < 0.97 * (shape ? c16 : c32) && (!a->stats || (64 % (a->Cout / a->groups)) == 0) && (!pro || a->C0 <= DbCfg<4, 64>::AFF_MAX_C)) shape = 2;
    if (const char* force = getenv("DIG_DB_TILE")) {     // test hook, read per call: "16" / "32" [+ "x64"]
        shape = (atoi(force) == 16 ? 1 : 0);
        if (strstr(force, "x64")) shape = shape ? 2 : 3;
    }
    switch (shape) {
        case 0: return db_launch<128, 128>(a, stream, false, pro);
        case 1: return db_launch<128, 128>(a, stream, true, pro);
        case 2: return db_launch<64, 128>(a, stream, true, pro);
        default: return db_launch<64, 128>(a, stream, false, pro);
    }
}
