// conv3x3 (pad 1), 128 -> 64 channels, bf16, REGISTER-stationary weights, persistent, gfx950.
//
// The 128 -> 64 convolutions of the U-Net's 256x256 level (ResnetBlock.block1 on the (x | skip) concat, ddp.py:423-441, and
// the last Upsample conv, ddp.py:90-94) move 805 MB per launch at 384 FLOP/B: they sit on the HBM / per-CU streaming roof.
// The generic kernel (conv3x3.hip) re-stages the whole 147 KB weight tensor from L2 for every 256-pixel tile - twice the
// bytes of the activations themselves - and the weights do not fit LDS beside a halo tile either.  Here they live in
// REGISTERS for the lifetime of a persistent block:
//
//   * 256 threads = 4 waves, ONE wave per SIMD, each with the whole 512-entry register file (amdgpu_waves_per_eu(1,1));
//     wave (ph, nh) owns pixel half ph (8 tile rows = 128 pixels = four 32-pixel accumulator tiles) x channel half nh
//     (32 channels) of a 16x16-pixel tile and keeps its 32 x 1152 weight slice as 72 MFMA A fragments (288 registers:
//     the compiler spreads them over the VGPR and accumulator halves, MFMA reads A/B operands from either).
//   * LDS holds nothing but activations: a ring of three 18x18-pixel x 64-channel halo chunks (one per K half: the
//     x part and the skip part of the concat, or the two channel halves of a 128-channel source), filled by LDS-DMA
//     (global_load_lds_dwordx4: no registers, no VALU, no LDS store instructions) one chunk ahead of the MFMAs.  Rows are
//     128 B with the 16-byte slots XOR-swizzled by (hx >> 1) & 7 - applied to the per-lane SOURCE address, the LDS side of
//     a DMA being linear - so every ds_read_b128 fragment read is conflict-free.  Fused here: channel concat, nearest-x2
//     upsampling (source addressing), zero padding (border tiles: out-of-image pixels are zeroed in LDS after the DMA landed).
//   * one s_barrier per chunk; DMA completion is tracked with counted s_waitcnt vmcnt(N) (the DMAs and the output stores are
//     the only vector-memory operations of the steady state, their numbers per chunk are fixed).
//   * per tile: accumulators (initialised with the bias) -> bf16 -> the wave's own 8 KB LDS slice (the only part of the
//     epilogue the MFMA pipe waits for); the slice -> 16-byte stores of 64-byte half rows (both halves of a 128-byte line are
//     written by the two nh waves within the same tile, they merge in L2) and the GroupNorm partial sums of the STORED bf16
//     values run between the MFMAs of the next tile's first chunk.  One (sum, sumsq) pair per (tile, group): the two pixel
//     halves are combined through LDS one barrier later, so the partial count equals the generic kernel's.
// Measured (B = 32, 256x256, MI355X): 329 us against the generic kernel's 354 us.  What bounds it (per-chunk s_memtime trace,
// -DRS_TRACE): with ONE wave per SIMD every vector-memory instruction that finds the CU's memory pipeline full blocks the
// wave that also issues the MFMAs; a tile moves 115 KB (DMA pieces + stores) at the CU's share of HBM (~14 B/clk), i.e.
// ~8.0 k cycles of issue blocking on top of 9.2 k cycles of MFMAs, whatever the prefetch distance or placement (DIST 1 / 2,
// early / late / spread issue all land within 2 %).  Hiding it needs a second wave per SIMD, which 288 weight registers
// exclude: the design trades the generic kernel's weight re-staging (262 KB of L2 traffic per tile) for that serialisation.
// A/B fragment reads use the inline-asm ring of conv3x3_ws.hip (tools/check_async_lds_asm.py checks this kernel as well).
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// This kernel reads the generic kernel's weight packing [chunk][9 taps][64][16] (BN = 64, KC = 16) and writes one GroupNorm
// partial per 16x16 tile (TH = 4 * CONV_WPG64 = 16 in dig_conv3x3_tiles): a variant build that overrides the generic kernel's
// tile macros (tools/lab/build_variant.sh) would silently disagree with it.
#if (defined(CONV_KC64) && CONV_KC64 != 16) || (defined(CONV_WPG64) && CONV_WPG64 != 4)
#error "conv3x3_rs128 assumes CONV_KC64 == 16 and CONV_WPG64 == 4 (the generic kernel's default packing and tile height)"
#endif

namespace {

constexpr int HPIX = 18 * 18;
constexpr int NJ = 41;                 // 1 KB DMA pieces per chunk: 41 x 8 pixels >= 324
constexpr int CHUNK = NJ * 1024;       // 41984 B: [328 px][64 ch] bf16, rows 128 B
constexpr int NBUF = 3;
constexpr int NDMA = 11;               // pieces per wave per chunk (44 issued, the last 3 repeat piece 37..39: equal counts per wave)
constexpr int EPI_OFF = NBUF * CHUNK;  // 125952
constexpr int SLICE = 8192;            // per-wave epilogue slice: [128 px][32 ch] bf16, rows 64 B
constexpr int ST_OFF = EPI_OFF + 4 * SLICE;
constexpr int RS_LDS = ST_OFF + 16 * 4 + 64 * 4;  // + stats exchange [2 nh][8] + bias [64]
constexpr int NSTORE = 8;              // output store instructions per wave per tile
#ifndef RS_DIST
#define RS_DIST 1                      // chunks between a DMA's issue and its use
#endif
#ifndef RS_S0
#define RS_S0 0                        // compute step of the first DMA piece of a chunk
#endif
#ifndef RS_SS
#define RS_SS 1                        // steps between pieces
#endif
constexpr int DIST = RS_DIST, DMA_S0 = RS_S0, DMA_SS = RS_SS, DMA_LAST = DMA_S0 + (NDMA - 1) * DMA_SS;
static_assert(DMA_LAST < 36, "all pieces of a chunk are issued within one compute phase");
// output stores are issued at steps 3 + 4 i of a K-half-0 chunk: how many of them come after the chunk's last DMA piece
constexpr int stores_after_dma() {
    int n = 0;
    for (int i = 0; i < NSTORE; i++) n += (3 + 4 * i >= DMA_LAST) ? 1 : 0;
    return n;
}
constexpr int STORES_AFTER = stores_after_dma();

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) { return (unsigned)(uintptr_t)(const lds_byte*)p; }
template <int OFF> __device__ __forceinline__ void ds_read16(i32x4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 immediate offset is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lgkm_wait(i32x4& a, i32x4& b, i32x4& c, i32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
#if defined(RS_NO_DMA) || defined(RS_NO_EPI)  // ablation builds: the operation counts below do not hold
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// one 1 KB LDS-DMA piece: lane l reads 16 B at sbase + voff and they land at LDS address lds + 16 l.  M0 carries the LDS
// base; the compiler reserves M0, so it is saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds)
                 : "memory");
}

#ifdef RS_TRACE
__device__ unsigned long long rs_trace[4 * 64 * 4];  // [wave][chunk][top, after barrier, after compute, after epilogue] of block RS_TRACE_BLOCK
__device__ unsigned long long rs_blk[256 * 2];
#ifndef RS_TRACE_BLOCK
#define RS_TRACE_BLOCK 36
#endif
#define RS_STAMP(c, slot) if (bid == RS_TRACE_BLOCK && (c) < 64) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) rs_trace[(wave * 64 + (c)) * 4 + (slot)] = t_; }
#else
#define RS_STAMP(c, slot)
#endif

struct RsParams {
    const bf16* src[2];     // source tensor of K half 0 / 1, already offset to its first channel
    long long bstride[2];   // elements between images
    const bf16* w;          // generic packing for (BN=64, KC=16): [chunk 8][tap 9][n 64][16]
    const float* bias;
    bf16* out;
    float* stats;
    int cpix;               // channels per source pixel (64: two 64-channel tensors; 128: one 128-channel tensor)
    int B, H, W, up2, tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_rs128_kernel(RsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, nh = wave & 1;
    const int r = lane & 31, hf = lane >> 5;
    const int Ws = p.up2 ? p.W >> 1 : p.W;
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int K = (p.ntiles - bid + G - 1) / G;  // tiles of this block: bid, bid + G, ...  (every one of them valid)
    const unsigned smem_l = lds_addr(smem);
    float* st_l = reinterpret_cast<float*>(smem + ST_OFF);
    float* bias_l = st_l + 16;
    if (tid < 64) bias_l[tid] = p.bias[tid];
#ifdef RS_TRACE
    const unsigned long long rs_t0 = __builtin_amdgcn_s_memtime(), rs_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- this wave's 32 x 1152 weight slice: A fragments (rows = channels nh*32 + r, k = 8 hf .. 8 hf + 7 of 16-chunk kk)
    Frag<bf16> wr[2][9][4];
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                frag_load<bf16>(wr[kh][tap][ks], p.w + ((size_t)((kh * 4 + ks) * 9 + tap) * 64 + nh * 32 + r) * 16 + hf * 8);

    // A tile's coordinates, decoded ONCE (two integer divisions) when its prefetch is set up and carried as scalars:
    // linear tile index -> (image, tile inside the image), rotated per image so that every block sees the same mix of
    // border and interior tiles (see conv3x3_ws.hip)
    struct Tile {
        int b, tl, ty0, tx0;
        bool interior;
    };
    auto decode = [&](int t) {
        Tile q;
        q.b = t / tiles_img;
        q.tl = t - q.b * tiles_img;
        q.tl = (int)(((unsigned)q.tl + (unsigned)q.b * 37u) % (unsigned)tiles_img);
        const int ty = q.tl / p.tiles_x;
        q.ty0 = ty * 16;
        q.tx0 = (q.tl - ty * p.tiles_x) * 16;
        q.interior = q.ty0 > 0 && q.tx0 > 0 && q.ty0 + 16 < p.H && q.tx0 + 16 < p.W;
        return q;
    };
    auto tile_of = [&](int k) { return bid + k * G; };

    // ---- DMA geometry: piece i of this wave is chunk piece j = wave + 4 i (i = 10 exists for wave 0 only; the other waves
    // repeat their piece i = 9 so that every wave issues NDMA pieces per chunk).  Lane l of a piece fills LDS slot sv = l & 7
    // of halo pixel hp = 8 j + (l >> 3); the swizzle makes it read source slot sv ^ ((hx >> 1) & 7).
    // off[] = per-lane byte offsets of the pieces from a scalar base: for an interior tile the base is the halo origin and
    // the offsets do not depend on the tile; for a border tile the base is the image and the offsets are clamped into it
    // (zero_fill() overwrites the out-of-image pixels once the DMA has landed).  They are recomputed only when the kind
    // of tile changes, OUTSIDE compute(): nothing inside the MFMA loop may load a kernel argument (an s_load in flight
    // would break the counted lgkmcnt waits of the fragment ring).
    auto piece_j = [&](int i) { return (wave + 4 * i < NJ) ? wave + 4 * i : wave + 36; };
    unsigned off[NDMA];
    bool off_interior = false;
    auto dma_setup = [&](const Tile& q, const bf16* (&base)[2]) {
        int lq = lane >> 3, sv = lane & 7;
        asm volatile("" : "+v"(lq), "+v"(sv));  // recompute per call instead of carrying hoisted address math in registers
        size_t origin = 0;
        if (q.interior) {
            const int sy0 = p.up2 ? (q.ty0 - 1) >> 1 : q.ty0 - 1, sx0 = p.up2 ? (q.tx0 - 1) >> 1 : q.tx0 - 1;
            origin = ((size_t)sy0 * Ws + sx0) * p.cpix;
            if (!off_interior) {
#pragma unroll
                for (int i = 0; i < NDMA; i++) {
                    const int hp = min(piece_j(i) * 8 + lq, HPIX - 1);
                    const int hy = hp / 18, hx = hp - hy * 18;
                    const int ry = p.up2 ? (hy + 1) >> 1 : hy, rx = p.up2 ? (hx + 1) >> 1 : hx;
                    off[i] = (unsigned)(((ry * Ws + rx) * p.cpix + ((sv ^ ((hx >> 1) & 7)) << 3)) * 2);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NDMA; i++) {
                const int hp = min(piece_j(i) * 8 + lq, HPIX - 1);
                const int hy = hp / 18, hx = hp - hy * 18;
                const int Yc = min(max(q.ty0 + hy - 1, 0), p.H - 1), Xc = min(max(q.tx0 + hx - 1, 0), p.W - 1);
                const int sy = p.up2 ? Yc >> 1 : Yc, sx = p.up2 ? Xc >> 1 : Xc;
                off[i] = (unsigned)(((sy * Ws + sx) * p.cpix + ((sv ^ ((hx >> 1) & 7)) << 3)) * 2);
            }
        }
        off_interior = q.interior;
#pragma unroll
        for (int kh = 0; kh < 2; kh++) base[kh] = p.src[kh] + (size_t)q.b * p.bstride[kh] + origin;
    };
    auto issue_dma = [&](const bf16* base, int c, int i) {  // piece i of chunk c into ring buffer c % 3
#ifndef RS_NO_DMA
        dma16(base, off[i], smem_l + (unsigned)((c % NBUF) * CHUNK + piece_j(i) * 1024));
#endif
    };
    auto zero_fill = [&](const Tile& q, int c) {  // after this wave's pieces of chunk c have landed, before the barrier that publishes them
        if (!q.interior) {
            int lq = lane >> 3, ln = lane;
            asm volatile("" : "+v"(lq), "+v"(ln));
            unsigned char* buf = smem + (c % NBUF) * CHUNK;
#pragma unroll
            for (int i = 0; i < NDMA; i++) {
                const int j = piece_j(i), hp = j * 8 + lq;
                const int hy = hp / 18, hx = hp - hy * 18;
                const int Y = q.ty0 + hy - 1, X = q.tx0 + hx - 1;
                if (hp < HPIX && !(Y >= 0 && Y < p.H && X >= 0 && X < p.W)) *reinterpret_cast<i32x4*>(buf + j * 1024 + ln * 16) = i32x4{0, 0, 0, 0};
            }
        }
    };

    // ---- second half of the epilogue of the PREVIOUS tile, run between the MFMAs of the next chunk (or alone after the last
    // tile): store row i of the wave's slice (tile row 8 ph + i: 16 pixels x 64 B; lane l: pixel l >> 2, 16-byte slot l & 3)
    // and accumulate the GroupNorm sums of the slot's 8 channels (group nh * 4 + (l & 3)) from the stored bf16 values.
    struct Epi {
        bf16* outw;      // scalar: out + ((b * H + ty0 + 8 ph) * W + tx0) * 64 + nh * 32
        unsigned rowb;   // scalar: W * 128 (bytes between tile rows)
        float s, q;      // per-lane running sum / sum of squares
        bool stats;
    };
    auto epi_read = [&](int i, i32x4& od) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int pl = i * 16 + (ln >> 2);
        od = *reinterpret_cast<const i32x4*>(smem + EPI_OFF + wave * SLICE + pl * 64 + (((ln & 3) ^ ((pl >> 1) & 3)) << 4));
    };
    auto epi_store = [&](Epi& e, int i, const i32x4& od) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        *reinterpret_cast<i32x4*>(reinterpret_cast<unsigned char*>(e.outw) + (size_t)i * e.rowb + (unsigned)((ln >> 2) * 128 + (ln & 3) * 16)) = od;
        if (e.stats) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float x0 = __builtin_bit_cast(float, od[j] << 16), x1 = __builtin_bit_cast(float, od[j] & (int)0xffff0000);
                e.s += x0;
                e.q = fmaf(x0, x0, e.q);
                e.s += x1;
                e.q = fmaf(x1, x1, e.q);
            }
        }
    };
    // wave totals of (s, q) per slot: lanes with equal (lane & 3) are reduced; afterwards lanes 0..3 hold group nh * 4 + lane.
    // The lower pixel half parks them in LDS, the upper one adds them one barrier later (stats_combine).
    float st_keep_s = 0.f, st_keep_q = 0.f;
    auto epi_finish = [&](Epi& e) {
        if (e.stats) {
            float s = e.s, q = e.q;
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {
                s += __shfl_xor(s, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (ph == 1) {
                if (lane < 4) {
                    st_l[nh * 8 + lane] = s;
                    st_l[nh * 8 + 4 + lane] = q;
                }
            } else {
                st_keep_s = s;
                st_keep_q = q;
            }
        }
    };
    auto stats_combine = [&](const Tile& q) {  // after the barrier that follows epi_finish of tile q
        if (p.stats && ph == 0 && lane < 4) {
            float* dst = p.stats + (((size_t)q.b * 8 + nh * 4 + lane) * tiles_img + q.tl) * 2;
            dst[0] = st_keep_s + st_l[nh * 8 + lane];
            dst[1] = st_keep_q + st_l[nh * 8 + 4 + lane];
        }
    };

    // ---- compute: 36 (tap, ks) steps of 4 MFMAs on ring buffer c % 3 with the weights of K half KH; the pieces of chunk c + 2
    // are issued between the steps, and so is the second half of the previous tile's epilogue (KH = 0 only).
    // B fragments: pixel tile m = tile rows 8 ph + 2 m, + 1 (32 pixels), k slot 2 ks + hf.
    f32x16 acc[4];
    auto compute = [&](auto khc, int c, const bf16* pf_base, bool do_epi, Epi& epi) {  // pf_base: source of chunk c + 2, or nullptr
        constexpr int KH = decltype(khc)::value;
#ifndef RS_PF
#define RS_PF 1
#endif
        constexpr int PF = RS_PF, NST = 36;
        i32x4 fb[PF + 1][4];
        i32x4 od = i32x4{0, 0, 0, 0};
        const unsigned hbase = smem_l + (unsigned)((c % NBUF) * CHUNK);
        int rr = r;
        asm volatile("" : "+v"(rr));
        const int py0 = ph * 8 + (rr >> 4), px0 = rr & 15;
        unsigned haddr[3][4];  // [kx][ks]; tap row and pixel tile are immediates
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            const int hx = px0 + kx;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) haddr[kx][ks] = hbase + (py0 * 18 + hx) * 128 + (((ks * 2 + hf) ^ ((hx >> 1) & 7)) << 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // start counting the ring's reads from zero
        auto fetch = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            constexpr int tap = st >> 2, ks = st & 3, ky = tap / 3, kx = tap - ky * 3, slot = st % (PF + 1);
            ds_read16<(ky + 0) * 18 * 128>(fb[slot][0], haddr[kx][ks]);
            ds_read16<(ky + 2) * 18 * 128>(fb[slot][1], haddr[kx][ks]);
            ds_read16<(ky + 4) * 18 * 128>(fb[slot][2], haddr[kx][ks]);
            ds_read16<(ky + 6) * 18 * 128>(fb[slot][3], haddr[kx][ks]);
        };
        static_for<0, PF>([&](auto i) { fetch(i); });
        static_for<0, NST>([&](auto stc) {
            constexpr int st = decltype(stc)::value, slot = st % (PF + 1), tap = st >> 2, ks = st & 3;
            if constexpr (st + PF < NST) fetch(std::integral_constant<int, (st + PF < NST ? st + PF : 0)>{});
            if constexpr (st >= DMA_S0 && (st - DMA_S0) % DMA_SS == 0 && (st - DMA_S0) / DMA_SS < NDMA) {
                if (pf_base) issue_dma(pf_base, c + DIST, (st - DMA_S0) / DMA_SS);  // scalar branch
            }
            if constexpr (KH == 0) {
                // previous tile's row i: read at step 1 + 4 i, stored (and summed) two steps later.  These are ordinary LDS /
                // global operations in program order after the ring's reads: they can only make a counted wait stricter.
                if constexpr (st % 4 == 1 && st / 4 < NSTORE) {
                    if (do_epi) epi_read(st / 4, od);
                }
                if constexpr (st % 4 == 3 && st / 4 < NSTORE) {
                    if (do_epi) epi_store(epi, st / 4, od);
                }
            }
            constexpr int ahead = (NST - 1 - st) < PF ? (NST - 1 - st) : PF;
            lgkm_wait<4 * ahead>(fb[slot][0], fb[slot][1], fb[slot][2], fb[slot][3]);
#pragma unroll
            for (int m = 0; m < 4; m++)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[KH][tap][ks].v, __builtin_bit_cast(bf16x8, fb[slot][m]), acc[m], 0, 0, 0);
        });
    };

    // ---- first half of the epilogue (the only part the MFMA pipe waits for): accumulators (bias included) -> bf16 -> the
    // wave's own LDS slice [128 px][32 ch] (rows 64 B, 16-byte slots XOR-swizzled by (px >> 1) & 3), accumulators back to bias
    auto acc_init = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + nh * 32 + 8 * q4 + 4 * (ln >> 5));
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[m][4 * q4 + j] = bv[j];
        }
    };
    auto epi_regs = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA retired before the accumulators are read
        __builtin_amdgcn_sched_barrier(0);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        unsigned char* os = smem + EPI_OFF + wave * SLICE;
        const int r = ln & 31, hf = ln >> 5;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int pl = m * 32 + r;
                *reinterpret_cast<bf16x4*>(os + pl * 64 + ((q4 ^ ((pl >> 1) & 3)) << 4) + hf * 8) =
                    bf16x4{(bf16)acc[m][4 * q4], (bf16)acc[m][4 * q4 + 1], (bf16)acc[m][4 * q4 + 2], (bf16)acc[m][4 * q4 + 3]};
            }
        acc_init();
    };
    auto epi_setup = [&](const Tile& q, Epi& e) {
        e.outw = p.out + (((size_t)q.b * p.H + q.ty0 + ph * 8) * p.W + q.tx0) * 64 + nh * 32;
        e.rowb = (unsigned)p.W * 128u;
        e.s = e.q = 0.f;
        e.stats = p.stats != nullptr;
    };

    // ---- main loop.  Vector-memory operations of the steady state, per wave: the NDMA pieces of a chunk (issued two chunks
    // ahead, during compute) and the NSTORE output stores of a tile (issued during the NEXT tile's first chunk); the counted
    // vmcnt waits below rely on exactly these numbers.  Allowing FEWER outstanding operations than were issued is always
    // safe, so the single statistics store of the ph = 0 waves and the last store of a chunk are left out of the counts.
    vm_wait<0>();  // weights in registers
    __syncthreads();  // bias in LDS
    acc_init();
    Tile cur = {}, nxt = {}, prv = {};
    const bf16* base[2] = {nullptr, nullptr};      // sources of the tile whose pieces are issued next
    Epi epi = {};
    if (K > 0) {
        cur = decode(tile_of(0));
        dma_setup(cur, base);
#pragma unroll
        for (int i = 0; i < NDMA; i++) issue_dma(base[0], 0, i);
        if (DIST == 2) {
#pragma unroll
            for (int i = 0; i < NDMA; i++) issue_dma(base[1], 1, i);
        }
    }
    for (int k = 0; k < K; k++) {
        const bool more = k + 1 < K;
        // ---- chunk 2k (K half 0).  Issued after its pieces - DIST 2: the pieces of chunk 2k + 1 and the late stores of
        // tile k - 2's... (none: K-half-1 chunks issue no stores) ; DIST 1: nothing (chunk 2k - 1 issues no stores)
        RS_STAMP(2 * k, 0)
        if (DIST == 2) vm_wait<NDMA>();
        else vm_wait<0>();
        zero_fill(cur, 2 * k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RS_STAMP(2 * k, 1)
        const bf16* pf0;  // what compute(2k) prefetches: chunk 2k + DIST
        if (DIST == 2) {
            if (more) {
                nxt = decode(tile_of(k + 1));
                dma_setup(nxt, base);  // offsets + bases of chunks 2k + 2, 2k + 3 (overwrites off[]: tile k's pieces are all issued)
            }
            pf0 = more ? base[0] : nullptr;
        } else {
            pf0 = base[1];  // chunk 2k + 1 of the same tile
        }
        if (k > 0) epi_setup(prv, epi);
        compute(std::integral_constant<int, 0>{}, 2 * k, pf0, k > 0, epi);
        if (k > 0) epi_finish(epi);
        RS_STAMP(2 * k, 2)
        // ---- chunk 2k + 1 (K half 1).  Issued after its pieces - DIST 2: in chunk 2k the pieces of chunk 2k + 2 (more) and
        // the NSTORE stores of tile k - 1 (k > 0); DIST 1: the stores of tile k - 1 that follow the last piece
        RS_STAMP(2 * k + 1, 0)
        if (DIST == 2) {
            if (more) {
                if (k == 0) vm_wait<NDMA>();
                else vm_wait<NDMA + NSTORE>();
            } else {
                vm_wait<0>();
            }
        } else {
            if (k == 0) vm_wait<0>();
            else vm_wait<STORES_AFTER>();
        }
        zero_fill(cur, 2 * k + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RS_STAMP(2 * k + 1, 1)
        if (k > 0) stats_combine(prv);
        const bf16* pf1;
        if (DIST == 2) {
            pf1 = more ? base[1] : nullptr;
        } else {
            if (more) {
                nxt = decode(tile_of(k + 1));
                dma_setup(nxt, base);  // tile k + 1: chunk 2k + 2 now, chunk 2k + 3 during compute(2k + 2)
            }
            pf1 = more ? base[0] : nullptr;
        }
        compute(std::integral_constant<int, 1>{}, 2 * k + 1, pf1, false, epi);
        RS_STAMP(2 * k + 1, 2)
#ifndef RS_NO_EPI
        epi_regs();
#else
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
#endif
        RS_STAMP(2 * k + 1, 3)
        prv = cur;
        cur = nxt;
    }
    // the last tile's stores and sums have no next chunk to hide under
    if (K > 0) {
#ifndef RS_NO_EPI
        epi_setup(prv, epi);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < NSTORE; i++) {
            i32x4 od;
            epi_read(i, od);
            epi_store(epi, i, od);
        }
        epi_finish(epi);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stats_combine(prv);
    }
#ifdef RS_TRACE
    if (tid == 0) {
        rs_blk[bid * 2] = __builtin_amdgcn_s_memtime() - rs_t0;
        rs_blk[bid * 2 + 1] = __builtin_amdgcn_s_memrealtime() - rs_r0;
    }
#endif
}

}  // namespace

// eligibility: bf16, 128 input channels (64 | 64 concat or one 128-channel tensor), 64 outputs, no GroupNorm prologue,
// 8 GroupNorm groups, 16-aligned output size
bool conv3x3_rs128_eligible(int H, int W, int C0, int C1, int Cout, int groups, int dtype, bool has_aff) {
    return dtype == DIG_BF16 && !has_aff && Cout == 64 && groups == 8 && H % 16 == 0 && W % 16 == 0 &&
           ((C0 == 64 && C1 == 64) || (C0 == 128 && C1 == 0));
}

int conv3x3_rs128_launch(const dig_conv3x3_args* a, hipStream_t stream) {
    RsParams p;
    const int C1 = a->in1 ? a->C1 : 0;
    const long long src_px = (long long)(a->up2 ? (a->H / 2) * (a->W / 2) : a->H * a->W);
    p.cpix = a->C0;
    p.src[0] = (const bf16*)a->in0;
    p.bstride[0] = a->in0_bstride > 0 ? a->in0_bstride : src_px * a->C0;
    if (C1) {
        p.src[1] = (const bf16*)a->in1;
        p.bstride[1] = src_px * C1;
    } else {
        p.src[1] = (const bf16*)a->in0 + 64;
        p.bstride[1] = p.bstride[0];
    }
    p.w = (const bf16*)a->w;
    p.bias = a->bias;
    p.out = (bf16*)a->out;
    p.stats = a->stats;
    p.B = a->B; p.H = a->H; p.W = a->W; p.up2 = a->up2;
    p.tiles_x = a->W / 16;
    p.tiles_y = a->H / 16;
    p.ntiles = a->B * p.tiles_x * p.tiles_y;
    // 32-bit byte offsets inside one image of a source
    DIG_CHECK(src_px * a->C0 * 2 < (1ll << 31), DIG_EINVAL, "conv3x3(rs128): source image too large for 32-bit offsets");
    static std::atomic<int> ncu[64];  // zero-initialised; racing first calls both query and store the same value
    int device = 0;
    (void)hipGetDevice(&device);
    const int dev = device & 63;
    int ncus = ncu[dev].load(std::memory_order_acquire);
    if (!ncus) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_rs128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "conv3x3(rs128): cannot raise dynamic LDS to %d: %s", RS_LDS, hipGetErrorString(e));
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
        DIG_CHECK(e == hipSuccess && n > 0, DIG_ELAUNCH, "conv3x3(rs128): cannot query the CU count");
        ncu[dev].store(n, std::memory_order_release);
        ncus = n;
    }
    int grid = p.ntiles < ncus ? p.ntiles : ncus;
    if (const char* g = getenv("DIG_RS_GRID")) {  // test hook: a small grid makes every block walk many tiles
        const int v = atoi(g);
        if (v > 0 && v < grid) grid = v;
    }
    hipLaunchKernelGGL(conv3x3_rs128_kernel, dim3(grid), dim3(256), RS_LDS, stream, p);
    DIG_LAUNCH_CHECK("conv3x3(rs128)");
    return DIG_OK;
}

#ifdef RS_TRACE
extern "C" int dig_debug_rs_trace(unsigned long long* host, unsigned long long* blk) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(rs_trace), sizeof(unsigned long long) * 4 * 64 * 4);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(blk, HIP_SYMBOL(rs_blk), sizeof(unsigned long long) * 256 * 2);
}
#endif
