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
//     (global_load_lds_dwordx4: no registers, no VALU, no LDS store instructions) two chunks ahead of the MFMAs.  Rows are
//     128 B with the 16-byte slots XOR-swizzled by (hx >> 1) & 7 - applied to the per-lane SOURCE address, the LDS side of
//     a DMA being linear - so every ds_read_b128 fragment read is conflict-free.  Fused here: channel concat, nearest-x2
//     upsampling (source addressing), zero padding (border tiles: out-of-image pixels are zeroed in LDS after the DMA landed).
//   * one s_barrier per chunk; DMA completion is tracked with counted s_waitcnt vmcnt(N) (the DMAs and the output stores are
//     the only vector-memory operations of the steady state, their numbers per chunk are fixed).
//   * per tile: accumulators (+bias) -> bf16 -> the wave's own 8 KB LDS slice -> 16-byte stores of 64-byte half rows (both
//     halves of a 128-byte line are written by the two nh waves within the same tile, they merge in L2); GroupNorm partial
//     sums from the fp32 accumulators, one (sum, sumsq) pair per (tile, group) - the two pixel halves are combined through
//     LDS one barrier later, so the partial count equals the generic kernel's (dig_conv3x3_tiles is unchanged).
// A/B fragment reads use the inline-asm ring of conv3x3_ws.hip (tools/check_async_lds_asm.py checks this kernel as well).
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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

    // ---- this wave's 32 x 1152 weight slice: A fragments (rows = channels nh*32 + r, k = 8 hf .. 8 hf + 7 of 16-chunk kk)
    Frag<bf16> wr[2][9][4];
#pragma unroll
    for (int kh = 0; kh < 2; kh++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                frag_load<bf16>(wr[kh][tap][ks], p.w + ((size_t)((kh * 4 + ks) * 9 + tap) * 64 + nh * 32 + r) * 16 + hf * 8);

    // linear tile index -> (image, tile inside the image), rotated per image so that every block sees the same mix of
    // border and interior tiles (see conv3x3_ws.hip)
    auto decode = [&](int t, int& b, int& ty0, int& tx0, int& tl) {
        b = t / tiles_img;
        tl = t - b * tiles_img;
        tl = (int)(((unsigned)tl + (unsigned)b * 37u) % (unsigned)tiles_img);
        ty0 = (tl / p.tiles_x) * 16;
        tx0 = (tl % p.tiles_x) * 16;
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
    struct Dma {
        const bf16* base[2];  // scalar: K half 0 / 1 of the tile being prefetched
    };
    auto dma_setup = [&](int t, Dma& d) {
        int lq = lane >> 3, sv = lane & 7;
        asm volatile("" : "+v"(lq), "+v"(sv));  // recompute per call instead of carrying hoisted address math in registers
        int b, ty0, tx0, tl;
        decode(t, b, ty0, tx0, tl);
        const bool interior = ty0 > 0 && tx0 > 0 && ty0 + 16 < p.H && tx0 + 16 < p.W;  // scalar
        size_t origin = 0;
        if (interior) {
            const int sy0 = p.up2 ? (ty0 - 1) >> 1 : ty0 - 1, sx0 = p.up2 ? (tx0 - 1) >> 1 : tx0 - 1;
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
                const int Yc = min(max(ty0 + hy - 1, 0), p.H - 1), Xc = min(max(tx0 + hx - 1, 0), p.W - 1);
                const int sy = p.up2 ? Yc >> 1 : Yc, sx = p.up2 ? Xc >> 1 : Xc;
                off[i] = (unsigned)(((sy * Ws + sx) * p.cpix + ((sv ^ ((hx >> 1) & 7)) << 3)) * 2);
            }
        }
        off_interior = interior;
#pragma unroll
        for (int kh = 0; kh < 2; kh++) d.base[kh] = p.src[kh] + (size_t)b * p.bstride[kh] + origin;
    };
    auto issue_dma = [&](const bf16* base, int c, int i) {  // piece i of chunk c into ring buffer c % 3
#ifndef RS_NO_DMA
        dma16(base, off[i], smem_l + (unsigned)((c % NBUF) * CHUNK + piece_j(i) * 1024));
#endif
    };
    auto zero_fill = [&](int c) {  // after this wave's pieces of chunk c have landed, before the barrier that publishes them
        int b, ty0, tx0, tl;
        decode(tile_of(c >> 1), b, ty0, tx0, tl);
        const bool interior = ty0 > 0 && tx0 > 0 && ty0 + 16 < p.H && tx0 + 16 < p.W;
        if (!interior) {
            int lq = lane >> 3, ln = lane;
            asm volatile("" : "+v"(lq), "+v"(ln));
            unsigned char* buf = smem + (c % NBUF) * CHUNK;
#pragma unroll
            for (int i = 0; i < NDMA; i++) {
                const int j = piece_j(i), hp = j * 8 + lq;
                const int hy = hp / 18, hx = hp - hy * 18;
                const int Y = ty0 + hy - 1, X = tx0 + hx - 1;
                if (hp < HPIX && !(Y >= 0 && Y < p.H && X >= 0 && X < p.W)) *reinterpret_cast<i32x4*>(buf + j * 1024 + ln * 16) = i32x4{0, 0, 0, 0};
            }
        }
    };

    // ---- compute: 36 (tap, ks) steps of 4 MFMAs on ring buffer c % 3 with the weights of K half KH; the pieces of chunk c + 2
    // are issued between the steps.  B fragments: pixel tile m = tile rows 8 ph + 2 m, + 1 (32 pixels), k slot 2 ks + hf.
    f32x16 acc[4];
    auto compute = [&](auto khc, int c, const bf16* pf_base) {  // pf_base: source of chunk c + 2, or nullptr
        constexpr int KH = decltype(khc)::value;
#ifndef RS_PF
#define RS_PF 1
#endif
        constexpr int PF = RS_PF, NST = 36;
        i32x4 fb[PF + 1][4];
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
            if constexpr (st % 3 == 0 && st / 3 < NDMA) {
                if (pf_base) issue_dma(pf_base, c + 2, st / 3);  // scalar branch
            }
            constexpr int ahead = (NST - 1 - st) < PF ? (NST - 1 - st) : PF;
            lgkm_wait<4 * ahead>(fb[slot][0], fb[slot][1], fb[slot][2], fb[slot][3]);
#pragma unroll
            for (int m = 0; m < 4; m++)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[KH][tap][ks].v, __builtin_bit_cast(bf16x8, fb[slot][m]), acc[m], 0, 0, 0);
        });
    };

    // ---- epilogue of a finished tile (wave-local: its own LDS slice, its own 64-byte half rows of the output)
    float st_keep = 0.f;
    auto epilogue = [&](int t) {
        int b, ty0, tx0, tl;
        decode(t, b, ty0, tx0, tl);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA retired before the accumulators are read
        __builtin_amdgcn_sched_barrier(0);
        unsigned char* os = smem + EPI_OFF + wave * SLICE;
        int ln = lane;
        asm volatile("" : "+v"(ln));  // per-lane epilogue addresses are recomputed per tile (not hoisted and kept live)
        const int r = ln & 31, hf = ln >> 5, lane = ln;
        float st[8];  // [kind][q4]
#pragma unroll
        for (int j = 0; j < 8; j++) st[j] = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + nh * 32 + 8 * q4 + 4 * hf);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    v[j] = acc[m][4 * q4 + j] + bv[j];
                    st[q4] += v[j];
                    st[4 + q4] += v[j] * v[j];
                }
                const int pl = m * 32 + r;
                *reinterpret_cast<bf16x4*>(os + pl * 64 + ((q4 ^ ((pl >> 1) & 3)) << 4) + hf * 8) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            }
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][e] = 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // store i = tile row 8 ph + i: 16 pixels x 64 B (lane l: pixel l >> 2, 16-byte slot l & 3)
        bf16* outw = p.out + (((size_t)b * p.H + ty0 + ph * 8) * p.W + tx0) * 64 + nh * 32;  // scalar
        const unsigned lo = (unsigned)((lane >> 2) * 64 + (lane & 3) * 8);
        bf16x8 od[NSTORE];
#pragma unroll
        for (int i = 0; i < NSTORE; i++) {
            const int pl = i * 16 + (lane >> 2);
            od[i] = *reinterpret_cast<const bf16x8*>(os + pl * 64 + (((lane & 3) ^ ((pl >> 1) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < NSTORE; i++) *reinterpret_cast<bf16x8*>(outw + (size_t)i * p.W * 64 + lo) = od[i];
        if (p.stats) {
            // 8 per-lane partials -> wave totals: halving butterfly on lane bits 5, 4, 3 (lane then holds value
            // j = 4 b5 + 2 b4 + b3 of its 8-lane class), three plain steps finish it; lanes with (lane & 7) == 0 hold j = lane >> 3
            float a4[4], a2[2], a1;
            {
                const bool up = (lane >> 5) & 1;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float keep = up ? st[4 + j] : st[j], send = up ? st[j] : st[4 + j];
                    a4[j] = keep + __shfl_xor(send, 32, 64);
                }
            }
            {
                const bool up = (lane >> 4) & 1;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float keep = up ? a4[2 + j] : a4[j], send = up ? a4[j] : a4[2 + j];
                    a2[j] = keep + __shfl_xor(send, 16, 64);
                }
            }
            {
                const bool up = (lane >> 3) & 1;
                const float keep = up ? a2[1] : a2[0], send = up ? a2[0] : a2[1];
                a1 = keep + __shfl_xor(send, 8, 64);
            }
            a1 += __shfl_xor(a1, 4, 64);
            a1 += __shfl_xor(a1, 2, 64);
            a1 += __shfl_xor(a1, 1, 64);
            // the lower pixel half parks its totals in LDS; the upper one adds them one barrier later (stats_combine)
            if (ph == 1) {
                if ((lane & 7) == 0) st_l[nh * 8 + (lane >> 3)] = a1;
            } else {
                st_keep = a1;
            }
        }
    };
    auto stats_combine = [&](int t) {  // after the barrier that follows epilogue(t)
        if (p.stats && ph == 0 && (lane & 7) == 0) {
            int b, ty0, tx0, tl;
            decode(t, b, ty0, tx0, tl);
            const int j = lane >> 3, kind = j >> 2, g = nh * 4 + (j & 3);
            p.stats[(((size_t)b * 8 + g) * tiles_img + tl) * 2 + kind] = st_keep + st_l[nh * 8 + j];
        }
    };

#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[m][e] = 0.f;
    // prologue: weights in registers, then chunks 0 and 1 (tile 0) in flight
    vm_wait<0>();
#ifdef RS_TRACE
    const unsigned long long rs_t0 = __builtin_amdgcn_s_memtime(), rs_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    Dma d;
    if (K > 0) {
        dma_setup(tile_of(0), d);
#pragma unroll
        for (int i = 0; i < NDMA; i++) issue_dma(d.base[0], 0, i);
#pragma unroll
        for (int i = 0; i < NDMA; i++) issue_dma(d.base[1], 1, i);
    }
    for (int k = 0; k < K; k++) {
        const bool more = k + 1 < K;
        // ---- chunk 2k (K half 0).  Vector-memory operations issued after its pieces: the pieces of chunk 2k + 1 (NDMA)
        // and, from the second tile on, the NSTORE output stores of tile k - 1 (the statistics store of the ph = 0 waves is
        // one more: allowing FEWER outstanding operations than were issued is always safe).
        RS_STAMP(2 * k, 0)
        if (k == 0) vm_wait<NDMA>();
        else vm_wait<NDMA + NSTORE>();
        zero_fill(2 * k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RS_STAMP(2 * k, 1)
        if (k > 0) stats_combine(tile_of(k - 1));
        if (more) dma_setup(tile_of(k + 1), d);  // offsets + bases of chunks 2k + 2, 2k + 3 (overwrites off[]: tile k's pieces are all issued)
        compute(std::integral_constant<int, 0>{}, 2 * k, more ? d.base[0] : nullptr);
        RS_STAMP(2 * k, 2)
        RS_STAMP(2 * k + 1, 0)
        // ---- chunk 2k + 1 (K half 1): after its pieces came the stores of tile k - 1 (k > 0) and the pieces of chunk 2k + 2 (more)
        if (more) {
            if (k == 0) vm_wait<NDMA>();
            else vm_wait<NDMA + NSTORE>();
        } else {
            vm_wait<0>();
        }
        zero_fill(2 * k + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RS_STAMP(2 * k + 1, 1)
        compute(std::integral_constant<int, 1>{}, 2 * k + 1, more ? d.base[1] : nullptr);
        RS_STAMP(2 * k + 1, 2)
#ifndef RS_NO_EPI
        epilogue(tile_of(k));
#else
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
#endif
        RS_STAMP(2 * k + 1, 3)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (K > 0) stats_combine(tile_of(K - 1));
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
    static int ncu[64] = {};
    int device = 0;
    (void)hipGetDevice(&device);
    const int dev = device & 63;
    if (!ncu[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_rs128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "conv3x3(rs128): cannot raise dynamic LDS to %d: %s", RS_LDS, hipGetErrorString(e));
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
        DIG_CHECK(e == hipSuccess && n > 0, DIG_ELAUNCH, "conv3x3(rs128): cannot query the CU count");
        ncu[dev] = n;
    }
    int grid = p.ntiles < ncu[dev] ? p.ntiles : ncu[dev];
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
