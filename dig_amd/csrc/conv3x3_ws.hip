// conv3x3 (pad 1), 64 -> 64 channels, bf16, weight-stationary persistent variant for gfx950.
//
// At the 256x256 / 128x128 levels of the U-Net a 64->64 conv has ~2M output pixels but only 72 KB of weights: the
// generic kernel (conv3x3.hip) re-stages the whole weight tensor for every 256-pixel tile and spends its time moving
// it.  Here one 512-thread block per CU keeps ALL weights in LDS for its lifetime and walks 16x16-pixel tiles:
//
//   LDS  = W [9 taps][64 n][64 k] (72 KB)  +  2 x halo tile [18x18 px][64 ch] (2 x 40.5 KB), all rows 128 B with the
//          16-byte slots XOR-swizzled by (index >> 1) & 7 so that every ds_read_b128 fragment read is conflict-free
//          without padding; + bias and the per-group GroupNorm affine.
//   The 8 waves form two groups of 4 that run half a period apart (one s_barrier per phase):
//          compute phase : 144 MFMAs per wave on the group's halo buffer (wave = 64 px x 64 ch); fragments come from
//                          inline-asm ds_read_b128 issued two steps ahead with counted lgkmcnt waits
//          stage phase   : issue the next tile's global loads, epilogue of the finished tile (GroupNorm partials, bf16
//                          transpose through the wave's own slice of the halo buffer, 16-byte NHWC stores), then the
//                          GroupNorm-apply + SiLU prologue on the prefetched registers and the LDS store of the halo
//   so the MFMA pipe of every SIMD has one wave in its compute phase while its partner does the VALU/LDS/HBM work.
//   Accumulators hold channels as ROWS and pixels as COLUMNS (A = weights, B = halo), which makes the four
//   consecutive channels of an accumulator group one 8-byte piece of an NHWC pixel row.
// GroupNorm partials are written per WAVE (4 per tile): dig_conv3x3_tiles() reports the matching count.
// tools/check_async_lds_asm.py checks the generated ISA for the one thing the compiler cannot know here: that nothing
// touches a fragment register between its inline-asm LDS read and the counted wait that retires it.
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HPIX = 18 * 18;
constexpr int W_BYTES = 9 * 64 * 128;
constexpr int HALO_BYTES = HPIX * 128;
constexpr int WS_LDS = W_BYTES + 2 * HALO_BYTES + 256 + 2 * 512;  // + bias[64] fp32 + per-group GroupNorm affine [64][2]
constexpr int NVEC = 11;  // 16-byte halo vectors per thread per tile: 8 in the wave's own 8 KB slice + 3 in the tail

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
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#ifndef WS_STAGE_PRIO
#define WS_STAGE_PRIO 2
#endif

#ifndef WS_TRACE_BLOCK
#define WS_TRACE_BLOCK 0
#endif
#ifdef WS_TRACE
// per-phase timestamps of block 0, kept in registers during the phase and written out after its barrier
__device__ unsigned long long ws_trace[8 * 64 * 8];
__device__ unsigned long long ws_blk[256 * 2];  // per block: cycles from kernel entry to exit, realtime at entry  // [wave][phase][start, role end, barrier end, stage: loads issued, E1, E2, -, -]
#define WS_STAMP(slot) ws_t[slot] = __builtin_amdgcn_s_memtime();
#define WS_FLUSH()                                                                                  \
    if (blockIdx.x == WS_TRACE_BLOCK && lane == 0 && ph + 1 < 64)                                                \
        for (int s_ = 0; s_ < 6; s_++) ws_trace[(wave * 64 + ph + 1) * 8 + s_] = ws_t[s_];
#else
#define WS_STAMP(slot)
#define WS_FLUSH()
#endif

struct WsParams {
    const bf16* in;
    const float* aff;
    const bf16* w;  // generic packing for (BN=64, KC=16): [chunk 4][tap 9][n 64][16]
    const float* bias;
    bf16* out;
    float* stats;
    long long bstride;
    int B, H, W, up2, tiles_x, tiles_y, ntiles;
#ifdef WS_ASSERT_ADDR
    long long in_elems, out_elems, stats_elems;  // extents of the three global buffers, for the debug build's address checks
#endif
};

#ifdef WS_ASSERT_ADDR
// Debug build (-DWS_ASSERT_ADDR): every global address the kernel forms is checked against the extent of
// its buffer before the access; a violation is counted (first offender recorded) and the access is redirected to element 0
// instead of being performed.  [0] = count, [1] = kind (1 halo load, 2 output store, 3 stats store), [2] = offending element
// offset, [3] = tile index.  Read back with dig_debug_ws_addr_violations().
__device__ long long ws_addr_viol[4];
#define WS_CHECK(kind, off_, lim_, tile_)                                                \
    (((long long)(off_) < 0 || (long long)(off_) > (long long)(lim_)) ?                    \
         (atomicAdd((unsigned long long*)&ws_addr_viol[0], 1ull) == 0                       \
              ? (ws_addr_viol[1] = (kind), ws_addr_viol[2] = (long long)(off_), ws_addr_viol[3] = (tile_), 0ll) : 0ll) \
         : (long long)(off_))
#endif

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_ws64_kernel(WsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: roles, tile indices and bases stay scalar
    const int grp = wave >> 2, wg = wave & 3;
    const int r = lane & 31, hf = lane >> 5;
    unsigned char* Wl = smem;
    unsigned char* Hl = smem + W_BYTES + grp * HALO_BYTES;
    const int Ws = p.up2 ? p.W >> 1 : p.W;
    const int tiles_img = p.tiles_x * p.tiles_y;

    for (int i = tid; i < 9 * 64 * 8; i += 512) {
        const int s = i & 7, n = (i >> 3) & 63, tap = i >> 9;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)((s >> 1) * 9 + tap) * 64 + n) * 16 + (s & 1) * 8);
        *reinterpret_cast<bf16x8*>(Wl + tap * 8192 + n * 128 + ((s ^ ((n >> 1) & 7)) << 4)) = v;
    }
    float* bias_l = reinterpret_cast<float*>(smem + W_BYTES + 2 * HALO_BYTES);
    if (tid < 64) bias_l[tid] = p.bias[tid];
    float* aff_l = bias_l + 64 + grp * 128;  // this group's [64 ch][scale, shift] of the image being staged
    const int tg = wg * 64 + lane;           // thread index inside the group

    const int K = (p.ntiles + 2 * (int)gridDim.x - 1) / (2 * (int)gridDim.x);
    auto tile_of = [&](int k) { return (k * (int)gridDim.x + (int)blockIdx.x) * 2 + grp; };
    // linear tile index -> (image, tile inside the image).  A block's tiles are 2 * gridDim.x apart, usually a whole number
    // of images, so without the rotation a block would sit on ONE tile position for its whole life and the blocks on the
    // image border (clamped addressing, zero fill: ~15 % more cycles per tile) would finish last.  Rotating the tile index
    // by an image-dependent offset gives every block the same mix; neighbouring blocks still work on neighbouring tiles.
    auto decode = [&](int t, int& b, int& tl) {
        b = t / tiles_img;
        tl = t - b * tiles_img;
#ifndef WS_NO_ROTATE
        tl = (int)(((unsigned)tl + (unsigned)b * 37u) % (unsigned)tiles_img);
#endif
    };

    // ---- staging geometry, fixed per thread: vector i covers halo pixel hp_i (pixel rows 64*wg..64*wg+63 are the wave's
    // own 8 KB slice, 256.. the shared tail) and channels 8*sv..8*sv+7.  goff: element offset of its source from the
    // halo origin of an INTERIOR tile (nearest-x2 folded in); loff: swizzled LDS byte offset.
    const int lq = lane >> 3, sv = lane & 7;
    int goff[NVEC], loff[NVEC];
#pragma unroll
    for (int i = 0; i < NVEC; i++) {
        const int hp_raw = i < 8 ? wg * 64 + i * 8 + lq : 256 + (i - 8) * 32 + wg * 8 + lq;
        const int hp = min(hp_raw, HPIX - 1);
        const int hy = hp / 18, hx = hp - hy * 18;
        const int ry = p.up2 ? (hy + 1) >> 1 : hy, rx = p.up2 ? (hx + 1) >> 1 : hx;
        goff[i] = (ry * Ws + rx) * 64 + sv * 8;
        loff[i] = hp_raw < HPIX ? hp * 128 + ((sv ^ ((hx >> 1) & 7)) << 4) : -1;
    }

    auto gaddr = [&](int t, unsigned (&off)[NVEC], const bf16*& base, unsigned& hok, bool& interior) {
        int b, tl;
        decode(t, b, tl);
        const int ty0 = (tl / p.tiles_x) * 16, tx0 = (tl % p.tiles_x) * 16;
        const bf16* img = p.in + (size_t)b * p.bstride;
        // one load sequence for both cases (no control flow around the loads): a scalar base + 32-bit element offsets
        interior = ty0 > 0 && tx0 > 0 && ty0 + 16 < p.H && tx0 + 16 < p.W;  // scalar
        if (interior) {
            // interior tile: the base is the halo origin, the offsets are the fixed per-thread ones
            const int sy0 = p.up2 ? (ty0 - 1) >> 1 : ty0 - 1, sx0 = p.up2 ? (tx0 - 1) >> 1 : tx0 - 1;
            base = img + ((size_t)sy0 * Ws + sx0) * 64;
#pragma unroll
            for (int i = 0; i < NVEC; i++) off[i] = (unsigned)goff[i];
            hok = 0xffffffffu;
        } else {
            // border tile: out-of-image lanes read a clamped (valid) address and are zeroed in lstore() via `hok`
            base = img;
            hok = 0;
#pragma unroll
            for (int i = 0; i < NVEC; i++) {
                const int hp = min(i < 8 ? wg * 64 + i * 8 + lq : 256 + (i - 8) * 32 + wg * 8 + lq, HPIX - 1);
                const int hy = hp / 18, hx = hp - hy * 18;
                const int Y = ty0 + hy - 1, X = tx0 + hx - 1;
                const bool ok = Y >= 0 && Y < p.H && X >= 0 && X < p.W;
                const int Yc = min(max(Y, 0), p.H - 1), Xc = min(max(X, 0), p.W - 1);
                const int sy = p.up2 ? Yc >> 1 : Yc, sx = p.up2 ? Xc >> 1 : Xc;
                off[i] = (unsigned)((sy * Ws + sx) * 64 + sv * 8);
                hok |= (ok ? 1u : 0u) << i;
            }
        }
    };
    // GroupNorm-apply + SiLU prologue: silu_pre (common.hpp) on the affine pre-scaled by -log2(e) in aff_l.  Plain scalar
    // f32 VALU on purpose: gfx950's SIMD-32 gains nothing from v_pk_*_f32 and pays extra issue cycles for them beside
    // MFMAs (MI355X_MICROARCH.md, cycle constants) - the file is built with -fno-slp-vectorize so the compiler keeps it so.
    struct Aff {
        float pa[8], pb[8];
    };
    auto lstore_setup = [&](Aff& A) {
        if (p.aff != nullptr) {
            const f32x4* af = reinterpret_cast<const f32x4*>(aff_l + sv * 16);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 v = af[j];  // (a, b) of channels 2j, 2j+1
                A.pa[2 * j] = v[0]; A.pb[2 * j] = v[1]; A.pa[2 * j + 1] = v[2]; A.pb[2 * j + 1] = v[3];
            }
        }
    };
    // one 16-byte halo vector: prologue, zero fill outside the image, swizzled LDS store
    auto lstore_vec = [&](const Aff& A, const bf16x8& prev, int i, int lo, unsigned hok, bool all) {
        if (i < 10 || lo >= 0) {
            i32x4 w = __builtin_bit_cast(i32x4, prev);
            if (p.aff != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float x0 = __builtin_bit_cast(float, w[j] << 16), x1 = __builtin_bit_cast(float, w[j] & (int)0xffff0000);
                    const float o0 = silu_pre(A.pa[2 * j], A.pb[2 * j], x0), o1 = silu_pre(A.pa[2 * j + 1], A.pb[2 * j + 1], x1);
                    w[j] = __builtin_bit_cast(int, bf16x2{(bf16)o0, (bf16)o1});
                }
            }
            if (!all) {  // scalar branch: border tiles only
                if (!(hok & (1u << i))) w = i32x4{0, 0, 0, 0};
            }
            *reinterpret_cast<i32x4*>(Hl + lo) = w;
        }
    };

    // ---- compute: wave = tile rows 4wg..4wg+3 (mr: two 32-pixel column tiles) x 64 channels (nr: two 32-row tiles).
    // The accumulators start from the bias.
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i >> 1][i & 1][e] = 0.f;
    auto compute = [&]() {
        int rr = r;
        asm volatile("" : "+v"(rr));  // recompute the fragment addresses per phase instead of carrying 20 registers
        const int py0 = wg * 4 + (rr >> 4), px0 = rr & 15;
#pragma unroll
        for (int nr = 0; nr < 2; nr++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + nr * 32 + 8 * q4 + 4 * hf);
#pragma unroll
                for (int mr = 0; mr < 2; mr++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[mr][nr][4 * q4 + j] = bv[j];
            }
        // 36 steps (tap, ks) of 4 MFMAs.  hipcc sinks plain LDS loads down to their first use (one step of latency
        // exposed per step), so the fragment reads are inline-asm ds_read_b128 issued PF steps ahead into a register
        // ring, with counted lgkmcnt waits that carry the ring registers as operands (the MFMAs depend on the wait).
#ifndef WS_PF
#define WS_PF 2
#endif
        constexpr int PF = WS_PF, NST = 36;
        i32x4 fa[PF + 1][2], fb[PF + 1][2];
        const unsigned hbase = lds_addr(Hl), wbase = lds_addr(Wl);
        unsigned haddr[3][4], waddr[2][4];  // halo [kx][ks] (tap row, mr: immediates); weights [taps 0-4 | 5-8][ks]
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            const int hx = px0 + kx;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) haddr[kx][ks] = hbase + (py0 * 18 + hx) * 128 + (((ks * 2 + hf) ^ ((hx >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            waddr[0][ks] = wbase + rr * 128 + (((ks * 2 + hf) ^ ((rr >> 1) & 7)) << 4);
            waddr[1][ks] = waddr[0][ks] + 5 * 8192;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the bias reads above are the compiler's; start counting from zero
        auto fetch = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            constexpr int tap = st >> 2, ks = st & 3, ky = tap / 3, kx = tap - ky * 3, slot = st % (PF + 1);
            ds_read16<(ky * 18) * 128>(fb[slot][0], haddr[kx][ks]);
            ds_read16<(ky * 18 + 36) * 128>(fb[slot][1], haddr[kx][ks]);
            constexpr int wt = tap < 5 ? tap : tap - 5;
            ds_read16<wt * 8192>(fa[slot][0], waddr[tap < 5 ? 0 : 1][ks]);
            ds_read16<wt * 8192 + 4096>(fa[slot][1], waddr[tap < 5 ? 0 : 1][ks]);
        };
        static_for<0, PF>([&](auto i) { fetch(i); });
        static_for<0, NST>([&](auto stc) {
            constexpr int st = decltype(stc)::value, slot = st % (PF + 1);
            if constexpr (st + PF < NST) fetch(std::integral_constant<int, (st + PF < NST ? st + PF : 0)>{});
            constexpr int ahead = (NST - 1 - st) < PF ? (NST - 1 - st) : PF;  // fetch groups still allowed in flight
            lgkm_wait<4 * ahead>(fa[slot][0], fa[slot][1], fb[slot][0], fb[slot][1]);
#pragma unroll
            for (int mr = 0; mr < 2; mr++)
#pragma unroll
                for (int nr = 0; nr < 2; nr++)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[slot][nr]),
                                                                           __builtin_bit_cast(bf16x8, fb[slot][mr]), acc[mr][nr], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMA retired before the accumulators are read (see conv3x3.hip)
        __builtin_amdgcn_sched_barrier(0);
#if defined(WS_NO_E1) || defined(WS_NO_STAGE)  // ablation builds: keep the MFMAs alive although nobody reads their result
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
#endif
    };

    // ---- epilogue of the finished tile.  E1: accumulators -> bf16 -> the wave's slice + per-lane GroupNorm partials
    // (after it the accumulators are dead); E2: slice -> 16-byte NHWC stores, wave-level reduction of the partials.
    auto epilogue_regs = [&](float (&st)[16], int nr) {  // st: [kind][nr][q4]; one 32-channel half per call
        unsigned char* os = Hl + wg * 8192;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                for (int mr = 0; mr < 2; mr++) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        v[j] = acc[mr][nr][4 * q4 + j];
                        st[nr * 4 + q4] += v[j];
                        st[8 + nr * 4 + q4] += v[j] * v[j];
                    }
                    const int px = mr * 32 + r;
                    *reinterpret_cast<bf16x4*>(os + px * 128 + (((nr * 4 + q4) ^ (r & 7)) << 4) + hf * 8) =
                        bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                }
    };
    auto epilogue_read = [&](bf16x8 (&od)[8]) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned char* os = Hl + wg * 8192;
        // vector i of lane l = pixel 8i + (l >> 3) of the wave's 64 (tile row 4wg + (i >> 1)), channels 8(l & 7)..
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int px = i * 8 + lq;
            od[i] = *reinterpret_cast<const bf16x8*>(os + px * 128 + ((sv ^ (px & 7)) << 4));
        }
    };
    auto epilogue_store = [&](int t, const bf16x8 (&od)[8], int i0, int i1) {
        int b, tl;
        decode(t, b, tl);
        const int ty0 = (tl / p.tiles_x) * 16, tx0 = (tl % p.tiles_x) * 16;
        bf16* outw = p.out + (((size_t)b * p.H + ty0 + wg * 4) * p.W + tx0) * 64;  // scalar
        const unsigned lo = (unsigned)(lq * 64 + sv * 8);
#pragma unroll
#ifdef WS_ASSERT_ADDR
        for (int i = i0; i < i1; i++)
            *reinterpret_cast<bf16x8*>(p.out + WS_CHECK(2, (outw - p.out) + (long long)(((i >> 1) * p.W + (i & 1) * 8) * 64) + lo, p.out_elems - 8, t)) = od[i];
#else
        for (int i = i0; i < i1; i++) *reinterpret_cast<bf16x8*>(outw + (size_t)(((i >> 1) * p.W + (i & 1) * 8) * 64) + lo) = od[i];
#endif
    };
    auto epilogue_stats = [&](int t, const float (&st)[16]) {
        int b, tl;
        decode(t, b, tl);
        {
        if (p.stats) {
            // 16 per-lane partials -> wave totals with a halving butterfly: after the steps on lane bits 5,4,3,2 lane l
            // holds value j = 8*b5 + 4*b4 + 2*b3 + b2 summed over its 4-lane class; two plain steps finish it.
            float a8[8], a4[4], a2[2];
            {
                const bool up = (lane >> 5) & 1;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float keep = up ? st[8 + j] : st[j], send = up ? st[j] : st[8 + j];
                    a8[j] = keep + __shfl_xor(send, 32, 64);
                }
            }
            {
                const bool up = (lane >> 4) & 1;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float keep = up ? a8[4 + j] : a8[j], send = up ? a8[j] : a8[4 + j];
                    a4[j] = keep + __shfl_xor(send, 16, 64);
                }
            }
            {
                const bool up = (lane >> 3) & 1;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float keep = up ? a4[2 + j] : a4[j], send = up ? a4[j] : a4[2 + j];
                    a2[j] = keep + __shfl_xor(send, 8, 64);
                }
            }
            float a1;
            {
                const bool up = (lane >> 2) & 1;
                const float keep = up ? a2[1] : a2[0], send = up ? a2[0] : a2[1];
                a1 = keep + __shfl_xor(send, 4, 64);
            }
            a1 += __shfl_xor(a1, 2, 64);
            a1 += __shfl_xor(a1, 1, 64);
            if ((lane & 3) == 0) {
                const int j = lane >> 2;  // = 8*kind + 4*nr + q4
                const int kind = j >> 3, g = j & 7;
#ifdef WS_ASSERT_ADDR
                p.stats[WS_CHECK(3, (((long long)b * 8 + g) * (tiles_img * 4) + tl * 4 + wg) * 2 + kind, p.stats_elems - 1, t)] = a1;
#else
                p.stats[(((size_t)b * 8 + g) * (tiles_img * 4) + tl * 4 + wg) * 2 + kind] = a1;
#endif
            }
        }
        }
    };

#ifdef WS_TRACE
    const unsigned long long ws_t_entry = __builtin_amdgcn_s_memtime();
    const unsigned long long ws_rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    if (p.aff && tile_of(0) < p.ntiles && tg < 128) aff_l[tg] = p.aff[(size_t)(tile_of(0) / tiles_img) * 128 + tg] * SILU_PRE;
    __syncthreads();  // weights, bias and the first affine are in LDS
    for (int ph = -1; ph <= 2 * K; ph++) {
        const int q = ph - grp;
#ifdef WS_TRACE
        unsigned long long ws_t[6] = {0, 0, 0, 0, 0, 0};
#endif
        WS_STAMP(0)
        if (q & 1) {
            // stage role.  Every step runs unconditionally on block-local arrays (a conditional definition would make the
            // staging registers live across the whole phase loop, i.e. through the other role as well); only the global
            // stores are predicated.
            __builtin_amdgcn_s_setprio(WS_STAGE_PRIO);
            const int k = (q - 1) >> 1;
            const bool fin = k >= 0 && tile_of(k) < p.ntiles;
            const int tn = min(tile_of(k + 1 < K ? k + 1 : K - 1), p.ntiles - 1);
            // (two copies of the body, with and without the global stores: a branch around the stores alone would make
            // the compiler count the prefetch waits as if no store had been issued, and those waits would drain them)
            auto stage_body = [&](auto finc) {
                constexpr bool fin = decltype(finc)::value;
                const int tk = min(tile_of(max(k, 0)), p.ntiles - 1);
                // The HBM pipe of a CU moves ~1 KB per ~270 cycles per wave here, and a VMEM instruction that finds the
                // queue full blocks the wave: loads and stores are therefore dealt out between the VALU/LDS pieces of the
                // epilogue instead of being issued in two bursts.  Loads first (their data is needed last); the stores
                // are issued before the prefetched halo is consumed, so the counted waits of lstore() leave them in flight.
                bf16x8 pre[NVEC], od[8];
                unsigned off[NVEC], hok;
                bool interior;
                const bf16* base;
                float st[16];
#pragma unroll
                for (int j = 0; j < 16; j++) st[j] = 0.f;
                gaddr(tn, off, base, hok, interior);
                __builtin_amdgcn_sched_barrier(0);
#ifndef WS_NO_GLOAD
#ifdef WS_ASSERT_ADDR
#define WS_LOADS(a, b) _Pragma("unroll") for (int i = a; i < b; i++) pre[i] = *reinterpret_cast<const bf16x8*>(p.in + WS_CHECK(1, (base - p.in) + (long long)off[i], p.in_elems - 8, tn));
#else
#define WS_LOADS(a, b) _Pragma("unroll") for (int i = a; i < b; i++) pre[i] = *reinterpret_cast<const bf16x8*>(base + off[i]);
#endif
#else
#define WS_LOADS(a, b) _Pragma("unroll") for (int i = a; i < b; i++) pre[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#endif
                WS_LOADS(0, 4)
                __builtin_amdgcn_sched_barrier(0);
                WS_STAMP(3)
#ifndef WS_NO_E1
                epilogue_regs(st, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                WS_LOADS(4, 8)
                __builtin_amdgcn_sched_barrier(0);
#ifndef WS_NO_E1
                epilogue_regs(st, 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
                WS_LOADS(8, NVEC)
                __builtin_amdgcn_sched_barrier(0);
                WS_STAMP(4)
                epilogue_read(od);
                __builtin_amdgcn_sched_barrier(0);
                // The output stores and the prologue + LDS store of the prefetched halo are interleaved, one 16-byte vector
                // each: a store that finds the CU's memory queue full blocks the wave for ~270 cycles, which is exactly the
                // VALU time of one vector's GroupNorm-apply + SiLU - issued back to back (stores, then the prologue) the two
                // costs add up; alternated, they overlap.  od[] is complete in registers before the first vector
                // overwrites the wave's slice.
                Aff A;
                lstore_setup(A);
                __builtin_amdgcn_sched_barrier(0);
#ifndef WS_NO_LSTORE
#define WS_LS(i) lstore_vec(A, pre[i], i, loff[i], hok, interior); __builtin_amdgcn_sched_barrier(0);
#else
#define WS_LS(i)
#endif
#ifdef WS_NO_E2
                if constexpr (false) {
#else
                if constexpr (fin) {
#endif
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        epilogue_store(tk, od, i, i + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        WS_LS(i)
                    }
                    epilogue_stats(tk, st);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 4; i < 8; i++) {
                        epilogue_store(tk, od, i, i + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        WS_LS(i)
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) { WS_LS(i) }
                }
                WS_STAMP(5)
#pragma unroll
                for (int i = 8; i < NVEC; i++) { WS_LS(i) }
#undef WS_LS
#undef WS_LOADS
            };
#ifndef WS_NO_STAGE
            if (fin) stage_body(std::true_type{});
            else stage_body(std::false_type{});
#endif
        } else {
            // compute role (unconditional: a group without a tile multiplies stale LDS contents and nobody stores the result);
            // the GroupNorm affine of the tile staged NEXT phase travels global -> register -> LDS around it
            __builtin_amdgcn_s_setprio(0);
            const int k = q >> 1;
            const bool nxt = p.aff && k >= 0 && k + 1 < K && tile_of(k + 1) < p.ntiles && tg < 128;
            float av = 0.f;
            if (nxt) av = p.aff[(size_t)(tile_of(k + 1) / tiles_img) * 128 + tg] * SILU_PRE;
#ifndef WS_NO_COMPUTE
            compute();
#endif
            if (nxt) aff_l[tg] = av;
        }
        WS_STAMP(1)
        // phase barrier: LDS traffic only.  __syncthreads() would also drain vmcnt (the epilogue's global stores), putting
        // an HBM write round trip on every phase; nothing in this kernel reads what it stored.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        WS_STAMP(2)
        WS_FLUSH()
    }
#ifdef WS_TRACE
    if (tid == 0) {
        ws_blk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - ws_t_entry;
        ws_blk[blockIdx.x * 2 + 1] = ws_rt_entry;
    }
#endif
}

}  // namespace

// eligibility: bf16, one 64-channel input, 64 outputs, 8 GroupNorm groups, 16-aligned output size
bool conv3x3_ws64_eligible(int H, int W, int C0, int C1, int Cout, int groups, int dtype) {
    return dtype == DIG_BF16 && C0 == 64 && C1 == 0 && Cout == 64 && groups == 8 && H % 16 == 0 && W % 16 == 0;
}

int conv3x3_ws64_launch(const dig_conv3x3_args* a, hipStream_t stream) {
    WsParams p;
    p.in = (const bf16*)a->in0;
    p.aff = a->aff0;
    p.w = (const bf16*)a->w;
    p.bias = a->bias;
    p.out = (bf16*)a->out;
    p.stats = a->stats;
    p.bstride = a->in0_bstride > 0 ? a->in0_bstride : (long long)(a->up2 ? (a->H / 2) * (a->W / 2) : a->H * a->W) * 64;
    p.B = a->B; p.H = a->H; p.W = a->W; p.up2 = a->up2;
    p.tiles_x = a->W / 16;
    p.tiles_y = a->H / 16;
    p.ntiles = a->B * p.tiles_x * p.tiles_y;
    static std::atomic<int> ncu[64];  // zero-initialised; racing first calls both query and store the same value
    int device = 0;
    (void)hipGetDevice(&device);
    const int dev = device & 63;
    int ncus = ncu[dev].load(std::memory_order_acquire);
    if (!ncus) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_ws64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
        DIG_CHECK(e == hipSuccess, DIG_ELAUNCH, "conv3x3(ws64): cannot raise dynamic LDS to %d: %s", WS_LDS, hipGetErrorString(e));
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
        DIG_CHECK(e == hipSuccess && n > 0, DIG_ELAUNCH, "conv3x3(ws64): cannot query the CU count");
        ncu[dev].store(n, std::memory_order_release);
        ncus = n;
    }
    int grid = (p.ntiles + 1) / 2;
    if (grid > ncus) grid = ncus;
    if (const char* g = getenv("DIG_WS_GRID")) {  // test hook: a small grid makes every block walk many tiles
        const int v = atoi(g);
        if (v > 0 && v < grid) grid = v;
    }
#ifdef WS_ASSERT_ADDR
    const long long src_px = (long long)(a->up2 ? (a->H / 2) * (a->W / 2) : a->H * a->W);
    p.in_elems = p.bstride * (a->B - 1) + src_px * 64;
    p.out_elems = (long long)a->B * a->H * a->W * 64;
    p.stats_elems = (long long)a->B * 8 * p.tiles_x * p.tiles_y * 4 * 2;
#endif
    hipLaunchKernelGGL(conv3x3_ws64_kernel, dim3(grid), dim3(512), WS_LDS, stream, p);
    DIG_LAUNCH_CHECK("conv3x3(ws64)");
    return DIG_OK;
}

#ifdef WS_ASSERT_ADDR
extern "C" int dig_debug_ws_addr_violations(long long* host4) {
    return (int)hipMemcpyFromSymbol(host4, HIP_SYMBOL(ws_addr_viol), sizeof(long long) * 4);
}
#endif

#ifdef WS_TRACE
extern "C" int dig_debug_ws_trace(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ws_trace), sizeof(unsigned long long) * 8 * 64 * 8);
}
extern "C" int dig_debug_ws_blk(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ws_blk), sizeof(unsigned long long) * 256 * 2);
}
#endif
