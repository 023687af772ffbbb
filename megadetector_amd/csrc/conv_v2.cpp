// Implicit-GEMM convolution, second generation main loop (gfx950 / MI355X).
//
// Same GEMM view, data layout, LDS image and epilogue contract as conv_igemm.cpp (read its header
// first); what changes is the schedule, driven by two measurements on MI355X (profiles/r1a_*):
//   * removing the LDS fragment reads from the v1 loop gains 4 %, removing the HBM/L2 -> LDS loads
//     gains 29 %: the matrix pipe idles while a wave *issues* its buffer_load...lds burst
//     (~60 cycles of issue per 1 KiB piece) and its address arithmetic in front of the MFMAs;
//   * 128-row tiles leave 6.25 tiles per workgroup slot on the 80x80 / 160x160 feature maps
//     (7 rounds for 6.25 rounds of work); 160-row tiles divide them exactly.
// So here every wave's instruction stream is a uniform mix: the step over one 64-deep K slab is
// split at the workgroup barrier into two halves of FM*FN MFMAs each, fragments are double-buffered
// in registers, and the LDS-DMA pieces of slab s+2, the fragment reads of slab s+1 and the MFMAs of
// slab s are interleaved instruction by instruction.  Two independent workgroups share a CU, so the
// partner wave on a SIMD always has matrix work ready while this wave issues memory operations.
//
//   step s (cur = s & 1):                                    LDS stage cur holds slab s
//     read  Y  <- stage cur, k 32..63                         } interleaved
//     mfma  X  (k 0..31 of slab s)                            }
//     s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier                 slab s+1 landed everywhere; stage cur free
//     DMA   slab s+2 -> stage cur                             }
//     read  X' <- stage cur^1, k 0..31 of slab s+1            } interleaved
//     mfma  Y  (k 32..63 of slab s)                           }
//     advance the loader (branch-free inside a tile) ; epilogue when the tile's last slab is done
//
// Bias comes in through the scalar cache in the epilogue (no LDS, no vector registers), so a
// workgroup's LDS is exactly the two stages: 160x160 tiles use 2 x 80 KiB = the whole CU.

#include <algorithm>
#include <type_traits>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) char lds_char;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;        // >= every descriptor's num_records: the lane reads zeros
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;

__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

constexpr int v2_lds_bytes(int bm, int bn) { return 2 * (bm + bn) * 128; }
constexpr int v2_ring_lds_bytes(int bm, int bn) { return (3 * bm + 2 * bn) * 128; }
constexpr int v2_blocks_per_cu(int bm, int bn, int nw) {
    int b = 163840 / v2_lds_bytes(bm, bn);
    if (b > 32 / nw) b = 32 / nw;
    if (b > 2) b = 2;
    return b < 1 ? 1 : b;
}
constexpr int v2_waves_per_simd(int bm, int bn, int nw) {
    int w = v2_blocks_per_cu(bm, bn, nw) * nw / 4;
    return w < 1 ? 1 : w;
}

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// PROF = 1: timing-only instrumentation (s_memtime at the phase boundaries of every step, summed per
// wave and written to p.dbg); used by tools/convbench.cpp, never by the product path
// UP = the variant that reads the first K slabs of a 1x1 conv from a low-resolution tensor (nearest-neighbour upsample in
// place, ConvArgs::in_up); its own instantiation so that the loader of every other launch stays as it was
// PW = the instantiations for 1x1 / stride 1 / unpadded convs (tile set-up without divisions, see init_tile): 1 = the channel
// count is a multiple of 64, 2 = it is not (the last K slab's chunks past the last channel are masked, see advance)
// RA = activation stages (pointwise instantiations only): 2 = one ring of (activation + weight) stages; 3 = the activation tile in
// a ring of three beside the weight tile's ring of two -- a short-K pointwise layer is bound by one memory round trip per K step
// (a two-stage ring prefetches ONE step ahead), its weights come from L2 in a third of the time its activations take from the
// memory side, so the LDS a third weight stage would need buys more as activation lead (DESIGN.md section 5 [r4])
// DEC = 1: the instantiation for the Detect 1x1 convs that decode in their epilogue (ConvArgs::dec_pred; conv_igemm.cpp on why it is one)
template <int BM, int BN, int WM, int WN, int PROF = 0, bool UP = false, int PW = 0, int RA = 2, int DEC = 0>
__global__ void __launch_bounds__(WM * WN * 64, v2_waves_per_simd(BM, BN, WM * WN))
conv_v2_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int B_INSTR = BN / 8;                       // 1 KiB pieces (8 rows x 128 B) of the weight tile
    constexpr int A_PER = BM / 8 / NW, B_PER = (B_INSTR + NW - 1) / NW;
    constexpr bool B_RAGGED = (B_INSTR % NW) != 0;        // the last piece exists only on the first waves
    static_assert((BM / 8) % NW == 0, "the activation tile must split evenly over the waves");
    static_assert(RA == 2 || (RA == 3 && PW != 0 && !UP), "the three-stage activation ring is a pointwise instantiation");
    // byte offsets of activation stage sa / weight stage sb
    auto a_stage = [](int sa) __attribute__((always_inline)) { return RA == 2 ? sa * STAGE : sa * A_BYTES; };
    auto b_stage = [](int sb) __attribute__((always_inline)) { return RA == 2 ? sb * STAGE + A_BYTES : RA * A_BYTES + sb * B_BYTES; };
    static_assert(TM % 16 == 0 && TN % 16 == 0 && TM % 8 == 0, "wave tile must be a multiple of 16x16");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- persistent streams (see conv_igemm.cpp): block b runs on XCD b % 8 ---------------------
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_n = slot % p.tiles_n;
    const int ms = slot / p.tiles_n;
    const int xcd_first = xcd * p.tiles_per_xcd;
    const int xcd_tiles = min(p.tiles_per_xcd, p.tiles_m - xcd_first);
    const int my_tiles = (xcd_tiles > ms) ? (xcd_tiles - ms + p.m_streams - 1) / p.m_streams : 0;
    if (my_tiles <= 0) return;
    const int first_tile = xcd_first + ms;
    const int tile_step = p.m_streams;
    const int last_tile = first_tile + (my_tiles - 1) * tile_step;
    const int n0 = tile_n * BN;
    const int KT = p.k_pad >> 6;
    const int total_steps = my_tiles * KT;

    // ---- weight side ----------------------------------------------------------------------------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;        // swizzled source chunk inside the 128-byte K slab
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt + (size_t)n0 * p.k_pad), 0, kNumRecords, 0x00020000);
    unsigned b_off[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int row = (i * NW + wave) * 8 + lr;
        b_off[i] = (n0 + row < p.n_rows) ? (unsigned)(row * p.k_pad + jj * 8) * 2u : kOOB;
    }

    // ---- activation side: loader state ----------------------------------------------------------
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    unsigned a_off[A_PER];
    uint32_t a_mask[A_PER];
    // nearest-neighbour upsample read in place (1x1 convs only): the slabs [0, up_slabs) of K come from the low-resolution
    // producer of the concatenated input's first part, at this lane's pixel halved
    const int up_slabs = UP ? p.up_slabs : 0;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(UP ? p.in_up : p.in), 0, kNumRecords, 0x00020000);
    [[maybe_unused]] unsigned u_off[UP ? A_PER : 1];
    // position of this lane's 16-byte chunk inside K (valid for C8 >= 8: at most one tap wrap per slab)
    int c8 = 0, ts = 0;
    uint32_t tapbit = 1;
    unsigned tapoff = 0;
    int l_kt = 0, l_tile = first_tile;
    bool l_live = true;                     // false once the stream has no more slabs to load
    [[maybe_unused]] int lb_kt = 0;         // (RA == 3) the weight loader's slab: it runs one step behind the activation loader
    const int kh = p.ntaps / p.kw;
    const unsigned wrap_c = (unsigned)(p.ld_in * 2 - p.C8 * 16);        // next tap, same row
    const unsigned wrap_r = (unsigned)((p.W - p.kw) * p.ld_in * 2);     // first tap of the next kernel row

    // 1x1 / stride 1 / no padding (most launches of this kernel, with 3 .. 10 slabs per tile): output pixel m IS input
    // pixel m, so a lane's offset from the tile's first pixel never changes and the only mask bit is "m < M".  The
    // general path below costs two integer divisions and a 9-tap mask per row -- ~5 000 cycles per tile and wave, a
    // fifth of the whole 1x1 launch at K = 160 .. 320 (profiles/r3_convbench_v2_ablation.txt: "advance").
    // (its own instantiation -- conv2_launch picks it -- so that the 3x3 / strided launches keep the code they had)
    auto init_tile = [&](int tile_m) __attribute__((always_inline)) {
        if constexpr (PW) {
            // the descriptor ends with the tensor: rows past the last pixel read zeros through the range check, no mask
            // test per piece; a lane's offset (row of the tile, 16-byte chunk) never changes, the slab goes into the
            // scalar offset -- a piece is issued with no VALU instruction at all (every instruction between two MFMA
            // chunks is matrix-pipe idle time: the waves of a SIMD run the same code between the same barriers)
            const int m0 = tile_m * BM;
            const long long left = ((long long)p.M - m0) * p.ld_in * 2;
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (long long)m0 * p.ld_in), 0,
                                                       (int)(left > 0x7fffffffLL ? 0x7fffffffLL : left), 0x00020000);
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int row = (i * NW + wave) * 8 + lr;
                a_off[i] = (unsigned)(row * p.ld_in * 2 + jj * 16);
                a_mask[i] = 1u;
            }
            c8 = jj; ts = 0; tapbit = 1; tapoff = 0;
            return;
        }
        const int m0 = tile_m * BM;
        const int b0 = conv_udiv(m0, p.HoWo, p.rcp_howo);
        const int rem0 = m0 - b0 * p.HoWo;
        const int oy0 = conv_udiv(rem0, p.Wo, p.rcp_wo);
        const int ox0 = rem0 - oy0 * p.Wo;
        const long long base_px = (long long)(b0 * p.H + oy0 * p.stride - p.pad) * p.W + (ox0 * p.stride - p.pad);
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + base_px * p.ld_in), 0, kNumRecords, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = (i * NW + wave) * 8 + lr;
            const int m = m0 + row;
            uint32_t mask = 0;
            unsigned off = 0;
            if (m < p.M) {
                const int b = conv_udiv(m, p.HoWo, p.rcp_howo);
                const int rem = m - b * p.HoWo;
                const int oy = conv_udiv(rem, p.Wo, p.rcp_wo);
                const int ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad;
                const int ix0 = ox * p.stride - p.pad;
                const long long px = (long long)(b * p.H + iy0) * p.W + ix0;
                off = (unsigned)((px - base_px) * p.ld_in * 2);
                if constexpr (UP)
                    u_off[i] = (unsigned)((((long long)b * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * p.ld_up * 2 + jj * 16);
                // tap (r, s) is inside the image iff row r and column s are: three row bits x three column bits
                // (kernels are 1x1 or 3x3, bit r * kw + s)
                uint32_t cols = 0, rows = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    cols |= (t < p.kw && (unsigned)(ix0 + t) < (unsigned)p.W) ? 1u << t : 0u;
                    rows |= (t < kh && (unsigned)(iy0 + t) < (unsigned)p.H) ? 1u << t : 0u;
                }
                mask = ((rows & 1u) ? cols : 0u) | ((rows & 2u) ? cols << p.kw : 0u) | ((rows & 4u) ? cols << (2 * p.kw) : 0u);
            }
            a_off[i] = off;
            a_mask[i] = mask;
        }
        c8 = jj; ts = 0; tapbit = 1; tapoff = (unsigned)jj * 16u;
    };

    // (PW) this lane's 16-byte chunk of the last K slab lies past the last input channel
    [[maybe_unused]] const bool pw_tail_bad = (p.C8 & 7) != 0 && jj >= (p.C8 & 7);
    // the loader moves on by one slab (branch-free inside a tile)
    auto advance = [&]() __attribute__((always_inline)) {
        if (++l_kt == KT) {
            l_kt = 0;
            if (l_tile == last_tile) {
                l_live = false;
#pragma unroll
                for (int i = 0; i < A_PER; ++i) a_mask[i] = 0;
            } else {
                l_tile += tile_step;
                init_tile(l_tile);
            }
        } else {
            if constexpr (PW) {
                // the tensor's last K slab when C_in is no multiple of 64 (C8 % 8 != 0: 80, 160, 480 channels): its
                // chunks past the last channel would read the NEXT pixel's first channels (or, in a channel slice of a
                // wider buffer, the neighbouring tensor's) against zero weights -- 0 * Inf / NaN would poison every output
                // channel of the pixel.  Those lanes read zeros through the range check instead; init_tile restores
                // the offsets with the next tile.
                // (the scalar test first: layers whose channel count is a multiple of 64 -- most -- branch over the block;
                // as one per-lane condition it cost a dozen VALU instructions in every step of every layer)
                // (its own instantiation: as a run-time test it cost a dozen VALU instructions in every step of every layer,
                // whichever way it was written -- the compiler folds the scalar and the per-lane condition into one exec mask)
                if constexpr (PW == 2) {
                    if (l_kt == KT - 1 && pw_tail_bad) {
#pragma unroll
                        for (int i = 0; i < A_PER; ++i) a_off[i] = kOOB;
                    }
                }
            }
            c8 += 8;
            tapoff += 128u;
            const bool w = c8 >= p.C8;
            c8 = w ? c8 - p.C8 : c8;
            tapoff += w ? wrap_c : 0u;
            tapbit = w ? tapbit << 1 : tapbit;
            ts += w ? 1 : 0;
            const bool w2 = ts == p.kw;
            ts = w2 ? 0 : ts;
            tapoff += w2 ? wrap_r : 0u;
        }
    };

    // one LDS-DMA piece of the loader's current slab into stage `buf`
    auto dma_a = [&](int buf, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if constexpr (UP) {
            if (l_kt < up_slabs) {                                               // wave-uniform
                MDHIP_DMA16(u_rsrc, smem + buf * STAGE + (i * NW + wave) * 1024, (a_mask[i] & tapbit) ? u_off[i] : kOOB, l_kt * 128);
                return;
            }
        }
        if constexpr (PW) {
            MDHIP_DMA16(a_rsrc, smem + a_stage(buf) + (i * NW + wave) * 1024, a_off[i], l_kt * 128);
        } else {
            const unsigned voff = (a_mask[i] & tapbit) ? a_off[i] + tapoff : kOOB;
            MDHIP_DMA16(a_rsrc, smem + buf * STAGE + (i * NW + wave) * 1024, voff, 0);
        }
    };
    auto dma_b = [&](int buf, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if (B_RAGGED && i == B_PER - 1 && wave >= B_INSTR % NW) return;     // wave-uniform
        // (PW: after the stream's last slab the loader re-reads slabs of the last tile into stages nobody reads)
        const unsigned voff = (PW || l_live) ? b_off[i] : kOOB;
        MDHIP_DMA16(b_rsrc, smem + b_stage(buf) + (i * NW + wave) * 1024, voff, (RA == 3 ? lb_kt : l_kt) * 128);
    };

    // ---- fragment reads ---------------------------------------------------------------------------
    const int frag_row_off = (lane & 15) * 128;
    const int frag_ch0 = (((lane >> 4) ^ (lane & 7)) * 16);      // k 0..31 ; k 32..63 is ^ 64
    const int a_frag_base = (wm * TM) * 128 + frag_row_off;
    const int b_frag_base = (wn * TN) * 128 + frag_row_off;
    auto read_x = [&](int buf, int kk, int i) -> frag8_t {
        if constexpr ((PROF & 32) != 0) return frag_dummy(lane + i);
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + a_stage(buf) + a_frag_base + i * 2048 +
                                                                 (frag_ch0 ^ (kk * 64)));
    };
    auto read_w = [&](int buf, int kk, int j) -> frag8_t {
        if constexpr ((PROF & 32) != 0) return frag_dummy(lane + j);
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + b_stage(buf) + b_frag_base + j * 2048 +
                                                                 (frag_ch0 ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue: lane holds channels n..n+3 of pixel m ----------------------------------------
    // Specialised on (residual, fp32 output) so that the common path has no wait between stores: a
    // vmcnt wait in front of every store would serialise them on the memory round trip (measured:
    // 36k cycles per 160x160 tile, as long as the tile's MFMAs).  Residual rows are fetched one
    // fragment column ahead of their use, so waiting for them never waits for a store.
    const int q4 = lane >> 4;
    auto epilogue_t = [&](int tile_m, auto has_res_t, auto out_f32_t, auto out_f8_t) {
        // (x * r and the residual add stay two roundings whatever the shape of the code between them)
#pragma clang fp contract(off)
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        constexpr bool OUT_F32 = decltype(out_f32_t)::value;
        constexpr bool OUT_F8 = decltype(out_f8_t)::value;
        const int m0 = tile_m * BM + wm * TM + (lane & 15);
        const int nbase = n0 + wn * TN + q4 * 4;
        // 16-bit outputs and the residual go through buffer instructions (see conv_v5.cpp): one 32-bit offset per lane
        // and tensor, tile-relative descriptors, rows past the end dropped / read as zeros by the range check -- no
        // 64-bit address arithmetic and no branch around every store.  The offsets come from an opaque copy of the lane
        // id so that they are computed here and not kept alive across the main loop.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        float neg_log2e = kNegLog2e;                             // (opaque: not a register pair kept across the main loop)
        asm volatile("" : "+v"(neg_log2e));
        const int ml = wm * TM + (lane_e & 15);
        const int qe = lane_e >> 4;
        const int npair0 = n0 + wn * TN + qe * 8;                 // first of this lane's 8 channels of column pair 0
        const int nlast = n0 + wn * TN + (FN - 1) * 16 + qe * 4;  // its 4 channels of an odd last column
        const long long rows_left = (long long)p.M - (long long)tile_m * BM;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((uint16_t*)p.out + (size_t)tile_m * BM * p.ld_out), 0,
            (OUT_F32 || OUT_F8) ? 0 : (int)min(rows_left * p.ld_out * 2, 0x7fffffffLL), 0x00020000);
        const unsigned o_pair = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)npair0) * 2u;
        const unsigned o_last = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)nlast) * 2u;
        const unsigned o_step = 16u * (unsigned)p.ld_out * 2u;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.res + (HAS_RES ? (size_t)tile_m * BM * p.ld_res : 0)), 0,
            HAS_RES ? (int)min(rows_left * p.ld_res * 2, 0x7fffffffLL) : 0, 0x00020000);
        const unsigned r_col = ((unsigned)ml * (unsigned)p.ld_res + (unsigned)(n0 + wn * TN + qe * 4)) * 2u;
        const unsigned r_step = 16u * (unsigned)p.ld_res * 2u;
        // bias of all fragment columns first (scalar loads), then pixel-row by pixel-row so that the
        // stores that complete one cache line are issued back to back
        // bias of all fragment columns first (scalar loads), then pixel-row by pixel-row so that the
        // stores that complete one cache line are issued back to back
        // (round 3, measured and not kept: the five columns as vector loads issued together -- the compiler's vmcnt wait
        // for them also waits for the LDS-DMA pieces of the next tile that are in flight: epilogue + 25 .. 45 %)
        float bv[FN][4];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int nb = n0 + wn * TN + j * 16;                // wave-uniform: bias comes through s_load
            bv[j][0] = bv[j][1] = bv[j][2] = bv[j][3] = 0.f;
            if (nb < p.n_rows) {
                f32x16 b16;
                const unsigned long long ba = (unsigned long long)(p.bias + nb);
                const unsigned long long bs =
                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ba >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)ba);
                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b16) : "s"(bs) : "memory");
                const f32x4 g0 = {b16[0], b16[1], b16[2], b16[3]}, g1 = {b16[4], b16[5], b16[6], b16[7]},
                            g2 = {b16[8], b16[9], b16[10], b16[11]}, g3 = {b16[12], b16[13], b16[14], b16[15]};
                const f32x4 g = q4 == 0 ? g0 : (q4 == 1 ? g1 : (q4 == 2 ? g2 : g3));
                bv[j][0] = g[0]; bv[j][1] = g[1]; bv[j][2] = g[2]; bv[j][3] = g[3];
            }
        }
        uint2 rrow[2][FN];
        auto fetch_res_row = [&](int i, uint2 (&r)[FN]) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, (int)(r_col + (unsigned)i * r_step + (unsigned)(j * 32)), 0, 0);
                r[j] = make_uint2(t[0], t[1]);
            }
        };
        if constexpr (HAS_RES) fetch_res_row(0, rrow[0]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (HAS_RES) {
                if (i + 1 < FM) fetch_res_row(i + 1, rrow[(i + 1) & 1]);
            }
            const int m = m0 + i * 16;
            float v[FN][4];
            // bias, then the activation of the whole pixel row under ONE uniform branch (no select per value), two values
            // per instruction wherever the instruction set has a packed form
#pragma unroll
            for (int j = 0; j < FN; ++j) {
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const mdhip_f32x2 t = mdhip_f32x2{acc[i][j][r], acc[i][j][r + 1]} + mdhip_f32x2{bv[j][r], bv[j][r + 1]};
                    v[j][r] = t[0];
                    v[j][r + 1] = t[1];
                }
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if ((PROF & 4) == 0 && p.act) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const mdhip_f32x2 t = silu_f32x2(mdhip_f32x2{v[j][r], v[j][r + 1]}, neg_log2e);
                        v[j][r] = t[0];
                        v[j][r + 1] = t[1];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (HAS_RES) {
                    const uint2 rv = rrow[i & 1][j];
                    v[j][0] += st_unpack((uint16_t)(rv.x & 0xffff));
                    v[j][1] += st_unpack((uint16_t)(rv.x >> 16));
                    v[j][2] += st_unpack((uint16_t)(rv.y & 0xffff));
                    v[j][3] += st_unpack((uint16_t)(rv.y >> 16));
                }
            }
            if constexpr ((PROF & 2) != 0) {
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(v[j][0]), "v"(v[j][1]), "v"(v[j][2]), "v"(v[j][3]));
            } else if constexpr (OUT_F32 && DEC != 0) {            // Detect decode in place (ConvArgs::dec_pred)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = nbase + j * 16;
                    if (m < p.M && n < p.N) mdhip_decode_store(p, m, n, v[j]);
                }
            } else if constexpr (OUT_F32) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = nbase + j * 16;
                    if (m < p.M && n < p.N)
                        *(float4*)((float*)p.out + (size_t)m * p.ld_out + n) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                }
            } else if constexpr (OUT_F8) {
                // e4m3 output (MDHIP_DTYPE_FP8: the hidden tensor of a bottleneck): 4 channels = 4 bytes per lane and
                // fragment; the same exchange as the 16-bit path leaves 8 consecutive channels = 8 bytes per lane
                const __amdgpu_buffer_rsrc_t o8_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((uint8_t*)p.out + (size_t)tile_m * BM * p.ld_out), 0, (int)min(rows_left * p.ld_out, 0x7fffffffLL), 0x00020000);
                const unsigned row8 = (unsigned)(ml + i * 16) * (unsigned)p.ld_out;
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    const unsigned a0 = pack_e4m3x4(v[j][0], v[j][1], v[j][2], v[j][3], p.out_qscale);
                    const unsigned b0 = pack_e4m3x4(v[j + 1][0], v[j + 1][1], v[j + 1][2], v[j + 1][3], p.out_qscale);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    const int n = npair0 + j * 16;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{t0[0], t0[1]}, o8_rsrc, (int)(n < p.N ? row8 + (unsigned)n : kOOB), 0, 0);
                }
                if (FN & 1) {
                    const int j = FN - 1;
                    __builtin_amdgcn_raw_buffer_store_b32(pack_e4m3x4(v[j][0], v[j][1], v[j][2], v[j][3], p.out_qscale), o8_rsrc,
                                                          (int)(nlast < p.N ? row8 + (unsigned)nlast : kOOB), 0, 0);
                }
            } else {
                // bf16: pairs of fragment columns are exchanged across the four 16-lane rows
                // (v_permlane32_swap, v_permlane16_swap) so that a lane holds 8 consecutive channels
                // = 16 bytes, and one store covers 64 contiguous bytes per pixel instead of 32
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                    unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                    auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                    // row q of the wave now holds channels q*8 .. q*8+7 of the 32 channels of this pair
                    const unsigned off = o_pair + (unsigned)i * o_step + (unsigned)(j * 32);
                    // (round 6, measured and not kept: non-temporal stores +- 0 .. - 19 %, write-through stores - 17 .. - 22 %,
                    // profiles/r6_pointwise_experiments.txt)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{t0[0], t1[0], t0[1], t1[1]}, o_rsrc,
                                                           (int)(npair0 + j * 16 < p.N ? off : kOOB), 0, 0);
                }
                if (FN & 1) {
                    const int j = FN - 1;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{st_pack2(v[j][0], v[j][1]), st_pack2(v[j][2], v[j][3])}, o_rsrc,
                                                          (int)(nlast < p.N ? o_last + (unsigned)i * o_step : kOOB), 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int tile_m) {
        if constexpr ((PROF & 8) != 0) __builtin_amdgcn_s_setprio(3);
        if constexpr (DEC != 0) { epilogue_t(tile_m, std::false_type{}, std::true_type{}, std::false_type{}); return; }
        if (p.out_f32) epilogue_t(tile_m, std::false_type{}, std::true_type{}, std::false_type{});
        else if (p.out_f8) epilogue_t(tile_m, std::false_type{}, std::false_type{}, std::true_type{});
        else if (p.res) epilogue_t(tile_m, std::true_type{}, std::false_type{}, std::false_type{});
        else epilogue_t(tile_m, std::false_type{}, std::false_type{}, std::false_type{});
        if constexpr ((PROF & 8) != 0) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: slabs 0 and 1 in flight, fragments X of slab 0 in registers ---------------------
    init_tile(first_tile);
    auto lb_advance = [&]() __attribute__((always_inline)) { lb_kt = (lb_kt + 1 == KT) ? 0 : lb_kt + 1; };
    if constexpr (RA == 3) {
        // activation slabs 0, 1, 2 and weight slabs 0, 1
#pragma unroll
        for (int st = 0; st < 3; ++st) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) dma_a(st, i);
            advance();
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) dma_b(st, i);
            lb_advance();
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) dma_a(0, i);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) dma_b(0, i);
        advance();
#pragma unroll
        for (int i = 0; i < A_PER; ++i) dma_a(1, i);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) dma_b(1, i);
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) xa[i] = read_x(0, 0, i);
#pragma unroll
    for (int j = 0; j < FN; ++j) wa[j] = read_w(0, 0, j);

    int c_kt = 0, c_tile = first_tile;
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int k) {
        if constexpr ((PROF & 1) != 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[k] += t - t_prev;
            t_prev = t;
        }
    };
    if constexpr ((PROF & 1) != 0) t_prev = __builtin_amdgcn_s_memtime();
    [[maybe_unused]] int ca = 0;                    // (RA == 3) activation stage of the step being computed: step % 3
    for (int step = 0; step < total_steps; ++step) {
        const int cur = step & 1;
        // activation stages of this step / of the next one (the weight stages are cur / cur ^ 1)
        const int xa_cur = RA == 3 ? ca : cur, xa_nxt = RA == 3 ? (ca == 2 ? 0 : ca + 1) : cur ^ 1;
        // The instruction mix of a step is pinned with sched_barrier(0) fences: left alone the compiler
        // sinks all fragment reads below the MFMAs of a half (the wave then waits for them at the
        // barrier) and issues the DMA pieces as one burst.  None of the reads of a half is consumed
        // inside that half, so the fences create no waits.  MFMA chunk g = fragment column g.
        constexpr int DMA_TOTAL = A_PER + B_PER, DMA_PER_G = (DMA_TOTAL + FN - 1) / FN;
        // ---- first half: k 0..31 of slab `step`, while its k 32..63 fragments are read ----------
#pragma unroll
        for (int g = 0; g < FN; ++g) {
            wb[g] = read_w(cur, 1, g);
            if (g < FM) xb[g] = read_x(xa_cur, 1, g);
            if (g == FN - 1) {
#pragma unroll
                for (int i = FN; i < FM; ++i) xb[i] = read_x(xa_cur, 1, i);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((PROF & 64) == 0) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][g] = MDHIP_MFMA(wa[g], xa[i], acc[i][g]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        stamp(0);
        // slab step+1 has landed (this wave's pieces), stage `cur` is fully in registers
        // (RA == 3: the activation pieces of slab step+2, issued BEHIND the weight pieces of slab step+1 in the half step before,
        // may still be in flight: loads complete in order)
        if constexpr (RA == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(1);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        __builtin_amdgcn_sched_barrier(0);

        // ---- second half: X fragments of slab step+1, MFMAs on k 32..63 of slab step, and the DMA
        //      pieces of slab step+2 (into stage cur) spread behind the MFMA chunks --------------------
#pragma unroll
        for (int g = 0; g < FN; ++g) {
            wa[g] = read_w(cur ^ 1, 0, g);
            if (g < FM) xa[g] = read_x(xa_nxt, 0, g);
            if (g == FN - 1) {
#pragma unroll
                for (int i = FN; i < FM; ++i) xa[i] = read_x(xa_nxt, 0, i);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((PROF & 64) == 0) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][g] = MDHIP_MFMA(wb[g], xb[i], acc[i][g]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = g * DMA_PER_G; d < (g + 1) * DMA_PER_G && d < DMA_TOTAL; ++d) {
                if constexpr (RA == 3) {
                    // weight slab step+2 first, then activation slab step+3 into the stage this step has just finished with
                    if (d < B_PER) dma_b(cur, d);
                    else dma_a(xa_cur, d - B_PER);
                } else {
                    if (d < A_PER) dma_a(cur, d);
                    else dma_b(cur, d - A_PER);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        stamp(3);
        advance();
        if constexpr (RA == 3) {
            lb_advance();
            ca = ca == 2 ? 0 : ca + 1;
        }
        stamp(4);
        if (++c_kt == KT) {
            epilogue(c_tile);
            c_kt = 0;
            c_tile += tile_step;
        }
        stamp(5);
    }
    if constexpr ((PROF & 1) != 0) {
        if (lane == 0 && p.dbg) {
            unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = t_acc[k];
            d[6] = (unsigned long long)total_steps;
            d[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) ;   // HW_REG_XCC_ID
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table
// ---------------------------------------------------------------------------------------
// id (local), BM, BN, waves along M, waves along N
#define MDHIP_CONV2_CFGS(X) \
    X(0, 160, 160, 2, 2)    \
    X(1, 128, 160, 2, 2)    \
    X(2, 160, 80, 2, 1)     \
    X(3, 128, 80, 4, 1)     \
    X(4, 96, 160, 2, 2)     \
    X(5, 64, 160, 1, 2)     \
    X(6, 192, 160, 4, 2)    \
    X(7, 256, 160, 4, 2)    \
    X(8, 320, 160, 4, 2)
// pointwise-only configurations with the three-stage activation ring (RA = 3): id, BM, BN, WM, WN
#define MDHIP_CONV2_RING(X) \
    X(9, 96, 160, 2, 2)     \
    X(10, 64, 160, 1, 2)    \
    X(11, 320, 160, 4, 2)   \
    X(12, 256, 160, 4, 2)
// the configurations that also exist as a decoding instantiation (DEC = 1; conv2_cfg_decodes): id, BM, BN, WM, WN
#define MDHIP_CONV2_DEC(X) \
    X(0, 160, 160, 2, 2)   \
    X(2, 160, 80, 2, 1)    \
    X(3, 128, 80, 4, 1)    \
    X(8, 320, 160, 4, 2)
// id, BM, BN, WM, WN, PROF bits (1 = s_memtime stamps, 2 = no stores, 4 = no SiLU)
#define MDHIP_CONV2_PROF(X) \
    X(13, 160, 160, 2, 2, 1)  \
    X(14, 320, 160, 4, 2, 1) \
    X(15, 160, 160, 2, 2, 54) \
    X(16, 320, 160, 4, 2, 22) \
    X(17, 320, 160, 4, 2, 16) \
    X(18, 160, 160, 2, 2, 16)
// the pointwise ring instantiation (PW = 1, RA = 3) with PROF bits (round 6: where a 1x1 launch's time goes): id, BM, BN, WM, WN, PROF
#define MDHIP_CONV2_PROFRING(X) \
    X(19, 320, 160, 4, 2, 1)  \
    X(20, 320, 160, 4, 2, 2)  \
    X(21, 320, 160, 4, 2, 16) \
    X(22, 320, 160, 4, 2, 64)

static const ConvCfg g_cfgs2[] = {
#define X(id, bm, bn, wm, wn)                                                                        \
    {bm, bn, (wm) * (wn) * 64, (size_t)v2_lds_bytes(bm, bn), v2_blocks_per_cu(bm, bn, (wm) * (wn)), \
     "v2:" #bm "x" #bn "/" #wm "x" #wn},
    MDHIP_CONV2_CFGS(X)
#undef X
#define X(id, bm, bn, wm, wn)                                                                        \
    {bm, bn, (wm) * (wn) * 64, (size_t)v2_ring_lds_bytes(bm, bn), 163840 / v2_ring_lds_bytes(bm, bn) >= 2 ? 2 : 1, "v2:" #bm "x" #bn "/" #wm "x" #wn "/a3"},
    MDHIP_CONV2_RING(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    {bm, bn, (wm) * (wn) * 64, (size_t)v2_lds_bytes(bm, bn), v2_blocks_per_cu(bm, bn, (wm) * (wn)), \
     "v2prof" #prof ":" #bm "x" #bn "/" #wm "x" #wn},
    MDHIP_CONV2_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    {bm, bn, (wm) * (wn) * 64, (size_t)v2_ring_lds_bytes(bm, bn), 1, "v2prof" #prof ":" #bm "x" #bn "/" #wm "x" #wn "/a3"},
    MDHIP_CONV2_PROFRING(X)
#undef X
};
constexpr int kNumProf = 10;   // trailing instrumented entries: reachable through conv2_launch only

int conv2_num_cfgs() { return (int)(sizeof(g_cfgs2) / sizeof(g_cfgs2[0])) - kNumProf; }
const ConvCfg& conv2_cfg(int i) { return g_cfgs2[i]; }

hipError_t conv2_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, bn, wm, wn)                                                                        \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn>,                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes); \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, 0, false, 1>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes); \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, 0, false, 2>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes);
    MDHIP_CONV2_CFGS(X)
#undef X
#define X(id, bm, bn, wm, wn)                                                                        \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, 0, false, 1, 3>,          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes); \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, 0, false, 2, 3>,          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes);
    MDHIP_CONV2_RING(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, prof>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes);
    MDHIP_CONV2_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, prof, false, 1, 3>,       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes);
    MDHIP_CONV2_PROFRING(X)
#undef X
#define X(id, bm, bn, wm, wn)                                                                        \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<bm, bn, wm, wn, 0, false, 1, 2, 1>,       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs2[id].lds_bytes);
    MDHIP_CONV2_DEC(X)
#undef X
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_v2_kernel<160, 160, 2, 2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)g_cfgs2[0].lds_bytes);
    return e;
}

// (the configurations with a three-stage activation ring: 1x1 / stride 1 / unpadded launches only, at least three K slabs)
bool conv2_cfg_is_ring(int cfg) { return cfg >= 9 && cfg < conv2_num_cfgs(); }
// the configurations with a decoding instantiation (pointwise, channel count a multiple of 64): the tiles the tables give the
// 24-channel Detect convs
bool conv2_cfg_decodes(int cfg) { return cfg == 0 || cfg == 2 || cfg == 3 || cfg == 8; }
bool conv2_is_pointwise(const ConvArgs& a) {
    return a.ntaps == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo && a.in_up == nullptr;
}
bool conv2_supports(const ConvArgs& a) {
    // the branch-free K walk needs at least one whole slab per tap; the epilogue stores 4 channels
    // (in_up goes with configuration 0 only: the family table in conv_igemm.cpp checks the id)
    if (a.in_up && !(a.ntaps == 1 && a.stride == 1 && (a.H % 2) == 0 && (a.W % 2) == 0 && a.up_slabs > 0 &&
                     a.up_slabs * 8 <= a.C8 && !a.in_f8))
        return false;
    return a.C8 >= 8 && a.kw <= 3 && a.ntaps <= 9 && (a.k_pad % 64) == 0;
}

hipError_t conv2_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (cfg < 0 || cfg >= conv2_num_cfgs() + kNumProf || !conv2_supports(a)) return hipErrorInvalidValue;
    const ConvCfg& c = g_cfgs2[cfg];
    ConvArgs p = a;
    conv_set_rcp(p);
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, (32 * c.blocks_per_cu) / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    if (a.in_up) {                                     // upsample read in place: configuration 0 (160x160) only
        hipLaunchKernelGGL((conv_v2_kernel<160, 160, 2, 2, 0, true>), grid, dim3(256), c.lds_bytes, s, p);
        return hipGetLastError();
    }
    // 1x1 / stride 1 / unpadded: the instantiation whose tile set-up needs no divisions (same loads, same results)
    const bool pw = a.ntaps == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo;
    if (a.dec_pred) {                                  // Detect conv that decodes in its epilogue
        if (!pw || (a.C8 & 7) != 0 || !a.out_f32 || !conv2_cfg_decodes(cfg)) return hipErrorInvalidValue;
        switch (cfg) {
#define X(id, bm, bn, wm, wn)                                                                        \
    case id:                                                                                       \
        hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, 0, false, 1, 2, 1>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
            MDHIP_CONV2_DEC(X)
#undef X
        }
        return hipGetLastError();
    }
    switch (cfg) {
#define X(id, bm, bn, wm, wn)                                                                        \
    case id:                                                                                       \
        if (pw && (a.C8 & 7) == 0) hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, 0, false, 1>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        else if (pw) hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, 0, false, 2>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        else hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV2_CFGS(X)
#undef X
#define X(id, bm, bn, wm, wn)                                                                        \
    case id:                                                                                       \
        if (!pw) return hipErrorInvalidValue;                                                      \
        if ((a.C8 & 7) == 0) hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, 0, false, 1, 3>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        else hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, 0, false, 2, 3>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV2_RING(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    case id:                                                                                       \
        hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, prof>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV2_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    case id:                                                                                       \
        if (!pw || (a.C8 & 7) != 0) return hipErrorInvalidValue;                                   \
        hipLaunchKernelGGL((conv_v2_kernel<bm, bn, wm, wn, prof, false, 1, 3>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV2_PROFRING(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
