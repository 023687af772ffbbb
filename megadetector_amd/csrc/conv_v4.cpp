// 3x3 / stride 1 convolution as a *row-patch direct convolution* on MFMA (gfx950 / MI355X).
//
// Why: the implicit-GEMM kernels (conv_igemm.cpp, conv_v2.cpp) gather every input pixel nine times
// (once per tap) through the L2 -> LDS path, and that path -- about half a cache-line request per clock
// per CU -- is what bounds them (a 160x160x64 slab is 80 FLOP/B against a CU
// balance of 64 FLOP/B; measured kernel time = DMA-only time + MFMA-only time).  Here a workgroup
// loads, per group of 64 input channels, ONE patch of (R+2) x (WT+2) input pixels into LDS and serves
// all nine taps from it: the A fragment of tap (r,s) is the same LDS image read at a pixel shift of
// r*(WT+2)+s rows.  Out-of-image halo pixels are lanes whose DMA offset is out of the descriptor's
// range (zeros).  Only the weights (one 160 x 64 slab per tap) are streamed per step.  With an 8 x 40
// pixel tile per CU: A traffic / 5.5, 250 FLOP per L2->LDS byte instead of 80.
//
// K order is (channel group, r, s, channel in group), so the weights are packed a second time in
// that order (mdhip_capi.cpp) and the fp32 summation order differs from the implicit-GEMM kernels:
// results agree with them to rounding (not bitwise); batch-composition invariance stays bitwise.
//
// Workgroup: 8 waves = 4 (pairs of tile rows) x 2 (80 output channels), wave tile 80 pixels x 80
// channels = 5x5 16x16x32 MFMA fragments, one workgroup per CU (LDS: 2 patch buffers of 53 KiB + 2
// weight stages of 20 KiB).  Schedule per step (= one tap of one channel group, 64 deep), as conv_v2:
//     read  Y  (k 32..63: patch @tap, weight stage cur)      } interleaved
//     mfma  X  (k 0..31)                                     }
//     s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier
//     DMA   weight slab of step+2 -> stage cur ; one piece of the NEXT group's patch
//     read  X' (k 0..31 of step+1)                           } interleaved
//     mfma  Y                                                }

#include <algorithm>
#include <type_traits>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) char lds_char;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;

__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

constexpr int v4_patch_pieces(int r, int wt) { return ((r + 2) * (wt + 2) + 7) / 8; }
// 2 weight stages + 2 patch buffers + 1 KiB dump slot (destination of predicated-off DMA pieces)
constexpr int v4_lds_bytes(int r, int wt, int bn) { return 2 * bn * 128 + 2 * v4_patch_pieces(r, wt) * 1024 + 1024; }

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// PROF bits (developer builds only): 2 = no stores, 4 = no SiLU, 16 = no DMA in the steady state
template <int R, int WT, int WM, int WN, int PROF = 0>
__global__ void __launch_bounds__(512, 2)
conv_v4_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    static_assert(NW == 8, "eight waves per workgroup");
    constexpr int BM = R * WT, BN = WN * 80;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int PW = WT + 2, PROWS = (R + 2) * PW;
    constexpr int P_PIECES = v4_patch_pieces(R, WT), P_BYTES = P_PIECES * 1024;
    constexpr int P_PER = (P_PIECES + NW - 1) / NW;                   // patch pieces per wave per group
    constexpr int B_BYTES = BN * 128, B_PIECES = BN / 8, B_PER = (B_PIECES + NW - 1) / NW;
    constexpr int P_OFF = 2 * B_BYTES;
    constexpr int DUMP_OFF = P_OFF + 2 * P_BYTES;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "16x16 fragments");
    static_assert(P_PER <= 9, "one patch piece per tap at most");

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
    const int G = p.groups;                       // 64-channel groups (the last one may be half full)
    const int steps_per_tile = 9 * G;
    const int total_steps = my_tiles * steps_per_tile;
    const int tiles_w = p.W / WT, tiles_img = tiles_w * (p.H / R);

    // tile index -> (first pixel of the tile, first pixel of its patch) in units of pixels
    auto tile_origin = [&](int t, int& b, int& y0, int& x0) __attribute__((always_inline)) {
        b = t / tiles_img;
        const int rem = t - b * tiles_img;
        const int band = rem / tiles_w;
        y0 = band * R;
        x0 = (rem - band * tiles_w) * WT;
    };

    // ---- weight stream: slab (cg, tap) = 128 bytes of every row at byte offset step * 128 ---------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt4 + (size_t)n0 * p.k_pad4), 0, kNumRecords, 0x00020000);
    unsigned b_off[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int row = (i * NW + wave) * 8 + lr;
        b_off[i] = (row < BN && n0 + row < p.n_rows) ? (unsigned)(row * p.k_pad4 + jj * 8) * 2u : kOOB;
    }
    int l_step = 0;                                // the weight loader's step inside a tile (same for every tile)
    // every wave issues exactly B_PER pieces per step (no branches inside the step: the scheduling
    // groups below need one basic block); a surplus piece reads zeros into the dump slot
    auto dma_b_piece = [&](int stage, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        const bool real = (B_PIECES % NW) == 0 || i < B_PER - 1 || wave < B_PIECES % NW;      // wave-uniform
        lds_char* dst = smem + (real ? stage * B_BYTES + (i * NW + wave) * 1024 : DUMP_OFF);
        MDHIP_DMA16(b_rsrc, dst, b_off[i], l_step * 128);
    };
    auto dma_b_done = [&]() __attribute__((always_inline)) { l_step = (l_step + 1 == steps_per_tile) ? 0 : l_step + 1; };
    auto dma_b = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) dma_b_piece(stage, i);
        dma_b_done();
    };

    // ---- patch loader: runs one channel group ahead of the consumer ----------------------------------
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    unsigned p_off[P_PER];                         // byte offset of this lane's patch pixel from the patch origin, or OOB
    int pl_tile = first_tile, pl_cg = 0;           // the group whose patch is being loaded
    bool pl_live = true;
    auto patch_tile = [&](int t) __attribute__((always_inline)) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        const long long origin = ((long long)(b * p.H + y0 - 1) * p.W + (x0 - 1));
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + origin * p.ld_in), 0, kNumRecords, 0x00020000);
#pragma unroll
        for (int i = 0; i < P_PER; ++i) {
            const int pr = (i * NW + wave) * 8 + lr;
            const int line = pr / PW, col = pr - line * PW;
            const int y = y0 - 1 + line, x = x0 - 1 + col;
            const bool ok = pr < PROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            p_off[i] = ok ? (unsigned)((line * p.W + col) * p.ld_in * 2) : kOOB;
        }
    };
    auto dma_patch_piece = [&](int buf, int i) __attribute__((always_inline)) {      // prologue: piece i, all of them
        if constexpr ((PROF & 16) != 0) return;
        if (i * NW + wave >= P_PIECES) return;                                                // wave-uniform
        const bool chunk_ok = pl_cg * 8 + jj < p.C8;
        const unsigned voff = (p_off[i] != kOOB && chunk_ok && pl_live) ? p_off[i] + (unsigned)(pl_cg * 128 + jj * 16) : kOOB;
        MDHIP_DMA16(a_rsrc, smem + P_OFF + buf * P_BYTES + (i * NW + wave) * 1024, voff, 0);
    };
    // steady state: tap t loads piece t of the next group's patch (taps >= P_PER: a zero piece into the dump slot)
    auto dma_patch_tap = [&](int buf, int t) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        unsigned po = kOOB;
#pragma unroll
        for (int i = 0; i < P_PER; ++i) po = (t == i) ? p_off[i] : po;                          // wave-uniform selects
        const int q = t * NW + wave;
        const bool real = t < P_PER && q < P_PIECES;
        const bool chunk_ok = pl_cg * 8 + jj < p.C8;
        const unsigned voff = (po != kOOB && chunk_ok && pl_live && real) ? po + (unsigned)(pl_cg * 128 + jj * 16) : kOOB;
        lds_char* dst = smem + (real ? P_OFF + buf * P_BYTES + q * 1024 : DUMP_OFF);
        MDHIP_DMA16(a_rsrc, dst, voff, 0);
    };
    auto patch_next_group = [&]() __attribute__((always_inline)) {
        if (++pl_cg == G) {
            pl_cg = 0;
            if (pl_tile == last_tile) {
                pl_live = false;
            } else {
                pl_tile += tile_step;
                patch_tile(pl_tile);
            }
        }
    };

    // ---- fragment addressing ----------------------------------------------------------------------
    // A: lane reads patch pixel-row idx = fidx[i] + r*PW + s, 16-byte chunk ((lane>>4) + 4*kk) ^ (idx & 7)
    int fidx[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int px = wm * TM + i * 16 + (lane & 15);
        const int o = px / WT;
        fidx[i] = o * PW + (px - o * WT);
    }
    const int c0 = lane >> 4;
    unsigned a_addr[FM] = {};                      // LDS byte address of the k 0..31 chunk for the tap being read
    auto set_a_addr = [&](int buf, int tap) __attribute__((always_inline)) {
        if constexpr ((PROF & 128) != 0) { if (a_addr[0] != 0) return; }      // ablation: addresses computed once
        const int r = tap / 3;
        const int shift = r * PW + (tap - r * 3);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int idx = fidx[i] + shift;
            a_addr[i] = (unsigned)(P_OFF + buf * P_BYTES + idx * 128 + ((c0 ^ (idx & 7)) << 4));
        }
    };
    auto set_a_addr_one = [&](int buf, int shift, int i) __attribute__((always_inline)) {
        const int idx = fidx[i] + shift;
        a_addr[i] = (unsigned)(P_OFF + buf * P_BYTES + idx * 128 + ((c0 ^ (idx & 7)) << 4));
    };
    auto read_x = [&](int i, int kk) __attribute__((always_inline)) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + (a_addr[i] ^ (unsigned)(kk * 64)));
    };
    // B: as conv_igemm.cpp: row (lane & 15) of a 16-row fragment, chunk (lane>>4) ^ (row & 7), k 32..63 is ^ 64
    const int b_frag_base = (wn * TN) * 128 + (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) * 16);
    auto read_w = [&](int stage, int kk, int j) __attribute__((always_inline)) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + stage * B_BYTES + ((b_frag_base + j * 2048) ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue -------------------------------------------------------------------------------------
    const int q4 = lane >> 4;
    auto epilogue_t = [&](int tile_m, auto has_res_t, auto out_f32_t) __attribute__((always_inline)) {
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        constexpr bool OUT_F32 = decltype(out_f32_t)::value;
        int b, y0, x0;
        tile_origin(tile_m, b, y0, x0);
        const long long m_org = (long long)(b * p.H + y0) * p.W + x0;        // first pixel of the tile
        const int esz = OUT_F32 ? 4 : 2;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((char*)p.out + (size_t)m_org * p.ld_out * esz), 0, kNumRecords, 0x00020000);
        int mrel[FM];                                                        // pixel offset from the tile's first pixel
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int px = wm * TM + i * 16 + (lane & 15);
            const int o = px / WT;
            mrel[i] = o * p.W + (px - o * WT);
        }
        const int nbase = n0 + wn * TN + q4 * 4;
        // bias of all fragment columns first (scalar loads), then pixel-row by pixel-row with 16-byte
        // stores (see conv_v2.cpp's epilogue)
        float bv[FN][4];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int nb = n0 + wn * TN + j * 16;                // wave-uniform
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
        auto fetch_res_row = [&](int i, uint2 (&r)[FN]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
                r[j] = *(const uint2*)(p.res + (size_t)(m_org + mrel[i]) * p.ld_res + min(nbase + j * 16, p.N - 4));
        };
        if constexpr (HAS_RES) fetch_res_row(0, rrow[0]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (HAS_RES) {
                if (i + 1 < FM) fetch_res_row(i + 1, rrow[(i + 1) & 1]);
            }
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                mdhip_bias4(acc[i][j], bv[j], v[j]);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if ((PROF & 4) == 0 && p.act) {
#pragma unroll
                for (int j = 0; j < FN; ++j) mdhip_silu4(v[j]);
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
            } else if constexpr (OUT_F32) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = nbase + j * 16;
                    const unsigned voff = (n < p.N) ? (unsigned)(mrel[i] * p.ld_out + n) * 4u : kOOB;
                    const u32x4 o = {__float_as_uint(v[j][0]), __float_as_uint(v[j][1]), __float_as_uint(v[j][2]), __float_as_uint(v[j][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(o, o_rsrc, voff, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    const unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                    const unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    const auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                    const int n = n0 + wn * TN + j * 16 + q4 * 8;
                    const unsigned voff = (n < p.N) ? (unsigned)(mrel[i] * p.ld_out + n) * 2u : kOOB;
                    const u32x4 o = {t0[0], t1[0], t0[1], t1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, o_rsrc, voff, 0, 0);
                }
                if (FN & 1) {
                    const int j = FN - 1;
                    const int n = nbase + j * 16;
                    const unsigned voff = (n < p.N) ? (unsigned)(mrel[i] * p.ld_out + n) * 2u : kOOB;
                    const u32x2 o = {st_pack2(v[j][0], v[j][1]), st_pack2(v[j][2], v[j][3])};
                    __builtin_amdgcn_raw_buffer_store_b64(o, o_rsrc, voff, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int tile_m) __attribute__((always_inline)) {
        if (p.out_f32) epilogue_t(tile_m, std::false_type{}, std::true_type{});
        else if (p.res) epilogue_t(tile_m, std::true_type{}, std::false_type{});
        else epilogue_t(tile_m, std::false_type{}, std::false_type{});
    };

    // ---- prologue: patch of (first tile, group 0) in buffer 0, weight slabs of steps 0 and 1 ---------
    patch_tile(first_tile);
#pragma unroll
    for (int i = 0; i < P_PER; ++i) dma_patch_piece(0, i);
    patch_next_group();
    dma_b(0);
    dma_b(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
    set_a_addr(0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) xa[i] = read_x(i, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) wa[j] = read_w(0, 0, j);

    // The instruction mix of a step is pinned with sched_barrier(0) fences: left alone the compiler
    // sinks all fragment reads below the 25 MFMAs of a half and the wave then waits for them at the
    // barrier (and the DMA issue gets no MFMA cover).  None of the reads of a half is consumed inside
    // that half, so the fences do not create waits.
    int tap = 0, cg = 0, pbuf = 0, c_tile = first_tile;
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr ((PROF & 1) != 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[k] += t - t_prev;
            t_prev = t;
        }
    };
    if constexpr ((PROF & 1) != 0) t_prev = __builtin_amdgcn_s_memtime();
#define MDHIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    // a half-full last channel group (C_in mod 64 <= 32) has nothing in k 32..63: its second-half MFMAs are skipped
    const bool tail_short = (p.C8 & 7) != 0 && (p.C8 & 7) <= 4;
    for (int step = 0; step < total_steps; ++step) {
        const int cur = step & 1;
        const bool wrap = tap == 8;
        const int ntap = wrap ? 0 : tap + 1;
        const int nbuf = wrap ? pbuf ^ 1 : pbuf;
        const int nr = ntap / 3;
        const int nshift = nr * PW + (ntap - nr * 3);
        const bool skip_y = tail_short && cg == G - 1;
        // ---- first half: k 0..31 of this step, while its k 32..63 fragments are read and the patch
        //      addresses of the next tap are computed; MFMA chunk g = fragment column g -----------------
#pragma unroll
        for (int g = 0; g < FN; ++g) {
            wb[g] = read_w(cur, 1, g);
            if (g < FM) { xb[g] = read_x(g, 1); set_a_addr_one(nbuf, nshift, g); }
            if (g == FN - 1) {
#pragma unroll
                for (int i = FN; i < FM; ++i) { xb[i] = read_x(i, 1); set_a_addr_one(nbuf, nshift, i); }
            }
            MDHIP_FENCE();
#pragma unroll
            for (int i = 0; i < FM; ++i)
                acc[i][g] = MDHIP_MFMA(wa[g], xa[i], acc[i][g]);
            MDHIP_FENCE();
        }

        stamp(0);
        // everything this wave requested has landed; its reads of weight stage `cur` are complete
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(1);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        MDHIP_FENCE();

        // ---- second half: the k 0..31 fragments of step+1 (possibly the next group's first tap) and
        //      MFMAs on k 32..63; one DMA piece (weight slab of step+2, then one piece of the next
        //      group's patch) behind every chunk of MFMAs, so that the eight waves' requests do not
        //      arrive as one burst -----------------------------------------------------------------------
        static_assert(B_PER + 1 <= FN, "one DMA piece per MFMA chunk");
#pragma unroll
        for (int g = 0; g < FN; ++g) {
            wa[g] = read_w(cur ^ 1, 0, g);
            if (g < FM) xa[g] = read_x(g, 0);
            if (g == FN - 1) {
#pragma unroll
                for (int i = FN; i < FM; ++i) xa[i] = read_x(i, 0);
            }
            MDHIP_FENCE();
            if (!skip_y) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][g] = MDHIP_MFMA(wb[g], xb[i], acc[i][g]);
            }
            MDHIP_FENCE();
            if (g < B_PER) dma_b_piece(cur, g);
            if (g == B_PER) dma_patch_tap(pbuf ^ 1, tap);
            MDHIP_FENCE();
        }
        dma_b_done();
#undef MDHIP_FENCE

        stamp(3);
        if (wrap) {                            // the group is done: the loader moves on, maybe the tile too
            patch_next_group();
            if (++cg == G) {
                cg = 0;
                epilogue(c_tile);
                c_tile += tile_step;
            }
        }
        tap = ntap;
        pbuf = nbuf;
        stamp(5);
    }
    if constexpr ((PROF & 1) != 0) {
        if (lane == 0 && p.dbg) {
            unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = t_acc[k];
            d[6] = (unsigned long long)total_steps;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table
// ---------------------------------------------------------------------------------------
// id (local), tile rows, tile columns, waves along M, waves along N (80 output channels each), PROF bits
#define MDHIP_CONV4_CFGS(X) \
    X(0, 8, 40, 4, 2, 0)    \
    X(1, 8, 32, 8, 1, 0)
#define MDHIP_CONV4_PROF(X) \
    X(2, 8, 40, 4, 2, 22)   \
    X(3, 8, 40, 4, 2, 1)    \
    X(4, 8, 32, 8, 1, 1)

struct V4Geom { int r, wt, bn; };
static const V4Geom g_geom4[] = {
#define X(id, r, wt, wm, wn, prof) {r, wt, (wn) * 80},
    MDHIP_CONV4_CFGS(X) MDHIP_CONV4_PROF(X)
#undef X
};
static const ConvCfg g_cfgs4[] = {
#define X(id, r, wt, wm, wn, prof) \
    {(r) * (wt), (wn) * 80, 512, (size_t)v4_lds_bytes(r, wt, (wn) * 80), 1, "v4:patch" #r "x" #wt "/" #wm "x" #wn "/" #prof},
    MDHIP_CONV4_CFGS(X) MDHIP_CONV4_PROF(X)
#undef X
};
constexpr int kNumProf4 = 3;

int conv4_num_cfgs() { return (int)(sizeof(g_cfgs4) / sizeof(g_cfgs4[0])) - kNumProf4; }
const ConvCfg& conv4_cfg(int i) { return g_cfgs4[i]; }

hipError_t conv4_init() {
    hipError_t e = hipSuccess;
#define X(id, r, wt, wm, wn, prof)                                                             \
    if (e == hipSuccess)                                                                     \
        e = hipFuncSetAttribute((const void*)conv_v4_kernel<r, wt, wm, wn, prof>,               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs4[id].lds_bytes);
    MDHIP_CONV4_CFGS(X) MDHIP_CONV4_PROF(X)
#undef X
    return e;
}

bool conv4_supports(int cfg, const ConvArgs& a) {
    if (cfg < 0 || cfg >= conv4_num_cfgs() + kNumProf4) return false;
    const int r = g_geom4[cfg].r, wt = g_geom4[cfg].wt;
    return a.wgt4 != nullptr && a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H &&
           a.Wo == a.W && (a.H % r) == 0 && (a.W % wt) == 0 && a.C8 >= 8 && (a.N % 8) == 0 &&
           (long long)(r + 2) * a.W * a.ld_in * 2 < 0x7fffffffLL && (long long)r * a.W * a.ld_out * 4 < 0x7fffffffLL;
}

hipError_t conv4_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (!conv4_supports(cfg, a)) return hipErrorInvalidValue;
    const ConvCfg& c = g_cfgs4[cfg];
    ConvArgs p = a;
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = a.M / c.bm;                                        // exact: whole tiles only
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, 32 / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    switch (cfg) {
#define X(id, r, wt, wm, wn, prof)                                                              \
    case id:                                                                                  \
        hipLaunchKernelGGL((conv_v4_kernel<r, wt, wm, wn, prof>), grid, dim3(512), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV4_CFGS(X) MDHIP_CONV4_PROF(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
