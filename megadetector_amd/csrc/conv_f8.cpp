// 3x3 / stride 1 convolution on fp8 (OCP e4m3) operands: conv_v5's row-segment structure with the gfx950
// block-scaled MFMA  v_mfma_scale_f32_16x16x128_f8f6f4  (K = 128 per instruction, twice the bf16 rate, unit block
// scales) -- BASELINE.json configs[4], the W8A8 half of the MDHIP_DTYPE_FP8 mode.
//
// What changes against conv_v5.cpp:
//   * an LDS row of 128 bytes holds 128 channels (a channel group) of one pixel / one weight row instead of 64;
//     a step (one tap of one channel group) is therefore 128 deep, with the same bytes through L2 -> LDS and the
//     same bytes read from LDS as a 64-deep bf16 step: per byte moved the kernel does twice the work;
//   * an MFMA operand is 32 bytes per lane = the two 16-byte chunks  kb  and  kb + 4  of the lane's row
//     (kb = lane >> 4).  Which 32 of the 128 k values a lane group holds is free as long as both operands agree,
//     and this choice makes the two reads exactly conv_v5's conflict-free `kk = 0 / 1` reads;
//   * one instruction consumes the whole 128-byte row, so a step cannot be halved along K.  It is halved along the
//     fragment COLUMNS instead: first half = columns [0, FN1) of this step while the weight fragments of the other
//     columns are read; barrier (every read of weight stage `cur` is complete); second half = the remaining columns
//     while the activation fragments of the NEXT step (second register set) and its first FN1 weight columns are
//     read and the DMA pieces of step + 2 are issued;
//   * the fp32 accumulator is scaled per output channel (activation scale x weight scale, ConvArgs::scale) before
//     bias and SiLU; the output is 16-bit (bf16 / fp16), residual 16-bit.
// Zero padding, tap masks, persistent XCD-local streams, K order (channel group, r, s, channel) as conv_v5.

#include <algorithm>
#include <type_traits>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((address_space(3))) char lds_char;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;
[[maybe_unused]] constexpr int kUnitScale = 0x7f7f7f7f;          // E8M0 127 = 2^0 in every byte (tools/probe_fp8.cpp)

__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

constexpr int f8_run_pieces(int bm) { return (bm + 2 + 7) / 8; }
// 2 run buffers + 2 weight stages + 2 KiB: the row of zeros, the staged bias and the staged scales
constexpr int f8_lds_bytes(int bm, int bn) { return 2 * f8_run_pieces(bm) * 1024 + 2 * bn * 128 + 2048; }
constexpr int f8_blocks_per_cu(int bm, int bn, int nw) {
    int b = 163840 / f8_lds_bytes(bm, bn);
    if (b > 8 / nw) b = 8 / nw;          // two waves per SIMD (256 registers each)
    return b < 1 ? 1 : b;
}
constexpr int f8_waves_per_simd(int bm, int bn, int nw) {
    int w = f8_blocks_per_cu(bm, bn, nw) * nw / 4;
    return w < 1 ? 1 : w;
}

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)
// The builtin form (__builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4) has no accumulate-in-place variant in this
// compiler: destination and C operand get different registers, the accumulators ping-pong between two register sets
// and the kernel spills (measured: 310 spilled VGPRs for the 128x160 tile).  The instruction is therefore written out
// with the accumulator tied.  hipcc pads nothing around an asm statement, so the statement carries its own
// s_nop for the VALU-write -> MFMA-read wait states (operands normally come straight from ds_read, which the
// compiler's s_waitcnt covers), and the epilogue starts with the MFMA-write -> VALU-read wait states.
#define MDHIP_MFMA8(a, b, c)                                                                                   \
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"          \
                 : "+v"(c)                                                                                     \
                 : "v"(a), "v"(b), "v"(unit_scale))

// PROF bits (developer builds only): 1 = s_memtime stamps, 2 = no stores, 4 = no SiLU, 16 = no DMA in the steady state
template <int BM, int BN, int WM, int WN, int PROF = 0>
__global__ void __launch_bounds__(WM * WN * 64, f8_waves_per_simd(BM, BN, WM * WN))
conv_f8_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int FN1 = FN / 2;                              // fragment columns of the first half-step
    constexpr int A_PIECES = f8_run_pieces(BM), A_BUF = A_PIECES * 1024;
    constexpr int A_PER = (A_PIECES + NW - 1) / NW;
    constexpr int A_H0 = (A_PER + 1) / 2;
    constexpr int B_BYTES = BN * 128, B_PIECES = BN / 8, B_PER = (B_PIECES + NW - 1) / NW;
    constexpr int B_OFF = 2 * A_BUF;
    constexpr int ZERO_OFF = B_OFF + 2 * B_BYTES;
    static_assert(TM % 16 == 0 && TN % 16 == 0 && FN >= 2, "16x16 fragments, at least two columns");

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
    const int G = p.groups8;                      // 128-channel groups (the last one may be partly filled)
    const int runs_per_tile = 3 * G;
    const int steps_per_tile = 9 * G;
    const int total_runs = my_tiles * runs_per_tile;

    // the row of zeros; behind it the bias (offset 256) and the per-channel scales (offset 256 + 4 BN) of this
    // workgroup's BN output channels, staged once and read by every tile's epilogue with ds_read_b128
    static_assert(BN * 8 + 256 <= 2048, "bias / scale staging area");
    constexpr int BIAS_OFF = ZERO_OFF + 256, SCALE_OFF = BIAS_OFF + BN * 4;
    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < BN; c += NW * 64) {
        const bool ok = n0 + c < p.n_rows;
        *(__attribute__((address_space(3))) float*)(smem + BIAS_OFF + c * 4) = ok ? p.bias[n0 + c] : 0.f;
        *(__attribute__((address_space(3))) float*)(smem + SCALE_OFF + c * 4) = ok ? p.scale[n0 + c] : 0.f;
    }

    // ---- weight stream: slab (cg, tap) = 128 bytes of every row at byte offset step * 128 ---------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt8 + (size_t)n0 * p.k_pad8), 0, kNumRecords, 0x00020000);
    unsigned b_off[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int row = (i * NW + wave) * 8 + lr;
        b_off[i] = (row < BN && n0 + row < p.n_rows) ? (unsigned)(row * p.k_pad8 + jj * 16) : kOOB;
    }
    int l_step = 0;
    auto dma_b_piece = [&](int stage, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if ((B_PIECES % NW) != 0 && i == B_PER - 1 && wave >= B_PIECES % NW) return;           // wave-uniform
        MDHIP_DMA16(b_rsrc, smem + B_OFF + stage * B_BYTES + (i * NW + wave) * 1024, b_off[i], l_step * 128);
    };
    auto dma_b_done = [&]() __attribute__((always_inline)) { l_step = (l_step + 1 == steps_per_tile) ? 0 : l_step + 1; };

    // ---- run loader (conv_v5.cpp): buffer row q of run (tile, cg, r) = input pixel tile*BM + (r-1)*W - 1 + q,
    //      channels cg*128 .. cg*128+127; one byte per channel, ld_in bytes per pixel ------------------------
    const uint8_t* const in8 = (const uint8_t*)p.in;
    // (as conv_v5.cpp: one descriptor over the whole tensor, a piece is addressed from the tensor's first byte and the range
    // check zeroes the pixels outside the batch; the channel test stays -- an e4m3 MFMA reads all 128 channels of a group, and
    // what lies behind the last channel must be zeros, not the next pixel)
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)in8, 0, (int)((unsigned)p.M * (unsigned)p.ld_in), 0x00020000);
    unsigned q_off[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int q = (i * NW + wave) * 8 + lr;
        q_off[i] = (unsigned)(q * p.ld_in + jj * 16);
    }
    int lg_tile = first_tile, lg_cg = 0, lg_r = 0;
    int lg_first = 0;
    unsigned lg_abs = 0;                           // byte offset of (run's first pixel, channel group) in the tensor, mod 2^32
    auto run_setup = [&]() __attribute__((always_inline)) {
        lg_first = lg_tile * BM + (lg_r - 1) * p.W - 1;
        lg_abs = (unsigned)lg_first * (unsigned)p.ld_in + (unsigned)(lg_cg * 128);
    };
    auto dma_run_piece = [&](int buf, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if (i * NW + NW - 1 >= A_PIECES && i * NW + wave >= A_PIECES) return;                   // wave-uniform (conv_v5.cpp)
        const bool ok = jj < p.C8 - lg_cg * 8;
        MDHIP_DMA16(a_rsrc, smem + buf * A_BUF + (i * NW + wave) * 1024, ok ? q_off[i] + lg_abs : kOOB, 0);
    };
    auto run_next = [&]() __attribute__((always_inline)) {
        if (++lg_r == 3) {
            lg_r = 0;
            if (++lg_cg == G) {
                lg_cg = 0;
                if (lg_tile != last_tile) lg_tile += tile_step;
            }
        }
        run_setup();
    };

    // ---- fragment reads: the 16-byte chunk c of buffer row q sits at position c ^ (q & 7); a lane's operand is
    //      chunk kb (h = 0) followed by chunk kb + 4 (h = 1) of its row, kb = lane >> 4 -----------------------
    const int c0 = lane >> 4;
    unsigned a_sh[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
        a_sh[s] = (unsigned)((wm * TM + (lane & 15) + s) * 128 + ((c0 ^ (((lane & 7) + s) & 7)) << 4));
    const unsigned z_addr = (unsigned)(ZERO_OFF + c0 * 16);
    const int b_frag_base = B_OFF + (wn * TN + (lane & 15)) * 128 + ((c0 ^ (lane & 7)) << 4);
    uint32_t vmask[FM];
    unsigned a_eff[FM];
    auto tile_masks = [&](int t) __attribute__((always_inline)) {
        const int mb = t * BM + wm * TM + (lane & 15);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mb + i * 16;
            uint32_t mask = 0;
            if (m < p.M) {
                const int b = m / p.HoWo;
                const int rem = m - b * p.HoWo;
                const int y = rem / p.W;
                const int x = rem - y * p.W;
                const uint32_t rows = (y > 0 ? 0x007u : 0u) | 0x038u | (y < p.H - 1 ? 0x1c0u : 0u);
                const uint32_t cols = (x > 0 ? 0x049u : 0u) | 0x092u | (x < p.W - 1 ? 0x124u : 0u);
                mask = rows & cols;
            }
            vmask[i] = mask;
        }
    };
    auto set_a_eff_one = [&](int buf, int r, int s, int i) __attribute__((always_inline)) {
        const unsigned a = a_sh[s] + (unsigned)(buf * A_BUF + i * 2048);
        a_eff[i] = ((vmask[i] >> (r * 3 + s)) & 1u) ? a : z_addr;
        asm volatile("" : "+v"(a_eff[i]));         // (pinned where it is written, conv_v5.cpp)
    };
    auto ld16 = [&](unsigned addr) -> i32x4 {
        return *(const __attribute__((address_space(3))) i32x4*)(smem + addr);
    };
    auto join = [](i32x4 lo, i32x4 hi) -> i32x8 {
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto read_x = [&](int i) -> i32x8 { return join(ld16(a_eff[i]), ld16(a_eff[i] ^ 64u)); };
    auto read_w = [&](int stage, int j) -> i32x8 {
        const unsigned a = (unsigned)(stage * B_BYTES + j * 2048 + b_frag_base);
        return join(ld16(a), ld16(a ^ 64u));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int unit_scale = kUnitScale;
    asm volatile("" : "+v"(unit_scale));             // a VGPR holding the E8M0 unit scale of both operands

    // ---- epilogue (conv_v5.cpp's, plus the per-channel scale) -----------------------------------------------
    auto epilogue_t = [&](int tile_m, auto has_res_t) __attribute__((always_inline)) {
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // MFMA (16 passes) write -> VALU read of the accumulators
        // (an opaque copy of the lane id: lane-only offsets must not be hoisted ahead of the main loop, conv_v5.cpp)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int q4 = lane_e >> 4;
        const int nbase = n0 + wn * TN + q4 * 4;
        // bias and scale are re-read from LDS for every pixel row (volatile: 40 live registers less than keeping them
        // across the rows, which made the kernel spill)
        auto ld4 = [&](int off) -> f32x4 { return *(const volatile __attribute__((address_space(3))) f32x4*)(smem + off); };
        // Outputs and the residual go through buffer instructions (conv_v5.cpp): one 32-bit offset register per lane and
        // tensor instead of 64-bit address pairs, which were spilled and reloaded -- with an s_waitcnt vmcnt(0) each --
        // in front of every store; rows past the tensor's end are dropped / read as zeros by the range check.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const int npair0 = n0 + wn * TN + q4 * 8;
        const int nlast = nbase + (FN - 1) * 16;
        // (descriptors start at the tile's first pixel: offsets stay small whatever the tensor's size)
        const long long rows_left = (long long)p.M - (long long)tile_m * BM;
        const int ml = wm * TM + (lane_e & 15);
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((uint16_t*)p.out + (size_t)tile_m * BM * p.ld_out), 0, (int)min(rows_left * p.ld_out * 2, 0x7fffffffLL), 0x00020000);
        const unsigned o_pair = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)npair0) * 2u;
        const unsigned o_last = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)nlast) * 2u;
        const unsigned o_step = 16u * (unsigned)p.ld_out * 2u;
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.res + (HAS_RES ? (size_t)tile_m * BM * p.ld_res : 0)), 0,
            HAS_RES ? (int)min(rows_left * p.ld_res * 2, 0x7fffffffLL) : 0, 0x00020000);
        const unsigned r_col = ((unsigned)ml * (unsigned)p.ld_res + (unsigned)nbase) * 2u;
        const unsigned r_step = 16u * (unsigned)p.ld_res * 2u;
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
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int cl = (wn * TN + j * 16 + q4 * 4) * 4;
                const f32x4 svj = ld4(SCALE_OFF + cl), bvj = ld4(BIAS_OFF + cl);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] * svj[r] + bvj[r];
                    if ((PROF & 4) == 0 && p.act) t = silu_f32(t);
                    v[j][r] = t;
                }
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
            } else {
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                    unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                    auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                    const unsigned off = o_pair + (unsigned)i * o_step + (unsigned)(j * 32);
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
    auto epilogue = [&](int tile_m) __attribute__((always_inline)) {
        if (p.res) epilogue_t(tile_m, std::true_type{});
        else epilogue_t(tile_m, std::false_type{});
    };

    // ---- prologue: run (first tile, group 0, r 0) in buffer 0, weight slabs of steps 0 and 1 ----------
    run_setup();
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        if (i * NW + wave < A_PIECES)
            MDHIP_DMA16(a_rsrc, smem + (i * NW + wave) * 1024, jj < p.C8 ? q_off[i] + lg_abs : kOOB, 0);
    }
    run_next();
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            if ((B_PIECES % NW) != 0 && i == B_PER - 1 && wave >= B_PIECES % NW) continue;
            MDHIP_DMA16(b_rsrc, smem + B_OFF + st * B_BYTES + (i * NW + wave) * 1024, b_off[i], l_step * 128);
        }
        dma_b_done();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // two activation register sets (this step / next step), one weight set (columns are reloaded as they die)
    i32x8 xs[2][FM], w[FN];
    tile_masks(first_tile);
#pragma unroll
    for (int i = 0; i < FM; ++i) set_a_eff_one(0, 0, 0, i);
#pragma unroll
    for (int i = 0; i < FM; ++i) xs[0][i] = read_x(i);
#pragma unroll
    for (int j = 0; j < FN1; ++j) w[j] = read_w(0, j);

    int c_r = 0, c_cg = 0, c_tile = first_tile, pa = 0, step = 0;
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
    constexpr int FN2 = FN - FN1;
    constexpr int DMA_MAX = B_PER + A_H0, DMA_PER_G = (DMA_MAX + FN2 - 1) / FN2;
    // reads of the second half: FM activation fragments of the next step + FN1 weight columns, spread over its columns
    constexpr int RD2 = FM + FN1, RD2_PER_G = (RD2 + FN2 - 1) / FN2;

    // one step; XC = which activation register set holds this step's fragments and S = the tap inside the run are
    // compile-time, so that every register array is addressed statically
    auto step_body = [&](auto xc_t, auto s_t, const bool tile_end, const int n_r) __attribute__((always_inline)) {
        constexpr int XC = decltype(xc_t)::value;
        constexpr int s = decltype(s_t)::value;
        const int cur = step & 1;
        constexpr int ns = (s + 1) % 3;
        const int nbuf = s == 2 ? pa ^ 1 : pa;
        const int nr = s == 2 ? n_r : c_r;
        if (s == 2 && tile_end) tile_masks(c_tile + tile_step);
        // ---- first half: columns [0, FN1); the weight fragments of the other columns are read, the fragment
        //      addresses of the next step are selected ----
#pragma unroll
        for (int g = 0; g < FN1; ++g) {
#pragma unroll
            for (int j = FN1 + g; j < FN; j += FN1) w[j] = read_w(cur, j);
#pragma unroll
            for (int i = g; i < FM; i += FN1) set_a_eff_one(nbuf, nr, ns, i);
            MDHIP_FENCE();
#pragma unroll
            for (int i = 0; i < FM; ++i) MDHIP_MFMA8(w[g], xs[XC][i], acc[i][g]);
            MDHIP_FENCE();
        }
        stamp(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(1);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        MDHIP_FENCE();
        // ---- second half: the remaining columns; reads of the next step (activation fragments into the other
        //      register set, its first FN1 weight columns from stage cur ^ 1); DMA of step + 2 / the next run ----
#pragma unroll
        for (int g = 0; g < FN2; ++g) {
#pragma unroll
            for (int d = g * RD2_PER_G; d < (g + 1) * RD2_PER_G && d < RD2; ++d) {
                if (d < FM) xs[XC ^ 1][d] = read_x(d);
                else w[d - FM] = read_w(cur ^ 1, d - FM);
            }
            MDHIP_FENCE();
#pragma unroll
            for (int i = 0; i < FM; ++i) MDHIP_MFMA8(w[FN1 + g], xs[XC][i], acc[i][FN1 + g]);
            MDHIP_FENCE();
#pragma unroll
            for (int d = g * DMA_PER_G; d < (g + 1) * DMA_PER_G && d < DMA_MAX; ++d) {
                if (d < B_PER) dma_b_piece(cur, d);
                else if (s == 0 && d - B_PER < A_H0) dma_run_piece(pa ^ 1, d - B_PER);
                else if (s == 1 && A_H0 + d - B_PER < A_PER) dma_run_piece(pa ^ 1, A_H0 + d - B_PER);
            }
            MDHIP_FENCE();
        }
        dma_b_done();
        ++step;
        stamp(3);
    };
    // one run = three steps; PAR = the register set of the run's first step
    auto run_body = [&](auto par_t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_t)::value;
        const bool tile_end = c_r == 2 && c_cg == G - 1;
        const int n_r = c_r == 2 ? 0 : c_r + 1;
        step_body(std::integral_constant<int, PAR>{}, std::integral_constant<int, 0>{}, tile_end, n_r);
        step_body(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, 1>{}, tile_end, n_r);
        step_body(std::integral_constant<int, PAR>{}, std::integral_constant<int, 2>{}, tile_end, n_r);
        run_next();
        pa ^= 1;
        c_r = n_r;
        if (n_r == 0 && ++c_cg == G) {
            c_cg = 0;
            epilogue(c_tile);
            c_tile += tile_step;
        }
        stamp(5);
    };
    // Two runs per iteration, so that the register set of every step is fixed at compile time along ONE straight
    // path (a loop body that branches between the two parities makes the register allocator merge the two
    // assignments of every array at the loop header: hundreds of spilled VGPRs); an odd last run follows the loop.
    int run = 0;
    for (; run + 1 < total_runs; run += 2) {
        run_body(std::integral_constant<int, 0>{});
        run_body(std::integral_constant<int, 1>{});
    }
    if (run < total_runs) run_body(std::integral_constant<int, 0>{});
#undef MDHIP_FENCE
    if constexpr ((PROF & 1) != 0) {
        if (lane == 0 && p.dbg) {
            unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = t_acc[k];
            d[6] = (unsigned long long)step;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table
// ---------------------------------------------------------------------------------------
// id (local), BM, BN, waves along M, waves along N, PROF bits
#define MDHIP_CONV8_CFGS(X) \
    X(0, 128, 160, 2, 2, 0) \
    X(1, 192, 80, 4, 1, 0)  \
    X(2, 256, 160, 4, 2, 0) \
    X(3, 128, 80, 4, 1, 0)  \
    X(4, 64, 160, 2, 2, 0)  \
    X(5, 64, 80, 2, 1, 0)
// (PROF bits: 1 = s_memtime stamps, 2 = no stores, 4 = no SiLU, 16 = no LDS-DMA in the steady state)
#define MDHIP_CONV8_PROF(X) \
    X(6, 128, 160, 2, 2, 1)  \
    X(7, 128, 160, 2, 2, 16) \
    X(8, 128, 160, 2, 2, 2)  \
    X(9, 128, 160, 2, 2, 6)  \
    X(10, 128, 160, 2, 2, 22) \
    X(11, 256, 160, 4, 2, 1) \
    X(12, 256, 160, 4, 2, 16) \
    X(13, 256, 160, 4, 2, 2) \
    X(14, 256, 160, 4, 2, 22)

static const ConvCfg g_cfgs8[] = {
#define X(id, bm, bn, wm, wn, prof)                                                                   \
    {bm, bn, (wm) * (wn) * 64, (size_t)f8_lds_bytes(bm, bn), f8_blocks_per_cu(bm, bn, (wm) * (wn)), \
     "f8:run" #bm "x" #bn "/" #wm "x" #wn "/" #prof},
    MDHIP_CONV8_CFGS(X) MDHIP_CONV8_PROF(X)
#undef X
};
constexpr int kNumProf8 = 9;

int conv8_num_cfgs() { return (int)(sizeof(g_cfgs8) / sizeof(g_cfgs8[0])) - kNumProf8; }
const ConvCfg& conv8_cfg(int i) { return g_cfgs8[i]; }

hipError_t conv8_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, bn, wm, wn, prof)                                                              \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_f8_kernel<bm, bn, wm, wn, prof>,                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs8[id].lds_bytes);
    MDHIP_CONV8_CFGS(X) MDHIP_CONV8_PROF(X)
#undef X
    return e;
}

bool conv8_supports(int cfg, const ConvArgs& a) {
    if (cfg < 0 || cfg >= conv8_num_cfgs() + kNumProf8) return false;
    return a.in_f8 && !a.out_f8 && !a.out_f32 && a.wgt8 != nullptr && a.scale != nullptr && a.ntaps == 9 && a.kw == 3 &&
           a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && a.C8 >= 1 && (a.N % 8) == 0 &&
           ((long long)a.M + 2 * a.W + g_cfgs8[cfg < conv8_num_cfgs() ? cfg : 0].bm + 16) * a.ld_in < 0x7fffffffLL;   // one descriptor over the tensor
}

hipError_t conv8_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (!conv8_supports(cfg, a)) return hipErrorInvalidValue;
    const ConvCfg& c = g_cfgs8[cfg];
    ConvArgs p = a;
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, (32 * c.blocks_per_cu) / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    switch (cfg) {
#define X(id, bm, bn, wm, wn, prof)                                                               \
    case id:                                                                                    \
        hipLaunchKernelGGL((conv_f8_kernel<bm, bn, wm, wn, prof>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV8_CFGS(X) MDHIP_CONV8_PROF(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
