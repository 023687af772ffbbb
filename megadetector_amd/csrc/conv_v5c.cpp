// 3x3 / stride 1 convolution of the C = 80 -> N = 80 layers (the four bottleneck 3x3s of the x6 stack's first C3 block:
// M = 3.3 M pixels per batch of 32, K = 720) as its own kernel (gfx950 / MI355X), in conv_v5's kernel family: same K
// order (channel group, kernel row, tap, k-half), same MFMA chain per accumulator, same operand values -- bit-identical
// results.
//
// Why: conv_v5<192,80> runs these launches at 0.69 ms = 2.2 TB/s / 540 TFLOP/s -- 21 % of the matrix pipe and 28 % of
// HBM (s_memtime stamps: per 64-deep step 2330 cycles for 480 cycles of MFMA; issuing the weight and run DMA pieces and
// a 24 % epilogue that starts with the round trip of its residual).  The layer's whole weight tensor is 115 KB and its
// activations need 480 bytes of HBM traffic per output pixel, so, as for the stem (conv_v6.cpp):
//   * weights never move: wave (wm, wn) owns output channels wn*16 .. +15 and keeps the 18 weight fragments of the
//     64-channel group in registers (72 VGPRs); the 9 half-empty fragments of the 16-channel group sit in LDS (23 KB);
//   * a workgroup walks DOWN a column strip of the image: a tile is BM consecutive output pixels of one image row, its
//     three input rows are segments of BM + 2 pixels x 160 bytes in a ring of four LDS slots, and going from row y to
//     y + 1 brings in ONE new segment (LDS-DMA, landing while row y computes): every input pixel crosses L2 -> LDS once
//     per strip instead of three times per tile; out-of-image pixels are zero-filled by the DMA, so there are no tap
//     masks;
//   * the 160-byte pixel rows are conflict-free for ds_read_b128 as they are (40-dword stride), fragment addresses are
//     one base register per kernel row + an immediate: no address arithmetic in the loop;
//   * the residual of a tile is requested before its first MFMA, stores are exact-count buffer stores: one counted
//     vmcnt + one barrier per tile.
// Ten waves (2 pixel halves x 5 channel fragments), one workgroup per CU; per tile and wave 27 half steps x BM/32 MFMAs.
//
// Measured (batch 32, 320x320 maps, tools/convbench l2_3x3): 0.565 ms = 2.8 TB/s / 668 TFLOP/s against 0.676 ms for
// conv_v5<192,80> (160-pixel tiles = half an image row; 128-pixel tiles leave every third tile half empty at this
// width: 0.73 ms; five waves per workgroup, two workgroups per CU, 64-pixel tiles: 0.65 ms).  What bounds it now is the
// price of keeping the weights in registers: a wave owns ONE channel fragment, so every activation fragment is read
// from LDS by five waves -- one ds_read_b128 per MFMA, 1.44 MB per tile and CU (reads alone, everything else
// removed: 0.36 ms) -- and ten equal waves sit 3 / 3 / 2 / 2 on the four SIMDs.

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) char lds_char;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;

__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

constexpr int kPixB = 160;                                     // bytes per pixel (80 channels)
constexpr int c80_run_bytes(int bm) { return ((bm + 2) * kPixB + 1023) & ~1023; }
constexpr int kW1Bytes = 9 * 80 * 32;                          // the 16-channel group's weights: 9 taps x 80 rows x 32 B
// zero region read by the lanes that hold k 16..31 of the 16-channel group (base + the largest immediate of a fragment)
constexpr int c80_zero_bytes(int fm) { return ((15 + fm * 16 + 2) * kPixB + 32 + 255) & ~255; }      // (fm + 1 fragments: the fused kernel's conversion)
constexpr int c80_lds_bytes(int bm, int wm) { return 4 * c80_run_bytes(bm) + kW1Bytes + c80_zero_bytes(bm / (16 * wm)) + 320 + 1024; }
constexpr int c80f_lds_bytes(int bm, int wm) { return c80_lds_bytes(bm, wm) + 80 * 192 + 320; }
constexpr int c80_blocks(int bm, int wm) { return 163840 / c80_lds_bytes(bm, wm) >= 2 && wm == 1 ? 2 : 1; }

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// ConvArgs as set by the launcher: tiles_n = column strips per image row, tiles_per_xcd = row segments per strip,
// m_streams = rows per segment, tiles_m = units = images x segments x strips
// WM x 5 waves: wm = wave / 5 owns BM / WM pixels of the tile, wn = wave % 5 its 16 output channels
template <int BM, int WM>
__global__ void __launch_bounds__(WM * 5 * 64, c80_blocks(BM, WM))
conv_c80_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int kNWc = WM * 5;
    constexpr int FM = BM / (16 * WM);                         // fragment rows per wave
    constexpr int RUNB = c80_run_bytes(BM), PIECES = RUNB / 1024, NP = (PIECES + kNWc - 1) / kNWc;
    constexpr int W1_OFF = 4 * RUNB, ZERO_OFF = W1_OFF + kW1Bytes, BIAS_OFF = ZERO_OFF + c80_zero_bytes(FM);
    constexpr int SCRATCH_OFF = BIAS_OFF + 320;

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 5, wn = wave - 5 * wm;
    const int m15 = lane & 15, kb = lane >> 4;

    const int strips = p.tiles_n, segs = p.tiles_per_xcd, seg_rows = p.m_streams, total = p.tiles_m;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (total + 7) / 8;
    const int u_lo = xcd * per_xcd, u_hi = min(u_lo + per_xcd, total);
    if (u_lo + slot >= u_hi) return;

    // ---- once per workgroup: zero region, bias, the 16-channel group's weights ------------------------------------
    for (int c = tid * 16; c < c80_zero_bytes(FM); c += kNWc * 64 * 16)
        *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + c) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < 80; c += kNWc * 64) *(__attribute__((address_space(3))) float*)(smem + BIAS_OFF + c * 4) = p.bias[c];
    for (int c = tid; c < 9 * 80 * 2; c += kNWc * 64) {
        const int t = c / 160, rem = c - t * 160, ch = rem >> 1, half = rem & 1;
        *(__attribute__((address_space(3))) uint4*)(smem + W1_OFF + c * 16) =
            *(const uint4*)(p.wgt4 + (size_t)ch * p.k_pad4 + (9 + t) * 64 + half * 8);
    }
    // the 64-channel group's weight fragments of this wave's 16 channels: (tap, k-half) -> row wn*16 + m15, k = kb*8 ..
    frag8_t wreg[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            wreg[t][kk] = *(const frag8_t*)(p.wgt4 + (size_t)(wn * 16 + m15) * p.k_pad4 + t * 64 + kk * 32 + kb * 8);
    const f32x4 bias4 = {p.bias[wn * 16 + kb * 4], p.bias[wn * 16 + kb * 4 + 1], p.bias[wn * 16 + kb * 4 + 2],
                         p.bias[wn * 16 + kb * 4 + 3]};

    // ---- row loader: piece pc = wave + 10 k of a row segment; its lane handles 16-byte chunk g = pc*64 + lane,
    //      pixel g / 10 of the segment, channels (g % 10) * 8 ..; pieces beyond the segment land in a scratch KiB --------
    int l_px[NP];
    unsigned l_off[NP], l_dst[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pc = wave + kNWc * k;
        const int g = pc * 64 + lane;
        const int px = g / 10, c16 = g - px * 10;
        l_px[k] = (pc < PIECES && px < BM + 2) ? px : 0x40000000;              // never inside an image
        l_off[k] = (unsigned)((px * p.ld_in + c16 * 8) * 2);
        l_dst[k] = pc < PIECES ? (unsigned)(pc * 1024) : (unsigned)(SCRATCH_OFF - 0);
    }
    __amdgpu_buffer_rsrc_t in_rsrc, res_rsrc, out_rsrc;
    const int img_in_bytes = p.HoWo * p.ld_in * 2, img_out_bytes = p.HoWo * p.ld_out * 2, img_res_bytes = p.HoWo * p.ld_res * 2;
    int x0 = 0;
    // image row iy of the current strip into ring slot (iy + 1) & 3 (out-of-image rows and pixels: zeros)
    auto issue_row = [&](int iy) __attribute__((always_inline)) {
        const unsigned sl = (unsigned)(((iy + 1) & 3) * RUNB);
        const bool row_ok = (unsigned)iy < (unsigned)p.H;
        const unsigned row_term = (unsigned)((iy * p.W + x0 - 1) * p.ld_in * 2);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const bool ok = row_ok && (unsigned)(x0 - 1 + l_px[k]) < (unsigned)p.W;
            MDHIP_DMA16(in_rsrc, smem + (l_dst[k] >= (unsigned)SCRATCH_OFF ? l_dst[k] : sl + l_dst[k]), ok ? row_term + l_off[k] : kOOB, 0);
        }
    };

    // ---- fragment addresses: base of kernel row r (ring slot of image row y - 1 + r) + immediates -----------------------
    const unsigned lane_a = (unsigned)((wm * (BM / WM) + m15) * kPixB + kb * 16);
    const unsigned lane_a1 = kb < 2 ? lane_a + 128u : 0xffffffffu;       // 16-channel group: chunks 8, 9 of the pixel
    // (the same 160-byte pixel pitch and 16-byte chunk offset as the real rows: without the chunk term the lanes of
    // k-chunks 2 and 3 collide pairwise on their banks -- PMC: 21 % of the LDS cycles were conflicts)
    const unsigned lane_z = (unsigned)(ZERO_OFF + m15 * kPixB + (kb & 1) * 16);
    const unsigned lane_w1 = kb < 2 ? (unsigned)(W1_OFF + (wn * 16 + m15) * 32 + kb * 16) : (unsigned)ZERO_OFF;
    const unsigned lane_w1_step = kb < 2 ? 80u * 32u : 0u;

    f32x4 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    bool first = true;
    for (int u = u_lo + slot; u < u_hi; u += slots) {
        // unit -> (image, row segment, column strip)
        const int xs = u % strips;
        const int t2 = u / strips;
        const int sg = t2 % segs, b = t2 / segs;
        x0 = xs * BM;
        const int y_lo = sg * seg_rows, y_hi = min(y_lo + seg_rows, p.H);
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * p.HoWo * p.ld_in), 0, img_in_bytes, 0x00020000);
        out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.out + (size_t)b * img_out_bytes), 0, img_out_bytes, 0x00020000);
        if (p.res)
            res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (size_t)b * p.HoWo * p.ld_res), 0, img_res_bytes, 0x00020000);
        // every wave is past its fragment reads of the previous unit before the ring is refilled
        if (!first) __builtin_amdgcn_s_barrier();
        issue_row(y_lo - 1);
        issue_row(y_lo);
        issue_row(y_lo + 1);
        const int xw = x0 + wm * (BM / WM) + m15;                           // this lane's pixel column of fragment 0
        for (int y = y_lo; y < y_hi; ++y) {
            // rows y-1 .. y+1 have landed (everything but the FM stores of the previous tile), in every wave
            if (first || y == y_lo) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FM) : "memory");
            first = false;
            __builtin_amdgcn_s_barrier();
            if (y + 1 < y_hi) issue_row(y + 2);                               // into the slot row y - 2 has left
            // residual of this tile (8 bytes per lane and fragment row), long before the epilogue needs it
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 rres[FM];
            if (p.res) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int x = xw + i * 16;
                    const unsigned off = x < p.W ? (unsigned)(((y * p.W + x) * p.ld_res + wn * 16 + kb * 4) * 2) : kOOB;
                    rres[i] = __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, off, 0, 0);
                }
            }
            unsigned rb[3], rb1[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned s0 = (unsigned)(((y + r) & 3) * RUNB);             // slot of image row y - 1 + r
                rb[r] = s0 + lane_a;
                rb1[r] = kb < 2 ? s0 + lane_a1 : lane_z;
            }
            // ---- 27 half steps: (group 0: r, s, kk) then (group 1: r, s) ------------------------------------------
            // (ten waves per CU = three on a SIMD: the LDS round trip of one wave's fragments is covered by the MFMAs of
            // the other two, so the fragments are single-buffered and the registers go to the resident weights)
#pragma unroll
            for (int hs = 0; hs < 27; ++hs) {
                frag8_t xa[FM], w;
                if (hs < 18) {
                    const int t = hs >> 1, kk = hs & 1, r = t / 3, s = t - 3 * r;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
                        xa[i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + rb[r] + (i * 16 + s) * kPixB + kk * 64);
                    w = wreg[t][kk];
                } else {
                    const int t = hs - 18, r = t / 3, s = t - 3 * r;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
                        xa[i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + rb1[r] + (i * 16 + s) * kPixB);
                    w = *(const __attribute__((address_space(3))) frag8_t*)(smem + lane_w1 + t * lane_w1_step);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i] = MDHIP_MFMA(w, xa[i], acc[i]);
            }
            // ---- epilogue: bias, SiLU, residual, 16-bit, 8-byte stores (exactly FM per wave and tile) ----------------
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float v[4];
                mdhip_bias4(acc[i], bias4, v);
                if (p.act) mdhip_silu4(v);
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    v[0] += st_unpack((uint16_t)(rres[i][0] & 0xffff));
                    v[1] += st_unpack((uint16_t)(rres[i][0] >> 16));
                    v[2] += st_unpack((uint16_t)(rres[i][1] & 0xffff));
                    v[3] += st_unpack((uint16_t)(rres[i][1] >> 16));
                }
                const u32x2 d = {st_pack2(v[0], v[1]), st_pack2(v[2], v[3])};
                const int x = xw + i * 16;
                const unsigned off = x < p.W ? (unsigned)(((y * p.W + x) * p.ld_out + wn * 16 + kb * 4) * 2) : kOOB;
                __builtin_amdgcn_raw_buffer_store_b64(d, out_rsrc, off, 0, 0);
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// The whole bottleneck  out = x + SiLU(W2 * SiLU(W1 x + b1) + b2)  of the C = 80 block in one launch: the hidden tensor
// T = SiLU(W1 x + b1) (the output of the block's 1x1 conv, ConvArgs::wgt_pre / bias_pre) never leaves the CU.
//
// The strip walk above makes this possible without recomputing anything: a workgroup needs T row y + 2 exactly once,
// when it moves from output row y to y + 1.  So the x row segment of image row y + 2 is what the DMA brings in (one
// staging slot), the ten waves turn it into T with three K = 32 MFMA steps per fragment (1x1 weights: 12 VGPRs per
// wave), bias, SiLU, 16-bit rounding -- the 1x1 kernels' arithmetic and K order, so T has the bits the separate launch
// writes to HBM -- and store it (zero outside the image: the 3x3's padding applies to T) into the ring slot that T row
// y - 1 has just left.  Three ring slots + one staging slot = the four slots of the plain kernel.  Costs: + 13 % MFMAs,
// as many SiLUs again, one more barrier per tile; saves the 1x1 launch and 1.05 GB of HBM traffic per bottleneck.
// Measured at batch 32: 0.72 ms per bottleneck against 0.33 + 0.53 ms for the two launches; the step 34.64 -> 33.87 ms
// (same box, MDHIP_FUSE=0 / 1).  The 1x1's weights and bias live in LDS: in registers they spilled (168 per wave with
// ten waves on a CU), and a scratch reload per tile cost more than the whole conversion.
//
// x and out must not alias (a neighbouring strip reads this strip's border columns of x after this strip may have
// written them): the caller ping-pongs the block's two buffers.
// ---------------------------------------------------------------------------------------
template <int BM, int WM>
__global__ void __launch_bounds__(WM * 5 * 64, 1)
conv_c80f_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int kNWc = WM * 5;
    constexpr int FM = BM / (16 * WM);
    constexpr int RUNB = c80_run_bytes(BM), PIECES = RUNB / 1024, NP = (PIECES + kNWc - 1) / kNWc;
    constexpr int STAGE_OFF = 3 * RUNB;
    constexpr int W1_OFF = 4 * RUNB, ZERO_OFF = W1_OFF + kW1Bytes, BIAS_OFF = ZERO_OFF + c80_zero_bytes(FM);
    constexpr int SCRATCH_OFF = BIAS_OFF + 320;
    constexpr int WPRE_OFF = SCRATCH_OFF + 1024;              // the 1x1 conv's weights: 80 rows x 96 k x 2 B, then its bias
    constexpr int BPRE_OFF = WPRE_OFF + 80 * 192;
    constexpr int TF = (BM + 2 + 15) / 16;                     // fragments of a T row (BM + 2 pixels)
    constexpr int TFW = (TF + WM - 1) / WM;                    // ... per pixel group
    static_assert(((TFW - 1) * 16 + 15) * kPixB + 16 <= c80_zero_bytes(FM) + 0, "zero region covers the conversion's immediates");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 5, wn = wave - 5 * wm;
    const int m15 = lane & 15, kb = lane >> 4;

    const int strips = p.tiles_n, segs = p.tiles_per_xcd, seg_rows = p.m_streams, total = p.tiles_m;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (total + 7) / 8;
    const int u_lo = xcd * per_xcd, u_hi = min(u_lo + per_xcd, total);
    if (u_lo + slot >= u_hi) return;

    for (int c = tid * 16; c < c80_zero_bytes(FM); c += kNWc * 64 * 16)
        *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + c) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < 9 * 80 * 2; c += kNWc * 64) {
        const int t = c / 160, rem = c - t * 160, ch = rem >> 1, half = rem & 1;
        *(__attribute__((address_space(3))) uint4*)(smem + W1_OFF + c * 16) =
            *(const uint4*)(p.wgt4 + (size_t)ch * p.k_pad4 + (9 + t) * 64 + half * 8);
    }
    frag8_t wreg[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            wreg[t][kk] = *(const frag8_t*)(p.wgt4 + (size_t)(wn * 16 + m15) * p.k_pad4 + t * 64 + kk * 32 + kb * 8);
    // the 1x1 conv's weights (k = channel, 80 -> 96 = three 32-deep steps) and bias: LDS (registers go to the 3x3's)
    for (int c = tid; c < 80 * 12; c += kNWc * 64) {
        const int ch = c / 12, q = c - ch * 12;
        *(__attribute__((address_space(3))) uint4*)(smem + WPRE_OFF + c * 16) = *(const uint4*)(p.wgt_pre + (size_t)ch * p.k_pad_pre + q * 8);
    }
    for (int c = tid; c < 80; c += kNWc * 64) *(__attribute__((address_space(3))) float*)(smem + BPRE_OFF + c * 4) = p.bias_pre[c];
    const unsigned lane_wp = (unsigned)(WPRE_OFF + (wn * 16 + m15) * 192 + kb * 16);
    const f32x4 bias4 = {p.bias[wn * 16 + kb * 4], p.bias[wn * 16 + kb * 4 + 1], p.bias[wn * 16 + kb * 4 + 2],
                         p.bias[wn * 16 + kb * 4 + 3]};

    __amdgpu_buffer_rsrc_t in_rsrc, res_rsrc, out_rsrc;
    const int img_in_bytes = p.HoWo * p.ld_in * 2, img_out_bytes = p.HoWo * p.ld_out * 2, img_res_bytes = p.HoWo * p.ld_res * 2;
    int x0 = 0;
    // x row iy of the current strip into the staging slot (out-of-image rows and pixels: zeros, never used)
    auto issue_row = [&](int iy) __attribute__((always_inline)) {
        const bool row_ok = (unsigned)iy < (unsigned)p.H;
        const unsigned row_term = (unsigned)((iy * p.W + x0 - 1) * p.ld_in * 2);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            // (piece -> pixel, chunk recomputed here: three divisions per tile instead of nine registers for the whole kernel)
            const int pc = wave + kNWc * k;
            const int g = pc * 64 + lane;
            const int px = g / 10, c16 = g - px * 10;
            const bool ok = row_ok && pc < PIECES && px < BM + 2 && (unsigned)(x0 - 1 + px) < (unsigned)p.W;
            MDHIP_DMA16(in_rsrc, smem + (pc < PIECES ? STAGE_OFF + pc * 1024 : SCRATCH_OFF),
                        ok ? row_term + (unsigned)((px * p.ld_in + c16 * 8) * 2) : kOOB, 0);
        }
    };

    const unsigned lane_a = (unsigned)((wm * (BM / WM) + m15) * kPixB + kb * 16);
    const unsigned lane_a1 = kb < 2 ? lane_a + 128u : 0xffffffffu;
    // (the same 160-byte pixel pitch and 16-byte chunk offset as the real rows: without the chunk term the lanes of
    // k-chunks 2 and 3 collide pairwise on their banks -- PMC: 21 % of the LDS cycles were conflicts)
    const unsigned lane_z = (unsigned)(ZERO_OFF + m15 * kPixB + (kb & 1) * 16);
    const unsigned lane_w1 = kb < 2 ? (unsigned)(W1_OFF + (wn * 16 + m15) * 32 + kb * 16) : (unsigned)ZERO_OFF;
    const unsigned lane_w1_step = kb < 2 ? 80u * 32u : 0u;
    // conversion: this wave's pixel fragments of the staged row are wm*TFW .. (TF - 1 at most)
    const unsigned lane_x = (unsigned)(STAGE_OFF + (wm * TFW * 16 + m15) * kPixB + kb * 16);
    const unsigned lane_x2 = kb < 2 ? lane_x + 128u : lane_z;                    // k 64 .. 95: channels 64 .. 79, then zeros
    const unsigned lane_t = (unsigned)((wm * TFW * 16 + m15) * kPixB + (wn * 16 + kb * 4) * 2);

    f32x4 acc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // T row iy = SiLU(W1 x + b1) of the staged x row into ring slot `ts` (zero outside the image)
    auto convert = [&](int iy, int ts) __attribute__((always_inline)) {
        const bool row_ok = (unsigned)iy < (unsigned)p.H;
        frag8_t wp[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) wp[t] = *(const __attribute__((address_space(3))) frag8_t*)(smem + lane_wp + t * 64);
        const f32x4 bpre4 = *(const __attribute__((address_space(3))) f32x4*)(smem + BPRE_OFF + (wn * 16 + kb * 4) * 4);
#pragma unroll
        for (int f = 0; f < TFW; ++f) {
            if (wm * TFW + f < TF) {                                                  // wave-uniform
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const frag8_t xf = *(const __attribute__((address_space(3))) frag8_t*)(smem + (t < 2 ? lane_x + t * 64 : lane_x2) + f * 16 * kPixB);
                    c = MDHIP_MFMA(wp[t], xf, c);
                }
                const int px = (wm * TFW + f) * 16 + m15;
                const bool ok = row_ok && px < BM + 2 && (unsigned)(x0 - 1 + px) < (unsigned)p.W;
                float v[4];
                mdhip_bias4(c, bpre4, v);
                mdhip_silu4(v);
                uint2 d;
                d.x = ok ? st_pack2(v[0], v[1]) : 0u;
                d.y = ok ? st_pack2(v[2], v[3]) : 0u;
                if (px < BM + 2)
                    *(__attribute__((address_space(3))) uint2*)(smem + ts * RUNB + lane_t + f * 16 * kPixB) = d;
            }
        }
    };

    bool first = true;
    for (int u = u_lo + slot; u < u_hi; u += slots) {
        const int xs = u % strips;
        const int t2 = u / strips;
        const int sg = t2 % segs, b = t2 / segs;
        x0 = xs * BM;
        const int y_lo = sg * seg_rows, y_hi = min(y_lo + seg_rows, p.H);
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * p.HoWo * p.ld_in), 0, img_in_bytes, 0x00020000);
        out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.out + (size_t)b * img_out_bytes), 0, img_out_bytes, 0x00020000);
        if (p.res)
            res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (size_t)b * p.HoWo * p.ld_res), 0, img_res_bytes, 0x00020000);
        // ring slot of T row iy: (iy - y_lo + 1) % 3.  The unit's first three T rows: staged and converted one by one
        // (the previous unit's fragment reads are over: barrier)
        for (int k = 0; k < 3; ++k) {
            if (!first || k > 0) __builtin_amdgcn_s_barrier();               // staging slot and ring slot k are free
            first = false;
            issue_row(y_lo - 1 + k);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            convert(y_lo - 1 + k, k);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                           // the three T rows are complete
        if (y_lo + 1 < y_hi) issue_row(y_lo + 2);
        const int xw = x0 + wm * (BM / WM) + m15;
        int s0 = 0;                                                             // ring slot of T row y - 1
        for (int y = y_lo; y < y_hi; ++y) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 rres[FM];
            if (p.res) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int x = xw + i * 16;
                    const unsigned off = x < p.W ? (unsigned)(((y * p.W + x) * p.ld_res + wn * 16 + kb * 4) * 2) : kOOB;
                    rres[i] = __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, off, 0, 0);
                }
            }
            unsigned rb[3], rb1[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int sl = s0 + r;
                sl = sl >= 3 ? sl - 3 : sl;
                rb[r] = (unsigned)(sl * RUNB) + lane_a;
                rb1[r] = kb < 2 ? (unsigned)(sl * RUNB) + lane_a1 : lane_z;
            }
#pragma unroll
            for (int hs = 0; hs < 27; ++hs) {
                frag8_t xa[FM], w;
                if (hs < 18) {
                    const int t = hs >> 1, kk = hs & 1, r = t / 3, s = t - 3 * r;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
                        xa[i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + rb[r] + (i * 16 + s) * kPixB + kk * 64);
                    w = wreg[t][kk];
                } else {
                    const int t = hs - 18, r = t / 3, s = t - 3 * r;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
                        xa[i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + rb1[r] + (i * 16 + s) * kPixB);
                    w = *(const __attribute__((address_space(3))) frag8_t*)(smem + lane_w1 + t * lane_w1_step);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i] = MDHIP_MFMA(w, xa[i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float v[4];
                mdhip_bias4(acc[i], bias4, v);
                if (p.act) mdhip_silu4(v);
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    v[0] += st_unpack((uint16_t)(rres[i][0] & 0xffff));
                    v[1] += st_unpack((uint16_t)(rres[i][0] >> 16));
                    v[2] += st_unpack((uint16_t)(rres[i][1] & 0xffff));
                    v[3] += st_unpack((uint16_t)(rres[i][1] >> 16));
                }
                const u32x2 d = {st_pack2(v[0], v[1]), st_pack2(v[2], v[3])};
                const int x = xw + i * 16;
                const unsigned off = x < p.W ? (unsigned)(((y * p.W + x) * p.ld_out + wn * 16 + kb * 4) * 2) : kOOB;
                __builtin_amdgcn_raw_buffer_store_b64(d, out_rsrc, off, 0, 0);
            }
            if (y + 1 < y_hi) {
                // x row y + 2 is staged (older than this tile's FM stores); every wave is past its reads of T row y - 1
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FM) : "memory");
                __builtin_amdgcn_s_barrier();
                convert(y + 2, s0);                                              // T row y + 2 takes the slot of row y - 1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (y + 2 < y_hi) issue_row(y + 3);                              // the staging slot is free again
                s0 = s0 == 2 ? 0 : s0 + 1;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// The fused bottleneck, FOUR output rows per wave ("r4", round 6).  What bounds the kernel above is LDS bandwidth: with
// the weights in registers every MFMA reads its own activation fragment (one ds_read_b128 per MFMA: 1.7 MB through the
// LDS per 160-pixel row for 0.55 MB worth of MFMA time; counters: LDS pipe ~75 % busy, MFMA pipe 0.33).  A T row q is
// kernel row 2 of output row q - 1, row 1 of q and row 0 of q + 1, and the WEIGHTS of all three sit in this wave's
// registers -- so a wave that owns RW consecutive output rows of its pixels reads each fragment of T rows y - 1 ..
// y + RW once and uses it for up to three MFMAs: (RW + 2) / (3 RW) = 0.5 reads per MFMA at RW = 4.
//   * tile = RW = 4 output rows x BM = 64 pixels of a column strip; ten waves = 2 pixel halves (2 fragments each) x 5
//     channel fragments; accumulators 4 rows x 2 fragments;
//   * LDS: a ring of RW + 2 T rows (66 pixels x 160 B) + RW staging slots for the x rows of the NEXT tile (LDS-DMA, in
//     flight under this tile's main loop), converted to T rows y + 5 .. y + 8 between two barriers when the main loop has
//     left rows y - 1 .. y + 2;
//   * the loop runs T row by T row (not tap by tap): pass 1 the 64-channel group, pass 2 the 16-channel group, so every
//     accumulator still sees (group, kernel row, tap, k half) in the family's order: bit-identical to the kernel above;
//   * no zero region: the lanes that hold k 16..31 of the 16-channel group read the 32 bytes behind the pixel (the next
//     pixel's first channels: finite, every LDS byte is initialised) against weight fragments that are zero there.
// ---------------------------------------------------------------------------------------
constexpr int kD_BM = 64, kD_RW = 4, kD_RUNB = c80_run_bytes(kD_BM), kD_RING = kD_RW + 2;
constexpr int kD_STAGE_OFF = kD_RING * kD_RUNB, kD_W1_OFF = kD_STAGE_OFF + kD_RW * kD_RUNB, kD_ZERO_OFF = kD_W1_OFF + kW1Bytes;
constexpr int kD_WPRE_OFF = kD_ZERO_OFF + 256, kD_BPRE_OFF = kD_WPRE_OFF + 80 * 192, kD_LDS = kD_BPRE_OFF + 320;
static_assert(kD_LDS <= 163840, "LDS of the four-row fused bottleneck");
constexpr int kD_NB = 2, kD_RL = 4;                               // the shipped variant (see the template parameters)

// PROF 1 (developer variant, tools/convbench with CONVBENCH_FUSED=1): s_memtime per phase of a tile into ConvArgs::dbg
// NB: fragment buffers (steps st + 1 .. st + NB - 1 are in flight under step st); RL: first tap whose 64-channel weight
// fragments are dropped after pass 1 and fetched again after the epilogue
template <int PROF = 0, int NB = 4, int RL = 3>
__global__ void __launch_bounds__(640, 1)
conv_c80d_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = kD_BM, RW = kD_RW, RUNB = kD_RUNB, NRING = kD_RING, FMW = 2;
    constexpr int PIECES = RUNB / 1024, NP = (PIECES + 9) / 10;
    constexpr int STAGE_OFF = kD_STAGE_OFF, W1_OFF = kD_W1_OFF, ZERO_OFF = kD_ZERO_OFF, WPRE_OFF = kD_WPRE_OFF, BPRE_OFF = kD_BPRE_OFF;
    constexpr int TF = (BM + 2 + 15) / 16;                      // fragments of a T row (BM + 2 pixels)
    static_assert(RUNB % 1024 == 0 && (BM + 2) * kPixB <= RUNB, "row slot");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave / 5, wn = wave - 5 * g;                 // pixel half (main loop) / row pair (conversion), channel fragment
    const int m15 = lane & 15, kb = lane >> 4;

    const int strips = p.tiles_n, segs = p.tiles_per_xcd, seg_rows = p.m_streams, total = p.tiles_m;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (total + 7) / 8;
    const int u_lo = xcd * per_xcd, u_hi = min(u_lo + per_xcd, total);
    if (u_lo + slot >= u_hi) return;

    // ---- once per workgroup: every row slot zeroed (bytes no DMA / conversion ever writes are READ by the k 16..31 lanes),
    //      the zero chunk, both convs' LDS-resident weights --------------------------------------------------------------
    for (int c = tid * 16; c < W1_OFF; c += 640 * 16)
        *(__attribute__((address_space(3))) uint4*)(smem + c) = make_uint4(0, 0, 0, 0);
    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < 9 * 80 * 2; c += 640) {
        const int t = c / 160, rem = c - t * 160, ch = rem >> 1, half = rem & 1;
        *(__attribute__((address_space(3))) uint4*)(smem + W1_OFF + c * 16) =
            *(const uint4*)(p.wgt4 + (size_t)ch * p.k_pad4 + (9 + t) * 64 + half * 8);
    }
    for (int c = tid; c < 80 * 12; c += 640) {
        const int ch = c / 12, q = c - ch * 12;
        *(__attribute__((address_space(3))) uint4*)(smem + WPRE_OFF + c * 16) = *(const uint4*)(p.wgt_pre + (size_t)ch * p.k_pad_pre + q * 8);
    }
    for (int c = tid; c < 80; c += 640) *(__attribute__((address_space(3))) float*)(smem + BPRE_OFF + c * 4) = p.bias_pre[c];
    frag8_t wreg[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            wreg[t][kk] = *(const frag8_t*)(p.wgt4 + (size_t)(wn * 16 + m15) * p.k_pad4 + t * 64 + kk * 32 + kb * 8);
    const f32x4 bias4 = {p.bias[wn * 16 + kb * 4], p.bias[wn * 16 + kb * 4 + 1], p.bias[wn * 16 + kb * 4 + 2],
                         p.bias[wn * 16 + kb * 4 + 3]};

    __amdgpu_buffer_rsrc_t in_rsrc, res_rsrc, out_rsrc;
    const int img_in_bytes = p.HoWo * p.ld_in * 2, img_out_bytes = p.HoWo * p.ld_out * 2, img_res_bytes = p.HoWo * p.ld_res * 2;
    int x0 = 0;
    // x row iy of the current strip into staging slot st (out-of-image rows and pixels, and the lanes behind the segment: zeros)
    auto issue_row = [&](int iy, int st) __attribute__((always_inline)) {
        const bool row_ok = (unsigned)iy < (unsigned)p.H;
        const unsigned row_term = (unsigned)((iy * p.W + x0 - 1) * p.ld_in * 2);
        // (everything that depends only on the lane is worked out again at each use, from an opaque copy of the lane id: hoisted
        // out of the tile loop it would sit in registers across the main loop, which has none to spare)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int pc = wave + 10 * k;
            if (pc >= PIECES) continue;                                             // wave-uniform
            const int gch = pc * 64 + lane_o;
            const int px = gch / 10, c16 = gch - px * 10;
            const bool ok = row_ok && px < BM + 2 && (unsigned)(x0 - 1 + px) < (unsigned)p.W;
            MDHIP_DMA16(in_rsrc, smem + STAGE_OFF + st * RUNB + pc * 1024, ok ? row_term + (unsigned)((px * p.ld_in + c16 * 8) * 2) : kOOB, 0);
        }
    };

    // main loop: this lane's pixel of fragment 0 at tap shift 0 (slot pixel 0 = image column x0 - 1)
    const unsigned lane_a = (unsigned)((g * (BM / 2) + m15) * kPixB + kb * 16);
    const unsigned lane_w1 = kb < 2 ? (unsigned)(W1_OFF + (wn * 16 + m15) * 32 + kb * 16) : (unsigned)ZERO_OFF;
    const unsigned lane_w1_step = kb < 2 ? 80u * 32u : 0u;

    f32x4 acc[RW][FMW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
        for (int i = 0; i < FMW; ++i) acc[rr][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // T row iy = SiLU(W1 x + b1) of staging slot st into ring slot ts (zero outside the image): this wave's channel fragment,
    // all TF pixel fragments
    auto convert = [&](int iy, int st, int ts, int cf) __attribute__((always_inline)) {
        const bool row_ok = (unsigned)iy < (unsigned)p.H;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int m15 = lane_o & 15, kb = lane_o >> 4;
        // fragment f of a staged row / of the T row it becomes; cf = the channel fragment of this task
        const unsigned lane_x = (unsigned)(STAGE_OFF + m15 * kPixB + kb * 16);
        const unsigned lane_t = (unsigned)(m15 * kPixB + (cf * 16 + kb * 4) * 2);
        const unsigned lane_wp = (unsigned)(WPRE_OFF + (cf * 16 + m15) * 192 + kb * 16);
        frag8_t wp[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) wp[t] = *(const __attribute__((address_space(3))) frag8_t*)(smem + lane_wp + t * 64);
        const f32x4 bpre4 = *(const __attribute__((address_space(3))) f32x4*)(smem + BPRE_OFF + (cf * 16 + kb * 4) * 4);
        // Every fragment's reads first, then the MFMAs (k step outer: TF independent chains), then the SiLUs, then the writes.
        // Fragment by fragment -- read, three dependent MFMAs, SiLU, ds_write, next fragment's read -- the LDS write of one
        // fragment orders the reads of the next behind it (both are LDS: the compiler must assume they alias), and a task was
        // five serial round trips: 500 cycles per fragment in the stamps.
        frag8_t xf[TF][3];
#pragma unroll
        for (int f = 0; f < TF; ++f)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                // (t = 2: channels 64 .. 79, then the 32 bytes behind the pixel against zero weights)
                xf[f][t] = *(const __attribute__((address_space(3))) frag8_t*)(smem + lane_x + st * RUNB + f * 16 * kPixB + t * 64);
        f32x4 c[TF];
#pragma unroll
        for (int f = 0; f < TF; ++f) c[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int f = 0; f < TF; ++f) c[f] = MDHIP_MFMA(wp[t], xf[f][t], c[f]);
        uint2 d[TF];
#pragma unroll
        for (int f = 0; f < TF; ++f) {
            const int px = f * 16 + m15;
            const bool ok = row_ok && px < BM + 2 && (unsigned)(x0 - 1 + px) < (unsigned)p.W;
            float v[4];
            mdhip_bias4(c[f], bpre4, v);
            mdhip_silu4(v);
            d[f].x = ok ? st_pack2(v[0], v[1]) : 0u;
            d[f].y = ok ? st_pack2(v[2], v[3]) : 0u;
        }
#pragma unroll
        for (int f = 0; f < TF; ++f) {
            const int px = f * 16 + m15;
            if (px < BM + 2)
                *(__attribute__((address_space(3))) uint2*)(smem + ts * RUNB + lane_t + f * 16 * kPixB) = d[f];
        }
    };
    auto ring_slot = [&](int i) __attribute__((always_inline)) -> int { return i >= NRING ? i - NRING : i; };
    // The conversion of RW rows is 5 RW tasks (row, channel fragment).  Ten waves sit 3 / 3 / 2 / 2 on the four SIMDs (wave w on
    // SIMD w mod 4), so the waves of the two-wave SIMDs take three and two tasks, those of the three-wave SIMDs two, two and one:
    // five tasks per SIMD.  First task of wave w: 5 bits each.
    constexpr unsigned long long kTask0 = 0ull | (2ull << 5) | (4ull << 10) | (7ull << 15) | (10ull << 20) | (12ull << 25) | (14ull << 30) |
                                          (16ull << 35) | (18ull << 40) | (19ull << 45) | (20ull << 50);
    static_assert(RW == 4, "task table of the conversion");
    const int task_lo = (int)((kTask0 >> (5 * wave)) & 31), task_hi = (int)((kTask0 >> (5 * wave + 5)) & 31);
    // rows y_first .. y_first + RW - 1 of the staging slots 0 .. RW - 1 into the ring slots slot0 + row
    auto convert_rows = [&](int y_first, int slot0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int id = task_lo + k;
            if (id < task_hi) {                                                    // wave-uniform
                const int row = id / 5, cf = id - 5 * row;
                convert(y_first + row, row, ring_slot(slot0 + row), cf);
            }
        }
    };
    // (PROF) slots: 0 pass 1, 1 pass 2, 2 epilogue + weight reload, 3 wait + barrier, 4 conversion, 5 barrier + DMA issue / unit start-up
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0, n_tiles = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr (PROF != 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[k] += t - t_prev;
            t_prev = t;
        }
    };
    if constexpr (PROF != 0) t_prev = __builtin_amdgcn_s_memtime();

    bool first = true;
    for (int u = u_lo + slot; u < u_hi; u += slots) {
        const int xs = u % strips;
        const int t2 = u / strips;
        const int sg = t2 % segs, b = t2 / segs;
        x0 = xs * BM;
        const int y_lo = sg * seg_rows, y_hi = min(y_lo + seg_rows, p.H);
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)b * p.HoWo * p.ld_in), 0, img_in_bytes, 0x00020000);
        out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.out + (size_t)b * img_out_bytes), 0, img_out_bytes, 0x00020000);
        // (no residual: an empty descriptor -- the loads below are issued either way and return zeros: a branch around them inside
        // the unrolled step loop made the compiler duplicate the rest of the loop and keep the fragment buffers in scratch)
        res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res + (size_t)b * p.HoWo * p.ld_res : p.in), 0,
                                                     p.res ? img_res_bytes : 0, 0x00020000);
        // ---- the unit's first RW + 2 T rows (y_lo - 1 .. y_lo + RW): RW staged and converted, then two more; ring slot of
        //      T row q of the unit: (q - y_lo + 1) mod NRING ----------------------------------------------------------------
        if (first) {                                                              // (the zeroing above is complete in every wave)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            first = false;
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) issue_row(y_lo - 1 + k, k);                  // (staging is free: the last conversion of a unit ends with a barrier)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                             // staged; every wave has left the previous unit's ring
        convert_rows(y_lo - 1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_row(y_lo - 1 + RW, 0);
        issue_row(y_lo + RW, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        convert(y_lo - 1 + RW + g, g, RW + g, wn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                             // ring complete, staging free
        if (y_lo + RW < y_hi) {
#pragma unroll
            for (int k = 0; k < RW; ++k) issue_row(y_lo + RW + 1 + k, k);         // x rows of the next tile's new T rows
        }
        int s0 = 0;                                                               // ring slot of T row y - 1
        stamp(5);
        for (int y = y_lo; y < y_hi; y += RW) {
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 rres[RW][FMW];
            ++n_tiles;
            // The loop runs T row by T row: step st < NS1 = (T row qi, tap shift s, k half kk) of the 64-channel group (pass 1),
            // then (qi, s) of the 16-channel group (pass 2); output row rr takes T row qi as its kernel row r = qi - rr.  The
            // fragments of step st + 1 are read while the MFMAs of step st run (two buffers, fenced: left alone the compiler
            // clusters dozens of reads ahead and spills the resident weights).
            constexpr int NS1 = (RW + 2) * 6, NS2 = (RW + 2) * 3, NS = NS1 + NS2;
            frag8_t xq[NB][FMW];
            frag8_t w1[9];                                                            // pass 2: the 16-channel group's weight fragments (in kernel rows 1, 2's registers)
            auto load_x = [&](int st) __attribute__((always_inline)) {
                if (st < NS1) {
                    const int qi = st / 6, s = (st % 6) >> 1, kk = st & 1;
                    const unsigned base = (unsigned)(ring_slot(s0 + qi) * RUNB) + lane_a;
#pragma unroll
                    for (int i = 0; i < FMW; ++i)
                        xq[st % NB][i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + base + (i * 16 + s) * kPixB + kk * 64);
                } else {
                    const int qi = (st - NS1) / 3, s = (st - NS1) % 3;
                    const unsigned base = (unsigned)(ring_slot(s0 + qi) * RUNB) + lane_a + 128u;
#pragma unroll
                    for (int i = 0; i < FMW; ++i)
                        xq[st % NB][i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + base + (i * 16 + s) * kPixB);
                }
            };
#pragma unroll
            for (int st = 0; st < NB - 1; ++st) load_x(st);
#pragma unroll
            for (int st = 0; st < NS1; ++st) {                                            // pass 1
                load_x(st + NB - 1);
                __builtin_amdgcn_sched_barrier(0);
                const int qi = st / 6, s = (st % 6) >> 1, kk = st & 1;
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    const int r = qi - rr;
                    if (r < 0 || r > 2) continue;
#pragma unroll
                    for (int i = 0; i < FMW; ++i) acc[rr][i] = MDHIP_MFMA(wreg[r * 3 + s][kk], xq[st % NB][i], acc[rr][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(0);
            // pass 2 begins: its nine weight fragments, and the residual of the tile (8 bytes per lane and fragment)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                w1[t] = *(const __attribute__((address_space(3))) frag8_t*)(smem + lane_w1 + t * lane_w1_step);
            {
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const int xw = x0 + g * (BM / 2) + (lane_o & 15), kb = lane_o >> 4;       // this lane's pixel column of fragment 0
#pragma unroll
                for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                    for (int i = 0; i < FMW; ++i) {
                        const int x = xw + i * 16;
                        const unsigned off = (x < p.W && y + rr < y_hi) ? (unsigned)((((y + rr) * p.W + x) * p.ld_res + wn * 16 + kb * 4) * 2) : kOOB;
                        rres[rr][i] = __builtin_amdgcn_raw_buffer_load_b64(res_rsrc, off, 0, 0);
                    }
            }
            // ---- epilogue of fragment (rr, i): bias, SiLU, residual, 16-bit, one 8-byte store (exactly RW x FMW per wave and tile).
            //      Output row rr is complete when T row rr + 2 is through: its two fragments are finished behind the MFMAs of the
            //      first two steps of T row rr + 3 (VALU work in the shadow of this wave's own matrix instructions), the last row
            //      behind the loop ------------------------------------------------------------------------------------------------
            auto epilogue_frag = [&](int rr, int i) __attribute__((always_inline)) {
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int xw = x0 + g * (BM / 2) + (lane_e & 15), kb = lane_e >> 4;
                float v[4];
                mdhip_bias4(acc[rr][i], bias4, v);
                if (p.act) mdhip_silu4(v);
                acc[rr][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                // (no residual: the loads returned zeros from the empty descriptor)
                v[0] += st_unpack((uint16_t)(rres[rr][i][0] & 0xffff));
                v[1] += st_unpack((uint16_t)(rres[rr][i][0] >> 16));
                v[2] += st_unpack((uint16_t)(rres[rr][i][1] & 0xffff));
                v[3] += st_unpack((uint16_t)(rres[rr][i][1] >> 16));
                const u32x2 d = {st_pack2(v[0], v[1]), st_pack2(v[2], v[3])};
                const int x = xw + i * 16;
                const unsigned off = (x < p.W && y + rr < y_hi) ? (unsigned)((((y + rr) * p.W + x) * p.ld_out + wn * 16 + kb * 4) * 2) : kOOB;
                __builtin_amdgcn_raw_buffer_store_b64(d, out_rsrc, off, 0, 0);
            };
#pragma unroll
            for (int st = NS1; st < NS; ++st) {                                           // pass 2
                if (st + NB - 1 < NS) load_x(st + NB - 1);
                __builtin_amdgcn_sched_barrier(0);
                const int qi = (st - NS1) / 3, s = (st - NS1) % 3;
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    const int r = qi - rr;
                    if (r < 0 || r > 2) continue;
#pragma unroll
                    for (int i = 0; i < FMW; ++i) acc[rr][i] = MDHIP_MFMA(w1[r * 3 + s], xq[st % NB][i], acc[rr][i]);
                }
                if (qi >= 3 && s < FMW) epilogue_frag(qi - 3, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(1);
#pragma unroll
            for (int i = 0; i < FMW; ++i) epilogue_frag(RW - 1, i);
            // The weight fragments of kernel rows 1 and 2 (48 registers) are dead from the end of pass 1 and fetched again here (L2
            // hits, landing under the conversion below): pass 2 and the epilogue get their registers -- the 16-channel group's nine
            // weight fragments (read from LDS once per tile instead of once per use) and the residual -- instead of spilling.
            {
                asm volatile("" ::: "memory");
                int lane_w = lane;
                asm volatile("" : "+v"(lane_w));
                const uint16_t* wrow = p.wgt4 + (size_t)(wn * 16 + (lane_w & 15)) * p.k_pad4 + (lane_w >> 4) * 8;
#pragma unroll
                for (int t = RL; t < 9; ++t)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) wreg[t][kk] = *(const frag8_t*)(wrow + t * 64 + kk * 32);
            }
            stamp(2);
            if (y + RW < y_hi) {
                // the staged x rows (older than this tile's RW x FMW stores and the twelve weight loads) have landed; every wave is
                // past its reads of T rows y - 1 .. y + RW - 2
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(RW * FMW + 2 * (9 - RL)) : "memory");
                __builtin_amdgcn_s_barrier();
                stamp(3);
                convert_rows(y + RW + 1, s0);                                      // T rows y + RW + 1 .. y + 2 RW take the slots of y - 1 .. y + RW - 2
                stamp(4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                                     // the staging slots are free again
                if (y + 2 * RW < y_hi) {
#pragma unroll
                    for (int k = 0; k < RW; ++k) issue_row(y + 2 * RW + 1 + k, k);
                }
                s0 = ring_slot(s0 + RW);
                stamp(5);
            }
        }
    }
    if constexpr (PROF != 0) {
        if (lane == 0 && p.dbg) {
            unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * 10 + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = t_acc[k];
            d[6] = n_tiles;
            d[7] = (unsigned long long)__builtin_amdgcn_s_getreg(((6 - 1) << 11) | (0 << 6) | 4) | 0x100;   // HW_REG_HW_ID[5:0]: wave slot [3:0], SIMD [5:4]
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// (Round 4, built, measured, removed -- commit history has it: the fused bottleneck with TWO channel fragments per wave
// (four waves with 144 weight registers and 5 x 2 accumulator fragments + four one-fragment quarter waves: 30 instead
// of 50 fragment reads per half step, 13 / 13 / 12 / 12 MFMAs per SIMD).  Bit-identical, and 11 % SLOWER (0.83 against
// 0.745 ms per bottleneck, profiles/r4_c80g_two_fragments.txt): with the weights of two fragments resident a wave has no
// register left to read the next half step's activation fragments ahead, so every half step exposes its LDS round
// trip, and two such waves per SIMD hide less of it than the three light ones above.  The kernel is bound by LDS
// latency at its occupancy, not by LDS bandwidth.)
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// configuration table (ids local to this file; conv_v5.cpp appends them to its own)
// ---------------------------------------------------------------------------------------
// id, BM (pixels of an image row per tile), pixel groups (x 5 channel fragments = waves)
#define MDHIP_CONV5C_CFGS(X) \
    X(0, 160, 2)             \
    X(1, 128, 2)
// id 2: the four-row fused bottleneck (conv_c80d_kernel); as a plain 3x3 (fusion off) it launches id 0's strip kernel -- same bits
constexpr int kCfgR4 = 2;

static const ConvCfg g_cfgs5c[] = {
#define X(id, bm, wm) {bm, 80, (wm) * 5 * 64, (size_t)c80_lds_bytes(bm, wm), c80_blocks(bm, wm), "v5:strip" #bm "x80/" #wm "x5"},
    MDHIP_CONV5C_CFGS(X)
#undef X
    {kD_BM, 80, 640, (size_t)kD_LDS, 1, "v5:strip64x80/2x5/r4"},
    // (phase stamps; run only with ConvArgs::dbg set: tools/convbench)
    {kD_BM, 80, 640, (size_t)kD_LDS, 1, "dev:strip64x80/2x5/r4/nb2rl4"},
    {kD_BM, 80, 640, (size_t)kD_LDS, 1, "dev:strip64x80/2x5/r4/nb2rl3"},
    {kD_BM, 80, 640, (size_t)kD_LDS, 1, "dev:strip64x80/2x5/r4/nb2rl5"},
    {kD_BM, 80, 640, (size_t)kD_LDS, 1, "dev:strip64x80/2x5/r4/nb3rl3"},
    {kD_BM, 80, 640, (size_t)kD_LDS, 1, "dev:strip64x80/2x5/r4/nb4rl3"},
};


int conv5c_num_cfgs() { return (int)(sizeof(g_cfgs5c) / sizeof(g_cfgs5c[0])); }
const ConvCfg& conv5c_cfg(int i) { return g_cfgs5c[i]; }

hipError_t conv5c_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, wm)                                                                            \
    if (e == hipSuccess)                                                                         \
        e = hipFuncSetAttribute((const void*)conv_c80_kernel<bm, wm>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)g_cfgs5c[id].lds_bytes);                                       \
    if (e == hipSuccess)                                                                         \
        e = hipFuncSetAttribute((const void*)conv_c80f_kernel<bm, wm>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                c80f_lds_bytes(bm, wm));

    MDHIP_CONV5C_CFGS(X)
#undef X
#define MDHIP_C80D_VARIANTS(X) X(0, 1, 2, 4) X(1, 1, 2, 3) X(2, 1, 2, 5) X(3, 1, 3, 3) X(4, 1, 4, 3)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_c80d_kernel<0, kD_NB, kD_RL>, hipFuncAttributeMaxDynamicSharedMemorySize, kD_LDS);
#define X(id, prof, nb, rl) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_c80d_kernel<prof, nb, rl>, hipFuncAttributeMaxDynamicSharedMemorySize, kD_LDS);
    MDHIP_C80D_VARIANTS(X)
#undef X
    return e;
}

// the caller (conv5_supports) has checked the shape conditions common to the family
// (a.wgt_pre != nullptr: the fused bottleneck -- in and out must be different tensors, the 1x1 has 80 input channels)
bool conv5c_supports(int cfg, const ConvArgs& a) {
    if (a.wgt_pre != nullptr && (a.bias_pre == nullptr || a.k_pad_pre < 96 || (const void*)a.in == (const void*)a.out || a.act != 1))
        return false;
    if (cfg > kCfgR4 && (a.dbg == nullptr || a.wgt_pre == nullptr)) return false;            // developer variants
    return cfg >= 0 && cfg < conv5c_num_cfgs() && !a.out_f32 && !a.out_f8 && !a.in_f8 && a.C8 == 10 && a.groups == 2 &&
           a.N == 80 && a.n_rows == 80 && (long long)a.HoWo * a.ld_in * 2 < 0x3fffffffLL &&
           (long long)a.HoWo * a.ld_out * 2 < 0x3fffffffLL && (a.res == nullptr || (long long)a.HoWo * a.ld_res * 2 < 0x3fffffffLL);
}

hipError_t conv5c_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    const int dev = cfg > kCfgR4 ? cfg - kCfgR4 - 1 : -1;
    if (dev >= 0) cfg = kCfgR4;
    if (cfg == kCfgR4 && !a.wgt_pre) cfg = 0;                   // (not fused: the strip kernel, same family)
    const ConvCfg& c = g_cfgs5c[cfg];
    ConvArgs p = a;
    const int n_img = a.M / a.HoWo;
    if (cfg == kCfgR4) {
        // units = (image, row segment, 64-pixel column strip); rows per segment: the multiple of four that minimises
        // (units per workgroup, rounded up) x (rows + the unit's start-up, ~ 3 rows' worth)
        const int strips = (a.W + kD_BM - 1) / kD_BM;
        int best_rows = 4;
        double best_cost = 1e30;
        for (int sr = 4; sr <= ((a.H + 3) / 4) * 4; sr += 4) {
            const long long units = (long long)n_img * strips * ((a.H + sr - 1) / sr);
            const double cost = (double)((units + 255) / 256) * (sr + 3.0);
            if (cost < best_cost) { best_cost = cost; best_rows = sr; }
        }
        p.tiles_n = strips;
        p.tiles_per_xcd = (a.H + best_rows - 1) / best_rows;
        p.m_streams = best_rows;
        p.tiles_m = n_img * p.tiles_per_xcd * strips;
        const int slots = std::max(1, std::min(32, (p.tiles_m + 7) / 8));
        const dim3 grid_d((unsigned)(8 * slots));
        switch (dev) {
#define X(id, prof, nb, rl) case id: hipLaunchKernelGGL((conv_c80d_kernel<prof, nb, rl>), grid_d, dim3(640), kD_LDS, s, p); break;
            MDHIP_C80D_VARIANTS(X)
#undef X
            default: hipLaunchKernelGGL((conv_c80d_kernel<0, kD_NB, kD_RL>), grid_d, dim3(640), kD_LDS, s, p); break;
        }
        return hipGetLastError();
    }
    const int strips = (a.W + c.bm - 1) / c.bm;
    const int wgs = 256 * c.blocks_per_cu;
    // rows per segment: ~4 units per workgroup when the batch allows it (the first tile of a unit waits for three row
    // segments instead of one), at least 8 rows
    // (1, 2, 4 or 8 units per workgroup: the same step time to 0.3 %)
    int seg_rows = (int)std::max(8LL, std::min<long long>(a.H, ((long long)n_img * strips * a.H + 4 * wgs - 1) / (4 * wgs)));
    const int segs = (a.H + seg_rows - 1) / seg_rows;
    p.tiles_n = strips;
    p.tiles_per_xcd = segs;
    p.m_streams = seg_rows;
    p.tiles_m = n_img * segs * strips;
    const int slots = std::max(1, std::min(32 * c.blocks_per_cu, (p.tiles_m + 7) / 8));
    const dim3 grid((unsigned)(8 * slots));
    switch (cfg) {
#define X(id, bm, wm)                                                                             \
    case id:                                                                                      \
        if (a.wgt_pre) hipLaunchKernelGGL((conv_c80f_kernel<bm, wm>), grid, dim3((wm) * 5 * 64), c80f_lds_bytes(bm, wm), s, p); \
        else hipLaunchKernelGGL((conv_c80_kernel<bm, wm>), grid, dim3((wm) * 5 * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV5C_CFGS(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
