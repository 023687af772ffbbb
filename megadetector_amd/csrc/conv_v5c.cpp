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

static const ConvCfg g_cfgs5c[] = {
#define X(id, bm, wm) {bm, 80, (wm) * 5 * 64, (size_t)c80_lds_bytes(bm, wm), c80_blocks(bm, wm), "v5:strip" #bm "x80/" #wm "x5"},
    MDHIP_CONV5C_CFGS(X)
#undef X
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
    return e;
}

// the caller (conv5_supports) has checked the shape conditions common to the family
// (a.wgt_pre != nullptr: the fused bottleneck -- in and out must be different tensors, the 1x1 has 80 input channels)
bool conv5c_supports(int cfg, const ConvArgs& a) {
    if (a.wgt_pre != nullptr && (a.bias_pre == nullptr || a.k_pad_pre < 96 || (const void*)a.in == (const void*)a.out || a.act != 1))
        return false;
    return cfg >= 0 && cfg < conv5c_num_cfgs() && !a.out_f32 && !a.out_f8 && !a.in_f8 && a.C8 == 10 && a.groups == 2 &&
           a.N == 80 && a.n_rows == 80 && (long long)a.HoWo * a.ld_in * 2 < 0x3fffffffLL &&
           (long long)a.HoWo * a.ld_out * 2 < 0x3fffffffLL && (a.res == nullptr || (long long)a.HoWo * a.ld_res * 2 < 0x3fffffffLL);
}

hipError_t conv5c_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    const ConvCfg& c = g_cfgs5c[cfg];
    ConvArgs p = a;
    const int n_img = a.M / a.HoWo;
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
