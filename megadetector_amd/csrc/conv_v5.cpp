// 3x3 / stride 1 convolution with *row-segment reuse* across the three taps of a kernel row
// (gfx950 / MI355X).
//
// Why: the implicit-GEMM kernels gather every input pixel nine times through the L2 -> LDS path, which
// is what bounds them.  An M tile is BM *consecutive* output pixels in raster
// order, so for kernel row r the three taps (r,0) (r,1) (r,2) read three windows of the same
// contiguous run of BM+2 input pixels, one pixel apart.  This kernel loads that run ONCE per
// (64-channel group, kernel row) into an LDS buffer and serves the three taps from it by reading the
// A fragments at a row shift of s: activation traffic / 2.95, 103 FLOP per L2->LDS byte for a
// 128x160 tile instead of 71 -- in the two-workgroups-per-CU structure of conv_v2 (the row-patch
// kernel conv_v4 reuses more but needs the whole CU for one lock-stepped workgroup).
//
// The run wraps around image-row ends (and image ends inside a batch): a tap that falls outside the
// image for some output pixel must contribute zero.  That is not a property of the LDS image any
// more, so it is handled at the fragment read: per tile every lane keeps a 9-bit tap-validity mask
// for each of its FM pixels, and an invalid (pixel, tap) reads a 128-byte row of zeros instead
// (one v_cndmask on the LDS address per fragment, no branches).
//
// K order is (channel group, r, s, channel in group) -- the second weight packing of conv_v4 -- so
// results equal the row-patch kernel's and differ from the implicit-GEMM kernels by fp32 summation
// order only.
//
// Schedule per step (= one tap of one channel group, 64 deep), as conv_v2:
//     read  Y  (k 32..63: run buffer @ shift s, weight stage cur)   } interleaved
//     mfma  X  (k 0..31)                                            }
//     s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier
//     DMA   weight slab of step+2 -> stage cur ; in steps s = 0, 1: pieces of the NEXT run
//     read  X' (k 0..31 of step+1)                                  } interleaved
//     mfma  Y                                                       }
//
// The two waves of a SIMD run this between the same barriers, so whatever sits between two MFMA chunks is
// matrix-pipe idle time: a DMA piece is issued with 3-4 instructions (weights: lane offset fixed, step and
// piece in the scalar offset; run pieces: addressed from the tensor's first byte, the descriptor's range check
// zeroes what lies outside the batch), the tap select of the next step is pinned in the first half, and the
// epilogue goes through buffer instructions with one 32-bit offset per lane (DESIGN.md section 5 [r3b]).

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) char lds_char;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;

__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
typedef mdhip_f32x2 f32x2;

constexpr int v5_run_pieces(int bm) { return (bm + 2 + 7) / 8; }
// the row of zeros (256 bytes) and, behind it, the staged bias of the workgroup's BN channels: whole KiB
constexpr int v5_zero_bytes(int bn) { return (256 + bn * 4 + 1023) / 1024 * 1024; }
// 2 run buffers + 2 weight stages + the zero row / bias area
constexpr int v5_lds_bytes(int bm, int bn) { return 2 * v5_run_pieces(bm) * 1024 + 2 * bn * 128 + v5_zero_bytes(bn); }
constexpr int v5_blocks_per_cu(int bm, int bn, int nw) {
    int b = 163840 / v5_lds_bytes(bm, bn);
    if (b > 8 / nw) b = 8 / nw;          // two waves per SIMD (256 registers each)
    return b < 1 ? 1 : b;
}
constexpr bool v5_is_lean(int bm, int bn, int wm, int wn) { return bm / wm == 80 && bn / wn == 80; }
// aligned mode (conv_v5_kernel AL): zero rows at ZERO_OFF + i * 2048, i = 1..4, behind the zero row / bias area
constexpr int v5_al_extra_lds = 4 * 2048;
constexpr int v5_waves_per_simd(int bm, int bn, int nw) {
    int w = v5_blocks_per_cu(bm, bn, nw) * nw / 4;
    return w < 1 ? 1 : w;
}

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// PROF bits (developer builds only): 1 = s_memtime stamps, 2 = no stores, 4 = no SiLU, 16 = no DMA in the steady state,
// 32 = no tap-validity selects (timing experiment: wrong at image borders), 64 = no paired / short last group
// TAIL: what the last channel group looks like, fixed at compile time so that the steps of the full groups carry no
// test for it (every instruction between two MFMA chunks is matrix-pipe idle time in the lock-stepped tiles; the run-time
// tests cost 4 % on the 320-channel layers, which have no tail at all): 0 = every group is full (or more than half full),
// 1 = a last group of <= 32 channels (its k 32..63 MFMAs are skipped), 2 = that group with paired taps,
// -1 = decided at run time (developer variants)
// AL ("aligned", 8-wave tiles only): the 80 pixels of a wave lie inside ONE image row (W a multiple of 80).  Tap validity
// is then a handful of per-wave scalars -- top / bottom row, first / last pixel of the row in fragment 0 / 4 -- instead of a
// 9-bit mask per lane and fragment: one address register for fragments 1..3 (fragment i at the immediate offset i * 2048, a
// row of zeros at each of those offsets for a wave whose tap falls above / below the image), two for the edge fragments;
// 5-6 VALU instructions a step instead of ~36 (a select per fragment, the shift arithmetic, the k-half XOR per fragment).
template <int BM, int BN, int WM, int WN, int PROF = 0, int TAIL = -1, bool AL = false>
__global__ void __launch_bounds__(WM * WN * 64, v5_waves_per_simd(BM, BN, WM * WN))
conv_v5_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_PIECES = v5_run_pieces(BM), A_BUF = A_PIECES * 1024;
    constexpr int A_PER = (A_PIECES + NW - 1) / NW;          // run pieces per wave (the last one may not exist)
    constexpr int A_H0 = (A_PER + 1) / 2;                    // issued in step s = 0; the rest in step s = 1
    constexpr int B_BYTES = BN * 128, B_PIECES = BN / 8, B_PER = (B_PIECES + NW - 1) / NW;
    constexpr int B_OFF = 2 * A_BUF;
    constexpr int ZERO_OFF = B_OFF + 2 * B_BYTES;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "16x16 fragments");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    // Fragment reads address LDS by NUMBER: the kernel has no static LDS, so the dynamic block starts at byte 0, but the
    // compiler only learns that at link time and emits `v_add_u32 v, <base>, v` in front of every ds_read whose address is
    // formed as smem + offset -- ten VALU instructions a step between the MFMA chunks (checked once below)
    auto lds_at = [](unsigned off) __attribute__((always_inline)) { return (const lds_char*)off; };
    if ((unsigned)(uintptr_t)smem != 0u) __builtin_trap();

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
    // (developer switch, p.dev_param == 55: a stream takes a CONTIGUOUS chunk of its XCD's tiles instead of every m_streams-th one --
    // the halo rows two vertically adjacent tiles share are then re-read by the same CU one tile later, not by its neighbour at the
    // same time; round 6, read amplification)
    const bool chunked = p.dev_param == 55;
    const int per_stream = (xcd_tiles + p.m_streams - 1) / p.m_streams;
    const int my_tiles = chunked ? max(0, min(per_stream, xcd_tiles - ms * per_stream))
                                 : ((xcd_tiles > ms) ? (xcd_tiles - ms + p.m_streams - 1) / p.m_streams : 0);
    if (my_tiles <= 0) return;
    const int first_tile = chunked ? xcd_first + ms * per_stream : xcd_first + ms;
    const int tile_step = chunked ? 1 : p.m_streams;
    const int last_tile = first_tile + (my_tiles - 1) * tile_step;
    const int n0 = tile_n * BN;
    const int G = p.groups;                       // 64-channel groups (the last one may be half full)
    const int runs_per_tile = 3 * G;              // (channel group, kernel row) pairs
    // a last channel group of at most 32 channels with the paired packing: its runs take two steps (taps 0 + 1, tap 2)
    constexpr bool RT = TAIL < 0;
    const bool pair = RT ? ((PROF & 64) == 0 && p.wgt4p != nullptr && (p.C8 & 7) != 0 && (p.C8 & 7) <= 4) : TAIL == 2;
    const int steps_per_tile = pair ? 9 * G - 3 : 9 * G;
    const int total_runs = my_tiles * runs_per_tile;

    // the row of zeros that invalid (pixel, tap) pairs read; behind it (offset 256 of the same KiB) the bias of this
    // workgroup's BN output channels, staged once: the epilogue of every tile reads its 4 channels per fragment column
    // with one ds_read_b128 instead of a scalar load + wait per column (5 dependent round trips per tile)
    static_assert(BN * 4 + 256 <= v5_zero_bytes(BN), "bias staging area");
    static_assert(!AL || (TM == 80 && TN == 80 && BN * 4 + 256 <= 2048), "aligned mode: 80x80 wave tiles");
    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    if constexpr (AL) {
        // (aligned mode: 256 bytes of zeros at ZERO_OFF + i * 2048 for every fragment row i; v5_al_extra_lds)
        for (int c = tid; c < (FM - 1) * 16; c += NW * 64)
            *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + (c / 16 + 1) * 2048 + (c % 16) * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int c = tid; c < BN; c += NW * 64)
        *(__attribute__((address_space(3))) float*)(smem + ZERO_OFF + 256 + c * 4) = (n0 + c < p.n_rows) ? p.bias[n0 + c] : 0.f;

    // ---- weight stream: slab (cg, tap) = 128 bytes of every row at byte offset step * 128 ---------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const int w_kpad = pair ? p.k_pad4p : p.k_pad4;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((pair ? p.wgt4p : p.wgt4) + (size_t)n0 * w_kpad), 0, kNumRecords, 0x00020000);
    // LEAN (the 8-wave tiles, which have no register to spare): one offset register for the wave's first piece, the
    // others are that + i * NW * 8 rows; needs every row of the tile to exist (conv5_supports: n_rows % BN == 0)
    constexpr bool LEAN = FM == 5 && FN == 5;
    unsigned b_off[LEAN ? 1 : B_PER];
#pragma unroll
    for (int i = 0; i < (LEAN ? 1 : B_PER); ++i) {
        const int row = (i * NW + wave) * 8 + lr;
        b_off[i] = (LEAN || (row < BN && n0 + row < p.n_rows)) ? (unsigned)(row * w_kpad + jj * 8) * 2u : kOOB;
    }
    const unsigned b_stride = (unsigned)(NW * 8 * w_kpad) * 2u;        // (LEAN) bytes between a wave's pieces
    int l_step = 0;                                // the weight loader's step inside a tile (same for every tile)
    // LEAN: a piece is issued with as few instructions as the hardware needs -- every instruction between two MFMA chunks
    // is matrix-pipe idle time, because the two waves of a SIMD run the same code between the same barriers.  Weight
    // pieces: the lane offset never changes, (step, piece) go into the scalar offset.  Run pieces: addressed from the
    // tensor's first byte through a descriptor that covers exactly the tensor, so that the range check does what a
    // compare + select per piece did (an offset "before" the tensor wraps to a huge one); see dma_run_piece.
    auto b_voff = [&](int i) __attribute__((always_inline)) -> unsigned {
        if constexpr (LEAN) {
            // (the multiple of the stride is made opaque: the compiler would otherwise keep one precomputed offset
            // register per piece alive through the loop -- the registers this form exists to save)
            unsigned d = (unsigned)i * b_stride;
            asm volatile("" : "+s"(d));
            return b_off[0] + d;
        } else return b_off[i];
    };
    auto dma_b_piece = [&](int stage, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if ((B_PIECES % NW) != 0 && i == B_PER - 1 && wave >= B_PIECES % NW) return;           // wave-uniform
        if constexpr (LEAN) {
            unsigned so = (unsigned)l_step * 128u + (unsigned)i * b_stride;
            asm volatile("" : "+s"(so));
            MDHIP_DMA16(b_rsrc, smem + B_OFF + stage * B_BYTES + (i * NW + wave) * 1024, b_off[0], so);
        } else {
            MDHIP_DMA16(b_rsrc, smem + B_OFF + stage * B_BYTES + (i * NW + wave) * 1024, b_voff(i), l_step * 128);
        }
    };
    auto dma_b_done = [&]() __attribute__((always_inline)) { l_step = (l_step + 1 == steps_per_tile) ? 0 : l_step + 1; };

    // ---- run loader: one (group, kernel row) ahead of the consumer ------------------------------------
    // Buffer row q of run (tile, cg, r) holds input pixel  tile*BM + (r-1)*W - 1 + q  (raster index over
    // the whole batch), channels cg*64 .. cg*64+63.  Pixels outside the batch read zeros; pixels that
    // are inside the batch but outside the image for some tap are dealt with at the fragment read.
    // (conv5_supports: the tensor is smaller than 4 GiB)
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((unsigned)p.M * (unsigned)p.ld_in * 2u), 0x00020000);
    unsigned q_off[LEAN ? 1 : A_PER];              // byte offset of this lane's pixel + chunk from the run's first pixel
#pragma unroll
    for (int i = 0; i < (LEAN ? 1 : A_PER); ++i) {
        const int q = (i * NW + wave) * 8 + lr;
        q_off[i] = (unsigned)(q * p.ld_in * 2 + jj * 16);
    }
    const unsigned q_stride = (unsigned)(NW * 8 * p.ld_in) * 2u;
    int lg_tile = first_tile, lg_cg = 0, lg_r = 0;
    int lg_first = 0;                              // raster index of the run's first pixel (may be negative)
    unsigned lg_abs = 0;                           // byte offset of (run's first pixel, channel group) in the tensor, mod 2^32
    auto run_setup = [&]() __attribute__((always_inline)) {
        lg_first = lg_tile * BM + (lg_r - 1) * p.W - 1;
        lg_abs = (unsigned)lg_first * (unsigned)p.ld_in * 2u + (unsigned)(lg_cg * 128);
    };
    // A run piece is addressed from the tensor's first byte: pixels outside the batch (before the first / behind the last
    // image) fall outside the descriptor and read zeros -- the range check does what a compare + select per piece would.
    // The 8-wave tiles do not test channels either: the tensors they take have a multiple of 32 channels (conv5_supports),
    // so the only chunks past the last channel are k 32..63 of a half-full last group, which no MFMA reads (tail_short).
    // After the stream's last tile the loader re-reads runs of that tile into buffers nobody reads.
    auto dma_run_piece = [&](int buf, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        // (tested only for the piece index where it can be true: the compiler does not know that wave < NW)
        if (i * NW + NW - 1 >= A_PIECES && i * NW + wave >= A_PIECES) return;                   // wave-uniform
        if constexpr (LEAN) {
            unsigned so = lg_abs + (unsigned)i * q_stride;
            asm volatile("" : "+s"(so));
            MDHIP_DMA16(a_rsrc, smem + buf * A_BUF + (i * NW + wave) * 1024, q_off[0] + so, 0);
        } else {
            // (the tiles with registers to spare keep the channel test: any channel count that is a multiple of 8)
            const bool ok = jj < p.C8 - lg_cg * 8;
            MDHIP_DMA16(a_rsrc, smem + buf * A_BUF + (i * NW + wave) * 1024, ok ? q_off[i] + lg_abs : 0xffffff00u, 0);
        }
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

    // ---- fragment reads ---------------------------------------------------------------------------
    // run buffer: fragment row = wave row + i*16 + (lane & 15) + s; the 16-byte chunk of k-chunk c of
    // buffer row q sits at position c ^ (q & 7)
    const int c0 = lane >> 4;
    // byte offset inside a run buffer of fragment 0 at shift s: three registers, or (LEAN, which has none to spare)
    // worked out again in every step from an opaque copy of the lane id, six VALU instructions under the MFMAs
    auto a_shift = [&](int l, int s) __attribute__((always_inline)) -> unsigned {
        return (unsigned)((wm * TM + (l & 15) + s) * 128 + (((l >> 4) ^ (((l & 7) + s) & 7)) << 4));
    };
    unsigned a_sh[LEAN ? 1 : 3];
    if constexpr (!LEAN) {
#pragma unroll
        for (int s = 0; s < 3; ++s) a_sh[s] = a_shift(lane, s);
    }
    auto a_shift_now = [&](int s) __attribute__((always_inline)) -> unsigned {
        if constexpr (LEAN) {
            int l = lane;
            asm volatile("" : "+v"(l));
            return a_shift(l, s);
        } else return a_sh[s];
    };
    const unsigned z_addr = (unsigned)(ZERO_OFF + c0 * 16);
    const int b_frag_base = B_OFF + (wn * TN + (lane & 15)) * 128 + ((c0 ^ (lane & 7)) << 4);
    uint32_t vmask[AL ? 1 : FM];                   // tap-validity bits of this lane's FM pixels (tile being read)
    unsigned a_eff[AL ? 1 : FM];                   // LDS address of the fragments of the step being read
    // aligned mode: X = k 0..31 addresses of the step being read next (fragments 1..3 / fragment 0 / fragment 4, each + i * 2048),
    // Y = the k 32..63 addresses of the step being computed
    [[maybe_unused]] unsigned al_bx = 0, al_e0 = 0, al_e4 = 0, al_by = 0, al_f0 = 0, al_f4 = 0;
    [[maybe_unused]] int wflags = 16;              // of this wave in the tile being read: 1 top row, 2 bottom row, 4 starts a row, 8 ends one, 16 outside the batch
    [[maybe_unused]] unsigned al_sh[3] = {0, 0, 0};
    if constexpr (AL) {
#pragma unroll
        for (int s = 0; s < 3; ++s) al_sh[s] = a_shift(lane, s);
    }
    [[maybe_unused]] const bool lane_p0 = (lane & 15) == 0, lane_p15 = (lane & 15) == 15;
    auto wave_flags = [&](int t) __attribute__((always_inline)) {
        const int mw = t * BM + wm * TM;           // (scalar: the wave's first pixel)
        int f = 16;
        if (mw < p.M) {
            const int b = mw / p.HoWo;
            const int rem = mw - b * p.HoWo;
            const int y = rem / p.W;
            const int x = rem - y * p.W;
            f = (y == 0 ? 1 : 0) | (y == p.H - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x + TM == p.W ? 8 : 0);
        }
        wflags = __builtin_amdgcn_readfirstlane(f);
    };
    // the k 0..31 addresses of step (buf, r, s) from the wave's flags: everything reads zeros when the kernel row falls above /
    // below the image (or the wave lies behind the batch); lane 0 of fragment 0 / lane 15 of fragment 4 when the tap falls
    // left / right of it
    auto al_x_addresses = [&](int buf, int r, int s) __attribute__((always_inline)) {
        const unsigned a = (s == 0 ? al_sh[0] : (s == 1 ? al_sh[1] : al_sh[2])) + (unsigned)(buf * A_BUF);
        // (bit arithmetic, not || chains: those become a branch per term)
        const int kill = wflags & (16 | (r == 0 ? 1 : 0) | (r == 2 ? 2 : 0));
        al_bx = kill != 0 ? z_addr : a;
        const int edge0 = wflags & (s == 0 ? 4 : 0), edge4 = wflags & (s == 2 ? 8 : 0);
        al_e0 = (edge0 != 0 && lane_p0) ? z_addr : al_bx;
        al_e4 = (edge4 != 0 && lane_p15) ? z_addr : al_bx;
        asm volatile("" : "+v"(al_bx), "+v"(al_e0), "+v"(al_e4));
    };
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
    auto set_a_eff_one = [&](int buf, int r, int s, int i, unsigned a_s) __attribute__((always_inline)) {
        const unsigned a = a_s + (unsigned)(buf * A_BUF + i * 2048);
        if constexpr ((PROF & 32) != 0) a_eff[i] = a;
        else a_eff[i] = ((vmask[i] >> (r * 3 + s)) & 1u) ? a : z_addr;
        // (pinned here, in the first half of a step: left alone the compiler sinks the select into the second half, next
        // to the read that uses it -- the half that also issues the DMA pieces and has no instruction slot to spare)
        asm volatile("" : "+v"(a_eff[i]));
    };
    auto read_x = [&](int i, int kk) -> frag8_t {
        if constexpr (AL) {
            const unsigned a = kk == 0 ? (i == 0 ? al_e0 : (i == FM - 1 ? al_e4 : al_bx)) : (i == 0 ? al_f0 : (i == FM - 1 ? al_f4 : al_by));
            return *(const __attribute__((address_space(3))) frag8_t*)lds_at(a + (unsigned)(i * 2048));
        } else return *(const __attribute__((address_space(3))) frag8_t*)lds_at(a_eff[i] ^ (unsigned)(kk * 64));
    };
    auto read_w = [&](int stage, int kk, int j) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)lds_at((unsigned)(stage * B_BYTES + j * 2048) +
                                                                        (unsigned)(b_frag_base ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (developer variants, PROF & 1: s_memtime stamps.  PROF & 128 turns the six slots into a profile of the EPILOGUE: slots 0 .. 4 = its
    // five pixel rows (residual wait + arithmetic + store issue of that row), slot 5 = everything else)
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr ((PROF & 1) != 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[(PROF & 128) != 0 ? 5 : k] += t - t_prev;
            t_prev = t;
        }
    };
    auto stamp_row = [&](int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 129) == 129) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[i < 5 ? i : 4] += t - t_prev;
            t_prev = t;
        }
    };
    // ---- epilogue (bias staged in LDS, pixel-row order, packed SiLU, 16-byte buffer stores) ----------
    const int q4 = lane >> 4;
    auto epilogue_t = [&](int tile_m, auto has_res_t, auto out_f32_t, auto act_t) __attribute__((always_inline)) {
        // (x * r and the residual add stay two roundings -- as in every other kernel family, where a select on the
        // activation flag sits between them: the results of the network do not depend on which tile a layer got)
#pragma clang fp contract(off)
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        constexpr bool OUT_F32 = decltype(out_f32_t)::value;
        constexpr bool ACT_FIXED = decltype(act_t)::value;
        // (set here, opaquely: as a compile-time constant it is put into a register pair ahead of the main loop, which
        // has none to spare)
        float neg_log2e = -0x1.715476p+0f;
        asm volatile("" : "+v"(neg_log2e));
        // After the exchanges below, lane (pixel p = lane & 15, q = lane >> 4) holds 16 bytes = channels q*8 .. q*8+7 of a
        // 32-channel column pair and stores them from there.  (Moving the chunk of (p, q) to lane 4*p + q first -- four
        // adjacent lanes write 64 contiguous bytes, a quarter wave touches 4 cache lines instead of 16 -- was measured:
        // the ds_bpermutes cost 2-5 % more than the address path saves, profiles/r3_convbench_epilogue.txt.)
        // (from an opaque copy of the lane id: everything below that depends only on the lane -- with tile-relative
        // descriptors that is every offset of every row -- would otherwise be computed once ahead of the main loop,
        // spilled there, and reloaded in front of each store behind an s_waitcnt vmcnt(0))
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int lp = lane_e & 15;
        const int lq = lane_e >> 4;
        const int m0 = tile_m * BM + wm * TM + lp;
        const int nbase = n0 + wn * TN + q4 * 4;                       // accumulator layout (fp32 outputs)
        const int nlast = n0 + wn * TN + (FN - 1) * 16 + lq * 4;       // this lane's 4 channels of an odd last column
        // 16-bit outputs and the residual go through buffer instructions: ONE 32-bit offset register per lane and tensor
        // (column pairs sit at immediate offsets), rows past the tensor's end are dropped / read as zeros by the range
        // check.  With 64-bit global addresses the compiler kept an address pair per column alive across the tile's main
        // loop, spilled them, and reloaded one before every store -- each reload an s_waitcnt vmcnt(0), i.e. a full
        // write round trip per store and no residual row ever in flight behind the one being finished.
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const int npair0 = n0 + wn * TN + lq * 8;                      // first of this lane's 8 channels of column pair 0
        // (descriptors start at the tile's first pixel: offsets stay small whatever the tensor's size)
        const long long rows_left = (long long)p.M - (long long)tile_m * BM;
        const int ml = wm * TM + lp;                                    // this lane's pixel inside the tile (+ 16 per row i)
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((uint16_t*)p.out + (size_t)tile_m * BM * p.ld_out), 0, (int)min(rows_left * p.ld_out * 2, 0x7fffffffLL), 0x00020000);
        const unsigned o_pair = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)npair0) * 2u;
        const unsigned o_last = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)nlast) * 2u;
        const unsigned o_step = 16u * (unsigned)p.ld_out * 2u;
        // (8-wave tiles: every channel of the tile exists, conv5_supports)
        auto n_ok = [&](int n) __attribute__((always_inline)) -> bool { return LEAN || n < p.N; };
        // (8-wave tiles: the bias is read from LDS again for every pixel row; 20 registers the epilogue does not have)
        constexpr bool BIAS_PER_ROW = LEAN;
        f32x4 bv[FN];
        auto read_bias = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
                // (address from the opaque lane copy: formed from `lane` it is hoisted above the main loop, one register per column --
                // the paired-tail instantiations spilled two of them and reloaded each behind an s_waitcnt vmcnt(0) in the epilogue [r5])
                bv[j] = *(const __attribute__((address_space(3))) f32x4*)lds_at((unsigned)(ZERO_OFF + 256 + (wn * TN + j * 16) * 4) + (unsigned)lq * 16u);
        };
        if constexpr (!BIAS_PER_ROW) read_bias();
        // The residual is read the way the output is written: 16 bytes per lane = 8 consecutive channels of a pair of
        // fragment columns (64 contiguous bytes per pixel and instruction, 3 loads per pixel row instead of 5), and
        // brought back to the accumulator layout by the inverse of the store exchange (v_permlane16_swap, then
        // v_permlane32_swap: both are involutions).
        constexpr int NPAIR = FN / 2;
        // residual rows in flight ahead of the row being finished: 1 (two workgroups per CU: the partner workgroup's
        // MFMAs cover the round trip) or 2 (LEAN = one 8-wave workgroup per CU: every wave of the CU is in its epilogue
        // at the same time, and each pixel row would otherwise wait out most of an HBM round trip on its own)
        constexpr int RA = LEAN ? 2 : 1, RS = RA + 1;     // (all five rows up front, round 4: 1 - 4 % slower; three, round 5: + 0 .. 1 %)
        uint4 rpair[RS][NPAIR > 0 ? NPAIR : 1];
        uint2 rlast[RS];
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.res + (HAS_RES ? (size_t)tile_m * BM * p.ld_res : 0)), 0,
            HAS_RES ? (int)min(rows_left * p.ld_res * 2, 0x7fffffffLL) : 0, 0x00020000);
        const unsigned r_pair = ((unsigned)ml * (unsigned)p.ld_res + (unsigned)npair0) * 2u;
        const unsigned r_last = ((unsigned)ml * (unsigned)p.ld_res + (unsigned)nlast) * 2u;
        const unsigned r_step = 16u * (unsigned)p.ld_res * 2u;
        auto fetch_res_row = [&](int i, uint4 (&rp)[NPAIR > 0 ? NPAIR : 1], uint2& rl) {
            // (no branches, no clamps: what lies outside the tensor reads as zero and is never stored)
#pragma unroll
            for (int jp = 0; jp < NPAIR; ++jp) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, (int)(r_pair + (unsigned)i * r_step + (unsigned)(jp * 64)), 0, 0);
                rp[jp] = make_uint4(t[0], t[1], t[2], t[3]);
            }
            if (FN & 1) {
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, (int)(r_last + (unsigned)i * r_step), 0, 0);
                rl = make_uint2(t[0], t[1]);
            }
        };
        if constexpr ((PROF & 129) == 129) stamp(5);
        if constexpr (HAS_RES) {
#pragma unroll
            for (int a = 0; a < RA && a < FM; ++a) fetch_res_row(a, rpair[a % RS], rlast[a % RS]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (HAS_RES) {
                if (i + RA < FM) fetch_res_row(i + RA, rpair[(i + RA) % RS], rlast[(i + RA) % RS]);
            }
            const int m = m0 + i * 16;
            if constexpr (BIAS_PER_ROW) read_bias();
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 t = f32x2{acc[i][j][r], acc[i][j][r + 1]} + f32x2{bv[j][r], bv[j][r + 1]};
                    if constexpr ((PROF & 4) == 0) {
                        if (ACT_FIXED || p.act) t = silu_f32x2(t, neg_log2e);
                    }
                    v[j][r] = t[0];
                    v[j][r + 1] = t[1];
                }
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (HAS_RES) {
                auto add4 = [&](int j, unsigned lo, unsigned hi) {
                    v[j][0] += st_unpack((uint16_t)(lo & 0xffff));
                    v[j][1] += st_unpack((uint16_t)(lo >> 16));
                    v[j][2] += st_unpack((uint16_t)(hi & 0xffff));
                    v[j][3] += st_unpack((uint16_t)(hi >> 16));
                };
#pragma unroll
                for (int jp = 0; jp < NPAIR; ++jp) {
                    const uint4 d = rpair[i % RS][jp];           // as stored: (t0[0], t1[0], t0[1], t1[1])
                    auto s0 = __builtin_amdgcn_permlane16_swap(d.x, d.z, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(d.y, d.w, false, false);
                    auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s0[1], false, false);     // (a0, b0)
                    auto a1 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);     // (a1, b1)
                    add4(2 * jp, a0[0], a1[0]);
                    add4(2 * jp + 1, a0[1], a1[1]);
                }
                if (FN & 1) {
                    add4(FN - 1, rlast[i % RS].x, rlast[i % RS].y);
                }
            }
            if constexpr ((PROF & 2) != 0) {
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(v[j][0]), "v"(v[j][1]), "v"(v[j][2]), "v"(v[j][3]));
            } else if constexpr (OUT_F32) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = nbase + j * 16;
                    if (m < p.M && n < p.N)
                        *(float4*)((float*)p.out + (size_t)m * p.ld_out + n) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                    unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                    auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                    const uint4 o = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                    const unsigned off = o_pair + (unsigned)i * o_step + (unsigned)(j * 32);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{o.x, o.y, o.z, o.w}, o_rsrc,
                                                           (int)(n_ok(npair0 + j * 16) ? off : kOOB), 0, 0);
                }
                if (FN & 1) {
                    const int j = FN - 1;
                    const uint2 o = make_uint2(st_pack2(v[j][0], v[j][1]), st_pack2(v[j][2], v[j][3]));
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{o.x, o.y}, o_rsrc,
                                                          (int)(n_ok(nlast) ? o_last + (unsigned)i * o_step : kOOB), 0, 0);
                }
            }
            stamp_row(i);
        }
    };
    // The 8-wave tiles are built for activated 16-bit outputs only (conv5_supports): two epilogues, no select per value.
    // The others take the activation flag at run time (std::false_type = "ask p.act").
    auto epilogue = [&](int tile_m) __attribute__((always_inline)) {
        if constexpr (LEAN) {
            // (s_setprio 2 on the first four waves for the epilogue -- so that one wave of a SIMD computes while the other waits
            // for the memory path -- was measured: no difference on any layer, profiles/r4_convbench_epilogue_prio.txt)
            if (p.res) epilogue_t(tile_m, std::true_type{}, std::false_type{}, std::true_type{});
            else epilogue_t(tile_m, std::false_type{}, std::false_type{}, std::true_type{});
        } else {
            if (p.out_f32) epilogue_t(tile_m, std::false_type{}, std::true_type{}, std::false_type{});
            else if (p.res) epilogue_t(tile_m, std::true_type{}, std::false_type{}, std::false_type{});
            else epilogue_t(tile_m, std::false_type{}, std::false_type{}, std::false_type{});
        }
    };

    // ---- prologue: run (first tile, group 0, r 0) in buffer 0, weight slabs of steps 0 and 1 ----------
    run_setup();
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        if (i * NW + wave < A_PIECES) {
            MDHIP_DMA16(a_rsrc, smem + (i * NW + wave) * 1024,
                        (LEAN || jj < p.C8) ? (LEAN ? q_off[0] + (unsigned)i * q_stride : q_off[i]) + lg_abs : 0xffffff00u, 0);
        }
    }
    run_next();
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            if ((B_PIECES % NW) != 0 && i == B_PER - 1 && wave >= B_PIECES % NW) continue;
            MDHIP_DMA16(b_rsrc, smem + B_OFF + st * B_BYTES + (i * NW + wave) * 1024, b_voff(i), l_step * 128);
        }
        dma_b_done();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
    if constexpr (AL) {
        wave_flags(first_tile);
        al_x_addresses(0, 0, 0);
    } else {
        tile_masks(first_tile);
#pragma unroll
        for (int i = 0; i < FM; ++i) set_a_eff_one(0, 0, 0, i, a_shift_now(0));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) xa[i] = read_x(i, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) wa[j] = read_w(0, 0, j);

    int c_r = 0, c_cg = 0, c_tile = first_tile, pa = 0, step = 0;
    [[maybe_unused]] bool after_epilogue = false;
    if constexpr ((PROF & 1) != 0) t_prev = __builtin_amdgcn_s_memtime();
#define MDHIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    // a half-full last channel group (C_in mod 64 <= 32) has nothing in k 32..63: its second-half MFMAs are skipped
    const bool tail_short = RT ? ((PROF & 64) == 0 && (p.C8 & 7) != 0 && (p.C8 & 7) <= 4) : TAIL >= 1;
    // DMA slots behind the MFMA chunks of a second half: the weight pieces of step + 2, then this step's share of the NEXT
    // run's pieces -- the first A_H0 in step 0 and the rest in step 1, or all of them in step 0 of a paired run
    constexpr int DMA_MAX = B_PER + A_PER, DMA_PER_G = (DMA_MAX + FN - 1) / FN;
    for (int run = 0; run < total_runs; ++run) {
        const bool last_cg = c_cg == G - 1;
        const bool tile_end = c_r == 2 && last_cg;
        const int n_r = c_r == 2 ? 0 : c_r + 1;
        // (TAIL == 0: both are compile-time false and every test below folds away.  Two copies of the steps -- one for the
        // full groups, one for the last -- were tried for TAIL 1 / 2: 320+ spilled registers in the 8-wave tiles)
        const bool pair_run = TAIL == 0 ? false : (pair && last_cg);           // this run: taps 0 + 1 in one step, then tap 2
        const bool short_run = TAIL == 0 ? false : (tail_short && last_cg);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s == 1 && pair_run) continue;
            const int cur = step & 1;
            // no MFMAs on k 32..63: a half-full last group's step, unless it carries the second tap of a pair there
            const bool skip_y = short_run && !(pair_run && s == 0);
            // the step being prefetched: the next tap of this run, or the first tap of the next run
            const int ns = s == 2 ? 0 : ((s == 0 && pair_run) ? 2 : s + 1);
            const int nbuf = s == 2 ? pa ^ 1 : pa;
            const int nr = s == 2 ? n_r : c_r;
            unsigned a_next = 0, a_pair = 0;
            if constexpr (AL) {
                // the k 32..63 addresses of THIS step: the other half of the same rows -- or, first step of a paired run, tap 1
                // of this kernel row: k 0..31 one pixel on, no pixel of it left or right of the image
                al_by = al_bx ^ 64u;
                al_f0 = al_e0 ^ 64u;
                al_f4 = al_e4 ^ 64u;
                if (s == 0 && pair_run) {
                    const int kill = wflags & (16 | (c_r == 0 ? 1 : 0) | (c_r == 2 ? 2 : 0));
                    al_by = al_f0 = al_f4 = kill != 0 ? z_addr : al_sh[1] + (unsigned)(pa * A_BUF);
                }
                if (s == 2 && tile_end) wave_flags(c_tile + tile_step);   // (flags of a tile past the stream's end are never used)
                al_x_addresses(nbuf, nr, ns);
            } else {
                if (s == 2 && tile_end) tile_masks(c_tile + tile_step);   // (masks of a tile past the stream's end are never used)
                a_next = a_shift_now(ns);
                // (paired step: the k 32..63 half is tap 1 of this kernel row -- the fragments one pixel on, k 0..31)
                a_pair = (s == 0 && pair_run) ? a_shift_now(1) + (unsigned)(pa * A_BUF) : 0u;
            }
            // ---- first half: k 0..31 of this step, while its k 32..63 fragments are read and the
            //      fragment addresses of the next step are selected; MFMA chunk g = fragment column g ----
            auto read_y = [&](int i) __attribute__((always_inline)) -> frag8_t {
                if constexpr (AL) return read_x(i, 1);
                else if (s == 0 && pair_run) {
                    const unsigned a = ((vmask[i] >> (c_r * 3 + 1)) & 1u) ? a_pair + (unsigned)(i * 2048) : z_addr;
                    return *(const __attribute__((address_space(3))) frag8_t*)lds_at(a);
                }
                return read_x(i, 1);
            };
#pragma unroll
            for (int g = 0; g < FN; ++g) {
                // LEAN: the weight fragment of the other k half goes into the registers the chunk before released (one
                // spare fragment, read first): 6 weight fragments live instead of 10
                if constexpr (LEAN) wb[(g + FN - 1) % FN] = read_w(cur, 1, (g + FN - 1) % FN);
                else wb[g] = read_w(cur, 1, g);
                if (g < FM) { xb[g] = read_y(g); if constexpr (!AL) set_a_eff_one(nbuf, nr, ns, g, a_next); }
                if (g == FN - 1) {
#pragma unroll
                    for (int i = FN; i < FM; ++i) { xb[i] = read_y(i); if constexpr (!AL) set_a_eff_one(nbuf, nr, ns, i, a_next); }
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
            // (stamps: the wait and the barrier of the first step after an epilogue -- whose stores this wait also
            // covers -- are booked under slot 4)
            if (after_epilogue) stamp(4); else stamp(1);
            __builtin_amdgcn_s_barrier();
            if (after_epilogue) stamp(4); else stamp(2);
            after_epilogue = false;
            MDHIP_FENCE();

            // ---- second half: the k 0..31 fragments of the next step, MFMAs on k 32..63, and the DMA
            //      pieces (weight slab of step+2; in steps 0 and 1 the next run) behind the MFMA chunks ----
#pragma unroll
            for (int g = 0; g < FN; ++g) {
                if constexpr (LEAN) wa[(g + FN - 1) % FN] = read_w(cur ^ 1, 0, (g + FN - 1) % FN);
                else wa[g] = read_w(cur ^ 1, 0, g);
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
#pragma unroll
                for (int d = g * DMA_PER_G; d < (g + 1) * DMA_PER_G && d < DMA_MAX; ++d) {
                    if (d < B_PER) dma_b_piece(cur, d);
                    else if (s == 0 && (d - B_PER < A_H0 || pair_run)) dma_run_piece(pa ^ 1, d - B_PER);
                    else if (s == 1 && A_H0 + d - B_PER < A_PER) dma_run_piece(pa ^ 1, A_H0 + d - B_PER);
                }
                MDHIP_FENCE();
            }
            dma_b_done();
            ++step;
            stamp(3);
        }
        // the run is consumed: the loader moves on; maybe the tile is complete
        run_next();
        pa ^= 1;
        c_r = n_r;
        if (n_r == 0 && ++c_cg == G) {
            c_cg = 0;
            epilogue(c_tile);
            after_epilogue = true;
            c_tile += tile_step;
        }
        stamp(5);
    }
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
#define MDHIP_CONV5_CFGS(X) \
    X(0, 128, 160, 2, 2, 0) \
    X(1, 128, 80, 4, 1, 0)  \
    X(2, 256, 160, 4, 2, 0) \
    X(3, 192, 80, 4, 1, 0)  \
    X(4, 64, 160, 2, 2, 0)  \
    X(5, 64, 80, 2, 1, 0)   \
    X(6, 160, 320, 2, 4, 0) \
    X(7, 320, 160, 4, 2, 0)
#define MDHIP_CONV5_PROF(X)  \
    X(8, 128, 160, 2, 2, 1)  \
    X(9, 128, 160, 2, 2, 16) \
    X(10, 128, 160, 2, 2, 22) \
    X(11, 320, 160, 4, 2, 1) \
    X(12, 320, 160, 4, 2, 16) \
    X(13, 320, 160, 4, 2, 6) \
    X(14, 320, 160, 4, 2, 22) \
    X(15, 320, 160, 4, 2, 2) \
    X(16, 320, 160, 4, 2, 3) \
    X(17, 320, 160, 4, 2, 32) \
    X(18, 320, 160, 4, 2, 64) \
    X(19, 320, 160, 4, 2, 96) \
    X(20, 320, 160, 4, 2, 118) \
    X(21, 320, 160, 4, 2, 66) \
    X(22, 320, 160, 4, 2, 68) \
    X(23, 320, 160, 4, 2, 70) \
    X(24, 320, 160, 4, 2, 129) \
    X(25, 320, 160, 4, 2, 131) \
    X(26, 320, 160, 4, 2, 133)

static const ConvCfg g_cfgs5[] = {
#define X(id, bm, bn, wm, wn, prof)                                                                   \
    {bm, bn, (wm) * (wn) * 64, (size_t)v5_lds_bytes(bm, bn), v5_blocks_per_cu(bm, bn, (wm) * (wn)), \
     "v5:run" #bm "x" #bn "/" #wm "x" #wn "/" #prof},
    MDHIP_CONV5_CFGS(X) MDHIP_CONV5_PROF(X)
#undef X
};
constexpr int kNumProf5 = 19;

// ids: [0, kNumMain5) the configurations above, then the small-launch configurations of conv_v5s.cpp and the C = 80
// strip kernel of conv_v5c.cpp (same K order, same results), then the developer variants
constexpr int kNumMain5 = (int)(sizeof(g_cfgs5) / sizeof(g_cfgs5[0])) - kNumProf5;
static int first5c() { return kNumMain5 + conv5s_num_cfgs(); }

int conv5_num_cfgs() { return first5c() + conv5c_num_cfgs(); }
const ConvCfg& conv5_cfg(int i) {
    if (i < kNumMain5) return g_cfgs5[i];
    if (i < first5c()) return conv5s_cfg(i - kNumMain5);
    if (i < conv5_num_cfgs()) return conv5c_cfg(i - first5c());
    return g_cfgs5[i - conv5_num_cfgs() + kNumMain5];
}

hipError_t conv5_init() {
    hipError_t e = conv5s_init();
    if (e == hipSuccess) e = conv5c_init();
#define X(id, bm, bn, wm, wn, prof)                                                              \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof>,                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs5[id].lds_bytes);
    MDHIP_CONV5_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                              \
    if constexpr ((bm) * (bn) == 320 * 160 && (wm) * (wn) == 8) {                                   \
        const int al_lds = (int)g_cfgs5[id].lds_bytes + v5_al_extra_lds;                          \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof, 0, v5_is_lean(bm, bn, wm, wn)>, hipFuncAttributeMaxDynamicSharedMemorySize, al_lds); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof, 1, v5_is_lean(bm, bn, wm, wn)>, hipFuncAttributeMaxDynamicSharedMemorySize, al_lds); \
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof, 2, v5_is_lean(bm, bn, wm, wn)>, hipFuncAttributeMaxDynamicSharedMemorySize, al_lds); \
    }                                                                                          \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof, 0>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs5[id].lds_bytes); \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof, 1>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs5[id].lds_bytes); \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v5_kernel<bm, bn, wm, wn, prof, 2>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs5[id].lds_bytes);
    MDHIP_CONV5_CFGS(X)
#undef X
    return e;
}

bool conv5_supports(int cfg, const ConvArgs& a) {
    if (cfg < 0 || cfg >= conv5_num_cfgs() + kNumProf5) return false;
    const bool ok = a.wgt4 != nullptr && a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H &&
                    a.Wo == a.W && a.C8 >= 8 && (a.N % 8) == 0 &&
                    (long long)(2 * a.W + conv5_cfg(cfg).bm + 16) * a.ld_in * 2 + 4096 < 0x7fffffffLL;
    // the run loader addresses the whole tensor through one descriptor (dma_run_piece)
    if (ok && (cfg < kNumMain5 || cfg >= conv5_num_cfgs()) && ((long long)a.M + 320 + 2 * a.W + 16) * a.ld_in * 2 >= 0xffffff00LL)
        return false;
    // the 80x80-wave-tile configurations (LEAN; the developer variants of that tile shape too): no channel test on stores
    // or weight rows, so every channel of every N tile must exist (N a multiple of BN, not merely of 8), activated 16-bit
    // outputs, 32-channel granularity of the input
    const int local = cfg < kNumMain5 ? cfg : (cfg >= conv5_num_cfgs() ? cfg - conv5_num_cfgs() + kNumMain5 : -1);
    if (ok && local >= 0 && g_cfgs5[local].threads >= 512 && g_cfgs5[local].bm * g_cfgs5[local].bn == 160 * 320 &&
        ((a.N % g_cfgs5[local].bn) != 0 || a.N != a.n_rows || !a.act || a.out_f32 || (a.C8 % 4) != 0))
        return false;
    if (ok && cfg >= kNumMain5 && cfg < first5c()) return conv5s_supports(cfg - kNumMain5, a);
    if (ok && cfg >= first5c() && cfg < conv5_num_cfgs()) return conv5c_supports(cfg - first5c(), a);
    return ok;
}

hipError_t conv5_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (!conv5_supports(cfg, a)) return hipErrorInvalidValue;
    if (cfg >= kNumMain5 && cfg < first5c()) return conv5s_launch(cfg - kNumMain5, a, s);
    if (cfg >= first5c() && cfg < conv5_num_cfgs()) return conv5c_launch(cfg - first5c(), a, s);
    if (cfg >= conv5_num_cfgs()) cfg -= conv5_num_cfgs() - kNumMain5;        // developer variants
    const ConvCfg& c = g_cfgs5[cfg];
    ConvArgs p = a;
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, (32 * c.blocks_per_cu) / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    // the last channel group (see TAIL): at most half full -> its k 32..63 MFMAs are skipped; with the paired packing two taps a step
    const bool tail_short = (a.C8 & 7) != 0 && (a.C8 & 7) <= 4;
    const int tail = !tail_short ? 0 : (a.wgt4p != nullptr ? 2 : 1);
    // the 8-wave tiles' aligned mode (conv_v5_kernel AL): a wave's 80 pixels inside one image row; same results
    const bool aligned = (a.W % 80) == 0 && a.dev_param != 77;
    switch (cfg) {
#define X(id, bm, bn, wm, wn, prof)                                                               \
    case id:                                                                                    \
        if (v5_is_lean(bm, bn, wm, wn) && aligned) {                                            \
            const size_t al_lds = c.lds_bytes + v5_al_extra_lds;                                  \
            if (tail == 0) hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof, 0, v5_is_lean(bm, bn, wm, wn)>), grid, dim3((wm) * (wn) * 64), al_lds, s, p); \
            else if (tail == 1) hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof, 1, v5_is_lean(bm, bn, wm, wn)>), grid, dim3((wm) * (wn) * 64), al_lds, s, p); \
            else hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof, 2, v5_is_lean(bm, bn, wm, wn)>), grid, dim3((wm) * (wn) * 64), al_lds, s, p); \
            break;                                                                              \
        }                                                                                       \
        if (tail == 0) hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof, 0>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        else if (tail == 1) hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof, 1>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        else hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof, 2>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV5_CFGS(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                               \
    case id:                                                                                    \
        hipLaunchKernelGGL((conv_v5_kernel<bm, bn, wm, wn, prof>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV5_PROF(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
