// 3x3 / STRIDE 2 convolution with row-run reuse across the three taps of a kernel row (gfx950 / MI355X).
//
// The stride-2 convs of the YOLOv5 stack (layers 1, 3, 5, 7, 24, 27 of the x6 topology: 4.1 of the 31 ms of a batch-32
// step) ran as implicit GEMMs (conv_v2.cpp), which gather every tap's pixels on their own through the L2 -> LDS path: 0.15
// (8-wave 320x160 tile) .. 0.2 (160x160) DMA pieces per MFMA, and the no-DMA ablation of those launches runs twice as fast
// (profiles/r3_convbench_v2_ablation.txt).  conv_v5.cpp's remedy for stride 1 -- one contiguous run of input pixels per
// kernel row serves all three taps -- carries over with two changes:
//
//   * An M tile is 320 consecutive OUTPUT pixels = 320 / Wo whole output rows (Wo in {40, 80, 160, 320}: every tile
//     starts at column 0).  Output pixel ox of kernel row r reads input columns 2 ox - 1, 2 ox, 2 ox + 1 of input row
//     2 oy + r - 1: the ODD columns serve tap 0 (one entry to the left) and tap 2, the EVEN columns tap 1.  So a run is two
//     sub-buffers of 320 entries (128-byte rows = one 64-channel group): O = odd columns (entry j <-> column 2 j + 1),
//     E = even columns (entry j <-> column 2 j), entry index = pixel index inside the tile.  Fragment rows are then
//     consecutive LDS rows exactly as for stride 1 (a stride of two rows would put all 16 rows of a fragment on the same
//     banks): the same XOR swizzle, conflict-free.  160 pieces for three taps instead of 240: 0.117 pieces per MFMA.
//   * 2 x 40 KiB of run + 2 x 20 KiB of weight stages leave no room for a second run buffer, so the sub-buffers are
//     single and the tap ORDER makes that possible: tap 0 (O), tap 2 (O), tap 1 (E).  O is free once the tap-2 step has
//     read its fragments -- the next run's O pieces are issued in the second half of that step; E is free after the tap-1
//     step -- the next run's E pieces follow in the second halves of that step and of the next run's first one.  Every
//     piece still has at least one whole step between its issue and the barrier that publishes it.
//
// (Round 4, built, measured, removed: a last channel group of at most 16 channels -- C_in = 80, layer 1, nine steps with a
// quarter of their k filled -- as ONE run with its nine taps four to a step (the 16 channels of the three kernel rows side
// by side in the 128-byte rows, the lanes of k 0..15 and k 16..31 of a k half reading different taps): 12 steps per tile
// instead of 18, correct on every shape, and 3 % SLOWER on layer 1 (616 against 634 TFLOP/s; profiles/
// r4_convbench_stride2_fat_tail.txt) -- that layer is bound by its 2.1 GB of input through L2 and by its epilogue, not
// by its step count.)
//
// No per-tap zero rows for the left border except tap 0 at ox = 0 and kernel row 0 at oy = 0 (three bits per pixel:
// in range, oy > 0, ox > 0); the right and bottom borders never leave the image (H = 2 Ho, W = 2 Wo, pad 1).
// Everything else -- 8 waves with 80x80 wave tiles, persistent XCD-local tile streams, weight slabs through a two-stage
// ring (conv_v4's packing, walked in the order (group, kernel row, tap 0 / 2 / 1)), the lane-local epilogue with packed
// SiLU and 16-byte buffer stores -- is conv_v5's 8-wave tile.  K order (group, r, tap 0 / 2 / 1, channel): a summation
// order of its own, so the layer's results equal the implicit-GEMM kernels' to fp32 rounding only (tolerance test), and
// a layer that takes this kernel at batch 32 takes it at every batch size (bitwise batch invariance, mdhip_capi.cpp).

#include <algorithm>
#include <type_traits>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) char lds_char;
typedef mdhip_f32x2 f32x2;

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;

constexpr int kBM7 = 320, kBN7 = 160, kWM7 = 4, kWN7 = 2, kNW7 = kWM7 * kWN7;
constexpr int kSub7 = kBM7 * 128;                                   // one sub-buffer: 320 entries x 128 bytes
constexpr int kStage7 = kBN7 * 128;                                 // one weight stage
constexpr int kZero7 = 1024;                                        // the row of zeros (256 bytes) + the staged bias
constexpr int kLds7 = 2 * kSub7 + 2 * kStage7 + kZero7;             // 123 904 bytes: one workgroup per CU
constexpr int kAlExtra7 = 4 * 2048;                                 // aligned mode: zero rows at ZERO_OFF + i * 2048, i = 1..4

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// TAIL: 0 = the channel count is a multiple of 64 (no partly full last group: its tests are compiled out), 1 = it is not.
// AL: a wave's 80 output pixels lie inside one output row (Wo a multiple of 80): tap validity from three per-wave flags --
// inside the batch, top row, starts a row -- one address register for fragments 1..4 at the immediate offsets i * 2048 (a
// row of zeros at each of those offsets), one for fragment 0, whose lane 0 may fall left of the image (conv_v5.cpp, AL).
template <int TAIL, bool AL>
__global__ void __launch_bounds__(kNW7 * 64, 2)
conv_v7_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = kBM7, BN = kBN7, WM = kWM7, WN = kWN7, NW = kNW7;
    constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
    static_assert(FM == 5 && FN == 5, "80x80 wave tiles");
    constexpr int A_PER = BM / 8 / NW;                       // pieces of ONE sub-buffer per wave: 5, all in one output row
    constexpr int E_H0 = 3;                                  // E pieces issued in the tap-1 step; the rest in the next run's first step
    constexpr int B_PIECES = BN / 8, B_PER = (B_PIECES + NW - 1) / NW;
    constexpr int O_OFF = 0, E_OFF = kSub7, B_OFF = 2 * kSub7, B_BYTES = kStage7, ZERO_OFF = B_OFF + 2 * B_BYTES;
    static_assert(BN * 4 + 256 <= kZero7, "bias staging area");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    // (fragment reads address LDS by number, conv_v5.cpp: the dynamic block must start at byte 0)
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
    const int my_tiles = (xcd_tiles > ms) ? (xcd_tiles - ms + p.m_streams - 1) / p.m_streams : 0;
    if (my_tiles <= 0) return;
    const int first_tile = xcd_first + ms;
    const int tile_step = p.m_streams;
    const int last_tile = first_tile + (my_tiles - 1) * tile_step;
    const int n0 = tile_n * BN;
    const int G = p.groups;                       // 64-channel groups (the last one may be partly full)
    const int runs_per_tile = 3 * G;
    const int total_runs = my_tiles * runs_per_tile;
    // [r6] developer switch (p.dev_param == 66; measured, not the default): every other tile of an image walks its kernel rows
    // BACKWARDS (r = 2, 1, 0 inside every channel group).  The input row below a tile's last output row is read by that tile's
    // kernel row 2 and by the next tile's kernel row 0; the two tiles run at the same time on neighbouring CUs of one XCD, but in
    // the common order the first reads it in the last third of a group's nine steps and the second in the first third -- six steps
    // apart, by which time the XCD has streamed twice its L2.  With alternating directions both read a shared row in the same
    // third: FETCH_SIZE of layer 1 falls by 20 % (5.03 -> 4.01 GB), of layer 3 by 11 % -- and the launches take the same time
    // (L1 + 2 %, L3 - 1.4 %, L5 / L7 / L24 / L27 +- 0.5 %; profiles/r6_read_amplification.txt): these layers are not bound by
    // their reads.  Since it would change the summation order of every second tile for nothing, the common order stays.  (The
    // parity is that of the tile's index INSIDE its image, so a pixel's order would not depend on its batch; tiles that straddle
    // images keep the common order.)
    const bool rev_on = (p.HoWo % BM) == 0 && p.dev_param == 66;
    const int tpi = rev_on ? p.HoWo / BM : 1;      // tiles per image
    auto tile_rev = [&](int t) __attribute__((always_inline)) -> bool { return rev_on && ((t % tpi) & 1) != 0; };

    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    if constexpr (AL) {
        for (int c = tid; c < (FM - 1) * 16; c += NW * 64)
            *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + (c / 16 + 1) * 2048 + (c % 16) * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int c = tid; c < BN; c += NW * 64)
        *(__attribute__((address_space(3))) float*)(smem + ZERO_OFF + 256 + c * 4) = p.bias[n0 + c];

    // ---- weight stream: conv_v4's packing ([n_rows][groups * 9 * 64], slab (group, tap) = 128 bytes of every row), walked
    //      in the order (group, kernel row, tap 0 / 2 / 1); one lane offset, (slab, piece) in the scalar offset --------------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt4 + (size_t)n0 * p.k_pad4), 0, kNumRecords, 0x00020000);
    const unsigned b_off = (unsigned)((wave * 8 + lr) * p.k_pad4 + jj * 8) * 2u;
    const unsigned b_stride = (unsigned)(NW * 8 * p.k_pad4) * 2u;
    int l_base = 0, l_j = 0;                       // the loader's slab: l_base = 9 * group + 3 * (kernel row), l_j = 0 / 1 / 2 -> tap 0 / 2 / 1
    int l_g = 0, l_r = 0, l_tile = first_tile;     // its group, its position in the tile's kernel-row walk, its tile
    bool l_rev = tile_rev(first_tile);
    l_base = l_rev ? 6 : 0;
    auto dma_b_piece = [&](int stage, int i) __attribute__((always_inline)) {
        if ((B_PIECES % NW) != 0 && i == B_PER - 1 && wave >= B_PIECES % NW) return;           // wave-uniform
        const int slab = l_base + (l_j == 0 ? 0 : (l_j == 1 ? 2 : 1));
        unsigned so = (unsigned)slab * 128u + (unsigned)i * b_stride;
        asm volatile("" : "+s"(so));
        MDHIP_DMA16(b_rsrc, smem + B_OFF + stage * B_BYTES + (i * NW + wave) * 1024, b_off, so);
    };
    auto dma_b_done = [&]() __attribute__((always_inline)) {
        if (++l_j == 3) {
            l_j = 0;
            if (++l_r == 3) {
                l_r = 0;
                if (++l_g == G) {
                    l_g = 0;
                    l_tile += tile_step;                   // (past the stream's last tile: slabs nobody reads)
                    l_rev = tile_rev(l_tile);
                }
            }
            l_base = 9 * l_g + 3 * (l_rev ? 2 - l_r : l_r);
        }
    };

    // ---- run loader ----------------------------------------------------------------------------------------------
    // Wave w loads entries 40 w .. 40 w + 39 of both sub-buffers: 40 divides Wo, so they lie in ONE output row
    // (b, oy) of the tile, from column ox_w on: entry j of that block <-> input pixel (b, 2 oy + r - 1, 2 (ox_w + j) + c),
    // c = 1 (O) / 0 (E).  Lane (lr, jj) of piece i handles entry 8 i + lr, 16-byte chunk jj: byte offset from the block's
    // first pixel  (8 i + lr) * 2 pixels + jj * 16  -- one lane register, the piece in the scalar part.  The tensor is
    // addressed from its first byte through a descriptor that covers exactly the tensor: a row above the image or past
    // the batch gets an offset outside it and reads zeros.
    // (The descriptor starts at the image of the loader tile's first pixel and covers what is left of the tensor, at most
    // 2 GiB: a tile spans at most 320 / 40 = 8 images, so no offset depends on the batch size -- whether this kernel takes a
    // layer must not depend on the batch an image travels in.)
    const unsigned q_off = (unsigned)(lr * 2 * p.ld_in * 2 + jj * 16);
    const unsigned q_stride = (unsigned)(8 * 2 * p.ld_in * 2);           // bytes between a wave's pieces (8 entries = 16 pixels)
    const bool tail_bad = (p.C8 & 7) != 0 && jj >= (p.C8 & 7);           // this lane's chunk of a partly full last group
    // a run as the loader sees it: descriptor of its tile, scalar byte offsets of the wave's block (kOOB: no such row)
    struct RunGeom { __amdgpu_buffer_rsrc_t rsrc; unsigned so_o, so_e; bool tail; };
    int lg_tile = first_tile, lg_cg = 0, lg_r = 0;
    int lg_px0 = 0;                                 // input pixel index of (b, 2 oy - 1, 2 ox_w) relative to the descriptor (may be negative)
    bool lg_top = false, lg_ok = false;             // oy > 0 ; the wave's row exists (inside the batch)
    bool lg_rev = false;                            // the loader's tile walks its kernel rows backwards
    __amdgpu_buffer_rsrc_t lg_rsrc = b_rsrc;
    const long long img_bytes = (long long)p.H * p.W * p.ld_in * 2;
    const int n_img = p.M / p.HoWo;
    auto tile_geom = [&]() __attribute__((always_inline)) {
        const int m_t = min(lg_tile * BM, p.M - 1);                        // the tile's first output pixel -> its image
        const int b0 = conv_udiv(m_t, p.HoWo, p.rcp_howo);
        const long long left = (long long)(n_img - b0) * img_bytes;
        lg_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + (long long)b0 * img_bytes), 0,
                                                    (int)(left > 0x7fffff00LL ? 0x7fffff00LL : left), 0x00020000);
        const int m_w = lg_tile * BM + 40 * wave;                          // first output pixel of the wave's block (wave-uniform)
        lg_ok = m_w < p.M;
        const int b = conv_udiv(min(m_w, p.M - 1), p.HoWo, p.rcp_howo);
        const int rem = m_w - b * p.HoWo;
        const int oy = conv_udiv(rem, p.Wo, p.rcp_wo);
        const int ox = rem - oy * p.Wo;
        lg_top = oy > 0;
        lg_px0 = ((b - b0) * p.H + 2 * oy - 1) * p.W + 2 * ox;
        lg_rev = tile_rev(lg_tile);
    };
    auto run_geom = [&]() __attribute__((always_inline)) -> RunGeom {
        RunGeom g;
        g.rsrc = lg_rsrc;
        const int re = lg_rev ? 2 - lg_r : lg_r;                          // the kernel row of this position of the walk
        const bool ok = lg_ok && (re > 0 || lg_top);
        const unsigned base = (unsigned)(lg_px0 + re * p.W) * (unsigned)p.ld_in * 2u + (unsigned)(lg_cg * 128);
        g.so_e = ok ? base : kOOB;
        g.so_o = ok ? base + (unsigned)p.ld_in * 2u : kOOB;
        g.tail = TAIL != 0 && lg_cg == G - 1 && (p.C8 & 7) != 0;
        return g;
    };
    auto run_next = [&]() __attribute__((always_inline)) {
        if (++lg_r == 3) {
            lg_r = 0;
            if (++lg_cg == G) {
                lg_cg = 0;
                if (lg_tile != last_tile) { lg_tile += tile_step; tile_geom(); }
            }
        }
    };
    // piece i of a sub-buffer (sub = O_OFF / E_OFF) of the run with geometry g
    auto dma_run_piece = [&](int sub, const __amdgpu_buffer_rsrc_t& a_rsrc, unsigned so_base, bool tail, int i) __attribute__((always_inline)) {
        unsigned so = so_base + (unsigned)i * q_stride;
        asm volatile("" : "+s"(so));
        unsigned voff = q_off + so;                                       // (so_base = kOOB: far outside the descriptor)
        if constexpr (TAIL != 0) {
            if (tail) voff = tail_bad ? kOOB : voff;                      // (wave-uniform branch: last group only)
        }
        MDHIP_DMA16(a_rsrc, smem + sub + (wave * A_PER + i) * 1024, voff, 0);
    };

    // ---- fragment reads ---------------------------------------------------------------------------
    // sub-buffer row = pixel index inside the tile (+ sh = -1 for tap 0); the 16-byte chunk of k-chunk c of row q sits at
    // position c ^ (q & 7)
    const int c0 = lane >> 4;
    auto a_shift = [&](int l, int sh) __attribute__((always_inline)) -> unsigned {
        return (unsigned)((wm * TM + (l & 15) + sh) * 128 + (((l >> 4) ^ (((l & 7) + sh) & 7)) << 4));
    };
    auto a_shift_now = [&](int sh) __attribute__((always_inline)) -> unsigned {
        int l = lane;
        asm volatile("" : "+v"(l));
        return a_shift(l, sh);
    };
    const unsigned z_addr = (unsigned)(ZERO_OFF + c0 * 16);
    const int b_frag_base = B_OFF + (wn * TN + (lane & 15)) * 128 + ((c0 ^ (lane & 7)) << 4);
    uint32_t vmask[AL ? 1 : FM];                   // per pixel: bit 0 = inside the batch, bit 1 = oy > 0, bit 2 = ox > 0
    unsigned a_eff[AL ? 1 : FM];                   // LDS address of the fragments of the step being read
    // aligned mode: k 0..31 addresses of the step read next (fragments 1..4 / fragment 0, each + i * 2048) and the k 32..63
    // addresses of the step being computed; the wave's flags in the tile being read: 1 inside the batch, 2 oy > 0, 4 ox > 0
    [[maybe_unused]] unsigned al_bx = 0, al_e0 = 0, al_by = 0, al_f0 = 0;
    [[maybe_unused]] int wflags = 0;
    [[maybe_unused]] unsigned al_sh[2] = {0, 0};   // shift -1 (tap 0), shift 0
    if constexpr (AL) {
        al_sh[0] = a_shift(lane, -1);
        al_sh[1] = a_shift(lane, 0);
    }
    [[maybe_unused]] const bool lane_p0 = (lane & 15) == 0;
    auto wave_flags = [&](int t) __attribute__((always_inline)) {
        const int mw = t * BM + wm * TM;           // (scalar: the wave's first output pixel)
        int f = 0;
        if (mw < p.M) {
            const int b = mw / p.HoWo;
            const int rem = mw - b * p.HoWo;
            const int y = rem / p.Wo;
            const int x = rem - y * p.Wo;
            f = 1 | (y > 0 ? 2 : 0) | (x > 0 ? 4 : 0);
        }
        wflags = __builtin_amdgcn_readfirstlane(f);
    };
    auto al_x_addresses = [&](int r, int st) __attribute__((always_inline)) {
        const unsigned a = (st == 0 ? al_sh[0] : al_sh[1]) + (unsigned)(st == 2 ? E_OFF : O_OFF);
        const int need = 1 | (r == 0 ? 2 : 0);
        al_bx = (wflags & need) == need ? a : z_addr;
        const int edge = st == 0 ? (~wflags & 4) : 0;                 // tap 0 of a wave that starts a row: lane 0 of fragment 0
        al_e0 = (edge != 0 && lane_p0) ? z_addr : al_bx;
        asm volatile("" : "+v"(al_bx), "+v"(al_e0));
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
                const int y = rem / p.Wo;
                const int x = rem - y * p.Wo;
                mask = 1u | (y > 0 ? 2u : 0u) | (x > 0 ? 4u : 0u);
            }
            vmask[i] = mask;
        }
    };
    // step st of kernel row r: 0 = tap 0 (O, one entry to the left), 1 = tap 2 (O), 2 = tap 1 (E)
    auto set_a_eff_one = [&](int r, int st, int i, unsigned a_s) __attribute__((always_inline)) {
        const unsigned need = 1u | (r == 0 ? 2u : 0u) | (st == 0 ? 4u : 0u);
        const unsigned a = a_s + (unsigned)(i * 2048);
        a_eff[i] = ((vmask[i] & need) == need) ? a : z_addr;
        asm volatile("" : "+v"(a_eff[i]));
    };
    auto read_x = [&](int i, int kk) -> frag8_t {
        if constexpr (AL) {
            const unsigned a = kk == 0 ? (i == 0 ? al_e0 : al_bx) : (i == 0 ? al_f0 : al_by);
            return *(const __attribute__((address_space(3))) frag8_t*)(const lds_char*)(a + (unsigned)(i * 2048));
        } else return *(const __attribute__((address_space(3))) frag8_t*)(const lds_char*)(a_eff[i] ^ (unsigned)(kk * 64));
    };
    auto read_w = [&](int stage, int kk, int j) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(const lds_char*)((unsigned)(stage * B_BYTES + j * 2048) +
                                                                                  (unsigned)(b_frag_base ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue: conv_v5's 8-wave one (bias from LDS per pixel row, packed SiLU, exchange, 16-byte buffer stores) -------
    const int q4 = lane >> 4;
    auto epilogue = [&](int tile_m) __attribute__((always_inline)) {
        float neg_log2e = -0x1.715476p+0f;
        asm volatile("" : "+v"(neg_log2e));
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int lp = lane_e & 15;
        const int lq = lane_e >> 4;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const int npair0 = n0 + wn * TN + lq * 8;
        const int nlast = n0 + wn * TN + (FN - 1) * 16 + lq * 4;
        const long long rows_left = (long long)p.M - (long long)tile_m * BM;
        const int ml = wm * TM + lp;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((uint16_t*)p.out + (size_t)tile_m * BM * p.ld_out), 0, (int)min(rows_left * p.ld_out * 2, 0x7fffffffLL), 0x00020000);
        const unsigned o_pair = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)npair0) * 2u;
        const unsigned o_last = ((unsigned)ml * (unsigned)p.ld_out + (unsigned)nlast) * 2u;
        const unsigned o_step = 16u * (unsigned)p.ld_out * 2u;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            f32x4 bv[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bv[j] = *(const __attribute__((address_space(3))) f32x4*)(smem + ZERO_OFF + 256 + (wn * TN + j * 16 + q4 * 4) * 4);
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 t = f32x2{acc[i][j][r], acc[i][j][r + 1]} + f32x2{bv[j][r], bv[j][r + 1]};
                    t = silu_f32x2(t, neg_log2e);
                    v[j][r] = t[0];
                    v[j][r + 1] = t[1];
                }
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j + 1 < FN; j += 2) {
                unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                const unsigned off = o_pair + (unsigned)i * o_step + (unsigned)(j * 32);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{t0[0], t1[0], t0[1], t1[1]}, o_rsrc, (int)off, 0, 0);
            }
            {
                constexpr int j = FN - 1;
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{st_pack2(v[j][0], v[j][1]), st_pack2(v[j][2], v[j][3])}, o_rsrc,
                                                      (int)(o_last + (unsigned)i * o_step), 0, 0);
            }
        }
    };

    // ---- prologue: both sub-buffers of run (first tile, group 0, r 0), weight slabs of steps 0 and 1 ----------------------
    tile_geom();
    RunGeom g_cur = run_geom();
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        dma_run_piece(O_OFF, g_cur.rsrc, g_cur.so_o, g_cur.tail, i);
        dma_run_piece(E_OFF, g_cur.rsrc, g_cur.so_e, g_cur.tail, i);
    }
    run_next();
    RunGeom g_nxt = run_geom();                      // the run after the one being consumed
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) dma_b_piece(st, i);
        dma_b_done();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
    bool c_rev = tile_rev(first_tile);             // the consumer's tile walks its kernel rows backwards
    if constexpr (AL) {
        wave_flags(first_tile);
        al_x_addresses(c_rev ? 2 : 0, 0);
    } else {
        tile_masks(first_tile);
        const unsigned a0 = a_shift_now(-1) + (unsigned)O_OFF;
#pragma unroll
        for (int i = 0; i < FM; ++i) set_a_eff_one(c_rev ? 2 : 0, 0, i, a0);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) xa[i] = read_x(i, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) wa[j] = read_w(0, 0, j);

    int c_r = 0, c_cg = 0, c_tile = first_tile, step = 0;
#define MDHIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    // a last channel group of at most 32 channels has nothing in k 32..63: its second-half MFMAs are skipped
    const bool tail_short = TAIL != 0 && (p.C8 & 7) != 0 && (p.C8 & 7) <= 4;
    constexpr int DMA_MAX = B_PER + A_PER, DMA_PER_G = (DMA_MAX + FN - 1) / FN;
    for (int run = 0; run < total_runs; ++run) {
        const bool skip_y = tail_short && c_cg == G - 1;
        const bool tile_end = c_r == 2 && c_cg == G - 1;
        const int n_r = c_r == 2 ? 0 : c_r + 1;
        const bool n_rev = tile_end ? tile_rev(c_tile + tile_step) : c_rev;     // (of the run after this one)
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int cur = step & 1;
            // the step being prefetched: tap 2 / tap 1 of this kernel row, or tap 0 of the next run; nr = its KERNEL ROW (a
            // backward tile's walk position w is kernel row 2 - w)
            const int nst = st == 2 ? 0 : st + 1;
            const int nr = st == 2 ? (n_rev ? 2 - n_r : n_r) : (c_rev ? 2 - c_r : c_r);
            unsigned a_next = 0;
            if constexpr (AL) {
                al_by = al_bx ^ 64u;                                       // the k 32..63 addresses of THIS step
                al_f0 = al_e0 ^ 64u;
                if (st == 2 && tile_end) wave_flags(c_tile + tile_step);   // (flags of a tile past the stream's end are never used)
                al_x_addresses(nr, nst);
            } else {
                if (st == 2 && tile_end) tile_masks(c_tile + tile_step);   // (masks of a tile past the stream's end are never used)
                a_next = a_shift_now(nst == 0 ? -1 : 0) + (unsigned)(nst == 2 ? E_OFF : O_OFF);
            }
            // ---- first half: k 0..31 of this step, while its k 32..63 fragments are read and the fragment addresses
            //      of the next step are selected; MFMA chunk g = fragment column g ----
#pragma unroll
            for (int g = 0; g < FN; ++g) {
                // (the weight fragment of the other k half goes into the registers the chunk before released)
                wb[(g + FN - 1) % FN] = read_w(cur, 1, (g + FN - 1) % FN);
                xb[g] = read_x(g, 1);
                if constexpr (!AL) set_a_eff_one(nr, nst, g, a_next);
                MDHIP_FENCE();
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][g] = MDHIP_MFMA(wa[g], xa[i], acc[i][g]);
                MDHIP_FENCE();
            }
            // everything this wave requested has landed; its reads of weight stage `cur` are complete
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            MDHIP_FENCE();
            // ---- second half: the k 0..31 fragments of the next step, MFMAs on k 32..63, and the DMA pieces behind the
            //      MFMA chunks: the weight slab of step + 2; tap 0: the rest of THIS run's E; tap 2: the next run's O
            //      (O is read for the last time in this step's first half); tap 1: the first pieces of the next run's E ----
#pragma unroll
            for (int g = 0; g < FN; ++g) {
                wa[(g + FN - 1) % FN] = read_w(cur ^ 1, 0, (g + FN - 1) % FN);
                xa[g] = read_x(g, 0);
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
                    else if (st == 0 && E_H0 + d - B_PER < A_PER) dma_run_piece(E_OFF, g_cur.rsrc, g_cur.so_e, g_cur.tail, E_H0 + d - B_PER);
                    else if (st == 1) dma_run_piece(O_OFF, g_nxt.rsrc, g_nxt.so_o, g_nxt.tail, d - B_PER);
                    else if (st == 2 && d - B_PER < E_H0) dma_run_piece(E_OFF, g_nxt.rsrc, g_nxt.so_e, g_nxt.tail, d - B_PER);
                }
                MDHIP_FENCE();
            }
            dma_b_done();
            ++step;
        }
        // the run is consumed: the loader moves on; maybe the tile is complete
        g_cur = g_nxt;
        run_next();
        g_nxt = run_geom();
        c_r = n_r;
        c_rev = n_rev;
        if (n_r == 0 && ++c_cg == G) {
            c_cg = 0;
            epilogue(c_tile);
            c_tile += tile_step;
        }
    }
#undef MDHIP_FENCE
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table: one configuration
// ---------------------------------------------------------------------------------------
static const ConvCfg g_cfg7 = {kBM7, kBN7, kNW7 * 64, (size_t)kLds7, 1, "v7:s2run320x160/4x2"};

int conv7_num_cfgs() { return 1; }
const ConvCfg& conv7_cfg(int) { return g_cfg7; }

hipError_t conv7_init() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_v7_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds7);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_v7_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds7);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_v7_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds7 + kAlExtra7);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_v7_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds7 + kAlExtra7);
    return e;
}

bool conv7_supports(int cfg, const ConvArgs& a) {
    // 3x3 / stride 2 / pad 1 over an even-sized map whose output rows tile the 320-pixel M tile exactly (Wo = 40, 80, 160,
    // 320) and hold whole 40-entry wave blocks; every output channel of every N tile exists (no channel test on stores);
    // activated 16-bit outputs, no residual (the stride-2 convs of the YOLOv5 family have none); at least one full
    // 64-channel group; the (at most eight) images a tile spans inside the 31-bit offset range of the run loader's descriptor
    return cfg == 0 && !a.in_f8 && !a.out_f8 && !a.out_f32 && a.res == nullptr && a.act == 1 && a.wgt4 != nullptr &&
           a.ntaps == 9 && a.kw == 3 && a.stride == 2 && a.pad == 1 && a.H == 2 * a.Ho && a.W == 2 * a.Wo &&
           a.Wo >= 40 && (kBM7 % a.Wo) == 0 && (a.Wo % 40) == 0 && a.HoWo == a.Ho * a.Wo && (a.M % a.HoWo) == 0 &&
           a.C8 >= 8 && (a.N % kBN7) == 0 && a.N == a.n_rows &&
           (9LL * a.H * a.W + 4LL * a.W + 64) * a.ld_in * 2 < 0x7fffff00LL;      // (per image: nothing here depends on the batch)
}

hipError_t conv7_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (!conv7_supports(cfg, a)) return hipErrorInvalidValue;
    ConvArgs p = a;
    conv_set_rcp(p);
    p.tiles_n = a.n_rows / kBN7;
    p.tiles_m = (a.M + kBM7 - 1) / kBM7;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, 32 / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    // same results from every instantiation: what the last channel group looks like, and whether a wave's 80 output pixels lie
    // inside one output row
    const bool tail = (a.C8 & 7) != 0, aligned = (a.Wo % 80) == 0 && a.dev_param != 77;
    if (aligned) {
        if (tail) hipLaunchKernelGGL((conv_v7_kernel<1, true>), grid, dim3(kNW7 * 64), kLds7 + kAlExtra7, s, p);
        else hipLaunchKernelGGL((conv_v7_kernel<0, true>), grid, dim3(kNW7 * 64), kLds7 + kAlExtra7, s, p);
    } else {
        if (tail) hipLaunchKernelGGL((conv_v7_kernel<1, false>), grid, dim3(kNW7 * 64), kLds7, s, p);
        else hipLaunchKernelGGL((conv_v7_kernel<0, false>), grid, dim3(kNW7 * 64), kLds7, s, p);
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
