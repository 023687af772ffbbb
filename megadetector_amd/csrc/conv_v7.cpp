// 3x3 / stride 1 convolution, row-segment reuse (conv_v5.cpp's data movement, K order and results) under a
// ROLE-SPLIT schedule: one 8-wave workgroup per CU, 80x80 wave tiles, two wave groups that alternate between a
// matrix phase and a memory phase (gfx950 / MI355X).
//
// Why (round 3, profiles/r3_*): in conv_v5 every wave mixes MFMAs, LDS fragment reads and LDS-DMA issue in one
// instruction stream.  A wave is in-order, so while it sits on a `buffer_load ... lds` that the texture path has not
// accepted yet (60 - 185 cycles per 1 KiB piece, more when the waves of a workgroup burst together after a barrier) its
// MFMAs do not issue; with two independent workgroups per CU the partner wave on the SIMD fills some of those holes by
// accident of phase.  The 8-wave tiles (conv_v5<160,320>, <320,160>) move 26 - 50 % fewer bytes per MFMA through
// L2 -> LDS and 11 % fewer through LDS -> registers and tile the 80x80 / 160x160 maps at batch 32 exactly (1280 / 2560
// tiles on 256 CUs), but run in lock step: every wave issues its DMA at the same moment and nobody is left to feed the
// matrix pipe (+ 2 ... 6 % only).  Here the two waves of a SIMD are never in the same phase:
//
//     slot      4s        4s+1      4s+2      4s+3      4s+4 ...
//     group 0   L(s,0)    M(s,0)    L(s,1)    M(s,1)    L(s+1,0)
//     group 1   M(s-1,1)  L(s,0)    M(s,0)    L(s,1)    M(s,1)
//
//   L(s,h): fragment reads of k-half h of step s (5 + 5 ds_read_b128), then this wave's share of the LDS-DMA pieces,
//           s_waitcnt lgkmcnt(0), s_barrier                                   -- no MFMA
//   M(s,h): s_setprio 1, 25 MFMAs (one 80x80 wave tile x 32 deep), s_setprio 0, s_waitcnt vmcnt(0), s_barrier
//                                                                             -- no memory instruction
// One raw s_barrier per slot for all eight waves; group 1 = waves 4..7 (the second wave of every SIMD) runs one slot
// behind (it passes one extra barrier first, group 0 one extra at the end).  A step is a 64-deep slab of K: (channel
// group, kernel row r, tap s), as in conv_v5.
//
// What is loaded when (LDS: two weight stages, two run buffers, as conv_v5):
//   * the weight slab of step s+1 goes into stage (s+1) & 1 during the L phases of step s.  The stage held slab s-1,
//     whose last fragment read (group 1, L(s-1,1)) was retired by that wave's lgkmcnt(0) before the barrier that ends
//     slot 4s-1: write-after-read safe from slot 4s on.  A piece issued in an L phase is waited for at the end of the
//     same wave's next M phase (vmcnt(0) in front of its barrier: one matrix phase of cover), so it is visible to every
//     wave from the slot after that: pieces of group 0 from both of its L phases and pieces of group 1 from L(s,0) are
//     visible by slot 4s+4, where group 0 starts reading slab s+1.  Group 1's L(s,1) is one slot too late for weights
//     and carries run pieces only.
//   * the run (channel group, kernel row) r+1 goes into run buffer (r+1) & 1 during the three steps of run r -- from
//     any L phase of group 0 and from every L phase of group 1 except the last one of the run.
//   * piece -> (phase, wave) is a static round-robin (weights: piece j to phase type j % 3 = {g0 L0, g0 L1, g1 L0},
//     SIMD (j / 3) % 4; run pieces: one per (phase, SIMD) slot in an order that fills group 1's weight-free L(.,1)
//     phases first): at most 4 + 1 pieces per phase.
//
// Same K order (group, r, s, c) and the same MFMA chain per accumulator as conv_v5 (k 0..31 then k 32..63 of every step,
// steps in order; the k 32..63 half of a half-full last channel group is skipped): bit-identical results
// (tests/test_gpu_headline.py).  Needs every weight row of the tile to exist (n_rows % BN == 0).

#include <algorithm>
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

constexpr int v7_run_pieces(int bm) { return (bm + 2 + 7) / 8; }
constexpr int v7_zero_bytes(int bn) { return (256 + bn * 4 + 1023) / 1024 * 1024; }
constexpr int v7_lds_bytes(int bm, int bn) { return 2 * v7_run_pieces(bm) * 1024 + 2 * bn * 128 + v7_zero_bytes(bn); }

// run-piece slots of a run (3 steps x 2 halves x 2 groups), in the order they are filled; -1 = not eligible
// index = (group * 3 + step) * 2 + half
constexpr int v7_run_slot_order(int group, int step, int half) {
    if (group == 1 && half == 1) return step == 0 ? 0 : (step == 1 ? 1 : -1);      // weight-free phases first
    if (group == 0 && half == 1) return 2 + step;
    if (group == 0 && half == 0) return 5 + step;
    return 8 + step;                                                                  // group 1, half 0
}
[[maybe_unused]] constexpr int kRunSlots = 11 * 4;

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// The same instruction written out, invisible to hipcc's wait-count bookkeeping.  hipcc treats an LDS-DMA builtin as a
// pending LDS store; where a `ds_read` follows one across control flow it cannot see through (the conditional pieces of
// the continuous schedule below) it inserts `s_waitcnt vmcnt(0)` in front of the read, i.e. the wave sits out the whole
// L2 / HBM latency of every piece it has just issued (found in the .s of the first version: 971 instead of 1106 TFLOP/s).
// Ordering is this file's job anyway: counted vmcnt in front of the barrier that precedes the first read of the data.
// M0 (the LDS destination) is written and restored inside the statement (MI355X guide, section 5.7).
typedef int v7_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v7_i32x4 v7_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    v7_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);     // stride 0, no swizzle
    r[2] = 0x7fffffff;                                                            // num_records (bytes)
    r[3] = 0x00020000;                                                            // raw buffer, 32-bit data format
    return r;
}
__device__ __forceinline__ void v7_dma16(v7_i32x4 rsrc, unsigned lds_byte, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff)
                 : "memory");
}

// PROF bits (developer builds): 1 = s_memtime stamps (L phase, barrier after L, M phase, barrier after M, epilogue),
// 2 = no s_setprio around the matrix phase, 4 = no stagger (both groups in the same phase: lock step)
template <int BM, int BN, int WM, int WN, int PROF = 0>
__global__ void __launch_bounds__(512, 2)
conv_v7_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    static_assert(NW == 8, "two groups of four waves");
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "16x16 fragments");
    constexpr int A_PIECES = v7_run_pieces(BM), A_BUF = A_PIECES * 1024;
    constexpr int B_BYTES = BN * 128, B_PIECES = BN / 8;
    constexpr int B_OFF = 2 * A_BUF;
    constexpr int ZERO_OFF = B_OFF + 2 * B_BYTES;
    static_assert(A_PIECES <= kRunSlots, "one run piece per (phase, SIMD) slot");
    static_assert(BN * 4 + 256 <= v7_zero_bytes(BN), "bias staging area");

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int group = wave >> 2;                    // waves w and w + 4 share a SIMD
    const int wq = wave & 3;

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
    const int G = p.groups;
    const int runs_per_tile = 3 * G;
    const int steps_per_tile = 9 * G;
    const int total_runs = my_tiles * runs_per_tile;

    // the row of zeros that invalid (pixel, tap) pairs read; behind it the bias of the workgroup's BN channels
    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < BN; c += NW * 64)
        *(__attribute__((address_space(3))) float*)(smem + ZERO_OFF + 256 + c * 4) = (n0 + c < p.n_rows) ? p.bias[n0 + c] : 0.f;

    // ---- weight stream: slab `step` = 128 bytes of every row at byte offset step * 128 ------------------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt4 + (size_t)n0 * p.k_pad4), 0, kNumRecords, 0x00020000);
    const unsigned b_voff = (unsigned)(lr * p.k_pad4 + jj * 8) * 2u;       // this lane inside piece 0; piece j: + j * b_piece
    const int b_piece = 8 * p.k_pad4 * 2;
    // piece j of the weight slab `lstep` into stage `stage`
    auto dma_w = [&](int stage, int j, int lstep) __attribute__((always_inline)) {
        MDHIP_DMA16(b_rsrc, smem + B_OFF + stage * B_BYTES + j * 1024, b_voff, j * b_piece + lstep * 128);
    };

    // ---- run loader: one (group, kernel row) ahead of the consumer ------------------------------------
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    const unsigned q_voff = (unsigned)(lr * p.ld_in * 2 + jj * 16);
    const int q_piece = 8 * p.ld_in * 2;
    int lg_tile = first_tile, lg_cg = 0, lg_r = 0;
    bool lg_live = true;
    int lg_first = 0;
    unsigned lg_soff = 0;
    auto run_tile = [&](int t) __attribute__((always_inline)) {
        const long long origin = (long long)t * BM - p.W - 1;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + origin * p.ld_in), 0, kNumRecords, 0x00020000);
    };
    auto run_setup = [&]() __attribute__((always_inline)) {
        lg_first = lg_tile * BM + (lg_r - 1) * p.W - 1;
        lg_soff = (unsigned)(lg_r * p.W * p.ld_in * 2 + lg_cg * 128);
    };
    auto dma_run = [&](int buf, int j) __attribute__((always_inline)) {
        const int q = j * 8 + lr;
        const bool ok = lg_live && (unsigned)(lg_first + q) < (unsigned)p.M && lg_cg * 8 + jj < p.C8;
        MDHIP_DMA16(a_rsrc, smem + buf * A_BUF + j * 1024, ok ? q_voff : kOOB, lg_soff + (unsigned)(j * q_piece));
    };
    auto run_next = [&]() __attribute__((always_inline)) {
        if (++lg_r == 3) {
            lg_r = 0;
            if (++lg_cg == G) {
                lg_cg = 0;
                if (lg_tile == last_tile) lg_live = false;
                else { lg_tile += tile_step; run_tile(lg_tile); }
            }
        }
        run_setup();
    };

    // ---- fragment reads (conv_v5's LDS image) -----------------------------------------------------------
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
    auto read_x = [&](int i, int kk) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + (a_eff[i] ^ (unsigned)(kk * 64)));
    };
    auto read_w = [&](int stage, int kk, int j) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + stage * B_BYTES + j * 2048 +
                                                                 (b_frag_base ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue (conv_v5's: bias from LDS, pixel-row order, 16-byte stores, residual in the store layout) ----
    const int q4 = lane >> 4;
    auto epilogue_t = [&](int tile_m, auto has_res_t) __attribute__((always_inline)) {
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        const int m0 = tile_m * BM + wm * TM + (lane & 15);
        const int nbase = n0 + wn * TN + q4 * 4;
        float bv[FN][4];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 g = *(const __attribute__((address_space(3))) f32x4*)(smem + ZERO_OFF + 256 + (wn * TN + j * 16 + q4 * 4) * 4);
            bv[j][0] = g[0]; bv[j][1] = g[1]; bv[j][2] = g[2]; bv[j][3] = g[3];
        }
        constexpr int NPAIR = FN / 2;
        uint4 rpair[2][NPAIR > 0 ? NPAIR : 1];
        uint2 rlast[2];
        auto fetch_res_row = [&](int i, uint4 (&rp)[NPAIR > 0 ? NPAIR : 1], uint2& rl) {
            const int m = min(m0 + i * 16, p.M - 1);
            const uint16_t* rrow_p = p.res + (size_t)m * p.ld_res;
#pragma unroll
            for (int jp = 0; jp < NPAIR; ++jp)
                rp[jp] = *(const uint4*)(rrow_p + min(n0 + wn * TN + jp * 32 + q4 * 8, p.N - 8));
            if (FN & 1) rl = *(const uint2*)(rrow_p + min(nbase + (FN - 1) * 16, p.N - 4));
        };
        if constexpr (HAS_RES) fetch_res_row(0, rpair[0], rlast[0]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (HAS_RES) {
                if (i + 1 < FM) fetch_res_row(i + 1, rpair[(i + 1) & 1], rlast[(i + 1) & 1]);
            }
            const int m = m0 + i * 16;
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] + bv[j][r];
                    if (p.act) t = silu_f32(t);
                    v[j][r] = t;
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
                    const uint4 d = rpair[i & 1][jp];
                    auto s0 = __builtin_amdgcn_permlane16_swap(d.x, d.z, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(d.y, d.w, false, false);
                    auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s0[1], false, false);
                    auto a1 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
                    add4(2 * jp, a0[0], a1[0]);
                    add4(2 * jp + 1, a0[1], a1[1]);
                }
                if (FN & 1) add4(FN - 1, rlast[i & 1].x, rlast[i & 1].y);
            }
            uint16_t* orow = (uint16_t*)p.out + (size_t)m * p.ld_out;
#pragma unroll
            for (int j = 0; j + 1 < FN; j += 2) {
                unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                const int n = n0 + wn * TN + j * 16 + q4 * 8;
                if (m < p.M && n < p.N) *(uint4*)(orow + n) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            }
            if (FN & 1) {
                const int j = FN - 1;
                const int n = nbase + j * 16;
                uint2 o;
                o.x = st_pack2(v[j][0], v[j][1]);
                o.y = st_pack2(v[j][2], v[j][3]);
                if (m < p.M && n < p.N) *(uint2*)(orow + n) = o;
            }
        }
    };
    auto epilogue = [&](int tile_m) __attribute__((always_inline)) {
        if (p.res) epilogue_t(tile_m, std::true_type{});
        else epilogue_t(tile_m, std::false_type{});
    };

    // ---- prologue: run (first tile, group 0, r 0) in buffer 0, weight slab 0 in stage 0 --------------------
    run_tile(first_tile);
    run_setup();
    for (int j = wave; j < A_PIECES; j += NW) dma_run(0, j);
    run_next();
    for (int j = wave; j < B_PIECES; j += NW) dma_w(0, j, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    tile_masks(first_tile);

    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr ((PROF & 1) != 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[k] += t - t_prev;
            t_prev = t;
        }
    };
    if constexpr ((PROF & 1) != 0) t_prev = __builtin_amdgcn_s_memtime();

    // stagger: the second wave of every SIMD runs one slot behind (its extra barrier pairs with the barrier that ends
    // group 0's first L phase; group 0 passes the matching one after its last phase)
    if constexpr ((PROF & 4) == 0) {
        if (group == 1) __builtin_amdgcn_s_barrier();
    }
    stamp(1);

#define MDHIP_FENCE() __builtin_amdgcn_sched_barrier(0)
    const bool tail_short = (p.C8 & 7) != 0 && (p.C8 & 7) <= 4;
    frag8_t fx[FM], fw[FN];
    int c_r = 0, c_cg = 0, c_tile = first_tile, pa = 0;
    int w_step = 0;                                 // step inside the tile of the slab being consumed
    int gpar = 0;                                   // weight stage of that slab (steps_per_tile may be odd: not w_step & 1)
    unsigned long long n_steps = 0;
    for (int run = 0; run < total_runs; ++run) {
        const bool skip_y = tail_short && c_cg == G - 1;
        const bool tile_end = c_r == 2 && c_cg == G - 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int cur = gpar;
            const int nstep = (w_step + 1 == steps_per_tile) ? 0 : w_step + 1;      // the slab loaded during this step
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // ================= L(s, h): fragment reads, then this wave's DMA pieces =================
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const unsigned a = a_sh[s] + (unsigned)(pa * A_BUF + i * 2048);
                        a_eff[i] = ((vmask[i] >> (c_r * 3 + s)) & 1u) ? a : z_addr;
                    }
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) fw[j] = read_w(cur, h, j);
#pragma unroll
                for (int i = 0; i < FM; ++i) fx[i] = read_x(i, h);
                MDHIP_FENCE();
                // weight pieces of the next slab: phase types 0 = (g0, L0), 1 = (g0, L1), 2 = (g1, L0)
                {
                    const int t = group == 0 ? h : (h == 0 ? 2 : -1);
                    if (t >= 0) {
#pragma unroll
                        for (int n = 0; n * 12 < B_PIECES; ++n) {
                            const int j = n * 12 + wq * 3 + t;
                            if (j < B_PIECES) dma_w(cur ^ 1, j, nstep);
                        }
                    }
                }
                // at most one run piece per phase: slot u = order(group, s, h) * 4 + SIMD
                {
                    const int o0 = v7_run_slot_order(0, s, h), o1 = v7_run_slot_order(1, s, h);
                    const int o = group == 0 ? o0 : o1;
                    if (o >= 0) {
                        const int u = o * 4 + wq;
                        if (u < A_PIECES) dma_run(pa ^ 1, u);
                    }
                }
                MDHIP_FENCE();
                stamp(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                MDHIP_FENCE();
                stamp(1);
                // ================= M(s, h): 25 MFMAs, nothing else =================
                if (!(h == 1 && skip_y)) {
                    if constexpr ((PROF & 2) == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int g = 0; g < FN; ++g)
#pragma unroll
                        for (int i = 0; i < FM; ++i)
                            acc[i][g] = MDHIP_MFMA(fw[g], fx[i], acc[i][g]);
                    if constexpr ((PROF & 2) == 0) __builtin_amdgcn_s_setprio(0);
                }
                MDHIP_FENCE();
                stamp(2);
                if (h == 1 && s == 2 && tile_end) {
                    // the tile is complete: its epilogue runs inside this slot (the other group waits at the barrier
                    // or runs its own, one slot later)
                    epilogue(c_tile);
                    tile_masks(c_tile + tile_step);
                    stamp(4);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                MDHIP_FENCE();
                stamp(3);
            }
            w_step = nstep;
            gpar ^= 1;
            ++n_steps;
        }
        // the run is consumed: the loader moves on
        run_next();
        pa ^= 1;
        c_r = c_r == 2 ? 0 : c_r + 1;
        if (c_r == 0 && ++c_cg == G) {
            c_cg = 0;
            c_tile += tile_step;
        }
    }
#undef MDHIP_FENCE
    if constexpr ((PROF & 4) == 0) {
        if (group == 0) __builtin_amdgcn_s_barrier();
    }
    if constexpr ((PROF & 1) != 0) {
        if (lane == 0 && p.dbg) {
            unsigned long long* d = (unsigned long long*)p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = t_acc[k];
            d[6] = n_steps;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}


// =====================================================================================================================
// CONTINUOUS-DMA schedule (the shipped kernel of this file).
//
// conv_v5's step (two halves of 25 MFMAs per wave, fragments double-buffered in registers, one barrier in the middle)
// with the 8-wave 80x80-wave-tile geometry, and the LDS-DMA pieces taken OUT of the burst after the barrier.  Measured
// on the lock-step 8-wave conv_v5<320,160> (profiles/r3_convbench_ablate.txt): the half that carries the DMA takes 1505
// cycles against 689 for the half without -- 816 cycles for 4.2 pieces per wave.  That is the texture path draining the
// workgroup's burst (8 waves x ~6 pieces x 16 cycles per 1 KiB piece) while every wave of the CU sits on its
// `buffer_load ... lds` and no MFMA issues: the DMA path is only ~55 % utilised over a step, but all of its work is
// requested in the same few hundred cycles.  Here
//   * a step's pieces are spread over ALL TEN MFMA chunks of the step (both halves), one piece at a time, and the
//     chunk a wave issues its k-th piece at is rotated by the wave (rot = wave * 10 / 8: the two waves of a SIMD are
//     half a step apart) -- the texture path sees a steady trickle instead of a burst;
//   * which needs the weight slab s+2 to be loadable during the FIRST half of step s, while slab s is still being read:
//     NST = 3 weight stages (slab s in stage s % 3; 320x160 tiles: 2 x 41 + 3 x 20 + 1 = 143 KiB).  With NST = 2
//     (160x320: no room for a third 40 KiB stage) weight pieces stay in the second half, spread over its five chunks,
//     and only the run pieces use the first half;
//   * the wait in front of the mid-step barrier is COUNTED: vmcnt(n) with n = the pieces this wave issued in the first
//     half of this step (they belong to slab s+2 / the next run and may stay in flight); everything older -- slab s+1,
//     read right after the barrier, and in step 2 of a run the whole next run -- has landed.
// Same K order, fragment reads, MFMA chains and epilogue as conv_v5: bit-identical results.
// =====================================================================================================================
namespace {
constexpr int v7c_lds_bytes(int bm, int bn, int nst) { return 2 * v7_run_pieces(bm) * 1024 + nst * bn * 128 + v7_zero_bytes(bn); }
// bit mask over chunk positions: `n` pieces at positions (i * span) / n of a span
constexpr unsigned v7c_mask(int n, int span) {
    unsigned m = 0;
    for (int i = 0; i < n; ++i) m |= 1u << ((i * span) / n);
    return m;
}
}  // namespace

// PROF bits: 1 = s_memtime stamps (as conv_v5), 8 = no rotation (every wave issues at the same chunks), 16 = no DMA in
// the steady state (timing only, wrong results)
template <int BM, int BN, int WM, int WN, int NST, int PROF = 0>
__global__ void __launch_bounds__(512, 2)
conv_v7c_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    static_assert(NW == 8 && (NST == 2 || NST == 3), "8 waves, 2 or 3 weight stages");
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    static_assert(TM % 16 == 0 && TN % 16 == 0 && FN == 5, "16x16 fragments, five chunks per half");
    constexpr int A_PIECES = v7_run_pieces(BM), A_BUF = A_PIECES * 1024;
    constexpr int A_PER = (A_PIECES + NW - 1) / NW;          // run pieces per wave (the last one may not exist)
    constexpr int A_H0 = (A_PER + 1) / 2;                    // issued during step 0 of a run; the rest during step 1
    constexpr int B_BYTES = BN * 128, B_PIECES = BN / 8, B_PER = (B_PIECES + NW - 1) / NW;
    constexpr int B_OFF = 2 * A_BUF;
    constexpr int ZERO_OFF = B_OFF + NST * B_BYTES;
    static_assert(BN * 4 + 256 <= v7_zero_bytes(BN), "bias staging area");
    static_assert(A_H0 <= 10 && B_PER <= (NST == 3 ? 10 : 5), "one piece per chunk position");
    // chunk positions (0..4 first half, 5..9 second half) at which a wave issues a weight piece / a run piece: B_PER
    // (A_H0) positions spread evenly, rotated per wave; the k-th position issues the wave's k-th piece of the step
    constexpr unsigned WMASK = NST == 3 ? v7c_mask(B_PER, 10) : v7c_mask(B_PER, 5);
    constexpr unsigned RMASK = v7c_mask(A_H0, 10);

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int rot10 = (PROF & 8) ? 0 : (wave * 10) / NW;         // waves w and w + 4 (one SIMD): half a step apart
    const int rot5 = (PROF & 8) ? 0 : (wave * 5) / NW;
    const unsigned rmask = ((RMASK << rot10) | (RMASK >> (10 - rot10))) & 0x3ffu;
    const unsigned wmask = NST == 3 ? (((WMASK << rot10) | (WMASK >> (10 - rot10))) & 0x3ffu)
                                    : ((((WMASK << rot5) | (WMASK >> (5 - rot5))) & 0x1fu) << 5);

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
    const int G = p.groups;
    const int runs_per_tile = 3 * G;
    const int steps_per_tile = 9 * G;
    const int total_runs = my_tiles * runs_per_tile;

    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < BN; c += NW * 64)
        *(__attribute__((address_space(3))) float*)(smem + ZERO_OFF + 256 + c * 4) = (n0 + c < p.n_rows) ? p.bias[n0 + c] : 0.f;

    // ---- weight stream ----------------------------------------------------------------------------------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt4 + (size_t)n0 * p.k_pad4), 0, kNumRecords, 0x00020000);
    const v7_i32x4 b_rsrc_a = v7_rsrc(p.wgt4 + (size_t)n0 * p.k_pad4);
    const unsigned b_voff = (unsigned)((wave * 8 + lr) * p.k_pad4 + jj * 8) * 2u;      // this wave's piece 0
    const int b_stride = NW * 8 * p.k_pad4 * 2;                                        // to its next piece
    int l_step = 0;                                // step inside the tile of the slab the loader is at
    int l_stage = 0;                               // and the stage it goes to
    int n_inflight = 0;                            // pieces this wave issued since the last counted wait
    // this wave's i-th piece (i may be a run-time scalar) of the loader's slab
    auto dma_w = [&](int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if (i * NW + wave < B_PIECES) {                                                  // wave-uniform
            v7_dma16(b_rsrc_a, (unsigned)(B_OFF + l_stage * B_BYTES + (i * NW + wave) * 1024), b_voff, (unsigned)(i * b_stride + l_step * 128));
            ++n_inflight;
        }
    };
    auto w_next = [&]() __attribute__((always_inline)) {
        l_step = (l_step + 1 == steps_per_tile) ? 0 : l_step + 1;
        l_stage = (l_stage + 1 == NST) ? 0 : l_stage + 1;
    };

    // ---- run loader: one (group, kernel row) ahead of the consumer ------------------------------------
    v7_i32x4 a_rsrc_a = b_rsrc_a;
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    const unsigned q_voff = (unsigned)((wave * 8 + lr) * p.ld_in * 2 + jj * 16);
    const int q_stride = NW * 8 * p.ld_in * 2;
    int lg_tile = first_tile, lg_cg = 0, lg_r = 0;
    bool lg_live = true;
    int lg_first = 0;
    unsigned lg_soff = 0;
    auto run_tile = [&](int t) __attribute__((always_inline)) {
        const long long origin = (long long)t * BM - p.W - 1;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + origin * p.ld_in), 0, kNumRecords, 0x00020000);
        a_rsrc_a = v7_rsrc(p.in + origin * p.ld_in);
    };
    auto run_setup = [&]() __attribute__((always_inline)) {
        lg_first = lg_tile * BM + (lg_r - 1) * p.W - 1;
        lg_soff = (unsigned)(lg_r * p.W * p.ld_in * 2 + lg_cg * 128);
    };
    auto dma_run = [&](int buf, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        if (i * NW + wave < A_PIECES) {                                                  // wave-uniform
            const int q = (i * NW + wave) * 8 + lr;
            const bool ok = lg_live && (unsigned)(lg_first + q) < (unsigned)p.M && lg_cg * 8 + jj < p.C8;
            v7_dma16(a_rsrc_a, (unsigned)(buf * A_BUF + (i * NW + wave) * 1024), ok ? q_voff : kOOB, lg_soff + (unsigned)(i * q_stride));
            ++n_inflight;
        }
    };
    auto run_next = [&]() __attribute__((always_inline)) {
        if (++lg_r == 3) {
            lg_r = 0;
            if (++lg_cg == G) {
                lg_cg = 0;
                if (lg_tile == last_tile) lg_live = false;
                else { lg_tile += tile_step; run_tile(lg_tile); }
            }
        }
        run_setup();
    };

    // ---- fragment reads (conv_v5's LDS image) -----------------------------------------------------------
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
    };
    auto read_x = [&](int i, int kk) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + (a_eff[i] ^ (unsigned)(kk * 64)));
    };
    auto read_w = [&](int stage, int kk, int j) -> frag8_t {
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + stage * B_BYTES + j * 2048 +
                                                                 (b_frag_base ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue (conv_v5's) ----------------------------------------------------------------------------
    const int q4 = lane >> 4;
    auto epilogue_t = [&](int tile_m, auto has_res_t) __attribute__((always_inline)) {
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        const int m0 = tile_m * BM + wm * TM + (lane & 15);
        const int nbase = n0 + wn * TN + q4 * 4;
        float bv[FN][4];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 g = *(const __attribute__((address_space(3))) f32x4*)(smem + ZERO_OFF + 256 + (wn * TN + j * 16 + q4 * 4) * 4);
            bv[j][0] = g[0]; bv[j][1] = g[1]; bv[j][2] = g[2]; bv[j][3] = g[3];
        }
        constexpr int NPAIR = FN / 2;
        uint4 rpair[2][NPAIR];
        uint2 rlast[2];
        auto fetch_res_row = [&](int i, uint4 (&rp)[NPAIR], uint2& rl) {
            const int m = min(m0 + i * 16, p.M - 1);
            const uint16_t* rrow_p = p.res + (size_t)m * p.ld_res;
#pragma unroll
            for (int jp = 0; jp < NPAIR; ++jp)
                rp[jp] = *(const uint4*)(rrow_p + min(n0 + wn * TN + jp * 32 + q4 * 8, p.N - 8));
            rl = *(const uint2*)(rrow_p + min(nbase + (FN - 1) * 16, p.N - 4));
        };
        if constexpr (HAS_RES) fetch_res_row(0, rpair[0], rlast[0]);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (HAS_RES) {
                if (i + 1 < FM) fetch_res_row(i + 1, rpair[(i + 1) & 1], rlast[(i + 1) & 1]);
            }
            const int m = m0 + i * 16;
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] + bv[j][r];
                    if (p.act) t = silu_f32(t);
                    v[j][r] = t;
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
                    const uint4 d = rpair[i & 1][jp];
                    auto s0 = __builtin_amdgcn_permlane16_swap(d.x, d.z, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(d.y, d.w, false, false);
                    auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s0[1], false, false);
                    auto a1 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
                    add4(2 * jp, a0[0], a1[0]);
                    add4(2 * jp + 1, a0[1], a1[1]);
                }
                add4(FN - 1, rlast[i & 1].x, rlast[i & 1].y);
            }
            uint16_t* orow = (uint16_t*)p.out + (size_t)m * p.ld_out;
#pragma unroll
            for (int j = 0; j + 1 < FN; j += 2) {
                unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                const int n = n0 + wn * TN + j * 16 + q4 * 8;
                if (m < p.M && n < p.N) *(uint4*)(orow + n) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            }
            {
                const int j = FN - 1;
                const int n = nbase + j * 16;
                uint2 o;
                o.x = st_pack2(v[j][0], v[j][1]);
                o.y = st_pack2(v[j][2], v[j][3]);
                if (m < p.M && n < p.N) *(uint2*)(orow + n) = o;
            }
        }
    };
    auto epilogue = [&](int tile_m) __attribute__((always_inline)) {
        if (p.res) epilogue_t(tile_m, std::true_type{});
        else epilogue_t(tile_m, std::false_type{});
    };

    // ---- prologue: run (first tile, group 0, r 0) in buffer 0, weight slabs 0 and 1 in stages 0 and 1 -----------
    run_tile(first_tile);
    run_setup();
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        if (i * NW + wave < A_PIECES) {
            const int q = (i * NW + wave) * 8 + lr;
            const bool ok = (unsigned)(lg_first + q) < (unsigned)p.M && jj < p.C8;
            v7_dma16(a_rsrc_a, (unsigned)((i * NW + wave) * 1024), ok ? q_voff : kOOB, lg_soff + (unsigned)(i * q_stride));
        }
    }
    run_next();
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            if (i * NW + wave < B_PIECES)
                v7_dma16(b_rsrc_a, (unsigned)(B_OFF + st * B_BYTES + (i * NW + wave) * 1024), b_voff, (unsigned)(i * b_stride + l_step * 128));
        w_next();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
    tile_masks(first_tile);
#pragma unroll
    for (int i = 0; i < FM; ++i) set_a_eff_one(0, 0, 0, i);
#pragma unroll
    for (int i = 0; i < FM; ++i) xa[i] = read_x(i, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) wa[j] = read_w(0, 0, j);

    int c_r = 0, c_cg = 0, c_tile = first_tile, pa = 0, step = 0;
    int cst = 0;                                   // weight stage of the step being computed; the next step's is cst + 1
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
    // the DMA piece(s) of chunk position `pos` (0..4 first half, 5..9 second half) of step s of the current run
    int wi = 0, ri = 0;                            // pieces of this step issued so far (weights / run)
    auto dma_at = [&](int pos, int s) __attribute__((always_inline)) {
        if ((NST == 3 || pos >= 5) && ((wmask >> pos) & 1u)) { dma_w(wi); ++wi; }
        if (s < 2 && ((rmask >> pos) & 1u)) {
            if (s == 0 || A_H0 + ri < A_PER) dma_run(pa ^ 1, s == 0 ? ri : A_H0 + ri);
            ++ri;
        }
    };
    const bool tail_short = (p.C8 & 7) != 0 && (p.C8 & 7) <= 4;
    for (int run = 0; run < total_runs; ++run) {
        const bool skip_y = tail_short && c_cg == G - 1;
        const bool tile_end = c_r == 2 && c_cg == G - 1;
        const int n_r = c_r == 2 ? 0 : c_r + 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int nst_ = (cst + 1 == NST) ? 0 : cst + 1;         // stage of the next step's slab
            const int ns = (s + 1) % 3;
            const int nbuf = s == 2 ? pa ^ 1 : pa;
            const int nr = s == 2 ? n_r : c_r;
            if (s == 2 && tile_end) tile_masks(c_tile + tile_step);
            n_inflight = 0;
            wi = 0;
            ri = 0;
            // ---- first half: k 0..31 of this step; reads of its k 32..63 fragments (weight fragments streamed into
            //      the registers the chunk before released); pieces of slab step+2 (NST = 3) and of the next run ----
#pragma unroll
            for (int g = 0; g < FN; ++g) {
                wb[(g + FN - 1) % FN] = read_w(cst, 1, (g + FN - 1) % FN);
                if (g < FM) { xb[g] = read_x(g, 1); set_a_eff_one(nbuf, nr, ns, g); }
                if (g == FN - 1) {
#pragma unroll
                    for (int i = FN; i < FM; ++i) { xb[i] = read_x(i, 1); set_a_eff_one(nbuf, nr, ns, i); }
                }
                MDHIP_FENCE();
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][g] = MDHIP_MFMA(wa[g], xa[i], acc[i][g]);
                MDHIP_FENCE();
                dma_at(g, s);
                MDHIP_FENCE();
            }
            stamp(0);
            // everything older than this half's own pieces has landed (slab step+1; in step 2 the next run); this
            // wave's reads of stage cst are complete
            if (n_inflight == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (n_inflight == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            else if (n_inflight == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else if (n_inflight == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
            else if (n_inflight == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            stamp(1);
            __builtin_amdgcn_s_barrier();
            stamp(2);
            MDHIP_FENCE();
            // ---- second half: k 32..63; reads of the next step's k 0..31 fragments; the rest of the pieces ----
#pragma unroll
            for (int g = 0; g < FN; ++g) {
                wa[(g + FN - 1) % FN] = read_w(nst_, 0, (g + FN - 1) % FN);
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
                dma_at(5 + g, s);
                MDHIP_FENCE();
            }
            w_next();
            cst = nst_;
            ++step;
            stamp(3);
        }
        run_next();
        pa ^= 1;
        c_r = n_r;
        if (n_r == 0 && ++c_cg == G) {
            c_cg = 0;
            epilogue(c_tile);
            c_tile += tile_step;
        }
        stamp(5);
    }
#undef MDHIP_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
// continuous-DMA kernel: id (local), BM, BN, waves along M, waves along N, weight stages, PROF bits
#define MDHIP_CONV7C_CFGS(X)   \
    X(0, 320, 160, 4, 2, 3, 0) \
    X(1, 160, 320, 2, 4, 2, 0)
#define MDHIP_CONV7C_PROF(X)   \
    X(2, 320, 160, 4, 2, 3, 1) \
    X(3, 320, 160, 4, 2, 3, 8) \
    X(4, 320, 160, 4, 2, 3, 16) \
    X(5, 160, 320, 2, 4, 2, 1)
// role-split kernel (developer variants only: measured slower than the lock-step conv_v5 tiles, see the header)
#define MDHIP_CONV7S_PROF(X) \
    X(6, 160, 320, 2, 4, 0)  \
    X(7, 320, 160, 4, 2, 0)  \
    X(8, 160, 320, 2, 4, 1)  \
    X(9, 160, 320, 2, 4, 4)

static const ConvCfg g_cfgs7[] = {
#define X(id, bm, bn, wm, wn, nst, prof) \
    {bm, bn, 512, (size_t)v7c_lds_bytes(bm, bn, nst), 1, "v7:cont" #bm "x" #bn "/" #wm "x" #wn "/s" #nst "/" #prof},
    MDHIP_CONV7C_CFGS(X) MDHIP_CONV7C_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof) \
    {bm, bn, 512, (size_t)v7_lds_bytes(bm, bn), 1, "v7:split" #bm "x" #bn "/" #wm "x" #wn "/" #prof},
    MDHIP_CONV7S_PROF(X)
#undef X
};
constexpr int kNumMain7 = 2;
constexpr int kNumAll7 = (int)(sizeof(g_cfgs7) / sizeof(g_cfgs7[0]));

int conv7_num_cfgs() { return kNumMain7; }
const ConvCfg& conv7_cfg(int i) { return g_cfgs7[i]; }

hipError_t conv7_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, bn, wm, wn, nst, prof)                                                         \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v7c_kernel<bm, bn, wm, wn, nst, prof>,         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs7[id].lds_bytes);
    MDHIP_CONV7C_CFGS(X) MDHIP_CONV7C_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                              \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v7_kernel<bm, bn, wm, wn, prof>,                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs7[id].lds_bytes);
    MDHIP_CONV7S_PROF(X)
#undef X
    return e;
}

bool conv7_supports(int cfg, const ConvArgs& a) {
    if (cfg < 0 || cfg >= kNumAll7) return false;
    const ConvCfg& c = g_cfgs7[cfg];
    return a.wgt4 != nullptr && a.ntaps == 9 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W &&
           a.C8 >= 8 && (a.N % 8) == 0 && (a.n_rows % c.bn) == 0 && a.N == a.n_rows && !a.out_f32 && !a.in_f8 && !a.out_f8 &&
           (long long)(2 * a.W + c.bm + 16) * a.ld_in * 2 + 4096 < 0x7fffffffLL &&
           (long long)c.bn * a.k_pad4 * 2 + 4096 < 0x7fffffffLL;
}

hipError_t conv7_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (!conv7_supports(cfg, a)) return hipErrorInvalidValue;
    const ConvCfg& c = g_cfgs7[cfg];
    ConvArgs p = a;
    p.tiles_n = a.n_rows / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, 32 / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    switch (cfg) {
#define X(id, bm, bn, wm, wn, nst, prof)                                                          \
    case id:                                                                                    \
        hipLaunchKernelGGL((conv_v7c_kernel<bm, bn, wm, wn, nst, prof>), grid, dim3(512), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV7C_CFGS(X) MDHIP_CONV7C_PROF(X)
#undef X
#define X(id, bm, bn, wm, wn, prof)                                                               \
    case id:                                                                                    \
        hipLaunchKernelGGL((conv_v7_kernel<bm, bn, wm, wn, prof>), grid, dim3(512), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV7S_PROF(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
