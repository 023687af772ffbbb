// Implicit-GEMM convolution, third generation main loop (gfx950 / MI355X): 32-deep K slabs in a
// 4-stage LDS ring, three slabs of LDS-DMA in flight behind counted vmcnt waits.
//
// Same GEMM view, operand order and epilogue contract as conv_igemm.cpp (read its header first).
// Why this schedule (measurements on MI355X, tools/convbench.cpp, shape M 204800 N 320 K 2880):
//   MFMA-only main loop of a 160x160 tile, 2 workgroups/CU     1.9 PFLOP/s
//   + LDS fragment reads                                          1.84
//   + HBM/L2 -> LDS DMA (2 stages of 64-deep slabs)              1.07   <- the loss
//   the same DMA stream with no MFMAs at all takes as long as the MFMAs alone: with two stages only
//   ONE slab per workgroup can be in flight (the other is being consumed), the L2 -> LDS round trip
//   under load is ~1.9k cycles, so every step waits for memory and nothing overlaps.
// Bytes in flight is what buys bandwidth and latency tolerance.  Here the same 80 KiB of LDS per
// workgroup is cut into four stages of 32-deep slabs: while slab p feeds the MFMAs (from
// registers) and slab p+1 is being read into the other register set, slabs p+2, p+3, p+4 are in
// flight.  A wave waits for its own pieces of slab p+1 with `s_waitcnt vmcnt(2 * pieces)` (never 0
// in steady state), one raw s_barrier per slab publishes them and frees stage p & 3 for slab p+4.
// Epilogue stores go through a buffer descriptor so that out-of-range lanes are dropped by the
// hardware instead of branched around: the number of VMEM operations per epilogue is exact, and the
// counted waits after a tile boundary account for them instead of draining them.
//
// LDS image of a stage: [BM rows | BN rows] x 64 B (32 bf16 of K), written by 1 KiB DMA pieces of
// 16 rows; the 16-byte chunk c of row r is stored at chunk c ^ (3 * ((r >> 3) & 1)), which makes
// the ds_read_b128 fragment reads (16 rows x one chunk per 16-lane group) bank-conflict free.

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

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;        // >= every descriptor's num_records: reads zeros / write dropped
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;

__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

constexpr int kStages = 4;
constexpr int v3_lds_bytes(int bm, int bn) { return kStages * (bm + bn) * 64; }
constexpr int v3_blocks_per_cu(int bm, int bn, int nw) {
    int b = 163840 / v3_lds_bytes(bm, bn);
    if (b > 32 / nw) b = 32 / nw;
    if (b > 2) b = 2;
    return b < 1 ? 1 : b;
}
constexpr int v3_waves_per_simd(int bm, int bn, int nw) {
    int w = v3_blocks_per_cu(bm, bn, nw) * nw / 4;
    return w < 1 ? 1 : w;
}

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

// PROF bits (developer builds only, tools/convbench.cpp): 2 = no stores, 4 = no SiLU, 16 = no DMA in
// the steady state, 32 = no fragment reads, 64 = no MFMAs
template <int BM, int BN, int WM, int WN, int PROF = 0>
__global__ void __launch_bounds__(WM * WN * 64, v3_waves_per_simd(BM, BN, WM * WN))
conv_v3_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int A_PIECES = BM / 16, PIECES = (BM + BN) / 16;    // 1 KiB DMA pieces: 16 rows x 64 B
    constexpr int PPW = PIECES / NW;                              // pieces per wave per slab
    constexpr int EPI_OPS = FM * FN;                              // VMEM stores of one epilogue
    static_assert(PIECES % NW == 0, "the pieces of a stage must split evenly over the waves");
    static_assert(BM % 16 == 0 && BN % 16 == 0 && TM % 16 == 0 && TN % 16 == 0, "16-row granularity");
    static_assert(2 * PPW + EPI_OPS < 64, "vmcnt is a 6-bit counter");

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
    const int KT = p.k_pad >> 5;                  // 32-deep slabs per tile (k_pad is a multiple of 64)
    const int total_steps = my_tiles * KT;

    // ---- loader: this wave's pieces q = i * NW + wave; q < A_PIECES is an activation piece ------
    const int prow = lane >> 2;                               // row inside a 16-row piece
    const int jj = (lane & 3) ^ (3 * ((lane >> 5) & 1));      // swizzled source chunk inside the 64-byte slab row
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt + (size_t)n0 * p.k_pad), 0, kNumRecords, 0x00020000);
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    unsigned off[PPW];          // byte offset of the row's first chunk: (tap 0, channel 0) for A, k = 0 for B
    uint32_t mask[PPW];         // A: bit t set when tap t of the row is inside the image; B: 1 when the row exists
    bool is_a[PPW];             // wave-uniform
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = i * NW + wave;
        is_a[i] = q < A_PIECES;
        off[i] = 0;
        mask[i] = 0;
        if (!is_a[i]) {
            const int row = (q - A_PIECES) * 16 + prow;
            off[i] = (unsigned)(row * p.k_pad + jj * 8) * 2u;
            mask[i] = (n0 + row < p.n_rows) ? 1u : 0u;
        }
    }
    // position of this lane's 16-byte chunk inside K (C8 >= 4: at most one tap wrap per slab)
    int c8 = 0, ts = 0;
    uint32_t tapbit = 1;
    unsigned tapoff = 0, b_run = 0;
    int l_kt = 0, l_tile = first_tile;
    bool l_live = true;
    const int kh = p.ntaps / p.kw;
    const unsigned wrap_c = (unsigned)(p.ld_in * 2 - p.C8 * 16);
    const unsigned wrap_r = (unsigned)((p.W - p.kw) * p.ld_in * 2);

    auto init_tile = [&](int tile_m) __attribute__((always_inline)) {
        const int m0 = tile_m * BM;
        const int b0 = m0 / p.HoWo;
        const int rem0 = m0 - b0 * p.HoWo;
        const int oy0 = rem0 / p.Wo;
        const int ox0 = rem0 - oy0 * p.Wo;
        const long long base_px = (long long)(b0 * p.H + oy0 * p.stride - p.pad) * p.W + (ox0 * p.stride - p.pad);
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + base_px * p.ld_in), 0, kNumRecords, 0x00020000);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (!is_a[i]) continue;                                  // wave-uniform
            const int row = (i * NW + wave) * 16 + prow;
            const int m = m0 + row;
            uint32_t mk = 0;
            unsigned of = 0;
            if (m < p.M) {
                const int b = m / p.HoWo;
                const int rem = m - b * p.HoWo;
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad;
                const int ix0 = ox * p.stride - p.pad;
                const long long px = (long long)(b * p.H + iy0) * p.W + ix0;
                of = (unsigned)((px - base_px) * p.ld_in * 2);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        if (r < kh && s < p.kw && (unsigned)(iy0 + r) < (unsigned)p.H &&
                            (unsigned)(ix0 + s) < (unsigned)p.W)
                            mk |= 1u << (r * p.kw + s);
            }
            off[i] = of;
            mask[i] = mk;
        }
        c8 = jj; ts = 0; tapbit = 1; tapoff = (unsigned)jj * 16u;
        b_run = 0;
    };

    // K walk of the loader inside a tile (branch-free)
    auto advance_k = [&]() __attribute__((always_inline)) {
        b_run += 64u;
        c8 += 4;
        tapoff += 64u;
        const bool w = c8 >= p.C8;
        c8 = w ? c8 - p.C8 : c8;
        tapoff += w ? wrap_c : 0u;
        tapbit = w ? tapbit << 1 : tapbit;
        ts += w ? 1 : 0;
        const bool w2 = ts == p.kw;
        ts = w2 ? 0 : ts;
        tapoff += w2 ? wrap_r : 0u;
    };
    // KT is even (k_pad is a multiple of 64), so a tile's last slab has an odd index: the loader
    // crosses into its next tile, and the consumer finishes a tile, only in the odd half of a step pair
    auto advance_odd = [&]() __attribute__((always_inline)) {
        if (++l_kt == KT) {
            l_kt = 0;
            if (l_tile == last_tile) {
                l_live = false;
            } else {
                l_tile += tile_step;
                init_tile(l_tile);
            }
        } else {
            advance_k();
        }
    };
    auto advance_even = [&]() __attribute__((always_inline)) {
        ++l_kt;
        advance_k();
    };

    // piece i of the loader's current slab into ring stage `st`
    auto dma = [&](int st, int i) __attribute__((always_inline)) {
        if constexpr ((PROF & 16) != 0) return;
        const uint32_t bit = is_a[i] ? tapbit : 1u;
        const unsigned run = is_a[i] ? tapoff : b_run;
        const unsigned voff = ((mask[i] & bit) && l_live) ? off[i] + run : kOOB;
        lds_char* dst = smem + st * STAGE + (i * NW + wave) * 1024;
        if (is_a[i]) MDHIP_DMA16(a_rsrc, dst, voff, 0);
        else MDHIP_DMA16(b_rsrc, dst, voff, 0);
    };

    // ---- fragment reads: lane reads row (lane & 15), 16-byte chunk (lane >> 4) of a 16-row fragment
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ (3 * ((lane >> 3) & 1))) * 16);
    const int a_frag_base = (wm * TM) * 64 + frag_off;
    const int b_frag_base = A_BYTES + (wn * TN) * 64 + frag_off;
    auto read_x = [&](int st, int i) __attribute__((always_inline)) -> frag8_t {
        if constexpr ((PROF & 32) != 0) return frag_dummy(lane + i);
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + st * STAGE + a_frag_base + i * 1024);
    };
    auto read_w = [&](int st, int j) __attribute__((always_inline)) -> frag8_t {
        if constexpr ((PROF & 32) != 0) return frag_dummy(lane + j);
        return *(const __attribute__((address_space(3))) frag8_t*)(smem + st * STAGE + b_frag_base + j * 1024);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue: lane holds channels n..n+3 of pixel m; exactly FM*FN buffer stores ------------
    const int q4 = lane >> 4;
    auto epilogue_t = [&](int tile_m, auto has_res_t, auto out_f32_t) __attribute__((always_inline)) {
        constexpr bool HAS_RES = decltype(has_res_t)::value;
        constexpr bool OUT_F32 = decltype(out_f32_t)::value;
        const int mt = tile_m * BM;                           // descriptor based at the tile's first row
        const int ml = wm * TM + (lane & 15);
        const int nbase = n0 + wn * TN + q4 * 4;
        const int esz = OUT_F32 ? 4 : 2;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((char*)p.out + (size_t)mt * p.ld_out * esz), 0, kNumRecords, 0x00020000);
        uint2 rbuf[2][FM];
        auto fetch_res = [&](int j, uint2 (&r)[FM]) {
            const int n = min(nbase + j * 16, p.N - 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = min(mt + ml + i * 16, p.M - 1);
                r[i] = *(const uint2*)(p.res + (size_t)m * p.ld_res + n);
            }
        };
        if constexpr (HAS_RES) fetch_res(0, rbuf[0]);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            if constexpr (HAS_RES) {
                if (j + 1 < FN) fetch_res(j + 1, rbuf[(j + 1) & 1]);
            }
            const int nb = n0 + wn * TN + j * 16;                // wave-uniform
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (nb < p.n_rows) {
                // 16 biases of this fragment column through the scalar cache (explicit s_load: a
                // compiler-chosen vector load would put a vmcnt(0) in front of the stores)
                f32x16 b16;
                const unsigned long long ba = (unsigned long long)(p.bias + nb);
                const unsigned long long bs =
                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ba >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)ba);
                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b16) : "s"(bs) : "memory");
                const f32x4 g0 = {b16[0], b16[1], b16[2], b16[3]}, g1 = {b16[4], b16[5], b16[6], b16[7]},
                            g2 = {b16[8], b16[9], b16[10], b16[11]}, g3 = {b16[12], b16[13], b16[14], b16[15]};
                const f32x4 g = q4 == 0 ? g0 : (q4 == 1 ? g1 : (q4 == 2 ? g2 : g3));
                bv[0] = g[0]; bv[1] = g[1]; bv[2] = g[2]; bv[3] = g[3];
            }
            const int n = nbase + j * 16;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = ml + i * 16;                        // row inside the tile
                float v0 = acc[i][j][0] + bv[0];
                float v1 = acc[i][j][1] + bv[1];
                float v2 = acc[i][j][2] + bv[2];
                float v3 = acc[i][j][3] + bv[3];
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((PROF & 4) == 0 && p.act) {
                    v0 = silu_f32(v0); v1 = silu_f32(v1); v2 = silu_f32(v2); v3 = silu_f32(v3);
                }
                if constexpr (HAS_RES) {
                    const uint2 rv = rbuf[j & 1][i];
                    v0 += st_unpack((uint16_t)(rv.x & 0xffff));
                    v1 += st_unpack((uint16_t)(rv.x >> 16));
                    v2 += st_unpack((uint16_t)(rv.y & 0xffff));
                    v3 += st_unpack((uint16_t)(rv.y >> 16));
                }
                const bool ok = (mt + m < p.M) && (n < p.N);
                const unsigned voff = ok ? (unsigned)(m * p.ld_out + n) * (unsigned)esz : kOOB;
                if constexpr ((PROF & 2) != 0) {
                    asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(voff));
                } else if constexpr (OUT_F32) {
                    const u32x4 o = {__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
                    __builtin_amdgcn_raw_buffer_store_b128(o, o_rsrc, voff, 0, 0);
                } else {
                    const u32x2 o = {st_pack2(v0, v1), st_pack2(v2, v3)};
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

    // ---- prologue: slabs 0..3 in flight, fragments of slab 0 in registers -------------------------
    init_tile(first_tile);
#pragma unroll
    for (int st = 0; st < kStages; ++st) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma(st, i);
        if (st & 1) advance_odd(); else advance_even();
    }
    wait_vm_lgkm0<3 * PPW>();
    __builtin_amdgcn_s_barrier();

    frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) xa[i] = read_x(0, i);
#pragma unroll
    for (int j = 0; j < FN; ++j) wa[j] = read_w(0, j);

    int c_kt = 0, c_tile = first_tile;
    int since_epi = 3;            // steps since the last epilogue (its stores are younger than in-flight slabs for 3 steps)
    const bool count_stores = KT >= 4 && (PROF & 2) == 0;

    // first part of a step: this wave's pieces of slab step+1 have landed (slabs step+2, step+3 stay
    // in flight) and its fragment reads of slab `step` are complete, so after the barrier stage
    // step & 3 is free for slab step+4; then slab step+1's fragments go to the other register set
    // while slab `step` multiplies
#define MDHIP_V3_STEP(ST_FREE, ST_NEXT, XC, WC, XN, WN_)                                              \
    {                                                                                              \
        if (since_epi < 3 && count_stores) wait_vm_lgkm0<2 * PPW + EPI_OPS>();                     \
        else wait_vm_lgkm0<2 * PPW>();                                                             \
        __builtin_amdgcn_s_barrier();                                                              \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i) dma(ST_FREE, i);                           \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) XN[i] = read_x(ST_NEXT, i);                 \
        _Pragma("unroll") for (int j = 0; j < FN; ++j) WN_[j] = read_w(ST_NEXT, j);                \
        if constexpr ((PROF & 64) == 0) {                                                          \
            _Pragma("unroll") for (int j = 0; j < FN; ++j)                                         \
                _Pragma("unroll") for (int i = 0; i < FM; ++i)                                     \
                    acc[i][j] = MDHIP_MFMA(WC[j], XC[i], acc[i][j]); \
        }                                                                                          \
    }

    // total_steps is even; the ring position of slab s is s & 3
    for (int step = 0; step < total_steps; step += 2) {
        const int r = step & 2;                       // 0 or 2
        MDHIP_V3_STEP(r, r + 1, xa, wa, xb, wb)
        advance_even();
        ++since_epi;
        MDHIP_V3_STEP(r + 1, (r + 2) & 3, xb, wb, xa, wa)
        advance_odd();
        ++since_epi;
        c_kt += 2;
        if (c_kt == KT) {
            epilogue(c_tile);
            c_kt = 0;
            c_tile += tile_step;
            since_epi = 0;
        }
    }
#undef MDHIP_V3_STEP
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table
// ---------------------------------------------------------------------------------------
// id (local), BM, BN, waves along M, waves along N, PROF bits
#define MDHIP_CONV3_CFGS(X)  \
    X(0, 160, 160, 2, 2, 0)  \
    X(1, 96, 160, 2, 2, 0)
#define MDHIP_CONV3_PROF(X)  \
    X(2, 160, 160, 2, 2, 54) \
    X(3, 160, 160, 2, 2, 22) \
    X(4, 160, 160, 2, 2, 38) \
    X(5, 160, 160, 2, 2, 6)  \
    X(6, 160, 160, 2, 2, 102)

static const ConvCfg g_cfgs3[] = {
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    {bm, bn, (wm) * (wn) * 64, (size_t)v3_lds_bytes(bm, bn), v3_blocks_per_cu(bm, bn, (wm) * (wn)), \
     "v3:" #bm "x" #bn "/" #wm "x" #wn "/" #prof},
    MDHIP_CONV3_CFGS(X) MDHIP_CONV3_PROF(X)
#undef X
};
constexpr int kNumProf3 = 5;    // trailing developer variants: reachable through conv3_launch only

int conv3_num_cfgs() { return (int)(sizeof(g_cfgs3) / sizeof(g_cfgs3[0])) - kNumProf3; }
const ConvCfg& conv3_cfg(int i) { return g_cfgs3[i]; }

hipError_t conv3_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    if (e == hipSuccess)                                                                           \
        e = hipFuncSetAttribute((const void*)conv_v3_kernel<bm, bn, wm, wn, prof>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs3[id].lds_bytes);
    MDHIP_CONV3_CFGS(X) MDHIP_CONV3_PROF(X)
#undef X
    return e;
}

bool conv3_supports(const ConvArgs& a) {
    // branch-free K walk: at least one whole 32-deep slab per tap; per-tile output descriptor: 32-bit offsets
    return a.C8 >= 4 && a.kw <= 3 && a.ntaps <= 9 && (a.k_pad % 64) == 0 && (a.N % 4) == 0 &&
           (long long)512 * a.ld_out * 4 < 0x7fffffffLL;
}

hipError_t conv3_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (cfg < 0 || cfg >= conv3_num_cfgs() + kNumProf3 || !conv3_supports(a)) return hipErrorInvalidValue;
    const ConvCfg& c = g_cfgs3[cfg];
    ConvArgs p = a;
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, (32 * c.blocks_per_cu) / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    switch (cfg) {
#define X(id, bm, bn, wm, wn, prof)                                                                  \
    case id:                                                                                       \
        hipLaunchKernelGGL((conv_v3_kernel<bm, bn, wm, wn, prof>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV3_CFGS(X) MDHIP_CONV3_PROF(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
