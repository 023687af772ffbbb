// conv_v5 for small launches (gfx950 / MI355X): the same 3x3 / stride 1 row-segment convolution, the same K order and
// the same MFMA chains per accumulator -- bit-identical results -- scheduled for the case conv_v5.cpp is worst at: fewer
// tiles than the chip has room for (batch 1..4 on the 20x20 .. 80x80 maps: 28..400 tiles of 64 pixels on 256 CUs).
//
// What bounds conv_v5 there (tools/convbench, CONVBENCH_B=1, 640 -> 640 channels on 20x20): one workgroup per CU, every
// step = { wait for the DMA, barrier, LDS round trip of the next fragments, 4..20 MFMAs } with nothing else on the
// SIMD to cover the three latencies: 0.53 us per 64-deep step whatever the tile (61 TFLOP/s on the whole chip), and
// deeper weight stages alone change nothing (the LDS round trip per half step remains).  Here:
//
//   * staging is per RUN (one kernel row of one 64-channel group = 3 taps = 6 half steps of 32 k): NS stages of
//     { run of BM + 2 pixels, its three weight slabs }, the loader NS runs ahead, ONE wait + barrier per run;
//   * fragments are read PD half steps ahead of the MFMAs that use them, into a register ring of 6 slots (one per
//     half step of a run): the LDS round trip is covered by PD half steps of MFMAs of the same wave, and the barrier
//     sits PD half steps before the end of a run, where the last fragment reads of that run have been issued;
//   * small tiles (64x32 .. 64x64 per workgroup, 32x32 per wave) so that a 20x20 map still gives 70..140 workgroups.
//
// Measured (same shape): 48 -> 26 us (640 ch, 20x20, batch 1 and 2), 39 -> 21.5 us (480 ch, 40x40, batch 1), 42 -> 27 us
// (batch 2); of what remains ~8 us is the floor of any launch here (dispatch, prologue latency, epilogue) and ~9 us the
// loop with its DMA, MFMAs and barriers all removed.
//
// Selected by the tile table at small batch sizes only (same kernel family as conv_v5: an image's result does not depend
// on the batch size).

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

constexpr int v5s_run_pieces(int bm) { return (bm + 2 + 7) / 8; }
constexpr int v5s_stage_bytes(int bm, int bn, int nw) {
    return (v5s_run_pieces(bm) + nw - 1) / nw * nw * 1024 + 3 * bn * 128;
}
// NS stages + 1 KiB (row of zeros, bias)
constexpr int v5s_lds_bytes(int bm, int bn, int nw, int ns) { return ns * v5s_stage_bytes(bm, bn, nw) + 1024; }
constexpr int v5s_blocks_per_cu(int bm, int bn, int nw, int ns) {
    int b = 163840 / v5s_lds_bytes(bm, bn, nw, ns);
    if (b > 8 / nw) b = 8 / nw;          // two waves per SIMD (256 registers each)
    return b < 1 ? 1 : b;
}
constexpr int v5s_waves_per_simd(int bm, int bn, int nw, int ns) {
    int w = v5s_blocks_per_cu(bm, bn, nw, ns) * nw / 4;
    return w < 1 ? 1 : w;
}

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

template <int BM, int BN, int WM, int WN, int NS, int PD>
__global__ void __launch_bounds__(WM * WN * 64, v5s_waves_per_simd(BM, BN, WM * WN, NS))
conv_v5s_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_PIECES = v5s_run_pieces(BM);
    constexpr int A_PER = (A_PIECES + NW - 1) / NW;          // run pieces per wave (every wave issues all of them)
    constexpr int A_PAD = A_PER * NW * 1024;                 // run buffer incl. the pieces that do not exist
    constexpr int B_BYTES = BN * 128, B_PIECES = BN / 8, B_PER = B_PIECES / NW;
    constexpr int STAGE = A_PAD + 3 * B_BYTES;
    constexpr int P_RUN = A_PER + 3 * B_PER;                 // DMA requests per wave and run
    constexpr int ZERO_OFF = NS * STAGE;
    constexpr int WAIT_LOOP = (NS - 2) * P_RUN < 63 ? (NS - 2) * P_RUN : 63;
    constexpr int WAIT_PRO = (NS - 1) * P_RUN < 63 ? (NS - 1) * P_RUN : 63;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "16x16 fragments");
    static_assert(B_PIECES % NW == 0, "counted vmcnt: the same number of requests in every wave");
    static_assert(NS >= 2 && PD >= 1 && PD <= 6, "stages / prefetch distance");
    static_assert(BN * 4 + 256 <= 1024, "bias staging area");

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
    const int G = p.groups;                       // 64-channel groups (the last one may be half full)
    const int runs_per_tile = 3 * G;              // (channel group, kernel row) pairs
    const int total_runs = my_tiles * runs_per_tile;

    // the row of zeros that invalid (pixel, tap) pairs read, and the bias of this workgroup's BN channels behind it
    if (tid < 16) *(__attribute__((address_space(3))) uint4*)(smem + ZERO_OFF + tid * 16) = make_uint4(0, 0, 0, 0);
    for (int c = tid; c < BN; c += NW * 64)
        *(__attribute__((address_space(3))) float*)(smem + ZERO_OFF + 256 + c * 4) = (n0 + c < p.n_rows) ? p.bias[n0 + c] : 0.f;

    // ---- loader: run (tile, cg, r) = BM + 2 pixels of channels cg*64 .. +63 starting one pixel before the tile's
    //      first pixel in image row r - 1 (raster index over the whole batch), and the slabs (cg, r, 0..2) ----------
    const int lr = lane >> 3;
    const int jj = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt4 + (size_t)n0 * p.k_pad4), 0, kNumRecords, 0x00020000);
    unsigned b_off[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int row = (i * NW + wave) * 8 + lr;
        b_off[i] = (n0 + row < p.n_rows) ? (unsigned)(row * p.k_pad4 + jj * 8) * 2u : kOOB;
    }
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    unsigned q_off[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int q = (i * NW + wave) * 8 + lr;
        q_off[i] = (unsigned)(q * p.ld_in * 2 + jj * 16);
    }
    int lg_tile = first_tile, lg_cg = 0, lg_r = 0, lg_run = 0;
    bool lg_live = true;
    int lg_first = 0;                              // raster index of the run's first pixel (may be negative)
    unsigned lg_soff = 0;
    auto run_tile = [&](int t) __attribute__((always_inline)) {
        const long long origin = (long long)t * BM - p.W - 1;          // first pixel of the r = 0 run
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + origin * p.ld_in), 0, kNumRecords, 0x00020000);
    };
    auto run_setup = [&]() __attribute__((always_inline)) {
        lg_first = lg_tile * BM + (lg_r - 1) * p.W - 1;
        lg_soff = (unsigned)(lg_r * p.W * p.ld_in * 2 + lg_cg * 128);
    };
    // all requests of the loader's run into stage `stage`; every wave issues every one of them (a run piece that does
    // not exist reads out of bounds into the padding)
    auto dma_run = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < A_PER; ++d) {
            const int q = (d * NW + wave) * 8 + lr;
            const bool ok = d * NW + wave < A_PIECES && lg_live && (unsigned)(lg_first + q) < (unsigned)p.M &&
                            lg_cg * 8 + jj < p.C8;
            MDHIP_DMA16(a_rsrc, smem + stage * STAGE + (d * NW + wave) * 1024, ok ? q_off[d] : kOOB, lg_soff);
        }
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
                MDHIP_DMA16(b_rsrc, smem + stage * STAGE + A_PAD + sl * B_BYTES + (i * NW + wave) * 1024, b_off[i],
                            (lg_run * 3 + sl) * 128);
    };
    auto run_next = [&]() __attribute__((always_inline)) {
        lg_run = (lg_run + 1 == runs_per_tile) ? 0 : lg_run + 1;
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

    // ---- fragment reads (layout as conv_v5: 128-byte rows, 16-byte chunk c of row q at position c ^ (q & 7)) ------
    const int c0 = lane >> 4;
    unsigned a_sh[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
        a_sh[s] = (unsigned)((wm * TM + (lane & 15) + s) * 128 + ((c0 ^ (((lane & 7) + s) & 7)) << 4));
    const unsigned z_addr = (unsigned)(ZERO_OFF + c0 * 16);
    const int b_frag_base = (wn * TN + (lane & 15)) * 128 + ((c0 ^ (lane & 7)) << 4);
    uint32_t vmask[FM];                            // tap-validity bits of this lane's FM pixels (tile being READ)
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
    frag8_t fx[6][FM], fw[6][FN];                  // ring slot = half step (s, kk) of a run
    // the fragments of half step hs (0..5) of the run in `stage` (kernel row r) into ring slot hs
    auto read_half = [&](int hs, int stage, int r) __attribute__((always_inline)) {
        const int s = hs >> 1, kk = hs & 1;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const unsigned a = a_sh[s] + (unsigned)(stage * STAGE + i * 2048);
            const unsigned e = ((vmask[i] >> (r * 3 + s)) & 1u) ? a : z_addr;
            fx[hs][i] = *(const __attribute__((address_space(3))) frag8_t*)(const lds_char*)(e ^ (unsigned)(kk * 64));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j)
            fw[hs][j] = *(const __attribute__((address_space(3))) frag8_t*)(const lds_char*)((unsigned)(stage * STAGE + A_PAD + s * B_BYTES + j * 2048) +
                                                                                           (unsigned)(b_frag_base ^ (kk * 64)));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue (as conv_v5: bias from LDS, SiLU, residual in store layout, 16-byte stores) ----------------------
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
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + i * 16;
            uint4 rpair[NPAIR > 0 ? NPAIR : 1];
            uint2 rlast = make_uint2(0, 0);
            if constexpr (HAS_RES) {
                const uint16_t* rrow_p = p.res + (size_t)min(m, p.M - 1) * p.ld_res;
#pragma unroll
                for (int jp = 0; jp < NPAIR; ++jp)
                    rpair[jp] = *(const uint4*)(rrow_p + min(n0 + wn * TN + jp * 32 + q4 * 8, p.N - 8));
                if (FN & 1) rlast = *(const uint2*)(rrow_p + min(nbase + (FN - 1) * 16, p.N - 4));
            }
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                mdhip_bias4(acc[i][j], bv[j], v[j]);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (p.act) {
#pragma unroll
                for (int j = 0; j < FN; ++j) mdhip_silu4(v[j]);
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
                    const uint4 d = rpair[jp];                    // as stored: (t0[0], t1[0], t0[1], t1[1])
                    auto s0 = __builtin_amdgcn_permlane16_swap(d.x, d.z, false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(d.y, d.w, false, false);
                    auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s0[1], false, false);
                    auto a1 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
                    add4(2 * jp, a0[0], a1[0]);
                    add4(2 * jp + 1, a0[1], a1[1]);
                }
                if (FN & 1) add4(FN - 1, rlast.x, rlast.y);
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

    // ---- prologue: runs 0 .. NS-1 requested, run 0 landed, its first PD half steps in the ring ----------------------
    run_tile(first_tile);
    run_setup();
#pragma unroll
    for (int r = 0; r < NS; ++r) {
        dma_run(r);
        run_next();
    }
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAIT_PRO) : "memory");
    __builtin_amdgcn_s_barrier();
    tile_masks(first_tile);
#pragma unroll
    for (int t = 0; t < PD && t < 6; ++t) read_half(t, 0, 0);

    int c_r = 0, c_cg = 0, c_tile = first_tile, pa = 0;
    // a half-full last channel group (C_in mod 64 <= 32) has nothing in k 32..63: its second-half MFMAs are skipped
    const bool tail_short = (p.C8 & 7) != 0 && (p.C8 & 7) <= 4;
    for (int run = 0; run < total_runs; ++run) {
        const bool skip_y = tail_short && c_cg == G - 1;
        const bool tile_end = c_r == 2 && c_cg == G - 1;
        const int n_r = c_r == 2 ? 0 : c_r + 1;
        const int pa_next = (pa + 1 == NS) ? 0 : pa + 1;
#pragma unroll
        for (int h = 0; h < 6; ++h) {
            if (h == 6 - PD) {
                // every fragment read of this run has been issued: once they are back the run's stage is free for the
                // loader (NS runs ahead); the run that starts next (requested NS - 1 runs ago) must have landed,
                // NS - 2 runs of requests behind it may stay in flight
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAIT_LOOP) : "memory");
                __builtin_amdgcn_s_barrier();
                dma_run(pa);
                run_next();
                if (tile_end) tile_masks(c_tile + tile_step);      // the reads from here on belong to the next tile
            }
            const int t = h + PD;
            if (t < 6) read_half(t, pa, c_r);
            else read_half(t - 6, pa_next, n_r);
            __builtin_amdgcn_sched_barrier(0);
            if (!((h & 1) && skip_y)) {
#pragma unroll
                for (int g = 0; g < FN; ++g)
#pragma unroll
                    for (int i = 0; i < FM; ++i) acc[i][g] = MDHIP_MFMA(fw[h][g], fx[h][i], acc[i][g]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pa = pa_next;
        c_r = n_r;
        if (n_r == 0 && ++c_cg == G) {
            c_cg = 0;
            epilogue(c_tile);
            c_tile += tile_step;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table (ids local to this file; conv_v5.cpp appends them to its own)
// ---------------------------------------------------------------------------------------
// id, BM, BN, waves along M, waves along N, run stages, fragment prefetch distance (half steps)
#define MDHIP_CONV5S_CFGS(X)   \
    X(0, 64, 64, 2, 2, 2, 5)   \
    X(1, 128, 64, 2, 2, 2, 5)

static const ConvCfg g_cfgs5s[] = {
#define X(id, bm, bn, wm, wn, ns, pd)                                                          \
    {bm, bn, (wm) * (wn) * 64, (size_t)v5s_lds_bytes(bm, bn, (wm) * (wn), ns),                  \
     v5s_blocks_per_cu(bm, bn, (wm) * (wn), ns), "v5:run" #bm "x" #bn "/" #wm "x" #wn "/r" #ns "p" #pd},
    MDHIP_CONV5S_CFGS(X)
#undef X
};

int conv5s_num_cfgs() { return (int)(sizeof(g_cfgs5s) / sizeof(g_cfgs5s[0])); }
const ConvCfg& conv5s_cfg(int i) { return g_cfgs5s[i]; }

hipError_t conv5s_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, bn, wm, wn, ns, pd)                                                          \
    if (e == hipSuccess)                                                                       \
        e = hipFuncSetAttribute((const void*)conv_v5s_kernel<bm, bn, wm, wn, ns, pd>,             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_cfgs5s[id].lds_bytes);
    MDHIP_CONV5S_CFGS(X)
#undef X
    return e;
}

// the caller (conv5_supports) has checked the shape conditions common to the family
bool conv5s_supports(int cfg, const ConvArgs& a) {
    return cfg >= 0 && cfg < conv5s_num_cfgs() && !a.out_f32;
}

hipError_t conv5s_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    const ConvCfg& c = g_cfgs5s[cfg];
    ConvArgs p = a;
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, (32 * c.blocks_per_cu) / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    switch (cfg) {
#define X(id, bm, bn, wm, wn, ns, pd)                                                              \
    case id:                                                                                    \
        hipLaunchKernelGGL((conv_v5s_kernel<bm, bn, wm, wn, ns, pd>), grid, dim3((wm) * (wn) * 64), c.lds_bytes, s, p); \
        break;
        MDHIP_CONV5S_CFGS(X)
#undef X
    }
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
