// Implicit-GEMM convolution for gfx950 (MI355X): bf16 operands, fp32 MFMA accumulation,
// bias + SiLU (+ residual) fused into the epilogue, output written straight into a channel
// slice of the consumer's buffer (concat-by-construction).
//
// Replaces the cuDNN/MIOpen conv2d + SiLU (+ add, + cat) kernels PyTorch launches under
//   reference megadetector/detection/pytorch_detector.py:1313   self.model(batch_tensor)
// for every Conv module of the YOLOv5x6 graph (SURVEY.md section 8(a) P4 layer table).
//
// GEMM view:  C[M][N] = A[M][K] * W[N][K]^T
//   M = batch*Ho*Wo output pixels (NHWC, pixel-major), N = C_out, K = kh*kw*C_in ordered
//   (r, s, c) so that every 16-byte chunk (8 channels) of an A row is contiguous in HBM.
//
// Data movement (per workgroup, per 64-wide K slab):
//   HBM/L2 --buffer_load_dwordx4 ... lds, 16 B/lane--> LDS  (no VGPR round trip; the per-lane
//   *source* offset does the im2col gather; conv padding, K padding and ragged tiles are lanes
//   whose offset is out of the descriptor's range, which the hardware returns as zeros; the
//   K-slab advance of the weight tile is a wave-uniform SGPR offset: no per-lane arithmetic)
//   LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row & 7) on the source
//   side and on the ds_read_b128 side (guide rule 21: linear LDS destination, same involution
//   on source and read) so that fragment reads are bank-conflict free.
//   NS LDS stages form a ring: the loads of slabs k+1 .. k+NS-1 are in flight while slab k feeds
//   the MFMAs.  A wave waits for its own slab-k loads with a *counted* s_waitcnt vmcnt(N)
//   (younger slabs stay in flight), then one raw s_barrier per slab makes every wave's part
//   of slab k visible and releases the stage that was consumed in the previous iteration
//   (guide: glds across barriers needs raw s_barrier + counted vmcnt, never __syncthreads()).
//
// MFMA: v_mfma_f32_16x16x32_bf16 with the operands swapped (weights as "A", pixels as "B") so
//   that each lane ends up with 4 *consecutive output channels* of one pixel: the epilogue is
//   lane-local (bias, SiLU, residual, bf16 pack) and stores 8 contiguous bytes per fragment.
//
// Workgroup -> tile mapping is XCD-aware: consecutive tiles (which share im2col halo rows
//   and the weight panel) are placed on the same XCD/L2.

#include <algorithm>

#include "mdhip_internal.h"

namespace mdhip {
namespace MDHIP_ST {

typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((address_space(3))) char lds_char;

// 16 bytes per lane, HBM/L2 -> LDS directly: LDS address = m0-base (wave-uniform) + lane*16;
// source = descriptor base + voff (per lane) + soff (wave-uniform SGPR).  A lane whose voff is
// >= num_records reads zeros: that is how conv padding, K padding and ragged tiles are done.
#define MDHIP_BLDS16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

[[maybe_unused]] constexpr unsigned kOOB = 0x80000000u;        // >= every descriptor's num_records
[[maybe_unused]] constexpr int kNumRecords = 0x7fffffff;


// SiLU = x * sigmoid(x): v_mul, v_exp, v_add, v_rcp, v_mul (each <= 1 ulp: far inside the bf16
// rounding that follows)
__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// two fp32 -> packed bf16, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS bytes of one workgroup and the residency (workgroups per CU, waves per SIMD) it is built for
constexpr int conv_lds_bytes(int bm, int bn, int ns) { return ns * (bm + bn) * 128 + 1024 + bn * 4; }
constexpr int conv_blocks_per_cu(int bm, int bn, int nw, int ns) {
    int b = 163840 / conv_lds_bytes(bm, bn, ns);
    if (b > 32 / nw) b = 32 / nw;
    if (b > 4) b = 4;
    return b < 1 ? 1 : b;
}
constexpr int conv_waves_per_simd(int bm, int bn, int nw, int ns) {
    int w = conv_blocks_per_cu(bm, bn, nw, ns) * nw / 4;
    return w < 1 ? 1 : w;
}

// FP = 1: fragment prefetch -- the LDS reads of the next 32-deep half slab are issued before the
// MFMAs of the current one (two register sets), so ds_read latency hides under the matrix pipe
// instead of in front of it; needs NS >= 3 because the next slab must already have landed.
// DEC = 1: the instantiation for the Detect 1x1 convs that decode in their epilogue (ConvArgs::dec_pred, mdhip_decode_store).  Its own
// instantiation, of the narrow tiles only: as a run-time branch the decode path (expf, a division, the grid arithmetic) cost every
// instantiation 10 - 24 registers and put scratch memory into the widest tiles of conv_v2.cpp (measured in the ISA, round 6).
template <int BM, int BN, int WM, int WN, int NS, int FP, int DEC = 0>
__global__ void __launch_bounds__(WM * WN * 64, conv_waves_per_simd(BM, BN, WM * WN, NS))
conv_igemm_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the buffer-resource
                                      // type and builtins below do not exist for the x86 target)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8;                 // 8 rows x 128 B per wave-instruction
    constexpr int B_INSTR = BN / 8;
    constexpr int A_PER = A_INSTR / NW;
    constexpr int B_PER = (B_INSTR + NW - 1) / NW;
    constexpr int LPS = A_PER + B_PER;              // loads per slab per wave (uniform)
    static_assert(A_INSTR % NW == 0, "A tile must split evenly over the waves");
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be a multiple of 16x16");
    static_assert(NS >= 2 && (NS - 2) * LPS < 64, "vmcnt is a 6-bit counter");
    static_assert(!FP || NS >= 3, "fragment prefetch needs three LDS stages");
    // every wave issues exactly LPS loads per slab (uniform vmcnt bookkeeping); when BN/8 does
    // not divide by the wave count the surplus loads are out-of-range (zeros) and land in one
    // shared 1 KiB dump area behind the last stage
    constexpr int DUMP_OFF = NS * STAGE;
    constexpr int BIAS_OFF = DUMP_OFF + 1024;       // BN fp32 biases of this stream's N tile

    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- persistent streams ------------------------------------------------------------------
    // Block b runs on XCD b % 8.  Each XCD owns one contiguous range of M tiles; inside the XCD the
    // streams interleave over that range (stream s takes tiles s, s+S, s+2S, ...), so at any moment
    // the workgroups behind one L2 work on neighbouring tiles: the 3x3 halo rows and the A rows
    // shared by the N tiles are L2 hits instead of fabric traffic.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile_n = slot % p.tiles_n;
    const int ms = slot / p.tiles_n;                       // M stream index inside the XCD
    const int xcd_first = xcd * p.tiles_per_xcd;
    const int xcd_tiles = min(p.tiles_per_xcd, p.tiles_m - xcd_first);
    const int my_tiles = (xcd_tiles > ms) ? (xcd_tiles - ms + p.m_streams - 1) / p.m_streams : 0;
    if (my_tiles <= 0) return;
    const int first_tile = xcd_first + ms;
    const int tile_step = p.m_streams;
    const int n0 = tile_n * BN;
    const int KT = p.k_pad >> 6;
    const int total_steps = my_tiles * KT;

    // ---- weight side: fixed for the whole stream -------------------------------------------
    const int lr = lane >> 3;              // row inside an 8-row load instruction (== row & 7)
    const int jj = (lane & 7) ^ lr;        // swizzled source chunk inside the 128-byte K slab
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wgt + (size_t)n0 * p.k_pad), 0, kNumRecords, 0x00020000);
    unsigned b_off[B_PER];                 // byte offset of (row, chunk jj); the slab adds kt*128 as SGPR offset
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int row = (i * NW + wave) * 8 + lr;
        b_off[i] = (row < BN && n0 + row < p.n_rows) ? (unsigned)(row * p.k_pad + jj * 8) * 2u : kOOB;
    }
    for (int i = tid; i < BN; i += NW * 64)
        *(__attribute__((address_space(3))) float*)(smem + BIAS_OFF + i * 4) =
            (n0 + i < p.n_rows) ? p.bias[n0 + i] : 0.f;
    __syncthreads();

    // ---- activation side: loader state of the tile whose slabs are being issued ---------------
    __amdgpu_buffer_rsrc_t a_rsrc = b_rsrc;
    unsigned a_off[A_PER];                 // byte offset of the row's (tap 0, channel 0) from the A base
    uint32_t a_mask[A_PER];                // bit t set: tap t of this row is inside the image
    int c8 = 0, tap = 0, tr = 0, ts = 0;   // this lane's position inside K
    int l_kt = KT, l_tile = first_tile - tile_step;
    const int kh = p.ntaps / p.kw;

    // 1x1 / stride 1 / no padding: output pixel m is input pixel m -- no divisions, one mask bit (see conv_v2.cpp)
    const bool pointwise = p.ntaps == 1 && p.stride == 1 && p.pad == 0 && p.H == p.Ho && p.W == p.Wo && p.C8 >= 8;
    auto init_loader_tile = [&](int tile_m) {
        const int m0 = tile_m * BM;
        if (pointwise) {                                                            // wave-uniform
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (long long)m0 * p.ld_in), 0, kNumRecords, 0x00020000);
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int row = (i * NW + wave) * 8 + lr;
                a_off[i] = (unsigned)(row * p.ld_in * 2);
                a_mask[i] = (m0 + row < p.M) ? 1u : 0u;
            }
            c8 = jj; tap = 0; tr = 0; ts = 0;            // (C8 >= 8: the lane's first chunk is inside tap 0)
            return;
        }
        // descriptor based at the tile's first pixel minus the conv padding: every in-range tap
        // of every row of the tile has a small non-negative byte offset
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
                uint32_t cols = 0, rows = 0;         // kernels are 1x1 or 3x3 (planner enforces it): bit r * kw + s
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
        c8 = jj; tap = 0; tr = 0; ts = 0;
        while (c8 >= p.C8) {
            c8 -= p.C8;
            ++tap;
            if (++ts == p.kw) { ts = 0; ++tr; }
        }
    };

    // issue the loads of the next slab (crossing into the stream's next tile when needed)
    auto issue = [&](int buf) {
        if (l_kt == KT) {
            l_tile += tile_step;
            init_loader_tile(l_tile);
            l_kt = 0;
        }
        lds_char* sA = smem + buf * STAGE;
        lds_char* sB = sA + A_BYTES;
        const unsigned tapoff = (unsigned)((tr * p.W + ts) * p.ld_in + c8 * 8) * 2u;
        const uint32_t bit = (tap < p.ntaps) ? (1u << (tap & 31)) : 0u;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const unsigned voff = (a_mask[i] & bit) ? a_off[i] + tapoff : kOOB;
            MDHIP_BLDS16(a_rsrc, sA + (i * NW + wave) * 1024, voff, 0);
        }
        const int soff = l_kt * 128;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int instr = i * NW + wave;
            lds_char* dst = (instr < B_INSTR) ? sB + instr * 1024 : smem + DUMP_OFF;
            MDHIP_BLDS16(b_rsrc, dst, b_off[i], soff);
        }
        c8 += 8;                           // advance this lane's K position by one slab
        while (c8 >= p.C8) {
            c8 -= p.C8;
            ++tap;
            if (++ts == p.kw) { ts = 0; ++tr; }
        }
        ++l_kt;
    };

    // ---- fragment read offsets ---------------------------------------------------------
    // lane reads row (lane & 15) of a 16-row fragment, 16-byte chunk (kk*4 + (lane >> 4)),
    // stored at chunk ^ (row & 7)
    const int frag_row_off = (lane & 15) * 128;
    const int frag_ch0 = (((lane >> 4) ^ (lane & 7)) * 16);      // kk = 0 ; kk = 1 is ^ 64
    const int a_frag_base = (wm * TM) * 128 + frag_row_off;
    const int b_frag_base = A_BYTES + (wn * TN) * 128 + frag_row_off;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue: lane holds channels n..n+3 of pixel m --------------------------------
    // Pixel-row by pixel-row, so that the stores completing one cache line are issued back to back;
    // bf16 outputs of two fragment columns are exchanged across the four 16-lane rows
    // (v_permlane32_swap + v_permlane16_swap) so that a lane stores 8 consecutive channels = 16 bytes
    // and one store instruction covers 64 contiguous bytes per pixel instead of 32 (measured on the
    // 160x160-tile kernel: epilogue 36k -> 14k cycles per tile under load, 1x1 convs +25 %).
    auto epilogue = [&](int tile_m) {
        const int mrow = tile_m * BM + wm * TM + (lane & 15);
        const int q4 = lane >> 4;
        const int nl0 = wn * TN + q4 * 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow + i * 16;
            const bool m_ok = m < p.M;
            float v[FN][4];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nl = nl0 + j * 16;
                const float4 bv = *(const __attribute__((address_space(3))) float4*)(smem + BIAS_OFF + nl * 4);
                const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
                mdhip_bias4(acc[i][j], b4, v[j]);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (p.act) {
#pragma unroll
                for (int j = 0; j < FN; ++j) mdhip_silu4(v[j]);
            }
            if (p.res) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = n0 + nl0 + j * 16;
                    if (m_ok && n < p.N) {
                        const uint2 rv = *(const uint2*)(p.res + (size_t)m * p.ld_res + n);
                        v[j][0] += st_unpack((uint16_t)(rv.x & 0xffff));
                        v[j][1] += st_unpack((uint16_t)(rv.x >> 16));
                        v[j][2] += st_unpack((uint16_t)(rv.y & 0xffff));
                        v[j][3] += st_unpack((uint16_t)(rv.y >> 16));
                    }
                }
            }
            if constexpr (DEC != 0) {                               // Detect decode in place (ConvArgs::dec_pred)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = n0 + nl0 + j * 16;
                    if (m_ok && n < p.N) mdhip_decode_store(p, m, n, v[j]);
                }
            } else if (p.out_f32) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = n0 + nl0 + j * 16;
                    if (m_ok && n < p.N)
                        *(float4*)((float*)p.out + (size_t)m * p.ld_out + n) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                }
            } else if (p.out_f8) {
                // e4m3 output (MDHIP_DTYPE_FP8: the hidden tensor of a bottleneck): 4 channels = 4 bytes per lane and
                // fragment; the same exchange as the 16-bit path leaves 8 consecutive channels = 8 bytes per lane
                uint8_t* orow8 = (uint8_t*)p.out + (size_t)m * p.ld_out;
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    const unsigned a0 = pack_e4m3x4(v[j][0], v[j][1], v[j][2], v[j][3], p.out_qscale);
                    const unsigned b0 = pack_e4m3x4(v[j + 1][0], v[j + 1][1], v[j + 1][2], v[j + 1][3], p.out_qscale);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    const int n = n0 + wn * TN + j * 16 + q4 * 8;
                    if (m_ok && n < p.N) *(uint2*)(orow8 + n) = make_uint2(t0[0], t0[1]);
                }
                if (FN & 1) {
                    const int j = FN - 1;
                    const int n = n0 + nl0 + j * 16;
                    if (m_ok && n < p.N) *(unsigned*)(orow8 + n) = pack_e4m3x4(v[j][0], v[j][1], v[j][2], v[j][3], p.out_qscale);
                }
            } else {
                uint16_t* orow = (uint16_t*)p.out + (size_t)m * p.ld_out;
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    const unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                    const unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                    const auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                    // row q of the wave now holds channels q*8 .. q*8+7 of this pair's 32 channels
                    const int n = n0 + wn * TN + j * 16 + q4 * 8;
                    if (m_ok && n < p.N) *(uint4*)(orow + n) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                }
                if (FN & 1) {
                    const int j = FN - 1;
                    const int n = n0 + nl0 + j * 16;
                    uint2 o;
                    o.x = st_pack2(v[j][0], v[j][1]);
                    o.y = st_pack2(v[j][2], v[j][3]);
                    if (m_ok && n < p.N) *(uint2*)(orow + n) = o;
                }
            }
        }
    };

    // ---- software pipeline over (tile, slab) steps ----------------------------------------
    auto load_frags = [&](frag8_t (&xf)[FM], frag8_t (&wf)[FN], int slot, int kk) {
        const lds_char* sbase = smem + slot * STAGE;
        const int choff = frag_ch0 ^ (kk * 64);
#pragma unroll
        for (int i = 0; i < FM; ++i)
            xf[i] = *(const __attribute__((address_space(3))) frag8_t*)(sbase + a_frag_base + i * 2048 + choff);
#pragma unroll
        for (int j = 0; j < FN; ++j)
            wf[j] = *(const __attribute__((address_space(3))) frag8_t*)(sbase + b_frag_base + j * 2048 + choff);
    };
    auto mfma_block = [&](const frag8_t (&xf)[FM], const frag8_t (&wf)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = MDHIP_MFMA(wf[j], xf[i], acc[i][j]);
    };

    int issued = 0;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (issued < total_steps) { issue(st); ++issued; }

    int cur = 0;                 // ring slot of the slab being consumed
    int nxt = NS - 1;            // ring slot the next issued slab goes to
    int c_kt = 0, c_tile = first_tile;

    if constexpr (FP) {
        frag8_t xa[FM], wa[FN], xb[FM], wb[FN];
        if (NS - 2 < total_steps) wait_vmcnt<(NS - 2) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        load_frags(xa, wa, 0, 0);
        for (int step = 0; step < total_steps; ++step) {
            load_frags(xb, wb, cur, 1);              // second half of this slab ...
            mfma_block(xa, wa);                      // ... while the first half multiplies
            // the next slab must have landed (this wave's part), then everyone's
            if (step + (NS - 2) < total_steps) wait_vmcnt<(NS - 3) * LPS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (issued < total_steps) { issue(nxt); ++issued; }
            const int nslot = (cur + 1 == NS) ? 0 : cur + 1;
            if (step + 1 < total_steps) load_frags(xa, wa, nslot, 0);
            mfma_block(xb, wb);
            if (++c_kt == KT) {
                epilogue(c_tile);
                c_kt = 0;
                c_tile += tile_step;
            }
            cur = nslot;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    } else {
        for (int step = 0; step < total_steps; ++step) {
            // this wave's loads of the current slab have landed; up to NS-2 younger slabs stay in flight
            if (step + (NS - 2) < total_steps) wait_vmcnt<(NS - 2) * LPS>();
            else wait_vmcnt<0>();    // tail: fewer slabs are outstanding than the steady-state count
            __builtin_amdgcn_s_barrier();
            if (issued < total_steps) { issue(nxt); ++issued; }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                frag8_t xf[FM], wf[FN];
                load_frags(xf, wf, cur, kk);
                mfma_block(xf, wf);
            }
            if (++c_kt == KT) {      // tile finished: write it out while the next tile's slabs stream in
                epilogue(c_tile);
                c_kt = 0;
                c_tile += tile_step;
            }
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table
// ---------------------------------------------------------------------------------------
// id, BM, BN, waves along M, waves along N, LDS stages, fragment prefetch
#define MDHIP_CONV_CFGS(X)     \
    X(0, 256, 160, 4, 2, 2, 0) \
    X(1, 128, 160, 2, 2, 2, 0) \
    X(2, 256, 80, 4, 1, 2, 0)  \
    X(3, 128, 80, 4, 1, 2, 0)  \
    X(4, 256, 32, 4, 1, 2, 0)  \
    X(5, 128, 64, 2, 2, 2, 0)  \
    X(6, 128, 128, 2, 2, 2, 0) \
    X(7, 256, 128, 4, 2, 2, 0) \
    X(8, 128, 320, 2, 4, 2, 0) \
    X(9, 64, 160, 1, 2, 2, 0)  \
    X(10, 64, 64, 1, 2, 2, 0)  \
    X(11, 256, 64, 4, 1, 2, 0) \
    X(12, 256, 160, 4, 2, 3, 0) \
    X(13, 128, 160, 2, 2, 3, 1) \
    X(14, 256, 160, 4, 2, 3, 1) \
    X(15, 128, 80, 4, 1, 3, 0)  \
    X(16, 128, 80, 4, 1, 3, 1)  \
    X(17, 256, 80, 4, 1, 3, 1)  \
    X(18, 256, 128, 4, 2, 3, 0) \
    X(19, 128, 128, 2, 2, 3, 1) \
    X(20, 256, 128, 4, 2, 3, 1) \
    X(21, 64, 160, 1, 2, 4, 0)  \
    X(22, 128, 64, 2, 2, 4, 1)  \
    X(23, 256, 32, 4, 1, 4, 0)  \
    X(24, 256, 320, 2, 4, 2, 0) \
    X(25, 128, 80, 2, 1, 3, 1)  \
    X(26, 256, 320, 4, 4, 2, 0) \
    X(27, 128, 160, 4, 2, 3, 1)

static const ConvCfg g_cfgs[] = {
#define X(id, bm, bn, wm, wn, ns, fp)                                                                 \
    {bm, bn, (wm) * (wn) * 64, (size_t)conv_lds_bytes(bm, bn, ns),                                  \
     conv_blocks_per_cu(bm, bn, (wm) * (wn), ns), #bm "x" #bn "/" #wm "x" #wn "/s" #ns "/p" #fp},
    MDHIP_CONV_CFGS(X)
#undef X
};

// configuration ids: [0, kNumV1) = this file's kernel, then the other families in the order of g_fams
constexpr int kNumV1 = (int)(sizeof(g_cfgs) / sizeof(g_cfgs[0]));
int conv_num_v1_cfgs() { return kNumV1; }

namespace {
// the first-generation configurations that exist in a decoding instantiation (DEC): the narrow tiles a 24-channel op is given
constexpr bool v1_decodes(int bn) { return bn <= 80; }
template <int BM, int BN, int WM, int WN, int NS, int FP>
hipError_t v1_set_lds(int lds) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, NS, FP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (v1_decodes(BN)) {
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, NS, FP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    return e;
}
template <int BM, int BN, int WM, int WN, int NS, int FP>
hipError_t v1_launch(const ConvArgs& p, dim3 grid, size_t lds, hipStream_t s) {
    if (p.dec_pred) {
        if constexpr (v1_decodes(BN)) {
            hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, NS, FP, 1>), grid, dim3(WM * WN * 64), lds, s, p);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, NS, FP>), grid, dim3(WM * WN * 64), lds, s, p);
    return hipGetLastError();
}
bool conv2_supports_l(int cfg, const ConvArgs& a) {
    if (conv2_cfg_is_ring(cfg) && !(conv2_is_pointwise(a) && a.k_pad >= 3 * 64)) return false;
    return conv2_supports(a) && (a.in_up == nullptr || cfg == 0);                    // in_up: 160x160 only
}
// one row per kernel family after the first: local configuration ids [0, num()), developer variants
// (tools/convbench.cpp) behind them at the negative ids  dev_base - k
struct Family {
    int (*num)();
    const ConvCfg& (*cfg)(int);
    bool (*supports)(int, const ConvArgs&);
    hipError_t (*launch)(int, const ConvArgs&, hipStream_t);
    hipError_t (*init)();
    bool bitwise;        // false: equals the implicit-GEMM kernels up to fp32 summation order only (other K order)
    int dev_base;        // developer variant k (0, 1, ...) is addressed as  dev_base - k
    bool f8_in;          // takes e4m3 activations (and nothing else)
    bool f8_out;         // its epilogue can write e4m3 (ConvArgs::out_f8)
};
const Family g_fams[] = {
    {conv2_num_cfgs, conv2_cfg, conv2_supports_l, conv2_launch, conv2_init, true, -1, false, true},
    // (conv_v4.cpp -- the row-patch direct convolution of round 1, conv_v5's predecessor and the origin of the (group, r, s, c)
    // weight packing -- left the build in round 5: no table entry had selected it since round 3)
    {conv5_num_cfgs, conv5_cfg, conv5_supports, conv5_launch, conv5_init, false, -301, false, false},
    {conv6_num_cfgs, conv6_cfg, conv6_supports, conv6_launch, conv6_init, true, -401, false, false},     // the stem kernel: same K order
    {conv8_num_cfgs, conv8_cfg, conv8_supports, conv8_launch, conv8_init, false, -801, true, false},
    {conv7_num_cfgs, conv7_cfg, conv7_supports, conv7_launch, conv7_init, false, -901, false, false},    // stride-2 row runs: a K order of its own
};
constexpr int kNumFams = (int)(sizeof(g_fams) / sizeof(g_fams[0]));
// family and local id of a global id >= kNumV1
const Family* find_family(int cfg, int* local) {
    int i = cfg - kNumV1;
    for (int f = 0; f < kNumFams; ++f) {
        const int n = g_fams[f].num();
        if (i < n) { *local = i; return &g_fams[f]; }
        i -= n;
    }
    return nullptr;
}
}  // namespace

int conv_num_cfgs() {
    int n = kNumV1;
    for (int f = 0; f < kNumFams; ++f) n += g_fams[f].num();
    return n;
}
const ConvCfg& conv_cfg(int i) {
    if (i < kNumV1) return g_cfgs[i];
    int l = 0;
    const Family* f = find_family(i, &l);
    return f ? f->cfg(l) : g_cfgs[0];
}
bool conv_cfg_is_bitwise_family(int cfg) {
    if (cfg < kNumV1) return true;
    int l = 0;
    const Family* f = find_family(cfg, &l);
    return f ? f->bitwise : true;
}

// (a.dec_pred: the op decodes in its epilogue -- only the configurations with such an instantiation take it)
bool conv_cfg_decodes(int cfg) {
    if (cfg < 0 || cfg >= conv_num_cfgs()) return false;
    if (cfg < kNumV1) return v1_decodes(g_cfgs[cfg].bn);
    int l = 0;
    const Family* f = find_family(cfg, &l);
    return f == &g_fams[0] && conv2_cfg_decodes(l);
}

bool conv_supports(int cfg, const ConvArgs& a) {
    if (cfg < 0 || cfg >= conv_num_cfgs()) return false;
    if (a.dec_pred && !(conv_cfg_decodes(cfg) && a.out_f32 && (a.N % 8) == 0)) return false;
    if (cfg < kNumV1) return !a.in_f8;                              // the first-generation kernel takes every 16-bit op
    int l = 0;
    const Family* f = find_family(cfg, &l);
    if (!f || f->f8_in != (a.in_f8 != 0) || (a.out_f8 && !f->f8_out)) return false;
    return f->supports(l, a);
}

hipError_t conv_init() {
    hipError_t e = hipSuccess;
#define X(id, bm, bn, wm, wn, ns, fp) \
    if (e == hipSuccess) e = v1_set_lds<bm, bn, wm, wn, ns, fp>((int)g_cfgs[id].lds_bytes);
    MDHIP_CONV_CFGS(X)
#undef X
    for (int f = 0; f < kNumFams && e == hipSuccess; ++f) e = g_fams[f].init();
    return e;
}

hipError_t conv_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    // negative ids address the developer variants (tools/convbench.cpp): family f's k-th variant is dev_base - k
    if (cfg >= conv_num_cfgs()) return hipErrorInvalidValue;
    if (cfg < 0) {
        for (int f = kNumFams - 1; f >= 0; --f)
            if (cfg <= g_fams[f].dev_base) return g_fams[f].launch(g_fams[f].num() + (g_fams[f].dev_base - cfg), a, s);
        return hipErrorInvalidValue;
    }
    if (cfg >= kNumV1) {
        int l = 0;
        const Family* f = find_family(cfg, &l);
        if (!f || f->f8_in != (a.in_f8 != 0) || (a.out_f8 && !f->f8_out)) return hipErrorInvalidValue;
        return f->launch(l, a, s);
    }
    if (a.in_f8) return hipErrorInvalidValue;
    const ConvCfg& c = g_cfgs[cfg];
    ConvArgs p = a;
    conv_set_rcp(p);
    p.tiles_n = (a.n_rows + c.bn - 1) / c.bn;
    p.tiles_m = (a.M + c.bm - 1) / c.bm;
    // persistent streams: about as many workgroups as fit on the chip at once (32 CUs per XCD);
    // the load pipeline of a stream runs across its tile boundaries
    p.tiles_per_xcd = (p.tiles_m + 7) / 8;
    p.m_streams = std::max(1, std::min(p.tiles_per_xcd, (32 * c.blocks_per_cu) / p.tiles_n));
    const dim3 grid((unsigned)(8 * p.tiles_n * p.m_streams));
    if (p.dec_pred && !conv_supports(cfg, p)) return hipErrorInvalidValue;
    switch (cfg) {
#define X(id, bm, bn, wm, wn, ns, fp) \
    case id: return v1_launch<bm, bn, wm, wn, ns, fp>(p, grid, c.lds_bytes, s);
        MDHIP_CONV_CFGS(X)
#undef X
    }
    return hipErrorInvalidValue;
}

}  // namespace MDHIP_ST
}  // namespace mdhip
