// The stem of the network as its own kernel (gfx950 / MI355X): 6x6 / stride 2 conv on the space-to-depth input =
// 3x3 / stride 1 over 16-channel (32-byte) pixels, 80 output channels, K = 144 (+ padding to 160).
//
// Why: for this GEMM (M = 13.1 M pixels per batch of 32, N = 80, K = 144) the generic implicit-GEMM kernel spends three
// 64-deep slabs of main loop per tile and then a full SiLU + store epilogue: 1.24 ms per step, 2.0 TB/s, 186 TFLOP/s --
// 0.24 of the HBM roofline of a layer that only has to read 0.42 GB and write 2.1 GB.  Here
//   * every wave keeps ALL weight fragments of its 80 output channels in registers for the whole kernel
//     (5 fragment columns x 5 k-steps x 4 VGPRs = 100): no weight traffic, no weight LDS reads, no weight stages;
//   * a tile is 128 consecutive output pixels of ONE image row; its three input row segments (130 pixels x 32 bytes)
//     come in by LDS-DMA (15 pieces of 1 KiB, out-of-image pixels read zeros) into one of two buffers while the previous
//     tile computes; a wave reads 10 activation fragments per tile;
//   * one barrier per tile; stores are buffer stores with exact instruction counts (out-of-row pixels are dropped by the
//     descriptor, not by a branch), so the wait in front of the next tile's fragments is a COUNTED vmcnt that leaves the
//     epilogue's stores in flight.
// K order (r, s, c) and operand roles are the implicit-GEMM kernel's, k-steps are accumulated in the same order: results
// are bit-identical to conv_igemm.cpp's (the configuration belongs to the bitwise family).

#include <algorithm>

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

[[maybe_unused]] constexpr int kBM = 128;                    // output pixels per tile (one image row segment)
constexpr int kNW = 4;                      // waves; each owns 32 pixels x 80 channels
[[maybe_unused]] constexpr int kFM = 2, kFN = 5, kKS = 5;    // fragment rows / columns per wave, 32-deep k-steps (K = 144 -> 160)
constexpr int kRunPieces = 5;               // 160 pixels x 32 bytes per input row segment
constexpr int kRunBytes = kRunPieces * 1024;
constexpr int kBufBytes = 3 * kRunBytes;    // three kernel rows
constexpr int kBiasOff = 2 * kBufBytes;
constexpr int kStemLds = kBiasOff + 80 * 4;

}  // namespace

#define MDHIP_DMA16(rsrc, lptr, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr), 16, (voff), (soff), 0, 0)

__global__ void __launch_bounds__(kNW * 64, 2)
conv_stem_kernel(const ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m15 = lane & 15, kb = lane >> 4;

    // tiles: (image, row, 128-pixel segment), contiguous ranges per XCD, interleaved over its workgroups
    const int tiles_x = (p.W + kBM - 1) / kBM;
    const int n_img = p.M / p.HoWo;
    const int total = n_img * p.H * tiles_x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (total + 7) / 8;
    const int t_lo = xcd * per_xcd, t_hi = min(t_lo + per_xcd, total);
    int tile = t_lo + slot;
    if (tile >= t_hi) return;

    for (int c = tid; c < 80; c += kNW * 64) *(__attribute__((address_space(3))) float*)(smem + kBiasOff + c * 4) = p.bias[c];

    // every weight fragment this wave will ever use: column j, k-step t -> rows j*16 + (lane & 15), k = t*32 + kb*8 ..
    frag8_t wreg[kFN][kKS];
#pragma unroll
    for (int j = 0; j < kFN; ++j)
#pragma unroll
        for (int t = 0; t < kKS; ++t)
            wreg[j][t] = *(const frag8_t*)(p.wgt + (size_t)(j * 16 + m15) * p.k_pad + t * 32 + kb * 8);

    // activation fragment of k-step t: taps 2t and 2t+1 (16 channels each); this lane's 8 k values = channels
    // (kb & 1) * 8 .. of tap 2t + (kb >> 1).  Tap 9 does not exist (its weights are zero): read tap 8 instead.
    unsigned aoff[kKS];
#pragma unroll
    for (int t = 0; t < kKS; ++t) {
        const int tap = min(2 * t + (kb >> 1), 8);
        const int r = tap / 3, s = tap - 3 * r;
        aoff[t] = (unsigned)(r * kRunBytes + (wave * 32 + m15 + s) * 32 + (kb & 1) * 16);
    }

    // run loader: piece pc (0 .. 14) = kernel row pc / 5, pixels 32 * (pc % 5) .. + 31 of the segment that starts at x0 - 1
    const int lq = lane >> 1, lh = lane & 1;
    auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
        const int xt = t % tiles_x;
        const int yb = t / tiles_x;                   // image * H + y
        const int y = yb % p.H, b = yb / p.H;
        const int x0 = xt * kBM;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.in + (size_t)b * p.HoWo * p.ld_in), 0, kNumRecords, 0x00020000);
#pragma unroll
        for (int k = 0; k < (3 * kRunPieces + kNW - 1) / kNW; ++k) {
            const int pc = k * kNW + wave;
            if (pc >= 3 * kRunPieces) break;                                          // wave-uniform
            const int r = pc / kRunPieces, q = (pc - r * kRunPieces) * 32 + lq;
            const int iy = y + r - 1, ix = x0 - 1 + q;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = ok ? (unsigned)(((size_t)iy * p.W + ix) * p.ld_in * 2 + lh * 16) : kOOB;
            MDHIP_DMA16(rsrc, smem + buf * kBufBytes + pc * 1024, off, 0);
        }
    };

    // output: buffer stores through a per-image descriptor; a pixel beyond the row end gets an offset the descriptor
    // rejects (the store instruction is still issued: exact vmcnt bookkeeping)
    const int img_out_bytes = p.HoWo * p.ld_out * 2;

    f32x4 acc[kFM][kFN];
#pragma unroll
    for (int i = 0; i < kFM; ++i)
#pragma unroll
        for (int j = 0; j < kFN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_tile(tile, 0);
    int buf = 0;
    const int q4 = kb;
    bool first = true;
    for (; tile < t_hi; tile += slots) {
        // the DMA pieces of this tile are older than the (exactly 6) stores of the previous tile's epilogue
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();
        if (tile + slots < t_hi) issue_tile(tile + slots, buf ^ 1);
        const unsigned base = (unsigned)(buf * kBufBytes);
#pragma unroll
        for (int t = 0; t < kKS; ++t) {
            frag8_t xa[kFM];
#pragma unroll
            for (int i = 0; i < kFM; ++i)
                xa[i] = *(const __attribute__((address_space(3))) frag8_t*)(smem + base + aoff[t] + i * 512);
#pragma unroll
            for (int j = 0; j < kFN; ++j)
#pragma unroll
                for (int i = 0; i < kFM; ++i) acc[i][j] = MDHIP_MFMA(wreg[j][t], xa[i], acc[i][j]);
        }
        // ---- epilogue: bias, SiLU, 16-bit, lane exchange -> 16-byte stores --------------------------------------------
        const int xt = tile % tiles_x;
        const int yb = tile / tiles_x;
        const int y_img = yb % p.H, b_img = yb / p.H;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((char*)p.out + (size_t)b_img * img_out_bytes), 0, img_out_bytes, 0x00020000);
        const int x0 = xt * kBM + wave * 32 + m15;
#pragma unroll
        for (int i = 0; i < kFM; ++i) {
            const int x = x0 + i * 16;
            const bool ok = x < p.W;
            const unsigned row_off = (unsigned)((y_img * p.W + x) * p.ld_out * 2);
            float v[kFN][4];
#pragma unroll
            for (int j = 0; j < kFN; ++j) {
                const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(smem + kBiasOff + (j * 16 + q4 * 4) * 4);
                mdhip_bias4(acc[i][j], bv, v[j]);
                mdhip_silu4(v[j]);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j + 1 < kFN; j += 2) {
                unsigned a0 = st_pack2(v[j][0], v[j][1]), a1 = st_pack2(v[j][2], v[j][3]);
                unsigned b0 = st_pack2(v[j + 1][0], v[j + 1][1]), b1 = st_pack2(v[j + 1][2], v[j + 1][3]);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                auto t0 = __builtin_amdgcn_permlane16_swap(s0[0], s0[1], false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                const u32x4 d = {t0[0], t1[0], t0[1], t1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(d, o_rsrc, ok ? row_off + (unsigned)((j * 16 + q4 * 8) * 2) : kOOB, 0, 0);
            }
            {
                constexpr int j = kFN - 1;
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                const u32x2 d = {st_pack2(v[j][0], v[j][1]), st_pack2(v[j][2], v[j][3])};
                __builtin_amdgcn_raw_buffer_store_b64(d, o_rsrc, ok ? row_off + (unsigned)((j * 16 + q4 * 4) * 2) : kOOB, 0, 0);
            }
        }
        buf ^= 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---------------------------------------------------------------------------------------
// configuration table: one configuration
// ---------------------------------------------------------------------------------------
static const ConvCfg g_cfg6 = {kBM, 80, kNW * 64, (size_t)kStemLds, 2, "stem:row128x80/4x1"};

int conv6_num_cfgs() { return 1; }
const ConvCfg& conv6_cfg(int) { return g_cfg6; }

hipError_t conv6_init() {
    return hipFuncSetAttribute((const void*)conv_stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kStemLds);
}

bool conv6_supports(int cfg, const ConvArgs& a) {
    // the stem as the planner lowers it: 3x3 / stride 1 / pad 1 over 16-channel pixels, exactly 80 output channels,
    // SiLU, 16-bit output, no residual; per-image offsets and the output tensor inside the 31-bit buffer range
    return cfg == 0 && !a.in_f8 && !a.out_f8 && !a.out_f32 && a.res == nullptr && a.act == 1 && a.ntaps == 9 && a.kw == 3 &&
           a.stride == 1 && a.pad == 1 && a.C8 == 2 && a.ld_in == 16 && a.N == 80 && a.n_rows == 80 && a.k_pad >= 160 &&
           a.Ho == a.H && a.Wo == a.W && (long long)a.HoWo * a.ld_in * 2 < 0x7fffffffLL &&
           (long long)a.HoWo * a.ld_out * 2 < 0x7fffffffLL;
}

hipError_t conv6_launch(int cfg, const ConvArgs& a, hipStream_t s) {
    if (!conv6_supports(cfg, a)) return hipErrorInvalidValue;
    const int tiles_x = (a.W + kBM - 1) / kBM;
    const long long total = (long long)(a.M / a.HoWo) * a.H * tiles_x;
    const int slots = (int)std::max(1LL, std::min<long long>(64, (total + 7) / 8));      // 2 workgroups on each of 32 CUs per XCD
    hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)(8 * slots)), dim3(kNW * 64), kStemLds, s, a);
    return hipGetLastError();
}

}  // namespace MDHIP_ST
}  // namespace mdhip
