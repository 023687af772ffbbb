// What the L2 -> LDS path (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction) delivers per CU on this box, with and
// without MFMAs issued between the pieces -- the ceiling the conv kernels' loaders run against.
//
//   source span  2 MiB  = every CU walks the same L2-resident buffer (what weights look like)
//               64 MiB  = memory-side cache
//                4 GiB  = HBM stream
//   waves / CU   4, 8, 16 (one workgroup per CU; 8 = the 8-wave conv tile)
//   MFMAs between two pieces of a wave: 0, 5 (1x1 conv, 160x160 tile: 0.2 pieces per MFMA), 12 (3x3 row-run tile: 0.085)
//   in flight    pieces a wave keeps outstanding (8 or 16)
//
// Build:  bash tools/build_convbench.sh   (builds build/dma_peak as well)     Run:  build/dma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) char lds_char;

// every wave: `rounds` rounds of R pieces (1 KiB each) into its own R KiB of LDS, M MFMAs behind every piece, the round
// before the last one awaited before the next is issued (2R pieces of the wave in flight at most)
template <int WAVES, int R, int M>
__global__ void __launch_bounds__(WAVES * 64) dma_kernel(const char* src, unsigned span_mask, int rounds, float* out,
                                                         unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    // a conv loader's pattern: 8 rows of 128 bytes per piece (lane -> row lane / 8, chunk lane % 8), rows 4 KiB apart
    const unsigned lane_off = (unsigned)((lane >> 3) * 4096 + (lane & 7) * 16);
    unsigned pos = ((unsigned)blockIdx.x * 40503u * 32768u + (unsigned)wave * 128u) & span_mask;
    bf16x8 a = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (short)blockIdx.x};
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            unsigned so = pos;
            asm volatile("" : "+s"(so));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, smem + (wave * R + i) * 1024, 16, lane_off, so, 0, 0);
            pos = (pos + 32768u + 128u * WAVES) & span_mask;      // next 8-row block (the waves of a CU interleave in a row)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 12], 0, 0, 0);
        }
        if (R == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += (float)smem[(threadIdx.x * 16) & 8191];
    if (s == 12345.678f) out[0] = s;
    if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <int WAVES, int R, int M>
static void run(const char* label, const char* src, size_t span, int blocks, int rounds, float* out, unsigned long long* cyc) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t lds = (size_t)WAVES * R * 1024;
    CK(hipFuncSetAttribute((const void*)dma_kernel<WAVES, R, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned mask = (unsigned)(span - 1) & ~127u;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((dma_kernel<WAVES, R, M>), dim3(blocks), dim3(WAVES * 64), lds, 0, src, mask, rounds, out, cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    unsigned long long hc[8];
    CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < 8; ++i) c += (double)hc[i] / 8;
    const double bytes = (double)blocks * WAVES * rounds * R * 1024.0;
    const double mfma = (double)blocks * WAVES * rounds * R * M;
    printf("%-6s %2d waves/CU, %2d in flight/wave, %2d MFMA/piece: %7.1f GB/s = %5.1f B/clk/CU  (%.3f ms, %.0f cycles, %.2f GHz; "
           "%.0f cycles per piece and CU", label, WAVES, 2 * R, M, bytes / best / 1e6, bytes / blocks / c, best, c, c / (best * 1e6),
           c / ((double)WAVES * rounds * R));
    if (M) printf("; MFMA pipe %.0f %% busy", 100.0 * (mfma / blocks / 4 * 16.0) / c);
    printf(")\n");
}


// ---------------------------------------------------------------------------------------------------------------------
// The main loop of the 8-wave conv tile as a skeleton (80x80 wave tiles: 25 accumulators, 5 + 5 fragments per k half;
// a step = two halves with a barrier between them; the second half issues the DMA pieces): which ingredient costs what.
//   READS   fragment reads from LDS (10 ds_read_b128 per half and wave) feeding the MFMAs; without them the operands are
//           constants
//   NB, NA  DMA pieces per wave and step from the L2-resident buffer (weights) / from the `span` buffer (activations)
//   BAR     s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier in the middle of every step
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short frag8;

template <bool READS, int NB, int NA, bool BAR>
__global__ void __launch_bounds__(512, 2) loop_model(const char* wsrc, const char* asrc, unsigned a_mask, int steps, float* out,
                                                     unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    lds_char* const smem = (lds_char*)smem_generic;
    constexpr int A_BUF = 41 * 1024, B_OFF = 2 * A_BUF, B_BYTES = 20 * 1024;       // the 320x160 tile's LDS map
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)asrc, 0, 0x7fffffff, 0x00020000);
    const unsigned lane_off = (unsigned)((lane >> 3) * 2880 + (lane & 7) * 16);     // rows of a 1440-deep weight matrix
    const unsigned lane_off_a = (unsigned)((lane >> 3) * 320 + (lane & 7) * 16);    // pixels of a 160-channel tensor
    unsigned wpos = (unsigned)wave * 23040u, apos = ((unsigned)blockIdx.x * 40503u * 65536u) & a_mask;
    // fragment addresses as in conv_v5 (128-byte rows, XOR swizzle)
    const unsigned a_frag = (unsigned)((wm * 80 + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4));
    const unsigned b_frag = (unsigned)(B_OFF + (wn * 80 + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4));
    auto rd = [&](unsigned a) -> frag8 { return *(const __attribute__((address_space(3))) frag8*)(smem + a); };
    frag8 xa[5], wa[5], xb[5], wb[5];
    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        xa[i] = xb[i] = frag8{(short)lane, 1, 2, 3, 4, 5, 6, (short)i};
        wa[i] = wb[i] = frag8{7, 6, 5, 4, 3, 2, 1, (short)(lane + i)};
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < steps; ++st) {
        const int cur = st & 1;
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            if (READS) {
                wb[g] = rd((b_frag + cur * B_BYTES + g * 2048) ^ 64u);
                xb[g] = rd((a_frag + cur * A_BUF + g * 2048) ^ 64u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], xa[i], acc[i][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            if (READS) {
                wa[g] = rd(b_frag + (cur ^ 1) * B_BYTES + g * 2048);
                xa[g] = rd(a_frag + (cur ^ 1) * A_BUF + g * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], xb[i], acc[i][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g < NB) {
                unsigned so = wpos + (unsigned)st * 128u + (unsigned)g * 184320u;
                asm volatile("" : "+s"(so));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, smem + B_OFF + cur * B_BYTES + (g * 8 + wave) * 1024 % B_BYTES, 16, lane_off, so & 0x1fffffu, 0, 0);
            } else if (g < NB + NA) {
                unsigned so = apos;
                asm volatile("" : "+s"(so));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, smem + (cur ^ 1) * A_BUF + (((g - NB) * 8 + wave) * 1024) % A_BUF, 16, lane_off_a + so, 0, 0, 0);
                apos = (apos + 2560u * 8u) & a_mask;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!BAR) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[0] = s;
    if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <bool READS, int NB, int NA, bool BAR>
static void run_model(const char* label, const char* wsrc, const char* asrc, size_t span, int blocks, float* out, unsigned long long* cyc) {
    const int steps = 2000;
    const size_t lds = 123904;
    CK(hipFuncSetAttribute((const void*)loop_model<READS, NB, NA, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned mask = (unsigned)(span - 1) & ~127u;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((loop_model<READS, NB, NA, BAR>), dim3(blocks), dim3(512), lds, 0, wsrc, asrc, mask, steps, out, cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    unsigned long long hc[64];
    CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < 64; ++i) c += (double)hc[i] / 64;
    const double flops = (double)blocks * 8 * steps * 50 * (2.0 * 16 * 16 * 32);
    printf("model %-5s reads %d, DMA %d (L2) + %d (%s) pieces per wave and step, barrier %d: %6.0f cycles per step (1600 = matrix pipe full), "
           "%6.1f TFLOP/s, %.2f GHz, DMA %.1f B/clk/CU\n", label, (int)READS, NB, NA, label, (int)BAR, c / steps, flops / best / 1e9,
           c / (best * 1e6), (double)(NB + NA) * 8 * 1024 / (c / steps));
}

// ---------------------------------------------------------------------------------------------------------------------
// The epilogue's stores: what a 16-byte-per-lane buffer store costs by shape.  Every wave writes `rounds` groups of 16 pixels
// x 80 channels (2560 bytes) of a [pixels][NCH] bf16 tensor, as the conv epilogue does: SHAPE 0 = fragment-shaped (lane =
// (pixel, 16-byte chunk): 2 x b128 + 1 x b64 per group, each instruction touches 16 pixels x 64 bytes), SHAPE 1 = line-shaped
// (the same bytes, but a wave instruction covers consecutive 16-byte chunks of consecutive pixels: 160 bytes per pixel -> 6.4
// pixels per instruction, 2.5 instructions per group), SHAPE 2 = fully contiguous 1 KiB per instruction (what an N tile that
// holds every channel of the tensor could write).  8 waves per CU, one workgroup per CU, every CU its own region.
// ---------------------------------------------------------------------------------------------------------------------
template <int SHAPE>
__global__ void __launch_bounds__(512) store_model(char* out, int nch, int rounds, unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned row = (unsigned)nch * 2u;                         // bytes per pixel
    const size_t region = (size_t)rounds * 64 * row;                  // a workgroup writes `rounds` groups of 4 x 16 pixels
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)blockIdx.x * region), 0, (int)region, 0x00020000);
    const u32x4 d = {(unsigned)lane, 1u, 2u, (unsigned)wave};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const unsigned g0 = (unsigned)(r * 64 + wm * 16) * row + (unsigned)(wn * 160);     // the wave's 16 pixels x 80 channels
        if (SHAPE == 0) {
            const unsigned o = g0 + (unsigned)(lane & 15) * row + (unsigned)(lane >> 4) * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)(o + 64u), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{d[0], d[1]}, rsrc, (int)(g0 + (unsigned)(lane & 15) * row + 128u + (unsigned)(lane >> 4) * 8u), 0, 0);
        } else if (SHAPE == 1) {
            // chunk c = 0..159 of the group: pixel c / 10, 16-byte chunk c % 10 of its 160 bytes
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned c = (unsigned)(k * 64 + lane);
                const unsigned px = c / 10u, ch = c - px * 10u;
                const unsigned o = c < 160u ? g0 + px * row + ch * 16u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)o, 0, 0);
            }
        } else {
            // 2560 contiguous bytes per wave and group (as if the tensor had exactly this tile's channels)
            const unsigned g1 = (unsigned)((r * 8 + wave) * 2560);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned c = (unsigned)(k * 64 + lane);
                __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)(c < 160u ? g1 + c * 16u : 0x80000000u), 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (cyc && threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <int SHAPE>
static void run_store(const char* label, char* out, int nch, int blocks, unsigned long long* cyc) {
    const int rounds = 180;                       // 180 x 64 pixels x 640 bytes x 256 workgroups = 1.9 GB: inside the 2 GiB buffer
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((store_model<SHAPE>), dim3(blocks), dim3(512), 0, 0, out, nch, rounds, cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    unsigned long long hc[64];
    CK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < 64; ++i) c += (double)hc[i] / 64;
    const double bytes = (double)blocks * rounds * 8 * 2560.0;
    printf("stores %-18s %3d channels per pixel: %7.1f GB/s, %6.0f cycles per group of 8 waves x 16 pixels x 80 channels and CU (%.3f ms, %.2f GHz)\n",
           label, nch, bytes / best / 1e6, c / rounds, best, c / (best * 1e6));
}

int main(int argc, char** argv) {
    const bool only_stores = argc > 1 && argv[1][0] == 's';
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, cus);
    const size_t big = (size_t)2 << 30;
    char* src;
    float* out;
    unsigned long long* cyc;
    CK(hipMalloc(&src, big + (16 << 20)));
    CK(hipMemset(src, 1, big + (16 << 20)));
    CK(hipMalloc(&out, 4));
    CK(hipMalloc(&cyc, 8 * 4096));
    if (only_stores) {
        for (int nch : {160, 320}) {
            run_store<0>("fragment-shaped", src, nch, cus, cyc);
            run_store<1>("line-shaped", src, nch, cus, cyc);
            run_store<2>("contiguous", src, nch, cus, cyc);
        }
        return 0;
    }
    struct { const char* label; size_t span; } spans[] = {{"L2", (size_t)2 << 20}, {"MALL", (size_t)64 << 20}, {"HBM", big}};
    for (auto& sp : spans) {
        const int rounds = 400;
        run<4, 8, 0>(sp.label, src, sp.span, cus, rounds * 2, out, cyc);
        run<8, 8, 0>(sp.label, src, sp.span, cus, rounds, out, cyc);
        run<16, 8, 0>(sp.label, src, sp.span, cus, rounds / 2, out, cyc);
        run<8, 16, 0>(sp.label, src, sp.span, cus, rounds / 2, out, cyc);
        run<8, 8, 5>(sp.label, src, sp.span, cus, rounds, out, cyc);
        run<8, 8, 12>(sp.label, src, sp.span, cus, rounds, out, cyc);
        run<8, 16, 12>(sp.label, src, sp.span, cus, rounds / 2, out, cyc);
        run<4, 8, 12>(sp.label, src, sp.span, cus, rounds, out, cyc);
    }
    for (auto& sp : spans) {
        run_model<false, 0, 0, false>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<false, 0, 0, true>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<true, 0, 0, false>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<true, 0, 0, true>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<false, 3, 2, true>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<true, 3, 0, true>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<true, 3, 2, true>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<true, 3, 2, false>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
        run_model<true, 5, 0, true>(sp.label, src, src + (4 << 20), sp.span, cus, out, cyc);
    }
    for (int nch : {160, 320}) {
        run_store<0>("fragment-shaped", src, nch, cus, cyc);
        run_store<1>("line-shaped", src, nch, cus, cyc);
        run_store<2>("contiguous", src, nch, cus, cyc);
    }
    return 0;
}
