// Measured peaks of the box next to the vendor figures: an HBM stream test
// (copy and read-only, 16 bytes per lane, grid-stride) and a register-only MFMA loop
// (v_mfma_f32_16x16x32_bf16, independent accumulators, no memory traffic).
//
// Build:  bash tools/build_convbench.sh   (builds build/peaks as well)
// Run:    build/peaks
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

__global__ void __launch_bounds__(256) copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

__global__ void __launch_bounds__(256) read_kernel(const uint4* __restrict__ in, unsigned* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345679u) out[0] = acc;      // never true for the test pattern: keeps the loads alive
}

__global__ void __launch_bounds__(256) write_kernel(uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

// 16 independent accumulators per wave, `iters` rounds of 16 MFMAs each
__global__ void __launch_bounds__(256) mfma_kernel(float* out, int iters, unsigned long long* cyc = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 a = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (short)blockIdx.x};
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
    if (cyc && blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = __builtin_amdgcn_s_memtime() - t0;
#endif
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, clock %d MHz, memory clock %d MHz, bus %d bit\n", prop.name, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth);
    const size_t bytes = (size_t)4 << 30;       // 4 GiB per buffer: far beyond L2 + MALL
    uint4 *a, *b;
    unsigned* flag;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&flag, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    const size_t n = bytes / 16;
    const int grid = prop.multiProcessorCount * 16;
    float ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n); }, 5);
    printf("HBM copy  (read + write): %7.1f GB/s  (%.2f ms for 2 x %.1f GiB)\n", 2.0 * bytes / ms / 1e6, ms, bytes / 1073741824.0);
    ms = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, flag, n); }, 5);
    printf("HBM read  only          : %7.1f GB/s  (%.2f ms)\n", 1.0 * bytes / ms / 1e6, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, b, n); }, 5);
    printf("HBM write only          : %7.1f GB/s  (%.2f ms)\n", 1.0 * bytes / ms / 1e6, ms);

    float* out;
    CK(hipMalloc(&out, 4));
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = prop.multiProcessorCount * waves_per_simd;      // 4 waves per block = one per SIMD
        unsigned long long* cyc;
        CK(hipMalloc(&cyc, 8));
        ms = time_ms([&] { hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); }, 3);
        unsigned long long hc = 0;
        CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
        const double flops = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
        printf("MFMA 16x16x32 bf16, %d wave(s)/SIMD: %7.1f TFLOP/s  (%.2f ms; %.0f s_memtime cycles per launch = %.2f GHz shader clock "
               "during the loop, %.1f cycles per MFMA and SIMD)\n", waves_per_simd, flops / ms / 1e9, ms, (double)hc,
               (double)hc / (ms * 1e6), (double)hc / ((double)iters * 16 * waves_per_simd));
    }
    return 0;
}
