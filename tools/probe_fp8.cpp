// Developer probe (not part of the product): pins down the operand layout of the gfx950 block-scaled
// MFMA  v_mfma_scale_f32_16x16x128_f8f6f4  with fp8 (e4m3, OCP) operands and unit scales, which the fp8
// convolution kernels (conv_f8.cpp) rely on.
//
//   D[i][j] = sum_k A[i][k] * B[k][j],   i, j < 16, k < 128
//
// Hypothesis H1 (what the kernels assume): lane l holds row i = l & 15 of A (column j = l & 15 of B) and the 32
// consecutive k values  32 * (l >> 4) .. + 31, byte b of the lane's 8 VGPRs = k offset b; scale byte 0x7f = 2^0.
// H2: the two 16-byte halves interleave: k = 64 * half + 16 * (l >> 4) + byte.
// The probe fills A and B with random e4m3 values, runs one MFMA and compares with a host evaluation.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe_fp8.cpp -o build/probe_fp8 && build/probe_fp8

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* a_lane, const uint8_t* b_lane, float* d, unsigned scale) {
    const int l = threadIdx.x;
    v8i a = *(const v8i*)(a_lane + l * 32);
    v8i b = *(const v8i*)(b_lane + l * 32);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)scale, 0, (int)scale);
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r
    for (int r = 0; r < 4; ++r) d[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

static float e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);                 // subnormal: m * 2^-3 * 2^-6
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(128 * 16);
    srand(7);
    auto rnd = [] {
        uint8_t v;
        do { v = (uint8_t)(rand() & 0xff); } while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 9);   // no NaN, |x| <= 7
        return v;
    };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    std::vector<double> ref(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 128; ++k) s += (double)e4m3_to_f32(A[i * 128 + k]) * e4m3_to_f32(B[k * 16 + j]);
            ref[i * 16 + j] = s;
        }
    uint8_t *da, *db;
    float* dd;
    hipMalloc(&da, 64 * 32);
    hipMalloc(&db, 64 * 32);
    hipMalloc(&dd, 256 * 4);
    for (int hyp = 1; hyp <= 2; ++hyp) {
        std::vector<uint8_t> al(64 * 32), bl(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int b = 0; b < 32; ++b) {
                const int k = hyp == 1 ? 32 * (l >> 4) + b : 64 * (b >> 4) + 16 * (l >> 4) + (b & 15);
                al[l * 32 + b] = A[(l & 15) * 128 + k];
                bl[l * 32 + b] = B[k * 16 + (l & 15)];
            }
        hipMemcpy(da, al.data(), al.size(), hipMemcpyHostToDevice);
        hipMemcpy(db, bl.data(), bl.size(), hipMemcpyHostToDevice);
        for (unsigned scale : {0x7f7f7f7fu, 0x0000007fu, 0u}) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, scale);
            std::vector<float> d(256);
            hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
            double emax = 0, rmax = 0;
            for (int i = 0; i < 256; ++i) {
                emax = fmax(emax, fabs(d[i] - ref[i]));
                rmax = fmax(rmax, fabs(ref[i]));
            }
            printf("H%d scale %08x: max|err| %.6g (max|ref| %.4g)  d[0] %.6g ref[0] %.6g  ratio %.6g %s\n", hyp, scale, emax, rmax,
                   d[0], ref[0], d[0] / ref[0], emax <= 1e-3 * rmax ? "MATCH" : "");
        }
    }
    return 0;
}
