// Developer micro-benchmark for the implicit-GEMM conv kernels (not part of the product):
// runs one conv shape with a list of tile configurations on random bf16 data, checks the
// result against a naive GPU reference and prints ms / TFLOP/s.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/convbench.cpp megadetector_amd/csrc/conv_igemm.o -o gpurun_out/convbench
//   ./convbench <shape> <iters> <cfg> [<cfg> ...]
// shapes:  name  batch H W C_in C_out k stride   (see table below)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../megadetector_amd/csrc/mdhip_internal.h"

using namespace mdhip;
using namespace mdhip::st_bf16;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct Shape { const char* name; int b, h, w, cin, cout, k, s; bool res; };
static const Shape kShapes[] = {
    {"l26_3x3", 32, 80, 80, 320, 320, 3, 1, false},     // M 204800 N 320 K 2880  (x20, 20% of the net)
    {"l6_3x3r", 32, 80, 80, 320, 320, 3, 1, true},      // same with residual (backbone)
    {"l23_3x3", 32, 160, 160, 160, 160, 3, 1, false},   // M 819200 N 160 K 1440
    {"l4_3x3r", 32, 160, 160, 160, 160, 3, 1, true},    // same with residual (backbone C3 of layer 4)
    {"l8_3x3r", 32, 40, 40, 480, 480, 3, 1, true},      // M 51200 N 480 K 4320 with residual
    {"l2_3x3", 32, 320, 320, 80, 80, 3, 1, true},       // M 3276800 N 80 K 720
    {"l29_3x3", 32, 40, 40, 480, 480, 3, 1, false},     // M 51200 N 480 K 4320
    {"l32_3x3", 32, 20, 20, 640, 640, 3, 1, false},     // M 12800 N 640 K 5760
    {"gemm_l26", 32, 80, 80, 2880, 320, 1, 1, false},   // the L26 3x3's GEMM shape as a plain GEMM (1x1 over 2880 channels): M 204800 N 320 K 2880
    {"gemm_l23", 32, 160, 160, 1472, 160, 1, 1, false}, // M 819200 N 160 K 1472 (= 23 slabs; the L23 3x3 has 1440)
    {"l26_1x1", 32, 80, 80, 320, 320, 1, 1, false},     // M 204800 N 320 K 320
    {"l23_1x1", 32, 160, 160, 160, 160, 1, 1, false},
    {"l2_1x1", 32, 320, 320, 80, 80, 1, 1, false},
    {"l26_cv3", 32, 80, 80, 640, 640, 1, 1, false},
    {"l2_cv3", 32, 320, 320, 160, 160, 1, 1, false},
    {"l19_cv12", 32, 80, 80, 1280, 640, 1, 1, false},   // M 204800 N 640 K 1280 (C3.cv1|cv2 behind a concat)
    {"l23_cv12", 32, 160, 160, 640, 320, 1, 1, false},  // M 819200 N 320 K 640
    {"l20_1x1", 32, 80, 80, 640, 320, 1, 1, false},     // M 204800 N 320 K 640
    {"l8_cv3", 32, 40, 40, 960, 960, 1, 1, false},      // M 51200 N 960 K 960
    {"l15_cv12", 32, 40, 40, 1920, 960, 1, 1, false},   // M 51200 N 960 K 1920
    {"l29_cv12", 32, 40, 40, 1280, 960, 1, 1, false},   // M 51200 N 960 K 1280
    {"l4_cv3", 32, 160, 160, 320, 320, 1, 1, false},    // M 819200 N 320 K 320
    {"l10_cv3", 32, 20, 20, 1280, 1280, 1, 1, false},   // M 12800 N 1280 K 1280
    {"l8_1x1", 32, 40, 40, 480, 480, 1, 1, false},      // M 51200 N 480 K 480
    {"l1_s2", 32, 640, 640, 80, 160, 3, 2, false},      // M 3276800 N 160 K 720
    {"l3_s2", 32, 320, 320, 160, 320, 3, 2, false},
    {"l5_s2", 32, 160, 160, 320, 640, 3, 2, false},
    {"l7_s2", 32, 80, 80, 640, 960, 3, 2, false},       // M 51200 N 960 K 5760
    {"l24_s2", 32, 160, 160, 320, 320, 3, 2, false},    // M 204800 N 320 K 2880
    {"l27_s2", 32, 80, 80, 640, 640, 3, 2, false},
    {"s2a", 3, 10, 80, 96, 160, 3, 2, false},           // stride-2 row-run kernel: Wo = 40, 600 output pixels (ragged last tile, tiles across images), 64 + 32 channels
    {"s2b", 3, 20, 160, 80, 320, 3, 2, false},          // Wo = 80, 64 + 16 channels, two N tiles
    {"s2c", 1, 8, 640, 160, 160, 3, 2, false},          // Wo = 320: one output row per tile
    {"s2d", 2, 48, 320, 128, 160, 3, 2, false},         // Wo = 160, full groups only
    {"s2e", 3, 14, 80, 72, 160, 3, 2, false},           // Wo = 40, 64 + 8 channels (one chunk in the last group), ragged, tiles across images
    {"s2f", 2, 6, 640, 208, 160, 3, 2, false},          // Wo = 320, 3 x 64 + 16 channels
    {"p32", 2, 16, 64, 80, 80, 3, 1, true},             // row-patch kernels: short tail group, N = 80
    {"p40", 2, 16, 80, 160, 160, 3, 1, false},          // 64 + 64 + 32 channels
    {"p40b", 3, 8, 40, 96, 200, 3, 1, true},            // 64 + 32 channels, ragged N
    {"odd", 3, 19, 37, 96, 200, 3, 1, true},            // nothing divides anything: masks, ragged M and N
    {"q320", 3, 11, 23, 96, 320, 3, 1, true},           // 8-wave tiles (N % 320 == 0): ragged M, masks, 64 + 32 channels
    {"q160", 3, 11, 23, 160, 160, 3, 1, true},
    {"small", 2, 24, 40, 64, 96, 3, 1, true},
    {"small1", 2, 24, 40, 64, 96, 1, 1, false},
    {"smalls2", 2, 24, 40, 64, 96, 3, 2, false},
};

__global__ void ref_conv(const uint16_t* in, const uint16_t* wgt, const float* bias, const uint16_t* res,
                         uint16_t* out, int B, int H, int W, int C, int Ho, int Wo, int N, int k, int s,
                         int pad, int k_pad, int cin_pad) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * Ho * Wo * N;
    if (t >= total) return;
    const int n = (int)(t % N);
    const long long m = t / N;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((long long)Wo * Ho));
    float acc = 0.f;
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q) {
            const int iy = oy * s - pad + r, ix = ox * s - pad + q;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            const uint16_t* ip = in + ((size_t)(b * H + iy) * W + ix) * C;
            const uint16_t* wp = wgt + (size_t)n * k_pad + (r * k + q) * cin_pad;
            for (int c = 0; c < C; ++c) acc += bf16_to_f32(ip[c]) * bf16_to_f32(wp[c]);
        }
    float v = acc + bias[n];
    v = v / (1.0f + expf(-v));
    if (res) v += bf16_to_f32(res[(size_t)m * N + n]);
    out[(size_t)m * N + n] = f32_to_bf16(v);
}

// fp8 reference: e4m3 activations [pixels][C] and weights [N][9][C] (dense, k = t*C + c), fp32 accumulation,
// per-channel scale, bias, SiLU, residual, bf16 output
__global__ void ref_conv_f8(const uint8_t* in, const uint8_t* wgt, const float* scale, const float* bias,
                            const uint16_t* res, uint16_t* out, int B, int H, int W, int C, int N) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * N;
    if (t >= total) return;
    const int n = (int)(t % N);
    const long long m = t / N;
    const int ox = (int)(m % W), oy = (int)((m / W) % H), b = (int)(m / ((long long)W * H));
    float acc = 0.f;
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
            const int iy = oy - 1 + r, ix = ox - 1 + q;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            const uint8_t* ip = in + ((size_t)(b * H + iy) * W + ix) * C;
            const uint8_t* wp = wgt + ((size_t)n * 9 + (r * 3 + q)) * C;
            for (int c = 0; c < C; ++c) acc += e4m3_to_f32(ip[c]) * e4m3_to_f32(wp[c]);
        }
    float v = acc * scale[n] + bias[n];
    v = v / (1.0f + expf(-v));
    if (res) v += bf16_to_f32(res[(size_t)m * N + n]);
    out[(size_t)m * N + n] = f32_to_bf16(v);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: convbench <shape> <iters> <cfg>...   (cfg -1 = all)\n");
        return 2;
    }
    const Shape* sh = nullptr;
    for (const Shape& s : kShapes)
        if (!strcmp(s.name, argv[1])) sh = &s;
    if (!sh) { fprintf(stderr, "unknown shape %s\n", argv[1]); return 2; }
    Shape sh_b;                                                   // CONVBENCH_B=<n>: the same layer at another batch
    if (const char* eb = getenv("CONVBENCH_B")) { sh_b = *sh; sh_b.b = atoi(eb); sh = &sh_b; }
    const int iters = atoi(argv[2]);
    std::vector<int> cfgs;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "all")) { for (int j = 0; j < conv_num_cfgs(); ++j) cfgs.push_back(j); continue; }
        if (argv[i][0] == 'p') { cfgs.push_back(-1 - atoi(argv[i] + 1)); continue; }   // p0, p1: instrumented v2 variants
        if (argv[i][0] == 't') { cfgs.push_back(-301 - atoi(argv[i] + 1)); continue; } // t0..: v5 developer variants
        if (argv[i][0] == 'u') { cfgs.push_back(-401 - atoi(argv[i] + 1)); continue; } // u0..: v6 developer variants
        if (argv[i][0] == 'f') { cfgs.push_back(-801 - atoi(argv[i] + 1)); continue; } // f0..: fp8 developer variants
        if (argv[i][0] == 'n') {                                                        // n<prefix>: every configuration whose name starts with prefix
            for (int j = 0; j < conv_num_cfgs(); ++j) if (!strncmp(conv_cfg(j).name, argv[i] + 1, strlen(argv[i] + 1))) cfgs.push_back(j);
            continue;
        }
        cfgs.push_back(atoi(argv[i]));
    }
    CK(conv_init());
    const int pad = sh->k / 2;
    const int Ho = sh->h / sh->s, Wo = sh->w / sh->s;
    const long long M = (long long)sh->b * Ho * Wo;
    const int cin_pad = (sh->cin + 7) / 8 * 8;
    const int k_real = sh->k * sh->k * sh->cin;
    const int k_pad = (sh->k * sh->k * cin_pad + 63) / 64 * 64;
    const int n_rows = (sh->cout + 15) / 16 * 16;
    const size_t in_elems = (size_t)sh->b * sh->h * sh->w * sh->cin, out_elems = (size_t)M * sh->cout;

    std::mt19937 rng(123);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> h_in(in_elems), h_w((size_t)n_rows * k_pad, 0), h_res(out_elems);
    std::vector<float> h_b(n_rows, 0.f);
    for (auto& v : h_in) v = f32_to_bf16(nd(rng));
    for (auto& v : h_res) v = f32_to_bf16(nd(rng));
    const float wstd = 1.0f / sqrtf((float)k_real);
    for (int n = 0; n < sh->cout; ++n) {
        h_b[n] = 0.1f * nd(rng);
        for (int t = 0; t < sh->k * sh->k; ++t)
            for (int c = 0; c < sh->cin; ++c) h_w[(size_t)n * k_pad + t * cin_pad + c] = f32_to_bf16(wstd * nd(rng));
    }
    // second weight packing for the row-patch kernel: k = (channel group of 64, tap, channel in group)
    const int groups = (cin_pad + 63) / 64, k_pad4 = groups * 9 * 64;
    std::vector<uint16_t> h_w4;
    if (sh->k == 3) {
        h_w4.assign((size_t)n_rows * k_pad4, 0);
        for (int n = 0; n < sh->cout; ++n)
            for (int t = 0; t < 9; ++t)
                for (int c = 0; c < sh->cin; ++c)
                    h_w4[(size_t)n * k_pad4 + ((c / 64) * 9 + t) * 64 + (c % 64)] = h_w[(size_t)n * k_pad + t * cin_pad + c];
    }
    uint16_t* d_w4 = nullptr;
    // the paired packing of conv_v5.cpp for a last group of at most 32 channels (CONVBENCH_PAIR=0: without)
    uint16_t* d_w4p = nullptr;
    int k_pad4p = 0;
    std::vector<uint16_t> h_w4p;
    if (sh->k == 3 && (cin_pad % 64) != 0 && (cin_pad % 64) <= 32 && !(getenv("CONVBENCH_PAIR") && atoi(getenv("CONVBENCH_PAIR")) == 0)) {
        const int tail = cin_pad % 64;
        k_pad4p = (9 * (groups - 1) + 6) * 64;
        h_w4p.assign((size_t)n_rows * k_pad4p, 0);
        for (int n = 0; n < n_rows; ++n) {
            const uint16_t* src = &h_w4[(size_t)n * k_pad4];
            uint16_t* dst = &h_w4p[(size_t)n * k_pad4p];
            std::copy(src, src + (size_t)9 * (groups - 1) * 64, dst);
            for (int r = 0; r < 3; ++r)
                for (int sx = 0; sx < 3; ++sx)
                    for (int c = 0; c < tail; ++c)
                        dst[((groups - 1) * 9 + 2 * r + (sx == 2 ? 1 : 0)) * 64 + (sx == 1 ? 32 : 0) + c] = src[((groups - 1) * 9 + r * 3 + sx) * 64 + c];
        }
    }
    uint16_t *d_in, *d_w, *d_res, *d_out, *d_ref, *d_zero;
    float* d_b;
    CK(hipMalloc(&d_in, in_elems * 2 + 4096));
    CK(hipMalloc(&d_w, h_w.size() * 2));
    CK(hipMalloc(&d_res, out_elems * 2));
    CK(hipMalloc(&d_out, out_elems * 2));
    CK(hipMalloc(&d_ref, out_elems * 2));
    CK(hipMalloc(&d_b, n_rows * 4));
    CK(hipMalloc(&d_zero, 256));
    CK(hipMemset(d_zero, 0, 256));
    CK(hipMemcpy(d_in, h_in.data(), in_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), h_w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_res, h_res.data(), out_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), n_rows * 4, hipMemcpyHostToDevice));
    if (!h_w4.empty()) {
        CK(hipMalloc(&d_w4, h_w4.size() * 2));
        CK(hipMemcpy(d_w4, h_w4.data(), h_w4.size() * 2, hipMemcpyHostToDevice));
    }
    if (!h_w4p.empty()) {
        CK(hipMalloc(&d_w4p, h_w4p.size() * 2));
        CK(hipMemcpy(d_w4p, h_w4p.data(), h_w4p.size() * 2, hipMemcpyHostToDevice));
    }

    {
        const long long total = M * sh->cout;
        hipLaunchKernelGGL(ref_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_in, d_w, d_b,
                           sh->res ? d_res : nullptr, d_ref, sh->b, sh->h, sh->w, sh->cin, Ho, Wo, sh->cout, sh->k,
                           sh->s, pad, k_pad, cin_pad);
        CK(hipDeviceSynchronize());
    }
    // CONVBENCH_FUSED=1 (80 -> 80 channel 3x3 shapes): the whole bottleneck out = x + SiLU(W2 * SiLU(W1 x + b1) + b2) in one
    // launch (conv_v5c.cpp's fused kernels: ConvArgs::wgt_pre / bias_pre, residual = the input); the reference is the two
    // reference convs in a row with the hidden tensor rounded to 16 bits in between
    const bool fused = getenv("CONVBENCH_FUSED") && atoi(getenv("CONVBENCH_FUSED")) != 0;
    uint16_t *d_wpre = nullptr, *d_hidden = nullptr;
    float* d_bpre = nullptr;
    const int k_pad_pre = (cin_pad + 63) / 64 * 64;
    if (fused) {
        if (sh->k != 3 || sh->s != 1 || sh->cin != sh->cout) { fprintf(stderr, "CONVBENCH_FUSED: a C -> C 3x3 / s1 shape\n"); return 2; }
        std::vector<uint16_t> h_wpre((size_t)n_rows * k_pad_pre, 0);
        std::vector<float> h_bpre(n_rows, 0.f);
        const float ws1 = 1.0f / sqrtf((float)sh->cin);
        for (int n = 0; n < sh->cout; ++n) {
            h_bpre[n] = 0.1f * nd(rng);
            for (int c = 0; c < sh->cin; ++c) h_wpre[(size_t)n * k_pad_pre + c] = f32_to_bf16(ws1 * nd(rng));
        }
        CK(hipMalloc(&d_wpre, h_wpre.size() * 2));
        CK(hipMalloc(&d_bpre, n_rows * 4));
        CK(hipMalloc(&d_hidden, in_elems * 2 + 4096));
        CK(hipMemcpy(d_wpre, h_wpre.data(), h_wpre.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_bpre, h_bpre.data(), n_rows * 4, hipMemcpyHostToDevice));
        const long long total = M * sh->cout;
        hipLaunchKernelGGL(ref_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_in, d_wpre, d_bpre,
                           (const uint16_t*)nullptr, d_hidden, sh->b, sh->h, sh->w, sh->cin, sh->h, sh->w, sh->cout, 1, 1, 0, k_pad_pre, cin_pad);
        hipLaunchKernelGGL(ref_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_hidden, d_w, d_b,
                           sh->res ? d_in : nullptr, d_ref, sh->b, sh->h, sh->w, sh->cin, Ho, Wo, sh->cout, sh->k, sh->s, pad, k_pad, cin_pad);
        CK(hipDeviceSynchronize());
    }
    std::vector<uint16_t> h_ref16(out_elems), h_out(out_elems);
    CK(hipMemcpy(h_ref16.data(), d_ref, out_elems * 2, hipMemcpyDeviceToHost));

    // fp8 operands of the same layer (conv_f8.cpp): activations e4m3 [pixels][C], weights quantised per output
    // channel (scale = amax / 448), packed [n_rows][groups8 * 9 * 128], k = (channel group of 128, tap, channel)
    const bool want_f8 = sh->k == 3 && sh->s == 1 && (sh->cin % 16) == 0;
    const int groups8 = (sh->cin + 127) / 128, k_pad8 = groups8 * 9 * 128;
    uint8_t *d_in8 = nullptr, *d_w8 = nullptr, *d_w8dense = nullptr;
    float* d_scale = nullptr;
    uint16_t* d_ref8 = nullptr;
    std::vector<uint16_t> h_ref8;
    if (want_f8) {
        std::vector<uint8_t> h_in8(in_elems), h_w8((size_t)n_rows * k_pad8, 0), h_w8d((size_t)sh->cout * 9 * sh->cin);
        std::vector<float> h_scale(n_rows, 0.f);
        for (size_t i = 0; i < in_elems; ++i) h_in8[i] = f32_to_e4m3(bf16_to_f32(h_in[i]));
        for (int n = 0; n < sh->cout; ++n) {
            float amax = 0.f;
            for (int t = 0; t < 9; ++t)
                for (int c = 0; c < sh->cin; ++c) amax = fmaxf(amax, fabsf(bf16_to_f32(h_w[(size_t)n * k_pad + t * cin_pad + c])));
            const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
            h_scale[n] = sc;
            for (int t = 0; t < 9; ++t)
                for (int c = 0; c < sh->cin; ++c) {
                    const uint8_t q = f32_to_e4m3(bf16_to_f32(h_w[(size_t)n * k_pad + t * cin_pad + c]) / sc);
                    h_w8d[((size_t)n * 9 + t) * sh->cin + c] = q;
                    h_w8[(size_t)n * k_pad8 + ((c / 128) * 9 + t) * 128 + (c % 128)] = q;
                }
        }
        CK(hipMalloc(&d_in8, in_elems + 4096));
        CK(hipMalloc(&d_w8, h_w8.size()));
        CK(hipMalloc(&d_w8dense, h_w8d.size()));
        CK(hipMalloc(&d_scale, n_rows * 4));
        CK(hipMalloc(&d_ref8, out_elems * 2));
        CK(hipMemcpy(d_in8, h_in8.data(), in_elems, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w8, h_w8.data(), h_w8.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w8dense, h_w8d.data(), h_w8d.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_scale, h_scale.data(), n_rows * 4, hipMemcpyHostToDevice));
        const long long total = M * sh->cout;
        hipLaunchKernelGGL(ref_conv_f8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_in8, d_w8dense, d_scale, d_b,
                           sh->res ? d_res : nullptr, d_ref8, sh->b, sh->h, sh->w, sh->cin, sh->cout);
        CK(hipDeviceSynchronize());
        h_ref8.resize(out_elems);
        CK(hipMemcpy(h_ref8.data(), d_ref8, out_elems * 2, hipMemcpyDeviceToHost));
    }

    ConvArgs a{};
    memset(&a, 0, sizeof(a));
    a.in = d_in; a.wgt = d_w; a.bias = d_b; a.out = d_out; a.res = sh->res ? d_res : nullptr; a.zero = d_zero;
    a.ld_in = sh->cin; a.ld_out = sh->cout; a.ld_res = sh->cout;
    a.H = sh->h; a.W = sh->w; a.C8 = cin_pad / 8; a.Ho = Ho; a.Wo = Wo; a.HoWo = Ho * Wo;
    a.M = (int)M; a.N = sh->cout; a.n_rows = n_rows; a.k_pad = k_pad; a.ntaps = sh->k * sh->k; a.kw = sh->k;
    a.stride = sh->s; a.pad = pad; a.act = 1; a.out_f32 = 0;
    a.wgt4 = d_w4; a.k_pad4 = k_pad4; a.groups = groups;
    a.wgt4p = d_w4p; a.k_pad4p = k_pad4p;
    if (fused) {
        a.wgt_pre = d_wpre; a.bias_pre = d_bpre; a.k_pad_pre = k_pad_pre;
        a.res = sh->res ? d_in : nullptr; a.ld_res = sh->cin;
    }
    const double flops = 2.0 * (double)M * sh->cout * (k_real + (fused ? sh->cin : 0));
    const size_t dbg_words = 8 * 16 * 4096;
    unsigned long long* d_dbg;
    CK(hipMalloc(&d_dbg, dbg_words * 8));
    CK(hipMemset(d_dbg, 0, dbg_words * 8));
    a.dbg = d_dbg;
    a.dev_param = getenv("MDHIP_DEV_PARAM") ? atoi(getenv("MDHIP_DEV_PARAM")) : 0;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%s: M=%lld N=%d K=%d (%.1f GFLOP)\n", sh->name, M, sh->cout, k_real, flops / 1e9);
    const ConvArgs a16 = a;
    ConvArgs af8 = a;
    af8.in = (const uint16_t*)d_in8; af8.in_f8 = 1; af8.wgt8 = d_w8; af8.scale = d_scale; af8.k_pad8 = k_pad8; af8.groups8 = groups8;
    af8.C8 = (sh->cin + 15) / 16; af8.wgt4 = nullptr; af8.wgt4p = nullptr;
    for (int cfg : cfgs) {
        const bool is_f8 = cfg <= -801 || (cfg >= 0 && !strncmp(conv_cfg(cfg).name, "f8:", 3));
        if (is_f8 && !want_f8) { printf("  cfg %2d: no fp8 form of this shape\n", cfg); continue; }
        a = is_f8 ? af8 : a16;
        const std::vector<uint16_t>& h_ref = is_f8 ? h_ref8 : h_ref16;
        CK(hipMemset(d_out, 0xff, out_elems * 2));
        hipError_t e = conv_launch(cfg, a, 0);
        if (e != hipSuccess) { printf("  cfg %2d launch failed: %s\n", cfg, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_out.data(), d_out, out_elems * 2, hipMemcpyDeviceToHost));
        double max_err = 0, max_ref = 0;
        size_t bad = 0;
        for (size_t i = 0; i < out_elems; ++i) {
            const float r = bf16_to_f32(h_ref[i]), o = bf16_to_f32(h_out[i]);
            const double d = fabs((double)r - (double)o);
            if (!(d <= 0.02 * fabs(r) + 0.02)) ++bad;
            if (d > max_err || d != d) max_err = d;
            if (fabs(r) > max_ref) max_ref = fabs(r);
        }
        float best = 1e30f, tot = 0;
        const int reps = 3;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) (void)conv_launch(cfg, a, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters;
            tot += ms;
            if (ms < best) best = ms;
        }
        printf("  cfg %2d %-22s %8.4f ms (best %8.4f)  %7.1f TF/s   max|err| %.3g (max|ref| %.3g) bad %zu%s\n", cfg,
               cfg <= -801 ? conv8_cfg(conv8_num_cfgs() - 801 - cfg).name : cfg <= -401 ? conv6_cfg(conv6_num_cfgs() - 401 - cfg).name : cfg <= -301 ? conv5_cfg(conv5_num_cfgs() - 301 - cfg).name : cfg < 0 ? conv2_cfg(conv2_num_cfgs() - 1 - cfg).name : conv_cfg(cfg).name, tot / reps, best, flops / (tot / reps * 1e-3) / 1e12, max_err, max_ref, bad,
               bad ? "  <-- MISMATCH" : "");
        if (cfg >= 0 && !strncmp(conv_cfg(cfg).name, "dev:strip", 9)) {
            // the stamped four-row fused bottleneck: cycles per tile and wave by phase
            std::vector<unsigned long long> h(dbg_words);
            CK(hipMemcpy(h.data(), d_dbg, dbg_words * 8, hipMemcpyDeviceToHost));
            double sum[6] = {0, 0, 0, 0, 0, 0}, tiles = 0;
            int waves = 0;
            for (size_t w = 0; w < dbg_words / 8; ++w) {
                if (!h[w * 8 + 6]) continue;
                ++waves;
                tiles += (double)h[w * 8 + 6];
                for (int k = 0; k < 6; ++k) sum[k] += (double)h[w * 8 + k];
            }
            static const char* nm[6] = {"pass 1 (64-channel group)", "pass 2 (16-channel group)", "epilogue + weight reload", "wait (DMA) + barrier",
                                        "conversion of the next T rows", "barrier + DMA issue, unit start-up"};
            double tot = 0;
            for (int k = 0; k < 6; ++k) tot += sum[k];
            {   // which SIMD each wave index of a workgroup sits on (HW_ID[5:4]); and the phase cycles by wave index
                int hist[10][4] = {};
                for (size_t w = 0; w < dbg_words / 8; ++w)
                    if (h[w * 8 + 7]) ++hist[w % 10][(h[w * 8 + 7] >> 4) & 3];
                printf("    SIMD of wave index 0..9 (workgroups per SIMD id 0/1/2/3):");
                for (int k = 0; k < 10; ++k) printf("  %d:%d/%d/%d/%d", k, hist[k][0], hist[k][1], hist[k][2], hist[k][3]);
                printf("\n");
                for (int k = 0; k < 10; ++k) {
                    double s6[6] = {0, 0, 0, 0, 0, 0}, tl = 0;
                    for (size_t w = k; w < dbg_words / 8; w += 10) {
                        if (!h[w * 8 + 6]) continue;
                        tl += (double)h[w * 8 + 6];
                        for (int q = 0; q < 6; ++q) s6[q] += (double)h[w * 8 + q];
                    }
                    if (tl > 0) printf("      wave %d: %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f\n", k, s6[0] / tl, s6[1] / tl, s6[2] / tl, s6[3] / tl, s6[4] / tl, s6[5] / tl);
                }
            }
            printf("    %d waves, %.1f tiles each; cycles per tile: total %.0f\n", waves, tiles / std::max(1, waves), tot / std::max(1.0, tiles));
            for (int k = 0; k < 6; ++k) printf("      %-36s %8.1f  (%4.1f%%)\n", nm[k], sum[k] / std::max(1.0, tiles), 100 * sum[k] / std::max(1.0, tot));
            CK(hipMemset(d_dbg, 0, dbg_words * 8));
        }
        if (cfg < 0) {
            // per-wave phase sums of the last launch: cycles per step, averaged over all waves that ran
            std::vector<unsigned long long> h(dbg_words);
            CK(hipMemcpy(h.data(), d_dbg, dbg_words * 8, hipMemcpyDeviceToHost));
            double sum[6] = {0, 0, 0, 0, 0, 0}, steps = 0;
            int waves = 0;
            {   // HW_ID census of the variants that record it (word 7): wave slot / SIMD / CU / SE of every wave
                int slot_hist[16] = {0}, n = 0, mixed = 0;
                for (size_t w = 0; w + 4 <= dbg_words / 8; w += 4) {
                    if (!h[w * 8 + 7]) continue;
                    for (int k = 0; k < 4; ++k) { ++slot_hist[h[(w + k) * 8 + 7] & 15]; ++n; }
                    if ((h[w * 8 + 7] & 1) != (h[(w + 3) * 8 + 7] & 1)) ++mixed;
                }
                if (n) {
                    printf("    HW_ID wave-slot histogram over %d waves:", n);
                    for (int k = 0; k < 16; ++k) if (slot_hist[k]) printf(" [%d]=%d", k, slot_hist[k]);
                    printf("  (workgroups whose waves disagree on slot parity: %d)\n", mixed);
                }
            }
            for (size_t w = 0; w < dbg_words / 8; ++w) {
                if (!h[w * 8 + 6]) continue;
                ++waves;
                steps += (double)h[w * 8 + 6];
                for (int k = 0; k < 6; ++k) sum[k] += (double)h[w * 8 + k];
            }
            static const char* nm5[6] = {"half1 (reads Y + mfma X)", "waitcnt vmcnt/lgkmcnt", "barrier", "half2 (dma + reads X + mfma Y)",
                                         "first step after an epilogue: wait + barrier", "epilogue (amortised)"};
            static const char* nm_epi[6] = {"epilogue pixel row 0", "epilogue pixel row 1", "epilogue pixel row 2", "epilogue pixel row 3",
                                            "epilogue pixel row 4", "everything else (main loop, waits, barriers)"};
            const char* cname = cfg <= -301 && cfg > -401 ? conv5_cfg(conv5_num_cfgs() - 301 - cfg).name : "";
            const int prof_bits = strrchr(cname, '/') ? atoi(strrchr(cname, '/') + 1) : 0;
            const char* const* nm = (prof_bits & 128) ? nm_epi : nm5;
            double tot = 0;
            for (int k = 0; k < 6; ++k) tot += sum[k];
            printf("    %d waves, %.0f steps each; cycles per step: total %.0f\n", waves, steps / waves, tot / steps);
            for (int k = 0; k < 6; ++k) printf("      %-34s %8.1f  (%4.1f%%)\n", nm[k], sum[k] / steps, 100 * sum[k] / tot);
            if (getenv("CONVBENCH_PER_WAVE")) {
                // the same per wave index inside its workgroup (8-wave tiles: who waits at the barrier, who arrives last)
                const int nwv = 8;
                for (int wv = 0; wv < nwv; ++wv) {
                    double s6[6] = {0, 0, 0, 0, 0, 0}, st = 0;
                    for (size_t w = wv; w < dbg_words / 8; w += nwv) {
                        if (!h[w * 8 + 6]) continue;
                        st += (double)h[w * 8 + 6];
                        for (int k = 0; k < 6; ++k) s6[k] += (double)h[w * 8 + k];
                    }
                    if (st > 0) printf("      wave %d: %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f\n", wv, s6[0] / st, s6[1] / st, s6[2] / st, s6[3] / st, s6[4] / st, s6[5] / st);
                }
            }
            CK(hipMemset(d_dbg, 0, dbg_words * 8));
        }
        fflush(stdout);
    }
    return 0;
}
