// Internal declarations shared by the kernel translation units and the C-ABI layer.
// gfx950 (MI355X / CDNA4) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdhip {

// ---------------------------------------------------------------------------------------
// storage-type helpers.  Activations and packed weights are 16-bit: bf16 (the configuration BASELINE.json
// names) or fp16 (same MFMA rate, 3 more mantissa bits; MDHIP_DTYPE_FP16).  Accumulation is fp32 either way.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // RNE
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    union { float f; uint32_t u; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}
__host__ __device__ inline uint16_t f32_to_f16(float f) {
    union { _Float16 h; uint16_t u; } v;
    v.h = (_Float16)f;                                                          // RNE, overflow -> inf
    return v.u;
}
__host__ __device__ inline float f16_to_f32(uint16_t h) {
    union { _Float16 h; uint16_t u; } v;
    v.u = h;
    return (float)v.h;
}
// OCP e4m3 (fn): round to nearest even, saturating at +-448 (the hardware conversion used by the kernels is fed
// clamped values, so both agree); NaN -> 0x7f
__host__ __device__ inline uint8_t f32_to_e4m3(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    const uint32_t sign = (v.u >> 24) & 0x80u;
    uint32_t a = v.u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint8_t)(sign | 0x7f);
    if (a >= 0x43e00000u) return (uint8_t)(sign | 0x7e);            // >= 448 (incl. inf): saturate
    if (a < 0x3a800000u) {                                          // < 2^-10: below half of the smallest subnormal 2^-9
        return (uint8_t)sign;                                       // (2^-10 itself is a tie to even = 0)
    }
    const int e = (int)(a >> 23) - 127;                             // unbiased exponent, -10 .. 8
    if (e < -6) {                                                   // subnormal range: quantum 2^-9
        v.u = a;
        const float q = v.f * 512.0f;                               // exact
        // round to nearest even integer in [0, 8]
        const float r = q + 12582912.0f;                            // 1.5 * 2^23: the add rounds to an integer, RNE
        union { float f; uint32_t u; } w;
        w.f = r;
        const uint32_t m = w.u & 0xfu;                              // 0 .. 8 (8 = the smallest normal 2^-6)
        return (uint8_t)(sign | m);
    }
    uint32_t mant = a & 0x7fffffu;
    uint32_t keep = mant >> 20, rest = mant & 0xfffffu;
    uint32_t out = ((uint32_t)(e + 7) << 3) | keep;
    if (rest > 0x80000u || (rest == 0x80000u && (keep & 1u))) ++out;   // RNE; a carry moves into the exponent
    if (out > 0x7eu) out = 0x7eu;
    return (uint8_t)(sign | out);
}
__host__ __device__ inline float e4m3_to_f32(uint8_t h) {
    const uint32_t s = h >> 7, e = (h >> 3) & 15u, m = h & 7u;
    union { float f; uint32_t u; } v;
    if (e == 15u && m == 7u) { v.u = 0x7fc00000u; return v.f; }
    float f;
    if (e == 0u) f = (float)m * (1.0f / 512.0f);
    else { v.u = ((e + 120u) << 23) | (m << 20); f = v.f; }
    return s ? -f : f;
}
__host__ __device__ inline uint16_t f32_to_st(float f, int f16) { return f16 ? f32_to_f16(f) : f32_to_bf16(f); }
__host__ __device__ inline float st_to_f32(uint16_t h, int f16) { return f16 ? f16_to_f32(h) : bf16_to_f32(h); }

// The convolution translation units are compiled twice: as is (bf16, namespace mdhip::st_bf16) and with
// -DMDHIP_ST_F16 (fp16, namespace mdhip::st_f16).  Inside them MDHIP_ST is that namespace, frag8_t the MFMA
// operand type, MDHIP_MFMA the 16x16x32 instruction, st_pack2 / st_unpack the epilogue conversions.
#if defined(MDHIP_ST_F16)
#define MDHIP_ST st_f16
#define MDHIP_ST_LABEL "fp16"
#else
#define MDHIP_ST st_bf16
#define MDHIP_ST_LABEL "bf16"
#endif
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#if defined(MDHIP_ST_F16)
typedef _Float16 frag8_t __attribute__((ext_vector_type(8)));
#define MDHIP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ uint32_t st_pack2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    const h2 r = __builtin_convertvector(v, h2);
    return *(const uint32_t*)&r;
}
__device__ __forceinline__ float st_unpack(uint16_t h) { return f16_to_f32(h); }
#else
typedef short frag8_t __attribute__((ext_vector_type(8)));
#define MDHIP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ uint32_t st_pack2(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    const b2 r = __builtin_convertvector(v, b2);
    return *(const uint32_t*)&r;
}
__device__ __forceinline__ float st_unpack(uint16_t h) { return bf16_to_f32(h); }
#endif
// four fp32 values -> four e4m3 bytes (RNE, saturating at +-448): the out_f8 epilogue of the 16-bit kernels
__device__ __forceinline__ uint32_t pack_e4m3x4(float a, float b, float c, float d, float qscale) {
    a = __builtin_fminf(__builtin_fmaxf(a * qscale, -448.0f), 448.0f);
    b = __builtin_fminf(__builtin_fmaxf(b * qscale, -448.0f), 448.0f);
    c = __builtin_fminf(__builtin_fmaxf(c * qscale, -448.0f), 448.0f);
    d = __builtin_fminf(__builtin_fmaxf(d * qscale, -448.0f), 448.0f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
// a fragment that is not read from LDS (ablation variants of the kernels only)
__device__ __forceinline__ frag8_t frag_dummy(int v) {
    frag8_t z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (__typeof__(z[0]))(v + k);
    asm volatile("" : "+v"(z));
    return z;
}
#endif

// ---------------------------------------------------------------------------------------
// implicit-GEMM convolution (conv_igemm.cpp)
// ---------------------------------------------------------------------------------------
struct ConvArgs {
    const uint16_t* in;     // NHWC bf16 view: channel 0 of this conv's input, pixel stride ld_in
    const uint16_t* wgt;    // packed [n_rows][k_pad] bf16, k = (r*kw + s)*C_in_pad + c
    const float*    bias;   // [n_rows] fp32 (zero padded)
    void*           out;    // bf16 (or fp32 when out_f32) view, pixel stride ld_out
    const uint16_t* res;    // residual view (added after the activation) or nullptr
    const uint16_t* zero;   // >= 16 bytes of zeros: source of every padded / out-of-range chunk
    int ld_in, ld_out, ld_res;
    int H, W, C8;           // input spatial size, C_in_pad / 8
    int Ho, Wo, HoWo;
    int M, N, n_rows;       // GEMM rows (= batch*Ho*Wo), real out channels, packed rows (mult of 16)
    int k_pad;              // multiple of 64
    int ntaps, kw;
    int stride, pad;
    int act;                // 1 = SiLU
    int out_f32;            // 1 = fp32 output, no rounding (Detect logits)
    int tiles_n, tiles_m, tiles_per_xcd, m_streams;   // filled by conv_launch
    // second packing (introduced by the row-patch kernel of round 1; read by conv_v5 / conv_v5s / conv_v5c / conv_v7 / conv_f8): weights packed [n_rows][groups*9*64], k = (cg, r, s, c % 64)
    const uint16_t* wgt4;   // nullptr when the op has no such packing
    int k_pad4, groups;
    // conv_v5.cpp, layers whose last channel group is at most half full (C_in mod 64 in 8 .. 32: the 160- and 480-channel
    // bottlenecks): the same packing with the last group's taps PAIRED -- per kernel row r one slab [ tap (r,0) ch 0..31 |
    // tap (r,1) ch 0..31 ] and one slab [ tap (r,2) ch 0..31 | zeros ], 6 slabs instead of 9 -- so that the group takes two
    // steps per kernel row instead of three half-empty ones (same MFMA chain per accumulator: same bits).  nullptr = none.
    const uint16_t* wgt4p;
    int k_pad4p;
    // fp8 path (conv_f8.cpp; MDHIP_DTYPE_FP8): the input view holds e4m3 bytes (ld_in, C8 then count BYTES and
    // 16-byte chunks = 16 channels), weights packed [n_rows][groups8*9*128] e4m3, k = (channel group of 128, tap,
    // channel in group), `scale` = per-output-channel fp32 factor (activation scale x weight scale) applied to the
    // fp32 accumulator before the bias; out_f8: the 16-bit kernels' epilogue writes e4m3(v * out_qscale) bytes
    const uint8_t* wgt8;
    const float*   scale;
    int k_pad8, groups8;
    int in_f8, out_f8;
    float out_qscale;
    void* dbg;              // instrumentation output of the profiling variants (tools/convbench.cpp), else nullptr
    int dev_param;          // free parameter of the developer variants (tools/convbench.cpp: env MDHIP_DEV_PARAM)
    // conv_v5c.cpp, fused bottleneck: the 1x1 conv in front of this 3x3 ([n_rows][k_pad_pre] 16-bit weights, k = input
    // channel; fp32 bias); nullptr = plain 3x3
    const uint16_t* wgt_pre;
    const float*    bias_pre;
    int k_pad_pre;
    // conv_v2.cpp, 1x1 convs behind Upsample + Concat: the first up_slabs 64-channel slabs of K are read from the
    // LOW-resolution tensor in_up (pixel (y, x) -> (y / 2, x / 2), pixel pitch ld_up) instead of their 4x copy in the
    // concat buffer; nullptr = everything from `in`
    const uint16_t* in_up;
    int ld_up, up_slabs;
    // Detect decode in the epilogue of the Detect 1x1 conv (out_f32 ops of a head with 8 outputs per anchor; conv_igemm.cpp /
    // conv_v2.cpp): dec_pred != nullptr -> instead of storing its four fp32 logits a lane decodes them (mdhip_decode_store
    // below: detect_decode_kernel's statements, the same bits) and writes 16 bytes of the prediction row
    // dec_pred[image][dec_level_off + (anchor * Ho + y) * Wo + x][8]; the logits tensor is then neither written nor read
    float* dec_pred;
    const float* dec_anchors;   // device, [na][2]: anchor sizes of this level in pixels
    float dec_stride;
    int dec_level_off, dec_n_anchors;
    // floor(2^32 / HoWo), floor(2^32 / Wo) (0xffffffff for a divisor of 1): the tile set-up of the implicit-GEMM kernels
    // splits an output pixel index into (image, row, column) with conv_udiv() instead of two run-time integer
    // divisions per row; filled by conv_launch / conv2_launch (conv_set_rcp), callers leave them alone
    unsigned rcp_howo, rcp_wo;
};
inline unsigned conv_rcp32(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }
inline void conv_set_rcp(ConvArgs& p) { p.rcp_howo = conv_rcp32(p.HoWo); p.rcp_wo = conv_rcp32(p.Wo); }
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// m / d for 0 <= m < 2^31, d >= 1, rcp = floor(2^32 / d): the estimate mulhi(m, rcp) is the quotient or one below it
// (m * (2^32 / d - rcp) / 2^32 < 1), one compare fixes it -- 5 instructions instead of the ~30 of a run-time division
__device__ __forceinline__ int conv_udiv(int m, int d, unsigned rcp) {
    unsigned q = __umulhi((unsigned)m, rcp);
    const unsigned r = (unsigned)m - q * (unsigned)d;
    return (int)(r >= (unsigned)d ? q + 1u : q);
}
// SiLU of two values at a time: the same operations and roundings as the scalar  x * rcp(1 + exp(-x))  of every kernel
// family (x * -log2(e), v_exp_f32, 1 + e, v_rcp_f32, x * r), with the four full-rate ones as packed fp32 instructions.
// The two transcendentals are quarter rate: for 1x1 convs with K <= 640 the activation is more VALU time than the
// layer's MFMAs are matrix time, so the epilogues are written around it.  (neg_log2e: -0x1.715476p+0f, passed in so that
// register-tight kernels can keep it out of the main loop.)
typedef __attribute__((ext_vector_type(2))) float mdhip_f32x2;
constexpr float kNegLog2e = -0x1.715476p+0f;
__device__ __forceinline__ mdhip_f32x2 silu_f32x2(mdhip_f32x2 x, float neg_log2e = kNegLog2e) {
    const mdhip_f32x2 u = x * mdhip_f32x2{neg_log2e, neg_log2e};
    const mdhip_f32x2 e = mdhip_f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + mdhip_f32x2{1.0f, 1.0f};
    return x * mdhip_f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}
// v[0..3] = a[0..3] + b[0..3] (packed), the first half of every epilogue; the activation follows under one uniform branch
// per pixel row (mdhip_silu4), not as a select per value
template <typename A, typename B>
__device__ __forceinline__ void mdhip_bias4(const A& a, const B& b, float (&v)[4]) {
    const mdhip_f32x2 t0 = mdhip_f32x2{a[0], a[1]} + mdhip_f32x2{b[0], b[1]};
    const mdhip_f32x2 t1 = mdhip_f32x2{a[2], a[3]} + mdhip_f32x2{b[2], b[3]};
    v[0] = t0[0]; v[1] = t0[1]; v[2] = t1[0]; v[3] = t1[1];
}
// Detect decode (yolov5 Detect.forward, inference): ONE definition for detect_decode_kernel
// (misc_kernels.cpp) and for the conv epilogues that decode in place -- same statements, no contraction: same bits
__device__ __forceinline__ float mdhip_sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void mdhip_decode_box(float l0, float l1, float l2, float l3, int x, int y, float stride, float aw, float ah,
                                                 float& cx, float& cy, float& bw, float& bh) {
#pragma clang fp contract(off)
    const float s0 = mdhip_sigmoid_exact(l0), s1 = mdhip_sigmoid_exact(l1);
    const float s2 = mdhip_sigmoid_exact(l2), s3 = mdhip_sigmoid_exact(l3);
    cx = (s0 * 2.0f + ((float)x - 0.5f)) * stride;
    cy = (s1 * 2.0f + ((float)y - 0.5f)) * stride;
    const float w2 = s2 * 2.0f, h2 = s3 * 2.0f;
    bw = (w2 * w2) * aw;
    bh = (h2 * h2) * ah;
}
// a lane's four consecutive logits (channels n .. n + 3 of output pixel m, n a multiple of 4, 8 outputs per anchor) -> its
// half of the prediction row of anchor n / 8: the box (n mod 8 == 0) or objectness + classes (n mod 8 == 4)
__device__ __forceinline__ void mdhip_decode_store(const ConvArgs& p, int m, int n, const float (&v)[4]) {
#pragma clang fp contract(off)
    const int b = conv_udiv(m, p.HoWo, p.rcp_howo);
    const int rem = m - b * p.HoWo;
    const int y = conv_udiv(rem, p.Wo, p.rcp_wo);
    const int x = rem - y * p.Wo;
    const int a = n >> 3;
    const int idx = p.dec_level_off + (a * p.Ho + y) * p.Wo + x;
    float* o = p.dec_pred + ((size_t)b * p.dec_n_anchors + idx) * 8 + (n & 4);
    float4 r;
    if ((n & 4) == 0) mdhip_decode_box(v[0], v[1], v[2], v[3], x, y, p.dec_stride, p.dec_anchors[a * 2 + 0], p.dec_anchors[a * 2 + 1], r.x, r.y, r.z, r.w);
    else r = make_float4(mdhip_sigmoid_exact(v[0]), mdhip_sigmoid_exact(v[1]), mdhip_sigmoid_exact(v[2]), mdhip_sigmoid_exact(v[3]));
    *(float4*)o = r;
}
__device__ __forceinline__ void mdhip_silu4(float (&v)[4], float neg_log2e = kNegLog2e) {
    const mdhip_f32x2 t0 = silu_f32x2(mdhip_f32x2{v[0], v[1]}, neg_log2e), t1 = silu_f32x2(mdhip_f32x2{v[2], v[3]}, neg_log2e);
    v[0] = t0[0]; v[1] = t0[1]; v[2] = t1[0]; v[3] = t1[1];
}
#endif

struct ConvCfg {
    int bm, bn, threads;
    size_t lds_bytes;
    int blocks_per_cu;      // residency the kernel is compiled for (LDS- and wave-limited)
    const char* name;
};

// The conv API exists once per storage type (see MDHIP_ST above):
//   conv_*  : dispatch over all kernels (conv_igemm.cpp); ids [0, conv_num_v1_cfgs()) run conv_igemm.cpp's
//             kernel (every shape), then conv_v2.cpp's, conv_v5.cpp's, conv_v6.cpp's
//   conv2_* : second-generation main loop (conv_v2.cpp); local ids, reached through conv_launch
//   conv5_* : 3x3 / stride 1 with row-segment reuse across the taps of a kernel row (conv_v5.cpp); its configurations
//             for small launches (conv5s_*, conv_v5s.cpp) are listed behind its own
//   conv6_* : the stem (3x3 over 16-channel space-to-depth pixels, N = 80) with its weights in registers (conv_v6.cpp)
//   conv8_* : conv_v5's structure on e4m3 operands with the block-scaled K = 128 MFMA (conv_f8.cpp)
//   conv7_* : 3x3 / STRIDE 2 with row-run reuse (odd / even input columns in two sub-buffers; conv_v7.cpp); listed last
// conv_launch returns hipSuccess or the launch error; conv_init raises the dynamic-LDS limits (one-off);
// conv_cfg_is_bitwise_family is false for kernels whose result equals the others' up to fp32 summation
// order only.
#define MDHIP_CONV_API \
    int conv_num_cfgs(); \
    const ConvCfg& conv_cfg(int i); \
    hipError_t conv_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv_init(); \
    bool conv_supports(int cfg, const ConvArgs& a); \
    bool conv_cfg_is_bitwise_family(int cfg); \
    bool conv_cfg_decodes(int cfg); \
    bool conv2_cfg_decodes(int cfg); \
    int conv_num_v1_cfgs(); \
    int conv2_num_cfgs(); \
    const ConvCfg& conv2_cfg(int i); \
    bool conv2_supports(const ConvArgs& a); \
    bool conv2_cfg_is_ring(int cfg); \
    bool conv2_is_pointwise(const ConvArgs& a); \
    hipError_t conv2_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv2_init(); \
    int conv5_num_cfgs(); \
    const ConvCfg& conv5_cfg(int i); \
    bool conv5_supports(int cfg, const ConvArgs& a); \
    hipError_t conv5_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv5_init(); \
    int conv5s_num_cfgs(); \
    const ConvCfg& conv5s_cfg(int i); \
    bool conv5s_supports(int cfg, const ConvArgs& a); \
    hipError_t conv5s_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv5s_init(); \
    int conv5c_num_cfgs(); \
    const ConvCfg& conv5c_cfg(int i); \
    bool conv5c_supports(int cfg, const ConvArgs& a); \
    hipError_t conv5c_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv5c_init(); \
    int conv6_num_cfgs(); \
    const ConvCfg& conv6_cfg(int i); \
    bool conv6_supports(int cfg, const ConvArgs& a); \
    hipError_t conv6_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv6_init(); \
    int conv8_num_cfgs(); \
    const ConvCfg& conv8_cfg(int i); \
    bool conv8_supports(int cfg, const ConvArgs& a); \
    hipError_t conv8_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv8_init(); \
    int conv7_num_cfgs(); \
    const ConvCfg& conv7_cfg(int i); \
    bool conv7_supports(int cfg, const ConvArgs& a); \
    hipError_t conv7_launch(int cfg, const ConvArgs& a, hipStream_t s); \
    hipError_t conv7_init();
namespace st_bf16 {
MDHIP_CONV_API
}
namespace st_f16 {
MDHIP_CONV_API
}
#undef MDHIP_CONV_API

// ---------------------------------------------------------------------------------------
// memory-bound helpers (misc_kernels.cpp)
// ---------------------------------------------------------------------------------------
struct LetterboxDev {     // device copy of mdhip_letterbox + source pointer
    const uint8_t* src;
    int src_h, src_w, resized_h, resized_w, top, left;
    int interp;           // 0: cv2.INTER_LINEAR, 1: cv2.INTER_AREA (shrinking only)
    // cv2's bilinear scales  1.0 / ((double)resized / (double)src)  per axis, computed on the host with that very expression (IEEE
    // double on both sides: the bits the kernels' own linear_scale() produces); read by letterbox_linear_s2d_kernel
    double sx, sy;
};
// u8 HWC -> space-to-depth bf16 [n][out_h/2][out_w/2][16] (12 real channels: (dy,dx,c)), /255
// Three kernels, chosen per batch from the geometry: a streaming copy (no image resampled), a streaming bilinear kernel
// (cv2.INTER_LINEAR, every real camera image) and the general one (INTER_AREA, very wide sources); force_general = the last
// one whatever the batch (tests, A/B).  geom_host: the geometry in host memory (source pointers are device pointers); when
// letterbox_geometry_travels_inline(...) it is passed in the kernel arguments and geom_dev is not read (the caller skips
// the upload)
bool letterbox_geometry_travels_inline(const LetterboxDev* geom_host, int n, int out_w, bool force_general);
hipError_t launch_letterbox_s2d(const LetterboxDev* geom_dev, const LetterboxDev* geom_host, int n, int out_h, int out_w,
                                uint16_t* out, int f16, bool force_general, hipStream_t s);
// SPPF: three chained 5x5/s1/p2 max pools of slice 0 written to slices 1..3 of the same buffer
hipError_t launch_sppf_pool(uint16_t* buf, int ld, int c, int n, int h, int w, int k, int f16, hipStream_t s);
// nearest x2 upsample of a view into a view
hipError_t launch_upsample2x(const uint16_t* in, int ld_in, uint16_t* out, int ld_out, int c,
                             int n, int h, int w, hipStream_t s);
// strided channel-slice copy
hipError_t launch_copy_view(const uint16_t* in, int ld_in, uint16_t* out, int ld_out, int c,
                            long long pixels, hipStream_t s);
// max |x| of a 16-bit view, atomically max-ed into *out (a non-negative float, zero it first)
hipError_t launch_absmax_view(const uint16_t* in, int ld, int c, long long pixels, int f16, float* out, hipStream_t s);
// Detect decode of one level: logits fp32 [n*ny*nx][ld] -> pred[n][n_anchors][no].  A test-time-augmentation
// pass keeps anchors [keep_from, keep_to) of its own numbering, de-scales / un-flips the boxes and writes them
// at out_off of the concatenated prediction; the default is the plain forward.
struct DecodeTta {
    int keep_from = 0, keep_to = 0x7fffffff, out_off = 0;
    float scale = 1.0f;
    int flip_lr = 0;
    float img_w = 0.f;
};
hipError_t launch_detect_decode(const float* logits, int ld, float* pred, int n, int ny, int nx,
                                int na, int no, int n_anchors, int level_off, float stride,
                                const float* anchors_px /*device, [na][2]*/, const DecodeTta& tta, hipStream_t s);
// scale_img of yolov5's augmented inference: s2d input (h x w) -> s2d (oh x ow), bilinear to (sh x sw), pad 0.447
hipError_t launch_tta_scale(const uint16_t* in, uint16_t* out, int n, int h, int w, int sh, int sw, int oh, int ow,
                            int flip_lr, int f16, hipStream_t s);
// debug readback: NHWC bf16 view -> NCHW fp32
hipError_t launch_nhwc_to_nchw_f32(const uint16_t* in, int ld, float* out, int n, int c, int h,
                                   int w, int f16, hipStream_t s);
hipError_t launch_s2d_to_nchw_f32(const uint16_t* in, float* out, int n, int h, int w, int f16, hipStream_t s);

// ---------------------------------------------------------------------------------------
// NMS (nms_kernels.cpp)
// ---------------------------------------------------------------------------------------
constexpr int kNmsScanParts = 16;   // workgroups per image of the candidate scan (nms_kernels.cpp stage A)
struct NmsScratch {
    uint32_t* keys[3];     // [n][cap] each: 0 / 2 = the band being sorted (ping-pong), 1 = every candidate in anchor order
    uint32_t* vals[3];
    int cap;               // candidates capacity per image (= max anchors)
    uint32_t* seg_cnt;     // [n][kNmsScanParts] candidates found by each scan workgroup
};
hipError_t launch_nms(const float* pred, int n, int n_anchors, int no, float conf_thres,
                      float iou_thres, int max_det, const NmsScratch& scr, float* out /*device [n][max_det][6]*/,
                      int* counts /*device [n]*/, hipStream_t s);
constexpr int kNmsMaxDet = 1024;

}  // namespace mdhip
