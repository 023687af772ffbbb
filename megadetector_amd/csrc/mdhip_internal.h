// Internal declarations shared by the kernel translation units and the C-ABI layer.
// gfx950 (MI355X / CDNA4) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdhip {

// ---------------------------------------------------------------------------------------
// bf16 helpers (storage type of every activation and packed weight)
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
    // row-patch direct convolution (conv_v4.cpp): weights packed [n_rows][groups*9*64], k = (cg, r, s, c % 64)
    const uint16_t* wgt4;   // nullptr when the op has no such packing
    int k_pad4, groups;
    void* dbg;              // instrumentation output of the profiling variants (tools/convbench.cpp), else nullptr
};

struct ConvCfg {
    int bm, bn, threads;
    size_t lds_bytes;
    int blocks_per_cu;      // residency the kernel is compiled for (LDS- and wave-limited)
    const char* name;
};

int conv_num_cfgs();
const ConvCfg& conv_cfg(int i);
// returns hipSuccess or the launch error
hipError_t conv_launch(int cfg, const ConvArgs& a, hipStream_t s);
hipError_t conv_init();   // one-off: raise dynamic-LDS limits
bool conv_supports(int cfg, const ConvArgs& a);
bool conv_cfg_is_bitwise_family(int cfg);   // false: same result up to fp32 summation order only   // can configuration `cfg` run this op?
int conv_num_v1_cfgs();   // ids below this run conv_igemm.cpp's kernel (every shape); the rest conv_v2.cpp's
// second-generation main loop (conv_v2.cpp); local ids, reached through conv_launch
int conv2_num_cfgs();
const ConvCfg& conv2_cfg(int i);
bool conv2_supports(const ConvArgs& a);
hipError_t conv2_launch(int cfg, const ConvArgs& a, hipStream_t s);
hipError_t conv2_init();
// third-generation main loop (conv_v3.cpp): 32-deep slabs, 4-stage ring, counted waits
int conv3_num_cfgs();
const ConvCfg& conv3_cfg(int i);
bool conv3_supports(const ConvArgs& a);
hipError_t conv3_launch(int cfg, const ConvArgs& a, hipStream_t s);
hipError_t conv3_init();
// row-patch direct convolution for 3x3 / stride 1 (conv_v4.cpp)
int conv4_num_cfgs();
const ConvCfg& conv4_cfg(int i);
bool conv4_supports(int cfg, const ConvArgs& a);
hipError_t conv4_launch(int cfg, const ConvArgs& a, hipStream_t s);
hipError_t conv4_init();
// 3x3 / stride 1 with row-segment reuse across the taps of a kernel row (conv_v5.cpp)
int conv5_num_cfgs();
const ConvCfg& conv5_cfg(int i);
bool conv5_supports(int cfg, const ConvArgs& a);
hipError_t conv5_launch(int cfg, const ConvArgs& a, hipStream_t s);
hipError_t conv5_init();

// ---------------------------------------------------------------------------------------
// memory-bound helpers (misc_kernels.cpp)
// ---------------------------------------------------------------------------------------
struct LetterboxDev {     // device copy of mdhip_letterbox + source pointer
    const uint8_t* src;
    int src_h, src_w, resized_h, resized_w, top, left;
};
// u8 HWC -> space-to-depth bf16 [n][out_h/2][out_w/2][16] (12 real channels: (dy,dx,c)), /255
hipError_t launch_letterbox_s2d(const LetterboxDev* geom_dev, int n, int out_h, int out_w,
                                uint16_t* out, hipStream_t s);
// SPPF: three chained 5x5/s1/p2 max pools of slice 0 written to slices 1..3 of the same buffer
hipError_t launch_sppf_pool(uint16_t* buf, int ld, int c, int n, int h, int w, int k, hipStream_t s);
// nearest x2 upsample of a view into a view
hipError_t launch_upsample2x(const uint16_t* in, int ld_in, uint16_t* out, int ld_out, int c,
                             int n, int h, int w, hipStream_t s);
// strided channel-slice copy
hipError_t launch_copy_view(const uint16_t* in, int ld_in, uint16_t* out, int ld_out, int c,
                            long long pixels, hipStream_t s);
// Detect decode of one level: logits fp32 [n*ny*nx][ld] -> pred[n][n_anchors][no]
hipError_t launch_detect_decode(const float* logits, int ld, float* pred, int n, int ny, int nx,
                                int na, int no, int n_anchors, int level_off, float stride,
                                const float* anchors_px /*device, [na][2]*/, hipStream_t s);
// debug readback: NHWC bf16 view -> NCHW fp32
hipError_t launch_nhwc_to_nchw_f32(const uint16_t* in, int ld, float* out, int n, int c, int h,
                                   int w, hipStream_t s);
hipError_t launch_s2d_to_nchw_f32(const uint16_t* in, float* out, int n, int h, int w, hipStream_t s);

// ---------------------------------------------------------------------------------------
// NMS (nms_kernels.cpp)
// ---------------------------------------------------------------------------------------
struct NmsScratch {
    uint32_t* keys[2];     // [n][cap] each
    uint32_t* vals[2];
    int cap;               // candidates capacity per image (= max anchors)
};
hipError_t launch_nms(const float* pred, int n, int n_anchors, int no, float conf_thres,
                      float iou_thres, int max_det, const NmsScratch& scr, float* out /*device [n][max_det][6]*/,
                      int* counts /*device [n]*/, hipStream_t s);
constexpr int kNmsMaxDet = 1024;

}  // namespace mdhip
