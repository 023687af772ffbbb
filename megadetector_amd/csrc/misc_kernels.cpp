// HBM-bound kernels of the MDv5 hot path (gfx950): letterbox + normalise + space-to-depth,
// SPPF max-pools, nearest upsample, Detect decode, and debug read-back helpers.
//
// Compiled with -ffp-contract=off: the fixed-point bilinear coordinates and the Detect
// decode must round exactly like the un-fused NumPy / PyTorch expressions they replace.

#include <algorithm>
#include <cstdlib>

#include "mdhip_internal.h"

namespace mdhip {

// ---------------------------------------------------------------------------------------
// letterbox_s2d: replaces  cv2.resize(INTER_LINEAR) + copyMakeBorder(114)  inside yolov5
// letterbox() (reference pytorch_detector.py:1104-1109) and  transpose/float()/ /255
// (reference pytorch_detector.py:1283-1306).
//
// Output layout: bf16 [n][out_h/2][out_w/2][16]; channel (dy*2+dx)*3 + c holds pixel
// (2Y+dy, 2X+dx) colour c, channels 12..15 are zero.  In this space-to-depth form the model's
// first layer (Conv 6x6 stride 2 pad 2 on 3 channels) is an ordinary 3x3 stride-1 pad-1
// convolution on 16 channels, i.e. one more implicit-GEMM call with 32-byte pixels.
//
// Bilinear = OpenCV's 8-bit path: 11-bit coefficients, horizontal pass to int, vertical
// pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double linear_scale(int dst_len, int src_len) { return 1.0 / ((double)dst_len / (double)src_len); }
__device__ __forceinline__ void linear_coef_s(int d, double scale, int src_len, int& s0, int& s1, int& w0, int& w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src_len - 1) { f = 0.f; s = src_len - 1; }
    w1 = __float2int_rn(f * 2048.0f);
    w0 = __float2int_rn((1.0f - f) * 2048.0f);
    s0 = s;
    s1 = min(s + 1, src_len - 1);
}
__device__ __forceinline__ void linear_coef(int d, int dst_len, int src_len, int& s0, int& s1,
                                            int& w0, int& w1) {
    const double scale = linear_scale(dst_len, src_len);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src_len - 1) { f = 0.f; s = src_len - 1; }
    w1 = __float2int_rn(f * 2048.0f);
    w0 = __float2int_rn((1.0f - f) * 2048.0f);
    s0 = s;
    s1 = min(s + 1, src_len - 1);
}

// One axis of cv::computeResizeAreaTab (INTER_AREA, shrinking): destination index d covers source
// [d*scale, (d+1)*scale); entries in accumulation order: optional left partial, whole cells, optional right partial.
struct AreaAxis {
    int s_left, s1, s2;          // left partial at s_left (if a_left >= 0), whole cells [s1, s2), right partial at s2
    float a_left, a_mid, a_right; // weights; a_left / a_right < 0 when absent
    __device__ int count() const { return (a_left >= 0.f) + (s2 - s1) + (a_right >= 0.f); }
    __device__ void entry(int k, int& si, float& a) const {
        if (a_left >= 0.f) { if (k == 0) { si = s_left; a = a_left; return; } --k; }
        if (k < s2 - s1) { si = s1 + k; a = a_mid; return; }
        si = s2; a = a_right;
    }
};
__device__ __forceinline__ AreaAxis area_axis(int d, int dst_len, int src_len) {
    const double scale = 1.0 / ((double)dst_len / (double)src_len);
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)src_len - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, src_len - 1);
    s1 = min(s1, s2);
    AreaAxis t;
    t.s_left = s1 - 1; t.s1 = s1; t.s2 = s2;
    t.a_left = ((double)s1 - f1 > 1e-3) ? (float)(((double)s1 - f1) / cell) : -1.f;
    t.a_mid = (float)(1.0 / cell);
    t.a_right = (f2 - (double)s2 > 1e-3) ? (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell) : -1.f;
    return t;
}

__global__ void __launch_bounds__(256)
letterbox_s2d_kernel(const LetterboxDev* __restrict__ geom, int out_h, int out_w,
                     uint16_t* __restrict__ out, int f16) {
    const int img = blockIdx.z;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;     // s2d column
    const int Y = blockIdx.y;                                // s2d row
    const int W2 = out_w >> 1, H2 = out_h >> 1;
    if (X >= W2) return;
    const LetterboxDev g = geom[img];
    const bool resize = (g.resized_h != g.src_h) || (g.resized_w != g.src_w);
    // INTER_AREA (compatibility_mode 'modern', shrinking): integer factors in both directions take OpenCV's
    // integer path, everything else the float table path
    const bool area = resize && g.interp == 1;
    int isx = 0, isy = 0;
    if (area) {
        const double sx = 1.0 / ((double)g.resized_w / (double)g.src_w), sy = 1.0 / ((double)g.resized_h / (double)g.src_h);
        const int rx = (int)rint(sx), ry = (int)rint(sy);
        if (fabs(sx - rx) < 2.220446049250313e-16 && fabs(sy - ry) < 2.220446049250313e-16) { isx = rx; isy = ry; }
    }

    uint16_t px[16];
#pragma unroll
    for (int i = 12; i < 16; ++i) px[i] = 0;

#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int y = 2 * Y + dy - g.top;
        int y0 = 0, y1 = 0, b0 = 2048, b1 = 0;
        const bool y_in = (unsigned)y < (unsigned)g.resized_h;
        if (y_in && resize && !area) linear_coef(y, g.resized_h, g.src_h, y0, y1, b0, b1);
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = 2 * X + dx - g.left;
            int v[3] = {114, 114, 114};
            if (y_in && (unsigned)x < (unsigned)g.resized_w) {
                if (!resize) {
                    const uint8_t* s = g.src + ((size_t)y * g.src_w + x) * 3;
                    v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
                } else if (area && isx > 0) {
                    // resizeAreaFast_: block sum; 2x2 as (s + 2) >> 2, else saturate_cast<uchar>(sum * (1.f / area))
                    int sum[3] = {0, 0, 0};
                    for (int yy = 0; yy < isy; ++yy) {
                        const uint8_t* r = g.src + ((size_t)(y * isy + yy) * g.src_w + (size_t)x * isx) * 3;
                        for (int xx = 0; xx < isx; ++xx) { sum[0] += r[xx * 3]; sum[1] += r[xx * 3 + 1]; sum[2] += r[xx * 3 + 2]; }
                    }
                    if (isx == 2 && isy == 2) {
                        v[0] = (sum[0] + 2) >> 2; v[1] = (sum[1] + 2) >> 2; v[2] = (sum[2] + 2) >> 2;
                    } else {
                        const float sc = 1.0f / (float)(isx * isy);
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[c] = min(max(__float2int_rn((float)sum[c] * sc), 0), 255);
                    }
                } else if (area) {
                    // resizeArea_<uchar,float>: per source row buf = sum_k S[sx_k]*alpha_k, then sum = beta_0*buf_0,
                    // sum += beta_j*buf_j; fp32 in table order; round half to even
                    const AreaAxis ax = area_axis(x, g.resized_w, g.src_w), ay = area_axis(y, g.resized_h, g.src_h);
                    const int nx = ax.count(), nyy = ay.count();
                    float acc[3] = {0.f, 0.f, 0.f};
                    for (int j = 0; j < nyy; ++j) {
                        int sy; float beta;
                        ay.entry(j, sy, beta);
                        const uint8_t* r = g.src + (size_t)sy * g.src_w * 3;
                        float buf[3] = {0.f, 0.f, 0.f};
                        for (int k = 0; k < nx; ++k) {
                            int sxk; float a;
                            ax.entry(k, sxk, a);
                            buf[0] = buf[0] + (float)r[sxk * 3] * a;
                            buf[1] = buf[1] + (float)r[sxk * 3 + 1] * a;
                            buf[2] = buf[2] + (float)r[sxk * 3 + 2] * a;
                        }
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[c] = j == 0 ? beta * buf[c] : acc[c] + beta * buf[c];
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = min(max(__float2int_rn(acc[c]), 0), 255);
                } else {
                    int x0, x1, a0, a1;
                    linear_coef(x, g.resized_w, g.src_w, x0, x1, a0, a1);
                    const uint8_t* r0 = g.src + (size_t)y0 * g.src_w * 3;
                    const uint8_t* r1 = g.src + (size_t)y1 * g.src_w * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int t0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
                        const int t1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
                        int o = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
                        v[c] = min(max(o, 0), 255);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                px[(dy * 2 + dx) * 3 + c] = f32_to_st((float)v[c] / 255.0f, f16);
        }
    }
    uint4* dst = (uint4*)(out + (((size_t)img * H2 + Y) * W2 + X) * 16);
    uint4 lo, hi;
    lo.x = px[0] | ((uint32_t)px[1] << 16);   lo.y = px[2] | ((uint32_t)px[3] << 16);
    lo.z = px[4] | ((uint32_t)px[5] << 16);   lo.w = px[6] | ((uint32_t)px[7] << 16);
    hi.x = px[8] | ((uint32_t)px[9] << 16);   hi.y = px[10] | ((uint32_t)px[11] << 16);
    hi.z = px[12] | ((uint32_t)px[13] << 16); hi.w = px[14] | ((uint32_t)px[15] << 16);
    dst[0] = lo;
    dst[1] = hi;
}

// [r5] The same transform for batches in which NO image is resampled (resized == source size: the synthetic 1280x1280
// workload of BASELINE configs[1], video frames and camera images that already have the network's long side) -- a
// streaming copy, so it is built like one.  The general kernel above spends its time in the address path: twelve scattered
// byte loads and twelve IEEE divisions per thread (0.22 ms for 32 images = 2.1 TB/s).  Here a workgroup owns one
// space-to-depth row (two source rows) of one image:
//   phase 0: a 256-entry table u8 -> storage type of (float)v / 255.0f, computed with the very expression of the general
//            kernel (bit-identical by construction), in LDS;
//   phase 1: the two source rows come in as ALIGNED dwords (two loads + v_alignbyte for a misaligned row start; the <= 2
//            groups per row that straddle the image's left / right edge byte by byte), go through the table and land
//            in LDS as 16-bit values at their output column (padding columns / rows: table[114]);
//   phase 2: one 16-byte store per lane, consecutive lanes = consecutive 16-byte chunks of the output row
//            ([row0 px 2X, 2X+1 | row1 px 2X, 2X+1 | 0 0 0 0] = dwords row0[3X..3X+2], row1[3X..3X+2], 0, 0).
// Measured (batch 32, 1280x1280, live events incl. the geometry upload): 0.223 -> 0.154 ms (3.07 TB/s).  Four s2d rows per
// workgroup with the next row's dwords prefetched into registers: 0.170 ms -- slower (a quarter of the workgroups, 16 more
// registers); not kept.
// Geometry of up to kLbInline images travels in the kernel arguments (scalar loads from the kernarg segment): no H2D copy
// in front of the launch (the 1.3 KB upload cost 10 - 15 us of stream time per batch); larger batches read `ptr`.
constexpr int kLbInline = 32;
struct LetterboxGeom {
    const LetterboxDev* ptr;
    LetterboxDev inl[kLbInline];
};
__global__ void __launch_bounds__(256)
letterbox_copy_s2d_kernel(const LetterboxGeom geom, int out_h, int out_w,
                          uint16_t* __restrict__ out, int f16) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lb_lds[];
    uint16_t* lut = (uint16_t*)lb_lds;                          // 256 entries
    uint32_t* rows = lb_lds + 128;                              // 2 rows x (out_w * 3 / 2) dwords of 16-bit pairs
    const int img = blockIdx.y, Y = blockIdx.x, t = threadIdx.x;
    const int W2 = out_w >> 1, H2 = out_h >> 1;
    const int row_dw = (out_w * 3) >> 1;                        // dwords per staged row (out_w is a multiple of 4)
    const LetterboxDev g = geom.ptr ? geom.ptr[img] : geom.inl[img];
    lut[t] = f32_to_st((float)t / 255.0f, f16);
    __syncthreads();
    const uint32_t pad1 = lut[114], pad2 = pad1 | (pad1 << 16);
    const int row_bytes = g.src_w * 3, left3 = g.left * 3;
    const int groups = (out_w * 3) >> 2;                        // groups of 4 output bytes per row
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int y = 2 * Y + dy - g.top;
        const bool y_in = (unsigned)y < (unsigned)g.src_h;
        const uint8_t* rb = g.src + (size_t)(y_in ? y : 0) * row_bytes;
        uint32_t* dst = rows + dy * row_dw;
        for (int gi = t; gi < groups; gi += 256) {
            const int o = 4 * gi - left3;                       // source byte offset of this group's first byte
            uint32_t lo16 = pad2, hi16 = pad2;
            if (y_in && o + 3 >= 0 && o < row_bytes) {
                uint32_t v;
                if (o >= 0 && o + 3 < row_bytes) {
                    const uintptr_t a = (uintptr_t)(rb + o);
                    const uint32_t* al = (const uint32_t*)(a & ~(uintptr_t)3);
                    const uint32_t sh = (uint32_t)(a & 3);
                    const uint32_t w0 = al[0];
                    const uint32_t w1 = sh ? al[1] : 0u;
                    v = __builtin_amdgcn_alignbyte(w1, w0, sh);
                } else {                                        // the <= 2 groups of a row that straddle an image edge
                    v = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ok = o + k;
                        v |= (((unsigned)ok < (unsigned)row_bytes) ? (uint32_t)rb[ok] : 114u) << (8 * k);   // padding byte = 114
                    }
                }
                lo16 = lut[v & 0xff] | ((uint32_t)lut[(v >> 8) & 0xff] << 16);
                hi16 = lut[(v >> 16) & 0xff] | ((uint32_t)lut[v >> 24] << 16);
            }
            *(uint2*)(dst + 2 * gi) = make_uint2(lo16, hi16);
        }
    }
    __syncthreads();
    uint4* orow = (uint4*)(out + ((size_t)img * H2 + Y) * (size_t)W2 * 16);
    const uint32_t* r0 = rows;
    const uint32_t* r1 = rows + row_dw;
    for (int q = t; q < 2 * W2; q += 256) {
        const int X = q >> 1;
        uint4 v;
        if (q & 1) v = make_uint4(r1[3 * X + 1], r1[3 * X + 2], 0u, 0u);
        else v = make_uint4(r0[3 * X], r0[3 * X + 1], r0[3 * X + 2], r1[3 * X]);
        orow[q] = v;
    }
}


// [r6] The bilinear path (cv2.INTER_LINEAR, what yolov5's letterbox() runs for every real camera image: reference
// pytorch_detector.py:1104-1109) with the general kernel's work per output value cut to what the arithmetic needs.  The general
// kernel spends a thread's time on 6 double-precision divisions (the scale of linear_coef, per call), 12 IEEE divisions by 255,
// and 48 byte loads (each a full 64-lane gather instruction).  Here, one thread per space-to-depth pixel as there:
//   * the two scales come with the geometry (computed on the host with the same IEEE double expression: same bits);
//   * the u8 -> storage-type conversion is the copy kernel's 256-entry table in LDS (the general kernel's expression);
//   * two RGB pixels of a source row = six consecutive bytes: ONE 12-byte load from the aligned address + two v_alignbyte; channel
//     c of both pixels as (p0 | p1 << 16) with one v_perm_b32, OpenCV's horizontal pass p0 * a0 + p1 * a1 with one
//     v_dot2_u32_u16 -- 8 memory instructions per thread instead of 48, integer arithmetic with the general kernel's values;
//   * the thread's 32 output bytes go through a wave-private 2 KiB of LDS so that each of its two stores covers 1 KiB of contiguous
//     output (16 bytes per lane on consecutive chunks) instead of every other 16 bytes of 2 KiB.
// (Sharing the horizontal pass of a source row between the thread's two output rows was tried: the selects it needs cost more
// than the passes it saves.)
// First built as a staged kernel (source rows in LDS as aligned dwords, output rows staged for 16-byte stores, two barriers): bit-exact
// and 10 - 50 % SLOWER than the general kernel on every shape (profiles/r6_letterbox.txt) -- at 15 waves per CU its three
// barrier-separated phases leave the dependent LDS chains (window -> table) uncovered.  This form has no phase structure.
// At the right edge (x0 == src_w - 1, where a1 == 0) the window starts one pixel earlier and the weights swap places, so that
// no window reaches past its row; a window whose 12-byte load would reach past the dword of the image's last byte (the last
// pixels of the last row) is read byte by byte.
constexpr int kLbRows = 2;
__global__ void __launch_bounds__(256)
letterbox_linear_s2d_kernel(const LetterboxGeom geom, int out_h, int out_w, uint16_t* __restrict__ out, int f16) {
    __shared__ uint16_t lut[256];
    __shared__ uint4 stage[4 * 128];                            // 2 KiB per wave (the store transpose below)
    const int img = blockIdx.z, t = threadIdx.x;
    const int X = blockIdx.x * 256 + t;                         // s2d column
    const int Y = blockIdx.y;                                   // s2d rows Y * kLbRows ..
    const int W2 = out_w >> 1, H2 = out_h >> 1;
    const LetterboxDev g = geom.ptr ? geom.ptr[img] : geom.inl[img];
    lut[t] = f32_to_st((float)t / 255.0f, f16);
    __syncthreads();
    const int row_bytes = g.src_w * 3;
    typedef const __attribute__((address_space(1))) uint8_t* gptr_t;
    const uintptr_t src = (uintptr_t)g.src;
    const long long image_bytes = (long long)g.src_h * row_bytes;
    // columns: window start (byte offset in a row) and the packed weights of its two pixels
    int wo[2];
    uint32_t aw[2];
    bool x_in[2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
        const int x = 2 * X + dx - g.left;
        x_in[dx] = X < W2 && (unsigned)x < (unsigned)g.resized_w;
        int x0 = 0, x1 = 0, a0 = 2048, a1 = 0;
        if (x_in[dx]) linear_coef_s(x, g.sx, g.src_w, x0, x1, a0, a1);
        const bool edge = x0 == g.src_w - 1 && g.src_w > 1;     // (then x1 == x0 and a1 == 0)
        wo[dx] = (edge ? x0 - 1 : x0) * 3;
        aw[dx] = edge ? ((uint32_t)a0 << 16) : ((uint32_t)a0 | ((uint32_t)a1 << 16));
    }
    // horizontal pass of source row r at window offset o: t[c] = p0[c] * a0 + p1[c] * a1.  The row's base address is scalar; a lane
    // adds a 32-bit offset: the aligned dword address is  (base & ~3) + ((base & 3) + o & ~3),  the shift  (base & 3) + o & 3
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto hpass = [&](int r, int o, uint32_t w2, uint32_t (&tc)[3]) __attribute__((always_inline)) {
        const long long row_off = (long long)r * row_bytes;                   // (scalar)
        const uintptr_t base = src + (uintptr_t)row_off;
        const uintptr_t base_al = base & ~(uintptr_t)3;
        const uint32_t bo = (uint32_t)(base & 3) + (uint32_t)o;
        const uint32_t voff = bo & ~3u, sh = bo & 3u;
        // bytes from base_al to the end of the dword that holds the image's last byte (scalar; an image is far below 4 GB)
        const uint32_t room = (uint32_t)((((src + (uintptr_t)image_bytes + 3) & ~(uintptr_t)3)) - base_al);
        uint32_t lo, hi;
        if (voff + 12u <= room && g.src_w > 1) {
            typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));
            const u32x3a w = *(const __attribute__((address_space(1))) u32x3a*)(base_al + voff);
            lo = __builtin_amdgcn_alignbyte(w[1], w[0], sh);                   // bytes o .. o + 3
            hi = __builtin_amdgcn_alignbyte(w[2], w[1], sh);                   // bytes o + 4 .. o + 7
        } else {                                                               // the image's last bytes (or a one-pixel-wide source)
            gptr_t q = (gptr_t)(base + (uintptr_t)o);
            const int left = (int)(image_bytes - row_off - o);                 // bytes of the image from there on (>= 3)
            lo = hi = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t v = k < left ? (uint32_t)q[k] : 0u;
                if (k < 4) lo |= v << (8 * k); else hi |= v << (8 * (k - 4));
            }
        }
        const u16x2 a2 = *(const u16x2*)&w2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t pp = __builtin_amdgcn_perm(hi, lo, (uint32_t)(c | (0x0c << 8) | ((c + 3) << 16) | (0x0cu << 24)));
            tc[c] = __builtin_amdgcn_udot2(*(const u16x2*)&pp, a2, 0u, false);
        }
    };
    // kLbRows space-to-depth rows per thread: the column weights above (a dozen double-precision instructions per column) are
    // computed once for them, and the loads of the second row are in flight while the first is being finished
#pragma unroll
    for (int rr = 0; rr < kLbRows; ++rr) {
        const int Yr = Y * kLbRows + rr;
        if (Yr >= H2) break;                                    // (workgroup-uniform)
        // rows: weights of the two output rows and the source rows they need.  The values are the same in every lane (they depend on
        // blockIdx only); v_readfirstlane says so to the compiler, which then keeps them -- and the row base addresses below -- in
        // scalar registers
        int ysrc[4] = {0, 0, 0, 0}, bw[4] = {2048, 0, 2048, 0};
        bool y_in[2];
    #pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * Yr + dy - g.top;
            y_in[dy] = (unsigned)y < (unsigned)g.resized_h;
            if (y_in[dy]) linear_coef_s(y, g.sy, g.src_h, ysrc[2 * dy], ysrc[2 * dy + 1], bw[2 * dy], bw[2 * dy + 1]);
        }
    #pragma unroll
        for (int k = 0; k < 4; ++k) {
            ysrc[k] = __builtin_amdgcn_readfirstlane(ysrc[k]);
            bw[k] = __builtin_amdgcn_readfirstlane(bw[k]);
        }
        uint32_t th[4][2][3];                                       // [source-row slot][dx][channel]
    #pragma unroll
        for (int k = 0; k < 4; ++k) {
    #pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                th[k][dx][0] = th[k][dx][1] = th[k][dx][2] = 0u;
                if (y_in[k >> 1] && x_in[dx]) hpass(ysrc[k], wo[dx], aw[dx], th[k][dx]);
            }
        }
        const uint32_t pad1 = lut[114];
        uint32_t o16[2][6];
    #pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int b0 = bw[2 * dy], b1 = bw[2 * dy + 1];
    #pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
    #pragma unroll
                for (int c = 0; c < 3; ++c) {
                    uint32_t v = pad1;
                    if (y_in[dy] && x_in[dx]) {
                        // (24-bit multiplies: the weights are <= 2048, the passes' sums >> 4 below 2^15 -- the 32-bit multiply is quarter rate)
                        const uint32_t t0 = th[2 * dy][dx][c], t1 = th[2 * dy + 1][dx][c];
                        const int o = (int)(((__umul24((unsigned)b0, t0 >> 4) >> 16) + (__umul24((unsigned)b1, t1 >> 4) >> 16) + 2u) >> 2);
                        v = lut[min(max(o, 0), 255)];
                    }
                    o16[dy][3 * dx + c] = v;
                }
            }
        }
        const uint4 v0 = make_uint4(o16[0][0] | (o16[0][1] << 16), o16[0][2] | (o16[0][3] << 16), o16[0][4] | (o16[0][5] << 16), o16[1][0] | (o16[1][1] << 16));
        const uint4 v1 = make_uint4(o16[1][2] | (o16[1][3] << 16), o16[1][4] | (o16[1][5] << 16), 0u, 0u);
        // wave-private transpose: lane l holds chunks 2l, 2l + 1 of its wave's 128 consecutive 16-byte chunks; it stores chunks l and 64 + l
        const int lane = t & 63, wv = t >> 6;
        uint4* st = stage + wv * 128;
        st[2 * lane] = v0;
        st[2 * lane + 1] = v1;
        __builtin_amdgcn_wave_barrier();
        const uint4 c0 = st[lane], c1 = st[64 + lane];
        // (the wave's s2d pixels X0 .. X0 + 63; those at or behind W2 belong to no row of this image)
        const int X0 = blockIdx.x * 256 + wv * 64;
        uint4* dst = (uint4*)(out + (((size_t)img * H2 + Yr) * W2 + X0) * 16);
        if (X0 + (lane >> 1) < W2) dst[lane] = c0;
        if (X0 + 32 + (lane >> 1) < W2) dst[64 + lane] = c1;
    }
}

// which kernel a batch takes: 1 = streaming copy (no image is resampled), 2 = streaming bilinear (every resampled image with
// cv2.INTER_LINEAR, source rows short enough for the LDS), 0 = the general kernel (INTER_AREA, very wide sources, odd widths)
struct LbPlan { int kind; size_t lds; };
static LbPlan lb_plan(const LetterboxDev* g, int n, int out_w, bool force_general) {
    LbPlan p{0, 0};
    if (force_general) return p;
    bool no_resampling = true, linear = true;
    for (int i = 0; i < n; ++i) {
        const bool rs = g[i].resized_h != g[i].src_h || g[i].resized_w != g[i].src_w;
        no_resampling = no_resampling && !rs;
        linear = linear && (!rs || g[i].interp == 0);
    }
    if (no_resampling) {
        p.lds = 512 + (size_t)out_w * 12;
        if ((out_w % 4) == 0 && p.lds <= 65536) p.kind = 1;
        else p.kind = 2;                                       // (the bilinear kernel with unit weights: exact)
        return p;
    }
    if (linear) p.kind = 2;
    return p;
}
bool letterbox_geometry_travels_inline(const LetterboxDev* geom_host, int n, int out_w, bool force_general) {
    return lb_plan(geom_host, n, out_w, force_general).kind != 0 && n <= kLbInline;
}

hipError_t launch_letterbox_s2d(const LetterboxDev* geom_dev, const LetterboxDev* geom_host, int n, int out_h, int out_w,
                                uint16_t* out, int f16, bool force_general, hipStream_t s) {
    const int W2 = out_w / 2, H2 = out_h / 2;
    const LbPlan plan = lb_plan(geom_host, n, out_w, force_general);
    if (plan.kind != 0) {
        LetterboxGeom geom;
        geom.ptr = geom_dev;
        if (n <= kLbInline) {
            geom.ptr = nullptr;
            for (int i = 0; i < n; ++i) geom.inl[i] = geom_host[i];
            for (int i = n; i < kLbInline; ++i) geom.inl[i] = LetterboxDev{nullptr, 0, 0, 0, 0, 0, 0, 0, 1.0, 1.0};
        }
        if (plan.kind == 1) hipLaunchKernelGGL(letterbox_copy_s2d_kernel, dim3(H2, n), dim3(256), plan.lds, s, geom, out_h, out_w, out, f16);
        else hipLaunchKernelGGL(letterbox_linear_s2d_kernel, dim3((W2 + 255) / 256, (H2 + kLbRows - 1) / kLbRows, n), dim3(256), 0, s, geom, out_h, out_w, out, f16);
        return hipGetLastError();
    }
    dim3 grid((W2 + 255) / 256, H2, n);
    hipLaunchKernelGGL(letterbox_s2d_kernel, grid, dim3(256), 0, s, geom_dev, out_h, out_w, out, f16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// 16-byte (8 x bf16 / fp16) vector helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t max2_st(uint32_t a, uint32_t b, int f16) {
    const float a0 = st_to_f32((uint16_t)(a & 0xffff), f16), a1 = st_to_f32((uint16_t)(a >> 16), f16);
    const float b0 = st_to_f32((uint16_t)(b & 0xffff), f16), b1 = st_to_f32((uint16_t)(b >> 16), f16);
    const uint32_t lo = (b0 > a0) ? (b & 0xffff) : (a & 0xffff);
    const uint32_t hi = (b1 > a1) ? (b >> 16) : (a >> 16);
    return lo | (hi << 16);
}
__device__ __forceinline__ uint4 max8_st(uint4 a, uint4 b, int f16) {
    return make_uint4(max2_st(a.x, b.x, f16), max2_st(a.y, b.y, f16), max2_st(a.z, b.z, f16), max2_st(a.w, b.w, f16));
}

// ---------------------------------------------------------------------------------------
// SPPF pools: y1 = maxpool k(x), y2 = maxpool k(y1), y3 = maxpool k(y2)  (stride 1, pad k/2,
// -inf padding).  x is channel slice 0 of buf, y1..y3 go to slices 1..3 (yolov5
// models/common.py:SPPF.forward).
// ---------------------------------------------------------------------------------------
// one k x k / stride 1 max pool of channel slice `src` into slice `src + 1` of the same buffer; the three
// chained pools of SPPF are three launches (75 loads per output instead of the 169 of the fused
// (k), (2k-1), (3k-2) windows: 0.45 -> 0.1 ms per step at batch 32)
__global__ void __launch_bounds__(256)
sppf_pool_kernel(uint16_t* __restrict__ buf, int ld, int c8, int n, int h, int w, int r1, int src, int f16) {
    const long long total = (long long)n * h * w * c8;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ch = (int)(t % c8);
    long long pix = t / c8;
    const int x = (int)(pix % w);
    const int y = (int)((pix / w) % h);
    const int b = (int)(pix / ((long long)w * h));
    const int c = c8 * 8;
    const uint16_t* base = buf + (size_t)b * h * w * ld + src * c + ch * 8;
    const uint32_t ninf2 = f16 ? 0xfc00fc00u : 0xff80ff80u;   // two -inf
    uint4 m = make_uint4(ninf2, ninf2, ninf2, ninf2);
    for (int dy = -r1; dy <= r1; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)h) continue;
        for (int dx = -r1; dx <= r1; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)w) continue;
            m = max8_st(m, *(const uint4*)(base + ((size_t)yy * w + xx) * ld), f16);
        }
    }
    *(uint4*)(buf + ((size_t)(b * h + y) * w + x) * ld + (src + 1) * c + ch * 8) = m;
}

// The three chained pools in ONE launch when an image's map fits in LDS (the 20x20 .. 40x40 maps SPPF sees): a workgroup
// takes CH 8-channel chunks of one image, loads them once, and runs row-max then column-max (5 + 5 instead of 25 reads
// per output) three times between two LDS buffers, writing each stage to its slice.  max is exact: same bits as the
// chained launches.  0.186 -> 0.10 ms per step at batch 32.  [r6] 1024 threads per workgroup when the map has at least that many
// (pixel, chunk) items: the kernel is latency-bound (six barrier-separated passes of a few dependent LDS reads per thread), so what
// helps is more waves per pass, not fewer bytes.
template <int CH>
__global__ void __launch_bounds__(1024)
sppf_pool_lds_kernel(uint16_t* __restrict__ buf, int ld, int c8, int h, int w, int r1, int f16) {
    extern __shared__ __attribute__((aligned(16))) uint4 sp[];
    const int hw = h * w, items = hw * CH;
    uint4* A = sp;
    uint4* T = sp + items;
    const int groups = (c8 + CH - 1) / CH;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int c = c8 * 8;
    uint16_t* img = buf + (size_t)b * hw * ld;
    const uint32_t ninf2 = f16 ? 0xfc00fc00u : 0xff80ff80u;
    for (int t = threadIdx.x; t < items; t += (int)blockDim.x) {
        const int pix = t / CH, ch = g * CH + (t - pix * CH);
        A[t] = ch < c8 ? *(const uint4*)(img + (size_t)pix * ld + ch * 8) : make_uint4(ninf2, ninf2, ninf2, ninf2);
    }
    __syncthreads();
    for (int stage = 0; stage < 3; ++stage) {
        for (int t = threadIdx.x; t < items; t += (int)blockDim.x) {
            const int pix = t / CH, k = t - pix * CH, y = pix / w, x = pix - y * w;
            uint4 m = make_uint4(ninf2, ninf2, ninf2, ninf2);
            for (int dx = -r1; dx <= r1; ++dx)
                if ((unsigned)(x + dx) < (unsigned)w) m = max8_st(m, A[(pix + dx) * CH + k], f16);
            T[t] = m;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < items; t += (int)blockDim.x) {
            const int pix = t / CH, k = t - pix * CH, y = pix / w, ch = g * CH + k;
            uint4 m = make_uint4(ninf2, ninf2, ninf2, ninf2);
            for (int dy = -r1; dy <= r1; ++dy)
                if ((unsigned)(y + dy) < (unsigned)h) m = max8_st(m, T[(pix + dy * w) * CH + k], f16);
            A[t] = m;
            if (ch < c8) *(uint4*)(img + (size_t)pix * ld + (stage + 1) * c + ch * 8) = m;
        }
        __syncthreads();
    }
}

hipError_t launch_sppf_pool(uint16_t* buf, int ld, int c, int n, int h, int w, int k, int f16, hipStream_t s) {
    const long long total = (long long)n * h * w * (c / 8);
    const size_t per_chunk = (size_t)h * w * 16 * 2;                // two LDS buffers of one 8-channel chunk of the map
    if (per_chunk * 4 <= 65536) {
        // [r6] two chunks (16 channels) per workgroup when that still gives a thread per item: twice the workgroups, each
        // with half the passes' work.  Measured against four chunks in one session: 0.0935 / 0.0947 ms at batch 32 -- no
        // difference: the kernel is bound by its six barrier-separated passes either way (profiles/r6_read_amplification.txt)
        const int items2 = h * w * 2;
        if (items2 >= 512 && items2 <= 1024) {
            hipLaunchKernelGGL(sppf_pool_lds_kernel<2>, dim3((unsigned)(n * ((c / 8 + 1) / 2))), dim3((items2 + 63) / 64 * 64), per_chunk * 2, s,
                               buf, ld, c / 8, h, w, k / 2, f16);
            return hipGetLastError();
        }
        const int threads = h * w * 4 >= 1024 ? 1024 : 256;
        hipLaunchKernelGGL(sppf_pool_lds_kernel<4>, dim3((unsigned)(n * ((c / 8 + 3) / 4))), dim3(threads), per_chunk * 4, s,
                           buf, ld, c / 8, h, w, k / 2, f16);
        return hipGetLastError();
    }
    if (per_chunk <= 65536) {
        hipLaunchKernelGGL(sppf_pool_lds_kernel<1>, dim3((unsigned)(n * (c / 8))), dim3(256), per_chunk, s,
                           buf, ld, c / 8, h, w, k / 2, f16);
        return hipGetLastError();
    }
    for (int src = 0; src < 3; ++src)
        hipLaunchKernelGGL(sppf_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           buf, ld, c / 8, n, h, w, k / 2, src, f16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// nearest x2 upsample (nn.Upsample(scale_factor=2, mode='nearest')) of a view into a view
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
upsample2x_kernel(const uint16_t* __restrict__ in, int ld_in, uint16_t* __restrict__ out, int ld_out,
                  int c8, int n, int h, int w) {
    const int ho = 2 * h, wo = 2 * w;
    const long long total = (long long)n * ho * wo * c8;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ch = (int)(t % c8);
    long long pix = t / c8;
    const int x = (int)(pix % wo);
    const int y = (int)((pix / wo) % ho);
    const int b = (int)(pix / ((long long)wo * ho));
    const uint4 v = *(const uint4*)(in + ((size_t)(b * h + (y >> 1)) * w + (x >> 1)) * ld_in + ch * 8);
    *(uint4*)(out + ((size_t)(b * ho + y) * wo + x) * ld_out + ch * 8) = v;
}

hipError_t launch_upsample2x(const uint16_t* in, int ld_in, uint16_t* out, int ld_out, int c,
                             int n, int h, int w, hipStream_t s) {
    const long long total = (long long)n * 4 * h * w * (c / 8);
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       in, ld_in, out, ld_out, c / 8, n, h, w);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
copy_view_kernel(const uint16_t* __restrict__ in, int ld_in, uint16_t* __restrict__ out, int ld_out,
                 int c8, long long pixels) {
    const long long total = pixels * c8;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ch = (int)(t % c8);
    const long long pix = t / c8;
    *(uint4*)(out + (size_t)pix * ld_out + ch * 8) = *(const uint4*)(in + (size_t)pix * ld_in + ch * 8);
}

hipError_t launch_copy_view(const uint16_t* in, int ld_in, uint16_t* out, int ld_out, int c,
                            long long pixels, hipStream_t s) {
    const long long total = pixels * (c / 8);
    hipLaunchKernelGGL(copy_view_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       in, ld_in, out, ld_out, c / 8, pixels);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// max |x| over a 16-bit view (fp8 calibration: the range of a bottleneck's hidden tensor)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
absmax_view_kernel(const uint16_t* __restrict__ in, int ld, int c8, long long pixels, int f16, unsigned* __restrict__ out) {
    const long long total = pixels * c8;
    float m = 0.f;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long pix = t / c8;
        const int ch = (int)(t - pix * c8);
        const uint4 v = *(const uint4*)(in + (size_t)pix * ld + ch * 8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m = fmaxf(m, fabsf(st_to_f32((uint16_t)(w[k] & 0xffff), f16)));
            m = fmaxf(m, fabsf(st_to_f32((uint16_t)(w[k] >> 16), f16)));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(out, __float_as_uint(m));      // non-negative floats order as integers
}

hipError_t launch_absmax_view(const uint16_t* in, int ld, int c, long long pixels, int f16, float* out, hipStream_t s) {
    const long long total = pixels * (c / 8);
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(absmax_view_kernel, dim3(blocks), dim3(256), 0, s, in, ld, c / 8, pixels, f16, (unsigned*)out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Detect decode (yolov5 models/yolo.py:Detect.forward, inference branch):
//   y = sigmoid(logits);  xy = (y*2 + grid) * stride, grid = (x-0.5, y-0.5)
//   wh = (y*2)^2 * anchor_px;  rows ordered (anchor, y, x) inside the level.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f32(float x) { return mdhip_sigmoid_exact(x); }

// one thread per (image, anchor, pixel): the general form (any number of anchors / outputs)
__global__ void __launch_bounds__(256)
detect_decode_kernel(const float* __restrict__ logits, int ld, float* __restrict__ pred, int n,
                     int ny, int nx, int na, int no, int n_anchors, int level_off, float stride,
                     const float* __restrict__ anchors_px, DecodeTta tta) {
    const long long total = (long long)n * na * ny * nx;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % nx);
    const int y = (int)((t / nx) % ny);
    const int a = (int)((t / ((long long)nx * ny)) % na);
    const int b = (int)(t / ((long long)nx * ny * na));
    // anchor index inside this forward pass; a test-time-augmentation pass keeps [keep_from, keep_to) of its
    // anchors and appends them at out_off of the concatenated prediction (identity for a plain forward)
    const int idx = level_off + (a * ny + y) * nx + x;
    if (idx < tta.keep_from || idx >= tta.keep_to) return;
    const float* l = logits + ((size_t)(b * ny + y) * nx + x) * ld + a * no;
    float* o = pred + ((size_t)b * n_anchors + tta.out_off + (idx - tta.keep_from)) * no;
    const float s0 = sigmoid_f32(l[0]), s1 = sigmoid_f32(l[1]);
    const float s2 = sigmoid_f32(l[2]), s3 = sigmoid_f32(l[3]);
    float cx = (s0 * 2.0f + ((float)x - 0.5f)) * stride;
    float cy = (s1 * 2.0f + ((float)y - 0.5f)) * stride;
    const float w2 = s2 * 2.0f, h2 = s3 * 2.0f;
    float bw = (w2 * w2) * anchors_px[a * 2 + 0];
    float bh = (h2 * h2) * anchors_px[a * 2 + 1];
    if (tta.scale != 1.0f) {          // yolov5 _descale_pred: p[..., :4] /= scale, then un-flip
        cx /= tta.scale; cy /= tta.scale; bw /= tta.scale; bh /= tta.scale;
    }
    if (tta.flip_lr) cx = tta.img_w - cx;
    o[0] = cx; o[1] = cy; o[2] = bw; o[3] = bh;
    for (int i = 4; i < no; ++i) o[i] = sigmoid_f32(l[i]);
}

// MegaDetector's head (3 anchors x 8 outputs = 24 floats per pixel): one thread per (image, pixel) reads its pixel's 96
// contiguous bytes once (the general kernel's threads read 32 of every 96 bytes, three times over) and writes the
// three anchors' rows; the same arithmetic, statement for statement: the same bits.
__global__ void __launch_bounds__(256)
detect_decode_3x8_kernel(const float* __restrict__ logits, int ld, float* __restrict__ pred, int n,
                         int ny, int nx, int n_anchors, int level_off, float stride,
                         const float* __restrict__ anchors_px, DecodeTta tta) {
    const long long total = (long long)n * ny * nx;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % nx);
    const int y = (int)((t / nx) % ny);
    const int b = (int)(t / ((long long)nx * ny));
    const float4* lp = (const float4*)(logits + ((size_t)(b * ny + y) * nx + x) * ld);
    float4 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = lp[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int idx = level_off + (a * ny + y) * nx + x;
        if (idx < tta.keep_from || idx >= tta.keep_to) continue;
        const float4 p = v[2 * a], q = v[2 * a + 1];
        const float s0 = sigmoid_f32(p.x), s1 = sigmoid_f32(p.y);
        const float s2 = sigmoid_f32(p.z), s3 = sigmoid_f32(p.w);
        float cx = (s0 * 2.0f + ((float)x - 0.5f)) * stride;
        float cy = (s1 * 2.0f + ((float)y - 0.5f)) * stride;
        const float w2 = s2 * 2.0f, h2 = s3 * 2.0f;
        float bw = (w2 * w2) * anchors_px[a * 2 + 0];
        float bh = (h2 * h2) * anchors_px[a * 2 + 1];
        if (tta.scale != 1.0f) {
            cx /= tta.scale; cy /= tta.scale; bw /= tta.scale; bh /= tta.scale;
        }
        if (tta.flip_lr) cx = tta.img_w - cx;
        float4* o = (float4*)(pred + ((size_t)b * n_anchors + tta.out_off + (idx - tta.keep_from)) * 8);
        o[0] = make_float4(cx, cy, bw, bh);
        o[1] = make_float4(sigmoid_f32(q.x), sigmoid_f32(q.y), sigmoid_f32(q.z), sigmoid_f32(q.w));
    }
}

hipError_t launch_detect_decode(const float* logits, int ld, float* pred, int n, int ny, int nx,
                                int na, int no, int n_anchors, int level_off, float stride,
                                const float* anchors_px, const DecodeTta& tta, hipStream_t s) {
    if (na == 3 && no == 8 && (ld % 4) == 0 && ((uintptr_t)logits % 16) == 0 && ((uintptr_t)pred % 16) == 0) {
        const long long total = (long long)n * ny * nx;
        hipLaunchKernelGGL(detect_decode_3x8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           logits, ld, pred, n, ny, nx, n_anchors, level_off, stride, anchors_px, tta);
        return hipGetLastError();
    }
    const long long total = (long long)n * na * ny * nx;
    hipLaunchKernelGGL(detect_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       logits, ld, pred, n, ny, nx, na, no, n_anchors, level_off, stride, anchors_px, tta);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// test-time augmentation input (yolov5 utils/torch_utils.py:scale_img on the optionally left-right
// flipped batch): bilinear resize (torch F.interpolate, align_corners=False) of the normalised
// network input to (sh, sw), padded with 0.447 to (oh, ow); space-to-depth in, space-to-depth out
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float s2d_load(const uint16_t* in, int b, int H2, int W2, int y, int x, int c, int f16) {
    return st_to_f32(in[(((size_t)b * H2 + (y >> 1)) * W2 + (x >> 1)) * 16 + ((y & 1) * 2 + (x & 1)) * 3 + c], f16);
}

__global__ void __launch_bounds__(256)
tta_scale_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int n, int h, int w, int sh, int sw,
                 int oh, int ow, int flip_lr, int f16) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x;     // s2d column of the output
    const int Y = blockIdx.y;
    const int b = blockIdx.z;
    const int OW2 = ow >> 1, OH2 = oh >> 1;
    if (X >= OW2) return;
    const float ry = (float)h / (float)sh, rx = (float)w / (float)sw;
    uint16_t px[16];
#pragma unroll
    for (int i = 12; i < 16; ++i) px[i] = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int y = 2 * Y + dy;
        float fy = ry * ((float)y + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        const int y0 = (int)fy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = 2 * X + dx;
            float v[3] = {0.447f, 0.447f, 0.447f};
            if (y < sh && x < sw) {
                float fx = rx * ((float)x + 0.5f) - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = (int)fx;
                const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const int sx0 = flip_lr ? w - 1 - x0 : x0, sx1 = flip_lr ? w - 1 - x1 : x1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float p00 = s2d_load(in, b, h >> 1, w >> 1, y0, sx0, c, f16);
                    const float p01 = s2d_load(in, b, h >> 1, w >> 1, y0, sx1, c, f16);
                    const float p10 = s2d_load(in, b, h >> 1, w >> 1, y1, sx0, c, f16);
                    const float p11 = s2d_load(in, b, h >> 1, w >> 1, y1, sx1, c, f16);
                    v[c] = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) px[(dy * 2 + dx) * 3 + c] = f32_to_st(v[c], f16);
        }
    }
    uint4* dst = (uint4*)(out + (((size_t)b * OH2 + Y) * OW2 + X) * 16);
    uint4 lo, hi;
    lo.x = px[0] | ((uint32_t)px[1] << 16);   lo.y = px[2] | ((uint32_t)px[3] << 16);
    lo.z = px[4] | ((uint32_t)px[5] << 16);   lo.w = px[6] | ((uint32_t)px[7] << 16);
    hi.x = px[8] | ((uint32_t)px[9] << 16);   hi.y = px[10] | ((uint32_t)px[11] << 16);
    hi.z = px[12] | ((uint32_t)px[13] << 16); hi.w = px[14] | ((uint32_t)px[15] << 16);
    dst[0] = lo;
    dst[1] = hi;
}

hipError_t launch_tta_scale(const uint16_t* in, uint16_t* out, int n, int h, int w, int sh, int sw, int oh, int ow,
                            int flip_lr, int f16, hipStream_t s) {
    dim3 grid((ow / 2 + 255) / 256, oh / 2, n);
    hipLaunchKernelGGL(tta_scale_kernel, grid, dim3(256), 0, s, in, out, n, h, w, sh, sw, oh, ow, flip_lr, f16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// debug read-back
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nhwc_to_nchw_f32_kernel(const uint16_t* __restrict__ in, int ld, float* __restrict__ out, int n,
                        int c, int h, int w, int f16) {
    const long long total = (long long)n * c * h * w;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % w);
    const int y = (int)((t / w) % h);
    const int ch = (int)((t / ((long long)w * h)) % c);
    const int b = (int)(t / ((long long)w * h * c));
    out[t] = st_to_f32(in[((size_t)(b * h + y) * w + x) * ld + ch], f16);
}

hipError_t launch_nhwc_to_nchw_f32(const uint16_t* in, int ld, float* out, int n, int c, int h,
                                   int w, int f16, hipStream_t s) {
    const long long total = (long long)n * c * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       s, in, ld, out, n, c, h, w, f16);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
s2d_to_nchw_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int n, int h, int w, int f16) {
    const long long total = (long long)n * 3 * h * w;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % w);
    const int y = (int)((t / w) % h);
    const int ch = (int)((t / ((long long)w * h)) % 3);
    const int b = (int)(t / ((long long)w * h * 3));
    const int H2 = h / 2, W2 = w / 2;
    out[t] = st_to_f32(in[(((size_t)b * H2 + (y >> 1)) * W2 + (x >> 1)) * 16 + ((y & 1) * 2 + (x & 1)) * 3 + ch], f16);
}

hipError_t launch_s2d_to_nchw_f32(const uint16_t* in, float* out, int n, int h, int w, int f16, hipStream_t s) {
    const long long total = (long long)n * 3 * h * w;
    hipLaunchKernelGGL(s2d_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       s, in, out, n, h, w, f16);
    return hipGetLastError();
}

}  // namespace mdhip
