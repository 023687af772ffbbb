/*
 * mdhip.h -- C ABI of the MI355X-native MegaDetector v5 batch-inference hot path.
 *
 * The reference (agentmorris/MegaDetector) is pure Python and has NO FFI on this path; its
 * plugin seam is the duck-typed detector object returned by
 *   megadetector/detection/run_detector.py:601  load_detector(...)
 * i.e. the class  megadetector/detection/pytorch_detector.py:739  PTDetector.
 * This header is the C ABI that sits *under* that Python seam: each
 * entry point replaces one stage of PTDetector._process_batch_group
 * (pytorch_detector.py:1257-1426) and is bound from Python with ctypes
 * (megadetector_amd/_lib.py; the stub a maintainer would add is shown in INTEGRATION.md).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative MDHIP_E* code, never throws, never exits.  mdhip_last_error() returns a
 * human-readable message for the most recent failure on that context (or, with ctx == NULL,
 * the most recent mdhip_create failure on the calling thread).  A context is bound to one GPU
 * and is not thread-safe; use one context per (process, GPU).  The library owns all device
 * buffers and packed weights; the caller owns every pointer it passes in.
 */
#ifndef MDHIP_H
#define MDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDHIP_OK            0
#define MDHIP_EINVAL       -1   /* bad argument / unsupported model description */
#define MDHIP_EHIP         -2   /* a HIP runtime call failed                     */
#define MDHIP_ENOMEM       -3   /* device arena too small for the request        */
#define MDHIP_EUNSUPPORTED -4   /* valid request this build does not implement   */

/* arithmetic type of the conv stack */
#define MDHIP_DTYPE_BF16 0
#define MDHIP_DTYPE_FP8  1      /* BASELINE.json configs[4]: bf16 storage, the 3x3 convs of the bottlenecks on e4m3
                                 * operands (W8A8, block-scaled K = 128 MFMA); needs mdhip_calibrate once */
#define MDHIP_DTYPE_FP16 2      /* fp16 storage of activations and weights (fp32 accumulate): same MFMA rate as
                                 * bf16, 3 more mantissa bits -- the accuracy mode */

/* module kinds of a YOLOv5 model description (yolov5 models/yolo.py:parse_model rows) */
#define MDHIP_CONV      0
#define MDHIP_C3        1
#define MDHIP_SPPF      2
#define MDHIP_UPSAMPLE  3
#define MDHIP_CONCAT    4
#define MDHIP_DETECT    5

/* One fused (conv + folded BatchNorm) of the checkpoint: what
 * pytorch_detector.py:957  checkpoint['model'].float().fuse()  leaves in each Conv module. */
typedef struct {
    const float* weight;   /* host, fp32, OIHW: [c_out][c_in][kh][kw] */
    const float* bias;     /* host, fp32, [c_out]                      */
    int32_t c_out, c_in, kh, kw;
} mdhip_conv;

/* One row of the model (same granularity as model.model[i] in the reference's checkpoint).
 * from[] holds absolute layer indices (-1 = the network input).
 * Conv:    convs[first_conv]                        k,s,p as in the module
 * C3:      cv1, cv2, cv3, then (m[j].cv1, m[j].cv2) for j < n        -> 3 + 2n convs
 * SPPF:    cv1, cv2 ; k = pool size
 * Detect:  one 1x1 conv per input level (bias, no activation)        -> n_from convs */
typedef struct {
    int32_t type;
    int32_t n_from;
    int32_t from[4];
    int32_t c_out;
    int32_t k, s, p;
    int32_t n;
    int32_t shortcut;
    int32_t first_conv;
} mdhip_layer;

typedef struct {
    int32_t n_layers;
    const mdhip_layer* layers;
    int32_t n_convs;
    const mdhip_conv* convs;
    int32_t nc;                 /* classes (3 for MDv5)                         */
    int32_t na;                 /* anchors per level (3)                        */
    int32_t nl;                 /* detection levels (4 for YOLOv5x6)            */
    const float* anchors_px;    /* host, [nl][na][2] = Detect.anchors * stride  */
    const float* strides;       /* host, [nl]                                   */
} mdhip_model;

/* Letterbox geometry of one image, computed on the host exactly as
 * yolov5 letterbox() does (restated at pytorch_detector.py:434-454). */
typedef struct {
    int32_t src_h, src_w;           /* original image                                   */
    int32_t resized_h, resized_w;   /* new_unpad: size after cv2.resize                  */
    int32_t top, left;              /* border offsets (copyMakeBorder, value 114)       */
    int32_t interp;                 /* 0 = cv2.INTER_LINEAR (yolov5 letterbox; 'modern' growing),
                                     * 1 = cv2.INTER_AREA ('modern' shrinking, pytorch_detector.py:1048-1062) */
} mdhip_letterbox;

typedef struct mdhip_ctx mdhip_ctx;

/* Replaces PTDetector.__init__/_load_model (pytorch_detector.py:745-959): packs weights to
 * the MFMA operand layout, plans and allocates every activation buffer for up to
 * max_batch images of max_h x max_w letterboxed pixels on GPU `device`. */
int mdhip_create(const mdhip_model* model, int device, int dtype,
                 int max_batch, int max_h, int max_w, mdhip_ctx** out);
void mdhip_destroy(mdhip_ctx* ctx);
const char* mdhip_last_error(mdhip_ctx* ctx);

/* Replaces letterbox() + HWC->CHW + float() + /255 (pytorch_detector.py:1104-1109,
 * :1283-1310).  images[i]: HWC uint8 RGB, src_h x src_w, host or device memory
 * (host images are copied to a device staging area first).  Output: the context's network input,
 * n x out_h x out_w.  out_h/out_w must be multiples of the model's largest stride.
 * Device images: the kernels read whole aligned dwords, so for an image pointer (or row pitch src_w * 3) that is not a
 * multiple of 4 up to 3 bytes in front of the first and behind the last pixel are READ (never used, never written): they
 * lie in the same aligned dword as an image byte, i.e. inside any hipMalloc'ed block that holds the image, but a
 * memory checker that tracks exact extents will report them.  (MDHIP_LETTERBOX_GENERAL in the environment at
 * mdhip_create selects the byte-wise kernel for every batch.)
 * Streams: the call may be enqueued on another stream than the forwards -- it first makes its stream wait
 * (hipStreamWaitEvent, inside the library) for the last mdhip_forward / mdhip_forward_tta enqueued before it to have read
 * the network input (the stem), so the letterbox of batch i + 1 can run next to the rest of forward i; the forward of
 * batch i + 1 must then be ordered behind this call by the caller (an event), as bench.py and the detector do. */
int mdhip_preprocess(mdhip_ctx* ctx, const uint8_t* const* images, const mdhip_letterbox* geom,
                     int n, int out_h, int out_w, void* hip_stream);

/* Replaces self.model(batch)[0] (pytorch_detector.py:1313): conv stack + Detect decode.
 * Leaves (n, n_anchors, 5+nc) fp32 predictions in a device buffer of the context. */
int mdhip_forward(mdhip_ctx* ctx, int n, int h, int w, void* hip_stream);

/* Test-time augmentation: replaces mdhip_forward for `model(batch, augment=True)` (reference
 * pytorch_detector.py:1313 -> yolov5 _forward_augment): three passes over the batch that mdhip_preprocess
 * left in the context -- scale 1, scale 0.83 left-right flipped, scale 0.67 (bilinear, padded with 0.447 to
 * the model stride) -- boxes de-scaled and un-flipped, the coarsest level of the first and the finest level of
 * the last pass dropped, predictions concatenated.  mdhip_nms / mdhip_read_predictions then work on
 * mdhip_last_num_anchors(ctx) anchors per image. */
int mdhip_forward_tta(mdhip_ctx* ctx, int n, int h, int w, void* hip_stream);
/* anchors per image of the prediction the context currently holds (last forward, augmented forward or mdhip_nms_on) */
int mdhip_last_num_anchors(mdhip_ctx* ctx);

/* MDHIP_DTYPE_FP8 contexts (BASELINE.json configs[4]; the reference has no reduced precision at all --
 * pytorch_detector.py:848 hard-wires half_precision = False -- so there is no upstream call this replaces).
 * Activations and weights stay bf16 in HBM except the hidden tensor of every C3 bottleneck: its 1x1 conv writes it
 * as OCP e4m3 with one scale per tensor, its 3x3 conv runs on e4m3 operands (weights quantised per output channel at
 * mdhip_create) with the block-scaled K = 128 MFMA and fp32 accumulation.
 * mdhip_calibrate runs the batch left by mdhip_preprocess once in bf16, records the largest magnitude of every such
 * tensor and derives the scales (2x head-room); repeated calls accumulate the ranges.  mdhip_forward fails with
 * MDHIP_EINVAL until the context has scales (from mdhip_calibrate or mdhip_fp8_set_scales).  Results of an image do
 * not depend on the batch it travels in; they do depend on the calibration data.
 * mdhip_fp8_get_scales: returns the number of e4m3 tensors and fills up to max_n entries (any pointer may be NULL):
 * the scale, the model layer (C3 index) and the op index of the producing 1x1 conv, in execution order.
 * mdhip_fp8_set_scales: installs scales saved from an earlier calibration (same model, same order). */
int mdhip_calibrate(mdhip_ctx* ctx, int n, int h, int w, void* hip_stream);
int mdhip_fp8_num_tensors(mdhip_ctx* ctx);
int mdhip_fp8_get_scales(mdhip_ctx* ctx, float* scales, int32_t* layers, int32_t* ops, int max_n);
int mdhip_fp8_set_scales(mdhip_ctx* ctx, const float* scales, int n);
/* the host-side weight quantiser of the fp8 mode (OCP e4m3, round to nearest even, saturating at +-448); exported so
 * that the CPU test-suite can pin it against torch.float8_e4m3fn without a GPU */
int mdhip_f32_to_e4m3(const float* in, uint8_t* out, int n);

/* Replaces nms() (pytorch_detector.py:502-610) on the predictions of the last forward.
 * out: host, [n][max_det][6] = x1,y1,x2,y2,conf,cls in letterboxed pixels, sorted by
 * confidence (descending; ties by anchor index); counts: host, [n].  Blocks until the
 * results are in host memory. */
int mdhip_nms(mdhip_ctx* ctx, int n, float conf_thres, float iou_thres, int max_det,
              float* out, int32_t* counts, void* hip_stream);

/* Asynchronous form of mdhip_nms, for overlapping the host-side formatting of batch i
 * (pytorch_detector.py:1361-1422) with the GPU work of batch i+1: the kernel and the D2H copies
 * are enqueued on the stream and the call returns; results land in pinned host slot `slot`
 * (0 <= slot < MDHIP_NMS_SLOTS) owned by the context.  mdhip_nms_wait blocks until that slot is
 * complete and returns pointers into it ([n][max_det][6] floats, [n] counts), valid until the
 * slot is enqueued again. 
 * Every forward writes the other of two prediction buffers, so mdhip_nms_enqueue may run on another stream
 * than the forward of the next batch.  Ordering contract: the CALLER orders the enqueue behind its own forward (an
 * event recorded after mdhip_forward, waited for on the NMS stream); the LIBRARY orders the forward after next behind
 * the enqueue -- mdhip_nms_enqueue records an event for the prediction buffer it reads, and the forward that is about
 * to overwrite that buffer (mdhip_forward / mdhip_forward_tta, any stream) waits for it.  The scratch buffers of the
 * NMS kernels are shared: keep all mdhip_nms* calls of one context on ONE stream. */
#define MDHIP_NMS_SLOTS 4
int mdhip_nms_enqueue(mdhip_ctx* ctx, int n, float conf_thres, float iou_thres, int max_det,
                      int slot, void* hip_stream);
int mdhip_nms_wait(mdhip_ctx* ctx, int slot, const float** out, const int32_t** counts);

/* nms() on caller-supplied predictions (host, [n][n_anchors][5+nc] fp32); any n_anchors up to
 * the context's capacity.  Used by the parity tests against the reference's NMS vectors. */
int mdhip_nms_on(mdhip_ctx* ctx, const float* pred, int n, int n_anchors, float conf_thres,
                 float iou_thres, int max_det, float* out, int32_t* counts, void* hip_stream);

/* ---- introspection / measurement (not on the product path) ---- */

int mdhip_num_anchors(mdhip_ctx* ctx, int h, int w);                 /* anchors per image  */
int mdhip_max_stride(mdhip_ctx* ctx);
/* copy the raw predictions of the last forward to host: [n][n_anchors][5+nc] fp32 */
int mdhip_read_predictions(mdhip_ctx* ctx, int n, float* out, void* hip_stream);
/* copy the network input of the last preprocess to host as [n][3][h][w] fp32 (in [0,1]) */
int mdhip_read_input(mdhip_ctx* ctx, int n, int h, int w, float* out, void* hip_stream);
/* copy the output of model layer `layer` (last forward) to host as NCHW fp32; returns
 * c,h,w through the out-params; out may be NULL to query the shape only */
int mdhip_read_layer(mdhip_ctx* ctx, int layer, int n, float* out, int* c, int* h, int* w,
                     void* hip_stream);

typedef struct {
    char    name[48];       /* e.g. "L6.m3.cv2 3x3"                           */
    int32_t kind;           /* 0 conv (implicit GEMM), 1 pool, 2 upsample, 3 decode, 4 copy */
    int32_t layer;          /* model layer index                              */
    int32_t m, n, k;        /* GEMM view of a conv (per call, for the last n,h,w) */
    double  flops;          /* algorithmic FLOPs of the op for the last (n,h,w)   */
    double  bytes;          /* algorithmic HBM bytes (read input once + write output once + weights) */
    int32_t cfg;            /* tile configuration chosen (a decode op: -2 = done in
                             * the epilogue of the conv in front, -1 = own launch) */
    int32_t ntaps, stride, has_res;   /* conv: kh*kw of the packed kernel, stride, residual added */
} mdhip_op_info;

int mdhip_num_ops(mdhip_ctx* ctx);
int mdhip_get_op_info(mdhip_ctx* ctx, int op, mdhip_op_info* out);
/* run the forward with a hipEvent pair around every op; ms[op] = duration in milliseconds */
int mdhip_forward_timed(mdhip_ctx* ctx, int n, int h, int w, float* ms, void* hip_stream);
/* live measurement for bench.py: with enable != 0 every mdhip_forward is bracketed by a hipEvent pair
 * recorded on the stream it is launched on (a ring of the 64 most recent forwards);
 * mdhip_forward_times waits for and returns the durations (ms) of the most recent min(max_n, 64,
 * forwards since enabling) forwards, oldest first, and returns how many it wrote (or a negative code). */
int mdhip_time_forwards(mdhip_ctx* ctx, int enable);
int mdhip_forward_times(mdhip_ctx* ctx, float* ms, int max_n);
/* force tile configuration `cfg` for op `op` (-1 = automatic choice); returns MDHIP_EINVAL
 * when cfg does not fit the op.  mdhip_num_conv_cfgs() = number of configurations. */
int mdhip_set_op_cfg(mdhip_ctx* ctx, int op, int cfg);
int mdhip_num_conv_cfgs(void);
/* 1 when tile configuration `cfg` can run conv op `op` (the first-generation configurations run every
 * op; the later main loops need C_in >= 64 or 32, kernels up to 3x3), 0 when not, negative on bad arguments */
int mdhip_op_supports_cfg(mdhip_ctx* ctx, int op, int cfg);
/* 1 when configuration `cfg` accumulates K in the canonical (r, s, c) order, i.e. produces bit-identical
 * results to every other such configuration; 0 for the row-patch kernel, whose K order is
 * (channel group, r, s, c) and whose results agree to fp32 summation-order rounding only */
int mdhip_cfg_is_bitwise(int cfg);
/* human-readable name of a tile configuration ("v2:160x160/2x2", ...); "" when out of range */
const char* mdhip_conv_cfg_name(int cfg);
/* measured tile choices (tools/autotune.py -> megadetector_amd/tuned_cfgs.json): a conv runs the
 * configuration of the entry with the same layer geometry (N, K, taps, stride, residual) whose
 * PER-IMAGE M (= m / batch = Ho*Wo) is equal or, failing that, nearest within a factor 4 (the same
 * layer at another image shape); everything else uses the built-in heuristic.  The choice does not
 * depend on the batch size of the call, so an image's result is bit-identical whatever batch it is
 * part of.  A configuration that does not support the op falls back to the heuristic choice. */
typedef struct {
    int32_t m, n, k;        /* GEMM view at measurement: M = batch*Ho*Wo, N = C_out, K = kh*kw*C_in */
    int32_t ntaps, stride;  /* kh*kw, conv stride */
    int32_t has_res;        /* 1 when the op adds a residual */
    int32_t cfg;
    int32_t batch;          /* batch size the entry was measured at (<= 0: taken as 32) */
} mdhip_tuned;
int mdhip_set_tuned(mdhip_ctx* ctx, const mdhip_tuned* entries, int n);
/* Fused bottlenecks (default on): a C3 block whose 3x3 convs all resolve to a strip configuration (the 80-channel block
 * of the x6 stack at batch >= 2) runs every bottleneck -- 1x1, SiLU, 3x3, SiLU, residual -- as ONE launch that keeps the
 * hidden tensor on chip; and a 1x1 conv behind Upsample + Concat reads the low-resolution tensor in place instead of its
 * 4x copy (the upsample launch is skipped).  Same arithmetic and summation order: bit-identical results.  on = 0 runs
 * the separate launches (A/B measurements, tests). */
int mdhip_set_fuse(mdhip_ctx* ctx, int on);
/* Graph replay of mdhip_forward (default off): the launches of one forward -- ~160 kernels, which at batch 1 .. 4 do a few
 * microseconds of work each -- are captured once per (batch, height, width) on an internal stream and replayed with one
 * hipGraphLaunch on the caller's stream.  mode 0 = off, 1 = every forward, 2 = forwards of at most max_n images (max_n <= 0
 * keeps the current bound, 8).  The first forward of a shape always runs eagerly; calls that change what a forward
 * launches (mdhip_set_tuned, mdhip_set_op_cfg, mdhip_set_fuse, fp8 scales) drop the captured graphs.  Same kernels, same
 * arguments: bit-identical results.  mdhip_forward_tta and the timed / per-op entry points are not replayed.
 * Replaces nothing in the reference (pytorch_detector.py:1313 runs eager PyTorch); this is launch plumbing. */
int mdhip_set_graph(mdhip_ctx* ctx, int mode, int max_n);
/* Named integer switches of a context (returns MDHIP_EINVAL for an unknown name; every change drops the captured graphs):
 *   "letterbox_general" 0 | 1   1 = mdhip_preprocess never takes the streaming-copy kernel (A/B measurements, tests;
 *                               the environment variable MDHIP_LETTERBOX_GENERAL at mdhip_create sets the same switch)
 *   "fuse_decode"       1 | 0   1 (default) = the Detect decode (yolov5 Detect.forward, pytorch_detector.py:1313) runs in the
 *                               epilogue of each level's 1x1 conv -- no fp32 logits tensor, four launches fewer -- for heads
 *                               with 8 outputs per anchor (MDv5: nc = 3) in the plain forward; the augmented forward and
 *                               other heads always use the separate decode kernel.  Same statements, bit-identical.
 * Replaces nothing in the reference; the switches exist for measurements and tests. */
int mdhip_set_option(mdhip_ctx* ctx, const char* name, int value);
/* time one op in isolation: `iters` back-to-back launches bracketed by events */
int mdhip_time_op(mdhip_ctx* ctx, int op, int n, int h, int w, int iters, float* ms_avg,
                  void* hip_stream);

const char* mdhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MDHIP_H */
