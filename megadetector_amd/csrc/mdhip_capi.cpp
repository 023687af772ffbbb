// C-ABI layer (include/mdhip.h): model planning, weight packing, buffer arena, executor.
//
// This is the native runtime under the Python detector seam
// (reference megadetector/detection/pytorch_detector.py:739 PTDetector):
//   mdhip_create      <- PTDetector.__init__/_load_model            (:745-959)
//   mdhip_preprocess  <- letterbox + tensor prep                    (:1104-1109, :1283-1310)
//   mdhip_forward     <- self.model(batch)[0]                       (:1313)
//   mdhip_nms         <- nms()                                      (:502-610, :1342)
//
// Planning turns the YOLOv5 module list into a flat list of ops over channel-strided NHWC
// bf16 views of one device arena:
//   * Concat never copies: producers write straight into their channel slice of the consumer's
//     buffer (a copy op is emitted only for a producer that already lives elsewhere).
//   * C3:  cv1 and cv2 read the same input -> ONE implicit GEMM with the two weight sets
//     stacked along N writes the [m-branch | cv2] concat buffer; the bottleneck chain then
//     updates the first half in place (1x1 -> scratch, 3x3 (+residual) -> slice).
//   * the 6x6/s2 stem runs as a 3x3/s1 conv over the space-to-depth input the letterbox
//     kernel produces.
//   * Detect: per level a 1x1 implicit GEMM with fp32 output followed by the decode kernel.

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mdhip.h"
#include "mdhip_internal.h"

using namespace mdhip;

namespace {

thread_local std::string g_create_error;

struct Tensor {
    size_t off = 0;   // byte offset into the arena
    int ld = 0;       // elements between consecutive pixels
    int c = 0;        // channels of the view
    int div = 1;      // spatial size = network input / div
    bool valid = false;
};

struct PackedConv {
    size_t w_off = 0, b_off = 0;     // byte offsets into the weight arena
    size_t w4_off = 0;               // second packing for the row-patch kernel (0 = none)
    int k_pad4 = 0, groups = 0;
    size_t w4p_off = 0;              // the same with the half-full last group's taps paired (conv_v5.cpp; 0 = none)
    int k_pad4p = 0;
    int n_rows = 0, k_pad = 0, cin_pad = 0, kh = 0, kw = 0, c_out = 0, k_real = 0;
    // fp8 form (MDHIP_DTYPE_FP8, 3x3 / stride-1 bottleneck convs): e4m3 weights [n_rows][groups8*9*128], quantised per
    // output channel (wscale[n] = max_k |w[n][k]| / 448); scale_off = device array of n_rows floats holding
    // activation scale x wscale[n], written by mdhip_calibrate / mdhip_fp8_set_scales
    size_t w8_off = 0, scale_off = 0;
    int k_pad8 = 0, groups8 = 0;
    std::vector<float> wscale;
};

enum OpKind { OP_CONV = 0, OP_POOL = 1, OP_UPSAMPLE = 2, OP_DECODE = 3, OP_COPY = 4 };

struct Op {
    int kind = OP_CONV;
    int layer = -1;
    std::string name;
    Tensor in, out, res;
    bool has_res = false;
    int pc = -1;
    int stride = 1, pad = 0, act = 1, out_f32 = 0;
    int pool_k = 5;
    int level = 0;            // decode
    size_t f32_off = 0;       // decode: logits buffer offset ; conv with out_f32: same
    int f32_ld = 0;
    int forced_cfg = -1;
    // fp8 mode: this op writes (f8_out) / reads (f8_in) an e4m3 tensor; f8_peer = the op at the other end of it;
    // act_scale = the tensor's scale (value = e4m3 x act_scale), 0 until calibrated; amax = largest |x| seen
    bool f8_out = false, f8_in = false;
    int f8_peer = -1;
    float act_scale = 0.f, amax = 0.f;
    // fused bottleneck (conv_v5c.cpp): fuse_role 1 = the 1x1 of bottleneck fuse_idx of C3 block fuse_group, 2 = its 3x3
    int fuse_group = -1, fuse_idx = -1, fuse_role = 0;
    double pre_flops = 0;
    // upsample read in place (conv_v2.cpp): an OP_UPSAMPLE whose only reader is the 1x1 conv `up_peer` (and vice versa)
    int up_peer = -1;
    // Detect: the 1x1 conv of a level and its OP_DECODE (the next op); dec_done = the conv of THIS forward decoded in its
    // epilogue, the decode op has nothing left to launch
    bool dec_done = false;
    size_t amax_off = 0;
    // the configuration chosen for the last (n, h, w): the table walk is not repeated on every launch
    int memo_n = 0, memo_h = 0, memo_w = 0, memo_cfg = -1;
    bool memo_from_table = false;
    int last_cfg = -1;
    // stats for the last (n,h,w)
    int gm = 0, gn = 0, gk = 0;
    double flops = 0, bytes = 0;
};

}  // namespace

struct mdhip_ctx {
    int device = 0;
    int dtype = 0;
    int max_batch = 0, max_h = 0, max_w = 0;
    int nc = 0, na = 0, nl = 0, no = 0;
    std::vector<float> strides;
    int max_stride = 0;
    std::vector<mdhip_layer> layers;
    std::vector<Tensor> layer_out;
    std::vector<PackedConv> packed;
    std::vector<Op> ops;
    Tensor input;                 // space-to-depth network input (16 channels, div 2)
    Tensor input_orig;            // copy of it during test-time augmentation (the scaled passes overwrite `input`)
    DecodeTta cur_tta;            // how the Detect decode of the running pass places its anchors
    int cur_A = 0;                // anchors per image of the prediction being written (row pitch of `pred`)
    int last_A = 0;               // anchors per image of the last forward (plain or augmented)
    int a_cap = 0;                // capacity of `pred` and of the NMS scratch, anchors per image
    size_t arena_bytes = 0;
    char* arena = nullptr;
    char* warena = nullptr;       // packed weights + biases + zero page + anchors
    size_t warena_bytes = 0;
    size_t zero_off = 0, anchors_off = 0;
    // fp32 predictions [max_batch][a_cap][no], two of them: every forward writes the other one, so that the NMS of
    // batch i (on its own stream) may still read its predictions while the forward of batch i+1 runs
    size_t pred_offs[2] = {0, 0};
    int pred_cur = 0;
    size_t pred_off = 0;          // = pred_offs[pred_cur]: the prediction of the last forward
    int a_max = 0;
    NmsScratch nms_scr{};
    size_t nms_out_off = 0, nms_cnt_off = 0;
    size_t geom_off = 0;
    char* stage = nullptr;        // device staging for host images
    size_t stage_bytes = 0;
    int last_n = 0, last_h = 0, last_w = 0;
    std::string err;
    // fp8 mode: until every e4m3 tensor has a scale (mdhip_calibrate / mdhip_fp8_set_scales) the forward refuses
    // to run; `calibrating` makes run_op execute every op in 16 bits and record the range of the tensors
    bool calibrated = false, calibrating = false;
    int n_f8 = 0;
    // C3 blocks whose bottlenecks can run as one launch each (1x1 -> LDS -> 3x3): op indices of the 3x3s per block
    std::vector<std::vector<int>> fuse_groups;
    bool fuse_enabled = true, fuse_suspended = false;
    bool pair_enabled = true;         // paired taps of a half-full last channel group (conv_v5.cpp); MDHIP_PAIR=0 at create: off
    bool fuse_decode = true;          // Detect decode in the epilogue of the Detect 1x1 convs (mdhip_set_option "fuse_decode")
    bool letterbox_general = false;   // MDHIP_LETTERBOX_GENERAL at create: never take the streaming-copy letterbox (A/B, tests)
    std::vector<hipEvent_t> events;
    std::vector<mdhip_tuned> tuned;   // measured tile choices (tools/autotune.py)
    // optional event pair around every mdhip_forward (bench.py's live roofline measurement)
    static constexpr int kFwdRing = 64;
    bool time_forward = false;
    hipEvent_t fwd_ev[kFwdRing][2] = {};
    long long fwd_count = 0;
    // pinned host staging: letterbox geometry ring + asynchronous NMS result slots
    LetterboxDev* geom_host = nullptr;
    int geom_slot = 0;
    hipEvent_t geom_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float* nms_host_out[MDHIP_NMS_SLOTS] = {};
    int32_t* nms_host_cnt[MDHIP_NMS_SLOTS] = {};
    hipEvent_t nms_ev[MDHIP_NMS_SLOTS] = {};
    int nms_slot_n[MDHIP_NMS_SLOTS] = {};
    // mdhip_set_graph: the op sequence of a forward captured once per (batch, height, width, prediction buffer) and
    // replayed with one hipGraphLaunch (small batches are bound by ~160 launches of a few microseconds of work each)
    int graph_mode = 0;                                   // 0 = off, 1 = on, 2 = on for batches <= graph_max_n
    int graph_max_n = 8;
    hipStream_t capture_stream = nullptr;
    // `disabled`: capture or instantiation failed once for this shape -- it runs eagerly from then on; `last_use`: LRU stamp
    struct GraphSlot { hipGraphExec_t exec = nullptr; int seen = 0; bool disabled = false; long long last_use = 0; };
    std::map<std::tuple<int, int, int, int>, GraphSlot> graphs;
    static constexpr int kMaxGraphs = 32;                 // cached executables (letterbox shapes x batch sizes x 2 buffers)
    long long graph_clock = 0;
    // recorded on the forward's stream behind the last op that reads the network input (the stem, op 0): a following
    // mdhip_preprocess -- possibly on ANOTHER stream, next to the rest of this forward -- waits for it before it overwrites
    // the input tensor
    hipEvent_t input_free = nullptr;
    bool input_free_valid = false;
    // the NMS that reads prediction buffer k (possibly on another stream: mdhip_nms_enqueue) records pred_read[k]; the
    // forward that is about to overwrite buffer k waits for it -- the ordering is the library's, not the caller's
    hipEvent_t pred_read[2] = {nullptr, nullptr};
    bool pred_read_valid[2] = {false, false};
};

namespace {

int fail(mdhip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

// every call that changes what a forward launches drops the captured graphs (after the device has finished with them: an
// executable may still be in flight on the caller's stream)
void drop_graphs(mdhip_ctx* ctx) {
    bool any = false;
    for (auto& kv : ctx->graphs) any |= kv.second.exec != nullptr;
    if (any) {
        (void)hipSetDevice(ctx->device);            // (the setters reach here without it: synchronise OUR device)
        (void)hipDeviceSynchronize();
    }
    for (auto& kv : ctx->graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    ctx->graphs.clear();
}

// room for one more cached graph: the least recently used executable goes (its stream work is waited for first)
void evict_graph_if_full(mdhip_ctx* ctx) {
    int live = 0;
    for (auto& kv : ctx->graphs) live += kv.second.exec != nullptr;
    if (live < mdhip_ctx::kMaxGraphs) return;
    auto victim = ctx->graphs.end();
    for (auto it = ctx->graphs.begin(); it != ctx->graphs.end(); ++it)
        if (it->second.exec && (victim == ctx->graphs.end() || it->second.last_use < victim->second.last_use)) victim = it;
    if (victim == ctx->graphs.end()) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    (void)hipGraphExecDestroy(victim->second.exec);
    ctx->graphs.erase(victim);
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(ctx, MDHIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------------------
struct Planner {
    mdhip_ctx* ctx;
    const mdhip_model* model;
    size_t cursor = 0;
    std::vector<std::vector<uint16_t>> w_host;   // packed weights per PackedConv
    std::vector<std::vector<uint16_t>> w4_host;  // row-patch packing (empty when not applicable)
    std::vector<std::vector<uint16_t>> w4p_host; // ... with the last group's taps paired (empty when not applicable)
    std::vector<std::vector<float>> b_host;
    std::vector<std::vector<uint8_t>> w8_host;   // e4m3 packing (empty when the conv has no fp8 form)
    std::vector<int> layer_c, layer_div;
    std::vector<int> concat_target, concat_choff;   // per producer layer
    std::vector<Tensor> concat_buf;                  // per concat layer
    std::string error;

    // ld >= c: pixel pitch in elements (a pitch that is a multiple of 64 keeps every 128-byte K-slab row of
    // a pixel inside one cache line; the pad channels are never read or written)
    Tensor alloc(int c, int div, int ld = 0) {
        Tensor t;
        t.off = cursor;
        t.ld = ld > c ? ld : c;
        t.c = c;
        t.div = div;
        t.valid = true;
        const size_t px = (size_t)ctx->max_batch * (ctx->max_h / div) * (ctx->max_w / div);
        cursor = align_up(cursor + px * t.ld * 2, 256);
        return t;
    }
    size_t alloc_bytes(size_t bytes) {
        const size_t off = cursor;
        cursor = align_up(cursor + bytes, 256);
        return off;
    }
    static Tensor slice(const Tensor& t, int ch_off, int c) {
        Tensor s = t;
        s.off = t.off + (size_t)ch_off * 2;
        s.c = c;
        return s;
    }

    // pack one or more OIHW fp32 convs (stacked along N) to bf16 / fp16 [n_rows][k_pad], k = (r,s,c)
    int pack(const std::vector<const mdhip_conv*>& cs, bool s2d_stem) {
        PackedConv pc;
        const int f16 = ctx->dtype == MDHIP_DTYPE_FP16;
        const mdhip_conv* c0 = cs[0];
        int c_out = 0;
        for (auto* c : cs) c_out += c->c_out;
        if (s2d_stem) {
            pc.kh = pc.kw = 3;
            pc.cin_pad = 16;
            pc.k_real = 6 * 6 * 3;
        } else {
            pc.kh = c0->kh;
            pc.kw = c0->kw;
            pc.cin_pad = round_up(c0->c_in, 8);
            pc.k_real = c0->kh * c0->kw * c0->c_in;
        }
        pc.c_out = c_out;
        pc.n_rows = round_up(c_out, 16);
        pc.k_pad = round_up(pc.kh * pc.kw * pc.cin_pad, 64);
        std::vector<uint16_t> w((size_t)pc.n_rows * pc.k_pad, 0);
        std::vector<float> b(pc.n_rows, 0.f);
        int row0 = 0;
        for (auto* c : cs) {
            for (int o = 0; o < c->c_out; ++o) {
                uint16_t* dst = &w[(size_t)(row0 + o) * pc.k_pad];
                b[row0 + o] = c->bias ? c->bias[o] : 0.f;
                if (s2d_stem) {
                    // w6[o][c][6][6] -> w3[o][r'][s'][(dy*2+dx)*3 + c], 6x6 index = 2*r'+dy
                    for (int ci = 0; ci < 3; ++ci)
                        for (int r = 0; r < 6; ++r)
                            for (int s = 0; s < 6; ++s) {
                                const float v = c->weight[(((size_t)o * 3 + ci) * 6 + r) * 6 + s];
                                const int rp = r >> 1, dy = r & 1, sp = s >> 1, dx = s & 1;
                                dst[(rp * 3 + sp) * 16 + (dy * 2 + dx) * 3 + ci] = f32_to_st(v, f16);
                            }
                } else {
                    for (int ci = 0; ci < c->c_in; ++ci)
                        for (int r = 0; r < c->kh; ++r)
                            for (int s = 0; s < c->kw; ++s) {
                                const float v =
                                    c->weight[(((size_t)o * c->c_in + ci) * c->kh + r) * c->kw + s];
                                dst[(r * c->kw + s) * pc.cin_pad + ci] = f32_to_st(v, f16);
                            }
                }
            }
            row0 += c->c_out;
        }
        // 3x3 convs with at least 64 input channels also get the row-patch order:
        // k = (channel group of 64, tap, channel in group), every (group, tap) slab 64 wide (zero padded)
        std::vector<uint16_t> w4;
        if (!s2d_stem && pc.kh == 3 && pc.kw == 3 && pc.cin_pad >= 64) {
            pc.groups = (pc.cin_pad + 63) / 64;
            pc.k_pad4 = pc.groups * 9 * 64;
            w4.assign((size_t)pc.n_rows * pc.k_pad4, 0);
            for (int o = 0; o < pc.n_rows; ++o)
                for (int t = 0; t < 9; ++t)
                    for (int ci = 0; ci < pc.cin_pad; ++ci)
                        w4[(size_t)o * pc.k_pad4 + ((ci / 64) * 9 + t) * 64 + (ci % 64)] =
                            w[(size_t)o * pc.k_pad + t * pc.cin_pad + ci];
        }
        // a last group of at most 32 channels: the paired packing of conv_v5.cpp -- groups 0 .. G-2 as above; last group:
        // per kernel row r the slabs [ tap (r,0) ch 0..31 | tap (r,1) ch 0..31 ] and [ tap (r,2) ch 0..31 | zeros ]
        std::vector<uint16_t> w4p;
        if (!w4.empty() && (pc.cin_pad % 64) != 0 && (pc.cin_pad % 64) <= 32) {
            const int G = pc.groups, tail = pc.cin_pad % 64;
            pc.k_pad4p = (9 * (G - 1) + 6) * 64;
            w4p.assign((size_t)pc.n_rows * pc.k_pad4p, 0);
            for (int o = 0; o < pc.n_rows; ++o) {
                const uint16_t* src = &w4[(size_t)o * pc.k_pad4];
                uint16_t* dst = &w4p[(size_t)o * pc.k_pad4p];
                std::copy(src, src + (size_t)9 * (G - 1) * 64, dst);
                for (int r = 0; r < 3; ++r)
                    for (int sx = 0; sx < 3; ++sx)
                        for (int ci = 0; ci < tail; ++ci)
                            dst[((G - 1) * 9 + 2 * r + (sx == 2 ? 1 : 0)) * 64 + (sx == 1 ? 32 : 0) + ci] =
                                src[((G - 1) * 9 + r * 3 + sx) * 64 + ci];
            }
        }
        // fp8 mode: 3x3 convs whose input channel count is a multiple of 16 also get the e4m3 packing of
        // conv_f8.cpp: k = (channel group of 128, tap, channel in group), quantised from the fp32 weights
        std::vector<uint8_t> w8;
        if (ctx->dtype == MDHIP_DTYPE_FP8 && !s2d_stem && cs.size() == 1 && pc.kh == 3 && pc.kw == 3 && (c0->c_in % 16) == 0) {
            pc.groups8 = (c0->c_in + 127) / 128;
            pc.k_pad8 = pc.groups8 * 9 * 128;
            pc.wscale.assign(pc.n_rows, 1.0f);
            w8.assign((size_t)pc.n_rows * pc.k_pad8, 0);
            for (int o = 0; o < c0->c_out; ++o) {
                const float* wo = c0->weight + (size_t)o * c0->c_in * 9;
                float amax = 0.f;
                for (int k = 0; k < c0->c_in * 9; ++k) amax = std::max(amax, std::fabs(wo[k]));
                const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
                pc.wscale[o] = sc;
                for (int ci = 0; ci < c0->c_in; ++ci)
                    for (int t = 0; t < 9; ++t)
                        w8[(size_t)o * pc.k_pad8 + ((ci / 128) * 9 + t) * 128 + (ci % 128)] = f32_to_e4m3(wo[ci * 9 + t] / sc);
            }
        }
        ctx->packed.push_back(pc);
        w_host.push_back(std::move(w));
        w4_host.push_back(std::move(w4));
        w4p_host.push_back(std::move(w4p));
        b_host.push_back(std::move(b));
        w8_host.push_back(std::move(w8));
        return (int)ctx->packed.size() - 1;
    }

    void add_conv(int layer, const std::string& name, const Tensor& in, const Tensor& out, int pc,
                  int stride, int pad, bool act, const Tensor* res) {
        Op op;
        op.kind = OP_CONV;
        op.layer = layer;
        op.name = name;
        op.in = in;
        op.out = out;
        op.pc = pc;
        op.stride = stride;
        op.pad = pad;
        op.act = act ? 1 : 0;
        if (res) { op.res = *res; op.has_res = true; }
        ctx->ops.push_back(op);
    }

    bool plan() {
        const int nL = model->n_layers;
        layer_c.assign(nL, 0);
        layer_div.assign(nL, 1);
        concat_target.assign(nL, -1);
        concat_choff.assign(nL, 0);
        concat_buf.assign(nL, Tensor());
        ctx->layer_out.assign(nL, Tensor());
        char nm[96];

        // pass 1: channels / divisors / concat targets
        for (int i = 0; i < nL; ++i) {
            const mdhip_layer& L = model->layers[i];
            for (int j = 0; j < L.n_from; ++j)
                if (L.from[j] >= i || L.from[j] < -1) { error = "layer 'from' index out of order"; return false; }
            const int f0 = L.n_from > 0 ? L.from[0] : -1;
            const int in_div = f0 < 0 ? 1 : layer_div[f0];
            switch (L.type) {
                case MDHIP_CONV:
                    layer_c[i] = L.c_out;
                    layer_div[i] = in_div * L.s;
                    break;
                case MDHIP_C3:
                case MDHIP_SPPF:
                    layer_c[i] = L.c_out;
                    layer_div[i] = in_div;
                    break;
                case MDHIP_UPSAMPLE:
                    if (f0 < 0 || in_div % 2) { error = "bad upsample input"; return false; }
                    layer_c[i] = layer_c[f0];
                    layer_div[i] = in_div / 2;
                    break;
                case MDHIP_CONCAT: {
                    int c = 0;
                    for (int j = 0; j < L.n_from; ++j) {
                        const int f = L.from[j];
                        if (f < 0 || layer_div[f] != in_div) { error = "concat inputs differ in size"; return false; }
                        if (concat_target[f] < 0) { concat_target[f] = i; concat_choff[f] = c; }
                        c += layer_c[f];
                    }
                    layer_c[i] = c;
                    layer_div[i] = in_div;
                    break;
                }
                case MDHIP_DETECT:
                    break;
                default:
                    error = "unknown layer type";
                    return false;
            }
            if (L.type != MDHIP_DETECT && L.type != MDHIP_CONCAT && (layer_c[i] % 8)) {
                error = "channel counts must be multiples of 8";
                return false;
            }
        }

        // network input (space-to-depth, 16 channels)
        ctx->input = alloc(16, 2);
        ctx->input_orig = alloc(16, 2);

        auto out_view = [&](int i) -> Tensor {
            const int tgt = concat_target[i];
            if (tgt >= 0) {
                if (!concat_buf[tgt].valid) concat_buf[tgt] = alloc(layer_c[tgt], layer_div[tgt]);
                return slice(concat_buf[tgt], concat_choff[i], layer_c[i]);
            }
            return alloc(layer_c[i], layer_div[i]);
        };

        // pass 2: ops
        for (int i = 0; i < nL; ++i) {
            const mdhip_layer& L = model->layers[i];
            const int f0 = L.n_from > 0 ? L.from[0] : -1;
            if (f0 < 0 && !(L.type == MDHIP_CONV && i == 0)) { error = "only the stem conv (layer 0) may read the network input"; return false; }
            if (L.type != MDHIP_DETECT && L.type != MDHIP_CONCAT && L.type != MDHIP_UPSAMPLE &&
                (L.first_conv < 0 || L.first_conv >= model->n_convs)) { error = "first_conv out of range"; return false; }
            switch (L.type) {
                case MDHIP_CONV: {
                    const mdhip_conv* c = &model->convs[L.first_conv];
                    Tensor out = out_view(i);
                    if (f0 < 0) {
                        if (!(c->c_in == 3 && c->kh == 6 && c->kw == 6 && L.s == 2 && L.p == 2)) {
                            error = "stem must be Conv(3->c, k=6, s=2, p=2)";
                            return false;
                        }
                        const int pc = pack({c}, true);
                        snprintf(nm, sizeof(nm), "L%d stem 6x6s2 (3x3 s2d)", i);
                        add_conv(i, nm, ctx->input, out, pc, 1, 1, true, nullptr);
                    } else {
                        if (c->c_in != layer_c[f0] || c->kh != L.k || c->kw != L.k) { error = "conv shape mismatch"; return false; }
                        if (L.k != 1 && L.k != 3) { error = "only 1x1 and 3x3 convs supported"; return false; }
                        const int pc = pack({c}, false);
                        snprintf(nm, sizeof(nm), "L%d conv %dx%ds%d", i, L.k, L.k, L.s);
                        add_conv(i, nm, ctx->layer_out[f0], out, pc, L.s, L.p, true, nullptr);
                    }
                    ctx->layer_out[i] = out;
                    break;
                }
                case MDHIP_C3: {
                    const mdhip_conv* cv = &model->convs[L.first_conv];
                    const int ch = cv[0].c_out;          // hidden width c_
                    if (ch % 8 || cv[1].c_out != ch || cv[0].c_in != layer_c[f0]) { error = "C3 shape mismatch"; return false; }
                    Tensor out = out_view(i);
                    Tensor Y = alloc(2 * ch, layer_div[i]);
                    // (a line-aligned pixel pitch for the hidden tensor -- 160 -> 192, 480 -> 512 channels -- was
                    // measured: no gain, 36.0 vs 35.9 ms per forward)
                    Tensor T = alloc(ch, layer_div[i]);
                    Tensor Y1 = slice(Y, 0, ch);
                    int pc = pack({&cv[0], &cv[1]}, false);
                    snprintf(nm, sizeof(nm), "L%d C3.cv1|cv2 1x1", i);
                    add_conv(i, nm, ctx->layer_out[f0], Y, pc, 1, 0, true, nullptr);
                    for (int j = 0; j < L.n; ++j) {
                        const mdhip_conv* b1 = &cv[3 + 2 * j];
                        const mdhip_conv* b2 = &cv[4 + 2 * j];
                        if (b1->kh != 1 || b2->kh != 3) { error = "bottleneck must be 1x1 then 3x3"; return false; }
                        pc = pack({b1}, false);
                        snprintf(nm, sizeof(nm), "L%d C3.m%d.cv1 1x1", i, j);
                        add_conv(i, nm, Y1, T, pc, 1, 0, true, nullptr);
                        pc = pack({b2}, false);
                        snprintf(nm, sizeof(nm), "L%d C3.m%d.cv2 3x3", i, j);
                        add_conv(i, nm, T, Y1, pc, 1, 1, true, L.shortcut ? &Y1 : nullptr);
                        // candidates for the fused bottleneck kernel (decided per forward from the 3x3s' tiles): an
                        // even number of bottlenecks, so that ping-ponging Y1 <-> T ends in Y1.  The 80-channel block
                        // (the shape conv_v5c.cpp takes) stays in 16 bits in the fp8 mode too: fused it is faster than
                        // its 1x1 -> e4m3 -> 3x3 pair (4.0 against 4.4 ms per 32 images) and exact.
                        const bool strip_block = ch == 80 && (L.n % 2) == 0;
                        if ((L.n % 2) == 0 && (ctx->dtype != MDHIP_DTYPE_FP8 || strip_block)) {
                            const int o2 = (int)ctx->ops.size() - 1, o1 = o2 - 1;
                            if (j == 0) ctx->fuse_groups.emplace_back();
                            ctx->ops[o1].fuse_group = ctx->ops[o2].fuse_group = (int)ctx->fuse_groups.size() - 1;
                            ctx->ops[o1].fuse_idx = ctx->ops[o2].fuse_idx = j;
                            ctx->ops[o1].fuse_role = 1;
                            ctx->ops[o2].fuse_role = 2;
                            ctx->fuse_groups.back().push_back(o2);
                        }
                        if (ctx->packed[pc].groups8 > 0 && !strip_block) {
                            // fp8 mode: the hidden tensor T of this bottleneck travels as e4m3 (1x1 writes, 3x3 reads)
                            const int o2 = (int)ctx->ops.size() - 1, o1 = o2 - 1;
                            ctx->ops[o1].f8_out = true;
                            ctx->ops[o1].f8_peer = o2;
                            ctx->ops[o2].f8_in = true;
                            ctx->ops[o2].f8_peer = o1;
                            ++ctx->n_f8;
                        }
                    }
                    pc = pack({&cv[2]}, false);
                    snprintf(nm, sizeof(nm), "L%d C3.cv3 1x1", i);
                    add_conv(i, nm, Y, out, pc, 1, 0, true, nullptr);
                    ctx->layer_out[i] = out;
                    break;
                }
                case MDHIP_SPPF: {
                    const mdhip_conv* cv = &model->convs[L.first_conv];
                    const int ch = cv[0].c_out;
                    if (ch % 8) { error = "SPPF hidden width must be a multiple of 8"; return false; }
                    Tensor out = out_view(i);
                    Tensor Y = alloc(4 * ch, layer_div[i]);
                    int pc = pack({&cv[0]}, false);
                    snprintf(nm, sizeof(nm), "L%d SPPF.cv1 1x1", i);
                    add_conv(i, nm, ctx->layer_out[f0], slice(Y, 0, ch), pc, 1, 0, true, nullptr);
                    Op pool;
                    pool.kind = OP_POOL;
                    pool.layer = i;
                    snprintf(nm, sizeof(nm), "L%d SPPF.pool x3 k%d", i, L.k);
                    pool.name = nm;
                    pool.in = slice(Y, 0, ch);
                    pool.out = Y;
                    pool.pool_k = L.k;
                    ctx->ops.push_back(pool);
                    pc = pack({&cv[1]}, false);
                    snprintf(nm, sizeof(nm), "L%d SPPF.cv2 1x1", i);
                    add_conv(i, nm, Y, out, pc, 1, 0, true, nullptr);
                    ctx->layer_out[i] = out;
                    break;
                }
                case MDHIP_UPSAMPLE: {
                    Tensor out = out_view(i);
                    Op op;
                    op.kind = OP_UPSAMPLE;
                    op.layer = i;
                    snprintf(nm, sizeof(nm), "L%d upsample x2", i);
                    op.name = nm;
                    op.in = ctx->layer_out[f0];
                    op.out = out;
                    ctx->ops.push_back(op);
                    ctx->layer_out[i] = out;
                    break;
                }
                case MDHIP_CONCAT: {
                    if (!concat_buf[i].valid) concat_buf[i] = alloc(layer_c[i], layer_div[i]);
                    // a concat feeding another concat keeps its own buffer and is copied below
                    int c = 0;
                    for (int j = 0; j < L.n_from; ++j) {
                        const int f = L.from[j];
                        if (!(concat_target[f] == i && concat_choff[f] == c)) {
                            Op op;
                            op.kind = OP_COPY;
                            op.layer = i;
                            snprintf(nm, sizeof(nm), "L%d concat copy of L%d", i, f);
                            op.name = nm;
                            op.in = ctx->layer_out[f];
                            op.out = slice(concat_buf[i], c, layer_c[f]);
                            ctx->ops.push_back(op);
                        }
                        c += layer_c[f];
                    }
                    ctx->layer_out[i] = concat_buf[i];
                    if (concat_target[i] >= 0) {
                        // nested concat: copy into the outer buffer
                        Tensor outer = out_view(i);
                        Op op;
                        op.kind = OP_COPY;
                        op.layer = i;
                        snprintf(nm, sizeof(nm), "L%d nested concat copy", i);
                        op.name = nm;
                        op.in = concat_buf[i];
                        op.out = outer;
                        ctx->ops.push_back(op);
                    }
                    break;
                }
                case MDHIP_DETECT: {
                    if (L.n_from != model->nl) { error = "Detect inputs != nl"; return false; }
                    for (int l = 0; l < L.n_from; ++l) {
                        const mdhip_conv* c = &model->convs[L.first_conv + l];
                        const int f = L.from[l];
                        if (c->c_out != ctx->na * ctx->no || c->c_in != layer_c[f] || c->kh != 1) { error = "Detect conv shape mismatch"; return false; }
                        if (std::fabs(ctx->strides[l] - (float)layer_div[f]) > 1e-6f) { error = "Detect stride does not match the graph"; return false; }
                        const int pc = pack({c}, false);
                        Op op;
                        op.kind = OP_CONV;
                        op.layer = i;
                        snprintf(nm, sizeof(nm), "L%d Detect.m%d 1x1", i, l);
                        op.name = nm;
                        op.in = ctx->layer_out[f];
                        op.pc = pc;
                        op.stride = 1;
                        op.pad = 0;
                        op.act = 0;
                        op.out_f32 = 1;
                        op.f32_ld = ctx->packed[pc].n_rows;
                        const size_t px = (size_t)ctx->max_batch * (ctx->max_h / layer_div[f]) * (ctx->max_w / layer_div[f]);
                        op.f32_off = alloc_bytes(px * op.f32_ld * 4);
                        op.out = op.in;   // spatial size only
                        op.out.c = c->c_out;
                        ctx->ops.push_back(op);
                        Op dec;
                        dec.kind = OP_DECODE;
                        dec.layer = i;
                        snprintf(nm, sizeof(nm), "L%d Detect.decode%d", i, l);
                        dec.name = nm;
                        dec.in = op.in;
                        dec.level = l;
                        dec.f32_off = op.f32_off;
                        dec.f32_ld = op.f32_ld;
                        ctx->ops.push_back(dec);
                    }
                    break;
                }
            }
        }
        // an upsample whose output is the first part of a concatenated tensor that exactly one op reads, a 1x1 / stride 1
        // conv: that conv can read the low-resolution tensor in place (conv_v2.cpp) and the upsample need not run
        for (size_t u = 0; u < ctx->ops.size(); ++u) {
            Op& up = ctx->ops[u];
            if (up.kind != OP_UPSAMPLE) continue;
            int reader = -1, readers = 0;
            for (size_t k = 0; k < ctx->ops.size(); ++k) {
                const Op& o = ctx->ops[k];
                if (o.kind == OP_DECODE) continue;
                const bool overlaps = o.in.off == up.out.off || (o.has_res && o.res.off == up.out.off);
                if (k != u && overlaps) { ++readers; reader = (int)k; }
            }
            if (readers != 1) continue;
            Op& c = ctx->ops[reader];
            const PackedConv& pc = ctx->packed[c.pc >= 0 ? c.pc : 0];
            if (c.kind == OP_CONV && reader > (int)u && c.stride == 1 && pc.kh == 1 && pc.kw == 1 && !c.f8_in &&
                c.in.off == up.out.off && c.in.ld == up.out.ld && c.in.c > up.out.c && c.in.div == up.out.div) {
                up.up_peer = reader;
                c.up_peer = (int)u;
            }
        }
        return true;
    }
};

int num_anchors_for(const mdhip_ctx* ctx, int h, int w) {
    int a = 0;
    for (int l = 0; l < ctx->nl; ++l) {
        const int s = (int)ctx->strides[l];
        a += ctx->na * (h / s) * (w / s);
    }
    return a;
}

int check_calibrated(mdhip_ctx* ctx) {
    if (ctx->dtype == MDHIP_DTYPE_FP8 && ctx->n_f8 > 0 && !ctx->calibrated)
        return fail(ctx, MDHIP_EINVAL, "fp8 context without activation scales: call mdhip_calibrate (or mdhip_fp8_set_scales) first");
    return MDHIP_OK;
}

// new range -> scale of an e4m3 tensor and the combined per-channel factors of the conv that reads it
int apply_fp8_scale(mdhip_ctx* ctx, Op& producer, float act_scale) {
    drop_graphs(ctx);                               // (the quantisation scale is a launch argument)
    producer.act_scale = act_scale;
    Op& consumer = ctx->ops[producer.f8_peer];
    consumer.act_scale = act_scale;
    const PackedConv& pc = ctx->packed[consumer.pc];
    std::vector<float> sc(pc.n_rows, 0.f);
    for (int n = 0; n < pc.n_rows; ++n) sc[n] = act_scale * pc.wscale[n];
    HIP_TRY(ctx, hipMemcpy(ctx->warena + pc.scale_off, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    return MDHIP_OK;
}

int check_shape(mdhip_ctx* ctx, int n, int h, int w) {
    if (n < 1 || n > ctx->max_batch) return fail(ctx, MDHIP_EINVAL, "batch %d outside [1,%d]", n, ctx->max_batch);
    if (h < ctx->max_stride || w < ctx->max_stride || h % ctx->max_stride || w % ctx->max_stride)
        return fail(ctx, MDHIP_EINVAL, "input %dx%d must be a positive multiple of the model stride %d", h, w, ctx->max_stride);
    if ((size_t)h * w > (size_t)ctx->max_h * ctx->max_w)
        return fail(ctx, MDHIP_ENOMEM, "input %dx%d exceeds the planned %dx%d", h, w, ctx->max_h, ctx->max_w);
    return MDHIP_OK;
}

// the conv API of the context's storage type (the kernels are compiled once per type, mdhip_internal.h)
struct ConvApi {
    int (*num_cfgs)();
    const ConvCfg& (*cfg)(int);
    hipError_t (*launch)(int, const ConvArgs&, hipStream_t);
    hipError_t (*init)();
    bool (*supports)(int, const ConvArgs&);
    bool (*is_bitwise_family)(int);
    int (*num_v1_cfgs)();
    bool (*cfg_decodes)(int);
};
const ConvApi g_conv_bf16 = {st_bf16::conv_num_cfgs, st_bf16::conv_cfg, st_bf16::conv_launch, st_bf16::conv_init,
                             st_bf16::conv_supports, st_bf16::conv_cfg_is_bitwise_family, st_bf16::conv_num_v1_cfgs, st_bf16::conv_cfg_decodes};
const ConvApi g_conv_f16 = {st_f16::conv_num_cfgs, st_f16::conv_cfg, st_f16::conv_launch, st_f16::conv_init,
                            st_f16::conv_supports, st_f16::conv_cfg_is_bitwise_family, st_f16::conv_num_v1_cfgs, st_f16::conv_cfg_decodes};
inline const ConvApi& conv_api(const mdhip_ctx* ctx) { return ctx->dtype == MDHIP_DTYPE_FP16 ? g_conv_f16 : g_conv_bf16; }
// tile configurations (count, names, families) are the same for both storage types
inline int conv_num_cfgs() { return g_conv_bf16.num_cfgs(); }
inline const ConvCfg& conv_cfg(int i) { return g_conv_bf16.cfg(i); }
inline bool conv_cfg_is_bitwise_family(int c) { return g_conv_bf16.is_bitwise_family(c); }
inline int conv_num_v1_cfgs() { return g_conv_bf16.num_v1_cfgs(); }

// heuristic tile choice; measured overrides arrive through mdhip_set_op_cfg
int choose_cfg(int M, int n_rows) {
    // prior from measurements on MI355X (profiles/autotune_r1.txt); tools/autotune.py refines it
    static const float quality[] = {0.92f, 1.00f, 0.55f, 0.95f, 0.45f, 0.70f, 0.85f, 0.85f, 0.95f, 0.55f, 0.45f, 0.65f,
                                    0.95f, 1.00f, 1.00f, 0.95f, 0.95f, 0.60f, 0.90f, 0.88f, 0.88f, 0.55f, 0.70f, 0.45f,
                                    0.50f, 0.50f, 0.50f, 0.50f};
    static_assert(sizeof(quality) / sizeof(quality[0]) == 28, "one prior per tile configuration");
    int best = 0;
    float best_score = -1.f;
    for (int i = 0; i < conv_num_v1_cfgs(); ++i) {
        const ConvCfg& c = conv_cfg(i);
        const int tn = (n_rows + c.bn - 1) / c.bn, tm = (M + c.bm - 1) / c.bm;
        const float useful = ((float)n_rows / (tn * c.bn)) * ((float)M / ((float)tm * c.bm));
        const float fill = std::min(1.0f, (float)tm * tn / 512.0f);
        const float score = useful * (0.35f + 0.65f * fill) * quality[i];
        if (score > best_score) { best_score = score; best = i; }
    }
    return best;
}

// tile choice without a table entry: the fill-aware heuristic over the first-generation configurations for a 16-bit
// op; for an op with an e4m3 operand, the best-filling configuration among those that take it
int choose_cfg_for(mdhip_ctx* ctx, const ConvArgs& a) {
    if (!a.in_f8 && !a.out_f8) return choose_cfg(a.M, a.n_rows);
    int best = -1;
    float best_score = -1.f;
    for (int i = 0; i < conv_num_cfgs(); ++i) {
        if (!conv_api(ctx).supports(i, a)) continue;
        const ConvCfg& c = conv_cfg(i);
        const int tn = (a.n_rows + c.bn - 1) / c.bn, tm = (a.M + c.bm - 1) / c.bm;
        const float useful = ((float)a.n_rows / (tn * c.bn)) * ((float)a.M / ((float)tm * c.bm));
        const float fill = std::min(1.0f, (float)tm * tn / (256.0f * c.blocks_per_cu));
        const float score = useful * (0.35f + 0.65f * fill) * (c.blocks_per_cu == 2 ? 1.0f : 0.95f);
        if (score > best_score) { best_score = score; best = i; }
    }
    return best < 0 ? 0 : best;
}

void fill_conv_args(mdhip_ctx* ctx, Op& op, int n, int h, int w, ConvArgs& a) {
    const PackedConv& pc = ctx->packed[op.pc];
    const int H = h / op.in.div, W = w / op.in.div;
    const int Ho = H / op.stride, Wo = W / op.stride;
    a.in = (const uint16_t*)(ctx->arena + op.in.off);
    a.wgt = (const uint16_t*)(ctx->warena + pc.w_off);
    a.bias = (const float*)(ctx->warena + pc.b_off);
    a.zero = (const uint16_t*)(ctx->warena + ctx->zero_off);
    a.ld_in = op.in.ld;
    a.H = H;
    a.W = W;
    a.C8 = pc.cin_pad / 8;
    a.Ho = Ho;
    a.Wo = Wo;
    a.HoWo = Ho * Wo;
    a.M = n * Ho * Wo;
    a.N = pc.c_out;
    a.n_rows = pc.n_rows;
    a.k_pad = pc.k_pad;
    a.ntaps = pc.kh * pc.kw;
    a.kw = pc.kw;
    a.stride = op.stride;
    a.pad = op.pad;
    a.act = op.act;
    a.out_f32 = op.out_f32;
    if (op.out_f32) {
        a.out = ctx->arena + op.f32_off;
        a.ld_out = op.f32_ld;
    } else {
        a.out = ctx->arena + op.out.off;
        a.ld_out = op.out.ld;
    }
    a.res = op.has_res ? (const uint16_t*)(ctx->arena + op.res.off) : nullptr;
    a.ld_res = op.has_res ? op.res.ld : 0;
    a.tiles_n = 1;
    a.dbg = nullptr;
    a.dev_param = 0;
    a.wgt8 = nullptr;
    a.scale = nullptr;
    a.k_pad8 = a.groups8 = 0;
    a.in_f8 = a.out_f8 = 0;
    a.out_qscale = 1.0f;
    a.wgt4 = pc.w4_off ? (const uint16_t*)(ctx->warena + pc.w4_off) : nullptr;
    a.k_pad4 = pc.k_pad4;
    a.groups = pc.groups;
    a.wgt4p = (pc.w4p_off && ctx->pair_enabled) ? (const uint16_t*)(ctx->warena + pc.w4p_off) : nullptr;
    a.k_pad4p = pc.k_pad4p;
    if (ctx->dtype == MDHIP_DTYPE_FP8 && !ctx->calibrating) {
        if (op.f8_in) {
            // the e4m3 tensor lives in the 16-bit tensor's allocation: same pixel pitch, counted in bytes
            a.in_f8 = 1;
            a.wgt8 = (const uint8_t*)(ctx->warena + pc.w8_off);
            a.scale = (const float*)(ctx->warena + pc.scale_off);
            a.k_pad8 = pc.k_pad8;
            a.groups8 = pc.groups8;
            a.C8 = (op.in.c + 15) / 16;
        }
        if (op.f8_out) {
            a.out_f8 = 1;
            a.out_qscale = 1.0f / op.act_scale;
        }
    }
    op.gm = a.M;
    op.gn = pc.c_out;
    op.gk = pc.k_real;
    op.flops = 2.0 * (double)a.M * pc.c_out * pc.k_real;
    const double in_px = (double)n * H * W;
    op.bytes = in_px * op.in.c * 2.0 + (double)a.M * pc.c_out * (op.out_f32 ? 4.0 : 2.0) +
               (double)pc.c_out * pc.k_real * 2.0 + (op.has_res ? (double)a.M * pc.c_out * 2.0 : 0.0);
}

// the tile configuration of a conv op for this call: forced (tests, autotune), remembered from the last call of the same
// shape, or from the table (see below); -1 = none of them applies (the caller falls back to the heuristic)
int select_cfg(mdhip_ctx* ctx, Op& op, const ConvArgs& a, int n, int h, int w, bool* from_table_out) {
    int cfg = op.forced_cfg;
    bool from_table = false;
    const bool memo_hit = cfg < 0 && op.memo_cfg >= 0 && op.memo_n == n && op.memo_h == h && op.memo_w == w;
    if (memo_hit) {
        cfg = op.memo_cfg;
        from_table = op.memo_from_table;
    } else if (cfg < 0) {
        const PackedConv& pc = ctx->packed[op.pc];
        // 1. The canonical entry: same layer geometry (N, K, taps, stride, residual), per-image M equal
        //    or nearest within 4x (the same layer at another image shape, e.g. 960x1280 instead of
        //    1280x1280), largest batch among equals.  It fixes the kernel FAMILY (= fp32 summation
        //    order) of the op -- per image, never per call, or an image's result would depend on the
        //    batch it travels in.
        // 2. Among the entries of that geometry, per-image M and family: the one measured at the
        //    nearest total M (= nearest batch size).  Small batches want smaller tiles.
        // 3. No entry within 4x of this call's total M: the fill-aware heuristic for the bitwise family,
        //    the canonical configuration otherwise.
        const double m_img = (double)a.M / n;
        const mdhip_tuned* canon = nullptr;
        double best = 1e30;
        for (const mdhip_tuned& t : ctx->tuned) {
            if (t.n != pc.c_out || t.k != pc.k_real || t.ntaps != a.ntaps || t.stride != a.stride ||
                t.has_res != (op.has_res ? 1 : 0) || t.m <= 0)
                continue;
            const int tb = t.batch > 0 ? t.batch : 32;
            const double t_img = (double)t.m / tb;
            const double r = t_img > m_img ? t_img / m_img : m_img / t_img;
            if (r > 4.0 || !conv_api(ctx).supports(t.cfg, a)) continue;
            const int cb = canon ? (canon->batch > 0 ? canon->batch : 32) : 0;
            if (r < best - 1e-9 || (r < best + 1e-9 && tb > cb)) {
                best = r;
                canon = &t;
            }
        }
        if (canon) {
            const bool fam = conv_cfg_is_bitwise_family(canon->cfg);
            const double c_img = (double)canon->m / (canon->batch > 0 ? canon->batch : 32);
            const mdhip_tuned* pick = nullptr;
            double best_m = 1e30;
            for (const mdhip_tuned& t : ctx->tuned) {
                if (t.n != canon->n || t.k != canon->k || t.ntaps != canon->ntaps || t.stride != canon->stride ||
                    t.has_res != canon->has_res || t.m <= 0 || conv_cfg_is_bitwise_family(t.cfg) != fam)
                    continue;
                const double t_img = (double)t.m / (t.batch > 0 ? t.batch : 32);
                if (t_img < c_img * 0.999 || t_img > c_img * 1.001 || !conv_api(ctx).supports(t.cfg, a)) continue;
                const double scaled = (double)t.m * (m_img / t_img);          // total M of that batch at this image shape
                const double r = scaled > a.M ? scaled / a.M : a.M / scaled;
                if (r < best_m) { best_m = r; pick = &t; }
            }
            if (pick && best_m <= 4.0) {
                cfg = pick->cfg;
                from_table = true;
            } else if (!fam) {
                cfg = canon->cfg;
                from_table = true;
            }                                   // else: heuristic below (bitwise family)
        }
    }
    *from_table_out = from_table;
    return cfg;
}

// ---- fused bottlenecks (conv_v5c.cpp) --------------------------------------------------------------------------
// A C3 block runs its bottlenecks as one launch each (1x1 -> T in LDS -> 3x3 + residual) when every 3x3 of the block
// resolves to a strip configuration for this call: the block then ping-pongs between its two buffers (Y1 -> T -> Y1 ..;
// the fused kernel must not write the tensor it reads halos from), so it is all bottlenecks of a block or none.
// Same arithmetic and K order as the two separate launches: the same bits.

// the 3x3 `op` (bottleneck j of its block) as a fused launch: x = Y1 for even j, T for odd j
void fused_args(mdhip_ctx* ctx, const Op& op, const Op& pre, ConvArgs& a) {
    const Tensor& X = (op.fuse_idx % 2 == 0) ? op.out : op.in;
    const Tensor& O = (op.fuse_idx % 2 == 0) ? op.in : op.out;
    const PackedConv& pp = ctx->packed[pre.pc];
    a.in = (const uint16_t*)(ctx->arena + X.off);
    a.ld_in = X.ld;
    a.out = ctx->arena + O.off;
    a.ld_out = O.ld;
    a.res = op.has_res ? a.in : nullptr;
    a.ld_res = op.has_res ? X.ld : 0;
    a.wgt_pre = (const uint16_t*)(ctx->warena + pp.w_off);
    a.bias_pre = (const float*)(ctx->warena + pp.b_off);
    a.k_pad_pre = pp.k_pad;
}

bool group_is_fused(mdhip_ctx* ctx, int group, int n, int h, int w) {
    if (group < 0 || !ctx->fuse_enabled || ctx->fuse_suspended) return false;
    for (int oi : ctx->fuse_groups[group]) {
        Op& op = ctx->ops[oi];
        const Op& pre = ctx->ops[oi - 1];
        const double f0 = op.flops, b0 = op.bytes;
        ConvArgs a{};
        fill_conv_args(ctx, op, n, h, w, a);
        op.flops = f0; op.bytes = b0;
        bool from_table = false;
        int cfg = select_cfg(ctx, op, a, n, h, w, &from_table);
        if (cfg < 0) cfg = choose_cfg_for(ctx, a);
        if (cfg < 0 || strncmp(conv_api(ctx).cfg(cfg).name, "v5:strip", 8) != 0) return false;
        fused_args(ctx, op, pre, a);
        if (!conv_api(ctx).supports(cfg, a)) return false;
    }
    return true;
}

// the 1x1 conv `conv` reads the first channels of its concatenated input from the low-resolution tensor of the upsample
// op in front of it (which is then not run) when its tile configuration is one of conv_v2.cpp's
void up_args(mdhip_ctx* ctx, const Op& up, ConvArgs& a) {
    a.in_up = (const uint16_t*)(ctx->arena + up.in.off);
    a.ld_up = up.in.ld;
    a.up_slabs = up.in.c / 64;
}

bool up_is_absorbed(mdhip_ctx* ctx, Op& conv, int n, int h, int w) {
    if (conv.up_peer < 0 || !ctx->fuse_enabled || ctx->fuse_suspended) return false;
    const Op& up = ctx->ops[conv.up_peer];
    if (up.in.c % 64) return false;
    const double f0 = conv.flops, b0 = conv.bytes;
    ConvArgs a{};
    fill_conv_args(ctx, conv, n, h, w, a);
    conv.flops = f0; conv.bytes = b0;
    bool from_table = false;
    int cfg = select_cfg(ctx, conv, a, n, h, w, &from_table);
    if (cfg < 0) cfg = choose_cfg_for(ctx, a);
    if (cfg < 0 || strncmp(conv_api(ctx).cfg(cfg).name, "v2:", 3) != 0) return false;
    up_args(ctx, up, a);
    return conv_api(ctx).supports(cfg, a);
}

int run_op(mdhip_ctx* ctx, Op& op, int n, int h, int w, hipStream_t s) {
    switch (op.kind) {
        case OP_CONV: {
            ConvArgs a{};
            const bool fused = op.fuse_role != 0 && group_is_fused(ctx, op.fuse_group, n, h, w);
            const bool up_in_place = op.up_peer >= 0 && up_is_absorbed(ctx, op, n, h, w);
            fill_conv_args(ctx, op, n, h, w, a);
            if (up_in_place) {
                up_args(ctx, ctx->ops[op.up_peer], a);
                op.bytes -= (double)a.M * ctx->ops[op.up_peer].in.c * 2.0 * 0.75;      // a quarter of those pixels is read
            }
            if (fused && op.fuse_role == 1) {            // this 1x1 runs inside the following 3x3's launch
                op.last_cfg = -1;
                op.pre_flops = op.flops;                  // accounted with the fused launch
                op.flops = op.bytes = 0;
                break;
            }
            if (fused) {
                const Op& pre = *(&op - 1);
                fused_args(ctx, op, pre, a);
                op.flops += pre.pre_flops;
                op.bytes -= (double)a.M * a.N * 2.0;       // T is neither written nor read
            }
            bool from_table = false;
            int cfg = select_cfg(ctx, op, a, n, h, w, &from_table);
            if (cfg < 0) cfg = choose_cfg_for(ctx, a);
            // Detect decode in this conv's epilogue (mdhip_decode_store): the plain forward of a head with 8 outputs per anchor,
            // on the two kernel families that take 1x1 / fp32-output ops; the augmented forward (anchors kept / de-scaled /
            // flipped per pass) and every other head keep the separate decode launch
            Op* dec = (op.out_f32 && (size_t)(&op - ctx->ops.data()) + 1 < ctx->ops.size() && (&op)[1].kind == OP_DECODE) ? &op + 1 : nullptr;
            if (dec) dec->dec_done = false;
            const bool plain_pass = ctx->cur_tta.keep_from == 0 && ctx->cur_tta.keep_to == 0x7fffffff && ctx->cur_tta.out_off == 0 &&
                                    ctx->cur_tta.scale == 1.0f && ctx->cur_tta.flip_lr == 0;
            // (pointwise with whole 64-channel slabs: what the decoding instantiations of conv_v2.cpp take)
            auto decodes_in_place = [&](int c) {
                return dec && ctx->fuse_decode && !ctx->fuse_suspended && ctx->no == 8 && plain_pass && !ctx->calibrating &&
                       conv_api(ctx).cfg_decodes(c) && (c < conv_num_v1_cfgs() || (a.C8 & 7) == 0);
            };
            auto set_decode = [&](int c) {
                a.dec_pred = nullptr;
                if (!decodes_in_place(c)) return;
                int level_off = 0;
                for (int l = 0; l < dec->level; ++l) {
                    const int sl = (int)ctx->strides[l];
                    level_off += ctx->na * (h / sl) * (w / sl);
                }
                a.dec_pred = (float*)(ctx->arena + ctx->pred_off);
                a.dec_anchors = (const float*)(ctx->warena + ctx->anchors_off) + dec->level * ctx->na * 2;
                a.dec_stride = ctx->strides[dec->level];
                a.dec_level_off = level_off;
                a.dec_n_anchors = ctx->cur_A;
            };
            set_decode(cfg);
            hipError_t le = conv_api(ctx).launch(cfg, a, s);
            if (le == hipErrorInvalidValue && from_table && !fused && !up_in_place) {
                // table entry from another build: not applicable.  Only for an op that is launched as planned: with
                // `fused` the arguments have in / out swapped and the 1x1 in front was skipped, with `up_in_place` the
                // upsample launch was skipped -- a kernel that ignores those fields would read tensors that were
                // never written, so those cases keep the error (group_is_fused / up_is_absorbed checked supports()
                // for this very configuration: reaching this is a bug, not a stale table).
                (void)hipGetLastError();
                cfg = choose_cfg_for(ctx, a);
                from_table = false;
                set_decode(cfg);
                le = conv_api(ctx).launch(cfg, a, s);
            }
            if (dec && a.dec_pred && le == hipSuccess) {
                dec->dec_done = true;
            }
            op.last_cfg = cfg;
            if (op.forced_cfg < 0 && le == hipSuccess) {
                op.memo_n = n; op.memo_h = h; op.memo_w = w; op.memo_cfg = cfg; op.memo_from_table = from_table;
            }
            HIP_TRY(ctx, le);
            if (ctx->calibrating && op.f8_out)
                HIP_TRY(ctx, launch_absmax_view((const uint16_t*)(ctx->arena + op.out.off), op.out.ld, op.out.c, (long long)a.M,
                                                0, (float*)(ctx->arena + op.amax_off), s));
            break;
        }
        case OP_POOL: {
            const int H = h / op.in.div, W = w / op.in.div;
            op.flops = 0;
            op.bytes = (double)n * H * W * op.in.c * 2.0 * 4.0;
            HIP_TRY(ctx, launch_sppf_pool((uint16_t*)(ctx->arena + op.out.off), op.out.ld, op.in.c, n, H, W, op.pool_k, ctx->dtype == MDHIP_DTYPE_FP16, s));
            break;
        }
        case OP_UPSAMPLE: {
            if (op.up_peer >= 0 && up_is_absorbed(ctx, ctx->ops[op.up_peer], n, h, w)) {     // read in place by its consumer
                op.bytes = 0;
                break;
            }
            const int H = h / op.in.div, W = w / op.in.div;
            op.bytes = (double)n * H * W * op.in.c * 2.0 * 5.0;
            HIP_TRY(ctx, launch_upsample2x((const uint16_t*)(ctx->arena + op.in.off), op.in.ld,
                                           (uint16_t*)(ctx->arena + op.out.off), op.out.ld, op.in.c, n, H, W, s));
            break;
        }
        case OP_COPY: {
            const long long px = (long long)n * (h / op.in.div) * (w / op.in.div);
            op.bytes = (double)px * op.in.c * 4.0;
            HIP_TRY(ctx, launch_copy_view((const uint16_t*)(ctx->arena + op.in.off), op.in.ld,
                                          (uint16_t*)(ctx->arena + op.out.off), op.out.ld, op.in.c, px, s));
            break;
        }
        case OP_DECODE: {
            if (op.dec_done) {                     // decoded in the epilogue of the conv in front (this forward)
                op.bytes = 0;
                op.last_cfg = -2;                  // (mdhip_get_op_info: -2 = folded into the conv in front, -1 = own launch)
                break;
            }
            op.last_cfg = -1;
            const int ny = h / op.in.div, nx = w / op.in.div;
            int level_off = 0;
            for (int l = 0; l < op.level; ++l) {
                const int sl = (int)ctx->strides[l];
                level_off += ctx->na * (h / sl) * (w / sl);
            }
            op.bytes = (double)n * ny * nx * ctx->na * ctx->no * 8.0;
            HIP_TRY(ctx, launch_detect_decode((const float*)(ctx->arena + op.f32_off), op.f32_ld,
                                              (float*)(ctx->arena + ctx->pred_off), n, ny, nx, ctx->na,
                                              ctx->no, ctx->cur_A, level_off, ctx->strides[op.level],
                                              (const float*)(ctx->warena + ctx->anchors_off) + op.level * ctx->na * 2,
                                              ctx->cur_tta, s));
            break;
        }
    }
    return MDHIP_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* mdhip_version(void) { return "mdhip 0.2 (gfx950: bf16 / fp16 storage, fp8 bottlenecks)"; }

const char* mdhip_last_error(mdhip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mdhip_create(const mdhip_model* model, int device, int dtype, int max_batch, int max_h,
                 int max_w, mdhip_ctx** out) {
    if (!out) return fail(nullptr, MDHIP_EINVAL, "out is NULL");
    *out = nullptr;
    if (!model || !model->layers || !model->convs || model->n_layers < 1)
        return fail(nullptr, MDHIP_EINVAL, "empty model description");
    if (dtype != MDHIP_DTYPE_BF16 && dtype != MDHIP_DTYPE_FP16 && dtype != MDHIP_DTYPE_FP8)
        return fail(nullptr, MDHIP_EUNSUPPORTED, "dtype %d not implemented (bf16, fp16, fp8)", dtype);
    if (max_batch < 1 || max_h < 64 || max_w < 64) return fail(nullptr, MDHIP_EINVAL, "bad capacity %d x %dx%d", max_batch, max_h, max_w);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, MDHIP_EHIP, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, MDHIP_EINVAL, "device %d outside [0,%d)", device, ndev);

    mdhip_ctx* ctx = new mdhip_ctx();
    if (const char* ef = getenv("MDHIP_FUSE")) ctx->fuse_enabled = atoi(ef) != 0;      // A/B measurements
    if (const char* ep = getenv("MDHIP_PAIR")) ctx->pair_enabled = atoi(ep) != 0;      // A/B measurements, bit-identity test
    ctx->letterbox_general = getenv("MDHIP_LETTERBOX_GENERAL") != nullptr;             // (read once, not per mdhip_preprocess)
    ctx->device = device;
    ctx->dtype = dtype;
    ctx->max_batch = max_batch;
    ctx->nc = model->nc;
    ctx->na = model->na;
    ctx->nl = model->nl;
    ctx->no = model->nc + 5;
    bool has_detect = false;
    for (int i = 0; i < model->n_layers; ++i) has_detect |= model->layers[i].type == MDHIP_DETECT;
    ctx->max_stride = 2;
    if (has_detect) {
        if (model->nl < 1 || !model->strides || !model->anchors_px || model->nc < 1 || model->nc > 250) {
            delete ctx;
            return fail(nullptr, MDHIP_EINVAL, "Detect layer needs nl/strides/anchors/nc");
        }
        ctx->strides.assign(model->strides, model->strides + model->nl);
        for (float s : ctx->strides) ctx->max_stride = std::max(ctx->max_stride, (int)s);
    }
    {   // largest stride of any layer (models without Detect, used by unit tests)
        std::vector<int> div(model->n_layers, 1);
        for (int i = 0; i < model->n_layers; ++i) {
            const mdhip_layer& L = model->layers[i];
            const int f0 = L.n_from > 0 ? L.from[0] : -1;
            const int d = (f0 < 0 || f0 >= i) ? 1 : div[f0];
            div[i] = L.type == MDHIP_CONV ? d * std::max(1, L.s) : (L.type == MDHIP_UPSAMPLE ? std::max(1, d / 2) : d);
            ctx->max_stride = std::max(ctx->max_stride, div[i]);
        }
    }
    ctx->max_h = round_up(max_h, ctx->max_stride);
    ctx->max_w = round_up(max_w, ctx->max_stride);
    ctx->layers.assign(model->layers, model->layers + model->n_layers);

    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = conv_api(ctx).init();
    if (e != hipSuccess) {
        delete ctx;
        return fail(nullptr, MDHIP_EHIP, "device init failed: %s", hipGetErrorString(e));
    }

    Planner P;
    P.ctx = ctx;
    P.model = model;
    if (!P.plan()) {
        std::string msg = P.error;
        delete ctx;
        return fail(nullptr, MDHIP_EINVAL, "model planning failed: %s", msg.c_str());
    }
    // predictions, NMS scratch, letterbox geometry
    ctx->a_max = has_detect ? num_anchors_for(ctx, ctx->max_h, ctx->max_w) : 1;
    // room for the concatenated predictions of test-time augmentation (three passes, <= 3 x a_max)
    ctx->a_cap = has_detect ? 3 * ctx->a_max : 1;
    ctx->pred_offs[0] = P.alloc_bytes((size_t)max_batch * ctx->a_cap * ctx->no * 4);
    ctx->pred_offs[1] = P.alloc_bytes((size_t)max_batch * ctx->a_cap * ctx->no * 4);
    ctx->pred_off = ctx->pred_offs[0];
    size_t nms_kv[6];
    for (int i = 0; i < 6; ++i) nms_kv[i] = P.alloc_bytes((size_t)max_batch * ctx->a_cap * 4);
    const size_t nms_seg = P.alloc_bytes((size_t)max_batch * kNmsScanParts * 4);
    ctx->nms_out_off = P.alloc_bytes((size_t)max_batch * kNmsMaxDet * 6 * 4);
    ctx->nms_cnt_off = P.alloc_bytes((size_t)max_batch * 4);
    ctx->geom_off = P.alloc_bytes((size_t)max_batch * sizeof(LetterboxDev));
    {   // fp8 calibration: one range word per e4m3 tensor
        const size_t base = P.alloc_bytes((size_t)std::max(1, ctx->n_f8) * 4);
        size_t k = 0;
        for (Op& op : ctx->ops)
            if (op.f8_out) op.amax_off = base + 4 * k++;
    }
    ctx->arena_bytes = P.cursor + 256;

    // weight arena
    size_t wcur = 0;
    ctx->zero_off = 0;
    wcur = 256;
    ctx->anchors_off = wcur;
    wcur = align_up(wcur + (size_t)std::max(1, ctx->nl * ctx->na * 2) * 4, 256);
    for (size_t i = 0; i < ctx->packed.size(); ++i) {
        ctx->packed[i].w_off = wcur;
        wcur = align_up(wcur + P.w_host[i].size() * 2, 256);
        ctx->packed[i].b_off = wcur;
        wcur = align_up(wcur + P.b_host[i].size() * 4, 256);
        if (!P.w4_host[i].empty()) {
            ctx->packed[i].w4_off = wcur;
            wcur = align_up(wcur + P.w4_host[i].size() * 2, 256);
        }
        if (!P.w4p_host[i].empty()) {
            ctx->packed[i].w4p_off = wcur;
            wcur = align_up(wcur + P.w4p_host[i].size() * 2, 256);
        }
        if (!P.w8_host[i].empty()) {
            ctx->packed[i].w8_off = wcur;
            wcur = align_up(wcur + P.w8_host[i].size(), 256);
            ctx->packed[i].scale_off = wcur;
            wcur = align_up(wcur + (size_t)ctx->packed[i].n_rows * 4, 256);
        }
    }
    ctx->warena_bytes = wcur;

#define CREATE_TRY(expr)                                                                        \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            std::string m = std::string(#expr) + " failed: " + hipGetErrorString(e__);          \
            mdhip_destroy(ctx);                                                                 \
            return fail(nullptr, e__ == hipErrorOutOfMemory ? MDHIP_ENOMEM : MDHIP_EHIP, "%s", m.c_str()); \
        }                                                                                       \
    } while (0)

    // mdhip_forward records `input_free` right behind op 0: nothing after it may read the network input
    for (size_t oi = 1; oi < ctx->ops.size(); ++oi)
        if ((ctx->ops[oi].in.valid && ctx->ops[oi].in.off == ctx->input.off) ||
            (ctx->ops[oi].has_res && ctx->ops[oi].res.off == ctx->input.off)) {
            const std::string m = "op " + std::to_string(oi) + " (" + ctx->ops[oi].name + ") reads the network input: only layer 0 may";
            mdhip_destroy(ctx);
            return fail(nullptr, MDHIP_EINVAL, "%s", m.c_str());
        }
    CREATE_TRY(hipMalloc((void**)&ctx->arena, ctx->arena_bytes));
    CREATE_TRY(hipMalloc((void**)&ctx->warena, ctx->warena_bytes));
    CREATE_TRY(hipMemset(ctx->warena, 0, 256));
    CREATE_TRY(hipHostMalloc((void**)&ctx->geom_host, (size_t)4 * max_batch * sizeof(LetterboxDev), hipHostMallocDefault));
    for (int i = 0; i < 4; ++i) CREATE_TRY(hipEventCreateWithFlags(&ctx->geom_ev[i], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->input_free, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) CREATE_TRY(hipEventCreateWithFlags(&ctx->pred_read[i], hipEventDisableTiming));
    for (int i = 0; i < MDHIP_NMS_SLOTS; ++i) {
        CREATE_TRY(hipHostMalloc((void**)&ctx->nms_host_out[i], (size_t)max_batch * kNmsMaxDet * 6 * 4, hipHostMallocDefault));
        CREATE_TRY(hipHostMalloc((void**)&ctx->nms_host_cnt[i], (size_t)max_batch * 4, hipHostMallocDefault));
        CREATE_TRY(hipEventCreateWithFlags(&ctx->nms_ev[i], hipEventDisableTiming));
    }
    // The whole arena starts as zeros: no kernel's result may depend on memory nobody wrote (K-slab tails against zero
    // weights, pad channels, halo rows of a neighbouring tensor).  MDHIP_ARENA_POISON=1 (tests) fills it with 0xFF bytes
    // instead -- NaN in bf16, fp16 and fp32 -- so that any such read shows up as NaN in the output.
    {
        const char* pz = getenv("MDHIP_ARENA_POISON");
        CREATE_TRY(hipMemset(ctx->arena, (pz && atoi(pz) != 0) ? 0xff : 0, ctx->arena_bytes));
    }
    if (has_detect)
        CREATE_TRY(hipMemcpy(ctx->warena + ctx->anchors_off, model->anchors_px, (size_t)ctx->nl * ctx->na * 2 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < ctx->packed.size(); ++i) {
        CREATE_TRY(hipMemcpy(ctx->warena + ctx->packed[i].w_off, P.w_host[i].data(), P.w_host[i].size() * 2, hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(ctx->warena + ctx->packed[i].b_off, P.b_host[i].data(), P.b_host[i].size() * 4, hipMemcpyHostToDevice));
        if (!P.w4_host[i].empty())
            CREATE_TRY(hipMemcpy(ctx->warena + ctx->packed[i].w4_off, P.w4_host[i].data(), P.w4_host[i].size() * 2, hipMemcpyHostToDevice));
        if (!P.w4p_host[i].empty())
            CREATE_TRY(hipMemcpy(ctx->warena + ctx->packed[i].w4p_off, P.w4p_host[i].data(), P.w4p_host[i].size() * 2, hipMemcpyHostToDevice));
        if (!P.w8_host[i].empty()) {
            CREATE_TRY(hipMemcpy(ctx->warena + ctx->packed[i].w8_off, P.w8_host[i].data(), P.w8_host[i].size(), hipMemcpyHostToDevice));
            CREATE_TRY(hipMemset(ctx->warena + ctx->packed[i].scale_off, 0, (size_t)ctx->packed[i].n_rows * 4));
        }
    }
    for (int i = 0; i < 3; ++i) {
        ctx->nms_scr.keys[i] = (uint32_t*)(ctx->arena + nms_kv[i]);
        ctx->nms_scr.vals[i] = (uint32_t*)(ctx->arena + nms_kv[3 + i]);
    }
    ctx->nms_scr.cap = ctx->a_cap;
    ctx->nms_scr.seg_cnt = (uint32_t*)(ctx->arena + nms_seg);
    CREATE_TRY(hipDeviceSynchronize());
#undef CREATE_TRY
    *out = ctx;
    return MDHIP_OK;
}

void mdhip_destroy(mdhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    drop_graphs(ctx);
    if (ctx->capture_stream) (void)hipStreamDestroy(ctx->capture_stream);
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    for (int i = 0; i < mdhip_ctx::kFwdRing; ++i)
        for (int k = 0; k < 2; ++k)
            if (ctx->fwd_ev[i][k]) (void)hipEventDestroy(ctx->fwd_ev[i][k]);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->warena) (void)hipFree(ctx->warena);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->geom_host) (void)hipHostFree(ctx->geom_host);
    for (int i = 0; i < 4; ++i) if (ctx->geom_ev[i]) (void)hipEventDestroy(ctx->geom_ev[i]);
    if (ctx->input_free) (void)hipEventDestroy(ctx->input_free);
    for (int i = 0; i < 2; ++i)
        if (ctx->pred_read[i]) (void)hipEventDestroy(ctx->pred_read[i]);
    for (int i = 0; i < MDHIP_NMS_SLOTS; ++i) {
        if (ctx->nms_host_out[i]) (void)hipHostFree(ctx->nms_host_out[i]);
        if (ctx->nms_host_cnt[i]) (void)hipHostFree(ctx->nms_host_cnt[i]);
        if (ctx->nms_ev[i]) (void)hipEventDestroy(ctx->nms_ev[i]);
    }
    delete ctx;
}

int mdhip_max_stride(mdhip_ctx* ctx) { return ctx ? ctx->max_stride : MDHIP_EINVAL; }

int mdhip_num_anchors(mdhip_ctx* ctx, int h, int w) {
    if (!ctx) return MDHIP_EINVAL;
    return num_anchors_for(ctx, h, w);
}

int mdhip_preprocess(mdhip_ctx* ctx, const uint8_t* const* images, const mdhip_letterbox* geom,
                     int n, int out_h, int out_w, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (!images || !geom) return fail(ctx, MDHIP_EINVAL, "images/geom is NULL");
    if (int rc = check_shape(ctx, n, out_h, out_w)) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<LetterboxDev> g(n);
    std::vector<bool> on_host(n);
    size_t host_bytes = 0;
    for (int i = 0; i < n; ++i) {
        const mdhip_letterbox& q = geom[i];
        if (!images[i] || q.src_h < 1 || q.src_w < 1 || q.resized_h < 1 || q.resized_w < 1 || q.top < 0 || q.left < 0 ||
            q.top + q.resized_h > out_h || q.left + q.resized_w > out_w)
            return fail(ctx, MDHIP_EINVAL, "image %d: letterbox geometry does not fit %dx%d", i, out_h, out_w);
        hipPointerAttribute_t attr;
        const hipError_t e = hipPointerGetAttributes(&attr, images[i]);
        bool host = true;
        if (e == hipSuccess) host = !(attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
        else (void)hipGetLastError();
        on_host[i] = host;
        if (host) host_bytes += align_up((size_t)q.src_h * q.src_w * 3, 256);
        if (q.interp != 0 && q.interp != 1) return fail(ctx, MDHIP_EINVAL, "image %d: interp %d (0 = linear, 1 = area)", i, q.interp);
        if (q.interp == 1 && (q.resized_h > q.src_h || q.resized_w > q.src_w))
            return fail(ctx, MDHIP_EINVAL, "image %d: INTER_AREA is implemented for shrinking only", i);
        g[i] = LetterboxDev{images[i], q.src_h, q.src_w, q.resized_h, q.resized_w, q.top, q.left, q.interp,
                            1.0 / ((double)q.resized_w / (double)q.src_w), 1.0 / ((double)q.resized_h / (double)q.src_h)};
    }
    if (host_bytes > ctx->stage_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(s));
        if (ctx->stage) HIP_TRY(ctx, hipFree(ctx->stage));
        ctx->stage = nullptr;
        ctx->stage_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->stage, host_bytes));
        ctx->stage_bytes = host_bytes;
    }
    size_t cur = 0;
    for (int i = 0; i < n; ++i) {
        if (!on_host[i]) continue;
        const size_t bytes = (size_t)g[i].src_h * g[i].src_w * 3;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->stage + cur, images[i], bytes, hipMemcpyHostToDevice, s));
        g[i].src = (const uint8_t*)(ctx->stage + cur);
        cur += align_up(bytes, 256);
    }
    // the forward that still reads the input tensor (its stem) comes first, whatever stream it runs on
    if (ctx->input_free_valid) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->input_free, 0));
    if (!letterbox_geometry_travels_inline(g.data(), n, out_w, ctx->letterbox_general)) {
        // geometry goes through a 4-deep pinned ring so that the call never blocks on the stream
        const int slot = ctx->geom_slot;
        ctx->geom_slot = (slot + 1) & 3;
        HIP_TRY(ctx, hipEventSynchronize(ctx->geom_ev[slot]));          // slot's previous copy has completed
        LetterboxDev* gh = ctx->geom_host + (size_t)slot * ctx->max_batch;
        memcpy(gh, g.data(), n * sizeof(LetterboxDev));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->arena + ctx->geom_off, gh, n * sizeof(LetterboxDev), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipEventRecord(ctx->geom_ev[slot], s));
    }
    HIP_TRY(ctx, launch_letterbox_s2d((const LetterboxDev*)(ctx->arena + ctx->geom_off), g.data(), n, out_h, out_w,
                                      (uint16_t*)(ctx->arena + ctx->input.off), ctx->dtype == MDHIP_DTYPE_FP16, ctx->letterbox_general, s));
    ctx->last_n = n;
    ctx->last_h = out_h;
    ctx->last_w = out_w;
    return MDHIP_OK;
}

int mdhip_forward(mdhip_ctx* ctx, int n, int h, int w, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (int rc = check_shape(ctx, n, h, w)) return rc;
    if (int rc = check_calibrated(ctx)) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)(ctx->fwd_count % mdhip_ctx::kFwdRing);
    if (ctx->time_forward) HIP_TRY(ctx, hipEventRecord(ctx->fwd_ev[slot][0], s));
    ctx->cur_tta = DecodeTta();
    ctx->cur_A = num_anchors_for(ctx, h, w);
    ctx->pred_cur ^= 1;
    ctx->pred_off = ctx->pred_offs[ctx->pred_cur];
    if (ctx->pred_read_valid[ctx->pred_cur])            // an NMS on another stream may still read the buffer this forward overwrites
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->pred_read[ctx->pred_cur], 0));
    const bool use_graph = (ctx->graph_mode == 1 || (ctx->graph_mode == 2 && n <= ctx->graph_max_n)) && !ctx->calibrating;
    bool launched = false;
    if (use_graph) {
        mdhip_ctx::GraphSlot& g = ctx->graphs[std::make_tuple(n, h, w, ctx->pred_cur)];
        g.last_use = ++ctx->graph_clock;
        if (g.exec) {
            HIP_TRY(ctx, hipGraphLaunch(g.exec, s));
            launched = true;
        } else if (!g.disabled && ++g.seen >= 2) {
            // the first forward of a shape runs eagerly (it settles the tile choices: a stale table entry is replaced on
            // its first failing launch); the second is captured on an internal stream and replayed from then on.  A
            // capture or instantiation that fails disables replay for this shape: the eager path below just worked.
            if (!ctx->capture_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->capture_stream, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamBeginCapture(ctx->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                for (Op& op : ctx->ops)
                    if (run_op(ctx, op, n, h, w, ctx->capture_stream) != MDHIP_OK) { ok = false; break; }
                ok = (hipStreamEndCapture(ctx->capture_stream, &graph) == hipSuccess) && ok && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec != nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            if (!ok) {
                (void)hipGetLastError();
                if (exec) (void)hipGraphExecDestroy(exec);
                g.disabled = true;
            } else {
                // (evict_graph_if_full may erase other map entries: std::map references to `g` stay valid)
                evict_graph_if_full(ctx);
                g.exec = exec;
                HIP_TRY(ctx, hipGraphLaunch(g.exec, s));
                launched = true;
            }
        }
    }
    if (!launched) {
        for (size_t oi = 0; oi < ctx->ops.size(); ++oi) {
            if (int rc = run_op(ctx, ctx->ops[oi], n, h, w, s)) return rc;
            if (oi == 0) {                                   // (the planner lets only layer 0 read the network input)
                HIP_TRY(ctx, hipEventRecord(ctx->input_free, s));
                ctx->input_free_valid = true;
            }
        }
    } else {
        HIP_TRY(ctx, hipEventRecord(ctx->input_free, s));
        ctx->input_free_valid = true;
    }
    if (ctx->time_forward) {
        HIP_TRY(ctx, hipEventRecord(ctx->fwd_ev[slot][1], s));
        ++ctx->fwd_count;
    }
    ctx->last_n = n;
    ctx->last_h = h;
    ctx->last_w = w;
    ctx->last_A = ctx->cur_A;
    return MDHIP_OK;
}

// yolov5 models/yolo.py:_forward_augment (what `model(batch, augment=True)` runs, reference
// pytorch_detector.py:1313): three passes at scales 1 / 0.83 (left-right flipped) / 0.67 of the letterboxed
// batch, boxes de-scaled and un-flipped, the largest-stride level of the first pass and the smallest-stride
// level of the last pass dropped (_clip_augmented), predictions concatenated along the anchor axis.
int mdhip_forward_tta(mdhip_ctx* ctx, int n, int h, int w, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (int rc = check_shape(ctx, n, h, w)) return rc;
    if (ctx->last_n < n || ctx->last_h != h || ctx->last_w != w)
        return fail(ctx, MDHIP_EINVAL, "mdhip_forward_tta needs mdhip_preprocess of the same batch first");
    if (int rc = check_calibrated(ctx)) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float scales[3] = {1.0f, 0.83f, 0.67f};
    const double scales_d[3] = {1.0, 0.83, 0.67};
    const int flips[3] = {0, 1, 0};
    const int gs = ctx->max_stride;
    int sh[3], sw[3], oh[3], ow[3], A[3];
    for (int k = 0; k < 3; ++k) {
        sh[k] = k ? (int)(h * scales_d[k]) : h;
        sw[k] = k ? (int)(w * scales_d[k]) : w;
        oh[k] = k ? (int)std::ceil(h * scales_d[k] / gs) * gs : h;
        ow[k] = k ? (int)std::ceil(w * scales_d[k] / gs) * gs : w;
        A[k] = num_anchors_for(ctx, oh[k], ow[k]);
    }
    long long g = 0, p4 = 1;
    for (int l = 0; l < ctx->nl; ++l) { g += p4; if (l + 1 < ctx->nl) p4 *= 4; }      // p4 = 4^(nl-1)
    const int i1 = (int)(A[0] / g), i3 = (int)((A[2] / g) * p4);
    const int total = (A[0] - i1) + A[1] + (A[2] - i3);
    if (total > ctx->a_cap || i1 >= A[0] || i3 >= A[2])
        return fail(ctx, MDHIP_ENOMEM, "augmented prediction of %d anchors exceeds the planned %d", total, ctx->a_cap);
    const size_t in_bytes = (size_t)n * (h / 2) * (w / 2) * 16 * 2;
    uint16_t* in = (uint16_t*)(ctx->arena + ctx->input.off);
    uint16_t* orig = (uint16_t*)(ctx->arena + ctx->input_orig.off);
    HIP_TRY(ctx, hipMemcpyAsync(orig, in, in_bytes, hipMemcpyDeviceToDevice, s));
    const int f16 = ctx->dtype == MDHIP_DTYPE_FP16;
    ctx->cur_A = total;
    ctx->pred_cur ^= 1;
    ctx->pred_off = ctx->pred_offs[ctx->pred_cur];
    if (ctx->pred_read_valid[ctx->pred_cur])            // an NMS on another stream may still read the buffer this forward overwrites
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->pred_read[ctx->pred_cur], 0));
    int out_off = 0;
    for (int k = 0; k < 3; ++k) {
        if (k) HIP_TRY(ctx, launch_tta_scale(orig, in, n, h, w, sh[k], sw[k], oh[k], ow[k], flips[k], f16, s));
        DecodeTta t;
        t.keep_from = k == 2 ? i3 : 0;
        t.keep_to = k == 0 ? A[0] - i1 : A[k];
        t.out_off = out_off;
        t.scale = scales[k];
        t.flip_lr = flips[k];
        t.img_w = (float)w;
        ctx->cur_tta = t;
        for (Op& op : ctx->ops)
            if (int rc = run_op(ctx, op, n, oh[k], ow[k], s)) return rc;
        out_off += t.keep_to - t.keep_from;
    }
    HIP_TRY(ctx, hipMemcpyAsync(in, orig, in_bytes, hipMemcpyDeviceToDevice, s));       // `input` holds the batch again
    HIP_TRY(ctx, hipEventRecord(ctx->input_free, s));
    ctx->input_free_valid = true;
    ctx->cur_tta = DecodeTta();
    ctx->last_n = n;
    ctx->last_h = h;
    ctx->last_w = w;
    ctx->last_A = total;
    return MDHIP_OK;
}

// fp8 mode (BASELINE.json configs[4]).  The reference has no reduced precision (pytorch_detector.py:848
// half_precision = False), so there is nothing upstream to mirror: post-training static quantisation of the hidden
// tensor of every bottleneck, per-tensor activation scale from the largest magnitude seen on the calibration batches
// (x FP8_RANGE_MARGIN head-room), per-output-channel weight scales fixed at mdhip_create.
static constexpr float kFp8RangeMargin = 2.0f;

int mdhip_calibrate(mdhip_ctx* ctx, int n, int h, int w, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (ctx->dtype != MDHIP_DTYPE_FP8) return fail(ctx, MDHIP_EINVAL, "mdhip_calibrate: not an fp8 context");
    if (int rc = check_shape(ctx, n, h, w)) return rc;
    if (ctx->last_n < n || ctx->last_h != h || ctx->last_w != w)
        return fail(ctx, MDHIP_EINVAL, "mdhip_calibrate needs mdhip_preprocess of the same batch first");
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (Op& op : ctx->ops)
        if (op.f8_out) HIP_TRY(ctx, hipMemsetAsync(ctx->arena + op.amax_off, 0, 4, s));
    // one forward in 16 bits (every op, the 3x3 convs through their bf16 weights), ranges recorded on the way
    ctx->calibrating = true;
    for (Op& op : ctx->ops) op.memo_cfg = -1;
    ctx->cur_tta = DecodeTta();
    ctx->cur_A = num_anchors_for(ctx, h, w);
    int rc = MDHIP_OK;
    for (Op& op : ctx->ops)
        if ((rc = run_op(ctx, op, n, h, w, s)) != MDHIP_OK) break;
    ctx->calibrating = false;
    for (Op& op : ctx->ops) op.memo_cfg = -1;
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->input_free_valid = false;                      // (the stream is idle: nothing to wait for)
    for (Op& op : ctx->ops) {
        if (!op.f8_out) continue;
        float m = 0.f;
        HIP_TRY(ctx, hipMemcpy(&m, ctx->arena + op.amax_off, 4, hipMemcpyDeviceToHost));
        if (!(m == m) || m > 3.0e38f) return fail(ctx, MDHIP_EINVAL, "calibration saw a non-finite activation in %s", op.name.c_str());
        op.amax = std::max(op.amax, m);                                   // ranges accumulate over calibration calls
        const float scale = std::max(op.amax, 1e-20f) * kFp8RangeMargin / 448.0f;
        if (int r2 = apply_fp8_scale(ctx, op, scale)) return r2;
    }
    ctx->calibrated = true;
    ctx->last_A = ctx->cur_A;
    return MDHIP_OK;
}

int mdhip_fp8_num_tensors(mdhip_ctx* ctx) { return ctx ? ctx->n_f8 : MDHIP_EINVAL; }

int mdhip_f32_to_e4m3(const float* in, uint8_t* out, int n) {
    if (!in || !out || n < 0) return MDHIP_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = f32_to_e4m3(in[i]);
    return MDHIP_OK;
}

int mdhip_fp8_get_scales(mdhip_ctx* ctx, float* scales, int32_t* layers, int32_t* ops, int max_n) {
    if (!ctx || max_n < 0) return MDHIP_EINVAL;
    int k = 0;
    for (size_t i = 0; i < ctx->ops.size(); ++i) {
        const Op& op = ctx->ops[i];
        if (!op.f8_out) continue;
        if (k < max_n) {
            if (scales) scales[k] = op.act_scale;
            if (layers) layers[k] = op.layer;
            if (ops) ops[k] = (int32_t)i;
        }
        ++k;
    }
    return k;
}

int mdhip_fp8_set_scales(mdhip_ctx* ctx, const float* scales, int n) {
    if (!ctx || !scales) return MDHIP_EINVAL;
    if (ctx->dtype != MDHIP_DTYPE_FP8) return fail(ctx, MDHIP_EINVAL, "mdhip_fp8_set_scales: not an fp8 context");
    if (n != ctx->n_f8) return fail(ctx, MDHIP_EINVAL, "%d scales for %d fp8 tensors", n, ctx->n_f8);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int k = 0;
    for (Op& op : ctx->ops) {
        if (!op.f8_out) continue;
        const float sc = scales[k++];
        if (!(sc > 0.f) || sc > 3.0e38f) return fail(ctx, MDHIP_EINVAL, "scale %d is not a positive finite number", k - 1);
        op.amax = sc * 448.0f / kFp8RangeMargin;
        if (int rc = apply_fp8_scale(ctx, op, sc)) return rc;
    }
    ctx->calibrated = true;
    for (Op& op : ctx->ops) op.memo_cfg = -1;
    return MDHIP_OK;
}

int mdhip_last_num_anchors(mdhip_ctx* ctx) { return ctx ? ctx->last_A : 0; }

int mdhip_time_forwards(mdhip_ctx* ctx, int enable) {
    if (!ctx) return MDHIP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !ctx->fwd_ev[0][0])
        for (int i = 0; i < mdhip_ctx::kFwdRing; ++i)
            for (int k = 0; k < 2; ++k) HIP_TRY(ctx, hipEventCreate(&ctx->fwd_ev[i][k]));
    ctx->time_forward = enable != 0;
    ctx->fwd_count = 0;
    return MDHIP_OK;
}

int mdhip_forward_times(mdhip_ctx* ctx, float* ms, int max_n) {
    if (!ctx || !ms || max_n < 0) return MDHIP_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long have = std::min<long long>(ctx->fwd_count, mdhip_ctx::kFwdRing);
    const int n = (int)std::min<long long>(have, max_n);
    for (int i = 0; i < n; ++i) {                       // the most recent n forwards, oldest first
        const int slot = (int)((ctx->fwd_count - n + i) % mdhip_ctx::kFwdRing);
        HIP_TRY(ctx, hipEventSynchronize(ctx->fwd_ev[slot][1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->fwd_ev[slot][0], ctx->fwd_ev[slot][1]));
    }
    return n;
}

int mdhip_forward_timed(mdhip_ctx* ctx, int n, int h, int w, float* ms, void* hip_stream) {
    if (!ctx || !ms) return MDHIP_EINVAL;
    if (int rc = check_shape(ctx, n, h, w)) return rc;
    if (int rc = check_calibrated(ctx)) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t need = ctx->ops.size() + 1;
    while (ctx->events.size() < need) {
        hipEvent_t ev;
        HIP_TRY(ctx, hipEventCreate(&ev));
        ctx->events.push_back(ev);
    }
    HIP_TRY(ctx, hipEventRecord(ctx->events[0], s));
    ctx->cur_tta = DecodeTta();
    ctx->cur_A = num_anchors_for(ctx, h, w);
    for (size_t i = 0; i < ctx->ops.size(); ++i) {
        if (int rc = run_op(ctx, ctx->ops[i], n, h, w, s)) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->events[i + 1], s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->input_free_valid = false;
    for (size_t i = 0; i < ctx->ops.size(); ++i)
        HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->events[i], ctx->events[i + 1]));
    ctx->last_n = n;
    ctx->last_h = h;
    ctx->last_w = w;
    ctx->last_A = ctx->cur_A;
    return MDHIP_OK;
}

int mdhip_time_op(mdhip_ctx* ctx, int op, int n, int h, int w, int iters, float* ms_avg, void* hip_stream) {
    if (!ctx || !ms_avg || op < 0 || op >= (int)ctx->ops.size() || iters < 1) return MDHIP_EINVAL;
    if (int rc = check_shape(ctx, n, h, w)) return rc;
    if (int rc = check_calibrated(ctx)) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    while (ctx->events.size() < 2) {
        hipEvent_t ev;
        HIP_TRY(ctx, hipEventCreate(&ev));
        ctx->events.push_back(ev);
    }
    ctx->cur_tta = DecodeTta();
    ctx->cur_A = num_anchors_for(ctx, h, w);
    // one op in isolation: a bottleneck's two convs as the two launches they are (no fusion)
    struct Suspend { mdhip_ctx* c; Suspend(mdhip_ctx* c_) : c(c_) { c->fuse_suspended = true; } ~Suspend() { c->fuse_suspended = false; } } suspend(ctx);
    ctx->ops[op].dec_done = false;                                     // (a decode op on its own: launched, whatever the last forward did)
    if (int rc = run_op(ctx, ctx->ops[op], n, h, w, s)) return rc;    // warm
    HIP_TRY(ctx, hipEventRecord(ctx->events[0], s));
    for (int i = 0; i < iters; ++i)
        if (int rc = run_op(ctx, ctx->ops[op], n, h, w, s)) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->events[1], s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->events[0], ctx->events[1]));
    *ms_avg = ms / iters;
    return MDHIP_OK;
}

static int nms_common(mdhip_ctx* ctx, const float* pred_dev, int n, int n_anchors, float conf_thres,
                      float iou_thres, int max_det, float* out, int32_t* counts, hipStream_t s) {
    if (!out || !counts) return fail(ctx, MDHIP_EINVAL, "out/counts is NULL");
    if (n < 1 || n > ctx->max_batch) return fail(ctx, MDHIP_EINVAL, "batch %d outside [1,%d]", n, ctx->max_batch);
    if (max_det < 1 || max_det > kNmsMaxDet) return fail(ctx, MDHIP_EINVAL, "max_det %d outside [1,%d]", max_det, kNmsMaxDet);
    if (n_anchors < 1 || n_anchors > ctx->a_cap) return fail(ctx, MDHIP_EINVAL, "n_anchors %d outside [1,%d]", n_anchors, ctx->a_cap);
    float* out_dev = (float*)(ctx->arena + ctx->nms_out_off);
    int* cnt_dev = (int*)(ctx->arena + ctx->nms_cnt_off);
    HIP_TRY(ctx, launch_nms(pred_dev, n, n_anchors, ctx->no, conf_thres, iou_thres, max_det, ctx->nms_scr, out_dev, cnt_dev, s));
    HIP_TRY(ctx, hipMemcpyAsync(out, out_dev, (size_t)n * max_det * 6 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(counts, cnt_dev, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MDHIP_OK;
}

int mdhip_nms(mdhip_ctx* ctx, int n, float conf_thres, float iou_thres, int max_det, float* out,
              int32_t* counts, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (ctx->last_h == 0) return fail(ctx, MDHIP_EINVAL, "mdhip_nms before mdhip_forward");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int A = ctx->last_A;
    return nms_common(ctx, (const float*)(ctx->arena + ctx->pred_off), n, A, conf_thres, iou_thres,
                      max_det, out, counts, (hipStream_t)hip_stream);
}

int mdhip_nms_enqueue(mdhip_ctx* ctx, int n, float conf_thres, float iou_thres, int max_det, int slot,
                      void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (ctx->last_h == 0) return fail(ctx, MDHIP_EINVAL, "mdhip_nms_enqueue before mdhip_forward");
    if (slot < 0 || slot >= MDHIP_NMS_SLOTS) return fail(ctx, MDHIP_EINVAL, "slot %d outside [0,%d)", slot, MDHIP_NMS_SLOTS);
    if (n < 1 || n > ctx->max_batch) return fail(ctx, MDHIP_EINVAL, "batch %d outside [1,%d]", n, ctx->max_batch);
    if (max_det < 1 || max_det > kNmsMaxDet) return fail(ctx, MDHIP_EINVAL, "max_det %d outside [1,%d]", max_det, kNmsMaxDet);
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int A = ctx->last_A;
    float* out_dev = (float*)(ctx->arena + ctx->nms_out_off);
    int* cnt_dev = (int*)(ctx->arena + ctx->nms_cnt_off);
    HIP_TRY(ctx, launch_nms((const float*)(ctx->arena + ctx->pred_off), n, A, ctx->no, conf_thres, iou_thres, max_det,
                            ctx->nms_scr, out_dev, cnt_dev, s));
    HIP_TRY(ctx, hipEventRecord(ctx->pred_read[ctx->pred_cur], s));
    ctx->pred_read_valid[ctx->pred_cur] = true;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->nms_host_out[slot], out_dev, (size_t)n * max_det * 6 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->nms_host_cnt[slot], cnt_dev, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(ctx->nms_ev[slot], s));
    ctx->nms_slot_n[slot] = n;
    return MDHIP_OK;
}

int mdhip_nms_wait(mdhip_ctx* ctx, int slot, const float** out, const int32_t** counts) {
    if (!ctx || !out || !counts) return MDHIP_EINVAL;
    if (slot < 0 || slot >= MDHIP_NMS_SLOTS || ctx->nms_slot_n[slot] == 0)
        return fail(ctx, MDHIP_EINVAL, "nothing enqueued in slot %d", slot);
    HIP_TRY(ctx, hipEventSynchronize(ctx->nms_ev[slot]));
    *out = ctx->nms_host_out[slot];
    *counts = ctx->nms_host_cnt[slot];
    return MDHIP_OK;
}

int mdhip_nms_on(mdhip_ctx* ctx, const float* pred, int n, int n_anchors, float conf_thres,
                 float iou_thres, int max_det, float* out, int32_t* counts, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (!pred) return fail(ctx, MDHIP_EINVAL, "pred is NULL");
    if (n < 1 || n > ctx->max_batch || n_anchors < 1 || n_anchors > ctx->a_cap)
        return fail(ctx, MDHIP_EINVAL, "n=%d n_anchors=%d outside the context capacity (%d x %d)", n, n_anchors, ctx->max_batch, ctx->a_cap);
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* pred_dev = (float*)(ctx->arena + ctx->pred_off);
    HIP_TRY(ctx, hipMemcpyAsync(pred_dev, pred, (size_t)n * n_anchors * ctx->no * 4, hipMemcpyHostToDevice, s));
    ctx->last_A = n_anchors;          // the context's prediction is now this tensor
    return nms_common(ctx, pred_dev, n, n_anchors, conf_thres, iou_thres, max_det, out, counts, s);
}

int mdhip_read_predictions(mdhip_ctx* ctx, int n, float* out, void* hip_stream) {
    if (!ctx || !out) return MDHIP_EINVAL;
    if (ctx->last_h == 0 || n < 1 || n > ctx->last_n) return fail(ctx, MDHIP_EINVAL, "no forward result for n=%d", n);
    hipStream_t s = (hipStream_t)hip_stream;
    const int A = ctx->last_A;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->arena + ctx->pred_off, (size_t)n * A * ctx->no * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MDHIP_OK;
}

int mdhip_read_input(mdhip_ctx* ctx, int n, int h, int w, float* out, void* hip_stream) {
    if (!ctx || !out) return MDHIP_EINVAL;
    if (int rc = check_shape(ctx, n, h, w)) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    float* tmp = nullptr;
    const size_t bytes = (size_t)n * 3 * h * w * 4;
    HIP_TRY(ctx, hipMalloc((void**)&tmp, bytes));
    hipError_t e = launch_s2d_to_nchw_f32((const uint16_t*)(ctx->arena + ctx->input.off), tmp, n, h, w, ctx->dtype == MDHIP_DTYPE_FP16, s);
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    HIP_TRY(ctx, e);
    return MDHIP_OK;
}

int mdhip_read_layer(mdhip_ctx* ctx, int layer, int n, float* out, int* c, int* h, int* w, void* hip_stream) {
    if (!ctx) return MDHIP_EINVAL;
    if (layer < 0 || layer >= (int)ctx->layer_out.size() || !ctx->layer_out[layer].valid)
        return fail(ctx, MDHIP_EINVAL, "layer %d has no readable output", layer);
    if (ctx->last_h == 0) return fail(ctx, MDHIP_EINVAL, "mdhip_read_layer before mdhip_forward");
    const Tensor& t = ctx->layer_out[layer];
    const int H = ctx->last_h / t.div, W = ctx->last_w / t.div;
    if (c) *c = t.c;
    if (h) *h = H;
    if (w) *w = W;
    if (!out) return MDHIP_OK;
    if (n < 1 || n > ctx->last_n) return fail(ctx, MDHIP_EINVAL, "n=%d outside the last forward's batch %d", n, ctx->last_n);
    hipStream_t s = (hipStream_t)hip_stream;
    float* tmp = nullptr;
    const size_t bytes = (size_t)n * t.c * H * W * 4;
    HIP_TRY(ctx, hipMalloc((void**)&tmp, bytes));
    hipError_t e = launch_nhwc_to_nchw_f32((const uint16_t*)(ctx->arena + t.off), t.ld, tmp, n, t.c, H, W, ctx->dtype == MDHIP_DTYPE_FP16, s);
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    HIP_TRY(ctx, e);
    return MDHIP_OK;
}

int mdhip_num_ops(mdhip_ctx* ctx) { return ctx ? (int)ctx->ops.size() : MDHIP_EINVAL; }

int mdhip_get_op_info(mdhip_ctx* ctx, int op, mdhip_op_info* out) {
    if (!ctx || !out || op < 0 || op >= (int)ctx->ops.size()) return MDHIP_EINVAL;
    const Op& o = ctx->ops[op];
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", o.name.c_str());
    out->kind = o.kind;
    out->layer = o.layer;
    out->m = o.gm;
    out->n = o.gn;
    out->k = o.gk;
    out->flops = o.flops;
    out->bytes = o.bytes;
    out->cfg = o.last_cfg;
    if (o.kind == OP_CONV) {
        const PackedConv& pc = ctx->packed[o.pc];
        out->ntaps = pc.kh * pc.kw;
        out->stride = o.stride;
        out->has_res = o.has_res ? 1 : 0;
    }
    return MDHIP_OK;
}

int mdhip_num_conv_cfgs(void) { return conv_num_cfgs(); }

int mdhip_op_supports_cfg(mdhip_ctx* ctx, int op, int cfg) {
    if (!ctx || op < 0 || op >= (int)ctx->ops.size()) return MDHIP_EINVAL;
    if (ctx->ops[op].kind != OP_CONV || cfg < 0 || cfg >= conv_num_cfgs()) return 0;
    ConvArgs a{};
    const int h = ctx->last_h ? ctx->last_h : ctx->max_stride, w = ctx->last_w ? ctx->last_w : ctx->max_stride;
    fill_conv_args(ctx, ctx->ops[op], ctx->last_n ? ctx->last_n : 1, h, w, a);
    return conv_api(ctx).supports(cfg, a) ? 1 : 0;
}

const char* mdhip_conv_cfg_name(int cfg) { return (cfg >= 0 && cfg < conv_num_cfgs()) ? conv_cfg(cfg).name : ""; }

int mdhip_cfg_is_bitwise(int cfg) { return (cfg >= 0 && cfg < conv_num_cfgs() && conv_cfg_is_bitwise_family(cfg)) ? 1 : 0; }

int mdhip_set_tuned(mdhip_ctx* ctx, const mdhip_tuned* entries, int n) {
    if (!ctx || n < 0 || (n > 0 && !entries)) return MDHIP_EINVAL;
    for (int i = 0; i < n; ++i)
        if (entries[i].cfg < 0 || entries[i].cfg >= conv_num_cfgs())
            return fail(ctx, MDHIP_EINVAL, "tuned entry %d: cfg %d outside [0,%d)", i, entries[i].cfg, conv_num_cfgs());
    ctx->tuned.assign(entries, entries + n);
    for (Op& op : ctx->ops) op.memo_cfg = -1;
    drop_graphs(ctx);
    return MDHIP_OK;
}

int mdhip_set_fuse(mdhip_ctx* ctx, int on) {
    if (!ctx) return MDHIP_EINVAL;
    ctx->fuse_enabled = on != 0;
    drop_graphs(ctx);
    return MDHIP_OK;
}

int mdhip_set_option(mdhip_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return MDHIP_EINVAL;
    if (!strcmp(name, "letterbox_general")) ctx->letterbox_general = value != 0;
    else if (!strcmp(name, "fuse_decode")) ctx->fuse_decode = value != 0;
    else return fail(ctx, MDHIP_EINVAL, "unknown option '%s'", name);
    drop_graphs(ctx);
    return MDHIP_OK;
}

int mdhip_set_graph(mdhip_ctx* ctx, int mode, int max_n) {
    if (!ctx || mode < 0 || mode > 2) return MDHIP_EINVAL;
    ctx->graph_mode = mode;
    if (max_n > 0) ctx->graph_max_n = max_n;
    if (mode == 0) drop_graphs(ctx);
    return MDHIP_OK;
}

int mdhip_set_op_cfg(mdhip_ctx* ctx, int op, int cfg) {
    if (!ctx || op < 0 || op >= (int)ctx->ops.size()) return MDHIP_EINVAL;
    if (ctx->ops[op].kind != OP_CONV) return fail(ctx, MDHIP_EINVAL, "op %d is not a conv", op);
    if (cfg < -1 || cfg >= conv_num_cfgs()) return fail(ctx, MDHIP_EINVAL, "cfg %d outside [-1,%d)", cfg, conv_num_cfgs());
    ctx->ops[op].forced_cfg = cfg;
    drop_graphs(ctx);
    return MDHIP_OK;
}

}  // extern "C"
