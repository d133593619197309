// Shared host-side pieces of libdfd_hip.so: status plumbing, the weights-blob reader and
// the handle.  Nothing here crosses the C ABI; include/dfd_hip.h is the public surface.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dfd_hip.h"
#include "blob_reader.h"
#include "imgproc_kernels.h"

namespace dfd {

struct ForensicState;   // forensic_api.hip
void forensic_destroy(dfd_handle* h);
struct FreqState;       // freq_kernels.hip
void freq_destroy(dfd_handle* h);
struct SsdState;        // ssd_api.hip
int ssd_init(dfd_handle* h);
void ssd_destroy(dfd_handle* h);
int ssd_warmup(dfd_handle* h, int n_frames);   // detector forward on n synthetic 300x300 inputs (tile measurement)
struct HaarState;       // haar_api.hip: Haar cascade fallback detector
int haar_init(dfd_handle* h);
void haar_destroy(dfd_handle* h);
struct CommState;       // comm_api.hip: RCCL communicator of the vote exchange
void comm_destroy(dfd_handle* h);
struct S6Table;         // gemm_split.hip: measured split-GEMM tiles of this handle
struct MtcnnState;      // mtcnn_api.hip
int mtcnn_init(dfd_handle* h);
void mtcnn_destroy(dfd_handle* h);

// Tensor + parse_blob: blob_reader.h (plain C++, also built into the sanitizer harness)

struct B0Block {
    int kernel, stride, expand, c_in, c_out, c_exp, c_se, h_in, h_out, pad_lo;
    bool skip;
    const float *exp_w, *exp_b, *dw_w, *dw_b, *se_w1, *se_b1, *se_w2, *se_b2, *proj_w, *proj_b;
    int dw_tiles;
};

struct B0Plan {
    std::vector<B0Block> blocks;
    const float *stem_w, *stem_b, *head_w, *head_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *fc3_w, *fc3_b;
    // workspace (floats per image)
    size_t io_floats = 0, exp_floats = 0, dw_floats = 0, pool_floats = 0, gate_floats = 0;
};

// Pixel budget of ONE batched request (dfd_analyze_stream_batch / jpeg_decode_batch_to): 2^27 = 64 frames of 1080p.  A
// flat 8192 x 8192 JPEG is ~1 MB, so a 32-part request inside the server's body limit could otherwise ask for 13 GB of
// pinned coefficients + 6 GB of frames before anything is rejected; the check runs on the parsed headers, before any
// allocation (the server mirrors it: backend_server.MAX_BATCH_PIXELS).
constexpr size_t kMaxBatchPixels = (size_t)1 << 27;

// a device buffer that only ever grows (re-allocated outside of steady state)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct B0Tap {
    const char* name = nullptr;   // stage to copy out, or null
    float* out = nullptr;
    size_t capacity = 0, count = 0;
    bool found = false;
};
struct B0Prof {
    std::vector<hipEvent_t> events;
    std::vector<const char*> names;
    bool enabled = false;
};

}  // namespace dfd

struct dfd_handle {
    int device = 0;
    int max_batch = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t order_ev = nullptr;       // dfd_wait_for: recorded on another handle's stream, waited for by this one
    // dfd_analyze_frames_host: copy stream + two staging slots (uploads overlap the previous batch's compute)
    // pinned mailbox for the small host<->device transfers of the batch path (mailbox_* below)
    char* mailbox = nullptr;
    size_t mailbox_cap = 0, mailbox_head = 0, mailbox_lap_end = 0;
    size_t mailbox_live = 0, mailbox_live_prev = 0;   // bytes handed out since the last / between the last two stream_sync
    std::vector<char*> mailbox_old, mailbox_old_prev; // replaced blocks: freed two stream_syncs after their retirement
    // second compute stream: the forensic launch set of a batch call runs beside detector / classifier
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_go = nullptr, aux_done = nullptr;
    bool overlap_forensics = true;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done[2] = {nullptr, nullptr}, slot_free[2] = {nullptr, nullptr};
    dfd::DevBuf stage[2];
    std::map<std::string, dfd::Tensor> tensors;
    std::vector<void*> owned;            // every hipMalloc the handle must free
    dfd::B0Plan b0;
    // classifier workspace
    float *in_nchw = nullptr, *io0 = nullptr, *io1 = nullptr, *expbuf = nullptr, *dwbuf = nullptr;
    float *pool = nullptr, *gate = nullptr, *headbuf = nullptr, *feat = nullptr, *fc1 = nullptr,
          *fc2 = nullptr, *logits = nullptr;
    // image pre-processing: colour LUTs + lazily grown scratch (frame upload, Lab/BGR crops)
    dfd::ColorTables color{};
    bool has_color = false;
    dfd::DevBuf frame_buf, lab_buf, crop_buf, lut_buf, desc_buf, u8_out, face_batch;
    int last_detections = 0;             // detections of the last single-frame detector run, before the max_out cut
    std::vector<char> crop_valid;        // per crop of the last preprocess: 0 = the MTCNN stage found no face
    int n_compact = 0;                   // rows of in_nchw the last preprocess filled (= crops with a face, in crop order, when it compacts)
    unsigned long long classifier_crops = 0;   // crops b0_forward has been asked for since dfd_create (dfd_classifier_crop_count)
    dfd::ForensicState* forensic = nullptr;   // per-stream temporal state + work buffers
    dfd::FreqState* freq = nullptr;           // compute_frequency_features tables + scratch
    dfd::SsdState* ssd = nullptr;             // detector plan + workspace (null: blob has no detector)
    dfd::MtcnnState* mtcnn = nullptr;         // MTCNN cascade (null: blob has none)
    dfd::HaarState* haar = nullptr;           // Haar cascade (null: blob has none)
    dfd::CommState* comm = nullptr;           // vote exchange (null until dfd_comm_init)
    bool use_mtcnn = true;                    // classify paths align each crop with the cascade when the blob has one
    bool fuse_stem = true;               // stem conv computed inside block 0's depthwise kernel
    bool fuse_expand = true;             // MBConv blocks 1-5: expand conv computed inside the depthwise kernel
    bool fuse_late = true;               // blocks 6-10 / 12-15: expand + depthwise of whole images in one launch (mbconv_late_kernel):
                                         // the faster configuration (round 3: +2.2-2.6 % per step, strictly fewer bytes), default since round 4
    unsigned fuse_late_skip = (1u << 8) | (1u << 9);   // blocks that keep expand GEMM + depthwise kernel although fuse_late is on:
                                         // measured per block at batch 256 (fused - separate, us): b6 -17.8, b7 -5.5, b8 +2.1, b9 +3.6,
                                         // b10 -1.4, b12 -9.8, b13 -4.9, b14 -4.7, b15 -2.3 (option "fuse_late_skip", a bit per block)
    bool fuse_se = false;                // squeeze-excite gate computed by the last block of each image inside the depthwise launch
                                         // (measured slower than the separate launch: DESIGN.md section 5, round 3; kept as an option)
    bool se_in_proj = false;             // squeeze-excite gate evaluated by the projection GEMM's blocks where the pool sums are final
                                         // per image (blocks 6-10 / 12-15 with fuse_late): no se_kernel launch there.  Built and
                                         // measured in round 4: gates bit-identical, step SLOWER - off
    bool se_thin = false;                // blocks 0-4: the gate evaluated by the narrow projection's own blocks (pw8_kernel prologue),
                                         // no se_kernel launch there.  Built and measured in round 4: the five projections +85 us
                                         // against 54 us of se_kernel launches saved (3.08 vs 2.99 ms per step) - off (option "se_thin")
    unsigned* se_counter = nullptr;      // [max_batch] arrival counters of that hand-off (zero between launches)
    bool split_gemm = true;              // 1x1 / k x k convs on the bf16x3-split MFMA path (gemm_split.hip)
    bool act_bf16 = false;               // classifier activations stored as bf16 (fp32 arithmetic): configs[3]
    int bf16_planes = 3;                 // weight planes the bf16-activation GEMMs use: 3 = fp32-exact weights, 1 = bf16 weights
    dfd::DevBuf jpeg_work;               // dfd_decode_jpeg: coefficients, component planes, quantisation tables
    dfd::DevBuf jpeg_raw[2];             // dfd_analyze_jpegs_host: the scans of the chunk being decoded / being uploaded
    dfd::DevBuf jpeg_work2[2];           // ... and the decoder's scratch of the chunk being decoded / the one before
    hipStream_t jpeg_stream = nullptr;   // ... whose decode runs beside the analysis of the previous chunk (= aux_stream)
    hipEvent_t jpeg_done[2] = {nullptr, nullptr}, frames_free[2] = {nullptr, nullptr};
    void* jpeg_host = nullptr;           // pinned host buffer the entropy decoder writes the coefficients into
    size_t jpeg_host_cap = 0;
    int jpeg_device_entropy = 2;         // batch calls: 2 = restart-less JPEGs are entropy-decoded on the device (jpeg_gpu_entropy.h) from
                                         // 1 MiB of scan data per call (below that the host pool is quicker), 1 = always, 0 = never
    int jpeg_rounds = 16;                // rounds of that decoder's fixed-point iteration (option "jpeg_rounds", 2 .. 32; converged rounds cost ~nothing)
    int jpeg_chunk_bytes = 512;          // bytes of de-stuffed scan per lane of that decoder (option "jpeg_chunk_bytes", >= 256, % 4)
    unsigned long long jpeg_frames_device = 0, jpeg_frames_host = 0;   // frames of batch calls decoded there / by the host decoder
    dfd::DevBuf tap_buf;                 // fp32 staging for taps of bf16 buffers
    std::map<const float*, unsigned short*> wsplit;   // fp32 weight tensor -> its three-plane bf16 split
    dfd::S6Table* gemm = nullptr;        // split-GEMM tile per shape (measured by dfd_warmup, heuristic otherwise)
    dfd::B0Prof prof;                    // layer events between profile_begin/end
    int prof_steps = 0;                  // forwards that carried events
    int prof_seen = 0, prof_stride = 1;  // forwards since profile_begin; every prof_stride-th one is instrumented
    std::string err;
};

namespace dfd {

extern thread_local std::string g_create_error;

int fail(dfd_handle* h, int code, const char* fmt, ...);

#define DFD_HIP_TRY(h, expr)                                                                    \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return dfd::fail((h), DFD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                               \
    } while (0)

// grows `b` to at least `bytes` (frees and re-allocates; contents are not preserved)
int ensure(dfd_handle* h, DevBuf* b, size_t bytes);
// Small transfers on the compute stream without the DMA engines: descriptors, detection rows and logits are a few KB,
// but as hipMemcpyAsync they queue on the same SDMA engine as the 200-400 MB frame upload of the next batch and the
// batch in flight stalls until that upload has finished (measured: 32-frame batches ran at upload + compute, not at
// max(upload, compute)).  A copy kernel moves them through a pinned, device-visible mailbox instead.
//   mailbox_h2d: `src` is copied into the mailbox before the call returns (reusable at once); the device copy is enqueued.
//   mailbox_d2h: enqueues the copy and returns the host address that holds the data once the stream is synchronised.
// Every wait on the handle's stream goes through stream_sync: it is the point after which mailbox regions handed out
// before the PREVIOUS sync can be reused (the host reads a d2h region right after the sync that completes it).
hipError_t stream_sync(dfd_handle* h);
int mailbox_h2d(dfd_handle* h, void* dst_dev, const void* src, size_t bytes);
const void* mailbox_d2h(dfd_handle* h, const void* src_dev, size_t bytes);
void copy_kernel_async(void* dst, const void* src, size_t bytes, hipStream_t s);
// builds h->color from the "lut.*" tensors of the blob (imgproc_api.hip)
int color_tables_init(dfd_handle* h);

// stages that work on a frame already resident in HBM (forensic_api / ssd_api / imgproc_api)
int forensics_run(dfd_handle* h, int stream_id, const uint8_t* frame_dev, int hh, int ww, int stride, int full,
                  double* scores_out, double* prob_out, double* stats_out);
int detect_run(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, float conf_thr, int32_t* xywh_out,
               float* conf_out, int max_out, int* n_out);
// haar_api.hip: detectMultiScale + groupRectangles on a resident frame; *n_total = groups before the max_out cut
int haar_run(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, float scale_factor, int min_neighbors,
             int min_size, int32_t* xywh_out, int max_out, int* n_out, int* n_candidates, int* n_total = nullptr);
// frame_offs: per-crop byte offset of its frame inside frame_dev (null = single frame)
int preprocess_run(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, const int32_t* xywh, int n,
                   int apply_clahe, const size_t* frame_offs = nullptr);
// boxes of resident frame(s) -> logits on the host, m <= max_batch: crop / CLAHE / MTCNN / 224 x 224, then the classifier
// at the batch of the crops the cascade KEPT (reference deepfake_detection.py:377-380: `mtcnn()` -> None returns before
// the model runs) - a rejected crop costs no classifier work and gets NaN.  Ends with a stream wait.
int classify_boxes(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, const int32_t* xywh, int m,
                   int apply_clahe, const size_t* frame_offs, float* logits_out);
// DetectionOutput of `n` frames already resized to 300x300 -> rows/count on the host
int detect_batch_run(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes,
                     float conf_thr, int max_faces, int32_t* xywh_out, int* n_out, int* n_total_out = nullptr);
// the analyzer over n consecutive frames of ONE stream (its temporal state advances n frames): frame i scored in full
// or fast mode as full[i] says - results identical to n forensics_run calls in order
int forensics_stream_batch_run(dfd_handle* h, int stream_id, const uint8_t* frames_dev, int n, int hh, int ww, int stride,
                               size_t frame_bytes, const int* full, double* scores_out, double* prob_out);
// jpeg_decode.hip: n JPEGs of one size -> packed BGR frames [n][hh][ww][3] at frames_dev (entropy decoding of the files
// in parallel on the host pool, one coefficient upload, IDCT / colour per frame).  *hh / *ww: in = expected size or 0
int jpeg_decode_batch_to(dfd_handle* h, const uint8_t* const* jpegs, const size_t* lens, int n, uint8_t* frames_dev_or_null,
                         int* hh, int* ww);
// stateless six-signal forensic probability of `n` device frames (temporal signal = first-frame value 0)
int forensics_batch_run(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes,
                        double* prob_out, double* scores_out);
// the same in two halves: begin enqueues the launch set on the handle's second stream (ordered after what the main
// stream holds at that moment) and returns; end waits for it and scores on the host.  Nothing else may use the
// forensic work buffers in between.
int forensics_batch_begin(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes);
int forensics_batch_end(dfd_handle* h, int n, double* prob_out, double* scores_out);

// mtcnn_api.hip: MTCNN.forward on a BGR image in HBM -> selected box (x1,y1,x2,y2,prob), *found, and the
// 160x160 BGR u8 crop at mtcnn_face_dev(h).  tap_* are for parity tests (null otherwise).
struct MtImage { const uint8_t* src; int h, w; size_t stride; };
// the same for `n` images of a step at once: faces_out [n][160*160*3] (device), boxes_out [n][5] or null, found [n]
// returns after the step's last stream wait: `found` is final (the classifier is then sized by the faces that are left)
int mtcnn_align_batch_device(dfd_handle* h, const MtImage* imgs, int n, uint8_t* faces_out, float* boxes_out, char* found,
                             const char* tap_name, std::vector<float>* tap, int* tap_dims);
int mtcnn_align_device(dfd_handle* h, const uint8_t* img_dev, int hh, int ww, size_t stride, float* box_out, int* found,
                       const char* tap_name, std::vector<float>* tap, int* tap_dims);
const uint8_t* mtcnn_face_dev(dfd_handle* h);

// imgproc_api.hip: the per-frame work on a frame that is already in h->frame_buf
int analyze_frame_resident(dfd_handle* h, int stream_id, int hh, int ww, int stride, int full_forensics, float conf_thr,
                           int max_faces, int apply_clahe, double* scores_out, double* forensic_prob_out, int32_t* xywh_out,
                           int* n_faces_out, float* logits_out);

// jpeg_decode.hip: baseline JPEG bytes -> packed BGR frame in h->frame_buf (stride width * 3)
int jpeg_decode_to_frame(dfd_handle* h, const uint8_t* jpeg, size_t len, int* hh, int* ww);

// b0_plan.hip
// three-plane bf16 split of a weight tensor of the handle, made on first use; null (+ error set) on failure
const unsigned short* split_weights(dfd_handle* h, const float* W, int N, int K, bool transposed = false);
// 1x1 conv through the split path when enabled and the shape allows, else the fp32 MFMA kernel
int pointwise(dfd_handle* h, const float* X, const float* W, const float* bias, const float* gate, const float* R,
              float* Y, int M, int K, int N, int HW, int act);
int b0_build_plan(dfd_handle* h);
// Runs the classifier on h->stream.  `stop_at_features`: leave after the pooled 1280-vector.
int b0_forward(dfd_handle* h, const float* nchw_dev, int n, float* logits_dev, B0Tap* tap,
               B0Prof* prof);

}  // namespace dfd
