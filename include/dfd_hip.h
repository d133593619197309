/*
 * dfd_hip.h - C ABI of libdfd_hip.so, the MI355X (gfx950) implementation of the
 * per-frame deepfake inference hot path.
 *
 * The reference (KrishTanna28/Real-Time-Video-Deepfake-Detection) is pure Python
 * and has no FFI of its own; each entry point below names the
 * reference call site whose arithmetic it replaces.  The ctypes binding that a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success and a negative dfd_status on failure;
 *     nothing throws or aborts across this boundary; the message for the last
 *     failure on a handle is dfd_last_error(handle) (dfd_last_error(NULL) for
 *     failures of dfd_create itself);
 *   - the caller owns every host buffer; pointers are borrowed for the call only;
 *   - "_device" variants take pointers into the handle's GPU (from dfd_device_alloc
 *     or any hipMalloc'd / torch CUDA tensor on that device), enqueue on the
 *     handle's HIP stream and return without synchronising;
 *   - a handle is not re-entrant: one caller thread per handle.
 */
#ifndef DFD_HIP_H
#define DFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dfd_handle dfd_handle;

typedef enum dfd_status {
    DFD_OK = 0,
    DFD_ERR_ARG = -1,      /* bad argument (null, size, shape)            */
    DFD_ERR_BLOB = -2,     /* weights blob malformed or a tensor missing  */
    DFD_ERR_HIP = -3,      /* a HIP runtime call failed                   */
    DFD_ERR_NO_DEVICE = -4,/* no usable gfx950 device                     */
    DFD_ERR_STATE = -5,    /* call order (e.g. detector weights not set)  */
    DFD_ERR_CAPACITY = -6, /* batch larger than the handle was created for*/
    DFD_ERR_UNSUPPORTED = -7 /* valid input of a kind this path does not handle (e.g. progressive JPEG) */
} dfd_status;

#define DFD_ABI_VERSION 1
#define DFD_CROP 224          /* classifier input edge, reference deepfake_detection.py:383 */
#define DFD_FEATURES 1280     /* backbone feature width, reference model.py:46              */

int dfd_abi_version(void);

/* ---- lifetime ------------------------------------------------------------------
 * Replaces the import-time model construction + weight load of reference
 * deepfake_detection.py:30-90 (DeepfakeEfficientNet + load_state_dict + .to(DEVICE).eval()).
 * `blob` is the packed classifier produced by weights.pack_b0 (BatchNorm folded,
 * NHWC layouts).  `max_batch` sizes the activation workspace (9.6 MB of HBM per crop). */
int dfd_create(int device, const void* blob, size_t blob_len, int max_batch, dfd_handle** out);
void dfd_destroy(dfd_handle* h);
const char* dfd_last_error(const dfd_handle* h);
int dfd_max_batch(const dfd_handle* h);
/* Tuning switches (results stay within the parity tolerances either way):
 *   "fuse_expand" (default 1, env DFD_FUSE_EXPAND): MBConv blocks 1-5 compute the 1x1 expand conv
 *   inside the depthwise kernel instead of writing the expanded tensor to HBM.
 *   "fuse_late" (default 1 since round 4, env DFD_FUSE_LATE; needs "fuse_expand"): blocks 6-10 and 12-15 (14 x 14 / 7 x 7
 *   maps) do the same with whole images per thread block - the faster configuration; 0 = expand
 *   GEMM and depthwise kernel as separate launches.
 *   "fuse_late_skip" (default blocks 8 and 9: bit b set = block b keeps separate launches although "fuse_late" is on; chosen
 *   per block by measurement at batch 256).
 *   "se_in_proj" (default 0, env DFD_SE_IN_PROJ; measured slower, kept for the measurement): where a depthwise launch leaves final per-image pool sums (the
 *   whole-image launches of "fuse_late") the projection GEMM's blocks evaluate the squeeze-excite gate themselves
 *   (se_kernel's arithmetic, identical gate bits) instead of a separate launch per block.
 *   "se_thin" (default 0; measured slower, kept for the measurement): blocks 0-4 - the blocks of the narrow projection
 *   kernel (pw8_kernel) evaluate the squeeze-excite gate of the images they meet in a prologue, no se_kernel launch.
 *   "fuse_stem" (default 1, env DFD_FUSE_STEM): the stem conv is computed inside block 0's depthwise
 *   kernel (the 112x112x32 stem activation stays in LDS).
 *   "split_gemm" (default 1, env DFD_SPLIT_GEMM): 1x1 convs (N >= 16) and the detector's k x k convs run on
 *   the split-precision GEMM (each fp32 operand = exact sum of three bf16 terms, six products on the bf16
 *   MFMA, fp32 accumulate: fp32-dot-product accuracy); 0 = the fp32 MFMA kernel everywhere.
 *   "mtcnn" (default 1): align every crop with the MTCNN cascade when the blob carries one.
 *   "overlap_forensics" (default 1): dfd_analyze_batch_device / dfd_analyze_frames_host run the six forensic signals of
 *   a batch on the handle's second stream beside the detector and the classifier and collect them at the end of the call;
 *   0 = in front of the detector on the main stream.  Same results.
 *   "profile_stride" (default 1): between dfd_b0_profile_begin/end only every k-th forward records events.
 *   "stream_priority" (1 high, 0 normal - the default -, -1 low): re-creates the handle's main stream at that priority
 *   (the handle is drained first).  For a process that keeps two handles busy on one device (two batches in flight): the
 *   runtime maps streams of ONE priority onto a small pool of hardware queues and two main streams that land on the same
 *   queue run in line; streams of different priorities come from different pools. */
int dfd_set_option(dfd_handle* h, const char* name, int value);
/*   "fuse_se" (default 0): the squeeze-excite gate is computed by the last-arriving block of each image inside the
 *   depthwise launch (measured slower than the separate launch: DESIGN.md section 5; kept for the measurement).
 *   "bf16_activations" (default 0): every classifier activation that reaches HBM is stored as bf16 (arithmetic,
 *   accumulators, SE pools / gates and the MLP head stay fp32): BASELINE.json configs[3], DESIGN.md section 4a.
 *   "bf16_weight_planes" (3 or 1, default 3): with bf16 activations, the 1x1 convs multiply against the three exact
 *   bf16 planes of the fp32 weights (3) or against bf16-rounded weights (1).
 *   "gemm_tile" (default -1): >= 0 forces split-GEMM instance number value % (candidates of the shape) for every
 *   1x1 / k x k conv - parity tests walk 0 .. dfd_gemm_tile_count()-1 and require identical bits; -1 = the
 *   handle's tile table (measured by dfd_warmup, heuristic for shapes it has not seen). */
int dfd_gemm_tile_count(void);

/* One untimed pass over the classifier at batch `n_crops` (0 = skip; <= max_batch) and the detector at
 * `n_frames` frames (0 = skip) on synthetic data.  Sizes workspaces, splits the weights and MEASURES the
 * split-GEMM tile of every layer shape at those batch sizes (the only entry point that synchronises for
 * tuning; the "_device" entry points never do - a shape that was not warmed up runs a heuristic tile, with
 * the same result bits).  Call once per (n_crops, n_frames) you intend to serve; DFD_S6_TUNE=0 in the
 * environment skips the measurement. */
int dfd_warmup(dfd_handle* h, int n_crops, int n_frames);

/* The handle's measured tiles as text (one "M K N mode kind wm wn mt nt ks" line per shape) and back: a later
 * process - a profiled run, a server restart - imports them and dfd_warmup then measures only what is missing.
 * text_out == NULL queries the length.  Entries that name an instance this build cannot launch are dropped. */
int dfd_tiles_export(dfd_handle* h, char* text_out, size_t capacity, size_t* length);
int dfd_tiles_import(dfd_handle* h, const char* text, size_t length, int* accepted);

/* Host-only arithmetic of the split GEMM's 32-bit addressing guard: how many of `rows` rows of `row_bytes`
 * bytes one kernel launch may cover (a multiple of rows_per_image, whole `rows` when everything fits below
 * 2^31 bytes, -1 when a single image does not).  Batches above that are issued as several launches, so every
 * max_batch dfd_create accepts is addressable.  No GPU needed. */
long long dfd_gemm_chunk_rows(long long rows, long long row_bytes, long long rows_per_image);

/* ---- device memory and stream plumbing (no reference counterpart) -------------- */
int dfd_device_alloc(dfd_handle* h, size_t bytes, void** dptr);
int dfd_device_free(dfd_handle* h, void* dptr);
int dfd_memcpy_h2d(dfd_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int dfd_memcpy_d2h(dfd_handle* h, void* dst_host, const void* src_dev, size_t bytes);
int dfd_sync(dfd_handle* h);
/* Device-side ordering between two handles on the same device, without a host wait: everything queued on `h` after this
 * call starts only when the work queued on `other` so far has finished (an event recorded on other's stream, waited for by
 * h's stream).  bench.py orders its two classifier lanes with it around the step that carries per-launch events.
 * Threading: as every entry point, one caller thread per handle - no other thread may be inside a call on `h` or `other`. */
int dfd_wait_for(dfd_handle* h, dfd_handle* other);
/* device address of the handle's frame buffer: the last frame uploaded by a host-frame entry point or decoded by
 * dfd_decode_jpeg (packed BGR); valid until the next such call */
void* dfd_frame_ptr(dfd_handle* h);
/* HIP events on the handle's own stream (what bench.py times kernels with). */
int dfd_timer_begin(dfd_handle* h);
int dfd_timer_end(dfd_handle* h, float* elapsed_ms);

/* ---- classifier ------------------------------------------------------------------
 * DeepfakeEfficientNet.forward, reference model.py:63-72 (eval mode): normalised RGB
 * float32 NCHW (n,3,224,224) -> logits (n,1).  sigmoid is applied by the caller as at
 * reference deepfake_detection.py:397-398. */
int dfd_classify_nchw(dfd_handle* h, const float* nchw_host, int n, float* logits_host);
int dfd_classify_nchw_device(dfd_handle* h, const float* nchw_dev, int n, float* logits_dev);
/* DeepfakeEfficientNet.extract_features, reference model.py:74-89: -> (n,1280). */
int dfd_extract_features(dfd_handle* h, const float* nchw_host, int n, float* feat_host);

/* Runs the forward on `nchw_dev` and copies one named intermediate to the host
 * (NHWC float32): "stem", "b<i>.exp", "b<i>.dw", "b<i>.gate", "b<i>.out", "head",
 * "feat", "logit".  For stage-by-stage parity tests; `count` receives the number of
 * floats written (<= capacity). */
int dfd_b0_tap(dfd_handle* h, const float* nchw_dev, int n, const char* name,
               float* out_host, size_t capacity, size_t* count);

/* Per-launch timing with HIP events on the handle's stream.  Between begin and end every
 * dfd_classify_nchw_device call records an event after each kernel launch; end synchronises
 * and returns, per launch position, the elapsed milliseconds summed over the `steps`
 * instrumented forwards (every forward, or every "profile_stride"-th one) ("stem", "b<i>.exp", "b<i>.dw", "b<i>.se", "b<i>.proj", "head", "avgpool",
 * "mlp").  `names` receives pointers to static strings. */
int dfd_b0_profile_begin(dfd_handle* h);
int dfd_b0_profile_end(dfd_handle* h, float* ms_sum, const char** names, int max_layers,
                       int* count, int* steps);

/* ---- per-face pre-processing ----------------------------------------------------------
 * All take an 8-bit BGR image in host memory (rows `stride` bytes apart) as cv2 hands it to
 * the reference.  Boxes are (x, y, w, h) int32 quadruples inside the frame. */

/* cv2.resize(frame, (dw, dh), interpolation=INTER_LINEAR) for 8-bit BGR: reference
 * frame_analysis.py:71,112 (256x256) and face_detection.py:77 (300x300).  out: dh*dw*3. */
int dfd_resize_bgr(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                   int dh, int dw, uint8_t* out);

/* DeepfakeDetector.preprocess_face_quality, reference deepfake_detection.py:357-370:
 * BGR->Lab, CLAHE(clipLimit 2.0, 8x8 tiles) on L, Lab->BGR.  out: height*width*3, packed. */
int dfd_preprocess_face_quality(dfd_handle* h, const uint8_t* bgr, int height, int width,
                                int stride, uint8_t* out);

/* One test-time-augmentation copy of a face crop, reference deepfake_detection.py:419-433:
 * cv2.flip(img, 1) when `flip`, cv2.convertScaleAbs(img, alpha=brightness, beta=0), then cv2.warpAffine(img,
 * cv2.getRotationMatrix2D((w/2, h/2), angle_deg, 1.0), (w, h)) with OpenCV's fixed-point bilinear sampling.
 * out: height*width*3, packed.  The random draws (flip probability .5, brightness 0.9..1.1, angle -3..3 degrees)
 * and the averaging of the per-copy probabilities stay on the host (DeepfakeDetector.analyze_face_with_tta). */
int dfd_tta_augment(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride, int flip, double brightness,
                    double angle_deg, uint8_t* out);

/* crop (reference backend_server.py:160-161 / deepfake_detection.py:612) -> optional CLAHE ->
 * BGR->RGB, bilinear 224x224 (align_corners=False), /255, ImageNet normalise (reference
 * deepfake_detection.py:376,382-389).  When the blob carries an MTCNN cascade and option "mtcnn" is on, the re-crop at
 * :377 runs in between (P-/R-/O-Net, best face resampled to 160x160; a crop without a face gives the zero-filled
 * face's row here and NaN from the classify entry points).  nchw_out: (n,3,224,224) float32. */
int dfd_preprocess_crops(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                         const int32_t* xywh, int n, int apply_clahe, float* nchw_out);

/* The same followed by the classifier: one logit per box (analyze_face without the
 * calibration/heuristic scalars, reference deepfake_detection.py:517-538). */
int dfd_classify_crops(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                       const int32_t* xywh, int n, int apply_clahe, float* logits_out);

/* compute_frequency_features, reference model.py:105-149: BGR (channels = 3) or gray (1) 8-bit
 * image -> gray -> cv2.resize 224x224 -> channel 0 = min-max-normalised log1p|fftshift(fft2)|,
 * channel 1 = min-max-normalised log1p|cv2.dct(gray/255)|.  out: (2,224,224) float32.  The model
 * ignores this tensor (reference model.py:63-72); provided for API parity. */
int dfd_frequency_features(dfd_handle* h, const uint8_t* img, int height, int width, int stride,
                           int channels, float* out);

/* ---- face detector ----------------------------------------------------------------------
 * _detect_dnn, reference face_detection.py:71-105, for a blob packed with detector weights
 * (weights.pack_all): cv2.resize to 300x300, mean (104,177,123) subtraction, SSD forward,
 * DetectionOutput (NMS 0.45, top_k 400, keep_top_k 200), then the reference's integer
 * post-processing: conf > conf_thr (strict), scale by [w,h,w,h], truncate, clamp, keep
 * w > 20 and h > 20.  Writes up to max_out (x, y, w, h) quadruples in network (descending
 * confidence) order; frames smaller than 30 pixels in either direction give n_out = 0
 * (reference :55-56).  conf_out may be NULL. */
int dfd_detect_faces(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                     float conf_thr, int32_t* xywh_out, float* conf_out, int max_out, int* n_out);
int dfd_has_detector(const dfd_handle* h);
/* Number of detections of the last dfd_detect_faces / dfd_analyze_frame call BEFORE the max_out / max_faces cut:
 * `len(faces)` of the reference (backend_server.py:181 reports it while classifying faces[0] only). */
int dfd_last_detection_count(const dfd_handle* h);
/* Crops the classifier has been run on since dfd_create (sum of its batch sizes over every entry point).  With the MTCNN
 * stage on, a crop the cascade rejects is never classified (reference deepfake_detection.py:377-380 returns before the
 * model runs): a call with n boxes of which k keep a face advances this by k.  Tests read it before / after a call. */
int dfd_classifier_crop_count(const dfd_handle* h, unsigned long long* total);
/* One named detector intermediate for parity tests: a layer name of ssd_arch.LAYERS,
 * "<source>.head", "prob", "boxes" (per prior) or "rows" (DetectionOutput: score,x1,y1,x2,y2). */
int dfd_ssd_tap(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                const char* name, float* out, size_t capacity, size_t* count);

/* ---- Haar cascade fallback -------------------------------------------------------------
 * _detect_haar, reference face_detection.py:108-123: cv2.CascadeClassifier.detectMultiScale(gray, scaleFactor,
 * minNeighbors, minSize=(min_size, min_size)) for a stump cascade with upright HAAR features packed into the blob
 * (haar.load_cascade_xml reads OpenCV's XML; weights.pack_all(..., haar=...)).  The reference uses this path
 * whenever its SSD files are missing.  Boxes are the grouped rectangles in OpenCV's order; n_candidates (may be
 * NULL) receives the number of windows that passed the cascade before grouping. */
int dfd_has_haar(const dfd_handle* h);
int dfd_detect_faces_haar(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride, float scale_factor,
                          int min_neighbors, int min_size, int32_t* xywh_out, int max_out, int* n_out, int* n_candidates);

/* ---- MTCNN align/crop -------------------------------------------------
 * The reference runs facenet-pytorch's MTCNN(select_largest=False, post_process=False) on every
 * already-cropped face before the classifier (deepfake_detection.py:24-28, 376-380): P-Net over a
 * 0.709 scale pyramid, R-Net, O-Net (thresholds .6/.7/.7, NMS .5/.7/.7/.7-Min), the box with the
 * highest probability, then PIL crop + 8-bit BILINEAR resize to 160x160.  Present when the blob
 * carries "mtcnn.*" tensors (weights.pack_mtcnn_tensors); the classify entry points
 * (dfd_classify_crops, dfd_preprocess_crops, dfd_analyze_frame, dfd_analyze_batch_device) then align
 * every crop with it and return a NaN logit where it finds no face (the reference returns None
 * there); dfd_set_option(h, "mtcnn", 0) bypasses the stage.  The box bookkeeping between the three networks (NMS,
 * regression, squaring, clipping, selection, extract_face geometry and resize tables) runs on the device, one thread
 * block per crop and stage (csrc/mtcnn_boxes.hip); environment DFD_MT_DEVICE_BOXES=0 keeps it on the library's host
 * side (identical results; also the fallback when a crop has more than 8192 P-Net candidates or 4096 windows).
 *   face_chw_out : NULL or 3*160*160 floats, RGB planes 0..255 (what MTCNN.forward returns)
 *   box_out      : NULL or 5 floats (x1, y1, x2, y2, probability) of the selected box
 *   found        : 1 / 0 */
int dfd_has_mtcnn(const dfd_handle* h);
int dfd_mtcnn_align(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                    float* face_chw_out, float* box_out, int* found);
/* One named cascade intermediate for parity tests: "pnet.prob.<level>" [H][W], "pnet.reg.<level>"
 * [H][W][4], "rnet.prob" / "onet.prob" [n] and "rnet.reg" / "onet.reg" [n][4] of the candidate windows in
 * order, "stage1" / "stage2" / "stage3" rows (x1,y1,x2,y2,score); dims[3] receives the shape. */
int dfd_mtcnn_tap(dfd_handle* h, const uint8_t* bgr, int height, int width, int stride,
                  const char* name, float* out, size_t capacity, size_t* count, int* dims);

/* ---- frame forensics ------------------------------------------------------------------
 * FrameForensicAnalyzer.analyze (full != 0) / analyze_fast (full == 0), reference
 * frame_analysis.py:58-126: cv2.resize to 256x256, then the six signals (:128-389), weighted
 * sum (:49-56,88 / :118-119), clip to [0,1].  Temporal state (previous gray frame, last 30
 * mean differences, frame counter; :35-37) is kept per `stream_id` inside the handle.
 *   scores_out[6] = frequency, noise, ela, edge, color, temporal   (NaN where the fast variant
 *                   does not compute a signal)
 *   prob_out      = fake_probability
 *   stats_out     = NULL or DFD_FORENSIC_NSTATS doubles: the quantities the thresholds act on
 *                   (low/mid/high band means, high ratio, mid ratio, mid cv, noise mean, noise cv,
 *                    ela mean, ela cv, edge density, laplacian variance, S std, V std, unique hues,
 *                    mean abs frame difference (-1 on a stream's first frame), temporal cv,
 *                    frame counter) */
#define DFD_FORENSIC_NSTATS 18
int dfd_forensics(dfd_handle* h, int stream_id, const uint8_t* bgr, int height, int width, int stride,
                  int full, double* scores_out, double* prob_out, double* stats_out);
/* FrameForensicAnalyzer.reset, reference frame_analysis.py:391-395. */
int dfd_forensics_reset(dfd_handle* h, int stream_id);
/* Mirrors the analyzer's attributes frame_count, len(temporal_diffs), prev_frame_gray is not None. */
int dfd_forensics_state(dfd_handle* h, int stream_id, int* frame_count, int* n_diffs, int* has_prev);

/* ---- one frame, end to end ---------------------------------------------------------------
 * The per-frame work of DeepfakeDetector.predict (reference deepfake_detection.py:597-626)
 * and of the /analyze handler (reference backend_server.py:147-164) with ONE upload of the
 * frame: forensics (full or fast) on `stream_id`, face detection, then crop -> CLAHE -> 224x224
 * -> classifier for the first min(n_detected, max_faces) boxes (predict uses all faces, the
 * server faces[0]; more faces than the handle's max_batch are classified in several passes).  scores_out[6]/forensic_prob_out as dfd_forensics; xywh_out receives
 * n_faces_out boxes and logits_out one logit per box.  Calibration, the +0.10 small-face
 * heuristic and the vote are host logic (scalars). */
int dfd_analyze_frame(dfd_handle* h, int stream_id, const uint8_t* bgr, int height, int width,
                      int stride, int full_forensics, float conf_thr, int max_faces, int apply_clahe,
                      double* scores_out, double* forensic_prob_out, int32_t* xywh_out,
                      int* n_faces_out, float* logits_out);

/* ---- image decode at the HTTP edge -----------------------------------------------------
 * cv2.imdecode(np.frombuffer(bytes), cv2.IMREAD_COLOR) of reference backend_server.py:139-145 for JPEG input (what
 * the extension sends): entropy decoding on the host, dequantisation + libjpeg's islow IDCT + fancy chroma
 * upsampling + YCbCr->RGB on the device - bit-identical to libjpeg's defaults.  8-bit sequential Huffman JPEGs, gray
 * or YCbCr 4:4:4 / 4:2:2 / 4:2:0, one interleaved scan, restart intervals; anything else returns
 * DFD_ERR_UNSUPPORTED (the host then decodes with its own library and calls dfd_analyze_frame).
 * dfd_decode_jpeg: bgr_out may be NULL (size query through height / width; the frame stays on the device).
 * dfd_analyze_jpeg: dfd_analyze_frame without the raw upload - the decoded frame never visits the host. */
/* The host half alone (no GPU needed; tests pin it against libjpeg through the oracle's IDCT): info[14] = width,
 * height, components, hmax, vmax, then per component (blocks_w, blocks_h, quantisation table index);
 * qtables_out = 4 x 64 uint16 in natural order (NULL: skip); coef_out = int16 quantised coefficients in natural
 * order, 64 per block, blocks row-major per component incl. MCU padding, components concatenated (NULL: count only).
 * Errors of this function are reported through dfd_last_error(NULL). */
int dfd_jpeg_coefficients(const uint8_t* jpeg, size_t len, int* info, uint16_t* qtables_out, int16_t* coef_out,
                          size_t capacity, size_t* count);
int dfd_decode_jpeg(dfd_handle* h, const uint8_t* jpeg, size_t len, uint8_t* bgr_out, size_t capacity, int* height, int* width);
/* n JPEGs of ONE size -> n packed BGR frames [n][H][W][3] (bgr_out may be NULL: the frames stay on the device).  Round 4:
 * in a batch the scans of restart-less files are entropy-decoded ON THE DEVICE (csrc/jpeg_gpu_entropy.h: a lane per
 * 512-byte chunk of the de-stuffed scan, the host decoder's speculative-chunk scheme as a fixed-point iteration) - the
 * JPEG bytes cross PCIe instead of 6.2 MB of coefficients per 1080p frame.  Restart-interval files, files of differing
 * sampling, and any frame the device decoder's own checks do not vouch for go through the host decoder; the result is
 * the same bits either way (tests pin both to libjpeg).  Options: "jpeg_device_entropy" (default 2: from 1 MiB of scan data
 * per call - a small batch is quicker on the host pool; 1 = always; 0 = never), "jpeg_chunk_bytes" (default 512), "jpeg_rounds".  dfd_jpeg_decode_counts: frames of batch calls decoded
 * on the device / by the host decoder since dfd_create. */
int dfd_decode_jpeg_batch(dfd_handle* h, int n, const uint8_t* const* jpegs, const size_t* lens, uint8_t* bgr_out, size_t capacity,
                          int* height, int* width);
int dfd_jpeg_decode_counts(const dfd_handle* h, unsigned long long* on_device, unsigned long long* on_host);
/* dfd_analyze_frames_host with JPEG files instead of raw frames: n_total files of one size and sampling (jpegs[i], lens[i]; host
 * memory, from dfd_host_alloc for full speed) are analysed `batch` at a time - the scans of chunk k + 1 cross PCIe on the
 * copy stream while chunk k is entropy-decoded on the device, turned into frames (IDCT, upsampling, colour) and run
 * through dfd_analyze_batch_device.  0.3 - 1.2 MB per 1080p frame over the link instead of 6.2 MB: the path that is not
 * bound by the raw upload.  DFD_ERR_UNSUPPORTED for files only the host decoder takes (restart intervals, mixed layouts).
 * Results as dfd_analyze_batch_device; height_out / width_out (may be NULL) receive the frame size. */
int dfd_analyze_jpegs_host(dfd_handle* h, const uint8_t* const* jpegs, const size_t* lens, int n_total, int batch,
                           const int32_t* forced_xywh, int forced_k, float conf_thr, int max_faces, int apply_clahe, int with_forensics,
                           int32_t* xywh_out, int* n_faces_out, float* logits_out, double* forensic_prob_out, int* height_out,
                           int* width_out);
int dfd_analyze_jpeg(dfd_handle* h, int stream_id, const uint8_t* jpeg, size_t len, int full_forensics, float conf_thr,
                     int max_faces, int apply_clahe, double* scores_out, double* forensic_prob_out, int32_t* xywh_out,
                     int* n_faces_out, float* logits_out, int* height_out, int* width_out);

/* ---- one request, several consecutive frames of ONE stream (POST /analyze_batch) --------------
 * The per-frame flow of dfd_analyze_frame / dfd_analyze_jpeg (reference backend_server.py:139-164, executed once per
 * request there) for n frames in stream order with every stage batched: data[i] is a JPEG of len[i] bytes (entropy
 * decoding of the files in parallel on the library's host threads, IDCT / colour on the device) or, with len[i] = 0, a
 * packed BGR frame of height x width; all frames of a call share one size.  full_forensics[i]: the caller's full / fast
 * schedule (reference deepfake_detection.py:509-512).  The stream's temporal forensic state advances n frames.
 * Outputs: scores [n][6], forensic_prob [n], xywh [n][max_faces][4], n_faces [n] (boxes returned and classified),
 * n_detected [n] or NULL (len(faces) before the max_faces cut, backend_server.py:181), logits [n][max_faces] (NaN:
 * the MTCNN stage found no face), *height / *width or NULL.  Results equal n single calls in the same order.
 * DFD_ERR_UNSUPPORTED / DFD_ERR_ARG for a JPEG the device path does not decode: nothing of the stream's state has
 * moved yet (headers of all parts are parsed before any stage runs). */
int dfd_analyze_stream_batch(dfd_handle* h, int stream_id, int n, const uint8_t* const* data, const size_t* len, int height,
                             int width, const int* full_forensics, float conf_thr, int max_faces, int apply_clahe,
                             double* scores_out, double* forensic_prob_out, int32_t* xywh_out, int* n_faces_out,
                             int* n_detected_out, float* logits_out, int* height_out, int* width_out);

/* ---- many frames, resident in HBM (throughput path; BASELINE.json configs[2]/[3]) ----------
 * frames_dev: n packed 8-bit BGR frames of height x width on the handle's device (row stride
 * width*3).  Runs the detector on all frames in one launch set; then, per frame, classifies
 * either the detected boxes (forced_xywh == NULL, at most max_faces) or the caller's forced_k
 * boxes per frame (forced_xywh: n*forced_k quadruples - used by the benchmark so that the crop
 * workload does not depend on what a random-weight detector fires on).  with_forensics != 0 also
 * computes the stateless six-signal probability of every frame.
 *   xywh_out [n][max_faces][4], n_faces_out [n], logits_out [n][max_faces], forensic_prob_out [n]. */
int dfd_analyze_batch_device(dfd_handle* h, const uint8_t* frames_dev, int n, int height, int width,
                             const int32_t* forced_xywh, int forced_k, float conf_thr, int max_faces,
                             int apply_clahe, int with_forensics, int32_t* xywh_out, int* n_faces_out,
                             float* logits_out, double* forensic_prob_out);

/* ---- many frames from host memory, upload overlapped with compute ---------------------------------------------
 * The PCIe-inclusive form of dfd_analyze_batch_device: n_total packed BGR frames in host memory (ideally from
 * dfd_host_alloc = pinned, so that the copies run asynchronously) are processed `batch` at a time; batch k + 1 is
 * uploaded on a second HIP stream while batch k is analysed.  Output arrays are sized for n_total frames. */
int dfd_host_alloc(dfd_handle* h, size_t bytes, void** ptr);
int dfd_host_free(dfd_handle* h, void* ptr);
int dfd_analyze_frames_host(dfd_handle* h, const uint8_t* frames_host, int n_total, int batch, int height, int width,
                            const int32_t* forced_xywh, int forced_k, float conf_thr, int max_faces, int apply_clahe,
                            int with_forensics, int32_t* xywh_out, int* n_faces_out, float* logits_out,
                            double* forensic_prob_out);

/* ---- frame-sharded streams (BASELINE.json configs[4]) ------------------------------
 * With frame t of a stream on rank t % G, the only state that crosses frames is the vote window (reference
 * deepfake_detection.py:111-118) and the analyzer's temporal signal (reference frame_analysis.py:349-389: the
 * previous gray frame and the last 30 mean differences).  A rank therefore computes, for each of its frames,
 * the five stateless signals and the mean absolute gray difference against the frame's predecessor (which it
 * also holds: recomputed, not communicated), all ranks exchange fixed-size records with ONE all-gather per wave,
 * and every rank replays temporal score, weighted sum and vote in frame order (host: streams.py).
 *
 * dfd_forensic_signals_device: n packed BGR frames resident in HBM; prev_index[f] = index (inside this batch) of
 * frame f's predecessor, -1 (none), or -2: frame f is itself only a predecessor - it gets a gray plane and no
 * signals (its outputs are set to -1); such frames must form the tail of the batch.  scores5_out [n][5] =
 * frequency, noise, ela, edge, color (all five computed; the caller drops noise/ela/color on "fast" frames);
 * mean_diff_out [n] = mean |gray - gray_prev| or -1. */
int dfd_forensic_signals_device(dfd_handle* h, const uint8_t* frames_dev, int n, int height, int width,
                                const int32_t* prev_index, double* scores5_out, double* mean_diff_out);

/* Vote exchange over RCCL (xGMI inside a node).  Rank 0 obtains an id (ncclGetUniqueId) and hands its
 * DFD_COMM_ID_BYTES bytes to the other ranks by any host channel (file, socket, torch.distributed store); every
 * rank then calls dfd_comm_init on its handle.  dfd_vote_allgather copies `bytes_per_rank` bytes of records
 * to the device, runs ONE ncclAllGather on the handle's stream and returns all ranks' blocks, rank-major, in
 * host memory (world * bytes_per_rank bytes): the collective named in SURVEY 8(b)/(e).  librccl is opened on
 * first use (DFD_RCCL_LIB overrides the name); without it these three return DFD_ERR_STATE. */
#define DFD_COMM_ID_BYTES 128
int dfd_comm_unique_id(void* id_out);
int dfd_comm_init(dfd_handle* h, const void* id, int rank, int world);
int dfd_comm_destroy(dfd_handle* h);
int dfd_comm_info(const dfd_handle* h, int* rank, int* world);      /* world = 0: no communicator */
int dfd_vote_allgather(dfd_handle* h, const void* local_records, size_t bytes_per_rank, void* all_records_out);
/* `waves` consecutive waves in one call: one upload of [waves][bytes_per_rank], one ncclAllGather PER WAVE (slot w of
 * all_records_out = [world][bytes_per_rank] of wave w), one download, one stream wait. */
int dfd_vote_allgather_waves(dfd_handle* h, const void* local_records, int waves, size_t bytes_per_rank, void* all_records_out);

#ifdef __cplusplus
}
#endif
#endif /* DFD_HIP_H */
