// C ABI entry points of the per-face pre-processing path (include/dfd_hip.h).
#include "b0_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstring>

#include "dfd_common.h"

using namespace dfd;

namespace dfd {

__global__ __launch_bounds__(256) void mailbox_copy_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

hipError_t stream_sync(dfd_handle* h) {
    const hipError_t e = hipStreamSynchronize(h->stream);
    // regions handed out before the previous sync have been copied from (h2d) and read by the host (d2h, read right
    // after the sync that completed them): only the last two generations stay live
    h->mailbox_live_prev = h->mailbox_live;
    h->mailbox_live = 0;
    for (char* p : h->mailbox_old_prev) hipHostFree(p);
    h->mailbox_old_prev.swap(h->mailbox_old);
    h->mailbox_old.clear();
    return e;
}

static char* mailbox_alloc(dfd_handle* h, size_t bytes) {
    constexpr size_t kCap = 8u << 20;
    bytes = (bytes + 255) & ~(size_t)255;
    // Bump allocation in a ring.  Live = what was handed out since the sync before last (mailbox_live +
    // mailbox_live_prev bytes ending at the head, continuing below mailbox_lap_end after a wrap).  A wrap-around to
    // offset 0 is taken only when the new region ends before the live span starts; otherwise - and for a request above
    // an eighth of the block - a new, larger block replaces this one (the old block stays allocated until two syncs
    // later: pointers into it remain valid).
    const size_t live = h->mailbox_live + h->mailbox_live_prev;
    bool grow = !h->mailbox || bytes > h->mailbox_cap / 8;
    if (!grow) {
        if (live > h->mailbox_head) {
            // the live span reaches back into the previous lap: [lap_end - (live - head), lap_end)
            if (h->mailbox_head + bytes > h->mailbox_lap_end - (live - h->mailbox_head)) grow = true;
        } else if (h->mailbox_head + bytes > h->mailbox_cap) {
            if (bytes <= h->mailbox_head - live) { h->mailbox_lap_end = h->mailbox_head; h->mailbox_head = 0; }
            else grow = true;
        }
    }
    if (grow && h->mailbox && bytes <= h->mailbox_cap / 8 && h->mailbox_cap >= ((size_t)256 << 20)) {
        // a caller that never waits on the stream between requests: wait for it here rather than grow without bound
        // (two generations retire: nothing handed out so far is still in flight; the caller's own unread d2h regions
        // are the ones just completed and sit behind the head)
        stream_sync(h);
        stream_sync(h);
        h->mailbox_lap_end = h->mailbox_head;
        if (h->mailbox_head + bytes > h->mailbox_cap) h->mailbox_head = 0;
        grow = false;
    }
    if (grow) {
        const size_t cap = std::max(std::max(kCap, bytes * 16), h->mailbox ? h->mailbox_cap * 2 : (size_t)0);
        char* p = nullptr;
        if (hipHostMalloc((void**)&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
        if (getenv("DFD_MAILBOX_VERBOSE")) fprintf(stderr, "[dfd] mailbox: new block of %zu bytes (request %zu, head %zu, live %zu)\n", cap, bytes, h->mailbox_head, live);
        if (h->mailbox) h->mailbox_old.push_back(h->mailbox);
        h->mailbox = p;
        h->mailbox_cap = cap;
        h->mailbox_head = 0;
        h->mailbox_lap_end = 0;
        h->mailbox_live = h->mailbox_live_prev = 0;                // the live span stayed behind in the old block
    }
    char* p = h->mailbox + h->mailbox_head;
    h->mailbox_head += bytes;
    h->mailbox_live += bytes;
    return p;
}

static void mailbox_launch(dfd_handle* h, void* dst, const void* src, size_t bytes) {
    const size_t words = (bytes + 3) / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256);
    hipLaunchKernelGGL(mailbox_copy_kernel, dim3(blocks), dim3(256), 0, h->stream, (unsigned*)dst, (const unsigned*)src, words);
}

// device -> pinned host memory with the copy kernel on any stream (results of work that runs beside the main stream)
void copy_kernel_async(void* dst, const void* src, size_t bytes, hipStream_t s) {
    const size_t words = (bytes + 3) / 4;
    if (!words) return;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256);
    hipLaunchKernelGGL(mailbox_copy_kernel, dim3(blocks), dim3(256), 0, s, (unsigned*)dst, (const unsigned*)src, words);
}

int mailbox_h2d(dfd_handle* h, void* dst_dev, const void* src, size_t bytes) {
    if (!bytes) return DFD_OK;
    char* p = mailbox_alloc(h, bytes);
    if (!p) {                                               // larger than the mailbox: the DMA path
        DFD_HIP_TRY(h, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, h->stream));
        DFD_HIP_TRY(h, stream_sync(h));
        return DFD_OK;
    }
    memcpy(p, src, bytes);
    mailbox_launch(h, dst_dev, p, bytes);
    DFD_HIP_TRY(h, hipGetLastError());
    return DFD_OK;
}

const void* mailbox_d2h(dfd_handle* h, const void* src_dev, size_t bytes) {
    char* p = mailbox_alloc(h, bytes ? bytes : 4);
    if (!p) return nullptr;
    if (bytes) mailbox_launch(h, p, src_dev, bytes);
    return p;
}

int ensure(dfd_handle* h, DevBuf* b, size_t bytes) {
    if (bytes <= b->cap) return DFD_OK;
    DFD_HIP_TRY(h, stream_sync(h));
    if (b->p) {
        DFD_HIP_TRY(h, hipFree(b->p));
        for (auto& o : h->owned)
            if (o == b->p) o = nullptr;
        b->p = nullptr;
        b->cap = 0;
    }
    const size_t want = (bytes + (1 << 20) - 1) & ~((size_t)(1 << 20) - 1);
    DFD_HIP_TRY(h, hipMalloc(&b->p, want));
    h->owned.push_back(b->p);
    b->cap = want;
    return DFD_OK;
}

// "lut.*" tensors hold integers as exactly-representable float32; convert to int32 tables.
int color_tables_init(dfd_handle* h) {
    struct Want { const char* name; size_t count; const int** slot; };
    ColorTables& T = h->color;
    const Want wants[] = {{"lut.gamma", 256, &T.gamma},       {"lut.cbrt", 3072, &T.cbrt},
                          {"lut.L_fy", 256, &T.L_fy},         {"lut.L_y", 256, &T.L_y},
                          {"lut.a_div", 256, &T.a_div},       {"lut.b_div", 256, &T.b_div},
                          {"lut.ab_xz", 36864, &T.ab_xz},     {"lut.inv_gamma", 4096, &T.inv_gamma},
                          {"lut.hsv_sdiv", 256, &T.hsv_sdiv}, {"lut.hsv_hdiv", 256, &T.hsv_hdiv}};
    if (h->tensors.find("lut.gamma") == h->tensors.end()) return DFD_OK;   // blob without tables
    std::vector<float> tmp;
    for (const Want& w : wants) {
        auto it = h->tensors.find(w.name);
        if (it == h->tensors.end() || it->second.count != w.count)
            return fail(h, DFD_ERR_BLOB, "weights blob: table '%s' missing or wrong size", w.name);
        tmp.resize(w.count);
        DFD_HIP_TRY(h, hipMemcpy(tmp.data(), it->second.dev, w.count * 4, hipMemcpyDeviceToHost));
        std::vector<int> iv(w.count);
        for (size_t i = 0; i < w.count; ++i) iv[i] = (int)tmp[i];
        void* d = nullptr;
        DFD_HIP_TRY(h, hipMalloc(&d, w.count * 4));
        h->owned.push_back(d);
        DFD_HIP_TRY(h, hipMemcpy(d, iv.data(), w.count * 4, hipMemcpyHostToDevice));
        *w.slot = static_cast<const int*>(d);
    }
    for (const char* name : {"lut.fwd_coef", "lut.inv_coef"}) {
        auto it = h->tensors.find(name);
        if (it == h->tensors.end() || it->second.count != 9)
            return fail(h, DFD_ERR_BLOB, "weights blob: table '%s' missing or wrong size", name);
        float c[9];
        DFD_HIP_TRY(h, hipMemcpy(c, it->second.dev, 36, hipMemcpyDeviceToHost));
        for (int i = 0; i < 9; ++i) {
            if (name[4] == 'f') T.fwd[i] = (int)c[i];
            else T.inv[i] = (long long)c[i];
        }
    }
    h->has_color = true;
    return DFD_OK;
}

}  // namespace dfd

namespace {

int upload_frame(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride) {
    if (!bgr || hh <= 0 || ww <= 0 || stride < ww * 3) return fail(h, DFD_ERR_ARG, "frame: bad pointer or geometry");
    int rc = ensure(h, &h->frame_buf, (size_t)hh * stride);
    if (rc) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    return DFD_OK;
}

// validates the boxes, lays the packed crops out in the scratch buffers, uploads descriptors
int stage_crops(dfd_handle* h, int hh, int ww, const int32_t* xywh, int n, size_t* total, int* max_pixels,
                const size_t* frame_offs = nullptr) {
    if (!xywh || n <= 0) return fail(h, DFD_ERR_ARG, "crops: null boxes or n <= 0");
    if (n > h->max_batch) return fail(h, DFD_ERR_CAPACITY, "crops: %d boxes exceed handle capacity %d", n, h->max_batch);
    std::vector<CropDesc> d(n);
    size_t off = 0;
    int mp = 0;
    for (int i = 0; i < n; ++i) {
        const int x = xywh[4 * i], y = xywh[4 * i + 1], w = xywh[4 * i + 2], hgt = xywh[4 * i + 3];
        if (w <= 0 || hgt <= 0 || x < 0 || y < 0 || x + w > ww || y + hgt > hh)
            return fail(h, DFD_ERR_ARG, "crops: box %d (%d,%d,%d,%d) outside the %dx%d frame", i, x, y, w, hgt, ww, hh);
        d[i] = CropDesc{x, y, w, hgt, off, frame_offs ? frame_offs[i] : 0};
        off += ((size_t)w * hgt * 3 + 255) & ~(size_t)255;
        if (w * hgt > mp) mp = w * hgt;
    }
    int rc;
    if ((rc = ensure(h, &h->desc_buf, n * sizeof(CropDesc)))) return rc;
    if ((rc = mailbox_h2d(h, h->desc_buf.p, d.data(), n * sizeof(CropDesc)))) return rc;      // `d` is copied before the call returns
    *total = off;
    *max_pixels = mp;
    return DFD_OK;
}

// frame already on the device -> normalised NCHW crops in h->in_nchw
// compact: in_nchw receives only the crops the MTCNN stage kept (h->n_compact rows, crop order); without the stage, or
// with compact = false, all n rows (a rejected crop's row is then the zero-filled face, as facenet-pytorch never returns)
int preprocess_on_device(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride,
                         const int32_t* xywh, int n, int apply_clahe, const size_t* frame_offs = nullptr, bool compact = false) {
    size_t total = 0;
    int mp = 0, rc;
    if ((rc = stage_crops(h, hh, ww, xywh, n, &total, &mp, frame_offs))) return rc;
    const CropDesc* dd = static_cast<const CropDesc*>(h->desc_buf.p);
    if (apply_clahe) {
        if (!h->has_color) return fail(h, DFD_ERR_STATE, "CLAHE needs the colour tables (blob packed without luts)");
        if ((rc = ensure(h, &h->lab_buf, total))) return rc;
        if ((rc = ensure(h, &h->crop_buf, total))) return rc;
        if ((rc = ensure(h, &h->lut_buf, (size_t)n * 64 * 256))) return rc;
        launch_clahe(frame_dev, stride, dd, n, (uint8_t*)h->lab_buf.p, (uint8_t*)h->lut_buf.p,
                     (uint8_t*)h->crop_buf.p, h->color, mp, h->stream);
    }
    h->crop_valid.assign(n, 1);
    h->n_compact = n;
    if (h->use_mtcnn && h->mtcnn) {
        // reference deepfake_detection.py:376-380: MTCNN.forward on the (CLAHE'd) crop picks the face window and
        // resamples it to 160x160; that image, not the detector crop, feeds the 224x224 bilinear + normalise
        // (:382-389).  All crops of the call go through the cascade together; a crop without a face yields no
        // prediction (NaN logit, `None` upstream).
        if ((rc = ensure(h, &h->face_batch, (size_t)n * 160 * 160 * 3))) return rc;
        std::vector<MtImage> imgs(n);
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            const int x = xywh[4 * i], y = xywh[4 * i + 1], w = xywh[4 * i + 2], hgt = xywh[4 * i + 3];
            const uint8_t* img = apply_clahe ? (const uint8_t*)h->crop_buf.p + off
                                             : frame_dev + (frame_offs ? frame_offs[i] : 0) + (size_t)y * stride + (size_t)x * 3;
            imgs[i] = MtImage{img, hgt, w, apply_clahe ? (size_t)w * 3 : (size_t)stride};
            off += ((size_t)w * hgt * 3 + 255) & ~(size_t)255;        // as stage_crops lays the packed crops out
        }
        // the cascade returns after its last stream wait with the per-crop flags (one 32-byte row per crop): what the
        // 224 x 224 resize and the classifier are sized by.  (Round 3 queued both for all n crops behind the extract
        // launches and read the flags afterwards: on the bench funnel 72 % of the classifier pass was thrown away.)
        if ((rc = mtcnn_align_batch_device(h, imgs.data(), n, (uint8_t*)h->face_batch.p, nullptr, h->crop_valid.data(), nullptr,
                                           nullptr, nullptr)))
            return rc;
        std::vector<CropDesc> fd;
        fd.reserve(n);
        for (int i = 0; i < n; ++i)
            if (!compact || h->crop_valid[i]) fd.push_back(CropDesc{0, 0, 160, 160, 0, (size_t)i * 160 * 160 * 3});
        h->n_compact = (int)fd.size();
        if (fd.empty()) return DFD_OK;
        if ((rc = mailbox_h2d(h, h->desc_buf.p, fd.data(), fd.size() * sizeof(CropDesc)))) return rc;
        launch_crop_norm((const uint8_t*)h->face_batch.p, 160 * 3, nullptr, dd, (int)fd.size(), h->in_nchw, false, h->stream);
        DFD_HIP_TRY(h, hipGetLastError());
        return DFD_OK;
    }
    launch_crop_norm(frame_dev, stride, (const uint8_t*)h->crop_buf.p, dd, n, h->in_nchw, apply_clahe != 0, h->stream);
    DFD_HIP_TRY(h, hipGetLastError());
    return DFD_OK;
}

}  // namespace

namespace dfd {
int preprocess_run(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, const int32_t* xywh, int n,
                   int apply_clahe, const size_t* frame_offs) {
    return preprocess_on_device(h, frame_dev, hh, ww, stride, xywh, n, apply_clahe, frame_offs);
}

int classify_boxes(dfd_handle* h, const uint8_t* frame_dev, int hh, int ww, int stride, const int32_t* xywh, int m,
                   int apply_clahe, const size_t* frame_offs, float* logits_out) {
    int rc;
    if ((rc = preprocess_on_device(h, frame_dev, hh, ww, stride, xywh, m, apply_clahe, frame_offs, true))) return rc;
    const int k = h->n_compact;
    const float* lg = nullptr;
    if (k > 0) {
        if ((rc = b0_forward(h, h->in_nchw, k, h->logits, nullptr, nullptr))) return rc;
        lg = (const float*)mailbox_d2h(h, h->logits, (size_t)k * 4);
        if (!lg) return fail(h, DFD_ERR_HIP, "classify: mailbox allocation failed");
        DFD_HIP_TRY(h, hipGetLastError());
    }
    DFD_HIP_TRY(h, stream_sync(h));
    for (int i = 0, j = 0; i < m; ++i) logits_out[i] = h->crop_valid[i] ? lg[j++] : NAN;      // NaN: MTCNN found no face in this crop
    return DFD_OK;
}
}  // namespace dfd

extern "C" {

// DeepfakeDetector's per-frame work with ONE upload of the frame: forensics -> detect -> crop/CLAHE ->
// classify (reference backend_server.py:147-164 / deepfake_detection.py:597-615).
int dfd_analyze_frame(dfd_handle* h, int stream_id, const uint8_t* bgr, int hh, int ww, int stride, int full_forensics,
                      float conf_thr, int max_faces, int apply_clahe, double* scores_out, double* forensic_prob_out,
                      int32_t* xywh_out, int* n_faces_out, float* logits_out) {
    if (!h) return DFD_ERR_ARG;
    if (!bgr || !scores_out || !forensic_prob_out || !xywh_out || !n_faces_out || !logits_out || max_faces <= 0 ||
        hh <= 0 || ww <= 0 || stride < ww * 3)
        return fail(h, DFD_ERR_ARG, "analyze_frame: bad pointer or geometry");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure(h, &h->frame_buf, (size_t)hh * stride);
    if (rc) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    return analyze_frame_resident(h, stream_id, hh, ww, stride, full_forensics, conf_thr, max_faces, apply_clahe, scores_out,
                                  forensic_prob_out, xywh_out, n_faces_out, logits_out);
}

// the same from JPEG bytes: the frame is decoded on the device (jpeg_decode.hip) instead of uploaded raw
int dfd_analyze_jpeg(dfd_handle* h, int stream_id, const uint8_t* jpeg, size_t len, int full_forensics, float conf_thr,
                     int max_faces, int apply_clahe, double* scores_out, double* forensic_prob_out, int32_t* xywh_out,
                     int* n_faces_out, float* logits_out, int* height_out, int* width_out) {
    if (!h) return DFD_ERR_ARG;
    if (!jpeg || !scores_out || !forensic_prob_out || !xywh_out || !n_faces_out || !logits_out || max_faces <= 0)
        return fail(h, DFD_ERR_ARG, "analyze_jpeg: bad pointer");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int hh = 0, ww = 0;
    int rc = jpeg_decode_to_frame(h, jpeg, len, &hh, &ww);
    if (rc) return rc;
    if (height_out) *height_out = hh;
    if (width_out) *width_out = ww;
    return analyze_frame_resident(h, stream_id, hh, ww, ww * 3, full_forensics, conf_thr, max_faces, apply_clahe, scores_out,
                                  forensic_prob_out, xywh_out, n_faces_out, logits_out);
}

}  // extern "C"

namespace dfd {
// the frame is in h->frame_buf: forensics -> detect -> crop/CLAHE -> classify
int analyze_frame_resident(dfd_handle* h, int stream_id, int hh, int ww, int stride, int full_forensics, float conf_thr,
                           int max_faces, int apply_clahe, double* scores_out, double* forensic_prob_out, int32_t* xywh_out,
                           int* n_faces_out, float* logits_out) {
    int rc;
    const uint8_t* fd = (const uint8_t*)h->frame_buf.p;
    if ((rc = forensics_run(h, stream_id, fd, hh, ww, stride, full_forensics, scores_out, forensic_prob_out, nullptr))) return rc;
    *n_faces_out = 0;
    h->last_detections = 0;
    if ((!h->ssd && !h->haar) || hh < 30 || ww < 30) return DFD_OK;  // no detector at all / tiny frame: no faces
    int n = 0;
    // reference face_detection.py:58-66: the DNN when its files were loaded, the Haar cascade otherwise - and after
    // a DNN failure (detectMultiScale(gray, 1.1, 5, minSize 30x30), :115-121)
    rc = h->ssd ? detect_run(h, fd, hh, ww, stride, conf_thr, xywh_out, nullptr, max_faces, &n) : DFD_ERR_STATE;
    if (rc) {
        if (!h->haar) return rc;
        if ((rc = haar_run(h, fd, hh, ww, stride, 1.1f, 5, 30, xywh_out, max_faces, &n, nullptr, &h->last_detections))) return rc;
    }
    *n_faces_out = n;
    // every returned face is classified, in chunks of the handle's batch capacity (predict votes on all detections,
    // reference deepfake_detection.py:611-626; dfd_last_detection_count gives len(faces) when max_faces cut the list)
    for (int start = 0; start < n; start += h->max_batch) {
        const int m = n - start < h->max_batch ? n - start : h->max_batch;
        if ((rc = classify_boxes(h, fd, hh, ww, stride, xywh_out + (size_t)start * 4, m, apply_clahe, nullptr, logits_out + start))) return rc;
    }
    return DFD_OK;
}
}  // namespace dfd

extern "C" {

// One test-time-augmentation copy of a face crop (reference deepfake_detection.py:419-433).
int dfd_tta_augment(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, int flip, double brightness, double angle_deg,
                    uint8_t* out) {
    if (!h) return DFD_ERR_ARG;
    if (!out) return fail(h, DFD_ERR_ARG, "tta_augment: null output");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = upload_frame(h, bgr, hh, ww, stride);
    if (rc) return rc;
    if ((rc = ensure(h, &h->u8_out, (size_t)hh * ww * 3))) return rc;
    // cv2.getRotationMatrix2D((w/2, h/2), angle, 1.0), then warpAffine's inversion (imgwarp.cpp), all in double
    const double cx = ww / 2.0, cy = hh / 2.0;
    const double a = std::cos(angle_deg * M_PI / 180.0), b = std::sin(angle_deg * M_PI / 180.0);
    double M[6] = {a, b, (1.0 - a) * cx - b * cy, -b, a, b * cx + (1.0 - a) * cy};
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    launch_tta_augment((const uint8_t*)h->frame_buf.p, hh, ww, stride, flip ? 1 : 0, (float)brightness, M, (uint8_t*)h->u8_out.p,
                       h->stream);
    DFD_HIP_TRY(h, hipMemcpyAsync(out, h->u8_out.p, (size_t)hh * ww * 3, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

int dfd_resize_bgr(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, int dh, int dw, uint8_t* out) {
    if (!h) return DFD_ERR_ARG;
    if (!out || dh <= 0 || dw <= 0) return fail(h, DFD_ERR_ARG, "resize: bad output");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = upload_frame(h, bgr, hh, ww, stride);
    if (rc) return rc;
    if ((rc = ensure(h, &h->u8_out, (size_t)dh * dw * 3))) return rc;
    launch_resize_bgr((const uint8_t*)h->frame_buf.p, 1, hh, ww, stride, 0, (uint8_t*)h->u8_out.p, dh, dw, h->stream);
    DFD_HIP_TRY(h, hipMemcpyAsync(out, h->u8_out.p, (size_t)dh * dw * 3, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

int dfd_preprocess_face_quality(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, uint8_t* out) {
    if (!h) return DFD_ERR_ARG;
    if (!out) return fail(h, DFD_ERR_ARG, "preprocess_face_quality: null output");
    if (!h->has_color) return fail(h, DFD_ERR_STATE, "CLAHE needs the colour tables (blob packed without luts)");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = upload_frame(h, bgr, hh, ww, stride);
    if (rc) return rc;
    const int32_t box[4] = {0, 0, ww, hh};
    size_t total = 0;
    int mp = 0;
    const int saved = h->max_batch;
    if ((rc = stage_crops(h, hh, ww, box, 1, &total, &mp))) return rc;
    (void)saved;
    if ((rc = ensure(h, &h->lab_buf, total))) return rc;
    if ((rc = ensure(h, &h->crop_buf, total))) return rc;
    if ((rc = ensure(h, &h->lut_buf, 64 * 256))) return rc;
    launch_clahe((const uint8_t*)h->frame_buf.p, stride, (const CropDesc*)h->desc_buf.p, 1, (uint8_t*)h->lab_buf.p,
                 (uint8_t*)h->lut_buf.p, (uint8_t*)h->crop_buf.p, h->color, mp, h->stream);
    DFD_HIP_TRY(h, hipMemcpyAsync(out, h->crop_buf.p, (size_t)hh * ww * 3, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

int dfd_preprocess_crops(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, const int32_t* xywh,
                         int n, int apply_clahe, float* nchw_out) {
    if (!h) return DFD_ERR_ARG;
    if (!nchw_out) return fail(h, DFD_ERR_ARG, "preprocess_crops: null output");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = upload_frame(h, bgr, hh, ww, stride);
    if (rc) return rc;
    if ((rc = preprocess_on_device(h, (const uint8_t*)h->frame_buf.p, hh, ww, stride, xywh, n, apply_clahe))) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(nchw_out, h->in_nchw, (size_t)n * 3 * 224 * 224 * 4, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

int dfd_classify_crops(dfd_handle* h, const uint8_t* bgr, int hh, int ww, int stride, const int32_t* xywh, int n,
                       int apply_clahe, float* logits_out) {
    if (!h) return DFD_ERR_ARG;
    if (!logits_out) return fail(h, DFD_ERR_ARG, "classify_crops: null output");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = upload_frame(h, bgr, hh, ww, stride);
    if (rc) return rc;
    if (n <= 0 || n > h->max_batch) return fail(h, n <= 0 ? DFD_ERR_ARG : DFD_ERR_CAPACITY, "classify_crops: %d boxes outside 1..%d", n, h->max_batch);
    if ((rc = classify_boxes(h, (const uint8_t*)h->frame_buf.p, hh, ww, stride, xywh, n, apply_clahe, nullptr, logits_out))) return rc;
    return DFD_OK;
}

}  // extern "C"
