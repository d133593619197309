// Batched, device-resident end-to-end path (BASELINE.json configs[2]/[3]): n frames already in
// HBM -> detector (one launch set for all frames) -> crops -> CLAHE -> 224x224 -> classifier,
// optionally with the six forensic signals.  The per-frame semantics are those of
// dfd_analyze_frame; this entry point exists for throughput (no per-frame host round trips
// except the small DetectionOutput read-back that sizes the crop batch).
#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>
#include "b0_kernels.h"
#include "dfd_common.h"

using namespace dfd;

extern "C" {

int dfd_analyze_batch_device(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, const int32_t* forced_xywh,
                             int forced_k, float conf_thr, int max_faces, int apply_clahe, int with_forensics,
                             int32_t* xywh_out, int* n_faces_out, float* logits_out, double* forensic_prob_out) {
    if (!h) return DFD_ERR_ARG;
    if (!frames_dev || n <= 0 || hh <= 0 || ww <= 0 || max_faces <= 0 || !xywh_out || !n_faces_out || !logits_out)
        return fail(h, DFD_ERR_ARG, "analyze_batch: bad pointer or geometry");
    if (forced_xywh && (forced_k <= 0 || forced_k > max_faces))
        return fail(h, DFD_ERR_ARG, "analyze_batch: forced_k must be in 1..max_faces");
    if (with_forensics && !forensic_prob_out) return fail(h, DFD_ERR_ARG, "analyze_batch: forensic_prob_out is null");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    const int stride = ww * 3;
    const size_t frame_bytes = (size_t)hh * stride;
    int rc;
    // the six signals depend on the frames only: their launch set runs on the handle's second stream beside the detector
    // and the classifier (it fills the detector's small tail layers and the waits for detection / window counts) and is
    // collected at the end of the call; option overlap_forensics = 0: in front of the detector on the main stream
    const bool beside = with_forensics && h->overlap_forensics;
    if (with_forensics) {
        rc = beside ? forensics_batch_begin(h, frames_dev, n, hh, ww, stride, frame_bytes)
                    : forensics_batch_run(h, frames_dev, n, hh, ww, stride, frame_bytes, forensic_prob_out, nullptr);
        if (rc) return rc;
    }
    struct Collect {                                                 // every exit waits for the second stream
        dfd_handle* h; bool on; int n; double* out;
        int finish() { if (!on) return DFD_OK; on = false; return forensics_batch_end(h, n, out, nullptr); }
        ~Collect() { if (on) hipEventSynchronize(h->aux_done); }
    } collect{h, beside, n, forensic_prob_out};
    // detector on every frame (its boxes are used unless the caller forces boxes)
    std::vector<int32_t> det((size_t)n * max_faces * 4);
    std::vector<int> ndet(n);
    if ((rc = detect_batch_run(h, frames_dev, n, hh, ww, stride, frame_bytes, conf_thr, max_faces, det.data(), ndet.data())))
        return rc;
    // crop list across frames
    std::vector<int32_t> boxes;
    std::vector<size_t> offs;
    for (int f = 0; f < n; ++f) {
        const int k = forced_xywh ? forced_k : ndet[f];
        const int32_t* src = forced_xywh ? forced_xywh + (size_t)f * forced_k * 4 : det.data() + (size_t)f * max_faces * 4;
        n_faces_out[f] = k;
        for (int i = 0; i < k; ++i) {
            for (int c = 0; c < 4; ++c) {
                boxes.push_back(src[4 * i + c]);
                xywh_out[((size_t)f * max_faces + i) * 4 + c] = src[4 * i + c];
            }
            offs.push_back((size_t)f * frame_bytes);
        }
    }
    // classify in chunks of the handle's batch capacity
    const int total = (int)offs.size();
    std::vector<float> logits(total);
    for (int start = 0; start < total; start += h->max_batch) {
        const int m = std::min(h->max_batch, total - start);
        if ((rc = classify_boxes(h, frames_dev, hh, ww, stride, boxes.data() + (size_t)start * 4, m, apply_clahe, offs.data() + start,
                                 logits.data() + start)))
            return rc;
    }
    int k = 0;
    for (int f = 0; f < n; ++f)
        for (int i = 0; i < n_faces_out[f]; ++i) logits_out[(size_t)f * max_faces + i] = logits[k++];
    return collect.finish();
}


// ---- POST /analyze_batch: n consecutive frames of ONE stream in one call -----------------------------------------
// The per-frame flow of dfd_analyze_frame / dfd_analyze_jpeg (reference backend_server.py:147-164: forensics with the
// stream's temporal state and the caller's full / fast schedule, detector - SSD or the Haar fallback - and the first
// max_faces faces of every frame classified), with every stage batched over the request's frames: the JPEG parts are
// entropy-decoded in parallel on the host pool and turned into frames on the device (raw BGR parts are uploaded),
// then ONE forensic launch set, ONE detector pass, ONE classifier batch.  Results equal n single calls in order.
// data[i] / len[i]: the bytes of a JPEG (len[i] > 0) or a packed BGR frame of hh x ww (len[i] = 0).
int dfd_analyze_stream_batch(dfd_handle* h, int stream_id, int n, const uint8_t* const* data, const size_t* len, int hh, int ww,
                             const int* full_forensics, float conf_thr, int max_faces, int apply_clahe, double* scores_out,
                             double* forensic_prob_out, int32_t* xywh_out, int* n_faces_out, int* n_detected_out,
                             float* logits_out, int* height_out, int* width_out) {
    if (!h) return DFD_ERR_ARG;
    if (n <= 0 || !data || !len || !full_forensics || max_faces <= 0 || !scores_out || !forensic_prob_out || !xywh_out ||
        !n_faces_out || !logits_out)
        return fail(h, DFD_ERR_ARG, "analyze_stream_batch: bad pointer or count");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc;
    // frame size: from the JPEG headers when there is a JPEG part, else the caller's
    std::vector<const uint8_t*> jp;
    std::vector<size_t> jl;
    std::vector<int> jidx;
    for (int i = 0; i < n; ++i) {
        if (!data[i]) return fail(h, DFD_ERR_ARG, "analyze_stream_batch: frame %d is null", i);
        if (len[i]) { jp.push_back(data[i]); jl.push_back(len[i]); jidx.push_back(i); }
    }
    if (jp.size() != (size_t)n && (hh <= 0 || ww <= 0)) return fail(h, DFD_ERR_ARG, "analyze_stream_batch: raw frames need hh and ww");
    if (!jp.empty() && (rc = jpeg_decode_batch_to(h, jp.data(), jl.data(), (int)jp.size(), nullptr, &hh, &ww))) return rc;
    if (height_out) *height_out = hh;
    if (width_out) *width_out = ww;
    if ((size_t)n * (size_t)hh * (size_t)ww > kMaxBatchPixels)         // before any allocation or upload (raw parts too)
        return fail(h, DFD_ERR_UNSUPPORTED, "analyze_stream_batch: %d frames of %d x %d exceed the %zu-pixel budget of one request", n,
                    ww, hh, kMaxBatchPixels);
    const int stride = ww * 3;
    const size_t frame_bytes = (size_t)hh * stride;
    if ((rc = ensure(h, &h->stage[0], (size_t)n * frame_bytes))) return rc;
    uint8_t* frames = static_cast<uint8_t*>(h->stage[0].p);
    for (int i = 0; i < n; ++i)
        if (!len[i]) DFD_HIP_TRY(h, hipMemcpyAsync(frames + (size_t)i * frame_bytes, data[i], frame_bytes, hipMemcpyHostToDevice, h->stream));
    if (!jp.empty()) {
        if (jp.size() == (size_t)n) {
            if ((rc = jpeg_decode_batch_to(h, jp.data(), jl.data(), n, frames, &hh, &ww))) return rc;
        } else {                                                     // mixed request: the JPEG parts one by one into their slots
            for (size_t k = 0; k < jp.size(); ++k)
                if ((rc = jpeg_decode_batch_to(h, &jp[k], &jl[k], 1, frames + (size_t)jidx[k] * frame_bytes, &hh, &ww))) return rc;
        }
    }
    if ((rc = forensics_stream_batch_run(h, stream_id, frames, n, hh, ww, stride, frame_bytes, full_forensics, scores_out,
                                         forensic_prob_out)))
        return rc;
    for (int f = 0; f < n; ++f) n_faces_out[f] = 0;
    if (n_detected_out) for (int f = 0; f < n; ++f) n_detected_out[f] = 0;
    h->last_detections = 0;
    if ((!h->ssd && !h->haar) || hh < 30 || ww < 30) return DFD_OK;
    std::vector<int> total(n, 0);
    rc = h->ssd ? detect_batch_run(h, frames, n, hh, ww, stride, frame_bytes, conf_thr, max_faces, xywh_out, n_faces_out, total.data())
                : DFD_ERR_STATE;
    if (rc) {                                                        // reference face_detection.py:58-66
        if (!h->haar) return rc;
        for (int f = 0; f < n; ++f)
            if ((rc = haar_run(h, frames + (size_t)f * frame_bytes, hh, ww, stride, 1.1f, 5, 30, xywh_out + (size_t)f * max_faces * 4,
                               max_faces, &n_faces_out[f], nullptr, &total[f])))
                return rc;
    }
    if (n_detected_out) for (int f = 0; f < n; ++f) n_detected_out[f] = total[f];
    h->last_detections = total[n - 1];
    std::vector<int32_t> boxes;
    std::vector<size_t> offs;
    for (int f = 0; f < n; ++f)
        for (int i = 0; i < n_faces_out[f]; ++i) {
            for (int c = 0; c < 4; ++c) boxes.push_back(xywh_out[((size_t)f * max_faces + i) * 4 + c]);
            offs.push_back((size_t)f * frame_bytes);
        }
    const int ncrops = (int)offs.size();
    std::vector<float> logits(ncrops);
    for (int start = 0; start < ncrops; start += h->max_batch) {
        const int m = std::min(h->max_batch, ncrops - start);
        if ((rc = classify_boxes(h, frames, hh, ww, stride, boxes.data() + (size_t)start * 4, m, apply_clahe, offs.data() + start,
                                 logits.data() + start)))
            return rc;
    }
    int k = 0;
    for (int f = 0; f < n; ++f)
        for (int i = 0; i < n_faces_out[f]; ++i) logits_out[(size_t)f * max_faces + i] = logits[k++];
    return DFD_OK;
}


// ---- host frames, PCIe-inclusive (BASELINE.json metric "frames/sec/GPU" with the upload counted) -------------
// n_total frames in (pinned) host memory are analysed `batch` at a time through two device staging buffers: the
// upload of batch k + 1 is issued on the handle's copy stream before batch k is computed on the compute stream, so
// the 6.2 MB per 1080p frame cross PCIe while the previous batch runs (reference flow: one cv2.imdecode'd frame per
// request, backend_server.py:139-164 - here many frames per call).  Results as dfd_analyze_batch_device.
int dfd_host_alloc(dfd_handle* h, size_t bytes, void** ptr) {
    if (!h || !ptr) return DFD_ERR_ARG;
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    DFD_HIP_TRY(h, hipHostMalloc(ptr, bytes ? bytes : 4, hipHostMallocDefault));
    return DFD_OK;
}

int dfd_host_free(dfd_handle* h, void* ptr) {
    if (!h) return DFD_ERR_ARG;
    DFD_HIP_TRY(h, hipHostFree(ptr));
    return DFD_OK;
}

int dfd_analyze_frames_host(dfd_handle* h, const uint8_t* frames_host, int n_total, int batch, int hh, int ww,
                            const int32_t* forced_xywh, int forced_k, float conf_thr, int max_faces, int apply_clahe,
                            int with_forensics, int32_t* xywh_out, int* n_faces_out, float* logits_out,
                            double* forensic_prob_out) {
    if (!h) return DFD_ERR_ARG;
    if (!frames_host || n_total <= 0 || batch <= 0 || hh <= 0 || ww <= 0 || max_faces <= 0)
        return fail(h, DFD_ERR_ARG, "analyze_frames_host: bad pointer or geometry");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    const size_t frame_bytes = (size_t)hh * ww * 3;
    int rc;
    if (!h->copy_stream) {
        DFD_HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->copy_done[i], hipEventDisableTiming));
            DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->slot_free[i], hipEventDisableTiming));
        }
    }
    for (int i = 0; i < 2; ++i)
        if ((rc = ensure(h, &h->stage[i], (size_t)batch * frame_bytes))) return rc;
    // chunks of `batch` frames (measured, bench.py e2e.detect_classify_h2d: 64 / 32 / 16 frames per chunk land within 3 %
    // of each other and tapering the last chunk does not pay - the gap to the pure upload rate is not a chunking effect)
    std::vector<std::pair<int, int>> chunks;                 // (first frame, count)
    for (int first = 0; first < n_total; first += batch) chunks.push_back({first, std::min(batch, n_total - first)});
    auto upload = [&](int k) -> int {                       // chunk k -> staging slot k & 1, on the copy stream
        const int slot = k & 1, first = chunks[k].first, cnt = chunks[k].second;
        if (k >= 2) DFD_HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->slot_free[slot], 0));   // chunk k - 2 done with it
        DFD_HIP_TRY(h, hipMemcpyAsync(h->stage[slot].p, frames_host + (size_t)first * frame_bytes, (size_t)cnt * frame_bytes,
                                      hipMemcpyHostToDevice, h->copy_stream));
        DFD_HIP_TRY(h, hipEventRecord(h->copy_done[slot], h->copy_stream));
        return DFD_OK;
    };
    const int nb = (int)chunks.size();
    if ((rc = upload(0))) return rc;
    for (int k = 0; k < nb; ++k) {
        const int slot = k & 1, first = chunks[k].first, cnt = chunks[k].second;
        if (k + 1 < nb && (rc = upload(k + 1))) return rc;              // in flight while chunk k computes
        DFD_HIP_TRY(h, hipStreamWaitEvent(h->stream, h->copy_done[slot], 0));
        rc = dfd_analyze_batch_device(h, (const uint8_t*)h->stage[slot].p, cnt, hh, ww,
                                      forced_xywh ? forced_xywh + (size_t)first * forced_k * 4 : nullptr, forced_k, conf_thr,
                                      max_faces, apply_clahe, with_forensics, xywh_out + (size_t)first * max_faces * 4,
                                      n_faces_out + first, logits_out + (size_t)first * max_faces,
                                      forensic_prob_out ? forensic_prob_out + first : nullptr);
        if (rc) return rc;
        DFD_HIP_TRY(h, hipEventRecord(h->slot_free[slot], h->stream));
    }
    DFD_HIP_TRY(h, stream_sync(h));
    return DFD_OK;
}

}  // extern "C"
