// C ABI of the forensic analyzer: device statistics -> the reference's threshold scoring
// (reference frame_analysis.py:58-389), with the per-stream temporal state kept here.
#include <cmath>
#include <deque>

#include "dfd_common.h"
#include "forensic_kernels.h"

using namespace dfd;

namespace dfd {

struct ForensicStream {
    void* prev_gray = nullptr;     // 65536 bytes on the device
    bool has_prev = false;
    std::deque<double> diffs;      // last 30 mean absolute differences
    int frame_count = 0;
};

struct ForensicState {
    std::map<int, ForensicStream> streams;
    DevBuf work;                   // carved by forensic_carve for `cap` frames
    int cap = 0;
    ForensicBuffers buf{};
    float2* twiddle = nullptr;
    double* diff_part = nullptr;   // 256 partial sums
    DevBuf pair_idx, pair_part;    // dfd_forensic_signals_device: predecessor indices, [n][256] partial sums
    double* host_res = nullptr;    // pinned: statistics of a batch that ran on the second stream (forensics_batch_begin)
    size_t host_res_cap = 0;
};

void forensic_destroy(dfd_handle* h) {
    if (h->forensic && h->forensic->host_res) hipHostFree(h->forensic->host_res);
    delete h->forensic;
    h->forensic = nullptr;
}

}  // namespace dfd

namespace {

int state_init(dfd_handle* h, int frames) {
    if (!h->forensic) {
        h->forensic = new ForensicState();
        float2 tw[128];
        for (int k = 0; k < 128; ++k) {
            const double a = -2.0 * M_PI * k / 256.0;
            tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        void* d = nullptr;
        DFD_HIP_TRY(h, hipMalloc(&d, sizeof tw));
        h->owned.push_back(d);
        DFD_HIP_TRY(h, hipMemcpy(d, tw, sizeof tw, hipMemcpyHostToDevice));
        h->forensic->twiddle = static_cast<float2*>(d);
        DFD_HIP_TRY(h, hipMalloc(&d, 256 * 8));
        h->owned.push_back(d);
        h->forensic->diff_part = static_cast<double*>(d);
    }
    ForensicState& F = *h->forensic;
    if (frames > F.cap) {
        const int rc = ensure(h, &F.work, forensic_bytes_per_frame() * frames + 65536);
        if (rc) return rc;
        forensic_carve(F.work.p, frames, &F.buf);
        F.cap = frames;
    }
    return DFD_OK;
}

double pop_std(const double* v, int n, double* mean_out) {
    double m = 0;
    for (int i = 0; i < n; ++i) m += v[i];
    m /= n;
    double q = 0;
    for (int i = 0; i < n; ++i) q += (v[i] - m) * (v[i] - m);
    *mean_out = m;
    return std::sqrt(q / n);
}

double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

// the five stateless signals from the device statistics (frame_analysis.py:150-347); sc[5] (temporal) = 0
void static_scores(const double* st, const double* noise, const double* ela, bool full, double* sc, double* ex) {
    const double nan = std::nan("");
    sc[0] = 0; sc[1] = nan; sc[2] = nan; sc[3] = 0; sc[4] = nan; sc[5] = 0;
    // ---- frequency (:150-180)
    const double lo = st[ST_FREQ_LOW], mi = st[ST_FREQ_MID], hi = st[ST_FREQ_HIGH];
    const double total = lo + mi + hi + 1e-10, hr = hi / total, mr = mi / total;
    const double mid_cv = st[ST_FREQ_MID_STD] / (mi + 1e-10);
    double s = 0.0;
    if (hr < 0.18) s += 0.4; else if (hr < 0.22) s += 0.2;
    if (mid_cv > 0.6) s += 0.25; else if (mid_cv > 0.45) s += 0.1;
    if (mr > 0.45 && hr < 0.2) s += 0.15;
    sc[0] = clip01(s);
    // ---- edges (:296-309)
    const double density = st[ST_EDGE_COUNT] / 65536.0, lap_var = st[ST_LAP_VAR];
    s = 0.0;
    if (density < 0.02) s += 0.35; else if (density < 0.04) s += 0.15;
    if (lap_var < 50) s += 0.3; else if (lap_var < 100) s += 0.1;
    sc[3] = clip01(s);
    double noise_mean = nan, noise_cv = nan, ela_mean = nan, ela_cv = nan;
    if (full) {
        // ---- noise (:207-225)
        noise_cv = pop_std(noise, 64, &noise_mean) / (noise_mean + 1e-10);
        s = 0.0;
        if (noise_cv > 0.7) s += 0.5; else if (noise_cv > 0.5) s += 0.25;
        if (noise_mean < 1.0) s += 0.3; else if (noise_mean < 2.0) s += 0.1;
        sc[1] = clip01(s);
        // ---- ELA (:258-276)
        ela_cv = pop_std(ela, 64, &ela_mean) / (ela_mean + 1e-10);
        s = 0.0;
        if (ela_cv > 0.9) s += 0.5; else if (ela_cv > 0.6) s += 0.2;
        if (ela_mean > 15) s += 0.2; else if (ela_mean > 10) s += 0.1;
        sc[2] = clip01(s);
        // ---- colour (:326-347)
        s = 0.0;
        if (st[ST_SAT_STD] < 15) s += 0.3; else if (st[ST_SAT_STD] < 25) s += 0.1;
        if (st[ST_VAL_STD] < 15) s += 0.25; else if (st[ST_VAL_STD] < 25) s += 0.1;
        if (st[ST_HUES] < 30) s += 0.25; else if (st[ST_HUES] < 50) s += 0.1;
        sc[4] = clip01(s);
    }
    const double e[10] = {lo, mi, hi, hr, mr, mid_cv, noise_mean, noise_cv, ela_mean, ela_cv};
    for (int i = 0; i < 10; ++i) ex[i] = e[i];
}

}  // namespace

namespace dfd {

// the analyzer on a frame that is already in HBM (shared by dfd_forensics and dfd_analyze_frame)
int forensics_run(dfd_handle* h, int stream_id, const uint8_t* frame_dev, int hh, int ww, int stride, int full,
                  double* scores_out, double* prob_out, double* stats_out) {
    if (!h->has_color) return fail(h, DFD_ERR_STATE, "forensics needs the colour tables (blob packed without luts)");
    int rc = state_init(h, 1);
    if (rc) return rc;
    ForensicState& F = *h->forensic;
    ForensicStream& S = F.streams[stream_id];
    if (!S.prev_gray) {
        DFD_HIP_TRY(h, hipMalloc(&S.prev_gray, 65536));
        h->owned.push_back(S.prev_gray);
    }
    S.frame_count += 1;                                              // frame_analysis.py:68,110

    launch_resize_bgr(frame_dev, 1, hh, ww, stride, 0, F.buf.rs, 256, 256, h->stream);
    launch_forensics(F.buf, 1, full != 0, h->color, F.twiddle, h->stream);
    double mean_diff = -1.0;
    if (S.has_prev) launch_absdiff(F.buf.gray, (const uint8_t*)S.prev_gray, F.diff_part, h->stream);
    double st[FORENSIC_STATS], noise[64], ela[64], dpart[256];
    DFD_HIP_TRY(h, hipMemcpyAsync(st, F.buf.stats, sizeof st, hipMemcpyDeviceToHost, h->stream));
    if (full) {
        DFD_HIP_TRY(h, hipMemcpyAsync(noise, F.buf.stats_noise, sizeof noise, hipMemcpyDeviceToHost, h->stream));
        DFD_HIP_TRY(h, hipMemcpyAsync(ela, F.buf.stats_ela, sizeof ela, hipMemcpyDeviceToHost, h->stream));
    }
    if (S.has_prev) DFD_HIP_TRY(h, hipMemcpyAsync(dpart, F.diff_part, sizeof dpart, hipMemcpyDeviceToHost, h->stream));
    DFD_HIP_TRY(h, hipMemcpyAsync(S.prev_gray, F.buf.gray, 65536, hipMemcpyDeviceToDevice, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    DFD_HIP_TRY(h, hipGetLastError());

    const double nan = std::nan("");
    double sc[6], ex[10];
    static_scores(st, noise, ela, full != 0, sc, ex);
    const double lo = ex[0], mi = ex[1], hi = ex[2], hr = ex[3], mr = ex[4], mid_cv = ex[5];
    const double noise_mean = ex[6], noise_cv = ex[7], ela_mean = ex[8], ela_cv = ex[9];
    const double density = st[ST_EDGE_COUNT] / 65536.0, lap_var = st[ST_LAP_VAR];
    double s = 0.0;
    // ---- temporal (:358-389)
    double temporal_cv = nan;
    if (!S.has_prev) {
        S.has_prev = true;
    } else {
        double sum = 0;
        for (int i = 0; i < 256; ++i) sum += dpart[i];
        mean_diff = sum / 65536.0;
        S.diffs.push_back(mean_diff);
        if (S.diffs.size() > 30) S.diffs.pop_front();
        if (S.diffs.size() >= 5) {
            std::vector<double> d(S.diffs.begin(), S.diffs.end());
            double dm;
            temporal_cv = pop_std(d.data(), (int)d.size(), &dm) / (dm + 1e-10);
            s = 0.0;
            if (temporal_cv > 1.5) s += 0.4; else if (temporal_cv > 1.0) s += 0.2;
            if (mean_diff < 0.3 && S.frame_count > 10) s += 0.3;
            else if (mean_diff < 0.8 && S.frame_count > 10) s += 0.1;
            sc[5] = clip01(s);
        }
    }
    // ---- weighted sum in the reference's dict order (:49-56,88 / :118-119)
    double comb = 0.0;
    if (full) {
        const double w[6] = {0.25, 0.20, 0.20, 0.15, 0.10, 0.10};
        for (int i = 0; i < 6; ++i) comb += sc[i] * w[i];
    } else {
        comb += sc[0] * 0.45;
        comb += sc[5] * 0.25;
        comb += sc[3] * 0.30;
    }
    for (int i = 0; i < 6; ++i) scores_out[i] = sc[i];
    *prob_out = clip01(comb);
    if (stats_out) {
        const double out[DFD_FORENSIC_NSTATS] = {lo, mi, hi, hr, mr, mid_cv, noise_mean, noise_cv, ela_mean, ela_cv,
                                                 density, lap_var, full ? st[ST_SAT_STD] : nan, full ? st[ST_VAL_STD] : nan,
                                                 full ? st[ST_HUES] : nan, mean_diff, temporal_cv, (double)S.frame_count};
        for (int i = 0; i < DFD_FORENSIC_NSTATS; ++i) stats_out[i] = out[i];
    }
    return DFD_OK;
}

// n consecutive frames of one stream in one launch set (POST /analyze_batch): the device statistics of all frames at
// once (full mode kernels when any frame is full), frame 0 differenced against the stream's stored gray plane and
// frame i against frame i - 1, then the host half of forensics_run replayed frame by frame in order - the temporal
// deque, the frame counter and the stored plane end exactly where n single calls would leave them.
int forensics_stream_batch_run(dfd_handle* h, int stream_id, const uint8_t* frames_dev, int n, int hh, int ww, int stride,
                               size_t frame_bytes, const int* full, double* scores_out, double* prob_out) {
    if (!h->has_color) return fail(h, DFD_ERR_STATE, "forensics needs the colour tables (blob packed without luts)");
    int rc = state_init(h, n);
    if (rc) return rc;
    ForensicState& F = *h->forensic;
    ForensicStream& S = F.streams[stream_id];
    if (!S.prev_gray) {
        DFD_HIP_TRY(h, hipMalloc(&S.prev_gray, 65536));
        h->owned.push_back(S.prev_gray);
    }
    bool any_full = false;
    for (int f = 0; f < n; ++f) any_full = any_full || full[f] != 0;
    if ((rc = ensure(h, &F.pair_idx, (size_t)n * 4))) return rc;
    if ((rc = ensure(h, &F.pair_part, (size_t)n * 256 * 8))) return rc;
    std::vector<int32_t> prev(n);
    for (int f = 0; f < n; ++f) prev[f] = f - 1;                   // frame 0: the stored plane (below)
    if ((rc = mailbox_h2d(h, F.pair_idx.p, prev.data(), (size_t)n * 4))) return rc;
    launch_resize_bgr(frames_dev, n, hh, ww, stride, frame_bytes, F.buf.rs, 256, 256, h->stream);
    launch_forensics(F.buf, n, any_full, h->color, F.twiddle, h->stream);
    const bool had_prev = S.has_prev;
    if (had_prev) launch_absdiff(F.buf.gray, (const uint8_t*)S.prev_gray, F.diff_part, h->stream);
    if (n > 1) launch_absdiff_pairs(F.buf.gray, (const int*)F.pair_idx.p, (double*)F.pair_part.p, n, h->stream);
    const double* st = (const double*)mailbox_d2h(h, F.buf.stats, (size_t)n * FORENSIC_STATS * 8);
    const double* noise = (const double*)mailbox_d2h(h, F.buf.stats_noise, (size_t)n * 64 * 8);
    const double* ela = (const double*)mailbox_d2h(h, F.buf.stats_ela, (size_t)n * 64 * 8);
    const double* part = (const double*)mailbox_d2h(h, F.pair_part.p, (size_t)n * 256 * 8);
    const double* part0 = (const double*)mailbox_d2h(h, F.diff_part, 256 * 8);
    if (!st || !noise || !ela || !part || !part0) return fail(h, DFD_ERR_HIP, "forensics: mailbox allocation failed");
    DFD_HIP_TRY(h, hipMemcpyAsync(S.prev_gray, F.buf.gray + (size_t)(n - 1) * 65536, 65536, hipMemcpyDeviceToDevice, h->stream));
    DFD_HIP_TRY(h, stream_sync(h));
    DFD_HIP_TRY(h, hipGetLastError());
    for (int f = 0; f < n; ++f) {
        S.frame_count += 1;
        const bool fl = full[f] != 0;
        double sc[6], ex[10];
        static_scores(&st[(size_t)f * FORENSIC_STATS], &noise[(size_t)f * 64], &ela[(size_t)f * 64], fl, sc, ex);
        if (!S.has_prev) {
            S.has_prev = true;
        } else {
            const double* dp = f == 0 ? part0 : part + (size_t)f * 256;
            double sum = 0;
            for (int i = 0; i < 256; ++i) sum += dp[i];
            const double mean_diff = sum / 65536.0;
            S.diffs.push_back(mean_diff);
            if (S.diffs.size() > 30) S.diffs.pop_front();
            if (S.diffs.size() >= 5) {
                std::vector<double> d(S.diffs.begin(), S.diffs.end());
                double dm;
                const double temporal_cv = pop_std(d.data(), (int)d.size(), &dm) / (dm + 1e-10);
                double s = 0.0;
                if (temporal_cv > 1.5) s += 0.4; else if (temporal_cv > 1.0) s += 0.2;
                if (mean_diff < 0.3 && S.frame_count > 10) s += 0.3;
                else if (mean_diff < 0.8 && S.frame_count > 10) s += 0.1;
                sc[5] = clip01(s);
            }
        }
        double comb = 0.0;
        if (fl) {
            const double w[6] = {0.25, 0.20, 0.20, 0.15, 0.10, 0.10};
            for (int i = 0; i < 6; ++i) comb += sc[i] * w[i];
        } else {
            comb += sc[0] * 0.45;
            comb += sc[5] * 0.25;
            comb += sc[3] * 0.30;
        }
        for (int i = 0; i < 6; ++i) scores_out[(size_t)f * 6 + i] = sc[i];
        prob_out[f] = clip01(comb);
    }
    return DFD_OK;
}

// Stateless batch variant for throughput runs: n device frames -> six-signal probability each, the
// temporal signal taking its first-frame value 0 (frame_analysis.py:358-360).  One launch set for all frames.
int forensics_batch_run(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes,
                        double* prob_out, double* scores_out) {
    if (!h->has_color) return fail(h, DFD_ERR_STATE, "forensics needs the colour tables (blob packed without luts)");
    int rc = state_init(h, n);
    if (rc) return rc;
    ForensicState& F = *h->forensic;
    launch_resize_bgr(frames_dev, n, hh, ww, stride, frame_bytes, F.buf.rs, 256, 256, h->stream);
    launch_forensics(F.buf, n, true, h->color, F.twiddle, h->stream);
    // a few KB per frame, through the mailbox (dfd_common.h) rather than the DMA engines
    const double* st = (const double*)mailbox_d2h(h, F.buf.stats, (size_t)n * FORENSIC_STATS * 8);
    const double* noise = (const double*)mailbox_d2h(h, F.buf.stats_noise, (size_t)n * 64 * 8);
    const double* ela = (const double*)mailbox_d2h(h, F.buf.stats_ela, (size_t)n * 64 * 8);
    if (!st || !noise || !ela) return fail(h, DFD_ERR_HIP, "forensics: mailbox allocation failed");
    DFD_HIP_TRY(h, stream_sync(h));
    DFD_HIP_TRY(h, hipGetLastError());
    const double w[6] = {0.25, 0.20, 0.20, 0.15, 0.10, 0.10};
    for (int f = 0; f < n; ++f) {
        double sc[6], ex[10];
        static_scores(&st[(size_t)f * FORENSIC_STATS], &noise[(size_t)f * 64], &ela[(size_t)f * 64], true, sc, ex);
        double comb = 0.0;
        for (int i = 0; i < 6; ++i) comb += sc[i] * w[i];
        prob_out[f] = clip01(comb);
        if (scores_out)
            for (int i = 0; i < 6; ++i) scores_out[(size_t)f * 6 + i] = sc[i];
    }
    return DFD_OK;
}

int forensics_batch_begin(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, int stride, size_t frame_bytes) {
    if (!h->has_color) return fail(h, DFD_ERR_STATE, "forensics needs the colour tables (blob packed without luts)");
    int rc = state_init(h, n);
    if (rc) return rc;
    ForensicState& F = *h->forensic;
    if (!h->aux_stream) {
        // LOWEST priority: the signals have the whole call to finish; their workgroups should take the CUs the main
        // stream leaves idle (DetectionOutput runs 64 blocks on 256 CUs, the detector's tail layers and the cascade's
        // R-/O-Net are small, four waits on the host) instead of competing with its large launches
        int least = 0, greatest = 0;
        DFD_HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        DFD_HIP_TRY(h, hipStreamCreateWithPriority(&h->aux_stream, hipStreamNonBlocking, least));
        DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->aux_go, hipEventDisableTiming));
        DFD_HIP_TRY(h, hipEventCreateWithFlags(&h->aux_done, hipEventDisableTiming));
    }
    const size_t per = FORENSIC_STATS + 128, need = (size_t)n * per * 8;
    if (need > F.host_res_cap) {
        if (F.host_res) DFD_HIP_TRY(h, hipHostFree(F.host_res));
        F.host_res = nullptr;
        F.host_res_cap = 0;
        DFD_HIP_TRY(h, hipHostMalloc((void**)&F.host_res, need, hipHostMallocDefault));
        F.host_res_cap = need;
    }
    // the frames are complete where the main stream stands now (an upload it waited for, a decode it ran)
    DFD_HIP_TRY(h, hipEventRecord(h->aux_go, h->stream));
    DFD_HIP_TRY(h, hipStreamWaitEvent(h->aux_stream, h->aux_go, 0));
    launch_resize_bgr(frames_dev, n, hh, ww, stride, frame_bytes, F.buf.rs, 256, 256, h->aux_stream);
    launch_forensics(F.buf, n, true, h->color, F.twiddle, h->aux_stream);
    copy_kernel_async(F.host_res, F.buf.stats, (size_t)n * FORENSIC_STATS * 8, h->aux_stream);
    copy_kernel_async(F.host_res + (size_t)n * FORENSIC_STATS, F.buf.stats_noise, (size_t)n * 64 * 8, h->aux_stream);
    copy_kernel_async(F.host_res + (size_t)n * (FORENSIC_STATS + 64), F.buf.stats_ela, (size_t)n * 64 * 8, h->aux_stream);
    DFD_HIP_TRY(h, hipGetLastError());
    DFD_HIP_TRY(h, hipEventRecord(h->aux_done, h->aux_stream));
    return DFD_OK;
}

int forensics_batch_end(dfd_handle* h, int n, double* prob_out, double* scores_out) {
    ForensicState& F = *h->forensic;
    DFD_HIP_TRY(h, hipEventSynchronize(h->aux_done));
    const double* st = F.host_res;
    const double* noise = F.host_res + (size_t)n * FORENSIC_STATS;
    const double* ela = F.host_res + (size_t)n * (FORENSIC_STATS + 64);
    const double w[6] = {0.25, 0.20, 0.20, 0.15, 0.10, 0.10};
    for (int f = 0; f < n; ++f) {
        double sc[6], ex[10];
        static_scores(&st[(size_t)f * FORENSIC_STATS], &noise[(size_t)f * 64], &ela[(size_t)f * 64], true, sc, ex);
        double comb = 0.0;
        for (int i = 0; i < 6; ++i) comb += sc[i] * w[i];
        prob_out[f] = clip01(comb);
        if (scores_out)
            for (int i = 0; i < 6; ++i) scores_out[(size_t)f * 6 + i] = sc[i];
    }
    return DFD_OK;
}

}  // namespace dfd

extern "C" {

int dfd_forensics(dfd_handle* h, int stream_id, const uint8_t* bgr, int hh, int ww, int stride, int full,
                  double* scores_out, double* prob_out, double* stats_out) {
    if (!h) return DFD_ERR_ARG;
    if (!bgr || !scores_out || !prob_out || hh <= 0 || ww <= 0 || stride < ww * 3)
        return fail(h, DFD_ERR_ARG, "forensics: bad pointer or geometry");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    const int rc = ensure(h, &h->frame_buf, (size_t)hh * stride);
    if (rc) return rc;
    DFD_HIP_TRY(h, hipMemcpyAsync(h->frame_buf.p, bgr, (size_t)hh * stride, hipMemcpyHostToDevice, h->stream));
    return forensics_run(h, stream_id, (const uint8_t*)h->frame_buf.p, hh, ww, stride, full, scores_out, prob_out, stats_out);
}

int dfd_forensic_signals_device(dfd_handle* h, const uint8_t* frames_dev, int n, int hh, int ww, const int32_t* prev_index,
                                double* scores5_out, double* mean_diff_out) {
    if (!h) return DFD_ERR_ARG;
    if (!frames_dev || n <= 0 || hh <= 0 || ww <= 0 || !prev_index || !scores5_out || !mean_diff_out)
        return fail(h, DFD_ERR_ARG, "forensic_signals: bad pointer or geometry");
    // prev_index[f] = -2: frame f is only somebody's predecessor - it needs a gray plane, no signals.  Such frames
    // form the tail of the batch (the kernels of the signals run on the leading ns frames).
    int ns = n;
    while (ns > 0 && prev_index[ns - 1] == -2) --ns;
    for (int f = 0; f < n; ++f) {
        if (prev_index[f] >= n) return fail(h, DFD_ERR_ARG, "forensic_signals: prev_index[%d] = %d outside the batch", f, prev_index[f]);
        if (prev_index[f] < -2 || (prev_index[f] == -2 && f < ns))
            return fail(h, DFD_ERR_ARG, "forensic_signals: predecessor-only frames (-2) must be the tail of the batch");
    }
    if (ns == 0) return fail(h, DFD_ERR_ARG, "forensic_signals: every frame of the batch is predecessor-only (-2): nothing to compute");
    if (!h->has_color) return fail(h, DFD_ERR_STATE, "forensics needs the colour tables (blob packed without luts)");
    DFD_HIP_TRY(h, hipSetDevice(h->device));
    int rc = state_init(h, n);
    if (rc) return rc;
    ForensicState& F = *h->forensic;
    if ((rc = ensure(h, &F.pair_idx, (size_t)n * 4))) return rc;
    if ((rc = ensure(h, &F.pair_part, (size_t)n * 256 * 8))) return rc;
    const int stride = ww * 3;
    if ((rc = mailbox_h2d(h, F.pair_idx.p, prev_index, (size_t)n * 4))) return rc;
    launch_resize_bgr(frames_dev, n, hh, ww, stride, (size_t)hh * stride, F.buf.rs, 256, 256, h->stream);
    launch_forensics(F.buf, ns, true, h->color, F.twiddle, h->stream, n - ns);
    if (ns > 0) launch_absdiff_pairs(F.buf.gray, (const int*)F.pair_idx.p, (double*)F.pair_part.p, ns, h->stream);
    const size_t nz = ns > 0 ? ns : 1;
    const double* st = (const double*)mailbox_d2h(h, F.buf.stats, nz * FORENSIC_STATS * 8);
    const double* noise = (const double*)mailbox_d2h(h, F.buf.stats_noise, nz * 64 * 8);
    const double* ela = (const double*)mailbox_d2h(h, F.buf.stats_ela, nz * 64 * 8);
    const double* part = (const double*)mailbox_d2h(h, F.pair_part.p, nz * 256 * 8);
    if (!st || !noise || !ela || !part) return fail(h, DFD_ERR_HIP, "forensics: mailbox allocation failed");
    DFD_HIP_TRY(h, stream_sync(h));
    DFD_HIP_TRY(h, hipGetLastError());
    for (int f = ns; f < n; ++f) {                           // predecessor-only frames: no signals
        for (int i = 0; i < 5; ++i) scores5_out[(size_t)f * 5 + i] = -1.0;
        mean_diff_out[f] = -1.0;
    }
    for (int f = 0; f < ns; ++f) {
        double sc[6], ex[10];
        static_scores(&st[(size_t)f * FORENSIC_STATS], &noise[(size_t)f * 64], &ela[(size_t)f * 64], true, sc, ex);
        for (int i = 0; i < 5; ++i) scores5_out[(size_t)f * 5 + i] = sc[i];
        if (prev_index[f] < 0) {
            mean_diff_out[f] = -1.0;
        } else {
            double sum = 0;                                      // the summation order of forensics_run
            for (int i = 0; i < 256; ++i) sum += part[(size_t)f * 256 + i];
            mean_diff_out[f] = sum / 65536.0;
        }
    }
    return DFD_OK;
}

int dfd_forensics_reset(dfd_handle* h, int stream_id) {
    if (!h) return DFD_ERR_ARG;
    if (!h->forensic) return DFD_OK;
    auto it = h->forensic->streams.find(stream_id);
    if (it == h->forensic->streams.end()) return DFD_OK;
    it->second.has_prev = false;                 // frame_analysis.py:391-395
    it->second.diffs.clear();
    it->second.frame_count = 0;
    return DFD_OK;
}

int dfd_forensics_state(dfd_handle* h, int stream_id, int* frame_count, int* n_diffs, int* has_prev) {
    if (!h) return DFD_ERR_ARG;
    int fc = 0, nd = 0, hp = 0;
    if (h->forensic) {
        auto it = h->forensic->streams.find(stream_id);
        if (it != h->forensic->streams.end()) {
            fc = it->second.frame_count;
            nd = (int)it->second.diffs.size();
            hp = it->second.has_prev ? 1 : 0;
        }
    }
    if (frame_count) *frame_count = fc;
    if (n_diffs) *n_diffs = nd;
    if (has_prev) *has_prev = hp;
    return DFD_OK;
}

}  // extern "C"
