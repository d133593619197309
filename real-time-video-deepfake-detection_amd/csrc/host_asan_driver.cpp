// Sanitizer harness for the host-only halves of libdfd_hip.so: the code that reads
// bytes nobody here wrote - JPEG markers / Huffman tables / entropy-coded data (jpeg_entropy.h, the body of
// dfd_jpeg_coefficients and of every /analyze JPEG request), the weights-blob table (blob_reader.h) - and the integer box
// logic of the detectors (host_boxes.h), compiled WITHOUT HIP and with -fsanitize=address,undefined.  CPU only:
// `make -C csrc asan-host` -> ../host_asan_driver; tests/test_host_asan.py feeds it the negative corpus.  Never part of
// the product and never run on the GPU box (GPU sanitizers are not available there; this file launches nothing).
//
//   host_asan_driver jpeg  FILE [CHUNKS]   -> "rc=<status> count=<coefficients> hash=<fnv1a of info + tables + coefficients>"
//   host_asan_driver jpegfuzz FILE SEED N  -> N seeded mutations of FILE (byte flips, truncations, marker-length edits) in
//                                             one process: "ok=<decoded> rejected=<errors>"; any memory error aborts
//   host_asan_driver blobfuzz FILE SEED N  -> the same for the weights blob's table
//   host_asan_driver blob  FILE            -> "rc=<0|-2> tensors=<n>"
//   host_asan_driver rows  FILE H W THR    -> SSD rows [n][5] f32 -> "n=<kept> total=<all> boxes..."
//   host_asan_driver rects FILE THR        -> int32 [n][4] -> grouped rectangles
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "blob_reader.h"
#include "host_boxes.h"
#include "jpeg_entropy.h"

namespace dfd {
int fail(dfd_handle*, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fputs("error: ", stderr);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
    return code;
}
}  // namespace dfd

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s jpeg|blob|rows|rects FILE ...\n", argv[0]); return 2; }
    const std::string cmd = argv[1];
    // the exact-size heap copy puts the file's last byte against a redzone
    const std::vector<uint8_t> file = slurp(argv[2]);
    std::vector<uint8_t> data(file.begin(), file.end());
    data.shrink_to_fit();
    if (cmd == "jpeg") {
        if (argc > 3) setenv("DFD_JPEG_CHUNKS", argv[3], 1);
        int info[16] = {0};
        uint16_t q[4 * 64] = {0};
        size_t count = 0;
        int rc = dfd_jpeg::coefficients(data.data(), data.size(), info, q, nullptr, 0, &count);
        uint64_t h = 1469598103934665603ull;
        if (rc == 0) {
            std::vector<int16_t> coef(count);
            rc = dfd_jpeg::coefficients(data.data(), data.size(), info, q, coef.data(), coef.size(), &count);
            h = fnv(h, info, sizeof info);
            h = fnv(h, q, sizeof q);
            h = fnv(h, coef.data(), coef.size() * 2);
        }
        printf("rc=%d count=%zu hash=%016llx\n", rc, count, (unsigned long long)h);
        return 0;
    }
    if ((cmd == "jpegfuzz" || cmd == "blobfuzz") && argc >= 5) {
        uint64_t st = strtoull(argv[3], nullptr, 10) * 2654435761ull + 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        const int N = atoi(argv[4]);
        int ok = 0, bad = 0;
        fclose(stderr);                                     // thousands of expected error lines
        stderr = fopen("/dev/null", "w");
        for (int it = 0; it < N; ++it) {
            std::vector<uint8_t> m(file.begin(), file.end());
            const int kind = (int)(rnd() % 4);
            if (kind == 0 && m.size() > 4) m.resize(4 + rnd() % (m.size() - 4));               // truncation
            const int flips = 1 + (int)(rnd() % 3);
            // headers are the first few hundred bytes: half of the edits land there
            for (int f = 0; f < flips && !m.empty(); ++f) {
                const size_t span = (rnd() & 1) ? std::min<size_t>(m.size(), 700) : m.size();
                m[rnd() % span] = (uint8_t)(kind == 3 ? 0xFF : rnd());
            }
            m.shrink_to_fit();
            if (cmd == "jpegfuzz") {
                int info[16];
                uint16_t q[256];
                size_t count = 0;
                if (it & 1) setenv("DFD_JPEG_CHUNKS", (it & 2) ? "7" : "64", 1); else unsetenv("DFD_JPEG_CHUNKS");
                int rc = dfd_jpeg::coefficients(m.data(), m.size(), info, q, nullptr, 0, &count);
                rc == 0 ? ++ok : ++bad;
            } else {
                std::map<std::string, dfd::Tensor> t;
                std::string err;
                if (dfd::parse_blob(m.data(), m.size(), &t, &err)) {
                    double sum = 0.0;
                    for (auto& kv : t)
                        for (size_t i = 0; i < kv.second.count; ++i) sum += kv.second.host[i];
                    ++ok;
                    if (sum == 12345.678) printf("!");
                } else ++bad;
            }
        }
        printf("ok=%d rejected=%d\n", ok, bad);
        return 0;
    }
    if (cmd == "blob") {
        std::map<std::string, dfd::Tensor> t;
        std::string err;
        const bool ok = dfd::parse_blob(data.data(), data.size(), &t, &err);
        double sum = 0.0;                                   // touch every payload the table admits
        if (ok)
            for (auto& kv : t)
                for (size_t i = 0; i < kv.second.count; ++i) sum += kv.second.host[i];
        printf("rc=%d tensors=%zu sum=%g %s\n", ok ? 0 : DFD_ERR_BLOB, t.size(), sum, err.c_str());
        return 0;
    }
    if (cmd == "rows" && argc >= 6) {
        const int n = (int)(data.size() / 20), hh = atoi(argv[3]), ww = atoi(argv[4]);
        const float thr = (float)atof(argv[5]);
        std::vector<float> rows(n * 5 + 1);
        memcpy(rows.data(), data.data(), (size_t)n * 20);
        const int max_out = 4;
        std::vector<int32_t> xywh(max_out * 4);
        std::vector<float> conf(max_out);
        int total = 0;
        const int k = dfd::ssd_postprocess(rows.data(), n, hh, ww, thr, xywh.data(), conf.data(), max_out, &total);
        printf("n=%d total=%d", k, total);
        for (int i = 0; i < k; ++i) printf(" (%d,%d,%d,%d)", xywh[4 * i], xywh[4 * i + 1], xywh[4 * i + 2], xywh[4 * i + 3]);
        printf("\n");
        return 0;
    }
    if (cmd == "rects" && argc >= 4) {
        const int n = (int)(data.size() / 16);
        std::vector<dfd::Rect> in(n);
        for (int i = 0; i < n; ++i) memcpy(&in[i], data.data() + 16 * (size_t)i, 16);
        const std::vector<dfd::Rect> out = dfd::group_rectangles(in, atoi(argv[3]), 0.2);
        printf("n=%zu", out.size());
        for (const dfd::Rect& r : out) printf(" (%d,%d,%d,%d)", r.x, r.y, r.w, r.h);
        printf("\n");
        return 0;
    }
    fprintf(stderr, "unknown command\n");
    return 2;
}
