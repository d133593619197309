// Host half of the JPEG path: markers, Huffman tables and the entropy-coded segment -> quantised coefficients
// (reference backend_server.py:139-145, cv2.imdecode = libjpeg).  Plain C++ - no HIP in here - so that the code that
// reads bytes from the network also builds into the sanitizer harness (csrc/host_asan_driver.cpp, `make asan-host`,
// tests/test_host_asan.py).  jpeg_decode.hip includes it in front of the device half.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/dfd_hip.h"

namespace dfd {
int fail(dfd_handle* h, int code, const char* fmt, ...);
}

namespace dfd_jpeg {
using dfd::fail;

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr size_t kMaxJpegPixels = (size_t)1 << 26;       // 8192 x 8192: 0.4 GB of coefficients at 4:4:4

struct HuffTable {
    bool present = false;
    uint8_t vals[256];
    int maxcode[18], valptr[17], mincode[17];
    uint16_t look10[1024];                    // 10-bit lookahead: (length << 8) | symbol, 0 = longer code
    // false: the code lengths over-subscribe the code space (jdhuff.c's "code >= 1 << si" check, the Kraft
    // inequality) - such a table would index past look10[] here and below vals[0] in huff_window
    bool build(const uint8_t* bits, const uint8_t* v, int nvals) {
        present = false;
        if (nvals < 0 || nvals > 256) return false;
        memset(vals, 0, sizeof vals);
        memcpy(vals, v, (size_t)nvals);
        int code = 0, k = 0;
        memset(look10, 0, sizeof look10);
        for (int l = 1; l <= 16; ++l) {
            if (code + (int)bits[l] > (1 << l)) return false;
            valptr[l] = k;
            mincode[l] = code;
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                if (l <= 10) {
                    const int base = code << (10 - l);
                    for (int f = 0; f < (1 << (10 - l)); ++f) look10[base + f] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, bw = 0, bh = 0; };   // bw/bh: blocks incl. MCU padding

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Parsed {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    Component comp[3];
    uint16_t q[4][64];                         // natural order
    bool qpresent[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    const uint8_t* scan = nullptr;
    const uint8_t* end = nullptr;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// -> DFD_OK, DFD_ERR_ARG (not a JPEG / truncated / corrupt) or DFD_ERR_UNSUPPORTED
inline int parse_headers(dfd_handle* h, const uint8_t* d, size_t len, Parsed* P) {
    if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(h, DFD_ERR_ARG, "decode_jpeg: not a JPEG (no SOI)");
    size_t pos = 2;
    P->end = d + len;
    bool have_sof = false;
    while (pos + 4 <= len) {
        if (d[pos] != 0xFF) return fail(h, DFD_ERR_ARG, "decode_jpeg: marker expected at byte %zu", pos);
        while (pos < len && d[pos] == 0xFF) ++pos;                 // fill bytes
        if (pos >= len) break;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > len) break;
        const int seg = be16(d + pos);
        if (seg < 2 || pos + seg > len) return fail(h, DFD_ERR_ARG, "decode_jpeg: truncated segment");
        const uint8_t* s = d + pos + 2;
        const int n = seg - 2;
        if (m == 0xDB) {                                           // DQT
            int i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad DQT");
                for (int k = 0; k < 64; ++k) {
                    P->q[tq][kZigzag[k]] = (uint16_t)(pq ? be16(s + i + 2 * k) : s[i + k]);
                }
                P->qpresent[tq] = true;
                i += pq ? 128 : 64;
            }
        } else if (m == 0xC4) {                                    // DHT
            int i = 0;
            while (i + 17 <= n) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                uint8_t bits[17];
                bits[0] = 0;
                int total = 0;
                for (int l = 1; l <= 16; ++l) { bits[l] = s[i + l]; total += bits[l]; }
                if (tc > 1 || th > 3 || total > 256 || i + 17 + total > n) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad DHT");
                if (!(tc ? P->ac[th] : P->dc[th]).build(bits, s + i + 17, total))
                    return fail(h, DFD_ERR_ARG, "decode_jpeg: bad DHT (code lengths over-subscribed)");
                i += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {                       // SOF0 / SOF1: sequential Huffman
            if (n < 6 || s[0] != 8) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: %d-bit samples", n >= 1 ? s[0] : 0);
            P->height = be16(s + 1);
            P->width = be16(s + 3);
            P->ncomp = s[5];
            if (P->ncomp != 1 && P->ncomp != 3) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: %d components", P->ncomp);
            if (n < 6 + 3 * P->ncomp || P->width <= 0 || P->height <= 0) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad SOF");
            // cv2.imdecode is capped by CV_IO_MAX_IMAGE_PIXELS (2^30) and Pillow by MAX_IMAGE_PIXELS; here a header
            // alone would make decode_scan allocate width x height coefficients, so the cap comes first
            if ((size_t)P->width * (size_t)P->height > kMaxJpegPixels)
                return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: %d x %d exceeds the %zu-pixel limit of the GPU path", P->width,
                            P->height, kMaxJpegPixels);
            for (int c = 0; c < P->ncomp; ++c) {
                Component& C = P->comp[c];
                C.id = s[6 + 3 * c];
                C.h = s[7 + 3 * c] >> 4;
                C.v = s[7 + 3 * c] & 15;
                C.tq = s[8 + 3 * c];
                if (C.tq > 3) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad quantisation table index");
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: SOF%d (progressive / lossless / arithmetic) is not decoded on the GPU path", m - 0xC0);
        } else if (m == 0xDD) {
            if (n >= 2) P->restart = be16(s);
        } else if (m == 0xDA) {                                    // SOS
            if (!have_sof) return fail(h, DFD_ERR_ARG, "decode_jpeg: SOS before SOF");
            if (n < 1 || s[0] != P->ncomp || n < 1 + 2 * P->ncomp + 3)
                return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: non-interleaved (multi-scan) file");
            for (int c = 0; c < P->ncomp; ++c) {
                if (s[1 + 2 * c] != P->comp[c].id) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: scan component order");
                P->comp[c].td = s[2 + 2 * c] >> 4;
                P->comp[c].ta = s[2 + 2 * c] & 15;
                if (P->comp[c].td > 3 || P->comp[c].ta > 3) return fail(h, DFD_ERR_ARG, "decode_jpeg: bad table selector");
            }
            P->scan = d + pos + seg;
            break;
        }
        pos += seg;
    }
    if (!have_sof || !P->scan) return fail(h, DFD_ERR_ARG, "decode_jpeg: no frame or scan found");
    // sampling: luma h x v in {1x1, 2x1, 2x2}, chroma 1x1
    if (P->ncomp == 1) { P->comp[0].h = P->comp[0].v = 1; }
    else {
        const Component &Y = P->comp[0], &B = P->comp[1], &R = P->comp[2];
        const bool ok = B.h == 1 && B.v == 1 && R.h == 1 && R.v == 1 &&
                        ((Y.h == 1 && Y.v == 1) || (Y.h == 2 && Y.v == 1) || (Y.h == 2 && Y.v == 2));
        if (!ok) return fail(h, DFD_ERR_UNSUPPORTED, "decode_jpeg: sampling %dx%d,%dx%d,%dx%d", Y.h, Y.v, B.h, B.v, R.h, R.v);
    }
    P->hmax = P->comp[0].h;
    P->vmax = P->comp[0].v;
    for (int c = 0; c < P->ncomp; ++c) {
        if (!P->qpresent[P->comp[c].tq]) return fail(h, DFD_ERR_ARG, "decode_jpeg: quantisation table %d missing", P->comp[c].tq);
        if (!P->dc[P->comp[c].td].present || !P->ac[P->comp[c].ta].present) return fail(h, DFD_ERR_ARG, "decode_jpeg: Huffman table missing");
    }
    return DFD_OK;
}

// ---- host thread pool ------------------------------------------------------------------------------------
// A few persistent threads for the entropy decoder (one request = one scan; spawning threads per request would cost
// what the parallel decode saves).  run(n, fn): fn(i) for i in [0, n) on the pool + the caller, returns when all are
// done.  DFD_HOST_THREADS caps the size (1 = everything on the calling thread).
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    int size() const { return owner_pid_ == getpid() ? (int)workers_.size() + 1 : 1; }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        // a fork()ed child (pre-fork servers) inherits the pool object but not its threads: everything on the caller there
        if (workers_.empty() || n == 1 || owner_pid_ != getpid()) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::unique_lock<std::mutex> call(call_mu_);                 // one parallel region at a time
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); done_ = 0; ++epoch_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return done_ == (int)workers_.size(); });
        fn_ = nullptr;
    }

private:
    HostPool() {
        const char* cap = getenv("DFD_HOST_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        int nt = std::min(cap ? std::max(atoi(cap), 1) : 16, std::max(hw / 2, 1));
        owner_pid_ = getpid();
        for (int t = 1; t < nt; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        if (owner_pid_ != getpid()) {                              // forked child: the threads do not exist here
            for (auto& w : workers_) w.detach();
            return;
        }
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void work() {
        for (int i = next_.fetch_add(1); i < n_; i = next_.fetch_add(1)) (*fn_)(i);
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            work();
            { std::lock_guard<std::mutex> lk(mu_); ++done_; }
            cv_done_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, done_ = 0;
    unsigned long epoch_ = 0;
    bool stop_ = false;
    pid_t owner_pid_ = 0;
};

// ---- entropy-coded segment -> coefficients ------------------------------------------------------------------
// Output: quantised coefficients in natural order per component, blocks row-major with MCU padding.
//
// The scan is first de-stuffed into a flat byte array (FF 00 -> FF, RSTn removed and remembered as segment starts,
// kScanPad zero bytes of padding): after that a bit position is one integer and a symbol is decoded from ONE unaligned
// 64-bit window (Huffman code <= 16 bits + value bits <= 15), with no marker checks in the loop.  A block is only
// started at a bit position inside the payload; inside a block the window is reloaded at most 63 times, and every
// reload checks the position against the payload (decode_block), so the furthest byte a window can touch is
// payload + one reload step (26 + 31 bits) + 8 < kScanPad.
//
// Parallelism over the pool's threads:
//   * files with restart intervals: the segments between RSTn markers are independent (byte aligned, DC prediction
//     reset) - one task per segment;
//   * files without (what browsers and cv2.imencode write): SPECULATIVE chunks.  A Huffman stream resynchronises
//     after a few wrong symbols, so thread t > 0 starts decoding at the first byte of chunk t as if a block of MCU
//     slot 0 began there, keeps its blocks in a private buffer and records the bit position at which each block
//     ended.  Thread 0 decodes the true stream.  Afterwards the chunks are stitched in order: the true decoder's
//     state at a block boundary is (bit position, slot of the next block in its MCU) - the DC predictions are not
//     part of it because blocks carry DC DIFFERENCES until the final pass - and a speculative decoder that ended a
//     block at the same bit position with the same next slot has, by determinism, produced exactly the true blocks
//     from there on.  If chunk t had not yet synchronised where chunk t - 1 ended, the true decoder continues from
//     there block by block until it meets one of chunk t's recorded states (a few blocks; in the worst case the
//     whole chunk, i.e. the sequential algorithm).  Then every block is copied to its place and one pass integrates
//     the DC differences per component.  The result does not depend on the number of threads or the chunk borders.
constexpr size_t kScanPad = 320;               // > the worst-case block (16 + 11 + 63 * (16 + 15) bits = 248 bytes) + a window

struct Destuffed {
    std::vector<uint8_t> bytes;                // + kScanPad bytes of zero padding
    std::vector<size_t> seg_start;             // byte offsets where a restart segment begins (first = 0)
    size_t nbits = 0;                          // payload bits
};

inline void destuff(const uint8_t* p, const uint8_t* end, Destuffed* out) {
    out->bytes.clear();
    out->bytes.reserve((size_t)(end - p) + kScanPad);
    out->seg_start.assign(1, 0);
    while (p < end) {
        const uint8_t* f = static_cast<const uint8_t*>(memchr(p, 0xFF, (size_t)(end - p)));
        if (!f) { out->bytes.insert(out->bytes.end(), p, end); break; }
        out->bytes.insert(out->bytes.end(), p, f);
        if (f + 1 >= end) break;
        const uint8_t m = f[1];
        if (m == 0x00) { out->bytes.push_back(0xFF); p = f + 2; }
        else if (m >= 0xD0 && m <= 0xD7) { out->seg_start.push_back(out->bytes.size()); p = f + 2; }
        else if (m == 0xFF) { p = f + 1; }                         // fill byte
        else break;                                                // EOI or any other marker: end of the scan
    }
    out->nbits = out->bytes.size() * 8;
    out->bytes.insert(out->bytes.end(), kScanPad, 0);
}

inline uint64_t window(const uint8_t* c, uint64_t bp) {           // the 57+ bits that follow bit position bp, left aligned
    uint64_t w;
    memcpy(&w, c + (bp >> 3), 8);
    return __builtin_bswap64(w) << (bp & 7);
}

// one Huffman symbol from the window: returns the symbol (or -1), *len = code length
inline int huff_window(const HuffTable& t, uint64_t w, int* len) {
    const uint16_t e = t.look10[w >> 54];
    if (e) { *len = e >> 8; return e & 0xff; }
    const int code = (int)(w >> 48);
    for (int l = 11; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (c <= t.maxcode[l]) {
            const int idx = t.valptr[l] + c - t.mincode[l];
            *len = l;
            return (idx >= 0 && idx < 256) ? t.vals[idx] : -1;
        }
    }
    *len = 16;
    return -1;
}

// One block at bit position *bp -> blk[64] (natural order; blk[0] = the DC DIFFERENCE).  The caller hands a zeroed
// block.  false: corrupt code / coefficient index (bits were consumed all the same, *bp has advanced).
// A 64-bit window serves several symbols: it is reloaded when fewer than 31 of its 57 guaranteed bits are left - and
// a reload at or past `nbits` (the payload's end) ends the block as truncated: one block can consume up to 1,980
// bits, so without that check a crafted pair of one-bit tables walks the window hundreds of bytes past the scan.
inline bool decode_block(const uint8_t* c, uint64_t* bp, uint64_t nbits, const HuffTable& D, const HuffTable& A, int16_t* blk) {
    uint64_t b = *bp;
    uint64_t w = window(c, b);
    int used = 0;                                                  // bits of `w` already consumed
    int len;
    int s = huff_window(D, w, &len);
    if (s < 0 || s > 11) { *bp = b + len; return false; }
    if (s) blk[0] = (int16_t)extend((int)((w << len) >> (64 - s)), s);
    used = len + s;
    for (int k = 1; k < 64;) {
        if (used > 26) {
            b += used;
            used = 0;
            if (b >= nbits) { *bp = b; return false; }
            w = window(c, b);
        }
        const uint64_t ww = w << used;
        const int rs = huff_window(A, ww, &len);
        if (rs < 0) { *bp = b + used + len; return false; }
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
            used += len;
            if (r != 15) break;
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) { *bp = b + used + len + sz; return false; }
        blk[kZigzag[k]] = (int16_t)extend((int)((ww << len) >> (64 - sz)), sz);
        used += len + sz;
        ++k;
    }
    *bp = b + used;
    return true;
}

struct ScanLayout {
    int bpm = 0;                               // blocks per MCU
    int slot_comp[6], slot_bx[6], slot_by[6];
    int mcux = 0, mcuy = 0;
    size_t total = 0;                          // blocks of the scan
    size_t comp_off[3];
    int bw[3];
};

inline int16_t* block_ptr(const ScanLayout& L, const Parsed& P, int16_t* coef, size_t mcu, int slot) {
    const int c = L.slot_comp[slot];
    const size_t my = mcu / L.mcux, mx = mcu - my * L.mcux;
    return coef + L.comp_off[c] + ((my * P.comp[c].v + L.slot_by[slot]) * (size_t)L.bw[c] + mx * P.comp[c].h + L.slot_bx[slot]) * 64;
}

// `coef` must hold `total` coefficients (any content: every block is written).  -> DFD_OK / DFD_ERR_ARG
// `inner`: use the pool inside this scan (false when the caller already runs one scan per pool thread)
inline int entropy_decode(dfd_handle* h, Parsed* P, int16_t* coef, const ScanLayout& L, bool inner = true) {
    const bool verbose = getenv("DFD_JPEG_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_last = now();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        const auto t = now();
        fprintf(stderr, "[dfd]   %-24s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    Destuffed ds;
    try {
        destuff(P->scan, P->end, &ds);
    } catch (const std::bad_alloc&) {
        return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: out of host memory");
    }
    lap("destuff");
    const uint8_t* c = ds.bytes.data();
    const size_t nbits = ds.nbits;
    HostPool& pool = HostPool::get();
    auto prun = [&](int n, const std::function<void(int)>& fn) {
        if (inner) pool.run(n, fn);
        else for (int i = 0; i < n; ++i) fn(i);
    };
    std::atomic<int> bad{0};
    auto tables = [&](int slot, const HuffTable** D, const HuffTable** A) {
        const Component& C = P->comp[L.slot_comp[slot]];
        *D = &P->dc[C.td];
        *A = &P->ac[C.ta];
    };
    const size_t n_mcu = (size_t)L.mcux * L.mcuy;

    if (P->restart) {
        // independent segments of `restart` MCUs each
        const size_t nseg = (n_mcu + P->restart - 1) / P->restart;
        if (ds.seg_start.size() < nseg) return fail(h, DFD_ERR_ARG, "decode_jpeg: restart marker missing");
        prun((int)nseg, [&](int sidx) {
            uint64_t bp = (uint64_t)ds.seg_start[sidx] * 8;
            int pred[3] = {0, 0, 0};
            const size_t m0 = (size_t)sidx * P->restart, m1 = std::min(n_mcu, m0 + P->restart);
            for (size_t m = m0; m < m1; ++m)
                for (int slot = 0; slot < L.bpm; ++slot) {
                    int16_t* blk = block_ptr(L, *P, coef, m, slot);
                    memset(blk, 0, 128);
                    const HuffTable *D, *A;
                    tables(slot, &D, &A);
                    if (bp >= nbits || !decode_block(c, &bp, nbits, *D, *A, blk)) { bad.store(1); return; }
                    const int cc = L.slot_comp[slot];
                    pred[cc] += blk[0];
                    blk[0] = (int16_t)pred[cc];
                }
        });
        if (bad.load()) return fail(h, DFD_ERR_ARG, "decode_jpeg: corrupt or truncated entropy-coded data");
        return DFD_OK;
    }

    // ---- no restart markers: speculative chunks
    struct Chunk {
        std::vector<int16_t> blocks;           // local blocks, 64 coefficients each (DC = difference)
        std::vector<uint64_t> end_bit;         // bit position after local block i
        uint64_t start = 0, stop = 0;          // bit range [start, stop)
        bool overflow = false;
    };
    int T = inner ? pool.size() : 1;
    if (nbits < (size_t)T * 32768) T = std::max<int>(1, (int)(nbits / 32768));   // small files: fewer chunks
    if (const char* e = getenv("DFD_JPEG_CHUNKS")) T = std::max(1, std::min(atoi(e), 256));     // tests: any chunking, same result
    if ((size_t)T > nbits / 64) T = std::max<int>(1, (int)(nbits / 64));
    std::vector<Chunk> ch(T);
    for (int t = 0; t < T; ++t) {
        ch[t].start = (nbits / 8 * t / T) * 8;
        ch[t].stop = t + 1 < T ? (nbits / 8 * (t + 1) / T) * 8 : nbits;
    }
    const size_t expect = L.total / T + 64;
    prun(T, [&](int t) {
        Chunk& K = ch[t];
        try {
            K.blocks.reserve((expect + expect / 2) * 64);
            K.end_bit.reserve(expect + expect / 2);
        } catch (const std::bad_alloc&) { K.overflow = true; return; }
        uint64_t bp = K.start;
        // thread 0 is the true stream; the others assume slot 0 at their first byte.  Each runs to the first block
        // boundary at or past its chunk's end (the last one: until the scan's blocks are done or the data ends).
        const size_t cap = t == 0 ? L.total : std::max<size_t>(4 * expect, 4096);
        size_t i = 0;
        while (bp < K.stop && i < cap) {
            const int slot = (int)(i % L.bpm);
            const HuffTable *D, *A;
            tables(slot, &D, &A);
            if ((i + 1) * 64 > K.blocks.size()) {
                try {
                    K.blocks.resize(K.blocks.size() + 4096 * 64, 0);         // zeroed blocks, 4096 at a time
                } catch (const std::bad_alloc&) { K.overflow = true; return; }
            }
            int16_t* blk = K.blocks.data() + i * 64;
            if (!decode_block(c, &bp, nbits, *D, *A, blk)) {
                if (t == 0) { bad.store(1); return; }              // the true stream is corrupt
                memset(blk, 0, 128);                               // speculative garbage: carry on from here
            }
            K.end_bit.push_back(bp);
            ++i;
        }
        if (i >= cap && bp < K.stop && t > 0) K.overflow = true;    // never synchronised into a sane block rate
        if (verbose) fprintf(stderr, "[dfd]   chunk %d done at +%.3f ms on thread %zu\n", t,
                             std::chrono::duration<double, std::milli>(now() - t_last).count(),
                             std::hash<std::thread::id>{}(std::this_thread::get_id()) % 1000);
    });
    if (bad.load()) return fail(h, DFD_ERR_ARG, "decode_jpeg: corrupt entropy-coded data");
    lap("speculative chunks");

    // ---- stitch: (source, first local block, count, global first block)
    struct Piece { const int16_t* src; size_t count, gstart; };
    std::vector<Piece> pieces;
    std::vector<std::vector<int16_t>> patches;                      // blocks the true decoder had to add between chunks
    size_t G = 0;                                                   // global blocks placed so far
    uint64_t bp = 0;                                                // true bit position after them
    {
        const Chunk& K = ch[0];
        if (K.overflow) return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: out of host memory");
        const size_t cnt = std::min(K.end_bit.size(), L.total);
        pieces.push_back(Piece{K.blocks.data(), cnt, 0});
        G = cnt;
        bp = cnt ? K.end_bit[cnt - 1] : 0;
    }
    for (int t = 1; t < T && G < L.total; ++t) {
        const Chunk& K = ch[t];
        // does the true state (bp, slot G % bpm) coincide with a recorded boundary of chunk t?
        size_t match = (size_t)-1;
        auto find = [&](uint64_t pos, size_t g) -> size_t {
            if (K.overflow) return (size_t)-1;
            auto it = std::lower_bound(K.end_bit.begin(), K.end_bit.end(), pos);
            for (; it != K.end_bit.end() && *it == pos; ++it) {    // (zero-length blocks cannot occur: every block consumes bits)
                const size_t i = (size_t)(it - K.end_bit.begin());
                if ((i + 1) % L.bpm == g % L.bpm) return i;
            }
            return (size_t)-1;
        };
        match = bp >= K.start ? find(bp, G) : (size_t)-1;
        if (match == (size_t)-1) {
            // the true decoder walks on until it meets chunk t's stream, the chunk ends, or the scan is complete
            patches.emplace_back();
            std::vector<int16_t>& pb = patches.back();
            const size_t g0 = G;
            while (G < L.total && bp < K.stop && bp < nbits) {
                const int slot = (int)(G % L.bpm);
                const HuffTable *D, *A;
                tables(slot, &D, &A);
                try {
                    pb.resize(pb.size() + 64, 0);
                } catch (const std::bad_alloc&) { return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: out of host memory"); }
                if (!decode_block(c, &bp, nbits, *D, *A, pb.data() + pb.size() - 64))
                    return fail(h, DFD_ERR_ARG, "decode_jpeg: corrupt entropy-coded data");
                ++G;
                if (bp >= K.start && (match = find(bp, G)) != (size_t)-1) break;
            }
            // src filled in below (vector may move).  A walk of zero blocks (the true decoder was already past this chunk:
            // chunks shorter than one block) leaves no piece - and must leave no patch either, or the creation-order
            // pairing below hands a later piece the empty buffer (found by the sanitizer harness at 256 chunks)
            if (G > g0) pieces.push_back(Piece{nullptr, G - g0, g0});
            else patches.pop_back();
        }
        if (match != (size_t)-1) {
            const size_t first = match + 1, avail = K.end_bit.size() - first;
            const size_t cnt = std::min(avail, L.total - G);
            if (cnt) {
                pieces.push_back(Piece{K.blocks.data() + first * 64, cnt, G});
                G += cnt;
                bp = K.end_bit[first + cnt - 1];
            }
        }
    }
    // whatever is still missing (the last chunk never synchronised, or ended early): the true decoder finishes
    if (G < L.total) {
        patches.emplace_back();
        std::vector<int16_t>& pb = patches.back();
        const size_t g0 = G;
        while (G < L.total) {
            if (bp >= nbits) return fail(h, DFD_ERR_ARG, "decode_jpeg: truncated entropy-coded data (%zu of %zu blocks)", G, L.total);
            const int slot = (int)(G % L.bpm);
            const HuffTable *D, *A;
            tables(slot, &D, &A);
            try {
                pb.resize(pb.size() + 64, 0);
            } catch (const std::bad_alloc&) { return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: out of host memory"); }
            if (!decode_block(c, &bp, nbits, *D, *A, pb.data() + pb.size() - 64))
                return fail(h, DFD_ERR_ARG, "decode_jpeg: corrupt entropy-coded data");
            ++G;
        }
        pieces.push_back(Piece{nullptr, G - g0, g0});
    }
    if (bp > nbits) return fail(h, DFD_ERR_ARG, "decode_jpeg: truncated entropy-coded data");      // the last block ran into the padding
    if (getenv("DFD_JPEG_VERBOSE")) {
        size_t patched = 0;
        for (auto& pb : patches) patched += pb.size() / 64;
        fprintf(stderr, "[dfd] jpeg entropy: %d chunks, %zu blocks, %zu decoded again by the stitcher (%zu patches)\n", T, L.total, patched, patches.size());
        for (int t = 0; t < T; ++t) fprintf(stderr, "[dfd]   chunk %d: %zu local blocks%s\n", t, ch[t].end_bit.size(), ch[t].overflow ? " (overflow)" : "");
    }
    {   // patch pieces -> their buffers (in creation order)
        size_t pi = 0;
        for (Piece& pc : pieces)
            if (!pc.src) pc.src = patches[pi++].data();
    }
    lap("stitch");
    // ---- every block to its place and the DC differences integrated, both piece-parallel: while copying, a piece turns
    // its differences into prefix sums that start at 0; the pieces' totals are chained in order (a few dozen adds); a
    // second sweep over the DC terms alone adds each piece's base (the lines are still in that thread's cache - the
    // one sequential pass over 49k blocks it replaces took 0.6 of the decoder's 2.6 ms: every line came from another core)
    struct Sum3 { int v[3]; };
    std::vector<Sum3> psum(pieces.size(), Sum3{{0, 0, 0}}), pbase(pieces.size(), Sum3{{0, 0, 0}});
    auto walk = [&](const Piece& pc, auto&& fn) {
        const size_t mcu0 = pc.gstart / L.bpm;
        size_t my = mcu0 / L.mcux, mx = mcu0 - my * L.mcux;
        int slot = (int)(pc.gstart % L.bpm);
        for (size_t k = 0; k < pc.count; ++k) {
            const int cc = L.slot_comp[slot];
            int16_t* dst = coef + L.comp_off[cc] +
                           ((my * P->comp[cc].v + L.slot_by[slot]) * (size_t)L.bw[cc] + mx * P->comp[cc].h + L.slot_bx[slot]) * 64;
            fn(k, cc, dst);
            if (++slot == L.bpm) {
                slot = 0;
                if (++mx == (size_t)L.mcux) { mx = 0; ++my; }
            }
        }
    };
    prun((int)pieces.size(), [&](int i) {
        const Piece& pc = pieces[i];
        int run[3] = {0, 0, 0};
        walk(pc, [&](size_t k, int cc, int16_t* dst) {
            memcpy(dst, pc.src + k * 64, 128);
            run[cc] += dst[0];
            dst[0] = (int16_t)run[cc];
        });
        for (int c2 = 0; c2 < 3; ++c2) psum[i].v[c2] = run[c2];
    });
    lap("scatter");
    for (size_t i = 1; i < pieces.size(); ++i)
        for (int c2 = 0; c2 < 3; ++c2) pbase[i].v[c2] = pbase[i - 1].v[c2] + psum[i - 1].v[c2];
    prun((int)pieces.size(), [&](int i) {
        if (i == 0) return;
        const Sum3 base = pbase[i];
        walk(pieces[i], [&](size_t, int cc, int16_t* dst) { dst[0] = (int16_t)(dst[0] + base.v[cc]); });
    });
    lap("DC pass");
    return DFD_OK;
}

// geometry of the scan: component block grids, MCU slots
inline void scan_layout(Parsed* P, ScanLayout* L) {
    const int mcu_w = 8 * P->hmax, mcu_h = 8 * P->vmax;
    L->mcux = (P->width + mcu_w - 1) / mcu_w;
    L->mcuy = (P->height + mcu_h - 1) / mcu_h;
    size_t total = 0;
    L->bpm = 0;
    for (int c = 0; c < P->ncomp; ++c) {
        Component& C = P->comp[c];
        C.bw = L->mcux * C.h;
        C.bh = L->mcuy * C.v;
        L->bw[c] = C.bw;
        L->comp_off[c] = total;
        total += (size_t)C.bw * C.bh * 64;
        for (int by = 0; by < C.v; ++by)
            for (int bx = 0; bx < C.h; ++bx) {
                L->slot_comp[L->bpm] = c;
                L->slot_bx[L->bpm] = bx;
                L->slot_by[L->bpm] = by;
                ++L->bpm;
            }
    }
    L->total = (size_t)L->mcux * L->mcuy * L->bpm;
}

inline int decode_scan(dfd_handle* h, Parsed* P, std::vector<int16_t>* coef, size_t* comp_off) {
    ScanLayout L;
    scan_layout(P, &L);
    for (int c = 0; c < P->ncomp; ++c) comp_off[c] = L.comp_off[c];
    try {
        coef->resize(L.total * 64);
    } catch (const std::bad_alloc&) {                              // never unwinds through the C ABI
        return fail(h, DFD_ERR_CAPACITY, "decode_jpeg: %zu coefficients do not fit in host memory", L.total * 64);
    }
    return entropy_decode(h, P, coef->data(), L);
}

// dfd_jpeg_coefficients without the C ABI around it (dfd_hip.h: info[16], qtables 4 x 64, coefficients)
inline int coefficients(const uint8_t* jpeg, size_t len, int* info, uint16_t* qtables_out, int16_t* coef_out, size_t capacity,
                        size_t* count) {
    if (!jpeg || !info || !count) return fail(nullptr, DFD_ERR_ARG, "jpeg_coefficients: null pointer");
    Parsed P;
    int rc = parse_headers(nullptr, jpeg, len, &P);
    if (rc) return rc;
    std::vector<int16_t> coef;
    size_t off[3] = {0, 0, 0};
    if ((rc = decode_scan(nullptr, &P, &coef, off))) return rc;
    info[0] = P.width; info[1] = P.height; info[2] = P.ncomp; info[3] = P.hmax; info[4] = P.vmax;
    for (int c = 0; c < 3; ++c) {
        info[5 + 3 * c] = c < P.ncomp ? P.comp[c].bw : 0;
        info[6 + 3 * c] = c < P.ncomp ? P.comp[c].bh : 0;
        info[7 + 3 * c] = c < P.ncomp ? P.comp[c].tq : 0;
    }
    *count = coef.size();
    if (qtables_out) memcpy(qtables_out, P.q, sizeof P.q);
    if (coef_out) {
        if (coef.size() > capacity) return fail(nullptr, DFD_ERR_ARG, "jpeg_coefficients: capacity");
        memcpy(coef_out, coef.data(), coef.size() * 2);
    }
    return DFD_OK;
}

}  // namespace dfd_jpeg
