// Weights-blob reader (plain C++, no HIP): shared by the library (dfd_create) and the sanitizer harness
// (host_asan_driver.cpp).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>

namespace dfd {

struct Tensor {
    const float* host = nullptr;   // into the caller's blob (valid during dfd_create only)
    float* dev = nullptr;          // device copy owned by the handle
    uint32_t ndim = 0;
    uint32_t dims[4] = {1, 1, 1, 1};
    size_t count = 0;
};

// Blob layout written by weights.serialize(): "DFDW" u32 version u32 count, then
// {char name[48]; u32 ndim; u32 dims[4]; u64 offset; u64 nbytes} entries, then payloads.
inline bool parse_blob(const void* blob, size_t len, std::map<std::string, Tensor>* out, std::string* err) {
    const uint8_t* p = static_cast<const uint8_t*>(blob);
    if (!p || len < 12 || memcmp(p, "DFDW", 4) != 0) { *err = "blob: bad magic"; return false; }
    uint32_t ver, cnt;
    memcpy(&ver, p + 4, 4);
    memcpy(&cnt, p + 8, 4);
    if (ver != 1) { *err = "blob: unsupported version"; return false; }
    const size_t esz = 48 + 4 + 16 + 8 + 8;
    if (12 + (size_t)cnt * esz > len) { *err = "blob: truncated table"; return false; }
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint8_t* e = p + 12 + (size_t)i * esz;
        char name[49];
        memcpy(name, e, 48);
        name[48] = 0;
        Tensor t;
        memcpy(&t.ndim, e + 48, 4);
        memcpy(t.dims, e + 52, 16);
        uint64_t off, nb;
        memcpy(&off, e + 68, 8);
        memcpy(&nb, e + 76, 8);
        if (t.ndim > 4 || off % 4 || nb > len || off > len - nb) { *err = std::string("blob: bad entry ") + name; return false; }
        t.count = 1;
        for (uint32_t d = 0; d < t.ndim; ++d) {
            // four 32-bit extents can wrap a size_t product back onto nb: the count may never exceed the blob itself
            if (t.dims[d] && t.count > len / t.dims[d]) { *err = std::string("blob: size mismatch for ") + name; return false; }
            t.count *= t.dims[d];
        }
        if (t.count * 4 != nb) { *err = std::string("blob: size mismatch for ") + name; return false; }
        t.host = reinterpret_cast<const float*>(p + off);
        (*out)[name] = t;
    }
    return true;
}

}  // namespace dfd
