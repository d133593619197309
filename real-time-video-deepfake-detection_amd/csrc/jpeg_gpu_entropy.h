// Entropy decoding of baseline JPEG scans ON THE DEVICE (round 4; SURVEY 8(f) N2, reference backend_server.py:139-145:
// cv2.imdecode = libjpeg).  With the Huffman decoder on host threads a batch of frames had to cross PCIe as 16-bit
// coefficients - as many bytes as the raw frames (6.2 MB per 1080p frame) - and the host managed ~0.9 k frames/s per 16
// cores.  Here the JPEG BYTES cross (0.3-1.2 MB per frame) and the scan is decoded by one lane per 512-byte chunk:
//
//   1. de-stuffing (FF 00 -> FF) as a stream compaction: jg_count_kernel counts the stuffed zeros per 16-KB block and
//      finds the first marker (end of the entropy-coded data), jg_compact_kernel writes the flat byte array - after
//      that a bit position is one integer, as in the host decoder (jpeg_entropy.h);
//   2. the host decoder's speculative chunks as a fixed-point iteration.  A Huffman stream resynchronises after a few
//      symbols, so lane t > 0 decodes chunk t from its first byte AS IF a block of MCU slot 0 began there and records
//      where it ended: the first block boundary (bit position, slot of the next block) at or past the chunk's end
//      (round 0).  In round r every lane takes its predecessor's end state of round r - 1 as its start; a lane whose
//      start did not change keeps its result.  After round r the chunks 0 .. r hold the true decode whatever the data,
//      and on real scans everything is final after round 1 (round 2 .. only confirm, at no cost: whole waves skip).  The
//      rounds count blocks and sum DC differences per component, nothing is written;
//   3. jg_scan_kernel: per frame, prefix sums over the chunks give every chunk its first block index and its DC
//      predictions; it also CHECKS the fixed point (start of chunk t == end of chunk t - 1, slot == block index mod
//      blocks per MCU, block total == the frame's) - a frame that fails any check is decoded by the host path instead;
//   4. jg_emit_kernel decodes every chunk once more from its true start and writes the coefficients (natural order,
//      DC integrated) where the IDCT kernel reads them; the buffer is zeroed first (most coefficients are zero).
//
// Decoding work: three passes over the scan instead of one, on ~2,400 lanes per 1080p frame.  The result does not depend
// on the chunk size or the number of rounds (tests: equal to the host decoder's coefficients, i.e. pinned to libjpeg
// through the oracle's IDCT, and the decoded frames equal Pillow's bit for bit).
// Restart-interval files, more than four distinct Huffman tables and anything the header parser rejects stay on the host
// path.  Included by jpeg_decode.hip (shares Parsed / ScanLayout / HuffTable with the host decoder).
#pragma once

namespace dfd_jpeg_gpu {

using dfd_jpeg::HuffTable;
using dfd_jpeg::Parsed;
using dfd_jpeg::ScanLayout;

constexpr int JG_LUT_BITS = 12, JG_LUT = 1 << JG_LUT_BITS;
constexpr int JG_TABLES = 4;                     // distinct Huffman tables of a frame (DC + AC)
constexpr int JG_MAX_ROUNDS = 32;                // rounds a call may ask for (option "jpeg_rounds"; the scan kernel verifies the result)
constexpr int JG_DS_THREADS = 256, JG_DS_PIECES = 4, JG_DS_BLOCK = JG_DS_THREADS * JG_DS_PIECES * 16;   // 16 KB per block
constexpr int JG_CB = 256;                       // chunks (lanes) per block of the decode kernels

struct JgTableSet {                              // device image of one frame's Huffman tables
    uint16_t lut[JG_TABLES][JG_LUT];             // (length << 8) | symbol for codes of <= 12 bits, 0 otherwise
    int32_t maxcode[JG_TABLES][4], mincode[JG_TABLES][4], valptr[JG_TABLES][4];   // lengths 13 .. 16
    uint8_t vals[JG_TABLES][256];
};

struct JgFrame {
    uint32_t raw_off, raw_len;                   // the scan's bytes in the uploaded buffer (any alignment: the de-stuffing kernels
                                                 // read whole 16-byte pieces from raw_off & ~15 and skip the bytes in front)
    uint32_t ds_off;                             // its de-stuffed bytes (% 16 == 0, capacity raw_len + 64)
    uint32_t chunk0, nchunks, cblk0;             // chunk range, first chunk block
    uint32_t dsblk0, ndsblk;                     // de-stuffing blocks
    uint32_t coef_off;                           // int16 elements from the coefficient base
    uint32_t tabset;
    int32_t bpm, total_blocks, mcux, chunk_bytes, cw_shift;   // cw_shift = log2(chunk_bytes / 4)
    uint8_t slot_comp[8], slot_bx[8], slot_by[8], slot_dc[8], slot_ac[8];
    int32_t comp_h[3], comp_v[3], comp_bw[3];
    uint32_t comp_off[3];
    // written by the device
    uint32_t marker_pos;                         // raw offset of the first marker (0xffffffff before jg_count_kernel)
    uint32_t nbits;                              // payload bits of the de-stuffed scan
    int32_t status;                              // 0 = decoded; else why not (JgStatus)
    uint32_t blocks_found;
};
enum JgStatus { JG_OK = 0, JG_NOT_CONVERGED = 1, JG_BLOCK_COUNT = 2, JG_BAD_CODE = 3, JG_SLOT = 4 };

struct JgChunks {                                // per chunk, structure of arrays
    uint2* st;                                   // start state (bit position, slot)
    uint2* en[2];                                // end state of the last two rounds
    uint32_t* cnt;                               // complete blocks that start in the chunk
    int32_t* dcs;                                // [3] sum of their DC differences per component
    uint32_t* gfirst;                            // first block index (scan kernel)
    int32_t* dcb;                                // [3] DC prediction at the chunk's first block
    uint8_t* err;                                // the chunk's decode met an invalid code / coefficient index
    uint32_t* redone;                            // [JG_MAX_ROUNDS] lanes that decoded in round r (diagnostics)
};

// Layout of a frame's de-stuffed scan in memory: CHUNK-INTERLEAVED dwords.  Lane t of a wave walks chunk t one dword at a
// time; stored flat, the 64 lanes of a wave streamed from 64 different cache lines per refill and every refill was an L2
// round trip (measured: 1.9 ms per pass over 64 x 1.2 MB, ~2,500 SIMD cycles per symbol step).  Here dword j of chunk c
// sits at ((c / 64) * CW + j) * 64 + c % 64 (CW = dwords per chunk, a power of two): the 64 lanes of a wave read
// neighbouring dwords of a few lines that stay in the L1 while the lanes drift apart by a few dwords.
__host__ __device__ __forceinline__ uint32_t jg_dword_at(uint32_t g, int cw_shift) {
    const uint32_t c = g >> cw_shift, j = g & ((1u << cw_shift) - 1u);
    return (((c >> 6) << cw_shift) + j) * 64u + (c & 63u);
}

// ---------------------------------------------------------------------------------------------- de-stuffing
// piece q of a block = its bytes [16 q, 16 q + 16): thread t takes the pieces t, t + 256, ... (coalesced 16-byte loads)
// (positions in [lead, raw_len) are the scan; a piece may start in front of it)
__device__ __forceinline__ void jg_piece_scan(uint32_t lead, uint32_t raw_len, uint32_t pos0, const uint8_t (&b)[16], uint8_t prev,
                                              int* removed, uint32_t* marker) {
    int rem = 0;
    uint32_t mk = 0xffffffffu;
    uint8_t p = prev;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t pos = pos0 + i;
        if (pos == lead) p = 0;                                    // nothing in front of the first byte
        if (pos >= lead && pos < raw_len) {
            if (p == 0xFF) {
                if (b[i] == 0) ++rem;
                else if (mk == 0xffffffffu) mk = pos - 1;
            }
            if (pos + 1 == raw_len && b[i] == 0xFF && mk == 0xffffffffu) mk = pos;   // an FF as the very last byte ends the data too
        }
        p = b[i];
    }
    *removed = rem;
    *marker = mk;
}

__global__ __launch_bounds__(JG_DS_THREADS) void jg_count_kernel(const uint8_t* __restrict__ raw, JgFrame* __restrict__ F,
                                                                 const uint16_t* __restrict__ blk_frame, uint32_t* __restrict__ blk_removed) {
    __shared__ int red[JG_DS_THREADS / 64];
    const int f = blk_frame[blockIdx.x], tid = threadIdx.x;
    // positions count from the 16-byte boundary in front of the scan: its bytes are [lead, lead + raw_len)
    const uint32_t lead = F[f].raw_off & 15u, raw_len = F[f].raw_len + lead;
    const uint32_t base = (blockIdx.x - F[f].dsblk0) * (uint32_t)JG_DS_BLOCK;
    const uint8_t* scan = raw + (F[f].raw_off - lead);
    int total = 0;
    uint32_t mk = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < JG_DS_PIECES; ++i) {
        const uint32_t pos0 = base + (uint32_t)(i * JG_DS_THREADS + tid) * 16;
        uint8_t b[16];
        const uint4 v = pos0 < raw_len ? *reinterpret_cast<const uint4*>(scan + pos0) : make_uint4(0, 0, 0, 0);   // (capacity is padded)
        memcpy(b, &v, 16);
        const uint8_t prev = pos0 > lead && pos0 <= raw_len ? scan[pos0 - 1] : 0;
        int rem;
        uint32_t m;
        jg_piece_scan(lead, raw_len, pos0, b, prev, &rem, &m);
        total += rem;
        mk = m < mk ? m : mk;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        total += __shfl_xor(total, off);
        const uint32_t o = __shfl_xor(mk, off);
        mk = o < mk ? o : mk;
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = total;
        if (mk != 0xffffffffu) atomicMin(&F[f].marker_pos, mk);
    }
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int w = 0; w < JG_DS_THREADS / 64; ++w) s += red[w];
        blk_removed[blockIdx.x] = (uint32_t)s;
    }
}

__global__ __launch_bounds__(JG_DS_THREADS) void jg_compact_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ ds,
                                                                   JgFrame* __restrict__ F, const uint16_t* __restrict__ blk_frame,
                                                                   const uint32_t* __restrict__ blk_removed) {
    __shared__ int cnt[JG_DS_THREADS * JG_DS_PIECES];            // stuffed zeros per piece, then their exclusive prefix
    __shared__ int wsum[JG_DS_THREADS / 64 + 1];
    __shared__ int before;
    const int f = blk_frame[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lead = F[f].raw_off & 15u, raw_len = F[f].raw_len + lead, b0 = F[f].dsblk0;
    const uint32_t base = (blockIdx.x - b0) * (uint32_t)JG_DS_BLOCK;
    const uint8_t* scan = raw + (F[f].raw_off - lead);
    uint8_t* out = ds + F[f].ds_off;
    const int cw_shift = F[f].cw_shift;
    // stuffed zeros in the frame's blocks before this one
    {
        int s = 0;
        for (uint32_t b = b0 + tid; b < blockIdx.x; b += JG_DS_THREADS) s += (int)blk_removed[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) wsum[wave] = s;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < JG_DS_THREADS / 64; ++w) t += wsum[w];
            before = t;
        }
        __syncthreads();
    }
    uint8_t b[JG_DS_PIECES][16];
    uint8_t prev[JG_DS_PIECES];
#pragma unroll
    for (int i = 0; i < JG_DS_PIECES; ++i) {
        const uint32_t pos0 = base + (uint32_t)(i * JG_DS_THREADS + tid) * 16;
        const uint4 v = pos0 < raw_len ? *reinterpret_cast<const uint4*>(scan + pos0) : make_uint4(0, 0, 0, 0);
        memcpy(b[i], &v, 16);
        prev[i] = pos0 > lead && pos0 <= raw_len ? scan[pos0 - 1] : 0;
        int rem;
        uint32_t m;
        jg_piece_scan(lead, raw_len, pos0, b[i], prev[i], &rem, &m);
        cnt[i * JG_DS_THREADS + tid] = rem;
    }
    __syncthreads();
    // exclusive prefix over the 1024 pieces in byte order: thread t scans the pieces 4 t .. 4 t + 3, then the block
    int mine[JG_DS_PIECES], tot = 0;
#pragma unroll
    for (int i = 0; i < JG_DS_PIECES; ++i) { mine[i] = tot; tot += cnt[JG_DS_PIECES * tid + i]; }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < JG_DS_PIECES; ++i) cnt[JG_DS_PIECES * tid + i] = wbase + incl - tot + mine[i];
    __syncthreads();
    const uint32_t end = F[f].marker_pos < raw_len ? F[f].marker_pos : raw_len;   // one past the last payload byte
#pragma unroll
    for (int i = 0; i < JG_DS_PIECES; ++i) {
        const int q = i * JG_DS_THREADS + tid;
        const uint32_t pos0 = base + (uint32_t)q * 16;
        if (pos0 >= raw_len) continue;
        // output index of the piece's first scan byte: scan bytes in front of it minus the stuffed zeros among them
        uint32_t o = (pos0 > lead ? pos0 - lead : 0u) - (uint32_t)before - (uint32_t)cnt[q];
        uint8_t p = prev[i];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t pos = pos0 + k;
            if (pos == lead) p = 0;
            if (pos >= lead && pos < raw_len) {
                const bool stuffed = p == 0xFF && b[i][k] == 0;
                if (!stuffed) { out[4u * jg_dword_at(o >> 2, cw_shift) + (o & 3u)] = b[i][k]; ++o; }
                if (pos + 1 == end) F[f].nbits = 8u * o;              // (end == 0: set by the host to 0 beforehand)
            }
            p = b[i][k];
        }
    }
}

// ---------------------------------------------------------------------------------------------- the decoder
struct JgLds {
    JgTableSet tab;
    JgFrame fr;
    uint4 slot[8];                               // per MCU slot: coefficient offset of its block in MCU (0, 0), per MCU row, per MCU column
    uint8_t zigzag[64];
};
__device__ const uint8_t jg_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ __forceinline__ int jg_extend(uint32_t v, int s) {
    return s == 0 ? 0 : ((int)v < (1 << (s - 1)) ? (int)v - (1 << s) + 1 : (int)v);
}

// Decodes blocks from bit position p0 (slot z0) of the de-stuffed scan `ds`: !EMIT - up to the first block boundary at
// or past `pend`; EMIT - exactly `want` blocks, written to coef.  Counts only COMPLETE blocks (every bit inside nbits).
template <bool EMIT>
__device__ __forceinline__ void jg_decode(const JgLds& L, const uint32_t* __restrict__ ds, uint32_t nbits, uint32_t p0, int z0,
                                          uint32_t pend, uint32_t want, uint32_t* p_out, int* z_out, uint32_t* n_out, int (&dsum)[3],
                                          bool* bad_out, int16_t* __restrict__ coef, uint32_t g0, const int (&pred0)[3]) {
    const int bpm = L.fr.bpm, cws = L.fr.cw_shift;
    // slot -> DC table / AC table / component as nibbles of three registers: no LDS read on the symbol loop's critical path
    uint32_t dcmap = 0, acmap = 0, cmap = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        dcmap |= (uint32_t)L.fr.slot_dc[i] << (4 * i);
        acmap |= (uint32_t)L.fr.slot_ac[i] << (4 * i);
        cmap |= (uint32_t)L.fr.slot_comp[i] << (4 * i);
    }
    uint32_t idx = p0 >> 5;
    uint64_t buf = ((uint64_t)__builtin_bswap32(ds[jg_dword_at(idx, cws)]) << 32 | __builtin_bswap32(ds[jg_dword_at(idx + 1, cws)])) << (p0 & 31);
    int have = 64 - (int)(p0 & 31);
    // The next dword of the lane's stream is requested in EVERY step (all lanes, whether the last one was consumed or
    // not) and used one step later: a refill under `if (have < 32)` made the whole wave wait for the load of whichever lane
    // refilled last - a memory round trip in every step of every wave.
    uint32_t ridx = idx + 2;
    uint32_t wnext = ds[jg_dword_at(ridx, cws)];
    uint32_t p = p0, pb = p0, n = 0;                              // pb: position of the last block boundary
    int z = z0, k = 0;
    bool bad = false;
    int pred[3] = {pred0[0], pred0[1], pred0[2]};
    // EMIT: where the current block goes, and - computed while it is decoded, off the loop's critical path - the next one
    uint32_t mx = 0, my = 0, dst = 0, dstn = 0;
    int zn = z0;
    auto block_at = [&](int zz) {
        const uint4 sl = L.slot[zz];
        return sl.x + my * sl.y + mx * sl.z;
    };
    auto advance = [&]() {                                         // (zn, mx, my) -> the block after it
        if (++zn == bpm) {
            zn = 0;
            if (++mx == (uint32_t)L.fr.mcux) { mx = 0; ++my; }
        }
    };
    if constexpr (EMIT) {
        const uint32_t mcu = g0 / (uint32_t)bpm;
        my = mcu / (uint32_t)L.fr.mcux;
        mx = mcu - my * (uint32_t)L.fr.mcux;
        dst = block_at(zn);
        advance();
        dstn = block_at(zn);
    }
    // One symbol.  Returns false when the lane is done (block count / chunk end reached, or the data ended); a coefficient
    // to store comes back as (eo, ev) - eo = 0xffffffff: none.
    auto step = [&](uint32_t& eo, int& ev) -> bool {
        eo = 0xffffffffu;
        ev = 0;
        if (EMIT ? n >= want : (k == 0 && p >= pend)) return false;
        if (p >= nbits) return false;                              // data ended inside a block (or before the next one)
        if (have <= 32) {
            buf |= (uint64_t)__builtin_bswap32(wnext) << (32 - have);
            have += 32;
            ++ridx;
        }
        wnext = ds[jg_dword_at(ridx, cws)];
        const bool isdc = k == 0;
        const int tb = (int)(((isdc ? dcmap : acmap) >> (4 * z)) & 15u);
        const uint32_t code = (uint32_t)(buf >> 48);
        uint32_t e = L.tab.lut[tb][code >> (16 - JG_LUT_BITS)];
        if (e == 0) {                                              // a code of 13 .. 16 bits (rare), or none
            e = (16u << 8);
            bool found = false;
#pragma unroll
            for (int l = 13; l <= 16; ++l) {
                const int c = (int)(code >> (16 - l));
                if (!found && c <= L.tab.maxcode[tb][l - 13]) {
                    found = true;
                    e = ((uint32_t)l << 8) | L.tab.vals[tb][(L.tab.valptr[tb][l - 13] + c - L.tab.mincode[tb][l - 13]) & 255];
                }
            }
            bad |= !found;
        }
        const int len = (int)(e >> 8), rs = (int)(e & 255);
        int s = isdc ? rs : (rs & 15);
        const int r = isdc ? 0 : (rs >> 4);
        if (isdc && s > 11) { bad = true; s &= 7; }                // (an 8-bit file has no DC category above 11)
        const uint32_t v = s ? (uint32_t)((buf << len) >> (64 - s)) : 0u;
        const int used = len + s;
        buf <<= used;
        have -= used;
        p += (uint32_t)used;
        if (p > nbits) return false;                               // the symbol ran past the data: not a block
        if (isdc) {
            const int c = (int)((cmap >> (4 * z)) & 15u);
            const int d = jg_extend(v, s);
            dsum[c] += d;
            if constexpr (EMIT) {
                pred[c] += d;
                eo = dst;
                ev = pred[c];
            }
            k = 1;
        } else if (s == 0) {
            k = r == 15 ? k + 16 : 64;
        } else {
            k += r;
            if (k > 63) { bad = true; k = 64; }
            else {
                if constexpr (EMIT) {
                    eo = dst + L.zigzag[k];
                    ev = jg_extend(v, s);
                }
                ++k;
            }
        }
        if (k >= 64) {                                             // block complete
            k = 0;
            ++n;
            pb = p;
            if (++z == bpm) z = 0;
            if constexpr (EMIT) {
                dst = dstn;
                advance();
                dstn = block_at(zn);
            }
        }
        return true;
    };
    if constexpr (!EMIT) {
        uint32_t eo;
        int ev;
        while (step(eo, ev)) {}
    } else {
        // Stores count in vmcnt like loads and retire in order: with a 2-byte store in every step, the wait for the stream's
        // next dword at the top of the following step was a wait for that store's round trip (measured: the emit pass 3.5 x
        // a counting pass).  Sixteen steps keep their coefficients in registers and store them together: one such wait per
        // sixteen symbols.
        constexpr int G = 16;
        bool live = true;
        while (live) {
            uint32_t eo[G];
            int ev[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                eo[u] = 0xffffffffu;
                ev[u] = 0;
                if (live) live = step(eo[u], ev[u]);
            }
#pragma unroll
            for (int u = 0; u < G; ++u)
                if (eo[u] != 0xffffffffu) coef[eo[u]] = (int16_t)ev[u];
        }
    }
    *p_out = pb;
    *z_out = z;
    *n_out = n;
    *bad_out = bad;
}

__device__ __forceinline__ void jg_load_lds(JgLds& L, const JgFrame* F, const JgTableSet* T, int f) {
    const int tid = threadIdx.x;
    {
        const uint4* src = reinterpret_cast<const uint4*>(&T[F[f].tabset]);
        uint4* dst = reinterpret_cast<uint4*>(&L.tab);
        for (int i = tid; i < (int)(sizeof(JgTableSet) / 16); i += JG_CB) dst[i] = src[i];
    }
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&F[f]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&L.fr);
        for (int i = tid; i < (int)(sizeof(JgFrame) / 4); i += JG_CB) dst[i] = src[i];
    }
    if (tid < 64) L.zigzag[tid] = jg_zigzag[tid];
    __syncthreads();
    if (tid < 8) {
        const int z = tid < L.fr.bpm ? tid : 0, c = L.fr.slot_comp[z];
        L.slot[tid] = make_uint4(L.fr.coef_off + L.fr.comp_off[c] + ((uint32_t)L.fr.slot_by[z] * (uint32_t)L.fr.comp_bw[c] + L.fr.slot_bx[z]) * 64u,
                                 (uint32_t)L.fr.comp_v[c] * (uint32_t)L.fr.comp_bw[c] * 64u, (uint32_t)L.fr.comp_h[c] * 64u, 0u);
    }
    __syncthreads();
}

__global__ __launch_bounds__(JG_CB) void jg_round_kernel(const uint8_t* __restrict__ ds_base, const JgFrame* __restrict__ F,
                                                         const JgTableSet* __restrict__ T, const uint16_t* __restrict__ cblk_frame,
                                                         JgChunks S, int round) {
    __shared__ __attribute__((aligned(16))) JgLds L;
    const int f = cblk_frame[blockIdx.x];
    jg_load_lds(L, F, T, f);
    const uint32_t t = (blockIdx.x - L.fr.cblk0) * JG_CB + threadIdx.x, gt = L.fr.chunk0 + t;
    const uint32_t nbits = L.fr.nbits, cbits = (uint32_t)L.fr.chunk_bytes * 8u;
    const uint32_t b0 = t * cbits;
    if (t >= L.fr.nchunks || b0 >= nbits) return;                 // (chunks past the payload take no part)
    const uint32_t b1 = b0 + cbits < nbits ? b0 + cbits : nbits;
    uint2 start;
    if (t == 0) start = make_uint2(0u, 0u);
    else if (round == 0) start = make_uint2(b0, 0u);
    else start = S.en[(round - 1) & 1][gt - 1];
    if (round > 0) {
        const uint2 old = S.st[gt];
        if (old.x == start.x && old.y == start.y) { S.en[round & 1][gt] = S.en[(round - 1) & 1][gt]; return; }
    }
    S.st[gt] = start;
    {   // diagnostics: lanes that decode in this round (one atomic per wave)
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1u) atomicAdd(&S.redone[round], (uint32_t)__popcll(m));
    }
    uint32_t pe = start.x, n = 0;
    int ze = (int)start.y, dsum[3] = {0, 0, 0};
    bool bad = false;
    const int zero3[3] = {0, 0, 0};
    if (start.x < b1)
        jg_decode<false>(L, reinterpret_cast<const uint32_t*>(ds_base + L.fr.ds_off), nbits, start.x, (int)start.y, b1, 0u, &pe, &ze, &n, dsum,
                         &bad, nullptr, 0u, zero3);
    S.en[round & 1][gt] = make_uint2(pe, (uint32_t)ze);
    S.cnt[gt] = n;
    S.dcs[3 * gt] = dsum[0];
    S.dcs[3 * gt + 1] = dsum[1];
    S.dcs[3 * gt + 2] = dsum[2];
    S.err[gt] = bad ? 1 : 0;
}

// one 1024-thread block per frame: prefix sums over its chunks + the checks of the fixed point
__global__ __launch_bounds__(1024) void jg_scan_kernel(JgFrame* __restrict__ F, JgChunks S, int last_round) {
    __shared__ int wsum[16][4];
    __shared__ int carry[4];
    __shared__ int status;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const JgFrame fr = F[f];
    const uint32_t cbits = (uint32_t)fr.chunk_bytes * 8u;
    const uint32_t nvalid = fr.nbits ? (fr.nbits + cbits - 1) / cbits : 0u;
    if (tid < 4) carry[tid] = 0;
    if (tid == 0) status = JG_OK;
    __syncthreads();
    const uint2* en = S.en[last_round & 1];
    for (uint32_t t0 = 0; t0 < nvalid; t0 += 1024) {
        const uint32_t t = t0 + tid, gt = fr.chunk0 + t;
        const bool live = t < nvalid;
        int v[4] = {0, 0, 0, 0};
        if (live) {
            v[0] = (int)S.cnt[gt];
            v[1] = S.dcs[3 * gt];
            v[2] = S.dcs[3 * gt + 1];
            v[3] = S.dcs[3 * gt + 2];
        }
        int incl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            incl[c] = v[c];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl[c], off);
                if (lane >= off) incl[c] += o;
            }
            if (lane == 63) wsum[wave][c] = incl[c];
        }
        __syncthreads();
        int excl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int wb = carry[c];
            for (int w = 0; w < wave; ++w) wb += wsum[w][c];
            excl[c] = wb + incl[c] - v[c];
        }
        if (live) {
            S.gfirst[gt] = (uint32_t)excl[0];
            S.dcb[3 * gt] = excl[1];
            S.dcb[3 * gt + 1] = excl[2];
            S.dcb[3 * gt + 2] = excl[3];
            // the fixed point: this chunk started where its predecessor ended, in the slot its block index says
            const uint2 st = S.st[gt];
            const uint2 want = t == 0 ? make_uint2(0u, 0u) : en[gt - 1];
            int bad = JG_OK;
            if (st.x != want.x || st.y != want.y) bad = JG_NOT_CONVERGED;
            else if ((uint32_t)excl[0] % (uint32_t)fr.bpm != st.y) bad = JG_SLOT;
            else if (S.err[gt] && (uint32_t)excl[0] + (uint32_t)v[0] < (uint32_t)fr.total_blocks) bad = JG_BAD_CODE;   // (the padding bits after the last block decode to anything)
            if (bad) atomicMax(&status, bad);
        }
        __syncthreads();
        if (tid == 1023) {
#pragma unroll
            for (int c = 0; c < 4; ++c) carry[c] = excl[c] + v[c];
        }
        __syncthreads();
    }
    if (tid == 0) {
        int st = status;
        if (st == JG_OK && carry[0] != fr.total_blocks) st = JG_BLOCK_COUNT;
        F[f].status = st;
        F[f].blocks_found = (uint32_t)carry[0];
    }
}

__global__ __launch_bounds__(JG_CB) void jg_emit_kernel(const uint8_t* __restrict__ ds_base, const JgFrame* __restrict__ F,
                                                        const JgTableSet* __restrict__ T, const uint16_t* __restrict__ cblk_frame,
                                                        JgChunks S, int16_t* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) JgLds L;
    const int f = cblk_frame[blockIdx.x];
    jg_load_lds(L, F, T, f);
    if (L.fr.status != JG_OK) return;
    const uint32_t t = (blockIdx.x - L.fr.cblk0) * JG_CB + threadIdx.x, gt = L.fr.chunk0 + t;
    const uint32_t nbits = L.fr.nbits, cbits = (uint32_t)L.fr.chunk_bytes * 8u;
    if (t >= L.fr.nchunks || t * cbits >= nbits) return;
    const uint32_t want = S.cnt[gt];
    if (want == 0) return;
    const uint2 start = S.st[gt];
    const int pred0[3] = {S.dcb[3 * gt], S.dcb[3 * gt + 1], S.dcb[3 * gt + 2]};
    uint32_t pe, n;
    int ze, dsum[3] = {0, 0, 0};
    bool bad;
    jg_decode<true>(L, reinterpret_cast<const uint32_t*>(ds_base + L.fr.ds_off), nbits, start.x, (int)start.y, 0u, want, &pe, &ze, &n, dsum, &bad,
                    coef, S.gfirst[gt], pred0);
}

// ---------------------------------------------------------------------------------------------- host side
// device image of a parsed file's tables; false: more than JG_TABLES distinct tables
inline bool jg_build_tables(const Parsed& P, JgTableSet* ts, uint8_t (&slot_dc)[8], uint8_t (&slot_ac)[8], const ScanLayout& L) {
    const HuffTable* used[JG_TABLES];
    int nused = 0;
    auto index_of = [&](const HuffTable* t) {
        for (int i = 0; i < nused; ++i)
            if (used[i] == t) return i;
        if (nused == JG_TABLES) return -1;
        used[nused] = t;
        return nused++;
    };
    for (int s = 0; s < L.bpm; ++s) {
        const dfd_jpeg::Component& C = P.comp[L.slot_comp[s]];
        const int d = index_of(&P.dc[C.td]), a = index_of(&P.ac[C.ta]);
        if (d < 0 || a < 0) return false;
        slot_dc[s] = (uint8_t)d;
        slot_ac[s] = (uint8_t)a;
    }
    memset(ts, 0, sizeof *ts);
    for (int i = 0; i < nused; ++i) {
        const HuffTable& H = *used[i];
        memcpy(ts->vals[i], H.vals, 256);
        for (int l = 1; l <= 16; ++l) {
            if (H.maxcode[l] < 0) { if (l >= 13) { ts->maxcode[i][l - 13] = -1; } continue; }
            if (l >= 13) {
                ts->maxcode[i][l - 13] = H.maxcode[l];
                ts->mincode[i][l - 13] = H.mincode[l];
                ts->valptr[i][l - 13] = H.valptr[l];
                continue;
            }
            for (int code = H.mincode[l]; code <= H.maxcode[l]; ++code) {
                const int sym = H.vals[(H.valptr[l] + code - H.mincode[l]) & 255];
                const int base = code << (JG_LUT_BITS - l);
                for (int fill = 0; fill < (1 << (JG_LUT_BITS - l)); ++fill) ts->lut[i][base + fill] = (uint16_t)((l << 8) | sym);
            }
        }
        for (int l = 13; l <= 16; ++l)
            if (H.maxcode[l] < 0) ts->maxcode[i][l - 13] = -1;
    }
    return true;
}

// can this parsed file take the device path?
inline bool jg_supported(const Parsed& P, const ScanLayout& L) {
    return P.restart == 0 && L.bpm <= 8 && (size_t)(P.end - P.scan) < ((size_t)1 << 30);
}

}  // namespace dfd_jpeg_gpu
