// Launchers and buffer layout of the forensic-signal kernels (256x256 analysis image).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "imgproc_kernels.h"

namespace dfd {

// per-frame scalar statistics produced on the device (doubles)
enum ForensicStat {
    ST_FREQ_LOW = 0, ST_FREQ_MID, ST_FREQ_HIGH, ST_FREQ_MID_STD, ST_LAP_VAR, ST_EDGE_COUNT,
    ST_SAT_STD, ST_VAL_STD, ST_HUES, FORENSIC_STATS
};

struct ForensicBuffers {
    uint8_t* rs;         // [n][256][256][3] resized BGR
    uint8_t* gray;       // [n][65536]
    float2* fft_tmp;     // [n][65536] row-FFT output, transposed
    double* fft_part;    // [n][256][7]
    short2* grad;        // [n][65536] Sobel dx,dy
    double* lap_part;    // [n][256][2]
    uint8_t* map;        // [n][65536] Canny labels
    double* edge_count;  // [n]
    uint8_t *jy, *jcb, *jcr;   // decoded JPEG planes [n][65536], [n][16384] x2
    double* hsv_part;    // [n][256][4]
    unsigned* hue_bits;  // [n][6]
    double* stats;       // [n][FORENSIC_STATS]
    double* stats_noise; // [n][64] block stds of the noise residual
    double* stats_ela;   // [n][64] block means of the ELA difference
};

size_t forensic_bytes_per_frame();
void forensic_carve(void* base, int n, ForensicBuffers* out);
void launch_forensics(const ForensicBuffers& B, int n, bool full, const ColorTables& T, const float2* tw, hipStream_t s,
                      int gray_only = 0);
void launch_absdiff(const uint8_t* gray, const uint8_t* prev, double* part256, hipStream_t s);
void launch_absdiff_pairs(const uint8_t* gray, const int* prev_index, double* part /*[n][256]*/, int n, hipStream_t s);

}  // namespace dfd
