// jpeg_host.h — structures and host-side parser of the JPEG decode row; no HIP types, builds for the CPU as well.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/vo_hip.h"

#ifndef JPG_NT
#define JPG_NT 256               // threads of the Huffman kernel = subsequences per image (512: 14 % less time for ONE launch of 257 files, 8 % / 3 % fewer pairs per second in the three-context pipeline on flat- / scene-chroma files)
#endif
#define JPG_LOOK 10              // bits of Huffman look-ahead table
#define JPG_LONG 4                // long-code prefixes per table with a second-level table
#define JPG_PAD 16               // zero bytes after every clean stream
#define JPG_MAX_BPM 10           // blocks per MCU (T.81 limit)
struct JpegImage {               // one file of a batch: geometry from the headers + where its data lives in the batch buffers
    uint32_t raw_off, raw_len, hdr_len;          // entropy-coded bytes in the blob (from the end of the SOS header to the end of the file)
    uint32_t clean_off, clean_len;               // the same without stuffing / restart markers (written by k_jpeg_unstuff)
    uint32_t rst_off, rst_cap, nrst;             // restart positions (clean-stream byte offsets)
    uint32_t sync_rounds;                        // diagnostic: propagation rounds k_jpeg_huffman needed
    uint32_t coef_blk;                           // first 8x8 block in the coefficient buffer
    uint32_t tab_idx;                            // its JpegTables in the batch's table array (files with identical headers share one)
    uint32_t out_stride;
    uint64_t plane_off[3], out_off;
    int32_t W, H, nc, mx, my, ri, ycc, bpm, total_blocks, mode /*0: 4:4:4 or grey, 1: h2v1, 2: h2v2*/, orientation;
    int32_t ch[3], cv[3], bw[3], bh[3], dw[3], dh[3], tq[3], td[3], ta[3];
    uint8_t blk_comp[JPG_MAX_BPM], blk_bx[JPG_MAX_BPM], blk_by[JPG_MAX_BPM];
};
struct JpegTables {
    uint16_t q[4][64];                           // quantisation tables, natural order
    // slots 0-3: DC tables, 4-7: AC tables, indexed by the next JPG_LOOK bits of the stream: (code length << 8) | symbol for the
    // codes that fit; 0x8000 | n for a prefix of longer codes that has second-level table n; 0 = longer code without one
    uint16_t lut[8][1 << JPG_LOOK];
    // codes longer than the look-ahead: the first JPG_LONG prefixes (of JPG_LOOK bits) that lead to such codes get a table over the
    // next 6 bits ((code length << 8) | symbol, 0 = no such code).  A code book with more long prefixes than that leaves the
    // rest at 0 (the canonical walk over the lengths, maxcode / valoff)
    uint16_t sub[8][JPG_LONG][64];
    int32_t maxcode[8][18], valoff[8][18];
    uint8_t vals[8][256];
};
int jpeg_info(const uint8_t* d, size_t n, int* h, int* w, int* ncomp, int* sampling, int* orientation);
int jpeg_parse(const uint8_t* d, size_t n, JpegImage* img, JpegTables* T, const char** why, const uint8_t* prev_hdr = nullptr,
               size_t prev_hdr_len = 0, const JpegImage* prev_img = nullptr, const JpegTables* prev_T = nullptr, bool* same_tables = nullptr);
