// jpeg_host.cpp — the host half of the JPEG decode row: marker segments SOI .. SOS of cv2.imread's input
// (/root/reference/src/visual_slam.py:346) parsed into JpegImage / JpegTables.  Plain C++ with no HIP in it, so that the
// same unit also builds for the CPU under AddressSanitizer (tests/test_jpeg_host_sanitize.py feeds it damaged files).
#include "jpeg_host.h"
#include <string.h>

static const uint8_t h_zigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------ host: marker segments up to the scan header
static inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

static int exif_orientation_of(const uint8_t* s, int len)
{
    if (len < 14 || memcmp(s, "Exif\0\0", 6) != 0) return 0;
    const uint8_t* t = s + 6; const uint32_t n = (uint32_t)(len - 6);
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 0;
    auto u16 = [&](uint32_t o) -> uint32_t { return le ? (uint32_t)(t[o] | (t[o + 1] << 8)) : (uint32_t)((t[o] << 8) | t[o + 1]); };
    auto u32 = [&](uint32_t o) -> uint32_t { return le ? (u16(o) | (u16(o + 2) << 16)) : ((u16(o) << 16) | u16(o + 2)); };
    const uint32_t ifd = u32(4);
    if (ifd > n || ifd + 2 > n) return 0;
    const uint32_t cnt = u16(ifd);
    for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t e = ifd + 2 + 12 * i;
        if (e + 12 > n) return 0;
        if (u16(e) == 0x0112) { const uint32_t v = u16(e + 8); return v >= 1 && v <= 8 ? (int)v : 0; }
    }
    return 0;
}

int jpeg_info(const uint8_t* d, size_t n, int* h, int* w, int* ncomp, int* sampling, int* orientation)
{
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return VO_ERR_INVALID;
    size_t pos = 2;
    int orient = 0;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) return VO_ERR_INVALID;
        while (pos < n && d[pos] == 0xFF) pos++;
        if (pos >= n) break;
        const int m = d[pos++];
        if (m == 0xD9 || m == 0xDA || pos + 2 > n) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        const int len = rd16(d + pos);
        if (len < 2 || pos + (size_t)len > n) return VO_ERR_INVALID;
        if (m == 0xE1 && !orient) orient = exif_orientation_of(d + pos + 2, len - 2);
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            if (len < 8) return VO_ERR_INVALID;
            if (h) *h = rd16(d + pos + 3);
            if (w) *w = rd16(d + pos + 5);
            if (ncomp) *ncomp = d[pos + 7];
            if (sampling) *sampling = len >= 11 ? d[pos + 9] : 0;
            if (orientation) *orientation = orient;
            return (m == 0xC0 || m == 0xC1) && d[pos + 2] == 8 ? VO_OK : VO_ERR_UNSUPPORTED;
        }
        pos += (size_t)len;
    }
    return VO_ERR_INVALID;
}

// canonical code book -> 9-bit look-ahead table + the (maxcode, value offset) pairs of the longer codes
static bool build_tables(const uint8_t* counts /*16*/, const uint8_t* vals, int nvals, JpegTables* T, int slot)
{
    uint16_t* lut = T->lut[slot];
    memset(lut, 0, sizeof(T->lut[slot]));
    memset(T->vals[slot], 0, 256);
    memcpy(T->vals[slot], vals, (size_t)nvals);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        const int cnt = counts[l - 1];
        T->valoff[slot][l] = k - code;
        if (code + cnt > (1 << l)) return false;      // over-subscribed code book: reject BEFORE anything is indexed by the codes
        if (cnt) {
            if (l <= JPG_LOOK)
                for (int i = 0; i < cnt; i++) {
                    const int c = (code + i) << (JPG_LOOK - l);
                    for (int j = 0; j < (1 << (JPG_LOOK - l)); j++) lut[c + j] = (uint16_t)((l << 8) | vals[k + i]);
                }
            code += cnt; k += cnt;
            T->maxcode[slot][l] = code - 1;
        } else T->maxcode[slot][l] = -1;
        code <<= 1;
    }
    // second level for the codes of 11 .. 16 bits: walk them in canonical order, one 64-entry table per 10-bit prefix
    memset(T->sub[slot], 0, sizeof(T->sub[slot]));
    {
        int c2 = 0, k2 = 0, nsub = 0, last_prefix = -1;
        for (int l = 1; l <= 16; l++) {
            const int cnt = counts[l - 1];
            for (int i = 0; i < cnt; i++, c2++, k2++) {
                if (l <= JPG_LOOK) continue;
                const int prefix = c2 >> (l - JPG_LOOK);
                if (prefix != last_prefix) { last_prefix = prefix; nsub++; if (nsub <= JPG_LONG) lut[prefix] = (uint16_t)(0x8000u | (uint32_t)(nsub - 1)); }
                if (nsub > JPG_LONG) continue;
                const int low = (c2 << (16 - l)) & 63;                     // the code's bits 10 .. 15, left-aligned in 6 bits
                for (int j = 0; j < (1 << (16 - l)); j++) T->sub[slot][nsub - 1][low + j] = (uint16_t)((l << 8) | vals[k2]);
            }
            c2 <<= 1;
        }
    }
    T->maxcode[slot][17] = 0x7fffffff; T->valoff[slot][17] = 0;
    T->maxcode[slot][0] = -1; T->valoff[slot][0] = 0;
    return k == nvals;
}

// Fills img (geometry, table selectors, where the entropy-coded bytes start) and T; the caller assigns buffer offsets.
// Frames of one camera / encoder carry identical DQT and DHT segments: when `prev` (the header bytes and tables of the
// file parsed before) starts with the same bytes up to the scan header, its tables are copied instead of rebuilt.
int jpeg_parse(const uint8_t* d, size_t n, JpegImage* img, JpegTables* T, const char** why, const uint8_t* prev_hdr, size_t prev_hdr_len,
               const JpegImage* prev_img, const JpegTables* prev_T, bool* same_tables)
{
    if (same_tables) *same_tables = false;
    if (prev_hdr && prev_hdr_len > 4 && prev_hdr_len <= n && !memcmp(d, prev_hdr, prev_hdr_len)) {
        static const char* dummy2; if (!why) why = &dummy2;
        *img = *prev_img;
        if (same_tables) *same_tables = true;                 // the caller lets this file share the previous file's tables
        else memcpy(T, prev_T, sizeof(*T));
        if (n - prev_hdr_len > 0x7fffffffull) { *why = "image too large"; return VO_ERR_UNSUPPORTED; }
        img->raw_len = (uint32_t)(n - prev_hdr_len);
        return VO_OK;
    }
    static const char* dummy; if (!why) why = &dummy;
    memset(img, 0, sizeof(*img)); memset(T, 0, sizeof(*T));
    bool qseen[4] = {false, false, false, false}, hseen[8] = {false, false, false, false, false, false, false, false};
    bool sof = false, jfif = false, adobe = false;
    int adobe_tr = 0, cid[3] = {0, 0, 0};
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { *why = "no SOI marker"; return VO_ERR_INVALID; }
    size_t pos = 2;
    for (;;) {
        if (pos + 4 > n || d[pos] != 0xFF) { *why = "broken marker structure"; return VO_ERR_INVALID; }
        while (pos < n && d[pos] == 0xFF) pos++;
        if (pos >= n) { *why = "truncated file"; return VO_ERR_INVALID; }
        const int m = d[pos++];
        if (m == 0xD9) { *why = "no scan before EOI"; return VO_ERR_INVALID; }
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) { *why = "truncated file"; return VO_ERR_INVALID; }
        const int len = rd16(d + pos);
        if (len < 2 || pos + (size_t)len > n) { *why = "segment length past the end of the file"; return VO_ERR_INVALID; }
        const uint8_t* s = d + pos + 2; const int sl = len - 2;
        pos += (size_t)len;
        if (m == 0xDB) {
            for (int o = 0; o < sl;) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3 || pq > 1 || o + 64 * (pq + 1) > sl) { *why = "bad DQT"; return VO_ERR_INVALID; }
                for (int i = 0; i < 64; i++) T->q[tq][h_zigzag[i]] = (uint16_t)(pq ? rd16(s + o + 2 * i) : s[o + i]);
                o += 64 * (pq + 1); qseen[tq] = true;
            }
        } else if (m == 0xC4) {
            for (int o = 0; o < sl;) {
                if (o + 17 > sl) { *why = "bad DHT"; return VO_ERR_INVALID; }
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                int cnt = 0;
                for (int l = 0; l < 16; l++) cnt += s[o + l];
                if (tc > 1 || th > 3 || cnt > 256 || o + 16 + cnt > sl || !build_tables(s + o, s + o + 16, cnt, T, tc * 4 + th)) { *why = "bad DHT"; return VO_ERR_INVALID; }
                hseen[tc * 4 + th] = true;
                o += 16 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sof || sl < 6) { *why = "bad SOF"; return VO_ERR_INVALID; }
            if (s[0] != 8) { *why = "12-bit samples"; return VO_ERR_UNSUPPORTED; }
            img->H = rd16(s + 1); img->W = rd16(s + 3); img->nc = s[5];
            if (img->H == 0 || img->W == 0) { *why = "height defined by a DNL marker"; return VO_ERR_UNSUPPORTED; }
            if (img->nc != 1 && img->nc != 3) { *why = "neither grey nor three components (CMYK?)"; return VO_ERR_UNSUPPORTED; }
            if (sl < 6 + 3 * img->nc) { *why = "bad SOF"; return VO_ERR_INVALID; }
            for (int i = 0; i < img->nc; i++) {
                cid[i] = s[6 + 3 * i]; img->ch[i] = s[7 + 3 * i] >> 4; img->cv[i] = s[7 + 3 * i] & 15; img->tq[i] = s[8 + 3 * i];
                if (img->ch[i] < 1 || img->ch[i] > 4 || img->cv[i] < 1 || img->cv[i] > 4 || img->tq[i] > 3) { *why = "bad SOF"; return VO_ERR_INVALID; }
            }
            sof = true;
        } else if (m >= 0xC2 && m <= 0xCF) {
            *why = "progressive, lossless or arithmetic-coded frame"; return VO_ERR_UNSUPPORTED;
        } else if (m == 0xDD) {
            if (sl < 2) { *why = "bad DRI"; return VO_ERR_INVALID; }
            img->ri = rd16(s);
        } else if (m == 0xE0) {
            if (sl >= 5 && !memcmp(s, "JFIF\0", 5)) jfif = true;
        } else if (m == 0xE1) {
            if (!img->orientation) img->orientation = exif_orientation_of(s, sl);
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_tr = s[11]; }
        } else if (m == 0xDA) {
            if (!sof || sl < 1) { *why = "scan before frame header"; return VO_ERR_INVALID; }
            if (s[0] != img->nc) { *why = "components spread over several scans"; return VO_ERR_UNSUPPORTED; }
            if (sl < 1 + 2 * img->nc + 3) { *why = "bad SOS"; return VO_ERR_INVALID; }
            for (int i = 0; i < img->nc; i++) {
                if (s[1 + 2 * i] != cid[i]) { *why = "scan component order differs from the frame's"; return VO_ERR_UNSUPPORTED; }
                img->td[i] = s[2 + 2 * i] >> 4; img->ta[i] = s[2 + 2 * i] & 15;
                if (img->td[i] > 3 || img->ta[i] > 3 || !hseen[img->td[i]] || !hseen[4 + img->ta[i]] || !qseen[img->tq[i]]) { *why = "scan refers to a missing table"; return VO_ERR_INVALID; }
            }
            break;
        }
    }
    if (img->nc == 1) img->ch[0] = img->cv[0] = 1;                 // a single-component scan is never interleaved
    int hmax = 1, vmax = 1;
    for (int i = 0; i < img->nc; i++) { hmax = img->ch[i] > hmax ? img->ch[i] : hmax; vmax = img->cv[i] > vmax ? img->cv[i] : vmax; }
    img->mode = 0;
    if (img->nc == 3) {
        if (img->ch[0] != hmax || img->cv[0] != vmax || img->ch[1] != img->ch[2] || img->cv[1] != img->cv[2]) { *why = "sampling factors outside 4:4:4 / 4:2:2 / 4:2:0"; return VO_ERR_UNSUPPORTED; }
        if (img->ch[1] == hmax && img->cv[1] == vmax) img->mode = 0;
        else if (img->ch[1] * 2 == hmax && img->cv[1] == vmax) img->mode = 1;
        else if (img->ch[1] * 2 == hmax && img->cv[1] * 2 == vmax) img->mode = 2;
        else { *why = "sampling factors outside 4:4:4 / 4:2:2 / 4:2:0"; return VO_ERR_UNSUPPORTED; }
        img->ycc = jfif ? 1 : adobe ? (adobe_tr != 0) : !(cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B');
    }
    img->mx = (img->W + 8 * hmax - 1) / (8 * hmax); img->my = (img->H + 8 * vmax - 1) / (8 * vmax);
    img->bpm = 0;
    for (int i = 0; i < img->nc; i++) {
        img->bw[i] = img->mx * img->ch[i]; img->bh[i] = img->my * img->cv[i];
        img->dw[i] = (img->W * img->ch[i] + hmax - 1) / hmax; img->dh[i] = (img->H * img->cv[i] + vmax - 1) / vmax;
        for (int by = 0; by < img->cv[i]; by++)
            for (int bx = 0; bx < img->ch[i]; bx++) {
                if (img->bpm >= JPG_MAX_BPM) { *why = "more than 10 blocks per MCU"; return VO_ERR_INVALID; }
                img->blk_comp[img->bpm] = (uint8_t)i; img->blk_bx[img->bpm] = (uint8_t)bx; img->blk_by[img->bpm] = (uint8_t)by; img->bpm++;
            }
    }
    const long long blocks = (long long)img->mx * img->my * img->bpm;
    if (blocks > 0x3fffffff || n - pos > 0x7fffffffull) { *why = "image too large"; return VO_ERR_UNSUPPORTED; }
    img->total_blocks = (int32_t)blocks;
    img->raw_len = (uint32_t)(n - pos);
    img->hdr_len = (uint32_t)pos;
    return VO_OK;
}

