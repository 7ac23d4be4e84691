/* voo_jpeg.c — CPU ORACLE (test infrastructure only, see voo.h) for the JPEG decode in front of the path:
 * cv2.imread(filename) at /root/reference/src/visual_slam.py:346 (also src/triangulate_points_from_images.py:14-15,
 * src/feature_detection.py:5,10).  cv2.imread hands a .jpg to libjpeg-turbo with its default decompression parameters:
 * baseline / extended-sequential Huffman, 8-bit; dct_method = JDCT_ISLOW (jidctint.c, 13-bit constants, two passes);
 * do_fancy_upsampling = TRUE (jdsample.c h2v1 / h2v2 triangle filters, plain replication when the chroma plane is at
 * most 2 samples wide); YCbCr -> RGB through the 16-bit fixed-point tables of jdcolor.c; output order B, G, R;
 * a grey-scale file is replicated into three channels (IMREAD_COLOR is imread's default).
 *
 * PINNED (unlike the rest of the oracle): Pillow is importable in this image and wraps the same library
 * (libjpeg-turbo, same defaults), so tests/test_oracle_jpeg.py compares this file byte for byte with
 * PIL.Image.open(...).convert("RGB") on JPEGs of every supported layout.
 *
 * Supported: SOF0 / SOF1, 1 or 3 components in ONE interleaved scan, sampling 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2), restart
 * intervals, 8- or 16-bit quantisation tables, JFIF / Adobe colour-space rules.  Not supported (VOO_JPEG_UNSUPPORTED):
 * progressive / arithmetic / lossless frames, 12-bit samples, CMYK, multi-scan sequential files, other sampling ratios.
 * The IDCT works in 32-bit wrap-around integers (libjpeg's C code uses `long`, its SIMD code 16/32-bit lanes: they agree
 * with this on every stream an encoder can produce).  EXIF orientation is reported, not applied. */
#include "voo.h"
#include <stdlib.h>
#include <string.h>

static const uint8_t ZIGZAG[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    int present;
    uint8_t nbits[17];          /* number of codes of each length 1..16 */
    uint8_t vals[256];
    int32_t mincode[17], maxcode[18], valptr[17];   /* JPEG spec F.2.2.3 */
} huff_t;

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;                 /* plane size in blocks (padded to whole MCUs) */
    int dw, dh;                 /* downsampled_width / _height: the real samples of the component */
    uint8_t* plane;             /* bw*8 x bh*8 samples */
    int pred;
} comp_t;

typedef struct {
    const uint8_t* p; size_t n, pos;
    uint32_t acc; int cnt;      /* bit accumulator */
    int marker;                 /* a marker was met inside the entropy-coded segment (0 = none) */
    int insufficient;           /* bits were requested after the data ran out (jdhuff.c insufficient_data) */
} bits_t;

static int next_bit(bits_t* b)
{
    if (b->cnt == 0) {
        int byte = 0;
        if (!b->marker && b->pos < b->n) {
            byte = b->p[b->pos];
            if (byte == 0xFF) {
                const int nx = b->pos + 1 < b->n ? b->p[b->pos + 1] : 0xD9;
                if (nx == 0) b->pos += 2;                    /* stuffed zero */
                else { b->marker = nx; byte = 0; }           /* libjpeg: feed zero bits once a marker is reached */
            } else b->pos++;
        } else if (!b->marker) b->marker = 0xD9;             /* the file simply ends */
        if (b->marker) b->insufficient = 1;
        b->acc = (uint32_t)byte; b->cnt = 8;
    }
    b->cnt--;
    return (int)((b->acc >> b->cnt) & 1u);
}

static int receive(bits_t* b, int s) { int v = 0; while (s--) v = (v << 1) | next_bit(b); return v; }
static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

static int decode_sym(bits_t* b, const huff_t* h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | next_bit(b);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->vals[h->valptr[l] + code - h->mincode[l]];
    }
    /* garbage (damaged or zero-filled data): jdhuff.c jpeg_huff_decode runs on to its sentinel length 17 — one more bit is
     * consumed — and "fakes a zero as the safest result" */
    (void)next_bit(b);
    return 0;
}

static int build_huff(huff_t* h)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k; h->mincode[l] = code;
        if (h->nbits[l]) { code += h->nbits[l]; k += h->nbits[l]; h->maxcode[l] = code - 1; }
        else h->maxcode[l] = -1;
        if (code > (1 << l)) return -1;
        code <<= 1;
    }
    return k <= 256 ? 0 : -1;
}

/* jidctint.c: jpeg_idct_islow */
#define CONST_BITS 13
#define PASS1_BITS 2
#define MUL(a, c) ((int32_t)((uint32_t)(a) * (uint32_t)(c)))
#define ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define SUB(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))
#define SHL(a, n) ((int32_t)((uint32_t)(a) << (n)))
static int32_t descale(int32_t x, int n) { return ADD(x, 1 << (n - 1)) >> n; }
static uint8_t idct_limit(int32_t v)
{
    /* jidctint.c looks the sample up in IDCT_range_limit[v & RANGE_MASK] — a 10-bit WRAP: 512 <= v < 896 comes out as 0, not
     * 255.  The libjpeg-turbo that cv2 (and Pillow) ship runs its SIMD transform (jidctint-sse2 / -avx2 / -neon), which narrows
     * with SATURATION (packssdw, packsswb) and then adds 128: an overshoot stays at the end of the range.  The two differ only
     * for samples more than four times out of range (quality 1 on saturated noise: one pixel in 53 000 random files,
     * tests/golden/jpeg_gray_q1_saturated_329x267.jpg); the SIMD behaviour is what cv2.imread returns. */
    const int r = v + 128;
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

static void idct_1d(const int32_t in[8], int32_t out[8], int shift, int pass1)
{
    int32_t z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    z2 = in[2]; z3 = in[6];
    z1 = MUL(ADD(z2, z3), 4433);
    tmp2 = ADD(z1, MUL(z3, -15137));
    tmp3 = ADD(z1, MUL(z2, 6270));
    z2 = in[0]; z3 = in[4];
    tmp0 = SHL(ADD(z2, z3), CONST_BITS); tmp1 = SHL(SUB(z2, z3), CONST_BITS);
    tmp10 = ADD(tmp0, tmp3); tmp13 = SUB(tmp0, tmp3); tmp11 = ADD(tmp1, tmp2); tmp12 = SUB(tmp1, tmp2);
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = ADD(tmp0, tmp3); z2 = ADD(tmp1, tmp2); z3 = ADD(tmp0, tmp2); z4 = ADD(tmp1, tmp3);
    z5 = MUL(ADD(z3, z4), 9633);
    tmp0 = MUL(tmp0, 2446); tmp1 = MUL(tmp1, 16819); tmp2 = MUL(tmp2, 25172); tmp3 = MUL(tmp3, 12299);
    z1 = MUL(z1, -7373); z2 = MUL(z2, -20995); z3 = MUL(z3, -16069); z4 = MUL(z4, -3196);
    z3 = ADD(z3, z5); z4 = ADD(z4, z5);
    tmp0 = ADD(tmp0, ADD(z1, z3)); tmp1 = ADD(tmp1, ADD(z2, z4)); tmp2 = ADD(tmp2, ADD(z2, z3)); tmp3 = ADD(tmp3, ADD(z1, z4));
    (void)pass1;
    out[0] = descale(ADD(tmp10, tmp3), shift); out[7] = descale(SUB(tmp10, tmp3), shift);
    out[1] = descale(ADD(tmp11, tmp2), shift); out[6] = descale(SUB(tmp11, tmp2), shift);
    out[2] = descale(ADD(tmp12, tmp1), shift); out[5] = descale(SUB(tmp12, tmp1), shift);
    out[3] = descale(ADD(tmp13, tmp0), shift); out[4] = descale(SUB(tmp13, tmp0), shift);
}

/* coef: 64 quantised coefficients in natural order; q: quantisation table in natural order */
static void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride)
{
    int32_t ws[64];
    for (int c = 0; c < 8; c++) {
        int32_t in[8], o[8];
        for (int r = 0; r < 8; r++) in[r] = MUL((int32_t)coef[8 * r + c], (int32_t)q[8 * r + c]);
        /* (libjpeg's all-AC-zero shortcut gives the same values as the full computation) */
        idct_1d(in, o, CONST_BITS - PASS1_BITS, 1);
        for (int r = 0; r < 8; r++) ws[8 * r + c] = o[r];
    }
    for (int r = 0; r < 8; r++) {
        int32_t o[8];
        idct_1d(ws + 8 * r, o, CONST_BITS + PASS1_BITS + 3, 0);
        for (int c = 0; c < 8; c++) out[r * stride + c] = idct_limit(o[c]);
    }
}

static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* jdsample.c h2v1_fancy_upsample: one row of dw samples -> 2*dw samples */
static void up_h2v1(const uint8_t* in, int dw, uint8_t* out)
{
    if (dw <= 2) { for (int i = 0; i < dw; i++) out[2 * i] = out[2 * i + 1] = in[i]; return; }     /* h2v1_upsample */
    out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < dw - 1; i++) {
        out[2 * i] = (uint8_t)((in[i] * 3 + in[i - 1] + 1) >> 2);
        out[2 * i + 1] = (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
    }
    out[2 * dw - 2] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2); out[2 * dw - 1] = in[dw - 1];
}

/* jdsample.c h2v2_fancy_upsample: output row from the nearer input row in0 and the farther one in1 */
static void up_h2v2(const uint8_t* in0, const uint8_t* in1, int dw, uint8_t* out)
{
    int last, cur = in0[0] * 3 + in1[0], next = in0[1] * 3 + in1[1];
    out[0] = (uint8_t)((cur * 4 + 8) >> 4); out[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
    last = cur; cur = next;
    for (int i = 1; i < dw - 1; i++) {
        next = in0[i + 1] * 3 + in1[i + 1];
        out[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4); out[2 * i + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
        last = cur; cur = next;
    }
    out[2 * dw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); out[2 * dw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
}

static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

/* EXIF orientation (1..8) from an APP1 segment body, 0 if absent */
static int exif_orientation(const uint8_t* s, int len)
{
    if (len < 14 || memcmp(s, "Exif\0\0", 6)) return 0;
    const uint8_t* t = s + 6; const int n = len - 6;
    const int le = t[0] == 'I';
    if (!((t[0] == 'I' && t[1] == 'I') || (t[0] == 'M' && t[1] == 'M'))) return 0;
#define RD16(o) (le ? (t[o] | (t[(o) + 1] << 8)) : ((t[o] << 8) | t[(o) + 1]))
#define RD32(o) (le ? ((uint32_t)t[o] | ((uint32_t)t[(o) + 1] << 8) | ((uint32_t)t[(o) + 2] << 16) | ((uint32_t)t[(o) + 3] << 24)) \
                    : (((uint32_t)t[o] << 24) | ((uint32_t)t[(o) + 1] << 16) | ((uint32_t)t[(o) + 2] << 8) | (uint32_t)t[(o) + 3]))
    const uint32_t ifd = RD32(4);
    if (ifd + 2 > (uint32_t)n) return 0;
    const int cnt = RD16(ifd);
    for (int i = 0; i < cnt; i++) {
        const uint32_t e = ifd + 2 + 12u * (uint32_t)i;
        if (e + 12 > (uint32_t)n) return 0;
        if (RD16(e) == 0x0112) { const int v = RD16(e + 8); return v >= 1 && v <= 8 ? v : 0; }
    }
#undef RD16
#undef RD32
    return 0;
}

/* Header only: size, component count, sampling of component 0 (h << 4 | v), EXIF orientation (0 = none).
 * Returns VOO_OK, VOO_JPEG_UNSUPPORTED or VOO_JPEG_CORRUPT. */
int voo_jpeg_info(const uint8_t* data, size_t n, int32_t* h, int32_t* w, int32_t* ncomp, int32_t* sampling, int32_t* orientation)
{
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return VOO_JPEG_CORRUPT;
    size_t pos = 2;
    int orient = 0;
    while (pos + 4 <= n) {
        if (data[pos] != 0xFF) return VOO_JPEG_CORRUPT;
        while (pos < n && data[pos] == 0xFF) pos++;
        if (pos >= n) break;
        const int m = data[pos++];
        if (m == 0xD9 || m == 0xDA) break;
        if (pos + 2 > n) break;
        const int len = be16(data + pos);
        if (len < 2 || pos + (size_t)len > n) return VOO_JPEG_CORRUPT;
        if (m == 0xE1 && !orient) orient = exif_orientation(data + pos + 2, len - 2);
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            if (len < 8) return VOO_JPEG_CORRUPT;
            if (h) *h = be16(data + pos + 3);
            if (w) *w = be16(data + pos + 5);
            if (ncomp) *ncomp = data[pos + 7];
            if (sampling) *sampling = len >= 11 ? data[pos + 9] : 0;
            if (orientation) *orientation = orient;
            return (m == 0xC0 || m == 0xC1) && data[pos + 2] == 8 ? VOO_OK : VOO_JPEG_UNSUPPORTED;
        }
        pos += (size_t)len;
    }
    return VOO_JPEG_CORRUPT;
}

/* cv2.imdecode(buf, IMREAD_COLOR) for a JPEG: out = h x w x 3, B G R, rows of out_stride bytes. */
int voo_jpeg_decode(const uint8_t* data, size_t n, uint8_t* out, int out_stride, int cap_h, int cap_w)
{
    uint16_t Q[4][64];
    int qpresent[4] = {0, 0, 0, 0};
    huff_t* H = (huff_t*)calloc(8, sizeof(huff_t));          /* [0..3] DC, [4..7] AC */
    comp_t C[3];
    int W = 0, Hh = 0, nc = 0, ri = 0, got_sof = 0, jfif = 0, adobe = 0, adobe_tr = 0, rc = VOO_JPEG_CORRUPT;
    memset(C, 0, sizeof(C)); memset(Q, 0, sizeof(Q));
    if (!H) return VOO_JPEG_CORRUPT;
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) goto done;
    size_t pos = 2;
    for (;;) {
        if (pos + 4 > n || data[pos] != 0xFF) goto done;
        while (pos < n && data[pos] == 0xFF) pos++;
        if (pos >= n) goto done;
        const int m = data[pos++];
        if (m == 0xD9) goto done;                                            /* EOI before any scan */
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;                 /* parameterless */
        if (pos + 2 > n) goto done;
        const int len = be16(data + pos);
        if (len < 2 || pos + (size_t)len > n) goto done;
        const uint8_t* s = data + pos + 2; const int sl = len - 2;
        pos += (size_t)len;
        if (m == 0xDB) {                                                     /* DQT */
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15; o++;
                if (tq > 3 || pq > 1 || o + 64 * (pq + 1) > sl) goto done;
                for (int i = 0; i < 64; i++) { Q[tq][ZIGZAG[i]] = (uint16_t)(pq ? be16(s + o + 2 * i) : s[o + i]); }
                o += 64 * (pq + 1); qpresent[tq] = 1;
            }
        } else if (m == 0xC4) {                                              /* DHT */
            int o = 0;
            while (o < sl) {
                if (o + 17 > sl) goto done;
                const int tc = s[o] >> 4, th = s[o] & 15; o++;
                if (tc > 1 || th > 3) goto done;
                huff_t* h = &H[tc * 4 + th];
                int cnt = 0;
                memset(h, 0, sizeof(*h));
                for (int l = 1; l <= 16; l++) { h->nbits[l] = s[o + l - 1]; cnt += s[o + l - 1]; }
                o += 16;
                if (cnt > 256 || o + cnt > sl) goto done;
                memcpy(h->vals, s + o, (size_t)cnt); o += cnt;
                if (build_huff(h)) goto done;
                h->present = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {                                 /* SOF0 / SOF1 */
            if (got_sof || sl < 6) goto done;
            if (s[0] != 8) { rc = VOO_JPEG_UNSUPPORTED; goto done; }
            Hh = be16(s + 1); W = be16(s + 3); nc = s[5];
            if (Hh == 0 || W == 0) { rc = VOO_JPEG_UNSUPPORTED; goto done; }  /* DNL-defined height */
            if (nc != 1 && nc != 3) { rc = VOO_JPEG_UNSUPPORTED; goto done; }
            if (sl < 6 + 3 * nc) goto done;
            for (int i = 0; i < nc; i++) {
                C[i].id = s[6 + 3 * i]; C[i].h = s[7 + 3 * i] >> 4; C[i].v = s[7 + 3 * i] & 15; C[i].tq = s[8 + 3 * i];
                if (C[i].h < 1 || C[i].h > 4 || C[i].v < 1 || C[i].v > 4 || C[i].tq > 3) goto done;
            }
            got_sof = 1;
        } else if (m >= 0xC2 && m <= 0xCF) {                                 /* progressive, lossless, arithmetic ... */
            rc = VOO_JPEG_UNSUPPORTED; goto done;
        } else if (m == 0xDD) {
            if (sl < 2) goto done;
            ri = be16(s);
        } else if (m == 0xE0) {
            if (sl >= 5 && !memcmp(s, "JFIF\0", 5)) jfif = 1;
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) { adobe = 1; adobe_tr = s[11]; }
        } else if (m == 0xDA) {                                              /* SOS */
            if (!got_sof || sl < 1) goto done;
            const int ns = s[0];
            if (ns != nc) { rc = VOO_JPEG_UNSUPPORTED; goto done; }          /* multi-scan sequential file */
            if (sl < 1 + 2 * ns + 3) goto done;
            for (int i = 0; i < ns; i++) {
                if (s[1 + 2 * i] != C[i].id) { rc = VOO_JPEG_UNSUPPORTED; goto done; }
                C[i].td = s[2 + 2 * i] >> 4; C[i].ta = s[2 + 2 * i] & 15;
                if (C[i].td > 3 || C[i].ta > 3 || !H[C[i].td].present || !H[4 + C[i].ta].present || !qpresent[C[i].tq]) goto done;
            }
            break;
        }
        /* every other segment (APPn, COM, ...) is skipped */
    }
    if (Hh > cap_h || W > cap_w) { rc = VOO_JPEG_TOO_SMALL; goto done; }
    {
        int hmax = 1, vmax = 1;
        if (nc == 1) { C[0].h = C[0].v = 1; }                                 /* a single-component scan is never interleaved */
        for (int i = 0; i < nc; i++) { if (C[i].h > hmax) hmax = C[i].h; if (C[i].v > vmax) vmax = C[i].v; }
        if (nc == 3) {
            /* the upsamplers built: full size, h2v1, h2v2 — luma at full resolution, both chroma planes alike */
            if (C[0].h != hmax || C[0].v != vmax || C[1].h != C[2].h || C[1].v != C[2].v) { rc = VOO_JPEG_UNSUPPORTED; goto done; }
            const int okc = (C[1].h == hmax && C[1].v == vmax) || (C[1].h * 2 == hmax && C[1].v == vmax) || (C[1].h * 2 == hmax && C[1].v * 2 == vmax);
            if (!okc) { rc = VOO_JPEG_UNSUPPORTED; goto done; }
        }
        const int mx = (W + 8 * hmax - 1) / (8 * hmax), my = (Hh + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < nc; i++) {
            C[i].bw = mx * C[i].h; C[i].bh = my * C[i].v;
            C[i].dw = (W * C[i].h + hmax - 1) / hmax; C[i].dh = (Hh * C[i].v + vmax - 1) / vmax;
            C[i].plane = (uint8_t*)malloc((size_t)C[i].bw * 8 * C[i].bh * 8);
            if (!C[i].plane) goto done;
        }
        /* entropy-coded segment: MCU by MCU */
        bits_t b = {data, n, pos, 0, 0, 0, 0};
        int16_t coef[64];
        const long total = (long)mx * my;
        for (long mcu = 0; mcu < total; mcu++) {
            if (ri && mcu && mcu % ri == 0) {                                 /* restart: byte-align, expect RSTn */
                b.cnt = 0;
                if (!b.marker) {                                              /* (padding bits were not all consumed) */
                    if (b.pos + 1 < n && b.p[b.pos] == 0xFF && b.p[b.pos + 1] >= 0xC0) b.marker = b.p[b.pos + 1];
                    else if (b.pos >= n) b.marker = 0xD9;                     /* the file simply ends (the data source supplies an EOI) */
                }
                if (b.marker >= 0xD0 && b.marker <= 0xD7) { b.pos += 2; b.marker = 0; b.insufficient = 0; }
                /* any other marker where RSTn should stand — a file cut exactly at the end of an interval: jdmarker.c
                 * jpeg_resync_to_restart, action 3: "valid non-restart marker: return without advancing" — the marker stays
                 * unread, process_restart resets the predictions (below), and the next MCU runs out of data as usual */
                else if (!b.marker && !b.insufficient) goto done;             /* neither a marker nor the end: damaged data */
                for (int i = 0; i < nc; i++) C[i].pred = 0;
            }
            /* jdhuff.c decode_mcu: "If we've run out of data, just leave the MCU set to zeroes" — the MCU during which the
             * data ran out is still decoded (from zero bits), every later one of the segment is left all-zero */
            const int skip = b.insufficient;
            const int mxi = (int)(mcu % mx), myi = (int)(mcu / mx);
            for (int i = 0; i < nc; i++)
                for (int by = 0; by < C[i].v; by++)
                    for (int bx = 0; bx < C[i].h; bx++) {
                        memset(coef, 0, sizeof(coef));
                        const int px = (mxi * C[i].h + bx) * 8, py = (myi * C[i].v + by) * 8;
                        if (skip) { idct_islow(coef, Q[C[i].tq], C[i].plane + (size_t)py * (C[i].bw * 8) + px, C[i].bw * 8); continue; }
                        int sy = decode_sym(&b, &H[C[i].td]);
                        if (sy < 0 || sy > 15) goto done;
                        C[i].pred += extend(receive(&b, sy), sy);
                        coef[0] = (int16_t)C[i].pred;
                        for (int k = 1; k < 64; k++) {
                            const int rs = decode_sym(&b, &H[4 + C[i].ta]);
                            if (rs < 0) goto done;
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) { if (r == 15) { k += 15; continue; } break; }
                            k += r;
                            /* a run that leaves the block (damaged or zero-filled data): jdhuff.c stores through the 16 extra
                             * entries of jpeg_natural_order[], all 63 — the value lands in the last coefficient, the block ends */
                            coef[ZIGZAG[k > 63 ? 63 : k]] = (int16_t)extend(receive(&b, sz), sz);
                        }
                        idct_islow(coef, Q[C[i].tq], C[i].plane + (size_t)py * (C[i].bw * 8) + px, C[i].bw * 8);
                    }
        }
        /* upsample + colour conversion, row by row */
        int ycc = 1;                                                          /* jdapimin.c default_decompress_parms */
        if (nc == 3) {
            if (jfif) ycc = 1;
            else if (adobe) ycc = adobe_tr != 0;
            else ycc = !(C[0].id == 'R' && C[1].id == 'G' && C[2].id == 'B');
        }
        uint8_t* cbrow = (uint8_t*)malloc((size_t)2 * (C[nc - 1].bw * 8 * 2 + 16));
        if (!cbrow) goto done;
        uint8_t* crrow = cbrow + (C[nc - 1].bw * 8 * 2 + 16);
        for (int y = 0; y < Hh; y++) {
            const uint8_t* yrow = C[0].plane + (size_t)y * (C[0].bw * 8);
            uint8_t* o = out + (size_t)y * out_stride;
            if (nc == 1) { for (int x = 0; x < W; x++) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = yrow[x]; continue; }
            const uint8_t* cb; const uint8_t* cr;
            const int st = C[1].bw * 8, dw = C[1].dw, dh = C[1].dh;
            if (C[1].h == C[0].h && C[1].v == C[0].v) { cb = C[1].plane + (size_t)y * st; cr = C[2].plane + (size_t)y * st; }
            else if (C[1].v == C[0].v) {                                      /* h2v1 */
                up_h2v1(C[1].plane + (size_t)y * st, dw, cbrow); up_h2v1(C[2].plane + (size_t)y * st, dw, crrow);
                cb = cbrow; cr = crrow;
            } else {                                                          /* h2v2 */
                const int r = y >> 1;
                if (dw <= 2) {                                                /* h2v2_upsample: replication */
                    for (int i = 0; i < dw; i++) { cbrow[2 * i] = cbrow[2 * i + 1] = C[1].plane[(size_t)r * st + i]; crrow[2 * i] = crrow[2 * i + 1] = C[2].plane[(size_t)r * st + i]; }
                } else {
                    int r1 = (y & 1) ? r + 1 : r - 1;                         /* the context row: above for even, below for odd rows */
                    if (r1 < 0) r1 = 0;                                       /* jdmainct.c: rows above the image = row 0 */
                    if (r1 > dh - 1) r1 = dh - 1;                             /* rows below = the last real row */
                    up_h2v2(C[1].plane + (size_t)r * st, C[1].plane + (size_t)r1 * st, dw, cbrow);
                    up_h2v2(C[2].plane + (size_t)r * st, C[2].plane + (size_t)r1 * st, dw, crrow);
                }
                cb = cbrow; cr = crrow;
            }
            for (int x = 0; x < W; x++) {
                const int Y = yrow[x], Cb = cb[x], Cr = cr[x];
                if (!ycc) { o[3 * x] = (uint8_t)Cr; o[3 * x + 1] = (uint8_t)Cb; o[3 * x + 2] = (uint8_t)Y; continue; }   /* stored R,G,B */
                const int xb = Cb - 128, xr = Cr - 128;                       /* jdcolor.c build_ycc_rgb_table */
                const int R = Y + ((91881 * xr + 32768) >> 16);
                const int G = Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16);
                const int B = Y + ((116130 * xb + 32768) >> 16);
                o[3 * x] = clamp8(B); o[3 * x + 1] = clamp8(G); o[3 * x + 2] = clamp8(R);
            }
        }
        free(cbrow);
        rc = VOO_OK;
    }
done:
    for (int i = 0; i < 3; i++) free(C[i].plane);
    free(H);
    return rc;
}
