// Driver of the sanitizer build of visual_odometry_amd/csrc/jpeg_host.cpp (CPU only): parses every file named on the
// command line with jpeg_info and jpeg_parse, the second one also through the "same header as the previous file" shortcut.
// Any status is acceptable; AddressSanitizer / UBSan abort the process on a memory error.
#include "jpeg_host.h"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv)
{
    std::vector<uint8_t> prev_hdr; JpegImage prev_img; JpegTables* prev_T = new JpegTables; bool have_prev = false;
    int ok = 0, rejected = 0;
    for (int a = 1; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        std::vector<uint8_t> d;
        uint8_t buf[4096]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        // an exactly-sized heap copy so that any read past the end is seen
        uint8_t* p = new uint8_t[d.size() ? d.size() : 1];
        for (size_t i = 0; i < d.size(); i++) p[i] = d[i];
        int h, w, nc, samp, orient;
        jpeg_info(p, d.size(), &h, &w, &nc, &samp, &orient);
        JpegImage img; JpegTables* T = new JpegTables; const char* why = nullptr;
        int rc = have_prev ? jpeg_parse(p, d.size(), &img, T, &why, prev_hdr.data(), prev_hdr.size(), &prev_img, prev_T)
                           : jpeg_parse(p, d.size(), &img, T, &why);
        if (rc == VO_OK) {
            ok++;
            if (img.hdr_len <= d.size()) { prev_hdr.assign(p, p + img.hdr_len); prev_img = img; *prev_T = *T; have_prev = true; }
        } else rejected++;
        delete T; delete[] p;
    }
    delete prev_T;
    printf("parsed %d accepted %d rejected\n", ok, rejected);
    return 0;
}
