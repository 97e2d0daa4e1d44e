// tiff_lzw.cpp -- TIFF-flavoured LZW encoder (host code behind the C-ABI, no GPU involved).
//
// The reference writes its GeoTIFF exports with rasterio's compress='lzw' (pydem/process_manager.py:905, :930); the writer
// of pydem_amd/raster.py offers the same codec through this encoder.  The stream is what libtiff's encoder emits for the
// same bytes (tif_lzw.c: MSB-first codes of 9..12 bits, ClearCode 256 first, EndOfInformation 257 last, the code width
// grows one entry EARLY -- "early change" --, a full table (4094 entries) is followed by a ClearCode, and so is a
// compression ratio that stopped improving between two 10000-byte checkpoints): GDAL, rasterio and
// the reader of raster.py decode it.  tests/test_raster.py holds the output byte for byte against libtiff's (Pillow) on
// strips of every dtype.
#include "internal.h"
#include <string.h>

namespace {
constexpr int CODE_CLEAR = 256, CODE_EOI = 257, CODE_FIRST = 258, BITS_MIN = 9, BITS_MAX = 12;
constexpr int CODE_MAX = (1 << BITS_MAX) - 1;       // 4095
constexpr int HSIZE = 9001;                         // 91 % occupancy, like libtiff
}

extern "C" int pydem_tiff_lzw_encode(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int64_t *out_n)
{
    int64_t op = 0;
    uint64_t acc = 0; int have = 0;
    int nbits = BITS_MIN, maxcode = (1 << BITS_MIN) - 1, free_ent = CODE_FIRST;
    bool full = false;
    // libtiff's compression-ratio watch (tif_lzw.c: CHECK_GAP, CALCRATIO): every 10000 input bytes the ratio input bytes /
    // output BITS (a 24.8 fixed-point number) is compared with the one of the previous check, and when it has not improved
    // the table is dropped with a ClearCode -- part of the stream, so it is reproduced
    long incount = 0, outcount = 0, checkpoint = 10000, enc_ratio = 0;
    auto put = [&](int code) {
        acc = (acc << nbits) | (uint64_t)code; have += nbits;
        while (have >= 8) { if (op < cap) dst[op] = (uint8_t)(acc >> (have - 8)); else full = true; op++; have -= 8; }
        outcount += nbits;
    };
    static thread_local int32_t h_key[HSIZE]; static thread_local int16_t h_code[HSIZE];
    auto clear_table = [&]() { memset(h_key, 0xff, sizeof(h_key)); };
    clear_table();
    put(CODE_CLEAR);
    if (n > 0) {
        int ent = src[0];
        incount++;
        for (int64_t i = 1; i < n; i++) {
            const int c = src[i];
            incount++;
            const int32_t key = (c << BITS_MAX) + ent;
            int h = (c << 5) ^ ent;                          // (xor hashing; secondary probe below)
            if (h >= HSIZE) h -= HSIZE;
            bool found = false;
            if (h_key[h] == key) { ent = h_code[h]; found = true; }
            else if (h_key[h] >= 0) {
                int disp = h == 0 ? 1 : HSIZE - h;
                for (;;) {
                    if ((h -= disp) < 0) h += HSIZE;
                    if (h_key[h] == key) { ent = h_code[h]; found = true; break; }
                    if (h_key[h] < 0) break;
                }
            }
            if (found) continue;
            put(ent);
            ent = c;
            h_code[h] = (int16_t)free_ent++;
            h_key[h] = key;
            auto restart = [&]() {
                clear_table();
                enc_ratio = 0; incount = 0; outcount = 0;
                free_ent = CODE_FIRST;
                put(CODE_CLEAR);
                nbits = BITS_MIN; maxcode = (1 << BITS_MIN) - 1;
            };
            if (free_ent == CODE_MAX - 1) restart();         // the table is full: ClearCode, start over
            else if (free_ent > maxcode) {
                nbits++; maxcode = (1 << nbits) - 1;
            } else if (incount >= checkpoint) {
                checkpoint = incount + 10000;
                long rat;
                if (incount > 0x007fffff) { rat = outcount >> 8; rat = rat == 0 ? 0x7fffffff : incount / rat; }
                else rat = (incount << 8) / outcount;
                if (rat <= enc_ratio) restart(); else enc_ratio = rat;
            }
        }
        // the last string; the decoder adds one more entry when it reads the code behind it, so the width may grow before
        // the EndOfInformation code (libtiff LZWPostEncode)
        put(ent);
        free_ent++;
        if (free_ent == CODE_MAX - 1) { put(CODE_CLEAR); nbits = BITS_MIN; }
        else if (free_ent > maxcode) { nbits++; }
    }
    put(CODE_EOI);
    if (have > 0) { if (op < cap) dst[op] = (uint8_t)(acc << (8 - have)); else full = true; op++; }
    *out_n = op;
    if (full) { pydem_set_error("pydem_tiff_lzw_encode: output buffer too small (%lld bytes needed)", (long long)op); return -2; }
    return 0;
}
