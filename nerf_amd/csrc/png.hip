// Host-side helper of the image reader: PNG scanline un-filtering (filter types 0-4 of the PNG
// specification).  The Average and Paeth filters depend on the already reconstructed pixel to the
// left, which makes them a per-byte sequential loop -- too slow in Python for 800x800 frames.
#include "common.h"

// raw: h rows of (1 filter byte + stride data bytes); out: h rows of stride bytes; bpp: bytes per pixel.
NERF_API int nerf_png_unfilter(const unsigned char* raw, int64_t h, int64_t stride, int bpp, unsigned char* out) {
    NERF_REQUIRE(raw && out && h >= 0 && stride > 0 && bpp > 0 && bpp <= 8, "nerf_png_unfilter: bad argument");
    for (int64_t y = 0; y < h; ++y) {
        const unsigned char* src = raw + y * (stride + 1);
        const int f = src[0];
        ++src;
        unsigned char* cur = out + y * stride;
        const unsigned char* up = y ? cur - stride : nullptr;
        NERF_REQUIRE(f >= 0 && f <= 4, "nerf_png_unfilter: row %lld has filter type %d", (long long)y, f);
        for (int64_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pred = 0;
            if (f == 1) pred = a;
            else if (f == 2) pred = b;
            else if (f == 3) pred = (a + b) >> 1;
            else if (f == 4) {
                const int p = a + b - c;
                const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[x] = (unsigned char)(src[x] + pred);
        }
    }
    return 0;
}
