/*
 * mcorb_synth.c -- deterministic synthetic rig frames.
 *
 * The reference ships no images, so tests and bench.py draw their input from
 * this integer-only generator (numpy mirror: mc-slam_amd/synth.py; the two are
 * checked against each other in tests/test_synth.py).
 *
 * Frame f of a rig of C cameras at W x H:
 *   canvas (W + 24*(C-1)) x H, filled with 128;
 *   nrect = 6000*W*H/921600 axis-aligned filled rectangles drawn in sequence
 *   from xorshift64* seeded with (0x4D435F53 ^ f): w,h in [6,60], top-left
 *   uniform over the canvas (clipped), gray uniform 0..255;
 *   per-pixel noise in [-6,+6] from a counter hash (splitmix64 finaliser of
 *   the canvas pixel index and the seed), added with clamping;
 *   camera c is the crop at x offset 24*c (pure horizontal disparity: rows
 *   stay aligned, so cross-camera matches are real).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t xs64star(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static inline uint32_t xs_uniform(uint64_t *s, uint32_t n) { return (uint32_t)((xs64star(s) >> 33) % n); }

static inline uint64_t splitmix_hash(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define MCORB_SYNTH_DISPARITY 24

/* Writes camera `cam` of frame `frame` into out (stride bytes per row).
 * Returns 0, or -1 on bad arguments. */
int mcorb_synth_rig_frame(uint32_t frame, int ncams, int cam, int w, int h, uint8_t *out, int stride)
{
    if (!out || ncams < 1 || cam < 0 || cam >= ncams || w < 1 || h < 1 || stride < w) return -1;
    const int cw = w + MCORB_SYNTH_DISPARITY * (ncams - 1);
    uint8_t *canvas = (uint8_t *)malloc((size_t)cw * h);
    if (!canvas) return -1;
    memset(canvas, 128, (size_t)cw * h);
    const uint64_t seed = 0x4D435F53ULL ^ (uint64_t)frame;
    uint64_t s = seed;
    const int nrect = (int)((6000LL * w * h) / 921600LL);
    for (int r = 0; r < nrect; r++) {
        int rw = 6 + (int)xs_uniform(&s, 55);
        int rh = 6 + (int)xs_uniform(&s, 55);
        int x0 = (int)xs_uniform(&s, (uint32_t)cw);
        int y0 = (int)xs_uniform(&s, (uint32_t)h);
        int g = (int)xs_uniform(&s, 256);
        int x1 = x0 + rw > cw ? cw : x0 + rw;
        int y1 = y0 + rh > h ? h : y0 + rh;
        for (int y = y0; y < y1; y++) memset(canvas + (size_t)y * cw + x0, g, (size_t)(x1 - x0));
    }
    const uint64_t nseed = seed * 0x9E3779B97F4A7C15ULL;
    const int xoff = MCORB_SYNTH_DISPARITY * cam;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            uint64_t idx = (uint64_t)y * (uint64_t)cw + (uint64_t)(x + xoff);
            int n = (int)((splitmix_hash(idx + nseed) >> 33) % 13) - 6;
            int v = canvas[(size_t)y * cw + x + xoff] + n;
            out[(size_t)y * stride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(canvas);
    return 0;
}
