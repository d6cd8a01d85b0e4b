// The signal word of the small-batch hand-off (mcorb_signal.h): every field survives the round trip at its extremes, fields do not
// bleed into each other, and the checksum notices a changed, missing or shifted entry.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mcorb_signal.h"

using namespace mcorb;

int main()
{
    int bad = 0;
    std::mt19937 rng(7);
    for (int it = 0; it < 200000; it++) {
        const int b = (int)(rng() % 3), c = (int)(rng() % (kSelSignalMaxCount + 1)), m = (int)(rng() % (c + 1));
        const uint32_t x = (uint32_t)rng();
        const unsigned long long w = sel_signal(b, c, m, x);
        if (!sel_signal_done(w) || sel_signal_bad(w) != b || sel_signal_count(w) != c || sel_signal_mono(w) != m || sel_signal_check(w) != x) bad++;
    }
    {   // extremes
        const unsigned long long w = sel_signal(2, kSelSignalMaxCount, kSelSignalMaxCount, 0xffffffffu);
        if (sel_signal_bad(w) != 2 || sel_signal_count(w) != kSelSignalMaxCount || sel_signal_mono(w) != kSelSignalMaxCount || sel_signal_check(w) != 0xffffffffu) bad++;
        if (sel_signal_done(0ull)) bad++;   // the host's reset value is "not done"
    }
    // checksum: any single changed value, a stale (zero) entry, or two swapped entries change the XOR
    for (int it = 0; it < 2000; it++) {
        const int n = 1 + (int)(rng() % 3000);
        std::vector<uint32_t> sel(n);
        std::vector<uint8_t> resp(n);
        for (int k = 0; k < n; k++) { sel[k] = (uint32_t)rng(); resp[k] = (uint8_t)rng(); }
        auto sum = [&]() { uint32_t x = 0; for (int k = 0; k < n; k++) x ^= sel_check(sel[k], resp[k], k); return x; };
        const uint32_t ref = sum();
        const int k = (int)(rng() % n);
        const uint32_t s0 = sel[k]; const uint8_t r0 = resp[k];
        sel[k] ^= 1u << (rng() % 32); if (sum() == ref) bad++; sel[k] = s0;
        resp[k] ^= (uint8_t)(1u << (rng() % 8)); if (sum() == ref) bad++; resp[k] = r0;
        if (s0 != 0 || r0 != 0) { sel[k] = 0; resp[k] = 0; if (sum() == ref) bad++; sel[k] = s0; resp[k] = r0; }
        if (n > 1) {
            const int j = (k + 1 + (int)(rng() % (n - 1))) % n;
            if (sel[j] != sel[k] || resp[j] != resp[k]) {
                std::swap(sel[j], sel[k]); std::swap(resp[j], resp[k]);
                if (sum() == ref) bad++;
                std::swap(sel[j], sel[k]); std::swap(resp[j], resp[k]);
            }
        }
    }
    printf("signal word: bad=%d\n", bad);
    return bad ? 1 : 0;
}
