// test_sortmodel.cpp -- known-answer test of mcorb_sortmodel.h (the statement of libstdc++'s std::sort the GPU selection kernel
// runs) against std::sort itself, with the comparator the host selection stage uses (upper 32 bits only):
//   * 12 000 random multisets, 0 .. 700 entries, key alphabets from 1 to 2^20 values (ties everywhere), incl. the (count << 12 | x)
//     shape DistributeOctTree sorts;
//   * sorted, reversed, all-equal, organ-pipe and saw-tooth sequences;
//   * median-of-three killer sequences (McIlroy's adversary run against std::sort itself) that drive the introsort loop into its
//     heap-sort branch -- the test asserts that branch was really taken.
// Prints "sortmodel cases=<n> heap_cases=<k> bad=<m>"; exit status 0 iff bad == 0 and heap_cases > 0.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mcorb_sortmodel.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

static int check(const std::vector<uint32_t> &keys, int *heap_ranges)
{
    const int n = (int)keys.size();
    std::vector<uint64_t> a(n), b(n);
    for (int i = 0; i < n; i++) a[i] = b[i] = ((uint64_t)keys[i] << 32) | (uint32_t)i;
    std::sort(a.begin(), a.end(), [](uint64_t x, uint64_t y) { return (x >> 32) < (y >> 32); });
    mcorb::sort_model(b.data(), n, heap_ranges);
    return a == b ? 0 : 1;
}

// McIlroy, "A killer adversary for quicksort": values are decided while the sort asks for comparisons
static std::vector<int> adv_val;
static int adv_nsolid, adv_candidate, adv_gas;
static bool adv_less(int x, int y)
{
    if (adv_val[x] == adv_gas && adv_val[y] == adv_gas) {
        if (x == adv_candidate) adv_val[x] = adv_nsolid++;
        else adv_val[y] = adv_nsolid++;
    }
    if (adv_val[x] == adv_gas) adv_candidate = x;
    else if (adv_val[y] == adv_gas) adv_candidate = y;
    return adv_val[x] < adv_val[y];
}
static std::vector<uint32_t> killer(int n)
{
    adv_val.assign(n, n - 1);
    adv_gas = n - 1; adv_nsolid = 0; adv_candidate = 0;
    std::vector<int> ptr(n);
    for (int i = 0; i < n; i++) ptr[i] = i;
    std::sort(ptr.begin(), ptr.end(), adv_less);
    std::vector<uint32_t> k(n);
    for (int i = 0; i < n; i++) k[i] = (uint32_t)adv_val[i];
    return k;
}

int main(int argc, char **argv)
{
    if (argc == 3 && std::string(argv[1]) == "--dump-killer") {   // the killer sequence of that length, one key per line (GPU KAT input)
        for (uint32_t v : killer(atoi(argv[2]))) printf("%u\n", v);
        return 0;
    }
    int bad = 0, cases = 0, heap_cases = 0;
    for (int it = 0; it < 12000; it++) {
        const int n = it < 40 ? it : (int)(rnd() % 701);
        const int shape = (int)(rnd() % 6);
        std::vector<uint32_t> k(n);
        const uint32_t alpha = shape == 0 ? 1 : shape == 1 ? 2 + rnd() % 3 : shape == 2 ? 8 + rnd() % 24 : shape == 3 ? 200 : 1u << 20;
        for (int i = 0; i < n; i++) {
            if (shape == 5) k[i] = ((2 + rnd() % 40) << 12) | ((rnd() % 24) * 39);   // (key count << 12 | UL.x): few counts, few columns
            else k[i] = rnd() % alpha;
        }
        int h = 0;
        bad += check(k, &h);
        heap_cases += h > 0;
        cases++;
    }
    for (int n : {17, 33, 64, 100, 128, 129, 255, 434, 512, 700}) {
        std::vector<uint32_t> k(n);
        int h;
        for (int i = 0; i < n; i++) k[i] = (uint32_t)i;
        bad += check(k, &h); cases++;
        for (int i = 0; i < n; i++) k[i] = (uint32_t)(n - i);
        bad += check(k, &h); cases++;
        for (int i = 0; i < n; i++) k[i] = (uint32_t)(i < n / 2 ? i : n - i);
        bad += check(k, &h); cases++;
        for (int i = 0; i < n; i++) k[i] = (uint32_t)(i % 7);
        bad += check(k, &h); cases++;
        for (int i = 0; i < n; i++) k[i] = (uint32_t)(i / 5);
        bad += check(k, &h); cases++;
    }
    for (int n : {40, 64, 100, 128, 200, 256, 434, 500, 700, 1024, 1519}) {
        const std::vector<uint32_t> k = killer(n);
        int h = 0;
        bad += check(k, &h);
        heap_cases += h > 0;
        cases++;
        // the same sequence with ties folded in
        std::vector<uint32_t> k2(k);
        for (auto &v : k2) v /= 3;
        bad += check(k2, &h);
        heap_cases += h > 0;
        cases++;
    }
    printf("sortmodel cases=%d heap_cases=%d bad=%d\n", cases, heap_cases, bad);
    return bad == 0 && heap_cases > 0 ? 0 : 1;
}
