// Host selection stage under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on this pool).
// Random and clustered candidate sets in raster order: mcorb::host_bucket_sort (what k_compact does) + mcorb::select_octree
// must return the indices and the ORDER of the oracle's literal DistributeOctTree restatement.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <set>
#include <vector>

#include "mcorb_oracle.h"
#include "mcorb_select.h"

static int run_case(std::mt19937 &rng, int W, int H, int N, int npts, int box)
{
    const int w = W - 32, h = H - 32;
    std::set<std::pair<int, int>> pts;   // (y, x): iteration order is raster order
    std::uniform_int_distribution<int> ux(0, w - 1), uy(0, h - 1);
    if (box > 0) {
        const int bx = ux(rng) % std::max(1, w - box), by = uy(rng) % std::max(1, h - box);
        std::uniform_int_distribution<int> b(0, box - 1);
        for (int i = 0; i < 4 * npts && (int)pts.size() < npts; i++) pts.insert({by + b(rng), bx + b(rng)});
        for (int i = 0; i < 20; i++) pts.insert({uy(rng), ux(rng)});
    } else {
        while ((int)pts.size() < npts) pts.insert({uy(rng), ux(rng)});
    }
    std::vector<uint32_t> packed;
    std::vector<float> fx, fy, fr;
    std::uniform_int_distribution<int> ur(20, 24);
    for (auto &p : pts) {
        const int r = ur(rng);
        packed.push_back(mcorb::pack_cand(p.second, p.first, r));
        fx.push_back((float)p.second); fy.push_back((float)p.first); fr.push_back((float)r);
    }
    const int n = (int)packed.size();
    std::vector<int> want(n + 8), got(N + 64 + 8);
    std::vector<uint32_t> gotv(N + 64 + 8);
    const int nw = orc_distribute_octree(fx.data(), fy.data(), fr.data(), n, 16, W - 16, 16, H - 16, N, want.data(), (int)want.size());
    const mcorb::SelectParams P = mcorb::make_select_params(16, W - 16, 16, H - 16, N, 0, 0);
    std::vector<uint32_t> sorted;
    std::vector<int> perm, bstart;
    std::vector<mcorb::BucketBest> bbest;
    mcorb::host_bucket_sort(packed.data(), n, P, sorted, perm, bstart, bbest);
    static mcorb::SelectScratch sc;
    const int ng = mcorb::select_octree(sorted.data(), bstart.data(), bbest.data(), n, P, got.data(), gotv.data(), sc);
    if (ng != nw) { fprintf(stderr, "count %d != %d (W %d H %d N %d n %d box %d)\n", ng, nw, W, H, N, n, box); return 1; }
    for (int i = 0; i < ng; i++)
        if (perm[got[i]] != want[i] || gotv[i] != packed[want[i]]) { fprintf(stderr, "entry %d differs (W %d H %d N %d n %d box %d)\n", i, W, H, N, n, box); return 1; }
    return 0;
}

// k_compact's table form of the path code: lutx[x] | luty[y] must equal path_code(x, y) for EVERY candidate position
static int run_path_tables(int W, int H, int N)
{
    const mcorb::SelectParams P = mcorb::make_select_params(16, W - 16, 16, H - 16, N, 0, 0);
    if (P.nIni < 1) return 0;
    const int W0 = W - 32, H0 = H - 32;
    std::vector<uint16_t> tx(W0), ty(H0);
    mcorb::path_code_tables(W0, H0, P.nIni, P.hX, P.depth, tx.data(), ty.data());
    for (int y = 0; y < H0; y++)
        for (int x = 0; x < W0; x++)
            if (((uint32_t)tx[x] | (uint32_t)ty[y]) != mcorb::path_code(x, y, W0, H0, P.nIni, P.hX, P.depth)) {
                fprintf(stderr, "path code table mismatch at (%d, %d), %dx%d N %d\n", x, y, W, H, N);
                return 1;
            }
    return 0;
}

extern "C" int mcorb_synth_rig_frame(uint32_t frame, int ncams, int cam, int w, int h, uint8_t *out, int stride);

// the oracle itself (the checker of every parity claim) through the same sanitizers: extraction, matching, track merge
static int run_oracle(int W, int H, int N)
{
    std::vector<uint8_t> img[2];
    std::vector<orc_keypoint> kps[2];
    std::vector<uint8_t> desc[2];
    int n[2] = {0, 0};
    orc_extractor *e = orc_create(N, 1.2f, 8, 20, 7, 0);
    for (int c = 0; c < 2; c++) {
        img[c].resize((size_t)W * H);
        mcorb_synth_rig_frame(1, 2, c, W, H, img[c].data(), W);
        kps[c].resize(N + 512); desc[c].resize((size_t)(N + 512) * 32);
        if (orc_extract(e, img[c].data(), W, H, W, 0, 0, kps[c].data(), desc[c].data(), N + 512, &n[c]) < 0) return 1;
    }
    orc_destroy(e);
    const uint8_t *dp[2] = {desc[0].data(), desc[1].data()};
    std::vector<int32_t> tracks((size_t)(n[0] + n[1] + 1) * 2);
    int mergeable = 0;
    const int nt = orc_intra_matches(dp, n, 2, 75.f, 0.85f, tracks.data(), n[0] + n[1] + 1, &mergeable);
    printf("oracle %dx%d: %d + %d keypoints, %d tracks\n", W, H, n[0], n[1], nt);
    return (n[0] > 50 && nt > 10) ? 0 : 1;
}

int main()
{
    std::mt19937 rng(12345);
    int bad = 0, cases = 0;
    const int cfg[][5] = {{640, 480, 217, 3000, 0}, {1280, 720, 434, 9000, 0}, {357, 201, 122, 900, 0}, {300, 300, 50, 40, 0},
                          {640, 480, 5, 700, 0},   {1280, 720, 434, 3000, 90}, {640, 480, 217, 2500, 60}, {1280, 720, 434, 6000, 200},
                          {1920, 1080, 434, 20000, 0}, {200, 120, 30, 1, 0}, {752, 480, 300, 2, 0}};
    for (auto &c : cfg)
        for (int rep = 0; rep < 3; rep++, cases++) bad += run_case(rng, c[0], c[1], c[2], c[3], c[4]);
    const int geo[][3] = {{1280, 720, 434}, {1067, 600, 362}, {357, 201, 122}, {1920, 1080, 434}, {640, 480, 5}, {536, 301, 122},
                          {4000, 300, 2000}, {300, 400, 30}, {803, 601, 3000}};
    for (auto &q : geo) bad += run_path_tables(q[0], q[1], q[2]);
    bad += run_oracle(320, 240, 500);
    bad += run_oracle(411, 305, 800);
    printf("select_sanitize cases=%d bad=%d\n", cases, bad);
    return bad ? 1 : 0;
}
