// Exercises the C++ host mirror (include/mcorb_adapter.hpp) the way MC-SLAM's own code uses
// ORBextractor / MultiCameraFrame: construct, setData, extractFeaturesParallel, BruteForceMatch,
// computeIntraMatches.  Prints FNV-1a checksums that tests/test_gpu_cpp_adapter.py compares with
// the CPU oracle's outputs for the same synthetic frames.
#include <stdio.h>
#include <stdlib.h>

#include "mcorb_adapter.hpp"

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 3, W = argc > 2 ? atoi(argv[2]) : 640, H = argc > 3 ? atoi(argv[3]) : 480;
    const int N = argc > 4 ? atoi(argv[4]) : 1000, frame = argc > 5 ? atoi(argv[5]) : 0;
    try {
        std::vector<std::vector<uint8_t>> imgs(C, std::vector<uint8_t>((size_t)W * H));
        std::vector<const uint8_t *> ptrs;
        for (int c = 0; c < C; c++) {
            mcorb_synth_rig_frame(frame, C, c, W, H, imgs[c].data(), W);
            ptrs.push_back(imgs[c].data());
        }
        // single-camera ORBextractor, as MultiCameraFrame::extractFeatureSingle calls it (:236)
        mcorb::ORBextractor ext(N, 1.2f, 8, 20, 7);
        std::vector<mcorb_keypoint> kps;
        std::vector<uint8_t> desc;
        std::vector<int> vLapping = {0, 0};
        const int mono = ext(imgs[0].data(), W, H, W, kps, desc, vLapping);
        uint64_t h = 1469598103934665603ULL;
        h = fnv(h, kps.data(), kps.size() * sizeof(mcorb_keypoint));
        h = fnv(h, desc.data(), desc.size());
        printf("extractor mono=%d n=%zu hash=%016llx levels=%d\n", mono, kps.size(), (unsigned long long)h, ext.GetLevels());
        std::vector<uint8_t> none;
        printf("empty=%d\n", ext(nullptr, 0, 0, 0, kps, none, vLapping));

        // rig: extractFeaturesParallel + all-pairs BruteForceMatch + computeIntraMatches(matches, false)
        mcorb_params p;
        mcorb_default_params(&p);
        p.nfeatures = N;
        mcorb::MultiCameraFrontEnd fr(C, W, H, p);
        fr.setData(ptrs, W);
        fr.extractFeaturesParallel();
        uint64_t hr = 1469598103934665603ULL;
        for (int c = 0; c < C; c++) {
            hr = fnv(hr, fr.image_kps[c].data(), fr.image_kps[c].size() * sizeof(mcorb_keypoint));
            hr = fnv(hr, fr.image_descriptors[c].data(), fr.image_descriptors[c].size());
        }
        uint64_t hm = 1469598103934665603ULL;
        size_t nm = 0;
        for (int i = 0; i < C - 1; i++)
            for (int j = i + 1; j < C; j++) {
                std::vector<unsigned int> i1, i2;
                std::vector<mcorb_keypoint> k1, k2;
                fr.BruteForceMatch(i, j, 75, 0.85f, i1, i2, k1, k2);
                hm = fnv(hm, i1.data(), i1.size() * 4);
                hm = fnv(hm, i2.data(), i2.size() * 4);
                nm += i1.size();
            }
        std::vector<mcorb::IntraMatch> matches;
        fr.computeIntraMatches(matches, false);
        uint64_t ht = 1469598103934665603ULL;
        for (auto &m : matches) ht = fnv(ht, m.matchIndex.data(), (size_t)C * sizeof(int));
        printf("rig features=%016llx matches=%zu/%016llx tracks=%zu/%016llx mergeable=%d\n", (unsigned long long)hr, nm,
               (unsigned long long)hm, matches.size(), (unsigned long long)ht, fr.cnt_mergable_matches);

        // computeIntraMatches(matches, true): side-by-side rig with K = I, F = [t]x for every pair
        std::vector<double> F;
        for (int q = 0; q < C * (C - 1) / 2; q++) {
            const double f[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
            F.insert(F.end(), f, f + 9);
        }
        fr.setFundamental(F);
        fr.computeIntraMatches(matches, true);
        uint64_t he = 1469598103934665603ULL;
        for (auto &m : matches) he = fnv(he, m.matchIndex.data(), (size_t)C * sizeof(int));
        printf("epipolar tracks=%zu/%016llx mergeable=%d\n", matches.size(), (unsigned long long)he, fr.cnt_mergable_matches);

        // computeIntraMatches(matches, words_) + transform() with a vocabulary text file (argv[6], optional)
        if (argc > 6) {
            mcorb::ORBVocabulary voc;
            if (!voc.loadFromTextFile(argv[6])) throw std::runtime_error("vocabulary did not load");
            std::vector<unsigned int> words_;
            fr.computeIntraMatches(matches, words_, voc, 0.85, 2);
            uint64_t hb = 1469598103934665603ULL;
            for (auto &m : matches) {
                hb = fnv(hb, m.matchIndex.data(), (size_t)C * sizeof(int));
                hb = fnv(hb, &m.n_rays, sizeof(int));
            }
            hb = fnv(hb, words_.data(), words_.size() * sizeof(unsigned int));
            mcorb::ORBVocabulary::BowVector bow;
            mcorb::ORBVocabulary::FeatureVector fv;
            fr.transform(0, voc, bow, fv, 2);
            uint64_t hv = 1469598103934665603ULL;
            for (auto &e : bow) { hv = fnv(hv, &e.first, 4); hv = fnv(hv, &e.second, 8); }
            for (auto &e : fv) { hv = fnv(hv, &e.first, 4); hv = fnv(hv, e.second.data(), e.second.size() * 4); }
            printf("bow tracks=%zu words=%zu hash=%016llx transform=%zu/%zu/%016llx\n", matches.size(), words_.size(),
                   (unsigned long long)hb, bow.size(), fv.size(), (unsigned long long)hv);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "FAILED: %s\n", e.what());
        return 1;
    }
    return 0;
}
