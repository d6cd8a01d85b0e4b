// mcorb_adapter.hpp -- header-only C++ host mirror of the reference interfaces this path
// replaces, on top of the C ABI (mcorb.h).  Same names, argument meaning and error behaviour as
//   class ORBextractor                          MCSlam/include/MCSlam/ORBextractor.h:43-116
//   MultiCameraFrame::extractFeaturesParallel   MCSlam/include/MCSlam/MultiCameraFrame.h:74
//   MultiCameraFrame::BruteForceMatch           MultiCameraFrame.h:86
//   MultiCameraFrame::computeIntraMatches       MultiCameraFrame.h:84
//   class IntraMatch                            MultiCameraFrame.h:42-57
//
// Two flavours in one header:
//   * plain types (std::vector, raw pointers) -- compiles anywhere, used by tests/cpp;
//   * with -DMCORB_WITH_OPENCV the exact cv:: signatures of the reference are added, so that
//     MCSlam/src/ORBextractor.cpp can be dropped from the build and this header included from
//     MCSlam/include/MCSlam/ORBextractor.h instead.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mcorb.h"

#ifdef MCORB_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#endif

namespace mcorb {

const int TH_HIGH = 100;      // ORBextractor.h:26
const int TH_LOW = 75;        // ORBextractor.h:27
const int HISTO_LENGTH = 30;  // ORBextractor.h:28

inline void check(int st, const char *what)
{
    if (st != MCORB_OK) throw std::runtime_error(std::string(what) + ": " + mcorb_last_error());
}

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
    {
        mcorb_default_params(&p_);
        p_.nfeatures = nfeatures; p_.scale_factor = scaleFactor; p_.nlevels = nlevels;
        p_.ini_th_fast = iniThFAST; p_.min_th_fast = minThFAST; p_.device_id = device;
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels); mnFeaturesPerLevel.resize(nlevels);
        check(mcorb_get_tables(&p_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                               mvInvLevelSigma2.data(), mnFeaturesPerLevel.data()), "mcorb_get_tables");
        check(mcorb_create(&p_, 0, 0, &h_), "mcorb_create");
    }
    ~ORBextractor() { mcorb_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // operator() on plain buffers.  Returns monoIndex, or -1 for an empty image exactly like the
    // reference (ORBextractor.cpp:1090-1091); throws on the conditions where the reference would
    // divide by zero or assert.  The reference ignores its mask argument; it has none here.
    int operator()(const uint8_t *gray, int w, int h, int stride, std::vector<mcorb_keypoint> &_keypoints,
                   std::vector<uint8_t> &_descriptors, std::vector<int> &vLappingArea)
    {
        const int cap = p_.nfeatures + 8 * p_.nlevels + 64;
        _keypoints.resize(cap);
        _descriptors.resize((size_t)cap * 32);
        int n = 0, mono = 0;
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        const int st = mcorb_extract(h_, gray, w, h, stride, lap0, lap1, _keypoints.data(), _descriptors.data(), cap, &n, &mono);
        if (st == MCORB_E_EMPTY) { _keypoints.clear(); _descriptors.clear(); return -1; }
        check(st, "mcorb_extract");
        _keypoints.resize(n);
        _descriptors.resize((size_t)n * 32);
        return mono;
    }

    int inline GetLevels() { return p_.nlevels; }
    float inline GetScaleFactor() { return p_.scale_factor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // DescriptorDistance (ORBextractor.cpp:1202-1218) on 32-byte rows
    int DescriptorDistance(const uint8_t *a, const uint8_t *b) { return mcorb_hamming256(a, b); }

    // knnMatch(k=2) + the ratio / threshold filter of BruteForceMatch (MultiCameraFrame.cpp:1053-1078)
    void BruteForceMatch(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float dist_thresh, float neigh_ratio,
                         std::vector<unsigned int> &indices_1, std::vector<unsigned int> &indices_2)
    {
        indices_1.assign(n1 > 0 ? n1 : 0, 0);
        indices_2.assign(n1 > 0 ? n1 : 0, 0);
        int n = 0;
        check(mcorb_match_ratio(h_, d1, n1, d2, n2, dist_thresh, neigh_ratio, indices_1.data(), indices_2.data(), n1, &n),
              "mcorb_match_ratio");
        indices_1.resize(n);
        indices_2.resize(n);
    }

    // getMatches_distRatio (ORBextractor.cpp:1228-1290): best / second-best over index subsets on the
    // GPU k-NN kernel, then the reference's one-to-one bookkeeping.  A and B are n x 32 row-major.
    void getMatches_distRatio(const uint8_t *A, const std::vector<unsigned int> &i_A, const uint8_t *B,
                              const std::vector<unsigned int> &i_B, std::vector<unsigned int> &i_match_A,
                              std::vector<unsigned int> &i_match_B, int &BookK)
    {
        i_match_A.resize(0);
        i_match_B.resize(0);
        BookK += (int)(i_A.size() * i_B.size());
        if (i_A.empty() || i_B.empty()) return;
        std::vector<uint8_t> qa(i_A.size() * 32), tb(i_B.size() * 32);
        for (size_t i = 0; i < i_A.size(); i++) memcpy(&qa[i * 32], A + (size_t)i_A[i] * 32, 32);
        for (size_t i = 0; i < i_B.size(); i++) memcpy(&tb[i * 32], B + (size_t)i_B[i] * 32, 32);
        std::vector<int32_t> idx(i_A.size() * 2), dist(i_A.size() * 2);
        check(mcorb_knn2(h_, qa.data(), (int)i_A.size(), tb.data(), (int)i_B.size(), idx.data(), dist.data()), "mcorb_knn2");
        for (size_t a = 0; a < i_A.size(); a++) {
            const double best_dist_1 = dist[2 * a], best_dist_2 = idx[2 * a + 1] >= 0 ? (double)dist[2 * a + 1] : 1e9;
            if (best_dist_1 <= TH_LOW && best_dist_1 / best_dist_2 <= max_neighbor_ratio) {
                const unsigned int idx_B = i_B[idx[2 * a]];
                size_t k = 0;
                while (k < i_match_B.size() && i_match_B[k] != idx_B) k++;
                if (k == i_match_B.size()) {
                    i_match_B.push_back(idx_B);
                    i_match_A.push_back(i_A[a]);
                } else {
                    const double d = DescriptorDistance(A + (size_t)i_match_A[k] * 32, B + (size_t)idx_B * 32);
                    BookK++;
                    if (best_dist_1 < d) i_match_A[k] = i_A[a];
                }
            }
        }
    }

#ifdef MCORB_WITH_OPENCV
    // The reference's exact signature (ORBextractor.h:57-59).  cv::KeyPoint and mcorb_keypoint share
    // their field order, so the keypoints are copied as a block.
    int operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint> &_keypoints,
                   cv::OutputArray _descriptors, std::vector<int> &vLappingArea)
    {
        (void)_mask;   // "Mask is ignored in the current implementation." (ORBextractor.h:56)
        if (_image.empty()) return -1;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        std::vector<mcorb_keypoint> k;
        std::vector<uint8_t> d;
        const int mono = (*this)(image.data, image.cols, image.rows, (int)image.step, k, d, vLappingArea);
        static_assert(sizeof(cv::KeyPoint) == sizeof(mcorb_keypoint), "cv::KeyPoint layout");
        _keypoints.resize(k.size());
        if (!k.empty()) memcpy((void *)_keypoints.data(), k.data(), k.size() * sizeof(mcorb_keypoint));
        if (k.empty()) _descriptors.release();
        else {
            _descriptors.create((int)k.size(), 32, CV_8U);
            memcpy(_descriptors.getMat().data, d.data(), d.size());
        }
        return mono;
    }
    int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return mcorb_hamming256(a.data, b.data); }
#endif

    double max_neighbor_ratio = 0.85;   // ORBextractor.h:90

protected:
    mcorb_params p_;
    mcorb_t *h_ = nullptr;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// MultiCameraFrame.h:42-57, matchIndex widened from array<int,5> to MCORB_MAX_CAMS entries
class IntraMatch {
public:
    std::array<int, MCORB_MAX_CAMS> matchIndex;
    bool mono;
    int n_rays;
    IntraMatch() : mono(true), n_rays(0) { matchIndex.fill(-1); }
};

// ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (ORBVocabulary.h:21-30), the members this path uses
class ORBVocabulary {
public:
    typedef std::map<unsigned int, double> BowVector;                        // DBoW2::BowVector (WordId -> WordValue)
    typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector; // DBoW2::FeatureVector (NodeId -> feature indices)
    explicit ORBVocabulary(int device = 0) : device_(device) {}
    ~ORBVocabulary() { mcorb_vocab_destroy(v_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;
    // loadFromTextFile (FrontEnd.h:137-138): false when the file cannot be read or parsed
    bool loadFromTextFile(const std::string &filename)
    {
        mcorb_vocab_destroy(v_);
        v_ = nullptr;
        return mcorb_vocab_load_text(filename.c_str(), device_, &v_) == MCORB_OK;
    }
    // nodes 1..nnodes in file order: parent id, leaf flag, 32-byte descriptor, weight
    void create(int k, int L, int scoring, int weighting, const std::vector<int32_t> &parent, const std::vector<uint8_t> &is_leaf,
                const std::vector<uint8_t> &desc, const std::vector<double> &weight)
    {
        mcorb_vocab_destroy(v_);
        v_ = nullptr;
        check(mcorb_vocab_create(k, L, scoring, weighting, parent.data(), is_leaf.data(), desc.data(), weight.data(), (int)parent.size(),
                                 device_, &v_), "mcorb_vocab_create");
    }
    // transform(features, BowVector&, FeatureVector&, levelsup) on n packed 32-byte descriptors
    void transform(const uint8_t *desc, int n, BowVector &v, FeatureVector &fv, int levelsup) const
    {
        std::vector<uint32_t> ids(n + 1), nodes(n + 1);
        std::vector<double> vals(n + 1);
        std::vector<int32_t> offs(n + 2), feats(n + 1);
        int nb = 0, nf = 0;
        check(mcorb_vocab_transform(v_, desc, n, levelsup, ids.data(), vals.data(), n + 1, &nb, nodes.data(), offs.data(), n + 1, &nf,
                                    feats.data(), n + 1), "mcorb_vocab_transform");
        fill(ids, vals, nb, nodes, offs, feats, nf, v, fv);
    }
    mcorb_vocab *handle() const { return v_; }
    static void fill(const std::vector<uint32_t> &ids, const std::vector<double> &vals, int nb, const std::vector<uint32_t> &nodes,
                     const std::vector<int32_t> &offs, const std::vector<int32_t> &feats, int nf, BowVector &v, FeatureVector &fv)
    {
        v.clear(); fv.clear();
        for (int i = 0; i < nb; i++) v.emplace_hint(v.end(), ids[i], vals[i]);
        for (int i = 0; i < nf; i++)
            fv.emplace_hint(fv.end(), nodes[i], std::vector<unsigned int>(feats.begin() + offs[i], feats.begin() + offs[i + 1]));
    }

private:
    int device_;
    mcorb_vocab *v_ = nullptr;
};

// The extract + intra-rig-match members of MultiCameraFrame for one rig frame.
class MultiCameraFrontEnd {
public:
    MultiCameraFrontEnd(int num_cams, int width, int height, const mcorb_params &p) : num_cams_(num_cams), w_(width), h_(height)
    {
        check(mcorb_rig_create(&p, num_cams, width, height, 1, 1, &rig_), "mcorb_rig_create");
        image_kps.resize(num_cams);
        image_descriptors.resize(num_cams);
    }
    ~MultiCameraFrontEnd() { mcorb_rig_destroy(rig_); }

    // setData (MultiCameraFrame.cpp:95-152): 8-bit gray frames ...
    void setData(const std::vector<const uint8_t *> &imgs, int stride)
    {
        if ((int)imgs.size() != num_cams_) throw std::runtime_error("ERROR:: number of images is wrong");
        check(mcorb_rig_upload_u8(rig_, 0, imgs.data(), num_cams_, stride), "mcorb_rig_upload_u8");
    }
    // ... or the reader's CV_32F [0,1] frames (DatasetReader.cpp:709-712)
    void setDataF32(const std::vector<const float *> &imgs, int stride_bytes, int channels)
    {
        if ((int)imgs.size() != num_cams_) throw std::runtime_error("ERROR:: number of images is wrong");
        check(mcorb_rig_upload_f32(rig_, 0, imgs.data(), num_cams_, stride_bytes, channels), "mcorb_rig_upload_f32");
    }
    // extractFeaturesParallel (MultiCameraFrame.cpp:203-228)
    void extractFeaturesParallel()
    {
        check(mcorb_rig_extract(rig_, 0, num_cams_, 0, 0), "mcorb_rig_extract");
        const int cap = mcorb_rig_kcap(rig_);
        for (int c = 0; c < num_cams_; c++) {
            image_kps[c].resize(cap);
            image_descriptors[c].resize((size_t)cap * 32);
            int n = 0, mono = 0;
            check(mcorb_rig_get_features(rig_, 0, c, image_kps[c].data(), image_descriptors[c].data(), cap, &n, &mono),
                  "mcorb_rig_get_features");
            image_kps[c].resize(n);
            image_descriptors[c].resize((size_t)n * 32);
        }
        image_kps_undist.clear();   // belongs to the previous frame
        matched_ = false;
    }
    // BruteForceMatch (MultiCameraFrame.cpp:1024-1086), cam1 < cam2 as at every reference call site
    void BruteForceMatch(int img1_ind, int img2_ind, float dist_thresh, float neigh_ratio,
                         std::vector<unsigned int> &indices_1, std::vector<unsigned int> &indices_2,
                         std::vector<mcorb_keypoint> &kps1, std::vector<mcorb_keypoint> &kps2)
    {
        ensure_match(dist_thresh, neigh_ratio);
        const int cap = mcorb_rig_kcap(rig_);
        indices_1.resize(cap); indices_2.resize(cap);
        int n = 0;
        check(mcorb_rig_get_pair_matches(rig_, 0, 0, img1_ind, img2_ind, indices_1.data(), indices_2.data(), cap, &n),
              "mcorb_rig_get_pair_matches");
        indices_1.resize(n); indices_2.resize(n);
        kps1.clear(); kps2.clear();
        // the reference returns image_kps_undist (:1069-1076): the caller's undistorted set when one was given
        // (setUndistorted), else the extracted keypoints (RECTIFY, or zero distortion: UndistortKeyPoints copies them, :302-305)
        const std::vector<mcorb_keypoint> &u1 = undist_ok(img1_ind) ? image_kps_undist[img1_ind] : image_kps[img1_ind];
        const std::vector<mcorb_keypoint> &u2 = undist_ok(img2_ind) ? image_kps_undist[img2_ind] : image_kps[img2_ind];
        for (int k = 0; k < n; k++) {
            kps1.push_back(u1[indices_1[k]]);
            kps2.push_back(u2[indices_2[k]]);
        }
    }
    // The per-pair fundamental matrices the old=true branch builds from camconfig_ (MultiCameraFrame.cpp:1126-1142),
    // row-major 3x3 each, pairs in the order (0,1),(0,2)..: x_j^T F x_i = 0.
    void setFundamental(const std::vector<double> &F_pairs)
    {
        if ((int)F_pairs.size() != 9 * num_cams_ * (num_cams_ - 1) / 2) throw std::runtime_error("ERROR:: need one 3x3 per camera pair");
        F_pairs_ = F_pairs;
    }
#ifdef MCORB_WITH_OPENCV
    // the same from K_mats_/R_mats_/t_mats_ (CV_64F), with the cv::Mat expressions of the reference so that the
    // entries are the ones it computes
    void setCalibration(const std::vector<cv::Mat> &K_mats, const std::vector<cv::Mat> &R_mats, const std::vector<cv::Mat> &t_mats)
    {
        F_pairs_.clear();
        for (int i = 0; i < num_cams_ - 1; i++)
            for (int j = i + 1; j < num_cams_; j++) {
                cv::Mat Ti = cv::Mat::eye(4, 4, CV_64F), Tj = cv::Mat::eye(4, 4, CV_64F);
                R_mats[i].copyTo(Ti(cv::Range(0, 3), cv::Range(0, 3)));
                t_mats[i].copyTo(Ti(cv::Range(0, 3), cv::Range(3, 4)));
                R_mats[j].copyTo(Tj(cv::Range(0, 3), cv::Range(0, 3)));
                t_mats[j].copyTo(Tj(cv::Range(0, 3), cv::Range(3, 4)));
                cv::Mat Tji = Tj * Ti.inv();
                const double x = Tji.at<double>(0, 3), y = Tji.at<double>(1, 3), z = Tji.at<double>(2, 3);
                cv::Mat skew = (cv::Mat_<double>(3, 3) << 0, -z, y, z, 0, -x, -y, x, 0);
                cv::Mat F = K_mats[j].t().inv() * skew * Tji(cv::Range(0, 3), cv::Range(0, 3)) * K_mats[i].inv();
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) F_pairs_.push_back(F.at<double>(r, c));
            }
    }
#endif
    // computeIntraMatches(matches, words_) (MultiCameraFrame.cpp:586-943), the variant FrontEnd.cpp:1009 calls;
    // levelsup as in extractFeatureSingle's transform call (:257)
    void computeIntraMatches(std::vector<IntraMatch> &matches, std::vector<unsigned int> &words_, const ORBVocabulary &voc,
                             double max_neighbor_ratio = 0.85, int levelsup = 4)
    {
        const int cap = mcorb_rig_kcap(rig_) * num_cams_;
        std::vector<int32_t> tr((size_t)cap * num_cams_), rays(cap);
        std::vector<uint32_t> w(cap);
        int n = 0, nw = 0;
        // the |dy| < 50 gate reads image_kps_undist[cam][k].pt.y (:708-716)
        std::vector<std::vector<float>> yu(num_cams_);
        std::vector<const float *> yp(num_cams_, nullptr);
        for (int c = 0; c < num_cams_; c++)
            if (undist_ok(c)) {
                for (const mcorb_keypoint &k : image_kps_undist[c]) yu[c].push_back(k.y);
                yp[c] = yu[c].data();
            }
        bool any = false;
        for (int c = 0; c < num_cams_; c++) any = any || yp[c];
        if (any)
            for (int c = 0; c < num_cams_; c++)
                if (!yp[c]) { for (const mcorb_keypoint &k : image_kps[c]) yu[c].push_back(k.y); yp[c] = yu[c].data(); }
        check(mcorb_rig_match_bow_frames(rig_, 0, 0, 1, voc.handle(), levelsup, max_neighbor_ratio, any ? yp.data() : nullptr), "mcorb_rig_match_bow_frames");
        check(mcorb_rig_get_bow_tracks(rig_, 0, 0, tr.data(), rays.data(), cap, &n, w.data(), cap, &nw), "mcorb_rig_get_bow_tracks");
        matches.clear();
        matches.resize(n);
        for (int m = 0; m < n; m++) {
            for (int c = 0; c < num_cams_; c++) matches[m].matchIndex[c] = tr[(size_t)m * num_cams_ + c];
            matches[m].n_rays = rays[m];
        }
        words_.insert(words_.end(), w.begin(), w.begin() + nw);
    }
    // orb_vocabulary->transform(image_descriptors[cam], BoW_vecs[cam], BoW_feats[cam], levelsup) (MultiCameraFrame.cpp:257),
    // reading the descriptors where extraction left them on the device
    void transform(int cam, const ORBVocabulary &voc, ORBVocabulary::BowVector &bow, ORBVocabulary::FeatureVector &fv, int levelsup = 4)
    {
        const int n = (int)image_kps[cam].size();
        std::vector<uint32_t> ids(n + 1), nodes(n + 1);
        std::vector<double> vals(n + 1);
        std::vector<int32_t> offs(n + 2), feats(n + 1);
        int nb = 0, nf = 0;
        check(mcorb_rig_transform_image(rig_, 0, cam, voc.handle(), levelsup, ids.data(), vals.data(), n + 1, &nb, nodes.data(), offs.data(),
                                        n + 1, &nf, feats.data(), n + 1), "mcorb_rig_transform_image");
        ORBVocabulary::fill(ids, vals, nb, nodes, offs, feats, nf, bow, fv);
    }

    // image_kps_undist (MultiCameraFrame.cpp:241-245): what UndistortKeyPoints (:300-347) produced for the current frame.  Read by
    // BruteForceMatch's returned keypoints, the epipolar check and the BoW-guided matcher's row gate.  Not needed when RECTIFY is
    // on or the distortion is zero: the reference then copies image_kps, which is the default here.  Call after
    // extractFeaturesParallel(); it is dropped by the next extraction.
    void setUndistorted(const std::vector<std::vector<mcorb_keypoint>> &kps_undist) { image_kps_undist = kps_undist; }
    bool undist_ok(int c) const { return (int)image_kps_undist.size() == num_cams_ && image_kps_undist[c].size() == image_kps[c].size() && !image_kps[c].empty(); }

    // computeIntraMatches(matches, old) (MultiCameraFrame.cpp:1100-1288)
    void computeIntraMatches(std::vector<IntraMatch> &matches, bool old)
    {
        ensure_match(75, 0.85f);
        const int cap = mcorb_rig_kcap(rig_) * num_cams_;
        std::vector<int32_t> tr((size_t)cap * num_cams_);
        int n = 0;
        if (old) {
            if (F_pairs_.empty()) throw std::runtime_error("ERROR:: computeIntraMatches(old=true) needs setFundamental/setCalibration");
            std::vector<const mcorb_keypoint *> und;
            if ((int)image_kps_undist.size() == num_cams_)
                for (int c = 0; c < num_cams_; c++) {
                    if (image_kps_undist[c].size() != image_kps[c].size()) throw std::runtime_error("ERROR:: image_kps_undist size");
                    und.push_back(image_kps_undist[c].data());
                }
            check(mcorb_rig_get_tracks_epipolar(rig_, 0, 0, F_pairs_.data(), und.empty() ? nullptr : und.data(), tr.data(), cap, &n,
                                                &cnt_mergable_matches), "mcorb_rig_get_tracks_epipolar");
        } else
            check(mcorb_rig_get_tracks(rig_, 0, 0, tr.data(), cap, &n, &cnt_mergable_matches), "mcorb_rig_get_tracks");
        matches.clear();
        matches.resize(n);
        for (int m = 0; m < n; m++)
            for (int c = 0; c < num_cams_; c++) matches[m].matchIndex[c] = tr[(size_t)m * num_cams_ + c];
    }

    int num_cams_;
    std::vector<std::vector<mcorb_keypoint>> image_kps;
    std::vector<std::vector<uint8_t>> image_descriptors;   // per camera, n x 32
    std::vector<std::vector<mcorb_keypoint>> image_kps_undist;
    int cnt_mergable_matches = 0;

private:
    void ensure_match(float thr, float ratio)
    {
        if (matched_ && thr == thr_ && ratio == ratio_) return;
        check(mcorb_rig_match(rig_, 0, 1, thr, ratio), "mcorb_rig_match");
        matched_ = true; thr_ = thr; ratio_ = ratio;
    }
    mcorb_rig *rig_ = nullptr;
    int w_, h_;
    bool matched_ = false;
    float thr_ = 0, ratio_ = 0;
    std::vector<double> F_pairs_;
};

}  // namespace mcorb
