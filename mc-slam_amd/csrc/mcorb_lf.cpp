// mcorb_lf.cpp -- FrontEnd::obtainLfFeatures (MCSlam/src/FrontEnd.cpp:213-593), SURVEY.md 8f row N3: the immediate
// consumer of the IntraMatch tracks.  Host code: per-track view filtering by the segmentation masks (:262-270), N-view
// triangulation (:280-307), the depth gate (:309), representative descriptor (:349, MultiCameraFrame.cpp:530-567), and the
// response-sorted mono fill to 3000 - intramatch_size (:478-523).  The tiling / refview branches are compile-time dead in the
// reference (`bool tiling = false; bool refview = false;`, :436-437) and are not restated.
//
// Integer / ordering work is exact: the same statements in the same order, and argsorte()'s std::sort (MCSlam/utils.h:21-30)
// is called on the same sequence with the same predicate (ties between equal responses land where libstdc++'s introsort
// puts them -- that IS the reference's order).
// Triangulation: cv::sfm::triangulatePoints is un-vendored third-party code (opencv_contrib / libmv): two views -> the 4x4 DLT
// design matrix, more views -> the 3n x (4+n) "x = alpha P X" design, null vector by SVD (cv::SVD::solveZ).  Restated here with
// the smallest eigenvector of the design's Gram matrix in FP64 (inverse iteration): the null vector of a
// full-column-rank-minus-one matrix is unique up to scale, so any stable method returns the same point to ~1e-11 relative;
// parity for this step is tolerance-based (1e-9) and UNPINNED.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <numeric>
#include <string>
#include <vector>

#include "mcorb_engine.h"

using namespace mcorb;

namespace {

// right singular vector of the smallest singular value of the m x n matrix A (row-major, m >= n, n <= 20)
// = eigenvector of the smallest eigenvalue of G = A^T A, found by inverse iteration: G + delta I is factored ONCE (LU with
// partial pivoting, n^3 / 3 multiplications), every further step is two triangular solves.  The design matrices of the
// triangulation have one singular value far below the rest (zero for exact correspondences): unshifted steps until the
// iterate has settled, then Rayleigh-quotient shifts (a fresh n^3 / 3 factorisation per step, nothing at n = 4 .. 8) until
// two successive iterates agree to 1e-15.  Accuracy: eps |G| / (lambda_2 - lambda_1) on the unit vector, 1e-12 .. 1e-11
// for this geometry (the one-sided Jacobi SVD it replaces worked on A itself, eps / sigma_2, at 50 x the cost: 60 sweeps
// of 6 .. 28 column pairs per track were 10 of the 13 ms a 4-camera frame took in round 2).  The test against LAPACK's SVD
// (600 cases of 2 .. 6 views, small and gross noise) holds its 1e-9.
constexpr int kLfMaxN = 4 + MCORB_MAX_CAMS, kLfMaxM = 3 * MCORB_MAX_CAMS;
// M, N > 0: compile-time shape (the 2-, 3- and 4-view designs are almost all the tracks of a 4-camera rig: with the loop
// bounds known the compiler unrolls and keeps G / LU in registers); M = N = 0: run-time shape
template <int M, int N>
inline void null_vector_t(const double *A, int m_rt, int n_rt, double *x)
{
    const int m = M > 0 ? M : m_rt, n = N > 0 ? N : n_rt;
    double G[kLfMaxN * kLfMaxN], LU[kLfMaxN * kLfMaxN];
    int piv[kLfMaxN];
    double tr = 0.0;
    for (int p = 0; p < n; p++)
        for (int q = p; q < n; q++) {
            double sacc = 0.0;
            for (int i = 0; i < m; i++) sacc += A[i * n + p] * A[i * n + q];
            G[p * n + q] = G[q * n + p] = sacc;
            if (p == q) tr += sacc;
        }
    if (!(tr > 0.0)) { for (int i = 0; i < n; i++) x[i] = i == n - 1 ? 1.0 : 0.0; return; }
    const double pivmin = 1e-30 * tr;
    // factor G - shift I (LU, partial pivoting); a vanishing pivot is nudged: the solve then blows up along the null vector,
    // which is the point of inverse iteration
    auto factor = [&](double shift) {
        for (int i = 0; i < n * n; i++) LU[i] = G[i];
        for (int i = 0; i < n; i++) LU[i * n + i] -= shift;
        for (int k = 0; k < n; k++) {
            int pk = k;
            for (int i = k + 1; i < n; i++) if (fabs(LU[i * n + k]) > fabs(LU[pk * n + k])) pk = i;
            piv[k] = pk;
            if (pk != k) for (int j = 0; j < n; j++) std::swap(LU[k * n + j], LU[pk * n + j]);
            double d = LU[k * n + k];
            // (a shift that IS an eigenvalue to the last bit can leave an exactly zero pivot; the nudge must keep the solve
            // finite -- 1 / 1e-300 squared overflows, and the normalisation then turns the iterate into NaNs)
            if (fabs(d) < pivmin) { d = d < 0 ? -pivmin : pivmin; LU[k * n + k] = d; }
            const double inv = 1.0 / d;
            for (int i = k + 1; i < n; i++) {
                const double f = LU[i * n + k] * inv;
                LU[i * n + k] = f;
                if (f != 0.0) for (int j = k + 1; j < n; j++) LU[i * n + j] -= f * LU[k * n + j];
            }
        }
    };
    double v[N > 0 ? N : kLfMaxN], y[N > 0 ? N : kLfMaxN];
    auto solve_step = [&]() -> double {   // v <- normalised (G - shift I)^-1 v, sign kept; returns the largest change of a component
        for (int i = 0; i < n; i++) y[i] = v[i];
        for (int k = 0; k < n; k++)            // P (all row exchanges first: the multipliers sit in their final rows)
            if (piv[k] != k) std::swap(y[k], y[piv[k]]);
        for (int k = 0; k < n; k++)            // L
            for (int i = k + 1; i < n; i++) y[i] -= LU[i * n + k] * y[k];
        for (int k = n - 1; k >= 0; k--) {     // U
            double t = y[k];
            for (int j = k + 1; j < n; j++) t -= LU[k * n + j] * y[j];
            y[k] = t / LU[k * n + k];
        }
        double nn = 0.0, dotp = 0.0;
        for (int i = 0; i < n; i++) nn += y[i] * y[i];
        nn = 1.0 / sqrt(nn);
        for (int i = 0; i < n; i++) { y[i] *= nn; dotp += y[i] * v[i]; }
        const double sgn = dotp < 0 ? -1.0 : 1.0;
        double diff = 0.0;
        for (int i = 0; i < n; i++) { y[i] *= sgn; diff = std::max(diff, fabs(y[i] - v[i])); v[i] = y[i]; }
        return diff;
    };
    auto rayleigh = [&]() {
        double r = 0.0;
        for (int p = 0; p < n; p++) {
            double gp = 0.0;
            for (int q = 0; q < n; q++) gp += G[p * n + q] * v[q];
            r += v[p] * gp;
        }
        return r;
    };
    for (int i = 0; i < n; i++) v[i] = 1.0 / sqrt((double)n) * (1.0 + 0.01 * i);   // any start with a component along the answer
    // Unshifted steps -- which can only converge to the eigenvector of the eigenvalue nearest zero, the smallest (G is positive
    // semi-definite) -- until the iterate has settled to 1e-3; only then Rayleigh-quotient shifts (cubic from there: two or
    // three steps where the unshifted iteration, contracting by lambda_1 / lambda_2 = 0.1 .. 0.5 on tracks with a wrong
    // correspondence, needs dozens).  Shifting earlier is not safe: a start vector that happens to be nearly orthogonal to the
    // answer still has its quotient near lambda_2 after a few steps, and the shifted iteration then converges THERE (seen on 2 %
    // of real tracks).  Should a shifted step move the iterate by more than 1e-2 it has left the basin: back to unshifted steps.
    factor(-1e-14 * tr);   // (keeps the factorisation away from an exactly singular matrix)
    double diff = 1.0;
    for (int it = 0; it < 400 && diff >= 1e-3; it++) diff = solve_step();
    bool shifted = false;
    for (int it = 0; it < 8 && diff >= 1e-15; it++) {
        factor(rayleigh());
        diff = solve_step();
        shifted = true;
        if (diff > 1e-2) break;
    }
    if (shifted) {
        // "settled" can also mean: sitting next to ANOTHER eigenvector with a tiny component along the wanted one (an unlucky
        // start); the shifted steps then polish that one.  Sylvester: G - (rho - tol) I has as many negative pivots in its LDL^T
        // as G has eigenvalues below rho - tol -- none when rho is the smallest.  Otherwise: unshifted iteration to the end.
        const double rho = rayleigh(), tol = 1e-10 * tr + 1e-3 * fabs(rho);
        bool smallest = diff <= 1e-2;   // (false for a NaN too)
        if (smallest) {
            for (int i = 0; i < n * n; i++) LU[i] = G[i];
            for (int i = 0; i < n; i++) LU[i * n + i] -= rho - tol;
            for (int k = 0; k < n && smallest; k++) {
                const double d = LU[k * n + k];
                if (!(d > 0.0)) { smallest = false; break; }
                for (int i = k + 1; i < n; i++) {
                    const double f = LU[i * n + k] / d;
                    for (int j = k + 1; j < n; j++) LU[i * n + j] -= f * LU[k * n + j];
                }
            }
        }
        if (!smallest) {
            for (int i = 0; i < n; i++) v[i] = 1.0 / sqrt((double)n) * (1.0 + 0.01 * i);
            factor(-1e-14 * tr);
            diff = 1.0;
            for (int k = 0; k < 2000 && diff >= 1e-15; k++) diff = solve_step();
        }
    }
    for (int i = 0; i < n; i++) x[i] = v[i];
}
void null_vector(const double *A, int m, int n, double *x)
{
    if (m == 4 && n == 4) null_vector_t<4, 4>(A, m, n, x);
    else if (m == 9 && n == 7) null_vector_t<9, 7>(A, m, n, x);
    else if (m == 12 && n == 8) null_vector_t<12, 8>(A, m, n, x);
    else null_vector_t<0, 0>(A, m, n, x);
}

// cv::sfm::triangulatePoints for one point seen in nv views: x = normalised image coordinates, P = 3x4 [R|t] (row-major)
void triangulate(const double *x, const double *const *P, int nv, double X[3])
{
    double h[4];
    if (nv == 2) {   // triangulateDLT
        double D[16];
        for (int i = 0; i < 4; i++) {
            D[0 * 4 + i] = x[0] * P[0][8 + i] - P[0][0 + i];
            D[1 * 4 + i] = x[1] * P[0][8 + i] - P[0][4 + i];
            D[2 * 4 + i] = x[2] * P[1][8 + i] - P[1][0 + i];
            D[3 * 4 + i] = x[3] * P[1][8 + i] - P[1][4 + i];
        }
        null_vector(D, 4, 4, h);
    } else {         // triangulateNViews: [-P_i | x_i in column 4+i] (X, alpha_1..alpha_n)^T = 0
        const int m = 3 * nv, n = 4 + nv;
        double D[kLfMaxM * kLfMaxN], sol[kLfMaxN];
        for (int i = 0; i < m * n; i++) D[i] = 0.0;
        for (int i = 0; i < nv; i++) {
            for (int jj = 0; jj < 3; jj++)
                for (int ii = 0; ii < 4; ii++) D[(size_t)(3 * i + jj) * n + ii] = -P[i][4 * jj + ii];
            D[(size_t)(3 * i + 0) * n + 4 + i] = x[2 * i];
            D[(size_t)(3 * i + 1) * n + 4 + i] = x[2 * i + 1];
            D[(size_t)(3 * i + 2) * n + 4 + i] = 1.0;
        }
        null_vector(D, m, n, sol);
        for (int i = 0; i < 4; i++) h[i] = sol[i];
    }
    for (int i = 0; i < 3; i++) X[i] = h[i] / h[3];   // homogeneousToEuclidean
}

}  // namespace

// host-stage hook for the CPU test-suite: the triangulation alone (x: nv normalised points, P: nv row-major 3x4 matrices)
extern "C" int mcorb_host_triangulate(const double *x, const double *P, int nv, double X[3])
{
    if (!x || !P || !X || nv < 2 || nv > MCORB_MAX_CAMS) return MCORB_E_ARG;
    const double *Pp[MCORB_MAX_CAMS];
    for (int i = 0; i < nv; i++) Pp[i] = P + 12 * i;
    triangulate(x, Pp, nv, X);
    return MCORB_OK;
}

// obtainLfFeatures of one frame of a slot.  parallel_tri: spread the triangulations over the worker pool (a single frame per
// call); the batched entry point runs whole frames as pool tasks instead and passes false.
static int lf_one_frame(Rig &R, Slot *s, int slot, int frame, const int32_t *tracks, int ntracks, const uint32_t *words,
                        const mcorb_camera *cams, const float *const *seg_masks, int seg_stride, const mcorb_keypoint *const *kps_undist,
                        int total_feats, mcorb_lf_feature *out, int cap, int *n_out, int *intramatch_size_out, int *mono_size_out,
                        uint32_t *words_fil, int cap_words, int *nwords_fil_out, bool parallel_tri)
{
    if (n_out) *n_out = 0;
    if (intramatch_size_out) *intramatch_size_out = 0;
    if (mono_size_out) *mono_size_out = 0;
    if (nwords_fil_out) *nwords_fil_out = 0;
    if (ntracks < 0 || (ntracks && !tracks) || !cams || !out || cap < 0) {
        set_error("obtain_lf_features: bad argument");
        return MCORB_E_ARG;
    }
    const int C = R.ncams, kcap = R.geom.kcap;
    static const bool prof = getenv("MCORB_HOST_PROF") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto T0 = now();
    if (frame < 0 || (frame + 1) * C > s->nimg_done) { set_error("obtain_lf_features: frame not extracted"); return MCORB_E_STATE; }
    const int m0 = frame * C;
    auto KP = [&](int c) -> const std::vector<mcorb_keypoint> & { return s->kps[m0 + c]; };
    auto KPU = [&](int c, int k) -> const mcorb_keypoint & { return kps_undist && kps_undist[c] ? kps_undist[c][k] : s->kps[m0 + c][k]; };
    auto DESC = [&](int c, int k) -> const uint8_t * { return s->h_desc + ((size_t)(m0 + c) * kcap + k) * 32; };
    auto seg = [&](int c, float px, float py) -> float {   // segMasks[i].at<float>(p.y, p.x): float -> int conversion truncates
        if (!seg_masks || !seg_masks[c]) return 0.f;
        return seg_masks[c][(size_t)(int)py * seg_stride + (int)px];
    };
    for (int t = 0; t < ntracks; t++)
        for (int c = 0; c < C; c++) {
            const int k = tracks[(size_t)t * C + c];
            if (k < -1 || k >= (int)KP(c).size()) { set_error("obtain_lf_features: track index out of range"); return MCORB_E_ARG; }
        }
    std::vector<std::vector<uint8_t>> keypoint_mask(C);          // (:221-227)
    for (int c = 0; c < C; c++) keypoint_mask[c].assign(KP(c).size(), 1);
    std::vector<const double *> prj(C);
    for (int c = 0; c < C; c++) prj[c] = cams[c].Rt;             // build_Rt(R, t) (:224)

    // mono candidates are kept as (camera, keypoint) only: every one of them becomes the same kind of feature (:396-412 and
    // :489-512 fill the same fields), and only the best total_feats - intramatch_size are materialised after the sort
    struct MonoRef { int cam, kp; };
    std::vector<mcorb_lf_feature> intra;
    std::vector<MonoRef> mono_keypoints;
    std::vector<float> responses;
    std::vector<uint32_t> wfil;
    size_t nkp_total = 0;
    for (int c = 0; c < C; c++) nkp_total += KP(c).size();
    intra.reserve((size_t)std::max(total_feats, ntracks) + 8); mono_keypoints.reserve(nkp_total + 8); responses.reserve(nkp_total + 8);
    int intramatch_size = 0, mono_size = 0;
    auto blank = [&]() {
        mcorb_lf_feature f;
        memset(&f, 0, sizeof(f));
        for (int c = 0; c < MCORB_MAX_CAMS; c++) f.match_index[c] = -1;
        return f;
    };

    // The triangulation of a track depends on nothing but the track: all of them are done up front on the worker pool
    // (a null vector by Jacobi sweeps per track was 10 of the 13 ms this call took for 2 100 tracks); the bookkeeping below
    // then walks the tracks in order, as the reference does.
    struct Tri { double X[3]; };
    std::vector<Tri> tri((size_t)ntracks);
    {
        constexpr int kChunk = 64;
        const std::function<void(int, int)> tri_task = [&](int chunk, int) {
            for (int ind = chunk * kChunk; ind < std::min(ntracks, (chunk + 1) * kChunk); ind++) {
                int views[MCORB_MAX_CAMS], nv = 0;
                for (int i = 0; i < C; i++) {
                    const int k = tracks[(size_t)ind * C + i];
                    if (k != -1 && (double)seg(i, KP(i)[k].x, KP(i)[k].y) < 0.7) views[nv++] = i;
                }
                if (nv < 2) continue;
                double xx[2 * MCORB_MAX_CAMS];
                const double *PJs[MCORB_MAX_CAMS];
                for (int ii = 0; ii < nv; ii++) {
                    const int v = views[ii];
                    const mcorb_keypoint &kp = KP(v)[tracks[(size_t)ind * C + v]];
                    PJs[ii] = prj[v];
                    xx[2 * ii] = ((double)kp.x - cams[v].K[2]) / cams[v].K[0];          // (pt.x - cx) / fx (:291-293)
                    xx[2 * ii + 1] = ((double)kp.y - cams[v].K[5]) / cams[v].K[4];
                }
                triangulate(xx, PJs, nv, tri[ind].X);
            }
        };
        const int nchunks = (ntracks + kChunk - 1) / kChunk;
        if (nchunks > 1 && parallel_tri) R.pool->parallel_for(nchunks, tri_task, R.pool_threads + slot);
        else for (int ch = 0; ch < nchunks; ch++) tri_task(ch, 0);
    }

    const auto T1 = now();
    for (int ind = 0; ind < ntracks; ind++) {                    // (:250-414)
        mcorb_lf_feature temp = blank();
        for (int c = 0; c < C; c++) temp.match_index[c] = tracks[(size_t)ind * C + c];
        int view_inds[MCORB_MAX_CAMS], num_views = 0;
        const uint8_t *descs[MCORB_MAX_CAMS];
        for (int i = 0; i < C; i++) {
            const int feat_ind = temp.match_index[i];
            if (feat_ind != -1) {
                const mcorb_keypoint &kp = KP(i)[feat_ind];
                // `segMasks[i].at<float>(p.y, p.x) < 0.7` (:264): the float is promoted and compared with the DOUBLE 0.7, so a
                // mask value of exactly 0.7f (0.699999988) still counts as static
                if ((double)seg(i, kp.x, kp.y) < 0.7) {
                    descs[num_views] = DESC(i, feat_ind);
                    view_inds[num_views++] = i;
                } else {
                    temp.match_index[i] = -1;
                }
            }
        }
        if (num_views > 1) {
            const double *X = tri[ind].X;                        // cv::sfm::triangulatePoints of the views kept above (:291-306)
            if (X[2] < 40 && X[2] > 0.5) {                       // (:309)
                // K_mats_[0] * pt3d (:339-341): the point is taken in the reference camera's frame
                const double *K0 = cams[0].K;
                const double px = K0[0] * X[0] + K0[1] * X[1] + K0[2] * X[2], py = K0[3] * X[0] + K0[4] * X[1] + K0[5] * X[2],
                             pz = K0[6] * X[0] + K0[7] * X[1] + K0[8] * X[2];
                if (words) wfil.push_back(words[ind]);
                uint8_t packed[MCORB_MAX_CAMS * 32];
                for (int ii = 0; ii < num_views; ii++) memcpy(packed + 32 * ii, descs[ii], 32);
                const int rep = mcorb_representative_desc(packed, num_views);   // computeRepresentativeDesc (:349)
                memcpy(temp.desc, descs[rep], 32);
                temp.point3d[0] = X[0]; temp.point3d[1] = X[1]; temp.point3d[2] = X[2];
                temp.uv_ref[0] = (float)(px / pz);               // cv::Point2f(expected_x, expected_y)
                temp.uv_ref[1] = (float)(py / pz);
                temp.mono = 0;
                temp.n_rays = num_views;
                intra.push_back(temp);
                intramatch_size++;
                for (int ii = 0; ii < num_views; ii++) keypoint_mask[view_inds[ii]][temp.match_index[view_inds[ii]]] = 0;
            }
        } else if (num_views == 1) {                             // (:396-412)
            const int v = view_inds[0], k = temp.match_index[v];
            mono_keypoints.push_back(MonoRef{v, k});
            responses.push_back(KPU(v, k).response);
            keypoint_mask[v][k] = 0;
        }
    }
    const auto T2 = now();
    // every keypoint no track used becomes a mono candidate, camera by camera (:489-512); the segmentation test is commented
    // out in this branch of the reference
    for (int i = 0; i < C; i++)
        for (int j = 0; j < (int)KP(i).size(); j++)
            if (keypoint_mask[i][j]) {
                mono_keypoints.push_back(MonoRef{i, j});
                responses.push_back(KPU(i, j).response);
                keypoint_mask[i][j] = 0;
            }
    const auto T3 = now();
    // argsorte(responses, false) (MCSlam/utils.h:21-30): std::sort of the index sequence, descending response
    std::vector<int> sorted((int)responses.size());
    std::iota(sorted.begin(), sorted.end(), 0);
    std::sort(sorted.begin(), sorted.end(), [&responses](int i, int j) -> bool { return responses[i] > responses[j]; });
    for (int i = 0; i < (int)sorted.size() && i < (total_feats - intramatch_size); ++i) {   // (:515-521)
        const MonoRef &mr = mono_keypoints[sorted[i]];
        mcorb_lf_feature f = blank();
        f.match_index[mr.cam] = mr.kp;
        memcpy(f.desc, DESC(mr.cam, mr.kp), 32);
        f.uv_ref[0] = KPU(mr.cam, mr.kp).x; f.uv_ref[1] = KPU(mr.cam, mr.kp).y;
        f.n_rays = 1;
        f.mono = 1;
        intra.push_back(f);
        mono_size++;
    }
    const auto T4 = now();
    // words_fil is a std::set: ascending, unique
    std::sort(wfil.begin(), wfil.end());
    wfil.erase(std::unique(wfil.begin(), wfil.end()), wfil.end());

    if (n_out) *n_out = (int)intra.size();
    if (intramatch_size_out) *intramatch_size_out = intramatch_size;
    if (mono_size_out) *mono_size_out = mono_size;
    if (nwords_fil_out) *nwords_fil_out = (int)wfil.size();
    if ((int)intra.size() > cap || (words_fil && (int)wfil.size() > cap_words)) { set_error("obtain_lf_features: output too small"); return MCORB_E_CAP; }
    if (!intra.empty()) memcpy(out, intra.data(), intra.size() * sizeof(mcorb_lf_feature));
    if (words_fil && !wfil.empty()) memcpy(words_fil, wfil.data(), wfil.size() * sizeof(uint32_t));
    if (prof && frame == 0)
        fprintf(stderr, "[mcorb host prof] obtain_lf_features frame 0: setup + triangulation %.0f us, track bookkeeping %.0f, mono pool %.0f, sort + fill %.0f, copy out %.0f\n",
                us(T0, T1), us(T1, T2), us(T2, T3), us(T3, T4), us(T4, now()));
    return MCORB_OK;
}

extern "C" int mcorb_rig_obtain_lf_features(mcorb_rig *r, int slot, int frame, const int32_t *tracks, int ntracks,
                                            const uint32_t *words, const mcorb_camera *cams, const float *const *seg_masks,
                                            int seg_stride, const mcorb_keypoint *const *kps_undist, int total_feats,
                                            mcorb_lf_feature *out, int cap, int *n_out, int *intramatch_size_out,
                                            int *mono_size_out, uint32_t *words_fil, int cap_words, int *nwords_fil_out)
{
    if (n_out) *n_out = 0;
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size()) { set_error("obtain_lf_features: bad argument"); return MCORB_E_ARG; }
    Rig &R = r->rig;
    Slot *s = R.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    return lf_one_frame(R, s, slot, frame, tracks, ntracks, words, cams, seg_masks, seg_stride, kps_undist, total_feats, out, cap, n_out,
                        intramatch_size_out, mono_size_out, words_fil, cap_words, nwords_fil_out, true);
}

// All frames [frame0, frame0 + nframes) of a slot in one call, one worker-pool task per frame (the per-frame call is 70 x the
// 0.03 ms per frame of the extraction that feeds it; FrontEnd.cpp:1024 calls obtainLfFeatures once per frame right after
// computeIntraMatches).  tracks / words: the frames' arrays back to back, ntracks[f] tracks (ncams ints each) and words per
// frame; seg_masks / kps_undist: nframes * ncams pointers (index f * ncams + cam) or NULL; out: nframes blocks of `cap` entries,
// words_fil: nframes blocks of cap_words; the four count arrays have nframes entries.  Returns the first failing frame's status.
extern "C" int mcorb_rig_obtain_lf_features_frames(mcorb_rig *r, int slot, int frame0, int nframes, const int32_t *tracks,
                                                   const int32_t *ntracks, const uint32_t *words, const mcorb_camera *cams,
                                                   const float *const *seg_masks, int seg_stride,
                                                   const mcorb_keypoint *const *kps_undist, int total_feats, mcorb_lf_feature *out,
                                                   int cap, int *n_out, int *intramatch_size_out, int *mono_size_out,
                                                   uint32_t *words_fil, int cap_words, int *nwords_fil_out)
{
    if (!r || slot < 0 || slot >= (int)r->rig.slots.size() || nframes < 1 || frame0 < 0 || !ntracks || !n_out || !intramatch_size_out ||
        !mono_size_out || !out || cap < 0 || (words_fil && !nwords_fil_out)) {
        set_error("obtain_lf_features_frames: bad argument");
        return MCORB_E_ARG;
    }
    Rig &R = r->rig;
    Slot *s = R.slots[slot];
    {
        std::lock_guard<std::mutex> lk(s->m);
        if (s->busy) { set_error("slot busy"); return MCORB_E_STATE; }
    }
    const int C = R.ncams;
    std::vector<size_t> off((size_t)nframes + 1, 0);
    for (int f = 0; f < nframes; f++) {
        if (ntracks[f] < 0) { set_error("obtain_lf_features_frames: negative track count"); return MCORB_E_ARG; }
        off[f + 1] = off[f] + (size_t)ntracks[f];
    }
    std::vector<int> status((size_t)nframes, MCORB_OK);
    std::vector<std::string> errs((size_t)nframes);
    R.pool->parallel_for(nframes, [&](int f, int) {
        int nw = 0;
        status[f] = lf_one_frame(R, s, slot, frame0 + f, tracks ? tracks + off[f] * C : nullptr, ntracks[f], words ? words + off[f] : nullptr, cams,
                                 seg_masks ? seg_masks + (size_t)f * C : nullptr, seg_stride, kps_undist ? kps_undist + (size_t)f * C : nullptr,
                                 total_feats, out + (size_t)f * cap, cap, &n_out[f], &intramatch_size_out[f], &mono_size_out[f],
                                 words_fil ? words_fil + (size_t)f * cap_words : nullptr, cap_words, &nw, false);
        if (nwords_fil_out) nwords_fil_out[f] = nw;
        if (status[f] != MCORB_OK) errs[f] = get_error();   // (the error text is per thread)
    }, R.pool_threads + slot);
    for (int f = 0; f < nframes; f++)
        if (status[f] != MCORB_OK) { set_error("frame " + std::to_string(frame0 + f) + ": " + errs[f]); return status[f]; }
    return MCORB_OK;
}
