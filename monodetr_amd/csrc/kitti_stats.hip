// monodetr_amd/csrc/kitti_stats.hip -- HOST code (no kernel): the sequential heart of the KITTI evaluation, one call per
// (class, difficulty, metric, overlap threshold).
//
// Reference: compute_statistics_jit / fused_compute_statistics / get_thresholds, numba-jitted CPU functions
// (lib/datasets/kitti/kitti_eval_python/eval.py:231-401, :9-27), driven from a Python loop over frames and over the 41
// recall thresholds (eval_class, :563-620).  The greedy detection-to-ground-truth assignment is inherently serial per
// frame and tiny (tens of boxes); what costs time in Python is the ~3 M calls of a full validation run, so the whole
// inner loop -- first pass collecting true-positive scores, recall thresholds, second pass accumulating
// tp / fp / fn / orientation similarity per threshold -- is one native call here.  The overlaps come from
// rotate_iou.hip (device) or the axis-aligned image overlap (host, numpy).
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/monodetr_amd.h"

namespace {

struct Frame {
    const double *ov;        // [nd, ng] row-major: overlap of detection j with ground truth i at ov[j * ng + i]
    const double *gt, *dt;   // [ng, 5] (bbox, alpha), [nd, 6] (bbox, alpha, score)
    const int64_t *ig, *id;  // ignore flags: 0 = counts, 1 = neutral, -1 = other class
    const double *dc;        // [ndc, 4] don't-care boxes
    int ng, nd, ndc;
};

struct Counts { int64_t tp, fp, fn; double similarity; };

constexpr double kNone = -10000000.0;

// eval.py:231-345.  `tp_scores` (first pass only) receives the score of every true positive.
Counts frame_statistics(const Frame &f, int metric, double min_overlap, double thresh, bool compute_fp, bool compute_aos,
                        std::vector<double> *tp_scores)
{
    std::vector<char> assigned(f.nd, 0), below(f.nd, 0);
    if (compute_fp)
        for (int j = 0; j < f.nd; ++j) below[j] = f.dt[j * 6 + 5] < thresh;
    Counts c{0, 0, 0, 0.0};
    std::vector<double> delta;
    for (int i = 0; i < f.ng; ++i) {
        if (f.ig[i] == -1) continue;
        int det = -1;
        double valid = kNone, max_overlap = 0.0;
        bool took_ignored = false;
        for (int j = 0; j < f.nd; ++j) {
            if (f.id[j] == -1 || assigned[j] || below[j]) continue;
            const double ov = f.ov[static_cast<int64_t>(j) * f.ng + i], score = f.dt[j * 6 + 5];
            if (!compute_fp && ov > min_overlap && score > valid) {
                det = j;
                valid = score;
            } else if (compute_fp && ov > min_overlap && (ov > max_overlap || took_ignored) && f.id[j] == 0) {
                max_overlap = ov;
                det = j;
                valid = 1.0;
                took_ignored = false;
            } else if (compute_fp && ov > min_overlap && valid == kNone && f.id[j] == 1) {
                det = j;
                valid = 1.0;
                took_ignored = true;
            }
        }
        if (valid == kNone && f.ig[i] == 0) {
            ++c.fn;
        } else if (valid != kNone && (f.ig[i] == 1 || f.id[det] == 1)) {
            assigned[det] = 1;
        } else if (valid != kNone) {
            ++c.tp;
            if (tp_scores) tp_scores->push_back(f.dt[det * 6 + 5]);
            if (compute_aos) delta.push_back(f.gt[i * 5 + 4] - f.dt[det * 6 + 4]);
            assigned[det] = 1;
        }
    }
    if (compute_fp) {
        for (int j = 0; j < f.nd; ++j)
            if (!(assigned[j] || f.id[j] == -1 || f.id[j] == 1 || below[j])) ++c.fp;
        int64_t stuff = 0;
        if (metric == 0) {                              // detections inside don't-care regions are not false positives
            for (int i = 0; i < f.ndc; ++i) {
                const double *q = f.dc + i * 4;
                for (int j = 0; j < f.nd; ++j) {
                    if (assigned[j] || f.id[j] == -1 || f.id[j] == 1 || below[j]) continue;
                    const double *b = f.dt + j * 6;
                    const double iw = std::min(b[2], q[2]) - std::max(b[0], q[0]);
                    double ov = 0.0;
                    if (iw > 0) {
                        const double ih = std::min(b[3], q[3]) - std::max(b[1], q[1]);
                        if (ih > 0) ov = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));      // image_box_overlap, criterion 0
                    }
                    if (ov > min_overlap) {
                        assigned[j] = 1;
                        ++stuff;
                    }
                }
            }
        }
        c.fp -= stuff;
        if (compute_aos) {
            if (c.tp > 0 || c.fp > 0) {
                double s = 0.0;
                for (double d : delta) s += (1.0 + cos(d)) / 2.0;
                c.similarity = s;
            } else {
                c.similarity = -1.0;
            }
        }
    }
    return c;
}

// eval.py:9-27
std::vector<double> recall_thresholds(std::vector<double> scores, int64_t num_gt, int num_sample_pts)
{
    std::sort(scores.begin(), scores.end(), [](double a, double b) { return a > b; });
    std::vector<double> out;
    double current = 0.0;
    const int64_t n = static_cast<int64_t>(scores.size());
    for (int64_t i = 0; i < n; ++i) {
        const double l_recall = static_cast<double>(i + 1) / static_cast<double>(num_gt);
        const double r_recall = i < n - 1 ? static_cast<double>(i + 2) / static_cast<double>(num_gt) : l_recall;
        if ((r_recall - current) < (current - l_recall) && i < n - 1) continue;
        out.push_back(scores[i]);
        current += 1.0 / (num_sample_pts - 1.0);
    }
    return out;
}

}  // namespace

extern "C" int mdetr_kitti_pr_curve(const double *overlaps, const int64_t *ov_start, const double *gt_datas,
                                    const double *dt_datas, const int64_t *gt_start, const int64_t *dt_start,
                                    const int64_t *ignored_gt, const int64_t *ignored_det, const double *dontcares,
                                    const int64_t *dc_start, int n_frames, int metric, double min_overlap,
                                    int64_t num_valid_gt, int compute_aos, int max_thresholds, double *pr,
                                    double *thresholds, int *n_thresholds)
{
    if (n_frames < 0 || !ov_start || !gt_start || !dt_start || !dc_start || !pr || !thresholds || !n_thresholds || max_thresholds < 1)
        return MDETR_E_ARG;
    std::vector<Frame> frames(n_frames);
    for (int f = 0; f < n_frames; ++f) {
        Frame &fr = frames[f];
        fr.ng = static_cast<int>(gt_start[f + 1] - gt_start[f]);
        fr.nd = static_cast<int>(dt_start[f + 1] - dt_start[f]);
        fr.ndc = static_cast<int>(dc_start[f + 1] - dc_start[f]);
        fr.ov = overlaps + ov_start[f];
        fr.gt = gt_datas + gt_start[f] * 5;
        fr.dt = dt_datas + dt_start[f] * 6;
        fr.ig = ignored_gt + gt_start[f];
        fr.id = ignored_det + dt_start[f];
        fr.dc = dontcares + dc_start[f] * 4;
    }
    std::vector<double> scores;
    for (const Frame &fr : frames) frame_statistics(fr, metric, min_overlap, 0.0, false, false, &scores);
    const std::vector<double> th = recall_thresholds(scores, num_valid_gt, 41);
    if (static_cast<int>(th.size()) > max_thresholds) return MDETR_E_ARG;
    *n_thresholds = static_cast<int>(th.size());
    for (size_t t = 0; t < th.size(); ++t) {
        thresholds[t] = th[t];
        pr[4 * t] = pr[4 * t + 1] = pr[4 * t + 2] = pr[4 * t + 3] = 0.0;
    }
    for (const Frame &fr : frames)
        for (size_t t = 0; t < th.size(); ++t) {
            const Counts c = frame_statistics(fr, metric, min_overlap, th[t], true, compute_aos != 0, nullptr);
            pr[4 * t] += static_cast<double>(c.tp);
            pr[4 * t + 1] += static_cast<double>(c.fp);
            pr[4 * t + 2] += static_cast<double>(c.fn);
            if (c.similarity != -1.0) pr[4 * t + 3] += c.similarity;
        }
    return MDETR_OK;
}
