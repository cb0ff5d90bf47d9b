// homography.h -- pair homography with RANSAC, standing in for
//   cv::findHomography(ref_pts, src_pts, cv::RANSAC, ransac_thresh, mask)      (data_io.cc:340-355, threshold 4 px).
// OpenCV is not available to this build.  The estimator follows the same scheme -- 4-point minimal samples, forward
// transfer error against the threshold, adaptive iteration count at confidence 0.995 up to 2000 iterations, a least-squares
// fit on the inliers (normalised DLT) followed by a few Gauss-Newton steps on the transfer error, H scaled to h33 = 1 --
// with its own deterministic generator, so on outlier-free or clearly separated data it agrees with OpenCV's result to the
// fit's precision, but it is not bit-identical.
#pragma once

#include <vector>

#include "types.h"

namespace ptzcalib {

// dst ~ H src.  Returns false (H untouched) for fewer than 4 correspondences or when no model gathers 4 inliers --
// the cv::Mat::empty() case of the reference.  inlier_mask (optional) gets one byte per correspondence.
bool FindHomographyRansac(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, double ransac_thresh, Mat33& H,
                          std::vector<unsigned char>* inlier_mask = nullptr);

// Least-squares homography of all given correspondences (normalised DLT).
bool FitHomographyDLT(const std::vector<Point2f>& src, const std::vector<Point2f>& dst, const std::vector<int>& idx, Mat33& H);

}  // namespace ptzcalib
