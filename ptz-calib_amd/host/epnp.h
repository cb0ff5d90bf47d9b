// epnp.h -- perspective-n-point initialisation used by the georeferencing stage.
// Replaces the reference's call cv::solvePnP(pts3d, pixels, K, dist, rvec, tvec, false, cv::SOLVEPNP_EPNP)
// (src/core/ptzray_optimizer.cc:572).  OpenCV is not available to this build, so this is the published EPnP
// algorithm (Lepetit, Moreno-Noguer, Fua, IJCV 2009: four virtual control points, null space of the 2n x 12
// system, three beta approximations + Gauss-Newton, absolute orientation) written from the paper, plus a
// plane-induced-homography branch for coplanar points (pitch markings), where the 4-control-point form is
// rank deficient, and a damped Gauss-Newton polish of the pose.  It is an INITIALISER: its output is gated (ptzray_optimizer.cc:583,602) and then refined by
// the bundle adjustment, so it is held to "recovers the pose", not to bit parity with OpenCV's implementation.
#pragma once

#include <vector>

#include "types.h"

namespace ptzcalib {

// World -> camera pose (X_c = R X_w + t) from >= 4 correspondences.  Pixels are undistorted with (K, dist) the way
// cv::undistortPoints does (5 fixed-point iterations, the reference's (k1,k2,k3,p1,p2) vector read in OpenCV's
// (k1,k2,p1,p2,k3) order) and the problem is solved in normalised coordinates.
// Returns false for fewer than 4 points, mismatched sizes or a degenerate (collinear) configuration.
bool SolvePnPEPnP(const std::vector<Point3d>& pts3d, const std::vector<Point2f>& pixels, const Mat33& K, const Vec5& dist,
                  Mat33& R, Vec3& t);

}  // namespace ptzcalib
