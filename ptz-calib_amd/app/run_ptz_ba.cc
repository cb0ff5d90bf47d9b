// run_ptz_ba -- PTZ-IBA + georeferencing from a directory of images, COLMAP-format features and matches, and an annotation
// file; writes <output>/<basename(images)>.json.  Same options, stages, messages and exit codes as the reference tool
// (src/app/run_ptz_ba.cc:24-154): 0 on success, -1 when a stage fails, 1 on bad options.  Every solve runs on the MI355X
// library through the C++ classes of ptz-calib_amd/host.
#include <cstdio>
#include <string>
#include <unordered_set>
#include <vector>

#include "../host/data_io.h"
#include "../host/ptz_incremental_optimizer.h"
#include "../host/ptzray_optimizer.h"
#include "args.h"

using namespace ptzcalib;

static bool RunPtzBA(const std::vector<std::string>& fnames, const std::vector<ImageFeatures>& features,
                     const std::vector<MatchesInfo>& matches_info, int max_iter, std::vector<Camera>& cameras,
                     std::unordered_set<long>& reg_image_ids)
{  // run_ptz_ba.cc:116-129
  cameras.clear();
  cameras.resize(fnames.size());
  PtzIncrementalOptimizer ptz_iba(features, matches_info, cameras, fnames, max_iter);
  reg_image_ids.clear();
  return ptz_iba.Solve(cameras, reg_image_ids);
}

static bool RunGeoreferencing(const std::vector<ImageFeatures>& features, const std::vector<MatchesInfo>& matches_info,
                              const std::vector<std::vector<Point2f>>& pixels, const std::vector<std::vector<Point3d>>& pts3d,
                              const std::unordered_set<long>& cam_ids, int max_iter, bool has_dist, std::vector<Camera>& cameras,
                              double& error_2d2d, double& error_2d3d)
{  // run_ptz_ba.cc:131-154
  PTZRayOptimizer optimizer(features, matches_info, cameras, pixels, pts3d, cam_ids, max_iter, has_dist ? PTZRayDist : PTZRay);
  std::vector<std::vector<Ray>> rays;
  if (!optimizer.Solve(cameras, rays)) {
    error_2d2d = error_2d3d = -1;
    return false;
  }
  error_2d2d = optimizer.final_reproj_error_2d2d();
  error_2d3d = optimizer.final_reproj_error_2d3d();
  return true;
}

int main(int argc, char** argv)
{
  ptzapp::Args parser;
  parser.Add("images", 'i', "Images directory", true);
  parser.Add("features", 'f', "Features and matches directory", true);
  parser.Add("annotation", 'a', "Annotation filepath", false);
  parser.Add("output", 'o', "Output directory", true);
  parser.AddFlag("dist", "Whether images have distortion");
  parser.ParseCheck(argc, argv);

  std::vector<std::string> fnames;
  std::vector<ImageFeatures> features;
  std::vector<Size> sizes;
  if (!LoadImgsAndFeatures(parser.Get("images"), parser.Get("features"), fnames, features, sizes)) {
    fprintf(stderr, "Error loading images and features. Exiting ...\n");
    return -1;
  }
  std::vector<MatchesInfo> matches_info;
  const std::string matches_path = parser.Get("features") + "/pairs_matches.txt";
  if (!LoadMatchesInfo(matches_path, fnames, features, matches_info)) {
    fprintf(stderr, "Error loading matches from %s. Exiting ...\n", matches_path.c_str());
    return -1;
  }
  fprintf(stderr, "================== PTZ-IBA Begin ==========================\n");
  std::vector<Camera> cameras;
  std::unordered_set<long> reg_image_ids;
  static const int MAX_ITER = 200;
  if (!RunPtzBA(fnames, features, matches_info, MAX_ITER, cameras, reg_image_ids)) {
    fprintf(stderr, "================== PTZ-IBA End: failed ==========================\n");
    return -1;
  }
  fprintf(stderr, "================== PTZ-IBA End: success ==========================\n");

  std::vector<std::vector<Point2f>> pixels;
  std::vector<std::vector<Point3d>> pts3d;
  if (!LoadAnnotation(parser.Get("annotation"), fnames, pixels, pts3d)) {
    fprintf(stderr, "Error loading annotation from %s. Exiting ...\n", parser.Get("annotation").c_str());
    return -1;
  }
  fprintf(stderr, "================== Georeferencing Begin ==========================\n");
  double error_2d2d, error_2d3d;
  if (!RunGeoreferencing(features, matches_info, pixels, pts3d, reg_image_ids, MAX_ITER, parser.Exist("dist"), cameras, error_2d2d, error_2d3d)) {
    fprintf(stderr, "================== Georeferencing End: failed ==========================\n");
    return -1;
  }
  fprintf(stderr, "================== Georeferencing End: success ==========================\n");

  const std::string cam_id = BaseName(parser.Get("images"));
  const std::string out_dir = parser.Get("output");
  MkdirIfNotExist(out_dir);
  const std::string out_path = out_dir + "/" + cam_id + ".json";
  SaveRegisteredCam(cameras, reg_image_ids, fnames, pixels, pts3d, out_path);

  fprintf(stderr, "================== Summary Begin ==========================\n");
  fprintf(stderr, "Registered/Total: %zu/%zu\n", reg_image_ids.size(), fnames.size());
  fprintf(stderr, "Error 2d-2d: %g\n", error_2d2d);
  fprintf(stderr, "Error 2d-3d: %g\n", error_2d3d);
  fprintf(stderr, "==================== Summary End ==========================\n");
  return 0;
}
