// run_ptz_reloc -- relocalise test images against calibrated reference images; same options, inputs and output file as the
// reference tool (src/app/run_ptz_reloc.cc:23-148).  Where the reference runs one KRTOptimizer per test image in a loop
// (:68-118), this tool gathers every test image's problem and solves them all in ONE ptz_krt_solve_batch launch.
#include <cstdio>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/ptz_calib_amd.h"
#include "../host/data_io.h"
#include "args.h"

using namespace ptzcalib;

typedef std::pair<std::string, std::vector<DMatch>> BestMatchT;

// the reference image with the most matches towards this test image; the first one wins a tie (run_ptz_reloc.cc:150-170)
static BestMatchT FindBestMatch(const std::string& fname, const std::vector<std::pair<std::string, std::string>>& img_pairs_name,
                                const std::vector<std::vector<DMatch>>& pairs_matches)
{
  BestMatchT best;
  for (size_t i = 0; i < img_pairs_name.size(); ++i) {
    if (img_pairs_name[i].second != fname) continue;
    if (pairs_matches[i].size() > best.second.size()) best = {img_pairs_name[i].first, pairs_matches[i]};
  }
  return best;
}

int main(int argc, char** argv)
{
  ptzapp::Args parser;
  parser.Add("ref_images", '\0', "Reference images directory", true);
  parser.Add("ref_features", '\0', "Reference images features directory", true);
  parser.Add("ref_params", '\0', "Reference camera parameters filepath", true);
  parser.Add("test_images", '\0', "Test images directory", true);
  parser.Add("test_features", '\0', "Test images features and matches directory", true);
  parser.Add("output", '\0', "Output directory", true);
  parser.AddFlag("dist", "Whether images have distortion");
  parser.ParseCheck(argc, argv);

  std::vector<std::string> ref_fnames, test_fnames;
  std::vector<ImageFeatures> ref_features, test_features;
  std::vector<Size> ref_sizes, test_sizes;
  if (!LoadImgsAndFeatures(parser.Get("ref_images"), parser.Get("ref_features"), ref_fnames, ref_features, ref_sizes)) {
    fprintf(stderr, "Error loading reference images and features. Exiting ...\n");
    return -1;
  }
  if (!LoadImgsAndFeatures(parser.Get("test_images"), parser.Get("test_features"), test_fnames, test_features, test_sizes)) {
    fprintf(stderr, "Error loading test images and features. Exiting ...\n");
    return -1;
  }
  std::vector<std::vector<DMatch>> pairs_matches;
  std::vector<std::pair<std::string, std::string>> img_pairs_name;
  ReadColmapMatches(parser.Get("test_features") + "/pairs_matches.txt", pairs_matches, img_pairs_name);
  std::vector<Camera> ref_cameras;
  if (!ReadCamFromJson(parser.Get("ref_params"), ref_fnames, ref_cameras)) {
    fprintf(stderr, "Error loading reference camera parameters. Exiting ...\n");
    return -1;
  }

  // gather one query per test image that has a usable reference (run_ptz_reloc.cc:68-104)
  std::vector<size_t> query_image;
  std::vector<int64_t> match_ptr{0};
  std::vector<float> uv_ref, uv_cur;
  std::vector<double> cam_ref, cam_cur;
  for (size_t test_idx = 0; test_idx < test_fnames.size(); ++test_idx) {
    const BestMatchT best = FindBestMatch(test_fnames[test_idx], img_pairs_name, pairs_matches);
    const long ref_idx = FindImgIndex(ref_fnames, best.first);
    bool usable = ref_idx != -1 && !best.second.empty();
    if (usable)
      for (const DMatch& m : best.second)
        usable &= m.queryIdx >= 0 && m.trainIdx >= 0 && static_cast<size_t>(m.queryIdx) < ref_features[ref_idx].keypoints.size() &&
                  static_cast<size_t>(m.trainIdx) < test_features[test_idx].keypoints.size();
    if (!usable) {
      fprintf(stderr, "Running ptz-reloc failed: %s\n", test_fnames[test_idx].c_str());
      continue;
    }
    const Camera& ref_cam = ref_cameras[ref_idx];
    const double f = ref_cam.K()[0];
    const double cx = 0.5 * test_sizes[test_idx].width, cy = 0.5 * test_sizes[test_idx].height;
    const Camera init(Mat33{f, 0, cx, 0, f, cy, 0, 0, 1}, ref_cam.R(), ref_cam.t(), ref_cam.dist());
    const std::vector<double> vr = ref_cam.ToVector(), vc = init.ToVector();
    cam_ref.insert(cam_ref.end(), vr.begin(), vr.end());
    cam_cur.insert(cam_cur.end(), vc.begin(), vc.end());
    for (const DMatch& m : best.second) {
      const Point2f a = ref_features[ref_idx].keypoints[m.queryIdx].pt, b = test_features[test_idx].keypoints[m.trainIdx].pt;
      uv_ref.push_back(a.x); uv_ref.push_back(a.y);
      uv_cur.push_back(b.x); uv_cur.push_back(b.y);
    }
    match_ptr.push_back(static_cast<int64_t>(uv_ref.size() / 2));
    query_image.push_back(test_idx);
  }

  std::vector<Camera> test_cameras(test_fnames.size());
  std::unordered_set<long> success_ids;
  if (!query_image.empty()) {
    static const int MAX_ITER = 200;
    static const double MAX_REPROJ_ERROR = 100.0;
    ptz_lm_options opt;
    ptz_lm_options_default(&opt);
    opt.max_num_iterations = MAX_ITER;
    const int32_t nq = static_cast<int32_t>(query_image.size());
    std::vector<ptz_lm_summary> summaries(nq);
    std::vector<int32_t> accepted(nq, 0);
    const int32_t rc = ptz_krt_solve_batch(nq, match_ptr.data(), uv_ref.data(), uv_cur.data(), cam_ref.data(), cam_cur.data(),
                                           parser.Exist("dist") ? PTZ_KRT_FDist : PTZ_KRT_F, MAX_REPROJ_ERROR, &opt, summaries.data(),
                                           accepted.data(), nullptr);
    if (rc != PTZ_OK) {
      fprintf(stderr, "ptz_krt_solve_batch failed with status %d (no usable HIP device?)\n", rc);
      return -1;
    }
    for (int32_t q = 0; q < nq; ++q) {
      const size_t test_idx = query_image[q];
      if (accepted[q]) {
        test_cameras[test_idx].FromVector(std::vector<double>(cam_cur.begin() + 15 * q, cam_cur.begin() + 15 * (q + 1)));
        success_ids.insert(static_cast<long>(test_idx));
        fprintf(stderr, "Running ptz-reloc success: %s\n", test_fnames[test_idx].c_str());
      }
      else fprintf(stderr, "Running ptz-reloc failed: %s\n", test_fnames[test_idx].c_str());
    }
  }

  const std::string cam_id = BaseName(parser.Get("test_images"));
  const std::string out_dir = parser.Get("output");
  MkdirIfNotExist(out_dir);
  std::vector<std::vector<Point2f>> pixels(test_fnames.size());
  std::vector<std::vector<Point3d>> pts3d(test_fnames.size());
  SaveRegisteredCam(test_cameras, success_ids, test_fnames, pixels, pts3d, out_dir + "/" + cam_id + ".json");
  return 0;
}
