#include "data_io.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>

#include "homography.h"
#include "image_size.h"
#include "json_mini.h"

namespace ptzcalib {

// ---- paths ------------------------------------------------------------------------------------------------------
std::string BaseName(const std::string& path)
{
  const size_t found = path.find_last_of("/\\");
  return found == std::string::npos ? path : path.substr(found + 1);
}

void SplitExt(const std::string& path, std::string* root, std::string* ext)
{
  // the dot must belong to the last path component and must not be its first character(s): ".bashrc" has no extension
  const size_t slash = path.find_last_of("/\\");
  const size_t first = slash == std::string::npos ? 0 : slash + 1;
  const size_t dot = path.find_last_of('.');
  bool has = dot != std::string::npos && dot >= first;
  if (has) {
    size_t k = first;
    while (k < dot && path[k] == '.') ++k;
    if (k == dot) has = false;  // only dots in front of it
  }
  if (!has) { *root = path; *ext = ""; return; }
  *root = path.substr(0, dot);
  *ext = path.substr(dot);
}

bool MkdirIfNotExist(const std::string& dir)
{
  struct stat st;
  if (stat(dir.c_str(), &st) == 0) return true;
  return mkdir(dir.c_str(), 0755) == 0;
}

std::vector<std::string> ListDir(const std::string& dir)
{
  std::vector<std::string> paths;
  DIR* d = opendir(dir.c_str());
  if (!d) return paths;
  while (struct dirent* e = readdir(d)) {
    if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
    std::string p = dir;
    if (!p.empty() && p.back() != '/') p.push_back('/');
    paths.push_back(p + e->d_name);
  }
  closedir(d);
  return paths;
}

long FindImgIndex(const std::vector<std::string>& fnames, const std::string& fname)
{  // names are compared without their extensions (data_io.cc:456-472)
  std::string want, e0;
  SplitExt(fname, &want, &e0);
  for (size_t i = 0; i < fnames.size(); ++i) {
    std::string root, ext;
    SplitExt(fnames[i], &root, &ext);
    if (root == want) return static_cast<long>(i);
  }
  return -1;
}

// ---- COLMAP text files ----------------------------------------------------------------------------------------------
void ReadColmapFeatures(const std::string& filepath, std::vector<KeyPoint>& kpts)
{
  kpts.clear();
  std::ifstream fin(filepath);
  if (!fin.good()) return;
  int num_kpts = 0, desc_dim = 0;
  fin >> num_kpts >> desc_dim;
  if (!fin || num_kpts < 0 || desc_dim < 0) return;
  kpts.resize(static_cast<size_t>(num_kpts));
  float scale, orientation, d;
  for (int i = 0; i < num_kpts; ++i) {
    fin >> kpts[i].pt.x >> kpts[i].pt.y >> scale >> orientation;
    for (int j = 0; j < desc_dim; ++j) fin >> d;  // descriptors are not used by the optimizers
  }
  // a short file leaves the remaining key points at (0, 0), as the reference's unchecked stream reads do
}

static bool HasEnding(const std::string& s, const std::string& ending)
{
  return s.length() >= ending.length() && 0 == s.compare(s.length() - ending.length(), ending.length(), ending);
}

void ReadColmapMatches(const std::string& filepath, std::vector<std::vector<DMatch>>& pairs_matches,
                       std::vector<std::pair<std::string, std::string>>& img_pairs_name)
{
  pairs_matches.clear();
  img_pairs_name.clear();
  std::vector<DMatch> matches;
  std::pair<std::string, std::string> img_pair;
  std::ifstream fin(filepath);
  std::string line;
  // A block is committed only by the blank line that follows it: the last block of a file that does not end with a blank
  // line is dropped, and so is a block without matches (data_io.cc:75-86).
  while (std::getline(fin, line)) {
    if (line.empty()) {
      if (!matches.empty()) {
        pairs_matches.push_back(matches);
        img_pairs_name.push_back(img_pair);
        matches.clear();
        img_pair = {};
      }
      continue;
    }
    std::string str1, str2;
    std::istringstream iss(line);
    iss >> str1 >> str2;
    if (HasEnding(str1, ".png") || HasEnding(str1, ".jpg") || HasEnding(str1, ".jpeg")) img_pair = {str1, str2};
    else {
      try {
        DMatch m;
        m.queryIdx = std::stoi(str1);
        m.trainIdx = std::stoi(str2);
        matches.push_back(m);
      }
      catch (const std::exception&) {
        return;  // std::stoi throws in the reference too: reading stops, what was committed so far is kept (:103-105)
      }
    }
  }
}

// ---- camera JSON ------------------------------------------------------------------------------------------------------
bool SaveToJson(const std::vector<Camera>& cameras, const std::vector<std::string>& names,
                const std::vector<std::vector<Point2f>>& pixels_gt, const std::vector<std::vector<Point3d>>& pts3d_gt,
                const std::string& filepath)
{
  Json all = Json::Object();
  Json& cams = all["cameras"];
  cams = Json::Object();
  for (size_t i = 0; i < cameras.size(); ++i) {
    Json j = Json::Object();
    std::string rootname, ext;
    SplitExt(names[i], &rootname, &ext);
    j["name"] = Json::String(rootname);
    const Vec3 twc = cameras[i].t_wc();
    j["pos"] = Json::FloatArray({twc[0], twc[1], twc[2]});
    const int width = static_cast<int>(2 * cameras[i].K()[2]);
    const int height = static_cast<int>(2 * cameras[i].K()[5]);
    Json res = Json::Array();
    res.push_back(Json::Int(width));
    res.push_back(Json::Int(height));
    j["res"] = res;
    j["K"] = Json::FloatArray(std::vector<double>(cameras[i].K().begin(), cameras[i].K().end()));
    j["R"] = Json::FloatArray(std::vector<double>(cameras[i].R().begin(), cameras[i].R().end()));
    j["t"] = Json::FloatArray(std::vector<double>(cameras[i].t().begin(), cameras[i].t().end()));
    j["dist"] = Json::FloatArray(std::vector<double>(cameras[i].dist().begin(), cameras[i].dist().end()));
    j["distType"] = Json::String(cameras[i].dist()[0] < 1e-5 ? "" : "k1");  // the sign-sensitive test of the reference (:140)
    Json pix = Json::Array(), pos = Json::Array();
    for (size_t k = 0; k < pixels_gt[i].size(); ++k) {
      // float division, stored as float, printed as the double it converts to
      const float px = pixels_gt[i][k].x / width, py = pixels_gt[i][k].y / height;
      pix.push_back(Json::FloatArray({static_cast<double>(px), static_cast<double>(py)}));
      pos.push_back(Json::FloatArray({pts3d_gt[i][k].x, pts3d_gt[i][k].y, pts3d_gt[i][k].z}));
    }
    Json marker = Json::Object();
    marker["pix"] = pix;
    marker["pos"] = pos;
    j["marker"] = marker;
    j["version"] = Json::String("2.0");
    cams[rootname] = j;
  }
  std::ofstream fout(filepath, std::ios_base::out);
  if (!fout.good()) return false;
  fout << all.dump(4) << std::endl;
  return true;
}

static Json ReadJsonFile(const std::string& filepath)
{
  if (filepath.empty()) return Json::Object();
  std::ifstream in(filepath, std::ios::in);
  if (!in.is_open() || in.fail()) return Json::Object();
  std::stringstream ss;
  ss << in.rdbuf();
  Json j;
  if (!Json::Parse(ss.str(), j)) return Json::Object();
  return j;
}

static void FillCamera(const Json& value, Camera& cam)
{
  const std::vector<double> K = value.at("K").number_array(), R = value.at("R").number_array(), t = value.at("t").number_array(),
                            d = value.at("dist").number_array();
  // the reference memcpy's vec.size() doubles into fixed-size matrices; sizes other than 9/9/3/5 would overrun there
  if (K.size() != 9 || R.size() != 9 || t.size() != 3 || d.size() != 5) throw std::runtime_error("camera entry with wrong sizes");
  for (int k = 0; k < 9; ++k) { cam.K()[k] = K[k]; cam.R()[k] = R[k]; }
  for (int k = 0; k < 3; ++k) cam.t()[k] = t[k];
  for (int k = 0; k < 5; ++k) cam.dist()[k] = d[k];
}

bool ReadFromJson(const std::string& filepath, std::vector<Camera>& cameras, std::vector<std::string>& names,
                  std::vector<std::vector<Point2f>>& pixels, std::vector<std::vector<Point3d>>& pts3d, std::vector<Size>& sizes)
{
  cameras.clear(); names.clear(); pixels.clear(); pts3d.clear(); sizes.clear();
  const Json j = ReadJsonFile(filepath);
  if (j.empty()) return false;
  try {
    const Json& jc = j.at("cameras");
    for (const std::string& name : jc.sorted_keys()) {  // nlohmann::json iterates objects in key order
      const Json& value = jc.at(name);
      Camera cam;
      FillCamera(value, cam);
      Size size;
      size.width = static_cast<int>(value.at("res").at(0).integer());
      size.height = static_cast<int>(value.at("res").at(1).integer());
      std::vector<Point2f> pixs;
      std::vector<Point3d> pts;
      for (const Json& p : value.at("marker").at("pix").items())  // stored normalised by the image size (:252-257)
        pixs.emplace_back(static_cast<float>(size.width * p.at(0).number()), static_cast<float>(size.height * p.at(1).number()));
      for (const Json& p : value.at("marker").at("pos").items()) pts.emplace_back(p.at(0).number(), p.at(1).number(), p.at(2).number());
      names.push_back(name);
      pixels.push_back(pixs);
      pts3d.push_back(pts);
      cameras.push_back(cam);
      sizes.push_back(size);
    }
    return true;
  }
  catch (const std::exception&) {
    return false;
  }
}

bool ReadCamFromJson(const std::string& filepath, const std::vector<std::string>& names, std::vector<Camera>& cameras)
{
  cameras.clear();
  cameras.resize(names.size());
  const Json j = ReadJsonFile(filepath);
  if (j.empty()) return false;
  try {
    const Json& jc = j.at("cameras");
    for (size_t i = 0; i < names.size(); ++i) {
      std::string rootname, ext;
      SplitExt(names[i], &rootname, &ext);
      if (!jc.contains(rootname)) return false;
      FillCamera(jc.at(rootname), cameras[i]);
    }
    return true;
  }
  catch (const std::exception&) {
    return false;
  }
}

// ---- images, features, matches ----------------------------------------------------------------------------------------
bool LoadImgsAndFeatures(const std::string& img_dir, const std::string& feature_dir, std::vector<std::string>& fnames,
                         std::vector<ImageFeatures>& features, std::vector<Size>& sizes)
{
  std::vector<std::string> fpaths = ListDir(img_dir);
  std::sort(fpaths.begin(), fpaths.end());
  fnames.clear(); features.clear(); sizes.clear();
  static const char* kValidExts[] = {".png", ".jpg", ".jpeg", ".bmp", ".tiff"};
  for (const std::string& fpath : fpaths) {
    const std::string fname = BaseName(fpath);
    std::string rootname, ext;
    SplitExt(fname, &rootname, &ext);
    bool valid = false;
    for (const char* e : kValidExts) valid |= (ext == e);
    if (!valid || fname == "mask.png") continue;
    Size size;
    if (!ReadImageSize(fpath, size)) continue;  // cv::imread(...).empty()
    ImageFeatures feature;
    feature.img_size = size;
    ReadColmapFeatures(feature_dir + "/" + fname + ".txt", feature.keypoints);
    fnames.push_back(fname);
    features.push_back(feature);
    sizes.push_back(size);
  }
  return fnames.size() >= 2;
}

bool LoadMatchesInfo(const std::string& matches_path, const std::vector<std::string>& fnames, const std::vector<ImageFeatures>& features,
                     std::vector<MatchesInfo>& matches_info)
{
  std::vector<std::vector<DMatch>> pairs_matches;
  std::vector<std::pair<std::string, std::string>> img_pairs_name;
  ReadColmapMatches(matches_path, pairs_matches, img_pairs_name);
  matches_info.clear();
  const size_t num_images = fnames.size();
  matches_info.resize(num_images * num_images);  // value-initialised cells: (0, 0), no matches, empty H, confidence 0
  for (size_t p = 0; p < pairs_matches.size(); ++p) {
    const long index_i = FindImgIndex(fnames, img_pairs_name[p].first);
    const long index_j = FindImgIndex(fnames, img_pairs_name[p].second);
    if (index_i < 0 || index_j < 0) continue;  // the reference indexes with -1 here (undefined behaviour); skip the pair
    const std::vector<DMatch>& ms = pairs_matches[p];
    std::vector<Point2f> a, b;
    bool in_range = true;
    for (const DMatch& m : ms) {
      if (m.queryIdx < 0 || m.trainIdx < 0 || static_cast<size_t>(m.queryIdx) >= features[index_i].keypoints.size() ||
          static_cast<size_t>(m.trainIdx) >= features[index_j].keypoints.size()) { in_range = false; break; }
      a.push_back(features[index_i].keypoints[m.queryIdx].pt);
      b.push_back(features[index_j].keypoints[m.trainIdx].pt);
    }
    if (!in_range) continue;
    MatchesInfo mi;
    mi.matches = ms;
    static const double kRansacThresh = 4.0;
    mi.H_empty = !FindHomographyRansac(a, b, kRansacThresh, mi.H);  // H_j_i: pixel of image i -> pixel of image j
    mi.inliers_mask.assign(ms.size(), 1);
    mi.num_inliers = static_cast<int>(ms.size());
    static const int kMaxNumMatches = 100;  // CalMatchingScore (:358-366): float ratio, stored in a double
    mi.confidence = static_cast<int>(ms.size()) >= kMaxNumMatches ? 1.0f : static_cast<float>(ms.size()) / static_cast<float>(kMaxNumMatches);
    mi.src_img_idx = index_i;
    mi.dst_img_idx = index_j;
    matches_info[static_cast<size_t>(index_i) * num_images + static_cast<size_t>(index_j)] = mi;
  }
  return true;
}

bool LoadAnnotation(const std::string& annot_path, const std::vector<std::string>& fnames, std::vector<std::vector<Point2f>>& pixels,
                    std::vector<std::vector<Point3d>>& pts3d)
{
  std::vector<std::string> gt_names;
  std::vector<std::vector<Point2f>> gt_pixels;
  std::vector<std::vector<Point3d>> gt_pts3d;
  std::vector<Camera> gt_cameras;
  std::vector<Size> gt_sizes;
  pixels.clear();
  pts3d.clear();
  if (!ReadFromJson(annot_path, gt_cameras, gt_names, gt_pixels, gt_pts3d, gt_sizes)) return false;
  pixels.resize(fnames.size());
  pts3d.resize(fnames.size());
  for (size_t i = 0; i < gt_cameras.size(); ++i) {
    const long idx = FindImgIndex(fnames, gt_names[i]);
    if (idx == -1) continue;
    pixels[idx] = gt_pixels[i];
    pts3d[idx] = gt_pts3d[i];
  }
  return true;
}

void SaveRegisteredCam(const std::vector<Camera>& cameras, const std::unordered_set<long>& reg_image_ids,
                       const std::vector<std::string>& fnames, const std::vector<std::vector<Point2f>>& pixels,
                       const std::vector<std::vector<Point3d>>& pts3d, const std::string& out_path)
{
  std::vector<Camera> cams;
  std::vector<std::string> names;
  std::vector<std::vector<Point2f>> pix;
  std::vector<std::vector<Point3d>> pts;
  for (size_t i = 0; i < cameras.size(); ++i) {
    if (reg_image_ids.find(static_cast<long>(i)) == reg_image_ids.end()) continue;
    cams.push_back(cameras[i]);
    names.push_back(fnames[i]);
    pix.push_back(pixels[i]);
    pts.push_back(pts3d[i]);
  }
  SaveToJson(cams, names, pix, pts, out_path);
}

}  // namespace ptzcalib
