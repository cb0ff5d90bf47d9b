// data_io.h -- on-disk formats of the reference (src/core/data_io.h / data_io.cc) without OpenCV / nlohmann:
//   * COLMAP text features  "N D\n x y scale ori d0 .. d(D-1)\n ..."                     (data_io.cc:24-52)
//   * pairs_matches.txt      blocks "nameA nameB\n i j\n ...\n<blank line>"                 (data_io.cc:64-110)
//   * camera / annotation JSON  {"cameras": {name: {name,pos,res,K,R,t,dist,distType,marker{pix,pos},version}}} (:112-295)
//   * image directory listing with sizes read from the file headers                       (data_io.cc:294-333)
//   * the N x N MatchesInfo table with one RANSAC homography per listed pair              (data_io.cc:336-400)
#pragma once

#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "types.h"

namespace ptzcalib {

void ReadColmapFeatures(const std::string& filepath, std::vector<KeyPoint>& kpts);
void ReadColmapMatches(const std::string& filepath, std::vector<std::vector<DMatch>>& pairs_matches,
                       std::vector<std::pair<std::string, std::string>>& img_pairs_name);
bool SaveToJson(const std::vector<Camera>& cameras, const std::vector<std::string>& names,
                const std::vector<std::vector<Point2f>>& pixels_gt, const std::vector<std::vector<Point3d>>& pts3d_gt,
                const std::string& filepath);
bool ReadFromJson(const std::string& filepath, std::vector<Camera>& cameras, std::vector<std::string>& names,
                  std::vector<std::vector<Point2f>>& pixels, std::vector<std::vector<Point3d>>& pts3d, std::vector<Size>& sizes);
bool ReadCamFromJson(const std::string& filepath, const std::vector<std::string>& names, std::vector<Camera>& cameras);
bool LoadImgsAndFeatures(const std::string& img_dir, const std::string& feature_dir, std::vector<std::string>& fnames,
                         std::vector<ImageFeatures>& features, std::vector<Size>& sizes);
bool LoadMatchesInfo(const std::string& matches_path, const std::vector<std::string>& fnames, const std::vector<ImageFeatures>& features,
                     std::vector<MatchesInfo>& matches_info);
bool LoadAnnotation(const std::string& annot_path, const std::vector<std::string>& fnames, std::vector<std::vector<Point2f>>& pixels,
                    std::vector<std::vector<Point3d>>& pts3d);
void SaveRegisteredCam(const std::vector<Camera>& cameras, const std::unordered_set<long>& reg_image_ids,
                       const std::vector<std::string>& fnames, const std::vector<std::vector<Point2f>>& pixels,
                       const std::vector<std::vector<Point3d>>& pts3d, const std::string& out_path);
long FindImgIndex(const std::vector<std::string>& fnames, const std::string& fname);

// path helpers with the semantics of the reference's utils/os_path.cc
std::string BaseName(const std::string& path);                                      // text after the last / or \ ("" for "dir/")
void SplitExt(const std::string& path, std::string* root, std::string* ext);         // ext keeps its dot and its case
bool MkdirIfNotExist(const std::string& dir);
std::vector<std::string> ListDir(const std::string& dir);                            // full paths, unsorted

}  // namespace ptzcalib
