// tracks.h -- feature-track builder with the reference's interface (src/core/tracks.h:37-65) and the
// reference's track ids: a track id is the union-find root index of its nodes after union by rank with path
// compression in file order of the pairs and matches (tracks.cc:47-60, union_find.h:71-95).  The ids and the
// (track asc, image asc) iteration order are part of the bit-exact indexing contract of the solver.
#pragma once

#include <map>
#include <set>
#include <utility>
#include <cstdint>
#include <vector>

#include "types.h"

namespace ptzcalib {

using IndexedFeaturePair = std::pair<int, int>;  // {ImageId, FeatureId}
using Track = std::map<int, int>;                 // {ImageId -> FeatureId}
using Tracks = std::map<int, Track>;              // {TrackId -> Track}

class TracksBuilder {
 public:
  void Build(const std::vector<MatchesInfo>& matches_info);
  void Filter(int min_track_length = 2);  // drop tracks with < N images or with a repeated image
  void ExportToSTL(Tracks& tracks);
  // flat form of the same export (ascending track id, ascending image id inside), without building the maps
  void ExportFlat(std::vector<int>& id, std::vector<int64_t>& ptr, std::vector<int>& img, std::vector<int>& feat) const;
  size_t NbTracks() const;

 private:
  int FindRoot(int i);
  int NodeIndex(const IndexedFeaturePair& node) const;
  std::vector<IndexedFeaturePair> nodes_;  // sorted unique (image, feature); position = node index
  std::vector<int> parent_, rank_, size_;
  std::vector<size_t> dense_offset_;  // per image: first slot of its feature range in dense_rank_ (empty = sorted path)
  std::vector<int> dense_rank_;       // node index of (image, feature), -1 = not a node
};

void Length(const Tracks& tracks, int& total_length, int& max_length, int& min_length);
void FindMaxCoVisible(const Tracks& tracks, int num_images, std::set<int>& max_connect_imgs);

}  // namespace ptzcalib
