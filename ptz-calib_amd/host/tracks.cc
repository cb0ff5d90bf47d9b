#include "tracks.h"

#include <algorithm>
#include <climits>
#include <limits>

namespace ptzcalib {

int TracksBuilder::NodeIndex(const IndexedFeaturePair& node) const
{
  if (!dense_rank_.empty()) return dense_rank_[dense_offset_[node.first] + node.second];
  return static_cast<int>(std::lower_bound(nodes_.begin(), nodes_.end(), node) - nodes_.begin());
}

int TracksBuilder::FindRoot(int i)
{
  int root = i;
  while (parent_[root] != root) root = parent_[root];
  while (parent_[i] != root) {  // full path compression, as the recursive form of union_find.h:58-68
    const int next = parent_[i];
    parent_[i] = root;
    i = next;
  }
  return root;
}

void TracksBuilder::Build(const std::vector<MatchesInfo>& matches_info)
{
  // Node numbering = rank of (image, feature) in the sorted set of matched features (tracks.cc:24-41); the union-find
  // roots, hence the track ids, depend on it.  Image and feature ids are small non-negative integers in practice: a dense
  // presence table gives the ranks in one pass; anything else goes through sort + unique + binary search.
  nodes_.clear();
  dense_offset_.clear();
  dense_rank_.clear();
  long max_img = -1;
  bool dense = true;
  for (const auto& mi : matches_info) {
    if (mi.matches.empty()) continue;
    if (mi.src_img_idx < 0 || mi.dst_img_idx < 0 || mi.src_img_idx > (1 << 24) || mi.dst_img_idx > (1 << 24)) { dense = false; break; }
    max_img = std::max(max_img, std::max(mi.src_img_idx, mi.dst_img_idx));
  }
  if (dense && max_img >= 0) {
    std::vector<int> max_feat(static_cast<size_t>(max_img) + 1, -1);
    for (const auto& mi : matches_info)
      for (const auto& m : mi.matches) {
        if (m.queryIdx < 0 || m.trainIdx < 0) { dense = false; break; }
        max_feat[mi.src_img_idx] = std::max(max_feat[mi.src_img_idx], m.queryIdx);
        max_feat[mi.dst_img_idx] = std::max(max_feat[mi.dst_img_idx], m.trainIdx);
      }
    size_t total = 0;
    if (dense) {
      dense_offset_.assign(static_cast<size_t>(max_img) + 2, 0);
      for (long i = 0; i <= max_img; ++i) { dense_offset_[i] = total; total += static_cast<size_t>(max_feat[i] + 1); }
      dense_offset_[max_img + 1] = total;
      if (total > (size_t(1) << 27)) dense = false;  // sparse, huge feature ids: not worth a table
    }
    if (dense) {
      dense_rank_.assign(total, -1);
      for (const auto& mi : matches_info)
        for (const auto& m : mi.matches) {
          dense_rank_[dense_offset_[mi.src_img_idx] + m.queryIdx] = 0;
          dense_rank_[dense_offset_[mi.dst_img_idx] + m.trainIdx] = 0;
        }
      int count = 0;
      for (long i = 0; i <= max_img; ++i)
        for (size_t s = dense_offset_[i]; s < dense_offset_[i + 1]; ++s)
          if (dense_rank_[s] == 0) {
            dense_rank_[s] = count++;
            nodes_.emplace_back(static_cast<int>(i), static_cast<int>(s - dense_offset_[i]));
          }
    }
  }
  else dense = false;
  if (!dense) {
    dense_offset_.clear();
    dense_rank_.clear();
    nodes_.clear();
    for (const auto& mi : matches_info)
      for (const auto& m : mi.matches) {
        nodes_.emplace_back(static_cast<int>(mi.src_img_idx), m.queryIdx);
        nodes_.emplace_back(static_cast<int>(mi.dst_img_idx), m.trainIdx);
      }
    std::sort(nodes_.begin(), nodes_.end());
    nodes_.erase(std::unique(nodes_.begin(), nodes_.end()), nodes_.end());
  }
  const int n = static_cast<int>(nodes_.size());
  parent_.resize(n);
  for (int i = 0; i < n; ++i) parent_[i] = i;
  rank_.assign(n, 0);
  size_.assign(n, 1);
  for (const auto& mi : matches_info)
    for (const auto& m : mi.matches) {
      const int ri = FindRoot(NodeIndex({static_cast<int>(mi.src_img_idx), m.queryIdx}));
      const int rj = FindRoot(NodeIndex({static_cast<int>(mi.dst_img_idx), m.trainIdx}));
      if (ri == rj) continue;
      if (rank_[ri] < rank_[rj]) {  // union by rank: the lower-rank root goes under the higher one
        parent_[ri] = rj;
        size_[rj] += size_[ri];
      }
      else {
        parent_[rj] = ri;
        size_[ri] += size_[rj];
        if (rank_[ri] == rank_[rj]) ++rank_[ri];
      }
    }
}

void TracksBuilder::Filter(int min_track_length)
{
  const int n = static_cast<int>(nodes_.size());
  std::vector<int> n_img(n, 0), last_img(n, -1);
  std::vector<char> bad(n, 0), seen(n, 0);
  for (int k = 0; k < n; ++k) {
    const int r = FindRoot(k);  // leaves parent_[k] == root for every node
    seen[r] = 1;
    if (last_img[r] == nodes_[k].first) bad[r] = 1;  // nodes are image-sorted: a repeat is adjacent per root
    else { last_img[r] = nodes_[k].first; ++n_img[r]; }
  }
  for (int r = 0; r < n; ++r)
    if (seen[r] && n_img[r] < min_track_length) bad[r] = 1;
  for (int k = 0; k < n; ++k) {
    const int r = parent_[k];
    if (r != std::numeric_limits<int>::max() && bad[r]) {
      size_[r] = 1;
      parent_[k] = std::numeric_limits<int>::max();
    }
  }
}

void TracksBuilder::ExportToSTL(Tracks& tracks)
{
  tracks.clear();
  for (size_t k = 0; k < nodes_.size(); ++k) {
    const int id = parent_[k];
    if (id != std::numeric_limits<int>::max() && size_[id] > 1) tracks[id].insert(nodes_[k]);
  }
}

// The same content as ExportToSTL without the node-per-entry maps: tracks in ascending id order, the views of a track in
// ascending image order (the nodes are image-sorted, and a filtered track holds an image once) -- the iteration order of the map.
void TracksBuilder::ExportFlat(std::vector<int>& id, std::vector<int64_t>& ptr, std::vector<int>& img, std::vector<int>& feat) const
{
  const int n = static_cast<int>(nodes_.size());
  const int none = std::numeric_limits<int>::max();
  std::vector<int> count(static_cast<size_t>(n) + 1, 0);
  for (int k = 0; k < n; ++k) {
    const int r = parent_[k];
    if (r != none && size_[r] > 1) ++count[r];
  }
  id.clear(); ptr.clear();
  std::vector<int64_t> start(static_cast<size_t>(n), -1);
  int64_t run = 0;
  ptr.push_back(0);
  for (int r = 0; r < n; ++r)
    if (count[r] > 0) {
      id.push_back(r);
      start[r] = run;
      run += count[r];
      ptr.push_back(run);
    }
  img.assign(static_cast<size_t>(run), 0);
  feat.assign(static_cast<size_t>(run), 0);
  for (int k = 0; k < n; ++k) {
    const int r = parent_[k];
    if (r == none || size_[r] <= 1) continue;
    const int64_t slot = start[r]++;
    img[slot] = nodes_[k].first;
    feat[slot] = nodes_[k].second;
  }
}

size_t TracksBuilder::NbTracks() const
{
  std::set<int> ids(parent_.begin(), parent_.end());
  ids.erase(std::numeric_limits<int>::max());
  return ids.size();
}

void Length(const Tracks& tracks, int& total_length, int& max_length, int& min_length)
{
  total_length = 0; max_length = 0; min_length = INT_MAX;
  for (const auto& t : tracks) {
    const int l = static_cast<int>(t.second.size());
    total_length += l;
    max_length = std::max(max_length, l);
    min_length = std::min(min_length, l);
  }
}

void FindMaxCoVisible(const Tracks& tracks, int num_images, std::set<int>& max_connect_imgs)
{
  // connected components of the image co-visibility graph (largest one), tracks.cc:147-205
  std::vector<int> comp(num_images);
  for (int i = 0; i < num_images; ++i) comp[i] = i;
  auto find = [&](int i) { while (comp[i] != i) i = comp[i] = comp[comp[i]]; return i; };
  std::vector<char> used(num_images, 0);
  for (const auto& t : tracks) {
    int first = -1;
    for (const auto& kv : t.second) {
      if (kv.first < 0 || kv.first >= num_images) continue;
      used[kv.first] = 1;
      if (first < 0) first = find(kv.first);
      else comp[find(kv.first)] = first;
    }
  }
  std::map<int, std::set<int>> groups;
  for (int i = 0; i < num_images; ++i)
    if (used[i]) groups[find(i)].insert(i);
  max_connect_imgs.clear();
  for (const auto& g : groups)
    if (g.second.size() > max_connect_imgs.size()) max_connect_imgs = g.second;
}

}  // namespace ptzcalib
