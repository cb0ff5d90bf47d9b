// ref_tracks_driver.cc -- TEST INFRASTRUCTURE.  Drives the reference's own UnionFind
// (union_find.h:33-106) and flat_pair_map (flat_pair_map.h:22-52), included from
// /root/reference/src/core where they lie (-I on the command line; never copied), through the
// Build / Filter / ExportToSTL sequence of tracks.cc:19-113.  tracks.cc itself cannot be compiled
// here (tracks.h -> types.h -> <opencv2/calib3d.hpp>, absent), so the three member functions are
// re-expressed below over plain integer match lists; the data structures that decide the track ids
// (union by rank + path compression, sorted flat map lookups) are the reference's.
//
// stdin:  n_pairs, then per pair: src dst n_matches, then n_matches lines "queryIdx trainIdx";
//         finally min_track_length.
// stdout: n_tracks, then per track: track_id n_entries, then "image feature" pairs.
#include <climits>
#include <cstdio>
#include <limits>
#include <map>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

#include "flat_pair_map.h"
#include "union_find.h"

using namespace ptzcalib;
typedef std::pair<int, int> IndexedFeaturePair;

struct PairMatches {
  long src, dst;
  std::vector<std::pair<int, int> > m;
};

int main()
{
  int n_pairs;
  if (scanf("%d", &n_pairs) != 1) return 1;
  std::vector<PairMatches> pairs(n_pairs);
  for (int p = 0; p < n_pairs; ++p) {
    int nm;
    if (scanf("%ld %ld %d", &pairs[p].src, &pairs[p].dst, &nm) != 3) return 1;
    pairs[p].m.resize(nm);
    for (int k = 0; k < nm; ++k)
      if (scanf("%d %d", &pairs[p].m[k].first, &pairs[p].m[k].second) != 2) return 1;
  }
  int min_track_length;
  if (scanf("%d", &min_track_length) != 1) return 1;

  flat_pair_map<IndexedFeaturePair, int> map_node_to_index;
  UnionFind uf_tree;

  // Build (tracks.cc:19-61)
  std::set<IndexedFeaturePair> all_features;
  for (size_t p = 0; p < pairs.size(); ++p)
    for (size_t k = 0; k < pairs[p].m.size(); ++k) {
      all_features.insert(IndexedFeaturePair((int)pairs[p].src, pairs[p].m[k].first));
      all_features.insert(IndexedFeaturePair((int)pairs[p].dst, pairs[p].m[k].second));
    }
  map_node_to_index.reserve(all_features.size());
  int count = 0;
  for (std::set<IndexedFeaturePair>::const_iterator it = all_features.begin(); it != all_features.end(); ++it) {
    map_node_to_index.emplace_back(*it, count);
    ++count;
  }
  map_node_to_index.sort();
  all_features.clear();
  uf_tree.InitSets(static_cast<int>(map_node_to_index.size()));
  for (size_t p = 0; p < pairs.size(); ++p)
    for (size_t k = 0; k < pairs[p].m.size(); ++k) {
      IndexedFeaturePair pair_i((int)pairs[p].src, pairs[p].m[k].first);
      IndexedFeaturePair pair_j((int)pairs[p].dst, pairs[p].m[k].second);
      uf_tree.Union(map_node_to_index[pair_i], map_node_to_index[pair_j]);
    }

  // Filter (tracks.cc:63-97)
  const flat_pair_map<IndexedFeaturePair, int>& cmap = map_node_to_index;
  std::map<int, std::set<int> > tracks;
  std::set<int> problematic;
  for (size_t k = 0; k < cmap.size(); ++k) {
    const int track_id = uf_tree.Find((int)k);
    if (tracks[track_id].insert(cmap[k].first.first).second == false) problematic.insert(track_id);
  }
  for (std::map<int, std::set<int> >::const_iterator it = tracks.begin(); it != tracks.end(); ++it)
    if ((int)it->second.size() < min_track_length) problematic.insert(it->first);
  for (size_t i = 0; i < uf_tree.m_cc_parent.size(); ++i) {
    int& root_index = uf_tree.m_cc_parent[i];
    if (problematic.count(root_index) > 0) {
      uf_tree.m_cc_size[root_index] = 1;
      root_index = std::numeric_limits<int>::max();
    }
  }

  // ExportToSTL (tracks.cc:99-113)
  std::map<int, std::map<int, int> > out;
  for (size_t k = 0; k < cmap.size(); ++k) {
    const int track_id = uf_tree.m_cc_parent[k];
    if (track_id != std::numeric_limits<int>::max() && uf_tree.m_cc_size[track_id] > 1) out[track_id].insert(cmap[k].first);
  }
  printf("%zu\n", out.size());
  for (std::map<int, std::map<int, int> >::const_iterator t = out.begin(); t != out.end(); ++t) {
    printf("%d %zu\n", t->first, t->second.size());
    for (std::map<int, int>::const_iterator e = t->second.begin(); e != t->second.end(); ++e) printf("%d %d\n", e->first, e->second);
  }
  return 0;
}
