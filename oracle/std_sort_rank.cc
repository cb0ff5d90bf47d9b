// std_sort_rank.cc -- oracle helper (test infrastructure).  The reference ranks images with
//   std::sort(indices.begin(), indices.end(), [&](int A, int B) { return confidences_rank[A] > confidences_rank[B]; });
// (src/core/ptz_incremental_optimizer.cc:192-194, 229-231, 281-283).  std::sort is not stable, and the scores are sums of
// quantised confidences, so equal scores are the norm: WHICH of two equal images comes first is decided by the standard
// library's sort.  The restatement therefore calls the same std::sort of the same toolchain on the same index array
// instead of imitating it in Python.
#include <algorithm>
#include <numeric>
#include <vector>

extern "C" int orc_rank_by_score(int n, const float* score, long* out)
{
  std::vector<long> indices(n);
  std::iota(indices.begin(), indices.end(), 0);
  std::sort(indices.begin(), indices.end(), [&](int A, int B) -> bool { return score[A] > score[B]; });
  int m = 0;
  for (long id : indices) {
    if (score[id] <= 0.0f) break;
    out[m++] = id;
  }
  return m;
}
