// Host build of the seed tree with scipy.spatial.cKDTree's layout (see ia3_kdtree.h for why the layout matters).
//
// Restates scipy/spatial/ckdtree/src/build.cxx for cKDTree(data) with its defaults — leafsize 16, compact_nodes=True
// (bounds recomputed from the points under every node), balanced_tree=True (split at the median) — as the reference
// constructs it (External/Fitting_v4.py:601).  Per node: bounds of its points; split axis = largest spread (first of
// equals); leaf if <= 16 points or zero spread; std::nth_element of the permutation range at its middle position by the
// coordinate on that axis (plain `<` on the coordinate); split = coordinate of the middle element; a Hoare pass that
// leaves [coordinate < split | coordinate >= split]; the sliding fix-ups when one side came out empty; children built
// less-side first, depth first (node numbering = creation order).  The permutation inside a leaf decides which of two
// equidistant seeds a query meets first, so the selection algorithm is part of the contract: SciPy's wheels and this
// library both get it from libstdc++ (introselect); tests/test_kdtree_cpu.py compares permutation and nodes with
// scipy's own (`tree.indices`, `tree.tree`) on thousands of points with many equal coordinates.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include "ia3_kdtree.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ia3k {

namespace {
struct Builder {
  const double* data;
  std::vector<int64_t> idx;
  std::vector<ia3::KdNode> nodes;
  int leafsize;
  int build(int64_t start_idx, int64_t end_idx, int parent, double* maxes, double* mins) {
    constexpr int m = 3;
    int64_t* indices = idx.data();
    nodes.push_back(ia3::KdNode());
    const int node_index = (int)nodes.size() - 1;
    {
      ia3::KdNode& n = nodes[node_index];
      n.split = 0.0; n.split_dim = -1; n.less = -1; n.greater = -1;
      n.start = (int)start_idx; n.end = (int)end_idx; n.parent = parent;
    }
    if (end_idx - start_idx <= leafsize) return node_index;
    {
      const double* p0 = data + indices[start_idx] * m;
      for (int i = 0; i < m; ++i) { maxes[i] = p0[i]; mins[i] = p0[i]; }
      for (int64_t j = start_idx + 1; j < end_idx; ++j) {
        const double* pp = data + indices[j] * m;
        for (int i = 0; i < m; ++i) {
          const double tmp = pp[i];
          maxes[i] = maxes[i] > tmp ? maxes[i] : tmp;
          mins[i] = mins[i] < tmp ? mins[i] : tmp;
        }
      }
    }
    int d = 0;
    double size = 0;
    for (int i = 0; i < m; ++i)
      if (maxes[i] - mins[i] > size) { d = i; size = maxes[i] - mins[i]; }
    if (maxes[d] == mins[d]) return node_index;   // all points identical: leaf
    double split;
    {
      const int64_t mid = (end_idx - start_idx) / 2;
      const double* dat = data;
      auto by_coordinate = [dat, d](int64_t a, int64_t b) { return dat[a * m + d] < dat[b * m + d]; };
      std::nth_element(indices + start_idx, indices + start_idx + mid, indices + end_idx, by_coordinate);
      split = data[indices[start_idx + mid] * m + d];
      // the median of a node whose smallest coordinate is that median would leave nothing below the split: cKDTree
      // moves the split to the next representable value, so that the points AT the minimum go to the less side
      if (split == mins[d]) split = std::nextafter(split, maxes[d]);
    }
    int64_t p = start_idx, q = end_idx - 1;
    while (p <= q) {
      if (data[indices[p] * m + d] < split) ++p;
      else if (data[indices[q] * m + d] >= split) --q;
      else { std::swap(indices[p], indices[q]); ++p; --q; }
    }
    if (p == start_idx) {          // no point below the split: slide it to the smallest coordinate
      int64_t j = start_idx;
      split = data[indices[j] * m + d];
      for (int64_t i = start_idx + 1; i < end_idx; ++i)
        if (data[indices[i] * m + d] < split) { j = i; split = data[indices[j] * m + d]; }
      std::swap(indices[start_idx], indices[j]);
      p = start_idx + 1;
    } else if (p == end_idx) {     // no point at or above it: slide to the largest
      int64_t j = end_idx - 1;
      split = data[indices[j] * m + d];
      for (int64_t i = start_idx; i < end_idx - 1; ++i)
        if (data[indices[i] * m + d] > split) { j = i; split = data[indices[j] * m + d]; }
      std::swap(indices[end_idx - 1], indices[j]);
      p = end_idx - 1;
    }
    const int less = build(start_idx, p, node_index, maxes, mins);
    const int greater = build(p, end_idx, node_index, maxes, mins);
    ia3::KdNode& n = nodes[node_index];
    n.less = less; n.greater = greater; n.split_dim = d; n.split = split;
    return node_index;
  }
};
}  // namespace

// points: n x 3 float64.  nodes / indices as ia3::KdTree wants them; mins / maxes: bounds of all points (cKDTree.mins/maxes)
void kd_build(const double* points, int n, std::vector<ia3::KdNode>& nodes, std::vector<int>& indices, double* mins,
              double* maxes) {
  Builder b;
  b.data = points; b.leafsize = 16;
  b.idx.resize((size_t)n);
  for (int i = 0; i < n; ++i) b.idx[(size_t)i] = i;
  for (int k = 0; k < 3; ++k) { mins[k] = n ? points[k] : 0.0; maxes[k] = mins[k]; }
  for (int j = 1; j < n; ++j)
    for (int k = 0; k < 3; ++k) {
      const double v = points[3 * (size_t)j + k];
      if (v > maxes[k]) maxes[k] = v;
      if (v < mins[k]) mins[k] = v;
    }
  double wmax[3] = {maxes[0], maxes[1], maxes[2]}, wmin[3] = {mins[0], mins[1], mins[2]};
  b.nodes.reserve((size_t)(n / 4 + 4));
  if (n > 0) b.build(0, n, -1, wmax, wmin);
  nodes.swap(b.nodes);
  indices.resize((size_t)n);
  for (int i = 0; i < n; ++i) indices[(size_t)i] = (int)b.idx[(size_t)i];
}

}  // namespace ia3k
