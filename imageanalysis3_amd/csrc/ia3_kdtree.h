// Nearest-seed query with scipy.spatial.cKDTree's traversal order.
//
// Reference: External/Fitting_v4.py:601 builds `KDTree(self.centers)` (scipy.spatial.cKDTree, defaults: leafsize 16,
// compact_nodes, balanced_tree) and :612 / :422-424 keeps a ball voxel for seed ic when
// `tree.query(xyz, distance_upper_bound=2*radius_fit)` returns ic.  A voxel equidistant from two seeds goes to the one
// the query meets FIRST (a candidate replaces the current best only if it is strictly nearer), so the tie winner is a
// property of the tree layout and of the traversal: best-first over the nodes by their lower bound, the near child
// followed directly, the far child queued in cKDTree's own binary heap.  SciPy is a dependency of the reference, not
// part of it; this restates the published algorithm (scipy/spatial/ckdtree/src/build.cxx, query.cxx; pinned against
// scipy 1.15.3 by tests/test_kdtree_cpu.py: permutation, nodes and query results identical on tie-rich integer fields).
//
// The build (kdtree.cpp, host: it is std::nth_element + a Hoare partition per node, sequential by nature) produces the
// node array and the point permutation; kd_nearest below is host/device agnostic.  Differences in FORM to query.cxx, none
// in result: a queued node carries its lower bound and its index only — the per-axis side distances cKDTree copies into
// every queue entry are a function of the path from the root (the deepest split on that axis where the path took the
// far side, else the root's distance), so they are recomputed by walking up when an entry is taken off the queue; this
// makes an entry 16 bytes, which lets every lane of a wavefront keep its own queue in LDS.
#pragma once
#include "ia3_model.h"   // IA3_HD

namespace ia3 {

struct KdNode {
  double split;
  int split_dim;     // -1: leaf
  int less, greater; // children (inner nodes)
  int start, end;    // range of the point permutation under this node
  int parent;        // -1: root
};

struct KdTree {
  const KdNode* nodes;
  const int* indices;    // permutation of the points, leaf ranges contiguous
  const double* data;    // n x 3 points
  double mins[3], maxes[3];
  int n;
};

// cKDTree's heap (ckdtree_decl.h `struct heap`): push sifts up while strictly smaller than the parent, remove moves the
// last entry to the root and sifts down (towards the smaller child, the right one only if strictly smaller than the left).
// Store: entry k of this queue at p[k * stride] (stride 1 on the host, 64 for a per-lane queue laid out [entry][lane]).
struct KdQEntry { double priority; int node; int pad; };

template <class EntryPtr>
struct KdQueue {
  EntryPtr h;
  int stride, cap, n;
  bool overflow;
  IA3_HD KdQueue(EntryPtr p, int stride_, int cap_) : h(p), stride(stride_), cap(cap_), n(0), overflow(false) {}
  IA3_HD void push(double priority, int node) {
    if (n >= cap) { overflow = true; return; }
    int i = n++;
    h[i * stride].priority = priority; h[i * stride].node = node;
    while (i > 0) {
      const int up = (i - 1) / 2;
      if (!(h[i * stride].priority < h[up * stride].priority)) break;
      const double tp = h[up * stride].priority; const int tn = h[up * stride].node;
      h[up * stride].priority = h[i * stride].priority; h[up * stride].node = h[i * stride].node;
      h[i * stride].priority = tp; h[i * stride].node = tn;
      i = up;
    }
  }
  IA3_HD void pop(double& priority, int& node) {
    priority = h[0].priority; node = h[0].node;
    h[0].priority = h[(n - 1) * stride].priority; h[0].node = h[(n - 1) * stride].node;
    --n;
    int i = 0, j = 1, k = 2;
    while ((j < n && h[i * stride].priority > h[j * stride].priority) ||
           (k < n && h[i * stride].priority > h[k * stride].priority)) {
      const int l = (k < n && h[j * stride].priority > h[k * stride].priority) ? k : j;
      const double tp = h[l * stride].priority; const int tn = h[l * stride].node;
      h[l * stride].priority = h[i * stride].priority; h[l * stride].node = h[i * stride].node;
      h[i * stride].priority = tp; h[i * stride].node = tn;
      i = l; j = 2 * i + 1; k = 2 * i + 2;
    }
  }
};

// per-axis squared distances from x to the region of `node` (query.cxx keeps them in nodeinfo::side_distances)
IA3_HD void kd_side_distances(const KdTree& t, int node, const double* x, const double* root_side, double* side) {
  bool have0 = false, have1 = false, have2 = false;
  side[0] = root_side[0]; side[1] = root_side[1]; side[2] = root_side[2];
  for (int c = node; t.nodes[c].parent >= 0;) {
    const int p = t.nodes[c].parent;
    const int d = t.nodes[p].split_dim;
    const double split = t.nodes[p].split;
    const bool far_side = (x[d] < split) != (t.nodes[p].less == c);
    if (far_side) {
      const double diff = x[d] < split ? split - x[d] : x[d] - split;
      if (d == 0 && !have0) { side[0] = diff * diff; have0 = true; }
      if (d == 1 && !have1) { side[1] = diff * diff; have1 = true; }
      if (d == 2 && !have2) { side[2] = diff * diff; have2 = true; }
    }
    c = p;
  }
}

// tree.query(x, k=1, p=2, eps=0, distance_upper_bound=upper): index of the nearest point, or t.n when none is nearer
// than `upper`.  *d2 (optional): its squared distance.
template <class Queue>
IA3_HD int kd_nearest(const KdTree& t, const double* x, double upper, Queue& q, double* d2 = nullptr) {
  double root_side[3], side[3];
  double mind = 0.0;
  for (int i = 0; i < 3; ++i) {
    double sd = x[i] - t.maxes[i];
    const double s2 = t.mins[i] - x[i];
    if (s2 > sd) sd = s2;
    if (!(sd > 0.0)) sd = 0.0;
    sd = sd * sd;
    mind += sd - 0.0;
    root_side[i] = sd; side[i] = sd;
  }
  double ub = upper * upper;
  int best = t.n;
  double best_d = 0.0;
  int node = 0;
  for (;;) {
    const KdNode nd = t.nodes[node];
    if (nd.split_dim < 0) {
      for (int i = nd.start; i < nd.end; ++i) {
        const int pi = t.indices[i];
        const double* pt = t.data + 3 * (size_t)pi;
        double s = 0.0;
        { const double dd = pt[0] - x[0]; s += dd * dd; }
        { const double dd = pt[1] - x[1]; s += dd * dd; }
        { const double dd = pt[2] - x[2]; s += dd * dd; }
        if (s < ub) { best = pi; best_d = s; ub = s; }   // strictly nearer only: the first of equals stays
      }
      if (q.n == 0) break;
      q.pop(mind, node);
      kd_side_distances(t, node, x, root_side, side);
    } else {
      if (mind > ub) break;   // the nearest open cell is already too far
      const int d = nd.split_dim;
      int near, far;
      double sdist;
      if (x[d] < nd.split) { near = nd.less; far = nd.greater; sdist = nd.split - x[d]; }
      else { near = nd.greater; far = nd.less; sdist = x[d] - nd.split; }
      sdist = sdist * sdist;
      const double far_mind = mind + (sdist - side[d]);
      if (mind > far_mind) {
        // (query.cxx "ensure ni1 is closer than ni2"; the far cell is never nearer without periodic boundaries —
        // kept so the order of operations is the published one)
        if (mind <= ub) q.push(mind, near);
        node = far; mind = far_mind; side[d] = sdist;
      } else {
        if (far_mind <= ub) q.push(far_mind, far);
        node = near;
      }
    }
  }
  if (d2) *d2 = best_d;
  return best;
}

}  // namespace ia3
