// TEST INFRASTRUCTURE — host build of the product's seed tree (csrc/kdtree.cpp + ia3_kdtree.h) so its layout and its
// queries can be compared with scipy.spatial.cKDTree on the CPU.  Not shipped, not a fallback.
#include <vector>
#include <cstring>
#include "../../imageanalysis3_amd/csrc/kdtree.cpp"

static std::vector<ia3::KdNode> g_nodes;
static std::vector<int> g_idx;
static std::vector<double> g_pts;
static ia3::KdTree g_tree;

extern "C" int ia3cpu_kd_build(const double* pts, int n, int* idx_out, double* nodes_out, int cap) {
  g_pts.assign(pts, pts + 3 * (size_t)n);
  ia3k::kd_build(g_pts.data(), n, g_nodes, g_idx, g_tree.mins, g_tree.maxes);
  g_tree.nodes = g_nodes.data(); g_tree.indices = g_idx.data(); g_tree.data = g_pts.data(); g_tree.n = n;
  if (idx_out) memcpy(idx_out, g_idx.data(), sizeof(int) * (size_t)n);
  int k = 0;
  for (const auto& nd : g_nodes) {   // [split_dim, split, start, end, less, greater, parent] per node, creation order
    if (k >= cap) break;
    double* o = nodes_out + 7 * (size_t)k++;
    o[0] = nd.split_dim; o[1] = nd.split; o[2] = nd.start; o[3] = nd.end; o[4] = nd.less; o[5] = nd.greater; o[6] = nd.parent;
  }
  return (int)g_nodes.size();
}

// queue entries laid out [entry][lane] with `stride` lanes, as the kernel keeps them in LDS (stride 1 = plain)
extern "C" int ia3cpu_kd_query(const double* xs, int nq, double upper, int stride, int cap, int* out_idx, double* out_d2,
                               int* max_queue) {
  std::vector<ia3::KdQEntry> store((size_t)stride * (size_t)cap);
  int overflow = 0;
  for (int i = 0; i < nq; ++i) {
    ia3::KdQueue<ia3::KdQEntry*> q(store.data() + (i % stride), stride, cap);
    double d2 = 0;
    out_idx[i] = ia3::kd_nearest(g_tree, xs + 3 * (size_t)i, upper, q, &d2);
    out_d2[i] = d2;
    if (q.overflow) { overflow = 1; out_idx[i] = -1; }
  }
  (void)max_queue;
  return overflow;
}
