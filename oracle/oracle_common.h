/*
 * oracle_common.h -- pieces shared by the CPU oracle's translation units (TEST INFRASTRUCTURE
 * ONLY, see the header of oracle.cpp): restatements of the Eigen / tf2 arithmetic the reference
 * uses and the exact kd-tree standing in for pcl::KdTreeFLANN.
 */
#ifndef DDDMR_ORACLE_COMMON_H_
#define DDDMR_ORACLE_COMMON_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

namespace oracle_detail {

struct F3 { float x, y, z; };

// --------------------------------------------------------------------------
// Small restatements of the Eigen / tf2 pieces the path uses (all double).
// --------------------------------------------------------------------------
struct Affine {      // Eigen::Affine3d: linear part l[r][c] + translation t[r]
  double l[3][3];
  double t[3];
};
struct Quat { double x, y, z, w; };

// Eigen::Quaternion::toRotationMatrix (Eigen/src/Geometry/Quaternion.h).
inline void quat_to_matrix(const Quat& q, double m[3][3]) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  m[0][0] = 1.0 - (tyy + tzz); m[0][1] = txy - twz;         m[0][2] = txz + twy;
  m[1][0] = txy + twz;         m[1][1] = 1.0 - (txx + tzz); m[1][2] = tyz - twx;
  m[2][0] = txz - twy;         m[2][1] = tyz + twx;         m[2][2] = 1.0 - (txx + tyy);
}

// Eigen quaternion from a rotation matrix (quaternionbase_assign_impl<..,3,3>).
inline Quat matrix_to_quat(const double m[3][3]) {
  Quat q;
  double c[4];  // x y z w (Eigen coeffs order)
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    c[3] = 0.5 * t;
    t = 0.5 / t;
    c[0] = (m[2][1] - m[1][2]) * t;
    c[1] = (m[0][2] - m[2][0]) * t;
    c[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    c[3] = (m[k][j] - m[j][k]) * t;
    c[j] = (m[j][i] + m[i][j]) * t;
    c[k] = (m[k][i] + m[i][k]) * t;
  }
  q.x = c[0]; q.y = c[1]; q.z = c[2]; q.w = c[3];
  return q;
}

// tf2::transformToEigen(geometry_msgs::Transform):
// Isometry3d(Translation3d(x,y,z) * Quaterniond(w,x,y,z)).
inline Affine transform_to_eigen(const double p[7]) {
  Affine a;
  Quat q{p[3], p[4], p[5], p[6]};
  quat_to_matrix(q, a.l);
  a.t[0] = p[0]; a.t[1] = p[1]; a.t[2] = p[2];
  return a;
}

// Eigen::AngleAxisd(angle, UnitZ()).toRotationMatrix() -> Affine3d with zero
// translation (dd_simple_trajectory_generator_theory.cpp:416).
inline Affine angle_axis_z(double angle) {
  Affine a;
  const double ax = 0.0, ay = 0.0, az = 1.0;
  const double s = std::sin(angle), c = std::cos(angle);
  const double sx = s * ax, sy = s * ay, sz = s * az;
  const double c1x = (1.0 - c) * ax, c1y = (1.0 - c) * ay, c1z = (1.0 - c) * az;
  double tmp;
  tmp = c1x * ay; a.l[0][1] = tmp - sz; a.l[1][0] = tmp + sz;
  tmp = c1x * az; a.l[0][2] = tmp + sy; a.l[2][0] = tmp - sy;
  tmp = c1y * az; a.l[1][2] = tmp - sx; a.l[2][1] = tmp + sx;
  a.l[0][0] = c1x * ax + c; a.l[1][1] = c1y * ay + c; a.l[2][2] = c1z * az + c;
  a.t[0] = a.t[1] = a.t[2] = 0.0;
  return a;
}

// Affine3d * Affine3d.
inline Affine mul(const Affine& A, const Affine& B) {
  Affine r;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      r.l[i][j] = A.l[i][0] * B.l[0][j] + A.l[i][1] * B.l[1][j] + A.l[i][2] * B.l[2][j];
    r.t[i] = A.l[i][0] * B.t[0] + A.l[i][1] * B.t[1] + A.l[i][2] * B.t[2] + A.t[i];
  }
  return r;
}

// Eigen Transform<double,3,Affine>::inverse(): general 3x3 inverse of the linear
// part by cofactors (Eigen/src/LU/InverseImpl.h, compute_inverse<..,3>), then
// translation = -inv * t.
inline Affine inverse(const Affine& A) {
  Affine r;
  const double (*m)[3] = A.l;
  double cof[3][3];
  cof[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  cof[1][0] = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  cof[2][0] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  const double det = m[0][0] * cof[0][0] + m[0][1] * cof[1][0] + m[0][2] * cof[2][0];
  const double invdet = 1.0 / det;
  cof[0][1] = m[0][2] * m[2][1] - m[0][1] * m[2][2];
  cof[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0];
  cof[2][1] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
  cof[0][2] = m[0][1] * m[1][2] - m[0][2] * m[1][1];
  cof[1][2] = m[0][2] * m[1][0] - m[0][0] * m[1][2];
  cof[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.l[i][j] = cof[i][j] * invdet;
  for (int i = 0; i < 3; ++i)
    r.t[i] = -(r.l[i][0] * A.t[0] + r.l[i][1] * A.t[1] + r.l[i][2] * A.t[2]);
  return r;
}

// tf2::eigenToTransform(Affine3d): translation + Quaterniond(T.linear()).
inline void eigen_to_transform(const Affine& A, double out[7]) {
  out[0] = A.t[0]; out[1] = A.t[1]; out[2] = A.t[2];
  Quat q = matrix_to_quat(A.l);
  out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// tf2::Matrix3x3(q).getEulerYPR(yaw, pitch, roll) -> yaw (solution 1)
// (tf2/LinearMath/Matrix3x3.h: setRotation + getEulerYPR).
inline double tf2_yaw_from_quat(const Quat& q) {
  const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const double s = 2.0 / d;
  const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  double m[3][3] = {{1.0 - (yy + zz), xy - wz, xz + wy},
                    {xy + wz, 1.0 - (xx + zz), yz - wx},
                    {xz - wy, yz + wx, 1.0 - (xx + yy)}};
  if (std::fabs(m[2][0]) >= 1.0) {
    return 0.0;  // gimbal branch: euler_out.yaw = 0
  }
  const double pitch = -std::asin(m[2][0]);
  return std::atan2(m[1][0] / std::cos(pitch), m[0][0] / std::cos(pitch));
}

// --------------------------------------------------------------------------
// Exact kd-tree over float xyz standing in for pcl::KdTreeFLANN (FLANN
// KDTreeSingleIndex, leaf size 15, L2_Simple<float>, eps = 0 => exact search).
// Distances are FLANN's: float, ((dx*dx) + dy*dy) + dz*dz.
// --------------------------------------------------------------------------
inline float l2_simple(const float* a, const float* b) {
  float result = 0.f, diff;
  diff = a[0] - b[0]; result += diff * diff;
  diff = a[1] - b[1]; result += diff * diff;
  diff = a[2] - b[2]; result += diff * diff;
  return result;
}

class KdTree {
 public:
  void build(const float* xyz, size_t n, size_t stride_floats) {
    pts_.resize(n * 3);
    for (size_t i = 0; i < n; ++i) {
      pts_[3 * i + 0] = xyz[i * stride_floats + 0];
      pts_[3 * i + 1] = xyz[i * stride_floats + 1];
      pts_[3 * i + 2] = xyz[i * stride_floats + 2];
    }
    idx_.resize(n);
    std::iota(idx_.begin(), idx_.end(), 0);
    nodes_.clear();
    nodes_.reserve(n / 4 + 16);
    if (n) build_rec(0, n);
  }
  size_t size() const { return idx_.size(); }
  const float* point(int i) const { return &pts_[3 * (size_t)i]; }

  // radiusSearch: all points with dist^2 < r^2 (FLANN RadiusResultSet::addPoint
  // uses a strict comparison; unverified offline -- fixtures stay away from it).
  int radius_search(const float q[3], float radius, std::vector<int>& out) const {
    return radius_search_r2(q, radius * radius, out);
  }
  // same with the squared radius given (PCL: static_cast<float>(radius * radius) of a double radius)
  int radius_search_r2(const float q[3], float r2, std::vector<int>& out) const {
    out.clear();
    if (nodes_.empty()) return 0;
    radius_rec(0, q, r2, out);
    return (int)out.size();
  }
  // nearestKSearch(K=1): returns index, writes float squared distance.
  int nearest(const float q[3], float& best_d2) const {
    if (nodes_.empty()) return -1;
    int best = -1;
    best_d2 = std::numeric_limits<float>::max();
    nearest_rec(0, q, best, best_d2);
    return best;
  }

 private:
  struct Node {
    int left = -1, right = -1;  // children, or -1 for leaf
    int begin = 0, end = 0;     // leaf range in idx_
    int dim = 0;
    float split = 0.f;
    float lo[3], hi[3];         // bounding box
  };
  std::vector<float> pts_;
  std::vector<int> idx_;
  std::vector<Node> nodes_;

  int build_rec(size_t b, size_t e) {
    Node nd;
    for (int d = 0; d < 3; ++d) { nd.lo[d] = std::numeric_limits<float>::max(); nd.hi[d] = -nd.lo[d]; }
    for (size_t i = b; i < e; ++i)
      for (int d = 0; d < 3; ++d) {
        const float v = pts_[3 * (size_t)idx_[i] + d];
        nd.lo[d] = std::min(nd.lo[d], v); nd.hi[d] = std::max(nd.hi[d], v);
      }
    nd.begin = (int)b; nd.end = (int)e;
    const int me = (int)nodes_.size();
    nodes_.push_back(nd);
    if (e - b > 15) {
      int dim = 0;
      float span = nd.hi[0] - nd.lo[0];
      for (int d = 1; d < 3; ++d)
        if (nd.hi[d] - nd.lo[d] > span) { span = nd.hi[d] - nd.lo[d]; dim = d; }
      if (span > 0.f) {
        const size_t mid = (b + e) / 2;
        std::nth_element(idx_.begin() + b, idx_.begin() + mid, idx_.begin() + e,
                         [&](int a, int c) { return pts_[3 * (size_t)a + dim] < pts_[3 * (size_t)c + dim]; });
        const float split = pts_[3 * (size_t)idx_[mid] + dim];
        const int l = build_rec(b, mid);
        const int r = build_rec(mid, e);
        nodes_[me].left = l; nodes_[me].right = r;
        nodes_[me].dim = dim; nodes_[me].split = split;
      }
    }
    return me;
  }
  static float box_d2(const Node& n, const float q[3]) {
    // conservative lower bound in double to never prune a true neighbour
    double d2 = 0.0;
    for (int d = 0; d < 3; ++d) {
      double diff = 0.0;
      if (q[d] < n.lo[d]) diff = (double)n.lo[d] - q[d];
      else if (q[d] > n.hi[d]) diff = (double)q[d] - n.hi[d];
      d2 += diff * diff;
    }
    return (float)(d2 * (1.0 - 1e-6));
  }
  void radius_rec(int ni, const float q[3], float r2, std::vector<int>& out) const {
    const Node& n = nodes_[ni];
    if (box_d2(n, q) >= r2) return;
    if (n.left < 0) {
      for (int i = n.begin; i < n.end; ++i) {
        const int id = idx_[i];
        if (l2_simple(&pts_[3 * (size_t)id], q) < r2) out.push_back(id);
      }
      return;
    }
    radius_rec(n.left, q, r2, out);
    radius_rec(n.right, q, r2, out);
  }
  void nearest_rec(int ni, const float q[3], int& best, float& best_d2) const {
    const Node& n = nodes_[ni];
    if (box_d2(n, q) > best_d2) return;
    if (n.left < 0) {
      for (int i = n.begin; i < n.end; ++i) {
        const int id = idx_[i];
        const float d2 = l2_simple(&pts_[3 * (size_t)id], q);
        if (d2 < best_d2) { best_d2 = d2; best = id; }
      }
      return;
    }
    const bool left_first = q[n.dim] < n.split;
    nearest_rec(left_first ? n.left : n.right, q, best, best_d2);
    nearest_rec(left_first ? n.right : n.left, q, best, best_d2);
  }
};

}  // namespace oracle_detail
#endif
