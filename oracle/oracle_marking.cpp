/*
 * oracle_marking.cpp -- CPU restatement of the global-mode marking / clearing layer.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of oracle.cpp): only tests/ and bench.py's checker legs
 * may use it.  PARITY UNPINNED: the reference holds no fixtures for this path and cannot be built
 * here (ROS 2 / PCL / FLANN / Eigen / tf2 absent).
 *
 * Restates, literally and with the reference's float / double mix:
 *   MultiLayerSpinningLidar::selfClear              plugins/multilayer_spinning_lidar.cpp:456-628
 *   MultiLayerSpinningLidar::selfMark               :306-455
 *   MultiLayerSpinningLidar::getCastingPointCloud   :630-651
 *   MultiLayerSpinningLidar::isinLidarObservation   :682-746
 *   Marking::addPCPtr / removePCPtr /
 *     computeMinDistanceFromObstacle2GroundNodes    plugins/cluster_marking.cpp:49-138
 *   DynamicGraph::setValue / clearValue / initial   src/graph/dynamic_graph.cpp:38-61
 * (paths relative to /root/reference/src/dddmr_perception_3d/), for is_local_planner = false,
 * get_first_tf_ / is_static_layer_ready_ / isAllLayersBeenReset() true.
 *
 * Third-party arithmetic restated from the libraries' published algorithms:
 *   PCL 1.15  EuclideanClusterExtraction (extractEuclideanClusters: seed order, radiusSearch with
 *             static_cast<float>(r * r), indices sorted; extract(): std::sort over reverse iterators
 *             by cluster size), VoxelGrid (floor(p * inverse_leaf) - min_b, voxel index order,
 *             CentroidPoint float accumulation), ProjectInliers / SampleConsensusModelPlane::
 *             projectPoints (normalised Vector4f normal, float dot and subtract; Eigen's SSE
 *             reduction order (a0 + a2) + (a1 + a3) for the 4-float dot product: x86-64 always has
 *             SSE2, so Eigen vectorises fixed-size Vector4f), KdTreeFLANN::radiusSearch.
 *   tf2       quatRotate, Quaternion(axis, angle), Matrix3x3::setRotation / getRotation / getRPY,
 *             Transform::inverse / mult (all double).
 *   angles    shortest_angular_distance = normalize_angle(to - from).
 * Where the reference's behaviour is undefined (radiusSearch on a kd-tree without an input cloud when
 * the last observation holds <= 5 points, :474-480 / :592) the restatement treats the search as
 * returning nothing.
 */
#include <map>
#include <unordered_map>

#include "../include/dddmr_rollout.h"
#include "oracle.h"
#include "oracle_common.h"

using namespace oracle_detail;

namespace {

struct F4 { float x, y, z, i; };
struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };

// ---- tf2 (LinearMath, tf2Scalar = double) -------------------------------------------------------
// operator*(Quaternion, Vector3) and Quaternion::operator*=, inverse() (Quaternion.h)
static Q4 q_mul_v(const Q4& q, const V3& w) {
  return Q4{q.w * w.x + q.y * w.z - q.z * w.y, q.w * w.y + q.z * w.x - q.x * w.z,
            q.w * w.z + q.x * w.y - q.y * w.x, -q.x * w.x - q.y * w.y - q.z * w.z};
}
static Q4 q_mul_q(const Q4& a, const Q4& q) {
  return Q4{a.w * q.x + a.x * q.w + a.y * q.z - a.z * q.y, a.w * q.y + a.y * q.w + a.z * q.x - a.x * q.z,
            a.w * q.z + a.z * q.w + a.x * q.y - a.y * q.x, a.w * q.w - a.x * q.x - a.y * q.y - a.z * q.z};
}
static V3 quat_rotate(const Q4& rotation, const V3& v) {      // tf2::quatRotate
  Q4 q = q_mul_v(rotation, v);
  q = q_mul_q(q, Q4{-rotation.x, -rotation.y, -rotation.z, rotation.w});
  return V3{q.x, q.y, q.z};
}
struct M3 { double m[3][3]; };
static M3 m_set_rotation(const Q4& q) {                         // Matrix3x3::setRotation
  const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const double s = 2.0 / d;
  const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  M3 r;
  r.m[0][0] = 1.0 - (yy + zz); r.m[0][1] = xy - wz; r.m[0][2] = xz + wy;
  r.m[1][0] = xy + wz; r.m[1][1] = 1.0 - (xx + zz); r.m[1][2] = yz - wx;
  r.m[2][0] = xz - wy; r.m[2][1] = yz + wx; r.m[2][2] = 1.0 - (xx + yy);
  return r;
}
static Q4 m_get_rotation(const M3& a) {                         // Matrix3x3::getRotation
  const double trace = a.m[0][0] + a.m[1][1] + a.m[2][2];
  double temp[4];
  if (trace > 0.0) {
    double s = std::sqrt(trace + 1.0);
    temp[3] = s * 0.5;
    s = 0.5 / s;
    temp[0] = (a.m[2][1] - a.m[1][2]) * s;
    temp[1] = (a.m[0][2] - a.m[2][0]) * s;
    temp[2] = (a.m[1][0] - a.m[0][1]) * s;
  } else {
    const int i = a.m[0][0] < a.m[1][1] ? (a.m[1][1] < a.m[2][2] ? 2 : 1) : (a.m[0][0] < a.m[2][2] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = std::sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0);
    temp[i] = s * 0.5;
    s = 0.5 / s;
    temp[3] = (a.m[k][j] - a.m[j][k]) * s;
    temp[j] = (a.m[j][i] + a.m[i][j]) * s;
    temp[k] = (a.m[k][i] + a.m[i][k]) * s;
  }
  return Q4{temp[0], temp[1], temp[2], temp[3]};
}
static M3 m_mul(const M3& a, const M3& b) {                     // operator*(Matrix3x3, Matrix3x3): rows . columns
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = b.m[0][j] * a.m[i][0] + b.m[1][j] * a.m[i][1] + b.m[2][j] * a.m[i][2];
  return r;
}

// geometry_msgs Transform of an Eigen affine: tf2::eigenToTransform (translation + Quaterniond(linear))
struct Tf { V3 t; Q4 q; };
static Tf to_tf(const Affine& A) {
  double p[7];
  eigen_to_transform(A, p);
  return Tf{V3{p[0], p[1], p[2]}, Q4{p[3], p[4], p[5], p[6]}};
}

// ---- PCL VoxelGrid<PointXYZI> with downsample_all_data (leaf as float) -------------------------
static void voxel_grid(std::vector<F4>& pts, float leaf) {
  if (pts.empty()) return;
  const float inv = 1.0f / leaf;
  F4 mn = pts[0], mx = pts[0];
  for (const F4& q : pts) {
    mn.x = std::min(mn.x, q.x); mn.y = std::min(mn.y, q.y); mn.z = std::min(mn.z, q.z);
    mx.x = std::max(mx.x, q.x); mx.y = std::max(mx.y, q.y); mx.z = std::max(mx.z, q.z);
  }
  const int minb[3] = {(int)std::floor(mn.x * inv), (int)std::floor(mn.y * inv), (int)std::floor(mn.z * inv)};
  const int maxb[3] = {(int)std::floor(mx.x * inv), (int)std::floor(mx.y * inv), (int)std::floor(mx.z * inv)};
  const int64_t d0 = maxb[0] - minb[0] + 1, d1 = maxb[1] - minb[1] + 1;
  std::vector<std::pair<int64_t, uint32_t>> order(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) {
    const int64_t i0 = (int64_t)std::floor(pts[i].x * inv) - minb[0];
    const int64_t i1 = (int64_t)std::floor(pts[i].y * inv) - minb[1];
    const int64_t i2 = (int64_t)std::floor(pts[i].z * inv) - minb[2];
    order[i] = {i0 + i1 * d0 + i2 * d0 * d1, (uint32_t)i};
  }
  std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  std::vector<F4> out;
  for (size_t i = 0; i < order.size();) {
    size_t j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    while (j < order.size() && order[j].first == order[i].first) {
      const F4& q = pts[order[j].second];
      sx += q.x; sy += q.y; sz += q.z; si += q.i;
      ++j;
    }
    const float cnt = (float)(j - i);
    out.push_back(F4{sx / cnt, sy / cnt, sz / cnt, si / cnt});
    i = j;
  }
  pts.swap(out);
}

struct PerMarking {                       // perception_3d::per_marking (cluster_marking.h:85-90)
  bool has_pc = false;
  std::vector<F4> pc;
  float mc[4] = {0, 0, 0, 0};
  std::unordered_map<int, float> nodes_of_min_distance;
};

}  // namespace

struct oracle_marking {
  dddmr_marking_config cfg;
  std::vector<float> ground, map_pts;      // xyz
  KdTree kd_ground, kd_map;
  std::vector<double> dgraph;              // DynamicGraph::graph_, keys 0..n_ground
  std::map<int, double> lethal_map;
  std::map<int, std::map<int, std::map<int, PerMarking>>> marking;
  std::vector<float> obs_prev;             // pcl_msg_gbl_ of the last selfMark (xyz)
  // transforms of the current update
  Affine gbl2b, gbl2s;
  Tf tf_gbl2b, tf_gbl2s;
  // diagnostics of the last update
  std::vector<int32_t> dec_voxels;         // [n][3] voxels selfClear decided on
  std::vector<float> dec_margin;           // smallest distance of that decision to one of its thresholds
  std::vector<uint8_t> dec_removed;
  std::vector<int32_t> mark_voxels;        // [n][3] voxel key of every cluster that reached the FOV test
  std::vector<float> mark_margin;
  std::vector<uint8_t> mark_added;
  oracle_marking_stats stats;
};

namespace {

// isinLidarObservation (:682-746); *margin = smallest |angle - threshold| in degrees
bool in_lidar_observation(const oracle_marking& M, const float pc[3], float* margin) {
  const dddmr_marking_config& c = M.cfg;
  const Tf& s = M.tf_gbl2s;
  const V3 n = quat_rotate(s.q, V3{0, 0, 1});
  const double d = -s.t.x * n.x - s.t.y * n.y - s.t.z * n.z;
  const double p2plane = pc[0] * n.x + pc[1] * n.y + pc[2] * n.z + d;
  const double dx = pc[0] - s.t.x, dy = pc[1] - s.t.y, dz = pc[2] - s.t.z;
  const double p2s = std::sqrt(dx * dx + dy * dy + dz * dz);
  const double result = std::asin(p2plane / p2s) * 180.0 / 3.1415926535;
  float mg = (float)std::min(std::fabs(result - c.vertical_FOV_bottom), std::fabs(result - c.vertical_FOV_top));
  if (margin) *margin = mg;
  if (result < c.vertical_FOV_bottom || result > c.vertical_FOV_top) return false;

  const double vx = pc[0] - s.t.x, vy = pc[1] - s.t.y, vz = pc[2] - s.t.z;
  const double unit = std::sqrt(vx * vx + vy * vy + vz * vz);
  const V3 axis{vx / unit, vy / unit, vz / unit};
  const V3 up{1.0, 0.0, 0.0};
  // right_vector = axis.cross(up); right_vector.normalized() discards its result (:716)
  const V3 right{axis.y * up.z - axis.z * up.y, axis.z * up.x - axis.x * up.z, axis.x * up.y - axis.y * up.x};
  const double angle = -1.0 * std::acos(axis.x * up.x + axis.y * up.y + axis.z * up.z);
  // tf2::Quaternion(axis, angle) -> setRotation
  const double len = std::sqrt(right.x * right.x + right.y * right.y + right.z * right.z);
  const double sn = std::sin(angle * 0.5) / len;
  Q4 q{right.x * sn, right.y * sn, right.z * sn, std::cos(angle * 0.5)};
  {                                                              // q.normalize(): *this /= length()
    const double l = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const double inv = 1.0 / l;                                  // operator/= multiplies by 1/s
    q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
  }
  const M3 pointing = m_set_rotation(q);
  const M3 sensor = m_set_rotation(s.q);                         // tf2::fromMsg(trans_gbl2s_)
  M3 inv;                                                        // inverse(): transpose
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) inv.m[i][j] = sensor.m[j][i];
  const M3 rel = m_mul(inv, pointing);                           // mult(): basis product
  const M3 m = m_set_rotation(m_get_rotation(rel));              // Matrix3x3 m(transform.getRotation())
  double yaw;                                                    // getRPY -> getEulerYPR, solution 1
  if (std::fabs(m.m[2][0]) >= 1) {
    yaw = 0;
  } else {
    const double pitch = -std::asin(m.m[2][0]);
    yaw = std::atan2(m.m[1][0] / std::cos(pitch), m.m[0][0] / std::cos(pitch));
  }
  {                                                              // angles::shortest_angular_distance(0.0, yaw)
    const double r = std::fmod(yaw - 0.0 + M_PI, 2.0 * M_PI);
    yaw = r <= 0.0 ? r + M_PI : r - M_PI;
  }
  yaw = yaw * 180.0 / 3.1415926535;
  const double ps = c.scan_effective_positive_start, pe = c.scan_effective_positive_end;
  const double ns = c.scan_effective_negative_start, ne = c.scan_effective_negative_end;
  mg = std::min(mg, (float)std::min(std::min(std::fabs(yaw - ps), std::fabs(yaw - pe)),
                                    std::min(std::min(std::fabs(yaw - ns), std::fabs(yaw - ne)), std::fabs(yaw))));
  if (margin) *margin = mg;
  if (yaw >= 0 && (yaw < ps || yaw > pe)) return false;
  else if (yaw < 0 && (yaw > ns || yaw < ne)) return false;
  else return true;
}

// Marking::computeMinDistanceFromObstacle2GroundNodes (cluster_marking.cpp:49-102)
void min_distance_to_ground_nodes(const oracle_marking& M, const std::vector<F4>& pc, const float mc_in[4],
                                  std::unordered_map<int, float>& nodes) {
  // ProjectInliers(SACMODEL_PLANE): SampleConsensusModelPlane::projectPoints, all points
  float mc[4] = {mc_in[0], mc_in[1], mc_in[2], 0.0f};
  {                                                              // Eigen Vector4f::normalize(): /= norm()
    const float sq = (mc[0] * mc[0] + mc[2] * mc[2]) + (mc[1] * mc[1] + mc[3] * mc[3]);
    const float nrm = std::sqrt(sq);
    for (float& v : mc) v = v / nrm;
  }
  const float tmp_mc[4] = {mc[0], mc[1], mc[2], mc_in[3]};
  std::vector<F4> proj(pc.size());
  for (size_t i = 0; i < pc.size(); ++i) {
    const float p[4] = {pc[i].x, pc[i].y, pc[i].z, 1.0f};
    const float dist = (tmp_mc[0] * p[0] + tmp_mc[2] * p[2]) + (tmp_mc[1] * p[1] + tmp_mc[3] * p[3]);
    proj[i] = F4{p[0] - mc[0] * dist, p[1] - mc[1] * dist, p[2] - mc[2] * dist, pc[i].i};
  }
  voxel_grid(proj, 0.1f);
  const float r2 = static_cast<float>(M.cfg.inflation_radius * M.cfg.inflation_radius);
  std::vector<int> id;
  for (const F4& pt : proj) {
    const float q[3] = {pt.x, pt.y, pt.z};
    if (M.kd_ground.radius_search_r2(q, r2, id)) {
      for (int g : id) {
        const float dx = pt.x - M.ground[3 * (size_t)g + 0];
        const float dy = pt.y - M.ground[3 * (size_t)g + 1];
        const float distance = std::sqrt(dx * dx + dy * dy);     // z dropped on purpose (:86-88)
        auto ins = nodes.insert(std::make_pair(g, distance));
        if (!ins.second) ins.first->second = std::min(ins.first->second, distance);
      }
    }
  }
}

void add_pc(oracle_marking& M, double cx, double cy, double cz, const std::vector<F4>& pc, const float mc[4]) {
  const int x = (int)(cx / M.cfg.xy_resolution), y = (int)(cy / M.cfg.xy_resolution), z = (int)(cz / M.cfg.height_resolution);
  PerMarking& pm = M.marking[x][y][z];
  pm.has_pc = true;
  pm.pc = pc;
  std::memcpy(pm.mc, mc, sizeof(pm.mc));
  std::unordered_map<int, float> nodes;
  min_distance_to_ground_nodes(M, pc, mc, nodes);
  pm.nodes_of_min_distance = nodes;
  for (const auto& kv : nodes) {
    M.dgraph[(size_t)kv.first] = std::min(M.dgraph[(size_t)kv.first], (double)kv.second);   // DynamicGraph::setValue
    if (kv.second <= M.cfg.inscribed_radius) M.lethal_map[kv.first] = kv.second;
  }
}

void remove_pc(oracle_marking& M, PerMarking& pm) {
  for (const auto& kv : pm.nodes_of_min_distance) {
    M.dgraph[(size_t)kv.first] = 9999.0;                         // DynamicGraph::clearValue(key, 9999.0)
    if (kv.second <= M.cfg.inscribed_radius) M.lethal_map.erase(kv.first);
  }
  pm.has_pc = false;
  pm.pc.clear();
}

void self_clear(oracle_marking& M) {
  const dddmr_marking_config& c = M.cfg;
  M.dec_voxels.clear(); M.dec_margin.clear(); M.dec_removed.clear();
  KdTree kd_last;
  const size_t n_prev = M.obs_prev.size() / 3;
  const bool observation_clear = !(n_prev > 5);
  if (!observation_clear) kd_last.build(M.obs_prev.data(), n_prev, 3);
  const V3& tb = M.tf_gbl2b.t;
  const int x_min = (int)((tb.x - c.perception_window_size) / c.xy_resolution);
  const int x_max = (int)((tb.x + c.perception_window_size) / c.xy_resolution);
  const int y_min = (int)((tb.y - c.perception_window_size) / c.xy_resolution);
  const int y_max = (int)((tb.y + c.perception_window_size) / c.xy_resolution);
  const int z_min = (int)((tb.z - c.marking_height) / c.height_resolution);
  const int z_max = (int)((tb.z + c.marking_height) / c.height_resolution);
  const V3 st{M.gbl2s.t[0], M.gbl2s.t[1], M.gbl2s.t[2]};          // trans_gbl2s_af3_.translation()
  std::vector<int> id;
  auto it_x_min = M.marking.lower_bound(x_min), it_x_max = M.marking.lower_bound(x_max);
  if (it_x_min == M.marking.end() && it_x_min == it_x_max) return;
  for (auto it_x = it_x_min; it_x != it_x_max; ++it_x) {
    auto it_y_min = it_x->second.lower_bound(y_min), it_y_max = it_x->second.lower_bound(y_max);
    if (it_y_min == it_x->second.end() && it_y_min == it_y_max) continue;
    for (auto it_y = it_y_min; it_y != it_y_max; ++it_y) {
      if (it_y->second.empty()) continue;
      auto it_z_min = it_y->second.lower_bound(z_min), it_z_max = it_y->second.lower_bound(z_max);
      if (it_z_min == it_y->second.end() && it_z_min == it_z_max) continue;
      for (auto it_z = it_z_min; it_z != it_z_max; ++it_z) {
        if (!it_z->second.has_pc) continue;
        M.stats.n_in_window++;
        float pt[3];
        pt[0] = (float)(it_x->first * c.xy_resolution);
        pt[1] = (float)(it_y->first * c.xy_resolution);
        pt[2] = (float)(it_z->first * c.height_resolution);
        float margin = 1e30f;
        bool removed = false;
        float fov_margin;
        if (!in_lidar_observation(M, pt, &fov_margin)) {
          margin = fov_margin;                                    // kept: outside the sensor's view
        } else {
          margin = fov_margin;
          bool skip_clear = false;
          if (!observation_clear) {
            // getCastingPointCloud (:630-651): a line of points every 5 cm from the sensor to the voxel
            const float dX = (float)(pt[0] - st.x), dY = (float)(pt[1] - st.y), dZ = (float)(pt[2] - st.z);
            float distance = std::sqrt(dX * dX + dY * dY + dZ * dZ);
            distance = (float)(distance / 0.05);
            const float dt = 1 / distance;
            for (float t = 0; t <= 1.0; t += dt) {
              float a[3];
              a[0] = (float)(st.x + dX * t);
              a[1] = (float)(st.y + dY * t);
              a[2] = (float)(st.z + dZ * t);
              const double ddx = pt[0] - a[0], ddy = pt[1] - a[1], ddz = pt[2] - a[2];   // getDistanceBTWPoints
              const float intensity = (float)std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
              margin = std::min(margin, std::fabs(intensity - 0.05f));
              if (intensity < 0.05) break;                        // last 5 cm ignored (:563-564)
              double search_distance = intensity / 20. + 0.01;
              search_distance = std::min(search_distance, 0.1);
              const float r2 = static_cast<float>(search_distance * search_distance);
              // margin of the ray probe: distance of the nearest observation point from the probe sphere
              float nd2;
              const int nn = kd_last.nearest(a, nd2);
              if (nn >= 0) margin = std::min(margin, std::fabs(std::sqrt(nd2) - (float)search_distance));
              if (kd_last.radius_search_r2(a, r2, id) > 0) { skip_clear = true; break; }
            }
          }
          if (!skip_clear) {
            int cnt = 0;
            if (!observation_clear) {
              const float r2 = static_cast<float>(c.xy_resolution * c.xy_resolution);
              cnt = kd_last.radius_search_r2(pt, r2, id);
              // margin: how far the 2nd nearest point is from the resolution sphere
              std::vector<float> d2s;
              std::vector<int> all;
              kd_last.radius_search_r2(pt, r2 * 4.0f, all);
              for (int g : all) d2s.push_back(l2_simple(kd_last.point(g), pt));
              std::sort(d2s.begin(), d2s.end());
              if (d2s.size() >= 2) margin = std::min(margin, std::fabs(std::sqrt(d2s[1]) - (float)c.xy_resolution));
            }
            if (!(cnt > 1)) {
              remove_pc(M, it_z->second);
              removed = true;
              M.stats.n_cleared++;
            }
          }
        }
        M.dec_voxels.push_back(it_x->first); M.dec_voxels.push_back(it_y->first); M.dec_voxels.push_back(it_z->first);
        M.dec_margin.push_back(margin);
        M.dec_removed.push_back(removed ? 1 : 0);
      }
    }
  }
}

void self_mark(oracle_marking& M, const float* obs, size_t n) {
  const dddmr_marking_config& c = M.cfg;
  M.mark_voxels.clear(); M.mark_margin.clear(); M.mark_added.clear();
  if (n <= 5) return;                                             // pcl_msg_->points.size() <= 5 (:320-321)
  M.obs_prev.assign(obs, obs + 3 * n);                            // pcl_msg_gbl_ (:322-324)
  M.stats.n_observation = (uint32_t)n;
  KdTree kd;
  kd.build(obs, n, 3);
  // pcl::extractEuclideanClusters
  const float tol2 = static_cast<float>(c.euclidean_cluster_extraction_tolerance * c.euclidean_cluster_extraction_tolerance);
  std::vector<std::vector<int>> clusters;
  {
    std::vector<uint8_t> processed(n, 0);
    std::vector<int> nn;
    for (size_t i = 0; i < n; ++i) {
      if (processed[i]) continue;
      std::vector<int> seed;
      size_t sq = 0;
      seed.push_back((int)i);
      processed[i] = 1;
      while (sq < seed.size()) {
        if (!kd.radius_search_r2(obs + 3 * (size_t)seed[sq], tol2, nn)) { ++sq; continue; }
        for (int j : nn) {
          if (processed[(size_t)j]) continue;
          seed.push_back(j);
          processed[(size_t)j] = 1;
        }
        ++sq;
      }
      if (seed.size() >= (size_t)std::max(0, c.euclidean_cluster_extraction_min_cluster_size) && seed.size() <= n) {
        std::sort(seed.begin(), seed.end());
        clusters.push_back(seed);
      }
    }
    // EuclideanClusterExtraction::extract: std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters)
    std::sort(clusters.rbegin(), clusters.rend(),
              [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() < b.size(); });
  }
  M.stats.n_clusters = (uint32_t)clusters.size();
  float intensity_cnt = 100;
  std::vector<int> id;
  for (const auto& idx : clusters) {
    std::vector<F4> cloud_cluster;
    float cx = 0.f, cy = 0.f, cz = 0.f;                           // pcl::PointXYZI centroid: x = y = z = 0
    for (int p : idx) {
      const F4 ip{obs[3 * (size_t)p], obs[3 * (size_t)p + 1], obs[3 * (size_t)p + 2], intensity_cnt};
      cx += ip.x; cy += ip.y; cz += ip.z;
      cloud_cluster.push_back(ip);
    }
    intensity_cnt += 100;
    const float sz = (float)idx.size();                           // float /= size_t
    cx /= sz; cy /= sz; cz /= sz;
    const float centroid[3] = {cx, cy, cz};
    float margin = 1e30f;
    {
      float nd2;
      if (M.kd_ground.nearest(centroid, nd2) >= 0) margin = std::min(margin, std::fabs(std::sqrt(nd2) - 0.05f));
      if (M.kd_map.size() && M.kd_map.nearest(centroid, nd2) >= 0) margin = std::min(margin, std::fabs(std::sqrt(nd2) - 0.1f));
    }
    // cluster centre attached to the ground -> ignore (:364-368)
    if (M.kd_ground.radius_search_r2(centroid, static_cast<float>(0.05 * 0.05), id)) continue;
    voxel_grid(cloud_cluster, 0.2f);
    size_t hit = 0;
    if (c.segmentation_ignore_ratio <= 0.999) {
      for (size_t a = 0; a < cloud_cluster.size(); ++a) {         // (searches with the CENTROID every time, :380)
        if (M.kd_map.radius_search_r2(centroid, static_cast<float>(0.1 * 0.1), id)) {
          hit++;
          if (hit > cloud_cluster.size() * c.segmentation_ignore_ratio) break;
        }
      }
    }
    if (hit <= cloud_cluster.size() * c.segmentation_ignore_ratio) {
      const Q4 rot = M.tf_gbl2b.q;
      const V3 nrm = quat_rotate(rot, V3{0, 0, 1});
      float mc[4];
      mc[0] = (float)nrm.x; mc[1] = (float)nrm.y; mc[2] = (float)nrm.z;
      const double d = -M.tf_gbl2b.t.x * nrm.x - M.tf_gbl2b.t.y * nrm.y - M.tf_gbl2b.t.z * nrm.z;
      mc[3] = (float)d;
      float vc[3];
      vc[0] = (float)((int)(cx / c.xy_resolution) * c.xy_resolution);
      vc[1] = (float)((int)(cy / c.xy_resolution) * c.xy_resolution);
      vc[2] = (float)((int)(cz / c.height_resolution) * c.height_resolution);
      // margin of the voxel key: distance of centroid / resolution from the next integer
      for (int a = 0; a < 3; ++a) {
        const double v = centroid[a] / (a < 2 ? c.xy_resolution : c.height_resolution);
        const double fr = std::fabs(v - std::nearbyint(v));
        margin = std::min(margin, (float)(fr * (a < 2 ? c.xy_resolution : c.height_resolution)));
      }
      float fov_margin;
      const bool in = in_lidar_observation(M, vc, &fov_margin);
      margin = std::min(margin, fov_margin);
      M.mark_voxels.push_back((int)(cx / c.xy_resolution));
      M.mark_voxels.push_back((int)(cy / c.xy_resolution));
      M.mark_voxels.push_back((int)(cz / c.height_resolution));
      M.mark_margin.push_back(margin);
      M.mark_added.push_back(in ? 1 : 0);
      if (in) {
        add_pc(M, cx, cy, cz, cloud_cluster, mc);
        M.stats.n_marked++;
      }
    }
  }
}

}  // namespace

extern "C" {

oracle_marking* oracle_marking_create(const dddmr_marking_config* cfg, const float* ground_xyz, size_t n_ground,
                                      size_t ground_stride_bytes, const float* map_xyz, size_t n_map,
                                      size_t map_stride_bytes) {
  if (!cfg) return nullptr;
  auto* M = new oracle_marking();
  M->cfg = *cfg;
  const size_t gs = ground_stride_bytes / sizeof(float), ms = map_stride_bytes / sizeof(float);
  M->ground.resize(3 * n_ground);
  for (size_t i = 0; i < n_ground; ++i)
    for (int a = 0; a < 3; ++a) M->ground[3 * i + a] = ground_xyz[i * gs + a];
  M->map_pts.resize(3 * n_map);
  for (size_t i = 0; i < n_map; ++i)
    for (int a = 0; a < 3; ++a) M->map_pts[3 * i + a] = map_xyz[i * ms + a];
  M->kd_ground.build(M->ground.data(), n_ground, 3);
  M->kd_map.build(M->map_pts.data(), n_map, 3);
  M->dgraph.assign(n_ground + 1, cfg->max_obstacle_distance);     // DynamicGraph::initial: i = 0..n inclusive
  std::memset(&M->stats, 0, sizeof(M->stats));
  return M;
}

void oracle_marking_destroy(oracle_marking* M) { delete M; }

void oracle_marking_reset(oracle_marking* M) {
  if (!M) return;
  M->marking.clear();
  M->lethal_map.clear();
  M->dgraph.assign(M->dgraph.size(), M->cfg.max_obstacle_distance);
}

// One StackedPerception::doClear_then_Mark pass of the lidar plugin (stacked_perception.cpp:72-90):
// obs_gbl = pcl_msg_gbl_ of THIS update (n points xyz, global frame).
int oracle_marking_update(oracle_marking* M, const float* obs_gbl_xyz, size_t n, const double T_base_sensor[7],
                          const double T_gbl_base[7], oracle_marking_stats* stats) {
  if (!M) return -1;
  std::memset(&M->stats, 0, sizeof(M->stats));
  const Affine b2s = transform_to_eigen(T_base_sensor);
  M->gbl2b = transform_to_eigen(T_gbl_base);
  M->gbl2s = mul(M->gbl2b, b2s);                                  // trans_gbl2s_af3_ (:236-237)
  M->tf_gbl2s = to_tf(M->gbl2s);                                  // tf2::eigenToTransform (:238)
  M->tf_gbl2b = Tf{V3{T_gbl_base[0], T_gbl_base[1], T_gbl_base[2]},
                   Q4{T_gbl_base[3], T_gbl_base[4], T_gbl_base[5], T_gbl_base[6]}};
  self_clear(*M);
  self_mark(*M, obs_gbl_xyz, n);
  size_t alive = 0;
  for (auto& x : M->marking)
    for (auto& y : x.second)
      for (auto& z : y.second) alive += z.second.has_pc ? 1 : 0;
  M->stats.n_alive = (uint32_t)alive;
  if (stats) *stats = M->stats;
  return 0;
}

size_t oracle_marking_get_voxels(oracle_marking* M, int32_t* xyz_out, size_t capacity) {
  size_t n = 0;
  for (auto& x : M->marking)
    for (auto& y : x.second)
      for (auto& z : y.second) {
        if (!z.second.has_pc) continue;
        if (xyz_out && n < capacity) { xyz_out[3 * n] = x.first; xyz_out[3 * n + 1] = y.first; xyz_out[3 * n + 2] = z.first; }
        ++n;
      }
  return n;
}

// generator points of every alive marking: pm.pc projected + 0.1 m VoxelGrid as min_distance_to_ground_nodes does
size_t oracle_marking_get_points(oracle_marking* M, float* xyz_out, int32_t* voxel_out, size_t capacity) {
  size_t n = 0;
  for (auto& x : M->marking)
    for (auto& y : x.second)
      for (auto& z : y.second) {
        const PerMarking& pm = z.second;
        if (!pm.has_pc) continue;
        float mc[4] = {pm.mc[0], pm.mc[1], pm.mc[2], 0.0f};
        const float nrm = std::sqrt((mc[0] * mc[0] + mc[2] * mc[2]) + (mc[1] * mc[1] + mc[3] * mc[3]));
        for (float& v : mc) v = v / nrm;
        const float tmp_mc[4] = {mc[0], mc[1], mc[2], pm.mc[3]};
        std::vector<F4> proj(pm.pc.size());
        for (size_t i = 0; i < pm.pc.size(); ++i) {
          const float p[4] = {pm.pc[i].x, pm.pc[i].y, pm.pc[i].z, 1.0f};
          const float dist = (tmp_mc[0] * p[0] + tmp_mc[2] * p[2]) + (tmp_mc[1] * p[1] + tmp_mc[3] * p[3]);
          proj[i] = F4{p[0] - mc[0] * dist, p[1] - mc[1] * dist, p[2] - mc[2] * dist, pm.pc[i].i};
        }
        voxel_grid(proj, 0.1f);
        for (const F4& q : proj) {
          if (xyz_out && n < capacity) { xyz_out[3 * n] = q.x; xyz_out[3 * n + 1] = q.y; xyz_out[3 * n + 2] = q.z; }
          if (voxel_out && n < capacity) { voxel_out[3 * n] = x.first; voxel_out[3 * n + 1] = y.first; voxel_out[3 * n + 2] = z.first; }
          ++n;
        }
      }
  return n;
}

size_t oracle_marking_get_dgraph(oracle_marking* M, double* out, size_t capacity) {
  const size_t n = std::min(capacity, M->dgraph.size());
  if (out) std::memcpy(out, M->dgraph.data(), n * sizeof(double));
  return M->dgraph.size();
}

size_t oracle_marking_get_lethal(oracle_marking* M, uint8_t* flags, size_t capacity) {
  if (flags) std::memset(flags, 0, capacity);
  for (auto& kv : M->lethal_map)
    if (flags && (size_t)kv.first < capacity) flags[kv.first] = 1;
  return M->lethal_map.size();
}

// decisions of the last update: which = 0 selfClear (voxel, margin, removed), 1 selfMark (voxel, margin, added)
size_t oracle_marking_get_decisions(oracle_marking* M, int which, int32_t* voxels, float* margins, uint8_t* flags,
                                    size_t capacity) {
  const auto& v = which == 0 ? M->dec_voxels : M->mark_voxels;
  const auto& m = which == 0 ? M->dec_margin : M->mark_margin;
  const auto& f = which == 0 ? M->dec_removed : M->mark_added;
  const size_t n = m.size();
  for (size_t i = 0; i < n && i < capacity; ++i) {
    if (voxels) { voxels[3 * i] = v[3 * i]; voxels[3 * i + 1] = v[3 * i + 1]; voxels[3 * i + 2] = v[3 * i + 2]; }
    if (margins) margins[i] = m[i];
    if (flags) flags[i] = f[i];
  }
  return n;
}

// isinLidarObservation alone (unit tests of the FOV logic)
int oracle_in_lidar_observation(const dddmr_marking_config* cfg, const double T_base_sensor[7], const double T_gbl_base[7],
                                const float* pts_xyz, size_t n, uint8_t* inside, float* margin) {
  oracle_marking M;
  M.cfg = *cfg;
  M.gbl2b = transform_to_eigen(T_gbl_base);
  M.gbl2s = mul(M.gbl2b, transform_to_eigen(T_base_sensor));
  M.tf_gbl2s = to_tf(M.gbl2s);
  for (size_t i = 0; i < n; ++i) {
    float mg;
    inside[i] = in_lidar_observation(M, pts_xyz + 3 * i, &mg) ? 1 : 0;
    if (margin) margin[i] = mg;
  }
  return 0;
}

}  // extern "C"
