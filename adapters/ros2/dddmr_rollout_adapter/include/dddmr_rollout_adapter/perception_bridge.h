// perception_bridge.h -- the perception_3d side of the boundary: what the patches under
// adapters/ros2/patches/ call from the reference's sensor plugins.
//
//   MultiLayerSpinningLidar::cbSensor, local mode   (plugins/multilayer_spinning_lidar.cpp:177-281)  -> feedScan()
//   ... selfClear + selfMark, global mode           (:306-628; StackedPerception::doClear_then_Mark,
//                                                    src/stacked_perception.cpp:72-90)               -> MarkingLayerBridge
//   ... get_dGraphValue / updateLethalPointCloud    (:838-841, :283-304)                             -> MarkingLayerBridge
//   PathBlockedStrategy::selfMark                   (plugins/path_blocked_strategy.cpp:56-100)       -> pathBlocked()
//
// Everything here is a template over the ROS / PCL types it is handed (geometry_msgs TransformStamped,
// pcl::PointCloud<...>): this header includes neither, so it is syntax-checked in a plain C++ toolchain
// (tests/test_adapters_cpu.py) and instantiated with the real types inside a dddmr_navigation workspace.
#ifndef DDDMR_ROLLOUT_ADAPTER_PERCEPTION_BRIDGE_H_
#define DDDMR_ROLLOUT_ADAPTER_PERCEPTION_BRIDGE_H_

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "dddmr_rollout.h"
#include "dddmr_rollout_adapter/shared_context.h"

namespace dddmr_rollout_adapter
{

// geometry_msgs::msg::TransformStamped -> x y z qx qy qz qw
template<class TransformStamped>
inline void toPose7(const TransformStamped & t, double out[7])
{
  out[0] = t.transform.translation.x; out[1] = t.transform.translation.y; out[2] = t.transform.translation.z;
  out[3] = t.transform.rotation.x; out[4] = t.transform.rotation.y; out[5] = t.transform.rotation.z;
  out[6] = t.transform.rotation.w;
}

// cbSensor, is_local_planner_ = true: the raw scan in the SENSOR frame (pcl::fromROSMsg output, before any of the
// reference's transforms / PassThrough / VoxelGrid passes) goes to the device, which does all of them and keeps
// the result as the aggregate observation.  Returns the library's code; on DDDMR_OK the caller skips the CPU
// passes of this callback and SharedContext::noteDeviceFeed() tells the planner not to upload a CPU aggregate.
template<class Cloud, class TransformStamped>
inline int feedScan(
  dddmr_rollout_ctx * ctx, const Cloud & scan_sensor_frame, const TransformStamped & trans_b2s,
  const TransformStamped & trans_gbl2b, double perception_window_size, double marking_height, int stitcher_num,
  uint32_t * n_out = nullptr)
{
  if (!ctx) {return DDDMR_ERR_BAD_ARG;}
  double b2s[7], g2b[7];
  toPose7(trans_b2s, b2s);
  toPose7(trans_gbl2b, g2b);
  int rc = dddmr_rollout_set_stitcher(ctx, stitcher_num > 0 ? stitcher_num : 0);    // (the library keeps the deque of raw scans)
  if (rc != DDDMR_OK) {return rc;}
  const size_t n = scan_sensor_frame.points.size();
  rc = dddmr_rollout_set_scan(
    ctx, n ? &scan_sensor_frame.points[0].x : nullptr, n, sizeof(scan_sensor_frame.points[0]), b2s, g2b,
    perception_window_size, marking_height, n_out);
  if (rc == DDDMR_OK) {SharedContext::noteDeviceFeed();}
  return rc;
}

// The same for plugin number `source` of several sensor plugins (its position in perception_3d's `plugins:` list,
// 0 .. DDDMR_MAX_SOURCES - 1): the device keeps one feed per source and publishes their concatenation in source order,
// which is StackedPerception::aggregateObservations (src/stacked_perception.cpp:128-140).
template<class Cloud, class TransformStamped>
inline int feedScanSource(
  dddmr_rollout_ctx * ctx, int source, const Cloud & scan_sensor_frame, const TransformStamped & trans_b2s,
  const TransformStamped & trans_gbl2b, double perception_window_size, double marking_height, int stitcher_num,
  uint32_t * n_source_out = nullptr, uint32_t * n_aggregate_out = nullptr)
{
  if (!ctx) {return DDDMR_ERR_BAD_ARG;}
  double b2s[7], g2b[7];
  toPose7(trans_b2s, b2s);
  toPose7(trans_gbl2b, g2b);
  int rc = dddmr_rollout_set_stitcher_source(ctx, source, stitcher_num > 0 ? stitcher_num : 0);
  if (rc != DDDMR_OK) {return rc;}
  const size_t n = scan_sensor_frame.points.size();
  rc = dddmr_rollout_set_scan_source(
    ctx, source, n ? &scan_sensor_frame.points[0].x : nullptr, n, sizeof(scan_sensor_frame.points[0]), b2s, g2b,
    perception_window_size, marking_height, n_source_out, n_aggregate_out);
  if (rc == DDDMR_OK) {SharedContext::noteDeviceFeed();}
  return rc;
}

// PathBlockedStrategy::selfMark on the device's aggregate observation.  pcl_prune_plan is
// shared_data_->pcl_prune_plan_ (pcl::PointXYZI, 32-byte records: repacked to x y z intensity).
template<class PlanCloud>
inline int pathBlocked(
  dddmr_rollout_ctx * ctx, const PlanCloud & pcl_prune_plan, double check_radius, double * blocked_ratio_percent,
  bool * path_blocked_wait)
{
  if (!ctx) {return DDDMR_ERR_BAD_ARG;}
  std::vector<float> xyzi(4 * pcl_prune_plan.points.size());
  for (size_t i = 0; i < pcl_prune_plan.points.size(); ++i) {
    const auto & p = pcl_prune_plan.points[i];
    xyzi[4 * i] = p.x; xyzi[4 * i + 1] = p.y; xyzi[4 * i + 2] = p.z; xyzi[4 * i + 3] = p.intensity;
  }
  int32_t opinion = DDDMR_OPINION_PASS;
  const int rc = dddmr_rollout_path_blocked(
    ctx, xyzi.data(), pcl_prune_plan.points.size(), check_radius, blocked_ratio_percent, &opinion, nullptr);
  if (rc == DDDMR_OK && path_blocked_wait) {*path_blocked_wait = opinion == DDDMR_OPINION_PATH_BLOCKED_WAIT;}
  return rc;
}

// The global-mode marking / clearing layer of ONE lidar plugin instance.  Mirrors what the plugin keeps:
// pct_marking_ (store + lethal_map_) and dGraph_ live on the device; host copies of the dGraph and the lethal set
// are refreshed after every update, because get_dGraphValue() is called per ground node by the global planner's
// A* (perception_3d_ros.cpp get_min_dGraphValue) and must not cost a device round trip each.
class MarkingLayerBridge
{
public:
  // resetdGraph (:831-839) / first use: pcl_ground = shared_data_->pcl_ground_ (static_ground_size_ nodes),
  // pcl_map = shared_data_->pcl_map_.  Parameters as the plugin read them in onInitialize (:58-170).
  template<class GroundCloud, class MapCloud>
  int create(
    dddmr_rollout_ctx * ctx, const dddmr_marking_config & cfg, const GroundCloud & pcl_ground, size_t static_ground_size,
    const MapCloud & pcl_map)
  {
    ctx_ = ctx;
    n_ground_ = static_ground_size;
    const size_t nm = pcl_map.points.size();
    const int rc = dddmr_rollout_marking_create(
      ctx, &cfg, n_ground_ ? &pcl_ground.points[0].x : nullptr, n_ground_, sizeof(pcl_ground.points[0]),
      nm ? &pcl_map.points[0].x : nullptr, nm, nm ? sizeof(pcl_map.points[0]) : 16);
    if (rc != DDDMR_OK) {return rc;}
    dgraph_.assign(n_ground_ + 1, cfg.max_obstacle_distance);
    lethal_.assign(n_ground_ + 1, 0);
    return DDDMR_OK;
  }
  bool ready() const {return ctx_ != nullptr;}

  // resetdGraph after create
  int reset(double max_obstacle_distance)
  {
    const int rc = dddmr_rollout_marking_reset(ctx_);
    if (rc == DDDMR_OK) {
      dgraph_.assign(n_ground_ + 1, max_obstacle_distance);
      lethal_.assign(n_ground_ + 1, 0);
    }
    return rc;
  }

  // one doClear_then_Mark pass: selfClear against the previous observation, selfMark of the observation the last
  // feedScan left on the device; then the host copies are refreshed
  template<class TransformStamped>
  int clearThenMark(const TransformStamped & trans_b2s, const TransformStamped & trans_gbl2b, dddmr_marking_stats * stats = nullptr)
  {
    double b2s[7], g2b[7];
    toPose7(trans_b2s, b2s);
    toPose7(trans_gbl2b, g2b);
    int rc = dddmr_rollout_marking_update(ctx_, b2s, g2b, stats);
    if (rc != DDDMR_OK) {return rc;}
    rc = dddmr_rollout_marking_get_dgraph(ctx_, dgraph_.data(), dgraph_.size());
    if (rc != DDDMR_OK) {return rc;}
    return dddmr_rollout_marking_get_lethal(ctx_, lethal_.data(), lethal_.size());
  }

  // Marking::get_dGraphValue(index)
  double dGraphValue(unsigned int index) const {return index < dgraph_.size() ? dgraph_[index] : 9999.0;}

  // updateLethalPointCloud (:283-304): the ground nodes of lethal_map_ as points of `out` (pcl::PointXYZI cloud)
  template<class GroundCloud, class LethalCloud>
  void lethalPointCloud(const GroundCloud & pcl_ground, LethalCloud & out) const
  {
    for (size_t i = 0; i < lethal_.size() && i < pcl_ground.points.size(); ++i) {
      if (!lethal_[i]) {continue;}
      typename LethalCloud::PointType ipt;
      ipt.x = pcl_ground.points[i].x; ipt.y = pcl_ground.points[i].y; ipt.z = pcl_ground.points[i].z;
      out.push_back(ipt);
    }
  }

  // the `global_marking` topic (pubUpdateLoop, :755-775): the reference publishes every alive marking's stored cluster
  // (its 0.2 m-downsampled points); the device stores what the dGraph is computed from instead -- the same clusters
  // projected on the robot's ground plane at 0.1 m -- so the published cloud is that: same footprints, z on the plane.
  // Fetched on demand only (call it when the topic has subscribers).
  template<class MarkingCloud>
  int markingPointCloud(MarkingCloud & out) const
  {
    size_t n = 0;
    int rc = dddmr_rollout_marking_get_points(ctx_, nullptr, nullptr, 0, &n);
    if (rc != DDDMR_OK || n == 0) {return rc;}
    std::vector<float> xyz(3 * n);
    rc = dddmr_rollout_marking_get_points(ctx_, xyz.data(), nullptr, n, &n);
    if (rc != DDDMR_OK) {return rc;}
    for (size_t i = 0; i < n; ++i) {
      typename MarkingCloud::PointType ipt;
      ipt.x = xyz[3 * i]; ipt.y = xyz[3 * i + 1]; ipt.z = xyz[3 * i + 2];
      out.push_back(ipt);
    }
    return DDDMR_OK;
  }

private:
  dddmr_rollout_ctx * ctx_ = nullptr;
  size_t n_ground_ = 0;
  std::vector<double> dgraph_;
  std::vector<uint8_t> lethal_;
};

// dddmr_marking_config from the plugin's members (names as in multilayer_spinning_lidar.h); capacities sized from the
// map: every ground node can carry a handful of markings over a long drive
inline dddmr_marking_config markingConfig(
  double resolution, double height_resolution, double marking_height, double perception_window_size,
  double vertical_FOV_top, double vertical_FOV_bottom, double scan_effective_positive_start,
  double scan_effective_positive_end, double scan_effective_negative_start, double scan_effective_negative_end,
  double euclidean_cluster_extraction_tolerance, int euclidean_cluster_extraction_min_cluster_size,
  double segmentation_ignore_ratio, double inscribed_radius, double inflation_radius, double max_obstacle_distance,
  size_t static_ground_size)
{
  dddmr_marking_config c;
  std::memset(&c, 0, sizeof(c));
  c.xy_resolution = resolution; c.height_resolution = height_resolution;
  c.marking_height = marking_height; c.perception_window_size = perception_window_size;
  c.vertical_FOV_top = vertical_FOV_top; c.vertical_FOV_bottom = vertical_FOV_bottom;
  c.scan_effective_positive_start = scan_effective_positive_start; c.scan_effective_positive_end = scan_effective_positive_end;
  c.scan_effective_negative_start = scan_effective_negative_start; c.scan_effective_negative_end = scan_effective_negative_end;
  c.euclidean_cluster_extraction_tolerance = euclidean_cluster_extraction_tolerance;
  c.euclidean_cluster_extraction_min_cluster_size = euclidean_cluster_extraction_min_cluster_size;
  c.segmentation_ignore_ratio = segmentation_ignore_ratio;
  c.inscribed_radius = inscribed_radius; c.inflation_radius = inflation_radius;
  c.max_obstacle_distance = max_obstacle_distance;
  size_t markings = 1u << 15;
  while (markings < 4 * static_ground_size && markings < (1u << 22)) {markings <<= 1;}
  c.max_markings = static_cast<uint32_t>(markings);
  c.max_cluster_points = static_cast<uint32_t>(markings * 16 > (1u << 26) ? (1u << 26) : markings * 16);
  return c;
}

}  // namespace dddmr_rollout_adapter
#endif
