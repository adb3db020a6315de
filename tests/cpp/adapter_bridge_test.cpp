// Unit test of the ROS-side binding logic (adapters/ros2/dddmr_rollout_adapter/include/dddmr_rollout_adapter/
// planner_bridge.h, perception_bridge.h, shared_context.h) WITHOUT ROS, PCL or a GPU: the bridges are templates over the
// message / cloud types, instantiated here with stand-ins that have the same members, against a fake C-ABI that records
// the calls and returns programmed codes.  What is checked is what VERDICT r2 found broken in the patched planner:
// every return code is looked at, a rejected observation never lets the tick run, and a failed tick is reported as such.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dddmr_rollout_adapter/perception_bridge.h"
#include "dddmr_rollout_adapter/planner_bridge.h"

// ---- stand-ins with the members the bridges touch ----
struct V3 { double x = 0, y = 0, z = 0; };
struct Q4 { double x = 0, y = 0, z = 0, w = 1; };
struct TransformStamped { struct { V3 translation; Q4 rotation; } transform; };
struct Pose { V3 position; Q4 orientation; };
struct PoseStamped { Pose pose; };
struct Path { std::vector<PoseStamped> poses; };
struct PoseArray { typedef std::vector<Pose> _poses_type; _poses_type poses; };
struct Odometry { struct { struct { V3 linear, angular; } twist; } twist; };
struct alignas(16) PointXYZI { float x = 0, y = 0, z = 0, pad = 1; float intensity = 0; float pad2[3] = {0, 0, 0}; };
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is a 32-byte record");
struct alignas(16) PointXYZ { float x = 0, y = 0, z = 0, pad = 1; };
template <class P> struct Cloud { typedef P PointType; std::vector<P> points; void push_back(const P& p) { points.push_back(p); } };
struct Trajectory { double xv_ = 0, yv_ = 0, thetav_ = 0, cost_ = -1; };

// ---- fake C-ABI ----
struct dddmr_rollout_ctx { int dummy; };
static struct Fake {
  int rc_set_cloud = DDDMR_OK, rc_plan = DDDMR_OK, rc_tick = DDDMR_OK, rc_scan = DDDMR_OK, rc_mark = DDDMR_OK;
  int n_set_cloud = 0, n_plan = 0, n_tick = 0, n_scan = 0, n_update = 0, stitcher = -1;
  size_t cloud_n = 0, cloud_stride = 0, plan_n = 0, scan_n = 0, scan_stride = 0;
  int best_index = 3;
  double plan0[7], pose[7], twist[3], allowed = 0, heading = 0;
  std::string theory, err = "fake error";
  std::vector<float> pb;
  double check_radius = 0;
} F;
extern "C" {
int dddmr_rollout_set_cloud(dddmr_rollout_ctx*, const float* p, size_t n, size_t stride) {
  ++F.n_set_cloud; F.cloud_n = n; F.cloud_stride = stride; (void)p; return F.rc_set_cloud; }
int dddmr_rollout_set_prune_plan(dddmr_rollout_ctx*, const double* p, size_t n) {
  ++F.n_plan; F.plan_n = n; if (n) std::memcpy(F.plan0, p, sizeof(F.plan0)); return F.rc_plan; }
int dddmr_rollout_tick(dddmr_rollout_ctx*, const char* name, const dddmr_tick_input* in, dddmr_rollout_result* out) {
  ++F.n_tick; F.theory = name; std::memcpy(F.pose, in->robot_pose, sizeof(F.pose)); std::memcpy(F.twist, in->robot_twist, sizeof(F.twist));
  F.allowed = in->allowed_max_linear_speed; F.heading = in->heading_deviation;
  std::memset(out, 0, sizeof(*out)); out->best_index = -1;               // (the library presets -1 on every path)
  if (F.rc_tick != DDDMR_OK) return F.rc_tick;
  out->best_index = F.best_index; out->planner_state = F.best_index >= 0 ? DDDMR_TRAJECTORY_FOUND : DDDMR_ALL_TRAJECTORIES_FAIL;
  out->best_cost = F.best_index >= 0 ? 1.25 : -1.0; out->vx = 0.4; out->vy = -0.1; out->wz = 0.2; return DDDMR_OK; }
const char* dddmr_rollout_last_error(dddmr_rollout_ctx*) { return F.err.c_str(); }
int dddmr_rollout_get_best_poses(dddmr_rollout_ctx*, double* out, size_t cap, size_t* n) {
  *n = 2; if (out) { assert(cap >= 2); for (int i = 0; i < 14; ++i) out[i] = i; } return DDDMR_OK; }
int dddmr_rollout_set_stitcher(dddmr_rollout_ctx*, int32_t n) { F.stitcher = n; return DDDMR_OK; }
int dddmr_rollout_set_scan(dddmr_rollout_ctx*, const float*, size_t n, size_t stride, const double b2s[7], const double g2b[7], double, double, uint32_t* n_out) {
  ++F.n_scan; F.scan_n = n; F.scan_stride = stride; assert(b2s[6] == 1.0 && g2b[0] == 2.0); if (n_out) *n_out = 7; return F.rc_scan; }
int dddmr_rollout_set_stitcher_source(dddmr_rollout_ctx*, int32_t src, int32_t n) { F.stitcher = 100 * src + n; return src >= 0 && src < DDDMR_MAX_SOURCES ? DDDMR_OK : DDDMR_ERR_BAD_ARG; }
int dddmr_rollout_set_scan_source(dddmr_rollout_ctx*, int32_t src, const float*, size_t n, size_t stride, const double b2s[7], const double g2b[7], double, double,
                                  uint32_t* n_src, uint32_t* n_all) {
  ++F.n_scan; F.scan_n = n; F.scan_stride = stride; assert(src == 1 && b2s[6] == 1.0 && g2b[0] == 2.0); if (n_src) *n_src = 7; if (n_all) *n_all = 19; return F.rc_scan; }
int dddmr_rollout_path_blocked(dddmr_rollout_ctx*, const float* p, size_t n, double r, double* ratio, int32_t* opinion, uint8_t*) {
  F.pb.assign(p, p + 4 * n); F.check_radius = r; *ratio = 25.0; *opinion = DDDMR_OPINION_PATH_BLOCKED_WAIT; return DDDMR_OK; }
int dddmr_rollout_marking_create(dddmr_rollout_ctx*, const dddmr_marking_config*, const float*, size_t, size_t gs, const float*, size_t, size_t) {
  assert(gs == 32); return DDDMR_OK; }
int dddmr_rollout_marking_update(dddmr_rollout_ctx*, const double*, const double*, dddmr_marking_stats* st) { ++F.n_update; if (st) st->n_alive = 5; return F.rc_mark; }
int dddmr_rollout_marking_reset(dddmr_rollout_ctx*) { return DDDMR_OK; }
int dddmr_rollout_marking_get_dgraph(dddmr_rollout_ctx*, double* v, size_t cap) { for (size_t i = 0; i < cap; ++i) v[i] = 0.5 * i; return DDDMR_OK; }
int dddmr_rollout_marking_get_lethal(dddmr_rollout_ctx*, uint8_t* f, size_t cap) { for (size_t i = 0; i < cap; ++i) f[i] = i % 2; return DDDMR_OK; }
int dddmr_rollout_marking_get_points(dddmr_rollout_ctx*, float* xyz, int32_t* vox, size_t cap, size_t* n) {
  *n = 3; if (xyz) { assert(cap >= 3 && vox == nullptr); for (int i = 0; i < 9; ++i) xyz[i] = 0.25f * i; } return DDDMR_OK; }
void dddmr_rollout_destroy(dddmr_rollout_ctx*) {}
}

using namespace dddmr_rollout_adapter;

int main() {
  dddmr_rollout_ctx ctx{0};
  Cloud<PointXYZI> obs;
  obs.points.resize(10);
  Path plan;
  plan.poses.resize(3);
  plan.poses[0].pose.position.x = 7.0; plan.poses[0].pose.orientation.w = 0.5;
  TransformStamped g2b;
  g2b.transform.translation.x = 2.0;
  Odometry odom;
  odom.twist.twist.linear.x = 0.3; odom.twist.twist.angular.z = -0.1;
  Trajectory best;
  dddmr_rollout_result res;
  std::string err;

  // 1. a tick that finds a trajectory: every input reaches the library, the command comes back
  auto o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "differential_drive_simple", best, &res, &err);
  assert(o == TickOutcome::kTrajectoryFound && F.n_set_cloud == 1 && F.n_plan == 1 && F.n_tick == 1);
  assert(F.cloud_n == 10 && F.cloud_stride == 32 && F.plan_n == 3 && F.plan0[0] == 7.0 && F.plan0[6] == 0.5);
  assert(F.pose[0] == 2.0 && F.pose[6] == 1.0 && F.twist[0] == 0.3 && F.twist[2] == -0.1 && F.allowed == 0.8 && F.heading == 0.25);
  assert(F.theory == "differential_drive_simple" && best.xv_ == 0.4 && best.yv_ == -0.1 && best.thetav_ == 0.2 && best.cost_ == 1.25 && err.empty());

  // 2. the observation is rejected (over max_points): the tick must NOT run, nothing stale is planned against
  F.rc_set_cloud = DDDMR_ERR_CAPACITY; F.err = "set_cloud: 700000 points > max_points 600000";
  best.xv_ = 9.0;
  o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(o == TickOutcome::kPerceptionMalfunction && F.n_tick == 1 && F.n_plan == 1 && best.cost_ == -1 && best.xv_ == 0.0);
  assert(err.find("max_points") != std::string::npos && res.best_index == -1);
  F.rc_set_cloud = DDDMR_OK;

  // 3. the tick itself fails (the library presets best_index = -1): reported as an engine error, not as "all rejected"
  F.rc_tick = DDDMR_ERR_HIP; F.err = "hipErrorLaunchFailure";
  o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(o == TickOutcome::kEngineError && err == "hipErrorLaunchFailure" && best.cost_ == -1);
  F.rc_tick = DDDMR_OK;
  F.rc_plan = DDDMR_ERR_CAPACITY;
  const int ticks_before = F.n_tick;
  o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(o == TickOutcome::kEngineError && F.n_tick == ticks_before);
  F.rc_plan = DDDMR_OK;

  // 4. every sample rejected by the critics
  F.best_index = -1;
  o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(o == TickOutcome::kAllTrajectoriesFail && best.cost_ == -1 && best.xv_ == 0.0 && err.empty());
  F.best_index = 3;

  // 5. the lidar plugin fed this cycle's scan: the planner must not overwrite the device's aggregate -- once
  Cloud<PointXYZ> scan;
  scan.points.resize(100);
  TransformStamped b2s;
  uint32_t n_out = 0;
  assert(feedScan(&ctx, scan, b2s, g2b, 5.0, 2.0, 2, &n_out) == DDDMR_OK && F.n_scan == 1 && F.scan_n == 100 && F.scan_stride == 16 && F.stitcher == 2 && n_out == 7);
  int before = F.n_set_cloud;
  o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(o == TickOutcome::kTrajectoryFound && F.n_set_cloud == before);
  o = rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(F.n_set_cloud == before + 1);
  F.rc_scan = DDDMR_ERR_CAPACITY;                                    // a failed feed does not claim the cycle
  assert(feedScan(&ctx, scan, b2s, g2b, 5.0, 2.0, 0) == DDDMR_ERR_CAPACITY && F.stitcher == 0);
  {                                                                      // a second sensor plugin: source 1
    const int rc_was = F.rc_scan, scans_was = F.n_scan;
    F.rc_scan = DDDMR_OK;
    uint32_t n_src = 0, n_all = 0;
    assert(feedScanSource(&ctx, 1, scan, b2s, g2b, 5.0, 2.0, 3, &n_src, &n_all) == DDDMR_OK);
    assert(F.n_scan == scans_was + 1 && F.stitcher == 103 && n_src == 7 && n_all == 19);
    assert(feedScanSource(&ctx, DDDMR_MAX_SOURCES, scan, b2s, g2b, 5.0, 2.0, 0) == DDDMR_ERR_BAD_ARG && F.n_scan == scans_was + 1);
    (void)SharedContext::consumeDeviceFeed();
    F.rc_scan = rc_was;
  }
  before = F.n_set_cloud;
  (void)rolloutTick(&ctx, obs, plan, g2b, odom, 0.8, 0.25, "t", best, &res, &err);
  assert(F.n_set_cloud == before + 1);
  F.rc_scan = DDDMR_OK;

  // 6. best poses, PathBlocked repack (32-byte records -> x y z intensity)
  PoseArray pa;
  assert(bestPoses(&ctx, pa) == DDDMR_OK && pa.poses.size() == 2 && pa.poses[1].position.x == 7.0 && pa.poses[1].orientation.w == 13.0);
  Cloud<PointXYZI> pp;
  pp.points.resize(2);
  pp.points[1].x = 1.f; pp.points[1].y = 2.f; pp.points[1].z = 3.f; pp.points[1].intensity = -1.f;
  double ratio = 0;
  bool wait = false;
  assert(pathBlocked(&ctx, pp, 0.4, &ratio, &wait) == DDDMR_OK && ratio == 25.0 && wait && F.check_radius == 0.4);
  assert(F.pb.size() == 8 && F.pb[4] == 1.f && F.pb[6] == 3.f && F.pb[7] == -1.f);

  // 7. marking layer: host copies after an update, lethal cloud from the ground nodes
  MarkingLayerBridge ml;
  assert(!ml.ready());
  Cloud<PointXYZI> ground, map, lethal;
  ground.points.resize(6);
  for (int i = 0; i < 6; ++i) ground.points[i].x = (float)i;
  const dddmr_marking_config mc = markingConfig(0.05, 0.05, 2.0, 5.0, 15, -15, 30, 180, -30, -180, 0.1, 1, 1.1, 0.5, 1.5, 9999.0, 6);
  assert(mc.max_markings >= (1u << 15) && mc.inflation_radius == 1.5 && mc.euclidean_cluster_extraction_min_cluster_size == 1);
  assert(ml.create(&ctx, mc, ground, 6, map) == DDDMR_OK && ml.ready() && ml.dGraphValue(3) == 9999.0);
  dddmr_marking_stats st;
  assert(ml.clearThenMark(b2s, g2b, &st) == DDDMR_OK && F.n_update == 1 && st.n_alive == 5 && ml.dGraphValue(4) == 2.0 && ml.dGraphValue(99) == 9999.0);
  ml.lethalPointCloud(ground, lethal);
  assert(lethal.points.size() == 3 && lethal.points[0].x == 1.f && lethal.points[2].x == 5.f);
  Cloud<PointXYZI> gbl_marking;
  assert(ml.markingPointCloud(gbl_marking) == DDDMR_OK && gbl_marking.points.size() == 3 && gbl_marking.points[1].x == 0.75f &&
         gbl_marking.points[2].z == 2.0f);
  F.rc_mark = DDDMR_ERR_CAPACITY;
  assert(ml.clearThenMark(b2s, g2b) == DDDMR_ERR_CAPACITY);

  // 8. shared context: publish / withdraw, acquire / release in a process without a planner
  assert(SharedContext::get() == nullptr);
  SharedContext::publish(&ctx);
  assert(SharedContext::get() == &ctx);
  SharedContext::publish(nullptr);
  int made = 0;
  dddmr_rollout_ctx own{1};
  assert(SharedContext::acquire([&]() { ++made; return &own; }) == &own && SharedContext::acquire([&]() { ++made; return &own; }) == &own && made == 1);
  SharedContext::release();
  assert(SharedContext::get() == &own);
  SharedContext::release();
  assert(SharedContext::get() == nullptr);
  std::puts("adapter bridges OK");
  return 0;
}
