// rollout_engine.hip -- context, host-side theory initialisation and the C-ABI
// (include/dddmr_rollout.h) of the MI355X local-planner rollout engine.
//
// Host responsibilities (cheap, per tick): the dynamic-window sample axes of the
// theory's initialise() (dd_simple_trajectory_generator_theory.cpp:236-295,
// omni_simple_...cpp:260-332, dd_rotate_inplace_theory.cpp:229-274), the local
// costmap tile's extent, and launching the five kernels of rollout_kernels.hip.h.
// Everything proportional to N_traj x N_steps or to the cloud runs on the GPU.
//
// Citations are relative to /root/reference/src/dddmr_local_planner/.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library is dlopen()ed by dddmr_rollout_comm_init

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rollout_kernels.hip.h"
#include "perception_kernels.hip.h"
#include "measure_kernels.hip.h"

using namespace dddmr;

namespace {

constexpr uint32_t kCapCells = 1u << 20;
constexpr int kMaxAxis = 4096;
constexpr int kScoreLdsMax = 150 * 1024;   // dynamic LDS one k_score workgroup may use
constexpr int kCloudBufs = 3;              // front / busy / free, see dddmr_rollout_ctx

struct Window {              // result of a theory's initialise()
  std::vector<float> ax, ay, ath;
  bool list_mode = false;
  std::vector<float4> list;  // explicit samples (rotate-in-place, motor-constraint filter)
  size_t count() const { return list_mode ? list.size() : ax.size() * ay.size() * ath.size(); }
};

// velocity_iterator.h:44-69 -- even samples in [lo,hi], max(2,n) of them, an
// extra 0.0 where the range straddles zero, last sample forced to hi.
void velocity_samples(double lo, double hi, int n, bool insert_zero, std::vector<float>& out) {
  out.clear();
  if (lo == hi) {
    out.push_back((float)lo);
    return;
  }
  n = std::max(2, n);
  const double step = (hi - lo) / double(std::max(1, n - 1));
  double next = lo;
  for (int j = 0; j < n - 1; ++j) {
    const double cur = next;
    next += step;
    out.push_back((float)cur);
    if (insert_zero && cur < 0 && next > 0) out.push_back(0.0f);
  }
  out.push_back((float)hi);
}

bool motor_rpm_ok(const dddmr_theory_config& c, float v, float w) {
  // dd_simple...cpp:297-312, dd_rotate_inplace_theory.cpp:276-286
  const double vr = v + c.robot_radius * w;
  const double vl = v - c.robot_radius * w;
  const double rpm_r = vr * c.gear_ratio * 60. / 3.1415926 / c.wheel_diameter;
  const double rpm_l = vl * c.gear_ratio * 60. / 3.1415926 / c.wheel_diameter;
  return !(std::fabs(rpm_r) >= c.max_motor_shaft_rpm || std::fabs(rpm_l) >= c.max_motor_shaft_rpm);
}

// The dynamic window is computed in float (Eigen::Vector3f max_vel/min_vel) from
// double limits, exactly like the theories' initialise().
void make_window(const dddmr_theory_config& c, const dddmr_tick_input& in, Window& w) {
  w = Window();
  if (!(c.linear_x_sample * c.angular_z_sample > 0)) {
    w.list_mode = true;  // no samples at all
    return;
  }
  const bool zero = c.bench_no_zero_insert == 0;
  const double period = 1.0 / c.controller_frequency;
  const double vx = in.robot_twist[0], vy = in.robot_twist[1], wz = in.robot_twist[2];
  const float accx = (float)c.acc_lim_x, accy = (float)c.acc_lim_y, acct = (float)c.acc_lim_theta;
  const double max_th = c.max_vel_theta, min_th = -1.0 * c.max_vel_theta;

  if (c.kind == DDDMR_THEORY_DD_ROTATE_INPLACE) {
    w.list_mode = true;
    const float sp = (float)c.rotation_speed, sn = (float)(-1.0 * c.rotation_speed);
    if (motor_rpm_ok(c, 0.f, sp)) w.list.push_back(make_float4(0.f, 0.f, sp, 0.f));
    if (motor_rpm_ok(c, 0.f, sn)) w.list.push_back(make_float4(0.f, 0.f, sn, 0.f));
    return;
  }

  float hi_x, lo_x, hi_t, lo_t;
  hi_t = (float)std::min(max_th, wz + acct * period);
  lo_t = (float)std::max(min_th, wz - acct * period);
  if (c.kind == DDDMR_THEORY_DD_SIMPLE) {
    double cap_x = c.max_vel_x;
    if (in.allowed_max_linear_speed > 0.0) cap_x = std::min(cap_x, in.allowed_max_linear_speed);
    hi_x = (float)std::min(cap_x, vx + accx * period);
    lo_x = (float)std::max(c.min_vel_x, vx / c.deceleration_ratio);
    if (hi_x < lo_x) {  // speed zone tighter than the robot can decelerate (:273-276)
      lo_x = (float)(vx / c.deceleration_ratio);
      hi_x = (float)(vx / c.deceleration_ratio);
    }
    velocity_samples(lo_x, hi_x, (int)c.linear_x_sample, zero, w.ax);
    velocity_samples(lo_t, hi_t, (int)c.angular_z_sample, zero, w.ath);
    w.ay.assign(1, 0.0f);
    if (c.use_motor_constraint) {  // filtered list keeps the x-major / theta-minor order
      w.list_mode = true;
      for (float x : w.ax)
        for (float t : w.ath)
          if (motor_rpm_ok(c, x, t)) w.list.push_back(make_float4(x, 0.f, t, 0.f));
    }
    return;
  }
  // omni (omni_simple...cpp:283-312)
  float hi_y, lo_y;
  hi_x = (float)std::min(c.max_vel_x, vx + accx * period);
  hi_y = (float)std::min(c.max_vel_y, vy + accy * period);
  lo_x = (float)std::max(c.min_vel_x, vx - accx * period);
  lo_y = (float)std::max(c.min_vel_y, vy - accy * period);
  if (vx >= c.max_vel_x / c.deceleration_ratio) lo_x = (float)std::max(c.min_vel_x, vx / c.deceleration_ratio);
  else if (vx <= c.min_vel_x / c.deceleration_ratio) hi_x = (float)std::min(c.max_vel_x, vx / c.deceleration_ratio);
  if (vy >= c.max_vel_y / c.deceleration_ratio) lo_y = (float)std::max(c.min_vel_y, vy / c.deceleration_ratio);
  else if (vy <= c.min_vel_y / c.deceleration_ratio) hi_y = (float)std::min(c.max_vel_y, vy / c.deceleration_ratio);
  velocity_samples(lo_x, hi_x, (int)c.linear_x_sample, zero, w.ax);
  velocity_samples(lo_y, hi_y, (int)c.linear_y_sample, zero, w.ay);
  velocity_samples(lo_t, hi_t, (int)c.angular_z_sample, zero, w.ath);
}

void quat_to_rot(const double p[7], double R[9]) {
  // Eigen::Quaterniond(w,x,y,z).toRotationMatrix(), as tf2::transformToEigen builds it
  const double x = p[3], y = p[4], z = p[5], w = p[6];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

float absmax(const std::vector<float>& v) {
  float m = 0.f;
  for (float x : v) m = std::max(m, std::fabs(x));
  return m;
}

struct MarkingState;                       // global-mode marking / clearing layer, marking_host.hip.h
void marking_free(MarkingState* m);

}  // namespace

struct dddmr_rollout_ctx {
  dddmr_rollout_config cfg{};
  std::vector<dddmr_theory_config> theories;
  int device = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  // k_rollout state arrays, grown on demand: [n_local][max steps of the tick]
  TrajInfo* traj_info = nullptr;
  double2* st_sc = nullptr;
  float2* st_xy = nullptr;
  size_t st_cap = 0;       // (trajectory, step) pairs the state arrays hold
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evs0 = nullptr, evs1 = nullptr, cloud_ready[kCloudBufs] = {nullptr, nullptr, nullptr};

  // device memory
  float4* cloud_dev[kCloudBufs] = {nullptr, nullptr, nullptr};
  uint32_t cloud_n[kCloudBufs] = {0, 0, 0};
  uint2* pt_slot = nullptr;
  Pt3* sorted = nullptr;
  uint32_t *cell_count = nullptr, *cell_start = nullptr;
  uint32_t* row_tab = nullptr;       // compact row-run index of the tick's grid (k_bin_scatter -> k_score)
  float* axes_dev = nullptr;
  float4* samples_dev = nullptr;
  float4* plan_dev = nullptr;
  double* costs = nullptr;
  int32_t* steps = nullptr;
  float4* samples_out = nullptr;
  int64_t* best_key = nullptr;
  uint32_t* overflow = nullptr;
  DevResult* result_dev = nullptr;   // device alias of result_host (host-mapped)
  uint32_t* tickets = nullptr;       // [0] binning ticket, [1] scoring ticket
  // load feedback (device): per-trajectory load of the last tick, tile assignment of this one
  float4* blocked_plan = nullptr;      // PathBlockedStrategy scratch (lazily allocated)
  uint32_t* blocked_flags = nullptr;
  uint32_t* traj_load = nullptr;
  uint32_t* assign = nullptr;
  int load_theory = -1, load_nlocal = -1;   // what traj_load describes
  float collided_share = 1.0f;              // share of the last tick's trajectories the collision critics rejected
  int probe_mode = -1;                      // DDDMR_PROBE: 1 / 0 force the walk's probe round on / off, -1 by collided_share
  bool no_assign = false;
  bool no_boxfast = false;   // DDDMR_NO_BOXFAST: always take the general vertex transform
  bool no_tab = false;       // DDDMR_NO_TAB: k_score reads the row-run index from L2 instead of staging it in LDS
  // DDDMR_POISON=1 (tests): fill the per-trajectory outputs with NaN / -1 patterns before every
  // tick, so a trajectory the scorer skipped cannot pass for scored with last tick's values
  bool poison = false;
  // DDDMR_HOST_PROF=1: host-side time of the tick's stages, printed at destroy
  bool host_prof = false;
  double prof_ns[4] = {0, 0, 0, 0};
  uint64_t prof_n = 0;
  bool gnz_one = false;
  double* poses_dev = nullptr;
  // perception feed scratch
  PerceptionScratch feed{};
  // several sensors feeding one aggregate (StackedPerception::aggregateObservations, stacked_perception.cpp:128-140):
  // sources 1.. get their own scratch (stitcher state, staging) and every source its latest observation on the device
  static constexpr int kMaxSources = DDDMR_MAX_SOURCES;
  PerceptionScratch* src_feed[kMaxSources] = {};       // [0] unused (source 0 feeds through `feed`)
  float4* src_cloud[kMaxSources] = {};                 // latest observation of a source, allocated on first use
  uint32_t src_n[kMaxSources] = {};
  bool multi_source = false;

  // pinned host memory
  float4* cloud_stage[kCloudBufs] = {nullptr, nullptr, nullptr};   // pinned staging, one per device cloud buffer
  float* small_stage = nullptr;  // axes / sample list / plan / scan upload
  DevResult* result_host = nullptr;

  // prune plan (host copy)
  uint32_t plan_m = 0;
  double plan_last[7] = {0, 0, 0, 0, 0, 0, 1};

  // Cloud triple buffer: `front` is the published observation the next tick reads, `busy` the one
  // a pending tick (or path_blocked) is reading, and there is always a third that is neither, so
  // set_cloud / set_scan never wait for a tick and never overwrite what one reads.
  std::mutex cloud_mu;
  std::mutex producer_mu;   // serialises set_cloud / set_scan callers (one producer at a time)
  int front = 0;
  int busy = -1;
  // buffer i was published but no wait on cloud_ready[i] has been enqueued on `stream` yet
  bool wait_pending[kCloudBufs] = {false, false, false};

  // multi-rank contexts (dddmr_rollout_comm_init): the library's own RCCL communicator; the tick then
  // runs k_score -> ncclAllReduce(min) of the ranks' (cost bits, -index) slots -> k_resolve on `stream`
  ncclComm_t comm = nullptr;
  int comm_ranks = 0;
  bool comm_loopback = false;        // single-device rehearsal of the exchange: the peers' words are set by the host
  int64_t* slots_dev = nullptr;      // [2 * comm_ranks] send: own slot written by k_score, INT64_MAX elsewhere
  int64_t* slots_red = nullptr;      // [2 * comm_ranks] receive
  DevResult* local_result_dev = nullptr;

  MarkingState* marking = nullptr;   // dddmr_rollout_marking_create

  std::mutex tick_mu;
  std::mutex err_mu;        // last_error is written by tick and sensor threads alike
  std::string last_error;

  // last tick (for get_debug / get_best_poses / resolve)
  DevTick last{};
  bool have_last = false;
  Window last_window;
  float cell_size = 0.25f;
  bool cell_forced = false;   // DDDMR_CELL given: no automatic growth on big shards
  int tile_override = 0;
  int rt_override = 0;        // DDDMR_RT: trajectories per rollout workgroup
  int threads_override = 0;   // DDDMR_THREADS: force the 256- or 512-lane k_score
  int n_cu = 256;   // compute units of the device
  bool tail_round = false;      // DDDMR_TAIL_ROUND=1: one last round of short k_score workgroups (measured: C3 +3 us, C4 -7 us; off)
  int timing = 1;   // DDDMR_TIMING: 0 no HIP events, 1 around k_score (score_ms), 2 also around the whole tick
  int timing_every = 1;   // DDDMR_TIMING_EVERY: record the events on every n-th tick only
  int spin = 1;     // DDDMR_SPIN: poll the host-mapped result instead of hipStreamSynchronize
  int final_mode = -1;   // DDDMR_FINAL: 1 always decode in k_finalize, 0 always in k_score's last workgroup, -1 by shard size
  uint32_t seq = 0;
  DevResult last_result{};   // host copy of the last COLLECTED tick's result
  float last_score_ms = 0.f, last_device_ms = 0.f;
  // a tick whose kernels are enqueued but whose result has not been collected yet
  struct Pending {
    bool active = false;
    DevTick k{};
    int s_tick = 0;
    bool timed = false, timed_all = false;
    dddmr_rollout_result head{};   // n_samples / shard fields known at enqueue time
    Window window;                 // becomes last_window when the tick is collected
  } pend;
};

namespace {

void release_cloud(dddmr_rollout_ctx* c);

// RCCL entry points, resolved at run time: the engine has no link-time dependency on librccl (hosts
// that never call dddmr_rollout_comm_init do not need it), and a process that already holds a copy --
// PyTorch ships its own librccl.so -- shares it instead of loading a second one.
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*error_string)(ncclResult_t) = nullptr;
  std::string why;
  bool ok() const { return handle != nullptr; }
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    if (const char* e = std::getenv("DDDMR_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    for (const char* n : names) {
      if (h) break;
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    r.comm_count = reinterpret_cast<decltype(r.comm_count)>(dlsym(h, "ncclCommCount"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(h, "ncclAllReduce"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.comm_count || !r.all_reduce || !r.error_string) {
      r.why = "librccl lacks an expected symbol";
      return;
    }
    r.handle = h;
  });
  return r;
}

int fail(dddmr_rollout_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) {
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    ctx->last_error = buf;
  }
  return code;
}

#define HIPCHK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, DDDMR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                      \
  } while (0)

const dddmr_theory_config* find_theory(const dddmr_rollout_ctx* ctx, const char* name) {
  for (const auto& t : ctx->theories)
    if (std::strncmp(t.name, name, DDDMR_NAME_LEN) == 0) return &t;
  return nullptr;
}

// The points the collision critics look at around one pose, in the body frame: the 8 cuboid vertices (the min-max
// critic tests their bounding box) AND the 8 corners of the region CollisionModel tests, { d : |d . a_i| <= h_i } around
// the mean of the vertices with a_i, h_i from the edges e_i = v_i - v_0 (collision_model.cpp:85-115).  For a cuboid
// that is a body-frame box that region is the cuboid; for any other vertex list the three slabs meet in the DUAL
// parallelepiped, centre +- g_1 +- g_2 +- g_3, g_i = (e_j x e_k) |e_i|^2 / (2 det), which reaches beyond the vertices'
// hull -- a tile / candidate range sized by the vertices alone never looks at the points in between (found by a soak).
// A degenerate vertex list leaves the region unbounded: the corners then go to the 1 m search ball's box.
void collision_extent_points(const dddmr_theory_config& c, double out[16][3]) {
  double ctr[3] = {0, 0, 0}, e[3][3], g[3][3];
  for (int k = 0; k < 8; ++k)
    for (int a = 0; a < 3; ++a) { out[k][a] = c.cuboid[k][a]; ctr[a] += c.cuboid[k][a] / 8.0; }
  for (int i = 0; i < 3; ++i)
    for (int a = 0; a < 3; ++a) e[i][a] = (double)c.cuboid[i + 1][a] - (double)c.cuboid[0][a];
  auto cross = [](const double* u, const double* v, double* w) {
    w[0] = u[1] * v[2] - u[2] * v[1]; w[1] = u[2] * v[0] - u[0] * v[2]; w[2] = u[0] * v[1] - u[1] * v[0];
  };
  double cr[3][3];
  cross(e[1], e[2], cr[0]); cross(e[2], e[0], cr[1]); cross(e[0], e[1], cr[2]);
  const double det = e[0][0] * cr[0][0] + e[0][1] * cr[0][1] + e[0][2] * cr[0][2];
  const bool ok = std::fabs(det) > 1e-12;
  for (int i = 0; i < 3; ++i) {
    const double n2 = e[i][0] * e[i][0] + e[i][1] * e[i][1] + e[i][2] * e[i][2];
    for (int a = 0; a < 3; ++a) g[i][a] = ok ? cr[i][a] * n2 / (2.0 * det) : 0.0;
  }
  for (int corner = 0; corner < 8; ++corner)
    for (int a = 0; a < 3; ++a) {
      double v = ctr[a];
      for (int i = 0; i < 3; ++i) v += ((corner >> i) & 1) ? g[i][a] : -g[i][a];
      if (!ok) v = ((corner >> a) & 1) ? 1.0 : -1.0;
      out[8 + corner][a] = std::max(-3.0, std::min(3.0, v));
    }
}

// Extent of the local costmap tile: every cloud point that can be inside any
// cuboid of any trajectory of this tick.  A pose stays within rho =
// max speed * sim_time of base_link in the body xy-plane, a cuboid vertex
// within rv of its pose (any yaw), so the body-frame box
// [-(rho+rv), rho+rv]^2 x [vz_min, vz_max] bounds all vertices; points further
// than 1 m from every pose are ignored by the critic's radius search anyway
// (collision_model.cpp:122).
void tile_extent(const dddmr_theory_config& c, const Window& w, const double R[9], const double t[3],
                 double sim_time, float rmin[3], float rmax[3]) {
  double rho;
  if (c.kind == DDDMR_THEORY_DD_ROTATE_INPLACE) {
    rho = 0.0;
  } else if (w.list_mode) {
    double m = 0;
    for (const auto& s : w.list) m = std::max(m, std::hypot((double)s.x, (double)s.y));
    rho = m * sim_time;
  } else {
    rho = std::hypot((double)absmax(w.ax), (double)absmax(w.ay)) * sim_time;
  }
  rho = rho * 1.001 + 0.01;  // float state rounding
  double rv = 0, vz0 = 1e30, vz1 = -1e30;
  double ext[16][3];
  collision_extent_points(c, ext);
  for (int k = 0; k < 16; ++k) {
    rv = std::max(rv, std::hypot(ext[k][0], ext[k][1]));
    vz0 = std::min(vz0, ext[k][2]);
    vz1 = std::max(vz1, ext[k][2]);
  }
  const double e = rho + rv;
  const double margin = 0.02;
  for (int i = 0; i < 3; ++i) {
    double lo = 1e30, hi = -1e30;
    for (int corner = 0; corner < 8; ++corner) {
      const double bx = (corner & 1) ? e : -e, by = (corner & 2) ? e : -e, bz = (corner & 4) ? vz1 : vz0;
      const double v = R[3 * i + 0] * bx + R[3 * i + 1] * by + R[3 * i + 2] * bz + t[i];
      lo = std::min(lo, v);
      hi = std::max(hi, v);
    }
    // radius criterion: within 1 m of some pose, poses within rho of base_link
    lo = std::max(lo, t[i] - (rho + 1.0));
    hi = std::min(hi, t[i] + (rho + 1.0));
    rmin[i] = (float)(lo - margin);
    rmax[i] = (float)(hi + margin);
  }
}

}  // namespace

extern "C" {

#ifdef DDDMR_PHASE_STAMPS
// diagnostic build only: copy the per-workgroup phase stamps of the last k_score launch
int dddmr_rollout_diag_stamps(unsigned long long* out, size_t n_words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n_words * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
// ... and of the last k_bin_count launch (binning, assignment and rollout workgroups)
int dddmr_rollout_diag_rstamps(unsigned long long* out, size_t n_words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rstamps), n_words * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

const char* dddmr_rollout_version(void) { return "dddmr-rollout-mi355x 0.1 (gfx950)"; }

size_t dddmr_rollout_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(dddmr_critic_config);
    case 1: return sizeof(dddmr_theory_config);
    case 2: return sizeof(dddmr_rollout_config);
    case 3: return sizeof(dddmr_tick_input);
    case 4: return sizeof(dddmr_rollout_result);
    case 5: return sizeof(dddmr_rollout_debug);
    case 6: return sizeof(dddmr_marking_config);
    case 7: return sizeof(dddmr_marking_stats);
    default: return 0;
  }
}

int64_t dddmr_rollout_pack_key(double cost, uint32_t global_index) { return pack_key(cost, global_index); }
int32_t dddmr_rollout_key_index(int64_t key) { return key_index(key); }

const char* dddmr_rollout_last_error(dddmr_rollout_ctx* ctx) {
  if (!ctx) return "null context";
  // a per-thread copy: another thread may replace the context's string at any time
  static thread_local std::string copy;
  {
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    copy = ctx->last_error;
  }
  return copy.c_str();
}

void dddmr_rollout_destroy(dddmr_rollout_ctx* ctx) {
  if (!ctx) return;
  if (ctx->host_prof && ctx->prof_n)
    std::fprintf(stderr, "[dddmr] host time per tick over %llu ticks: prepare %.2f us, launches %.2f us, wait for result %.2f us\n",
                 (unsigned long long)ctx->prof_n, ctx->prof_ns[0] / ctx->prof_n * 1e-3, ctx->prof_ns[1] / ctx->prof_n * 1e-3,
                 ctx->prof_ns[2] / ctx->prof_n * 1e-3);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  if (ctx->marking) { marking_free(ctx->marking); ctx->marking = nullptr; }
  if (ctx->comm) (void)rccl().comm_destroy(ctx->comm);
  if (ctx->slots_dev) (void)hipFree(ctx->slots_dev);
  if (ctx->slots_red) (void)hipFree(ctx->slots_red);
  if (ctx->local_result_dev) (void)hipFree(ctx->local_result_dev);
  for (int i = 0; i < kCloudBufs; ++i) {
    if (ctx->cloud_dev[i]) (void)hipFree(ctx->cloud_dev[i]);
    if (ctx->cloud_ready[i]) (void)hipEventDestroy(ctx->cloud_ready[i]);
  }
  void* dev[] = {ctx->row_tab, ctx->pt_slot, ctx->sorted, ctx->cell_count, ctx->cell_start, ctx->axes_dev,
                 ctx->samples_dev, ctx->plan_dev, ctx->costs, ctx->steps, ctx->samples_out,
                 ctx->best_key, ctx->overflow, ctx->tickets, ctx->poses_dev, ctx->traj_load, ctx->assign, ctx->blocked_plan, ctx->blocked_flags};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  perception_free(ctx->feed);
  for (int i = 0; i < dddmr_rollout_ctx::kMaxSources; ++i) {
    if (ctx->src_feed[i]) { perception_free(*ctx->src_feed[i]); delete ctx->src_feed[i]; }
    if (ctx->src_cloud[i]) (void)hipFree(ctx->src_cloud[i]);
  }
  for (int i = 0; i < kCloudBufs; ++i)
    if (ctx->cloud_stage[i]) (void)hipHostFree(ctx->cloud_stage[i]);
  if (ctx->small_stage) (void)hipHostFree(ctx->small_stage);
  if (ctx->result_host) (void)hipHostFree(ctx->result_host);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->evs0) (void)hipEventDestroy(ctx->evs0);
  if (ctx->evs1) (void)hipEventDestroy(ctx->evs1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->traj_info) (void)hipFree(ctx->traj_info);
  if (ctx->st_sc) (void)hipFree(ctx->st_sc);
  if (ctx->st_xy) (void)hipFree(ctx->st_xy);
  delete ctx;
}

int dddmr_rollout_create(const dddmr_rollout_config* cfg, dddmr_rollout_ctx** out) {
  if (!cfg || !out) return DDDMR_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->abi_version != DDDMR_ROLLOUT_ABI_VERSION) return DDDMR_ERR_BAD_ARG;
  if (cfg->n_theories <= 0 || !cfg->theories) return DDDMR_ERR_BAD_ARG;
  if (cfg->max_points == 0 || cfg->max_trajectories == 0 || cfg->max_steps == 0) return DDDMR_ERR_BAD_ARG;
  if (cfg->max_trajectories >= (1u << kKeyIndexBits)) return DDDMR_ERR_CAPACITY;
  if (cfg->max_points >= (1u << 20)) return DDDMR_ERR_CAPACITY;   // a row run's length is packed into 20 bits
  if (cfg->max_steps > 4096) return DDDMR_ERR_CAPACITY;          // a pair index is packed into 12 bits
  if (cfg->max_plan_poses > (uint32_t)kMaxPlan) return DDDMR_ERR_CAPACITY;
  for (int i = 0; i < cfg->n_theories; ++i) {
    const auto& t = cfg->theories[i];
    if (t.n_critics < 0 || t.n_critics > DDDMR_MAX_CRITICS) return DDDMR_ERR_BAD_ARG;
    if (t.kind < 0 || t.kind > DDDMR_THEORY_DD_ROTATE_INPLACE) return DDDMR_ERR_BAD_ARG;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return DDDMR_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= n_dev) return DDDMR_ERR_NO_DEVICE;

  auto* ctx = new dddmr_rollout_ctx();
  ctx->cfg = *cfg;
  ctx->theories.assign(cfg->theories, cfg->theories + cfg->n_theories);
  ctx->cfg.theories = ctx->theories.data();
  ctx->device = cfg->device;
  if (const char* e = std::getenv("DDDMR_CELL")) {
    const float v = (float)std::atof(e);
    if (v > 0.01f && v < 10.f) { ctx->cell_size = v; ctx->cell_forced = true; }
  }
  if (const char* e = std::getenv("DDDMR_TILE")) ctx->tile_override = std::atoi(e);
  if (const char* e = std::getenv("DDDMR_RT")) ctx->rt_override = std::atoi(e);
  if (const char* e = std::getenv("DDDMR_THREADS")) ctx->threads_override = std::atoi(e) == 512 ? 512 : 256;
  if (const char* e = std::getenv("DDDMR_TAIL_ROUND")) ctx->tail_round = std::atoi(e) != 0;
  if (const char* e = std::getenv("DDDMR_TIMING")) ctx->timing = std::atoi(e);
  if (const char* e = std::getenv("DDDMR_TIMING_EVERY")) ctx->timing_every = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("DDDMR_SPIN")) ctx->spin = std::atoi(e);
  if (const char* e = std::getenv("DDDMR_FINAL")) ctx->final_mode = std::atoi(e) ? 1 : 0;
  if (const char* e = std::getenv("DDDMR_PROBE")) ctx->probe_mode = std::atoi(e) ? 1 : 0;

  auto init = [&]() -> int {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
      hipDeviceProp_t prop;
      HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
      ctx->n_cu = std::max(1, prop.multiProcessorCount);
    }
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreate(&ctx->ev0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev1));
    HIPCHK(ctx, hipEventCreate(&ctx->evs0));
    HIPCHK(ctx, hipEventCreate(&ctx->evs1));
    const size_t P = cfg->max_points, N = cfg->max_trajectories;
    const size_t plan_cap = std::max<uint32_t>(cfg->max_plan_poses, 1);
    for (int i = 0; i < kCloudBufs; ++i) {
      HIPCHK(ctx, hipMalloc(&ctx->cloud_dev[i], P * sizeof(float4)));
      HIPCHK(ctx, hipEventCreateWithFlags(&ctx->cloud_ready[i], hipEventDisableTiming));
    }
    HIPCHK(ctx, hipMalloc(&ctx->pt_slot, P * sizeof(uint2)));
    HIPCHK(ctx, hipMalloc(&ctx->sorted, (P + kItem) * sizeof(Pt3)));
    HIPCHK(ctx, hipMemset(ctx->sorted, 0, (P + kItem) * sizeof(Pt3)));
    HIPCHK(ctx, hipMalloc(&ctx->cell_count, (kCapCells + 1) * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->cell_start, (kCapCells + 1) * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(ctx->cell_count, 0, (kCapCells + 1) * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->row_tab, (size_t)kTabCap * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->axes_dev, 3 * kMaxAxis * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&ctx->samples_dev, N * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&ctx->plan_dev, plan_cap * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&ctx->costs, N * sizeof(double)));
    HIPCHK(ctx, hipMalloc(&ctx->steps, N * sizeof(int32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->samples_out, N * sizeof(float4)));
    // words 0..2: reduced argmin words; from kSlotBase on: four words per k_score workgroup (single-round shards)
    HIPCHK(ctx, hipMalloc(&ctx->best_key, ((size_t)kSlotBase + (size_t)kSlotWords * ctx->cfg.max_trajectories) * sizeof(int64_t)));
    HIPCHK(ctx, hipMalloc(&ctx->overflow, sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->traj_load, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(ctx->traj_load, 0, N * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->assign, N * sizeof(uint32_t)));
    ctx->no_assign = std::getenv("DDDMR_NO_ASSIGN") != nullptr;
    ctx->no_boxfast = std::getenv("DDDMR_NO_BOXFAST") != nullptr;
    ctx->no_tab = std::getenv("DDDMR_NO_TAB") != nullptr;
    ctx->poison = std::getenv("DDDMR_POISON") != nullptr;
    ctx->host_prof = std::getenv("DDDMR_HOST_PROF") != nullptr;
    ctx->gnz_one = std::getenv("DDDMR_GNZ_ONE") != nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->tickets, 2 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemset(ctx->tickets, 0, 2 * sizeof(uint32_t)));
    HIPCHK(ctx, hipMalloc(&ctx->poses_dev, (size_t)cfg->max_steps * 7 * sizeof(double)));
    for (int i = 0; i < kCloudBufs; ++i) HIPCHK(ctx, hipHostMalloc(&ctx->cloud_stage[i], P * sizeof(float4), hipHostMallocDefault));
    const size_t small = std::max<size_t>({3 * kMaxAxis * sizeof(float), N * sizeof(float4),
                                           plan_cap * sizeof(float4)});
    HIPCHK(ctx, hipHostMalloc(&ctx->small_stage, small, hipHostMallocDefault));
    HIPCHK(ctx, hipHostMalloc(&ctx->result_host, sizeof(DevResult), hipHostMallocMapped));
    HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->result_dev), ctx->result_host, 0));
    const int rc = perception_alloc(ctx->feed, P);
    if (rc != 0) return fail(ctx, DDDMR_ERR_HIP, "perception scratch allocation failed");
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout), hipFuncAttributeMaxDynamicSharedMemorySize, kScoreLdsMax));
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_count), hipFuncAttributeMaxDynamicSharedMemorySize, kScoreLdsMax));
    {
      const void* score_kernels[] = {
          reinterpret_cast<const void*>(k_score<256, false, false>), reinterpret_cast<const void*>(k_score<256, false, true>),
          reinterpret_cast<const void*>(k_score<256, true, false>),  reinterpret_cast<const void*>(k_score<256, true, true>),
          reinterpret_cast<const void*>(k_score<512, false, false>), reinterpret_cast<const void*>(k_score<512, false, true>),
          reinterpret_cast<const void*>(k_score<512, true, false>),  reinterpret_cast<const void*>(k_score<512, true, true>)};
      for (const void* f : score_kernels) HIPCHK(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kScoreLdsMax));
    }
    HIPCHK(ctx, hipDeviceSynchronize());
    return DDDMR_OK;
  };
  const int rc = init();
  if (rc != DDDMR_OK) {
    fprintf(stderr, "dddmr_rollout_create: %s\n", ctx->last_error.c_str());
    dddmr_rollout_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return DDDMR_OK;
}

// Publish a device-side cloud buffer as the new front buffer.
static void publish_cloud(dddmr_rollout_ctx* ctx, int idx, uint32_t n) {
  std::lock_guard<std::mutex> lk(ctx->cloud_mu);
  ctx->cloud_n[idx] = n;
  ctx->front = idx;
  ctx->wait_pending[idx] = true;
}

// Pick the buffer to fill: neither the published front nor the one a pending tick reads.
// Never waits (the round-1 double buffer blocked -- on one thread forever -- when a second
// observation arrived while a tick_begin was pending); producer_mu must be held.
static int acquire_back(dddmr_rollout_ctx* ctx) {
  std::lock_guard<std::mutex> lk(ctx->cloud_mu);
  for (int i = 0; i < kCloudBufs; ++i)
    if (i != ctx->front && i != ctx->busy) return i;
  return -1;   // unreachable: three buffers, two exclusions
}

// Pin the front buffer for a consumer on ctx->stream (tick_mu held).  *need_wait says whether the
// stream still has to wait for the buffer's upload; call cloud_wait_done() once that wait is enqueued.
static int pin_front(dddmr_rollout_ctx* ctx, bool* need_wait) {
  std::lock_guard<std::mutex> lk(ctx->cloud_mu);
  const int idx = ctx->front;
  ctx->busy = idx;
  *need_wait = ctx->wait_pending[idx];
  return idx;
}
static void cloud_wait_done(dddmr_rollout_ctx* ctx, int idx) {
  std::lock_guard<std::mutex> lk(ctx->cloud_mu);
  ctx->wait_pending[idx] = false;   // (idx is pinned, so it cannot have been republished meanwhile)
}

int dddmr_rollout_set_cloud(dddmr_rollout_ctx* ctx, const float* xyzi, size_t n_points,
                            size_t stride_bytes) {
  if (!ctx) return DDDMR_ERR_BAD_ARG;
  if (n_points > 0 && (!xyzi || stride_bytes < 12 || stride_bytes % 4 != 0))
    return fail(ctx, DDDMR_ERR_BAD_ARG, "set_cloud: bad pointer/stride");
  if (n_points > ctx->cfg.max_points)
    return fail(ctx, DDDMR_ERR_CAPACITY, "set_cloud: %zu points > max_points %u", n_points, ctx->cfg.max_points);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> prod(ctx->producer_mu);
  const int back = acquire_back(ctx);
  // The staging buffer of this slot may still feed the copy of an earlier call.
  HIPCHK(ctx, hipEventSynchronize(ctx->cloud_ready[back]));
  // Repack to float4 (PCL PointXYZI is 32 bytes wide) in pinned memory, in a few chunks so
  // that the DMA of one chunk runs while the next is repacked.  The call does not wait for
  // the copies: consumers wait on cloud_ready[back] (the tick's stream does so on the device).
  const size_t sf = stride_bytes / 4;
  const bool has_i = stride_bytes >= 16;
  float4* stage = ctx->cloud_stage[back];
  const size_t kChunks = n_points >= 32768 ? 4 : 1;
  for (size_t c = 0; c < kChunks; ++c) {
    const size_t b = n_points * c / kChunks, e = n_points * (c + 1) / kChunks;
    if (stride_bytes == 16) {
      std::memcpy(stage + b, xyzi + 4 * b, (e - b) * sizeof(float4));
    } else {
      for (size_t i = b; i < e; ++i) {
        const float* p = xyzi + i * sf;
        stage[i] = make_float4(p[0], p[1], p[2], has_i ? p[3] : 0.f);
      }
    }
    if (e > b)
      HIPCHK(ctx, hipMemcpyAsync(ctx->cloud_dev[back] + b, stage + b, (e - b) * sizeof(float4), hipMemcpyHostToDevice,
                                 ctx->copy_stream));
  }
  HIPCHK(ctx, hipEventRecord(ctx->cloud_ready[back], ctx->copy_stream));
  publish_cloud(ctx, back, (uint32_t)n_points);
  return DDDMR_OK;
}

// one sensor's scan through the feed; source < 0: the single-producer form (the result replaces the aggregate)
static int set_scan_impl(dddmr_rollout_ctx* ctx, int source, const float* xyz, size_t n_points, size_t stride_bytes,
                         const double T_base_sensor[7], const double T_gbl_base[7], double perception_window_size,
                         double marking_height, uint32_t* n_out_points, uint32_t* n_aggregate) {
  if (!ctx || !T_base_sensor || !T_gbl_base) return DDDMR_ERR_BAD_ARG;
  if (source >= dddmr_rollout_ctx::kMaxSources) return fail(ctx, DDDMR_ERR_BAD_ARG, "set_scan: source %d (at most %d sensors)", source, dddmr_rollout_ctx::kMaxSources);
  if (n_points > 0 && (!xyz || stride_bytes < 12 || stride_bytes % 4 != 0))
    return fail(ctx, DDDMR_ERR_BAD_ARG, "set_scan: bad pointer/stride");
  if (n_points > ctx->cfg.max_points)
    return fail(ctx, DDDMR_ERR_CAPACITY, "set_scan: %zu points > max_points %u", n_points, ctx->cfg.max_points);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> prod(ctx->producer_mu);
  if (source < 0 && ctx->multi_source) source = 0;              // once several sensors feed, plain set_scan is sensor 0
  PerceptionScratch* scratch = &ctx->feed;
  if (source > 0) {
    if (!ctx->src_feed[source]) {
      auto* ps = new PerceptionScratch();
      if (perception_alloc(*ps, ctx->cfg.max_points) != 0) { perception_free(*ps); delete ps; return fail(ctx, DDDMR_ERR_HIP, "set_scan: scratch of source %d", source); }
      ctx->src_feed[source] = ps;
    }
    scratch = ctx->src_feed[source];
  }
  if (scratch->stitcher_num > 0 && n_points == 0) return fail(ctx, DDDMR_ERR_BAD_ARG, "set_scan: empty scan with the stitcher on");
  if (source >= 0) {
    ctx->multi_source = true;
    if (!ctx->src_cloud[source]) HIPCHK(ctx, hipMalloc(&ctx->src_cloud[source], (size_t)std::max<uint32_t>(ctx->cfg.max_points, 1) * sizeof(float4)));
  }
  const int back = acquire_back(ctx);
  FeedParams fp;
  quat_to_rot(T_base_sensor, fp.Rbs);
  quat_to_rot(T_gbl_base, fp.Rgb);
  for (int i = 0; i < 3; ++i) {
    fp.tbs[i] = T_base_sensor[i];
    fp.tgb[i] = T_gbl_base[i];
  }
  fp.n = (int)n_points;
  fp.window = (float)perception_window_size;
  fp.height = (float)marking_height;
  uint32_t n_out = 0;
  float4* dst = source >= 0 ? ctx->src_cloud[source] : ctx->cloud_dev[back];
  const int rc = perception_feed(*scratch, fp, xyz, stride_bytes, dst, ctx->copy_stream, &n_out);
  if (rc == -2) return fail(ctx, DDDMR_ERR_CAPACITY, "set_scan: the (stitched) scan exceeds max_points %u", ctx->cfg.max_points);
  if (rc != 0) return fail(ctx, DDDMR_ERR_HIP, "set_scan: perception feed failed (%d)", rc);
  uint32_t n_all = n_out;
  if (source >= 0) {
    // the aggregate = every sensor's latest observation, in sensor order (aggregateObservations' loop over the plugins)
    ctx->src_n[source] = n_out;
    size_t total = 0;
    for (int i = 0; i < dddmr_rollout_ctx::kMaxSources; ++i) total += ctx->src_n[i];
    if (total > ctx->cfg.max_points) return fail(ctx, DDDMR_ERR_CAPACITY, "set_scan: the sensors' observations together (%zu points) exceed max_points %u", total, ctx->cfg.max_points);
    size_t at = 0;
    for (int i = 0; i < dddmr_rollout_ctx::kMaxSources; ++i) {
      if (!ctx->src_n[i]) continue;
      HIPCHK(ctx, hipMemcpyAsync(ctx->cloud_dev[back] + at, ctx->src_cloud[i], (size_t)ctx->src_n[i] * sizeof(float4), hipMemcpyDeviceToDevice, ctx->copy_stream));
      at += ctx->src_n[i];
    }
    n_all = (uint32_t)total;
  }
  // the tick's stream waits on this event, so the feed kernels need not have retired yet
  HIPCHK(ctx, hipEventRecord(ctx->cloud_ready[back], ctx->copy_stream));
  publish_cloud(ctx, back, n_all);
  if (n_out_points) *n_out_points = n_out;
  if (n_aggregate) *n_aggregate = n_all;
  return DDDMR_OK;
}

int dddmr_rollout_set_scan(dddmr_rollout_ctx* ctx, const float* xyz, size_t n_points,
                           size_t stride_bytes, const double T_base_sensor[7],
                           const double T_gbl_base[7], double perception_window_size,
                           double marking_height, uint32_t* n_out_points) {
  return set_scan_impl(ctx, -1, xyz, n_points, stride_bytes, T_base_sensor, T_gbl_base, perception_window_size, marking_height, n_out_points, nullptr);
}

int dddmr_rollout_set_scan_source(dddmr_rollout_ctx* ctx, int32_t source_id, const float* xyz, size_t n_points,
                                  size_t stride_bytes, const double T_base_sensor[7], const double T_gbl_base[7],
                                  double perception_window_size, double marking_height, uint32_t* n_source_points,
                                  uint32_t* n_aggregate_points) {
  if (source_id < 0) return ctx ? fail(ctx, DDDMR_ERR_BAD_ARG, "set_scan_source: source %d", source_id) : DDDMR_ERR_BAD_ARG;
  return set_scan_impl(ctx, source_id, xyz, n_points, stride_bytes, T_base_sensor, T_gbl_base, perception_window_size, marking_height,
                       n_source_points, n_aggregate_points);
}

int dddmr_rollout_set_stitcher(dddmr_rollout_ctx* ctx, int32_t stitcher_num) {
  if (!ctx || stitcher_num < 0) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> prod(ctx->producer_mu);
  ctx->feed.stitcher_num = stitcher_num;
  ctx->feed.stitched.clear();
  return DDDMR_OK;
}

int dddmr_rollout_set_stitcher_source(dddmr_rollout_ctx* ctx, int32_t source_id, int32_t stitcher_num) {
  if (!ctx || stitcher_num < 0 || source_id < 0 || source_id >= dddmr_rollout_ctx::kMaxSources) return DDDMR_ERR_BAD_ARG;
  if (source_id == 0) return dddmr_rollout_set_stitcher(ctx, stitcher_num);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> prod(ctx->producer_mu);
  if (!ctx->src_feed[source_id]) {
    auto* ps = new PerceptionScratch();
    if (perception_alloc(*ps, ctx->cfg.max_points) != 0) { perception_free(*ps); delete ps; return fail(ctx, DDDMR_ERR_HIP, "set_stitcher: scratch of source %d", source_id); }
    ctx->src_feed[source_id] = ps;
  }
  ctx->src_feed[source_id]->stitcher_num = stitcher_num;
  ctx->src_feed[source_id]->stitched.clear();
  return DDDMR_OK;
}

int dddmr_rollout_get_cloud(dddmr_rollout_ctx* ctx, float* xyzi_out, size_t capacity, size_t* n_points) {
  if (!ctx || !n_points) return DDDMR_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "get_cloud while a tick_begin is pending");
  bool pending;
  const int idx = pin_front(ctx, &pending);     // a producer must not recycle the buffer while it is copied
  struct Release { dddmr_rollout_ctx* c; ~Release() { release_cloud(c); } } release{ctx};
  const uint32_t n = ctx->cloud_n[idx];
  *n_points = n;
  if (!xyzi_out) return DDDMR_OK;
  if (capacity < n) return fail(ctx, DDDMR_ERR_CAPACITY, "get_cloud: capacity %zu < %u", capacity, n);
  HIPCHK(ctx, hipEventSynchronize(ctx->cloud_ready[idx]));      // set_cloud returns before its copy has landed
  if (n) HIPCHK(ctx, hipMemcpy(xyzi_out, ctx->cloud_dev[idx], (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
  return DDDMR_OK;
}

int dddmr_rollout_path_blocked(dddmr_rollout_ctx* ctx, const float* plan_xyzi, size_t n_plan, double check_radius,
                               double* ratio, int32_t* opinion, uint8_t* blocked_flags) {
  if (!ctx || !ratio || !opinion) return DDDMR_ERR_BAD_ARG;
  if (n_plan > 0 && !plan_xyzi) return fail(ctx, DDDMR_ERR_BAD_ARG, "path_blocked: null plan");
  if (n_plan > (size_t)kBlockedMaxPlan)
    return fail(ctx, DDDMR_ERR_CAPACITY, "path_blocked: %zu plan points > %d", n_plan, kBlockedMaxPlan);
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "path_blocked while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  *ratio = 0.0;
  *opinion = DDDMR_OPINION_PASS;
  if (blocked_flags) std::memset(blocked_flags, 0, n_plan);
  // pin the front cloud like a tick does
  bool pending;
  const int cidx = pin_front(ctx, &pending);
  struct Release { dddmr_rollout_ctx* c; ~Release() { release_cloud(c); } } release{ctx};
  const uint32_t n_points = ctx->cloud_n[cidx];
  if (n_points <= 5 || n_plan == 0) return DDDMR_OK;                  // path_blocked_strategy.cpp:62-64
  BlockedParams b;
  b.n_points = (int)n_points;
  b.m = (int)n_plan;
  b.r2 = static_cast<float>(check_radius * check_radius);             // pcl::KdTreeFLANN::radiusSearch
  bool any = false;
  for (int a = 0; a < 3; ++a) { b.lo[a] = 3.402823466e+38f; b.hi[a] = -3.402823466e+38f; }
  for (size_t i = 0; i < n_plan; ++i) {
    if (plan_xyzi[4 * i + 3] < 0) continue;
    any = true;
    for (int a = 0; a < 3; ++a) {
      b.lo[a] = std::min(b.lo[a], plan_xyzi[4 * i + a]);
      b.hi[a] = std::max(b.hi[a], plan_xyzi[4 * i + a]);
    }
  }
  if (any) {
    // conservative reject box: a point farther than r from the plan's box along an axis cannot be within r
    const float grow = (float)(std::fabs(check_radius) * 1.000001 + 1e-6);
    for (int a = 0; a < 3; ++a) {
      b.lo[a] = std::nextafter(b.lo[a] - grow, -3.402823466e+38f);
      b.hi[a] = std::nextafter(b.hi[a] + grow, 3.402823466e+38f);
    }
    if (!ctx->blocked_plan) {
      HIPCHK(ctx, hipMalloc(&ctx->blocked_plan, kBlockedMaxPlan * sizeof(float4)));
      HIPCHK(ctx, hipMalloc(&ctx->blocked_flags, (kBlockedMaxPlan / 32) * sizeof(uint32_t)));
    }
    if (pending) {
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->cloud_ready[cidx], 0));
      cloud_wait_done(ctx, cidx);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->blocked_plan, plan_xyzi, n_plan * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->blocked_flags, 0, (kBlockedMaxPlan / 32) * sizeof(uint32_t), ctx->stream));
    const int blocks = (int)std::min<uint32_t>(1024, (n_points + 255) / 256);
    hipLaunchKernelGGL(k_path_blocked, dim3(blocks), dim3(256), 0, ctx->stream, b, ctx->cloud_dev[cidx],
                       ctx->blocked_plan, ctx->blocked_flags);
    uint32_t words[kBlockedMaxPlan / 32];
    HIPCHK(ctx, hipMemcpyAsync(words, ctx->blocked_flags, sizeof(words), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    size_t blocked = 0;
    for (size_t i = 0; i < n_plan; ++i) {
      const bool hit = (words[i >> 5] >> (i & 31)) & 1u;
      if (hit) ++blocked;
      if (blocked_flags) blocked_flags[i] = hit ? 1 : 0;
    }
    const float orig = (float)n_plan, blk = (float)blocked;           // float division, double scale (:91-93)
    *ratio = (blk) / (orig) * 100.0;
  }
  if (*ratio > 0.0) *opinion = DDDMR_OPINION_PATH_BLOCKED_WAIT;       // :96-97
  return DDDMR_OK;
}

int dddmr_rollout_set_prune_plan(dddmr_rollout_ctx* ctx, const double* poses, size_t n_poses) {
  if (!ctx) return DDDMR_ERR_BAD_ARG;
  if (n_poses > 0 && !poses) return fail(ctx, DDDMR_ERR_BAD_ARG, "set_prune_plan: null poses");
  if (n_poses > ctx->cfg.max_plan_poses)
    return fail(ctx, DDDMR_ERR_CAPACITY, "set_prune_plan: %zu poses > max_plan_poses %u", n_poses,
                ctx->cfg.max_plan_poses);
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "set_prune_plan while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // ModelSharedData::updateData: positions as float PointXYZI (model_shared_data.h:83-91)
  std::vector<float4> xyz(std::max<size_t>(n_poses, 1));
  for (size_t i = 0; i < n_poses; ++i)
    xyz[i] = make_float4((float)poses[7 * i + 0], (float)poses[7 * i + 1], (float)poses[7 * i + 2], 0.f);
  if (n_poses) {
    HIPCHK(ctx, hipMemcpy(ctx->plan_dev, xyz.data(), n_poses * sizeof(float4), hipMemcpyHostToDevice));
    std::memcpy(ctx->plan_last, poses + 7 * (n_poses - 1), 7 * sizeof(double));
  }
  ctx->plan_m = (uint32_t)n_poses;
  return DDDMR_OK;
}

}  // extern "C"

namespace {

void release_cloud(dddmr_rollout_ctx* c) {
  std::lock_guard<std::mutex> lk(c->cloud_mu);
  c->busy = -1;
}

// Enqueue one tick (host-side initialise() + 3 launches); tick_mu must be held.
int tick_enqueue(dddmr_rollout_ctx* ctx, const char* theory_name, const dddmr_tick_input* in) {
  dddmr_rollout_result head_storage;
  dddmr_rollout_result* out = &head_storage;
  std::memset(out, 0, sizeof(*out));
  out->planner_state = DDDMR_ALL_TRAJECTORIES_FAIL;
  out->best_index = -1;
  out->best_cost = -1.0;
  out->key = kKeyNone;
  const auto prof_t0 = std::chrono::steady_clock::now();
  const dddmr_theory_config* th = find_theory(ctx, theory_name);
  if (!th) return fail(ctx, DDDMR_ERR_UNKNOWN_THEORY, "unknown theory '%s'", theory_name);
  HIPCHK(ctx, hipSetDevice(ctx->device));

  // ---- initialise(): velocity samples of this tick ----
  Window& w = ctx->pend.window;
  make_window(*th, *in, w);
  const size_t N = w.count();
  if (N > ctx->cfg.max_trajectories)
    return fail(ctx, DDDMR_ERR_CAPACITY, "%zu samples > max_trajectories %u", N, ctx->cfg.max_trajectories);
  if (!w.list_mode && (w.ax.size() > (size_t)kMaxAxis || w.ay.size() > (size_t)kMaxAxis || w.ath.size() > (size_t)kMaxAxis))
    return fail(ctx, DDDMR_ERR_CAPACITY, "sample axis longer than %d", kMaxAxis);
  const int world = std::max(1, ctx->cfg.world_size);
  const int rank = std::min(std::max(0, ctx->cfg.rank), world - 1);
  const uint32_t begin = (uint32_t)((uint64_t)rank * N / world);
  const uint32_t end = (uint32_t)((uint64_t)(rank + 1) * N / world);
  out->n_samples = (uint32_t)N;
  out->local_begin = begin;
  out->n_local = end - begin;

  DevTick k{};
  k.kind = th->kind;
  k.fixed_steps = th->bench_fixed_steps > 0 ? th->bench_fixed_steps : 0;
  k.list_mode = w.list_mode ? 1 : 0;
  k.n_global = (int)N;
  k.begin = (int)begin;
  k.n_local = (int)(end - begin);
  k.nx = (int)std::max<size_t>(w.ax.size(), 1);
  k.ny = (int)std::max<size_t>(w.ay.size(), 1);
  k.nth = (int)std::max<size_t>(w.ath.size(), 1);
  k.ay_ofs = kMaxAxis;
  k.ath_ofs = 2 * kMaxAxis;
  k.sim_time = th->sim_time;
  k.sim_gran = th->sim_granularity;
  k.ang_gran = th->angular_sim_granularity;
  k.min_vel_x = th->min_vel_x;
  k.max_vel_x = th->max_vel_x;
  k.min_vel_theta = th->min_vel_theta;
  k.min_vel_trans = th->min_vel_trans;
  k.max_vel_trans = th->max_vel_trans;
  k.allowed_max = in->allowed_max_linear_speed;
  quat_to_rot(in->robot_pose, k.R);
  for (int i = 0; i < 3; ++i) k.t[i] = in->robot_pose[i];
  for (int v = 0; v < 8; ++v)
    for (int j = 0; j < 3; ++j) k.cub[3 * v + j] = th->cuboid[v][j];
  k.m = (int)ctx->plan_m;
  quat_to_rot(ctx->plan_last, k.planR);
  for (int i = 0; i < 3; ++i) k.planT[i] = ctx->plan_last[i];
  k.n_critics = th->n_critics;
  for (int m = 0; m < th->n_critics; ++m) {
    k.ckind[m] = th->critics[m].kind;
    k.cw[m] = th->critics[m].weight;
    k.ctw[m] = th->critics[m].translation_weight;
    k.cow[m] = th->critics[m].orientation_weight;
    if (k.ckind[m] == DDDMR_CRITIC_COLLISION) k.want_collision = 1;
    if (k.ckind[m] == DDDMR_CRITIC_COLLISION_MIN_MAX) k.want_minmax = 1;
  }
  k.heading_dev = in->heading_deviation;

  // horizon of this tick (monotone in |v| and |w|, so the axis extremes bound it)
  double sim_time_eff = th->sim_time;
  int s_tick;
  {
    double vmax, wmax;
    if (w.list_mode) {
      vmax = 0; wmax = 0;
      for (const auto& s : w.list) {
        vmax = std::max(vmax, std::hypot((double)s.x, (double)s.y));
        wmax = std::max(wmax, std::fabs((double)s.z));
      }
    } else {
      vmax = std::hypot((double)absmax(w.ax), (double)absmax(w.ay));
      wmax = (double)absmax(w.ath);
    }
    if (th->bench_fixed_steps > 0) {
      s_tick = th->bench_fixed_steps;
    } else if (th->kind == DDDMR_THEORY_DD_ROTATE_INPLACE) {
      s_tick = (int)std::ceil(std::max(0.0, 6.28 / th->angular_sim_granularity)) + 1;
      sim_time_eff = 0.0;
    } else {
      s_tick = (int)std::ceil(std::max(vmax * th->sim_time / th->sim_granularity,
                                       wmax * th->sim_time / th->angular_sim_granularity)) + 1;
    }
    s_tick = std::max(s_tick, 1);
  }
  if ((uint32_t)s_tick > ctx->cfg.max_steps)
    return fail(ctx, DDDMR_ERR_CAPACITY, "horizon of %d steps > max_steps %u", s_tick, ctx->cfg.max_steps);
  k.max_steps = s_tick;

  // ---- cloud front buffer + local costmap tile ----
  // (the "upload still pending" flag is only cleared once the stream wait below is enqueued: an
  // early error return in between must not lose the ordering against copy_stream)
  bool pending;
  const int cidx = pin_front(ctx, &pending);
  struct Unbusy {          // releases the cloud buffer again if enqueueing fails half-way
    dddmr_rollout_ctx* c;
    bool armed = true;
    ~Unbusy() { if (armed) release_cloud(c); }
  } unbusy{ctx};
  k.n_points = (int)ctx->cloud_n[cidx];
  tile_extent(*th, w, k.R, k.t, sim_time_eff, k.rmin, k.rmax);
  // A cuboid's AABB (clipped to the 2 m wide search ball) must not span more than
  // kRows cell rows: rows <= span / cell + 2.
  double diam = 0;
  {
    double ext[16][3];
    collision_extent_points(*th, ext);
    for (int a = 0; a < 16; ++a)
      for (int b = a + 1; b < 16; ++b) {
        const double dx = ext[a][0] - ext[b][0], dy = ext[a][1] - ext[b][1], dz = ext[a][2] - ext[b][2];
        diam = std::max(diam, std::sqrt(dx * dx + dy * dy + dz * dz));
      }
    diam += 4e-4;     // the candidate range's margin on both sides (k_score phase D1)
  }
  float cell = std::max(ctx->cell_size, (float)(std::min(diam, 2.0) * 1.001 / (kRows - 2)));
  float cell_z = cell;     // z cells do not grow with the x/y cells below
  // Big shards run many 256-lane workgroups per CU and are bound by how many (trajectory,
  // step) slots fit a CU's LDS; a slot's row segments are the largest part of it, so there the
  // cells grow until a cuboid spans at most 4 rows (C3 k_score 143 -> 122 us, C4 342 -> 298 us
  // at 0.42 m).  Shards that fit one round of 512-lane workgroups keep the small cells: their
  // LDS is not the limit and bigger cells make the counting atomics collide (C2 binning
  // +3 us at 0.42 m, +7 us at 0.5 m).
  if (!ctx->cell_forced && k.n_local > ctx->n_cu * kMaxTile)
    cell = std::max(cell, std::min(0.5f, (float)(std::min(diam, 2.0) * 1.001 / 2.9)));
  for (;;) {
    k.gnx = std::max(1, (int)std::ceil((k.rmax[0] - k.rmin[0]) / cell));
    k.gny = std::max(1, (int)std::ceil((k.rmax[1] - k.rmin[1]) / cell));
    // Candidate runs always take every z of a row, but one cell column per (x, y) makes the
    // counting atomics of wall points collide (measured: k_bin_count 13 -> 17 us); keep z.
    k.gnz = ctx->gnz_one ? 1 : std::max(1, (int)std::ceil((k.rmax[2] - k.rmin[2]) / cell_z));
    const uint64_t nc = (uint64_t)k.gnx * k.gny * k.gnz;
    if (nc <= kCapCells && k.gnx < 32000 && k.gny < 32000) { k.n_cells = (int)nc; break; }
    cell *= 1.5f;
    cell_z *= 1.5f;
  }
  k.inv_cell = 1.0f / cell;
  k.inv_cell_z = 1.0f / cell_z;
  for (int i = 0; i < 3; ++i) k.gmin[i] = k.rmin[i];
  // rows <= floor(span / cell) + 2 (span = cuboid diameter clipped to the 2 m search ball)
  k.rows_cap = std::min(kRows, (int)std::floor(std::min(diam, 2.0) * 1.001 / cell) + 2);
  {
    // box in the body frame, vertices in the push order blb brb blt flb brt frt flt frb
    // (dd_simple_trajectory_generator_theory.cpp:211-218)?  Then k_score shares the products.
    const float (*c)[3] = th->cuboid;
    const float X0 = c[0][0], X1 = c[3][0], Y0 = c[0][1], Y1 = c[1][1], Z0 = c[0][2], Z1 = c[2][2];
    const float want[8][3] = {{X0, Y0, Z0}, {X0, Y1, Z0}, {X0, Y0, Z1}, {X1, Y0, Z0},
                              {X0, Y1, Z1}, {X1, Y1, Z1}, {X1, Y0, Z1}, {X1, Y1, Z0}};
    bool box = true;
    for (int v = 0; v < 8; ++v)
      for (int a = 0; a < 3; ++a) box = box && (c[v][a] == want[v][a]);
    k.box_fast = (box && !ctx->no_boxfast) ? 1 : 0;
  }

  // trajectories per workgroup: ~one (trajectory, step) pair per lane
  // OBB records carry the pose only if some pair can need the 1 m radius test: a point
  // inside the box is within max|vertex| of the pose, so a cuboid that lies inside the
  // search ball never does (the min-max critic always needs it).
  double vnorm = 0;
  {
    double ext[16][3];       // (the corners of the box the collision critic derives count too)
    collision_extent_points(*th, ext);
    for (int v = 0; v < 16; ++v) vnorm = std::max(vnorm, std::sqrt(ext[v][0] * ext[v][0] + ext[v][1] * ext[v][1] + ext[v][2] * ext[v][2]));
  }
  k.rec_pose = (vnorm >= 0.985 || k.want_minmax) ? 1 : 0;
  const int rec_words = rec_words_of(k.rec_pose != 0, k.want_minmax != 0);
  {
    const long te = (long)(k.gnx + 1) * k.gny;
    k.tab_entries = (te <= kTabCap && k.n_points >= 5 && (k.want_collision || k.want_minmax) && !ctx->no_tab) ? (int)te : 0;
  }
  // Workgroup shape.  Default: 256 lanes and the tile that keeps the most (trajectory, step)
  // slots resident per CU with the slots of one workgroup fitting its lanes (one pair per
  // lane in D1/D2) -- measured on the big batches: C4 (50 steps) tile 3 / 4 / 5 / 6
  // -> 420 / 342 / 364 / 519 us (39 KB of LDS at tile 4: four workgroups per CU, 47 KB at
  // tile 5: three), C3 (80-step rows) tile 2 / 3 / 4 -> 158 / 143 / 174 us.
  // When the whole shard fits ONE round of resident 512-lane workgroups (2 per CU at 4 waves
  // per SIMD and <= 80 KB of LDS), that shape wins instead: the launch is bound by its
  // heaviest tile's collision walk and 512 lanes both halve it and average over more
  // trajectories (C2).
  // Workgroup shape.  Per-workgroup fixed costs (staging the plan and the row-run index, ~13 barriers, the wave-0
  // scans) make few, fat workgroups win: measured on the r02 scenes, k_score at C3 (80-step rows) 256 lanes x tile
  // 2 / 3 / 4 -> 222 / 158 / 152 us, 512 lanes x tile 6 -> 132 us; C4 (50-step rows) 256 lanes x tile 3 / 5 / 7 ->
  // 564 / 344 / 336 us, 512 lanes x tile 8 / 10 / 11 -> 347 / 304 / 320 us.  So: 512 lanes (two workgroups per CU
  // at 4 waves per SIMD and <= 80 KB of LDS each) and
  //  - a shard that fits ONE round of resident workgroups is spread evenly over them (C2: tile 8, 512 workgroups);
  //  - a bigger shard takes the largest tile whose (trajectory, step) pairs still fit the lanes (one pair per lane
  //    in D1 / D2) and whose LDS fits twice into a CU.
  // (1024-lane workgroups, one per CU, lose again: C3 143 us at tile 12, C4 390 us at tile 16.)
  // DDDMR_THREADS=256 / DDDMR_TILE keep the 256-lane shape reachable for experiments.
  int thr = 512;
  auto lds_of = [&](int t) { return score_lds_bytes(t, s_tick, k.m, rec_words, k.tab_entries, k.rows_cap); };
  auto tile_for_256 = [&]() {
    // most (trajectory, step) slots resident per CU: workgroups per CU (by registers, fewer by LDS) x slots per
    // workgroup, slots <= lanes
    int t_best = 1;
    long best = 0;
    for (int t = 1; t <= kMaxTile; ++t) {
      if (t > 1 && t * s_tick > 256) break;
      const size_t need = lds_of(t) + 1024;   // + static LDS
      const long wgs = std::min<long>(DDDMR_SCORE_WPE_256, (long)((size_t)(160 * 1024) / need));
      const long resident = wgs * t * s_tick;
      if (resident >= best) { best = resident; t_best = t; }
    }
    return t_best;
  };
  int tile = 1;
  if (ctx->tile_override > 0) {
    tile = std::min(ctx->tile_override, kMaxTile);
    thr = ctx->threads_override > 0 ? ctx->threads_override : 256;
  } else if (ctx->threads_override == 256) {
    thr = 256;
    tile = tile_for_256();
  } else if (k.n_local > 0) {
    const int slots512 = ctx->n_cu * 2;
    const int fit = (k.n_local + slots512 - 1) / slots512;
    if (fit <= kMaxTile && fit * s_tick <= 2 * 512 && lds_of(fit) <= (size_t)80 * 1024) {
      tile = std::max(fit, 1);
    } else {
      for (int t = 2; t <= kMaxTile; ++t) {
        if (t * s_tick > 512 || lds_of(t) > (size_t)80 * 1024) break;
        tile = t;
      }
    }
  }
  while (tile > 1 && score_lds_bytes(tile, s_tick, k.m, rec_words, k.tab_entries, k.rows_cap) > (size_t)(160 * 1024) / 2) --tile;
  const size_t lds = score_lds_bytes(tile, s_tick, k.m, rec_words, k.tab_entries, k.rows_cap);
  if (lds > (size_t)kScoreLdsMax) return fail(ctx, DDDMR_ERR_CAPACITY, "horizon needs %zu bytes of LDS", lds);
  if (ctx->seq == 3 && std::getenv("DDDMR_DEBUG_GRID"))
    std::fprintf(stderr, "[dddmr] k_score shape: %d lanes, tile %d, %d-step rows, %zu bytes of dynamic LDS (tile+1 would need %zu)\n", thr, tile, s_tick, lds,
                 score_lds_bytes(tile + 1, s_tick, k.m, rec_words, k.tab_entries, k.rows_cap));
  k.tile = tile;
  // Who decodes the winner: shards that run as ONE round of workgroups let the last workgroup do it (a
  // finalize launch would cost the tick ~3 us); bigger shards run several rounds, where every workgroup's ticket
  // round trip holds a slot that the next workgroup is waiting for -- there a one-wave k_finalize follows.
  // The collision walk's probe round (every lane first walks ONE item, spread evenly over the tile's list) settles
  // colliding trajectories early; when few collide it is a barrier and a scan for nothing.  Measured: 86 %
  // colliding (C3, r01 scene) k_score 114 us with / 155 us without; 25 % colliding (r02 scenes) C3 126.5 / 124.0 us,
  // C4 295.6 / 285.2 us.  Decided by the share the previous tick of the same theory and shard measured; either way
  // gives identical results.
  const bool same_as_last = ctx->load_theory == (int)(th - ctx->theories.data()) && ctx->load_nlocal == k.n_local;
  k.probe = ctx->probe_mode >= 0 ? ctx->probe_mode : ((!same_as_last || ctx->collided_share > 0.5f) ? 1 : 0);
  const bool one_round = k.n_local <= 0 || (k.n_local + tile - 1) / tile <= ctx->n_cu * (thr == 512 ? 2 : 4);
  k.final_kernel = ctx->final_mode >= 0 ? ctx->final_mode : (one_round ? 0 : 1);

  // small per-tick uploads (sample axes or explicit list).  A rank with an EMPTY shard needs them too when the context
  // has a communicator: k_resolve decodes the global winner's command from them on every rank (rotate-in-place has two
  // samples, so rank 0 of three or more ranks owns none).
  if (k.n_local > 0 || ctx->comm || ctx->comm_loopback) {
    if (w.list_mode) {
      std::memcpy(ctx->small_stage, w.list.data(), N * sizeof(float4));
      HIPCHK(ctx, hipMemcpyAsync(ctx->samples_dev, ctx->small_stage, N * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
    } else if (w.ax.size() + w.ay.size() + w.ath.size() <= (size_t)kInlineAxes) {
      k.axes_inline = 1;                      // axes ride in the kernel arguments
      k.ay_ofs = (int)w.ax.size();
      k.ath_ofs = (int)(w.ax.size() + w.ay.size());
      std::memcpy(k.axes_inl, w.ax.data(), w.ax.size() * sizeof(float));
      std::memcpy(k.axes_inl + k.ay_ofs, w.ay.data(), w.ay.size() * sizeof(float));
      std::memcpy(k.axes_inl + k.ath_ofs, w.ath.data(), w.ath.size() * sizeof(float));
    } else {
      float* a = ctx->small_stage;
      std::memcpy(a, w.ax.data(), w.ax.size() * sizeof(float));
      std::memcpy(a + k.ay_ofs, w.ay.data(), w.ay.size() * sizeof(float));
      std::memcpy(a + k.ath_ofs, w.ath.data(), w.ath.size() * sizeof(float));
      HIPCHK(ctx, hipMemcpyAsync(ctx->axes_dev, a, 3 * kMaxAxis * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
  }
  if (k.n_local > 0) {
    // state arrays of the body-frame rollout
    const size_t need = (size_t)k.n_local * (size_t)s_tick;
    if (need > ctx->st_cap || !ctx->traj_info) {
      HIPCHK(ctx, hipDeviceSynchronize());
      if (ctx->st_sc) (void)hipFree(ctx->st_sc);
      if (ctx->st_xy) (void)hipFree(ctx->st_xy);
          ctx->st_sc = nullptr; ctx->st_xy = nullptr; ctx->st_cap = 0;
      const size_t cap = need + need / 4 + 1024;
      HIPCHK(ctx, hipMalloc(&ctx->st_sc, cap * sizeof(double2)));
      HIPCHK(ctx, hipMalloc(&ctx->st_xy, cap * sizeof(float2)));
      ctx->st_cap = cap;
      if (!ctx->traj_info) HIPCHK(ctx, hipMalloc(&ctx->traj_info, (size_t)ctx->cfg.max_trajectories * sizeof(TrajInfo)));
    }
    // Rollout workgroups ride along with k_bin_count.  Few, fat workgroups win: dispatching a
    // 1024-lane workgroup costs ~12 ns, which is what bounds the launch on big shards (C4:
    // 64 trajectories per workgroup 44 us, 32: 56 us, 16: 86 us), and on small ones ~128
    // workgroups are the sweet spot (C2: 16 per workgroup 14.4 us, 32: 12.1 us, 64: 13.2 us).
    // Round 2: the launch's dynamic LDS (the rollout rows, 16 bytes per pair) is allocated by EVERY workgroup of
    // k_bin_count, and a CU holds two 1024-lane workgroups at most (wave slots).  Rows sized for two per CU
    // (<= 74 KB beside ~6 KB of static LDS) keep the whole launch resident in one round at C3 (the rollout
    // workgroups used to start in two rounds: k_bin_count 28 -> ~18 us).  Within a quarter of that cap the row count
    // that fills phase B's 1024-lane passes best wins (C3: 50 x 81 pairs = 3.96 passes, C4: 80 x 51 = 3.98).
    const int s1 = s_tick + 1;
    const int rt_lds = (int)std::min<size_t>((size_t)kRolloutMax, ((size_t)74 * 1024 - 16) / ((size_t)s1 * 16));
    int rt = std::min(std::max((k.n_local + 127) / 128, 4), std::max(rt_lds, 1));
    if (rt == rt_lds && rt > 4) {
      double best_fill = 0.0;
      for (int c = rt_lds; c >= rt_lds - rt_lds / 4; --c) {
        const int items = c * s1;
        const double fill = (double)items / (double)((items + kBinThreads - 1) / kBinThreads * kBinThreads);
        if (fill > best_fill + 1e-9) { best_fill = fill; rt = c; }
      }
    }
    if (ctx->rt_override > 0) rt = std::min(ctx->rt_override, kRolloutMax);
    while (rt > 1 && rollout_lds_bytes(rt, s_tick) > (size_t)128 * 1024) --rt;
    k.rt = rt;
  }
  if (pending) {
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->cloud_ready[cidx], 0));
    cloud_wait_done(ctx, cidx);
  }

  const auto prof_t1 = std::chrono::steady_clock::now();
  if (ctx->poison && k.n_local > 0) {
    HIPCHK(ctx, hipMemsetAsync(ctx->costs, 0xFF, (size_t)k.n_local * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->steps, 0xFF, (size_t)k.n_local * sizeof(int32_t), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->samples_out, 0xFF, (size_t)k.n_local * sizeof(float4), ctx->stream));
  }
  k.seq = ++ctx->seq;
  if (k.seq == 0) k.seq = ctx->seq = 1;
  // HIP events serialise the queue around them (~3 us each); timed ticks are sampled
  const bool timed = ctx->timing >= 1 && (ctx->seq % (uint32_t)ctx->timing_every) == 0;
  const bool timed_all = timed && ctx->timing >= 2;
  if (timed_all) HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const int bin_blocks = std::max(1, std::min(2048, (k.n_points + 255) / 256));
  // one point per lane while that needs few workgroups (latency), kBinPer per lane beyond (dispatch cost)
  const int per_wg = k.n_points <= 128 * kBinThreads ? kBinThreads : kBinThreads * kBinPer;
  const int cnt_blocks = std::max(1, std::min(512, (k.n_points + per_wg - 1) / per_wg));
  const int roll_blocks = k.n_local > 0 ? (k.n_local + k.rt - 1) / k.rt : 0;
  const size_t roll_lds = k.n_local > 0 ? rollout_lds_bytes(k.rt, s_tick) : 0;
  k.bin_blocks = cnt_blocks;
  k.roll_blocks = roll_blocks;
  // Load feedback: valid when the previous tick scored the same shard of the same theory
  // (its loads are indexed by local trajectory).  Otherwise this tick deals strided.
  const int theory_id = (int)(th - ctx->theories.data());
  k.n_tiles = k.n_local > 0 ? (k.n_local + tile - 1) / tile : 0;
  k.assign_groups = std::max(1, (k.n_local + kAssignPer * kBinThreads - 1) / (kAssignPer * kBinThreads));
  k.use_assign = (!ctx->no_assign && k.n_tiles > 1 && k.n_local <= kAssignMax && ctx->load_theory == theory_id &&
                  ctx->load_nlocal == k.n_local) ? 1 : 0;
  if (k.use_assign) k.n_tiles = (k.n_tiles + k.assign_groups - 1) / k.assign_groups * k.assign_groups;
  k.nb_tiles = k.n_tiles;
  k.r0 = 0;
  // Several rounds of resident workgroups: full workgroups for the whole rounds, ONE last round of short workgroups
  // for the rest (rollout_kernels.hip.h, tile_slot()); the lightest trajectories of the load-feedback deal land in it.
  // Built, bit-identical, measured and left OFF (DDDMR_TAIL_ROUND=1): a k_score workgroup's life is mostly fixed cost
  // (staging, ~13 barriers, scans), so 512 two-trajectory workgroups cost the C3 launch what its 171 full ones did:
  // C3 tick 156.2 -> 159.1 us, C4 337.3 -> 330.3 us (profiles/r03_tail_round.txt).
  if (!one_round && tile > 1 && ctx->tail_round && k.n_local <= kAssignMax) {
    const int G = k.assign_groups;
    const long slots = (long)ctx->n_cu * (thr == 512 ? 2 : 4);
    const long whole = (long)k.n_local / (slots * tile);                         // rounds of full workgroups
    const long rem = (long)k.n_local - whole * slots * tile;
    const int t2 = (int)((rem + slots - 1) / slots);                             // trajectories of a short workgroup
    if (whole >= 1 && rem > 0 && t2 < tile) {
      const int nb = (int)((whole * slots + G - 1) / G * G), ns = (int)((slots + G - 1) / G * G);
      if ((long)nb * tile + (long)ns * t2 >= k.n_local) {
        k.nb_tiles = nb;
        k.n_tiles = nb + ns;
        k.r0 = tile - t2;
      }
    }
  }
  ctx->load_theory = theory_id;
  ctx->load_nlocal = k.n_local;
  if (k.n_points > 0) {
    hipLaunchKernelGGL(k_bin_count, dim3(cnt_blocks + roll_blocks + (k.use_assign ? k.assign_groups : 0)), dim3(kBinThreads), roll_lds, ctx->stream,
                       k, ctx->cloud_dev[cidx], ctx->cell_count, ctx->cell_start, ctx->pt_slot, ctx->tickets,
                       ctx->best_key, ctx->overflow, ctx->axes_dev, ctx->samples_dev, ctx->traj_info, ctx->st_sc,
                       ctx->st_xy, ctx->traj_load, ctx->assign);
    hipLaunchKernelGGL(k_bin_scatter, dim3(bin_blocks), dim3(256), 0, ctx->stream, k, ctx->cloud_dev[cidx],
                       ctx->pt_slot, ctx->cell_start, ctx->sorted, ctx->row_tab);
  } else {
    hipLaunchKernelGGL(k_bin_reset, dim3(1), dim3(256), 0, ctx->stream, k, ctx->cell_count, ctx->cell_start,
                       ctx->best_key, ctx->overflow);
    if (roll_blocks > 0)
      hipLaunchKernelGGL(k_rollout, dim3(roll_blocks), dim3(256), roll_lds, ctx->stream, k, ctx->axes_dev,
                         ctx->samples_dev, ctx->traj_info, ctx->st_sc, ctx->st_xy);
    if (k.use_assign)
      hipLaunchKernelGGL(k_assign, dim3(k.assign_groups), dim3(kBinThreads), 0, ctx->stream, k, ctx->traj_load, ctx->assign);
  }
  if (timed) HIPCHK(ctx, hipEventRecord(ctx->evs0, ctx->stream));
  // multi-rank context: k_score leaves the shard's winner on the device (staging record + its slot of
  // the all-reduce), k_resolve publishes the global one
  const bool exchange = ctx->comm || ctx->comm_loopback;
  DevResult* score_result = exchange ? ctx->local_result_dev : ctx->result_dev;
  int64_t* score_words = exchange ? ctx->slots_dev + 2 * rank : nullptr;
  if (k.n_local > 0) {
    const int wgs = k.n_tiles;
    const bool lean = !k.want_minmax && !k.rec_pose && k.box_fast;
#define DDDMR_LAUNCH_SCORE(T, L)                                                                                   \
  do { if (k.probe) DDDMR_LAUNCH_SCORE_P(T, L, true); else DDDMR_LAUNCH_SCORE_P(T, L, false); } while (0)
#define DDDMR_LAUNCH_SCORE_P(T, L, P)                                                                              \
  hipLaunchKernelGGL((k_score<T, L, P>), dim3(wgs), dim3(T), lds, ctx->stream, k, ctx->traj_info, ctx->st_sc,     \
                     ctx->st_xy, ctx->plan_dev, ctx->cell_start, ctx->sorted, ctx->costs, ctx->steps,             \
                     ctx->samples_out, ctx->best_key, ctx->overflow, ctx->tickets + 1, score_result, ctx->assign, \
                     ctx->traj_load, score_words, ctx->row_tab)
    if (thr == 512) { if (lean) DDDMR_LAUNCH_SCORE(512, true); else DDDMR_LAUNCH_SCORE(512, false); }
    else            { if (lean) DDDMR_LAUNCH_SCORE(256, true); else DDDMR_LAUNCH_SCORE(256, false); }
#undef DDDMR_LAUNCH_SCORE
#undef DDDMR_LAUNCH_SCORE_P
  } else {
    hipLaunchKernelGGL(k_empty_result, dim3(1), dim3(64), 0, ctx->stream, k, ctx->cell_start, score_result, score_words);
  }
  if (timed) HIPCHK(ctx, hipEventRecord(ctx->evs1, ctx->stream));
  if (k.n_local > 0 && k.final_kernel)
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, ctx->stream, k, ctx->best_key, ctx->costs, ctx->samples_out,
                       ctx->cell_start, ctx->overflow, score_result, score_words);
  if (exchange) {
    // ONE collective per tick, on the tick's stream: ordered after k_score / k_finalize (which wrote this rank's two
    // words of slots_dev) and before k_resolve by stream order alone; no host synchronisation in between.
    if (ctx->comm) {
      const int nrc = rccl().all_reduce(ctx->slots_dev, ctx->slots_red, (size_t)2 * ctx->comm_ranks, ncclInt64, ncclMin,
                                        ctx->comm, ctx->stream);
      if (nrc != ncclSuccess) return fail(ctx, DDDMR_ERR_HIP, "ncclAllReduce failed: %s", rccl().error_string((ncclResult_t)nrc));
    } else {
      HIPCHK(ctx, hipMemcpyAsync(ctx->slots_red, ctx->slots_dev, (size_t)2 * ctx->comm_ranks * sizeof(int64_t),
                                 hipMemcpyDeviceToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(k_resolve, dim3(1), dim3(64), 0, ctx->stream, k, ctx->slots_red, ctx->comm_ranks,
                       ctx->local_result_dev, ctx->axes_dev, ctx->samples_dev, ctx->result_dev);
  }
  if (timed_all) HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(ctx, hipGetLastError());
  if (ctx->host_prof) {
    const auto prof_t2 = std::chrono::steady_clock::now();
    ctx->prof_ns[0] += std::chrono::duration<double, std::nano>(prof_t1 - prof_t0).count();
    ctx->prof_ns[1] += std::chrono::duration<double, std::nano>(prof_t2 - prof_t1).count();
    ++ctx->prof_n;
  }
  unbusy.armed = false;   // the cloud buffer stays pinned until tick_collect
  ctx->pend.active = true;
  ctx->pend.k = k;
  ctx->pend.s_tick = s_tick;
  ctx->pend.timed = timed;
  ctx->pend.timed_all = timed_all;
  ctx->pend.head = *out;
  return DDDMR_OK;
}

// Wait for the enqueued tick and decode its result; tick_mu must be held.
int tick_collect(dddmr_rollout_ctx* ctx, dddmr_rollout_result* out) {
  if (!ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "tick_end without tick_begin");
  ctx->pend.active = false;
  struct Release { dddmr_rollout_ctx* c; ~Release() { release_cloud(c); } } release{ctx};
  const DevTick& k = ctx->pend.k;
  *out = ctx->pend.head;
  const auto prof_c0 = std::chrono::steady_clock::now();
  // The last k_score workgroup stores the result into host-mapped memory and then
  // the tick's sequence number (system-scope release): polling it beats a stream
  // synchronise by several microseconds.  Bounded; falls back to the stream sync.
  bool seen = false;
  if (ctx->spin) {
    volatile uint32_t* seq_p = &ctx->result_host->seq;
    for (uint64_t spins = 0; spins < (1ull << 26); ++spins) {
      if (*seq_p == k.seq) { seen = true; break; }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!seen) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->pend.timed) {
    HIPCHK(ctx, hipEventSynchronize(ctx->pend.timed_all ? ctx->ev1 : ctx->evs1));
    if (ctx->pend.timed_all) HIPCHK(ctx, hipEventElapsedTime(&ctx->last_device_ms, ctx->ev0, ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(&ctx->last_score_ms, ctx->evs0, ctx->evs1));
  }
  const float ms = ctx->last_device_ms, score_ms = ctx->last_score_ms;   // latest sampled values

  if (ctx->host_prof) ctx->prof_ns[2] += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - prof_c0).count();
  const DevResult r = *ctx->result_host;
  ctx->last_result = r;
  if (k.n_local > 0) ctx->collided_share = (float)r.n_collided / (float)k.n_local;
  ctx->last = k;
  ctx->last_window = ctx->pend.window;
  ctx->have_last = true;
  if (r.overflow)
    return fail(ctx, DDDMR_ERR_CAPACITY, "device capacity flag %u (1: trajectory longer than %d steps, 2: cuboid spans more than %d cell rows)",
                r.overflow, ctx->pend.s_tick, kRows);
  out->device_ms = ms;
  out->score_ms = score_ms;
  out->n_points_binned = r.n_binned;
  out->key = r.key;
  if (r.index >= 0) {
    out->planner_state = DDDMR_TRAJECTORY_FOUND;
    out->best_index = r.index;
    out->best_cost = r.cost;
    out->vx = r.vx; out->vy = r.vy; out->wz = r.wz;
  }
  return DDDMR_OK;
}

}  // namespace

extern "C" {

int dddmr_rollout_tick(dddmr_rollout_ctx* ctx, const char* theory_name, const dddmr_tick_input* in,
                       dddmr_rollout_result* out) {
  if (!ctx || !theory_name || !in || !out) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "tick while a tick_begin is pending");
  std::memset(out, 0, sizeof(*out));
  out->planner_state = DDDMR_ALL_TRAJECTORIES_FAIL;
  out->best_index = -1;
  out->best_cost = -1.0;
  out->key = kKeyNone;
  const int rc = tick_enqueue(ctx, theory_name, in);
  if (rc != DDDMR_OK) return rc;
  return tick_collect(ctx, out);
}

int dddmr_rollout_tick_begin(dddmr_rollout_ctx* ctx, const char* theory_name, const dddmr_tick_input* in) {
  if (!ctx || !theory_name || !in) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "tick_begin while another tick_begin is pending");
  return tick_enqueue(ctx, theory_name, in);
}

int dddmr_rollout_tick_end(dddmr_rollout_ctx* ctx, dddmr_rollout_result* out) {
  if (!ctx || !out) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  return tick_collect(ctx, out);
}

// the command of global sample `idx` of the last collected tick (samples are a closed-form grid, or the
// tick's explicit list: every rank can recompute the winner's command from its index)
static void sample_of(const Window& w, int idx, float* vx, float* vy, float* wz) {
  if (w.list_mode) {
    *vx = w.list[idx].x; *vy = w.list[idx].y; *wz = w.list[idx].z;
  } else {
    const int nth = (int)w.ath.size(), ny = (int)w.ay.size();
    *vx = w.ax[(idx / nth) / ny];
    *vy = w.ay[(idx / nth) % ny];
    *wz = w.ath[idx % nth];
  }
}

int dddmr_rollout_resolve(dddmr_rollout_ctx* ctx, int64_t reduced_key, dddmr_rollout_result* inout) {
  if (!ctx || !inout) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->have_last) return fail(ctx, DDDMR_ERR_STATE, "resolve before any tick");
  inout->key = reduced_key;
  const int32_t idx = key_index(reduced_key);
  if (idx < 0) {
    inout->planner_state = DDDMR_ALL_TRAJECTORIES_FAIL;
    inout->best_index = -1;
    inout->best_cost = -1.0;
    inout->vx = inout->vy = inout->wz = 0.0;
    return DDDMR_OK;
  }
  if (idx >= ctx->last.n_global) return fail(ctx, DDDMR_ERR_BAD_ARG, "resolve: index %d out of range", idx);
  float vx, vy, wz;
  sample_of(ctx->last_window, idx, &vx, &vy, &wz);
  inout->planner_state = DDDMR_TRAJECTORY_FOUND;
  inout->best_index = idx;
  inout->vx = vx; inout->vy = vy; inout->wz = wz;
  if (ctx->last_result.index == idx) {
    inout->best_cost = ctx->last_result.cost;   // this rank's own winner: exact
  } else {
    // winner lives on another rank (or an older tick): the key carries the cost's top 40 bits
    union { double d; uint64_t u; } cv;
    cv.u = (uint64_t)reduced_key & ~((1ull << kKeyIndexBits) - 1);
    inout->best_cost = cv.d;
  }
  return DDDMR_OK;
}

void dddmr_rollout_winner_words(const dddmr_rollout_result* r, int64_t words[2]) {
  words[0] = words[1] = INT64_MAX;
  if (!r || r->best_index < 0) return;
  words[0] = cost_bits(r->best_cost);
  if (words[0] != INT64_MAX) words[1] = -(int64_t)r->best_index;
}

int dddmr_rollout_resolve_words(dddmr_rollout_ctx* ctx, const int64_t* words, int32_t n_ranks,
                                dddmr_rollout_result* inout) {
  if (!ctx || !inout || !words || n_ranks <= 0) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->have_last) return fail(ctx, DDDMR_ERR_STATE, "resolve before any tick");
  // minimum cost (full doubles), equal costs -> highest index: local_planner.cpp:460 over the whole batch
  int64_t c = INT64_MAX, ni = INT64_MAX;
  for (int r = 0; r < n_ranks; ++r) {
    const int64_t cr = words[2 * r], ir = words[2 * r + 1];
    if (cr == INT64_MAX) continue;
    if (cr < c || (cr == c && ir < ni)) { c = cr; ni = ir; }
  }
  if (c == INT64_MAX) {
    inout->planner_state = DDDMR_ALL_TRAJECTORIES_FAIL;
    inout->best_index = -1;
    inout->best_cost = -1.0;
    inout->vx = inout->vy = inout->wz = 0.0;
    inout->key = kKeyNone;
    return DDDMR_OK;
  }
  const int64_t idx = -ni;
  if (idx < 0 || idx >= ctx->last.n_global) return fail(ctx, DDDMR_ERR_BAD_ARG, "resolve_words: index %lld out of range", (long long)idx);
  float vx, vy, wz;
  sample_of(ctx->last_window, (int)idx, &vx, &vy, &wz);
  union { double d; int64_t i; } cv;
  cv.i = c;
  inout->planner_state = DDDMR_TRAJECTORY_FOUND;
  inout->best_index = (int32_t)idx;
  inout->best_cost = cv.d;
  inout->vx = vx; inout->vy = vy; inout->wz = wz;
  inout->key = pack_key(cv.d, (uint32_t)idx);
  return DDDMR_OK;
}

int dddmr_rollout_comm_unique_id(uint8_t id_out[DDDMR_COMM_ID_BYTES]) {
  static_assert(DDDMR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (!id_out) return DDDMR_ERR_BAD_ARG;
  if (!rccl().ok()) return DDDMR_ERR_NO_DEVICE;
  ncclUniqueId id;
  if (rccl().get_unique_id(&id) != ncclSuccess) return DDDMR_ERR_HIP;
  std::memcpy(id_out, id.internal, DDDMR_COMM_ID_BYTES);
  return DDDMR_OK;
}

// slot vectors of the per-tick exchange: [2 * n_ranks] words to send (own pair written by k_score / k_finalize,
// INT64_MAX elsewhere: the other ranks' slots never change) and to receive
static int alloc_exchange_buffers(dddmr_rollout_ctx* ctx, int n_ranks) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->slots_dev) { (void)hipFree(ctx->slots_dev); ctx->slots_dev = nullptr; }      // (re-initialisation after comm_destroy)
  if (ctx->slots_red) { (void)hipFree(ctx->slots_red); ctx->slots_red = nullptr; }
  if (ctx->local_result_dev) { (void)hipFree(ctx->local_result_dev); ctx->local_result_dev = nullptr; }
  HIPCHK(ctx, hipMalloc(&ctx->slots_dev, (size_t)2 * n_ranks * sizeof(int64_t)));
  HIPCHK(ctx, hipMalloc(&ctx->slots_red, (size_t)2 * n_ranks * sizeof(int64_t)));
  HIPCHK(ctx, hipMalloc(&ctx->local_result_dev, sizeof(DevResult)));
  std::vector<int64_t> none((size_t)2 * n_ranks, INT64_MAX);
  HIPCHK(ctx, hipMemcpy(ctx->slots_dev, none.data(), none.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->slots_red, none.data(), none.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  return DDDMR_OK;
}

int dddmr_rollout_comm_init(dddmr_rollout_ctx* ctx, const uint8_t id[DDDMR_COMM_ID_BYTES], int32_t rank, int32_t n_ranks) {
  if (!ctx || !id) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "comm_init while a tick_begin is pending");
  if (ctx->comm) return fail(ctx, DDDMR_ERR_STATE, "comm_init: the context already has a communicator");
  const int world = std::max(1, ctx->cfg.world_size);
  if (n_ranks != world || rank != std::min(std::max(0, ctx->cfg.rank), world - 1))
    return fail(ctx, DDDMR_ERR_BAD_ARG, "comm_init: rank %d of %d does not match the context's shard (rank %d of %d)", rank,
                n_ranks, ctx->cfg.rank, world);
  if (ctx->comm_loopback) return fail(ctx, DDDMR_ERR_STATE, "comm_init: the context is in loopback mode");
  if (!rccl().ok()) return fail(ctx, DDDMR_ERR_NO_DEVICE, "comm_init: %s", rccl().why.c_str());
  const int ab = alloc_exchange_buffers(ctx, n_ranks);
  if (ab != DDDMR_OK) return ab;
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, DDDMR_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = rccl().comm_init_rank(&comm, n_ranks, uid, rank);     // collective: every rank calls it
  if (rc != ncclSuccess) return fail(ctx, DDDMR_ERR_HIP, "ncclCommInitRank failed: %s", rccl().error_string(rc));
  ctx->comm = comm;
  ctx->comm_ranks = n_ranks;
  return DDDMR_OK;
}

int dddmr_rollout_comm_destroy(dddmr_rollout_ctx* ctx) {
  if (!ctx) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "comm_destroy while a tick_begin is pending");
  if (!ctx->comm && !ctx->comm_loopback) return DDDMR_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->comm) (void)rccl().comm_destroy(ctx->comm);
  ctx->comm = nullptr;
  ctx->comm_loopback = false;
  ctx->comm_ranks = 0;
  return DDDMR_OK;
}

int dddmr_rollout_device_count(int32_t* n_out) {
  if (!n_out) return DDDMR_ERR_BAD_ARG;
  int n = 0;
  *n_out = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return DDDMR_ERR_NO_DEVICE; }
  *n_out = n;
  return DDDMR_OK;
}

int dddmr_rollout_comm_ranks(dddmr_rollout_ctx* ctx, int32_t* n_ranks_out) {
  if (!ctx || !n_ranks_out) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  *n_ranks_out = 0;
  if (ctx->comm_loopback) { *n_ranks_out = ctx->comm_ranks; return DDDMR_OK; }
  if (!ctx->comm) return DDDMR_OK;
  int n = 0;
  const ncclResult_t rc = rccl().comm_count(ctx->comm, &n);      // what RCCL itself says, not what we asked for
  if (rc != ncclSuccess) return fail(ctx, DDDMR_ERR_HIP, "ncclCommCount failed: %s", rccl().error_string(rc));
  *n_ranks_out = n;
  return DDDMR_OK;
}

int dddmr_rollout_comm_loopback(dddmr_rollout_ctx* ctx) {
  if (!ctx) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "comm_loopback while a tick_begin is pending");
  if (ctx->comm || ctx->comm_loopback) return fail(ctx, DDDMR_ERR_STATE, "comm_loopback: the context already has an exchange");
  const int world = std::max(1, ctx->cfg.world_size);
  const int ab = alloc_exchange_buffers(ctx, world);
  if (ab != DDDMR_OK) return ab;
  ctx->comm_loopback = true;
  ctx->comm_ranks = world;
  return DDDMR_OK;
}

int dddmr_rollout_comm_loopback_set_peer(dddmr_rollout_ctx* ctx, int32_t peer_rank, const int64_t words[2]) {
  if (!ctx || !words) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->comm_loopback) return fail(ctx, DDDMR_ERR_STATE, "comm_loopback_set_peer: the context is not in loopback mode");
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "comm_loopback_set_peer while a tick_begin is pending");
  const int world = ctx->comm_ranks, rank = std::min(std::max(0, ctx->cfg.rank), world - 1);
  if (peer_rank < 0 || peer_rank >= world || peer_rank == rank)
    return fail(ctx, DDDMR_ERR_BAD_ARG, "comm_loopback_set_peer: peer %d of %d (own rank %d)", peer_rank, world, rank);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipMemcpy(ctx->slots_dev + 2 * peer_rank, words, 2 * sizeof(int64_t), hipMemcpyHostToDevice));
  return DDDMR_OK;
}

// Stream ceiling of this GPU ("a measured stream-copy ceiling on the same GPU ... both
// denominators"): `bytes` per buffer (>= 1 GiB defeats the 256 MB of MALL), `reps` launches each.
int dddmr_rollout_selftest_sincos(dddmr_rollout_ctx* ctx, const double* angles, size_t n, double* sin_out,
                                  double* cos_out) {
  if (!ctx || !angles || !sin_out || !cos_out || n == 0 || n > ((size_t)1 << 24)) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "selftest_sincos while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double* buf = nullptr;
  HIPCHK(ctx, hipMalloc(&buf, 3 * n * sizeof(double)));
  int rc = DDDMR_OK;
  auto run = [&]() -> int {
    HIPCHK(ctx, hipMemcpyAsync(buf, angles, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_selftest_sincos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, buf, (int)n,
                       buf + n, buf + 2 * n);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(sin_out, buf + n, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(cos_out, buf + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return DDDMR_OK;
  };
  rc = run();
  (void)hipFree(buf);
  return rc;
}

int dddmr_rollout_stream_ceiling(dddmr_rollout_ctx* ctx, size_t bytes, int32_t reps, double* copy_gbps,
                                 double* read_gbps) {
  if (!ctx || !copy_gbps || !read_gbps || reps <= 0 || bytes < (1u << 20)) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "stream_ceiling while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = bytes / sizeof(float4);
  // one pass of the four-loads-in-flight loop per lane measured best (grid sweep 1k ... 64k workgroups:
  // copy 4.5 -> 5.2 TB/s, read 5.7 -> 6.0 TB/s; tools/exp_ceiling.py)
  int blocks = (int)std::min<size_t>(65536, std::max<size_t>(1, n / (256 * 4)));
  if (const char* e = std::getenv("DDDMR_CEIL_BLOCKS")) blocks = std::max(1, std::atoi(e));
  float4 *src = nullptr, *dst = nullptr;
  float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = DDDMR_OK;
  auto run = [&]() -> int {
    HIPCHK(ctx, hipMalloc(&src, n * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&dst, n * sizeof(float4)));
    HIPCHK(ctx, hipMalloc(&sink, (size_t)blocks * 256 * sizeof(float)));
    HIPCHK(ctx, hipMemsetAsync(src, 0x11, n * sizeof(float4), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(dst, 0, n * sizeof(float4), ctx->stream));
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    float ms = 0.f;
    hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, ctx->stream, src, dst, n);   // warm-up
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, ctx->stream, src, dst, n);
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    *copy_gbps = 2.0 * (double)(n * sizeof(float4)) * reps / ((double)ms * 1e-3) / 1e9;
    hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, ctx->stream, src, sink, n);
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, ctx->stream, src, sink, n);
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(e1));
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    *read_gbps = (double)(n * sizeof(float4)) * reps / ((double)ms * 1e-3) / 1e9;
    HIPCHK(ctx, hipGetLastError());
    return DDDMR_OK;
  };
  rc = run();
  (void)hipStreamSynchronize(ctx->stream);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  if (sink) (void)hipFree(sink);
  return rc;
}

// The theory's initialise() alone: the velocity samples a tick with these inputs would roll out, in the
// reference's generation order (x-major, y, theta-minor).  Host-only, no device work.
int dddmr_rollout_samples(dddmr_rollout_ctx* ctx, const char* theory_name, const dddmr_tick_input* in, float* samples_out,
                          size_t capacity, size_t* n_samples) {
  if (!ctx || !theory_name || !in || !n_samples) return DDDMR_ERR_BAD_ARG;
  const dddmr_theory_config* th = find_theory(ctx, theory_name);
  if (!th) return fail(ctx, DDDMR_ERR_UNKNOWN_THEORY, "unknown theory '%s'", theory_name);
  Window w;
  make_window(*th, *in, w);
  const size_t n = w.count();
  *n_samples = n;
  if (!samples_out) return DDDMR_OK;
  if (capacity < n) return fail(ctx, DDDMR_ERR_CAPACITY, "samples: capacity %zu < %zu", capacity, n);
  for (size_t i = 0; i < n; ++i) sample_of(w, (int)i, samples_out + 3 * i, samples_out + 3 * i + 1, samples_out + 3 * i + 2);
  return DDDMR_OK;
}

int dddmr_rollout_get_debug(dddmr_rollout_ctx* ctx, dddmr_rollout_debug* dbg) {
  if (!ctx || !dbg) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->have_last) return fail(ctx, DDDMR_ERR_STATE, "get_debug before any tick");
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "get_debug while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // the tick may have returned on the polled sequence number
  const size_t n = (size_t)ctx->last.n_local;
  if (n == 0) return DDDMR_OK;
  if (dbg->costs) HIPCHK(ctx, hipMemcpy(dbg->costs, ctx->costs, n * sizeof(double), hipMemcpyDeviceToHost));
  if (dbg->steps) HIPCHK(ctx, hipMemcpy(dbg->steps, ctx->steps, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (dbg->samples) {
    std::vector<float4> tmp(n);
    HIPCHK(ctx, hipMemcpy(tmp.data(), ctx->samples_out, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      dbg->samples[3 * i + 0] = tmp[i].x;
      dbg->samples[3 * i + 1] = tmp[i].y;
      dbg->samples[3 * i + 2] = tmp[i].z;
    }
  }
  return DDDMR_OK;
}

int dddmr_rollout_get_pose_arrays(dddmr_rollout_ctx* ctx, int32_t which, double* poses_out, size_t capacity,
                                  size_t* n_poses) {
  if (!ctx || !n_poses || (which != 0 && which != 1)) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->have_last) return fail(ctx, DDDMR_ERR_STATE, "get_pose_arrays before any tick");
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "get_pose_arrays while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const int n = ctx->last.n_local;
  *n_poses = 0;
  if (n <= 0) return DDDMR_OK;
  std::vector<int32_t> steps(n), off(n);
  std::vector<double> costs(n);
  HIPCHK(ctx, hipMemcpy(steps.data(), ctx->steps, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(costs.data(), ctx->costs, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    // generated: nextTrajectory returned true (steps > 0); accepted: cost_ >= 0 (local_planner.cpp:463)
    const bool want = steps[i] > 0 && (which == 0 || costs[i] >= 0.0);
    off[i] = want ? (int32_t)total : -1;
    if (want) total += (size_t)steps[i];
  }
  *n_poses = total;
  if (!poses_out || total == 0) return DDDMR_OK;
  if (capacity < total) return fail(ctx, DDDMR_ERR_CAPACITY, "get_pose_arrays: capacity %zu < %zu", capacity, total);
  if (total > (size_t)INT32_MAX) return fail(ctx, DDDMR_ERR_CAPACITY, "get_pose_arrays: %zu poses", total);
  int32_t* off_dev = nullptr;
  double* out_dev = nullptr;
  HIPCHK(ctx, hipMalloc(&off_dev, (size_t)n * sizeof(int32_t)));
  if (hipMalloc(&out_dev, total * 7 * sizeof(double)) != hipSuccess) {
    (void)hipFree(off_dev);
    return fail(ctx, DDDMR_ERR_HIP, "get_pose_arrays: out of device memory for %zu poses", total);
  }
  int rc = DDDMR_OK;
  const size_t pairs = (size_t)n * (size_t)ctx->last.max_steps;
  if (hipMemcpyAsync(off_dev, off.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
    rc = DDDMR_ERR_HIP;
  if (rc == DDDMR_OK) {
    hipLaunchKernelGGL(k_pose_arrays, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, ctx->stream, ctx->last, off_dev,
                       ctx->steps, ctx->st_sc, ctx->st_xy, out_dev);
    if (hipMemcpyAsync(poses_out, out_dev, total * 7 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
      rc = DDDMR_ERR_HIP;
  }
  (void)hipFree(off_dev);
  (void)hipFree(out_dev);
  if (rc != DDDMR_OK) return fail(ctx, rc, "get_pose_arrays: %s", hipGetErrorString(hipGetLastError()));
  return DDDMR_OK;
}

int dddmr_rollout_get_best_poses(dddmr_rollout_ctx* ctx, double* poses_out, size_t capacity, size_t* n_poses) {
  if (!ctx || !n_poses) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->have_last) return fail(ctx, DDDMR_ERR_STATE, "get_best_poses before any tick");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "get_best_poses while a tick_begin is pending");
  const int32_t idx = ctx->last_result.index;
  *n_poses = 0;
  if (idx < 0) return DDDMR_OK;
  const int li = idx - ctx->last.begin;
  if (li < 0 || li >= ctx->last.n_local) return DDDMR_OK;  // winner is on another rank
  int32_t ns = 0;
  HIPCHK(ctx, hipMemcpy(&ns, ctx->steps + li, sizeof(int32_t), hipMemcpyDeviceToHost));
  *n_poses = (size_t)ns;
  if (!poses_out) return DDDMR_OK;
  if (capacity < (size_t)ns) return fail(ctx, DDDMR_ERR_CAPACITY, "get_best_poses: capacity %zu < %d", capacity, ns);
  hipLaunchKernelGGL(k_trajectory_poses, dim3(1), dim3(64), 0, ctx->stream, ctx->last, li, ctx->samples_out,
                     ctx->steps, ctx->poses_dev, (float*)nullptr);
  HIPCHK(ctx, hipMemcpyAsync(poses_out, ctx->poses_dev, (size_t)ns * 7 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return DDDMR_OK;
}

int dddmr_rollout_get_best_cuboids(dddmr_rollout_ctx* ctx, float* vertices_out, size_t capacity_poses, size_t* n_poses) {
  if (!ctx || !n_poses) return DDDMR_ERR_BAD_ARG;
  std::lock_guard<std::mutex> tk(ctx->tick_mu);
  if (!ctx->have_last) return fail(ctx, DDDMR_ERR_STATE, "get_best_cuboids before any tick");
  if (ctx->pend.active) return fail(ctx, DDDMR_ERR_STATE, "get_best_cuboids while a tick_begin is pending");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t idx = ctx->last_result.index;
  *n_poses = 0;
  if (idx < 0) return DDDMR_OK;
  const int li = idx - ctx->last.begin;
  if (li < 0 || li >= ctx->last.n_local) return DDDMR_OK;  // winner is on another rank
  int32_t ns = 0;
  HIPCHK(ctx, hipMemcpy(&ns, ctx->steps + li, sizeof(int32_t), hipMemcpyDeviceToHost));
  *n_poses = (size_t)ns;
  if (!vertices_out || ns == 0) return DDDMR_OK;
  if (capacity_poses < (size_t)ns) return fail(ctx, DDDMR_ERR_CAPACITY, "get_best_cuboids: capacity %zu < %d", capacity_poses, ns);
  float* cub_dev = nullptr;
  HIPCHK(ctx, hipMalloc(&cub_dev, (size_t)ns * 24 * sizeof(float)));
  hipLaunchKernelGGL(k_trajectory_poses, dim3(1), dim3(64), 0, ctx->stream, ctx->last, li, ctx->samples_out,
                     ctx->steps, ctx->poses_dev, cub_dev);
  const hipError_t e = hipMemcpyAsync(vertices_out, cub_dev, (size_t)ns * 24 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);
  (void)hipFree(cub_dev);
  if (e != hipSuccess || e2 != hipSuccess) return fail(ctx, DDDMR_ERR_HIP, "get_best_cuboids: copy failed");
  return DDDMR_OK;
}

}  // extern "C"

#include "marking_host.hip.h"
