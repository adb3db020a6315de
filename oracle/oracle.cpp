/*
 * oracle.cpp -- CPU restatement of the dddmr_local_planner rollout hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (dddmr_navigation_amd/,
 * include/) links, imports or executes this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / reported baseline.
 *
 * PARITY UNPINNED: the reference has no tests, fixtures or golden vectors for
 * this path and cannot be built here (needs ROS 2 Humble,
 * PCL 1.15, FLANN, Eigen, tf2 -- none present, no network).  This file is a
 * literal restatement of the cited reference source with the reference's mixed
 * float/double arithmetic; the third-party arithmetic it depends on (Eigen
 * Affine3d/AngleAxisd/Quaterniond, PCL transformPointCloud / KdTreeFLANN,
 * tf2 Matrix3x3::getEulerYPR) is restated from those libraries' published
 * algorithms.  Known-answer counts derivable by hand from the reference source
 * (55 trajectories, 2363 / 967 steps for the playground / shipped configs,
 * 126 steps for in-place rotation) pin the step-count arithmetic.
 *
 * All file:line citations are relative to
 * /root/reference/src/dddmr_local_planner/ .
 *
 * Build: see oracle/Makefile (g++ -O3 -ffp-contract=off: the reference is
 * built for baseline x86-64, so no fused multiply-add is ever formed).
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../include/dddmr_rollout.h"
#include "oracle.h"

#include "oracle_common.h"

using namespace oracle_detail;

namespace {


// --------------------------------------------------------------------------
// base_trajectory::Trajectory (base_trajectory/include/base_trajectory/trajectory.h:47-126)
// --------------------------------------------------------------------------
struct Step {
  double pose[7];   // PoseStamped position + orientation
  F3 cuboid[8];     // transformed cuboid vertices (pcl::PointXYZ)
  F3 mn, mx;        // cuboid_min_max_t
  F3 pcl;           // PointXYZI position (trajectory.cpp:69-75)
};
struct Trajectory {
  double xv = 0.0, yv = 0.0, thetav = 0.0;
  double cost = -1.0;           // trajectory.cpp:34-37
  double time_delta = 0.0;
  std::vector<Step> steps;
};

// velocity_iterator.h:44-69.  no_zero_insert is the bench-mode extension.
static std::vector<double> velocity_iterator(double mn, double mx, int num_samples,
                                             bool no_zero_insert) {
  std::vector<double> samples;
  if (mn == mx) {
    samples.push_back(mn);
  } else {
    num_samples = std::max(2, num_samples);
    const double step_size = (mx - mn) / double(std::max(1, (num_samples - 1)));
    double current;
    double next = mn;
    for (int j = 0; j < num_samples - 1; ++j) {
      current = next;
      next += step_size;
      samples.push_back(current);
      if (!no_zero_insert && (current < 0) && (next > 0)) samples.push_back(0.0);
    }
    samples.push_back(mx);
  }
  return samples;
}

struct Sample { float v[3]; };

static bool dd_motor_ok(const dddmr_theory_config& c, const Sample& s, bool always) {
  // dd_simple...cpp:297-312 ; dd_rotate_inplace_theory.cpp:276-286 (always on)
  if (!always && !c.use_motor_constraint) return true;
  double vr = s.v[0] + c.robot_radius * s.v[2];
  double vl = s.v[0] - c.robot_radius * s.v[2];
  double rpm_r = vr * c.gear_ratio * 60. / 3.1415926 / c.wheel_diameter;
  double rpm_l = vl * c.gear_ratio * 60. / 3.1415926 / c.wheel_diameter;
  if (std::fabs(rpm_r) >= c.max_motor_shaft_rpm || std::fabs(rpm_l) >= c.max_motor_shaft_rpm)
    return false;
  return true;
}

// initialise() of the three theories.
static std::vector<Sample> make_samples(const dddmr_theory_config& c, const dddmr_tick_input& in) {
  std::vector<Sample> out;
  const bool nz = c.bench_no_zero_insert != 0;
  const double tx = in.robot_twist[0], ty = in.robot_twist[1], tw = in.robot_twist[2];
  const double max_vel_th = c.max_vel_theta;
  const double min_vel_th = -1.0 * max_vel_th;
  const float acc0 = (float)c.acc_lim_x, acc1 = (float)c.acc_lim_y, acc2 = (float)c.acc_lim_theta;
  double min_vel_x = c.min_vel_x, max_vel_x = c.max_vel_x;
  if (!(c.linear_x_sample * c.angular_z_sample > 0)) return out;
  const double sim_period = 1.0 / c.controller_frequency;
  float max_vel[3] = {0.f, 0.f, 0.f}, min_vel[3] = {0.f, 0.f, 0.f};

  if (c.kind == DDDMR_THEORY_DD_SIMPLE) {
    // dd_simple_trajectory_generator_theory.cpp:236-295
    if (in.allowed_max_linear_speed > 0.0)
      max_vel_x = std::min(max_vel_x, in.allowed_max_linear_speed);
    max_vel[0] = (float)std::min(max_vel_x, tx + acc0 * sim_period);
    max_vel[2] = (float)std::min(max_vel_th, tw + acc2 * sim_period);
    min_vel[0] = (float)std::max(min_vel_x, tx / c.deceleration_ratio);
    min_vel[2] = (float)std::max(min_vel_th, tw - acc2 * sim_period);
    if (max_vel[0] < min_vel[0]) {
      min_vel[0] = (float)(tx / c.deceleration_ratio);
      max_vel[0] = (float)(tx / c.deceleration_ratio);
    }
    auto xs = velocity_iterator(min_vel[0], max_vel[0], (int)c.linear_x_sample, nz);
    auto ths = velocity_iterator(min_vel[2], max_vel[2], (int)c.angular_z_sample, nz);
    for (double xv : xs) {
      Sample s{{(float)xv, 0.f, 0.f}};
      for (double th : ths) {
        s.v[2] = (float)th;
        if (dd_motor_ok(c, s, false)) out.push_back(s);
      }
    }
  } else if (c.kind == DDDMR_THEORY_OMNI_SIMPLE) {
    // omni_simple_trajectory_generator_theory.cpp:260-332
    const double min_vel_y = c.min_vel_y, max_vel_y = c.max_vel_y;
    max_vel[0] = (float)std::min(max_vel_x, tx + acc0 * sim_period);
    max_vel[1] = (float)std::min(max_vel_y, ty + acc1 * sim_period);
    max_vel[2] = (float)std::min(max_vel_th, tw + acc2 * sim_period);
    min_vel[0] = (float)std::max(min_vel_x, tx - acc0 * sim_period);
    min_vel[1] = (float)std::max(min_vel_y, ty - acc1 * sim_period);
    min_vel[2] = (float)std::max(min_vel_th, tw - acc2 * sim_period);
    if (tx >= max_vel_x / c.deceleration_ratio) {
      min_vel[0] = (float)std::max(min_vel_x, tx / c.deceleration_ratio);
    } else if (tx <= min_vel_x / c.deceleration_ratio) {
      max_vel[0] = (float)std::min(max_vel_x, tx / c.deceleration_ratio);
    }
    if (ty >= max_vel_y / c.deceleration_ratio) {
      min_vel[1] = (float)std::max(min_vel_y, ty / c.deceleration_ratio);
    } else if (ty <= min_vel_y / c.deceleration_ratio) {
      max_vel[1] = (float)std::min(max_vel_y, ty / c.deceleration_ratio);
    }
    auto xs = velocity_iterator(min_vel[0], max_vel[0], (int)c.linear_x_sample, nz);
    auto ys = velocity_iterator(min_vel[1], max_vel[1], (int)c.linear_y_sample, nz);
    auto ths = velocity_iterator(min_vel[2], max_vel[2], (int)c.angular_z_sample, nz);
    for (double xv : xs) {
      Sample s{{(float)xv, 0.f, 0.f}};
      for (double yv : ys) {
        s.v[1] = (float)yv;
        for (double th : ths) {
          s.v[2] = (float)th;
          out.push_back(s);  // isMotorConstraintSatisfied always true (:334-343)
        }
      }
    }
  } else {
    // dd_rotate_inplace_theory.cpp:229-274: two samples (0,0,+-rotation_speed)
    Sample pos{{0.f, 0.f, (float)c.rotation_speed}};
    Sample neg{{0.f, 0.f, (float)(-1.0 * c.rotation_speed)}};
    if (dd_motor_ok(c, pos, true)) out.push_back(pos);
    if (dd_motor_ok(c, neg, true)) out.push_back(neg);
  }
  return out;
}

// generateTrajectory() + computeNewPositions() of the three theories.
// dd_simple...cpp:351-464, omni_simple...cpp:382-505, dd_rotate_inplace...cpp:325-427
static bool generate_trajectory(const dddmr_theory_config& c, const dddmr_tick_input& in,
                                const Sample& sample, Trajectory& traj) {
  const Affine pos_af3 = transform_to_eigen(in.robot_pose);
  const float sv0 = sample.v[0], sv1 = sample.v[1], sv2 = sample.v[2];
  const double eps = 1e-4;
  traj.cost = 0.0;
  traj.steps.clear();
  double vmag;
  double sim_time = c.sim_time;

  if (c.kind == DDDMR_THEORY_DD_SIMPLE) {
    vmag = std::fabs((double)sv0);
    if ((c.min_vel_x >= 0 && vmag + eps < c.min_vel_x) &&
        (c.min_vel_theta >= 0 && std::fabs((double)sv2) + eps < c.min_vel_theta))
      return false;
    if (c.max_vel_x >= 0 && vmag - eps > c.max_vel_x) return false;
  } else if (c.kind == DDDMR_THEORY_OMNI_SIMPLE) {
    vmag = std::hypot((double)sv0, (double)sv1);
    if ((c.min_vel_trans >= 0 && vmag + eps < c.min_vel_trans) &&
        (c.min_vel_theta >= 0 && std::fabs((double)sv2) + eps < c.min_vel_theta))
      return false;
    if (c.max_vel_trans >= 0 && vmag - eps > c.max_vel_trans) return false;
    if (in.allowed_max_linear_speed > 0.0) {
      if (vmag - eps > in.allowed_max_linear_speed) return false;
    }
  } else {
    vmag = std::fabs((double)sv0);
    sim_time = 6.28 / std::fabs((double)sv2);  // a_rad_sim_time (:337)
  }

  int num_steps;
  if (c.bench_fixed_steps > 0) {
    num_steps = c.bench_fixed_steps;  // fixed variant, dd_simple...cpp:391-394
  } else {
    const double sim_time_distance = vmag * sim_time;
    const double sim_time_angle = std::fabs((double)sv2) * sim_time;
    num_steps = (int)std::ceil(std::max(sim_time_distance / c.sim_granularity,
                                        sim_time_angle / c.angular_sim_granularity));
  }
  if (num_steps == 0) return false;

  const double dt = sim_time / num_steps;
  traj.time_delta = dt;
  traj.xv = sv0;
  if (c.kind == DDDMR_THEORY_OMNI_SIMPLE) traj.yv = sv1;
  traj.thetav = sv2;

  float pos[3] = {0.f, 0.f, 0.f};
  traj.steps.reserve(num_steps);
  for (int i = 0; i < num_steps; ++i) {
    // computeNewPositions: state is Eigen::Vector3f, dt is double; cos/sin of a
    // float argument resolve to the float overloads (<math.h> is pulled in by
    // tf2/LinearMath/Scalar.h, so ::cos(float) is std::cos(float)).
    float np[3];
    if (c.kind == DDDMR_THEORY_OMNI_SIMPLE) {
      np[0] = (float)(pos[0] + (sv0 * cosf(pos[2]) + sv1 * std::cos(M_PI_2 + pos[2])) * dt);
      np[1] = (float)(pos[1] + (sv0 * sinf(pos[2]) + sv1 * std::sin(M_PI_2 + pos[2])) * dt);
    } else {
      np[0] = (float)(pos[0] + (sv0 * cosf(pos[2])) * dt);
      np[1] = (float)(pos[1] + (sv0 * sinf(pos[2])) * dt);
    }
    np[2] = (float)(pos[2] + sv2 * dt);
    pos[0] = np[0]; pos[1] = np[1]; pos[2] = np[2];

    Affine b2traj = angle_axis_z((double)pos[2]);
    b2traj.t[0] = pos[0];
    b2traj.t[1] = pos[1];
    const Affine g2t = mul(pos_af3, b2traj);

    Step st;
    eigen_to_transform(g2t, st.pose);
    // pcl::transformPointCloud(cuboid, out, Affine3d): double multiply-add per
    // coordinate, cast to float (pcl/common/impl/transforms.hpp, SSE/AVX off).
    for (int k = 0; k < 8; ++k) {
      const double px = c.cuboid[k][0], py = c.cuboid[k][1], pz = c.cuboid[k][2];
      st.cuboid[k].x = (float)(g2t.l[0][0] * px + g2t.l[0][1] * py + g2t.l[0][2] * pz + g2t.t[0]);
      st.cuboid[k].y = (float)(g2t.l[1][0] * px + g2t.l[1][1] * py + g2t.l[1][2] * pz + g2t.t[1]);
      st.cuboid[k].z = (float)(g2t.l[2][0] * px + g2t.l[2][1] * py + g2t.l[2][2] * pz + g2t.t[2]);
    }
    // pcl::getMinMax3D
    st.mn = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
             std::numeric_limits<float>::max()};
    st.mx = {-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(),
             -std::numeric_limits<float>::max()};
    for (int k = 0; k < 8; ++k) {
      st.mn.x = std::min(st.mn.x, st.cuboid[k].x); st.mx.x = std::max(st.mx.x, st.cuboid[k].x);
      st.mn.y = std::min(st.mn.y, st.cuboid[k].y); st.mx.y = std::max(st.mx.y, st.cuboid[k].y);
      st.mn.z = std::min(st.mn.z, st.cuboid[k].z); st.mx.z = std::max(st.mx.z, st.cuboid[k].z);
    }
    // Trajectory::addPoint (trajectory.cpp:64-80)
    st.pcl = {(float)st.pose[0], (float)st.pose[1], (float)st.pose[2]};
    traj.steps.push_back(st);
  }
  return true;
}


// --------------------------------------------------------------------------
// mpc_critics::ModelSharedData (include/mpc_critics/model_shared_data.h:67-116)
// --------------------------------------------------------------------------
struct SharedData {
  const float* cloud = nullptr;  // aggregate observation
  size_t n_points = 0, stride = 0;
  KdTree perception_kdtree;                 // built iff n_points >= 5 (:78-81)
  std::vector<double> prune_plan;           // nav_msgs::Path poses [m][7]
  std::vector<float> pcl_prune_plan;        // PointXYZI positions [m][3] (:83-91)
  double heading_deviation = 0.0;
  // accounting for SURVEY.md 8(d): neighbours gathered / steps evaluated by the
  // collision critic (per scoring thread, summed by the caller)
};
struct Counters { uint64_t k_sum = 0, steps_eval = 0; };

// collision_model.cpp:51-148
static double collision_model(const SharedData& sd, const Trajectory& traj, Counters& cnt,
                              std::vector<int>& id) {
  if (sd.n_points < 5) return 0.0;
  for (size_t i = 0; i < traj.steps.size(); ++i) {
    const Step& st = traj.steps[i];
    const float pose[3] = {st.pcl.x, st.pcl.y, st.pcl.z};
    F3 c{0.f, 0.f, 0.f};
    for (int k = 0; k < 8; ++k) { c.x += st.cuboid[k].x; c.y += st.cuboid[k].y; c.z += st.cuboid[k].z; }
    c.x /= 8; c.y /= 8; c.z /= 8;
    F3 dx{st.cuboid[3].x - st.cuboid[0].x, st.cuboid[3].y - st.cuboid[0].y, st.cuboid[3].z - st.cuboid[0].z};
    F3 dy{st.cuboid[1].x - st.cuboid[0].x, st.cuboid[1].y - st.cuboid[0].y, st.cuboid[1].z - st.cuboid[0].z};
    F3 dz{st.cuboid[2].x - st.cuboid[0].x, st.cuboid[2].y - st.cuboid[0].y, st.cuboid[2].z - st.cuboid[0].z};
    const double half_x = sqrtf(dx.x * dx.x + dx.y * dx.y + dx.z * dx.z) / 2.;
    const double half_y = sqrtf(dy.x * dy.x + dy.y * dy.y + dy.z * dy.z) / 2.;
    const double half_z = sqrtf(dz.x * dz.x + dz.y * dz.y + dz.z * dz.z) / 2.;
    dx.x = (float)(dx.x / (2. * half_x)); dx.y = (float)(dx.y / (2. * half_x)); dx.z = (float)(dx.z / (2. * half_x));
    dy.x = (float)(dy.x / (2. * half_y)); dy.y = (float)(dy.y / (2. * half_y)); dy.z = (float)(dy.z / (2. * half_y));
    dz.x = (float)(dz.x / (2. * half_z)); dz.y = (float)(dz.y / (2. * half_z)); dz.z = (float)(dz.z / (2. * half_z));

    sd.perception_kdtree.radius_search(pose, 1.0f, id);
    cnt.k_sum += id.size();
    cnt.steps_eval += 1;
    for (int pid : id) {
      const float* p = sd.cloud + (size_t)pid * sd.stride;
      const F3 dp{p[0] - c.x, p[1] - c.y, p[2] - c.z};
      const double x_value = fabsf(dp.x * dx.x + dp.y * dx.y + dp.z * dx.z);
      const double y_value = fabsf(dp.x * dy.x + dp.y * dy.y + dp.z * dy.z);
      const double z_value = fabsf(dp.x * dz.x + dp.y * dz.y + dp.z * dz.z);
      if (x_value <= half_x && y_value <= half_y && z_value <= half_z) return -1.0;
    }
  }
  return 0.0;
}

// collision_min_max_model.cpp:51-88
static double collision_min_max_model(const SharedData& sd, const Trajectory& traj, Counters& cnt,
                                      std::vector<int>& id) {
  if (sd.n_points < 5) return 0.0;
  for (size_t i = 0; i < traj.steps.size(); ++i) {
    const Step& st = traj.steps[i];
    const float pose[3] = {st.pcl.x, st.pcl.y, st.pcl.z};
    sd.perception_kdtree.radius_search(pose, 1.0f, id);
    cnt.k_sum += id.size();
    cnt.steps_eval += 1;
    for (int pid : id) {
      const float* p = sd.cloud + (size_t)pid * sd.stride;
      if (p[0] >= st.mn.x && p[0] <= st.mx.x && p[1] >= st.mn.y && p[1] <= st.mx.y &&
          p[2] >= st.mn.z && p[2] <= st.mx.z)
        return -1.0;
    }
  }
  return 0.0;
}

// stick_path_model.cpp:51-77 (a kd-tree is built per call there; exact 1-NN is
// what it returns, computed here by scan with FLANN's float distance).
static int plan_nearest(const SharedData& sd, const float q[3], float& d2) {
  const size_t m = sd.pcl_prune_plan.size() / 3;
  int best = -1;
  d2 = std::numeric_limits<float>::max();
  for (size_t i = 0; i < m; ++i) {
    const float v = l2_simple(&sd.pcl_prune_plan[3 * i], q);
    if (v < d2) { d2 = v; best = (int)i; }
  }
  return best;
}

static double stick_path_model(const SharedData& sd, const Trajectory& traj) {
  const size_t m = sd.pcl_prune_plan.size() / 3;
  if (m < 3) return 10.0;
  double normalized_distance = 0.0;
  for (size_t i = 0; i < traj.steps.size(); ++i) {
    const float q[3] = {traj.steps[i].pcl.x, traj.steps[i].pcl.y, traj.steps[i].pcl.z};
    float d2;
    if (plan_nearest(sd, q, d2) >= 0) normalized_distance += sqrtf(d2);
    else normalized_distance += 3.0;
  }
  normalized_distance /= (double)m;
  return normalized_distance;
}

// toward_global_plan_model.cpp:52-78
static double toward_global_plan_model(const SharedData& sd, const Trajectory& traj, double weight) {
  const size_t m = sd.pcl_prune_plan.size() / 3;
  if (m < 3) return 10.0;
  const Step& last = traj.steps.back();
  const float q[3] = {last.pcl.x, last.pcl.y, last.pcl.z};
  float d2;
  if (plan_nearest(sd, q, d2) >= 0) return sqrtf(d2) * weight;
  return -12.0;
}

// pure_pursuit_model.cpp:60-114
static double pure_pursuit_model(const SharedData& sd, const Trajectory& traj, double tw, double ow) {
  const size_t m = sd.prune_plan.size() / 7;
  if (m == 0 || traj.steps.size() < 2) return -4.0;
  const double* last_traj = traj.steps.back().pose;
  const double* last_plan = &sd.prune_plan[7 * (m - 1)];
  Affine a = transform_to_eigen(last_traj);
  a = inverse(a);
  const Affine b = transform_to_eigen(last_plan);
  const Affine d = mul(a, b);
  double tfd[7];
  eigen_to_transform(d, tfd);
  Quat q{tfd[3], tfd[4], tfd[5], tfd[6]};
  double y = tf2_yaw_from_quat(q);
  y = std::fmod((y + 3.1416), 3.1416);
  const double distance = std::sqrt(tfd[0] * tfd[0] + tfd[1] * tfd[1] + tfd[2] * tfd[2]);
  return (tw * distance + ow * y);
}

// shortest_angle_model.cpp:51-69
static double shortest_angle_model(const SharedData& sd, const Trajectory& traj, double weight_) {
  double weight;
  if (sd.heading_deviation >= 0) {
    if (traj.thetav >= 0) weight = weight_;
    else weight = weight_ * 2;
  } else {
    if (traj.thetav >= 0) weight = weight_ * 2;
    else weight = weight_;
  }
  return weight;
}

// twirling_model.cpp:51-55
static double twirling_model(const Trajectory& traj, double weight) {
  return std::fabs(traj.thetav) * weight;
}

// Diagnostic for the parity harness ("fragile" verdicts).  For
// every step (no early exit) and every cloud point near the pose, the signed
// margin mm = max(box margin, radius margin), box margin = max_i(|proj_i| -
// half_i), radius margin = dist - 1.0; mm <= 0 <=> the point makes the critic
// return -1.  Returns min mm over the trajectory: the verdict is robust unless
// |min mm| is below the harness tolerance.
static float collision_min_margin(const SharedData& sd, const Trajectory& traj, bool minmax_model,
                                  std::vector<int>& id) {
  double min_m = 1e30;
  if (sd.n_points < 5) return (float)min_m;
  for (size_t i = 0; i < traj.steps.size(); ++i) {
    const Step& st = traj.steps[i];
    const float pose[3] = {st.pcl.x, st.pcl.y, st.pcl.z};
    F3 c{0.f, 0.f, 0.f};
    for (int k = 0; k < 8; ++k) { c.x += st.cuboid[k].x; c.y += st.cuboid[k].y; c.z += st.cuboid[k].z; }
    c.x /= 8; c.y /= 8; c.z /= 8;
    F3 ax[3] = {{st.cuboid[3].x - st.cuboid[0].x, st.cuboid[3].y - st.cuboid[0].y, st.cuboid[3].z - st.cuboid[0].z},
                {st.cuboid[1].x - st.cuboid[0].x, st.cuboid[1].y - st.cuboid[0].y, st.cuboid[1].z - st.cuboid[0].z},
                {st.cuboid[2].x - st.cuboid[0].x, st.cuboid[2].y - st.cuboid[0].y, st.cuboid[2].z - st.cuboid[0].z}};
    double half[3], ux[3][3];
    for (int a = 0; a < 3; ++a) {
      half[a] = std::sqrt((double)ax[a].x * ax[a].x + (double)ax[a].y * ax[a].y + (double)ax[a].z * ax[a].z) / 2.;
      ux[a][0] = ax[a].x / (2. * half[a]); ux[a][1] = ax[a].y / (2. * half[a]); ux[a][2] = ax[a].z / (2. * half[a]);
    }
    sd.perception_kdtree.radius_search(pose, 1.01f, id);  // a little wider than the critic's ball
    for (int pid : id) {
      const float* p = sd.cloud + (size_t)pid * sd.stride;
      double m;
      if (minmax_model) {
        m = std::max({(double)st.mn.x - p[0], (double)p[0] - st.mx.x, (double)st.mn.y - p[1],
                      (double)p[1] - st.mx.y, (double)st.mn.z - p[2], (double)p[2] - st.mx.z});
      } else {
        const double dpx = (double)p[0] - c.x, dpy = (double)p[1] - c.y, dpz = (double)p[2] - c.z;
        m = -1e30;
        for (int a = 0; a < 3; ++a)
          m = std::max(m, std::fabs(dpx * ux[a][0] + dpy * ux[a][1] + dpz * ux[a][2]) - half[a]);
      }
      const double ddx = (double)p[0] - pose[0], ddy = (double)p[1] - pose[1], ddz = (double)p[2] - pose[2];
      const double mr = std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz) - 1.0;
      min_m = std::min(min_m, std::max(m, mr));
    }
  }
  return (float)min_m;
}

// stacked_scoring_model.cpp:75-93
static void score_trajectory(const dddmr_theory_config& c, const SharedData& sd, Trajectory& traj,
                             Counters& cnt, std::vector<int>& scratch) {
  for (int m = 0; m < c.n_critics; ++m) {
    const dddmr_critic_config& k = c.critics[m];
    double r = 0.0;
    switch (k.kind) {
      case DDDMR_CRITIC_COLLISION: r = collision_model(sd, traj, cnt, scratch); break;
      case DDDMR_CRITIC_COLLISION_MIN_MAX: r = collision_min_max_model(sd, traj, cnt, scratch); break;
      case DDDMR_CRITIC_STICK_PATH: r = stick_path_model(sd, traj); break;
      case DDDMR_CRITIC_PURE_PURSUIT:
        r = pure_pursuit_model(sd, traj, k.translation_weight, k.orientation_weight); break;
      case DDDMR_CRITIC_TOWARD_GLOBAL_PLAN: r = toward_global_plan_model(sd, traj, k.weight); break;
      case DDDMR_CRITIC_SHORTEST_ANGLE: r = shortest_angle_model(sd, traj, k.weight); break;
      case DDDMR_CRITIC_TWIRLING: r = twirling_model(traj, k.weight); break;
      default: r = 0.0;
    }
    if (r < 0) {
      traj.cost = r;
      break;
    } else {
      traj.cost += r;
    }
  }
}

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int oracle_velocity_iterator(double mn, double mx, int num_samples, int no_zero_insert,
                             double* out, int capacity) {
  auto v = velocity_iterator(mn, mx, num_samples, no_zero_insert != 0);
  for (int i = 0; i < (int)v.size() && i < capacity; ++i) out[i] = v[i];
  return (int)v.size();
}

int oracle_samples(const dddmr_theory_config* theory, const dddmr_tick_input* in, float* out,
                   int capacity) {
  auto s = make_samples(*theory, *in);
  for (int i = 0; i < (int)s.size() && i < capacity; ++i) {
    out[3 * i + 0] = s[i].v[0]; out[3 * i + 1] = s[i].v[1]; out[3 * i + 2] = s[i].v[2];
  }
  return (int)s.size();
}

int oracle_generate(const dddmr_theory_config* theory, const dddmr_tick_input* in,
                    const float sample[3], double* poses, float* cuboids, float* minmax,
                    int capacity) {
  Trajectory t;
  Sample s{{sample[0], sample[1], sample[2]}};
  if (!generate_trajectory(*theory, *in, s, t)) return 0;
  for (int i = 0; i < (int)t.steps.size() && i < capacity; ++i) {
    if (poses) std::memcpy(poses + 7 * i, t.steps[i].pose, 7 * sizeof(double));
    if (cuboids) std::memcpy(cuboids + 24 * i, t.steps[i].cuboid, 24 * sizeof(float));
    if (minmax) {
      std::memcpy(minmax + 6 * i, &t.steps[i].mn, 3 * sizeof(float));
      std::memcpy(minmax + 6 * i + 3, &t.steps[i].mx, 3 * sizeof(float));
    }
  }
  return (int)t.steps.size();
}

int oracle_radius_count(const float* xyz, size_t n_points, size_t stride_bytes, const float* q,
                        size_t n_queries, float radius, int32_t* counts) {
  KdTree kd;
  kd.build(xyz, n_points, stride_bytes / sizeof(float));
  std::vector<int> id;
  for (size_t i = 0; i < n_queries; ++i) counts[i] = kd.radius_search(q + 3 * i, radius, id);
  return 0;
}

// --------------------------------------------------------------------------
// PathBlockedStrategy::selfMark
// (dddmr_perception_3d/plugins/path_blocked_strategy.cpp:56-100): the share of the
// prune-plan cloud (pcl_prune_plan_, local_planner.cpp:402-430: backward points tagged
// intensity -1, forward points 0 / 1) whose FORWARD points have an observation point
// within check_radius.  A fresh kd-tree on the aggregate observation (:68-70);
// pcl::KdTreeFLANN::radiusSearch(point, double radius) hands
// static_cast<float>(radius * radius) to FLANN.  blocked / size are floats, * 100.0
// in double (:91-93).  opinion: 0 = PASS, 1 = PATH_BLOCKED_WAIT (:96-97).
// --------------------------------------------------------------------------
int oracle_path_blocked(const float* cloud, size_t n_points, size_t stride_bytes, const float* plan_xyzi,
                        size_t n_plan, double check_radius, double* ratio, int32_t* opinion,
                        uint8_t* blocked_flags) {
  *ratio = 0.0;
  *opinion = 0;
  if (blocked_flags) std::memset(blocked_flags, 0, n_plan);
  if (n_points <= 5 || n_plan == 0) return 0;                       // :62-64
  KdTree kd;
  kd.build(cloud, n_points, stride_bytes / sizeof(float));
  const float r2 = static_cast<float>(check_radius * check_radius);
  std::vector<int> id;
  size_t blocked = 0;
  for (size_t i = 0; i < n_plan; ++i) {
    const float* p = plan_xyzi + 4 * i;
    if (p[3] < 0) continue;                                         // :80-81
    if (kd.radius_search_r2(p, r2, id) > 0) {                       // :83
      ++blocked;
      if (blocked_flags) blocked_flags[i] = 1;
    }
  }
  const float orig = (float)n_plan, blk = (float)blocked;           // :91-92
  *ratio = (blk) / (orig) * 100.0;                                  // :93
  if (*ratio > 0.0) *opinion = 1;
  return 0;
}

// --------------------------------------------------------------------------
// Local-mode perception feed: MultiLayerSpinningLidar::cbSensor
// (dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp:177-281), stitcher
// off.  Third-party arithmetic restated from PCL 1.15: transformPointCloud
// (double multiply-add -> float), PassThrough (keep min <= v <= max, drop
// non-finite), VoxelGrid::applyFilter with downsample_all_data (voxel =
// floor(p * inverse_leaf) - min_b, points sorted by voxel index, centroid by
// CentroidPoint = float accumulation / n).  PCL's std::sort is unstable, so the
// summation order inside a voxel is unspecified there; here it is input order.
// Output order = ascending voxel index, like PCL.
// --------------------------------------------------------------------------
int oracle_feed(const float* scan, size_t n, size_t stride_bytes, const double T_base_sensor[7],
                const double T_gbl_base[7], double window, double height, float* out_xyz,
                size_t capacity, size_t* n_out) {
  const size_t sf = stride_bytes / sizeof(float);
  const Affine b2s = transform_to_eigen(T_base_sensor);
  const Affine g2b = transform_to_eigen(T_gbl_base);
  std::vector<F3> pts;
  pts.reserve(n);
  const float lim = (float)window, zmax = (float)height;
  for (size_t i = 0; i < n; ++i) {
    const float* p = scan + i * sf;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    F3 q;
    q.x = (float)(b2s.l[0][0] * p[0] + b2s.l[0][1] * p[1] + b2s.l[0][2] * p[2] + b2s.t[0]);
    q.y = (float)(b2s.l[1][0] * p[0] + b2s.l[1][1] * p[1] + b2s.l[1][2] * p[2] + b2s.t[1]);
    q.z = (float)(b2s.l[2][0] * p[0] + b2s.l[2][1] * p[1] + b2s.l[2][2] * p[2] + b2s.t[2]);
    if (q.x < -lim || q.x > lim) continue;   // PassThrough "x"
    if (q.y < -lim || q.y > lim) continue;   // PassThrough "y" (same limits, :246-247)
    if (q.z < 0.0f || q.z > zmax) continue;  // PassThrough "z"
    pts.push_back(q);
  }
  *n_out = 0;
  if (pts.empty()) return 0;
  const float inv = 1.0f / 0.1f;  // inverse_leaf_size_
  F3 mn = pts[0], mx = pts[0];
  for (const F3& q : pts) {
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
  std::stable_sort(order.begin(), order.end(),
                   [](const auto& a, const auto& b) { return a.first < b.first; });
  size_t out = 0;
  for (size_t i = 0; i < order.size();) {
    size_t j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    while (j < order.size() && order[j].first == order[i].first) {
      const F3& q = pts[order[j].second];
      sx += q.x; sy += q.y; sz += q.z;
      ++j;
    }
    const float cnt = (float)(j - i);
    const float cx = sx / cnt, cy = sy / cnt, cz = sz / cnt;
    if (out < capacity && out_xyz) {
      out_xyz[3 * out + 0] = (float)(g2b.l[0][0] * cx + g2b.l[0][1] * cy + g2b.l[0][2] * cz + g2b.t[0]);
      out_xyz[3 * out + 1] = (float)(g2b.l[1][0] * cx + g2b.l[1][1] * cy + g2b.l[1][2] * cz + g2b.t[1]);
      out_xyz[3 * out + 2] = (float)(g2b.l[2][0] * cx + g2b.l[2][1] * cy + g2b.l[2][2] * cz + g2b.t[2]);
    }
    ++out;
    i = j;
  }
  *n_out = out;
  return 0;
}

// One control tick: local_planner.cpp:535-587 + getBestTrajectory (:447-480).
// [begin,end) selects a contiguous range of the global sample list (sharding
// tests, bounded CPU-baseline samples); pass 0,UINT32_MAX for everything.
int oracle_tick(const dddmr_theory_config* theory, const float* cloud, size_t n_points,
                size_t stride_bytes, const double* plan, size_t n_plan,
                const dddmr_tick_input* in, uint32_t begin, uint32_t end, int n_threads,
                oracle_result* out, double* costs, int32_t* steps, float* samples_out,
                double* last_poses, float* min_margin) {
  if (!theory || !in || !out) return -1;
  std::memset(out, 0, sizeof(*out));
  const double t0 = now_s();

  // trajectory_generators: initialise() + hasMore/nextTrajectory loop (:535-557)
  const std::vector<Sample> samples = make_samples(*theory, *in);
  const uint32_t n = (uint32_t)samples.size();
  begin = std::min(begin, n);
  end = std::min(end, n);
  if (end < begin) end = begin;
  const uint32_t nl = end - begin;
  std::vector<Trajectory> trajs(nl);
  std::vector<uint8_t> generated(nl, 0);

  if (n_threads < 1) n_threads = 1;
  auto parallel_for = [&](auto&& fn) {
    if (n_threads == 1 || nl < 2) { fn(0, 0u, nl); return; }
    std::vector<std::thread> th;
    const uint32_t chunk = (nl + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
      const uint32_t b = std::min(nl, t * chunk), e = std::min(nl, b + chunk);
      if (b < e) th.emplace_back(fn, t, b, e);
    }
    for (auto& x : th) x.join();
  };

  parallel_for([&](int, uint32_t b, uint32_t e) {
    for (uint32_t i = b; i < e; ++i)
      generated[i] = generate_trajectory(*theory, *in, samples[begin + i], trajs[i]) ? 1 : 0;
  });
  const double t1 = now_s();

  // mpc_critics updateSharedData (model_shared_data.h:74-92)
  SharedData sd;
  sd.cloud = cloud;
  sd.n_points = n_points;
  sd.stride = stride_bytes / sizeof(float);
  if (n_points >= 5) sd.perception_kdtree.build(cloud, n_points, sd.stride);
  sd.prune_plan.assign(plan, plan + 7 * n_plan);
  sd.pcl_prune_plan.resize(3 * n_plan);
  for (size_t i = 0; i < n_plan; ++i) {
    sd.pcl_prune_plan[3 * i + 0] = (float)plan[7 * i + 0];
    sd.pcl_prune_plan[3 * i + 1] = (float)plan[7 * i + 1];
    sd.pcl_prune_plan[3 * i + 2] = (float)plan[7 * i + 2];
  }
  sd.heading_deviation = in->heading_deviation;
  const double t2 = now_s();

  // scoring loop (local_planner.cpp:456-469)
  std::vector<Counters> cnts(n_threads);
  parallel_for([&](int t, uint32_t b, uint32_t e) {
    std::vector<int> scratch;
    for (uint32_t i = b; i < e; ++i)
      if (generated[i]) score_trajectory(*theory, sd, trajs[i], cnts[t], scratch);
  });
  const double t3 = now_s();

  // getBestTrajectory: best.cost_ = -1; scan in generation order with <=.
  double minimum_cost = 9999999;
  int best = -1;
  uint64_t steps_total = 0;
  uint32_t n_gen = 0;
  for (uint32_t i = 0; i < nl; ++i) {
    if (!generated[i]) continue;
    ++n_gen;
    steps_total += trajs[i].steps.size();
    if (trajs[i].cost >= 0 && trajs[i].cost <= minimum_cost) {
      best = (int)i;
      minimum_cost = trajs[i].cost;
    }
  }
  out->n_samples = n;
  out->n_local = nl;
  out->n_generated = n_gen;
  out->steps_total = steps_total;
  for (auto& c : cnts) { out->k_sum += c.k_sum; out->steps_eval += c.steps_eval; }
  if (best >= 0) {
    out->planner_state = DDDMR_TRAJECTORY_FOUND;
    out->best_index = (int32_t)(begin + best);
    out->best_cost = trajs[best].cost;
    out->vx = trajs[best].xv; out->vy = trajs[best].yv; out->wz = trajs[best].thetav;
  } else {
    out->planner_state = DDDMR_ALL_TRAJECTORIES_FAIL;
    out->best_index = -1;
    out->best_cost = -1.0;
    out->vx = out->vy = out->wz = 0.0;
  }
  out->t_generate_s = t1 - t0;
  out->t_kdtree_s = t2 - t1;
  out->t_score_s = t3 - t2;

  if (min_margin) {
    int mm_kind = -1;  // first collision critic of the stack, if any
    for (int m = 0; m < theory->n_critics && mm_kind < 0; ++m)
      if (theory->critics[m].kind == DDDMR_CRITIC_COLLISION ||
          theory->critics[m].kind == DDDMR_CRITIC_COLLISION_MIN_MAX)
        mm_kind = theory->critics[m].kind;
    parallel_for([&](int, uint32_t b, uint32_t e) {
      std::vector<int> scratch;
      for (uint32_t i = b; i < e; ++i)
        min_margin[i] = (generated[i] && mm_kind >= 0)
                            ? collision_min_margin(sd, trajs[i], mm_kind == DDDMR_CRITIC_COLLISION_MIN_MAX, scratch)
                            : 1e30f;
    });
  }
  for (uint32_t i = 0; i < nl; ++i) {
    if (costs) costs[i] = generated[i] ? trajs[i].cost : DDDMR_COST_NOT_GENERATED;
    if (steps) steps[i] = generated[i] ? (int32_t)trajs[i].steps.size() : 0;
    if (samples_out) {
      samples_out[3 * i + 0] = samples[begin + i].v[0];
      samples_out[3 * i + 1] = samples[begin + i].v[1];
      samples_out[3 * i + 2] = samples[begin + i].v[2];
    }
    if (last_poses) {
      if (generated[i] && !trajs[i].steps.empty())
        std::memcpy(last_poses + 7 * i, trajs[i].steps.back().pose, 7 * sizeof(double));
      else
        std::memset(last_poses + 7 * i, 0, 7 * sizeof(double));
    }
  }
  return 0;
}

}  // extern "C"
