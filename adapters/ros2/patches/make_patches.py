#!/usr/bin/env python3
"""Regenerates adapters/ros2/patches/*.patch from the reference checkout (only where /root/reference exists):
every edit below is (file, text to find, replacement); the files are copied to a scratch a/ b/ pair, edited in b/ and
diffed.  The text to find quotes a few of the reference's own lines as context, like any patch does.

usage: python adapters/ros2/patches/make_patches.py [/root/reference]"""
import os
import shutil
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
LP = "src/dddmr_local_planner/local_planner"
RB = "src/dddmr_local_planner/recovery_behaviors"
P3 = "src/dddmr_perception_3d"

TICK_CALL = '''  // ---- MI355X rollout engine (libdddmr_rollout): replaces re-initialising the theories, the
  // hasMoreTrajectories / nextTrajectory loop, updateSharedData() (kd-tree build) and getBestTrajectory()
  // of this tick by ONE batched call (dddmr_rollout_adapter/planner_bridge.h).  Shared data of generators and
  // critics is still filled above / below for the code that reads it elsewhere (isInitialHeadingAligned,
  // recovery behaviours).
  std::unique_lock<mpc_critics::StackedScoringModel::model_mutex_t> critics_lock(*(mpc_critics_ros_->getStackedScoringModelPtr()->getMutex()));
  mpc_critics_ros_->getSharedDataPtr()->robot_pose_ = trans_gbl2b_;
  mpc_critics_ros_->getSharedDataPtr()->robot_state_ = robot_state_;
  mpc_critics_ros_->getSharedDataPtr()->prune_plan_ = prune_plan_;
  if(!gpu_rollout_){
    //@ the planner owns the process's rollout context; the perception plugins borrow it (shared_context.h)
    gpu_rollout_ = dddmr_rollout_adapter::createContextFromNodes(*trajectory_generators_ros_, *mpc_critics_ros_);
    dddmr_rollout_adapter::SharedContext::publish(gpu_rollout_);
  }
  {
    dddmr_rollout_result res;
    std::string gpu_error;
    const auto outcome = dddmr_rollout_adapter::rolloutTick(
      gpu_rollout_, *(perception_3d_ros_->getSharedDataPtr()->aggregate_observation_), prune_plan_, trans_gbl2b_, robot_state_,
      perception_3d_ros_->getSharedDataPtr()->current_allowed_max_linear_speed_,
      mpc_critics_ros_->getSharedDataPtr()->heading_deviation_, traj_gen_name, best_traj, &res, &gpu_error);
    if(outcome==dddmr_rollout_adapter::TickOutcome::kPerceptionMalfunction){
      //@ the observation did not reach the device: never plan against the previous one
      RCLCPP_ERROR(this->get_logger().get_child(name_), "GPU rollout: aggregate observation rejected: %s", gpu_error.c_str());
      return dddmr_sys_core::PERCEPTION_MALFUNCTION;
    }
    if(outcome==dddmr_rollout_adapter::TickOutcome::kEngineError){
      RCLCPP_ERROR(this->get_logger().get_child(name_), "GPU rollout failed: %s", gpu_error.c_str());
      return dddmr_sys_core::ALL_TRAJECTORIES_FAIL;
    }
    if(pub_best_trajectory_pose_->get_subscription_count()>0 && res.best_index>=0){
      geometry_msgs::msg::PoseArray best_pose_arr;
      if(dddmr_rollout_adapter::bestPoses(gpu_rollout_, best_pose_arr)==DDDMR_OK){
        best_pose_arr.header.frame_id = perception_3d_ros_->getGlobalUtils()->getGblFrame();
        best_pose_arr.header.stamp = clock_->now();
        pub_best_trajectory_pose_->publish(best_pose_arr);
      }
    }
  }
'''

EDITS = {
    "local_planner_gpu_rollout.patch": [
        (LP + "/include/local_planner/local_planner.h",
         "#include <trajectory_generators/trajectory_generators_ros.h>\n",
         "#include <trajectory_generators/trajectory_generators_ros.h>\n"
         "#include <dddmr_rollout_adapter/theory_config_from_params.h>\n#include <dddmr_rollout_adapter/planner_bridge.h>\n"),
        (LP + "/include/local_planner/local_planner.h",
         "      std::shared_ptr<trajectory_generators::Trajectory_Generators_ROS> trajectory_generators_ros_;\n",
         "      std::shared_ptr<trajectory_generators::Trajectory_Generators_ROS> trajectory_generators_ros_;\n"
         "      //@ MI355X rollout engine (libdddmr_rollout), created on the first tick from the two nodes' parameters\n"
         "      dddmr_rollout_ctx* gpu_rollout_ = nullptr;\n"),
        (LP + "/src/local_planner.cpp",
         "  tf2Buffer_.reset();\n",
         "  tf2Buffer_.reset();\n"
         "  if(gpu_rollout_){\n"
         "    std::lock_guard<std::recursive_mutex> gpu_lock(dddmr_rollout_adapter::SharedContext::mutex());\n"
         "    dddmr_rollout_adapter::SharedContext::publish(nullptr);\n"
         "    dddmr_rollout_destroy(gpu_rollout_);\n"
         "  }\n"),
        (LP + "/src/local_planner.cpp", ("BLOCK", "  trajectory_generators_ros_->initializeTheories_wi_Shared_data();\n",
                                         "  getBestTrajectory(traj_gen_name, best_traj);\n"), TICK_CALL),
    ],
    "rotate_inplace_behavior_gpu_rollout.patch": [
        (RB + "/include/recovery_behaviors/robot_behavior.h",
         "#include <trajectory_generators/trajectory_generators_ros.h>\n",
         "#include <trajectory_generators/trajectory_generators_ros.h>\n#include <dddmr_rollout.h>\n"),
        (RB + "/include/recovery_behaviors/robot_behavior.h",
         "    std::shared_ptr<trajectory_generators::Trajectory_Generators_ROS> trajectory_generators_ros_;\n",
         "    std::shared_ptr<trajectory_generators::Trajectory_Generators_ROS> trajectory_generators_ros_;\n"
         "    //@ MI355X rollout engine (libdddmr_rollout): the process's shared context when the local planner published one,\n"
         "    //@ else created here on the first iteration from the two nodes' parameters\n"
         "    dddmr_rollout_ctx* gpu_rollout_ = nullptr;\n"),
        (RB + "/behaviors/rotate_inplace_behavior.cpp",
         "#include <recovery_behaviors/rotate_inplace_behavior.h>\n",
         "#include <recovery_behaviors/rotate_inplace_behavior.h>\n#include <dddmr_rollout_adapter/theory_config_from_params.h>\n"
         "#include <dddmr_rollout_adapter/planner_bridge.h>\n"),
        (RB + "/behaviors/rotate_inplace_behavior.cpp",
         ("BLOCK", "    trajectory_generators_ros_->initializeTheories_wi_Shared_data();\n", "    getBestTrajectory(trajectory_generator_name_, best_traj);\n"),
         '''    // ---- MI355X rollout engine: one batched call instead of the generate -> score -> argmin loop ----
    std::unique_lock<mpc_critics::StackedScoringModel::model_mutex_t> critics_lock(*(mpc_critics_ros_->getStackedScoringModelPtr()->getMutex()));
    base_trajectory::Trajectory best_traj;      // cost_ -1, zero velocities
    if(!gpu_rollout_)
      gpu_rollout_ = dddmr_rollout_adapter::createContextFromNodes(*trajectory_generators_ros_, *mpc_critics_ros_);
    {
      dddmr_rollout_result res;
      std::string gpu_error;
      nav_msgs::msg::Path no_plan;               // (the rotate-in-place stacks hold no path critic)
      const auto outcome = dddmr_rollout_adapter::rolloutTick(
        gpu_rollout_, *(perception_3d_ros_->getSharedDataPtr()->aggregate_observation_), no_plan, trans_gbl2b, shared_data_->robot_state_,
        -1.0, mpc_critics_ros_->getSharedDataPtr()->heading_deviation_, trajectory_generator_name_, best_traj, &res, &gpu_error);
      if(outcome==dddmr_rollout_adapter::TickOutcome::kPerceptionMalfunction || outcome==dddmr_rollout_adapter::TickOutcome::kEngineError){
        //@ best_traj stays rejected (cost_ -1): the behaviour stops the robot below, as for ALL_TRAJECTORIES_FAIL
        RCLCPP_ERROR(node_->get_logger().get_child(name_), "GPU rollout failed: %s", gpu_error.c_str());
      }
    }
'''),
    ],
    "multilayer_spinning_lidar_gpu.patch": [
        (P3 + "/include/perception_3d/multilayer_spinning_lidar.h",
         "#include <perception_3d/sensor.h>\n",
         "#include <perception_3d/sensor.h>\n#include <dddmr_rollout_adapter/perception_bridge.h>\n"),
        (P3 + "/include/perception_3d/multilayer_spinning_lidar.h",
         "    std::shared_ptr<Marking> pct_marking_;\n",
         "    std::shared_ptr<Marking> pct_marking_;\n"
         "    //@ MI355X rollout engine: the global-mode marking / clearing layer of this plugin on the device\n"
         "    //@ (store, dGraph and lethal set live there; host copies for get_dGraphValue / getLethal)\n"
         "    dddmr_rollout_adapter::MarkingLayerBridge gpu_marking_;\n"
         "    bool gpu_marking_owner_ = false;\n"
         "    pcl::PointCloud<pcl::PointXYZ>::Ptr raw_scan_;\n"),
        # cbSensor: keep the raw scan (sensor frame); local mode: the whole callback's filtering runs on the device
        (P3 + "/plugins/multilayer_spinning_lidar.cpp",
         "  get_first_tf_ = true;\n",
         '''  get_first_tf_ = true;

  //@ ---- MI355X rollout engine: hand the RAW scan (sensor frame) to the device, which runs this callback's transforms,
  //@ PassThrough crop and 0.1 m VoxelGrid itself and keeps the result as the aggregate observation (local mode) / as the
  //@ observation of the next doClear_then_Mark (global mode).  While no context is published the CPU code below runs.
  {
    std::lock_guard<std::recursive_mutex> gpu_lock(dddmr_rollout_adapter::SharedContext::mutex());
    if(dddmr_rollout_ctx* gpu = dddmr_rollout_adapter::SharedContext::get()){
      pcl::PointCloud<pcl::PointXYZ> latest;       //@ the library keeps the stitcher's deque: it gets the newest scan only
      pcl::fromROSMsg(*msg, latest);
      const int rc = dddmr_rollout_adapter::feedScan(gpu, latest, trans_b2s_, trans_gbl2b_, perception_window_size_, marking_height_, stitcher_num_);
      if(rc!=DDDMR_OK)
        RCLCPP_ERROR_THROTTLE(node_->get_logger().get_child(name_), *clock_, 5000, "GPU feed failed (%d): %s", rc, dddmr_rollout_last_error(gpu));
    }
  }
'''),
        # selfClear + selfMark, global mode: one update on the device replaces both; selfClear does it, selfMark returns
        (P3 + "/plugins/multilayer_spinning_lidar.cpp",
         "  if(shared_data_->dgraph_update_request_[name_]){\n    //@ need to regenerate dynamic graph\n    resetdGraph();\n    shared_data_->dgraph_update_request_[name_] = false;\n  }\n",
         '''  if(shared_data_->dgraph_update_request_[name_]){
    //@ need to regenerate dynamic graph
    resetdGraph();
    shared_data_->dgraph_update_request_[name_] = false;
  }

  //@ ---- MI355X rollout engine: selfClear (against the previous observation) AND selfMark (of the scan cbSensor fed)
  //@ are ONE dddmr_rollout_marking_update; selfMark() below then has nothing left to do for this cycle
  {
    std::lock_guard<std::recursive_mutex> gpu_lock(dddmr_rollout_adapter::SharedContext::mutex());
    if(gpu_marking_.ready() && dddmr_rollout_adapter::SharedContext::get()){
      dddmr_marking_stats gpu_stats;
      const int rc = gpu_marking_.clearThenMark(trans_b2s_, trans_gbl2b_, &gpu_stats);
      if(rc!=DDDMR_OK)
        RCLCPP_ERROR_THROTTLE(node_->get_logger().get_child(name_), *clock_, 5000, "GPU marking update failed (%d): %s", rc,
                              dddmr_rollout_last_error(dddmr_rollout_adapter::SharedContext::get()));
      return;
    }
  }
'''),
        (P3 + "/plugins/multilayer_spinning_lidar.cpp",
         "  if(!shared_data_->isAllLayersBeenReset()){\n    return;\n  }\n",
         "  if(!shared_data_->isAllLayersBeenReset()){\n    return;\n  }\n\n"
         "  //@ MI355X rollout engine: this cycle's selfMark already ran inside selfClear's update\n"
         "  if(gpu_marking_.ready() && dddmr_rollout_adapter::SharedContext::get()) return;\n"),
        # resetdGraph: (re)create the device layer from the ground / map clouds
        (P3 + "/plugins/multilayer_spinning_lidar.cpp",
         "  RCLCPP_INFO(node_->get_logger().get_child(name_), \"%s done dynamic graph regeneration.\", name_.c_str());\n",
         '''  {
    //@ MI355X rollout engine: the same store / dGraph on the device.  Global mode only; the context is the process's
    //@ shared one, or -- in a process without a local planner -- created and owned by the first plugin that needs it.
    std::lock_guard<std::recursive_mutex> gpu_lock(dddmr_rollout_adapter::SharedContext::mutex());
    if(!is_local_planner_){
      dddmr_rollout_ctx* gpu = dddmr_rollout_adapter::SharedContext::get();
      if(!gpu && !gpu_marking_owner_){
        gpu = dddmr_rollout_adapter::SharedContext::acquire([](){
          dddmr_rollout_config cfg{};
          cfg.abi_version = DDDMR_ROLLOUT_ABI_VERSION; cfg.world_size = 1; cfg.max_points = 1u << 18; cfg.max_trajectories = 64;
          cfg.max_steps = 64; cfg.max_plan_poses = 8;
          dddmr_rollout_ctx* c = nullptr;
          return dddmr_rollout_create(&cfg, &c)==DDDMR_OK ? c : nullptr;});
        gpu_marking_owner_ = gpu!=nullptr;
      }
      if(gpu){
        const dddmr_marking_config mc = dddmr_rollout_adapter::markingConfig(
          resolution_, height_resolution_, marking_height_, perception_window_size_, vertical_FOV_top_, vertical_FOV_bottom_,
          scan_effective_positive_start_, scan_effective_positive_end_, scan_effective_negative_start_, scan_effective_negative_end_,
          euclidean_cluster_extraction_tolerance_, euclidean_cluster_extraction_min_cluster_size_, segmentation_ignore_ratio_,
          gbl_utils_->getInscribedRadius(), gbl_utils_->getInflationRadius(), gbl_utils_->getMaxObstacleDistance(), shared_data_->static_ground_size_);
        const int rc = gpu_marking_.create(gpu, mc, *(shared_data_->pcl_ground_), shared_data_->static_ground_size_, *(shared_data_->pcl_map_));
        if(rc!=DDDMR_OK)
          RCLCPP_ERROR(node_->get_logger().get_child(name_), "GPU marking layer not created (%d): %s", rc, dddmr_rollout_last_error(gpu));
      }
    }
  }
  RCLCPP_INFO(node_->get_logger().get_child(name_), "%s done dynamic graph regeneration.", name_.c_str());
'''),
        (P3 + "/plugins/multilayer_spinning_lidar.cpp",
         "  std::unique_lock<std::recursive_mutex> lock(shared_data_->ground_kdtree_cb_mutex_);\n  return pct_marking_->get_dGraphValue(index);\n",
         "  std::unique_lock<std::recursive_mutex> lock(shared_data_->ground_kdtree_cb_mutex_);\n"
         "  if(gpu_marking_.ready()) return gpu_marking_.dGraphValue(index);      //@ host copy refreshed by every update\n"
         "  return pct_marking_->get_dGraphValue(index);\n"),
        (P3 + "/plugins/multilayer_spinning_lidar.cpp",
         "  current_lethal_.reset(new pcl::PointCloud<pcl::PointXYZI>);\n  for(auto it=pct_marking_->lethal_map_.begin(); it!=pct_marking_->lethal_map_.end(); it++){\n",
         "  current_lethal_.reset(new pcl::PointCloud<pcl::PointXYZI>);\n"
         "  if(gpu_marking_.ready()) gpu_marking_.lethalPointCloud(*(shared_data_->pcl_ground_), *current_lethal_);\n"
         "  else\n  for(auto it=pct_marking_->lethal_map_.begin(); it!=pct_marking_->lethal_map_.end(); it++){\n"),
    ],
    "path_blocked_strategy_gpu.patch": [
        (P3 + "/plugins/path_blocked_strategy.cpp",
         "#include <perception_3d/path_blocked_strategy.h>\n",
         "#include <perception_3d/path_blocked_strategy.h>\n#include <dddmr_rollout_adapter/perception_bridge.h>\n"),
        (P3 + "/plugins/path_blocked_strategy.cpp",
         "  //@this method will return percent of point of pruneplan which conflict with obstacle  \n  else{\n",
         '''  //@ ---- MI355X rollout engine: the radius probes of the prune plan against the aggregate observation the device already
  //@ holds (binned for this tick), instead of a second kd-tree build per cycle.  Any failure falls through to the CPU code.
  else if([&](){
      std::lock_guard<std::recursive_mutex> gpu_lock(dddmr_rollout_adapter::SharedContext::mutex());
      dddmr_rollout_ctx* gpu = dddmr_rollout_adapter::SharedContext::get();
      double ratio = 0.0;
      if(!gpu || dddmr_rollout_adapter::pathBlocked(gpu, shared_data_->pcl_prune_plan_, check_radius_, &ratio, nullptr)!=DDDMR_OK)
        return false;
      prune_plan_blocked_ratio_ = ratio;
      return true;}()){
  }
  //@this method will return percent of point of pruneplan which conflict with obstacle
  else{
'''),
    ],
}


def apply(text, find, repl, path):
    if isinstance(find, tuple):                      # ("BLOCK", first line, last line): replace the whole span
        _, first, last = find
        a = text.index(first)
        b = text.index(last, a) + len(last)
        return text[:a] + repl + text[b:]
    assert text.count(find) == 1, (path, find[:60], text.count(find))
    return text.replace(find, repl, 1)


def main():
    for patch, edits in EDITS.items():
        with tempfile.TemporaryDirectory() as tmp:
            files = sorted({e[0] for e in edits}, key=lambda f: [e[0] for e in edits].index(f))
            for side in ("a", "b"):
                for f in files:
                    os.makedirs(os.path.dirname(os.path.join(tmp, side, f)), exist_ok=True)
                    shutil.copy(os.path.join(REF, f), os.path.join(tmp, side, f))
            for f, find, repl in edits:
                p = os.path.join(tmp, "b", f)
                text = apply(open(p).read(), find, repl, f)
                open(p, "w").write(text)
            out = []
            for f in files:
                r = subprocess.run(["diff", "-U2", os.path.join("a", f), os.path.join("b", f)], cwd=tmp, capture_output=True, text=True)
                assert r.returncode == 1, (f, r.returncode, r.stderr)
                lines = r.stdout.split("\n")
                lines[0], lines[1] = "--- a/" + f, "+++ b/" + f          # no timestamps
                out.append("\n".join(lines))
            open(os.path.join(HERE, patch), "w").write("".join(out))
            print(patch, sum(len(o.split("\n")) for o in out), "lines")


if __name__ == "__main__":
    main()
