/*
 * dddmr_rollout.h -- C ABI of the MI355X local-planner rollout engine.
 *
 * One call (dddmr_rollout_tick) replaces the inner section of the reference's
 * Local_Planner::computeVelocityCommand
 *   (src/dddmr_local_planner/local_planner/src/local_planner.cpp:535-587):
 *   re-initialise theories -> generate all trajectories -> update critic shared
 *   data (kd-tree build) -> score every trajectory -> pick the best one.
 *
 * All paths cited below are relative to /root/reference/src/dddmr_local_planner/
 * unless they start with dddmr_perception_3d/ or dddmr_sys_core/.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; the caller owns every buffer it passes
 *     in and it only has to stay valid for the duration of the call;
 *   - every function returns 0 (DDDMR_OK) or a negative dddmr_status;
 *   - no exceptions cross this boundary, no callbacks into the caller;
 *   - a context is externally serialised exactly like the reference serialises
 *     ticks (perception mutex local_planner.cpp:498, critics mutex :577); the
 *     library additionally takes an internal mutex per call.  set_cloud /
 *     set_scan may be called from the sensor-callback thread while another
 *     thread ticks (the device cloud is triple-buffered: published / being read by
 *     a tick / free, so a producer never waits for a tick and never overwrites
 *     what one reads; producers are serialised among themselves).
 *   - the library never runs any of this on the CPU: with no usable HIP device
 *     dddmr_rollout_create fails with DDDMR_ERR_NO_DEVICE.
 */
#ifndef DDDMR_ROLLOUT_H_
#define DDDMR_ROLLOUT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDDMR_ROLLOUT_ABI_VERSION 2
#define DDDMR_MAX_CRITICS 8
#define DDDMR_NAME_LEN 64
#define DDDMR_COMM_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */

/* library status codes (return values) */
typedef enum {
  DDDMR_OK = 0,
  DDDMR_ERR_BAD_ARG = -1,
  DDDMR_ERR_NO_DEVICE = -2,
  DDDMR_ERR_HIP = -3,
  /* stacked_generator.cpp:82-91: unknown theory name is FATAL-logged and yields
     zero trajectories; here it is an explicit error. */
  DDDMR_ERR_UNKNOWN_THEORY = -4,
  DDDMR_ERR_CAPACITY = -5,
  DDDMR_ERR_STATE = -6
} dddmr_status;

/* dddmr_sys_core/include/dddmr_sys_core/dddmr_enum_states.h:46-54 (numeric
   values are part of the boundary).  The engine itself only produces
   ALL_TRAJECTORIES_FAIL or TRAJECTORY_FOUND; the other codes stay in the host
   driver (local_planner.cpp:484-524,597-607). */
typedef enum {
  DDDMR_TF_FAIL = 0,
  DDDMR_PRUNE_PLAN_FAIL = 1,
  DDDMR_ALL_TRAJECTORIES_FAIL = 2,
  DDDMR_PERCEPTION_MALFUNCTION = 3,
  DDDMR_TRAJECTORY_FOUND = 4,
  DDDMR_PATH_BLOCKED_WAIT = 5,
  DDDMR_PATH_BLOCKED_REPLANNING = 6
} dddmr_planner_state;

/* trajectory_generators.xml: the three theory plugins. */
typedef enum {
  DDDMR_THEORY_DD_SIMPLE = 0,        /* theories/dd_simple_trajectory_generator_theory.cpp */
  DDDMR_THEORY_OMNI_SIMPLE = 1,      /* theories/omni_simple_trajectory_generator_theory.cpp */
  DDDMR_THEORY_DD_ROTATE_INPLACE = 2 /* theories/dd_rotate_inplace_theory.cpp */
} dddmr_theory_kind;

/* mpc_critics.xml: the seven critic plugins. */
typedef enum {
  DDDMR_CRITIC_COLLISION = 0,          /* models/collision_model.cpp */
  DDDMR_CRITIC_COLLISION_MIN_MAX = 1,  /* models/collision_min_max_model.cpp */
  DDDMR_CRITIC_STICK_PATH = 2,         /* models/stick_path_model.cpp */
  DDDMR_CRITIC_PURE_PURSUIT = 3,       /* models/pure_pursuit_model.cpp */
  DDDMR_CRITIC_TOWARD_GLOBAL_PLAN = 4, /* models/toward_global_plan_model.cpp */
  DDDMR_CRITIC_SHORTEST_ANGLE = 5,     /* models/shortest_angle_model.cpp */
  DDDMR_CRITIC_TWIRLING = 6            /* models/twirling_model.cpp */
} dddmr_critic_kind;

/* Per-trajectory codes reported in costs[] (stacked_scoring_model.cpp:75-93:
   the first negative critic return becomes the trajectory cost). */
#define DDDMR_COST_COLLISION (-1.0)      /* collision_model.cpp:136-139 */
#define DDDMR_COST_PURE_PURSUIT_GUARD (-4.0) /* pure_pursuit_model.cpp:62-64 */
#define DDDMR_COST_NN_FAIL (-12.0)       /* toward_global_plan_model.cpp:74 */
/* Sample whose generateTrajectory() returned false (dd_simple...cpp:364-385);
   the reference never queues such a trajectory (local_planner.cpp:551-555). */
#define DDDMR_COST_NOT_GENERATED (-100.0)

/* One critic of a theory's stack.  `weight` is the plugin's ".weight"
   parameter (read by every model; unused by collision / stick_path exactly as
   in the reference), translation/orientation weights are PurePursuitModel's
   (pure_pursuit_model.cpp:50-56). */
typedef struct {
  int32_t kind; /* dddmr_critic_kind */
  int32_t reserved;
  double weight;
  double translation_weight;
  double orientation_weight;
} dddmr_critic_config;

/* One named theory = limits + params of the generator plugin
   (dd_simple...cpp:47-134, omni_simple...cpp:47-158, dd_rotate_inplace...cpp:47-129)
   + robot cuboid + the ordered critic stack bound to it through
   "<critic>.trajectory_generator" (mpc_critics/src/mpc_critics_ros.cpp:71-79). */
typedef struct {
  char name[DDDMR_NAME_LEN];
  int32_t kind; /* dddmr_theory_kind */
  int32_t use_motor_constraint;

  /* limits */
  double min_vel_x, max_vel_x;
  double min_vel_y, max_vel_y;         /* omni only */
  double min_vel_trans, max_vel_trans; /* omni only */
  double min_vel_theta, max_vel_theta;
  double acc_lim_x, acc_lim_y, acc_lim_theta;
  double deceleration_ratio;
  double max_motor_shaft_rpm, wheel_diameter, gear_ratio, robot_radius;

  /* params */
  double controller_frequency;
  double sim_time;
  double linear_x_sample, linear_y_sample, angular_z_sample;
  double sim_granularity, angular_sim_granularity;
  double rotation_speed; /* rotate-in-place only (dd_rotate_inplace_theory.cpp:127) */

  /* 8 cuboid vertices in base_link, in the reference push order
     blb, brb, blt, flb, brt, frt, flt, frb (dd_simple...cpp:211-218); the
     collision critic derives the box axes from [0]->[3], [0]->[1], [0]->[2]. */
  float cuboid[8][3];

  /* Bench-mode extensions; 0 = reference behaviour.
     bench_fixed_steps > 0: every trajectory uses exactly that many steps and
       dt = sim_time / steps (the reference's commented-out fixed variant,
       dd_simple...cpp:391-394).
     bench_no_zero_insert != 0: VelocityIterator does not insert the extra 0.0
       sample (velocity_iterator.h:63-65) so sample counts are exact powers. */
  int32_t bench_fixed_steps;
  int32_t bench_no_zero_insert;

  int32_t n_critics;
  int32_t reserved;
  dddmr_critic_config critics[DDDMR_MAX_CRITICS];
} dddmr_theory_config;

typedef struct {
  uint32_t abi_version; /* DDDMR_ROLLOUT_ABI_VERSION */
  int32_t device;       /* HIP device ordinal */
  /* Trajectory shard of this context: the context scores the
     contiguous global sample range [rank*N/world, (rank+1)*N/world).
     world_size <= 1 means the whole batch. */
  int32_t rank;
  int32_t world_size;
  uint32_t max_points;       /* capacity of the aggregate observation cloud (< 2^20) */
  uint32_t max_trajectories; /* capacity of one tick's sample list (global N, < 2^24) */
  uint32_t max_steps;        /* capacity of one trajectory's horizon (<= 4096).  A tick whose longest
                                trajectory does not fit one workgroup's LDS (about 700 poses with the
                                collision critic; the shipped configs need <= 252) fails with
                                DDDMR_ERR_CAPACITY instead of truncating */
  uint32_t max_plan_poses;   /* capacity of the prune plan (<= 512) */
  int32_t n_theories;
  int32_t reserved;
  const dddmr_theory_config* theories;
} dddmr_rollout_config;

/* Inputs of one control tick; mirrors what computeVelocityCommand copies into
   the generator and critic shared data (local_planner.cpp:528-533, 580-583). */
typedef struct {
  double robot_pose[7];  /* trans_gbl2b_: x y z qx qy qz qw */
  double robot_twist[3]; /* robot_state_.twist.twist: linear.x linear.y angular.z */
  /* perception shared data current_allowed_max_linear_speed_ (<= 0: none,
     dd_simple...cpp:260-262, omni_simple...cpp:406-411) */
  double allowed_max_linear_speed;
  /* ModelSharedData::heading_deviation_ (local_planner.cpp:262-263,295-296) */
  double heading_deviation;
} dddmr_tick_input;

typedef struct {
  int32_t planner_state;     /* DDDMR_ALL_TRAJECTORIES_FAIL or DDDMR_TRAJECTORY_FOUND */
  int32_t best_index;        /* global sample index of the winner, -1 if none */
  double best_cost;          /* -1.0 if none (local_planner.cpp:450) */
  double vx, vy, wz;         /* best_traj.{xv_,yv_,thetav_}; 0 if none (trajectory.cpp:34-37) */
  uint32_t n_samples;        /* global number of velocity samples this tick */
  uint32_t n_local;          /* samples scored by this context's shard */
  uint32_t local_begin;      /* first global sample index of the shard */
  uint32_t n_points_binned;  /* cloud points inside the local costmap tile */
  /* packed argmin key of this shard, see dddmr_rollout_pack_key */
  int64_t key;
  float device_ms;           /* HIP-event time of the tick's kernels */
  float score_ms;            /* HIP-event time of the fused rollout+critics kernel alone */
} dddmr_rollout_result;

/* Optional per-trajectory outputs of the last tick (caller-allocated). */
typedef struct {
  double* costs;      /* [n_local] accumulated cost or reject code */
  int32_t* steps;     /* [n_local] generated steps (0 = not generated) */
  float* samples;     /* [n_local][3] vx vy wz of each sample */
} dddmr_rollout_debug;

typedef struct dddmr_rollout_ctx dddmr_rollout_ctx;

/* Replaces Trajectory_Generators_ROS / MPC_Critics_ROS plugin loading
   (trajectory_generators/src/trajectory_generators_ros.cpp:45-84,
    mpc_critics/src/mpc_critics_ros.cpp:45-83). */
int dddmr_rollout_create(const dddmr_rollout_config* cfg, dddmr_rollout_ctx** out);
void dddmr_rollout_destroy(dddmr_rollout_ctx* ctx);

/* Aggregate observation in the global frame = output of
   StackedPerception::aggregateObservations
   (dddmr_perception_3d/src/stacked_perception.cpp:128-140).  xyzi points at
   n_points records, stride_bytes apart, each starting with float x,y,z
   (PCL PointXYZI: stride 32; packed xyzi: 16; packed xyz: 12). */
int dddmr_rollout_set_cloud(dddmr_rollout_ctx* ctx, const float* xyzi,
                            size_t n_points, size_t stride_bytes);

/* Fused local-mode perception feed = MultiLayerSpinningLidar::cbSensor
   (dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp:177-281):
   sensor->base transform, PassThrough crop |x|,|y| <= window, 0 <= z <= height,
   0.1 m VoxelGrid centroid downsample, base->global transform; the result
   becomes the aggregate observation without leaving the device.
   T_* are x y z qx qy qz qw. */
int dddmr_rollout_set_scan(dddmr_rollout_ctx* ctx, const float* xyz, size_t n_points,
                           size_t stride_bytes, const double T_base_sensor[7],
                           const double T_gbl_base[7], double perception_window_size,
                           double marking_height, uint32_t* n_out_points);

/* cbSensor's stitcher (multilayer_spinning_lidar.cpp:185-200, parameter `stitcher_num`): with
   stitcher_num > 0 every dddmr_rollout_set_scan feeds the last stitcher_num RAW scans, oldest first,
   through the CURRENT transforms (exactly what the reference does: the queued scans are not
   re-registered).  0 switches it off and empties the queue.  The queued scans together must fit
   max_points. */
int dddmr_rollout_set_stitcher(dddmr_rollout_ctx* ctx, int32_t stitcher_num);

/* Several sensors on one aggregate: StackedPerception::aggregateObservations
   (dddmr_perception_3d/src/stacked_perception.cpp:128-140) concatenates every sensor plugin's
   current observation, in plugin order.  source_id = the sensor's position in that order
   (0 .. DDDMR_MAX_SOURCES - 1); each call runs that sensor's cbSensor on the device and the aggregate
   becomes the concatenation, in source order, of every source's LATEST observation (a source that has
   not reported yet contributes nothing).  *n_source_points / *n_aggregate_points (either may be NULL)
   receive the two sizes.  The first call with a source id switches the context to this mode; plain
   dddmr_rollout_set_scan then means source 0.  The observations together must fit max_points. */
#define DDDMR_MAX_SOURCES 4
int dddmr_rollout_set_scan_source(dddmr_rollout_ctx* ctx, int32_t source_id, const float* xyz,
                                  size_t n_points, size_t stride_bytes, const double T_base_sensor[7],
                                  const double T_gbl_base[7], double perception_window_size,
                                  double marking_height, uint32_t* n_source_points,
                                  uint32_t* n_aggregate_points);
int dddmr_rollout_set_stitcher_source(dddmr_rollout_ctx* ctx, int32_t source_id, int32_t stitcher_num);

/* Copy the current aggregate observation back (debug / parity of set_scan). */
int dddmr_rollout_get_cloud(dddmr_rollout_ctx* ctx, float* xyzi_out, size_t capacity,
                            size_t* n_points);

/* Prune plan = output of Local_Planner::prunePlan (local_planner.cpp:374-445);
   poses are x y z qx qy qz qw in the global frame. */
int dddmr_rollout_set_prune_plan(dddmr_rollout_ctx* ctx, const double* poses_xyz_qxyzw,
                                 size_t n_poses);

/* Perception opinions (dddmr_perception_3d/include/perception_3d/sensor.h, PerceptionOpinion). */
typedef enum {
  DDDMR_OPINION_PASS = 0,
  DDDMR_OPINION_PATH_BLOCKED_WAIT = 1
} dddmr_perception_opinion;

/* PathBlockedStrategy::selfMark
   (dddmr_perception_3d/plugins/path_blocked_strategy.cpp:56-100) on the current aggregate
   observation: plan_xyzi is pcl_prune_plan_ as Local_Planner::prunePlan fills it
   (local_planner.cpp:402-430: n_points records x,y,z,intensity, backward points tagged
   intensity < 0).  *blocked_ratio_percent = blocked forward points / n_points * 100,
   *opinion = PATH_BLOCKED_WAIT when that is > 0 (computeVelocityCommand then returns
   dddmr_sys_core::PATH_BLOCKED_WAIT, local_planner.cpp:597-602).  blocked_flags (may be
   NULL) receives one byte per plan point.  Replaces the second per-tick kd-tree build on
   the observation (:68-70). */
int dddmr_rollout_path_blocked(dddmr_rollout_ctx* ctx, const float* plan_xyzi, size_t n_points,
                               double check_radius, double* blocked_ratio_percent, int32_t* opinion,
                               uint8_t* blocked_flags);

/* The theory's initialise() alone (dd_simple...cpp:236-295, omni_simple...cpp:260-332,
   dd_rotate_inplace_theory.cpp:229-274): the velocity samples a tick with these inputs rolls out, in
   generation order, samples_out[n][3] = vx vy wz.  Host-only (no device work); call with NULL for the
   count.  An iterator-protocol adapter (hasMoreTrajectories / nextTrajectory) needs the list before the
   batch has been scored. */
int dddmr_rollout_samples(dddmr_rollout_ctx* ctx, const char* theory_name, const dddmr_tick_input* in,
                          float* samples_out, size_t capacity, size_t* n_samples);

/* One control tick for the named theory. */
int dddmr_rollout_tick(dddmr_rollout_ctx* ctx, const char* theory_name,
                       const dddmr_tick_input* in, dddmr_rollout_result* out);

/* Split form of dddmr_rollout_tick for hosts that want to overlap their own work
   (e.g. the previous tick's all-reduce) with the GPU: tick_begin enqueues the
   tick and returns at once, tick_end waits for it and fills the result.  Exactly
   one tick may be pending per context; set_prune_plan / tick / get_* return
   DDDMR_ERR_STATE while one is.  set_cloud / set_scan stay allowed, any number of
   times, from any thread (they fill a free buffer; the pending tick keeps the
   observation it started with).  dddmr_rollout_tick == tick_begin + tick_end. */
int dddmr_rollout_tick_begin(dddmr_rollout_ctx* ctx, const char* theory_name,
                             const dddmr_tick_input* in);
int dddmr_rollout_tick_end(dddmr_rollout_ctx* ctx, dddmr_rollout_result* out);

/* Multi-rank hosts: every rank ticks its shard, then ONE small min all-reduce
   picks the global winner and every rank resolves its command from the index.

   Exact form (use this one): rank r contributes two int64 words, dddmr_rollout_winner_words
   = { bit pattern of its best cost, -best_index } (INT64_MAX, INT64_MAX when its shard has no
   acceptable trajectory).  All-reduce with MIN a vector of 2*n_ranks words in which rank r fills
   slots [2r, 2r+1] and leaves INT64_MAX elsewhere (16*n_ranks bytes, latency-bound), then call
   dddmr_rollout_resolve_words on every rank: minimum cost compared as full doubles, equal costs
   -> highest index, exactly the reference's `<=` scan over the whole batch
   (local_planner.cpp:456-463).

   8-byte form: min-reduce result.key (dddmr_rollout_pack_key) and call dddmr_rollout_resolve.
   The key carries the top 40 bits of the cost, so across ranks costs closer than 3.7e-9
   relative resolve to the higher index; inside one shard the winner is always exact. */
int dddmr_rollout_resolve(dddmr_rollout_ctx* ctx, int64_t reduced_key,
                          dddmr_rollout_result* inout);
void dddmr_rollout_winner_words(const dddmr_rollout_result* r, int64_t words[2]);
int dddmr_rollout_resolve_words(dddmr_rollout_ctx* ctx, const int64_t* words, int32_t n_ranks,
                                dddmr_rollout_result* inout);

/* In-library exchange (SURVEY.md 8b "Context owns ... RCCL communicators", 8e): a C++ host needs no
   collective code of its own.  One rank calls dddmr_rollout_comm_unique_id (ncclGetUniqueId) and
   hands the 128 bytes to the others by any means it has; every rank then calls
   dddmr_rollout_comm_init with the rank / n_ranks its context was created with (collective:
   ncclCommInitRank, one process per GPU, RCCL over xGMI).  From then on every tick of the context
   runs, on the context's stream, k_score -> ONE ncclAllReduce(ncclInt64, ncclMin) of the 2*n_ranks
   slot vector described above -> a resolve kernel, and dddmr_rollout_tick / tick_end return the
   GLOBAL winner (best_index, exact best_cost, command) on every rank; `key` is the global winner's
   packed key, n_local / local_begin still describe the shard.  All ranks must tick in lockstep (same
   theory, same inputs): the all-reduce is a collective.  librccl is loaded at run time
   (DDDMR_RCCL_LIB overrides the search), so hosts that never call this need no RCCL.
   dddmr_rollout_comm_destroy returns the context to single-rank results (the host-side
   resolve_words path keeps working either way). */
int dddmr_rollout_comm_unique_id(uint8_t id_out[DDDMR_COMM_ID_BYTES]);
int dddmr_rollout_comm_init(dddmr_rollout_ctx* ctx, const uint8_t id[DDDMR_COMM_ID_BYTES],
                            int32_t rank, int32_t n_ranks);
int dddmr_rollout_comm_destroy(dddmr_rollout_ctx* ctx);
/* HIP devices this process sees (hipGetDeviceCount): what a multi-GPU host places its contexts by
   (`cfg.device`); the reference has no counterpart (one CPU process per robot). */
int dddmr_rollout_device_count(int32_t* n_out);
/* Ranks of the context's exchange as the communicator itself reports them (ncclCommCount), 0 without
   one: what a multi-GPU run prints next to its numbers. */
int dddmr_rollout_comm_ranks(dddmr_rollout_ctx* ctx, int32_t* n_ranks_out);
/* Single-device rehearsal of the exchange (RCCL refuses two ranks on one GPU): the context, created
   as rank r of W, runs the same k_score -> slot vector -> resolve-kernel sequence on its stream with
   the all-reduce replaced by a device copy of its own send vector; the peers' (cost bits, -index)
   pairs (dddmr_rollout_winner_words of their ticks) are written into it by the host.  Exercises
   exactly the device code a W-rank communicator runs, including ranks whose shard is empty. */
int dddmr_rollout_comm_loopback(dddmr_rollout_ctx* ctx);
int dddmr_rollout_comm_loopback_set_peer(dddmr_rollout_ctx* ctx, int32_t peer_rank, const int64_t words[2]);

/* Per-trajectory outputs of the last tick (any pointer may be NULL). */
int dddmr_rollout_get_debug(dddmr_rollout_ctx* ctx, dddmr_rollout_debug* dbg);

/* Debug pose arrays of the last tick, poses_out[n][7] x y z qx qy qz qw, trajectory by
   trajectory in sample order, poses in step order: which = 0 the `trajectory` topic
   (every generated trajectory, local_planner.cpp:549-569), which = 1 the
   `accepted_trajectory` topic (cost_ >= 0, local_planner.cpp:461-470).  Call with
   poses_out == NULL to get the count.  Computed on demand from the rollout state the
   tick left on the device; nothing is copied unless this is called. */
int dddmr_rollout_get_pose_arrays(dddmr_rollout_ctx* ctx, int32_t which, double* poses_out,
                                  size_t capacity, size_t* n_poses);

/* Best trajectory poses of the last tick for visualisation
   (local_planner.cpp:472-478): poses_out[n][7] x y z qx qy qz qw. */
int dddmr_rollout_get_best_poses(dddmr_rollout_ctx* ctx, double* poses_out,
                                 size_t capacity, size_t* n_poses);

/* Cuboids of the best trajectory, vertices_out[n][8][3] floats in the vertex order of the theory's
   cuboid: base_trajectory::Trajectory::getCuboid(i) of every pose (trajectory.cpp:52-54, filled at
   dd_simple...cpp:443).  The reference collects them for a `trajectory_cuboids` debug cloud whose
   publication is commented out (local_planner.cpp:118,454,572-573,631); its `robot_cuboid` topic is a
   static marker built from the YAML vertices (:159-190,364-367) and needs no compute. */
int dddmr_rollout_get_best_cuboids(dddmr_rollout_ctx* ctx, float* vertices_out, size_t capacity_poses,
                                   size_t* n_poses);

/* Argmin key: min over keys == minimum cost, ties -> highest index (the
   reference's `<=` scan keeps the LAST minimal trajectory,
   local_planner.cpp:460-463).  cost < 0 (rejected) or cost > 9999999 (the scan's
   initial minimum_cost, :452: never accepted) -> INT64_MAX. */
int64_t dddmr_rollout_pack_key(double cost, uint32_t global_index);
int32_t dddmr_rollout_key_index(int64_t key); /* -1 for the "none" key */

/* ---------------------------------------------------------------------------------------------
   Global-mode marking / clearing layer: MultiLayerSpinningLidar::selfClear /
   selfMark with is_local_planner = false
   (dddmr_perception_3d/plugins/multilayer_spinning_lidar.cpp:306-628, isinLidarObservation :682-746,
   getCastingPointCloud :630-651) and the cluster store Marking::addPCPtr / removePCPtr /
   computeMinDistanceFromObstacle2GroundNodes (plugins/cluster_marking.cpp:49-138) with its
   DynamicGraph (src/graph/dynamic_graph.cpp).  The persistent voxel -> cluster store, the dGraph
   (per-ground-node obstacle distance) and the lethal set live on the device.

   One dddmr_rollout_marking_update() = one StackedPerception::doClear_then_Mark() pass of the lidar
   plugin (src/stacked_perception.cpp:72-90): selfClear against the PREVIOUS update's observation
   and the current sensor pose, then selfMark of the current observation.  The observation is the
   context's current aggregate cloud in the global frame -- what dddmr_rollout_set_scan leaves on
   the device (pcl_msg_gbl_, :322-323) or dddmr_rollout_set_cloud uploaded.
   Parameters carry the plugin's YAML names (multilayer_spinning_lidar.cpp:73-139) and the node's
   inscribed_radius / inflation_radius / max_obstacle_distance. */
typedef struct {
  double xy_resolution, height_resolution;
  double marking_height, perception_window_size;
  double vertical_FOV_top, vertical_FOV_bottom;                 /* degrees */
  double scan_effective_positive_start, scan_effective_positive_end;
  double scan_effective_negative_start, scan_effective_negative_end;
  double euclidean_cluster_extraction_tolerance;
  int32_t euclidean_cluster_extraction_min_cluster_size;
  int32_t reserved;
  double segmentation_ignore_ratio;
  double inscribed_radius, inflation_radius, max_obstacle_distance;
  uint32_t max_markings;        /* slots of the persistent store; size it for about twice the markings alive at a time:
                                   cleared voxels keep their slot until the store's garbage collection drops them
                                   (it runs when half the slots hold a key) */
  uint32_t max_cluster_points;  /* capacity of the pool of stored cluster points (0.2 m downsampled) */
} dddmr_marking_config;

typedef struct {
  uint32_t n_observation;   /* points of the observation this update marked */
  uint32_t n_clusters;      /* Euclidean clusters found in it */
  uint32_t n_marked;        /* clusters stored by Marking::addPCPtr */
  uint32_t n_in_window;     /* stored markings selfClear looked at */
  uint32_t n_cleared;       /* ... of which removePCPtr'ed */
  uint32_t n_alive;         /* markings alive after the update */
  float clear_ms, mark_ms;  /* HIP-event times of the two halves */
} dddmr_marking_stats;

/* ground = shared_data_->pcl_ground_ (the nodes of the dGraph, kdtree_ground_), map =
   shared_data_->pcl_map_ (kdtree_map_, the static layer's cloud); both x y z records `stride`
   bytes apart, copied to the device once.  Also initialises the dGraph (resetdGraph, :831-839). */
int dddmr_rollout_marking_create(dddmr_rollout_ctx* ctx, const dddmr_marking_config* cfg,
                                 const float* ground_xyz, size_t n_ground, size_t ground_stride_bytes,
                                 const float* map_xyz, size_t n_map, size_t map_stride_bytes);
/* One doClear_then_Mark pass.  May be called between dddmr_rollout_tick_begin and _tick_end: the
   reference runs the perception thread's pass and the planner's tick side by side, and so does the
   device (the update takes a stream of its own next to the tick's kernels) -- provided no newer
   observation was published after tick_begin, else DDDMR_ERR_STATE (call it after tick_end). */
int dddmr_rollout_marking_update(dddmr_rollout_ctx* ctx, const double T_base_sensor[7],
                                 const double T_gbl_base[7], dddmr_marking_stats* stats);
/* resetdGraph: empty store, dGraph back to max_obstacle_distance. */
int dddmr_rollout_marking_reset(dddmr_rollout_ctx* ctx);
/* The generator points of every alive marking -- its cluster projected on the robot's ground plane and
   downsampled at 0.1 m, what computeMinDistanceFromObstacle2GroundNodes (cluster_marking.cpp:54-64)
   searches the ground nodes with -- as x y z floats, with the voxel key (x y z ints) of the marking each
   belongs to in voxel_out (may be NULL); call with xyz_out NULL for the count.  Debug / visualisation
   (the reference publishes its markings as a cloud). */
int dddmr_rollout_marking_get_points(dddmr_rollout_ctx* ctx, float* xyz_out, int32_t* voxel_out, size_t capacity,
                                     size_t* n);
/* Alive markings as voxel keys (x y z ints, xyz_out[n][3]); call with NULL for the count. */
int dddmr_rollout_marking_get_voxels(dddmr_rollout_ctx* ctx, int32_t* xyz_out, size_t capacity, size_t* n);
/* dGraph values of ground nodes 0..n_ground (DynamicGraph::initial fills n + 1 entries) and the
   lethal set (lethal_map_ keys) as one byte per ground node. */
int dddmr_rollout_marking_get_dgraph(dddmr_rollout_ctx* ctx, double* values_out, size_t capacity);
int dddmr_rollout_marking_get_lethal(dddmr_rollout_ctx* ctx, uint8_t* flags_out, size_t capacity);
/* Which route the updates took (observations of up to 16384 points run fused: six launches, no
   copies; larger ones take the general route with library sorts; DDDMR_MARKING_ROUTE=general|fused
   forces one) and how many kernels / memsets the last update launched.  Diagnostics, any pointer
   may be NULL. */
int dddmr_rollout_marking_route_counts(dddmr_rollout_ctx* ctx, uint32_t* updates_fused,
                                       uint32_t* updates_general, uint32_t* launches_last_update);

/* Measurement aid ("a measured stream-copy ceiling on the same GPU"): streams
   `bytes` (>= 1 GiB recommended: beyond the 256 MB of MALL) `reps` times through a float4 copy
   kernel and a read-only kernel on the context's device; *copy_gbps counts read + write bytes.
   Not part of the reference's surface and not on the tick's path. */
int dddmr_rollout_stream_ceiling(dddmr_rollout_ctx* ctx, size_t bytes, int32_t reps,
                                 double* copy_gbps, double* read_gbps);

/* Self-test aid: sine and cosine of n heading angles from the rollout's own double-precision routine
   (the stand-in for the libm sin / cos the reference's theories call, dd_simple...cpp:416,457-464,
   omni_simple...cpp:498-505), so that a test can bound its error against a higher-precision value.
   Not part of the reference's surface and not on the tick's path. */
int dddmr_rollout_selftest_sincos(dddmr_rollout_ctx* ctx, const double* angles, size_t n,
                                  double* sin_out, double* cos_out);

const char* dddmr_rollout_last_error(dddmr_rollout_ctx* ctx);
const char* dddmr_rollout_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DDDMR_ROLLOUT_H_ */
