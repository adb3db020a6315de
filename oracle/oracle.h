/*
 * oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see the
 * header of oracle.cpp).  Config / input structs are the ones declared in
 * include/dddmr_rollout.h so the checker and the product are driven by the
 * very same bytes.
 */
#ifndef DDDMR_ORACLE_H_
#define DDDMR_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#include "../include/dddmr_rollout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t planner_state;
  int32_t best_index;
  double best_cost;
  double vx, vy, wz;
  uint32_t n_samples;   /* global sample count */
  uint32_t n_local;     /* samples in [begin,end) */
  uint32_t n_generated; /* of those, generateTrajectory() == true */
  uint32_t reserved;
  uint64_t k_sum;       /* sum of radiusSearch result sizes */
  uint64_t steps_eval;  /* steps the collision critic evaluated (early exit) */
  uint64_t steps_total; /* all generated steps */
  double t_generate_s, t_kdtree_s, t_score_s;
} oracle_result;

int oracle_velocity_iterator(double mn, double mx, int num_samples, int no_zero_insert,
                             double* out, int capacity);
int oracle_samples(const dddmr_theory_config* theory, const dddmr_tick_input* in, float* out,
                   int capacity);
int oracle_generate(const dddmr_theory_config* theory, const dddmr_tick_input* in,
                    const float sample[3], double* poses, float* cuboids, float* minmax,
                    int capacity);
int oracle_radius_count(const float* xyz, size_t n_points, size_t stride_bytes, const float* q,
                        size_t n_queries, float radius, int32_t* counts);
int oracle_path_blocked(const float* cloud, size_t n_points, size_t stride_bytes, const float* plan_xyzi,
                        size_t n_plan, double check_radius, double* ratio, int32_t* opinion,
                        uint8_t* blocked_flags);
int oracle_tick(const dddmr_theory_config* theory, const float* cloud, size_t n_points,
                size_t stride_bytes, const double* plan, size_t n_plan,
                const dddmr_tick_input* in, uint32_t begin, uint32_t end, int n_threads,
                oracle_result* out, double* costs, int32_t* steps, float* samples_out,
                double* last_poses, float* min_margin);

int oracle_feed(const float* scan, size_t n, size_t stride_bytes, const double T_base_sensor[7],
                const double T_gbl_base[7], double window, double height, float* out_xyz,
                size_t capacity, size_t* n_out);

/* ---- global-mode marking / clearing layer (oracle_marking.cpp) ---- */
typedef dddmr_marking_stats oracle_marking_stats;
typedef struct oracle_marking oracle_marking;
oracle_marking* oracle_marking_create(const dddmr_marking_config* cfg, const float* ground_xyz, size_t n_ground,
                                      size_t ground_stride_bytes, const float* map_xyz, size_t n_map,
                                      size_t map_stride_bytes);
void oracle_marking_destroy(oracle_marking* m);
void oracle_marking_reset(oracle_marking* m);
int oracle_marking_update(oracle_marking* m, const float* obs_gbl_xyz, size_t n, const double T_base_sensor[7],
                          const double T_gbl_base[7], oracle_marking_stats* stats);
size_t oracle_marking_get_voxels(oracle_marking* m, int32_t* xyz_out, size_t capacity);
size_t oracle_marking_get_points(oracle_marking* m, float* xyz_out, int32_t* voxel_out, size_t capacity);
size_t oracle_marking_get_dgraph(oracle_marking* m, double* out, size_t capacity);
size_t oracle_marking_get_lethal(oracle_marking* m, uint8_t* flags, size_t capacity);
size_t oracle_marking_get_decisions(oracle_marking* m, int which, int32_t* voxels, float* margins, uint8_t* flags,
                                    size_t capacity);
int oracle_in_lidar_observation(const dddmr_marking_config* cfg, const double T_base_sensor[7],
                                const double T_gbl_base[7], const float* pts_xyz, size_t n, uint8_t* inside,
                                float* margin);

#ifdef __cplusplus
}
#endif
#endif
