// gpu_rollout_theory.h -- TrajectoryGeneratorTheory plugin of variant (i).  NOT compiled in this
// repository's containers.  Interface: trajectory_generators/include/trajectory_generators/
// trajectory_generator_theory.h:49-73; exported like dd_simple_trajectory_generator_theory.cpp:33.
#ifndef DDDMR_ROLLOUT_ADAPTER_GPU_ROLLOUT_THEORY_H_
#define DDDMR_ROLLOUT_ADAPTER_GPU_ROLLOUT_THEORY_H_

#include <string>
#include <vector>

#include <trajectory_generators/trajectory_generator_theory.h>

#include "dddmr_rollout.h"

namespace dddmr_rollout_adapter
{

class GpuRolloutTheory : public trajectory_generators::TrajectoryGeneratorTheory
{
public:
  GpuRolloutTheory() = default;

  bool hasMoreTrajectories() override;
  bool nextTrajectory(base_trajectory::Trajectory & traj) override;
  void initialise() override;

protected:
  void onInitialize() override;

private:
  dddmr_theory_config config_{};
  const std::vector<float> * samples_ = nullptr;   // owned by the bridge, valid until the next initialise()
  size_t next_ = 0;
};

}  // namespace dddmr_rollout_adapter
#endif
