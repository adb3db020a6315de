// gpu_pass_through_model.h -- ScoringModel plugin of variant (i).  NOT compiled in this repository's
// containers.  Interface: mpc_critics/include/mpc_critics/scoring_model.h:44-73.
#ifndef DDDMR_ROLLOUT_ADAPTER_GPU_PASS_THROUGH_MODEL_H_
#define DDDMR_ROLLOUT_ADAPTER_GPU_PASS_THROUGH_MODEL_H_

#include <string>

#include <mpc_critics/scoring_model.h>

namespace dddmr_rollout_adapter
{

class GpuPassThroughModel : public mpc_critics::ScoringModel
{
public:
  GpuPassThroughModel() = default;
  double scoreTrajectory(base_trajectory::Trajectory & traj) override;

protected:
  void onInitialize() override;

private:
  std::string theory_name_;
};

}  // namespace dddmr_rollout_adapter
#endif
