// shared_context.h -- who owns the rollout context when planner and perception plugins both use it.
//
// One process of the reference hosts one Local_Planner node and its perception_3d plugins
// (dddmr_p2p_move_base: p2p_move_base_node.cpp creates both in the same executable), and all of them
// look at the SAME aggregate observation each control cycle.  So there is one context per process:
//
//   owner      Local_Planner (patches/local_planner_gpu_rollout.patch) creates it on its first tick from the
//              generator / critic nodes' parameters, publishes it here, and destroys it in its destructor
//              (after withdrawing it).  A process without a Local_Planner (the global planner's
//              perception_3d_ros node) lets the first plugin that needs one create and own it through
//              acquire(): the last release() destroys it.
//   borrowers  MultiLayerSpinningLidar (cbSensor feed, global-mode marking / clearing) and
//              PathBlockedStrategy get() it under mutex(); while none is published they run the
//              reference's CPU code of that cycle unchanged.
//
// The library serialises calls on a context itself (include/dddmr_rollout.h, "Threading"); mutex() only
// guards the pointer's lifetime against the owner's destructor.
#ifndef DDDMR_ROLLOUT_ADAPTER_SHARED_CONTEXT_H_
#define DDDMR_ROLLOUT_ADAPTER_SHARED_CONTEXT_H_

#include <atomic>
#include <mutex>

#include "dddmr_rollout.h"

namespace dddmr_rollout_adapter
{

class SharedContext
{
public:
  static std::recursive_mutex & mutex() {return state().mu;}

  // borrowed pointer, nullptr while no owner has published one (hold mutex() while using it)
  static dddmr_rollout_ctx * get() {return state().ctx;}

  // owner: publish after dddmr_rollout_create, withdraw (nullptr) before dddmr_rollout_destroy
  static void publish(dddmr_rollout_ctx * ctx)
  {
    std::lock_guard<std::recursive_mutex> lock(state().mu);
    state().ctx = ctx;
    state().device_feed_seq = 0;
  }

  // processes without a planner: create on first use, destroy with the last user
  template<class MakeContext>
  static dddmr_rollout_ctx * acquire(MakeContext && make)
  {
    std::lock_guard<std::recursive_mutex> lock(state().mu);
    if (!state().ctx) {
      state().ctx = make();
      state().owned_here = state().ctx != nullptr;
    }
    if (state().ctx && state().owned_here) {++state().users;}
    return state().ctx;
  }
  static void release()
  {
    std::lock_guard<std::recursive_mutex> lock(state().mu);
    if (state().owned_here && state().users > 0 && --state().users == 0) {
      dddmr_rollout_destroy(state().ctx);
      state().ctx = nullptr;
      state().owned_here = false;
    }
  }

  // The lidar plugin's cbSensor feed (dddmr_rollout_set_scan) leaves the aggregate observation on the
  // device: the planner must not overwrite it with the CPU aggregate of the same cycle.  The feed bumps a
  // sequence number; the planner's tick consumes it.
  static void noteDeviceFeed() {++state().device_feed_seq;}
  static bool consumeDeviceFeed()
  {
    const unsigned now = state().device_feed_seq.load();
    const bool fresh = now != state().device_feed_seen;
    state().device_feed_seen = now;
    return fresh;
  }

private:
  struct State
  {
    std::recursive_mutex mu;
    dddmr_rollout_ctx * ctx = nullptr;
    bool owned_here = false;
    int users = 0;
    std::atomic<unsigned> device_feed_seq{0};
    unsigned device_feed_seen = 0;
  };
  static State & state()
  {
    static State s;
    return s;
  }
};

}  // namespace dddmr_rollout_adapter
#endif
