// CosineAnnealingWarmupRestarts stepped ON THE DEVICE (scheduler/cosine_annearing_with_warmup.py:53-89, stepped once per
// batch: train.py:57-61): the optimiser reads its learning rate from a device scalar, and the schedule's own state
// (cycle, step inside the cycle, current cycle length, decayed max_lr) lives next to it.  One thread advances the state and
// writes the next learning rate, in f64 like the host class - so a training step has no host-computed scalar in it and the
// whole step can be replayed from a captured hipGraph.
#include "common.h"
#include <math.h>

namespace lasr {

// mirrors the fields of the reference class (and of lightning_asr_amd/schedule.py)
struct LrState {
  double base_max_lr, max_lr, min_lr, cycle_mult, gamma;
  long long first_cycle_steps, cur_cycle_steps, warmup_steps, cycle, step_in_cycle, last_epoch;
};

__global__ void lr_schedule_step_kernel(LrState* st, float* lr_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  LrState s = *st;
  s.last_epoch += 1;
  s.step_in_cycle += 1;
  if (s.step_in_cycle >= s.cur_cycle_steps) {
    s.cycle += 1;
    s.step_in_cycle -= s.cur_cycle_steps;
    s.cur_cycle_steps = (long long)((double)(s.cur_cycle_steps - s.warmup_steps) * s.cycle_mult) + s.warmup_steps;   // int(...) truncates
  }
  s.max_lr = s.base_max_lr * pow(s.gamma, (double)s.cycle);
  double lr;
  if (s.step_in_cycle == -1) lr = s.min_lr;
  else if (s.step_in_cycle < s.warmup_steps) lr = (s.max_lr - s.min_lr) * (double)s.step_in_cycle / (double)s.warmup_steps + s.min_lr;
  else
    lr = s.min_lr + (s.max_lr - s.min_lr) *
                        (1.0 + cos(3.141592653589793 * (double)(s.step_in_cycle - s.warmup_steps) / (double)(s.cur_cycle_steps - s.warmup_steps))) / 2.0;
  *st = s;
  *lr_out = (float)lr;
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_lr_schedule_state_bytes(void) { return sizeof(LrState); }

// state_host_out: sizeof(LrState) bytes the caller uploads to the device (any copy it likes), describing the schedule right
// after construction (the reference's constructor steps once and then resets the rate to min_lr).
extern "C" int lasr_lr_schedule_init(void* state_host_out, size_t bytes, int64_t first_cycle_steps, double cycle_mult, double max_lr,
                                     double min_lr, int64_t warmup_steps, double gamma, int64_t cycle, int64_t step_in_cycle,
                                     int64_t cur_cycle_steps, int64_t last_epoch) {
  LASR_CHECK_ARG(state_host_out && bytes >= sizeof(LrState), "lasr_lr_schedule_init: buffer of %zu bytes needed", sizeof(LrState));
  LASR_CHECK_ARG(first_cycle_steps > 0 && warmup_steps >= 0 && warmup_steps < first_cycle_steps, "lasr_lr_schedule_init: warmup_steps < first_cycle_steps");
  LrState s;
  s.base_max_lr = max_lr; s.max_lr = max_lr * pow(gamma, (double)cycle); s.min_lr = min_lr; s.cycle_mult = cycle_mult; s.gamma = gamma;
  s.first_cycle_steps = first_cycle_steps; s.cur_cycle_steps = cur_cycle_steps; s.warmup_steps = warmup_steps; s.cycle = cycle;
  s.step_in_cycle = step_in_cycle; s.last_epoch = last_epoch;
  memcpy(state_host_out, &s, sizeof(s));
  return 0;
}

// advance the schedule by one step and write the new rate: lr_dev[0] is what the NEXT lasr_novograd_step reads
extern "C" int lasr_lr_schedule_step(void* state_dev, float* lr_dev, void* stream) {
  LASR_CHECK_ARG(state_dev && lr_dev, "lasr_lr_schedule_step: null pointer");
  hipLaunchKernelGGL(lr_schedule_step_kernel, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<LrState*>(state_dev), lr_dev);
  LASR_LAUNCH_CHECK("lr_schedule_step_kernel");
  return 0;
}
