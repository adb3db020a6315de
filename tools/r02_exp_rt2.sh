for RT in 36 40 44 46; do
  export DDDMR_RT=$RT
  python bench.py --workload C3 --steps 300 --no-cpu-baseline --no-ceiling > gpurun_out/exp_rt/C3_$RT.json 2> gpurun_out/exp_rt/C3_$RT.err || exit 1
  python - gpurun_out/exp_rt/C3_$RT.json C3 $RT <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d['roofline']
print(sys.argv[2], "rt", sys.argv[3], "ms/step", d['ms_per_step'], "k_score", r['kernel_ms'], "M/s %.1f" % (d['value']/1e6))
PY
done
export DDDMR_LIB_NAME=libdddmr_rollout_diag.so PYTHONPATH=$PWD
DDDMR_RT=44 python tools/bin_stamps.py C3 4
