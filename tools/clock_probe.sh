#!/bin/bash
# Dev tool (GPU box): the shader clock and board power rocm-smi reports while the default bench step replays (and idle, before it).
# usage: bash tools/clock_probe.sh [steps]
root=${GRAFT_REPO_ROOT:-$PWD}
steps=${1:-4000}
echo "== idle"; rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|Power|Performance" | head -8
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
python $root/bench.py --steps $steps --warmup 10 --no-cpu-baseline > /tmp/clock_probe_bench.json 2>/dev/null &
pid=$!
sleep 12          # import + set-up + capture
echo "== while the step replays"
for i in $(seq 1 12); do
  kill -0 $pid 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power|Power \(W\)" | tr '\n' ' '; echo
  sleep 0.4
done
wait $pid
python -c "import json; d=json.loads(open('/tmp/clock_probe_bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], 'ms/step over', d['steps'], 'steps')"
