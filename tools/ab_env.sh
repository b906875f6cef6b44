#!/bin/bash
# Dev tool (GPU box): same-call A/B of environment switches on the default bench (cfg2 unless LASR_BENCH_CONFIG is in the case).
# usage: bash tools/ab_env.sh <out.txt> <steps> <repeats> "<case: VAR=val VAR=val | ->" ...     ("-" = defaults)
# Cases run round-robin <repeats> times so that slow drift of the box hits all of them alike.
out=$1; steps=$2; reps=$3; shift 3
root=${GRAFT_REPO_ROOT:-$PWD}
: > $out
for r in $(seq 1 $reps); do
  for c in "$@"; do
    envs=""; [ "$c" = "-" ] || envs="$c"
    ms=$(env $envs python $root/bench.py --steps $steps --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $r  [$c]  $ms ms/step" | tee -a $out
  done
done
python - "$out" <<'PY'
import sys, re, collections
d = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"rep \d+\s+\[(.*)\]\s+([\d.]+) ms", l)
    if m: d.setdefault(m.group(1), []).append(float(m.group(2)))
with open(sys.argv[1], "a") as f:
    for k, v in d.items():
        line = "mean [%s] %.4f ms/step (min %.4f, n=%d)" % (k, sum(v) / len(v), min(v), len(v))
        print(line); f.write(line + "\n")
PY
