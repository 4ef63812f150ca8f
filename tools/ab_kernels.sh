# Per-kernel A/B of measurement builds on one box (run through gpurun):  tools/ab_kernels.sh <precision> base v1 v2 ...
# For every variant: rocprofv3 --kernel-trace --stats of tools/perf/perf_train_leg.py (bench.py's training leg alone), then the
# average duration of every library kernel of a step and the step time -> gpurun_out/ab/kernels_<precision>.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; export TMPDIR=/tmp
precision=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=$GRAFT_REPO_ROOT/build/variants/libplayrender_$v.so; fi
  out=/tmp/ab_$v_$precision; rm -rf $out
  PR_PERF_LIB=$lib PR_PERF_PRECISION=$precision rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python tools/perf/perf_train_leg.py 15 5 > /tmp/ab_line.json 2>/dev/null
  python - "$v" "$precision" $out <<'PY' | tee -a gpurun_out/ab/kernels_$precision.log
import csv, glob, json, sys
v, precision, out = sys.argv[1:4]
line = json.loads([l for l in open('/tmp/ab_line.json') if l.startswith('{')][-1])
rows = list(csv.DictReader(open(glob.glob(out + '/**/*kernel_stats.csv', recursive=True)[0])))
steps = 20      # 15 timed + 5 warm-up steps (+ the profiled pass of perf_train_leg: 15 more) - normalise per CALL instead
keep = {}
for r in rows:
    name = r['Name'].split('(')[0].replace('pr::', '').replace('void ', '')
    if name.startswith('k_'):
        keep[name] = (float(r['AverageNs']) / 1e3, int(r['Calls']), float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3)
order = sorted(keep, key=lambda k: -keep[k][0] * keep[k][1])
print(f"{v:10s} {precision} step {line['ms_per_step']:.3f} ms (median {line['ms_per_step_median']:.3f})  " +
      "  ".join(f"{k}:{keep[k][0]:.0f}[{keep[k][2]:.0f}-{keep[k][3]:.0f}]" for k in order[:8]))
PY
done
