# Which leg of the default bench run leaves the rare 25-35 ms frame in the duplex leg behind?  Runs bench.py with one extra leg at a time.
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  for i in 1 2 3; do
    python bench.py "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); du=d['duplex']; print('$tag', round(du['p50_frame_step_ms'],2), round(du['p99_frame_step_ms'],2), round(du['max_frame_step_ms'],2), du['slowest_frames'][0], du['gc']['collections_in_timed_frames'])"
  done
}
run cli-only --steps 20 --no-cpu-baseline --no-trim-leg --no-bf16-leg
run steps282 --no-cpu-baseline --no-trim-leg --no-bf16-leg --no-cli-leg
run bf16+trim --steps 20 --no-cpu-baseline --no-cli-leg
