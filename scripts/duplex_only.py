import json, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_codec_agent_amd.duplex_bench import run_duplex_bench
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    r = run_duplex_bench(torch.device('cuda', 0), secs=125.0)
    print(json.dumps({k: r[k] for k in ("p50_frame_step_ms", "p99_frame_step_ms", "max_frame_step_ms", "slowest_frames", "trims_in_timed_window", "gc")}), flush=True)
