"""A/B of the duplex loop: the whole frame as one graph replay (rca_duplex_frame; RCA_DUPLEX_FORK=0 keeps its encode tail in line instead
of beside the first LM step) vs one replay per LM chunk between the separate codec calls.
python scripts/duplex_ab.py [secs] [weight_format]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from realtime_codec_agent_amd.duplex_bench import run_duplex_bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
fmt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
for dg, fork in ((True, "1"), (True, "0"), (False, "1"), (True, "1"), (True, "0"), (False, "1")):
    os.environ["RCA_DUPLEX_FORK"] = fork
    r = run_duplex_bench(secs=secs, weight_format=fmt, duplex_graph=dg)
    keep = {k: r[k] for k in ("xRT", "p50_frame_step_ms", "p95_frame_step_ms", "p99_frame_step_ms", "max_frame_step_ms", "frames",
                              "one_replay_frames", "stage_p50_ms", "lm_step_ms", "lm_ctx_tokens")}
    print(json.dumps({"duplex_graph": dg, "fork": fork, **keep}), flush=True)
