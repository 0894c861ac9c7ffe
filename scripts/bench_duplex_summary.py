"""Prints the duplex fields of a bench.py output file.  usage: bench_duplex_summary.py <bench stdout file>"""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric'):
        d = json.loads(l)
        du = d["duplex"]
        print({k: du[k] for k in ("xRT", "p50_frame_step_ms", "p95_frame_step_ms", "p99_frame_step_ms", "max_frame_step_ms", "one_replay_frames", "frames")},
              [round(t["frame_ms"], 1) for t in du["trims_in_timed_window"]])
        for f in ("duplex_q8_0", "duplex_q4_k"):
            if f in d:
                print(f, d[f]["p50_frame_step_ms"], d[f]["max_frame_step_ms"])
