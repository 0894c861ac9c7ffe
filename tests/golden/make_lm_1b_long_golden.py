"""Generates tests/golden/lm_1b_long_topk.npz (and, with the argument q4_k, lm_1b_long_q4k_topk.npz): top-100 ids / logits, a
strided slice and the moments of the last-token logits of the ~1B random-init model behind a 2 200-token context
(tests/lm_long_case.py), computed by the CPU oracle oracle/lm_ref.py::LMRef -- itself pinned to the reference's codec_llama.py
classes by tests/golden/lm_tiny.npz.  About 4.5 TFLOP and 12 GB of host memory.
    python tests/golden/make_lm_1b_long_golden.py
    python tests/golden/make_lm_1b_long_golden.py q4_k
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import lm_1b_case as case  # noqa: E402
import lm_long_case as lc  # noqa: E402


def main():
    t = time.time()
    fmt = sys.argv[1] if len(sys.argv) > 1 else None
    pts = lc.long_1b_oracle_points(fmt)
    out = {}
    for i, lg in enumerate(pts):
        for k, v in case.summarize(lg).items():
            out[f"p{i}/{k}"] = v
        print(f"point {i}: top id {int(out[f'p{i}/top_ids'][0])} {float(out[f'p{i}/top_vals'][0]):.5f}  std {float(out[f'p{i}/std']):.5f}")
    ctx, steps = lc.long_1b_ids()
    out["ctx_ids"], out["step_ids"] = ctx, np.stack(steps)
    path = os.path.join(HERE, "lm_1b_long_q4k_topk.npz" if fmt == "q4_k" else "lm_1b_long_topk.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes) in {time.time() - t:.0f} s")


if __name__ == "__main__":
    main()
