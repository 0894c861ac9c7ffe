"""Generates tests/golden/lm_1b_topk.npz: the top-100 ids / logits, a strided slice and the moments of the last-token
logits of the ~1B random-init model (tests/lm_1b_case.py) computed by the CPU oracle oracle/lm_ref.py::LMRef, which is
itself pinned to the reference's codec_llama.py classes by tests/golden/lm_tiny.npz (SURVEY.md 8c asks for exactly this
slice at 1B).  About 0.3 TFLOP and 10 GB of host memory.
    python tests/golden/make_lm_1b_golden.py          # bf16 model -> lm_1b_topk.npz
    python tests/golden/make_lm_1b_golden.py q8_0     # its q8_0 twin (llama.cpp's quantize_row_q8_0 rule) -> lm_1b_q8_topk.npz
    python tests/golden/make_lm_1b_golden.py q4_k     # its Q4_K twin (oracle/q4k_ref.py) -> lm_1b_q4k_topk.npz
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import lm_1b_case as case  # noqa: E402


def main():
    t = time.time()
    fmt = sys.argv[1] if len(sys.argv) > 1 else None      # "q8_0": the quantised twin -> lm_1b_q8_topk.npz
    pts = case.oracle_points(fmt)
    out = {}
    for i, lg in enumerate(pts):
        for k, v in case.summarize(lg).items():
            out[f"p{i}/{k}"] = v
        print(f"point {i}: top id {int(out[f'p{i}/top_ids'][0])} {float(out[f'p{i}/top_vals'][0]):.5f}  std {float(out[f'p{i}/std']):.5f}")
    ctx, steps = case.token_ids()
    out["ctx_ids"], out["step_ids"] = ctx, np.stack(steps)
    path = os.path.join(HERE, {"q8_0": "lm_1b_q8_topk.npz", "q4_k": "lm_1b_q4k_topk.npz"}.get(fmt, "lm_1b_topk.npz"))
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes) in {time.time() - t:.0f} s")


if __name__ == "__main__":
    main()
