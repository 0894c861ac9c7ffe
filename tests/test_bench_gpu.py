"""bench.py's contract with the driver: stdout carries exactly ONE line, a JSON object with the agreed keys."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_roofline_and_legs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-duplex"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # the batch-CLI leg's own summary must not reach stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["dtype"] == "f32" and d["vs_baseline"] is None
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.3 < rf["frac"] < 1.0
    cfg = d["config"]
    assert cfg["receptive_field_trimmed"]["codes_identical_to_full_windows"] is True
    assert cfg["bf16_mfma_opt_in"]["bf16_hi_lo_split"]["code_ids_equal_to_f32_path"] > 0.99
    assert cfg["batch_cli"]["value"] > 0
