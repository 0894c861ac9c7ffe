"""CPU oracle for the duplex codec-LM hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (realtime_codec_agent_amd) never does.
"""
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "librca_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatements -> librca_oracle.so (gcc, seconds)."""
    srcs = [os.path.join(_DIR, f) for f in ("codec_oracle.c", "sampler_oracle.c", "lm_init_oracle.c")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-B", "librca_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH
