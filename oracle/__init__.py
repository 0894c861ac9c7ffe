"""CPU oracle for the duplex codec-LM hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (realtime_codec_agent_amd) never does.
"""
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "librca_oracle.so")


def build(force: bool = False) -> str:
    """Compile codec_oracle.c -> librca_oracle.so (gcc, seconds)."""
    src = os.path.join(_DIR, "codec_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-B", "librca_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH
