"""Shared by summarize_profile.py / collect_profiles.py: never glob(...)[0] -- a gpurun_out/prof_<tag>/ directory on the
build machine accumulates the output of every run merged back into it (one sub-directory per GPU host), so every lookup takes
the NEWEST matching file, and the sources the numbers belong to are identified by a hash written next to them."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HASHED = ["bench.py", "realtime_codec_agent_amd/csrc/rca_codec.hip", "realtime_codec_agent_amd/csrc/rca_lm.hip",
          "realtime_codec_agent_amd/csrc/rca_common.h", "include/rca.h"]


def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in HASHED:
        p = os.path.join(root, rel)
        h.update(rel.encode())
        h.update(open(p, "rb").read() if os.path.exists(p) else b"-")
    return h.hexdigest()[:16]
