"""Calibration sort time of every candidate placement of the scratch arrays (GLU_VERBOSE output of prepare):
GLU_HIP_SCRATCH_TUNE_LIST=step_mib:count python tools/tune_scan.py [u32|u64] [log2 count]"""
import os, sys
sys.path.insert(0, "gl-radix-sort_amd")
os.environ["GLU_VERBOSE"] = "1"
import glu_hip as G
kind = sys.argv[1] if len(sys.argv) > 1 else "u32"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
s = G.RadixSort(); s.prepare_internal_buffers(1 << log2n, key_bytes=8 if kind == "u64" else 4)
print(s.scratch_placement())
