import sys
sys.path.insert(0, "gl-radix-sort_amd")
import glu_hip as G
s = G.RadixSort(); s.prepare_internal_buffers(1 << 28)
