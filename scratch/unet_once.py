"""One eager UNet forward (768x768, CFG batch 2) + one decode after a warm-up pass: the workload of the PMC passes
(rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter set per run) behind roofline.traffic."""
import sys, torch
sys.path.insert(0, ".")
import bench
from synth_inputs import *  # noqa (placeholder, replaced below)
