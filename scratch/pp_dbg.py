import sys, numpy as np
sys.path.insert(0, "lio-mapping_amd")
from lio_amd import capi, synth
hip = capi.load_hip()
ds = synth.make_dataset("outdoor", 2, 0.3)
pp = capi.PointProcessor(hip, ds.lidar.lower_deg, ds.lidar.upper_deg, ds.lidar.rings)
for k in range(3):
    pp.process(ds.frames[k % 2].scan)
