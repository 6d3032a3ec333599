import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stereo_orb_slam_amd import pg
ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
info = np.diag([0.01, 0.01, 0.01, 1.0, 1.0, 1.0]).reshape(36)
est = np.stack([ident, np.array([0.3, -0.1, 0.2, 0, 0, 0, 1.0])])
meas = np.array([[1.0, 0.0, 0.0, 0.0, 0.0, np.sin(0.05), np.cos(0.05)]])
with pg.PoseGraph(pg.default_options(verbose=1, max_iterations=2)) as h:
    h.set_graph(est.copy(), np.array([1, 0], np.uint8), np.array([0], np.uint32), np.array([1], np.uint32), meas, info)
    s = h.optimize()
    print("est", h.estimates())
