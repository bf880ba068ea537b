"""FeatureRegistrator on a 4096^2 cell-like image with a known similarity transform: wall time and error."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O
from microaligner_amd import FeatureRegistrator, synthetic
H = W = 4096
ref = synthetic.make_cells(H, W, seed=3)
th = np.deg2rad(0.4)
M = np.array([[np.cos(th), -np.sin(th), 23.0], [np.sin(th), np.cos(th), -15.0]])
t0 = time.time(); mov = O.warp_affine(ref, M); print('warp oracle', time.time() - t0)
f = FeatureRegistrator(); f.verbose = False
f.ref_img, f.mov_img = ref, mov
t0 = time.time(); T = f.register(); print('register (cold)', time.time() - t0, 's')
if '--profile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); T = f.register(); pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
t0 = time.time(); T = f.register(); print('register', time.time() - t0, 's')
Mi = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
print(T, '\n', Mi, '\nerr', np.abs(T - Mi).max())
