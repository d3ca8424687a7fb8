import importlib, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import helpers as Hh
from oracle import binding as ob
P = importlib.import_module("pl-svo_amd")
ctx = P.capi.Context(0)
for seed, (W, H, npts, nseg, nlev, maxl, minl) in ((1235, (640, 480, 200, 80, 4, 3, 1)), (4, (320, 240, 100, 24, 4, 3, 1)), (1236, (1280, 720, 400, 150, 5, 4, 2))):
    st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl)
    ctx.config_pyramids(2, W, H, nlev); ctx.upload_pyramid(0, ref); ctx.upload_pyramid(1, cur)
    outs = set()
    for k in range(200):
        r = ctx.sparse_align(job)
        outs.add((np.asarray(r.T).tobytes(), r.n_meas, tuple(r.iters_per_level[:5]), np.asarray(r.seg_alive).tobytes()))
    print(seed, "distinct results over 200 runs:", len(outs))
