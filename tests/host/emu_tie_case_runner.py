"""One config-2 (or config-3) frame (seed given) at a given launch shape through whatever library PLSVO_HIP_LIB names -- the emulated one in the
CPU suite, the gfx950 one on a GPU box (tools/r04_ab.sh) --; prints a JSON line:
whether the device followed the oracle's Gauss-Newton path, the near-tie counters, the inter-frame pose error.  tests/test_emu_parity.py
uses it on a seed whose near tie falls on an iteration whose per-pixel terms the default build does not keep.  usage: ... <seed> [threads per frame, default 64]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob  # noqa: E402
import helpers as Hh  # noqa: E402

ob.build()
seed = int(sys.argv[1])
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = P.capi.Context(0)
cfg3 = len(sys.argv) > 3 and sys.argv[3] == "config3"
st, ref, cur, job = Hh.make_case(ob, seed, 1280, 720, 400, 150, 5, 4, 2) if cfg3 else Hh.make_case(ob, seed, 640, 480, 200, 80, 4, 3, 1)
res_o, log_o = ob.sparse_align(job, ref, cur, max_log=200)
ctx.set_launch_shapes(align_threads=threads)
ctx.config_pyramids(2, 1280, 720, 5) if cfg3 else ctx.config_pyramids(2, 640, 480, 4)
ctx.upload_pyramid(0, ref)
ctx.upload_pyramid(1, cur)
ctx.align_set_trace(200)
res_d = ctx.sparse_align(job)
log_d = ctx.align_fetch_trace(0)
ang, tr, _ = Hh.pose_close(res_d.T, res_o.T)
iters, ties, unarmed = ctx.align_chi2_ties()
print(json.dumps({"seed": seed, "same_path": bool(Hh.same_path(log_o, log_d)), "gn_iterations": int(iters), "decided_on_exact_sums": int(ties),
                  "near_ties_without_terms": int(unarmed), "inter_rot_rad": float(ang), "inter_trans_rel": float(tr),
                  "iters_device": list(res_d.iters_per_level)[:4], "iters_oracle": list(res_o.iters_per_level)[:4]}))
