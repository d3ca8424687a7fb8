"""Per-call wall time of the C++ drop-in (adapter_driver --bench) on a BASELINE config-2 frame pair, with the host-side split of
plsvo_sparse_align_batch (PLSVO_HOST_TIMING).  usage: python tools/adapter_latency.py [calls]"""
import importlib, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh
from oracle import binding as ob
P = importlib.import_module("pl-svo_amd")
calls = sys.argv[1] if len(sys.argv) > 1 else "100"
st, ref, cur, job = Hh.make_case(ob, 1234, 640, 480, 200, 80, 4, 3, 1)
fr = P.synth.make_poseopt_frame(1234, 200, 80, 640, 480)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "in.bin")
    P.adapter_io.write_adapter_input(path, st, ref, cur, fr, 4, 3, 1)
    r = subprocess.run([os.path.join(ROOT, "pl-svo_amd/host/adapter_driver"), "--bench", calls, path], env=dict(os.environ, PLSVO_HOST_TIMING="1"),
                       capture_output=True, text=True)
print(r.stdout.strip())
print("\n".join([l for l in r.stderr.splitlines() if "sparse_align_batch" in l][-3:]))
