"""Run the frame-to-frame harness (pl-svo_amd/sequence.py) on the GPU and write the trajectory in the reference
harness's format (app/run_pipeline.cpp:433-451: `timestamp tx ty tz qx qy qz qw` of T_f_w^-1).
With mapping (default on) 40 % of the point landmarks start as depth-filter seeds and structure optimisation runs at every
fifth frame, so all of align / reproject / match / pose-opt / structure-opt / seed update are exercised.
usage: python tools/run_sequence.py [out.txt] [n_frames] [seed] [mapping 0|1]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("pl-svo_amd")
seqm = importlib.import_module("pl-svo_amd.sequence")

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "trajectory.txt")
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
mapping = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
seq = seqm.make_sequence(seed, n_frames, 640, 480, 200, 80, total=0.35)      # 35 % of the scene depth + 0.09 rad over the whole run
ctx = P.capi.Context(0)
res = seqm.run_sequence(seqm.HipBackend(ctx), seq, mapping=mapping)
n = P.trajectory.write_trajectory(out, ["%.6f" % (0.05 * k) for k in range(n_frames)], [r["T"] for r in res], [r["cov"] for r in res])
err = seqm.pose_errors(res, seq)
print(json.dumps({"frames": n_frames, "lines_written": n, "trajectory": out, "max_rot_err_rad": max(e[0] for e in err),
                  "max_trans_err_m": max(e[1] for e in err), "matched_points_last": res[-1]["n_matched_pt"], "matched_segments_last": res[-1]["n_matched_seg"],
                  "mapping": mapping, "landmarks_first_last": [res[1].get("n_known"), res[-1].get("n_known")],
                  "seeds_first_last": [res[1].get("n_seeds"), res[-1].get("n_seeds")]}))
ctx.close()
