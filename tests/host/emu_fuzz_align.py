"""Random-configuration fuzz of the alignment through whatever library PLSVO_HIP_LIB names (meant for the host emulation build: a case
takes 0.05 - 0.3 s): image size, pyramid depth, level range, feature counts (0 .. 257 points, 0 .. 80 segments), segment lengths, iteration
cap, motion size and launch shape drawn at random; every case compared with the oracle -- same Gauss-Newton path => same counts, culls
and poses to 1e-7; a different path is reported with the records at which the two part.
Round 3, 900 cases on the emulated library: no crash, no inconsistency.  What the report lists is of three known kinds: (1) one or
two features (rank-deficient H: the reference's own step is rounding residue there, tests/test_solve_model.py); (2) alignments the
oracle itself loses (step_rot > 0.1 rad); (3) float-tie partings and 1e-7 .. 1e-6 differences of segment-dominated frames (the per-line
float sums of |res|), all inside the parity bar.
usage: PLSVO_HIP_LIB=<emu>/libplsvo_hip_emu.so python tests/host/emu_fuzz_align.py <cases> [first case index]"""
import importlib, sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob
ob.build()
import helpers as Hh
ctx = P.capi.Context(0)
N=int(sys.argv[1]); start=int(sys.argv[2]) if len(sys.argv)>2 else 0
rng=np.random.default_rng(12345+start)
bad=[]; parted=[]; n_done=0
for k in range(N):
    W = int(rng.choice([96,128,160,200,322,320,640,500])); H = int(rng.choice([72,96,120,150,242,240,480,376]))
    nlev = int(rng.integers(2,6))
    while (min(W,H) >> (nlev-1)) < 20: nlev -= 1
    nlev = max(nlev,1)
    maxl = int(rng.integers(0,nlev)); minl = int(rng.integers(0,maxl+1))
    npts = int(rng.choice([0,1,2,5,17,33,64,100,200,257])); nseg = int(rng.choice([0,1,3,10,31,32,33,80]))
    if npts+nseg==0: npts=7
    T = int(rng.choice([64,128,256,512,0]))
    n_iter = int(rng.choice([1,3,10,30]))
    ms = float(rng.choice([0.05,0.2,0.5,1.0]))
    slr = None if rng.random()<0.6 else (float(rng.choice([8,40])), float(rng.choice([60,300])))
    seed = 90000+start+k
    cfg=dict(seed=seed,W=W,H=H,nlev=nlev,maxl=maxl,minl=minl,npts=npts,nseg=nseg,T=T,n_iter=n_iter,ms=ms,slr=slr)
    try:
        st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl, n_iter, motion_scale=ms, seg_len_range=slr)
        res_o, log_o = ob.sparse_align(job, ref, cur, max_log=400)
        ctx.set_launch_shapes(align_threads=T)
        ctx.config_pyramids(2, W, H, nlev); ctx.upload_pyramid(0, ref); ctx.upload_pyramid(1, cur); ctx.align_set_trace(400)
        res_d = ctx.sparse_align(job); log_d = ctx.align_fetch_trace(0)
        same = Hh.same_path(log_o, log_d)
        ang, dist = P.synth.se3_log_angle_dist(res_d.T, res_o.T)
        step_rot, step_tr = P.synth.se3_log_angle_dist(res_o.T, st.T_init)
        n_done+=1
        if same:
            ok = np.array_equal(res_d.seg_alive, res_o.seg_alive) and res_d.n_meas==res_o.n_meas and list(res_d.iters_per_level)==list(res_o.iters_per_level) and res_d.status==res_o.status
            okp = (ang < 1e-7 and dist < 1e-7) or (np.isnan(ang) and np.isnan(P.synth.se3_log_angle_dist(res_o.T, res_o.T)[0])) or step_rot>0.1
            if not (ok and okp): bad.append(dict(cfg, why='same path but results differ', ang=float(ang), dist=float(dist), nm=(res_d.n_meas,res_o.n_meas)))
        else:
            parted.append(dict(cfg, ang=float(ang), dist=float(dist), prefix=Hh.common_prefix(log_o,log_d), n=(len(log_o),len(log_d)), step_rot=float(step_rot), last_o=[(r["level"],r["iter"],r["accepted"],r["new_chi2"],float(np.max(np.abs(r["x"])))) for r in log_o[max(0,Hh.common_prefix(log_o,log_d)-1):][:2]], last_d=[(r["level"],r["iter"],r["accepted"],r["new_chi2"],float(np.max(np.abs(r["x"])))) for r in log_d[max(0,Hh.common_prefix(log_o,log_d)-1):][:2]], ties=ctx.align_chi2_ties()))
    except Exception as e:
        bad.append(dict(cfg, why='exception '+repr(e)[:200]))
print(json.dumps({"done":n_done,"bad":bad,"parted":parted}))
