"""Host logic of bench.py that decides what the bench line may claim (no GPU): the hash that ties the offline counter traffic
(profiles/hbm_traffic*.json) to the kernel sources it was measured on, and the synthetic moving-sequence inputs of the launch-order leg."""
import importlib
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench

P = importlib.import_module("pl-svo_amd")


def test_kernel_source_hash_is_stable_and_covers_the_kernel_sources():
    a, b = bench.kernel_source_sha(), bench.kernel_source_sha()
    assert a == b and len(a) == 16 and all(c in "0123456789abcdef" for c in a)
    for f in bench.KERNEL_SOURCES:
        assert os.path.exists(os.path.join(ROOT, "pl-svo_amd", "csrc", f)), f


def test_offline_traffic_is_used_only_for_the_kernel_it_was_measured_on(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    csrc = tmp_path / "pl-svo_amd" / "csrc"
    csrc.mkdir(parents=True)
    for f in bench.KERNEL_SOURCES:
        (csrc / f).write_text("// " + f)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    sha = bench.kernel_source_sha()
    rec = {"batch": 32768, "measured_at": "abc1234", "kernel_source_sha": sha, "align_fused_kernel_bytes_per_stream": 1000.0,
           "align_fused_kernel_bytes_per_stream_uncorrected": 700.0}
    (prof / "hbm_traffic.json").write_text(json.dumps(rec))
    t = bench.offline_traffic(32768, 2)
    assert t["same_kernel"] and t["same_batch"] and t["bytes"] == 32768 * 1000 and t["raw"] == 32768 * 700 and "not scaled" in t["source"]
    t = bench.offline_traffic(8192, 2)                       # another batch: usable, but flagged as scaled
    assert t["same_kernel"] and not t["same_batch"] and t["bytes"] == 8192 * 1000 and "scaled to 8192" in t["source"]
    (csrc / bench.KERNEL_SOURCES[0]).write_text("// edited")   # the kernel changed: the counters describe another build
    t = bench.offline_traffic(32768, 2)
    assert not t["same_kernel"] and t["bytes"] == 32768 * 1000 and "NOT this tree" in t["source"]
    assert bench.offline_traffic(32768, 3)["bytes"] is None     # no file for that workload


def test_committed_traffic_files_parse_and_say_what_they_measured():
    for cfg, batch in ((2, 32768), (3, 16384)):
        t = bench.offline_traffic(batch, cfg)
        assert t["bytes"] is None or (t["bytes"] > 0 and isinstance(t["same_kernel"], bool) and t["source"])
        if t["bytes"] and not t["same_kernel"]:
            warnings.warn(f"profiles/hbm_traffic{'' if cfg == 2 else '_config3'}.json was measured on other kernel sources: bench.py will report "
                          "formulation_min as its roofline basis until tools/r06_pmc.sh is run again")


def test_moving_sequence_inputs():
    sy = P.synth
    st = sy.make_align_stream(77, 160, 120, 12, 4)
    # motions: deterministic, different per image, the two models differ in how they vary
    ind = [sy.stream_motion(st, k) for k in range(6)]
    smo = [sy.stream_motion(st, k, model="smooth") for k in range(6)]
    assert all(np.array_equal(sy.stream_motion(st, k), ind[k]) for k in range(6))
    assert len({tuple(np.round(T, 12)) for T in ind}) == 6 and len({tuple(np.round(T, 12)) for T in smo}) == 6
    mean_dir = np.mean([T[4:] / np.linalg.norm(T[4:]) for T in smo], axis=0)
    assert np.linalg.norm(mean_dir) > 0.9                     # smooth: one direction of travel; independent motions point anywhere
    assert abs(np.linalg.norm(ind[0][:4]) - 1.0) < 1e-12
    # views: the reference view and T_true reproduce render_streams' pair up to their own noise realisation
    pair = sy.render_streams([st], noise_sigma=0.0)
    ref = sy.render_views([st], [None], noise_sigma=0.0)
    cur = sy.render_views([st], [st.T_true], noise_sigma=0.0)
    assert np.array_equal(pair[0, 0].numpy(), ref[0].numpy()) and np.array_equal(pair[0, 1].numpy(), cur[0].numpy())
    other = sy.render_views([st], [ind[0]], noise_sigma=0.0)
    assert not np.array_equal(other[0].numpy(), cur[0].numpy())
