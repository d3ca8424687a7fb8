"""ctypes binding of the product library libplsvo_hip.so (C ABI: include/plsvo_hip.h).

There is NO CPU fallback here: loading fails loudly if the HIP extension is missing, and Context()
raises if no gfx950 device can be used.  Nothing in this module imports or calls the CPU oracle.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLSVO_HIP_LIB", os.path.join(_HERE, "libplsvo_hip.so"))  # override only for instrumented builds

# every symbol include/plsvo_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "plsvo_hip_create", "plsvo_hip_create_on_stream", "plsvo_hip_set_option", "plsvo_align_slot_layout", "plsvo_hip_destroy", "plsvo_hip_last_error", "plsvo_hip_stream", "plsvo_hip_synchronize",
    "plsvo_hip_config_pyramids", "plsvo_hip_upload_pyramid", "plsvo_hip_build_pyramid", "plsvo_hip_build_pyramids_dev",
    "plsvo_hip_download_level", "plsvo_hip_copy_slots",
    "plsvo_sparse_align", "plsvo_sparse_align_batch", "plsvo_align_stage", "plsvo_align_run", "plsvo_align_fetch",
    "plsvo_align_set_trace", "plsvo_align_fetch_trace", "plsvo_align_poses_dev", "plsvo_align_copy_poses", "plsvo_align_work", "plsvo_align_work_points", "plsvo_align_chi2_ties", "plsvo_align_launch_order",
    "plsvo_pose_optimize", "plsvo_pose_optimize_batch", "plsvo_poseopt_stage", "plsvo_poseopt_run", "plsvo_poseopt_fetch",
    "plsvo_poseopt_set_trace", "plsvo_poseopt_fetch_trace", "plsvo_poseopt_poses_dev", "plsvo_poseopt_copy_poses", "plsvo_poseopt_work",
    "plsvo_structure_optimize", "plsvo_match_direct", "plsvo_reproject", "plsvo_trajectory_record", "plsvo_update_seeds",
    "plsvo_chain_stage", "plsvo_chain_run", "plsvo_chain_fetch", "plsvo_frame_step_batch", "plsvo_chain_poses_dev",
    "plsvo_pack_pose_records", "plsvo_fetch_pose_records", "plsvo_gather_poses",
    "plsvo_hip_set_profiling", "plsvo_hip_kernel_time", "plsvo_hip_reset_profiling",
    "plsvo_hip_version", "plsvo_hip_build_flags", "plsvo_hip_device_info",
]


class PlsvoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"plsvo_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libplsvo_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C pl-svo_amd/csrc` "
                          f"(or __graft_entry__.build()); plsvo_hip has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    if hasattr(L, "plsvo_emu_build"):
        # tests/host/build_emu.sh: the device sources on a CPU wave emulator.  Only ever reached through an explicit PLSVO_HIP_LIB
        # (tests/test_emu_parity.py); said out loud so that nothing measured or shipped can pass for the gfx950 library by accident.
        sys.stderr.write(f"pl-svo_amd: {LIB_PATH} is a HOST EMULATION build of the kernels (test infrastructure), not the gfx950 library\n")
    ctxp = C.c_void_p
    vp = C.c_void_p
    sig = {
        "plsvo_hip_create": (C.c_int, [C.c_int, vp, C.POINTER(ctxp)]),
        "plsvo_hip_create_on_stream": (C.c_int, [C.c_int, vp, C.POINTER(ctxp)]),
        "plsvo_hip_destroy": (None, [ctxp]),
        "plsvo_hip_set_option": (C.c_int, [ctxp, C.c_int, C.c_int]),
        "plsvo_hip_last_error": (C.c_char_p, [ctxp]),
        "plsvo_hip_stream": (vp, [ctxp]),
        "plsvo_hip_synchronize": (C.c_int, [ctxp]),
        "plsvo_hip_config_pyramids": (C.c_int, [ctxp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "plsvo_hip_upload_pyramid": (C.c_int, [ctxp, C.c_int, C.c_int, C.POINTER(abi.c_u8_p), abi.c_i32_p, abi.c_i32_p, abi.c_i32_p]),
        "plsvo_hip_build_pyramid": (C.c_int, [ctxp, C.c_int, abi.c_u8_p, C.c_int, C.c_int]),
        "plsvo_hip_build_pyramids_dev": (C.c_int, [ctxp, C.c_int, C.c_int, vp, C.c_int, C.c_size_t, C.c_int]),
        "plsvo_hip_download_level": (C.c_int, [ctxp, C.c_int, C.c_int, abi.c_u8_p]),
        "plsvo_hip_copy_slots": (C.c_int, [ctxp, C.c_int, C.c_int, C.c_int]),
        "plsvo_sparse_align": (C.c_int, [ctxp, C.POINTER(abi.AlignIn), C.POINTER(abi.AlignOut)]),
        "plsvo_sparse_align_batch": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.AlignIn), C.POINTER(abi.AlignOut)]),
        "plsvo_align_stage": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.AlignIn)]),
        "plsvo_align_run": (C.c_int, [ctxp]),
        "plsvo_align_fetch": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.AlignOut)]),
        "plsvo_align_set_trace": (C.c_int, [ctxp, C.c_int]),
        "plsvo_align_fetch_trace": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.AlignIterLog), C.c_int, C.POINTER(C.c_int)]),
        "plsvo_align_poses_dev": (vp, [ctxp]),
        "plsvo_align_copy_poses": (C.c_int, [ctxp, vp]),
        "plsvo_poseopt_copy_poses": (C.c_int, [ctxp, vp]),
        "plsvo_align_work": (C.c_int, [ctxp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "plsvo_align_chi2_ties": (C.c_int, [ctxp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "plsvo_align_work_points": (C.c_int, [ctxp, C.POINTER(C.c_uint64)]),
        "plsvo_align_launch_order": (C.c_int, [ctxp, C.c_int, C.POINTER(C.c_int32)]),
        "plsvo_pose_optimize": (C.c_int, [ctxp, C.POINTER(abi.PoseOptIn), C.POINTER(abi.PoseOptOut)]),
        "plsvo_pose_optimize_batch": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.PoseOptIn), C.POINTER(abi.PoseOptOut)]),
        "plsvo_poseopt_stage": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.PoseOptIn)]),
        "plsvo_poseopt_run": (C.c_int, [ctxp]),
        "plsvo_poseopt_fetch": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.PoseOptOut)]),
        "plsvo_poseopt_set_trace": (C.c_int, [ctxp, C.c_int]),
        "plsvo_poseopt_fetch_trace": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.PoseOptIterLog), C.c_int, C.POINTER(C.c_int)]),
        "plsvo_poseopt_poses_dev": (vp, [ctxp]),
        "plsvo_poseopt_work": (C.c_int, [ctxp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "plsvo_structure_optimize": (C.c_int, [ctxp, C.POINTER(abi.StructOptIn), C.POINTER(abi.StructOptOut)]),
        "plsvo_match_direct": (C.c_int, [ctxp, C.POINTER(abi.MatchIn), C.POINTER(abi.MatchOut)]),
        "plsvo_reproject": (C.c_int, [ctxp, C.POINTER(abi.ReprojectIn), C.POINTER(abi.ReprojectOut)]),
        "plsvo_update_seeds": (C.c_int, [ctxp, C.POINTER(abi.SeedsIn), C.POINTER(abi.SeedsOut)]),
        "plsvo_trajectory_record": (C.c_int, [abi.c_double_p, abi.c_double_p, abi.c_double_p]),
        "plsvo_chain_stage": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.ChainIn), C.POINTER(abi.ChainParams)]),
        "plsvo_chain_run": (C.c_int, [ctxp]),
        "plsvo_chain_fetch": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.ChainOut)]),
        "plsvo_frame_step_batch": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.ChainIn), C.POINTER(abi.ChainParams), C.POINTER(abi.ChainOut)]),
        "plsvo_chain_poses_dev": (vp, [ctxp]),
        "plsvo_align_slot_layout": (C.c_int, [C.POINTER(abi.AlignIn), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_longlong)]),
        "plsvo_pack_pose_records": (C.c_int, [ctxp, vp, C.POINTER(C.c_int)]),
        "plsvo_fetch_pose_records": (C.c_int, [ctxp, C.c_int, C.POINTER(abi.PoseRecord)]),
        "plsvo_gather_poses": (C.c_int, [ctxp, vp, vp, C.c_int, vp]),
        "plsvo_hip_set_profiling": (C.c_int, [ctxp, C.c_int]),
        "plsvo_hip_kernel_time": (C.c_int, [ctxp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
        "plsvo_hip_reset_profiling": (C.c_int, [ctxp]),
        "plsvo_hip_version": (C.c_char_p, []),
        "plsvo_hip_build_flags": (C.c_char_p, []),
        "plsvo_hip_device_info": (C.c_int, [ctxp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    }
    for name, (res, args) in sig.items():
        if "PLSVO_HIP_LIB" in os.environ and not hasattr(L, name):
            continue   # A/B against an older instrumented build (experiments only): a missing entry point fails when it is called
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


class Context:
    """One plsvo_ctx: a device + stream + HBM buffers.  Mirrors the C ABI one-to-one."""

    def __init__(self, device=0, stream=None):
        """stream: None -> the ctx creates a private non-blocking stream; an integer hipStream_t handle -> every launch goes
        on THAT stream, 0 included (0 is HIP's default stream, e.g. torch.cuda.current_stream().cuda_stream)."""
        self.L = lib()
        h = C.c_void_p()
        if stream is None:
            rc = self.L.plsvo_hip_create(int(device), None, C.byref(h))
        else:
            rc = self.L.plsvo_hip_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if rc != 0:
            raise PlsvoError(rc, (self.L.plsvo_hip_last_error(None) or b"").decode())
        self.h = h
        self._keep = None

    def set_ldlt_flavour(self, flavour):
        """320 (Eigen 3.1 ... 3.2.1, default) or 330 (Eigen 3.2.2+): zero-pivot rule of the optimisers' 6x6 LDLT solve"""
        self._chk(self.L.plsvo_hip_set_option(self.h, 1, int(flavour)))

    def set_launch_shapes(self, align_threads=None, poseopt_threads=None):
        """fix the threads-per-frame of the alignment / pose-optimiser kernels (0 = automatic); for tests and measurements"""
        if align_threads is not None:
            self._chk(self.L.plsvo_hip_set_option(self.h, 2, int(align_threads)))
        if poseopt_threads is not None:
            self._chk(self.L.plsvo_hip_set_option(self.h, 3, int(poseopt_threads)))

    def set_launch_order_refresh(self, align=None, poseopt=None):
        """launch order of a RE-RUN staged batch: True (default) = longest-first by the previous launch's measured work, False = the stage
        call's order for every launch (PLSVO_OPT_ALIGN_REORDER / PLSVO_OPT_POSEOPT_REORDER); scheduling only"""
        if align is not None:
            self._chk(self.L.plsvo_hip_set_option(self.h, 4, 1 if align else 0))
        if poseopt is not None:
            self._chk(self.L.plsvo_hip_set_option(self.h, 5, 1 if poseopt else 0))

    def close(self):
        if getattr(self, "h", None):
            self.L.plsvo_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PlsvoError(rc, (self.L.plsvo_hip_last_error(self.h) or b"").decode())

    # ---- info / timing ----
    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = C.c_size_t(0)
        self._chk(self.L.plsvo_hip_device_info(self.h, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value

    def stream(self):
        return self.L.plsvo_hip_stream(self.h)

    def synchronize(self):
        self._chk(self.L.plsvo_hip_synchronize(self.h))

    def set_profiling(self, on):
        self._chk(self.L.plsvo_hip_set_profiling(self.h, 1 if on else 0))

    def reset_profiling(self):
        self._chk(self.L.plsvo_hip_reset_profiling(self.h))

    def kernel_time(self, k):
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._chk(self.L.plsvo_hip_kernel_time(self.h, k, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- pyramids ----
    def config_pyramids(self, n_slots, width, height, n_levels):
        self._chk(self.L.plsvo_hip_config_pyramids(self.h, n_slots, width, height, n_levels))
        self.n_levels = n_levels
        self.width, self.height = width, height

    def upload_pyramid(self, slot, levels):
        keep = [np.ascontiguousarray(l, dtype=np.uint8) for l in levels]
        n = len(keep)
        ptrs = (abi.c_u8_p * n)(*[l.ctypes.data_as(abi.c_u8_p) for l in keep])
        w = (C.c_int32 * n)(*[l.shape[1] for l in keep])
        h = (C.c_int32 * n)(*[l.shape[0] for l in keep])
        s = (C.c_int32 * n)(*[l.strides[0] for l in keep])
        self._chk(self.L.plsvo_hip_upload_pyramid(self.h, slot, n, ptrs, w, h, s))

    def build_pyramid(self, slot, img0, rounding=0):
        img0 = np.ascontiguousarray(img0, dtype=np.uint8)
        self._chk(self.L.plsvo_hip_build_pyramid(self.h, slot, img0.ctypes.data_as(abi.c_u8_p), img0.strides[0], rounding))

    def build_pyramids_dev(self, first_slot, n, d_ptr, stride_bytes, image_pitch_bytes, rounding=0):
        self._chk(self.L.plsvo_hip_build_pyramids_dev(self.h, first_slot, n, C.c_void_p(d_ptr), stride_bytes,
                                                      image_pitch_bytes, rounding))

    def download_level(self, slot, level):
        w, h = self.width >> level, self.height >> level
        out = np.empty((h, w), dtype=np.uint8)
        self._chk(self.L.plsvo_hip_download_level(self.h, slot, level, out.ctypes.data_as(abi.c_u8_p)))
        return out

    def copy_slots(self, dst_first, src_first, n):
        self._chk(self.L.plsvo_hip_copy_slots(self.h, int(dst_first), int(src_first), int(n)))

    def download_pyramid(self, slot):
        return [self.download_level(slot, l) for l in range(self.n_levels)]

    # ---- sparse image alignment ----
    def align_set_trace(self, max_records):
        self._chk(self.L.plsvo_align_set_trace(self.h, max_records))
        self._align_trace = max_records

    def align_stage(self, jobs):
        arr = (abi.AlignIn * len(jobs))(*[j.c for j in jobs])
        self._chk(self.L.plsvo_align_stage(self.h, len(jobs), arr))
        self._align_jobs = jobs

    def align_run(self):
        self._chk(self.L.plsvo_align_run(self.h))

    def align_fetch(self):
        jobs = self._align_jobs
        n = len(jobs)
        outs = (abi.AlignOut * n)()
        alive = [np.ones(max(j.n_seg, 1), dtype=np.uint8) for j in jobs]
        for o, a in zip(outs, alive):
            o.seg_alive_out = a.ctypes.data_as(abi.c_u8_p)
        self._chk(self.L.plsvo_align_fetch(self.h, n, outs))
        return [abi.AlignResult(o, a[:j.n_seg].copy()) for o, a, j in zip(outs, alive, jobs)]

    def align_fetch_trace(self, job):
        cap = getattr(self, "_align_trace", 0)
        log = (abi.AlignIterLog * max(cap, 1))()
        n = C.c_int(0)
        self._chk(self.L.plsvo_align_fetch_trace(self.h, job, log, cap, C.byref(n)))
        return abi.align_log_to_dicts(log, n.value)

    def sparse_align_batch(self, jobs):
        self.align_stage(jobs)
        self.align_run()
        return self.align_fetch()

    def sparse_align(self, job):
        return self.sparse_align_batch([job])[0]

    def align_work(self):
        a = C.c_uint64(0)
        b = C.c_uint64(0)
        self._chk(self.L.plsvo_align_work(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def align_work_points(self):
        a = C.c_uint64(0)
        self._chk(self.L.plsvo_align_work_points(self.h, C.byref(a)))
        return a.value

    def align_launch_order(self, n):
        """plsvo_align_launch_order: the job each block of the next align_run of the resident batch works on"""
        out = np.zeros(n, dtype=np.int32)
        self._chk(self.L.plsvo_align_launch_order(self.h, n, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def align_chi2_ties(self):
        """(Gauss-Newton iterations, iterations decided on the exact float chi2 sums, near ties whose terms had not been kept) of the last align_run"""
        a = C.c_uint64(0)
        b = C.c_uint64(0)
        c = C.c_uint64(0)
        self._chk(self.L.plsvo_align_chi2_ties(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def align_poses_dev(self):
        return self.L.plsvo_align_poses_dev(self.h)

    def align_copy_poses(self, d_dst):
        self._chk(self.L.plsvo_align_copy_poses(self.h, C.c_void_p(d_dst)))

    def poseopt_copy_poses(self, d_dst):
        self._chk(self.L.plsvo_poseopt_copy_poses(self.h, C.c_void_p(d_dst)))

    # ---- pose optimisation ----
    def poseopt_set_trace(self, max_records):
        self._chk(self.L.plsvo_poseopt_set_trace(self.h, max_records))
        self._pose_trace = max_records

    def poseopt_stage(self, jobs):
        arr = (abi.PoseOptIn * len(jobs))(*[j.c for j in jobs])
        self._chk(self.L.plsvo_poseopt_stage(self.h, len(jobs), arr))
        self._pose_jobs = jobs

    def poseopt_run(self):
        self._chk(self.L.plsvo_poseopt_run(self.h))

    def poseopt_fetch(self):
        jobs = self._pose_jobs
        n = len(jobs)
        outs = (abi.PoseOptOut * n)()
        pk = [np.ones(max(j.n_pts, 1), dtype=np.uint8) for j in jobs]
        sk = [np.ones(max(j.n_seg, 1), dtype=np.uint8) for j in jobs]
        for o, a, b in zip(outs, pk, sk):
            o.pt_keep = a.ctypes.data_as(abi.c_u8_p)
            o.seg_keep = b.ctypes.data_as(abi.c_u8_p)
        self._chk(self.L.plsvo_poseopt_fetch(self.h, n, outs))
        return [abi.PoseOptResult(o, a[:j.n_pts].copy(), b[:j.n_seg].copy()) for o, a, b, j in zip(outs, pk, sk, jobs)]

    def poseopt_fetch_trace(self, job):
        cap = getattr(self, "_pose_trace", 0)
        log = (abi.PoseOptIterLog * max(cap, 1))()
        n = C.c_int(0)
        self._chk(self.L.plsvo_poseopt_fetch_trace(self.h, job, log, cap, C.byref(n)))
        return abi.poseopt_log_to_dicts(log, n.value)

    def pose_optimize_batch(self, jobs):
        self.poseopt_stage(jobs)
        self.poseopt_run()
        return self.poseopt_fetch()

    def pose_optimize(self, job):
        return self.pose_optimize_batch([job])[0]

    def poseopt_work(self):
        a = C.c_uint64(0)
        b = C.c_uint64(0)
        self._chk(self.L.plsvo_poseopt_work(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def poseopt_poses_dev(self):
        return self.L.plsvo_poseopt_poses_dev(self.h)

    # ---- resident frame step ----
    def chain_stage(self, jobs, cam, n_pyr_levels=3, align_max_iter=10, cell_size=30, cell_rule=False, max_fts=120, cell_order=None,
                    reproj_thresh=2.0, poseopt_n_iter=10, seg_cell_size=0, max_fts_segs=100, seg_cell_order=None):
        n = len(jobs)
        arr = (abi.ChainIn * n)(*[j.c for j in jobs])
        pr = abi.ChainParams()
        pr.cam = cam if isinstance(cam, abi.Pinhole) else abi.Pinhole(*cam)
        pr.n_pyr_levels, pr.align_max_iter, pr.cell_size, pr.cell_rule = int(n_pyr_levels), int(align_max_iter), int(cell_size), int(bool(cell_rule))
        pr.max_fts, pr.poseopt_n_iter, pr.reproj_thresh = int(max_fts), int(poseopt_n_iter), float(reproj_thresh)
        self._chain_order = None if cell_order is None else np.ascontiguousarray(cell_order, dtype=np.int32)
        pr.cell_order = C.cast(None, abi.c_i32_p) if self._chain_order is None else self._chain_order.ctypes.data_as(abi.c_i32_p)
        # the segments' grid (gridls_): a segment filed under both end-point cells, one success per cell, max_fts_segs
        pr.seg_cell_size, pr.max_fts_segs = int(seg_cell_size), int(max_fts_segs)
        self._chain_seg_order = None if seg_cell_order is None else np.ascontiguousarray(seg_cell_order, dtype=np.int32)
        pr.seg_cell_order = C.cast(None, abi.c_i32_p) if self._chain_seg_order is None else self._chain_seg_order.ctypes.data_as(abi.c_i32_p)
        self._chain_seg_mult = 2 if (cell_rule and seg_cell_size > 0) else 1     # a segment that wins both of its cells is a feature twice
        self._chk(self.L.plsvo_chain_stage(self.h, n, arr, C.byref(pr)))
        self._chain_jobs = list(jobs)
        self._align_jobs = [j.align_job for j in jobs]

    def chain_run(self):
        self._chk(self.L.plsvo_chain_run(self.h))

    def chain_fetch(self):
        jobs = self._chain_jobs
        n = len(jobs)
        outs = (abi.ChainOut * n)()
        bufs = []
        for o, j in zip(outs, jobs):
            b = dict(alive=np.ones(max(j.align_job.n_seg, 1), np.uint8), pk=np.zeros(max(j.n_cand_pt, 1), np.uint8), sk=np.zeros(max(self._chain_seg_mult * j.n_cand_seg, 1), np.uint8),
                     found=np.zeros(max(j.n_cand, 1), np.uint8), px=np.zeros((max(j.n_cand, 1), 2)), level=np.zeros(max(j.n_cand, 1), np.int32),
                     sel_pt=np.zeros(max(j.n_cand_pt, 1), np.int32), sel_seg=np.zeros(max(self._chain_seg_mult * j.n_cand_seg, 1), np.int32))
            o.align.seg_alive_out = b["alive"].ctypes.data_as(abi.c_u8_p)
            o.pose.pt_keep = b["pk"].ctypes.data_as(abi.c_u8_p)
            o.pose.seg_keep = b["sk"].ctypes.data_as(abi.c_u8_p)
            o.found = b["found"].ctypes.data_as(abi.c_u8_p)
            o.px = b["px"].ctypes.data_as(abi.c_double_p)
            o.search_level = b["level"].ctypes.data_as(abi.c_i32_p)
            o.sel_pt = b["sel_pt"].ctypes.data_as(abi.c_i32_p)
            o.sel_seg = b["sel_seg"].ctypes.data_as(abi.c_i32_p)
            bufs.append(b)
        self._chk(self.L.plsvo_chain_fetch(self.h, n, outs))
        return [abi.ChainResult(o, j, b["alive"][:j.align_job.n_seg].copy(), b["pk"], b["sk"], b["found"], b["px"], b["level"], b["sel_pt"], b["sel_seg"])
                for o, j, b in zip(outs, jobs, bufs)]

    def frame_step_batch(self, jobs, cam, **params):
        self.chain_stage(jobs, cam, **params)
        self.chain_run()
        return self.chain_fetch()

    def chain_poses_dev(self):
        return self.L.plsvo_chain_poses_dev(self.h)

    # ---- structure optimisation ----
    def structure_optimize(self, job):
        out, bufs = job.make_out()
        self._chk(self.L.plsvo_structure_optimize(self.h, C.byref(job.c), C.byref(out)))
        return job.trim(bufs)

    # ---- direct feature matching ----
    def match_direct(self, job):
        out, bufs = job.make_out()
        self._chk(self.L.plsvo_match_direct(self.h, C.byref(job.c), C.byref(out)))
        return job.trim(bufs)

    def reproject(self, job):
        out, bufs = job.make_out()
        self._chk(self.L.plsvo_reproject(self.h, C.byref(job.c), C.byref(out)))
        return job.trim(bufs)

    def update_seeds(self, job):
        out, bufs = job.make_out()
        self._chk(self.L.plsvo_update_seeds(self.h, C.byref(job.c), C.byref(out)))
        return job.trim(bufs)

    def pack_pose_records(self, d_dst):
        """plsvo_pack_pose_records: one 96-byte plsvo_pose_record per stream of the resident batch into device memory at d_dst (enqueued on
        the ctx stream); returns the record count"""
        n = C.c_int(0)
        self._chk(self.L.plsvo_pack_pose_records(self.h, C.c_void_p(d_dst), C.byref(n)))
        return int(n.value)

    def fetch_pose_records(self, n):
        """plsvo_fetch_pose_records: the resident batch's n records as a numpy structured array (abi.POSE_RECORD_DTYPE)"""
        out = np.zeros(n, dtype=abi.POSE_RECORD_DTYPE)
        self._chk(self.L.plsvo_fetch_pose_records(self.h, n, out.ctypes.data_as(C.POINTER(abi.PoseRecord))))
        return out

    def gather_poses(self, rccl_comm, d_local, n_local, d_all):
        """plsvo_gather_poses: all-gather of n_local plsvo_pose_record (96 B each, device pointers) over an RCCL communicator"""
        self._chk(self.L.plsvo_gather_poses(self.h, C.c_void_p(rccl_comm), C.c_void_p(d_local), n_local, C.c_void_p(d_all)))


def trajectory_record(T_f_w, cov):
    """plsvo_trajectory_record: (write?, [tx ty tz qx qy qz qw] of T_f_w^-1) -- host-only, no ctx needed"""
    T = np.ascontiguousarray(T_f_w, dtype=np.float64)
    Cv = np.ascontiguousarray(cov, dtype=np.float64).reshape(36)
    out = np.empty(7)
    ok = lib().plsvo_trajectory_record(T.ctypes.data_as(abi.c_double_p), Cv.ctypes.data_as(abi.c_double_p), out.ctypes.data_as(abi.c_double_p))
    return bool(ok), out


def align_slot_layout(job, level):
    """plsvo_align_slot_layout: the static patch-slot layout of one alignment job at one level -- host-only, no ctx needed.
    Returns (first slot per segment or -1, samples per segment, slots in use, long_lines flag, patches)."""
    import numpy as np
    n_seg = int(job.c.n_seg)
    codes = (C.c_int32 * max(n_seg, 1))()
    n_slots, long_lines, n_patches = C.c_int32(0), C.c_int32(0), C.c_longlong(0)
    rc = lib().plsvo_align_slot_layout(C.byref(job.c), int(level), codes, C.byref(n_slots), C.byref(long_lines), C.byref(n_patches))
    if rc != 0:
        raise PlsvoError(rc, "plsvo_align_slot_layout failed")
    code = np.array(codes[:n_seg], dtype=np.int64)
    first = np.where(code >= 0, code & 0xfffff, -1)
    n = np.where(code >= 0, code >> 20, 0)
    return first, n, int(n_slots.value), bool(long_lines.value), int(n_patches.value)
