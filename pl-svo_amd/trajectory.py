"""Trajectory wire format of the reference's harness (app/run_pipeline.cpp:433-451): one line per frame,

    <timestamp> tx ty tz qx qy qz qw        (camera pose in the world = T_f_w^-1, TUM RGB-D convention)

written with a default-constructed std::ofstream, i.e. operator<<(double) at precision 6 in %g style.  The record
itself (inverse pose + the reference's skip rules) comes from the C ABI (plsvo_trajectory_record); this module only
formats it, so files written here can be diffed against run_pipeline's."""
from . import capi


def format_number(x):
    """std::ostream << double with default flags: %g with 6 significant digits"""
    return "%g" % x


def tum_line(timestamp, T_f_w, cov):
    """-> the text line, or None when the reference skips the frame (:425-443)"""
    ok, rec = capi.trajectory_record(T_f_w, cov)
    if not ok:
        return None
    return " ".join([str(timestamp)] + [format_number(v) for v in rec])


def write_trajectory(path, timestamps, poses_T_f_w, covs):
    n = 0
    with open(path, "w") as f:
        for ts, T, cv in zip(timestamps, poses_T_f_w, covs):
            line = tum_line(ts, T, cv)
            if line is not None:
                f.write(line + "\n")
                n += 1
    return n
