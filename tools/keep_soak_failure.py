#!/usr/bin/env python3
"""Turns the record a session soak leaves after a mismatch (gpurun_out/soak_session_failure.npz, written by
tools/soak_session.py's diagnose()) into a test fixture: the frame's inputs only - parameters, camera list, object matrices,
scene name, size and the three targets as they stood before the frame.  What the frame must look like is not stored: the test
renders it with the exhaustive oracle.

    python tools/keep_soak_failure.py gpurun_out/soak_session_failure.npz tests/golden/session_far_plane_frame.npz

tests/golden/session_far_plane_frame.npz was made this way from the run `tools/soak_session.py 90 20261112` (frame 90 767)."""
import sys

import numpy as np

KEEP = ("params", "cams", "n_cams", "objs", "scene", "W", "H", "before")


def main(src, dst):
    z = np.load(src)
    np.savez_compressed(dst, **{k: z[k] for k in KEEP})
    print(dst, {k: z[k].shape for k in KEEP})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
