"""nanorq_amd -- MI355X-native RaptorQ precode solve / symbol generation (nanorq-compatible).

The product is the native library nanorq_amd/libnanorq_hip.so (gfx950 HIP kernels behind the C ABI
of include/nanorq_hip.h, plus the drop-in nanorq.h / io.h layer).  This package only loads it and
offers a thin ctypes mirror for tests and bench.py; there is no Python or CPU implementation of
the hot path, and every entry point raises when the HIP library or a GPU is missing.
"""
import os as _os

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues, 4 by default; a context runs up to six
# streams beside the caller's, and two streams on one queue run one after the other (the receiver pipeline: 313 -> 216 Gbit/s).
# The variable belongs to the HOST process and must be there before the runtime initialises (include/nanorq.h);
# for a Python host this package is that place.  A value already in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .binding import (NrqError, Context, lib, lib_path, params, host_plan, host_kconst, PLAN_FIELDS,
                      plan_header, plan_ops, plan_ops_store)

__all__ = ["NrqError", "Context", "lib", "lib_path", "params", "host_plan", "host_kconst", "PLAN_FIELDS",
           "plan_header", "plan_ops", "plan_ops_store"]
