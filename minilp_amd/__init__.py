"""minilp_amd — MI355X-native revised-simplex pivot engine behind minilp's Problem/Solution API.

The hot path (pricing, FTRAN/BTRAN, ratio tests, tableau-row product, basis update) is hand-written
HIP for gfx950 in `csrc/`, reached through the C ABI of `include/minilp_hip.h`.  This package is the
Python mirror of the reference's public interface (lib.rs:61-464).  There is no CPU fallback: without
the built extension or without a GPU every solve raises.
"""
from .api import (EQ, GE, LE, MAXIMIZE, MINIMIZE, Infeasible, InternalError, LinearExpr, MpsFile, Problem, Solution,
                  Unbounded, device_count, lib, lib_path, min_cut, set_device)

__all__ = ["Problem", "Solution", "MpsFile", "LinearExpr", "MINIMIZE", "MAXIMIZE", "EQ", "LE", "GE", "Infeasible", "Unbounded",
           "InternalError", "device_count", "set_device", "lib", "lib_path", "min_cut"]
