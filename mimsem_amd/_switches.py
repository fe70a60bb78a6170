"""Run-time switches.  The product reads a short list of documented user switches (MIMSEM_LIB, MIMSEM_VERBOSE,
MIMSEM_COLUMN_PIVOT_FALLBACK, MIMSEM_REFINE, MIMSEM_NO_REFINE, MIMSEM_WAVE, MIMSEM_SCHUR_FUSED, MIMSEM_SCHUR3_CHAIN, MIMSEM_SCHUR3_SUPERBLOCKS, MIMSEM_PATCH_MAP, MIMSEM_MASS_SOLVER, MIMSEM_CHEB_CALIBRATE,
MIMSEM_SW_CHEB, MIMSEM_SW_GRAPH_ITER, MIMSEM_NEWTON_FUSED, MIMSEM_PCG).  Everything else -- the switches of variants that were built,
measured and declined, whose records live under profiles/ -- is a CLOSED EXPERIMENT: in this package read only when MIMSEM_EXPERIMENTS=1 is
set (scripts/ab_*.sh and the parity tests of those variants set it); in the library not even compiled in unless it was built with
-DMIMSEM_WITH_EXPERIMENTS (csrc/ctx.hpp: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS", then MIMSEM_LIB=build_ab/libmimsem_hip_exp.so)."""
import os


def experiments_on():
    return os.environ.get("MIMSEM_EXPERIMENTS", "0") not in ("", "0")


def experiment(name, default):
    """the value of a closed experiment's switch: the environment's when MIMSEM_EXPERIMENTS=1, else the default (the measured best)"""
    return os.environ.get(name, default) if experiments_on() else default
