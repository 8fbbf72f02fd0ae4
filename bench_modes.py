"""Timed entries of bench.py beyond independent streams (filled in as the modes are built)."""


def _todo(name):
    raise SystemExit(f"bench.py: {name} is not available in this build")


def run_bands(a, D, wl, mask, cfg):
    _todo("--mode bands")


def run_timeshard(a, D, wl, mask, cfg):
    _todo("--mode timeshard")


def run_c1e(a, D, wl, mask, cfg):
    _todo("--workload C1e")
