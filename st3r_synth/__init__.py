"""Synthetic scenes, condensed alignment problems, pairwise predictions and stand-in networks for the tests, bench.py,
tools/ and examples/ (SYNTH-1M and the synthetic condensed problems).  Pure numpy / torch-CPU generators:
NOT part of the product package -- nothing under starst3r_amd/ imports this."""
from . import synth, synth_align, synth_model, synth_pairs  # noqa: F401
