"""MI355X-native 3D Monte Carlo photon tracer (drop-in for the raytracer_3d.cu path of Chiil/raytracer)."""
from .tracer import (PhotonTracer, RtError, TALLY_NAMES, EVENT_NAMES, load_library, save_tallies,
                     photon_balance, algorithmic_bytes, split_tally_words, choose_launch_order)
from . import scenes
from . import distributed

__all__ = ["PhotonTracer", "RtError", "TALLY_NAMES", "EVENT_NAMES", "load_library", "save_tallies",
           "photon_balance", "algorithmic_bytes", "split_tally_words", "choose_launch_order", "scenes", "distributed"]
