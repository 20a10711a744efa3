"""Test-side alias of the package's synthetic burst generators."""
import _pkg

_pkg.load()
from openbts_ttsou_amd.synth import *  # noqa: F401,F403,E402
