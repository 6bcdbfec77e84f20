"""everyvoice_amd — MI355X (gfx950) native implementation of the EveryVoice TTS hot path.

Python here is the host-side mirror of the reference's interface for this path only
(``HiFiGANGenerator`` / ``load_hifigan_from_checkpoint`` / ``expand`` ...); the arithmetic runs in
hand-written HIP kernels behind the C ABI of ``libevmi_hip.so`` (``include/evmi.h``).
"""

__version__ = "0.5.0"  # tracks the reference's major.minor (everyvoice/_version.py:5)
