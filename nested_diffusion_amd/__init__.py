"""nested_diffusion_amd -- MI355X-native inference hot path of nested-diffusion (LaDiNE).

Host-side mirror of the reference's Python interface for the path
  mapping network -> p_sample_loop over ConditionalModel -> averaged class probabilities
(diffusion/diffusion_utils.py, diffusion/latent_model.py, diffusion/classification_train_separately.py
:749-794 of the reference) on top of libnd_hip.so (hand-written HIP for gfx950, C ABI in
include/nested_diffusion.h).  No CPU fallback: operators raise when the HIP library is missing.
"""
__version__ = "0.1.0"
