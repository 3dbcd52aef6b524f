"""omnivggt-official_amd: MI355X-native (gfx950) implementation of the OmniVGGT
multi-view aggregator hot path behind the reference's OmniVGGT.forward contract.

The directory name carries a hyphen (it mirrors the upstream repo name); import it
as `omnivggt_official_amd` (repo-root shim module) -- see DESIGN.md.
"""
__all__ = ["lib", "build", "ops", "model", "aggregator", "heads", "heads_hip", "sharding", "postprocess", "weights", "camera_math"]
__version__ = "0.1.0"
