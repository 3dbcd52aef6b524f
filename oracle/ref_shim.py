"""Import shim for the upstream reference (TEST INFRASTRUCTURE -- only usable where
/root/reference exists, i.e. in the build container, never on the GPU box).

The reference's model module pulls in cv2 and evo transitively
(omnivggt/utils/misc.py:5,7 <- utils/geometry.py:12 <- models/omnivggt_aggregator.py:10)
and calls torch.hub.load at construction (models/aggregator.py:191-193).  Neither is
touched by the forward pass, so they are stubbed here WITHOUT modifying the reference.
"""
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("OVG_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "omnivggt"))


def install():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    for name in ["cv2", "evo", "evo.main_ape", "evo.main_rpe", "evo.core", "evo.core.sync", "evo.core.metrics",
                 "evo.core.trajectory", "evo.tools", "evo.tools.file_interface", "evo.tools.plot",
                 "evo.core.geometry", "evo.core.lie_algebra"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def build_reference_model():
    """Instantiate the reference OmniVGGT with torch.hub.load patched out (no network)."""
    install()
    import torch

    class _NoHub:
        def state_dict(self):
            return {}

    with mock.patch.object(torch.hub, "load", lambda *a, **k: _NoHub()):
        from omnivggt.models.omnivggt import OmniVGGT
        model = OmniVGGT()
    return model.eval()
