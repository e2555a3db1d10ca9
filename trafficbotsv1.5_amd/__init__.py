"""MI355X-native hot path of TrafficBots V1.5 (HPTR / KNARPE transformer + closed-loop rollout).

The directory name carries a dot, so the package is loaded under the module name ``trafficbots_amd`` by
``__graft_entry__.load_package()``; sub-packages mirror the reference's ``src/`` tree
(``models/``, ``models/modules/``, ``utils/``, ``pl_modules/``) and ``csrc/`` holds the HIP kernels + C ABI.
"""
from . import config, synthetic, utils  # noqa: F401

__all__ = ["config", "synthetic", "utils"]
