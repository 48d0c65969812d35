"""amatsukaze_amd -- MI355X-native logo / CM / KFM analysis hot path of nekopanda/Amatsukaze.

Product code only: HIP kernels + C ABI (csrc/, libamt_gpu.so) and the Python mirror of the reference's
filter interface (api.py).  Nothing here imports the CPU oracle.
"""
from .api import (AMTAnalyzeLogo, AMTEraseLogo, AmtError, AmtsFile, Context, DeviceClip, FrameStats, Logo, LogoFrame,
                  LogoScan, ScanLogo, ScanLogoFile, weave_fields)

__all__ = ["AMTAnalyzeLogo", "AMTEraseLogo", "AmtError", "AmtsFile", "Context", "DeviceClip", "FrameStats", "Logo", "LogoFrame",
           "LogoScan", "ScanLogo", "ScanLogoFile", "weave_fields"]
