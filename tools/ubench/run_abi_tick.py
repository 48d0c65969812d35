"""writes a logo file without touching the GPU, then runs tools/ubench/abi_tick on it"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import amt_synth as S
from amatsukaze_amd import binding
lib = binding.load()
data = np.ascontiguousarray(S.make_logo(256, 128)[0], np.float32)
h = lib.amtgpu_logo_from_planes(None, 256, 128, 1, 1, 1440, 1080, 1120, 64, data.ctypes.data)
p = os.path.join(tempfile.mkdtemp(), "logo.lgd")
assert h and lib.amtgpu_logo_save(None, h, p.encode(), b"x", 1)
env = dict(os.environ)
env.update(dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a))
print(subprocess.run([os.path.join(ROOT, "tools", "ubench", "abi_tick"), p], capture_output=True, text=True, env=env).stdout)
