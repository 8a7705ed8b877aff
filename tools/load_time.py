import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import movi_amd
from tools import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
six = synth.synth_index(rows, mode=6, seed=77)
img = six.image()
os.makedirs("/tmp/idx_lt", exist_ok=True)
np.asarray(img).tofile("/tmp/idx_lt/index.movi")
t0 = time.perf_counter(); ix = movi_amd.MoveIndex.from_image(img); t1 = time.perf_counter()
print("from_image (resident host memory, one hipMemcpy), %.2f GB: %.3f s" % (img.nbytes / 1e9, t1 - t0))
ix.close()
for rep in range(3):
    t0 = time.perf_counter(); ix = movi_amd.MoveIndex.load("/tmp/idx_lt"); t1 = time.perf_counter()
    print("movi_index_load (mmap, page cache warm), run %d: %.3f s = %.1f GB/s" % (rep, t1 - t0, img.nbytes / (t1 - t0) / 1e9))
    ix.close()
t0 = time.perf_counter(); b = open("/tmp/idx_lt/index.movi", "rb").read(); t1 = time.perf_counter()
print("for scale: reading the file into a process buffer (what the loader did before): %.3f s" % (t1 - t0))
os.remove("/tmp/idx_lt/index.movi")
