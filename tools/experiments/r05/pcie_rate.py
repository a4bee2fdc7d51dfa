"""PCIe-inclusive rate of the batch entry points with HOST in/out buffers (images uploaded, features downloaded by the call)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rgbd_pl_slam_amd import ORBextractor, LineSegment
from rgbd_pl_slam_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imgs = synth_batch(100, 16)
imgs = np.ascontiguousarray(np.concatenate([imgs] * ((B + 15) // 16))[:B])
ext = ORBextractor(nfeatures=1000, max_width=640, max_height=480, max_batch=B)
ls = LineSegment(nlines=100, max_width=640, max_height=480, max_batch=B)
for _ in range(2): ext.extract_batch(imgs); ls.extract_batch(imgs)
t = time.perf_counter(); K = 3
for _ in range(K): ext.extract_batch(imgs)
dt_o = (time.perf_counter() - t) / K
t = time.perf_counter()
for _ in range(K): ls.extract_batch(imgs)
dt_l = (time.perf_counter() - t) / K
print("host in/out, B=%d: ORB %.1f ms (%.0f fps), lines %.1f ms (%.0f fps), sequential both %.0f fps" % (B, dt_o * 1e3, B / dt_o, dt_l * 1e3, B / dt_l, B / (dt_o + dt_l)))
