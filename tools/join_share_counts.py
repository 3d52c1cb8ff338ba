import sys, numpy as np
sys.path.insert(0, '/root/repo')
from kspider_amd import engine, synth
sk = synth.generate("C2")
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
e.build_blocks(dk.ptr.value, sk.offsets)
cap = int(min(e.edge_bound(0, e.num_tiles), 1 << 26)) * 2 + 1
de = engine.DeviceBuffer(cap * 16)
w, j = [], []
for _ in range(120):
    e.build_blocks(dk.ptr.value, sk.offsets)
    e.join(0, e.num_tiles, de.ptr.value, cap)
    st = e.stats()
    w.append(int(st["n_join_workgroups"])); j.append(st["ms_join"])
w = np.array(w); j = np.array(j)
print("workgroups: min %d median %d max %d; over 512: %d of %d" % (w.min(), np.median(w), w.max(), (w > 512).sum(), len(w)))
print("join ms: min %.3f median %.3f max %.3f" % (j.min(), np.median(j), j.max()))
for lo, hi in ((0, 480), (480, 496), (496, 512), (512, 528), (528, 9999)):
    m = (w >= lo) & (w < hi)
    if m.any(): print("  wgs [%d, %d): n=%d join median %.3f max %.3f" % (lo, hi, m.sum(), np.median(j[m]), j[m].max()))
