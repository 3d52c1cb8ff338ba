"""C2-shaped sketches plus keys held by a random 10 % of all sources (conserved k-mers): how many tiles stay
inactive with / without ignoring such keys in the label pass.   python tools/hub_experiment.py [N] [hubs]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sk = synth.generate("C2", n_sources=N)
rng = np.random.default_rng(7)
runs = [sk.run(s) for s in range(N)]
hubs = (np.uint64(1) << np.uint64(53)) + np.arange(H, dtype=np.uint64) * np.uint64(977)
for h in hubs:
    for s in rng.choice(N, size=N // 10, replace=False):
        runs[s] = np.append(runs[s], h)
sk2 = synth.from_runs(runs)
dk = engine.DeviceBuffer.from_numpy(sk2.keys)
e = engine.Engine(0)
for it in range(3):
    e.build_blocks(dk.ptr.value, sk2.offsets)
    cap = int(e.edge_bound(0, e.num_tiles)) + 1
    de = engine.DeviceBuffer(min(cap, 1 << 27) * 16)
    try:
        cnt = e.join(0, e.num_tiles, de.ptr.value, min(cap, 1 << 27))
    except engine.KspError as ex:
        cnt = -1
    st = e.stats()
print(f"label_max={os.environ.get('KSP_DEBUG_LABEL_MAX', 'default')}: words {st['n_block_keys']}, active tiles {st['n_active_tiles']} of {st['n_tiles']}, "
      f"build {st['ms_build']:.2f} ms, join {st['ms_join']:.2f} ms, edges {cnt}")
