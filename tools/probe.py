import time, sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
t=time.time(); sk = synth.generate("C2", n_sources=N); print("gen", time.time()-t, sk.n_sources, int(sk.offsets[-1]), flush=True)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
P = N*(N-1)//2
cap = 1<<26
de = engine.DeviceBuffer(cap*16)
for it in range(4):
    t=time.time()
    e.build_blocks(dk.ptr.value, sk.offsets)
    t1=time.time()
    cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
    t2=time.time()
    st = e.stats()
    print(f"it{it} build {1e3*(t1-t):.2f} ms (ev {st['ms_build']:.2f}) join {1e3*(t2-t1):.2f} ms (ev {st['ms_join']:.2f}) edges {cnt} pairs/s {P/(t2-t):.3e} stream GB {st['last_stream_bytes']/1e9:.2f} -> {st['last_stream_bytes']/st['ms_join']/1e6:.1f} GB/s; blockkeys {st['n_block_keys']} keybits {st['key_bits']}", flush=True)
