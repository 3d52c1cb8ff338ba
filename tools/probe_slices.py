import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kspider_amd import engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 28284
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sk = synth.generate("C2", n_sources=N)
dev = torch.device("cuda", 0)
keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
e = engine.Engine(0)
for it in range(2):
    t = time.time(); e.build_blocks(keys_d.data_ptr(), sk.offsets); torch.cuda.synchronize(); t_full = time.time() - t
nb = e.stats()["n_blocks"]
sizes = []; rows = []
# the ranks' MIN all-reduce of the source labels, emulated on one GPU
labels = torch.full((N,), 2**31 - 1, dtype=torch.int32, device=dev); lab = torch.empty_like(labels)
for p in range(G):
    e.build_slice(keys_d.data_ptr(), sk.offsets, p, G); e.slice_labels(lab.data_ptr()); labels = torch.minimum(labels, lab)
for p in range(G):
    for it in range(2 if p == 0 else 1):
        torch.cuda.synchronize(); t = time.time(); e.build_slice(keys_d.data_ptr(), sk.offsets, p, G); e.slice_finish(labels.data_ptr()); torch.cuda.synchronize(); t_slice = time.time() - t
    sz = e.slice_sizes(); sizes.append(sz)
    L, nbig = int(sz[0]), int(sz[2])
    r = [torch.zeros(L, dtype=torch.int32, device=dev), torch.zeros(L, dtype=torch.int32, device=dev), None,
         torch.zeros(nb + 1, dtype=torch.int32, device=dev), torch.zeros(nb + 1, dtype=torch.int32, device=dev),
         torch.zeros(max(4, 4 * nbig), dtype=torch.int32, device=dev)]
    t = time.time(); e.slice_export(r[0].data_ptr(), r[1].data_ptr(), 0, r[3].data_ptr(), r[4].data_ptr(), r[5].data_ptr()); t_exp = time.time() - t
    rows.append(r)
    if p == 0: print(f"full build {t_full*1e3:.2f} ms; slice build {t_slice*1e3:.2f} ms; export {t_exp*1e3:.2f} ms; L={L} nbig={nbig}", flush=True)
sizes = np.concatenate(sizes)
ls = int(sizes[0::4].max()); bs = max(1, int(sizes[2::4].max()))
def stack(i, cols):
    out = torch.zeros((G, cols), dtype=torch.int32, device=dev)
    for p in range(G): out[p, :rows[p][i].numel()] = rows[p][i][:cols]
    return out
brk, info, raw, pos, big = stack(0, ls), stack(1, ls), stack(3, nb + 1), stack(4, nb + 1), stack(5, 4 * bs)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    e.assemble(sizes, brk.data_ptr(), info.data_ptr(), 0, ls, raw.data_ptr(), pos.data_ptr(), big.data_ptr(), bs)
    torch.cuda.synchronize(); t_asm = time.time() - t
print(f"assemble {t_asm*1e3:.2f} ms; gathered bytes per rank {(brk.numel()+info.numel())*4*(G-1)/G/1e6:.0f} MB; block keys {e.stats()['n_block_keys']}")
T = e.num_tiles; cuts = e.balanced_cuts(G); print('cuts', cuts, 'active', e.stats()['n_active_tiles']); t0, t1 = cuts[G // 2], cuts[G // 2 + 1]
cap = 1 << 24; de = torch.empty((cap, 16), dtype=torch.uint8, device=dev)
for it in range(2):
    cnt = e.join(t0, t1, de.data_ptr(), cap); st = e.stats()
print(f"join of 1/{G} of the tiles: {st['ms_join']:.2f} ms, edges {cnt}")
