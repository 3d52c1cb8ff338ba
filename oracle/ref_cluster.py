"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's clustering step, /root/reference/pykSpider/kSpider2/ks_clustering.py:
  load_seq_to_kmers :48-53, tsv_get_namesmap :55-61, construct_graph :63-116 (non-ANI branch :95-116, ANI branch
  :70-94), cluster_graph :118-137, cutoff scaling :156.
Connected components by union-find instead of rustworkx (absent here).  Differences from the reference that the
product shares and INTEGRATION.md states: no edge is dropped when a 10 000 000-edge batch fills (:107-113), and
the output order is canonical (components by smallest node, members ascending) where the reference has
rustworkx's set order.  Parity pin: tests/golden/clusters/* were produced by the reference's own Clusters class
(run by tests/golden/make_cluster_golden.py with a pure-Python stand-in for the rustworkx calls); this file is
checked against them as SETS of components.
"""
from __future__ import annotations


def output_path(index_prefix: str, cutoff: float) -> str:
    return index_prefix + f"_kSpider_clusters_{float(cutoff) * 100}%.tsv"     # :33-34 with :156


def clusters(index_prefix: str, dist_type: str = "max_cont", cutoff: float = 0.0):
    """-> list of components, each a list of names; components by smallest node, members ascending."""
    col = {"min_cont": 3, "avg_cont": 4, "max_cont": 5, "ani": 6}[dist_type]           # :12-17
    threshold = float(cutoff) * 100                                                        # :156
    with open(index_prefix + "_kSpider_seqToKmersNo.tsv") as f:                            # :48-53
        next(f)
        for line in f:
            seq_id, kmers = tuple(line.strip().split("\t")[1:])
            int(seq_id), int(kmers)
    names = {}
    with open(index_prefix + ".namesMap") as f:                                            # :55-61
        next(f)
        for row in f:
            row = row.strip().split()
            names[int(row[0])] = row[1]
    n = len(names)
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    ani = None
    if dist_type == "ani":
        ani = open(index_prefix + "_kSpider_pairwise.ani_col.tsv")
        next(ani)
    with open(index_prefix + "_kSpider_pairwise.tsv") as f:                                # :67-116
        next(f)
        for row in f:
            row = row.strip().split("\t")
            a, b = int(row[0]) - 1, int(row[1]) - 1
            distance = float(next(ani).strip()) * 100.0 if ani else float(row[col]) * 100
            if distance < threshold:
                continue
            if not (0 <= a < n and 0 <= b < n):
                raise IndexError("edge names a node that .namesMap does not have")
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    if ani:
        ani.close()
    comps = {}
    for v in range(n):
        comps.setdefault(find(v), []).append(v)
    return [[names[v + 1] for v in comps[r]] for r in sorted(comps)]                       # :133-137


def write_clusters(index_prefix: str, dist_type: str = "max_cont", cutoff: float = 0.0) -> str:
    out = output_path(index_prefix, cutoff)
    with open(out, "w") as f:
        for comp in clusters(index_prefix, dist_type, cutoff):
            f.write(",".join(comp) + "\n")
    return out
