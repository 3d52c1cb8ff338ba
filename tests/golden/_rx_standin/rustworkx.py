"""Stand-in for the three rustworkx calls the reference's clustering makes (ks_clustering.py:42-44, 87-94,
107-116, 119): PyGraph.add_nodes_from / add_edges_from and connected_components.  rustworkx is not installed in
the build container; this lets tests/golden/make_cluster_golden.py run the REFERENCE's own Clusters class
unmodified.  What it cannot reproduce is rustworkx's ordering of the components and of the nodes inside a
component, so the golden files pin the components as sets.  Test infrastructure only."""


class PyGraph:
    def __init__(self):
        self._nodes = []
        self._adj = []

    def add_nodes_from(self, payloads):
        first = len(self._nodes)
        for p in payloads:
            self._nodes.append(p)
            self._adj.append([])
        return list(range(first, len(self._nodes)))

    def add_edges_from(self, edges):
        out = []
        for a, b, _w in edges:
            if not (0 <= a < len(self._nodes) and 0 <= b < len(self._nodes)):
                raise IndexError("node index out of range")
            self._adj[a].append(b)
            self._adj[b].append(a)
            out.append(len(out))
        return out


def connected_components(graph):
    seen = [False] * len(graph._nodes)
    comps = []
    for s in range(len(graph._nodes)):
        if seen[s]:
            continue
        seen[s] = True
        comp, stack = set(), [s]
        while stack:
            v = stack.pop()
            comp.add(v)
            for w in graph._adj[v]:
                if not seen[w]:
                    seen[w] = True
                    stack.append(w)
        comps.append(comp)
    return comps
