"""Test-only stand-in for the slice of ``graph_tool`` that the VStrains reference touches.

graph-tool (C++/Boost, conda-only) cannot be installed in the build container, so the reference's
graph stages are run behind this module when ``tests/golden/make_graph_golden.py`` produces
fixtures.  It is NOT product code and nothing under ``vstrains_amd/`` imports it.

It models graph-tool's ``adj_list`` as remembered from its ``graph_adjacency.hh`` (one vector
per vertex, out-edges in ``[0, n_out)``, in-edges behind them):

* ``add_edge(s, t)``: the new out-edge goes to slot ``n_out`` of ``s``; if an in-edge lives there
  it is moved to the back of the vector.  The in-edge is appended to ``t``'s vector.
* ``remove_edge``: erased in place (order kept); the index goes on a FIFO free list that
  ``add_edge`` reuses; property values at a reused index are NOT reset.
* ``vertices()``: index order.  ``edges()``: vertex-major, out-list order.
* ``copy()``: vertices first, then edges re-added in ``edges()`` order with fresh indices.
* ``hash(Vertex) == index``; ``hash(Edge) == edge index``.

``GT_STANDIN_INEDGE=plain`` switches to plain insertion order for in-edges, ``=lifo`` / ``=swappop`` vary the
free-list and the remove_edge rule (below), ``=inrev`` / ``=plain_outrev`` reverse the in- or the out-entries; the
fixture maker runs every case under all of them and records which outputs agree ("parity unpinned at the graph-tool boundary").
"""
import os
from collections import deque

MODEL = os.environ.get("GT_STANDIN_INEDGE", "rotate")
INEDGE_ROTATION = MODEL in ("rotate", "lifo", "swappop")  # (lifo / swappop vary ONE other rule each, on top of the default)
# Two further models vary the OTHER recalled rules (fixtures count as "adjacency invariant" only if they survive all of
# rotate / plain / lifo / swappop): "lifo" reuses the index of the edge removed LAST (the free list as a stack instead of
# a queue: changes hash(Edge), hence the order of the sets of edges the reference builds, IO.py:207, and which stale
# property value a new edge meets); "swappop" lets remove_edge move the last entry of the segment into the hole instead
# of closing it (order not kept).
LIFO_REUSE = MODEL == "lifo"
SWAP_POP = MODEL == "swappop"
# Reported only (case.json lists what changes, nothing is required to survive them): in-entries in REVERSE insertion
# order, and out-entries in reverse insertion order.  Either changes already which way gfa_to_graph's orientation walk
# goes (IO.py:137-229 iterates all_edges()), so hardly any file stays the same: these are not small doubts about a rule
# but different containers.
IN_REVERSED = MODEL == "inrev"
OUT_REVERSED = MODEL == "plain_outrev"


class _PropMap:
    def __init__(self, default_factory):
        self._d = []
        self._mk = default_factory

    def _grow(self, i):
        while len(self._d) <= i:
            self._d.append(self._mk())

    def __getitem__(self, key):
        i = key._pm_index()
        self._grow(i)
        return self._d[i]

    def __setitem__(self, key, val):
        i = key._pm_index()
        self._grow(i)
        self._d[i] = self._conv(val)

    def _conv(self, v):
        return v

    def copy_from(self, other):
        self._d = list(other._d)


def _typed(type_name, val):
    if type_name == "string":
        dflt = "" if val is None else str(val)
        pm = _PropMap(lambda: dflt)
        pm._conv = str
    elif type_name == "double":
        dflt = 0.0 if val is None else float(val)
        pm = _PropMap(lambda: dflt)
        pm._conv = float
    elif type_name in ("int", "int16_t", "int32_t", "int64_t"):
        dflt = 0 if val is None else int(val)
        pm = _PropMap(lambda: dflt)
        pm._conv = int
    else:
        raise NotImplementedError(type_name)
    pm._type = type_name
    pm._default = val
    return pm


class _Props:
    """``g.vp`` / ``g.ep``: attribute-style container of property maps."""

    def __init__(self):
        object.__setattr__(self, "_maps", {})

    def __setattr__(self, k, v):
        self._maps[k] = v

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_maps")[k]
        except KeyError:
            raise AttributeError(k)


class Vertex:
    __slots__ = ("_g", "_i")

    def __init__(self, g, i):
        self._g = g
        self._i = i

    def _pm_index(self):
        return self._i

    def __int__(self):
        return self._i

    def __index__(self):
        return self._i

    def __hash__(self):
        return hash(self._i)

    def __eq__(self, o):
        return isinstance(o, Vertex) and o._i == self._i

    def __ne__(self, o):
        return not self.__eq__(o)

    def __lt__(self, o):
        return self._i < int(o)

    def __gt__(self, o):
        return self._i > int(o)

    def __le__(self, o):
        return self._i <= int(o)

    def __ge__(self, o):
        return self._i >= int(o)

    def __repr__(self):
        return "<Vertex %d>" % self._i

    def out_degree(self):
        return self._g._nout[self._i]

    def in_degree(self):
        return len(self._g._adj[self._i]) - self._g._nout[self._i]

    def out_edges(self):
        g = self._g
        for (t, idx) in list(g._adj[self._i][: g._nout[self._i]]):
            yield Edge(g, self._i, t, idx)

    def in_edges(self):
        g = self._g
        for (s, idx) in list(g._adj[self._i][g._nout[self._i]:]):
            yield Edge(g, s, self._i, idx)

    def all_edges(self):
        yield from self.out_edges()
        yield from self.in_edges()

    def out_neighbors(self):
        for e in self.out_edges():
            yield e.target()

    def in_neighbors(self):
        for e in self.in_edges():
            yield e.source()

    def all_neighbors(self):
        yield from self.out_neighbors()
        yield from self.in_neighbors()


class Edge:
    __slots__ = ("_g", "_s", "_t", "_idx")

    def __init__(self, g, s, t, idx):
        self._g = g
        self._s = s
        self._t = t
        self._idx = idx

    def _pm_index(self):
        return self._idx

    def source(self):
        return Vertex(self._g, self._s)

    def target(self):
        return Vertex(self._g, self._t)

    def __hash__(self):
        return hash(self._idx)

    def __eq__(self, o):
        return isinstance(o, Edge) and (o._s, o._t, o._idx) == (self._s, self._t, self._idx)

    def __ne__(self, o):
        return not self.__eq__(o)

    def __repr__(self):
        return "<Edge %d->%d #%d>" % (self._s, self._t, self._idx)


class Graph:
    def __init__(self, g=None, directed=True):
        self.vp = _Props()
        self.ep = _Props()
        self._adj = []   # per vertex: list of (neighbour, edge idx); out part then in part
        self._nout = []
        self._free = deque()
        self._next_eidx = 0
        self._ne = 0
        if g is not None:
            self._copy_from(g)

    # --- property maps
    def new_vertex_property(self, type_name, val=None):
        return _typed(type_name, val)

    def new_edge_property(self, type_name, val=None):
        return _typed(type_name, val)

    # --- structure
    def add_vertex(self):
        self._adj.append([])
        self._nout.append(0)
        return Vertex(self, len(self._adj) - 1)

    def add_edge(self, source, target):
        s, t = int(source), int(target)
        if self._free:
            idx = self._free.pop() if LIFO_REUSE else self._free.popleft()
        else:
            idx = self._next_eidx
            self._next_eidx += 1
        ses = self._adj[s]
        pos = self._nout[s]
        if OUT_REVERSED:
            ses.insert(0, (t, idx))
        elif pos < len(ses):
            if INEDGE_ROTATION:
                ses.append(ses[pos])
                ses[pos] = (t, idx)
            else:
                ses.insert(pos, (t, idx))
        else:
            ses.append((t, idx))
        self._nout[s] = pos + 1
        if IN_REVERSED:
            self._adj[t].insert(self._nout[t], (s, idx))
        else:
            self._adj[t].append((s, idx))
        self._ne += 1
        return Edge(self, s, t, idx)

    def remove_edge(self, e):
        s, t, idx = e._s, e._t, e._idx
        ses = self._adj[s]
        for i in range(self._nout[s]):
            if ses[i] == (t, idx):
                if SWAP_POP:  # the last out-entry fills the hole, then the last in-entry fills ITS place
                    last_out = self._nout[s] - 1
                    ses[i] = ses[last_out]
                    ses[last_out] = ses[-1]
                    ses.pop()
                else:
                    del ses[i]
                break
        else:
            raise ValueError("edge not found")
        self._nout[s] -= 1
        tes = self._adj[t]
        for i in range(self._nout[t], len(tes)):
            if tes[i] == (s, idx):
                if SWAP_POP:
                    tes[i] = tes[-1]
                    tes.pop()
                else:
                    del tes[i]
                break
        else:
            raise ValueError("edge not found")
        self._free.append(idx)
        self._ne -= 1

    def vertices(self):
        for i in range(len(self._adj)):
            yield Vertex(self, i)

    def vertex(self, i):
        return Vertex(self, int(i))

    def edges(self):
        for i in range(len(self._adj)):
            for (t, idx) in list(self._adj[i][: self._nout[i]]):
                yield Edge(self, i, t, idx)

    def edge(self, s, t):
        s, t = int(s), int(t)
        for (tt, idx) in self._adj[s][: self._nout[s]]:
            if tt == t:
                return Edge(self, s, t, idx)
        return None

    def num_vertices(self):
        return len(self._adj)

    def num_edges(self):
        return self._ne

    def copy(self):
        return Graph(self)

    def _copy_from(self, g):
        for _ in range(g.num_vertices()):
            self.add_vertex()
        for name, pm in g.vp._maps.items():
            npm = _typed(pm._type, pm._default)
            npm.copy_from(pm)
            self.vp._maps[name] = npm
        for name, pm in g.ep._maps.items():
            self.ep._maps[name] = _typed(pm._type, pm._default)
        for e in g.edges():
            ne = self.add_edge(e._s, e._t)
            for name, pm in g.ep._maps.items():
                self.ep._maps[name][ne] = pm[e]
