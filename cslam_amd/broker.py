"""Broker — which endpoint of every selected edge ships its local features; host mirror of
cslam/broker.py:8-129.

The reference delegates to networkx (maximum bipartite matching by Hopcroft-Karp, König's
theorem for the cover, the local-ratio 2-approximation otherwise).  The graphs are tiny (at most
`frontend.inter_robot_loop_closure_budget` edges), so this stays host Python — but without the
networkx dependency, and reproducing networkx's traversal orders (dict/set insertion orders,
component copies) so the SAME vertices are chosen: tests/golden/broker_ref.json pins it against
the reference.  networkx functions restated: `connected_components`, `Graph.subgraph().copy()`,
`bipartite.color/sets`, `bipartite.hopcroft_karp_matching`, `bipartite.to_vertex_cover`,
`approximation.min_weighted_vertex_cover` (networkx 3.4.2).
"""
import collections

import numpy as np


class _Graph(object):
    """Insertion-ordered undirected simple graph (the part of nx.Graph the broker needs)."""

    def __init__(self):
        self.adj = {}

    def __contains__(self, v):
        return v in self.adj

    def __iter__(self):
        return iter(self.adj)

    def __len__(self):
        return len(self.adj)

    def add_node(self, v):
        if v not in self.adj:
            self.adj[v] = {}

    def add_edge(self, u, v):
        self.add_node(u)
        self.add_node(v)
        self.adj[u][v] = True
        self.adj[v][u] = True

    def edges(self):
        seen = set()
        for u, nbrs in self.adj.items():
            for v in nbrs:
                if v not in seen:
                    yield (u, v)
            seen.add(u)

    def connected_components(self):
        seen = set()
        n = len(self.adj)
        for source in self.adj:
            if source in seen:
                continue
            comp = {source}
            level = [source]
            while level:
                nxt = []
                for v in level:
                    for w in self.adj[v]:
                        if w not in comp:
                            comp.add(w)
                            nxt.append(w)
                    if len(comp) == n:
                        break
                else:
                    level = nxt
                    continue
                break
            seen.update(comp)
            yield comp

    def component_copy(self, comp):
        """`G.subgraph(comp).copy()` (networkx 3.4.2): the view's filter holds `set(comp)`; its
        nodes iterate that SET when it is less than half of the graph and the graph order
        otherwise; neighbours always keep the adjacency order."""
        nodes = set(v for v in comp)  # show_nodes(...) re-builds the set from an iteration
        if 2 * len(nodes) < len(self.adj):
            order = [v for v in nodes if v in self.adj]
        else:
            order = [v for v in self.adj if v in nodes]
        g = _Graph()
        for v in order:
            g.add_node(v)
        for u in order:
            for w in self.adj[u]:
                if w in nodes:
                    g.add_edge(u, w)
        return g


def _bipartite_sets(g):
    color = {}
    for n in g:
        if n in color or len(g.adj[n]) == 0:
            continue
        queue = [n]
        color[n] = 1
        while queue:
            v = queue.pop()
            c = 1 - color[v]
            for w in g.adj[v]:
                if w not in color:
                    color[w] = c
                    queue.append(w)
    for n in g:
        if len(g.adj[n]) == 0:
            color[n] = 0
    left = {n for n, top in color.items() if top}
    right = {n for n, top in color.items() if not top}
    return left, right


def _hopcroft_karp(g):
    INF = float("inf")
    left, right = _bipartite_sets(g)
    lm = {v: None for v in left}
    rm = {v: None for v in right}
    dist = {}
    queue = collections.deque()

    def bfs():
        for v in left:
            if lm[v] is None:
                dist[v] = 0
                queue.append(v)
            else:
                dist[v] = INF
        dist[None] = INF
        while queue:
            v = queue.popleft()
            if dist[v] < dist[None]:
                for u in g.adj[v]:
                    if dist[rm[u]] is INF:
                        dist[rm[u]] = dist[v] + 1
                        queue.append(rm[u])
        return dist[None] is not INF

    def dfs(v):
        if v is not None:
            for u in g.adj[v]:
                if dist[rm[u]] == dist[v] + 1:
                    if dfs(rm[u]):
                        rm[u] = v
                        lm[v] = u
                        return True
            dist[v] = INF
            return False
        return True

    while bfs():
        for v in left:
            if lm[v] is None:
                dfs(v)
    matching = {k: v for k, v in lm.items() if v is not None}
    matching.update({k: v for k, v in rm.items() if v is not None})
    return matching, left, right


def _alternating_reach(g, v, matched, unmatched, targets):
    def walk(u, along_matched):
        visited = set()
        stack = [(u, iter(g.adj[u]), 0 if along_matched else 1)]
        while stack:
            parent, children, depth = stack[-1]
            valid = matched if depth % 2 else unmatched
            try:
                child = next(children)
                if child not in visited:
                    if (parent, child) in valid or (child, parent) in valid:
                        if child in targets:
                            return True
                        visited.add(child)
                        stack.append((child, iter(g.adj[child]), depth + 1))
            except StopIteration:
                stack.pop()
        return False

    return walk(v, True) or walk(v, False)


def _koenig_cover(g):
    matching, left, right = _hopcroft_karp(g)
    pairs = {frozenset((u, v)) for u, v in matching.items()}
    matched = {tuple(p) for p in pairs}
    unmatched = {(u, v) for (u, v) in g.edges() if frozenset((u, v)) not in pairs}
    targets = (set(g) - set(matching)) & left
    z = {v for v in g if v in targets or _alternating_reach(g, v, matched, unmatched, targets)}
    return (left - z) | (right & z)


def _local_ratio_cover(g):
    cost = {v: 1 for v in g}
    cover = set()
    for u, v in g.edges():
        if u in cover or v in cover:
            continue
        if cost[u] <= cost[v]:
            cover.add(u)
            cost[v] -= cost[u]
        else:
            cover.add(v)
            cost[u] -= cost[v]
    return cover


class Broker(object):
    """The broker decides which vertices in the matching graph are to be shared between the
    robots."""

    def __init__(self, edges, robots_involved):
        """
        Args:
            edges (list(EdgeInterRobot)): selected inter-robot edges
            robots_involved (list(int)): ids of the robots taking part in the exchange
        """
        self.edges = edges
        with_edges = set()
        for e in self.edges:
            if e.robot0_id in robots_involved:
                with_edges.add(e.robot0_id)
            if e.robot1_id in robots_involved:
                with_edges.add(e.robot1_id)
        with_edges = list(with_edges)
        self.is_multi_robot_graph = len(with_edges) >= 2
        if not self.is_multi_robot_graph:
            return
        self.is_bipartite = len(with_edges) == 2
        self.matching_graph = _Graph()
        for e in self.edges:
            ends = [(e.robot0_id, e.robot0_keyframe_id), (e.robot1_id, e.robot1_keyframe_id)]
            for vertex in ends:
                if vertex not in self.matching_graph:
                    if self.is_bipartite:
                        if vertex[0] == with_edges[0] or vertex[0] == with_edges[1]:
                            self.matching_graph.add_node(vertex)
                    elif vertex[0] in with_edges:
                        self.matching_graph.add_node(vertex)
            if ends[0][0] in with_edges and ends[1][0] in with_edges:
                self.matching_graph.add_edge(ends[0], ends[1])

    def brokerage(self, use_vertex_cover):
        """Vertices to transmit: list of sets of (robot_id, keyframe_id), one per component."""
        if not self.is_multi_robot_graph:
            return []
        return self.vertex_cover() if use_vertex_cover else self.simple_dialog()

    def vertex_cover(self):
        """Minimum vertex cover per connected component (bipartite: maximum matching + König;
        otherwise the greedy local-ratio 2-approximation) — broker.py:84-110."""
        covers = []
        comps = [self.matching_graph.component_copy(c)
                 for c in self.matching_graph.connected_components()]
        for g in comps:
            covers.append(_koenig_cover(g) if self.is_bipartite else _local_ratio_cover(g))
        return covers

    def simple_dialog(self):
        """For each edge transmit one endpoint at random unless one is already chosen
        (broker.py:112-129; global numpy RNG like the reference)."""
        chosen = set()
        for e in self.edges:
            ends = [(e.robot0_id, e.robot0_keyframe_id), (e.robot1_id, e.robot1_keyframe_id)]
            if ends[0] not in chosen and ends[1] not in chosen:
                chosen.add(ends[np.random.randint(2)])
        return [chosen]
