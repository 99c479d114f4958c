#!/usr/bin/env python3
"""Shape of a hipGraph dumped by CUDAGraph.debug_dump (DOT): nodes, edges, roots, leaves, nodes with more than one successor /
predecessor -- is the captured update step ONE chain, or does it fork (and where do the forks join)?  usage: dot_graph_shape.py file.dot"""
import re
import sys
from collections import defaultdict

txt = open(sys.argv[1]).read()
edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
labels = dict(re.findall(r'"?([\w.]+)"?\s*\[[^\]]*label="([^"]*)"', txt))
succ, pred, nodes = defaultdict(set), defaultdict(set), set(labels)
for a, b in edges:
    succ[a].add(b), pred[b].add(a), nodes.update((a, b))
roots = [n for n in nodes if not pred[n]]
leaves = [n for n in nodes if not succ[n]]
forks = [n for n in nodes if len(succ[n]) > 1]
joins = [n for n in nodes if len(pred[n]) > 1]
print("nodes %d edges %d roots %d leaves %d forks %d joins %d" % (len(nodes), len(edges), len(roots), len(leaves), len(forks), len(joins)))
short = lambda n: (labels.get(n, "")[:110]).replace("\\n", " | ")
for name, lst in (("root", roots), ("leaf", leaves), ("fork", forks), ("join", joins)):
    for n in lst[:12]:
        print("%s %s  ->%d <-%d  %s" % (name, n, len(succ[n]), len(pred[n]), short(n)))
