"""Graph / embedding readers with the reference's semantics (``src/utils.py:12-67``).

Neighbour order matters downstream (it is the BFS child order and the order of the D-step
positives), so the adjacency lists keep file order exactly like the reference's dict of lists.
"""
import numpy as np


def read_edges_from_file(filename):
    """One edge per line, whitespace separated integer ids (utils.py:50-54)."""
    edges = []
    with open(filename, "r") as f:
        for line in f:
            edges.append([int(tok) for tok in line.split()])
    return edges


def read_edges(train_filename, test_filename):
    """-> (n_node, graph) with graph[v] = neighbours of v over the TRAIN edges in file order, both
    directions (a self-loop lists v twice); nodes that only occur in the test file get an empty
    list; n_node = number of distinct ids seen (utils.py:12-47)."""
    graph = {}
    train = read_edges_from_file(train_filename)
    test = read_edges_from_file(test_filename) if test_filename != "" else []
    for a, b in train:
        graph.setdefault(a, []).append(b)
        graph.setdefault(b, []).append(a)
    for a, b in test:
        graph.setdefault(a, [])
        graph.setdefault(b, [])
    return len(graph), graph


def read_embeddings(filename, n_node, n_embed):
    """``.emb`` text -> float64 [n_node, n_embed]; the first line is a header; rows whose id is
    absent keep uniform [0, 1) draws from the global numpy RNG (utils.py:57-67)."""
    emb = np.random.rand(n_node, n_embed)
    with open(filename, "r") as f:
        f.readline()
        for line in f:
            tok = line.split()
            if tok:
                emb[int(tok[0]), :] = [float(x) for x in tok[1:]]
    return emb


def read_embeddings_bin(filename):
    """Binary side-car written next to the ``.emb`` text (``Engine.write_embeddings_bin``) -> float32 [n_node, n_emb]."""
    with open(filename, "rb") as f:
        head = f.read(20)
        if head[:4] != b"GGEB" or np.frombuffer(head, "<i4", 1, 4)[0] != 1:
            raise ValueError("%s is not a GGEB v1 file" % filename)
        n_emb = int(np.frombuffer(head, "<i4", 1, 8)[0])
        n_node = int(np.frombuffer(head, "<i8", 1, 12)[0])
        return np.fromfile(f, dtype="<f4", count=n_node * n_emb).reshape(n_node, n_emb)
