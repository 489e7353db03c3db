import numpy as np, sys, os, ctypes, threading, time
sys.path.insert(0, os.getcwd())
import graphgan_amd as ga
from graphgan_amd._lib import lib
from tests.helpers import load_small
os.environ["GG_LZ_ARENA"]="40"
g, n, graph = load_small(0)
rowptr, col = ga.graph_to_csr(n, graph)
roots=np.arange(n,dtype=np.int32)
deg = (rowptr[1:] - rowptr[:-1]).astype(np.int32)
eng = ga.Engine(g["E"], g["E"]); eng.set_bias(0, g["b"]); eng.set_tree_mode(1, 6); eng.set_graph_csr(rowptr, col)
eng.build_trees(roots, device=True)
done=[False]
def run():
    got = eng.walk_sample(np.arange(n,dtype=np.int32), deg[roots], True, 5, 0, stride=20)
    print("walk ok", got["path_len"][:10], flush=True); done[0]=True
t=threading.Thread(target=run, daemon=True); t.start()
time.sleep(6)
out = np.zeros(160, np.uint64)
lib.gg_debug_words(eng._ctx, 1600, 160, out.ctypes.data_as(ctypes.c_void_p))
print("done", done[0])
print("oob", [int(x) for x in out[:12]])
print("pre/q0", [(int(x >> np.uint64(32)), int(x & np.uint64(0xffffffff))) for x in out[12:28]])
print("entry", [int(x) for x in out[40:52]])
print("calls", int(out[60]))
print("exec scan", int(out[90]), hex(int(out[91])), int(out[92]), "exec entry", int(out[94]), hex(int(out[95])))
print("first", [(int(x >> np.uint64(48)), int((x >> np.uint64(32)) & np.uint64(0xffff)), int((x >> np.uint64(16)) & np.uint64(0xffff)), int(x & np.uint64(0xffff))) for x in out[70:134]])
sys.stdout.flush(); os._exit(0)
