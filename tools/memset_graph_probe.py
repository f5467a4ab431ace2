"""Round-2 open question (VERDICT 7a / ADVICE): the L=200 d=128 step went non-finite after 12-116 replays when its zero
fills were hipMemsetAsync nodes.  Was the memset node ordered before the kernel that accumulates into its buffer in THIS
capture?  Captures the step (ACATTN_ZERO_MEMSET=1: memset nodes; default: fill kernels), walks the captured hipGraph with
hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType / hipGraphKernelNodeGetParams / hipGraphMemsetNodeGetParams and
reports, for every memset (or zero_fill kernel) node, its predecessors and successors and which later kernel nodes take a
pointer inside the filled range as an argument.

    ACATTN_ZERO_MEMSET=1 python tools/memset_graph_probe.py gpurun_out/r3/graph_memset
    python tools/memset_graph_probe.py gpurun_out/r3/graph_kernel
"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ac_tsr_amd as A

out = sys.argv[1]
sys.argv = [sys.argv[0], "--config", "cfg4", "--batch", os.environ.get("PROBE_BATCH", "64")]
a = bench.parse()
device = torch.device("cuda:0")
torch.manual_seed(42)
model = getattr(A, a.model)(A.DictConfig(bench.model_config(a)), A.ItemCount(a.items)).to(device)
trainer = A.AttackSASRecTrainer(A.DictConfig(learner='adam', learning_rate=1e-4), model)
model.train()
gen = torch.Generator().manual_seed(1000)
pool = [bench.synthetic_batch(a.batch, a.seq_len, a.items, gen, device) for _ in range(2)]
trainer.enable_graph(pool[0], debug_dump=os.path.abspath(out + ".dot"))
torch.cuda.synchronize()
print("dot file written:", os.path.exists(out + ".dot"))
raw = getattr(trainer, "_raw_graph", None)
if raw is None:
    sys.exit("no raw graph handle (torch.cuda.CUDAGraph.raw_cuda_graph missing)")
hip = C.CDLL("libamdhip64.so")
graph = C.c_void_p(raw)
n = C.c_size_t(0)
assert hip.hipGraphGetNodes(graph, None, C.byref(n)) == 0
nodes = (C.c_void_p * n.value)()
assert hip.hipGraphGetNodes(graph, nodes, C.byref(n)) == 0
ne = C.c_size_t(0)
assert hip.hipGraphGetEdges(graph, None, None, C.byref(ne)) == 0
src, dst = (C.c_void_p * ne.value)(), (C.c_void_p * ne.value)()
assert hip.hipGraphGetEdges(graph, src, dst, C.byref(ne)) == 0
print(f"{n.value} nodes, {ne.value} edges")
TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}


class KernelParams(C.Structure):  # hipKernelNodeParams
    _fields_ = [("blockDim", C.c_uint * 3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", C.c_uint * 3),
                ("kernelParams", C.POINTER(C.c_void_p)), ("sharedMemBytes", C.c_uint)]


class MemsetParams(C.Structure):  # hipMemsetParams
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint), ("height", C.c_size_t), ("pitch", C.c_size_t),
                ("value", C.c_uint), ("width", C.c_size_t)]


hip.hipKernelNameRefByPtr.restype = C.c_char_p
hip.hipKernelNameRefByPtr.argtypes = [C.c_void_p, C.c_void_p]
info = {}
order = []
for k in range(n.value):
    h = nodes[k]
    t = C.c_int(-1)
    hip.hipGraphNodeGetType(C.c_void_p(h), C.byref(t))
    d = {"type": TYPES.get(t.value, str(t.value)), "name": "", "idx": k}
    if t.value == 0:
        kp = KernelParams()
        if hip.hipGraphKernelNodeGetParams(C.c_void_p(h), C.byref(kp)) == 0:
            nm = hip.hipKernelNameRefByPtr(kp.func, None)
            d["name"] = (nm.decode() if nm else "?")[:90]
            d["grid"] = tuple(kp.gridDim)
    elif t.value == 2:
        mp = MemsetParams()
        if hip.hipGraphMemsetNodeGetParams(C.c_void_p(h), C.byref(mp)) == 0:
            d.update(dst=mp.dst, bytes=mp.width * mp.elementSize * max(mp.height, 1), value=mp.value)
            d["name"] = f"memset dst=0x{mp.dst or 0:x} bytes={d['bytes']}"
    info[h] = d
    order.append(h)
succ, pred = {}, {}
for u, v in zip(src, dst):
    succ.setdefault(u, []).append(v)
    pred.setdefault(v, []).append(u)
import collections
print("node types:", dict(collections.Counter(d["type"] for d in info.values())))
hits = [h for h in order if info[h]["type"] == "memset" or "zero_fill" in info[h]["name"]]
print(f"{len(hits)} zero-fill nodes")
for h in hits:
    d = info[h]
    print(f"NODE #{d['idx']} {d['type']} {d['name']}")
    for p in pred.get(h, []):
        print(f"   pred: #{info[p]['idx']} {info[p]['type']} {info[p]['name']}")
    for q in succ.get(h, []):
        print(f"   succ: #{info[q]['idx']} {info[q]['type']} {info[q]['name']}")
    if not succ.get(h):
        print("   succ: NONE  <-- no dependency edge leaves this node")
    # reachability: which nodes are ordered after this one (transitively)?
    seen, stack = set(), list(succ.get(h, []))
    while stack:
        x = stack.pop()
        if x in seen:
            continue
        seen.add(x)
        stack.extend(succ.get(x, []))
    after = [info[x] for x in seen if info[x]["type"] == "kernel"]
    names = collections.Counter(x["name"].split("(")[0][-60:] for x in after)
    acc = [x for x in after if any(s in x["name"] for s in ("bwd_row", "bwd_key", "ce_bwd"))]
    print(f"   ordered before {len(seen)} nodes ({len(after)} kernels); accumulating kernels among them: "
          f"{sorted(set(x['name'].split('(')[0][-50:] for x in acc))}")
# and the converse: accumulating kernels that are NOT ordered after any zero-fill node
