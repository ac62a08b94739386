"""Structure of the captured training step (DataParallelTrainer(graph=True)): dumps the hipGraph with hipGraphDebugDotPrint
(FU_GRAPH_DOT) and counts nodes, edges, forks and joins -- is the weight-gradient side chain still a second branch in the graph?
    python3 tools/graph_dot.py [out.dot]        (FU_NO_SIDE_STREAM=1 for the one-chain capture)"""
import os, re, sys, collections
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/fu_step.dot"
os.environ["FU_GRAPH_DOT"] = out
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from floodplanet_code_amd.unet import HipUNet
from floodplanet_code_amd.distributed import DataParallelTrainer
dev = torch.device("cuda:0")
net = HipUNet(8, 3, bilinear=True, base_channels=64, precision="bf16").to(dev).train()
tr = DataParallelTrainer(net, lr=1e-4, world_size=1, rank=0, graph=True)
x = torch.rand(16, 8, 256, 256, device=dev); t = (torch.rand(16, 256, 256, device=dev) > 0.5).long()
for _ in range(3):
    tr.step(x, t, 0)
torch.cuda.synchronize()
txt = open(out).read()
edges = re.findall(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', txt)
nodes = set(re.findall(r'^\s*"?([\w.]+)"?\s*\[', txt, flags=re.M)) | {a for a, _ in edges} | {b for _, b in edges}
outd, ind = collections.Counter(a for a, _ in edges), collections.Counter(b for _, b in edges)
kern = len(re.findall(r'KERNEL|kernel', txt))
print(f"{out}: {len(nodes)} nodes, {len(edges)} edges, ~{kern} kernel mentions")
print(f"forks (nodes with > 1 successor): {sum(1 for n in nodes if outd[n] > 1)}, joins (> 1 predecessor): {sum(1 for n in nodes if ind[n] > 1)}")
print(f"roots: {sum(1 for n in nodes if ind[n] == 0)}, leaves: {sum(1 for n in nodes if outd[n] == 0)}")
# longest path (critical chain length in nodes) against the node count: a linear chain has length == nodes
succ = collections.defaultdict(list)
for a, b in edges: succ[a].append(b)
memo = {}
def depth(n):
    st = [(n, 0)]
    while st:
        v, i = st[-1]
        if v in memo: st.pop(); continue
        if i < len(succ[v]):
            st[-1] = (v, i + 1)
            if succ[v][i] not in memo: st.append((succ[v][i], 0))
        else:
            memo[v] = 1 + max((memo[w] for w in succ[v]), default=0); st.pop()
    return memo[n]
print(f"longest dependency chain: {max(depth(n) for n in nodes)} nodes")
