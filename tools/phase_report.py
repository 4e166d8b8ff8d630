"""dev (make DEV=1 OUT_DIR=../lib_dev; LGNN_LIB_DIR=lib_dev LGNN_PHASE_REPORT=1): per-phase cycle counts of paths_fused_kernel
over a few arxiv-shaped fits.  usage: python tools/phase_report.py [workload] [fits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import laplace_gnn_amd as lg

wl = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, ei, X, tri, try_ = bench.make_workload(wl, "cuda")
torch.manual_seed(0)
cls = lg.GraphSAGE if w.get("kind") == "sage" else lg.GCN
model = cls(w["F"], w["H"], w["C"], 2, X, ei, symmetric=True).to("cuda")
loader = lg.TensorBatchLoader(tri.cuda(), try_.cuda(), batch_size=w["batch"])
la = lg.KronLaplace(model, "classification")
for _ in range(fits):
    la.fit(loader)
torch.cuda.synchronize()
model.engine.close()
