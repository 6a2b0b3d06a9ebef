"""Times kgat_eval_recall_ndcg_f32 at the amazon-book shape (developer tool)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dgl_kgat_amd import metrics, ops
dev = torch.device("cuda:0")
n_u, n_i, F, K = int(os.environ.get("EVAL_PROBE_USERS", 70679)), int(os.environ.get("EVAL_PROBE_ITEMS", 24915)), int(os.environ.get("EVAL_PROBE_F", 176)), 20
g = torch.Generator(device="cpu").manual_seed(5)
emb = torch.randn((n_u + n_i, F), generator=g).to(dev)
rng = np.random.default_rng(6)
train = {u: np.unique(rng.integers(0, n_i, 8)) for u in range(n_u)}
test = {u: np.unique(rng.integers(0, n_i, 3)) for u in range(n_u)}
plan = metrics.EvalPlan(train, test, np.arange(n_u, n_u + n_i), dev)
def run():
    return ops.eval_recall_ndcg(emb, plan.user_ids, plan.item_ids, plan.train_ptr, plan.train_items, plan.test_ptr, plan.test_items, K)
for probe in ["-"]:
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        with ops.KernelTimer() as kt:
            run()
        torch.cuda.synchronize()
        ts.append(kt.summary()["eval_recall_ndcg"][0][1])
    print("probe %s: eval_recall_ndcg %.2f ms (min of 5; %.1f TFLOP/s)" % (probe, min(ts), 2.0 * n_u * n_i * F / min(ts) / 1e9))
