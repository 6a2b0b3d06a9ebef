#!/usr/bin/env python3
"""The 1/2/4/8-GPU curve as a MODEL (points that cannot be measured on a one-GPU box are shipped as a
model, labelled as such): per-rank local step time MEASURED on one GPU with the exchange stubbed out - every rank
of a P-way destination partition in turn, P = 1, 2, 4, 8 - plus the link arithmetic of SURVEY 5 for the per-layer
exchange.  Writes profiles/r06_scaling_model.json; every predicted field is named model_*.

    python scripts/scaling_model.py [--configs 2,3,4] [--scale4 1.0] [--out profiles/r06_scaling_model.json]

Link model (MI355X node: 8 GPUs fully connected, 7 xGMI links per GPU, 153.6 GB/s per link and direction):
  all-reduce (ring, the north star's exchange): every link carries 2 (P-1)/P x S      -> S x 2 (P-1)/P / 153.6 GB/s
  all-gather of the owned row slices (direct):  every rank receives P-1 slices of ~S/P,
                                                one per link, side by side            -> max slice bytes / 153.6 GB/s
  + LAUNCH_US per collective (a fixed cost of starting an RCCL kernel on every rank; not measured here).
S = N x D_out x 4 bytes of a layer's output; the last layer's exchange is part of the step as bench.py runs it (the
readout needs all rows on every rank).  Reference: single device (kgat.py:69-71); the split is BASELINE.json's."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgl_kgat_amd as K  # noqa: E402
from dgl_kgat_amd import ops, partition, synth  # noqa: E402

LINK_GBS = 153.6
LAUNCH_US = 20.0

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="2,3,4")
ap.add_argument("--scale4", type=float, default=1.0, help="scale of configs[4] (1.0 = 10 M nodes / 200 M edges)")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--graphs", type=int, default=1,
                help="1: also measure every rank's local step replayed as HIP graphs (partition.GraphedShardForward)")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_scaling_model.json"))
args = ap.parse_args()
dev = torch.device("cuda:0")
K.enable_lazy_edge_weights()   # as bench.py's headline steps


def workload(cfg):
    if cfg in (2, 3):
        n, trip, R = synth.amazon_book_ckg()
        d = 64 if cfg == 2 else 128
        return "configs[%d]" % cfg, "amazon-book-shaped CKG, d = %d" % d, n, R, d, synth.build_graph(n, trip, dev), len(trip)
    n, e = int(10_000_000 * args.scale4), int(200_000_000 * args.scale4)
    src, dst, et = synth.power_law_coo_device(n, e, 64, dev)
    return "configs[4]", "power-law CKG N=%d E=%d (scale %g), d = 64" % (n, e, args.scale4), n, 64, 64, \
        synth.build_graph_device(n, src, dst, et), e


rows = []
for cfg in [int(c) for c in args.configs.split(",")]:
    key, desc, n, R, d, g, E = workload(cfg)
    torch.manual_seed(1234)
    model = K.KGATPropagation(n, R, d, d, 3, d, dropout=0.0).to(dev)
    widths = [layer.res_fc_2.out_features for layer in model.layers]
    steps = args.steps if E < 20_000_000 else max(args.steps // 4, 3)
    for P in (1, 2, 4, 8):
        local, graphed, att, rows_per_rank, edges_per_rank = [], [], [], [], []
        bounds = None
        for r in range(P):
            if P == 1:
                sg = g
            else:
                sg, keep = partition.shard_graph(g, r, P, bounds=bounds)
                bounds = sg.partition.bounds
                sg.partition.exchange_enabled = False     # the collective is skipped: local work only

            def step():
                with torch.no_grad():
                    sg.edata["w"] = model.compute_attention(sg)
                    return model.gnn(sg)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            local.append((time.perf_counter() - t0) / steps * 1e3)
            with torch.no_grad():                           # the attention refresh alone (no exchange inside it)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    sg.edata["w"] = model.compute_attention(sg)
                torch.cuda.synchronize()
                att.append((time.perf_counter() - t0) / steps * 1e3)
            if args.graphs:
                gs = partition.GraphedShardForward(model, sg)
                for _ in range(3):
                    gs()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    gs()
                torch.cuda.synchronize()
                graphed.append((time.perf_counter() - t0) / steps * 1e3)
                del gs
            rows_per_rank.append(n if P == 1 else int(bounds[r + 1] - bounds[r]))
            edges_per_rank.append(int(sg.number_of_edges()))
            if P > 1:
                del sg
                torch.cuda.empty_cache()
        t_eager = max(local)
        t_local = max(graphed) if graphed else t_eager      # the step ends when the slowest rank has its rows
        S = [n * w * 4 for w in widths]
        frac_max = max(rows_per_rank) / n
        ar = [0.0 if P == 1 else (2 * (P - 1) / P * s / (LINK_GBS * 1e9) * 1e3 + LAUNCH_US * 1e-3) for s in S]
        ag = [0.0 if P == 1 else (frac_max * s / (LINK_GBS * 1e9) * 1e3 + LAUNCH_US * 1e-3) for s in S]
        # exchange overlapped with compute (Partition.propagate_overlapped, K row blocks per layer: block k travels
        # while block k + 1 is computed): a layer then costs max(c, x) + min(c, x) / K instead of c + x, with c the
        # slowest rank's per-layer compute ((local - attention) / layers: an even split, not measured per layer)
        K_CH = 4
        slow = int(np.argmax(graphed if graphed else local))
        c_layer = max(t_local - att[slow], 0.0) / len(widths)

        def overlapped(xs):
            return att[slow] + sum(max(c_layer, x) + min(c_layer, x) / K_CH for x in xs)
        row = {"config": key, "workload": desc, "P": P, "n_nodes": n, "n_edges": E, "layer_output_bytes": S,
               "measured_attention_ms_per_rank": [round(x, 4) for x in att],
               "model_overlap_chunks": K_CH,
               "model_ms_per_step_allreduce_overlapped": round(t_local if P == 1 else overlapped(ar), 4),
               "model_ms_per_step_allgather_overlapped": round(t_local if P == 1 else overlapped(ag), 4),
               "model_exposed_exchange_ms_allgather_overlapped": round(0.0 if P == 1 else overlapped(ag) - t_local, 4),
               "model_edges_per_s_allgather_overlapped": round(len(widths) * E / ((t_local if P == 1 else overlapped(ag)) * 1e-3), 1),
               "measured_local_ms_per_rank_eager_launches": [round(x, 4) for x in local],
               "measured_local_ms_per_rank_hip_graphs": [round(x, 4) for x in graphed],
               "measured_local_ms_slowest_rank": round(t_local, 4),
               "measured_local_ms_slowest_rank_eager_launches": round(t_eager, 4),
               "rows_per_rank": rows_per_rank, "edges_per_rank": edges_per_rank,
               "model_allreduce_ms_per_layer": [round(x, 4) for x in ar],
               "model_allgather_ms_per_layer": [round(x, 4) for x in ag],
               "model_ms_per_step_allreduce": round(t_local + sum(ar), 4),
               "model_ms_per_step_allgather": round(t_local + sum(ag), 4),
               "model_edges_per_s_allreduce": round(len(widths) * E / ((t_local + sum(ar)) * 1e-3), 1),
               "model_edges_per_s_allgather": round(len(widths) * E / ((t_local + sum(ag)) * 1e-3), 1)}
        rows.append(row)
        print("%s P=%d: local (slowest rank) %.3f ms (eager launches %.3f) | model all-reduce +%.3f -> %.3f ms (%.2f G edges/s) | model all-gather "
              "+%.3f -> %.3f ms (%.2f G edges/s), overlapped %.3f ms" % (key, P, t_local, t_eager, sum(ar), t_local + sum(ar),
                                                      row["model_edges_per_s_allreduce"] / 1e9, sum(ag), t_local + sum(ag),
                                                      row["model_edges_per_s_allgather"] / 1e9,
                                                      row["model_ms_per_step_allgather_overlapped"]), flush=True)
    del g, model
    torch.cuda.empty_cache()

base = {r["config"]: r for r in rows if r["P"] == 1}
for r in rows:
    b = base[r["config"]]
    r["model_speedup_vs_P1_allreduce"] = round(b["model_ms_per_step_allreduce"] / r["model_ms_per_step_allreduce"], 3)
    r["model_speedup_vs_P1_allgather"] = round(b["model_ms_per_step_allgather"] / r["model_ms_per_step_allgather"], 3)
out = {"what": "MODEL of the 1/2/4/8-GPU curve: measured_* fields are per-rank local step times measured on ONE MI355X with "
               "the exchange stubbed (each rank of the P-way destination partition in turn); model_* fields add the link "
               "arithmetic of SURVEY 5 and are NOT measurements",
       "link_model": {"xgmi_links_per_gpu": 7, "GBs_per_link_per_direction": LINK_GBS, "launch_us_per_collective": LAUNCH_US,
                      "allreduce": "ring: 2 (P-1)/P x S per link", "allgather": "direct: largest row slice per link",
                      "overlap": "model_ms_per_step_*: none (exchange fully exposed after each layer); model_*_overlapped: "
                                 "Partition.propagate_overlapped with 4 row blocks per layer, max(c, x) + min(c, x) / 4 per layer"},
       "device": torch.cuda.get_device_properties(dev).name, "rows": rows}
with open(args.out, "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote", args.out)
