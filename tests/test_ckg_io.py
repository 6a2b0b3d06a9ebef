"""CPU: on-disk formats + CKG construction (SURVEY 8f "next" #4) against golden vectors produced
by the reference's own dataset.DataLoader (tests/golden/make_golden_dataset.py)."""
import os

import numpy as np

from conftest import GOLDEN_DIR

from dgl_kgat_amd import ckg_io


def _golden():
    z = np.load(os.path.join(GOLDEN_DIR, "toy_dataset.npz"))
    return {k: z[k] for k in z.files}


def _rows_sorted(a):
    return a[np.lexsort(a.T[::-1])]


def test_loader_matches_reference_dataloader(tmp_path):
    g = _golden()
    d = str(tmp_path / "data")
    ckg_io.save_ckg_files(d, int(g["n_users"]), g["uv_train"], g["uv_val"], g["uv_test"], g["kg"])
    ds = ckg_io.CKGDataset(d)
    assert (ds.n_users, ds.n_items, ds.n_KG_relation, ds.n_KG_entity) == (
        int(g["n_users"]), int(g["n_items"]), int(g["n_KG_relation"]), int(g["n_KG_entity"]))
    assert np.array_equal(ds.item_id_range, g["item_id_range"])
    n_kg = len(g["kg"])
    for mine, ref in ((ds.train_KG_triplet, g["train_KG_triplet"]), (ds.test_KG_triplet, g["test_KG_triplet"])):
        assert mine.dtype == np.int32 and mine.shape == ref.shape
        # the item-KG block is in file order; the interaction blocks are sorted by user (the order
        # inside one user is not specified by the reference: pandas' unstable default sort)
        assert np.array_equal(mine[:n_kg], ref[:n_kg])
        half = (len(ref) - n_kg) // 2
        for lo, hi, key in ((n_kg, n_kg + half, 0), (n_kg + half, len(ref), 2)):
            assert np.array_equal(mine[lo:hi, key], ref[lo:hi, key])  # same user order
            assert np.array_equal(_rows_sorted(mine[lo:hi]), _rows_sorted(ref[lo:hi]))
        # the reversed block mirrors the forward block row by row
        assert np.array_equal(mine[n_kg:n_kg + half][:, [2, 0]], mine[n_kg + half:][:, [0, 2]])
    graph = ds.train_graph()
    src, dst = graph.edges()
    assert graph.number_of_nodes() == int(g["n_KG_entity"]) and graph.number_of_edges() == len(g["train_g_src"])
    assert np.array_equal(np.sort(src.numpy() * 10_000 + dst.numpy()),
                          np.sort(g["train_g_src"].astype(np.int64) * 10_000 + g["train_g_dst"]))
    assert np.array_equal(src.numpy(), ds.train_KG_triplet[:, 2]) and np.array_equal(dst.numpy(), ds.train_KG_triplet[:, 0])
    assert graph.edata["type"].dtype.is_floating_point is False and graph.ndata["id"].tolist() == list(range(ds.n_KG_entity))
    assert np.array_equal(np.bincount(graph.edata["type"].numpy()), np.bincount(g["train_g_type"]))


def test_tables_round_trip(tmp_path):
    p = str(tmp_path / "t.pd")
    a = np.array([[3, 1, 7], [0, 0, 2]], np.int32)
    ckg_io.write_table(p, ["h", "r", "t"], a)
    assert open(p).readline() == "h\tr\tt\n"
    names, b = ckg_io.read_table(p, ("h", "r", "t"))
    assert names == ["h", "r", "t"] and np.array_equal(a, b)
    ckg_io.write_table(p, ["u", "v"], np.zeros((0, 2), np.int32))
    assert ckg_io.read_table(p)[1].shape == (0, 2)


def test_ranking_metrics_match_reference():
    """recall@K / ndcg@K (metric.py:36-68) against the value the reference's own code returns."""
    import torch
    from dgl_kgat_amd import metrics
    g = _golden()
    as_dict = lambda users, items: {int(u): np.array([int(x) for x in str(s).split(";")]) for u, s in zip(users, items)}  # noqa: E731
    train = as_dict(g["train_users"], g["train_user_items"])
    test = as_dict(g["test_users"], g["test_user_items"])
    # (the matmul + stable-sort form runs on CPU tensors; the HIP kernel behind calc_recall_ndcg is checked against
    # the same value in tests/test_gpu_eval.py)
    rec, ndcg = metrics.calc_recall_ndcg_sorted(torch.as_tensor(g["metric_embedding"]), train, test, g["item_id_range"],
                                                K=5, batch_users=4)
    assert abs(rec - g["metric_recall_ndcg_at5"][0]) < 1e-12 and abs(ndcg - g["metric_recall_ndcg_at5"][1]) < 1e-12
    # the per-user restatement the GPU test checks larger cases against: pinned to the same value
    from oracle import kgat_oracle as orc
    rec2, ndcg2 = orc.recall_ndcg_per_user(g["metric_embedding"], train, test, g["item_id_range"], 5)
    assert abs(rec2 - g["metric_recall_ndcg_at5"][0]) < 1e-12 and abs(ndcg2 - g["metric_recall_ndcg_at5"][1]) < 1e-12


def test_raw_kgat_release_files(tmp_path):
    """train.txt / test.txt / kg_final.txt of the KGAT release (process_kgat_data.py): parsing
    rules, the "seen" validation split invariants, and the round trip into a CKGDataset."""
    from dgl_kgat_amd import ckg_io
    raw = tmp_path / "raw"
    raw.mkdir()
    rng = np.random.default_rng(3)
    n_users, n_items, n_ent, n_rel = 40, 30, 55, 4
    lines, pairs = [], set()
    for u in range(n_users):
        its = rng.choice(n_items, rng.integers(1, 9), replace=False).tolist()
        lines.append(" ".join(map(str, [u] + its + its[:1])))       # one repeated item per user
        pairs |= {(u, v) for v in its}
    lines.insert(5, "999")                                           # a user without items: skipped
    covered = {v for _, v in pairs}
    missing = sorted(set(range(n_items)) - covered)
    (raw / "train.txt").write_text("\n".join(lines) + "\n")
    (raw / "test.txt").write_text("\n".join("%d %s" % (u, " ".join(map(str, missing + [u % n_items])))
                                            for u in range(0, n_users, 3)) + "\n")
    kg = np.stack([rng.integers(0, n_ent, 300), rng.integers(0, n_rel, 300), rng.integers(0, n_ent, 300)], 1)
    kg[:n_ent, 0] = np.arange(n_ent)                                 # every entity / relation id appears
    kg[:n_rel, 1] = np.arange(n_rel)
    kg = np.vstack([kg, kg[:20]])                                    # duplicate rows
    np.savetxt(raw / "kg_final.txt", kg, fmt="%d")
    got = ckg_io.read_kgat_interactions(str(raw / "train.txt"))
    assert {tuple(p) for p in got.tolist()} == pairs and len(got) == len(pairs)
    kg_read = ckg_io.read_kgat_kg(str(raw / "kg_final.txt"))
    assert len(kg_read) == len({tuple(r) for r in kg.tolist()}) and np.array_equal(kg_read[:5], kg[:5])
    train, valid = ckg_io.split_validation(got, 0.3, seed=1)
    assert len(train) + len(valid) == len(got) and len(valid) <= int(0.3 * len(got))
    assert set(np.unique(train[:, 0])) == set(np.unique(got[:, 0]))  # no user / item lost from training
    assert set(np.unique(train[:, 1])) == set(np.unique(got[:, 1]))
    t2, v2 = ckg_io.split_validation(got, 0.3, seed=1)
    assert np.array_equal(t2, train) and np.array_equal(v2, valid)
    ds = ckg_io.convert_kgat_release(str(raw), str(tmp_path / "out"), val_ratio=0.2, seed=2)
    assert ds.n_users == n_users and ds.n_items == n_items
    assert ds.n_train + ds.n_valid == len(pairs)
    assert ds.train_KG_triplet.shape[1] == 3 and ds.n_KG_relation == n_rel + 2
    g = ds.train_graph()
    assert g.number_of_edges() == len(kg_read) + 2 * ds.n_train
