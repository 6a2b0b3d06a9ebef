"""On-disk formats and collaborative-KG construction (SURVEY.md 8f "next" #4).

Reads the files the reference's ``DataLoader`` reads (``dataset.py:14-33``, format described in
``datasets/README.md``) and builds the same triplet arrays and graphs (``dataset.py:57-120``):

* ``uv_train.pd`` / ``uv_val.pd`` / ``uv_test.pd`` - tab separated, header ``u\\tv`` (an optional
  third column ``r`` holds an interaction type), int32 ids starting at 0;
* ``kg_item.pd`` - tab separated, header ``h\\tr\\tt``, item ids shared with the interaction
  files, attribute entities numbered after the items.

Node ids are laid out ``<users | items | attribute entities>`` (items and entities are shifted by
``n_users``, ``dataset.py:57-73``); the CKG triplets are the item-KG triplets followed by the
user->item pairs (relation ``R_kg``) and their reverses (relation ``R_kg + n_uv_rel``)
(``dataset.py:76-89``); a graph has one edge per triplet row, ``src = t``, ``dst = h``,
``edata['type'] = r`` (``dataset.py:112-120``).  Pure host code (numpy); the graphs it returns
are this package's DGLGraph, whose kernels run on the GPU.
"""
import os

import numpy as np
import torch

from .graph import DGLGraph

UV_FILES = ("uv_train.pd", "uv_val.pd", "uv_test.pd")
KG_FILE = "kg_item.pd"


def read_table(path, columns=None):
    """A tab-separated int32 table with a header line -> (column names, (rows, cols) int32)."""
    with open(path) as f:
        names = f.readline().rstrip("\n").split("\t")
        body = f.read()
    if body.strip():
        data = np.loadtxt(body.splitlines(), dtype=np.int32, delimiter="\t", ndmin=2)
    else:
        data = np.zeros((0, len(names)), np.int32)
    if columns is not None and names[:len(columns)] != list(columns):
        raise ValueError("%s: expected header starting with %s, found %s" % (path, list(columns), names))
    return names, data


def write_table(path, names, data):
    np.savetxt(path, np.asarray(data, dtype=np.int64), fmt="%d", delimiter="\t", header="\t".join(names),
               comments="")


def _uv_triplets(uv, offset_rel, n_uv_rel, symmetric=True):
    """dataset.py:165-185: [u, R, v] rows, then (symmetric) the reversed [v, R + n_uv_rel, u] rows."""
    rel = np.zeros(len(uv), np.int32) if uv.shape[1] == 2 else uv[:, 2]
    fwd = np.stack([uv[:, 0], rel + offset_rel, uv[:, 1]], 1).astype(np.int32)
    if not symmetric:
        return fwd
    rev = np.stack([uv[:, 1], rel + offset_rel + n_uv_rel, uv[:, 0]], 1).astype(np.int32)
    return np.vstack([fwd, rev])


class CKGDataset:
    """The arrays the reference's DataLoader derives from a data directory (dataset.py:20-110)."""

    def __init__(self, data_dir, symmetric=True, add_uv2kg=True):
        cols, train = read_table(os.path.join(data_dir, UV_FILES[0]), ("u", "v"))
        _, valid = read_table(os.path.join(data_dir, UV_FILES[1]), ("u", "v"))
        _, test = read_table(os.path.join(data_dir, UV_FILES[2]), ("u", "v"))
        # the reference sorts every split by user (a stable sort would be order preserving inside a
        # user; pandas' default sort_values is quicksort, so only the user order is specified)
        train = train[np.argsort(train[:, 0], kind="stable")]
        valid = valid[np.argsort(valid[:, 0], kind="stable")]
        test = test[np.argsort(test[:, 0], kind="stable")]
        self.n_users = len(np.unique(train[:, 0]))
        self.n_items = len(np.unique(train[:, 1]))
        self.n_train, self.n_valid, self.n_test = len(train), len(valid), len(test)
        n_uv_rel = 1 if train.shape[1] == 2 else len(np.unique(train[:, 2]))
        _, kg = read_table(os.path.join(data_dir, KG_FILE), ("h", "r", "t"))
        # <user> | <item> <attribute entity>: shift items / entities behind the users
        off = self.n_users
        for uv in (train, valid, test):
            uv[:, 1] += off
        kg = kg.copy()
        kg[:, 0] += off
        kg[:, 2] += off
        self.item_id_range = np.arange(off, off + self.n_items)
        n_kg_rel = len(np.unique(kg[:, 1]))
        train_uv = _uv_triplets(train, n_kg_rel, n_uv_rel, symmetric)
        train_valid_uv = _uv_triplets(np.vstack([train, valid]), n_kg_rel, n_uv_rel, symmetric)
        if add_uv2kg:
            self.train_KG_triplet = np.vstack([kg, train_uv]).astype(np.int32)
            self.test_KG_triplet = np.vstack([kg, train_valid_uv]).astype(np.int32)
        else:
            self.train_KG_triplet = self.test_KG_triplet = kg.astype(np.int32)
        self.n_KG_relation = len(np.unique(self.train_KG_triplet[:, 1]))
        self.n_KG_entity = len(np.unique(np.concatenate([self.train_KG_triplet[:, 0], self.train_KG_triplet[:, 2]])))
        self.train_pairs, self.valid_pairs, self.test_pairs = train, valid, test

    def _graph(self, triplets, device=None):
        g = DGLGraph()
        g.add_nodes(self.n_KG_entity)
        g.add_edges(triplets[:, 2], triplets[:, 0])
        g.readonly()
        ids = torch.arange(self.n_KG_entity, dtype=torch.long)
        et = torch.as_tensor(triplets[:, 1].astype(np.int64))
        g.ndata["id"] = ids if device is None else ids.to(device)
        g.edata["type"] = et if device is None else et.to(device)
        return g

    def train_graph(self, device=None):
        """dataset.py:112-120 ``train_g``."""
        return self._graph(self.train_KG_triplet, device)

    def test_graph(self, device=None):
        """dataset.py:122-130 ``test_g`` (train + validation interactions)."""
        return self._graph(self.test_KG_triplet, device)


def save_ckg_files(data_dir, n_users, uv_train, uv_val, uv_test, kg):
    """Write a data directory in the reference's format from raw (un-shifted) id arrays."""
    os.makedirs(data_dir, exist_ok=True)
    for name, arr in zip(UV_FILES, (uv_train, uv_val, uv_test)):
        write_table(os.path.join(data_dir, name), ["u", "v"] + (["r"] if np.asarray(arr).shape[1] == 3 else []), arr)
    write_table(os.path.join(data_dir, KG_FILE), ["h", "r", "t"], kg)


# ---------------------------------------------------------------------------------------------
# Raw KGAT release files -> the reference's data directory (datasets/process_kgat_data.py)
def read_kgat_interactions(path):
    """``train.txt`` / ``test.txt`` of the KGAT release: one line per user, ``user item item ...``
    (process_kgat_data.py:139-149 ``read2u_v_dict``: lines with no item are skipped, repeated
    items of a user collapse).  Returns (P, 2) int64 (user, item) pairs, users in file order and
    each user's items ascending (the reference's order inside a user is that of a Python set -
    unspecified)."""
    pairs = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if len(tok) > 1:
                u = int(tok[0])
                pairs.extend((u, v) for v in sorted({int(t) for t in tok[1:]}))
    return np.asarray(pairs, dtype=np.int64).reshape(-1, 2)


def read_kgat_kg(path):
    """``kg_final.txt``: ``h r t`` per line; duplicate rows dropped, first occurrence kept
    (process_kgat_data.py:213-243 ``read_kg2pd`` without remapping: entity and relation ids must
    already be 0..max, as in the released files)."""
    kg = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if kg.size == 0:
        return kg.reshape(0, 3)
    _, first = np.unique(kg, axis=0, return_index=True)
    kg = kg[np.sort(first)]
    ents, rels = np.unique(kg[:, [0, 2]]), np.unique(kg[:, 1])
    if ents.max() + 1 != ents.size or rels.max() + 1 != rels.size:
        raise ValueError("%s: entity / relation ids are not consecutive from 0" % path)
    return kg


def split_validation(pairs, val_ratio=0.1, seed=0):
    """process_kgat_data.py:179-211 ``split_val(mode="seen")``: a random ``val_ratio`` of the training
    pairs becomes the validation split, then one pair of every user / item that lost all its
    training pairs moves back.  Seeded numpy permutation (the reference uses the global numpy
    state, so its exact split is not reproducible either).  Returns (train, valid)."""
    pairs = np.asarray(pairs)
    rng = np.random.default_rng(seed)
    idx = rng.permutation(len(pairs))
    n_val = int(len(pairs) * val_ratio)
    is_val = np.zeros(len(pairs), bool)
    is_val[idx[:n_val]] = True
    for col in (0, 1):
        seen = np.unique(pairs[~is_val, col])
        lost = np.setdiff1d(np.unique(pairs[:, col]), seen)
        if len(lost):
            cand = np.nonzero(is_val & np.isin(pairs[:, col], lost))[0]
            _, first = np.unique(pairs[cand, col], return_index=True)  # first validation pair of each lost id
            is_val[cand[first]] = False
    return pairs[~is_val], pairs[is_val]


def convert_kgat_release(raw_dir, out_dir, val_ratio=0.1, seed=0):
    """Raw KGAT directory (``train.txt``, ``test.txt``, ``kg_final.txt``) -> the four tables the
    reference's DataLoader (and CKGDataset above) read.  Users / items are re-numbered in sorted
    order over train + test as process_kgat_data.py:151-177 does; item ids of the released files
    already coincide with their KG entity ids."""
    train_all = read_kgat_interactions(os.path.join(raw_dir, "train.txt"))
    test = read_kgat_interactions(os.path.join(raw_dir, "test.txt"))
    kg = read_kgat_kg(os.path.join(raw_dir, "kg_final.txt"))
    both = np.vstack([train_all, test])
    users, items = np.unique(both[:, 0]), np.unique(both[:, 1])
    if items.max() + 1 != items.size:
        raise ValueError("item ids are not consecutive from 0: the KG entity numbering would not line up")

    def remap(p):
        return np.stack([np.searchsorted(users, p[:, 0]), np.searchsorted(items, p[:, 1])], 1)
    train, valid = split_validation(remap(train_all), val_ratio, seed)
    save_ckg_files(out_dir, len(users), train, valid, remap(test), kg)
    return CKGDataset(out_dir)
