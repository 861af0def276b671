"""Frozen full-size counts (tests/golden/counts.json): the oracle re-derives config C2
(60k-point nuScenes-shape clouds, the cheapest of the three) and must reproduce generator SHA, counts
and index-table checksums.  The GPU twin (tests/test_gpu_counts.py) checks C2, C3 and C5."""
import json
import os

from tests.golden import make_counts


def load_frozen():
    with open(os.path.join(os.path.dirname(os.path.abspath(make_counts.__file__)), "counts.json")) as f:
        return json.load(f)


def test_table_checksum_is_order_sensitive():
    import numpy as np
    a = np.array([[3, -1, 5], [7, 0, -1]], np.int32)
    b = a.copy()
    b[0, 0], b[1, 0] = a[1, 0], a[0, 0]
    assert make_counts.table_checksum(a) != make_counts.table_checksum(b)
    # exact value by hand: sum((v+2) * (pos+1))
    assert make_counts.table_checksum(a) == sum((int(v) + 2) * (i + 1) for i, v in enumerate(a.ravel()))


def test_oracle_reproduces_frozen_counts_c2():
    frozen = load_frozen()
    assert set(frozen) == {"c2", "c3", "c5"}
    got = make_counts.derive_oracle("c2")
    assert got == frozen["c2"]


def test_frozen_counts_are_consistent():
    """Structural facts the frozen numbers must satisfy whatever produced them."""
    for name, rec in load_frozen().items():
        lv = rec["levels"]
        assert lv["subm1"]["rows"] == sum(s["n_voxels"] for s in rec["samples"])
        for key in ("subm1", "subm2", "subm3", "subm4"):
            cnt = lv[key]["pair_cnt"]
            assert cnt == cnt[::-1] and cnt[13] == lv[key]["rows"] and sum(cnt) == lv[key]["pairs"]
        for conv, nxt in (("spconv2", "subm2"), ("spconv3", "subm3"), ("spconv4", "subm4")):
            assert lv[conv]["rows_out"] == lv[nxt]["rows"] and lv[conv]["shape_out"] == lv[nxt]["shape"]
            assert lv[conv]["pairs"] >= lv[conv]["rows_in"]          # every input reaches >= 1 output
        assert lv["spconv_down2"]["pairs"] <= 2 * lv["spconv_down2"]["rows_in"]   # k (3,1,1), s (2,1,1): <= ceil(3/2) outputs per input
    c3 = load_frozen()["c3"]
    assert [s["n_voxels"] for s in c3["samples"]] == [150000, 150000]  # the MAX_NUMBER_OF_VOXELS cap bites at full size
