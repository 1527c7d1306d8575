"""factorized_amd/data.py against a literal restatement of the reference's rules (data_loader.py:131-160,
mfm_mosi.py:76-125, 391-393) on ragged synthetic segments.  CPU only."""
import os
import tempfile

import numpy as np

from factorized_amd import data as D


def _segments(rs, n, dims, max_words=14):
    out = {"text": [], "acoustic": [], "visual": []}
    for _ in range(n):
        ln = int(rs.randint(1, max_words + 1))
        out["text"].append(rs.normal(size=(ln, dims[0])).astype(np.float32))
        a = rs.normal(size=(ln, dims[1])).astype(np.float32)
        a[rs.rand(ln, dims[1]) < 0.05] = np.nan
        a[rs.rand(ln, dims[1]) < 0.02] = -np.inf
        out["acoustic"].append(a)
        out["visual"].append((rs.normal(size=(ln, dims[2])) * rs.uniform(0.1, 30, size=dims[2])).astype(np.float32))
    out["label"] = rs.uniform(-3, 3, size=n).astype(np.float32)
    return out


def _reference_rules(sp, max_len):
    """get_data's loops, written like the reference writes them"""
    res = {"text": [], "acoustic": [], "visual": []}
    for i in range(len(sp["text"])):
        for key in res:
            fts = sp[key][i]
            rows = []
            if max_len >= len(fts):
                for _ in range(max_len - len(fts)):
                    rows.append(np.zeros(fts.shape[1]))
                for w in fts:
                    rows.append(w)
            else:
                for w in fts[len(fts) - max_len:]:
                    rows.append(w)
            res[key].append(rows)
    return {k: np.array(v) for k, v in res.items()}


def test_adapter_matches_reference_rules_and_feeds_the_dataset():
    rs = np.random.RandomState(0)
    dims, max_len = (6, 40, 5), 9
    splits = {"train": _segments(rs, 23, dims), "valid": _segments(rs, 7, dims), "test": _segments(rs, 11, dims)}
    splits["train"]["visual"] = [np.concatenate([v[:, :4], np.zeros((v.shape[0], 1), np.float32)], 1) for v in splits["train"]["visual"]]
    out = D.build_splits(splits, max_len)
    ref = {}
    for name, sp in splits.items():
        sp2 = dict(sp)
        sp2["acoustic"] = []
        for a in sp["acoustic"]:
            a = a.copy(); a[np.isnan(a)] = 0; a[np.isneginf(a)] = 0
            sp2["acoustic"].append(a)
        ref[name] = _reference_rules(sp2, max_len)
    fmax = np.max(np.max(np.abs(ref["train"]["visual"]), axis=0), axis=0)
    fmax[fmax == 0] = 1
    for name in splits:
        X, y, ln = out[name]
        want = np.concatenate((ref[name]["text"], ref[name]["acoustic"][:, :, 1:35], ref[name]["visual"] / fmax), axis=2).swapaxes(0, 1)
        assert X.shape == (max_len, len(splits[name]["text"]), 6 + 34 + 5) and X.dtype == np.float32 and X.flags["C_CONTIGUOUS"]
        assert np.allclose(X, want, rtol=1e-6, atol=1e-6)
        assert np.all(np.isfinite(X))
        assert ln.tolist() == [len(s) for s in splits[name]["text"]]
        assert np.array_equal(y, splits[name]["label"])
    # the all-zero visual feature of the train split keeps scale 1 (mfm_mosi.py:96)
    assert D.visual_scale(ref["train"]["visual"].astype(np.float32))[4] == 1.0
    # npz round trip
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "aligned.npz")
        D.save_aligned(path, splits)
        assert all(v.dtype != object for v in np.load(path, allow_pickle=False).values())
        again = D.load_aligned(path, max_len)
        assert np.array_equal(again["test"][0], out["test"][0])
