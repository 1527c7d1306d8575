"""Data adapter: word-aligned multimodal segments -> the `[T, N, D]` tensors the MFM step consumes.

The reference builds its inputs in `data_loader.py` (private CMU-MOSI file tree: FACET CSVs, COVAREP .mat files,
word-aligned transcripts) and in the preamble of `mfm_mosi.py:55-125`.  The raw-file readers are tied to that tree and
are not reproduced; what IS restated here is everything between "per-word feature vectors of every segment" (what a
CMU-MultimodalSDK word-level alignment exports) and the model input, with the reference's exact rules:

  * a segment longer than `max_len` keeps its LAST `max_len` words; a shorter one is FRONT-padded with zeros
    (data_loader.py:131-160);
  * acoustic features: NaN and -inf averages become 0 (data_loader.py:98-99); columns 1:35 are used when no feature
    selection mask is given (mfm_mosi.py:76-79);
  * visual features are divided by the TRAIN split's per-feature max-abs, zeros replaced by 1 (mfm_mosi.py:94-102);
    acoustic features are NOT normalised (the lines are commented out in the reference);
  * X = concat(word embedding, acoustic, visual) along the feature axis (mfm_mosi.py:108-125), then
    `swapaxes(0, 1)` to time-major `[T, N, D]` (mfm_mosi.py:391-393).

`save_aligned` / `load_aligned` write / read a pickle-free .npz of such segments (concatenated arrays + offsets); `train.DeviceDataset.from_arrays` puts the
result into HBM in the `[nb, T, B, D]` batch layout of the fused step.
"""
import numpy as np


def clean_acoustic(a):
    """data_loader.py:98-99: NaN / -inf word averages -> 0 (in a copy)."""
    a = np.array(a, dtype=np.float32, copy=True)
    a[np.isnan(a)] = 0.0
    a[np.isneginf(a)] = 0.0
    return a


def pad_segments(segments, max_len):
    """segments: list of [len_i, d] arrays (len_i >= 1) -> ([N, max_len, d] float32, lengths[N]).
    Front-pad with zeros / keep the last `max_len` words (data_loader.py:131-160)."""
    n = len(segments)
    d = int(np.asarray(segments[0]).shape[1]) if n else 0
    out = np.zeros((n, max_len, d), dtype=np.float32)
    lengths = np.zeros(n, dtype=np.int64)
    for i, seg in enumerate(segments):
        seg = np.asarray(seg, dtype=np.float32)
        lengths[i] = seg.shape[0]                      # the reference records the UNtruncated length
        if seg.shape[0] >= max_len:
            out[i] = seg[seg.shape[0] - max_len:]
        else:
            out[i, max_len - seg.shape[0]:] = seg
    return out, lengths


def visual_scale(visual_train):
    """mfm_mosi.py:94-96: per-feature max-abs over the train split, zeros -> 1."""
    m = np.max(np.max(np.abs(visual_train), axis=0), axis=0)
    m = np.array(m, dtype=np.float32, copy=True)
    m[m == 0] = 1.0
    return m


def assemble(text_emb, acoustic, visual, vis_scale, acoustic_cols=slice(1, 35)):
    """[N, T, d_l], [N, T, d_a_raw], [N, T, d_v] -> time-major X [T, N, d_l + d_a + d_v] float32
    (mfm_mosi.py:76-79, 98-125, 391-393).  `acoustic_cols=None` keeps every acoustic column."""
    a = acoustic if acoustic_cols is None else acoustic[:, :, acoustic_cols]
    x = np.concatenate([np.asarray(text_emb, dtype=np.float32), np.asarray(a, dtype=np.float32),
                        np.asarray(visual, dtype=np.float32) / vis_scale], axis=2)
    return np.ascontiguousarray(x.swapaxes(0, 1))


def build_splits(splits, max_len, acoustic_cols=slice(1, 35)):
    """splits: {'train'|'valid'|'test': dict(text=[segments of [len, d_l]], acoustic=[...], visual=[...], label=[N])}
    -> {'train': (X[T,N,D], y[N], lengths[N]), ...} with the train split's visual scale applied to all three."""
    padded = {}
    for name, sp in splits.items():
        t, ln = pad_segments(sp["text"], max_len)
        a, _ = pad_segments([clean_acoustic(s) for s in sp["acoustic"]], max_len)
        v, _ = pad_segments(sp["visual"], max_len)
        padded[name] = (t, a, v, np.asarray(sp["label"], dtype=np.float32), ln)
    scale = visual_scale(padded["train"][2])
    return {name: (assemble(t, a, v, scale, acoustic_cols), y, ln) for name, (t, a, v, y, ln) in padded.items()}


def save_aligned(path, splits):
    """Write word-aligned segments in the pickle-free layout `load_aligned` reads: per split and modality ONE
    concatenated array `<split>_<modality>` [sum(len_i), d] plus `<split>_offsets` [N + 1] (segment i = rows
    offsets[i]:offsets[i+1] of all three modalities -- they are aligned word by word) and `<split>_label` [N]."""
    blob = {}
    for name, sp in splits.items():
        lens = [int(np.asarray(s).shape[0]) for s in sp["text"]]
        blob[name + "_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        for k in ("text", "acoustic", "visual"):
            assert [int(np.asarray(s).shape[0]) for s in sp[k]] == lens, "modalities must be aligned word by word"
            blob["%s_%s" % (name, k)] = np.concatenate([np.asarray(s, dtype=np.float32) for s in sp[k]], axis=0)
        blob[name + "_label"] = np.asarray(sp["label"], dtype=np.float32)
    np.savez(path, **blob)


def load_aligned(path, max_len, acoustic_cols=slice(1, 35)):
    """Read an .npz of word-aligned segments written by `save_aligned` (plain numeric arrays: concatenated segments +
    offsets).  Loaded with allow_pickle=False -- a dataset file cannot execute code."""
    z = np.load(path, allow_pickle=False)
    splits = {}
    for name in ("train", "valid", "test"):
        off = z[name + "_offsets"]
        sp = {"label": z[name + "_label"]}
        for k in ("text", "acoustic", "visual"):
            flat = z["%s_%s" % (name, k)]
            sp[k] = [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        splits[name] = sp
    return build_splits(splits, max_len, acoustic_cols)
