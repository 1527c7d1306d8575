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

`load_aligned(path)` reads an .npz with per-split arrays (see its docstring); `train.DeviceDataset.from_arrays` puts the
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


def load_aligned(path, max_len, acoustic_cols=slice(1, 35)):
    """Read an .npz of word-aligned segments: for split in train/valid/test the object arrays `<split>_text`,
    `<split>_acoustic`, `<split>_visual` (one [len_i, d] array per segment) and `<split>_label` [N]."""
    z = np.load(path, allow_pickle=True)
    splits = {}
    for name in ("train", "valid", "test"):
        splits[name] = dict(text=list(z[name + "_text"]), acoustic=list(z[name + "_acoustic"]),
                            visual=list(z[name + "_visual"]), label=z[name + "_label"])
    return build_splits(splits, max_len, acoustic_cols)
