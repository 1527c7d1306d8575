"""Deterministic synthetic inputs and weights (numpy only, no torch RNG).

There is no network and the reference's CMU-MOSI pickles are private
(reference data_loader.py:10, mfm_mosi.py:61), so every test, fixture and
benchmark runs on data of the reference's *shape*:

* `make_batch` -- word-aligned feature sequences, time-major `[T, B, D]` as the
  reference feeds its model after `swapaxes(0,1)` (mfm_mosi.py:391); language
  features ~N(0,0.4^2) (GloVe-like), acoustic ~N(0,1) (COVAREP is not normalised,
  mfm_mosi.py:96-103), visual ~U(-1,1) (FACET divided by its max-abs,
  mfm_mosi.py:94-102); a random number of *leading* timesteps is zero, as the
  reference front-pads short utterances (data_loader.py:139-143).
* `make_weights` -- one RandomState per tensor so the reference model, the oracle
  and the HIP path can be loaded with bit-identical parameters without relying on
  torch's initialisation streams.  Bounds follow torch's defaults (U(+-1/sqrt(h))
  for LSTMCell, U(+-1/sqrt(fan_in)) for Linear).
"""
import numpy as np


def make_batch(input_dims, B, T, seed=7, output_dim=1, classes=0, max_pad=10):
    d_l, d_a, d_v = input_dims
    rs = np.random.RandomState(seed)
    x_l = rs.normal(0.0, 0.4, size=(T, B, d_l))
    x_a = rs.normal(0.0, 1.0, size=(T, B, d_a))
    x_v = rs.uniform(-1.0, 1.0, size=(T, B, d_v))
    x = np.concatenate([x_l, x_a, x_v], axis=2).astype(np.float32)
    pad = rs.randint(0, min(max_pad, max(T - 1, 0)) + 1, size=B)
    for b in range(B):
        x[:pad[b], b, :] = 0.0
    if classes:
        y = rs.randint(0, classes, size=B).astype(np.int64)
    elif output_dim == 1:
        y = rs.uniform(-3.0, 3.0, size=B).astype(np.float32)
    else:
        y = rs.uniform(-3.0, 3.0, size=(B, output_dim)).astype(np.float32)
    return x, y


def make_dataset(input_dims, N, T, seed=11, output_dim=1, classes=0):
    """`(X[T,N,D], y[N])` -- N samples laid out time-major like a swapped
    reference split; batches are contiguous column slices (mfm_mosi.py:425-429)."""
    return make_batch(input_dims, N, T, seed=seed, output_dim=output_dim, classes=classes)


def make_weights(shapes, seed=1234):
    """`shapes`: ordered mapping name -> shape (a state_dict's keys/shapes).
    Returns an ordered dict name -> float32 ndarray."""
    out = {}
    for i, (name, shape) in enumerate(shapes.items()):
        shape = tuple(int(s) for s in shape)
        rs = np.random.RandomState(seed + i)
        if ".lstm" in name or name.startswith("lstm_"):
            # weight_ih [4h,d], weight_hh [4h,h], bias_* [4h] -> bound 1/sqrt(h)
            h = shape[0] // 4
            k = 1.0 / np.sqrt(h)
        elif name.endswith(".weight"):
            k = 1.0 / np.sqrt(shape[1])
        else:
            # Linear bias: bound 1/sqrt(fan_in); fan_in is not visible from the
            # bias shape, so use the sibling weight if present.
            w = name[:-len("bias")] + "weight"
            fan_in = shapes[w][1] if w in shapes else shape[0]
            k = 1.0 / np.sqrt(fan_in)
        out[name] = rs.uniform(-k, k, size=shape).astype(np.float32)
    return out
