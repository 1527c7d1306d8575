"""Train-mode dropout of the latent kernels (reference mfm_model.py:599-617, 644-647, 657: nn.Dropout between fc1
and fc2 of the four z->f MLPs and of the classifier).  The GPU draws its masks from a counter-based generator,
not torch's Philox stream, so dropout cannot be compared sample by sample with a CPU run; instead the masks the
kernel actually used are read back from the plan's latent record and

  * every mask value is 0 or 1/(1-p), the kept fraction is 1-p within 5 sigma, per site, on BOTH latent kernel
    families (row-per-workgroup at B <= 256, staged-LDS above);
  * the stored post-dropout activation equals relu(fc1(input)) * mask recomputed on the host;
  * the masks are INJECTED into the CPU oracle (its nn.Dropout modules replaced by fixed multipliers): forward
    losses and all 78 gradients must then match at 1e-4 -- which pins that the backward applies the same mask
    and scale as the forward;
  * masks change from call to call and with the seed (per-rank seeds of the data-parallel step)."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import configs, synth
from tests.cases import grad_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
SITES = {"zl_to_fl": ("zl_to_fl_dropout", "l"), "za_to_fa": ("za_to_fa_dropout", "a"),
         "zv_to_fv": ("zv_to_fv_dropout", "v"), "zy_to_fy": ("zy_to_fy_dropout", "y"),
         "fy_to_y": ("fy_to_y_dropout", None)}
P = dict(zl_to_fl_dropout=0.2, za_to_fa_dropout=0.5, zv_to_fv_dropout=0.7, zy_to_fy_dropout=0.3, fy_to_y_dropout=0.4)


class _FixedMask(torch.nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask


@pytest.mark.parametrize("B,path", [(200, "row"), (512, "staged"), (64, "forced-staged")])
def test_dropout_masks_statistics_and_gradients(B, path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    if path == "forced-staged":
        monkeypatch.setenv("MFM_LATENT_PATH", "staged")
    else:
        monkeypatch.delenv("MFM_LATENT_PATH", raising=False)
    T = 6
    cfgs = configs.canonical_configs(dropout=True, **P)
    cfg = cfgs[0]
    e = engine.MFMEngine(cfgs)
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=5)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    out = e.forward(x, y, train=True, want_xhat=False)
    rec, grd, lay = e.latent_record(T, B)
    assert lay["row_path"] == (path == "row")
    rec = rec.cpu().numpy().copy()
    masks = {}
    for site, (key, seg) in SITES.items():
        p = cfg[key]
        n = lay["width"][site]
        mk = rec[:, lay["mask"][site]: lay["mask"][site] + n]
        keep = 1.0 / (1.0 - p)
        assert np.all((mk == 0.0) | (np.abs(mk - keep) < 1e-6)), site          # 0 or 1/(1-p), nothing else
        frac = float((mk != 0.0).mean())
        sigma = np.sqrt(p * (1 - p) / mk.size)
        assert abs(frac - (1 - p)) < 5 * sigma + 1e-9, (site, frac, 1 - p)
        # rows draw different masks (the generator is keyed by row and column, not by column alone)
        assert not np.all(mk == mk[:1])
        # the stored activation is relu(fc1(input)) * mask
        if seg is None:
            inp = rec[:, lay["f"]["y"]: lay["f"]["y"] + lay["width"]["zy_to_fy"]]
            wt, bs = w["fy_to_y_fc1.weight"], w["fy_to_y_fc1.bias"]
        else:
            inp = rec[:, lay["mu"][seg]: lay["mu"][seg] + lay["z_n"][seg]]
            wt, bs = w[site + "_fc1.weight"], w[site + "_fc1.bias"]
        want = np.maximum(inp.astype(np.float64) @ wt.T.astype(np.float64) + bs, 0.0) * mk
        got = rec[:, lay["act"][site]: lay["act"][site] + n]
        assert np.max(np.abs(got - want)) < 1e-5 * max(1.0, np.abs(want).max()), site
        masks[site] = mk.copy()
    # ---- the same masks in the oracle: losses and every gradient must agree
    e.backward(x, y, stage=0)
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    for site, (key, _) in SITES.items():
        setattr(m, key, _FixedMask(torch.from_numpy(masks[site])))
    torch.set_num_threads(4)
    terms = O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfg)
    terms["loss"].backward()
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen", "reg", "loss"):
        ref = float(terms[k].detach())
        assert abs(ld[k] - ref) <= TOL * max(abs(ref), 1e-3), (k, ld[k], ref)
    gv = e.grad_views()
    for n, p in m.named_parameters():
        assert grad_err(gv[n].cpu().numpy(), p.grad.numpy()) < TOL, n
    # dropped units carry exactly no gradient into their pre-activation
    g = grd.cpu().numpy()
    for site in SITES:
        seg = g[:, lay["act"][site]: lay["act"][site] + lay["width"][site]]
        assert np.all(seg[masks[site] == 0.0] == 0.0), site
        assert np.any(seg != 0.0), site
    # ---- fresh masks on the next call; eval mode applies none
    e.forward(x, y, train=True, want_xhat=False)
    rec2 = e.latent_record(T, B)[0].cpu().numpy()
    for site in SITES:
        a = rec2[:, lay["mask"][site]: lay["mask"][site] + lay["width"][site]]
        assert not np.array_equal(a, masks[site]), site
    e.forward(x, y, train=False, want_xhat=False)
    rec3 = e.latent_record(T, B)[0].cpu().numpy()
    for site in SITES:
        assert np.all(rec3[:, lay["mask"][site]: lay["mask"][site] + lay["width"][site]] == 1.0), site


def test_dropout_streams_differ_between_data_parallel_ranks():
    """DataParallelStep gives every rank its own seed: two engines that differ only in the
    rank must draw different masks for the same local row numbers."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine, train
    cfgs = configs.canonical_configs(dropout=True, **P)
    B, T = 32, 4
    xn, yn = synth.make_batch(cfgs[0]["input_dims"], B, T, seed=5)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    got = []
    for rank in (0, 1):
        e = engine.MFMEngine(cfgs)
        e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
        train.DataParallelStep(e, world=2, allreduce=lambda t: None, rank=rank)
        assert e.reg_scale == 2.0
        e.forward(x, y, train=True, want_xhat=False)
        rec, _, lay = e.latent_record(T, B)
        o = lay["mask"]["zv_to_fv"]
        got.append(rec[:, o:o + lay["width"]["zv_to_fv"]].cpu().numpy().copy())
    assert not np.array_equal(got[0], got[1])
