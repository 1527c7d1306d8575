"""Pins the CPU oracle (oracle/mfm_oracle.py) to the reference's own outputs
(tests/golden/*.npz, produced by tests/golden/make_golden.py from an import of
/root/reference/mfm_model.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import configs, synth
from tests import cases

TOL = 2e-6   # same torch ops on the same host -> essentially bit-equal


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference(name):
    torch.set_num_threads(1)
    cs = cases.load_case(name)
    gold, cfg = cs["gold"], cs["cfg"]
    model = O.build(cs["variant"], cs["cfgs"])
    w = synth.make_weights(O.state_shapes(model), seed=1234)
    O.load_numpy_weights(model, w)
    if cs["variant"] == "mmd":
        g = torch.from_numpy(gold["mmd_gauss"])
        sizes = [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]]
        model.mmd_gauss = list(torch.split(g, sizes, dim=1))
    model.train()
    x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
    opt = torch.optim.Adam(model.parameters())
    trace = []
    for s in range(cs["steps"]):
        opt.zero_grad()
        terms = O.loss_terms(model, x, y, cfg, cs["loss_kind"])
        terms["loss"].backward()
        if s == 0:
            for k in ("disc", "gen", "gen_l", "gen_a", "gen_v", "reg", "loss"):
                assert abs(terms[k].item() - float(gold["fwd_" + k])) <= TOL * max(1.0, abs(float(gold["fwd_" + k]))), k
            dec = terms["decoded"]
            assert cases.rel_err(dec[3].detach().numpy(), gold["y_hat"]) < TOL
            assert cases.rel_err(dec[1].detach().numpy(), gold["x_a_hat"]) < TOL
            assert cases.rel_err(dec[0][0].detach().numpy(), gold["x_l_hat_first"]) < TOL
            assert cases.rel_err(dec[2][-1].detach().numpy(), gold["x_v_hat_last"]) < TOL
            names = [n for n, _ in model.named_parameters()]
            assert names == list(gold["param_names"])
            gs = np.stack([cases.summarize(p.grad.numpy()) if p.grad is not None else np.full(10, np.nan)
                           for p in model.parameters()])
            both_nan = np.isnan(gs) & np.isnan(gold["grad_summary"])
            assert np.allclose(np.where(both_nan, 0, gs), np.where(both_nan, 0, gold["grad_summary"]),
                               rtol=1e-5, atol=1e-6)
        trace.append([terms[k].item() for k in ("loss", "disc", "gen", "reg")])
        opt.step()
        if s == 0:
            p1 = np.stack([cases.summarize(p.detach().numpy()) for p in model.parameters()])
            assert np.allclose(p1, gold["param_after1"], rtol=1e-5, atol=1e-6)
    pl = np.stack([cases.summarize(p.detach().numpy()) for p in model.parameters()])
    assert np.allclose(pl, gold["param_after_last"], rtol=1e-5, atol=1e-6)
    assert np.allclose(np.array(trace), gold["trace"], rtol=1e-5, atol=1e-6)


def test_state_dict_keys_match_reference_listing():
    """78 tensors / 477,294 parameters for MFM_KL_EF at the canonical sizes
    (probe of the reference in the build container)."""
    cs = cases.load_case("klef_b32_t20")
    m = O.build("kl_ef", cs["cfgs"])
    sd = m.state_dict()
    assert len(sd) == 78
    assert sum(v.numel() for v in sd.values()) == 477294
    assert list(sd)[0] == "encoder_l.lstm.weight_ih" and list(sd)[-1] == "fy_to_y_fc2.bias"


def test_numpy_cell_gate_order():
    rs = np.random.RandomState(0)
    d, h, B = 7, 5, 3
    cell = torch.nn.LSTMCell(d, h)
    x = rs.normal(size=(B, d)).astype(np.float32)
    h0 = rs.normal(size=(B, h)).astype(np.float32)
    c0 = rs.normal(size=(B, h)).astype(np.float32)
    h1, c1 = cell(torch.from_numpy(x), (torch.from_numpy(h0), torch.from_numpy(c0)))
    h2, c2 = O.lstm_cell_numpy(x, h0, c0, cell.weight_ih.detach().numpy(), cell.weight_hh.detach().numpy(),
                               cell.bias_ih.detach().numpy(), cell.bias_hh.detach().numpy())
    assert np.allclose(h1.detach().numpy(), h2, atol=1e-6)
    assert np.allclose(c1.detach().numpy(), c2, atol=1e-6)


@pytest.mark.parametrize("variant,gname", [("kl_ef", "klef_staged_b32_t20"), ("kl", "kl_staged_b32_t20"),
                                           ("mmd", "mmd_staged_b32_t20")])
@pytest.mark.parametrize("mode", ["frozen", "legacy"])
def test_oracle_staged_training_matches_reference(mode, variant, gname):
    """train_beta_vae's two-stage schedule on the oracle vs the reference's own trajectory (run_staged in
    tests/golden/make_golden.py): pins oracle.stage_loss and both zero_grad semantics, for all three classes."""
    torch.set_num_threads(1)
    gold = np.load(cases.GOLDEN + "/%s.npz" % gname)
    B, T, n1, n2 = (int(v) for v in gold["meta"])
    from factorized_amd import configs
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    model = O.build(variant, cfgs)
    O.load_numpy_weights(model, synth.make_weights(O.state_shapes(model), seed=1234))
    model.train()
    if variant == "mmd":
        g = torch.from_numpy(np.ascontiguousarray(gold["mmd_gauss"]))
        model.mmd_gauss = list(torch.split(g, [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]], dim=1))
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    opt = torch.optim.Adam(model.parameters())
    if mode == "frozen":          # gradients of both stage losses at the initial weights; NaN rows = .grad is None
        for stage in (1, 2):
            opt.zero_grad(set_to_none=True)
            O.stage_loss(O.loss_terms(model, x, y, cfg), cfg, stage).backward()
            gs = gold["grad_summary_stage%d" % stage]
            for i, p in enumerate(model.parameters()):
                assert (p.grad is None) == bool(np.isnan(gs[i, 0])), (stage, i)
                if p.grad is not None:
                    assert np.allclose(cases.summarize(p.grad.numpy()), gs[i], rtol=2e-4, atol=2e-6), (stage, i)
    trace = []
    for s in range(n1 + n2):
        stage = 1 if s < n1 else 2
        opt.zero_grad(set_to_none=(mode == "frozen"))
        terms = O.loss_terms(model, x, y, cfg)
        loss = O.stage_loss(terms, cfg, stage)
        loss.backward()
        opt.step()
        trace.append([loss.item(), terms["disc"].item(), terms["gen"].item(), terms["reg"].item()])
        if s == n1 - 1:
            p1 = np.stack([cases.summarize(p.detach().numpy()) for p in model.parameters()])
            assert np.allclose(p1, gold[mode + "_param_after_stage1"], rtol=1e-5, atol=1e-6)
    p2 = np.stack([cases.summarize(p.detach().numpy()) for p in model.parameters()])
    assert np.allclose(p2, gold[mode + "_param_after_stage2"], rtol=1e-5, atol=1e-6)
    assert np.allclose(np.array(trace), gold[mode + "_trace"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", ["mosei", "you"])
def test_oracle_matches_reference_t50_forward(shape):
    """BASELINE configs 4 / 3 at the sequence length they name (MOSEI shape: 7 regression outputs; YouTube shape: cross-entropy
    head, D = 410; T=50, B=256; light goldens: summaries only): the oracle's loss terms of the first step against the
    reference's (one forward: the CPU suite stays within minutes)."""
    gold = np.load(cases.GOLDEN + "/klef_%s_b256_t50.npz" % shape)
    B, T, _ = (int(v) for v in gold["meta"])
    assert (B, T) == (256, 50)
    cfgs = (configs.mosei_configs if shape == "mosei" else configs.you_configs)(dropout=False)
    cfg = cfgs[0]
    model = O.build("kl_ef", cfgs)
    O.load_numpy_weights(model, synth.make_weights(O.state_shapes(model), seed=1234))
    model.train()
    lk = cfg.get("loss", "l1")
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7, output_dim=cfg["output_dim"], classes=cfg["output_dim"] if lk == "ce" else 0)
    torch.set_num_threads(4)
    with torch.no_grad():
        terms = O.loss_terms(model, torch.from_numpy(xn), torch.from_numpy(yn), cfg, lk)
    for k in ("disc", "gen", "gen_l", "gen_a", "gen_v", "reg", "loss"):
        ref = float(gold["fwd_" + k])
        assert abs(float(terms[k]) - ref) <= 2e-6 * max(abs(ref), 1.0), (k, float(terms[k]), ref)
    assert np.allclose(cases.summarize(terms["decoded"][3].numpy()), gold["y_hat_sum"], rtol=1e-5, atol=1e-5)
