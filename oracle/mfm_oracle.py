"""CPU ORACLE for the MFM training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file.  `factorized_amd/` never does: the product path is the HIP
library and fails loudly without it.

What this is: a plain PyTorch-CPU restatement of the reference's hot path
(pliang279/factorized, `mfm_model.py` + the joint-loss step of `mfm_mosi.py`),
written from the reference's behaviour with the same op sequence -- a Python
`for` loop over timesteps around `nn.LSTMCell`, `nn.Linear` heads, sum-KLD,
mean-MSE / mean-L1, `optim.Adam` -- so it doubles as the honest "reference CPU
path" timing in bench.py (`cpu_baseline.kind == "port"`).

Parity pin: the reference ships no tests or golden vectors,
so the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF: tests/golden/
make_golden.py imports /root/reference/mfm_model.py in the build container, runs
it on the seeded recipe of factorized_amd/synth.py and commits the results as
tests/golden/*.npz; tests/test_oracle_golden.py checks this file against them.

Reference citations (file:line into /root/reference):
  loss_KLD            mfm_model.py:36-38      (sum, not mean)
  compute_kernel/MMD  mfm_model.py:14-34      (note the double division by dim)
  encoderLSTM         mfm_model.py:40-62      (returns fc1(h_T), no activation)
  decoderLSTM         mfm_model.py:64-91      (step>0 input is its own hidden state)
  MFN                 mfm_model.py:93-199
  MFM / MFM_KL_EF / MFM_KL forward   mfm_model.py:522-555 / 619-660 / 723-764
  joint loss + step   mfm_mosi.py:424-442     (train_mfm.train)
  staged loss         mfm_mosi.py:255-285     (train_beta_vae.train)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- losses
def kld_sum(mu, logvar):
    # mfm_model.py:36-38
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())


def gauss_kernel(a, b):
    # mfm_model.py:14-23: exp(-mean_k((a_i-b_j)^2)/k)   (mean AND a further /k)
    k = a.size(1)
    diff = a.unsqueeze(1) - b.unsqueeze(0)
    return torch.exp(-(diff.pow(2).mean(2) / float(k)))


def mmd(z, gauss):
    # mfm_model.py:25-34 with the N(0,1) sample injected (the reference draws it
    # with torch.randn on the host; parity needs the same numbers on both sides).
    return gauss_kernel(gauss, gauss).mean() + gauss_kernel(z, z).mean() \
        - 2.0 * gauss_kernel(gauss, z).mean()


# ----------------------------------------------------------------------------- blocks
class SeqEncoder(nn.Module):
    """mfm_model.py:40-62."""

    def __init__(self, d, h):
        super().__init__()
        self.lstm = nn.LSTMCell(d, h)
        self.fc1 = nn.Linear(h, h)
        self.h = h

    def run(self, x, keep=False):
        T, B = x.shape[0], x.shape[1]
        hx = x.new_zeros(B, self.h)
        cx = x.new_zeros(B, self.h)
        hs = []
        for t in range(T):
            hx, cx = self.lstm(x[t], (hx, cx))
            if keep:
                hs.append(hx)
        return (self.fc1(hx), hs) if keep else self.fc1(hx)

    def forward(self, x):
        return self.run(x)


class SeqDecoder(nn.Module):
    """mfm_model.py:64-91: LSTMCell(h,h); step 0 consumes the embedding, every
    later step consumes the previous hidden state as its input."""

    def __init__(self, h, d):
        super().__init__()
        self.lstm = nn.LSTMCell(h, h)
        self.fc1 = nn.Linear(h, d)
        self.h, self.d = h, d

    def forward(self, hT, t):
        B = hT.shape[0]
        hx = hT.new_zeros(B, self.h)
        cx = hT.new_zeros(B, self.h)
        inp = hT
        hs = []
        for _ in range(t):
            hx, cx = self.lstm(inp, (hx, cx))
            inp = hx
            hs.append(hx)
        return self.fc1(torch.stack(hs, 0))


class MemFusion(nn.Module):
    """Memory Fusion Network encoder, mfm_model.py:93-199.  `out_fc1/out_fc2`
    exist (state_dict keys) but are unused in forward, as in the reference."""

    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__()
        self.d_l, self.d_a, self.d_v = config["input_dims"]
        self.dh_l, self.dh_a, self.dh_v = config["h_dims"]
        tot = self.dh_l + self.dh_a + self.dh_v
        self.mem_dim = config["memsize"]
        att_in = tot * config["windowsize"]
        gam_in = att_in + self.mem_dim
        self.lstm_l = nn.LSTMCell(self.d_l, self.dh_l)
        self.lstm_a = nn.LSTMCell(self.d_a, self.dh_a)
        self.lstm_v = nn.LSTMCell(self.d_v, self.dh_v)
        self.att1_fc1 = nn.Linear(att_in, NN1Config["shapes"])
        self.att1_fc2 = nn.Linear(NN1Config["shapes"], att_in)
        self.att1_dropout = nn.Dropout(NN1Config["drop"])
        self.att2_fc1 = nn.Linear(att_in, NN2Config["shapes"])
        self.att2_fc2 = nn.Linear(NN2Config["shapes"], self.mem_dim)
        self.att2_dropout = nn.Dropout(NN2Config["drop"])
        self.gamma1_fc1 = nn.Linear(gam_in, gamma1Config["shapes"])
        self.gamma1_fc2 = nn.Linear(gamma1Config["shapes"], self.mem_dim)
        self.gamma1_dropout = nn.Dropout(gamma1Config["drop"])
        self.gamma2_fc1 = nn.Linear(gam_in, gamma2Config["shapes"])
        self.gamma2_fc2 = nn.Linear(gamma2Config["shapes"], self.mem_dim)
        self.gamma2_dropout = nn.Dropout(gamma2Config["drop"])
        self.out_fc1 = nn.Linear(tot + self.mem_dim, outConfig["shapes"])
        self.out_fc2 = nn.Linear(outConfig["shapes"], config["output_dim"])
        self.out_dropout = nn.Dropout(outConfig["drop"])

    def forward(self, x):
        T, B = x.shape[0], x.shape[1]
        xs = torch.split(x, [self.d_l, self.d_a, self.d_v], dim=2)
        cells = (self.lstm_l, self.lstm_a, self.lstm_v)
        hs = [x.new_zeros(B, n) for n in (self.dh_l, self.dh_a, self.dh_v)]
        cs = [x.new_zeros(B, n) for n in (self.dh_l, self.dh_a, self.dh_v)]
        mem = x.new_zeros(B, self.mem_dim)
        for t in range(T):
            prev = torch.cat(cs, 1)
            new = [cell(xm[t], (h, c)) for cell, xm, h, c in zip(cells, xs, hs, cs)]
            hs = [n[0] for n in new]
            cs = [n[1] for n in new]
            c_star = torch.cat([prev, torch.cat(cs, 1)], 1)
            att = F.softmax(self.att1_fc2(self.att1_dropout(F.relu(self.att1_fc1(c_star)))), dim=1)
            attended = att * c_star
            c_hat = torch.tanh(self.att2_fc2(self.att2_dropout(F.relu(self.att2_fc1(attended)))))
            both = torch.cat([attended, mem], 1)
            g1 = torch.sigmoid(self.gamma1_fc2(self.gamma1_dropout(F.relu(self.gamma1_fc1(both)))))
            g2 = torch.sigmoid(self.gamma2_fc2(self.gamma2_dropout(F.relu(self.gamma2_fc1(both)))))
            mem = g1 * mem + g2 * c_hat
        return torch.cat(hs + [mem], 1)


# ----------------------------------------------------------------------------- models
class _Factorized(nn.Module):
    """Shared wiring of MFM / MFM_KL / MFM_KL_EF (mfm_model.py:469-764).

    variant: 'kl_ef' (early-fusion LSTM for zy + KLD), 'kl' (MFN for zy + KLD),
             'mmd'   (MFN for zy, no logvar heads, MMD regulariser)."""

    def __init__(self, variant, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__()
        self.variant = variant
        self.d_l, self.d_a, self.d_v = config["input_dims"]
        zy, zl, za, zv = (config[k] for k in ("zy_size", "zl_size", "za_size", "zv_size"))
        fy, fl, fa, fv = (config[k] for k in ("fy_size", "fl_size", "fa_size", "fv_size"))
        self.encoder_l = SeqEncoder(self.d_l, zl)
        self.encoder_a = SeqEncoder(self.d_a, za)
        self.encoder_v = SeqEncoder(self.d_v, zv)
        self.decoder_l = SeqDecoder(fy + fl, self.d_l)
        self.decoder_a = SeqDecoder(fy + fa, self.d_a)
        self.decoder_v = SeqDecoder(fy + fv, self.d_v)
        if variant == "kl_ef":
            last = zl + za + zv
            self.ef_encoder = SeqEncoder(self.d_l + self.d_a + self.d_v, last)
        else:
            last = sum(config["h_dims"]) + config["memsize"]
            self.mfn_encoder = MemFusion(config, NN1Config, NN2Config, gamma1Config,
                                         gamma2Config, outConfig)
        self.last_to_zy_fc1 = nn.Linear(last, zy)
        if variant != "mmd":
            self.last_to_logvarzy_fc1 = nn.Linear(last, zy)
            self.last_to_zl_fc1 = nn.Linear(zl, zl)
            self.last_to_za_fc1 = nn.Linear(za, za)
            self.last_to_zv_fc1 = nn.Linear(zv, zv)
            self.last_to_logvarzl_fc1 = nn.Linear(zl, zl)
            self.last_to_logvarza_fc1 = nn.Linear(za, za)
            self.last_to_logvarzv_fc1 = nn.Linear(zv, zv)
        for tag, zin, fout in (("zy_to_fy", zy, fy), ("zl_to_fl", zl, fl),
                               ("za_to_fa", za, fa), ("zv_to_fv", zv, fv)):
            setattr(self, tag + "_fc1", nn.Linear(zin, fout))
            setattr(self, tag + "_fc2", nn.Linear(fout, fout))
            setattr(self, tag + "_dropout", nn.Dropout(config[tag + "_dropout"]))
        self.fy_to_y_fc1 = nn.Linear(fy, fy)
        self.fy_to_y_fc2 = nn.Linear(fy, config["output_dim"])
        self.fy_to_y_dropout = nn.Dropout(config["fy_to_y_dropout"])
        self.mmd_gauss = None  # optional injected N(0,1) samples [zl, za, zv, zy]

    def _z_to_f(self, tag, z):
        fc1, fc2 = getattr(self, tag + "_fc1"), getattr(self, tag + "_fc2")
        drop = getattr(self, tag + "_dropout")
        return F.relu(fc2(drop(F.relu(fc1(z)))))

    def forward(self, x):
        T = x.shape[0]
        x_l = x[:, :, :self.d_l]
        x_a = x[:, :, self.d_l:self.d_l + self.d_a]
        x_v = x[:, :, self.d_l + self.d_a:]
        l_last = self.encoder_l(x_l)
        a_last = self.encoder_a(x_a)
        v_last = self.encoder_v(x_v)
        y_last = self.ef_encoder(x) if self.variant == "kl_ef" else self.mfn_encoder(x)
        zy = self.last_to_zy_fc1(y_last)
        if self.variant == "mmd":
            zl, za, zv = l_last, a_last, v_last
            g = self.mmd_gauss
            if g is None:
                g = [torch.randn(z.size()) for z in (zl, za, zv, zy)]
            reg = mmd(zl, g[0]) + mmd(za, g[1]) + mmd(zv, g[2]) + mmd(zy, g[3])
        else:
            zl = self.last_to_zl_fc1(l_last)
            za = self.last_to_za_fc1(a_last)
            zv = self.last_to_zv_fc1(v_last)
            reg = kld_sum(zl, self.last_to_logvarzl_fc1(l_last)) \
                + kld_sum(za, self.last_to_logvarza_fc1(a_last)) \
                + kld_sum(zv, self.last_to_logvarzv_fc1(v_last)) \
                + kld_sum(zy, self.last_to_logvarzy_fc1(y_last))
        fy = self._z_to_f("zy_to_fy", zy)
        fl = self._z_to_f("zl_to_fl", zl)
        fa = self._z_to_f("za_to_fa", za)
        fv = self._z_to_f("zv_to_fv", zv)
        x_l_hat = self.decoder_l(torch.cat([fy, fl], 1), T)
        x_a_hat = self.decoder_a(torch.cat([fy, fa], 1), T)
        x_v_hat = self.decoder_v(torch.cat([fy, fv], 1), T)
        y_hat = self.fy_to_y_fc2(self.fy_to_y_dropout(F.relu(self.fy_to_y_fc1(fy))))
        return [x_l_hat, x_a_hat, x_v_hat, y_hat], reg, 0.0


def build(variant, configs):
    """variant in {'kl_ef','kl','mmd'} <-> reference MFM_KL_EF / MFM_KL / MFM."""
    return _Factorized(variant, *configs)


def load_numpy_weights(model, weights):
    sd = model.state_dict()
    assert list(sd.keys()) == list(weights.keys()), "state_dict key mismatch"
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()})
    return model


def state_shapes(model):
    return {k: tuple(v.shape) for k, v in model.state_dict().items()}


# ----------------------------------------------------------------------------- the step
def loss_terms(model, x, y, cfg, loss_kind="l1"):
    """mfm_mosi.py:430-439.  Returns dict of scalar tensors + the decoded list."""
    d_l, d_a, _ = cfg["input_dims"]
    decoded, reg, missing = model.forward(x)
    x_l_hat, x_a_hat, x_v_hat, y_hat = decoded
    x_l = x[:, :, :d_l]
    x_a = x[:, :, d_l:d_l + d_a]
    x_v = x[:, :, d_l + d_a:]
    gen_l = F.mse_loss(x_l_hat, x_l)
    gen_a = F.mse_loss(x_a_hat, x_a)
    gen_v = F.mse_loss(x_v_hat, x_v)
    gen = cfg["lda_xl"] * gen_l + cfg["lda_xa"] * gen_a + cfg["lda_xv"] * gen_v
    if loss_kind == "ce":
        disc = F.cross_entropy(y_hat, y)                       # mfm_you.py:451,484
    elif y_hat.shape[1] == 1:
        disc = F.l1_loss(y_hat.squeeze(1), y)                  # mfm_mosi.py:432,438
    else:
        disc = F.l1_loss(y_hat, y)
    reg_w = cfg["lda_mmd"] * reg
    return dict(disc=disc, gen=gen, gen_l=gen_l, gen_a=gen_a, gen_v=gen_v, reg=reg,
                loss=disc + gen + reg_w + missing, decoded=decoded)


def stage_loss(terms, cfg, stage):
    """train_beta_vae (mfm_mosi.py:278-281): stage 1 = gen + reg, stage 2 = disc + reg;
    stage 0 = the joint loss of train_mfm (mfm_mosi.py:439)."""
    reg_w = cfg["lda_mmd"] * terms["reg"]
    if stage == 1:
        return terms["gen"] + reg_w
    if stage == 2:
        return terms["disc"] + reg_w
    return terms["loss"]


def train_step(model, optimizer, x, y, cfg, stage=0, loss_kind="l1"):
    optimizer.zero_grad()
    terms = loss_terms(model, x, y, cfg, loss_kind)
    loss = stage_loss(terms, cfg, stage)
    loss.backward()
    optimizer.step()
    return terms, loss


def time_cpu_steps(configs, B, T, budget_s=10.0, max_steps=200, warmup=2, threads=None, variant="kl_ef"):
    """Reference-CPU-path timing for bench.py's cpu_baseline leg: train mode, joint loss, Adam
    defaults -- the op sequence of mfm_mosi.py:424-442.  Bounded by wall time (`budget_s`)."""
    import time
    from factorized_amd import synth  # data recipe only (numpy)
    if threads:
        torch.set_num_threads(int(threads))
    cfg = configs[0]
    model = build(variant, configs)
    opt = torch.optim.Adam(model.parameters())
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7, output_dim=cfg["output_dim"])
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    model.train()
    for _ in range(warmup):
        train_step(model, opt, x, y, cfg)
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps:
        train_step(model, opt, x, y, cfg)
        steps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(ms_per_step=1e3 * dt / steps, samples_per_s=B * steps / dt,
                threads=torch.get_num_threads(), steps=steps)


# ----------------------------------------------------------------------------- scalar math
def lstm_cell_numpy(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """Explicit fp64 LSTM cell (torch gate order i,f,g,o) used to pin the gate
    layout the HIP kernels assume.  x[B,d] h,c[B,H]."""
    import numpy as np
    g = x.astype(np.float64) @ w_ih.T.astype(np.float64) + b_ih + h @ w_hh.T.astype(np.float64) + b_hh
    H = h.shape[1]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    i, f, gg, o = sig(g[:, :H]), sig(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), sig(g[:, 3 * H:])
    c2 = f * c + i * gg
    return o * np.tanh(c2), c2
