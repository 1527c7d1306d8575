"""CPU ORACLE for the reference's ablation and missing-modality models -- TEST INFRASTRUCTURE, NOT PRODUCT CODE
(same rules as oracle/mfm_oracle.py: only tests/ may import it).

Plain PyTorch-CPU restatements, built from the oracle's SeqEncoder / SeqDecoder / MemFusion blocks, of

  M_A, M_B, M_C, M_D            reference mfm_model.py:201-467  (ablations of the factorization)
  MFM_missing                   reference mfm_model.py:766-898  (surrogate encoders for a missing modality)
  seq2seq, basic_missing        reference mfm_model.py:900-1017 (baselines of the missing-modality study)

Pinned against outputs of the reference itself: tests/golden/extra_*.npz (make_golden.py::run_extra).  The N(0,1)
samples loss_MMD draws (mfm_model.py:26) are injected in call order through `mmd_gauss`."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mfm_oracle import MemFusion, SeqDecoder, SeqEncoder, mmd


class _Base(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.d_l, self.d_a, self.d_v = config["input_dims"]
        self.mmd_gauss = None
        self._gi = 0

    def _split(self, x):
        return x[:, :, :self.d_l], x[:, :, self.d_l:self.d_l + self.d_a], x[:, :, self.d_l + self.d_a:]

    def _mmd(self, z):
        g = self.mmd_gauss[self._gi] if self.mmd_gauss is not None else torch.randn(z.size())
        self._gi += 1
        return mmd(z, g)

    def _zf(self, tag, cfg, zin, fout):
        setattr(self, tag + "_fc1", nn.Linear(zin, fout))
        setattr(self, tag + "_fc2", nn.Linear(fout, fout))
        setattr(self, tag + "_dropout", nn.Dropout(cfg[tag + "_dropout"]))

    def _z_to_f(self, tag, z):
        fc1, fc2, drop = getattr(self, tag + "_fc1"), getattr(self, tag + "_fc2"), getattr(self, tag + "_dropout")
        return F.relu(fc2(drop(F.relu(fc1(z)))))

    def _classify(self, f):
        return self.fy_to_y_fc2(self.fy_to_y_dropout(F.relu(self.fy_to_y_fc1(f))))


def _sizes(c):
    return (c["zy_size"], c["zl_size"], c["za_size"], c["zv_size"], c["fy_size"], c["fl_size"], c["fa_size"], c["fv_size"])


class M_A(_Base):                                   # mfm_model.py:201-266
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        last = sum(config["h_dims"]) + config["memsize"]
        self.encoder_l = SeqEncoder(self.d_l + self.d_a + self.d_v, zl)
        self.decoder_l = SeqDecoder(fy + fl, self.d_l)
        self.decoder_a = SeqDecoder(fy + fl, self.d_a)
        self.decoder_v = SeqDecoder(fy + fl, self.d_v)
        self.mfn_encoder = MemFusion(config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig)
        self.last_to_zy_fc1 = nn.Linear(last, zy)
        self._zf("zy_to_fy", config, zy, fy)
        self._zf("zl_to_fl", config, zl, fl)
        self.fy_to_y_fc1 = nn.Linear(fy, fy)
        self.fy_to_y_fc2 = nn.Linear(fy, config["output_dim"])
        self.fy_to_y_dropout = nn.Dropout(config["fy_to_y_dropout"])

    def forward(self, x):
        self._gi = 0
        t = x.shape[0]
        zl = self.encoder_l(x)
        zy = self.last_to_zy_fc1(self.mfn_encoder(x))
        reg = self._mmd(zl) + self._mmd(zy)
        fy, fl = self._z_to_f("zy_to_fy", zy), self._z_to_f("zl_to_fl", zl)
        h = torch.cat([fy, fl], 1)
        return [self.decoder_l(h, t), self.decoder_a(h, t), self.decoder_v(h, t), self._classify(fy)], reg, 0.0


class M_B(_Base):                                   # mfm_model.py:268-335
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        self.encoder_l = SeqEncoder(self.d_l, zl)
        self.encoder_a = SeqEncoder(self.d_a, za)
        self.encoder_v = SeqEncoder(self.d_v, zv)
        self.decoder_l = SeqDecoder(fl, self.d_l)
        self.decoder_a = SeqDecoder(fa, self.d_a)
        self.decoder_v = SeqDecoder(fv, self.d_v)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)
        self.fy_to_y_fc1 = nn.Linear(fl + fa + fv, fy)
        self.fy_to_y_fc2 = nn.Linear(fy, config["output_dim"])
        self.fy_to_y_dropout = nn.Dropout(config["fy_to_y_dropout"])

    def forward(self, x):
        self._gi = 0
        t = x.shape[0]
        x_l, x_a, x_v = self._split(x)
        zl, za, zv = self.encoder_l(x_l), self.encoder_a(x_a), self.encoder_v(x_v)
        reg = self._mmd(zl) + self._mmd(za) + self._mmd(zv)
        fl, fa, fv = self._z_to_f("zl_to_fl", zl), self._z_to_f("za_to_fa", za), self._z_to_f("zv_to_fv", zv)
        y_hat = self._classify(torch.cat([fl, fa, fv], 1))
        return [self.decoder_l(fl, t), self.decoder_a(fa, t), self.decoder_v(fv, t), y_hat], reg, 0.0


class M_C(_Base):                                   # mfm_model.py:337-387
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        last = sum(config["h_dims"]) + config["memsize"]
        self.decoder_l = SeqDecoder(fy, self.d_l)
        self.decoder_a = SeqDecoder(fy, self.d_a)
        self.decoder_v = SeqDecoder(fy, self.d_v)
        self.mfn_encoder = MemFusion(config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig)
        self.last_to_zy_fc1 = nn.Linear(last, zy)
        self._zf("zy_to_fy", config, zy, fy)
        self.fy_to_y_fc1 = nn.Linear(fy, fy)
        self.fy_to_y_fc2 = nn.Linear(fy, config["output_dim"])
        self.fy_to_y_dropout = nn.Dropout(config["fy_to_y_dropout"])

    def forward(self, x):
        self._gi = 0
        t = x.shape[0]
        zy = self.last_to_zy_fc1(self.mfn_encoder(x))
        reg = self._mmd(zy)
        fy = self._z_to_f("zy_to_fy", zy)
        return [self.decoder_l(fy, t), self.decoder_a(fy, t), self.decoder_v(fy, t), self._classify(fy)], reg, 0.0


class M_D(_Base):                                   # mfm_model.py:389-467
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        self.encoder_l = SeqEncoder(self.d_l, zl)
        self.encoder_a = SeqEncoder(self.d_a, za)
        self.encoder_v = SeqEncoder(self.d_v, zv)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)
        self.fs_to_y = nn.Linear(fl + fa + fv, config["output_dim"])

    def forward(self, x):
        x_l, x_a, x_v = self._split(x)
        zl, za, zv = self.encoder_l(x_l), self.encoder_a(x_a), self.encoder_v(x_v)
        fl, fa, fv = self._z_to_f("zl_to_fl", zl), self._z_to_f("za_to_fa", za), self._z_to_f("zv_to_fv", zv)
        return [x_l, x_a, x_v, self.fs_to_y(torch.cat([fl, fa, fv], 1))], 0.0, 0.0


class MFM_missing(_Base):                           # mfm_model.py:766-898
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        d_l, d_a, d_v = self.d_l, self.d_a, self.d_v
        last = sum(config["h_dims"]) + config["memsize"]
        self.encoder_l = SeqEncoder(d_l, zl)
        self.encoder_a = SeqEncoder(d_a, za)
        self.encoder_v = SeqEncoder(d_v, zv)
        self.encoder_la_to_v = SeqEncoder(d_l + d_a, zv)
        self.encoder_lv_to_a = SeqEncoder(d_l + d_v, za)
        self.encoder_av_to_l = SeqEncoder(d_a + d_v, zl)
        self.encoder_la_to_y = SeqEncoder(d_l + d_a, zy)
        self.encoder_lv_to_y = SeqEncoder(d_l + d_v, zy)
        self.encoder_av_to_y = SeqEncoder(d_a + d_v, zy)
        self.decoder_l = SeqDecoder(fy + fl, d_l)
        self.decoder_a = SeqDecoder(fy + fa, d_a)
        self.decoder_v = SeqDecoder(fy + fv, d_v)
        self.mfn_encoder = MemFusion(config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig)
        self.last_to_zy_fc1 = nn.Linear(last, zy)
        self._zf("zy_to_fy", config, zy, fy)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)
        self.fy_to_y_fc1 = nn.Linear(fy, fy)
        self.fy_to_y_fc2 = nn.Linear(fy, config["output_dim"])
        self.fy_to_y_dropout = nn.Dropout(config["fy_to_y_dropout"])

    def forward(self, x):
        self._gi = 0
        t = x.shape[0]
        x_l, x_a, x_v = self._split(x)
        zl, za, zv = self.encoder_l(x_l), self.encoder_a(x_a), self.encoder_v(x_v)
        zy = self.last_to_zy_fc1(self.mfn_encoder(x))
        la, lv, av = torch.cat([x_l, x_a], 2), torch.cat([x_l, x_v], 2), torch.cat([x_a, x_v], 2)
        zv_nov, za_noa, zl_nol = self.encoder_la_to_v(la), self.encoder_lv_to_a(lv), self.encoder_av_to_l(av)
        zy_nov, zy_noa, zy_nol = self.encoder_la_to_y(la), self.encoder_lv_to_y(lv), self.encoder_av_to_y(av)
        reg = self._mmd(zl) + self._mmd(za) + self._mmd(zv) + self._mmd(zy)
        missing = F.mse_loss(zv_nov, zv) + F.mse_loss(za_noa, za) + F.mse_loss(zl_nol, zl) \
            + F.mse_loss(zy_nov, zy) + F.mse_loss(zy_noa, zy) + F.mse_loss(zy_nol, zy)

        def decode(zl, za, zv, zy):
            fy = self._z_to_f("zy_to_fy", zy)
            fl, fa, fv = self._z_to_f("zl_to_fl", zl), self._z_to_f("za_to_fa", za), self._z_to_f("zv_to_fv", zv)
            return [self.decoder_l(torch.cat([fy, fl], 1), t), self.decoder_a(torch.cat([fy, fa], 1), t),
                    self.decoder_v(torch.cat([fy, fv], 1), t), self._classify(fy)]
        return (decode(zl, za, zv, zy), decode(zl_nol, za, zv, zy_nol), decode(zl, za_noa, zv, zy_noa),
                decode(zl, za, zv_nov, zy_nov), reg, missing)


class seq2seq(_Base):                               # mfm_model.py:900-960
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        d_l, d_a, d_v = self.d_l, self.d_a, self.d_v
        self.encoder_la_to_v = SeqEncoder(d_l + d_a, zv)
        self.encoder_lv_to_a = SeqEncoder(d_l + d_v, za)
        self.encoder_av_to_l = SeqEncoder(d_a + d_v, zl)
        self.decoder_l = SeqDecoder(fl, d_l)
        self.decoder_a = SeqDecoder(fa, d_a)
        self.decoder_v = SeqDecoder(fv, d_v)
        self._zf("zl_to_fl", config, zl, fl)
        self._zf("za_to_fa", config, za, fa)
        self._zf("zv_to_fv", config, zv, fv)

    def forward(self, x):
        self._gi = 0
        t = x.shape[0]
        x_l, x_a, x_v = self._split(x)
        zv_nov = self.encoder_la_to_v(torch.cat([x_l, x_a], 2))
        za_noa = self.encoder_lv_to_a(torch.cat([x_l, x_v], 2))
        zl_nol = self.encoder_av_to_l(torch.cat([x_a, x_v], 2))
        reg = self._mmd(zv_nov) + self._mmd(za_noa) + self._mmd(zl_nol)
        fl, fa, fv = self._z_to_f("zl_to_fl", zl_nol), self._z_to_f("za_to_fa", za_noa), self._z_to_f("zv_to_fv", zv_nov)
        return [self.decoder_l(fl, t)], [self.decoder_a(fa, t)], [self.decoder_v(fv, t)], reg


class basic_missing(_Base):                         # mfm_model.py:962-1017
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super().__init__(config)
        zy, zl, za, zv, fy, fl, fa, fv = _sizes(config)
        d_l, d_a, d_v = self.d_l, self.d_a, self.d_v
        self.encoder_la_to_y = SeqEncoder(d_l + d_a, zy)
        self.encoder_lv_to_y = SeqEncoder(d_l + d_v, zy)
        self.encoder_av_to_y = SeqEncoder(d_a + d_v, zy)
        for tag in ("zy_nol_to_y", "zy_noa_to_y", "zy_nov_to_y"):
            setattr(self, tag + "_fc1", nn.Linear(zy, fy))
            setattr(self, tag + "_fc2", nn.Linear(fy, config["output_dim"]))
            setattr(self, tag + "_dropout", nn.Dropout(config["zy_to_fy_dropout"]))

    def _head(self, tag, z):
        return getattr(self, tag + "_fc2")(getattr(self, tag + "_dropout")(F.relu(getattr(self, tag + "_fc1")(z))))

    def forward(self, x):
        self._gi = 0
        x_l, x_a, x_v = self._split(x)
        zy_nov = self.encoder_la_to_y(torch.cat([x_l, x_a], 2))
        zy_noa = self.encoder_lv_to_y(torch.cat([x_l, x_v], 2))
        zy_nol = self.encoder_av_to_y(torch.cat([x_a, x_v], 2))
        reg = self._mmd(zy_nov) + self._mmd(zy_noa) + self._mmd(zy_nol)
        return self._head("zy_nol_to_y", zy_nol), self._head("zy_noa_to_y", zy_noa), self._head("zy_nov_to_y", zy_nov), reg


CLASSES = {"M_A": M_A, "M_B": M_B, "M_C": M_C, "M_D": M_D, "MFM_missing": MFM_missing, "seq2seq": seq2seq,
           "basic_missing": basic_missing}
N_GAUSS = {"M_A": ["zl", "zy"], "M_B": ["zl", "za", "zv"], "M_C": ["zy"], "M_D": [], "MFM_missing": ["zl", "za", "zv", "zy"],
           "seq2seq": ["zv", "za", "zl"], "basic_missing": ["zy", "zy", "zy"]}


def flatten_outputs(out):
    """every tensor a forward returns, in a fixed order (lists are walked), floats as 0-d tensors"""
    flat = []

    def walk(o):
        if isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
        elif torch.is_tensor(o):
            flat.append(o)
        else:
            flat.append(torch.tensor(float(o)))
    walk(out)
    return flat


def test_objective(out):
    """a scalar that depends on EVERY output of a forward (used to compare gradients): sum over outputs of
    mean(o^2) * (1 + 0.1 * index) for tensors with a graph, plus the scalar regularisers"""
    total = 0.0
    for i, o in enumerate(flatten_outputs(out)):
        if o.requires_grad:
            total = total + (o * o).mean() * (1.0 + 0.1 * i) if o.dim() else total + o * (1.0 + 0.1 * i)
    return total
